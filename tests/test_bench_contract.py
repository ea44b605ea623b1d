"""bench.py prints ONE JSON line with the driver's contract keys (metric/value/unit/..., roofline, cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"}


def _run(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_2d_small_batch():
    d = _run("--steps", "2", "--warmup", "1", "--batch", "64")
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "env-steps/s" and d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 64 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["algorithmic_bytes_per_launch"] - 64 * 50 * 10 * 3 * 6144 * 8) < 1
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and c["unit"] == "env-steps/s"
    assert d["nan_envs"] == 0


@pytest.mark.gpu
def test_bench_line_3d_small_batch():
    d = _run("--dim", "3", "--steps", "2", "--warmup", "1", "--batch", "4", "--no-cpu-baseline")
    assert REQUIRED <= set(d) and d["nan_envs"] == 0 and d["value"] > 0
    assert abs(d["roofline"]["algorithmic_bytes_per_launch"] - 4 * 13 * 10 * 4 * 73728 * 8) < 1
