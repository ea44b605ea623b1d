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
    # the stated core count must describe compute actually obtained: N workers deliver at least half of N single workers
    assert c["cores"] <= c["affinity"] and c["single_worker"] > 0 and c["cores_effective"] > 0.5 * c["cores"], c
    assert d["nan_envs"] == 0 and d["per_rank"]["nan_envs"] == [0]


@pytest.mark.gpu
def test_bench_line_3d_small_batch():
    d = _run("--dim", "3", "--steps", "2", "--warmup", "1", "--batch", "4", "--no-cpu-baseline")
    assert REQUIRED <= set(d) and d["nan_envs"] == 0 and d["value"] > 0
    assert abs(d["roofline"]["algorithmic_bytes_per_launch"] - 4 * 13 * 10 * 4 * 73728 * 8) < 1


def test_gpus_without_launcher_never_prints_a_one_gpu_line():
    """`python bench.py --gpus 8` without torchrun must either start its own 8 ranks or fail: it may never print an
    n_gpus=1 line with rc 0.  In this container there is no GPU, so the self-launcher refuses (fewer devices than asked)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=300, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    import torch
    if torch.cuda.device_count() < 8:
        assert out.returncode != 0 and not lines, (out.returncode, out.stdout[-500:])
        assert "--gpus 8" in out.stderr
    else:                                              # a full node: the self-launched ranks print the 8-GPU line
        assert out.returncode == 0 and len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 8


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_self_launched_two_rank_rehearsal_on_one_gpu():
    """`--gpus 2` with no launcher: bench.py starts its own two ranks (gloo rehearsal: both share the one GPU of a test
    box) and rank 0 prints one n_gpus=2 line whose value counts the envs of both shards."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2",
                          "--warmup", "1", "--batch", "32"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 64 and d["nan_envs"] == 0
    # the CPU baseline is measured by the self-launching parent, before the ranks exist, and handed to rank 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["single_worker"] > 0
    # a straggler GPU must be visible: per-rank timings and NaN counts next to the max / sum
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == 2 and pr["nan_envs"] == [0, 0] and abs(max(pr["ms_per_step"]) - d["ms_per_step"]) < 1e-9
    assert abs(d["value"] - 64 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_self_launched_four_rank_rehearsal_on_one_gpu():
    """SCALE readiness without a node: `--gpus 4` with no launcher -- the self-launch is decided before any GPU call, four ranks
    (this pool lets six processes touch one card, the test runner is one of them, and one slot is left as margin; the eight-rank
    reductions run on the CPU in tests/test_sharding_gloo.py::test_eight_rank_reductions_of_the_scale_line) create four handles on the
    one device, step their shards between gloo barriers, and rank 0 prints ONE line: n_gpus 4, global batch 4 x 128, `per_rank` of
    length four, the CPU baseline taken by the parent before the ranks existed."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dist-backend", "gloo", "--steps", "3",
                          "--warmup", "1", "--batch", "128"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["config"]["global_batch"] == 512 and d["nan_envs"] == 0 and d["scaling"] == "weak"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    pr = d["per_rank"]
    assert len(pr["ms_per_step"]) == 4 and pr["nan_envs"] == [0] * 4 and abs(max(pr["ms_per_step"]) - d["ms_per_step"]) < 1e-9
    assert abs(d["value"] - 512 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert "extra" not in d                                   # the extras belong to the N = 1 line


@pytest.mark.gpu
def test_extra_keys_are_measured_in_the_same_run():
    d = _run("--steps", "2", "--warmup", "1", "--batch", "64", "--no-cpu-baseline")
    x = d["extra"]
    assert "error" not in x, x
    assert x["copy_ceiling"]["kernel_gbs"] > 500 and x["copy_ceiling"]["memcpy_d2d_gbs"] > 500
    assert x["gym_api_env_steps_per_s"]["no_info_state"] > 0 and x["gym_api_env_steps_per_s"]["default_info_state"] > 0 and x["gym_api_env_steps_per_s"]["always_new_arrays"] > 0
    c4 = x["config4_3d"]
    assert c4["nan_envs"] == 0 and c4["value"] > 0 and "32x48x48" in c4["config"]["workload"]
    assert d["roofline"]["on_box_copy_gbs"] == x["copy_ceiling"]["kernel_gbs"]
    rs = x["ra_sweep_config3"]                              # BASELINE.json configs[3] per GPU: thirds at Ra 1e4 / 1e5 / 1e6
    assert set(rs["nan_envs_per_ra"]) == {"10000", "100000", "1e+06"} and sum(rs["nan_envs_per_ra"].values()) == 0 and rs["value"] > 0
    assert 2.0 < rs["mean_nusselt_state_per_ra"]["10000"] < 6.5 and rs["mean_nusselt_state_per_ra"]["1e+06"] > rs["mean_nusselt_state_per_ra"]["10000"]
    sc = x["steady_ckpt"]                                   # the reference's stored steady states stay put under zero action
    assert sc["nan_envs"] == 0 and abs(sc["kinetic_energy"] / sc["reference_kinetic_energy"] - 1) < 1e-4 and sc["max_rel_ke_drift_per_env_step"] < 1e-4
    s2 = x["streaming_2d_128x64"]                           # a grid without an LDS-resident kernel: the streaming 2D path
    assert s2["nan_envs"] == 0 and s2["value"] > 0 and s2["steps"] >= 10 and 0.0 < s2["mean_nusselt_state"] < 12.0


def test_cpu_baseline_reports_the_compute_it_actually_got():
    """The CPU-baseline leg on THIS machine (no GPU needed): worker count bounded by affinity, cgroup quota and physical cores;
    the N-worker total within a factor of two of N single workers; the per-core rate is what the bench line's reader needs."""
    sys.path.insert(0, ROOT)
    import bench
    c = bench.cpu_baseline(dim=2, envs=8, budget_s=3.0)
    assert c["kind"] == "port" and c["unit"] == "env-steps/s" and 1 <= c["cores"] <= c["affinity"] and c["cores"] <= c["physical_cores"]
    if c["cgroup_cpu_quota"] is not None:
        assert c["cores"] <= int(c["cgroup_cpu_quota"] + 0.5) or c["cores"] == 1
    assert c["single_worker"] > 0 and abs(c["per_core"] * c["cores"] - c["value"]) < 1e-9 * c["value"]
    assert 0.5 * c["cores"] < c["cores_effective"] < 2.0 * c["cores"], c
    q = bench._cpu_quota()
    assert q is None or q > 0
