"""GPU: randomised HIP <-> oracle parity over configurations nobody picked by hand (scripts/fuzz_parity.py).

Every grid the C ABI accepts (2D: >= 8 x 8; 3D: >= 8 cells per direction -- odd and prime sizes included) must give the oracle's answer
through whichever kernels the host picks for it, for random Rayleigh / Prandtl numbers, domains, plate temperatures, heater counts,
sensor grids, solver / control steps (ragged last substep), either clock, and batch sizes on both sides of the host's thresholds (env
groups, chain counts, tile shapes: envs beyond the first two or three repeat them and must reproduce them bit for bit).  The sweep is
seeded, so a failure names its configuration.
(First run of the sweep, round 4: 260 float64 + 195 float32 draws; the five draws outside the bars were all odd nx and the error was
the ORACLE's -- tests/test_oracle_golden.py::test_projection_is_exact_on_any_grid_odd_sizes_included.)"""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzz():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "scripts", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.oracle_py.build_oracle()
    return mod


@pytest.mark.parametrize("precision,seed,n2,n3", [("f64", 11, 36, 10), ("f32", 12, 24, 6)])
def test_random_configurations_match_the_oracle(fuzz, precision, seed, n2, n3):
    rng = np.random.default_rng(seed)
    bar_f, bar_nu, bar_obs = fuzz.BARS[precision]
    odd = 0
    for n in range(n2):
        cfg, obs, clock = fuzz.draw_2d(rng)
        odd += cfg["nx"] % 2
        B = fuzz.draw_batch(rng, cfg)
        w, wn, wo = fuzz.run_2d(cfg, obs, clock, seed * 1000 + n, precision, B)
        assert w < bar_f and wn < bar_nu and wo < bar_obs, (cfg, obs, clock, B, w, wn, wo)
    for n in range(n3):
        cfg, clock = fuzz.draw_3d(rng)
        B = fuzz.draw_batch(rng, cfg)
        w, wn = fuzz.run_3d(cfg, clock, seed * 1000 + 500 + n, precision, B)
        assert w < bar_f and wn < bar_nu, (cfg, clock, B, w, wn)
    assert odd >= 1                                                # the draw that found the oracle's even-nx assumption stays in the sweep


@pytest.fixture(scope="module")
def fuzz_seq():
    spec = importlib.util.spec_from_file_location("fuzz_sequences", os.path.join(ROOT, "scripts", "fuzz_sequences.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.oracle_py.build_oracle()
    return mod


@pytest.mark.parametrize("target", ["resident-2d", "streaming-2d", "3d"])
def test_random_call_sequences_match_per_env_oracle_mirrors(fuzz_seq, target):
    """per-env Rayleigh numbers, then a seeded random walk over step / zero-action step / masked random reset / masked reset from arrays
    under either clock; after every call every env's fields, Nusselt number, t / step counters against its own oracle instance"""
    rng = np.random.default_rng({"resident-2d": 41, "streaming-2d": 42, "3d": 43}[target])
    for q in range(3):
        worst, log = fuzz_seq.run_sequence(target, rng, 8)
        assert worst < 1e-9, log
