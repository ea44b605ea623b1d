"""GPU parity, 3D: the streaming HIP path (through the C ABI) against the 3D CPU oracle on
identical inputs (fp64; tolerances per test).  The oracle itself is anchored on the 2D one
(tests/test_oracle3d.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SHAPE = (16, 24, 32)          # (nz, ny, nx): non-square slab exercises both FFT factorisations (4x6, 4x8)
DOMAIN = (2.0, 3 * np.pi, 4 * np.pi)


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


@pytest.fixture(scope="module")
def native():
    from rbc_gym import _native
    return _native


@pytest.fixture(scope="module")
def o3():
    import oracle_py
    oracle_py.build_oracle()
    return oracle_py


def _developed_state(o3, seed, ra=5000.0):
    """a few oracle substeps from a kicked state so that every term of the operator is active"""
    o = o3.Oracle3D(ra=ra, shape=SHAPE, domain=DOMAIN, kick=0.2)
    o.reset_random(seed)
    act = np.random.default_rng(seed).uniform(-1, 1, (8, 8)).astype(np.float32)
    o.set_action(act); o.update_state()
    for _ in range(3):
        o.substep(0.04)
    return o.fields()


def test_random_reset_and_projection(native, o3):
    seeds = np.array([3, 77], dtype=np.uint64)
    sim = native.NativeSim3D(batch=2, shape=SHAPE, domain=DOMAIN, ra=5000.0)
    sim.reset(seeds)
    b, u, v, w = sim.get_fields()
    nu = sim.get_nusselt()
    for e in range(2):
        o = o3.Oracle3D(ra=5000.0, shape=SHAPE, domain=DOMAIN)
        o.reset_random(int(seeds[e]))
        ob, ou, ov, ow = o.fields()
        assert np.abs(b[e] - ob).max() < 1e-14
        for x, y in ((u[e], ou), (v[e], ov), (w[e], ow)):
            assert np.abs(x - y).max() < 1e-13
        dx, dy, dz = DOMAIN[2] / SHAPE[2], DOMAIN[1] / SHAPE[1], DOMAIN[0] / SHAPE[0]
        div = (np.roll(u[e], -1, 2) - u[e]) / dx + (np.roll(v[e], -1, 1) - v[e]) / dy + (w[e][1:] - w[e][:-1]) / dz
        assert np.abs(div).max() < 1e-13
        assert abs(nu[e] - o.nusselt()) < 1e-10
    t, s = sim.get_info()
    assert np.all(t == 0) and np.all(s == 1)


def test_tendencies_match_oracle(native, o3):
    ics = [_developed_state(o3, 11), _developed_state(o3, 12, ra=20000.0)]
    ras = [5000.0, 20000.0]
    sim = native.NativeSim3D(batch=2, shape=SHAPE, domain=DOMAIN)
    sim.set_rayleigh(ras)
    sim.reset_from_arrays(*[np.stack([ic[q] for ic in ics]) for q in range(4)])
    act = np.random.default_rng(5).uniform(-1, 1, (2, 8, 8)).astype(np.float32)
    g = sim.debug_tendencies(act)
    for e in range(2):
        o = o3.Oracle3D(ra=ras[e], shape=SHAPE, domain=DOMAIN)
        o.reset_from_arrays(*ics[e])
        o.set_action(act[e]); o.update_state()
        go = o.tendencies()
        for f in "uvwb":
            assert np.abs(g[f][e] - go[f]).max() < 1e-11 * max(np.abs(go[f]).max(), 1.0), f


def test_env_step_matches_oracle(native, o3):
    ic = _developed_state(o3, 21)
    sim = native.NativeSim3D(batch=1, shape=SHAPE, domain=DOMAIN, ra=5000.0, dt_control=0.055, dt_solver=0.01)   # 5 x 0.04 + 0.02
    sim.reset_from_arrays(*[x[None] for x in ic])
    o = o3.Oracle3D(ra=5000.0, shape=SHAPE, domain=DOMAIN, dt_control=0.055, dt_solver=0.01)
    o.reset_from_arrays(*ic)
    for n in range(2):
        act = np.random.default_rng(30 + n).uniform(-1, 1, (1, 8, 8)).astype(np.float32)
        assert sim.step(act) and o.step(act[0])
        for x, y in zip(sim.get_fields(), o.fields()):
            assert rel_l2(x[0], y) < 1e-11
        assert abs(sim.get_nusselt()[0] - o.nusselt()) < 1e-9 * abs(o.nusselt())
        assert np.allclose(sim.get_state()[0], o.state(), rtol=1e-5, atol=1e-6)
    t, s = sim.get_info()
    assert abs(t[0] - 2 * 0.055 * 4) < 1e-12 and s[0] == 3            # rbc_sim3D_api.jl:89: time += dt * t_ff


def test_masked_reset_batch_independence_and_nan(native, o3):
    ic = _developed_state(o3, 41)
    B = 3
    arrs = [np.stack([x] * B) for x in ic]
    act = np.random.default_rng(2).uniform(-1, 1, (B, 8, 8)).astype(np.float32)
    act[2] = act[0]
    sim = native.NativeSim3D(batch=B, shape=SHAPE, domain=DOMAIN, ra=5000.0, dt_control=0.02)
    with pytest.raises(native.RbcError):
        sim.step(act)
    sim.reset_from_arrays(*arrs)
    assert sim.step(act)
    f = sim.get_fields()
    for q in range(4):
        assert np.array_equal(f[q][0], f[q][2]) and not np.array_equal(f[q][0], f[q][1])
    bad = [a.copy() for a in arrs]
    bad[0][1, 3, 3, 3] = np.nan
    sim.reset_from_arrays(*bad, mask=[0, 1, 0])
    g = sim.get_fields()
    assert np.array_equal(g[1][0], f[1][0]) and np.array_equal(g[1][2], f[1][2])      # unmasked envs untouched
    t, s = sim.get_info()
    assert list(s) == [2, 1, 2]
    assert not sim.step(act)
    assert list(sim.get_flags()) == [0, 1, 0]


def test_nan_env_recovers_after_masked_reset(native, o3):
    """An env that blew up must come back clean from a masked reset (the vector env's autoreset path tolerates a
    flagged env that is re-initialised in the same step, vector.py): its stale G^- may hold NaN, and stage 1 of the
    next substep multiplies G^- by zeta^1 = 0 -- it must not read it at all.  The reference rebuilds the whole model on
    reset (rbc_sim3D_api.jl:52-58)."""
    ic = _developed_state(o3, 43)
    B = 2
    arrs = [np.stack([x] * B) for x in ic]
    act = np.random.default_rng(8).uniform(-1, 1, (B, 8, 8)).astype(np.float32)
    act[1] = act[0]
    sim = native.NativeSim3D(batch=B, shape=SHAPE, domain=DOMAIN, ra=5000.0, dt_control=0.02)
    bad = [a.copy() for a in arrs]
    bad[1][0, 2, 5, 7] = np.nan                                     # u of env 0
    sim.reset_from_arrays(*bad)
    assert not sim.step(act)                                         # NaN spreads into env 0's state AND its G^-
    assert list(sim.get_flags()) == [1, 0]
    sim.reset_from_arrays(*arrs, mask=[1, 0])                        # masked reset of the broken env only
    sim.reset_from_arrays(*arrs, mask=[0, 1])                        # and bring env 1 back to the same state
    assert sim.step(act)
    f = sim.get_fields()
    assert list(sim.get_flags()) == [0, 0]
    for q in range(4):
        assert np.isfinite(f[q]).all()
        assert np.array_equal(f[q][0], f[q][1])                      # the recovered env equals one that never broke
    # same through a random reset
    sim.reset_from_arrays(*bad)
    assert not sim.step(act)
    sim.reset(np.array([5, 5], dtype=np.uint64))
    assert sim.step(act)
    g = sim.get_fields()
    for q in range(4):
        assert np.isfinite(g[q]).all() and np.array_equal(g[q][0], g[q][1])


def test_config4_shape_matches_oracle(native, o3):
    """BASELINE.json configs[4] exactly: 32x48x48, Ra=1e4, heater_duration 0.125, dt_solver 0.01 (x t_ff=4: 12 x 0.04 +
    0.02 = 13 RK3 substeps), B=2, from a DEVELOPED convecting state (the GPU runs from a kicked start under random actions
    until Nu > 1.5; the fields then go to both sides through reset_from_arrays).  48 = 6 x 8 takes the register-blocked
    FFT in x AND y, ny % 16 == 0 the 16x4 tiles, nz = 32 the packed z solve.  Tendencies 1e-11, one env-step 1e-10."""
    shape, ra = (32, 48, 48), 1e4
    B = 2
    gen = native.NativeSim3D(batch=B, shape=shape, ra=ra, dt_control=0.125, dt_solver=0.01, random_kick=0.1)
    gen.reset(np.array([11, 12], dtype=np.uint64))
    rng = np.random.default_rng(4)
    nu = None
    for n in range(400):
        assert gen.step(rng.uniform(-1, 1, (B, 8, 8)).astype(np.float32))
        nu = gen.get_nusselt()
        if n >= 60 and np.all(nu > 1.5):
            break
    assert np.all(nu > 1.5), nu                                       # convection has set in: every operator term is active
    ics = gen.get_fields()
    gen.close()
    assert np.abs(ics[3]).max() > 0.05
    sim = native.NativeSim3D(batch=B, shape=shape, ra=ra, dt_control=0.125, dt_solver=0.01)
    sim.reset_from_arrays(*ics)
    act = rng.uniform(-1, 1, (B, 8, 8)).astype(np.float32)
    g = sim.debug_tendencies(act)
    orcs = []
    for e in range(B):
        o = o3.Oracle3D(ra=ra, shape=shape, dt_control=0.125, dt_solver=0.01)
        o.reset_from_arrays(*[x[e] for x in ics])
        o.set_action(act[e]); o.update_state()
        go = o.tendencies()
        for f in "uvwb":
            assert np.abs(g[f][e] - go[f]).max() < 1e-11 * max(np.abs(go[f]).max(), 1.0), f
        orcs.append(o)
    assert sim.step(act)
    fields, nus, st = sim.get_fields(), sim.get_nusselt(), sim.get_state()
    for e, o in enumerate(orcs):
        assert o.step(act[e])
        for x, y in zip(fields, o.fields()):
            assert rel_l2(x[e], y) < 1e-10
        assert abs(nus[e] - o.nusselt()) < 1e-9 * abs(o.nusselt())
        assert np.allclose(st[e], o.state(), rtol=1e-5, atol=1e-6)
    t, s = sim.get_info()
    assert np.allclose(t, 0.5) and np.all(s == 2)


def test_float32_3d_within_stated_tolerances_of_the_float64_oracle(native, o3):
    """rbc_config.precision = float32 on the 3D path (the rbc3f instantiation of every streaming kernel; rbc3D.py:229-232 hands
    out float32 observations anyway) at BASELINE.json configs[4]'s shape, from a DEVELOPED convecting state, one env-step of 13
    substeps.  Stated tolerances against the float64 oracle on the identical initial fields (float32-rounded on upload):
    tendencies 2e-5 of their maximum, b 1e-6 rel-L2, u, v, w 1e-4, Nusselt number 1e-5; divergence at float32 round-off."""
    shape, ra, B = (32, 48, 48), 1e4, 2
    gen = native.NativeSim3D(batch=B, shape=shape, ra=ra, dt_control=0.125, dt_solver=0.01, random_kick=0.1)
    gen.reset(np.array([11, 12], dtype=np.uint64))
    rng = np.random.default_rng(4)
    for n in range(400):
        assert gen.step(rng.uniform(-1, 1, (B, 8, 8)).astype(np.float32))
        if n >= 60 and np.all(gen.get_nusselt() > 1.5):
            break
    ics = gen.get_fields()
    gen.close()
    act = rng.uniform(-1, 1, (B, 8, 8)).astype(np.float32)
    sim = native.NativeSim3D(batch=B, shape=shape, ra=ra, dt_control=0.125, dt_solver=0.01, precision="f32")
    sim.reset_from_arrays(*ics)
    g = sim.debug_tendencies(act)
    orcs = []
    for e in range(B):
        o = o3.Oracle3D(ra=ra, shape=shape, dt_control=0.125, dt_solver=0.01)
        o.reset_from_arrays(*[x[e] for x in ics])
        o.set_action(act[e]); o.update_state()
        go = o.tendencies()
        for f in "uvwb":
            assert np.abs(g[f][e] - go[f]).max() < 2e-5 * max(np.abs(go[f]).max(), 1.0), f
        orcs.append(o)
    assert sim.step(act)
    fields, nus, st = sim.get_fields(), sim.get_nusselt(), sim.get_state()
    dx, dz = 4 * np.pi / 48, 2.0 / 32
    for e, o in enumerate(orcs):
        assert o.step(act[e])
        errs = [rel_l2(x[e], y) for x, y in zip(fields, o.fields())]
        assert errs[0] < 1e-6 and max(errs[1:]) < 1e-4, errs
        assert abs(nus[e] - o.nusselt()) < 1e-5 * abs(o.nusselt())
        assert np.allclose(st[e], o.state(), rtol=1e-4, atol=1e-5)
        b, u, v, w = [x[e] for x in fields]
        div = (np.roll(u, -1, axis=2) - u) / dx + (np.roll(v, -1, axis=1) - v) / dx + (w[1:] - w[:-1]) / dz
        assert np.abs(div).max() < 2e-5 * np.abs(u).max() / dx
    t, s = sim.get_info()
    assert np.allclose(t, 0.5) and np.all(s == 2)
    sim.close()


def test_float32_3d_flowstats_statistics(native):
    """The reference's flow-statistics protocol (pin P4, flowstats_ra.py:27-36) in float32: last-100-step mean Nusselt numbers
    of all 14 Rayleigh numbers within the same bars as the float64 run (5 %, 12 % just above onset), Nu[0] at Ra = 500."""
    ref = {500: 1.368, 750: 1.513, 1000: 1.497, 1500: 1.668, 2000: 1.762, 4000: 2.128, 8000: 2.411, 16000: 2.851,
           32000: 3.453, 64000: 4.232, 128000: 5.233, 256000: 6.422, 512000: 7.886, 1000000: 9.212}
    ras = sorted(ref)
    seeds = 2
    B = seeds * len(ras)
    sim = native.NativeSim3D(batch=B, shape=(32, 64, 64), dt_control=0.25, dt_solver=0.005, precision="f32")
    sim.set_rayleigh(np.tile(np.array(ras, dtype=np.float64), seeds))
    sim.reset(np.arange(B, dtype=np.uint64) + 2024)
    zero = np.zeros((B, 8, 8), np.float32)
    nus = []
    for n in range(300):
        assert sim.step(zero)
        nus.append(sim.get_nusselt().copy())
    nus = np.array(nus)
    assert abs(nus[0, 0] - 1.0) < 1e-4
    means = nus[200:].mean(0).reshape(seeds, len(ras)).mean(0)
    for j, ra in enumerate(ras):
        tol = 0.12 if 750 <= ra <= 1500 else 0.05
        assert abs(means[j] - ref[ra]) < tol * ref[ra], (ra, means[j], ref[ra])
    sim.close()


def test_float32_3d_linear_growth_follows_the_theory(native, golden_dir):
    """The float32 3D path against the discretisation's linear theory (tests/linear_theory3d.py, no reference data, no oracle): 32
    members at Ra = 16000 on the flow-statistics protocol, increments of log(Nu-1) per env-step within 1 % of the theory's for 50
    solver steps (float64 ensembles: 0.05-0.3 %)."""
    import sys
    sys.path.insert(0, os.path.dirname(golden_dir))
    from linear_theory3d import LinearRBC3D
    B = 32
    sim = native.NativeSim3D(batch=B, shape=(32, 64, 64), ra=16000.0, dt_control=0.25, dt_solver=0.005, precision="f32")
    sim.reset(np.arange(B, dtype=np.uint64) + 600)
    zero = np.zeros((B, 8, 8), np.float32)
    nus = []
    for n in range(8):
        assert sim.step(zero)
        nus.append(sim.get_nusselt().copy())
    sim.close()
    la = np.log(np.array(nus) - 1.0).mean(1)
    th = np.log(LinearRBC3D(16000.0).nusselt_series(8, 0.02, 50))
    assert np.all(np.abs(np.diff(la)[2:7] / np.diff(th)[2:7] - 1.0) < 0.01), np.diff(la) / np.diff(th)
    assert abs(la[0] - th[0]) < 0.03


def test_fast_fft_sizes(native, o3):
    """48 = 6x8 and 64 = 8x8 take the register-blocked slab FFT (dft6 / dft8 x dft8); the other tests
    run 32 = 4x8 and the generic 24 = 4x6 path."""
    shape, dom = (8, 48, 64), (2.0, 4 * np.pi, 5 * np.pi)
    o = o3.Oracle3D(ra=8000.0, shape=shape, domain=dom, kick=0.1, dt_control=0.02)
    o.reset_random(9)
    sim = native.NativeSim3D(batch=1, shape=shape, domain=dom, ra=8000.0, dt_control=0.02)
    sim.reset_from_arrays(*[x[None] for x in o.fields()])
    for x, y in zip(sim.get_fields(), o.fields()):
        assert np.abs(x[0] - y).max() < 1e-13
    act = np.random.default_rng(4).uniform(-1, 1, (1, 8, 8)).astype(np.float32)
    assert sim.step(act) and o.step(act[0])
    for x, y in zip(sim.get_fields(), o.fields()):
        assert rel_l2(x[0], y) < 1e-11


def test_flowstats_pin_p4(native):
    """pin P4: the reference's experiments/flowstats run (grid 32x64x64, heater_duration 0.25,
    dt_solver 0.005, zero action, 300 steps; flowstats_ra.py:27-36) on the native 3D stepper at all 14 Rayleigh
    numbers, two seeds each; mean Nusselt of the last 100 steps vs the values measured from the reference's
    flowstats_ra.pkl (quoted from SURVEY.md: the pickle itself may not be loaded, scripts/flowstats3d.py).
    Statistical pin: the reference is ONE realisation per Ra whose own last-100 std is 1-3 %, so 5 % here; the three
    points just above onset (750-1500, where the reference's own series is not monotonic: 1.513 -> 1.497 -> 1.668)
    depend on the pattern selected and get 12 %.  Recorded single-seed run: tests/golden/flowstats3d_gpu.json
    (13 of 14 within 3.2 %, Ra=1000 at +10 %)."""
    ref = {500: 1.368, 750: 1.513, 1000: 1.497, 1500: 1.668, 2000: 1.762, 4000: 2.128, 8000: 2.411, 16000: 2.851,
           32000: 3.453, 64000: 4.232, 128000: 5.233, 256000: 6.422, 512000: 7.886, 1000000: 9.212}
    ras = sorted(ref)
    seeds = 2
    B = seeds * len(ras)
    sim = native.NativeSim3D(batch=B, shape=(32, 64, 64), dt_control=0.25, dt_solver=0.005)
    sim.set_rayleigh(np.tile(np.array(ras, dtype=np.float64), seeds))
    sim.reset(np.arange(B, dtype=np.uint64) + 2024)
    zero = np.zeros((B, 8, 8), np.float32)
    nus = []
    for n in range(300):
        assert sim.step(zero)
        nus.append(sim.get_nusselt().copy())
    nus = np.array(nus)
    assert np.all(np.abs(nus[0] - 1.0) < 3e-2) and abs(nus[0, 0] - 1.0) < 1e-4      # Nu[0] = 1.0000024 at Ra=500 in the reference data
    means = nus[200:].mean(0).reshape(seeds, len(ras)).mean(0)
    for j, ra in enumerate(ras):
        tol = 0.12 if 750 <= ra <= 1500 else 0.05
        assert abs(means[j] - ref[ra]) < tol * ref[ra], (ra, means[j], ref[ra])
    # the power law of the notebook's fit, Nu_max ~ 0.2211 Ra^0.2742 (flowstats_plots.ipynb cell 4), bounds the series
    assert np.all(nus[200:].max(0).reshape(seeds, len(ras)) < 1.25 * 0.2211 * np.array(ras) ** 0.2742 + 1.0)


def test_flowstats_velocity_maxima_in_the_steady_regime(native, golden_dir):
    """The three velocity series of the reference's flow-statistics data (max|u|, max|v|, max|w| of the float32 state after every
    env-step, flowstats_ra.py:65-67) in the statistically steady regime: last-100-step means on the native stepper (two members per
    Rayleigh number) against the reference's single realisation.  Recorded with four members (tests/golden/flowstats3d_velstats.json):
    from Ra = 8000 up all three maxima within 3 % (e.g. Ra = 64000: 0.7667 / 0.7773 / 0.8315 against 0.7673 / 0.7914 / 0.8489; the
    reference's own temporal std is 6-7 %); below, where the pattern selected decides, within 16 %.  Bars: 8 % / 20 %."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(golden_dir), "..", "scripts"))
    from flowstats3d_series import run_series
    ref = np.load(os.path.join(golden_dir, "flowstats_ref_series.npz"))
    out = run_series(ref["ra"], 2, 300, seed0=31)
    worst = 0.0
    for i, ra in enumerate(ref["ra"]):
        for k in ("umax", "vmax", "wmax"):
            mine, r = out[k][i, :, 200:].mean(), ref[k][i, 200:].mean()
            tol = 0.08 if ra >= 8000 else 0.20
            assert abs(mine - r) < tol * r, (ra, k, mine, r)
            if ra >= 8000:
                worst = max(worst, abs(mine / r - 1))
        nu, rn = out["nusselt"][i, :, 200:].mean(), ref["nusselt"][i, 200:].mean()
        assert abs(nu - rn) < (0.12 if 750 <= ra <= 1500 else 0.05) * rn, (ra, nu, rn)
    print(f"steady-regime velocity maxima: worst deviation from the reference at Ra >= 8000: {100 * worst:.1f} %")


def test_flowstats_series_pin(native, golden_dir):
    """Time-resolved pin on the reference's own data: the four per-step series of the 14 zero-action runs behind
    experiments/flowstats (flowstats_ra.py:55-66; tests/golden/flowstats_ref_series.npz, extracted from flowstats_ra.pkl by a
    non-executing opcode walk, tests/golden/extract_flowstats.py) against 8-member ensembles per Rayleigh number on the
    native 3D stepper at the reference's protocol (32x64x64, heater_duration 0.25, dt_solver 0.005; flowstats_ra.py:27-36).
    The reference is ONE realisation per Ra drawn from Julia's RNG, so every check is statistical.

      (a) steps 1-3, all 14 Ra: the decay of the random kick and the first growth, |z| < 4.5 (recorded 16-member run:
          max 2.6 of 42).  Nu-1 scales with kick^2: this pins the IC amplitude and the first control interval.
      (b) overshoot: the step of the first Nusselt maximum equals the ensemble's within one step (Ra >= 750); peak within 6 %.
      (c) the BUILD's clock, with no reference data: the ensemble's increments of log(Nu-1) equal those of the linear theory
          of the discretisation (tests/linear_theory3d.py: 50 RK3 substeps of 0.02 per env-step, no free parameter) within
          1 % per env-step and 0.4 % on average (recorded 16-member run: 0.05-0.3 %).
      (d) the SHAPE of the reference's deviation, tau(n) of scripts/flowstats3d_tau.py on all four series: a straight line
          tau(n) = 1 + 0.98 (n - 1) -- the first env-step agrees with the documented clock, every later one carries the growth
          of 49 solver steps instead of 50, from Nu-1 ~ 1e-4 on (not an amplitude / resolution effect) and at every Ra >= 4000
          alike (recorded: slope 0.9825 +- 0.0015, tau(1) = 1.007 +- 0.010; velocity maxima 0.976 +- 0.008, the same line).
      (e) the explanation run: 50 solver steps in the first env-step, 49 in every later one -> slope 1.000 +- 0.003 and the
          reference inside the ensemble through growth, overshoot and decay (rms z 1.0 over 30 env-steps at Ra <= 512000;
          documented clock: 1.8).
    The reference's sources integrate every env-step to stop_time (rbc_sim3D_api.jl:63,88-89), i.e. 50 solver steps, and
    that is what this build ships; DESIGN.md section 4 has the full account of what (d) and (e) can and cannot decide."""
    import sys
    for p in (os.path.join(os.path.dirname(golden_dir), "..", "scripts"), os.path.dirname(golden_dir)):
        sys.path.insert(0, p)
    from flowstats3d_series import run_series
    from flowstats3d_tau import analyse, summary
    from linear_theory3d import LinearRBC3D
    ref = np.load(os.path.join(golden_dir, "flowstats_ref_series.npz"))
    ras, seeds, steps = ref["ra"], 8, 45
    out = run_series(ras, seeds, steps, seed0=4242)
    nu = out["nusselt"]                                              # [ra, member, step]
    infl = np.sqrt(1.0 + 1.0 / seeds)
    worst = 0.0
    for i, ra in enumerate(ras):
        m, s = (nu[i] - 1).mean(0), (nu[i] - 1).std(0, ddof=1)
        r = ref["nusselt"][i, :steps] - 1
        z = (r[:3] - m[:3]) / (s[:3] * infl)
        worst = max(worst, float(np.abs(z).max()))
        assert np.all(np.abs(z) < 4.5), (ra, z)                       # (a)
        if ra < 750:
            continue                                                  # Ra=500 peaks at step ~65, outside this window
        am = nu[i].argmax(1)
        ar = int(np.argmax(ref["nusselt"][i, :steps]))
        assert am.max() < steps - 1, (ra, am)
        assert abs(ar - np.median(am)) <= 1, (ra, ar, am)             # (b)
        pk = nu[i].max(1)
        assert abs(ref["nusselt"][i, :steps].max() - pk.mean()) < 0.06 * pk.mean(), (ra, pk.mean())
    ratios = []
    for ra in (8000.0, 16000.0):                                      # (c)
        i = int(np.argmin(np.abs(ras - ra)))
        th = np.diff(np.log(LinearRBC3D(ra).nusselt_series(8, 0.02, 50)))[2:7]
        en = np.diff(np.log(nu[i][:, :8] - 1).mean(0))[2:7]
        assert np.all(np.abs(en / th - 1.0) < 0.01), (ra, en / th)
        ratios.append(en / th)
    assert abs(np.mean(ratios) - 1.0) < 0.004, ratios
    base = summary(analyse(ref, out))                                # (d)
    sn = base["nusselt"]
    assert 0.976 < sn["slope_mean"] < 0.989 and sn["slope_sem"] < 0.003 and abs(sn["tau1_mean"] - 1.0) < 0.05, sn
    vel = np.mean([base[k]["slope_mean"] for k in ("umax", "vmax", "wmax")])
    assert 0.93 < vel < 1.01, base
    # (e) the PRODUCT switch reference_clock="recorded" (rbc_config.reference_clock; no test-side hook): over the growth window as before ...
    out49 = run_series(ras, seeds, 30, seed0=4242, reference_clock="recorded")
    s49 = summary(analyse(ref, out49))["nusselt"]
    assert abs(s49["slope_mean"] - 1.0) < 0.008, s49
    def zscores(o):
        zs = []
        for i, ra in enumerate(ras):
            if 4000 <= ra <= 512000:
                la = np.log(o["nusselt"][i][:, :30] - 1)
                zs.append((np.log(ref["nusselt"][i, :30] - 1) - la.mean(0)) / (la.std(0, ddof=1) * infl))
        return np.array(zs)
    z50, z49 = zscores(out), zscores(out49)
    rms50, rms49 = float(np.sqrt((z50 ** 2).mean())), float(np.sqrt((z49 ** 2).mean()))
    # 8 members: z is t-distributed with 7 degrees of freedom, so the bar is on the rms and the 3-sigma fraction, not on the
    # maximum (recorded with 16 members: rms 1.02, all 240 below 2.6 -- against rms 1.84, 11 % beyond 3 for the documented clock)
    assert rms49 < 1.6 and (np.abs(z49) < 3).mean() > 0.96 and rms50 > 1.25 * rms49, (rms49, rms50)
    print(f"flowstats series pin: max |z| steps 1-3 = {worst:.2f}; build/theory increments {np.mean(ratios):.4f}; tau slope {sn['slope_mean']:.4f} "
          f"+- {sn['slope_sem']:.4f}, tau(1) {sn['tau1_mean']:.3f}; velocity slopes {vel:.3f}; recorded clock: slope {s49['slope_mean']:.4f}, rms z {rms49:.2f} (documented clock {rms50:.2f})")


def recorded_window_zscores(ref, out, steps):
    """z of the reference's single realisation against the ensemble, all four series of flowstats_ra.py:55-66 on their
    log-amplitudes, [series][ra][step]; members' spread inflated by sqrt(1 + 1/M) for the ensemble mean's own error"""
    from flowstats3d_tau import SERIES, log_amplitude
    z = {}
    for name in SERIES:
        M = out[name].shape[1]
        la = log_amplitude(name, out[name][:, :, :steps])
        z[name] = (log_amplitude(name, ref[name][:, :steps]) - la.mean(1)) / (la.std(1, ddof=1) * np.sqrt(1.0 + 1.0 / M))
    return z


def test_recorded_clock_reproduces_the_whole_recorded_window(native, golden_dir):
    """`reference_clock="recorded"` against EVERYTHING the reference recorded with a time axis: 14 Rayleigh numbers x the first
    60 env-steps x four series (Nu, max|u|, max|v|, max|w| of experiments/flowstats/flowstats_ra.pkl, flowstats_ra.py:55-66) --
    decay of the kick, linear growth, overshoot, decay towards the statistically steady state.  16 members per Ra on the native 3D
    stepper at the reference's protocol; the reference is one realisation, so z = (reference - ensemble mean) / member spread on
    the log-amplitudes.  Bars: rms z over the whole window <= 1.3 per series (a t-distribution with 15 degrees of freedom has
    rms 1.07), tau-slope of the Nusselt series 1.000 +- 0.005; the documented clock fails both (it is the default all the same:
    it is what rbc_sim3D_api.jl:88-89 says)."""
    import sys
    for p in (os.path.join(os.path.dirname(golden_dir), "..", "scripts"), os.path.dirname(golden_dir)):
        sys.path.insert(0, p)
    from flowstats3d_series import run_series
    from flowstats3d_tau import SERIES, analyse, summary
    ref = np.load(os.path.join(golden_dir, "flowstats_ref_series.npz"))
    ras, seeds, steps = ref["ra"], 16, 60
    rec = run_series(ras, seeds, steps, seed0=9090, reference_clock="recorded")
    doc = run_series(ras, seeds, steps, seed0=9090)
    srec, sdoc = summary(analyse(ref, rec))["nusselt"], summary(analyse(ref, doc))["nusselt"]
    zr, zd = recorded_window_zscores(ref, rec, steps), recorded_window_zscores(ref, doc, steps)
    rms = lambda z: float(np.sqrt((z ** 2).mean()))
    line = {k: (round(rms(zr[k]), 2), round(rms(zd[k]), 2)) for k in SERIES}
    print(f"recorded clock, 14 Ra x {steps} steps, 16 members: rms z per series (recorded, documented) {line}; tau slope "
          f"{srec['slope_mean']:.4f} +- {srec['slope_sem']:.4f} (documented {sdoc['slope_mean']:.4f}); worst |z| "
          f"{max(float(np.abs(zr[k]).max()) for k in SERIES):.1f}")
    assert abs(srec["slope_mean"] - 1.0) < 0.005 and abs(sdoc["slope_mean"] - 0.9825) < 0.006, (srec, sdoc)
    for k in SERIES:
        assert rms(zr[k]) <= 1.3, (k, line)
    assert rms(zd["nusselt"]) > 1.3 * rms(zr["nusselt"]), line
    all_r = np.concatenate([zr[k].ravel() for k in SERIES])
    assert (np.abs(all_r) < 3).mean() > 0.98, float((np.abs(all_r) < 3).mean())


def test_roundoff_sized_extra_substep_changes_nothing(native):
    """Oceananigans' `run!` clips its last step to stop_time - t; in double precision the clock often ends an env-step one
    ulp short and a 51st solver step of ~1e-16 follows (Simulations/run.jl `aligned_time_step`; rbc_sim3D_api.jl:88).  Such
    a step divides a round-off divergence by a round-off dt in the pressure solve; here it is shown to leave the state where
    it was (relative change < 1e-12) and the next env-step on the same trajectory -- it cannot be the missing 2 %."""
    B = 2
    zero = np.zeros((B, 8, 8), np.float32)
    sims = [native.NativeSim3D(batch=B, shape=(16, 32, 32), ra=20000.0, dt_control=0.25, dt_solver=0.005) for _ in range(2)]
    for sim in sims:
        sim.reset(np.arange(B, dtype=np.uint64) + 5)
    for n in range(6):
        for k, sim in enumerate(sims):
            assert sim.step(zero)
            if k == 1:
                before = sim.get_fields()
                sim.debug_substeps(zero, 1, 1.1e-16)
                after = sim.get_fields()
                for x, y in zip(before, after):
                    assert np.abs(x - y).max() <= 1e-12 * max(1.0, np.abs(x).max())
    fa, fb = sims[0].get_fields(), sims[1].get_fields()
    for x, y in zip(fa, fb):
        assert np.linalg.norm(x - y) <= 1e-10 * max(np.linalg.norm(x), 1e-30)
    for sim in sims:
        sim.close()


@pytest.mark.parametrize("shape", [SHAPE, (16, 32, 32)], ids=["tiles-8x8", "tiles-16x4"])
def test_marching_and_generic_tendency_kernels_agree(native, o3, monkeypatch, shape):
    """The LDS-tiled tendency kernels (nz % 8 == 0, ny % 8 == 0), the z-marching ones (nz % 4 == 0; RBC_NO_TILE=1) and
    the cell-per-thread ones (any grid; RBC_NO_MARCH=1) evaluate the same expressions; a grid with nz = 10 takes the
    generic path and is checked against the oracle."""
    o = o3.Oracle3D(ra=5000.0, shape=shape, domain=DOMAIN, kick=0.2)
    o.reset_random(5)
    o.set_action(np.zeros((8, 8), np.float32)); o.update_state()
    for _ in range(2):
        o.substep(0.04)
    ic = o.fields()
    act = np.random.default_rng(6).uniform(-1, 1, (1, 8, 8)).astype(np.float32)
    outs = []
    # LDS-tiled / z-marching / cell-per-thread tendencies; last: the unpacked Poisson path (one FFT per slab)
    for no_tile, no_march, no_pair in (("0", "0", "0"), ("1", "0", "0"), ("1", "1", "0"), ("0", "0", "1")):
        monkeypatch.setenv("RBC_NO_TILE", no_tile)
        monkeypatch.setenv("RBC_NO_MARCH", no_march)
        monkeypatch.setenv("RBC_NO_PAIR", no_pair)
        sim = native.NativeSim3D(batch=2, shape=shape, domain=DOMAIN, ra=5000.0, dt_control=0.03, dt_solver=0.01)
        sim.reset_from_arrays(*[np.stack([x, x[::1]]) for x in ic])
        assert sim.step(np.concatenate([act, act]))
        outs.append(sim.get_fields())
        sim.close()
    monkeypatch.setenv("RBC_NO_TILE", "0"); monkeypatch.setenv("RBC_NO_MARCH", "0"); monkeypatch.setenv("RBC_NO_PAIR", "0")
    monkeypatch.setenv("RBC_NO_FUSE_Z", "1")              # two-launch z sweeps instead of the fused one (nz = 16 here: HALF = 8)
    sim = native.NativeSim3D(batch=2, shape=shape, domain=DOMAIN, ra=5000.0, dt_control=0.03, dt_solver=0.01)
    sim.reset_from_arrays(*[np.stack([x, x[::1]]) for x in ic])
    assert sim.step(np.concatenate([act, act]))
    outs.append(sim.get_fields())
    sim.close()
    monkeypatch.delenv("RBC_NO_FUSE_Z")
    for other in outs[1:]:
        for x, y in zip(outs[0], other):
            assert rel_l2(x, y) < 1e-12
    ref = o3.Oracle3D(ra=5000.0, shape=shape, domain=DOMAIN, dt_control=0.03, dt_solver=0.01)
    ref.reset_from_arrays(*ic)
    assert ref.step(act[0])
    for x, y in zip(outs[0], ref.fields()):
        assert rel_l2(x[0], y) < 1e-11 and np.array_equal(x[0], x[1])          # oracle parity; batch members identical
    monkeypatch.delenv("RBC_NO_TILE")
    monkeypatch.delenv("RBC_NO_PAIR")
    monkeypatch.delenv("RBC_NO_MARCH")
    if shape != SHAPE:
        return
    shape = (10, 24, 32)
    o = o3.Oracle3D(ra=5000.0, shape=shape, domain=DOMAIN, kick=0.2, dt_control=0.03, dt_solver=0.01)
    o.reset_random(3)
    ic2 = o.fields()
    sim = native.NativeSim3D(batch=1, shape=shape, domain=DOMAIN, ra=5000.0, dt_control=0.03, dt_solver=0.01)
    sim.reset_from_arrays(*[x[None] for x in ic2])
    o.reset_from_arrays(*ic2)
    assert sim.step(act) and o.step(act[0])
    for x, y in zip(sim.get_fields(), o.fields()):
        assert rel_l2(x[0], y) < 1e-11


def test_env_groups_on_separate_streams_change_nothing(native, monkeypatch):
    """The batch is cut into env groups whose stage chains run on their own streams, replayed as one captured graph (default:
    4 groups from B = 16 up, tall 16 x 16 tiles).  Against the single chain on the handle's stream (RBC_3D_GROUPS=1, no graph)
    with the same tile shape nothing may change, bit for bit: three env-steps (both ping-pong parities of the graph and a
    replay), and uneven groups (16 envs in 3 groups: 6 + 6 + 4) launched without a graph.  Other tile shapes (the default
    picks one by the workgroup count of a launch) are other instantiations of the same expressions (the compiler's fma
    contraction may differ): round-off."""
    B, shape = 16, (16, 32, 32)
    act = np.random.default_rng(8).uniform(-1, 1, (3, B, 8, 8)).astype(np.float32)
    outs = []
    for env in (dict(RBC_3D_GROUPS="1", RBC_TILE_SHAPE="16x16"), dict(RBC_TILE_SHAPE="16x16"), dict(RBC_3D_GROUPS="3", RBC_USE_GRAPH="0", RBC_TILE_SHAPE="16x16"),
                dict(RBC_3D_SLICES="2", RBC_3D_GROUPS="2", RBC_TILE_SHAPE="16x16"),       # two sequential time slices of 8 envs, two chains each
                dict(), dict(RBC_3D_GROUPS="1"), dict(RBC_3D_GROUPS="2", RBC_TILE_SHAPE="16x8")):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sim = native.NativeSim3D(batch=B, shape=shape, domain=DOMAIN, ra=5000.0, dt_control=0.05, dt_solver=0.01, random_kick=0.2)
        sim.reset(np.arange(40, 40 + B, dtype=np.uint64))
        for n in range(3):
            assert sim.step(act[n])
        outs.append((sim.get_fields(), sim.get_nusselt(), sim.get_state(), sim.get_flags()))
        sim.close()
        for k in env:
            monkeypatch.delenv(k)
    for f, nu, st, fl in outs[1:4]:
        for x, y in zip(f, outs[0][0]):
            assert np.array_equal(x, y)
        assert np.array_equal(nu, outs[0][1]) and np.array_equal(st, outs[0][2]) and np.array_equal(fl, outs[0][3])
    for f, nu, st, fl in outs[4:]:
        for x, y in zip(f, outs[0][0]):
            assert rel_l2(x, y) < 1e-12
    assert len({float(x) for x in outs[0][1]}) == B             # 16 different envs, not copies


@pytest.mark.parametrize("shape,B,precision", [((32, 48, 48), 17, "f64"), ((32, 48, 48), 33, "f64"), ((24, 64, 16), 17, "f64"), ((16, 32, 32), 21, "f32")])
def test_an_env_result_does_not_depend_on_the_group_it_sits_in(native, shape, B, precision):
    """Batches that do not divide into equal env groups (17 = 5 + 4 + 4 + 4, 33 = 9 + 8 + 8 + 8): the tile shape follows the workgroup
    count of a launch, and 33 envs put the groups on both sides of the 16 x 16 threshold (9 envs: 108 workgroups, 8: 96 -- at the
    threshold, 6 would fall below).  Every group runs the instantiations picked for the handle's NOMINAL group size, so envs with the same seed and
    actions end bit for bit the same wherever they sit (found by scripts/fuzz_parity.py: the remainder group of 17 = 5 + 5 + 5 + 2
    ran shorter tiles and differed from its twins at round-off)."""
    sim = native.NativeSim3D(batch=B, shape=shape, ra=8300.0, heaters=3, dt_solver=0.0092, dt_control=0.0414, random_kick=0.1, precision=precision)
    sim.reset((np.arange(B, dtype=np.uint64) % np.uint64(2)) + np.uint64(77))
    rng = np.random.default_rng(0)
    for n in range(3):
        act = rng.uniform(-1, 1, (2, 3, 3)).astype(np.float32)[np.arange(B) % 2]
        assert sim.step(act)
        for a in list(sim.get_fields()) + [sim.get_nusselt(), sim.get_state()]:
            for e in range(2, B):
                assert np.array_equal(a[e], a[e % 2]), (n, e)
    f = sim.get_fields()
    assert not np.array_equal(f[0][0], f[0][1])                  # two different envs, repeated
    sim.close()


def test_marching_inverse_fft_is_bitwise_the_separate_vertical_correction(native, monkeypatch):
    """k3_ifft_march (float64 default since round 4: the inverse-FFT workgroup walks two adjacent mirror-packed slab pairs, keeps
    the previous pair's potential at its own columns in registers and applies w -= dts dphi/dz itself) against k3_ifft_pair +
    k3_correct_w (RBC_IFFT_MARCH=0): the same expressions on the same operands in the same order, so the state is the same BIT
    FOR BIT -- random reset (the masked projection of set!), three actuated env-steps with a clipped last substep, a masked
    reset in between; on configs[4]'s grid (6 columns per thread), the registry default (16 x 32 x 32: nz/2 = 8, 4 columns) and
    the flow-statistics grid in float32 (64 x 64: 8 columns per thread; in float64 that grid keeps the separate pass, so both runs take
    it); float32 goes through the switch (its default is the separate pass)."""
    for B, shape, prec in ((8, (32, 48, 48), "f64"), (3, (16, 32, 32), "f64"), (2, (32, 64, 64), "f64"), (4, (32, 48, 48), "f32"), (2, (32, 64, 64), "f32")):
        act = np.random.default_rng(B).uniform(-1, 1, (3, B, 8, 8)).astype(np.float32)
        outs = []
        for flag in ("0", "2"):
            monkeypatch.setenv("RBC_IFFT_MARCH", flag)
            sim = native.NativeSim3D(batch=B, shape=shape, ra=9000.0, dt_control=0.035, dt_solver=0.01, random_kick=0.2, precision=prec)
            sim.reset(np.arange(7, 7 + B, dtype=np.uint64))
            f0 = sim.get_fields()
            assert sim.step(act[0]) and sim.step(act[1])
            mask = np.zeros(B, np.uint8); mask[B - 1] = 1
            sim.reset(np.arange(70, 70 + B, dtype=np.uint64), mask=mask)
            assert sim.step(act[2])
            outs.append((f0, sim.get_fields(), sim.get_nusselt(), sim.get_state()))
            sim.close()
        monkeypatch.delenv("RBC_IFFT_MARCH")
        for x, y in zip(outs[0][0] + outs[0][1], outs[1][0] + outs[1][1]):
            assert np.array_equal(x, y), (shape, prec)
        assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])
        b, u, v, w = outs[1][1]
        assert np.all(w[:, 0] == 0) and np.all(w[:, -1] == 0) and np.isfinite(w).all()


def test_constant_grid_instantiations_agree_with_the_generic_ones(native, monkeypatch):
    """configs[4]'s 48 x 48 planes run tile kernels instantiated with nx, ny as compile-time constants (index arithmetic by
    multiplication); RBC_NO_CONST_GRID=1 selects the generic instantiations of the same bodies.  B = 8 takes the 16 x 16 tiles,
    B = 2 the 16 x 4 ones."""
    for B, shape in ((8, (32, 48, 48)), (2, (32, 48, 48)), (16, (16, 32, 32)), (2, (16, 64, 64))):      # + the registry default and the flowstats planes
        act = np.random.default_rng(B).uniform(-1, 1, (B, 8, 8)).astype(np.float32)
        outs = []
        for flag in ("0", "1"):
            monkeypatch.setenv("RBC_NO_CONST_GRID", flag)
            sim = native.NativeSim3D(batch=B, shape=shape, ra=1e4, random_kick=0.1)
            sim.reset(np.arange(70, 70 + B, dtype=np.uint64))
            assert sim.step(act)
            outs.append(sim.get_fields())
            sim.close()
        monkeypatch.delenv("RBC_NO_CONST_GRID")
        for x, y in zip(*outs):
            assert np.isfinite(x).all() and rel_l2(x, y) < 1e-12


def test_3d_env_groups_on_the_legacy_default_stream(native):
    """rbc_set_stream(hipStreamLegacy): the env-step cannot be captured into a graph on that stream, so the group chains are
    launched directly, forked from and joined into the legacy stream by events -- same results, and a torch kernel queued behind
    the step on torch's default stream sees the finished state without an explicit synchronisation."""
    torch = pytest.importorskip("torch")
    B, shape = 16, (16, 32, 32)
    kw = dict(batch=B, shape=shape, domain=DOMAIN, ra=5000.0, dt_control=0.05, dt_solver=0.01, random_kick=0.2)
    ref, sim = native.NativeSim3D(**kw), native.NativeSim3D(**kw)
    seeds = np.arange(5, 5 + B, dtype=np.uint64)
    ref.reset(seeds); sim.reset(seeds)
    sim._check(sim.lib.rbc_set_stream(sim.h, native.torch_stream_handle(torch.cuda.current_stream())))
    from rbc_gym.vector import DeviceArray
    nz, ny, nx = shape
    view = torch.as_tensor(DeviceArray(sim.lib.rbc_dev_state(sim.h), (B, 4, nz, ny, nx), "<f4", sim), device="cuda")
    gen = torch.Generator(device="cuda"); gen.manual_seed(2)
    for step in range(2):
        acts = (torch.rand((B, 8, 8), device="cuda", generator=gen) * 2 - 1).contiguous()      # produced on torch's stream, never synchronised
        sim.step_dev(acts.data_ptr())
        got = view.clone().cpu().numpy()                                                       # torch reads the view on the same stream
        assert ref.step(acts.cpu().numpy())
        assert np.array_equal(ref.get_state(), got)
    sim._check(sim.lib.rbc_set_stream(sim.h, None))
    sim.close(); ref.close()


def test_env_groups_replay_their_own_graphs_on_their_own_hardware_queues(native):
    """The runtime spreads a process's streams over four hardware queues by a rule of its own (scripts/probes/queue_map.hip): four
    streams created in a row can land on two queues, and two chains on one queue run one after the other.  rbc_create probes
    (include/rbc_hip.h rbc_debug_launch_plan) -- also with other streams and other handles alive in the process, which shift the
    rule's phase -- and the per-group graphs give bit for bit the state of the one graph over all groups (RBC_3D_GROUP_GRAPHS=0)."""
    import torch
    extra = [torch.cuda.Stream() for _ in range(3)]                 # somebody else's streams, created first
    kw = dict(batch=16, shape=(16, 32, 32), ra=5000.0, dt_control=0.05, dt_solver=0.01, random_kick=0.2)
    sims = [native.NativeSim3D(**kw) for _ in range(3)]             # three live handles: 12 group streams on 4 queues
    for sim in sims:
        assert sim.launch_plan() == (4, 1)
    small = native.NativeSim3D(batch=3, shape=(16, 32, 32))
    assert small.launch_plan() == (1, 0)                            # one group: the handle's stream, direct launches
    small.close()
    act = np.random.default_rng(8).uniform(-1, 1, (3, 16, 8, 8)).astype(np.float32)
    os.environ["RBC_3D_GROUP_GRAPHS"] = "0"
    try:
        ref = native.NativeSim3D(**kw)
    finally:
        del os.environ["RBC_3D_GROUP_GRAPHS"]
    assert ref.launch_plan() == (4, 0)
    seeds = np.arange(16, dtype=np.uint64) + 9
    for sim in sims + [ref]:
        sim.reset(seeds)
    for n in range(3):                                              # the handles step alternately: both parities of every graph
        for sim in sims + [ref]:
            assert sim.step(act[n])
    want = ref.get_fields()
    for sim in sims:
        for x, y in zip(sim.get_fields(), want):
            assert np.array_equal(x, y)
        assert np.array_equal(sim.get_nusselt(), ref.get_nusselt()) and np.array_equal(sim.get_state(), ref.get_state())
    for sim in sims + [ref]:
        sim.close()
    del extra
