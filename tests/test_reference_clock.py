"""GPU: the product switch `reference_clock="recorded"` (rbc_config.reference_clock, include/rbc_hip.h).

What the switch is: the reference's sources integrate every env-step to `stop_time` (rbc_sim2D_api.jl:84-85,
rbc_sim3D_api.jl:88-89: heater_duration / dt_solver solver steps), its only recorded time series
(experiments/flowstats/flowstats_ra.pkl) shows the growth of ALL solver steps in the first env-step after a reset and of ONE
LESS in every later env-step.  "documented" (default) ships the former, "recorded" the latter.  These
tests pin the mechanics on every path -- LDS-resident 2D kernel (float64, packed float32 pairs), streaming 2D, 3D (captured
graphs of both lengths) --: which env takes how many solver steps, per env across masked resets, with `t` / `step` advancing
as documented; tests/test_gpu_parity3d.py::test_recorded_clock_reproduces_the_whole_recorded_window holds the switch against
the reference's data.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DT = 0.03


@pytest.fixture(scope="module")
def native():
    from rbc_gym import _native
    return _native


def _ics(ckpt, B):
    idx = np.arange(B) % ckpt["b"].shape[0]
    return ckpt["b"][idx], ckpt["u"][idx], ckpt["w"][idx]


def _same(fa, fb):
    return all(np.array_equal(x, y) for x, y in zip(fa, fb))


@pytest.mark.parametrize("path", ["resident-f64", "resident-f32-pairs", "streaming-2d"])
def test_recorded_clock_2d_first_interval_full_later_ones_one_solver_step_short(native, ckpt_ra1e5, monkeypatch, path):
    """env-step 1 == the documented env-step bit for bit; env-step n >= 2 == (heater_duration / dt_solver - 1) solver steps;
    a masked reset gives the reset envs their full first interval again while the others keep the short one (mixed batch)."""
    if path == "streaming-2d":
        monkeypatch.setenv("RBC_FORCE_STREAM2D", "1")
    kw = {"precision": native.PRECISIONS["f32"]} if path == "resident-f32-pairs" else {}
    B, nsub = 5, 10                                                # odd batch: the packed variant's last pair is half empty
    rng = np.random.default_rng(11)
    acts = rng.uniform(-1, 1, (3, B, 12)).astype(np.float32)
    ics = _ics(ckpt_ra1e5, B)
    rec = native.NativeSim(batch=B, ra=1e5, dt_control=nsub * DT, reference_clock="recorded", **kw)
    doc = native.NativeSim(batch=B, ra=1e5, dt_control=nsub * DT, **kw)
    for sim in (rec, doc):
        sim.reset_from_arrays(*ics)
    assert rec.step(acts[0]) and doc.step(acts[0])
    assert _same(rec.get_fields(), doc.get_fields())              # first env-step after the reset: all nsub solver steps
    assert np.array_equal(rec.get_obs(5), doc.get_obs(5))
    assert rec.step(acts[1])
    doc.debug_substeps(acts[1], nsub - 1, DT)                     # every later one: nsub - 1
    assert _same(rec.get_fields(), doc.get_fields())
    t, step = rec.get_info()
    assert np.allclose(t, 2 * nsub * DT) and np.all(step == 3)     # the clocks advance as documented
    # masked reset of envs 1 and 4 (one of them shares a float32 pair with an env that is NOT reset)
    mask = np.zeros(B, np.uint8); mask[[1, 4]] = 1
    for sim in (rec, doc):
        sim.reset_from_arrays(*ics, mask=mask)
    assert rec.step(acts[2])
    doc.debug_substeps(acts[2], nsub - 1, DT)
    short = doc.get_fields()
    ref = native.NativeSim(batch=B, ra=1e5, dt_control=nsub * DT, **kw)       # the reset envs' full interval
    ref.reset_from_arrays(*ics)
    assert ref.step(acts[2])
    full = ref.get_fields()
    got = rec.get_fields()
    for e in range(B):
        want = full if mask[e] else short
        for x, y in zip(got, want):
            assert np.array_equal(x[e], y[e]), (path, e)
    t, step = rec.get_info()
    assert np.all(step == np.where(mask, 2, 4)) and np.allclose(t, np.where(mask, 1, 3) * nsub * DT)
    for sim in (rec, doc, ref):
        sim.close()


def test_recorded_clock_2d_matches_the_oracle_with_one_substep_less(native, ckpt_ra1e4):
    """the same schedule on the CPU oracle: set the action, refresh the state, nsub solver steps in the first interval and
    nsub - 1 in the second; a clipped last substep (heater_duration 1: 33 x 0.03 + 0.01, example/timing.py:10) is kept"""
    import oracle_py
    for dtc, nfull, rem in ((0.3, 10, 0.0), (0.25, 8, 0.01)):
        act = np.random.default_rng(5).uniform(-1, 1, (2, 1, 12)).astype(np.float32)
        ic = [ckpt_ra1e4[k][:1] for k in ("b", "u", "w")]
        sim = native.NativeSim(batch=1, dt_control=dtc, reference_clock="recorded")
        sim.reset_from_arrays(*ic)
        o = oracle_py.OracleSim(ra=1e4, dt_control=dtc)
        o.reset_from_arrays(*[x[0] for x in ic])
        for n in range(2):
            assert sim.step(act[n])
            o.set_action(act[n, 0]); o.update_state()
            for _ in range(nfull - (1 if n else 0)):
                o.substep(DT)
            if rem:
                o.substep(rem)
        for x, y in zip(sim.get_fields(), o.fields()):
            assert np.linalg.norm(x[0] - y) <= 1e-10 * np.linalg.norm(y), dtc
        sim.close()


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_recorded_clock_3d_graphs_of_both_lengths_and_mixed_batches(native, precision):
    """3D: B = 16 runs as four env groups replayed as captured graphs -- one graph per interval length; a masked reset makes a
    mixed batch (the fresh envs' extra solver step is taken by the whole batch and undone for the others)."""
    B, nsub, shape = 16, 5, (16, 32, 32)
    kw = dict(batch=B, shape=shape, ra=20000.0, dt_control=0.05, dt_solver=0.01, precision=precision)
    rng = np.random.default_rng(3)
    acts = rng.uniform(-1, 1, (4, B, 8, 8)).astype(np.float32)
    seeds = np.arange(B, dtype=np.uint64) + 40
    rec = native.NativeSim3D(reference_clock="recorded", **kw)
    doc = native.NativeSim3D(**kw)
    dts = 0.01 * 4.0                                               # solver step in model time (t_ff = Lz^2 = 4, rbc_sim3D_api.jl:43)
    for sim in (rec, doc):
        sim.reset(seeds)
    assert rec.step(acts[0]) and doc.step(acts[0])
    assert _same(rec.get_fields(), doc.get_fields())
    for n in (1, 2):                                               # two short intervals: both ping-pong parities of the short graph
        assert rec.step(acts[n])
        doc.debug_substeps(acts[n], nsub - 1, dts)
        assert _same(rec.get_fields(), doc.get_fields()), n
    t, step = rec.get_info()
    assert np.allclose(t, 3 * 0.05 * 4.0) and np.all(step == 4)
    mask = np.zeros(B, np.uint8); mask[[0, 5, 15]] = 1
    for sim in (rec, doc):
        sim.reset(seeds + 100, mask=mask)
    assert rec.step(acts[3])
    doc.debug_substeps(acts[3], nsub - 1, dts)
    short = doc.get_fields()
    ref = native.NativeSim3D(**kw)
    ref.reset(seeds + 100)
    assert ref.step(acts[3])
    full, got = ref.get_fields(), rec.get_fields()
    for e in range(B):
        want = full if mask[e] else short
        for x, y in zip(got, want):
            assert np.array_equal(x[e], y[e]), e
    assert np.array_equal(rec.get_flags(), np.zeros(B, np.int32))
    for sim in (rec, doc, ref):
        sim.close()


def test_recorded_clock_through_the_gym_layer(native):
    """env kwarg `reference_clock` on the single env and the vector env; NEXT_STEP autoreset restores the full first interval"""
    import rbc_gym  # noqa: F401
    from rbc_gym.envs import RayleighBenardConvection2DEnv
    from rbc_gym.vector import RayleighBenardConvection2DVectorEnv
    with pytest.raises(ValueError):
        RayleighBenardConvection2DEnv(reference_clock="julia")
    env = RayleighBenardConvection2DEnv(heater_duration=0.3, episode_length=0.6, reference_clock="recorded")
    dox = RayleighBenardConvection2DEnv(heater_duration=0.3, episode_length=0.6)
    a = np.linspace(-1, 1, 12).astype(np.float32)
    env.reset(seed=3); dox.reset(seed=3)
    o1, r1, _, tr1, i1 = env.step(a)
    p1, q1, _, _, j1 = dox.step(a)
    assert np.array_equal(o1, p1) and r1 == q1 and i1["t"] == j1["t"] == 0.3 and not tr1
    o2, r2, _, tr2, i2 = env.step(a)
    p2, q2, _, _, j2 = dox.step(a)
    assert not np.array_equal(o2, p2) and i2["t"] == j2["t"] and i2["step"] == j2["step"] == 3 and tr2
    env.close(); dox.close()
    venv = RayleighBenardConvection2DVectorEnv(num_envs=3, heater_duration=0.3, episode_length=0.6, reference_clock="recorded")
    vdoc = RayleighBenardConvection2DVectorEnv(num_envs=3, heater_duration=0.3, episode_length=0.6)
    acts = np.tile(a, (3, 1))
    venv.reset(seed=3); vdoc.reset(seed=3)
    for n in range(2):
        ov, *_ , trunc, _ = venv.step(acts)
        od, *_ = vdoc.step(acts)
    assert trunc.all() and not np.array_equal(ov, od)
    venv.step(acts); vdoc.step(acts)                                # NEXT_STEP: this call re-initialises every env
    ov, *_ = venv.step(acts)
    od, *_ = vdoc.step(acts)
    assert np.array_equal(ov, od)                                   # first interval after the autoreset: full again
    assert np.array_equal(ov[0], o1)                                # ... and the single env's (seed s + 0, same action)
    venv.close(); vdoc.close()


def test_recorded_clock_on_a_sharded_vector_env(native):
    """devices=[0, 0]: two handles behind one env object, each with its own fresh-env bookkeeping; bitwise the single-handle env,
    through a NEXT_STEP autoreset that restarts all envs and a later masked restart of one shard's env only"""
    from rbc_gym.vector import RayleighBenardConvection2DVectorEnv
    kw = dict(num_envs=5, heater_duration=0.3, episode_length=0.9, reference_clock="recorded")
    one = RayleighBenardConvection2DVectorEnv(**kw)
    two = RayleighBenardConvection2DVectorEnv(devices=[0, 0], **kw)
    rng = np.random.default_rng(9)
    one.reset(seed=21); two.reset(seed=21)
    for n in range(6):                                             # 3 steps to truncation, the autoreset step, two more
        a = rng.uniform(-1, 1, (5, 12)).astype(np.float32)
        ro, rt = one.step(a), two.step(a)
        assert np.array_equal(ro[0], rt[0]) and np.array_equal(ro[1], rt[1]) and np.array_equal(ro[3], rt[3]), n
        assert np.array_equal(ro[4]["t"], rt[4]["t"]) and np.array_equal(ro[4]["step"], rt[4]["step"])
    m = np.array([0, 0, 0, 1, 0], np.uint8)                        # env 3 lives on the second shard
    for env in (one, two):
        env._reset_envs(m)
    a = rng.uniform(-1, 1, (5, 12)).astype(np.float32)
    ro, rt = one.step(a), two.step(a)
    assert np.array_equal(ro[0], rt[0]) and np.array_equal(ro[4]["step"], rt[4]["step"]) and ro[4]["step"][3] == 2
    one.close(); two.close()
