"""GPU parity tests of the streaming 2D path: 2D grids the LDS-resident kernel is not built for (float64 128x64, 192x32, any
other nx, nz >= 8) run on the 3D streaming kernels with ny = 1 (rbc3d_kernels.hpp, "Streaming-2D mode").  The reference takes any
`state_shape` (rbc2D.py:40-41 -> initialize_simulation(grid=...), rbc_sim2D_api.jl:17-25); these tests hold the streaming path
to the same oracle and the same tolerances as the resident kernel (tests/test_gpu_parity.py), and compare the two HIP paths
with each other on the default grid (RBC_FORCE_STREAM2D=1)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


@pytest.fixture(scope="module")
def native():
    from rbc_gym import _native
    return _native


@pytest.fixture(scope="module")
def oracle():
    import oracle_py
    oracle_py.build_oracle()
    return oracle_py


@pytest.mark.parametrize("cfg", [
    dict(nx=128, nz=64, heaters=12, heater_limit=0.75, obs=(8, 64), dt_solver=0.02, dt_control=0.06, ra=1e4),     # float64 state too large for a CU's LDS
    dict(nx=192, nz=32, heaters=12, heater_limit=0.75, obs=(8, 48), dt_solver=0.015, dt_control=0.05, ra=2e4),
    dict(nx=256, nz=64, heaters=16, heater_limit=0.9, obs=(16, 32), dt_solver=0.01, dt_control=0.03, ra=1e5, lx=4 * np.pi),
    dict(nx=100, nz=40, heaters=7, heater_limit=0.6, obs=(5, 25), dt_solver=0.03, dt_control=0.08, ra=1e4),       # nx = 10 x 10: generic two-factor DFT
    dict(nx=72, nz=27, heaters=5, heater_limit=0.5, obs=(9, 36), dt_solver=0.03, dt_control=0.07, ra=5e3, lz=1.5, min_b=0.5, delta_b=2.0),  # odd nz: unpacked z solve, cell-per-thread tendencies
    dict(nx=64, nz=40, heaters=8, heater_limit=0.75, obs=(8, 32), dt_solver=0.03, dt_control=0.07, ra=1e4),        # one-kernel projection, DFT-8 x 8 rows
    dict(nx=48, nz=24, heaters=6, heater_limit=0.75, obs=(6, 24), dt_solver=0.03, dt_control=0.07, ra=3e3),        # DFT-6 x 8
    dict(nx=32, nz=24, heaters=4, heater_limit=0.75, obs=(6, 16), dt_solver=0.03, dt_control=0.07, ra=3e3, lx=3.0),  # DFT-4 x 8
    dict(nx=192, nz=128, heaters=12, heater_limit=0.75, obs=(8, 48), dt_solver=0.01, dt_control=0.025, ra=1e5),     # spectrum too large for one workgroup: in-place separate kernels (DFT-24 x 8)
    dict(nx=256, nz=128, heaters=12, heater_limit=0.75, obs=(8, 64), dt_solver=0.01, dt_control=0.025, ra=1e6),     # DFT-32 x 8
    dict(nx=384, nz=32, heaters=12, heater_limit=0.75, obs=(8, 48), dt_solver=0.01, dt_control=0.025, ra=1e5, lx=8 * np.pi),  # nx > 256: no FLAT tiles, no in-place FFT -- generic DFT, z-marching tendencies
], ids=["128x64", "192x32", "256x64", "100x40", "72x27", "64x40", "48x24", "32x24", "192x128", "256x128", "384x32"])
def test_streaming_2d_grids_match_oracle(native, oracle, cfg):
    """random reset, then two actuated control intervals (incl. a clipped last substep where dt_control is not a multiple
    of dt_solver): fields at round-off of the oracle, Nusselt numbers, float32 observations (all five channels)."""
    cfg = dict(cfg)
    obs = cfg.pop("obs")
    sim = native.NativeSim(batch=2, obs_nz=obs[0], obs_nx=obs[1], random_kick=0.05, **cfg)
    seeds = np.array([5, 6], dtype=np.uint64)
    sim.reset(seeds)
    orcs = []
    for e in range(2):
        o = oracle.OracleSim(obs=obs, kick=0.05, **cfg)
        o.reset_random(int(seeds[e]))
        orcs.append(o)
    b, u, w = sim.get_fields()
    nus, nuo = sim.get_nusselt()
    for e, o in enumerate(orcs):
        ob, ou, ow = o.fields()
        assert np.abs(b[e] - ob).max() < 1e-14                            # the counter RNG and the clamp
        assert rel_l2(u[e], ou) < 1e-12 and rel_l2(w[e], ow) < 1e-12
        assert np.all(w[e][0] == 0) and np.all(w[e][-1] == 0)
        assert abs(nus[e] - o.nusselt(True)) < 1e-9 * max(1.0, abs(o.nusselt(True)))
        assert abs(nuo[e] - o.nusselt(False)) < 1e-9 * max(1.0, abs(o.nusselt(False)))
    rng = np.random.default_rng(1)
    for n in range(2):
        act = rng.uniform(-1.5, 1.5, (2, cfg["heaters"])).astype(np.float32)      # beyond [-1, 1]: exercises the K2 rescaling
        assert sim.step(act)
        b, u, w = sim.get_fields()
        nus, nuo = sim.get_nusselt()
        ob5 = sim.get_obs(5)
        st5 = sim.get_state(5)
        dx, dz = cfg.get("lx", 2 * np.pi) / cfg["nx"], cfg.get("lz", 2.0) / cfg["nz"]
        for e, o in enumerate(orcs):
            assert o.step(act[e])
            for x, y in zip((b[e], u[e], w[e]), o.fields()):
                assert rel_l2(x, y) < 1e-10
            div = (np.roll(u[e], -1, 1) - u[e]) / dx + (w[e][1:] - w[e][:-1]) / dz
            assert np.abs(div).max() < 1e-11
            assert abs(nus[e] - o.nusselt(True)) < 1e-7 * max(1.0, abs(o.nusselt(True)))
            assert abs(nuo[e] - o.nusselt(False)) < 1e-7 * max(1.0, abs(o.nusselt(False)))
            oo, so = o.obs_f32(5), o.state(5)
            assert np.allclose(ob5[e][:4], oo[:4], rtol=1e-5, atol=1e-5)
            assert np.allclose(st5[e][:4], so[:4], rtol=1e-5, atol=1e-5)
            assert np.abs(st5[e][4] - so[4]).max() < 1e-5 * max(np.abs(so[4]).max(), 1e-3)    # pNHS: zero-mean potential of the last stage
    t, s = sim.get_info()
    assert np.allclose(t, 2 * cfg["dt_control"]) and np.all(s == 3)


def test_one_kernel_projection_agrees_with_the_separate_kernels(native, monkeypatch):
    """Where an env's packed spectrum fits the LDS the projection is one kernel (k2s_project_fused: 128x64, 192x32, ...);
    RBC_NO_FUSE_PROJECT=1 runs the same grid on the separate kernels (row FFTs of several rows per workgroup, z sweeps,
    inverse FFTs, corrections)."""
    kw = dict(batch=3, nx=128, nz=64, obs_nx=64, obs_nz=8, dt_control=0.09, random_kick=0.05)
    act = np.random.default_rng(3).uniform(-1, 1, (2, 3, 12)).astype(np.float32)
    outs = []
    for flag in ("0", "1"):
        monkeypatch.setenv("RBC_NO_FUSE_PROJECT", flag)
        sim = native.NativeSim(**kw)
        sim.reset(np.array([1, 2, 3], dtype=np.uint64))
        for n in range(2):
            assert sim.step(act[n])
        outs.append((sim.get_fields(), sim.get_state(5), sim.get_nusselt()))
        sim.close()
    monkeypatch.delenv("RBC_NO_FUSE_PROJECT")
    for x, y in zip(outs[0][0], outs[1][0]):
        assert rel_l2(x, y) < 1e-11
    assert np.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-5) and np.allclose(outs[0][2], outs[1][2], rtol=1e-9)


def test_streaming_2d_tendencies_and_heater_profile_match_oracle(native, oracle):
    """128x64 with the reference's 12 heaters: the cubic blends of collate_actions_colin touch cell centres (Nx >= 128)."""
    kw = dict(nx=128, nz=64, heaters=12, dt_solver=0.02, dt_control=0.06)
    sim = native.NativeSim(batch=2, obs_nz=8, obs_nx=64, random_kick=0.1, **kw)
    seeds = np.array([3, 4], dtype=np.uint64)
    sim.reset(seeds)
    act = np.random.default_rng(0).uniform(-1, 1, (2, 12)).astype(np.float32)
    g = sim.debug_tendencies(act)
    for e in range(2):
        o = oracle.OracleSim(obs=(8, 64), kick=0.1, **kw)
        o.reset_random(int(seeds[e]))
        o.set_action(act[e]); o.update_state()
        go = o.tendencies()
        for f in "buw":
            assert np.abs(g[f][e] - go[f]).max() < 1e-11 * max(np.abs(go[f]).max(), 1.0), f


def test_streaming_and_resident_kernels_agree_on_the_default_grid(native, oracle, monkeypatch, ckpt_ra1e4, ckpt_ra1e5):
    """The two HIP designs on the same inputs (96x64, stored Ra=1e4 steady states and a chaotic Ra=1e5 state, two control
    intervals): same fields to round-off, same outputs; a masked checkpoint reset and the NaN flag behave alike."""
    ics = [(1e4, ckpt_ra1e4["b"][0], ckpt_ra1e4["u"][0], ckpt_ra1e4["w"][0]), (1e5, ckpt_ra1e5["b"][0], ckpt_ra1e5["u"][0], ckpt_ra1e5["w"][0]),
           (1e4, ckpt_ra1e4["b"][1], ckpt_ra1e4["u"][1], ckpt_ra1e4["w"][1])]
    B = len(ics)
    sims = []
    for force in ("0", "1"):
        monkeypatch.setenv("RBC_FORCE_STREAM2D", force)
        sim = native.NativeSim(batch=B)
        sim.set_rayleigh([ic[0] for ic in ics])
        sim.reset_from_arrays(np.stack([ic[1] for ic in ics]), np.stack([ic[2] for ic in ics]), np.stack([ic[3] for ic in ics]))
        sims.append(sim)
    monkeypatch.delenv("RBC_FORCE_STREAM2D")
    res, strm = sims
    assert strm.algorithmic_bytes_per_env_step() == res.algorithmic_bytes_per_env_step()
    rng = np.random.default_rng(4)
    for n in range(2):
        act = rng.uniform(-1, 1, (B, 12)).astype(np.float32)
        assert res.step(act) and strm.step(act)
        for x, y in zip(res.get_fields(), strm.get_fields()):
            assert rel_l2(y, x) < 1e-9
        for x, y in zip(res.get_nusselt(), strm.get_nusselt()):
            assert np.allclose(y, x, rtol=1e-7)
        assert np.allclose(strm.get_obs(5)[:, :4], res.get_obs(5)[:, :4], rtol=1e-5, atol=1e-5)
        s5, r5 = strm.get_state(5), res.get_state(5)
        assert np.allclose(s5[:, :4], r5[:, :4], rtol=1e-5, atol=1e-5)
        assert np.abs(s5[:, 4] - r5[:, 4]).max() < 1e-5 * max(np.abs(r5[:, 4]).max(), 1e-3)
    # the oracle on the streaming path directly (first IC, same two actions)
    o = oracle.OracleSim(ra=1e4)
    o.reset_from_arrays(*ics[0][1:])
    rng = np.random.default_rng(4)
    for n in range(2):
        assert o.step(rng.uniform(-1, 1, (B, 12)).astype(np.float32)[0])
    for x, y in zip(strm.get_fields(), o.fields()):
        assert rel_l2(x[0], y) < 1e-9
    # masked reset + NaN flag
    f_before = strm.get_fields()
    bad = np.stack([ic[1] for ic in ics]); bad[1, 10, 10] = np.nan
    strm.reset_from_arrays(bad, np.stack([ic[2] for ic in ics]), np.stack([ic[3] for ic in ics]), mask=[0, 1, 0])
    t, s = strm.get_info()
    assert t[1] == 0 and s[1] == 1 and t[0] == 3.0 and s[0] == 3
    f_after = strm.get_fields()
    assert np.array_equal(f_before[0][0], f_after[0][0]) and np.array_equal(f_before[1][2], f_after[1][2])
    assert not strm.step(np.zeros((B, 12), np.float32))
    assert list(strm.get_flags()) == [0, 1, 0]
    # a blown-up env comes back after a masked reset (the NaN must not survive in either state buffer)
    good = [np.stack([ic[q] for ic in ics]) for q in (1, 2, 3)]
    strm.reset_from_arrays(*good, mask=[0, 1, 0])
    res.reset_from_arrays(*good, mask=[0, 1, 0])
    act = rng.uniform(-1, 1, (B, 12)).astype(np.float32)
    assert strm.step(act) and res.step(act)
    for x, y in zip(res.get_fields(), strm.get_fields()):
        assert np.isfinite(y).all() and rel_l2(y[1], x[1]) < 1e-9
    with pytest.raises(native.RbcError):
        native.NativeSim(batch=1, nx=128, nz=64, obs_nx=64, obs_nz=8, precision=1).step(np.zeros((1, 12), np.float32))   # float32 128x64 is resident; not initialised
    native.NativeSim(batch=1, nx=100, nz=40, obs_nx=50, obs_nz=8, precision=1).close()     # float32 on the streaming path (round 3): see the test below


def test_streaming_2d_batch_members_are_independent_and_deterministic(native):
    B = 5
    kw = dict(nx=128, nz=64, obs_nx=64, obs_nz=8, dt_control=0.09)
    seeds = np.arange(11, 11 + B, dtype=np.uint64)
    act = np.random.default_rng(2).uniform(-1, 1, (B, 12)).astype(np.float32)
    sim = native.NativeSim(batch=B, **kw)
    sim.reset(seeds)
    assert sim.step(act)
    fb, nb, ob = sim.get_fields(), sim.get_nusselt(), sim.get_obs(5)
    for e in (0, 3):
        one = native.NativeSim(batch=1, **kw)
        one.reset(seeds[e:e + 1])
        assert one.step(act[e:e + 1])
        for a, c in zip(fb, one.get_fields()):
            assert np.array_equal(a[e], c[0])
        assert nb[0][e] == one.get_nusselt()[0][0] and np.array_equal(ob[e], one.get_obs(5)[0])


def test_gym_env_on_a_streaming_grid(native):
    """The drop-in env with a state_shape the resident kernel has no instantiation for: reset, wrapped steps, normalised
    observations written by the device, the reward-shaping wrapper's cell distances, render."""
    import rbc_gym  # noqa: F401  (registers the ids)
    from rbc_gym._gym import gym
    from rbc_gym.wrappers import VectorRBCNormalizeObservation
    env = gym.make("rbc_gym/RayleighBenardConvection2D-v0", state_shape=[64, 128], observation_shape=[8, 64], heater_duration=0.3, render_mode="rgb_array")
    obs, info = env.reset(seed=3)
    assert obs.shape == (3, 8, 64) and info["state"].shape == (3, 64, 128)
    obs2, rew, term, trunc, info = env.step(env.action_space.sample())
    assert np.isfinite(obs2).all() and np.isfinite(rew) and not term
    frame = env.render()
    assert frame.ndim == 3 and frame.shape[2] == 3
    env.close()
    venv = gym.make_vec("rbc_gym/RayleighBenardConvection2D-v0", num_envs=3, state_shape=[64, 128], observation_shape=[8, 64], heater_duration=0.3)
    raw, _ = venv.reset(seed=5)
    wrapped = VectorRBCNormalizeObservation(venv, heater_limit=0.75)
    o1, _ = wrapped.reset(seed=5)
    assert o1.shape == raw.shape and np.abs(o1).max() <= 1.3 + 1e-6 and not np.allclose(o1, raw)
    d = venv.sim.get_cell_distances(0.001)
    assert d.shape == (3,) and np.isfinite(d).all()
    venv.close()


def test_reward_shaping_falls_back_to_the_numpy_search_on_a_wide_grid(native):
    """`rbc_get_cell_distances` is one wave per mid-line (nx <= 256); on a wider streaming grid the vector wrapper must take
    the numpy peak search (wrappers.cell_distances on info['state']) instead of raising on every step."""
    import rbc_gym  # noqa: F401
    from rbc_gym._gym import gym
    from rbc_gym.wrappers import VectorRBCRewardShaping, cell_distances
    venv = gym.make_vec("rbc_gym/RayleighBenardConvection2D-v0", num_envs=2, state_shape=[32, 384], observation_shape=[8, 48], heater_duration=0.06)
    with pytest.raises(native.RbcError):
        venv.reset(seed=1)
        venv.sim.get_cell_distances(0.001)
    shaped = VectorRBCRewardShaping(venv, shaping_weight=0.5)
    shaped.reset(seed=1)
    for _ in range(2):
        obs, rew, term, trunc, info = shaped.step(np.zeros((2, 12), np.float32))
    assert shaped.device_search is False and np.isfinite(rew).all()
    state = venv.sim.get_state(3)
    assert np.array_equal(info["cell_dist"], cell_distances(state[:, 2, 32 // 2 - 1]))
    venv.close()


def test_two_live_handles_of_different_grids_step_alternately(native):
    """The dynamic-LDS ceiling of a kernel is process state: creating a handle on a smaller grid must not lower it under a
    live handle on a larger one (128x128 needs > 64 KiB in the row-FFT kernels, 128x32 far less; 3D: 64x64 vs 32x32 slabs)."""
    big = native.NativeSim(batch=2, nx=192, nz=128, obs_nx=48, obs_nz=8, dt_control=0.03, dt_solver=0.01)
    big.reset(np.array([1, 2], dtype=np.uint64))
    small = native.NativeSim(batch=2, nx=128, nz=64, obs_nx=64, obs_nz=8, dt_control=0.03, dt_solver=0.01)      # created AFTER: would have lowered the ceiling
    small.reset(np.array([3, 4], dtype=np.uint64))
    b3 = native.NativeSim3D(batch=4, shape=(16, 64, 64), dt_control=0.02, dt_solver=0.01)
    b3.reset(np.arange(4, dtype=np.uint64))
    s3 = native.NativeSim3D(batch=4, shape=(16, 32, 32), dt_control=0.02, dt_solver=0.01)
    s3.reset(np.arange(4, dtype=np.uint64))
    for _ in range(2):
        assert big.step(np.zeros((2, 12), np.float32)) and small.step(np.zeros((2, 12), np.float32))
        assert b3.step(np.zeros((4, 8, 8), np.float32)) and s3.step(np.zeros((4, 8, 8), np.float32))
    for sim in (big, small):
        assert np.isfinite(sim.get_nusselt()[0]).all()
    for sim in (b3, s3):
        assert np.isfinite(sim.get_nusselt()).all()
    for sim in (big, small, b3, s3):
        sim.close()


@pytest.mark.parametrize("cfg", [
    dict(nx=256, nz=64, heaters=16, heater_limit=0.9, obs=(16, 32), dt_solver=0.01, dt_control=0.03, ra=1e5, lx=4 * np.pi),     # FLAT tiles + one-kernel projection
    dict(nx=100, nz=40, heaters=7, heater_limit=0.6, obs=(5, 25), dt_solver=0.03, dt_control=0.08, ra=1e4),                   # generic DFT, z-marching tendencies
    dict(nx=192, nz=128, heaters=12, heater_limit=0.75, obs=(8, 48), dt_solver=0.01, dt_control=0.025, ra=1e5),                # in-place separate kernels
], ids=["256x64", "100x40", "192x128"])
def test_float32_streaming_2d_within_stated_tolerances_of_the_float64_oracle(native, oracle, cfg):
    """rbc_config.precision = float32 on grids without an LDS-resident float32 kernel: the rbc3f instantiation of the streaming
    kernels (state, tendencies, spectrum and potential in float32; outputs and Nusselt sums float64).  Stated tolerances
    against the float64 oracle from the same random reset over two actuated control intervals: b 2e-6 rel-L2, u and w 1e-4,
    Nusselt numbers 1e-4, divergence at float32 round-off."""
    cfg = dict(cfg)
    obs = cfg.pop("obs")
    sim = native.NativeSim(batch=2, obs_nz=obs[0], obs_nx=obs[1], random_kick=0.05, precision=1, **cfg)
    seeds = np.array([5, 6], dtype=np.uint64)
    sim.reset(seeds)
    orcs = []
    for e in range(2):
        o = oracle.OracleSim(obs=obs, kick=0.05, **cfg)
        o.reset_random(int(seeds[e]))
        orcs.append(o)
    b, u, w = sim.get_fields()
    for e, o in enumerate(orcs):
        ob, ou, ow = o.fields()
        assert np.abs(b[e] - ob).max() < 3e-7 and rel_l2(u[e], ou) < 2e-5 and rel_l2(w[e], ow) < 2e-5       # float32 storage of the same deviates
    rng = np.random.default_rng(3)
    for n in range(2):
        act = rng.uniform(-1, 1, (2, cfg["heaters"])).astype(np.float32)
        assert sim.step(act)
        for e, o in enumerate(orcs):
            assert o.step(act[e])
    b, u, w = sim.get_fields()
    nus, nuo = sim.get_nusselt()
    dx, dz = cfg.get("lx", 2 * np.pi) / cfg["nx"], cfg.get("lz", 2.0) / cfg["nz"]
    for e, o in enumerate(orcs):
        ob, ou, ow = o.fields()
        assert rel_l2(b[e], ob) < 2e-6 and rel_l2(u[e], ou) < 1e-4 and rel_l2(w[e], ow) < 1e-4, (rel_l2(b[e], ob), rel_l2(u[e], ou), rel_l2(w[e], ow))
        assert abs(nus[e] - o.nusselt(True)) < 1e-4 * abs(o.nusselt(True)) and abs(nuo[e] - o.nusselt(False)) < 1e-4 * abs(o.nusselt(False))
        div = (np.roll(u[e], -1, axis=1) - u[e]) / dx + (w[e][1:] - w[e][:-1]) / dz
        assert np.abs(div).max() < 2e-5 * max(np.abs(u[e]).max() / dx, 1e-3)
    sim.close()


def test_streaming_chunk_heights_and_chain_counts_change_nothing(native, monkeypatch):
    """The FLAT tile kernel marches 4 / 8 / 16 / 32 / 64 levels per workgroup (RBC_FLAT_KT; default by grid and batch) and the batch
    runs on 1-4 stream chains (RBC_3D_GROUPS; default 3 from B = 768 up): other instantiations and launch orders of the same
    per-cell expressions -- round-off between chunk heights (the compiler's contraction may differ), bitwise between chain counts."""
    B = 12
    act = np.random.default_rng(2).uniform(-1, 1, (2, B, 12)).astype(np.float32)
    outs = {}
    for name, env in (("kt16_g1", dict(RBC_FLAT_KT="16", RBC_3D_GROUPS="1")), ("kt16_g3", dict(RBC_FLAT_KT="16", RBC_3D_GROUPS="3")),
                      ("kt32", dict(RBC_FLAT_KT="32", RBC_3D_GROUPS="1")), ("kt64", dict(RBC_FLAT_KT="64", RBC_3D_GROUPS="1")),
                      ("kt4", dict(RBC_FLAT_KT="4", RBC_3D_GROUPS="2"))):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sim = native.NativeSim(batch=B, nx=128, nz=64, obs_nx=64, obs_nz=8, dt_control=0.09, dt_solver=0.03, random_kick=0.05)
        sim.reset(np.arange(B, dtype=np.uint64) + 3)
        for n in range(2):
            assert sim.step(act[n])
        outs[name] = (sim.get_fields(), sim.get_nusselt())
        sim.close()
        for k in env:
            monkeypatch.delenv(k)
    for x, y in zip(outs["kt16_g1"][0], outs["kt16_g3"][0]):
        assert np.array_equal(x, y)
    for name in ("kt32", "kt64", "kt4"):
        for x, y in zip(outs["kt16_g1"][0], outs[name][0]):
            assert rel_l2(y, x) < 1e-12, name
