"""CPU: the oracle against the golden vectors derived from the reference's checkpoint DATA
(tests/golden/*, generator make_fixtures.py).  These pin the restated discretisation; see
DESIGN.md "Oracle" for what each pin can and cannot discriminate."""
import json
import os

import numpy as np
import pytest

import oracle_py
from oracle_py import OracleSim, VAR_BOUNDS, VAR_POISSON, VAR_SYMLEVEL, VAR_BUOYANCY, VAR_VISCOUS

DX, DZ = 2 * np.pi / 96, 2 / 64


@pytest.fixture(scope="module")
def pins(golden_dir):
    return json.load(open(os.path.join(golden_dir, "ckpt2d_pins.json")))


def test_checkpoint_states_are_discretely_incompressible(ckpt_ra1e4, ckpt_ra1e5, pins):
    """pin P1: C-grid staggering + exact projection (max|div| <= 1.3e-14 in the data)."""
    for ck in (ckpt_ra1e4, ckpt_ra1e5):
        for e in range(ck["b"].shape[0]):
            u, w = ck["u"][e], ck["w"][e]
            div = (np.roll(u, -1, 1) - u) / DX + (w[1:] - w[:-1]) / DZ
            assert np.abs(div).max() < 2e-14
            assert np.all(w[0] == 0) and np.all(w[-1] == 0)
    worst = max(ep["max_abs_div"] for f in pins.values() for ep in f["episodes"])
    assert worst < 2e-14


def test_nusselt_restatement_matches_numpy_pins(ckpt_ra1e4, pins):
    """get_nusselt (rbc_sim2D_api.jl:142-163) restated in C == restated in numpy on the same data."""
    eps = pins["train/ckpt_ra10000"]["episodes"]
    for e in range(3):
        s = OracleSim(ra=1e4)
        s.load_raw(ckpt_ra1e4["b"][e], ckpt_ra1e4["u"][e], ckpt_ra1e4["w"][e])
        assert abs(s.nusselt(True) - eps[e]["nusselt_state"]) < 1e-11
        assert abs(s.nusselt(False) - eps[e]["nusselt_obs"]) < 1e-11
        assert abs(s.kinetic_energy() - eps[e]["ke"]) < 1e-14
        assert s.max_divergence() < 2e-14


def test_projection_is_exact_and_leaves_checkpoint_unchanged(ckpt_ra1e4):
    s = OracleSim(ra=1e4)
    b0, u0, w0 = ckpt_ra1e4["b"][0], ckpt_ra1e4["u"][0], ckpt_ra1e4["w"][0]
    s.reset_from_arrays(b0, u0, w0)       # set!'s projection on an already solenoidal state
    b, u, w = s.fields()
    assert np.array_equal(b, b0)
    assert np.abs(u - u0).max() < 1e-13 and np.abs(w - w0).max() < 1e-13
    assert s.max_divergence() < 2e-14
    assert s.info() == (0.0, 1)


def test_two_poisson_formulations_agree(ckpt_ra1e5):
    """FFT-x + tridiagonal-z vs the reference solver's eigenfunction expansion (DFT x DCT-II)."""
    b0, u0, w0 = ckpt_ra1e5["b"][0], ckpt_ra1e5["u"][0], ckpt_ra1e5["w"][0]
    rng = np.random.default_rng(0)
    u1 = u0 + 0.05 * rng.standard_normal(u0.shape)          # make the projection do real work
    w1 = w0.copy(); w1[1:-1] += 0.05 * rng.standard_normal(w0[1:-1].shape)
    out = []
    for var in (0, 1):
        s = OracleSim(ra=1e5, variants={VAR_POISSON: var})
        s.reset_from_arrays(b0, u1, w1)
        assert s.max_divergence() < 1e-12
        out.append(s.fields())
    assert np.abs(out[0][1] - out[1][1]).max() < 1e-12 and np.abs(out[0][2] - out[1][2]).max() < 1e-12


@pytest.mark.parametrize("nx,nz", [(61, 32), (17, 27), (9, 8), (62, 32), (100, 40)])
def test_projection_is_exact_on_any_grid_odd_sizes_included(nx, nz):
    """The randomised GPU<->oracle sweep (scripts/fuzz_parity.py) found the oracle's real inverse DFT assuming an even nx (a Nyquist
    mode that an odd nx does not have): the HIP path was divergence-free on 61 x 32, the oracle was not.  Both Poisson formulations
    of the oracle project exactly and agree on odd, prime and even grids."""
    rng = np.random.default_rng(nx)
    b0 = 1.5 + 0.1 * rng.standard_normal((nz, nx)); u0 = 0.1 * rng.standard_normal((nz, nx))
    w0 = 0.1 * rng.standard_normal((nz + 1, nx)); w0[0] = w0[-1] = 0
    out = []
    for var in (0, 1):
        s = OracleSim(nx=nx, nz=nz, heaters=4, obs=(nz, nx), dt_solver=0.01, dt_control=0.03, ra=2e4, variants={VAR_POISSON: var})
        s.reset_from_arrays(b0, u0, w0)
        assert s.max_divergence() < 1e-12
        assert s.step(np.linspace(-1, 1, 4).astype(np.float32)) and s.max_divergence() < 1e-12
        out.append(s.fields())
    for x, y in zip(*out):
        assert np.abs(x - y).max() < 1e-11


def test_stored_states_are_near_fixed_points_of_the_restated_operator(ckpt_ra1e4):
    """pin P2 (operator level).  The Ra=1e4 states sit on a weakly damped oscillation around a
    steady roll pattern: d/dt of the restated semi-discrete system must be at the oscillation's
    own level (~1e-3) everywhere INCLUDING the wall-adjacent rows.  The alternative near-wall
    stencil selections leave a 10x larger residual there, which is how the pinned variant was
    chosen."""
    b0, u0, w0 = ckpt_ra1e4["b"][0], ckpt_ra1e4["u"][0], ckpt_ra1e4["w"][0]
    s = OracleSim(ra=1e4)
    s.load_raw(b0, u0, w0)
    r = s.projected_rate()
    assert np.abs(r["b"]).max() < 1.5e-3 and np.abs(r["u"]).max() < 3e-3 and np.abs(r["w"]).max() < 2e-3
    wall = np.r_[0, 1, 62, 63]
    assert np.abs(r["b"][wall]).max() < 1.0e-3
    for bad in (1, 2):
        s2 = OracleSim(ra=1e4, variants={VAR_BOUNDS: bad})
        s2.load_raw(b0, u0, w0)
        r2 = s2.projected_rate()
        assert np.abs(r2["b"][wall]).max() > 1e-2


def test_equivalent_formulations_agree():
    """split hydrostatic pressure vs buoyancy in w, stress-divergence vs Laplacian viscosity:
    identical after the exact projection (up to round-off)."""
    base = OracleSim(ra=1e5)
    base.reset_random(5)
    f0 = base.fields()
    act = np.linspace(-1, 1, 12).astype(np.float32)
    ref = None
    for var in ({}, {VAR_BUOYANCY: 1}, {VAR_VISCOUS: 1}):
        s = OracleSim(ra=1e5, dt_control=0.09, variants=var)
        s.reset_from_arrays(*f0)
        assert s.step(act)
        f = s.fields()
        if ref is None:
            ref = f
        else:
            for a, c in zip(f, ref):
                assert np.abs(a - c).max() < 1e-11


def test_oracle_stays_on_the_reference_attractor(ckpt_ra1e4, pins):
    """pin P2 (trajectory level): continuing a stored Ra=1e4 state with zero action keeps kinetic
    energy and both Nusselt numbers inside the band spanned by the 40 reference episodes."""
    eps = [e for sp in ("train", "val", "test") for e in pins[f"{sp}/ckpt_ra10000"]["episodes"]]
    ke = np.array([e["ke"] for e in eps]); nus = np.array([e["nusselt_state"] for e in eps]); nuo = np.array([e["nusselt_obs"] for e in eps])
    s = OracleSim(ra=1e4)
    s.reset_from_arrays(ckpt_ra1e4["b"][0], ckpt_ra1e4["u"][0], ckpt_ra1e4["w"][0])
    for n in range(4):
        assert s.step(None)
        assert ke.min() - 3e-7 < s.kinetic_energy() < ke.max() + 3e-7
        assert nus.min() - 5e-5 < s.nusselt(True) < nus.max() + 5e-5
        assert nuo.min() - 1e-4 < s.nusselt(False) < nuo.max() + 1e-4
        assert s.max_divergence() < 5e-14
    t, step = s.info()
    assert t == 6.0 and step == 5


def test_the_lost_solver_step_of_a_run_reentry_is_lost_time_not_lost_amplitude(ckpt_ra1e4, pins):
    """The reference's recorded series carry the growth of one solver step less per `run!` re-entry (DESIGN.md section 4;
    `reference_clock="recorded"`).  In the linear phase a loss of TIME (one solver step not taken) and a loss of AMPLITUDE
    (perturbations scaled by e^-0.006, i.e. Nu - 1 by e^-0.012, once per re-entry) fit those series equally well.  The
    reference's own 2D checkpoints tell them apart: their generator re-enters `run!` every 10 solver steps
    (rbc_sim2D.jl:189-194 with --delta_t_snap 0.3, scripts/create_checkpoints_2D.sh:18-20), 2000 times per episode, and its 40
    stored Ra = 1e4 states sit on ONE steady state (kinetic energy spread 6e-7).  A time loss leaves a fixed point where it
    is; an amplitude loss per re-entry moves it to where the flow regrows the loss every 10 steps.  Here both on the oracle,
    from a stored state: 9 instead of 10 solver steps per re-entry stays inside the 40-episode band, the amplitude factor
    leaves it by four orders of magnitude more than the band is wide.  So the recorded loss is a loss of time."""
    eps = [e for sp in ("train", "val", "test") for e in pins[f"{sp}/ckpt_ra10000"]["episodes"]]
    ke = np.array([e["ke"] for e in eps])
    cond = (1.0 + (2.0 - (np.arange(64) + 0.5) / 32.0) * 0.5)[:, None]        # conduction profile min_b + (Lz - z) delta_b / 2

    def reentries(nsub, factor, n):
        s = OracleSim(ra=1e4)
        s.reset_from_arrays(ckpt_ra1e4["b"][0], ckpt_ra1e4["u"][0], ckpt_ra1e4["w"][0])
        for _ in range(n):
            s.set_action(np.zeros(12, np.float32)); s.update_state()
            for _ in range(nsub):
                s.substep(0.03)
            if factor != 1.0:
                b, u, w = s.fields()
                s.load_raw(cond + (b - cond) * factor, u * factor, w * factor)
        return s.kinetic_energy()

    time_loss = reentries(9, 1.0, 30)
    amp_loss = reentries(10, np.exp(-0.006), 30)
    assert ke.min() - 1e-6 < time_loss < ke.max() + 1e-6, time_loss
    assert ke.mean() - amp_loss > 5e-3 > 1000 * (ke.max() - ke.min()), (amp_loss, ke.mean())


def test_recorded_ensemble_pin(golden_dir):
    """The slow from-rest ensemble (tests/golden/oracle_ensemble.py, run by hand, result committed):
    12 independent oracle runs to t=600 vs the 40 Oceananigans episodes."""
    r = json.load(open(os.path.join(golden_dir, "oracle_ensemble_ra10000.json")))
    ref, me, alt = r["reference"], r["oracle_symlevel0"], r["oracle_symlevel1"]
    assert abs(me["ke_mean"] - ref["ke_mean"]) / ref["ke_mean"] < 5e-6
    assert abs(me["ke_z"]) < 2.0 and me["profile_max_abs_z"] < 3.0
    assert abs(me["nu_state_mean"] - ref["nu_state_mean"]) < 3 * np.hypot(me["nu_state_sem"], ref["nu_state_sem"])
    assert alt["profile_max_abs_z"] > 5.0       # the rejected advecting-velocity variant is detectably off


def test_heater_profile(ckpt_ra1e4):
    """collate_actions_colin (rbc_sim2D.jl:87-133): at Nx=96 no cell centre falls into a blend zone,
    so the profile is piecewise constant, 8 cells per heater, mean 2, |T-2| <= heater_limit."""
    s = OracleSim()
    rng = np.random.default_rng(1)
    for _ in range(5):
        a = rng.uniform(-1, 1, 12).astype(np.float32)
        s.set_action(a)
        tb = s.bottom_profile()
        v = 0.75 * a.astype(np.float64)
        k2 = max(1.0, np.abs(v - v.mean()).max() / 0.75)
        expect = np.repeat(2 + (v - v.mean()) / k2, 8)
        assert np.allclose(tb, expect, rtol=0, atol=1e-15)
        assert abs(tb.mean() - 2) < 1e-14 and np.abs(tb - 2).max() <= 0.75 + 1e-15
    s.set_action(np.zeros(12, np.float32))
    assert np.all(s.bottom_profile() == 2.0)
    s128 = OracleSim(nx=128, obs=(8, 64))                # blends are hit for Nx >= 128
    a = np.zeros(12, np.float32); a[3] = 1
    s128.set_action(a)
    tb = s128.bottom_profile()
    assert len(np.unique(np.round(tb, 12))) > 3


def test_random_initial_state():
    """initialize_model (rbc_sim2D.jl:163-171) with the build's own counter-based RNG."""
    s = OracleSim()
    s.reset_random(1234)
    b, u, w = s.fields()
    z = (np.arange(64) + 0.5) * DZ
    lin = 1 + (2 - z) / 2
    assert b.min() >= 1.0 and b.max() <= 2.0
    assert np.abs(b.mean(1) - lin)[2:-2].max() < 5e-3
    assert 0.008 < u.std() < 0.012 and s.max_divergence() < 1e-13
    s2 = OracleSim(); s2.reset_random(1234)
    assert all(np.array_equal(a, c) for a, c in zip(s.fields(), s2.fields()))
    s3 = OracleSim(); s3.reset_random(1235)
    assert not np.array_equal(s3.fields()[0], b)
    n = np.array([oracle_py.lib().rbco_normal(7, 0, i) for i in range(20000)])
    assert abs(n.mean()) < 0.03 and abs(n.std() - 1) < 0.03


def test_control_interval_substeps():
    """run! advances by aligned steps: 50 x 0.03 for heater_duration 1.5; 33 x 0.03 + 0.01 for 1.0."""
    a = OracleSim(dt_control=1.0); a.reset_random(3)
    b = OracleSim(dt_control=1.0); b.reset_from_arrays(*a.fields())
    act = np.zeros(12, np.float32)
    a.step(act)
    b.set_action(act); b.update_state()
    for _ in range(33):
        b.substep(0.03)
    b.substep(1.0 - 33 * 0.03)
    for x, y in zip(a.fields(), b.fields()):      # b went through one more (idempotent up to round-off) projection
        assert np.abs(x - y).max() < 1e-12


def test_observation_layout(ckpt_ra1e4):
    """get_state / get_observation (rbc_sim2D_api.jl:102-129) + rbc2D.py:184-196 transposes."""
    s = OracleSim(ra=1e4)
    b0, u0, w0 = ckpt_ra1e4["b"][0], ckpt_ra1e4["u"][0], ckpt_ra1e4["w"][0]
    s.reset_from_arrays(b0, u0, w0)
    st = s.state(5)
    assert st.shape == (5, 64, 96) and st.dtype == np.float32
    b, u, w = s.fields()
    assert np.array_equal(st[0], b.astype(np.float32)) and np.array_equal(st[2], w[:64].astype(np.float32))
    ob = s.obs_f32(3)
    assert ob.shape == (3, 8, 48)
    assert np.array_equal(ob, st[:3, 0:64:8, 0:96:2])
    assert np.all(st[2, 0] == 0)                      # w channel starts at the wall face
    # pHY'[top cell] = -min_b*dz ; pHY' decreases downward by b_face*dz
    assert np.allclose(st[3, 63], -1.0 * DZ, atol=1e-7)
