"""CPU-side account of the reference's time-resolved data (experiments/flowstats/flowstats_ra.pkl -> tests/golden/
flowstats_ref_series.npz): linear theory of the restated discretisation (tests/linear_theory3d.py, no free parameter)
against (a) the native stepper's recorded ensembles and (b) the reference's series.  DESIGN.md section 4 has the prose."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "scripts"))
from linear_theory3d import LinearRBC3D  # noqa: E402

GOLD = os.path.join(HERE, "golden")
RAS = (8000.0, 16000.0)            # linear for >= 8 env-steps, growth well resolved on the 32x64x64 grid
STEPS = 9


@pytest.fixture(scope="module")
def data():
    ref = np.load(os.path.join(GOLD, "flowstats_ref_series.npz"))
    mom = np.load(os.path.join(GOLD, "flowstats3d_moments.npz"))
    th = {}
    for ra in RAS:
        lin = LinearRBC3D(ra)
        th[ra] = {"50": np.log(lin.nusselt_series(STEPS, 0.02, 50)),
                  "50then49": np.log(lin.nusselt_series(STEPS, 0.02, 49, nsub_first=50)),
                  "sigma_max": lin.sigma_max}
    return ref, mom, th


def test_build_follows_the_linear_theory_of_the_discretisation(data):
    """The recorded 16-member ensembles of the native stepper grow, env-step by env-step, exactly as the discretisation's
    linear theory says 50 RK3 substeps of 0.02 must: increments of log(Nu-1) within 0.4 % (recorded 0.05-0.3 %), level
    within 2 member standard errors.  This pins the BUILD's clock, linear operator, initial-condition statistics and
    Nusselt definition with no reference data involved."""
    ref, mom, th = data
    for ra in RAS:
        i = int(np.argmin(np.abs(mom["ra"] - ra)))
        m, s = mom["clock50_nusselt_mean"][i, :STEPS], mom["clock50_nusselt_sd"][i, :STEPS]
        t = th[ra]["50"]
        se = s / np.sqrt(float(mom["clock50_members"]))
        assert np.all(np.abs(t - m) < 3.0 * se + 0.005), (ra, t - m, se)
        dt_, dm = np.diff(t)[2:7], np.diff(m)[2:7]                       # env-steps 4..8: Nu-1 < 0.05, member spread < 1.3 %
        assert np.all(np.abs(dt_ / dm - 1.0) < 0.004), (ra, dt_ / dm)
        # the same theory with 49 substeps after the first env-step is 2 % away: the ensemble rejects it
        d49 = np.diff(th[ra]["50then49"])[2:7]
        assert np.all(dm / d49 - 1.0 > 0.015), (ra, dm / d49)
        m49 = mom["clock50then49_nusselt_mean"][i, :STEPS]
        assert np.all(np.abs(np.diff(m49)[2:7] / d49 - 1.0) < 0.004), (ra, np.diff(m49)[2:7] / d49)


def test_reference_series_grows_like_49_solver_steps_per_env_step(data):
    """What the reference's own series does against the same theory: the first env-step agrees with the documented
    clock (level within the member spread), every later one carries the growth of 49 solver steps, not 50 -- already
    at Nu-1 ~ 1e-4 (so not an amplitude / resolution effect), at the same 2 % in every env-step (so clock-like)."""
    ref, mom, th = data
    for ra in RAS:
        i = int(np.argmin(np.abs(ref["ra"] - ra)))
        r = np.log(ref["nusselt"][i, :STEPS] - 1.0)
        s = mom["clock50_nusselt_sd"][i, :STEPS]                         # spread of ONE realisation's level
        g = np.maximum(mom["clock50_nusselt_growth3_sd"][i, :STEPS], 1e-9)   # ... and of its growth since env-step 3
        assert abs(r[0] - th[ra]["50"][0]) < 3.0 * s[0]
        z50 = ((r - r[2]) - (th[ra]["50"] - th[ra]["50"][2])) / g
        z49 = ((r - r[2]) - (th[ra]["50then49"] - th[ra]["50then49"][2])) / g
        # one realisation's growth over six env-steps scatters by 4 %, the missing 6 x 2 % are 2-3 of those spreads; the
        # 11-sigma statement is the slope over nine Rayleigh numbers in test_recorded_time_offset_curves
        assert z50[-1] < -1.8 and np.all(np.diff(z50[3:]) < 0.0), (ra, z50)       # falls behind the documented clock, step after step
        assert np.all(np.abs(z49[3:]) < 1.5), (ra, z49)
        ratio = np.diff(r)[2:7] / np.diff(th[ra]["50"])[2:7]
        assert np.all((ratio > 0.965) & (ratio < 0.992)), (ra, ratio)


def test_recorded_time_offset_curves():
    """tau(n) of scripts/flowstats3d_tau.py on the recorded ensembles (tests/golden/flowstats3d_tau_*.json), all four
    series of flowstats_ra.py:55-66, Ra >= 4000: against the documented clock the Nusselt curve is a straight line of
    slope 0.9825 +- 0.0015 through tau(1) = 1.007 +- 0.010; with 49 solver steps after the first env-step the slope is
    1.0008 +- 0.0015.  The three velocity maxima (single-cell extrema of one realisation: ten times noisier) give
    0.985 / 0.969 / 0.975 +- 0.013: the same line, no separate information."""
    def load(name):
        with open(os.path.join(GOLD, f"flowstats3d_tau_{name}.json")) as f:
            return json.load(f)["summary"]
    base, fix, all49 = load("clock50"), load("clock50then49"), load("clock49")
    assert abs(base["nusselt"]["slope_mean"] - 0.9825) < 0.002 and base["nusselt"]["slope_sem"] < 0.002
    assert abs(base["nusselt"]["tau1_mean"] - 1.0) < 2.0 * base["nusselt"]["tau1_sem"] + 0.005
    assert abs(fix["nusselt"]["slope_mean"] - 1.0) < 2.0 * fix["nusselt"]["slope_sem"]
    assert abs(all49["nusselt"]["slope_mean"] - 1.0) < 2.0 * all49["nusselt"]["slope_sem"]
    assert all49["nusselt"]["tau1_mean"] - 1.0 > 4.0 * all49["nusselt"]["tau1_sem"]      # the FIRST env-step is not short
    vel = np.mean([base[k]["slope_mean"] for k in ("umax", "vmax", "wmax")])
    vel_sem = np.sqrt(sum(base[k]["slope_sem"] ** 2 for k in ("umax", "vmax", "wmax"))) / 3.0
    assert abs(vel - base["nusselt"]["slope_mean"]) < 2.0 * vel_sem and 1.0 - vel > 2.0 * vel_sem, (vel, vel_sem)


def test_oracle_itself_follows_the_theory_at_the_flowstats_protocol(data):
    """The 3D ORACLE run directly at the reference's protocol (tests/golden/oracle3d_flowstats.py: 8 members, 8 env-steps at
    Ra = 16000, three minutes of CPU time; recorded in oracle3d_flowstats_ra16000.json): its increments of log(Nu-1) are the
    theory's for 50 solver steps per env-step (1 %, mean 0.5 %) -- and not the reference's, which lie 2 % lower."""
    ref, mom, th = data
    with open(os.path.join(GOLD, "oracle3d_flowstats_ra16000.json")) as f:
        rec = json.load(f)
    nu = np.array(rec["nusselt"])
    assert nu.shape == (8, 8) and rec["ra"] == 16000.0
    la = np.log(nu - 1.0)
    inc = np.diff(la.mean(0))[2:7]
    t = np.diff(th[16000.0]["50"])[2:7]
    assert np.all(np.abs(inc / t - 1.0) < 0.01) and abs(np.mean(inc / t) - 1.0) < 0.005, inc / t
    # level after the first env-step: within 2.5 % (both ensembles, GPU and oracle, sit 1-1.7 % below the theory's level there: the
    # theory treats the clamp of the wall cells' noise through its variance only; the increments are what pins the clock)
    assert abs(la[:, 0].mean() - th[16000.0]["50"][0]) < 0.025
    i = int(np.argmin(np.abs(ref["ra"] - 16000.0)))
    r = np.diff(np.log(ref["nusselt"][i, :8] - 1.0))[2:7]
    assert np.all(r / inc < 0.99) and np.mean(r / inc) < 0.982


def test_linear_theory_reproduces_the_classical_onset():
    """Independent check of the theory tool, hence of the restated linear operator (C-grid Laplacians, buoyancy averaged to the w
    faces, no-slip / fixed-temperature ghost cells, exact projection): the discrete operator on the protocol's 32 x 64 x 64 grid
    goes unstable at Ra = 212.4 in the reference's units, i.e. 8 x 212.4 = 1699 in the classical ones (nu, kappa assume
    H = 1 but H = 2), at |k| = 1.576 -- Rayleigh-Benard convection between rigid plates: 1707.76 at k H = 3.117 (k = 1.5585);
    0.5 % below, the second-order error of 32 cells across the layer."""
    from linear_theory3d import critical_rayleigh
    rc, kt2 = critical_rayleigh()
    assert abs(8.0 * rc / 1707.76 - 1.0) < 0.008, rc
    assert abs(np.sqrt(kt2) - 1.5585) < 0.06, kt2
