"""Randomised HIP <-> oracle parity over configurations nobody picked by hand (GPU box; the oracle is the checker).

Every grid the C ABI accepts (2D: >= 8 x 8 cells; 3D: >= 8 cells per direction) has to give the oracle's answer through whichever
kernels the host picks for it -- resident or streaming, packed or generic z solve, fast or generic DFT, tiled / marching / cell-per-
thread tendencies --, for any Rayleigh / Prandtl number, domain, plate temperatures, heater count and limit, sensor grid, solver and
control step (ragged last substep included) and either clock.  Draws N2 2D and N3 3D configurations from one seed, runs a random
reset and two actuated control intervals on both sides and prints the worst relative L2 difference of the fields per configuration.

    python scripts/fuzz_parity.py [seed] [n2d] [n3d] [f64|f32]          exit code 1 if any configuration exceeds the bars of the parity tests
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from rbc_gym import _native  # noqa: E402
import oracle_py  # noqa: E402


# (fields rel. L2, Nusselt rel., float32 observations abs.): float64 = the bars of the parity tests; float32 against the float64 oracle =
# round-off of the solver's own precision over two short intervals from a kicked state (worst of 195 draws: 1.6e-5 / 1.3e-5 / 9.5e-7),
# far from the O(1) of a wrong index or a wrong kernel choice, which is what this sweep is after
BARS = {"f64": (1e-10, 1e-7, 2e-5), "f32": (1e-4, 1e-4, 1e-5)}


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def divisors(n, lo=1):
    return [d for d in range(lo, n + 1) if n % d == 0]


def draw_2d(rng):
    pool_x = [8, 12, 16, 17, 20, 24, 30, 32, 36, 40, 48, 50, 56, 60, 64, 72, 80, 90, 96, 100, 112, 120, 128, 144, 160, 192, 200, 224, 256, 320]
    nx = int(rng.choice(pool_x)) if rng.random() < 0.8 else int(rng.integers(8, 200))
    nz = int(rng.choice([8, 9, 12, 16, 20, 24, 27, 32, 36, 40, 48, 56, 64, 80, 96])) if rng.random() < 0.8 else int(rng.integers(8, 100))
    if rng.random() < 0.25:                                          # the grids with an LDS-resident kernel in either precision (rbc_api.hip bind_grid)
        nx, nz = [(96, 64), (96, 48), (96, 32), (64, 64), (64, 48), (64, 32), (128, 32), (128, 64), (192, 32)][int(rng.integers(0, 9))]
    heaters = int(rng.integers(1, min(nx // 2, 32) + 1))
    obs_nz = int(rng.choice(divisors(nz, 2)))
    obs_nx = int(rng.choice(divisors(nx)))
    lx = float(rng.choice([2 * np.pi, 4.0, 3 * np.pi, 4 * np.pi]))
    lz = float(rng.choice([2.0, 1.0, 1.5]))
    dx, dz = lx / nx, lz / nz
    ra = float(10 ** rng.uniform(3.3, 5.5))
    pr = float(rng.choice([0.7, 1.0, 0.5]))
    kap = 1.0 / np.sqrt(pr * ra)
    nu = np.sqrt(pr / ra)
    # explicit diffusion: 4 max(nu, kappa) dt (1/dx^2 + 1/dz^2) < 2.51 (RK3); keep half of it, and an advective CFL below ~0.5 at |u| ~ 0.3
    dt_max = min(0.5 * 2.51 / (4 * max(nu, kap) * (1 / dx ** 2 + 1 / dz ** 2)), 0.5 * min(dx, dz) / 0.3, 0.03)
    dt = float(np.round(dt_max * rng.uniform(0.5, 1.0), 4)) or 1e-4
    nsub = int(rng.integers(2, 6))
    dtc = nsub * dt + (float(rng.choice([0.0, 0.37, 0.5])) * dt)          # a clipped last substep in two draws of three
    cfg = dict(nx=nx, nz=nz, heaters=heaters, heater_limit=float(rng.choice([0.3, 0.6, 0.75, 0.9])), dt_solver=dt, dt_control=float(dtc),
               ra=ra, pr=pr, lx=lx, lz=lz, min_b=float(rng.choice([1.0, 0.5])), delta_b=float(rng.choice([1.0, 2.0])))
    return cfg, (obs_nz, obs_nx), str(rng.choice(["documented", "recorded"]))


def clones_equal(fields, U):
    """envs e >= U carry the seed and the actions of env e % U: bitwise the same state, whatever env group / chain / tile shape ran them"""
    return all(np.array_equal(f[e], f[e % U]) for f in fields for e in range(U, f.shape[0]))


def run_2d(cfg, obs, clock, seed, precision="f64", B=3):
    """the first U = 3 envs are checked against the oracle; a larger batch (other env groups, chain counts and tile shapes: the host
    picks them from the batch size) repeats them and must reproduce them bit for bit"""
    U = min(B, 3)
    sim = _native.NativeSim(batch=B, obs_nz=obs[0], obs_nx=obs[1], random_kick=0.05, reference_clock=clock, precision=_native.PRECISIONS[precision], **cfg)
    seeds = (np.arange(B, dtype=np.uint64) % np.uint64(U)) + np.uint64(seed)
    sim.reset(seeds)
    orcs = []
    for e in range(U):
        o = oracle_py.OracleSim(obs=obs, kick=0.05, **cfg)
        o.reset_random(int(seeds[e]))
        orcs.append(o)
    worst = 0.0
    for x, o in zip(zip(*sim.get_fields()), orcs):
        worst = max(worst, max(rel_l2(a, b) for a, b in zip(x, o.fields())))
    rng = np.random.default_rng(seed)
    dt, dtc = cfg["dt_solver"], cfg["dt_control"]
    nfull = int(np.floor(dtc / dt + 1e-9))
    rem = dtc - nfull * dt
    worst_obs = worst_nu = 0.0
    for n in range(2):
        act = rng.uniform(-1.5, 1.5, (U, cfg["heaters"])).astype(np.float32)[np.arange(B) % U]
        assert sim.step(act), "NaN flag"
        for e, o in enumerate(orcs):
            if clock == "recorded" and n > 0:                      # one full solver step less in every env-step but the first
                o.set_action(act[e]); o.update_state()
                for _ in range(nfull - 1):
                    o.substep(dt)
                if rem > 1e-12:
                    o.substep(rem)
            else:
                assert o.step(act[e])
        f = sim.get_fields()
        nus, nuo = sim.get_nusselt()
        ob = sim.get_obs(5)
        assert clones_equal(f, U) and clones_equal((ob, nus, nuo), U), "batch members with the same seed and actions differ"
        for e, o in enumerate(orcs):
            worst = max(worst, max(rel_l2(a[e], b) for a, b in zip(f, o.fields())))
            worst_nu = max(worst_nu, abs(nus[e] - o.nusselt(True)) / max(1.0, abs(o.nusselt(True))), abs(nuo[e] - o.nusselt(False)) / max(1.0, abs(o.nusselt(False))))
            worst_obs = max(worst_obs, float(np.abs(ob[e][:4] - o.obs_f32(5)[:4]).max()))
    sim.close()
    return worst, worst_nu, worst_obs


def draw_batch(rng, cfg):
    """batch sizes on both sides of the host's thresholds (env groups of >= 4 envs, three 2D chains from 768 envs, tile shapes by workgroup count)"""
    if "shape" in cfg:
        return int(rng.choice([2, 5, 17, 40]))
    return int(rng.choice([3, 3, 8, 70, 800 if cfg["nx"] * cfg["nz"] <= 16384 else 70]))


def draw_3d(rng):
    pool = [8, 9, 10, 12, 16, 18, 20, 24, 28, 32, 36, 40, 48, 56, 64]
    nx, ny = int(rng.choice(pool)), int(rng.choice(pool))
    nz = int(rng.choice([8, 9, 12, 16, 20, 24, 32]))
    if rng.random() < 0.3:
        nx = ny = int(rng.choice([16, 32, 48, 64]))
    heaters = int(rng.choice([h for h in (1, 2, 3, 4, 6, 8) if h <= min(nx, ny) // 2]))
    lx, ly, lz = float(rng.choice([4 * np.pi, 2 * np.pi, 6.0])), float(rng.choice([4 * np.pi, 3 * np.pi, 5.0])), 2.0
    ra = float(10 ** rng.uniform(3.0, 4.6))
    dxm = min(lx / nx, ly / ny, lz / nz)
    kap = 1.0 / np.sqrt(0.7 * ra)
    dt_model = min(0.5 * 2.51 / (4 * kap * (nx * nx / lx ** 2 + ny * ny / ly ** 2 + nz * nz / lz ** 2)), 0.5 * dxm / 0.3, 0.04)
    dts = float(np.round(dt_model / 4.0 * rng.uniform(0.5, 1.0), 5)) or 1e-5          # in free-fall units (x t_ff = 4, rbc_sim3D_api.jl:43)
    nsub = int(rng.integers(2, 5))
    dtc = nsub * dts + float(rng.choice([0.0, 0.5])) * dts
    cfg = dict(shape=(nz, ny, nx), domain=(lz, ly, lx), ra=ra, heaters=heaters, heater_limit=float(rng.choice([0.5, 0.9])),
               dt_solver=dts, dt_control=float(dtc))
    return cfg, str(rng.choice(["documented", "recorded"]))


def run_3d(cfg, clock, seed, precision="f64", B=2):
    U = min(B, 2)
    sim = _native.NativeSim3D(batch=B, random_kick=0.1, reference_clock=clock, precision=precision, **cfg)
    seeds = (np.arange(B, dtype=np.uint64) % np.uint64(U)) + np.uint64(seed)
    sim.reset(seeds)
    orcs = []
    for e in range(U):
        o = oracle_py.Oracle3D(kick=0.1, **cfg)
        o.reset_random(int(seeds[e]))
        orcs.append(o)
    worst = 0.0
    f = sim.get_fields()
    for e, o in enumerate(orcs):
        worst = max(worst, max(rel_l2(a[e], b) for a, b in zip(f, o.fields())))
    rng = np.random.default_rng(seed)
    dts, dtc = cfg["dt_solver"], cfg["dt_control"]
    nfull = int(np.floor(dtc / dts + 1e-9))
    rem = dtc - nfull * dts
    H = cfg["heaters"]
    worst_nu = 0.0
    for n in range(2):
        act = rng.uniform(-1, 1, (U, H, H)).astype(np.float32)[np.arange(B) % U]
        assert sim.step(act), "NaN flag"
        for e, o in enumerate(orcs):
            if clock == "recorded" and n > 0:
                o.set_action(act[e]); o.update_state()
                for _ in range(nfull - 1):
                    o.substep(dts * 4.0)
                if rem > 1e-12:
                    o.substep(rem * 4.0)
            else:
                assert o.step(act[e])
        f = sim.get_fields()
        nu = sim.get_nusselt()
        assert clones_equal(f, U) and clones_equal((nu,), U), "batch members with the same seed and actions differ"
        for e, o in enumerate(orcs):
            worst = max(worst, max(rel_l2(a[e], b) for a, b in zip(f, o.fields())))
            worst_nu = max(worst_nu, abs(nu[e] - o.nusselt()) / max(1.0, abs(o.nusselt())))
    sim.close()
    return worst, worst_nu


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n2 = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    n3 = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    prec = sys.argv[4] if len(sys.argv) > 4 else "f64"
    bar_f, bar_nu, bar_obs = BARS[prec]
    oracle_py.build_oracle()
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for n in range(n2):
        cfg, obs, clock = draw_2d(rng)
        tag = f"2D {cfg['nx']}x{cfg['nz']} heaters={cfg['heaters']} obs={obs} ra={cfg['ra']:.3g} pr={cfg['pr']} dt={cfg['dt_solver']} dtc={cfg['dt_control']:.5f} l=({cfg['lx']:.3f},{cfg['lz']}) b=({cfg['min_b']},{cfg['delta_b']}) {clock}"
        try:
            B = draw_batch(rng, cfg)
            tag += f" B={B}"
            w, wn, wo = run_2d(cfg, obs, clock, seed * 1000 + n, prec, B)
            ok = w < bar_f and wn < bar_nu and wo < bar_obs
            print(f"{'ok ' if ok else 'BAD'} {tag}: fields {w:.2e} nusselt {wn:.2e} obs {wo:.2e}", flush=True)
        except Exception as exc:                                   # a refusal is a finding too: every drawn grid is inside the documented bounds
            ok = False
            print(f"ERR {tag}: {type(exc).__name__}: {exc}", flush=True)
        bad += not ok
    for n in range(n3):
        cfg, clock = draw_3d(rng)
        tag = f"3D {cfg['shape']} heaters={cfg['heaters']} ra={cfg['ra']:.3g} dt={cfg['dt_solver']} dtc={cfg['dt_control']:.5f} domain=({cfg['domain'][2]:.3f},{cfg['domain'][1]:.3f}) {clock}"
        try:
            B = draw_batch(rng, cfg)
            tag += f" B={B}"
            w, wn = run_3d(cfg, clock, seed * 1000 + 500 + n, prec, B)
            ok = w < bar_f and wn < bar_nu
            print(f"{'ok ' if ok else 'BAD'} {tag}: fields {w:.2e} nusselt {wn:.2e}", flush=True)
        except Exception as exc:
            ok = False
            print(f"ERR {tag}: {type(exc).__name__}: {exc}", flush=True)
        bad += not ok
    print(f"{n2 + n3} configurations, {bad} outside the bars, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
