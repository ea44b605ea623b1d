#!/usr/bin/env python3
"""Pin P3 on the GPU path: the reference's checkpoint protocol (kick 0.02, dt 0.03, zero action, t = 600) at all seven
Rayleigh numbers it ships checkpoints for, 128 members each in one batch, against the per-Ra statistics of its 40
episodes (tests/golden/ckpt2d_pins.json).  python scripts/ensemble_ra_sweep.py  (needs an MI355X)"""
import json, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
from rbc_gym import _native  # noqa: E402

RAS = [10000, 30000, 100000, 300000, 1000000, 3000000, 10000000]


def reference_stats():
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "ckpt2d_pins.json")))
    out = {}
    for ra in RAS:
        eps = [e for sp in ("train", "val", "test") for e in pins[f"{sp}/ckpt_ra{ra}"]["episodes"]]
        out[ra] = {k: np.array([e[k] for e in eps]) for k in ("ke", "nusselt_state", "nusselt_obs", "umax", "wmax")}
    return out


def run(per=128, seed0=777, steps=400):
    B = per * len(RAS)
    sim = _native.NativeSim(batch=B, random_kick=0.02, write_state=0)
    sim.set_rayleigh(np.repeat(np.array(RAS, dtype=np.float64), per))
    sim.reset(np.arange(B, dtype=np.uint64) + seed0)
    zero = np.zeros((B, 12), np.float32)
    for _ in range(steps):
        if not sim.step(zero):
            raise RuntimeError(f"NaN envs: {np.nonzero(sim.get_flags())[0]}")
    b, u, w = sim.get_fields()
    ke = 0.5 * ((u ** 2).mean((1, 2)) + (w[:, :64] ** 2).mean((1, 2)))
    nus, nuo = sim.get_nusselt()
    res = {}
    for j, ra in enumerate(RAS):
        sl = slice(j * per, (j + 1) * per)
        res[ra] = {"ke": ke[sl], "nusselt_state": nus[sl], "nusselt_obs": nuo[sl],
                   "umax": np.abs(u[sl]).max((1, 2)), "wmax": np.abs(w[sl]).max((1, 2))}
    sim.close()
    return res


if __name__ == "__main__":
    ref, got = reference_stats(), run()
    for ra in RAS:
        line = [f"Ra={ra:>8d}"]
        for k in ("ke", "nusselt_state", "nusselt_obs", "umax", "wmax"):
            a, r = got[ra][k], ref[ra][k]
            sem = np.hypot(a.std(ddof=1) / np.sqrt(a.size), r.std(ddof=1) / np.sqrt(r.size))
            line.append(f"{k} {a.mean():.4f} vs {r.mean():.4f} (z={(a.mean() - r.mean()) / sem:+.1f})")
        print("  ".join(line))
        q = [0, .1, .25, .5, .75, .9, 1]
        print("      ke quantiles  mine", np.round(np.quantile(got[ra]["ke"], q), 4), " ref", np.round(np.quantile(ref[ra]["ke"], q), 4))
