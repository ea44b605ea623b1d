#!/usr/bin/env python3
"""P4 with error bars: the reference's flowstats protocol (32x64x64, heater_duration 0.25, dt_solver 0.005, zero action,
300 steps; experiments/flowstats/flowstats_ra.py:27-36) for SEEDS independent initialisations per Rayleigh number in one
batch; prints mean +- spread over seeds of the last-100-step mean Nusselt next to the reference's single realisation.
    python scripts/flowstats3d_ensemble.py [seeds]     (needs an MI355X; 8 seeds x 14 Ra = 112 envs, about a minute)"""
import json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
from rbc_gym import _native  # noqa: E402

REF = {500: 1.368, 750: 1.513, 1000: 1.497, 1500: 1.668, 2000: 1.762, 4000: 2.128, 8000: 2.411, 16000: 2.851,
       32000: 3.453, 64000: 4.232, 128000: 5.233, 256000: 6.422, 512000: 7.886, 1000000: 9.212}

if __name__ == "__main__":
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ras = sorted(REF)
    B = seeds * len(ras)
    sim = _native.NativeSim3D(batch=B, shape=(32, 64, 64), dt_control=0.25, dt_solver=0.005)
    sim.set_rayleigh(np.tile(np.array(ras, dtype=np.float64), seeds))
    sim.reset(np.arange(B, dtype=np.uint64) + 777)
    zero = np.zeros((B, 8, 8), np.float32)
    nus, t0 = [], time.time()
    for n in range(300):
        assert sim.step(zero), f"NaN at step {n}"
        nus.append(sim.get_nusselt().copy())
        if n % 50 == 49:
            print(f"step {n + 1} ({time.time() - t0:.0f} s)", flush=True)
    m = np.array(nus)[200:].mean(0).reshape(seeds, len(ras))
    out = {}
    for j, ra in enumerate(ras):
        out[str(ra)] = {"mean": float(m[:, j].mean()), "std_over_seeds": float(m[:, j].std(ddof=1)), "min": float(m[:, j].min()),
                        "max": float(m[:, j].max()), "reference": REF[ra], "rel_diff_of_mean": float(m[:, j].mean() / REF[ra] - 1)}
        print(f"Ra={ra:>8d}  Nu = {m[:, j].mean():.4f} +- {m[:, j].std(ddof=1):.4f}  [{m[:, j].min():.4f}, {m[:, j].max():.4f}]   reference {REF[ra]:.3f}"
              f"  ({100 * (m[:, j].mean() / REF[ra] - 1):+.2f} %)")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"seeds": seeds, "results": out}, open(os.path.join(ROOT, "gpurun_out", "flowstats3d_ensemble.json"), "w"), indent=1)
