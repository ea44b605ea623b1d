#!/usr/bin/env python3
"""3D statistical pin P4: restates experiments/flowstats/flowstats_ra.py:27-36 on the
native 3D stepper (grid 32x64x64, heater_duration 0.25, dt_solver 0.005, zero action, 300 steps) and
compares the mean Nusselt number of the last 100 steps with the values the survey measured from the
reference's experiments/flowstats/flowstats_ra.pkl.  (That pickle cannot be loaded here:
torch.load(weights_only=True) refuses it and no other loader is allowed, so the numbers come from
SURVEY.md and the fit from the text output of flowstats_plots.ipynb.)

    python scripts/flowstats3d.py  -> tests/golden/flowstats3d_gpu.json   (needs an MI355X)
"""
import json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
from rbc_gym import _native  # noqa: E402

REF = {500: 1.368, 750: 1.513, 1000: 1.497, 1500: 1.668, 2000: 1.762, 4000: 2.128, 8000: 2.411, 16000: 2.851,
       32000: 3.453, 64000: 4.232, 128000: 5.233, 256000: 6.422, 512000: 7.886, 1000000: 9.212}

if __name__ == "__main__":
    ras = [int(a) for a in sys.argv[1:]] or [2000, 8000, 32000]
    B = len(ras)
    sim = _native.NativeSim3D(batch=B, shape=(32, 64, 64), heater_duration=None) if False else \
        _native.NativeSim3D(batch=B, shape=(32, 64, 64), dt_control=0.25, dt_solver=0.005)
    sim.set_rayleigh(np.array(ras, dtype=np.float64))
    sim.reset(np.arange(B, dtype=np.uint64) + 2024)
    zero = np.zeros((B, 8, 8), np.float32)
    nus = []
    t0 = time.time()
    for n in range(300):
        assert sim.step(zero), f"NaN at step {n}"
        nus.append(sim.get_nusselt().copy())
        if n % 50 == 49:
            print(f"step {n + 1}: Nu = {nus[-1]}  ({time.time() - t0:.1f} s)", flush=True)
    nus = np.array(nus)
    out = {}
    for j, ra in enumerate(ras):
        m = float(nus[200:, j].mean())
        out[str(ra)] = {"nu_first": float(nus[0, j]), "nu_last100_mean": m, "nu_last100_std": float(nus[200:, j].std()),
                        "reference_last100_mean": REF.get(ra), "rel_diff": (m - REF[ra]) / REF[ra] if ra in REF else None}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "flowstats3d_gpu.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
