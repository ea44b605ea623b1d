#!/usr/bin/env python3
"""The 3D ORACLE itself at the reference's flow-statistics protocol (32x64x64, heater_duration 0.25, dt_solver 0.005, zero action;
flowstats_ra.py:27-36): MEMBERS independent runs of STEPS env-steps at one Rayleigh number on the CPU, Nu after every env-step
-> tests/golden/oracle3d_flowstats_ra<Ra>.json.  Closes the loop oracle <-> linear theory <-> GPU ensemble <-> reference series
directly (the GPU <-> oracle parity tests are per-step at 1e-10).  Minutes of CPU time: run by hand, the recorded numbers are
asserted by tests/test_flowstats_theory.py.

    python tests/golden/oracle3d_flowstats.py [ra=16000] [members=8] [steps=7]
"""
import json
import os
import sys
import time
from multiprocessing import get_context

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def member(args):
    ra, seed, steps = args
    import oracle_py
    o = oracle_py.Oracle3D(ra=ra, shape=(32, 64, 64), dt_control=0.25, dt_solver=0.005)
    o.reset_random(seed)
    out = []
    for _ in range(steps):
        assert o.step(None)
        out.append(o.nusselt())
    return out


if __name__ == "__main__":
    ra = float(sys.argv[1]) if len(sys.argv) > 1 else 16000.0
    members = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    import oracle_py
    oracle_py.build_oracle()
    t0 = time.time()
    with get_context("fork").Pool(min(members, os.cpu_count() or 1)) as pool:
        nus = pool.map(member, [(ra, 4242 + m, steps) for m in range(members)])
    nus = np.array(nus)
    out = {"ra": ra, "members": members, "steps": steps, "seeds": [4242 + m for m in range(members)], "nusselt": nus.tolist(),
           "cpu_seconds_wall": time.time() - t0}
    with open(os.path.join(HERE, f"oracle3d_flowstats_ra{int(ra)}.json"), "w") as f:
        json.dump(out, f, indent=1)
    la = np.log(nus - 1)
    print("mean log(Nu-1):", la.mean(0), "increments:", np.diff(la.mean(0)), f"({time.time() - t0:.0f} s)")
