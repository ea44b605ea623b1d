#!/usr/bin/env python3
"""Member moments of the native stepper's flow-statistics ensembles -> tests/golden/flowstats3d_moments.npz.

Input: the .npz files scripts/flowstats3d_series.py wrote on an MI355X (16 members per Rayleigh number, protocol of
experiments/flowstats/flowstats_ra.py:27-36): the documented clock (50 solver steps per env-step), 49 per env-step, and
50 in the first env-step followed by 49 in every later one.  Stored: mean and standard deviation over the members of the
log-amplitudes log(Nu-1), log max|u|, |v|, |w| for the first 30 env-steps (and the
standard deviation of the growth since env-step 3) -- what tests/test_flowstats_theory.py (no GPU)
compares with the linear theory of the discretisation and with the reference's series.

    python tests/golden/make_flowstats_moments.py gpurun_out/fs_series_lemoin.npz gpurun_out/fs_e1_49sub.npz gpurun_out/fs_e5_lead1_49.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ("clock50", "clock49", "clock50then49")
STEPS = 30

if __name__ == "__main__":
    out = {}
    for name, path in zip(NAMES, sys.argv[1:4]):
        e = np.load(path)
        out["ra"] = e["ra"]
        for k in ("nusselt", "umax", "vmax", "wmax"):
            x = e[k][:, :, :STEPS]
            la = np.log(np.abs(x - 1.0)) if k == "nusselt" else np.log(np.abs(x))
            out[f"{name}_{k}_mean"] = la.mean(1).astype(np.float64)
            out[f"{name}_{k}_sd"] = la.std(1, ddof=1).astype(np.float32)
            # growth since env-step 3 (index 2): the members' frozen-in level offsets drop out, the spread is that of the growth alone
            out[f"{name}_{k}_growth3_sd"] = (la - la[:, :, 2:3]).std(1, ddof=1).astype(np.float32)
        out[f"{name}_members"] = np.int64(e["nusselt"].shape[1])
    np.savez_compressed(os.path.join(HERE, "flowstats3d_moments.npz"), **out)
    print("wrote", os.path.join(HERE, "flowstats3d_moments.npz"))
