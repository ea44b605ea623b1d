#!/usr/bin/env python3
"""Statistically steady regime of the flow-statistics protocol: last-100-step means of Nu, max|u|, max|v|, max|w| per Rayleigh
number on the native 3D stepper (SEEDS members each) against the reference's single realisation (flowstats_ra.py:55-66,
tests/golden/flowstats_ref_series.npz).   python scripts/flowstats3d_velstats.py [seeds=4] [precision=f64]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from flowstats3d_series import run_series
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ref = np.load(os.path.join(ROOT, "tests", "golden", "flowstats_ref_series.npz"))
out = run_series(ref["ra"], seeds, 300, seed0=99, progress=lambda s: print(s, flush=True))
rows = []
for i, ra in enumerate(ref["ra"]):
    row = {"ra": float(ra)}
    for k in ("nusselt", "umax", "vmax", "wmax"):
        mine = out[k][i, :, 200:].mean(1)                       # per member
        r = ref[k][i, 200:]
        row[k] = {"ref_mean": float(r.mean()), "ref_std": float(r.std()), "build_mean": float(mine.mean()), "build_member_std": float(mine.std(ddof=1)) if seeds > 1 else None,
                  "build_time_std": float(out[k][i, :, 200:].std(1).mean())}
    rows.append(row)
    print(f"Ra={ra:9.0f} " + "  ".join(f"{k}: ref {row[k]['ref_mean']:.4f}+-{row[k]['ref_std']:.3f} build {row[k]['build_mean']:.4f}" for k in ("nusselt", "umax", "vmax", "wmax")))
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "flowstats3d_velstats.json"), "w"), indent=1)
