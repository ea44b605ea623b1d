#!/usr/bin/env python3
"""The reference's WHOLE recorded output against the native 3D stepper under both clocks: 14 Rayleigh numbers x 300 env-steps x 4 series
(experiments/flowstats/flowstats_ra.pkl via tests/golden/flowstats_ref_series.npz), 16 members per Ra.  z = (reference - ensemble mean)
/ member spread on the log-amplitudes; rms z per series over windows of the run.  Writes tests/golden/flowstats3d_full_window.json when
given `record`.   python scripts/flowstats3d_full_window.py [members=16] [record]     (needs an MI355X; ~3 min)"""
import json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "rbc-gym_amd"), os.path.join(ROOT, "scripts"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from flowstats3d_series import run_series  # noqa: E402
from flowstats3d_tau import SERIES, analyse, log_amplitude, summary  # noqa: E402

if __name__ == "__main__":
    members = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 16
    ref = np.load(os.path.join(ROOT, "tests", "golden", "flowstats_ref_series.npz"))
    steps = ref["nusselt"].shape[1]
    windows = [(0, 30), (30, 60), (60, 150), (150, steps), (0, steps)]
    rec = {"members": members, "steps": int(steps), "ra": [float(r) for r in ref["ra"]], "clocks": {}}
    for clock in ("recorded", "documented"):
        t0 = time.time()
        out = run_series(ref["ra"], members, steps, seed0=20250, reference_clock=clock, progress=lambda s: print(f"  {clock}: {s}", flush=True))
        z = {}
        for name in SERIES:
            la = log_amplitude(name, out[name])
            z[name] = (log_amplitude(name, ref[name]) - la.mean(1)) / (la.std(1, ddof=1) * np.sqrt(1.0 + 1.0 / members))
        tau = summary(analyse(ref, {k: v[:, :, :60] for k, v in out.items()}))["nusselt"]
        row = {"tau_slope": tau["slope_mean"], "tau_slope_sem": tau["slope_sem"], "seconds": time.time() - t0, "rms_z": {}, "frac_abs_z_below_3": {}}
        for a, b in windows:
            row["rms_z"][f"{a}-{b}"] = {name: float(np.sqrt((z[name][:, a:b] ** 2).mean())) for name in SERIES}
            row["frac_abs_z_below_3"][f"{a}-{b}"] = float(np.mean([(np.abs(z[name][:, a:b]) < 3).mean() for name in SERIES]))
        rec["clocks"][clock] = row
        print(f"{clock}: tau slope {tau['slope_mean']:.4f} +- {tau['slope_sem']:.4f}; rms z per window " +
              "; ".join(f"[{w}] " + " ".join(f"{k} {v:.2f}" for k, v in r.items()) for w, r in row["rms_z"].items()), flush=True)
    if "record" in sys.argv:
        with open(os.path.join(ROOT, "tests", "golden", "flowstats3d_full_window.json"), "w") as f:
            json.dump(rec, f, indent=1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "flowstats3d_full_window.json"), "w") as f:
        json.dump(rec, f, indent=1)
