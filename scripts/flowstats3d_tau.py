#!/usr/bin/env python3
"""Time-offset curves tau(n) of the reference's flow-statistics series against an ensemble of the native 3D stepper.

For each Rayleigh number and each of the four recorded series (flowstats_ra.py:55-66: Nu, max|u|, max|v|, max|w|),
tau(n) is the (interpolated) ensemble time at which the ensemble-mean log-amplitude equals the reference's value after
env-step n.  A stepper whose clock and operator equal the reference's gives tau(n) = n within the member spread; a clock
that runs a fixed fraction slow gives a straight line of that slope FROM n = 1 ON IN ALL FOUR SERIES; an
amplitude-dependent (resolution) effect gives tau(n) = n while the perturbation is small and a deficit that opens with
amplitude.  Pure numpy; works on the .npz files scripts/flowstats3d_series.py writes.

    python scripts/flowstats3d_tau.py gpurun_out/fs_series_lemoin.npz [out.json]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SERIES = ("nusselt", "wmax", "umax", "vmax")


def log_amplitude(name, x):
    return np.log(np.abs(x - 1.0)) if name == "nusselt" else np.log(np.abs(x))


def growth_window(mean_log):
    """[n0, n1] (0-based step indices): from the step after the initial decay's minimum to the last step whose
    increment still exceeds 40 % of the largest one (the exponential phase, before the overshoot bends the curve)."""
    d = np.diff(mean_log)
    n0 = int(np.argmin(mean_log[: max(3, np.argmax(d) + 1)]))
    peak = int(np.argmax(d))
    n1 = peak
    while n1 + 1 < len(d) and d[n1 + 1] > 0.4 * d[peak]:
        n1 += 1
    return n0 + 1, n1 + 1


def tau_curve(ref_log, ens_log):
    """-> (steps n (1-based), tau(n), sd of tau from the member spread) over the growth window of the ensemble mean."""
    m = ens_log.mean(0)
    s = ens_log.std(0, ddof=1)
    n0, n1 = growth_window(m)
    n = np.arange(1, len(m) + 1, dtype=float)
    xs = np.linspace(n[n0], n[n1], 4001)
    ys = np.interp(xs, n[n0:n1 + 1], m[n0:n1 + 1])          # log-linear between samples: exact for exponential growth
    steps, taus, sds = [], [], []
    for k in range(n0, n1 + 1):
        v = ref_log[k]
        if v < ys[0] or v > ys[-1]:
            continue
        j = int(np.searchsorted(ys, v))
        slope = (m[min(k + 1, n1)] - m[max(k - 1, n0)]) / (n[min(k + 1, n1)] - n[max(k - 1, n0)])
        steps.append(k + 1); taus.append(xs[min(j, len(xs) - 1)]); sds.append(s[k] / max(slope, 1e-9))
    return np.array(steps, float), np.array(taus), np.array(sds)


def analyse(ref, ens, ra_min=4000.0):
    rows = []
    for i, ra in enumerate(ref["ra"]):
        if ra < ra_min:
            continue
        row = {"ra": float(ra)}
        for name in SERIES:
            nsteps = ens[name].shape[2]
            steps, taus, sds = tau_curve(log_amplitude(name, ref[name][i, :nsteps]), log_amplitude(name, ens[name][i]))
            if len(steps) >= 3:
                w = 1.0 / np.maximum(sds, 1e-3) ** 2
                A = np.stack([np.ones_like(steps), steps - 1.0], axis=1)
                coef, *_ = np.linalg.lstsq(A * np.sqrt(w)[:, None], taus * np.sqrt(w), rcond=None)
                t1, slope = float(coef[0]), float(coef[1])          # tau(n) ~ t1 + slope * (n - 1)
            else:
                t1 = slope = float("nan")
            row[name] = {"steps": steps.tolist(), "tau": np.round(taus, 4).tolist(), "tau_sd": np.round(sds, 4).tolist(),
                         "tau_at_1": t1, "slope": slope}
        rows.append(row)
    return rows


def summary(rows):
    out = {}
    for name in SERIES:
        sl = np.array([r[name]["slope"] for r in rows]); t1 = np.array([r[name]["tau_at_1"] for r in rows])
        ok = np.isfinite(sl)
        out[name] = {"slope_mean": float(sl[ok].mean()), "slope_sem": float(sl[ok].std(ddof=1) / np.sqrt(ok.sum())),
                     "tau1_mean": float(t1[ok].mean()), "tau1_sem": float(t1[ok].std(ddof=1) / np.sqrt(ok.sum())), "n_ra": int(ok.sum())}
    return out


if __name__ == "__main__":
    ens = np.load(sys.argv[1])
    ref = np.load(os.path.join(ROOT, "tests", "golden", "flowstats_ref_series.npz"))
    rows = analyse(ref, ens)
    for r in rows:
        print(f"Ra={r['ra']:9.0f} " + "  ".join(f"{k}: tau(1)={r[k]['tau_at_1']:+.3f} slope={r[k]['slope']:.4f} (n={len(r[k]['steps'])})" for k in SERIES))
    s = summary(rows)
    for k in SERIES:
        print(f"{k:8s} slope {s[k]['slope_mean']:.4f} +- {s[k]['slope_sem']:.4f}   tau(1) {s[k]['tau1_mean']:.3f} +- {s[k]['tau1_sem']:.3f}   ({s[k]['n_ra']} Ra)")
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            json.dump({"source": os.path.basename(sys.argv[1]), "rows": rows, "summary": s}, f, indent=1)
