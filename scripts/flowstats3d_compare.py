#!/usr/bin/env python3
"""Compare a series file written by scripts/flowstats3d_series.py with the reference's own series
(tests/golden/flowstats_ref_series.npz): early-time Nu-1, linear-phase growth and the best uniform time factor.
    python scripts/flowstats3d_compare.py gpurun_out/flowstats3d_series.npz [more.npz ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def linear_phase_end(mean_nu_minus_1, cap=0.3, least=4):
    n = int(np.argmax(mean_nu_minus_1 > cap)) if np.any(mean_nu_minus_1 > cap) else len(mean_nu_minus_1)
    return max(n, least)


def time_factor(ens_log_mean, ref_log, nl, grid=np.linspace(0.90, 1.10, 401)):
    """f minimising rms[ ref(n) - ens(f n) ] over steps 1..nl in log space (ens interpolated in time)"""
    t = np.arange(1, len(ens_log_mean) + 1, dtype=float)
    best = (np.inf, 1.0)
    for f in grid:
        err = np.sqrt(np.mean((ref_log[:nl] - np.interp(f * t[:nl], t, ens_log_mean)) ** 2))
        if err < best[0]:
            best = (err, f)
    return best[1], best[0]


def compare(path, ref):
    g = np.load(path)
    steps = g["nusselt"].shape[2]
    print(f"== {path}: {g['nusselt'].shape[1]} members, {steps} steps")
    fs = []
    for i, ra in enumerate(ref["ra"]):
        nu = g["nusselt"][i] - 1
        m, s = nu.mean(0), nu.std(0, ddof=1)
        r = ref["nusselt"][i, :steps] - 1
        nl = min(linear_phase_end(m), steps)
        z = (r[:nl] - m[:nl]) / s[:nl]
        f, res = time_factor(np.log(m), np.log(r), nl)
        res1 = np.sqrt(np.mean((np.log(r[:nl]) - np.log(m[:nl])) ** 2))
        fs.append(f)
        print(f"Ra={ra:>9.0f} linear steps 1..{nl:2d}: max|z| {np.abs(z).max():5.2f}  z(1..3) {z[0]:+.2f} {z[1]:+.2f} {z[2]:+.2f}   "
              f"rms log-resid {100 * res1:5.2f}% -> {100 * res:5.2f}% at time factor {f:.3f}")
    print(f"   median time factor {np.median(fs):.4f}  (Ra >= 4000: {np.median(fs[5:]):.4f})")


if __name__ == "__main__":
    ref = np.load(os.path.join(ROOT, "tests", "golden", "flowstats_ref_series.npz"))
    for p in sys.argv[1:]:
        compare(p, ref)


def time_map(ens_log_mean, ref_log, nl):
    """(F, a, rms) of the map  t_ref = a + F n  minimising rms[ ref(n) - ens(a + F n) ] over steps 1..nl (log space)"""
    t = np.arange(1, len(ens_log_mean) + 1, dtype=float)
    best = (np.inf, 1.0, 0.0)
    for F in np.linspace(0.94, 1.04, 201):
        for a in np.linspace(-0.1, 0.1, 41):
            err = np.sqrt(np.mean((ref_log[:nl] - np.interp(a + F * t[:nl], t, ens_log_mean)) ** 2))
            if err < best[0]:
                best = (err, F, a)
    return best[1], best[2], best[0]


def peak_time(y):
    """time of the maximum of a series sampled at steps 1, 2, ... (parabola through the three samples around it)"""
    k = int(np.argmax(y))
    if k == 0 or k == len(y) - 1:
        return float(k + 1)
    a, b, c = y[k - 1], y[k], y[k + 1]
    return k + 1 + 0.5 * (a - c) / (a - 2 * b + c)


def summary(path, ref):
    """numbers recorded in tests/golden/flowstats3d_series_experiments.json"""
    g = np.load(path)
    steps = g["nusselt"].shape[2]
    rows = []
    for i, ra in enumerate(ref["ra"]):
        nu = g["nusselt"][i] - 1
        m, s = nu.mean(0), nu.std(0, ddof=1)
        r = ref["nusselt"][i, :steps] - 1
        nl = min(linear_phase_end(m), steps)
        f, _ = time_factor(np.log(m), np.log(r), nl)
        npk = int(np.argmax(m))
        row = {"ra": float(ra), "z_steps_1_3": [float((r[n] - m[n]) / s[n]) for n in range(3)], "linear_steps": int(nl),
               "linear_time_factor": float(f)}
        if 3 < npk < steps - 1:
            F, a, rms = time_map(np.log(m), np.log(r), npk - 1)
            tp = np.array([peak_time(x) for x in g["nusselt"][i]])
            row.update({"prepeak_steps": int(npk - 1), "prepeak_time_map_F": float(F), "prepeak_time_map_a": float(a),
                        "prepeak_rms_log_resid": float(rms), "peak_time_members_mean": float(tp.mean()),
                        "peak_time_members_std": float(tp.std(ddof=1)), "peak_time_reference": float(peak_time(ref["nusselt"][i, :steps]))})
        rows.append(row)
    return rows
