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
