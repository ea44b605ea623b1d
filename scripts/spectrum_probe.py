#!/usr/bin/env python3
"""x power spectra (row-averaged) of b', u, w of an ensemble at the generator's protocol against the reference's 40 episodes
(tests/golden/ckpt2d_ra{Ra}_profiles.npz: xspec), wavenumber by wavenumber: where a difference in the grid-scale dissipation of the
advection scheme would show.   python scripts/spectrum_probe.py [ra=1e7] [members=1024]      (needs an MI355X)"""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
from rbc_gym import _native  # noqa: E402

if __name__ == "__main__":
    ra = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    ref = np.load(os.path.join(ROOT, "tests", "golden", f"ckpt2d_ra{int(ra)}_profiles.npz"))["xspec"]        # (40, 3, 49)
    sim = _native.NativeSim(batch=n, ra=ra, random_kick=0.02, write_state=0)
    sim.reset(np.arange(n, dtype=np.uint64) + 31)
    zero = np.zeros((n, 12), np.float32)
    for _ in range(400):
        assert sim.step(zero)
    b, u, w = sim.get_fields()
    sim.close()
    mine = np.stack([(np.abs(np.fft.rfft(f - f.mean(2, keepdims=True), axis=2)) ** 2).mean(1) / f.shape[2] ** 2 for f in (b, u, w[:, :-1])], axis=1)   # (n, 3, 49)
    rm, me = ref.mean(0), mine.mean(0)
    se = np.sqrt(ref.var(0, ddof=1) / len(ref) + mine.var(0, ddof=1) / n)
    print(f"Ra = {ra:.0e}, {n} members vs 40 episodes: ratio mine / reference of the row-averaged power per wavenumber (z in brackets)")
    for name, q in (("b'", 0), ("u", 1), ("w", 2)):
        ks = [1, 2, 3, 4, 6, 8, 12, 16, 20, 24, 28, 32, 36, 40, 44, 48]
        print(f"  {name:2s} " + "  ".join(f"k{k}: {me[q, k] / rm[q, k]:.3f} ({(me[q, k] - rm[q, k]) / se[q, k]:+.1f})" for k in ks))
        for lo, hi in ((1, 3), (3, 9), (9, 17), (17, 33), (33, 49)):
            a, r = mine[:, q, lo:hi].sum(1), ref[:, q, lo:hi].sum(1)
            z = (a.mean() - r.mean()) / np.hypot(a.std(ddof=1) / np.sqrt(n), r.std(ddof=1) / np.sqrt(len(r)))
            print(f"       band k {lo}..{hi - 1}: {a.mean() / r.mean():.3f} (z {z:+.1f})", end="")
        print()
