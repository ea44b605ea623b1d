#!/usr/bin/env python3
"""Whole-field pin of the 2D solver on the reference's 40 Ra=1e4 episodes: from-rest ensemble at the reference's checkpoint
protocol (random kick 0.02, dt 0.03, zero action, t = 600), k=2 members, z-scores of the translation-invariant spectra
(tests/spectral_invariants.py) against tests/golden/ckpt2d_ra10000_spectra.npz.
    python scripts/spectral_pin.py [members=256]                 (RBC_HIP_LIB=<other build> to test a variant)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rbc_gym import _native  # noqa: E402
from spectral_invariants import SPEC_K, spectral_z_scores  # noqa: E402

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    ref = np.load(os.path.join(ROOT, "tests", "golden", "ckpt2d_ra10000_spectra.npz"))
    sim = _native.NativeSim(batch=n, random_kick=0.02, write_state=0)
    sim.reset(np.arange(n, dtype=np.uint64) + 4242)
    zero = np.zeros((n, 12), np.float32)
    for _ in range(400):
        assert sim.step(zero)
    b, u, w = sim.get_fields()
    on = (np.abs(np.fft.rfft(w[:, 32], axis=1))[:, 1:].argmax(1) + 1) == 2
    zm, zp, mods, big = spectral_z_scores(b[on], u[on], w[on], ref)
    print(f"library {os.environ.get('RBC_HIP_LIB', 'default')}: {int(on.sum())} of {n} members on the k=2 state")
    print(f"moduli: {zm.size} z-scores, max |z| {np.abs(zm).max():.2f}, rms {np.sqrt(np.mean(zm ** 2)):.2f}")
    print(f"phases: {zp.size} z-scores (strong modes), max |z| {np.abs(zp).max():.2f}, rms {np.sqrt(np.mean(zp ** 2)):.2f}")
    rel = np.abs(mods - ref["mod_mean"])[big] / ref["mod_mean"][big]
    print(f"relative difference of the mean moduli: max {rel.max():.2e}, median {np.median(rel):.2e}  (k = {SPEC_K})")
