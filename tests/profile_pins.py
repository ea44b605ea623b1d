"""Row-wise profile pins at the chaotic Rayleigh numbers (test infrastructure; numpy only).

The reference ships 40 independent episodes per Rayleigh number (data/checkpoints/{train,val,test}/ckpt_ra*.h5, written by
rbc_sim2D.jl:41-43,64-66 at t = 600).  tests/golden/ckpt2d_ra{Ra}_profiles.npz holds, per episode, the per-row means over x of
b, u^2, w^2, w b, b^2 (make_fixtures.py::row_moments).  At Ra >= 1e6 the thermal boundary layer is one to two cells thick, so
the rows next to the walls are decided by the order reduction 5 -> 3 -> 1 of the upwind stencils there (a
[RECALL] item): an ensemble of the build at the generator's protocol against these rows is the pin for it.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MOMENTS = ("b", "u2", "w2", "wb", "b2")
CHAOTIC_RAS = (30000, 100000, 300000, 1000000, 3000000, 10000000)


def row_moments_batch(b, u, w):
    """(B, nz, nx) fields (w with nz + 1 faces) -> (B, 5, nz): the same moments as make_fixtures.py::row_moments"""
    wc = w[:, :-1]
    return np.stack([b.mean(2), (u ** 2).mean(2), (wc ** 2).mean(2), (b * wc).mean(2), (b ** 2).mean(2)], axis=1)


def reference_profiles(ra):
    return np.load(os.path.join(GOLDEN, f"ckpt2d_ra{int(ra)}_profiles.npz"))["profiles"]      # (40, 5, nz)


def symmetrised(p):
    """The equations are symmetric under (z -> Lz - z, b -> 3 - b, w -> -w) and so are the statistics: fold the upper half onto
    the lower one (b -> 3 - b, <b^2> -> <(3 - b)^2>; u^2, w b unchanged; w^2 moves from face k to face nz - k, i.e. the cell
    ABOVE) to halve the noise of a profile.  p: (..., 5, nz) -> (..., 5, nz)"""
    q = p[..., ::-1].copy()
    q[..., 0, :] = 3.0 - p[..., 0, ::-1]
    q[..., 4, :] = 9.0 - 6.0 * p[..., 0, ::-1] + p[..., 4, ::-1]
    w2 = p[..., 2, :]
    q[..., 2, 1:] = w2[..., :0:-1]            # face k of the mirrored state = face nz - k of the original; face 0 is a wall
    q[..., 2, 0] = 0.0
    # <w b> at (cell k, face k): the mirror image pairs cell nz-1-k with face nz-k, the face ABOVE that cell -- not the same
    # statistic, so wb is left unfolded
    q[..., 3, :] = p[..., 3, :]
    return 0.5 * (p + q)


def profile_z(ens, ref):
    """z[moment][row] of the ensemble-mean rows against the reference's 40 episodes (Welch), and the relative difference"""
    me, mr = ens.mean(0), ref.mean(0)
    se = np.sqrt(ens.var(0, ddof=1) / len(ens) + ref.var(0, ddof=1) / len(ref))
    with np.errstate(divide="ignore", invalid="ignore"):
        z = np.where(se > 0, (me - mr) / se, 0.0)
        rel = np.where(np.abs(mr) > 0, me / mr - 1.0, 0.0)
    return z, rel


def summarise(z, skip_wall_w=True):
    """max |z| and rms z over the rows that carry information (w^2 and w b vanish identically at the wall face, row 0)"""
    zz = z.copy()
    if skip_wall_w:
        zz[2, 0] = 0.0; zz[3, 0] = 0.0
    n = zz.size - (2 if skip_wall_w else 0)
    return float(np.abs(zz).max()), float(np.sqrt((zz ** 2).sum() / n))
