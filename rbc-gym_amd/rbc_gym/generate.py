"""Checkpoint generation on the device: the batched counterpart of the reference's command-line generator
(`simulate` in rbc_sim2D.jl:20-70, driven by scripts/create_checkpoints_2D.sh:18-20).

The reference runs `random_inits` independent simulations one after another on the CPU (seed + i each), without
actuation, for `duration` time units and stores the final b,u,w of each as one episode of `ckpt_ra{Ra}.h5`.
Here all episodes are one batch on one GPU and the file is written by `checkpoint.write_checkpoint` in the same
on-disk format.  (Julia's RNG stream cannot be reproduced, so episode i is *a* random initialisation with seed
start_seed + i of this library's counter-based generator, not the reference's episode i.)"""
import os

import numpy as np

from . import _native
from .checkpoint import write_checkpoint


def generate_checkpoints_2d(directory, ra=1e4, random_inits=20, seed=42, n=(96, 64), min_b=1.0, random_kick=0.02,
                            delta_t=0.03, duration=600.0, chunk=1.5, device=0, progress=None):
    """-> path of the written file.  Arguments follow the Julia CLI (rbc_sim2D.jl:231-302): n = (Nx, Nz)."""
    nx, nz = int(n[0]), int(n[1])
    sim = _native.NativeSim(batch=int(random_inits), device=device, nx=nx, nz=nz, ra=float(ra), min_b=float(min_b),
                            random_kick=float(random_kick), dt_solver=float(delta_t), dt_control=float(chunk), write_state=0)
    try:
        sim.reset(np.arange(1, random_inits + 1, dtype=np.uint64) + np.uint64(seed))      # seed + i, i = 1..E
        zero = np.zeros((random_inits, sim.heaters), np.float32)
        steps = int(round(duration / chunk))
        for s in range(steps):
            if not sim.step(zero):
                raise RuntimeError("checkpoint generation: NaN values in the simulation")    # rbc_sim2D.jl:60-62
            if progress and (s + 1) % max(1, steps // 10) == 0:
                progress(s + 1, steps)
        b, u, w = sim.get_fields()
    finally:
        sim.close()
    os.makedirs(directory, exist_ok=True)
    ra_tag = int(ra) if float(ra).is_integer() else ra
    path = os.path.join(directory, f"ckpt_ra{ra_tag}.h5")
    write_checkpoint(path, b, u, w, start_seed=seed)
    return path


def generate_checkpoints_3d(directory, ra=2500, pr=0.7, random_inits=20, seed=42, n=(32, 32, 16), domain=(4 * np.pi, 4 * np.pi, 2.0),
                            b=(1.0, 2.0), random_kick=0.01, delta_t=0.01, delta_t_snap=0.25, duration=200.0, device=0, progress=None):
    """3D counterpart (`simulate_3d_rb`, rbc_sim3D.jl:13-95, driven by scripts/create_checkpoints_3D.sh): n and domain in
    the Julia CLI's (x, y, z) order, times in free-fall units (x t_ff = Lz^2); writes `3D_ckpt_ra{Ra}.h5` with datasets
    b, u, v, w.  -> path."""
    nx, ny, nz = (int(v) for v in n)
    lx, ly, lz = (float(v) for v in domain)
    sim = _native.NativeSim3D(batch=int(random_inits), device=device, shape=(nz, ny, nx), domain=(lz, ly, lx), ra=float(ra), pr=float(pr),
                              t_diff=(float(b[0]), float(b[1])), dt_control=float(delta_t_snap), dt_solver=float(delta_t),
                              random_kick=float(random_kick))
    try:
        sim.reset(np.arange(1, random_inits + 1, dtype=np.uint64) + np.uint64(seed))
        heaters = sim.heaters
        zero = np.zeros((random_inits, heaters, heaters), np.float32)
        steps = int(duration // (delta_t_snap * lz * lz))          # totalsteps = div(duration, dt_snap * t_ff)  (rbc_sim3D.jl:37)
        for s in range(steps):
            if not sim.step(zero):
                raise RuntimeError("checkpoint generation: NaN values in the simulation")
            if progress and (s + 1) % max(1, steps // 10) == 0:
                progress(s + 1, steps)
        fb, fu, fv, fw = sim.get_fields()
    finally:
        sim.close()
    os.makedirs(directory, exist_ok=True)
    ra_tag = int(ra) if float(ra).is_integer() else ra
    path = os.path.join(directory, f"3D_ckpt_ra{ra_tag}.h5")
    write_checkpoint(path, fb, fu, fw, start_seed=seed, v=fv)
    return path
