#!/usr/bin/env python3
"""Time the streaming 2D path (2D grids without an LDS-resident kernel) and, on the default grid, against the resident kernel.
usage: python scripts/stream2d_timing.py [B] [steps] [nx nz [precision]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rbc-gym_amd"))
from rbc_gym import _native

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def run(label, **kw):
    sim = _native.NativeSim(batch=B, **kw)
    sim.reset(np.arange(B, dtype=np.uint64))
    act = np.random.default_rng(0).uniform(-1, 1, (B, 12)).astype(np.float32)
    sim.step(act)
    sim.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step(act)
    sim.synchronize()
    dt = (time.perf_counter() - t0) / steps
    gb = sim.algorithmic_bytes_per_env_step() * B / dt / 1e9
    print(f"{label}: {dt * 1e3:.2f} ms per batched env-step, {B / dt:.0f} env-steps/s, {gb:.0f} GB/s algorithmic", flush=True)
    sim.close()


if len(sys.argv) > 4:            # one grid only (profiling): B steps nx nz
    nx, nz = int(sys.argv[3]), int(sys.argv[4])
    prec = sys.argv[5] if len(sys.argv) > 5 else "f64"
    run(f"{nx}x{nz} streaming {prec}", nx=nx, nz=nz, obs_nx=nx // 2, obs_nz=8, precision=_native.PRECISIONS[prec])
    sys.exit(0)
run("96x64 resident")
os.environ["RBC_FORCE_STREAM2D"] = "1"
run("96x64 streaming")
del os.environ["RBC_FORCE_STREAM2D"]
run("128x64 streaming", nx=128, nz=64, obs_nx=64, obs_nz=8)
run("192x32 streaming", nx=192, nz=32, obs_nx=48, obs_nz=8)
run("256x128 streaming", nx=256, nz=128, obs_nx=64, obs_nz=8)
