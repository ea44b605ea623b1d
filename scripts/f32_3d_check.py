#!/usr/bin/env python3
"""float32 against float64 on the streaming paths (3D configs[4] shape and a streaming-2D grid): agreement after two control
intervals from the same seeds, and the rate of both.  usage: python scripts/f32_3d_check.py [B] [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rbc-gym_amd"))
from rbc_gym import _native

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def rel(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def run3d(prec):
    sim = _native.NativeSim3D(batch=B, shape=(32, 48, 48), ra=1e4, precision=prec)
    sim.reset(np.arange(B, dtype=np.uint64) + 1234)
    act = np.random.default_rng(0).uniform(-1, 1, (B, 8, 8)).astype(np.float32)
    for _ in range(2):
        assert sim.step(act)
    f, nu = sim.get_fields(), sim.get_nusselt().copy()
    sim.step(act); sim.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sim.step(act)
    sim.synchronize()
    dt = (time.perf_counter() - t0) / steps
    nan = int(sim.get_flags().sum())
    sim.close()
    return f, nu, B / dt, nan


def run2d(prec, nx=128, nz=64):
    b2 = 1024
    sim = _native.NativeSim(batch=b2, nx=nx, nz=nz, obs_nx=nx // 2, obs_nz=8, precision=prec)
    sim.reset(np.arange(b2, dtype=np.uint64))
    act = np.random.default_rng(0).uniform(-1, 1, (b2, 12)).astype(np.float32)
    for _ in range(2):
        assert sim.step(act)
    f, nu = sim.get_fields(), sim.get_nusselt()[0].copy()
    sim.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        sim.step(act)
    sim.synchronize()
    dt = (time.perf_counter() - t0) / 3
    sim.close()
    return f, nu, b2 / dt


f64, nu64, r64, n64 = run3d(0)
f32, nu32, r32, n32 = run3d(1)
print(f"3D 32x48x48 B={B}: f64 {r64:.0f} env-steps/s (nan {n64}), f32 {r32:.0f} env-steps/s (nan {n32}), ratio {r32 / r64:.2f}")
for name, a, b in zip("buvw", f32, f64):
    print(f"   rel-L2 {name}: {rel(a, b):.2e}   max|{name}| {np.abs(b).max():.3e}")
print(f"   Nusselt: max rel diff {np.abs(nu32 / nu64 - 1).max():.2e}")
g64, m64, s64 = run2d(0)
g32, m32, s32 = run2d(1)
print(f"2D 128x64 streaming B=1024: f64 {s64:.0f}, f32 {s32:.0f} env-steps/s, ratio {s32 / s64:.2f}")
for name, a, b in zip("buw", g32, g64):
    print(f"   rel-L2 {name}: {rel(a, b):.2e}")
print(f"   Nusselt: max rel diff {np.abs(m32 / m64 - 1).max():.2e}")
