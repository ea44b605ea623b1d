#!/usr/bin/env python3
"""configs[1] device-resident rate of the LDS-resident 2D kernel over 30 env-steps: python scripts/rate_2d.py [f64|f32] [B] [nx nz]  (RBC_HIP_LIB selects the build)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rbc-gym_amd"))
from rbc_gym import _native
import torch
prec = 1 if "f32" in sys.argv else 0
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
B = nums[0] if nums else 1024
nx, nz = (nums[1], nums[2]) if len(nums) >= 3 else (96, 64)
sim = _native.NativeSim(batch=B, precision=prec, write_state=1, nx=nx, nz=nz, obs_nx=nx // 2, obs_nz=8, ra=1e5 if nx > 96 else 1e4)
sim.reset(np.arange(B, dtype=np.uint64) + 1234)
g = torch.Generator(device="cuda"); g.manual_seed(1)
act = (torch.rand((B, 12), device="cuda", generator=g) * 2 - 1).contiguous(); torch.cuda.synchronize()
for _ in range(3): sim.step_dev(act.data_ptr())
sim.synchronize(); t0 = time.perf_counter()
for _ in range(30): sim.step_dev(act.data_ptr())
sim.synchronize(); dt = (time.perf_counter() - t0) / 30
print(f"{'f32' if prec else 'f64'} {nx}x{nz} B={B}: {B / dt:.0f} env-steps/s  ({dt * 1e3:.3f} ms per batched step)  mean Nu {sim.get_nusselt()[0].mean():.12f}")
