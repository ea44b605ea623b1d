#!/usr/bin/env python3
"""configs[4] device-resident rate over 100 env-steps: python scripts/rate_3d.py [f64|f32] [B] (env knobs apply: RBC_3D_GROUPS, ...)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rbc-gym_amd"))
from rbc_gym import _native
import torch
prec = 1 if "f32" in sys.argv else 0
nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
B = nums[0] if nums else 32
sim = _native.NativeSim3D(batch=B, shape=(32, 48, 48), ra=1e4, precision=prec)
sim.reset(np.arange(B, dtype=np.uint64) + 1234)
g = torch.Generator(device="cuda"); g.manual_seed(1)
act = (torch.rand((B, 8, 8), device="cuda", generator=g) * 2 - 1).contiguous(); torch.cuda.synchronize()
for _ in range(5): sim.step_dev(act.data_ptr())
sim.synchronize(); t0 = time.perf_counter()
for _ in range(100): sim.step_dev(act.data_ptr())
sim.synchronize(); dt = (time.perf_counter() - t0) / 100
print(f"{'f32' if prec else 'f64'} B={B}: {B / dt:.0f} env-steps/s  ({dt * 1e3:.3f} ms per env-step of the batch)  mean Nu {sim.get_nusselt().mean():.12f}")
