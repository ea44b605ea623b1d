"""a few SYNCHRONOUS 3D env-steps for a kernel trace (rocprofv3 --kernel-trace): when does each chain's first kernel start?"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
from rbc_gym import _native
B = 32
sim = _native.NativeSim3D(batch=B, shape=(32, 48, 48), ra=1e4, dt_control=0.125, dt_solver=0.01, random_kick=0.1)
sim.reset(np.arange(B, dtype=np.uint64))
act = np.random.default_rng(0).uniform(-1, 1, (B, 8, 8)).astype(np.float32)
for _ in range(4):
    sim.step(act)
    time.sleep(0.02)
