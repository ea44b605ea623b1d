#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-facing path: VectorEnv.step with host actions,
host observations/rewards (and optionally the float32 full state in info) vs the device-resident loop."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
import rbc_gym  # noqa
from rbc_gym._gym import gym

B = 1024
for info_state in (False, "fresh", True, "pinned"):
    env = gym.make_vec("rbc_gym/RayleighBenardConvection2D-v0", num_envs=B, info_state=info_state)
    env.reset(seed=0)
    a = np.random.default_rng(0).uniform(-1, 1, (B, 12)).astype(np.float32)
    for _ in range(2):
        env.step(a)
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        env.step(a)
    dt = (time.perf_counter() - t0) / n
    print(f"VectorEnv.step, host actions/obs/rewards, info_state={info_state}: {dt * 1e3:.2f} ms/step = {B / dt:.0f} env-steps/s")
    env.close()
