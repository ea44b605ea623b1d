"""The reference's example/timing.py on this backend: construction, reset and step times of one env
(README.md:60-62 of the reference: 7.68 s Julia init, 0.12 s per step at dt=1 on one CPU core)."""
import os
import sys
import time
from timeit import timeit

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rbc-gym_amd"))
import rbc_gym  # noqa: E402,F401
from rbc_gym._gym import gym  # noqa: E402

t1 = time.perf_counter()
env = gym.make("rbc_gym/RayleighBenardConvection2D-v0", heater_duration=1, render_mode=None, use_gpu=True)
env.reset()
print(f"init time (library load, handle, first reset): {time.perf_counter() - t1:.2f} seconds")

iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reset_time = timeit("env.reset()", number=iterations, globals=globals())
print(f"Average time to reset env: {reset_time / iterations * 1e3:.3f} ms")


def step():
    env.step(env.action_space.sample())


step_time = timeit("step()", number=iterations, globals=globals())
print(f"Average time to step one timestep (dt=1): {step_time / iterations * 1e3:.3f} ms")
