"""gym-API rate of the 3D vector env at configs[4]: B x VectorEnv.step() per wall second, host actions in, host observations (the
float32 states, 38 MB per step) out; always-new arrays ("fresh") against the default (pooled page-locked arrays, reused once dropped) and unconditionally rotating page-locked buffers ("pinned"), both precisions."""
import sys, time, numpy as np
sys.path.insert(0, "rbc-gym_amd")
import rbc_gym
from rbc_gym._gym import gym
for prec, bufs in (("f64", "fresh"), ("f64", None), ("f64", "pinned"), ("f32", "fresh"), ("f32", None), ("f32", "pinned")):
    venv = gym.make_vec("rbc_gym/RayleighBenardConvection3D-v0", num_envs=32, state_shape=(32, 48, 48), rayleigh_number=10000, episode_length=10**9, precision=prec, obs_buffers=bufs)
    venv.reset(seed=1)
    acts = np.random.default_rng(0).uniform(-1, 1, (12, 32, 8, 8)).astype(np.float32)
    for n in range(8):                    # (pinned buffers: one graph per buffer and ping-pong parity is captured on first use)
        venv.step(acts[n])
    t0 = time.perf_counter()
    for n in range(10):
        venv.step(acts[n + 1])
    dt = (time.perf_counter() - t0) / 10
    print(prec, bufs, f"gym API: {32 / dt:.0f} env-steps/s ({dt * 1e3:.2f} ms per step of 32 envs)")
    # breakdown: sim only
    t0 = time.perf_counter()
    for n in range(10):
        venv.sim.step(acts[n + 1])
    dt2 = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for n in range(10):
        venv.sim.get_state()
    dt3 = (time.perf_counter() - t0) / 10
    print(f"   sim.step {dt2 * 1e3:.2f} ms, get_state {dt3 * 1e3:.2f} ms")
    venv.close()
