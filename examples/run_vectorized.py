"""Vectorised rollout as in the reference's example/run_vectorized.py; here `make_vec` returns ONE batched env
whose envs all advance in a single kernel launch (no worker processes), so the call works for 6 or 6000 envs."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rbc-gym_amd"))
import rbc_gym  # noqa: E402,F401
from rbc_gym._gym import gym  # noqa: E402


def main(num_envs=6, steps=100):
    env = gym.make_vec("rbc_gym/RayleighBenardConvection2D-v0", num_envs=num_envs, vectorization_mode="async",
                       vector_kwargs={"copy": True, "daemon": True}, render_mode="rgb_array")
    obs, info = env.reset()
    print(f"Observation shape: {obs.shape}")
    t0 = time.perf_counter()
    for _ in range(steps):
        action = env.action_space.sample()
        observation, reward, terminated, truncated, info = env.step(action)
        if truncated.any():
            break
    dt = time.perf_counter() - t0
    frames = env.render()
    print(f"{num_envs} envs x {steps} steps in {dt:.2f} s = {num_envs * steps / dt:.0f} env-steps/s; "
          f"mean reward {reward.mean():.4f}; {len(frames)} frames of shape {frames[0].shape}")
    env.close()


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:3]))
