"""The reference's example/run_3D.py without the pyvista window: 3D env, zero action, prints the Nusselt number."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rbc-gym_amd"))
import rbc_gym  # noqa: E402,F401
from rbc_gym._gym import gym  # noqa: E402

env = gym.make("rbc_gym/RayleighBenardConvection3D-v0", rayleigh_number=2500, heater_duration=0.25,
               episode_length=int(sys.argv[1]) if len(sys.argv) > 1 else 50)
obs, info = env.reset()
while True:
    action = env.action_space.sample() * 0
    observation, reward, terminated, truncated, info = env.step(action)
    if info["step"] % 20 == 0 or truncated:
        print(f"t={info['t']:7.2f} step={info['step']:4d} reward={reward:.5f} obs {observation.shape}")
    if truncated:
        break
env.close()
