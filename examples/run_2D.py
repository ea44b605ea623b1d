"""Single 2D env, zero action, as the reference's example/run_2D.py drives it (pressure channels, full-grid
observations); frames are rendered off-screen (rgb_array) because the GPU boxes have no display."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rbc-gym_amd"))
import rbc_gym  # noqa: E402,F401
from rbc_gym._gym import gym  # noqa: E402  (gymnasium when installed)

env = gym.make("rbc_gym/RayleighBenardConvection2D-v0", render_mode="rgb_array", pressure=True, observation_shape=[64, 96])
obs, info = env.reset()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else env.unwrapped.episode_steps
for step in range(steps):
    action = env.action_space.sample() * 0
    observation, reward, terminated, truncated, info = env.step(action)
    frame = env.render()
    if step % 20 == 0:
        print(f"step {info['step']:4d}  t={info['t']:7.2f}  Nu_state={info['nusselt_state']:.4f}  reward={reward:.4f}  frame {frame.shape}")
    if truncated:
        break
env.close()
