"""The reference's example/run_checkpoint.py: episodes start from converged states stored in ckpt_ra<Ra>.h5.
If the file is not there it is generated on the GPU first (a short spin-up here; scripts/create_checkpoints_2D.py
writes the full t = 600 train/test/val sets)."""
import logging
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rbc-gym_amd"))
import rbc_gym  # noqa: E402,F401
from rbc_gym._gym import gym  # noqa: E402
from rbc_gym.generate import generate_checkpoints_2d  # noqa: E402

logging.basicConfig(format="%(levelname)s:%(message)s", level=logging.INFO)
path = sys.argv[1] if len(sys.argv) > 1 else "data/checkpoints/train/ckpt_ra10000.h5"
if not os.path.exists(path):
    path = generate_checkpoints_2d(os.path.dirname(path) or ".", ra=1e4, random_inits=4, seed=42,
                                   duration=float(os.environ.get("RBC_SPINUP", "60")))
env = gym.make("rbc_gym/RayleighBenardConvection2D-v0", checkpoint=path, render_mode="rgb_array")
obs, info = env.reset()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else env.unwrapped.episode_steps
for step in range(steps):
    observation, reward, terminated, truncated, info = env.step(env.action_space.sample())
    env.render()
    if truncated:
        break
print(f"t={info['t']:.2f} step={info['step']} Nu_state={info['nusselt_state']:.4f}")
env.close()
