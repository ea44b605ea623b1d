#!/usr/bin/env python3
"""Device-side counterpart of the reference's scripts/create_checkpoints_2D.sh: writes
<dir>/{train,test,val}/ckpt_ra<Ra>.h5 (20/10/10 episodes, seeds 42/62/72, 96x64, kick 0.02, dt 0.03, t = 600)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rbc-gym_amd"))
from rbc_gym.generate import generate_checkpoints_2d  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("ra", type=float)
ap.add_argument("--dir", default="data/checkpoints")
ap.add_argument("--duration", type=float, default=600.0)
ap.add_argument("--device", type=int, default=0)
a = ap.parse_args()
for split, seed, n in (("train", 42, 20), ("test", 62, 10), ("val", 72, 10)):
    p = generate_checkpoints_2d(os.path.join(a.dir, split), ra=a.ra, random_inits=n, seed=seed, duration=a.duration,
                                device=a.device, progress=lambda s, t: print(f"  {split}: {s}/{t} intervals", flush=True))
    print("Saved data to:", p)
