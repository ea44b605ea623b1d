#!/usr/bin/env python3
"""Device-side counterpart of the reference's scripts/create_checkpoints_3D.sh: writes
<dir>/{train,test,val}/3D_ckpt_ra<Ra>.h5 (20/10/10 episodes, seeds 42/62/72, 32x32x16, Pr 0.7, plates 1/2, t = 200)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rbc-gym_amd"))
from rbc_gym.generate import generate_checkpoints_3d  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ra", type=float, default=2500)
ap.add_argument("--dir", default="data/checkpoints")
ap.add_argument("--duration", type=float, default=200.0)
ap.add_argument("--device", type=int, default=0)
a = ap.parse_args()
for split, seed, n in (("train", 42, 20), ("test", 62, 10), ("val", 72, 10)):
    p = generate_checkpoints_3d(os.path.join(a.dir, split), ra=a.ra, random_inits=n, seed=seed, duration=a.duration, device=a.device,
                                progress=lambda s, t: print(f"  {split}: {s}/{t} intervals", flush=True))
    print("Saved data to:", p)
