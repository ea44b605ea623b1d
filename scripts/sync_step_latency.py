"""Where a SYNCHRONOUS 3D env-step (gym semantics: actions in, wait, results out) loses time against the back-to-back device rate:
host time inside rbc_step_dev (graph launch), time to completion, and the same with the graph off (RBC_USE_GRAPH=0) / one group."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
from rbc_gym import _native
B = 32
sim = _native.NativeSim3D(batch=B, shape=(32, 48, 48), ra=1e4, dt_control=0.125, dt_solver=0.01, random_kick=0.1, precision=sys.argv[1] if len(sys.argv) > 1 else "f64")
sim.reset(np.arange(B, dtype=np.uint64))
act = np.random.default_rng(0).uniform(-1, 1, (B, 8, 8)).astype(np.float32)
for _ in range(6):
    sim.step(act)
lib, h = sim.lib, sim.h
import torch
dact = torch.from_numpy(act).cuda()
torch.cuda.synchronize()
dptr = dact.data_ptr()
# synchronous: launch, then wait
tl, tw = [], []
for _ in range(20):
    t0 = time.perf_counter()
    if dptr: sim.step_dev(dptr)
    else: lib.rbc_step(h, act.ctypes.data_as(_native._fp))
    t1 = time.perf_counter()
    lib.rbc_synchronize(h)
    t2 = time.perf_counter()
    tl.append(t1 - t0); tw.append(t2 - t0)
print(f"synchronous: call returns after {np.median(tl) * 1e3:.3f} ms, complete after {np.median(tw) * 1e3:.3f} ms (dev actions: {bool(dptr)})")
# back to back
if dptr:
    lib.rbc_synchronize(h)
    t0 = time.perf_counter()
    for _ in range(20):
        sim.step_dev(dptr)
    lib.rbc_synchronize(h)
    print(f"back to back: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step")
t0 = time.perf_counter()
for _ in range(20):
    sim.step(act)
print(f"rbc_step (host actions, flags back): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per step")
