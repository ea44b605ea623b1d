"""diagnostic: a 3D batch whose envs e >= 2 repeat env e % 2 -- which env, which quantity, after which call do they part?"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
from rbc_gym import _native
shape = tuple(int(v) for v in sys.argv[1].split("x")); B = int(sys.argv[2]); prec = sys.argv[3] if len(sys.argv) > 3 else "f64"
sim = _native.NativeSim3D(batch=B, shape=shape, ra=8300.0, heaters=3, dt_solver=0.0092, dt_control=0.0414, random_kick=0.1, precision=prec)
seeds = (np.arange(B, dtype=np.uint64) % np.uint64(2)) + np.uint64(77)
sim.reset(seeds)
def report(tag):
    f = sim.get_fields(); nu = sim.get_nusselt()
    bad = {}
    for name, a in zip("buvw", f):
        for e in range(2, B):
            d = float(np.abs(a[e] - a[e % 2]).max())
            if d: bad.setdefault(name, []).append((e, d))
    dn = [(e, float(abs(nu[e] - nu[e % 2]))) for e in range(2, B) if nu[e] != nu[e % 2]]
    print(tag, "fields:", {k: v[:6] for k, v in bad.items()}, "nusselt:", dn[:8], flush=True)
report("reset")
rng = np.random.default_rng(0)
for n in range(3):
    act = rng.uniform(-1, 1, (2, 3, 3)).astype(np.float32)[np.arange(B) % 2]
    assert sim.step(act)
    report(f"step {n + 1}")
