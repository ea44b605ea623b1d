"""diagnostic: where does a grid leave the oracle?  reset (RNG + projection), projection alone, tendencies, one solver step"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rbc_gym import _native
import oracle_py
oracle_py.build_oracle()
def rel(a, b): return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))
def div(u, w, dx, dz): return float(np.abs((np.roll(u, -1, 1) - u) / dx + (w[1:] - w[:-1]) / dz).max())
for nx, nz in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]:
    cfg = dict(nx=nx, nz=nz, heaters=4, dt_solver=0.01, dt_control=0.03, ra=2e4)
    obs = (2, 1)
    dx, dz = 2 * np.pi / nx, 2.0 / nz
    sim = _native.NativeSim(batch=1, obs_nz=obs[0], obs_nx=obs[1], random_kick=0.05, **cfg)
    sim.reset(np.array([7], np.uint64))
    o = oracle_py.OracleSim(obs=obs, kick=0.05, **cfg); o.reset_random(7)
    b, u, w = sim.get_fields(); ob, ou, ow = o.fields()
    print(nx, nz, "random reset", [f"{rel(a[0], c):.1e}" for a, c in zip((b, u, w), (ob, ou, ow))], f"div gpu {div(u[0], w[0], dx, dz):.1e} oracle {div(ou, ow, dx, dz):.1e}")
    rng = np.random.default_rng(3)
    b0 = 1.5 + 0.1 * rng.standard_normal((nz, nx)); u0 = 0.1 * rng.standard_normal((nz, nx)); w0 = 0.1 * rng.standard_normal((nz + 1, nx)); w0[0] = w0[-1] = 0
    sim.reset_from_arrays(b0[None], u0[None], w0[None]); o.reset_from_arrays(b0, u0, w0)
    b, u, w = sim.get_fields(); ob, ou, ow = o.fields()
    print("   projection of given arrays", [f"{rel(a[0], c):.1e}" for a, c in zip((b, u, w), (ob, ou, ow))], f"div gpu {div(u[0], w[0], dx, dz):.1e} oracle {div(ou, ow, dx, dz):.1e}")
    sim.reset_from_arrays(ob[None], ou[None], ow[None])
    b, u, w = sim.get_fields()
    print("   projection of the oracle's projected state (idempotence)", [f"{rel(a[0], c):.1e}" for a, c in zip((b, u, w), (ob, ou, ow))])
    act = np.random.default_rng(0).uniform(-1, 1, (1, cfg["heaters"])).astype(np.float32)
    g = sim.debug_tendencies(act)
    o.set_action(act[0]); o.update_state(); go = o.tendencies()
    print("   tendencies", {k: f"{np.abs(g[k][0] - go[k]).max() / max(np.abs(go[k]).max(), 1e-30):.1e}" for k in go})
    sim.debug_substeps(act, 1, cfg["dt_solver"]); o.substep(cfg["dt_solver"])
    b, u, w = sim.get_fields(); ob, ou, ow = o.fields()
    print("   one solver step", [f"{rel(a[0], c):.1e}" for a, c in zip((b, u, w), (ob, ou, ow))], f"div gpu {div(u[0], w[0], dx, dz):.1e} oracle {div(ou, ow, dx, dz):.1e}")
    sim.close()
