#!/usr/bin/env python3
"""Diagnostic: where a tile workgroup of the 3D tendency kernel spends its level loop (build with -DRBC_STAMPS=1 into a separate
.so; the shipped library never executes a stamp).  Wave 0 of every workgroup accumulates s_memtime ticks per phase.
    python scripts/tile_stamps.py --build        (here: cross-compiles rbc-gym_amd/lib/librbc_hip_stamps.so)
    python scripts/tile_stamps.py [f64|f32]      (on the GPU box; `2d` : the FLAT tiles of the streaming-2D path at 128 x 64, B = 1024)
"""
import ctypes as C
import os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
SO = os.path.join(ROOT, "rbc-gym_amd", "lib", "librbc_hip_stamps.so")
NAMES = ["shift windows + issue next level's loads", "first field (u | w)", "second field (v | b)", "barrier 1 (all reads of the planes done)",
         "plane stores to LDS (waits for the prefetched planes)", "barrier 2", "loop back-edge"]

if __name__ == "__main__":
    if "--build" in sys.argv:
        sys.path.insert(0, ROOT)
        import __graft_entry__ as ge                       # the library's own two-unit build, with the stamp hooks compiled in
        ge.build_hip(force=True, extra_flags=("-DRBC_STAMPS=1",), out=SO)
        sys.exit(0)
    os.environ["RBC_HIP_LIB"] = SO
    os.environ.setdefault("RBC_3D_GROUPS", "1")
    from rbc_gym import _native
    prec = 1 if "f32" in sys.argv else 0
    if "2d" in sys.argv:
        B = 1024
        sim = _native.NativeSim(batch=B, nx=128, nz=64, obs_nx=64, obs_nz=8, precision=prec)
        act = np.random.default_rng(0).uniform(-1, 1, (B, 12)).astype(np.float32)
    else:
        B = 32
        sim = _native.NativeSim3D(batch=B, shape=(32, 48, 48), ra=1e4, precision=prec)
        act = np.random.default_rng(0).uniform(-1, 1, (B, 8, 8)).astype(np.float32)
    sim.lib.rbc_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    sim.reset(np.arange(B, dtype=np.uint64) + 1234)
    for _ in range(3):
        sim.step(act)
    st = np.zeros((B, 64), np.uint64)
    sim.lib.rbc_debug_stamps(sim.h, st.ctypes.data_as(C.POINTER(C.c_uint64)))
    flat = st.ravel().astype(np.float64)
    for body, name in ((0, "(u, v) body"), (1, "(w, b) body")):
        acc, n = flat[16 * body:16 * body + 7], flat[16 * body + 15]
        print(f"{name}: {int(n)} workgroups, {acc.sum() / n:.0f} ticks per workgroup in the level loop")
        for a, nm in zip(acc, NAMES):
            print(f"   {100 * a / acc.sum():5.1f} %  {a / n:9.0f}  {nm}")
