#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the step kernel (build with -DRBC_STAMPS=1 into a
separate .so; the shipped library never executes a stamp).  Run on the GPU box:
    python scripts/phase_stamps.py [batch]
"""
import ctypes as C
import os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
SO = os.path.join(ROOT, "rbc-gym_amd", "lib", "librbc_hip_stamps.so")
NAMES = ["setup+outputs", "prepass barrier wait", "u tend", "b tend", "stash+w tend", "sync+write U*", "rhs", "fft A", "fft B",
         "pack", "ifft B", "ifft A", "correct", "b write", "loop top", "prepass compute", "unpack", "thomas fwd", "thomas jct+bwd", "-", "-", "-", "-", "-"]


def build():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge                           # the library's own two-unit build, with the stamp hooks compiled in
    ge.build_hip(force=True, extra_flags=("-DRBC_STAMPS=1",), out=SO)


if __name__ == "__main__":
    if not os.path.exists(SO) or "--build" in sys.argv:
        build()
        if "--build" in sys.argv:
            sys.exit(0)
    os.environ["RBC_HIP_LIB"] = SO
    from rbc_gym import _native
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
    sim = _native.NativeSim(batch=B)
    sim.lib.rbc_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    sim.reset(np.arange(B, dtype=np.uint64) + 1234)
    act = np.random.default_rng(0).uniform(-1, 1, (B, 12)).astype(np.float32)
    for _ in range(2):
        sim.step(act)
    st = np.zeros((B, 64), np.uint64)
    sim.lib.rbc_debug_stamps(sim.h, st.ctypes.data_as(C.POINTER(C.c_uint64)))
    m0 = st[:, :24].astype(np.float64).mean(0)
    m1 = st[:, 32:56].astype(np.float64).mean(0)
    tot = m0.sum()
    print(f"mean cycles per env-step per workgroup: {tot:.0f}  ({tot / 150:.0f} per stage); first wave | last wave")
    for n, v, w in zip(NAMES, m0, m1):
        print(f"  {n:22s} {100 * v / tot:5.1f}%  {v / 150:8.0f} | {w / 150:8.0f} cyc/stage")
