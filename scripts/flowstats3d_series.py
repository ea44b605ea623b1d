#!/usr/bin/env python3
"""Per-step series of the reference's flow-statistics protocol on the native 3D stepper, SEEDS members per Rayleigh
number in one batch: Nu(t), max|u|, max|v|, max|w| of the float32 state after every env.step (what
experiments/flowstats/flowstats_ra.py:55-66 records), for the first STEPS steps.  Compared by tests/test_gpu_parity3d.py
and scripts/flowstats3d_compare.py with the reference's own series (tests/golden/flowstats_ref_series.npz).

    python scripts/flowstats3d_series.py [seeds=16] [steps=100] [out=gpurun_out/flowstats3d_series.npz] [dt_control=0.25] [dt_solver=0.005] [lead_substeps=0] [clock=documented|recorded]
Needs an MI355X.  clock=recorded runs the product switch `reference_clock="recorded"` (rbc_config.reference_clock: all solver
steps in the first env-step after a reset, one less in every later one).  A library built with -DRBC_EXPERIMENTS=1 reads
RBC_EXPERIMENT_RK3 (rbc3d_host_body.hpp): deliberately wrong RK3 coefficients for the "does the pin discriminate" experiment.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
from rbc_gym import _native  # noqa: E402


def shape_tff(shape, lz=2.0):
    """free-fall time the 3D API scales its steps with (t_ff = Lz^2, rbc_sim3D_api.jl:43)"""
    return lz * lz


def run_series(ras, seeds, steps, seed0=777, shape=(32, 64, 64), dt_control=0.25, dt_solver=0.005, progress=None, lead_substeps=0,
               reference_clock="documented"):
    """-> dict of arrays [len(ras), seeds, steps]: nusselt, umax, vmax, wmax (flowstats_ra.py:27-36 protocol, zero action)."""
    import torch
    from rbc_gym.vector import DeviceArray
    ras = np.asarray(ras, dtype=np.float64)
    R, B = len(ras), len(ras) * seeds
    sim = _native.NativeSim3D(batch=B, shape=shape, dt_control=dt_control, dt_solver=dt_solver, reference_clock=reference_clock)
    sim.set_rayleigh(np.repeat(ras, seeds))                       # env index = ra_index * seeds + member
    sim.reset(np.arange(B, dtype=np.uint64) + np.uint64(seed0))
    nz, ny, nx = shape
    state = torch.as_tensor(DeviceArray(sim.lib.rbc_dev_state(sim.h), (B, 4, nz * ny * nx), "<f4", sim), device="cuda")
    zero = np.zeros((B, 8, 8), np.float32)
    if lead_substeps:                      # experiment: extra solver steps in front of the first env-step (clock hypotheses)
        sim.debug_substeps(zero, int(lead_substeps), dt_solver * shape_tff(shape))
    out = {k: np.zeros((R, seeds, steps)) for k in ("nusselt", "umax", "vmax", "wmax")}
    t0 = time.time()
    for n in range(steps):
        if not sim.step(zero):
            raise RuntimeError(f"NaN at step {n}")
        out["nusselt"][:, :, n] = sim.get_nusselt().reshape(R, seeds)
        mx = state.abs().amax(dim=2).cpu().numpy().astype(np.float64)          # (B, 4): b, u, v, w
        for q, k in ((1, "umax"), (2, "vmax"), (3, "wmax")):
            out[k][:, :, n] = mx[:, q].reshape(R, seeds)
        if progress and n % 10 == 9:
            progress(f"step {n + 1}/{steps} ({time.time() - t0:.0f} s)")
    sim.close()
    return out


if __name__ == "__main__":
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "flowstats3d_series.npz")
    dt_control = float(sys.argv[4]) if len(sys.argv) > 4 else 0.25        # experiments only: the protocol's values are the defaults
    dt_solver = float(sys.argv[5]) if len(sys.argv) > 5 else 0.005
    lead = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    clock = sys.argv[7] if len(sys.argv) > 7 else "documented"
    ref = np.load(os.path.join(ROOT, "tests", "golden", "flowstats_ref_series.npz"))
    out = run_series(ref["ra"], seeds, steps, dt_control=dt_control, dt_solver=dt_solver, progress=lambda s: print(s, flush=True), lead_substeps=lead,
                     reference_clock=clock)
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    np.savez_compressed(dst, ra=ref["ra"], **out)
    for i, ra in enumerate(ref["ra"]):
        nu = out["nusselt"][i]
        print(f"Ra={ra:>9.0f}  Nu-1 at steps 1,2,3,10: " + " ".join(
            f"{nu[:, n].mean() - 1:.3e}+-{nu[:, n].std(ddof=1) / np.sqrt(seeds):.1e} (ref {ref['nusselt'][i, n] - 1:.3e})" for n in (0, 1, 2, 9)))
