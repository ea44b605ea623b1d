#!/usr/bin/env python3
"""Does the reference's 2D data carry the run!-re-entry time loss?  

The 2D checkpoints are single snapshots, but at Ra = 1e4 every episode is a from-rest run caught at nominal t = 600 while it is still
ringing down onto the steady state: the distance from the fixed point decays like exp(-sigma t), so its distribution over episodes is a
CLOCK.  The generator re-enters `run!` every 10 solver steps (rbc_sim2D.jl:189-194): if each re-entry loses one solver step of time, as the
3D series show, the episodes were caught at an effective t = 540.  This script measures the kinetic-energy scatter of the k = 2 members
of a from-rest ensemble (generator parameters: kick 0.02, dt 0.03) as a function of time, under both clocks, next to the scatter of the
reference's 40 episodes (tests/golden/ckpt2d_pins.json).   python scripts/steady_clock_probe.py [members=2048]
"""
import json, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
from rbc_gym import _native  # noqa: E402

KE_INF = 0.0983448


def series(n, clock, marks, seed0=4242):
    """-> {t_nominal: KE of every member}; clock "documented": env-steps of 1.5 (50 solver steps);
    "generator": env-steps of 0.3 under reference_clock="recorded" (10 solver steps in the first, 9 in every later one)"""
    kw = dict(batch=n, random_kick=0.02, write_state=0)
    if clock == "generator":
        kw.update(dt_control=0.3, reference_clock="recorded")
    sim = _native.NativeSim(**kw)
    dtc = sim.cfg.dt_control
    sim.reset(np.arange(n, dtype=np.uint64) + seed0)
    zero = np.zeros((n, 12), np.float32)
    out, t_end = {}, max(marks)
    want = {int(round(t / dtc)): t for t in marks}
    for k in range(1, int(round(t_end / dtc)) + 1):
        assert sim.step(zero)
        if k in want:
            _, u, w = sim.get_fields()
            out[want[k]] = 0.5 * ((u ** 2).mean((1, 2)) + (w[:, :-1] ** 2).mean((1, 2)))
    sim.close()
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "ckpt2d_pins.json")))
    ref = np.array([e["ke"] for sp in ("train", "val", "test") for e in pins[f"{sp}/ckpt_ra10000"]["episodes"]])
    rs = ref.std(ddof=1)
    print(f"reference: 40 episodes, KE {ref.mean():.9f}, std {rs:.3e} (+- {rs / np.sqrt(2 * 39):.1e}), |KE - mean| median {np.median(np.abs(ref - ref.mean())):.3e}")
    marks = [420.0, 480.0, 540.0, 570.0, 600.0, 630.0, 660.0, 720.0]
    for clock in ("documented", "generator"):
        res = series(n, clock, marks)
        on = np.abs(res[600.0] - KE_INF) < 1e-4                   # the k = 2 members (the others sit on the k = 1 state)
        print(f"{clock} clock, {on.sum()} k=2 members of {n}:")
        for t in marks:
            k = res[t][on]
            s = k.std(ddof=1)
            print(f"   nominal t = {t:5.0f}: KE {k.mean():.9f}  std {s:.3e}  (reference std / this = {rs / s:.2f})")
