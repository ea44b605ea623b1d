"""Randomised CALL SEQUENCES on the C ABI against per-env oracle mirrors (GPU box; the oracle is the checker).

scripts/fuzz_parity.py varies the configuration; this one varies what a caller does with a handle: per-env Rayleigh numbers, then a
random walk over  step(random actions) | masked random reset | masked reset from arrays | step with the zero action  under either
clock, on the LDS-resident 2D kernel, a streaming 2D grid and a 3D grid.  After every call: fields (rel. L2), Nusselt numbers, t / step
counters and the NaN flags of every env against its own oracle instance, which is driven by the same calls one env at a time (the
reference's process-per-env picture: rbc2D.py:124-182).  The recorded clock is mirrored env by env: all solver steps in the first env-step
after a reset of THAT env, one fewer in every later one (include/rbc_hip.h).

    python scripts/fuzz_sequences.py [seed] [sequences per target] [calls per sequence]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from rbc_gym import _native  # noqa: E402
import oracle_py  # noqa: E402


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


TARGETS = {
    "resident-2d": dict(dim=2, cfg=dict(nx=96, nz=64, heaters=12, dt_solver=0.03, dt_control=0.1), obs=(8, 48)),          # 3 x 0.03 + 0.01
    "streaming-2d": dict(dim=2, cfg=dict(nx=80, nz=36, heaters=5, dt_solver=0.02, dt_control=0.06, lx=4.0), obs=(6, 20)),
    "3d": dict(dim=3, cfg=dict(shape=(16, 24, 32), domain=(2.0, 3 * np.pi, 4 * np.pi), heaters=4, dt_solver=0.01, dt_control=0.035), obs=None),
}


class Mirror:
    """one env of the batch on the oracle, driven by the same calls"""

    def __init__(self, target, ra, clock):
        t = TARGETS[target]
        self.dim, self.clock = t["dim"], clock
        if self.dim == 2:
            self.o = oracle_py.OracleSim(obs=t["obs"], kick=0.05, ra=ra, **t["cfg"])
            self.dt, self.tff = t["cfg"]["dt_solver"], 1.0
        else:
            self.o = oracle_py.Oracle3D(kick=0.1, ra=ra, **t["cfg"])
            self.dt, self.tff = t["cfg"]["dt_solver"], 4.0
        dtc = t["cfg"]["dt_control"]
        self.nfull = int(np.floor(dtc / self.dt + 1e-9))
        self.rem = dtc - self.nfull * self.dt
        self.dtc = dtc
        self.fresh = True
        self.t, self.step_no = 0.0, 1

    def reset_random(self, seed):
        self.o.reset_random(int(seed)); self.fresh = True; self.t, self.step_no = 0.0, 1

    def reset_from_arrays(self, arrs):
        self.o.reset_from_arrays(*arrs); self.fresh = True; self.t, self.step_no = 0.0, 1

    def step(self, act):
        if self.clock == "recorded" and not self.fresh:
            self.o.set_action(act); self.o.update_state()
            for _ in range(self.nfull - 1):
                self.o.substep(self.dt * self.tff)
            if self.rem > 1e-12:
                self.o.substep(self.rem * self.tff)
        else:
            assert self.o.step(act)
        self.fresh = False
        self.t += self.dtc * self.tff; self.step_no += 1


def run_sequence(target, rng, ncalls):
    t = TARGETS[target]
    B = int(rng.choice([3, 4, 6]))
    clock = str(rng.choice(["documented", "recorded"]))
    ras = 10 ** rng.uniform(3.5, 4.6, B)
    if t["dim"] == 2:
        sim = _native.NativeSim(batch=B, obs_nz=t["obs"][0], obs_nx=t["obs"][1], random_kick=0.05, reference_clock=clock, **t["cfg"])
        ashape = (t["cfg"]["heaters"],)
    else:
        sim = _native.NativeSim3D(batch=B, random_kick=0.1, reference_clock=clock, **t["cfg"])
        ashape = (t["cfg"]["heaters"],) * 2
    sim.set_rayleigh(ras)
    mir = [Mirror(target, float(ras[e]), clock) for e in range(B)]
    seeds = rng.integers(1, 2 ** 40, B).astype(np.uint64)
    sim.reset(seeds)
    for e in range(B):
        mir[e].reset_random(seeds[e])
    worst, log = 0.0, [f"B={B} {clock}"]

    def check(tag):
        nonlocal worst
        f = sim.get_fields()
        tt, ss = sim.get_info()
        nu = sim.get_nusselt()
        nus = nu[0] if t["dim"] == 2 else nu
        assert not sim.get_flags().any(), (tag, "NaN flag")
        for e in range(B):
            d = max(rel_l2(a[e], b) for a, b in zip(f, mir[e].o.fields()))
            worst = max(worst, d)
            assert d < 1e-9, (tag, e, d, log)
            assert abs(tt[e] - mir[e].t) < 1e-9 and ss[e] == mir[e].step_no, (tag, e, tt[e], mir[e].t, ss[e], mir[e].step_no, log)
            onu = mir[e].o.nusselt(True) if t["dim"] == 2 else mir[e].o.nusselt()
            assert abs(nus[e] - onu) < 1e-7 * max(1.0, abs(onu)), (tag, e, nus[e], onu, log)

    check("reset")
    for n in range(ncalls):
        op = rng.choice(["step", "step", "step", "zero", "reset", "arrays"])
        if op in ("step", "zero"):
            act = rng.uniform(-1.2, 1.2, (B,) + ashape).astype(np.float32) if op == "step" else np.zeros((B,) + ashape, np.float32)
            assert sim.step(act)
            for e in range(B):
                mir[e].step(act[e])
            log.append(op)
        elif op == "reset":
            mask = (rng.random(B) < 0.4).astype(np.uint8)
            if not mask.any():
                mask[int(rng.integers(0, B))] = 1
            seeds = rng.integers(1, 2 ** 40, B).astype(np.uint64)
            sim.reset(seeds, mask=mask)
            for e in range(B):
                if mask[e]:
                    mir[e].reset_random(seeds[e])
            log.append("reset:" + "".join(str(int(m)) for m in mask))
        else:                                                      # masked reset from arrays: every marked env gets the state of env 0's oracle
            mask = (rng.random(B) < 0.4).astype(np.uint8)
            if not mask.any():
                mask[int(rng.integers(0, B))] = 1
            src = [x.copy() for x in mir[0].o.fields()]
            sim.reset_from_arrays(*[np.stack([x] * B) for x in src], mask=mask)
            for e in range(B):
                if mask[e]:
                    mir[e].reset_from_arrays(src)
            log.append("arrays:" + "".join(str(int(m)) for m in mask))
        check(f"call {n} {log[-1]}")
    sim.close()
    return worst, log


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    nseq = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ncalls = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    oracle_py.build_oracle()
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for target in TARGETS:
        for q in range(nseq):
            try:
                w, log = run_sequence(target, rng, ncalls)
                print(f"ok  {target} #{q}: worst field difference {w:.2e}   {' '.join(log)}", flush=True)
            except AssertionError as exc:
                bad += 1
                print(f"BAD {target} #{q}: {exc}", flush=True)
    print(f"{3 * nseq} sequences, {bad} failed, {time.time() - t0:.0f} s")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
