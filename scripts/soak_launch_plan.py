"""soak: per-group graphs against the one multi-stream graph over many env-steps with masked resets in between -- bitwise equal states,
no NaN flag, both precisions, both clocks (the launch plan reuses its start / done events every step)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
from rbc_gym import _native
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for prec in ("f64", "f32"):
    for clock in ("documented", "recorded"):
        kw = dict(batch=32, shape=(32, 48, 48), ra=1e4, dt_control=0.125, dt_solver=0.01, random_kick=0.1, precision=prec, reference_clock=clock)
        a = _native.NativeSim3D(**kw)
        os.environ["RBC_3D_GROUP_GRAPHS"] = "0"
        b = _native.NativeSim3D(**kw)
        del os.environ["RBC_3D_GROUP_GRAPHS"]
        assert a.launch_plan() == (4, 1) and b.launch_plan() == (4, 0), (a.launch_plan(), b.launch_plan())
        seeds = np.arange(32, dtype=np.uint64) + 3
        a.reset(seeds); b.reset(seeds)
        rng = np.random.default_rng(1)
        for n in range(steps):
            act = rng.uniform(-1, 1, (32, 8, 8)).astype(np.float32)
            assert a.step(act) and b.step(act), n
            if n % 37 == 36:
                m = (rng.random(32) < 0.3).astype(np.uint8)
                s2 = rng.integers(1, 1 << 30, 32).astype(np.uint64)
                a.reset(s2, mask=m); b.reset(s2, mask=m)
            if n % 50 == 49 or n == steps - 1:
                for x, y in zip(a.get_fields(), b.get_fields()):
                    assert np.array_equal(x, y), (prec, clock, n)
                assert np.array_equal(a.get_nusselt(), b.get_nusselt()) and np.array_equal(a.get_state(), b.get_state())
        print(prec, clock, steps, "env-steps: bitwise equal, mean Nu", float(a.get_nusselt().mean()), flush=True)
        a.close(); b.close()
