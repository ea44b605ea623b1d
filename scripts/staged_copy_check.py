"""large float32 outputs into pageable memory: the staged, multi-threaded copy against the plain hipMemcpy2D (RBC_STAGED_COPY=0) --
same bytes, and how long each takes (2D: 1024 x 3 of 5 channels = 75 MB strided; 3D configs[4]: 38 MB contiguous)"""
import os, sys, time, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
if len(sys.argv) > 1:
    from rbc_gym import _native
    if sys.argv[1] == "2d":
        sim = _native.NativeSim(batch=1024); sim.reset(np.arange(1024, dtype=np.uint64)); get = lambda: sim.get_state(3)
        sim.step(np.random.default_rng(0).uniform(-1, 1, (1024, 12)).astype(np.float32))
    else:
        sim = _native.NativeSim3D(batch=32, shape=(32, 48, 48), ra=1e4, random_kick=0.1); sim.reset(np.arange(32, dtype=np.uint64)); get = lambda: sim.get_state()
        sim.step(np.random.default_rng(0).uniform(-1, 1, (32, 8, 8)).astype(np.float32))
    a = get()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); b = get(); ts.append(time.perf_counter() - t0)
    np.save(sys.argv[2], a)
    print(f"{sys.argv[1]} RBC_STAGED_COPY={os.environ.get('RBC_STAGED_COPY', '1')}: {a.nbytes / 2**20:.0f} MiB in {np.median(ts) * 1e3:.2f} ms (fresh array each call), checksum {float(a.astype(np.float64).sum()):.6f}")
else:
    for dim in ("2d", "3d"):
        outs = []
        for st in ("1", "0"):
            f = f"/tmp/staged_{dim}_{st}.npy"
            subprocess.check_call([sys.executable, __file__, dim, f], env=dict(os.environ, RBC_STAGED_COPY=st))
            outs.append(np.load(f))
        print("   identical:", np.array_equal(outs[0], outs[1]))
