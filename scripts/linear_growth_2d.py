import os, sys, numpy as np
sys.path.insert(0, "rbc-gym_amd"); sys.path.insert(0, "tests")
from rbc_gym import _native
from linear_theory3d import energy_series_2d
np.set_printoptions(precision=4, linewidth=200)
for grid, B, seed0 in (((64, 128), 4096, 9000), ((64, 128), 4096, 50000), ((64, 96), 4096, 50000)):
    nz, nx = grid
    th = energy_series_2d(1e4, 7, shape=grid)
    sim = _native.NativeSim(batch=B, ra=1e4, nx=nx, nz=nz, obs_nx=nx // 2, obs_nz=8)
    sim.reset(np.arange(B, dtype=np.uint64) + seed0)
    zero = np.zeros((B, 12), np.float32)
    ke = []
    for n in range(7):
        assert sim.step(zero)
        b, u, w = sim.get_fields()
        ke.append(0.5 * ((u ** 2).mean(axis=(1, 2)) + (w[:, :-1] ** 2).mean(axis=(1, 2))))
    sim.close()
    ke = np.array(ke); m = ke.mean(1); se = ke.std(1, ddof=1) / np.sqrt(B)
    print(grid, B, seed0, "KE/theory", m / th, "se/m", se / m)
    print("   increments/theory", np.diff(np.log(m)) / np.diff(np.log(th)))
