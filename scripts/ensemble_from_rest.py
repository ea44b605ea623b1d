import sys, os, json
sys.path.insert(0, "rbc-gym_amd")
import numpy as np
from rbc_gym import _native
n = 1024
sim = _native.NativeSim(batch=n, random_kick=0.02, write_state=0)
sim.reset(np.arange(n, dtype=np.uint64) + 4242)
zero = np.zeros((n, 12), np.float32)
for _ in range(400):
    assert sim.step(zero)
b, u, w = sim.get_fields()
ke = 0.5 * ((u ** 2).mean((1, 2)) + (w[:, :64] ** 2).mean((1, 2)))
nus, nuo = sim.get_nusselt()
print("ke quantiles", np.quantile(ke, [0, .01, .05, .25, .5, .75, .95, .99, 1]))
on = np.abs(ke - 0.0983448) < 1e-4
print("on-branch", on.sum(), "of", n)
print("on-branch ke mean %.9f sem %.2e ; nus %.6f sem %.2e ; nuo %.6f sem %.2e" % (ke[on].mean(), ke[on].std(ddof=1)/np.sqrt(on.sum()), nus[on].mean(), nus[on].std(ddof=1)/np.sqrt(on.sum()), nuo[on].mean(), nuo[on].std(ddof=1)/np.sqrt(on.sum())))
off = ~on
print("off-branch ke values", np.sort(ke[off])[:20], "nus", np.sort(nus[off])[:10])
# dominant wavenumber of mid-height w
spec = np.abs(np.fft.rfft(w[:, 32], axis=1))
dom = spec[:, 1:].argmax(1) + 1
print("dominant k counts", np.bincount(dom))
print("off-branch dominant k", np.bincount(dom[off]))
prof = b[on].mean((0, 2)); print("profile bottom", prof[:3], "top", prof[-3:])
# continue to t=1200 and see if off-branch members converge
for _ in range(400):
    sim.step(zero)
b2, u2, w2 = sim.get_fields()
ke2 = 0.5 * ((u2 ** 2).mean((1, 2)) + (w2[:, :64] ** 2).mean((1, 2)))
print("t=1200: on-branch", (np.abs(ke2 - 0.0983448) < 1e-4).sum(), "ke quantiles", np.quantile(ke2, [0, .01, .5, .99, 1]))
