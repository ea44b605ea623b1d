#!/usr/bin/env python3
"""Where does the +2.4 % of kinetic energy at Ra = 1e7 come from?  
The generator's protocol at ONE Rayleigh number with a solver step / kick / clock of choice; prints the ensemble's kinetic-energy
distribution next to the reference's 40 episodes.   python scripts/ra1e7_probe.py [ra=1e7] [members=1024] [dt=0.03] [kick=0.02] [t_end=600] [precision=f64]
"""
import json, os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
from rbc_gym import _native  # noqa: E402

if __name__ == "__main__":
    ra = float(sys.argv[1]) if len(sys.argv) > 1 else 1e7
    members = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    dt = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
    kick = float(sys.argv[4]) if len(sys.argv) > 4 else 0.02
    t_end = float(sys.argv[5]) if len(sys.argv) > 5 else 600.0
    prec = sys.argv[6] if len(sys.argv) > 6 else "f64"
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "ckpt2d_pins.json")))
    ref = np.array([e["ke"] for sp in ("train", "val", "test") for e in pins[f"{sp}/ckpt_ra{int(ra)}"]["episodes"]])
    sim = _native.NativeSim(batch=members, ra=ra, random_kick=kick, write_state=0, dt_solver=dt, precision=_native.PRECISIONS[prec])
    sim.reset(np.arange(members, dtype=np.uint64) + 99)
    zero = np.zeros((members, 12), np.float32)
    q = [0, .05, .25, .5, .75, .95, 1]
    for n in range(1, int(round(t_end / 1.5)) + 1):
        sim.step(zero)
        if n * 1.5 in (300.0, 450.0, t_end):
            _, u, w = sim.get_fields()
            ke = 0.5 * ((u ** 2).mean((1, 2)) + (w[:, :-1] ** 2).mean((1, 2)))
            ok = np.isfinite(ke)
            print(f"Ra={ra:.0e} dt={dt} kick={kick} {prec} t={n * 1.5:.0f}: KE {ke[ok].mean():.5f} +- {ke[ok].std(ddof=1) / np.sqrt(ok.sum()):.5f} (std {ke[ok].std(ddof=1):.5f}; {members - ok.sum()} NaN envs)  "
                  f"ref {ref.mean():.5f} +- {ref.std(ddof=1) / np.sqrt(ref.size):.5f} (std {ref.std(ddof=1):.5f})   z = "
                  f"{(ke[ok].mean() - ref.mean()) / np.hypot(ke[ok].std(ddof=1) / np.sqrt(ok.sum()), ref.std(ddof=1) / np.sqrt(ref.size)):+.1f}", flush=True)
    d = np.load(os.path.join(ROOT, "tests", "golden", f"ckpt2d_ra{int(ra)}_profiles.npz"))
    spec = np.abs(np.fft.rfft(w[:, w.shape[1] // 2], axis=1))[:, :9] / w.shape[2]
    for name, sp, k_e, um in (("mine", spec[ok], ke[ok], u[ok].mean(2)), ("ref ", d["wmid_spec"], d["ke"], d["umean"])):
        dom = sp[:, 1:].argmax(1) + 1
        zon = np.sqrt((um ** 2).mean(1))                     # rms over rows of the horizontal-mean flow
        print(f"   {name}: dominant k of w at mid-height: " + ", ".join(f"k={k}: {100 * (dom == k).mean():.1f} % (KE {k_e[dom == k].mean():.4f})" for k in np.unique(dom)) +
              f";  |W_2| {sp[:, 2].mean():.4f}  |W_1| {sp[:, 1].mean():.4f}  |W_3| {sp[:, 3].mean():.4f}  |W_4| {sp[:, 4].mean():.4f}; zonal-flow rms {zon.mean():.4f} (max {zon.max():.4f}); corr(KE, zonal) {np.corrcoef(k_e, zon)[0, 1]:+.2f}")
    print("   quantiles mine", np.round(np.quantile(ke[ok], q), 4), " ref", np.round(np.quantile(ref, q), 4), flush=True)
    sim.close()
