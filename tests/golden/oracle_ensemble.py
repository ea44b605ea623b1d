#!/usr/bin/env python3
"""Ensemble pin of the oracle's discretisation (slow: ~10 min on 8 cores; run by hand).

Restates the reference's checkpoint generator (scripts/create_checkpoints_2D.sh:18-20 ->
rbc_sim2D.jl:15-72: N=96x64, Ra=1e4, kick 0.02, dt 0.03, run! every 0.3, duration 600, zero
action) on the CPU oracle from independent random initial states and compares the ensemble at
t=600 with the 40 Oceananigans episodes stored in the reference's ckpt_ra10000.h5 files
(tests/golden/ckpt2d_pins.json, ckpt2d_ra10000_profiles.npz).  Julia's RNG stream cannot be
reproduced, so the comparison is statistical: mean kinetic energy, Nusselt numbers and the
horizontal-mean profiles must agree within the ensembles' standard errors.

Writes tests/golden/oracle_ensemble_ra10000.json (the result recorded in DESIGN.md).
usage: python tests/golden/oracle_ensemble.py [n_members=12] [symlevel ...]
"""
import json, os, sys
import numpy as np
from multiprocessing import Pool

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_py import OracleSim, VAR_SYMLEVEL  # noqa: E402


def prof(s):
    b, u, w = s.fields()
    wc = w[:-1]
    return np.stack([b.mean(1), (u**2).mean(1), (wc**2).mean(1), (b * wc).mean(1), (b**2).mean(1)])


def run(args):
    sym, seed = args
    s = OracleSim(ra=1e4, kick=0.02, dt_control=0.3, variants={VAR_SYMLEVEL: sym})
    s.reset_random(seed)
    for _ in range(2000):
        s.step(None)
    return sym, seed, s.kinetic_energy(), s.nusselt(True), s.nusselt(False), prof(s)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    syms = [int(a) for a in sys.argv[2:]] or [0, 1]
    pins = json.load(open(f"{HERE}/ckpt2d_pins.json"))
    P = np.load(f"{HERE}/ckpt2d_ra10000_profiles.npz")["profiles"]
    ref = [e for sp in ("train", "val", "test") for e in pins[f"{sp}/ckpt_ra10000"]["episodes"]]
    ke = np.array([e["ke"] for e in ref]); nus = np.array([e["nusselt_state"] for e in ref]); nuo = np.array([e["nusselt_obs"] for e in ref])
    out = {"reference": {"n": len(ref), "ke_mean": ke.mean(), "ke_sem": ke.std(ddof=1) / np.sqrt(len(ref)),
                         "nu_state_mean": nus.mean(), "nu_state_sem": nus.std(ddof=1) / np.sqrt(len(ref)),
                         "nu_obs_mean": nuo.mean(), "nu_obs_sem": nuo.std(ddof=1) / np.sqrt(len(ref))}}
    with Pool(min(8, os.cpu_count() or 1)) as pool:
        res = pool.map(run, [(sym, 1000 + i) for i in range(n) for sym in syms])
    for sym in syms:
        r = [x for x in res if x[0] == sym]
        k = np.array([x[2] for x in r]); a = np.array([x[3] for x in r]); o = np.array([x[4] for x in r]); q = np.array([x[5] for x in r])
        se = np.sqrt(q.var(0, ddof=1) / len(r) + P.var(0, ddof=1) / P.shape[0])
        z = (q.mean(0) - P.mean(0)) / np.maximum(se, 1e-300)
        z[2, 0] = z[3, 0] = 0.0
        out[f"oracle_symlevel{sym}"] = {"n": len(r), "ke_mean": k.mean(), "ke_sem": k.std(ddof=1) / np.sqrt(len(r)),
                                        "nu_state_mean": a.mean(), "nu_state_sem": a.std(ddof=1) / np.sqrt(len(r)),
                                        "nu_obs_mean": o.mean(), "nu_obs_sem": o.std(ddof=1) / np.sqrt(len(r)),
                                        "ke_z": (k.mean() - ke.mean()) / np.hypot(k.std(ddof=1) / np.sqrt(len(r)), out["reference"]["ke_sem"]),
                                        "profile_chi2_per_row": [float((z[j] ** 2).mean()) for j in range(5)],
                                        "profile_max_abs_z": float(np.abs(z).max())}
    json.dump(out, open(f"{HERE}/oracle_ensemble_ra10000.json", "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
