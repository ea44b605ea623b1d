#!/usr/bin/env python3
"""Row-wise profile pin at the chaotic Rayleigh numbers on the GPU path (tests/profile_pins.py).

    python scripts/profile_pin.py [members=1024] [protocol=env|generator] [out=gpurun_out/profile_pin.json]
    RBC_HIP_LIB=build/librbc_hip_sym1.so python scripts/profile_pin.py ...      (the rejected advecting-velocity variant)

protocol=env:        400 env-steps of heater_duration 1.5 (50 solver steps each) to t = 600 -- the generator's model time with the
                     documented clock.
protocol=generator:  what the reference's generator did: `run!` re-entered every 10 solver steps (rbc_sim2D.jl:189-194,
                     --delta_t_snap 0.3), 2000 times, under reference_clock="recorded" (one solver step lost per re-entry).
Prints per Ra the z-scores of every row of <b>, <u^2>, <w^2>, <w b>, <b^2> against the reference's 40 episodes, the kinetic
energy, and where the kinetic-energy difference sits (which rows, which moment).  Needs an MI355X.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rbc_gym import _native  # noqa: E402
import profile_pins as pp  # noqa: E402


def run(ras, members, protocol="env", seed0=2024, t_end=600.0, ke_times=()):
    """-> {ra: (members, 5, nz) row moments at t_end}, {ra: {t: ensemble-mean KE}}"""
    B = members * len(ras)
    kw = dict(batch=B, random_kick=0.02, write_state=0)
    if protocol == "generator":
        kw.update(dt_control=0.3, reference_clock="recorded")
    sim = _native.NativeSim(**kw)
    dtc = sim.cfg.dt_control
    sim.set_rayleigh(np.repeat(np.array(ras, dtype=np.float64), members))
    sim.reset(np.arange(B, dtype=np.uint64) + seed0)
    zero = np.zeros((B, 12), np.float32)
    steps = int(round(t_end / dtc))
    marks = {int(round(t / dtc)): t for t in ke_times}
    series = {ra: {} for ra in ras}
    for n in range(1, steps + 1):
        if not sim.step(zero):
            raise RuntimeError(f"NaN envs: {np.nonzero(sim.get_flags())[0]}")
        if n in marks:
            _, u, w = sim.get_fields()
            ke = 0.5 * ((u ** 2).mean((1, 2)) + (w[:, :-1] ** 2).mean((1, 2)))
            for j, ra in enumerate(ras):
                series[ra][marks[n]] = (float(ke[j * members:(j + 1) * members].mean()), float(ke[j * members:(j + 1) * members].std(ddof=1) / np.sqrt(members)))
    b, u, w = sim.get_fields()
    sim.close()
    mom = pp.row_moments_batch(b, u, w)
    return {ra: mom[j * members:(j + 1) * members] for j, ra in enumerate(ras)}, series


if __name__ == "__main__":
    members = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    protocol = sys.argv[2] if len(sys.argv) > 2 else "env"
    dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "profile_pin.json")
    ras = list(pp.CHAOTIC_RAS)
    t0 = time.time()
    ens, series = run(ras, members, protocol, ke_times=(300.0, 450.0, 540.0, 600.0))
    print(f"{protocol}: {members} members x {len(ras)} Ra in {time.time() - t0:.1f} s  (library: {os.environ.get('RBC_HIP_LIB', 'shipped')})", flush=True)
    rec = {"protocol": protocol, "members": members, "library": os.environ.get("RBC_HIP_LIB", "shipped"), "per_ra": {}}
    for ra in ras:
        ref = pp.reference_profiles(ra)
        z, rel = pp.profile_z(ens[ra], ref)
        zmax, zrms = pp.summarise(z)
        zs, _ = pp.profile_z(pp.symmetrised(ens[ra]), pp.symmetrised(ref))
        smax, srms = pp.summarise(zs)
        ke_e = 0.5 * (ens[ra][:, 1].mean(1) + ens[ra][:, 2].mean(1))
        ke_r = 0.5 * (ref[:, 1].mean(1) + ref[:, 2].mean(1))
        zke = (ke_e.mean() - ke_r.mean()) / np.hypot(ke_e.std(ddof=1) / np.sqrt(len(ke_e)), ke_r.std(ddof=1) / np.sqrt(len(ke_r)))
        du2 = ens[ra][:, 1].mean(0) - ref[:, 1].mean(0)             # where the kinetic-energy difference sits, row by row
        dw2 = ens[ra][:, 2].mean(0) - ref[:, 2].mean(0)
        nz = du2.size
        wall = list(range(0, 4)) + list(range(nz - 4, nz))
        print(f"Ra={ra:>8d}  rows: max|z| {zmax:.2f} rms {zrms:.2f}  (folded: {smax:.2f} / {srms:.2f})   KE {ke_e.mean():.5f} vs {ke_r.mean():.5f} (z={zke:+.1f}, {100 * (ke_e.mean() / ke_r.mean() - 1):+.2f} %)"
              f"   dKE from u^2 {0.5 * du2.mean():+.5f}, from w^2 {0.5 * dw2.mean():+.5f}; share of the 8 wall rows {(du2[wall].sum() + dw2[wall].sum()) / (du2.sum() + dw2.sum() + 1e-300):.2f}")
        for m, name in enumerate(pp.MOMENTS):
            k = int(np.abs(z[m]).argmax())
            print(f"      {name:3s}: max|z| {np.abs(z[m]).max():.2f} at row {k} (rel {100 * rel[m, k]:+.2f} %), rms {np.sqrt((z[m] ** 2).mean()):.2f};  wall rows z: " +
                  " ".join(f"{v:+.1f}" for v in z[m, :4]) + " | " + " ".join(f"{v:+.1f}" for v in z[m, -4:]))
        print("      KE(t): " + "  ".join(f"t={t:.0f}: {v[0]:.5f}+-{v[1]:.5f}" for t, v in sorted(series[ra].items())), flush=True)
        rec["per_ra"][str(ra)] = {"z": np.round(z, 3).tolist(), "z_folded": np.round(zs, 3).tolist(), "rows_max_abs_z": zmax, "rows_rms_z": zrms,
                                  "folded_max_abs_z": smax, "folded_rms_z": srms, "ke": float(ke_e.mean()), "ke_ref": float(ke_r.mean()), "ke_z": float(zke),
                                  "ke_series": {str(t): v for t, v in series[ra].items()}}
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as f:
        json.dump(rec, f)
    print("wrote", dst)
