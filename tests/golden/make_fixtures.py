#!/opt/conda/bin/python3.9
"""Generate the golden fixtures under tests/golden/ from the reference's DATA files.

Run in the build container only (needs /root/reference and h5py, which lives in
/opt/conda/bin/python3.9 there):

    /opt/conda/bin/python3.9 tests/golden/make_fixtures.py

Inputs : /root/reference/data/checkpoints/{train,val,test}/ckpt_ra*.h5
         (Oceananigans output written by the reference's checkpoint generator,
          rbc_sim2D.jl:33-70; h5py sees b,u as (Nz,1,Nx,E) and w as (Nz+1,1,Nx,E), f64)
Outputs: ckpt2d_ra10000.npz   - 3 episodes of train/ckpt_ra10000 (b,u,w f64)
         ckpt2d_ra100000.npz  - 1 episode of train/ckpt_ra100000
         ckpt2d_ra{30000,...,10000000}_profiles.npz - per-episode row-wise moments of the 40 episodes per chaotic Ra
                                (`make_fixtures.py profiles` writes only these)
         ckpt2d_pins.json     - per-file known-answer table (divergence, KE, Nusselt
                                on the full state and on the 8x48 sensor grid,
                                horizontal-mean b profile) computed with numpy from the data.
Only DATA travels: no reference code is executed or copied.
"""
import json, os, sys
import numpy as np
import h5py

REF = "/root/reference/data/checkpoints"
OUT = os.path.dirname(os.path.abspath(__file__))
LX, LZ = 2 * np.pi, 2.0
PR = 0.7


def array_gradient(a):
    # restates rbc_sim2D.jl:206-220 (index-unit gradient, one-sided ends)
    g = np.empty_like(a)
    g[0] = a[1] - a[0]
    g[-1] = a[-1] - a[-2]
    g[1:-1] = (a[2:] - a[:-2]) / 2
    return g


def nusselt(T, w, kappa):
    # restates rbc_sim2D_api.jl:142-163 on arrays laid out (z, x)
    q1 = np.mean(T * w)
    Tx = T.mean(axis=1)
    q2 = kappa * np.mean(array_gradient(Tx))
    return (q1 - q2) / (kappa * 1.0 / LZ)


def analyse(path, ra):
    with h5py.File(path, "r") as f:
        b = f["b"][...][:, 0]  # (Nz, Nx, E)
        u = f["u"][...][:, 0]
        w = f["w"][...][:, 0]  # (Nz+1, Nx, E)
        attrs = {k: int(v) for k, v in f.attrs.items()}
    nz, nx, E = b.shape
    dx, dz = LX / nx, LZ / nz
    kappa = 1 / np.sqrt(PR * ra)
    rows = []
    for e in range(E):
        be, ue, we = b[..., e], u[..., e], w[..., e]
        div = (np.roll(ue, -1, axis=1) - ue) / dx + (we[1:] - we[:-1]) / dz
        ke = 0.5 * (np.mean(ue**2) + np.mean(we[:nz] ** 2))
        nu_state = nusselt(be, we[:nz], kappa)
        nu_obs = nusselt(be[0:nz:nz // 8, 0:nx:2], we[0:nz:nz // 8, 0:nx:2], kappa)
        rows.append(dict(max_abs_div=float(np.abs(div).max()), ke=float(ke),
                         nusselt_state=float(nu_state), nusselt_obs=float(nu_obs),
                         w_bottom_max=float(np.abs(we[0]).max()), w_top_max=float(np.abs(we[nz]).max()),
                         mean_b=float(be.mean()), umax=float(np.abs(ue).max()), wmax=float(np.abs(we).max())))
    prof = b.mean(axis=(1, 2))
    return dict(attrs=attrs, shape=[int(nz), int(nx), int(E)], episodes=rows,
                mean_b_profile=[float(x) for x in prof]), (b, u, w)


def main():
    pins = {}
    for split in ("train", "val", "test"):
        for ra in (10000, 30000, 100000, 300000, 1000000, 3000000, 10000000):
            p = f"{REF}/{split}/ckpt_ra{ra}.h5"
            info, (b, u, w) = analyse(p, ra)
            pins[f"{split}/ckpt_ra{ra}"] = info
            if split == "train" and ra == 10000:
                ep = [0, 1, 2]
                np.savez_compressed(f"{OUT}/ckpt2d_ra10000.npz",
                                    b=np.ascontiguousarray(np.moveaxis(b[..., ep], -1, 0)),
                                    u=np.ascontiguousarray(np.moveaxis(u[..., ep], -1, 0)),
                                    w=np.ascontiguousarray(np.moveaxis(w[..., ep], -1, 0)),
                                    episodes=np.array(ep), ra=np.array(ra))
            if split == "train" and ra == 100000:
                ep = [0]
                np.savez_compressed(f"{OUT}/ckpt2d_ra100000.npz",
                                    b=np.ascontiguousarray(np.moveaxis(b[..., ep], -1, 0)),
                                    u=np.ascontiguousarray(np.moveaxis(u[..., ep], -1, 0)),
                                    w=np.ascontiguousarray(np.moveaxis(w[..., ep], -1, 0)),
                                    episodes=np.array(ep), ra=np.array(ra))
    # per-episode horizontal-mean profiles of all 40 Ra=1e4 episodes (ensemble pin for the
    # from-rest run of the oracle, tests/golden/oracle_ensemble.py): [episode][<b>,<u^2>,<w^2>,<wb>,<b^2>][k]
    prof = []
    for split in ("train", "val", "test"):
        with h5py.File(f"{REF}/{split}/ckpt_ra10000.h5", "r") as f:
            b = f["b"][...][:, 0]; u = f["u"][...][:, 0]; w = f["w"][...][:, 0]
        for e in range(b.shape[-1]):
            be, ue, wc = b[..., e], u[..., e], w[:-1, :, e]
            prof.append(np.stack([be.mean(1), (ue**2).mean(1), (wc**2).mean(1), (be * wc).mean(1), (be**2).mean(1)]))
    np.savez_compressed(f"{OUT}/ckpt2d_ra10000_profiles.npz", profiles=np.array(prof))
    spectra()
    with open(f"{OUT}/ckpt2d_pins.json", "w") as f:
        json.dump(pins, f, indent=1)
    r = pins["train/ckpt_ra10000"]["episodes"]
    print("Ra=1e4 train: KE", np.mean([x["ke"] for x in r]), "Nu_state", np.mean([x["nusselt_state"] for x in r]),
          "Nu_obs", np.mean([x["nusselt_obs"] for x in r]), "max div", max(x["max_abs_div"] for x in r))


sys.path.insert(0, os.path.dirname(OUT))
from spectral_invariants import SPEC_K, field_spectra  # noqa: E402  (tests/spectral_invariants.py)


def spectra():
    """ckpt2d_ra10000_spectra.npz: mean and standard deviation over the reference's 40 Ra=1e4 episodes (train 20, val 10,
    test 10; all are x-translates of ONE steady k=2 state) of the translation invariants above: the whole steady field
    structure, not just five profile moments.  Phases are stored as mean/std of cos and sin."""
    mods, phs = [], []
    for split in ("train", "val", "test"):
        with h5py.File(f"{REF}/{split}/ckpt_ra10000.h5", "r") as f:
            b = f["b"][...][:, 0]; u = f["u"][...][:, 0]; w = f["w"][...][:, 0]
        for e in range(b.shape[-1]):
            m, p = field_spectra(b[..., e], u[..., e], w[:-1, :, e])
            mods.append(m); phs.append(p)
    mods, phs = np.array(mods), np.array(phs)
    np.savez_compressed(f"{OUT}/ckpt2d_ra10000_spectra.npz", k=np.array(SPEC_K), episodes=np.array(len(mods)),
                        mod_mean=mods.mean(0), mod_std=mods.std(0, ddof=1),
                        cos_mean=np.cos(phs).mean(0), cos_std=np.cos(phs).std(0, ddof=1),
                        sin_mean=np.sin(phs).mean(0), sin_std=np.sin(phs).std(0, ddof=1))
    print("spectra: |B_2| rows 0..2", mods.mean(0)[0, 1, :3], "rel spread", (mods.std(0, ddof=1)[0, 1, :3] / mods.mean(0)[0, 1, :3]))


PROFILE_MOMENTS = ("b", "u2", "w2", "wb", "b2")


def row_moments(b, u, w):
    """per-row means over x of one state (z, x): <b>, <u^2>, <w^2>, <w b>, <b^2> with w at the face BELOW the cell (index k,
    the convention of get_nusselt, rbc_sim2D_api.jl:142-163) -> (5, nz)"""
    wc = w[:-1]
    return np.stack([b.mean(1), (u ** 2).mean(1), (wc ** 2).mean(1), (b * wc).mean(1), (b ** 2).mean(1)])


def chaotic_profiles():
    """ckpt2d_ra{Ra}_profiles.npz for the six chaotic Rayleigh numbers: the row-wise moments above of each of the reference's 40
    episodes (train 20, val 10, test 10: independent random initial conditions at t = 600, rbc_sim2D.jl:41-43,64-66) --
    where the boundary layers are 1-2 cells thick these rows are what the near-wall advection stencils decide.  Data only."""
    for ra in (30000, 100000, 300000, 1000000, 3000000, 10000000):
        prof, spec, kes, umeans, xspec = [], [], [], [], []
        for split in ("train", "val", "test"):
            with h5py.File(f"{REF}/{split}/ckpt_ra{ra}.h5", "r") as f:
                b = f["b"][...][:, 0]; u = f["u"][...][:, 0]; w = f["w"][...][:, 0]
            for e in range(b.shape[-1]):
                prof.append(row_moments(b[..., e], u[..., e], w[..., e]))
                # which large-scale pattern the episode is in: |W_k| / nx of the mid-height w row, k = 0..8 (translation invariant)
                spec.append(np.abs(np.fft.rfft(w[w.shape[0] // 2, :, e]))[:9] / w.shape[1])
                kes.append(0.5 * ((u[..., e] ** 2).mean() + (w[:-1, :, e] ** 2).mean()))
                umeans.append(u[..., e].mean(1))                       # horizontal-mean (zonal) flow per row
                # x power spectra averaged over the rows, |F_k|^2 / nx^2 for b', u, w (b' = b minus its row mean): where the grid-scale
                # dissipation of the advection scheme shows (49 wavenumbers x 3 fields per episode)
                xspec.append(np.stack([(np.abs(np.fft.rfft(f - f.mean(1, keepdims=True), axis=1)) ** 2).mean(0) / f.shape[1] ** 2
                                       for f in (b[..., e], u[..., e], w[:-1, :, e])]))
        prof = np.array(prof)
        np.savez_compressed(f"{OUT}/ckpt2d_ra{ra}_profiles.npz", profiles=prof, moments=np.array(PROFILE_MOMENTS), ra=np.array(ra),
                            wmid_spec=np.array(spec), ke=np.array(kes), umean=np.array(umeans), xspec=np.array(xspec))
        m, se = prof.mean(0), prof.std(0, ddof=1) / np.sqrt(len(prof))
        print(f"Ra={ra}: {len(prof)} episodes; <b> rows 0..2 {np.round(m[0, :3], 4)} +- {np.round(se[0, :3], 4)}; <w^2> rows 1..3 {np.round(m[2, 1:4], 5)}")


def small_h5():
    """ckpt2d_small.h5: 2 episodes of train/ckpt_ra10000 re-written with h5py in the reference
    writer's layout (attrs num_episodes/start_seed, contiguous f64 datasets (Nz[+1],1,Nx,E)) --
    input for the dependency-free HDF5 reader test."""
    with h5py.File(f"{REF}/train/ckpt_ra10000.h5", "r") as f, h5py.File(f"{OUT}/ckpt2d_small.h5", "w", libver="earliest") as g:
        g.attrs["num_episodes"] = np.int64(2)
        g.attrs["start_seed"] = np.int64(f.attrs["start_seed"])
        for k in ("b", "u", "w"):
            g.create_dataset(k, data=f[k][..., :2])


if __name__ == "__main__":
    if sys.argv[1:] == ["profiles"]:
        chaotic_profiles()
    else:
        main()
        small_h5()
        chaotic_profiles()
