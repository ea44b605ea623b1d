"""x-translation invariants of a 2D state (test infrastructure; shared by tests/golden/make_fixtures.py, which computes them
for the reference's 40 Ra=1e4 episodes, and by the GPU ensemble tests)."""
import numpy as np

SPEC_K = (0, 2, 4, 6, 8)


def field_spectra(b, u, w):
    """One state (arrays (nz, nx), w without its top wall row): per row the moduli |F_k| of the x-DFT of b, u, w for
    k in SPEC_K -> (3, 5, nz), and the cross phases arg(B_k conj W_k), arg(U_k conj W_k) for k in SPEC_K[1:] -> (2, 4, nz).
    A shift of the state by s cells multiplies every F_k by the same unit phase exp(-2 pi i k s / nx), so both sets are
    identical for all x-translates of one steady state."""
    F = [np.fft.rfft(f, axis=1) / f.shape[1] for f in (b, u, w)]
    mod = np.stack([np.stack([np.abs(f[:, k]) for k in SPEC_K]) for f in F])
    ks = SPEC_K[1:]
    ph = np.stack([np.stack([np.angle(F[0][:, k] * np.conj(F[2][:, k])) for k in ks]),
                   np.stack([np.angle(F[1][:, k] * np.conj(F[2][:, k])) for k in ks])])
    return mod, ph


def spectral_z_scores(b, u, w, ref, floor=1e-7):
    """z-scores of an ensemble of states (b, u: (m, nz, nx), w: (m, nz+1, nx), all on the k=2 steady state) against the
    reference's episode statistics `ref` (tests/golden/ckpt2d_ra10000_spectra.npz):
    (z of the moduli whose reference mean exceeds `floor`, z of the mean cos/sin of the cross phases of the strong modes,
    mean moduli, mask of the moduli used)."""
    m = b.shape[0]
    mods, phs = zip(*[field_spectra(b[e], u[e], w[e, :-1]) for e in range(m)])
    mods, phs = np.array(mods), np.array(phs)
    n_ref = int(ref["episodes"])

    def z(mine, mean, std):
        sem = np.hypot(mine.std(0, ddof=1) / np.sqrt(m), std / np.sqrt(n_ref))
        return (mine.mean(0) - mean) / np.maximum(sem, 1e-300)

    zm = z(mods, ref["mod_mean"], ref["mod_std"])
    big = ref["mod_mean"] > floor
    strong = ref["mod_mean"] > 1e-4                    # phases of weak modes are dominated by the residual oscillation
    both = np.stack([strong[0, 1:] & strong[2, 1:], strong[1, 1:] & strong[2, 1:]])
    # Cross phases: where the rolls' mirror symmetry locks them (cos = +-1, sin = 0, or the reverse) the spread is round-off;
    # where it does not, a state and its mirror image (x -> -x, another steady state) have opposite sin (b-w) or cos (u-w),
    # and each ensemble holds both in binomial proportions -- so the standard error gets an absolute floor (1e-5, the
    # relative spread of the moduli) and the comparison stays a z-score.
    def zfloor(mine, mean, std):
        sem = np.hypot(mine.std(0, ddof=1) / np.sqrt(m), std / np.sqrt(n_ref))
        return (mine.mean(0) - mean) / np.maximum(sem, 1e-5)
    zc, zs = zfloor(np.cos(phs), ref["cos_mean"], ref["cos_std"]), zfloor(np.sin(phs), ref["sin_mean"], ref["sin_std"])
    return zm[big], np.concatenate([zc[both], zs[both]]), mods.mean(0), big
