"""Linear theory of the restated 3D discretisation (test infrastructure, numpy only).

While the perturbation is small, the flow-statistics protocol (experiments/flowstats/flowstats_ra.py:27-66: random
kick on the conduction profile, zero action, Nu after every env-step) is a LINEAR stochastic problem: every horizontal
Fourier mode of the C-grid operators (second-order differences -> modified wavenumbers, rbc_sim3D.jl:156-164 model) evolves
independently, the RK3 substep with an exact projection after every stage is the polynomial 1 + z + z^2/2 + z^3/6 of
dt * P L (P: discrete Leray projector, L: diffusion + buoyancy + advection of the conduction profile), and the ensemble
mean of Nu - 1 = <w b'> / kappa (rbc_sim3D_api.jl:134-159) follows from propagating the covariance of the white-noise
initial condition (rbc_sim3D.jl:169-178, then `set!`'s projection).  No sampling, no free parameter.

Use: the growth of Nu - 1 per env-step that the discretisation MUST show, against (a) the native stepper's ensemble
(pins the build's clock and linear operator independently of any reference data) and (b) the reference's series.
"""
import numpy as np
from math import erf, exp, pi, sqrt


def _clamped_normal_variance(c):
    """variance of min(xi, c), xi ~ N(0,1) (the clamp of rbc_sim3D.jl:174 acting on the wall-adjacent cells)."""
    Phi = 0.5 * (1.0 + erf(c / sqrt(2.0)))
    phi = exp(-0.5 * c * c) / sqrt(2.0 * pi)
    m1 = -phi + c * (1.0 - Phi)
    m2 = (Phi - c * phi) + c * c * (1.0 - Phi)
    return m2 - m1 * m1


class LinearRBC3D:
    def __init__(self, ra, pr=0.7, shape=(32, 64, 64), domain=(2.0, 4 * pi, 4 * pi), kick=0.01, delta_b=1.0):
        self.nz, self.ny, self.nx = shape
        self.lz, self.ly, self.lx = domain
        self.nu = sqrt(pr / ra)
        self.kappa = 1.0 / sqrt(pr * ra)
        self.kick = kick
        self.beta = delta_b / self.lz                  # -dB/dz of the conduction profile
        self.dz = self.lz / self.nz
        nz, dz = self.nz, self.dz
        # vertical operators.  centre fields (u~, b'): ghost = -interior at both walls (no-slip / fixed wall value)
        d2c = np.zeros((nz, nz))
        for k in range(nz):
            d2c[k, k] = -2.0
            if k > 0: d2c[k, k - 1] = 1.0
            if k < nz - 1: d2c[k, k + 1] = 1.0
        d2c[0, 0] = -3.0
        d2c[nz - 1, nz - 1] = -3.0
        self.d2c = d2c / dz ** 2
        nf = nz - 1                                    # interior w faces (walls carry w = 0)
        d2f = np.zeros((nf, nf))
        for k in range(nf):
            d2f[k, k] = -2.0
            if k > 0: d2f[k, k - 1] = 1.0
            if k < nf - 1: d2f[k, k + 1] = 1.0
        self.d2f = d2f / dz ** 2
        # dzw: cell k <- (w_{k+1} - w_k)/dz with interior face f (0-based) lying between cells f and f+1
        dzw = np.zeros((nz, nf))
        avw = np.zeros((nz, nf))
        for f in range(nf):
            dzw[f, f] += 1.0 / dz;  dzw[f + 1, f] -= 1.0 / dz
            avw[f, f] += 0.5;       avw[f + 1, f] += 0.5
        self.dzw, self.avw = dzw, avw
        self.gzp = -dzw.T                              # face f <- (p_{f+1} - p_f)/dz
        self.avb = avw.T                               # face f <- (b_f + b_{f+1})/2
        # pairing of Nu: cell k with ITS LOWER face (w[1:Nz] of rbc_sim3D_api.jl:118 starts at the wall face)
        pair = np.zeros((nz, nf))
        for k in range(1, nz):
            pair[k, k - 1] = 1.0
        self.pair = pair

    def mode_operator(self, kt2):
        """(M, P) for modified horizontal wavenumber^2 kt2 on the state [u~' (nz), w (nz-1), b' (nz)]."""
        nz, nf = self.nz, self.nz - 1
        kt = sqrt(kt2)
        n = 2 * nz + nf
        iu, iw, ib = slice(0, nz), slice(nz, nz + nf), slice(nz + nf, n)
        L = np.zeros((n, n))
        L[iu, iu] = self.nu * (self.d2c - kt2 * np.eye(nz))
        L[iw, iw] = self.nu * (self.d2f - kt2 * np.eye(nf))
        L[iw, ib] = self.avb
        L[ib, ib] = self.kappa * (self.d2c - kt2 * np.eye(nz))
        L[ib, iw] = self.beta * self.avw
        D = np.zeros((nz, n));  D[:, iu] = kt * np.eye(nz);  D[:, iw] = self.dzw
        G = np.zeros((n, nz));  G[iu, :] = -kt * np.eye(nz);  G[iw, :] = self.gzp
        P = np.eye(n) - G @ np.linalg.solve(D @ G, D)
        return P @ L, P, (iu, iw, ib)

    def mode_set(self):
        """unique modified wavenumbers^2 of the horizontal grid with their multiplicities (kt2 = 0 left out: w == 0 there)."""
        def kt2_1d(n, l):
            m = np.arange(n)
            return (2.0 * n / l * np.sin(pi * m / n)) ** 2
        kx2, ky2 = kt2_1d(self.nx, self.lx), kt2_1d(self.ny, self.ly)
        allk = np.round((kx2[:, None] + ky2[None, :]).ravel(), 10)
        vals, counts = np.unique(allk, return_counts=True)
        keep = vals > 1e-12
        return vals[keep], counts[keep]

    def nusselt_series(self, steps, dt, nsub, nsub_first=None, time_scale=1.0):
        """E[Nu - 1] after each of `steps` env-steps of nsub RK3 substeps of size dt (the first one nsub_first)."""
        nz, nf = self.nz, self.nz - 1
        vals, counts = self.mode_set()
        nmodes = self.nx * self.ny
        out = np.zeros(steps)
        s2 = self.kick ** 2
        var_b = np.full(nz, s2)
        c = (self.beta * 0.5 * self.dz) / self.kick                # distance of the wall cells' mean from the clamp, in sigmas
        var_b[0] = var_b[-1] = s2 * _clamped_normal_variance(c)
        sig_max = 0.0
        for kt2, mult in zip(vals, counts):
            M, P, (iu, iw, ib) = self.mode_operator(kt2)
            n = M.shape[0]
            z = dt * time_scale * M
            R = np.eye(n) + z + z @ z / 2.0 + z @ z @ z / 6.0
            Rn = np.linalg.matrix_power(R, nsub)
            R1 = Rn if nsub_first in (None, nsub) else np.linalg.matrix_power(R, nsub_first)
            C = np.zeros((n, n))
            C[iu, iu] = s2 * np.eye(nz);  C[iw, iw] = s2 * np.eye(nf);  C[ib, ib] = np.diag(var_b)
            C = P @ C @ P.T                                          # set!'s projection of the initial velocities
            for s in range(steps):
                A = R1 if s == 0 else Rn
                C = A @ C @ A.T
                cwb = C[ib, iw]                                      # E[b_k w_f]
                out[s] += mult * np.sum(self.pair * cwb) / nz
            sig_max = max(sig_max, np.max(np.linalg.eigvals(M).real))
        self.sigma_max = sig_max
        return out / nmodes / self.kappa


def energy_series_2d(ra, steps, dt=0.03, nsub=50, shape=(64, 96), domain=(2.0, 2 * pi), kick=0.01, pr=0.7):
    """E[KE] = E[(<u^2> + <w^2>) / 2] of a 2D env (rbc_sim2D.jl:163-171 initial condition, `set!`'s projection, zero action) after each of
    `steps` env-steps of nsub RK3 substeps -- the same covariance propagation as LinearRBC3D.nusselt_series with ny = 1; the
    horizontal-mean mode of u (no pressure, pure diffusion between the no-slip walls) is carried separately.  What the linear
    phase of a from-rest run of the 2D kernel MUST show: pins its clock and linear operator with no reference data."""
    nz, nx = shape
    lin = LinearRBC3D(ra, pr=pr, shape=(nz, 1, nx), domain=(domain[0], 1.0, domain[1]), kick=kick)
    vals, counts = lin.mode_set()
    s2 = kick ** 2
    nf = nz - 1
    var_b = np.full(nz, s2)
    c = (lin.beta * 0.5 * lin.dz) / kick
    var_b[0] = var_b[-1] = s2 * _clamped_normal_variance(c)
    out = np.zeros(steps)
    for kt2, mult in zip(vals, counts):
        M, P, (iu, iw, ib) = lin.mode_operator(kt2)
        n = M.shape[0]
        z = dt * M
        Rn = np.linalg.matrix_power(np.eye(n) + z + z @ z / 2.0 + z @ z @ z / 6.0, nsub)
        C = np.zeros((n, n))
        C[iu, iu] = s2 * np.eye(nz); C[iw, iw] = s2 * np.eye(nf); C[ib, ib] = np.diag(var_b)
        C = P @ C @ P.T
        for s in range(steps):
            C = Rn @ C @ Rn.T
            out[s] += mult * 0.5 * (np.trace(C[iu, iu]) + np.trace(C[iw, iw])) / nz
    # horizontal mean of u: du/dt = nu d2c u
    z = dt * lin.nu * lin.d2c
    Rn = np.linalg.matrix_power(np.eye(nz) + z + z @ z / 2.0 + z @ z @ z / 6.0, nsub)
    C = s2 * np.eye(nz)
    for s in range(steps):
        C = Rn @ C @ Rn.T
        out[s] += 0.5 * np.trace(C) / nz
    return out / nx


def growth_rate(ra, kt2_window=(1.0, 4.5), **kw):
    """largest growth rate over the horizontal modes with modified wavenumber^2 inside `kt2_window` (the null eigenvalues the
    projector adds are left out) and the mode that has it"""
    lin = LinearRBC3D(ra, **kw)
    vals, _ = lin.mode_set()
    best, at = -np.inf, None
    for kt2 in vals[(vals >= kt2_window[0]) & (vals <= kt2_window[1])]:
        ev = np.linalg.eigvals(lin.mode_operator(kt2)[0]).real
        ev = ev[np.abs(ev) > 1e-9]
        if ev.max() > best:
            best, at = float(ev.max()), float(kt2)
    return best, at


def critical_rayleigh(lo=190.0, hi=230.0, iters=10, **kw):
    """onset of convection of the DISCRETE operator on the protocol's grid, in the reference's units (Ra_classical = 8 Ra: SURVEY P5)"""
    at = None
    for _ in range(iters):
        mid = 0.5 * (lo + hi)
        s, k = growth_rate(mid, **kw)
        if s > 0:
            hi, at = mid, k
        else:
            lo = mid
    return 0.5 * (lo + hi), at


if __name__ == "__main__":
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ref = np.load(os.path.join(root, "tests", "golden", "flowstats_ref_series.npz"))
    ens = np.load(sys.argv[1]) if len(sys.argv) > 1 else None
    np.set_printoptions(linewidth=200, precision=4, suppress=True)
    for i, ra in enumerate(ref["ra"]):
        if ra < 4000:
            continue
        lin = LinearRBC3D(ra)
        th = lin.nusselt_series(8, 0.02, 50)
        print(f"Ra={ra:9.0f} sigma_max={lin.sigma_max:.4f}")
        print("   theory Nu-1 :", th)
        print("   ref    Nu-1 :", ref["nusselt"][i, :8] - 1)
        if ens is not None:
            print("   build  Nu-1 :", ens["nusselt"][i, :, :8].mean(0) - 1)
