// rbc3d_kernels_body.hpp -- the precision-dependent kernels of the streaming path; included by rbc3d_kernels.hpp once with
// RBC3_NS = rbc3, RBC3_REAL = double and once with RBC3_NS = rbc3f, RBC3_REAL = float (no include guard on purpose).
namespace RBC3_NS {

using namespace rbc3c;
using real = RBC3_REAL;
using f64 = double;
using real2 = HIP_vector_type<real, 2>;
__host__ __device__ __forceinline__ real2 make_real2(real x, real y) { real2 r; r.x = x; r.y = y; return r; }


using rbc::left3; using rbc::left5; using rbc::right3; using rbc::right5; using rbc::sym4;


__device__ __forceinline__ real upw(real vel, real L, real R) { return vel * (vel > real(0.0) ? L : R); }
// vel * (the left- or the right-biased fifth-order reconstruction at the face between c | d; a..f = psi[-3..+2]).  float32: both
// reconstructions as ONE chain of packed instructions -- lane .x the left-biased, lane .y the right-biased one, every tap a
// v_pk_fma_f32 with the value broadcast to both lanes (op_sel) and the two weights in an SGPR pair: six packed instructions instead
// of ten scalar ones.  The float32 tile kernels are bound by VALU issue, and these reconstructions are most of their arithmetic.
template <class R>
__device__ __forceinline__ R upw5(R vel, R a, R b, R c, R d, R e, R f)
{
    if constexpr (std::is_same<R, float>::value) {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 acc = v2{2.0f / 60.0f, 0.0f} * v2{a, a};
        acc += v2{-13.0f / 60.0f, -3.0f / 60.0f} * v2{b, b};
        acc += v2{47.0f / 60.0f, 27.0f / 60.0f} * v2{c, c};
        acc += v2{27.0f / 60.0f, 47.0f / 60.0f} * v2{d, d};
        acc += v2{-3.0f / 60.0f, -13.0f / 60.0f} * v2{e, e};
        acc += v2{0.0f, 2.0f / 60.0f} * v2{f, f};
        return vel * (vel > 0.0f ? acc.x : acc.y);
    } else return upw(vel, left5(a, b, c, d, e), right5(b, c, d, e, f));
}


// z (Bounded, N cells).  centre field -> face k: p[j] = psi[k-3+j], j=0..5 (face between p[2] | p[3])
__device__ __forceinline__ real zfL(const real *p, int k, int N)
{ const int k1 = k + 1; return (k1 >= 4 && k1 <= N - 2) ? left5(p[0], p[1], p[2], p[3], p[4]) : ((k1 >= 3 && k1 <= N - 1) ? left3(p[1], p[2], p[3]) : p[2]); }
__device__ __forceinline__ real zfR(const real *p, int k, int N)
{ const int k1 = k + 1; return (k1 >= 4 && k1 <= N - 2) ? right5(p[1], p[2], p[3], p[4], p[5]) : ((k1 >= 3 && k1 <= N - 1) ? right3(p[2], p[3], p[4]) : p[3]); }
__device__ __forceinline__ real zfS(const real *p, int k, int N)
{ const int k1 = k + 1; return (k1 >= 4 && k1 <= N - 2) ? sym4(p[1], p[2], p[3], p[4]) : real(0.5) * (p[2] + p[3]); }
// the same from a four-level window p4[j] = psi[k-2+j] (all zfS reads)
__device__ __forceinline__ real zfS4(const real *p4, int k, int N)
{ const int k1 = k + 1; return (k1 >= 4 && k1 <= N - 2) ? sym4(p4[0], p4[1], p4[2], p4[3]) : real(0.5) * (p4[1] + p4[2]); }
// face field -> centre k: p[j] = psi_face[k-2+j], j=0..5 (centre between p[2] | p[3])
__device__ __forceinline__ real zcL(const real *p, int k, int N)
{ const int k1 = k + 1; return (k1 >= 3 && k1 <= N - 2) ? left5(p[0], p[1], p[2], p[3], p[4]) : ((k1 >= 2 && k1 <= N - 1) ? left3(p[1], p[2], p[3]) : p[2]); }
__device__ __forceinline__ real zcR(const real *p, int k, int N)
{ const int k1 = k + 1; return (k1 >= 3 && k1 <= N - 2) ? right5(p[1], p[2], p[3], p[4], p[5]) : ((k1 >= 2 && k1 <= N - 1) ? right3(p[2], p[3], p[4]) : p[3]); }
__device__ __forceinline__ real zcS(const real *p, int k, int N)
{ const int k1 = k + 1; return (k1 >= 3 && k1 <= N - 2) ? sym4(p[1], p[2], p[3], p[4]) : real(0.5) * (p[2] + p[3]); }

__device__ __forceinline__ real l5a(const real *p) { return left5(p[0], p[1], p[2], p[3], p[4]); }
__device__ __forceinline__ real r5a(const real *p) { return right5(p[1], p[2], p[3], p[4], p[5]); }
// accessor of one env's state buffer
struct Fields {
    const real *b, *u, *v, *w;
    int nx, ny, nz;
    __device__ __forceinline__ size_t at(int i, int j, int k) const { return ((size_t)k * ny + j) * nx + i; }
};



// ---- hydrostatic pressure anomaly: thread per column --------------------------------------------
__global__ void k3_hydrostatic(Geo3 g, const real *state, real *phy, int B)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int ncol = g.nx * g.ny;
    if (col >= ncol * B) return;
    const int env = col / ncol, ij = col - env * ncol;
    const real *b = state + (size_t)env * g.env_stride;
    real *p = phy + (size_t)env * g.nc;
    const real hz = g.dz / 2;
    const real cN = b[(size_t)(g.nz - 1) * ncol + ij];
    const real halo = cN + ((g.min_b - cN) / hz) * g.dz;
    real acc = -(real(0.5) * (cN + halo)) * g.dz;
    p[(size_t)(g.nz - 1) * ncol + ij] = acc;
    real up = cN;
    constexpr int BK = 8;                 // latency-bound column scan: fetch BK levels at a time
    int k = g.nz - 2;
    for (; k - BK + 1 >= 0; k -= BK) {
        real c[BK];
#pragma unroll
        for (int q = 0; q < BK; ++q) c[q] = b[(size_t)(k - q) * ncol + ij];
#pragma unroll
        for (int q = 0; q < BK; ++q) {
            acc = acc - (real(0.5) * (c[q] + up)) * g.dz;
            p[(size_t)(k - q) * ncol + ij] = acc;
            up = c[q];
        }
    }
    for (; k >= 0; --k) {
        const real c = b[(size_t)k * ncol + ij];
        acc = acc - (real(0.5) * (c + up)) * g.dz;
        p[(size_t)k * ncol + ij] = acc;
        up = c;
    }
}

// ---- tendencies + RK update; FIELD 0:u 1:v 2:w 3:b ------------------------------------------------
// cur: state read; nxt: U* written; gm: G^- read / G written (same layout as the state buffer).
template <int FIELD>
__global__ void k3_tendency(Geo3 g, const real *cur, real *nxt, real *gm, const real *phy, const f64 *actT,
                            const f64 *nu_kappa, real dt, real gam, real zet, int B, real *dbg)
{
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= g.nc * B) return;
    const int env = cell / g.nc, c0 = cell - env * g.nc;
    const int nx = g.nx, ny = g.ny, nz = g.nz;
    const int k = c0 / (nx * ny), j = (c0 - k * nx * ny) / nx, i = c0 - (k * ny + j) * nx;
    const real *sb = cur + (size_t)env * g.env_stride;
    const real *b = sb, *u = sb + g.nc, *v = sb + 2 * (size_t)g.nc, *w = sb + 3 * (size_t)g.nc;
    const real nu = nu_kappa[2 * env], ka = nu_kappa[2 * env + 1];
    const real rdx = g.rdx, rdy = g.rdy, rdz = g.rdz, dz = g.dz, hz = dz / 2;
    int xi[7], yj[7];
    wrap7(i, nx, xi); wrap7(j, ny, yj);
    const int pl = nx * ny;
    auto A = [&](const real *f, int a, int bb, int kk) -> real { return f[(size_t)kk * pl + yj[bb + 3] * nx + xi[a + 3]]; };
    // z access with clamping (values read outside the domain are never used) for centre fields
    auto Zc = [&](const real *f, int a, int bb, int kk) -> real { return A(f, a, bb, min(max(kk, 0), nz - 1)); };
    // w has nz+1 levels; beyond the walls -> clamp (unused)
    auto Zw = [&](int a, int bb, int kk) -> real { return A(w, a, bb, min(max(kk, 0), nz)); };
    // ghost values of the Value BCs (no-slip for u,v; plate temperatures for b)
    auto ghost_lo = [&](real c1, real bc) -> real { return c1 + ((c1 - bc) / hz) * (-dz); };
    auto ghost_hi = [&](real cN, real bc) -> real { return cN + ((bc - cN) / hz) * dz; };

    real G;
    real old;
    if (FIELD == 0) {
        // ---- u at (x-face i, j, k) ----
        real p[6];
        // flux_uu at centres i-1 and i
        real fe, fw;
        { real q[6]; for (int t = 0; t < 6; ++t) q[t] = A(u, t - 2, 0, k); fe = upw(sym4(q[1], q[2], q[3], q[4]), l5a(q), r5a(q)); }
        { real q[6]; for (int t = 0; t < 6; ++t) q[t] = A(u, t - 3, 0, k); fw = upw(sym4(q[1], q[2], q[3], q[4]), l5a(q), r5a(q)); }
        // flux_vu at (xf i, yf j+1) and (xf i, yf j): v interpolated in x, u biased in y
        real fn, fs;
        { const real vt = sym4(A(v, -2, 1, k), A(v, -1, 1, k), A(v, 0, 1, k), A(v, 1, 1, k));
          for (int t = 0; t < 6; ++t) p[t] = A(u, 0, t - 2, k); fn = upw(vt, l5a(p), r5a(p)); }
        { const real vt = sym4(A(v, -2, 0, k), A(v, -1, 0, k), A(v, 0, 0, k), A(v, 1, 0, k));
          for (int t = 0; t < 6; ++t) p[t] = A(u, 0, t - 3, k); fs = upw(vt, l5a(p), r5a(p)); }
        // flux_wu at (xf i, zf k+1) and (xf i, zf k)
        real ft = real(0.0), fb = real(0.0);
        if (k + 1 < nz) { const real wt = sym4(Zw(-2, 0, k + 1), Zw(-1, 0, k + 1), Zw(0, 0, k + 1), Zw(1, 0, k + 1));
                          for (int t = 0; t < 6; ++t) p[t] = Zc(u, 0, 0, k - 2 + t); ft = upw(wt, zfL(p, k + 1, nz), zfR(p, k + 1, nz)); }
        if (k > 0) { const real wt = sym4(Zw(-2, 0, k), Zw(-1, 0, k), Zw(0, 0, k), Zw(1, 0, k));
                     for (int t = 0; t < 6; ++t) p[t] = Zc(u, 0, 0, k - 3 + t); fb = upw(wt, zfL(p, k, nz), zfR(p, k, nz)); }
        const real adv = (fe - fw) * rdx + (fn - fs) * rdy + (ft - fb) * rdz;
        const real u0 = A(u, 0, 0, k);
        const real uup = (k + 1 < nz) ? A(u, 0, 0, k + 1) : ghost_hi(u0, real(0.0));
        const real udn = (k > 0) ? A(u, 0, 0, k - 1) : ghost_lo(u0, real(0.0));
        const real dwt = (k + 1 < nz) ? (Zw(0, 0, k + 1) - Zw(-1, 0, k + 1)) : real(0.0);
        const real dwb = (k > 0) ? (Zw(0, 0, k) - Zw(-1, 0, k)) : real(0.0);
        const real vis = nu * (real(2.0) * ((A(u, 1, 0, k) - u0) - (u0 - A(u, -1, 0, k))) * rdx * rdx
                                 + (((A(u, 0, 1, k) - u0) * rdy + (A(v, 0, 1, k) - A(v, -1, 1, k)) * rdx)
                                    - ((u0 - A(u, 0, -1, k)) * rdy + (A(v, 0, 0, k) - A(v, -1, 0, k)) * rdx)) * rdy
                                 + (((uup - u0) * rdz + dwt * rdx) - ((u0 - udn) * rdz + dwb * rdx)) * rdz);
        const real *ph = phy + (size_t)env * g.nc;
        G = vis - adv - (A(ph, 0, 0, k) - A(ph, -1, 0, k)) * rdx;
        old = u0;
    } else if (FIELD == 1) {
        // ---- v at (i, y-face j, k): mirror of u with x<->y ----
        real p[6];
        real fe, fw;   // flux_uv at (xf i+1, yf j) and (xf i, yf j): u interpolated in y, v biased in x
        { const real ut = sym4(A(u, 1, -2, k), A(u, 1, -1, k), A(u, 1, 0, k), A(u, 1, 1, k));
          for (int t = 0; t < 6; ++t) p[t] = A(v, t - 2, 0, k); fe = upw(ut, l5a(p), r5a(p)); }
        { const real ut = sym4(A(u, 0, -2, k), A(u, 0, -1, k), A(u, 0, 0, k), A(u, 0, 1, k));
          for (int t = 0; t < 6; ++t) p[t] = A(v, t - 3, 0, k); fw = upw(ut, l5a(p), r5a(p)); }
        real fn, fs;   // flux_vv at centres j and j-1
        { real q[6]; for (int t = 0; t < 6; ++t) q[t] = A(v, 0, t - 2, k); fn = upw(sym4(q[1], q[2], q[3], q[4]), l5a(q), r5a(q)); }
        { real q[6]; for (int t = 0; t < 6; ++t) q[t] = A(v, 0, t - 3, k); fs = upw(sym4(q[1], q[2], q[3], q[4]), l5a(q), r5a(q)); }
        real ft = real(0.0), fb = real(0.0);
        if (k + 1 < nz) { const real wt = sym4(Zw(0, -2, k + 1), Zw(0, -1, k + 1), Zw(0, 0, k + 1), Zw(0, 1, k + 1));
                          for (int t = 0; t < 6; ++t) p[t] = Zc(v, 0, 0, k - 2 + t); ft = upw(wt, zfL(p, k + 1, nz), zfR(p, k + 1, nz)); }
        if (k > 0) { const real wt = sym4(Zw(0, -2, k), Zw(0, -1, k), Zw(0, 0, k), Zw(0, 1, k));
                     for (int t = 0; t < 6; ++t) p[t] = Zc(v, 0, 0, k - 3 + t); fb = upw(wt, zfL(p, k, nz), zfR(p, k, nz)); }
        const real adv = (fe - fw) * rdx + (fn - fs) * rdy + (ft - fb) * rdz;
        const real v0 = A(v, 0, 0, k);
        const real vup = (k + 1 < nz) ? A(v, 0, 0, k + 1) : ghost_hi(v0, real(0.0));
        const real vdn = (k > 0) ? A(v, 0, 0, k - 1) : ghost_lo(v0, real(0.0));
        const real dwt = (k + 1 < nz) ? (Zw(0, 0, k + 1) - Zw(0, -1, k + 1)) : real(0.0);
        const real dwb = (k > 0) ? (Zw(0, 0, k) - Zw(0, -1, k)) : real(0.0);
        const real vis = nu * ((((A(u, 1, 0, k) - A(u, 1, -1, k)) * rdy + (A(v, 1, 0, k) - v0) * rdx)
                                  - ((A(u, 0, 0, k) - A(u, 0, -1, k)) * rdy + (v0 - A(v, -1, 0, k)) * rdx)) * rdx
                                 + real(2.0) * ((A(v, 0, 1, k) - v0) - (v0 - A(v, 0, -1, k))) * rdy * rdy
                                 + (((vup - v0) * rdz + dwt * rdy) - ((v0 - vdn) * rdz + dwb * rdy)) * rdz);
        const real *ph = phy + (size_t)env * g.nc;
        G = vis - adv - (A(ph, 0, 0, k) - A(ph, 0, -1, k)) * rdy;
        old = v0;
    } else if (FIELD == 2) {
        // ---- w at (i, j, z-face k); wall face k=0 never evolves ----
        const real w0 = A(w, 0, 0, k);
        old = w0;
        if (k == 0) { G = real(0.0); }
        else {
            real p[6], q[6];
            real fe, fw, fn, fs;
            // flux_uw: u interpolated in z to face k (at x-faces i+1 and i), w biased in x
            for (int t = 0; t < 6; ++t) q[t] = Zc(u, 1, 0, k - 3 + t);
            for (int t = 0; t < 6; ++t) p[t] = A(w, t - 2, 0, k);
            fe = upw(zfS(q, k, nz), l5a(p), r5a(p));
            for (int t = 0; t < 6; ++t) q[t] = Zc(u, 0, 0, k - 3 + t);
            for (int t = 0; t < 6; ++t) p[t] = A(w, t - 3, 0, k);
            fw = upw(zfS(q, k, nz), l5a(p), r5a(p));
            for (int t = 0; t < 6; ++t) q[t] = Zc(v, 0, 1, k - 3 + t);
            for (int t = 0; t < 6; ++t) p[t] = A(w, 0, t - 2, k);
            fn = upw(zfS(q, k, nz), l5a(p), r5a(p));
            for (int t = 0; t < 6; ++t) q[t] = Zc(v, 0, 0, k - 3 + t);
            for (int t = 0; t < 6; ++t) p[t] = A(w, 0, t - 3, k);
            fs = upw(zfS(q, k, nz), l5a(p), r5a(p));
            // flux_ww at centres k and k-1
            for (int t = 0; t < 6; ++t) p[t] = Zw(0, 0, k - 2 + t);
            const real ft = upw(zcS(p, k, nz), zcL(p, k, nz), zcR(p, k, nz));
            for (int t = 0; t < 6; ++t) p[t] = Zw(0, 0, k - 3 + t);
            const real fb = upw(zcS(p, k - 1, nz), zcL(p, k - 1, nz), zcR(p, k - 1, nz));
            const real adv = (fe - fw) * rdx + (fn - fs) * rdy + (ft - fb) * rdz;
            const real vis = nu * ((((A(u, 1, 0, k) - A(u, 1, 0, k - 1)) * rdz + (A(w, 1, 0, k) - w0) * rdx)
                                      - ((A(u, 0, 0, k) - A(u, 0, 0, k - 1)) * rdz + (w0 - A(w, -1, 0, k)) * rdx)) * rdx
                                     + (((A(v, 0, 1, k) - A(v, 0, 1, k - 1)) * rdz + (A(w, 0, 1, k) - w0) * rdy)
                                        - ((A(v, 0, 0, k) - A(v, 0, 0, k - 1)) * rdz + (w0 - A(w, 0, -1, k)) * rdy)) * rdy
                                     + real(2.0) * ((A(w, 0, 0, k + 1) - w0) - (w0 - A(w, 0, 0, k - 1))) * rdz * rdz);
            G = vis - adv;
        }
    } else {
        // ---- b at centre ----
        real p[6];
        real fe, fw, fn, fs, ft = real(0.0), fb = real(0.0);
        for (int t = 0; t < 6; ++t) p[t] = A(b, t - 2, 0, k); fe = upw(A(u, 1, 0, k), l5a(p), r5a(p));
        for (int t = 0; t < 6; ++t) p[t] = A(b, t - 3, 0, k); fw = upw(A(u, 0, 0, k), l5a(p), r5a(p));
        for (int t = 0; t < 6; ++t) p[t] = A(b, 0, t - 2, k); fn = upw(A(v, 0, 1, k), l5a(p), r5a(p));
        for (int t = 0; t < 6; ++t) p[t] = A(b, 0, t - 3, k); fs = upw(A(v, 0, 0, k), l5a(p), r5a(p));
        if (k + 1 < nz) { for (int t = 0; t < 6; ++t) p[t] = Zc(b, 0, 0, k - 2 + t); ft = upw(A(w, 0, 0, k + 1), zfL(p, k + 1, nz), zfR(p, k + 1, nz)); }
        if (k > 0) { for (int t = 0; t < 6; ++t) p[t] = Zc(b, 0, 0, k - 3 + t); fb = upw(A(w, 0, 0, k), zfL(p, k, nz), zfR(p, k, nz)); }
        const real adv = (fe - fw) * rdx + (fn - fs) * rdy + (ft - fb) * rdz;
        const real b0 = A(b, 0, 0, k);
        const real bup = (k + 1 < nz) ? A(b, 0, 0, k + 1) : ghost_hi(b0, g.min_b);
        const real bdn = (k > 0) ? A(b, 0, 0, k - 1) : ghost_lo(b0, bottom_T(g, actT + (size_t)env * wall_stride(g), i, j));
        const real dif = ka * (((A(b, 1, 0, k) - b0) - (b0 - A(b, -1, 0, k))) * rdx * rdx
                                 + ((A(b, 0, 1, k) - b0) - (b0 - A(b, 0, -1, k))) * rdy * rdy
                                 + ((bup - b0) - (b0 - bdn)) * rdz * rdz);
        G = dif - adv;
        old = b0;
    }
    // field offsets in the state layout: b,u,v,w
    const size_t foff = (FIELD == 3) ? 0 : ((FIELD == 0) ? (size_t)g.nc : ((FIELD == 1) ? 2 * (size_t)g.nc : 3 * (size_t)g.nc));
    const size_t o = (size_t)env * g.env_stride + foff + c0;
    if (dbg) { dbg[((size_t)env * 4 + FIELD) * g.nc + c0] = G; return; }
    // zeta^1 = 0: stage 1 must not read G^- at all (0 * NaN = NaN: a stale G^- of an env that blew up would survive
    // its reset; the reference rebuilds the model on every reset, rbc_sim3D_api.jl:52-58).  Uniform branch.
    const real gprev = (zet != real(0.0)) ? gm[o] : real(0.0);
    nxt[o] = old + dt * (gam * G + zet * gprev);
    gm[o] = G;
    if (FIELD == 2 && k == nz - 1) nxt[(size_t)env * g.env_stride + 3 * (size_t)g.nc + c0 + nx * ny] = real(0.0);   // top wall face
}


// ---- z-marching tendency kernels ---------------------------------------------------------------------
// Thread (env, chunk, j, i) walks KC3 (=4) levels upward keeping its own column's z window and the carried
// face fluxes in registers; only the horizontal neighbours of the current level are loaded (about 90 cached
// loads per cell for all four fields instead of ~380 in the cell-per-thread kernels above, which remain the
// generic / debug path).  Same arithmetic, same order of operations per cell.
#ifndef RBC_KC3
#define RBC_KC3 4
#endif
constexpr int KC3 = RBC_KC3;

template <int FIELD>
__global__ void __launch_bounds__(128) k3_tend_march(Geo3 g, const real *cur, real *nxt, real *gm, const real *phy, const f64 *actT,
                              const f64 *nu_kappa, real dt, real gam, real zet, int B)
{
    const int nx = g.nx, ny = g.ny, nz = g.nz, pl = nx * ny, nch = nz / KC3;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pl * nch * B) return;
    const int env = t / (pl * nch), r0 = t - env * pl * nch, ch = r0 / pl, ij = r0 - ch * pl, j = ij / nx, i = ij - j * nx;
    const int k0 = ch * KC3;
    const real *sb = cur + (size_t)env * g.env_stride;
    const real *b = sb, *u = sb + g.nc, *v = sb + 2 * (size_t)g.nc, *w = sb + 3 * (size_t)g.nc;
    const real nu = nu_kappa[2 * env], ka = nu_kappa[2 * env + 1];
    const real rdx = g.rdx, rdy = g.rdy, rdz = g.rdz, dz = g.dz, hz = dz / 2;
    int xi[7], yj[7];
    wrap7(i, nx, xi); wrap7(j, ny, yj);
    auto A = [&](const real *f, int a, int bb, int kk) -> real { return f[(size_t)kk * pl + yj[bb + 3] * nx + xi[a + 3]]; };
    auto Zc = [&](const real *f, int a, int bb, int kk) -> real { return A(f, a, bb, min(max(kk, 0), nz - 1)); };
    auto Zw = [&](int a, int bb, int kk) -> real { return A(w, a, bb, min(max(kk, 0), nz)); };
    auto ghost_lo = [&](real c1, real bc) -> real { return c1 + ((c1 - bc) / hz) * (-dz); };
    auto ghost_hi = [&](real cN, real bc) -> real { return cN + ((bc - cN) / hz) * dz; };
    const size_t foff = (FIELD == 3) ? 0 : ((FIELD == 0) ? (size_t)g.nc : ((FIELD == 1) ? 2 * (size_t)g.nc : 3 * (size_t)g.nc));
    const size_t ebase = (size_t)env * g.env_stride + foff;
    auto commit = [&](int k, real old, real G) {
        const size_t o = ebase + (size_t)k * pl + ij;
        const real gprev = (zet != real(0.0)) ? gm[o] : real(0.0);     // stage 1 never reads G^- (see k3_tendency)
        nxt[o] = old + dt * (gam * G + zet * gprev);
        gm[o] = G;
    };

    if (FIELD == 0 || FIELD == 1) {
        // u (FIELD 0) and v (FIELD 1) are mirror images: `a` runs along the field's own face direction
        const real *f = (FIELD == 0) ? u : v;             // advected component
        const real *o2 = (FIELD == 0) ? v : u;            // the other horizontal component
        const real rda = (FIELD == 0) ? rdx : rdy, rdb = (FIELD == 0) ? rdy : rdx;
        // accessors in (along, across) coordinates
        auto F = [&](const real *q, int da, int db, int kk) -> real { return (FIELD == 0) ? A(q, da, db, kk) : A(q, db, da, kk); };
        auto Wl = [&](int da, int db, int kk) -> real { return (FIELD == 0) ? Zw(da, db, kk) : Zw(db, da, kk); };
        const real *ph = phy + (size_t)env * g.nc;
        real win[6];
        for (int q = 0; q < 6; ++q) win[q] = F(f, 0, 0, min(max(k0 - 3 + q, 0), nz - 1));
        real fb = real(0.0), dwb = real(0.0), fdn;
        if (k0 > 0) {
            const real wm = Wl(-1, 0, k0), wc = Wl(0, 0, k0);
            fb = upw(sym4(Wl(-2, 0, k0), wm, wc, Wl(1, 0, k0)), zfL(win, k0, nz), zfR(win, k0, nz));
            dwb = wc - wm;
            fdn = win[2];
        } else fdn = ghost_lo(win[3], real(0.0));
        for (int k = k0; k < k0 + KC3; ++k) {
            for (int q = 0; q < 5; ++q) win[q] = win[q + 1];
            win[5] = F(f, 0, 0, min(k + 3, nz - 1));
            const real f0 = win[2];
            // own-direction flux at centres a-1 and a
            real q7[7];
            for (int q = 0; q < 7; ++q) q7[q] = (q == 3) ? f0 : F(f, q - 3, 0, k);
            const real fe = upw(sym4(q7[2], q7[3], q7[4], q7[5]), left5(q7[1], q7[2], q7[3], q7[4], q7[5]), right5(q7[2], q7[3], q7[4], q7[5], q7[6]));
            const real fw = upw(sym4(q7[1], q7[2], q7[3], q7[4]), left5(q7[0], q7[1], q7[2], q7[3], q7[4]), right5(q7[1], q7[2], q7[3], q7[4], q7[5]));
            // cross flux at across-faces b+1 and b: other component interpolated along `a`, f biased across
            real c7[7];
            for (int q = 0; q < 7; ++q) c7[q] = (q == 3) ? f0 : F(f, 0, q - 3, k);
            const real on_m = F(o2, -1, 1, k), on_c = F(o2, 0, 1, k), os_m = F(o2, -1, 0, k), os_c = F(o2, 0, 0, k);
            const real fn = upw(sym4(F(o2, -2, 1, k), on_m, on_c, F(o2, 1, 1, k)), left5(c7[1], c7[2], c7[3], c7[4], c7[5]), right5(c7[2], c7[3], c7[4], c7[5], c7[6]));
            const real fs = upw(sym4(F(o2, -2, 0, k), os_m, os_c, F(o2, 1, 0, k)), left5(c7[0], c7[1], c7[2], c7[3], c7[4]), right5(c7[1], c7[2], c7[3], c7[4], c7[5]));
            // vertical flux at face k+1
            real ft = real(0.0), dwt = real(0.0), fup;
            if (k + 1 < nz) {
                const real wm = Wl(-1, 0, k + 1), wc = Wl(0, 0, k + 1);
                ft = upw(sym4(Wl(-2, 0, k + 1), wm, wc, Wl(1, 0, k + 1)), zfL(win, k + 1, nz), zfR(win, k + 1, nz));
                dwt = wc - wm;
                fup = win[3];
            } else fup = ghost_hi(f0, real(0.0));
            const real adv = (fe - fw) * rda + (fn - fs) * rdb + (ft - fb) * rdz;
            const real vis = nu * (real(2.0) * ((q7[4] - f0) - (f0 - q7[2])) * rda * rda
                                     + (((c7[4] - f0) * rdb + (on_c - on_m) * rda) - ((f0 - c7[2]) * rdb + (os_c - os_m) * rda)) * rdb
                                     + (((fup - f0) * rdz + dwt * rda) - ((f0 - fdn) * rdz + dwb * rda)) * rdz);
            const real G = vis - adv - (F(ph, 0, 0, k) - F(ph, -1, 0, k)) * rda;
            commit(k, f0, G);
            fb = ft; dwb = dwt; fdn = f0;
        }
    } else if (FIELD == 2) {
        real win[6], au[6], eu[6], av[6], ev[6];   // w faces; u at x-faces i, i+1; v at y-faces j, j+1 (levels k-3..k+2)
        for (int q = 0; q < 6; ++q) {
            win[q] = Zw(0, 0, k0 - 3 + q);
            au[q] = Zc(u, 0, 0, k0 - 4 + q); eu[q] = Zc(u, 1, 0, k0 - 4 + q);
            av[q] = Zc(v, 0, 0, k0 - 4 + q); ev[q] = Zc(v, 0, 1, k0 - 4 + q);
        }
        // flux_ww at centre k0-1: needs faces k0-3..k0+2 = win
        real fb = (k0 > 0) ? upw(zcS(win, k0 - 1, nz), zcL(win, k0 - 1, nz), zcR(win, k0 - 1, nz)) : real(0.0);
        for (int k = k0; k < k0 + KC3; ++k) {
            for (int q = 0; q < 5; ++q) { win[q] = win[q + 1]; au[q] = au[q + 1]; eu[q] = eu[q + 1]; av[q] = av[q + 1]; ev[q] = ev[q + 1]; }
            win[5] = Zw(0, 0, k + 3);                       // faces k-2..k+3 (centre k between win[2]|win[3])
            au[5] = Zc(u, 0, 0, k + 2); eu[5] = Zc(u, 1, 0, k + 2); av[5] = Zc(v, 0, 0, k + 2); ev[5] = Zc(v, 0, 1, k + 2);   // levels k-3..k+2
            const real w0 = win[2];
            const real ft = upw(zcS(win, k, nz), zcL(win, k, nz), zcR(win, k, nz));
            if (k == 0) { fb = ft; continue; }              // wall face: never evolves (nxt keeps 0)
            real q7[7], c7[7];
            for (int q = 0; q < 7; ++q) { q7[q] = (q == 3) ? w0 : A(w, q - 3, 0, k); c7[q] = (q == 3) ? w0 : A(w, 0, q - 3, k); }
            const real fe = upw(zfS(eu, k, nz), left5(q7[1], q7[2], q7[3], q7[4], q7[5]), right5(q7[2], q7[3], q7[4], q7[5], q7[6]));
            const real fw = upw(zfS(au, k, nz), left5(q7[0], q7[1], q7[2], q7[3], q7[4]), right5(q7[1], q7[2], q7[3], q7[4], q7[5]));
            const real fn = upw(zfS(ev, k, nz), left5(c7[1], c7[2], c7[3], c7[4], c7[5]), right5(c7[2], c7[3], c7[4], c7[5], c7[6]));
            const real fs = upw(zfS(av, k, nz), left5(c7[0], c7[1], c7[2], c7[3], c7[4]), right5(c7[1], c7[2], c7[3], c7[4], c7[5]));
            const real adv = (fe - fw) * rdx + (fn - fs) * rdy + (ft - fb) * rdz;
            const real vis = nu * ((((eu[3] - eu[2]) * rdz + (q7[4] - w0) * rdx) - ((au[3] - au[2]) * rdz + (w0 - q7[2]) * rdx)) * rdx
                                     + (((ev[3] - ev[2]) * rdz + (c7[4] - w0) * rdy) - ((av[3] - av[2]) * rdz + (w0 - c7[2]) * rdy)) * rdy
                                     + real(2.0) * ((win[3] - w0) - (w0 - win[1])) * rdz * rdz);
            commit(k, w0, vis - adv);
            fb = ft;
        }
        if (k0 + KC3 == nz) nxt[(size_t)env * g.env_stride + 3 * (size_t)g.nc + (size_t)nz * pl + ij] = real(0.0);   // top wall face
        if (k0 == 0) { const size_t o = ebase + ij; nxt[o] = real(0.0); gm[o] = real(0.0); }                                // bottom wall face
    } else {
        real win[6];
        for (int q = 0; q < 6; ++q) win[q] = Zc(b, 0, 0, k0 - 3 + q);
        real fb = (k0 > 0) ? upw(A(w, 0, 0, k0), zfL(win, k0, nz), zfR(win, k0, nz)) : real(0.0);
        real bdn = (k0 > 0) ? win[2] : ghost_lo(win[3], bottom_T(g, actT + (size_t)env * wall_stride(g), i, j));
        for (int k = k0; k < k0 + KC3; ++k) {
            for (int q = 0; q < 5; ++q) win[q] = win[q + 1];
            win[5] = Zc(b, 0, 0, k + 3);
            const real b0 = win[2];
            real q7[7], c7[7];
            for (int q = 0; q < 7; ++q) { q7[q] = (q == 3) ? b0 : A(b, q - 3, 0, k); c7[q] = (q == 3) ? b0 : A(b, 0, q - 3, k); }
            const real fe = upw(A(u, 1, 0, k), left5(q7[1], q7[2], q7[3], q7[4], q7[5]), right5(q7[2], q7[3], q7[4], q7[5], q7[6]));
            const real fw = upw(A(u, 0, 0, k), left5(q7[0], q7[1], q7[2], q7[3], q7[4]), right5(q7[1], q7[2], q7[3], q7[4], q7[5]));
            const real fn = upw(A(v, 0, 1, k), left5(c7[1], c7[2], c7[3], c7[4], c7[5]), right5(c7[2], c7[3], c7[4], c7[5], c7[6]));
            const real fs = upw(A(v, 0, 0, k), left5(c7[0], c7[1], c7[2], c7[3], c7[4]), right5(c7[1], c7[2], c7[3], c7[4], c7[5]));
            real ft = real(0.0), bup;
            if (k + 1 < nz) { ft = upw(A(w, 0, 0, k + 1), zfL(win, k + 1, nz), zfR(win, k + 1, nz)); bup = win[3]; }
            else bup = ghost_hi(b0, g.min_b);
            const real adv = (fe - fw) * rdx + (fn - fs) * rdy + (ft - fb) * rdz;
            const real dif = ka * (((q7[4] - b0) - (b0 - q7[2])) * rdx * rdx + ((c7[4] - b0) - (b0 - c7[2])) * rdy * rdy
                                     + ((bup - b0) - (b0 - bdn)) * rdz * rdz);
            commit(k, b0, dif - adv);
            fb = ft; bdn = b0;
        }
    }
}

// ---- LDS-tiled tendency kernels ------------------------------------------------------------------------
// A workgroup owns a (all x) x (TY rows of y) tile of one env and marches KT levels upward.  At every level the
// horizontal planes the stencils reach (TY + 6 rows: periodic halo of 3 in y; x wraps inside the row) are staged
// once in LDS by the whole group -- fetched into registers one level ahead so the loads fly under the arithmetic --
// and every cross-column stencil value is an LDS read; each thread keeps its own column's z windows and the
// carried face fluxes in registers.  Two kernels: (u, v) share the planes u, v, w(k+1); (w, b) the planes w, b.
// About 21 global loads per cell for all four fields (marching kernels: ~90, cell-per-thread: ~380).
// Buoyancy: Oceananigans splits off a hydrostatic pressure anomaly pHY' (d pHY'/dz = b at the w faces) and puts
// -grad_h pHY' into G_u, G_v and nothing into G_w ([OC] update_hydrostatic_pressure.jl; the z-marching and cell-per-thread
// kernels above and the oracle do exactly that).  The tiled kernels use the un-split form instead -- +b_face in G_w, no pHY'
// anywhere: the two tendency fields differ by the discrete gradient of pHY' with its wall-normal components dropped, which
// is exactly what the pressure projection removes (its potential becomes phi - pHY'), so U after every stage is the same to
// round-off (tests/test_gpu_parity3d.py holds it to the oracle at 1e-11) while the column scan kernel, its 3 loads per
// cell in (u, v) and its store disappear from the stage.  store_g = 0 in the last stage of a substep, whose tendencies
// are never read again (zeta^1 = 0).
constexpr int NXP3 = 64;               // padded row length of an LDS plane (nx <= 64): row/plane offsets become immediates

// -DRBC_EXPERIMENT_HALF_FLUX=1: timing bound only (WRONG numerics, never shipped): the west / south face fluxes of every field are not
// computed -- what sharing each horizontal face flux between the two cells it separates could save at most if the exchange were free.
// Measured at configs[4] (NOTES.md R4): float64 +3 %, float32 +7.6 %.  Not worth an LDS exchange and a third barrier per level.
#ifndef RBC_EXPERIMENT_HALF_FLUX
#define RBC_EXPERIMENT_HALF_FLUX 0
#endif

// One env's [b | u | v | w] block of a state-sized buffer (state, next state or G^-) as the tile kernels address it: a UNIFORM
// element offset (field + level, computed on the scalar unit) plus a 32-bit per-lane BYTE offset (own column, or the plane element
// a thread stages).  BUF = false (shipped): plain pointer arithmetic; hipcc folds `base + column` into a 64-bit VGPR pair and spends
// one v_lshl_add_u64 per access on the uniform part (99 of them in the configs[4] tile kernel: a third of its non-fp64 VALU
// instructions).  BUF = true (-DRBC_TILE_BUFFER_ADDR=1): through a buffer descriptor that covers exactly this env's
// block -- `buffer_load_dwordx2 v, v_off, s[rsrc:rsrc+3], s_off offen`, NO vector instruction per access: 1 v_lshl_add_u64 left in
// the kernel, 166 -> 141 VGPRs (float32: 130 -> 100).  Measured on one box, three interleaved repeats (scripts/ab_rate3d.sh,
// configs[4]): float64 6.32k against 6.36k env-steps/s (noise), float32 9.16k against 9.48k (-3.4 %).  The address arithmetic was
// never what the kernel waits for (it is bound by operand latency and the two barriers of a level, DESIGN.md section 3b, NOTES.md section 5b); the
// descriptor path is kept as a build flag for the record, off.
#ifndef RBC_TILE_BUFFER_ADDR
#define RBC_TILE_BUFFER_ADDR 0
#endif
__device__ __forceinline__ void buf_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff, double &x) { x = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0)); }
__device__ __forceinline__ void buf_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff, float &x) { x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, int voff, int soff, double x)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(decltype(__builtin_amdgcn_raw_buffer_load_b64(r, 0, 0, 0)), x), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_st(__amdgpu_buffer_rsrc_t r, int voff, int soff, float x) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), r, voff, soff, 0); }

template <bool BUF>
struct EnvMem {
    real *p;
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ EnvMem(const real *base, size_t elems) : p(const_cast<real *>(base))
    {
        if constexpr (BUF) r = __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)(elems * sizeof(real)), 0x00020000);
    }
    __device__ __forceinline__ real ld(size_t off, unsigned boff) const
    {
        if constexpr (BUF) { real x; buf_ld(r, (int)boff, (int)(off * sizeof(real)), x); return x; }
        else return *reinterpret_cast<const real *>(reinterpret_cast<const char *>(p + off) + boff);
    }
    __device__ __forceinline__ void st(size_t off, unsigned boff, real x) const
    {
        if constexpr (BUF) buf_st(r, (int)boff, (int)(off * sizeof(real)), x);
        else *reinterpret_cast<real *>(reinterpret_cast<char *>(p + off) + boff) = x;
    }
};

// Diagnostic build only (-DRBC_STAMPS=1, never shipped; scripts/tile_stamps.py): wave 0 of every tile workgroup accumulates
// s_memtime ticks per phase of its level loop and adds them to stamps[16 * body + id] (the kernel's last argument carries the buffer, null in the shipped build).
#undef TSTAMP
#undef TSTAMP_INIT
#undef TSTAMP_FLUSH
#if RBC_STAMPS
#define TSTAMP_INIT                                                                                 \
    const bool ts_on = (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 0);                \
    unsigned int ts_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                              \
    unsigned long long ts_last = ts_on ? __builtin_amdgcn_s_memtime() : 0ull
#define TSTAMP(id)                                                                                  \
    do {                                                                                            \
        if (ts_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ts_acc[id] += (unsigned int)(t_ - ts_last); ts_last = t_; } \
    } while (0)
#define TSTAMP_FLUSH(body, stamps)                                                                  \
    do {                                                                                            \
        if (ts_on && threadIdx.x == 0 && (stamps)) {                                                \
            for (int q_ = 0; q_ < 8; ++q_) atomicAdd((unsigned long long *)(stamps) + 16 * (body) + q_, (unsigned long long)ts_acc[q_]); \
            atomicAdd((unsigned long long *)(stamps) + 16 * (body) + 15, 1ull);                     \
        }                                                                                           \
    } while (0)
#else
#define TSTAMP_INIT
#define TSTAMP(id)
#define TSTAMP_FLUSH(body, stamps)
#endif

// FLAT (streaming-2D mode, ny = 1): a plane is ONE row of up to NXP values (no y halo), every y offset of a stencil read
// aliases that row, and everything that involves v or a y-difference is compiled out (it is identically zero).
// NXC > 0: nx (and, NYC, ny) known at compile time -- the index arithmetic of the plane staging (idx / nx, the row wrap) turns
// into multiplications by constants instead of 32-bit division sequences (about 17 VALU instructions each).
template <int TY3, int KT3, int NXP = NXP3, bool FLAT = false, int NXC = 0, int NYC = 0>
struct TileGeo {
    static constexpr int ROWS3 = FLAT ? 1 : TY3 + 6;
    static constexpr int PLANE3 = ROWS3 * NXP;
    static constexpr int NXPAD = NXP;
    static constexpr bool IS_FLAT = FLAT;
    int nx, ny, nz, pl, rows, plane, tiles, chunks, env, j0, k0, i, jl, j, tid, nthreads;
    __device__ __forceinline__ TileGeo(const Geo3 &g, int blk)
    {
        nx = NXC ? NXC : g.nx; ny = NYC ? NYC : g.ny; nz = g.nz; pl = nx * ny; rows = ROWS3; plane = rows * nx;
        tiles = ny / TY3; chunks = nz / KT3;
        const int zc = blk % chunks, yt = (blk / chunks) % tiles;
        env = blk / (chunks * tiles); j0 = yt * TY3; k0 = zc * KT3;
        tid = threadIdx.x; nthreads = blockDim.x; jl = tid / nx; i = tid - jl * nx; j = j0 + jl;
    }
    // global offset (inside one level) of tile element idx = row * nx + column
    __device__ __forceinline__ int src(int idx) const
    {
        if (FLAT) return idx;
        const int r = idx / nx, c = idx - r * nx;
        int jy = j0 - 3 + r; jy += (jy < 0) ? ny : 0; jy -= (jy >= ny) ? ny : 0;
        return jy * nx + c;
    }
};

template <int NPF, class TG, class MEM>
__device__ __forceinline__ void tile_fetch(const TG &t, const MEM &m, size_t lev, real (&pf)[NPF])
{
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
        const int idx = t.tid + q * t.nthreads;
        pf[q] = (idx < t.plane) ? m.ld(lev, (unsigned)t.src(idx) * (unsigned)sizeof(real)) : real(0.0);
    }
}
template <int NPF, class TG>
__device__ __forceinline__ void tile_store(const TG &t, real *dst, const real (&pf)[NPF])
{
#pragma unroll
    for (int q = 0; q < NPF; ++q) {
        const int idx = t.tid + q * t.nthreads;
        if (idx < t.plane) {
            if (TG::IS_FLAT) dst[idx] = pf[q];
            else { const int r = idx / t.nx, c = idx - r * t.nx; dst[r * TG::NXPAD + c] = pf[q]; }
        }
    }
}

// (u, v): blockDim = nx * TY3, B * (ny/TY3) * (nz/KT3) workgroups, LDS = 3 planes.  Two shapes are built:
// 16 rows x 4 levels (768 threads = 12 waves, three per SIMD, register budget 168) where ny % 16 == 0, else 8 x 8
// (up to 512 threads, budget 256); the first is 3 % faster at 48 x 48 x 32 (smaller halo share, even SIMD load).
template <int TY3, int KT3, int NPF, int NXP = NXP3, bool FLAT = false, int NXC = 0, int NYC = 0>
__device__ __forceinline__ void tile_uv_body(const Geo3 &g, const real *cur, real *nxt, real *gm,
                                             const f64 *nu_kappa, real dt, real gam, real zet, int store_g, int blk,
                                             const real *stamps = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) real tile_sm[];
    const TileGeo<TY3, KT3, NXP, FLAT, NXC, NYC> t(g, blk);
    constexpr int PLANE3 = TileGeo<TY3, KT3, NXP, FLAT, NXC, NYC>::PLANE3;
    const int nx = t.nx, nz = t.nz, pl = t.pl;
    real *PU = tile_sm, *PV = tile_sm + PLANE3, *PW = tile_sm + 2 * PLANE3;        // u(k), v(k), w(k+1)
    constexpr int IU = 0, IV = PLANE3, IW = 2 * PLANE3;
    const size_t eb = (size_t)t.env * g.env_stride;
    const EnvMem<(!FLAT && RBC_TILE_BUFFER_ADDR)> S(cur + eb, g.env_stride), N(nxt + eb, g.env_stride), G(gm + eb, g.env_stride);      // state, next state, G^- of this env
    const size_t u = g.nc, v = 2 * (size_t)g.nc, w = 3 * (size_t)g.nc;      // element offsets of the fields inside an env's block
    const real nu = nu_kappa[2 * t.env];
    const real rdx = g.rdx, rdy = g.rdy, rdz = g.rdz, dz = g.dz, hz = dz / 2;
    int xi[7];
    wrap7(t.i, nx, xi);
    const int col = t.j * nx + t.i;                               // own column inside a level
    // one LDS address per x offset; the plane and the y offset are immediates of the ds_read
    const real *xb[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) xb[a] = tile_sm + (FLAT ? 0 : (t.jl + 3) * NXP) + xi[a];
    auto L = [&](int P, int a, int b) -> real { return xb[a + 3][P + (FLAT ? 0 : b * NXP)]; };
    auto ghost_lo = [&](real c1, real bc) -> real { return c1 + ((c1 - bc) / hz) * (-dz); };
    auto ghost_hi = [&](real cN, real bc) -> real { return cN + ((bc - cN) / hz) * dz; };
    const unsigned colb = (unsigned)col * (unsigned)sizeof(real);       // byte offset of the own column inside a level
    auto own = [&](size_t f, int kk) -> real { return S.ld(f + (size_t)min(max(kk, 0), nz - 1) * pl, colb); };

    // stage level k0 (u, v) and k0+1 (w; level k0 itself is only needed for the chunk's bottom face, read below).  Every
    // global load of the prologue is issued here, before the first barrier, so that the workgroup waits for memory once
    // instead of once per barrier (a workgroup lives for four levels: an exposed round trip is a tenth of its life).
    const bool use_gm = (zet != real(0.0));                             // stage 1 never reads G^- (see k3_tendency)
    real pfu[NPF], pfv[NPF], pfw[NPF], pfw1[NPF];
    tile_fetch(t, S, u + (size_t)t.k0 * pl, pfu);
    if constexpr (!FLAT) tile_fetch(t, S, v + (size_t)t.k0 * pl, pfv);
    tile_fetch(t, S, w + (size_t)t.k0 * pl, pfw);
    tile_fetch(t, S, w + (size_t)min(t.k0 + 1, nz) * pl, pfw1);
    real winu[6], winv[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) { winu[q] = own(u, t.k0 - 3 + q); winv[q] = FLAT ? real(0.0) : own(v, t.k0 - 3 + q); }
    // per-level global operands of this thread travel one level ahead of their use, like the planes
    real nu5 = own(u, t.k0 + 3), nv5 = FLAT ? real(0.0) : own(v, t.k0 + 3);
    real ngu = use_gm ? G.ld(u + (size_t)t.k0 * pl, colb) : real(0.0), ngv = (use_gm && !FLAT) ? G.ld(v + (size_t)t.k0 * pl, colb) : real(0.0);
    tile_store(t, PW, pfw);                                       // w(k0) first: bottom-face terms of the chunk
    __syncthreads();
    real fbu = real(0.0), dwbu = real(0.0), fdnu, fbv = real(0.0), dwbv = real(0.0), fdnv;
    if (t.k0 > 0) {
        const real wc = L(IW, 0, 0), wmx = L(IW, -1, 0);
        fbu = upw(sym4(L(IW, -2, 0), wmx, wc, L(IW, 1, 0)), zfL(winu, t.k0, nz), zfR(winu, t.k0, nz));
        dwbu = wc - wmx;
        if constexpr (!FLAT) {
            const real wmy = L(IW, 0, -1);
            fbv = upw(sym4(L(IW, 0, -2), wmy, wc, L(IW, 0, 1)), zfL(winv, t.k0, nz), zfR(winv, t.k0, nz));
            dwbv = wc - wmy;
        }
        fdnu = winu[2]; fdnv = winv[2];
    } else { fdnu = ghost_lo(winu[3], real(0.0)); fdnv = ghost_lo(winv[3], real(0.0)); }
    __syncthreads();
    tile_store(t, PU, pfu);
    if constexpr (!FLAT) tile_store(t, PV, pfv);
    tile_store(t, PW, pfw1);
    __syncthreads();

    TSTAMP_INIT;
    for (int k = t.k0; k < t.k0 + KT3; ++k) {
        const bool more = (k + 1 < t.k0 + KT3);
        TSTAMP(6);
#pragma unroll
        for (int q = 0; q < 5; ++q) { winu[q] = winu[q + 1]; winv[q] = winv[q + 1]; }
        winu[5] = nu5; winv[5] = nv5;
        const real gpu_ = ngu, gpv_ = ngv;
        if (more) {                                               // next level's planes and operands fly under this level's arithmetic
            tile_fetch(t, S, u + (size_t)(k + 1) * pl, pfu);
            if constexpr (!FLAT) tile_fetch(t, S, v + (size_t)(k + 1) * pl, pfv);
            tile_fetch(t, S, w + (size_t)min(k + 2, nz) * pl, pfw);
            nu5 = own(u, k + 4);
            if constexpr (!FLAT) nv5 = own(v, k + 4);
            if (use_gm) { ngu = G.ld(u + (size_t)(k + 1) * pl, colb); if constexpr (!FLAT) ngv = G.ld(v + (size_t)(k + 1) * pl, colb); }
        }
        const bool top = (k + 1 >= nz);
        const real wc = top ? real(0.0) : L(IW, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        TSTAMP(0);
        // ---- u at (x-face i, j, k): `a` along x, `b` along y ----
        {
            const real f0 = winu[2];
            real q7[7], c7[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) { q7[q] = (q == 3) ? f0 : L(IU, q - 3, 0); c7[q] = (q == 3 || FLAT) ? f0 : L(IU, 0, q - 3); }
            const real fe = upw5<real>(sym4(q7[2], q7[3], q7[4], q7[5]), q7[1], q7[2], q7[3], q7[4], q7[5], q7[6]);
#if RBC_EXPERIMENT_HALF_FLUX
            const real fw = fe * real(0.5);      /* timing bound only, WRONG numerics */
#else
            const real fw = upw5<real>(sym4(q7[1], q7[2], q7[3], q7[4]), q7[0], q7[1], q7[2], q7[3], q7[4], q7[5]);
#endif
            real on_m = real(0.0), on_c = real(0.0), os_m = real(0.0), os_c = real(0.0), fn = real(0.0), fs = real(0.0);      // FLAT: v == 0, no y fluxes
            if constexpr (!FLAT) {
                on_m = L(IV, -1, 1); on_c = L(IV, 0, 1); os_m = L(IV, -1, 0); os_c = winv[2];
                fn = upw5<real>(sym4(L(IV, -2, 1), on_m, on_c, L(IV, 1, 1)), c7[1], c7[2], c7[3], c7[4], c7[5], c7[6]);
#if RBC_EXPERIMENT_HALF_FLUX
                fs = fn * real(0.5);
#else
                fs = upw5<real>(sym4(L(IV, -2, 0), os_m, os_c, L(IV, 1, 0)), c7[0], c7[1], c7[2], c7[3], c7[4], c7[5]);
#endif
            }
            real ft = real(0.0), dwt = real(0.0), fup;
            if (!top) {
                const real wm = L(IW, -1, 0);
                ft = upw(sym4(L(IW, -2, 0), wm, wc, L(IW, 1, 0)), zfL(winu, k + 1, nz), zfR(winu, k + 1, nz));
                dwt = wc - wm; fup = winu[3];
            } else fup = ghost_hi(f0, real(0.0));
            const real adv = (fe - fw) * rdx + (fn - fs) * rdy + (ft - fbu) * rdz;
            const real vis = nu * (real(2.0) * ((q7[4] - f0) - (f0 - q7[2])) * rdx * rdx
                                     + (((c7[4] - f0) * rdy + (on_c - on_m) * rdx) - ((f0 - c7[2]) * rdy + (os_c - os_m) * rdx)) * rdy
                                     + (((fup - f0) * rdz + dwt * rdx) - ((f0 - fdnu) * rdz + dwbu * rdx)) * rdz);
            const real Gn = vis - adv;                         // buoyancy sits in G_w (see the header comment of these kernels)
            const size_t o = u + (size_t)k * pl;
            N.st(o, colb, f0 + dt * (gam * Gn + zet * gpu_));
            if (store_g) G.st(o, colb, Gn);
            fbu = ft; dwbu = dwt; fdnu = f0;
        }
        __builtin_amdgcn_sched_barrier(0);           // keep the two sections' live ranges apart
        TSTAMP(1);
        // ---- v at (i, y-face j, k): the mirror image, `a` along y, `b` along x ----
        if constexpr (!FLAT) {
            const real f0 = winv[2];
            real q7[7], c7[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) { q7[q] = (q == 3) ? f0 : L(IV, 0, q - 3); c7[q] = (q == 3) ? f0 : L(IV, q - 3, 0); }
            const real fe = upw5<real>(sym4(q7[2], q7[3], q7[4], q7[5]), q7[1], q7[2], q7[3], q7[4], q7[5], q7[6]);
#if RBC_EXPERIMENT_HALF_FLUX
            const real fw = fe * real(0.5);      /* timing bound only, WRONG numerics */
#else
            const real fw = upw5<real>(sym4(q7[1], q7[2], q7[3], q7[4]), q7[0], q7[1], q7[2], q7[3], q7[4], q7[5]);
#endif
            const real on_m = L(IU, 1, -1), on_c = L(IU, 1, 0), os_m = L(IU, 0, -1), os_c = winu[2];
            const real fn = upw5<real>(sym4(L(IU, 1, -2), on_m, on_c, L(IU, 1, 1)), c7[1], c7[2], c7[3], c7[4], c7[5], c7[6]);
#if RBC_EXPERIMENT_HALF_FLUX
            const real fs = fn * real(0.5);
#else
            const real fs = upw5<real>(sym4(L(IU, 0, -2), os_m, os_c, L(IU, 0, 1)), c7[0], c7[1], c7[2], c7[3], c7[4], c7[5]);
#endif
            real ft = real(0.0), dwt = real(0.0), fup;
            if (!top) {
                const real wm = L(IW, 0, -1);
                ft = upw(sym4(L(IW, 0, -2), wm, wc, L(IW, 0, 1)), zfL(winv, k + 1, nz), zfR(winv, k + 1, nz));
                dwt = wc - wm; fup = winv[3];
            } else fup = ghost_hi(f0, real(0.0));
            const real adv = (fe - fw) * rdy + (fn - fs) * rdx + (ft - fbv) * rdz;
            const real vis = nu * (real(2.0) * ((q7[4] - f0) - (f0 - q7[2])) * rdy * rdy
                                     + (((c7[4] - f0) * rdx + (on_c - on_m) * rdy) - ((f0 - c7[2]) * rdx + (os_c - os_m) * rdy)) * rdx
                                     + (((fup - f0) * rdz + dwt * rdy) - ((f0 - fdnv) * rdz + dwbv * rdy)) * rdz);
            const real Gn = vis - adv;
            const size_t o = v + (size_t)k * pl;
            N.st(o, colb, f0 + dt * (gam * Gn + zet * gpv_));
            if (store_g) G.st(o, colb, Gn);
            fbv = ft; dwbv = dwt; fdnv = f0;
        }
        TSTAMP(2);
        if (more) {
            __syncthreads();                                      // every read of this level's planes is done
            TSTAMP(3);
            tile_store(t, PU, pfu);
            if constexpr (!FLAT) tile_store(t, PV, pfv);
            tile_store(t, PW, pfw);
            TSTAMP(4);
            __syncthreads();
            TSTAMP(5);
        }
    }
    TSTAMP_FLUSH(0, stamps);
}

// (w, b): same shape, LDS = 2 planes (w and b at the current level)
template <int TY3, int KT3, int NPF, int NXP = NXP3, bool FLAT = false, int NXC = 0, int NYC = 0>
__device__ __forceinline__ void tile_wb_body(const Geo3 &g, const real *cur, real *nxt, real *gm, const f64 *actT,
                                             const f64 *nu_kappa, real dt, real gam, real zet, int store_g, int blk,
                                             const real *stamps = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) real tile_sm[];
    const TileGeo<TY3, KT3, NXP, FLAT, NXC, NYC> t(g, blk);
    constexpr int PLANE3 = TileGeo<TY3, KT3, NXP, FLAT, NXC, NYC>::PLANE3;
    const int nx = t.nx, nz = t.nz, pl = t.pl;
    real *PW = tile_sm, *PB = tile_sm + PLANE3;
    constexpr int IW = 0, IB = PLANE3;
    const size_t eb = (size_t)t.env * g.env_stride;
    const EnvMem<(!FLAT && RBC_TILE_BUFFER_ADDR)> S(cur + eb, g.env_stride), N(nxt + eb, g.env_stride), G(gm + eb, g.env_stride);      // state, next state, G^- of this env
    const size_t b = 0, u = g.nc, v = 2 * (size_t)g.nc, w = 3 * (size_t)g.nc;      // element offsets of the fields inside an env's block
    const real nu = nu_kappa[2 * t.env], ka = nu_kappa[2 * t.env + 1];
    const real rdx = g.rdx, rdy = g.rdy, rdz = g.rdz, dz = g.dz, hz = dz / 2;
    int xi[7];
    wrap7(t.i, nx, xi);
    const int col = t.j * nx + t.i;
    const int colE = t.j * nx + xi[4];                            // column (i+1, j)
    const int colN = ((t.j + 1 == t.ny) ? 0 : t.j + 1) * nx + t.i;   // column (i, j+1)
    const real *xb[7];
#pragma unroll
    for (int a = 0; a < 7; ++a) xb[a] = tile_sm + (FLAT ? 0 : (t.jl + 3) * NXP) + xi[a];
    auto L = [&](int P, int a, int bb) -> real { return xb[a + 3][P + (FLAT ? 0 : bb * NXP)]; };
    auto ghost_lo = [&](real c1, real bc) -> real { return c1 + ((c1 - bc) / hz) * (-dz); };
    auto ghost_hi = [&](real cN, real bc) -> real { return cN + ((bc - cN) / hz) * dz; };
    // byte offsets of the own column and of its east / north neighbours inside a level (see EnvMem)
    const unsigned colb = (unsigned)col * (unsigned)sizeof(real), colEb = (unsigned)colE * (unsigned)sizeof(real), colNb = (unsigned)colN * (unsigned)sizeof(real);
    auto cen = [&](size_t f, unsigned cb, int kk) -> real { return S.ld(f + (size_t)min(max(kk, 0), nz - 1) * pl, cb); };
    auto fac = [&](int kk) -> real { return S.ld(w + (size_t)min(max(kk, 0), nz) * pl, colb); };      // own-column w face

    // every global load of the prologue before the first barrier (see k3_tile_uv)
    const bool use_gm = (zet != real(0.0));                             // stage 1 never reads G^- (see k3_tendency)
    real pfw[NPF], pfb[NPF];
    tile_fetch(t, S, w + (size_t)t.k0 * pl, pfw); tile_fetch(t, S, b + (size_t)t.k0 * pl, pfb);
    // z windows: w faces and b centres six deep (5-point stencils), the advecting u, v columns four deep (levels k-2..k+1: all the
    // centred interpolation to the w face reads)
    real winw[6], winb[6], au[4], eu[4], av[4], ev[4];
#pragma unroll
    for (int q = 0; q < 6; ++q) { winw[q] = fac(t.k0 - 3 + q); winb[q] = cen(b, colb, t.k0 - 3 + q); }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        au[q] = cen(u, colb, t.k0 - 3 + q); eu[q] = cen(u, colEb, t.k0 - 3 + q);
        av[q] = FLAT ? real(0.0) : cen(v, colb, t.k0 - 3 + q); ev[q] = FLAT ? real(0.0) : cen(v, colNb, t.k0 - 3 + q);
    }
    // per-level global operands of this thread travel one level ahead of their use, like the planes
    real nw5 = fac(t.k0 + 3), nb5 = cen(b, colb, t.k0 + 3);
    real nau = cen(u, colb, t.k0 + 1), neu = cen(u, colEb, t.k0 + 1), nav = FLAT ? real(0.0) : cen(v, colb, t.k0 + 1), nev = FLAT ? real(0.0) : cen(v, colNb, t.k0 + 1);
    real ngw = use_gm ? G.ld(w + (size_t)t.k0 * pl, colb) : real(0.0), ngb = use_gm ? G.ld(b + (size_t)t.k0 * pl, colb) : real(0.0);
    real fbw = (t.k0 > 0) ? upw(zcS(winw, t.k0 - 1, nz), zcL(winw, t.k0 - 1, nz), zcR(winw, t.k0 - 1, nz)) : real(0.0);
    real fbb = (t.k0 > 0) ? upw(winw[3], zfL(winb, t.k0, nz), zfR(winb, t.k0, nz)) : real(0.0);
    real bdn = (t.k0 > 0) ? winb[2] : ghost_lo(winb[3], bottom_T(g, actT + (size_t)t.env * wall_stride(g), t.i, t.j));
    tile_store(t, PW, pfw); tile_store(t, PB, pfb);
    __syncthreads();

    TSTAMP_INIT;
    for (int k = t.k0; k < t.k0 + KT3; ++k) {
        const bool more = (k + 1 < t.k0 + KT3);
        TSTAMP(6);
#pragma unroll
        for (int q = 0; q < 5; ++q) { winw[q] = winw[q + 1]; winb[q] = winb[q + 1]; }
#pragma unroll
        for (int q = 0; q < 3; ++q) { au[q] = au[q + 1]; eu[q] = eu[q + 1]; av[q] = av[q + 1]; ev[q] = ev[q + 1]; }
        winw[5] = nw5; winb[5] = nb5;                             // w faces k-2..k+3, b centres k-2..k+3
        au[3] = nau; eu[3] = neu; av[3] = nav; ev[3] = nev;       // u, v levels k-2..k+1
        const real gpw_ = ngw, gpb_ = ngb;
        if (more) {
            tile_fetch(t, S, w + (size_t)(k + 1) * pl, pfw); tile_fetch(t, S, b + (size_t)(k + 1) * pl, pfb);
            nw5 = fac(k + 4); nb5 = cen(b, colb, k + 4);
            nau = cen(u, colb, k + 2); neu = cen(u, colEb, k + 2);
            if constexpr (!FLAT) { nav = cen(v, colb, k + 2); nev = cen(v, colNb, k + 2); }
            if (use_gm) { ngw = G.ld(w + (size_t)(k + 1) * pl, colb); ngb = G.ld(b + (size_t)(k + 1) * pl, colb); }
        }
        __builtin_amdgcn_sched_barrier(0);
        TSTAMP(0);
        // ---- w at (i, j, z-face k); the wall face k = 0 never evolves ----
        {
            const real w0 = winw[2];
            const real ft = upw(zcS(winw, k, nz), zcL(winw, k, nz), zcR(winw, k, nz));
            const size_t o = w + (size_t)k * pl;
            if (k > 0) {
                real q7[7], c7[7];
#pragma unroll
                for (int q = 0; q < 7; ++q) { q7[q] = (q == 3) ? w0 : L(IW, q - 3, 0); c7[q] = (q == 3 || FLAT) ? w0 : L(IW, 0, q - 3); }
                const real fe = upw5<real>(zfS4(eu, k, nz), q7[1], q7[2], q7[3], q7[4], q7[5], q7[6]);
#if RBC_EXPERIMENT_HALF_FLUX
            const real fw = fe * real(0.5);      /* timing bound only, WRONG numerics */
#else
                const real fw = upw5<real>(zfS4(au, k, nz), q7[0], q7[1], q7[2], q7[3], q7[4], q7[5]);
#endif
                real fn = real(0.0), fs = real(0.0);
                if constexpr (!FLAT) {
                    fn = upw5<real>(zfS4(ev, k, nz), c7[1], c7[2], c7[3], c7[4], c7[5], c7[6]);
#if RBC_EXPERIMENT_HALF_FLUX
                fs = fn * real(0.5);
#else
                    fs = upw5<real>(zfS4(av, k, nz), c7[0], c7[1], c7[2], c7[3], c7[4], c7[5]);
#endif
                }
                const real adv = (fe - fw) * rdx + (fn - fs) * rdy + (ft - fbw) * rdz;
                const real vis = nu * ((((eu[2] - eu[1]) * rdz + (q7[4] - w0) * rdx) - ((au[2] - au[1]) * rdz + (w0 - q7[2]) * rdx)) * rdx
                                         + (((ev[2] - ev[1]) * rdz + (c7[4] - w0) * rdy) - ((av[2] - av[1]) * rdz + (w0 - c7[2]) * rdy)) * rdy
                                         + real(2.0) * ((winw[3] - w0) - (w0 - winw[1])) * rdz * rdz);
                const real Gn = vis - adv + real(0.5) * (winb[1] + winb[2]);     // + b at the face: the un-split buoyancy term
                N.st(o, colb, w0 + dt * (gam * Gn + zet * gpw_));
                if (store_g) G.st(o, colb, Gn);
            } else { N.st(o, colb, real(0.0)); if (store_g) G.st(o, colb, real(0.0)); }
            fbw = ft;
        }
        __builtin_amdgcn_sched_barrier(0);
        TSTAMP(1);
        // ---- b at the centre (i, j, k) ----
        {
            const real b0 = winb[2];
            real q7[7], c7[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) { q7[q] = (q == 3) ? b0 : L(IB, q - 3, 0); c7[q] = (q == 3 || FLAT) ? b0 : L(IB, 0, q - 3); }
            const real fe = upw5<real>(eu[2], q7[1], q7[2], q7[3], q7[4], q7[5], q7[6]);
#if RBC_EXPERIMENT_HALF_FLUX
            const real fw = fe * real(0.5);      /* timing bound only, WRONG numerics */
#else
            const real fw = upw5<real>(au[2], q7[0], q7[1], q7[2], q7[3], q7[4], q7[5]);
#endif
            real fn = real(0.0), fs = real(0.0);
            if constexpr (!FLAT) {
                fn = upw5<real>(ev[2], c7[1], c7[2], c7[3], c7[4], c7[5], c7[6]);
#if RBC_EXPERIMENT_HALF_FLUX
                fs = fn * real(0.5);
#else
                fs = upw5<real>(av[2], c7[0], c7[1], c7[2], c7[3], c7[4], c7[5]);
#endif
            }
            real ft = real(0.0), bup;
            if (k + 1 < nz) { ft = upw(winw[3], zfL(winb, k + 1, nz), zfR(winb, k + 1, nz)); bup = winb[3]; }
            else bup = ghost_hi(b0, g.min_b);
            const real adv = (fe - fw) * rdx + (fn - fs) * rdy + (ft - fbb) * rdz;
            const real dif = ka * (((q7[4] - b0) - (b0 - q7[2])) * rdx * rdx + ((c7[4] - b0) - (b0 - c7[2])) * rdy * rdy
                                     + ((bup - b0) - (b0 - bdn)) * rdz * rdz);
            const real Gn = dif - adv;
            const size_t o = b + (size_t)k * pl;
            N.st(o, colb, b0 + dt * (gam * Gn + zet * gpb_));
            if (store_g) G.st(o, colb, Gn);
            fbb = ft; bdn = b0;
        }
        TSTAMP(2);
        if (more) {
            __syncthreads();
            TSTAMP(3);
            tile_store(t, PW, pfw); tile_store(t, PB, pfb);
            TSTAMP(4);
            __syncthreads();
            TSTAMP(5);
        }
    }
    TSTAMP_FLUSH(1, stamps);
    if (t.k0 + KT3 == nz) N.st(w + (size_t)nz * pl, colb, real(0.0));      // top wall face
}

// Both tendency kernels as ONE launch: the first half of the grid runs the (u, v) body, the second half the (w, b) body (they
// read the same state buffer and write disjoint fields).  A kernel boundary on the dependent stream costs about 10 us on this
// path whatever the kernels do (2.5 ms per env-step at B = 1, where all 234 launches are nearly empty); one launch fewer per
// stage is worth more here than anything done inside the kernels.
template <int TY3, int KT3, int NPF, int MAXT, int WAVES, int NXP = NXP3, bool FLAT = false, int NXC = 0, int NYC = 0>
__global__ void __launch_bounds__(MAXT, WAVES) k3_tile_all(Geo3 g, const real *cur, real *nxt, real *gm, const f64 *actT,
                                                   const f64 *nu_kappa, real dt, real gam, real zet, int store_g,
                                                   const real *stamps)
{
    const int half = gridDim.x >> 1;                 // first half of the grid: (u, v); second half: (w, b), starting as the first drains
    if ((int)blockIdx.x >= half) tile_wb_body<TY3, KT3, NPF, NXP, FLAT, NXC, NYC>(g, cur, nxt, gm, actT, nu_kappa, dt, gam, zet, store_g, (int)blockIdx.x - half, stamps);
    else tile_uv_body<TY3, KT3, NPF, NXP, FLAT, NXC, NYC>(g, cur, nxt, gm, nu_kappa, dt, gam, zet, store_g, (int)blockIdx.x, stamps);
}

// ---- generic two-factor DFT of every line of a slab held in LDS ------------------------------------
// data: [rows][n] complex (re,im interleaved), n = n1*n2, line stride `ls`, element stride `es`
// (so the same routine does rows and columns).  out-of-place src -> dst.  sign=-1 forward, +1 inverse.
__device__ inline void slab_dft(const real2 *src, real2 *dst, int lines, int n, int n1, int n2, int ls, int es,
                                const real2 *tw, int sign, real2 *tmp)
{
    // stage 1: tmp[line][k1][n2'] = W_n^(n2' k1) * sum_{a<n1} src[line][n2*a + n2'] W_n1^(a k1)
    for (int idx = threadIdx.x; idx < lines * n; idx += blockDim.x) {
        const int line = idx / n, r = idx - line * n, k1 = r / n2, b2 = r - k1 * n2;
        real sr = real(0.0), si = real(0.0);
        for (int a = 0; a < n1; ++a) {
            const real2 x = src[line * ls + (n2 * a + b2) * es];
            const real2 t = tw[((a * k1) % n1) * n2];            // W_n1^(a k1) = W_n^(n2 a k1)
            const real ti = sign < 0 ? -t.y : t.y;
            sr += x.x * t.x - x.y * ti; si += x.x * ti + x.y * t.x;
        }
        const real2 t = tw[(b2 * k1) % n];
        const real ti = sign < 0 ? -t.y : t.y;
        tmp[line * n + k1 * n2 + b2] = make_real2(sr * t.x - si * ti, sr * ti + si * t.x);
    }
    __syncthreads();
    // stage 2: dst[line][k1 + n1*k2] = sum_{b<n2} tmp[line][k1][b] W_n2^(b k2)
    for (int idx = threadIdx.x; idx < lines * n; idx += blockDim.x) {
        const int line = idx / n, r = idx - line * n, k2 = r / n1, k1 = r - k2 * n1;
        real sr = real(0.0), si = real(0.0);
        for (int b2 = 0; b2 < n2; ++b2) {
            const real2 x = tmp[line * n + k1 * n2 + b2];
            const real2 t = tw[((b2 * k2) % n2) * n1];
            const real ti = sign < 0 ? -t.y : t.y;
            sr += x.x * t.x - x.y * ti; si += x.x * ti + x.y * t.x;
        }
        dst[line * ls + (k1 + n1 * k2) * es] = make_real2(sr, si);
    }
    __syncthreads();
}

struct FftPlan {
    int nx1, nx2, ny1, ny2;
    const real2 *tw;          // twiddles e^{+2 pi i t / n}: nx values for the rows, then ny for the columns (k3_twiddles)
};

__global__ void k3_twiddles(real2 *tw, int nx, int ny)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nx + ny) return;
    const int n = (t < nx) ? nx : ny, q = (t < nx) ? t : t - nx;
    f64 s, c;                                   // computed in float64 whatever the precision of the transform
    sincospi(2.0 * q / n, &s, &c);
    tw[t] = make_real2((real)c, (real)s);
}

// ---- register-blocked slab FFT for n = N1*8 (N1 in {4,6,8}: n = 32, 48, 64) ----------------------
// Same two-factor Cooley-Tukey as slab_dft, but each work item holds a whole small DFT in registers
// (N1 loads for N1 outputs, then 8 for 8) instead of re-reading LDS for every output.
__device__ __forceinline__ void dft4(real *re, real *im)
{
    const real s0r = re[0] + re[2], s0i = im[0] + im[2], d0r = re[0] - re[2], d0i = im[0] - im[2];
    const real s1r = re[1] + re[3], s1i = im[1] + im[3], d1r = re[1] - re[3], d1i = im[1] - im[3];
    re[0] = s0r + s1r; im[0] = s0i + s1i;
    re[1] = d0r + d1i; im[1] = d0i - d1r;
    re[2] = s0r - s1r; im[2] = s0i - s1i;
    re[3] = d0r - d1i; im[3] = d0i + d1r;
}
// DFT-6 by the prime-factor map n=(3a+2b)%6, k=(3c+4d)%6 (a,c<2; b,d<3): no twiddles
__device__ __forceinline__ void dft6(real *re, real *im)
{
    real sr[3], si[3], dr[3], di[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int p0 = (2 * b) % 6, p1 = (3 + 2 * b) % 6;
        sr[b] = re[p0] + re[p1]; si[b] = im[p0] + im[p1];
        dr[b] = re[p0] - re[p1]; di[b] = im[p0] - im[p1];
    }
    rbc::dft3(sr[0], si[0], sr[1], si[1], sr[2], si[2]);
    rbc::dft3(dr[0], di[0], dr[1], di[1], dr[2], di[2]);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        re[(4 * d) % 6] = sr[d]; im[(4 * d) % 6] = si[d];
        re[(3 + 4 * d) % 6] = dr[d]; im[(3 + 4 * d) % 6] = di[d];
    }
}
// DFT-32 as two DFT-16 (even / odd samples) + one radix-2 butterfly with W32^k
__device__ __forceinline__ void dft32(real *re, real *im)
{
    real er[16], ei[16], orr[16], oi[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) { er[j] = re[2 * j]; ei[j] = im[2 * j]; orr[j] = re[2 * j + 1]; oi[j] = im[2 * j + 1]; }
    rbc::dft16(er, ei);
    rbc::dft16(orr, oi);
    // cos(2 pi k / 32), k = 0..8 (the rest by symmetry); W32^k = c[k] - i s[k]
    constexpr real c32[9] = {real(1.0), real(0.98078528040323044913), real(0.92387953251128675613), real(0.83146961230254523708), real(0.70710678118654752440),
                               real(0.55557023301960222474), real(0.38268343236508977173), real(0.19509032201612826785), real(0.0)};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const real wr = (k <= 8) ? c32[k] : -c32[16 - k], wi = -((k <= 8) ? c32[8 - k] : c32[k - 8]);
        const real tr = orr[k] * wr - oi[k] * wi, ti = orr[k] * wi + oi[k] * wr;
        re[k] = er[k] + tr; im[k] = ei[k] + ti;
        re[k + 16] = er[k] - tr; im[k + 16] = ei[k] - ti;
    }
}

template <int N1> __device__ __forceinline__ void dftN(real *re, real *im)
{
    static_assert(N1 == 4 || N1 == 6 || N1 == 8 || N1 == 12 || N1 == 16 || N1 == 24 || N1 == 32, "register-blocked slab FFT: n = 8 * {4, 6, 8, 12, 16, 24, 32}");
    if (N1 == 32) { dft32(re, im); return; }
    if (N1 == 4) dft4(re, im);
    else if (N1 == 6) dft6(re, im);
    else if (N1 == 8) rbc::dft8(re, im);
    else if (N1 == 12) rbc::dft12(re, im);
    else if (N1 == 16) rbc::dft16(re, im);
    else rbc::dft24(re, im);
}

// Lanes run along LINES in both stages (consecutive lanes = consecutive lines, all at the same element): with a line stride that
// is odd in units of 16 bytes (rows: nxp = nx | 1 complex values; columns: 1) the 16 lanes of an LDS access group fall on 16
// different bank quads, whatever the element offsets are.  Stage 1 goes S -> D and leaves (k1, b2) at element k1 + N1 b2, so
// that stage 2 is in place in D and ends in natural order k = k1 + N1 k2.
template <int N1>
__device__ inline void slab_fft(const real2 *S, real2 *D, int lines, int ls, int es, const real2 *tw, int sign)
{
    // stage 1: DFT-N1 over a for fixed (line, b2), times W_n^(b2 k1)
    for (int item = threadIdx.x; item < lines * 8; item += blockDim.x) {
        const int b2 = item / lines, line = item - b2 * lines;
        const real2 *src = S + line * ls + b2 * es;
        real re[N1], im[N1];
#pragma unroll
        for (int a = 0; a < N1; ++a) { const real2 x = src[8 * a * es]; re[a] = x.x; im[a] = x.y; }
        if (sign < 0) dftN<N1>(re, im); else dftN<N1>(im, re);
        real2 *dst = D + line * ls + N1 * b2 * es;
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) {
            const real2 t = tw[b2 * k1];
            const real ti = sign < 0 ? -t.y : t.y;
            dst[k1 * es] = make_real2(re[k1] * t.x - im[k1] * ti, re[k1] * ti + im[k1] * t.x);
        }
    }
    __syncthreads();
    // stage 2 (in place): DFT-8 over b2 for fixed (line, k1) -> element k1 + N1*k2
    for (int item = threadIdx.x; item < lines * N1; item += blockDim.x) {
        const int k1 = item / lines, line = item - k1 * lines;
        real2 *p = D + line * ls + k1 * es;
        real re[8], im[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) { const real2 x = p[N1 * b * es]; re[b] = x.x; im[b] = x.y; }
        if (sign < 0) rbc::dft8(re, im); else rbc::dft8(im, re);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) p[N1 * k2 * es] = make_real2(re[k2], im[k2]);
    }
    __syncthreads();
}

// padded row length of a slab in LDS (complex values): odd, see slab_fft
__host__ __device__ __forceinline__ int slab_row(int nx) { return nx | 1; }

// 2D transform of the slab in A (rows of slab_row(nx) values; result back in A); the generic routine for other sizes
__device__ inline void slab_fft2d(real2 *A, real2 *T, int nx, int ny, const FftPlan &pl, const real2 *twx, const real2 *twy,
                                  int sign)
{
    const int nxp = slab_row(nx);
    real2 *S = A, *D = T;
    // along x: lines = rows (stride nxp, element stride 1)
    if (pl.nx2 == 8 && pl.nx1 == 6) { slab_fft<6>(S, D, ny, nxp, 1, twx, sign); real2 *t = S; S = D; D = t; }
    else if (pl.nx2 == 8 && pl.nx1 == 4) { slab_fft<4>(S, D, ny, nxp, 1, twx, sign); real2 *t = S; S = D; D = t; }
    else if (pl.nx2 == 8 && pl.nx1 == 8) { slab_fft<8>(S, D, ny, nxp, 1, twx, sign); real2 *t = S; S = D; D = t; }
    else slab_dft(S, S, ny, nx, pl.nx1, pl.nx2, nxp, 1, twx, sign, D);      // (the long rows of streaming-2D grids: slab_fft_rows below)
    // along y: lines = columns (stride 1, element stride nxp)
    if (pl.ny2 == 8 && pl.ny1 == 6) { slab_fft<6>(S, D, nx, 1, nxp, twy, sign); real2 *t = S; S = D; D = t; }
    else if (pl.ny2 == 8 && pl.ny1 == 4) { slab_fft<4>(S, D, nx, 1, nxp, twy, sign); real2 *t = S; S = D; D = t; }
    else if (pl.ny2 == 8 && pl.ny1 == 8) { slab_fft<8>(S, D, nx, 1, nxp, twy, sign); real2 *t = S; S = D; D = t; }
    else slab_dft(S, S, nx, ny, pl.ny1, pl.ny2, 1, nxp, twy, sign, D);
    if (S != A) {
        for (int idx = threadIdx.x; idx < nxp * ny; idx += blockDim.x) A[idx] = S[idx];
        __syncthreads();
    }
}


// forward: rhs slab (divergence of U*/dts) -> 2D spectrum.  One workgroup per (env, k).
__global__ void k3_rhs_fft(Geo3 g, FftPlan pl, const real *st, real2 *spec, real dts)
{
    extern __shared__ __attribute__((aligned(16))) real2 sm[];
    const int nx = g.nx, ny = g.ny, nz = g.nz, pln = nx * ny;
    const int env = blockIdx.x / nz, k = blockIdx.x - env * nz;
    const int nxp = slab_row(nx), lpl = nxp * ny;                  // LDS slab: rows padded to nxp values
    real2 *A = sm, *T = sm + lpl, *twx = sm + 2 * lpl, *twy = twx + nx;
    for (int t = threadIdx.x; t < nx + ny; t += blockDim.x) twx[t] = pl.tw[t];      // twy = twx + nx, like the table
    const real *sb = st + (size_t)env * g.env_stride;
    const real *u = sb + g.nc, *v = sb + 2 * (size_t)g.nc, *w = sb + 3 * (size_t)g.nc;
    const real rdt = real(1.0) / dts;
    for (int idx = threadIdx.x; idx < pln; idx += blockDim.x) {
        const int j = idx / nx, i = idx - j * nx;
        const int ip = (i + 1 == nx) ? 0 : i + 1, jp = (j + 1 == ny) ? 0 : j + 1;
        const size_t c = (size_t)k * pln + idx;
        const real wt = (k + 1 < nz) ? w[c + pln] : real(0.0);
        const real wb = (k > 0) ? w[c] : real(0.0);
        const real d = (u[(size_t)k * pln + j * nx + ip] - u[c]) * g.rdx + (v[(size_t)k * pln + jp * nx + i] - v[c]) * g.rdy + (wt - wb) * g.rdz;
        A[j * nxp + i] = make_real2(d * rdt, real(0.0));
    }
    __syncthreads();
    slab_fft2d(A, T, nx, ny, pl, twx, twy, -1);
    real2 *o = spec + ((size_t)env * nz + k) * pln;
    for (int idx = threadIdx.x; idx < pln; idx += blockDim.x) { const int j = idx / nx; o[idx] = A[idx + j * (nxp - nx)]; }
}

// z sweep per (env, n, m): tab[k][n][m] = 1/pivot; mean mode pinned (its z-mean is removed on output)
__global__ void k3_thomas(Geo3 g, real2 *spec, const real *tab, int B)
{
    const int pln = g.nx * g.ny;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pln * B) return;
    const int env = t / pln, mn = t - env * pln;
    real2 *s = spec + (size_t)env * g.nz * pln + mn;
    const real o = g.rdz * g.rdz;
    // the sweeps are latency-bound (about one wave per SIMD): fetch BK levels ahead of the recurrence, the
    // stores of one block overlap the loads of the next
    constexpr int BK = 8;
    real yr = real(0.0), yi = real(0.0);
    int k = 0;
    for (; k + BK <= g.nz; k += BK) {
        real inv[BK]; real2 r[BK];
#pragma unroll
        for (int q = 0; q < BK; ++q) { inv[q] = tab[(size_t)(k + q) * pln + mn]; r[q] = s[(size_t)(k + q) * pln]; }
#pragma unroll
        for (int q = 0; q < BK; ++q) {
            yr = r[q].x * inv[q] - (inv[q] * o) * yr;
            yi = r[q].y * inv[q] - (inv[q] * o) * yi;
            s[(size_t)(k + q) * pln] = make_real2(yr, yi);
        }
    }
    for (; k < g.nz; ++k) {
        const real inv = tab[(size_t)k * pln + mn];
        const real2 r = s[(size_t)k * pln];
        yr = r.x * inv - (inv * o) * yr;
        yi = r.y * inv - (inv * o) * yi;
        s[(size_t)k * pln] = make_real2(yr, yi);
    }
    real xr = real(0.0), xi = real(0.0);
    k = g.nz - 1;
    for (; k - BK + 1 >= 0; k -= BK) {
        real cp[BK]; real2 y[BK];
#pragma unroll
        for (int q = 0; q < BK; ++q) { cp[q] = tab[(size_t)(k - q) * pln + mn] * o; y[q] = s[(size_t)(k - q) * pln]; }
#pragma unroll
        for (int q = 0; q < BK; ++q) {
            xr = y[q].x - cp[q] * xr; xi = y[q].y - cp[q] * xi;
            s[(size_t)(k - q) * pln] = make_real2(xr, xi);
        }
    }
    for (; k >= 0; --k) {
        const real cp = tab[(size_t)k * pln + mn] * o;
        const real2 y = s[(size_t)k * pln];
        xr = y.x - cp * xr; xi = y.y - cp * xi;
        s[(size_t)k * pln] = make_real2(xr, xi);
    }
}

// inverse 2D FFT of one slab -> phi[env][k][j][i] (real part, normalised)
__global__ void k3_ifft(Geo3 g, FftPlan pl, const real2 *spec, real *phi)
{
    extern __shared__ __attribute__((aligned(16))) real2 sm[];
    const int nx = g.nx, ny = g.ny, nz = g.nz, pln = nx * ny;
    const int env = blockIdx.x / nz, k = blockIdx.x - env * nz;
    const int nxp = slab_row(nx), lpl = nxp * ny;                  // LDS slab: rows padded to nxp values
    real2 *A = sm, *T = sm + lpl, *twx = sm + 2 * lpl, *twy = twx + nx;
    for (int t = threadIdx.x; t < nx + ny; t += blockDim.x) twx[t] = pl.tw[t];      // twy = twx + nx, like the table
    const real2 *in = spec + ((size_t)env * nz + k) * pln;
    for (int idx = threadIdx.x; idx < pln; idx += blockDim.x) { const int j = idx / nx; A[idx + j * (nxp - nx)] = in[idx]; }
    __syncthreads();
    slab_fft2d(A, T, nx, ny, pl, twx, twy, +1);
    const real sc = real(1.0) / (real)pln;
    real *o = phi + ((size_t)env * nz + k) * pln;
    for (int idx = threadIdx.x; idx < pln; idx += blockDim.x) { const int j = idx / nx; o[idx] = A[idx + j * (nxp - nx)].x * sc; }
}


// ---- mirror-slab packing (even nz) ----------------------------------------------------------------------
// Slab k (real part) and its mirror image nz-1-k (imaginary part) go through ONE complex 2D FFT, and the packed
// spectrum is never unpacked: the z operator is real and mirror symmetric, so eliminating from both walls applies,
// at step k, the same real recurrence to slab k (upward) and slab nz-1-k (downward), i.e. to the real and the
// imaginary part of the packed value alike ("Poisson", has the argument).  Only the 2x2
// junction between slabs nz/2-1 and nz/2 mixes the two sweeps, through the packed value of the conjugate mode
// (-kx, -ky):  X = jf (P - i c conj(P')).  Half the FFTs, half the spectrum, sweeps of half the length.
__global__ void k3_rhs_fft_pair(Geo3 g, FftPlan pl, const real *st, real2 *spec, real dts)
{
    extern __shared__ __attribute__((aligned(16))) real2 sm[];
    const int nx = g.nx, ny = g.ny, nz = g.nz, pln = nx * ny, half = nz / 2;
    const int env = blockIdx.x / half, k = blockIdx.x - env * half, km = nz - 1 - k;
    const int nxp = slab_row(nx), lpl = nxp * ny;                  // LDS slab: rows padded to nxp values
    real2 *A = sm, *T = sm + lpl, *twx = sm + 2 * lpl, *twy = twx + nx;
    for (int t = threadIdx.x; t < nx + ny; t += blockDim.x) twx[t] = pl.tw[t];      // twy = twx + nx, like the table
    const real *sb = st + (size_t)env * g.env_stride;
    const real *u = sb + g.nc, *v = sb + 2 * (size_t)g.nc, *w = sb + 3 * (size_t)g.nc;
    const real rdt = real(1.0) / dts;
    for (int idx = threadIdx.x; idx < pln; idx += blockDim.x) {
        const int j = idx / nx, i = idx - j * nx;
        const int ip = (i + 1 == nx) ? 0 : i + 1, jp = (j + 1 == ny) ? 0 : j + 1;
        const int e = j * nx + ip, n = jp * nx + i;
        auto div = [&](int kk) -> real {
            const size_t c = (size_t)kk * pln;
            const real wt = (kk + 1 < nz) ? w[c + pln + idx] : real(0.0);
            const real wb = (kk > 0) ? w[c + idx] : real(0.0);
            return (u[c + e] - u[c + idx]) * g.rdx + (v[c + n] - v[c + idx]) * g.rdy + (wt - wb) * g.rdz;
        };
        A[j * nxp + i] = make_real2(div(k) * rdt, div(km) * rdt);
    }
    __syncthreads();
    slab_fft2d(A, T, nx, ny, pl, twx, twy, -1);
    real2 *o = spec + ((size_t)env * half + k) * pln;
    for (int idx = threadIdx.x; idx < pln; idx += blockDim.x) { const int j = idx / nx; o[idx] = A[idx + j * (nxp - nx)]; }
}

// forward elimination of the packed spectrum over k = 0..nz/2-1; the junction value also goes to jct[env][mode]
__global__ void k3_thomas_pair_fwd(Geo3 g, real2 *spec, real2 *jct, const real *tab, int B)
{
    const int pln = g.nx * g.ny, half = g.nz / 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pln * B) return;
    const int env = t / pln, mn = t - env * pln;
    real2 *s = spec + (size_t)env * half * pln + mn;
    const real o = g.rdz * g.rdz;
    constexpr int BK = 8;
    real yr = real(0.0), yi = real(0.0);
    int k = 0;
    for (; k + BK <= half; k += BK) {
        real inv[BK]; real2 r[BK];
#pragma unroll
        for (int q = 0; q < BK; ++q) { inv[q] = tab[(size_t)(k + q) * pln + mn]; r[q] = s[(size_t)(k + q) * pln]; }
#pragma unroll
        for (int q = 0; q < BK; ++q) {
            yr = r[q].x * inv[q] - (inv[q] * o) * yr;
            yi = r[q].y * inv[q] - (inv[q] * o) * yi;
            s[(size_t)(k + q) * pln] = make_real2(yr, yi);
        }
    }
    for (; k < half; ++k) {
        const real inv = tab[(size_t)k * pln + mn];
        const real2 r = s[(size_t)k * pln];
        yr = r.x * inv - (inv * o) * yr;
        yi = r.y * inv - (inv * o) * yi;
        s[(size_t)k * pln] = make_real2(yr, yi);
    }
    jct[(size_t)env * pln + mn] = make_real2(yr, yi);
}

// junction with the conjugate mode, then back-substitution outward
// partner != nullptr: the spectrum is in another order than the natural one (in-place row FFTs of the streaming-2D mode); tab is
// then permuted likewise and partner[p] is the position of the conjugate mode
__global__ void k3_thomas_pair_bwd(Geo3 g, real2 *spec, const real2 *jct, const real *tab, int B, const int *partner = nullptr)
{
    const int nx = g.nx, ny = g.ny, pln = nx * ny, half = g.nz / 2;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pln * B) return;
    const int env = t / pln, mn = t - env * pln;
    const int n = mn / nx, m = mn - n * nx;
    const int mp = partner ? partner[mn] : ((n == 0) ? 0 : ny - n) * nx + ((m == 0) ? 0 : nx - m);     // mode (-kx, -ky)
    real2 *s = spec + (size_t)env * half * pln + mn;
    const real o = g.rdz * g.rdz;
    const real2 P = jct[(size_t)env * pln + mn], Pc = jct[(size_t)env * pln + mp];
    const real c = tab[(size_t)(half - 1) * pln + mn] * o;
    real xr, xi;
    if (mn == 0) { xr = P.x; xi = real(0.0); }                                        // singular mean mode: pin phi = 0 in slab nz/2
    else { const real jf = real(1.0) / (real(1.0) - c * c); xr = jf * (P.x - c * Pc.y); xi = jf * (P.y - c * Pc.x); }
    s[(size_t)(half - 1) * pln] = make_real2(xr, xi);
    constexpr int BK = 8;
    int k = half - 2;
    for (; k - BK + 1 >= 0; k -= BK) {
        real cp[BK]; real2 y[BK];
#pragma unroll
        for (int q = 0; q < BK; ++q) { cp[q] = tab[(size_t)(k - q) * pln + mn] * o; y[q] = s[(size_t)(k - q) * pln]; }
#pragma unroll
        for (int q = 0; q < BK; ++q) {
            xr = y[q].x - cp[q] * xr; xi = y[q].y - cp[q] * xi;
            s[(size_t)(k - q) * pln] = make_real2(xr, xi);
        }
    }
    for (; k >= 0; --k) {
        const real cp = tab[(size_t)k * pln + mn] * o;
        const real2 y = s[(size_t)k * pln];
        xr = y.x - cp * xr; xi = y.y - cp * xi;
        s[(size_t)k * pln] = make_real2(xr, xi);
    }
}

// Both sweeps in one launch, the packed spectrum read and written ONCE: a thread owns a mode (kx, ky) together with its
// conjugate partner (-kx, -ky) -- the only other mode its junction needs -- and keeps the forward results of both columns
// (2 x HALF complex values) in registers between the sweeps.  Threads of the partner modes exit at once (whole waves, except
// in the two self-conjugate rows).  Same recurrences in the same order as k3_thomas_pair_fwd / _bwd: bitwise the same result.
template <int HALF>
__global__ void __launch_bounds__(128) k3_thomas_pair_fused(Geo3 g, real2 *spec, const real *tab, int B, const int *partner = nullptr)
{
    const int nx = g.nx, ny = g.ny, pln = nx * ny;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= pln * B) return;
    const int env = t / pln, mn = t - env * pln;
    const int n = mn / nx, m = mn - n * nx;
    const int mp = partner ? partner[mn] : ((n == 0) ? 0 : ny - n) * nx + ((m == 0) ? 0 : nx - m);     // mode (-kx, -ky)
    if (mp < mn) return;                                                       // the partner's thread does this pair
    const bool self = (mp == mn);
    real2 *sa = spec + (size_t)env * HALF * pln + mn, *sb = spec + (size_t)env * HALF * pln + mp;
    const real o = g.rdz * g.rdz;
    real2 ya[HALF], yb[HALF];
    real inv[HALF];
#pragma unroll
    for (int k = 0; k < HALF; ++k) { inv[k] = tab[(size_t)k * pln + mn]; ya[k] = sa[(size_t)k * pln]; yb[k] = sb[(size_t)k * pln]; }
    real ar = real(0.0), ai = real(0.0), br = real(0.0), bi = real(0.0);
#pragma unroll
    for (int k = 0; k < HALF; ++k) {                                           // forward elimination (pivots depend on |kx|, |ky| only)
        ar = ya[k].x * inv[k] - (inv[k] * o) * ar; ai = ya[k].y * inv[k] - (inv[k] * o) * ai;
        br = yb[k].x * inv[k] - (inv[k] * o) * br; bi = yb[k].y * inv[k] - (inv[k] * o) * bi;
        ya[k] = make_real2(ar, ai); yb[k] = make_real2(br, bi);
    }
    const real c = inv[HALF - 1] * o;
    real xar, xai, xbr, xbi;
    if (mn == 0) { xar = ar; xai = real(0.0); xbr = ar; xbi = real(0.0); }                  // singular mean mode: pin phi = 0 in slab nz/2
    else {
        const real jf = real(1.0) / (real(1.0) - c * c);
        xar = jf * (ar - c * bi); xai = jf * (ai - c * br);                    // X = jf (P - i c conj(P'))
        xbr = jf * (br - c * ai); xbi = jf * (bi - c * ar);
    }
    sa[(size_t)(HALF - 1) * pln] = make_real2(xar, xai);
    if (!self) sb[(size_t)(HALF - 1) * pln] = make_real2(xbr, xbi);
#pragma unroll
    for (int k = HALF - 2; k >= 0; --k) {                                      // back-substitution outward
        const real cp = inv[k] * o;
        xar = ya[k].x - cp * xar; xai = ya[k].y - cp * xai;
        xbr = yb[k].x - cp * xbr; xbi = yb[k].y - cp * xbi;
        sa[(size_t)k * pln] = make_real2(xar, xai);
        if (!self) sb[(size_t)k * pln] = make_real2(xbr, xbi);
    }
}

// inverse 2D FFT of a packed slab pair -> phi of slab k (real part) and of slab nz-1-k (imaginary part), and, with both
// potentials of the slab still in LDS, the horizontal half of pressure_correct_velocities!: u -= dts dphi/dx, v -= dts dphi/dy
// on the two slabs (st != nullptr).  The vertical half needs phi of the slab below: k3_correct_w.
__global__ void k3_ifft_pair(Geo3 g, FftPlan pl, const real2 *spec, real *phi, real *st, real dts, const uint8_t *mask)
{
    extern __shared__ __attribute__((aligned(16))) real2 sm[];
    const int nx = g.nx, ny = g.ny, nz = g.nz, pln = nx * ny, half = nz / 2;
    const int env = blockIdx.x / half, k = blockIdx.x - env * half;
    if (mask && !mask[env]) return;                               // masked reset: this env is not being projected
    const int nxp = slab_row(nx), lpl = nxp * ny;                  // LDS slab: rows padded to nxp values
    real2 *A = sm, *T = sm + lpl, *twx = sm + 2 * lpl, *twy = twx + nx;
    for (int t = threadIdx.x; t < nx + ny; t += blockDim.x) twx[t] = pl.tw[t];      // twy = twx + nx, like the table
    const real2 *in = spec + ((size_t)env * half + k) * pln;
    for (int idx = threadIdx.x; idx < pln; idx += blockDim.x) { const int j = idx / nx; A[idx + j * (nxp - nx)] = in[idx]; }
    __syncthreads();
    slab_fft2d(A, T, nx, ny, pl, twx, twy, +1);
    const real sc = real(1.0) / (real)pln;
    real *lo = phi + ((size_t)env * nz + k) * pln, *hi = phi + ((size_t)env * nz + (nz - 1 - k)) * pln;
    for (int idx = threadIdx.x; idx < pln; idx += blockDim.x) { const int j = idx / nx; const real2 a = A[idx + j * (nxp - nx)]; lo[idx] = a.x * sc; hi[idx] = a.y * sc; }
    if (!st) return;
    real *sb = st + (size_t)env * g.env_stride;
    real *ulo = sb + g.nc + (size_t)k * pln, *uhi = sb + g.nc + (size_t)(nz - 1 - k) * pln;
    real *vlo = sb + 2 * (size_t)g.nc + (size_t)k * pln, *vhi = sb + 2 * (size_t)g.nc + (size_t)(nz - 1 - k) * pln;
    // all global loads of a batch first (the workgroup has only four waves to hide their latency), then the updates
    constexpr int UB = 4;
    for (int base = threadIdx.x; base < pln; base += UB * blockDim.x) {
        real ul[UB], uh[UB], vl[UB], vh[UB];
#pragma unroll
        for (int q = 0; q < UB; ++q) {
            const int idx = base + q * blockDim.x;
            if (idx < pln) { ul[q] = ulo[idx]; uh[q] = uhi[idx]; vl[q] = vlo[idx]; vh[q] = vhi[idx]; }
        }
#pragma unroll
        for (int q = 0; q < UB; ++q) {
            const int idx = base + q * blockDim.x;
            if (idx < pln) {
                const int j = idx / nx, i = idx - j * nx;
                const int w_ = j * nxp + ((i == 0) ? nx - 1 : i - 1), s_ = ((j == 0) ? ny - 1 : j - 1) * nxp + i;
                const real2 c = A[j * nxp + i], pw = A[w_], ps = A[s_];
                // same operation order as k3_correct: (phi_c - phi_w) * rdx * dts on the normalised potentials
                ulo[idx] = ul[q] - (c.x * sc - pw.x * sc) * g.rdx * dts; uhi[idx] = uh[q] - (c.y * sc - pw.y * sc) * g.rdx * dts;
                vlo[idx] = vl[q] - (c.x * sc - ps.x * sc) * g.rdy * dts; vhi[idx] = vh[q] - (c.y * sc - ps.y * sc) * g.rdy * dts;
            }
        }
    }
}

// The whole correction in the inverse-FFT kernel (round 4): a workgroup MARCHES over CH adjacent packed slab pairs p = k0 .. k0+CH-1
// of one env.  The vertical correction of face f needs phi of the cells below and above it, at the SAME column -- and with mirror
// packing both neighbours of a face live in ADJACENT pairs: face p (lower half) needs Re of pairs p-1 and p, face nz-p (upper half)
// needs Im of pairs p and p-1, and the junction face nz/2 is Im - Re of pair nz/2-1 alone.  So every thread keeps the (normalised)
// packed potential of the previous pair at the columns it owns in registers (pln / blockDim complex values) and applies
//     w(p)    -= (Re phi_p - Re phi_{p-1}) rdz dts,      w(nz-p) -= (Im phi_{p-1} - Im phi_p) rdz dts
// right after the horizontal half, with the same expression and operand order as k3_correct_w: bit for bit the same state.  A
// chunk that does not start at the wall transforms pair k0-1 once more to seed its registers ((CH+1)/CH of the FFT work); the
// k3_correct_w launch, its read of w and two reads of phi, and the store of phi (never an output of a 3D env) are gone.
// Timing bound measured before this kernel existed (a build that simply skips k3_correct_w, wrong numerics): +8.3 % float64,
// +6.3 % float32 at configs[4]; realised with CH = 2: +6.6 % float64 (rbc3d_host_body.hpp, create3d, has the sweep over CH).
template <int CH, int NP>                                        // NP: columns a thread owns, pln <= NP * blockDim (chosen by the host)
__global__ void __launch_bounds__(512, 2) k3_ifft_march(Geo3 g, FftPlan pl, const real2 *spec, real *st, real dts, const uint8_t *mask)
{
    // (two waves per SIMD, up to 256 VGPRs: a launch has B * nz / 4 workgroups -- 64 per env group at configs[4] -- so residency is no
    //  constraint, and the registers buy the prefetch of EVERYTHING a pair's correction reads before its transform starts)
    extern __shared__ __attribute__((aligned(16))) real2 sm[];
    const int nx = g.nx, ny = g.ny, nz = g.nz, pln = nx * ny, half = nz / 2, nchunk = half / CH;
    const int env = blockIdx.x / nchunk, k0 = (blockIdx.x - env * nchunk) * CH;
    if (mask && !mask[env]) return;
    const int nxp = slab_row(nx), lpl = nxp * ny;
    real2 *A = sm, *T = sm + lpl, *twx = sm + 2 * lpl, *twy = twx + nx;
    for (int t = threadIdx.x; t < nx + ny; t += blockDim.x) twx[t] = pl.tw[t];
    const real sc = real(1.0) / (real)pln;
    real *sb = st + (size_t)env * g.env_stride;
    real *ub = sb + g.nc, *vb = sb + 2 * (size_t)g.nc, *wb = sb + 3 * (size_t)g.nc;
    const int s0 = (k0 > 0) ? -1 : 0;
    real2 prev[NP], nxt[NP];                                       // potential of the previous pair / spectrum of the next one, own columns
    {
        const real2 *in = spec + ((size_t)env * half + (k0 + s0)) * pln;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int idx = threadIdx.x + q * blockDim.x;
            prev[q] = make_real2(real(0.0), real(0.0));
            nxt[q] = (idx < pln) ? in[idx] : make_real2(real(0.0), real(0.0));
        }
    }
    for (int s = s0; s < CH; ++s) {
        const int p = k0 + s, pm = nz - 1 - p;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int idx = threadIdx.x + q * blockDim.x;
            if (idx < pln) { const int j = idx / nx; A[idx + j * (nxp - nx)] = nxt[q]; }
        }
        if (s + 1 < CH) {                                         // the next pair's spectrum flies under this pair's transform
            const real2 *in = spec + ((size_t)env * half + (p + 1)) * pln;
#pragma unroll
            for (int q = 0; q < NP; ++q) { const int idx = threadIdx.x + q * blockDim.x; if (idx < pln) nxt[q] = in[idx]; }
        }
        real *ulo = ub + (size_t)p * pln, *uhi = ub + (size_t)pm * pln, *vlo = vb + (size_t)p * pln, *vhi = vb + (size_t)pm * pln;
        real *wlo = wb + (size_t)p * pln, *whi = wb + (size_t)(nz - p) * pln, *wj = wb + (size_t)half * pln;      // faces p, nz - p, nz/2
        const bool low = (p > 0), jct = (p == half - 1);          // the wall faces 0 and nz carry no correction
        // ... and so does everything this pair's correction will read: u, v of both slabs, w of its two (three) faces
        constexpr int PF = NP;      // (float64 with 8 columns per thread -- 64 x 64 planes -- does not fit 256 registers this way: the host keeps the separate pass there)
        real ul[NP], uh[NP], vl[NP], vh[NP], wl[NP], wh[NP], wc[NP];
        if (s >= 0) {
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int idx = threadIdx.x + q * blockDim.x;
                if (idx < pln) {
                    ul[q] = ulo[idx]; uh[q] = uhi[idx]; vl[q] = vlo[idx]; vh[q] = vhi[idx];
                    if (low) { wl[q] = wlo[idx]; wh[q] = whi[idx]; }
                    if (jct) wc[q] = wj[idx];
                }
            }
        }
        __syncthreads();
        slab_fft2d(A, T, nx, ny, pl, twx, twy, +1);
        if (s >= 0) {
#pragma unroll
            for (int q = PF; q < NP; ++q) {
                const int idx = threadIdx.x + q * blockDim.x;
                if (idx < pln) {
                    ul[q] = ulo[idx]; uh[q] = uhi[idx]; vl[q] = vlo[idx]; vh[q] = vhi[idx];
                    if (low) { wl[q] = wlo[idx]; wh[q] = whi[idx]; }
                    if (jct) wc[q] = wj[idx];
                }
            }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int idx = threadIdx.x + q * blockDim.x;
                if (idx < pln) {
                    const int j = idx / nx, i = idx - j * nx;
                    const int w_ = j * nxp + ((i == 0) ? nx - 1 : i - 1), s_ = ((j == 0) ? ny - 1 : j - 1) * nxp + i;
                    const real2 c = A[j * nxp + i], pw = A[w_], ps = A[s_];
                    // same operation order as k3_correct: (phi_c - phi_w) * rdx * dts on the normalised potentials
                    ulo[idx] = ul[q] - (c.x * sc - pw.x * sc) * g.rdx * dts; uhi[idx] = uh[q] - (c.y * sc - pw.y * sc) * g.rdx * dts;
                    vlo[idx] = vl[q] - (c.x * sc - ps.x * sc) * g.rdy * dts; vhi[idx] = vh[q] - (c.y * sc - ps.y * sc) * g.rdy * dts;
                    const real2 cn = make_real2(c.x * sc, c.y * sc), pv = prev[q];
                    if (low) { wlo[idx] = wl[q] - (cn.x - pv.x) * g.rdz * dts; whi[idx] = wh[q] - (pv.y - cn.y) * g.rdz * dts; }
                    if (jct) wj[idx] = wc[q] - (cn.y - cn.x) * g.rdz * dts;
                    prev[q] = cn;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int idx = threadIdx.x + q * blockDim.x;
                if (idx < pln) { const int j = idx / nx; const real2 c = A[idx + j * (nxp - nx)]; prev[q] = make_real2(c.x * sc, c.y * sc); }
            }
        }
        __syncthreads();                                          // A is overwritten by the next pair's spectrum
    }
}

// vertical half of pressure_correct_velocities!: w -= dts dphi/dz on the interior faces (thread per cell, k >= 1)
__global__ void k3_correct_w(Geo3 g, real *st, const real *phi, real dts, int B, const uint8_t *mask)
{
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    const int pln = g.nx * g.ny, per = g.nc - pln;
    if (cell >= per * B) return;
    const int env = cell / per, c0 = cell - env * per + pln;
    if (mask && !mask[env]) return;
    const real *p = phi + (size_t)env * g.nc;
    st[(size_t)env * g.env_stride + 3 * (size_t)g.nc + c0] -= (p[c0] - p[c0 - pln]) * g.rdz * dts;
}

// pressure_correct_velocities!
__global__ void k3_correct(Geo3 g, real *st, const real *phi, real dts, int B, const uint8_t *mask)
{
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= g.nc * B) return;
    const int env = cell / g.nc, c0 = cell - env * g.nc;
    if (mask && !mask[env]) return;
    const int nx = g.nx, ny = g.ny, pln = nx * ny;
    const int k = c0 / pln, j = (c0 - k * pln) / nx, i = c0 - k * pln - j * nx;
    const int im = (i == 0) ? nx - 1 : i - 1, jm = (j == 0) ? ny - 1 : j - 1;
    real *sb = st + (size_t)env * g.env_stride;
    const real *p = phi + (size_t)env * g.nc;
    const real pc = p[c0];
    sb[g.nc + c0] -= (pc - p[(size_t)k * pln + j * nx + im]) * g.rdx * dts;
    sb[2 * (size_t)g.nc + c0] -= (pc - p[(size_t)k * pln + jm * nx + i]) * g.rdy * dts;
    if (k > 0) sb[3 * (size_t)g.nc + c0] -= (pc - p[c0 - pln]) * g.rdz * dts;
}

// copy b (unchanged by a projection-only pass) between state buffers
__global__ void k3_copy(real *dst, const real *src, size_t n)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[t] = src[t];
}

// random IC: initialize_model, rbc_sim3D.jl:169-179 (fields 0:u 1:v 2:w 3:b of the counter RNG)
__global__ void k3_random(Geo3 g, real *st, const uint64_t *seeds, const uint8_t *mask, int B)
{
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= g.nw * B) return;
    const int env = cell / g.nw, c0 = cell - env * g.nw;
    if (mask && !mask[env]) return;
    const int pln = g.nx * g.ny, k = c0 / pln;
    real *sb = st + (size_t)env * g.env_stride;
    const uint64_t seed = seeds[env];
    sb[3 * (size_t)g.nc + c0] = (k == 0 || k == g.nz) ? real(0.0) : g.kick * rbc::normal_deviate(seed, 2, (uint32_t)c0);
    if (k < g.nz) {
        sb[g.nc + c0] = g.kick * rbc::normal_deviate(seed, 0, (uint32_t)c0);
        sb[2 * (size_t)g.nc + c0] = g.kick * rbc::normal_deviate(seed, 1, (uint32_t)c0);
        const real z = (k + real(0.5)) * g.dz;
        const real val = g.min_b + (g.lz - z) * g.delta_b / 2 + g.kick * rbc::normal_deviate(seed, 3, (uint32_t)c0);
        sb[c0] = fmin(fmax(val, g.min_b), g.min_b + g.delta_b);
    }
}

// outputs: float32 state (b,u,v,w), Nusselt (rbc_sim3D_api.jl:134-159), NaN flag.  OUT_SPLIT workgroups per env, each over a
// contiguous share of the cells; the partial sums meet in `part` and the workgroup that arrives last (a counter per env) adds
// them in index order, so the result does not depend on the arrival order (bitwise reproducible).
constexpr int OUT_SPLIT = 16;

// =====================================================================================================================
// Streaming-2D mode: 2D grids whose state does not fit a CU's LDS (float64: 128x64, 192x32, ... any nx, nz the slab FFT
// and the z kernels take) run on the 3D streaming kernels above with ny = 1: every y-neighbour of a cell is the cell itself,
// so all y-fluxes, the y-viscous terms and the y-pressure gradient vanish identically, v stays exactly 0 and the (x, z)
// operators are the 2D ones (tests/test_oracle3d.py pins a y-independent 3D state on the 2D oracle; here ny = 1 makes it
// exact and removes the 3D instability of 2D rolls).  What differs from a 3D env is supplied here: the heater profile
// (collate_actions_colin), the 2D random initial condition, and the 2D outputs (5-channel float32 state / observations,
// the two Nusselt numbers of rbc_sim2D_api.jl:142-163).
// =====================================================================================================================

// x transform of `lines` rows held in LDS (rows of slab_row(nx) values), result pointer returned (S or D)
__device__ inline real2 *slab_fft_rows(real2 *S, real2 *D, int lines, int nx, const FftPlan &pl, const real2 *twx, int sign)
{
    const int nxp = slab_row(nx);
#define RBC_ROWS(N1_) if (pl.nx2 == 8 && pl.nx1 == N1_) { slab_fft<N1_>(S, D, lines, nxp, 1, twx, sign); return D; }
    RBC_ROWS(4) RBC_ROWS(6) RBC_ROWS(8) RBC_ROWS(12) RBC_ROWS(16)      // (a DFT-24 in registers costs the kernel its occupancy)
#undef RBC_ROWS
    slab_dft(S, S, lines, nx, pl.nx1, pl.nx2, nxp, 1, twx, sign, D);
    return S;
}
#if RBC_EXPERIMENT_NOFFT
#define slab_fft_rows(S_, D_, l_, n_, p_, t_, s_) (S_)          /* timing experiment only (WRONG numerics) */
#endif

// Projection of a streaming-2D state (ny = 1): a "slab" is one row, so a workgroup takes R consecutive rows k0 .. k0+R-1 of one
// env, each packed with its mirror row nz-1-k as in k3_rhs_fft_pair, and transforms them together (R lines of the row FFT).
// grid = B * (nz/2 / R) workgroups; same arithmetic per value as the 3D kernels (divergence, scaling, correction order).
__global__ void __launch_bounds__(256, 3) k2s_rhs_fft_pair(Geo3 g, FftPlan pl, const real *st, real2 *spec, real dts, int R)
{
    extern __shared__ __attribute__((aligned(16))) real2 sm[];
    const int nx = g.nx, nz = g.nz, half = nz / 2, nxp = slab_row(nx), per = half / R;
    const int env = blockIdx.x / per, k0 = (blockIdx.x - env * per) * R;
    real2 *A = sm, *T = sm + R * nxp, *twx = sm + 2 * R * nxp;
    for (int t = threadIdx.x; t < nx; t += blockDim.x) twx[t] = pl.tw[t];
    const real *sb = st + (size_t)env * g.env_stride;
    const real *u = sb + g.nc, *w = sb + 3 * (size_t)g.nc;
    const real rdt = real(1.0) / dts;
    for (int idx = threadIdx.x; idx < R * nx; idx += blockDim.x) {
        const int r = idx / nx, i = idx - r * nx, k = k0 + r, km = nz - 1 - k;
        const int ip = (i + 1 == nx) ? 0 : i + 1;
        auto div = [&](int kk) -> real {
            const size_t c = (size_t)kk * nx;
            const real wt = (kk + 1 < nz) ? w[c + nx + i] : real(0.0);
            const real wb = (kk > 0) ? w[c + i] : real(0.0);
            return (u[c + ip] - u[c + i]) * g.rdx + (wt - wb) * g.rdz;
        };
        A[r * nxp + i] = make_real2(div(k) * rdt, div(km) * rdt);
    }
    __syncthreads();
    const real2 *S = slab_fft_rows(A, T, R, nx, pl, twx, -1);
    real2 *o = spec + ((size_t)env * half + k0) * nx;
    for (int idx = threadIdx.x; idx < R * nx; idx += blockDim.x) { const int r = idx / nx, i = idx - r * nx; o[idx] = S[r * nxp + i]; }
}

__global__ void __launch_bounds__(256, 3) k2s_ifft_pair(Geo3 g, FftPlan pl, const real2 *spec, real *phi, real *st, real dts, const uint8_t *mask, int R)
{
    extern __shared__ __attribute__((aligned(16))) real2 sm[];
    const int nx = g.nx, nz = g.nz, half = nz / 2, nxp = slab_row(nx), per = half / R;
    const int env = blockIdx.x / per, k0 = (blockIdx.x - env * per) * R;
    if (mask && !mask[env]) return;                               // masked reset: this env is not being projected
    real2 *A = sm, *T = sm + R * nxp, *twx = sm + 2 * R * nxp;
    for (int t = threadIdx.x; t < nx; t += blockDim.x) twx[t] = pl.tw[t];
    const real2 *in = spec + ((size_t)env * half + k0) * nx;
    for (int idx = threadIdx.x; idx < R * nx; idx += blockDim.x) { const int r = idx / nx, i = idx - r * nx; A[r * nxp + i] = in[idx]; }
    __syncthreads();
    const real2 *S = slab_fft_rows(A, T, R, nx, pl, twx, +1);
    const real sc = real(1.0) / (real)nx;
    real *sb = st + (size_t)env * g.env_stride;
    real *ph = phi + (size_t)env * g.nc, *u = sb + g.nc;
    for (int idx = threadIdx.x; idx < R * nx; idx += blockDim.x) {
        const int r = idx / nx, i = idx - r * nx, k = k0 + r, km = nz - 1 - k;
        const real2 c = S[r * nxp + i], pw = S[r * nxp + ((i == 0) ? nx - 1 : i - 1)];
        ph[(size_t)k * nx + i] = c.x * sc; ph[(size_t)km * nx + i] = c.y * sc;
        // same operation order as k3_correct / k3_ifft_pair: (phi_c - phi_w) * rdx * dts on the normalised potentials
        u[(size_t)k * nx + i] -= (c.x * sc - pw.x * sc) * g.rdx * dts;
        u[(size_t)km * nx + i] -= (c.y * sc - pw.y * sc) * g.rdx * dts;
    }
}

// ---- whole projection of a streaming-2D env in ONE kernel ------------------------------------------------------------------
// Where the packed spectrum of an env (nz/2 rows of nx complex values) fits the LDS, one workgroup per env does what the five
// launches above do without the spectrum or the potential ever travelling through memory: divergence of U*/dts of row k and of
// its mirror row nz-1-k packed as one complex row -> row FFTs -> the z solve from both walls on the packed rows (the recurrences
// of k3_thomas_pair_fwd / _bwd, one thread per mode, on LDS) -> inverse row FFTs -> u -= dts d(phi)/dx, w -= dts d(phi)/dz
// (-> phi itself for the pNHS output when asked).  Reads u, w once and writes them once.
// Row FFT, n = 8 N1, fully in place: forward = DFT-N1 over a at stride 8 (x index 8a + b2) with the twiddle W^(b2 k1), then
// DFT-8 over the 8 contiguous values, which leaves mode m = k1 + N1 k2 at position 8 k1 + k2; the inverse undoes the two
// steps in reverse order and ends in natural x order.  Nothing in spectral space needs the natural order: the pivots are looked
// up by mode, the conjugate partner by its position.  Lanes run along rows (odd row stride: no bank conflicts).
template <int N1>
__device__ __forceinline__ void rowfft_inplace(real2 *A, int lines, int ls, const real2 *tw, int sign)
{
    if (sign < 0) {
        for (int item = threadIdx.x; item < lines * 8; item += blockDim.x) {
            const int b2 = item / lines, line = item - b2 * lines;
            real2 *p = A + line * ls + b2;
            real re[N1], im[N1];
#pragma unroll
            for (int a = 0; a < N1; ++a) { const real2 x = p[8 * a]; re[a] = x.x; im[a] = x.y; }
            dftN<N1>(re, im);
#pragma unroll
            for (int k1 = 0; k1 < N1; ++k1) {
                const real2 t = tw[b2 * k1];
                p[8 * k1] = make_real2(re[k1] * t.x + im[k1] * t.y, im[k1] * t.x - re[k1] * t.y);      // times conj(t)
            }
        }
        __syncthreads();
        for (int item = threadIdx.x; item < lines * N1; item += blockDim.x) {
            const int k1 = item / lines, line = item - k1 * lines;
            real2 *p = A + line * ls + 8 * k1;
            real re[8], im[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) { const real2 x = p[b]; re[b] = x.x; im[b] = x.y; }
            rbc::dft8(re, im);
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) p[k2] = make_real2(re[k2], im[k2]);
        }
        __syncthreads();
    } else {
        for (int item = threadIdx.x; item < lines * N1; item += blockDim.x) {
            const int k1 = item / lines, line = item - k1 * lines;
            real2 *p = A + line * ls + 8 * k1;
            real re[8], im[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) { const real2 x = p[k2]; re[k2] = x.x; im[k2] = x.y; }
            rbc::dft8(im, re);
#pragma unroll
            for (int b2 = 0; b2 < 8; ++b2) {
                const real2 t = tw[b2 * k1];
                p[b2] = make_real2(re[b2] * t.x - im[b2] * t.y, re[b2] * t.y + im[b2] * t.x);          // times t
            }
        }
        __syncthreads();
        for (int item = threadIdx.x; item < lines * 8; item += blockDim.x) {
            const int b2 = item / lines, line = item - b2 * lines;
            real2 *p = A + line * ls + b2;
            real re[N1], im[N1];
#pragma unroll
            for (int k1 = 0; k1 < N1; ++k1) { const real2 x = p[8 * k1]; re[k1] = x.x; im[k1] = x.y; }
            dftN<N1>(im, re);
#pragma unroll
            for (int a = 0; a < N1; ++a) p[8 * a] = make_real2(re[a], im[a]);
        }
        __syncthreads();
    }
}

// The separate rhs / inverse kernels with the in-place row FFT (grids whose packed spectrum does not fit one workgroup's LDS:
// 192x128, 256x128, ...): R row pairs per workgroup, half the LDS of the out-of-place version, spectrum in position order
// (mode k1 + N1 k2 at 8 k1 + k2; the z-sweep kernels get the pivot table permuted alike and a partner table).
template <int N1>
__global__ void __launch_bounds__(256) k2s_rhs_fft_pair_ip(Geo3 g, FftPlan pl, const real *st, real2 *spec, real dts, int R)
{
    extern __shared__ __attribute__((aligned(16))) real2 sm[];
    const int nx = g.nx, nz = g.nz, half = nz / 2, ls = slab_row(nx), per = half / R;
    const int env = blockIdx.x / per, k0 = (blockIdx.x - env * per) * R;
    real2 *A = sm, *twx = sm + R * ls;
    for (int t = threadIdx.x; t < nx; t += blockDim.x) twx[t] = pl.tw[t];
    const real *sb = st + (size_t)env * g.env_stride;
    const real *u = sb + g.nc, *w = sb + 3 * (size_t)g.nc;
    const real rdt = real(1.0) / dts;
    for (int idx = threadIdx.x; idx < R * nx; idx += blockDim.x) {
        const int r = idx / nx, i = idx - r * nx, k = k0 + r, km = nz - 1 - k;
        const int ip = (i + 1 == nx) ? 0 : i + 1;
        auto div = [&](int kk) -> real {
            const size_t c = (size_t)kk * nx;
            const real wt = (kk + 1 < nz) ? w[c + nx + i] : real(0.0);
            const real wb = (kk > 0) ? w[c + i] : real(0.0);
            return (u[c + ip] - u[c + i]) * g.rdx + (wt - wb) * g.rdz;
        };
        A[r * ls + i] = make_real2(div(k) * rdt, div(km) * rdt);
    }
    __syncthreads();
    rowfft_inplace<N1>(A, R, ls, twx, -1);
    real2 *o = spec + ((size_t)env * half + k0) * nx;
    for (int idx = threadIdx.x; idx < R * nx; idx += blockDim.x) { const int r = idx / nx, i = idx - r * nx; o[idx] = A[r * ls + i]; }
}

template <int N1>
__global__ void __launch_bounds__(256) k2s_ifft_pair_ip(Geo3 g, FftPlan pl, const real2 *spec, real *phi, real *st, real dts, const uint8_t *mask, int R)
{
    extern __shared__ __attribute__((aligned(16))) real2 sm[];
    const int nx = g.nx, nz = g.nz, half = nz / 2, ls = slab_row(nx), per = half / R;
    const int env = blockIdx.x / per, k0 = (blockIdx.x - env * per) * R;
    if (mask && !mask[env]) return;
    real2 *A = sm, *twx = sm + R * ls;
    for (int t = threadIdx.x; t < nx; t += blockDim.x) twx[t] = pl.tw[t];
    const real2 *in = spec + ((size_t)env * half + k0) * nx;
    for (int idx = threadIdx.x; idx < R * nx; idx += blockDim.x) { const int r = idx / nx, i = idx - r * nx; A[r * ls + i] = in[idx]; }
    __syncthreads();
    rowfft_inplace<N1>(A, R, ls, twx, +1);
    const real sc = real(1.0) / (real)nx;
    real *sb = st + (size_t)env * g.env_stride;
    real *ph = phi + (size_t)env * g.nc, *u = sb + g.nc;
    for (int idx = threadIdx.x; idx < R * nx; idx += blockDim.x) {
        const int r = idx / nx, i = idx - r * nx, k = k0 + r, km = nz - 1 - k;
        const real2 c = A[r * ls + i], pw = A[r * ls + ((i == 0) ? nx - 1 : i - 1)];
        ph[(size_t)k * nx + i] = c.x * sc; ph[(size_t)km * nx + i] = c.y * sc;
        u[(size_t)k * nx + i] -= (c.x * sc - pw.x * sc) * g.rdx * dts;
        u[(size_t)km * nx + i] -= (c.y * sc - pw.y * sc) * g.rdx * dts;
    }
}

template <int N1>
__global__ void __launch_bounds__(N1 >= 32 ? 256 : 512) k2s_project_fused(Geo3 g, FftPlan pl, real *st, real *phi, const real *tab, real dts,
                                                         const uint8_t *mask, int store_phi)
{
    extern __shared__ __attribute__((aligned(16))) real2 sm[];
    const int nx = g.nx, nz = g.nz, half = nz / 2, ls = slab_row(nx), env = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    if (mask && !mask[env]) return;                               // masked reset: this env is not being projected
    real2 *A = sm, *twx = sm + half * ls;
    for (int t = tid; t < nx; t += nthr) twx[t] = pl.tw[t];
    real *sb = st + (size_t)env * g.env_stride;
    real *u = sb + g.nc, *w = sb + 3 * (size_t)g.nc;
    const real rdt = real(1.0) / dts;
    // The loops over an env's cells below run 8-16 trips per thread with global loads in every trip; trip by trip each load's
    // round trip is exposed (hipcc does not pipeline them: the trip count is a run-time value), and these loops were most of the
    // kernel's 42 us.  Every loop therefore works on UB trips at once: all their loads first, then the arithmetic and the stores.
    constexpr int UB = 4;
    for (int base = tid; base < half * nx; base += nthr * UB) {
        real uE[UB][2], uC[UB][2], wT[UB][2], wB[UB][2];
#pragma unroll
        for (int q = 0; q < UB; ++q) {
            const int idx = base + q * nthr;
            if (idx < half * nx) {
                const int r = idx / nx, i = idx - r * nx, km = nz - 1 - r;
                const int ip = (i + 1 == nx) ? 0 : i + 1;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int kk = h2 ? km : r;
                    const size_t c = (size_t)kk * nx;
                    uE[q][h2] = u[c + ip]; uC[q][h2] = u[c + i];
                    wT[q][h2] = (kk + 1 < nz) ? w[c + nx + i] : real(0.0);
                    wB[q][h2] = (kk > 0) ? w[c + i] : real(0.0);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < UB; ++q) {
            const int idx = base + q * nthr;
            if (idx < half * nx) {
                const int r = idx / nx, i = idx - r * nx;
                const real d0 = (uE[q][0] - uC[q][0]) * g.rdx + (wT[q][0] - wB[q][0]) * g.rdz;
                const real d1 = (uE[q][1] - uC[q][1]) * g.rdx + (wT[q][1] - wB[q][1]) * g.rdz;
                A[r * ls + i] = make_real2(d0 * rdt, d1 * rdt);
            }
        }
    }
    __syncthreads();
    rowfft_inplace<N1>(A, half, ls, twx, -1);
    // z solve, one thread per position p (mode m = k1 + N1 k2 at p = 8 k1 + k2)
    {
        const int p = tid;
        const bool on = p < nx;
        const int m = on ? (p >> 3) + N1 * (p & 7) : 0;
        const int mc = (m == 0) ? 0 : nx - m, pc = 8 * (mc % N1) + mc / N1;          // conjugate partner and its position
        const real o = g.rdz * g.rdz;
        real yr = real(0.0), yi = real(0.0);
        if (on) {
            constexpr int BK = 8;
            int k = 0;
            for (; k + BK <= half; k += BK) {
                real inv[BK];
#pragma unroll
                for (int q = 0; q < BK; ++q) inv[q] = tab[(size_t)(k + q) * nx + m];
#pragma unroll
                for (int q = 0; q < BK; ++q) {
                    const real2 r = A[(k + q) * ls + p];
                    yr = r.x * inv[q] - (inv[q] * o) * yr;
                    yi = r.y * inv[q] - (inv[q] * o) * yi;
                    A[(k + q) * ls + p] = make_real2(yr, yi);
                }
            }
            for (; k < half; ++k) {
                const real inv = tab[(size_t)k * nx + m];
                const real2 r = A[k * ls + p];
                yr = r.x * inv - (inv * o) * yr;
                yi = r.y * inv - (inv * o) * yi;
                A[k * ls + p] = make_real2(yr, yi);
            }
        }
        __syncthreads();
        real xr = real(0.0), xi = real(0.0);
        if (on) {
            const real2 P = A[(half - 1) * ls + p], Pc = A[(half - 1) * ls + pc];
            const real c = tab[(size_t)(half - 1) * nx + m] * o;
            if (m == 0) { xr = P.x; xi = real(0.0); }                                       // singular mean mode: pin phi = 0 in row nz/2
            else { const real jf = real(1.0) / (real(1.0) - c * c); xr = jf * (P.x - c * Pc.y); xi = jf * (P.y - c * Pc.x); }
        }
        __syncthreads();                                                              // every junction value read before it is replaced
        if (on) {
            A[(half - 1) * ls + p] = make_real2(xr, xi);
            constexpr int BK = 8;
            int k = half - 2;
            for (; k - BK + 1 >= 0; k -= BK) {
                real cp[BK];
#pragma unroll
                for (int q = 0; q < BK; ++q) cp[q] = tab[(size_t)(k - q) * nx + m] * o;
#pragma unroll
                for (int q = 0; q < BK; ++q) {
                    const real2 y = A[(k - q) * ls + p];
                    xr = y.x - cp[q] * xr; xi = y.y - cp[q] * xi;
                    A[(k - q) * ls + p] = make_real2(xr, xi);
                }
            }
            for (; k >= 0; --k) {
                const real cp = tab[(size_t)k * nx + m] * o;
                const real2 y = A[k * ls + p];
                xr = y.x - cp * xr; xi = y.y - cp * xi;
                A[k * ls + p] = make_real2(xr, xi);
            }
        }
        __syncthreads();
    }
    rowfft_inplace<N1>(A, half, ls, twx, +1);
    // corrections (same operation order as k3_ifft_pair / k3_correct_w: differences of the normalised potentials)
    const real sc = real(1.0) / (real)nx;
    real *ph = phi + (size_t)env * g.nc;
    for (int base = tid; base < half * nx; base += nthr * UB) {
        real u0[UB], u1[UB];
#pragma unroll
        for (int q = 0; q < UB; ++q) {
            const int idx = base + q * nthr;
            if (idx < half * nx) {
                const int r = idx / nx, i = idx - r * nx, km = nz - 1 - r;
                u0[q] = u[(size_t)r * nx + i]; u1[q] = u[(size_t)km * nx + i];
            }
        }
#pragma unroll
        for (int q = 0; q < UB; ++q) {
            const int idx = base + q * nthr;
            if (idx < half * nx) {
                const int r = idx / nx, i = idx - r * nx, km = nz - 1 - r;
                const real2 c = A[r * ls + i], pw = A[r * ls + ((i == 0) ? nx - 1 : i - 1)];
                u[(size_t)r * nx + i] = u0[q] - (c.x * sc - pw.x * sc) * g.rdx * dts;
                u[(size_t)km * nx + i] = u1[q] - (c.y * sc - pw.y * sc) * g.rdx * dts;
                if (store_phi) { ph[(size_t)r * nx + i] = c.x * sc; ph[(size_t)km * nx + i] = c.y * sc; }
            }
        }
    }
    auto phi_at = [&](int cell, int i) -> real { return (cell < half) ? A[cell * ls + i].x : A[(nz - 1 - cell) * ls + i].y; };
    constexpr int WB = 8;
    for (int base = tid; base < (nz - 1) * nx; base += nthr * WB) {
        real w0[WB];
#pragma unroll
        for (int q = 0; q < WB; ++q) {
            const int idx = base + q * nthr;
            if (idx < (nz - 1) * nx) w0[q] = w[(size_t)nx + idx];                       // face f = idx / nx + 1, column i: offset f * nx + i
        }
#pragma unroll
        for (int q = 0; q < WB; ++q) {
            const int idx = base + q * nthr;
            if (idx < (nz - 1) * nx) {
                const int f = idx / nx + 1, i = idx - (f - 1) * nx;                   // face f between cells f-1 and f
                w[(size_t)nx + idx] = w0[q] - (phi_at(f, i) * sc - phi_at(f - 1, i) * sc) * g.rdz * dts;
            }
        }
    }
}


// random IC of the 2D envs: initialize_model, rbc_sim2D.jl:163-171, with the 2D kernel's counters (fields 0:u 1:w 2:b, index k*nx+i)
__global__ void k2s_random(Geo3 g, real *st, const uint64_t *seeds, const uint8_t *mask, int B)
{
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= g.nc * B) return;
    const int env = cell / g.nc, c0 = cell - env * g.nc;
    if (mask && !mask[env]) return;
    const int k = c0 / g.nx;
    real *sb = st + (size_t)env * g.env_stride;
    const uint64_t seed = seeds[env];
    sb[g.nc + c0] = g.kick * rbc::normal_deviate(seed, 0, (uint32_t)c0);                       // u
    sb[2 * (size_t)g.nc + c0] = real(0.0);                                                           // v
    sb[3 * (size_t)g.nc + c0] = (k == 0) ? real(0.0) : g.kick * rbc::normal_deviate(seed, 1, (uint32_t)c0);   // w (wall face 0)
    if (k == g.nz - 1) sb[3 * (size_t)g.nc + c0 + g.nx] = real(0.0);                                 // top wall face
    const real z = (k + real(0.5)) * g.dz;
    const real val = g.min_b + (g.lz - z) * g.delta_b / 2 + g.kick * rbc::normal_deviate(seed, 2, (uint32_t)c0);
    sb[c0] = fmin(fmax(val, g.min_b), g.min_b + g.delta_b);
}

// v of the envs being reset, in BOTH state buffers: the v tendency kernel is not launched in streaming-2D mode (v == 0), so a
// NaN that a blown-up env left in the other buffer would otherwise survive the reset
__global__ void k2s_clear_v(Geo3 g, real *st0, real *st1, const uint8_t *mask, int B)
{
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= g.nc * B) return;
    const int env = cell / g.nc, c0 = cell - env * g.nc;
    if (mask && !mask[env]) return;
    const size_t o = (size_t)env * g.env_stride + 2 * (size_t)g.nc + c0;
    st0[o] = real(0.0); st1[o] = real(0.0);
}

__global__ void __launch_bounds__(256) k3_output(Geo3 g, const real *st, const f64 *nu_kappa, float *state32, f64 *nusselt, int *flags,
                                                 const uint8_t *mask, f64 *part, unsigned int *arrive, ObsNorm3 norm)
{
    __shared__ f64 red[256];
    __shared__ int bad, last;
    const int env = blockIdx.x / OUT_SPLIT, sp = blockIdx.x - env * OUT_SPLIT;
    if (mask && !mask[env]) return;
    const real *sb = st + (size_t)env * g.env_stride;
    const int pln = g.nx * g.ny;
    if (threadIdx.x == 0) { bad = 0; last = 0; }
    __syncthreads();
    f64 acc = real(0.0);
    int nan = 0;
    float *o = state32 + (size_t)env * 4 * g.nc;
    const int per = (g.nc + OUT_SPLIT - 1) / OUT_SPLIT, c_end = min(g.nc, (sp + 1) * per);
    for (int c0 = sp * per + threadIdx.x; c0 < c_end; c0 += blockDim.x) {
        const int k = c0 / pln;
        const f64 b = sb[c0], u = sb[g.nc + c0], v = sb[2 * (size_t)g.nc + c0], w = sb[3 * (size_t)g.nc + c0];
        o[c0] = obs_value3(norm, 0, (float)b); o[g.nc + c0] = obs_value3(norm, 1, (float)u);
        o[2 * (size_t)g.nc + c0] = obs_value3(norm, 2, (float)v); o[3 * (size_t)g.nc + c0] = obs_value3(norm, 3, (float)w);
        const f64 zc = (k + real(0.5)) / g.nz, tc = (real(1.0) - zc) * g.delta_b + g.min_b;
        acc += (b - tc) * w;
        nan |= (isnan(b) || isnan(u) || isnan(v) || isnan(w)) ? 1 : 0;
    }
    red[threadIdx.x] = acc;
    if (nan) bad = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        f64 s = real(0.0);
        for (int t = 0; t < (int)blockDim.x; ++t) s += red[t];
        part[(size_t)env * 2 * OUT_SPLIT + sp] = s;
        part[(size_t)env * 2 * OUT_SPLIT + OUT_SPLIT + sp] = bad ? real(1.0) : real(0.0);
        __threadfence();
        if (atomicAdd(&arrive[env], 1u) == OUT_SPLIT - 1) {       // every share of this env is in
            __threadfence();
            f64 tot = real(0.0), anybad = real(0.0);
            for (int q = 0; q < OUT_SPLIT; ++q) {
                tot += __builtin_nontemporal_load(&part[(size_t)env * 2 * OUT_SPLIT + q]);
                anybad += __builtin_nontemporal_load(&part[(size_t)env * 2 * OUT_SPLIT + OUT_SPLIT + q]);
            }
            nusselt[env] = real(1.0) + (tot / (f64)g.nc) / nu_kappa[2 * env + 1];
            flags[env] = anybad > real(0.0) ? 1 : 0;
            arrive[env] = 0;                                       // ready for the next call
        }
    }
}

// 2D outputs of one env per workgroup (256 threads; serial sums in index order: deterministic): channels b,u,w,pHY',pNHS as
// float32 state and strided observations (rbc_sim2D_api.jl:102-129), Nusselt numbers on the state and on the sensor grid
// (:142-163 with array_gradient, rbc_sim2D.jl:206-220), NaN flag.  phi = the last stage's potential (its mean removed).
// When the tendencies used the un-split buoyancy (FLAT tile kernels) the projection's potential is phi + (gamma pHY'(stage) +
// zeta pHY'(previous stage)) / (gamma + zeta) (see the note at the tile kernels): phyA / phyB are those two scans of the last
// substep's stages 3 and 2, ca / cb their weights; phyA == nullptr (after a reset, and on the hydrostatic-split fallback kernels):
// phi is pNHS itself.
__global__ void __launch_bounds__(256) k2s_output(Geo3 g, const real *st, const real *phi, const f64 *nu_kappa, Out2D P, const uint8_t *mask,
                                                  const real *phyA, const real *phyB, f64 ca, f64 cb)
{
    extern __shared__ f64 sm2[];          // [nz] row means (state grid) | [nz] scratch | [256] reduction
    const int env = blockIdx.x, tid = threadIdx.x, nx = g.nx, nz = g.nz, nc = g.nc;
    if (mask && !mask[env]) return;
    f64 *rowmean = sm2, *red = sm2 + 2 * nz;
    const real *sb = st + (size_t)env * g.env_stride;
    const real *b = sb, *u = sb + nc, *w = sb + 3 * (size_t)nc, *ph = phi + (size_t)env * nc;
    const real *pa = phyA ? phyA + (size_t)env * nc : nullptr, *pb = phyA ? phyB + (size_t)env * nc : nullptr;
    auto pnhs = [&](int c) -> f64 { return pa ? ph[c] - (ca * pa[c] + cb * pb[c]) : ph[c]; };
    const f64 kap = nu_kappa[2 * env + 1];
    auto block_sum = [&](f64 v) -> f64 {
        __syncthreads();
        red[tid] = v;
        __syncthreads();
        if (tid == 0) { f64 s = real(0.0); for (int t = 0; t < 256; ++t) s += red[t]; red[0] = s; }
        __syncthreads();
        const f64 r = red[0];
        return r;
    };
    // NaN flag + mean of phi
    f64 bad = real(0.0), psum = real(0.0);
    for (int c = tid; c < nc; c += 256) { bad += (isnan(b[c]) || isnan(u[c]) || isnan(w[c])) ? real(1.0) : real(0.0); psum += pnhs(c); }
    bad = block_sum(bad);
    psum = block_sum(psum);
    if (tid == 0) P.flags[env] = bad > real(0.0) ? 1 : 0;
    const f64 pmean = psum / (f64)nc;
    // float32 state + observations; pHY' by a column scan from the top (thread per column)
    const int stx = nx / P.obs_nx, stz = nz / P.obs_nz;
    float *ob = P.obs + (size_t)env * 5 * P.obs_nz * P.obs_nx, *sbf = P.state32 + (size_t)env * 5 * nc;
    const size_t och = (size_t)P.obs_nz * P.obs_nx;
    const f64 hz = g.dz / 2;
    for (int i = tid; i < nx; i += 256) {
        f64 up = real(0.0), acc = real(0.0);
        for (int k = nz - 1; k >= 0; --k) {
            const int c = k * nx + i;
            const f64 bc = b[c];
            const f64 above = (k == nz - 1) ? (bc + ((g.min_b - bc) / hz) * g.dz) : up;      // Value-BC halo above the top cell
            acc = acc - (real(0.5) * (bc + above)) * g.dz;
            up = bc;
            const f64 vals[5] = {bc, u[c], w[c], acc, pnhs(c) - pmean};
            if (P.write_state)
                for (int q = 0; q < 5; ++q) sbf[(size_t)q * nc + c] = (float)vals[q];
            if ((i % stx) == 0 && (k % stz) == 0) {
                const size_t o = (size_t)(k / stz) * P.obs_nx + (i / stx);
                for (int q = 0; q < 5; ++q) ob[q * och + o] = obs_value2(P, q, vals[q]);
            }
        }
    }
    // Nusselt numbers
    for (int which = 0; which < 2; ++which) {
        const int sx = which ? stx : 1, sz = which ? stz : 1;
        const int mx = nx / sx, mz = nz / sz;
        f64 q1 = real(0.0);
        for (int c = tid; c < nc; c += 256) { const int k = c / nx, i = c - k * nx; if ((i % sx) == 0 && (k % sz) == 0) q1 += b[c] * w[c]; }
        q1 = block_sum(q1);
        __syncthreads();
        for (int k = tid; k < nz; k += 256) {
            f64 s = real(0.0);
            if ((k % sz) == 0) for (int i = 0; i < nx; i += sx) s += b[k * nx + i];
            rowmean[k] = s / (f64)mx;
        }
        __syncthreads();
        if (tid == 0) {
            f64 gsum = real(0.0);
            for (int kk = 0; kk < mz; ++kk) {
                const f64 cur = rowmean[kk * sz];
                if (kk == 0) gsum += rowmean[sz] - cur;
                else if (kk == mz - 1) gsum += cur - rowmean[(kk - 1) * sz];
                else gsum += (rowmean[(kk + 1) * sz] - rowmean[(kk - 1) * sz]) / 2;
            }
            const f64 q2 = kap * (gsum / mz), q1m = q1 / ((f64)mx * mz);
            P.nusselt[(size_t)env * 2 + which] = (q1m - q2) / (kap * g.delta_b / g.lz);
        }
        __syncthreads();
    }
}

}  // namespace RBC3_NS
