// rbc2d_kernel.hpp -- the LDS-resident 2D Boussinesq integrator for gfx950 (MI355X).
//
// One workgroup owns ONE env for a whole control interval (reference: one `step_simulation`
// call, rbc_sim2D_api.jl:75-97, i.e. n_sub RK3 substeps x 3 stages of Oceananigans'
// NonhydrostaticModel).  The prognostic fields u, w, b of the env (3 x NZ x NX float64 =
// 144 KiB at 64x96) stay in the CU's 160 KiB LDS from the first stage to the last; of the
// previous-stage tendencies G^-, G^-_u lives in registers and G^-_b, G^-_w are parked in an
// L2-resident global workspace; apart from that HBM is touched only to load the state at the
// start and to store state + observations at the end.
//
//   thread (i, c)  i = x index (lanes run along x -> conflict-free ds_read_b64 rows)
//                  c = z chunk of CZ=8 cells; NX*NZ/8 = 768 threads = 12 wave64 / CU
//
// Per stage (what Oceananigans' time_step!/update_state! do, see DESIGN.md for file map):
//   hydrostatic-pressure column scan -> tendencies (UpwindBiased(5) advection with
//   boundary-adjacent order reduction, ScalarDiffusivity stress divergence; interior waves run a
//   copy without the wall cases) -> RK3 update in registers -> divergence -> exact Poisson solve:
//   complex FFT-96 (8x12; row p packed with its mirror row NZ-1-p) along x in LDS, tridiagonal
//   solve along z per wavenumber from both walls at once directly on the packed transform, inverse
//   FFT -> projection.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rbc {

constexpr int CZ = 8;              // cells per thread along z
constexpr int MAX_HEATERS = 32;

enum Mode : int { MODE_STEP = 0, MODE_PROJECT = 1, MODE_RANDOM = 2, MODE_TENDENCY = 3 };

// Advecting-velocity reconstruction near the walls (see oracle RBCO_VAR_SYMLEVEL):
// 0 (pinned by the reference's checkpoint ensemble, DESIGN.md "Oracle") = Centered(4) only where
// the 5th-order upwind stencil is allowed, else Centered(2); 1 = Centered(4) with its own buffer.
#ifndef RBC_SYMLEVEL
#define RBC_SYMLEVEL 0
#endif

// Diagnostic build only (-DRBC_STAMPS=1, never shipped): thread 0 of every workgroup accumulates
// shader-clock cycles per phase into stamps[env][32]; outputs are unaffected.
#ifndef RBC_STAMPS
#define RBC_STAMPS 0
#endif
#ifndef RBC_EXPERIMENT_NOPROJECT
#define RBC_EXPERIMENT_NOPROJECT 0   // timing experiment only (WRONG numerics): skips the pressure projection
#endif
// Wave priorities (0..3) of the tendency passes, wall-wave copy / interior copy.  There is no barrier between the u and
// the b pass, so a wave still in the u pass outranks one already in the b pass (they meet at the barrier before the w
// pass); the wall copy is the longer one and outranks the interior copy where both race to a barrier.  Values from a
// sweep on the MI355X (12.38-12.60 ms per launch over eleven combinations; 12.77 ms with no priorities at all).
#ifndef RBC_PU_W
#define RBC_PU_W 3
#define RBC_PU_I 3
#define RBC_PB_W 1
#define RBC_PB_I 0
#define RBC_PW_W 3
#define RBC_PW_I 1
#endif
#ifndef RBC_EXPERIMENT_NOPHYPRE
#define RBC_EXPERIMENT_NOPHYPRE 0   // timing experiment only (WRONG numerics): drops the hydrostatic pre-pass (chunk totals of the column scan)
#endif
#ifndef RBC_EXPERIMENT_NOG0
#define RBC_EXPERIMENT_NOG0 0   // timing experiment only (WRONG numerics): drops the G^- registers
#endif
#if RBC_STAMPS
#define STAMP(id)                                                                                   \
    do {                                                                                            \
        if (tid == 0 || tid == (int)blockDim.x - 1) {                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                             \
            stamp_acc[id] += t_ - stamp_last;                                                       \
            stamp_last = t_;                                                                        \
        }                                                                                           \
    } while (0)
#else
#define STAMP(id) asm volatile("; PHASE_MARK " #id)
#endif

struct Params2D {
    double *fields;            // [B][ b(NZ*NX) | u(NZ*NX) | w((NZ+1)*NX) ] float64
    const float *actions;      // [B][heaters]
    const double *nu_kappa;    // [B][2]: nu = sqrt(Pr/Ra), kappa = 1/sqrt(Pr*Ra) (rbc_sim2D_api.jl:40-41), host-computed
    const uint8_t *mask;       // [B] or nullptr
    const uint64_t *seeds;     // [B] (MODE_RANDOM)
    const double *tri_inv;     // [NZ/2+1][NX/2+1]: half-sweep pivots 1/(piv*NX*f) + junction row (see tri_table)
    float *obs;                // [B][5][obs_nz][obs_nx]
    float *state32;            // [B][5][NZ][NX]
    double *nusselt;           // [B][2]
    int *flags;                // [B]
    double *dbg_g;             // [B][3][NZ][NX] (MODE_TENDENCY)
    double *gpark;             // [B][2][NT][CZ]: previous-stage tendencies of b and w parked between stages
    unsigned long long *stamps; // [B][64] (RBC_STAMPS builds)
    double lx, lz, min_b, delta_b, heater_limit, kick;
    double dx, dz, rdx, rdz, rdx2, rdz2, rhz;   // uniform grid metrics (host-computed so they stay scalar-loadable)
    double dt, dt_last;
    int batch;                 // number of envs B (the packed variant pairs them up: the last workgroup of an odd batch runs one)
    int nsub;                  // number of RK3 substeps (the last one uses dt_last)
    int heaters;
    int mode;
    int write_state;
    int obs_nx, obs_nz;
    int obs_norm, obs_clip;    // rbc_set_obs_normalization: channels [0, obs_norm) of obs are normalised
    float obs_min[5], obs_rng[5], obs_maxval;
    // (kept at the END of the struct: the one-env kernels never read them and keep their argument offsets)
    // float copies of the uniform constants for the packed float32 variant: converted on the host they arrive as SCALAR kernel arguments
    // and are broadcast by op_sel where they are used; converted in the kernel (v_cvt_f32_f64) each one is a VGPR for the whole launch --
    // a dozen of them in a kernel that sits at its 168-register cap.  cpf = (rdz * rdz) * NX in float, the z sweeps' off-diagonal factor.
    float f_dx, f_dz, f_rdx, f_rdz, f_rdx2, f_rdz2, f_rhz, f_min_b, f_dt, f_dt_last, f_cpf;
};

// RBCNormalizeObservation.observation (rbc_normalize_observation.py:66-74) on one float32 sample: the same
// float32 operations in the same order as the numpy expression (no contraction), so results are bit-identical.
__device__ __forceinline__ float obs_value(const Params2D &P, int c, double x)
{
    float o = (float)x;
    if (c < P.obs_norm) {
        o = __fmul_rn(P.obs_maxval, __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, __fsub_rn(o, P.obs_min[c])), P.obs_rng[c]), 1.0f));
        if (P.obs_clip && o == o) o = fminf(fmaxf(o, -P.obs_maxval), P.obs_maxval);      // (a NaN goes through, as in np.clip)
    }
    return o;
}

// ------------------------------------------------------------------------------------------
// counter-based normal deviates (same construction as the test oracle's rbco_normal)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ inline double normal_deviate(uint64_t seed, uint32_t field, uint32_t index)
{
    uint64_t ctr = ((uint64_t)field << 32) | (uint64_t)index;
    uint64_t h0 = splitmix64(seed ^ splitmix64(ctr));
    uint64_t r1 = splitmix64(h0 + 0x9E3779B97F4A7C15ull);
    uint64_t r2 = splitmix64(h0 + 2 * 0x9E3779B97F4A7C15ull);
    double u1 = (double)((r1 >> 11) + 1) * (1.0 / 9007199254740992.0);
    double u2 = (double)(r2 >> 11) * (1.0 / 9007199254740992.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}

// ------------------------------------------------------------------------------------------
// reconstruction stencils (uniform grid).  a..f are six consecutive values psi[-3..+2]
// around the target: centre->face: target face sits between c and d; face->centre alike.
// ------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T left5(T a, T b, T c, T d, T e)
{ return T(2.0 / 60.0) * a - T(13.0 / 60.0) * b + T(47.0 / 60.0) * c + T(27.0 / 60.0) * d - T(3.0 / 60.0) * e; }
template <typename T> __device__ __forceinline__ T right5(T b, T c, T d, T e, T f)
{ return T(27.0 / 60.0) * c - T(3.0 / 60.0) * b + T(47.0 / 60.0) * d - T(13.0 / 60.0) * e + T(2.0 / 60.0) * f; }
template <typename T> __device__ __forceinline__ T left3(T b, T c, T d)
{ return T(5.0 / 6.0) * c - T(1.0 / 6.0) * b + T(2.0 / 6.0) * d; }
template <typename T> __device__ __forceinline__ T right3(T c, T d, T e)
{ return T(2.0 / 6.0) * c + T(5.0 / 6.0) * d - T(1.0 / 6.0) * e; }
template <typename T> __device__ __forceinline__ T sym4(T b, T c, T d, T e)
{ return T(7.0 / 12.0) * (c + d) - T(1.0 / 12.0) * (b + e); }

// upwinded value: vel>0 takes the left-biased reconstruction (== upwind_biased_product/vel).
// Both reconstructions are materialised (empty asm) so hipcc emits one v_cndmask pair instead of
// a divergent if/else around each stencil: the sign of the velocity varies from lane to lane.
#ifndef RBC_BRANCHFREE
#define RBC_BRANCHFREE 1
#endif
template <typename T> __device__ __forceinline__ T pick(T vel, T L, T R)
{
#if RBC_BRANCHFREE
    asm volatile("" : "+v"(L), "+v"(R));
#endif
    return vel * (vel > T(0) ? L : R);
}
template <typename T> __device__ __forceinline__ T upw5(T vel, T a, T b, T c, T d, T e, T f)
{ return pick(vel, left5(a, b, c, d, e), right5(b, c, d, e, f)); }

// wall-aware version: ok5/ok3 select 5th / 3rd / 1st order (both biases share one test)
template <typename T> __device__ __forceinline__ T upwz(T vel, T a, T b, T c, T d, T e, T f, bool ok5, bool ok3)
{
    T L = ok5 ? left5(a, b, c, d, e) : (ok3 ? left3(b, c, d) : c);
    T R = ok5 ? right5(b, c, d, e, f) : (ok3 ? right3(c, d, e) : d);
    return pick(vel, L, R);
}
template <typename T> __device__ __forceinline__ T symz(T b, T c, T d, T e, bool ok4)
{ return ok4 ? sym4(b, c, d, e) : T(0.5) * (c + d); }

// ------------------------------------------------------------------------------------------
// small complex DFTs in registers (forward, e^{-i...}); inverse = call with re/im swapped
// ------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void dft8(T *re, T *im)
{
    const T h = T(0.70710678118654752440);
    // stage 1: pairs (j, j+4)
    T ar[8], ai[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ar[j] = re[j] + re[j + 4]; ai[j] = im[j] + im[j + 4];
        ar[j + 4] = re[j] - re[j + 4]; ai[j + 4] = im[j] - im[j + 4];
    }
    // twiddle the odd half by W8^j
    { T t;
      t = ar[5]; ar[5] = h * (ar[5] + ai[5]); ai[5] = h * (ai[5] - t);           // *(1-i)/sqrt2
      t = ar[6]; ar[6] = ai[6]; ai[6] = -t;                                      // *(-i)
      t = ar[7]; ar[7] = h * (ai[7] - ar[7]); ai[7] = -h * (t + ai[7]); }        // *(-1-i)/sqrt2
    // two DFT-4 on (0,1,2,3) -> even outputs, (4,5,6,7) -> odd outputs
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        T *xr = ar + 4 * g, *xi = ai + 4 * g;
        T s0r = xr[0] + xr[2], s0i = xi[0] + xi[2];
        T d0r = xr[0] - xr[2], d0i = xi[0] - xi[2];
        T s1r = xr[1] + xr[3], s1i = xi[1] + xi[3];
        T d1r = xr[1] - xr[3], d1i = xi[1] - xi[3];
        // outputs k=0..3 of the DFT-4: s0+s1, d0 - i d1, s0-s1, d0 + i d1
        re[g + 0] = s0r + s1r; im[g + 0] = s0i + s1i;
        re[g + 2] = d0r + d1i; im[g + 2] = d0i - d1r;
        re[g + 4] = s0r - s1r; im[g + 4] = s0i - s1i;
        re[g + 6] = d0r - d1i; im[g + 6] = d0i + d1r;
    }
}

template <typename T> __device__ __forceinline__ void dft3(T &r0, T &i0, T &r1, T &i1, T &r2, T &i2)
{
    const T s = T(0.86602540378443864676);
    T tr = r1 + r2, ti = i1 + i2;
    T mr = r0 - T(0.5) * tr, mi = i0 - T(0.5) * ti;
    T dr = s * (r1 - r2), di = s * (i1 - i2);
    r0 += tr; i0 += ti;
    r1 = mr + di; i1 = mi - dr;
    r2 = mr - di; i2 = mi + dr;
}

// DFT-12 by the prime-factor (Good-Thomas) map: n=(4n1+3n2)%12, k=(4k1+9k2)%12; no twiddles
template <typename T> __device__ __forceinline__ void dft12(T *re, T *im)
{
    T tr[3][4], ti[3][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        T r0 = re[(3 * n2) % 12], i0 = im[(3 * n2) % 12];
        T r1 = re[(4 + 3 * n2) % 12], i1 = im[(4 + 3 * n2) % 12];
        T r2 = re[(8 + 3 * n2) % 12], i2 = im[(8 + 3 * n2) % 12];
        dft3(r0, i0, r1, i1, r2, i2);
        tr[0][n2] = r0; ti[0][n2] = i0;
        tr[1][n2] = r1; ti[1][n2] = i1;
        tr[2][n2] = r2; ti[2][n2] = i2;
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
        T s0r = tr[k1][0] + tr[k1][2], s0i = ti[k1][0] + ti[k1][2];
        T d0r = tr[k1][0] - tr[k1][2], d0i = ti[k1][0] - ti[k1][2];
        T s1r = tr[k1][1] + tr[k1][3], s1i = ti[k1][1] + ti[k1][3];
        T d1r = tr[k1][1] - tr[k1][3], d1i = ti[k1][1] - ti[k1][3];
        re[(4 * k1) % 12] = s0r + s1r;          im[(4 * k1) % 12] = s0i + s1i;
        re[(4 * k1 + 9) % 12] = d0r + d1i;      im[(4 * k1 + 9) % 12] = d0i - d1r;
        re[(4 * k1 + 18) % 12] = s0r - s1r;     im[(4 * k1 + 18) % 12] = s0i - s1i;
        re[(4 * k1 + 27) % 12] = d0r - d1i;     im[(4 * k1 + 27) % 12] = d0i + d1r;
    }
}

// DFT-16 as two DFT-8 (even / odd samples) + one radix-2 butterfly with W16^k
template <typename T> __device__ __forceinline__ void dft16(T *re, T *im)
{
    T er[8], ei[8], orr[8], oi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { er[j] = re[2 * j]; ei[j] = im[2 * j]; orr[j] = re[2 * j + 1]; oi[j] = im[2 * j + 1]; }
    dft8(er, ei);
    dft8(orr, oi);
    const T c1 = T(0.92387953251128675613), s1 = T(0.38268343236508977173), h = T(0.70710678118654752440);
    // W16^k = wr[k] + i wi[k], k = 0..7
    const T wr[8] = {T(1), c1, h, s1, T(0), -s1, -h, -c1};
    const T wi[8] = {T(0), -s1, -h, -c1, T(-1), -c1, -h, -s1};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const T tr = orr[k] * wr[k] - oi[k] * wi[k], ti = orr[k] * wi[k] + oi[k] * wr[k];
        re[k] = er[k] + tr; im[k] = ei[k] + ti;
        re[k + 8] = er[k] - tr; im[k + 8] = ei[k] - ti;
    }
}

// DFT-24 by the prime-factor map n = (8 n1 + 3 n2) % 24, k = (16 k1 + 9 k2) % 24 (n1, k1 < 3; n2, k2 < 8): no twiddles
template <typename T> __device__ __forceinline__ void dft24(T *re, T *im)
{
    T tr[3][8], ti[3][8];
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) {
        T r0 = re[(3 * n2) % 24], i0 = im[(3 * n2) % 24];
        T r1 = re[(8 + 3 * n2) % 24], i1 = im[(8 + 3 * n2) % 24];
        T r2 = re[(16 + 3 * n2) % 24], i2 = im[(16 + 3 * n2) % 24];
        dft3(r0, i0, r1, i1, r2, i2);
        tr[0][n2] = r0; ti[0][n2] = i0;
        tr[1][n2] = r1; ti[1][n2] = i1;
        tr[2][n2] = r2; ti[2][n2] = i2;
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
        dft8(tr[k1], ti[k1]);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) { re[(16 * k1 + 9 * k2) % 24] = tr[k1][k2]; im[(16 * k1 + 9 * k2) % 24] = ti[k1][k2]; }
    }
}

// the second factor of the x transform: NX = 8 * N2
template <int N2, typename T> __device__ __forceinline__ void dft_n2(T *re, T *im)
{
    static_assert(N2 == 8 || N2 == 12 || N2 == 16 || N2 == 24, "x transform sizes: NX = 8 * {8, 12, 16, 24}");
    if (N2 == 8) dft8(re, im);
    else if (N2 == 12) dft12(re, im);
    else if (N2 == 16) dft16(re, im);
    else dft24(re, im);
}

// ------------------------------------------------------------------------------------------
// Real types of the kernel.  double and float are one env per workgroup.  f32x2 is the PACKED float32 variant: every value
// is a pair (env 2w, env 2w+1) of the workgroup's two envs and every arithmetic instruction is a v_pk_{add,mul,fma}_f32 that
// advances both -- CDNA4 issues a packed-f32 instruction in the slot of one f64 instruction, so the pair runs through exactly
// the f64 kernel's instruction stream (same LDS layout with 8-byte slots, same registers) at two envs per instruction.
// ------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct LaneT { static constexpr int N = 1; typedef T S; };
template <> struct LaneT<f32x2> { static constexpr int N = 2; typedef float S; };
template <typename T> __device__ __forceinline__ T bc(double x) { return (T)x; }                  // a double constant in every lane
template <> __device__ __forceinline__ f32x2 bc<f32x2>(double x) { return f32x2((float)x); }
// the same for a constant that exists in both precisions (Params2D: double for the one-env kernels, float for the packed variant)
template <typename T> __device__ __forceinline__ T bc2(double d, float) { return (T)d; }
template <> __device__ __forceinline__ f32x2 bc2<f32x2>(double, float f) { return f32x2(f); }
template <> __device__ __forceinline__ float bc2<float>(double, float f) { return f; }      // one env per workgroup in float32: same reason
__device__ __forceinline__ double lane(double x, int) { return x; }
__device__ __forceinline__ float lane(float x, int) { return x; }
__device__ __forceinline__ float lane(f32x2 x, int e) { return e ? x.y : x.x; }
__device__ __forceinline__ void set_lane(double &x, int, double v) { x = v; }
__device__ __forceinline__ void set_lane(float &x, int, float v) { x = v; }
__device__ __forceinline__ void set_lane(f32x2 &x, int e, float v) { if (e) x.y = v; else x.x = v; }
// two consecutive reals as one aligned unit (16-byte accesses for 8-byte reals) of the G^- workspace
template <typename T> struct alignas(2 * sizeof(T)) Two { T x, y; };

// ------------------------------------------------------------------------------------------
// LDS layout: rows interleave the three fields, F(k, f, i) = lds[(3k + f) * NX + i] with
// f = 0:u 1:w 2:b (the b slot doubles as the Poisson right-hand side / potential).  One
// address VGPR per stencil column then reaches every field and row through the 16-bit
// immediate offset of ds_read_b64.
// ------------------------------------------------------------------------------------------
template <int NX, int NZ, typename T = double>
struct Geo {
    static_assert(NX % 64 == 0 || NX == 96, "lanes run along x: NX in {64, 96, 128, 192}");
    static_assert(NZ % (2 * CZ) == 0, "NZ must be a multiple of 16");
    static constexpr int N2 = NX / 8;                     // x transform = 8 x N2 Cooley-Tukey (N2 in {8, 12, 16, 24})
    static constexpr int NC = NZ / CZ;
    static constexpr int NT = NX * NC;                    // threads: one per column and z chunk
    static_assert(NT <= 1024, "one workgroup per env: NX * NZ <= 8192");
    static constexpr int NCELL = NX * NZ;
    static constexpr int NH = NX / 2 + 1;                 // stored Fourier columns
    static constexpr int NXP = (NX + 63) / 64 * 64;       // NX rounded up to whole waves (z sweeps: lanes [0,NX) up, [NXP,NXP+NX) down)
    static_assert(NT >= 2 * NXP, "the two z sweeps need 2 * NXP threads");
    static constexpr int RS = 3 * NX;                     // LDS row stride (reals)
    static constexpr int FU = 0, FW = NX, FB = 2 * NX;    // field offsets inside a row
    static constexpr int GUARD = 3;                       // stencil rows that may be touched beyond a wall
    // [scratch | NZ field rows | GUARD rows]: the scratch (twiddles 2*NX, then reductions / column-scan partials / pivot rows:
    // NT + 130) doubles as the guard below row 0, so every stencil row offset is a compile-time immediate (values read from
    // guards are selected away)
    static constexpr int TW = 2 * NX;
    static constexpr int SCRATCH = (TW + NT + 130 > GUARD * RS) ? TW + NT + 130 : GUARD * RS;
    // The z solve stages the pivot table (NZ/2+1 rows x NH modes) in LDS that is idle during the projection: rows [0, TSPLIT)
    // + one junction row (NX reals) over the reduction scratch, the rest + a junction row over the top guard rows; the little
    // that does not fit there extends the allocation.  TSPLIT is a multiple of the sweeps' prefetch block.
    static constexpr int TROWS = NZ / 2 + 1, ZBLK = 8;
    static constexpr int tsplit()
    {
        int t = (SCRATCH - TW - NX) / NH / ZBLK * ZBLK;
        if (t > 16) t = 16;
        if (t > NZ / 2) t = NZ / 2;
        return t < 0 ? 0 : t;
    }
    static constexpr int TSPLIT = tsplit();
    static constexpr int ZTAIL = ((TROWS - TSPLIT) * NH + NX > GUARD * RS) ? (TROWS - TSPLIT) * NH + NX - GUARD * RS : 0;
    // packed variant: the per-env pairs (nu, kappa) -- scalars of the workgroup in the one-env kernels, VGPR pairs that live (spilled) through
    // the whole launch otherwise -- sit behind the tail and are read where a pass needs them
    static constexpr int XTRA = (LaneT<T>::N > 1) ? 2 : 0;
    static constexpr int XOFF = 3 * NCELL + GUARD * RS + ZTAIL;           // relative to the first field row
    static constexpr size_t LDS_BYTES = (size_t)(SCRATCH + 3 * NCELL + GUARD * RS + ZTAIL + XTRA) * sizeof(T);
    static_assert(LDS_BYTES <= 163840, "LDS budget of one CU");
    static constexpr size_t ENV_STRIDE = (size_t)(3 * NZ + 1) * NX;   // doubles per env in `fields`
    // workgroups that share a CU (LDS and the 2048-thread limit decide) -> waves per SIMD the register allocation must allow
    static constexpr int WG_PER_CU = (2 * LDS_BYTES <= 163840 && 2 * NT <= 2048) ? 2 : 1;
    static constexpr int WAVES_PER_SIMD = (WG_PER_CU * NT / 64 + 3) / 4;
};

// Re-derive per-thread indices inside each phase instead of keeping dozens of loop-invariant
// address registers alive across the whole stage loop (hipcc hoists them, runs out of VGPRs and
// then spills/reloads them through scratch at ~1.5k cycles a reload): an empty volatile asm makes
// the value opaque, so everything computed from it is rematerialised where it is used.
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// Workgroup barrier for LDS hand-offs only.  __syncthreads() also drains the vector-memory queue
// (s_waitcnt vmcnt(0)), which would expose the latency of the G^- prefetch loads and park stores
// that are deliberately left in flight across phases; nothing inside the stage loop hands GLOBAL
// data between threads, so waiting for this wave's LDS traffic is sufficient.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// position of Fourier mode m inside a transformed row (digit-reversed 8 x N2 order)
template <int N2> __device__ __forceinline__ int mode_pos(int m) { return N2 * (m & 7) + (m >> 3); }

// deterministic block reduction (sum) through LDS scratch (NT + 65 reals); partial sums are stored in the working
// precision, accumulated in double; result broadcast to all threads
template <int NT, typename T>
__device__ inline double block_sum(double v, T *scr, int tid)
{
    __syncthreads();
    scr[tid] = (T)v;
    __syncthreads();
    if (tid < 64) {
        double s = 0.0;
        for (int j = tid; j < NT; j += 64) s += (double)scr[j];
        scr[NT + tid] = (T)s;
    }
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int j = 0; j < 64; ++j) s += (double)scr[NT + j];
        scr[NT + 64] = (T)s;
    }
    __syncthreads();
    return (double)scr[NT + 64];
}

// ------------------------------------------------------------------------------------------
// Poisson solve + projection for one stage.  On entry the u,w slots of the thread's own cells
// hold U* (written by the caller, not yet fenced); the b slot is scratch (rhs -> phi).  On exit
// u,w hold the projected velocities (own cells written, not fenced) and the b slot holds phi
// (its mean is NOT removed).
// ------------------------------------------------------------------------------------------
template <int NX, int NZ, typename T>
__device__ __forceinline__ void project(T *__restrict__ lds, const T *__restrict__ tw,
                                        const double *__restrict__ tri_inv, T dts, T rdx, T rdz,
                                        int tid_in, unsigned long long *stamp_acc, unsigned long long &stamp_last,
                                        T (&un)[CZ], T (&wn)[CZ], float cpf_host = 0.0f)
{
    using G = Geo<NX, NZ, T>;
    constexpr int RS = G::RS, FU = G::FU, FW = G::FW, FB = G::FB, N2 = G::N2;
    (void)stamp_acc; (void)stamp_last;
    int tid = tid_in;
    lds_barrier();
    STAMP(5);
    tid = opaque(tid_in);
    // Stage the pivot table of the z solve (NZ/2+1 rows x NH modes, L2-resident) into LDS that is idle during the
    // projection: rows 0..15 over the reduction scratch (the column-scan partials of this stage are consumed), the
    // rest over the top guard rows (their content is never used).  Loads are issued here, stored after the rhs.
    constexpr int TROWS = G::TROWS, TSPLIT = G::TSPLIT, TN = TROWS * G::NH, TPER = (TN + G::NT - 1) / G::NT;
    static_assert(TSPLIT * G::NH + NX <= G::SCRATCH - G::TW, "pivot rows [0, TSPLIT) + a junction row must fit the reduction scratch");
    static_assert((TROWS - TSPLIT) * G::NH + NX <= G::GUARD * RS + G::ZTAIL, "remaining pivot rows + a junction row must fit above the fields");
    T *tabA = const_cast<T *>(tw) + G::TW;
    T *tabB = lds + NZ * RS - TSPLIT * G::NH;               // indexed with the global row number
    T tstage[TPER];
#pragma unroll
    for (int q = 0; q < TPER; ++q) { const int idx = tid + q * G::NT; tstage[q] = bc<T>(tri_inv[min(idx, TN - 1)]); }
    // rhs = div(U*)/dts   (solve_for_pressure!, [OC] solve_for_pressure.jl).  un/wn = this thread's own U* cells
    // (also in LDS for the neighbours); only the east u and the w face above the chunk are read.
    {
        const int c = tid / NX, i = tid - c * NX, k0 = c * CZ, ip1 = (i + 1 == NX) ? 0 : i + 1;
        const bool top = (c == G::NC - 1);
        const T rdt = T(1) / dts;
        const T *me = lds + k0 * RS;
        T ue[CZ];
#pragma unroll
        for (int r = 0; r < CZ; ++r) ue[r] = me[r * RS + FU + ip1];
        const T wtop = top ? T(0) : me[CZ * RS + FW + i];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < CZ; ++r) {
            const T whi = (r == CZ - 1) ? wtop : wn[r + 1];
            lds[(k0 + r) * RS + FB + i] = ((ue[r] - un[r]) * rdx + (whi - wn[r]) * rdz) * rdt;
        }
    }
#pragma unroll
    for (int q = 0; q < TPER; ++q) {
        const int idx = tid + q * G::NT;
        if (idx < TN) (idx < TSPLIT * G::NH ? tabA : tabB)[idx] = tstage[q];
    }
    lds_barrier();
    STAMP(6);
    tid = opaque(tid_in);
    // ---- forward FFT along x, two rows packed as one complex sequence: row p (real part) with its mirror
    //      image NZ-1-p (imaginary part), see the z solve below ------------------------------------------
    if (tid < N2 * (NZ / 2)) {          // pass A: DFT-8 over n1 for fixed n2, twiddle W_NX^(n2*k1)
        const int p = tid / N2, n2 = tid - N2 * p;
        T *R = lds + p * RS + FB, *I = lds + (NZ - 1 - p) * RS + FB;
        T re[8], im[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) { re[n1] = R[N2 * n1 + n2]; im[n1] = I[N2 * n1 + n2]; }
        dft8(re, im);
#pragma unroll
        for (int k1 = 1; k1 < 8; ++k1) {
            const T c = tw[2 * (n2 * 8 + k1)], s = tw[2 * (n2 * 8 + k1) + 1];  // W = c - i s
            T t = re[k1];
            re[k1] = t * c + im[k1] * s;
            im[k1] = im[k1] * c - t * s;
        }
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) { R[N2 * k1 + n2] = re[k1]; I[N2 * k1 + n2] = im[k1]; }
    }
    lds_barrier();
    STAMP(7);
    tid = opaque(tid_in);
    if (tid < 8 * (NZ / 2)) {           // pass B: DFT-N2 over n2 for fixed k1 -> mode k1+8*k2 at N2*k1+k2
        const int p = tid / 8, k1 = tid - 8 * p;
        T *R = lds + p * RS + FB + N2 * k1, *I = lds + (NZ - 1 - p) * RS + FB + N2 * k1;
        T re[N2], im[N2];
#pragma unroll
        for (int n = 0; n < N2; ++n) { re[n] = R[n]; im[n] = I[n]; }
        dft_n2<N2>(re, im);
#pragma unroll
        for (int n = 0; n < N2; ++n) { R[n] = re[n]; I[n] = im[n]; }
    }
    lds_barrier();
    STAMP(8);
    tid = opaque(tid_in);
    // ---- z solve per wavenumber -------------------------------------------------------------
    // The packed transform Z_p[m] = A_p[m] + i A_{NZ-1-p}[m] (A_k = spectrum of row k) is never unpacked.  The
    // operator is real and mirror symmetric in z, so eliminating from both walls applies, at step p, the SAME
    // real recurrence to row p (sweeping up) and to row NZ-1-p (sweeping down): by linearity it can run on
    // Re Z_p (stored in row p) and Im Z_p (stored in row NZ-1-p) directly, for all NX columns (column of mode
    // m at mode_pos(m), eigenvalue index min(m, NX-m)).  Only the 2x2 junction between rows NZ/2-1 and NZ/2
    // mixes the two sweeps: with P = y_up + i y_dn at mode m and P' at mode NX-m, conj(P') = y_up - i y_dn, so
    //   x_up + i x_dn = jf [ P - i c conj(P') ]  =>  Re: jf (Re P[m] - c Im P[NX-m]),  Im: jf (Im P[m] - c Re P[NX-m]).
    // The inverse transform of the packed solution returns phi_p (real part) and phi_{NZ-1-p} (imaginary part).
    // The NX real tridiagonal recurrences per packed array, eliminated from BOTH
    //     walls at once: lanes 0..95 sweep rows 0..NZ/2-1 upward, lanes 128..223 sweep rows NZ-1..NZ/2
    //     downward (the operator is mirror symmetric, so both use the same pivots), they meet in a
    //     2x2 junction and substitute back outward.  y/x overwrite the column in place.
    constexpr int HALF = NZ / 2;
    constexpr int NXP = G::NXP;
    const bool sw_up = tid < NX, sw_dn = (tid >= NXP) && (tid < NXP + NX);
    const int tj = sw_up ? tid : tid - NXP;
    const int tm = min(tj, NX - tj);
    // cp_k = tab_k * cpf.  Packed pairs: computed here the product is a loop-invariant VGPR pair that hipcc hoists out of the stage loop,
    // spills, and reloads from scratch at the head of both z sweeps -- on the serial recurrence of the phase in which four waves work;
    // the host hands the same float product over as a kernel argument, which stays a scalar operand.
    T cpf;
    if constexpr (LaneT<T>::N > 1) cpf = bc2<T>(0.0, cpf_host); else cpf = rdz * rdz * bc<T>((double)NX);
    const int tp = (tj == 0) ? 0 : NX - tj;                   // junction partner column (mode NX-m)
    T *colb = lds + FB + mode_pos<N2>(tj);
    T *jctA = tabA + TSPLIT * G::NH, *jctB = tabB + TROWS * G::NH;
    // Both sweeps are written with the direction as a compile-time constant and fully unrolled, so every LDS
    // access is base + immediate and a row costs two loads, two fp64 ops and a store: the four active waves
    // sit alone on their SIMDs and the phase is bound by their instruction count.
    constexpr int BLK = G::ZBLK;                               // rows fetched ahead of the recurrence
    static_assert(HALF % BLK == 0 && TSPLIT % BLK == 0, "z-sweep blocks must not straddle the table split");
    // A ds instruction's immediate offset is 16 bits: from ONE base register only the first RPB rows of a column are reachable
    // (28 at 96 float64 columns), and hipcc spends a v_add_u32 on every access beyond -- two per row in the down sweep, whose rows
    // all lie beyond, in a phase that four lone waves issue instruction by instruction.  Opaque per-base offsets give every
    // block of RPB rows its own base register, computed once per sweep.
    constexpr int RPB = 65535 / (RS * (int)sizeof(T));
    constexpr int NBASE = (NZ + RPB - 1) / RPB;
    int rbase[NBASE];
#pragma unroll
    for (int q = 0; q < NBASE; ++q) rbase[q] = (q == 0) ? 0 : opaque(q * RPB * RS);
    auto row = [&](int k) -> T & { return colb[rbase[k / RPB] + (k % RPB) * RS]; };
    auto fwd = [&](auto dir) {
        constexpr int DIR = decltype(dir)::value;
        T y = T(0);
        T rr[BLK], tt[BLK];
#pragma unroll
        for (int j = 0; j < BLK; ++j) { rr[j] = row(DIR > 0 ? j : NZ - 1 - j); tt[j] = (j < TSPLIT ? tabA : tabB)[j * G::NH + tm]; }
#pragma unroll
        for (int s0 = 0; s0 < HALF; s0 += BLK) {
            T rn[BLK], tn[BLK];
            if (s0 + BLK < HALF) {
#pragma unroll
                for (int j = 0; j < BLK; ++j) {
                    const int sidx = s0 + BLK + j;
                    rn[j] = row(DIR > 0 ? sidx : NZ - 1 - sidx);
                    tn[j] = (sidx < TSPLIT ? tabA : tabB)[sidx * G::NH + tm];
                }
            }
#pragma unroll
            for (int j = 0; j < BLK; ++j) {
                y = tt[j] * (rr[j] - cpf * y);                       // y_k = (r_k - c y_{k-1}) / piv_k
                row(DIR > 0 ? s0 + j : NZ - 1 - s0 - j) = y;
            }
            if (s0 + BLK < HALF) {
#pragma unroll
                for (int j = 0; j < BLK; ++j) { rr[j] = rn[j]; tt[j] = tn[j]; }
            }
        }
        (DIR > 0 ? jctA : jctB)[tj] = y;       // junction values travel through side rows: the other sweep overwrites in place
    };
    if (tid < NXP) { if (sw_up) fwd(std::integral_constant<int, 1>{}); }
    else if (tid < 2 * NXP) { if (sw_dn) fwd(std::integral_constant<int, -1>{}); }
    lds_barrier();
    STAMP(17);
    auto bwd = [&](auto dir) {
        constexpr int DIR = decltype(dir)::value;
        const T ya = DIR > 0 ? jctA[tj] : jctA[tp], yb = DIR > 0 ? jctB[tp] : jctB[tj];
        const T c = (HALF - 1 < TSPLIT ? tabA : tabB)[(HALF - 1) * G::NH + tm] * cpf;
        const T jf = (HALF < TSPLIT ? tabA : tabB)[HALF * G::NH + tm];   // 1/(1-c^2); 0 for the singular mean mode
        T x = DIR > 0 ? (ya - c * yb) * jf : (yb - c * ya) * jf;
        if (tm == 0) x = DIR > 0 ? ya : T(0);                                  // pin the mean mode (mean removed on output)
        row(DIR > 0 ? HALF - 1 : HALF) = x;
        // rows HALF-2 .. 0 (sweep-local numbering), fetched BLK ahead; the first block is one row short
        T yy[BLK], cc[BLK];
#pragma unroll
        for (int j = 0; j < BLK; ++j) {
            const int sidx = HALF - 2 - j;
            yy[j] = row(DIR > 0 ? sidx : NZ - 1 - sidx);
            cc[j] = (sidx < TSPLIT ? tabA : tabB)[sidx * G::NH + tm];
        }
#pragma unroll
        for (int s0 = HALF - 2; s0 >= 0; s0 -= BLK) {
            T yn[BLK], cn[BLK];
#pragma unroll
            for (int j = 0; j < BLK; ++j) {
                const int sidx = s0 - BLK - j;
                if (sidx >= 0) {
                    yn[j] = row(DIR > 0 ? sidx : NZ - 1 - sidx);
                    cn[j] = (sidx < TSPLIT ? tabA : tabB)[sidx * G::NH + tm];
                }
            }
#pragma unroll
            for (int j = 0; j < BLK; ++j) {
                const int sidx = s0 - j;
                if (sidx >= 0) {
                    x = yy[j] - (cc[j] * cpf) * x;
                    row(DIR > 0 ? sidx : NZ - 1 - sidx) = x;
                }
            }
#pragma unroll
            for (int j = 0; j < BLK; ++j) { yy[j] = yn[j]; cc[j] = cn[j]; }
        }
    };
    if (tid < NXP) { if (sw_up) bwd(std::integral_constant<int, 1>{}); }
    else if (tid < 2 * NXP) { if (sw_dn) bwd(std::integral_constant<int, -1>{}); }
    lds_barrier();
    STAMP(18);
    tid = opaque(tid_in);
    // ---- inverse FFT (swap re<->im roles) --------------------------------------------------
    if (tid < 8 * (NZ / 2)) {
        const int p = tid / 8, k1 = tid - 8 * p;
        T *R = lds + p * RS + FB + N2 * k1, *I = lds + (NZ - 1 - p) * RS + FB + N2 * k1;
        T re[N2], im[N2];
#pragma unroll
        for (int n = 0; n < N2; ++n) { re[n] = R[n]; im[n] = I[n]; }
        dft_n2<N2>(im, re);
        // inverse twiddle W_NX^(-n2*k1): (a+ib)(c+is)
#pragma unroll
        for (int n2 = 1; n2 < N2; ++n2) {
            const T c = tw[2 * (n2 * 8 + k1)], s = tw[2 * (n2 * 8 + k1) + 1];
            T t = re[n2];
            re[n2] = t * c - im[n2] * s;
            im[n2] = im[n2] * c + t * s;
        }
#pragma unroll
        for (int n = 0; n < N2; ++n) { R[n] = re[n]; I[n] = im[n]; }
    }
    lds_barrier();
    STAMP(10);
    tid = opaque(tid_in);
    if (tid < N2 * (NZ / 2)) {
        const int p = tid / N2, n2 = tid - N2 * p;
        T *R = lds + p * RS + FB, *I = lds + (NZ - 1 - p) * RS + FB;
        T re[8], im[8];
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) { re[k1] = R[N2 * k1 + n2]; im[k1] = I[N2 * k1 + n2]; }
        dft8(im, re);
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) { R[N2 * n1 + n2] = re[n1]; I[N2 * n1 + n2] = im[n1]; }
    }
    lds_barrier();
    STAMP(11);
    tid = opaque(tid_in);
    // ---- pressure_correct_velocities! ([OC] pressure_correction.jl) -------------------------
    {
        const int c = tid / NX, i = tid - c * NX, k0 = c * CZ, im1 = (i == 0) ? NX - 1 : i - 1;
        T *me = lds + k0 * RS;
        T pc[CZ], pw[CZ];
#pragma unroll
        for (int r = 0; r < CZ; ++r) { pc[r] = me[r * RS + FB + i]; pw[r] = me[r * RS + FB + im1]; }
        T pdn = (k0 > 0) ? me[-RS + FB + i] : T(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < CZ; ++r) {
            un[r] -= (pc[r] - pw[r]) * rdx * dts;
            if (k0 + r > 0) wn[r] -= (pc[r] - pdn) * rdz * dts;
            pdn = pc[r];
            me[r * RS + FU + i] = un[r];
            me[r * RS + FW + i] = wn[r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------
// DBG = true: the instantiation rbc_debug_tendencies launches (MODE_TENDENCY: one stage, the three tendency fields written to
// dbg_g).  The production kernel is compiled WITHOUT the hook: its ~24 uniform `if (dbg)` store blocks per stage put a possible
// store in front of every use of a prefetched G^- value, which turns those waits into vmcnt(0) (they then also wait for the
// park loads issued since), and cost registers: 82.6k -> 85.2k env-steps/s in float64, 143.7k -> 154.7k in packed float32.
template <int NX, int NZ, typename T, bool DBG = false>
__global__ __launch_bounds__(NX *(NZ / CZ), (Geo<NX, NZ, T>::WAVES_PER_SIMD)) void rbc2d_kernel(const Params2D P)
{
    using G = Geo<NX, NZ, T>;
    typedef T real;                  // working precision: double (the reference's Float64) or float (the fp32 variant)
    constexpr int RS = G::RS, FU = G::FU, FW = G::FW, FB = G::FB, N2 = G::N2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_bytes_raw[];
    real *lds_raw = reinterpret_cast<real *>(lds_bytes_raw);
    real *Sc = lds_raw;
    real *lds = lds_raw + G::SCRATCH;   // field rows start here
    real *tw = Sc;                   // [N2][8][2] twiddles c,s of W_NX^(n2*k1) = c - i s
    real *scr = Sc + G::TW;          // reductions / column-scan partials (NT + 130 reals)

    constexpr int NL = LaneT<T>::N;
    typedef typename LaneT<T>::S scal;
    if constexpr (NL == 1) {
        if (P.mask && !P.mask[blockIdx.x]) return;
    }
    const int tid = threadIdx.x;
    unsigned long long stamp_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = 0;
#if RBC_STAMPS
    if (tid == 0 || tid == (int)blockDim.x - 1) stamp_last = __builtin_amdgcn_s_memtime();
#endif
    const int c = tid / NX, i = tid - c * NX;
    const int k0 = c * CZ;
    const bool top = (c == G::NC - 1);

    const real dx = bc2<real>(P.dx, P.f_dx), dz = bc2<real>(P.dz, P.f_dz), rdx = bc2<real>(P.rdx, P.f_rdx), rdz = bc2<real>(P.rdz, P.f_rdz),
               rdx2 = bc2<real>(P.rdx2, P.f_rdx2), rdz2 = bc2<real>(P.rdz2, P.f_rdz2);
    const real min_b = bc2<real>(P.min_b, P.f_min_b);
    real nu, kap, Tb;
    real bn[CZ];
    if (tid < NX) {                  // N2 * 8 = NX twiddles
        const int n2 = tid / 8, k1 = tid - 8 * n2;
        double s, cc;
        sincospi(2.0 * (double)(n2 * k1) / (double)NX, &s, &cc);
        tw[2 * tid] = bc<real>(cc); tw[2 * tid + 1] = bc<real>(s);
    }

    if constexpr (NL == 1) {
    // one env per workgroup: the round-1 prologue verbatim (see the note at the outputs: the kernel sits at hipcc's scalar
    // register limit and the shape of this code decides where the spills land)
    const int env = blockIdx.x;
    nu = (real)P.nu_kappa[2 * env]; kap = (real)P.nu_kappa[2 * env + 1];
    double *gf = P.fields + (size_t)env * G::ENV_STRIDE;
    double *gb_ = gf, *gu_ = gf + G::NCELL, *gw_ = gf + 2 * G::NCELL;
    // ---- A10: heater profile of this column (collate_actions_colin, rbc_sim2D.jl:87-133) ----
    {
        double Tbd;
        const int n = P.heaters;
        const double ampl = P.heater_limit, hdx = 0.03;
        const bool zero_action = (P.mode == MODE_PROJECT || P.mode == MODE_RANDOM || P.actions == nullptr);
        const float *act = zero_action ? nullptr : P.actions + (size_t)env * n;
        double mean = 0.0, dev = 0.0;
        for (int a = 0; a < n; ++a) mean += ampl * (act ? (double)act[a] : 0.0);
        mean /= n;
        for (int a = 0; a < n; ++a) dev = fmax(dev, fabs(ampl * (act ? (double)act[a] : 0.0) - mean));
        double K2 = dev / ampl;
        if (!(K2 > 1.0)) K2 = 1.0;
        const double seg = P.lx / n, x = (i + 0.5) * dx;
        int xs = (int)floor(x / seg) + 1;
        if (xs > n) xs = n;
        const int a0 = (xs == 1) ? n : xs - 1, a2 = (xs == n) ? 1 : xs + 1;
        const double T0 = 2 + (ampl * (act ? (double)act[a0 - 1] : 0.0) - mean) / K2;
        const double T1 = 2 + (ampl * (act ? (double)act[xs - 1] : 0.0) - mean) / K2;
        const double T2 = 2 + (ampl * (act ? (double)act[a2 - 1] : 0.0) - mean) / K2;
        const double xp = x - (xs - 1) * seg;
        if (xp < hdx) Tbd = T0 + ((T0 - T1) / (4 * hdx * hdx * hdx)) * (xp - 2 * hdx) * (xp + hdx) * (xp + hdx);
        else if (xp >= seg - hdx) Tbd = T1 + ((T1 - T2) / (4 * hdx * hdx * hdx)) * (xp - seg - 2 * hdx) * (xp - seg + hdx) * (xp - seg + hdx);
        else Tbd = T1;
        Tb = (real)Tbd;
    }

    // ---- load (or generate) the state of the own cells --------------------------------------
    {
        real *me = lds + k0 * RS + i;
        if (P.mode == MODE_RANDOM) {   // initialize_model, rbc_sim2D.jl:163-171
            const uint64_t seed = P.seeds[env];
            for (int r = 0; r < CZ; ++r) {
                const int k = k0 + r;
                const uint32_t id = (uint32_t)(k * NX + i);
                me[r * RS + FU] = (real)(P.kick * normal_deviate(seed, 0, id));
                me[r * RS + FW] = (k == 0) ? real(0) : (real)(P.kick * normal_deviate(seed, 1, id));
                const double z = (k + 0.5) * P.dz;
                const double v = P.min_b + (P.lz - z) * P.delta_b / 2 + P.kick * normal_deviate(seed, 2, id);
                me[r * RS + FB] = (real)fmin(fmax(v, P.min_b), P.min_b + P.delta_b);
            }
        } else {
#pragma unroll
            for (int r = 0; r < CZ; ++r) {
                const int k = k0 + r;
                me[r * RS + FB] = (real)gb_[k * NX + i];
                me[r * RS + FU] = (real)gu_[k * NX + i];
                me[r * RS + FW] = (k == 0) ? real(0) : (real)gw_[k * NX + i];
            }
        }
#pragma unroll
        for (int r = 0; r < CZ; ++r) bn[r] = me[r * RS + FB];
    }

    } else {
    // lanes: the env(s) of this workgroup.  A lane beyond the batch (odd batch, packed variant) or masked out of a reset
    // computes along on a copy of lane 0's env and stores nothing.
    const int wg = blockIdx.x;
    int envs[NL];
    bool live[NL];
    bool any_live = false;
#pragma unroll
    for (int e = 0; e < NL; ++e) {
        const int id = wg * NL + e;
        envs[e] = (id < P.batch) ? id : wg * NL;
        live[e] = (id < P.batch) && !(P.mask && !P.mask[envs[e]]);
        any_live = any_live || live[e];
    }
    if (!any_live) return;
#pragma unroll
    for (int e = 0; e < NL; ++e) { set_lane(nu, e, (scal)P.nu_kappa[2 * envs[e]]); set_lane(kap, e, (scal)P.nu_kappa[2 * envs[e] + 1]); }
    if (tid == 0) { lds[G::XOFF] = nu; lds[G::XOFF + 1] = kap; }      // read back per pass (NU / KAP below); a barrier precedes the stage loop

    double *gf[NL];
#pragma unroll
    for (int e = 0; e < NL; ++e) gf[e] = P.fields + (size_t)envs[e] * G::ENV_STRIDE;      // b | u | w of lane e

    // ---- A10: heater profile of this column (collate_actions_colin, rbc_sim2D.jl:87-133) ----
#pragma unroll
    for (int e = 0; e < NL; ++e) {
        double Tbd;
        const int n = P.heaters;
        const double ampl = P.heater_limit, hdx = 0.03;
        const bool zero_action = (P.mode == MODE_PROJECT || P.mode == MODE_RANDOM || P.actions == nullptr);
        const float *act = zero_action ? nullptr : P.actions + (size_t)envs[e] * n;
        double mean = 0.0, dev = 0.0;
        for (int a = 0; a < n; ++a) mean += ampl * (act ? (double)act[a] : 0.0);
        mean /= n;
        for (int a = 0; a < n; ++a) dev = fmax(dev, fabs(ampl * (act ? (double)act[a] : 0.0) - mean));
        double K2 = dev / ampl;
        if (!(K2 > 1.0)) K2 = 1.0;
        const double seg = P.lx / n, x = (i + 0.5) * P.dx;
        int xs = (int)floor(x / seg) + 1;
        if (xs > n) xs = n;
        const int a0 = (xs == 1) ? n : xs - 1, a2 = (xs == n) ? 1 : xs + 1;
        const double T0 = 2 + (ampl * (act ? (double)act[a0 - 1] : 0.0) - mean) / K2;
        const double T1 = 2 + (ampl * (act ? (double)act[xs - 1] : 0.0) - mean) / K2;
        const double T2 = 2 + (ampl * (act ? (double)act[a2 - 1] : 0.0) - mean) / K2;
        const double xp = x - (xs - 1) * seg;
        if (xp < hdx) Tbd = T0 + ((T0 - T1) / (4 * hdx * hdx * hdx)) * (xp - 2 * hdx) * (xp + hdx) * (xp + hdx);
        else if (xp >= seg - hdx) Tbd = T1 + ((T1 - T2) / (4 * hdx * hdx * hdx)) * (xp - seg - 2 * hdx) * (xp - seg + hdx) * (xp - seg + hdx);
        else Tbd = T1;
        set_lane(Tb, e, (scal)Tbd);
    }

    // ---- load (or generate) the state of the own cells --------------------------------------
    {
        real *me = lds + k0 * RS + i;
        if (P.mode == MODE_RANDOM) {   // initialize_model, rbc_sim2D.jl:163-171
            for (int r = 0; r < CZ; ++r) {
                const int k = k0 + r;
                const uint32_t id = (uint32_t)(k * NX + i);
                real vu, vw, vb;
#pragma unroll
                for (int e = 0; e < NL; ++e) {
                    const uint64_t seed = P.seeds[envs[e]];
                    set_lane(vu, e, (scal)(P.kick * normal_deviate(seed, 0, id)));
                    set_lane(vw, e, (k == 0) ? (scal)0 : (scal)(P.kick * normal_deviate(seed, 1, id)));
                    const double z = (k + 0.5) * P.dz;
                    const double v = P.min_b + (P.lz - z) * P.delta_b / 2 + P.kick * normal_deviate(seed, 2, id);
                    set_lane(vb, e, (scal)fmin(fmax(v, P.min_b), P.min_b + P.delta_b));
                }
                me[r * RS + FU] = vu; me[r * RS + FW] = vw; me[r * RS + FB] = vb;
            }
        } else {
#pragma unroll
            for (int r = 0; r < CZ; ++r) {
                const int k = k0 + r;
                real vu, vw, vb;
#pragma unroll
                for (int e = 0; e < NL; ++e) {
                    set_lane(vb, e, (scal)gf[e][k * NX + i]);
                    set_lane(vu, e, (scal)gf[e][G::NCELL + k * NX + i]);
                    set_lane(vw, e, (k == 0) ? (scal)0 : (scal)gf[e][2 * G::NCELL + k * NX + i]);
                }
                me[r * RS + FB] = vb; me[r * RS + FU] = vu; me[r * RS + FW] = vw;
            }
        }
#pragma unroll
        for (int r = 0; r < CZ; ++r) bn[r] = me[r * RS + FB];
    }

    }
    // G^- (previous stage tendencies).  Only G^-_u stays in registers across stages; G^-_b and
    // G^-_w are parked in an L2-resident global workspace ([field][tid][8]: four 16-byte accesses
    // per thread, a wave covers 4 KiB contiguously) and fetched back one pass ahead of their use, so
    // their load latency hides under the preceding pass and the Poisson phases run with 32 fewer
    // live VGPRs.
    real g0u[CZ];
#pragma unroll
    for (int r = 0; r < CZ; ++r) g0u[r] = real(0);
    typedef Two<real> dbl2;
    real *gpark = reinterpret_cast<real *>(P.gpark);       // the workspace holds the working precision, one slice per workgroup
    // this thread's 64-byte slot of field f (0: b, 1: w), re-derived from the opaque tid at each of its four uses per stage: an
    // address pair kept across the stage is spilled by hipcc (three of them were, with their reloads), and ANY scratch access
    // inside the stage loop makes the s_waitcnt in front of it a vmcnt(0) that also waits for the park loads just issued
    auto park = [&](int f) -> dbl2 * { return reinterpret_cast<dbl2 *>(gpark + (((size_t)blockIdx.x * 2 + f) * G::NT + opaque(tid)) * CZ); };

    const real rhz = bc2<real>(P.rhz, P.f_rhz);   // 1/(dz/2); dz/2 is a power of two at the reference sizes, so *rhz == /(dz/2) bitwise
    auto NU = [&]() -> real { if constexpr (NL == 1) return nu; else return lds[G::XOFF]; };
    auto KAP = [&]() -> real { if constexpr (NL == 1) return kap; else return lds[G::XOFF + 1]; };
    const int nstage = (P.mode == MODE_STEP) ? 3 * P.nsub : ((P.mode == MODE_TENDENCY) ? 1 : 0);

    if (P.mode == MODE_PROJECT || P.mode == MODE_RANDOM) {
        // set!'s incompressibility projection with unit time step ([OC] set_nonhydrostatic_model.jl)
        real u0[CZ], w0[CZ];
        lds_barrier();
#pragma unroll
        for (int r = 0; r < CZ; ++r) { u0[r] = lds[(k0 + r) * RS + FU + i]; w0[r] = lds[(k0 + r) * RS + FW + i]; }
        project<NX, NZ, T>(lds, tw, P.tri_inv, real(1), rdx, rdz, tid, stamp_acc, stamp_last, u0, w0, P.f_cpf);
        // the b slot now holds phi (pNHS); b stays in registers
    } else {
        lds_barrier();
    }

    // row offset of (relative row rr, field f).  Rows outside the domain fall into the guard
    // rows / front scratch: whatever is read there is selected away by the wall-adjacent stencils
    auto off = [&](int rr, int f) -> int { return rr * RS + f; };

    const int wave0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x) & ~63;       // first thread of this wave
    const bool wall_wave = (wave0 / NX == 0) || ((wave0 + 63) / NX >= G::NC - 1);
    STAMP(0);
    for (int st = 0; st < nstage; ++st) {
        const int sub = st / 3, ph = st - 3 * sub;
        const bool last_sub = (sub == P.nsub - 1);
        const real dt = bc2<real>(last_sub ? P.dt_last : P.dt, last_sub ? P.f_dt_last : P.f_dt);
        // Le-Moin RK3 ([OC] TimeSteppers/runge_kutta_3.jl)
        const real gam = (ph == 0) ? real(8.0 / 15.0) : (ph == 1 ? real(5.0 / 12.0) : real(3.0 / 4.0));
        const real zet = (ph == 0) ? real(0.0) : (ph == 1 ? real(-17.0 / 60.0) : real(-5.0 / 12.0));
        const real dts = (gam + zet) * dt;
        const bool dbg = DBG && (P.mode == MODE_TENDENCY);
        double *dg = dbg ? P.dbg_g + (size_t)blockIdx.x * NL * 3 * G::NCELL : nullptr;   // operator-level hook: lane 0 (the packed variant is not bound to it)
        STAMP(14);
        // column addresses of the 7-point x stencil, shared by all fields and rows; re-derived every
        // stage (see opaque()) so they do not occupy registers through the Poisson phases
        const int ti = opaque(tid);
        const int c = ti / NX, i = ti - c * NX, k0 = c * CZ;
        const int ip1 = (i + 1 == NX) ? 0 : i + 1, ip2 = (ip1 + 1 == NX) ? 0 : ip1 + 1, ip3 = (ip2 + 1 == NX) ? 0 : ip2 + 1;
        const int im1 = (i == 0) ? NX - 1 : i - 1, im2 = (im1 == 0) ? NX - 1 : im1 - 1, im3 = (im2 == 0) ? NX - 1 : im2 - 1;
        const real *cm3 = lds + k0 * RS + im3, *cm2 = lds + k0 * RS + im2, *cm1 = lds + k0 * RS + im1;
        const real *cc0 = lds + k0 * RS + i;
        const real *cp1 = lds + k0 * RS + ip1, *cp2 = lds + k0 * RS + ip2, *cp3 = lds + k0 * RS + ip3;

        const bool use_g0 = (ph != 0);      // zeta^1 = 0: the first stage of a substep needs no G^-
        const bool keep_g = (ph != 2);      // the tendencies of the last stage are never reused
        real un[CZ], wn[CZ];
        // The three tendency passes exist twice: waves whose threads all sit in interior chunks (8 of the 12 at
        // NZ=64, two per SIMD) run a copy in which bot/top are compile-time false, i.e. without the 3rd/1st-order
        // fallbacks, wall selects and halo rows.  Both copies execute the same barriers.
        auto tendencies = [&](auto wall_tag) __attribute__((always_inline)) {
            constexpr bool WALL = decltype(wall_tag)::value;
            const bool bot = WALL && (c == 0), top = WALL && (c == G::NC - 1);
            // ---- hydrostatic pressure anomaly ([OC] update_hydrostatic_pressure.jl) --------------
            // pHY'[k] = pHY'[k+1] - b_face(k+1) dz.  G_u needs pHY'[i,k]-pHY'[i-1,k]
            //         = -dz * sum_{k'>=k} mean(db[k'], db[k'+1]),  db[k] = b[i,k]-b[i-1,k].
            // Pre-pass: the chunk totals; the u pass below walks DOWN its chunk and accumulates.
            real db_top;   // db at the first row above the chunk (halo row for the top chunk)
#if RBC_EXPERIMENT_NOPHYPRE
            db_top = real(0); scr[c * NX + i] = real(0);
#else
            {
                {   // branch-free: the top chunk takes the Value-BC halo row, the others the row above
                    const real cN = cc0[(CZ - 1) * RS + FB], cM = cm1[(CZ - 1) * RS + FB];
                    const real hN = cN + ((min_b - cN) * rhz) * dz, hM = cM + ((min_b - cM) * rhz) * dz;
                    const real dn = cc0[off(CZ, FB)] - cm1[off(CZ, FB)];
                    db_top = top ? (hN - hM) : dn;
                }
                real acc = real(0), dbu = db_top;
                real dcol[CZ], dmin[CZ];       // all loads first: hipcc otherwise serialises eight LDS round trips
#pragma unroll
                for (int r = 0; r < CZ; ++r) { dcol[r] = cc0[r * RS + FB]; dmin[r] = cm1[r * RS + FB]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = CZ - 1; r >= 0; --r) {
                    const real d = dcol[r] - dmin[r];
                    acc += real(0.5) * (d + dbu);
                    dbu = d;
                }
                scr[c * NX + i] = acc;
            }
#endif
            STAMP(15);
            real g0b[CZ], g0w[CZ];
            lds_barrier();
            STAMP(1);
#pragma unroll
            for (int r = 0; r < CZ; ++r) g0b[r] = real(0);
            if (use_g0) {                                                             // uniform branch: zeta^1 = 0 needs no G^-; lands under the u pass
                const dbl2 *pk = park(0);
#pragma unroll
                for (int r = 0; r < CZ; r += 2) { const dbl2 v = pk[r / 2]; g0b[r] = v.x; g0b[r + 1] = v.y; }
            }
            __builtin_amdgcn_s_setprio(WALL ? RBC_PU_W : RBC_PU_I);      // waves still in an earlier pass outrank those ahead of them
            // ======================= u tendency (walks down the chunk) ==============================
            {
                real above = real(0);
                for (int cc = G::NC - 1; cc > c; --cc) above += scr[cc * NX + i];
                real pacc = real(0), dbu = db_top;
                // z window of u around face k+1: rows k-2..k+3  (w0..w5), face between w2|w3
                real w0, w1, w2, w3, w4, w5;
                w0 = cc0[off(CZ - 3, FU)]; w1 = cc0[off(CZ - 2, FU)]; w2 = cc0[off(CZ - 1, FU)];
                w3 = cc0[off(CZ, FU)]; w4 = cc0[off(CZ + 1, FU)]; w5 = cc0[off(CZ + 2, FU)];
                // top face of the chunk (face k0+CZ): advecting w in x (Centered(4), periodic)
                real wm_hi, wc_hi, fz_hi, uup;
                {   // branch-free (clamped loads, selects): the top wall face carries no flux and w=0
                    wm_hi = top ? real(0) : cm1[off(CZ, FW)]; wc_hi = top ? real(0) : cc0[off(CZ, FW)];
                    const real wt = sym4(cm2[off(CZ, FW)], wm_hi, wc_hi, cp1[off(CZ, FW)]);
                    const real f = upwz(wt, w0, w1, w2, w3, w4, w5, true, true);
                    fz_hi = top ? real(0) : f;
                    uup = top ? (w2 + ((real(0) - w2) * rhz) * dz) : w3;   // halo row above the top cell
                }
#pragma unroll
                for (int r = CZ - 1; r >= 0; --r) {
                    // slide the window down: now around face k (rows k-3..k+2)
                    w5 = w4; w4 = w3; w3 = w2; w2 = w1; w1 = w0; w0 = cc0[off(r - 3, FU)];
                    const real u0 = w3;
                    const real um3 = cm3[r * RS + FU], um2 = cm2[r * RS + FU], um1 = cm1[r * RS + FU];
                    const real up1 = cp1[r * RS + FU], up2 = cp2[r * RS + FU], up3 = cp3[r * RS + FU];
                    // flux_uu at centres i-1 and i  (advective_momentum_flux_Uu)
                    const real ut_w = sym4(um2, um1, u0, up1);
                    const real ut_e = sym4(um1, u0, up1, up2);
                    const real fx_w = upw5(ut_w, um3, um2, um1, u0, up1, up2);
                    const real fx_e = upw5(ut_e, um2, um1, u0, up1, up2, up3);
                    // bottom face k of this cell
                    real fz_lo, udn, wm_lo, wc_lo;
                    {   // the bottom wall row of w is identically 0 in LDS, so its flux vanishes by itself
                        wm_lo = cm1[r * RS + FW]; wc_lo = cc0[r * RS + FW];
                        const real wt = sym4(cm2[r * RS + FW], wm_lo, wc_lo, cp1[r * RS + FW]);
                        // face k: 5th if 3<=k<=NZ-3, 3rd if 2<=k<=NZ-2, else 1st
                        const bool ok5 = ((r >= 3) || !bot) && ((r <= CZ - 3) || !top);
                        const bool ok3 = ((r >= 2) || !bot) && ((r <= CZ - 2) || !top);
                        fz_lo = upwz(wt, w0, w1, w2, w3, w4, w5, ok5, ok3);
                        const bool wall = (r == 0) && bot;
                        fz_lo = wall ? real(0) : fz_lo;
                        udn = wall ? (u0 + ((u0 - real(0)) * rhz) * (-dz)) : w2;
                    }
                    const real adv = (fx_e - fx_w) * rdx + (fz_hi - fz_lo) * rdz;
                    // -d_j tau_1j, tau = -2 nu Sigma ([OC] TurbulenceClosures, isotropic ScalarDiffusivity)
                    const real vis = NU() * (real(2) * ((up1 - u0) - (u0 - um1)) * rdx2
                                             + (((uup - u0) * rdz + (wc_hi - wm_hi) * rdx) - ((u0 - udn) * rdz + (wc_lo - wm_lo) * rdx)) * rdz);
                    // hydrostatic pressure gradient
                    const real d = cc0[r * RS + FB] - cm1[r * RS + FB];
                    pacc += real(0.5) * (d + dbu);
                    dbu = d;
                    const real dphy = -(pacc + above) * dz;
                    const real g = vis - adv - dphy * rdx;
                    if (dbg) dg[G::NCELL + (k0 + r) * NX + i] = (double)lane(g, 0);
    #if RBC_EXPERIMENT_NOG0
                    un[r] = u0 + dt * (gam * g);
    #else
                    un[r] = u0 + dt * (gam * g + zet * g0u[r]);
                    asm volatile("" : "+v"(un[r]));   // pin the update here: hipcc otherwise sinks it past the Poisson solve
                    g0u[r] = g;
    #endif
                    fz_hi = fz_lo; uup = u0; wm_hi = wm_lo; wc_hi = wc_lo;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            STAMP(2);
            __builtin_amdgcn_s_setprio(WALL ? RBC_PB_W : RBC_PB_I);
            // ======================= b tendency (walks up) ===========================================
#pragma unroll
            for (int r = 0; r < CZ; ++r) g0w[r] = real(0);
            if (use_g0) {                                                             // lands under the b pass
                const dbl2 *pk = park(1);
#pragma unroll
                for (int r = 0; r < CZ; r += 2) { const dbl2 v = pk[r / 2]; g0w[r] = v.x; g0w[r + 1] = v.y; }
            }
            {
                real w0, w1, w2, w3, w4, w5;   // b rows k-3..k+2 around face k (face between w2|w3)
                w0 = cc0[off(-3, FB)]; w1 = cc0[off(-2, FB)]; w2 = cc0[off(-1, FB)];
                w3 = cc0[off(0, FB)]; w4 = cc0[off(1, FB)]; w5 = cc0[off(2, FB)];
                real fz_lo = bot ? real(0) : upwz(cc0[FW], w0, w1, w2, w3, w4, w5, true, true);
                real bdn = bot ? (w3 + ((w3 - Tb) * rhz) * (-dz)) : w2;      // Value BC halo below the first cell
#pragma unroll
                for (int r = 0; r < CZ; ++r) {
                    w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = cc0[off(r + 3, FB)];   // now around face k+1
                    const real b0 = w2;
                    const real bm3 = cm3[r * RS + FB], bm2 = cm2[r * RS + FB], bm1 = cm1[r * RS + FB];
                    const real bp1 = cp1[r * RS + FB], bp2 = cp2[r * RS + FB], bp3 = cp3[r * RS + FB];
                    const real ui = cc0[r * RS + FU], ue = cp1[r * RS + FU];
                    const real fx_i = upw5(ui, bm3, bm2, bm1, b0, bp1, bp2);
                    const real fx_e = upw5(ue, bm2, bm1, b0, bp1, bp2, bp3);
                    real fz_hi, bup;
                    {
                        const bool ok5 = ((r + 1 >= 3) || !bot) && ((r + 1 <= CZ - 3) || !top);
                        const bool ok3 = ((r + 1 >= 2) || !bot) && ((r + 1 <= CZ - 2) || !top);
                        const bool wall = (r == CZ - 1) && top;
                        fz_hi = upwz(cc0[off(r + 1, FW)], w0, w1, w2, w3, w4, w5, ok5, ok3);
                        fz_hi = wall ? real(0) : fz_hi;
                        bup = wall ? (b0 + ((min_b - b0) * rhz) * dz) : w3;
                    }
                    const real adv = (fx_e - fx_i) * rdx + (fz_hi - fz_lo) * rdz;
                    const real dif = KAP() * (((bp1 - b0) - (b0 - bm1)) * rdx2 + ((bup - b0) - (b0 - bdn)) * rdz2);
                    const real g = dif - adv;
                    if (dbg) dg[(k0 + r) * NX + i] = (double)lane(g, 0);
    #if RBC_EXPERIMENT_NOG0
                    bn[r] = b0 + dt * (gam * g);
    #else
                    bn[r] = b0 + dt * (gam * g + zet * g0b[r]);
                    asm volatile("" : "+v"(bn[r]));
                    g0b[r] = g;
    #endif
                    fz_lo = fz_hi; bdn = b0;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (keep_g) {
dbl2 *pk = park(0);
#pragma unroll
                for (int r = 0; r < CZ; r += 2) { dbl2 v; v.x = g0b[r]; v.y = g0b[r + 1]; pk[r / 2] = v; }
            }
            STAMP(3);
            if (!dbg) {
                // every thread has finished reading the old b: its slot now parks the new u so that the
                // heaviest pass below runs with one new-value array fewer in registers
                lds_barrier();
                real *me = lds + k0 * RS + i;
#pragma unroll
                for (int r = 0; r < CZ; ++r) me[r * RS + FB] = un[r];
            }
            __builtin_amdgcn_s_setprio(WALL ? RBC_PW_W : RBC_PW_I);
            // ======================= w tendency (faces k0..k0+7, walks up) ===========================
            {
                auto wld = [&](int rr) -> real { return (WALL && k0 + rr >= NZ) ? real(0) : cc0[off(rr, FW)]; };
                real w0, w1, w2, w3, w4, w5;   // w faces k-2..k+3 around centre k (between w2|w3)
                w0 = wld(-3); w1 = wld(-2); w2 = wld(-1); w3 = wld(0); w4 = wld(1); w5 = wld(2);   // centre k0-1
                real fz_lo = upwz(sym4(w1, w2, w3, w4), w0, w1, w2, w3, w4, w5, true, true);   // flux_ww at centre k0-1
                fz_lo = bot ? real(0) : fz_lo;
                // u columns at x-faces i and i+1: rows k-2..k+1 around z-face k
                real a0 = cc0[off(-2, FU)], a1 = cc0[off(-1, FU)], a2 = cc0[off(0, FU)], a3 = cc0[off(1, FU)];
                real e0 = cp1[off(-2, FU)], e1 = cp1[off(-1, FU)], e2 = cp1[off(0, FU)], e3 = cp1[off(1, FU)];
#pragma unroll
                for (int r = 0; r < CZ; ++r) {
                    if (r > 0) {
                        a0 = a1; a1 = a2; a2 = a3; a3 = cc0[off(r + 1, FU)];
                        e0 = e1; e1 = e2; e2 = e3; e3 = cp1[off(r + 1, FU)];
                    }
                    w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = wld(r + 3);   // centre k: faces k-2..k+3
                    const real wc = w2;          // w at face k
                    // flux_ww at centre k: 5th if 2<=k<=NZ-3, 3rd if 1<=k<=NZ-2, else 1st
                    const bool c5 = ((r >= 2) || !bot) && ((r <= CZ - 3) || !top);
                    const bool c3 = ((r >= 1) || !bot) && ((r <= CZ - 2) || !top);
    #if RBC_SYMLEVEL
                    const bool c4 = c3;
    #else
                    const bool c4 = c5;
    #endif
                    const real fz_hi = upwz(symz(w1, w2, w3, w4, c4), w0, w1, w2, w3, w4, w5, c5, c3);
                    real g;
                    {
    #if RBC_SYMLEVEL
                        const bool f4 = ((r >= 2) || !bot) && ((r <= CZ - 2) || !top);       // 2<=k<=NZ-2
    #else
                        const bool f4 = ((r >= 3) || !bot) && ((r <= CZ - 3) || !top);       // 3<=k<=NZ-3
    #endif
                        const real ut_w = symz(a0, a1, a2, a3, f4);
                        const real ut_e = symz(e0, e1, e2, e3, f4);
                        const real wm3 = cm3[r * RS + FW], wm2 = cm2[r * RS + FW], wm1 = cm1[r * RS + FW];
                        const real wp1 = cp1[r * RS + FW], wp2 = cp2[r * RS + FW], wp3 = cp3[r * RS + FW];
                        const real fx_w = upw5(ut_w, wm3, wm2, wm1, wc, wp1, wp2);
                        const real fx_e = upw5(ut_e, wm2, wm1, wc, wp1, wp2, wp3);
                        const real adv = (fx_e - fx_w) * rdx + (fz_hi - fz_lo) * rdz;
                        const real vis = NU() * ((((e2 - e1) * rdz + (wp1 - wc) * rdx) - ((a2 - a1) * rdz + (wc - wm1) * rdx)) * rdx
                                                 + real(2) * ((w3 - wc) - (wc - w1)) * rdz2);
                        g = ((r == 0) && bot) ? real(0) : (vis - adv);
                    }
                    if (dbg) dg[2 * G::NCELL + (k0 + r) * NX + i] = (double)lane(g, 0);
    #if RBC_EXPERIMENT_NOG0
                    wn[r] = wc + dt * (gam * g);
    #else
                    wn[r] = wc + dt * (gam * g + zet * g0w[r]);
                    asm volatile("" : "+v"(wn[r]));
                    g0w[r] = g;
    #endif
                    fz_lo = fz_hi;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (dbg) return;
            if (keep_g) {
dbl2 *pk = park(1);
#pragma unroll
                for (int r = 0; r < CZ; r += 2) { dbl2 v; v.x = g0w[r]; v.y = g0w[r + 1]; pk[r / 2] = v; }
            }
        };
        // wave priorities inside: the wall copy is the longer one and wins issue arbitration on its SIMD
        if (wall_wave) tendencies(std::true_type{}); else tendencies(std::false_type{});
        __builtin_amdgcn_s_setprio(0);
        if (dbg) return;
        STAMP(4);
        lds_barrier();   // every read of the old state is done
        {
            real *me = lds + k0 * RS + i;
#pragma unroll
            for (int r = 0; r < CZ; ++r) un[r] = me[r * RS + FB];          // U*_u comes back from its parking slot
#pragma unroll
            for (int r = 0; r < CZ; ++r) { me[r * RS + FU] = un[r]; me[r * RS + FW] = wn[r]; }
        }
#if RBC_EXPERIMENT_NOPROJECT
        lds_barrier();          // timing experiment only (WRONG numerics): what the stage costs without the Poisson solve
#else
        project<NX, NZ, T>(lds, tw, P.tri_inv, dts, rdx, rdz, tid, stamp_acc, stamp_last, un, wn, P.f_cpf);
#endif
        STAMP(12);
        if (st + 1 < nstage) {
            lds_barrier();   // phi reads done -> the b slot takes the new b
            real *me = lds + k0 * RS + i;
#pragma unroll
            for (int r = 0; r < CZ; ++r) me[r * RS + FB] = bn[r];
            lds_barrier();
        }
        STAMP(13);
    }

    // =========================== outputs ========================================================
    if constexpr (LaneT<T>::N == 1) {
    // one env per workgroup: this block is the round-1 code verbatim -- hipcc's scalar-register allocation for the whole
    // kernel is at its limit (106 SGPRs, 60 spilled into VGPR lanes) and any change of shape here moves v_readlane
    // instructions into the hot passes (measured: +6 % per launch with the lane-generic block below)
    const int env = blockIdx.x;
    double *gf1 = P.fields + (size_t)env * G::ENV_STRIDE;
    double *gb_ = gf1, *gu_ = gf1 + G::NCELL, *gw_ = gf1 + 2 * G::NCELL;
    // here: LDS u,w = final velocities (own cells), b slot = phi of the last stage, bn = final b
    __syncthreads();
    real un[CZ], wn[CZ];
    {
        const real *me = lds + k0 * RS + i;
#pragma unroll
        for (int r = 0; r < CZ; ++r) { un[r] = me[r * RS + FU]; wn[r] = me[r * RS + FW]; }
    }
    // state back to HBM (always float64: the layout of the reference's checkpoint datasets)
#pragma unroll
    for (int r = 0; r < CZ; ++r) {
        const int k = k0 + r;
        gb_[k * NX + i] = (double)bn[r];
        gu_[k * NX + i] = (double)un[r];
        gw_[k * NX + i] = (double)wn[r];
    }
    if (top) gw_[NZ * NX + i] = 0.0;

    // A13 NaN flag
    double bad = 0.0;
#pragma unroll
    for (int r = 0; r < CZ; ++r) bad += (isnan(bn[r]) || isnan(un[r]) || isnan(wn[r])) ? 1.0 : 0.0;
    bad = block_sum<G::NT>(bad, scr, tid);
    if (tid == 0) P.flags[env] = (bad > 0.0) ? 1 : 0;

    // pNHS = phi - mean(phi)   (the reference solver zeroes the mean mode)
    real ph[CZ];
    double psum = 0.0;
#pragma unroll
    for (int r = 0; r < CZ; ++r) { ph[r] = lds[(k0 + r) * RS + FB + i]; psum += (double)ph[r]; }
    psum = block_sum<G::NT>(psum, scr, tid);
    const real pmean = (real)(psum / (double)G::NCELL);
#pragma unroll
    for (int r = 0; r < CZ; ++r) ph[r] -= pmean;

    // pHY' (absolute) by the same column scan; the b slot (free now) takes the final b
    real phy[CZ];
    {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CZ; ++r) lds[(k0 + r) * RS + FB + i] = bn[r];
        __syncthreads();
        const real babove = top ? (bn[CZ - 1] + ((min_b - bn[CZ - 1]) * rhz) * dz) : lds[(k0 + CZ) * RS + FB + i];
        real acc = real(0);
#pragma unroll
        for (int r = CZ - 1; r >= 0; --r) {
            const real bup = (r == CZ - 1) ? babove : bn[r + 1];
            acc += real(0.5) * (bn[r] + bup);
            phy[r] = acc;
        }
        scr[c * NX + i] = acc;
        __syncthreads();
        real above = real(0);
        for (int cc = G::NC - 1; cc > c; --cc) above += scr[cc * NX + i];
#pragma unroll
        for (int r = 0; r < CZ; ++r) phy[r] = -(phy[r] + above) * dz;
    }
    // now LDS: u, w, b slots = final fields

    // A11 observation / state (float32, channel order b,u,w,pHY',pNHS; layout [c][z][x])
    {
        const int stx = NX / P.obs_nx, stz = NZ / P.obs_nz;
        float *ob = P.obs + (size_t)env * 5 * P.obs_nz * P.obs_nx;
        const size_t och = (size_t)P.obs_nz * P.obs_nx;
        const bool xs = (i % stx) == 0;
#pragma unroll
        for (int r = 0; r < CZ; ++r) {
            const int k = k0 + r;
            if (xs && (k % stz) == 0) {
                const size_t o = (size_t)(k / stz) * P.obs_nx + (i / stx);
                ob[o] = obs_value(P, 0, (double)bn[r]); ob[och + o] = obs_value(P, 1, (double)un[r]); ob[2 * och + o] = obs_value(P, 2, (double)wn[r]);
                ob[3 * och + o] = obs_value(P, 3, (double)phy[r]); ob[4 * och + o] = obs_value(P, 4, (double)ph[r]);
            }
        }
        if (P.write_state) {
            float *sb = P.state32 + (size_t)env * 5 * G::NCELL;
#pragma unroll
            for (int r = 0; r < CZ; ++r) {
                const int o = (k0 + r) * NX + i;
                sb[o] = (float)bn[r]; sb[G::NCELL + o] = (float)un[r]; sb[2 * G::NCELL + o] = (float)wn[r];
                sb[3 * G::NCELL + o] = (float)phy[r]; sb[4 * G::NCELL + o] = (float)ph[r];
            }
        }
    }

    // A12 Nusselt numbers (get_nusselt rbc_sim2D_api.jl:142-163, array_gradient rbc_sim2D.jl:206-220); sums in float64
    const double kapd = P.nu_kappa[2 * env + 1];
    for (int which = 0; which < 2; ++which) {   // 0: full state, 1: sensor grid
        const int stx = which ? NX / P.obs_nx : 1, stz = which ? NZ / P.obs_nz : 1;
        const int mx = NX / stx, mz = NZ / stz;
        double q1 = 0.0;
        const bool xs = (i % stx) == 0;
#pragma unroll
        for (int r = 0; r < CZ; ++r)
            if (xs && ((k0 + r) % stz) == 0) q1 += (double)bn[r] * (double)wn[r];
        q1 = block_sum<G::NT>(q1, scr, tid);
        // row means of T on the (sub)grid: thread t sums segment seg of row
        __syncthreads();
        {
            const int row = tid / N2, seg = tid - N2 * row;   // NT = N2*NZ threads: NX/8 = N2 segments of 8 cells per row
            double s = 0.0;
            if ((row % stz) == 0)
                for (int j = 0; j < 8; ++j) { const int x = 8 * seg + j; if ((x % stx) == 0) s += (double)lds[row * RS + FB + x]; }
            scr[tid] = (real)s;
        }
        __syncthreads();
        if (tid < NZ) {
            double s = 0.0;
            for (int j = 0; j < N2; ++j) s += (double)scr[tid * N2 + j];
            scr[G::NT + 66 + tid] = (real)(s / (double)mx);
        }
        __syncthreads();
        if (tid == 0) {
            const real *tx = scr + G::NT + 66;
            double g = 0.0;
            for (int kk = 0; kk < mz; ++kk) {
                const double cur = (double)tx[kk * stz];
                if (kk == 0) g += (double)tx[stz] - cur;
                else if (kk == mz - 1) g += cur - (double)tx[(kk - 1) * stz];
                else g += ((double)tx[(kk + 1) * stz] - (double)tx[(kk - 1) * stz]) / 2;
            }
            const double q2 = kapd * (g / mz);
            const double q1m = q1 / ((double)mx * mz);
            P.nusselt[(size_t)env * 2 + which] = (q1m - q2) / (kapd * P.delta_b / P.lz);
        }
        __syncthreads();
    }
    STAMP(0);
#if RBC_STAMPS
    if ((tid == 0 || tid == (int)blockDim.x - 1) && P.stamps)
        for (int j = 0; j < 24; ++j) P.stamps[(size_t)env * 64 + (tid ? 32 : 0) + j] = stamp_acc[j];
#endif
    } else {
    // here: LDS u,w = final velocities (own cells), b slot = phi of the last stage, bn = final b
    __syncthreads();
    real un[CZ], wn[CZ];
    {
        const real *me = lds + k0 * RS + i;
#pragma unroll
        for (int r = 0; r < CZ; ++r) { un[r] = me[r * RS + FU]; wn[r] = me[r * RS + FW]; }
    }
    // phi of the last stage (b slot) before the slot takes the final b
    real ph[CZ];
#pragma unroll
    for (int r = 0; r < CZ; ++r) ph[r] = lds[(k0 + r) * RS + FB + i];

    // pHY' (absolute) by the same column scan; the b slot (free now) takes the final b
    real phy[CZ];
    {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CZ; ++r) lds[(k0 + r) * RS + FB + i] = bn[r];
        __syncthreads();
        const real babove = top ? (bn[CZ - 1] + ((min_b - bn[CZ - 1]) * rhz) * dz) : lds[(k0 + CZ) * RS + FB + i];
        real acc = real(0);
#pragma unroll
        for (int r = CZ - 1; r >= 0; --r) {
            const real bup = (r == CZ - 1) ? babove : bn[r + 1];
            acc += real(0.5) * (bn[r] + bup);
            phy[r] = acc;
        }
        scr[c * NX + i] = acc;
        __syncthreads();
        real above = real(0);
        for (int cc = G::NC - 1; cc > c; --cc) above += scr[cc * NX + i];
#pragma unroll
        for (int r = 0; r < CZ; ++r) phy[r] = -(phy[r] + above) * dz;
    }
    // now LDS: u, w, b slots = final fields (of every lane)

    scal *sscr = reinterpret_cast<scal *>(scr);          // reductions run per lane on scalars
    // the lane -> env map is re-derived here from an opaque copy of the workgroup index: kept alive across the stage loop it
    // costs scalar registers that hipcc then spills into VGPR lanes (v_readlane in the hot passes: +6 % on the f64 kernel)
    // The final u, w, b are in LDS as well as in registers: this block reads them from LDS, so that the three 8-row arrays of pairs
    // are dead before it starts (with them alive across both lanes' output code hipcc spilled nine VGPRs here)
    auto FB_ = [&](int r) -> real { return lds[(k0 + r) * RS + FB + i]; };
    auto FU_ = [&](int r) -> real { return lds[(k0 + r) * RS + FU + i]; };
    auto FW_ = [&](int r) -> real { return lds[(k0 + r) * RS + FW + i]; };
    int wg_o = blockIdx.x;
    asm volatile("" : "+s"(wg_o));
#pragma unroll
    for (int e = 0; e < NL; ++e) {
        const int id = wg_o * NL + e;
        if (id >= P.batch || (P.mask && !P.mask[id])) continue;      // uniform over the workgroup: this lane stores nothing
        const int env = id;
        // state back to HBM (always float64: the layout of the reference's checkpoint datasets)
        {
            double *gf_e = P.fields + (size_t)env * G::ENV_STRIDE;
            double *gb_ = gf_e, *gu_ = gf_e + G::NCELL, *gw_ = gf_e + 2 * G::NCELL;
#pragma unroll
            for (int r = 0; r < CZ; ++r) {
                const int k = k0 + r;
                gb_[k * NX + i] = (double)lane(FB_(r), e);
                gu_[k * NX + i] = (double)lane(FU_(r), e);
                gw_[k * NX + i] = (double)lane(FW_(r), e);
            }
            if (top) gw_[NZ * NX + i] = 0.0;
        }

        // A13 NaN flag
        double bad = 0.0;
#pragma unroll
        for (int r = 0; r < CZ; ++r) bad += (isnan(lane(FB_(r), e)) || isnan(lane(FU_(r), e)) || isnan(lane(FW_(r), e))) ? 1.0 : 0.0;
        bad = block_sum<G::NT>(bad, sscr, tid);
        if (tid == 0) P.flags[env] = (bad > 0.0) ? 1 : 0;

        // pNHS = phi - mean(phi)   (the reference solver zeroes the mean mode)
        double psum = 0.0;
#pragma unroll
        for (int r = 0; r < CZ; ++r) psum += (double)lane(ph[r], e);
        psum = block_sum<G::NT>(psum, sscr, tid);
        const scal pmean = (scal)(psum / (double)G::NCELL);

        // A11 observation / state (float32, channel order b,u,w,pHY',pNHS; layout [c][z][x])
        {
            const int stx = NX / P.obs_nx, stz = NZ / P.obs_nz;
            float *ob = P.obs + (size_t)env * 5 * P.obs_nz * P.obs_nx;
            const size_t och = (size_t)P.obs_nz * P.obs_nx;
            const bool xs = (i % stx) == 0;
#pragma unroll
            for (int r = 0; r < CZ; ++r) {
                const int k = k0 + r;
                if (xs && (k % stz) == 0) {
                    const size_t o = (size_t)(k / stz) * P.obs_nx + (i / stx);
                    ob[o] = obs_value(P, 0, (double)lane(FB_(r), e)); ob[och + o] = obs_value(P, 1, (double)lane(FU_(r), e));
                    ob[2 * och + o] = obs_value(P, 2, (double)lane(FW_(r), e));
                    ob[3 * och + o] = obs_value(P, 3, (double)lane(phy[r], e)); ob[4 * och + o] = obs_value(P, 4, (double)(lane(ph[r], e) - pmean));
                }
            }
            if (P.write_state) {
                float *sb = P.state32 + (size_t)env * 5 * G::NCELL;
#pragma unroll
                for (int r = 0; r < CZ; ++r) {
                    const int o = (k0 + r) * NX + i;
                    sb[o] = (float)lane(FB_(r), e); sb[G::NCELL + o] = (float)lane(FU_(r), e); sb[2 * G::NCELL + o] = (float)lane(FW_(r), e);
                    sb[3 * G::NCELL + o] = (float)lane(phy[r], e); sb[4 * G::NCELL + o] = (float)(lane(ph[r], e) - pmean);
                }
            }
        }

        // A12 Nusselt numbers (get_nusselt rbc_sim2D_api.jl:142-163, array_gradient rbc_sim2D.jl:206-220); sums in float64
        const double kapd = P.nu_kappa[2 * env + 1];
        for (int which = 0; which < 2; ++which) {   // 0: full state, 1: sensor grid
            const int stx = which ? NX / P.obs_nx : 1, stz = which ? NZ / P.obs_nz : 1;
            const int mx = NX / stx, mz = NZ / stz;
            double q1 = 0.0;
            const bool xs = (i % stx) == 0;
#pragma unroll
            for (int r = 0; r < CZ; ++r)
                if (xs && ((k0 + r) % stz) == 0) q1 += (double)lane(FB_(r), e) * (double)lane(FW_(r), e);
            q1 = block_sum<G::NT>(q1, sscr, tid);
            // row means of T on the (sub)grid: thread t sums segment seg of row
            __syncthreads();
            {
                const int row = tid / N2, seg = tid - N2 * row;   // NT = N2*NZ threads: NX/8 = N2 segments of 8 cells per row
                double s = 0.0;
                if ((row % stz) == 0)
                    for (int j = 0; j < 8; ++j) { const int x = 8 * seg + j; if ((x % stx) == 0) s += (double)lane(lds[row * RS + FB + x], e); }
                sscr[tid] = (scal)s;
            }
            __syncthreads();
            if (tid < NZ) {
                double s = 0.0;
                for (int j = 0; j < N2; ++j) s += (double)sscr[tid * N2 + j];
                sscr[G::NT + 66 + tid] = (scal)(s / (double)mx);
            }
            __syncthreads();
            if (tid == 0) {
                const scal *tx = sscr + G::NT + 66;
                double g = 0.0;
                for (int kk = 0; kk < mz; ++kk) {
                    const double cur = (double)tx[kk * stz];
                    if (kk == 0) g += (double)tx[stz] - cur;
                    else if (kk == mz - 1) g += cur - (double)tx[(kk - 1) * stz];
                    else g += ((double)tx[(kk + 1) * stz] - (double)tx[(kk - 1) * stz]) / 2;
                }
                const double q2 = kapd * (g / mz);
                const double q1m = q1 / ((double)mx * mz);
                P.nusselt[(size_t)env * 2 + which] = (q1m - q2) / (kapd * P.delta_b / P.lz);
            }
            __syncthreads();
        }
    }
    STAMP(0);
#if RBC_STAMPS
    if ((tid == 0 || tid == (int)blockDim.x - 1) && P.stamps)
        for (int j = 0; j < 24; ++j) P.stamps[(size_t)blockIdx.x * 64 + (tid ? 32 : 0) + j] = stamp_acc[j];
#endif
    }
}

#ifndef RBC2D_TEMPLATE_ONLY      /* the two small non-template kernels below belong to rbc_api.hip's translation unit only */
// RBCRewardShaping.compute_cell_distances (wrappers/rbc_reward_shaping.py:85-140) for a batch of mid-line signals, one
// wave64 per env: uy[env * stride + i], i < nx <= 256 (the float32 w channel of the state at row nz/2 - 1).
//   peaks = scipy.signal.find_peaks(uy, height): strict rise before, strict fall after, a flat top counts once at its
//           middle sample (rounded down), the end samples never count; kept if uy[peak] >= height (compared in float32,
//           as numpy compares a float32 array with a python float);
//   for every pair i < j of peaks: direct = |x_j - x_i| with x_k = k * (lx / nx)  (np.linspace(0, lx, nx, endpoint=False)),
//           around = lx - direct, d = min(direct, around); d = 0 if the signal stays positive between the two peaks the
//           short way round ([i, j) if direct < around, else [j, nx) and [0, i));  result = max d (0 with fewer than 2 peaks).
// Same float32 comparisons and float64 arithmetic as the numpy code: results are bit-identical (tests/test_env_gpu.py).
__global__ __launch_bounds__(64) void cell_distance_kernel(const float *__restrict__ uy, size_t stride, int nx, double lx, float height,
                                                           double *__restrict__ out)
{
    __shared__ float x[256];
    __shared__ int cnt[257];        // cnt[i] = number of samples j < i with uy[j] <= 0 (or NaN)
    __shared__ int peaks[128];
    __shared__ int npk;
    const int env = blockIdx.x, lane = threadIdx.x;
    for (int i = lane; i < nx; i += 64) x[i] = uy[(size_t)env * stride + i];
    if (lane == 0) npk = 0;
    __syncthreads();
    if (lane == 0) {
        int c = 0;
        for (int i = 0; i < nx; ++i) { cnt[i] = c; c += (x[i] > 0.0f) ? 0 : 1; }
        cnt[nx] = c;
    }
    for (int i = lane + 1; i < nx - 1; i += 64) {
        if (x[i - 1] < x[i]) {                                   // a rise: i starts a (possibly flat) top
            int j = i + 1;
            while (j < nx - 1 && x[j] == x[i]) ++j;
            if (x[j] < x[i]) {
                const int p = (i + j - 1) / 2;
                if (x[p] >= height) peaks[atomicAdd(&npk, 1)] = p;
            }
        }
    }
    __syncthreads();
    const int P = npk;
    const double step = lx / (double)nx;
    double best = 0.0;
    for (int a = 0; a < P; ++a)
        for (int b = a + 1 + lane; b < P; b += 64) {
            const int i = min(peaks[a], peaks[b]), j = max(peaks[a], peaks[b]);
            // x_j - x_i with each product rounded on its own, as numpy does: an fma contraction would change the last bit
            // (HIP's __dmul_rn is a plain multiply that hipcc may still contract, hence the opaque values)
            double xj = (double)j * step, xi = (double)i * step;
            asm volatile("" : "+v"(xj), "+v"(xi));
            const double direct = fabs(xj - xi), around = lx - direct;
            double d = (around < direct) ? around : direct;
            if (direct < around) { if (cnt[j] - cnt[i] == 0) d = 0.0; }
            else if (cnt[nx] - cnt[j] == 0 && cnt[i] == 0) d = 0.0;
            best = fmax(best, d);
        }
    for (int off = 32; off > 0; off >>= 1) best = fmax(best, __shfl_xor(best, off, 64));
    if (lane == 0) out[env] = best;
}

// streaming-copy kernel for rbc_copy_ceiling: 16 bytes per lane, grid-stride, 4096 workgroups of 256 threads
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy16_kernel(const uint4 *__restrict__ src_, uint4 *__restrict__ dst_, size_t n)
{
    const u32x4 *__restrict__ src = reinterpret_cast<const u32x4 *>(src_);
    u32x4 *__restrict__ dst = reinterpret_cast<u32x4 *>(dst_);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {                 // four independent 16-byte loads in flight per lane
        const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
        __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
#endif   /* RBC2D_TEMPLATE_ONLY */

}  // namespace rbc
