// rbc3d_kernels.hpp -- streaming 3D Boussinesq integrator for gfx950.
//
// Replaces `step_simulation`/`initialize_simulation` of src/rbc_gym/sim/rbc_sim3D_api.jl (77-101,
// 17-72) for a batch of envs.  A 3D env (32x48x48: 2.4 MB of fp64 state) does not fit one CU's LDS,
// so unlike the 2D kernel the state lives in HBM/L2 and every RK3 stage is a short sequence of launches
// (rbc3d_host.hpp: advance3d / project3d; DESIGN.md section 3b, NOTES.md section 5b):
//   k3_tile_all (LDS-tiled tendencies of (u,v) and (w,b), U* into the other state buffer; z-marching and
//   cell-per-thread kernels as fallbacks for other grid shapes) -> k3_rhs_fft_pair (divergence of two mirror
//   slabs as one complex 2D FFT in LDS) -> k3_thomas_pair_fused (z solve per (kx,ky) from both walls on the
//   packed spectrum) -> k3_ifft_pair (inverse FFT + u, v correction) -> k3_correct_w.
// The batch is cut into env groups whose chains run on separate streams (one captured graph per env-step).
// Same discretisation as the 2D kernel; layouts [k][j][i].
// The second half of the file is the streaming-2D mode: 2D grids without an LDS-resident kernel on these
// kernels with ny = 1 (FLAT tiles, k2s_* kernels, the one-kernel projection k2s_project_fused).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rbc2d_kernel.hpp"   // normal_deviate, stencil helpers

// Precision: the kernels below exist twice, `rbc3::` with real = double (the reference's arithmetic; every parity test runs on
// it) and `rbc3f::` with real = float (rbc_config.precision = RBC_PRECISION_F32: state, tendencies, spectrum and potential in
// float32 -- half the bytes of a path that is bound by memory traffic and latency; outputs, Nusselt sums, the bottom-plate
// table and the diffusivities stay float64).  rbc3d_kernels_body.hpp is included once per precision; what does not depend on it
// (geometry, action preprocessing, the 2D heater profile, the float32 observation helpers) lives in rbc3c.
namespace rbc3c {

struct Geo3 {
    int nx, ny, nz;
    int nc;            // nx*ny*nz
    int nw;            // nx*ny*(nz+1)
    size_t env_stride; // doubles per env in one state buffer: b,u,v (nc each) + w (nw)
    double dx, dy, dz, rdx, rdy, rdz, lx, ly, lz;
    double min_b, delta_b, heater_limit, kick;
    int heaters;
    int wall_nx;       // 0: the 3D envs' heaters x heaters table (bottom_T below); > 0: streaming-2D mode (ny = 1), the bottom-plate
                       // temperature is a per-column table of wall_nx values per env (collate_actions_colin, k2s_wall)
};

// periodic 7-point index table around i
__device__ __forceinline__ void wrap7(int i, int n, int *o)
{
    if (n >= 3) {
#pragma unroll
        for (int d = -3; d <= 3; ++d) { int t = i + d; t += (t < 0) ? n : 0; t -= (t >= n) ? n : 0; o[d + 3] = t; }
    } else {               // ny = 1 (streaming-2D mode) or 2: more than one wrap
#pragma unroll
        for (int d = -3; d <= 3; ++d) { int t = (i + d) % n; o[d + 3] = t + ((t < 0) ? n : 0); }
    }
}

// wall temperature of column (i,j): bottom_T, rbc_sim3D.jl:131-141 (act = preprocessed 8x8 table)
__device__ __forceinline__ size_t wall_stride(const Geo3 &g) { return g.wall_nx ? (size_t)g.wall_nx : (size_t)g.heaters * g.heaters; }

__device__ __forceinline__ double bottom_T(const Geo3 &g, const double *act, int i, int j)
{
    if (g.wall_nx) return act[i];
    const int n = g.heaters;
    const double x = (i + 0.5) * g.dx, y = (j + 0.5) * g.dy;
    int a = (int)floor(x / g.lx * n) + 1, c = (int)floor(y / g.ly * n) + 1;
    a = min(max(a, 1), n); c = min(max(c, 1), n);
    return act[(a - 1) * n + (c - 1)];
}

// ---- preprocess_action, rbc_sim3D.jl:111-128: act_T[env][n*n] -----------------------------------
__global__ void __launch_bounds__(64) k3_preprocess(Geo3 g, const float *actions, double *actT, int raw_zero)
{
    const int env = blockIdx.x, nn = g.heaters * g.heaters, lane = threadIdx.x;
    double *o = actT + (size_t)env * nn;
    if (raw_zero || actions == nullptr) { for (int a = lane; a < nn; a += 64) o[a] = 0.0; return; }
    const float *in = actions + (size_t)env * nn;
    __shared__ double sh[2];
    if (lane == 0) {                       // the two reductions stay serial in index order (same sums as the oracle, bit for bit)
        double mean = 0.0, mx = 0.0;
        for (int a = 0; a < nn; ++a) mean += (double)in[a];
        mean /= nn;
        for (int a = 0; a < nn; ++a) mx = fmax(mx, fabs((double)in[a] - mean));
        sh[0] = mean; sh[1] = mx > 1.0 ? mx : 1.0;
    }
    __syncthreads();
    const double mean = sh[0], K = sh[1];
    for (int a = lane; a < nn; a += 64) o[a] = (g.min_b + g.delta_b) + (((double)in[a] - mean) / K) * g.heater_limit;
}

// bottom-plate temperature of every column: collate_actions_colin (rbc_sim2D.jl:87-133), same arithmetic as rbc2d_kernel
__global__ void __launch_bounds__(128) k2s_wall(Geo3 g, const float *actions, double *wall, int zero_action, int B)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= g.nx * B) return;
    const int env = t / g.nx, i = t - env * g.nx;
    const int n = g.heaters;
    const double ampl = g.heater_limit, hdx = 0.03;
    const float *act = (zero_action || actions == nullptr) ? nullptr : actions + (size_t)env * n;
    double mean = 0.0, dev = 0.0;
    for (int a = 0; a < n; ++a) mean += ampl * (act ? (double)act[a] : 0.0);
    mean /= n;
    for (int a = 0; a < n; ++a) dev = fmax(dev, fabs(ampl * (act ? (double)act[a] : 0.0) - mean));
    double K2 = dev / ampl;
    if (!(K2 > 1.0)) K2 = 1.0;
    const double seg = g.lx / n, x = (i + 0.5) * g.dx;
    int xs = (int)floor(x / seg) + 1;
    if (xs > n) xs = n;
    const int a0 = (xs == 1) ? n : xs - 1, a2 = (xs == n) ? 1 : xs + 1;
    const double T0 = 2 + (ampl * (act ? (double)act[a0 - 1] : 0.0) - mean) / K2;
    const double T1 = 2 + (ampl * (act ? (double)act[xs - 1] : 0.0) - mean) / K2;
    const double T2 = 2 + (ampl * (act ? (double)act[a2 - 1] : 0.0) - mean) / K2;
    const double xp = x - (xs - 1) * seg;
    double Tb;
    if (xp < hdx) Tb = T0 + ((T0 - T1) / (4 * hdx * hdx * hdx)) * (xp - 2 * hdx) * (xp + hdx) * (xp + hdx);
    else if (xp >= seg - hdx) Tb = T1 + ((T1 - T2) / (4 * hdx * hdx * hdx)) * (xp - seg - 2 * hdx) * (xp - seg + hdx) * (xp - seg + hdx);
    else Tb = T1;
    wall[(size_t)env * g.nx + i] = Tb;
}

// state of every env NOT marked in `mark` copied back from `src` (host3::lead_substep3d); 4-byte words, blockIdx.x = env
__global__ void __launch_bounds__(256) k3_restore_unmarked(uint32_t *dst, const uint32_t *src, const uint8_t *mark, size_t words_per_env)
{
    const int env = blockIdx.x;
    if (mark[env]) return;
    const size_t base = (size_t)env * words_per_env;
    for (size_t i = (size_t)blockIdx.y * blockDim.x + threadIdx.x; i < words_per_env; i += (size_t)gridDim.y * blockDim.x)
        dst[base + i] = src[base + i];
}

// Occupies its stream's hardware queue for `ticks` of the 100 MHz wall clock (one wave; bounded by construction).  create3d uses it
// to find out which of its streams share a hardware queue (host3::streams_run_side_by_side).
__global__ void __launch_bounds__(64) k3_hold(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// RBCNormalizeObservation fused into the 3D output kernel's float32 state write (the 3D observation IS the state, rbc3D.py:229-232):
// obs[c] <- maxval * (2 * (obs[c] - min[c]) / (max[c] - min[c]) - 1), optionally clipped -- the same float32 operations in the same
// order as the numpy expression of rbc_normalize_observation.py:66-74 (no contraction), so results are bit-identical to the host wrapper.
struct ObsNorm3 {
    int n, clip;                // channels normalised (0: off), clip to [-maxval, maxval]
    float mn[4], rng[4], maxval;
};
__device__ __forceinline__ float obs_value3(const ObsNorm3 &P, int c, float o)
{
    if (c < P.n) {
        o = __fmul_rn(P.maxval, __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, __fsub_rn(o, P.mn[c])), P.rng[c]), 1.0f));
        if (P.clip && o == o) o = fminf(fmaxf(o, -P.maxval), P.maxval);      // np.clip hands a NaN through; fminf / fmaxf would drop it
    }
    return o;
}

struct Out2D {
    float *obs, *state32;       // [B][5][obs_nz][obs_nx], [B][5][nz][nx]
    double *nusselt;            // [B][2]
    int *flags;
    int obs_nx, obs_nz, write_state, obs_norm, obs_clip;
    float obs_min[5], obs_rng[5], obs_maxval;
};

__device__ __forceinline__ float obs_value2(const Out2D &P, int c, double x)
{
    float o = (float)x;
    if (c < P.obs_norm) {      // RBCNormalizeObservation, same float32 operations as rbc::obs_value
        o = __fmul_rn(P.obs_maxval, __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, __fsub_rn(o, P.obs_min[c])), P.obs_rng[c]), 1.0f));
        if (P.obs_clip && o == o) o = fminf(fmaxf(o, -P.obs_maxval), P.obs_maxval);      // (NaN goes through, as in np.clip)
    }
    return o;
}

}  // namespace rbc3c

#define RBC3_NS rbc3
#define RBC3_REAL double
#include "rbc3d_kernels_body.hpp"
#undef RBC3_NS
#undef RBC3_REAL
#define RBC3_NS rbc3f
#define RBC3_REAL float
#include "rbc3d_kernels_body.hpp"
#undef RBC3_NS
#undef RBC3_REAL
