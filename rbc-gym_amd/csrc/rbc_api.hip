// rbc_api.hip -- host side of librbc_hip.so: the C ABI declared in include/rbc_hip.h.
//
// Replaces the reference's Julia plugin API (src/rbc_gym/sim/rbc_sim2D_api.jl) for a BATCH of
// envs resident on one MI355X.  All PDE work happens in rbc2d_kernel (rbc2d_kernel.hpp); this
// file only owns device buffers, the per-env clocks (api:12-13,67-68,87-88) and the launches.
// There is deliberately no CPU fallback: every entry point that needs the GPU fails with
// RBC_ERR_DEVICE when HIP does.
#include "rbc2d_instances.hpp"

#include "../../include/rbc_hip.h"

// the resident 2D kernels are defined in rbc2d_instances.hip (own translation unit, own scheduling strategy)
#define RBC2D_DECLARE(NX, NZ, T)                                                                    \
    extern template __global__ void rbc::rbc2d_kernel<NX, NZ, T, false>(const rbc::Params2D); \
    extern template __global__ void rbc::rbc2d_kernel<NX, NZ, T, true>(const rbc::Params2D);
#define RBC2D_DECLARE_PROD(NX, NZ, T) extern template __global__ void rbc::rbc2d_kernel<NX, NZ, T, false>(const rbc::Params2D);
RBC2D_INSTANCES_F64(RBC2D_DECLARE)
RBC2D_INSTANCES_F32(RBC2D_DECLARE)
RBC2D_INSTANCES_F32X2(RBC2D_DECLARE_PROD)
#undef RBC2D_DECLARE
#undef RBC2D_DECLARE_PROD

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

// -DRBC_EXPERIMENTS=1: A/B builds made by scripts/ that read the numerics- or launch-shape-changing RBC_EXPERIMENT_* environment
// variables (rbc3d_host_body.hpp).  The shipped library is built without it and contains none of them
// (tests/test_host_api.py::test_shipped_library_has_no_experiment_knobs).
#ifndef RBC_EXPERIMENTS
#define RBC_EXPERIMENTS 0
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(RBC_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

}  // namespace

struct rbc_handle {
    rbc_config cfg;
    void *s3 = nullptr;                // streaming path (rbc3d_host.hpp: host3::rbc3_state or host3f::rbc3_state): dim == 3, and dim == 2 grids without an LDS-resident kernel
    bool s3_f32 = false;               // ... in float32 (rbc_config.precision): the state behind s3 belongs to host3f / rbc3f
    bool stream2d = false;             // dim == 2 on the streaming path (ny = 1): 2D actions, resets and outputs around the 3D stage kernels
    int obs_norm = 0, obs_clip = 0;    // rbc_set_obs_normalization
    float obs_min[5] = {0, 0, 0, 0, 0}, obs_rng[5] = {1, 1, 1, 1, 1}, obs_maxval = 1.0f;
    bool no_pair = false;              // RBC_NO_PAIR=1: unpacked 3D Poisson path (one FFT per slab; A/B and odd nz)
    bool no_tile = false;              // RBC_NO_TILE=1: skip the LDS-tiled 3D tendency kernels (A/B and debug)
    bool no_march = false;             // RBC_NO_MARCH=1: use the cell-per-thread 3D tendency kernels (A/B and debug)
    bool no_fuse_z = false;            // RBC_NO_FUSE_Z=1: two-kernel z sweeps (A/B reference of k3_thomas_pair_fused)
    bool no_graph = true;              // RBC_USE_GRAPH=1 replays the 3D env-step as a captured HIP graph (measured: +1 %, so off by default)
    int B = 0, nx = 0, nz = 0;
    size_t ncell = 0, env_stride = 0, obs_sz = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    double *d_fields = nullptr, *d_ra = nullptr, *d_tri = nullptr, *d_nu = nullptr, *d_dbg = nullptr, *d_park = nullptr, *d_celld = nullptr;
    float *d_actions = nullptr, *d_obs = nullptr, *d_state = nullptr;
    uint8_t *d_mask = nullptr;
    uint64_t *d_seeds = nullptr;
    int *d_flags = nullptr;
    unsigned long long *d_stamps = nullptr;   // RBC_STAMPS diagnostic builds only
    std::vector<double> t;
    std::vector<int64_t> step;
    std::vector<uint8_t> inited;
    std::vector<double> stage;        // host staging for reset_from_arrays
    std::vector<uint8_t> fresh;       // RBC_CLOCK_RECORDED: envs whose next env-step is the first after their reset (staging of d_mask)
    void *out_stage = nullptr;         // page-locked staging of large float32 outputs bound for pageable caller memory (copy_channels)
    size_t out_stage_cap = 0;
    std::vector<hipEvent_t> out_ev;    // one per staged chunk
    std::vector<hipEvent_t> ev;       // profiling: (start, stop) pairs, one pair per timed launch
    size_t ev_used = 0;               // pairs recorded since the last rbc_profile_read
    bool profiling = false;
    int nsub = 0;
    double dt_last = 0.0, dt_solver_eff = 0.0;
    void (*kernel)(const rbc::Params2D) = nullptr;
    void (*kernel_dbg)(const rbc::Params2D) = nullptr;   // the instantiation with the MODE_TENDENCY hook (rbc_debug_tendencies); none for packed pairs
    size_t lds_bytes = 0;
    int threads = 0;
    int lanes = 1;                     // envs per workgroup of the 2D kernel (2 for the packed float32 variant)
};

#include "rbc3d_host.hpp"

namespace {

inline bool is3d(const rbc_handle *h) { return h->cfg.dim == 3; }
inline size_t actions_per_env(const rbc_handle *h) { return is3d(h) ? (size_t)h->cfg.heaters * h->cfg.heaters : (size_t)h->cfg.heaters; }

template <int NX, int NZ, typename T>
void bind_kernel(rbc_handle *h)
{
    h->kernel = rbc::rbc2d_kernel<NX, NZ, T>;
    if constexpr (rbc::LaneT<T>::N == 1) h->kernel_dbg = rbc::rbc2d_kernel<NX, NZ, T, true>;
    h->lds_bytes = rbc::Geo<NX, NZ, T>::LDS_BYTES;
    h->threads = rbc::Geo<NX, NZ, T>::NT;
    h->lanes = rbc::LaneT<T>::N;
}

// The LDS-resident 2D kernel is instantiated for these grids (x periodic, lanes along x: NX = 8 * {8, 12, 16, 24}; NZ a
// multiple of 16; NX * NZ <= 8192 threads-times-8; three fields of NX * NZ reals + scratch within the CU's 160 KiB).
// float64 is the reference's arithmetic; float32 (RBC_PRECISION_F32) halves the footprint, so two workgroups share a CU
// and the larger grids fit.
bool bind_grid(rbc_handle *h, int nx, int nz, int precision)
{
#define RBC_GRID(NX_, NZ_, T_) if (nx == NX_ && nz == NZ_) { bind_kernel<NX_, NZ_, T_>(h); return true; }
    if (precision == RBC_PRECISION_F64) {
        RBC2D_INSTANCES_F64(RBC_GRID)
    } else if (precision == RBC_PRECISION_F32) {
        // packed pairs (two envs per workgroup, v_pk_*_f32: the f64 kernel's instruction stream at two envs per instruction)
        // where a pair fits the LDS; RBC_F32_SCALAR=1 selects the one-env-per-workgroup float kernel instead (A/B reference)
        const char *e = std::getenv("RBC_F32_SCALAR");
        if (!(e && e[0] == '1')) { RBC2D_INSTANCES_F32X2(RBC_GRID) }
        RBC2D_INSTANCES_F32(RBC_GRID)
    }
#undef RBC_GRID
    return false;
}

// Pivots of the z-direction operator of every Fourier mode (pressure solve):
//   (phi[k-1] - 2 phi[k] + phi[k+1])/dz^2 - lam_x(m) phi[k] = r[k], mirror (Neumann) ends.
// The kernel eliminates from both walls at once (the operator is mirror symmetric), so only the
// pivots of rows 0..nz/2-1 are needed: tab[k][m] = 1/(piv_k * nx), where nx undoes the
// unnormalised FFT pair.  Row nz/2 holds the junction factor 1/(1 - c^2), c = o/piv_{nz/2-1}
// (0 for the singular mean mode m = 0, which the kernel pins instead).
std::vector<double> tri_table(int nx, int nz, double lx, double lz)
{
    const int nh = nx / 2 + 1, half = nz / 2;
    const double dx = lx / nx, dz = lz / nz, o = 1.0 / (dz * dz), pi = 3.14159265358979323846;
    std::vector<double> tab((size_t)(half + 1) * nh);
    for (int m = 0; m < nh; ++m) {
        const double s = 2.0 * std::sin(m * pi / nx) / dx, lam = s * s;   // poisson_eigenvalues, Periodic
        double piv = 0.0;
        for (int k = 0; k < half; ++k) {
            const double d = -((k == 0) ? 1.0 : 2.0) * o - lam;
            piv = (k == 0) ? d : d - o * o / piv;
            tab[(size_t)k * nh + m] = 1.0 / (piv * nx);
        }
        const double c = o / piv;
        tab[(size_t)half * nh + m] = (m == 0) ? 0.0 : 1.0 / (1.0 - c * c);
    }
    return tab;
}

rbc::Params2D base_params(const rbc_handle *h)
{
    rbc::Params2D p{};
    p.fields = h->d_fields;
    p.actions = nullptr;
    p.nu_kappa = h->d_ra;
    p.mask = nullptr;
    p.seeds = h->d_seeds;
    p.tri_inv = h->d_tri;
    p.obs = h->d_obs;
    p.state32 = h->d_state;
    p.nusselt = h->d_nu;
    p.flags = h->d_flags;
    p.dbg_g = h->d_dbg;
    p.gpark = h->d_park;
    p.stamps = h->d_stamps;
    p.lx = h->cfg.lx; p.lz = h->cfg.lz;
    p.dx = h->cfg.lx / h->nx; p.dz = h->cfg.lz / h->nz;
    p.rdx = 1.0 / p.dx; p.rdz = 1.0 / p.dz; p.rdx2 = p.rdx * p.rdx; p.rdz2 = p.rdz * p.rdz;
    p.rhz = 1.0 / (p.dz / 2);
    p.min_b = h->cfg.min_b; p.delta_b = h->cfg.delta_b;
    p.heater_limit = h->cfg.heater_limit; p.kick = h->cfg.random_kick;
    p.dt = h->cfg.dt_solver; p.dt_last = h->dt_last; p.nsub = h->nsub;
    // float copies for the packed float32 kernel (scalar kernel arguments instead of in-kernel conversions, rbc2d_kernel.hpp Params2D)
    p.f_dx = (float)p.dx; p.f_dz = (float)p.dz; p.f_rdx = (float)p.rdx; p.f_rdz = (float)p.rdz; p.f_rdx2 = (float)p.rdx2; p.f_rdz2 = (float)p.rdz2;
    p.f_rhz = (float)p.rhz; p.f_min_b = (float)p.min_b; p.f_dt = (float)p.dt; p.f_dt_last = (float)p.dt_last;
    p.f_cpf = (p.f_rdz * p.f_rdz) * (float)h->nx;
    p.batch = h->B;
    p.heaters = h->cfg.heaters;
    p.mode = rbc::MODE_STEP;
    p.write_state = h->cfg.write_state;
    p.obs_nx = h->cfg.obs_nx; p.obs_nz = h->cfg.obs_nz;
    p.obs_norm = h->obs_norm; p.obs_clip = h->obs_clip; p.obs_maxval = h->obs_maxval;
    for (int c = 0; c < 5; ++c) { p.obs_min[c] = h->obs_min[c]; p.obs_rng[c] = h->obs_rng[c]; }
    return p;
}

int launch(rbc_handle *h, const rbc::Params2D &p, bool timed)
{
    void (*const kernel)(const rbc::Params2D) = (p.mode == rbc::MODE_TENDENCY) ? h->kernel_dbg : h->kernel;
    if (!kernel) return fail(RBC_ERR_INVALID, "no kernel for this mode");
    const bool rec = timed && h->profiling && 2 * (h->ev_used + 1) <= h->ev.size();
    if (rec) HIP_TRY(hipEventRecord(h->ev[2 * h->ev_used], h->stream));
    hipLaunchKernelGGL(kernel, dim3((h->B + h->lanes - 1) / h->lanes), dim3(h->threads), h->lds_bytes, h->stream, p);
    HIP_TRY(hipGetLastError());
    if (rec) {
        HIP_TRY(hipEventRecord(h->ev[2 * h->ev_used + 1], h->stream));
        h->ev_used++;
    }
    return RBC_OK;
}

int check_handle(const rbc_handle *h)
{
    if (!h) return fail(RBC_ERR_INVALID, "null handle");
    return RBC_OK;
}

int all_initialized(const rbc_handle *h)
{
    for (int e = 0; e < h->B; ++e)
        if (!h->inited[e])
            return fail(RBC_ERR_NOT_INITIALIZED, "Simulation not initialized. Call rbc_reset first.");   // api:79-81
    return RBC_OK;
}

}  // namespace

extern "C" {

int rbc_abi_version(void) { return RBC_ABI_VERSION; }

const char *rbc_last_error(void) { return g_err.c_str(); }

int rbc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rbc_has_precision(int precision)
{
    return (precision == RBC_PRECISION_F64 || precision == RBC_PRECISION_F32) ? 1 : 0;
}

void rbc_default_config(rbc_config *c)
{
    // gym registry defaults, src/rbc_gym/__init__.py:7-18, + constants rbc_sim2D_api.jl:28-38
    std::memset(c, 0, sizeof(*c));
    c->abi_version = RBC_ABI_VERSION;
    c->dim = 2;
    c->nx = 96; c->ny = 1; c->nz = 64;
    c->lx = 2.0 * 3.14159265358979323846; c->ly = 1.0; c->lz = 2.0;
    c->ra = 1e4; c->pr = 0.7;
    c->min_b = 1.0; c->delta_b = 1.0;
    c->heaters = 12; c->heater_limit = 0.75;
    c->dt_solver = 0.03; c->dt_control = 1.5;
    c->random_kick = 0.01;
    c->obs_nx = 48; c->obs_nz = 8;
    c->batch = 1; c->device = 0; c->write_state = 1;
    c->precision = RBC_PRECISION_F64;
    c->reference_clock = RBC_CLOCK_DOCUMENTED;
}

int rbc_create(const rbc_config *cfg, rbc_handle **out)
{
    if (!cfg || !out) return fail(RBC_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->abi_version != RBC_ABI_VERSION) return fail(RBC_ERR_INVALID, "rbc_config.abi_version mismatch");
    if (cfg->dim != 2 && cfg->dim != 3) return fail(RBC_ERR_INVALID, "dim must be 2 or 3");
    if (cfg->batch < 1) return fail(RBC_ERR_INVALID, "batch must be >= 1");
    if (!rbc_has_precision(cfg->precision)) return fail(RBC_ERR_INVALID, "precision must be RBC_PRECISION_F64 or RBC_PRECISION_F32");
    if (cfg->reference_clock != RBC_CLOCK_DOCUMENTED && cfg->reference_clock != RBC_CLOCK_RECORDED)
        return fail(RBC_ERR_INVALID, "reference_clock must be RBC_CLOCK_DOCUMENTED or RBC_CLOCK_RECORDED");
    if (cfg->heaters < 1 || cfg->heaters > rbc::MAX_HEATERS) return fail(RBC_ERR_INVALID, "heaters out of range");
    if (!(cfg->ra > 0) || !(cfg->pr > 0) || !(cfg->dt_solver > 0) || !(cfg->dt_control > 0))
        return fail(RBC_ERR_INVALID, "ra, pr, dt_solver, dt_control must be positive");
    if (cfg->dim == 2 && (cfg->nx < 8 || cfg->nz < 8))
        return fail(RBC_ERR_INVALID, "unsupported 2D grid: at least 8 cells in x and z");
    if (cfg->dim == 2 && (cfg->obs_nx < 1 || cfg->obs_nz < 2 || cfg->nx % cfg->obs_nx || cfg->nz % cfg->obs_nz))
        return fail(RBC_ERR_INVALID, "sensor grid must divide the state grid (and have >= 2 rows)");
    if (cfg->dim == 3 && (cfg->nx < 8 || cfg->ny < 8 || cfg->nz < 8 || !(cfg->ly > 0)))
        return fail(RBC_ERR_INVALID, "3D grid must be at least 8 cells in every direction");

    auto *h = new rbc_handle();
    h->cfg = *cfg;
    h->B = cfg->batch; h->nx = cfg->nx; h->nz = cfg->nz;
    { const char *e = std::getenv("RBC_USE_GRAPH"); h->no_graph = !(e && e[0] == '1'); }
    { const char *e = std::getenv("RBC_NO_MARCH"); h->no_march = e && e[0] == '1'; }
    { const char *e = std::getenv("RBC_NO_TILE"); h->no_tile = e && e[0] == '1'; }
    { const char *e = std::getenv("RBC_NO_PAIR"); h->no_pair = e && e[0] == '1'; }
    { const char *e = std::getenv("RBC_NO_FUSE_Z"); h->no_fuse_z = e && e[0] == '1'; }
    // 2D: the LDS-resident kernel where it is built for the grid; any other float64 grid runs on the streaming kernels with
    // ny = 1 (RBC_FORCE_STREAM2D=1 sends every float64 2D handle there: the A/B partner of the resident kernel in the tests)
    bool force_stream = false;
    { const char *e = std::getenv("RBC_FORCE_STREAM2D"); force_stream = e && e[0] == '1'; }
    if (cfg->dim == 3) { /* streaming kernels, any grid whose horizontal slab fits the LDS FFT */ }
    else if (force_stream || !bind_grid(h, cfg->nx, cfg->nz, cfg->precision)) {
        h->stream2d = true;                               // (float32: the rbc3f instantiation of the streaming kernels)
        h->cfg.ny = 1; h->cfg.ly = 1.0;
    }
    h->ncell = (size_t)h->nx * h->nz * (cfg->dim == 3 ? cfg->ny : 1);
    h->env_stride = (size_t)(3 * h->nz + 1) * h->nx;
    h->obs_sz = (size_t)cfg->obs_nx * cfg->obs_nz;
    {
        // 3D: the reference runs in free-fall units, solver/control steps are scaled by t_ff = lz^2 (rbc_sim3D_api.jl:43,65)
        const double tff = (cfg->dim == 3) ? cfg->lz * cfg->lz : 1.0;
        const double T = cfg->dt_control * tff, dt = cfg->dt_solver * tff;
        h->dt_solver_eff = dt;
        int nfull = (int)std::floor(T / dt + 1e-9);
        double rem = T - nfull * dt;
        if (rem > 1e-9 * dt) { h->nsub = nfull + 1; h->dt_last = rem; }
        else { h->nsub = nfull; h->dt_last = dt; }
        if (h->nsub < 1) { delete h; return fail(RBC_ERR_INVALID, "dt_control shorter than one solver step"); }
        // RBC_CLOCK_RECORDED drops one FULL solver step from every env-step but the first after a reset
        if (cfg->reference_clock == RBC_CLOCK_RECORDED && (h->nsub < 2 || !(T - dt > 1e-9 * dt))) {
            delete h;
            return fail(RBC_ERR_INVALID, "reference_clock = RBC_CLOCK_RECORDED needs dt_control > dt_solver");
        }
    }

#define CREATE_TRY(expr)                                                                           \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            int rc_ = fail(RBC_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));     \
            rbc_destroy(h);                                                                        \
            return rc_;                                                                            \
        }                                                                                          \
    } while (0)

    int ndev = 0;
    CREATE_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) { delete h; return fail(RBC_ERR_DEVICE, "no such HIP device"); }
    CREATE_TRY(hipSetDevice(cfg->device));
    CREATE_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    const size_t B = h->B;
    const size_t nact = actions_per_env(h);
    CREATE_TRY(hipMalloc(&h->d_ra, B * 2 * sizeof(double)));   // (nu, kappa) per env
    CREATE_TRY(hipMalloc(&h->d_actions, B * nact * sizeof(float)));
    CREATE_TRY(hipMalloc(&h->d_mask, B));
    CREATE_TRY(hipMalloc(&h->d_seeds, B * sizeof(uint64_t)));
    CREATE_TRY(hipMalloc(&h->d_nu, B * 2 * sizeof(double)));
    CREATE_TRY(hipMalloc(&h->d_flags, B * sizeof(int)));
    CREATE_TRY(hipMemset(h->d_flags, 0, B * sizeof(int)));
    {
        std::vector<double> nk(2 * B);
        for (size_t e = 0; e < B; ++e) { nk[2 * e] = std::sqrt(cfg->pr / cfg->ra); nk[2 * e + 1] = 1.0 / std::sqrt(cfg->pr * cfg->ra); }
        CREATE_TRY(hipMemcpy(h->d_ra, nk.data(), nk.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (cfg->dim == 3 || h->stream2d) {
        if (h->stream2d) {
            CREATE_TRY(hipMalloc(&h->d_obs, B * 5 * h->obs_sz * sizeof(float)));
            CREATE_TRY(hipMalloc(&h->d_state, B * 5 * h->ncell * sizeof(float)));
            CREATE_TRY(hipMemset(h->d_state, 0, B * 5 * h->ncell * sizeof(float)));
        } else CREATE_TRY(hipMalloc(&h->d_state, B * 4 * h->ncell * sizeof(float)));
        h->s3_f32 = (cfg->precision == RBC_PRECISION_F32);
#if RBC_STAMPS
        CREATE_TRY(hipMalloc(&h->d_stamps, B * 64 * sizeof(unsigned long long)));
        CREATE_TRY(hipMemset(h->d_stamps, 0, B * 64 * sizeof(unsigned long long)));
#endif
        if (int rc = RBC_S3(h, create3d, h)) { rbc_destroy(h); return rc; }
        h->t.assign(B, 0.0);
        h->step.assign(B, 1);
        h->inited.assign(B, 0);
        *out = h;
        return RBC_OK;
    }
    CREATE_TRY(hipMalloc(&h->d_fields, B * h->env_stride * sizeof(double)));
    CREATE_TRY(hipMemset(h->d_fields, 0, B * h->env_stride * sizeof(double)));
    CREATE_TRY(hipMalloc(&h->d_obs, B * 5 * h->obs_sz * sizeof(float)));
    CREATE_TRY(hipMalloc(&h->d_state, B * 5 * h->ncell * sizeof(float)));
    CREATE_TRY(hipMalloc(&h->d_park, B * 2 * rbc::CZ * (size_t)h->threads * sizeof(double)));
#if RBC_STAMPS
    CREATE_TRY(hipMalloc(&h->d_stamps, B * 64 * sizeof(unsigned long long)));
    CREATE_TRY(hipMemset(h->d_stamps, 0, B * 64 * sizeof(unsigned long long)));
#endif
    {
        std::vector<double> tab = tri_table(h->nx, h->nz, cfg->lx, cfg->lz);
        CREATE_TRY(hipMalloc(&h->d_tri, tab.size() * sizeof(double)));
        CREATE_TRY(hipMemcpy(h->d_tri, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(h->kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)h->lds_bytes));
    if (h->kernel_dbg)
        CREATE_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(h->kernel_dbg), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)h->lds_bytes));
#undef CREATE_TRY
    h->t.assign(B, 0.0);
    h->step.assign(B, 1);
    h->inited.assign(B, 0);
    *out = h;
    return RBC_OK;
}

int rbc_destroy(rbc_handle *h)
{
    if (!h) return RBC_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->own_stream) (void)hipStreamSynchronize(h->own_stream);
    if (h->s3) RBC_S3(h, destroy3d, h);
    void *bufs[] = {h->d_fields, h->d_ra, h->d_tri, h->d_nu, h->d_dbg, h->d_park, h->d_celld, h->d_actions, h->d_obs, h->d_state,
                    h->d_mask, h->d_seeds, h->d_flags, h->d_stamps};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->out_ev) (void)hipEventDestroy(e);
    if (h->out_stage) (void)hipHostFree(h->out_stage);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return RBC_OK;
}

int rbc_set_stream(rbc_handle *h, void *hip_stream)
{
    if (int rc = check_handle(h)) return rc;
    // hipStreamLegacy is kept as the null stream internally (the same stream; every HIP entry point takes it, which is not
    // true of the (hipStream_t)1 alias: hipEventRecord / hipStreamWaitEvent fault on it in this runtime)
    if (hip_stream == reinterpret_cast<void *>(hipStreamLegacy)) h->stream = nullptr;
    else h->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : h->own_stream;
    return RBC_OK;
}

void *rbc_get_stream(rbc_handle *h)
{
    if (!h) return nullptr;
    return h->stream ? reinterpret_cast<void *>(h->stream) : reinterpret_cast<void *>(hipStreamLegacy);
}

int rbc_synchronize(rbc_handle *h)
{
    if (int rc = check_handle(h)) return rc;
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RBC_OK;
}

int rbc_set_rayleigh(rbc_handle *h, const double *ra)
{
    if (int rc = check_handle(h)) return rc;
    if (!ra) return fail(RBC_ERR_INVALID, "null ra");
    for (int e = 0; e < h->B; ++e)
        if (!(ra[e] > 0)) return fail(RBC_ERR_INVALID, "Rayleigh numbers must be positive");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    std::vector<double> nk(2 * (size_t)h->B);
    for (int e = 0; e < h->B; ++e) {   // rbc_sim2D_api.jl:40-41
        nk[2 * e] = std::sqrt(h->cfg.pr / ra[e]);
        nk[2 * e + 1] = 1.0 / std::sqrt(h->cfg.pr * ra[e]);
    }
    HIP_TRY(hipMemcpy(h->d_ra, nk.data(), nk.size() * sizeof(double), hipMemcpyHostToDevice));
    return RBC_OK;
}

static int upload_mask(rbc_handle *h, const uint8_t *mask, std::vector<uint8_t> &m)
{
    m.assign(h->B, 1);
    if (mask)
        for (int e = 0; e < h->B; ++e) m[e] = mask[e] ? 1 : 0;
    HIP_TRY(hipMemcpyAsync(h->d_mask, m.data(), h->B, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RBC_OK;
}

static void mark_reset(rbc_handle *h, const std::vector<uint8_t> &m)
{
    for (int e = 0; e < h->B; ++e)
        if (m[e]) {
            h->t[e] = 0.0;      // api:68
            h->step[e] = 1;     // api:67
            h->inited[e] = 1;
        }
}

int rbc_reset(rbc_handle *h, const uint8_t *mask, const uint64_t *seeds)
{
    if (int rc = check_handle(h)) return rc;
    if (!seeds) return fail(RBC_ERR_INVALID, "null seeds");
    HIP_TRY(hipSetDevice(h->cfg.device));
    std::vector<uint8_t> m;
    if (int rc = upload_mask(h, mask, m)) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_seeds, seeds, (size_t)h->B * sizeof(uint64_t), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->s3) {
        if (int rc = RBC_S3(h, random_reset3d, h)) return rc;
        mark_reset(h, m);
        return RBC_OK;
    }
    rbc::Params2D p = base_params(h);
    p.mode = rbc::MODE_RANDOM;
    p.mask = h->d_mask;
    if (int rc = launch(h, p, false)) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    mark_reset(h, m);
    return RBC_OK;
}

int rbc_reset_from_arrays3(rbc_handle *h, const uint8_t *mask, const double *b, const double *u, const double *v, const double *w)
{
    if (int rc = check_handle(h)) return rc;
    if (!is3d(h)) return fail(RBC_ERR_INVALID, "rbc_reset_from_arrays3 needs a dim=3 handle");
    if (!b || !u || !v || !w) return fail(RBC_ERR_INVALID, "null field array");
    HIP_TRY(hipSetDevice(h->cfg.device));
    std::vector<uint8_t> m;
    if (int rc = upload_mask(h, mask, m)) return rc;
    const size_t nc = h->ncell, nw = RBC_S3(h, faces3d, h);
    h->stage.resize(RBC_S3(h, env_stride3d, h));
    for (int e = 0; e < h->B; ++e) {
        if (!m[e]) continue;
        std::memcpy(h->stage.data(), b + (size_t)e * nc, nc * sizeof(double));
        std::memcpy(h->stage.data() + nc, u + (size_t)e * nc, nc * sizeof(double));
        std::memcpy(h->stage.data() + 2 * nc, v + (size_t)e * nc, nc * sizeof(double));
        std::memcpy(h->stage.data() + 3 * nc, w + (size_t)e * nw, nw * sizeof(double));
        if (int rc = RBC_S3(h, put_env3d, h, e, h->stage.data())) return rc;
    }
    if (int rc = RBC_S3(h, finish_reset3d, h)) return rc;
    mark_reset(h, m);
    return RBC_OK;
}

int rbc_reset_from_arrays(rbc_handle *h, const uint8_t *mask, const double *b, const double *u, const double *w)
{
    if (int rc = check_handle(h)) return rc;
    if (is3d(h)) return fail(RBC_ERR_INVALID, "dim=3 handles take rbc_reset_from_arrays3 (b,u,v,w)");
    if (!b || !u || !w) return fail(RBC_ERR_INVALID, "null field array");
    HIP_TRY(hipSetDevice(h->cfg.device));
    std::vector<uint8_t> m;
    if (int rc = upload_mask(h, mask, m)) return rc;
    const size_t nc = h->ncell, nw = nc + h->nx;
    if (h->stream2d) {                                  // streaming layout [b | u | v = 0 | w]
        h->stage.assign(RBC_S3(h, env_stride3d, h), 0.0);
        for (int e = 0; e < h->B; ++e) {
            if (!m[e]) continue;
            std::memcpy(h->stage.data(), b + (size_t)e * nc, nc * sizeof(double));
            std::memcpy(h->stage.data() + nc, u + (size_t)e * nc, nc * sizeof(double));
            std::memcpy(h->stage.data() + 3 * nc, w + (size_t)e * nw, nw * sizeof(double));
            if (int rc = RBC_S3(h, put_env3d, h, e, h->stage.data())) return rc;
        }
        if (int rc = RBC_S3(h, finish_reset3d, h)) return rc;
        mark_reset(h, m);
        return RBC_OK;
    }
    h->stage.resize(h->env_stride);
    for (int e = 0; e < h->B; ++e) {
        if (!m[e]) continue;
        std::memcpy(h->stage.data(), b + (size_t)e * nc, nc * sizeof(double));
        std::memcpy(h->stage.data() + nc, u + (size_t)e * nc, nc * sizeof(double));
        std::memcpy(h->stage.data() + 2 * nc, w + (size_t)e * nw, nw * sizeof(double));
        HIP_TRY(hipMemcpy(h->d_fields + (size_t)e * h->env_stride, h->stage.data(), h->env_stride * sizeof(double),
                          hipMemcpyHostToDevice));
    }
    rbc::Params2D p = base_params(h);
    p.mode = rbc::MODE_PROJECT;
    p.mask = h->d_mask;
    if (int rc = launch(h, p, false)) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    mark_reset(h, m);
    return RBC_OK;
}

static void advance_clocks(rbc_handle *h)
{
    for (int e = 0; e < h->B; ++e) {
        h->t[e] += h->cfg.dt_control * (is3d(h) ? h->cfg.lz * h->cfg.lz : 1.0);   // api:87 (3D: rbc_sim3D_api.jl:89)
        h->step[e] += 1;                // api:88
    }
}

// One env-step of the 2D resident kernel or of the streaming path: `nsub` solver steps for every env (the last of size dt_last).
static int step_uniform(rbc_handle *h, const float *actions_dev, int nsub, bool timed)
{
    if (h->s3) return RBC_S3(h, step3d, h, actions_dev, nsub, h->dt_solver_eff, h->dt_last, timed);
    rbc::Params2D p = base_params(h);
    p.actions = actions_dev;
    p.nsub = nsub;
    return launch(h, p, timed);
}

int rbc_step_dev(rbc_handle *h, const float *actions_dev)
{
    if (int rc = check_handle(h)) return rc;
    if (int rc = all_initialized(h)) return rc;
    if (!actions_dev) return fail(RBC_ERR_INVALID, "null actions");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->cfg.reference_clock == RBC_CLOCK_RECORDED) {
        // The reference's recorded series (rbc_config.reference_clock): the first env-step after a reset integrates all
        // nsub solver steps, every later one nsub - 1.  step[e] == 1 (api:67) marks an env that has not stepped since its reset.
        int nfresh = 0;
        h->fresh.resize(h->B);
        for (int e = 0; e < h->B; ++e) { h->fresh[e] = (h->step[e] == 1); nfresh += h->fresh[e]; }
        if (nfresh == h->B) { if (int rc = step_uniform(h, actions_dev, h->nsub, true)) return rc; }
        else if (nfresh == 0) { if (int rc = step_uniform(h, actions_dev, h->nsub - 1, true)) return rc; }
        else {
            // mixed batch (a masked reset put some envs at their first env-step): the fresh envs take their extra solver step
            // alone -- the same arithmetic as the first substep of a full interval, nothing is carried from one RK3 substep
            // into the next (zeta^1 = 0) --, then every env runs the nsub - 1 steps of a later interval
            HIP_TRY(hipMemcpyAsync(h->d_mask, h->fresh.data(), (size_t)h->B, hipMemcpyHostToDevice, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));      // `fresh` is pageable and rewritten by the next call; mixed batches are rare
            if (h->s3) { if (int rc = RBC_S3(h, lead_substep3d, h, actions_dev, h->d_mask)) return rc; }
            else {
                rbc::Params2D p = base_params(h);
                p.actions = actions_dev;
                p.nsub = 1; p.dt_last = h->cfg.dt_solver; p.f_dt_last = (float)p.dt_last;
                p.mask = h->d_mask;
                if (int rc = launch(h, p, false)) return rc;
            }
            if (int rc = step_uniform(h, actions_dev, h->nsub - 1, true)) return rc;
        }
        advance_clocks(h);
        return RBC_OK;
    }
    if (int rc = step_uniform(h, actions_dev, h->nsub, true)) return rc;
    advance_clocks(h);
    return RBC_OK;
}

int rbc_step(rbc_handle *h, const float *actions)
{
    if (int rc = check_handle(h)) return rc;
    if (int rc = all_initialized(h)) return rc;
    if (!actions) return fail(RBC_ERR_INVALID, "null actions");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const size_t nact = actions_per_env(h);
    HIP_TRY(hipMemcpyAsync(h->d_actions, actions, (size_t)h->B * nact * sizeof(float), hipMemcpyHostToDevice,
                           h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));   // `actions` may be pageable: do not return before it is consumed
    if (int rc = rbc_step_dev(h, h->d_actions)) return rc;
    std::vector<int32_t> fl(h->B);
    HIP_TRY(hipMemcpyAsync(fl.data(), h->d_flags, (size_t)h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int e = 0; e < h->B; ++e)
        if (fl[e]) return fail(RBC_ERR_NAN, "Error in simulation step, probably NaN values");   // rbc2D.py:171
    return RBC_OK;
}

static int copy_channels(rbc_handle *h, float *out, const float *dev, size_t chan, int nch)
{
    const int total = is3d(h) ? 4 : 5;       // 3D: b,u,v,w (rbc_sim3D_api.jl:106-121); 2D: b,u,w,pHY',pNHS
    if (!out) return fail(RBC_ERR_INVALID, "null output");
    if (nch < 1 || nch > total) return fail(RBC_ERR_INVALID, "nch out of range");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t row = (size_t)nch * chan * sizeof(float), bytes = row * h->B;
    // Large outputs into PAGEABLE caller memory -- the fresh numpy array of every step that the reference's API returns (rbc2D.py:185-196):
    // 75 MB of float32 states at B = 1024 in 2D, 38 MB of observations at configs[4] in 3D.  hipMemcpy stages such a copy through
    // its own bounce buffers at ~11 GB/s, most of it the first touch of freshly mapped pages by one thread.  Here the block crosses
    // PCIe in chunks into a page-locked buffer of the handle (~50 GB/s) and a few threads move each chunk on as it lands, touching
    // the destination's pages in parallel.  Page-locked destinations (rbc_host_alloc, obs_buffers="pinned") keep the direct copy.
    static const bool staged_off = [] { const char *e = std::getenv("RBC_STAGED_COPY"); return e && e[0] == '0'; }();
    bool pageable = false;
    if (bytes >= ((size_t)8 << 20) && !staged_off) {
        hipPointerAttribute_t at;
        const hipError_t e = hipPointerGetAttributes(&at, out);
        if (e != hipSuccess) { (void)hipGetLastError(); pageable = true; }          // not known to the runtime: ordinary host memory
        else pageable = (at.type == hipMemoryTypeUnregistered);
    }
    if (!pageable) {
        HIP_TRY(hipMemcpy2D(out, row, dev, total * chan * sizeof(float), row, h->B, hipMemcpyDeviceToHost));
        return RBC_OK;
    }
    if (h->out_stage_cap < bytes) {
        if (h->out_stage) { (void)hipHostFree(h->out_stage); h->out_stage = nullptr; h->out_stage_cap = 0; }
        HIP_TRY(hipHostMalloc(&h->out_stage, bytes, hipHostMallocDefault));
        h->out_stage_cap = bytes;
    }
    const unsigned hw = std::thread::hardware_concurrency();
    const int T = hw >= 8 ? 4 : (hw >= 4 ? 2 : 1), K = 2 * T;
    while ((int)h->out_ev.size() < K) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        h->out_ev.push_back(ev);
    }
    const int per = (h->B + K - 1) / K;
    char *stage = static_cast<char *>(h->out_stage);
    int chunks = 0;
    for (int c = 0; c < K && c * per < h->B; ++c, ++chunks) {
        const int r0 = c * per, nr = (r0 + per <= h->B) ? per : h->B - r0;
        HIP_TRY(hipMemcpy2DAsync(stage + (size_t)r0 * row, row, dev + (size_t)r0 * total * chan, total * chan * sizeof(float), row, nr,
                                 hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipEventRecord(h->out_ev[c], h->stream));
    }
    std::vector<hipError_t> res((size_t)T, hipSuccess);
    auto mover = [&](int t) {
        (void)hipSetDevice(h->cfg.device);
        for (int c = t; c < chunks; c += T) {
            const hipError_t e = hipEventSynchronize(h->out_ev[c]);
            if (e != hipSuccess) { res[(size_t)t] = e; return; }
            const int r0 = c * per, nr = (r0 + per <= h->B) ? per : h->B - r0;
            std::memcpy(reinterpret_cast<char *>(out) + (size_t)r0 * row, stage + (size_t)r0 * row, (size_t)nr * row);
        }
    };
    std::vector<std::thread> th;
    int started = 1;                                     // mover 0 is this thread
    try {
        for (int t = 1; t < T; ++t) { th.emplace_back(mover, t); ++started; }
    } catch (...) { /* no more threads to be had: this thread moves their chunks too (nothing may be thrown across the C ABI) */ }
    mover(0);
    for (int t = started; t < T; ++t) mover(t);
    for (auto &x : th) x.join();
    for (hipError_t e : res)
        if (e != hipSuccess) return fail(RBC_ERR_DEVICE, std::string("staged copy: ") + hipGetErrorString(e));
    return RBC_OK;
}

int rbc_set_obs_normalization(rbc_handle *h, const double *min_vals, const double *max_vals, int nch, double maxval, int clip)
{
    if (int rc = check_handle(h)) return rc;
    if (nch < 0 || nch > (is3d(h) ? 4 : 5)) return fail(RBC_ERR_INVALID, "rbc_set_obs_normalization: nch must be in [0, 5] (dim=3: [0, 4], the channels b, u, v, w)");
    if (nch > 0 && (!min_vals || !max_vals)) return fail(RBC_ERR_INVALID, "rbc_set_obs_normalization: NULL bounds");
    for (int c = 0; c < nch; ++c) {
        // the numpy expression divides by the python float (max - min) cast to float32
        const float rng = (float)(max_vals[c] - min_vals[c]);
        if (!(rng > 0.0f)) return fail(RBC_ERR_INVALID, "rbc_set_obs_normalization: max_vals must exceed min_vals");
    }
    for (int c = 0; c < nch; ++c) { h->obs_min[c] = (float)min_vals[c]; h->obs_rng[c] = (float)(max_vals[c] - min_vals[c]); }
    h->obs_norm = nch; h->obs_clip = clip ? 1 : 0; h->obs_maxval = (float)maxval;
    if (h->s3) RBC_S3(h, drop_graphs3d, h);          // a captured env-step carries the old parameters as kernel arguments
    if (is3d(h)) {                                   // the 3D observation IS the float32 state buffer: rewrite it for the current state
        bool any = false;
        for (uint8_t v : h->inited) any = any || v;
        if (any) {
            HIP_TRY(hipSetDevice(h->cfg.device));
            if (int rc = RBC_S3(h, refresh_outputs3d, h)) return rc;
        }
    }
    return RBC_OK;
}

int rbc_get_cell_distances(rbc_handle *h, double height, double *out)
{
    if (int rc = check_handle(h)) return rc;
    if (int rc = all_initialized(h)) return rc;
    if (is3d(h)) return fail(RBC_ERR_INVALID, "rbc_get_cell_distances: 2D envs only (the reference's wrapper reads a 2D mid-line)");
    if (!h->cfg.write_state) return fail(RBC_ERR_INVALID, "rbc_get_cell_distances needs write_state=1 (it reads the float32 state)");
    if (!out) return fail(RBC_ERR_INVALID, "null output");
    if (h->nx > 256) return fail(RBC_ERR_INVALID, "rbc_get_cell_distances: nx <= 256");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (!h->d_celld) HIP_TRY(hipMalloc(&h->d_celld, (size_t)h->B * sizeof(double)));
    const float *mid = h->d_state + 2 * h->ncell + (size_t)(h->nz / 2 - 1) * h->nx;     // channel UY = w, row int(nz/2) - 1
    hipLaunchKernelGGL(rbc::cell_distance_kernel, dim3(h->B), dim3(64), 0, h->stream, mid, 5 * h->ncell, h->nx, h->cfg.lx, (float)height, h->d_celld);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, h->d_celld, (size_t)h->B * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RBC_OK;
}

void *rbc_dev_cell_dist(rbc_handle *h) { return h ? h->d_celld : nullptr; }

int rbc_debug_cell_distances(int device, const float *uy, int B, int nx, double lx, double height, double *out)
{
    if (!uy || !out || B < 1 || nx < 3 || nx > 256) return fail(RBC_ERR_INVALID, "rbc_debug_cell_distances: bad argument (3 <= nx <= 256)");
    HIP_TRY(hipSetDevice(device));
    float *d_in = nullptr;
    double *d_out = nullptr;
    HIP_TRY(hipMalloc(&d_in, (size_t)B * nx * sizeof(float)));
    hipError_t e = hipMalloc(&d_out, (size_t)B * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(d_in, uy, (size_t)B * nx * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rbc::cell_distance_kernel, dim3(B), dim3(64), 0, nullptr, d_in, (size_t)nx, nx, lx, (float)height, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)B * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(RBC_ERR_DEVICE, std::string("rbc_debug_cell_distances: ") + hipGetErrorString(e));
    return RBC_OK;
}

void *rbc_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable)   /* portable: page-locked for every device of the node (multi-GPU vector env) */ != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}

void rbc_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int rbc_get_obs(rbc_handle *h, float *out, int nch)
{
    if (int rc = check_handle(h)) return rc;
    if (int rc = all_initialized(h)) return rc;
    if (is3d(h)) return copy_channels(h, out, h->d_state, h->ncell, nch);     // 3D: the observation IS the state (rbc3D.py:229-232)
    return copy_channels(h, out, h->d_obs, h->obs_sz, nch);
}

int rbc_get_state(rbc_handle *h, float *out, int nch)
{
    if (int rc = check_handle(h)) return rc;
    if (int rc = all_initialized(h)) return rc;
    if (!is3d(h) && !h->cfg.write_state) return fail(RBC_ERR_INVALID, "handle was created with write_state=0");
    return copy_channels(h, out, h->d_state, h->ncell, nch);
}

int rbc_get_fields3(rbc_handle *h, double *b, double *u, double *v, double *w)
{
    if (int rc = check_handle(h)) return rc;
    if (!is3d(h)) return fail(RBC_ERR_INVALID, "rbc_get_fields3 needs a dim=3 handle");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    double *outs[4] = {b, u, v, w};
    for (int f = 0; f < 4; ++f)
        if (outs[f])
            if (int rc = RBC_S3(h, get_field3d, h, f, outs[f])) return rc;
    return RBC_OK;
}

int rbc_get_fields(rbc_handle *h, double *b, double *u, double *w)
{
    if (int rc = check_handle(h)) return rc;
    if (is3d(h)) return fail(RBC_ERR_INVALID, "dim=3 handles take rbc_get_fields3");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->stream2d) {
        double *outs[4] = {b, u, nullptr, w};              // streaming layout [b | u | v = 0 | w]
        for (int f = 0; f < 4; ++f)
            if (outs[f])
                if (int rc = RBC_S3(h, get_field3d, h, f, outs[f])) return rc;
        return RBC_OK;
    }
    const size_t nc = h->ncell, nw = nc + h->nx, pitch = h->env_stride * sizeof(double);
    if (b) HIP_TRY(hipMemcpy2D(b, nc * sizeof(double), h->d_fields, pitch, nc * sizeof(double), h->B, hipMemcpyDeviceToHost));
    if (u) HIP_TRY(hipMemcpy2D(u, nc * sizeof(double), h->d_fields + nc, pitch, nc * sizeof(double), h->B, hipMemcpyDeviceToHost));
    if (w) HIP_TRY(hipMemcpy2D(w, nw * sizeof(double), h->d_fields + 2 * nc, pitch, nw * sizeof(double), h->B, hipMemcpyDeviceToHost));
    return RBC_OK;
}

int rbc_get_nusselt(rbc_handle *h, double *nu_state, double *nu_obs)
{
    if (int rc = check_handle(h)) return rc;
    if (int rc = all_initialized(h)) return rc;
    HIP_TRY(hipSetDevice(h->cfg.device));
    std::vector<double> tmp((size_t)h->B * 2);
    HIP_TRY(hipMemcpyAsync(tmp.data(), h->d_nu, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int e = 0; e < h->B; ++e) {
        if (is3d(h)) {                       // 3D has one Nusselt number (rbc_sim3D_api.jl:134): d_nu is [B]
            if (nu_state) nu_state[e] = tmp[e];
            if (nu_obs) nu_obs[e] = tmp[e];
            continue;
        }
        if (nu_state) nu_state[e] = tmp[2 * e];
        if (nu_obs) nu_obs[e] = tmp[2 * e + 1];
    }
    return RBC_OK;
}

int rbc_get_info(rbc_handle *h, double *t, int64_t *step)
{
    if (int rc = check_handle(h)) return rc;
    for (int e = 0; e < h->B; ++e) {
        if (t) t[e] = h->t[e];
        if (step) step[e] = h->step[e];
    }
    return RBC_OK;
}

int rbc_get_flags(rbc_handle *h, int32_t *flags)
{
    if (int rc = check_handle(h)) return rc;
    if (!flags) return fail(RBC_ERR_INVALID, "null flags");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipMemcpyAsync(flags, h->d_flags, (size_t)h->B * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RBC_OK;
}

void *rbc_dev_obs(rbc_handle *h) { return h ? h->d_obs : nullptr; }
void *rbc_dev_state(rbc_handle *h) { return h ? h->d_state : nullptr; }
void *rbc_dev_nusselt(rbc_handle *h) { return h ? h->d_nu : nullptr; }
void *rbc_dev_flags(rbc_handle *h) { return h ? h->d_flags : nullptr; }
void *rbc_dev_fields(rbc_handle *h) { return h ? (h->s3 ? RBC_S3(h, dev_fields3d, h) : (void *)h->d_fields) : nullptr; }

int rbc_set_profiling(rbc_handle *h, int max_launches)
{
    if (int rc = check_handle(h)) return rc;
    HIP_TRY(hipSetDevice(h->cfg.device));
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    h->ev.clear();
    h->ev_used = 0;
    h->profiling = max_launches > 0;
    for (int n = 0; n < 2 * max_launches; ++n) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        h->ev.push_back(e);
    }
    return RBC_OK;
}

int rbc_profile_read(rbc_handle *h, double *ms, int capacity)
{
    if (!h || !ms) return -1;
    if (hipSetDevice(h->cfg.device) != hipSuccess) return -1;
    int n = 0;
    for (size_t j = 0; j < h->ev_used && n < capacity; ++j, ++n) {
        float v = -1.0f;
        if (hipEventSynchronize(h->ev[2 * j + 1]) != hipSuccess) return -1;
        if (hipEventElapsedTime(&v, h->ev[2 * j], h->ev[2 * j + 1]) != hipSuccess) return -1;
        ms[n] = (double)v;
    }
    h->ev_used = 0;
    return n;
}

int rbc_copy_ceiling(int device, size_t bytes, int iters, double *kernel_gbs, double *memcpy_gbs)
{
    if (iters < 1 || bytes < (size_t)1 << 20) return fail(RBC_ERR_INVALID, "rbc_copy_ceiling: need iters >= 1 and at least 1 MiB");
    HIP_TRY(hipSetDevice(device));
    const size_t n16 = bytes / 16;
    void *src = nullptr, *dst = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = RBC_OK;
    auto bail = [&](hipError_t e, const char *what) { if (e != hipSuccess && rc == RBC_OK) rc = fail(RBC_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); };
    bail(hipMalloc(&src, n16 * 16), "hipMalloc");
    if (rc == RBC_OK) bail(hipMalloc(&dst, n16 * 16), "hipMalloc");
    if (rc == RBC_OK) bail(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    if (rc == RBC_OK) bail(hipEventCreate(&e0), "hipEventCreate");
    if (rc == RBC_OK) bail(hipEventCreate(&e1), "hipEventCreate");
    if (rc == RBC_OK) bail(hipMemsetAsync(src, 1, n16 * 16, st), "hipMemsetAsync");
    // which = 0: hipMemcpyAsync D2D; which >= 1: the copy kernel at several grid sizes (the best one is reported)
    const int grids[] = {0, 256 * 4, 256 * 8, 256 * 16, 256 * 32, 256 * 64};
    if (kernel_gbs) *kernel_gbs = 0.0;
    for (int which = 0; which < 6 && rc == RBC_OK; ++which) {
        if (which == 0 ? !memcpy_gbs : !kernel_gbs) continue;
        for (int it = -2; it < iters && rc == RBC_OK; ++it) {     // two warm-up rounds
            if (it == 0) bail(hipEventRecord(e0, st), "hipEventRecord");
            if (which == 0) bail(hipMemcpyAsync(dst, src, n16 * 16, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
            else {
                hipLaunchKernelGGL(rbc::copy16_kernel, dim3(grids[which]), dim3(256), 0, st, (const uint4 *)src, (uint4 *)dst, n16);
                bail(hipGetLastError(), "copy16_kernel");
            }
        }
        if (rc == RBC_OK) bail(hipEventRecord(e1, st), "hipEventRecord");
        if (rc == RBC_OK) bail(hipEventSynchronize(e1), "hipEventSynchronize");
        float ms = 0.0f;
        if (rc == RBC_OK) bail(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
        if (rc == RBC_OK) {
            const double gbs = 2.0 * (double)(n16 * 16) * iters / ((double)ms * 1e-3) / 1e9;
            if (which == 0) *memcpy_gbs = gbs;
            else if (gbs > *kernel_gbs) *kernel_gbs = gbs;
        }
    }
    if (st) (void)hipStreamSynchronize(st);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    return rc;
}

double rbc_algorithmic_bytes_per_env_step(rbc_handle *h)
{
    if (!h) return 0.0;
    // B_sub = 10 * F * C * s (F prognostic fields: 3 in 2D, 4 in 3D; C cells; s = 8 bytes, 4 for the float32 variant) per RK3 substep
    return (double)h->nsub * 10.0 * (is3d(h) ? 4.0 : 3.0) * (double)h->ncell * (h->cfg.precision == RBC_PRECISION_F32 ? 4.0 : 8.0);
}

int rbc_debug_tendencies3(rbc_handle *h, const float *actions, double *gu, double *gv, double *gw, double *gb)
{
    if (int rc = check_handle(h)) return rc;
    if (!is3d(h)) return fail(RBC_ERR_INVALID, "needs a dim=3 handle");
    if (!actions || !gu || !gv || !gw || !gb) return fail(RBC_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipMemcpy(h->d_actions, actions, (size_t)h->B * actions_per_env(h) * sizeof(float), hipMemcpyHostToDevice));
    double *const outs[4] = {gu, gv, gw, gb};
    return RBC_S3(h, debug_tendencies3d, h, outs);
}

int rbc_debug_tendencies(rbc_handle *h, const float *actions, double *gb, double *gu, double *gw)
{
    if (int rc = check_handle(h)) return rc;
    if (is3d(h)) return fail(RBC_ERR_INVALID, "dim=3 handles take rbc_debug_tendencies3");
    if (!actions || !gb || !gu || !gw) return fail(RBC_ERR_INVALID, "null argument");
    if (h->stream2d) {                     // the cell-per-thread tendency kernels on the ny = 1 state
        HIP_TRY(hipSetDevice(h->cfg.device));
        HIP_TRY(hipMemcpy(h->d_actions, actions, (size_t)h->B * actions_per_env(h) * sizeof(float), hipMemcpyHostToDevice));
        double *const outs[4] = {gu, nullptr, gw, gb};
        return RBC_S3(h, debug_tendencies3d, h, outs);
    }
    if (h->lanes != 1) return fail(RBC_ERR_INVALID, "rbc_debug_tendencies: not available on the packed float32 kernel (RBC_F32_SCALAR=1 selects the scalar one)");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const size_t nc = h->ncell;
    if (!h->d_dbg) HIP_TRY(hipMalloc(&h->d_dbg, (size_t)h->B * 3 * nc * sizeof(double)));
    HIP_TRY(hipMemcpy(h->d_actions, actions, (size_t)h->B * h->cfg.heaters * sizeof(float), hipMemcpyHostToDevice));
    rbc::Params2D p = base_params(h);
    p.mode = rbc::MODE_TENDENCY;
    p.actions = h->d_actions;
    p.nsub = 1;
    if (int rc = launch(h, p, false)) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    const size_t pitch = 3 * nc * sizeof(double);
    HIP_TRY(hipMemcpy2D(gb, nc * sizeof(double), h->d_dbg, pitch, nc * sizeof(double), h->B, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy2D(gu, nc * sizeof(double), h->d_dbg + nc, pitch, nc * sizeof(double), h->B, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy2D(gw, nc * sizeof(double), h->d_dbg + 2 * nc, pitch, nc * sizeof(double), h->B, hipMemcpyDeviceToHost));
    return RBC_OK;
}

/* diagnostic builds (-DRBC_STAMPS=1): per-phase shader-clock cycles of the last launch, [B][32] */
int rbc_debug_stamps(rbc_handle *h, unsigned long long *out)
{
    if (int rc = check_handle(h)) return rc;
    if (!h->d_stamps || !out) return fail(RBC_ERR_INVALID, "not a stamp build");
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(out, h->d_stamps, (size_t)h->B * 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RBC_OK;
}

int rbc_debug_launch_plan(rbc_handle *h, int groups[2])
{
    if (!h || !groups) return fail(RBC_ERR_INVALID, "null argument");
    groups[0] = 1; groups[1] = 0;
    if (h->s3) {
        if (h->s3_f32) { groups[0] = host3f::S3(h)->groups; groups[1] = host3f::S3(h)->own_queues ? 1 : 0; }
        else { groups[0] = host3::S3(h)->groups; groups[1] = host3::S3(h)->own_queues ? 1 : 0; }
    }
    return RBC_OK;
}

int rbc_debug_substeps(rbc_handle *h, const float *actions, int nsub, double dt)
{
    if (int rc = check_handle(h)) return rc;
    if (!actions || nsub < 1 || !(dt > 0)) return fail(RBC_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->cfg.device));
    if (h->s3) {
        HIP_TRY(hipMemcpy(h->d_actions, actions, (size_t)h->B * actions_per_env(h) * sizeof(float), hipMemcpyHostToDevice));
        if (int rc = RBC_S3(h, step3d, h, h->d_actions, nsub, dt, dt, false)) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));
        return RBC_OK;
    }
    HIP_TRY(hipMemcpy(h->d_actions, actions, (size_t)h->B * h->cfg.heaters * sizeof(float), hipMemcpyHostToDevice));
    rbc::Params2D p = base_params(h);
    p.actions = h->d_actions;
    p.nsub = nsub;
    p.dt = dt; p.f_dt = (float)dt;
    p.dt_last = dt; p.f_dt_last = (float)dt;
    if (int rc = launch(h, p, false)) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return RBC_OK;
}

}  // extern "C"
