// rbc3d_host.hpp -- host side of the 3D path (included by rbc_api.hip after `struct rbc_handle`).
// Replaces rbc_sim3D_api.jl (initialize_simulation :17, step_simulation :77, get_state :106,
// get_info :126, get_nusselt :134) for a batch of envs; see rbc3d_kernels.hpp for the launch sequence.
// Included by rbc3d_host.hpp once per precision: RBC3_HOST = host3 with K3 = rbc3 (real = double), RBC3_HOST = host3f with
// K3 = rbc3f (real = float).  No include guard on purpose.
namespace RBC3_HOST {

namespace K3 = RBC3_NS;
namespace K3C = rbc3c;
using real = K3::real;
using real2 = K3::real2;

struct rbc3_state {
    K3::Geo3 g;
    K3::FftPlan plan;
    real *st[2] = {nullptr, nullptr};     // ping-pong state buffers [B][b|u|v|w]
    int cur = 0;
    real *phy2 = nullptr;              // streaming-2D: pHY' of the stage before the last (k2s_output's correction of pNHS)
    bool unsplit_phi = false;          // the potential in `phi` belongs to un-split tendencies (set by a step, cleared by a reset)
    real *gm = nullptr, *phy = nullptr, *phi = nullptr, *tab = nullptr, *dbg = nullptr;
    double *actT = nullptr;            // bottom-plate table (float64 in both precisions)
    real2 *jct = nullptr;              // junction values of the packed z solve, [env][mode]
    real2 *spec = nullptr;
    real2 *tw = nullptr;               // FFT twiddle table (FftPlan::tw)
    double *out_part = nullptr;        // k3_output: per-env partial sums of its OUT_SPLIT workgroups
    unsigned int *out_arrive = nullptr;
    size_t fft_lds = 0;
    int thr2d = 256;
    int rows2d = 0;                    // streaming-2D: rows (each packed with its mirror) per FFT workgroup, 0 = per-slab kernels
    size_t fft2d_lds = 0;
    int ip2d = 0;                      // streaming-2D: N1 of the in-place separate kernels (k2s_*_ip), spectrum in position order
    size_t ip2d_lds = 0;
    real *tab_perm = nullptr;        // the pivot table with its columns in position order, and the conjugate partner of every position
    int *partner = nullptr;
    int fuse2d = 0;                    // streaming-2D: N1 of the one-kernel projection (k2s_project_fused), 0 = the separate kernels
    size_t fuse2d_lds = 0;
    int fft_threads = 256;             // slab-FFT workgroup: one round of work items for the larger of nx, ny (8 items per line)
    int march = 0;                     // 3D: 2 = k3_ifft_march (inverse FFT + the whole correction), 0 = k3_ifft_pair + k3_correct_w
    double tff = 1.0;
    // Le-Moin RK3 ([OC] TimeSteppers/runge_kutta_3.jl).  Builds made with -DRBC_EXPERIMENTS=1 (never the shipped library) read
    // RBC_EXPERIMENT_RK3="g1,g2,g3,z2,z3" for the "does the flowstats pin discriminate the time integrator" experiment.
    double gam[3] = {8.0 / 15.0, 5.0 / 12.0, 3.0 / 4.0}, zet[3] = {0.0, -17.0 / 60.0, -5.0 / 12.0};
    // one captured HIP graph per ping-pong parity of the standard env-step (39 stages is odd, so the
    // starting buffer alternates): ~350 launches replayed as one graph launch
    // (second index: 0 = the handle's nsub solver steps, 1 = nsub - 1, the later env-steps of RBC_CLOCK_RECORDED)
    hipGraphExec_t gexec[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    real *lead_save = nullptr;         // RBC_CLOCK_RECORDED, mixed batches: the envs that sit out the fresh envs' extra solver step
    // env groups: the batch is cut into `groups` contiguous ranges and every range runs its own chain of stage kernels on its
    // own stream (envs are independent), so that one group's latency-bound phases (single-round FFT launches, kernel tails)
    // overlap another group's tendency kernels.  1 = the whole batch on the handle's stream.
    int groups = 1;
    // time slices: the batch is cut into `slices` contiguous ranges that run their WHOLE env-step one after the other (each
    // slice on its `groups` chains), so that the working set of what is in flight -- two state buffers, G^-, potential,
    // spectrum of the slice's envs -- stays inside the 256 MB Infinity Cache: a streaming copy whose working set fits runs at
    // 7.1 TB/s on this chip, 5.2 TB/s from HBM (scripts/mall_probe.py).  1 = the whole batch at once (the default: slicing
    // measured slower, see create3d).
    int slices = 1;
    std::vector<hipStream_t> gstream;
    std::vector<hipEvent_t> gdone;
    hipEvent_t gstart = nullptr;
    // One captured graph PER env group ([parity][length][group]), each a single-stream graph on its group's stream, where the group
    // streams sit on different hardware queues (own_queues, found out by create3d).  Replaying a single-stream graph costs the
    // host ~0.1 ms; the one graph over all groups costs ~4 us per node -- 2.3 ms (float64) / 2.8 ms (float32) for configs[4], the
    // last chain starting 1.5 ms after the first.  Back to back that hides under the previous step; a SYNCHRONOUS caller (gym
    // semantics: actions in, wait, observations out) pays it on every step: 5.7 -> 4.9 ms float64, 4.7 -> 3.3 ms float32
    // (scripts/sync_step_latency.py).  RBC_3D_GROUP_GRAPHS=0: the one graph (also the fallback when the queues are shared).
    std::vector<hipGraphExec_t> ggexec[2][2];
    bool own_queues = false;
};

// a contiguous range of envs and the stream its launches go to
// (Bn: the NOMINAL group size of the handle -- every kernel-choice heuristic reads it instead of B, so that a shorter remainder group runs
// the same instantiations as the others and an env's result does not depend on which group it sits in)
struct rbc3_grp { int e0, B; hipStream_t st; int Bn = 0; int nominal() const { return Bn > 0 ? Bn : B; } };

namespace {

void factor2(int n, int &n1, int &n2)
{
    if (n == 32 || n == 48 || n == 64 || n == 96 || n == 128 || n == 192 || n == 256) { n1 = n / 8; n2 = 8; return; }   // register-blocked fast path (N1 x 8; the y pass is instantiated up to 64)
    n1 = 1;
    for (int d = 1; d * d <= n; ++d)
        if (n % d == 0) n1 = d;
    n2 = n / n1;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is per-kernel PROCESS state: setting it to one handle's need would lower it under
// another live handle on a larger grid (whose next launch then fails with hipErrorInvalidValue).  It is a ceiling, not a
// reservation -- occupancy follows the size a launch actually asks for -- so every kernel gets the CU's whole LDS once.
constexpr int RBC_LDS_CEILING = 160 * 1024;
#define RBC_LDS_ATTR(fn) HIP3(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, RBC_LDS_CEILING))

#define HIP3(expr)                                                                                 \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(RBC_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// waves per SIMD the 768-thread tile kernels are compiled for: 3 in float64 (168 VGPRs, one 12-wave workgroup per CU); the float32
// instantiation halves every window and plane register
#ifndef RBC_F32_TILE_WAVES
#define RBC_F32_TILE_WAVES 3
#endif
constexpr int TW3 = std::is_same<real, float>::value ? RBC_F32_TILE_WAVES : 3;

inline rbc3_state *S3(const rbc_handle *h) { return static_cast<rbc3_state *>(h->s3); }

// true if a kernel on `b` can run while one is running on `a`, i.e. the two streams do not share a hardware queue: `a` is held
// for half a millisecond, an empty kernel goes to `b`; if `a` is already done when `b` has finished, `b` waited behind it.
int streams_run_side_by_side(hipStream_t a, hipStream_t b, hipEvent_t ea, hipEvent_t eb, bool *yes)
{
    hipLaunchKernelGGL(K3C::k3_hold, dim3(1), dim3(64), 0, a, 50000LL);
    HIP3(hipEventRecord(ea, a));
    hipLaunchKernelGGL(K3C::k3_hold, dim3(1), dim3(64), 0, b, 0LL);
    HIP3(hipEventRecord(eb, b));
    HIP3(hipEventSynchronize(eb));
    const hipError_t q = hipEventQuery(ea);
    if (q != hipSuccess && q != hipErrorNotReady) return fail(RBC_ERR_DEVICE, std::string("hipEventQuery: ") + hipGetErrorString(q));
    *yes = (q == hipErrorNotReady);
    HIP3(hipEventSynchronize(ea));
    return RBC_OK;
}

// n streams for the env groups.  The runtime spreads the streams of a process over four hardware queues by a rule of its own
// (the n-th stream of a fresh process lands on queue 1, 2, 3, 4, 4, 3, 2, 1, 4, 3, ...: scripts/probes/queue_map.hip), so four streams
// created in a row can sit on two queues -- and two chains on one queue run one after the other (configs[4]: 7.1 ms instead of
// 4.9).  With `probe`, streams are created one at a time and kept only if they run side by side with every stream already kept (at
// most 12 tries, ~0.6 ms each); *own_queues says whether that found n of them.  Otherwise the first n are taken as they come.
int group_streams(rbc_handle *h, int n, bool probe, std::vector<hipStream_t> &out, bool *own_queues)
{
    (void)h;
    out.clear();
    *own_queues = false;
    std::vector<hipStream_t> rejected;
    hipEvent_t ea = nullptr, eb = nullptr;
    if (probe) { HIP3(hipEventCreateWithFlags(&ea, hipEventDisableTiming)); HIP3(hipEventCreateWithFlags(&eb, hipEventDisableTiming)); }
    int rc = RBC_OK;
    for (int tries = 0; (int)out.size() < n && tries < (probe ? 12 : n) && rc == RBC_OK; ++tries) {
        hipStream_t c = nullptr;
        if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) { rc = fail(RBC_ERR_DEVICE, "hipStreamCreateWithFlags failed"); break; }
        bool keep = true;
        for (size_t k = 0; probe && keep && k < out.size() && rc == RBC_OK; ++k) rc = streams_run_side_by_side(out[k], c, ea, eb, &keep);
        (keep ? out : rejected).push_back(c);
    }
    if (ea) (void)hipEventDestroy(ea);
    if (eb) (void)hipEventDestroy(eb);
    *own_queues = probe && (int)out.size() == n && rc == RBC_OK;
    while ((int)out.size() < n && !rejected.empty()) { out.push_back(rejected.back()); rejected.pop_back(); }      // shared queues: still n streams
    for (hipStream_t c : rejected) (void)hipStreamDestroy(c);
    while ((int)out.size() < n && rc == RBC_OK) {
        hipStream_t c = nullptr;
        if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) rc = fail(RBC_ERR_DEVICE, "hipStreamCreateWithFlags failed");
        else out.push_back(c);
    }
    return rc;
}

int create3d(rbc_handle *h)
{
    const rbc_config &c = h->cfg;
    auto *s = new rbc3_state();
    h->s3 = s;
    K3::Geo3 &g = s->g;
    // streaming-2D mode (a dim = 2 handle on a grid the LDS-resident kernel is not built for): ny = 1, see rbc3d_kernels.hpp
    const int ny = h->stream2d ? 1 : c.ny;
    const double ly = h->stream2d ? 1.0 : c.ly;
    g.nx = c.nx; g.ny = ny; g.nz = c.nz;
    g.nc = c.nx * ny * c.nz; g.nw = c.nx * ny * (c.nz + 1);
    g.env_stride = (size_t)3 * g.nc + g.nw;
    g.lx = c.lx; g.ly = ly; g.lz = c.lz;
    g.dx = c.lx / c.nx; g.dy = ly / ny; g.dz = c.lz / c.nz;
    g.rdx = 1.0 / g.dx; g.rdy = 1.0 / g.dy; g.rdz = 1.0 / g.dz;
    g.min_b = c.min_b; g.delta_b = c.delta_b; g.heater_limit = c.heater_limit; g.kick = c.random_kick;
    g.heaters = c.heaters;
    g.wall_nx = h->stream2d ? c.nx : 0;
    s->tff = h->stream2d ? 1.0 : c.lz * c.lz;                // rbc_sim3D_api.jl:43
#if RBC_EXPERIMENTS           /* numerics-changing knob: only in builds made with -DRBC_EXPERIMENTS=1, never in the shipped library */
    if (const char *e = std::getenv("RBC_EXPERIMENT_RK3")) {
        double v[5];
        if (std::sscanf(e, "%lf,%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3], &v[4]) == 5) {
            s->gam[0] = v[0]; s->gam[1] = v[1]; s->gam[2] = v[2]; s->zet[1] = v[3]; s->zet[2] = v[4];
        } else return fail(RBC_ERR_INVALID, "RBC_EXPERIMENT_RK3 must be g1,g2,g3,z2,z3");
    }
#endif
    factor2(c.nx, s->plan.nx1, s->plan.nx2);
    factor2(ny, s->plan.ny1, s->plan.ny2);
    s->fft_lds = ((size_t)2 * K3::slab_row(c.nx) * ny + c.nx + ny) * sizeof(real2);
    { const int items = 8 * (c.nx > ny ? c.nx : ny); s->fft_threads = items >= 512 ? 512 : (items <= 256 ? 256 : (items + 63) / 64 * 64); }
    if (h->stream2d) s->fft_threads = c.nx >= 256 ? 256 : (c.nx + 63) / 64 * 64;      // a "slab" is one row: one work item per point
#if RBC_EXPERIMENTS
    if (const char *e = std::getenv("RBC_EXPERIMENT_FFT_THREADS")) {            // A/B knob of the slab-FFT workgroup size
        const int v = std::atoi(e);
        if (v < 64 || v > 1024 || v % 64) return fail(RBC_ERR_INVALID, "RBC_EXPERIMENT_FFT_THREADS must be a multiple of 64 in [64, 1024]");
        s->fft_threads = v;
    }
#endif
    if (h->stream2d && c.nz % 2 == 0 && !h->no_pair) {      // several row pairs per workgroup (k2s_rhs_fft_pair / k2s_ifft_pair)
        int R = 16;
#if RBC_EXPERIMENTS
        if (const char *e = std::getenv("RBC_EXPERIMENT_FFT_ROWS")) {
            R = std::atoi(e);
            if (R < 1 || R > 64 || (R & (R - 1))) return fail(RBC_ERR_INVALID, "RBC_EXPERIMENT_FFT_ROWS must be a power of two in [1, 64]");
        }
        if (const char *e = std::getenv("RBC_EXPERIMENT_FFT2D_THREADS")) s->thr2d = std::atoi(e);
#endif
        if (s->thr2d > 256 || s->thr2d < 64) s->thr2d = 256;          // the kernels' launch bound
        while (R > 1 && ((c.nz / 2) % R != 0 || (size_t)(2 * R * K3::slab_row(c.nx) + c.nx) * sizeof(real2) > 128 * 1024)) R /= 2;
        s->rows2d = R < 1 ? 1 : R;
        s->fft2d_lds = (size_t)(2 * s->rows2d * K3::slab_row(c.nx) + c.nx) * sizeof(real2);
        if (s->fft2d_lds > (size_t)RBC_LDS_CEILING)
            return fail(RBC_ERR_INVALID, "streaming 2D: a row pair of this nx does not fit the LDS FFT (nx <= ~3400)");
        RBC_LDS_ATTR(K3::k2s_rhs_fft_pair);
        RBC_LDS_ATTR(K3::k2s_ifft_pair);
        // the whole projection as one kernel where an env's packed spectrum fits the LDS and nx = 8 * {4, 6, 8, 12, 16, 24}
        const size_t need = ((size_t)(c.nz / 2) * K3::slab_row(c.nx) + c.nx) * sizeof(real2);
        const char *nf = std::getenv("RBC_NO_FUSE_PROJECT");
        const int n1 = s->plan.nx1;
        const bool fast_rows = s->plan.nx2 == 8 && (n1 == 4 || n1 == 6 || n1 == 8 || n1 == 12 || n1 == 16 || n1 == 24 || n1 == 32) && c.nx <= 256;
        if (fast_rows && need <= 150 * 1024 && !(nf && nf[0] == '1')) {
            s->fuse2d = n1; s->fuse2d_lds = need;
#define RBC_FUSE_ATTR(N1_) RBC_LDS_ATTR(K3::k2s_project_fused<N1_>);
            RBC_FUSE_ATTR(4) RBC_FUSE_ATTR(6) RBC_FUSE_ATTR(8) RBC_FUSE_ATTR(12) RBC_FUSE_ATTR(16) RBC_FUSE_ATTR(24) RBC_FUSE_ATTR(32)
#undef RBC_FUSE_ATTR
        } else if (fast_rows) {          // the spectrum of an env does not fit: separate kernels with the in-place row FFT
            int Rr = 16;
            while (Rr > 1 && (c.nz / 2) % Rr != 0) Rr /= 2;
            s->ip2d = n1; s->rows2d = Rr;
            s->ip2d_lds = ((size_t)Rr * K3::slab_row(c.nx) + c.nx) * sizeof(real2);
#define RBC_IP_ATTR(N1_)                                                                                                                 \
            RBC_LDS_ATTR(K3::k2s_rhs_fft_pair_ip<N1_>); \
            RBC_LDS_ATTR(K3::k2s_ifft_pair_ip<N1_>);
            RBC_IP_ATTR(4) RBC_IP_ATTR(6) RBC_IP_ATTR(8) RBC_IP_ATTR(12) RBC_IP_ATTR(16) RBC_IP_ATTR(24) RBC_IP_ATTR(32)
#undef RBC_IP_ATTR
        }
    }
    if (s->fft_lds > 160 * 1024) return fail(RBC_ERR_INVALID, "3D horizontal slab too large for the LDS FFT (nx*ny <= ~5000)");
    const size_t B = h->B;
    for (int q = 0; q < 2; ++q) {
        HIP3(hipMalloc(&s->st[q], B * g.env_stride * sizeof(real)));
        HIP3(hipMemset(s->st[q], 0, B * g.env_stride * sizeof(real)));
    }
    HIP3(hipMalloc(&s->gm, B * g.env_stride * sizeof(real)));
    HIP3(hipMemset(s->gm, 0, B * g.env_stride * sizeof(real)));
    HIP3(hipMalloc(&s->phy, B * (size_t)g.nc * sizeof(real)));
    HIP3(hipMalloc(&s->phi, B * (size_t)g.nc * sizeof(real)));
    if (h->stream2d) HIP3(hipMalloc(&s->phy2, B * (size_t)g.nc * sizeof(real)));
    HIP3(hipMalloc(&s->spec, B * (size_t)g.nc * sizeof(real2)));
    HIP3(hipMalloc(&s->jct, B * (size_t)g.nx * g.ny * sizeof(real2)));
    HIP3(hipMalloc(&s->tw, (size_t)(g.nx + g.ny) * sizeof(real2)));
    hipLaunchKernelGGL(K3::k3_twiddles, dim3((unsigned)((g.nx + g.ny + 127) / 128)), dim3(128), 0, h->stream, s->tw, g.nx, g.ny);
    HIP3(hipGetLastError());
    s->plan.tw = s->tw;
    HIP3(hipMalloc(&s->out_part, B * 2 * K3::OUT_SPLIT * sizeof(double)));
    HIP3(hipMalloc(&s->out_arrive, B * sizeof(unsigned int)));
    HIP3(hipMemset(s->out_arrive, 0, B * sizeof(unsigned int)));
    const size_t nwall = h->stream2d ? (size_t)c.nx : (size_t)c.heaters * c.heaters;     // bottom-plate table per env
    HIP3(hipMalloc(&s->actT, B * nwall * sizeof(double)));
    HIP3(hipMemset(s->actT, 0, B * nwall * sizeof(double)));
    {   // pivots of the z operator for every horizontal mode: tab[k][n][m] = 1/piv_k
        const double o = 1.0 / (g.dz * g.dz), pi = 3.14159265358979323846;
        std::vector<double> tab((size_t)g.nc);
        for (int n = 0; n < ny; ++n)
            for (int m = 0; m < c.nx; ++m) {
                const double tx = 2.0 * std::sin(m * pi / c.nx) / g.dx, ty = 2.0 * std::sin(n * pi / ny) / g.dy;
                const double lam = tx * tx + ty * ty;
                double piv = 0.0;
                for (int k = 0; k < c.nz; ++k) {
                    double d = -((k == 0 || k == c.nz - 1) ? 1.0 : 2.0) * o - lam;
                    if (m == 0 && n == 0 && k == c.nz - 1) d -= o;       // pin the singular mean mode
                    piv = (k == 0) ? d : d - o * o / piv;
                    tab[((size_t)k * ny + n) * c.nx + m] = 1.0 / piv;
                }
            }
        {
            std::vector<real> tr(tab.begin(), tab.end());          // pivots are computed in float64, stored in the solver's precision
            HIP3(hipMalloc(&s->tab, tr.size() * sizeof(real)));
            HIP3(hipMemcpy(s->tab, tr.data(), tr.size() * sizeof(real), hipMemcpyHostToDevice));
        }
        if (s->ip2d) {                   // position p = 8 k1 + k2 holds mode m = k1 + N1 k2 (rowfft_inplace)
            const int N1 = s->ip2d, nx = c.nx;
            std::vector<real> tp((size_t)c.nz * nx);
            std::vector<int> pa(nx);
            for (int p = 0; p < nx; ++p) {
                const int m = (p >> 3) + N1 * (p & 7), mc = (m == 0) ? 0 : nx - m;
                pa[p] = 8 * (mc % N1) + mc / N1;
                for (int k = 0; k < c.nz; ++k) tp[(size_t)k * nx + p] = (real)tab[(size_t)k * nx + m];
            }
            HIP3(hipMalloc(&s->tab_perm, tp.size() * sizeof(real)));
            HIP3(hipMemcpy(s->tab_perm, tp.data(), tp.size() * sizeof(real), hipMemcpyHostToDevice));
            HIP3(hipMalloc(&s->partner, pa.size() * sizeof(int)));
            HIP3(hipMemcpy(s->partner, pa.data(), pa.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    {   // RBC_3D_SLICES=n: experiment knob, default 1.  Measured (B = 1024 at 128 x 64, 650 MB in flight): 2 / 4 / 8 slices run at
        // 22.9k / 21.0k / 16.2k env-steps/s against 24.5k unsliced -- what the Infinity Cache returns (+37 % on a pure copy) is less
        // than what the four-times-smaller launches lose; configs[4] (B = 32, 304 MB): 4.4k sliced in two against 4.8k.
        const int min_envs = h->stream2d ? 64 : 8;
        int sl = 1;
        if (const char *e = std::getenv("RBC_3D_SLICES")) sl = std::atoi(e);
        if (sl > 64) sl = 64;
        while (sl > 1 && h->B / sl < min_envs) --sl;
        s->slices = sl < 1 ? 1 : sl;
    }
    {   // RBC_3D_GROUPS=n overrides the default: 4 groups of >= 4 envs for 3D handles; streaming-2D: 3 chains once every chain still
        // fills the chip (B >= 768) -- the bandwidth-bound tile kernel of one chain then runs under the latency-bound one-kernel
        // projection of another (128 x 64, B = 1024: 24.2k / 25.3k / 26.3k / 26.1k env-steps/s with 1 / 2 / 3 / 4 chains)
        int want = h->stream2d ? (h->B >= 768 ? 3 : 1) : 4;
        if (const char *e = std::getenv("RBC_3D_GROUPS")) want = std::atoi(e);
        if (want > 16) want = 16;
        const int per_slice = (h->B + s->slices - 1) / s->slices;
        while (want > 1 && per_slice / want < 4) --want;
        s->groups = want < 1 ? 1 : want;
        if (s->slices > 1 && !std::getenv("RBC_USE_GRAPH")) h->no_graph = false;     // many short launches: replay them as one captured graph
        if (s->groups > 1 && !std::getenv("RBC_USE_GRAPH")) h->no_graph = false;     // several chains: replay them as one captured graph
        if (s->groups > 1) {
            s->gdone.resize(s->groups);
            for (int q = 0; q < s->groups; ++q) HIP3(hipEventCreateWithFlags(&s->gdone[q], hipEventDisableTiming));
            HIP3(hipEventCreateWithFlags(&s->gstart, hipEventDisableTiming));
            const char *gg = std::getenv("RBC_3D_GROUP_GRAPHS");
            if (int rc = group_streams(h, s->groups, !h->no_graph && s->slices == 1 && !(gg && gg[0] == '0'), s->gstream, &s->own_queues)) return rc;
        }
    }
    // 3D, mirror-packed path, float64: the inverse FFT marches 2 adjacent slab pairs per workgroup and applies the vertical correction
    // itself (k3_ifft_march).  Measured at configs[4], three interleaved repeats on one box (scripts/ab_march.sh): float64 6.28k ->
    // 6.69k env-steps/s (+6.6 %; 1 pair per workgroup, i.e. every pair transformed twice: +5.3 %; 4 pairs: -3 %; 8: -21 % -- the
    // workgroups are latency-bound, a longer serial chain costs more than the saved transforms); with everything a pair's correction
    // reads prefetched before its transform (two waves per SIMD, 247 VGPRs): 6.31k -> 6.74k (+6.8 %; 1 pair +5.3 %, 4 pairs +0.4 %).
    // float32: 9.36k -> 9.10k (-2.8 %), with the prefetch 9.37k -> 9.31k (-0.6 %): keeps k3_ifft_pair + k3_correct_w.  float64 planes
    // beyond 6 columns per thread (64 x 64) keep the separate pass too (the prefetch does not fit 256 registers).
    // RBC_IFFT_MARCH=0 | 2 overrides (0: the A/B partner, bitwise the same state; also what the streaming-2D grids run).
    if (!h->stream2d && c.nz % 4 == 0 && !h->no_pair && (size_t)c.nx * ny <= (size_t)(std::is_same<real, double>::value ? 6 : 8) * s->fft_threads) {
        int m = std::is_same<real, double>::value ? 2 : 0;
        if (const char *e = std::getenv("RBC_IFFT_MARCH")) m = std::atoi(e);
        s->march = (m == 2) ? 2 : 0;
#if RBC_EXPERIMENTS
        if ((m == 1 || m == 4) && (c.nz / 2) % m == 0) s->march = m;
#endif
    }
    RBC_LDS_ATTR((K3::k3_ifft_march<2, 4>)); RBC_LDS_ATTR((K3::k3_ifft_march<2, 6>));
    if constexpr (!std::is_same<real, double>::value) RBC_LDS_ATTR((K3::k3_ifft_march<2, 8>));
#if RBC_EXPERIMENTS
    RBC_LDS_ATTR((K3::k3_ifft_march<1, 6>)); RBC_LDS_ATTR((K3::k3_ifft_march<4, 6>));
#endif
    RBC_LDS_ATTR(K3::k3_rhs_fft);
    RBC_LDS_ATTR(K3::k3_ifft);
    RBC_LDS_ATTR(K3::k3_rhs_fft_pair);
    RBC_LDS_ATTR(K3::k3_ifft_pair);
    return RBC_OK;
}

void drop_graphs3d(rbc_handle *h)
{
    if (!h->s3) return;
    (void)hipStreamSynchronize(h->stream);
    for (auto &gp : S3(h)->gexec)
        for (auto &g : gp)
            if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    for (auto &gp : S3(h)->ggexec)
        for (auto &gv : gp) {
            for (hipGraphExec_t g : gv) if (g) (void)hipGraphExecDestroy(g);
            gv.clear();
        }
}

void destroy3d(rbc_handle *h)
{
    rbc3_state *s = S3(h);
    if (!s) return;
    drop_graphs3d(h);
    for (hipStream_t q : s->gstream) if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
    for (hipEvent_t e : s->gdone) if (e) (void)hipEventDestroy(e);
    if (s->gstart) (void)hipEventDestroy(s->gstart);
    void *bufs[] = {s->st[0], s->st[1], s->lead_save, s->gm, s->phy, s->phy2, s->tab_perm, s->partner, s->phi, s->spec, s->jct, s->tw, s->actT, s->tab, s->dbg, s->out_part, s->out_arrive};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    delete s;
    h->s3 = nullptr;
}

inline dim3 grid_for(size_t n, int bs) { return dim3((unsigned)((n + bs - 1) / bs)); }

inline rbc3_grp whole_batch(const rbc_handle *h) { return rbc3_grp{0, h->B, h->stream}; }

// exact projection of state buffer `which` (0/1) of the group's envs with stage step dts (mask: device pointer [B] or null)
// (want_phi = false: the caller does not need the potential itself afterwards -- only the one-kernel streaming-2D projection can skip its store)
int project3d(rbc_handle *h, const rbc3_grp &q, int which, double dts, const uint8_t *mask, bool want_phi = true)
{
    rbc3_state *s = S3(h);
    const K3::Geo3 &g = s->g;
    const int B = q.B;
    const size_t pln = (size_t)g.nx * g.ny;
    real *buf = s->st[which] + (size_t)q.e0 * g.env_stride;
    real *phi = s->phi + (size_t)q.e0 * g.nc;
    const uint8_t *mk = mask ? mask + q.e0 : nullptr;
    if (s->fuse2d && g.nz % 2 == 0 && !h->no_pair) {
        const int fthr = 512;             // measured: 256 -> 512 threads +12 % (more loads in flight around the LDS phases); 768 / 1024: no better
#define RBC_FUSE_LAUNCH(N1_) if (s->fuse2d == N1_) hipLaunchKernelGGL(K3::k2s_project_fused<N1_>, dim3(B), dim3(fthr), s->fuse2d_lds, q.st, g, s->plan, buf, phi, s->tab, dts, mk, want_phi ? 1 : 0);
        RBC_FUSE_LAUNCH(4) RBC_FUSE_LAUNCH(6) RBC_FUSE_LAUNCH(8) RBC_FUSE_LAUNCH(12) RBC_FUSE_LAUNCH(16) RBC_FUSE_LAUNCH(24)
        if (s->fuse2d == 32) hipLaunchKernelGGL(K3::k2s_project_fused<32>, dim3(B), dim3(256), s->fuse2d_lds, q.st, g, s->plan, buf, phi, s->tab, dts, mk, want_phi ? 1 : 0);
#undef RBC_FUSE_LAUNCH
        HIP3(hipGetLastError());
        return RBC_OK;
    }
    if (g.nz % 2 == 0 && !h->no_pair) {      // mirror slabs packed as one complex transform, z solve on the packed spectrum
        real2 *spec = s->spec + (size_t)q.e0 * (g.nz / 2) * pln, *jct = s->jct + (size_t)q.e0 * pln;
        const dim3 gm_ = grid_for((size_t)B * pln, 128);
        const int thr2d = s->thr2d;
        const real *tabz = s->ip2d ? s->tab_perm : s->tab;            // z-sweep tables in the order of the spectrum
        const int *partner = s->ip2d ? s->partner : nullptr;
#define RBC_IP_RHS(N1_) if (s->ip2d == N1_) hipLaunchKernelGGL(K3::k2s_rhs_fft_pair_ip<N1_>, dim3(B * (g.nz / 2 / s->rows2d)), dim3(256), s->ip2d_lds, q.st, g, s->plan, buf, spec, dts, s->rows2d);
#define RBC_IP_INV(N1_) if (s->ip2d == N1_) hipLaunchKernelGGL(K3::k2s_ifft_pair_ip<N1_>, dim3(B * (g.nz / 2 / s->rows2d)), dim3(256), s->ip2d_lds, q.st, g, s->plan, spec, phi, buf, dts, mk, s->rows2d);
        if (s->ip2d) { RBC_IP_RHS(4) RBC_IP_RHS(6) RBC_IP_RHS(8) RBC_IP_RHS(12) RBC_IP_RHS(16) RBC_IP_RHS(24) RBC_IP_RHS(32) }
        else
        if (s->rows2d) hipLaunchKernelGGL(K3::k2s_rhs_fft_pair, dim3(B * (g.nz / 2 / s->rows2d)), dim3(thr2d), s->fft2d_lds, q.st, g, s->plan, buf, spec, dts, s->rows2d);
        else hipLaunchKernelGGL(K3::k3_rhs_fft_pair, dim3(B * (g.nz / 2)), dim3(s->fft_threads), s->fft_lds, q.st, g, s->plan, buf, spec, dts);
        if (g.nz == 32 && !h->no_fuse_z) hipLaunchKernelGGL(K3::k3_thomas_pair_fused<16>, gm_, dim3(128), 0, q.st, g, spec, tabz, B, partner);
        else if (g.nz == 16 && !h->no_fuse_z) hipLaunchKernelGGL(K3::k3_thomas_pair_fused<8>, gm_, dim3(128), 0, q.st, g, spec, tabz, B, partner);
        else {
            hipLaunchKernelGGL(K3::k3_thomas_pair_fwd, gm_, dim3(128), 0, q.st, g, spec, jct, tabz, B);
            hipLaunchKernelGGL(K3::k3_thomas_pair_bwd, gm_, dim3(128), 0, q.st, g, spec, jct, tabz, B, partner);
        }
        if (s->ip2d) { RBC_IP_INV(4) RBC_IP_INV(6) RBC_IP_INV(8) RBC_IP_INV(12) RBC_IP_INV(16) RBC_IP_INV(24) RBC_IP_INV(32) }
        else
        if (s->rows2d) hipLaunchKernelGGL(K3::k2s_ifft_pair, dim3(B * (g.nz / 2 / s->rows2d)), dim3(thr2d), s->fft2d_lds, q.st, g, s->plan, spec, phi, buf, dts, mk, s->rows2d);
        else if (s->march) {            // inverse FFT + the WHOLE correction, two slab pairs per workgroup: no k3_correct_w, no phi
            const dim3 gr(B * (g.nz / 2 / s->march));
            const int np = (int)((pln + s->fft_threads - 1) / s->fft_threads);        // columns per thread: 4, 6 or 8
#define RBC_MARCH(CH_, NP_) hipLaunchKernelGGL((K3::k3_ifft_march<CH_, NP_>), gr, dim3(s->fft_threads), s->fft_lds, q.st, g, s->plan, spec, buf, dts, mk)
#if RBC_EXPERIMENTS      /* chunk-length sweep (scripts/ab_march.sh with an -DRBC_EXPERIMENTS=1 library): 1 or 4 pairs per workgroup, configs[4]-sized planes only */
            if (s->march == 1 && np <= 6) RBC_MARCH(1, 6); else if (s->march == 4 && np <= 6) RBC_MARCH(4, 6); else
#endif
            if (np <= 4) RBC_MARCH(2, 4); else if (np <= 6) RBC_MARCH(2, 6);
            else if constexpr (!std::is_same<real, double>::value) RBC_MARCH(2, 8);      // (float64: 8 columns per thread do not fit the registers, create3d)
#undef RBC_MARCH
            HIP3(hipGetLastError());
            return RBC_OK;
        }
        else hipLaunchKernelGGL(K3::k3_ifft_pair, dim3(B * (g.nz / 2)), dim3(s->fft_threads), s->fft_lds, q.st, g, s->plan, spec, phi, buf, dts, mk);
#if RBC_EXPERIMENTS      /* timing bound only (WRONG numerics): what a stage costs without the separate vertical correction */
        static const bool skip_cw = [] { const char *e = std::getenv("RBC_EXPERIMENT_SKIP_CW"); return e && e[0] == '1'; }();
        if (!skip_cw)
#endif
        hipLaunchKernelGGL(K3::k3_correct_w, grid_for((size_t)B * (g.nc - pln), 256), dim3(256), 0, q.st, g, buf, phi, dts, B, mk);
#undef RBC_IP_RHS
#undef RBC_IP_INV
        HIP3(hipGetLastError());
        return RBC_OK;
    }
    real2 *spec = s->spec + (size_t)q.e0 * g.nz * pln;
    hipLaunchKernelGGL(K3::k3_rhs_fft, dim3(B * g.nz), dim3(s->fft_threads), s->fft_lds, q.st, g, s->plan, buf, spec, dts);
    hipLaunchKernelGGL(K3::k3_thomas, grid_for((size_t)B * pln, 128), dim3(128), 0, q.st, g, spec, s->tab, B);
    hipLaunchKernelGGL(K3::k3_ifft, dim3(B * g.nz), dim3(s->fft_threads), s->fft_lds, q.st, g, s->plan, spec, phi);
    hipLaunchKernelGGL(K3::k3_correct, grid_for((size_t)B * g.nc, 256), dim3(256), 0, q.st, g, buf, phi, dts, B, mk);
    HIP3(hipGetLastError());
    return RBC_OK;
}

// outputs of the group's envs from state buffer `which`
int output3d(rbc_handle *h, const rbc3_grp &q, int which, const uint8_t *mask)
{
    rbc3_state *s = S3(h);
    const K3::Geo3 &g = s->g;
    const real *st = s->st[which] + (size_t)q.e0 * g.env_stride;
    const uint8_t *mk = mask ? mask + q.e0 : nullptr;
    if (h->stream2d) {
        K3::Out2D o{};
        o.obs = h->d_obs + (size_t)q.e0 * 5 * h->obs_sz; o.state32 = h->d_state + (size_t)q.e0 * 5 * g.nc;
        o.nusselt = h->d_nu + (size_t)q.e0 * 2; o.flags = h->d_flags + q.e0;
        o.obs_nx = h->cfg.obs_nx; o.obs_nz = h->cfg.obs_nz; o.write_state = h->cfg.write_state;
        o.obs_norm = h->obs_norm; o.obs_clip = h->obs_clip; o.obs_maxval = h->obs_maxval;
        for (int c = 0; c < 5; ++c) { o.obs_min[c] = h->obs_min[c]; o.obs_rng[c] = h->obs_rng[c]; }
        const double gs = s->gam[2] + s->zet[2];
        hipLaunchKernelGGL(K3::k2s_output, dim3(q.B), dim3(256), (2 * (size_t)g.nz + 256) * sizeof(double), q.st, g, st, s->phi + (size_t)q.e0 * g.nc,
                           h->d_ra + (size_t)q.e0 * 2, o, mk, s->unsplit_phi ? s->phy + (size_t)q.e0 * g.nc : (const real *)nullptr,
                           s->phy2 + (size_t)q.e0 * g.nc, s->gam[2] / gs, s->zet[2] / gs);
        HIP3(hipGetLastError());
        return RBC_OK;
    }
    K3::ObsNorm3 nrm{};
    nrm.n = h->obs_norm > 4 ? 4 : h->obs_norm; nrm.clip = h->obs_clip; nrm.maxval = h->obs_maxval;
    for (int c = 0; c < 4; ++c) { nrm.mn[c] = h->obs_min[c]; nrm.rng[c] = h->obs_rng[c]; }
    hipLaunchKernelGGL(K3::k3_output, dim3(q.B * K3::OUT_SPLIT), dim3(256), 0, q.st, g, st, h->d_ra + (size_t)q.e0 * 2, h->d_state + (size_t)q.e0 * 4 * g.nc,
                       h->d_nu + q.e0, h->d_flags + q.e0, mk, s->out_part + (size_t)q.e0 * 2 * K3::OUT_SPLIT, s->out_arrive + q.e0, nrm);
    HIP3(hipGetLastError());
    return RBC_OK;
}

// bottom-plate table of the group's envs from the raw actions: preprocess_action (3D) / collate_actions_colin per column (streaming 2D)
void wall3d(rbc_handle *h, const rbc3_grp &q, const float *actions_dev, int zero)
{
    rbc3_state *s = S3(h);
    const K3::Geo3 &g = s->g;
    if (h->stream2d)
        hipLaunchKernelGGL(K3::k2s_wall, grid_for((size_t)q.B * g.nx, 128), dim3(128), 0, q.st, g, actions_dev ? actions_dev + (size_t)q.e0 * g.heaters : nullptr,
                           s->actT + (size_t)q.e0 * g.nx, zero, q.B);
    else
        hipLaunchKernelGGL(K3::k3_preprocess, dim3(q.B), dim3(64), 0, q.st, g, actions_dev ? actions_dev + (size_t)q.e0 * g.heaters * g.heaters : nullptr,
                           s->actT + (size_t)q.e0 * g.heaters * g.heaters, zero);
}

// the stage list of `nsub` substeps (the last of size dt_last) for one group of envs, starting from state buffer `which`;
// actions already on the device.  Returns the buffer that holds the state afterwards through *which_out.
// (st0, st1: the range of RK3 stages, counted 0 .. 3 nsub - 1 over the whole interval, this call issues -- `which` is the buffer the
// first of them reads; the default is the whole interval.  run_step3d issues the groups' chains stage by stage, see there.)
int advance3d(rbc_handle *h, const rbc3_grp &q, int which, const float *actions_dev, int nsub, double dt, double dt_last, int *which_out,
              int st0 = 0, int st1 = -1)
{
    rbc3_state *s = S3(h);
    const K3::Geo3 &g = s->g;
    const int B = q.B;
    if (st1 < 0) st1 = 3 * nsub;
    if (st0 == 0) wall3d(h, q, actions_dev, 0);
    const double *gam = s->gam, *zet = s->zet;
    const dim3 gc = grid_for((size_t)B * g.nc, 128), bc(128);
    const size_t eo = (size_t)q.e0 * g.env_stride;
    real *gm = s->gm + eo, *phy = s->phy + (size_t)q.e0 * g.nc;
    const double *actT = s->actT + (size_t)q.e0 * (g.wall_nx ? (size_t)g.wall_nx : (size_t)g.heaters * g.heaters);
    const double *ra = h->d_ra + (size_t)q.e0 * 2;
    auto tiles_fit = [&](int ty, int kt, int maxt) {
        const int thr = g.nx * ty;
        return !h->no_tile && g.nz % kt == 0 && g.ny % ty == 0 && thr <= maxt && thr % 64 == 0 && g.nx <= K3::NXP3 &&
               (ty + 6) * g.nx <= 2 * thr;
    };
    // LDS-tiled tendency kernels (planes staged once per level; first half of the grid: (u, v), second half: (w, b)).  Tile shapes
    // (rows of y x levels of z): tall tiles win (fewer chunk prologues: the z windows of a column are loaded once per chunk) as
    // long as the launch still has about a hundred workgroups (a group of 8 configs[4] envs: 96); small batches take
    // 16 x 4.  RBC_TILE_SHAPE=16x16|16x8|16x4|8x8 forces one (A/B runs).
    const char *tshape = std::getenv("RBC_TILE_SHAPE");
    auto want = [&](const char *name, bool dflt) { return tshape ? std::strcmp(tshape, name) == 0 : dflt; };
    const bool no_nxc = [] { const char *e = std::getenv("RBC_NO_CONST_GRID"); return e && e[0] == '1'; }();      // A/B: the generic instantiations
    int shape = 0;
    const int Bn = q.nominal();
    auto wgs = [&](int ty, int kt) { return 2 * Bn * (g.ny / ty) * (g.nz / kt); };      // workgroups of one (full-sized) group's launch
    // float32: the tall tiles only once a launch has ~two hundred workgroups.  With four chains in flight a 96-workgroup launch per chain
    // is 384 workgroups on 256 CUs -- a round and a half -- and the half-empty second round costs the float32 kernel (shorter levels,
    // more issue-bound) more than the extra chunk prologues of 16 x 8 tiles: configs[4] float32 9.37k -> 9.94k env-steps/s (+6 %;
    // float64 prefers the tall tiles: 6.66k against 6.40k).  B = 64: 16 x 16 again (192 workgroups per launch), 10.8k.
    const int tall_min = std::is_same<real, float>::value ? 192 : 96;
    if (want("16x16", wgs(16, 16) >= tall_min) && tiles_fit(16, 16, 768)) shape = 1;
    else if (want("16x8", wgs(16, 8) >= 96) && tiles_fit(16, 8, 768)) shape = 2;
    else if (want("16x4", true) && tiles_fit(16, 4, 768)) shape = 3;
    else if (want("8x8", true) && tiles_fit(8, 8, 512)) shape = 4;
    // streaming-2D mode: the same bodies with one-row planes (FLAT), a workgroup = one row of nx threads marching 16 / 8 / 4 levels
    // (tall chunks while the launch keeps a few hundred workgroups; a small batch takes 4 levels: its step is a chain of short kernels)
    if (h->stream2d && !h->no_tile && g.nx <= 256 && g.nz % 4 == 0) {
        auto enough = [&](int kt) { return g.nz % kt == 0 && 2 * (size_t)Bn * (g.nz / kt) >= 256; };
        shape = (g.nz >= 64 && enough(32)) ? 8 : (enough(16) ? 5 : (enough(8) ? 6 : 7));     // 32 levels: +1.8 % at 128 x 64 (64: -6 %)
        if (const char *e = std::getenv("RBC_FLAT_KT")) {                                      // A/B knob: 4, 8, 16, 32, 64
            const int kt = std::atoi(e);
            if (kt > 0 && g.nz % kt == 0) shape = kt == 64 ? 9 : (kt == 32 ? 8 : (kt == 16 ? 5 : (kt == 8 ? 6 : (kt == 4 ? 7 : shape))));
        }
    }
    for (int n = st0 / 3; n < nsub && 3 * n < st1; ++n) {
        const double d = (n == nsub - 1) ? dt_last : dt;
        for (int ph = (n == st0 / 3) ? st0 % 3 : 0; ph < 3 && 3 * n + ph < st1; ++ph) {
            real *cur = s->st[which] + eo, *nxt = s->st[which ^ 1] + eo;
            const int store_g = (ph != 2);                     // the last stage's tendencies are never read again (zeta^1 = 0)
            if (!shape)                                        // the fallback kernels use the hydrostatic split (pHY' column scan)
                hipLaunchKernelGGL(K3::k3_hydrostatic, grid_for((size_t)B * g.nx * g.ny, 128), dim3(128), 0, q.st, g, cur, phy, B);
            else if (h->stream2d && n == nsub - 1 && ph >= 1)  // un-split tendencies: the two scans k2s_output needs to return pNHS
                hipLaunchKernelGGL(K3::k3_hydrostatic, grid_for((size_t)B * g.nx * g.ny, 128), dim3(128), 0, q.st, g, cur,
                                   ph == 2 ? phy : s->phy2 + (size_t)q.e0 * g.nc, B);
#undef RBC_TILE_STAMPS
#if RBC_STAMPS            /* diagnostic build: the last argument carries the stamp buffer (TSTAMP in rbc3d_kernels_body.hpp) */
#define RBC_TILE_STAMPS reinterpret_cast<const real *>(h->d_stamps)
#else
#define RBC_TILE_STAMPS (const real *)nullptr
#endif
#define RBC_TILE_LAUNCH(TY, KT, THR, WAVES)                                                                                              \
            {                                                                                                                            \
                const dim3 gt((unsigned)(2 * (size_t)B * (g.ny / TY) * (g.nz / KT))), bt(g.nx * TY);                                      \
                const size_t pb = (size_t)(TY + 6) * K3::NXP3 * sizeof(real);                                                        \
                hipLaunchKernelGGL((K3::k3_tile_all<TY, KT, 2, THR, WAVES>), gt, bt, 3 * pb, q.st, g, cur, nxt, gm, actT, ra, d, gam[ph], zet[ph], store_g, RBC_TILE_STAMPS); \
            }
            // (48, 48) horizontal planes -- configs[4] --, the registry default (32, 32) and the flowstats experiment's (64, 64) have
            // instantiations with nx, ny as compile-time constants: the index arithmetic of the plane staging becomes multiplications
            // (a third fewer VALU instructions in the kernel, +4-5 % env-steps/s)
#define RBC_TILE_LAUNCHC(TY, KT, THR, WAVES, NXC_, NYC_)                                                                                 \
            {                                                                                                                            \
                const dim3 gt((unsigned)(2 * (size_t)B * (g.ny / TY) * (g.nz / KT))), bt(g.nx * TY);                                      \
                const size_t pb = (size_t)(TY + 6) * K3::NXP3 * sizeof(real);                                                        \
                hipLaunchKernelGGL((K3::k3_tile_all<TY, KT, 2, THR, WAVES, K3::NXP3, false, NXC_, NYC_>), gt, bt, 3 * pb, q.st, g, cur, nxt, gm, actT, ra, d, gam[ph], zet[ph], store_g, RBC_TILE_STAMPS); \
            }
            const bool c48 = (g.nx == 48 && g.ny == 48 && !no_nxc), c32 = (g.nx == 32 && g.ny == 32 && !no_nxc), c64 = (g.nx == 64 && g.ny == 64 && !no_nxc);
            if (shape == 1 && c48) RBC_TILE_LAUNCHC(16, 16, 768, TW3, 48, 48)
            else if (shape == 2 && c48) RBC_TILE_LAUNCHC(16, 8, 768, TW3, 48, 48)
            else if (shape == 3 && c48) RBC_TILE_LAUNCHC(16, 4, 768, TW3, 48, 48)
            else if (shape == 1 && c32) RBC_TILE_LAUNCHC(16, 16, 768, TW3, 32, 32)
            else if (shape == 2 && c32) RBC_TILE_LAUNCHC(16, 8, 768, TW3, 32, 32)
            else if (shape == 3 && c32) RBC_TILE_LAUNCHC(16, 4, 768, TW3, 32, 32)
            else if (shape == 4 && c64) RBC_TILE_LAUNCHC(8, 8, 512, 2, 64, 64)
#undef RBC_TILE_LAUNCHC
            else if (shape == 1) RBC_TILE_LAUNCH(16, 16, 768, TW3)
            else if (shape == 2) RBC_TILE_LAUNCH(16, 8, 768, TW3)
            else if (shape == 3) RBC_TILE_LAUNCH(16, 4, 768, TW3)
            else if (shape == 4) RBC_TILE_LAUNCH(8, 8, 512, 2)
#undef RBC_TILE_LAUNCH
#define RBC_FLAT_LAUNCH(KT)                                                                                                              \
            {                                                                                                                            \
                const dim3 gt((unsigned)(2 * (size_t)B * (g.nz / KT))), bt(g.nx);                                                         \
                hipLaunchKernelGGL((K3::k3_tile_all<1, KT, 1, 256, 3, 256, true>), gt, bt, 3 * 256 * sizeof(real), q.st, g, cur, nxt, gm, actT, ra, d, gam[ph], zet[ph], store_g, RBC_TILE_STAMPS); \
            }
            else if (shape == 5) RBC_FLAT_LAUNCH(16)
            else if (shape == 6) RBC_FLAT_LAUNCH(8)
            else if (shape == 7) RBC_FLAT_LAUNCH(4)
            else if (shape == 8) RBC_FLAT_LAUNCH(32)
            else if (shape == 9) RBC_FLAT_LAUNCH(64)
#undef RBC_FLAT_LAUNCH
            else if (g.nz % K3::KC3 == 0 && !h->no_march) {      // z-marching kernels (register reuse along z)
                const dim3 gm_(grid_for((size_t)B * g.nx * g.ny * (g.nz / K3::KC3), 128));
                hipLaunchKernelGGL(K3::k3_tend_march<0>, gm_, bc, 0, q.st, g, cur, nxt, gm, phy, actT, ra, d, gam[ph], zet[ph], B);
                if (!h->stream2d)                              // ny = 1: v and its tendency are identically zero in both state buffers
                    hipLaunchKernelGGL(K3::k3_tend_march<1>, gm_, bc, 0, q.st, g, cur, nxt, gm, phy, actT, ra, d, gam[ph], zet[ph], B);
                hipLaunchKernelGGL(K3::k3_tend_march<2>, gm_, bc, 0, q.st, g, cur, nxt, gm, phy, actT, ra, d, gam[ph], zet[ph], B);
                hipLaunchKernelGGL(K3::k3_tend_march<3>, gm_, bc, 0, q.st, g, cur, nxt, gm, phy, actT, ra, d, gam[ph], zet[ph], B);
            } else {
                hipLaunchKernelGGL(K3::k3_tendency<0>, gc, bc, 0, q.st, g, cur, nxt, gm, phy, actT, ra, d, gam[ph], zet[ph], B, (real *)nullptr);
                if (!h->stream2d)
                    hipLaunchKernelGGL(K3::k3_tendency<1>, gc, bc, 0, q.st, g, cur, nxt, gm, phy, actT, ra, d, gam[ph], zet[ph], B, (real *)nullptr);
                hipLaunchKernelGGL(K3::k3_tendency<2>, gc, bc, 0, q.st, g, cur, nxt, gm, phy, actT, ra, d, gam[ph], zet[ph], B, (real *)nullptr);
                hipLaunchKernelGGL(K3::k3_tendency<3>, gc, bc, 0, q.st, g, cur, nxt, gm, phy, actT, ra, d, gam[ph], zet[ph], B, (real *)nullptr);
            }
            // the potential is an output (pNHS) only after the last stage of the control interval; that stage also completes
            // its own projection (outputs and the next env-step read the projected state)
            const bool last = (n == nsub - 1 && ph == 2);
            const double dts = (gam[ph] + zet[ph]) * d;
            if (int rc = project3d(h, q, which ^ 1, dts, nullptr, last)) return rc;
            which ^= 1;
        }
    }
    HIP3(hipGetLastError());
    s->unsplit_phi = h->stream2d && shape != 0;
    *which_out = which;
    return RBC_OK;
}

// advance + outputs for the whole batch: every env group's chain on its own stream, forked from and joined back into the
// handle's stream (the same calls capture into a graph)
int run_step3d(rbc_handle *h, const float *actions_dev, int nsub, double dt, double dt_last, bool with_output = true)
{
    rbc3_state *s = S3(h);
    int which = s->cur;
    const int per_slice = (h->B + s->slices - 1) / s->slices;
    for (int sl = 0; sl < s->slices; ++sl) {                      // slices run one after the other (stream order / join events)
        const int s0 = sl * per_slice, Bs = (s0 + per_slice <= h->B) ? per_slice : h->B - s0;
        if (Bs <= 0) continue;
        if (s->groups <= 1) {
            const rbc3_grp q{s0, Bs, h->stream, per_slice};
            if (int rc = advance3d(h, q, s->cur, actions_dev, nsub, dt, dt_last, &which)) return rc;
            if (with_output) if (int rc = output3d(h, q, which, nullptr)) return rc;
            continue;
        }
        HIP3(hipEventRecord(s->gstart, h->stream));
        const int per = (per_slice + s->groups - 1) / s->groups, base = Bs / s->groups, rem = Bs % s->groups;      // balanced: 17 envs = 5 + 4 + 4 + 4
        std::vector<rbc3_grp> qs;
        for (int gi = 0, e0 = s0; gi < s->groups; ++gi) {
            const int Bg = base + (gi < rem ? 1 : 0);
            if (Bg <= 0) continue;
            qs.push_back(rbc3_grp{e0, Bg, s->gstream[gi], per});
            e0 += Bg;
            HIP3(hipStreamWaitEvent(s->gstream[gi], s->gstart, 0));
        }
        // The chains are issued STAGE BY STAGE across the groups, not chain by chain: a captured graph is replayed in the order
        // its nodes were recorded (~4 us of host time per node), so chain by chain the last of four chains started 1.5-1.8 ms after
        // the first -- invisible back to back, a quarter of the step for a synchronous caller (scripts/sync_step_latency.py).
        static const bool by_chain = [] { const char *v = std::getenv("RBC_3D_ISSUE_ORDER"); return v && v[0] == 'c'; }();
        const int nst = 3 * nsub, chunk = by_chain ? nst : 1;
        int w = s->cur;
        for (int st = 0; st < nst; st += chunk) {
            int wn = w;
            for (const rbc3_grp &q : qs)
                if (int rc = advance3d(h, q, w, actions_dev, nsub, dt, dt_last, &wn, st, st + chunk)) return rc;
            w = wn;
        }
        which = w;
        for (size_t gi = 0; gi < qs.size(); ++gi) {
            if (with_output) if (int rc = output3d(h, qs[gi], which, nullptr)) return rc;
            HIP3(hipEventRecord(s->gdone[gi], qs[gi].st));
            HIP3(hipStreamWaitEvent(h->stream, s->gdone[gi], 0));
        }
    }
    s->cur = which;
    return RBC_OK;
}

int step3d(rbc_handle *h, const float *actions_dev, int nsub, double dt, double dt_last, bool timed)
{
    rbc3_state *s = S3(h);
    const bool rec = timed && h->profiling && 2 * (h->ev_used + 1) <= h->ev.size();
    // (the legacy default stream -- rbc_set_stream(h, hipStreamLegacy) -- cannot be captured: direct launches there)
    const bool recorded = (h->cfg.reference_clock == RBC_CLOCK_RECORDED);
    const int var = (nsub == h->nsub) ? 0 : ((recorded && nsub == h->nsub - 1) ? 1 : -1);      // which captured graph this step is
    const bool standard = (var >= 0) && (dt == h->dt_solver_eff) && (dt_last == h->dt_last) && !h->no_graph &&
                          h->stream != nullptr && h->stream != hipStreamPerThread;      // (legacy = the null stream here, see rbc_set_stream)
    if (standard && actions_dev != h->d_actions)     // the graph reads the handle's own action buffer
        HIP3(hipMemcpyAsync(h->d_actions, actions_dev, (size_t)h->B * (h->stream2d ? 1 : s->g.heaters) * s->g.heaters * sizeof(float),
                            hipMemcpyDeviceToDevice, h->stream));
    if (rec) HIP3(hipEventRecord(h->ev[2 * h->ev_used], h->stream));
    if (standard && s->own_queues) {
        // per-group graphs: captured once per (parity, length) on the groups' own streams, replayed side by side
        const int par = s->cur, G = s->groups;
        const int base = h->B / G, rem = h->B % G, per = (h->B + G - 1) / G;
        std::vector<hipGraphExec_t> &gg = s->ggexec[par][var];
        if (gg.empty()) {
            gg.assign((size_t)G, nullptr);
            int which = par;
            for (int gi = 0, e0 = 0; gi < G; ++gi) {
                const int Bg = base + (gi < rem ? 1 : 0);
                if (Bg <= 0) continue;
                const rbc3_grp q{e0, Bg, s->gstream[gi], per};
                e0 += Bg;
                hipGraph_t graph = nullptr;
                HIP3(hipStreamBeginCapture(q.st, hipStreamCaptureModeRelaxed));
                int rc = advance3d(h, q, par, h->d_actions, nsub, dt, dt_last, &which);
                if (!rc) rc = output3d(h, q, which, nullptr);
                hipError_t e = hipStreamEndCapture(q.st, &graph);
                if (rc || e != hipSuccess) {
                    if (graph) (void)hipGraphDestroy(graph);
                    drop_graphs3d(h);
                    return rc ? rc : fail(RBC_ERR_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
                }
                e = hipGraphInstantiate(&gg[(size_t)gi], graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (e != hipSuccess) { drop_graphs3d(h); return fail(RBC_ERR_DEVICE, std::string("hipGraphInstantiate: ") + hipGetErrorString(e)); }
            }
        }
        HIP3(hipEventRecord(s->gstart, h->stream));                // the action upload (and whatever the caller ordered before it)
        for (int gi = 0; gi < G; ++gi) {
            if (!gg[(size_t)gi]) continue;
            HIP3(hipStreamWaitEvent(s->gstream[gi], s->gstart, 0));
            HIP3(hipGraphLaunch(gg[(size_t)gi], s->gstream[gi]));
            HIP3(hipEventRecord(s->gdone[gi], s->gstream[gi]));
        }
        for (int gi = 0; gi < G; ++gi)
            if (gg[(size_t)gi]) HIP3(hipStreamWaitEvent(h->stream, s->gdone[gi], 0));
        s->cur = par ^ ((3 * nsub) & 1);
    } else if (standard) {
        const int par = s->cur;
        if (!s->gexec[par][var]) {
            hipGraph_t graph = nullptr;
            HIP3(hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed));
            int rc = run_step3d(h, h->d_actions, nsub, dt, dt_last);
            hipError_t e = hipStreamEndCapture(h->stream, &graph);
            s->cur = par;                               // capture executed nothing: undo the host-side flips
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            if (e != hipSuccess) return fail(RBC_ERR_DEVICE, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
            HIP3(hipGraphInstantiate(&s->gexec[par][var], graph, nullptr, nullptr, 0));
            (void)hipGraphDestroy(graph);
        }
        HIP3(hipGraphLaunch(s->gexec[par][var], h->stream));
        s->cur = par ^ ((3 * nsub) & 1);
    } else {
        if (int rc = run_step3d(h, actions_dev, nsub, dt, dt_last)) return rc;
    }
    if (rec) {
        HIP3(hipEventRecord(h->ev[2 * h->ev_used + 1], h->stream));
        h->ev_used++;
    }
    return RBC_OK;
}

// RBC_CLOCK_RECORDED with a mixed batch: one solver step for the envs marked in `fresh_dev` only.  The stage kernels have no env
// mask, so the whole batch takes the step (same env groups, hence the same tile instantiations, as an env-step) and the envs that
// were not marked get their state back afterwards; the marked ones are then exactly -- bit for bit -- where the first substep of
// a full interval would have put them.
int lead_substep3d(rbc_handle *h, const float *actions_dev, const uint8_t *fresh_dev)
{
    rbc3_state *s = S3(h);
    const size_t n = (size_t)h->B * s->g.env_stride;
    if (!s->lead_save) HIP3(hipMalloc(&s->lead_save, n * sizeof(real)));
    HIP3(hipMemcpyAsync(s->lead_save, s->st[s->cur], n * sizeof(real), hipMemcpyDeviceToDevice, h->stream));
    if (int rc = run_step3d(h, actions_dev, 1, h->dt_solver_eff, h->dt_solver_eff, false)) return rc;
    const int which = s->cur;
    const size_t words = s->g.env_stride * (sizeof(real) / 4);
    hipLaunchKernelGGL(K3C::k3_restore_unmarked, dim3((unsigned)h->B, 64), dim3(256), 0, h->stream, reinterpret_cast<uint32_t *>(s->st[which]),
                       reinterpret_cast<const uint32_t *>(s->lead_save), fresh_dev, words);
    HIP3(hipGetLastError());
    return RBC_OK;
}

// finish a reset of the masked envs: set!'s projection with unit step + outputs
int finish_reset3d(rbc_handle *h)
{
    rbc3_state *s = S3(h);
    const rbc3_grp q = whole_batch(h);
    wall3d(h, q, nullptr, 1);
    s->unsplit_phi = false;                                      // set!'s projection: phi is the potential itself
    if (h->stream2d)
        hipLaunchKernelGGL(K3::k2s_clear_v, grid_for((size_t)h->B * s->g.nc, 256), dim3(256), 0, h->stream, s->g, s->st[0], s->st[1], h->d_mask, h->B);
    if (int rc = project3d(h, q, s->cur, 1.0, h->d_mask)) return rc;
    if (int rc = output3d(h, q, s->cur, h->d_mask)) return rc;
    HIP3(hipStreamSynchronize(h->stream));
    return RBC_OK;
}

// ---- state I/O of the C ABI (float64 arrays at the boundary whatever the solver's precision) ------------------------------------
// random initial condition of the masked envs (seeds already uploaded to h->d_seeds), then set!'s projection and the outputs
int random_reset3d(rbc_handle *h)
{
    rbc3_state *s = S3(h);
    if (h->stream2d)
        hipLaunchKernelGGL(K3::k2s_random, grid_for((size_t)h->B * s->g.nc, 256), dim3(256), 0, h->stream, s->g, s->st[s->cur], h->d_seeds, h->d_mask, h->B);
    else
        hipLaunchKernelGGL(K3::k3_random, grid_for((size_t)h->B * s->g.nw, 256), dim3(256), 0, h->stream, s->g, s->st[s->cur], h->d_seeds, h->d_mask, h->B);
    return finish_reset3d(h);
}

// one env's state from a float64 staging array in the streaming layout [b | u | v | w] (env_stride values)
int put_env3d(rbc_handle *h, int e, const double *stage)
{
    rbc3_state *s = S3(h);
    const size_t n = s->g.env_stride;
    if constexpr (std::is_same<real, double>::value) {
        HIP3(hipMemcpy(s->st[s->cur] + (size_t)e * n, stage, n * sizeof(double), hipMemcpyHostToDevice));
    } else {
        std::vector<real> tmp(stage, stage + n);
        HIP3(hipMemcpy(s->st[s->cur] + (size_t)e * n, tmp.data(), n * sizeof(real), hipMemcpyHostToDevice));
    }
    return RBC_OK;
}

// field `f` (0: b, 1: u, 2: v, 3: w) of every env into a dense float64 array [B][count]
int get_field3d(rbc_handle *h, int f, double *out)
{
    rbc3_state *s = S3(h);
    const size_t nc = s->g.nc, count = (f == 3) ? (size_t)s->g.nw : nc, pitch = s->g.env_stride * sizeof(real);
    const real *base = s->st[s->cur] + (size_t)f * nc;
    if constexpr (std::is_same<real, double>::value) {
        HIP3(hipMemcpy2D(out, count * sizeof(double), base, pitch, count * sizeof(double), h->B, hipMemcpyDeviceToHost));
    } else {
        std::vector<real> tmp((size_t)h->B * count);
        HIP3(hipMemcpy2D(tmp.data(), count * sizeof(real), base, pitch, count * sizeof(real), h->B, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); ++i) out[i] = (double)tmp[i];
    }
    return RBC_OK;
}

// outputs of the current state again (after the observation normalisation changed)
int refresh_outputs3d(rbc_handle *h)
{
    if (int rc = output3d(h, whole_batch(h), S3(h)->cur, nullptr)) return rc;
    HIP3(hipStreamSynchronize(h->stream));
    return RBC_OK;
}

void *dev_fields3d(rbc_handle *h) { rbc3_state *s = S3(h); return (void *)s->st[s->cur]; }
size_t env_stride3d(rbc_handle *h) { return S3(h)->g.env_stride; }
size_t faces3d(rbc_handle *h) { return (size_t)S3(h)->g.nw; }

// tendencies G of the current state for the actions already in h->d_actions (cell-per-thread kernels with the hydrostatic split):
// outs[q] (q = 0: u, 1: v, 2: w, 3: b; null = skip) as dense float64 arrays [B][nc]
int debug_tendencies3d(rbc_handle *h, double *const outs[4])
{
    rbc3_state *s = S3(h);
    const K3::Geo3 &g = s->g;
    const int B = h->B;
    const size_t nc = g.nc;
    if (!s->dbg) HIP3(hipMalloc(&s->dbg, (size_t)B * 4 * nc * sizeof(real)));
    wall3d(h, whole_batch(h), h->d_actions, 0);
    real *cur = s->st[s->cur];
    hipLaunchKernelGGL(K3::k3_hydrostatic, grid_for((size_t)B * g.nx * g.ny, 128), dim3(128), 0, h->stream, g, cur, s->phy, B);
    const dim3 gc = grid_for((size_t)B * g.nc, 128), bc(128);
    hipLaunchKernelGGL(K3::k3_tendency<0>, gc, bc, 0, h->stream, g, cur, cur, s->gm, s->phy, s->actT, h->d_ra, 0.0, 1.0, 0.0, B, s->dbg);
    if (!h->stream2d)
        hipLaunchKernelGGL(K3::k3_tendency<1>, gc, bc, 0, h->stream, g, cur, cur, s->gm, s->phy, s->actT, h->d_ra, 0.0, 1.0, 0.0, B, s->dbg);
    hipLaunchKernelGGL(K3::k3_tendency<2>, gc, bc, 0, h->stream, g, cur, cur, s->gm, s->phy, s->actT, h->d_ra, 0.0, 1.0, 0.0, B, s->dbg);
    hipLaunchKernelGGL(K3::k3_tendency<3>, gc, bc, 0, h->stream, g, cur, cur, s->gm, s->phy, s->actT, h->d_ra, 0.0, 1.0, 0.0, B, s->dbg);
    HIP3(hipGetLastError());
    HIP3(hipStreamSynchronize(h->stream));
    const size_t pitch = 4 * nc * sizeof(real);
    std::vector<real> tmp;
    for (int q = 0; q < 4; ++q) {
        if (!outs[q]) continue;
        if constexpr (std::is_same<real, double>::value) {
            HIP3(hipMemcpy2D(outs[q], nc * sizeof(double), s->dbg + q * nc, pitch, nc * sizeof(double), B, hipMemcpyDeviceToHost));
        } else {
            tmp.resize((size_t)B * nc);
            HIP3(hipMemcpy2D(tmp.data(), nc * sizeof(real), s->dbg + q * nc, pitch, nc * sizeof(real), B, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < tmp.size(); ++i) outs[q][i] = (double)tmp[i];
        }
    }
    return RBC_OK;
}

}  // namespace

}  // namespace RBC3_HOST
