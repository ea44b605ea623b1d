/*
 * rbc_hip.h -- C ABI of the MI355X-native batched Rayleigh-Benard stepper (librbc_hip.so).
 *
 * This is the drop-in boundary for the hot path of MichielStraat/RBC-Gym: it replaces the
 * Julia "plugin API" that the reference's gym envs call through juliacall
 *   src/rbc_gym/sim/rbc_sim2D_api.jl : initialize_simulation (:17), step_simulation (:75),
 *                                      get_state (:102), get_observation (:123),
 *                                      get_info (:134), get_nusselt (:142)
 * (call sites rbc2D.py:143,169,185,192,199,203-205) together with the un-vendored
 * Oceananigans.jl v0.92.0 solver underneath it (juliapkg.json:7-10).  Differences from the
 * reference interface, by design:
 *   - BATCHED: one handle owns B independent env instances resident in the HBM of one GPU
 *     (the reference holds one env in Julia module globals, one Julia runtime per env);
 *   - opaque handle instead of module globals; integer status + rbc_last_error() instead of
 *     Julia exceptions; nothing throws across the ABI;
 *   - arrays are C-order [env][channel][z][x] (what rbc2D.py produces AFTER its transposes),
 *     float32 for observations/state exactly like the Python boundary (rbc2D.py:185,192);
 *     all solver arithmetic is float64 like the reference.
 * Plain pointers and sizes only; no torch / HIP types in any signature (streams travel as
 * void*).  "_dev" entry points take DEVICE pointers (zero-copy PyTorch-ROCm tensors).
 *
 * Threading: calls on one handle must be serialised by the caller; different handles
 * (different GPUs) may be driven from different threads.  The library owns device state;
 * the caller owns every I/O buffer.
 */
#ifndef RBC_HIP_H
#define RBC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RBC_ABI_VERSION 3

/* status codes */
enum {
    RBC_OK = 0,
    RBC_ERR_INVALID = 1,      /* bad argument / unsupported configuration              */
    RBC_ERR_DEVICE = 2,       /* HIP runtime error (message in rbc_last_error)         */
    RBC_ERR_NAN = 3,          /* at least one env produced NaNs (see rbc_get_flags)    */
    RBC_ERR_NOT_INITIALIZED = 4 /* step/get before the first reset (rbc_sim2D_api.jl:79-81) */
};

/* Mirrors the kwargs of initialize_simulation (rbc_sim2D_api.jl:17) plus the constants it
   hard-codes (:28-38).  Shapes are in the reference's Julia order (x, z). */
typedef struct rbc_config {
    int32_t abi_version;     /* = RBC_ABI_VERSION                                            */
    int32_t dim;             /* 2, or 3 (rbc_sim3D_api.jl: grid (nx,ny,nz), L (lx,ly,lz), heaters x heaters
                                actions, dt_* in free-fall units scaled by t_ff = lz^2, 4 channels b,u,v,w) */
    int32_t nx, ny, nz;      /* grid = state_shape[::-1]  (96, 1, 64).  dim=2, float64: any nx, nz >= 8 (the reference
                                takes any grid: rbc_sim2D_api.jl:17-25); the LDS-resident kernel where it is built for
                                the grid, the streaming kernels with ny = 1 otherwise          */
    double  lx, ly, lz;      /* L = [2*pi, 2]                                   (api:28)     */
    double  ra;              /* default Rayleigh number for every env            (api:17)     */
    double  pr;              /* 0.7                                             (api:35)     */
    double  min_b, delta_b;  /* 1, 1                                            (api:30,36)  */
    int32_t heaters;         /* 12                                              (api:31)     */
    double  heater_limit;    /* 0.75                                            (api:32)     */
    double  dt_solver;       /* 0.03                                            (api:38)     */
    double  dt_control;      /* heater_duration = 1.5                           (api:33)     */
    double  random_kick;     /* 0.01                                            (api:37)     */
    int32_t obs_nx, obs_nz;  /* sensors = observation_shape[::-1] (48, 8)       (api:27)     */
    int32_t batch;           /* number of env instances B on this device                     */
    int32_t device;          /* HIP device ordinal                                           */
    int32_t write_state;     /* 1: rbc_step also refreshes the float32 full-state buffer     */
    int32_t precision;       /* arithmetic of the solver: RBC_PRECISION_F64 (0, the reference's Float64, default) or
                                RBC_PRECISION_F32 (1: float32 state and arithmetic) -- every path:
                                the LDS-resident 2D kernels, the streaming 2D grids and dim=3 (rbc3D.py:229-232 hands out
                                float32 observations anyway).  The I/O types of the ABI do not change with it (float64
                                fields in and out, float32 obs/state; Nusselt sums stay float64); only rbc_dev_fields()
                                exposes the solver's own element type (float for a float32 streaming / 3D handle).  */
    int32_t reference_clock; /* how many solver steps an env-step integrates (ABI 3).
                                RBC_CLOCK_DOCUMENTED (0, default): what the reference's SOURCES say -- run!(simulation) to
                                stop_time, stop_time += dt (rbc_sim2D_api.jl:84-85, rbc_sim3D_api.jl:88-89): every env-step
                                integrates dt_control, i.e. ceil(dt_control / dt_solver) solver steps.
                                RBC_CLOCK_RECORDED (1): what the reference's only RECORDED time series shows
                                (experiments/flowstats/flowstats_ra.pkl, written by flowstats_ra.py:55-66 through
                                rbc_sim3D_api.jl:77-101): the first env-step after a reset carries the growth of all its
                                solver steps, every later one of one solver step less (tau(n) = 1 + 0.98 (n - 1) at 50
                                solver steps per env-step on all four series).  With this value the
                                first rbc_step after a reset of an env integrates dt_control and every later one
                                dt_control - dt_solver (one full solver step dropped, a clipped last substep kept), while
                                t and step reported by rbc_get_info advance as documented.  Per env: a masked reset
                                restores the full first interval for the envs it resets, whatever the others are doing.
                                Needs dt_control > dt_solver.  dim = 2 and 3, every precision and path.                */
} rbc_config;

enum { RBC_PRECISION_F64 = 0, RBC_PRECISION_F32 = 1 };
enum { RBC_CLOCK_DOCUMENTED = 0, RBC_CLOCK_RECORDED = 1 };

typedef struct rbc_handle rbc_handle;

/* library / device */
int         rbc_abi_version(void);
const char *rbc_last_error(void);                 /* thread-local message of the last failure */
int         rbc_device_count(void);               /* number of visible HIP devices (0 if none) */
int         rbc_has_precision(int precision);     /* 1 if this build carries kernels for RBC_PRECISION_* */
void        rbc_default_config(rbc_config *cfg);  /* the gym registry defaults (__init__.py:7-18) */

/* lifetime: replaces gym.make()'s juliacall.newmodule + include (rbc2D.py:111-115) */
int  rbc_create(const rbc_config *cfg, rbc_handle **out);
int  rbc_destroy(rbc_handle *h);
/* Stream all launches of the handle go to.  NULL = the handle's own (non-blocking) stream, which has NO implicit ordering
   with any other stream: complete what the sim will read (actions written by another stream) before calling rbc_step_dev,
   and rbc_synchronize (or an event) before another stream reads the rbc_dev_* views.  To run ON the legacy default stream
   (what PyTorch's default stream, handle 0, is) pass hipStreamLegacy = (void*)1; any other value is a hipStream_t.      */
int  rbc_set_stream(rbc_handle *h, void *hip_stream);
void *rbc_get_stream(rbc_handle *h);
int  rbc_synchronize(rbc_handle *h);

/* per-env Rayleigh numbers (Ra sweep); ra[B] host pointer.  nu, kappa per api:40-41 */
int  rbc_set_rayleigh(rbc_handle *h, const double *ra);

/* RBCNormalizeObservation (wrappers/rbc_normalize_observation.py:66-74) fused into the observation write of
   the 2D step kernel: obs[c] = maxval * (2 * (obs[c] - min_vals[c]) / (max_vals[c] - min_vals[c]) - 1), evaluated
   in float32 in exactly that order on the float32-rounded sample, the python-float bounds rounded to float32 where
   numpy rounds them (bit-identical to the numpy wrapper), then
   clipped to [-maxval, maxval] if clip != 0 (a NaN stays a NaN, as with np.clip: a blown-up env is visible in its observation as
   well as in rbc_get_flags).  nch <= 5 channels (b,u,w,pHY',pNHS); channels >= nch stay raw;
   nch = 0 switches the transform off.  Takes effect from the next reset/step; rbc_get_state is never transformed.
   dim=3: nch <= 4 channels (b,u,v,w).  The 3D observation IS the float32 state buffer (rbc3D.py:229-232), so there the
   transform applies to what rbc_get_obs / rbc_get_state / rbc_dev_state hand out (rewritten at once for the current state);
   Nusselt number, NaN flag and rbc_get_fields3 always come from the raw float64 state. */
int  rbc_set_obs_normalization(rbc_handle *h, const double *min_vals, const double *max_vals, int nch, double maxval, int clip);

/* RBCRewardShaping.compute_cell_distances (wrappers/rbc_reward_shaping.py:85-140) on the device, for every env of a dim=2
   handle created with write_state=1: the largest periodic distance between two up-welling plumes (peaks of the float32
   vertical-velocity channel on the row nz/2 - 1 that reach `height`, the reference uses 0.001), 0 if the signal stays
   positive between them; domain length = cfg.lx.  out[B] float64 on the host; bit-identical to the numpy wrapper.
   rbc_dev_cell_dist: the device buffer the last call filled (NULL before the first call).
   rbc_debug_cell_distances: the same kernel on caller-provided host signals uy[B][nx] (parity tests).                  */
int   rbc_get_cell_distances(rbc_handle *h, double height, double *out);
void *rbc_dev_cell_dist(rbc_handle *h);
int   rbc_debug_cell_distances(int device, const float *uy, int B, int nx, double lx, double height, double *out);

/* Page-locked host buffers for the rbc_get_* outputs: a pinned destination takes the float32 state copy of 1024 envs (75 MB)
   at PCIe speed.  Plain helpers over hipHostMalloc / hipHostFree; any rbc_get_* accepts either kind of pointer.  Into PAGEABLE
   memory, rbc_get_obs / rbc_get_state outputs of 8 MiB and more are staged in chunks through a page-locked buffer of the handle
   and moved on by up to four host threads while the next chunk is in flight (RBC_STAGED_COPY=0: one hipMemcpy2D).      */
void *rbc_host_alloc(size_t bytes);
void  rbc_host_free(void *p);

/* initialize_simulation (api:17-70).  mask[B] (NULL = all): which envs to reset.
   Random IC (rbc_sim2D.jl:163-171) from the library's counter-based RNG, seeds[B].        */
int  rbc_reset(rbc_handle *h, const uint8_t *mask, const uint64_t *seeds);
/* checkpoint / identical-IC path (initialize_from_checkpoint, rbc_sim2D.jl:173-186):
   host arrays b,u: [B][nz][nx], w: [B][nz+1][nx] float64; only masked envs are read.      */
int  rbc_reset_from_arrays(rbc_handle *h, const uint8_t *mask,
                           const double *b, const double *u, const double *w);

/* 3D variants (initialize_from_checkpoint rbc_sim3D.jl:181-199; get_state rbc_sim3D_api.jl:106-121):
   b,u,v: [B][nz][ny][nx], w: [B][nz+1][ny][nx] float64                                      */
int  rbc_reset_from_arrays3(rbc_handle *h, const uint8_t *mask,
                            const double *b, const double *u, const double *v, const double *w);
int  rbc_get_fields3(rbc_handle *h, double *b, double *u, double *v, double *w);

/* step_simulation (api:75-97): actions [B][heaters] float32 in [-1,1] (host pointer;
   dim=3: [B][heaters][heaters], preprocess_action rbc_sim3D.jl:111-128 is applied on the device).
   Advances every env by dt_control. Returns RBC_OK, or RBC_ERR_NAN if any env has NaNs.   */
int  rbc_step(rbc_handle *h, const float *actions);
int  rbc_step_dev(rbc_handle *h, const float *actions_dev);  /* async on the handle's stream */

/* get_observation (api:123-129) after rbc2D.py:191-196: [B][nch][obs_nz][obs_nx] float32,
   channels b,u,w,(pHY',pNHS); nch = 3 or 5.                                               */
int  rbc_get_obs(rbc_handle *h, float *out, int nch);
/* get_state (api:102-118) after rbc2D.py:184-189: [B][nch][nz][nx] float32                */
int  rbc_get_state(rbc_handle *h, float *out, int nch);
/* raw float64 prognostic fields (checkpoint writer / parity tests): b,u [B][nz][nx], w [B][nz+1][nx] */
int  rbc_get_fields(rbc_handle *h, double *b, double *u, double *w);
/* get_nusselt (api:142-163): nu_state[B], nu_obs[B] float64 (either may be NULL)          */
int  rbc_get_nusselt(rbc_handle *h, double *nu_state, double *nu_obs);
/* get_info (api:134-137): t[B] float64, step[B] int64                                     */
int  rbc_get_info(rbc_handle *h, double *t, int64_t *step);
/* step_contains_NaNs (rbc_sim2D.jl:223-228): flags[B], 1 = NaN in b,u or w                */
int  rbc_get_flags(rbc_handle *h, int32_t *flags);

/* device-resident views for zero-copy consumers (valid until rbc_destroy):
   obs float32 [B][5][obs_nz][obs_nx], state float32 [B][5][nz][nx],
   nusselt float64 [B][2] (state, obs), flags int32 [B]                                    */
void *rbc_dev_obs(rbc_handle *h);
void *rbc_dev_state(rbc_handle *h);
void *rbc_dev_nusselt(rbc_handle *h);
void *rbc_dev_flags(rbc_handle *h);
void *rbc_dev_fields(rbc_handle *h);   /* float64 [B][ b(nz*nx) | u(nz*nx) | w((nz+1)*nx) ]; 2D handles on the streaming path
                                          (grids without an LDS-resident kernel): [B][ b | u | v = 0 | w ] of the buffer that
                                          currently holds the state (it alternates between two with every RK3 stage) */

/* measurement support for bench.py: HIP events are recorded on the handle's stream around
   every step-kernel launch of rbc_step / rbc_step_dev (up to max_launches launches between
   two reads; 0 disables).  rbc_profile_read waits for the recorded launches, writes their
   durations in milliseconds to ms[0..capacity) and returns how many it wrote (-1 on error). */
int    rbc_set_profiling(rbc_handle *h, int max_launches);
int    rbc_profile_read(rbc_handle *h, double *ms, int capacity);
/* On-box streaming ceiling to report next to the 8 TB/s spec: copies `bytes` (rounded down to 16) from one
   device buffer to another `iters` times after a warm-up, with a 16-bytes-per-lane grid-stride copy kernel (best of five grid
   sizes) and with hipMemcpyAsync device-to-device; rates in GB/s count bytes read + bytes written.  Either output may be NULL.   */
int    rbc_copy_ceiling(int device, size_t bytes, int iters, double *kernel_gbs, double *memcpy_gbs);
/* algorithmic HBM bytes of one env-step per env under SURVEY.md 8(d)'s convention          */
double rbc_algorithmic_bytes_per_env_step(rbc_handle *h);

/* operator-level test hooks (used by the parity tests only; not needed by the env layer):
   tendencies G(b,u,w) of the current state for the given actions: gb,gu,gw [B][nz][nx]    */
int  rbc_debug_tendencies(rbc_handle *h, const float *actions, double *gb, double *gu, double *gw);
int  rbc_debug_tendencies3(rbc_handle *h, const float *actions, double *gu, double *gv, double *gw, double *gb);
/* run `nsub` RK3 substeps of size dt with the given actions (no counters touched)          */
int  rbc_debug_substeps(rbc_handle *h, const float *actions, int nsub, double dt);
/* how the streaming path (3D, streaming 2D) replays an env-step: groups[0] = env groups (stream chains) of the handle, groups[1] =
   1 if every group replays its own captured graph on a stream with a hardware queue of its own (found by probing at rbc_create), 0 if
   the groups are branches of one graph (or launches are direct).  A handle on the LDS-resident 2D kernel reports {1, 0}.          */
int  rbc_debug_launch_plan(rbc_handle *h, int groups[2]);
/* diagnostic builds only (-DRBC_STAMPS=1): per-phase shader-clock cycles of the last launch,
   out[B][64]; returns RBC_ERR_INVALID in the shipped build                                 */
int  rbc_debug_stamps(rbc_handle *h, unsigned long long *out);

#ifdef __cplusplus
}
#endif
#endif
