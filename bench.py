#!/usr/bin/env python3
"""bench.py -- env-steps/s of the batched 2D Rayleigh-Benard stepper on N MI355X (one process per GPU).

A "step" is one env.step() of ALL B envs resident on a GPU: one control interval of
heater_duration=1.5 = 50 RK3 substeps (150 stages, 150 pressure solves) per env, the reference's
default `gym.make("rbc_gym/RayleighBenardConvection2D-v0")` settings (64x96, Ra=1e4, dt 0.03).
Workload = BASELINE.json configs[1]: B=1024 envs per GPU, synthetic random-perturbation initial
states (per-env seed 1234+i) and U(-1,1) actions re-drawn every step, both resident in HBM before
the timed region.  N>1: env instances are sharded across ranks, no data-path collective (weak
scaling: 1024 envs per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      # without a launcher: starts the N ranks itself (before any GPU call)

The JSON line carries, besides the contract's keys: `roofline` (SURVEY.md 8(d)'s algorithmic-bytes convention, with
the PMC-measured HBM rate beside it), `roofline_valu` (the physically binding one: fp64 VALU issue), `copy_ceiling`
(an on-box streaming-copy rate next to the 8 TB/s spec), `cpu_baseline` and `extra` (gym-API rate, configs[4] 3D,
fp32 variant, a 128x64 grid on the streaming 2D path -- all measured in this run).
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_SIMD = 256 * 4             # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9             # peak engine clock; a wave64 fp64 VALU instruction occupies its SIMD for 4 cycles
FP64_WAVE_INST_PEAK = N_SIMD * CLOCK_HZ / 4      # = 6.144e11 /s  (x 64 lanes x 2 flop = 78.6 TFLOP/s for FMA)


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle ("port"), envs spread over ALL host cores of the box.  Runs BEFORE
# anything touches the GPU (fork-based pool).  SURVEY.md 8(d): B = 64 envs of the bench workload.
# ---------------------------------------------------------------------------------------------
def _cpu_worker(args):
    idxs, nsteps, dim = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_py
    sims, rngs = [], []
    for idx in idxs:
        rng = np.random.default_rng(1234 + idx)
        if dim == 3:
            sim = oracle_py.Oracle3D(ra=1e4, shape=(32, 48, 48), dt_control=0.125, dt_solver=0.01)
        else:
            sim = oracle_py.OracleSim(ra=1e4)
        sim.reset_random(1234 + idx)
        sims.append(sim); rngs.append(rng)
    shape = (8, 8) if dim == 3 else (12,)
    t0 = time.perf_counter()
    for _ in range(nsteps):
        for sim, rng in zip(sims, rngs):
            sim.step(rng.uniform(-1, 1, shape).astype(np.float32))
    return time.perf_counter() - t0


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_quota():
    """CPU time this process may use per wall second, in cores, from the cgroup (v2 cpu.max, v1 cfs quota); None = unlimited."""
    try:
        rel = "/"
        for line in open("/proc/self/cgroup"):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                rel = parts[2]
        best = None
        path = os.path.join("/sys/fs/cgroup", rel.lstrip("/"))
        while True:                                   # the tightest limit on the way up to the root
            f = os.path.join(path, "cpu.max")
            if os.path.exists(f):
                q, per = open(f).read().split()[:2]
                if q != "max":
                    c = float(q) / float(per)
                    best = c if best is None else min(best, c)
            if os.path.realpath(path) == "/sys/fs/cgroup":
                break
            path = os.path.dirname(path)
        if best is not None:
            return best
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def _physical_cores(allowed):
    """distinct (package, core) pairs among the logical CPUs this process may run on (SMT siblings count once)"""
    seen = set()
    for c in allowed:
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology"
            seen.add((open(f"{base}/physical_package_id").read().strip(), open(f"{base}/core_id").read().strip()))
        except OSError:
            seen.add(("?", c))
    return len(seen)


def cpu_baseline(dim=2, envs=64, budget_s=12.0):
    """The C oracle on the host cores this process REALLY gets: worker count = min(affinity, cgroup quota, physical cores,
    envs); when no quota is published and the affinity mask is wider than 32 CPUs, a short scaling probe (1 / 16 / 64 /
    all workers, one env-step each) picks the widest pool that still scales.  Reports the single-worker rate beside the
    total, so `cores_effective` = total / single-worker rate says how much compute was actually obtained."""
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    oracle_py.build_oracle()
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    nproc = os.cpu_count() or len(allowed)
    quota = _cpu_quota()
    phys = _physical_cores(allowed)
    limit = min(len(allowed), phys, envs if dim == 2 else 32)
    if quota is not None:
        limit = max(1, min(limit, int(quota + 0.5)))
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    _cpu_worker(([0], 1, dim))                                # warm (page cache, tables, clocks) ...
    single = _cpu_worker(([0], 2, dim)) / 2.0                 # ... then one env on one core: the per-core yardstick
    probe = {}

    def rate(workers, nsteps):
        n_env = max(envs, workers) if dim == 2 else workers
        shards = [list(range(r, n_env, workers)) for r in range(workers)]
        with ctx.Pool(workers) as pool:
            per = pool.map(_cpu_worker, [(s, nsteps, dim) for s in shards])
        return n_env, per

    workers = limit
    if quota is None and len(allowed) > 32:                   # no published quota on a wide mask: measure where scaling stops
        for wk in sorted({1, 16, min(64, limit), limit}):
            n_env, per = rate(wk, 1)
            probe[wk] = n_env / max(per)
        workers = max(probe, key=lambda k: probe[k])
    n_env, cal = rate(workers, 1)
    spent = time.perf_counter() - t0
    nsteps = int(max(1, min(200, (budget_s - spent) / max(max(cal), 1e-3))))
    n_env, per = rate(workers, nsteps)
    wall = time.perf_counter() - t0
    stepping = max(per)
    total = n_env * nsteps / stepping
    single_rate = 1.0 / single
    what = ("64x96, Ra=1e4, 50 RK3 substeps, random ICs seeds 1234+i, U(-1,1) actions" if dim == 2 else
            "32x48x48, Ra=1e4, 13 RK3 substeps, random ICs seeds 1234+i, U(-1,1) 8x8 actions")
    return {"value": total, "unit": "env-steps/s", "cores": workers, "kind": "port",
            "per_core": total / workers, "single_worker": single_rate, "cores_effective": total / single_rate,
            "nproc": nproc, "affinity": len(allowed), "physical_cores": phys, "cgroup_cpu_quota": quota,
            "scaling_probe": {str(k): v for k, v in probe.items()} or None, "cpu_model": _cpu_model(),
            "sample": f"{n_env} envs spread over {workers} worker processes (min of affinity {len(allowed)}, cgroup quota {quota}, "
                      f"physical cores {phys}, envs; nproc={nproc}) x {nsteps} env-steps of the bench workload ({what}) on the C oracle "
                      f"(oracle/, readable scalar C, direct O(N^2) DFT in its Poisson solve: a deliberately plain port, a baseline and not "
                      f"a tuned CPU code); one worker alone: {single_rate:.2f} env-steps/s; {stepping:.1f} s stepping, {wall:.1f} s wall incl. "
                      f"calibration. Julia reference unavailable on this box (README.md:62 publishes 0.12 s/step at 34 substeps on one "
                      f"Apple-silicon core = 5.7 env-steps/s at 50)"}


# ---------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------
def self_launch(args):
    """`--gpus N` without a launcher: become the launcher.  Runs before this process touches the GPU
    (torch.cuda.device_count() does not initialise it) and never prints a line of its own."""
    import socket
    import torch
    have = torch.cuda.device_count()
    if args.dist_backend == "nccl" and have < args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {have} HIP device(s) visible; refusing to print a line for "
                         f"fewer GPUs than asked (use --dist-backend gloo to rehearse with ranks sharing a GPU)\n")
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def latest_profile(pattern, match):
    """newest profiles/<pattern> whose recorded bench configuration matches this run (else None, None)"""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        key = d.get("workload_key", {})
        if all(key.get(k, "f64" if k == "precision" else None) == v for k, v in match.items()):     # (profiles older than the float32 3D path carry no precision: float64)
            return d, os.path.relpath(path, ROOT)
    return None, None


def timed_loop(torch, dev, barrier, fn, K):
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for n in range(K):
        fn(n)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0


def run_3d(ctx, B, K, W, ra, precision="f64"):
    """configs[4]: 3D 32x48x48, Ra=1e4, heater_duration 0.125, dt_solver 0.01 (13 RK3 substeps per env-step)."""
    torch, np, _native, sharding = ctx["torch"], ctx["np"], ctx["_native"], ctx["sharding"]
    dev, rank, world = ctx["dev"], ctx["rank"], ctx["world"]
    sim = _native.NativeSim3D(batch=B, device=ctx["local_rank"], shape=(32, 48, 48), ra=ra, precision=precision)
    start, count = sharding.shard(world * B, world, rank)
    sim.reset(sharding.env_seeds(1234, start, count))
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)
    actions = (torch.rand((K + W, B, 8, 8), device=dev, generator=gen, dtype=torch.float32) * 2 - 1).contiguous()
    torch.cuda.synchronize(dev)          # the sim runs on its own stream: the actions must be complete before it reads them
    stride = B * 64 * 4
    for n in range(W):
        sim.step_dev(actions.data_ptr() + n * stride)
    sim.set_profiling(K)
    elapsed = timed_loop(torch, dev, ctx["barrier"], lambda n: sim.step_dev(actions.data_ptr() + (W + n) * stride), K)
    ms = sim.profile_read(K)
    nan_envs = int(sim.get_flags().sum())
    nu = sim.get_nusselt()
    rank_elapsed, rank_nan = sharding.gather_run(elapsed, nan_envs, device=ctx["red_dev"], dist=ctx["dist"] if world > 1 else None)
    elapsed, nan_total = sharding.reduce_run(elapsed, nan_envs, device=ctx["red_dev"], dist=ctx["dist"] if world > 1 else None)
    alg = sim.algorithmic_bytes_per_env_step() * B
    avg = float(np.mean(ms)) if len(ms) else float("nan")
    ach = alg / (avg * 1e-3) / 1e9
    prof, src = latest_profile("*_3d*_summary.json", {"dim": 3, "batch": B, "ra": ra, "precision": precision, "groups": None})
    traffic = prof.get("hbm_traffic_bytes_per_env_step_batch") if prof else None
    sim.close()
    return {
        "metric": "env-steps/sec (batched 3D RBC 32x48x48 Ra=1e4)", "value": world * B * K / elapsed, "unit": "env-steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": precision, "data": "synthetic",
        "config": {"workload": f"configs[4]: batched 3D envs, batch={B} per GPU, grid 32x48x48, Ra={ra:g}, heater_duration 0.125 "
                               "(x t_ff=4) = 13 RK3 substeps per env-step, dt_solver 0.01", "global_batch": world * B,
                   "parallelism": f"env-sharded x{world} (no collective on the step path)"},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": src,
                     "measured_hbm_gbs": (traffic / (avg * 1e-3) / 1e9) if traffic else None,
                     "kernel": "rbc3 stage sequence x 39 per env-step (one 'launch' = one env-step of the batch)",
                     "kernel_ms_avg": avg, "algorithmic_bytes_per_launch": alg,
                     "note": f"algorithmic bytes = 10*F*C*s per substep (F=4, C=73728, s={8 if precision == 'f64' else 4}) x 13 x batch"},
        "nan_envs": nan_total, "mean_nusselt": float(np.mean(nu)),
        "per_rank": {"ms_per_step": [e / K * 1e3 for e in rank_elapsed], "nan_envs": rank_nan}}


def gym_api_rate(ctx, B, steps, info_state):
    """The gym-level number of SURVEY.md 8(d)'s metric text: B x VectorEnv.step() calls per wall second, host actions in,
    host observations / rewards / infos out (PCIe inclusive)."""
    np = ctx["np"]
    from rbc_gym.vector import RayleighBenardConvection2DVectorEnv
    venv = RayleighBenardConvection2DVectorEnv(num_envs=B, device=ctx["local_rank"], info_state=info_state, episode_length=10**9)
    venv.reset(seed=1234)
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1, 1, (steps + 2, B, 12)).astype(np.float32)
    venv.step(acts[0]); venv.step(acts[1])
    t0 = time.perf_counter()
    for n in range(steps):
        venv.step(acts[n + 2])
    dt = time.perf_counter() - t0
    venv.close()
    return B * steps / dt


def ra_sweep_extra(ctx, B, spin, K):
    """configs[3] on one GPU: contiguous thirds of the batch at Ra = 1e4 / 1e5 / 1e6, fixed dt = 0.03 as in the reference
    (substep / CFL stress).  `spin` untimed env-steps first, so that the high-Ra thirds are convecting
    (and at their advective CFL) when the K timed ones run."""
    torch, np, _native, sharding = ctx["torch"], ctx["np"], ctx["_native"], ctx["sharding"]
    dev = ctx["dev"]
    ras = (1e4, 1e5, 1e6)
    ra_env = np.array([ras[min(2, e * 3 // B)] for e in range(B)])
    sim = _native.NativeSim(batch=B, device=ctx["local_rank"], ra=1e4)
    sim.set_rayleigh(ra_env)
    sim.reset(sharding.env_seeds(1234, 0, B))
    gen = torch.Generator(device=dev)
    gen.manual_seed(99)
    acts = (torch.rand((8, B, 12), device=dev, generator=gen, dtype=torch.float32) * 2 - 1).contiguous()
    torch.cuda.synchronize(dev)
    stride = B * 12 * 4
    for n in range(spin):
        sim.step_dev(acts.data_ptr() + (n % 8) * stride)
    e = timed_loop(torch, dev, ctx["barrier"], lambda n: sim.step_dev(acts.data_ptr() + (n % 8) * stride), K)
    flags = sim.get_flags()
    nus, _ = sim.get_nusselt()
    out = {"value": B * K / e, "unit": "env-steps/s", "ms_per_step": e / K * 1e3, "steps": K, "spin_up_steps": spin, "batch": B,
           "rayleigh_numbers": list(ras), "dt_solver": 0.03,
           "nan_envs_per_ra": {f"{r:g}": int(flags[ra_env == r].sum()) for r in ras},
           "mean_nusselt_state_per_ra": {f"{r:g}": float(np.nanmean(nus[ra_env == r])) for r in ras},
           "note": "BASELINE.json configs[3] per GPU: batch split into contiguous thirds, one Rayleigh number each, random U(-1,1) actions"}
    sim.close()
    return out


def steady_ckpt_extra(ctx, B, K):
    """SURVEY.md 8(d) C2 'steady' variant: the reference's stored Ra=1e4 steady states (tests/golden/ckpt2d_ra10000.npz: three
    episodes of data/checkpoints/train/ckpt_ra10000.h5) tiled across the batch, zero action: throughput, and how far the
    kinetic energy and the Nusselt number move per env-step (the reference's 40 episodes scatter by 1e-6 / 3.5e-6 relative)."""
    torch, np, _native = ctx["torch"], ctx["np"], ctx["_native"]
    ck = np.load(os.path.join(ROOT, "tests", "golden", "ckpt2d_ra10000.npz"))
    n_ep = ck["b"].shape[0]
    idx = np.arange(B) % n_ep
    sim = _native.NativeSim(batch=B, device=ctx["local_rank"], ra=1e4)
    sim.reset_from_arrays(ck["b"][idx], ck["u"][idx], ck["w"][idx])

    def ke():
        b, u, w = sim.get_fields()
        return 0.5 * ((u ** 2).mean(axis=(1, 2)) + (w[:, :-1] ** 2).mean(axis=(1, 2)))
    ke0, nu0 = ke(), sim.get_nusselt()[0].copy()
    zero = torch.zeros((B, 12), device=ctx["dev"], dtype=torch.float32)
    torch.cuda.synchronize(ctx["dev"])
    e = timed_loop(torch, ctx["dev"], ctx["barrier"], lambda n: sim.step_dev(zero.data_ptr()), K)
    ke1, nu1 = ke(), sim.get_nusselt()[0]
    out = {"value": B * K / e, "unit": "env-steps/s", "ms_per_step": e / K * 1e3, "steps": K, "batch": B, "episodes_tiled": int(n_ep),
           "kinetic_energy": float(ke1.mean()), "reference_kinetic_energy": 0.0983448,
           "max_rel_ke_drift_per_env_step": float(np.abs(ke1 / ke0 - 1).max() / K),
           "max_rel_nusselt_drift_per_env_step": float(np.abs(nu1 / nu0 - 1).max() / K),
           "nusselt_state": float(nu1.mean()), "nan_envs": int(sim.get_flags().sum())}
    sim.close()
    return out


def gym_api_rate_3d(ctx, B, steps, obs_buffers, precision="f64"):
    """configs[4] through the gym API: B x VectorEnv.step() per wall second, host actions in, host observations (the float32
    states: 38 MB per step at B = 32) / rewards / infos out -- what the reference's SubprocVecEnv workers deliver to PPO."""
    np = ctx["np"]
    from rbc_gym.vector import RayleighBenardConvection3DVectorEnv
    venv = RayleighBenardConvection3DVectorEnv(num_envs=B, state_shape=(32, 48, 48), rayleigh_number=1e4, device=ctx["local_rank"],
                                               episode_length=10**9, obs_buffers=obs_buffers, precision=precision)
    venv.reset(seed=1234)
    acts = np.random.default_rng(5).uniform(-1, 1, (steps + 2, B, 8, 8)).astype(np.float32)
    venv.step(acts[0]); venv.step(acts[1])                           # both ping-pong parities of the captured graphs
    t0 = time.perf_counter()
    for n in range(steps):
        venv.step(acts[n + 2])
    dt = time.perf_counter() - t0
    venv.close()
    return B * steps / dt


def sync_step_ms_3d(ctx, B, steps, precision="f64"):
    """configs[4], one SYNCHRONOUS rbc_step (host actions in, wait, NaN flags back): what a gym-style caller pays per step of B envs,
    against `ms_per_step` of the back-to-back device-resident run (launch plan)"""
    np = ctx["np"]
    from rbc_gym import _native
    sim = _native.NativeSim3D(batch=B, device=ctx["local_rank"], shape=(32, 48, 48), ra=1e4, dt_control=0.125, dt_solver=0.01, precision=precision)
    sim.reset(np.arange(B, dtype=np.uint64) + 1234)
    acts = np.random.default_rng(5).uniform(-1, 1, (steps + 2, B, 8, 8)).astype(np.float32)
    sim.step(acts[0]); sim.step(acts[1])
    t0 = time.perf_counter()
    for n in range(steps):
        sim.step(acts[n + 2])
    dt = (time.perf_counter() - t0) / steps
    plan = sim.launch_plan()
    sim.close()
    return {"ms": dt * 1e3, "env_groups": plan[0], "own_graph_and_hardware_queue_per_group": bool(plan[1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="envs per GPU (default 1024; 32 with --dim 3)")
    ap.add_argument("--ra", type=float, default=1e4)
    ap.add_argument("--ra-sweep", type=str, default=None,
                    help="comma list, e.g. 1e4,1e5,1e6: BASELINE.json configs[3] stress case; the GLOBAL batch is split "
                         "into contiguous equal parts, one Rayleigh number each (fixed dt=0.03 as in the reference)")
    ap.add_argument("--precision", default="f64", choices=("f64", "f32"),
                    help="arithmetic of the 2D kernel; f64 (the reference's) is the headline, f32 the variant of SURVEY 8(d) C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra measurements (copy ceiling, gym-API rate, 3D, fp32)")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl (= RCCL) on a multi-GPU node; gloo only to rehearse the N>1 flow with several ranks sharing one GPU")
    ap.add_argument("--dim", type=int, default=2, choices=(2, 3),
                    help="3: BASELINE.json configs[4] (3D 32x48x48, Ra=1e4, 32 envs per GPU); the default line is the 2D configs[1]")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # the ranks print the line; this process never touches the GPU.  The CPU baseline runs HERE, before the ranks exist
        # (they would compete for the host cores), and travels to rank 0 through a file.
        tmp = None
        if not args.no_cpu_baseline:
            import tempfile
            fd, tmp = tempfile.mkstemp(prefix="rbc_cpu_baseline_", suffix=".json")
            with os.fdopen(fd, "w") as f:
                json.dump(cpu_baseline(dim=args.dim), f)
            os.environ["RBC_BENCH_CPU_BASELINE_FILE"] = tmp
        rc = self_launch(args)
        if tmp and os.path.exists(tmp):
            os.unlink(tmp)
        sys.exit(rc)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(dim=args.dim)
    elif rank == 0 and os.environ.get("RBC_BENCH_CPU_BASELINE_FILE"):      # measured by the self-launching parent
        try:
            cpu = json.load(open(os.environ["RBC_BENCH_CPU_BASELINE_FILE"]))
        except (OSError, ValueError):
            cpu = None

    import numpy as np
    import torch
    import torch.distributed as dist
    from rbc_gym import _native, sharding

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if args.dist_backend == "gloo":          # rehearsal: ranks may outnumber the GPUs of the box
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    red_dev = dev if args.dist_backend == "nccl" else None      # gloo reduces host tensors

    def barrier():
        sharding.barrier(dist if world > 1 else None)

    ctx = dict(torch=torch, np=np, _native=_native, sharding=sharding, dist=dist, dev=dev, red_dev=red_dev, rank=rank,
               local_rank=local_rank, world=world, barrier=barrier)
    K, W = args.steps, args.warmup
    if args.dim == 3:
        out = run_3d(ctx, args.batch or 32, K, W, args.ra, args.precision)
        if rank == 0:
            out["cpu_baseline"] = cpu
            print(json.dumps(out))
        if world > 1:
            dist.destroy_process_group()
        return

    B = args.batch or 1024
    f32 = args.precision == "f32"
    sim = _native.NativeSim(batch=B, device=local_rank, ra=args.ra, precision=1 if f32 else 0)
    # the simulation runs on its own (non-blocking) stream; everything torch enqueues for it is completed with a device
    # synchronisation before the sim reads it, and the timed region is bracketed by device synchronisations
    start, count = sharding.shard(world * B, world, rank)      # weak scaling: B envs on every GPU
    assert count == B
    ra_env = None
    if args.ra_sweep:
        ras = [float(x) for x in args.ra_sweep.split(",")]
        ra_env = np.array([ras[min(len(ras) - 1, (start + e) * len(ras) // (world * B))] for e in range(B)])
        sim.set_rayleigh(ra_env)
    sim.reset(sharding.env_seeds(1234, start, count))
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)
    actions = (torch.rand((K + W, B, 12), device=dev, generator=gen, dtype=torch.float32) * 2 - 1).contiguous()
    torch.cuda.synchronize(dev)
    stride = B * 12 * 4

    for n in range(W):
        sim.step_dev(actions.data_ptr() + n * stride)
    sim.set_profiling(K)
    elapsed = timed_loop(torch, dev, barrier, lambda n: sim.step_dev(actions.data_ptr() + (W + n) * stride), K)

    kern_ms = sim.profile_read(K)
    flags = sim.get_flags()
    nan_envs = int(flags.sum())
    nus, _ = sim.get_nusselt()
    rank_elapsed, rank_nan = sharding.gather_run(elapsed, nan_envs, device=red_dev, dist=dist if world > 1 else None)
    elapsed, nan_total = sharding.reduce_run(elapsed, nan_envs, device=red_dev, dist=dist if world > 1 else None)
    alg_bytes = sim.algorithmic_bytes_per_env_step() * B          # per launch
    sim.close()

    if rank == 0:
        avg_ms = float(np.mean(kern_ms)) if len(kern_ms) else float("nan")
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms == avg_ms else None
        # counters from the separate rocprofv3 --pmc passes, attached only when they were taken on THIS workload
        key = {"dim": 2, "batch": B, "ra": args.ra, "precision": args.precision, "ra_sweep": args.ra_sweep}
        prof, prof_src = latest_profile("*_summary.json", key)
        traffic = prof.get("hbm_traffic_bytes_per_launch") if prof else None
        sq, sq_src = latest_profile("*_sq_counters.json", key)
        valu = None
        if sq:
            pl = sq["per_launch_mean"]
            n64 = pl.get("SQ_INSTS_VALU_ADD_F64", 0) + pl.get("SQ_INSTS_VALU_FMA_F64", 0) + pl.get("SQ_INSTS_VALU_MUL_F64", 0)
            n_all = pl.get("SQ_INSTS_VALU", 0)
            flop = 64 * (pl.get("SQ_INSTS_VALU_ADD_F64", 0) + 2 * pl.get("SQ_INSTS_VALU_FMA_F64", 0) + pl.get("SQ_INSTS_VALU_MUL_F64", 0))
            rate = n64 / (avg_ms * 1e-3)
            valu = {"bound": "valu_fp64_issue", "achieved": rate, "peak": FP64_WAVE_INST_PEAK, "unit": "fp64 wave64-instructions/s",
                    "frac": rate / FP64_WAVE_INST_PEAK, "fp64_wave_instructions_per_launch": n64, "valu_wave_instructions_per_launch": n_all,
                    "all_valu_issue_frac": (n_all / (avg_ms * 1e-3)) / FP64_WAVE_INST_PEAK if n_all else None,
                    "achieved_tflops_fp64": flop / (avg_ms * 1e-3) / 1e12, "peak_tflops_fp64_vector": 78.6,
                    "source": sq_src,
                    "note": "instruction counts per launch are constants of the build (SQ_INSTS_VALU_{ADD,FMA,MUL}_F64, separate "
                            "rocprofv3 --pmc passes); peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 fp64 instruction. This is the "
                            "limiter that binds: the state never leaves LDS, HBM carries ~14 % of the algorithmic bytes"}
        out = {
            "metric": "env-steps/sec (batched 2D RBC 64x96 Ra=1e4)",
            "value": world * B * K / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {"workload": f"configs[1]: batched 2D envs, batch={B} per GPU, grid 64x96, Ra={args.ra:g}, "
                                   "heater_duration 1.5 = 50 RK3 substeps (150 stages) per env-step, dt 0.03",
                       "global_batch": world * B, "substeps_per_env_step": 50,
                       "parallelism": f"env-sharded x{world} (no collective on the step path)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (FETCH_SIZE+WRITE_SIZE)*1024, separate rocprofv3 --pmc passes",
                         "traffic_source": prof_src,
                         "measured_hbm_gbs": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
                         "measured_hbm_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "kernel": f"rbc2d_kernel<96,64,{'float2 pairs' if f32 else 'double'},DBG=false>", "kernel_ms_avg": avg_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "achieved/frac follow SURVEY.md 8(d)'s convention (algorithmic bytes = 10*F*C*s per RK3 substep, F=3, "
                                 "C=6144, x 50 x batch, / kernel time): a CONVENTION number for this kernel, which keeps the state in LDS "
                                 "for the whole control interval and is NOT HBM-bound -- measured_hbm_* is what the memory system "
                                 "actually carries (state in/out + G^- tendencies parked through L2), roofline_valu the limiter that binds"},
            "roofline_valu": valu,
            "cpu_baseline": cpu,
            "nan_envs": nan_total,
            "per_rank": {"ms_per_step": [e / K * 1e3 for e in rank_elapsed], "nan_envs": rank_nan,
                         "note": "value uses the slowest rank (max over ranks); a straggler GPU shows here"},
            "ra_sweep": ({"values": [float(x) for x in args.ra_sweep.split(",")],
                          "rank0_nan_envs_per_ra": {f"{r:g}": int(flags[ra_env == r].sum()) for r in sorted(set(ra_env))},
                          "rank0_mean_nusselt_per_ra": {f"{r:g}": float(np.mean(nus[ra_env == r])) for r in sorted(set(ra_env))}}
                         if args.ra_sweep else None),
            "mean_nusselt_state": float(np.mean(nus)),
        }
        if world == 1 and not args.no_extra:
            extra = {}
            try:
                extra["copy_ceiling"] = _native.copy_ceiling(local_rank)
                out["roofline"]["on_box_copy_gbs"] = extra["copy_ceiling"]["kernel_gbs"]
                extra["gym_api_env_steps_per_s"] = {
                    "no_info_state": gym_api_rate(ctx, B, 5, False), "default_info_state": gym_api_rate(ctx, B, 5, True),
                    "always_new_arrays": gym_api_rate(ctx, B, 5, "fresh"),
                    "note": "B x VectorEnv.step() per wall second, host actions in, host obs/reward/info out; 5 steps after 2 warm-up steps. "
                            "default_info_state: info['state'] (75 MB of float32 per step) in pooled page-locked arrays that are reused once the "
                            "caller has dropped them; always_new_arrays: np.empty every step"}
                d3 = run_3d(ctx, 32, 40, 5, 1e4)
                extra["config4_3d"] = {k: d3[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline",
                                                          "nan_envs", "mean_nusselt")}
                extra["gym_api_3d_env_steps_per_s"] = {
                    "default": gym_api_rate_3d(ctx, 32, 8, None), "always_new_arrays": gym_api_rate_3d(ctx, 32, 8, "fresh"),
                    "default_fp32": gym_api_rate_3d(ctx, 32, 8, None, "f32") if _native.has_precision("f32") else None,
                    "sync_step": {"f64": sync_step_ms_3d(ctx, 32, 10), "f32": sync_step_ms_3d(ctx, 32, 10, "f32") if _native.has_precision("f32") else None},
                    "note": "configs[4], B x VectorEnv.step() per wall second with the 38 MB of float32 observations crossing PCIe every step; "
                            "8 steps after 2 warm-up steps. default: pooled page-locked observation arrays, reused once the caller has dropped them"}
                if _native.has_precision("f32"):           # the float32 instantiation of the same 3D kernels (153 MB algorithmic per env-step)
                    d3f = run_3d(ctx, 32, 40, 5, 1e4, "f32")
                    extra["config4_3d_fp32"] = {k: d3f[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config",
                                                                    "roofline", "nan_envs", "mean_nusselt")}
                if not f32 and _native.has_precision("f32"):
                    s32 = _native.NativeSim(batch=B, device=local_rank, ra=args.ra, precision=1)
                    s32.reset(sharding.env_seeds(1234, 0, B))
                    for n in range(2):
                        s32.step_dev(actions.data_ptr() + n * stride)
                    k32 = min(K, 10)
                    s32.set_profiling(k32)
                    e32 = timed_loop(torch, dev, barrier, lambda n: s32.step_dev(actions.data_ptr() + (W + n) * stride), k32)
                    ms32 = float(np.mean(s32.profile_read(k32)))
                    alg32 = s32.algorithmic_bytes_per_env_step() * B
                    extra["fp32_variant"] = {"value": B * k32 / e32, "unit": "env-steps/s", "ms_per_step": e32 / k32 * 1e3, "kernel_ms_avg": ms32,
                                             "roofline_frac_vs_36.86MB_per_env_step": alg32 / (ms32 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "nan_envs": int(s32.get_flags().sum()), "mean_nusselt_state": float(np.mean(s32.get_nusselt()[0])),
                                             "mean_nusselt_state_note": f"after 2 + {k32} env-steps from the same seeds (the f64 line's value is after "
                                                                        f"{W} + {K}): a different time of the transient, not a parity signal -- parity of "
                                                                        "the float32 variant is tests/test_gpu_parity.py"}
                    s32.close()
                extra["ra_sweep_config3"] = ra_sweep_extra(ctx, B, 50, 10)
                extra["steady_ckpt"] = steady_ckpt_extra(ctx, B, 10)
                # a 2D grid the LDS-resident kernel has no instantiation for: the streaming path
                # (at Ra = 1e5: with Ra = 1e4 this finer grid leaves the explicit scheme's stability range at dt = 0.03 once the flow is up --
                #  4 kappa dt (1/dx^2 + 1/dz^2) = 2.06 of RK3's 2.51 before advection is added, 1.80 on the default grid -- and the oracle
                #  blows up alike after 8 actuated env-steps; the work per env-step does not depend on Ra)
                s2 = _native.NativeSim(batch=B, device=local_rank, ra=1e5, nx=128, nz=64, obs_nx=64, obs_nz=8)
                s2.reset(sharding.env_seeds(1234, 0, B))
                s2.step_dev(actions.data_ptr())
                K2 = 10
                e2 = timed_loop(torch, dev, barrier, lambda n: s2.step_dev(actions.data_ptr() + ((1 + n) % (K + W)) * stride), K2)
                alg2 = s2.algorithmic_bytes_per_env_step() * B
                p2, p2src = latest_profile("*_stream2d_128x64_summary.json", {"dim": 2, "batch": B, "nx": 128, "nz": 64, "precision": "f64"})
                tr2 = p2.get("hbm_traffic_bytes_per_env_step_batch") if p2 else None
                extra["streaming_2d_128x64"] = {"value": B * K2 / e2, "unit": "env-steps/s", "ms_per_step": e2 / K2 * 1e3, "steps": K2, "warmup": 1,
                                                "algorithmic_gbs": alg2 * K2 / e2 / 1e9,
                                                "roofline": {"bound": "hbm", "achieved": alg2 * K2 / e2 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                             "frac": alg2 * K2 / e2 / 1e9 / HBM_PEAK_GBS, "traffic": tr2, "traffic_source": p2src,
                                                             "traffic_over_algorithmic": (tr2 / alg2) if tr2 else None,
                                                             "measured_hbm_gbs": (tr2 * K2 / e2 / 1e9) if tr2 else None,
                                                             "algorithmic_bytes_per_launch": alg2,
                                                             "note": "one 'launch' = one env-step of the batch (150 stages x [tile kernel + one-kernel projection] on three stream chains)"},
                                                "rayleigh_number": 1e5, "nan_envs": int(s2.get_flags().sum()), "mean_nusselt_state": float(np.mean(s2.get_nusselt()[0]))}
                s2.close()
            except Exception as e:                       # an extra must never take the contract line down with it
                extra["error"] = f"{type(e).__name__}: {e}"
            out["extra"] = extra
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
