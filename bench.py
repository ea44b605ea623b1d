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
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rbc-gym_amd"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle ("port"), one env per process, on the box's host cores.  Runs BEFORE
# anything touches the GPU (fork-based pool).
# ---------------------------------------------------------------------------------------------
def _cpu_worker(args):
    idx, nsteps = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_py
    rng = np.random.default_rng(1234 + idx)
    sim = oracle_py.OracleSim(ra=1e4)
    sim.reset_random(1234 + idx)
    sim.step(rng.uniform(-1, 1, 12).astype(np.float32))          # warm-up
    t0 = time.perf_counter()
    for _ in range(nsteps):
        sim.step(rng.uniform(-1, 1, 12).astype(np.float32))
    return time.perf_counter() - t0


def cpu_baseline(nsteps=6):
    import multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_py
    oracle_py.build_oracle()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        per = pool.map(_cpu_worker, [(i, nsteps) for i in range(cores)])
    wall = time.perf_counter() - t0
    stepping = max(per)
    return {"value": cores * nsteps / stepping, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "per_core": nsteps / (sum(per) / len(per)),
            "sample": f"{cores} envs (one per core) x {nsteps} env-steps of the bench workload (64x96, Ra=1e4, "
                      f"50 RK3 substeps, random ICs seeds 1234+i, U(-1,1) actions) on the C oracle; "
                      f"{stepping:.1f} s stepping, {wall:.1f} s wall. Julia reference unavailable on this box "
                      f"(README.md:62 publishes 0.12 s/step at 34 substeps on one Apple-silicon core)"}


def bench3d(args, rank, local_rank, world, dev, red_dev, dist, sharding, _native, np, torch):
    """configs[4]: 3D 32x48x48, Ra=1e4, heater_duration 0.125, dt_solver 0.01 (13 RK3 substeps per env-step)."""
    B = args.batch if args.batch != 1024 else 32
    K, W = args.steps, args.warmup
    sim = _native.NativeSim3D(batch=B, device=local_rank, shape=(32, 48, 48), ra=args.ra)
    stream = torch.cuda.current_stream(dev)
    sim.lib.rbc_set_stream(sim.h, stream.cuda_stream)
    start, count = sharding.shard(world * B, world, rank)
    sim.reset(sharding.env_seeds(1234, start, count))
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)
    actions = (torch.rand((K + W, B, 8, 8), device=dev, generator=gen, dtype=torch.float32) * 2 - 1).contiguous()
    stride = B * 64 * 4
    for n in range(W):
        sim.step_dev(actions.data_ptr() + n * stride)
    sim.set_profiling(K)
    torch.cuda.synchronize(dev)
    sharding.barrier(dist if world > 1 else None)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for n in range(K):
        sim.step_dev(actions.data_ptr() + (W + n) * stride)
    torch.cuda.synchronize(dev)
    sharding.barrier(dist if world > 1 else None)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    ms = sim.profile_read(K)
    nan_envs = int(sim.get_flags().sum())
    nu = sim.get_nusselt()
    elapsed, nan_total = sharding.reduce_run(elapsed, nan_envs, device=red_dev, dist=dist if world > 1 else None)
    if rank == 0:
        alg = sim.algorithmic_bytes_per_env_step() * B
        avg = float(np.mean(ms)) if len(ms) else float("nan")
        ach = alg / (avg * 1e-3) / 1e9
        print(json.dumps({
            "metric": "env-steps/sec (batched 3D RBC 32x48x48 Ra=1e4)", "value": world * B * K / elapsed, "unit": "env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[4]: batched 3D envs, batch={B} per GPU, grid 32x48x48, Ra={args.ra:g}, heater_duration 0.125 "
                                   "(x t_ff=4) = 13 RK3 substeps per env-step, dt_solver 0.01", "global_batch": world * B,
                       "parallelism": f"env-sharded x{world} (no collective on the step path)"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "rbc3 stage sequence (hydrostatic, 4x tendency, slab FFT, z sweep, inverse FFT, correct) x 39",
                         "kernel_ms_avg": avg, "algorithmic_bytes_per_launch": alg,
                         "note": "algorithmic bytes = 10*F*C*s per substep (F=4, C=73728, s=8) x 13 x batch; 'launch' = one env-step of the batch"},
            "cpu_baseline": None, "nan_envs": nan_total, "mean_nusselt": float(np.mean(nu))}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="envs per GPU")
    ap.add_argument("--ra", type=float, default=1e4)
    ap.add_argument("--ra-sweep", type=str, default=None,
                    help="comma list, e.g. 1e4,1e5,1e6: BASELINE.json configs[3] stress case; the GLOBAL batch is split "
                         "into contiguous equal parts, one Rayleigh number each (fixed dt=0.03 as in the reference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="nccl (= RCCL) on a multi-GPU node; gloo only to rehearse the N>1 flow with several ranks sharing one GPU")
    ap.add_argument("--dim", type=int, default=2, choices=(2, 3),
                    help="3: BASELINE.json configs[4] (3D 32x48x48, Ra=1e4, 32 envs per GPU); the default line is the 2D configs[1]")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

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

    B, K, W = args.batch, args.steps, args.warmup
    if args.dim == 3:
        return bench3d(args, rank, local_rank, world, dev, red_dev, dist, sharding, _native, np, torch)
    sim = _native.NativeSim(batch=B, device=local_rank, ra=args.ra)
    stream = torch.cuda.current_stream(dev)
    sim.lib.rbc_set_stream(sim.h, stream.cuda_stream)
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
    stride = B * 12 * 4

    def barrier():
        sharding.barrier(dist if world > 1 else None)

    for n in range(W):
        sim.step_dev(actions.data_ptr() + n * stride)
    sim.set_profiling(K)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for n in range(K):
        sim.step_dev(actions.data_ptr() + (W + n) * stride)
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    kern_ms = sim.profile_read(K)
    flags = sim.get_flags()
    nan_envs = int(flags.sum())
    nus, _ = sim.get_nusselt()

    elapsed, nan_total = sharding.reduce_run(elapsed, nan_envs, device=red_dev, dist=dist if world > 1 else None)

    if rank == 0:
        traffic, traffic_src = None, None
        try:                       # HBM bytes per launch from the separate rocprofv3 --pmc passes (profiles/)
            import glob
            latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json")))[-1]
            summ = json.load(open(latest))
            if summ.get("bench_line_under_profiler", {}).get("config", {}).get("global_batch") == world * B or world == 1:
                traffic, traffic_src = summ.get("hbm_traffic_bytes_per_launch"), os.path.relpath(latest, ROOT)
        except Exception:
            pass
        alg_bytes = sim.algorithmic_bytes_per_env_step() * B          # per launch
        avg_ms = float(np.mean(kern_ms)) if len(kern_ms) else float("nan")
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms == avg_ms else None
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
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"configs[1]: batched 2D envs, batch={B} per GPU, grid 64x96, Ra={args.ra:g}, "
                                   "heater_duration 1.5 = 50 RK3 substeps (150 stages) per env-step, dt 0.03",
                       "global_batch": world * B, "substeps_per_env_step": 50,
                       "parallelism": f"env-sharded x{world} (no collective on the step path)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (FETCH_SIZE+WRITE_SIZE)*1024, separate rocprofv3 --pmc passes",
                         "traffic_source": traffic_src,
                         "kernel": "rbc2d_kernel<96,64>", "kernel_ms_avg": avg_ms,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "algorithmic bytes = 10*F*C*s per RK3 substep (F=3,C=6144,s=8) x 50 x batch; the kernel "
                                 "keeps the state in LDS for the whole control interval: real HBM traffic is ~0.4 MB/env of state and outputs "
                                 "plus ~10 MB/env of G^- tendencies parked through L2 (write-through stores, reads served by L2)"},
            "cpu_baseline": cpu,
            "nan_envs": nan_total,
            "ra_sweep": ({"values": [float(x) for x in args.ra_sweep.split(",")],
                          "rank0_nan_envs_per_ra": {f"{r:g}": int(flags[ra_env == r].sum()) for r in sorted(set(ra_env))},
                          "rank0_mean_nusselt_per_ra": {f"{r:g}": float(np.mean(nus[ra_env == r])) for r in sorted(set(ra_env))}}
                         if args.ra_sweep else None),
            "mean_nusselt_state": float(np.mean(nus)),
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
