"""Multi-GPU: env instances are independent, so N GPUs = N contiguous shards of the global env
batch, one process per GPU, and NO collective on the step path.  torch.distributed
(RCCL on the GPU box, gloo in the CPU tests) is only used to line the ranks up around the timed
region and to reduce the timing / NaN counters."""
import numpy as np


def shard(global_batch, world, rank):
    """contiguous env range [start, start+count) of `rank`; the first global_batch % world ranks get one extra env."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    base, extra = divmod(int(global_batch), int(world))
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def env_seeds(base_seed, start, count):
    """per-env seeds of the synthetic workload: base_seed + global env index."""
    return np.uint64(base_seed) + np.arange(start, start + count, dtype=np.uint64)


def barrier(dist=None):
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def reduce_run(elapsed, nan_envs, device=None, dist=None):
    """-> (max elapsed over ranks, total NaN envs) ; identity when not distributed."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed), int(nan_envs)
    import torch
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    n = torch.tensor([int(nan_envs)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())


def gather_run(elapsed, nan_envs, device=None, dist=None):
    """-> (per-rank elapsed seconds, per-rank NaN-env counts), rank-major lists: a straggler GPU shows up here while
    `reduce_run` only keeps the maximum.  Identity (one-element lists) when not distributed."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(elapsed)], [int(nan_envs)]
    import torch
    world = dist.get_world_size()
    mine = torch.tensor([float(elapsed), float(nan_envs)], dtype=torch.float64, device=device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = torch.stack(parts).cpu()
    return [float(x) for x in out[:, 0]], [int(round(float(x))) for x in out[:, 1]]


def gather_observations(local_obs, dist=None):
    """Optional exchange step when the policy lives on one GPU: all-gather of the per-rank
    observation batches, rank-major, i.e. in global env order for contiguous equal shards.  `local_obs` is a torch
    tensor -- e.g. `torch.as_tensor(venv.device_views()["obs"], device="cuda")`, zero-copy over the library's
    buffer, so on the GPU box this is one RCCL all-gather over xGMI (8192 envs x 4.6 KB = 38 MB per step) with
    no host hop.  Not on the step path: the solver itself never communicates."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local_obs
    world = dist.get_world_size()
    local = local_obs.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out
