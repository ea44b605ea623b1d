"""CPU, world_size 2 over gloo: the N>1 path of bench.py (env sharding, per-env seeds, timing /
NaN reduction).  The stepper stand-in on CPU is the oracle (tests may use it): two ranks stepping
their shards must reproduce a single process stepping all envs, env for env."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rbc_gym import sharding


def test_shards_partition_the_batch():
    for gb, w in ((8192, 8), (1024, 1), (10, 4), (7, 8)):
        seen = []
        for r in range(w):
            s, c = sharding.shard(gb, w, r)
            seen += list(range(s, s + c))
        assert seen == list(range(gb))
    assert np.array_equal(sharding.env_seeds(1234, 1024, 3), np.array([2258, 2259, 2260], dtype=np.uint64))
    with pytest.raises(ValueError):
        sharding.shard(8, 2, 2)


def _step_envs(ids):
    import oracle_py
    out = {}
    for i in ids:
        s = oracle_py.OracleSim(ra=1e4, dt_control=0.06)
        s.reset_random(int(sharding.env_seeds(1234, i, 1)[0]))
        a = np.random.default_rng(100 + i).uniform(-1, 1, 12).astype(np.float32)
        assert s.step(a)
        out[i] = (s.nusselt(True), s.kinetic_energy())
    return out


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = sharding.shard(4, world, rank)
    sharding.barrier(dist)
    res = _step_envs(range(start, start + count))
    sharding.barrier(dist)
    elapsed, nans = sharding.reduce_run(1.0 + rank, rank, dist=dist)     # rank-dependent inputs: MAX / SUM visible
    import torch
    local = torch.arange(start, start + count, dtype=torch.float32).reshape(count, 1, 1).expand(count, 3, 2).contiguous()
    obs = sharding.gather_observations(local, dist)                       # optional policy-side exchange
    q.put((rank, res, elapsed, nans, obs[:, 0, 0].tolist()))
    dist.destroy_process_group()


def test_two_ranks_reproduce_one_process():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    merged = {}
    for rank, res, elapsed, nans, order in got:
        assert elapsed == 2.0 and nans == 1                 # max over ranks, sum over ranks
        assert order == [0.0, 1.0, 2.0, 3.0]                # gathered observations arrive in global env order
        merged.update(res)
    serial = _step_envs(range(4))
    assert sorted(merged) == [0, 1, 2, 3]
    for i in range(4):
        assert merged[i] == serial[i]                       # bitwise: no cross-env coupling anywhere
