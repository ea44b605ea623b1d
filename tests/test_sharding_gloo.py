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
    per_rank = sharding.gather_run(1.0 + rank, rank, dist=dist)          # ... and the per-rank lists of the N>1 bench line
    assert per_rank == ([1.0, 2.0], [0, 1]), per_rank
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


def _worker8(rank, world, port, q):
    """the reductions of the N = 8 bench line with nothing but the host: shards of BASELINE.json configs[2] (8192 envs over 8 GPUs)"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = sharding.shard(8192, world, rank)
    seeds = sharding.env_seeds(1234, start, count)
    sharding.barrier(dist)
    elapsed, nans = sharding.reduce_run(0.5 + 0.01 * rank, rank % 2, dist=dist)
    per_rank = sharding.gather_run(0.5 + 0.01 * rank, rank % 2, dist=dist)
    q.put((rank, start, count, int(seeds[0]), int(seeds[-1]), elapsed, nans, per_rank))
    dist.destroy_process_group()


def test_eight_rank_reductions_of_the_scale_line():
    """world_size 8 over gloo on the CPU: the shards of configs[2] tile the 8192 envs, seeds run on across the shard boundaries,
    `value` would use the slowest rank, `per_rank` lists all eight.  (On the GPU box at most six processes may touch the card,
    so the GPU rehearsal of the self-launching bench, tests/test_bench_contract.py, runs four ranks next to the test runner; the
    eight-rank case is covered here without a GPU and by the driver on a real node.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    nxt = 0
    for rank, start, count, s0, s1, elapsed, nans, per_rank in got:
        assert (start, count) == (nxt, 1024) and (s0, s1) == (1234 + start, 1234 + start + 1023)
        nxt += count
        assert abs(elapsed - 0.57) < 1e-12 and nans == 4
        assert np.allclose(per_rank[0], [0.5 + 0.01 * r for r in range(8)]) and per_rank[1] == [0, 1] * 4
    assert nxt == 8192


class _FakeSim:
    """stand-in for NativeSim in the host-logic test of rbc_gym.sharded.ShardedSim: env e of the global batch holds the
    value `100 * device + local index`, a step adds the env's action sum; records the thread it was driven from"""

    def __init__(self, count, dev):
        import threading
        self.B, self.dev, self.heaters, self.lib = count, dev, 12, None
        self.x = 100.0 * dev + np.arange(count, dtype=np.float64)
        self.seeds = None
        self.threads = set()
        self._t = threading

    def reset(self, seeds, mask=None):
        self.threads.add(self._t.get_ident())
        m = np.ones(self.B, bool) if mask is None else np.asarray(mask, bool)
        assert seeds.shape == (self.B,) and m.shape == (self.B,)
        self.seeds = np.where(m, seeds, 0 if self.seeds is None else self.seeds)
        self.x = np.where(m, 100.0 * self.dev + np.arange(self.B), self.x)

    def step(self, a):
        self.threads.add(self._t.get_ident())
        assert a.shape == (self.B, 12) and a.dtype == np.float32
        self.x = self.x + a.sum(1)
        return True

    def get_obs(self, nch=3):
        return np.repeat(self.x[:, None], nch, 1)

    def get_nusselt(self):
        return self.x.copy(), -self.x

    def get_flags(self):
        return np.zeros(self.B, np.int32)

    def close(self):
        pass


def test_sharded_sim_concatenates_in_global_env_order():
    """rbc_gym.sharded.ShardedSim (the multi-GPU VectorEnv's engine) with three fake device handles: contiguous env
    ranges, per-shard slices of seeds / masks / actions, results back in global order, one driver thread per handle."""
    from rbc_gym.sharded import ShardedSim
    made = []

    def make(count, dev):
        made.append(_FakeSim(count, dev))
        return made[-1]
    sh = ShardedSim(make, 10, [0, 1, 2])
    assert sh.ranges == [(0, 4), (4, 3), (7, 3)] and [m.B for m in made] == [4, 3, 3]
    sh.reset(np.arange(10, dtype=np.uint64) + 50)
    assert [list(m.seeds) for m in made] == [[50, 51, 52, 53], [54, 55, 56], [57, 58, 59]]
    a = np.zeros((10, 12), np.float32)
    a[:, 0] = np.arange(10) / 10
    assert sh.step(a)
    nus, nuo = sh.get_nusselt()
    want = np.concatenate([100.0 * d + np.arange(c) for d, c in ((0, 4), (1, 3), (2, 3))]) + np.arange(10, dtype=np.float32) / 10
    assert np.allclose(nus, want) and np.allclose(nuo, -want) and sh.get_obs(5).shape == (10, 5)
    mask = np.zeros(10, np.uint8)
    mask[[3, 4]] = 1                                      # straddles the first shard boundary; shard 2 is not touched at all
    before = made[2].x.copy()
    sh.reset(np.arange(10, dtype=np.uint64) + 900, mask=mask)
    assert made[0].seeds[3] == 903 and made[1].seeds[0] == 904 and made[0].seeds[0] == 50 and np.array_equal(made[2].x, before)
    assert made[0].x[3] == 3.0 and made[1].x[0] == 100.0 and abs(made[0].x[1] - 1.1) < 1e-6      # env 1 kept its stepped value
    assert len(set().union(*[m.threads for m in made])) >= 1
    with pytest.raises(ValueError):
        sh.step(np.zeros((9, 12), np.float32))
    with pytest.raises(ValueError):
        ShardedSim(make, 2, [0, 1, 2])
    sh.close()


def test_sharded_sim_marks_itself_broken_after_a_failed_mutating_call():
    """A step that fails on one shard has still advanced the others (clocks and autoreset masks out of step across the
    batch): the sharded env then refuses every further call but close().  Host logic only: fake handles, no GPU."""
    import numpy as np
    from rbc_gym.sharded import ShardedSim

    class Fake:
        lib, heaters = None, 12

        def __init__(self, count, dev):
            self.count, self.dev, self.steps, self.closed = count, dev, 0, False

        def step(self, a):
            if self.dev == 1 and self.steps == 1:
                raise RuntimeError("device lost")
            self.steps += 1
            return True

        def get_flags(self):
            return np.zeros(self.count, np.int32)

        def set_obs_normalization(self, lo, hi):
            if len(lo) != len(hi):
                raise ValueError("min_vals and max_vals must be of equal length")      # refused before anything is mutated

        def close(self):
            self.closed = True

    s = ShardedSim(Fake, 5, devices=[0, 1])
    acts = np.zeros((5, 12), np.float32)
    assert s.step(acts) is True
    with pytest.raises(ValueError):                                     # every shard refuses the same bad argument: nothing diverged
        s.set_obs_normalization([0, 0, 0], [1, 1])
    assert s.get_flags().shape == (5,)                                  # ... so the env stays usable
    with pytest.raises(RuntimeError, match="device lost"):
        s.step(acts)
    assert [f.steps for f in s.sims] == [2, 1]                         # the healthy shard went ahead: the batch is inconsistent
    with pytest.raises(RuntimeError, match="broken"):
        s.get_flags()
    with pytest.raises(RuntimeError, match="broken"):
        s.step(acts)
    sims = list(s.sims)
    s.close()
    assert all(f.closed for f in sims)
