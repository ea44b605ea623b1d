"""One vectorised env over several GPUs of a node (BASELINE.json north_star: "independent env instances sharded across the
8 GPUs").  The reference's user entry is `gym.make_vec(id, num_envs=N)` (example/run_vectorized.py:11-20), which fans out
to N processes; here `devices=[0, 1, ...]` splits the N envs into contiguous ranges (sharding.shard), one library handle
per device, and every call is issued to all handles at once from one thread per handle -- ctypes drops the GIL for the
duration of a foreign call and the C ABI allows different handles to be driven from different threads
(include/rbc_hip.h, "Threading").  Results are concatenated in global env order, so the sharded env is
indistinguishable from a single-handle one: same seeds (s + i across the shard boundaries), same autoreset masks, same
arrays.  No collective: env instances never communicate.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from .sharding import shard


class ShardedSim:
    """A list of NativeSim / NativeSim3D handles behind the single-handle interface the env layer uses."""

    def __init__(self, make_sim, num_envs, devices):
        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one GPU")
        if num_envs < len(devices):
            raise ValueError(f"{num_envs} envs cannot be sharded over {len(devices)} devices")
        self.devices = devices
        self.ranges = [shard(num_envs, len(devices), r) for r in range(len(devices))]       # (start, count) per shard
        self.sims = []
        try:
            for dev, (_, count) in zip(devices, self.ranges):
                self.sims.append(make_sim(count, dev))
        except Exception:
            self.close()
            raise
        self.B = int(num_envs)
        self._broken = None
        self._pool = ThreadPoolExecutor(max_workers=len(devices), thread_name_prefix="rbc-shard")
        first = self.sims[0]
        self.lib, self.heaters = first.lib, first.heaters
        for name in ("nx", "ny", "nz", "obs_shape"):
            if hasattr(first, name):
                setattr(self, name, getattr(first, name))

    # -- plumbing ------------------------------------------------------------------------------------
    def _slice(self, r, a):
        s, c = self.ranges[r]
        return a[s:s + c]

    def _all(self, fn, mutating=False):
        """fn(r, sim) on every shard concurrently; results in shard order; the first exception is re-raised.
        A MUTATING call (reset / step / set_*) that fails on one shard has still run on the others: their clocks,
        states and autoreset masks are then out of step with the failed one, so the sharded env marks itself broken and
        refuses every further call but close() -- build a new env (the reference's per-process envs die the same way:
        a RuntimeError out of one worker's step ends the AsyncVectorEnv)."""
        if self._broken is not None:
            raise RuntimeError(f"sharded env is broken after a failed call on one shard ({self._broken}); close it and make a new one")
        futs = [self._pool.submit(fn, r, sim) for r, sim in enumerate(self.sims)]
        out, first, failed = [], None, []
        for r, f in enumerate(futs):
            try:
                out.append(f.result())
            except Exception as e:                      # collect every shard before raising: no call is left in flight
                out.append(None)
                failed.append(e)
                if first is None:
                    first = (r, e)
        if first is not None:
            # An argument check that every shard refuses alike (ValueError / TypeError raised before anything is mutated: a
            # wrong action shape, a wrong normalisation vector) leaves all shards where they were: the env stays usable.
            uniform_refusal = len(failed) == len(self.sims) and all(isinstance(e, (ValueError, TypeError)) for e in failed)
            if mutating and not uniform_refusal:
                self._broken = f"shard {first[0]} on device {self.devices[first[0]]}: {first[1]}"
            raise first[1]
        return out

    def _split(self, a, dtype=None):
        if a is None:
            return [None] * len(self.sims)
        a = np.asarray(a) if dtype is None else np.asarray(a, dtype=dtype)
        if a.shape[0] != self.B:
            raise ValueError(f"leading dimension must be num_envs={self.B}, got {a.shape}")
        return [self._slice(r, a) for r in range(len(self.sims))]

    # -- initialize_simulation -------------------------------------------------------------------------
    def reset(self, seeds, mask=None):
        s = self._split(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (self.B,)))
        m = self._split(mask)
        self._all(lambda r, sim: None if (m[r] is not None and not np.any(m[r])) else sim.reset(s[r], mask=m[r]), mutating=True)

    def reset_from_arrays(self, *fields, mask=None):
        f = [self._split(x) for x in fields]
        m = self._split(mask)
        self._all(lambda r, sim: None if (m[r] is not None and not np.any(m[r])) else sim.reset_from_arrays(*[x[r] for x in f], mask=m[r]), mutating=True)

    def set_rayleigh(self, ra):
        v = self._split(np.broadcast_to(np.asarray(ra, np.float64), (self.B,)))
        self._all(lambda r, sim: sim.set_rayleigh(v[r]), mutating=True)

    def set_obs_normalization(self, *a, **kw):
        self._all(lambda r, sim: sim.set_obs_normalization(*a, **kw), mutating=True)

    # -- step_simulation ---------------------------------------------------------------------------------
    def step(self, actions):
        a = self._split(actions, np.float32)
        return all(self._all(lambda r, sim: sim.step(a[r]), mutating=True))

    def step_dev(self, actions_dev_ptrs):
        """one device pointer per shard (each on that shard's GPU), float32 [count][heaters...]"""
        ptrs = list(actions_dev_ptrs)
        if len(ptrs) != len(self.sims):
            raise ValueError("step_dev on a sharded env takes one device pointer per shard")
        self._all(lambda r, sim: sim.step_dev(ptrs[r]), mutating=True)

    # -- getters ---------------------------------------------------------------------------------------------
    def _cat(self, parts):
        if isinstance(parts[0], tuple):
            return tuple(np.concatenate([p[i] for p in parts]) for i in range(len(parts[0])))
        return np.concatenate(parts)

    def get_obs(self, *a):
        return self._cat(self._all(lambda r, sim: sim.get_obs(*a)))

    def get_state(self, *a, out=None):                       # (2D: nch positional; 3D: no positional argument)
        if out is None:
            return self._cat(self._all(lambda r, sim: sim.get_state(*a)))
        self._all(lambda r, sim: sim.get_state(*a, out=self._slice(r, out)))          # shard slices of a C-contiguous array are contiguous
        return out

    def get_fields(self):
        return self._cat(self._all(lambda r, sim: tuple(sim.get_fields())))

    def get_nusselt(self):
        return self._cat(self._all(lambda r, sim: sim.get_nusselt()))

    def get_info(self):
        return self._cat(self._all(lambda r, sim: tuple(sim.get_info())))

    def get_flags(self):
        return self._cat(self._all(lambda r, sim: sim.get_flags()))

    def get_cell_distances(self, *a):
        return self._cat(self._all(lambda r, sim: sim.get_cell_distances(*a)))

    def synchronize(self):
        self._all(lambda r, sim: sim.synchronize())

    def dev_ptrs(self):
        return [sim.dev_ptrs() for sim in self.sims]

    def close(self):
        for sim in getattr(self, "sims", []):
            sim.close()
        self.sims = []
        pool = getattr(self, "_pool", None)
        if pool is not None:
            pool.shutdown(wait=True)
            self._pool = None
