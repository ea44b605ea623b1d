"""ctypes binding of librbc_hip.so (the C ABI in include/rbc_hip.h).

This is the Python side of the drop-in boundary: where the reference does
``juliacall.newmodule("RBCGymAPI")`` + ``include("rbc_sim2D_api.jl")`` (rbc2D.py:111-115),
this package loads one shared library.  There is no fallback: if the library is missing or
no MI355X is visible, constructing a simulation raises.
"""
import ctypes as C
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)                      # rbc-gym_amd/
LIB_PATH = os.path.join(ROOT, "lib", "librbc_hip.so")

RBC_OK, RBC_ERR_INVALID, RBC_ERR_DEVICE, RBC_ERR_NAN, RBC_ERR_NOT_INITIALIZED = range(5)
ABI_VERSION = 3


class RbcConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("dim", C.c_int32),
                ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("lx", C.c_double), ("ly", C.c_double), ("lz", C.c_double),
                ("ra", C.c_double), ("pr", C.c_double),
                ("min_b", C.c_double), ("delta_b", C.c_double),
                ("heaters", C.c_int32), ("heater_limit", C.c_double),
                ("dt_solver", C.c_double), ("dt_control", C.c_double),
                ("random_kick", C.c_double),
                ("obs_nx", C.c_int32), ("obs_nz", C.c_int32),
                ("batch", C.c_int32), ("device", C.c_int32), ("write_state", C.c_int32), ("precision", C.c_int32),
                ("reference_clock", C.c_int32)]

PRECISIONS = {"f64": 0, "f32": 1}
# rbc_config.reference_clock (include/rbc_hip.h): "documented" = every env-step integrates heater_duration, what the reference's
# sources say (rbc_sim2D_api.jl:84-85); "recorded" = what its recorded flowstats series show: the first env-step after a reset
# carries all its solver steps, every later one of them one less
CLOCKS = {"documented": 0, "recorded": 1}


def clock_code(name):
    if name in CLOCKS:
        return CLOCKS[name]
    if name in CLOCKS.values():
        return int(name)
    raise ValueError(f"reference_clock must be one of {sorted(CLOCKS)}, got {name!r}")


# every symbol include/rbc_hip.h declares: name -> (restype, argtypes)
_vp, _dp, _fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)
_u8p, _u64p, _i64p, _i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
SYMBOLS = {
    "rbc_abi_version": (C.c_int, []),
    "rbc_last_error": (C.c_char_p, []),
    "rbc_device_count": (C.c_int, []),
    "rbc_has_precision": (C.c_int, [C.c_int]),
    "rbc_copy_ceiling": (C.c_int, [C.c_int, C.c_size_t, C.c_int, _dp, _dp]),
    "rbc_default_config": (None, [C.POINTER(RbcConfig)]),
    "rbc_create": (C.c_int, [C.POINTER(RbcConfig), C.POINTER(_vp)]),
    "rbc_destroy": (C.c_int, [_vp]),
    "rbc_set_stream": (C.c_int, [_vp, _vp]),
    "rbc_get_stream": (_vp, [_vp]),
    "rbc_synchronize": (C.c_int, [_vp]),
    "rbc_set_rayleigh": (C.c_int, [_vp, _dp]),
    "rbc_set_obs_normalization": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_double, C.c_int]),
    "rbc_get_cell_distances": (C.c_int, [_vp, C.c_double, _dp]),
    "rbc_dev_cell_dist": (_vp, [_vp]),
    "rbc_debug_cell_distances": (C.c_int, [C.c_int, _fp, C.c_int, C.c_int, C.c_double, C.c_double, _dp]),
    "rbc_host_alloc": (_vp, [C.c_size_t]),
    "rbc_host_free": (None, [_vp]),
    "rbc_reset": (C.c_int, [_vp, _u8p, _u64p]),
    "rbc_reset_from_arrays": (C.c_int, [_vp, _u8p, _dp, _dp, _dp]),
    "rbc_step": (C.c_int, [_vp, _fp]),
    "rbc_step_dev": (C.c_int, [_vp, _vp]),
    "rbc_get_obs": (C.c_int, [_vp, _fp, C.c_int]),
    "rbc_get_state": (C.c_int, [_vp, _fp, C.c_int]),
    "rbc_get_fields": (C.c_int, [_vp, _dp, _dp, _dp]),
    "rbc_get_nusselt": (C.c_int, [_vp, _dp, _dp]),
    "rbc_get_info": (C.c_int, [_vp, _dp, _i64p]),
    "rbc_get_flags": (C.c_int, [_vp, _i32p]),
    "rbc_dev_obs": (_vp, [_vp]),
    "rbc_dev_state": (_vp, [_vp]),
    "rbc_dev_nusselt": (_vp, [_vp]),
    "rbc_dev_flags": (_vp, [_vp]),
    "rbc_dev_fields": (_vp, [_vp]),
    "rbc_set_profiling": (C.c_int, [_vp, C.c_int]),
    "rbc_profile_read": (C.c_int, [_vp, _dp, C.c_int]),
    "rbc_algorithmic_bytes_per_env_step": (C.c_double, [_vp]),
    "rbc_debug_tendencies": (C.c_int, [_vp, _fp, _dp, _dp, _dp]),
    "rbc_debug_substeps": (C.c_int, [_vp, _fp, C.c_int, C.c_double]),
    "rbc_debug_stamps": (C.c_int, [_vp, _u64p]),
    "rbc_debug_launch_plan": (C.c_int, [_vp, _i32p]),
    "rbc_reset_from_arrays3": (C.c_int, [_vp, _u8p, _dp, _dp, _dp, _dp]),
    "rbc_get_fields3": (C.c_int, [_vp, _dp, _dp, _dp, _dp]),
    "rbc_debug_tendencies3": (C.c_int, [_vp, _fp, _dp, _dp, _dp, _dp]),
}

_lib = None


class RbcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7) and libc10_hip.so asks
    for it by file name, so if librbc_hip.so pulled in /opt/rocm's copy first, a later `import torch` would
    load a second HIP runtime into the process and see no GPUs.  When torch is installed (and not yet
    imported) preload its copy, so both bind to one runtime whatever the import order.
    RBC_HIP_SYSTEM_RUNTIME=1 skips this."""
    if os.environ.get("RBC_HIP_SYSTEM_RUNTIME") == "1" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass                      # fall back to the system runtime


def load_library(path=None):
    """Load librbc_hip.so and bind every declared symbol.  Raises if the library is absent:
    the product path never falls back to a CPU implementation."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RBC_HIP_LIB", LIB_PATH)
    if not os.path.exists(p):
        raise ImportError(
            f"librbc_hip.so not found at {p}: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). rbc_gym has no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(p)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.rbc_abi_version() != ABI_VERSION:
        raise ImportError("librbc_hip.so ABI version mismatch")
    if path is None:
        _lib = lib
    return lib


class _PinnedBlock:
    """owner of one hipHostMalloc block; numpy arrays made from it keep it alive through their .base chain"""

    def __init__(self, lib, nbytes):
        self.lib, self.ptr, self.nbytes = lib, lib.rbc_host_alloc(nbytes), nbytes
        if not self.ptr:
            raise MemoryError(f"rbc_host_alloc({nbytes}) failed")
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            if self.ptr:
                self.lib.rbc_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32):
    """numpy array over page-locked host memory (rbc_host_alloc): a fast destination for get_state/get_obs(out=...)."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    block = _PinnedBlock(load_library(), max(n, 1))
    return np.asarray(block)[:n].view(dt).reshape(shape)


class PinnedPool:
    """Output arrays that behave like fresh ones and copy like page-locked ones.

    The reference's API returns a new array from every step (rbc2D.py:185-196) and the caller may keep it for ever.  A new 38 MB (3D
    observations at configs[4]) or 75 MB (2D states at B = 1024) array per step costs more than the step's device-to-host copy:
    the allocator maps fresh pages, the kernel zero-fills them, the copy into pageable memory is staged.  The pool hands out
    page-locked arrays (rbc_host_alloc: direct DMA) and reuses one ONLY when nothing but the pool references it any more
    (sys.getrefcount: a view, a dict entry, a list slot of the caller's all count) -- so an array the caller still holds is
    never written again, exactly as with fresh arrays.  At most `cap` arrays; when the caller holds them all, `take()` returns
    None and the caller of take() allocates an ordinary array."""

    def __init__(self, shape, dtype=np.float32, cap=4):
        self.shape, self.dtype, self.cap = tuple(int(x) for x in shape), np.dtype(dtype), int(cap)
        self._items = []

    def take(self):
        for i in range(len(self._items)):
            if sys.getrefcount(self._items[i]) == 2:     # the pool's list + getrefcount's own argument: nobody else
                return self._items[i]
        if len(self._items) < self.cap:
            try:
                self._items.append(pinned_empty(self.shape, self.dtype))
            except (MemoryError, RbcError, OSError):
                self.cap = len(self._items)              # no more page-locked memory to be had: stop asking
                return None
            return self._items[-1]
        return None


POOL_MIN_BYTES = 8 << 20         # smaller outputs are not worth a pool (a fresh 4.7 MB observation array costs ~0.3 ms)


def default_config():
    cfg = RbcConfig()
    load_library().rbc_default_config(C.byref(cfg))
    return cfg


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


def has_precision(name):
    """does this build of librbc_hip.so carry kernels for "f64" / "f32"?"""
    return bool(load_library().rbc_has_precision(PRECISIONS[name]))


def copy_ceiling(device=0, nbytes=1 << 30, iters=10):
    """on-box streaming-copy rate (GB/s, bytes read + written) by a 16-bytes-per-lane copy kernel and by hipMemcpyAsync D2D"""
    lib = load_library()
    k, m = C.c_double(), C.c_double()
    rc = lib.rbc_copy_ceiling(int(device), int(nbytes), int(iters), C.byref(k), C.byref(m))
    if rc != RBC_OK:
        raise RbcError(rc, lib.rbc_last_error().decode())
    return {"kernel_gbs": k.value, "memcpy_d2d_gbs": m.value, "bytes": int(nbytes), "iters": int(iters),
            "note": "GB/s of bytes read + bytes written, device buffer to device buffer"}


def debug_cell_distances(uy, lx=2 * np.pi, height=0.001, device=0):
    """the device kernel behind NativeSim.get_cell_distances on caller-provided signals uy[B, nx] (parity tests)"""
    lib = load_library()
    a = np.ascontiguousarray(uy, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError("uy must be (B, nx)")
    o = np.empty(a.shape[0])
    rc = lib.rbc_debug_cell_distances(int(device), _ptr(a, _fp), a.shape[0], a.shape[1], float(lx), float(height), _ptr(o, _dp))
    if rc != RBC_OK:
        raise RbcError(rc, lib.rbc_last_error().decode())
    return o


def torch_stream_handle(stream):
    """value for rbc_set_stream that makes the sim run ON a torch stream: torch's default stream has handle 0, which the
    C ABI reserves for "the handle's own stream", so it is passed as hipStreamLegacy (1) instead."""
    return int(stream.cuda_stream) or 1


class NativeSim:
    """A batch of B envs on one GPU.  Method names follow the reference's Julia API
    (rbc_sim2D_api.jl): initialize_simulation -> reset*, step_simulation -> step, get_state,
    get_observation, get_info, get_nusselt."""

    def __init__(self, batch=1, device=0, **kw):
        self.lib = load_library()
        cfg = default_config()
        cfg.batch, cfg.device = int(batch), int(device)
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown config field {k}")
            setattr(cfg, k, clock_code(v) if k == "reference_clock" else v)
        self.cfg = cfg
        self.h = _vp()
        self._check(self.lib.rbc_create(C.byref(cfg), C.byref(self.h)))
        self.B, self.nx, self.nz = cfg.batch, cfg.nx, cfg.nz
        self.obs_shape = (cfg.obs_nz, cfg.obs_nx)
        self.heaters = cfg.heaters

    def _check(self, rc):
        if rc != RBC_OK:
            raise RbcError(rc, self.lib.rbc_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.rbc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- initialize_simulation ---------------------------------------------------------------
    def _mask(self, mask):
        if mask is None:
            return None, None
        m = np.ascontiguousarray(mask, dtype=np.uint8)
        if m.shape != (self.B,):
            raise ValueError(f"mask must have shape {(self.B,)}, got {m.shape}")
        return m, _ptr(m, _u8p)

    def reset(self, seeds, mask=None):
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (self.B,)))
        m, mp = self._mask(mask)
        self._check(self.lib.rbc_reset(self.h, mp, _ptr(s, _u64p)))

    def reset_from_arrays(self, b, u, w, mask=None):
        b = np.ascontiguousarray(b, np.float64); u = np.ascontiguousarray(u, np.float64); w = np.ascontiguousarray(w, np.float64)
        want = {"b": (self.B, self.nz, self.nx), "u": (self.B, self.nz, self.nx), "w": (self.B, self.nz + 1, self.nx)}
        for name, a in (("b", b), ("u", u), ("w", w)):          # raw pointers go to memcpy: never trust the caller's shapes
            if a.shape != want[name]:
                raise ValueError(f"reset_from_arrays: {name} must have shape {want[name]} (batch, z, x), got {a.shape}")
        m, mp = self._mask(mask)
        self._check(self.lib.rbc_reset_from_arrays(self.h, mp, _ptr(b, _dp), _ptr(u, _dp), _ptr(w, _dp)))

    def set_rayleigh(self, ra):
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(ra, np.float64), (self.B,)))
        self._check(self.lib.rbc_set_rayleigh(self.h, _ptr(r, _dp)))

    def set_obs_normalization(self, min_vals=None, max_vals=None, maxval=1.0, clip=False):
        """RBCNormalizeObservation fused into the kernel's observation write; None switches it off."""
        if min_vals is None:
            self._check(self.lib.rbc_set_obs_normalization(self.h, None, None, 0, 1.0, 0))
            return
        lo = np.ascontiguousarray(min_vals, np.float64)
        hi = np.ascontiguousarray(max_vals, np.float64)
        if lo.shape != hi.shape or lo.ndim != 1:
            raise ValueError("set_obs_normalization: min_vals and max_vals must be 1-D and of equal length")
        self._check(self.lib.rbc_set_obs_normalization(self.h, _ptr(lo, _dp), _ptr(hi, _dp), int(lo.size), float(maxval), int(bool(clip))))

    # -- step_simulation ---------------------------------------------------------------------
    def _actions(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        if a.shape != (self.B, self.heaters):
            raise ValueError(f"actions must have shape {(self.B, self.heaters)}, got {a.shape}")
        return a

    def step(self, actions):
        a = self._actions(actions)
        rc = self.lib.rbc_step(self.h, _ptr(a, _fp))
        if rc == RBC_ERR_NAN:
            return False
        self._check(rc)
        return True

    def step_dev(self, actions_dev_ptr):
        self._check(self.lib.rbc_step_dev(self.h, _vp(actions_dev_ptr)))

    # -- getters -----------------------------------------------------------------------------
    def get_obs(self, nch=3):
        o = np.empty((self.B, nch) + self.obs_shape, np.float32)
        self._check(self.lib.rbc_get_obs(self.h, _ptr(o, _fp), nch))
        return o

    def get_state(self, nch=3, out=None):
        """float32 (B, nch, nz, nx); `out` may be a caller-owned array of that shape, e.g. from pinned_empty()"""
        o = np.empty((self.B, nch, self.nz, self.nx), np.float32) if out is None else out
        if o.shape != (self.B, nch, self.nz, self.nx) or o.dtype != np.float32 or not o.flags.c_contiguous:
            raise ValueError(f"get_state: out must be a C-contiguous float32 array of shape {(self.B, nch, self.nz, self.nx)}")
        self._check(self.lib.rbc_get_state(self.h, _ptr(o, _fp), nch))
        return o

    def get_fields(self):
        b = np.empty((self.B, self.nz, self.nx)); u = np.empty_like(b); w = np.empty((self.B, self.nz + 1, self.nx))
        self._check(self.lib.rbc_get_fields(self.h, _ptr(b, _dp), _ptr(u, _dp), _ptr(w, _dp)))
        return b, u, w

    def get_nusselt(self):
        a = np.empty(self.B); o = np.empty(self.B)
        self._check(self.lib.rbc_get_nusselt(self.h, _ptr(a, _dp), _ptr(o, _dp)))
        return a, o

    def get_info(self):
        t = np.empty(self.B); s = np.empty(self.B, np.int64)
        self._check(self.lib.rbc_get_info(self.h, _ptr(t, _dp), _ptr(s, _i64p)))
        return t, s

    def launch_plan(self):
        """(env groups, 1 if every group replays its own graph on a hardware queue of its own) -- include/rbc_hip.h rbc_debug_launch_plan"""
        g = np.zeros(2, np.int32)
        self._check(self.lib.rbc_debug_launch_plan(self.h, _ptr(g, _i32p)))
        return int(g[0]), int(g[1])

    def get_flags(self):
        f = np.empty(self.B, np.int32)
        self._check(self.lib.rbc_get_flags(self.h, _ptr(f, _i32p)))
        return f

    def get_cell_distances(self, height=0.001):
        """RBCRewardShaping.compute_cell_distances for every env, evaluated on the device (needs write_state=1)"""
        o = np.empty(self.B)
        self._check(self.lib.rbc_get_cell_distances(self.h, float(height), _ptr(o, _dp)))
        return o

    def synchronize(self):
        self._check(self.lib.rbc_synchronize(self.h))

    # -- measurement / test hooks --------------------------------------------------------------
    def set_profiling(self, max_launches):
        self._check(self.lib.rbc_set_profiling(self.h, int(max_launches)))

    def profile_read(self, capacity=4096):
        ms = np.empty(capacity)
        n = self.lib.rbc_profile_read(self.h, _ptr(ms, _dp), capacity)
        if n < 0:
            raise RbcError(RBC_ERR_DEVICE, "rbc_profile_read failed")
        return ms[:n].copy()

    def algorithmic_bytes_per_env_step(self):
        return self.lib.rbc_algorithmic_bytes_per_env_step(self.h)

    def debug_tendencies(self, actions):
        a = self._actions(actions)
        g = [np.empty((self.B, self.nz, self.nx)) for _ in range(3)]
        self._check(self.lib.rbc_debug_tendencies(self.h, _ptr(a, _fp), _ptr(g[0], _dp), _ptr(g[1], _dp), _ptr(g[2], _dp)))
        return dict(b=g[0], u=g[1], w=g[2])

    def debug_substeps(self, actions, nsub, dt):
        a = self._actions(actions)
        self._check(self.lib.rbc_debug_substeps(self.h, _ptr(a, _fp), int(nsub), float(dt)))

    def dev_ptrs(self):
        L = self.lib
        return dict(obs=L.rbc_dev_obs(self.h), state=L.rbc_dev_state(self.h), nusselt=L.rbc_dev_nusselt(self.h),
                    flags=L.rbc_dev_flags(self.h), fields=L.rbc_dev_fields(self.h))


class NativeSim3D:
    """A batch of B 3D envs on one GPU (rbc_sim3D_api.jl semantics; array shapes (nz, ny, nx))."""

    def __init__(self, batch=1, device=0, shape=(16, 32, 32), domain=(2.0, 4 * np.pi, 4 * np.pi), ra=2500.0, pr=0.7,
                 t_diff=(1.0, 2.0), heaters=8, heater_limit=0.9, dt_control=0.125, dt_solver=0.01, random_kick=None, precision=0,
                 reference_clock="documented"):
        self.lib = load_library()
        cfg = default_config()
        cfg.precision = int(PRECISIONS.get(precision, precision))
        cfg.reference_clock = clock_code(reference_clock)
        nz, ny, nx = shape
        lz, ly, lx = domain
        cfg.dim, cfg.nx, cfg.ny, cfg.nz = 3, int(nx), int(ny), int(nz)
        cfg.lx, cfg.ly, cfg.lz = float(lx), float(ly), float(lz)
        cfg.ra, cfg.pr = float(ra), float(pr)
        cfg.min_b, cfg.delta_b = float(t_diff[0]), float(t_diff[1] - t_diff[0])
        cfg.heaters, cfg.heater_limit = int(heaters), float(heater_limit)
        cfg.dt_control, cfg.dt_solver = float(dt_control), float(dt_solver)
        cfg.obs_nx, cfg.obs_nz = int(nx), int(nz)
        if random_kick is not None:
            cfg.random_kick = float(random_kick)
        cfg.batch, cfg.device = int(batch), int(device)
        self.cfg = cfg
        self.h = _vp()
        rc = self.lib.rbc_create(C.byref(cfg), C.byref(self.h))
        if rc != RBC_OK:
            raise RbcError(rc, self.lib.rbc_last_error().decode())
        self.B, self.nx, self.ny, self.nz, self.heaters = cfg.batch, nx, ny, nz, heaters

    def _check(self, rc):
        if rc != RBC_OK:
            raise RbcError(rc, self.lib.rbc_last_error().decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.rbc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _mask(self, mask):
        if mask is None:
            return None
        self._m = np.ascontiguousarray(mask, dtype=np.uint8)
        if self._m.shape != (self.B,):
            raise ValueError(f"mask must have shape {(self.B,)}, got {self._m.shape}")
        return _ptr(self._m, _u8p)

    def reset(self, seeds, mask=None):
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint64), (self.B,)))
        self._check(self.lib.rbc_reset(self.h, self._mask(mask), _ptr(s, _u64p)))

    def reset_from_arrays(self, b, u, v, w, mask=None):
        a = [np.ascontiguousarray(x, np.float64) for x in (b, u, v, w)]
        cen, fac = (self.B, self.nz, self.ny, self.nx), (self.B, self.nz + 1, self.ny, self.nx)
        for name, x, want in (("b", a[0], cen), ("u", a[1], cen), ("v", a[2], cen), ("w", a[3], fac)):
            if x.shape != want:
                raise ValueError(f"reset_from_arrays: {name} must have shape {want} (batch, z, y, x), got {x.shape}")
        self._check(self.lib.rbc_reset_from_arrays3(self.h, self._mask(mask), *[_ptr(x, _dp) for x in a]))

    def set_rayleigh(self, ra):
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(ra, np.float64), (self.B,)))
        self._check(self.lib.rbc_set_rayleigh(self.h, _ptr(r, _dp)))

    def set_obs_normalization(self, min_vals=None, max_vals=None, maxval=1.0, clip=False):
        """RBCNormalizeObservation fused into the output kernel's float32 state write (channels b, u, v, w); None switches it off."""
        if min_vals is None:
            self._check(self.lib.rbc_set_obs_normalization(self.h, None, None, 0, 1.0, 0))
            return
        lo = np.ascontiguousarray(min_vals, np.float64)
        hi = np.ascontiguousarray(max_vals, np.float64)
        if lo.shape != hi.shape or lo.ndim != 1:
            raise ValueError("set_obs_normalization: min_vals and max_vals must be 1-D and of equal length")
        self._check(self.lib.rbc_set_obs_normalization(self.h, _ptr(lo, _dp), _ptr(hi, _dp), int(lo.size), float(maxval), int(bool(clip))))

    def _actions(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float32)
        if a.shape != (self.B, self.heaters, self.heaters):
            raise ValueError(f"actions must have shape {(self.B, self.heaters, self.heaters)}, got {a.shape}")
        return a

    def step(self, actions):
        a = self._actions(actions)
        rc = self.lib.rbc_step(self.h, _ptr(a, _fp))
        if rc == RBC_ERR_NAN:
            return False
        self._check(rc)
        return True

    def step_dev(self, actions_dev_ptr):
        self._check(self.lib.rbc_step_dev(self.h, _vp(actions_dev_ptr)))

    def get_state(self, out=None):
        """float32 (B, 4, nz, ny, nx); `out` may be a caller-owned array of that shape, e.g. from pinned_empty()"""
        shape = (self.B, 4, self.nz, self.ny, self.nx)
        o = np.empty(shape, np.float32) if out is None else out
        if o.shape != shape or o.dtype != np.float32 or not o.flags.c_contiguous:
            raise ValueError(f"get_state: out must be a C-contiguous float32 array of shape {shape}")
        self._check(self.lib.rbc_get_state(self.h, _ptr(o, _fp), 4))
        return o

    get_obs = get_state          # rbc3D.py:229-232: the observation is the full state

    def get_fields(self):
        s = (self.B, self.nz, self.ny, self.nx)
        b, u, v, w = np.empty(s), np.empty(s), np.empty(s), np.empty((self.B, self.nz + 1, self.ny, self.nx))
        self._check(self.lib.rbc_get_fields3(self.h, *[_ptr(x, _dp) for x in (b, u, v, w)]))
        return b, u, v, w

    def get_nusselt(self):
        a = np.empty(self.B)
        self._check(self.lib.rbc_get_nusselt(self.h, _ptr(a, _dp), None))
        return a

    def get_info(self):
        t = np.empty(self.B); s = np.empty(self.B, np.int64)
        self._check(self.lib.rbc_get_info(self.h, _ptr(t, _dp), _ptr(s, _i64p)))
        return t, s

    def launch_plan(self):
        """(env groups, 1 if every group replays its own graph on a hardware queue of its own) -- include/rbc_hip.h rbc_debug_launch_plan"""
        g = np.zeros(2, np.int32)
        self._check(self.lib.rbc_debug_launch_plan(self.h, _ptr(g, _i32p)))
        return int(g[0]), int(g[1])

    def get_flags(self):
        f = np.empty(self.B, np.int32)
        self._check(self.lib.rbc_get_flags(self.h, _ptr(f, _i32p)))
        return f

    def synchronize(self):
        self._check(self.lib.rbc_synchronize(self.h))

    def set_profiling(self, n):
        self._check(self.lib.rbc_set_profiling(self.h, int(n)))

    def profile_read(self, capacity=4096):
        ms = np.empty(capacity)
        k = self.lib.rbc_profile_read(self.h, _ptr(ms, _dp), capacity)
        return ms[:max(k, 0)].copy()

    def algorithmic_bytes_per_env_step(self):
        return self.lib.rbc_algorithmic_bytes_per_env_step(self.h)

    def debug_tendencies(self, actions):
        a = self._actions(actions)
        g = [np.empty((self.B, self.nz, self.ny, self.nx)) for _ in range(4)]
        self._check(self.lib.rbc_debug_tendencies3(self.h, _ptr(a, _fp), *[_ptr(x, _dp) for x in g]))
        return dict(u=g[0], v=g[1], w=g[2], b=g[3])

    def debug_substeps(self, actions, nsub, dt):
        a = self._actions(actions)
        self._check(self.lib.rbc_debug_substeps(self.h, _ptr(a, _fp), int(nsub), float(dt)))

    def dev_ptrs(self):
        L = self.lib
        return dict(state=L.rbc_dev_state(self.h), nusselt=L.rbc_dev_nusselt(self.h), flags=L.rbc_dev_flags(self.h),
                    fields=L.rbc_dev_fields(self.h))
