"""ctypes binding of the CPU oracle (oracle/librbc_oracle.so).  Test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "librbc_oracle.so")

VAR_BOUNDS, VAR_SYMLEVEL, VAR_BUOYANCY, VAR_VISCOUS, VAR_POISSON = range(5)


class OracleConfig(C.Structure):
    _fields_ = [("nx", C.c_int32), ("nz", C.c_int32), ("lx", C.c_double), ("lz", C.c_double),
                ("ra", C.c_double), ("pr", C.c_double), ("min_b", C.c_double), ("delta_b", C.c_double),
                ("heaters", C.c_int32), ("heater_limit", C.c_double), ("dt_solver", C.c_double),
                ("dt_control", C.c_double), ("random_kick", C.c_double), ("obs_nx", C.c_int32),
                ("obs_nz", C.c_int32)]


def build_oracle(force=False):
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("rbc_oracle.c", "rbc_oracle3d.c", "rbc_oracle.h")]
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(x) for x in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(ORACLE_SO)
        dp = C.POINTER(C.c_double)
        fp = C.POINTER(C.c_float)
        L.rbco_create.restype = C.c_void_p
        L.rbco_create.argtypes = [C.POINTER(OracleConfig)]
        L.rbco_destroy.argtypes = [C.c_void_p]
        L.rbco_set_variant.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.rbco_reset_random.argtypes = [C.c_void_p, C.c_uint64]
        L.rbco_reset_from_arrays.argtypes = [C.c_void_p, dp, dp, dp]
        L.rbco_load_raw.argtypes = [C.c_void_p, dp, dp, dp]
        L.rbco_step.restype = C.c_int
        L.rbco_step.argtypes = [C.c_void_p, fp]
        L.rbco_set_action.argtypes = [C.c_void_p, fp]
        L.rbco_update_state.argtypes = [C.c_void_p]
        L.rbco_substep.argtypes = [C.c_void_p, C.c_double]
        L.rbco_get_tendencies.argtypes = [C.c_void_p, dp, dp, dp]
        L.rbco_projected_rate.argtypes = [C.c_void_p, dp, dp, dp]
        L.rbco_bottom_profile.argtypes = [C.c_void_p, dp]
        L.rbco_get_state.argtypes = [C.c_void_p, dp, C.c_int]
        L.rbco_get_state_f32.argtypes = [C.c_void_p, fp, C.c_int]
        L.rbco_get_obs_f32.argtypes = [C.c_void_p, fp, C.c_int]
        L.rbco_get_fields.argtypes = [C.c_void_p, dp, dp, dp]
        L.rbco_nusselt.restype = C.c_double
        L.rbco_nusselt.argtypes = [C.c_void_p, C.c_int]
        L.rbco_get_info.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int64)]
        L.rbco_max_divergence.restype = C.c_double
        L.rbco_max_divergence.argtypes = [C.c_void_p]
        L.rbco_kinetic_energy.restype = C.c_double
        L.rbco_kinetic_energy.argtypes = [C.c_void_p]
        L.rbco_normal.restype = C.c_double
        L.rbco_normal.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class OracleSim:
    """One env on the CPU oracle; method names follow the reference's Julia API
    (rbc_sim2D_api.jl: initialize_simulation, step_simulation, get_state, ...)."""

    def __init__(self, ra=1e4, nx=96, nz=64, heaters=12, heater_limit=0.75, dt_control=1.5,
                 dt_solver=0.03, obs=(8, 48), pr=0.7, min_b=1.0, delta_b=1.0, kick=0.01,
                 lx=2 * np.pi, lz=2.0, variants=None):
        self.cfg = OracleConfig(nx, nz, lx, lz, float(ra), pr, min_b, delta_b, heaters, heater_limit,
                                dt_solver, dt_control, kick, obs[1], obs[0])
        self.L = lib()
        self.h = C.c_void_p(self.L.rbco_create(C.byref(self.cfg)))
        if not self.h:
            raise RuntimeError("rbco_create failed")
        self.nx, self.nz, self.obs = nx, nz, tuple(obs)
        for k, v in (variants or {}).items():
            self.L.rbco_set_variant(self.h, k, v)

    def __del__(self):
        try:
            if self.h:
                self.L.rbco_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_variant(self, which, value):
        self.L.rbco_set_variant(self.h, which, value)

    def reset_random(self, seed):
        self.L.rbco_reset_random(self.h, int(seed))

    def _chk(self, b, u, w):
        b = np.ascontiguousarray(b, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        w = np.ascontiguousarray(w, dtype=np.float64)
        assert b.shape == (self.nz, self.nx) and u.shape == (self.nz, self.nx) and w.shape == (self.nz + 1, self.nx)
        return b, u, w

    def reset_from_arrays(self, b, u, w):
        b, u, w = self._chk(b, u, w)
        self.L.rbco_reset_from_arrays(self.h, _dp(b), _dp(u), _dp(w))

    def load_raw(self, b, u, w):
        b, u, w = self._chk(b, u, w)
        self.L.rbco_load_raw(self.h, _dp(b), _dp(u), _dp(w))

    def step(self, action=None):
        a = np.zeros(self.cfg.heaters, np.float32) if action is None else np.ascontiguousarray(action, np.float32)
        return bool(self.L.rbco_step(self.h, _fp(a)))

    def set_action(self, action):
        a = np.ascontiguousarray(action, np.float32)
        self.L.rbco_set_action(self.h, _fp(a))

    def update_state(self):
        self.L.rbco_update_state(self.h)

    def substep(self, dt):
        self.L.rbco_substep(self.h, float(dt))

    def tendencies(self):
        g = [np.empty((self.nz, self.nx)) for _ in range(3)]
        self.L.rbco_get_tendencies(self.h, _dp(g[0]), _dp(g[1]), _dp(g[2]))
        return dict(b=g[0], u=g[1], w=g[2])

    def projected_rate(self):
        g = [np.empty((self.nz, self.nx)) for _ in range(3)]
        self.L.rbco_projected_rate(self.h, _dp(g[0]), _dp(g[1]), _dp(g[2]))
        return dict(u=g[0], w=g[1], b=g[2])

    def bottom_profile(self):
        t = np.empty(self.nx)
        self.L.rbco_bottom_profile(self.h, _dp(t))
        return t

    def fields(self):
        b, u, w = np.empty((self.nz, self.nx)), np.empty((self.nz, self.nx)), np.empty((self.nz + 1, self.nx))
        self.L.rbco_get_fields(self.h, _dp(b), _dp(u), _dp(w))
        return b, u, w

    def state(self, nch=3, f32=True):
        if f32:
            o = np.empty((nch, self.nz, self.nx), np.float32)
            self.L.rbco_get_state_f32(self.h, _fp(o), nch)
        else:
            o = np.empty((nch, self.nz, self.nx))
            self.L.rbco_get_state(self.h, _dp(o), nch)
        return o

    def obs_f32(self, nch=3):
        o = np.empty((nch,) + self.obs, np.float32)
        self.L.rbco_get_obs_f32(self.h, _fp(o), nch)
        return o

    def nusselt(self, state=True):
        return self.L.rbco_nusselt(self.h, 1 if state else 0)

    def info(self):
        t, s = C.c_double(), C.c_int64()
        self.L.rbco_get_info(self.h, C.byref(t), C.byref(s))
        return t.value, s.value

    def max_divergence(self):
        return self.L.rbco_max_divergence(self.h)

    def kinetic_energy(self):
        return self.L.rbco_kinetic_energy(self.h)


# ---------------------------------------------------------------------------------------------
# 3D oracle (oracle/rbc_oracle3d.c)
# ---------------------------------------------------------------------------------------------
class Oracle3Config(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("lx", C.c_double), ("ly", C.c_double), ("lz", C.c_double),
                ("ra", C.c_double), ("pr", C.c_double), ("min_b", C.c_double), ("delta_b", C.c_double),
                ("heaters", C.c_int32), ("heater_limit", C.c_double), ("dt_solver", C.c_double),
                ("dt_control", C.c_double), ("random_kick", C.c_double)]


_lib3_ready = False


def lib3():
    global _lib3_ready
    L = lib()
    if not _lib3_ready:
        dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.rbco3_create.restype = C.c_void_p
        L.rbco3_create.argtypes = [C.POINTER(Oracle3Config)]
        L.rbco3_destroy.argtypes = [C.c_void_p]
        L.rbco3_reset_random.argtypes = [C.c_void_p, C.c_uint64]
        L.rbco3_reset_from_arrays.argtypes = [C.c_void_p, dp, dp, dp, dp]
        L.rbco3_step.restype = C.c_int
        L.rbco3_step.argtypes = [C.c_void_p, fp]
        L.rbco3_set_action.argtypes = [C.c_void_p, fp, C.c_int]
        L.rbco3_update_state.argtypes = [C.c_void_p]
        L.rbco3_substep.argtypes = [C.c_void_p, C.c_double]
        L.rbco3_get_fields.argtypes = [C.c_void_p, dp, dp, dp, dp]
        L.rbco3_get_tendencies.argtypes = [C.c_void_p, dp, dp, dp, dp]
        L.rbco3_get_state_f32.argtypes = [C.c_void_p, fp]
        L.rbco3_nusselt.restype = C.c_double
        L.rbco3_nusselt.argtypes = [C.c_void_p]
        L.rbco3_get_info.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int64)]
        L.rbco3_max_divergence.restype = C.c_double
        L.rbco3_max_divergence.argtypes = [C.c_void_p]
        _lib3_ready = True
    return L


class Oracle3D:
    """One 3D env on the CPU oracle (rbc_sim3D_api.jl semantics; shapes (nz, ny, nx))."""

    def __init__(self, ra=2500.0, shape=(16, 32, 32), domain=(2.0, 4 * np.pi, 4 * np.pi), heaters=8, heater_limit=0.9,
                 dt_control=0.125, dt_solver=0.01, pr=0.7, t_diff=(1.0, 2.0), kick=0.01):
        nz, ny, nx = shape
        lz, ly, lx = domain
        self.cfg = Oracle3Config(nx, ny, nz, lx, ly, lz, float(ra), pr, t_diff[0], t_diff[1] - t_diff[0], heaters,
                                 heater_limit, dt_solver, dt_control, kick)
        self.L = lib3()
        self.h = C.c_void_p(self.L.rbco3_create(C.byref(self.cfg)))
        if not self.h:
            raise RuntimeError("rbco3_create failed")
        self.nx, self.ny, self.nz, self.heaters = nx, ny, nz, heaters

    def __del__(self):
        try:
            if self.h:
                self.L.rbco3_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def reset_random(self, seed):
        self.L.rbco3_reset_random(self.h, int(seed))

    def reset_from_arrays(self, b, u, v, w):
        a = [np.ascontiguousarray(x, np.float64) for x in (b, u, v, w)]
        assert a[0].shape == (self.nz, self.ny, self.nx) and a[3].shape == (self.nz + 1, self.ny, self.nx)
        self.L.rbco3_reset_from_arrays(self.h, *[_dp(x) for x in a])

    def step(self, action=None):
        a = np.zeros((self.heaters, self.heaters), np.float32) if action is None else np.ascontiguousarray(action, np.float32)
        return bool(self.L.rbco3_step(self.h, _fp(a)))

    def set_action(self, action):
        a = np.ascontiguousarray(action, np.float32)
        self.L.rbco3_set_action(self.h, _fp(a), 0)

    def update_state(self):
        self.L.rbco3_update_state(self.h)

    def substep(self, dt):
        self.L.rbco3_substep(self.h, float(dt))

    def fields(self):
        s = (self.nz, self.ny, self.nx)
        b, u, v, w = np.empty(s), np.empty(s), np.empty(s), np.empty((self.nz + 1, self.ny, self.nx))
        self.L.rbco3_get_fields(self.h, _dp(b), _dp(u), _dp(v), _dp(w))
        return b, u, v, w

    def tendencies(self):
        g = [np.empty((self.nz, self.ny, self.nx)) for _ in range(4)]
        self.L.rbco3_get_tendencies(self.h, *[_dp(x) for x in g])
        return dict(u=g[0], v=g[1], w=g[2], b=g[3])

    def state(self):
        o = np.empty((4, self.nz, self.ny, self.nx), np.float32)
        self.L.rbco3_get_state_f32(self.h, _fp(o))
        return o

    def nusselt(self):
        return self.L.rbco3_nusselt(self.h)

    def info(self):
        t, s = C.c_double(), C.c_int64()
        self.L.rbco3_get_info(self.h, C.byref(t), C.byref(s))
        return t.value, s.value

    def max_divergence(self):
        return self.L.rbco3_max_divergence(self.h)
