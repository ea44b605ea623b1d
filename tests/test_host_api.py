"""CPU: the C-ABI library loads, exports every symbol include/rbc_hip.h declares, agrees with the
ctypes struct layout, validates arguments and FAILS LOUDLY without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rbc_hip.h")


@pytest.fixture(scope="module")
def native():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.build_hip()
    from rbc_gym import _native
    return _native


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rbc_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(native):
    lib = C.CDLL(native.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in rbc_hip.h but not exported"
    assert set(native.SYMBOLS) == set(names)
    assert lib.rbc_abi_version() == native.ABI_VERSION


def test_config_struct_layout_matches_header(native, tmp_path):
    prog = tmp_path / "layout.c"
    fields = [f for f, _ in native.RbcConfig._fields_]
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rbc_hip.h"\nint main(void){printf("%zu",sizeof(rbc_config));'
                    + "".join(f'printf(" %zu",offsetof(rbc_config,{f}));' for f in fields) + "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert out[0] == C.sizeof(native.RbcConfig)
    assert out[1:] == [getattr(native.RbcConfig, f).offset for f in fields]


def test_default_config_is_the_registry_default(native):
    c = native.default_config()
    assert (c.nx, c.nz, c.obs_nx, c.obs_nz, c.heaters) == (96, 64, 48, 8, 12)
    assert c.ra == 1e4 and c.pr == 0.7 and c.heater_limit == 0.75 and c.dt_solver == 0.03 and c.dt_control == 1.5
    assert abs(c.lx - 2 * np.pi) < 1e-15 and c.lz == 2.0 and c.min_b == 1 and c.delta_b == 1 and c.random_kick == 0.01
    assert c.reference_clock == native.CLOCKS["documented"] == 0         # the sources' clock is the default; "recorded" is opt-in


def test_invalid_configs_are_rejected_before_touching_the_device(native):
    lib = native.load_library()
    for field, value, frag in (("dim", 4, "dim must be"), ("nx", 4, "unsupported 2D grid"), ("precision", 7, "precision"), ("batch", 0, "batch"),
                               ("heaters", 0, "heaters"), ("obs_nx", 5, "sensor"), ("abi_version", 99, "abi_version"),
                               ("reference_clock", 2, "reference_clock")):
        cfg = native.default_config()
        setattr(cfg, field, value)
        h = C.c_void_p()
        rc = lib.rbc_create(C.byref(cfg), C.byref(h))
        assert rc == native.RBC_ERR_INVALID and not h.value
        assert frag in lib.rbc_last_error().decode()


def test_recorded_clock_needs_more_than_one_solver_step(native):
    """RBC_CLOCK_RECORDED drops one full solver step from every env-step but the first: refused where there is none to drop"""
    lib = native.load_library()
    cfg = native.default_config()
    cfg.reference_clock, cfg.dt_control = native.CLOCKS["recorded"], cfg.dt_solver
    h = C.c_void_p()
    assert lib.rbc_create(C.byref(cfg), C.byref(h)) == native.RBC_ERR_INVALID and "dt_control > dt_solver" in lib.rbc_last_error().decode()
    with pytest.raises(ValueError):
        native.clock_code("julia")


def test_shipped_library_has_no_experiment_knobs(native):
    """numerics-changing RBC_EXPERIMENT_* environment knobs exist only in -DRBC_EXPERIMENTS=1 builds made by scripts/"""
    blob = open(native.LIB_PATH, "rb").read()
    assert b"RBC_EXPERIMENT" not in blob


def test_no_gpu_means_failure_not_fallback(native):
    lib = native.load_library()
    if lib.rbc_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(native.RbcError) as e:
        native.NativeSim(batch=2)
    assert e.value.code == native.RBC_ERR_DEVICE


def test_missing_library_raises(native, tmp_path):
    with pytest.raises(ImportError):
        native.load_library(str(tmp_path / "nope.so"))


def test_product_never_references_the_oracle():
    """the oracle is test infrastructure: nothing under rbc-gym_amd/ may import, link or open it."""
    pkg = os.path.join(ROOT, "rbc-gym_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "librbc_oracle" not in txt and "oracle_py" not in txt and "rbc_oracle.h" not in txt, f
