"""Register / scratch budget of the hot kernels, read from the gfx950 code object inside the built library (no GPU):
a compiler bump or a source change that starts spilling fails HERE instead of showing up as a few per cent on the GPU.
The bounds are today's numbers (scripts/kernel_resources.py prints the whole table)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import kernel_resources as kr  # noqa: E402

LIB = os.path.join(ROOT, "rbc-gym_amd", "lib", "librbc_hip.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(os.path.join(kr.LLVM, "llvm-readelf")) and shutil.which("c++filt")),
                                reason="needs the built library and ROCm's llvm tools")

# kernel -> upper bounds (vgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size [bytes per lane])
BOUNDS = {
    # LDS-resident 2D kernels: one 12-wave workgroup per CU -> 168 VGPRs is the cap for 3 waves per SIMD
    # (the production instantiations: DBG = false, no MODE_TENDENCY hook)
    "rbc::rbc2d_kernel<96, 64, double, false>": (168, 0, 48, 0),             # no scratch at all since round 3
    # packed float32 pairs: 20 spilled VGPRs / 44 B scratch -> NONE in round 4 (uniform constants as float kernel arguments, nu / kappa
    # pairs in LDS, the output block reads the final fields from LDS instead of holding three 8-row arrays of pairs across both lanes)
    "rbc::rbc2d_kernel<96, 64, float __vector(2), false>": (168, 0, 96, 0),
    # 3D tendency tiles, configs[4]'s 48 x 48 planes as compile-time constants (12 waves, 3 per SIMD).  NO spill: a reload from
    # scratch shares vmcnt with the planes prefetched one level ahead and waits for them -- six spilled VGPRs cost 7 % of the
    # env-step rate until round 3 (NOTES.md section 5b, scripts/tile_stamps.py)
    "rbc3::k3_tile_all<16, 16, 2, 768, 3, 64, false, 48, 48>": (168, 0, 0, 0),
    "rbc3::k3_tile_all<16, 16, 2, 768, 3, 64, false, 32, 32>": (168, 0, 0, 0),
    "rbc3::k3_tile_all<16, 16, 2, 768, 3, 64, false, 0, 0>": (168, 0, 0, 0),
    "rbc3f::k3_tile_all<16, 16, 2, 768, 3, 64, false, 48, 48>": (168, 0, 0, 0),
    "rbc3::k3_tile_all<16, 4, 2, 768, 3, 64, false, 48, 48>": (168, 0, 0, 0),
    "rbc3f::k3_tile_all<16, 8, 2, 768, 3, 64, false, 48, 48>": (168, 0, 0, 0),      # what configs[4] runs in float32 since round 4 (B = 32: 16 x 8 tiles)
    # streaming-2D: FLAT tiles and the one-kernel projection at 128 x 64 (N1 = 16, two workgroups per CU)
    "rbc3::k3_tile_all<1, 16, 1, 256, 3, 256, true, 0, 0>": (128, 0, 0, 0),
    "rbc3::k2s_project_fused<16>": (104, 0, 0, 0),
    # projection kernels of the 3D path
    "rbc3::k3_rhs_fft_pair": (64, 0, 0, 0),
    "rbc3::k3_ifft_pair": (72, 0, 0, 0),
    "rbc3::k3_thomas_pair_fused<16>": (256, 0, 0, 0),
    # inverse FFT + the whole correction (round 4, float64 default at configs[4]: 6 columns per thread): compiled for two waves per
    # SIMD -- a launch has 64 workgroups per env group, residency is no constraint -- so that everything a pair's correction reads is
    # prefetched before its transform; no spill to memory (its spilled SGPRs live in VGPR lanes)
    "rbc3::k3_ifft_march<2, 6>": (256, 0, 80, 0),
    "rbc3::k3_ifft_march<2, 4>": (256, 0, 80, 0),
    # generic two-factor DFT row kernels of the streaming-2D mode (grids like 100 x 40 whose nx is not 8 * {4 ... 32}): today's
    # numbers, scratch included -- a rarely taken path, pinned so that it does not get worse unnoticed
    "rbc3::k2s_rhs_fft_pair": (168, 0, 0, 216),
    "rbc3::k2s_ifft_pair": (168, 0, 0, 216),
}


@pytest.fixture(scope="module")
def table():
    return kr.kernel_resources(LIB)


def test_one_gfx950_code_object_with_every_hot_kernel(table):
    missing = [k for k in BOUNDS if k not in table]
    assert not missing, (missing, sorted(table)[:10])


@pytest.mark.parametrize("kernel", sorted(BOUNDS))
def test_register_and_scratch_budget(table, kernel):
    vgpr, vspill, sspill, scratch = BOUNDS[kernel]
    r = table[kernel]
    got = (r["vgpr_count"], r["vgpr_spill_count"], r["sgpr_spill_count"], r["private_segment_fixed_size"])
    assert r["vgpr_count"] <= vgpr and r["vgpr_spill_count"] <= vspill and r["sgpr_spill_count"] <= sspill and \
        r["private_segment_fixed_size"] <= scratch, (kernel, "vgpr, vgpr spills, sgpr spills, scratch bytes:", got, "bounds", BOUNDS[kernel])
    assert r["agpr_count"] == 0            # no MFMA anywhere on this path (fp64 vector peak = fp64 MFMA peak on gfx950; FFT, not DFT-GEMM)
