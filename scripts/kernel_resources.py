#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel in librbc_hip.so, read from the metadata of its gfx950 code objects (one per translation unit).

    python scripts/kernel_resources.py [path/to/librbc_hip.so] [substring ...]

The .so carries a clang offload bundle in its .hip_fatbin section; llvm-objcopy dumps it, clang-offload-bundler takes the
hipv4-amdgcn-amd-amdhsa--gfx950 entry out, llvm-readelf --notes prints the AMDGPU metadata (one YAML map per kernel).  All
three tools ship with ROCm (/opt/rocm/lib/llvm/bin); nothing is executed on a GPU.  tests/test_kernel_resources.py pins
today's numbers of the hot kernels as upper bounds, so that a compiler or source change that starts spilling fails on the
CPU box instead of showing up as a few per cent on the GPU.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
        "group_segment_fixed_size", "max_flat_workgroup_size")


def demangle(names):
    import shutil
    filt = os.path.join(LLVM, "llvm-cxxfilt")
    if not os.path.exists(filt):
        filt = shutil.which("c++filt")
    if not filt:
        return list(names)
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return out.splitlines()


def kernel_resources(lib=None):
    """-> {demangled kernel name: {key: int}} for the gfx950 code object inside `lib`."""
    lib = lib or os.path.join(ROOT, "rbc-gym_amd", "lib", "librbc_hip.so")
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", lib])
        # one offload bundle per translation unit of the library (rbc2d_instances.hip, rbc_api.hip), back to back in the section
        blob, magic = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        notes = ""
        for n, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
            part, co = os.path.join(tmp, f"fat{n}.bin"), os.path.join(tmp, f"k{n}.co")
            open(part, "wb").write(blob[a:b])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                                   f"--targets={TARGET}", f"--output={co}"])
            notes += subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout + "\n"
    kernels, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*(- )?\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        first, key, val = m.group(1), m.group(2), m.group(3).strip()
        if key == "agpr_count" and first:            # first key of a kernel's map (keys are sorted)
            cur = {}
            kernels.append(cur)
        if cur is None:
            continue
        if key == "name":
            cur["name"] = val.strip("'\"")
        elif key in KEYS:
            cur[key] = int(val)
    kernels = [k for k in kernels if "name" in k]
    names = demangle([k["name"] for k in kernels])
    def short(n):                                 # drop the return type and the parameter list, keep template arguments
        n = n.replace("void ", "")
        depth = 0
        for i, c in enumerate(n):
            depth += (c == "<") - (c == ">")
            if c == "(" and depth == 0:
                return n[:i].strip()
        return n.strip()
    return {short(n): {q: k.get(q, 0) for q in KEYS} for n, k in zip(names, kernels)}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else None
    pats = [a for a in sys.argv[1:] if not a.endswith(".so")]
    res = kernel_resources(lib)
    print(f"{'kernel':78s} vgpr agpr sgpr vspill sspill scratch   lds")
    for name, r in res.items():
        if pats and not any(p in name for p in pats):
            continue
        print(f"{name[:78]:78s} {r['vgpr_count']:4d} {r['agpr_count']:4d} {r['sgpr_count']:4d} {r['vgpr_spill_count']:6d} {r['sgpr_spill_count']:6d} "
              f"{r['private_segment_fixed_size']:7d} {r['group_segment_fixed_size']:5d}")
