#!/usr/bin/env python3
"""Copy the judged rocprofv3 summaries from gpurun_out/<dir> into profiles/ (tracked).

usage: python scripts/summarize_profile.py gpurun_out/prof_r1 r01
Expects <dir>/trace (--kernel-trace --stats), <dir>/pmc_fetch (--pmc FETCH_SIZE) and
<dir>/pmc_write (--pmc WRITE_SIZE), each produced by its own rocprofv3 pass over bench.py.
"""
import csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
ks = glob.glob(f"{src}/trace/*/*_kernel_stats.csv")[0]
shutil.copy(ks, f"{out}/{tag}_kernel_stats.csv")
kt = glob.glob(f"{src}/trace/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(kt)) if "rbc2d_kernel" in r["Kernel_Name"]]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
step = [d for d in dur if d > 0.25 * max(dur)]          # drops the (short) reset launch
summ = {"kernel": rows[0]["Kernel_Name"], "launches": len(dur), "step_launches": len(step),
        "step_avg_ms": sum(step) / len(step), "step_min_ms": min(step), "step_max_ms": max(step),
        "vgpr": int(rows[0]["VGPR_Count"]), "sgpr": int(rows[0]["SGPR_Count"]), "lds_bytes": int(rows[0]["LDS_Block_Size"]),
        "scratch_bytes_per_lane": int(rows[0]["Scratch_Size"]), "workgroup": int(rows[0]["Workgroup_Size_X"]), "grid": int(rows[0]["Grid_Size_X"])}
pmc = {}
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = glob.glob(f"{src}/{name}/*/*_counter_collection.csv")
    if not f:
        continue
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "rbc2d_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr]
    v = [x for x in v if x > 0.25 * max(v)]
    pmc[ctr] = {"launches": len(v), "mean_KB": sum(v) / len(v), "min_KB": min(v), "max_KB": max(v)}
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # units: KB (guide: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024).  Calibration of FETCH_SIZE for THIS access pattern
    # (8-byte-per-lane coalesced row loads): the kernel must read 3*NZ*NX*8 B of state per env; see DESIGN.md.
    summ["hbm_traffic_bytes_per_launch"] = (pmc["FETCH_SIZE"]["mean_KB"] + pmc["WRITE_SIZE"]["mean_KB"]) * 1024
summ["pmc"] = pmc
bl = glob.glob(f"{src}/bench_trace.log")
if bl:
    for line in open(bl[0]):
        if line.startswith('{"metric"'):
            summ["bench_line_under_profiler"] = json.loads(line)
json.dump(summ, open(f"{out}/{tag}_summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in summ.items() if k != "bench_line_under_profiler"}, indent=1))
