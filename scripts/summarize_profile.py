#!/usr/bin/env python3
"""Copy the judged rocprofv3 summaries from gpurun_out/prof_<tag> (scripts/collect_profiles.sh) into profiles/ (tracked).

usage: python scripts/summarize_profile.py gpurun_out/prof_r02a r02a
Writes profiles/<tag>_kernel_stats.csv, <tag>_summary.json (2D step kernel: trace durations, register / LDS footprint,
FETCH_SIZE / WRITE_SIZE per launch), <tag>_sq_counters.json (SQ instruction and activity counters per launch),
<tag>_3d_kernel_stats.csv and <tag>_3d_summary.json (per-kernel HBM counters of the configs[4] env-step).
bench.py attaches the counter-derived numbers only when `workload_key` matches the run it is printing.
"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    f = glob.glob(pattern)
    return f[0] if f else None


def bench_line(path):
    if path and os.path.exists(path):
        for line in open(path):
            if line.startswith('{"metric"'):
                return json.loads(line)
    return None


def counters(dirname, match):
    """{counter: {kernel: [values per dispatch]}} of one --pmc pass"""
    f = one(f"{src}/{dirname}/*/*_counter_collection.csv")
    res = defaultdict(lambda: defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                res[r["Counter_Name"]][r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return res


# ---------------------------------------------------------------- 2D headline workload
ks = one(f"{src}/trace/*/*_kernel_stats.csv")
if ks:
    shutil.copy(ks, f"{out}/{tag}_kernel_stats.csv")
    kt = one(f"{src}/trace/*/*_kernel_trace.csv")
    rows = [r for r in csv.DictReader(open(kt)) if "rbc2d_kernel" in r["Kernel_Name"]]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    step = [d for d in dur if d > 0.25 * max(dur)]          # drops the (short) reset launch
    line = bench_line(f"{src}/bench_trace.log")
    key = {"dim": 2, "batch": 1024, "ra": 1e4, "precision": (line or {}).get("dtype", "f64"), "ra_sweep": None}
    summ = {"workload_key": key, "kernel": rows[0]["Kernel_Name"], "launches": len(dur), "step_launches": len(step),
            "step_avg_ms": sum(step) / len(step), "step_min_ms": min(step), "step_max_ms": max(step),
            # the first profiled launch carries the tracer's start-up (14.3 against 12.1 ms in round 3): the median and the average
            # without it are what the bench line's HIP-event average has to agree with
            "step_median_ms": sorted(step)[len(step) // 2], "step_avg_ms_without_first": (sum(step[1:]) / (len(step) - 1)) if len(step) > 1 else step[0],
            "vgpr": int(rows[0]["VGPR_Count"]), "sgpr": int(rows[0]["SGPR_Count"]), "lds_bytes": int(rows[0]["LDS_Block_Size"]),
            "scratch_bytes_per_lane": int(rows[0]["Scratch_Size"]), "workgroup": int(rows[0]["Workgroup_Size_X"]), "grid": int(rows[0]["Grid_Size_X"])}
    pmc = {}
    for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        c = counters(name, "rbc2d_kernel").get(ctr)
        if not c:
            continue
        v = [x for vals in c.values() for x in vals]
        v = [x for x in v if x > 0.25 * max(v)]
        pmc[ctr] = {"launches": len(v), "mean_KB": sum(v) / len(v), "min_KB": min(v), "max_KB": max(v)}
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # units: KB (guide: hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024).  Calibration of FETCH_SIZE for THIS access pattern
        # (8-byte-per-lane coalesced row loads): the kernel must read 3*NZ*NX*8 B of state per env; see DESIGN.md section 5.
        summ["hbm_traffic_bytes_per_launch"] = (pmc["FETCH_SIZE"]["mean_KB"] + pmc["WRITE_SIZE"]["mean_KB"]) * 1024
    summ["pmc"] = pmc
    if line:
        summ["bench_line_under_profiler"] = line
    json.dump(summ, open(f"{out}/{tag}_summary.json", "w"), indent=1)
    print(json.dumps({k: v for k, v in summ.items() if k != "bench_line_under_profiler"}, indent=1))

    sq = {}
    for name in ("pmc_sq1", "pmc_sq2"):
        for ctr, per_kernel in counters(name, "rbc2d_kernel").items():
            v = [x for vals in per_kernel.values() for x in vals]
            v = [x for x in v if x > 0.25 * max(v)] if max(v) > 0 else v
            sq[ctr] = sum(v) / len(v)
    if sq:
        ms = summ["step_avg_ms"]
        nthr_stage = 1024 * 768 / 64 * 150                  # waves per launch x stages: per-thread-and-stage normaliser
        derived = {}
        if "SQ_INSTS_VALU" in sq:
            f64 = sq.get("SQ_INSTS_VALU_ADD_F64", 0) + sq.get("SQ_INSTS_VALU_FMA_F64", 0) + sq.get("SQ_INSTS_VALU_MUL_F64", 0)
            derived["valu_wave_instructions_per_thread_and_stage"] = sq["SQ_INSTS_VALU"] / nthr_stage
            derived["fp64_share_of_valu"] = f64 / sq["SQ_INSTS_VALU"]
            derived[f"fp64_issue_fraction (4 cycles per wave64 instruction, 1024 SIMDs, {ms:.2f} ms at 2.4 GHz)"] = f64 * 4 / (1024 * ms * 1e-3 * 2.4e9)
            derived["all_valu_issue_fraction"] = sq["SQ_INSTS_VALU"] * 4 / (1024 * ms * 1e-3 * 2.4e9)
        if "SQ_LDS_IDX_ACTIVE" in sq:
            derived["lds_array_active_fraction (SQ_LDS_IDX_ACTIVE / 256 CUs x cycles)"] = sq["SQ_LDS_IDX_ACTIVE"] / (256 * ms * 1e-3 * 2.4e9)
            if "SQ_LDS_BANK_CONFLICT" in sq:
                derived["lds_bank_conflict_share_of_lds_cycles"] = sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]
        json.dump({"workload_key": key, "command": "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass> -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra",
                   "kernel": summ["kernel"] + ", B=1024 (one launch = 150 stages x 1024 workgroups)", "kernel_ms_in_trace_pass": ms,
                   "per_launch_mean": sq, "derived": derived}, open(f"{out}/{tag}_sq_counters.json", "w"), indent=1)
        print(json.dumps(derived, indent=1))

# ---------------------------------------------------------------- float32 (packed) variant of the 2D workload
ksf = one(f"{src}/trace_f32/*/*_kernel_stats.csv")
if ksf:
    shutil.copy(ksf, f"{out}/{tag}_f32_kernel_stats.csv")
    ktf = one(f"{src}/trace_f32/*/*_kernel_trace.csv")
    rows = [r for r in csv.DictReader(open(ktf)) if "rbc2d_kernel" in r["Kernel_Name"]]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    step = [d for d in dur if d > 0.25 * max(dur)]
    sq = {}
    for ctr, per_kernel in counters("pmc_f32", "rbc2d_kernel").items():
        v = [x for vals in per_kernel.values() for x in vals]
        v = [x for x in v if x > 0.25 * max(v)] if max(v) > 0 else v
        sq[ctr] = sum(v) / len(v)
    ms = sum(step) / len(step)
    summf = {"workload_key": {"dim": 2, "batch": 1024, "ra": 1e4, "precision": "f32", "ra_sweep": None}, "kernel": rows[0]["Kernel_Name"],
             "step_launches": len(step), "step_avg_ms": ms, "vgpr": int(rows[0]["VGPR_Count"]), "scratch_bytes_per_lane": int(rows[0]["Scratch_Size"]),
             "workgroup": int(rows[0]["Workgroup_Size_X"]), "grid": int(rows[0]["Grid_Size_X"]), "sq_per_launch_mean": sq,
             "bench_line_under_profiler": bench_line(f"{src}/bench_trace_f32.log")}
    if "SQ_INSTS_VALU" in sq:
        summf["all_valu_issue_fraction (4 cycles per wave64 instruction)"] = sq["SQ_INSTS_VALU"] * 4 / (1024 * ms * 1e-3 * 2.4e9)
    json.dump(summf, open(f"{out}/{tag}_f32_summary.json", "w"), indent=1)
    print(json.dumps({k: v for k, v in summf.items() if k != "bench_line_under_profiler"}, indent=1))

# ---------------------------------------------------------------- streaming paths: 3D configs[4] (default chains, one chain, float32), streaming 2D
def streaming(trace, fetch, write, dst, key, steps_default, bench_log, what):
    ks3 = one(f"{src}/{trace}/*/*_kernel_stats.csv")
    if not ks3:
        return
    shutil.copy(ks3, f"{out}/{tag}_{dst}_kernel_stats.csv")
    line3 = bench_line(f"{src}/{bench_log}") if bench_log else None
    steps = ((line3 or {}).get("steps", 0) + (line3 or {}).get("warmup", 0)) or steps_default
    per_kernel = {}
    for r in csv.DictReader(open(ks3)):
        if "rbc3" in r["Name"]:
            per_kernel[r["Name"].split("(")[0].replace("void ", "")] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                                       "total_ms": float(r["TotalDurationNs"]) / 1e6}
    tot = {}
    for name, ctr in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
        c = counters(name, "rbc3").get(ctr, {})
        for kern, vals in c.items():
            k = kern.split("(")[0].replace("void ", "")
            per_kernel.setdefault(k, {})[f"{ctr}_KB_per_call"] = sum(vals) / len(vals)
            per_kernel[k][f"{ctr}_KB_total"] = sum(vals)
        if c:
            tot[ctr] = sum(x for vals in c.values() for x in vals)
    summ3 = {"workload_key": key, "what": what, "env_steps_profiled": steps, "kernels": per_kernel,
             "kernel_time_ms_per_env_step_batch": sum(k.get("total_ms", 0.0) for k in per_kernel.values()) / steps,
             "note": "FETCH_SIZE / WRITE_SIZE in KB, raw counter values (separate --pmc passes); totals cover reset + all profiled env-steps of the batch; "
                     "under rocprofv3 concurrent stream chains serialise, so per-kernel times are those of the kernels alone, not of the overlapped run"}
    if len(tot) == 2:
        summ3["hbm_traffic_bytes_per_env_step_batch"] = (tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / steps
        summ3["fetch_bytes_per_env_step_batch"] = tot["FETCH_SIZE"] * 1024 / steps
        summ3["write_bytes_per_env_step_batch"] = tot["WRITE_SIZE"] * 1024 / steps
    if line3:
        summ3["bench_line_under_profiler"] = line3
        alg = (line3.get("roofline") or {}).get("algorithmic_bytes_per_launch")
        if alg and "hbm_traffic_bytes_per_env_step_batch" in summ3:
            summ3["traffic_over_algorithmic"] = summ3["hbm_traffic_bytes_per_env_step_batch"] / alg
    json.dump(summ3, open(f"{out}/{tag}_{dst}_summary.json", "w"), indent=1)
    print(dst, json.dumps({k: v for k, v in summ3.items() if k not in ("bench_line_under_profiler", "kernels")}, indent=1)[:1500])


streaming("trace3d", "pmc_fetch3d", "pmc_write3d", "3d", {"dim": 3, "batch": 32, "ra": 1e4, "precision": "f64"}, 7, "bench_trace3d.log",
          "configs[4] float64, default env groups (four stream chains, one graph)")
streaming("trace3d_g1", "pmc_fetch3d_g1", "pmc_write3d_g1", "3d_g1", {"dim": 3, "batch": 32, "ra": 1e4, "precision": "f64", "groups": 1}, 7, "bench_trace3d_g1.log",
          "configs[4] float64, RBC_3D_GROUPS=1: one chain on the handle's stream (clean per-kernel counters)")
streaming("trace3d_f32", "pmc_fetch3d_f32", "pmc_write3d_f32", "3d_f32", {"dim": 3, "batch": 32, "ra": 1e4, "precision": "f32"}, 7, "bench_trace3d_f32.log",
          "configs[4] float32 (rbc3f kernels)")
streaming("trace_s2d", "pmc_fetch_s2d", "pmc_write_s2d", "stream2d_128x64", {"dim": 2, "batch": 1024, "nx": 128, "nz": 64, "precision": "f64"}, 4, None,
          "streaming 2D, 128x64 float64, B = 1024, 1 warm-up + 3 env-steps (scripts/stream2d_timing.py 1024 3 128 64)")
