# instruction-cache counters of the 2D step kernel (its stage loop is ~70 KB of code against a 64 KB instruction cache)
set -o pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_icache; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
B2="python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra ${1:+--precision $1}"
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d "$OUT/p1" --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU -- $B2 > "$OUT/p1.log" 2>&1 || { echo FAILED; tail -8 "$OUT/p1.log"; exit 1; }
f=$(find "$OUT/p1" -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "rbc2d_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:22s} {sum(v[1:]) / max(1, len(v) - 1):.4g} per launch ({len(v)} launches)")
PY
