#!/bin/bash
# rocprofv3 evidence for one build, run ON THE GPU BOX from the repo root (gpurun -- 'bash scripts/collect_profiles.sh r02a'):
# kernel-trace stats and, in SEPARATE passes (the pool refuses other combinations), the HBM counters and two groups of SQ
# counters, for the 2D headline workload (bench.py, B=1024) and the 3D configs[4] workload (bench.py --dim 3, B=32).
# Everything lands in gpurun_out/prof_<tag>/; scripts/summarize_profile.py copies the judged summaries into profiles/.
set -o pipefail
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B2="python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra"
B3="python3 $ROOT/bench.py --dim 3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
run() { # name, extra rocprof args..., -- command
  local name=$1; shift
  echo "== $name" >&2
  timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$name" "$@" > "$OUT/$name.log" 2>&1 || { echo "FAILED $name" >&2; tail -5 "$OUT/$name.log" >&2; return 1; }
}
ONLY=${2:-all}          # "3d": only the configs[4] passes
if [ "$ONLY" != "3d" ]; then
run trace --stats -- $B2 && grep '^{"metric"' "$OUT/trace.log" > "$OUT/bench_trace.log"
run pmc_fetch --pmc FETCH_SIZE -- $B2
run pmc_write --pmc WRITE_SIZE -- $B2
run pmc_sq1 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES -- $B2
run pmc_sq2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY -- $B2
BF="python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --precision f32"
run trace_f32 --stats -- $BF && grep '^{"metric"' "$OUT/trace_f32.log" > "$OUT/bench_trace_f32.log"
run pmc_f32 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- $BF
fi
run trace3d --stats -- $B3 && grep '^{"metric"' "$OUT/trace3d.log" > "$OUT/bench_trace3d.log"
run pmc_fetch3d --pmc FETCH_SIZE -- $B3
run pmc_write3d --pmc WRITE_SIZE -- $B3
# one chain on the handle's stream: per-kernel times and counters without the other chains' kernels in between
export RBC_3D_GROUPS=1
run trace3d_g1 --stats -- $B3 && grep '^{"metric"' "$OUT/trace3d_g1.log" > "$OUT/bench_trace3d_g1.log"
run pmc_fetch3d_g1 --pmc FETCH_SIZE -- $B3
run pmc_write3d_g1 --pmc WRITE_SIZE -- $B3
unset RBC_3D_GROUPS
# the float32 instantiation of the same kernels (bench.py --precision f32 --dim 3)
B3F="$B3 --precision f32"
run trace3d_f32 --stats -- $B3F && grep '^{"metric"' "$OUT/trace3d_f32.log" > "$OUT/bench_trace3d_f32.log"
run pmc_fetch3d_f32 --pmc FETCH_SIZE -- $B3F
run pmc_write3d_f32 --pmc WRITE_SIZE -- $B3F
# streaming 2D at 128x64 (no LDS-resident float64 kernel): 1 warm-up + 3 env-steps of 1024 envs
S2="python3 $ROOT/scripts/stream2d_timing.py 1024 3 128 64"
run trace_s2d --stats -- $S2
run pmc_fetch_s2d --pmc FETCH_SIZE -- $S2
run pmc_write_s2d --pmc WRITE_SIZE -- $S2
ls "$OUT"
