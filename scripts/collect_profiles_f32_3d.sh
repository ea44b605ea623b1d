set -o pipefail
ROOT=$(pwd); TAG=${1:-r03b}; OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
B3F="python3 $ROOT/bench.py --dim 3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --precision f32"
run() { local name=$1; shift; timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$name" "$@" > "$OUT/$name.log" 2>&1 || { echo "FAILED $name" >&2; tail -5 "$OUT/$name.log" >&2; return 1; }; }
run trace3d_f32 --stats -- $B3F && grep '^{"metric"' "$OUT/trace3d_f32.log" > "$OUT/bench_trace3d_f32.log"
run pmc_fetch3d_f32 --pmc FETCH_SIZE -- $B3F
run pmc_write3d_f32 --pmc WRITE_SIZE -- $B3F
ls $OUT
