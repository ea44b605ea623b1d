#!/bin/bash
# per-kernel average durations of the configs[4] step (rocprofv3 serialises the chains: kernels alone) for one library build and precision:
# scripts/kernel_times_3d.sh lib.so f64|f32 tag
LIB=$1; P=$2; TAG=$3; ROOT=$(pwd); OUT=$ROOT/gpurun_out/kt_$TAG
cd /tmp && export TMPDIR=/tmp
RBC_HIP_LIB=$ROOT/$LIB timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --dim 3 --steps 5 --warmup 2 --no-cpu-baseline --no-extra --precision $P > $OUT.log 2>&1
cd $ROOT && python3 scripts/kernel_stats_top.py $OUT
rm -rf $OUT
