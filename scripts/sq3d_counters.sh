# SQ counters of the 3D configs[4] kernels (two --pmc passes of 8 counters; pool rule: --pmc only with --kernel-trace), one chain
# (RBC_3D_GROUPS=1) so that a kernel's counters are its own.  usage (GPU box, repo root): bash scripts/sq3d_counters.sh [f64|f32] [tag]
set -o pipefail
PREC=${1:-f64}; TAG=${2:-sq3d_$PREC}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
export RBC_3D_GROUPS=1
B3="python3 $ROOT/bench.py --dim 3 --steps 3 --warmup 1 --no-cpu-baseline --no-extra --precision $PREC"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -- $B3 > $OUT/p1.log 2>&1 || tail -3 $OUT/p1.log
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM -- $B3 > $OUT/p2.log 2>&1 || tail -3 $OUT/p2.log
ls $OUT/*/
