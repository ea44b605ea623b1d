#!/bin/bash
# 3D env-step rate against experiment knobs; run on the GPU box.
run() { echo -n "$*: "; env "$@" timeout -k 10 200 python bench.py --dim 3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms', 'nan', d['nan_envs'], 'Nu', round(d['mean_nusselt'],6))" || exit 1; }
for i in 1 2; do
run A=1
run RBC_HIP_LIB=$PWD/rbc-gym_amd/lib/librbc_hip_u2.so
run RBC_HIP_LIB=$PWD/rbc-gym_amd/lib/librbc_hip_u3.so
done
