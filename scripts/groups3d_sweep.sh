#!/bin/bash
# 3D env-step rate against experiment knobs; run on the GPU box.
run() { echo -n "$*: "; env "$@" timeout -k 10 200 python bench.py --dim 3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms', 'nan', d['nan_envs'], 'Nu', round(d['mean_nusselt'],6))" || exit 1; }
for i in 1 2 3; do
run RBC_EXPERIMENT_NO_NXC=1
run A=1
done
run RBC_3D_GROUPS=1 RBC_EXPERIMENT_NO_NXC=1
run RBC_3D_GROUPS=1
