#!/bin/bash
# 3D env-step rate against the number of env groups (streams), graph replay and experiment knobs; run on the GPU box.
run() { echo -n "$*: "; env "$@" timeout -k 10 200 python bench.py --dim 3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms', 'nan', d['nan_envs'], 'Nu', round(d['mean_nusselt'],6))" || exit 1; }
for q in 8 16; do
for g in 4 6 8; do run GPU_MAX_HW_QUEUES=$q RBC_3D_GROUPS=$g; done
run GPU_MAX_HW_QUEUES=$q RBC_3D_GROUPS=8 RBC_TILE_SHAPE=16x8
run GPU_MAX_HW_QUEUES=$q RBC_3D_GROUPS=8 RBC_TILE_SHAPE=16x4
done
