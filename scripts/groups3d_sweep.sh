#!/bin/bash
# 3D (configs[4]) env-step rate against the runtime knobs of the streaming path; run on the GPU box:
#   gpurun -- 'bash scripts/groups3d_sweep.sh'
# RBC_3D_GROUPS (env groups / streams), RBC_USE_GRAPH (0/1), RBC_TILE_SHAPE (16x16|16x8|16x4|8x8), RBC_NO_CONST_GRID=1.
run() { echo -n "$*: "; env "$@" timeout -k 10 200 python bench.py --dim 3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms', 'nan', d['nan_envs'], 'Nu', round(d['mean_nusselt'],6))" || exit 1; }
for g in 1 2 3 4 5 8; do run RBC_3D_GROUPS=$g; done
run RBC_3D_GROUPS=4 RBC_USE_GRAPH=0
for t in 16x16 16x8 16x4; do run RBC_TILE_SHAPE=$t; done
run RBC_NO_CONST_GRID=1
