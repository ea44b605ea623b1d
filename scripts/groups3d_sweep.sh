#!/bin/bash
# 3D env-step rate against the number of env groups (streams) and graph replay; run on the GPU box.
for g in 1 2 4 8; do
  for gr in 0 1; do
    echo -n "groups=$g graph=$gr: "
    RBC_3D_GROUPS=$g RBC_USE_GRAPH=$gr timeout -k 10 200 python bench.py --dim 3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms', 'nan', d['nan_envs'], 'Nu', round(d['mean_nusselt'],6))" || exit 1
  done
done
