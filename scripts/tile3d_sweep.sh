#!/bin/bash
for tv in ${TILES:-x b c d e f}; do
  for g in ${GROUPS_:-2 4 8}; do
    echo -n "tile=$tv groups=$g: "
    RBC_EXPERIMENT_TILE=$tv RBC_3D_GROUPS=$g RBC_USE_GRAPH=1 timeout -k 10 200 python bench.py --dim 3 --steps 10 --warmup 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), 'env-steps/s', round(d['ms_per_step'],3), 'ms', 'nan', d['nan_envs'], 'Nu', round(d['mean_nusselt'],6))" || exit 1
  done
done
