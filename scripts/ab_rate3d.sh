#!/bin/bash
# A/B of configs[4] device-resident rates between library builds on ONE box, interleaved: scripts/ab_rate3d.sh libA.so libB.so [reps]
# (box-to-box spread of this path is +-3 %, so only same-box interleaved repeats decide a few per cent)
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in "$A" "$B"; do
    for p in f64 f32; do echo -n "$(basename $L) "; RBC_HIP_LIB=$L python scripts/rate_3d.py $p 2>&1 | grep -v amdgpu.ids; done
  done
done
