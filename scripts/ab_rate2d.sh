#!/bin/bash
# A/B of the resident 2D kernel between library builds on ONE box, interleaved: scripts/ab_rate2d.sh libA.so libB.so [reps]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do for L in "$A" "$B"; do for p in f64 f32; do echo -n "$(basename $L) "; RBC_HIP_LIB=$L python scripts/rate_2d.py $p 2>&1 | grep -v amdgpu.ids; done; done; done
