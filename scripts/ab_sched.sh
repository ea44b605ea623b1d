#!/bin/bash
# A/B of whole-library builds (e.g. -mllvm --amdgpu-sched-strategy=...) on ONE box, interleaved: 2D resident f64 / packed f32, configs[4] f64 / f32
# scripts/ab_sched.sh libA.so libB.so [libC.so] -- three rounds
for i in 1 2 3; do for L in "$@"; do
  for p in f64 f32; do echo -n "$(basename $L) 2D "; RBC_HIP_LIB=$L python scripts/rate_2d.py $p 2>&1 | grep -v amdgpu.ids; done
  for p in f64 f32; do echo -n "$(basename $L) 3D $p "; RBC_HIP_LIB=$L python scripts/rate_3d.py $p 2>&1 | grep -v amdgpu.ids | tail -1; done
done; done
