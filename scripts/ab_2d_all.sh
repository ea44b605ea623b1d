#!/bin/bash
# every resident 2D instantiation between two library builds, interleaved on one box: scripts/ab_2d_all.sh libA.so libB.so
for i in 1 2 3; do for L in "$@"; do
  for g in "96 64" "96 48" "96 32" "64 64" "64 48" "64 32" "128 32"; do echo -n "$(basename $L) "; RBC_HIP_LIB=$L python scripts/rate_2d.py f64 1024 $g 2>&1 | grep -v amdgpu; done
  for g in "96 64" "64 64"; do echo -n "$(basename $L) pairs "; RBC_HIP_LIB=$L python scripts/rate_2d.py f32 1024 $g 2>&1 | grep -v amdgpu; done
  for g in "128 64" "192 32"; do echo -n "$(basename $L) "; RBC_HIP_LIB=$L python scripts/rate_2d.py f32 1024 $g 2>&1 | grep -v amdgpu; done
  echo -n "$(basename $L) scalar "; RBC_F32_SCALAR=1 RBC_HIP_LIB=$L python scripts/rate_2d.py f32 1024 2>&1 | grep -v amdgpu
done; done
