# One-env-per-workgroup float32 2D kernels (RBC_F32_SCALAR=1 at 96x64; 128x64 and 192x32) between a baseline build (build/librbc_hip_h3.so:
# the same hipcc command as __graft_entry__.build_hip on the commit to compare with) and the in-tree library, interleaved on one box.
for i in 1 2 3; do for L in build/librbc_hip_h3.so rbc-gym_amd/lib/librbc_hip.so; do
  echo -n "$(basename $L) "; RBC_HIP_LIB=$L python scripts/rate_2d.py f32 1024 128 64 2>&1 | grep -v amdgpu
  echo -n "$(basename $L) "; RBC_HIP_LIB=$L python scripts/rate_2d.py f32 1024 192 32 2>&1 | grep -v amdgpu
  echo -n "$(basename $L) scalar "; RBC_F32_SCALAR=1 RBC_HIP_LIB=$L python scripts/rate_2d.py f32 1024 2>&1 | grep -v amdgpu
done; done
