for i in 1 2 3; do
  for v in 0 1; do for p in f64 f32; do echo -n "skip_cw=$v "; RBC_EXPERIMENT_SKIP_CW=$v RBC_HIP_LIB=build/librbc_hip_exp.so python scripts/rate_3d.py $p 2>&1 | grep -v amdgpu.ids; done; done
done
for p in f64 f32; do echo -n "head "; RBC_HIP_LIB=build/librbc_hip_head.so python scripts/rate_3d.py $p 2>&1 | grep -v amdgpu.ids; done
