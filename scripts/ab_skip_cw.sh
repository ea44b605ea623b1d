# Timing bound of removing k3_correct_w (WRONG numerics).  Needs an experiments build and a baseline build:
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DRBC_EXPERIMENTS=1 -o build/librbc_hip_exp.so rbc-gym_amd/csrc/rbc_api.hip
#   (baseline: the same command without -DRBC_EXPERIMENTS=1 on the commit to compare with -> build/librbc_hip_head.so)
for i in 1 2 3; do
  for v in 0 1; do for p in f64 f32; do echo -n "skip_cw=$v "; RBC_EXPERIMENT_SKIP_CW=$v RBC_HIP_LIB=build/librbc_hip_exp.so python scripts/rate_3d.py $p 2>&1 | grep -v amdgpu.ids; done; done
done
for p in f64 f32; do echo -n "head "; RBC_HIP_LIB=build/librbc_hip_head.so python scripts/rate_3d.py $p 2>&1 | grep -v amdgpu.ids; done
