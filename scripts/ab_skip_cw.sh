# Timing bound of removing k3_correct_w (WRONG numerics).  Needs an experiments build and a baseline build:
#   python -c 'import __graft_entry__ as g; g.build_hip(True, ("-DRBC_EXPERIMENTS=1",), out="build/librbc_hip_exp.so")'
#   (baseline: the same command without -DRBC_EXPERIMENTS=1 on the commit to compare with -> build/librbc_hip_head.so)
for i in 1 2 3; do
  for v in 0 1; do for p in f64 f32; do echo -n "skip_cw=$v "; RBC_EXPERIMENT_SKIP_CW=$v RBC_HIP_LIB=build/librbc_hip_exp.so python scripts/rate_3d.py $p 2>&1 | grep -v amdgpu.ids; done; done
done
for p in f64 f32; do echo -n "head "; RBC_HIP_LIB=build/librbc_hip_head.so python scripts/rate_3d.py $p 2>&1 | grep -v amdgpu.ids; done
