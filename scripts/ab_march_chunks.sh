# Chunk-length sweep of k3_ifft_march (1 / 2 / 4 slab pairs per workgroup).  Needs build/librbc_hip_exp.so built with -DRBC_EXPERIMENTS=1
# (python -c 'import __graft_entry__ as g; g.build_hip(True, ("-DRBC_EXPERIMENTS=1",), out="build/librbc_hip_exp.so")').
for i in 1 2 3; do
  for m in 1 2 4; do for p in f64 f32; do echo -n "march=$m "; RBC_HIP_LIB=build/librbc_hip_exp.so RBC_IFFT_MARCH=$m python scripts/rate_3d.py $p 2>&1 | grep -v amdgpu.ids; done; done
done
