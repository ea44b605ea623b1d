# configs[4] rate against the number of env groups (stream chains) and against the batch size, in-tree library.
for i in 1 2; do for g in 2 3 4; do for p in f64 f32; do echo -n "groups=$g "; RBC_3D_GROUPS=$g python scripts/rate_3d.py $p 2>&1 | grep -v amdgpu.ids; done; done; done
for B in 64 128; do for p in f64 f32; do echo -n "B=$B "; python scripts/rate_3d.py $p $B 2>&1 | grep -v amdgpu.ids; done; done
