# configs[4] rate for the tile shapes x env groups, in-tree library (RBC_TILE_SHAPE, RBC_3D_GROUPS), two interleaved repeats.
for i in 1 2; do for sh in 16x16 16x8 16x4; do for g in 4 2; do for p in f64 f32; do echo -n "shape=$sh groups=$g "; RBC_TILE_SHAPE=$sh RBC_3D_GROUPS=$g python scripts/rate_3d.py $p 2>&1 | grep -v amdgpu.ids; done; done; done; done
