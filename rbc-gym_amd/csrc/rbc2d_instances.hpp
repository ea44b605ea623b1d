// The instantiations of the LDS-resident 2D kernel (rbc2d_kernel.hpp), listed once: rbc2d_instances.hip DEFINES them -- a translation
// unit of its own, compiled with `-mllvm --amdgpu-sched-strategy=iterative-maxocc` (the scheduler that wins on these long,
// register-bound kernels: float64 96x64 +0.5 %, packed float32 +0.75 %, 885 -> ~600 hazard nops between dependent v_pk_* in the stage
// loop; the streaming kernels lose 5 % under it and keep the default) -- and rbc_api.hip DECLARES them extern and binds them
// (bind_grid).  X(NX, NZ, T): production kernel, and for one-env-per-workgroup types the instantiation with the tendency hook.
#pragma once
#include "rbc2d_kernel.hpp"

#define RBC2D_INSTANCES_F64(X) X(96, 64, double) X(96, 48, double) X(96, 32, double) X(64, 64, double) X(64, 48, double) X(64, 32, double) X(128, 32, double)
#define RBC2D_INSTANCES_F32(X) X(96, 64, float) X(128, 64, float) X(64, 64, float) X(192, 32, float)
#define RBC2D_INSTANCES_F32X2(X) X(96, 64, rbc::f32x2) X(64, 64, rbc::f32x2)
