// Explicit instantiations of the LDS-resident 2D kernel; see rbc2d_instances.hpp for why they live in a translation unit of their own.
#include <hip/hip_runtime.h>

#define RBC2D_TEMPLATE_ONLY 1
#include "rbc2d_instances.hpp"

#define RBC2D_DEFINE(NX, NZ, T)                                                              \
    template __global__ void rbc::rbc2d_kernel<NX, NZ, T, false>(const rbc::Params2D); \
    template __global__ void rbc::rbc2d_kernel<NX, NZ, T, true>(const rbc::Params2D);
#define RBC2D_DEFINE_PROD(NX, NZ, T) template __global__ void rbc::rbc2d_kernel<NX, NZ, T, false>(const rbc::Params2D);
RBC2D_INSTANCES_F64(RBC2D_DEFINE)
RBC2D_INSTANCES_F32(RBC2D_DEFINE)
RBC2D_INSTANCES_F32X2(RBC2D_DEFINE_PROD)
