// rbc3d_host.hpp -- host side of the streaming path (included by rbc_api.hip after `struct rbc_handle`).
// Replaces rbc_sim3D_api.jl (initialize_simulation :17, step_simulation :77, get_state :106, get_info :126, get_nusselt :134)
// for a batch of envs; see rbc3d_kernels.hpp for the launch sequence.  Like the kernels it exists once per precision:
// host3:: drives rbc3:: (float64, the reference's arithmetic), host3f:: drives rbc3f:: (rbc_config.precision = float32); a handle
// belongs to one of them for its lifetime (rbc_handle::s3_f32) and rbc_api.hip dispatches with RBC_S3.
#pragma once
#include <type_traits>

#include "rbc3d_kernels.hpp"

#define RBC3_HOST host3
#define RBC3_NS rbc3
#include "rbc3d_host_body.hpp"
#undef RBC3_HOST
#undef RBC3_NS
#define RBC3_HOST host3f
#define RBC3_NS rbc3f
#include "rbc3d_host_body.hpp"
#undef RBC3_HOST
#undef RBC3_NS

// call `fn(args)` of the handle's precision
#define RBC_S3(h, fn, ...) ((h)->s3_f32 ? host3f::fn(__VA_ARGS__) : host3::fn(__VA_ARGS__))
