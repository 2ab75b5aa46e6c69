// Shared device/host helpers for the gfx950 kernels of libdadd_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dadd_hip.h"

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef float f4 __attribute__((ext_vector_type(4)));

#define DADD_WAVE 64

// ---- host-side error plumbing ---------------------------------------------------------------
void dadd_set_error(const char* fmt, ...);
#define DADD_REQUIRE(cond, ...)     \
  do {                              \
    if (!(cond)) {                  \
      dadd_set_error(__VA_ARGS__);  \
      return DADD_EINVAL;           \
    }                               \
  } while (0)
#define DADD_HIP(call)                                                                \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      dadd_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                     __LINE__);                                                       \
      return DADD_EHIP;                                                               \
    }                                                                                 \
  } while (0)
#define DADD_LAUNCH_CHECK() DADD_HIP(hipGetLastError())

static inline bool dadd_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// profiling hooks (api.hip)
bool dadd_prof_active(int kind);
void dadd_prof_pre(hipStream_t s);
void dadd_prof_post(hipStream_t s, double flop);

// ---- device helpers -------------------------------------------------------------------------
__device__ __forceinline__ float dadd_silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float dadd_gelu(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
