// Shared device/host helpers for the gfx950 kernels of libdadd_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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

// ---- launch wrapper with per-kernel timing (api.hip) ------------------------------------------------------
// Every kernel of the library is launched through dadd_launch().  While dadd_prof_begin() is active (eager
// launches only, never during stream capture) the launch goes through hipExtLaunchKernelGGL with a start/stop
// event pair: the pair carries the dispatch packet's OWN begin/end timestamps — the same clock rocprofv3's
// kernel trace reads — so the per-launch time excludes queue gaps and event-record overhead.  `name` is the
// rocprofv3 kernel name (without namespace / argument list), `flop` / `bytes` the launch's ALGORITHMIC work
// (2*M*N*K; compulsory reads + writes) for the roofline columns.
struct DaddLaunchTag {
  const char* name;
  double flop;
  double bytes;
};
extern int g_dadd_prof_on;
bool dadd_prof_slot(const DaddLaunchTag& tag, hipEvent_t* e0, hipEvent_t* e1);
template <typename... KA, typename... A>
inline void dadd_launch(const DaddLaunchTag& tag, void (*kernel)(KA...), dim3 grid, dim3 block, unsigned smem,
                        hipStream_t s, A... args) {
  hipEvent_t e0, e1;
  if (g_dadd_prof_on && dadd_prof_slot(tag, &e0, &e1))
    hipExtLaunchKernelGGL(kernel, grid, block, smem, s, e0, e1, 0, static_cast<KA>(args)...);
  else
    hipLaunchKernelGGL(kernel, grid, block, smem, s, static_cast<KA>(args)...);
}

// ---- device helpers -------------------------------------------------------------------------
// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division (ten instructions): GroupNorm + SiLU
// runs per element in gn_apply and, inside the 3x3 conv, on the loader waves of conv3x3_halo_kernel
__device__ __forceinline__ float dadd_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// Exact-form GELU 0.5 x (1 + erf(x/sqrt2)) with erf from Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, three orders below the fp16 rounding of the result): one v_rcp, one v_exp and
// a 5-term Horner instead of libm's branchy erff — the GEGLU epilogue evaluates 32 of these per
// thread behind a 5-tile main loop, where erff cost more cycles than the MFMAs.
__device__ __forceinline__ float dadd_gelu(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  const float erf_abs = fmaf(-p * t, e, 1.0f);           // erf(|x|/sqrt2)
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}
// Two at a time on packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: full rate on two lanes' worth of data per
// instruction) — same formula and rounding order per element as dadd_gelu, transcendentals stay scalar.
typedef float dadd_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dadd_f2 dadd_gelu2(dadd_f2 x) {
  const dadd_f2 one = {1.0f, 1.0f};
  const dadd_f2 z = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
  const dadd_f2 d = __builtin_elementwise_fma(dadd_f2{0.3275911f, 0.3275911f}, z, one);
  const dadd_f2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  dadd_f2 p = __builtin_elementwise_fma(dadd_f2{1.061405429f, 1.061405429f}, t, dadd_f2{-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(p, t, dadd_f2{1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(p, t, dadd_f2{-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(p, t, dadd_f2{0.254829592f, 0.254829592f});
  const dadd_f2 a = -z * z * 1.4426950408889634f;
  const dadd_f2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  const dadd_f2 erf_abs = __builtin_elementwise_fma(-p * t, e, one);
  return 0.5f * x * (one + __builtin_elementwise_copysign(erf_abs, x));
}
// Exchange with the lane 16 (32) away WITHOUT the LDS crossbar: __shfl_xor(v, 16 / 32, 64) compiles to ds_bpermute_b32
// plus an s_waitcnt lgkmcnt(0) — ~120 cycles of exposed latency per call in the softmax reductions, eight per 64-key
// tile in flash_kernel — while gfx950's v_permlane16_swap / v_permlane32_swap are plain VALU instructions.
// swap(x, x) leaves {r0, r0, r2, r2} / {r1, r1, r3, r3} (16-lane rows; resp. the two 32-lane halves) in its two results:
// both partners of every pair hold both values.
__device__ __forceinline__ void dadd_pair16(float v, float& a, float& b) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];   // (a bit_cast of the vector element itself reads element 0 for both)
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void dadd_pair32(float v, float& a, float& b) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];   // (a bit_cast of the vector element itself reads element 0 for both)
  a = __builtin_bit_cast(float, r0);
  b = __builtin_bit_cast(float, r1);
}
// sum / max over the four lanes {l, l^16, l^32, l^48} (every lane gets the result; the order of the additions is the
// same in all four lanes: (r0 + r1) + (r2 + r3))
__device__ __forceinline__ float dadd_sum_x16x32(float v) {
  float a, b;
  dadd_pair16(v, a, b);
  v = a + b;
  dadd_pair32(v, a, b);
  return a + b;
}
__device__ __forceinline__ float dadd_max_x16x32(float v) {
  float a, b;
  dadd_pair16(v, a, b);
  v = fmaxf(a, b);
  dadd_pair32(v, a, b);
  return fmaxf(a, b);
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
