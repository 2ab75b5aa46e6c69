// Micro-benchmark: cycles (s_memtime ticks) per instruction for ONE wave per SIMD issuing 16 independent copies of one
// instruction back to back - the vector instructions of the attention softmax / GELU epilogues and the MFMA shapes.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/inst_cost.hip -o /tmp/inst_cost && /tmp/inst_cost
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* ticks) {
  float v[16], w[16];
  f2 p[16];
  for (int i = 0; i < 16; ++i) { v[i] = out[threadIdx.x + i] + 1.5f; w[i] = out[threadIdx.x + 16 + i] + 0.25f; p[i] = f2{v[i], w[i]}; }
  h8 a = *reinterpret_cast<const h8*>(out + threadIdx.x * 8), b = *reinterpret_cast<const h8*>(out + 4096 + threadIdx.x * 8);
  h4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
  f4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define X(i)                                                                                                            \
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "v"(w[(i + 1) & 15]));               \
  else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, 1.0, %1" : "+v"(v[i]) : "v"(w[i]));                               \
  else if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 15]), "v"(p[(i + 2) & 15])); \
  else if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));                      \
  else if (KIND == 4) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));                                                    \
  else if (KIND == 5) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));                                                    \
  else if (KIND == 6) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v[i]) : "v"(w[i]), "v"(w[(i + 1) & 15]));       \
  else if (KIND == 7) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(w[i]), "v"(w[(i + 1) & 15]));         \
  else if (KIND == 8) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));                                    \
  else if (KIND == 9) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));                                                    \
  else if (KIND == 10) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));                                   \
  else if (KIND == 11) asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(v[i]), "+v"(w[i]));                 \
  else if (KIND == 12) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);                          \
  else if (KIND == 13) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[i], 0, 0, 0);                         \
  else if (KIND == 14) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));                                   \
  else if (KIND == 15) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 15]));
    REP16(X)
#undef X
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += v[i] + w[i] + p[i][0] + p[i][1] + acc[i][0] + acc[i][3];
  out[8192 + blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* ticks) {
  const int iters = 1000;
  k<KIND><<<256, 256>>>(out, iters, ticks);
  hipDeviceSynchronize();
  k<KIND><<<256, 256>>>(out, iters, ticks);
  hipDeviceSynchronize();
  unsigned long long h;
  hipMemcpy(&h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-46s %7.2f ticks per instruction\n", name, (double)h / (iters * 16));
}

int main() {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, (8192 + 256 * 256) * 4); hipMemset(out, 0, (8192 + 256 * 256) * 4); hipMalloc(&ticks, 256 * 8);
  run<0>("v_fma_f32 (three VGPR operands)", out, ticks);
  run<1>("v_fma_f32 (two VGPR operands + constant)", out, ticks);
  run<2>("v_pk_fma_f32 (two fp32 per lane)", out, ticks);
  run<3>("v_pk_mul_f32", out, ticks);
  run<15>("v_pk_add_f32", out, ticks);
  run<14>("v_mul_f32", out, ticks);
  run<10>("v_sub_f32", out, ticks);
  run<8>("v_max_f32", out, ticks);
  run<7>("v_max3_f32", out, ticks);
  run<6>("v_cvt_pk_f16_f32", out, ticks);
  run<4>("v_exp_f32", out, ticks);
  run<5>("v_exp_f16", out, ticks);
  run<9>("v_rcp_f32", out, ticks);
  run<11>("v_permlane32_swap_b32 (+ s_nop 1)", out, ticks);
  run<12>("v_mfma_f32_16x16x32_f16", out, ticks);
  run<13>("v_mfma_f32_16x16x16_f16", out, ticks);
  return 0;
}
