// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 from ONE wave per SIMD
// (20 independent accumulators, operands in registers, no memory): shader cycles per MFMA and TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k16(unsigned long long* out, float* sink, int iters, float seed) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed * (threadIdx.x % 13 + i)); b[i] = (_Float16)(seed * (threadIdx.x % 7 + 2 * i)); }
  f4 acc[20];
  for (int j = 0; j < 20; ++j) acc[j] = f4{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 20; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 20; ++j) s += acc[j][0] + acc[j][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k32(unsigned long long* out, float* sink, int iters, float seed) {
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed * (threadIdx.x % 13 + i)); b[i] = (_Float16)(seed * (threadIdx.x % 7 + 2 * i)); }
  f16v acc[5];
  for (int j = 0; j < 5; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 5; ++j) s += acc[j][0] + acc[j][15];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <typename K>
void run(const char* name, K kern, int waves, int mfma_per_iter, double flop_per_mfma) {
  const int blocks = 256, iters = 2000;
  unsigned long long* d; float* sink;
  hipMalloc(&d, blocks * 8); hipMalloc(&sink, blocks * 64 * waves * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), 0, 0, d, sink, iters, 0.01f);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0; for (int i = 0; i < blocks; ++i) cyc += (double)h[i]; cyc /= blocks;
  const double n = (double)iters * mfma_per_iter;
  printf("%-34s waves/CU=%d: %.2f cycles per MFMA per wave, kernel %.3f ms -> %.0f TFLOP/s, implied clock %.2f GHz\n", name, waves,
         cyc / n, ms, blocks * waves * n * flop_per_mfma / (ms * 1e-3) / 1e12, cyc / (ms * 1e-3) / 1e9);
}

int main() {
  run("v_mfma_f32_16x16x32_f16", k16<4>, 4, 20, 16384.0);
  run("v_mfma_f32_16x16x32_f16", k16<8>, 8, 20, 16384.0);
  run("v_mfma_f32_32x32x16_f16", k32<4>, 4, 10, 32768.0);
  run("v_mfma_f32_32x32x16_f16", k32<8>, 8, 10, 32768.0);
  return 0;
}
