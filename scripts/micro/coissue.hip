// Micro-benchmark: do vector-ALU instructions of one wave and MFMAs of ANOTHER wave on the same SIMD overlap?
// 256 workgroups x 8 waves (two per SIMD: wave w and w+4 share SIMD w%4).  Waves 0-3 run NM independent
// v_mfma_f32_16x16x32_f16, waves 4-7 run NV independent v_fma_f32 / v_exp_f32 / v_max3_f32; each role alone, then both.
// If the together time is max(alone) the pipes overlap across waves; if it is the sum they do not.  Also the same with
// BOTH waves of a SIMD running the same role (port contention).
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// mode bit 0: waves 0-3 do MFMA; bit 1: waves 4-7 do VALU; bit 2: waves 4-7 do MFMA too; bit 3: waves 0-3 do VALU too
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, unsigned long long* ticks) {
  const int wave = threadIdx.x >> 6;
  const bool lo = wave < 4;
  const bool do_mfma = lo ? (mode & 1) : (mode & 4);
  const bool do_valu = lo ? (mode & 8) : (mode & 2);
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * threadIdx.x + i); b[i] = (_Float16)(0.002f * i); }
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * threadIdx.x + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    }
  }
  if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == 0) v[i] = fmaf(v[i], 1.0001f, 0.5f);
        else if (KIND == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);
        else asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 2) & 15]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, unsigned long long* ticks) {
  const int iters = 2000;
  const int modes[] = {1, 2, 3, 5, 10, 15};
  const char* mn[] = {"MFMA on one wave/SIMD", "VALU on one wave/SIMD", "MFMA wave + VALU wave", "MFMA on both waves", "VALU on both waves",
                      "both roles on both waves"};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int m = 0; m < 6; ++m) {
    k<KIND><<<256, 512>>>(out, iters, modes[m], ticks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<256, 512>>>(out, iters, modes[m], ticks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    // per instruction: MFMA count = iters*8, VALU count = iters*16 (per wave)
    printf("%-10s %-28s %8.1f us   wave0 %8llu ticks (%.2f /MFMA)  wave4 %8llu ticks (%.2f /VALU, %.2f /MFMA)\n", name, mn[m], ms * 1e3, h[0],
           (double)h[0] / (iters * 8), h[4], (double)h[4] / (iters * 16), (double)h[4] / (iters * 8));
  }
}

int main() {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 256 * 8 * 8);
  run<0>("v_fma", out, ticks);
  run<1>("v_exp", out, ticks);
  run<2>("v_max3", out, ticks);
  return 0;
}
