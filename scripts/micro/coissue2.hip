// Follow-up of coissue.hip (VALU of another wave is starved while a wave issues MFMAs back to back): what DOES overlap?
//   A: ONE wave per SIMD interleaving 1 MFMA + NV independent v_fma_f32 in its own instruction stream (NV = 0..4)
//   B: MFMA wave + VALU wave per SIMD as before, the VALU wave at s_setprio 3
//   C: the same, the MFMA wave at s_setprio 0 and an s_nop 7 / s_sleep after every MFMA? (yield variants)
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 scripts/micro/coissue2.hip -o /tmp/coissue2 && /tmp/coissue2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(512) void k_interleave(float* out, int iters, unsigned long long* ticks) {
  h8 a = *reinterpret_cast<const h8*>(out + threadIdx.x * 8), b = *reinterpret_cast<const h8*>(out + 4096 + threadIdx.x * 8);
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = 0.001f * threadIdx.x + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) v[i * 4 + j] = fmaf(v[i * 4 + j], 1.0001f, 0.5f);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
      if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);  // NV VALU
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 32; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

// waves 0-3: MFMA (YIELD: 0 none, 1 = s_nop 7 after each MFMA, 2 = s_setprio 0 on this wave); waves 4-7: VALU at priority PRIO
template <int PRIO, int YIELD>
__global__ __launch_bounds__(512) void k_prio(float* out, int iters, unsigned long long* ticks) {
  const int wave = threadIdx.x >> 6;
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * threadIdx.x + i); b[i] = (_Float16)(0.002f * i); }
  f4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = 0.001f * threadIdx.x + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        if (YIELD == 1) {
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 7");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
    if (PRIO == 3) __builtin_amdgcn_s_setprio(3);
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

template <typename F>
void time_it(const char* name, F launch, unsigned long long* ticks, int iters, int nv) {
  launch(); hipDeviceSynchronize(); launch(); hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-44s wave0 %8llu ticks = %6.2f per MFMA", name, h[0], (double)h[0] / (iters * 8));
  if (nv < 0) printf("   wave4 %8llu ticks = %6.2f per v_fma", h[4], (double)h[4] / (iters * 16));
  printf("\n");
}

int main() {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, 256 * 512 * 4); hipMemset(out, 0, 256 * 512 * 4); hipMalloc(&ticks, 256 * 8 * 8);
  const int iters = 2000;
  time_it("A one wave/SIMD: MFMA only", [&] { k_interleave<0><<<256, 256>>>(out, iters, ticks); }, ticks, iters, 0);
  time_it("A one wave/SIMD: MFMA + 1 v_fma interleaved", [&] { k_interleave<1><<<256, 256>>>(out, iters, ticks); }, ticks, iters, 1);
  time_it("A one wave/SIMD: MFMA + 2 v_fma interleaved", [&] { k_interleave<2><<<256, 256>>>(out, iters, ticks); }, ticks, iters, 2);
  time_it("A one wave/SIMD: MFMA + 3 v_fma interleaved", [&] { k_interleave<3><<<256, 256>>>(out, iters, ticks); }, ticks, iters, 3);
  time_it("A one wave/SIMD: MFMA + 4 v_fma interleaved", [&] { k_interleave<4><<<256, 256>>>(out, iters, ticks); }, ticks, iters, 4);
  time_it("A two waves/SIMD: MFMA + 3 v_fma interleaved", [&] { k_interleave<3><<<256, 512>>>(out, iters, ticks); }, ticks, iters, 3);
  time_it("B MFMA wave + VALU wave, prio 0", [&] { k_prio<0, 0><<<256, 512>>>(out, iters, ticks); }, ticks, iters, -1);
  time_it("B MFMA wave + VALU wave, VALU at s_setprio 1", [&] { k_prio<1, 0><<<256, 512>>>(out, iters, ticks); }, ticks, iters, -1);
  time_it("B MFMA wave + VALU wave, VALU at s_setprio 3", [&] { k_prio<3, 0><<<256, 512>>>(out, iters, ticks); }, ticks, iters, -1);
  time_it("C MFMA wave with s_nop 7 after each + VALU", [&] { k_prio<0, 1><<<256, 512>>>(out, iters, ticks); }, ticks, iters, -1);
  time_it("C ... and VALU at s_setprio 3", [&] { k_prio<3, 1><<<256, 512>>>(out, iters, ticks); }, ticks, iters, -1);
  return 0;
}
