// Kernel argument block shared by the two implicit-GEMM kernels (igemm.hip, igemm_dma.hip).
#pragma once
#include "dadd_common.h"

struct IgemmArgs {
  const half_t* x;
  const half_t* x2;
  const half_t* w;
  half_t* out;
  float* partial;
  const float* bias;
  const float* rowvec;
  const half_t* residual;
  int* counters;   // split-K tickets, one per output tile, zero between launches (nullptr: finish kernel)
  int B, Hi, Wi, C1, C2, Ho, Wo, N;
  int taps, stride, ups, pad;
  int ldo, ldr, ld_rowvec;
  int splitk, flags;
  int M, K, nkt, kps, ntiles;
};

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (id % 8 shares an XCD, each
// with a private 4 MiB L2).  Remap so that one XCD owns a CONTIGUOUS range of logical tiles: row
// tiles that share halo rows and all column tiles of one row tile then hit the same L2, and the nine
// taps of a 3x3 window re-read the same ~1/8 of the activation from L2 instead of from the Infinity
// Cache.  Bijective for any grid size (cdna_hip_programming.md §5 "XCD swizzle must be bijective");
// placement only affects speed, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

int dadd_init_igemm_dma();
int dadd_launch_igemm_dma(const IgemmArgs& a, int tile_n, int nsplit, hipStream_t s);
