// Folded LayerNorm with the row statistics supplied by the producer of x (DADD_EPI_LNFOLD + ln_stats_in): staging of
// everything the epilogue needs — the row partials of the tile's 128 rows, c1 and the composed bias of its columns —
// in LDS, by LDS-DMA issued at the START of the tile's K loop.  The epilogue then pays LDS latency only; read straight
// from global memory the same operands cost one exposed round trip for the partials plus one per column block
// (measured: 4.8 us per 128x160 tile on the persistent ring, more than the LayerNorm launch the fold removes).
// No registers are held across the K loop.  LDS-DMA ring kernel: the loader waves issue the (at most 4) extra DMA
// instructions each inside their counted-wait protocol (igemm_dma.hip) — the MFMA waves' loop is untouched (a
// vmcnt(0) there would also wait for the previous tile's output stores).  Register-staged kernel: every wave issues
// its share before the K loop and waits before the epilogue.
#pragma once
#include "igemm_args.h"

// LDS layout behind a kernel's K-tile buffers: c1 and the composed bias of the tile's columns (1 KB each: an LDS-DMA
// instruction writes 64 lanes x 16 B, zeros for the lanes past the tile), then [part][128 rows] float2 (1 KB per part).
// CAP = bytes the kernel instantiation can spare.  c1 / bias are staged whenever the path is on (they let the epilogue
// keep the row-outer store order), the partials when they fit too — otherwise the epilogue fetches them from global
// memory in one batch (the register-staged 128x160 tile keeps two workgroups per CU with 8 KB: 6 parts).
constexpr int LN_LDS_BYTES = 10240;
constexpr int LN_LDS_STATS = 2048;

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) void* ln_lptr_t;

__device__ __forceinline__ bool ln_lds_usable(const IgemmArgs& p, int bm, int nk) {
  return (p.flags & DADD_EPI_LNFOLD) && p.ln_stats_in != nullptr && bm == 128 && nk >= 3;
}
__device__ __forceinline__ bool ln_lds_stats(const IgemmArgs& p, int cap) { return LN_LDS_STATS + p.ln_parts_in * 1024 <= cap; }
// DMA instructions wave `w` (0..3) issues per tile
__device__ __forceinline__ int ln_lds_count(const IgemmArgs& p, int w, bool stats) {
  return (w == 0 ? 1 : 0) + ((w == 1 && (p.flags & DADD_EPI_BIAS)) ? 1 : 0) + (stats ? (p.ln_parts_in - w + 3) >> 2 : 0);
}

// `w`: wave index 0..3 (wave-uniform), BN: tile columns.  Rows past M read the next part's rows or zeros (past the
// end of the buffer); they are never stored.
template <int BN>
__device__ __forceinline__ void ln_lds_issue(const IgemmArgs& p, char* scr, int m0, int n0, int w, int lane, bool stats) {
  const unsigned cv = lane < BN / 4 ? (unsigned)((n0 + lane * 4) * 4) : 0x80000000u;
  if (w == 0) {
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)p.ln_c1, 0, p.N * 4, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsC, (ln_lptr_t)scr, 16, cv, 0, 0, 0);
  }
  if (w == 1 && (p.flags & DADD_EPI_BIAS)) {
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, p.N * 4, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (ln_lptr_t)(scr + 1024), 16, cv, 0, 0, 0);
  }
  if (!stats) return;
  // (32-bit scalar arithmetic only: a 64-bit product runs on the VALU, the descriptor then lives in VGPRs and every DMA
  // instruction sits in a waterfall loop — tests/test_isa_guards.py; the host checks parts * M * 8 < 2^31)
  const __amdgpu_buffer_rsrc_t rsS =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.ln_stats_in, 0, p.ln_parts_in * p.M * 8, 0x00020000);
  for (int pp = w; pp < p.ln_parts_in; pp += 4)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (ln_lptr_t)(scr + LN_LDS_STATS + pp * 1024), 16,
                                             (unsigned)(lane * 16), (pp * p.M + m0) * 8, 0, 0);
}
#endif
