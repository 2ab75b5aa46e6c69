// 3x3 / stride-1 / pad-1 convolution with the activation HALO resident in LDS.
//
// Why.  The LDS-DMA implicit GEMM (igemm_dma.hip) re-loads the 128-pixel activation tile once per tap:
// 16 KB of activations + 20 KB of weights per 64-deep K tile.  Its DMA stream is bound by the CU's
// vector-memory path (~32 B/clk/CU measured, profiles/r01_x_dma_limits.txt), i.e. ~1120 cycles per K tile
// against 640 cycles of MFMA.  The nine taps of one 64-channel chunk read shifted copies of the same
// (R+2) x (W+2) pixel halo (R = 128 / W rows of the output tile), so this kernel loads the halo ONCE per
// chunk (33.8 KB at W = 64, i.e. 3.8 KB per tap instead of 16 KB) and forms the nine A operands by
// shifted fragment reads: 23.8 KB instead of 36 KB of DMA per K tile.
//
// Structure (same tile, MFMA, epilogue and wave specialisation as igemm_dma.hip):
//   * 128 output pixels (whole image rows) x 160 channels per workgroup, 8 waves: waves 0-3 MFMA,
//     waves 4-7 DMA; K is walked chunk-major, taps fastest: k = tap * Cin + chunk * 64 + [0, 64);
//   * LDS: weight ring 4 x 20 KB | halo buffer x 2 (34 KB each) | 4 KB dump for dead DMA slots;
//     halo pixel p lives at p * 128 B with its eight 16-B chunks XOR-swizzled by p & 7: a ds_read_b128
//     fragment read of 16 consecutive pixels starting at ANY pixel (any tap shift) is bank-conflict free
//     (the (p >> 1) & 7 swizzle of the aligned GEMM tiles is not: 23 % conflict cycles measured);
//   * per K tile (one tap) a loader wave issues ONE halo piece of the NEXT chunk (8 pixels x 128 B,
//     nine slots per chunk; out-of-image pixels use an out-of-range offset = hardware zero fill) and
//     then its five weight pieces; the counted wait `vmcnt(5)` therefore covers every halo piece and
//     the weight tile of the next tap; one raw s_barrier per tap;
//   * fragments of the next tap are prefetched under the MFMAs, except across a chunk boundary, where
//     the last halo piece of the next chunk is still in flight (one exposed LDS latency per 9 taps).
// Split-K slices are ranges of chunks (fp32 slabs + splitk_finish_kernel, as for the implicit GEMM).
#include <cstdlib>

#include "dadd_common.h"
#include "igemm_args.h"
#include "igemm_epilogue.h"

// Diagnostic build only (-DDADD_IGEMM_EXP=3, scripts/exp_stamps.sh): s_memtime stamps around the waits of
// one MFMA wave and one loader wave per workgroup, summed into p.partial[workgroup][8] (as uint64 counts).
#ifndef DADD_IGEMM_EXP
#define DADD_IGEMM_EXP 0
#endif
#if DADD_IGEMM_EXP == 3
#define DADD_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define DADD_ACC(dst, a, b) dst += (b) - (a)
#else
#define DADD_STAMP(var)
#define DADD_ACC(dst, a, b)
#endif

namespace {

constexpr int BM = 128, BK = 64, BN = 160;
constexpr int WN = BN / 2, J = WN / 16, NBJ = BN / 32;
constexpr int B_BYTES = BN * BK * 2;               // one weight tile
constexpr int W_RING = 4 * B_BYTES;
constexpr int HALO_PIECES = 36;                    // 9 slots x 4 loader waves, 8 pixels each
constexpr int HALO_MAX_PIX = 272;                  // >= (R+2)*(W+2) for W in {16, 32, 64}
constexpr int HALO_BYTES = HALO_MAX_PIX * 128;
constexpr int DUMP_OFF = W_RING + 2 * HALO_BYTES;
constexpr int SMEM_BYTES = DUMP_OFF + 4 * 1024;

typedef __attribute__((address_space(3))) void* lptr_t;
constexpr unsigned OOB = 0x80000000u;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// REGST: after the prologue the loader waves stage through REGISTERS — buffer_load_dwordx4 into VGPRs two
// taps ahead, ds_write_b128 into the ring / halo slot when it frees — instead of LDS-DMA.  In-kernel stamps
// (profiles/r01_zi_stamps_conv_halo.txt) show that an LDS-DMA instruction costs its wave ~115 cycles at issue and
// that this time is taken from the MFMA wave on the same SIMD (per tap: 640 cycles of MFMA + ~690 of DMA issue,
// no waiting at the barrier on either side); plain loads + LDS writes issue in a fraction of that, the loaders'
// idle VGPRs hide the memory latency, and the compiler's own counted waits order load -> store.
template <bool REGST>
__global__ __launch_bounds__(512, 1) void conv3x3_halo_kernel(const IgemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wave_all >= 4;
  const int wave = wave_all & 3;
  const int wm = wave >> 1, wn = wave & 1;

  const int tile_id = xcd_remap(blockIdx.x, gridDim.x);
  int mt, nt;
  tile_decode(p, tile_id, mt, nt);
  const int m0 = mt * BM, n0 = nt * BN;
  const int z = blockIdx.y;
  const int Cin = p.C1 + p.C2;
  const int nchunk = Cin / BK;
  const int c0 = z * p.kps;                         // p.kps: chunks per K slice for this kernel
  const int c1 = min(nchunk, c0 + p.kps);
  const int n_it = (c1 - c0) * 9;

  const int W = p.Wo, H = p.Ho, WH = W + 2;
  const int R = BM / W;                             // output rows per tile
  const int HP = (R + 2) * WH;                      // halo pixels
  const int tpi = (H * W) / BM;                     // tiles per image
  const int b = mt / tpi;
  const int y0 = (mt - b * tpi) * R;

  if (loader) {
    // ---- per-lane DMA state
    const int lrow = lane >> 3, lch = lane & 7;
    unsigned hv1[9], hv2[9];                        // halo piece `slot`: byte offset of this lane's 16 B
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const int pi = s * 4 + wave;
      const int px = pi * 8 + lrow;
      const int hy = px / WH, hx = px - hy * WH;
      const int y = y0 - 1 + hy, x = hx - 1;
      const bool ok = px < HP && y >= 0 && y < H && x >= 0 && x < W;
      const int chunk = lch ^ (px & 7);
      const int pix = (b * H + y) * W + x;
      hv1[s] = ok ? (unsigned)((pix * p.C1 + chunk * 8) * 2) : OOB;
      hv2[s] = ok ? (unsigned)((pix * p.C2 + chunk * 8) * 2) : OOB;
    }
    const int live_pieces = (HP + 7) >> 3;
    unsigned w_v[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      const int row = (j * 4 + wave) * 8 + lrow;
      const int n = n0 + row;
      w_v[j] = (n < p.N) ? (unsigned)(((size_t)n * p.K + (lch ^ ((row >> 1) & 7)) * 8) * 2) : OOB;
    }
    const size_t pix_total = (size_t)p.B * H * W;
    const int rec1 = (int)(pix_total * p.C1 * 2), rec2 = (int)(pix_total * p.C2 * 2);
    const int recW = (int)((size_t)p.N * p.K * 2);
    const half_t* base2 = p.x2 ? p.x2 : p.x;

    // halo piece `slot` of chunk c -> buffer (c - c0) & 1; dead slots go to the dump area
    auto issue_halo = [&](int c, int slot) {
      const bool have = c < c1;
      const bool second = (c * BK) >= p.C1;
      const int cb = second ? c * BK - p.C1 : c * BK;
      const bool live = have && (slot * 4 + wave) < live_pieces;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(second ? base2 : p.x), 0, have ? (second ? rec2 : rec1) : 0, 0x00020000);
      const int dst = live ? W_RING + ((c - c0) & 1) * HALO_BYTES + (slot * 4 + wave) * 1024
                           : DUMP_OFF + wave * 1024;
      const unsigned vo = second ? hv2[slot] : hv1[slot];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + dst), 16, live ? vo : OOB,
                                               (unsigned)(cb * 2), 0, 0);
    };
    // weight tile of local iteration gi (chunk c0 + gi / 9, tap gi % 9) -> ring slot gi & 3
    int wk_tap = 0, wk_c = c0 * BK, wk_gi = 0;      // cursor of the NEXT weight tile to issue
    auto issue_w = [&]() {
      const bool live = wk_gi < n_it;
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, live ? recW : 0, 0x00020000);
      char* dst = smem + (wk_gi & 3) * B_BYTES + wave * 1024;
      const unsigned koff = (unsigned)((wk_tap * Cin + wk_c) * 2);
#pragma unroll
      for (int j = 0; j < NBJ; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(dst + j * 4096), 16, w_v[j], koff, 0, 0);
      ++wk_gi;
      const int t1 = wk_tap + 1;
      const int wrap = t1 == 9 ? 1 : 0;
      wk_tap = wrap ? 0 : t1;
      wk_c += wrap ? BK : 0;
    };

    // prologue: the whole halo of the first chunk, then weight tiles 0, 1, 2
#pragma unroll
    for (int s = 0; s < 9; ++s) issue_halo(c0, s);
    issue_w();
    issue_w();
    issue_w();
    if constexpr (REGST) wait_vmcnt<0>();
    else wait_vmcnt<NBJ>();                         // everything but weight tile 2
    __builtin_amdgcn_s_barrier();
    int cur_c = c0, cur_t = 0;
    [[maybe_unused]] unsigned long long st_wait = 0, st_bar = 0, st_issue = 0;
    DADD_STAMP(l_begin);
    if constexpr (REGST) {
      // ---- register-staged stream.  Group g = {halo piece (chunk(g) + 1, slot t(g)), weight tile g + 3}; it is
      // LOADED during iteration g - 2 and STORED to LDS during iteration g (after that iteration's barrier, when
      // the slot of tile g - 1 is free).  Three register groups, the loop is unrolled by 3 (n_it = 9 x chunks).
      typedef unsigned u4v __attribute__((ext_vector_type(4)));
      struct Grp { u4v h; u4v w[NBJ]; };
      int ld_c = c0, ld_t = 0;                      // load-side cursor (group to load next)
      // the halo source offset is computed on the fly from a running (row, column) of the lane's pixel:
      // indexing the nine precomputed offsets with a runtime slot puts the array in scratch, and the
      // scratch load's vmcnt wait drains the whole prefetch queue
      const int px0 = wave * 8 + lrow;              // this lane's pixel in slot 0; slot s adds 32 pixels
      const int hy0 = px0 / WH, hx0 = px0 - hy0 * WH;
      const int sdy = 32 / WH, sdx = 32 - sdy * WH;
      const int hchunk = (lch ^ lrow) * 8;          // (px & 7) == lrow for every slot
      int l_px = px0, l_hy = hy0, l_hx = hx0;
      auto load_halo = [&](int c) -> u4v {
        const bool have = c < c1;
        const bool second = (c * BK) >= p.C1;
        const int cb = second ? c * BK - p.C1 : c * BK;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(second ? base2 : p.x), 0, have ? (second ? rec2 : rec1) : 0, 0x00020000);
        const int y = y0 - 1 + l_hy, x = l_hx - 1;
        const bool ok = l_px < HP && y >= 0 && y < H && x >= 0 && x < W;
        const int pix = (b * H + y) * W + x;
        const unsigned vo = ok ? (unsigned)((pix * (second ? p.C2 : p.C1) + hchunk) * 2) : OOB;
        u4v v = {0u, 0u, 0u, 0u};
        if constexpr (DADD_IGEMM_EXP != 5 && DADD_IGEMM_EXP != 6)   // diagnostic builds 5/6: no global loads
          v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (unsigned)(cb * 2), 0);
        const bool last = ld_t == 8;                // next slot (selects only)
        const int nx = l_hx + sdx, carry = nx >= WH ? 1 : 0;
        l_px = last ? px0 : l_px + 32;
        l_hx = last ? hx0 : nx - (carry ? WH : 0);
        l_hy = last ? hy0 : l_hy + sdy + carry;
        return v;
      };
      auto load_group = [&](Grp& g) {
        g.h = load_halo(ld_c + 1);
        const bool live = wk_gi < n_it;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, live ? recW : 0, 0x00020000);
        const unsigned koff = (unsigned)((wk_tap * Cin + wk_c) * 2);
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
          if constexpr (DADD_IGEMM_EXP != 5 && DADD_IGEMM_EXP != 6)
            g.w[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, w_v[j], koff, 0);
          else
            g.w[j] = u4v{koff, w_v[j], 0u, 0u};
        }
        ++wk_gi;
        const int w1 = wk_tap + 1, ww = w1 == 9 ? 1 : 0;
        wk_tap = ww ? 0 : w1;
        wk_c += ww ? BK : 0;
        const int l1 = ld_t + 1, lw = l1 == 9 ? 1 : 0;
        ld_t = lw ? 0 : l1;
        ld_c += lw;
      };
      int st_wgi = 3;                               // weight tile the next stored group carries
      auto store_group = [&](const Grp& g) {        // store-side cursor = (cur_c, cur_t)
        const int pi = cur_t * 4 + wave;
        const bool live = (cur_c + 1) < c1 && pi < live_pieces;
        const int lmask = live ? -1 : 0;            // mask arithmetic: a branch here splits the step
        const int hdst = ((W_RING + ((cur_c + 1 - c0) & 1) * HALO_BYTES + pi * 1024) & lmask) |
                         ((DUMP_OFF + wave * 1024) & ~lmask);
        char* wdst = smem + (st_wgi & 3) * B_BYTES + wave * 1024 + lane * 16;
        if constexpr (DADD_IGEMM_EXP != 4 && DADD_IGEMM_EXP != 6) {   // diagnostic builds 4/6: no LDS writes
          *reinterpret_cast<u4v*>(smem + hdst + lane * 16) = g.h;
#pragma unroll
          for (int j = 0; j < NBJ; ++j) *reinterpret_cast<u4v*>(wdst + j * 4096) = g.w[j];
        } else {                                    // keep the loads alive
          unsigned acc = g.h[0];
#pragma unroll
          for (int j = 0; j < NBJ; ++j) acc ^= g.w[j][0] ^ g.w[j][3];
          if (acc == 0x12345678u) *reinterpret_cast<unsigned*>(smem + hdst) = acc;
        }
        ++st_wgi;
        const int t1 = cur_t + 1, wrap = t1 == 9 ? 1 : 0;
        cur_t = wrap ? 0 : t1;
        cur_c += wrap;
      };
      Grp ga, gb, gc;
      load_group(ga);                               // groups 0 and 1 in flight before the loop
      load_group(gb);
      auto step = [&](const Grp& st, Grp& ld) {     // branch-free: the compiler's counted vmcnt stays exact
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // last iteration's LDS writes are done
        __builtin_amdgcn_s_barrier();
        store_group(st);                            // the compiler's counted vmcnt waits are exact here (11..6)
        __builtin_amdgcn_sched_barrier(0);          // keep the new loads behind the stores (and the wait exact)
        load_group(ld);                             // group gi + 2
      };
      for (int gi = 0; gi < n_it; gi += 3) {
        step(ga, gc);
        step(gb, ga);
        step(gc, gb);
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      return;
    }
    for (int gi = 0; gi < n_it; ++gi) {
      DADD_STAMP(l0);
      wait_vmcnt<NBJ>();                            // all but the weight tile issued last iteration
      DADD_STAMP(l1);
      __builtin_amdgcn_s_barrier();
      DADD_STAMP(l2);
      // the halo piece FIRST: the next wait (all but the NBJ youngest) then covers it
      switch (cur_t) {                              // hv1/hv2 stay in registers: constant indices only
        case 0: issue_halo(cur_c + 1, 0); break;
        case 1: issue_halo(cur_c + 1, 1); break;
        case 2: issue_halo(cur_c + 1, 2); break;
        case 3: issue_halo(cur_c + 1, 3); break;
        case 4: issue_halo(cur_c + 1, 4); break;
        case 5: issue_halo(cur_c + 1, 5); break;
        case 6: issue_halo(cur_c + 1, 6); break;
        case 7: issue_halo(cur_c + 1, 7); break;
        default: issue_halo(cur_c + 1, 8); break;
      }
      issue_w();                                    // tile gi + 3 -> the slot of tile gi - 1
      const int t1 = cur_t + 1;
      const int wrap = t1 == 9 ? 1 : 0;
      cur_t = wrap ? 0 : t1;
      cur_c += wrap;
      DADD_STAMP(l3);
      DADD_ACC(st_wait, l0, l1);
      DADD_ACC(st_bar, l1, l2);
      DADD_ACC(st_issue, l2, l3);
    }
    wait_vmcnt<0>();
#if DADD_IGEMM_EXP == 3
    if (wave == 0 && lane == 0 && p.partial && blockIdx.y == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.partial) + (size_t)blockIdx.x * 8;
      o[4] = st_wait; o[5] = st_bar; o[6] = st_issue; o[7] = __builtin_amdgcn_s_memtime() - l_begin;
    }
#endif
    return;
  }

  // ---- compute waves
  f4 acc[J][4];
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f4{0.f, 0.f, 0.f, 0.f};
  const int fq = lane >> 4;
  int p0[4];                                        // halo pixel of tap (0, 0) for fragment i
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = wm * 64 + i * 16 + (lane & 15);
    const int r = q / W;
    p0[i] = r * WH + (q - r * W);
  }
  const int swb = (wn * WN + (lane & 15));
  const int fb0 = (swb * 8 + (fq ^ ((swb >> 1) & 7))) * 16;         // weight fragment, K half 0
  const int fb1 = fb0 ^ 64;                                         // K half 1: chunk index ^ 4
  // (rows j*16 further down keep the swizzle term: (row + 16 j) >> 1 & 7 == (row >> 1) & 7)
  auto a_addr = [&](int i, int dt) {                // byte offset inside a halo buffer, K half 0
    const int px = p0[i] + dt;
    return px * 128 + ((fq ^ (px & 7)) << 4);
  };

  __builtin_amdgcn_s_barrier();                     // halo of the first chunk and weight tiles 0, 1 landed
  __builtin_amdgcn_sched_barrier(0);
  h8 xa0[4], xa1[4], wb0[J], wb1[J];
  int aoff[4];                                      // addresses of the CURRENT tap (half 0)
  {
    const char* hb = smem + W_RING;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      aoff[i] = a_addr(i, 0);
      xa0[i] = *reinterpret_cast<const h8*>(hb + aoff[i]);
    }
#pragma unroll
    for (int j = 0; j < J; ++j) wb0[j] = *reinterpret_cast<const h8*>(smem + fb0 + j * 2048);
  }
  int cur_t = 0, cur_ky = 0, cur_kx = 0, hsel = 0;  // tap of this iteration, halo buffer of this chunk
  [[maybe_unused]] unsigned long long sc_bar = 0, sc_h0 = 0, sc_h1 = 0;
  DADD_STAMP(c_begin);
  for (int gi = 0; gi < n_it; ++gi) {
    DADD_STAMP(c0s);
    __builtin_amdgcn_s_barrier();                   // one barrier per tap on both sides (also for gi == 0)
    DADD_STAMP(c1s);
    __builtin_amdgcn_sched_barrier(0);
    const char* hb = smem + W_RING + hsel * HALO_BYTES;
    const char* wcur1 = smem + (gi & 3) * B_BYTES + fb1;
    const char* wnext = smem + ((gi + 1) & 3) * B_BYTES + fb0;
    if (cur_t == 0 && gi > 0) {                     // first tap of a new chunk: no prefetch was possible
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        aoff[i] = a_addr(i, 0);
        xa0[i] = *reinterpret_cast<const h8*>(hb + aoff[i]);
      }
    }
    // next tap
    const int kx1 = cur_kx + 1;
    const int w3 = kx1 == 3 ? 1 : 0;
    const int nkx = w3 ? 0 : kx1;
    const int nky = cur_ky + w3;                    // 3 == chunk boundary
    const bool same_chunk = nky < 3;
    const int ndt = nky * WH + nkx;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4 * J; ++k) {               // K half 0; the half-1 fragments stream in behind
      const int jj = k / 4, ii = k % 4;
      acc[jj][ii] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb0[jj], xa0[ii], acc[jj][ii], 0, 0, 0);
      if (k == 0) wb1[0] = *reinterpret_cast<const h8*>(wcur1);
      else if (k <= 4) xa1[k - 1] = *reinterpret_cast<const h8*>(hb + (aoff[k - 1] ^ 64));
      else if (k < 4 + J) wb1[k - 4] = *reinterpret_cast<const h8*>(wcur1 + (k - 4) * 2048);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (same_chunk) {
#pragma unroll
      for (int i = 0; i < 4; ++i) aoff[i] = a_addr(i, ndt);
    }
    DADD_STAMP(c2s);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4 * J; ++k) {               // K half 1; prefetch of the next tap's half 0
      const int jj = k / 4, ii = k % 4;
      acc[jj][ii] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb1[jj], xa1[ii], acc[jj][ii], 0, 0, 0);
      if (k == 0) wb0[0] = *reinterpret_cast<const h8*>(wnext);
      else if (k <= 4) {
        if (same_chunk) xa0[k - 1] = *reinterpret_cast<const h8*>(hb + aoff[k - 1]);
      } else if (k < 4 + J) wb0[k - 4] = *reinterpret_cast<const h8*>(wnext + (k - 4) * 2048);
      __builtin_amdgcn_sched_barrier(0);
    }
    cur_kx = nkx;
    cur_ky = same_chunk ? nky : 0;
    cur_t = same_chunk ? cur_t + 1 : 0;
    hsel = same_chunk ? hsel : hsel ^ 1;
    __builtin_amdgcn_sched_barrier(0);
    DADD_STAMP(c3s);
    DADD_ACC(sc_bar, c0s, c1s);
    DADD_ACC(sc_h0, c1s, c2s);
    DADD_ACC(sc_h1, c2s, c3s);
  }
#if DADD_IGEMM_EXP == 3
  if (wave == 0 && lane == 0 && p.partial && blockIdx.y == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.partial) + (size_t)blockIdx.x * 8;
    o[0] = sc_bar; o[1] = sc_h0; o[2] = sc_h1; o[3] = __builtin_amdgcn_s_memtime() - c_begin;
  }
#endif
  igemm_epilogue<J, 4, 64, WN>(p, acc, m0, n0, wm, wn, lane, z, smem);
#endif
}

}  // namespace

int dadd_init_conv_halo() {
  DADD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
  DADD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES));
  return DADD_OK;
}

// Shapes this kernel takes (everything else stays on the implicit GEMM).
bool dadd_conv_halo_applicable(const IgemmArgs& a, int tile_n) {
  static const bool off = getenv("DADD_NO_HALO") != nullptr;   // A/B measurements only
  return !off && a.taps == 9 && a.stride == 1 && a.pad == 1 && !a.ups && tile_n == BN &&
         a.Hi == a.Ho && a.Wi == a.Wo && (a.Wo == 16 || a.Wo == 32 || a.Wo == 64) &&
         (a.Ho * a.Wo) % BM == 0 && (BM / a.Wo + 2) * (a.Wo + 2) <= HALO_MAX_PIX &&
         !(a.flags & (DADD_EPI_GEGLU | DADD_TUNE_PERSIST));
}

// `a.kps` = chunks per K slice, `a.splitk` = nsplit (set by the caller).
int dadd_launch_conv_halo(const IgemmArgs& a, int nsplit, hipStream_t s) {
  DADD_REQUIRE((size_t)a.B * a.Hi * a.Wi * (size_t)(a.C1 > a.C2 ? a.C1 : a.C2) * 2 < 0x7FF00000ull &&
                   (size_t)a.N * a.K * 2 < 0x7FF00000ull,
               "conv_halo: operand larger than the 2 GiB buffer window");
  dim3 grid(a.mtiles * a.ntiles, nsplit);
  static const bool regst = getenv("DADD_HALO_REGST") ? atoi(getenv("DADD_HALO_REGST")) != 0 : true;   // A/B
  if (regst) hipLaunchKernelGGL(conv3x3_halo_kernel<true>, grid, dim3(512), SMEM_BYTES, s, a);
  else hipLaunchKernelGGL(conv3x3_halo_kernel<false>, grid, dim3(512), SMEM_BYTES, s, a);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}
