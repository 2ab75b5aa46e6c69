// Implicit-GEMM convolution / linear with LDS-DMA staging (global_load_lds_dwordx4) — the fast path
// for K >= 4 tiles.  Same math, tile shape (128 x BN x 64), LDS image (XOR-swizzled 128-byte rows),
// swapped MFMA and epilogue as igemm.hip; what changes is how tiles reach LDS:
//
//   * no register staging, no ds_write: every wave issues 16-byte-per-lane DMA loads whose per-lane
//     SOURCE address carries the swizzle and the im2col gather (padding lanes read a zero line), the
//     LDS destination is the wave-linear 1 KiB the hardware requires (8 rows x 128 B);
//   * buffer-addressed DMA (buffer_load_dwordx4 ... offen lds): 32-bit per-lane offsets, the per-tile
//     tap / channel / K offset rides in the SCALAR soffset operand, and out-of-image taps are a
//     precomputed 9-bit mask per row that selects an out-of-range offset (the hardware returns zeros);
//   * a ring of 4 stages behind a COUNTED s_waitcnt vmcnt(N) and a raw s_barrier (a __syncthreads()
//     would drain the queue): one barrier per K tile; at iteration it, tile it computes, tile it+1 has
//     landed and its first fragments are prefetched under the MFMAs, tiles it+2 / it+3 are in flight;
//   * one block per CU (up to 144 KiB of LDS), latency hidden by the ring instead of by occupancy.
//
// The register-staged kernel in igemm.hip measured LDS-write-bound (ds_write_b128 moves ~79 B/clk/CU
// against 36.8 KB per K tile); LDS-DMA removes that traffic from the VGPR->LDS path entirely.
#include "dadd_common.h"
#include "igemm_args.h"
#include <cstdlib>
#include "igemm_epilogue.h"

namespace {

constexpr int BK = 64;

typedef __attribute__((address_space(3))) void* lptr_t;
constexpr unsigned OOB = 0x80000000u;   // beyond num_records of every descriptor: the load returns zeros

__device__ __forceinline__ int lds_off(int row, int chunk) {
  return (row * 8 + (chunk ^ ((row >> 1) & 7))) * 8;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Tile BM x BN x 64 with BM in {64, 128}, BN in {64, 128, 160}.  The 64-row tiles exist for the short GEMMs of the
// 16x16 / 8x8 levels (M = 1024 / 256): they fill the chip without split-K slabs, and two of their workgroups
// (64 KB of LDS each at 64x64) share a CU, so one's prologue / epilogue hides under the other's K loop.
// LNK: how a LayerNorm folded into this linear (igemm_args.h) gets its row statistics.  0: no fold (or, on 64-row tiles,
// the epilogue reads the producer's partials from global memory); 1: the MFMA waves accumulate them from the A
// fragments; 2 (128-row tiles): the producer's partials, c1 and the composed bias are staged in LDS by the loader
// waves (ln_lds.h).  Separate instantiations: the plain kernels carry none of the staging code (it cost the
// persistent qkv / GEGLU launches 3-4 us in scalar spills when it was a run-time branch).
template <int BM, int BN, bool UPS, bool PERS, int LNK>
__global__ __launch_bounds__(512, (BM + BN) <= 128 ? 4 : 2) void igemm_dma_kernel(const IgemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type exists only in device code; the host
                                      // pass needs just the launch stub of this signature
  constexpr bool WS = true;
  constexpr bool LNF = LNK == 1;
  constexpr int WM = BM / 2, MI = WM / 16;   // compute waves: 2 (M) x 2 (N); rows / 16-row fragments per wave
  constexpr int WN = BN / 2;
  constexpr int J = WN / 16;
  constexpr int NA = BM / 32;   // DMA instructions per loader wave per tile, activations (2 or 4)
  constexpr int NBJ = BN / 32;  // DMA instructions per wave per tile, weights (4 or 5)
  constexpr int LPT = NA + NBJ;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  // WS (wave specialisation): 8 waves, two per SIMD.  Waves 0-3 only read fragments and issue MFMAs,
  // waves 4-7 only issue the DMA stream.  Measured with the diagnostic builds (profiles/r01_u_dma_limits.txt):
  // in ONE instruction stream the two do not overlap — a DMA instruction holds its wave for ~60-180
  // cycles at issue and with a single wave per SIMD the matrix pipe drains meanwhile (16384x640x5760:
  // DMA alone 84 us, MFMA alone 91 us, together 154 us).
  const int t = threadIdx.x, lane = t & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = WS && wave_all >= 4;
  const int wave = WS ? (wave_all & 3) : wave_all;   // compute role: (wm, wn); loader role: row share
  const int wm = wave >> 1, wn = wave & 1;
  // Tiles of this workgroup.  One tile per workgroup, or (PERS, short-K GEMMs with more tiles than
  // CUs) a contiguous range of logical tiles per workgroup: the DMA ring then runs as ONE stream of K
  // tiles across output tiles, so the loads of tile t+1 are in flight under the MFMAs and the
  // epilogue of tile t instead of every tile paying its own pipeline fill.
  int tile_first, tile_count;
  if constexpr (PERS) {
    const int total = p.mtiles * p.ntiles, nwg = gridDim.x;
    const int l = xcd_remap(blockIdx.x, nwg);
    const int q = total / nwg, r = total - q * nwg;
    tile_first = l * q + min(l, r);
    tile_count = q + (l < r ? 1 : 0);
  } else {
    tile_first = xcd_remap(blockIdx.x, gridDim.x);
    tile_count = 1;
  }
  const int z = PERS ? 0 : blockIdx.y;
  const int kt0 = PERS ? 0 : z * p.kps;
  const int kt1 = PERS ? p.nkt : min(p.nkt, kt0 + p.kps);
  const int nk = kt1 - kt0;
  const int Cin = p.C1 + p.C2;
  const int Hv = UPS ? 2 * p.Hi : p.Hi;
  const int Wv = UPS ? 2 * p.Wi : p.Wi;
  const int HoWo = p.Ho * p.Wo;
  const int lrow = lane >> 3, lch = lane & 7;

  // Buffer descriptors.  For the affine gathers (everything but the 2x-upsample) the "-pad" of the
  // window origin is folded into the descriptor base, so per-lane offsets and the per-tap scalar
  // offset are both non-negative; lanes whose tap falls outside the image get voffset = OOB.
  const int shift1 = UPS ? 0 : (p.pad * p.Wi + p.pad) * p.C1;
  const int shift2 = UPS ? 0 : (p.pad * p.Wi + p.pad) * p.C2;
  const size_t pix_total = (size_t)p.B * p.Hi * p.Wi;
  const int rec1 = (int)((pix_total * p.C1 + shift1) * 2);
  const int rec2 = (int)((pix_total * p.C2 + shift2) * 2);
  const int recW = (int)((size_t)p.N * p.K * 2);
  const half_t* base1 = p.x - shift1;
  const half_t* base2 = (p.x2 ? p.x2 : p.x) - shift2;

  // per-lane gather state: byte offsets of the window origin in both sources, validity bit per tap
  unsigned a_v1[NA], a_v2[NA], a_mask[NA];
  int a_pix[NA], a_y[NA], a_x[NA], a_cc[NA];   // only the upsample path uses these per tile
  unsigned w_v[NBJ];
  auto setup_tile = [&](int tile_id) {
  int nt, mt;
  tile_decode(p, tile_id, mt, nt);
  const int m0 = mt * BM, n0 = nt * BN;
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (i * 4 + wave) * 8 + lrow;
    const int m = m0 + row;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int b = mm / HoWo;
    const int rem = mm - b * HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    const int cc = (lch ^ ((row >> 1) & 7)) * 8;   // logical chunk this lane fetches (halfs)
    const int pix = b * p.Hi * p.Wi;
    const int y0 = oy * p.stride - p.pad, x0 = ox * p.stride - p.pad;
    a_pix[i] = pix; a_y[i] = ok ? y0 : -100000; a_x[i] = x0; a_cc[i] = cc;
    const int org = pix + oy * p.stride * p.Wi + ox * p.stride;
    a_v1[i] = (unsigned)((org * p.C1 + cc) * 2);
    a_v2[i] = (unsigned)((org * p.C2 + cc) * 2);
    unsigned mask = 0;
    const int ntap = p.taps;
    for (int tp = 0; tp < ntap; ++tp) {
      const int ky = (ntap == 9) ? tp / 3 : 0, kx = (ntap == 9) ? tp - 3 * ky : 0;
      const int iy = y0 + ky, ix = x0 + kx;
      if (ok && iy >= 0 && iy < Hv && ix >= 0 && ix < Wv) mask |= 1u << tp;
    }
    a_mask[i] = mask;
  }
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int row = (j * 4 + wave) * 8 + lrow;
    const int n = n0 + row;
    w_v[j] = (n < p.N) ? (unsigned)(((size_t)n * p.K + (lch ^ ((row >> 1) & 7)) * 8) * 2) : OOB;
  }
  };
  setup_tile(tile_first);
  int iss_tile = tile_first, iss_left = tile_count;   // PERS: the tile the DMA cursor is in / tiles left

  // ---- wave-uniform tile cursor, advanced incrementally (no divisions in the loop): K tile -> tap,
  // (ky, kx), channel offset; plus the byte offsets of the ring slots being filled / computed.
  // K order.  korder 0: channels fastest (k = tap*Cin + c ascending, the order of igemm.hip).  korder 1
  // (3x3 only): TAP fastest — the nine taps of one 64-channel chunk are consecutive K tiles, so their
  // activation tiles are shifted copies of the same ~264 pixel lines (one 128-B line per pixel per
  // chunk) and can hit in the CU's L1 instead of each going to L2.  Same products, different
  // summation order.
  const bool tapfast = p.korder != 0;
  int cur_kt = kt0;
  // (the runtime division runs on the VALU: pin the wave-uniform results to SGPRs, or the scalar offset
  // operand of every weight DMA is legalised with a waterfall loop)
  int cur_tap = __builtin_amdgcn_readfirstlane(tapfast ? kt0 % 9 : (kt0 * BK) / Cin);
  int cur_c = __builtin_amdgcn_readfirstlane(tapfast ? (kt0 / 9) * BK : kt0 * BK - cur_tap * Cin);
  int cur_koff = __builtin_amdgcn_readfirstlane((cur_tap * Cin + cur_c) * 2);   // byte offset of the K tile in a weight row
  int cur_ky = (p.taps == 9) ? cur_tap / 3 : 0;
  int cur_kx = (p.taps == 9) ? cur_tap - 3 * cur_ky : 0;
  int fill_off = wave * 1024;                 // LDS byte offset of this wave's share of the slot to fill

  // One K tile of DMA work, split so that the pieces can be interleaved with MFMAs:
  //   IssueCtx c = issue_begin();  issue_a(c, i) x NA;  issue_w(c, j) x NBJ;  issue_advance();
  struct IssueCtx {
    __amdgpu_buffer_rsrc_t rsA, rsW;
    unsigned soff, koff;
    char* sa;
    int cs, cb, ky, kx, tap;
    bool second;
  };
  auto issue_begin = [&]() {
    IssueCtx c;
    const bool live = PERS ? iss_left > 0 : cur_kt < kt1;   // past the end: zero-record descriptors
    c.second = cur_c >= p.C1;
    c.cs = c.second ? p.C2 : p.C1;
    c.cb = c.second ? cur_c - p.C1 : cur_c;
    c.ky = cur_ky; c.kx = cur_kx; c.tap = cur_tap;
    c.rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(c.second ? base2 : base1), 0,
                                              live ? (c.second ? rec2 : rec1) : 0, 0x00020000);
    c.rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, live ? recW : 0, 0x00020000);
    c.sa = smem + fill_off;
    c.soff = UPS ? (unsigned)(c.cb * 2) : (unsigned)(((cur_ky * p.Wi + cur_kx) * c.cs + c.cb) * 2);
    c.koff = (unsigned)cur_koff;
    return c;
  };
  auto issue_w = [&](const IssueCtx& c, int j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rsW, (lptr_t)(c.sa + A_BYTES + j * 4096), 16, w_v[j], c.koff, 0, 0);
  };
  auto issue_advance = [&]() {
    // selects only: a branch here would split the loop body into two scheduling regions
    ++cur_kt;
    // selects only (a branch here makes the compiler treat the cursor — and with it the buffer
    // descriptors — as divergent: every DMA instruction then sits in a waterfall loop, 2.7x slower)
    const int c1 = cur_c + BK;
    const int cwrap = c1 >= Cin ? 1 : 0;
    const int kx1 = cur_kx + (tapfast ? 1 : cwrap);
    const int w3 = kx1 == 3 ? 1 : 0;
    cur_kx = w3 ? 0 : kx1;
    const int ky1 = cur_ky + w3;
    const int w9 = (tapfast && ky1 == 3) ? 1 : 0;
    cur_ky = w9 ? 0 : ky1;
    cur_koff += tapfast ? (w9 ? (BK - 8 * Cin) * 2 : Cin * 2) : BK * 2;   // own scalar chain: cur_tap lives in a VGPR
    cur_tap = tapfast ? (w9 ? 0 : cur_tap + 1) : cur_tap + cwrap;
    cur_c = tapfast ? cur_c + (w9 ? BK : 0) : (cwrap ? 0 : c1);
    const int f1 = fill_off + STAGE;
    fill_off = f1 >= 4 * STAGE ? f1 - 4 * STAGE : f1;
    if constexpr (PERS) {
      if (cur_kt == kt1) {   // wave-uniform: the cursor rolls over into the next output tile
        cur_kt = 0; cur_c = 0; cur_tap = 0; cur_ky = 0; cur_kx = 0; cur_koff = 0;
        ++iss_tile;
        --iss_left;
        if (iss_left > 0) setup_tile(iss_tile);
      }
    }
  };
  // Loader waves share their SIMD's VALU issue with an MFMA wave (conv_halo.hip: every vector instruction of
  // the loaders shows up as lost MFMA issue), so the per-piece "tap inside the image? which source?" selects
  // are cached and recomputed only when the tap or the source changes — never for a plain linear layer.
  unsigned a_eff[NA];
  int eff_key = -1;
  auto issue = [&]() {   // whole tile at once (prologue; loader waves)
    const IssueCtx c = issue_begin();
    {
      const int key = c.tap * 2 + (c.second ? 1 : 0) + (PERS ? iss_tile * 32 : 0);
      if (key != eff_key) {                         // wave-uniform
        eff_key = key;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          if constexpr (!UPS) {
            a_eff[i] = ((a_mask[i] >> c.tap) & 1u) ? (c.second ? a_v2[i] : a_v1[i]) : OOB;
          } else {   // nearest-2x upsample: the source pixel is (iy>>1, ix>>1) of the virtual image
            const int iy = a_y[i] + c.ky, ix = a_x[i] + c.kx;
            const bool ok = (iy >= 0) & (iy < Hv) & (ix >= 0) & (ix < Wv);
            a_eff[i] = ok ? (unsigned)(((a_pix[i] + (iy >> 1) * p.Wi + (ix >> 1)) * c.cs + a_cc[i]) * 2) : OOB;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NA; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(c.rsA, (lptr_t)(c.sa + i * 4096), 16, a_eff[i], c.soff, 0, 0);
#pragma unroll
      for (int jj = 0; jj < NBJ; ++jj) issue_w(c, jj);
    }
    issue_advance();
  };

  // ---- ring protocol (4 stages).  Loader waves: tiles 0..2 up front, then per K tile one counted wait
  // "all but my youngest tile landed", the workgroup barrier, and the DMA of tile it+3 into the slot that
  // tile it-1 has just vacated.  Compute waves: one barrier per K tile, MFMAs of tile it while the fragments
  // of the next K half / next tile are prefetched behind them.  With PERS the stream of K tiles runs across
  // this workgroup's output tiles (the cursor rolls over inside issue(), including the per-tile address
  // setup — off the MFMA waves).
  const int total_kt = tile_count * nk;
  // Folded LayerNorm with the producer's row partials (ln_lds.h): the operands of a tile's epilogue are staged in the
  // scratch behind the ring by the LOADER waves, as extra DMA behind the K tile they issue right after the barrier
  // that opens the output tile (every MFMA wave has then left the previous tile's epilogue, which read the scratch).
  // The counted wait of the next iteration allows for them (LPT + extras youngest operations may be in flight); one
  // iteration later they are as old as the K tile issued with them and the plain wait covers both.  nk >= 3, so the
  // barrier that publishes them precedes the tile's epilogue.
  char* const ln_scr = smem + 4 * STAGE;
  const bool ln_stage = LNK == 2 && !UPS && ln_lds_usable(p, BM, nk);
  const bool ln_sts = ln_stage && ln_lds_stats(p, LN_LDS_BYTES);
  if (loader) {
    const int ln_extra = ln_stage ? ln_lds_count(p, wave, ln_sts) : 0;      // extras this wave issues per output tile
    issue();
    issue();
    issue();
    wait_vmcnt<LPT>();
    __builtin_amdgcn_s_barrier();
    bool ext_young = false;
    int kc = 0, ext_tile = tile_first;
    for (int it = 0; it < total_kt; ++it) {
      if (it > 0) {
        if (ext_young) {     // wave-uniform
          switch (ln_extra) {
            case 1: wait_vmcnt<LPT + 1>(); break;
            case 2: wait_vmcnt<LPT + 2>(); break;
            case 3: wait_vmcnt<LPT + 3>(); break;
            case 4: wait_vmcnt<LPT + 4>(); break;
            default: wait_vmcnt<LPT>(); break;
          }
          ext_young = false;
        } else {
          wait_vmcnt<LPT>();   // tile it+1 landed (this wave's share); tile it+2 may be in flight
        }
        __builtin_amdgcn_s_barrier();
      }
      issue();               // tile it+3 -> the slot of tile it-1, free since this barrier
      if (ln_stage) {
        if (kc == 0) {
          int nt, mt;
          tile_decode(p, ext_tile, mt, nt);
          ln_lds_issue<BN>(p, ln_scr, __builtin_amdgcn_readfirstlane(mt) * BM, __builtin_amdgcn_readfirstlane(nt) * BN, wave,
                           lane, ln_sts);
          ext_young = ln_extra > 0;
          ++ext_tile;
        }
        kc = kc + 1 == nk ? 0 : kc + 1;
      }
    }
    wait_vmcnt<0>();
    return;
  }
  // ---- compute waves
  f4 acc[J][MI];
  // fragment addresses inside a slot: per lane one byte offset per operand and K half (the XOR swizzle term is
  // the same for all 16-row fragments of a wave), fragment i / j adds an immediate i*2048
  const int fq = lane >> 4;
  int fa[2], fb[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    fa[s2] = lds_off(wm * WM + (lane & 15), s2 * 4 + fq) * 2;
    fb[s2] = A_BYTES + lds_off(wn * WN + (lane & 15), s2 * 4 + fq) * 2;
  }
  __builtin_amdgcn_s_barrier();   // tiles 0 and 1 landed
  __builtin_amdgcn_sched_barrier(0);
  h8 xa0[MI], wb0[J], xa1[MI], wb1[J];
#pragma unroll
  for (int i = 0; i < MI; ++i) xa0[i] = *reinterpret_cast<const h8*>(smem + fa[0] + i * 2048);
#pragma unroll
  for (int j = 0; j < J; ++j) wb0[j] = *reinterpret_cast<const h8*>(smem + fb[0] + j * 2048);
  int comp_off = 0;
  bool started = false;
  for (int tl = 0; tl < tile_count; ++tl) {
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = f4{0.f, 0.f, 0.f, 0.f};
    [[maybe_unused]] float ls1[MI], ls2[MI];          // LNF: per-lane sum x, sum x^2 of rows (lane & 15) + 16 i
    constexpr int LQ = MI * 4, LPER = (LQ + MI * J - 1) / (MI * J);   // (fragment, register pair) steps per MFMA slot
    if constexpr (LNF) {
#pragma unroll
      for (int i = 0; i < MI; ++i) ls1[i] = ls2[i] = 0.f;
    }
    for (int it = 0; it < nk; ++it) {
      if (started) __builtin_amdgcn_s_barrier();
      started = true;
      __builtin_amdgcn_sched_barrier(0);
      const int n1 = comp_off + STAGE;
      const int next_off = n1 >= 4 * STAGE ? 0 : n1;
      const char* ca = smem + comp_off + fa[1];
      const char* cb1 = smem + comp_off + fb[1];
      const char* na = smem + next_off + fa[0];
      const char* nb = smem + next_off + fb[0];
      // hand-interleaved: every MFMA is followed by at most one fragment read (MI + J reads behind MI * J MFMAs),
      // issued in the order the next half consumes them: wb[0], xa[0..MI), wb[1..J)
#pragma unroll
      for (int k = 0; k < MI * J; ++k) {     // first K half; the second half's fragments stream in behind
        const int jj = k / MI, ii = k % MI;
        acc[jj][ii] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb0[jj], xa0[ii], acc[jj][ii], 0, 0, 0);
        if constexpr (LNF) {
#pragma unroll
          for (int t2 = 0; t2 < LPER; ++t2)
            if (k * LPER + t2 < LQ) ln_acc_pair(xa0[(k * LPER + t2) / 4], (k * LPER + t2) % 4, ls1[(k * LPER + t2) / 4], ls2[(k * LPER + t2) / 4]);
        }
        if (k == 0) wb1[0] = *reinterpret_cast<const h8*>(cb1);
        else if (k <= MI) xa1[k - 1] = *reinterpret_cast<const h8*>(ca + (k - 1) * 2048);
        else if (k < MI + J) wb1[k - MI] = *reinterpret_cast<const h8*>(cb1 + (k - MI) * 2048);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int k = 0; k < MI * J; ++k) {     // second K half; prefetch of tile it+1's first half
        const int jj = k / MI, ii = k % MI;
        acc[jj][ii] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb1[jj], xa1[ii], acc[jj][ii], 0, 0, 0);
        if constexpr (LNF) {
#pragma unroll
          for (int t2 = 0; t2 < LPER; ++t2)
            if (k * LPER + t2 < LQ) ln_acc_pair(xa1[(k * LPER + t2) / 4], (k * LPER + t2) % 4, ls1[(k * LPER + t2) / 4], ls2[(k * LPER + t2) / 4]);
        }
        if (k == 0) wb0[0] = *reinterpret_cast<const h8*>(nb);
        else if (k <= MI) xa0[k - 1] = *reinterpret_cast<const h8*>(na + (k - 1) * 2048);
        else if (k < MI + J) wb0[k - MI] = *reinterpret_cast<const h8*>(nb + (k - MI) * 2048);
        __builtin_amdgcn_sched_barrier(0);
      }
      comp_off = next_off;
      __builtin_amdgcn_sched_barrier(0);
    }
    int nt, mt;
    tile_decode(p, tile_first + tl, mt, nt);
    if constexpr (LNF) {
      ln_finish<MI>(ls1, ls2, p.K, p.ln_eps);
      igemm_epilogue<J, MI, WM, WN>(p, acc, mt * BM, nt * BN, wm, wn, lane, z, smem, ls1, ls2);
    } else {
      igemm_epilogue<J, MI, WM, WN>(p, acc, mt * BM, nt * BN, wm, wn, lane, z, smem, nullptr, nullptr, smem + 4 * STAGE,
                                    ln_sts ? ln_scr + LN_LDS_STATS : nullptr, ln_stage ? ln_scr : nullptr);
    }
  }
#endif
}

template <int BM, int BN>
constexpr int smem_bytes() {   // ring + the epilogue's scratch (GroupNorm partials; 128-row tiles: the staged LayerNorm operands)
  return 4 * (BM + BN) * BK * 2 + (BM == 128 ? LN_LDS_BYTES : 4096);
}

template <int BM, int BN, bool UPS, bool PERS, int LNK>
int set_attr() {
  DADD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_dma_kernel<BM, BN, UPS, PERS, LNK>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes<BM, BN>()));
  return DADD_OK;
}

int g_num_cu = 0;

template <int BM, int BN, bool UPS, bool PERS, int LNK>
void launch1(const char* name, const IgemmArgs& a, dim3 grid, hipStream_t s) {
  dadd_launch({name, dadd_igemm_flop(a), dadd_igemm_bytes(a)}, igemm_dma_kernel<BM, BN, UPS, PERS, LNK>, grid, dim3(512),
              smem_bytes<BM, BN>(), s, a);
}
// plain linears / convs and (no upsample) their LayerNorm-folded twins; `names`: kernel name per LNK (profiling records)
template <int BM, int BN, bool UPS, bool PERS>
void launch(const char* const (&names)[3], const IgemmArgs& a, dim3 grid, hipStream_t s) {
  if constexpr (!UPS) {
    if ((a.flags & DADD_EPI_LNFOLD) && a.ln_stats_in == nullptr) {     // the kernel sums the rows itself
      launch1<BM, BN, false, PERS, 1>(names[1], a, grid, s);
      return;
    }
    if constexpr (BM == 128) {
      if (a.flags & DADD_EPI_LNFOLD) {                                   // the producer's partials, staged in LDS
        launch1<BM, BN, false, PERS, 2>(names[2], a, grid, s);
        return;
      }
    }
  }
  launch1<BM, BN, UPS, PERS, 0>(names[0], a, grid, s);
}
template <int BM, int BN, bool PERS>
int set_attr2() {
  int rc = set_attr<BM, BN, false, PERS, 0>();
  if (rc == DADD_OK) rc = set_attr<BM, BN, false, PERS, 1>();
  if constexpr (BM == 128) {
    if (rc == DADD_OK) rc = set_attr<BM, BN, false, PERS, 2>();
  }
  return rc;
}

}  // namespace

int dadd_init_igemm_dma() {
  int rc = set_attr2<128, 128, true>();
  if (rc == DADD_OK) rc = set_attr2<128, 160, true>();
  if (rc == DADD_OK) rc = set_attr2<128, 128, false>();
  if (rc == DADD_OK) rc = set_attr2<128, 160, false>();
  if (rc == DADD_OK) rc = set_attr<128, 128, true, false, 0>();
  if (rc == DADD_OK) rc = set_attr<128, 160, true, false, 0>();
  if (rc == DADD_OK) rc = set_attr2<64, 64, false>();
  if (rc == DADD_OK) rc = set_attr2<64, 128, false>();
  if (rc == DADD_OK) rc = set_attr2<64, 160, false>();
  int dev = 0;
  DADD_HIP(hipGetDevice(&dev));
  DADD_HIP(hipDeviceGetAttribute(&g_num_cu, hipDeviceAttributeMultiprocessorCount, dev));
  return rc;
}

// persistent ring over several output tiles: 128-row tiles, more tiles than CUs, no split-K, no upsample gather
bool dadd_igemm_dma_persistent(const IgemmArgs& a, int nsplit) {
  return (a.flags & DADD_TUNE_PERSIST) && nsplit == 1 && !a.ups && g_num_cu > 0 && a.mtiles * a.ntiles > g_num_cu;
}

// `a.mtiles` / `a.ntiles` are the tile counts for (tile_m, tile_n), set by the caller
int dadd_launch_igemm_dma(const IgemmArgs& a, int tile_m, int tile_n, int nsplit, hipStream_t s) {
  const int total = a.mtiles * a.ntiles;
  // buffer offsets are 32-bit with bit 31 reserved as the out-of-range marker
  DADD_REQUIRE((size_t)a.B * a.Hi * a.Wi * (size_t)(a.C1 > a.C2 ? a.C1 : a.C2) * 2 < 0x7FF00000ull &&
                   (size_t)a.N * a.K * 2 < 0x7FF00000ull,
               "igemm(dma): operand larger than the 2 GiB buffer window");
  if (tile_m == 128 && dadd_igemm_dma_persistent(a, nsplit)) {
    dim3 grid(g_num_cu);
    if (tile_n == 160) { static const char* const nm[3] = {"igemm_dma_kernel<128, 160, false, true, 0>", "igemm_dma_kernel<128, 160, false, true, 1>", "igemm_dma_kernel<128, 160, false, true, 2>"}; launch<128, 160, false, true>(nm, a, grid, s); }
    else { static const char* const nm[3] = {"igemm_dma_kernel<128, 128, false, true, 0>", "igemm_dma_kernel<128, 128, false, true, 1>", "igemm_dma_kernel<128, 128, false, true, 2>"}; launch<128, 128, false, true>(nm, a, grid, s); }
    DADD_LAUNCH_CHECK();
    return DADD_OK;
  }
  dim3 grid(total, nsplit);
  if (tile_m == 64) {
    DADD_REQUIRE(!a.ups, "igemm(dma): the 64-row tiles have no upsample gather");
    if (tile_n == 160) { static const char* const nm[3] = {"igemm_dma_kernel<64, 160, false, false, 0>", "igemm_dma_kernel<64, 160, false, false, 1>", "igemm_dma_kernel<64, 160, false, false, 2>"}; launch<64, 160, false, false>(nm, a, grid, s); }
    else if (tile_n == 128) { static const char* const nm[3] = {"igemm_dma_kernel<64, 128, false, false, 0>", "igemm_dma_kernel<64, 128, false, false, 1>", "igemm_dma_kernel<64, 128, false, false, 2>"}; launch<64, 128, false, false>(nm, a, grid, s); }
    else { static const char* const nm[3] = {"igemm_dma_kernel<64, 64, false, false, 0>", "igemm_dma_kernel<64, 64, false, false, 1>", "igemm_dma_kernel<64, 64, false, false, 2>"}; launch<64, 64, false, false>(nm, a, grid, s); }
  } else if (tile_n == 160) {
    if (a.ups) { static const char* const nm[3] = {"igemm_dma_kernel<128, 160, true, false, 0>", "igemm_dma_kernel<128, 160, true, false, 1>", "igemm_dma_kernel<128, 160, true, false, 2>"}; launch<128, 160, true, false>(nm, a, grid, s); }
    else { static const char* const nm[3] = {"igemm_dma_kernel<128, 160, false, false, 0>", "igemm_dma_kernel<128, 160, false, false, 1>", "igemm_dma_kernel<128, 160, false, false, 2>"}; launch<128, 160, false, false>(nm, a, grid, s); }
  } else {
    if (a.ups) { static const char* const nm[3] = {"igemm_dma_kernel<128, 128, true, false, 0>", "igemm_dma_kernel<128, 128, true, false, 1>", "igemm_dma_kernel<128, 128, true, false, 2>"}; launch<128, 128, true, false>(nm, a, grid, s); }
    else { static const char* const nm[3] = {"igemm_dma_kernel<128, 128, false, false, 0>", "igemm_dma_kernel<128, 128, false, false, 1>", "igemm_dma_kernel<128, 128, false, false, 2>"}; launch<128, 128, false, false>(nm, a, grid, s); }
  }
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}
