// 3x3 / stride-1 / pad-1 convolution with the activation HALO resident in LDS.
//
// Why.  The LDS-DMA implicit GEMM (igemm_dma.hip) re-loads the 128-pixel activation tile once per tap:
// 16 KB of activations + 20 KB of weights per 64-deep K tile.  The nine taps of one 64-channel chunk read
// shifted copies of the same (R+2) x (W+2) pixel halo (R = 128 / W rows of the output tile), so this kernel
// loads the halo ONCE per chunk (33.8 KB at W = 64, i.e. 3.8 KB per tap instead of 16 KB) and forms the nine
// A operands by shifted fragment reads: 23.8 KB instead of 36 KB of global->LDS traffic per K tile.
//
// Structure (same tile, MFMA and epilogue as igemm_dma.hip):
//   * 128 output pixels (whole image rows) x 160 channels per workgroup, 8 waves: waves 0-3 (one per SIMD)
//     only read fragments and issue MFMAs, waves 4-7 only move data; K is walked chunk-major, taps fastest:
//     k = tap * Cin + chunk * 64 + [0, 64);
//   * LDS: weight ring 4 x 20 KB | halo buffer x 2 (34 KB each) | 4 KB dump for dead pieces;
//     halo pixel p lives at p * 128 B with its eight 16-B chunks XOR-swizzled by p & 7: a ds_read_b128
//     fragment read of 16 consecutive pixels starting at ANY pixel (any tap shift) is bank-conflict free
//     (the (p >> 1) & 7 swizzle of the aligned GEMM tiles is not: 23 % conflict cycles measured);
//   * loader waves move data by LDS-DMA (buffer_load_dwordx4 ... offen lds, 8 pixels x 128 B or 8 weight rows
//     per instruction; out-of-image pixels use an out-of-range offset = hardware zero fill): per tap one halo
//     piece of the NEXT chunk and then the five weight pieces of tap + 3, behind one counted `vmcnt(5)` and one
//     raw s_barrier per tap;
//   * the MFMA loop is branch-free: every tap prefetches the next tap's fragments under its MFMAs, also across
//     chunk boundaries (the pixels tap (0, 0) needs are in the halo pieces issued during taps 0..4).
// Split-K slices are ranges of chunks (fp32 slabs + splitk_finish_kernel, as for the implicit GEMM).
#include <cstdlib>
#include <type_traits>

#include "dadd_common.h"
#include "igemm_args.h"
#include "igemm_epilogue.h"

// Diagnostic builds only (the product library is built without the macro): 3 = s_memtime stamps around the
// barrier and the two K halves of one MFMA wave per workgroup, summed into p.partial[workgroup][8] (uint64);
// 5 / 6 = loader waves without their DMA stream (results are garbage).
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
constexpr int HALO_MAX_PIX = 272;                  // >= (R+2)*(W+2) for W in {16, 32, 64}; 34 pieces of 8 pixels
constexpr int HALO_BYTES = HALO_MAX_PIX * 128;
constexpr int DUMP_OFF = W_RING + 2 * HALO_BYTES;
constexpr int GN_OFF = DUMP_OFF + 4 * 1024;        // statistics scratch of the epilogue (igemm_epilogue.h)
constexpr int SMEM_BYTES = GN_OFF + 4 * 1024;
constexpr int SMEM_BYTES_GNIN = GN_OFF + 8 * 1024;  // DADD_PRE_GN: scale / shift of <= 1024 input channels (the whole 160 KB)
constexpr bool DO_LOAD = DADD_IGEMM_EXP != 5 && DADD_IGEMM_EXP != 6;

// DADD_PRE_GN: where the (scale, shift) pair of input channel `ch` lives.  Channels 0..1023 in the 8 KB behind the dump
// area; the rest in the tail of the second halo buffer that the narrower maps never fill (their dead pieces go to the
// dump area): 128 more channels at W = 64, 1024 at W = 32, 1408 at W = 16.
template <int WT>
constexpr int gn_tab_extra() { return (HALO_BYTES - (((BM / WT + 2) * (WT + 2) + 7) / 8) * 1024) / 8; }
template <int WT>
__device__ __forceinline__ int gn_tab_off(int ch) {
  return ch < 1024 ? GN_OFF + ch * 8 : W_RING + 2 * HALO_BYTES - gn_tab_extra<WT>() * 8 + (ch - 1024) * 8;
}

typedef __attribute__((address_space(3))) void* lptr_t;
typedef unsigned u4v __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0x80000000u;

// WT = image width (64, 32, 16): compile-time so that the fragments of one image row share ONE address register
// (they are 16 pixels = 2048 B apart and 16 = 0 mod 8 keeps the swizzle term) and the MFMA waves compute
// 64 / WT addresses per tap instead of four.
//
// DUO: TWO MFMA waves per SIMD (12 waves: 0-3 take K half 0 of every tap, 4-7 K half 1, 8-11 move data).  With one
// MFMA wave per SIMD the matrix pipe idles whenever that wave waits for a fragment read (measured with s_memtime
// stamps: 1390 cycles per tap for 640 cycles of MFMA); the two waves of a SIMD hold partial sums over disjoint halves
// of K, read the same number of fragments in total (no extra LDS traffic) and fill each other's stalls.  Fragments
// are single-buffered: a register is reloaded for the next tap right behind the MFMA that used it last.  After the
// last tap waves 4-7 park their accumulators in LDS, waves 0-3 add them and run the epilogue.
// MEASURED (profiles/r02_zn_halo_duo_ab.txt, same box, one UNet step): 42.1 / 36.4 / 35.9 us against 40.5 / 35.5 / 36.1 us
// for the one-wave build at W = 64 / 32 / 16 — no gain: the tap time is set by the LDS-DMA stream (three weight tiles
// = 60 KB in flight per CU at ~1.5 us of loaded latency), not by stalls of the MFMA waves.  Kept as the A/B build
// (DADD_TUNE_SHALLOW on a 3x3 / stride-1 conv), not the default.
//
// GNIN (DADD_PRE_GN): GroupNorm (+ SiLU) of the INPUT applied to the halo in LDS.  While the prologue DMA is in flight
// the four MFMA waves (idle then) reduce the chunk partials of the sample to mean / rstd — each wave its own eight
// groups, the arithmetic of gn_apply_kernel — and write scale / shift of every input channel to an LDS table (8 KB
// behind the dump area: Cin <= 1024).  Every loader wave then normalises the halo piece IT loaded, one tap after
// issuing it (its own counted wait has covered it): one ds_read_b128 of data, four of table, ~80 vector instructions,
// one ds_write_b128 per lane and tap — affordable because the tap time is set by the DMA stream, not by issue slots.
// Out-of-image pixels stay the zeros the DMA wrote (the conv pads the NORMALISED tensor).  Replaces a gn_apply
// launch (7 us + a read and a write of the tensor) per conv.
template <int WT, bool DUO, bool GNIN = false>
__global__ __launch_bounds__(DUO ? 768 : 512, 1) void conv3x3_halo_kernel(const IgemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool loader = wave_all >= (DUO ? 8 : 4);
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

  constexpr int W = WT, WH = WT + 2;
  const int H = p.Ho;
  constexpr int R = BM / W;                         // output rows per tile
  constexpr int HP = (R + 2) * WH;                  // halo pixels
  const int tpi = (H * W) / BM;                     // tiles per image
  const int b = mt / tpi;
  const int y0 = (mt - b * tpi) * R;

  if (loader) {
    const int lrow = lane >> 3, lch = lane & 7;
    const int live_pieces = (HP + 7) >> 3;
    const size_t pix_total = (size_t)p.B * H * W;
    const int rec1 = (int)(pix_total * p.C1 * 2), rec2 = (int)(pix_total * p.C2 * 2);
    const int recW = (int)((size_t)p.N * p.K * 2);
    const half_t* base2 = p.x2 ? p.x2 : p.x;
    unsigned w_v[NBJ];                              // weight piece j: byte offset of this lane's 16 B
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
      const int row = (j * 4 + wave) * 8 + lrow;
      const int n = n0 + row;
      w_v[j] = (n < p.N) ? (unsigned)(((size_t)n * p.K + (lch ^ ((row >> 1) & 7)) * 8) * 2) : OOB;
    }
    // Halo piece `s` (s = 0..8) of this wave covers pixels (s * 4 + wave) * 8 + [0, 8): byte offsets of this
    // lane's 16 B in both sources, precomputed once; the tap loop below is unrolled by nine so that the slot
    // index is a constant (a runtime-indexed table would land in scratch) and the loader waves — which share
    // their SIMD's VALU issue with an MFMA wave — run no vector arithmetic in the steady state.
    unsigned hv1[9], hv2[9];
    [[maybe_unused]] unsigned okmask = 0;           // GNIN: bit sl = this lane's pixel of slot sl lies inside the image
#pragma unroll
    for (int sl = 0; sl < 9; ++sl) {
      const int px = (sl * 4 + wave) * 8 + lrow;
      const int hy = px / WH, hx = px - hy * WH;
      const int y = y0 - 1 + hy, x = hx - 1;
      const bool ok = px < HP && y >= 0 && y < H && x >= 0 && x < W;
      okmask |= ok ? 1u << sl : 0u;
      const int chunk = lch ^ (px & 7);
      const int pix = (b * H + y) * W + x;
      hv1[sl] = ok ? (unsigned)((pix * p.C1 + chunk * 8) * 2) : OOB;
      hv2[sl] = ok ? (unsigned)((pix * p.C2 + chunk * 8) * 2) : OOB;
    }
    int wk_tap = 0, wk_c = c0 * BK, wk_gi = 0;      // cursor of the NEXT weight tile to fetch
    // halo piece `SL` of chunk c -> buffer (c - c0) & 1; pieces past the halo go to the dump area
    auto issue_halo = [&](int c, auto SLC) {
      constexpr int SL = decltype(SLC)::value;
      const bool have = c < c1;
      const bool second = (c * BK) >= p.C1;
      const int cb = second ? c * BK - p.C1 : c * BK;
      const bool live = have && (SL * 4 + wave) < live_pieces;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(second ? base2 : p.x), 0, have ? (second ? rec2 : rec1) : 0, 0x00020000);
      const int lmask = live ? -1 : 0;              // mask arithmetic, no branch
      const int dst = ((W_RING + ((c - c0) & 1) * HALO_BYTES + (SL * 4 + wave) * 1024) & lmask) |
                      ((DUMP_OFF + wave * 1024) & ~lmask);
      if constexpr (DO_LOAD)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(smem + dst), 16, second ? hv2[SL] : hv1[SL],
                                                 (unsigned)(cb * 2), 0, 0);
    };
    auto issue_w = [&]() {                          // weight tile wk_gi -> ring slot wk_gi & 3 (dead past the end)
      const __amdgpu_buffer_rsrc_t rw =
          __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, wk_gi < n_it ? recW : 0, 0x00020000);
      char* wdst = smem + (wk_gi & 3) * B_BYTES + wave * 1024;
      const unsigned koff = (unsigned)((wk_tap * Cin + wk_c) * 2);
#pragma unroll
      for (int j = 0; j < NBJ; ++j)
        if constexpr (DO_LOAD)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lptr_t)(wdst + j * 4096), 16, w_v[j], koff, 0, 0);
      ++wk_gi;
      const int w1 = wk_tap + 1, ww = w1 == 9 ? 1 : 0;
      wk_tap = ww ? 0 : w1;
      wk_c += ww ? BK : 0;
    };

    // GNIN: normalise (+ SiLU) the piece of slot SL of chunk `cp` that this wave loaded, in place
    [[maybe_unused]] auto gn_piece = [&](int cp, auto SLC) {
      constexpr int SL = decltype(SLC)::value;
      if (cp >= c1 || (SL * 4 + wave) >= live_pieces) return;        // wave-uniform: no such piece
      char* ptr = smem + W_RING + ((cp - c0) & 1) * HALO_BYTES + (SL * 4 + wave) * 1024 + lane * 16;
      const int px = (SL * 4 + wave) * 8 + lrow;
      const int ch0 = cp * BK + (lch ^ (px & 7)) * 8;                 // first of this lane's eight input channels
      const h8 v = *reinterpret_cast<const h8*>(ptr);
      const f4* tb = reinterpret_cast<const f4*>(smem + gn_tab_off<WT>(ch0));   // (scale, shift) of eight channels
      const f4 t0 = tb[0], t1 = tb[1], t2 = tb[2], t3 = tb[3];
      const float sc[8] = {t0[0], t0[2], t1[0], t1[2], t2[0], t2[2], t3[0], t3[2]};
      const float sh[8] = {t0[1], t0[3], t1[1], t1[3], t2[1], t2[3], t3[1], t3[3]};
      h8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)v[e] * sc[e] + sh[e];
        if (p.flags & DADD_PRE_GN_SILU) f = dadd_silu(f);
        o[e] = (half_t)f;
      }
      if ((okmask >> SL) & 1u) *reinterpret_cast<h8*>(ptr) = o;      // padding pixels keep their zeros
    };

    // ---- prologue: the whole halo of the first chunk, weight tiles 0, 1, 2
    issue_halo(c0, std::integral_constant<int, 0>{});
    issue_halo(c0, std::integral_constant<int, 1>{});
    issue_halo(c0, std::integral_constant<int, 2>{});
    issue_halo(c0, std::integral_constant<int, 3>{});
    issue_halo(c0, std::integral_constant<int, 4>{});
    issue_halo(c0, std::integral_constant<int, 5>{});
    issue_halo(c0, std::integral_constant<int, 6>{});
    issue_halo(c0, std::integral_constant<int, 7>{});
    issue_halo(c0, std::integral_constant<int, 8>{});
    issue_w();
    issue_w();
    issue_w();
    if constexpr (GNIN) __builtin_amdgcn_s_barrier();            // the MFMA waves have written the scale / shift table
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBJ) : "memory");   // everything but weight tile 2
    if constexpr (GNIN) {      // slots 0..7 of the first chunk (slot 8 follows the rule of the stream: one tap later)
      gn_piece(c0, std::integral_constant<int, 0>{});
      gn_piece(c0, std::integral_constant<int, 1>{});
      gn_piece(c0, std::integral_constant<int, 2>{});
      gn_piece(c0, std::integral_constant<int, 3>{});
      gn_piece(c0, std::integral_constant<int, 4>{});
      gn_piece(c0, std::integral_constant<int, 5>{});
      gn_piece(c0, std::integral_constant<int, 6>{});
      gn_piece(c0, std::integral_constant<int, 7>{});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    // ---- LDS-DMA stream.  Iteration g (after its barrier, when the ring slot of tile g - 1 is free) issues halo
    // piece t(g) of chunk(g) + 1 and THEN weight tile g + 3: the counted wait "all but the NBJ youngest" at the
    // top of the next iteration therefore covers every halo piece and weight tile g + 1.  The last piece of a
    // halo lands one tap after its chunk started, which is early enough: tap (0, 0) — the only tap prefetched
    // across the chunk boundary — reads pixels < R * (W + 2), i.e. pieces of slots 0..4 (issued at taps 0..4),
    // and slot 8 holds pixels of halo rows >= R that no tap before (2, 0) touches.
    // (Variants measured slower: two halo slots per tap so that the halo completes a tap early, 151 vs 142 us
    // on 16384x640x5760 — a dead LDS-DMA slot still costs ~115 issue cycles; register-staged loaders —
    // buffer_load into VGPRs two taps ahead, ds_write_b128 — 171 us, the loaders became the critical path.)
    auto ltap = [&](int c, auto SLC) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBJ) : "memory");
      if constexpr (GNIN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // last tap's normalised piece is written
      __builtin_amdgcn_s_barrier();
      issue_halo(c + 1, SLC);
      issue_w();
      if constexpr (GNIN) {    // BEHIND the DMA issue (the stream is the critical path; this runs while it is in flight):
        // the piece issued one tap ago has landed (the counted wait above) — slot SL - 1 of chunk c + 1, or slot 8 of
        // chunk c.  It becomes visible at the barrier after next; its first reader comes at least two taps later.
        constexpr int SL = decltype(SLC)::value;
        if constexpr (SL == 0) gn_piece(c, std::integral_constant<int, 8>{});
        else gn_piece(c + 1, std::integral_constant<int, (SL + 8) % 9>{});
      }
    };
    for (int c = c0; c < c1; ++c) {
      ltap(c, std::integral_constant<int, 0>{});
      ltap(c, std::integral_constant<int, 1>{});
      ltap(c, std::integral_constant<int, 2>{});
      ltap(c, std::integral_constant<int, 3>{});
      ltap(c, std::integral_constant<int, 4>{});
      ltap(c, std::integral_constant<int, 5>{});
      ltap(c, std::integral_constant<int, 6>{});
      ltap(c, std::integral_constant<int, 7>{});
      ltap(c, std::integral_constant<int, 8>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ---- MFMA waves
  f4 acc[J][4];
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f4{0.f, 0.f, 0.f, 0.f};
  const int fq = lane >> 4;
  constexpr int NL = 64 / W;                        // image rows (= address leaders) in a wave's 64 pixels
  constexpr int FPL = 4 / NL;                       // fragments per leader
  int p0[NL];                                       // halo pixel of tap (0, 0) for leader l
#pragma unroll
  for (int l = 0; l < NL; ++l) {
    const int q = wm * 64 + l * FPL * 16 + (lane & 15);
    const int r = q / W;
    p0[l] = r * WH + (q - r * W);
  }
  const int swb = (wn * WN + (lane & 15));
  const int fb0 = (swb * 8 + (fq ^ ((swb >> 1) & 7))) * 16;         // weight fragment, K half 0
  const int fb1 = fb0 ^ 64;                                         // K half 1: chunk index ^ 4
  // (rows j*16 further down keep the swizzle term: (row + 16 j) >> 1 & 7 == (row >> 1) & 7)
  auto a_addr = [&](int l, int dt) {                // byte offset inside a halo buffer, K half 0, leader l
    const int px = p0[l] + dt;
    return px * 128 + ((fq ^ (px & 7)) << 4);
  };

  // Leader addresses of all nine taps (independent of the chunk): the tap loop is unrolled by nine, every
  // address is a register and the MFMA waves — the critical path — run no address or tap arithmetic at all
  // (measured on 16384x640x5760: 134 us with ~30 VALU/SALU per tap on these waves, 118 us with one address
  // per image row, see DESIGN.md for this version).
  constexpr bool PRE = NL <= 2;                     // W = 16 (36 addresses) would spill: computed per tap there
  int aoff9[PRE ? 9 : 1][NL];
  if constexpr (PRE) {
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int l = 0; l < NL; ++l) aoff9[tp][l] = a_addr(l, (tp / 3) * WH + (tp % 3));
  }
  auto addr_of = [&](auto TP, int l) {
    constexpr int T = decltype(TP)::value;
    if constexpr (PRE) return aoff9[T][l];
    else {                                          // opaque copy: keeps the compiler from hoisting all 36 again
      int pz = p0[l];
      asm volatile("" : "+v"(pz));
      const int px = pz + (T / 3) * WH + (T % 3);
      return px * 128 + ((fq ^ (px & 7)) << 4);
    }
  };

  if constexpr (DUO) {
    const int hx = (wave_all >> 2) * 64;            // K half of this wave, as the XOR term of its fragment addresses
    const int fbh = fb0 ^ hx;
    __builtin_amdgcn_s_barrier();                   // halo of the first chunk and weight tiles 0..2 landed
    __builtin_amdgcn_sched_barrier(0);
    h8 xa[4], wb[J];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      xa[i] = *reinterpret_cast<const h8*>(smem + W_RING + (addr_of(std::integral_constant<int, 0>{}, i / FPL) ^ hx) + (i % FPL) * 2048);
#pragma unroll
    for (int j = 0; j < J; ++j) wb[j] = *reinterpret_cast<const h8*>(smem + fbh + j * 2048);
    int gi = 0;
    const char* hb = smem + W_RING;                 // halo buffer of the current chunk
    const char* hbo = smem + W_RING + HALO_BYTES;   // ... of the next chunk
    auto tap2 = [&](auto TP) {
      constexpr int T = decltype(TP)::value, NT = (T + 1) % 9;
      __builtin_amdgcn_s_barrier();                 // one barrier per tap on all twelve waves
      __builtin_amdgcn_sched_barrier(0);
      const char* wnext = smem + ((gi + 1) & 3) * B_BYTES + fbh;
      const char* hbn = T == 8 ? hbo : hb;          // tap (0, 0) of the next chunk reads the other buffer
#pragma unroll
      for (int k = 0; k < 4 * J; ++k) {
        const int jj = k / 4, ii = k % 4;
        acc[jj][ii] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[jj], xa[ii], acc[jj][ii], 0, 0, 0);
        if (ii == 3) wb[jj] = *reinterpret_cast<const h8*>(wnext + jj * 2048);       // last use of wb[jj] this tap
        if (jj == J - 1)                                                             // last use of xa[ii] this tap
          xa[ii] = *reinterpret_cast<const h8*>(hbn + (addr_of(std::integral_constant<int, NT>{}, ii / FPL) ^ hx) + (ii % FPL) * 2048);
        __builtin_amdgcn_sched_barrier(0);
      }
      ++gi;
      __builtin_amdgcn_sched_barrier(0);
    };
    for (int c = c0; c < c1; ++c) {
      tap2(std::integral_constant<int, 0>{});
      tap2(std::integral_constant<int, 1>{});
      tap2(std::integral_constant<int, 2>{});
      tap2(std::integral_constant<int, 3>{});
      tap2(std::integral_constant<int, 4>{});
      tap2(std::integral_constant<int, 5>{});
      tap2(std::integral_constant<int, 6>{});
      tap2(std::integral_constant<int, 7>{});
      tap2(std::integral_constant<int, 8>{});
      const char* sw = hb;
      hb = hbo;
      hbo = sw;
    }
    // ---- add the two K halves.  First barrier: every loader wave has terminated, i.e. all its DMA — the zero fills of
    // the dead tiles past the end included — has landed and the weight ring can be overwritten; the waves of half 1 park
    // their accumulators there ([register][lane] f4: conflict free), second barrier, the waves of half 0 add them.
    __builtin_amdgcn_s_barrier();
    f4* park = reinterpret_cast<f4*>(smem + wave * (J * 4 * 64 * 16));
    if (wave_all >= 4) {
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) park[(j * 4 + i) * 64 + lane] = acc[j][i];
    }
    __builtin_amdgcn_s_barrier();
    if (wave_all >= 4) return;
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j][i] += park[(j * 4 + i) * 64 + lane];
    igemm_epilogue<J, 4, 64, WN>(p, acc, m0, n0, wm, wn, lane, z, smem, nullptr, nullptr, smem + GN_OFF);
    return;
  }
  if constexpr (GNIN) {
    // mean / rstd of this wave's eight groups from the chunk partials (order and precision of gn_apply_kernel), then
    // scale / shift of their channels into the LDS table; the loader waves read it after the barrier below
    const int cg = Cin >> 5;
    const int g = wave * 8 + (lane >> 3), sub = lane & 7;
    // Skip-concat input: group g of the concatenation [x | x2] is the union of `r` consecutive groups of ONE source
    // (the host checked that the group widths nest), whose partials carry that source's own 32-group layout.
    const bool second = p.C2 > 0 && g * cg >= p.C1;
    const float* wsp = second ? p.gni_ws2 : p.gni_ws;
    const int nch = second ? p.gni_nchunk2 : p.gni_nchunk;
    const int wsrc = (second ? p.C2 : p.C1) >> 5;              // group width of the source
    const int r = p.C2 > 0 ? cg / wsrc : 1;
    const int g0 = p.C2 > 0 ? ((second ? g * cg - p.C1 : g * cg) / wsrc) : g;
    double a = 0.0, q = 0.0;
    for (int j = 0; j < r; ++j)
      for (int k = sub; k < nch; k += 32) {
        dadd_f2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kk = min(k + 8 * u, nch - 1);
          v[u] = *reinterpret_cast<const dadd_f2*>(wsp + (((size_t)b * nch + kk) * 32 + g0 + j) * 2);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (k + 8 * u < nch) {
            a += (double)v[u][0];
            q += (double)v[u][1];
          }
        }
      }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      a += __shfl_xor(a, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    const double n = (double)(H * W) * (double)cg;
    const double mu = a / n;
    double var = q / n - mu * mu;
    if (var < 0.0) var = 0.0;
    const float mean_f = (float)mu, rstd_f = (float)(1.0 / sqrt(var + (double)p.gni_eps));
    for (int cc = sub; cc < cg; cc += 8) {
      const int ch = g * cg + cc;
      const float sc = rstd_f * p.gni_gamma[ch];
      *reinterpret_cast<dadd_f2*>(smem + gn_tab_off<WT>(ch)) = dadd_f2{sc, p.gni_beta[ch] - mean_f * sc};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  __builtin_amdgcn_s_barrier();                     // halo of the first chunk and weight tiles 0..2 landed
  __builtin_amdgcn_sched_barrier(0);
  h8 xa0[4], xa1[4], wb0[J], wb1[J];
#pragma unroll
  for (int i = 0; i < 4; ++i)
    xa0[i] = *reinterpret_cast<const h8*>(smem + W_RING + addr_of(std::integral_constant<int, 0>{}, i / FPL) + (i % FPL) * 2048);
#pragma unroll
  for (int j = 0; j < J; ++j) wb0[j] = *reinterpret_cast<const h8*>(smem + fb0 + j * 2048);
  [[maybe_unused]] unsigned long long sc_bar = 0, sc_h0 = 0, sc_h1 = 0;
  DADD_STAMP(c_begin);
  int gi = 0;
  const char* hb = smem + W_RING;                   // halo buffer of the current chunk
  const char* hbo = smem + W_RING + HALO_BYTES;     // ... of the next chunk
  auto tap = [&](auto TP) {
    constexpr int T = decltype(TP)::value, NT = (T + 1) % 9;
    DADD_STAMP(c0s);
    __builtin_amdgcn_s_barrier();                   // one barrier per tap on both sides
    DADD_STAMP(c1s);
    __builtin_amdgcn_sched_barrier(0);
    const char* wcur1 = smem + (gi & 3) * B_BYTES + fb1;
    const char* wnext = smem + ((gi + 1) & 3) * B_BYTES + fb0;
    const char* hbn = T == 8 ? hbo : hb;            // tap (0, 0) of the next chunk reads the other buffer
#pragma unroll
    for (int k = 0; k < 4 * J; ++k) {               // K half 0; the half-1 fragments stream in behind
      const int jj = k / 4, ii = k % 4;
      acc[jj][ii] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb0[jj], xa0[ii], acc[jj][ii], 0, 0, 0);
      if (k == 0) wb1[0] = *reinterpret_cast<const h8*>(wcur1);
      else if (k <= 4) xa1[k - 1] = *reinterpret_cast<const h8*>(hb + (addr_of(std::integral_constant<int, T>{}, (k - 1) / FPL) ^ 64) + ((k - 1) % FPL) * 2048);
      else if (k < 4 + J) wb1[k - 4] = *reinterpret_cast<const h8*>(wcur1 + (k - 4) * 2048);
      __builtin_amdgcn_sched_barrier(0);
    }
    DADD_STAMP(c2s);
#pragma unroll
    for (int k = 0; k < 4 * J; ++k) {               // K half 1; prefetch of the next tap's half 0
      const int jj = k / 4, ii = k % 4;
      acc[jj][ii] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb1[jj], xa1[ii], acc[jj][ii], 0, 0, 0);
      if (k == 0) wb0[0] = *reinterpret_cast<const h8*>(wnext);
      else if (k <= 4) xa0[k - 1] = *reinterpret_cast<const h8*>(hbn + addr_of(std::integral_constant<int, NT>{}, (k - 1) / FPL) + ((k - 1) % FPL) * 2048);
      else if (k < 4 + J) wb0[k - 4] = *reinterpret_cast<const h8*>(wnext + (k - 4) * 2048);
      __builtin_amdgcn_sched_barrier(0);
    }
    ++gi;
    __builtin_amdgcn_sched_barrier(0);
    DADD_STAMP(c3s);
    DADD_ACC(sc_bar, c0s, c1s);
    DADD_ACC(sc_h0, c1s, c2s);
    DADD_ACC(sc_h1, c2s, c3s);
  };
  for (int c = c0; c < c1; ++c) {
    tap(std::integral_constant<int, 0>{});
    tap(std::integral_constant<int, 1>{});
    tap(std::integral_constant<int, 2>{});
    tap(std::integral_constant<int, 3>{});
    tap(std::integral_constant<int, 4>{});
    tap(std::integral_constant<int, 5>{});
    tap(std::integral_constant<int, 6>{});
    tap(std::integral_constant<int, 7>{});
    tap(std::integral_constant<int, 8>{});
    const char* sw = hb;
    hb = hbo;
    hbo = sw;
  }
#if DADD_IGEMM_EXP == 3
  if (wave == 0 && lane == 0 && p.partial && blockIdx.y == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.partial) + (size_t)blockIdx.x * 8;
    o[0] = sc_bar; o[1] = sc_h0; o[2] = sc_h1; o[3] = __builtin_amdgcn_s_memtime() - c_begin;
  }
#endif
  igemm_epilogue<J, 4, 64, WN>(p, acc, m0, n0, wm, wn, lane, z, smem, nullptr, nullptr, smem + GN_OFF);
#endif
}

template <int WT, bool DUO, bool GNIN = false>
int set_attr_halo() {
  DADD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<WT, DUO, GNIN>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, GNIN ? SMEM_BYTES_GNIN : SMEM_BYTES));
  return DADD_OK;
}

}  // namespace

int dadd_init_conv_halo() {
  int rc = set_attr_halo<64, false>();
  if (rc == DADD_OK) rc = set_attr_halo<32, false>();
  if (rc == DADD_OK) rc = set_attr_halo<16, false>();
  if (rc == DADD_OK) rc = set_attr_halo<64, true>();
  if (rc == DADD_OK) rc = set_attr_halo<32, true>();
  if (rc == DADD_OK) rc = set_attr_halo<16, true>();
  if (rc == DADD_OK) rc = set_attr_halo<64, false, true>();
  if (rc == DADD_OK) rc = set_attr_halo<32, false, true>();
  if (rc == DADD_OK) rc = set_attr_halo<16, false, true>();
  return rc;
}

// DADD_PRE_GN: input channels whose (scale, shift) fit in LDS for a map of width Wo
int dadd_conv_halo_gn_channels(int Wo) {
  return 1024 + (Wo == 64 ? gn_tab_extra<64>() : (Wo == 32 ? gn_tab_extra<32>() : gn_tab_extra<16>()));
}

// Shapes this kernel takes (everything else stays on the implicit GEMM).
bool dadd_conv_halo_applicable(const IgemmArgs& a, int tile_n) {
  return a.taps == 9 && a.stride == 1 && a.pad == 1 && !a.ups && tile_n == BN &&
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
  const double flop = dadd_igemm_flop(a), bytes = dadd_igemm_bytes(a);
  const bool duo = (a.flags & DADD_TUNE_SHALLOW) != 0;   // A/B switch: measured equal to the one-wave build (see the kernel comment)
  if (a.flags & DADD_PRE_GN) {
    if (a.Wo == 64) dadd_launch({"conv3x3_halo_kernel<64, false, true>", flop, bytes}, conv3x3_halo_kernel<64, false, true>, grid, dim3(512), SMEM_BYTES_GNIN, s, a);
    else if (a.Wo == 32) dadd_launch({"conv3x3_halo_kernel<32, false, true>", flop, bytes}, conv3x3_halo_kernel<32, false, true>, grid, dim3(512), SMEM_BYTES_GNIN, s, a);
    else dadd_launch({"conv3x3_halo_kernel<16, false, true>", flop, bytes}, conv3x3_halo_kernel<16, false, true>, grid, dim3(512), SMEM_BYTES_GNIN, s, a);
  } else if (duo) {
    if (a.Wo == 64) dadd_launch({"conv3x3_halo_kernel<64, true, false>", flop, bytes}, conv3x3_halo_kernel<64, true>, grid, dim3(768), SMEM_BYTES, s, a);
    else if (a.Wo == 32) dadd_launch({"conv3x3_halo_kernel<32, true, false>", flop, bytes}, conv3x3_halo_kernel<32, true>, grid, dim3(768), SMEM_BYTES, s, a);
    else dadd_launch({"conv3x3_halo_kernel<16, true, false>", flop, bytes}, conv3x3_halo_kernel<16, true>, grid, dim3(768), SMEM_BYTES, s, a);
  } else {
    if (a.Wo == 64) dadd_launch({"conv3x3_halo_kernel<64, false, false>", flop, bytes}, conv3x3_halo_kernel<64, false>, grid, dim3(512), SMEM_BYTES, s, a);
    else if (a.Wo == 32) dadd_launch({"conv3x3_halo_kernel<32, false, false>", flop, bytes}, conv3x3_halo_kernel<32, false>, grid, dim3(512), SMEM_BYTES, s, a);
    else dadd_launch({"conv3x3_halo_kernel<16, false, false>", flop, bytes}, conv3x3_halo_kernel<16, false>, grid, dim3(512), SMEM_BYTES, s, a);
  }
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}
