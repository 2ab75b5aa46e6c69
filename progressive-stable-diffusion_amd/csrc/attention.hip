// Attention on gfx950 MFMA.
//
// Both kernels compute the TRANSPOSED score tile  S^T = K * Q^T  (keys on the MFMA rows, queries on
// the lanes): a lane then owns one query column, so the row-wise softmax needs only two xor
// shuffles (lanes +16, +32) and the probabilities come out of the accumulator already shaped as the
// B operand of the next product  O^T = V^T * P^T  (the contraction index — the key — may be
// permuted freely as long as both operands use the same permutation: slot (g,j) of a 32-key step is
// key 4g+j of the first 16-key fragment for j<4 and key 4g+j-4 of the second for j>=4).  V stays
// row-major in LDS and is read transposed by ds_read_b64_tr_b16.  No score matrix, no P round trip.
//
//  * flash_kernel   : self-attention, online softmax over 64-key tiles (attn1, VAE mid block)
//  * xattn_kernel   : the DADD cross-attention — 2 or 3 sixteen-key pathways with INDEPENDENT
//                     softmaxes, gate/lambda folded into the probabilities, K/V fragments resident
//                     in registers for the whole block (they are step-invariant and tiny)
#include <stdlib.h>

#include <string>

#include "dadd_common.h"
#include "igemm_args.h"   // xcd_remap

namespace {

constexpr int round_up(int a, int b) { return (a + b - 1) / b * b; }
// V row stride (halfs) such that 8 consecutive rows x 32 B hit distinct banks for the tr reads.
constexpr int v_stride(int dvp) { return ((dvp * 2) % 64 == 32) ? dvp : dvp + 16; }

__device__ __forceinline__ h8 tr_pair(const half_t* lds_row_lo, const half_t* lds_row_hi) {
  typedef __attribute__((address_space(3))) fp16x4 lds_v4;
  const fp16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_v4*)lds_row_lo);
  const fp16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_v4*)lds_row_hi);
  h8 r;
  r[0] = (half_t)a[0]; r[1] = (half_t)a[1]; r[2] = (half_t)a[2]; r[3] = (half_t)a[3];
  r[4] = (half_t)b[0]; r[5] = (half_t)b[1]; r[6] = (half_t)b[2]; r[7] = (half_t)b[3];
  return r;
}

__device__ __forceinline__ h8 pack_p(const f4& lo, const f4& hi) {
  h8 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[j] = (half_t)lo[j];
    r[j + 4] = (half_t)hi[j];
  }
  return r;
}

// Maxima as the bare instructions: through fmaxf the compiler first canonicalises every operand that comes out of an
// MFMA or a cross-lane shuffle (v_max_f32 v, v, v — IEEE maxnum must quiet signalling NaNs): 20 of the ~560 vector
// instructions per 64-key tile of the VALU-bound d = 40 kernel.  Scores are finite here (masking uses -3e38, not inf).
__device__ __forceinline__ float vmax2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

struct FlashArgs {
  const half_t* q;
  const half_t* k;
  const half_t* v;
  half_t* out;
  int B, Nq, N, H, ldq, ld, ldo;   // Nq query rows (ldq), N key / value rows (ld) per sample
  float scale_log2;  // log2(e)/sqrt(d)
#ifdef DADD_FLASH_STAMPS
  unsigned long long* dbg;   // diagnostics build only (scripts/flash_stamps.py): per (block, wave) sums of phase cycles
#endif
};

// Diagnostics build (-DDADD_FLASH_STAMPS, never the shipped library): s_memtime between the phases of a tile, pinned with
// scheduling barriers (which also forbid the interleaving the product build relies on: the sums tell where a wave waits, not
// what the product kernel costs).
#ifdef DADD_FLASH_STAMPS
#define FSTAMP(i)                                                    \
  {                                                                  \
    __builtin_amdgcn_sched_barrier(0);                               \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
    stamp_acc[i] += now_ - stamp_last;                               \
    stamp_last = now_;                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
  }
#else
#define FSTAMP(i)
#endif

typedef float f2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ h8 pack_p8(const f4& lo, const f4& hi) {   // 8 floats -> 8 halfs, pairwise
  const h2 a = __builtin_convertvector((f2v){lo[0], lo[1]}, h2);
  const h2 b = __builtin_convertvector((f2v){lo[2], lo[3]}, h2);
  const h2 c = __builtin_convertvector((f2v){hi[0], hi[1]}, h2);
  const h2 d = __builtin_convertvector((f2v){hi[2], hi[3]}, h2);
  h8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = b[0]; r[3] = b[1];
  r[4] = c[0]; r[5] = c[1]; r[6] = d[0]; r[7] = d[1];
  return r;
}

// K tile image in LDS: KP panels of [64 keys][64 halfs], the igemm XOR swizzle inside a panel
// (measured conflict-free for the 16-row ds_read_b128 fragment reads).
__device__ __forceinline__ int kpanel_off(int row, int chunk) {
  return (row * 8 + (chunk ^ ((row >> 1) & 7))) * 8;
}

// DR = real head dim. Block = 4 waves x (16*QF) queries; KV tile = 64 keys.
//
// VALU is the bound of this kernel (PMC: VALU issue 73 % of the time, MFMA 23 %), so the inner loop
// is built to minimise vector instructions: all tile addressing is hoisted out of the loop, scores
// stay unscaled in the accumulator with the scale fused into the exponent (one fma + one v_exp_f32
// per score), the mask runs only on a ragged last tile, O is rescaled only when a row maximum moved,
// probabilities are converted pairwise, and — when the head dim leaves a spare column in its
// 16-multiple (d = 40 -> 48) — a column of ones in V makes the PV MFMA produce the softmax row sums.
// (two blocks per CU for the small head dims: 256 VGPRs per lane at most — at 260 the d = 40 kernel ran one block per
// CU and took 242 instead of 172 us.  Round 3 built the one-block-per-CU variant VERDICT r2 asked for — S(t+1) and
// PV(t-1) MFMAs in one basic block with the softmax of tile t, two score and two probability register sets, four K/V
// buffers: it needs ~304 live registers per lane, beyond the 256 ARCHITECTURAL VGPRs a wave has even when it owns all
// 512 (the other 256 are AGPRs, reachable by VALU only through v_accvgpr moves): hipcc emitted 1,160 of those moves per
// iteration pair and the kernel took 327 us against 171 (same box, profiles/r03_o_flash_pipe_ab.txt).  Not kept; what
// the counters say about this kernel is in profiles/r03_n_pmc_flash.txt: VALU issuing 50 % of the time, MFMA pipe 30 %.)
template <int DR, int QF, bool PREFETCH, int NW = 4>
__global__ __launch_bounds__(NW * 64, DR <= 96 ? 2 : 1) void flash_kernel(const FlashArgs p) {
  constexpr int NT = NW * 64, QB = NW * 16 * QF;   // threads, queries per block
  constexpr bool DEEP = PREFETCH && DR <= 40;   // two tiles in flight where the registers allow it
  constexpr int D = round_up(DR, 32), DVP = round_up(DR, 16);
  constexpr int KS = D / 32, DF = DVP / 16, DC = DR / 8;
  constexpr int KP = (D + 63) / 64, KTILE = KP * 64 * 64;       // halfs
  constexpr int VLD = v_stride(DVP), VTILE = 64 * VLD;
  constexpr int TILE_HALFS = KTILE + VTILE;
  constexpr int NL = (64 * DC + NT - 1) / NT;  // 16-byte loads per thread per tile per tensor
  constexpr bool SUMCOL = DVP > DR;          // spare V column available for the row sums
  constexpr int NBUFS = PREFETCH ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* Ks = reinterpret_cast<half_t*>(smem);   // buffer b: K at Ks + b*TILE_HALFS, V right after
  half_t* Vs = Ks + KTILE;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, li = lane & 15;
  // 1-D grid, XCD-aware: the query blocks of one (batch, head) run on ONE XCD, so its K/V (0.6-1.3 MB)
  // is fetched into one L2 instead of all eight (FETCH_SIZE 170 MB -> ~1/5 for 4x4096x8x40)
  const int nqb = (p.Nq + QB - 1) / QB;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int bh = tile / nqb, qb = tile - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int qw0 = qb * QB + wave * (16 * QF);
  const size_t tok0 = (size_t)b * p.N, qtok0 = (size_t)b * p.Nq;
  const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // zero both buffers once (padding columns are never written again); ones column for the row sums
  for (int i = t; i < NBUFS * TILE_HALFS / 8; i += NT) *reinterpret_cast<h8*>(Ks + i * 8) = zero8;
  __syncthreads();
  if (SUMCOL)
    for (int i = t; i < NBUFS * 64; i += NT) Vs[(i >> 6) * TILE_HALFS + (i & 63) * VLD + DR] = (half_t)1.0f;

  // Q fragments (B operand of S^T = K Q^T): lane -> query li, d-chunk 32s + 8g
  h8 qf[QF][KS];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    int qrow = qw0 + f * 16 + li;
    if (qrow > p.Nq - 1) qrow = p.Nq - 1;
    const half_t* qp = p.q + (qtok0 + qrow) * p.ldq + h * DR;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int dc = 32 * s + 8 * g;
      qf[f][s] = (dc < DR) ? *reinterpret_cast<const h8*>(qp + dc) : zero8;
      // the softmax scale (in log2 units) rides on Q: the accumulator then holds the exponent itself (see compute())
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[f][s][e] = (half_t)((float)qf[f][s][e] * p.scale_log2);
    }
  }

  f4 oacc[DF][QF];
  // negm[f] = -(reference maximum of query column li of fragment f, log2 units), four copies: it is the C operand of the
  // first S MFMA of a tile, so the accumulator comes out as  s*scale - m  and goes into v_exp_f32 as it is (MFMA and
  // vector instructions do not overlap on this SIMD - profiles/r03_y_mfma_valu_coissue.txt - so the fma per score that
  // this saves is time saved: 64 of ~250 vector instructions per tile)
  f4 negm[QF];
  float lrow[QF];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    negm[f] = f4{0.f, 0.f, 0.f, 0.f};
    lrow[f] = 0.f;
#pragma unroll
    for (int df = 0; df < DF; ++df) oacc[df][f] = f4{0.f, 0.f, 0.f, 0.f};
  }

  // hoisted tile addressing: what this thread loads / stores for every tile
  constexpr bool SLIM = DR <= 40;     // d = 40 sits at the 256-VGPR limit of two blocks per CU: one offset array (below)
  int g_off[SLIM ? 1 : NL], g_off_c[NL], k_lds[NL], v_lds[NL], t_row[NL];
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int idx = t + NT * u;
    const int row = idx / DC, ch = idx - row * DC;
    t_row[u] = (idx < 64 * DC) ? row : 1 << 30;      // rows past the tile never pass the key test
    // in-tile address for every thread: a thread without a row (row >= 64) gets a clamped address and never stores; for
    // the rows that exist this IS row * ld, so with SLIM it serves the ragged path too (4 spilled VGPRs less at d = 40;
    // the larger head dims keep their own array — dropping it there changed the schedule for the worse: d = 80 33 -> 35 us)
    g_off_c[u] = min(row, 63) * p.ld + h * DR + ch * 8;
    if constexpr (!SLIM) g_off[u] = row * p.ld + h * DR + ch * 8;
    k_lds[u] = (ch >> 3) * 4096 + kpanel_off(row, ch & 7);
    v_lds[u] = row * VLD + ch * 8;
  }
  const half_t* kbase = p.k + tok0 * p.ld;
  const half_t* vbase = p.v + tok0 * p.ld;
  // fragment read addresses (lane-constant): K row li of panel-half sp, V tr-read block of lane
  int ka_off[2];
#pragma unroll
  for (int sp = 0; sp < 2; ++sp) ka_off[sp] = kpanel_off(li, sp * 4 + g);
  const int va_off = (4 * g + (li >> 2)) * VLD + 4 * (li & 3);

  const int nkt = (p.N + 63) / 64;
  h8 rk0[PREFETCH ? NL : 1], rv0[PREFETCH ? NL : 1], rk1[PREFETCH ? NL : 1], rv1[PREFETCH ? NL : 1];
  (void)rk1; (void)rv1;
  auto tile_direct = [&](int kt) {   // large head dims: no register staging across the compute phase
    const size_t toff = (size_t)kt * 64 * p.ld;
    const int kmax = p.N - kt * 64;
#pragma unroll 4
    for (int u = 0; u < NL; ++u) {
      if (t_row[u] < 64) {
        const bool ok = t_row[u] < kmax;
        *reinterpret_cast<h8*>(Ks + k_lds[u]) = ok ? *reinterpret_cast<const h8*>(kbase + toff + (SLIM ? g_off_c[u] : g_off[u])) : zero8;
        *reinterpret_cast<h8*>(Vs + v_lds[u]) = ok ? *reinterpret_cast<const h8*>(vbase + toff + (SLIM ? g_off_c[u] : g_off[u])) : zero8;
      }
    }
  };
  auto tile_load = [&](int kt, h8 (&rk)[PREFETCH ? NL : 1], h8 (&rv)[PREFETCH ? NL : 1]) {
    const size_t toff = (size_t)kt * 64 * p.ld;
    const int kmax = p.N - kt * 64;                  // rows < kmax are real keys
    if (kmax >= 64) {                                // full tile (wave-uniform): no per-register selects — the address
#pragma unroll                                       // of a thread without a row is clamped, its store is skipped
      for (int u = 0; u < (PREFETCH ? NL : 1); ++u) {
        rk[u] = *reinterpret_cast<const h8*>(kbase + toff + g_off_c[u]);
        rv[u] = *reinterpret_cast<const h8*>(vbase + toff + g_off_c[u]);
      }
    } else {
#pragma unroll
      for (int u = 0; u < (PREFETCH ? NL : 1); ++u) {
        const bool ok = t_row[u] < kmax;
        rk[u] = ok ? *reinterpret_cast<const h8*>(kbase + toff + (SLIM ? g_off_c[u] : g_off[u])) : zero8;
        rv[u] = ok ? *reinterpret_cast<const h8*>(vbase + toff + (SLIM ? g_off_c[u] : g_off[u])) : zero8;
      }
    }
  };
  auto tile_store = [&](int buf, const h8 (&rk)[PREFETCH ? NL : 1], const h8 (&rv)[PREFETCH ? NL : 1]) {
    half_t* kd = Ks + buf * TILE_HALFS;
    half_t* vd = Vs + buf * TILE_HALFS;
#pragma unroll
    for (int u = 0; u < (PREFETCH ? NL : 1); ++u) {
      if (t_row[u] < 64) {
        *reinterpret_cast<h8*>(kd + k_lds[u]) = rk[u];
        *reinterpret_cast<h8*>(vd + v_lds[u]) = rv[u];
      }
    }
  };

  const bool ragged = (p.N & 63) != 0;
#ifdef DADD_FLASH_STAMPS
  unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = __builtin_amdgcn_s_memtime();
#endif
  auto compute = [&](int kt, int buf) {
    const half_t* Kc = Ks + buf * TILE_HALFS;
    const half_t* Vc = Vs + buf * TILE_HALFS;
    // ---- S^T = K Q^T : sacc[kf][f], rows = keys kf*16 + 4g + r, column = query li.  The first K step takes a literal
    // zero as its C operand (an inline constant of the MFMA): no per-tile zeroing of the 16 * QF score registers — this
    // kernel is bound by VALU issue, and those moves were 64 of its ~560 vector instructions per tile.
    f4 sacc[4][QF];
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const half_t* kp = Kc + (s >> 1) * 4096 + ka_off[s & 1];
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) {
        const h8 ka = *reinterpret_cast<const h8*>(kp + kf * 1024);
#pragma unroll
        for (int f = 0; f < QF; ++f)
          sacc[kf][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qf[f][s], s == 0 ? negm[f] : sacc[kf][f], 0, 0, 0);
      }
    }

    FSTAMP(2)   // K fragment reads + S MFMAs issued
    // ---- online softmax per query column
    h8 pb[QF][2];
    if (ragged && kt == nkt - 1) {
      const int kbase_i = kt * 64 + 4 * g;
#pragma unroll
      for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kbase_i + kf * 16 + r >= p.N) sacc[kf][f][r] = -3.0e38f;
    }
    // One basic block for the whole tile (measured with the DADD_FLASH_STAMPS build, profiles/r03_w_flash_stamps.txt: the
    // per-fragment "did a maximum move" branches and the eight-deep maximum chains left a wave 2,280 ticks in this phase for
    // ~250 vector instructions).  Now: the 16 exponents of a query column reduce in a depth-3 tree of three-input maxima, the
    // QF fragments are independent instruction streams for the scheduler, and the reference maximum is only MOVED when a
    // column's exponents exceed 8 (the probabilities then stay <= 256: exact in fp16 / fp32 sums, the final division by the row
    // sum cancels the stale reference) or in the first tile - after it the wave-uniform re-centring branch is rarely taken.
    float mc[QF];
#pragma unroll
    for (int f = 0; f < QF; ++f) {
      const float t0 = vmax3(sacc[0][f][0], sacc[0][f][1], sacc[0][f][2]);
      const float t1 = vmax3(sacc[0][f][3], sacc[1][f][0], sacc[1][f][1]);
      const float t2 = vmax3(sacc[1][f][2], sacc[1][f][3], sacc[2][f][0]);
      const float t3 = vmax3(sacc[2][f][1], sacc[2][f][2], sacc[2][f][3]);
      const float t4 = vmax3(sacc[3][f][0], sacc[3][f][1], sacc[3][f][2]);
      mc[f] = vmax2(vmax3(t0, t1, t2), vmax3(t3, t4, sacc[3][f][3]));
    }
    // A column is spread over four lanes (16 keys each), but whether ANY exponent of the wave exceeds 8 needs no cross-lane
    // maximum: the lane-local maxima decide the (rare) branch, and only inside it are the columns' maxima united
    // (v_permlane16/32_swap cost 18 ticks each with their hazard nops, scripts/micro/inst_cost.hip: eight per tile before).
    float hot = mc[0];
#pragma unroll
    for (int f = 1; f < QF; ++f) hot = vmax2(hot, mc[f]);
    if (__any(kt == 0 || hot > 8.0f)) {                            // wave-uniform branch
#pragma unroll
      for (int f = 0; f < QF; ++f) {   // the maximum over the four lanes that share a query column: permlane swaps (VALU)
        float ua, ub;
        dadd_pair16(mc[f], ua, ub);
        mc[f] = vmax2(ua, ub);
      }
#pragma unroll
      for (int f = 0; f < QF; ++f) {
        float ua, ub;
        dadd_pair32(mc[f], ua, ub);
        mc[f] = vmax2(ua, ub);
      }
#pragma unroll
      for (int f = 0; f < QF; ++f) {
        const float dlt = (kt == 0 || mc[f] > 8.0f) ? mc[f] : 0.f;    // this column's reference moves by dlt (0: it stays)
        const float alpha = __builtin_amdgcn_exp2f(-dlt);
        negm[f] -= dlt;
        if (!SUMCOL) lrow[f] *= alpha;
#pragma unroll
        for (int df = 0; df < DF; ++df) oacc[df][f] *= alpha;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf) sacc[kf][f] -= dlt;
      }
    }
#pragma unroll
    for (int f = 0; f < QF; ++f) {
      float rs = 0.f;
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(sacc[kf][f][r]);
          sacc[kf][f][r] = pv;
          if (!SUMCOL) rs += pv;
        }
      if (!SUMCOL) {
        rs = dadd_sum_x16x32(rs);
        lrow[f] += rs;
      }
      pb[f][0] = pack_p8(sacc[0][f], sacc[1][f]);
      pb[f][1] = pack_p8(sacc[2][f], sacc[3][f]);
    }

    FSTAMP(3)   // softmax (waits for the S MFMAs)
    // ---- O^T += V^T P^T   (with SUMCOL, row DR of O^T accumulates sum_k P = the softmax denominator)
#pragma unroll
    for (int df = 0; df < DF; ++df) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const half_t* base = Vc + va_off + kb * 32 * VLD + df * 16;
        const h8 va = tr_pair(base, base + 16 * VLD);
#pragma unroll
        for (int f = 0; f < QF; ++f)
          oacc[df][f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, pb[f][kb], oacc[df][f], 0, 0, 0);
      }
    }
    FSTAMP(4)   // V fragment reads + PV MFMAs issued
  };

  if (DEEP) {
    // two tiles in flight: registers set 0 carries even tiles, set 1 odd tiles; LDS buffer = parity.
    // A load issued at iteration kt is stored at iteration kt+1 (after that tile's compute), i.e. it
    // has a whole compute phase plus a barrier to land.
    tile_load(0, rk0, rv0);
    __syncthreads();   // ones column / zero fill done
    tile_store(0, rk0, rv0);
    if (nkt > 1) tile_load(1, rk1, rv1);
    int kt = 0;
    while (kt < nkt) {
      __syncthreads();   // tile kt visible in buffer 0; everyone finished reading tile kt-1 (buffer 1)
      FSTAMP(0)   // barrier
      if (kt + 2 < nkt) tile_load(kt + 2, rk0, rv0);
      FSTAMP(1)   // global loads issued
      compute(kt, 0);
      if (kt + 1 < nkt) tile_store(1, rk1, rv1);
      FSTAMP(5)   // LDS stores of the next tile (wait for its global loads)
      if (++kt >= nkt) break;
      __syncthreads();
      FSTAMP(0)
      if (kt + 2 < nkt) tile_load(kt + 2, rk1, rv1);
      FSTAMP(1)
      compute(kt, 1);
      if (kt + 1 < nkt) tile_store(0, rk0, rv0);
      FSTAMP(5)
      ++kt;
    }
#ifdef DADD_FLASH_STAMPS
    if (p.dbg != nullptr && lane == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) p.dbg[((size_t)blockIdx.x * NW + wave) * 8 + i] = stamp_acc[i];
    }
#endif
  } else if (PREFETCH) {
    tile_load(0, rk0, rv0);
    __syncthreads();
    tile_store(0, rk0, rv0);
    for (int kt = 0; kt < nkt; ++kt) {
      __syncthreads();   // tile kt visible; everyone finished reading tile kt-1 (the other buffer)
      if (kt + 1 < nkt) tile_load(kt + 1, rk0, rv0);
      compute(kt, kt & 1);
      if (kt + 1 < nkt) tile_store((kt + 1) & 1, rk0, rv0);
    }
  } else {
    for (int kt = 0; kt < nkt; ++kt) {
      __syncthreads();
      tile_direct(kt);
      __syncthreads();
      compute(kt, 0);
    }
  }

  // ---- normalise and store: lane owns O[q = li][d = df*16 + 4g .. +3]
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    float l = lrow[f];
    if (SUMCOL) {   // row DR of O^T: fragment DR/16, lane group (DR%16)/4, register DR%4
      l = __shfl(oacc[DR / 16][f][DR % 4], ((DR % 16) / 4) * 16 + li, 64);
    }
    const int qrow = qw0 + f * 16 + li;
    if (qrow >= p.Nq) continue;
    const float inv = 1.0f / l;
    half_t* op = p.out + (qtok0 + qrow) * p.ldo + h * DR;
#pragma unroll
    for (int df = 0; df < DF; ++df) {
      const int dcol = df * 16 + 4 * g;
      if (dcol < DR) {
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)(oacc[df][f][r] * inv);
        *reinterpret_cast<h4*>(op + dcol) = o;
      }
    }
  }
}

// --------------------------------------------------------------------------------------------
struct XattnArgs {
  const half_t* q;
  const half_t* kv;
  half_t* out;
  const float* gates;
  const float* lambda_dev;   // lambda read from device memory when non-null (one captured graph serves a lambda sweep)
  float lambda;
  int B, N, H, T, ldkv, C;
  int qpb;           // queries per workgroup: 256 (four 16-query fragments per wave) or 64 (one) on the small maps
  float scale_log2;
};

// NF sixteen-key fragments; JOINT = one softmax over all of them (baseline) instead of one each.
template <int DR, int NF, bool JOINT>
__global__ __launch_bounds__(256) void xattn_kernel(const XattnArgs p) {
  constexpr int D = round_up(DR, 32), DVP = round_up(DR, 16);
  constexpr int KS = D / 32, DF = DVP / 16, DC = DR / 8;
  constexpr int VLD = v_stride(DVP);
  constexpr int KB = (NF + 1) / 2;  // 32-key steps of the PV product
  __shared__ __attribute__((aligned(16))) half_t Vs[64 * VLD];

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, g = lane >> 4, li = lane & 15;
  const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // lambda: by value, or from device memory (then NF = 3 is launched and lambda == 0 skips the delta pathway here,
  // exactly as attention_processor_routing_gates.py:160,177-178 does: its tokens are never read, NaN cannot leak)
  const float lam = p.lambda_dev ? *p.lambda_dev : p.lambda;
  const bool use_delta = JOINT || NF < 3 || lam != 0.f;      // wave-uniform
  // fragment descriptors (wave-uniform)
  int tb[NF], kcol[NF], vcol[NF];
  float wgt[NF];
  if (JOINT) {
#pragma unroll
    for (int f = 0; f < NF; ++f) { tb[f] = 16 * f; kcol[f] = 0; vcol[f] = p.C; wgt[f] = 1.f; }
  } else {
    tb[0] = 16; kcol[0] = 0;       vcol[0] = p.C;     wgt[0] = p.gates[0];  // anatomy
    tb[1] = 0;  kcol[1] = 2 * p.C; vcol[1] = 3 * p.C; wgt[1] = p.gates[1];  // disease
    if (NF > 2) { tb[NF - 1] = 32; kcol[NF - 1] = 2 * p.C; vcol[NF - 1] = 3 * p.C; wgt[NF - 1] = lam; }
  }

  // V rows -> LDS (row f*16 + key), rest zero so that empty key slots contribute exactly 0
  for (int i = t; i < 64 * VLD / 8; i += 256) *reinterpret_cast<h8*>(Vs + i * 8) = zero8;
  __syncthreads();
  const int nf_live = use_delta ? NF : NF - 1;
  for (int idx = t; idx < nf_live * 16 * DC; idx += 256) {
    const int row = idx / DC, ch = idx - row * DC;
    const int f = row >> 4, key = row & 15;
    const size_t off = ((size_t)b * p.T + tb[f] + key) * p.ldkv + vcol[f] + h * DR + ch * 8;
    *reinterpret_cast<h8*>(Vs + row * VLD + ch * 8) = *reinterpret_cast<const h8*>(p.kv + off);
  }
  __syncthreads();

  // resident operands: K fragments (A of S^T) straight from HBM, V^T fragments via tr reads
  h8 kA[NF][KS];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const half_t* kp = p.kv + ((size_t)b * p.T + tb[f] + li) * p.ldkv + kcol[f] + h * DR;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int dc = 32 * s + 8 * g;
      kA[f][s] = (dc < DR && f < nf_live) ? *reinterpret_cast<const h8*>(kp + dc) : zero8;
    }
  }
  h8 vA[DF][KB];
#pragma unroll
  for (int df = 0; df < DF; ++df)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const half_t* base = Vs + (kb * 32 + 4 * g + (li >> 2)) * VLD + df * 16 + 4 * (li & 3);
      vA[df][kb] = tr_pair(base, base + 16 * VLD);
    }

  const size_t tok0 = (size_t)b * p.N;
  const int q0 = blockIdx.x * p.qpb;
  for (int qfi = wave; qfi < (p.qpb >> 4); qfi += 4) {
    const int qbase = q0 + qfi * 16;
    if (qbase >= p.N) break;  // wave-uniform
    int qrow = qbase + li;
    const bool qok = qrow < p.N;
    if (!qok) qrow = p.N - 1;
    const half_t* qp = p.q + (tok0 + qrow) * p.C + h * DR;
    h8 qB[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int dc = 32 * s + 8 * g;
      qB[s] = (dc < DR) ? *reinterpret_cast<const h8*>(qp + dc) : zero8;
    }
    f4 sacc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      sacc[f] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s)
        sacc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kA[f][s], qB[s], sacc[f], 0, 0, 0);
    }
    // softmax over the 16 keys of each fragment (or over all NF*16 when JOINT)
    float mx[NF], sm[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      float m = -1e30f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sacc[f][r] *= p.scale_log2;
        m = fmaxf(m, sacc[f][r]);
      }
      mx[f] = dadd_max_x16x32(m);
    }
    if (JOINT) {
      float m = mx[0];
#pragma unroll
      for (int f = 1; f < NF; ++f) m = fmaxf(m, mx[f]);
#pragma unroll
      for (int f = 0; f < NF; ++f) mx[f] = m;
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sacc[f][r] = exp2f(sacc[f][r] - mx[f]);
        s += sacc[f][r];
      }
      sm[f] = dadd_sum_x16x32(s);
    }
    if (JOINT) {
      float s = 0.f;
#pragma unroll
      for (int f = 0; f < NF; ++f) s += sm[f];
#pragma unroll
      for (int f = 0; f < NF; ++f) sm[f] = s;
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const float w = (f < nf_live) ? wgt[f] / sm[f] : 0.f;    // skipped pathway: P = 0 against V rows that stayed 0
#pragma unroll
      for (int r = 0; r < 4; ++r) sacc[f][r] *= w;
    }
    h8 pb[KB];
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
      pb[kb] = pack_p(sacc[2 * kb], (2 * kb + 1 < NF) ? sacc[(2 * kb + 1 < NF) ? 2 * kb + 1 : 0] : zero4);

    half_t* op = p.out + (tok0 + qrow) * p.C + h * DR;
#pragma unroll
    for (int df = 0; df < DF; ++df) {
      f4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vA[df][kb], pb[kb], o, 0, 0, 0);
      const int dcol = df * 16 + 4 * g;
      if (qok && dcol < DR) {
        h4 ov;
#pragma unroll
        for (int r = 0; r < 4; ++r) ov[r] = (half_t)o[r];
        *reinterpret_cast<h4*>(op + dcol) = ov;
      }
    }
  }
}

template <int DR, int QF, bool PF, int NW = 4>
int flash_attr() {
  constexpr int D = round_up(DR, 32), DVP = round_up(DR, 16);
  constexpr int smem = (PF ? 2 : 1) * (((D + 63) / 64) * 4096 + 64 * v_stride(DVP)) * (int)sizeof(half_t);
  DADD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_kernel<DR, QF, PF, NW>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  return DADD_OK;
}

template <int DR, int QF, bool PF, int NW = 4>
int launch_flash(const FlashArgs& a, hipStream_t s) {
  constexpr int D = round_up(DR, 32), DVP = round_up(DR, 16);
  constexpr int smem = (PF ? 2 : 1) * (((D + 63) / 64) * 4096 + 64 * v_stride(DVP)) * (int)sizeof(half_t);
  constexpr int QB = NW * 16 * QF;
  dim3 grid(((a.Nq + QB - 1) / QB) * a.B * a.H);
  static const std::string name = "flash_kernel<" + std::to_string(DR) + ", " + std::to_string(QF) + ", " + (PF ? "true" : "false") +
                                  (NW == 4 ? "" : ", " + std::to_string(NW)) + ">";
  const double tok = (double)a.B * a.Nq, c = (double)a.H * DR;
  dadd_launch({name.c_str(), 4.0 * tok * a.N * c, (tok + (double)a.B * a.N) * c * 2.0 * 2.0}, flash_kernel<DR, QF, PF, NW>, grid, dim3(NW * 64), smem, s, a);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

template <int DR>
int launch_xattn(XattnArgs a, int mode, hipStream_t s) {
  // 256 queries per workgroup leave the small maps on a fraction of the CUs (16x16 at B = 4: 32 workgroups); with 64 every
  // wave handles one 16-query fragment and the grid is four times as large
  a.qpb = ((long)a.B * a.H * ((a.N + 255) / 256) < 512) ? 64 : 256;
  dim3 grid((a.N + a.qpb - 1) / a.qpb, a.B * a.H);
  static const std::string nm = "xattn_kernel<" + std::to_string(DR);
  static const std::string n2t = nm + ", 2, true>", n3f = nm + ", 3, false>", n2f = nm + ", 2, false>";
  const double tok = (double)a.B * a.N, c = (double)a.C;
  const double bytes = tok * c * 4.0;
  if (mode == DADD_XATTN_BASELINE)
    dadd_launch({n2t.c_str(), 4.0 * tok * 32 * c, bytes}, xattn_kernel<DR, 2, true>, grid, dim3(256), 0, s, a);
  else if (a.lambda_dev != nullptr || a.lambda != 0.0f)
    dadd_launch({n3f.c_str(), 4.0 * tok * 48 * c, bytes}, xattn_kernel<DR, 3, false>, grid, dim3(256), 0, s, a);
  else
    dadd_launch({n2f.c_str(), 4.0 * tok * 32 * c, bytes}, xattn_kernel<DR, 2, false>, grid, dim3(256), 0, s, a);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

}  // namespace

int dadd_init_attention() {
  int rc = flash_attr<40, 2, true>();
  if (rc == DADD_OK) rc = flash_attr<40, 4, true>();
  if (rc == DADD_OK) rc = flash_attr<40, 2, true, 8>();
  if (rc == DADD_OK) rc = flash_attr<64, 2, true>();
  if (rc == DADD_OK) rc = flash_attr<96, 1, true>();
  if (rc == DADD_OK) rc = flash_attr<80, 2, true>();
  if (rc == DADD_OK) rc = flash_attr<160, 2, true>();
  if (rc == DADD_OK) rc = flash_attr<160, 1, true>();
  if (rc == DADD_OK) rc = flash_attr<512, 1, false>();
  return rc;
}

#ifdef DADD_FLASH_STAMPS
static unsigned long long* g_flash_dbg = nullptr;
extern "C" int dadd_attn_debug(void* buf) {     // diagnostics build only: 8 x u64 per (block, wave) of the next d = 40 launches
  g_flash_dbg = static_cast<unsigned long long*>(buf);
  return DADD_OK;
}
#endif

extern "C" int dadd_attn_f16(const void* q, const void* k, const void* v, void* out, int B, int Nq, int Nk,
                             int heads, int d, int ld_q, int ld_kv, int ld_out, void* stream) {
  DADD_REQUIRE(q && k && v && out, "attn: null pointer");
  DADD_REQUIRE(B > 0 && Nq > 0 && Nk > 0 && heads > 0, "attn: empty problem");
  DADD_REQUIRE(ld_q % 8 == 0 && ld_kv % 8 == 0 && ld_out % 4 == 0 && ld_q >= heads * d && ld_kv >= heads * d &&
                   ld_out >= heads * d, "attn: bad leading dimensions");
  DADD_REQUIRE(dadd_aligned16(q) && dadd_aligned16(k) && dadd_aligned16(v) && dadd_aligned16(out),
               "attn: pointers must be 16-byte aligned");
  FlashArgs a;
  a.q = static_cast<const half_t*>(q);
  a.k = static_cast<const half_t*>(k);
  a.v = static_cast<const half_t*>(v);
  a.out = static_cast<half_t*>(out);
  a.B = B; a.Nq = Nq; a.N = Nk; a.H = heads; a.ldq = ld_q; a.ld = ld_kv; a.ldo = ld_out;
  a.scale_log2 = 1.4426950408889634f / sqrtf((float)d);
#ifdef DADD_FLASH_STAMPS
  a.dbg = g_flash_dbg;
#endif
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (d) {
    case 40: {  // 256 queries per block once the grid still fills the chip (>= 2 blocks per CU): eight waves x 32 queries
      // (four waves per SIMD, 124 VGPRs) - 149 us against 158 for four waves x 64 queries at 4 x 4096 x 8 heads
      // (profiles/r03_x_flash_bench.txt; DADD_FLASH40=0 selects the latter for A/B runs)
      static const int var = getenv("DADD_FLASH40") ? atoi(getenv("DADD_FLASH40")) : 1;
      if ((long)B * heads * ((Nq + 255) / 256) >= 512) return var == 1 ? launch_flash<40, 2, true, 8>(a, s) : launch_flash<40, 4, true>(a, s);
      return launch_flash<40, 2, true>(a, s);
    }
    case 64: return launch_flash<64, 2, true>(a, s);     // CLIP ViT towers (257 tokens, 16 x 64)
    case 80: return launch_flash<80, 2, true>(a, s);     // (16 queries per wave measured slower at 32x32, B = 4: 29.7 against 27.2 us)
    case 96: return launch_flash<96, 1, true>(a, s);     // nn.MultiheadAttention(768, 8) of the resampler / purifier
    case 160:   // 16 queries per wave while 32 would leave most of the chip idle (16x16 maps at B = 4: 64 -> 128 workgroups,
                // 15.9 -> 10.7 us, profiles/r03_zi_flash_small.txt; two-wave workgroups measured 12.8)
      return ((long)B * heads * ((Nq + 127) / 128) < 256) ? launch_flash<160, 1, true>(a, s) : launch_flash<160, 2, true>(a, s);
    case 512: return launch_flash<512, 1, false>(a, s);
    default:
      dadd_set_error("attn: unsupported head dim %d (40, 64, 80, 96, 160, 512)", d);
      return DADD_EINVAL;
  }
}

extern "C" int dadd_self_attn_f16(const void* q, const void* k, const void* v, void* out, int B,
                                  int N, int heads, int d, int ld_qkv, int ld_out, void* stream) {
  return dadd_attn_f16(q, k, v, out, B, N, N, heads, d, ld_qkv, ld_qkv, ld_out, stream);
}

extern "C" int dadd_tri_xattn_f16(const void* q, const void* kv, void* out, const float* gates,
                                  float lambda, const float* lambda_dev, int mode, int B, int N, int heads, int d,
                                  int T, int ld_kv, void* stream) {
  DADD_REQUIRE(q && kv && out, "tri_xattn: null pointer");
  DADD_REQUIRE(mode == DADD_XATTN_SPLIT || mode == DADD_XATTN_BASELINE, "tri_xattn: bad mode");
  const int C = heads * d;
  if (mode == DADD_XATTN_SPLIT) {
    DADD_REQUIRE(gates != nullptr, "tri_xattn: split mode needs gates");
    DADD_REQUIRE(T == 48 && ld_kv >= 4 * C, "tri_xattn: split mode needs T=48, ld_kv>=4C");
  } else {
    DADD_REQUIRE(T == 32 && ld_kv >= 2 * C, "tri_xattn: baseline mode needs T=32, ld_kv>=2C");
  }
  DADD_REQUIRE(B > 0 && N > 0 && ld_kv % 8 == 0, "tri_xattn: bad extents");
  DADD_REQUIRE(dadd_aligned16(q) && dadd_aligned16(kv) && dadd_aligned16(out),
               "tri_xattn: pointers must be 16-byte aligned");
  XattnArgs a;
  a.q = static_cast<const half_t*>(q);
  a.kv = static_cast<const half_t*>(kv);
  a.out = static_cast<half_t*>(out);
  a.gates = gates;
  a.lambda = lambda;
  a.lambda_dev = (mode == DADD_XATTN_SPLIT) ? lambda_dev : nullptr;
  a.B = B; a.N = N; a.H = heads; a.T = T; a.ldkv = ld_kv; a.C = C;
  a.scale_log2 = 1.4426950408889634f / sqrtf((float)d);
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (d) {
    case 40: return launch_xattn<40>(a, mode, s);
    case 80: return launch_xattn<80>(a, mode, s);
    case 160: return launch_xattn<160>(a, mode, s);
    default:
      dadd_set_error("tri_xattn: unsupported head dim %d (40, 80, 160)", d);
      return DADD_EINVAL;
  }
}
