// DADD cross-attention (attn2) of one transformer block as ONE kernel.
//
// With 16 keys per pathway the projections fold into the step-invariant conditioning:
//     q K_p^T        = x (W_q K_p^T)        -> Mcat  [B][384][C],  row (h*3+p)*16+t, scale*log2(e) folded in
//     P_p V_p W_o^T  = P_p (V_p W_o^T)      -> VW    [B][C][384],  gate_p / lambda folded in
// (8 heads x 3 pathways x 16 tokens = 384), so per 128-token tile:
//     S = x Mcat^T (K = C)  ->  24 independent 16-wide softmaxes per token, in registers  ->
//     out = P VW^T (K = 384) + bias + residual.
// Replaces to_q GEMM + dadd_tri_xattn_f16 + to_out GEMM (three launches, four tensor round trips) of
// SplitInjectionAttentionProcessor.__call__ (src/models/attention_processor_routing_gates.py:118-190); the fold is
// exact algebra, the rounding points move (Mcat / VW are rounded to fp16 instead of q / the attention output).
//
// One workgroup (4 waves, one per SIMD) per 128-token tile:
//   phase 1: S[128 x 384] in registers (wave tile 64 x 192: 48 f4 accumulators), operands by LDS-DMA into a
//            2-stage ring (x tile 16 KB + Mcat tile 48 KB per 64-deep K tile);
//   softmax: a 16-column group is exactly one MFMA fragment column block: 4 registers x 4 lanes (xor 16, 32);
//   phase 2: P (fp16) parked in LDS (row stride 896 B: = 128 mod 256, so the (row>>1)&7 chunk swizzle stays
//            conflict free), VW tiles [160 x 64] streamed through a 2-stage ring, 64 x 80 wave tiles, epilogue
//            bias + residual, 8-byte stores.
#include "dadd_common.h"
#include "igemm_args.h"   // xcd_remap

namespace {

constexpr int BK = 64, NS = 384, BN2 = 160;
constexpr int B1_BYTES = NS * BK * 2;         // 48 KB
constexpr int P_STRIDE = 896;                 // bytes per P row (768 used)
constexpr int B2_BYTES = BN2 * BK * 2;        // 20 KB
// BM = tokens per workgroup: 128, or 64 when 128-token tiles would leave half of the CUs idle
template <int BM> constexpr int a_bytes() { return BM * BK * 2; }
template <int BM> constexpr int stage1() { return a_bytes<BM>() + B1_BYTES; }
template <int BM> constexpr int p_bytes() { return BM * P_STRIDE; }
// VW ring of phase 2: four stages (three tiles in flight) where they fit beside P (BM = 64: 56 + 80 KB), else two.  With two
// stages every one of the twelve 20 KB tiles paid its whole L2 -> LDS latency (one tile in flight under 20 MFMAs per wave).
template <int BM> constexpr int nst2() { return p_bytes<BM>() + 4 * B2_BYTES <= 160 * 1024 - 1024 ? 4 : 2; }
template <int BM> constexpr int smem_bytes() {
  return (p_bytes<BM>() + nst2<BM>() * B2_BYTES) > 2 * stage1<BM>() ? (p_bytes<BM>() + nst2<BM>() * B2_BYTES) : 2 * stage1<BM>();
}

typedef __attribute__((address_space(3))) void* lptr_t;
constexpr unsigned OOB = 0x80000000u;

struct Attn2Args {
  const half_t* x;         // [B*HW][C]   LayerNorm output
  const half_t* mcat;      // [B][384][C]
  const half_t* vw;        // [B][C][384]
  const float* bias;       // [C] or null
  const half_t* residual;  // [B*HW][C]
  half_t* out;             // [B*HW][C]
  float* ln_stats;         // null, or LayerNorm row partials of `out`: [C / 80][B*HW][2] (DADD_EPI_LNSTAT of igemm)
  // LayerNorm (norm2) folded into the score GEMM: x is the UN-normalised hidden state, mcat carries gamma,
  //   S = rstd_m (x mcat^T - mu_m c1) + d,   c1[b][n] = sum_c mcat[b][n][c],  d[b][n] = sum_c mcat0[b][n][c] beta_c
  const float* ln_in;      // null (no fold), or the row partials of x written by its producer: [ln_parts][B*HW][2]
  const float* ln_c1;      // [B][384]
  const float* ln_d;       // [B][384]
  float ln_eps;
  int ln_parts;
  int B, HW, C;
};

__device__ __forceinline__ int lds_off(int row, int chunk) {   // halfs, 128-byte rows
  return (row * 8 + (chunk ^ ((row >> 1) & 7))) * 8;
}

template <int BM>
__global__ __launch_bounds__(256, 1) void attn2_fused_kernel(const Attn2Args p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MI = BM / 32;                  // 16-row fragments per wave (wave tile BM/2 rows)
  constexpr int WMR = BM / 2;
  constexpr int NXP = BM / 32;                 // x DMA pieces per wave
  constexpr int A_BYTES = a_bytes<BM>(), STAGE1 = stage1<BM>(), P_BYTES = p_bytes<BM>();
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = tile * BM;
  const int b = m0 / p.HW;
  const int C = p.C;
  const int nk1 = C / BK;
  const int lrow = lane >> 3, lch = lane & 7;
  const int fq = lane >> 4, mc = lane & 15, g = fq;

  // ---------------------------------------------------------------- phase 1: S = x Mcat^T
  const __amdgpu_buffer_rsrc_t rsX =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((size_t)p.B * p.HW * C * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsM = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.mcat + (size_t)b * NS * C), 0, NS * C * 2, 0x00020000);
  unsigned xv[NXP], mv[12];     // per-lane byte offsets of this wave's DMA pieces (8 rows x 128 B each)
#pragma unroll
  for (int i = 0; i < NXP; ++i) {
    const int row = (i * 4 + wave) * 8 + lrow;
    xv[i] = (unsigned)(((size_t)(m0 + row) * C + (lch ^ ((row >> 1) & 7)) * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    const int row = (j * 4 + wave) * 8 + lrow;
    mv[j] = (unsigned)(((size_t)row * C + (lch ^ ((row >> 1) & 7)) * 8) * 2);
  }
  auto issue1 = [&](int kt, int stage) {
    char* sa = smem + stage * STAGE1 + wave * 1024;
    const unsigned ko = (unsigned)(kt * BK * 2);
#pragma unroll
    for (int i = 0; i < NXP; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lptr_t)(sa + i * 4096), 16, xv[i], ko, 0, 0);
#pragma unroll
    for (int j = 0; j < 12; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsM, (lptr_t)(sa + A_BYTES + j * 4096), 16, mv[j], ko, 0, 0);
  };

  f4 acc[12][MI];
#pragma unroll
  for (int j = 0; j < 12; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f4{0.f, 0.f, 0.f, 0.f};
  // folded LayerNorm: row mean / rstd of this lane's MI rows from the producer's partials (loaded under the first DMA)
  const bool fold = p.ln_in != nullptr;
  float lmu[MI], lrs[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) lmu[i] = 0.f, lrs[i] = 1.f;
  if (fold) {
    const dadd_f2* st = reinterpret_cast<const dadd_f2*>(p.ln_in);
    const size_t mtot = (size_t)p.B * p.HW;
    float sa[MI], sq[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) sa[i] = sq[i] = 0.f;
    for (int pp = 0; pp < p.ln_parts; ++pp)
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const dadd_f2 v = st[(size_t)pp * mtot + m0 + wm * WMR + i * 16 + mc];
        sa[i] += v[0];
        sq[i] += v[1];
      }
    const float inv = __builtin_amdgcn_rcpf((float)C);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const float mu = sa[i] * inv;
      lmu[i] = mu;
      lrs[i] = rsqrtf(fmaxf(sq[i] * inv - mu * mu, 0.f) + p.ln_eps);
    }
  }
  int fa[2], fb[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    fa[s] = lds_off(wm * WMR + mc, s * 4 + fq) * 2;
    fb[s] = A_BYTES + lds_off(wn * 192 + mc, s * 4 + fq) * 2;
  }
  issue1(0, 0);
  for (int kt = 0; kt < nk1; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // tile kt landed; everyone is done with the other stage
    if (kt + 1 < nk1) issue1(kt + 1, (kt + 1) & 1);
    const char* st = smem + (kt & 1) * STAGE1;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      h8 xa[MI], wb[12];
#pragma unroll
      for (int i = 0; i < MI; ++i) xa[i] = *reinterpret_cast<const h8*>(st + fa[s] + i * 2048);
#pragma unroll
      for (int j = 0; j < 12; ++j) wb[j] = *reinterpret_cast<const h8*>(st + fb[s] + j * 2048);
#pragma unroll
      for (int j = 0; j < 12; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[i], acc[j][i], 0, 0, 0);
    }
  }
  // c1 / d of this lane's 12 column quads (issued before the barrier and the first VW DMA: their latency hides there)
  f4 fc1[12], fd[12];
  if (fold) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int k = wn * 192 + j * 16 + g * 4;
      fc1[j] = *reinterpret_cast<const f4*>(p.ln_c1 + (size_t)b * NS + k);
      fd[j] = *reinterpret_cast<const f4*>(p.ln_d + (size_t)b * NS + k);
    }
  }
  __syncthreads();                                     // all fragment reads of phase 1 done: LDS is free

  // VW stream of phase 2 starts now (its ring lives behind the P region)
  const int nt2 = C / BN2;
  const int nk2 = NS / BK;                             // 6
  const int n_it2 = nt2 * nk2;
  const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.vw + (size_t)b * C * NS), 0, C * NS * 2, 0x00020000);
  unsigned vv[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int row = (j * 4 + wave) * 8 + lrow;         // row of the 160-row tile
    vv[j] = (unsigned)(((size_t)row * NS + (lch ^ ((row >> 1) & 7)) * 8) * 2);
  }
  auto issue2 = [&](int it, int stage) {
    const int nt = it / nk2, kt = it - nt * nk2;
    char* sa = smem + P_BYTES + stage * B2_BYTES + wave * 1024;
    const unsigned so = (unsigned)(((size_t)nt * BN2 * NS + kt * BK) * 2);
#pragma unroll
    for (int j = 0; j < 5; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, (lptr_t)(sa + j * 4096), 16, vv[j], so, 0, 0);
  };
  constexpr int NST2 = nst2<BM>();
  issue2(0, 0);
  if constexpr (NST2 == 4) {
    if (1 < n_it2) issue2(1, 1);
    if (2 < n_it2) issue2(2, 2);
  }

  // ---------------------------------------------------------------- softmax per 16-column group, P -> LDS (fp16)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = wm * WMR + i * 16 + mc;
    char* prow = smem + row * P_STRIDE;
    const int swz = (row >> 1) & 7;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      f4 v = acc[j][i];
      if (fold) v = lrs[i] * (v - lmu[i] * fc1[j]) + fd[j];
      float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
      mx = dadd_max_x16x32(mx);
      const float e0 = __builtin_amdgcn_exp2f(v[0] - mx), e1 = __builtin_amdgcn_exp2f(v[1] - mx);
      const float e2 = __builtin_amdgcn_exp2f(v[2] - mx), e3 = __builtin_amdgcn_exp2f(v[3] - mx);
      float sum = (e0 + e1) + (e2 + e3);
      sum = dadd_sum_x16x32(sum);
      const float inv = __builtin_amdgcn_rcpf(sum);
      const h4 o = {(half_t)(e0 * inv), (half_t)(e1 * inv), (half_t)(e2 * inv), (half_t)(e3 * inv)};
      const int k = wn * 192 + j * 16 + g * 4;         // column of P = K index of phase 2
      const int seg = k >> 6, ch = (k & 63) >> 3;
      *reinterpret_cast<h4*>(prow + seg * 128 + ((ch ^ swz) << 4) + (k & 7) * 2) = o;
    }
  }

  // ---------------------------------------------------------------- phase 2: out = P VW^T + bias + residual
  const int fp0 = (wm * WMR + mc) * P_STRIDE;           // P fragment row base (fragment i adds i*16 rows)
  const int pswz = ((wm * WMR + mc) >> 1) & 7;           // (row + 16 i) >> 1 & 7 == (row >> 1) & 7
  const int swb = wn * 80 + mc;
  const int fv0 = (swb * 8 + (fq ^ ((swb >> 1) & 7))) * 16;
  f4 acc2[5][MI];
  for (int it = 0; it < n_it2; ++it) {
    const int nt = it / nk2, kt = it - nt * nk2;
    if (kt == 0) {
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i) acc2[j][i] = f4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (NST2 == 4) {      // tiles it + 1 and it + 2 (five DMA instructions each, issued after tile `it`) may stay in flight
      const int younger = min(2, n_it2 - 1 - it);
      if (younger == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                   // VW tile `it` landed (first time: P writes visible too)
    if (it + NST2 - 1 < n_it2) issue2(it + NST2 - 1, (it + NST2 - 1) & (NST2 - 1));
    const char* vt = smem + P_BYTES + (it & (NST2 - 1)) * B2_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      h8 pa[MI], wb[5];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        pa[i] = *reinterpret_cast<const h8*>(smem + fp0 + i * 16 * P_STRIDE + kt * 128 + (((s * 4 + fq) ^ pswz) << 4));
#pragma unroll
      for (int j = 0; j < 5; ++j) wb[j] = *reinterpret_cast<const h8*>(vt + (fv0 ^ (s * 64)) + j * 2048);
#pragma unroll
      for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
          acc2[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], pa[i], acc2[j][i], 0, 0, 0);
    }
    if (kt == nk2 - 1) {                               // epilogue of column tile nt
      // All bias / residual loads of the tile first, then the arithmetic and the stores: written load - use - store per
      // (i, j) the compiler kept that order (the stores may alias the next loads) and the wave paid 2 x 5 x MI dependent
      // round trips behind s_waitcnt vmcnt(0) - about half of this kernel's time.
      f4 bias4[5];
      h4 res4[5][MI];
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int n = nt * BN2 + wn * 80 + j * 16 + g * 4;
        bias4[j] = p.bias ? *reinterpret_cast<const f4*>(p.bias + n) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MI; ++i)
          res4[j][i] = *reinterpret_cast<const h4*>(p.residual + ((size_t)m0 + wm * WMR + i * 16 + mc) * C + n);
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const size_t m = (size_t)m0 + wm * WMR + i * 16 + mc;
        float rs1 = 0.f, rs2 = 0.f;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const int n = nt * BN2 + wn * 80 + j * 16 + g * 4;
          const f4 v = acc2[j][i] + bias4[j];
          const h4 rv = res4[j][i];
          const h4 o = {(half_t)(v[0] + (float)rv[0]), (half_t)(v[1] + (float)rv[1]),
                        (half_t)(v[2] + (float)rv[2]), (half_t)(v[3] + (float)rv[3])};
          *reinterpret_cast<h4*>(p.out + m * C + n) = o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float f = (float)o[r];
            rs1 += f;
            rs2 = fmaf(f, f, rs2);
          }
        }
        if (p.ln_stats) {     // the next LayerNorm's row partials (sum, sum of squares over this wave's 80 columns)
          rs1 = dadd_sum_x16x32(rs1);
          rs2 = dadd_sum_x16x32(rs2);
          if (g == 0)
            reinterpret_cast<dadd_f2*>(p.ln_stats)[(size_t)(nt * 2 + wn) * ((size_t)p.B * p.HW) + m] = dadd_f2{rs1, rs2};
        }
      }
    }
  }
#endif
}

}  // namespace

int g_attn2_cus = 0;

int dadd_init_attn2_fused() {
  DADD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_fused_kernel<128>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes<128>()));
  DADD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn2_fused_kernel<64>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes<64>()));
  int dev = 0;
  DADD_HIP(hipGetDevice(&dev));
  DADD_HIP(hipDeviceGetAttribute(&g_attn2_cus, hipDeviceAttributeMultiprocessorCount, dev));
  return DADD_OK;
}

extern "C" int dadd_attn2_fused_f16(const void* x, const void* mcat, const void* vw, const float* bias,
                                    const void* residual, void* out, float* ln_stats_out, const float* ln_stats_in,
                                    int ln_parts_in, const float* ln_c1, const float* ln_d, float ln_eps, int B, int HW,
                                    int C, void* stream) {
  DADD_REQUIRE(x && mcat && vw && residual && out, "attn2_fused: null pointer");
  DADD_REQUIRE(B > 0 && HW > 0 && HW % 128 == 0, "attn2_fused: H*W=%d must be a multiple of 128", HW);
  DADD_REQUIRE(C > 0 && C % BN2 == 0 && C % BK == 0, "attn2_fused: C=%d must be a multiple of 320", C);
  DADD_REQUIRE((size_t)B * HW * C * 2 < 0x7FF00000ull, "attn2_fused: activation larger than the 2 GiB buffer window");
  DADD_REQUIRE(dadd_aligned16(x) && dadd_aligned16(mcat) && dadd_aligned16(vw) && dadd_aligned16(residual) &&
                   dadd_aligned16(out) && (!bias || dadd_aligned16(bias)),
               "attn2_fused: pointers must be 16-byte aligned");
  Attn2Args a;
  a.x = static_cast<const half_t*>(x);
  a.mcat = static_cast<const half_t*>(mcat);
  a.vw = static_cast<const half_t*>(vw);
  a.bias = bias;
  a.residual = static_cast<const half_t*>(residual);
  a.out = static_cast<half_t*>(out);
  a.ln_stats = ln_stats_out;
  DADD_REQUIRE(ln_stats_in == nullptr || (ln_parts_in > 0 && ln_c1 && ln_d && ln_eps > 0.f),
               "attn2_fused: a folded LayerNorm needs the row partials of x, their count, c1, d and eps");
  a.ln_in = ln_stats_in;
  a.ln_parts = ln_parts_in;
  a.ln_c1 = ln_c1;
  a.ln_d = ln_d;
  a.ln_eps = ln_eps;
  a.B = B; a.HW = HW; a.C = C;
  // algorithmic work: two GEMMs against the folded conditioning (K = C, N = 384 and K = 384, N = C)
  const double flop = 4.0 * (double)B * HW * C * 384.0;
  const double bytes = (double)B * HW * C * 2.0 * 3.0 + (double)B * 2.0 * 384.0 * C * 2.0;
  if (B * HW / 128 < g_attn2_cus)     // 128-token tiles would not fill the chip
    dadd_launch({"attn2_fused_kernel<64>", flop, bytes}, attn2_fused_kernel<64>, dim3(B * HW / 64), dim3(256),
                smem_bytes<64>(), static_cast<hipStream_t>(stream), a);
  else
    dadd_launch({"attn2_fused_kernel<128>", flop, bytes}, attn2_fused_kernel<128>, dim3(B * HW / 128), dim3(256),
                smem_bytes<128>(), static_cast<hipStream_t>(stream), a);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}
