// Implicit-GEMM convolution / linear on gfx950 MFMA (v_mfma_f32_16x16x32_f16).
//
//   out[m][n] = epi( sum_k A[m][k] * W[n][k] ),  m = output pixel (NHWC row), k = (tap, cin)
//
// Block = 256 threads = 4 waves (2 along M x 2 along N); tile 128 x BN x 64 with BN = 128 or 160
// (every UNet width is a multiple of 160, every VAE width a multiple of 128).  Both operands are
// K-contiguous in HBM (NHWC activations, [Cout][tap][Cin] weights), so both are staged with 16-byte
// loads into an XOR-swizzled LDS image and read back as ds_read_b128 MFMA fragments without bank
// conflicts.  The next K tile is fetched into registers while the current one feeds the MFMAs
// (double-buffered LDS, one barrier per K tile).  The MFMA is issued "swapped" (weights as the A
// operand) so each lane ends up with 4 consecutive output channels of one pixel: 8-byte fp16
// stores and vector reads of bias / time-embedding row / residual in the epilogue.
//
// Zero padding, stride 2, the asymmetric VAE-encoder padding, nearest-2x upsampling and the
// skip-connection concat are all folded into the A-tile address generation.
#include "dadd_common.h"
#include "igemm_args.h"
#include <cstdlib>
#include <string>
#include "igemm_epilogue.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 64;

// LDS image: rows of 64 halfs (8 chunks of 16 B); chunk index XORed with (row>>1)&7 so that the 16
// rows one MFMA fragment reads at a fixed chunk land on 16 distinct 16-byte slots of the bank row.
__device__ __forceinline__ int lds_off(int row, int chunk) {
  return (row * 8 + (chunk ^ ((row >> 1) & 7))) * 8;
}

template <int BM, int BN, bool DEEP>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const IgemmArgs p) {
  constexpr int WM = BM / 2;       // rows per wave (64 or 32)
  constexpr int MI = WM / 16;      // m-fragments per wave (4 or 2)
  constexpr int WN = BN / 2;       // columns per wave
  constexpr int J = WN / 16;       // n-fragments per wave (4 or 5)
  constexpr int NA = BM * 8 / 256; // 16-byte activation loads per thread per K tile
  constexpr int NB = BN * 8 / 256; // 16-byte weight loads per thread per K tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* As = reinterpret_cast<half_t*>(smem);
  half_t* Bs = As + 2 * BM * BK;

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_id = xcd_remap(blockIdx.x, gridDim.x);
  int nt, mt;
  tile_decode(p, tile_id, mt, nt);
  const int m0 = mt * BM, n0 = nt * BN;
  const int z = blockIdx.y;
  const int kt0 = z * p.kps;
  const int kt1 = min(p.nkt, kt0 + p.kps);
  const int Cin = p.C1 + p.C2;
  const int q = t & 7;
  const int r0 = t >> 3;
  const int Hv = p.ups ? 2 * p.Hi : p.Hi;
  const int Wv = p.ups ? 2 * p.Wi : p.Wi;
  const int HoWo = p.Ho * p.Wo;

  int a_pix[NA], a_y[NA], a_x[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int m = m0 + r0 + 32 * i;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int b = mm / HoWo;
    const int rem = mm - b * HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    a_pix[i] = b * p.Hi * p.Wi;
    a_y[i] = ok ? oy * p.stride - p.pad : -100000;  // invalid rows fail every bounds test
    a_x[i] = ox * p.stride - p.pad;
  }

  f4 acc[J][MI];
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int i = 0; i < MI; ++i) acc[j][i] = f4{0.f, 0.f, 0.f, 0.f};

  // folded LayerNorm with the producer's row partials: its operands are staged behind the two K-tile buffers (ln_lds.h)
  char* const ln_scr = smem + 2 * (BM + BN) * BK * 2;
  constexpr int LN_CAP = BM == 128 ? (BN == 160 ? 8192 : LN_LDS_BYTES) : 4096;   // two workgroups per CU stay resident
  bool ln_stage = false, ln_sts = false;
#if defined(__HIP_DEVICE_COMPILE__)
  ln_stage = ln_lds_usable(p, BM, 3);
  ln_sts = ln_stage && ln_lds_stats(p, LN_CAP);
  if (ln_stage) ln_lds_issue<BN>(p, ln_scr, m0, n0, __builtin_amdgcn_readfirstlane(wave), lane, ln_sts);
#endif

  const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  auto gload = [&](int kt, h8 (&ra)[NA], h8 (&rb)[NB]) {
    const int kk = kt * BK;
    const int tap = kk / Cin;
    const int c = kk - tap * Cin;
    const int ky = (p.taps == 9) ? tap / 3 : 0;
    const int kx = (p.taps == 9) ? tap - 3 * ky : 0;
    const bool second = c >= p.C1;
    const half_t* src = second ? p.x2 : p.x;
    const int cs = second ? p.C2 : p.C1;
    const int cc = (second ? c - p.C1 : c) + q * 8;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      int iy = a_y[i] + ky, ix = a_x[i] + kx;
      const bool ok = (iy >= 0) & (iy < Hv) & (ix >= 0) & (ix < Wv);
      if (p.ups) {
        iy >>= 1;
        ix >>= 1;
      }
      const size_t off = (size_t)(a_pix[i] + iy * p.Wi + ix) * cs + cc;
      ra[i] = ok ? *reinterpret_cast<const h8*>(src + off) : zero8;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = n0 + r0 + 32 * i;
      rb[i] = (n < p.N) ? *reinterpret_cast<const h8*>(p.w + (size_t)n * p.K + kk + q * 8) : zero8;
    }
  };
  auto lstore = [&](int buf, const h8 (&ra)[NA], const h8 (&rb)[NB]) {
    half_t* a = As + buf * BM * BK;
    half_t* b = Bs + buf * BN * BK;
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<h8*>(a + lds_off(r0 + 32 * i, q)) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<h8*>(b + lds_off(r0 + 32 * i, q)) = rb[i];
  };
  auto compute = [&](int buf) {
    const half_t* a = As + buf * BM * BK;
    const half_t* b = Bs + buf * BN * BK;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int chunk = s * 4 + (lane >> 4);
      h8 xa[MI], wb[J];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        xa[i] = *reinterpret_cast<const h8*>(a + lds_off(wm * WM + i * 16 + (lane & 15), chunk));
#pragma unroll
      for (int j = 0; j < J; ++j)
        wb[j] = *reinterpret_cast<const h8*>(b + lds_off(wn * WN + j * 16 + (lane & 15), chunk));
#pragma unroll
      for (int j = 0; j < J; ++j)
#pragma unroll
        for (int i = 0; i < MI; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[i], acc[j][i], 0, 0, 0);
    }
  };

  if constexpr (!DEEP) {
    // one K tile in flight: fetch kt+1 into registers while kt feeds the MFMAs
    h8 ra[NA], rb[NB];
    if (kt0 < kt1) {
      gload(kt0, ra, rb);
      lstore(0, ra, rb);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      const bool more = kt + 1 < kt1;
      if (more) gload(kt + 1, ra, rb);
      compute(cur);
      if (more) lstore(cur ^ 1, ra, rb);
      __syncthreads();
    }
  } else {
    // two K tiles in flight: while tile kt computes, tile kt+1 is already in registers (issued one
    // iteration ago) and tile kt+2 is being fetched; loop unrolled by two so both register sets are
    // statically named (runtime-indexed register arrays would go to scratch).
    h8 raA[NA], rbA[NB], raB[NA], rbB[NB];
    if (kt0 < kt1) {
      gload(kt0, raA, rbA);
      if (kt0 + 1 < kt1) gload(kt0 + 1, raB, rbB);
      lstore(0, raA, rbA);
    }
    __syncthreads();
    int kt = kt0;
    while (kt < kt1) {
      // even phase: LDS buf 0 holds kt, regs B hold kt+1, fetch kt+2 into regs A
      if (kt + 2 < kt1) gload(kt + 2, raA, rbA);
      compute(0);
      if (kt + 1 < kt1) lstore(1, raB, rbB);
      __syncthreads();
      if (++kt >= kt1) break;
      // odd phase: LDS buf 1 holds kt, regs A hold kt+1, fetch kt+2 into regs B
      if (kt + 2 < kt1) gload(kt + 2, raB, rbB);
      compute(1);
      if (kt + 1 < kt1) lstore(0, raA, rbA);
      __syncthreads();
      ++kt;
    }
  }

  // ---- epilogue (shared with igemm_dma.hip): lane holds out[m][n .. n+3] of the swapped MFMA result
  if (ln_stage) {     // (the DMA are the oldest vector-memory operations of every wave)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  igemm_epilogue<J, MI, WM, WN>(p, acc, m0, n0, wm, wn, lane, z, smem, nullptr, nullptr, ln_scr, ln_sts ? ln_scr + LN_LDS_STATS : nullptr,
                                ln_stage ? ln_scr : nullptr);
}

// Finishes a split-K launch: sums the fp32 slabs and applies the (non-GEGLU) epilogue.
__global__ __launch_bounds__(256) void splitk_finish_kernel(const IgemmArgs p, int nsplit) {
  const int n4 = p.N >> 2;
  const size_t total = (size_t)p.M * n4;
  const int HoWo = p.Ho * p.Wo;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / n4);
    const int n = (int)(idx - (size_t)m * n4) * 4;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    const float* slab = p.partial + (size_t)m * p.N + n;
    const size_t sstride = (size_t)p.M * p.N;
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {     // four slabs in flight, summed in slice order (bit-reproducible)
      const f4 t0 = *reinterpret_cast<const f4*>(slab + (size_t)s * sstride);
      const f4 t1 = *reinterpret_cast<const f4*>(slab + (size_t)(s + 1) * sstride);
      const f4 t2 = *reinterpret_cast<const f4*>(slab + (size_t)(s + 2) * sstride);
      const f4 t3 = *reinterpret_cast<const f4*>(slab + (size_t)(s + 3) * sstride);
      v += t0;
      v += t1;
      v += t2;
      v += t3;
    }
    for (; s < nsplit; ++s) v += *reinterpret_cast<const f4*>(slab + (size_t)s * sstride);
    if (p.flags & DADD_EPI_BIAS) v += *reinterpret_cast<const f4*>(p.bias + n);
    if (p.flags & DADD_EPI_ROWVEC)
      v += *reinterpret_cast<const f4*>(p.rowvec + (size_t)(m / HoWo) * p.ld_rowvec + n);
    if (p.flags & DADD_EPI_RESIDUAL) {
      const h4 rv = *reinterpret_cast<const h4*>(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
    }
    h4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = (half_t)v[r];
    *reinterpret_cast<h4*>(p.out + (size_t)m * p.ldo + n) = o;
  }
}

// Finish of a split-K launch whose output feeds a GroupNorm (DADD_EPI_GNSTAT): same sums and epilogue as above, but a
// block owns R rows x CB columns (whole groups) and also writes their (sum, sum of squares) per group into the
// GroupNorm's chunk partials [B][Ho*Wo/R][32][2] — the consumer then needs no statistics pass.  Fixed summation order.
// R = 16 (the engine's choice): 4x the blocks of the 64-row version on the 16x16 / 8x8 maps, where that one ran 32-128
// blocks of serial, dependent slab loads (16-21 us per launch for 23-31 MB); here every thread has its (up to three) rows
// times four slabs in flight.
template <int CB, int R>
__global__ __launch_bounds__(256) void splitk_finish_gn_kernel(const IgemmArgs p, int nsplit) {
  constexpr int Q = CB / 4, RL = 256 / Q;          // channel quads per row, row lanes (40 x 6 or 32 x 8)
  constexpr int NR = (R + RL - 1) / RL;            // rows per thread
  __shared__ float red[RL][CB][2];
  const int t = threadIdx.x, q = t % Q, rl = t / Q;
  const int n = blockIdx.y * CB + q * 4;
  const int m_base = blockIdx.x * R;
  const int HoWo = p.Ho * p.Wo;
  const int bsmp = m_base / HoWo;
  const size_t sstride = (size_t)p.M * p.N;
  float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
  if (rl < RL) {
    f4 add = {0.f, 0.f, 0.f, 0.f};
    if (p.flags & DADD_EPI_BIAS) add = *reinterpret_cast<const f4*>(p.bias + n);
    if (p.flags & DADD_EPI_ROWVEC) add += *reinterpret_cast<const f4*>(p.rowvec + (size_t)bsmp * p.ld_rowvec + n);
    f4 v[NR];
    const float* slab[NR];
    bool ok[NR];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int r = rl + j * RL;
      ok[j] = r < R;
      slab[j] = p.partial + (size_t)(m_base + (ok[j] ? r : 0)) * p.N + n;
      v[j] = add;
    }
    for (int s2 = 0; s2 < nsplit; s2 += 4) {        // NR rows x four slabs in flight, summed in slice order
      f4 tt[NR][4];
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) tt[j][u] = *reinterpret_cast<const f4*>(slab[j] + (size_t)min(s2 + u, nsplit - 1) * sstride);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float keep = (s2 + u < nsplit) ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[j][e] = fmaf(keep, tt[j][u][e], v[j][e]);
      }
    }
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      if (!ok[j]) continue;
      const int m = m_base + rl + j * RL;
      if (p.flags & DADD_EPI_RESIDUAL) {
        const h4 rv = *reinterpret_cast<const h4*>(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[j][e] += (float)rv[e];
      }
      h4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (half_t)v[j][e];
        const float f = (float)o[e];
        cs[e] += f;
        cq[e] = fmaf(f, f, cq[e]);
      }
      *reinterpret_cast<h4*>(p.out + (size_t)m * p.ldo + n) = o;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[rl][q * 4 + e][0] = cs[e];
      red[rl][q * 4 + e][1] = cq[e];
    }
  }
  __syncthreads();
  const int ngrp = CB / p.gn_cg;
  if (t < ngrp) {
    float a = 0.f, qq = 0.f;
    for (int c = 0; c < p.gn_cg; ++c)
      for (int r = 0; r < RL; ++r) {
        a += red[r][t * p.gn_cg + c][0];
        qq += red[r][t * p.gn_cg + c][1];
      }
    const int chunk = (m_base - bsmp * HoWo) / R;
    float* w = p.gn_ws + (((size_t)bsmp * p.gn_nchunk + chunk) * 32 + (blockIdx.y * CB) / p.gn_cg + t) * 2;
    w[0] = a;
    w[1] = qq;
  }
}

// Finish of a split-K launch on a small map whose output feeds a GroupNorm (DADD_EPI_GNAPPLY): one block per (group,
// sample) sums the slabs of its Ho*Wo x cg slab (the finish kernel's order: slices, bias, rowvec, residual; one rounding),
// writes it, keeps it in LDS, reduces (the single-launch GroupNorm's order: per-thread strided pairs, wave butterfly, four
// waves in double) and writes the normalised (+ SiLU) copy the next conv reads.  Replaces splitk_finish_kernel +
// gn_fused_kernel on the 8x8 maps (two launches of ~5 us each for 0.6 MB) with bit-identical results.
template <int VEC>
__global__ __launch_bounds__(512) void splitk_finish_gnapply_kernel(const IgemmArgs p, int nsplit) {
  typedef float fv __attribute__((ext_vector_type(VEC)));
  typedef _Float16 hv __attribute__((ext_vector_type(VEC)));
  extern __shared__ __attribute__((aligned(16))) char fg_smem[];
  half_t* slab = reinterpret_cast<half_t*>(fg_smem);       // [HW][cg]
  __shared__ float red[2][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int g = blockIdx.x, b = blockIdx.y;
  const int HW = p.Ho * p.Wo, cg = p.N >> 5, hp = cg >> 1, c0 = g * cg, npair = HW * hp;
  const int vp = cg / VEC, nitem = HW * vp;
  const size_t sstride = (size_t)p.M * p.N;
  // 1. the finish: every item (row, VEC channels) with eight slabs in flight, summed in slice order
  for (int idx = t; idx < nitem; idx += 512) {
    const int row = idx / vp, cv = idx - row * vp;
    const int c = c0 + VEC * cv;
    const size_t m = (size_t)b * HW + row;
    const float* sl = p.partial + m * p.N + c;
    fv v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = 0.f;
    for (int k0 = 0; k0 < nsplit; k0 += 8) {
      fv tt[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) tt[u] = *reinterpret_cast<const fv*>(sl + (size_t)min(k0 + u, nsplit - 1) * sstride);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float keep = (k0 + u < nsplit) ? 1.f : 0.f;      // fma(1, t, v) == v + t: the finish kernel's sums exactly
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[e] = fmaf(keep, tt[u][e], v[e]);
      }
    }
    if (p.flags & DADD_EPI_BIAS) v += *reinterpret_cast<const fv*>(p.bias + c);
    if (p.flags & DADD_EPI_ROWVEC) v += *reinterpret_cast<const fv*>(p.rowvec + (size_t)b * p.ld_rowvec + c);
    if (p.flags & DADD_EPI_RESIDUAL) {
      const hv rv = *reinterpret_cast<const hv*>(p.residual + m * p.ldr + c);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] += (float)rv[e];
    }
    hv o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = (half_t)v[e];
    *reinterpret_cast<hv*>(p.out + m * p.ldo + c) = o;
    *reinterpret_cast<hv*>(slab + row * cg + VEC * cv) = o;
  }
  __syncthreads();
  // 2. the statistics, in the single-launch GroupNorm's order: 256 strided lanes of channel pairs, butterfly, four waves
  float s = 0.f, ss = 0.f;
  if (t < 256) {
    for (int idx = t; idx < npair; idx += 256) {
      const h2 v = *reinterpret_cast<const h2*>(slab + 2 * idx);
      const float a = (float)v[0], d = (float)v[1];
      s += a + d;
      ss += a * a + d * d;
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; }
  }
  __syncthreads();
  const double n = (double)HW * (double)cg;
  const double sum = (double)red[0][0] + (double)red[0][1] + (double)red[0][2] + (double)red[0][3];
  const double sq = (double)red[1][0] + (double)red[1][1] + (double)red[1][2] + (double)red[1][3];
  const double mu = sum / n;
  double var = sq / n - mu * mu;
  if (var < 0.0) var = 0.0;
  const float mean = (float)mu, rstd = (float)(1.0 / sqrt(var + (double)p.gno_eps));
  const bool silu = (p.flags & DADD_EPI_GNAPPLY_SILU) != 0;
  // 3. normalise from LDS
  for (int idx = t; idx < npair; idx += 512) {
    const int row = idx / hp, cp = idx - row * hp;
    const int c = c0 + 2 * cp;
    const h2 v = *reinterpret_cast<const h2*>(slab + 2 * idx);
    h2 o;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float sc = rstd * p.gno_gamma[c + e];
      float f = ((float)v[e] - mean) * sc + p.gno_beta[c + e];
      if (silu) f = dadd_silu(f);
      o[e] = (half_t)f;
    }
    *reinterpret_cast<h2*>(p.gno_out + ((size_t)b * HW + row) * p.N + c) = o;
  }
}

constexpr double L2_KEEP = 3.0e6;   // bytes of one operand an XCD's 4 MB L2 can be trusted to keep while the other streams

template <int BM, int BN, bool DEEP>
int set_attr() {
  constexpr int smem = 2 * (BM + BN) * BK * (int)sizeof(half_t) + (BM == 128 ? (BN == 160 ? 8192 : LN_LDS_BYTES) : 4096);
  DADD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<BM, BN, DEEP>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  return DADD_OK;
}

template <int BM, int BN, bool DEEP>
int launch(const IgemmArgs& a, int nsplit, hipStream_t s) {
  constexpr int smem = 2 * (BM + BN) * BK * (int)sizeof(half_t) + (BM == 128 ? (BN == 160 ? 8192 : LN_LDS_BYTES) : 4096);   // + the epilogue's scratch
  const int mtiles = (a.M + BM - 1) / BM;
  dim3 grid(mtiles * a.ntiles, nsplit);
  static const std::string name = "igemm_kernel<" + std::to_string(BM) + ", " + std::to_string(BN) + ", " + (DEEP ? "true" : "false") + ">";
  dadd_launch({name.c_str(), dadd_igemm_flop(a), dadd_igemm_bytes(a)}, igemm_kernel<BM, BN, DEEP>, grid, dim3(256), smem, s, a);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

template <int BM, int BN>
int launch2(const IgemmArgs& a, int nsplit, bool deep, hipStream_t s) {
  return deep ? launch<BM, BN, true>(a, nsplit, s) : launch<BM, BN, false>(a, nsplit, s);
}

}  // namespace

int dadd_init_igemm() {
  int rc = dadd_init_igemm_dma();
  if (rc == DADD_OK) rc = dadd_init_conv_halo();
  if (rc == DADD_OK) rc = set_attr<128, 128, false>();
  if (rc == DADD_OK) rc = set_attr<128, 128, true>();
  if (rc == DADD_OK) rc = set_attr<128, 160, false>();
  if (rc == DADD_OK) rc = set_attr<64, 128, false>();
  if (rc == DADD_OK) rc = set_attr<64, 128, true>();
  if (rc == DADD_OK) rc = set_attr<64, 160, false>();
  if (rc == DADD_OK) rc = set_attr<64, 160, true>();
  return rc;
}

extern "C" int dadd_conv_igemm_f16(const dadd_igemm_desc* d, void* stream) {
  DADD_REQUIRE(d != nullptr, "igemm: null descriptor");
  hipStream_t s = static_cast<hipStream_t>(stream);
  IgemmArgs a;
  a.x = static_cast<const half_t*>(d->x);
  a.x2 = static_cast<const half_t*>(d->x2);
  a.w = static_cast<const half_t*>(d->w);
  a.out = static_cast<half_t*>(d->out);
  a.partial = d->partial;
  a.bias = d->bias;
  a.rowvec = d->rowvec;
  a.residual = static_cast<const half_t*>(d->residual);
  a.counters = d->counters;
  a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.C1 = d->C1; a.C2 = d->C2;
  a.Ho = d->Ho; a.Wo = d->Wo; a.N = d->N;
  a.taps = d->taps; a.stride = d->stride; a.ups = d->ups; a.pad = d->pad;
  a.flags = d->flags & (15 | DADD_TUNE_PERSIST | DADD_EPI_LNFOLD | DADD_EPI_ACT_MASK | DADD_EPI_GNSTAT | DADD_EPI_LNSTAT |
                        DADD_PRE_GN | DADD_PRE_GN_SILU | DADD_EPI_GNAPPLY | DADD_EPI_GNAPPLY_SILU);
  a.gno_out = static_cast<half_t*>(d->gn_out);
  a.gno_gamma = d->gn_out_gamma;
  a.gno_beta = d->gn_out_beta;
  a.gno_eps = d->gn_out_eps;
  a.gn_ws = d->gn_ws;
  a.gn_nchunk = d->gn_nchunk;
  a.gn_cg = d->gn_cg;   // epilogue bits + the persistent-ring request
  a.ln_c1 = d->ln_c1;
  a.ln_eps = d->ln_eps;
  a.gni_ws = d->gn_in_ws;
  a.gni_gamma = d->gn_in_gamma;
  a.gni_beta = d->gn_in_beta;
  a.gni_nchunk = d->gn_in_nchunk;
  a.gni_eps = d->gn_in_eps;
  a.gni_ws2 = d->gn_in_ws2;
  a.gni_nchunk2 = d->gn_in_nchunk2;
  a.ln_stats_out = d->ln_stats_out;
  a.ln_stats_in = (a.flags & DADD_EPI_LNFOLD) ? d->ln_stats_in : nullptr;
  a.ln_parts_in = d->ln_parts_in;
  const int Cin = a.C1 + a.C2;
  const bool geglu = (a.flags & DADD_EPI_GEGLU) != 0;
  a.ldo = d->ldo > 0 ? d->ldo : (geglu ? a.N / 2 : a.N);
  a.ldr = d->ldr > 0 ? d->ldr : a.N;
  a.ld_rowvec = d->ld_rowvec > 0 ? d->ld_rowvec : a.N;

  DADD_REQUIRE(a.x && a.w && a.out, "igemm: null x/w/out");
  DADD_REQUIRE(a.taps == 1 || a.taps == 9, "igemm: taps must be 1 or 9, got %d", a.taps);
  DADD_REQUIRE(a.B > 0 && a.Hi > 0 && a.Wi > 0 && a.Ho > 0 && a.Wo > 0 && a.N > 0,
               "igemm: non-positive extent");
  DADD_REQUIRE(Cin % 64 == 0 && a.C1 % 64 == 0 && a.C1 > 0,
               "igemm: channels must be multiples of 64 (C1=%d C2=%d)", a.C1, a.C2);
  DADD_REQUIRE(a.C2 == 0 || a.x2 != nullptr, "igemm: C2>0 needs x2");
  DADD_REQUIRE(a.N % 8 == 0, "igemm: N must be a multiple of 8, got %d", a.N);
  DADD_REQUIRE(a.stride == 1 || a.stride == 2, "igemm: stride must be 1 or 2");
  DADD_REQUIRE(!(a.ups && (a.stride != 1 || a.taps != 9)), "igemm: ups needs a stride-1 3x3");
  DADD_REQUIRE(dadd_aligned16(a.x) && dadd_aligned16(a.w) && dadd_aligned16(a.out) &&
                   (a.x2 == nullptr || dadd_aligned16(a.x2)),
               "igemm: pointers must be 16-byte aligned");
  DADD_REQUIRE(!(a.flags & DADD_EPI_BIAS) || a.bias, "igemm: bias flag without bias");
  DADD_REQUIRE(!(a.flags & DADD_EPI_ROWVEC) || a.rowvec, "igemm: rowvec flag without rowvec");
  DADD_REQUIRE(!(a.flags & DADD_EPI_RESIDUAL) || a.residual, "igemm: residual flag without ptr");
  DADD_REQUIRE(!(a.flags & DADD_EPI_ACT_MASK) || (!geglu && d->splitk <= 1),
               "igemm: an activation epilogue excludes GEGLU and split-K");
  DADD_REQUIRE(!(a.flags & DADD_EPI_LNFOLD) || (a.ln_c1 && a.taps == 1 && a.C2 == 0 && d->splitk <= 1 && a.ln_eps > 0.f),
               "igemm: a folded LayerNorm needs c1, a plain linear over one source (K = C) and no split-K");
  DADD_REQUIRE(a.ln_stats_in == nullptr ||
                   (a.ln_parts_in > 0 && (size_t)a.ln_parts_in * a.B * a.Ho * a.Wo * 8 < 0x7FF00000ull),
               "igemm: ln_stats_in needs ln_parts_in > 0 and fewer than 2 GiB of partials");
  DADD_REQUIRE(a.ldo % 4 == 0 && a.ldr % 4 == 0 && a.ld_rowvec % 4 == 0,
               "igemm: leading dimensions must be multiples of 4");

  a.M = a.B * a.Ho * a.Wo;
  a.K = a.taps * Cin;
  a.nkt = a.K / BK;
  int tile_n = d->tile_n;
  if (geglu) {
    DADD_REQUIRE(a.N % 128 == 0 && (tile_n == 0 || tile_n == 128), "igemm: GEGLU needs N%%128==0");
    tile_n = 128;
  }
  if (tile_n == 0) tile_n = (a.N % 160 == 0) ? 160 : 128;
  DADD_REQUIRE(tile_n == 64 || tile_n == 128 || tile_n == 160, "igemm: tile_n must be 64, 128 or 160");
  a.ntiles = (a.N + tile_n - 1) / tile_n;

  int splitk = d->splitk > 1 ? d->splitk : 1;
  if (splitk > a.nkt) splitk = a.nkt;
  a.kps = (a.nkt + splitk - 1) / splitk;
  const int nsplit = (a.nkt + a.kps - 1) / a.kps;
  a.splitk = nsplit;
  DADD_REQUIRE(nsplit == 1 || (a.partial != nullptr && !geglu),
               "igemm: split-K needs a partial buffer and no GEGLU");

  // two K tiles in flight except where that spills (128x160) or when the caller asks for one
  bool deep = (d->flags & DADD_TUNE_SHALLOW) == 0;
  int tile_m = d->tile_m;
  DADD_REQUIRE(tile_m == 0 || tile_m == 64 || tile_m == 128, "igemm: tile_m must be 0, 64 or 128");
  if (tile_m == 0) tile_m = 128;
  // tile order (igemm_args.h tile_decode): split the larger operand across the XCDs' L2s
  a.mtiles = (a.M + tile_m - 1) / tile_m;
  {
    const double a_bytes = 2.0 * a.B * a.Hi * a.Wi * Cin, w_bytes = 2.0 * a.N * a.K;
    if (a_bytes >= w_bytes) {
      a.gm = (a.mtiles + 7) / 8;
      a.gn = a.ntiles;
    } else {
      a.gm = a.mtiles;
      a.gn = (a.ntiles + 7) / 8;
    }
    // Linears whose A does not fit one XCD's L2 although W is the larger operand (the GEGLU projection at 32x32: A 5.2 MB,
    // W 6.5 MB): with "all row tiles x 1/8 of the column tiles" per XCD every XCD streamed all of A once per column tile
    // (FETCH_SIZE 242 MB for 12 MB of operands).  2-D groups - 1/2 (1/4) of the row tiles x 1/4 (1/2) of the
    // column tiles per XCD, row tile fastest - keep the XCD's A block (<= L2_KEEP) resident while its weight tiles stream.
    if (a.taps == 1 && !(a_bytes >= w_bytes) && a_bytes > L2_KEEP) {
      double best = 1e30;
      for (int ms = 2; ms <= 4; ms *= 2) {
        const int ns = 8 / ms;
        if (a.mtiles % ms || a.ntiles % ns || a_bytes / ms > L2_KEEP) continue;
        const double fetch = a_bytes / ms + w_bytes / ns;
        if (fetch < best) { best = fetch; a.gm = a.mtiles / ms; a.gn = a.ntiles / ns; }
      }
    }
  }
  // K order of the LDS-DMA kernel: channels fastest.  (Taps fastest — the nine taps of a 64-channel chunk as
  // consecutive K tiles — measured no different: the DMA stream is bound by the L2->LDS fill rate of a CU, not by
  // L1 hits: profiles/r01_x_dma_limits.txt.)
  a.korder = 0;
  int rc;
  // LDS-DMA ring kernel (igemm_dma.hip) for 128-row tiles; the register-staged kernel below keeps
  // the 64-row tiles and serves as the A/B reference (DADD_TUNE_NODMA)
  // (the 64-row LDS-DMA tiles have no upsample gather: such a request runs on the register-staged kernel; a folded
  // LayerNorm exists on the LDS-DMA kernel only)
  // (... unless its row statistics come from the producer of x: then the fold is epilogue arithmetic on every kernel)
  const bool dma = ((d->flags & DADD_TUNE_NODMA) == 0 || ((a.flags & DADD_EPI_LNFOLD) && !a.ln_stats_in)) &&
                   !(tile_m == 64 && a.ups);
  DADD_REQUIRE(tile_n != 64 || (dma && tile_m == 64 && !geglu), "igemm: 64-column tiles exist for the 64-row LDS-DMA kernel only");
  // persistent ring: a workgroup walks a contiguous run of tiles.  Activation-heavy GEMMs: column tile fastest, the
  // run keeps ONE activation row tile (L2-hot after the first tile) and streams the (L2-resident) weight tiles.
  // Weight-heavy GEMMs (the GEGLU projection of the 16x16 map: 26 MB of weights against 2.6 MB of activations) keep
  // the grouped order: row tile fastest inside a group of column tiles, so the runs of neighbouring workgroups —
  // same XCD, started together — read each weight tile from HBM once instead of once per row tile.
  if (dma && tile_m == 128 && dadd_igemm_dma_persistent(a, nsplit) &&
      ((d->flags & DADD_TUNE_SHALLOW) ? false : 2.0 * a.B * a.Hi * a.Wi * Cin >= 2.0 * a.N * a.K))
    a.gm = a.gn = 0;
  // 3x3 / stride 1 on whole-row tiles: the halo-resident kernel (conv_halo.hip); K slices = channel chunks
  const bool halo = dma && tile_m == 128 && dadd_conv_halo_applicable(a, tile_n);
  int halo_ns = 1;
  if (halo) {
    const int chunks = Cin / BK;
    int sk = d->splitk > 1 ? d->splitk : 1;
    if (sk > chunks) sk = chunks;
    a.kps = (chunks + sk - 1) / sk;
    halo_ns = (chunks + a.kps - 1) / a.kps;
    a.splitk = halo_ns;
    DADD_REQUIRE(halo_ns == 1 || a.partial != nullptr, "igemm: split-K needs a partial buffer");
  }
  if ((a.flags & DADD_EPI_GNSTAT) && (halo ? halo_ns : nsplit) > 1) {
    // split-K: the finish kernel writes the partials, per chunk of 16 (or 64) rows and 160- (or 128-) column block of whole groups
    const int cb = (a.N % 160 == 0) ? 160 : 128, howo = a.Ho * a.Wo;
    const int rows = a.gn_nchunk > 0 ? howo / a.gn_nchunk : 0;
    DADD_REQUIRE(a.gn_ws && a.gn_cg > 0 && a.N == 32 * a.gn_cg && !geglu && a.counters == nullptr && a.N % cb == 0 &&
                     cb % a.gn_cg == 0 && (rows == 16 || rows == 64) && a.M % rows == 0 && howo % rows == 0 &&
                     a.gn_nchunk * rows == howo && !(a.flags & (DADD_EPI_LNFOLD | DADD_EPI_ACT_MASK)),
                 "igemm: GroupNorm statistics with split-K need the finish kernel (no tickets), N == 32 groups in blocks of "
                 "160 / 128 columns, and chunks of 16 or 64 rows: gn_nchunk == Ho*Wo / 16 (or / 64)");
  } else if (a.flags & DADD_EPI_GNSTAT) {
    const int wm_rows = tile_m / 2, wn_cols = tile_n / 2, howo = a.Ho * a.Wo;
    DADD_REQUIRE(a.gn_ws && a.gn_cg > 0 && a.N == 32 * a.gn_cg && !geglu && (tile_n == 128 || tile_n == 160) &&
                     wn_cols % a.gn_cg == 0 && a.M % tile_m == 0 && a.N % tile_n == 0 && howo % wm_rows == 0 &&
                     a.gn_nchunk == howo / wm_rows && !(a.flags & (DADD_EPI_LNFOLD | DADD_EPI_ACT_MASK)),
                 "igemm: GroupNorm statistics need full tiles of 128/160 columns holding whole groups, whole-wave row blocks "
                 "inside one sample (Ho*Wo %% %d == 0, gn_nchunk == Ho*Wo / %d), N == 32 groups, no split-K", wm_rows, wm_rows);
  }
  if (a.flags & DADD_EPI_GNAPPLY) {
    DADD_REQUIRE((halo ? halo_ns : nsplit) > 1 && a.counters == nullptr && a.gno_out && a.gno_gamma && a.gno_beta &&
                     a.gno_eps > 0.f && a.N % 64 == 0 && a.ldo % 2 == 0 && a.ldr % 2 == 0 &&
                     (size_t)a.Ho * a.Wo * (a.N / 32) * 2 <= 16 * 1024 && dadd_aligned16(a.gno_out) &&
                     !(a.flags & (DADD_EPI_GNSTAT | DADD_EPI_LNFOLD | DADD_EPI_LNSTAT | DADD_EPI_ACT_MASK)),
                 "igemm: GroupNorm of the output in the finish kernel needs a finish-kernel split-K launch, gn_out / gamma / "
                 "beta / eps, N %% 64 == 0, a (sample, group) slab of <= 16 KiB and no other statistics / activation flags");
  } else {
    DADD_REQUIRE(!(a.flags & DADD_EPI_GNAPPLY_SILU), "igemm: DADD_EPI_GNAPPLY_SILU without DADD_EPI_GNAPPLY");
  }
  if (a.flags & DADD_EPI_LNSTAT) {
    const int wn_cols = tile_n / 2;
    DADD_REQUIRE(a.ln_stats_out && !geglu && !halo && a.N % wn_cols == 0 && d->ln_parts_out == a.N / wn_cols &&
                     (nsplit == 1 || a.counters != nullptr) && !(a.flags & DADD_EPI_GNSTAT),
                 "igemm: LayerNorm row partials need ln_stats_out, N %% (tile_n/2) == 0, ln_parts_out == N / (tile_n/2) = %d, "
                 "no GEGLU and no finish-kernel split-K", a.N / wn_cols);
  }
  if (a.flags & DADD_PRE_GN) {
    const int cgc = Cin / 32;      // group width of the (concatenated) input
    DADD_REQUIRE(a.C2 == 0 || (a.gni_ws2 && a.gni_nchunk2 >= 1 && a.gni_nchunk2 <= 256 && cgc > 0 && a.C1 % 32 == 0 &&
                               a.C2 % 32 == 0 && cgc % (a.C1 / 32) == 0 && cgc % (a.C2 / 32) == 0 && a.C1 % cgc == 0),
                 "igemm: GroupNorm over a concatenation needs the partials of both sources and group widths that nest "
                 "((C1+C2)/32 a multiple of C1/32 and C2/32, dividing C1)");
    DADD_REQUIRE(halo && !(d->flags & DADD_TUNE_SHALLOW) && Cin <= dadd_conv_halo_gn_channels(a.Wo) &&
                     Cin % 32 == 0 && a.gni_ws &&
                     a.gni_gamma && a.gni_beta && a.gni_nchunk >= 1 && a.gni_nchunk <= 256 && a.gni_eps > 0.f,
                 "igemm: GroupNorm on the way in needs the 3x3 halo kernel (stride 1, 64/32/16-wide map, 128x160 tiles), "
                 "Cin <= 1152 / 2048 / 2432 (W = 64 / 32 / 16), the chunk partials (<= 256 chunks), gamma, beta and eps");
  } else {
    DADD_REQUIRE(!(a.flags & DADD_PRE_GN_SILU), "igemm: DADD_PRE_GN_SILU without DADD_PRE_GN");
  }
  if (halo) {
    if (d->flags & DADD_TUNE_SHALLOW) a.flags |= DADD_TUNE_SHALLOW;   // A/B: the two-MFMA-waves-per-SIMD build
    rc = dadd_launch_conv_halo(a, halo_ns, s);
  }
  else if (dma)
    rc = dadd_launch_igemm_dma(a, tile_m, tile_n, nsplit, s);
  else if (tile_m == 128)
    rc = (tile_n == 160) ? launch<128, 160, false>(a, nsplit, s) : launch2<128, 128>(a, nsplit, deep, s);
  else
    rc = (tile_n == 160) ? launch2<64, 160>(a, nsplit, deep, s) : launch2<64, 128>(a, nsplit, deep, s);
  if (rc != DADD_OK) return rc;
  if ((halo ? halo_ns : nsplit) > 1 && a.counters == nullptr) {
    const size_t total = (size_t)a.M * (a.N / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    const int ns = halo ? halo_ns : nsplit;
    const DaddLaunchTag tag = {(a.flags & DADD_EPI_GNSTAT) ? "splitk_finish_gn_kernel" : "splitk_finish_kernel", 0.0,
                               (double)a.M * a.N * (4.0 * ns + 2.0 + ((a.flags & DADD_EPI_RESIDUAL) ? 2.0 : 0.0))};
    if (a.flags & DADD_EPI_GNAPPLY) {
      const DaddLaunchTag tag2 = {"splitk_finish_gnapply_kernel", 0.0, tag.bytes + 2.0 * a.M * a.N};
      const unsigned slab_bytes = (unsigned)(a.Ho * a.Wo * (a.N / 32) * 2);
      if ((a.N / 32) % 4 == 0 && a.ldo % 4 == 0 && a.ldr % 4 == 0 && a.ld_rowvec % 4 == 0)
        dadd_launch(tag2, splitk_finish_gnapply_kernel<4>, dim3(32, a.B), dim3(512), slab_bytes, s, a, ns);
      else
        dadd_launch(tag2, splitk_finish_gnapply_kernel<2>, dim3(32, a.B), dim3(512), slab_bytes, s, a, ns);
    } else if (!(a.flags & DADD_EPI_GNSTAT))
      dadd_launch(tag, splitk_finish_kernel, dim3(blocks), dim3(256), 0, s, a, ns);
    else {
      const int rows = (a.Ho * a.Wo) / a.gn_nchunk;
      if (a.N % 160 == 0) {
        if (rows == 16) dadd_launch(tag, splitk_finish_gn_kernel<160, 16>, dim3(a.M / 16, a.N / 160), dim3(256), 0, s, a, ns);
        else dadd_launch(tag, splitk_finish_gn_kernel<160, 64>, dim3(a.M / 64, a.N / 160), dim3(256), 0, s, a, ns);
      } else {
        if (rows == 16) dadd_launch(tag, splitk_finish_gn_kernel<128, 16>, dim3(a.M / 16, a.N / 128), dim3(256), 0, s, a, ns);
        else dadd_launch(tag, splitk_finish_gn_kernel<128, 64>, dim3(a.M / 64, a.N / 128), dim3(256), 0, s, a, ns);
      }
    }
    DADD_LAUNCH_CHECK();
  }
  return DADD_OK;
}
