// GroupNorm(+SiLU) over NHWC fp16 with a fused skip-concat, and LayerNorm — HBM-bound kernels:
// 16-byte loads/stores, fp32 statistics, wavefront-shuffle / LDS reductions, no atomics (results
// are bit-reproducible run to run).
#include "dadd_common.h"

namespace {

constexpr int GN_MAXV = 2;  // 16-byte channel vectors per thread: C <= 8*256*2 = 4096

struct GnArgs {
  const half_t* x1;
  const half_t* x2;
  const float* gamma;
  const float* beta;
  half_t* out;
  float* ws;     // [B][nchunk][groups][2] chunk partials
  float* stats;  // [B][groups][2] mean, rstd (tail of the workspace)
  int C1, C2, C, HW, groups, cg, nchunk, rows_per_chunk, rows_per_block, TV, RP, silu;
  float eps;
};

__device__ __forceinline__ h8 gn_load(const GnArgs& p, size_t pix, int c) {
  return (c < p.C1) ? *reinterpret_cast<const h8*>(p.x1 + pix * p.C1 + c)
                    : *reinterpret_cast<const h8*>(p.x2 + pix * p.C2 + (c - p.C1));
}

// pass 1: per (batch, row-chunk) partial sums per group.  grid (nchunk, B)
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnArgs p) {
  extern __shared__ float sm[];  // [RP][C] sums, [RP][C] squares
  const int t = threadIdx.x, tv = t % p.TV, tr = t / p.TV;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int row0 = chunk * p.rows_per_chunk;
  const int row1 = min(p.HW, row0 + p.rows_per_chunk);
  float s[GN_MAXV][8], ss[GN_MAXV][8];
#pragma unroll
  for (int u = 0; u < GN_MAXV; ++u)
#pragma unroll
    for (int e = 0; e < 8; ++e) s[u][e] = ss[u][e] = 0.f;
  const int nvec = p.C >> 3;
  if (tr < p.RP) {
    // four rows per trip, all loads issued before the first use: one 16-B load in flight per thread
    // made this pass latency-bound (9.5 us for 10.5 MB)
    for (int row = row0 + tr; row < row1; row += 4 * p.RP) {
      h8 xv[4][GN_MAXV];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int rr = row + q * p.RP;
        const size_t pix = (size_t)b * p.HW + (rr < row1 ? rr : row);
#pragma unroll
        for (int u = 0; u < GN_MAXV; ++u) {
          const int v = tv + u * p.TV;
          if (v < nvec) xv[q][u] = gn_load(p, pix, v * 8);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (row + q * p.RP >= row1) continue;
#pragma unroll
        for (int u = 0; u < GN_MAXV; ++u) {
          if (tv + u * p.TV < nvec) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float f = (float)xv[q][u][e];
              s[u][e] += f;
              ss[u][e] += f * f;
            }
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < GN_MAXV; ++u) {
      const int v = tv + u * p.TV;
      if (v < nvec) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sm[tr * p.C + v * 8 + e] = s[u][e];
          sm[(p.RP + tr) * p.C + v * 8 + e] = ss[u][e];
        }
      }
    }
  }
  __syncthreads();
  {   // group sums: 8 lanes per group (fixed order: lane-strided partials, then a 3-step butterfly)
    const int g = t >> 3, sub = t & 7;
    float a = 0.f, q = 0.f;
    if (g < p.groups) {
      const int n = p.RP * p.cg;
      for (int idx = sub; idx < n; idx += 8) {
        const int r = idx / p.cg, c = g * p.cg + (idx - r * p.cg);
        a += sm[r * p.C + c];
        q += sm[(p.RP + r) * p.C + c];
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      a += __shfl_xor(a, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if (g < p.groups && sub == 0) {
      float* w = p.ws + (((size_t)b * p.nchunk + chunk) * p.groups + g) * 2;
      w[0] = a;
      w[1] = q;
    }
  }
}

// pass 2: every block combines the (<= 64) chunk partials of its sample in a fixed order (double), folds
// gamma/beta into per-channel scale/shift (C values per block) and applies them to its rows.  grid (row blocks, B).
// (A separate finalize launch cost 5 us per GroupNorm for 16 KB of partials.)
template <bool FIN>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnArgs p) {
  extern __shared__ float sm[];  // [C] scale, [C] shift, [2*groups] mean/rstd (FIN)
  float* scale = sm;
  float* shift = sm + p.C;
  const int t = threadIdx.x, tv = t % p.TV, tr = t / p.TV;
  const int b = blockIdx.y;
  const float* st = p.stats + (size_t)b * p.groups * 2;
  if (FIN) {
    float* lst = sm + 2 * p.C;
    const int g = t >> 3, sub = t & 7;
    double a = 0.0, q = 0.0;
    if (g < p.groups) {
      for (int k = sub; k < p.nchunk; k += 32) {     // same order of summation as gn_finalize_kernel
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kk = k + 8 * u;
          v[u] = kk < p.nchunk ? *reinterpret_cast<const float2*>(p.ws + (((size_t)b * p.nchunk + kk) * p.groups + g) * 2)
                               : float2{0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a += (double)v[u].x;
          q += (double)v[u].y;
        }
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      a += __shfl_xor(a, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if (g < p.groups && sub == 0) {
      const double n = (double)p.HW * (double)p.cg;
      const double mu = a / n;
      double var = q / n - mu * mu;
      if (var < 0.0) var = 0.0;
      lst[2 * g] = (float)mu;
      lst[2 * g + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    __syncthreads();
    st = lst;
  }
  for (int c = t; c < p.C; c += 256) {
    const int g = c / p.cg;
    const float sc = st[2 * g + 1] * p.gamma[c];
    scale[c] = sc;
    shift[c] = p.beta[c] - st[2 * g] * sc;
  }
  __syncthreads();
  if (tr >= p.RP) return;
  const int nvec = p.C >> 3;
  const int row0 = blockIdx.x * p.rows_per_block;
  const int row1 = min(p.HW, row0 + p.rows_per_block);
  float sc[GN_MAXV][8], sh[GN_MAXV][8];     // this thread's channels are the same for every row
#pragma unroll
  for (int u = 0; u < GN_MAXV; ++u) {
    const int v = tv + u * p.TV;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[u][e] = v < nvec ? scale[v * 8 + e] : 0.f;
      sh[u][e] = v < nvec ? shift[v * 8 + e] : 0.f;
    }
  }
  for (int row = row0 + tr; row < row1; row += 4 * p.RP) {   // four rows in flight per thread
    h8 xv[4][GN_MAXV];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = row + q * p.RP;
      const size_t pix = (size_t)b * p.HW + (rr < row1 ? rr : row);
#pragma unroll
      for (int u = 0; u < GN_MAXV; ++u)
        if (tv + u * p.TV < nvec) xv[q][u] = gn_load(p, pix, (tv + u * p.TV) * 8);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rr = row + q * p.RP;
      if (rr >= row1) continue;
      const size_t pix = (size_t)b * p.HW + rr;
#pragma unroll
      for (int u = 0; u < GN_MAXV; ++u) {
        const int v = tv + u * p.TV;
        if (v < nvec) {
          h8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = (float)xv[q][u][e] * sc[u][e] + sh[u][e];
            if (p.silu) f = dadd_silu(f);
            o[e] = (half_t)f;
          }
          *reinterpret_cast<h8*>(p.out + pix * p.C + v * 8) = o;
        }
      }
    }
  }
}

// Single-launch GroupNorm for feature maps whose (batch, group) slab fits LDS (every UNet level
// below 64x64x640): one block per (group, batch) stages its HW x cg slab in LDS while summing,
// reduces, then normalises from LDS.  One launch and two HBM passes instead of two launches and
// three passes; the slab rows are cg*2 bytes (20..160 B) so the accesses are 4-byte pairs, which
// is fine for tensors that live in L2.
__global__ __launch_bounds__(256) void gn_fused_kernel(const GnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smraw[];
  h2* slab = reinterpret_cast<h2*>(smraw);                 // [HW][cg/2]
  __shared__ float red[2][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int g = blockIdx.x, b = blockIdx.y;
  const int hp = p.cg >> 1;                                // channel pairs per row of the slab
  const int c0 = g * p.cg;
  const int npair = p.HW * hp;
  float s = 0.f, ss = 0.f;
  for (int idx = t; idx < npair; idx += 256) {
    const int row = idx / hp, cp = idx - row * hp;
    const int c = c0 + 2 * cp;
    const size_t pix = (size_t)b * p.HW + row;
    const h2 v = (c < p.C1) ? *reinterpret_cast<const h2*>(p.x1 + pix * p.C1 + c)
                            : *reinterpret_cast<const h2*>(p.x2 + pix * p.C2 + (c - p.C1));
    slab[idx] = v;
    const float a = (float)v[0], d = (float)v[1];
    s += a + d;
    ss += a * a + d * d;
  }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if (lane == 0) { red[0][wave] = s; red[1][wave] = ss; }
  __syncthreads();
  const double n = (double)p.HW * (double)p.cg;
  const double sum = (double)red[0][0] + (double)red[0][1] + (double)red[0][2] + (double)red[0][3];
  const double sq = (double)red[1][0] + (double)red[1][1] + (double)red[1][2] + (double)red[1][3];
  const double mu = sum / n;
  double var = sq / n - mu * mu;
  if (var < 0.0) var = 0.0;
  const float mean = (float)mu, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
  for (int idx = t; idx < npair; idx += 256) {
    const int row = idx / hp, cp = idx - row * hp;
    const int c = c0 + 2 * cp;
    const h2 v = slab[idx];
    h2 o;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float sc = rstd * p.gamma[c + e];
      float f = ((float)v[e] - mean) * sc + p.beta[c + e];
      if (p.silu) f = dadd_silu(f);
      o[e] = (half_t)f;
    }
    *reinterpret_cast<h2*>(p.out + ((size_t)b * p.HW + row) * p.C + c) = o;
  }
}

constexpr int GN_CHUNK_MAX = 64;
constexpr int GN_FUSED_MAX_BYTES = 16 * 1024;   // measured: wins only below ~16 KiB per (batch, group) slab

// LayerNorm: one wave per row, the row lives in registers (exact two-pass variance).
constexpr int LN_MAXV = 4;  // C <= 8*64*4 = 2048
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        half_t* __restrict__ out, int M, int C,
                                                        float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nvec = C >> 3;
  const half_t* xr = x + (size_t)row * C;
  h8 v[LN_MAXV];
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    const int i = lane + 64 * u;
    if (i < nvec) {
      v[u] = *reinterpret_cast<const h8*>(xr + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += (float)v[u][e];
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    const int i = lane + 64 * u;
    if (i < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = (float)v[u][e] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int u = 0; u < LN_MAXV; ++u) {
    const int i = lane + 64 * u;
    if (i < nvec) {
      h8 o;
      const f4 g0 = *reinterpret_cast<const f4*>(gamma + i * 8);
      const f4 g1 = *reinterpret_cast<const f4*>(gamma + i * 8 + 4);
      const f4 b0 = *reinterpret_cast<const f4*>(beta + i * 8);
      const f4 b1 = *reinterpret_cast<const f4*>(beta + i * 8 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        o[e] = (half_t)(((float)v[u][e] - mean) * rstd * g0[e] + b0[e]);
        o[e + 4] = (half_t)(((float)v[u][e + 4] - mean) * rstd * g1[e] + b1[e]);
      }
      *reinterpret_cast<h8*>(out + (size_t)row * C + i * 8) = o;
    }
  }
}

}  // namespace

int dadd_init_norm() {
  DADD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gn_fused_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, GN_FUSED_MAX_BYTES));
  return DADD_OK;
}

// Producer statistics of a large map (VAE: 4096 chunk partials per sample at 512x512): folded down to <= 64 chunks per
// sample by one small launch, so that every apply block can still combine them itself.  Block (r, b) sums the chunks
// r, r + R, r + 2R, ... of sample b: thread = (quarter of those chunks, one of the 64 (group, sum | sum of squares)
// floats), fixed order, fp32 partials of at most a few hundred terms (the apply kernel continues in fp64).
__global__ __launch_bounds__(256) void gn_reduce_kernel(const float* __restrict__ ws, float* __restrict__ red, int nchunk,
                                                        int R) {
  __shared__ float part[4][64];
  const int t = threadIdx.x, idx = t & 63, q = t >> 6;
  const int r = blockIdx.x, b = blockIdx.y;
  float a = 0.f;
  for (int k = r + q * R; k < nchunk; k += 4 * R) a += ws[((size_t)b * nchunk + k) * 64 + idx];
  part[q][idx] = a;
  __syncthreads();
  if (t < 64) red[((size_t)b * R + r) * 64 + t] = ((part[0][t] + part[1][t]) + part[2][t]) + part[3][t];
}

extern "C" int dadd_groupnorm_f16(const void* x1, int C1, const void* x2, int C2,
                                  const float* gamma, const float* beta, void* out, float* ws,
                                  int B, int HW, int groups, float eps, int silu, int ws_chunks, void* stream) {
  const int C = C1 + C2;
  DADD_REQUIRE(x1 && gamma && beta && out && ws, "groupnorm: null pointer");
  DADD_REQUIRE(C1 > 0 && C1 % 8 == 0 && C2 >= 0 && C2 % 8 == 0, "groupnorm: C1/C2 must be x8");
  DADD_REQUIRE(C2 == 0 || x2, "groupnorm: C2>0 needs x2");
  DADD_REQUIRE(groups > 0 && groups <= 32 && C % groups == 0, "groupnorm: groups must be <= 32 and divide C");
  DADD_REQUIRE(C <= 8 * 256 * GN_MAXV, "groupnorm: C=%d too large", C);
  DADD_REQUIRE(B > 0 && HW > 0, "groupnorm: empty input");
  DADD_REQUIRE(dadd_aligned16(x1) && dadd_aligned16(out) && (!x2 || dadd_aligned16(x2)),
               "groupnorm: pointers must be 16-byte aligned");
  GnArgs p;
  p.x1 = static_cast<const half_t*>(x1);
  p.x2 = static_cast<const half_t*>(x2);
  p.gamma = gamma;
  p.beta = beta;
  p.out = static_cast<half_t*>(out);
  p.ws = ws;
  p.C1 = C1; p.C2 = C2; p.C = C; p.HW = HW; p.groups = groups; p.cg = C / groups;
  p.silu = silu; p.eps = eps;
  const int nvec = C / 8;
  p.TV = nvec < 256 ? nvec : 256;
  p.RP = 256 / p.TV;
  // stat chunks: at least two passes of rows per block, at most GN_CHUNK_MAX per sample (few enough that every
  // apply block combines them itself: no finalize launch)
  int nchunk = HW / (2 * p.RP > 16 ? 2 * p.RP : 16);
  if ((long)B * nchunk < 256) nchunk = HW / (4 * p.RP);      // small maps: one four-row trip per workgroup fills more of the chip
  if (nchunk > GN_CHUNK_MAX) nchunk = GN_CHUNK_MAX;
  if (nchunk < 1) nchunk = 1;
  p.stats = ws + (size_t)B * (DADD_GN_MAX_CHUNKS - 1) * groups * 2;   // last chunk slot of the workspace
  p.rows_per_chunk = (HW + nchunk - 1) / nchunk;
  p.nchunk = (HW + p.rows_per_chunk - 1) / p.rows_per_chunk;
  p.rows_per_block = 8 * p.RP;
  if ((long)B * ((HW + p.rows_per_block - 1) / p.rows_per_block) < 256) p.rows_per_block = 4 * p.RP;   // small maps: fill the chip
  const int nrb = (HW + p.rows_per_block - 1) / p.rows_per_block;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (ws_chunks > 0) {     // partials already written by the producer's epilogue: [B][ws_chunks][groups][2] at ws
    DADD_REQUIRE(C2 == 0 && groups == 32, "groupnorm: producer statistics need one source and 32 groups");
    p.nchunk = ws_chunks;
    if (ws_chunks > 2 * GN_CHUNK_MAX) {       // many chunks: fold them to R <= 64 per sample behind the partials
      const int R = 64;
      float* red = ws + (size_t)B * ws_chunks * 64;
      dadd_launch({"gn_reduce_kernel", 0.0, (double)B * ws_chunks * 256.0}, gn_reduce_kernel, dim3(R, B), dim3(256), 0, s,
                  (const float*)ws, red, ws_chunks, R);
      DADD_LAUNCH_CHECK();
      p.ws = red;
      p.nchunk = R;
    }
    p.rows_per_chunk = (HW + p.nchunk - 1) / p.nchunk;
    const size_t sm2p = ((size_t)2 * C + 2 * groups) * sizeof(float);
    dadd_launch({"gn_apply_kernel<true>", 0.0, (double)B * HW * C * 4.0}, gn_apply_kernel<true>, dim3(nrb, B), dim3(256), (unsigned)sm2p, s, p);
    DADD_LAUNCH_CHECK();
    return DADD_OK;
  }
  const size_t slab_bytes = (size_t)HW * p.cg * sizeof(half_t);
  const double act_bytes = (double)B * HW * C * 2.0;
  if (slab_bytes <= (size_t)GN_FUSED_MAX_BYTES && p.cg % 2 == 0 && C1 % 2 == 0) {
    dadd_launch({"gn_fused_kernel", 0.0, act_bytes * 2.0}, gn_fused_kernel, dim3(groups, B), dim3(256), (unsigned)slab_bytes, s, p);
    DADD_LAUNCH_CHECK();
    return DADD_OK;
  }
  const size_t sm1 = (size_t)2 * p.RP * C * sizeof(float);
  const size_t sm2 = ((size_t)2 * C + 2 * groups) * sizeof(float);
  dadd_launch({"gn_stats_kernel", 0.0, act_bytes}, gn_stats_kernel, dim3(p.nchunk, B), dim3(256), (unsigned)sm1, s, p);
  DADD_LAUNCH_CHECK();
  dadd_launch({"gn_apply_kernel<true>", 0.0, act_bytes * 2.0}, gn_apply_kernel<true>, dim3(nrb, B), dim3(256), (unsigned)sm2, s, p);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_layernorm_f16(const void* x, const float* gamma, const float* beta, void* out,
                                  int M, int C, float eps, void* stream) {
  DADD_REQUIRE(x && gamma && beta && out, "layernorm: null pointer");
  DADD_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 8 * 64 * LN_MAXV,
               "layernorm: C=%d must be a multiple of 8 and <= %d", C, 8 * 64 * LN_MAXV);
  DADD_REQUIRE(dadd_aligned16(x) && dadd_aligned16(out) && dadd_aligned16(gamma) &&
                   dadd_aligned16(beta), "layernorm: pointers must be 16-byte aligned");
  dadd_launch({"layernorm_kernel", 0.0, (double)M * C * 4.0}, layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0,
              static_cast<hipStream_t>(stream), static_cast<const half_t*>(x), gamma, beta,
              static_cast<half_t*>(out), M, C, eps);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}
