// Thin-channel convolutions at the ends of the UNet / VAE, layout conversion, time-embedding
// rows, and the DDIM update — the small HBM/latency-bound pieces around the MFMA kernels.
#include "dadd_common.h"

namespace {

// fp32 NCHW (C<=8 real channels) -> fp16 NHWC8, optional per-pixel CxC matrix + bias, then scale
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ x, half_t* __restrict__ out,
                                                   int B, int C, int HW, float scale,
                                                   const float* __restrict__ mat,
                                                   const float* __restrict__ vec) {
  const size_t total = (size_t)B * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t b = i / HW, pix = i - b * HW;
    float v[8], o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (c < C) ? x[(b * C + c) * HW + pix] * scale : 0.f;
    if (mat) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float a = 0.f;
        if (c < C) {
          a = vec ? vec[c] : 0.f;
          for (int k = 0; k < C; ++k) a += mat[c * C + k] * v[k];
        }
        o[c] = a;
      }
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = v[c];
    }
    h8 r;
#pragma unroll
    for (int c = 0; c < 8; ++c) r[c] = (half_t)o[c];
    *reinterpret_cast<h8*>(out + i * 8) = r;
  }
}

// conv3x3 pad1 from 8 stored channels: thread = (pixel, group of 8 output channels).  Consecutive
// threads are consecutive pixels of one output-channel group, so the weights are wave-uniform.
__global__ __launch_bounds__(256) void conv_cin8_kernel(const half_t* __restrict__ x,
                                                        const half_t* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        half_t* __restrict__ out, int B, int H, int W,
                                                        int Cout) {
  const int npix = B * H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int co0 = blockIdx.y * 8;
  if (pix >= npix) return;
  const int b = pix / (H * W), rem = pix - b * H * W, oy = rem / W, ox = rem - oy * W;
  float acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = bias ? bias[co0 + o] : 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
    const h8 xv = *reinterpret_cast<const h8*>(x + ((size_t)(b * H + iy) * W + ix) * 8);
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const h8 wv = *reinterpret_cast<const h8*>(w + ((size_t)(co0 + o) * 9 + tap) * 8);
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[o] += (float)xv[c] * (float)wv[c];
    }
  }
  h8 r;
#pragma unroll
  for (int o = 0; o < 8; ++o) r[o] = (half_t)acc[o];
  *reinterpret_cast<h8*>(out + (size_t)pix * Cout + co0) = r;
}

// conv_in of the UNet straight from the fp32 NCHW latents (C <= 4 channels): the layout change + fp16 rounding of
// pack_kernel happens in registers (same rounding point), two v_dot2_f32_f16 per tap and output channel instead of
// 24 converts + multiply-adds over the zero-padded 8 channels.  thread = (pixel, group of 8 output channels); the
// weights [Cout][9][8] (channels C.. zero) are wave-uniform.  Replaces pack_kernel + conv_cin8_kernel (4.3 + 28 us at
// 4x64x64 -> 320 channels: the old kernel was VALU-bound).
__global__ __launch_bounds__(256) void conv_in_nchw_kernel(const float* __restrict__ x, const half_t* __restrict__ w,
                                                           const float* __restrict__ bias, half_t* __restrict__ out,
                                                           int B, int C, int H, int W, int Cout) {
  typedef _Float16 ci_h2 __attribute__((ext_vector_type(2)));
  const int HW = H * W, npix = B * HW;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int co0 = blockIdx.y * 8;
  if (pix >= npix) return;
  const int b = pix / HW, rem = pix - b * HW, oy = rem / W, ox = rem - oy * W;
  ci_h2 x01[9], x23[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
    const float* xp = x + (size_t)b * C * HW + (ok ? iy * W + ix : rem);
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (c < C && ok) ? xp[(size_t)c * HW] : 0.f;
    x01[tap] = ci_h2{(_Float16)v[0], (_Float16)v[1]};
    x23[tap] = ci_h2{(_Float16)v[2], (_Float16)v[3]};
  }
  h8 r;
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    float acc = bias ? bias[co0 + o] : 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const h4 wv = *reinterpret_cast<const h4*>(w + ((size_t)(co0 + o) * 9 + tap) * 8);
      acc = __builtin_amdgcn_fdot2(x01[tap], ci_h2{wv[0], wv[1]}, acc, false);
      acc = __builtin_amdgcn_fdot2(x23[tap], ci_h2{wv[2], wv[3]}, acc, false);
    }
    r[o] = (half_t)acc;
  }
  *reinterpret_cast<h8*>(out + (size_t)pix * Cout + co0) = r;
}

// The same conv with the GroupNorm chunk partials of its output (the first ResNet's norm1 and the last up block's
// skip-concat then need no statistics pass and can normalise inside their 3x3 conv): a block owns 256 pixels (one chunk)
// and FOUR whole groups (4 * CG channels, eight at a time), every thread sums its pixel's rounded outputs per group, the
// block reduces them in a fixed order (butterfly per wave, then the four waves) and writes [b][chunk][group](sum, sum of
// squares).  grid (B*HW / 256, Cout / (4*CG)); HW % 256 == 0 so that a chunk never straddles two samples.
template <int CG>
__global__ __launch_bounds__(256) void conv_in_nchw_gn_kernel(const float* __restrict__ x, const half_t* __restrict__ w,
                                                              const float* __restrict__ bias, half_t* __restrict__ out,
                                                              float* __restrict__ gn_ws, int B, int C, int H, int W, int Cout) {
  typedef _Float16 ci_h2 __attribute__((ext_vector_type(2)));
  static_assert((4 * CG) % 8 == 0, "four groups must be whole 8-channel stores");
  __shared__ float red[4][8];
  const int HW = H * W;
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int co0 = blockIdx.y * 4 * CG;
  const int b = pix / HW, rem = pix - b * HW, oy = rem / W, ox = rem - oy * W;
  ci_h2 x01[9], x23[9];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
    const float* xp = x + (size_t)b * C * HW + (ok ? iy * W + ix : rem);
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (c < C && ok) ? xp[(size_t)c * HW] : 0.f;
    x01[tap] = ci_h2{(_Float16)v[0], (_Float16)v[1]};
    x23[tap] = ci_h2{(_Float16)v[2], (_Float16)v[3]};
  }
  float st[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // (sum, sum of squares) of the four groups
#pragma unroll
  for (int o8 = 0; o8 < 4 * CG / 8; ++o8) {
    h8 r;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const int co = co0 + o8 * 8 + o;
      float acc = bias ? bias[co] : 0.f;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const h4 wv = *reinterpret_cast<const h4*>(w + ((size_t)co * 9 + tap) * 8);
        acc = __builtin_amdgcn_fdot2(x01[tap], ci_h2{wv[0], wv[1]}, acc, false);
        acc = __builtin_amdgcn_fdot2(x23[tap], ci_h2{wv[2], wv[3]}, acc, false);
      }
      r[o] = (half_t)acc;
      const float f = (float)r[o];
      const int g = (o8 * 8 + o) / CG;      // compile-time after unrolling
      st[2 * g] += f;
      st[2 * g + 1] += f * f;
    }
    *reinterpret_cast<h8*>(out + (size_t)pix * Cout + co0 + o8 * 8) = r;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) st[i] = wave_sum(st[i]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) red[wave][i] = st[i];
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int i = threadIdx.x;
    const float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    const int chunk = (blockIdx.x * 256 - b * HW) >> 8, nchunk = HW >> 8;
    gn_ws[(((size_t)b * nchunk + chunk) * 32 + blockIdx.y * 4 + (i >> 1)) * 2 + (i & 1)] = v;
  }
}

// conv3x3 pad1 to <=4 output channels, fp32 NCHW output.  16 lanes share a pixel (channel chunks of 8 strided over the
// 16 lanes -> 256 contiguous bytes per tap) and every thread carries TWO pixels 16 apart, so one weight fragment read
// from LDS serves both; the nine taps of a chunk are loaded back to back (18 independent 16-byte loads in flight per
// lane) and multiplied with v_dot2_f32_f16 (exact fp16 products, fp32 sums).  The weights [Cout][9][C] are staged in
// LDS once per block.  (The first version — one pixel per thread, a tap loop with early-outs, converts + fma — took
// 50 us for the UNet's 320 -> 4 conv_out at 4x64x64: as long as a 3x3 conv with 80x the flops.)
typedef _Float16 cc_h2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void conv_cout4_kernel(const half_t* __restrict__ x,
                                                         const half_t* __restrict__ w,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ out, int B, int H, int W,
                                                         int C, int Cout, int mode,
                                                         const float* __restrict__ coef) {
#pragma clang fp contract(off)      // mode 3 is the DDIM update: individually rounded fp32 operations, as ddim_kernel
  extern __shared__ __attribute__((aligned(16))) char cc_smem[];
  half_t* ws = reinterpret_cast<half_t*>(cc_smem);          // [Cout][9][C]
  const int wtot = Cout * 9 * C;
  for (int i = threadIdx.x * 8; i < wtot; i += 256 * 8) *reinterpret_cast<h8*>(ws + i) = *reinterpret_cast<const h8*>(w + i);
  __syncthreads();
  const int npix = B * H * W, HW = H * W;
  const int sub = threadIdx.x & 15;
  const int nchunk = C >> 3;
  int pixs[2], bb[2], oy[2], ox[2];
  bool live[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int pix = blockIdx.x * 32 + q * 16 + (threadIdx.x >> 4);
    live[q] = pix < npix;
    if (!live[q]) pix = npix - 1;
    pixs[q] = pix;
    bb[q] = pix / HW;
    const int rem = pix - bb[q] * HW;
    oy[q] = rem / W;
    ox[q] = rem - oy[q] * W;
  }
  float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};   // pixel 0 / pixel 1 (separate arrays: a 2-D one went to scratch)
  const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int ch = sub; ch < nchunk; ch += 16) {
    h8 xv[2][9];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int iy = oy[q] + tap / 3 - 1, ix = ox[q] + tap % 3 - 1;
        const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
        const half_t* xp = x + ((size_t)(bb[q] * H + (ok ? iy : oy[q])) * W + (ok ? ix : ox[q])) * C + ch * 8;
        const h8 v = *reinterpret_cast<const h8*>(xp);      // unconditional load (clamped address), masked after
        xv[q][tap] = ok ? v : zero8;
      }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        if (o < Cout) {
          const h8 wv = *reinterpret_cast<const h8*>(ws + (o * 9 + tap) * C + ch * 8);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const cc_h2 wp = {wv[2 * c], wv[2 * c + 1]};
            a0[o] = __builtin_amdgcn_fdot2(cc_h2{xv[0][tap][2 * c], xv[0][tap][2 * c + 1]}, wp, a0[o], false);
            a1[o] = __builtin_amdgcn_fdot2(cc_h2{xv[1][tap][2 * c], xv[1][tap][2 * c + 1]}, wp, a1[o], false);
          }
        }
      }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o) {
#pragma unroll
    for (int s = 8; s > 0; s >>= 1) {
      a0[o] += __shfl_xor(a0[o], s, 64);
      a1[o] += __shfl_xor(a1[o], s, 64);
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    if (live[q] && sub < Cout) {
      float v = q ? a1[0] : a0[0];
      if (sub == 1) v = q ? a1[1] : a0[1];
      if (sub == 2) v = q ? a1[2] : a0[2];
      if (sub == 3) v = q ? a1[3] : a0[3];
      v += bias ? bias[sub] : 0.f;
      if (mode == 1) {
        v = fminf(fmaxf(v, -1.f), 1.f);
        v = (v + 1.f) / 2.f;
        v = fminf(fmaxf(v, 0.f), 1.f);
      } else if (mode == 2) {
        v = fminf(fmaxf(v, -30.f), 20.f);    // DiagonalGaussianDistribution clamps logvar on construction
      }
      float* dst = out + ((size_t)bb[q] * Cout + sub) * HW + (pixs[q] - bb[q] * HW);
      if (mode == 3) {                       // v = eps of this latent element: the DDIM step in place (ddim_kernel, no CFG)
        const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3];
        const float sx = c1 * v;
        const float num = *dst - sx;
        float x0 = num / c0;
        x0 = fminf(fmaxf(x0, -4.0f), 4.0f);
        const float ta = c2 * x0;
        const float tb = c3 * v;
        v = (c2 < 0.f) ? x0 : ta + tb;
      }
      *dst = v;
    }
  }
}

__global__ void timestep_features_kernel(const int64_t* __restrict__ t, float* __restrict__ out, int M,
                                         int dim) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * half) return;
  const int m = i / half, j = i - m * half;
  // freqs = exp(-ln(10000) * j / half), computed in fp32 like the reference engine
  const float freq = expf(-9.210340371976184f * (float)j / (float)half);
  const float ang = (float)t[m] * freq;
  out[(size_t)m * dim + j] = cosf(ang);
  out[(size_t)m * dim + half + j] = sinf(ang);
}

__device__ __forceinline__ float rows_act(float v, int act) {   // 0 none, 1 SiLU, 2 exact GELU (libm erf: fp32 path)
  return act == 1 ? dadd_silu(v) : (act == 2 ? 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)) : v);
}
__device__ __forceinline__ void rows_load8(const half_t* p, float (&o)[8]) {
  const h8 v = *reinterpret_cast<const h8*>(p);
#pragma unroll
  for (int c = 0; c < 8; ++c) o[c] = (float)v[c];
}
__device__ __forceinline__ void rows_load8(const float* p, float (&o)[8]) {
  const f4 a = *reinterpret_cast<const f4*>(p), b = *reinterpret_cast<const f4*>(p + 4);
#pragma unroll
  for (int c = 0; c < 4; ++c) { o[c] = a[c]; o[c + 4] = b[c]; }
}

// out[m][n] = act_out(sum_k act_in(x[m][k]) w[n][k] + bias[n]); one wave per output column,
// up to 8 rows of x staged in LDS per pass.  WT = half_t (time-embedding path) or float (AOE projector: its delta
// tokens are differences of two outputs, kept at fp32 weight fidelity).
template <typename WT>
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x,
                                                          const WT* __restrict__ w,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ out, int M, int K, int N,
                                                          int act_in, int act_out) {
  extern __shared__ float xs[];  // [8][K]
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nchunk = K >> 3;
  for (int m0 = 0; m0 < M; m0 += 8) {
    const int mr = min(8, M - m0);
    __syncthreads();
    for (int i = threadIdx.x; i < mr * K; i += 256) {
      xs[i] = rows_act(x[(size_t)m0 * K + i], act_in);
    }
    __syncthreads();
    if (n >= N) continue;
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
    for (int ch = lane; ch < nchunk; ch += 64) {
      float wv[8];
      rows_load8(w + (size_t)n * K + ch * 8, wv);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (r < mr) {
          const float* xr = xs + r * K + ch * 8;
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[r] += xr[c] * wv[c];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float s = wave_sum(acc[r]);
      if (lane == 0 && r < mr) {
        out[(size_t)(m0 + r) * N + n] = rows_act(s + (bias ? bias[n] : 0.f), act_out);
      }
    }
  }
}

// step[0] = row of this step, step[1] = arrival ticket (zero between launches).  Every block reads the row first; the
// block that draws the last ticket — after all the others have read — advances the row and resets the ticket.  (One
// block of 1024 threads took 17.6 us for the 0.4 MB of a B = 4 step; 80 blocks take the launch floor.)
__global__ __launch_bounds__(256) void begin_step_kernel(const float* __restrict__ table,
                                                         float* __restrict__ cur, int B, int ncols,
                                                         const float* __restrict__ coef,
                                                         float* __restrict__ cur_coef,
                                                         int32_t* __restrict__ step) {
  const int row = __hip_atomic_load(step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < ncols) {
    const float v = table[(size_t)row * ncols + i];
    for (int b = 0; b < B; ++b) cur[(size_t)b * ncols + i] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) cur_coef[threadIdx.x] = coef[row * 4 + threadIdx.x];
  __syncthreads();                  // every thread of this block has read `row`
  if (threadIdx.x == 0) {
    const int prev = __hip_atomic_fetch_add(step + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == (int)gridDim.x - 1) {
      __hip_atomic_store(step + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(step, row + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Every operation is an individually rounded fp32 op in the reference's order (no FMA contraction).
__global__ __launch_bounds__(256) void ddim_kernel(float* __restrict__ x, const float* __restrict__ ec,
                                                   const float* __restrict__ eu, float g_val,
                                                   const float* __restrict__ g_dev,
                                                   const float* __restrict__ coef, int64_t n) {
#pragma clang fp contract(off)  // __fmul_rn/__fsub_rn are plain * and - to the optimiser: forbid FMA fusion
  const float g = g_dev ? *g_dev : g_val;
  const float c0 = coef[0], c1 = coef[1], c2 = coef[2], c3 = coef[3];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float e = ec[i];
    if (eu) {
      const float u = eu[i];
      const float d = e - u;
      const float gd = g * d;
      e = u + gd;
    }
    const float s = c1 * e;
    const float num = x[i] - s;
    float x0 = num / c0;
    x0 = fminf(fmaxf(x0, -4.0f), 4.0f);
    const float a = c2 * x0;
    const float b = c3 * e;
    x[i] = (c2 < 0.f) ? x0 : a + b;
  }
}

// z = (mean + exp(0.5 * logvar) * noise) * scale : DiagonalGaussianDistribution.sample() followed by the
// latent scale (diffusers AutoencoderKL; src/models/diffusion_module_ip.py:410-411)
__global__ __launch_bounds__(256) void gaussian_sample_kernel(const float* __restrict__ mean,
                                                              const float* __restrict__ logvar,
                                                              const float* __restrict__ noise, float scale,
                                                              float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float lv = fminf(fmaxf(logvar[i], -30.f), 20.f);
    out[i] = (mean[i] + expf(0.5f * lv) * noise[i]) * scale;
  }
}

// frames fp32 NCHW in [0, 1] -> uint8 NHWC (what PIL / the BMP writers take): v * 255 truncated, exactly
// ``tensor.permute(1, 2, 0).mul(255).to(torch.uint8)`` of the reference's writers.  One thread = 4 pixels x 3 channels
// = three packed 32-bit stores of 12 contiguous bytes.
__global__ __launch_bounds__(256) void frames_u8_kernel(const float* __restrict__ x, uint8_t* __restrict__ out, int B,
                                                        int HW) {
  const size_t quads = (size_t)B * (HW / 4);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < quads; i += (size_t)gridDim.x * 256) {
    const size_t b = i / (HW / 4), q = i - b * (HW / 4);
    const float* src = x + b * 3 * HW + q * 4;
    const f4 r = *reinterpret_cast<const f4*>(src), g = *reinterpret_cast<const f4*>(src + HW),
             bl = *reinterpret_cast<const f4*>(src + 2 * (size_t)HW);
    uint8_t px[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      px[3 * k] = (uint8_t)(r[k] * 255.0f);
      px[3 * k + 1] = (uint8_t)(g[k] * 255.0f);
      px[3 * k + 2] = (uint8_t)(bl[k] * 255.0f);
    }
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + (b * HW + q * 4) * 3);
#pragma unroll
    for (int w = 0; w < 3; ++w)
      dst[w] = (uint32_t)px[4 * w] | ((uint32_t)px[4 * w + 1] << 8) | ((uint32_t)px[4 * w + 2] << 16) | ((uint32_t)px[4 * w + 3] << 24);
  }
}

// Forward diffusion (src/models/diffusion_module_ip.py:299-303): x_t = sqrt(ab[t_b]) x0 + sqrt(1 - ab[t_b]) noise,
// per-sample timestep; fp32, individually rounded ops in the reference's order.
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const int64_t* __restrict__ t, const float* __restrict__ ab,
                                                       float* __restrict__ out, int B, int64_t per) {
#pragma clang fp contract(off)
  const int64_t n = (int64_t)B * per;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float a = ab[t[i / per]];
    const float s0 = sqrtf(a), s1 = sqrtf(1.0f - a);
    const float u = s0 * x0[i];
    const float v = s1 * noise[i];
    out[i] = u + v;
  }
}

// base_loss[b] = mean over the sample of (pred - target)^2 (F.mse_loss(reduction="none").mean(dim=(1,2,3)),
// diffusion_module_ip.py:440-441): one block per sample, fixed-order tree (bit-reproducible).
__global__ __launch_bounds__(256) void mse_rows_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                       float* __restrict__ out, int64_t per) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* p = pred + (size_t)b * per;
  const float* q = target + (size_t)b * per;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < per; i += 256) {
    const float d = p[i] - q[i];
    acc += d * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[b] = (red[0] + red[1] + red[2] + red[3]) / (float)per;
}

}  // namespace

extern "C" int dadd_q_sample_f32(const float* x0, const float* noise, const int64_t* t, const float* alphas_cumprod,
                                 float* out, int B, int64_t per_sample, void* stream) {
  DADD_REQUIRE(x0 && noise && t && alphas_cumprod && out && B > 0 && per_sample > 0, "q_sample: bad arguments");
  const int64_t n = (int64_t)B * per_sample;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  dadd_launch({"q_sample_kernel", 0.0, (double)n * 12.0}, q_sample_kernel, dim3(blocks), dim3(256), 0,
              static_cast<hipStream_t>(stream), x0, noise, t, alphas_cumprod, out, B, per_sample);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_mse_rows_f32(const float* pred, const float* target, float* out, int B, int64_t per_sample,
                                 void* stream) {
  DADD_REQUIRE(pred && target && out && B > 0 && per_sample > 0, "mse_rows: bad arguments");
  dadd_launch({"mse_rows_kernel", 0.0, (double)B * per_sample * 8.0}, mse_rows_kernel, dim3(B), dim3(256), 0,
              static_cast<hipStream_t>(stream), pred, target, out, per_sample);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_frames_to_u8(const float* frames_nchw, void* out_nhwc_u8, int B, int H, int W, void* stream) {
  DADD_REQUIRE(frames_nchw && out_nhwc_u8 && B > 0 && H > 0 && W > 0 && (H * W) % 4 == 0,
               "frames_to_u8: H*W must be a multiple of 4");
  DADD_REQUIRE(dadd_aligned16(frames_nchw) && (((uintptr_t)out_nhwc_u8) & 3) == 0, "frames_to_u8: alignment");
  const size_t quads = (size_t)B * (H * W / 4);
  int blocks = (int)((quads + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  dadd_launch({"frames_u8_kernel", 0.0, (double)B * H * W * 15.0}, frames_u8_kernel, dim3(blocks), dim3(256), 0,
              static_cast<hipStream_t>(stream), frames_nchw, static_cast<uint8_t*>(out_nhwc_u8), B, H * W);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_gaussian_sample_f32(const float* mean, const float* logvar, const float* noise, float scale,
                                        float* out, int64_t n, void* stream) {
  DADD_REQUIRE(mean && logvar && noise && out && n > 0, "gaussian_sample: bad arguments");
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  dadd_launch({"gaussian_sample_kernel", 0.0, (double)n * 16.0}, gaussian_sample_kernel, dim3(blocks), dim3(256), 0,
              static_cast<hipStream_t>(stream), mean, logvar, noise, scale, out, n);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_pack_nchw_f32_to_nhwc8_f16(const float* x, void* out, int B, int C, int H, int W,
                                               float scale, const float* mat, const float* vec,
                                               void* stream) {
  DADD_REQUIRE(x && out, "pack: null pointer");
  DADD_REQUIRE(B > 0 && C > 0 && C <= 8 && H > 0 && W > 0, "pack: C must be in 1..8");
  DADD_REQUIRE(dadd_aligned16(out), "pack: out must be 16-byte aligned");
  const size_t total = (size_t)B * H * W;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  dadd_launch({"pack_kernel", 0.0, (double)total * (C * 4.0 + 16.0)}, pack_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     static_cast<half_t*>(out), B, C, H * W, scale, mat, vec);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_conv3x3_cin8_f16(const void* x, const void* w, const float* bias, void* out, int B,
                                     int H, int W, int Cout, void* stream) {
  DADD_REQUIRE(x && w && out, "conv_cin8: null pointer");
  DADD_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cout % 8 == 0, "conv_cin8: Cout must be x8");
  DADD_REQUIRE(dadd_aligned16(x) && dadd_aligned16(w) && dadd_aligned16(out),
               "conv_cin8: pointers must be 16-byte aligned");
  const int npix = B * H * W;
  dadd_launch({"conv_cin8_kernel", 2.0 * npix * Cout * 72.0, (double)npix * (16.0 + 2.0 * Cout)}, conv_cin8_kernel, dim3((npix + 255) / 256, Cout / 8), dim3(256), 0,
                     static_cast<hipStream_t>(stream), static_cast<const half_t*>(x),
                     static_cast<const half_t*>(w), bias, static_cast<half_t*>(out), B, H, W, Cout);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_conv_in_nchw_f16(const float* x_nchw, const void* w, const float* bias, void* out, int B, int C,
                                    int H, int W, int Cout, float* gn_ws, int gn_nchunk, void* stream) {
  DADD_REQUIRE(x_nchw && w && out, "conv_in_nchw: null pointer");
  DADD_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 1 && C <= 4 && Cout > 0 && Cout % 8 == 0,
               "conv_in_nchw: C must be 1..4 and Cout a multiple of 8");
  DADD_REQUIRE(dadd_aligned16(w) && dadd_aligned16(out), "conv_in_nchw: w / out must be 16-byte aligned");
  const int npix = B * H * W;
  if (gn_ws != nullptr) {     // GroupNorm chunk partials of the output: [B][gn_nchunk][32][2], 256 pixels per chunk
    DADD_REQUIRE(Cout == 320 && (H * W) % 256 == 0 && gn_nchunk == (H * W) / 256 && gn_nchunk <= DADD_GN_MAX_CHUNKS,
                 "conv_in_nchw: GroupNorm partials need Cout == 320, H*W %% 256 == 0 and gn_nchunk == H*W/256 (<= %d)",
                 DADD_GN_MAX_CHUNKS);
    dadd_launch({"conv_in_nchw_gn_kernel", 2.0 * npix * Cout * 9.0 * C, (double)npix * (4.0 * C + 2.0 * Cout)},
                conv_in_nchw_gn_kernel<10>, dim3(npix / 256, Cout / 40), dim3(256), 0, static_cast<hipStream_t>(stream),
                x_nchw, static_cast<const half_t*>(w), bias, static_cast<half_t*>(out), gn_ws, B, C, H, W, Cout);
    DADD_LAUNCH_CHECK();
    return DADD_OK;
  }
  dadd_launch({"conv_in_nchw_kernel", 2.0 * npix * Cout * 9.0 * C, (double)npix * (4.0 * C + 2.0 * Cout)}, conv_in_nchw_kernel,
              dim3((npix + 255) / 256, Cout / 8), dim3(256), 0, static_cast<hipStream_t>(stream), x_nchw,
              static_cast<const half_t*>(w), bias, static_cast<half_t*>(out), B, C, H, W, Cout);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

static int conv_cout4_launch(const void* x, const void* w, const float* bias, float* out_nchw, int B, int H, int W,
                             int C, int Cout, int mode, const float* coef, void* stream) {
  DADD_REQUIRE(x && w && out_nchw, "conv_cout4: null pointer");
  DADD_REQUIRE(B > 0 && H > 0 && W > 0 && C % 8 == 0 && Cout >= 1 && Cout <= 4,
               "conv_cout4: C must be x8 and Cout in 1..4");
  DADD_REQUIRE(dadd_aligned16(x) && dadd_aligned16(w), "conv_cout4: pointers must be 16-byte aligned");
  const int npix = B * H * W;
  const unsigned smem = (unsigned)(Cout * 9 * C * 2);
  DADD_REQUIRE(smem <= 160 * 1024 - 1024, "conv_cout4: %d channels do not fit the LDS weight stage", C);
  static unsigned smem_set = 64 * 1024;     // (no attribute call inside a stream capture for the usual sizes)
  if (smem > smem_set) {
    DADD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_cout4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)smem));
    smem_set = smem;
  }
  dadd_launch({"conv_cout4_kernel", 2.0 * npix * Cout * 9.0 * C, (double)npix * (2.0 * C + 4.0 * Cout)}, conv_cout4_kernel, dim3((npix + 31) / 32), dim3(256), smem,
                     static_cast<hipStream_t>(stream), static_cast<const half_t*>(x),
                     static_cast<const half_t*>(w), bias, out_nchw, B, H, W, C, Cout, mode, coef);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_conv3x3_cout4_f16(const void* x, const void* w, const float* bias, float* out_nchw,
                                      int B, int H, int W, int C, int Cout, int mode, void* stream) {
  DADD_REQUIRE(mode >= 0 && mode <= 2, "conv_cout4: mode must be 0, 1 or 2");
  return conv_cout4_launch(x, w, bias, out_nchw, B, H, W, C, Cout, mode, nullptr, stream);
}

extern "C" int dadd_conv_out_ddim_f16(const void* x, const void* w, const float* bias, float* latents,
                                      const float* coef, int B, int H, int W, int C, int Cout, void* stream) {
  DADD_REQUIRE(coef != nullptr, "conv_out_ddim: null coefficient row");
  return conv_cout4_launch(x, w, bias, latents, B, H, W, C, Cout, 3, coef, stream);
}

extern "C" int dadd_timestep_features_f32(const int64_t* t, float* out, int M, int dim, void* stream) {
  DADD_REQUIRE(t && out && M > 0 && dim > 0 && dim % 2 == 0, "timestep_features: bad arguments");
  const int n = M * dim / 2;
  dadd_launch({"timestep_features_kernel", 0.0, (double)M * dim * 4.0}, timestep_features_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     static_cast<hipStream_t>(stream), t, out, M, dim);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_linear_rows_f32(const float* x, const void* w, const float* bias, float* out, int M,
                                    int K, int N, int act_in, int act_out, int w_f32, void* stream) {
  DADD_REQUIRE(x && w && out, "linear_rows: null pointer");
  DADD_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0 && K <= 2048, "linear_rows: K must be x8, <=2048");
  DADD_REQUIRE(dadd_aligned16(w), "linear_rows: w must be 16-byte aligned");
  DADD_REQUIRE(act_in >= 0 && act_in <= 2 && act_out >= 0 && act_out <= 2, "linear_rows: act must be 0, 1 (SiLU) or 2 (GELU)");
  const DaddLaunchTag tag = {w_f32 ? "linear_rows_kernel<float>" : "linear_rows_kernel<_Float16>", 2.0 * M * N * K,
                             (w_f32 ? 4.0 : 2.0) * N * K + 4.0 * M * (K + N)};
  if (w_f32)
    dadd_launch(tag, linear_rows_kernel<float>, dim3((N + 3) / 4), dim3(256), (unsigned)(8 * K * sizeof(float)),
                static_cast<hipStream_t>(stream), x, static_cast<const float*>(w), bias, out, M, K, N, act_in, act_out);
  else
    dadd_launch(tag, linear_rows_kernel<half_t>, dim3((N + 3) / 4), dim3(256), (unsigned)(8 * K * sizeof(float)),
                static_cast<hipStream_t>(stream), x, static_cast<const half_t*>(w), bias, out, M, K, N, act_in, act_out);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_begin_step(const float* table, float* cur_rows, int B, int ncols, const float* coef,
                               float* cur_coef, int32_t* step, void* stream) {
  DADD_REQUIRE(table && cur_rows && coef && cur_coef && step && B > 0 && ncols > 0,
               "begin_step: bad arguments");
  dadd_launch({"begin_step_kernel", 0.0, 4.0 * ncols * (1.0 + B)}, begin_step_kernel, dim3((ncols + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                     table, cur_rows, B, ncols, coef, cur_coef, step);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_ddim_update_f32(float* x, const float* eps_c, const float* eps_u, float guidance,
                                    const float* guidance_dev, const float* coef, int64_t n, void* stream) {
  DADD_REQUIRE(x && eps_c && coef && n > 0, "ddim_update: bad arguments");
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  dadd_launch({"ddim_kernel", 0.0, (double)n * (eps_u ? 16.0 : 12.0)}, ddim_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                     eps_c, eps_u, guidance, guidance_dev, coef, n);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}
