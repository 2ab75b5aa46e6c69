// Small kernels of the conditioning front-end (CLIP patch rows, AOE class interpolation, FeaturePurifier tail).
// The matrix work of the front-end runs on the implicit-GEMM and attention kernels; these are the pieces around it.
#include "dadd_common.h"

namespace {

// pixel_values [B][3][H][W] fp32 -> rows [B][1 + gh*gw][Kp] fp16 for the patch-embedding GEMM: row 0 of a sample
// (the class-token slot) is zero, row 1 + py*gw + px holds the patch in conv-weight order k = (c*P + ky)*P + kx,
// zero padded to Kp.  Replaces the unfold of nn.Conv2d(3, hidden, P, stride=P) in CLIPVisionEmbeddings.
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ px, half_t* __restrict__ out, int B,
                                                       int H, int W, int P, int Kp) {
  const int gw = W / P, gh = H / P, T = 1 + gh * gw, K = 3 * P * P;
  const size_t total = (size_t)B * T * Kp;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % Kp);
    const size_t row = i / Kp;
    const int t = (int)(row % T), b = (int)(row / T);
    float v = 0.f;
    if (t > 0 && k < K) {
      const int c = k / (P * P), r = k - c * P * P, ky = r / P, kx = r - ky * P;
      const int py = (t - 1) / gw, pxi = (t - 1) - py * gw;
      v = px[(((size_t)b * 3 + c) * H + py * P + ky) * W + pxi * P + kx];
    }
    out[i] = (half_t)v;
  }
}

// AdditiveOrdinalEmbedder class interpolation (src/models/ordinal_embedder.py:129-182): table[c] = base +
// cumsum(deltas)[c-1]; label clamped to [0, C-1]; out[b] = table[lo] (1 - frac) + table[hi] frac.  fp32.
__global__ __launch_bounds__(256) void aoe_interp_kernel(const float* __restrict__ labels,
                                                         const float* __restrict__ base,
                                                         const float* __restrict__ deltas, float* __restrict__ out,
                                                         int B, int D, int classes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, dcol = i - b * D;
  const float top = (float)(classes - 1);
  const float y = fminf(fmaxf(labels[b], 0.f), top);
  const float lo = floorf(y);
  const float frac = y - lo;
  const int lo_i = (int)lo, hi_i = min(lo_i + 1, classes - 1);
  float t = base[dcol], tlo = 0.f, thi = 0.f;
  for (int c = 0; c < classes; ++c) {          // running cumulative sum in the reference's order
    if (c > 0) t += deltas[(size_t)(c - 1) * D + dcol];
    if (c == lo_i) tlo = t;
    if (c == hi_i) thi = t;
  }
  out[i] = tlo * (1.0f - frac) + thi * frac;
}

// FeaturePurifier tail (src/models/feature_purifier.py:88-95): out = LayerNorm(img - gate * dis), one wave per row,
// gate = already-sigmoided fp16 output of the gate MLP.  img / dis / gate fp16 [M][C], out fp32 [M][C].
constexpr int PT_MAXV = 4;   // C <= 8*64*4 = 2048
__global__ __launch_bounds__(256) void purifier_tail_kernel(const half_t* __restrict__ img,
                                                            const half_t* __restrict__ dis,
                                                            const half_t* __restrict__ gate,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ out,
                                                            int M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nvec = C >> 3;
  float v[PT_MAXV][8];
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < PT_MAXV; ++u) {
    const int i = lane + 64 * u;
    if (i < nvec) {
      const h8 a = *reinterpret_cast<const h8*>(img + (size_t)row * C + i * 8);
      const h8 d = *reinterpret_cast<const h8*>(dis + (size_t)row * C + i * 8);
      const h8 g = *reinterpret_cast<const h8*>(gate + (size_t)row * C + i * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[u][e] = (float)a[e] - (float)g[e] * (float)d[e];
        sum += v[u][e];
      }
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int u = 0; u < PT_MAXV; ++u)
    if (lane + 64 * u < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[u][e] - mean;
        sq += d * d;
      }
    }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int u = 0; u < PT_MAXV; ++u) {
    const int i = lane + 64 * u;
    if (i < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        out[(size_t)row * C + i * 8 + e] = (v[u][e] - mean) * rstd * gamma[i * 8 + e] + beta[i * 8 + e];
    }
  }
}

}  // namespace

extern "C" int dadd_clip_patch_rows_f16(const float* pixels, void* out, int B, int H, int W, int patch, int Kp,
                                        void* stream) {
  DADD_REQUIRE(pixels && out && B > 0 && patch > 0 && H % patch == 0 && W % patch == 0,
               "clip_patch_rows: image side must be a multiple of the patch size");
  DADD_REQUIRE(Kp >= 3 * patch * patch && Kp % 64 == 0, "clip_patch_rows: Kp must be a multiple of 64 >= 3*P*P");
  const size_t total = (size_t)B * (1 + (H / patch) * (W / patch)) * Kp;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  dadd_launch({"patchify_kernel", 0.0, (double)B * 3 * H * W * 4.0 + (double)total * 2.0}, patchify_kernel, dim3(blocks),
              dim3(256), 0, static_cast<hipStream_t>(stream), pixels, static_cast<half_t*>(out), B, H, W, patch, Kp);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_aoe_interp_f32(const float* labels, const float* base, const float* deltas, float* out, int B,
                                   int D, int classes, void* stream) {
  DADD_REQUIRE(labels && base && deltas && out && B > 0 && D > 0 && classes >= 2, "aoe_interp: bad arguments");
  dadd_launch({"aoe_interp_kernel", 0.0, (double)B * D * 4.0 + (double)classes * D * 4.0}, aoe_interp_kernel,
              dim3((B * D + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), labels, base, deltas, out, B, D,
              classes);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}

extern "C" int dadd_purifier_tail_f16(const void* img, const void* dis, const void* gate, const float* gamma,
                                      const float* beta, float* out, int M, int C, float eps, void* stream) {
  DADD_REQUIRE(img && dis && gate && gamma && beta && out, "purifier_tail: null pointer");
  DADD_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 8 * 64 * PT_MAXV, "purifier_tail: C=%d must be x8 and <= %d", C,
               8 * 64 * PT_MAXV);
  DADD_REQUIRE(dadd_aligned16(img) && dadd_aligned16(dis) && dadd_aligned16(gate),
               "purifier_tail: pointers must be 16-byte aligned");
  dadd_launch({"purifier_tail_kernel", 0.0, (double)M * C * 10.0}, purifier_tail_kernel, dim3((M + 3) / 4), dim3(256), 0,
              static_cast<hipStream_t>(stream), static_cast<const half_t*>(img), static_cast<const half_t*>(dis),
              static_cast<const half_t*>(gate), gamma, beta, out, M, C, eps);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}
