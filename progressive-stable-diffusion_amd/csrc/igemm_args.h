// Kernel argument block shared by the two implicit-GEMM kernels (igemm.hip, igemm_dma.hip).
#pragma once
#include "dadd_common.h"

struct IgemmArgs {
  const half_t* x;
  const half_t* x2;
  const half_t* w;
  half_t* out;
  float* partial;
  const float* bias;
  const float* rowvec;
  const half_t* residual;
  int* counters;   // split-K tickets, one per output tile, zero between launches (nullptr: finish kernel)
  float* gn_ws;         // DADD_EPI_GNSTAT: GroupNorm chunk partials [B][gn_nchunk][32][2] written by the epilogue
  int gn_nchunk, gn_cg;  //   chunks per sample (= Ho*Wo / rows per MFMA wave), channels per group
  const float* ln_c1;   // DADD_EPI_LNFOLD: c1[n] = sum_k w[n][k] (w already carries the LayerNorm gamma)
  float ln_eps;
  float* ln_stats_out;        // DADD_EPI_LNSTAT: row partials of the OUTPUT, [N / WN][M][2] (WN = columns per MFMA wave)
  const float* ln_stats_in;   // DADD_EPI_LNFOLD with the statistics of x supplied by its producer: [ln_parts_in][M][2]
  int ln_parts_in;
  const float* gni_ws;        // DADD_PRE_GN: chunk partials of the input [B][gni_nchunk][32][2], affine, eps
  const float* gni_gamma;
  const float* gni_beta;
  int gni_nchunk;
  float gni_eps;
  const float* gni_ws2;       // ... of the second source (skip-concat): its own 32 groups over C2 channels
  int gni_nchunk2;
  half_t* gno_out;            // DADD_EPI_GNAPPLY: GroupNorm (+ SiLU) of the OUTPUT written beside it by the split-K finish
  const float* gno_gamma;
  const float* gno_beta;
  float gno_eps;
  int B, Hi, Wi, C1, C2, Ho, Wo, N;
  int taps, stride, ups, pad;
  int ldo, ldr, ld_rowvec;
  int splitk, flags;
  int M, K, nkt, kps, ntiles;
  int korder;           // LDS-DMA kernel only: 0 = channels fastest, 1 = taps fastest (3x3)
  int mtiles, gm, gn;   // tile order: groups of gm x gn tiles (one of them spans its whole dimension); gm == 0: n fastest
};

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (id % 8 shares an XCD, each
// with a private 4 MiB L2).  Remap so that one XCD owns a CONTIGUOUS range of logical tiles: row
// tiles that share halo rows and all column tiles of one row tile then hit the same L2, and the nine
// taps of a 3x3 window re-read the same ~1/8 of the activation from L2 instead of from the Infinity
// Cache.  Bijective for any grid size (cdna_hip_programming.md §5 "XCD swizzle must be bijective");
// placement only affects speed, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// Logical tile id -> (mt, nt).  An XCD owns a contiguous range of ids (xcd_remap), so the order of ids
// decides what each 4 MiB L2 sees.  Measured with FETCH_SIZE (profiles/r01_q_traffic.txt): with n fastest
// the ntiles column tiles of one row tile start together and each of them pulls the SAME activation
// lines through the fabric (GEGLU N=2560: 198 MB fetched for 12 MB of operands), and a weight matrix
// larger than the activation is re-read by all 8 XCDs.  So: the larger operand is the one split across
// XCDs (gm < mtiles: row tiles split, A-heavy;  gn < ntiles: column tiles split, W-heavy) and inside a
// group the row tile runs fastest, so concurrent workgroups read different activation rows and share
// one or two weight tiles.  Bijective for any sizes.
__device__ __forceinline__ void tile_decode(const IgemmArgs& p, int tile_id, int& mt, int& nt) {
  if (p.gm == 0) {
    nt = tile_id % p.ntiles;
    mt = tile_id / p.ntiles;
  } else if (p.gn >= p.ntiles) {          // groups of gm row tiles x all column tiles
    const int per = p.gm * p.ntiles;
    const int g = tile_id / per, idx = tile_id - g * per;
    const int base = g * p.gm;
    const int gsz = min(p.gm, p.mtiles - base);
    mt = base + idx % gsz;
    nt = idx / gsz;
  } else if (p.gm >= p.mtiles) {          // groups of all row tiles x gn column tiles
    const int per = p.mtiles * p.gn;
    const int g = tile_id / per, idx = tile_id - g * per;
    mt = idx % p.mtiles;
    nt = g * p.gn + idx / p.mtiles;
  } else {                                // 2-D groups of gm x gn tiles (the host guarantees gm | mtiles and gn | ntiles): an XCD
    const int per = p.gm * p.gn;          // owns a block of A rows that STAYS in its L2 while the block's weight tiles stream by
    const int g = tile_id / per, idx = tile_id - g * per;
    const int ngn = p.ntiles / p.gn;
    const int gmi = g / ngn, gni = g - gmi * ngn;
    mt = gmi * p.gm + idx % p.gm;
    nt = gni * p.gn + idx / p.gm;
  }
}

// algorithmic work of one implicit-GEMM launch (roofline columns of the profiling records)
static inline double dadd_igemm_flop(const IgemmArgs& a) { return 2.0 * (double)a.M * (double)a.N * (double)a.K; }
static inline double dadd_igemm_bytes(const IgemmArgs& a) {
  const double n_out = (a.flags & DADD_EPI_GEGLU) ? a.N / 2 : a.N;
  return 2.0 * ((double)a.B * a.Hi * a.Wi * (a.C1 + a.C2) + (double)a.N * a.K + (double)a.M * n_out +
                ((a.flags & DADD_EPI_RESIDUAL) ? (double)a.M * a.N : 0.0));
}

// ---- LayerNorm folded into the consuming linear (DADD_EPI_LNFOLD) -----------------------------------------------
//   LN(x) W^T + b = rstd_m * (x (gamma o W)^T - mu_m * c1) + (W beta + b),   c1[n] = sum_k gamma_k W[n][k]
// The caller passes gamma o W as the weight, c1 and the composed bias; the kernel needs mu_m and rstd_m of every
// row of its tile.  K = C for these linears, so the MFMA waves see whole rows pass through their A fragments: each
// lane accumulates sum x and sum x^2 of the 8-element pieces it reads anyway (v_dot2_f32_f16: exact products, fp32
// sums), and two xor-shuffles over the four k-quarters of a fragment finish the row — whose outputs that same lane
// owns in the swapped-MFMA layout.  No statistics pass, no normalised copy of the activation, no extra launch.
typedef _Float16 dadd_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ln_acc_pair(const h8& x, int p, float& s1, float& s2) {
  const dadd_h2 v = {x[2 * p], x[2 * p + 1]};
  const dadd_h2 one = {(_Float16)1.0f, (_Float16)1.0f};
  s1 = __builtin_amdgcn_fdot2(v, one, s1, false);
  s2 = __builtin_amdgcn_fdot2(v, v, s2, false);
}
template <int MI>
__device__ __forceinline__ void ln_finish(float (&s1)[MI], float (&s2)[MI], int K, float eps) {   // -> mu, rstd
  const float inv = 1.0f / (float)K;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    float a = s1[i], q = s2[i];
    a = dadd_sum_x16x32(a);
    q = dadd_sum_x16x32(q);
    const float mu = a * inv;
    const float var = fmaxf(q * inv - mu * mu, 0.f);
    s1[i] = mu;
    s2[i] = rsqrtf(var + eps);
  }
}

int dadd_init_igemm_dma();
int dadd_launch_igemm_dma(const IgemmArgs& a, int tile_m, int tile_n, int nsplit, hipStream_t s);
bool dadd_igemm_dma_persistent(const IgemmArgs& a, int nsplit);
int dadd_init_conv_halo();
bool dadd_conv_halo_applicable(const IgemmArgs& a, int tile_n);
int dadd_conv_halo_gn_channels(int Wo);
int dadd_launch_conv_halo(const IgemmArgs& a, int nsplit, hipStream_t s);   // a.kps = chunks per K slice
