// Epilogue shared by the two implicit-GEMM kernels: bias / time-embedding row / residual / GEGLU,
// fp16 stores of 4 consecutive channels per lane, and the split-K combine.
//
// Split-K.  Every K slice writes its fp32 slab; with `counters` the combine happens IN the launch:
// the slice that draws the last ticket of a tile re-reads all slabs in slice order (so the sum is
// bit-reproducible) and runs the epilogue.  The hand-off follows cdna_hip_programming.md §6
// Guideline 16 in its counter form: plain slab stores -> every wave s_waitcnt vmcnt(0) -> barrier ->
// one lane: agent-scope release fence, asm vmcnt(0), relaxed agent fetch_add; the last arriver does
// one agent-scope acquire, asm vmcnt(0), resets the ticket for the next launch, and a barrier
// releases the other waves to plain loads.  No spinning, no residency assumption, results do not
// depend on block placement.  Without `counters` a separate finish kernel combines the slabs.
#pragma once
#include "igemm_args.h"
#include "ln_lds.h"

// sum over the 16 lanes of a DPP row (all 16 end up with the total): xor 1, xor 2, half-row mirror, row mirror
__device__ __forceinline__ float dadd_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}

// `ln_mu` / `ln_rs` (MI values each, or nullptr): row mean and 1/std of the folded LayerNorm (igemm_args.h).
// `gn_scratch`: 4 KB of LDS (4 MFMA waves x [80 columns][2] floats) that no LDS-DMA can still be writing when the
// epilogue runs — the staging area of the GroupNorm statistics (DADD_EPI_GNSTAT).
template <int J, int MI, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& p, f4 (&acc)[J][MI], int m0, int n0,
                                               int wm, int wn, int lane, int z, char* smem,
                                               const float* ln_mu = nullptr, const float* ln_rs = nullptr,
                                               char* gn_scratch = nullptr, const char* ln_lds = nullptr,
                                               const char* ln_cb = nullptr) {
  const int g = lane >> 4, mc = lane & 15;
  const int HoWo = p.Ho * p.Wo;
  if (p.splitk > 1) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * WM + i * 16 + mc;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int n = n0 + wn * WN + j * 16 + g * 4;
        if (n < p.N) *reinterpret_cast<f4*>(p.partial + ((size_t)z * p.M + m) * p.N + n) = acc[j][i];
      }
    }
    if (p.counters == nullptr) return;   // combined by splitk_finish_kernel
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                     // every wave's slab stores are issued and drained; LDS is free
    volatile int* flag = reinterpret_cast<volatile int*>(smem);
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int* ticket = p.counters + blockIdx.x;
      const int prev = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = (prev == p.splitk - 1) ? 1 : 0;
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      }
      *flag = last;
    }
    __syncthreads();
    if (*flag == 0) return;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * WM + i * 16 + mc;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int n = n0 + wn * WN + j * 16 + g * 4;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (m < p.M && n < p.N)
          for (int s = 0; s < p.splitk; ++s)
            v += *reinterpret_cast<const f4*>(p.partial + ((size_t)s * p.M + m) * p.N + n);
        acc[j][i] = v;
      }
    }
  }
  if (p.flags & DADD_EPI_GNSTAT) {
    // ---- GroupNorm statistics of the OUTPUT from this epilogue: the consumer's gn_stats pass — a full
    // read of the tensor and a launch — disappears.  Host contract: full tiles, Ho*Wo % WM == 0 (a wave's rows lie in
    // one sample), WN % cg == 0 and tile origin aligned to cg (a wave's columns hold whole groups), no split-K.
    // Column-block outer / row-fragment inner, so only one block's partial sums are live beside the accumulators.
    // Sums are taken over the ROUNDED fp16 values (what the consumer reads), in a fixed order: bit-reproducible.
    float* scratch = reinterpret_cast<float*>(gn_scratch) + (wm * 2 + wn) * 256;   // [WN <= 80][2] per wave
    const int mrow0 = m0 + wm * WM;
    const int bsmp = mrow0 / HoWo;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int n = n0 + wn * WN + j * 16 + g * 4;
      f4 bias4 = {0.f, 0.f, 0.f, 0.f};
      if (p.flags & DADD_EPI_BIAS) bias4 = *reinterpret_cast<const f4*>(p.bias + n);
      if (p.flags & DADD_EPI_ROWVEC) bias4 += *reinterpret_cast<const f4*>(p.rowvec + (size_t)bsmp * p.ld_rowvec + n);
      float cs[4] = {0.f, 0.f, 0.f, 0.f}, cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = mrow0 + i * 16 + mc;
        f4 v = acc[j][i] + bias4;
        if (p.flags & DADD_EPI_RESIDUAL) {
          const h4 rv = *reinterpret_cast<const h4*>(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
        }
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = (half_t)v[r];
          const float f = (float)o[r];
          cs[r] += f;
          cq[r] = fmaf(f, f, cq[r]);
        }
        *reinterpret_cast<h4*>(p.out + (size_t)m * p.ldo + n) = o;
      }
      // 16 lanes (mc) share a column quad: butterfly inside the DPP row (quad perms, half mirror, mirror)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        cs[r] = dadd_row16_sum(cs[r]);
        cq[r] = dadd_row16_sum(cq[r]);
      }
      if (mc == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          scratch[(j * 16 + g * 4 + r) * 2] = cs[r];
          scratch[(j * 16 + g * 4 + r) * 2 + 1] = cq[r];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int ngrp = WN / p.gn_cg;
    if (lane < ngrp) {
      float a = 0.f, q = 0.f;
      for (int c = 0; c < p.gn_cg; ++c) {
        a += scratch[(lane * p.gn_cg + c) * 2];
        q += scratch[(lane * p.gn_cg + c) * 2 + 1];
      }
      const int chunk = (mrow0 - bsmp * HoWo) / WM;
      const int grp = (n0 + wn * WN) / p.gn_cg + lane;
      float* w = p.gn_ws + (((size_t)bsmp * p.gn_nchunk + chunk) * 32 + grp) * 2;
      w[0] = a;
      w[1] = q;
    }
    return;
  }
  // ---- folded LayerNorm.  Row mean / rstd either come from the caller (the LDS-DMA kernel summed the rows of its A
  // fragments) or from the GEMM that produced x (its DADD_EPI_LNSTAT): ln_parts_in float2 partials per row, lanes of
  // a 16-lane row read 16 consecutive rows (128 B); four parts x MI rows are in flight together.
  // (plain register arrays, filled by fully unrolled code: selecting between two arrays through a pointer would put
  // them in scratch)
  float lmu[MI], lrs[MI];
  bool fold = false;
  if (ln_mu != nullptr) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      lmu[i] = ln_mu[i];
      lrs[i] = ln_rs[i];
    }
    fold = true;
  } else if ((p.flags & DADD_EPI_LNFOLD) && p.ln_stats_in != nullptr) {
    const dadd_f2* st = reinterpret_cast<const dadd_f2*>(p.ln_stats_in);
    float sa[MI], sq[MI];
    int mrow[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      sa[i] = sq[i] = 0.f;
      mrow[i] = min(m0 + wm * WM + i * 16 + mc, p.M - 1);
    }
    if (ln_lds != nullptr) {            // staged in LDS by ln_lds_issue(): [part][tile row][2]
      for (int pp = 0; pp < p.ln_parts_in; ++pp)
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const dadd_f2 v = *reinterpret_cast<const dadd_f2*>(ln_lds + pp * 1024 + (wm * WM + i * 16 + mc) * 8);
          sa[i] += v[0];
          sq[i] += v[1];
        }
    } else
    for (int pp = 0; pp < p.ln_parts_in; pp += 4) {
      dadd_f2 v[4][MI];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < MI; ++i)
          v[u][i] = st[(size_t)min(pp + u, p.ln_parts_in - 1) * p.M + mrow[i]];   // unconditional: no branch, no wait per load
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float keep = (pp + u < p.ln_parts_in) ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          sa[i] = fmaf(keep, v[u][i][0], sa[i]);
          sq[i] = fmaf(keep, v[u][i][1], sq[i]);
        }
      }
    }
    const float inv = __builtin_amdgcn_rcpf((float)p.K);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const float mu = sa[i] * inv;
      lmu[i] = mu;
      lrs[i] = rsqrtf(fmaxf(sq[i] * inv - mu * mu, 0.f) + p.ln_eps);
    }
    fold = true;
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i) lmu[i] = 0.f, lrs[i] = 1.f;
  }
  if (fold && ln_cb != nullptr) {
    // ---- folded LayerNorm, c1 and the composed bias staged in LDS: row outer / column block inner, the store order of
    // the plain epilogue (the 32-byte pieces of one output row leave back to back and combine into full lines; the
    // column-outer order below measured +6 us on the 128x160 qkv tile of the persistent ring).
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = m0 + wm * WM + i * 16 + mc;
      if (m >= p.M) continue;
      if (p.flags & DADD_EPI_GEGLU) {
        if constexpr (J == 4) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int ch = wn * WN + j * 16 + g * 4, cg2 = ch + 32;      // tile-relative columns: hidden, gate
            if (n0 + cg2 >= p.N) continue;
            f4 hv = lrs[i] * (acc[j][i] - lmu[i] * *reinterpret_cast<const f4*>(ln_cb + ch * 4));
            f4 gv = lrs[i] * (acc[j + 2][i] - lmu[i] * *reinterpret_cast<const f4*>(ln_cb + cg2 * 4));
            if (p.flags & DADD_EPI_BIAS) {
              hv += *reinterpret_cast<const f4*>(ln_cb + 1024 + ch * 4);
              gv += *reinterpret_cast<const f4*>(ln_cb + 1024 + cg2 * 4);
            }
            const int no = (n0 >> 1) + wn * 32 + j * 16 + g * 4;
            const dadd_f2 g01 = dadd_gelu2(dadd_f2{gv[0], gv[1]}), g23 = dadd_gelu2(dadd_f2{gv[2], gv[3]});
            const h4 o = {(half_t)(hv[0] * g01[0]), (half_t)(hv[1] * g01[1]), (half_t)(hv[2] * g23[0]),
                          (half_t)(hv[3] * g23[1])};
            *reinterpret_cast<h4*>(p.out + (size_t)m * p.ldo + no) = o;
          }
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int c = wn * WN + j * 16 + g * 4, n = n0 + c;
        if (n >= p.N) continue;
        f4 v = lrs[i] * (acc[j][i] - lmu[i] * *reinterpret_cast<const f4*>(ln_cb + c * 4));
        if (p.flags & DADD_EPI_BIAS) v += *reinterpret_cast<const f4*>(ln_cb + 1024 + c * 4);
        if (p.flags & DADD_EPI_ROWVEC) v += *reinterpret_cast<const f4*>(p.rowvec + (size_t)(m / HoWo) * p.ld_rowvec + n);
        if (p.flags & DADD_EPI_ACT_MASK) {
          if (p.flags & DADD_EPI_QUICKGELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + __expf(-1.702f * v[r]));
          } else if (p.flags & DADD_EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = dadd_gelu(v[r]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = 1.0f / (1.0f + __expf(-v[r]));
          }
        }
        if (p.flags & DADD_EPI_RESIDUAL) {
          const h4 rv = *reinterpret_cast<const h4*>(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
        }
        const h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<h4*>(p.out + (size_t)m * p.ldo + n) = o;
      }
    }
    return;
  }
  if (fold) {
    // ---- folded LayerNorm, operands from global memory (64-row tiles, whose MFMA waves wait on the LDS fill anyway):
    // column block outer, rows inner — c1 and the composed bias of a column quad are loaded once (inside the row
    // loop every load is followed by its own wait: measured ~9 us per 128x160 tile).  These linears (qkv, attn2.to_q, the GEGLU projection, the CLIP / resampler
    // blocks) have no row partials to write.
    if (p.flags & DADD_EPI_GEGLU) {
      if constexpr (J == 4) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int nh = n0 + wn * WN + j * 16 + g * 4;  // physical (interleaved) weight rows; the gate is column block j + 2
          const int ng = nh + 32;
          if (ng >= p.N) continue;
          const f4 c1h = *reinterpret_cast<const f4*>(p.ln_c1 + nh), c1g = *reinterpret_cast<const f4*>(p.ln_c1 + ng);
          f4 bh = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
          if (p.flags & DADD_EPI_BIAS) {
            bh = *reinterpret_cast<const f4*>(p.bias + nh);
            bg = *reinterpret_cast<const f4*>(p.bias + ng);
          }
          const int no = (n0 >> 1) + wn * 32 + j * 16 + g * 4;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int m = m0 + wm * WM + i * 16 + mc;
            if (m >= p.M) continue;
            const f4 hv = lrs[i] * (acc[j][i] - lmu[i] * c1h) + bh;
            const f4 gv = lrs[i] * (acc[j + 2][i] - lmu[i] * c1g) + bg;
            const dadd_f2 g01 = dadd_gelu2(dadd_f2{gv[0], gv[1]}), g23 = dadd_gelu2(dadd_f2{gv[2], gv[3]});
            const h4 o = {(half_t)(hv[0] * g01[0]), (half_t)(hv[1] * g01[1]), (half_t)(hv[2] * g23[0]),
                          (half_t)(hv[3] * g23[1])};
            *reinterpret_cast<h4*>(p.out + (size_t)m * p.ldo + no) = o;
          }
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int n = n0 + wn * WN + j * 16 + g * 4;
      if (n >= p.N) continue;
      const f4 c4 = *reinterpret_cast<const f4*>(p.ln_c1 + n);
      f4 b4 = {0.f, 0.f, 0.f, 0.f};
      if (p.flags & DADD_EPI_BIAS) b4 = *reinterpret_cast<const f4*>(p.bias + n);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int m = m0 + wm * WM + i * 16 + mc;
        if (m >= p.M) continue;
        f4 v = lrs[i] * (acc[j][i] - lmu[i] * c4) + b4;
        if (p.flags & DADD_EPI_ROWVEC) v += *reinterpret_cast<const f4*>(p.rowvec + (size_t)(m / HoWo) * p.ld_rowvec + n);
        if (p.flags & DADD_EPI_ACT_MASK) {
          if (p.flags & DADD_EPI_QUICKGELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + __expf(-1.702f * v[r]));
          } else if (p.flags & DADD_EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = dadd_gelu(v[r]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = 1.0f / (1.0f + __expf(-v[r]));
          }
        }
        if (p.flags & DADD_EPI_RESIDUAL) {
          const h4 rv = *reinterpret_cast<const h4*>(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
        }
        const h4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<h4*>(p.out + (size_t)m * p.ldo + n) = o;
      }
    }
    return;
  }
  const bool lnstat = (p.flags & DADD_EPI_LNSTAT) != 0 && n0 + wn * WN < p.N;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * WM + i * 16 + mc;
    if (m >= p.M) continue;
    const int b = m / HoWo;
    float rs1 = 0.f, rs2 = 0.f;      // DADD_EPI_LNSTAT: this lane's share of row m (J x 4 columns)
    if (p.flags & DADD_EPI_GEGLU) {
      if constexpr (J == 4) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int nh = n0 + wn * WN + j * 16 + g * 4;  // physical (interleaved) weight rows; the gate is column block j + 2
          const int ng = nh + 32;
          if (ng >= p.N) continue;
          f4 hv = acc[j][i], gv = acc[j + 2][i];
          if (p.flags & DADD_EPI_BIAS) {
            hv += *reinterpret_cast<const f4*>(p.bias + nh);
            gv += *reinterpret_cast<const f4*>(p.bias + ng);
          }
          const int no = (n0 >> 1) + wn * 32 + j * 16 + g * 4;
          const dadd_f2 g01 = dadd_gelu2(dadd_f2{gv[0], gv[1]}), g23 = dadd_gelu2(dadd_f2{gv[2], gv[3]});
          const h4 o = {(half_t)(hv[0] * g01[0]), (half_t)(hv[1] * g01[1]), (half_t)(hv[2] * g23[0]),
                        (half_t)(hv[3] * g23[1])};
          *reinterpret_cast<h4*>(p.out + (size_t)m * p.ldo + no) = o;
        }
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int n = n0 + wn * WN + j * 16 + g * 4;
      if (n >= p.N) continue;
      f4 v = acc[j][i];
      if (p.flags & DADD_EPI_BIAS) v += *reinterpret_cast<const f4*>(p.bias + n);
      if (p.flags & DADD_EPI_ROWVEC) v += *reinterpret_cast<const f4*>(p.rowvec + (size_t)b * p.ld_rowvec + n);
      if (p.flags & DADD_EPI_ACT_MASK) {      // activation of the linear itself, before any residual
        if (p.flags & DADD_EPI_QUICKGELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] / (1.0f + __expf(-1.702f * v[r]));
        } else if (p.flags & DADD_EPI_GELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = dadd_gelu(v[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = 1.0f / (1.0f + __expf(-v[r]));
        }
      }
      if (p.flags & DADD_EPI_RESIDUAL) {
        const h4 rv = *reinterpret_cast<const h4*>(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
      }
      h4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = (half_t)v[r];
        const float f = (float)o[r];      // statistics of the ROUNDED values, what the consumer multiplies
        rs1 += f;
        rs2 = fmaf(f, f, rs2);
      }
      *reinterpret_cast<h4*>(p.out + (size_t)m * p.ldo + n) = o;
    }
    if (lnstat) {     // wave-uniform.  The four lanes (lane & 15) + 16 g hold the WN columns of row m between them.
      rs1 = dadd_sum_x16x32(rs1);
      rs2 = dadd_sum_x16x32(rs2);
      if (g == 0) {
        const int part = (n0 + wn * WN) / WN;
        reinterpret_cast<dadd_f2*>(p.ln_stats_out)[(size_t)part * p.M + m] = dadd_f2{rs1, rs2};
      }
    }
  }
}
