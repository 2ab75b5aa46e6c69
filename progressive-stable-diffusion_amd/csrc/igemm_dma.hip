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

namespace {

constexpr int BM = 128;
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

template <int BN>
__global__ __launch_bounds__(256, 1) void igemm_dma_kernel(const IgemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type exists only in device code; the host
                                      // pass needs just the launch stub of this signature
  constexpr int NBUF = 4;
  constexpr int WN = BN / 2;
  constexpr int J = WN / 16;
  constexpr int NA = BM / 32;   // DMA instructions per wave per tile, activations (4)
  constexpr int NBJ = BN / 32;  // DMA instructions per wave per tile, weights (4 or 5)
  constexpr int LPT = NA + NBJ;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_id = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = tile_id % p.ntiles, mt = tile_id / p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;
  const int z = blockIdx.y;
  const int kt0 = z * p.kps;
  const int kt1 = min(p.nkt, kt0 + p.kps);
  const int nk = kt1 - kt0;
  const int Cin = p.C1 + p.C2;
  const int Hv = p.ups ? 2 * p.Hi : p.Hi;
  const int Wv = p.ups ? 2 * p.Wi : p.Wi;
  const int HoWo = p.Ho * p.Wo;
  const int lrow = lane >> 3, lch = lane & 7;

  // Buffer descriptors.  For the affine gathers (everything but the 2x-upsample) the "-pad" of the
  // window origin is folded into the descriptor base, so per-lane offsets and the per-tap scalar
  // offset are both non-negative; lanes whose tap falls outside the image get voffset = OOB.
  const int shift1 = p.ups ? 0 : (p.pad * p.Wi + p.pad) * p.C1;
  const int shift2 = p.ups ? 0 : (p.pad * p.Wi + p.pad) * p.C2;
  const size_t pix_total = (size_t)p.B * p.Hi * p.Wi;
  const auto rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.x - shift1), 0, (int)((pix_total * p.C1 + shift1) * 2), 0x00020000);
  const auto rsrc2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((p.x2 ? p.x2 : p.x) - shift2), 0, (int)((pix_total * p.C2 + shift2) * 2), 0x00020000);
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((size_t)p.N * p.K * 2),
                                                       0x00020000);

  // per-lane gather state: byte offsets of the window origin in both sources, validity bit per tap
  unsigned a_v1[NA], a_v2[NA], a_mask[NA];
  int a_pix[NA], a_y[NA], a_x[NA], a_cc[NA];   // only the upsample path needs these per tile
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
  unsigned w_v[NBJ];
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int row = (j * 4 + wave) * 8 + lrow;
    const int n = n0 + row;
    w_v[j] = (n < p.N) ? (unsigned)(((size_t)n * p.K + (lch ^ ((row >> 1) & 7)) * 8) * 2) : OOB;
  }

  auto issue = [&](int kt, int slot) {   // LPT DMA loads for K tile kt into ring slot `slot`
    const int kk = kt * BK;
    const int tap = kk / Cin;
    const int c = kk - tap * Cin;
    const int ky = (p.taps == 9) ? tap / 3 : 0;
    const int kx = (p.taps == 9) ? tap - 3 * ky : 0;
    const bool second = c >= p.C1;
    const int cs = second ? p.C2 : p.C1;
    const int cb = second ? c - p.C1 : c;
    char* sa = smem + slot * STAGE + wave * 1024;
    if (!p.ups) {
      const unsigned soff = (unsigned)(((ky * p.Wi + kx) * cs + cb) * 2);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool ok = (a_mask[i] >> tap) & 1u;
        if (second)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc2, (lptr_t)(sa + i * 4096), 16,
                                                   ok ? a_v2[i] : OOB, soff, 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (lptr_t)(sa + i * 4096), 16,
                                                   ok ? a_v1[i] : OOB, soff, 0, 0);
      }
    } else {   // nearest-2x upsample: the source pixel is (iy>>1, ix>>1) of the virtual image
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int iy = a_y[i] + ky, ix = a_x[i] + kx;
        const bool ok = (iy >= 0) & (iy < Hv) & (ix >= 0) & (ix < Wv);
        const unsigned vo = (unsigned)(((a_pix[i] + (iy >> 1) * p.Wi + (ix >> 1)) * cs + a_cc[i]) * 2);
        if (second)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc2, (lptr_t)(sa + i * 4096), 16, ok ? vo : OOB,
                                                   (unsigned)(cb * 2), 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (lptr_t)(sa + i * 4096), 16, ok ? vo : OOB,
                                                   (unsigned)(cb * 2), 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NBJ; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lptr_t)(sa + A_BYTES + j * 4096), 16, w_v[j],
                                               (unsigned)(kk * 2), 0, 0);
  };

  f4 acc[J][4];
#pragma unroll
  for (int j = 0; j < J; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j][i] = f4{0.f, 0.f, 0.f, 0.f};

  // fragment reads of one 32-deep K step (s = 0,1) of a ring slot, and the MFMAs that consume them
  const int frow_a = wm * 64 + (lane & 15), frow_b = wn * WN + (lane & 15), fq = lane >> 4;
  auto read_frags = [&](int slot, int s, h8 (&xa)[4], h8 (&wb)[J]) {
    const half_t* a = reinterpret_cast<const half_t*>(smem + slot * STAGE);
    const half_t* b = reinterpret_cast<const half_t*>(smem + slot * STAGE + A_BYTES);
    const int chunk = s * 4 + fq;
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[i] = *reinterpret_cast<const h8*>(a + lds_off(frow_a + i * 16, chunk));
#pragma unroll
    for (int j = 0; j < J; ++j) wb[j] = *reinterpret_cast<const h8*>(b + lds_off(frow_b + j * 16, chunk));
  };
  auto mma = [&](const h8 (&xa)[4], const h8 (&wb)[J]) {
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[j], xa[i], acc[j][i], 0, 0, 0);
  };

  // ---- ring schedule.  Roles at iteration `it`: tile it computes, tile it+1 has LANDED (its first
  // fragments are prefetched while tile it's second half runs), tiles it+2, it+3 are in flight, and
  // the slot of tile it-1 (fully read before this iteration's barrier) is the one being refilled.
#pragma unroll
  for (int d = 0; d < 3; ++d)
    if (d < nk) issue(kt0 + d, d);
  if (nk >= 3) wait_vmcnt<LPT>(); else wait_vmcnt<0>();   // tiles 0 and 1 landed (this wave's share)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  h8 xa0[4], wb0[J], xa1[4], wb1[J];
  if (nk > 0) read_frags(0, 0, xa0, wb0);
  for (int it = 0; it < nk; ++it) {
    if (it > 0) {
      if (it + 2 <= nk - 1) wait_vmcnt<LPT>(); else wait_vmcnt<0>();   // tile it+1 landed
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (it + 3 < nk) issue(kt0 + it + 3, (it + 3) & 3);
    read_frags(it & 3, 1, xa1, wb1);
    mma(xa0, wb0);
    if (it + 1 < nk) read_frags((it + 1) & 3, 0, xa0, wb0);
    mma(xa1, wb1);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue (identical to igemm.hip)
  const int g = lane >> 4, mc = lane & 15;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + mc;
    if (m >= p.M) continue;
    const int b = m / HoWo;
    if (p.splitk > 1) {
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int n = n0 + wn * WN + j * 16 + g * 4;
        if (n < p.N) *reinterpret_cast<f4*>(p.partial + ((size_t)z * p.M + m) * p.N + n) = acc[j][i];
      }
      continue;
    }
    if (p.flags & DADD_EPI_GEGLU) {
      if constexpr (J == 4) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int nh = n0 + wn * WN + j * 16 + g * 4;
          const int ng = nh + 32;
          if (ng >= p.N) continue;
          f4 hv = acc[j][i], gv = acc[j + 2][i];
          if (p.flags & DADD_EPI_BIAS) {
            hv += *reinterpret_cast<const f4*>(p.bias + nh);
            gv += *reinterpret_cast<const f4*>(p.bias + ng);
          }
          const int no = (n0 >> 1) + wn * 32 + j * 16 + g * 4;
          h4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (half_t)(hv[r] * dadd_gelu(gv[r]));
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
      if (p.flags & DADD_EPI_ROWVEC)
        v += *reinterpret_cast<const f4*>(p.rowvec + (size_t)b * p.ld_rowvec + n);
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
#endif
}

template <int BN>
constexpr int smem_bytes() { return 4 * (BM + BN) * BK * 2; }

template <int BN>
int set_attr() {
  DADD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_dma_kernel<BN>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes<BN>()));
  return DADD_OK;
}

}  // namespace

int dadd_init_igemm_dma() {
  int rc = set_attr<128>();
  return rc != DADD_OK ? rc : set_attr<160>();
}

int dadd_launch_igemm_dma(const IgemmArgs& a, int tile_n, int nsplit, hipStream_t s) {
  const int mtiles = (a.M + BM - 1) / BM;
  dim3 grid(mtiles * a.ntiles, nsplit);
  // buffer offsets are 32-bit with bit 31 reserved as the out-of-range marker
  DADD_REQUIRE((size_t)a.B * a.Hi * a.Wi * (size_t)(a.C1 > a.C2 ? a.C1 : a.C2) * 2 < 0x7FF00000ull &&
                   (size_t)a.N * a.K * 2 < 0x7FF00000ull,
               "igemm(dma): operand larger than the 2 GiB buffer window");
  if (tile_n == 160)
    hipLaunchKernelGGL(igemm_dma_kernel<160>, grid, dim3(256), smem_bytes<160>(), s, a);
  else
    hipLaunchKernelGGL(igemm_dma_kernel<128>, grid, dim3(256), smem_bytes<128>(), s, a);
  DADD_LAUNCH_CHECK();
  return DADD_OK;
}
