"""Static execution plans for the UNet epsilon-prediction, the VAE decoder and the DDIM loop.

A plan is built once per (batch, latent side): weights are repacked into the kernels' layouts
(fp16, ``[Cout][ky][kx][Cin]``), every intermediate gets a fixed buffer from a pool, and the
operator sequence is recorded as a list of bound launches.  Running the plan allocates nothing and
never synchronises, so one denoising step (``begin_step`` + UNet + DDIM update) is captured into a
hipGraph and replayed ``steps`` times with a device-side step counter.

What replaces what (reference paths):
  * ``UNetPlan``      — ``DiffusionModuleWithIP.forward`` -> ``OrdinalUNet.forward`` ->
                        diffusers ``UNet2DConditionModel`` (src/models/diffusion_module_ip.py:383-390,
                        src/models/unet/unet.py:96-146) incl. the 16 attention processors
                        (src/models/attention_processor_routing_gates.py:84-196 / _base.py:39-138)
  * ``VaeDecoderPlan``— ``SDVAE.decode`` (src/models/vae/vae.py:90-112)
  * ``DdimLoop``      — the loop body of ``_ddim_sample_ip``
                        (src/pipelines/inference/inference_pipeline_ip.py:423-468)
Step-invariant work is hoisted exactly: the K/V projections of the conditioning
tokens (16 sites) and the whole time-embedding MLP + 22 ``time_emb_proj`` rows for all steps are
computed once per sampling run.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import lib as L
from .routing import get_block_type  # noqa: F401  (re-exported for callers)

F16, F32 = torch.float16, torch.float32
UNET_CH = (320, 640, 1280, 1280)
VAE_CH = (128, 256, 512, 512)
HEADS = 8
GROUPS = 32
N_CU = 256
# attn2 folded into one kernel (csrc/attn2_fused.hip).  Used where one launch has at least A2_MIN_TILES 128-token
# tiles (B=4: the 64x64 sites; same-box A/B +0.9 % end to end); on the smaller maps its 32 / 8 workgroups leave the
# chip idle (-6 % when forced everywhere), so those keep to_q + xattn + to_out.  DADD_FUSED_ATTN2=0 switches it off.
FUSED_ATTN2 = os.environ.get("DADD_FUSED_ATTN2", "1") == "1"
A2_MIN_TILES = int(os.environ.get("DADD_A2_MIN_TILES", "128"))


# ----------------------------------------------------------------------------- weight packing
def pack_conv(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin,kh,kw] (or [Cout,Cin]) fp32 -> [Cout, kh*kw*Cin] fp16, K = (tap, cin)."""
    if w.dim() == 2:
        return w.to(F16).contiguous()
    co, ci, kh, kw = w.shape
    return w.permute(0, 2, 3, 1).reshape(co, kh * kw * ci).to(F16).contiguous()


def pack_conv_cin8(w: torch.Tensor) -> torch.Tensor:
    """[Cout,Cin<=8,3,3] -> [Cout,9,8] fp16, input channels zero-padded to 8."""
    co, ci, _, _ = w.shape
    out = torch.zeros(co, 9, 8, dtype=F16)
    out[:, :, :ci] = w.permute(0, 2, 3, 1).reshape(co, 9, ci).to(F16)
    return out.contiguous()


def pack_conv_cout4(w: torch.Tensor) -> torch.Tensor:
    """[Cout<=4,C,3,3] -> [Cout,9,C] fp16."""
    co, ci, _, _ = w.shape
    return w.permute(0, 2, 3, 1).reshape(co, 9, ci).to(F16).contiguous()


def geglu_interleave(w: torch.Tensor, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Row order for the fused GEGLU epilogue: every 128-row tile = 2 waves x [32 hidden | 32 gate]
    rows of the same 32 output columns (diffusers GEGLU: ``hidden, gate = proj(x).chunk(2)``)."""
    n2 = w.shape[0] // 2
    idx = torch.arange(w.shape[0])
    tile, r = idx // 128, idx % 128
    wn, rr = r // 64, r % 64
    col = tile * 64 + wn * 32 + (rr % 32)
    src = torch.where(rr < 32, col, n2 + col)
    return w[src].contiguous(), b[src].contiguous()


def fold_layernorm(w: torch.Tensor, b: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
    """LayerNorm folded into the linear that consumes it (csrc/igemm_args.h, DADD_EPI_LNFOLD):
    LN(x) W^T + b = rstd (x (gamma o W)^T - mu c1) + (W beta + b).  Returns (gamma o W in fp16, c1, composed bias);
    c1 sums the ROUNDED weights, so that the mean term cancels exactly against what the MFMAs accumulate."""
    w, gamma, beta = w.double(), gamma.double(), beta.double()
    w16 = (w * gamma[None, :]).to(F16)
    bias = w @ beta + (b.double() if b is not None else 0.0)
    return w16, w16.double().sum(dim=1).float(), bias.float()


def lds_image(w: torch.Tensor) -> torch.Tensor:
    """[R, 64] fp16 slab -> the LDS image the DMA kernels read fragments from: 128-byte rows, the eight 16-byte chunks
    of row r stored at position ``chunk ^ ((r >> 1) & 7)`` (csrc/igemm_dma.hip ``lds_off``; an involution)."""
    r = w.shape[0]
    sw = (torch.arange(r) >> 1) & 7
    idx = (torch.arange(8)[None, :] ^ sw[:, None])                       # image position p holds chunk p ^ sw
    return w.reshape(r, 8, 8).gather(1, idx[:, :, None].expand(r, 8, 8)).reshape(r, 64).contiguous()


def pack_ffn_stream(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, wp: torch.Tensor):
    """Weights of a transformer block's tail in the order ``csrc/ffn_block.hip`` consumes them, as ready LDS images:
    per 64 hidden channels five [128 x 64] pieces of ``ff.net.0.proj`` (rows: wave column wn -> [h0 g0 h1 g1], 16 rows
    each, hidden index = chunk*64 + wn*32 + u*16 + e, gate row = 4C + hidden: diffusers ``GEGLU`` is
    ``hidden, gate = proj(x).chunk(2)``), then two [160 x 64] pieces of ``ff.net.2`` (output channels x the chunk's
    hidden channels); after the 20 chunks the ten [160 x 64] pieces of ``proj_out``.  Returns (stream fp16 1-D, bias of
    the GEGLU rows in the same order)."""
    c, hid = 320, 1280
    assert tuple(w1.shape) == (2 * hid, c) and tuple(w2.shape) == (c, hid) and tuple(wp.shape)[:2] == (c, c)
    w1, w2, wp = w1.to(F16), w2.to(F16), wp.reshape(c, c).to(F16)
    pieces, bias = [], []
    e = torch.arange(16)
    for ch in range(hid // 64):
        rows = []
        for wn in range(2):
            for u in range(2):
                h = ch * 64 + wn * 32 + u * 16 + e
                rows += [h, hid + h]
                bias += [b1[h].float(), b1[hid + h].float()]
        rows = torch.cat(rows)                                   # 128 rows of ff.net.0.proj
        for kt in range(c // 64):
            pieces.append(lds_image(w1[rows][:, kt * 64:(kt + 1) * 64]))
        for nh in range(2):
            pieces.append(lds_image(w2[nh * 160:(nh + 1) * 160, ch * 64:(ch + 1) * 64]))
    for nh in range(2):
        for kt in range(c // 64):
            pieces.append(lds_image(wp[nh * 160:(nh + 1) * 160, kt * 64:(kt + 1) * 64]))
    return torch.cat([p.reshape(-1) for p in pieces]).contiguous(), torch.cat(bias).contiguous()


def pack_head_stream(wp: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor) -> torch.Tensor:
    """Weights of a transformer block's head in the order ``csrc/tf_head.hip`` consumes them: ten [160 x 64] LDS images of
    ``proj_in`` (column half, K tile), then thirty of ``to_q | to_k | to_v`` (six column blocks of 160, K tile)."""
    c = 320
    wp = wp.reshape(c, c).to(F16)
    wqkv = torch.cat([wq, wk, wv]).to(F16)
    assert tuple(wqkv.shape) == (3 * c, c)
    pieces = []
    for w in (wp, wqkv):
        for nh in range(w.shape[0] // 160):
            for kt in range(c // 64):
                pieces.append(lds_image(w[nh * 160:(nh + 1) * 160, kt * 64:(kt + 1) * 64]))
    return torch.cat([p.reshape(-1) for p in pieces]).contiguous()


# GroupNorm (+ SiLU) applied inside the consuming 3x3 conv (DADD_PRE_GN, csrc/conv_halo.hip GNIN): the halo is normalised
# in LDS by the loader waves; needs the producer's chunk partials, one source, <= 1024 input channels, the halo kernel
GN_IN_CONV = True

# conv3x3_halo_kernel with two MFMA waves per SIMD (csrc/conv_halo.hip, DUO): measured equal to the one-wave build
# (profiles/r02_zn_halo_duo_ab.txt), so off; True is the A/B switch
HALO_DUO = False

# On the 8x8 maps the split-K finish kernel of a conv also writes the GroupNorm (+ SiLU) of its output for the next conv
# (DADD_EPI_GNAPPLY, csrc/igemm.hip splitk_finish_gnapply_kernel): one launch instead of finish + single-launch GroupNorm
FINISH_GN_APPLY = True

# Weight prefetch on a side branch of the graph (csrc/api.hip dadd_prefetch): how many weight-bearing launches ahead of
# its consumer a weight tensor is read into the Infinity Cache (0 = off), and the smallest tensor worth a branch.
# MEASURED AND OFF (profiles/r03_zh_weight_prefetch_ab.txt, same box, A B A B): 385.7 / 385.0 ms per pass without, 535.4 /
# 534.8 ms with - a captured graph with ~80 side branches per step makes hipGraphLaunch host-bound (host queue time 277 ->
# 504 ms per pass), far more than the 3-8 us per launch that hot weights save (profiles/r03_zg_cold_hot_weights.txt).
WEIGHT_PREFETCH_AHEAD = int(os.environ.get("DADD_WEIGHT_PREFETCH", "0"))
WEIGHT_PREFETCH_MIN_BYTES = 2 * 1024 * 1024

# GroupNorm statistics written by the producing GEMM's epilogue (no gn_stats launch, one read of the tensor less).
GN_FROM_EPILOGUE = True
GN_FUSED_MAX_BYTES = 16 * 1024       # csrc/norm.hip: below this slab size the single-launch LDS GroupNorm runs
SPLITK_GN_ROWS = 16                  # rows per block / per GroupNorm chunk of splitk_finish_gn_kernel (16 or 64)


# LayerNorm folded into the consuming linear (qkv, attn2.to_q, GEGLU projection) instead of a LayerNorm launch and a
# normalised copy of the hidden states.  Measured in situ (profiles/r02_d_step_profile_lnfold_everywhere.txt): the row
# statistics are 64 v_dot2c per K tile in the MFMA waves — free where those waves wait on the LDS fill (64-row tiles:
# 15.7 vs 15.3 us, minus a 4.7 us LayerNorm launch), but +11..19 us on the 128-row tiles, whose MFMA waves are the
# critical path (qkv 64x64: 45 vs 26 us).  So "auto" folds exactly the linears planned on 64-row tiles; True / False
# force it everywhere / nowhere (tests, A/B measurements).
LN_FOLD = "auto"
# ... and on the 128-row tiles the fold is epilogue arithmetic only when the row statistics come from the GEMM that
# produced the hidden states (DADD_EPI_LNSTAT partials of its output, csrc/igemm_epilogue.h): no LayerNorm launch, no
# normalised copy, no statistics in the consumer's K loop.
LN_STATS_FROM_PRODUCER = True
LN_STATS_MAX_PARTS = 8          # what the consumer stages in LDS (csrc/ln_lds.h); more parts (1280 channels from 64-column
                                # tiles: 40) would be read from global memory in the epilogue — slower than the LayerNorm launch


# The tail of a 320-channel transformer block (norm3 -> GEGLU -> FF-out -> proj_out) as ONE launch per 64-token row block
# (csrc/ffn_block.hip): the (B*N) x 4C GEGLU output never leaves the CU.  Each workgroup streams the block's 2.6 MB of
# weights, so it pays only while there are about as many row blocks as CUs (B = 4 at 64x64: 256); below FFN_MIN_BLOCKS
# the three GEMM launches stay.
FUSED_FFN = True
FFN_MIN_BLOCKS = 128
# ... and its head (GroupNorm -> proj_in -> norm1 -> q|k|v) the same way (csrc/tf_head.hip), where the block's input
# comes with GroupNorm chunk partials from its producer
FUSED_HEAD = True


def fold_here(m: int, n: int, k: int, geglu: bool = False) -> bool:
    if LN_FOLD == "auto":
        tile_m, _, _, tune = plan_tiling(m, n, k, 1, geglu, False)
        return tile_m == 64 and not (tune & L.TUNE_NODMA)       # the statistics live in the LDS-DMA kernel
    return bool(LN_FOLD)


def choose_tiling(m: int, n: int, k: int, tile_n: int, geglu: bool = False,
                  residual: bool = True) -> Tuple[int, int, int]:
    """(tile_m, splitk, tune_flags) for one implicit GEMM with 128- or 160-column tiles, from measurements over
    every layer shape of the B=4 / 512x512 step on MI355X (scripts/op_bench.py, profiles/r01_*_op_bench.txt):
      * K >= 12 tiles: the wave-specialised LDS-DMA ring kernel, 128-row tiles, ONE workgroup per CU;
        split K only while a slice keeps >= 16 K tiles (8-10 on the small 8x8..32x32 grids) and the grid
        stays near 256 workgroups — the fp32 slabs and the finish kernel cost more than idle CUs below that;
        with >= 8 tiles per CU (VAE) the ring is kept running over a run of tiles (DADD_TUNE_PERSIST);
      * short K with more 128-row tiles than CUs (qkv, the GEGLU projection at 32x32): the same kernel
        with its ring kept running over a run of output tiles per workgroup (DADD_TUNE_PERSIST);
      * other short-K linears: without a residual input the LDS-DMA kernel (its loaders run no vector
        arithmetic), with one the register-staged kernel, two workgroups per CU, 64-row tiles when 128-row
        tiles would not give ~512 workgroups;
      * GEGLU at 64x64 (2560 tiles, erf epilogue as long as the MFMAs): register-staged 64-row tiles."""
    nkt = k // 64
    nt = math.ceil(n / tile_n)
    t128 = math.ceil(m / 128) * nt
    if geglu:
        if m >= 16384:
            return 64, 1, L.TUNE_NODMA | L.TUNE_SHALLOW
        return 128, 1, (L.TUNE_PERSIST if t128 >= 4 * N_CU else 0)
    if nkt >= 12:
        if n <= 128 and nkt <= 18:      # VAE 128-channel convs: one column tile, the DMA stream is the longer side
            return 128, 1, L.TUNE_NODMA
        per = 16 if t128 > 64 else (10 if t128 > 16 else 8)
        sk = max(1, min(round(N_CU / t128), nkt // per, 32))
        if not residual and nkt <= 20:      # plain projections: one pass beats slabs + finish (15.0 vs 18.9 us)
            sk = 1
        return 128, sk, (L.TUNE_PERSIST if sk == 1 and t128 >= 8 * N_CU else 0)   # VAE: >= 8 tiles per CU
    if t128 > N_CU:
        return 128, 1, L.TUNE_PERSIST
    if not residual and nkt >= 5:           # q / proj_in style linears without a residual read: DMA ring wins
        return 128, 1, 0
    return (128 if t128 >= 2 * N_CU else 64), 1, L.TUNE_NODMA


def choose_splitk(m: int, n: int, k: int, tile_n: int) -> int:
    return choose_tiling(m, n, k, tile_n)[1]


# Measured tile table: (m, n, k, taps, geglu, residual, ups, stride) -> (tile_m, tile_n, splitk, tune), written by
# scripts/tile_sweep.py from in-situ per-launch timings of the B=4 / 512x512 step; shapes outside it use the rules of
# choose_tiling().  TILING_OVERRIDE (same keys) is the sweep's own hook and wins over both.
TILING_OVERRIDE: Dict[Tuple, Tuple[int, int, int, int]] = {}
try:
    from .tiling_table import TABLE as TILING_TABLE, TABLE_R3 as TILING_TABLE_R3
except ImportError:          # no table committed yet
    TILING_TABLE, TILING_TABLE_R3 = {}, {}
TILING_R3 = True             # the round-3 entries of the table (64-row tiles with K slices on the small maps)


def tiling_key(m, n, k, taps, geglu, residual, ups=0, stride=1):
    return (int(m), int(n), int(k), int(taps), bool(geglu), bool(residual), int(bool(ups)), int(stride))


def plan_tiling(m, n, k, taps, geglu, residual, ups=0, stride=1) -> Tuple[int, int, int, int]:
    """(tile_m, tile_n, splitk, tune) of one implicit GEMM."""
    key = tiling_key(m, n, k, taps, geglu, residual, ups, stride)
    hit = TILING_OVERRIDE.get(key) or (TILING_TABLE_R3.get(key) if TILING_R3 else None) or TILING_TABLE.get(key)
    if hit is not None:
        return tuple(hit)
    tile_n = 128 if geglu or n % 160 else 160
    tile_m, sk, tune = choose_tiling(m, n, k, tile_n, geglu, residual)
    # short linears of the small maps (<= 128 tiles of 128 rows): 64x64 LDS-DMA tiles, two workgroups per CU, fill
    # the chip without fp32 slabs and a finish launch
    if (taps == 1 and not geglu and not ups and n % 64 == 0 and k // 64 <= 40
            and math.ceil(m / 128) * math.ceil(n / tile_n) <= N_CU // 2):
        return 64, 64, 1, 0
    return tile_m, tile_n, sk, tune


class Pool:
    """Plan-time buffer pool: fixed addresses, explicit release, reuse by (shape, dtype)."""

    def __init__(self, be):
        self.be = be
        self.free: Dict[Tuple, List[torch.Tensor]] = {}
        self.bytes = 0
        self.on_put = None

    def get(self, shape, dtype=F16) -> torch.Tensor:
        key = (tuple(shape), dtype)
        lst = self.free.get(key)
        if lst:
            return lst.pop()
        t = self.be.empty(shape, dtype)
        self.bytes += t.numel() * t.element_size()
        return t

    def put(self, *ts):
        for t in ts:
            if t is not None:
                if self.on_put is not None:
                    self.on_put(t)
                self.free.setdefault((tuple(t.shape), t.dtype), []).append(t)


class _Plan:
    def __init__(self, be, wcache=None):
        self.be = be
        self.wcache = wcache
        self.pool = Pool(be)
        self.ops: List = []
        self.keep: List[torch.Tensor] = []  # weights & persistent buffers
        self.gn_partials: Dict[int, Tuple[torch.Tensor, int]] = {}   # output buffer -> (chunk partials, chunks)
        self.ln_partials: Dict[int, torch.Tensor] = {}               # output buffer -> LayerNorm row partials [P][M][2]

        self._memo: Dict = {}
        self.gn_ready: Dict[int, Tuple] = {}      # output buffer -> (its GroupNorm written by the finish kernel, gamma, beta, eps, silu)

        def _forget(t):                 # a recycled buffer loses its statistics
            self.gn_partials.pop(t.data_ptr(), None)
            self.ln_partials.pop(t.data_ptr(), None)
            self.gn_ready.pop(t.data_ptr(), None)
        self.pool.on_put = _forget
        self.gn_ws = None
        # Split-K slabs are combined by the finish kernel.  The in-launch combine (last-arriving slice
        # reduces; `counters` of dadd_conv_igemm_f16) is implemented and tested but measured 2-4x SLOWER
        # on MI355X for these shapes (M=256 N=1280 K=11520, 16 slices: 122 us vs 27 us): every slice
        # pays an agent-scope release (L2 write-back) — cdna_hip_programming.md "cut GEMM->GEMM seams".
        self.sk_counters = None

    def rec(self, fn, *a, **k):
        self.ops.append((fn, a, k))

    def run(self):
        for fn, a, k in self.ops:
            fn(*a, **k)

    def dev(self, t, dtype=None):
        d = self.be.to_device(t, dtype)
        self.keep.append(d)
        return d

    def cached(self, key, make):
        """Device tensor for ``key`` from the optional cross-plan cache (``wcache``: plans for other batch sizes
        or tilings over the same state dict share the packed weights instead of re-packing 1.9 GB each)."""
        if self.wcache is None:         # plan-local memo: one device copy per key (fuse_gn hints name the same tensors twice)
            t = self._memo.get(key)
            if t is None:
                t = self._memo[key] = make()
            return t
        t = self.wcache.get(key)
        if t is None:
            t = self.wcache[key] = make()
        else:
            self.keep.append(t)
        return t

    # ---- recorded building blocks -----------------------------------------------------------
    def conv(self, x, w, out_shape, *, x2=None, bias=None, rowvec=None, residual=None, taps=9,
             stride=1, ups=0, pad=1, flags=0, ln_c1=None, ln_eps=1e-5, gn_stats=False, ln_stats=False,
             ln_stats_in=None, gn_in=None, fuse_gn=None):
        """``gn_in`` = (partials, nchunk, gamma, beta, eps, silu): GroupNorm of ``x`` on the way in (``gn_in_conv_ok``).
        ``fuse_gn`` = (gamma, beta, eps, silu) of the GroupNorm that consumes the output: on the small maps, where the
        conv runs split-K and a (sample, group) slab is <= 16 KiB, the finish kernel writes the normalised copy too
        (DADD_EPI_GNAPPLY) and the consuming ``gn()`` finds it in ``self.gn_ready`` instead of launching.
        ``ln_stats``: the output feeds a LayerNorm whose consumer is a folded linear — have the epilogue write the row
        partials (DADD_EPI_LNSTAT) where the launch allows (plain linear, one K pass, whole wave column blocks); they
        are found again through ``self.ln_partials``.  ``ln_stats_in``: such partials of ``x`` for this folded linear.
        ``gn_stats``: the output feeds a GroupNorm — have the epilogue write its chunk partials (DADD_EPI_GNSTAT)
        where the tiling allows (full 128/160-column tiles holding whole groups, row blocks inside one sample, <= 128
        chunks, no split-K); the consuming ``gn()`` then skips its statistics pass."""
        out = self.pool.get(out_shape)
        n = w.shape[0]
        m = out_shape[0] * out_shape[1] * out_shape[2]
        tile_m, tile_n, sk, tune = plan_tiling(m, n, w.shape[1], taps, bool(flags & L.EPI_GEGLU), residual is not None,
                                               ups, stride)
        if taps == 9 and HALO_DUO:
            tune |= L.TUNE_SHALLOW
        if ln_c1 is not None:
            sk = 1                      # the row statistics come from whole rows of A: no K slices
        partial = self.pool.get((sk * m * n,), F32) if sk > 1 else None
        gkw = {}
        if gn_stats and GN_FROM_EPILOGUE and not (flags & L.EPI_GEGLU) and n % 32 == 0:
            howo, cg = out_shape[1] * out_shape[2], n // 32
            if sk == 1:     # the GEMM's own epilogue writes the partials: one chunk per MFMA wave's row block
                wm_rows = tile_m // 2
                ok = tile_n in (128, 160) and (tile_n // 2) % cg == 0 and m % tile_m == 0 and n % tile_n == 0
            else:           # split-K: the finish kernel writes them, per 16 rows x 160 (128) columns
                wm_rows = SPLITK_GN_ROWS if howo // SPLITK_GN_ROWS <= 128 else 64      # (<= 128 chunks keeps GroupNorm-in-conv eligible)
                cb = 160 if n % 160 == 0 else 128
                ok = n % cb == 0 and cb % cg == 0 and m % wm_rows == 0
            nchunk = howo // wm_rows
            # (maps whose (batch, group) slab fits the single-launch LDS GroupNorm keep that path)
            # (> 128 chunks — the VAE maps — are folded to 64 per sample by gn_reduce_kernel inside dadd_groupnorm_f16:
            # one small launch instead of a statistics pass over the tensor; `ws` carries the room for them)
            if ok and howo % wm_rows == 0 and nchunk >= 1 and howo * cg * 2 > GN_FUSED_MAX_BYTES:
                extra = 64 if nchunk > 128 else 0
                ws = self.be.zeros((out_shape[0] * (nchunk + extra) * GROUPS * 2,), F32)
                self.keep.append(ws)
                self.gn_partials[out.data_ptr()] = (ws, nchunk)
                gkw = dict(gn_ws=ws, gn_nchunk=nchunk)
                flags |= L.EPI_GNSTAT
        if (ln_stats and LN_STATS_FROM_PRODUCER and not (flags & (L.EPI_GEGLU | L.EPI_GNSTAT)) and sk == 1
                and n % (tile_n // 2) == 0 and n // (tile_n // 2) <= LN_STATS_MAX_PARTS):
            st = self.be.zeros((n // (tile_n // 2), m, 2), F32)
            self.keep.append(st)
            self.ln_partials[out.data_ptr()] = st
            gkw["ln_stats_out"] = st
            flags |= L.EPI_LNSTAT
        f = flags | (L.EPI_BIAS if bias is not None else 0) | (L.EPI_ROWVEC if rowvec is not None else 0) \
            | (L.EPI_RESIDUAL if residual is not None else 0) | (L.EPI_LNFOLD if ln_c1 is not None else 0) | tune
        kw = dict(ln_c1=ln_c1, ln_eps=ln_eps) if ln_c1 is not None else {}
        if (fuse_gn is not None and FINISH_GN_APPLY and sk > 1 and self.sk_counters is None and not (f & L.EPI_GNSTAT)
                and n % 64 == 0 and out_shape[1] * out_shape[2] * (n // 32) * 2 <= GN_FUSED_MAX_BYTES):
            g_out = self.pool.get(out_shape)
            f |= L.EPI_GNAPPLY | (L.EPI_GNAPPLY_SILU if fuse_gn[3] else 0)
            kw["gn_apply"] = (g_out, fuse_gn[0], fuse_gn[1], fuse_gn[2])
            self.gn_ready[out.data_ptr()] = (g_out, fuse_gn[0].data_ptr(), fuse_gn[1].data_ptr(), float(fuse_gn[2]), bool(fuse_gn[3]))
        if gn_in is not None:       # (partials, chunks, gamma, beta, eps, silu[, partials of x2, chunks of x2])
            f |= L.PRE_GN | (L.PRE_GN_SILU if gn_in[5] else 0)
            kw["gn_in"] = tuple(gn_in[:5]) + tuple(gn_in[6:])
        if ln_c1 is not None and ln_stats_in is not None:
            kw["ln_stats_in"] = ln_stats_in
        self.rec(self.be.igemm, x, w, out, x2=x2, bias=bias, rowvec=rowvec, residual=residual,
                 taps=taps, stride=stride, ups=ups, pad=pad, flags=f, splitk=sk, partial=partial,
                 tile_n=tile_n, tile_m=tile_m, counters=self.sk_counters if sk > 1 else None, **kw, **gkw)  # None: finish kernel
        self.pool.put(partial)
        return out

    def gn_in_conv_ok(self, x, x2, w, out_shape, residual) -> Optional[Tuple]:
        """(partials, chunks) of ``x`` if the 3x3 conv (x -> out_shape, weights w) can apply the GroupNorm itself: the conv
        must land on the halo kernel (csrc/igemm.hip: stride 1 / pad 1, 64-, 32- or 16-wide square map in whole 128-pixel
        tiles, 128x160 LDS-DMA tiles, no persistent ring) and ``x`` must be a single source of <= 1152 / 2048 / 2432 channels whose
        producer wrote <= 128 chunk partials."""
        cmax = {64: 1152, 32: 2048, 16: 2432}.get(x.shape[2], 0)      # (scale, shift) pairs that fit beside the halo in LDS
        cin = x.shape[-1] + (0 if x2 is None else x2.shape[-1])
        if not GN_IN_CONV or cin > cmax or x.shape[-1] % 64 or cin % 64:
            return None
        part = self.gn_partials.get(x.data_ptr())
        if part is None or part[1] > 128:
            return None
        if x2 is not None:          # skip-concat: both sources' partials, nesting group widths (DADD_PRE_GN contract)
            part2 = self.gn_partials.get(x2.data_ptr())
            c1_, c2_, cgc = x.shape[-1], x2.shape[-1], cin // 32
            if (part2 is None or part2[1] > 128 or c1_ % 32 or c2_ % 32 or cgc % (c1_ // 32) or cgc % (c2_ // 32)
                    or c1_ % cgc):
                return None
            part = (part[0], part[1], part2[0], part2[1])
        b, h, w_, _ = x.shape
        n, m = w.shape[0], out_shape[0] * out_shape[1] * out_shape[2]
        if tuple(out_shape[1:3]) != (h, w_) or h != w_ or w_ not in (16, 32, 64) or (h * w_) % 128:
            return None
        tile_m, tile_n, _, tune = plan_tiling(m, n, w.shape[1], 9, False, residual is not None, 0, 1)
        if tile_m != 128 or tile_n != 160 or (tune & (L.TUNE_NODMA | L.TUNE_PERSIST)) or HALO_DUO:
            return None
        return part

    def gn(self, x1, x2, gamma, beta, eps, silu):
        c = x1.shape[-1] + (0 if x2 is None else x2.shape[-1])
        ready = self.gn_ready.get(x1.data_ptr()) if x2 is None else None
        if ready is not None and ready[1:] == (gamma.data_ptr(), beta.data_ptr(), float(eps), bool(silu)):
            del self.gn_ready[x1.data_ptr()]       # written by the producer's split-K finish (DADD_EPI_GNAPPLY)
            return ready[0]
        out = self.pool.get((*x1.shape[:3], c))
        part = self.gn_partials.get(x1.data_ptr()) if x2 is None else None
        if part is not None:            # statistics already written by the producing epilogue
            self.rec(self.be.groupnorm, x1, None, gamma, beta, out, part[0], GROUPS, eps, silu, ws_chunks=part[1])
        else:
            self.rec(self.be.groupnorm, x1, x2, gamma, beta, out, self.gn_ws, GROUPS, eps, silu)
        return out


# ----------------------------------------------------------------------------- UNet
class UNetPlan(_Plan):
    """SD-1.x UNet forward for a fixed (B, S); eps = plan(latents) with cond/time prepared apart."""

    def __init__(self, be, sd: Dict[str, torch.Tensor], batch: int, side: int, *,
                 prefix="unet.unet", use_routing_gates=True, use_frequency_strategy=True, wcache=None):
        super().__init__(be, wcache)
        assert side % 8 == 0, "latent side must be a multiple of 8 (three stride-2 levels)"
        self.B, self.S = batch, side
        self.gates_mode = use_routing_gates
        self.prefix = prefix + "."
        self.sd = sd
        self.lam = 0.0
        # device-side scalars read by the kernels of a captured step: [0] = lambda (delta steering), [1] = CFG scale
        self.params = be.zeros((2,), F32)
        self.params_dev = False      # True while a DdimLoop drives the plan: kernels take lambda from `params`
        self.ddim_coef = None        # set by DdimLoop for a step without CFG: conv_out applies the DDIM update itself
        self.gn_ws = be.empty((batch * L.GN_MAX_CHUNKS * GROUPS * 2,), F32)
        u = self.prefix
        # ---- time path: linear_1, linear_2, all 22 time_emb_proj concatenated
        self.w_t1 = self.dev(sd[u + "time_embedding.linear_1.weight"], F16)
        self.b_t1 = self.dev(sd[u + "time_embedding.linear_1.bias"])
        self.w_t2 = self.dev(sd[u + "time_embedding.linear_2.weight"], F16)
        self.b_t2 = self.dev(sd[u + "time_embedding.linear_2.bias"])
        names = self._resnet_names()
        self.temb_off, off = {}, 0
        for nme in names:
            self.temb_off[nme] = off
            off += sd[u + nme + ".time_emb_proj.weight"].shape[0]
        self.temb_cols = off
        self.w_tp = self.dev(torch.cat([sd[u + n + ".time_emb_proj.weight"] for n in names]), F16)
        self.b_tp = self.dev(torch.cat([sd[u + n + ".time_emb_proj.bias"] for n in names]))
        self.temb_rows = be.zeros((batch, self.temb_cols), F32)     # current step's rows
        # ---- cross-attention conditioning caches (step invariant)
        self.sites = self._attn_sites()
        self.T = 48 if use_routing_gates else 32
        self.kv_w, self.kv, self.gates = {}, {}, {}
        for site, c in self.sites:
            ap = f"{u}{site}.transformer_blocks.0.attn2"
            ws = [sd[ap + ".to_k.weight"], sd[ap + ".to_v.weight"]]
            if use_routing_gates:
                ws += [sd[ap + ".processor.to_k_dis.weight"], sd[ap + ".processor.to_v_dis.weight"]]
                self.gates[site] = self.dev(torch.stack([sd[ap + ".processor.anat_gate"],
                                                         sd[ap + ".processor.dis_gate"]]).float())
            else:
                self.gates[site] = None
            self.kv_w[site] = self.dev(torch.cat(ws), F16)
            self.kv[site] = [be.zeros((batch, 1, self.T, self.kv_w[site].shape[0]), F16)
                             for _ in range(2)]                     # [cond, uncond]
        self.cond16 = be.zeros((batch, 1, self.T, 768), F16)
        self.kv_slot = 0
        self.cond_gen = 0      # advanced by every set_cond(): callers that cache projections key on it
        # ---- attn2 folded into ONE kernel per block (dadd_attn2_fused_f16): with 16 keys per pathway
        # q K^T = x (W_q K^T) and P V W_o^T = P (V W_o^T), so the step-invariant conditioning absorbs both
        # projections.  Sites whose map is smaller than one 128-token tile (8x8) keep the three-kernel path.
        self.fused_attn2 = bool(use_routing_gates) and FUSED_ATTN2
        self.a2, self._a2_dirty, self._a2_lam = {}, True, None
        if self.fused_attn2:
            for site, c in self.sites:
                hw = self._site_hw(site)
                # one workgroup per 128-token tile: sites with few tiles leave most of the chip idle
                if hw % 128 == 0 and c % 320 == 0 and batch * hw // 128 >= A2_MIN_TILES:
                    ap = f"{u}{site}.transformer_blocks.0.attn2"
                    self.a2[site] = dict(
                        wq=self.dev(sd[ap + ".to_q.weight"].float()), wo=self.dev(sd[ap + ".to_out.0.weight"].float()),
                        mcat=be.zeros((batch, 384, c), F16), vw=be.zeros((batch, c, 384), F16),
                        # norm2 folded into the score GEMM (set by _transformer when the producer supplies row partials)
                        fold=False, ln_c1=be.zeros((batch, 384), F32), ln_d=be.zeros((batch, 384), F32),
                        gamma=self.dev(sd[f"{u}{site}.transformer_blocks.0.norm2.weight"].float()),
                        beta=self.dev(sd[f"{u}{site}.transformer_blocks.0.norm2.bias"].float()))
        # The fold runs once per conditioning (every sampler pass): sites of equal width share ONE set of batched torch
        # ops over stacked weights / buffers (five sites at 64x64: ~25 launches instead of ~125, profiles/r03_*_pass_breakdown)
        self.a2_groups = []
        by_c: Dict[int, List[str]] = {}
        for site, c in self.sites:
            if site in self.a2:
                by_c.setdefault(c, []).append(site)
        for c, names in by_c.items():
            self.a2_groups.append(self._a2_group(c, names, batch))
        # ---- I/O
        self.lat_in = be.zeros((batch, 4, side, side), F32)
        self.eps_out = [be.zeros((batch, 4, side, side), F32) for _ in range(2)]
        self._build()

    def _a2_group(self, c, names, batch):
        """Stacked weights and output buffers of the fused-attn2 sites of width ``c`` (``prepare_attn2``)."""
        be = self.be
        ns, d = len(names), c // HEADS
        with be.ctx():                      # (the stacking reads tensors uploaded on the backend's stream)
            grp = dict(sites=names, c=c,
                       wq=torch.stack([self.a2[n]["wq"].view(HEADS, d, c) for n in names]),
                       wo=torch.stack([self.a2[n]["wo"].view(c, HEADS, d) for n in names]),
                       gamma=torch.stack([self.a2[n]["gamma"] for n in names]).view(ns, 1, 1, c),
                       beta=torch.stack([self.a2[n]["beta"] for n in names]).view(ns, 1, 1, c),
                       gates=torch.stack([self.gates[n] for n in names]).float(),
                       mcat=be.zeros((ns, batch, 384, c), F16), vw=be.zeros((ns, batch, c, 384), F16),
                       ln_c1=be.zeros((ns, batch, 384), F32), ln_d=be.zeros((ns, batch, 384), F32))
        for i, n in enumerate(names):       # the per-site tensors the kernels read are slices of the stacked ones
            for key in ("mcat", "vw", "ln_c1", "ln_d"):
                self.a2[n][key] = grp[key][i]
            del self.a2[n]["wq"], self.a2[n]["wo"]
        self.keep += [grp[k] for k in ("wq", "wo", "gamma", "beta", "gates", "mcat", "vw", "ln_c1", "ln_d")]
        return grp

    # -- inventory -------------------------------------------------------------------------------
    @staticmethod
    def _resnet_names():
        n = [f"down_blocks.{i}.resnets.{j}" for i in range(4) for j in range(2)]
        n += ["mid_block.resnets.0", "mid_block.resnets.1"]
        n += [f"up_blocks.{i}.resnets.{j}" for i in range(4) for j in range(3)]
        return n

    def _site_hw(self, site: str) -> int:
        """Tokens per sample at an attention site (latent side S: 64x64 -> S^2, S^2/4, S^2/16, mid S^2/64)."""
        if site.startswith("down_blocks."):
            r = self.S >> int(site.split(".")[1])
        elif site.startswith("up_blocks."):
            r = self.S >> (3 - int(site.split(".")[1]))
        else:
            r = self.S >> 3
        return r * r

    @staticmethod
    def _attn_sites():
        s = [(f"down_blocks.{i}.attentions.{j}", UNET_CH[i]) for i in range(3) for j in range(2)]
        s.append(("mid_block.attentions.0", 1280))
        s += [(f"up_blocks.{i}.attentions.{j}", UNET_CH[3 - i]) for i in (1, 2, 3) for j in range(3)]
        return s

    def w(self, key, pack=pack_conv):
        return self.cached((self.prefix + key, pack.__name__), lambda: self.dev(pack(self.sd[self.prefix + key])))

    def f(self, key):
        return self.cached((self.prefix + key, "f32"), lambda: self.dev(self.sd[self.prefix + key].float()))

    # -- blocks ----------------------------------------------------------------------------------
    def _gn_of(self, key, eps, silu):
        """(gamma, beta, eps, silu) of the GroupNorm ``key`` — the ``fuse_gn`` hint of the conv that feeds it."""
        return (self.f(key + ".weight"), self.f(key + ".bias"), eps, silu)

    def _resnet(self, name, x, skip=None, next_gn=None):
        """``next_gn``: the GroupNorm that consumes this block's output first (``_gn_of``), if the caller knows it."""
        b, h, w_, _ = x.shape
        cin = x.shape[-1] + (0 if skip is None else skip.shape[-1])
        cout = self.sd[self.prefix + name + ".conv1.weight"].shape[0]
        off = self.temb_off[name]
        w1 = self.w(name + ".conv1.weight")
        p1 = self.gn_in_conv_ok(x, skip, w1, (b, h, w_, cout), None)
        if p1 is not None:           # norm1 + SiLU inside conv1 (p1 carries the skip's partials too when there is one)
            h1 = self.conv(x, w1, (b, h, w_, cout), x2=skip, bias=self.f(name + ".conv1.bias"),
                           rowvec=self.temb_rows[:, off:off + cout], gn_stats=True, fuse_gn=self._gn_of(name + ".norm2", 1e-5, 1),
                           gn_in=(p1[0], p1[1], self.f(name + ".norm1.weight"), self.f(name + ".norm1.bias"), 1e-5, 1) + tuple(p1[2:]))
        else:
            g1 = self.gn(x, skip, self.f(name + ".norm1.weight"), self.f(name + ".norm1.bias"), 1e-5, 1)
            h1 = self.conv(g1, w1, (b, h, w_, cout), bias=self.f(name + ".conv1.bias"),
                           rowvec=self.temb_rows[:, off:off + cout], gn_stats=True, fuse_gn=self._gn_of(name + ".norm2", 1e-5, 1))
            self.pool.put(g1)
        if cin != cout:
            res = self.conv(x, self.w(name + ".conv_shortcut.weight"), (b, h, w_, cout), x2=skip,
                            bias=self.f(name + ".conv_shortcut.bias"), taps=1, pad=0)
        else:
            assert skip is None
            res = x
        w2 = self.w(name + ".conv2.weight")
        p2 = self.gn_in_conv_ok(h1, None, w2, (b, h, w_, cout), res)
        if p2 is not None:           # norm2 + SiLU inside conv2
            out = self.conv(h1, w2, (b, h, w_, cout), bias=self.f(name + ".conv2.bias"), residual=res, gn_stats=True, fuse_gn=next_gn,
                            gn_in=(p2[0], p2[1], self.f(name + ".norm2.weight"), self.f(name + ".norm2.bias"), 1e-5, 1))
            self.pool.put(h1)
        else:
            g2 = self.gn(h1, None, self.f(name + ".norm2.weight"), self.f(name + ".norm2.bias"), 1e-5, 1)
            self.pool.put(h1)
            out = self.conv(g2, w2, (b, h, w_, cout), bias=self.f(name + ".conv2.bias"), residual=res, gn_stats=True, fuse_gn=next_gn)
            self.pool.put(g2)
        if res is not x:
            self.pool.put(res)
        return out

    def _ln_linear(self, tb, norm, name, wkey_list, bias_key=None, geglu=False):
        """Device tensors (w16, c1, bias) of ``norm`` folded into the linear(s) ``wkey_list`` (rows concatenated)."""
        u = self.prefix

        def make():
            w = torch.cat([self.sd[u + tb + k] for k in wkey_list])
            b = self.sd[u + tb + bias_key] if bias_key else None
            w16, c1, bias = fold_layernorm(w, b, self.sd[u + tb + norm + ".weight"], self.sd[u + tb + norm + ".bias"])
            if geglu:               # row order of the fused GEGLU epilogue; c1 / bias follow their rows
                idx = geglu_interleave(torch.arange(w16.shape[0])[:, None].float(), torch.zeros(w16.shape[0]))[0][:, 0].long()
                w16, c1, bias = w16[idx], c1[idx], bias[idx]
            return self.dev(w16.contiguous()), self.dev(c1.contiguous()), self.dev(bias.contiguous())
        t = self.cached((u + tb, "lnfold", name), make)
        if self.wcache is not None:
            self.keep += list(t)
        return t

    def _transformer(self, site, x):
        b, h, w_, c = x.shape
        shp = (b, h, w_, c)
        tb = site + ".transformer_blocks.0"
        m_rows = b * h * w_
        fold1, fold2, fold3 = (fold_here(m_rows, 3 * c, c), fold_here(m_rows, c, c) and site not in self.a2,
                               fold_here(m_rows, 8 * c, c, True))
        fused_tail = FUSED_FFN and c == 320 and (h * w_) % 64 == 0 and m_rows // 64 >= FFN_MIN_BLOCKS
        xpart = self.gn_partials.get(x.data_ptr())
        fused_head = (FUSED_HEAD and c == 320 and (h * w_) % 64 == 0 and m_rows // 64 >= FFN_MIN_BLOCKS
                      and xpart is not None and xpart[1] <= 256)
        ext = LN_STATS_FROM_PRODUCER and LN_FOLD == "auto"
        if fused_head:               # norm -> proj_in -> norm1 -> q|k|v in one launch (csrc/tf_head.hip)
            u = self.prefix

            def _head():
                return self.dev(pack_head_stream(self.sd[u + site + ".proj_in.weight"], *[self.sd[u + tb + f".attn1.to_{n}.weight"] for n in "qkv"]))
            hstream = self.cached((self.prefix + tb, "head_stream"), _head)
            if self.wcache is not None:
                self.keep.append(hstream)
            hs = self.pool.get(shp)
            qkv = self.pool.get((b, h, w_, 3 * c))
            self.rec(self.be.tf_head, x.view(b, h * w_, c), hstream, xpart[0], xpart[1], self.f(site + ".norm.weight"),
                     self.f(site + ".norm.bias"), self.f(site + ".proj_in.bias"), self.f(tb + ".norm1.weight"),
                     self.f(tb + ".norm1.bias"), hs.view(b, h * w_, c), qkv.view(b, h * w_, 3 * c))
        else:
            g = self.gn(x, None, self.f(site + ".norm.weight"), self.f(site + ".norm.bias"), 1e-6, 0)
            hs = self.conv(g, self.w(site + ".proj_in.weight"), shp, bias=self.f(site + ".proj_in.bias"),
                           taps=1, pad=0, ln_stats=LN_STATS_FROM_PRODUCER and LN_FOLD == "auto" and not fold1)
            self.pool.put(g)
        ln_box = [None]                 # the normalised copy, allocated only if some LayerNorm still runs as a kernel

        def ln_of(src, norm):
            if ln_box[0] is None:
                ln_box[0] = self.pool.get(shp)
            self.rec(self.be.layernorm, src, self.f(tb + norm + ".weight"), self.f(tb + norm + ".bias"), ln_box[0])
            return ln_box[0]
        # attn1 (self)
        st1 = self.ln_partials.get(hs.data_ptr())
        if fused_head:
            pass
        elif fold1 or st1 is not None:
            wqkv, c1, bqkv = self._ln_linear(tb, ".norm1", "qkv", [f".attn1.to_{n}.weight" for n in "qkv"])
            qkv = self.conv(hs, wqkv, (b, h, w_, 3 * c), bias=bqkv, taps=1, pad=0, ln_c1=c1,
                            ln_stats_in=None if fold1 else st1)
        else:
            wqkv = self.cached((self.prefix + tb, "qkv"), lambda: self.dev(
                torch.cat([self.sd[self.prefix + tb + f".attn1.to_{n}.weight"] for n in "qkv"]), F16))
            qkv = self.conv(ln_of(hs, ".norm1"), wqkv, (b, h, w_, 3 * c), taps=1, pad=0)
        att = self.pool.get(shp)
        self.rec(self.be.self_attn, qkv.view(b, h * w_, 3 * c), att.view(b, h * w_, c), HEADS)
        self.pool.put(qkv)
        fused2 = site in self.a2
        h2 = self.conv(att, self.w(tb + ".attn1.to_out.0.weight"), shp,
                       bias=self.f(tb + ".attn1.to_out.0.bias"), residual=hs, taps=1, pad=0,
                       ln_stats=ext and not fold2)
        self.pool.put(hs)
        # attn2 (DADD cross-attention)
        st3 = None
        if fused2:                   # one kernel: x (W_q K^T) -> 24 softmaxes -> P (V W_o^T) + bias + residual
            st = self.a2[site]
            st2 = self.ln_partials.get(h2.data_ptr())
            kw2 = {}
            if st2 is not None:          # norm2 folded into the score GEMM: mcat carries gamma (prepare_attn2)
                st["fold"] = True
                lnx = h2
                kw2 = dict(ln_stats_in=st2, ln_c1=st["ln_c1"], ln_d=st["ln_d"], ln_eps=1e-5)
            else:
                lnx = ln_of(h2, ".norm2")
            h3 = self.pool.get(shp)
            if ext and not fold3 and c % 80 == 0 and not fused_tail:
                st3 = self.be.zeros((c // 80, m_rows, 2), F32)
                self.keep.append(st3)
                kw2["ln_stats_out"] = st3
            self.rec(self.be.attn2_fused, lnx.view(b, h * w_, c), st["mcat"], st["vw"],
                     self.f(tb + ".attn2.to_out.0.bias"), h2.view(b, h * w_, c), h3.view(b, h * w_, c), **kw2)
        else:
            st2 = self.ln_partials.get(h2.data_ptr())
            if fold2 or st2 is not None:
                wq, c1, bq = self._ln_linear(tb, ".norm2", "to_q", [".attn2.to_q.weight"])
                q = self.conv(h2, wq, shp, bias=bq, taps=1, pad=0, ln_c1=c1, ln_stats_in=None if fold2 else st2)
            else:
                q = self.conv(ln_of(h2, ".norm2"), self.w(tb + ".attn2.to_q.weight"), shp, taps=1, pad=0)
            self.rec(self._xattn, site, q.view(b, h * w_, c), att.view(b, h * w_, c))
            self.pool.put(q)
            h3 = self.conv(att, self.w(tb + ".attn2.to_out.0.weight"), shp,
                           bias=self.f(tb + ".attn2.to_out.0.bias"), residual=h2, taps=1, pad=0,
                           ln_stats=ext and not fold3 and not fused_tail)
            st3 = self.ln_partials.get(h3.data_ptr())
        self.pool.put(h2, att)
        if fused_tail:               # norm3 -> GEGLU -> FF-out + h3 -> proj_out + x in one launch (csrc/ffn_block.hip)
            def _tail():
                u = self.prefix
                st, b1p = pack_ffn_stream(self.sd[u + tb + ".ff.net.0.proj.weight"], self.sd[u + tb + ".ff.net.0.proj.bias"],
                                          self.sd[u + tb + ".ff.net.2.weight"], self.sd[u + site + ".proj_out.weight"])
                return self.dev(st), self.dev(b1p)
            stream, b1p = self.cached((self.prefix + tb, "ffn_stream"), _tail)
            if self.wcache is not None:
                self.keep += [stream, b1p]
            out = self.pool.get(shp)
            nchunk = (h * w_) // 32
            gkw = {}
            if GN_FROM_EPILOGUE and nchunk <= 128 and (h * w_) * (c // 32) * 2 > GN_FUSED_MAX_BYTES:
                ws = self.be.zeros((b * nchunk * GROUPS * 2,), F32)
                self.keep.append(ws)
                self.gn_partials[out.data_ptr()] = (ws, nchunk)
                gkw = dict(gn_ws=ws, gn_nchunk=nchunk)
            if ln_box[0] is not None:
                self.pool.put(ln_box[0])
            self.rec(self.be.ffn_block, h3.view(b, h * w_, c), stream, self.f(tb + ".norm3.weight"), self.f(tb + ".norm3.bias"),
                     b1p, self.f(tb + ".ff.net.2.bias"), self.f(site + ".proj_out.bias"), x.view(b, h * w_, c),
                     out.view(b, h * w_, c), **gkw)
            self.pool.put(h3)
            return out
        # GEGLU feed-forward
        if fold3 or st3 is not None:
            wf, c1, bf = self._ln_linear(tb, ".norm3", "geglu", [".ff.net.0.proj.weight"], ".ff.net.0.proj.bias", geglu=True)
            ff = self.conv(h3, wf, (b, h, w_, 4 * c), bias=bf, taps=1, pad=0, flags=L.EPI_GEGLU, ln_c1=c1,
                           ln_stats_in=None if fold3 else st3)
        else:
            def _geglu():
                wf, bf = geglu_interleave(self.sd[self.prefix + tb + ".ff.net.0.proj.weight"],
                                          self.sd[self.prefix + tb + ".ff.net.0.proj.bias"])
                return self.dev(wf, F16), self.dev(bf.float())
            wf, bf = self.cached((self.prefix + tb, "geglu"), _geglu)
            if self.wcache is not None:
                self.keep += [wf, bf]
            ff = self.conv(ln_of(h3, ".norm3"), wf, (b, h, w_, 4 * c), bias=bf, taps=1, pad=0, flags=L.EPI_GEGLU)
        ln = ln_box[0]
        self.pool.put(ln)
        h4 = self.conv(ff, self.w(tb + ".ff.net.2.weight"), shp, bias=self.f(tb + ".ff.net.2.bias"),
                       residual=h3, taps=1, pad=0)
        self.pool.put(ff, h3)
        out = self.conv(h4, self.w(site + ".proj_out.weight"), shp, bias=self.f(site + ".proj_out.bias"),
                        residual=x, taps=1, pad=0, gn_stats=True)
        self.pool.put(h4)
        return out

    def _xattn(self, site, q, out):
        mode = L.XATTN_SPLIT if self.gates_mode else L.XATTN_BASELINE
        kv = self.kv[site][self.kv_slot]
        self.be.tri_xattn(q, kv.view(self.B, self.T, -1), out, self.gates[site], self.lam, mode, HEADS,
                          lam_dev=self.params[0:1] if (self.params_dev and self.gates_mode) else None)

    def _emit_eps(self, x, w, bias):
        if self.ddim_coef is not None:      # sampler step without CFG: conv_out and the DDIM update in one launch
            self.be.conv_out_ddim(x, w, bias, self.lat_in, self.ddim_coef)
        else:
            self.be.conv_cout4(x, w, bias, self.eps_out[self.kv_slot], 0)

    def _build(self):
        b, s = self.B, self.S
        h = self.pool.get((b, s, s, 320))
        gkw = {}
        if GN_FROM_EPILOGUE and (s * s) % 256 == 0 and (s * s) // 256 <= 128 and s * s * 10 * 2 > GN_FUSED_MAX_BYTES:
            # conv_in writes the GroupNorm chunk partials of its output (256 pixels per chunk): no statistics pass for the
            # first ResNet's norm1 nor for the last up block's skip-concat, and both can normalise inside their 3x3 conv
            nchunk = (s * s) // 256
            ws = self.be.zeros((b * nchunk * GROUPS * 2,), F32)
            self.keep.append(ws)
            self.gn_partials[h.data_ptr()] = (ws, nchunk)
            gkw = dict(gn_ws=ws, gn_nchunk=nchunk)
        self.rec(self.be.conv_in_nchw, self.lat_in, self.w("conv_in.weight", pack_conv_cin8), self.f("conv_in.bias"), h, **gkw)
        skips = [h]
        for i in range(4):
            for j in range(2):
                if i < 3:
                    nxt = self._gn_of(f"down_blocks.{i}.attentions.{j}.norm", 1e-6, 0)
                else:
                    nxt = self._gn_of("down_blocks.3.resnets.1.norm1" if j == 0 else "mid_block.resnets.0.norm1", 1e-5, 1)
                hn = self._resnet(f"down_blocks.{i}.resnets.{j}", h, next_gn=nxt)
                if h is not skips[-1]:
                    self.pool.put(h)
                h = hn
                if i < 3:
                    hn = self._transformer(f"down_blocks.{i}.attentions.{j}", h)
                    self.pool.put(h)
                    h = hn
                skips.append(h)
            if i < 3:
                c = h.shape[-1]
                h = self.conv(h, self.w(f"down_blocks.{i}.downsamplers.0.conv.weight"),
                              (b, h.shape[1] // 2, h.shape[2] // 2, c),
                              bias=self.f(f"down_blocks.{i}.downsamplers.0.conv.bias"), stride=2, pad=1, gn_stats=True,
                              fuse_gn=self._gn_of(f"down_blocks.{i + 1}.resnets.0.norm1", 1e-5, 1))
                skips.append(h)
        hn = self._resnet("mid_block.resnets.0", h, next_gn=self._gn_of("mid_block.attentions.0.norm", 1e-6, 0))        # h is skips[-1]: keep
        h = hn
        hn = self._transformer("mid_block.attentions.0", h)
        self.pool.put(h)
        h = self._resnet("mid_block.resnets.1", hn)
        self.pool.put(hn)
        for i in range(4):
            for j in range(3):
                skip = skips.pop()
                hn = self._resnet(f"up_blocks.{i}.resnets.{j}", h, skip,
                                  next_gn=self._gn_of(f"up_blocks.{i}.attentions.{j}.norm", 1e-6, 0) if i > 0 else None)
                self.pool.put(h, skip)
                h = hn
                if i > 0:
                    hn = self._transformer(f"up_blocks.{i}.attentions.{j}", h)
                    self.pool.put(h)
                    h = hn
            if i < 3:
                c = h.shape[-1]
                hn = self.conv(h, self.w(f"up_blocks.{i}.upsamplers.0.conv.weight"),
                               (b, h.shape[1] * 2, h.shape[2] * 2, c),
                               bias=self.f(f"up_blocks.{i}.upsamplers.0.conv.bias"), ups=1, pad=1)
                self.pool.put(h)
                h = hn
        g = self.gn(h, None, self.f("conv_norm_out.weight"), self.f("conv_norm_out.bias"), 1e-5, 1)
        self.rec(self._emit_eps, g, self.w("conv_out.weight", pack_conv_cout4), self.f("conv_out.bias"))
        self.pool.put(h, g)
        assert not self.gn_ready, "a GroupNorm written by a finish kernel was never consumed"
        self._insert_weight_prefetch()

    def _insert_weight_prefetch(self):
        """Every layer's weights are cold when its kernel starts (1.76 GB per step against 256 MB of Infinity Cache).  A
        launch whose weights are at least WEIGHT_PREFETCH_MIN_BYTES gets a ``be.prefetch`` of them WEIGHT_PREFETCH_AHEAD
        weight-bearing launches earlier - a side branch of the captured graph (``dadd_prefetch``) - and the plan ends with
        the join.  Reads only: results are unchanged."""
        if not WEIGHT_PREFETCH_AHEAD:
            return
        wops = []                       # (op index, weight tensor) of the launches that stream a weight operand
        for i, (fn, a, k) in enumerate(self.ops):
            name = getattr(fn, "__name__", "")
            if name == "igemm":
                wops.append((i, a[1]))
            elif name in ("ffn_block", "tf_head"):
                wops.append((i, a[1]))
        inserts = []
        for j, (i, w) in enumerate(wops):
            if w.numel() * w.element_size() < WEIGHT_PREFETCH_MIN_BYTES or j < WEIGHT_PREFETCH_AHEAD:
                continue
            inserts.append((wops[j - WEIGHT_PREFETCH_AHEAD][0], w))
        for at, w in sorted(inserts, key=lambda t: -t[0]):       # back to front: earlier indices stay valid
            self.ops.insert(at, (self.be.prefetch, (w,), {}))
        if inserts:
            self.ops.append((self.be.prefetch_join, (), {}))

    # -- step-invariant preparation ----------------------------------------------------------------
    def set_cond(self, cond: torch.Tensor, slot: int = 0):
        """Project the conditioning tokens through to_k/to_v(/to_k_dis/to_v_dis) of all 16 sites."""
        if cond.dim() == 2:
            cond = cond.unsqueeze(1)
        if cond.dim() != 3:
            raise ValueError(f"cond_embed must have shape (B, D) or (B, seq_len, D), got {tuple(cond.shape)}")
        if cond.shape[0] != self.B or cond.shape[1] != self.T or cond.shape[2] != 768:
            raise ValueError(f"cond_embed must be ({self.B}, {self.T}, 768) for this plan, got {tuple(cond.shape)}")
        self.be.copy_(self.cond16, cond)
        for site, _ in self.sites:
            self.be.igemm(self.cond16, self.kv_w[site], self.kv[site][slot], taps=1, pad=0)
        self.cond_gen += 1
        if slot == 0:
            self._a2_dirty = True

    def prepare_attn2(self, lam: float):
        """Fold W_q / W_o and the gates into the conditioning of the fused attn2 sites (once per run;
        never inside a graph capture: a no-op when neither the conditioning nor lambda changed)."""
        if not self.a2 or (not self._a2_dirty and self._a2_lam == float(lam)):
            return
        B = self.B
        with self.be.ctx(), torch.no_grad():
            for grp in self.a2_groups:
                names, c = grp["sites"], grp["c"]
                ns, d = len(names), c // HEADS
                kv = torch.stack([self.kv[n][0].view(B, self.T, 4 * c) for n in names]).float()     # s b t 4c
                hd = lambda t: t.reshape(ns, B, 16, HEADS, d)                     # noqa: E731
                ks = [hd(kv[:, :, 16:32, 0:c]), hd(kv[:, :, 0:16, 2 * c:3 * c]), hd(kv[:, :, 32:48, 2 * c:3 * c])]
                vs = [hd(kv[:, :, 16:32, c:2 * c]), hd(kv[:, :, 0:16, 3 * c:4 * c]), hd(kv[:, :, 32:48, 3 * c:4 * c])]
                g = grp["gates"]                                                  # [s][anat, dis]
                gp = [g[:, 0].view(ns, 1, 1, 1, 1), g[:, 1].view(ns, 1, 1, 1, 1),
                      torch.full((), float(lam), device=g.device)]                # (a fill, not an upload: no host wait)
                scale = math.log2(math.e) / math.sqrt(d)
                m = torch.stack([torch.einsum("sbthd,shdc->sbhtc", k, grp["wq"]) for k in ks], dim=3) * scale   # s b h p t c
                v = torch.stack([torch.einsum("snhd,sbthd->sbnht", grp["wo"], x) * gg for x, gg in zip(vs, gp)], dim=4)  # s b n h p t
                if float(lam) == 0.0:      # routing_gates.py:160,177-178: the delta pathway is skipped, not scaled —
                    m[:, :, :, 2] = 0.0    # zero scores and zero values: garbage (NaN) delta tokens cannot leak
                    v[:, :, :, :, 2] = 0.0
                m = m.reshape(ns, B, 384, c)
                folds = [bool(self.a2[n]["fold"]) for n in names]
                if any(folds):             # LayerNorm 2 folded in: S = rstd (x (gamma o M)^T - mu c1) + M beta
                    assert all(folds), "fused attn2 sites of one width fold norm2 alike"
                    grp["ln_d"].copy_((m * grp["beta"]).sum(dim=-1))
                    m = m * grp["gamma"]
                grp["mcat"].copy_(m)
                if any(folds):             # c1 sums the ROUNDED rows, so that the mean term cancels exactly
                    grp["ln_c1"].copy_(grp["mcat"].float().sum(dim=-1))
                grp["vw"].copy_(v.reshape(ns, B, c, 384))
        self._a2_dirty, self._a2_lam = False, float(lam)

    def time_rows(self, t: torch.Tensor, out: torch.Tensor):
        """rows[m] = cat_r time_emb_proj_r(SiLU(MLP(sinusoid(t[m]))))  for a vector of timesteps."""
        m = t.shape[0]
        feat = self.be.empty((m, 320), F32)
        t1 = self.be.empty((m, 1280), F32)
        t2 = self.be.empty((m, 1280), F32)
        self.be.timestep_features(t, feat)
        self.be.linear_rows(feat, self.w_t1, self.b_t1, t1, 0, 1)      # Linear -> SiLU
        self.be.linear_rows(t1, self.w_t2, self.b_t2, t2, 0, 0)        # Linear  (= temb)
        self.be.linear_rows(t2, self.w_tp, self.b_tp, out, 1, 0)       # SiLU -> 22 x Linear
        return t2

    def run_slot(self, slot: int, lam: float):
        self.kv_slot, self.lam = slot, lam
        self.run()

    def forward(self, latents: torch.Tensor, t: torch.Tensor, cond: Optional[torch.Tensor],
                lam: float = 0.0) -> torch.Tensor:
        """General ``module(latents, t, cond)`` call: eps for arbitrary per-sample timesteps.
        ``cond=None`` reuses the projections of the previous ``set_cond``."""
        if tuple(latents.shape) != tuple(self.lat_in.shape):
            raise ValueError(f"latents must be {tuple(self.lat_in.shape)}, got {tuple(latents.shape)}")
        if t.dim() == 0:
            t = t[None]
        t = t.reshape(-1).long()
        if t.shape[0] == 1:
            t = t.expand(self.B)
        if t.shape[0] != self.B:
            raise ValueError(f"timesteps must have {self.B} entries, got {t.shape[0]}")
        if cond is not None:
            self.set_cond(cond, 0)
        self.time_rows(self.be.to_device(t.contiguous()), self.temb_rows)
        self.be.copy_(self.lat_in, latents.float())
        self.prepare_attn2(lam)
        self.run_slot(0, lam)
        return self.be.clone(self.eps_out[0])


# ----------------------------------------------------------------------------- VAE decoder
class VaeDecoderPlan(_Plan):
    def __init__(self, be, sd, batch: int, side: int, *, prefix="vae.vae", latent_scale=0.18215):
        super().__init__(be)
        self.B, self.S = batch, side
        self.sd, self.prefix = sd, prefix + "."
        self.gn_ws = be.empty((batch * L.GN_MAX_CHUNKS * GROUPS * 2,), F32)
        self.z_in = be.zeros((batch, 4, side, side), F32)
        self.img_out = be.zeros((batch, 3, side * 8, side * 8), F32)
        self.inv_scale = 1.0 / latent_scale
        self._build()

    def w(self, key, pack=pack_conv):
        return self.dev(pack(self.sd[self.prefix + key]))

    def f(self, key):
        return self.dev(self.sd[self.prefix + key].float())

    def _res(self, name, x):
        b, h, w_, cin = x.shape
        cout = self.sd[self.prefix + name + ".conv1.weight"].shape[0]
        g1 = self.gn(x, None, self.f(name + ".norm1.weight"), self.f(name + ".norm1.bias"), 1e-6, 1)
        h1 = self.conv(g1, self.w(name + ".conv1.weight"), (b, h, w_, cout), bias=self.f(name + ".conv1.bias"),
                       gn_stats=True)
        self.pool.put(g1)
        g2 = self.gn(h1, None, self.f(name + ".norm2.weight"), self.f(name + ".norm2.bias"), 1e-6, 1)
        self.pool.put(h1)
        res = x
        if cin != cout:
            res = self.conv(x, self.w(name + ".conv_shortcut.weight"), (b, h, w_, cout),
                            bias=self.f(name + ".conv_shortcut.bias"), taps=1, pad=0)
        out = self.conv(g2, self.w(name + ".conv2.weight"), (b, h, w_, cout),
                        bias=self.f(name + ".conv2.bias"), residual=res, gn_stats=True)
        self.pool.put(g2)
        if res is not x:
            self.pool.put(res)
        return out

    def _attn(self, name, x):
        b, h, w_, c = x.shape
        g = self.gn(x, None, self.f(name + ".group_norm.weight"), self.f(name + ".group_norm.bias"), 1e-6, 0)
        wqkv = self.dev(torch.cat([self.sd[self.prefix + name + f".to_{n}.weight"] for n in "qkv"]), F16)
        bqkv = self.dev(torch.cat([self.sd[self.prefix + name + f".to_{n}.bias"] for n in "qkv"]).float())
        qkv = self.conv(g, wqkv, (b, h, w_, 3 * c), bias=bqkv, taps=1, pad=0)
        self.pool.put(g)
        att = self.pool.get((b, h, w_, c))
        self.rec(self.be.self_attn, qkv.view(b, h * w_, 3 * c), att.view(b, h * w_, c), 1)
        self.pool.put(qkv)
        out = self.conv(att, self.w(name + ".to_out.0.weight"), (b, h, w_, c),
                        bias=self.f(name + ".to_out.0.bias"), residual=x, taps=1, pad=0, gn_stats=True)
        self.pool.put(att)
        return out

    def _build(self):
        b, s = self.B, self.S
        d = "decoder."
        z8 = self.pool.get((b, s, s, 8))
        pq_w = self.dev(self.sd[self.prefix + "post_quant_conv.weight"].reshape(4, 4).float())
        pq_b = self.f("post_quant_conv.bias")
        self.rec(self.be.pack_latents, self.z_in, z8, self.inv_scale, pq_w, pq_b)
        h = self.pool.get((b, s, s, 512))
        self.rec(self.be.conv_cin8, z8, self.w(d + "conv_in.weight", pack_conv_cin8), self.f(d + "conv_in.bias"), h)
        for blk in (lambda x: self._res(d + "mid_block.resnets.0", x),
                    lambda x: self._attn(d + "mid_block.attentions.0", x),
                    lambda x: self._res(d + "mid_block.resnets.1", x)):
            hn = blk(h)
            self.pool.put(h)
            h = hn
        for i in range(4):
            for j in range(3):
                hn = self._res(d + f"up_blocks.{i}.resnets.{j}", h)
                self.pool.put(h)
                h = hn
            if i < 3:
                c = h.shape[-1]
                hn = self.conv(h, self.w(d + f"up_blocks.{i}.upsamplers.0.conv.weight"),
                               (b, h.shape[1] * 2, h.shape[2] * 2, c),
                               bias=self.f(d + f"up_blocks.{i}.upsamplers.0.conv.bias"), ups=1, pad=1, gn_stats=True)
                self.pool.put(h)
                h = hn
        g = self.gn(h, None, self.f(d + "conv_norm_out.weight"), self.f(d + "conv_norm_out.bias"), 1e-6, 1)
        self.rec(self.be.conv_cout4, g, self.w(d + "conv_out.weight", pack_conv_cout4),
                 self.f(d + "conv_out.bias"), self.img_out, 1)
        self.pool.put(h, g)


# ----------------------------------------------------------------------------- VAE encoder
class VaeEncoderPlan(VaeDecoderPlan):
    """``SDVAE.encode`` (src/models/vae/vae.py:71-88) -> diffusers ``AutoencoderKL.encode``: images (B,3,H,W) in
    [-1,1] -> moments (mean, logvar), each (B,4,H/8,W/8) fp32 NCHW.  Same kernels as the decoder; the three
    downsamplers are 3x3 / stride-2 convolutions with the asymmetric (0,1,0,1) padding folded into the gather
    (pad = 0: the window runs one pixel past the bottom / right edge, where the gather returns zeros), and
    ``quant_conv`` (1x1, 8 -> 8) is composed into ``conv_out`` at plan time (two linear maps: exact algebra, one
    rounding of the composed weights to fp16).  ``side`` is the LATENT side (image side / 8)."""

    def __init__(self, be, sd, batch: int, side: int, *, prefix="vae.vae", wcache=None):
        _Plan.__init__(self, be, wcache)
        self.B, self.S = batch, side
        self.sd, self.prefix = sd, prefix + "."
        self.gn_ws = be.empty((batch * L.GN_MAX_CHUNKS * GROUPS * 2,), F32)
        self.img_in = be.zeros((batch, 3, side * 8, side * 8), F32)
        self.mean = be.zeros((batch, 4, side, side), F32)
        self.logvar = be.zeros((batch, 4, side, side), F32)
        self._build()

    def _build(self):
        b, s = self.B, self.S * 8
        e = "encoder."
        x8 = self.pool.get((b, s, s, 8))
        self.rec(self.be.pack_latents, self.img_in, x8)
        h = self.pool.get((b, s, s, VAE_CH[0]))
        self.rec(self.be.conv_cin8, x8, self.w(e + "conv_in.weight", pack_conv_cin8), self.f(e + "conv_in.bias"), h)
        self.pool.put(x8)
        for i in range(4):
            for j in range(2):
                hn = self._res(e + f"down_blocks.{i}.resnets.{j}", h)
                self.pool.put(h)
                h = hn
            if i < 3:
                c = h.shape[-1]
                hn = self.conv(h, self.w(e + f"down_blocks.{i}.downsamplers.0.conv.weight"),
                               (b, h.shape[1] // 2, h.shape[2] // 2, c),
                               bias=self.f(e + f"down_blocks.{i}.downsamplers.0.conv.bias"), stride=2, pad=0,
                               gn_stats=True)
                self.pool.put(h)
                h = hn
        for blk in (lambda x: self._res(e + "mid_block.resnets.0", x),
                    lambda x: self._attn(e + "mid_block.attentions.0", x),
                    lambda x: self._res(e + "mid_block.resnets.1", x)):
            hn = blk(h)
            self.pool.put(h)
            h = hn
        g = self.gn(h, None, self.f(e + "conv_norm_out.weight"), self.f(e + "conv_norm_out.bias"), 1e-6, 1)
        # moments = quant_conv(conv_out(g)):  W' = Q W,  b' = Q b + q   (Q: 8x8 of the 1x1 conv)
        u = self.prefix
        q = self.sd[u + "quant_conv.weight"].reshape(8, 8).double()
        w = self.sd[u + e + "conv_out.weight"].double()
        wq = torch.einsum("oc,cikl->oikl", q, w).float()
        bq = (q @ self.sd[u + e + "conv_out.bias"].double() + self.sd[u + "quant_conv.bias"].double()).float()
        self.rec(self.be.conv_cout4, g, self.dev(pack_conv_cout4(wq[:4])), self.dev(bq[:4].contiguous()), self.mean, 0)
        self.rec(self.be.conv_cout4, g, self.dev(pack_conv_cout4(wq[4:])), self.dev(bq[4:].contiguous()), self.logvar, 2)
        self.pool.put(h, g)


# ----------------------------------------------------------------------------- DDIM loop
class DdimLoop:
    """Deterministic (eta = 0) DDIM loop over a UNetPlan: ONE captured step graph, replayed."""

    def __init__(self, unet: UNetPlan, max_steps: int = 1000):
        self.u = unet
        self.be = unet.be
        self.step = self.be.zeros((2,), torch.int32)       # row of the next step, block ticket of dadd_begin_step
        self.cur_coef = self.be.zeros((4,), F32)
        self.graphs: Dict[Tuple, object] = {}
        self.cap = 0
        self.table = None
        self.coef = None
        self.nsteps = 0
        self._prepared = None           # (timestep grid, schedule identity) the tables on the device were built for
        self._params = None             # (lambda, guidance) held by ``unet.params`` on the device
        self._reserve(64)

    def _reserve(self, n: int):
        if n <= self.cap:
            return
        for g in self.graphs.values():       # captured graphs hold the old table addresses
            self.be.graph_destroy(g)
        self.graphs.clear()
        self.cap = n
        self._prepared = None
        self.table = self.be.zeros((n, self.u.temb_cols), F32)
        self.coef = self.be.zeros((n, 4), F32)

    def prepare(self, timesteps: torch.Tensor, alphas_cumprod: torch.Tensor):
        """Coefficient rows computed on the host in fp32 exactly as the reference does per step
        (inference_pipeline_ip.py:434-450), and the time-embedding rows of every step.  Both depend on the timestep
        grid and the weights only: a repeated grid (every pass of a sweep) reuses the tables on the device — no
        device-to-host read of the grid, no upload, nothing that makes the host wait for the GPU, so the next pass is
        queued while the current one runs.  Callers on the hot path hand in a CPU grid (``torch.linspace`` gives the
        same integers on both devices: tests/test_gpu_parity.py::test_ddim_timestep_grid_on_device)."""
        ts = timesteps.detach()
        ts = (ts if ts.device.type == "cpu" else ts.cpu()).long()
        n = ts.shape[0]
        key = (tuple(ts.tolist()), alphas_cumprod.data_ptr(), alphas_cumprod._version)
        if key == self._prepared and n <= self.cap:
            self.nsteps = n
            return
        self._reserve(n)
        ac = alphas_cumprod.detach().cpu().float()
        coef = torch.empty(n, 4, dtype=F32)
        for i in range(n):
            a_t = ac[int(ts[i])]
            coef[i, 0], coef[i, 1] = torch.sqrt(a_t), torch.sqrt(1.0 - a_t)
            if i == n - 1:
                coef[i, 2], coef[i, 3] = -1.0, 0.0          # last step returns x0 (:441-443)
            else:
                a_p = ac[int(ts[i + 1])]
                coef[i, 2], coef[i, 3] = torch.sqrt(a_p), torch.sqrt(1.0 - a_p)
        self.be.copy_(self.coef[:n], self.be.to_device(coef))
        self.u.time_rows(self.be.to_device(ts), self.table[:n])
        self.nsteps = n
        self._prepared = key

    def set_params(self, lam: float, guidance: float):
        """lambda and the CFG scale live in device memory (``unet.params``): the captured step reads them there, so
        ONE graph per (cfg on/off) serves every (lambda, guidance) — the reference reads ``delta_scale`` per call
        (attention_processor_routing_gates.py:160).  The fused attn2 sites fold lambda into their step-invariant
        conditioning (``prepare_attn2``), which is device memory too."""
        if self._params != (float(lam), float(guidance)):      # (an upload from pageable memory blocks the host)
            self.be.copy_(self.u.params, torch.tensor([float(lam), float(guidance)], dtype=F32))
            self._params = (float(lam), float(guidance))
        self.u.prepare_attn2(lam)

    def sample(self, latents: torch.Tensor, timesteps: torch.Tensor, alphas_cumprod: torch.Tensor, lam: float,
               do_cfg: bool = False, guidance: float = 1.0, use_graph: bool = True,
               trace: Optional[list] = None) -> torch.Tensor:
        """All prepared steps from ``latents`` -> final latents, WITH both stream hand-offs: the backend stream waits
        for whatever torch's current stream has queued (conditioning caches written by torch ops, the latents), and
        the current stream waits for the returned tensor.  Callers that drive ``prepare`` / ``run`` by hand owe the
        same two calls (``be.wait_current()`` before, ``be.release_to_current()`` after)."""
        be = self.be
        be.wait_current()
        self.prepare(timesteps, alphas_cumprod)
        be.copy_(self.u.lat_in, latents)
        self.run(lam, do_cfg, guidance, use_graph=use_graph, trace=trace)
        out = be.clone(self.u.lat_in)
        be.release_to_current()
        return out

    def _one_step(self, lam: float, do_cfg: bool, guidance: float, keep_eps: bool = False):
        """``keep_eps``: eps of the step stays readable in ``unet.eps_out`` (traces); otherwise, without CFG, conv_out
        applies the DDIM update itself (``dadd_conv_out_ddim_f16``) and eps is never stored."""
        u = self.u
        u.params_dev = True
        fused = not do_cfg and not keep_eps
        try:
            self.be.begin_step(self.table, u.temb_rows, self.coef, self.cur_coef, self.step)
            u.ddim_coef = self.cur_coef if fused else None
            u.run_slot(0, lam)
            if do_cfg:
                u.run_slot(1, lam)
            if not fused:
                self.be.ddim_update(u.lat_in, u.eps_out[0], u.eps_out[1] if do_cfg else None, guidance,
                                    self.cur_coef, guidance_dev=u.params[1:2])
        finally:
            u.params_dev = False
            u.ddim_coef = None

    def run(self, lam: float, do_cfg: bool = False, guidance: float = 1.0, use_graph: bool = True,
            trace: Optional[list] = None):
        """Runs all prepared steps in place on ``unet.lat_in``."""
        self.be.zero_(self.step)
        self.set_params(lam, guidance)
        if trace is not None or not use_graph:
            for _ in range(self.nsteps):
                self._one_step(lam, do_cfg, guidance, keep_eps=trace is not None)
                if trace is not None:
                    trace.append((self.be.clone(self.u.eps_out[0]), self.be.clone(self.u.lat_in)))
            return
        key = bool(do_cfg)
        g = self.graphs.get(key)
        if g is None:
            self.be.synchronize()
            self.be.graph_begin()
            try:
                self._one_step(lam, do_cfg, guidance)
            finally:
                g = self.be.graph_end()
            self.graphs[key] = g
        for _ in range(self.nsteps):
            self.be.graph_launch(g)
