"""Operator backend: torch tensors in, C-ABI kernel launches out.

``HipBackend`` is the only backend the product ships.  Every method takes device tensors whose
memory torch owns, validates what the kernels assume, and launches on the backend's stream.
(The engine takes the backend as a constructor argument so that the CPU test-suite can check the
*wiring* of the plan with a reference implementation that lives under ``tests/``; the product
never constructs anything but ``HipBackend`` and fails loudly when the library is missing.)
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import lib as L


def _p(t: Optional[torch.Tensor]):
    """Device pointer of a tensor argument.  A host tensor here would hand the kernel a host address (a GPU memory
    fault, which can take the whole node down): refuse it as an ordinary Python error instead."""
    if t is None:
        return None
    if t.device.type != "cuda":
        raise ValueError(f"kernel argument lives on {t.device}, not on the GPU (shape {tuple(t.shape)}, {t.dtype})")
    return t.data_ptr()


class HipBackend:
    name = "hip-gfx950"

    def __init__(self, device: torch.device):
        if not torch.cuda.is_available():
            raise RuntimeError("HipBackend needs a ROCm device (torch.cuda.is_available() is False)")
        self.lib = L.load()
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self._desc = L.IgemmDesc()
        self._capturing = False     # between graph_begin() and graph_end(): no cross-stream waits may be recorded
        self._prof_on = False
        with torch.cuda.device(self.device):
            L.check(self.lib.dadd_init())

    # ------------------------------------------------------------------ plumbing
    @property
    def s(self):
        return self.stream.cuda_stream

    def ctx(self):
        """All torch-side work of the engine (allocation, copy_, zero_) runs on the backend stream,
        so the caching allocator and the kernels agree on one stream order."""
        return torch.cuda.stream(self.stream)

    def empty(self, shape, dtype):
        with self.ctx():
            return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        with self.ctx():
            return torch.zeros(shape, dtype=dtype, device=self.device)

    def _after_producer(self, src: torch.Tensor):
        """A device tensor handed in from outside was produced (or is still being produced) on torch's CURRENT
        stream; the backend stream must not read it earlier, and the caching allocator must not hand its block to
        another current-stream allocation while the backend stream still reads it."""
        if src.device.type == "cuda" and not self._capturing:
            cur = torch.cuda.current_stream(self.device)
            if cur != self.stream:
                self.stream.wait_stream(cur)
                src.record_stream(self.stream)

    def to_device(self, t: torch.Tensor, dtype=None):
        self._after_producer(t)
        with self.ctx():
            return t.to(device=self.device, dtype=dtype or t.dtype).contiguous()

    def copy_(self, dst: torch.Tensor, src: torch.Tensor):
        self._after_producer(src)
        with self.ctx():
            dst.copy_(src.reshape(dst.shape))

    def zero_(self, t: torch.Tensor):
        with self.ctx():
            t.zero_()

    def clone(self, t: torch.Tensor):
        """Copy made on the backend stream; the caller reads it on torch's current stream only after
        ``release_to_current()`` (or ``synchronize()``)."""
        with self.ctx():
            return t.detach().clone()

    def synchronize(self):
        self.stream.synchronize()

    def wait_current(self):
        """Order this backend's stream after torch's current stream (inputs produced by torch ops)."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def release_to_current(self):
        torch.cuda.current_stream(self.device).wait_stream(self.stream)

    # ------------------------------------------------------------------ ops
    def pack_latents(self, x, out, scale=1.0, mat=None, vec=None):
        b, c, h, w = x.shape
        assert x.dtype == torch.float32 and out.dtype == torch.float16 and out.shape == (b, h, w, 8)
        L.check(self.lib.dadd_pack_nchw_f32_to_nhwc8_f16(_p(x), _p(out), b, c, h, w, float(scale),
                                                         _p(mat), _p(vec), self.s))

    def conv_cin8(self, x, w, bias, out):
        b, h, wd, c8 = x.shape
        assert c8 == 8 and w.shape[1:] == (9, 8) and out.shape == (b, h, wd, w.shape[0])
        L.check(self.lib.dadd_conv3x3_cin8_f16(_p(x), _p(w), _p(bias), _p(out), b, h, wd, w.shape[0],
                                               self.s))

    def conv_in_nchw(self, x, w, bias, out, gn_ws=None, gn_nchunk=0):
        """fp32 NCHW latents (<= 4 channels) -> fp16 NHWC: pack_latents + conv_cin8 in one launch; with ``gn_ws`` also
        the GroupNorm chunk partials of the output (256 pixels per chunk)."""
        b, c, h, wd = x.shape
        assert x.dtype == torch.float32 and c <= 4 and w.shape[1:] == (9, 8) and out.shape == (b, h, wd, w.shape[0])
        assert gn_ws is None or (gn_ws.dtype == torch.float32 and gn_ws.numel() >= b * gn_nchunk * 64)
        L.check(self.lib.dadd_conv_in_nchw_f16(_p(x), _p(w), _p(bias), _p(out), b, c, h, wd, w.shape[0],
                                               _p(gn_ws), gn_nchunk, self.s))

    def conv_cout4(self, x, w, bias, out, mode=0):
        b, h, wd, c = x.shape
        co = w.shape[0]
        assert w.shape == (co, 9, c) and out.shape == (b, co, h, wd) and out.dtype == torch.float32
        L.check(self.lib.dadd_conv3x3_cout4_f16(_p(x), _p(w), _p(bias), _p(out), b, h, wd, c, co,
                                                int(mode), self.s))

    def conv_out_ddim(self, x, w, bias, latents, coef):
        """conv_out fused with the DDIM update: ``latents`` (fp32 NCHW) are stepped in place, eps is not stored."""
        b, h, wd, c = x.shape
        co = w.shape[0]
        assert w.shape == (co, 9, c) and latents.shape == (b, co, h, wd) and latents.dtype == torch.float32 \
            and coef.numel() == 4 and coef.dtype == torch.float32
        L.check(self.lib.dadd_conv_out_ddim_f16(_p(x), _p(w), _p(bias), _p(latents), _p(coef), b, h, wd, c, co, self.s))

    def q_sample(self, x0, noise, t, alphas_cumprod, out):
        b = x0.shape[0]
        assert x0.dtype == torch.float32 and t.dtype == torch.int64 and t.shape == (b,) and noise.shape == x0.shape
        L.check(self.lib.dadd_q_sample_f32(_p(x0), _p(noise), _p(t), _p(alphas_cumprod), _p(out), b,
                                           x0.numel() // b, self.s))

    def mse_rows(self, pred, target, out):
        b = pred.shape[0]
        assert pred.shape == target.shape and out.shape == (b,) and pred.dtype == torch.float32
        L.check(self.lib.dadd_mse_rows_f32(_p(pred), _p(target), _p(out), b, pred.numel() // b, self.s))

    def frames_to_u8(self, frames, out):
        b, c, h, w = frames.shape
        assert c == 3 and frames.dtype == torch.float32 and out.dtype == torch.uint8 and out.shape == (b, h, w, 3)
        L.check(self.lib.dadd_frames_to_u8(_p(frames), _p(out), b, h, w, self.s))

    def gaussian_sample(self, mean, logvar, noise, out, scale=1.0):
        assert mean.shape == logvar.shape == noise.shape == out.shape and mean.dtype == torch.float32
        L.check(self.lib.dadd_gaussian_sample_f32(_p(mean), _p(logvar), _p(noise), float(scale), _p(out),
                                                  mean.numel(), self.s))

    def igemm(self, x, w, out, *, x2=None, bias=None, rowvec=None, residual=None, taps=1, stride=1,
              ups=0, pad=0, flags=0, splitk=1, partial=None, tile_n=0, tile_m=0, counters=None, ln_c1=None,
              ln_eps=1e-5, gn_ws=None, gn_nchunk=0, ln_stats_out=None, ln_stats_in=None, gn_in=None, gn_apply=None):
        """x [B,Hi,Wi,C1] (x2 [B,Hi,Wi,C2]); w [N, taps*(C1+C2)]; out [B,Ho,Wo,N] (N/2 for GEGLU).
        ``gn_apply`` = (normalised output [B,Ho,Wo,N], gamma, beta, eps) with EPI_GNAPPLY: the split-K finish kernel also
        writes GroupNorm(out) (+ SiLU with EPI_GNAPPLY_SILU).
        ``gn_in`` = (partials [B*nchunk*64] fp32, nchunk, gamma, beta, eps) with PRE_GN: GroupNorm of x on the way in.
        ``ln_stats_out`` [P][M][2] fp32 (EPI_LNSTAT): row partials of the output, P = N / (tile_n/2);
        ``ln_stats_in`` [P'][M][2] (EPI_LNFOLD): the partials of x written by its producer."""
        b, hi, wi, c1 = x.shape
        c2 = 0 if x2 is None else x2.shape[-1]
        n = w.shape[0]
        ho, wo = out.shape[1], out.shape[2]
        assert w.shape[1] == taps * (c1 + c2), (w.shape, taps, c1, c2)
        assert out.shape[-1] == (n // 2 if flags & L.EPI_GEGLU else n) and out.shape[0] == b
        d = self._desc
        d.x, d.x2, d.w, d.out, d.partial = _p(x), _p(x2), _p(w), _p(out), _p(partial)
        d.bias, d.rowvec, d.residual = _p(bias), _p(rowvec), _p(residual)
        d.B, d.Hi, d.Wi, d.C1, d.C2, d.Ho, d.Wo, d.N = b, hi, wi, c1, c2, ho, wo, n
        d.taps, d.stride, d.ups, d.pad = taps, stride, ups, pad
        d.ldo, d.ldr = out.stride(-2), (residual.stride(-2) if residual is not None else 0)
        d.ld_rowvec = rowvec.stride(0) if rowvec is not None else 0
        d.splitk, d.flags, d.tile_n, d.tile_m = splitk, flags, tile_n, tile_m
        d.counters = _p(counters)
        d.ln_c1, d.ln_eps = _p(ln_c1), float(ln_eps)
        d.gn_ws, d.gn_nchunk, d.gn_cg = _p(gn_ws), int(gn_nchunk), (n // 32 if gn_ws is not None else 0)
        if gn_ws is not None:
            assert flags & L.EPI_GNSTAT and gn_ws.numel() >= b * gn_nchunk * 64 and gn_ws.dtype == torch.float32
        if ln_c1 is not None:
            assert flags & L.EPI_LNFOLD and ln_c1.numel() == n and ln_c1.dtype == torch.float32
        m_rows = b * ho * wo
        d.ln_stats_out, d.ln_stats_in, d.ln_parts_out, d.ln_parts_in = _p(ln_stats_out), _p(ln_stats_in), 0, 0
        if ln_stats_out is not None:
            assert flags & L.EPI_LNSTAT and ln_stats_out.dtype == torch.float32 and ln_stats_out.dim() == 3 \
                and ln_stats_out.shape[1:] == (m_rows, 2) and ln_stats_out.is_contiguous()
            d.ln_parts_out = ln_stats_out.shape[0]
        if ln_stats_in is not None:
            assert flags & L.EPI_LNFOLD and ln_stats_in.dtype == torch.float32 and ln_stats_in.dim() == 3 \
                and ln_stats_in.shape[1:] == (b * hi * wi, 2) and ln_stats_in.is_contiguous()
            d.ln_parts_in = ln_stats_in.shape[0]
        d.gn_in_ws = d.gn_in_gamma = d.gn_in_beta = d.gn_in_ws2 = None
        d.gn_in_nchunk, d.gn_in_eps, d.gn_in_nchunk2 = 0, 0.0, 0
        if gn_in is not None:       # (partials, chunks, gamma, beta, eps[, partials of x2, chunks of x2])
            ws_in, nch_in, gam, bet, eps_in = gn_in[:5]
            assert flags & L.PRE_GN and ws_in.dtype == gam.dtype == bet.dtype == torch.float32 \
                and ws_in.numel() >= b * nch_in * 64 and gam.numel() == c1 + c2 and bet.numel() == c1 + c2
            d.gn_in_ws, d.gn_in_gamma, d.gn_in_beta = _p(ws_in), _p(gam), _p(bet)
            d.gn_in_nchunk, d.gn_in_eps = int(nch_in), float(eps_in)
            if len(gn_in) > 5:
                ws2, nch2 = gn_in[5], gn_in[6]
                assert x2 is not None and ws2.dtype == torch.float32 and ws2.numel() >= b * nch2 * 64
                d.gn_in_ws2, d.gn_in_nchunk2 = _p(ws2), int(nch2)
        d.gn_out = d.gn_out_gamma = d.gn_out_beta = None
        d.gn_out_eps = 0.0
        if gn_apply is not None:
            g_out, gam, bet, eps_o = gn_apply
            assert flags & L.EPI_GNAPPLY and g_out.shape == out.shape and g_out.dtype == torch.float16 and g_out.is_contiguous() \
                and gam.dtype == bet.dtype == torch.float32 and gam.numel() == n and bet.numel() == n
            d.gn_out, d.gn_out_gamma, d.gn_out_beta, d.gn_out_eps = _p(g_out), _p(gam), _p(bet), float(eps_o)
        if partial is not None:
            assert partial.numel() >= splitk * b * ho * wo * n
        L.check(self.lib.dadd_conv_igemm_f16(C.byref(d), self.s))

    def groupnorm(self, x1, x2, gamma, beta, out, ws, groups, eps, silu, ws_chunks=0):
        """``ws_chunks`` > 0: ``ws`` holds the chunk partials written by the producing GEMM's epilogue."""
        b = x1.shape[0]
        hw = x1.shape[1] * x1.shape[2]
        c1 = x1.shape[-1]
        c2 = 0 if x2 is None else x2.shape[-1]
        need = (ws_chunks + (64 if ws_chunks > 128 else 0)) if ws_chunks else L.GN_MAX_CHUNKS
        assert out.shape[-1] == c1 + c2 and ws.numel() >= b * need * groups * 2
        L.check(self.lib.dadd_groupnorm_f16(_p(x1), c1, _p(x2), c2, _p(gamma), _p(beta), _p(out),
                                            _p(ws), b, hw, groups, float(eps), int(silu), int(ws_chunks), self.s))

    def layernorm(self, x, gamma, beta, out, eps=1e-5):
        c = x.shape[-1]
        m = x.numel() // c
        L.check(self.lib.dadd_layernorm_f16(_p(x), _p(gamma), _p(beta), _p(out), m, c, float(eps),
                                            self.s))

    def self_attn(self, qkv, out, heads):
        """qkv [B,N,3C] (q|k|v blocks of C columns); out [B,N,C]."""
        b, n, c3 = qkv.shape
        c = c3 // 3
        base = qkv.data_ptr()
        L.check(self.lib.dadd_self_attn_f16(base, base + 2 * c, base + 4 * c, _p(out), b, n, heads,
                                            c // heads, c3, out.stride(-2), self.s))

    def tri_xattn(self, q, kv, out, gates, lam, mode, heads, lam_dev=None):
        """``lam_dev`` (device float32[1]) overrides ``lam``: lambda is then a device-side parameter."""
        b, n, c = q.shape
        L.check(self.lib.dadd_tri_xattn_f16(_p(q), _p(kv), _p(out), _p(gates), float(lam), _p(lam_dev), int(mode),
                                            b, n, heads, c // heads, kv.shape[1], kv.stride(1),
                                            self.s))

    def attn2_fused(self, x, mcat, vw, bias, residual, out, ln_stats_out=None, ln_stats_in=None, ln_c1=None, ln_d=None,
                    ln_eps=1e-5):
        """x, residual, out [B,HW,C]; mcat [B,384,C]; vw [B,C,384] (include/dadd_hip.h); ``ln_stats_out`` [C/80][B*HW][2]
        fp32: LayerNorm row partials of ``out`` for the linear behind the next LayerNorm.  ``ln_stats_in`` [P][B*HW][2]
        with ``ln_c1`` / ``ln_d`` [B,384]: norm2 folded in (x un-normalised, mcat carrying gamma)."""
        b, hw, c = x.shape
        assert mcat.shape == (b, 384, c) and vw.shape == (b, c, 384) and out.shape == x.shape
        if ln_stats_out is not None:
            assert ln_stats_out.shape == (c // 80, b * hw, 2) and ln_stats_out.dtype == torch.float32 \
                and ln_stats_out.is_contiguous()
        parts = 0
        if ln_stats_in is not None:
            assert ln_stats_in.dim() == 3 and ln_stats_in.shape[1:] == (b * hw, 2) and ln_stats_in.dtype == torch.float32 \
                and ln_stats_in.is_contiguous() and ln_c1.shape == (b, 384) and ln_d.shape == (b, 384) \
                and ln_c1.dtype == ln_d.dtype == torch.float32 and ln_c1.is_contiguous() and ln_d.is_contiguous()
            parts = ln_stats_in.shape[0]
        L.check(self.lib.dadd_attn2_fused_f16(_p(x), _p(mcat), _p(vw), _p(bias), _p(residual), _p(out), _p(ln_stats_out),
                                              _p(ln_stats_in), parts, _p(ln_c1), _p(ln_d), float(ln_eps), b, hw, c, self.s))

    def ffn_block(self, x, stream, ln_g, ln_b, b1, b2, bp, xres, out, gn_ws=None, gn_nchunk=0, ln_eps=1e-5):
        """Transformer-block tail in one launch (csrc/ffn_block.hip): x, xres, out [B,HW,320] fp16; ``stream`` / ``b1``
        from ``engine.pack_ffn_stream``; ``gn_ws`` [B*gn_nchunk*64] fp32 receives GroupNorm partials of ``out``
        (chunks of 32 rows)."""
        b, hw, c = x.shape
        assert out.shape == x.shape == xres.shape and x.dtype == out.dtype == xres.dtype == torch.float16
        assert stream.dtype == torch.float16 and stream.numel() * 2 == self.lib.dadd_ffn_block_bytes()
        assert b1.numel() == 2560 and b2.numel() == c and bp.numel() == c and ln_g.numel() == c and ln_b.numel() == c
        assert all(t.dtype == torch.float32 for t in (ln_g, ln_b, b1, b2, bp))
        assert x.is_contiguous() and xres.is_contiguous() and out.is_contiguous()
        if gn_ws is not None:
            assert gn_ws.dtype == torch.float32 and gn_ws.numel() >= b * gn_nchunk * 64
        L.check(self.lib.dadd_ffn_block_f16(_p(x), _p(stream), _p(ln_g), _p(ln_b), float(ln_eps), _p(b1), _p(b2), _p(bp),
                                            _p(xres), _p(out), _p(gn_ws), int(gn_nchunk), b * hw, hw, c, self.s))

    def tf_head(self, x, stream, gn_ws, gn_nchunk, gn_g, gn_b, bp, ln_g, ln_b, hs, qkv, gn_eps=1e-6, ln_eps=1e-5):
        """Transformer-block head in one launch (csrc/tf_head.hip): x, hs [B,HW,320], qkv [B,HW,960] fp16; ``gn_ws`` the
        producer's GroupNorm chunk partials of x ([B*gn_nchunk*64] fp32); ``stream`` from ``engine.pack_head_stream``."""
        b, hw, c = x.shape
        assert hs.shape == x.shape and qkv.shape == (b, hw, 3 * c) and x.dtype == hs.dtype == qkv.dtype == torch.float16
        assert stream.dtype == torch.float16 and stream.numel() * 2 == self.lib.dadd_tf_head_bytes()
        assert gn_ws.dtype == torch.float32 and gn_ws.numel() >= b * gn_nchunk * 64
        assert all(t.dtype == torch.float32 and t.numel() == c for t in (gn_g, gn_b, bp, ln_g, ln_b))
        assert x.is_contiguous() and hs.is_contiguous() and qkv.is_contiguous()
        L.check(self.lib.dadd_tf_head_f16(_p(x), _p(stream), _p(gn_ws), int(gn_nchunk), _p(gn_g), _p(gn_b), float(gn_eps),
                                          _p(bp), _p(ln_g), _p(ln_b), float(ln_eps), _p(hs), _p(qkv), b * hw, hw, c, self.s))

    def timestep_features(self, t, out):
        assert t.dtype == torch.int64 and out.dtype == torch.float32
        L.check(self.lib.dadd_timestep_features_f32(_p(t), _p(out), out.shape[0], out.shape[1], self.s))

    def linear_rows(self, x, w, bias, out, act_in=0, act_out=0):
        """fp32 rows x (fp16 or fp32) weights; act 0 none, 1 SiLU, 2 GELU."""
        m, k = x.shape
        n = w.shape[0]
        assert w.shape[1] == k and out.shape == (m, n) and x.dtype == torch.float32
        assert w.dtype in (torch.float16, torch.float32)
        L.check(self.lib.dadd_linear_rows_f32(_p(x), _p(w), _p(bias), _p(out), m, k, n, act_in,
                                              act_out, int(w.dtype == torch.float32), self.s))

    def attention(self, q, k, v, out, heads):
        """softmax(q k^T / sqrt(d)) v with separate query / key-value lengths: q [B,Nq,*], k, v [B,Nk,*] (views
        into wider rows are fine: the row strides travel), out [B,Nq,C]."""
        b, nq, c = out.shape
        nk = k.shape[1]
        assert q.shape[:2] == (b, nq) and v.shape[:2] == (b, nk) and k.stride(1) == v.stride(1)
        L.check(self.lib.dadd_attn_f16(_p(q), _p(k), _p(v), _p(out), b, nq, nk, heads, c // heads, q.stride(1),
                                       k.stride(1), out.stride(1), self.s))

    def clip_patch_rows(self, pixels, out, patch):
        b, _, h, w = pixels.shape
        assert pixels.dtype == torch.float32 and out.dtype == torch.float16 and out.shape[0] == b
        L.check(self.lib.dadd_clip_patch_rows_f16(_p(pixels), _p(out), b, h, w, patch, out.shape[-1], self.s))

    def aoe_interp(self, labels, base, deltas, out):
        L.check(self.lib.dadd_aoe_interp_f32(_p(labels), _p(base), _p(deltas), _p(out), labels.shape[0],
                                             out.shape[1], deltas.shape[0] + 1, self.s))

    def purifier_tail(self, img, dis, gate, gamma, beta, out, eps=1e-5):
        c = img.shape[-1]
        L.check(self.lib.dadd_purifier_tail_f16(_p(img), _p(dis), _p(gate), _p(gamma), _p(beta), _p(out),
                                                img.numel() // c, c, float(eps), self.s))

    def begin_step(self, table, cur_rows, coef, cur_coef, step):
        assert step.numel() == 2 and step.dtype == torch.int32
        L.check(self.lib.dadd_begin_step(_p(table), _p(cur_rows), cur_rows.shape[0], table.shape[1],
                                         _p(coef), _p(cur_coef), _p(step), self.s))

    def ddim_update(self, x, eps_c, eps_u, guidance, coef, guidance_dev=None):
        L.check(self.lib.dadd_ddim_update_f32(_p(x), _p(eps_c), _p(eps_u), float(guidance), _p(guidance_dev),
                                              _p(coef), x.numel(), self.s))

    def prefetch(self, t: torch.Tensor):
        """Read ``t`` on the library's side stream (a parallel branch inside a graph capture): its lines are in the
        Infinity Cache when a later kernel streams them.  ``prefetch_join`` before the capture ends."""
        if self._prof_on:            # per-launch timing runs one kernel at a time: no side branch
            return
        L.check(self.lib.dadd_prefetch(_p(t), t.numel() * t.element_size(), self.s))

    def prefetch_join(self):
        L.check(self.lib.dadd_prefetch_join(self.s))

    # ------------------------------------------------------------------ graphs / profiling
    def graph_begin(self):
        L.check(self.lib.dadd_graph_begin(self.s))
        self._capturing = True

    def graph_end(self):
        self._capturing = False
        g = C.c_void_p()
        L.check(self.lib.dadd_graph_end(self.s, C.byref(g)))
        return g

    def graph_launch(self, g):
        L.check(self.lib.dadd_graph_launch(g, self.s))

    def graph_destroy(self, g):
        L.check(self.lib.dadd_graph_destroy(g))

    def prof_begin(self):
        """Start recording every kernel launch of the library (eager launches only)."""
        L.check(self.lib.dadd_prof_begin())
        self._prof_on = True

    def prof_end(self):
        """-> list of (kernel name, microseconds, algorithmic flop, algorithmic bytes) in issue order; the time
        is the dispatch's own begin/end timestamp pair (what rocprofv3's kernel trace reports)."""
        n = C.c_int(0)
        self._prof_on = False
        L.check(self.lib.dadd_prof_end(C.byref(n)))
        out, name, vals = [], C.c_char_p(), (C.c_double * 3)()
        for i in range(n.value):
            L.check(self.lib.dadd_prof_record(i, C.byref(name), vals))
            out.append((name.value.decode(), vals[0] * 1e3, vals[1], vals[2]))
        return out
