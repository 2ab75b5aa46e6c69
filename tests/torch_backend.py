"""TEST-ONLY operator backend: the op contract of ``progressive_stable_diffusion_amd.backend``
restated with plain torch on the CPU (fp16 storage, fp32 arithmetic).

Purpose: (1) let the CPU suite check the *wiring* of the engine's plans (which buffer feeds which
op, weight packing, GEGLU interleave, skip order, split-K bookkeeping) against the oracle without
a GPU; (2) serve as the per-op torch reference the ``-m gpu`` kernel tests compare the HIP kernels
with.  The product never imports this module and has no CPU path.
"""
from __future__ import annotations

import contextlib
import math

import torch
import torch.nn.functional as F

EPI_BIAS, EPI_ROWVEC, EPI_RESIDUAL, EPI_GEGLU = 1, 2, 4, 8
EPI_GNAPPLY_SILU = 65536


def geglu_deinterleave_index(n: int) -> torch.Tensor:
    """Inverse of engine.geglu_interleave: physical row -> logical row."""
    n2 = n // 2
    idx = torch.arange(n)
    tile, r = idx // 128, idx % 128
    wn, rr = r // 64, r % 64
    col = tile * 64 + wn * 32 + (rr % 32)
    return torch.where(rr < 32, col, n2 + col)


class TorchRefBackend:
    name = "torch-ref"

    def __init__(self, device="cpu"):
        self.device = torch.device(device)
        self.launches = 0

    # plumbing -------------------------------------------------------------------------------
    def ctx(self):
        return contextlib.nullcontext()

    def empty(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    zeros = empty

    def to_device(self, t, dtype=None):
        return t.to(device=self.device, dtype=dtype or t.dtype).contiguous()

    def copy_(self, dst, src):
        dst.copy_(src.reshape(dst.shape))

    def zero_(self, t):
        t.zero_()

    def clone(self, t):
        return t.detach().clone()

    def synchronize(self):
        pass

    def wait_current(self):
        pass

    def release_to_current(self):
        pass

    # ops ------------------------------------------------------------------------------------
    def pack_latents(self, x, out, scale=1.0, mat=None, vec=None):
        v = x.float() * scale
        if mat is not None:
            v = torch.einsum("oc,bchw->bohw", mat.float(), v)
            if vec is not None:
                v = v + vec.float()[None, :, None, None]
        out.zero_()
        out[..., : x.shape[1]] = v.permute(0, 2, 3, 1).to(out.dtype)

    def conv_cin8(self, x, w, bias, out):
        co = w.shape[0]
        wt = w.float().reshape(co, 3, 3, 8).permute(0, 3, 1, 2)
        y = F.conv2d(x.float().permute(0, 3, 1, 2), wt, None if bias is None else bias.float(), padding=1)
        out.copy_(y.permute(0, 2, 3, 1).to(out.dtype))

    def conv_in_nchw(self, x, w, bias, out, gn_ws=None, gn_nchunk=0):
        x8 = torch.zeros(x.shape[0], x.shape[2], x.shape[3], 8, dtype=torch.float16)
        self.pack_latents(x, x8)
        self.conv_cin8(x8, w, bias, out)
        if gn_ws is not None:           # chunk partials of the rounded output, [B][nchunk][32][2]
            b, h, wd, c = out.shape
            o = out.float().reshape(b, gn_nchunk, (h * wd) // gn_nchunk, 32, c // 32)
            st = torch.stack([o.sum(dim=(2, 4)), (o * o).sum(dim=(2, 4))], dim=-1)
            gn_ws[:b * gn_nchunk * 64].copy_(st.reshape(-1))

    def conv_out_ddim(self, x, w, bias, latents, coef):
        eps = torch.empty_like(latents)
        self.conv_cout4(x, w, bias, eps, 0)
        self.ddim_update(latents, eps, None, 1.0, coef)

    def conv_cout4(self, x, w, bias, out, mode=0):
        co, _, c = w.shape
        wt = w.float().reshape(co, 3, 3, c).permute(0, 3, 1, 2)
        y = F.conv2d(x.float().permute(0, 3, 1, 2), wt, None if bias is None else bias.float(), padding=1)
        if mode == 1:
            y = ((y.clamp(-1, 1) + 1.0) / 2.0).clamp(0, 1)
        elif mode == 2:
            y = y.clamp(-30.0, 20.0)
        out.copy_(y)

    def q_sample(self, x0, noise, t, alphas_cumprod, out):
        a = alphas_cumprod[t].view(-1, 1, 1, 1)
        out.copy_(torch.sqrt(a) * x0 + torch.sqrt(1.0 - a) * noise)

    def mse_rows(self, pred, target, out):
        out.copy_(((pred - target) ** 2).mean(dim=tuple(range(1, pred.dim()))))

    def frames_to_u8(self, frames, out):
        out.copy_(frames.permute(0, 2, 3, 1).mul(255).to(torch.uint8))

    def gaussian_sample(self, mean, logvar, noise, out, scale=1.0):
        out.copy_((mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise) * scale)

    def igemm(self, x, w, out, *, x2=None, bias=None, rowvec=None, residual=None, taps=1, stride=1,
              ups=0, pad=0, flags=0, splitk=1, partial=None, tile_n=0, tile_m=0, counters=None, ln_c1=None,
              ln_eps=1e-5, gn_ws=None, gn_nchunk=0, ln_stats_out=None, ln_stats_in=None, gn_in=None, gn_apply=None):
        self.launches += 1
        if gn_in is not None:           # PRE_GN: GroupNorm (+ SiLU) of x from its chunk partials, rounded to fp16 like gn_apply
            ws_in, nch_in, gam, bet, eps_in = gn_in[:5]
            if len(gn_in) > 5:          # skip-concat: the partials of both sources, each in its own 32-group layout
                ws2, nch2 = gn_in[5], gn_in[6]
                bsz, c1_, c2_ = x.shape[0], x.shape[-1], x2.shape[-1]
                cgc = (c1_ + c2_) // 32
                s1 = ws_in[: bsz * nch_in * 64].reshape(bsz, nch_in, 32, 2).double().sum(dim=1)
                s2 = ws2[: bsz * nch2 * 64].reshape(bsz, nch2, 32, 2).double().sum(dim=1)
                r1, r2 = cgc // (c1_ // 32), cgc // (c2_ // 32)
                cat = torch.cat([s1.reshape(bsz, 32 // r1, r1, 2).sum(dim=2), s2.reshape(bsz, 32 // r2, r2, 2).sum(dim=2)], dim=1)
                wsc = cat.float().reshape(bsz, 1, 32, 2).contiguous().reshape(-1)
                xc = torch.cat([x, x2], dim=-1)
                xn = torch.empty_like(xc)
                self.groupnorm(xc, None, gam, bet, xn, wsc, 32, eps_in, 1 if flags & 16384 else 0, ws_chunks=1)
                x, x2 = xn[..., :c1_].contiguous(), xn[..., c1_:].contiguous()
            else:
                xn = torch.empty_like(x)
                self.groupnorm(x, None, gam, bet, xn, ws_in, 32, eps_in, 1 if flags & 16384 else 0, ws_chunks=nch_in)
                x = xn
        xin = x.float() if x2 is None else torch.cat([x.float(), x2.float()], dim=-1)
        b, hi, wi, cin = xin.shape
        n = w.shape[0]
        assert cin % 64 == 0 and x.shape[-1] % 64 == 0 and n % 8 == 0
        assert w.shape[1] == taps * cin
        ho, wo = out.shape[1], out.shape[2]
        xn = xin.permute(0, 3, 1, 2)
        if ups:
            xn = F.interpolate(xn, scale_factor=2.0, mode="nearest")
        k = 3 if taps == 9 else 1
        wt = w.float().reshape(n, k, k, cin).permute(0, 3, 1, 2)
        if pad == 0 and k == 3:       # asymmetric (0,1,0,1) padding of the VAE-encoder downsample
            xn = F.pad(xn, (0, 2, 0, 2))
            y = F.conv2d(xn, wt, None, stride=stride)[:, :, :ho, :wo]
        else:
            y = F.conv2d(xn, wt, None, stride=stride, padding=pad)
        assert y.shape[2] == ho and y.shape[3] == wo, (y.shape, out.shape)
        y = y.permute(0, 2, 3, 1)
        if flags & 128:                 # EPI_LNFOLD: rstd * (x (gamma o W)^T - mu * c1), statistics of the fp16 rows
            if ln_stats_in is not None:     # row partials [P][M][2] written by the producer of x (EPI_LNSTAT)
                st = ln_stats_in.sum(dim=0).reshape(b, hi, wi, 2) / cin
                mu, var = st[..., 0:1], st[..., 1:2] - st[..., 0:1] ** 2
            else:
                mu = xin.mean(dim=-1, keepdim=True)
                var = (xin * xin).mean(dim=-1, keepdim=True) - mu * mu
            y = torch.rsqrt(var.clamp_min(0.0) + ln_eps) * (y - mu * ln_c1.float())
        act = flags & (256 | 512 | 1024)
        gn_apply_silu = bool(flags & EPI_GNAPPLY_SILU)
        flags &= 15                     # tuning bits (16, 32) do not change the math
        if flags & EPI_BIAS:
            y = y + bias.float()
        if flags & EPI_ROWVEC:
            y = y + rowvec.float()[:, None, None, :]
        if act & 256:
            y = y * torch.sigmoid(1.702 * y)
        elif act & 512:
            y = F.gelu(y)
        elif act & 1024:
            y = torch.sigmoid(y)
        if flags & EPI_GEGLU:
            y = y[..., geglu_deinterleave_index(n).argsort()]   # undo the physical row order
            hid, gate = y.chunk(2, dim=-1)
            y = hid * F.gelu(gate)
        if flags & EPI_RESIDUAL:
            y = y + residual.float()
        out.copy_(y.to(out.dtype))
        if ln_stats_out is not None:    # EPI_LNSTAT: row partials of the rounded output over blocks of N / P columns
            o = out.float().reshape(b * ho * wo, ln_stats_out.shape[0], -1)
            ln_stats_out.copy_(torch.stack([o.sum(dim=-1), (o * o).sum(dim=-1)], dim=-1).permute(1, 0, 2))
        if gn_ws is not None:           # EPI_GNSTAT: chunk partials of the rounded output, [B][nchunk][32][2]
            o = out.float().reshape(b, gn_nchunk, -1, 32, n // 32)
            part = torch.stack([o.sum(dim=(2, 4)), (o * o).sum(dim=(2, 4))], dim=-1)
            gn_ws[: part.numel()].copy_(part.reshape(-1))
        if gn_apply is not None:        # EPI_GNAPPLY: GroupNorm (+ SiLU) of the rounded output, written beside it
            g_out, gam, bet, eps_o = gn_apply
            self.groupnorm(out, None, gam, bet, g_out, None, 32, eps_o, gn_apply_silu)

    def prefetch(self, t):
        pass

    def prefetch_join(self):
        pass

    def groupnorm(self, x1, x2, gamma, beta, out, ws, groups, eps, silu, ws_chunks=0):
        x = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], dim=-1)
        if ws_chunks:                   # statistics come from the producer's partials, not from x
            b, c = x.shape[0], x.shape[-1]
            part = ws[: b * ws_chunks * groups * 2].reshape(b, ws_chunks, groups, 2).double().sum(dim=1)
            cnt = x.shape[1] * x.shape[2] * (c // groups)
            mean = part[..., 0] / cnt
            rstd = torch.rsqrt((part[..., 1] / cnt - mean * mean).clamp_min(0) + eps)
            xg = x.reshape(b, -1, groups, c // groups)
            y = ((xg - mean.float()[:, None, :, None]) * rstd.float()[:, None, :, None]).reshape(x.shape)
            y = (y * gamma.float() + beta.float()).permute(0, 3, 1, 2)
        else:
            y = F.group_norm(x.permute(0, 3, 1, 2), groups, gamma.float(), beta.float(), eps)
        if silu:
            y = F.silu(y)
        out.copy_(y.permute(0, 2, 3, 1).to(out.dtype))

    def layernorm(self, x, gamma, beta, out, eps=1e-5):
        out.copy_(F.layer_norm(x.float(), (x.shape[-1],), gamma.float(), beta.float(), eps).to(out.dtype))

    def self_attn(self, qkv, out, heads):
        b, n, c3 = qkv.shape
        c = c3 // 3
        d = c // heads
        q, k, v = (t.float().view(b, n, heads, d).transpose(1, 2) for t in qkv.split(c, dim=-1))
        p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1)
        out.copy_((p @ v).transpose(1, 2).reshape(b, n, c).to(out.dtype))

    def attn2_fused(self, x, mcat, vw, bias, residual, out, ln_stats_out=None, ln_stats_in=None, ln_c1=None, ln_d=None,
                    ln_eps=1e-5):
        b, hw, c = x.shape
        s = torch.einsum("bmc,bkc->bmk", x.float(), mcat.float())            # log2(e)/sqrt(d) folded in
        if ln_stats_in is not None:     # norm2 folded in: S = rstd (x mcat^T - mu c1) + d
            st = ln_stats_in.sum(dim=0).reshape(b, hw, 2) / c
            mu, var = st[..., 0:1], (st[..., 1:2] - st[..., 0:1] ** 2).clamp_min(0.0)
            s = torch.rsqrt(var + ln_eps) * (s - mu * ln_c1.float()[:, None, :]) + ln_d.float()[:, None, :]
        pr = torch.softmax(s.view(b, hw, 24, 16) * math.log(2.0), dim=-1).view(b, hw, 384)
        pr = pr.to(torch.float16).float()                                     # P is stored in fp16
        y = torch.einsum("bmk,bnk->bmn", pr, vw.float())
        if bias is not None:
            y = y + bias.float()
        out.copy_((y + residual.float()).to(out.dtype))
        if ln_stats_out is not None:
            o = out.float().reshape(b * hw, ln_stats_out.shape[0], -1)
            ln_stats_out.copy_(torch.stack([o.sum(dim=-1), (o * o).sum(dim=-1)], dim=-1).permute(1, 0, 2))

    @staticmethod
    def unpack_ffn_stream(stream, b1):
        """Inverse of ``engine.pack_ffn_stream``: (w1 [2560,320], b1 [2560], w2 [320,1280], wp [320,320]) from the piece
        stream — decoded piece by piece with the kernel's own addressing (image position p of row r holds chunk
        p ^ ((r >> 1) & 7)), so the CPU suite checks the packing the device consumes."""
        c, hid = 320, 1280

        def unimage(t, rows):
            t = t.reshape(rows, 8, 8)
            sw = (torch.arange(rows) >> 1) & 7
            idx = torch.arange(8)[None, :] ^ sw[:, None]                 # chunk q sits at position q ^ sw
            return t.gather(1, idx[:, :, None].expand(rows, 8, 8)).reshape(rows, 64)
        w1, w2, wp = torch.zeros(2 * hid, c), torch.zeros(c, hid), torch.zeros(c, c)
        bias = torch.zeros(2 * hid)
        st, off, boff = stream.float().cpu(), 0, 0
        e = torch.arange(16)
        for ch in range(hid // 64):
            rows = []
            for wn in range(2):
                for u in range(2):
                    h = ch * 64 + wn * 32 + u * 16 + e
                    rows += [h, hid + h]
            rows = torch.cat(rows)
            bias[rows] = b1.float().cpu()[boff:boff + 128]
            boff += 128
            for kt in range(c // 64):
                w1[rows, kt * 64:(kt + 1) * 64] = unimage(st[off:off + 128 * 64], 128)
                off += 128 * 64
            for nh in range(2):
                w2[nh * 160:(nh + 1) * 160, ch * 64:(ch + 1) * 64] = unimage(st[off:off + 160 * 64], 160)
                off += 160 * 64
        for nh in range(2):
            for kt in range(c // 64):
                wp[nh * 160:(nh + 1) * 160, kt * 64:(kt + 1) * 64] = unimage(st[off:off + 160 * 64], 160)
                off += 160 * 64
        assert off == st.numel()
        return w1, bias, w2, wp

    def ffn_block(self, x, stream, ln_g, ln_b, b1, b2, bp, xres, out, gn_ws=None, gn_nchunk=0, ln_eps=1e-5):
        """csrc/ffn_block.hip in torch: the four rounding points of the unfused launches (normalised rows, GEGLU output,
        h4, result), everything else fp32."""
        import torch.nn.functional as Fn
        b, hw, c = x.shape
        w1, b1u, w2, wp = self.unpack_ffn_stream(stream, b1)
        xn = Fn.layer_norm(x.float(), (c,), ln_g.float(), ln_b.float(), ln_eps).to(torch.float16).float()
        h = xn @ w1.T + b1u
        gg = (h[..., :1280] * Fn.gelu(h[..., 1280:])).to(torch.float16).float()
        h4 = (gg @ w2.T + b2.float() + x.float()).to(torch.float16).float()
        out.copy_((h4 @ wp.T + bp.float() + xres.float()).to(out.dtype))
        if gn_ws is not None:
            o = out.float().reshape(b, gn_nchunk, hw // gn_nchunk, 32, c // 32)
            st = torch.stack([o.sum(dim=(2, 4)), (o * o).sum(dim=(2, 4))], dim=-1)      # [b][chunk][32][2]
            gn_ws[:b * gn_nchunk * 64].copy_(st.reshape(-1))

    @staticmethod
    def unpack_head_stream(stream):
        """Inverse of ``engine.pack_head_stream``: (proj_in [320,320], to_q|to_k|to_v [960,320])."""
        c = 320
        st, off = stream.float().cpu(), 0
        sw = (torch.arange(160) >> 1) & 7
        idx = (torch.arange(8)[None, :] ^ sw[:, None])[:, :, None].expand(160, 8, 8)
        out = []
        for rows in (c, 3 * c):
            w = torch.zeros(rows, c)
            for nh in range(rows // 160):
                for kt in range(c // 64):
                    w[nh * 160:(nh + 1) * 160, kt * 64:(kt + 1) * 64] = st[off:off + 160 * 64].reshape(160, 8, 8).gather(1, idx).reshape(160, 64)
                    off += 160 * 64
            out.append(w)
        assert off == st.numel()
        return out

    def tf_head(self, x, stream, gn_ws, gn_nchunk, gn_g, gn_b, bp, ln_g, ln_b, hs, qkv, gn_eps=1e-6, ln_eps=1e-5):
        """csrc/tf_head.hip in torch: GroupNorm from the producer's chunk partials (x * scale + shift, fp16), proj_in + bias
        (fp16 hs), LayerNorm 1 (fp16), q|k|v (fp16)."""
        import torch.nn.functional as Fn
        b, hw, c = x.shape
        wp, wqkv = self.unpack_head_stream(stream)
        st = gn_ws[:b * gn_nchunk * 64].double().reshape(b, gn_nchunk, 32, 2).sum(dim=1)
        n = hw * (c // 32)
        mu = st[..., 0] / n
        rstd = 1.0 / torch.sqrt((st[..., 1] / n - mu * mu).clamp_min(0.0) + gn_eps)
        sc = rstd.float().repeat_interleave(c // 32, dim=1) * gn_g.float()                 # [b][c]
        sh = gn_b.float() - mu.float().repeat_interleave(c // 32, dim=1) * sc
        g = (x.float() * sc[:, None, :] + sh[:, None, :]).to(torch.float16).float()
        h = (g @ wp.T + bp.float()).to(torch.float16)
        hs.copy_(h)
        ln = Fn.layer_norm(h.float(), (c,), ln_g.float(), ln_b.float(), ln_eps).to(torch.float16).float()
        qkv.copy_((ln @ wqkv.T).to(qkv.dtype))

    def tri_xattn(self, q, kv, out, gates, lam, mode, heads, lam_dev=None):
        if lam_dev is not None:
            lam = float(lam_dev.reshape(-1)[0])
        b, n, c = q.shape
        d = c // heads
        qh = q.float().view(b, n, heads, d).transpose(1, 2)

        def path(tok0, ntok, kcol, vcol):
            k = kv[:, tok0:tok0 + ntok, kcol:kcol + c].float().view(b, ntok, heads, d).transpose(1, 2)
            v = kv[:, tok0:tok0 + ntok, vcol:vcol + c].float().view(b, ntok, heads, d).transpose(1, 2)
            return torch.softmax(qh @ k.transpose(-1, -2) / math.sqrt(d), dim=-1) @ v

        if mode == 0:
            z = gates[0].float() * path(16, 16, 0, c) + gates[1].float() * path(0, 16, 2 * c, 3 * c)
            if lam != 0.0:
                z = z + lam * path(32, 16, 2 * c, 3 * c)
        else:
            z = path(0, 32, 0, c)
        out.copy_(z.transpose(1, 2).reshape(b, n, c).to(out.dtype))

    def timestep_features(self, t, out):
        half = out.shape[1] // 2
        freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
        ang = t.float()[:, None] * freqs[None, :]
        out.copy_(torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1))

    def linear_rows(self, x, w, bias, out, act_in=0, act_out=0):
        act = {0: lambda t: t, 1: F.silu, 2: F.gelu}
        y = F.linear(act[act_in](x.float()), w.float(), None if bias is None else bias.float())
        out.copy_(act[act_out](y))

    def attention(self, q, k, v, out, heads):
        b, nq, c = out.shape
        d = c // heads
        qh, kh, vh = (t[..., :c].float().reshape(b, t.shape[1], heads, d).transpose(1, 2) for t in (q, k, v))
        p = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(d), dim=-1)
        out.copy_((p @ vh).transpose(1, 2).reshape(b, nq, c).to(out.dtype))

    def clip_patch_rows(self, pixels, out, patch):
        b, _, h, w = pixels.shape
        cols = F.unfold(pixels.float(), kernel_size=patch, stride=patch).transpose(1, 2)     # [B, T-1, 3*P*P]
        out.zero_()
        out[:, 1:, : cols.shape[-1]] = cols.to(out.dtype)

    def aoe_interp(self, labels, base, deltas, out):
        steps = torch.cumsum(deltas.float(), dim=0)
        tab = base.float() + torch.cat([torch.zeros_like(steps[:1]), steps], dim=0)
        top = tab.shape[0] - 1
        y = labels.float().clamp(0.0, float(top))
        lo = y.floor()
        frac = (y - lo)[:, None]
        lo_i = lo.long()
        hi_i = (lo_i + 1).clamp(max=top)
        out.copy_(tab[lo_i] * (1.0 - frac) + tab[hi_i] * frac)

    def purifier_tail(self, img, dis, gate, gamma, beta, out, eps=1e-5):
        v = img.float() - gate.float() * dis.float()
        out.copy_(F.layer_norm(v, (v.shape[-1],), gamma.float(), beta.float(), eps).reshape(out.shape))

    def begin_step(self, table, cur_rows, coef, cur_coef, step):
        r = int(step[0].item())
        cur_rows.copy_(table[r][None, :].expand_as(cur_rows))
        cur_coef.copy_(coef[r])
        step[0] += 1

    def ddim_update(self, x, eps_c, eps_u, guidance, coef, guidance_dev=None):
        if guidance_dev is not None:
            guidance = float(guidance_dev.reshape(-1)[0])
        e = eps_c if eps_u is None else eps_u + guidance * (eps_c - eps_u)
        x0 = ((x - coef[1] * e) / coef[0]).clamp(-4.0, 4.0)
        x.copy_(x0 if coef[2] < 0 else coef[2] * x0 + coef[3] * e)

    # graphs: the reference backend just replays eagerly ---------------------------------------
    def graph_begin(self):
        self._cap = []

    def graph_end(self):
        raise NotImplementedError("the reference backend has no graphs; run with use_graph=False")
