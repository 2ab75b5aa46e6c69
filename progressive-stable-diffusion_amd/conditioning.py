"""Conditioning front-end of the sampler on the HIP kernels: AOE, CLIP vision
tower, image projection / Perceiver resampler, FeaturePurifier.

Runs ONCE per batch.  Every matrix product goes through the implicit-GEMM kernel (``dadd_conv_igemm_f16``: fp16
operands, fp32 accumulate; LayerNorms folded into the consuming linear, quick-GELU / GELU / sigmoid and residual adds
in the epilogue), every attention through the flash kernel (``dadd_attn_f16``: d = 64 for CLIP, d = 96 for the
``nn.MultiheadAttention(768, 8)`` blocks), the AOE projector through the fp32-weight rows kernel (its delta tokens
are differences of two outputs).  torch is used for memory, dtype conversion at the class boundaries and views.
Call signatures mirror the reference classes so that ``_prepare_conditioning`` reads the same on both sides:
  * ``AdditiveOrdinalEmbedder``  — src/models/ordinal_embedder.py:43-309
  * ``FeaturePurifier``          — src/models/feature_purifier.py:29-95
  * ``ImageProjection[Plus]``    — src/models/image_encoder.py:91-228
  * ``ImageEncoder``             — src/models/image_encoder.py:17-88 (transformers CLIPVisionModelWithProjection:
                                   the tower is restated on the kernels; tests compare with transformers itself)
Class boundaries are fp32 torch tensors on the device (what the reference modules exchange); inside, activations
are fp16 rows.  There is no CPU path: the classes need a backend (the product passes ``HipBackend``).
"""
from __future__ import annotations

from typing import Dict, Iterator, Optional

import torch

from . import lib as L
from .engine import fold_layernorm, plan_tiling

F16, F32 = torch.float16, torch.float32


class _Params:
    """Slice of a flat state dict under ``prefix`` (fp32 masters on the device, for ``parameters()`` and packing)
    plus the GEMM helpers shared by the modules below."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, device, be):
        if be is None:
            raise RuntimeError("the conditioning modules run on the HIP backend: pass be=HipBackend(...)")
        n = len(prefix) + 1
        self.be = be
        self.device = torch.device(device)
        self.p = {k[n:]: be.to_device(v.detach().float()) for k, v in sd.items() if k.startswith(prefix + ".")}
        if not self.p:
            raise KeyError(f"no parameters under '{prefix}.' in the state dict")
        self._packed: Dict = {}

    def parameters(self) -> Iterator[torch.Tensor]:
        return iter(self.p.values())

    def to(self, *a, **k):
        return self

    # ---- packing (once per module) -----------------------------------------------------------------------------
    def pk(self, key, make):
        t = self._packed.get(key)
        if t is None:
            t = self._packed[key] = make()
        return t

    def w16(self, key):
        return self.pk(("w", key), lambda: self.p[key + ".weight"].to(F16).contiguous())

    def b32(self, key):
        return self.p.get(key + ".bias")

    def folded(self, name, wkeys, bkeys, norm):
        """(w16, c1, bias) of LayerNorm ``norm`` folded into the row-concatenated linears ``wkeys``."""
        def make():
            w = torch.cat([self.p[k] for k in wkeys]).cpu()
            b = torch.cat([self.p[k] for k in bkeys]).cpu() if bkeys else None
            w16, c1, bias = fold_layernorm(w, b, self.p[norm + ".weight"].cpu(), self.p[norm + ".bias"].cpu())
            return tuple(self.be.to_device(t.contiguous()) for t in (w16, c1, bias))
        return self.pk(("fold", name), make)

    # ---- ops ---------------------------------------------------------------------------------------------------
    def rows16(self, x: torch.Tensor) -> torch.Tensor:
        """fp32 (..., C) -> fp16 rows [1, M, 1, C] on the backend stream."""
        c = x.shape[-1]
        out = self.be.empty((1, x.numel() // c, 1, c), F16)
        self.be.copy_(out, x)
        return out

    def gemm(self, x, w, bias=None, *, residual=None, x2=None, flags=0, ln_c1=None, eps=1e-5):
        """rows [1,M,1,K] (x2: a second source concatenated along K) times w[N,K]^T -> rows [1,M,1,N]."""
        m, n = x.shape[1], w.shape[0]
        out = self.be.empty((1, m, 1, n), F16)
        tile_m, tile_n, sk, tune = plan_tiling(m, n, w.shape[1], 1, False, residual is not None)
        if ln_c1 is not None or flags & (L.EPI_QUICKGELU | L.EPI_GELU | L.EPI_SIGMOID):
            sk = 1
        partial = self.be.empty((sk * m * n,), F32) if sk > 1 else None
        f = flags | tune | (L.EPI_BIAS if bias is not None else 0) | (L.EPI_RESIDUAL if residual is not None else 0) \
            | (L.EPI_LNFOLD if ln_c1 is not None else 0)
        kw = dict(ln_c1=ln_c1, ln_eps=eps) if ln_c1 is not None else {}
        self.be.igemm(x, w, out, x2=x2, bias=bias, residual=residual, flags=f, splitk=sk, partial=partial,
                      tile_m=tile_m, tile_n=tile_n, **kw)
        return out

    def layernorm(self, x, key, eps=1e-5):
        out = self.be.empty(tuple(x.shape), F16)
        self.be.layernorm(x, self.p[key + ".weight"], self.p[key + ".bias"], out, eps)
        return out

    def mha(self, key, q_rows, kv_rows, bq, nq, nk, heads, q_norm=None, kv_norm=None, residual=None):
        """``nn.MultiheadAttention`` (packed in_proj, batch_first) over fp16 rows: q_rows [1,bq*nq,1,E], kv_rows
        [1,bq*nk,1,E]; ``q_norm`` / ``kv_norm`` name LayerNorms folded into the q / kv projections; ``residual`` is
        added in the out_proj epilogue."""
        e = q_rows.shape[-1]
        wk, bk = key + ".in_proj_weight", key + ".in_proj_bias"

        def split(lo, hi, norm, name):
            if norm is not None:
                def make():
                    w16, c1, bias = fold_layernorm(self.p[wk][lo:hi].cpu(), self.p[bk][lo:hi].cpu(),
                                                   self.p[norm + ".weight"].cpu(), self.p[norm + ".bias"].cpu())
                    return tuple(self.be.to_device(t.contiguous()) for t in (w16, c1, bias))
                return self.pk(("mha", key, name), make)
            return self.pk(("mha", key, name), lambda: (self.p[wk][lo:hi].to(F16).contiguous(), None,
                                                        self.p[bk][lo:hi].contiguous()))
        wq, cq, bq_ = split(0, e, q_norm, "q")
        wkv, ckv, bkv = split(e, 3 * e, kv_norm, "kv")
        q = self.gemm(q_rows, wq, bq_, ln_c1=cq)
        kv = self.gemm(kv_rows, wkv, bkv, ln_c1=ckv)
        att = self.be.empty((bq, nq, e), F16)
        kvv = kv.view(bq, nk, 2 * e)
        self.be.attention(q.view(bq, nq, e), kvv[:, :, :e], kvv[:, :, e:], att, heads)
        return self.gemm(att.view(1, bq * nq, 1, e), self.w16(key + ".out_proj"), self.b32(key + ".out_proj"),
                         residual=residual)

    def out32(self, rows, shape):
        out = self.be.empty(shape, F32)
        self.be.copy_(out, rows)
        return out


def _io(fn):
    """Class-boundary stream discipline: inputs produced on torch's current stream, outputs consumed there."""
    def wrapped(self, *a, **k):
        self.be.wait_current()
        try:
            return fn(self, *a, **k)
        finally:
            self.be.release_to_current()
    wrapped.__doc__ = fn.__doc__
    return wrapped


class AdditiveOrdinalEmbedder(_Params):
    """fp32 throughout: class interpolation kernel + the projector MLP on the fp32-weight rows kernel."""

    def __init__(self, sd, device, num_classes=4, embedding_dim=768, num_tokens=16,
                 prefix="ordinal_embedder", be=None):
        super().__init__(sd, prefix, device, be)
        if num_classes < 2:
            raise ValueError("num_classes must be ≥ 2 for ordinal modeling.")
        self.num_classes, self.embedding_dim, self.num_tokens = num_classes, embedding_dim, num_tokens

    def _interp(self, labels):
        lab = self.be.to_device(labels.detach().reshape(-1).float())
        out = self.be.empty((lab.shape[0], self.embedding_dim), F32)
        self.be.aoe_interp(lab, self.p["base"], self.p["deltas"], out)
        return out

    def _tokens(self, emb):
        p = self.p
        h = self.be.empty((emb.shape[0], p["projector.0.weight"].shape[0]), F32)
        self.be.linear_rows(emb, p["projector.0.weight"], p["projector.0.bias"], h, 0, 2)        # Linear -> GELU
        out = self.be.empty((emb.shape[0], p["projector.2.weight"].shape[0]), F32)
        self.be.linear_rows(h, p["projector.2.weight"], p["projector.2.bias"], out, 0, 0)
        return out.view(-1, self.num_tokens, self.embedding_dim)

    @_io
    def __call__(self, labels, is_training=False, unconditional=False, noise_std=0.005):
        if unconditional:
            return self.p["null_embedding"].expand(labels.shape[0] if labels.dim() else 1, -1)
        scalar = labels.dim() == 0
        emb = self._interp(labels[None] if scalar else labels)
        if is_training and noise_std > 0:
            with self.be.ctx():
                emb = emb + torch.randn_like(emb) * noise_std
        out = self._tokens(emb)
        return out[0] if scalar else out

    forward = __call__

    def get_negative_embedding(self, labels, is_training=False, noise_std=0.005):
        scalar = labels.dim() == 0
        lab = labels[None] if scalar else labels
        return self(torch.clamp(1.0 - lab, 0.0, 1.0), is_training=is_training, noise_std=noise_std)

    @_io
    def get_ordinal_delta_embedding(self, source_labels, target_labels):
        scalar = source_labels.dim() == 0
        if scalar:
            source_labels, target_labels = source_labels[None], target_labels[None]
        t, s = self._tokens(self._interp(target_labels)), self._tokens(self._interp(source_labels))
        with self.be.ctx():
            d = t - s                   # exactly 0 when source == target (ordinal_embedder.py:254-255)
        return d[0] if scalar else d

    def get_disease_delta_embedding(self, source_labels):
        return self.get_ordinal_delta_embedding(source_labels, torch.zeros_like(source_labels))


class FeaturePurifier(_Params):
    def __init__(self, sd, device, num_heads=8, prefix="feature_purifier", be=None):
        super().__init__(sd, prefix, device, be)
        self.num_heads = num_heads

    @_io
    def __call__(self, image_embeds, source_aoe):
        b, n_img, e = image_embeds.shape
        n_aoe = source_aoe.shape[1]
        img = self.rows16(image_embeds)
        aoe = self.rows16(source_aoe)
        dis = self.mha("cross_attn", img, aoe, b, n_img, n_aoe, self.num_heads, q_norm="norm_img", kv_norm="norm_aoe")
        img_n = self.layernorm(img, "norm_img")
        h = self.gemm(dis, self.w16("gate.0"), self.b32("gate.0"), x2=img_n, flags=L.EPI_GELU)   # cat([dis, img_n])
        gate = self.gemm(h, self.w16("gate.2"), self.b32("gate.2"), flags=L.EPI_SIGMOID)
        out = self.be.empty((b, n_img, e), F32)
        self.be.purifier_tail(img, dis, gate, self.p["norm_out.weight"], self.p["norm_out.bias"], out)
        return out

    forward = __call__


class ImageProjectionPlus(_Params):
    def __init__(self, sd, device, num_tokens=16, num_heads=8, prefix="image_projection", be=None):
        super().__init__(sd, prefix, device, be)
        self.num_tokens, self.num_heads = num_tokens, num_heads
        self.depth = len({k.split(".")[1] for k in self.p if k.startswith("layers.")})

    @_io
    def __call__(self, hidden_states):
        b, t, _ = hidden_states.shape
        e = self.p["latents"].shape[-1]
        ctx = self.rows16(hidden_states)
        if "proj_in.weight" in self.p:
            ctx = self.gemm(ctx, self.w16("proj_in"), self.b32("proj_in"))
        lat = self.rows16(self.p["latents"].expand(b, -1, -1))
        for i in range(self.depth):
            lp = f"layers.{i}"
            lat = self.mha(lp + ".cross_attn", lat, ctx, b, self.num_tokens, t, self.num_heads, q_norm=lp + ".norm1",
                           residual=lat)
            w0, c0, b0 = self.folded(lp + ".ff0", [lp + ".ff.0.weight"], [lp + ".ff.0.bias"], lp + ".norm2")
            h = self.gemm(lat, w0, b0, flags=L.EPI_GELU, ln_c1=c0)
            lat = self.gemm(h, self.w16(lp + ".ff.2"), self.b32(lp + ".ff.2"), residual=lat)
        return self.out32(self.layernorm(lat, "norm_out"), (b, self.num_tokens, e))

    forward = __call__


class ImageProjection(_Params):
    def __init__(self, sd, device, num_tokens=16, prefix="image_projection", be=None):
        super().__init__(sd, prefix, device, be)
        self.num_tokens = num_tokens
        self.cross_attention_dim = self.p["norm.weight"].shape[0]

    @_io
    def __call__(self, image_embeds):
        b = image_embeds.shape[0]
        x = self.gemm(self.rows16(image_embeds), self.w16("projection"), self.b32("projection"))
        x = x.view(1, b * self.num_tokens, 1, self.cross_attention_dim)
        return self.out32(self.layernorm(x, "norm"), (b, self.num_tokens, self.cross_attention_dim))

    forward = __call__


CLIP_VIT_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24,
                    num_attention_heads=16, image_size=224, patch_size=14, projection_dim=768)


def clip_config_from_state_dict(sd) -> dict:
    """CLIP vision config read off the tensor shapes of an ``image_encoder.image_encoder.*`` slice (the reference
    takes it from ``diff_cfg.image_encoder_path`` on the hub, image_encoder.py:34-42; offline the checkpoint is
    the only source).  Head width is 64 in every released CLIP ViT."""
    emb = "vision_model.embeddings."
    hidden = sd[emb + "class_embedding"].shape[0]
    patch = sd[emb + "patch_embedding.weight"].shape[-1]
    n_pos = sd[emb + "position_embedding.weight"].shape[0]
    layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith("vision_model.encoder.layers."))
    return dict(hidden_size=hidden, intermediate_size=sd["vision_model.encoder.layers.0.mlp.fc1.weight"].shape[0],
                num_hidden_layers=layers, num_attention_heads=max(1, hidden // 64),
                image_size=int(round((n_pos - 1) ** 0.5)) * patch, patch_size=patch,
                projection_dim=sd["visual_projection.weight"].shape[0])


class ImageEncoder(_Params):
    """Frozen CLIP vision tower (transformers ``CLIPVisionModelWithProjection``, image_encoder.py:34-88) on the HIP
    kernels: patch rows -> patch-embedding GEMM (+ class / position rows as its residual) -> pre-LayerNorm -> L x
    [LN1 folded into the fused q|k|v GEMM -> flash attention (d = hidden / heads) -> out_proj + residual -> LN2
    folded into fc1 with the quick-GELU epilogue -> fc2 + residual] (five launches per layer) -> hidden_states[-1];
    ``image_embeds`` = visual_projection(post_layernorm(class token)).  The state dict carries the tower under
    ``image_encoder.image_encoder.*`` (transformers key names)."""

    def __init__(self, sd, device, clip_config: Optional[dict] = None, prefix="image_encoder.image_encoder", be=None):
        super().__init__(sd, prefix, device, be)
        cfg = dict(clip_config or clip_config_from_state_dict({k: v for k, v in self.p.items()}))
        self.cfg = cfg
        self.hidden_size, self.projection_dim = cfg["hidden_size"], cfg["projection_dim"]
        self.heads, self.layers, self.patch = cfg["num_attention_heads"], cfg["num_hidden_layers"], cfg["patch_size"]
        if (self.hidden_size // self.heads) not in (40, 64, 80, 96, 160):
            raise ValueError(f"CLIP head width {self.hidden_size // self.heads} has no attention kernel (64 in released towers)")
        self.eps = float(cfg.get("layer_norm_eps", 1e-5))
        self.kp = -(-3 * self.patch * self.patch // 64) * 64
        e = "vision_model.embeddings."
        w = self.p[e + "patch_embedding.weight"].reshape(self.hidden_size, -1)
        wp = torch.zeros(self.hidden_size, self.kp, dtype=F16, device=w.device)
        wp[:, : w.shape[1]] = w.to(F16)
        self.w_patch = wp
        tok = self.p[e + "position_embedding.weight"].clone()
        tok[0] += self.p[e + "class_embedding"]
        self.tok_rows = tok.to(F16).contiguous()          # residual of the patch GEMM: cls + pos[0] | pos[1:]

    def _layer(self, i, x):
        lp = f"vision_model.encoder.layers.{i}."
        h = self.hidden_size
        b_t = x.shape[1]
        wqkv, c1, bqkv = self.folded(f"qkv{i}", [lp + f"self_attn.{n}_proj.weight" for n in "qkv"],
                                     [lp + f"self_attn.{n}_proj.bias" for n in "qkv"], lp + "layer_norm1")
        qkv = self.gemm(x, wqkv, bqkv, ln_c1=c1, eps=self.eps)
        bsz = self._bsz
        t = b_t // bsz
        att = self.be.empty((bsz, t, h), F16)
        q3 = qkv.view(bsz, t, 3 * h)
        self.be.attention(q3[:, :, :h], q3[:, :, h:2 * h], q3[:, :, 2 * h:], att, self.heads)
        x2 = self.gemm(att.view(1, b_t, 1, h), self.w16(lp + "self_attn.out_proj"), self.b32(lp + "self_attn.out_proj"),
                       residual=x)
        w1, c2, b1 = self.folded(f"fc1{i}", [lp + "mlp.fc1.weight"], [lp + "mlp.fc1.bias"], lp + "layer_norm2")
        hmid = self.gemm(x2, w1, b1, flags=L.EPI_QUICKGELU, ln_c1=c2, eps=self.eps)
        return self.gemm(hmid, self.w16(lp + "mlp.fc2"), self.b32(lp + "mlp.fc2"), residual=x2)

    def _tower(self, clip_images):
        """-> fp16 rows [1, B*T, 1, H] of hidden_states[-1], B, T."""
        px = self.be.to_device(clip_images.detach().float())
        b = px.shape[0]
        if b > 1 and clip_images.stride(0) == 0:      # one structure image expanded over the batch (:377): encode once
            px, b = px[:1].contiguous(), 1
        t = 1 + (px.shape[2] // self.patch) * (px.shape[3] // self.patch)
        if t != self.tok_rows.shape[0]:
            raise ValueError(f"pixel_values give {t} tokens, the tower has {self.tok_rows.shape[0]} position embeddings")
        self._bsz = b
        rows = self.be.empty((b, t, self.kp), F16)
        self.be.clip_patch_rows(px, rows, self.patch)
        res = self.rows16(self.tok_rows.expand(b, -1, -1)) if b > 1 else self.tok_rows.view(1, t, 1, -1)
        x = self.gemm(rows.view(1, b * t, 1, self.kp), self.w_patch, None, residual=res)
        out = self.be.empty(tuple(x.shape), F16)
        self.be.layernorm(x, self.p["vision_model.pre_layrnorm.weight"], self.p["vision_model.pre_layrnorm.bias"], out, self.eps)
        x = out
        for i in range(self.layers):
            x = self._layer(i, x)
        return x, b, t

    @_io
    def get_hidden_states(self, clip_images):
        x, b, t = self._tower(clip_images)
        hs = self.out32(x, (b, t, self.hidden_size))
        return hs.expand(clip_images.shape[0], -1, -1) if b != clip_images.shape[0] else hs

    @_io
    def __call__(self, clip_images):
        x, b, t = self._tower(clip_images)
        pooled = self.be.empty((1, b, 1, self.hidden_size), F16)
        self.be.copy_(pooled, x.view(b, t, self.hidden_size)[:, 0])
        ln = self.be.empty(tuple(pooled.shape), F16)
        self.be.layernorm(pooled, self.p["vision_model.post_layernorm.weight"], self.p["vision_model.post_layernorm.bias"],
                          ln, self.eps)
        emb = self.out32(self.gemm(ln, self.w16("visual_projection"), None), (b, self.projection_dim))
        return emb.expand(clip_images.shape[0], -1) if b != clip_images.shape[0] else emb

    forward = __call__
