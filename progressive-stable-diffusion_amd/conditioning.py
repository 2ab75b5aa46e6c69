"""Conditioning front-end of the sampler: AOE, image projection, FeaturePurifier, CLIP wrapper.

These run ONCE per batch (not per denoising step), on the device, as
fp32 torch tensor algebra; the fused HIP version of this front-end is the next item of the scope
table (§8f-1).  Call signatures mirror the reference classes so that
``_prepare_conditioning`` reads the same on both sides:
  * ``AdditiveOrdinalEmbedder``  — src/models/ordinal_embedder.py:43-309
  * ``FeaturePurifier``          — src/models/feature_purifier.py:29-95
  * ``ImageProjection[Plus]``    — src/models/image_encoder.py:91-228
  * ``ImageEncoder``             — src/models/image_encoder.py:17-88 (CLIP ViT-L/14 via transformers)
"""
from __future__ import annotations

import math
from typing import Dict, Iterator, Optional

import torch
import torch.nn.functional as F


class _Params:
    """Holds the slice of a flat state dict under ``prefix`` (fp32, on one device)."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, device):
        n = len(prefix) + 1
        self.p = {k[n:]: v.detach().to(device=device, dtype=torch.float32)
                  for k, v in sd.items() if k.startswith(prefix + ".")}
        if not self.p:
            raise KeyError(f"no parameters under '{prefix}.' in the state dict")

    def parameters(self) -> Iterator[torch.Tensor]:
        return iter(self.p.values())

    def to(self, *a, **k):
        self.p = {n: v.to(*a, **k) for n, v in self.p.items()}
        return self

    def lin(self, key, x):
        return F.linear(x, self.p[key + ".weight"], self.p.get(key + ".bias"))

    def ln(self, key, x):
        w = self.p[key + ".weight"]
        return F.layer_norm(x, (w.shape[0],), w, self.p[key + ".bias"], 1e-5)

    def mha(self, key, q_in, kv_in, heads):
        """Packed-projection multi-head attention (torch ``nn.MultiheadAttention`` semantics)."""
        e = q_in.shape[-1]
        w, b = self.p[key + ".in_proj_weight"], self.p[key + ".in_proj_bias"]
        q, k, v = (F.linear(src, w[i * e:(i + 1) * e], b[i * e:(i + 1) * e])
                   for i, src in enumerate((q_in, kv_in, kv_in)))
        bs, dh = q.shape[0], e // heads
        q, k, v = (t.reshape(bs, -1, heads, dh).transpose(1, 2) for t in (q, k, v))
        att = torch.softmax(q @ k.transpose(-1, -2) * (1.0 / math.sqrt(dh)), dim=-1) @ v
        return self.lin(key + ".out_proj", att.transpose(1, 2).reshape(bs, -1, e))


class AdditiveOrdinalEmbedder(_Params):
    def __init__(self, sd, device, num_classes=4, embedding_dim=768, num_tokens=16,
                 prefix="ordinal_embedder"):
        super().__init__(sd, prefix, device)
        if num_classes < 2:
            raise ValueError("num_classes must be ≥ 2 for ordinal modeling.")
        self.num_classes, self.embedding_dim, self.num_tokens = num_classes, embedding_dim, num_tokens

    def _class_table(self):
        steps = torch.cumsum(self.p["deltas"], dim=0)
        return self.p["base"] + torch.cat([torch.zeros_like(steps[:1]), steps], dim=0)

    def _interp(self, labels):
        tab = self._class_table()
        top = self.num_classes - 1
        y = labels.to(tab).clamp(0.0, float(top))
        lo = y.floor()
        frac = (y - lo)[..., None]
        lo_i = lo.long()
        hi_i = (lo_i + 1).clamp(max=top)
        return tab[lo_i] * (1.0 - frac) + tab[hi_i] * frac

    def _tokens(self, emb):
        h = self.lin("projector.2", F.gelu(self.lin("projector.0", emb)))
        return h.view(-1, self.num_tokens, self.embedding_dim)

    def __call__(self, labels, is_training=False, unconditional=False, noise_std=0.005):
        if unconditional:
            return self.p["null_embedding"].expand(labels.shape[0] if labels.dim() else 1, -1)
        scalar = labels.dim() == 0
        emb = self._interp(labels[None] if scalar else labels)
        if is_training and noise_std > 0:
            emb = emb + torch.randn_like(emb) * noise_std
        out = self._tokens(emb)
        return out[0] if scalar else out

    forward = __call__

    def get_negative_embedding(self, labels, is_training=False, noise_std=0.005):
        scalar = labels.dim() == 0
        lab = labels[None] if scalar else labels
        return self(torch.clamp(1.0 - lab, 0.0, 1.0), is_training=is_training, noise_std=noise_std)

    def get_ordinal_delta_embedding(self, source_labels, target_labels):
        scalar = source_labels.dim() == 0
        if scalar:
            source_labels, target_labels = source_labels[None], target_labels[None]
        d = self._tokens(self._interp(target_labels)) - self._tokens(self._interp(source_labels))
        return d[0] if scalar else d

    def get_disease_delta_embedding(self, source_labels):
        return self.get_ordinal_delta_embedding(source_labels, torch.zeros_like(source_labels))


class FeaturePurifier(_Params):
    def __init__(self, sd, device, num_heads=8, prefix="feature_purifier"):
        super().__init__(sd, prefix, device)
        self.num_heads = num_heads

    def __call__(self, image_embeds, source_aoe):
        img = self.ln("norm_img", image_embeds)
        dis = self.mha("cross_attn", img, self.ln("norm_aoe", source_aoe), self.num_heads)
        gate = torch.sigmoid(self.lin("gate.2", F.gelu(self.lin("gate.0", torch.cat([dis, img], -1)))))
        return self.ln("norm_out", image_embeds - gate * dis)

    forward = __call__


class ImageProjectionPlus(_Params):
    def __init__(self, sd, device, num_tokens=16, num_heads=8, prefix="image_projection"):
        super().__init__(sd, prefix, device)
        self.num_tokens, self.num_heads = num_tokens, num_heads
        self.depth = len({k.split(".")[1] for k in self.p if k.startswith("layers.")})

    def __call__(self, hidden_states):
        ctx = self.lin("proj_in", hidden_states) if "proj_in.weight" in self.p else hidden_states
        lat = self.p["latents"].expand(hidden_states.shape[0], -1, -1)
        for i in range(self.depth):
            lp = f"layers.{i}"
            lat = lat + self.mha(lp + ".cross_attn", self.ln(lp + ".norm1", lat), ctx, self.num_heads)
            lat = lat + self.lin(lp + ".ff.2", F.gelu(self.lin(lp + ".ff.0", self.ln(lp + ".norm2", lat))))
        return self.ln("norm_out", lat)

    forward = __call__


class ImageProjection(_Params):
    def __init__(self, sd, device, num_tokens=16, prefix="image_projection"):
        super().__init__(sd, prefix, device)
        self.num_tokens = num_tokens
        self.cross_attention_dim = self.p["norm.weight"].shape[0]

    def __call__(self, image_embeds):
        x = self.lin("projection", image_embeds).reshape(-1, self.num_tokens, self.cross_attention_dim)
        return self.ln("norm", x)

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


class ImageEncoder:
    """Frozen CLIP vision tower.  The reference loads ``openai/clip-vit-large-patch14`` from the hub
    (image_encoder.py:34-42); offline the same architecture is built from its config with seeded
    random weights (``clip_config`` overrides it, e.g. a 2-layer tower in the tests)."""

    def __init__(self, device, seed: int = 0, clip_config: Optional[dict] = None, state_dict=None):
        from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
        if clip_config is None and state_dict is not None:
            clip_config = clip_config_from_state_dict(state_dict)
        cfg = CLIPVisionConfig(**(clip_config or CLIP_VIT_L14))
        rng = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self.image_encoder = CLIPVisionModelWithProjection(cfg)
        torch.random.set_rng_state(rng)
        if state_dict is not None:
            own = self.image_encoder.state_dict()
            sd = {k: v for k, v in state_dict.items() if not k.endswith("position_ids")}   # buffer in old layouts
            lacking = [k for k in own if k not in sd and not k.endswith("position_ids")]
            extra = [k for k in sd if k not in own]
            if lacking or extra:
                raise KeyError(f"CLIP tower state dict mismatch: missing {lacking[:3]} unexpected {extra[:3]}")
            self.image_encoder.load_state_dict(sd, strict=False)
        self.image_encoder.requires_grad_(False).eval().to(device=device, dtype=torch.float32)
        self.hidden_size = cfg.hidden_size
        self.projection_dim = cfg.projection_dim

    def parameters(self):
        return self.image_encoder.parameters()

    def to(self, *a, **k):
        self.image_encoder.to(*a, **k)
        return self

    @torch.no_grad()
    def __call__(self, clip_images):
        return self.image_encoder(pixel_values=clip_images, output_hidden_states=True).image_embeds

    forward = __call__

    @torch.no_grad()
    def get_hidden_states(self, clip_images):
        return self.image_encoder(pixel_values=clip_images, output_hidden_states=True).hidden_states[-1]
