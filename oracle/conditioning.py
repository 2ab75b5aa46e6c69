"""Oracle (test infrastructure): conditioning front-end of the DADD sampler.

Pure functions over a flat state dict ``sd`` (checkpoint key layout).
PINNED against the imported reference modules via ``tests/golden/conditioning_*.npz``.

Reference followed (relative to the reference repo root):
  * ``src/models/ordinal_embedder.py:15-40,107-180,182-221,246-294``  (AOE)
  * ``src/models/feature_purifier.py:64-95``                          (FeaturePurifier)
  * ``src/models/image_encoder.py:123-133,205-228``                   (ImageProjection[Plus])
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _lin(sd, key, x):
    return F.linear(x, sd[key + ".weight"], sd.get(key + ".bias"))


def _ln(sd, key, x, eps=1e-5):
    w = sd[key + ".weight"]
    return F.layer_norm(x, (w.shape[0],), w, sd[key + ".bias"], eps)


def mha(sd, key, q_in, kv_in, heads):
    """``nn.MultiheadAttention(batch_first=True)`` with packed in_proj, no mask, no dropout."""
    e = q_in.shape[-1]
    w, b = sd[key + ".in_proj_weight"], sd[key + ".in_proj_bias"]
    q = F.linear(q_in, w[:e], b[:e])
    k = F.linear(kv_in, w[e:2 * e], b[e:2 * e])
    v = F.linear(kv_in, w[2 * e:], b[2 * e:])
    bsz, nq, _ = q.shape
    nk = k.shape[1]
    dh = e // heads
    q = q.view(bsz, nq, heads, dh).transpose(1, 2)
    k = k.view(bsz, nk, heads, dh).transpose(1, 2)
    v = v.view(bsz, nk, heads, dh).transpose(1, 2)
    p = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(bsz, nq, e)
    return _lin(sd, key + ".out_proj", o)


# --------------------------------------------------------------------------- AOE
def aoe_table(sd, p="ordinal_embedder"):
    """E[k] = base + sum(deltas[:k])  (ordinal_embedder.py:107-127)."""
    deltas = sd[p + ".deltas"]
    offs = torch.cat([torch.zeros_like(deltas[:1]), torch.cumsum(deltas, 0)], 0)
    return sd[p + ".base"][None, :] + offs


def aoe_lerp(sd, labels, p="ordinal_embedder"):
    """Clamp, floor, lerp between neighbouring class rows (ordinal_embedder.py:15-40,155-171)."""
    table = aoe_table(sd, p)
    kmax = table.shape[0] - 1
    y = labels.to(table.dtype).clamp(0.0, float(kmax))
    lo = torch.floor(y)
    hi = torch.clamp(lo + 1, max=kmax)
    a = (y - lo).unsqueeze(-1)
    return table[lo.long()] * (1.0 - a) + table[hi.long()] * a


def aoe_project(sd, emb, num_tokens, p="ordinal_embedder"):
    """Linear(D,2D) -> exact GELU -> Linear(2D, T*D) -> (B,T,D)  (ordinal_embedder.py:80-84,177-178)."""
    h = F.gelu(_lin(sd, p + ".projector.0", emb))
    h = _lin(sd, p + ".projector.2", h)
    return h.view(-1, num_tokens, emb.shape[-1])


def aoe_forward(sd, labels, num_tokens=16, p="ordinal_embedder"):
    """Inference-mode AOE tokens (is_training=False: no noise) (ordinal_embedder.py:129-180)."""
    return aoe_project(sd, aoe_lerp(sd, labels, p), num_tokens, p)


def aoe_negative(sd, labels, num_tokens=16, p="ordinal_embedder"):
    """CFG negative: label' = clamp(1-label, 0, 1)  (ordinal_embedder.py:182-221)."""
    return aoe_forward(sd, torch.clamp(1.0 - labels, min=0.0, max=1.0), num_tokens, p)


def aoe_delta(sd, source, target, num_tokens=16, p="ordinal_embedder"):
    """proj(E[target]) - proj(E[source]), subtraction after projection (ordinal_embedder.py:246-294)."""
    return (aoe_project(sd, aoe_lerp(sd, target, p), num_tokens, p)
            - aoe_project(sd, aoe_lerp(sd, source, p), num_tokens, p))


# --------------------------------------------------------------------------- purifier
def feature_purifier(sd, image_embeds, source_aoe, heads=8, p="feature_purifier"):
    """LN -> MHA(q=img, kv=aoe) -> sigmoid gate MLP -> img - gate*d -> LN (feature_purifier.py:64-95)."""
    img_n = _ln(sd, p + ".norm_img", image_embeds)
    aoe_n = _ln(sd, p + ".norm_aoe", source_aoe)
    d = mha(sd, p + ".cross_attn", img_n, aoe_n, heads)
    g = _lin(sd, p + ".gate.0", torch.cat([d, img_n], -1))
    g = torch.sigmoid(_lin(sd, p + ".gate.2", F.gelu(g)))
    return _ln(sd, p + ".norm_out", image_embeds - g * d)


# --------------------------------------------------------------------------- projections
def image_projection_plus(sd, hidden, heads=8, p="image_projection"):
    """Perceiver resampler: proj_in, depth x {pre-LN MHA + res, pre-LN MLP + res}, LN (image_encoder.py:205-228)."""
    x = _lin(sd, p + ".proj_in", hidden) if (p + ".proj_in.weight") in sd else hidden
    lat = sd[p + ".latents"].expand(hidden.shape[0], -1, -1)
    i = 0
    while (p + f".layers.{i}.norm1.weight") in sd:
        lp = p + f".layers.{i}"
        lat = lat + mha(sd, lp + ".cross_attn", _ln(sd, lp + ".norm1", lat), x, heads)
        h = F.gelu(_lin(sd, lp + ".ff.0", _ln(sd, lp + ".norm2", lat)))
        lat = lat + _lin(sd, lp + ".ff.2", h)
        i += 1
    return _ln(sd, p + ".norm_out", lat)


def image_projection(sd, image_embeds, num_tokens=16, p="image_projection"):
    """Linear -> (B,T,D) -> LN  (image_encoder.py:123-133)."""
    d = sd[p + ".norm.weight"].shape[0]
    x = _lin(sd, p + ".projection", image_embeds).reshape(-1, num_tokens, d)
    return _ln(sd, p + ".norm", x)
