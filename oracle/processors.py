"""Oracle (test infrastructure): the reference's cross-attention processors + plain self-attention.

Pure functions over the flat state dict.  ``ap`` is the attention-layer prefix, e.g.
``unet.unet.down_blocks.0.attentions.0.transformer_blocks.0.attn2``; the processor's own
tensors live under ``ap + ".processor."``.
PINNED via ``tests/golden/xattn_*.npz`` (outputs of the imported reference classes).

Reference followed:
  * ``src/models/attention_processor_routing_gates.py:123-196``  SplitInjectionAttentionProcessor.__call__
  * ``src/models/attention_processor_base.py:85-136``            OrdinalIPAttnProcessor2_0.__call__
  * ``src/models/attention_processor_routing_gates.py:199-230``  get_block_type
  * ``src/models/attention_processor_base.py:141-167``           get_frequency_mode_for_block
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _heads(x, h):
    b, n, c = x.shape
    return x.view(b, n, h, c // h).transpose(1, 2)


def _merge(x):
    b, h, n, d = x.shape
    return x.transpose(1, 2).reshape(b, n, h * d)


def _sm_av(q, k, v):
    d = q.shape[-1]
    return torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(d), dim=-1) @ v


def self_attention(sd, ap, x, heads):
    """diffusers AttnProcessor2_0 on attn1: q/k/v no bias, SDPA, to_out with bias."""
    q = _heads(F.linear(x, sd[ap + ".to_q.weight"]), heads)
    k = _heads(F.linear(x, sd[ap + ".to_k.weight"]), heads)
    v = _heads(F.linear(x, sd[ap + ".to_v.weight"]), heads)
    o = _merge(_sm_av(q, k, v))
    return F.linear(o, sd[ap + ".to_out.0.weight"], sd[ap + ".to_out.0.bias"])


def split_injection_attention(sd, ap, x, cond, heads, delta_scale,
                              n_aoe=16, n_img=16, n_delta=16):
    """Triple-pathway cross-attention: three INDEPENDENT softmaxes, gate/lambda-weighted sum.

    Token slicing ``[:n_aoe]`` = disease, ``[n_aoe:n_aoe+n_img]`` = anatomy, ``[-n_delta:]`` = delta
    (routing_gates.py:129-131); anatomy uses ``to_k/to_v``, disease AND delta use
    ``to_k_dis/to_v_dis`` (:133-137,161-162); the delta pathway exists only when
    ``delta_scale != 0`` (:160).  SD-1.x attn2 has no spatial_norm/group_norm/residual and
    rescale_output_factor 1, so those branches (:95-121,191-194) are inert.
    """
    pp = ap + ".processor"
    q = _heads(F.linear(x, sd[ap + ".to_q.weight"]), heads)
    dis, anat, delta = cond[:, :n_aoe], cond[:, n_aoe:n_aoe + n_img], cond[:, -n_delta:]
    z = sd[pp + ".anat_gate"] * _sm_av(
        q, _heads(F.linear(anat, sd[ap + ".to_k.weight"]), heads),
        _heads(F.linear(anat, sd[ap + ".to_v.weight"]), heads))
    z = z + sd[pp + ".dis_gate"] * _sm_av(
        q, _heads(F.linear(dis, sd[pp + ".to_k_dis.weight"]), heads),
        _heads(F.linear(dis, sd[pp + ".to_v_dis.weight"]), heads))
    if delta_scale != 0.0:
        z = z + delta_scale * _sm_av(
            q, _heads(F.linear(delta, sd[pp + ".to_k_dis.weight"]), heads),
            _heads(F.linear(delta, sd[pp + ".to_v_dis.weight"]), heads))
    return F.linear(_merge(z), sd[ap + ".to_out.0.weight"], sd[ap + ".to_out.0.bias"])


def ordinal_ip_attention(sd, ap, x, cond, heads, frequency_mode="both"):
    """Baseline 2-segment [AOE|Image] cross-attention, one joint softmax (base.py:85-136).

    For mode != "both" the reference multiplies the probabilities by an all-ones vector
    (scale_aoe = scale_ip = 1, base.py:29-37) and renormalises (:103-116); that is kept
    here literally so rounding matches.
    """
    q = _heads(F.linear(x, sd[ap + ".to_q.weight"]), heads)
    k = _heads(F.linear(cond, sd[ap + ".to_k.weight"]), heads)
    v = _heads(F.linear(cond, sd[ap + ".to_v.weight"]), heads)
    p = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(q.shape[-1]), dim=-1)
    if frequency_mode != "both":
        p = p * torch.ones_like(p[:1, :1, :1])
        p = p / p.sum(-1, keepdim=True)
    return F.linear(_merge(p @ v), sd[ap + ".to_out.0.weight"], sd[ap + ".to_out.0.bias"])


def block_role(name: str) -> str:
    """anatomy / disease role by UNet position (routing_gates.py:199-230)."""
    if "mid_block" in name:
        return "disease"
    for tag, disease_if in (("down_blocks.", lambda i: i >= 2), ("up_blocks.", lambda i: i <= 1)):
        if tag in name:
            return "disease" if disease_if(int(name.split(tag)[1].split(".")[0])) else "anatomy"
    return "both"


def frequency_mode(name: str) -> str:
    """Baseline per-block mode table (base.py:141-167)."""
    if "mid_block" in name:
        return "aoe_dominant"
    for tag, aoe_if in (("down_blocks.", lambda i: i > 1), ("up_blocks.", lambda i: i <= 1)):
        if tag in name:
            try:
                i = int(name.split(tag)[1].split(".")[0])
            except (IndexError, ValueError):
                return "both"
            return "aoe_dominant" if aoe_if(i) else "image_dominant"
    return "both"
