"""Oracle (test infrastructure): SD-1.x ``UNet2DConditionModel`` forward, fp32, CPU.

PARITY UNPINNED at this boundary: the arithmetic lives in the third-party ``diffusers``
package (``diffusers>=0.31.0``, reference ``pyproject.toml:27``), which is neither
vendored in the reference nor installed here, and the reference holds no fixture for
it.  This file restates the published SD-1.4 architecture with
diffusers' ``state_dict`` key names; it is anchored by the reference call site
``src/models/unet/unet.py:122-146`` (argument normalisation restated in
``normalise_unet_args``), by the block/width facts the reference itself encodes
(``src/models/attention_processor_routing_gates.py:207-215,273-282``) and by the
859.5 M parameter count (``tests/test_weights.py``).
The 16 ``attn2`` sites call the PINNED processors in ``oracle/processors.py``.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import processors as P

BLOCK_OUT = (320, 640, 1280, 1280)
HEADS = 8
GROUPS = 32


def normalise_unet_args(t: torch.Tensor, cond: torch.Tensor):
    """OrdinalUNet.forward argument handling (src/models/unet/unet.py:122-138)."""
    if cond.ndim == 2:
        cond = cond.unsqueeze(1)
    elif cond.ndim != 3:
        raise ValueError(f"cond_embed must have shape (B, D) or (B, seq_len, D), got {cond.shape}")
    if t.ndim == 0:
        t = t[None]
    elif t.ndim > 1:
        t = t.view(-1)
    return t, cond


def timestep_embedding(t: torch.Tensor, dim: int = 320) -> torch.Tensor:
    """diffusers ``Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)``."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    ang = t.float()[:, None] * freqs[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def _gn(sd, key, x, eps):
    return F.group_norm(x, GROUPS, sd[key + ".weight"], sd[key + ".bias"], eps)


def _conv(sd, key, x, stride=1, padding=1):
    return F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=padding)


def resnet(sd, p, x, temb, eps=1e-5):
    """ResnetBlock2D: GN-SiLU-conv3x3 (+temb proj) GN-SiLU-conv3x3, 1x1 shortcut iff Cin != Cout."""
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, eps)))
    if temb is not None:
        h = h + F.linear(F.silu(temb), sd[p + ".time_emb_proj.weight"],
                         sd[p + ".time_emb_proj.bias"])[:, :, None, None]
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, eps)))
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def transformer(sd, p, x, cond, attn2):
    """Transformer2DModel with one BasicTransformerBlock (conv 1x1 proj_in/out, GEGLU FF)."""
    b, c, hh, ww = x.shape
    res = x
    h = _conv(sd, p + ".proj_in", _gn(sd, p + ".norm", x, 1e-6), padding=0)
    h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    tb = p + ".transformer_blocks.0"

    def ln(k, v):
        return F.layer_norm(v, (c,), sd[tb + k + ".weight"], sd[tb + k + ".bias"], 1e-5)

    h = h + P.self_attention(sd, tb + ".attn1", ln(".norm1", h), HEADS)
    h = h + attn2(tb + ".attn2", ln(".norm2", h), cond)
    y = F.linear(ln(".norm3", h), sd[tb + ".ff.net.0.proj.weight"], sd[tb + ".ff.net.0.proj.bias"])
    hid, gate = y.chunk(2, dim=-1)
    h = h + F.linear(hid * F.gelu(gate), sd[tb + ".ff.net.2.weight"], sd[tb + ".ff.net.2.bias"])
    h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    return _conv(sd, p + ".proj_out", h, padding=0) + res


def unet_forward(sd, x, t, cond, *, prefix="unet.unet", use_routing_gates=True,
                 delta_scale=0.0, use_frequency_strategy=True, taps=None):
    """epsilon prediction.  x (B,4,S,S) fp32, t (B,) int, cond (B,48|32,768).

    ``taps``: optional dict that receives named intermediates (for layer-wise parity tests).
    """
    t, cond = normalise_unet_args(t, cond)
    u = prefix + "."

    def attn2(ap, h, c):
        if use_routing_gates:
            return P.split_injection_attention(sd, ap, h, c, HEADS, delta_scale)
        mode = P.frequency_mode(ap[len(u):]) if use_frequency_strategy else "both"
        return P.ordinal_ip_attention(sd, ap, h, c, HEADS, mode)

    def tap(name, v):
        if taps is not None:
            taps[name] = v

    temb = timestep_embedding(t.expand(x.shape[0]) if t.shape[0] == 1 else t)
    temb = F.linear(temb, sd[u + "time_embedding.linear_1.weight"], sd[u + "time_embedding.linear_1.bias"])
    temb = F.linear(F.silu(temb), sd[u + "time_embedding.linear_2.weight"],
                    sd[u + "time_embedding.linear_2.bias"])
    tap("temb", temb)

    h = _conv(sd, u + "conv_in", x)
    tap("conv_in", h)
    skips = [h]
    for i in range(4):
        bp = u + f"down_blocks.{i}"
        for j in range(2):
            h = resnet(sd, f"{bp}.resnets.{j}", h, temb)
            if i < 3:
                h = transformer(sd, f"{bp}.attentions.{j}", h, cond, attn2)
            skips.append(h)
            tap(f"down{i}.{j}", h)
        if i < 3:
            h = _conv(sd, f"{bp}.downsamplers.0.conv", h, stride=2, padding=1)
            skips.append(h)

    h = resnet(sd, u + "mid_block.resnets.0", h, temb)
    h = transformer(sd, u + "mid_block.attentions.0", h, cond, attn2)
    h = resnet(sd, u + "mid_block.resnets.1", h, temb)
    tap("mid", h)

    for i in range(4):
        bp = u + f"up_blocks.{i}"
        for j in range(3):
            h = resnet(sd, f"{bp}.resnets.{j}", torch.cat([h, skips.pop()], dim=1), temb)
            if i > 0:
                h = transformer(sd, f"{bp}.attentions.{j}", h, cond, attn2)
            tap(f"up{i}.{j}", h)
        if i < 3:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"{bp}.upsamplers.0.conv", h)

    h = F.silu(_gn(sd, u + "conv_norm_out", h, 1e-5))
    return _conv(sd, u + "conv_out", h)


def attn2_sites():
    """The 16 cross-attention sites as (block path, channels, latent-side divisor)."""
    out = []
    for i in range(3):
        for j in range(2):
            out.append((f"down_blocks.{i}.attentions.{j}", BLOCK_OUT[i], 2 ** i))
    out.append(("mid_block.attentions.0", 1280, 8))
    for i in (1, 2, 3):
        for j in range(3):
            out.append((f"up_blocks.{i}.attentions.{j}", BLOCK_OUT[3 - i], 2 ** (3 - i)))
    return out
