"""Oracle (test infrastructure): SD-1.x ``AutoencoderKL`` encode / decode, fp32, CPU.

PARITY UNPINNED at this boundary (third-party ``diffusers>=0.31.0``, reference
``pyproject.toml:27``; absent here; no reference fixture).  Restates the published SD-1.4
VAE with diffusers key names; anchored by the reference call
sites ``src/models/vae/vae.py:71-112`` and the 83.7 M parameter count.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

GROUPS = 32
EPS = 1e-6
VAE_CH = (128, 256, 512, 512)


def _gn(sd, key, x):
    return F.group_norm(x, GROUPS, sd[key + ".weight"], sd[key + ".bias"], EPS)


def _conv(sd, key, x, stride=1, padding=1):
    return F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=stride, padding=padding)


def _res(sd, p, x):
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x)))
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h)))
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def _mid_attn(sd, p, x):
    """Single-head (d = C = 512) spatial self-attention with GroupNorm and residual."""
    b, c, hh, ww = x.shape
    h = _gn(sd, p + ".group_norm", x).view(b, c, hh * ww).transpose(1, 2)
    q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    a = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(c), dim=-1) @ v
    a = F.linear(a, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return a.transpose(1, 2).reshape(b, c, hh, ww) + x


def _mid(sd, p, x):
    x = _res(sd, p + ".resnets.0", x)
    x = _mid_attn(sd, p + ".attentions.0", x)
    return _res(sd, p + ".resnets.1", x)


def vae_decode(sd, z, prefix="vae.vae"):
    """z (B,4,S,S) (already divided by latent_scale) -> (B,3,8S,8S)."""
    v = prefix + "."
    h = _conv(sd, v + "post_quant_conv", z, padding=0)
    h = _conv(sd, v + "decoder.conv_in", h)
    h = _mid(sd, v + "decoder.mid_block", h)
    for i in range(4):
        for j in range(3):
            h = _res(sd, v + f"decoder.up_blocks.{i}.resnets.{j}", h)
        if i < 3:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, v + f"decoder.up_blocks.{i}.upsamplers.0.conv", h)
    h = F.silu(_gn(sd, v + "decoder.conv_norm_out", h))
    return _conv(sd, v + "decoder.conv_out", h)


def vae_encode_moments(sd, x, prefix="vae.vae"):
    """x (B,3,H,W) in [-1,1] -> (mean, logvar) each (B,4,H/8,W/8); logvar clamped to [-30,20]."""
    v = prefix + "."
    h = _conv(sd, v + "encoder.conv_in", x)
    for i in range(4):
        for j in range(2):
            h = _res(sd, v + f"encoder.down_blocks.{i}.resnets.{j}", h)
        if i < 3:
            h = F.pad(h, (0, 1, 0, 1))
            h = _conv(sd, v + f"encoder.down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=0)
    h = _mid(sd, v + "encoder.mid_block", h)
    h = F.silu(_gn(sd, v + "encoder.conv_norm_out", h))
    h = _conv(sd, v + "encoder.conv_out", h)
    m = _conv(sd, v + "quant_conv", h, padding=0)
    mean, logvar = m.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def vae_encode_sample(sd, x, noise, prefix="vae.vae"):
    """``latent_dist.sample()`` with the noise injected: mean + exp(0.5*logvar)*noise."""
    mean, logvar = vae_encode_moments(sd, x, prefix)
    return mean + torch.exp(0.5 * logvar) * noise
