"""Parameter inventory and seeded random initialisation for the DADD module.

The reference obtains its weights from the hub (``src/models/unet/unet.py:70-75``,
``src/models/vae/vae.py:60-65``, ``src/models/image_encoder.py:34-42``) and from a
Lightning checkpoint (``src/pipelines/inference/inference_pipeline_ip.py:587-592``);
neither exists offline, so the benchmark and the parity tests use seeded random
weights of the same architecture.  Key names follow the checkpoint layout the
reference produces, so a real ``state_dict`` is a plain load.

Every tensor is drawn from its own generator seeded by ``crc32(key) ^ seed``: any subset
of the dict can be regenerated bit-identically, in any order, on any host.
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Optional, Tuple

import torch

Shape = Tuple[int, ...]

UNET_BLOCK_OUT = (320, 640, 1280, 1280)
VAE_BLOCK_OUT = (128, 256, 512, 512)
TEMB_DIM = 1280
COND_DIM = 768


# ----------------------------------------------------------------------------- shapes
def _conv(d, k, cout, cin, ks):
    d[k + ".weight"] = (cout, cin, ks, ks)
    d[k + ".bias"] = (cout,)


def _lin(d, k, out, inp, bias=True):
    d[k + ".weight"] = (out, inp)
    if bias:
        d[k + ".bias"] = (out,)


def _norm(d, k, c):
    d[k + ".weight"] = (c,)
    d[k + ".bias"] = (c,)


def _resnet(d, p, cin, cout, temb=True):
    _norm(d, p + ".norm1", cin)
    _conv(d, p + ".conv1", cout, cin, 3)
    if temb:
        _lin(d, p + ".time_emb_proj", cout, TEMB_DIM)
    _norm(d, p + ".norm2", cout)
    _conv(d, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(d, p + ".conv_shortcut", cout, cin, 1)


def _transformer(d, p, c, routing_gates):
    _norm(d, p + ".norm", c)
    _conv(d, p + ".proj_in", c, c, 1)
    tb = p + ".transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        _norm(d, f"{tb}.{n}", c)
    for a, kv_in in (("attn1", c), ("attn2", COND_DIM)):
        _lin(d, f"{tb}.{a}.to_q", c, c, bias=False)
        _lin(d, f"{tb}.{a}.to_k", c, kv_in, bias=False)
        _lin(d, f"{tb}.{a}.to_v", c, kv_in, bias=False)
        _lin(d, f"{tb}.{a}.to_out.0", c, c)
    if routing_gates:
        # SplitInjectionAttentionProcessor state (attention_processor_routing_gates.py:74-82)
        d[f"{tb}.attn2.processor.anat_gate"] = ()
        d[f"{tb}.attn2.processor.dis_gate"] = ()
        _lin(d, f"{tb}.attn2.processor.to_k_dis", c, COND_DIM, bias=False)
        _lin(d, f"{tb}.attn2.processor.to_v_dis", c, COND_DIM, bias=False)
    _lin(d, f"{tb}.ff.net.0.proj", 8 * c, c)
    _lin(d, f"{tb}.ff.net.2", c, 4 * c)
    _conv(d, p + ".proj_out", c, c, 1)


def unet_shapes(prefix="unet.unet", routing_gates=True) -> "OrderedDict[str, Shape]":
    """SD-1.x UNet2DConditionModel parameters (+ DADD processor tensors), diffusers names."""
    d: "OrderedDict[str, Shape]" = OrderedDict()
    u = prefix + "."
    ch = UNET_BLOCK_OUT
    _conv(d, u + "conv_in", ch[0], 4, 3)
    _lin(d, u + "time_embedding.linear_1", TEMB_DIM, ch[0])
    _lin(d, u + "time_embedding.linear_2", TEMB_DIM, TEMB_DIM)
    skip = [ch[0]]
    cur = ch[0]
    for i in range(4):
        for j in range(2):
            _resnet(d, u + f"down_blocks.{i}.resnets.{j}", cur, ch[i])
            cur = ch[i]
            if i < 3:
                _transformer(d, u + f"down_blocks.{i}.attentions.{j}", cur, routing_gates)
            skip.append(cur)
        if i < 3:
            _conv(d, u + f"down_blocks.{i}.downsamplers.0.conv", cur, cur, 3)
            skip.append(cur)
    _resnet(d, u + "mid_block.resnets.0", cur, cur)
    _transformer(d, u + "mid_block.attentions.0", cur, routing_gates)
    _resnet(d, u + "mid_block.resnets.1", cur, cur)
    rev = tuple(reversed(ch))
    for i in range(4):
        for j in range(3):
            _resnet(d, u + f"up_blocks.{i}.resnets.{j}", cur + skip.pop(), rev[i])
            cur = rev[i]
            if i > 0:
                _transformer(d, u + f"up_blocks.{i}.attentions.{j}", cur, routing_gates)
        if i < 3:
            _conv(d, u + f"up_blocks.{i}.upsamplers.0.conv", cur, cur, 3)
    _norm(d, u + "conv_norm_out", cur)
    _conv(d, u + "conv_out", 4, cur, 3)
    return d


def _vae_mid(d, p, c):
    _resnet(d, p + ".resnets.0", c, c, temb=False)
    a = p + ".attentions.0"
    _norm(d, a + ".group_norm", c)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(d, f"{a}.{n}", c, c)
    _resnet(d, p + ".resnets.1", c, c, temb=False)


def vae_shapes(prefix="vae.vae", encoder=True, decoder=True) -> "OrderedDict[str, Shape]":
    """SD-1.x AutoencoderKL parameters, diffusers names."""
    d: "OrderedDict[str, Shape]" = OrderedDict()
    v = prefix + "."
    ch = VAE_BLOCK_OUT
    if encoder:
        _conv(d, v + "encoder.conv_in", ch[0], 3, 3)
        cur = ch[0]
        for i in range(4):
            for j in range(2):
                _resnet(d, v + f"encoder.down_blocks.{i}.resnets.{j}", cur, ch[i], temb=False)
                cur = ch[i]
            if i < 3:
                _conv(d, v + f"encoder.down_blocks.{i}.downsamplers.0.conv", cur, cur, 3)
        _vae_mid(d, v + "encoder.mid_block", cur)
        _norm(d, v + "encoder.conv_norm_out", cur)
        _conv(d, v + "encoder.conv_out", 8, cur, 3)
        _conv(d, v + "quant_conv", 8, 8, 1)
    if decoder:
        _conv(d, v + "post_quant_conv", 4, 4, 1)
        rev = tuple(reversed(ch))
        _conv(d, v + "decoder.conv_in", rev[0], 4, 3)
        _vae_mid(d, v + "decoder.mid_block", rev[0])
        cur = rev[0]
        for i in range(4):
            for j in range(3):
                _resnet(d, v + f"decoder.up_blocks.{i}.resnets.{j}", cur, rev[i], temb=False)
                cur = rev[i]
            if i < 3:
                _conv(d, v + f"decoder.up_blocks.{i}.upsamplers.0.conv", cur, cur, 3)
        _norm(d, v + "decoder.conv_norm_out", cur)
        _conv(d, v + "decoder.conv_out", 3, cur, 3)
    return d


def _mha(d, p, e):
    d[p + ".in_proj_weight"] = (3 * e, e)
    d[p + ".in_proj_bias"] = (3 * e,)
    _lin(d, p + ".out_proj", e, e)


def conditioning_shapes(num_classes=4, dim=COND_DIM, num_tokens=16, clip_hidden=1024,
                        clip_proj=768, projection_plus=True, purifier=True, purifier_ff_mult=2,
                        resampler_depth=2) -> "OrderedDict[str, Shape]":
    """AOE / image projection / purifier parameters (reference module attribute names)."""
    d: "OrderedDict[str, Shape]" = OrderedDict()
    p = "ordinal_embedder"
    d[p + ".base"] = (dim,)
    d[p + ".deltas"] = (num_classes - 1, dim)
    d[p + ".null_embedding"] = (1, dim)
    _lin(d, p + ".projector.0", 2 * dim, dim)
    _lin(d, p + ".projector.2", num_tokens * dim, 2 * dim)
    _norm(d, p + ".norm", num_tokens * dim)
    p = "image_projection"
    if projection_plus:
        d[p + ".latents"] = (1, num_tokens, dim)
        if clip_hidden != dim:
            _lin(d, p + ".proj_in", dim, clip_hidden)
        for i in range(resampler_depth):
            lp = f"{p}.layers.{i}"
            _mha(d, lp + ".cross_attn", dim)
            _lin(d, lp + ".ff.0", 4 * dim, dim)
            _lin(d, lp + ".ff.2", dim, 4 * dim)
            _norm(d, lp + ".norm1", dim)
            _norm(d, lp + ".norm2", dim)
        _norm(d, p + ".norm_out", dim)
    else:
        _lin(d, p + ".projection", dim * num_tokens, clip_proj)
        _norm(d, p + ".norm", dim)
    if purifier:
        p = "feature_purifier"
        for n in ("norm_img", "norm_aoe", "norm_out"):
            _norm(d, f"{p}.{n}", dim)
        _mha(d, p + ".cross_attn", dim)
        _lin(d, p + ".gate.0", dim * purifier_ff_mult, dim * 2)
        _lin(d, p + ".gate.2", dim, dim * purifier_ff_mult)
    return d


CLIP_VIT_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                    image_size=224, patch_size=14, projection_dim=768)


def clip_shapes(cfg: Optional[dict] = None, prefix="image_encoder.image_encoder") -> "OrderedDict[str, Shape]":
    """transformers ``CLIPVisionModelWithProjection`` parameters (its key names) under the reference's module path
    (``ImageEncoder.image_encoder``, src/models/image_encoder.py:34-42)."""
    c = dict(cfg or CLIP_VIT_L14)
    h, inter, p = c["hidden_size"], c["intermediate_size"], c["patch_size"]
    d: "OrderedDict[str, Shape]" = OrderedDict()
    v = prefix + ".vision_model."
    d[v + "embeddings.class_embedding"] = (h,)
    d[v + "embeddings.patch_embedding.weight"] = (h, 3, p, p)
    d[v + "embeddings.position_embedding.weight"] = ((c["image_size"] // p) ** 2 + 1, h)
    _norm(d, v + "pre_layrnorm", h)
    for i in range(c["num_hidden_layers"]):
        lp = v + f"encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            _lin(d, lp + "self_attn." + n, h, h)
        _norm(d, lp + "layer_norm1", h)
        _lin(d, lp + "mlp.fc1", inter, h)
        _lin(d, lp + "mlp.fc2", h, inter)
        _norm(d, lp + "layer_norm2", h)
    _norm(d, v + "post_layernorm", h)
    d[prefix + ".visual_projection.weight"] = (c["projection_dim"], h)
    return d


# ----------------------------------------------------------------------------- init
def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFFFFFFFFFF)
    return g


def _is_norm_key(key: str) -> bool:
    leaf = key.rsplit(".", 2)[-2] if key.count(".") >= 1 else ""
    return (leaf.startswith(("norm", "layer_norm")) or leaf in ("group_norm", "conv_norm_out", "pre_layrnorm", "post_layernorm")
            or leaf in ("norm_img", "norm_aoe", "norm_out"))


def init_tensor(key: str, shape: Shape, seed: int, gates: Optional[Dict[str, Tuple[float, float]]] = None,
                aoe_delta_scale: float = 0.05) -> torch.Tensor:
    """One seeded fp32 tensor.  Linear/conv: U(-1/sqrt(fan_in), +) like PyTorch's default;
    norm affine parameters are randomised around (1, 0) so that a kernel that drops gamma/beta
    cannot pass a parity test."""
    g = _gen(key, seed)
    if key.endswith(("anat_gate", "dis_gate")):
        from .routing import get_block_type  # local: avoids a cycle at import time
        role = get_block_type(key)
        a, d = (gates or {}).get(role, (0.5, 0.5))
        return torch.tensor(a if key.endswith("anat_gate") else d, dtype=torch.float32)
    if key == "ordinal_embedder.base":
        return torch.randn(shape, generator=g) * 0.02
    if key == "ordinal_embedder.deltas":
        # monotone init of ordinal_embedder.py:92-105: N(delta_scale, 0.02) * (1 + 0.1 i)
        t = aoe_delta_scale + 0.02 * torch.randn(shape, generator=g)
        return t * (1.0 + 0.1 * torch.arange(shape[0], dtype=torch.float32))[:, None]
    if key == "ordinal_embedder.null_embedding":
        return torch.zeros(shape)
    if key == "image_projection.latents":
        return torch.randn(shape, generator=g) * 0.02
    if key.endswith(("class_embedding", "position_embedding.weight")):
        return torch.randn(shape, generator=g) * 0.02
    if _is_norm_key(key):
        n = torch.randn(shape, generator=g) * 0.1
        return 1.0 + n if key.endswith(".weight") else n
    if key.endswith("in_proj_bias"):
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.02
    if key.endswith(".bias"):
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.05
    fan_in = int(math.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    bound = 1.0 / math.sqrt(fan_in)
    return (torch.rand(shape, generator=g) * 2 - 1) * bound


def init_state_dict(shapes: "Dict[str, Shape]", seed: int = 0,
                    gates: Optional[Dict[str, Tuple[float, float]]] = None,
                    warm_start_dis: bool = True, keys: Optional[Iterable[str]] = None,
                    aoe_delta_scale: float = 0.05) -> "OrderedDict[str, torch.Tensor]":
    """Seeded fp32 CPU state dict.  ``warm_start_dis`` copies to_k/to_v into to_k_dis/to_v_dis as
    ``set_split_injection_processors`` does at construction (routing_gates.py:308-314); tests turn
    it off so the disease and anatomy projections differ."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k in (keys if keys is not None else shapes.keys()):
        out[k] = init_tensor(k, shapes[k], seed, gates, aoe_delta_scale)
    if warm_start_dis:
        for k in list(out.keys()):
            if k.endswith("processor.to_k_dis.weight"):
                out[k] = out[k.replace("processor.to_k_dis", "to_k")].clone()
            elif k.endswith("processor.to_v_dis.weight"):
                out[k] = out[k.replace("processor.to_v_dis", "to_v")].clone()
    return out


def count_params(shapes: "Dict[str, Shape]", exclude=("anat_gate", "dis_gate")) -> int:
    return sum(int(math.prod(s)) for k, s in shapes.items() if not k.endswith(exclude))
