"""diffusers attention-processor protocol, backed by the fused HIP cross-attention kernel.

Operator-level seam of SURVEY.md §8b: processors with the reference's call signature
``processor(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, ...)``
and state (``anat_gate``, ``dis_gate``, ``to_k_dis``, ``to_v_dis``, ``delta_scale`` / ``frequency_mode``),
so they can be installed on a real diffusers ``UNet2DConditionModel`` with ``unet.set_attn_processor``:
  * ``SplitInjectionAttentionProcessor``  — src/models/attention_processor_routing_gates.py:12-196
  * ``OrdinalIPAttnProcessor2_0``         — src/models/attention_processor_base.py:12-138
  * ``get_block_type`` / ``get_frequency_mode_for_block`` / ``set_*_processors`` — same files :199-316 / :141-216
``attn`` only needs what the reference reads from it: ``to_q``, ``to_k``, ``to_v``, ``to_out`` (callables with
``.weight``), ``heads`` and the optional-norm attributes.  The three softmaxes, the gate/lambda-weighted sum and
the K/V projections run in ``dadd_tri_xattn_f16`` / ``dadd_conv_igemm_f16``; the step-invariant K/V projection
of the conditioning tokens is cached per conditioning tensor.
"""
from __future__ import annotations

from typing import Dict, Literal, Optional, Tuple

import torch
import torch.nn as nn

from . import lib as L
from .routing import get_block_type, get_frequency_mode_for_block  # noqa: F401  (reference names)

F16 = torch.float16
_BACKENDS: Dict[str, object] = {}


def _backend(device: torch.device):
    from .backend import HipBackend
    key = str(device)
    if key not in _BACKENDS:
        _BACKENDS[key] = HipBackend(device)
    return _BACKENDS[key]


class _HipCrossAttention(nn.Module):
    """Shared machinery: q projection, cached K/V projection of the tokens, fused kernel, out projection."""

    mode = L.XATTN_SPLIT

    def _kv_weight(self, attn) -> torch.Tensor:
        raise NotImplementedError

    def _w16(self, be, name: str, w: torch.Tensor, dtype=F16) -> torch.Tensor:
        """fp16 (bias: fp32) device copy of a parameter, refreshed when the parameter is updated."""
        cache = self.__dict__.setdefault("_wcache", {})
        hit = cache.get(name)
        if hit is None or hit[0] is not w or hit[1] != w._version:
            cache[name] = hit = (w, w._version, be.to_device(w.detach(), dtype))
        return hit[2]

    def _run(self, attn, hidden_states, encoder_hidden_states, gates, lam):
        if attn.spatial_norm is not None or attn.group_norm is not None:
            raise NotImplementedError("spatial_norm / group_norm are inactive on SD-1.x attn2 and not built")
        residual = hidden_states
        input_ndim = hidden_states.ndim
        if input_ndim == 4:
            b, c, h, w = hidden_states.shape
            hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
        if encoder_hidden_states is None:
            raise NotImplementedError("self-attention stays on diffusers' AttnProcessor2_0 (attn1)")
        dev = hidden_states.device
        be = _backend(dev)
        b, n, c = hidden_states.shape
        t = encoder_hidden_states.shape[1]
        heads = attn.heads
        be.wait_current()
        x16 = be.to_device(hidden_states.detach().reshape(b, 1, n, c), F16)
        # step-invariant: K/V projections of the conditioning tokens, cached per tensor
        # (the cache holds the tensor itself, so its storage cannot be recycled under the same address)
        key = encoder_hidden_states._version
        if getattr(self, "_kv_src", None) is not encoder_hidden_states or self._kv_key != key:
            w_kv = be.to_device(self._kv_weight(attn).detach(), F16)
            cond16 = be.to_device(encoder_hidden_states.detach().reshape(b, 1, t, -1), F16)
            kv = be.empty((b, 1, t, w_kv.shape[0]), F16)
            be.igemm(cond16, w_kv, kv, flags=L.TUNE_NODMA, tile_m=64)
            self._kv, self._kv_key, self._kv_src = kv, key, encoder_hidden_states
        q = be.empty((b, 1, n, c), F16)
        be.igemm(x16, self._w16(be, "q", attn.to_q.weight), q, flags=L.TUNE_NODMA, tile_m=64)
        att = be.empty((b, n, c), F16)
        be.tri_xattn(q.view(b, n, c), self._kv.view(b, t, -1), att, gates, lam, self.mode, heads)
        out = be.empty((b, 1, n, c), F16)
        to_out = attn.to_out[0]
        bias = None if getattr(to_out, "bias", None) is None else self._w16(be, "ob", to_out.bias, torch.float32)
        be.igemm(att.view(b, 1, n, c), self._w16(be, "o", to_out.weight), out, bias=bias,
                 flags=(L.EPI_BIAS if bias is not None else 0) | L.TUNE_NODMA, tile_m=64)
        be.release_to_current()
        hidden_states = attn.to_out[1](out.view(b, n, c).to(residual.dtype))
        if input_ndim == 4:
            hidden_states = hidden_states.transpose(-1, -2).reshape(b, c, h, w)
        if attn.residual_connection:
            hidden_states = hidden_states + residual
        return hidden_states / attn.rescale_output_factor


class SplitInjectionAttentionProcessor(_HipCrossAttention):
    """Triple-pathway cross-attention (anatomy / disease / delta) with fixed per-block gates."""

    mode = L.XATTN_SPLIT

    def __init__(self, hidden_size: int, cross_attention_dim: Optional[int] = None, num_image_tokens: int = 16,
                 num_aoe_tokens: int = 16, num_delta_tokens: int = 16,
                 block_type: Literal["anatomy", "disease", "both"] = "both",
                 anat_gate_init: Optional[float] = None, dis_gate_init: Optional[float] = None,
                 delta_scale: float = 0.0) -> None:
        super().__init__()
        if (num_image_tokens, num_aoe_tokens, num_delta_tokens) != (16, 16, 16):
            raise NotImplementedError("the fused kernel is built for 16 tokens per pathway")
        self.hidden_size, self.cross_attention_dim = hidden_size, cross_attention_dim
        self.num_image_tokens, self.num_aoe_tokens, self.num_delta_tokens = 16, 16, 16
        self.block_type, self.delta_scale = block_type, delta_scale
        self.register_buffer("anat_gate", torch.tensor(0.5 if anat_gate_init is None else anat_gate_init))
        self.register_buffer("dis_gate", torch.tensor(0.5 if dis_gate_init is None else dis_gate_init))
        d_in = cross_attention_dim or hidden_size
        self.to_k_dis = nn.Linear(d_in, hidden_size, bias=False)
        self.to_v_dis = nn.Linear(d_in, hidden_size, bias=False)

    def _kv_weight(self, attn):
        return torch.cat([attn.to_k.weight, attn.to_v.weight, self.to_k_dis.weight, self.to_v_dis.weight])

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 *args, **kwargs):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is not used on this path (routing_gates.py:110-116)")
        if encoder_hidden_states is not None and encoder_hidden_states.shape[1] != 48:
            raise ValueError("expected [Source_AOE(16) | E_clean(16) | Delta_AOE(16)] = 48 conditioning tokens")
        be = _backend(hidden_states.device)
        gates = be.to_device(torch.stack([self.anat_gate, self.dis_gate]).float())
        return self._run(attn, hidden_states, encoder_hidden_states, gates, float(self.delta_scale))


class OrdinalIPAttnProcessor2_0(_HipCrossAttention):
    """Baseline 2-segment [AOE | Image] cross-attention, one joint softmax."""

    mode = L.XATTN_BASELINE

    def __init__(self, hidden_size: int, cross_attention_dim: Optional[int] = None, num_image_tokens: int = 16,
                 num_aoe_tokens: int = 16,
                 frequency_mode: Literal["both", "aoe_dominant", "image_dominant"] = "both") -> None:
        super().__init__()
        self.hidden_size, self.cross_attention_dim = hidden_size, cross_attention_dim
        self.num_image_tokens, self.num_aoe_tokens = num_image_tokens, num_aoe_tokens
        self.frequency_mode = frequency_mode
        self.scale_aoe = self.scale_ip = 1.0        # base.py:29-37: every mode scales by 1 (a no-op)

    def _kv_weight(self, attn):
        return torch.cat([attn.to_k.weight, attn.to_v.weight])

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None,
                 *args, **kwargs):
        if encoder_hidden_states is not None and attn.norm_cross:
            raise NotImplementedError("Cross-attention with separate encoder hidden states is not implemented "
                                      "in OrdinalIPAttnProcessor2_0.")
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is not used on this path")
        if encoder_hidden_states is not None and encoder_hidden_states.shape[1] != 32:
            raise ValueError("expected [AOE(16) | Image(16)] = 32 conditioning tokens")
        return self._run(attn, hidden_states, encoder_hidden_states, None, 0.0)


def _hidden_size(unet, name: str) -> int:
    ch = unet.config.block_out_channels
    if name.startswith("mid_block"):
        return ch[-1]
    if name.startswith("up_blocks"):
        return list(reversed(ch))[int(name[len("up_blocks.")])]
    if name.startswith("down_blocks"):
        return ch[int(name[len("down_blocks.")])]
    return ch[0]


def set_split_injection_processors(unet, num_image_tokens: int = 16, num_aoe_tokens: int = 16,
                                   num_delta_tokens: int = 16, use_frequency_strategy: bool = True,
                                   delta_scale: float = 0.0,
                                   gate_inits: Optional[Dict[str, Tuple[float, float]]] = None) -> dict:
    """Install the HIP triple-pathway processor on every attn2 of a diffusers UNet and warm-start the disease
    K/V from the text K/V (routing_gates.py:233-316); attn1 keeps whatever processor it has."""
    gate_inits = gate_inits or {"anatomy": (0.5, 0.5), "disease": (0.5, 0.5), "both": (0.5, 0.5)}
    procs = dict(unet.attn_processors)
    for name in list(procs.keys()):
        if name.endswith("attn1.processor"):
            continue
        role = get_block_type(name) if use_frequency_strategy else "both"
        a, d = gate_inits.get(role, (0.5, 0.5))
        procs[name] = SplitInjectionAttentionProcessor(
            _hidden_size(unet, name), unet.config.cross_attention_dim, num_image_tokens, num_aoe_tokens,
            num_delta_tokens, role, a, d, delta_scale)
    unet.set_attn_processor(procs)
    for _n, mod in unet.named_modules():
        if hasattr(mod, "processor") and isinstance(mod.processor, SplitInjectionAttentionProcessor):
            with torch.no_grad():
                mod.processor.to_k_dis.weight.copy_(mod.to_k.weight)
                mod.processor.to_v_dis.weight.copy_(mod.to_v.weight)
    return procs


def set_ordinal_ip_attention_processors(unet, num_image_tokens: int = 16, num_aoe_tokens: int = 16,
                                        use_frequency_strategy: bool = True) -> dict:
    """base.py:170-216."""
    procs = dict(unet.attn_processors)
    for name in list(procs.keys()):
        if name.endswith("attn1.processor"):
            continue
        mode = get_frequency_mode_for_block(name) if use_frequency_strategy else "both"
        procs[name] = OrdinalIPAttnProcessor2_0(_hidden_size(unet, name), unet.config.cross_attention_dim,
                                                num_image_tokens, num_aoe_tokens, mode)
    unet.set_attn_processor(procs)
    return procs
