"""On-disk weights -> the flat fp32 state dict the engine packs.

What the reference's checkpoints look like and where each rule below comes from:
  * Lightning ``.ckpt`` written with the EMA callback active: ``state_dict`` holds the AVERAGED weights
    and ``current_model_state`` the raw ones (src/callbacks/ema_callback.py:291-324); what
    ``DiffusionModuleWithIP.load_from_checkpoint(..., strict=False)`` then loads is ``state_dict``
    (src/pipelines/inference/inference_pipeline_ip.py:587-592) -> ``which="ema"`` is the default;
  * module prefixes ``unet.unet.*`` (diffusers names, the attention processors registered as sub-modules:
    ``...attn2.processor.{anat_gate,dis_gate,to_k_dis.weight,to_v_dis.weight}``), ``vae.vae.*``,
    ``image_encoder.image_encoder.*``, ``image_projection.*``, ``ordinal_embedder.*``, ``feature_purifier.*``;
  * the schedule buffers are ``persistent=False`` (src/models/diffusion_module_ip.py:180-193): never read
    from the file, always recomputed;
  * the frozen VAE may be stored in fp16 (16-mixed training): everything is widened to fp32 here and packed
    to the kernels' fp16 layouts by the engine;
  * base Stable-Diffusion weights come as diffusers-layout safetensors (``unet/diffusion_pytorch_model
    .safetensors``, ``vae/...``): keys without the module prefixes, recognised by their first component.

Only loaders that execute nothing from the file are used: ``safetensors`` and ``torch.load(weights_only=True)``.
A Lightning file whose ``hyper_parameters`` hold pickled config objects is refused by the safe loader; the
error says so and names the way out (re-save the tensors with safetensors).
"""
from __future__ import annotations

import warnings
from dataclasses import dataclass, field
from typing import Dict, List, Mapping, Optional, Tuple

import torch

NON_PERSISTENT = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "snr_values")
WRAPPER_PREFIXES = ("_orig_mod.", "module.")            # torch.compile / AveragedModel / DDP wrappers
_UNET_ROOTS = ("conv_in", "time_embedding", "down_blocks", "mid_block", "up_blocks", "conv_norm_out", "conv_out")
_VAE_ROOTS = ("encoder", "decoder", "quant_conv", "post_quant_conv")


@dataclass
class LoadReport:
    """What ``load_state_dict(strict=False)`` would have returned, plus what was done about it."""
    source: str = ""
    which: str = "ema"
    missing: List[str] = field(default_factory=list)        # expected by the module, absent from the file
    unexpected: List[str] = field(default_factory=list)     # in the file, unknown to the module
    widened: int = 0                                        # tensors stored narrower than fp32
    filled_from_seed: List[str] = field(default_factory=list)
    skipped_buffers: List[str] = field(default_factory=list)

    def summary(self) -> str:
        return (f"{self.source} [{self.which}]: {len(self.missing)} missing, {len(self.unexpected)} unexpected, "
                f"{self.widened} widened to fp32, {len(self.filled_from_seed)} filled from the seeded init")


def _strip(key: str) -> str:
    changed = True
    while changed:
        changed = False
        for p in WRAPPER_PREFIXES:
            if key.startswith(p):
                key, changed = key[len(p):], True
    # wrappers can also sit below the Lightning attribute: "unet.unet._orig_mod.conv_in.weight"
    return key.replace("._orig_mod.", ".")


def _with_module_prefix(key: str) -> str:
    """diffusers-layout file -> the reference's module attribute path."""
    root = key.split(".", 1)[0]
    if root in _UNET_ROOTS:     # (the VAE's own conv_in / mid_block / ... sit under encoder. / decoder.)
        return "unet.unet." + key
    if root in _VAE_ROOTS:
        return "vae.vae." + key
    return key


def read_raw(path: str) -> Mapping:
    """The file's top-level object through a loader that cannot execute code."""
    path = str(path)
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:      # pickled config objects (OmegaConf in hyper_parameters), custom classes, ...
        raise RuntimeError(
            f"{path}: the safe loader (torch.load(weights_only=True)) refused this file: {e}.  Checkpoints are "
            "never unpickled with code execution; export the tensors (state_dict / current_model_state) with "
            "safetensors or torch.save of a plain dict of tensors and load that.") from e


def select_state(blob: Mapping, which: str = "ema") -> Tuple[Dict[str, torch.Tensor], str]:
    """Pick the tensor dict out of a Lightning checkpoint / plain dict.  ``which``: "ema" = ``state_dict`` (the
    averaged weights when the EMA callback saved the file), "raw" = ``current_model_state`` when present."""
    if which not in ("ema", "raw"):
        raise ValueError("which must be 'ema' or 'raw'")
    if isinstance(blob, Mapping) and "state_dict" in blob and isinstance(blob["state_dict"], Mapping):
        if which == "raw" and isinstance(blob.get("current_model_state"), Mapping):
            return dict(blob["current_model_state"]), "raw"
        return dict(blob["state_dict"]), ("ema" if "current_model_state" in blob else "state_dict")
    if not isinstance(blob, Mapping):
        raise TypeError(f"checkpoint top level is {type(blob).__name__}, expected a mapping")
    return dict(blob), "flat"


def normalise(sd: Mapping[str, torch.Tensor], report: Optional[LoadReport] = None) -> Dict[str, torch.Tensor]:
    """Wrapper prefixes off, diffusers-layout keys under the module prefixes, non-persistent buffers and the
    legacy CLIP ``position_ids`` buffer dropped, every tensor widened to fp32 (gates stay 0-dim)."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if not isinstance(v, torch.Tensor):
            continue
        k = _with_module_prefix(_strip(k))
        if k in NON_PERSISTENT or k.endswith("position_ids"):
            if report is not None:
                report.skipped_buffers.append(k)
            continue
        if v.dtype != torch.float32:
            if report is not None and v.is_floating_point():
                report.widened += 1
            v = v.float()
        out[k] = v.detach().contiguous()
    return out


def reconcile(sd: Dict[str, torch.Tensor], shapes: Mapping[str, Tuple[int, ...]], *, strict: bool,
              seed: int = 0, init_kwargs: Optional[dict] = None,
              report: Optional[LoadReport] = None) -> Dict[str, torch.Tensor]:
    """Check ``sd`` against the module's inventory.  A shape mismatch always raises (as ``load_state_dict`` does
    under either strictness); missing tensors raise when ``strict`` and are otherwise filled from the seeded
    initialiser — the reference would leave its hub-pretrained / constructor values there, which do not exist
    offline — with a warning that names them."""
    from . import weights as W
    report = report if report is not None else LoadReport()
    bad = [f"{k}: file {tuple(sd[k].shape)} vs module {tuple(s)}" for k, s in shapes.items()
           if k in sd and tuple(sd[k].shape) != tuple(s)]
    if bad:
        raise RuntimeError("size mismatch for " + "; ".join(bad[:8]) + (" ..." if len(bad) > 8 else ""))
    report.missing = [k for k in shapes if k not in sd]
    report.unexpected = [k for k in sd if k not in shapes]
    if strict and (report.missing or report.unexpected):
        raise RuntimeError(f"Error(s) in loading state_dict: missing {report.missing[:5]} "
                           f"unexpected {report.unexpected[:5]} ({len(report.missing)}/{len(report.unexpected)})")
    if report.missing:
        fill = W.init_state_dict(dict(shapes), seed, keys=report.missing, warm_start_dis=False,
                                 **(init_kwargs or {}))
        # warm start of the disease projections from to_k / to_v when the file has those
        # (set_split_injection_processors, src/models/attention_processor_routing_gates.py:308-314)
        for k in report.missing:
            src = k.replace("processor.to_k_dis", "to_k").replace("processor.to_v_dis", "to_v")
            sd[k] = sd[src].clone() if (src != k and src in sd) else fill[k]
        report.filled_from_seed = list(report.missing)
        heads = sorted({k.split(".")[0] for k in report.missing})
        warnings.warn(f"checkpoint lacks {len(report.missing)} tensors (under {heads}); they are SEEDED RANDOM "
                      "(or warm-started) here, not pretrained", RuntimeWarning, stacklevel=3)
    return sd


def load_state(path: str, shapes: Optional[Mapping[str, Tuple[int, ...]]] = None, *, which: str = "ema",
               strict: bool = False, seed: int = 0,
               init_kwargs: Optional[dict] = None) -> Tuple[Dict[str, torch.Tensor], LoadReport, Mapping]:
    """File -> (flat fp32 state dict, report, raw top-level mapping)."""
    blob = read_raw(path)
    sd, kind = select_state(blob, which)
    rep = LoadReport(source=str(path), which=kind)
    sd = normalise(sd, rep)
    if shapes is not None:
        sd = reconcile(sd, shapes, strict=strict, seed=seed, init_kwargs=init_kwargs, report=rep)
    return sd, rep, blob


def merge_diffusers_files(unet_path: Optional[str] = None, vae_path: Optional[str] = None,
                          extra: Optional[Mapping[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """Base SD-1.x weights from diffusers-layout files (what ``OrdinalUNet`` / ``SDVAE`` pull from the hub,
    src/models/unet/unet.py:70-75, src/models/vae/vae.py:60-65) merged with DADD tensors from ``extra``."""
    out: Dict[str, torch.Tensor] = {}
    for p in (unet_path, vae_path):
        if p is not None:
            out.update(normalise(read_raw(p)))
    if extra:
        out.update(normalise(extra))
    return out
