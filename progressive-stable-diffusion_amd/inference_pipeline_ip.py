"""DDIM inference for MES progression with IP-Adapter conditioning — HIP engine behind the
function surface of src/pipelines/inference/inference_pipeline_ip.py (same names, argument
meaning, defaults and error behaviour; file:line of each counterpart in the docstrings).
"""
from __future__ import annotations

import argparse
import time
from pathlib import Path
from typing import Optional, Tuple

import torch
from torch import Tensor

from .config import load_config as _load_config  # noqa: F401  (:178-181)
from .diffusion_module_ip import DiffusionModuleWithIP


def _load_leace_projection(leace_path: Path, device: torch.device) -> dict:
    """(:24-33)"""
    data = torch.load(leace_path, map_location="cpu", weights_only=True)
    data["P_null"] = data["P_null"].to(device)
    data["mu"] = data["mu"].to(device)
    return data


def _apply_leace(image_embeds: Tensor, leace: dict) -> Tensor:
    """(flat - mu) @ P_null^T + mu over the flattened (T*D) token vector (:36-57)."""
    b, t, d = image_embeds.shape
    p_null = leace["P_null"].to(device=image_embeds.device, dtype=image_embeds.dtype)
    mu = leace["mu"].to(device=image_embeds.device, dtype=image_embeds.dtype)
    flat = (image_embeds.reshape(b, t * d) - mu[None, :]) @ p_null.T
    return (flat + mu[None, :]).reshape(b, t, d)


def _parse_args(argv=None) -> argparse.Namespace:
    """Same flags and defaults as the reference CLI (:60-162)."""
    p = argparse.ArgumentParser(description="Generate MES progression with patient-specific anatomical structure.")
    p.add_argument("--checkpoint", type=Path, required=True)
    p.add_argument("--config", type=Path, default=Path("configs/train_ip.yaml"))
    p.add_argument("--structure-image", type=Path, required=True)
    p.add_argument("--output-dir", type=Path, default=Path("outputs/inference_ip"))
    p.add_argument("--mes-steps", type=int, default=13)
    p.add_argument("--sampling-steps", type=int, default=50)
    p.add_argument("--device", type=str, default="auto")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--image-scale", type=float, default=1.0)
    p.add_argument("--eta", type=float, default=0.0)
    p.add_argument("--zero-image", action="store_true", default=False)
    p.add_argument("--leace", type=Path, default=None)
    p.add_argument("--source-label", type=float, default=None)
    p.add_argument("--steer-scale", type=float, default=0.0)
    p.add_argument("--guidance-scale", type=float, default=None)
    return p.parse_args(argv)


def _resolve_device(device_str: str) -> torch.device:
    if device_str != "auto":
        return torch.device(device_str)
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def _set_seed(seed: int) -> None:
    """(:171-175)"""
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def _build_labels(num_steps: int, start: float = 0.0, end: float = 3.0,
                  device: Optional[torch.device] = None) -> Tensor:
    """(:184-195)"""
    device = device or torch.device("cpu")
    if num_steps <= 0:
        raise ValueError("`mes_steps` must be a positive integer.")
    return torch.linspace(start, end, steps=num_steps, device=device, dtype=torch.float32)


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _clip_preprocess(display: Tensor) -> Tensor:
    """CLIPImageProcessor defaults for ViT-L/14: bicubic resize of the short side to 224, centre
    crop 224, normalise.  The reference instantiates the processor from the hub on every call
    (:223); its constants are restated so the path works offline."""
    import torch.nn.functional as F
    _, h, w = display.shape
    s = 224.0 / min(h, w)
    nh, nw = max(224, round(h * s)), max(224, round(w * s))
    x = F.interpolate(display[None], size=(nh, nw), mode="bicubic", align_corners=False, antialias=True)
    top, left = (nh - 224) // 2, (nw - 224) // 2
    x = x[:, :, top:top + 224, left:left + 224].clamp(0, 1)
    mean = torch.tensor(CLIP_MEAN)[None, :, None, None]
    std = torch.tensor(CLIP_STD)[None, :, None, None]
    return (x - mean) / std


def _load_and_preprocess_structure_image(image_path: Path, target_size: int,
                                         device: torch.device) -> Tuple[Tensor, Tensor]:
    """(:198-229) -> (CLIP pixel_values (1,3,224,224) on device, display tensor (3,H,W) in [0,1])."""
    import numpy as np
    from PIL import Image
    pil = Image.open(image_path).convert("RGB").resize((target_size, target_size), Image.BILINEAR)
    display = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).float() / 255.0
    return _clip_preprocess(display).to(device), display


def _prepare_conditioning(module: DiffusionModuleWithIP, target_labels: Tensor, source_labels: Tensor,
                          structure_image: Tensor, image_scale: float = 1.0,
                          leace: Optional[dict] = None, zero_aoe: bool = False) -> Tensor:
    """(:232-308) 3-segment [Source_AOE | E_clean | Delta_AOE] with routing gates, else [AOE | Image]."""
    batch_size = target_labels.shape[0]
    use_routing_gates = getattr(module.diff_cfg, "use_routing_gates", True)

    target_aoe = module.ordinal_embedder(target_labels, is_training=False)
    if target_aoe.dim() == 2:
        target_aoe = target_aoe.unsqueeze(1)
    if zero_aoe:
        target_aoe = module.ordinal_embedder.get_negative_embedding(target_labels, is_training=False)
        if target_aoe.dim() == 2:
            target_aoe = target_aoe.unsqueeze(1)
    source_aoe = module.ordinal_embedder(source_labels, is_training=False)
    if source_aoe.dim() == 2:
        source_aoe = source_aoe.unsqueeze(1)

    image_embeds = module._get_image_embeds(structure_image.expand(batch_size, -1, -1, -1))
    if leace is not None:
        image_embeds = _apply_leace(image_embeds, leace)
    if module.feature_purifier is not None:
        image_embeds = module.feature_purifier(image_embeds, source_aoe)
    if image_scale != 1.0:
        image_embeds = image_embeds * image_scale

    if use_routing_gates:
        delta = module.ordinal_embedder.get_ordinal_delta_embedding(source_labels, target_labels)
        if delta.dim() == 2:
            delta = delta.unsqueeze(1)
        return torch.cat([source_aoe, image_embeds, delta], dim=1)
    return torch.cat([target_aoe, image_embeds], dim=1)


def _set_delta_scale_on_processors(module: DiffusionModuleWithIP, delta_scale: float) -> None:
    """(:311-318)"""
    for _name, mod in module.unet.unet.named_modules():
        if hasattr(mod, "processor") and hasattr(mod.processor, "delta_scale"):
            mod.processor.delta_scale = delta_scale


def _ddim_sample_ip(module: DiffusionModuleWithIP, target_labels: Tensor, source_labels: Tensor,
                    structure_image: Tensor, sampling_steps: int, device: torch.device,
                    eta: float = 0.0, image_scale: float = 1.0, leace: Optional[dict] = None,
                    steer_scale: float = 0.0, guidance_scale: float = 1.0, *,
                    latents: Optional[Tensor] = None, use_graph: bool = True,
                    trace: Optional[list] = None, step_noise: Optional[Tensor] = None) -> Tensor:
    """(:321-470).  Keyword-only extras: ``latents`` injects the initial noise (B,4,S,S) instead of
    drawing it on the device (CPU and device RNG streams differ — parity tests need this);
    ``use_graph`` / ``trace`` select eager execution and per-step capture of (eps, latents);
    ``step_noise`` (steps-1,B,4,S,S) injects the per-step noise of the eta > 0 branch (:462-466)."""
    use_routing_gates = getattr(module.diff_cfg, "use_routing_gates", True)
    do_cfg = (not use_routing_gates) and (guidance_scale != 1.0)
    num_samples = target_labels.shape[0]
    height = module.cfg.dataset.image_size
    T = module.diff_cfg.num_train_timesteps
    if sampling_steps > T:
        raise ValueError(f"sampling_steps={sampling_steps} must be <= num_train_timesteps={T}")

    if latents is None:   # same noise for all MES levels (:377-385)
        single = torch.randn(1, module.cfg.model.latent_channels, height // 8, height // 8,
                             device=device, dtype=torch.float32)
        latents = single.repeat(num_samples, 1, 1, 1)
    else:
        latents = latents.to(device=device, dtype=torch.float32)
    side = latents.shape[-1]

    # (:389-395) the reference builds this grid on the device; the integers are the same on the CPU (asserted on the GPU
    # in tests/test_gpu_parity.py), where the engine's host-side step tables need them without a device read
    timesteps = torch.linspace(T - 1, 0, steps=sampling_steps, dtype=torch.long)
    embed_cond = _prepare_conditioning(module, target_labels, source_labels, structure_image,
                                       image_scale=image_scale, leace=leace)
    embed_uncond = None
    if do_cfg:
        embed_uncond = _prepare_conditioning(module, target_labels, source_labels, structure_image,
                                             image_scale=image_scale, leace=leace, zero_aoe=True)
    _set_delta_scale_on_processors(module, steer_scale)

    if eta != 0.0:
        return _ddim_stochastic(module, latents, timesteps, embed_cond, embed_uncond, guidance_scale, eta,
                                step_noise=step_noise)

    loop = module.ddim_loop(num_samples, side)
    plan, be = loop.u, loop.be
    be.wait_current()
    plan.set_cond(embed_cond, 0)
    if do_cfg:
        plan.set_cond(embed_uncond, 1)
    return loop.sample(latents, timesteps, module.alphas_cumprod, float(steer_scale) if use_routing_gates else 0.0,
                       do_cfg, float(guidance_scale), use_graph=use_graph, trace=trace)


def _ddim_sample_batched(module: DiffusionModuleWithIP, target_labels: Tensor, source_labels: Tensor,
                         structure_images: Tensor, sampling_steps: int, device: torch.device,
                         eta: float = 0.0, image_scale: float = 1.0, steer_scale: float = 0.0,
                         guidance_scale: float = 1.0, *, latents: Optional[Tensor] = None,
                         use_graph: bool = True) -> Tensor:
    """The sampler body as the reference's data-augmentation / evaluation pipelines copy it
    (src/pipelines/inference/inference_pipeline_ip_data_augment.py:211-297,
    src/pipelines/evaluation/evaluation_pipeline.py:472-565): an arbitrary batch of (source, target)
    pairs, ONE structure image per sample (B,3,224,224) and independent initial noise per sample.
    Same engine path as ``_ddim_sample_ip``."""
    if structure_images.shape[0] != target_labels.shape[0]:
        raise ValueError(f"structure_images batch {structure_images.shape[0]} != labels batch {target_labels.shape[0]}")
    if latents is None:
        side = module.cfg.dataset.image_size // 8
        latents = torch.randn(target_labels.shape[0], module.cfg.model.latent_channels, side, side,
                              device=device, dtype=torch.float32)
    return _ddim_sample_ip(module, target_labels, source_labels, structure_images, sampling_steps, device,
                           eta=eta, image_scale=image_scale, steer_scale=steer_scale,
                           guidance_scale=guidance_scale, latents=latents, use_graph=use_graph)


def _decode_latents(module: DiffusionModuleWithIP, latents: Tensor) -> Tensor:
    """(inference_pipeline_ip_data_augment.py:300-310) decode -> [0,1] RGB fp32 on the CPU for saving."""
    return _latents_to_images(module, latents).float().cpu()


def _ddim_stochastic(module, latents, timesteps, cond, uncond, guidance_scale, eta, step_noise=None):
    """eta > 0 (:457-468): per-step engine calls, update in torch (device RNG noise unless injected)."""
    ac = module.alphas_cumprod
    n = timesteps.shape[0]
    for i in range(n):
        t_int = int(timesteps[i])
        t = torch.full((latents.shape[0],), t_int, dtype=torch.long, device=latents.device)
        eps = module(latents, t, cond)
        if uncond is not None:
            eps_u = module(latents, t, uncond)
            eps = eps_u + guidance_scale * (eps - eps_u)
        a_t = ac[t_int].to(latents.dtype)
        x0 = ((latents - torch.sqrt(1.0 - a_t) * eps) / torch.sqrt(a_t)).clamp(-4.0, 4.0)
        if i == n - 1:
            return x0
        a_p = ac[int(timesteps[i + 1])].to(latents.dtype)
        sigma = eta * torch.sqrt((1 - a_p) / (1 - a_t) * (1 - a_t / a_p))
        noise = (torch.randn_like(latents) if step_noise is None
                 else step_noise[i].to(device=latents.device, dtype=latents.dtype))
        latents = torch.sqrt(a_p) * x0 + torch.sqrt(1 - a_p - sigma ** 2) * eps + sigma * noise
    return latents


def _latents_to_images(module: DiffusionModuleWithIP, latents: Tensor) -> Tensor:
    """(:473-486) decode with the frozen VAE and map to [0, 1] RGB."""
    with torch.no_grad():
        decoded = module.vae.decode(latents / module.diff_cfg.latent_scale)
    images = decoded.sample if hasattr(decoded, "sample") else decoded
    images = images.clamp(-1.0, 1.0)
    images = (images + 1.0) / 2.0
    return images.clamp(0.0, 1.0)


def _save_sequence(images: Tensor, labels: Tensor, output_dir: Path,
                   structure_image: Optional[Tensor] = None) -> None:
    """(:489-510) ``mes_{label:.2f}_{idx:02d}.png`` + ``structure_reference.png``."""
    from PIL import Image
    output_dir.mkdir(parents=True, exist_ok=True)
    images = images.cpu()

    def to_pil(t):
        return Image.fromarray(t.permute(1, 2, 0).mul(255).to(torch.uint8).numpy())

    if structure_image is not None:
        to_pil(structure_image).save(output_dir / "structure_reference.png")
    for idx, (image, label) in enumerate(zip(images, labels)):
        to_pil(image).save(output_dir / f"mes_{label.item():.2f}_{idx:02d}.png")


def _create_progression_grid(images: Tensor, labels: Tensor, structure_image: Optional[Tensor] = None,
                             output_path: Path = None):
    """(:513-563) <=7 columns, 4 px padding, optional centred structure row on top."""
    from PIL import Image
    images = images.cpu()
    n = len(images)
    ncols = min(n, 7)
    nrows = (n + ncols - 1) // ncols + (1 if structure_image is not None else 0)
    ih, iw = images.shape[2], images.shape[3]
    pad = 4
    gw, gh = ncols * (iw + pad) + pad, nrows * (ih + pad) + pad
    grid = Image.new("RGB", (gw, gh), color=(255, 255, 255))

    def to_pil(t):
        return Image.fromarray(t.permute(1, 2, 0).mul(255).to(torch.uint8).numpy())

    row0 = 0
    if structure_image is not None:
        grid.paste(to_pil(structure_image).resize((iw, ih)), ((gw - iw) // 2, pad))
        row0 = 1
    for idx, image in enumerate(images):
        r, c = idx // ncols + row0, idx % ncols
        grid.paste(to_pil(image), (pad + c * (iw + pad), pad + r * (ih + pad)))
    if output_path:
        grid.save(output_path)
    return grid


def main(argv=None) -> None:
    """(:566-669)"""
    args = _parse_args(argv)
    device = _resolve_device(args.device)
    seed = args.seed if args.seed is not None else int(time.time() * 1000) % (2 ** 32)
    print(f"Using {'fixed' if args.seed is not None else 'random'} seed: {seed}")
    _set_seed(seed)
    cfg = _load_config(args.config)
    target_steps = args.mes_steps
    module = DiffusionModuleWithIP.load_from_checkpoint(str(args.checkpoint), cfg=cfg, weights_only=False,
                                                        strict=False, device=device, batch_size=target_steps)
    module = module.to(device).to(torch.float32)
    module.eval()
    structure_tensor, display_tensor = _load_and_preprocess_structure_image(
        args.structure_image, target_size=cfg.dataset.image_size, device=device)
    target_labels = _build_labels(target_steps, 0.0, float(cfg.dataset.num_classes - 1), device)
    source_value = args.source_label if args.source_label is not None else 0.0
    source_labels = torch.full_like(target_labels, source_value)

    use_routing_gates = getattr(module.diff_cfg, "use_routing_gates", True)
    guidance_scale = (args.guidance_scale if args.guidance_scale is not None
                      else getattr(module.diff_cfg, "guidance_scale", 1.0))
    if use_routing_gates:
        guidance_scale = 1.0          # (:629-630)
    effective_image_scale = 0.0 if args.zero_image else args.image_scale
    leace = _load_leace_projection(args.leace, device) if args.leace is not None else None

    with torch.no_grad():
        latents = _ddim_sample_ip(module, target_labels, source_labels, structure_tensor,
                                  args.sampling_steps, device, eta=args.eta,
                                  image_scale=effective_image_scale, leace=leace,
                                  steer_scale=args.steer_scale, guidance_scale=guidance_scale)
        images = _latents_to_images(module, latents)
        _save_sequence(images, target_labels, args.output_dir, display_tensor)
        grid_path = args.output_dir / "progression_grid.png"
        _create_progression_grid(images, target_labels, display_tensor, grid_path)
    print(f"Saved {len(target_labels)} progression images to {args.output_dir}")
    print(f"Saved progression grid to {grid_path}")


if __name__ == "__main__":
    main()
