"""Oracle (test infrastructure): the DDIM sampler of the DADD / IP-Adapter inference path.

Restates, as pure functions over the flat state dict:
  * ``DiffusionModuleWithIP._build_noise_schedule`` + buffers  (src/models/diffusion_module_ip.py:274-287,180-193)
  * ``_prepare_conditioning``   (src/pipelines/inference/inference_pipeline_ip.py:232-308)
  * ``_apply_leace``            (…/inference_pipeline_ip.py:36-57)
  * ``_ddim_sample_ip``         (…/inference_pipeline_ip.py:321-470) — with the initial latents
    INJECTED (device RNG streams differ from CPU, SURVEY.md §7 "RNG parity")
  * ``_latents_to_images``      (…/inference_pipeline_ip.py:473-486)
Known answers are asserted in ``tests/test_oracle_golden.py``.
The CLIP encoder itself is third-party (``transformers``) in the reference too
(src/models/image_encoder.py:34-42); the oracle takes its output as an input.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import conditioning as C
from .sd_unet import unet_forward
from .sd_vae import vae_decode


@dataclass
class OracleCfg:
    image_size: int = 256
    latent_channels: int = 4
    num_train_timesteps: int = 1000
    beta_start: float = 0.00085
    beta_end: float = 0.012
    noise_schedule: str = "linear"
    latent_scale: float = 0.18215
    num_image_tokens: int = 16
    num_aoe_tokens: int = 16
    use_routing_gates: bool = True
    use_image_projection_plus: bool = True
    use_feature_purifier: bool = True
    use_frequency_strategy: bool = True
    purifier_num_heads: int = 8


def noise_schedule(cfg: OracleCfg):
    """betas linear in beta (NOT SD's scaled-linear); alphas_cumprod fp32 cumprod."""
    if cfg.noise_schedule != "linear":
        raise NotImplementedError("Only linear noise schedule is supported.")
    betas = torch.linspace(cfg.beta_start, cfg.beta_end, cfg.num_train_timesteps, dtype=torch.float32)
    ac = torch.cumprod(1.0 - betas, dim=0)
    prev = torch.cat([torch.ones(1, dtype=ac.dtype), ac[:-1]])
    snr = ac / (1.0 - ac + 1e-8)
    return betas, ac, prev, snr


def ddim_timesteps(T: int, steps: int) -> torch.Tensor:
    """``torch.linspace(T-1, 0, steps, dtype=long)`` — the irregular integer grid of App. C."""
    return torch.linspace(T - 1, 0, steps=steps, dtype=torch.long)


def apply_leace(image_embeds, leace):
    b, t, d = image_embeds.shape
    mu = leace["mu"].to(image_embeds.dtype)
    flat = (image_embeds.reshape(b, t * d) - mu[None]) @ leace["P_null"].to(image_embeds.dtype).T
    return (flat + mu[None]).reshape(b, t, d)


def image_embeds(sd, cfg: OracleCfg, clip_features):
    if cfg.use_image_projection_plus:
        return C.image_projection_plus(sd, clip_features)
    return C.image_projection(sd, clip_features, cfg.num_image_tokens)


def prepare_conditioning(sd, cfg: OracleCfg, target, source, clip_features, image_scale=1.0,
                         leace=None, zero_aoe=False):
    """[source_aoe | purified image * scale | delta]  or  [target_aoe(or negative) | image]."""
    bsz = target.shape[0]
    tgt = C.aoe_negative(sd, target, cfg.num_aoe_tokens) if zero_aoe \
        else C.aoe_forward(sd, target, cfg.num_aoe_tokens)
    src = C.aoe_forward(sd, source, cfg.num_aoe_tokens)
    img = image_embeds(sd, cfg, clip_features.expand(bsz, *clip_features.shape[1:]))
    if leace is not None:
        img = apply_leace(img, leace)
    if cfg.use_feature_purifier:
        img = C.feature_purifier(sd, img, src, cfg.purifier_num_heads)
    if image_scale != 1.0:
        img = img * image_scale
    if cfg.use_routing_gates:
        return torch.cat([src, img, C.aoe_delta(sd, source, target, cfg.num_aoe_tokens)], dim=1)
    return torch.cat([tgt, img], dim=1)


def ddim_update(latents, eps, ac, t_int, t_prev_int, last, eta=0.0, noise=None):
    """One DDIM update, op-for-op as inference_pipeline_ip.py:434-468."""
    a_t = ac[t_int].to(latents.dtype)
    s0, s1 = torch.sqrt(a_t), torch.sqrt(1.0 - a_t)
    x0 = ((latents - s1 * eps) / s0).clamp(-4.0, 4.0)
    if last:
        return x0
    a_p = ac[t_prev_int].to(latents.dtype)
    if eta == 0.0:
        return torch.sqrt(a_p) * x0 + torch.sqrt(1.0 - a_p) * eps
    sigma = eta * torch.sqrt((1 - a_p) / (1 - a_t) * (1 - a_t / a_p))
    return torch.sqrt(a_p) * x0 + torch.sqrt(1 - a_p - sigma ** 2) * eps + sigma * noise


def ddim_sample(sd, cfg: OracleCfg, target, source, clip_features, sampling_steps, init_latents,
                eta=0.0, image_scale=1.0, leace=None, steer_scale=0.0, guidance_scale=1.0,
                trace=None, step_noise=None):
    """Whole sampler.  ``init_latents`` (B,4,S,S) replaces the device ``randn`` (:377-385).

    ``trace``: optional list receiving (eps, latents after the step, latents before it, CFG parts) per step for
    step-wise (teacher-forced) parity tests;
    ``step_noise`` (steps-1,B,4,S,S) replaces the per-step ``randn_like`` of the eta > 0 branch (:462-466).
    """
    do_cfg = (not cfg.use_routing_gates) and (guidance_scale != 1.0)
    T = cfg.num_train_timesteps
    if sampling_steps > T:
        raise ValueError(f"sampling_steps={sampling_steps} must be <= num_train_timesteps={T}")
    _, ac, _, _ = noise_schedule(cfg)
    ts = ddim_timesteps(T, sampling_steps)
    cond = prepare_conditioning(sd, cfg, target, source, clip_features, image_scale, leace)
    uncond = prepare_conditioning(sd, cfg, target, source, clip_features, image_scale, leace,
                                  zero_aoe=True) if do_cfg else None
    x = init_latents.clone().float()
    bsz = x.shape[0]

    def unet(c):
        return unet_forward(sd, x, t, c, use_routing_gates=cfg.use_routing_gates,
                            delta_scale=steer_scale,
                            use_frequency_strategy=cfg.use_frequency_strategy)

    for i in range(sampling_steps):
        t_int = int(ts[i])
        t = torch.full((bsz,), t_int, dtype=torch.long)
        x_in, parts = x, None
        if do_cfg:
            e_c, e_u = unet(cond), unet(uncond)
            eps = e_u + guidance_scale * (e_c - e_u)
            parts = (e_c, e_u)
        else:
            eps = unet(cond)
        last = i == sampling_steps - 1
        noise = None
        if eta != 0.0 and not last:
            noise = torch.randn_like(x) if step_noise is None else step_noise[i].to(x)
        x = ddim_update(x, eps, ac, t_int, None if last else int(ts[i + 1]), last, eta, noise)
        if trace is not None:       # (eps, x_{t-1}, x_t the step started from, (eps_cond, eps_uncond) under CFG)
            trace.append((eps, x.clone(), x_in, parts))
    return x


def latents_to_images(sd, cfg: OracleCfg, latents):
    img = vae_decode(sd, latents / cfg.latent_scale).clamp(-1.0, 1.0)
    return ((img + 1.0) / 2.0).clamp(0.0, 1.0)
