"""Oracle (test infrastructure): the FORWARD half of ``DiffusionModuleWithIP.training_step``
(src/models/diffusion_module_ip.py:392-462) as pure functions over the flat state dict, with every random draw
injected: ``_q_sample`` (:299-303), ``_min_snr_weight`` (:305-313), training-time ``_prepare_conditioning`` (:334-381,
source == target, zero delta segment), CFG image-token dropout (:433-438) and the weighted eps-MSE (:440-443).
Plus ``LinearWarmupCosineAnnealingLR.get_lr`` (src/models/lr_scheduler.py:42-64).  Only tests import this."""
from __future__ import annotations

import math

import torch

from . import conditioning as C
from .sampler import OracleCfg, image_embeds, noise_schedule
from .sd_unet import unet_forward
from .sd_vae import vae_encode_sample


def q_sample(ac, x0, t, noise):
    a = ac[t].view(-1, 1, 1, 1)
    return torch.sqrt(a) * x0 + torch.sqrt(1.0 - a) * noise


def min_snr_weight(snr_values, t, gamma=1.0, enabled=True):
    if not enabled:
        return torch.ones_like(t, dtype=torch.float32)
    snr = snr_values[t]
    return torch.minimum(snr, torch.tensor(gamma)) / (snr + 1e-8)


def training_loss(sd, cfg: OracleCfg, images, labels, clip_features, t, noise, latent_noise, drop_mask,
                  min_snr_gamma=1.0, aoe_noise=None):
    _, ac, _, snr = noise_schedule(cfg)
    latents = vae_encode_sample(sd, images, latent_noise) * cfg.latent_scale
    noisy = q_sample(ac, latents, t, noise)
    aoe = C.aoe_forward(sd, labels, cfg.num_aoe_tokens) if aoe_noise is None else None
    if aoe is None:
        raise NotImplementedError("AOE training noise is drawn inside the embedder; tests run with noise_std = 0")
    img = image_embeds(sd, cfg, clip_features)
    if cfg.use_feature_purifier:
        img = C.feature_purifier(sd, img, aoe, cfg.purifier_num_heads)
    img = torch.where(drop_mask.view(-1, 1, 1).expand_as(img), torch.zeros_like(img), img)
    cond = torch.cat([aoe, img, torch.zeros_like(aoe)] if cfg.use_routing_gates else [aoe, img], dim=1)
    pred = unet_forward(sd, noisy, t, cond, use_routing_gates=cfg.use_routing_gates, delta_scale=0.0)
    base = ((pred - noise) ** 2).mean(dim=(1, 2, 3))
    return (min_snr_weight(snr, t, min_snr_gamma) * base).mean(), base


def warmup_cosine_lr(epoch, base_lrs, warmup_epochs, max_epochs, warmup_start_lr, eta_min=0.0):
    warmup_epochs, max_epochs = max(0, int(warmup_epochs)), max(1, int(max_epochs))
    if warmup_epochs > 0 and epoch < warmup_epochs:
        return [warmup_start_lr + (b - warmup_start_lr) * (epoch / float(warmup_epochs)) for b in base_lrs]
    prog = min((epoch - warmup_epochs) / float(max(1, max_epochs - warmup_epochs)), 1.0)
    return [eta_min + (b - eta_min) * 0.5 * (1.0 + math.cos(math.pi * prog)) for b in base_lrs]
