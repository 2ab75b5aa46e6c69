"""YAML config with attribute access — stands in for OmegaConf (absent offline) with the schema of
the reference's ``configs/train_ip.yaml`` (read at src/pipelines/inference/inference_pipeline_ip.py:178-181
and consumed by attribute access + ``getattr(..., default)`` at src/models/diffusion_module_ip.py:86-117).
"""
from __future__ import annotations

import copy
from pathlib import Path
from typing import Any

import yaml


class AttrDict(dict):
    """dict whose keys are also attributes; missing attributes raise AttributeError so that
    ``getattr(cfg.model, "key", default)`` behaves as it does on a DictConfig."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_attr(obj: Any) -> Any:
    if isinstance(obj, dict):
        return AttrDict({k: to_attr(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [to_attr(v) for v in obj]
    return obj


DEFAULTS = {
    "model": {
        "name": "ordinal_progressive_sd_ip", "embedding_dim": 768, "conditioning_dim": 768,
        "base_channels": 320, "num_res_blocks": [2, 2, 2, 2], "attention_heads": 8,
        "cfg_drop_prob": 0.0, "latent_channels": 4,
        "pretrained_vae_path": "CompVis/stable-diffusion-v1-4",
        "pretrained_unet_path": "CompVis/stable-diffusion-v1-4",
        "image_encoder_path": "openai/clip-vit-large-patch14",
        "num_image_tokens": 16, "num_aoe_tokens": 16, "use_image_projection_plus": True,
        "use_frequency_strategy": True, "use_routing_gates": True, "use_feature_purifier": True,
        "gate_init_anatomy": [0.1, 0.9], "gate_init_disease": [0.9, 0.1],
        "purifier_num_heads": 8, "purifier_ff_mult": 2, "delta_scale": 0.0,
        "ordinal_embedder": {"type": "aoe", "num_classes": 4, "interpolation_steps": 101,
                             "aoe": {"delta_scale": 0.05}},
    },
    "dataset": {"image_size": 256, "num_classes": 4},
    "training": {"precision": "16-mixed", "seed": 42, "use_min_snr_weighting": True},
    "diffusion": {"noise_schedule": "linear", "beta_start": 0.00085, "beta_end": 0.012,
                  "num_train_timesteps": 1000, "sampling_steps": 50, "guidance_scale": 1.0,
                  "min_snr_gamma": 1.0, "ema_update_interval": 1},
}


def default_config(**overrides) -> AttrDict:
    """The shipped ``train_ip.yaml`` values for every key the sampler path reads.
    ``overrides`` are dotted: ``default_config(**{"dataset.image_size": 512})``."""
    cfg = to_attr(copy.deepcopy(DEFAULTS))
    for dotted, v in overrides.items():
        node = cfg
        parts = dotted.split(".")
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = to_attr(v)
    return cfg


def load_config(config_path) -> AttrDict:
    """``_load_config`` (inference_pipeline_ip.py:178-181): FileNotFoundError when missing."""
    config_path = Path(config_path)
    if not config_path.exists():
        raise FileNotFoundError(f"Config not found at {config_path}")
    with open(config_path) as f:
        return to_attr(yaml.safe_load(f))
