"""``DiffusionModuleWithIP`` with the attribute protocol the reference pipelines use, backed by the
HIP engine.  Mirrors src/models/diffusion_module_ip.py (config block :33-64, construction :81-201,
schedule :274-287, ``_get_image_embeds`` :315-332, ``forward`` :383-390) and the two thin wrappers
src/models/unet/unet.py:52-146 and src/models/vae/vae.py:32-112.

What callers touch and where it lives here:
  module(latents, t, cond)           -> UNetPlan.forward              (eps, fp32 NCHW)
  module.diff_cfg / module.cfg       -> DiffusionIPConfig / AttrDict
  module.alphas_cumprod[...]         -> fp32 tensor, linear-beta schedule
  module.ordinal_embedder(...)       -> conditioning.AdditiveOrdinalEmbedder
  module._get_image_embeds(x)        -> CLIP tower + (Plus) projection
  module.feature_purifier            -> conditioning.FeaturePurifier | None
  module.unet.unet.named_modules()   -> 16 attn2 facades whose .processor.delta_scale is settable
  module.vae.decode(x).sample        -> VaeDecoderPlan
There is no CPU path: constructing the module needs the HIP library and a ROCm device.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any, Dict, Iterator, Optional, Tuple

import torch

from . import weights as W
from .conditioning import (AdditiveOrdinalEmbedder, FeaturePurifier, ImageEncoder, ImageProjection,
                           ImageProjectionPlus)
from .engine import DdimLoop, UNetPlan, VaeDecoderPlan
from .routing import get_block_type, get_frequency_mode_for_block


@dataclass
class DiffusionIPConfig:
    num_train_timesteps: int
    beta_start: float
    beta_end: float
    noise_schedule: str = "linear"
    sampling_steps: int = 50
    guidance_scale: float = 2.0
    min_snr_gamma: float = 1.0
    ema_update_interval: int = 10
    latent_scale: float = 0.18215
    input_perturbation: float = 0.0
    image_encoder_path: str = "openai/clip-vit-base-patch16"
    num_image_tokens: int = 16
    num_aoe_tokens: int = 16
    use_frequency_strategy: bool = True
    use_image_projection_plus: bool = False
    use_feature_purifier: bool = True
    purifier_num_heads: int = 8
    purifier_ff_mult: int = 2
    delta_scale: float = 0.0
    use_routing_gates: bool = True
    gate_init_anatomy: tuple = (0.5, 0.5)
    gate_init_disease: tuple = (0.5, 0.5)


def diff_cfg_from(cfg: Any) -> DiffusionIPConfig:
    g = getattr
    m, d, t = cfg.model, cfg.diffusion, getattr(cfg, "training", SimpleNamespace())
    return DiffusionIPConfig(
        num_train_timesteps=d.num_train_timesteps, beta_start=d.beta_start, beta_end=d.beta_end,
        noise_schedule=d.noise_schedule, sampling_steps=g(d, "sampling_steps", 50),
        guidance_scale=g(d, "guidance_scale", 2.0), min_snr_gamma=g(d, "min_snr_gamma", 1.0),
        ema_update_interval=g(d, "ema_update_interval", 10), latent_scale=g(d, "latent_scale", 0.18215),
        input_perturbation=g(t, "input_perturbation", 0.0),
        image_encoder_path=g(m, "image_encoder_path", "openai/clip-vit-base-patch16"),
        num_image_tokens=g(m, "num_image_tokens", 16), num_aoe_tokens=g(m, "num_aoe_tokens", 16),
        use_frequency_strategy=g(m, "use_frequency_strategy", True),
        use_image_projection_plus=g(m, "use_image_projection_plus", False),
        use_feature_purifier=g(m, "use_feature_purifier", True),
        purifier_num_heads=g(m, "purifier_num_heads", 8), purifier_ff_mult=g(m, "purifier_ff_mult", 2),
        delta_scale=g(m, "delta_scale", 0.0), use_routing_gates=g(m, "use_routing_gates", True),
        gate_init_anatomy=tuple(g(m, "gate_init_anatomy", [0.5, 0.5])),
        gate_init_disease=tuple(g(m, "gate_init_disease", [0.5, 0.5])))


def build_noise_schedule(dc: DiffusionIPConfig) -> Tuple[torch.Tensor, torch.Tensor]:
    """Linear-in-beta schedule (NOT SD's scaled-linear), fp32 cumprod (:274-287)."""
    if dc.noise_schedule != "linear":
        raise NotImplementedError("Only linear noise schedule is supported.")
    betas = torch.linspace(dc.beta_start, dc.beta_end, dc.num_train_timesteps, dtype=torch.float32)
    return betas, torch.cumprod(1.0 - betas, dim=0)


# ---------------------------------------------------------------------------------------------
class _Processor:
    """State holder of one attn2 site (what ``_set_delta_scale_on_processors`` walks)."""

    def __init__(self, name: str, routing: bool, gates: Optional[torch.Tensor]):
        self.name = name
        if routing:
            self.delta_scale = 0.0
            self.block_type = get_block_type(name)
            self.anat_gate, self.dis_gate = (gates[0], gates[1]) if gates is not None else (None, None)
        else:
            self.frequency_mode = get_frequency_mode_for_block(name)


class _AttnSite:
    def __init__(self, proc: _Processor):
        self.processor = proc


class _InnerUNet:
    """Stands where diffusers' ``UNet2DConditionModel`` stands (``module.unet.unet``)."""

    def __init__(self, plan: UNetPlan, routing: bool):
        self._plan = plan
        self.config = SimpleNamespace(in_channels=4, out_channels=4, cross_attention_dim=768,
                                      block_out_channels=[320, 640, 1280, 1280], sample_size=64)
        self._sites = {}
        for site, _ in plan.sites:
            n = f"{site}.transformer_blocks.0.attn2"
            self._sites[n] = _AttnSite(_Processor(n, routing, plan.gates.get(site)))

    def named_modules(self):
        return iter(self._sites.items())

    @property
    def attn_processors(self):
        return {n + ".processor": s.processor for n, s in self._sites.items()}

    def delta_scale(self) -> float:
        vals = {float(getattr(s.processor, "delta_scale", 0.0)) for s in self._sites.values()}
        if len(vals) > 1:
            raise ValueError(f"attention processors disagree on delta_scale: {sorted(vals)}")
        return vals.pop()

    def __call__(self, sample, timestep, encoder_hidden_states=None, **_):
        """``UNet2DConditionModel.forward`` as OrdinalUNet.forward uses it (src/models/unet/unet.py:140-144):
        ``unet(sample, timestep, encoder_hidden_states=cond).sample``."""
        eps = self._plan.forward(sample.float(), timestep, encoder_hidden_states, lam=self.delta_scale())
        return SimpleNamespace(sample=eps)


class OrdinalUNet:
    """src/models/unet/unet.py:52-146: argument normalisation + sanity checks, then the engine."""

    def __init__(self, plan: UNetPlan, routing: bool, conditioning_dim=768, in_channels=4, out_channels=4):
        self.compiled_handle = None
        self.unet = _InnerUNet(plan, routing)
        c = self.unet.config
        if c.in_channels != in_channels:
            raise ValueError(f"UNet in_channels mismatch: {c.in_channels} (from weights) vs {in_channels} (config).")
        if c.out_channels != out_channels:
            raise ValueError(f"UNet out_channels mismatch: {c.out_channels} (from weights) vs {out_channels} (config).")
        if c.cross_attention_dim != conditioning_dim:
            raise ValueError(f"UNet cross_attention_dim mismatch: {c.cross_attention_dim} (from weights) "
                             f"vs {conditioning_dim} (config).")
        self._plan = plan
        self._cond_ref, self._cond_ver, self._cond_gen = None, -1, -1

    # ``module.unet.unet = torch.compile(module.unet.unet, mode="reduce-overhead")`` (the reference's --compile switch,
    # src/pipelines/inference/inference_pipeline_ip_data_augment.py:398-400): the assignment is accepted and the wrapper kept
    # (``compiled_handle``), but the step keeps running on the engine - its captured hipGraph IS the launch-overhead removal
    # that switch asks for, and there is nothing for a tracing compiler to trace.
    @property
    def unet(self):
        return self._inner

    @unet.setter
    def unet(self, value):
        if isinstance(value, _InnerUNet):
            self._inner = value
        else:
            self.compiled_handle = value

    def parameters(self) -> Iterator[torch.Tensor]:
        return iter(self._plan.keep)

    def __call__(self, latents, timesteps, cond_embed):
        cond_in = cond_embed                  # identity of the caller's tensor (views below are new objects)
        if cond_embed.ndim == 2:
            cond_embed = cond_embed.unsqueeze(1)
        elif cond_embed.ndim != 3:
            raise ValueError(f"cond_embed must have shape (B, D) or (B, seq_len, D), got {cond_embed.shape}")
        if timesteps.ndim == 0:
            timesteps = timesteps[None]
        elif timesteps.ndim > 1:
            timesteps = timesteps.view(-1)
        plan = self._plan
        be = plan.be
        # dtype / device / shape normalisation runs on torch's CURRENT stream, i.e. before wait_current()
        # orders the backend stream behind it (a scalar, CPU or non-fp32 input is produced here)
        timesteps = timesteps.to(device=latents.device, dtype=torch.long).reshape(-1)
        if timesteps.shape[0] == 1:
            timesteps = timesteps.expand(plan.B)
        timesteps = timesteps.contiguous()
        latents = latents.float().contiguous()
        # Conditioning tensors are step-invariant: re-project only when the caller passes a new one.  The cache
        # HOLDS the tensor (an address can be recycled by the allocator) and is tied to the plan's conditioning
        # generation, which every set_cond() — e.g. a sampler run on the same plan — advances.
        fresh = not (cond_in is self._cond_ref and cond_in._version == self._cond_ver
                     and plan.cond_gen == self._cond_gen)
        if fresh:
            cond_embed = cond_embed.to(device=latents.device)
        be.wait_current()
        eps = plan.forward(latents, timesteps, cond_embed if fresh else None, lam=self.unet.delta_scale())
        if fresh:
            self._cond_ref, self._cond_ver, self._cond_gen = cond_in, cond_in._version, plan.cond_gen
        be.release_to_current()
        return eps

    forward = __call__


class SDVAE:
    """src/models/vae/vae.py:32-112 — decode() through the HIP decoder plan."""

    def __init__(self, be, sd, batch, side, latent_scale):
        self._be, self._sd = be, sd
        self._plans: Dict[Tuple[int, int], VaeDecoderPlan] = {}
        self._enc_plans: Dict[Tuple[int, int], Any] = {}
        self._latent_scale = latent_scale
        self._plan(batch, side)

    def _plan(self, batch, side) -> VaeDecoderPlan:
        p = self._plans.get((batch, side))
        if p is None:
            # the plan applies 1/latent_scale itself; decode() receives already-unscaled latents
            p = VaeDecoderPlan(self._be, self._sd, batch, side, latent_scale=1.0)
            self._plans[(batch, side)] = p
        return p

    def parameters(self):
        return iter(next(iter(self._plans.values())).keep)

    @torch.no_grad()
    def decode(self, latents, *, return_dict: bool = True):
        b, c, h, w = latents.shape
        if c != 4 or h != w:
            raise ValueError(f"latents must be (B, 4, S, S), got {tuple(latents.shape)}")
        plan = self._plan(b, h)
        be = self._be
        be.wait_current()
        be.copy_(plan.z_in, latents.float())
        plan.run()
        img = be.clone(plan.img_out)           # [0,1] frames; see sample_is_unit_range
        be.release_to_current()
        # The reference decoder returns [-1,1] and _latents_to_images maps to [0,1]; the HIP decoder
        # fuses that tail into its last kernel, so hand back the equivalent [-1,1] tensor here.
        out = img * 2.0 - 1.0
        return SimpleNamespace(sample=out) if return_dict else out

    def _enc_plan(self, batch, side):
        from .engine import VaeEncoderPlan
        p = self._enc_plans.get((batch, side))
        if p is None:
            if "vae.vae.encoder.conv_in.weight" not in self._sd:
                raise KeyError("the state dict holds no VAE encoder ('vae.vae.encoder.*'): encode() is unavailable")
            p = self._enc_plans[(batch, side)] = VaeEncoderPlan(self._be, self._sd, batch, side)
        return p

    @torch.no_grad()
    def encode(self, images, *, return_dict: bool = True):
        """(vae.py:71-88) images (B,3,H,W) in [-1,1] -> ``.latent_dist`` with ``mean`` / ``logvar`` (clamped to
        [-30,20]) / ``std`` / ``var`` / ``mode()`` / ``sample(generator=None)``, as diffusers'
        ``AutoencoderKLOutput`` + ``DiagonalGaussianDistribution``."""
        b, c, h, w = images.shape
        if c != 3 or h != w or h % 8:
            raise ValueError(f"images must be (B, 3, S, S) with S a multiple of 8, got {tuple(images.shape)}")
        plan = self._enc_plan(b, h // 8)
        be = self._be
        x = images.float().contiguous()
        be.wait_current()
        be.copy_(plan.img_in, x)
        plan.run()
        dist = DiagonalGaussian(be, be.clone(plan.mean), be.clone(plan.logvar))
        be.release_to_current()
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)


class DiagonalGaussian:
    """The slice of diffusers' ``DiagonalGaussianDistribution`` the reference touches (``.sample()`` at
    src/models/diffusion_module_ip.py:410-411); sampling runs in the HIP library."""

    def __init__(self, be, mean, logvar):
        self._be, self.mean, self.logvar = be, mean, logvar       # logvar already clamped to [-30, 20]

    @property
    def std(self):
        return torch.exp(0.5 * self.logvar)

    @property
    def var(self):
        return torch.exp(self.logvar)

    def mode(self):
        return self.mean

    def sample(self, generator=None, *, noise=None, scale: float = 1.0):
        """mean + std * noise (``scale`` folds the latent scale of the training step in).  ``noise`` injects the
        draw (parity tests: CPU and device RNG streams differ)."""
        be = self._be
        if noise is None:
            noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=torch.float32)
        else:
            noise = noise.to(device=self.mean.device, dtype=torch.float32).contiguous()
        be.wait_current()
        out = be.empty(self.mean.shape, torch.float32)
        be.gaussian_sample(self.mean, self.logvar, noise, out, scale)
        be.release_to_current()
        return out


class DiffusionModuleWithIP:
    def __init__(self, cfg: Any, state_dict: Optional[Dict[str, torch.Tensor]] = None, *,
                 device=None, seed: int = 0, batch_size: Optional[int] = None,
                 clip_config: Optional[dict] = None, backend=None, warm_start_dis: bool = True,
                 _strict: Optional[bool] = None, _report=None):
        from .backend import HipBackend
        self.cfg = cfg
        self.diff_cfg = diff_cfg_from(cfg)
        dc = self.diff_cfg
        betas, ac = build_noise_schedule(dc)      # raises NotImplementedError first, like the reference
        self.device = torch.device(device if device is not None else "cuda:0")
        self.be = backend if backend is not None else HipBackend(self.device)
        self.training = False
        routing = dc.use_routing_gates
        gates = {"anatomy": dc.gate_init_anatomy, "disease": dc.gate_init_disease, "both": (0.5, 0.5)}
        emb = cfg.model.ordinal_embedder
        clip_pref = "image_encoder.image_encoder."
        if clip_config is None and state_dict is not None and (clip_pref + "visual_projection.weight") in state_dict:
            from .conditioning import clip_config_from_state_dict     # tower geometry from the file's tensors
            clip_config = clip_config_from_state_dict(
                {k[len(clip_pref):]: v for k, v in state_dict.items() if k.startswith(clip_pref)})
        shapes = dict(W.unet_shapes(routing_gates=routing))
        has_enc = state_dict is None or "vae.vae.encoder.conv_in.weight" in state_dict
        shapes.update(W.vae_shapes(encoder=has_enc))
        shapes.update(W.conditioning_shapes(
            num_classes=emb.num_classes, dim=cfg.model.embedding_dim, num_tokens=dc.num_aoe_tokens,
            clip_hidden=(clip_config or {}).get("hidden_size", 1024),
            clip_proj=(clip_config or {}).get("projection_dim", 768),
            projection_plus=dc.use_image_projection_plus, purifier=dc.use_feature_purifier,
            purifier_ff_mult=dc.purifier_ff_mult))
        user_sd = state_dict is not None
        if state_dict is None:
            state_dict = W.init_state_dict(shapes, seed, gates=gates, warm_start_dis=warm_start_dis,
                                           aoe_delta_scale=getattr(emb.aoe, "delta_scale", 0.1))
        if _strict is None:          # direct construction: every tensor of the inventory must be there
            missing = [k for k in shapes if k not in state_dict]
            if missing:
                raise KeyError(f"state dict lacks {len(missing)} tensors, e.g. {missing[:3]}")
        else:                        # load_from_checkpoint: load_state_dict(strict=...) semantics + a report
            # The inventory is the MODULE's, not the file's: the reference keeps the whole AutoencoderKL (encoder
            # included) and the CLIP tower as child modules (diffusion_module_ip.py:120-141), so a file without them
            # fails ``strict=True`` there; with ``strict=False`` they are reported missing and filled from the seed
            # (``load_report.filled_from_seed``), never silently.
            from . import checkpoint as CK
            inventory = {**shapes, **W.vae_shapes(encoder=True), **W.clip_shapes(clip_config)}
            state_dict = CK.reconcile(dict(state_dict), inventory, strict=_strict, seed=seed,
                                      init_kwargs=dict(gates=gates, aoe_delta_scale=getattr(emb.aoe, "delta_scale", 0.1)),
                                      report=_report)
        self.load_report = _report
        self._sd = state_dict
        self.latent_side = cfg.dataset.image_size // 8
        self.batch_size = batch_size or 1

        self.betas = betas.to(self.device)
        self.alphas_cumprod = ac.to(self.device)
        self.alphas_cumprod_prev = torch.cat([torch.ones(1), ac[:-1]]).to(self.device)
        self.snr_values = (ac / (1.0 - ac + 1e-8)).to(self.device)

        self.ordinal_embedder = AdditiveOrdinalEmbedder(
            state_dict, self.device, emb.num_classes, cfg.model.embedding_dim, dc.num_aoe_tokens, be=self.be)
        # The reference keeps the CLIP tower as an nn.Module child, so a Lightning state_dict carries
        # ``image_encoder.image_encoder.*``: those tensors are the tower.  A caller-supplied state
        # dict WITHOUT them gets a seeded random tower and a loud warning (bench / tests only).
        if (clip_pref + "visual_projection.weight") not in state_dict:
            if user_sd:
                import warnings
                warnings.warn("state dict has no 'image_encoder.image_encoder.*' tensors: the CLIP tower is SEEDED RANDOM "
                              "(anatomy tokens are noise for a real checkpoint)", RuntimeWarning, stacklevel=2)
            state_dict = dict(state_dict)
            state_dict.update(W.init_state_dict(W.clip_shapes(clip_config), seed))
        self.image_encoder = ImageEncoder(state_dict, self.device, clip_config=clip_config, be=self.be)
        proj_cls = ImageProjectionPlus if dc.use_image_projection_plus else ImageProjection
        self.image_projection = proj_cls(state_dict, self.device, dc.num_image_tokens, be=self.be)
        self.feature_purifier = (FeaturePurifier(state_dict, self.device, dc.purifier_num_heads, be=self.be)
                                 if dc.use_feature_purifier else None)
        self._unets: Dict[Tuple[int, int], OrdinalUNet] = {}
        self._loops: Dict[Tuple[int, int], DdimLoop] = {}
        self.unet = self._unet_for(self.batch_size, self.latent_side)
        self.vae = SDVAE(self.be, state_dict, self.batch_size, self.latent_side, dc.latent_scale)

    # ---- plans per (batch, side) ----------------------------------------------------------------
    def _unet_for(self, batch: int, side: int) -> OrdinalUNet:
        u = self._unets.get((batch, side))
        if u is None:
            dc = self.diff_cfg
            plan = UNetPlan(self.be, self._sd, batch, side, use_routing_gates=dc.use_routing_gates,
                            use_frequency_strategy=dc.use_frequency_strategy)
            u = OrdinalUNet(plan, dc.use_routing_gates, self.cfg.model.conditioning_dim,
                            self.cfg.model.latent_channels, self.cfg.model.latent_channels)
            if self._unets:     # keep delta_scale consistent across plans
                lam = next(iter(self._unets.values())).unet.delta_scale()
                for _, s in u.unet.named_modules():
                    if hasattr(s.processor, "delta_scale"):
                        s.processor.delta_scale = lam
            self._unets[(batch, side)] = u
        return u

    def ddim_loop(self, batch: int, side: int) -> DdimLoop:
        lp = self._loops.get((batch, side))
        if lp is None:
            lp = DdimLoop(self._unet_for(batch, side)._plan)
            self._loops[(batch, side)] = lp
        return lp

    # ---- reference protocol -----------------------------------------------------------------------
    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, cfg=None, weights_only=False, strict=False,
                             map_location=None, *, which: str = "ema", **kw):
        """Lightning ``.ckpt`` / plain state-dict / safetensors loader (checkpoint.py).
        ``state_dict`` of a file saved by the EMA callback holds the averaged weights (``which="ema"``, what the
        reference's inference loads), ``current_model_state`` the raw ones (``which="raw"``).  ``strict=False``
        (the reference's call, inference_pipeline_ip.py:587-592) tolerates missing / unexpected keys and reports
        them in ``module.load_report``.  Only loaders that execute nothing from the file are used, whatever
        ``weights_only`` says."""
        from . import checkpoint as CK
        from .config import to_attr
        sd, rep, blob = CK.load_state(str(checkpoint_path), which=which)
        if cfg is None:
            hp = blob.get("hyper_parameters") if isinstance(blob, dict) else None
            if isinstance(hp, dict) and isinstance(hp.get("cfg"), dict):
                cfg = to_attr(hp["cfg"])
            else:
                raise ValueError("cfg is required (the file carries no plain-dict hyper_parameters['cfg'])")
        mod = cls(cfg, state_dict=sd, _strict=bool(strict), _report=rep, **kw)
        return mod

    def to(self, *args, **kwargs):
        for a in args:
            if isinstance(a, (torch.device, str)) and torch.device(a).type != self.device.type:
                raise RuntimeError(f"this module lives on {self.device}; there is no {a} path")
        return self            # storage precision is fixed by the engine (fp16 tiles, fp32 accumulate)

    def half(self):
        return self

    def float(self):
        return self

    def eval(self):
        self.training = False
        return self

    def parameters(self):
        yield from self.unet.parameters()
        yield from self.ordinal_embedder.parameters()
        yield from self.image_projection.parameters()
        if self.feature_purifier is not None:
            yield from self.feature_purifier.parameters()

    def _get_image_embeds(self, structure_images: torch.Tensor) -> torch.Tensor:
        if self.diff_cfg.use_image_projection_plus:
            feats = self.image_encoder.get_hidden_states(structure_images)
        else:
            feats = self.image_encoder(structure_images)
        return self.image_projection(feats)

    # ---- training-step forward (src/models/diffusion_module_ip.py:289-462) -----------------------------------------
    def _sample_timesteps(self, batch_size: int) -> torch.Tensor:
        """(:289-297)"""
        return torch.randint(0, self.diff_cfg.num_train_timesteps, (batch_size,), device=self.device, dtype=torch.long)

    def _q_sample(self, x0: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
        """(:299-303) forward diffusion on the HIP kernel."""
        be = self.be
        x0 = x0.to(device=self.device, dtype=torch.float32).contiguous()
        noise = noise.to(device=self.device, dtype=torch.float32).contiguous()
        t = t.to(device=self.device, dtype=torch.long).contiguous()
        be.wait_current()
        out = be.empty(tuple(x0.shape), torch.float32)
        be.q_sample(x0, noise, t, self.alphas_cumprod, out)
        be.release_to_current()
        return out

    def _min_snr_weight(self, t: torch.Tensor) -> torch.Tensor:
        """(:305-313) Min-SNR-gamma weight per sample."""
        if not getattr(self.cfg.training, "use_min_snr_weighting", True):
            return torch.ones_like(t, dtype=torch.float32, device=self.device)
        snr = self.snr_values[t]
        return torch.minimum(snr, torch.tensor(self.diff_cfg.min_snr_gamma, device=snr.device)) / (snr + 1e-8)

    def _prepare_conditioning(self, labels, structure_images, is_training: bool = True):
        """(:334-381) training-time conditioning: source == target label, delta segment exactly zero."""
        aoe = self.ordinal_embedder(labels, is_training=is_training)
        if aoe.dim() == 2:
            aoe = aoe.unsqueeze(1)
        img = self._get_image_embeds(structure_images)
        if self.feature_purifier is not None:
            img = self.feature_purifier(img, aoe)
        if self.diff_cfg.use_routing_gates:
            return aoe, img, torch.zeros_like(aoe)
        return aoe, img

    def training_step(self, batch, batch_idx: int = 0, *, noise=None, t=None, drop_mask=None, latent_noise=None,
                      is_training: bool = True):
        """FORWARD half of ``training_step`` (:392-462): VAE encode -> latent sample x latent_scale -> q_sample ->
        conditioning (is_training noise on the AOE interpolation, CFG image-token dropout) -> eps prediction -> per-sample
        MSE x Min-SNR weight -> mean.  Returns the loss VALUE (a tensor without autograd history): the backward
        kernels, AdamW and EMA of BASELINE config 4 are not built, so ``loss.backward()`` raises as any leaf without
        grad does.  Keyword-only extras inject the random draws (parity tests): ``noise`` (eps), ``t``, ``drop_mask``,
        ``latent_noise`` (the VAE posterior draw); ``is_training=False`` switches the AOE's own interpolation noise off."""
        images, labels, structure_images = batch
        b = images.shape[0]
        dist = self.vae.encode(images).latent_dist
        latents = dist.sample(noise=latent_noise, scale=self.diff_cfg.latent_scale)
        if noise is None:
            noise = torch.randn_like(latents)
        noise = noise.to(device=self.device, dtype=torch.float32)
        if t is None:
            t = self._sample_timesteps(b)
        t = t.to(self.device)
        noisy = self._q_sample(latents, t, noise)
        parts = self._prepare_conditioning(labels, structure_images, is_training=is_training)
        aoe, img = parts[0], parts[1]
        if drop_mask is None:
            drop_mask = torch.rand(b, device=self.device) < getattr(self.cfg.model, "cfg_drop_prob", 0.1)
        img = torch.where(drop_mask.to(self.device).view(-1, 1, 1).expand_as(img), torch.zeros_like(img), img)
        cond = torch.cat([aoe, img, parts[2]] if self.diff_cfg.use_routing_gates else [aoe, img], dim=1)
        pred = self(noisy, t, cond)
        be = self.be
        be.wait_current()
        base = be.empty((b,), torch.float32)
        be.mse_rows(pred.contiguous(), noise.contiguous(), base)
        be.release_to_current()
        return (self._min_snr_weight(t) * base).mean()

    def configure_optimizers(self):
        raise NotImplementedError("training (backward kernels, fused AdamW, EMA: BASELINE config 4 / SURVEY.md §8f-4) is not "
                                  "built; training_step() evaluates the loss only")

    def __call__(self, latents: torch.Tensor, timesteps: torch.Tensor, cond_embed: torch.Tensor):
        b, _, s, _ = latents.shape
        u = self._unet_for(b, s)
        if u is not self.unet:          # processors of every plan follow the public one
            lam = self.unet.unet.delta_scale()
            for _, site in u.unet.named_modules():
                if hasattr(site.processor, "delta_scale"):
                    site.processor.delta_scale = lam
        return u(latents, timesteps, cond_embed)

    forward = __call__
