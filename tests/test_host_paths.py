"""CPU checks of the host-side functions around the sampler that no other file covers: the conditioning cache
of ``OrdinalUNet``, ``_apply_leace`` / ``_load_leace_projection``, the eta > 0 branch, the checkpoint
loader (Lightning EMA layout, nesting, safetensors, wrappers, fp16 tensors, strictness), ``load_config`` on a
YAML of the ``train_ip.yaml`` schema, CLIP preprocessing and the PNG writers.  Engine calls go through the
TEST-ONLY torch backend; the same functions run on the HIP backend in test_gpu_parity.py.
"""
import os

import pytest
import torch
import yaml

from oracle import sampler as OS
from progressive_stable_diffusion_amd import checkpoint as CK
from progressive_stable_diffusion_amd import config as CFG
from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
from progressive_stable_diffusion_amd import weights as W
from progressive_stable_diffusion_amd.config import default_config
from progressive_stable_diffusion_amd.diffusion_module_ip import DiffusionModuleWithIP, diff_cfg_from
from tests.torch_backend import TorchRefBackend

GATES = {"anatomy": (0.1, 0.9), "disease": (0.9, 0.1), "both": (0.5, 0.5)}
TINY_CLIP = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                 image_size=224, patch_size=14, projection_dim=32)


@pytest.fixture(scope="module")
def full_sd():
    shapes = dict(W.unet_shapes())
    shapes.update(W.vae_shapes(encoder=True))       # a Lightning file carries the whole AutoencoderKL
    shapes.update(W.conditioning_shapes(clip_hidden=TINY_CLIP["hidden_size"], clip_proj=TINY_CLIP["projection_dim"]))
    return W.init_state_dict(shapes, 0, gates=GATES, warm_start_dis=False)


@pytest.fixture(scope="module")
def mod(full_sd):
    cfg = default_config(**{"dataset.image_size": 64})
    with pytest.warns(RuntimeWarning, match="SEEDED RANDOM"):      # caller-supplied dict without a CLIP tower
        return DiffusionModuleWithIP(cfg, state_dict=full_sd, device="cpu", seed=0, batch_size=2,
                                     clip_config=TINY_CLIP, backend=TorchRefBackend())


def _ocfg(mod):
    return OS.OracleCfg(image_size=mod.cfg.dataset.image_size)


# ------------------------------------------------------------------------------------------- cond cache
def test_cond_cache_follows_plan_state(mod):
    """module(x,t,cA); plan.set_cond(cB); module(x,t,cA) must give eps(cA) again: the cache
    holds the tensor and is tied to the plan's conditioning generation."""
    torch.manual_seed(0)
    x, t = torch.randn(2, 4, 8, 8), torch.tensor([500, 20])
    cA, cB = torch.randn(2, 48, 768) * 0.5, torch.randn(2, 48, 768) * 0.5
    PIPE._set_delta_scale_on_processors(mod, 1.5)
    with torch.no_grad():
        eA = mod(x, t, cA).clone()
        gen = mod.unet._plan.cond_gen
        assert torch.equal(mod(x, t, cA), eA) and mod.unet._plan.cond_gen == gen      # cached: no re-projection
        mod.unet._plan.set_cond(cB, 0)                                               # what a sampler run does
        assert torch.equal(mod(x, t, cA), eA)
        eB = mod(x, t, cB)
        assert (eB - eA).abs().max().item() > 1e-3
        cA.mul_(2.0)                                                                 # in-place edit: new version
        assert (mod(x, t, cA) - eA).abs().max().item() > 1e-3
        # scalar / CPU-int timesteps and a (B, D)-shaped cond are normalised as unet.py:129-140 does
        e1 = mod(x, torch.tensor(7), cB)
        e2 = mod(x, torch.tensor([7, 7]), cB)
        assert torch.equal(e1, e2)


# ------------------------------------------------------------------------------------------- LEACE
def test_apply_leace_and_loader(tmp_path):
    torch.manual_seed(1)
    b, t, d = 2, 16, 8
    emb = torch.randn(b, t, d)
    q, _ = torch.linalg.qr(torch.randn(t * d, 3))
    p_null = torch.eye(t * d) - q @ q.T                      # projector onto the complement of 3 concept directions
    mu = torch.randn(t * d)
    path = tmp_path / "leace.pt"
    torch.save({"P_null": p_null, "mu": mu}, path)
    leace = PIPE._load_leace_projection(path, torch.device("cpu"))
    out = PIPE._apply_leace(emb, leace)
    assert out.shape == emb.shape
    ref = OS.apply_leace(emb, {"P_null": p_null, "mu": mu})
    assert torch.allclose(out, ref, atol=1e-6)
    # properties of the projection (inference_pipeline_ip.py:36-57): the erased directions vanish, idempotent
    assert ((out.reshape(b, -1) - mu) @ q).abs().max().item() < 1e-4
    assert torch.allclose(PIPE._apply_leace(out, leace), out, atol=1e-5)


def test_sampler_with_leace_and_image_scale_matches_oracle(mod):
    torch.manual_seed(2)
    target, source = torch.tensor([3.0, 1.0]), torch.tensor([0.0, 2.0])
    pix = torch.randn(1, 3, 224, 224)
    td = 16 * 768
    v = torch.randn(td, 1)
    v = v / v.norm()
    leace = {"P_null": torch.eye(td) - v @ v.T, "mu": torch.randn(td) * 0.1}
    with torch.no_grad():
        got = PIPE._prepare_conditioning(mod, target, source, pix, image_scale=0.5, leace=leace)
        feats = mod.image_encoder.get_hidden_states(pix)
        ref = OS.prepare_conditioning(mod._sd, _ocfg(mod), target, source, feats, image_scale=0.5, leace=leace)
    assert got.shape == (2, 48, 768) and (got - ref).abs().max().item() < 8e-3      # image segment: fp16 rows


# ------------------------------------------------------------------------------------------- eta > 0
def test_ddim_stochastic_matches_oracle(mod):
    """eta > 0 branch (inference_pipeline_ip.py:457-468) with the per-step noise injected on both sides."""
    torch.manual_seed(3)
    steps = 4
    target, source = torch.tensor([2.0, 0.5]), torch.tensor([0.0, 3.0])
    pix = torch.randn(1, 3, 224, 224)
    lat = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(7))
    noise = torch.randn(steps - 1, 2, 4, 8, 8, generator=torch.Generator().manual_seed(8))
    with torch.no_grad():
        got = PIPE._ddim_sample_ip(mod, target, source, pix, steps, torch.device("cpu"), eta=0.7, steer_scale=2.0,
                                   latents=lat, step_noise=noise)
        feats = mod.image_encoder.get_hidden_states(pix)
        ref = OS.ddim_sample(mod._sd, _ocfg(mod), target, source, feats, steps, lat, eta=0.7, steer_scale=2.0,
                             step_noise=noise)
        det = PIPE._ddim_sample_ip(mod, target, source, pix, steps, torch.device("cpu"), eta=0.0, steer_scale=2.0,
                                   latents=lat, use_graph=False)
    assert (got - ref).abs().max().item() < 5e-2
    assert (got - det).abs().max().item() > 1e-2            # the noise really entered
    # one update, op for op, against the oracle's restatement of :457-468
    ac = mod.alphas_cumprod
    x, e, n = torch.randn(2, 4, 8, 8), torch.randn(2, 4, 8, 8), torch.randn(2, 4, 8, 8)
    a_t, a_p = ac[500], ac[400]
    x0 = ((x - torch.sqrt(1 - a_t) * e) / torch.sqrt(a_t)).clamp(-4, 4)
    sig = 0.7 * torch.sqrt((1 - a_p) / (1 - a_t) * (1 - a_t / a_p))
    mine = torch.sqrt(a_p) * x0 + torch.sqrt(1 - a_p - sig ** 2) * e + sig * n
    assert torch.equal(mine, OS.ddim_update(x, e, ac, 500, 400, False, 0.7, n))


# ------------------------------------------------------------------------------------------- checkpoints
def _tiny_clip_sd():
    return W.init_state_dict(W.clip_shapes(TINY_CLIP), 3)


def _load(path, **kw):
    cfg = default_config(**{"dataset.image_size": 64})
    return DiffusionModuleWithIP.load_from_checkpoint(str(path), cfg=cfg, device="cpu", batch_size=1,
                                                      backend=TorchRefBackend(), **kw)


def test_checkpoint_layouts_round_trip(full_sd, tmp_path):
    """Flat dict, {"state_dict": ...} nesting, Lightning + EMA layout (averaged weights in ``state_dict``, raw in
    ``current_model_state``: ema_callback.py:291-324), wrapper prefixes, fp16 VAE tensors, non-persistent buffers,
    safetensors — all through the safe loaders — give the module the tensors that were saved."""
    from safetensors.torch import save_file
    sd = dict(full_sd)
    sd.update(_tiny_clip_sd())
    ema = {k: (v * 0.5 if k.startswith("ordinal_embedder.") else v) for k, v in sd.items()}
    lightning = {
        "state_dict": {**{("vae.vae._orig_mod." + k[8:] if k.startswith("vae.vae.") else k):
                          (v.half() if k.startswith("vae.vae.") else v) for k, v in ema.items()},
                       "alphas_cumprod": torch.zeros(1000)},       # must be ignored: recomputed (:180-193)
        "current_model_state": sd, "averaging_state": {"n_averaged": torch.tensor(12)},
        "callbacks": {"EMAWeightAveraging": {"latest_update_step": 48}}, "epoch": 3, "global_step": 50,
        "hyper_parameters": {"cfg": yaml.safe_load(yaml.safe_dump(_plain(default_config(**{"dataset.image_size": 64}))))},
    }
    p_l, p_flat, p_st = tmp_path / "last.ckpt", tmp_path / "flat.pt", tmp_path / "w.safetensors"
    torch.save(lightning, p_l)
    torch.save(sd, p_flat)
    save_file({k: v.contiguous() for k, v in sd.items()}, str(p_st))

    m_ema = _load(p_l)
    rep = m_ema.load_report
    assert rep.which == "ema" and rep.missing == [] and rep.unexpected == [] and "alphas_cumprod" in rep.skipped_buffers
    assert rep.widened == sum(1 for k in sd if k.startswith("vae.vae."))
    assert torch.equal(m_ema._sd["ordinal_embedder.base"], sd["ordinal_embedder.base"] * 0.5)
    k = "vae.vae.decoder.conv_in.weight"
    assert torch.equal(m_ema._sd[k], sd[k].half().float())
    assert m_ema.alphas_cumprod[999].item() == pytest.approx(0.001578963, rel=2e-6)
    m_raw = _load(p_l, which="raw")
    assert m_raw.load_report.which == "raw" and torch.equal(m_raw._sd["ordinal_embedder.base"], sd["ordinal_embedder.base"])
    # cfg recovered from plain-dict hyper-parameters when the caller passes none
    m_hp = DiffusionModuleWithIP.load_from_checkpoint(str(p_l), device="cpu", batch_size=1, backend=TorchRefBackend())
    assert m_hp.cfg.dataset.image_size == 64 and m_hp.diff_cfg.use_routing_gates
    for p in (p_flat, p_st):
        m = _load(p)
        assert m.load_report.missing == [] and torch.equal(m._sd["unet.unet.conv_in.weight"], sd["unet.unet.conv_in.weight"])
        # the CLIP tower came from the file, not from the seed
        w = m.image_encoder.p["vision_model.embeddings.class_embedding"]
        assert torch.equal(w.cpu(), sd["image_encoder.image_encoder.vision_model.embeddings.class_embedding"])
        assert m.image_encoder.cfg["hidden_size"] == 64 and m.image_encoder.layers == 2      # geometry from the tensors
    # same eps from the loaded module as from the in-memory one
    torch.manual_seed(4)
    x, t, c = torch.randn(1, 4, 8, 8), torch.tensor([300]), torch.randn(1, 48, 768) * 0.5
    with pytest.warns(RuntimeWarning):
        m0 = DiffusionModuleWithIP(default_config(**{"dataset.image_size": 64}), state_dict=dict(full_sd), device="cpu",
                                   batch_size=1, clip_config=TINY_CLIP, backend=TorchRefBackend())
    with torch.no_grad():
        assert torch.equal(m0(x, t, c), m_raw(x, t, c))


def _plain(o):
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    return o


def test_checkpoint_strictness_and_reports(full_sd, tmp_path):
    sd = dict(full_sd)
    sd.update(_tiny_clip_sd())
    part = {k: v for k, v in sd.items() if not k.startswith("feature_purifier.") and "to_k_dis" not in k}
    part["some.legacy.tensor"] = torch.zeros(3)
    p = tmp_path / "partial.pt"
    torch.save({"state_dict": part}, p)
    with pytest.warns(RuntimeWarning, match="SEEDED RANDOM"):
        m = _load(p, strict=False)
    rep = m.load_report
    assert "some.legacy.tensor" in rep.unexpected and any(k.startswith("feature_purifier.") for k in rep.missing)
    assert set(rep.filled_from_seed) == set(rep.missing)
    kd = "unet.unet.mid_block.attentions.0.transformer_blocks.0.attn2.processor.to_k_dis.weight"
    assert torch.equal(m._sd[kd], sd[kd.replace("processor.to_k_dis", "to_k")])     # warm start (:308-314)
    with pytest.raises(RuntimeError, match="missing"):
        _load(p, strict=True)
    # the inventory is the module's, not the file's — a file WITHOUT the CLIP tower or the VAE
    # encoder (both child modules of the reference's LightningModule) fails strict=True and is reported under strict=False
    for drop in ("image_encoder.image_encoder.", "vae.vae.encoder."):
        lacking = {k: v for k, v in sd.items() if not k.startswith(drop)}
        torch.save(lacking, tmp_path / "lacking.pt")
        with pytest.raises(RuntimeError, match="missing"):
            _load(tmp_path / "lacking.pt", strict=True, clip_config=TINY_CLIP)
        with pytest.warns(RuntimeWarning, match="SEEDED RANDOM"):
            m2 = _load(tmp_path / "lacking.pt", strict=False, clip_config=TINY_CLIP)
        miss = m2.load_report.missing
        assert miss and all(k.startswith(drop) for k in miss) and set(m2.load_report.filled_from_seed) == set(miss)
    bad = dict(sd)
    bad["unet.unet.conv_in.weight"] = torch.zeros(320, 4, 1, 1)
    torch.save(bad, tmp_path / "bad.pt")
    with pytest.raises(RuntimeError, match="size mismatch"):
        _load(tmp_path / "bad.pt")
    # a file the safe loader refuses is reported, never unpickled
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    with open(tmp_path / "evil.ckpt", "wb") as f:
        pickle.dump({"state_dict": {}, "hyper_parameters": Evil()}, f)
    with pytest.raises(RuntimeError, match="refused"):
        _load(tmp_path / "evil.ckpt")
    with pytest.raises(ValueError, match="cfg is required"):
        DiffusionModuleWithIP.load_from_checkpoint(str(tmp_path / "bad.pt"), device="cpu", backend=TorchRefBackend())


def test_diffusers_layout_files_merge(full_sd, tmp_path):
    from safetensors.torch import save_file
    unet = {k[len("unet.unet."):]: v.contiguous() for k, v in full_sd.items()
            if k.startswith("unet.unet.") and ".processor." not in k}
    vae = {k[len("vae.vae."):]: v.half().contiguous() for k, v in full_sd.items() if k.startswith("vae.vae.")}
    save_file(unet, str(tmp_path / "unet.safetensors"))
    save_file(vae, str(tmp_path / "vae.safetensors"))
    dadd = {k: v for k, v in full_sd.items() if not k.startswith(("unet.unet.", "vae.vae.")) or ".processor." in k}
    merged = CK.merge_diffusers_files(str(tmp_path / "unet.safetensors"), str(tmp_path / "vae.safetensors"), dadd)
    assert set(merged) == set(full_sd)
    assert torch.equal(merged["unet.unet.conv_in.weight"], full_sd["unet.unet.conv_in.weight"])
    assert merged["vae.vae.decoder.conv_out.weight"].dtype == torch.float32


# ------------------------------------------------------------------------------------------- config
def test_load_config_yaml_with_train_ip_schema(tmp_path):
    """A YAML with the reference's ``configs/train_ip.yaml`` schema -> attribute access + ``diff_cfg_from``."""
    doc = _plain(default_config())
    doc["hydra"] = {"output_subdir": None}
    doc["optimizer"] = {"name": "adamw", "lr": 1e-4, "weight_decay": 0.001, "betas": [0.9, 0.999]}
    doc["model"]["gate_init_anatomy"] = [0.1, 0.9]
    path = tmp_path / "train_ip.yaml"
    path.write_text(yaml.safe_dump(doc))
    cfg = PIPE._load_config(path)
    assert cfg.model.ordinal_embedder.aoe.delta_scale == 0.05 and cfg.optimizer.betas == [0.9, 0.999]
    assert getattr(cfg.model, "not_a_key", 7) == 7                 # DictConfig-style getattr default
    dc = diff_cfg_from(cfg)
    assert dc.gate_init_anatomy == (0.1, 0.9) and dc.latent_scale == 0.18215 and dc.num_train_timesteps == 1000
    assert dc.image_encoder_path == "openai/clip-vit-large-patch14" and dc.use_image_projection_plus is True
    with pytest.raises(FileNotFoundError):
        CFG.load_config(tmp_path / "nope.yaml")
    ref_yaml = "/root/reference/configs/train_ip.yaml"      # present in the build container only
    if os.path.exists(ref_yaml):
        rc = CFG.load_config(ref_yaml)
        rdc = diff_cfg_from(rc)
        assert rdc == diff_cfg_from(default_config()), "DEFAULTS drifted from the shipped train_ip.yaml"
        assert rc.dataset.image_size == 256 and rc.training.gradient_clip_val == 1.0


# ------------------------------------------------------------------------------------------- images
def test_clip_preprocess_and_png_writers(tmp_path):
    from PIL import Image
    import numpy as np
    g = torch.Generator().manual_seed(5)
    disp = torch.rand(3, 96, 128, generator=g)
    px = PIPE._clip_preprocess(disp)
    assert px.shape == (1, 3, 224, 224)
    # a constant image stays constant under resize/crop: (c - mean)/std per channel
    flat = PIPE._clip_preprocess(torch.full((3, 64, 64), 0.5))
    for c in range(3):
        assert flat[0, c].std().item() < 1e-5
        assert flat[0, c].mean().item() == pytest.approx((0.5 - PIPE.CLIP_MEAN[c]) / PIPE.CLIP_STD[c], abs=1e-4)
    img_path = tmp_path / "s.png"
    Image.fromarray((np.random.RandomState(0).rand(80, 100, 3) * 255).astype("uint8")).save(img_path)
    clip, display = PIPE._load_and_preprocess_structure_image(img_path, 64, torch.device("cpu"))
    assert clip.shape == (1, 3, 224, 224) and display.shape == (3, 64, 64) and 0.0 <= float(display.min())
    imgs = torch.rand(9, 3, 32, 32, generator=g)
    labels = PIPE._build_labels(9)
    PIPE._save_sequence(imgs, labels, tmp_path / "out", display)
    names = sorted(os.listdir(tmp_path / "out"))
    assert "structure_reference.png" in names and "mes_0.00_00.png" in names and "mes_3.00_08.png" in names
    back = torch.from_numpy(np.asarray(Image.open(tmp_path / "out" / "mes_0.00_00.png"))).permute(2, 0, 1)
    assert torch.equal(back, imgs[0].mul(255).to(torch.uint8))
    grid = PIPE._create_progression_grid(imgs, labels, display, tmp_path / "grid.png")
    # 9 images -> 7 columns, 2 rows + the structure row; 4 px padding (:513-563)
    assert grid.size == (7 * 36 + 4, 3 * 36 + 4) and (tmp_path / "grid.png").exists()


def test_seed_and_device_helpers():
    PIPE._set_seed(11)
    a = torch.rand(3)
    PIPE._set_seed(11)
    assert torch.equal(a, torch.rand(3))
    args = PIPE._parse_args(["--checkpoint", "c.ckpt", "--structure-image", "s.png"])
    assert args.mes_steps == 13 and args.sampling_steps == 50 and args.steer_scale == 0.0 and args.eta == 0.0


# ------------------------------------------------------------------------------------------- training-step forward
def test_training_step_forward_matches_oracle(full_sd):
    """The forward half of ``training_step`` (diffusion_module_ip.py:392-462) through the engine (torch test backend)
    vs ``oracle.training``: VAE encode + posterior sample, q_sample, training-time conditioning with a zero delta
    segment, CFG image-token dropout, eps-MSE x Min-SNR.  Every random draw is injected."""
    from oracle import training as OT
    sd = dict(full_sd)
    sd.update(W.init_state_dict(W.vae_shapes(decoder=False), 0))
    cfg = default_config(**{"dataset.image_size": 64})
    with pytest.warns(RuntimeWarning):
        m = DiffusionModuleWithIP(cfg, state_dict=sd, device="cpu", batch_size=2, clip_config=TINY_CLIP, backend=TorchRefBackend())
    g = torch.Generator().manual_seed(31)
    images = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    labels = torch.tensor([1.0, 3.0])
    pix = torch.randn(2, 3, 224, 224, generator=g)
    t = torch.tensor([700, 35])
    noise, lat_noise = torch.randn(2, 4, 8, 8, generator=g), torch.randn(2, 4, 8, 8, generator=g)
    drop = torch.tensor([False, True])
    with torch.no_grad():
        loss = m.training_step((images, labels, pix), 0, noise=noise, t=t, drop_mask=drop, latent_noise=lat_noise, is_training=False)
        feats = m.image_encoder.get_hidden_states(pix)
        ref, base = OT.training_loss(m._sd, _ocfg(m), images, labels, feats, t, noise, lat_noise, drop)
    assert loss.shape == () and not loss.requires_grad
    assert abs(loss.item() - ref.item()) < 2e-2 * max(1.0, abs(ref.item())), (loss.item(), ref.item())
    with pytest.raises(RuntimeError):
        loss.backward()                                     # no autograd history: the backward pass is not built
    with pytest.raises(NotImplementedError):
        m.configure_optimizers()
    # pieces: q_sample and the Min-SNR weight against their restatements, known answers of App. C
    x0 = torch.randn(2, 4, 8, 8, generator=g)
    assert torch.allclose(m._q_sample(x0, t, noise), OT.q_sample(m.alphas_cumprod, x0, t, noise), atol=1e-6)
    w = m._min_snr_weight(torch.tensor([999, 0]))
    assert w[0].item() == pytest.approx(1.0, abs=1e-5)      # snr[999] = 0.00158 < gamma = 1 -> clipped / snr = 1
    assert w[1].item() == pytest.approx(1.0 / (0.999149978 / (1 - 0.999149978 + 1e-8)), rel=1e-3)
    assert m._sample_timesteps(5).shape == (5,) and m._sample_timesteps(5).dtype == torch.long
    parts = m._prepare_conditioning(labels, pix, is_training=False)
    assert len(parts) == 3 and float(parts[2].abs().max()) == 0.0 and parts[1].shape == (2, 16, 768)


def test_warmup_cosine_schedule_known_answers():
    from oracle import training as OT
    from progressive_stable_diffusion_amd.lr_scheduler import warmup_cosine_lr
    kw = dict(warmup_epochs=2, max_epochs=150, warmup_start_lr=1e-6, eta_min=1e-6)       # configs/train_ip.yaml:45-56
    base = [1e-4, 1e-4, 2e-4, 2e-4]                                                       # four parameter groups (:500-515)
    assert warmup_cosine_lr(0, base, **kw) == pytest.approx([1e-6] * 4)
    assert warmup_cosine_lr(1, base, **kw) == pytest.approx([(1e-6 + 1e-4) / 2] * 2 + [(1e-6 + 2e-4) / 2] * 2)
    assert warmup_cosine_lr(2, base, **kw) == pytest.approx(base)
    assert warmup_cosine_lr(150, base, **kw) == pytest.approx([1e-6] * 4)
    assert warmup_cosine_lr(400, base, **kw) == pytest.approx([1e-6] * 4)
    mid = warmup_cosine_lr(76, base, **kw)
    assert mid[0] == pytest.approx(1e-6 + (1e-4 - 1e-6) * 0.5, rel=1e-9)
    for e in (0, 1, 5, 76, 149, 150):
        assert warmup_cosine_lr(e, base, **kw) == pytest.approx(OT.warmup_cosine_lr(e, base, **kw))
