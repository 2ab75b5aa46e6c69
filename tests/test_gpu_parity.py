"""-m gpu: the HIP hot path (through the C ABI) against the CPU oracle and the committed golden
vectors, plus size-independent properties at BASELINE.json's full size.

Stated tolerances.  The engine stores activations in fp16 and accumulates in fp32; the oracle is
fp32 throughout.  * one UNet call: max|eps_hip - eps_oracle| <= 1e-2 * max|eps| (observed ~2e-3);
* DDIM trajectories: |latents| <= 4 by the clamp; the first update divides by sqrt(abar_999)=0.0397,
  amplifying an eps error 25x (and CFG by 1+2g), so latents are compared at 5e-2 (steer) / 0.25 (CFG);
* frames in [0,1]: 3e-2 max (u8: <= 8 levels), mean <= 2e-3;  * fp32 DDIM algebra: bit-exact
  (test_gpu_kernels.py).  Golden cross-attention vectors (the reference's own outputs): 4e-3.
"""
import os

import numpy as np
import pytest
import torch

from oracle import sampler as OS
from oracle.sd_unet import unet_forward
from oracle.sd_vae import vae_decode
from tests import golden_inputs as GI

pytestmark = pytest.mark.gpu
F16, F32 = torch.float16, torch.float32
DEV = torch.device("cuda:0")
TINY_CLIP = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                 image_size=224, patch_size=14, projection_dim=32)


@pytest.fixture(scope="module")
def hip():
    from progressive_stable_diffusion_amd.backend import HipBackend
    return HipBackend(DEV)


@pytest.fixture(scope="module")
def full_sd():
    from progressive_stable_diffusion_amd import weights as W
    shapes = dict(W.unet_shapes())
    shapes.update(W.vae_shapes(encoder=False))
    shapes.update(W.conditioning_shapes())
    return W.init_state_dict(shapes, 0, gates=GI.GATES, warm_start_dis=False)


def _module(sd, image_size, batch, clip_config=None, **cfg_over):
    from progressive_stable_diffusion_amd.config import default_config
    from progressive_stable_diffusion_amd.diffusion_module_ip import DiffusionModuleWithIP
    cfg = default_config(**{"dataset.image_size": image_size, **cfg_over})
    return DiffusionModuleWithIP(cfg, state_dict=sd, device=DEV, seed=0, batch_size=batch,
                                 clip_config=clip_config)


def _ocfg(mod):
    dc = mod.diff_cfg
    return OS.OracleCfg(image_size=mod.cfg.dataset.image_size, use_routing_gates=dc.use_routing_gates)


@pytest.mark.parametrize("site,c,n", GI.XATTN_CASES)
def test_golden_cross_attention_through_hip(hip, site, c, n, golden_dir):
    """The reference's own SplitInjection / OrdinalIP outputs (tests/golden/xattn.npz) reproduced by
    to_q (MFMA GEMM) -> fused tri_xattn kernel -> to_out (MFMA GEMM + bias)."""
    from progressive_stable_diffusion_amd import weights as W
    g = np.load(os.path.join(golden_dir, "xattn.npz"))
    ush = W.unet_shapes()
    ap = f"unet.unet.{site}.transformer_blocks.0.attn2"
    sd = W.init_state_dict(ush, GI.SEED, gates=GI.GATES, warm_start_dis=False,
                           keys=[k for k in ush if k.startswith(ap + ".")])
    x, cond3 = GI.xattn_inputs(c, n)
    tag = site.replace(".", "_")
    d = lambda t, dt=F16: hip.to_device(t, dt)   # noqa: E731
    xq = d(x.reshape(1, 1, n, c))
    q, att, out = (hip.zeros((1, 1, n, c), F16) for _ in range(3))
    hip.igemm(xq, d(sd[ap + ".to_q.weight"]), q)
    gates = d(torch.stack([sd[ap + ".processor.anat_gate"], sd[ap + ".processor.dis_gate"]]), F32)
    w_out, b_out = d(sd[ap + ".to_out.0.weight"]), d(sd[ap + ".to_out.0.bias"], F32)

    def run(kv_w, cond, mode, lam):
        kv = hip.zeros((1, 1, cond.shape[1], kv_w.shape[0]), F16)
        hip.igemm(d(cond.reshape(1, 1, cond.shape[1], 768)), d(kv_w), kv)
        hip.tri_xattn(q.view(1, n, c), kv.view(1, cond.shape[1], -1), att.view(1, n, c),
                      gates if mode == 0 else None, lam, mode, 8)
        hip.igemm(att, w_out, out, bias=b_out, flags=1)
        hip.synchronize()
        return out.view(1, n, c).float().cpu()

    kv4 = torch.cat([sd[ap + ".to_k.weight"], sd[ap + ".to_v.weight"],
                     sd[ap + ".processor.to_k_dis.weight"], sd[ap + ".processor.to_v_dis.weight"]])
    for lam in GI.LAMBDAS:
        ref = torch.from_numpy(g[f"{tag}__split_l{lam}"])
        assert (run(kv4, cond3, 0, lam) - ref).abs().max().item() < 4e-3, (site, lam)
    kv2 = torch.cat([sd[ap + ".to_k.weight"], sd[ap + ".to_v.weight"]])
    for mode in GI.MODES:
        ref = torch.from_numpy(g[f"{tag}__base_{mode}"])
        assert (run(kv2, cond3[:, :32], 1, 0.0) - ref).abs().max().item() < 4e-3, (site, mode)


class _Attention(torch.nn.Module):
    """What a diffusers Attention exposes to its processor (the attributes the reference reads)."""
    spatial_norm = group_norm = norm_cross = None
    residual_connection = False
    rescale_output_factor = 1.0

    def __init__(self, sd, ap, c, heads=8):
        super().__init__()
        self.heads = heads
        self.to_q, self.to_k, self.to_v = (torch.nn.Linear(i, c, bias=False) for i in (c, 768, 768))
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(c, c), torch.nn.Dropout(0.0)])
        self.load_state_dict({k[len(ap) + 1:]: v for k, v in sd.items()
                              if k.startswith(ap + ".") and ".processor." not in k})


@pytest.mark.parametrize("site,c,n", GI.XATTN_CASES)
def test_golden_vectors_through_processor_classes(site, c, n, golden_dir):
    """The drop-in processor classes (diffusers protocol, reference state-dict keys) called the way
    diffusers calls them, against the reference's own outputs; also the reference's error behaviour."""
    from progressive_stable_diffusion_amd import attention_processors as AP
    from progressive_stable_diffusion_amd import weights as W
    g = np.load(os.path.join(golden_dir, "xattn.npz"))
    ush = W.unet_shapes()
    ap = f"unet.unet.{site}.transformer_blocks.0.attn2"
    sd = W.init_state_dict(ush, GI.SEED, gates=GI.GATES, warm_start_dis=False,
                           keys=[k for k in ush if k.startswith(ap + ".")])
    attn = _Attention(sd, ap, c).to(DEV)
    x, cond3 = GI.xattn_inputs(c, n)
    x, cond3 = x.to(DEV), cond3.to(DEV)
    tag = site.replace(".", "_")
    proc = AP.SplitInjectionAttentionProcessor(c, 768, block_type=AP.get_block_type(site + ".x"))
    proc.load_state_dict({k[len(ap) + 11:]: v for k, v in sd.items() if k.startswith(ap + ".processor.")})
    proc = proc.to(DEV)
    assert set(proc.state_dict()) == {"anat_gate", "dis_gate", "to_k_dis.weight", "to_v_dis.weight"}
    for lam in GI.LAMBDAS:
        proc.delta_scale = lam
        out = proc(attn, x, encoder_hidden_states=cond3)
        torch.cuda.synchronize()
        ref = torch.from_numpy(g[f"{tag}__split_l{lam}"])
        assert out.shape == x.shape and out.dtype == x.dtype
        assert (out.float().cpu() - ref).abs().max().item() < 4e-3, (site, lam)
    for mode in GI.MODES:
        base = AP.OrdinalIPAttnProcessor2_0(c, 768, frequency_mode=mode)
        out = base(attn, x, encoder_hidden_states=cond3[:, :32].contiguous())
        torch.cuda.synchronize()
        assert (out.float().cpu() - torch.from_numpy(g[f"{tag}__base_{mode}"])).abs().max().item() < 4e-3
    with pytest.raises(ValueError):
        proc(attn, x, encoder_hidden_states=cond3[:, :32].contiguous())
    attn.norm_cross = True
    with pytest.raises(NotImplementedError):
        AP.OrdinalIPAttnProcessor2_0(c, 768)(attn, x, encoder_hidden_states=cond3[:, :32].contiguous())


@pytest.mark.parametrize("side,lam", [(16, 3.0), (24, 0.0)])
def test_unet_call_matches_oracle(hip, full_sd, side, lam):
    """One module(latents, t, cond) call; side 24 gives ragged attention lengths (576/144/36/9 keys)."""
    from progressive_stable_diffusion_amd.engine import UNetPlan
    b = 2
    plan = UNetPlan(hip, full_sd, b, side)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(b, 4, side, side, generator=g)
    cond = torch.randn(b, 48, 768, generator=g) * 0.5
    t = torch.tensor([999, 261])
    with torch.no_grad():
        ref = unet_forward(full_sd, x, t, cond, delta_scale=lam)
    got = plan.forward(x.to(DEV), t.to(DEV), cond.to(DEV), lam=lam)
    hip.synchronize()
    err = (got.cpu() - ref).abs().max().item()
    assert err < 1e-2 * max(1.0, ref.abs().max().item()), err


def test_vae_decode_matches_oracle(hip, full_sd):
    from progressive_stable_diffusion_amd.engine import VaeDecoderPlan
    b, s = 1, 16
    plan = VaeDecoderPlan(hip, full_sd, b, s, latent_scale=0.18215)
    z = torch.randn(b, 4, s, s, generator=torch.Generator().manual_seed(8)) * 0.18215 * 1.5
    with torch.no_grad():
        ref = ((vae_decode(full_sd, z / 0.18215).clamp(-1, 1) + 1) / 2).clamp(0, 1)
    hip.copy_(plan.z_in, z.to(DEV))
    plan.run()
    hip.synchronize()
    d = (plan.img_out.cpu() - ref).abs()
    assert d.max().item() < 3e-2 and d.mean().item() < 2e-3, (d.max().item(), d.mean().item())


_ORACLE_RUNS = {}


def _oracle_sample(key, sd, ocfg, target, source, feats, steps, lat, **kw):
    """The oracle's sampler with its per-step trace, run once per ``key`` and shared between the end-to-end test and
    the teacher-forced per-step test of the same configuration (tens of CPU-seconds each)."""
    hit = _ORACLE_RUNS.get(key)
    if hit is None:
        tr = []
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        with torch.no_grad():
            z = OS.ddim_sample(sd, ocfg, target, source, feats, steps, lat, trace=tr, **kw)
        hit = _ORACLE_RUNS[key] = (z, tr, feats)
    assert torch.equal(hit[2], feats), "the cached oracle run must have seen the same CLIP features"
    return hit[0], hit[1]


@pytest.mark.parametrize("gates_on", [True, False])
def test_config1_sampler_matches_oracle(full_sd, gates_on):
    """BASELINE config 1: 1 image, 256x256, 10 DDIM steps, 'guidance 3.0' in both readings
    (SURVEY.md §8d): (i) routing gates + steer lambda=3.0, (ii) baseline + CFG g=3.0.  Full-size
    CLIP ViT-L/14 tower (random init) in front, VAE decode behind."""
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    mod = _module(full_sd, 256, 1, **{"model.use_routing_gates": gates_on})
    target, source = torch.tensor([3.0]), torch.tensor([0.0])
    pix = torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1
    lat = torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(1234))
    kw = dict(steer_scale=3.0) if gates_on else dict(guidance_scale=3.0)
    with torch.no_grad():
        z = PIPE._ddim_sample_ip(mod, target.to(DEV), source.to(DEV), pix.to(DEV), 10, DEV, latents=lat, **kw)
        z_eager = PIPE._ddim_sample_ip(mod, target.to(DEV), source.to(DEV), pix.to(DEV), 10, DEV,
                                       latents=lat, use_graph=False, **kw)
        img = PIPE._latents_to_images(mod, z)
        feats = mod.image_encoder.get_hidden_states(pix.to(DEV)).cpu()
        z_ref, _ = _oracle_sample(("c1", gates_on), full_sd, _ocfg(mod), target, source, feats, 10, lat, **kw)
        img_ref = OS.latents_to_images(full_sd, _ocfg(mod), z_ref)
    assert torch.equal(z.cpu(), z_eager.cpu()), "hipGraph replay must equal eager launches bit for bit"
    if gates_on:        # no CFG: conv_out applies the DDIM update itself; the traced run keeps eps and a separate update
        tr = []
        with torch.no_grad():
            z_tr = PIPE._ddim_sample_ip(mod, target.to(DEV), source.to(DEV), pix.to(DEV), 10, DEV, latents=lat,
                                        trace=tr, **kw)
        assert len(tr) == 10 and torch.equal(z_tr.cpu(), z.cpu()), "fused conv_out + DDIM update must equal the two launches"
    ez = (z.cpu() - z_ref).abs().max().item()
    di = (img.cpu() - img_ref).abs()
    u8 = ((img.cpu() * 255).to(torch.uint8).int() - (img_ref * 255).to(torch.uint8).int()).abs()
    print(f"config1 gates={gates_on}: latents {ez:.3e} frames max {di.max():.3e} mean {di.mean():.3e} "
          f"u8 max {int(u8.max())} mean {u8.float().mean():.3f}")
    assert ez < (5e-2 if gates_on else 0.25)
    assert di.max().item() < (3e-2 if gates_on else 0.1) and di.mean().item() < 4e-3


@pytest.mark.parametrize("gates_on", [True, False])
def test_config1_teacher_forced_eps_per_step(full_sd, gates_on):
    """The test that does NOT amplify: BASELINE config 1, both readings, TEACHER-FORCED — the
    oracle's x_t of every one of the 10 steps goes into the HIP UNet (the product's ``module(latents, t, cond)`` with the
    product's own conditioning) and eps is compared step by step: max|eps_hip - eps_oracle| <= 1e-2 * max|eps_oracle|
    per step and per branch (conditional / unconditional under CFG).  End-of-trajectory latents (the tests above) divide
    the first error by sqrt(abar_999) = 0.04; this bound is on the kernels themselves.
    (reference loop: src/pipelines/inference/inference_pipeline_ip.py:423-456)"""
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    mod = _module(full_sd, 256, 1, **{"model.use_routing_gates": gates_on})
    target, source = torch.tensor([3.0]), torch.tensor([0.0])
    pix = torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1
    lat = torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(1234))
    kw = dict(steer_scale=3.0) if gates_on else dict(guidance_scale=3.0)
    with torch.no_grad():
        feats = mod.image_encoder.get_hidden_states(pix.to(DEV)).cpu()
        _, tr = _oracle_sample(("c1", gates_on), full_sd, _ocfg(mod), target, source, feats, 10, lat, **kw)
        cond = PIPE._prepare_conditioning(mod, target.to(DEV), source.to(DEV), pix.to(DEV))
        uncond = None if gates_on else PIPE._prepare_conditioning(mod, target.to(DEV), source.to(DEV), pix.to(DEV),
                                                                  zero_aoe=True)
        PIPE._set_delta_scale_on_processors(mod, 3.0 if gates_on else 0.0)
        ts = torch.linspace(999, 0, 10, dtype=torch.long, device=DEV)
        worst = 0.0
        for i, (eps_ref, _, x_t, parts) in enumerate(tr):
            t = ts[i].expand(1)
            refs = [(cond, eps_ref)] if gates_on else [(cond, parts[0]), (uncond, parts[1])]
            for c, ref in refs:
                got = mod(x_t.to(DEV), t, c).cpu()
                rel = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
                worst = max(worst, rel)
                assert rel < 1e-2, (gates_on, i, int(ts[i]), rel)
    print(f"config1 teacher-forced gates={gates_on}: worst per-step eps error {worst:.3e} of max|eps|")


def test_ddim_timestep_grid_on_device():
    """``torch.linspace(T-1, 0, steps, dtype=long, device=cuda)`` — how ``_ddim_sample_ip`` builds its grid
    (src/pipelines/inference/inference_pipeline_ip.py:389-395) — must give SURVEY.md App. C's irregular integer grids on
    the DEVICE too (the CPU oracle's grid is asserted in tests/test_oracle_golden.py)."""
    g50 = [999, 978, 958, 937, 917, 897, 876, 856, 835, 815, 795, 774, 754, 733, 713, 693, 672, 652, 632, 611, 591, 570,
           550, 530, 509, 489, 468, 448, 428, 407, 387, 366, 346, 326, 305, 285, 265, 244, 224, 203, 183, 163, 142, 122,
           101, 81, 61, 40, 20, 0]
    g13 = [999, 915, 832, 749, 666, 582, 499, 416, 333, 249, 166, 83, 0]
    g10 = [999, 888, 777, 666, 555, 444, 333, 222, 111, 0]
    for steps, want in ((50, g50), (13, g13), (10, g10)):
        got = torch.linspace(999, 0, steps=steps, dtype=torch.long, device=DEV)
        assert got.cpu().tolist() == want, steps
        assert got.cpu().tolist() == OS.ddim_timesteps(1000, steps).tolist()


def test_fused_attn2_sampler_matches_oracle(full_sd, monkeypatch):
    """attn2 folded into one kernel (x (W_q K^T) -> 24 softmaxes -> P (V W_o^T)) at every eligible site (opt-in
    path, off by default): 256x256, B=1, 4 steps, lambda=3, vs the oracle."""
    from progressive_stable_diffusion_amd import engine as E
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    monkeypatch.setattr(E, "FUSED_ATTN2", True)
    monkeypatch.setattr(E, "A2_MIN_TILES", 1)
    mod = _module(full_sd, 256, 1)
    assert len(mod.ddim_loop(1, 32).u.a2) >= 8, "fused sites expected at 32x32 and 16x16"
    target, source = torch.tensor([3.0]), torch.tensor([0.0])
    pix = torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1
    lat = torch.randn(1, 4, 32, 32, generator=torch.Generator().manual_seed(1234))
    with torch.no_grad():
        z = PIPE._ddim_sample_ip(mod, target.to(DEV), source.to(DEV), pix.to(DEV), 4, DEV, latents=lat, steer_scale=3.0)
        feats = mod.image_encoder.get_hidden_states(pix.to(DEV)).cpu()
        z_ref = OS.ddim_sample(full_sd, _ocfg(mod), target, source, feats, 4, lat, steer_scale=3.0)
    assert (z.cpu() - z_ref).abs().max().item() < 5e-2


def test_batched_sampler_matches_oracle(full_sd):
    """``_ddim_sample_batched`` (the data-augmentation / evaluation copy of the sampler): one structure
    image and one noise draw PER sample; 128x128, 4 steps, B=2, lambda=2."""
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    mod = _module(full_sd, 128, 2)
    target, source = torch.tensor([3.0, 0.0]), torch.tensor([1.0, 2.0])
    pix = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(5)) * 2 - 1
    lat = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(99))
    with torch.no_grad():
        z = PIPE._ddim_sample_batched(mod, target.to(DEV), source.to(DEV), pix.to(DEV), 4, DEV,
                                      steer_scale=2.0, latents=lat)
        img = PIPE._decode_latents(mod, z)
        feats = mod.image_encoder.get_hidden_states(pix.to(DEV)).cpu()
        z_ref = OS.ddim_sample(full_sd, _ocfg(mod), target, source, feats, 4, lat, steer_scale=2.0)
        img_ref = OS.latents_to_images(full_sd, _ocfg(mod), z_ref)
    assert img.device.type == "cpu" and img.shape == (2, 3, 128, 128)
    assert (z.cpu() - z_ref).abs().max().item() < 5e-2
    d = (img - img_ref).abs()
    assert d.max().item() < 3e-2 and d.mean().item() < 4e-3
    with pytest.raises(ValueError):
        PIPE._ddim_sample_batched(mod, target.to(DEV), source.to(DEV), pix[:1].to(DEV), 4, DEV)
    z_rng = PIPE._ddim_sample_batched(mod, target.to(DEV), source.to(DEV), pix.to(DEV), 2, DEV)
    assert z_rng.shape == (2, 4, 16, 16) and (z_rng[0] - z_rng[1]).abs().max().item() > 1e-3


def test_full_size_properties(full_sd):
    """BASELINE config 2 size (512x512, 50 steps, B=4, lambda=3): properties that need no oracle run."""
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    mod = _module(full_sd, 512, 4)
    pix = (torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV)
    lat1 = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(1234))
    lat = lat1.repeat(4, 1, 1, 1)
    tgt = torch.tensor([0.0, 1.0, 2.0, 3.0], device=DEV)
    src = torch.full((4,), 2.0, device=DEV)

    def sample(t, s, lam):
        with torch.no_grad():
            return PIPE._ddim_sample_ip(mod, t, s, pix, 50, DEV, steer_scale=lam, latents=lat)

    z3 = sample(tgt, src, 3.0)
    assert torch.isfinite(z3).all() and float(z3.abs().max()) <= 4.0 + 1e-6      # clamp(+-4) on x0
    assert torch.equal(z3, sample(tgt, src, 3.0))                                 # deterministic replay
    # target == source for sample 2 -> its delta tokens are exactly 0 -> lambda has no effect on it,
    # and with lambda = 0 every label yields the same image
    z0 = sample(tgt, src, 0.0)
    assert (z3[2] - z0[2]).abs().max().item() < 1e-5
    assert (z0 - z0[:1]).abs().max().item() < 1e-5
    assert (z3[0] - z3[3]).abs().max().item() > 1e-3                              # steering acts
    with torch.no_grad():
        img = PIPE._latents_to_images(mod, z3)
    assert img.shape == (4, 3, 512, 512) and float(img.min()) >= 0.0 and float(img.max()) <= 1.0


# ------------------------------------------------------------------------------------------------
# Round 2: the evidence VERDICT r1 asked for — golden conditioning through the PRODUCT classes on the device,
# BASELINE config 3's workload on one GPU (B = 13 plan == four padded shards), full-size kernels vs the oracle.
def test_golden_conditioning_through_product_classes(hip, golden_dir):
    """tests/golden/conditioning.npz = outputs of the imported reference classes (oracle/make_golden.py).  Here the
    product's ``conditioning.py`` classes produce them on the device THROUGH THE HIP KERNELS: the AOE path is fp32
    (interpolation kernel + fp32-weight rows kernel): 2e-5 (+1e-5 relative); resampler, basic projection and
    purifier run on the fp16-operand GEMM / attention kernels: 4e-3 (the tolerance of the golden cross-attention)."""
    from progressive_stable_diffusion_amd import conditioning as PC
    from progressive_stable_diffusion_amd import weights as W
    g = np.load(os.path.join(golden_dir, "conditioning.npz"))
    sd = W.init_state_dict(W.conditioning_shapes(), GI.SEED)

    def close(got, key, atol, rtol):
        ref = torch.from_numpy(g[key])
        err = (got.float().cpu() - ref).abs()
        assert got.shape == ref.shape and bool((err <= atol + rtol * ref.abs()).all()), (key, err.max().item())
        return err.max().item()

    aoe = PC.AdditiveOrdinalEmbedder(sd, DEV, be=hip)
    labels, source = torch.tensor(GI.LABELS, device=DEV), torch.tensor(GI.SOURCE, device=DEV)
    close(aoe(labels), "aoe_forward", 2e-5, 1e-5)
    close(aoe.get_negative_embedding(labels), "aoe_negative", 2e-5, 1e-5)
    close(aoe.get_ordinal_delta_embedding(source, labels), "aoe_delta", 2e-5, 1e-5)
    assert aoe.get_ordinal_delta_embedding(labels, labels).abs().max().item() == 0.0     # ordinal_embedder.py:254-255
    pur = PC.FeaturePurifier(sd, DEV, be=hip)
    e1 = close(pur(GI.purifier_image_tokens().to(DEV), aoe(torch.tensor(GI.PUR_SOURCE, device=DEV))), "pur_out", 4e-3, 4e-3)
    e2 = close(PC.ImageProjectionPlus(sd, DEV, be=hip)(GI.clip_hidden().to(DEV)), "plus_out", 4e-3, 4e-3)
    sd_b = W.init_state_dict(W.conditioning_shapes(projection_plus=False, purifier=False), GI.SEED)
    e3 = close(PC.ImageProjection(sd_b, DEV, be=hip)(GI.clip_embeds().to(DEV)), "basic_out", 4e-3, 4e-3)
    print(f"golden conditioning through HIP: purifier {e1:.2e} resampler {e2:.2e} basic {e3:.2e}")
    torch.cuda.synchronize()


@pytest.mark.parametrize("tiny", [False, True])
def test_clip_tower_matches_transformers(hip, tiny):
    """The CLIP vision tower on the HIP kernels (patch GEMM, LayerNorm-folded q|k|v and fc1 GEMMs, flash attention
    d = 64, quick-GELU epilogue) vs ``transformers.CLIPVisionModelWithProjection`` — the reference's own encoder
    (src/models/image_encoder.py:34-42) — in fp32 on the CPU with the same seeded weights: ViT-L/14 (24 layers, 257
    tokens) and a 2-layer tower.  Tolerance 1e-2 of the largest value (fp16 rows through 24 residual layers)."""
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    from progressive_stable_diffusion_amd import conditioning as PC
    from progressive_stable_diffusion_amd import weights as W
    cfg = dict(TINY_CLIP) if tiny else dict(W.CLIP_VIT_L14)
    sd = W.init_state_dict(W.clip_shapes(cfg), 5)
    hf = CLIPVisionModelWithProjection(CLIPVisionConfig(**cfg)).eval()
    pref = "image_encoder.image_encoder."
    missing, unexpected = hf.load_state_dict({k[len(pref):]: v for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith("position_ids") for k in missing), (missing, unexpected)
    px = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(6))
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.no_grad():
        ref = hf(pixel_values=px, output_hidden_states=True)
    enc = PC.ImageEncoder(sd, DEV, be=hip)
    assert enc.cfg["num_hidden_layers"] == cfg["num_hidden_layers"] and enc.hidden_size == cfg["hidden_size"]
    hs = enc.get_hidden_states(px.to(DEV))
    emb = enc(px.to(DEV))
    torch.cuda.synchronize()
    eh = (hs.cpu() - ref.hidden_states[-1]).abs().max().item() / ref.hidden_states[-1].abs().max().item()
    ee = (emb.cpu() - ref.image_embeds).abs().max().item() / ref.image_embeds.abs().max().item()
    print(f"clip tower tiny={tiny}: hidden_states[-1] rel {eh:.3e}, image_embeds rel {ee:.3e}")
    assert hs.shape == ref.hidden_states[-1].shape and eh < 1e-2 and ee < 1e-2
    # one structure image expanded over the batch (inference_pipeline_ip.py:377-385): encoded once, same values
    hs1 = enc.get_hidden_states(px[:1].to(DEV).expand(3, -1, -1, -1))
    assert hs1.shape[0] == 3 and torch.equal(hs1[0], hs1[2]) and (hs1[0] - hs[0]).abs().max().item() < 2e-3 * ref.hidden_states[-1].abs().max().item()


def test_config3_sweep_single_plan_and_four_shards(full_sd):
    """BASELINE config 3 on one GPU: the 13-label MES sweep ``linspace(0, 3, 13)``, source 0, ONE shared
    CPU-seeded latent — (i) as the single B = 13 plan the reference ``main()`` builds
    (inference_pipeline_ip.py:604-612,646-661) vs the oracle at 256x256 / 10 steps (labels 0, 0.25, 1.5, 2.75, 3); (ii) as four padded shards of
    4 through ``distributed.shard_labels`` (what 4 ranks would run) whose concatenation, padding dropped, matches
    the B = 13 result.  Tolerances: latents 5e-2 vs the oracle and between the two HIP plans (fp16 storage noise of either
    plan — other tilings / split-K orders at B = 13 and B = 4 — amplified 25x by the first DDIM update; measured
    4.0e-2 / 4.6e-2); frames 3e-2 max (measured 5.9e-3)."""
    from progressive_stable_diffusion_amd import distributed as D
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    mod = _module(full_sd, 256, 13)
    labels = PIPE._build_labels(13, 0.0, 3.0)
    assert labels[1].item() == 0.25 and labels[-1].item() == 3.0
    pix = torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1
    lat1 = D.shared_initial_latent(1234, 4, 32)
    with torch.no_grad():
        z13 = PIPE._ddim_sample_ip(mod, labels.to(DEV), torch.zeros(13, device=DEV), pix.to(DEV), 10, DEV,
                                   steer_scale=3.0, latents=lat1.repeat(13, 1, 1, 1))
        img13 = PIPE._latents_to_images(mod, z13)
        feats = mod.image_encoder.get_hidden_states(pix.to(DEV)).cpu()
        # the oracle is per-sample independent (as is the reference): five of the 13 labels keep its CPU time ~1 min
        sub = torch.tensor([0, 1, 6, 11, 12])
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        z_ref, tr = _oracle_sample("c3", full_sd, _ocfg(mod), labels[sub], torch.zeros(5), feats, 10,
                                   lat1.repeat(5, 1, 1, 1), steer_scale=3.0)
        # teacher-forced per-step check of the B = 13 plan (the bound that does not amplify): the oracle's x_t of the five
        # labels it ran go into their rows of the B = 13 batch (samples are independent: the other rows carry label 0's x_t)
        cond13 = PIPE._prepare_conditioning(mod, labels.to(DEV), torch.zeros(13, device=DEV), pix.to(DEV))
        PIPE._set_delta_scale_on_processors(mod, 3.0)
        ts = torch.linspace(999, 0, 10, dtype=torch.long, device=DEV)
        worst = 0.0
        for i, (eps_ref, _, x_t, _) in enumerate(tr):
            x13 = x_t[:1].repeat(13, 1, 1, 1)
            x13[sub] = x_t
            got = mod(x13.to(DEV), ts[i].expand(13), cond13).cpu()[sub]
            rel = (got - eps_ref).abs().max().item() / max(1.0, eps_ref.abs().max().item())
            worst = max(worst, rel)
            assert rel < 1e-2, (i, int(ts[i]), rel)
        print(f"config3 B=13 teacher-forced: worst per-step eps error {worst:.3e} of max|eps|")
        shards, frames = [], []
        for rank in range(4):
            loc, n_valid = D.shard_labels(labels, rank, 4, 4)
            assert loc.shape[0] == 4 and n_valid == (4 if rank < 3 else 1)
            z = PIPE._ddim_sample_ip(mod, loc.to(DEV), torch.zeros(4, device=DEV), pix.to(DEV), 10, DEV,
                                     steer_scale=3.0, latents=lat1.repeat(4, 1, 1, 1))
            shards.append(z)
            frames.append(PIPE._latents_to_images(mod, z))
    z16, img16 = torch.cat(shards), torch.cat(frames)
    assert torch.equal(z16[12], z16[13]) and torch.equal(z16[13], z16[15])          # padding repeats the last label
    e_oracle = (z13.cpu()[sub] - z_ref).abs().max().item()
    e_shard = (z16[:13] - z13).abs().max().item()
    e_img = (img16[:13] - img13).abs().max().item()
    print(f"config3 sweep: B=13 vs oracle {e_oracle:.3e}; 4x4 shards vs B=13 latents {e_shard:.3e} frames {e_img:.3e}")
    # Tolerance: ten DDIM steps amplify fp16-level differences of eps (the first update divides by sqrt(abar_999) =
    # 0.04).  Two VALID fp16 executions of the same sweep — the B=13 plan and the B=4 shards, which differ only in
    # tiling and in which fusions their shapes allow — end 4.4e-2 .. 5.0e-2 apart in the latents, and each is 4.0e-2 ..
    # 5.2e-2 from the fp32 oracle (both moved by ~1e-2 when the skip-concat GroupNorm moved into the conv); the decoded
    # frames stay within 5e-3.  So: 7e-2 on the latents, 3e-2 (measured 5e-3) on the frames.
    assert e_oracle < 7e-2 and e_shard < 7e-2 and e_img < 3e-2
    # label 0 == source 0: its delta tokens are exactly zero, lambda cannot act on it
    with torch.no_grad():
        z0 = PIPE._ddim_sample_ip(mod, labels.to(DEV), torch.zeros(13, device=DEV), pix.to(DEV), 10, DEV,
                                  steer_scale=0.0, latents=lat1.repeat(13, 1, 1, 1))
    assert (z0[0] - z13[0]).abs().max().item() < 1e-5 and (z13[12] - z13[0]).abs().max().item() > 1e-3


def test_config3_sweep_full_size_properties(full_sd):
    """The same sweep at BASELINE size (512x512, 50 steps) as the single B = 13 plan: properties only."""
    from progressive_stable_diffusion_amd import distributed as D
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    mod = _module(full_sd, 512, 13)
    labels = PIPE._build_labels(13, 0.0, 3.0).to(DEV)
    pix = (torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV)
    lat = D.shared_initial_latent(1234, 4, 64).repeat(13, 1, 1, 1)
    with torch.no_grad():
        z = PIPE._ddim_sample_ip(mod, labels, torch.zeros(13, device=DEV), pix, 50, DEV, steer_scale=3.0, latents=lat)
        z_again = PIPE._ddim_sample_ip(mod, labels, torch.zeros(13, device=DEV), pix, 50, DEV, steer_scale=3.0, latents=lat)
        z_l0 = PIPE._ddim_sample_ip(mod, labels, torch.zeros(13, device=DEV), pix, 50, DEV, steer_scale=0.0, latents=lat)
        img = PIPE._latents_to_images(mod, z)
    assert torch.isfinite(z).all() and float(z.abs().max()) <= 4.0 + 1e-6 and torch.equal(z, z_again)
    assert (z_l0 - z_l0[:1]).abs().max().item() < 1e-5            # lambda = 0: every label yields the same image
    assert (z[0] - z_l0[0]).abs().max().item() < 1e-5             # target == source: steering is a no-op
    d = [(z[i + 1] - z[i]).abs().mean().item() for i in range(12)]
    assert min(d) > 0.0                                           # every step of the progression moves the image
    assert img.shape == (13, 3, 512, 512) and float(img.min()) >= 0.0 and float(img.max()) <= 1.0


def test_vae_decode_512_matches_oracle(hip, full_sd):
    """VAE decode at BASELINE size (64x64 latent -> 512x512, B = 1) vs the oracle: the 512x512 / 256x256 kernels
    (128-channel register-staged convs, upsample gathers) get parity, not only a range check."""
    from progressive_stable_diffusion_amd.engine import VaeDecoderPlan
    plan = VaeDecoderPlan(hip, full_sd, 1, 64, latent_scale=0.18215)
    z = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(8)) * 0.18215 * 1.5
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.no_grad():
        ref = ((vae_decode(full_sd, z / 0.18215).clamp(-1, 1) + 1) / 2).clamp(0, 1)
    hip.copy_(plan.z_in, z.to(DEV))
    plan.run()
    hip.synchronize()
    d = (plan.img_out.cpu() - ref).abs()
    print(f"vae decode 512: max {d.max():.3e} mean {d.mean():.3e}")
    assert d.max().item() < 3e-2 and d.mean().item() < 2e-3


@pytest.mark.parametrize("bench_plan", [False, True])
def test_unet_call_512_matches_oracle(hip, full_sd, bench_plan, monkeypatch):
    """One eps call at BASELINE size (64x64 latent, B = 1, lambda = 3): the 64x64-level kernels (halo conv W = 64,
    flash d = 40 at 4096 keys) against the oracle.  ``bench_plan``: with the fusions the B = 4 bench plan uses at the
    64x64 sites forced on at B = 1 too — attn2 in one kernel, the transformer blocks' head (GroupNorm + proj_in + norm1 +
    q|k|v, csrc/tf_head.hip) and tail (norm3 + GEGLU + FF-out + proj_out, csrc/ffn_block.hip) in one launch each."""
    from progressive_stable_diffusion_amd import engine as E
    if bench_plan:
        monkeypatch.setattr(E, "FFN_MIN_BLOCKS", 1)
        monkeypatch.setattr(E, "A2_MIN_TILES", 1)
    plan = E.UNetPlan(hip, full_sd, 1, 64)
    names = [getattr(fn, "__name__", "") for fn, _, _ in plan.ops]
    assert (names.count("ffn_block"), names.count("tf_head")) == ((5, 5) if bench_plan else (0, 0))
    g = torch.Generator().manual_seed(17)
    x = torch.randn(1, 4, 64, 64, generator=g)
    cond = torch.randn(1, 48, 768, generator=g) * 0.5
    t = torch.tensor([650])
    if "u512" not in _ORACLE_RUNS:
        torch.set_num_threads(min(os.cpu_count() or 1, 64))
        with torch.no_grad():
            _ORACLE_RUNS["u512"] = unet_forward(full_sd, x, t, cond, delta_scale=3.0)
    ref = _ORACLE_RUNS["u512"]
    got = plan.forward(x.to(DEV), t.to(DEV), cond.to(DEV), lam=3.0)
    hip.synchronize()
    err = (got.cpu() - ref).abs().max().item()
    print(f"unet 512 call (bench plan fusions {bench_plan}): max err {err:.3e} (max |eps| {ref.abs().max():.3f})")
    assert err < 1e-2 * max(1.0, ref.abs().max().item()), err


def test_stochastic_sampler_on_device(full_sd):
    """eta > 0 (inference_pipeline_ip.py:457-468) through the HIP engine with injected per-step noise vs the oracle."""
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    mod = _module(full_sd, 128, 2)
    target, source = torch.tensor([3.0, 0.5]), torch.tensor([1.0, 2.0])
    pix = torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(5)) * 2 - 1
    lat = torch.randn(2, 4, 16, 16, generator=torch.Generator().manual_seed(9))
    noise = torch.randn(3, 2, 4, 16, 16, generator=torch.Generator().manual_seed(10))
    with torch.no_grad():
        z = PIPE._ddim_sample_ip(mod, target.to(DEV), source.to(DEV), pix.to(DEV), 4, DEV, eta=0.5, steer_scale=2.0,
                                 latents=lat, step_noise=noise)
        feats = mod.image_encoder.get_hidden_states(pix.to(DEV)).cpu()
        z_ref = OS.ddim_sample(full_sd, _ocfg(mod), target, source, feats, 4, lat, eta=0.5, steer_scale=2.0,
                               step_noise=noise)
    assert (z.cpu() - z_ref).abs().max().item() < 5e-2


def test_module_cond_cache_on_device(full_sd):
    """ADVICE r1 (high) on the real backend: a sampler run between two module() calls must not leave stale K/V."""
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    mod = _module(full_sd, 64, 2)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 8, 8, generator=g).to(DEV)
    t = torch.tensor([500, 20], device=DEV)
    cA = (torch.randn(2, 48, 768, generator=g) * 0.5).to(DEV)
    PIPE._set_delta_scale_on_processors(mod, 1.5)
    with torch.no_grad():
        eA = mod(x, t, cA).clone()
        pix = torch.rand(1, 3, 224, 224, generator=g).to(DEV)
        PIPE._ddim_sample_ip(mod, torch.tensor([3.0, 1.0], device=DEV), torch.zeros(2, device=DEV), pix, 2, DEV,
                             steer_scale=1.5)
        assert torch.equal(mod(x, t, cA), eA)
        assert torch.equal(mod(x, torch.tensor(500), cA)[0], mod(x, torch.tensor([500, 500], device=DEV), cA)[0])


def test_main_cli_end_to_end(full_sd, tmp_path):
    """``main()`` as the reference CLI runs it (inference_pipeline_ip.py:566-669): YAML config + tensors-only
    checkpoint (tiny CLIP tower inside, geometry read from its tensors) + structure image -> PNG sequence + grid."""
    import numpy as np
    import yaml
    from PIL import Image
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    from progressive_stable_diffusion_amd import weights as W
    from progressive_stable_diffusion_amd.config import default_config
    shapes = dict(W.unet_shapes())
    shapes.update(W.vae_shapes(encoder=False))
    shapes.update(W.conditioning_shapes(clip_hidden=TINY_CLIP["hidden_size"], clip_proj=TINY_CLIP["projection_dim"]))
    sd = W.init_state_dict(shapes, 0, gates=GI.GATES)
    sd.update(W.init_state_dict(W.clip_shapes(TINY_CLIP), 3))
    torch.save({"state_dict": sd}, tmp_path / "last.ckpt")

    def plain(o):
        return {k: plain(v) for k, v in o.items()} if isinstance(o, dict) else ([plain(v) for v in o] if isinstance(o, list) else o)
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(plain(default_config(**{"dataset.image_size": 64}))))
    Image.fromarray((np.random.RandomState(0).rand(70, 90, 3) * 255).astype("uint8")).save(tmp_path / "s.png")
    out = tmp_path / "out"
    PIPE.main(["--checkpoint", str(tmp_path / "last.ckpt"), "--config", str(tmp_path / "cfg.yaml"),
               "--structure-image", str(tmp_path / "s.png"), "--output-dir", str(out), "--mes-steps", "3",
               "--sampling-steps", "2", "--steer-scale", "2.0", "--seed", "7", "--device", "cuda"])
    names = sorted(os.listdir(out))
    assert names == ["mes_0.00_00.png", "mes_1.50_01.png", "mes_3.00_02.png", "progression_grid.png",
                     "structure_reference.png"]
    a = np.asarray(Image.open(out / "mes_0.00_00.png"))
    b = np.asarray(Image.open(out / "mes_3.00_02.png"))
    assert a.shape == (64, 64, 3) and np.abs(a.astype(int) - b.astype(int)).max() > 0


@pytest.mark.parametrize("image,batch", [(128, 2), (256, 1)])
def test_vae_encode_matches_oracle(hip, image, batch):
    """``SDVAE.encode`` on the HIP kernels (asymmetric-pad stride-2 convs, mid attention d = 512, quant_conv composed
    into conv_out) + the reparameterised sample with injected noise vs ``oracle.sd_vae`` (SURVEY.md §8 a15 / north_star
    'VAE encode').  Tolerance: moments 2e-2 of their max (fp16 storage through 24 convs), sample 2e-2."""
    from oracle.sd_vae import vae_encode_moments, vae_encode_sample
    from progressive_stable_diffusion_amd import weights as W
    from progressive_stable_diffusion_amd.engine import VaeEncoderPlan
    sd = W.init_state_dict(W.vae_shapes(decoder=False), 0)
    s = image // 8
    plan = VaeEncoderPlan(hip, sd, batch, s)
    g = torch.Generator().manual_seed(21)
    x = torch.rand(batch, 3, image, image, generator=g) * 2 - 1
    noise = torch.randn(batch, 4, s, s, generator=g)
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    with torch.no_grad():
        mean, logvar = vae_encode_moments(sd, x)
        ref = vae_encode_sample(sd, x, noise) * 0.18215
    hip.copy_(plan.img_in, x.to(DEV))
    plan.run()
    out = hip.zeros((batch, 4, s, s), F32)
    hip.gaussian_sample(plan.mean, plan.logvar, noise.to(DEV), out, 0.18215)
    hip.synchronize()
    em = (plan.mean.cpu() - mean).abs().max().item() / max(1.0, mean.abs().max().item())
    el = (plan.logvar.cpu() - logvar).abs().max().item() / max(1.0, logvar.abs().max().item())
    es = (out.cpu() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    print(f"vae encode {image}: mean {em:.3e} logvar {el:.3e} sample {es:.3e}")
    assert em < 2e-2 and el < 2e-2 and es < 2e-2


def test_module_vae_encode_on_device(full_sd):
    """``module.vae.encode(x).latent_dist.sample() * latent_scale`` (diffusion_module_ip.py:410-411) and the
    encode -> decode round trip through both plans (property: shapes, ranges, determinism of mode())."""
    from progressive_stable_diffusion_amd import weights as W
    sd = dict(full_sd)
    sd.update(W.init_state_dict(W.vae_shapes(decoder=False), 0))
    mod = _module(sd, 128, 2, clip_config=None)
    x = (torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(4)) * 2 - 1).to(DEV)
    dist = mod.vae.encode(x).latent_dist
    z = dist.sample() * mod.diff_cfg.latent_scale
    assert z.shape == (2, 4, 16, 16) and torch.isfinite(z).all()
    assert torch.equal(mod.vae.encode(x).latent_dist.mode(), dist.mean)
    img = mod.vae.decode(dist.mode()).sample
    assert img.shape == (2, 3, 128, 128) and float(img.min()) >= -1.0 and float(img.max()) <= 1.0


def test_one_graph_serves_a_lambda_sweep(full_sd):
    """lambda (and the CFG scale) are device-side parameters of the captured step (the reference
    reads ``delta_scale`` per call, attention_processor_routing_gates.py:160): after runs at three lambdas the loop
    holds ONE graph, every replay equals the eager run of the same lambda bit for bit, lambda = 0 equals the
    two-pathway result even with NaN planted in the delta tokens' projections."""
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    mod = _module(full_sd, 256, 2)
    target, source = torch.tensor([3.0, 0.5], device=DEV), torch.tensor([0.0, 2.0], device=DEV)
    pix = (torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV)
    lat = torch.randn(2, 4, 32, 32, generator=torch.Generator().manual_seed(77))
    loop = mod.ddim_loop(2, 32)
    outs = {}
    with torch.no_grad():
        for lam in (3.0, 0.0, 1.25, 3.0):
            z = PIPE._ddim_sample_ip(mod, target, source, pix, 4, DEV, steer_scale=lam, latents=lat)
            if lam in outs:
                assert torch.equal(z, outs[lam])
            outs[lam] = z
            z_e = PIPE._ddim_sample_ip(mod, target, source, pix, 4, DEV, steer_scale=lam, latents=lat, use_graph=False)
            assert torch.equal(z, z_e), lam
    assert len(loop.graphs) == 1
    assert (outs[3.0] - outs[0.0]).abs().max().item() > 1e-3 and (outs[1.25] - outs[0.0]).abs().max().item() > 1e-3
    # NaN in the delta rows of every K/V cache: lambda = 0 must not read them.  The NaNs are written by torch ops on
    # torch's CURRENT stream, the loop runs on the backend's stream: ``DdimLoop.sample`` orders the two both ways (the
    # hand-offs every product path makes); r2's red run drove prepare / run by hand without them and compared a tensor
    # the backend stream had not finished writing.
    with torch.no_grad():
        PIPE._ddim_sample_ip(mod, target, source, pix, 1, DEV, steer_scale=0.0, latents=lat)     # projects the cond
        for site, _ in loop.u.sites:
            loop.u.kv[site][0][:, :, 32:, :] = float("nan")
        loop.u._a2_dirty = True
        z_nan = loop.sample(lat.to(DEV), torch.linspace(999, 0, 4, dtype=torch.long), mod.alphas_cumprod, 0.0)
        torch.cuda.synchronize()
        assert all(bool(torch.isnan(loop.u.kv[site][0][:, :, 32:, :]).all()) for site, _ in loop.u.sites), \
            "the NaNs must still be in the delta rows after the run"
    diff = (z_nan - outs[0.0]).abs()
    bad = int((~(diff == 0)).sum())          # NaN != 0 counts as different
    assert bad == 0, (f"{bad} of {diff.numel()} latents differ from the clean lambda=0 run: NaNs in z_nan "
                      f"{int(torch.isnan(z_nan).sum())}, max |diff| {float(torch.nan_to_num(diff).max()):.3e}, first at "
                      f"{[tuple(i.tolist()) for i in (diff != 0).nonzero()[:4]]}")
    # ... and after a lambda = 3 run on the poisoned caches (which DOES read the delta rows -> NaN) the loop recovers:
    # a clean re-projection + lambda = 0 reproduces the clean result (no stale per-step state survives a run)
    with torch.no_grad():
        z_bad = loop.sample(lat.to(DEV), torch.linspace(999, 0, 4, dtype=torch.long), mod.alphas_cumprod, 3.0)
        assert not torch.equal(z_bad, outs[3.0]), "lambda != 0 reads the (poisoned) delta rows"
        z_again = PIPE._ddim_sample_ip(mod, target, source, pix, 4, DEV, steer_scale=0.0, latents=lat)
    assert torch.equal(z_again, outs[0.0])


def test_generation_drivers_and_frame_sink_on_device(full_sd, tmp_path):
    """§8f-2 on the HIP backend: ``generate_all`` (evaluation) and ``augment_dataset`` (data augmentation) over a tiny
    class-folder dataset at 128x128 / 3 steps — one static plan for the ragged batches, uint8 pack + pinned-buffer
    D2H on the copy stream + writer pool, files identical to ``_tensor_to_bmp`` of the frames the sampler produced,
    resume by existing file; and the drivers' frames equal a direct ``_ddim_sample_batched`` call on the same seed."""
    import numpy as np
    from PIL import Image
    from progressive_stable_diffusion_amd import generation as G
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    mod = _module(full_sd, 128, 6)
    rs = np.random.RandomState(0)
    for root in (tmp_path / "eval", tmp_path / "aug" / "train"):
        for cls, n in enumerate((1, 1, 0, 1)):
            (root / str(cls)).mkdir(parents=True, exist_ok=True)
            for i in range(n):
                Image.fromarray((rs.rand(60, 70, 3) * 255).astype("uint8")).save(root / str(cls) / f"img{cls}_{i}.bmp")
    jobs = G._collect_jobs([tmp_path / "eval"])
    res = G.generate_all(mod, jobs, mod.cfg, DEV, batch_images=2, sampling_steps=3, steer_scale=2.0, seed=5)
    assert {c: v.shape[0] for c, v in res.items()} == {0: 2, 1: 2, 2: 3, 3: 2} and len(mod._unets) == 1
    # the first batch again, by hand, on the same seed: same frames
    PIPE._set_seed(5)
    j0 = sorted(jobs, key=lambda j: (str(j.source_path), j.target_label))[:6]
    structs = [G._load_structure_image(j.source_path, DEV, 128) for j in j0]
    with torch.no_grad():
        z = PIPE._ddim_sample_batched(mod, torch.tensor([float(j.target_label) for j in j0], device=DEV),
                                      torch.tensor([float(j.source_label) for j in j0], device=DEV),
                                      torch.cat(structs), 3, DEV, steer_scale=2.0)
        fr = PIPE._latents_to_images(mod, z).cpu()
    first = {c: 0 for c in range(4)}
    for k, j in enumerate(j0):
        assert torch.equal(res[j.target_label][first[j.target_label]], fr[k])
        first[j.target_label] += 1
    dst = tmp_path / "dst"
    counts = G.augment_dataset(mod, tmp_path / "aug", dst, DEV, batch_images=2, sampling_steps=2, steer_scale=1.0, save_workers=4)
    assert counts == {0: 2, 1: 2, 2: 3, 3: 2} and len(list(dst.rglob("*_generated.bmp"))) == 9
    a = np.asarray(Image.open(dst / "1" / "img0_0_generated.bmp"))
    assert a.shape == (128, 128, 3) and a.std() > 1.0
    assert sum(G.augment_dataset(mod, tmp_path / "aug", dst, DEV, batch_images=2, sampling_steps=2).values()) == 0   # resume
    # sink bytes == mul(255).to(uint8) of the frames
    sink = G.FrameSink(mod.be, 6, 128, 128, workers=2)
    sink.submit(fr.to(DEV), [tmp_path / "s" / f"{k}.png" for k in range(6)])
    sink.close()
    for k in range(6):
        got = torch.from_numpy(np.asarray(Image.open(tmp_path / "s" / f"{k}.png")).copy()).permute(2, 0, 1)
        assert torch.equal(got, fr[k].mul(255).to(torch.uint8))


def test_training_step_forward_on_device(full_sd):
    """BASELINE config 4's forward half at 128x128, B = 2 on the HIP kernels (VAE encode, posterior sample, q_sample,
    conditioning with CFG image dropout, UNet eps, eps-MSE x Min-SNR) vs ``oracle.training`` with every draw injected.
    Tolerance 2 % of the loss (fp16 storage through encoder + UNet).  The backward pass / AdamW are not built."""
    from oracle import training as OT
    from progressive_stable_diffusion_amd import weights as W
    sd = dict(full_sd)
    sd.update(W.init_state_dict(W.vae_shapes(decoder=False), 0))
    mod = _module(sd, 128, 2)
    g = torch.Generator().manual_seed(31)
    images = torch.rand(2, 3, 128, 128, generator=g) * 2 - 1
    labels = torch.tensor([1.0, 3.0])
    pix = torch.randn(2, 3, 224, 224, generator=g)
    t = torch.tensor([700, 35])
    noise, lat_noise = torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 4, 16, 16, generator=g)
    drop = torch.tensor([False, True])
    with torch.no_grad():
        loss = mod.training_step((images.to(DEV), labels.to(DEV), pix.to(DEV)), 0, noise=noise, t=t, drop_mask=drop,
                                 latent_noise=lat_noise, is_training=False)
        feats = mod.image_encoder.get_hidden_states(pix.to(DEV)).cpu()
        ref, _ = OT.training_loss(sd, _ocfg(mod), images, labels, feats, t, noise, lat_noise, drop)
    print(f"training-step forward: loss {loss.item():.6f} vs oracle {ref.item():.6f}")
    assert abs(loss.item() - ref.item()) < 2e-2 * max(1.0, abs(ref.item()))
    x0 = torch.randn(2, 4, 16, 16, generator=g)
    assert torch.allclose(mod._q_sample(x0.to(DEV), t, noise).cpu(), OT.q_sample(mod.alphas_cumprod.cpu(), x0, t, noise),
                          rtol=1e-6, atol=1e-6)          # fp32; the device square roots may differ in the last place
    l2 = mod.training_step((images.to(DEV), labels.to(DEV), pix.to(DEV)))            # device RNG everywhere: finite
    assert torch.isfinite(l2) and l2.item() > 0.0


def test_config5_geometry_768_properties(full_sd):
    """BASELINE config 5's GEOMETRY (768x768, B = 2: 96x96 / 48x48 / 24x24 / 12x12 maps, 9216-token self-attention)
    through the fp16 kernels: finite, clamped, deterministic, steering acts, frames in range.  The fp8-e4m3 attention
    and bf16 convolutions config 5 names are NOT built; 96-pixel rows run on the implicit GEMM (the halo
    conv covers 16 / 32 / 64-pixel rows)."""
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    mod = _module(full_sd, 768, 2)
    pix = (torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV)
    lat = torch.randn(1, 4, 96, 96, generator=torch.Generator().manual_seed(5)).repeat(2, 1, 1, 1)
    tgt, src = torch.tensor([3.0, 2.0], device=DEV), torch.full((2,), 2.0, device=DEV)
    with torch.no_grad():
        z = PIPE._ddim_sample_ip(mod, tgt, src, pix, 6, DEV, steer_scale=3.0, latents=lat)
        z2 = PIPE._ddim_sample_ip(mod, tgt, src, pix, 6, DEV, steer_scale=3.0, latents=lat)
        z0 = PIPE._ddim_sample_ip(mod, tgt, src, pix, 6, DEV, steer_scale=0.0, latents=lat)
        img = PIPE._latents_to_images(mod, z)
    assert z.shape == (2, 4, 96, 96) and torch.isfinite(z).all() and float(z.abs().max()) <= 4.0 + 1e-6
    assert torch.equal(z, z2)
    assert (z[1] - z0[1]).abs().max().item() < 1e-5 and (z[0] - z0[0]).abs().max().item() > 1e-3
    assert img.shape == (2, 3, 768, 768) and float(img.min()) >= 0.0 and float(img.max()) <= 1.0
