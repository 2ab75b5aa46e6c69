"""CPU checks of the HOST logic: the engine's plans (op order, buffer reuse, weight packing, GEGLU
row interleave, skip-concat order, split-K bookkeeping, time-row tables, DDIM coefficient table,
module/pipeline protocol) executed through the TEST-ONLY torch backend and compared with the
oracle.  No HIP kernel runs here; the same comparisons run on the GPU in test_gpu_parity.py.
"""
import pytest
import torch

from oracle import sampler as OS
from oracle.sd_unet import unet_forward
from oracle.sd_vae import vae_decode
from progressive_stable_diffusion_amd import engine as E
from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
from progressive_stable_diffusion_amd import weights as W
from progressive_stable_diffusion_amd.config import default_config
from progressive_stable_diffusion_amd.diffusion_module_ip import DiffusionModuleWithIP
from tests.torch_backend import TorchRefBackend

GATES = {"anatomy": (0.1, 0.9), "disease": (0.9, 0.1), "both": (0.5, 0.5)}
TINY_CLIP = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                 image_size=224, patch_size=14, projection_dim=32)


@pytest.fixture(scope="module")
def full_sd():
    """One seeded state dict for every test of this file (UNet incl. processors, VAE decoder,
    conditioning modules sized for the tiny CLIP tower); baseline mode ignores the extra keys."""
    shapes = dict(W.unet_shapes())
    shapes.update(W.vae_shapes(encoder=False))
    shapes.update(W.conditioning_shapes(clip_hidden=TINY_CLIP["hidden_size"],
                                        clip_proj=TINY_CLIP["projection_dim"]))
    return W.init_state_dict(shapes, 0, gates=GATES, warm_start_dis=False)


@pytest.fixture(scope="module")
def unet_sd(full_sd):
    return full_sd


def test_geglu_interleave_roundtrip():
    w = torch.arange(512 * 3, dtype=torch.float32).reshape(512, 3)
    b = torch.arange(512, dtype=torch.float32)
    wp, bp = E.geglu_interleave(w, b)
    # tile 0, wave 0: rows 0..31 are hidden columns 0..31, rows 32..63 the matching gates
    assert bp[:32].tolist() == list(range(32))
    assert bp[32:64].tolist() == list(range(256, 288))
    assert bp[64:96].tolist() == list(range(32, 64))
    assert sorted(bp.tolist()) == list(range(512))
    assert torch.equal(wp[:, 0] / 3, bp)


@pytest.fixture(autouse=True)
def _fused_attn2_everywhere(monkeypatch):
    """The folded attn2 path (one kernel per block) is opt-in; switch it on for every eligible site so that the
    folded algebra (W_q K^T, V W_o^T, gates and lambda in the conditioning) stays under test on the CPU."""
    monkeypatch.setattr(E, "FUSED_ATTN2", True)
    monkeypatch.setattr(E, "A2_MIN_TILES", 1)


def test_splitk_heuristic():
    # policy measured with scripts/op_bench.py on MI355X (profiles/r01_*_op_bench.txt)
    from progressive_stable_diffusion_amd import lib as L
    assert E.choose_tiling(16384, 2560, 320, 128, geglu=True) == (64, 1, L.TUNE_NODMA | L.TUNE_SHALLOW)
    assert E.choose_tiling(4096, 5120, 640, 128, geglu=True) == (128, 1, L.TUNE_PERSIST)   # 1280 tiles, 5 per CU
    assert E.choose_tiling(1024, 10240, 1280, 128, geglu=True) == (128, 1, 0)
    assert E.choose_tiling(16384, 320, 320, 160) == (64, 1, L.TUNE_NODMA)   # short-K linear: register kernel
    assert E.choose_tiling(16384, 960, 320, 160) == (128, 1, L.TUNE_PERSIST)  # qkv: 768 tiles, ring over 3 tiles
    assert E.choose_tiling(4096, 1920, 640, 160) == (128, 1, L.TUNE_PERSIST)
    assert E.choose_tiling(16384, 320, 5760, 160) == (128, 1, 0)           # 256 tiles: one per CU, DMA ring
    assert E.choose_tiling(4096, 640, 5760, 160) == (128, 2, 0)
    assert E.choose_tiling(4096, 640, 1280, 160) == (128, 1, 0)            # 20 K tiles: splitting costs more
    assert E.choose_tiling(1024, 1280, 11520, 160) == (128, 4, 0)
    assert E.choose_tiling(256, 1280, 11520, 160) == (128, 16, 0)           # 8x8 level: 16 tiles x 16 splits
    assert E.choose_tiling(1024, 1280, 1280, 160) == (128, 2, 0)
    assert E.choose_tiling(1024, 1280, 1280, 160, residual=False) == (128, 1, 0)
    assert E.choose_tiling(16384, 320, 320, 160, residual=False) == (128, 1, 0)   # q projection: DMA ring
    assert E.choose_tiling(1048576, 128, 1152, 128) == (128, 1, L.TUNE_NODMA)      # VAE 512x512, 128 channels
    assert E.choose_tiling(262144, 256, 2304, 128) == (128, 1, L.TUNE_PERSIST)    # VAE 256x256: 16 tiles per CU
    assert E.choose_splitk(64, 1280, 768, 160) == 1
    for m, n, k in ((256, 1280, 23040), (1024, 640, 5760), (4096, 640, 5760), (1024, 1280, 1280)):
        tm, s, _ = E.choose_tiling(m, n, k, 160)
        assert tm in (64, 128) and 1 <= s <= 32 and (k // 64) // s >= 4


@pytest.mark.parametrize("lam,fold", [(0.0, True), (3.0, True), (3.0, False)])
def test_unet_plan_matches_oracle(unet_sd, lam, fold, monkeypatch):
    """fold: LayerNorm folded into the consuming linears (48 launches less per step) or as its own op."""
    monkeypatch.setattr(E, "LN_FOLD", fold)
    torch.manual_seed(1)
    b, s = 2, 8
    plan = E.UNetPlan(TorchRefBackend(), unet_sd, b, s)
    x, cond = torch.randn(b, 4, s, s), torch.randn(b, 48, 768) * 0.5
    t = torch.tensor([999, 333])
    with torch.no_grad():
        ref = unet_forward(unet_sd, x, t, cond, delta_scale=lam)
        got = plan.forward(x, t, cond, lam=lam)
    assert (ref - got).abs().max().item() < 6e-3 * max(1.0, ref.abs().max().item())
    # every pooled buffer is handed back exactly once: no leak of plan-time buffers
    n_ln = sum(1 for fn, _, _ in plan.ops if getattr(fn, "__name__", "") == "layernorm")
    assert n_ln == (len(plan.a2) if fold else 48) and len(plan.ops) > 200     # fused-attn2 sites keep LayerNorm 2
    # on these small maps every conv runs split-K and its finish kernel writes the next GroupNorm (DADD_EPI_GNAPPLY): only
    # the GroupNorms over a skip-concat, behind a non-split linear (transformer -> ResNet) and conv_norm_out stay launches
    n_gn = sum(1 for fn, _, _ in plan.ops if getattr(fn, "__name__", "") == "groupnorm")
    monkeypatch.setattr(E, "FINISH_GN_APPLY", False)
    n_gn_plain = sum(1 for fn, _, _ in E.UNetPlan(TorchRefBackend(), unet_sd, b, s).ops if getattr(fn, "__name__", "") == "groupnorm")
    monkeypatch.setattr(E, "FINISH_GN_APPLY", True)
    assert n_gn_plain == 61 and n_gn <= 30, (n_gn_plain, n_gn)
    monkeypatch.setattr(E, "LN_FOLD", "auto")                  # the shipped policy: fold on 64-row tiles only
    n_auto = sum(1 for fn, _, _ in E.UNetPlan(TorchRefBackend(), unet_sd, b, s).ops if getattr(fn, "__name__", "") == "layernorm")
    assert 0 <= n_auto <= 48


@pytest.mark.parametrize("lam", [0.0, 3.0])
def test_unet_plan_layernorm_statistics_from_producer(unet_sd, lam, monkeypatch):
    """The other half of the "auto" policy: where the consumer does not sum the rows itself (128-row tiles on the GPU;
    forced everywhere here), the GEMM that PRODUCES the hidden states writes the LayerNorm row partials
    (DADD_EPI_LNSTAT / attn2_fused ln_stats_out) and the folded consumer reads them: only LayerNorm 2 in front of a
    fused attn2 kernel is left as a launch, and the result is the oracle's."""
    from progressive_stable_diffusion_amd import lib as L
    monkeypatch.setattr(E, "LN_FOLD", "auto")
    monkeypatch.setattr(E, "fold_here", lambda *a, **k: False)
    monkeypatch.setattr(E, "LN_STATS_MAX_PARTS", 1000)     # (the small maps of this test run on 64-column tiles: 40 parts)
    torch.manual_seed(1)
    b, s = 2, 8
    plan = E.UNetPlan(TorchRefBackend(), unet_sd, b, s)
    x, cond = torch.randn(b, 4, s, s), torch.randn(b, 48, 768) * 0.5
    t = torch.tensor([999, 333])
    with torch.no_grad():
        ref = unet_forward(unet_sd, x, t, cond, delta_scale=lam)
        got = plan.forward(x, t, cond, lam=lam)
    assert (ref - got).abs().max().item() < 6e-3 * max(1.0, ref.abs().max().item())
    names = [getattr(fn, "__name__", "") for fn, _, _ in plan.ops]
    assert names.count("layernorm") == len(plan.a2)
    n_out = sum(1 for fn, _, k in plan.ops if k.get("ln_stats_out") is not None)
    n_in = sum(1 for fn, _, k in plan.ops if k.get("ln_stats_in") is not None)
    assert n_in == 48 - len(plan.a2) and n_out == n_in
    for fn, _, k in plan.ops:       # flags and buffers go together
        if getattr(fn, "__name__", "") == "igemm":
            assert bool(k["flags"] & L.EPI_LNSTAT) == (k.get("ln_stats_out") is not None)
            assert k.get("ln_stats_in") is None or (k["flags"] & L.EPI_LNFOLD)
    monkeypatch.setattr(E, "LN_STATS_FROM_PRODUCER", False)     # switch off: LayerNorm launches are back
    plan2 = E.UNetPlan(TorchRefBackend(), unet_sd, b, s)
    assert sum(1 for fn, _, _ in plan2.ops if getattr(fn, "__name__", "") == "layernorm") == 48


def test_unet_plan_fused_attn2_with_norm2_folded(unet_sd, monkeypatch):
    """Fused attn2 sites (maps of >= 128 tokens): norm2 is folded into the score GEMM of attn2_fused — the hidden state goes
    in un-normalised with the row partials of the attn1 out-projection, mcat carries gamma, c1 / d come from
    prepare_attn2 — and the kernel writes the partials for norm3.  No LayerNorm launch is left at those sites; lambda
    changes re-fold the conditioning; the result is the oracle's."""
    monkeypatch.setattr(E, "LN_FOLD", "auto")
    monkeypatch.setattr(E, "fold_here", lambda *a, **k: False)
    monkeypatch.setattr(E, "LN_STATS_MAX_PARTS", 1000)
    monkeypatch.setattr(E, "A2_MIN_TILES", 1)
    torch.manual_seed(4)
    b, s = 1, 16                       # 256 / 64 tokens at the two upper levels: the 320-channel sites fuse
    plan = E.UNetPlan(TorchRefBackend(), unet_sd, b, s)
    assert len(plan.a2) >= 1 and all(st["fold"] for st in plan.a2.values())
    names = [getattr(fn, "__name__", "") for fn, _, _ in plan.ops]
    assert names.count("layernorm") == 0
    fused = [k for fn, _, k in plan.ops if getattr(fn, "__name__", "") == "attn2_fused"]
    assert len(fused) == len(plan.a2) and all(k.get("ln_stats_in") is not None and k.get("ln_stats_out") is not None for k in fused)
    x, cond = torch.randn(b, 4, s, s), torch.randn(b, 48, 768) * 0.5
    t = torch.tensor([700])
    for lam in (3.0, 0.0):
        with torch.no_grad():
            ref = unet_forward(unet_sd, x, t, cond, delta_scale=lam)
            got = plan.forward(x, t, cond, lam=lam)
        assert (ref - got).abs().max().item() < 6e-3 * max(1.0, ref.abs().max().item()), lam


def test_unet_plan_groupnorm_inside_conv(unet_sd, monkeypatch):
    """DADD_PRE_GN: where a 3x3 conv of a 16/32/64-wide map reads a single source whose producer wrote the GroupNorm
    partials, the conv applies norm + SiLU itself (no groupnorm op, no normalised copy); switching the policy off
    brings the groupnorm ops back and both plans give the oracle's result."""
    from progressive_stable_diffusion_amd import lib as L
    torch.manual_seed(6)
    b, s = 1, 16
    x, cond = torch.randn(b, 4, s, s), torch.randn(b, 48, 768) * 0.5
    t = torch.tensor([400])
    with torch.no_grad():
        ref = unet_forward(unet_sd, x, t, cond, delta_scale=3.0)
    counts = {}
    monkeypatch.setattr(E, "GN_FUSED_MAX_BYTES", 0)          # (small maps: keep the partials path instead of the LDS GroupNorm)
    for on in (True, False):
        monkeypatch.setattr(E, "GN_IN_CONV", on)
        plan = E.UNetPlan(TorchRefBackend(), unet_sd, b, s)
        names = [getattr(fn, "__name__", "") for fn, _, _ in plan.ops]
        n_in = sum(1 for fn, _, k in plan.ops if k.get("gn_in") is not None)
        counts[on] = (names.count("groupnorm"), n_in)
        for fn, _, k in plan.ops:
            if k.get("gn_in") is not None:
                assert k["flags"] & L.PRE_GN and k["flags"] & L.PRE_GN_SILU and k["taps"] == 9
                assert (k.get("x2") is not None) == (len(k["gn_in"]) == 7)      # skip-concat: the skip's partials ride along
        with torch.no_grad():
            got = plan.forward(x, t, cond, lam=3.0)
        assert (ref - got).abs().max().item() < 6e-3 * max(1.0, ref.abs().max().item()), on
    assert counts[True][1] >= 3 and counts[False][1] == 0
    # skip-concat inputs: both sources need producer partials — at this toy size the 1x1 linears run on 64-column tiles
    # (no GroupNorm statistics), so plan them on 64x160 tiles as the 512x512 workload does
    monkeypatch.setattr(E, "GN_IN_CONV", True)
    real = E.plan_tiling
    monkeypatch.setattr(E, "plan_tiling", lambda m, n, k, taps, *a, **kw: (64, 160, 1, 0) if taps == 1 and n % 160 == 0 and m % 64 == 0
                        and not (a and a[0]) else real(m, n, k, taps, *a, **kw))
    plan = E.UNetPlan(TorchRefBackend(), unet_sd, b, s)
    assert any(k.get("gn_in") is not None and k.get("x2") is not None for _, _, k in plan.ops), "a skip-concat conv normalises too"
    with torch.no_grad():
        got = plan.forward(x, t, cond, lam=3.0)
    assert (ref - got).abs().max().item() < 6e-3 * max(1.0, ref.abs().max().item())
    assert counts[False][0] - counts[True][0] == counts[True][1]      # one groupnorm op less per conv that normalises


def test_unet_plan_fused_transformer_tail(unet_sd, monkeypatch):
    """norm3 -> GEGLU -> FF-out -> proj_out as one op (csrc/ffn_block.hip) at the 320-channel sites: the plan replaces
    three GEMM launches per block by ``ffn_block``, hands the GroupNorm partials of its output to the next resnet, and
    matches the oracle; also the host-side packing of the weight stream against the kernel's addressing (unpack)."""
    monkeypatch.setattr(E, "FFN_MIN_BLOCKS", 1)
    monkeypatch.setattr(E, "GN_FUSED_MAX_BYTES", 0)
    torch.manual_seed(2)
    b, s = 1, 16                        # level 0: 256 tokens = 4 row blocks of 64
    plan = E.UNetPlan(TorchRefBackend(), unet_sd, b, s)
    names = [getattr(fn, "__name__", "") for fn, _, _ in plan.ops]
    assert names.count("ffn_block") == 5 and names.count("tf_head") == 5       # head and tail of the five 320-channel blocks
    monkeypatch.setattr(E, "FUSED_FFN", False)
    monkeypatch.setattr(E, "FUSED_HEAD", False)
    plain = E.UNetPlan(TorchRefBackend(), unet_sd, b, s)
    assert len(plain.ops) - len(plan.ops) >= 20          # GEGLU + FF-out + proj_out -> one op and norm + proj_in + qkv -> one op, five times
    x, cond = torch.randn(b, 4, s, s), torch.randn(b, 48, 768) * 0.5
    t = torch.tensor([500])
    with torch.no_grad():
        ref = unet_forward(unet_sd, x, t, cond, delta_scale=3.0)
        got = plan.forward(x, t, cond, lam=3.0)
        got_plain = plain.forward(x, t, cond, lam=3.0)
    tol = 6e-3 * max(1.0, ref.abs().max().item())
    assert (ref - got).abs().max().item() < tol and (got - got_plain).abs().max().item() < tol
    u = "unet.unet.down_blocks.0.attentions.0"
    st, b1p = E.pack_ffn_stream(unet_sd[u + ".transformer_blocks.0.ff.net.0.proj.weight"], unet_sd[u + ".transformer_blocks.0.ff.net.0.proj.bias"],
                                unet_sd[u + ".transformer_blocks.0.ff.net.2.weight"], unet_sd[u + ".proj_out.weight"])
    w1, b1, w2, wp = TorchRefBackend.unpack_ffn_stream(st, b1p)
    assert torch.equal(w1, unet_sd[u + ".transformer_blocks.0.ff.net.0.proj.weight"].half().float())
    assert torch.equal(b1, unet_sd[u + ".transformer_blocks.0.ff.net.0.proj.bias"].float())
    assert torch.equal(w2, unet_sd[u + ".transformer_blocks.0.ff.net.2.weight"].half().float())
    assert torch.equal(wp, unet_sd[u + ".proj_out.weight"].reshape(320, 320).half().float())
    hst = E.pack_head_stream(unet_sd[u + ".proj_in.weight"], *[unet_sd[u + f".transformer_blocks.0.attn1.to_{n}.weight"] for n in "qkv"])
    wpi, wqkv = TorchRefBackend.unpack_head_stream(hst)
    assert torch.equal(wpi, unet_sd[u + ".proj_in.weight"].reshape(320, 320).half().float())
    assert torch.equal(wqkv, torch.cat([unet_sd[u + f".transformer_blocks.0.attn1.to_{n}.weight"] for n in "qkv"]).half().float())


def test_unet_plan_baseline_mode(full_sd):
    torch.manual_seed(2)
    sd = full_sd
    b, s = 1, 8
    plan = E.UNetPlan(TorchRefBackend(), sd, b, s, use_routing_gates=False)
    x, cond = torch.randn(b, 4, s, s), torch.randn(b, 32, 768) * 0.5
    t = torch.tensor([500])
    with torch.no_grad():
        ref = unet_forward(sd, x, t, cond, use_routing_gates=False)
        got = plan.forward(x, t, cond)
    assert (ref - got).abs().max().item() < 6e-3 * max(1.0, ref.abs().max().item())
    with pytest.raises(ValueError):
        plan.set_cond(torch.zeros(b, 48, 768))        # wrong token count for this mode
    with pytest.raises(ValueError):
        plan.set_cond(torch.zeros(b, 2, 3, 768))      # unet.py:129-131


def test_vae_decoder_plan_matches_oracle(full_sd):
    torch.manual_seed(3)
    sd = full_sd
    b, s = 1, 8
    plan = E.VaeDecoderPlan(TorchRefBackend(), sd, b, s, latent_scale=0.18215)
    z = torch.randn(b, 4, s, s) * 0.18215
    with torch.no_grad():
        ref = ((vae_decode(sd, z / 0.18215).clamp(-1, 1) + 1) / 2).clamp(0, 1)
    plan.z_in.copy_(z)
    plan.run()
    assert plan.img_out.shape == (b, 3, 8 * s, 8 * s)
    assert (ref - plan.img_out).abs().max().item() < 1e-2


def _module(cfg, sd, seed=0):
    return DiffusionModuleWithIP(cfg, state_dict=sd, device="cpu", seed=seed, batch_size=2,
                                 clip_config=TINY_CLIP, backend=TorchRefBackend())


def _oracle_cfg(mod):
    dc = mod.diff_cfg
    return OS.OracleCfg(image_size=mod.cfg.dataset.image_size, use_routing_gates=dc.use_routing_gates,
                        use_image_projection_plus=dc.use_image_projection_plus,
                        use_feature_purifier=dc.use_feature_purifier)


def test_module_casts_and_compile_assignment_keep_the_engine_path(full_sd):
    """The reference's augmentation pipeline does ``module = module.to(device); module = module.half(); module.eval()``
    and, with --compile, ``module.unet.unet = torch.compile(module.unet.unet, mode="reduce-overhead")``
    (src/pipelines/inference/inference_pipeline_ip_data_augment.py:372-379,398-400).  Here the casts return the module
    itself (operand precision is the engine's: fp16 tiles, fp32 accumulation), a cast to another device type is refused,
    the compile assignment is accepted without replacing the engine handle, and eps is bit-identical before and after."""
    cfg = default_config(**{"dataset.image_size": 64})
    mod = _module(cfg, full_sd)
    lat = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(7))
    t = torch.tensor([500, 20])
    cond = torch.randn(2, 48, 768, generator=torch.Generator().manual_seed(8)) * 0.5
    with torch.no_grad():
        e0 = mod(lat, t, cond).clone()
    assert mod.to(torch.device("cpu")) is mod and mod.to(torch.float16) is mod and mod.half() is mod and mod.float() is mod
    assert mod.eval() is mod and mod.training is False
    with pytest.raises(RuntimeError):
        mod.to("cuda" if mod.device.type == "cpu" else "cpu")
    inner = mod.unet.unet
    mod.unet.unet = torch.compile(mod.unet.unet, mode="reduce-overhead")       # lazy: nothing is traced by the assignment
    assert mod.unet.unet is inner and mod.unet.compiled_handle is not None
    assert len(dict(mod.unet.unet.named_modules())) == 16                     # the processors stay reachable (delta_scale)
    with torch.no_grad():
        e1 = mod(lat, t, cond)
        e2 = inner(lat, t, encoder_hidden_states=cond).sample                 # the diffusers-style call of unet.py:140-144
    assert torch.equal(e0, e1) and torch.equal(e0, e2)


@pytest.mark.parametrize("gates_on", [True, False])
def test_sampler_matches_oracle_config1_shape(gates_on, full_sd):
    """BASELINE config 1 in miniature (64x64 image, 4 DDIM steps): variant (i) gates on with
    steer lambda = 3.0, variant (ii) gates off with CFG g = 3.0 (two UNet calls per step)."""
    cfg = default_config(**{"dataset.image_size": 64, "model.use_routing_gates": gates_on})
    mod = _module(cfg, full_sd)
    torch.manual_seed(5)
    target = torch.tensor([3.0, 1.25])
    source = torch.tensor([0.0, 2.0])
    pix = torch.randn(1, 3, 224, 224)
    lat = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(1234))
    kw = dict(steer_scale=3.0) if gates_on else dict(guidance_scale=3.0)
    trace = []
    with torch.no_grad():
        got = PIPE._ddim_sample_ip(mod, target, source, pix, 4, torch.device("cpu"), latents=lat,
                                   use_graph=False, trace=trace, **kw)
        feats = mod.image_encoder.get_hidden_states(pix)
        ref_trace = []
        ref = OS.ddim_sample(mod._sd, _oracle_cfg(mod), target, source, feats, 4, lat,
                             trace=ref_trace, **kw)
        # conditioning tokens: AOE segments fp32 on both sides, the image segment through fp16 rows on the engine side
        c_ref = OS.prepare_conditioning(mod._sd, _oracle_cfg(mod), target, source, feats)
        c_got = PIPE._prepare_conditioning(mod, target, source, pix)
    assert (c_ref - c_got).abs().max().item() < 8e-3
    assert (c_ref[:, :16] - c_got[:, :16]).abs().max().item() < 1e-4      # AOE tokens: fp32 path
    assert len(trace) == 4
    for (e_g, x_g), (e_r, x_r, _, _) in zip(trace, ref_trace):
        if gates_on:   # with CFG the trace holds the conditional branch only
            assert (e_g - e_r).abs().max().item() < 2e-2
        # x0 = (x - s1*eps)/s0 amplifies an eps error by 1/sqrt(abar_999) = 25x at the first step,
        # and CFG by another (1 + 2g): fp16-storage noise, not a wiring difference
        assert (x_g - x_r).abs().max().item() < (5e-2 if gates_on else 0.2)
    assert (got - ref).abs().max().item() < (5e-2 if gates_on else 0.2)
    with torch.no_grad():
        img = PIPE._latents_to_images(mod, got)
        img_ref = OS.latents_to_images(mod._sd, _oracle_cfg(mod), ref)
    assert img.shape == (2, 3, 64, 64) and float(img.min()) >= 0.0 and float(img.max()) <= 1.0
    assert (img - img_ref).abs().max().item() < (3e-2 if gates_on else 0.1)


def test_module_protocol_and_errors(full_sd):
    cfg = default_config(**{"dataset.image_size": 64})
    mod = _module(cfg, full_sd)
    assert mod.diff_cfg.latent_scale == 0.18215 and mod.diff_cfg.num_train_timesteps == 1000
    assert mod.alphas_cumprod[999].item() == pytest.approx(0.001578963, rel=2e-6)
    procs = [m for _, m in mod.unet.unet.named_modules() if hasattr(m.processor, "delta_scale")]
    assert len(procs) == 16
    PIPE._set_delta_scale_on_processors(mod, 0.7)
    assert mod.unet.unet.delta_scale() == 0.7
    roles = {n.split(".transformer")[0]: p.processor.block_type for n, p in mod.unet.unet.named_modules()}
    assert roles["mid_block.attentions.0"] == "disease" and roles["down_blocks.0.attentions.0"] == "anatomy"
    with pytest.raises(ValueError):
        PIPE._ddim_sample_ip(mod, torch.zeros(2), torch.zeros(2), torch.zeros(1, 3, 224, 224), 1001,
                             torch.device("cpu"))
    with pytest.raises(ValueError):
        PIPE._build_labels(0)
    assert PIPE._build_labels(13).tolist()[1] == 0.25
    with pytest.raises(FileNotFoundError):
        PIPE._load_config("/nonexistent/train_ip.yaml")
    bad = default_config(**{"diffusion.noise_schedule": "cosine"})
    with pytest.raises(NotImplementedError):
        _module(bad, full_sd)
    with pytest.raises(ValueError):
        mod(torch.zeros(2, 4, 8, 8), torch.zeros(2, dtype=torch.long), torch.zeros(2, 3, 4, 768))


def test_vae_encoder_plan_matches_oracle():
    """``SDVAE.encode`` wiring (asymmetric stride-2 padding, quant_conv composed into conv_out, logvar clamp,
    reparameterised sample with injected noise) through the torch test backend vs the oracle."""
    from oracle.sd_vae import vae_encode_moments, vae_encode_sample
    sd = W.init_state_dict(W.vae_shapes(decoder=False), 3)
    b, s = 2, 4
    plan = E.VaeEncoderPlan(TorchRefBackend(), sd, b, s)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(b, 3, 8 * s, 8 * s, generator=g) * 2 - 1
    noise = torch.randn(b, 4, s, s, generator=g)
    with torch.no_grad():
        mean, logvar = vae_encode_moments(sd, x)
        ref = vae_encode_sample(sd, x, noise) * 0.18215
    plan.img_in.copy_(x)
    plan.run()
    assert (plan.mean - mean).abs().max().item() < 2e-2 * max(1.0, mean.abs().max().item())
    assert (plan.logvar - logvar).abs().max().item() < 2e-2 * max(1.0, logvar.abs().max().item())
    out = torch.zeros_like(mean)
    plan.be.gaussian_sample(plan.mean, plan.logvar, noise, out, 0.18215)
    assert (out - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())


def test_module_vae_encode_protocol(full_sd):
    """``module.vae.encode(x).latent_dist.sample()`` (diffusion_module_ip.py:410-411) on the module facade."""
    from oracle.sd_vae import vae_encode_sample
    sd = dict(full_sd)
    sd.update(W.init_state_dict(W.vae_shapes(decoder=False), 0))
    mod = _module(default_config(**{"dataset.image_size": 64}), sd)
    g = torch.Generator().manual_seed(12)
    x = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    noise = torch.randn(2, 4, 8, 8, generator=g)
    dist = mod.vae.encode(x).latent_dist
    z = dist.sample(noise=noise) * mod.diff_cfg.latent_scale
    with torch.no_grad():
        ref = vae_encode_sample(sd, x, noise) * 0.18215
    assert z.shape == (2, 4, 8, 8) and (z - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())
    assert torch.equal(dist.mode(), dist.mean) and float(dist.logvar.max()) <= 20.0 and torch.allclose(dist.std ** 2, dist.var)
    assert dist.sample().shape == (2, 4, 8, 8)                       # device RNG path
    with pytest.raises(ValueError):
        mod.vae.encode(torch.zeros(1, 3, 60, 60))
    no_enc = _module(default_config(**{"dataset.image_size": 64}), full_sd)
    with pytest.raises(KeyError):
        no_enc.vae.encode(x)


def test_groupnorm_statistics_from_the_producing_epilogue(monkeypatch):
    """conv(..., gn_stats=True) -> gn(): the epilogue's chunk partials (DADD_EPI_GNSTAT) replace the statistics pass.
    The torch backend's groupnorm NORMALISES FROM THE PARTIALS, so a wrong chunk / group layout shows up here."""
    import torch.nn.functional as Fn
    be = TorchRefBackend()
    plan = E._Plan(be)
    b, hw, cin, cout = 2, 32, 64, 320
    g = torch.Generator().manual_seed(3)
    x = torch.randn(b, hw, hw, cin, generator=g).to(torch.float16)
    w = (torch.randn(cout, 9 * cin, generator=g) / 24).to(torch.float16)
    bias = torch.randn(cout, generator=g) * 0.1
    res = torch.randn(b, hw, hw, cout, generator=g).to(torch.float16)
    gamma, beta = 1 + 0.1 * torch.randn(cout, generator=g), 0.1 * torch.randn(cout, generator=g)
    plan.gn_ws = be.empty((b * 256 * 32 * 2,), torch.float32)
    monkeypatch.setitem(E.TILING_OVERRIDE, E.tiling_key(b * hw * hw, cout, 9 * cin, 9, False, True), (128, 160, 1, 0))
    out = plan.conv(x, w, (b, hw, hw, cout), bias=bias, residual=res, gn_stats=True)
    assert out.data_ptr() in plan.gn_partials and plan.gn_partials[out.data_ptr()][1] == hw * hw // 64
    y = plan.gn(out, None, gamma, beta, 1e-5, 1)
    kinds = [(getattr(fn, "__name__", ""), k.get("ws_chunks", 0), bool(k.get("flags", 0) & 2048)) for fn, _, k in plan.ops]
    assert kinds == [("igemm", 0, True), ("groupnorm", 16, False)]
    plan.run()
    ref = Fn.silu(Fn.group_norm(out.float().permute(0, 3, 1, 2), 32, gamma, beta, 1e-5)).permute(0, 2, 3, 1)
    assert (y.float() - ref).abs().max().item() < 4e-3
    plan.pool.put(out)                                   # a recycled buffer must not keep its statistics
    assert out.data_ptr() not in plan.gn_partials
    # split-K: the finish kernel writes the partials (SPLITK_GN_ROWS-row chunks); 64-column tiles: no partials, two-pass GroupNorm
    monkeypatch.setitem(E.TILING_OVERRIDE, E.tiling_key(b * hw * hw, cout, 9 * cin, 9, False, True), (128, 160, 2, 0))
    out2 = plan.conv(x, w, (b, hw, hw, cout), bias=bias, residual=res, gn_stats=True)
    assert plan.gn_partials[out2.data_ptr()][1] == hw * hw // E.SPLITK_GN_ROWS
    y2 = plan.gn(out2, None, gamma, beta, 1e-5, 1)
    plan.ops[-2][0](*plan.ops[-2][1], **plan.ops[-2][2])
    plan.ops[-1][0](*plan.ops[-1][1], **plan.ops[-1][2])
    assert (y2.float() - ref).abs().max().item() < 4e-3
    w1 = (torch.randn(cout, cin, generator=g) / 8).to(torch.float16)
    monkeypatch.setitem(E.TILING_OVERRIDE, E.tiling_key(b * hw * hw, cout, cin, 1, False, False), (64, 64, 1, 0))
    out3 = plan.conv(x, w1, (b, hw, hw, cout), taps=1, pad=0, gn_stats=True)
    assert out3.data_ptr() not in plan.gn_partials
