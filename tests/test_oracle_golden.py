"""The oracle replayed against the reference's own outputs (tests/golden, made by
oracle/make_golden.py from the imported reference modules) and against the known-answer
constants of SURVEY.md Appendix C.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import conditioning as OC
from oracle import processors as OP
from oracle import sampler as OS
from progressive_stable_diffusion_amd import routing as R
from progressive_stable_diffusion_amd import weights as W
from tests import golden_inputs as GI


def _close(a, ref, tol=2e-5):
    ref = torch.from_numpy(ref)
    assert a.shape == ref.shape
    assert (a - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.fixture(scope="module")
def cond_sd():
    return W.init_state_dict(W.conditioning_shapes(), GI.SEED)


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "conditioning.npz"))


def test_aoe_matches_reference(cond_sd, gold):
    labels, source = torch.tensor(GI.LABELS), torch.tensor(GI.SOURCE)
    _close(OC.aoe_forward(cond_sd, labels), gold["aoe_forward"])
    _close(OC.aoe_negative(cond_sd, labels), gold["aoe_negative"])
    _close(OC.aoe_delta(cond_sd, source, labels), gold["aoe_delta"])


def test_aoe_delta_is_exactly_zero_for_same_label(cond_sd, gold):
    # ordinal_embedder.py:254-255 promises an exact zero when source == target
    labels = torch.tensor(GI.LABELS)
    assert np.abs(gold["aoe_delta_same"]).max() == 0.0
    assert OC.aoe_delta(cond_sd, labels, labels).abs().max().item() == 0.0


def test_feature_purifier_matches_reference(cond_sd, gold):
    src = OC.aoe_forward(cond_sd, torch.tensor(GI.PUR_SOURCE))
    _close(OC.feature_purifier(cond_sd, GI.purifier_image_tokens(), src), gold["pur_out"])


def test_resampler_matches_reference(cond_sd, gold):
    _close(OC.image_projection_plus(cond_sd, GI.clip_hidden()), gold["plus_out"])


def test_basic_projection_matches_reference(gold):
    sd = W.init_state_dict(W.conditioning_shapes(projection_plus=False, purifier=False), GI.SEED)
    _close(OC.image_projection(sd, GI.clip_embeds()), gold["basic_out"])


@pytest.mark.parametrize("site,c,n", GI.XATTN_CASES)
def test_cross_attention_processors_match_reference(site, c, n, golden_dir):
    g = np.load(os.path.join(golden_dir, "xattn.npz"))
    ush = W.unet_shapes()
    ap = f"unet.unet.{site}.transformer_blocks.0.attn2"
    sd = W.init_state_dict(ush, GI.SEED, gates=GI.GATES, warm_start_dis=False,
                           keys=[k for k in ush if k.startswith(ap + ".")])
    x, cond3 = GI.xattn_inputs(c, n)
    tag = site.replace(".", "_")
    gates = g[f"{tag}__gates"]
    assert float(sd[ap + ".processor.anat_gate"]) == pytest.approx(float(gates[0]))
    assert float(sd[ap + ".processor.dis_gate"]) == pytest.approx(float(gates[1]))
    outs = {}
    for lam in GI.LAMBDAS:
        outs[lam] = OP.split_injection_attention(sd, ap, x, cond3, 8, lam)
        _close(outs[lam], g[f"{tag}__split_l{lam}"])
    assert (outs[3.0] - outs[0.0]).abs().max() > 1e-3          # lambda changes the result
    for mode in GI.MODES:
        _close(OP.ordinal_ip_attention(sd, ap, x, cond3[:, :32], 8, mode), g[f"{tag}__base_{mode}"])
    # base.py:103-116 with scales 1 is the identity up to rounding
    a = OP.ordinal_ip_attention(sd, ap, x, cond3[:, :32], 8, "both")
    b = OP.ordinal_ip_attention(sd, ap, x, cond3[:, :32], 8, "aoe_dominant")
    assert (a - b).abs().max().item() < 1e-6


def test_block_role_tables(golden_dir):
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(golden_dir, "block_roles.tsv"))]
    assert len(rows) == 18
    for name, role, mode in rows:
        assert OP.block_role(name) == role and R.get_block_type(name) == role
        assert OP.frequency_mode(name) == mode and R.get_frequency_mode_for_block(name) == mode


def test_schedule_known_answers():
    # SURVEY.md Appendix C
    assert OS.ddim_timesteps(1000, 10).tolist() == [999, 888, 777, 666, 555, 444, 333, 222, 111, 0]
    assert OS.ddim_timesteps(1000, 13).tolist() == [999, 915, 832, 749, 666, 582, 499, 416, 333,
                                                    249, 166, 83, 0]
    t50 = OS.ddim_timesteps(1000, 50).tolist()
    assert t50[:6] == [999, 978, 958, 937, 917, 897] and t50[-4:] == [61, 40, 20, 0]
    _, ac, prev, snr = OS.noise_schedule(OS.OracleCfg())
    for i, v in ((0, 0.999149978), (1, 0.998289526), (499, 0.161812171), (978, 0.002029766),
                 (999, 0.001578963)):
        assert ac[i].item() == pytest.approx(v, rel=2e-6)
    assert torch.sqrt(1 - ac[999]).item() == pytest.approx(0.99921018, rel=1e-6)
    assert snr[999].item() == pytest.approx(0.00158146, rel=1e-5)
    assert prev[0].item() == 1.0 and prev[1].item() == ac[0].item()
    with pytest.raises(NotImplementedError):
        OS.noise_schedule(OS.OracleCfg(noise_schedule="cosine"))
