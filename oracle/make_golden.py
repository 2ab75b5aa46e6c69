"""Generate ``tests/golden/*.npz`` from the IMPORTED reference modules (build container only).

Run from the repo root:  ``python oracle/make_golden.py``
Needs ``/root/reference`` (absent on the GPU box — the fixtures, not this script, travel).

What is pinned (SURVEY.md §8c "Golden vectors / KATs available"):
  1. AdditiveOrdinalEmbedder: forward / get_negative_embedding / get_ordinal_delta_embedding
  2. FeaturePurifier(768, 8, 2)
  3. ImageProjectionPlus(1024->768, 16 tokens), ImageProjection(768->768x16)
  4. SplitInjectionAttentionProcessor for (N,C) in {(96,320),(64,640),(32,1280)}, both gate
     settings, lambda in {0, 0.5, 3.0}; OrdinalIPAttnProcessor2_0 for its three modes
  5. block-role tables;  6. schedule constants (recomputed in the tests, App. C)
Weights are NOT stored: they are regenerated from the seed by
``progressive_stable_diffusion_amd.weights`` (per-key generators), loaded here into the
reference modules with ``load_state_dict`` and into the oracle as a flat dict.  Only inputs
and the reference's outputs are written (fp32, a few hundred KB).

The two attention-processor files import ``diffusers.models.attention_processor.AttnProcessor2_0``
(used only as the default for attn1); ``diffusers`` is not installed, so an empty stand-in
class is seeded into ``sys.modules`` for that ONE symbol before import.
Nothing of the reference's arithmetic is stubbed.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from progressive_stable_diffusion_amd import weights as W  # noqa: E402
from oracle import conditioning as OC  # noqa: E402
from oracle import processors as OP  # noqa: E402

from tests import golden_inputs as GI  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = GI.SEED


def _stub_diffusers():
    d = types.ModuleType("diffusers")
    m = types.ModuleType("diffusers.models")
    a = types.ModuleType("diffusers.models.attention_processor")
    a.AttnProcessor2_0 = type("AttnProcessor2_0", (), {})
    sys.modules.update({"diffusers": d, "diffusers.models": m,
                        "diffusers.models.attention_processor": a})


def _sub(sd, prefix):
    return {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}


def _check(name, ref, mine, tol=2e-5):
    err = (ref - mine).abs().max().item()
    scale = ref.abs().max().item()
    print(f"  {name:42s} max|ref-oracle|={err:.3e}  (|ref|max={scale:.3e})")
    assert err <= tol * max(1.0, scale), name


@torch.no_grad()
def main():
    os.makedirs(OUT, exist_ok=True)
    _stub_diffusers()
    from src.models.ordinal_embedder import AdditiveOrdinalEmbedder
    from src.models.feature_purifier import FeaturePurifier
    from src.models.image_encoder import ImageProjection, ImageProjectionPlus
    from src.models.attention_processor_routing_gates import (
        SplitInjectionAttentionProcessor, get_block_type)
    from src.models.attention_processor_base import (
        OrdinalIPAttnProcessor2_0, get_frequency_mode_for_block)

    # ------------------------------------------------------------------ conditioning
    shapes = W.conditioning_shapes()
    sd = W.init_state_dict(shapes, SEED)
    aoe = AdditiveOrdinalEmbedder(num_classes=4, embedding_dim=768, delta_scale=0.05, num_tokens=16)
    aoe.load_state_dict(_sub(sd, "ordinal_embedder"))
    pur = FeaturePurifier(768, 8, 2)
    pur.load_state_dict(_sub(sd, "feature_purifier"))
    plus = ImageProjectionPlus(clip_hidden_dim=1024, cross_attention_dim=768, num_tokens=16)
    plus.load_state_dict(_sub(sd, "image_projection"))
    aoe.eval(), pur.eval(), plus.eval()

    labels, source = torch.tensor(GI.LABELS), torch.tensor(GI.SOURCE)
    g = {}
    g["aoe_forward"] = aoe(labels, is_training=False)
    g["aoe_negative"] = aoe.get_negative_embedding(labels, is_training=False)
    g["aoe_delta"] = aoe.get_ordinal_delta_embedding(source, labels)
    g["aoe_delta_same"] = aoe.get_ordinal_delta_embedding(labels, labels)
    print("AOE")
    _check("forward", g["aoe_forward"], OC.aoe_forward(sd, labels))
    _check("negative", g["aoe_negative"], OC.aoe_negative(sd, labels))
    _check("delta", g["aoe_delta"], OC.aoe_delta(sd, source, labels))
    assert g["aoe_delta_same"].abs().max().item() == 0.0

    img = GI.purifier_image_tokens()
    src_aoe = aoe(torch.tensor(GI.PUR_SOURCE), is_training=False)
    g["pur_out"] = pur(img, src_aoe)
    print("FeaturePurifier")
    _check("purifier", g["pur_out"], OC.feature_purifier(sd, img, src_aoe))

    hid = GI.clip_hidden()
    g["plus_out"] = plus(hid)
    print("ImageProjectionPlus")
    _check("resampler", g["plus_out"], OC.image_projection_plus(sd, hid))

    shapes_b = W.conditioning_shapes(projection_plus=False, purifier=False)
    sd_b = W.init_state_dict(shapes_b, SEED)
    basic = ImageProjection(clip_embedding_dim=768, cross_attention_dim=768, num_tokens=16)
    basic.load_state_dict(_sub(sd_b, "image_projection"))
    emb = GI.clip_embeds()
    g["basic_out"] = basic(emb)
    print("ImageProjection")
    _check("basic", g["basic_out"], OC.image_projection(sd_b, emb))
    np.savez_compressed(os.path.join(OUT, "conditioning.npz"),
                        **{k: v.numpy().astype(np.float32) for k, v in g.items()})

    # ------------------------------------------------------------------ attention processors
    class Attn:  # duck-typed diffusers Attention, exactly the attributes the processors read
        spatial_norm = None
        group_norm = None
        norm_cross = None
        residual_connection = False
        rescale_output_factor = 1.0

        def __init__(self, sd, ap, heads=8):
            self.heads = heads
            mk = lambda w, b=None: (lambda x: torch.nn.functional.linear(x, w, b))  # noqa: E731
            self.to_q = mk(sd[ap + ".to_q.weight"])
            self.to_k = mk(sd[ap + ".to_k.weight"])
            self.to_v = mk(sd[ap + ".to_v.weight"])
            self.to_out = [mk(sd[ap + ".to_out.0.weight"], sd[ap + ".to_out.0.bias"]), lambda x: x]

    ushapes = W.unet_shapes()
    gates = GI.GATES
    cases = GI.XATTN_CASES
    x_out = {}
    print("SplitInjectionAttentionProcessor / OrdinalIPAttnProcessor2_0")
    for site, c, n in cases:
        ap = f"unet.unet.{site}.transformer_blocks.0.attn2"
        keys = [k for k in ushapes if k.startswith(ap + ".")]
        sdu = W.init_state_dict(ushapes, SEED, gates=gates, warm_start_dis=False, keys=keys)
        role = get_block_type(site)
        proc = SplitInjectionAttentionProcessor(
            hidden_size=c, cross_attention_dim=768, block_type=role,
            anat_gate_init=gates[role][0], dis_gate_init=gates[role][1])
        proc.load_state_dict(_sub(sdu, ap + ".processor"))
        attn = Attn(sdu, ap)
        x, cond3 = GI.xattn_inputs(c, n)
        tag = site.replace(".", "_")
        x_out[f"{tag}__gates"] = torch.stack([proc.anat_gate, proc.dis_gate])
        for lam in GI.LAMBDAS:
            proc.delta_scale = lam
            ref = proc(attn, x, encoder_hidden_states=cond3)
            x_out[f"{tag}__split_l{lam}"] = ref
            _check(f"{site} split lambda={lam}", ref,
                   OP.split_injection_attention(sdu, ap, x, cond3, 8, lam))
        cond2 = cond3[:, :32]
        for mode in GI.MODES:
            bproc = OrdinalIPAttnProcessor2_0(hidden_size=c, cross_attention_dim=768,
                                              frequency_mode=mode)
            ref = bproc(attn, x, encoder_hidden_states=cond2)
            x_out[f"{tag}__base_{mode}"] = ref
            _check(f"{site} baseline {mode}", ref,
                   OP.ordinal_ip_attention(sdu, ap, x, cond2, 8, mode))
    np.savez_compressed(os.path.join(OUT, "xattn.npz"),
                        **{k: v.numpy().astype(np.float32) for k, v in x_out.items()})

    # ------------------------------------------------------------------ role tables
    names = ([f"down_blocks.{i}.attentions.{j}.transformer_blocks.0.attn2.processor"
              for i in range(3) for j in range(2)]
             + ["mid_block.attentions.0.transformer_blocks.0.attn2.processor"]
             + [f"up_blocks.{i}.attentions.{j}.transformer_blocks.0.attn2.processor"
                for i in (1, 2, 3) for j in range(3)] + ["conv_in", "time_embedding.linear_1"])
    with open(os.path.join(OUT, "block_roles.tsv"), "w") as f:
        for nme in names:
            f.write(f"{nme}\t{get_block_type(nme)}\t{get_frequency_mode_for_block(nme)}\n")
            assert get_block_type(nme) == OP.block_role(nme)
            assert get_frequency_mode_for_block(nme) == OP.frequency_mode(nme)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
