"""Seeded inputs shared by ``oracle/make_golden.py`` (writer) and the golden tests (readers).

Only the reference's OUTPUTS are stored in ``tests/golden/*.npz``; the random inputs are
regenerated here from fixed seeds (CPU ``torch.Generator`` streams are platform-independent).
"""
import torch

SEED = 0
GATES = {"anatomy": (0.1, 0.9), "disease": (0.9, 0.1), "both": (0.5, 0.5)}
LABELS = [0.0, 0.25, 1.0, 1.6, 3.0, 3.7, -0.5]
SOURCE = [2.0, 0.0, 1.0, 3.0, 2.5, 0.0, 3.0]
PUR_SOURCE = [2.0, 1.5]
# (attn2 site, channels, query tokens)
XATTN_CASES = [("down_blocks.0.attentions.1", 320, 48), ("up_blocks.2.attentions.0", 640, 32),
               ("mid_block.attentions.0", 1280, 16)]
LAMBDAS = (0.0, 0.5, 3.0)
MODES = ("both", "aoe_dominant", "image_dominant")


def rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def purifier_image_tokens():
    return rand((2, 16, 768), 11)


def clip_hidden():
    return rand((1, 257, 1024), 12)


def clip_embeds():
    return rand((2, 768), 13)


def xattn_inputs(c, n):
    return rand((1, n, c), 100 + c), rand((1, 48, 768), 200 + c, 0.7)
