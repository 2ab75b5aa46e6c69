"""Per-block role tables of the DADD cross-attention sites.

Mirrors the two reference functions by name and behaviour:
  * ``get_block_type``               — src/models/attention_processor_routing_gates.py:199-230
  * ``get_frequency_mode_for_block`` — src/models/attention_processor_base.py:141-167
(SURVEY.md Appendix A.3 has the resulting table for the 16 attn2 sites.)
"""
from __future__ import annotations

import re

_DOWN = re.compile(r"down_blocks\.(\d+)")
_UP = re.compile(r"up_blocks\.(\d+)")


def _index(rx, name):
    m = rx.search(name)
    return int(m.group(1)) if m else None


def get_block_type(block_name: str) -> str:
    """'disease' for the low-resolution sites (mid, down>=2, up<=1), 'anatomy' for the rest."""
    if "mid_block" in block_name:
        return "disease"
    i = _index(_DOWN, block_name)
    if i is not None:
        return "disease" if i >= 2 else "anatomy"
    i = _index(_UP, block_name)
    if i is not None:
        return "disease" if i <= 1 else "anatomy"
    return "both"


def get_frequency_mode_for_block(block_name: str) -> str:
    """Baseline processor mode; same geography as ``get_block_type``."""
    if "mid_block" in block_name:
        return "aoe_dominant"
    i = _index(_DOWN, block_name)
    if i is not None:
        return "image_dominant" if i <= 1 else "aoe_dominant"
    i = _index(_UP, block_name)
    if i is not None:
        return "aoe_dominant" if i <= 1 else "image_dominant"
    return "both"
