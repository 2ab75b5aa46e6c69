#!/usr/bin/env python
"""Launch the self-attention kernel a few times (for rocprofv3 --pmc passes).
usage: python scripts/flash_pmc.py N heads d [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from progressive_stable_diffusion_amd.backend import HipBackend  # noqa: E402

n, heads, d = (int(a) for a in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
be = HipBackend(torch.device("cuda:0"))
b, c = 4, heads * d
qkv = be.to_device(torch.randn(b, n, 3 * c, generator=torch.Generator().manual_seed(0)).half())
out = be.zeros((b, n, c), torch.float16)
for _ in range(iters):
    be.self_attn(qkv, out, heads)
be.synchronize()
print("done")
