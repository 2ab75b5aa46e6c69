#!/usr/bin/env python
"""Launch one implicit-GEMM shape a few times (for rocprofv3 --pmc passes).
usage: python scripts/igemm_pmc.py M_side Cin Cout taps splitk tune [iters]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from progressive_stable_diffusion_amd.backend import HipBackend  # noqa: E402

side, cin, cout, taps, sk, tune = (int(a) for a in sys.argv[1:7])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
be = HipBackend(torch.device("cuda:0"))
b = 4
g = torch.Generator().manual_seed(0)
x = be.to_device((torch.randn(b, side, side, cin, generator=g)).half())
w = be.to_device((torch.randn(cout, taps * cin, generator=g) / math.sqrt(taps * cin)).half())
out = be.zeros((b, side, side, cout), torch.float16)
partial = be.zeros((sk * b * side * side * cout,), torch.float32) if sk > 1 else None
for _ in range(iters):
    be.igemm(x, w, out, taps=taps, pad=taps // 9, flags=tune, splitk=sk, partial=partial, tile_m=128)
be.synchronize()
print("done", side, cin, cout, taps, sk, tune)
