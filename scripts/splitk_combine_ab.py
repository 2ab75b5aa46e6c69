#!/usr/bin/env python
"""Split-K combine: finish kernel against the in-launch combine (tickets: the last-arriving slice reduces) for the
few-slice convs of the 16x16 / 32x32 maps.  usage: python scripts/splitk_combine_ab.py"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from progressive_stable_diffusion_amd.backend import HipBackend  # noqa: E402

be = HipBackend(torch.device("cuda:0"))
b = 4
g = torch.Generator().manual_seed(0)
for side, cin, cout, taps, sk in ((16, 1280, 1280, 9, 4), (32, 640, 640, 9, 2), (8, 1280, 1280, 9, 16), (16, 1280, 1280, 1, 4)):
    x = be.to_device((torch.randn(b, side, side, cin, generator=g)).half())
    w = be.to_device((torch.randn(cout, taps * cin, generator=g) / math.sqrt(taps * cin)).half())
    res = be.to_device((torch.randn(b, side, side, cout, generator=g)).half())
    bias = be.to_device(torch.randn(cout, generator=g))
    out = [be.zeros((b, side, side, cout), torch.float16) for _ in range(2)]
    partial = be.zeros((sk * b * side * side * cout,), torch.float32)
    cnt = be.zeros((4096,), torch.int32)
    t = {}
    for mode, counters in (("finish kernel", None), ("in-launch", cnt)):
        o = out[0] if counters is None else out[1]
        for _ in range(3):
            be.igemm(x, w, o, bias=bias, residual=res, taps=taps, pad=taps // 9, flags=5, splitk=sk, partial=partial, tile_m=128,
                     counters=counters)
        be.synchronize()
        be.prof_begin()
        for _ in range(10):
            be.igemm(x, w, o, bias=bias, residual=res, taps=taps, pad=taps // 9, flags=5, splitk=sk, partial=partial, tile_m=128,
                     counters=counters)
        rec = be.prof_end()
        t[mode] = sum(r[1] for r in rec) / 10
        names = sorted(set(r[0] for r in rec))
    same = torch.equal(out[0].cpu(), out[1].cpu())
    print(f"{side}x{side} {cin}->{cout} taps {taps} split-K {sk}: " + ", ".join(f"{k} {v:.2f} us" for k, v in t.items()) + f"; identical {same}")
