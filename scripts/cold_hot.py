#!/usr/bin/env python
"""One implicit-GEMM shape with its weights cold (a 1 GB copy in between evicts L2 and the 256 MB Infinity Cache) against
hot (the same launch repeated): what a weight prefetch ahead of the launch could buy.
usage: python scripts/cold_hot.py side Cin Cout taps splitk [tile_m tile_n]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from progressive_stable_diffusion_amd.backend import HipBackend  # noqa: E402

side, cin, cout, taps, sk = (int(a) for a in sys.argv[1:6])
tile_m = int(sys.argv[6]) if len(sys.argv) > 6 else 128
tile_n = int(sys.argv[7]) if len(sys.argv) > 7 else 0
be = HipBackend(torch.device("cuda:0"))
b = 4
g = torch.Generator().manual_seed(0)
x = be.to_device((torch.randn(b, side, side, cin, generator=g)).half())
w = be.to_device((torch.randn(cout, taps * cin, generator=g) / math.sqrt(taps * cin)).half())
out = be.zeros((b, side, side, cout), torch.float16)
partial = be.zeros((sk * b * side * side * cout,), torch.float32) if sk > 1 else None
big_a, big_b = be.zeros((256 * 1024 * 1024,), torch.float32), be.zeros((256 * 1024 * 1024,), torch.float32)


def conv():
    be.igemm(x, w, out, taps=taps, pad=taps // 9, splitk=sk, partial=partial, tile_m=tile_m, tile_n=tile_n)


for _ in range(3):
    conv()
be.synchronize()
res = {}
for mode in ("hot", "cold", "cold_w_only", "w_prefetched"):
    ts = []
    for _ in range(6):
        if mode != "hot":
            be.copy_(big_b, big_a)          # 2 GB of traffic: nothing of x / w / out is left in L2 or the Infinity Cache
            if mode in ("cold_w_only", "w_prefetched"):   # ... then x and the output back in (what the previous kernel of a step leaves)
                be.copy_(out, out.clone())
                x.add_(0)
            if mode == "w_prefetched":      # another kernel has READ the weights (whatever XCD): Infinity-Cache-hot, L2 mostly not
                with be.ctx():
                    sink = w.view(-1)[::8].float().sum()
        be.synchronize()
        be.prof_begin()
        conv()
        rec = be.prof_end()
        ts.append(sum(r[1] for r in rec))
        names = [r[0] for r in rec]
    res[mode] = sorted(ts)[len(ts) // 2]
print(f"{side}x{side} {cin}->{cout} taps {taps} splitk {sk} tile {tile_m}x{tile_n or 'auto'}: {' + '.join(names)}: "
      + ", ".join(f"{k} {v:.2f} us" for k, v in res.items()))
