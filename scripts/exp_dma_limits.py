#!/usr/bin/env python
"""Time the LDS-DMA GEMM on the dominant shapes with the product library (0) or a diagnostic one
(1: DMA stream only, 2: MFMA + LDS reads only).  usage: python scripts/exp_dma_limits.py {0,1,2}"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from progressive_stable_diffusion_amd import lib as L  # noqa: E402

exp = int(sys.argv[1])
if exp:
    L.LIB_PATH = os.path.join(ROOT, "progressive-stable-diffusion_amd", "exp", f"libdadd_exp{exp}.so")
from progressive_stable_diffusion_amd.backend import HipBackend  # noqa: E402

be = HipBackend(torch.device("cuda:0"))
g = torch.Generator().manual_seed(0)
SHAPES = [  # b, side, cin, cout, taps, splitk
    (4, 64, 320, 320, 9, 1), (4, 64, 640, 320, 9, 1), (4, 64, 640, 640, 9, 1), (4, 32, 640, 640, 9, 2),
    (4, 64, 320, 960, 1, 1), (4, 64, 320, 2560, 1, 1), (4, 64, 1280, 320, 1, 1), (4, 64, 320, 320, 1, 1),
]
for b, side, cin, cout, taps, sk in SHAPES:
    x = be.to_device(torch.randn(b, side, side, cin, generator=g).half())
    k = taps * cin
    w = be.to_device((torch.randn(cout, k, generator=g) / math.sqrt(k)).half())
    out = be.zeros((b, side, side, cout), torch.float16)
    partial = be.zeros((sk * b * side * side * cout,), torch.float32) if sk > 1 else None
    run = lambda: be.igemm(x, w, out, taps=taps, pad=taps // 9, splitk=sk, partial=partial, tile_m=128)  # noqa: E731
    for _ in range(3):
        run()
    be.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    with be.ctx():
        e0.record()
        for _ in range(n):
            run()
        e1.record()
    be.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    m = b * side * side
    nkt = k // 64
    tiles = (m // 128) * ((cout + 159) // 160)
    print(f"exp{exp} M{m} N{cout} K{k} sk{sk}: {us:7.1f} us  {2.0 * m * cout * k / us / 1e6:7.1f} TF/s-equiv  "
          f"per K tile per wg-round {us / (nkt / sk * math.ceil(tiles * sk / 256)):.3f} us")
