#!/usr/bin/env python
"""In-kernel s_memtime stamps of conv3x3_halo_kernel (diagnostic library built with -DDADD_IGEMM_EXP=3):
where one MFMA wave and one loader wave per workgroup spend their time.  usage: python scripts/exp_stamps.py"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from progressive_stable_diffusion_amd import lib as L  # noqa: E402

L.LIB_PATH = os.path.join(ROOT, "progressive-stable-diffusion_amd", "exp", "libdadd_exp3.so")
from progressive_stable_diffusion_amd.backend import HipBackend  # noqa: E402

be = HipBackend(torch.device("cuda:0"))
g = torch.Generator().manual_seed(0)
for b, side, cin, cout in ((4, 64, 640, 640), (4, 64, 320, 320), (4, 32, 640, 640)):
    x = be.to_device(torch.randn(b, side, side, cin, generator=g).half())
    k = 9 * cin
    w = be.to_device((torch.randn(cout, k, generator=g) / math.sqrt(k)).half())
    out = be.zeros((b, side, side, cout), torch.float16)
    nwg = (b * side * side // 128) * (cout // 160)
    stamps = be.zeros((max(nwg * 16, b * side * side * cout),), torch.float32)   # 8 x uint64 per workgroup (sized as a split-K slab)
    for _ in range(3):
        be.igemm(x, w, out, taps=9, pad=1, splitk=1, partial=stamps, tile_m=128)
    be.synchronize()
    s = stamps[:nwg * 16].view(torch.int64).view(nwg, 8).double().cpu()
    m = s.mean(0)
    nit = k // 64
    print(f"conv {b}x{side}x{side} {cin}->{cout}: {nit} taps x chunks per workgroup, stamps in s_memtime ticks (shader-clock cycles)")
    print(f"  MFMA wave : loop {m[3]:9.0f} = barrier {m[0]:9.0f} ({100 * m[0] / m[3]:4.1f} %) + K half 0 {m[1]:9.0f} "
          f"({100 * m[1] / m[3]:4.1f} %) + K half 1 {m[2]:9.0f} ({100 * m[2] / m[3]:4.1f} %);  per tap {m[3] / nit:.0f} cycles (MFMA alone: 640)")
