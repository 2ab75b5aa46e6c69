#!/usr/bin/env python
"""Launch the fused transformer-block head / tail kernels (csrc/tf_head.hip, csrc/ffn_block.hip) a few times at the bench
shape (B = 4, 64x64 tokens), a 600 MB rewrite in between so their weight streams come from HBM as in the step — the
program profiled by scripts/pmc_kernel.sh.   usage: python scripts/rowblock_pmc.py [iters]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from progressive_stable_diffusion_amd.backend import HipBackend  # noqa: E402
from progressive_stable_diffusion_amd.engine import pack_ffn_stream, pack_head_stream  # noqa: E402

F16, F32 = torch.float16, torch.float32
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
be = HipBackend(torch.device("cuda:0"))
b, hw, c, hid, nchunk = 4, 4096, 320, 1280, 64
g = torch.Generator().manual_seed(0)
rn = lambda shape, s=1.0, dt=F16: (torch.randn(shape, generator=g) * s).to(dt)     # noqa: E731
x, xres = rn((b, hw, c)), rn((b, hw, c))
xs = x.float().reshape(b, nchunk, hw // nchunk, 32, c // 32)
ws = torch.stack([xs.sum(dim=(2, 4)), (xs * xs).sum(dim=(2, 4))], dim=-1).reshape(-1).contiguous()
st, b1p = pack_ffn_stream(rn((2 * hid, c), 1 / math.sqrt(c)), rn((2 * hid,), 0.2, F32), rn((c, hid), 1 / math.sqrt(hid)),
                          rn((c, c, 1, 1), 1 / math.sqrt(c)))
hst = pack_head_stream(rn((c, c, 1, 1), 1 / math.sqrt(c)), *[rn((c, c), 1 / math.sqrt(c)) for _ in range(3)])
vec = [be.to_device(1 + 0.2 * rn((c,), 1, F32)) if i % 2 == 0 else be.to_device(rn((c,), 0.2, F32)) for i in range(6)]
xd, xr, std, b1d, hstd, wsd = (be.to_device(t) for t in (x, xres, st, b1p, hst, ws))
out, hs, qkv = be.zeros((b, hw, c), F16), be.zeros((b, hw, c), F16), be.zeros((b, hw, 3 * c), F16)
gws = be.zeros((b * (hw // 32) * 64,), F32)
flush = be.zeros((150 * 1024 * 1024,), F32)
for _ in range(iters):
    be.zero_(flush)
    be.tf_head(xd, hstd, wsd, nchunk, vec[0], vec[1], vec[3], vec[2], vec[5], hs, qkv)
    be.zero_(flush)
    be.ffn_block(xd, std, vec[0], vec[1], b1d, vec[3], vec[5], xr, out, gn_ws=gws, gn_nchunk=hw // 32)
be.synchronize()
print("done")
