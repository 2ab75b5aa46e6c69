#!/usr/bin/env python
"""Where a wave of flash_kernel<40> spends its cycles (diagnostics build: rebuilds the library with -DDADD_FLASH_STAMPS on
the box it runs on; the snapshot's library is scratch there).  Prints the per-tile average of each phase over all waves.
usage: python scripts/flash_stamps.py   (DADD_FLASH40 selects the variant)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DADD_EXTRA_CFLAGS"] = "-DDADD_FLASH_STAMPS"
from progressive_stable_diffusion_amd import lib as L  # noqa: E402

if not os.environ.get("DADD_STAMPS_NOBUILD"):
    L.build(force=True)
import torch  # noqa: E402

from progressive_stable_diffusion_amd.backend import HipBackend  # noqa: E402

n, heads, d, b = 4096, 8, 40, 4
be = HipBackend(torch.device("cuda:0"))
c = heads * d
qkv = be.to_device(torch.randn(b, n, 3 * c, generator=torch.Generator().manual_seed(0)).half())
out = be.zeros((b, n, c), torch.float16)
var = int(os.environ.get("DADD_FLASH40", "1"))
nw = 8 if var == 1 else 4
nblk = b * heads * (n // 256)
buf = be.zeros((nblk * nw * 8,), torch.int64)
raw = C.CDLL(L.LIB_PATH)
raw.dadd_attn_debug.argtypes = [C.c_void_p]
for _ in range(2):
    be.self_attn(qkv, out, heads)
raw.dadd_attn_debug(buf.data_ptr())
be.self_attn(qkv, out, heads)
be.synchronize()
raw.dadd_attn_debug(None)
st = buf.cpu().view(nblk * nw, 8).double()
names = ["barrier", "global loads issued", "K reads + S MFMAs issued", "softmax (waits for S)", "V reads + PV MFMAs issued",
         "LDS stores of next tile"]
ntile = n // 64
tot = st[:, :6].sum(dim=1)
print(f"variant {var} ({nw} waves per block), {nblk} blocks; cycles per tile and wave (s_memtime ticks; mean over waves, min..max)")
for i, nm in enumerate(names):
    v = st[:, i] / ntile
    print(f"  {nm:28s} {v.mean():8.1f}   ({v.min():.0f} .. {v.max():.0f})   {100 * st[:, i].sum() / tot.sum():5.1f} %")
print(f"  {'sum':28s} {(tot / ntile).mean():8.1f}")
