#!/usr/bin/env python
"""Per-launch time of the self-attention kernel (dispatch timestamps), B=4 at N tokens / heads / d, and the maximum
difference from a float64 softmax(QK^T)V of the same fp16 operands.  DADD_FLASH40 selects the d = 40 variant.
usage: python scripts/flash_bench.py [N heads d [iters]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from progressive_stable_diffusion_amd.backend import HipBackend  # noqa: E402

n, heads, d = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 8, 40)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
be = HipBackend(torch.device("cuda:0"))
b, c = 4, heads * d
qkv_h = torch.randn(b, n, 3 * c, generator=torch.Generator().manual_seed(0)).half()
qkv = be.to_device(qkv_h)
out = be.zeros((b, n, c), torch.float16)
for _ in range(3):
    be.self_attn(qkv, out, heads)
be.synchronize()
be.prof_begin()
for _ in range(iters):
    be.self_attn(qkv, out, heads)
rec = be.prof_end()
us = sorted(r[1] for r in rec)
q, k, v = (t.double().view(1, n, heads, d).transpose(1, 2) for t in qkv_h[:1].cuda().split(c, dim=-1))
ref = (torch.softmax(q @ k.transpose(-1, -2) / d ** 0.5, dim=-1) @ v).transpose(1, 2).reshape(n, c)
err = (out[0].double() - ref).abs().max().item()
print(f"variant {os.environ.get('DADD_FLASH40', '1')}: {rec[0][0]}  N={n} heads={heads} d={d}: median {us[len(us) // 2]:.2f} us, min {us[0]:.2f}, "
      f"max {us[-1]:.2f}  ({4.0 * b * n * n * c / us[len(us) // 2] * 1e-6:.0f} TF/s);  max |err| vs float64 {err:.2e}")
