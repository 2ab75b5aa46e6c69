#!/usr/bin/env python
"""Where a tf_head launch (csrc/tf_head.hip) spends its time: per-workgroup s_memtime stamps (diagnostic hook
dadd_tf_head_debug) of MFMA wave 0 and loader wave 0, medians over the 256 workgroups of the bench shape, in us at the
100 MHz s_memtime clock... (s_memtime counts shader-clock cycles; the launch's wall time is printed beside it).

stamps: 0 start | 1 token tile landed (B0) | 2 GroupNorm applied (Bc) | 3 proj_in done | 4 LayerNorm 1 done (E2) |
        5 q|k|v MFMAs + stores issued | 6 stores drained ; loader: 8 start | 9 five pieces issued | 10 token tile landed |
        11 barrier(0) passed | 12 sum of cycles in counted waits | 13 sum of cycles in barriers | 14 loader done
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from progressive_stable_diffusion_amd.backend import HipBackend
    from progressive_stable_diffusion_amd.engine import pack_head_stream
    F16, F32 = torch.float16, torch.float32
    be = HipBackend(torch.device("cuda:0"))
    b, hw, c, nchunk = 4, 4096, 320, 64
    g = torch.Generator().manual_seed(0)
    rn = lambda shape, s=1.0, dt=F16: (torch.randn(shape, generator=g) * s).to(dt)     # noqa: E731
    x = rn((b, hw, c))
    xs = x.float().reshape(b, nchunk, hw // nchunk, 32, c // 32)
    ws = torch.stack([xs.sum(dim=(2, 4)), (xs * xs).sum(dim=(2, 4))], dim=-1).reshape(-1).contiguous()
    stream = pack_head_stream(rn((c, c, 1, 1), 1 / math.sqrt(c)), *[rn((c, c), 1 / math.sqrt(c)) for _ in range(3)])
    args = [be.to_device(t) for t in (x, stream, ws)] + [nchunk] + [be.to_device(rn((c,), 0.2, F32)) for _ in range(5)]
    hs, qkv = be.zeros((b, hw, c), F16), be.zeros((b, hw, 3 * c), F16)
    nwg = b * hw // 64
    dbg = be.zeros((nwg * 16,), torch.int64)
    flush = be.zeros((150 * 1024 * 1024,), F32)
    for rep in range(4):
        be.zero_(flush)
        be.lib.dadd_tf_head_debug(dbg.data_ptr() if rep == 3 else None)
        be.prof_begin()
        be.tf_head(*args, hs, qkv)
        rec = be.prof_end()
    be.lib.dadd_tf_head_debug(None)
    be.synchronize()
    d = dbg.cpu().reshape(nwg, 16).double()
    t0 = d[:, 0:1]
    print(f"launch {rec[0][1]:.2f} us (dispatch timestamps)")
    names = {1: "token tile landed", 2: "GroupNorm applied", 3: "proj_in done", 4: "LayerNorm 1 done", 5: "q|k|v issued",
             6: "stores drained", 8: "loader start", 9: "loader: 5 pieces issued", 10: "loader: tile landed",
             11: "loader: barrier(0)", 14: "loader done"}
    ref = d[:, 6] - d[:, 0]
    print(f"cycles start -> end per workgroup: median {ref.median():.0f}  min {ref.min():.0f}  max {ref.max():.0f}  "
          f"(=> {rec[0][1] * 1e3 / ref.median():.3f} ns per tick if the median workgroup spans the launch)")
    for k in sorted(names):
        v = (d[:, k] - d[:, 0])
        print(f"  {k:2d} {names[k]:26s} median {v.median():9.0f}  min {v.min():9.0f}  max {v.max():9.0f}")
    print(f"  loader: cycles in counted waits median {d[:, 12].median():.0f}, in barriers median {d[:, 13].median():.0f}")
    start = d[:, 0] - d[:, 0].min()
    print(f"  workgroup start skew: median {start.median():.0f}  max {start.max():.0f} ticks")


if __name__ == "__main__":
    main()
