#!/usr/bin/env python
"""Micro-benchmark of the fused transformer-block tail (csrc/ffn_block.hip) at the bench shape (B = 4, 64x64 tokens,
C = 320): the product kernel and its timing-only diagnostic builds, interleaved in ONE process (same box, same clocks):

    DADD_FFN_EXP  0 product | 1 stream only | 2 DMA issued against a zero-record descriptor | 3 no DMA
    DADD_FFN_ROT  chunk rotation multiplier (0 = every workgroup sweeps the hidden chunks in the same order)

Times are the dispatch's own begin/end timestamps (backend.prof_begin / prof_end).  Inputs are random (zero operands
flatter MFMA clocks); between launches a 600 MB buffer is rewritten so weights come from beyond the L2s, as in the step.

    python scripts/ffn_bench.py [--reps 12] [--variants "0:7,0:0,1:7,2:7,3:7"]
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--hw", type=int, default=4096)
    ap.add_argument("--variants", default="0:0,1:0,2:0,3:0")
    ap.add_argument("--no-flush", action="store_true")
    a = ap.parse_args()
    from progressive_stable_diffusion_amd.backend import HipBackend
    from progressive_stable_diffusion_amd.engine import pack_ffn_stream
    F16, F32 = torch.float16, torch.float32
    be = HipBackend(torch.device("cuda:0"))
    b, hw, c, hid = a.batch, a.hw, 320, 1280
    g = torch.Generator().manual_seed(0)
    rn = lambda shape, s=1.0, dt=F16: (torch.randn(shape, generator=g) * s).to(dt)     # noqa: E731
    x, xres = rn((b, hw, c)), rn((b, hw, c))
    streams = []
    for _ in range(5):       # five sites per step, each with its own weights
        st, b1p = pack_ffn_stream(rn((2 * hid, c), 1 / math.sqrt(c)), rn((2 * hid,), 0.2, F32), rn((c, hid), 1 / math.sqrt(hid)),
                                  rn((c, c, 1, 1), 1 / math.sqrt(c)))
        streams.append((be.to_device(st), be.to_device(b1p)))
    gam, bet, b2, bp = (be.to_device(t) for t in (1 + 0.2 * rn((c,), 1, F32), rn((c,), 0.2, F32), rn((c,), 0.2, F32), rn((c,), 0.2, F32)))
    xd, xr = be.to_device(x), be.to_device(xres)
    out = be.zeros((b, hw, c), F16)
    ws = be.zeros((b * (hw // 32) * 64,), F32)
    flush = be.zeros((150 * 1024 * 1024,), F32)
    variants = [tuple(v.split(":")) for v in a.variants.split(",")]
    res = {v: [] for v in variants}
    for rep in range(a.reps + 2):
        for v in variants:
            os.environ["DADD_FFN_EXP"], os.environ["DADD_FFN_ROT"] = v
            st, b1p = streams[rep % 5]
            if not a.no_flush:
                be.zero_(flush)
            be.prof_begin()
            be.ffn_block(xd, st, gam, bet, b1p, b2, bp, xr, out, gn_ws=ws, gn_nchunk=hw // 32)
            rec = be.prof_end()
            if rep >= 2:
                res[v].append(rec[0][1])
    os.environ.pop("DADD_FFN_EXP"), os.environ.pop("DADD_FFN_ROT")
    flop = 2.0 * b * hw * (c * 2 * hid + hid * c + c * c)
    print(f"ffn_block B={b} HW={hw}: {flop / 1e9:.2f} GF per launch, stream {be.lib.dadd_ffn_block_bytes() / 1e6:.2f} MB per workgroup")
    for v in variants:
        t = sorted(res[v])
        med = t[len(t) // 2]
        print(f"  exp {v[0]} rot {v[1]:>2s}: median {med:7.2f} us  min {t[0]:7.2f}  max {t[-1]:7.2f}   {flop / med / 1e6:7.1f} TF/s-equiv")


if __name__ == "__main__":
    main()
