#!/usr/bin/env python
"""Per-op timing over the REAL layer shapes of one UNet step / one VAE decode (B=4, 512x512).

Builds the engine plans, groups the recorded launches by shape signature and times each signature in
isolation with HIP events on the launch stream; for the implicit GEMM it sweeps the tuning knobs
(tile_m, split-K, one/two K tiles in flight) and prints the best variant per shape.  Output:
a table on stdout and gpurun_out/<tag>/op_bench.json.   usage: python scripts/op_bench.py [tag] [--vae]
"""
from __future__ import annotations

import json
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from progressive_stable_diffusion_amd import engine as E  # noqa: E402
from progressive_stable_diffusion_amd import lib as L  # noqa: E402
from progressive_stable_diffusion_amd.backend import HipBackend  # noqa: E402

F16, F32 = torch.float16, torch.float32


def fast_sd(shapes, seed=0):
    """Cheap random weights (values are irrelevant for timing, but must be random: zero operands
    raise the clock and flatter the numbers — cdna_hip_programming.md §5.4 rule 25)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, s in shapes.items():
        if len(s) == 0:
            sd[k] = torch.tensor(0.5)
        else:
            fan = 1
            for d in s[1:]:
                fan *= d
            sd[k] = (torch.rand(s, generator=g) * 2 - 1) * (1.0 / max(1.0, fan) ** 0.5)
    return sd


def timeit(be, fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(be.stream):
        s.record()
        for _ in range(iters):
            fn()
        e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters      # us


def igemm_sig(a, k):
    x, w, out = a[0], a[1], a[2]
    m = out.shape[0] * out.shape[1] * out.shape[2]
    return ("igemm", m, w.shape[0], w.shape[1], k.get("taps", 1), k.get("stride", 1), k.get("ups", 0),
            k.get("x2") is not None, k.get("flags", 0) & 15)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "opbench"
    do_vae = "--vae" in sys.argv
    out_dir = os.path.join(ROOT, "gpurun_out", tag)
    os.makedirs(out_dir, exist_ok=True)
    from progressive_stable_diffusion_amd import weights as W
    be = HipBackend(torch.device("cuda:0"))
    B, S = 4, 64
    if do_vae:
        plan = E.VaeDecoderPlan(be, fast_sd(W.vae_shapes(encoder=False)), B, S)
    else:
        plan = E.UNetPlan(be, fast_sd(W.unet_shapes()), B, S)
        cond = torch.randn(B, 48, 768, device="cuda:0") * 0.5
        plan.set_cond(cond, 0)
        plan.lam = 3.0
    be.synchronize()

    groups = OrderedDict()
    for fn, a, k in plan.ops:
        name = getattr(fn, "__name__", str(fn))
        if name == "igemm":
            sig = igemm_sig(a, k)
        else:
            shp = tuple(tuple(t.shape) for t in a if isinstance(t, torch.Tensor))[:2]
            sig = (name,) + shp
        groups.setdefault(sig, []).append((fn, a, k))

    partial = be.zeros((64 * 1024 * 1024,), F32)      # 256 MB: enough for any split-K slab here
    rows, total_cur, total_best = [], 0.0, 0.0
    for sig, lst in groups.items():
        fn, a, k = lst[0]
        cnt = len(lst)
        if sig[0] != "igemm":
            t = timeit(be, lambda: fn(*a, **k))
            rows.append({"op": sig[0], "sig": str(sig[1:]), "count": cnt, "us": t, "total_us": t * cnt})
            total_cur += t * cnt
            total_best += t * cnt
            continue
        _, m, n, kk, taps, stride, ups, cat, flags = sig
        nkt = kk // 64
        flop = 2.0 * m * n * kk
        t_cur = timeit(be, lambda: fn(*a, **k))
        best = (t_cur, "current")
        variants = []
        geglu = bool(flags & 8)
        for tm in (128, 64):
            for sk in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
                if sk > 1 and (geglu or nkt // sk < 2):
                    continue
                tiles = -(-m // tm) * -(-n // (k.get("tile_n") or 160))
                if sk > 1 and tiles * sk > 2048:
                    continue
                for shallow in ((0, 2, 3) if tm == 128 else (0, 1)):  # 2 = LDS-DMA ring kernel, 3 = persistent ring
                    if shallow == 3 and (sk > 1 or ups or tiles <= 256):
                        continue
                    tune = {0: L.TUNE_NODMA, 1: L.TUNE_NODMA | L.TUNE_SHALLOW, 2: 0, 3: L.TUNE_PERSIST}[shallow]
                    kw = dict(k)
                    kw.update(tile_m=tm, splitk=sk, partial=partial if sk > 1 else None,
                              flags=(k.get("flags", 0) & 15) | tune)
                    if geglu:
                        kw["tile_n"] = 128
                    try:
                        t = timeit(be, lambda: fn(*a, **kw), iters=10, warm=2)
                    except Exception as ex:      # noqa: BLE001
                        print("variant failed", sig, tm, sk, shallow, ex)
                        continue
                    variants.append((t, tm, sk, shallow))
                    if t < best[0]:
                        best = (t, f"tm{tm} sk{sk} {('reg2', 'reg1', 'dma', 'dmaP')[shallow]}")
        variants.sort()
        rows.append({"op": "igemm", "M": m, "N": n, "K": kk, "taps": taps, "stride": stride, "ups": ups,
                     "cat": cat, "flags": flags, "count": cnt, "us_current": t_cur,
                     "tflops_current": flop / t_cur / 1e6, "us_best": best[0], "best": best[1],
                     "tflops_best": flop / best[0] / 1e6, "total_us": t_cur * cnt,
                     "top3": [(round(v[0], 1), v[1], v[2], v[3]) for v in variants[:3]]})
        total_cur += t_cur * cnt
        total_best += best[0] * cnt

    rows.sort(key=lambda r: -r["total_us"])
    print(f"{'op':8s} {'shape':44s} {'cnt':>3s} {'us':>8s} {'tot_us':>9s} {'TF/s':>7s} | best")
    for r in rows:
        if r["op"] == "igemm":
            shp = f"M{r['M']} N{r['N']} K{r['K']} t{r['taps']} s{r['stride']} u{r['ups']} f{r['flags']}"
            print(f"{'igemm':8s} {shp:44s} {r['count']:3d} {r['us_current']:8.1f} {r['total_us']:9.1f} "
                  f"{r['tflops_current']:7.1f} | {r['us_best']:7.1f}us {r['tflops_best']:6.1f}TF {r['best']}  {r['top3']}")
        else:
            print(f"{r['op']:8s} {r['sig'][:44]:44s} {r['count']:3d} {r['us']:8.1f} {r['total_us']:9.1f}")
    print(f"sum of isolated op times: current {total_cur / 1e3:.2f} ms, with best igemm variants {total_best / 1e3:.2f} ms")
    json.dump({"rows": rows, "total_cur_us": total_cur, "total_best_us": total_best},
              open(os.path.join(out_dir, "op_bench_vae.json" if do_vae else "op_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
