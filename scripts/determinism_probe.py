#!/usr/bin/env python
"""Run one UNet step twice on identical inputs, checksum every tensor argument of every recorded op after it ran, and
report the first op whose checksums differ between the two runs (a data race / uninitialised read shows up here).

    python scripts/determinism_probe.py [--batch 13] [--image-size 512] [--repeats 3]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=13)
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--repeats", type=int, default=3)
    a = ap.parse_args()
    from progressive_stable_diffusion_amd import engine as E
    from progressive_stable_diffusion_amd import weights as W
    from progressive_stable_diffusion_amd.backend import HipBackend
    dev = torch.device("cuda:0")
    be = HipBackend(dev)
    side = a.image_size // 8
    sd = W.init_state_dict(W.unet_shapes(), 0, gates={"anatomy": (0.1, 0.9), "disease": (0.9, 0.1), "both": (0.5, 0.5)})
    plan = E.UNetPlan(be, sd, a.batch, side)
    g = torch.Generator().manual_seed(0)
    plan.set_cond((torch.randn(a.batch, 48, 768, generator=g) * 0.5).to(dev), 0)
    lat = torch.randn(a.batch, 4, side, side, generator=g).to(dev)
    t = torch.full((a.batch,), 500, dtype=torch.long, device=dev)
    plan.time_rows(t, plan.temb_rows)
    plan.lam = 3.0
    plan.prepare_attn2(3.0)

    def csum(x):
        v = x.contiguous().view(torch.uint8).to(torch.int64) if x.dtype != torch.float32 else x.contiguous().view(torch.int32).to(torch.int64)
        return int(v.sum().item())

    def run_once():
        be.copy_(plan.lat_in, lat)
        sums = []
        for fn, args, kw in plan.ops:
            fn(*args, **kw)
            be.synchronize()
            ts = [x for x in list(args) + list(kw.values()) if isinstance(x, torch.Tensor)]
            with be.ctx():
                sums.append(tuple(csum(x) for x in ts))
        return sums

    ref = run_once()
    bad = 0
    for r in range(a.repeats):
        cur = run_once()
        for i, (x, y) in enumerate(zip(ref, cur)):
            if x != y:
                fn, args, kw = plan.ops[i]
                shapes = [tuple(v.shape) for v in list(args) + list(kw.values()) if isinstance(v, torch.Tensor)]
                print(f"repeat {r}: first differing op #{i}: {getattr(fn, '__name__', fn)} shapes {shapes} "
                      f"flags {kw.get('flags')} tile {kw.get('tile_m')}x{kw.get('tile_n')} sk {kw.get('splitk')} "
                      f"diff slots {[j for j, (p, q) in enumerate(zip(x, y)) if p != q]}")
                bad += 1
                break
    print("deterministic" if bad == 0 else f"NON-DETERMINISTIC in {bad}/{a.repeats} repeats")


if __name__ == "__main__":
    main()
