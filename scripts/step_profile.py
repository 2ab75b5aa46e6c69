#!/usr/bin/env python
"""Live per-kernel profile of ONE UNet denoising step (and optionally the VAE decode): every launch of the
library is issued eagerly with its own begin/end timestamps (backend.prof_begin / prof_end — the clock of
rocprofv3's kernel trace), grouped by kernel name like a `rocprofv3 --stats` summary, plus the per-launch list
with shapes' algorithmic TFLOP/s and GB/s.

    python scripts/step_profile.py [--batch 4] [--image-size 512] [--steps 3] [--vae] [--list] [--out FILE]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--image-size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--vae", action="store_true")
    ap.add_argument("--list", action="store_true", help="also print every launch of the last step in issue order")
    ap.add_argument("--out", default=None)
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                    help="override an engine policy variable for an A/B on one box, e.g. --set LN_STATS_FROM_PRODUCER=False")
    a = ap.parse_args()
    from bench import kernel_table
    from progressive_stable_diffusion_amd import weights as W
    from progressive_stable_diffusion_amd.backend import HipBackend
    from progressive_stable_diffusion_amd import engine as E
    from progressive_stable_diffusion_amd.engine import DdimLoop, UNetPlan, VaeDecoderPlan
    import ast
    for kv in a.set:
        k, v = kv.split("=", 1)
        assert hasattr(E, k), k
        setattr(E, k, ast.literal_eval(v))
    dev = torch.device("cuda:0")
    be = HipBackend(dev)
    side = a.image_size // 8
    gates = {"anatomy": (0.1, 0.9), "disease": (0.9, 0.1), "both": (0.5, 0.5)}
    lines = []

    def emit(s=""):
        print(s)
        lines.append(s)

    def report(title, rec, n):
        rows, total = kernel_table(rec, n)
        emit(f"== {title}: {len(rec) / n:.0f} launches, {total / n / 1e3:.3f} ms of kernel time per pass")
        emit(f"{'kernel':58s} {'calls':>6s} {'avg us':>8s} {'share':>7s} {'TF/s':>7s} {'GB/s':>7s}")
        for t in rows:
            emit(f"{t['name'][:58]:58s} {t['calls_per_step']:6.1f} {t['avg_us']:8.2f} {100 * t['share']:6.2f}% "
                 f"{'' if t['tflops'] is None else format(t['tflops'], '7.1f'):>7s} {t['gbs']:7.0f}")
        if a.list:
            emit("-- launches of the last pass, issue order")
            per = len(rec) // n
            for name, us, flop, byt in rec[-per:]:
                emit(f"  {name[:56]:56s} {us:8.2f} us  {flop / 1e9:9.2f} GF {byt / 1e6:8.2f} MB  "
                     f"{flop / us / 1e6 if flop else 0:7.1f} TF/s {byt / us / 1e3:7.0f} GB/s")

    if not a.vae:
        sd = W.init_state_dict(W.unet_shapes(), 0, gates=gates)
        plan = UNetPlan(be, sd, a.batch, side)
        loop = DdimLoop(plan)
        g = torch.Generator().manual_seed(0)
        plan.set_cond((torch.randn(a.batch, 48, 768, generator=g) * 0.5).to(dev), 0)
        import progressive_stable_diffusion_amd.diffusion_module_ip as DM
        _, ac = DM.build_noise_schedule(DM.DiffusionIPConfig(1000, 0.00085, 0.012))
        loop.prepare(torch.linspace(999, 0, 50, dtype=torch.long), ac)
        be.copy_(plan.lat_in, torch.randn(a.batch, 4, side, side, generator=g).to(dev))
        be.zero_(loop.step)
        loop._one_step(3.0, False, 1.0)
        be.synchronize()
        be.prof_begin()
        for _ in range(a.steps):
            loop._one_step(3.0, False, 1.0)
        report(f"UNet step B={a.batch} {a.image_size}x{a.image_size}", be.prof_end(), a.steps)
    else:
        sd = W.init_state_dict(W.vae_shapes(encoder=False), 0)
        plan = VaeDecoderPlan(be, sd, a.batch, side)
        be.copy_(plan.z_in, torch.randn(a.batch, 4, side, side).to(dev) * 0.18215)
        plan.run()
        be.synchronize()
        be.prof_begin()
        for _ in range(a.steps):
            plan.run()
        report(f"VAE decode B={a.batch} {a.image_size}x{a.image_size}", be.prof_end(), a.steps)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
