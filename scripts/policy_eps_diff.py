#!/usr/bin/env python
"""eps of ONE UNet call at B=4 / 512x512 under two settings of the engine's fusion policies, against each other:
how far a fusion moves a single prediction (the sampler's 10- / 50-step latents amplify this).
    python scripts/policy_eps_diff.py GN_IN_CONV=False [NAME=VALUE ...]     # B = defaults, A = with the overrides"""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    from progressive_stable_diffusion_amd import engine as E
    from progressive_stable_diffusion_amd import weights as W
    from progressive_stable_diffusion_amd.backend import HipBackend
    dev = torch.device("cuda:0")
    be = HipBackend(dev)
    gates = {"anatomy": (0.1, 0.9), "disease": (0.9, 0.1), "both": (0.5, 0.5)}
    sd = W.init_state_dict(W.unet_shapes(), 0, gates=gates)
    g = torch.Generator().manual_seed(0)
    b, side = 4, 64
    x, cond = torch.randn(b, 4, side, side, generator=g), torch.randn(b, 48, 768, generator=g) * 0.5
    t = torch.tensor([999, 700, 333, 20])
    outs = []
    for overrides in ([], sys.argv[1:]):
        saved = {}
        for kv in overrides:
            k, v = kv.split("=", 1)
            saved[k] = getattr(E, k)
            setattr(E, k, ast.literal_eval(v))
        plan = E.UNetPlan(be, sd, b, side)
        eps = plan.forward(x.to(dev), t.to(dev), cond.to(dev), lam=3.0)
        be.synchronize()             # the plan runs on the backend's stream
        outs.append(eps.float().cpu())
        n_launch = len(plan.ops)
        for k, v in saved.items():
            setattr(E, k, v)
        print(f"{'defaults' if not overrides else ' '.join(overrides)}: {n_launch} plan ops, max |eps| {outs[-1].abs().max():.3f}")
        del plan
    d = (outs[0] - outs[1]).abs()
    print(f"max |eps_B - eps_A| = {d.max():.3e}   mean {d.mean():.3e}   (relative to max |eps|: {d.max() / outs[0].abs().max():.3e})")


if __name__ == "__main__":
    main()
