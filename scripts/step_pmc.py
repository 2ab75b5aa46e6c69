#!/usr/bin/env python
"""Two eager denoising steps of the benchmark workload (B=4, 512x512, lambda=3) — the program run under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes) to get HBM traffic per launch of
the dominant kernel.  No graph replay, no CLIP tower (random conditioning): only the step's kernels."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from progressive_stable_diffusion_amd import weights as W  # noqa: E402
from progressive_stable_diffusion_amd.config import default_config  # noqa: E402
from progressive_stable_diffusion_amd.diffusion_module_ip import DiffusionModuleWithIP  # noqa: E402

dev = torch.device("cuda:0")
tiny = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
            image_size=224, patch_size=14, projection_dim=32)
shapes = dict(W.unet_shapes())
shapes.update(W.conditioning_shapes(clip_hidden=64))
gates = {"anatomy": (0.1, 0.9), "disease": (0.9, 0.1), "both": (0.5, 0.5)}
shapes.update(W.vae_shapes(encoder=False))
sd = W.init_state_dict(shapes, 0, gates=gates)
mod = DiffusionModuleWithIP(default_config(**{"dataset.image_size": 512}), state_dict=sd, device=dev, seed=0,
                            batch_size=4, clip_config=tiny)
loop = mod.ddim_loop(4, 64)
be = loop.be
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    loop.u.set_cond((torch.randn(4, 48, 768, generator=g) * 0.5).to(dev), 0)
    loop.prepare(torch.linspace(999, 0, steps=50, dtype=torch.long, device=dev), mod.alphas_cumprod)
    be.copy_(loop.u.lat_in, torch.randn(4, 4, 64, 64, generator=g).to(dev))
    be.zero_(loop.step)
    for _ in range(2):
        loop._one_step(3.0, False, 1.0)
    be.synchronize()
print("done")
