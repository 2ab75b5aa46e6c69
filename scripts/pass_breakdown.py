#!/usr/bin/env python
"""Wall-clock breakdown of one benchmark pass (B=4, 512x512, 50 steps): conditioning prep, K/V + time-row
preparation, the 50 graph replays, VAE decode."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE, weights as W
from progressive_stable_diffusion_amd.config import default_config
from progressive_stable_diffusion_amd.diffusion_module_ip import DiffusionModuleWithIP
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from op_bench import fast_sd

dev = torch.device("cuda:0")
cfg = default_config(**{"dataset.image_size": 512})
shapes = dict(W.unet_shapes()); shapes.update(W.vae_shapes(encoder=False)); shapes.update(W.conditioning_shapes())
sd = fast_sd(shapes)
for k in sd:
    if k.endswith(("anat_gate", "dis_gate")): sd[k] = torch.tensor(0.5)
mod = DiffusionModuleWithIP(cfg, state_dict=sd, device=dev, batch_size=4)
tgt = torch.tensor([0., 1., 2., 3.], device=dev); src = torch.full_like(tgt, 2.0)
pix = torch.rand(1, 3, 224, 224, device=dev) * 2 - 1
lat = torch.randn(4, 4, 64, 64)

def sync(): torch.cuda.synchronize(dev)
def timed(f):
    sync(); t0 = time.perf_counter(); r = f(); sync(); return r, (time.perf_counter() - t0) * 1e3

with torch.no_grad():
    for _ in range(2):
        z = PIPE._ddim_sample_ip(mod, tgt, src, pix, 50, dev, steer_scale=3.0, latents=lat)
        PIPE._latents_to_images(mod, z)
    cond, t_cond = timed(lambda: PIPE._prepare_conditioning(mod, tgt, src, pix))
    _, t_clip = timed(lambda: mod.image_encoder.get_hidden_states(pix.expand(4, -1, -1, -1)))
    loop = mod.ddim_loop(4, 64); plan, be = loop.u, loop.be
    _, t_kv = timed(lambda: plan.set_cond(cond, 0))
    _, t_a2 = timed(lambda: plan.prepare_attn2(3.0))          # (dirty after set_cond: the fold of the fused attn2 sites)
    ts = torch.linspace(999, 0, steps=50, dtype=torch.long)
    _, t_prep = timed(lambda: loop.prepare(ts, mod.alphas_cumprod))
    be.copy_(plan.lat_in, lat.to(dev))
    _, t_loop = timed(lambda: loop.run(3.0, False, 1.0))
    _, t_loop_eager = timed(lambda: loop.run(3.0, False, 1.0, use_graph=False))
    z, t_all = timed(lambda: PIPE._ddim_sample_ip(mod, tgt, src, pix, 50, dev, steer_scale=3.0, latents=lat))
    _, t_dec = timed(lambda: PIPE._latents_to_images(mod, z))
print(f"prepare_conditioning {t_cond:.1f} ms (of which CLIP tower {t_clip:.1f}) | cond K/V projection {t_kv:.1f} | attn2 fold {t_a2:.1f} | "
      f"step tables {t_prep:.1f} | 50 graph replays {t_loop:.1f} ({t_loop/50:.2f}/step; eager {t_loop_eager:.1f}) | "
      f"_ddim_sample_ip total {t_all:.1f} | decode {t_dec:.1f}")
