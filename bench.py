#!/usr/bin/env python
"""Headline benchmark: images/sec at 512x512, 50 DDIM steps, batch-per-GPU 4 (BASELINE.json).

One "step" = one pass of the hot path over one batch of synthetic input per GPU:
``_ddim_sample_ip`` (conditioning prep + 50 hipGraph-replayed UNet/DDIM steps, delta steering
lambda = 3.0) + ``_latents_to_images`` (VAE decode) + the uint8 pack of the frames, then — for N > 1 — the single
RCCL all-gather of the decoded (uint8) frames.  Workload = BASELINE.json configs[1] per rank (weak scaling: 4 images per GPU).
Weights are seeded random tensors of the SD-1.4 / DADD architecture (no checkpoints offline).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...      (no torchrun environment: starts that same command itself, as a child process)

Rank 0 prints ONE JSON line.  ``roofline`` is measured live: after the timed region the same UNet
step is launched eagerly with every implicit-GEMM launch bracketed by HIP events on its own stream
(graph replays cannot carry per-kernel events); ``cpu_baseline`` times the CPU oracle (a port of the
reference's CPU path) on a bounded sample on rank 0 at N = 1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F16_TFLOPS = 2500.0           # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md chip table
FLOP_PER_IMAGE = 42.56e12          # 50 x 800.8 GF + 2514.5 GF


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3, help="timed passes (each = 4 images x 50 DDIM steps + decode)")
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--image-size", type=int, default=512)
    p.add_argument("--batch", type=int, default=4)
    p.add_argument("--ddim-steps", type=int, default=50)
    p.add_argument("--steer-scale", type=float, default=3.0)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-roofline", action="store_true")
    p.add_argument("--tiny-clip", action="store_true", help="2-layer CLIP tower (debug only)")
    p.add_argument("--launch-selftest", action="store_true",
                   help="ranks only rendezvous (gloo without a GPU), gather one tensor and print a JSON line: the "
                        "CPU test of the N > 1 launcher")
    p.add_argument("--set", action="append", default=[], metavar="NAME=VALUE",
                   help="A/B hook: override an engine policy variable (e.g. LN_STATS_FROM_PRODUCER=False); echoed in config")
    return p.parse_args()


def cpu_baseline(sd, image_size):
    """The oracle (CPU port of the reference path: eager fp32 torch, materialised attention) on the GPU box's
    host cores, the procedure of SURVEY.md §8d / BASELINE.md §3:
      * C1 (BASELINE configs[0]) timed IN FULL, both readings of "guidance 3.0": 1 image, 256x256, 10 DDIM steps +
        decode — (i) routing gates + steer lambda = 3.0, (ii) baseline processors + CFG g = 3.0 (two UNet calls/step);
      * C2 (the metric's config) on a BOUNDED sample: one warm-up eps call, then 2 DDIM steps + 1 VAE decode at
        512x512, B = 1; per image = 25 x (2-step time) + decode time (extrapolated linearly, stated).
    `value` is the C2 figure in the metric's unit."""
    from oracle import sampler as OS
    from oracle.sd_unet import unet_forward
    from oracle.sd_vae import vae_decode
    cores = min(os.cpu_count() or 1, 64)       # eager fp32 torch stops scaling well before 64 threads
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(1, 257, 1024, generator=g)
    c1 = {}
    with torch.no_grad():
        for tag, gates_on, kw in (("gates_lambda3", True, dict(steer_scale=3.0)), ("baseline_cfg3", False, dict(guidance_scale=3.0))):
            ocfg = OS.OracleCfg(image_size=256, use_routing_gates=gates_on)
            lat = torch.randn(1, 4, 32, 32, generator=g)
            t0 = time.perf_counter()
            z = OS.ddim_sample(sd, ocfg, torch.tensor([3.0]), torch.tensor([0.0]), feats, 10, lat, **kw)
            OS.latents_to_images(sd, ocfg, z)
            c1[tag] = time.perf_counter() - t0
        side = image_size // 8
        ocfg = OS.OracleCfg(image_size=image_size)
        _, ac, _, _ = OS.noise_schedule(ocfg)
        cond = OS.prepare_conditioning(sd, ocfg, torch.tensor([3.0]), torch.tensor([2.0]), feats)
        x = torch.randn(1, 4, side, side, generator=g)
        ts = OS.ddim_timesteps(1000, 50)
        unet_forward(sd, x, ts[:1], cond, delta_scale=3.0)                     # warm-up (allocator, thread pool)
        t0 = time.perf_counter()
        for i in range(2):
            eps = unet_forward(sd, x, ts[i:i + 1], cond, delta_scale=3.0)
            x = OS.ddim_update(x, eps, ac, int(ts[i]), int(ts[i + 1]), False)
        t_steps = time.perf_counter() - t0
        t0 = time.perf_counter()
        vae_decode(sd, x / ocfg.latent_scale)
        t_dec = time.perf_counter() - t0
    per_image = 25.0 * t_steps + t_dec
    return {"value": 1.0 / per_image, "unit": "images/sec", "cores": cores, "kind": "port",
            "c1_seconds_per_image": c1, "c1_images_per_sec": {k: 1.0 / v for k, v in c1.items()},
            "sample": f"CPU oracle (fp32 torch, {cores} threads).  C1 timed in full (1 image, 256x256, 10 DDIM steps + decode): "
                      f"gates+lambda=3 {c1['gates_lambda3']:.1f}s, baseline+CFG g=3 {c1['baseline_cfg3']:.1f}s.  C2 ({image_size}x{image_size}, B=1): "
                      f"1 warm-up eps call, then 2 DDIM steps = {t_steps:.2f}s and 1 VAE decode = {t_dec:.2f}s; per image = 25 x "
                      f"{t_steps:.2f} + {t_dec:.2f} = {per_image:.1f}s (50 steps extrapolated linearly from 2)"}


PEAK_HBM_GBS = 8000.0              # HBM3E peak, MI355X_MICROARCH.md chip table


def kernel_table(records, n_steps):
    """Per-kernel-name summary of a list of (name, us, flop, bytes) launch records — the live equivalent of a
    ``rocprofv3 --kernel-trace --stats`` summary (same clock: the dispatch packets' begin/end timestamps)."""
    tab = {}
    for name, us, flop, byt in records:
        t = tab.setdefault(name, {"name": name, "calls": 0, "us": 0.0, "flop": 0.0, "bytes": 0.0})
        t["calls"] += 1
        t["us"] += us
        t["flop"] += flop
        t["bytes"] += byt
    total = sum(t["us"] for t in tab.values()) or 1.0
    rows = sorted(tab.values(), key=lambda t: -t["us"])
    for t in rows:
        t["share"] = t["us"] / total
        t["avg_us"] = t["us"] / t["calls"]
        t["calls_per_step"] = t["calls"] / n_steps
        t["tflops"] = t["flop"] / (t["us"] * 1e-6) / 1e12 if t["flop"] > 0 else None
        t["gbs"] = t["bytes"] / (t["us"] * 1e-6) / 1e9
    return rows, total


def live_roofline(mod, a, side, lat, dev, n_steps=2):
    """Roofline of the kernel with the largest share of GPU time in the UNet step, measured live: right after the
    timed region the same step is launched eagerly ``n_steps`` times (a graph replay cannot carry per-kernel
    events) with every launch's own begin/end timestamps recorded (hipExtLaunchKernelGGL event pair = the clock of
    rocprofv3's kernel trace, so ``avg_launch_us`` is directly comparable with the AverageNs column of the
    committed profiles/*_kernel_stats.csv).  achieved = sum of algorithmic flop (2*M*N*K) or bytes over the
    launches of that kernel / sum of their durations."""
    loop = mod.ddim_loop(a.batch, side)
    be = loop.be
    with torch.no_grad():    # same state as the timed passes: cond projected, tables prepared
        be.copy_(loop.u.lat_in, lat)
        be.zero_(loop.step)
        loop._one_step(a.steer_scale, False, 1.0)      # warm: the eager path's first launch of each kernel
        be.synchronize()
        be.prof_begin()
        for _ in range(n_steps):
            loop._one_step(a.steer_scale, False, 1.0)
        rec = be.prof_end()
    rows, total_us = kernel_table(rec, n_steps)
    if not rows:
        return None
    dom = rows[0]
    mfma = dom["tflops"] is not None
    ach = dom["tflops"] if mfma else dom["gbs"]
    peak = PEAK_F16_TFLOPS if mfma else PEAK_HBM_GBS
    traffic = traffic_src = None
    try:    # HBM-side bytes per launch come from separate rocprofv3 --pmc passes (scripts/pmc_traffic.sh)
        with open(os.path.join(ROOT, "profiles", "traffic_latest.json")) as f:
            tj = json.load(f)
        want = dom["name"].replace(" ", "")
        table = dict(tj.get("per_kernel", {}))
        if "dominant" in tj:
            table.setdefault(tj["dominant"]["kernel"], tj["dominant"])
        # exact kernel name first (counter files and launch tags print template arguments alike), then the bare name
        hit = [v for k, v in table.items() if k.replace(" ", "") == want] or \
              [v for k, v in table.items() if k.split("<")[0] == dom["name"].split("<")[0] and "<" not in dom["name"]]
        if hit:
            traffic, traffic_src = hit[0]["traffic_bytes_per_launch"], "profiles/traffic_latest.json (rocprofv3 --pmc, scripts/pmc_traffic.sh)"
    except (OSError, KeyError, ValueError):
        pass
    gemm = [t for t in rows if t["tflops"] is not None]
    return {"bound": "mfma" if mfma else "hbm", "kernel": dom["name"], "achieved": ach, "peak": peak,
            "unit": "TFLOP/s" if mfma else "GB/s", "frac": ach / peak, "traffic": traffic,
            "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
            "share_of_step_gpu_time": dom["share"], "launches_per_step": dom["calls_per_step"],
            "avg_launch_us": dom["avg_us"],
            "algorithmic_per_launch": (dom["flop"] if mfma else dom["bytes"]) / dom["calls"],
            "timing": "per-launch begin/end timestamps of the dispatch (hipExtLaunchKernelGGL events) over an eager "
                      f"replay of {n_steps} steps right after the timed region; same clock as rocprofv3 --kernel-trace",
            "step_kernel_us": total_us / n_steps, "step_launches": len(rec) / n_steps,
            "all_mfma_kernels_tflops": sum(t["flop"] for t in gemm) / (sum(t["us"] for t in gemm) * 1e-6) / 1e12,
            "top_kernels": [{"name": t["name"], "share": round(t["share"], 4), "calls_per_step": t["calls_per_step"],
                             "avg_us": round(t["avg_us"], 2),
                             "tflops": None if t["tflops"] is None else round(t["tflops"], 1),
                             "gbs": round(t["gbs"], 1)} for t in rows[:10]]}


def launcher_argv(n_gpus: int, argv, port: int):
    """The command the driver itself uses for N > 1 (one rank per GPU, RCCL rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def self_launch(a, argv) -> int:
    """``bench.py --gpus N`` started as ONE plain process (no torchrun environment): start the N ranks as a CHILD
    process — before this process has made any GPU call; nothing here touches HIP, and the child is never exec'ed over
    a process that did — relay its output (rank 0 prints the JSON line) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this pool
    proc = subprocess.run(launcher_argv(a.gpus, argv, port), env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return proc.returncode if (proc.returncode or lines) else 1


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        raise SystemExit(self_launch(a, sys.argv[1:]))
    from progressive_stable_diffusion_amd import distributed as D
    rank, world, local = D.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.launch_selftest:
        got = D.all_gather_frames(torch.full((1, 3, 1, 1), float(rank)))
        t = D.max_over_ranks(float(rank))
        D.barrier()
        if rank == 0:
            print(json.dumps({"metric": "launch-selftest", "n_gpus": world, "ranks_seen": got[:, 0, 0, 0].tolist(),
                              "max_over_ranks": t}), flush=True)
        return
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    if a.set:
        import ast
        from progressive_stable_diffusion_amd import engine as E
        for kv in a.set:
            k, v = kv.split("=", 1)
            if not hasattr(E, k):
                raise SystemExit(f"--set: engine has no policy variable {k}")
            setattr(E, k, ast.literal_eval(v))
    from progressive_stable_diffusion_amd import lib
    from progressive_stable_diffusion_amd import weights as W
    from progressive_stable_diffusion_amd.config import default_config
    from progressive_stable_diffusion_amd.diffusion_module_ip import DiffusionModuleWithIP
    if rank == 0:
        lib.build()
    D.barrier()

    cfg = default_config(**{"dataset.image_size": a.image_size})
    gates = {"anatomy": (0.1, 0.9), "disease": (0.9, 0.1), "both": (0.5, 0.5)}
    tiny = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                image_size=224, patch_size=14, projection_dim=32) if a.tiny_clip else None
    shapes = dict(W.unet_shapes())
    shapes.update(W.vae_shapes(encoder=False))
    shapes.update(W.conditioning_shapes(clip_hidden=64 if tiny else 1024))
    sd = W.init_state_dict(shapes, 0, gates=gates)
    mod = DiffusionModuleWithIP(cfg, state_dict=sd, device=dev, seed=0, batch_size=a.batch, clip_config=tiny)

    side = a.image_size // 8
    n_total = a.batch * world                       # weak scaling: 4 labels per GPU
    labels = torch.linspace(0.0, 3.0, n_total)
    target, _ = D.shard_labels(labels, rank, world, a.batch)
    target = target.to(dev)
    source = torch.full_like(target, 2.0)
    pix = (torch.rand(1, 3, 224, 224, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(dev)
    # inputs resident in HBM before the timed region (an upload from pageable memory would also make the host wait for
    # the previous pass: with everything on the device the host queues pass i+1 while pass i runs)
    lat = D.shared_initial_latent(1234, 4, side).repeat(a.batch, 1, 1, 1).to(dev)

    def one_pass():
        with torch.no_grad():
            z = PIPE._ddim_sample_ip(mod, target, source, pix, a.ddim_steps, dev,
                                     steer_scale=a.steer_scale, latents=lat)
            frames = PIPE._latents_to_images(mod, z)
            return D.all_gather_frames_u8(mod.be, frames)      # uint8 NHWC frames: what the writers consume

    for _ in range(a.warmup):
        one_pass()
    torch.cuda.synchronize(dev)
    D.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    t_queue = 0.0
    for _ in range(a.steps):
        tq = time.perf_counter()
        frames = one_pass()
        t_queue += time.perf_counter() - tq       # host time to QUEUE a pass (no wait for the GPU inside a pass)
    torch.cuda.synchronize(dev)
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)
    assert frames.shape == (n_total, a.image_size, a.image_size, 3) and frames.dtype == torch.uint8

    roof = None
    if not a.no_roofline:
        roof = live_roofline(mod, a, side, lat, dev)

    if rank == 0:
        images = n_total * a.steps
        value = images / elapsed
        out = {
            "metric": "images/sec at 512x512, 50 DDIM steps, bs/GPU=4", "value": value, "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
            "host_queue_ms_per_step": t_queue / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
            "data": "synthetic (seeded random-init SD-1.4/DADD weights, CPU-seeded noise, random structure image)",
            "config": {"workload": f"{a.image_size}x{a.image_size}, {a.ddim_steps} DDIM steps, bs={a.batch}/GPU, "
                                   f"delta-steer lambda={a.steer_scale}, routing gates on, conditioning prep + "
                                   "VAE decode included", "global_batch": n_total,
                       "parallelism": f"batch-shard x{world} + 1 all-gather of uint8 frames",
                       **({"policy_overrides": a.set} if a.set else {})},
            "achieved_tflops_whole_job": value * FLOP_PER_IMAGE / 1e12 if a.image_size == 512 else None,
            "frac_of_mfma_peak_whole_job": (value * FLOP_PER_IMAGE / 1e12) / (PEAK_F16_TFLOPS * world)
            if a.image_size == 512 else None,
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, a.image_size)
        print(json.dumps(out), flush=True)
    D.barrier()


if __name__ == "__main__":
    main()
