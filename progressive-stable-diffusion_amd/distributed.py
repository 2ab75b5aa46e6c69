"""Batch sharding of the sampler over the GPUs of one node: one process per GPU, weights replicated,
each rank denoises its own labels, ONE all-gather of the decoded frames (RCCL over xGMI on the
GPU; gloo in the CPU tests).  Latents are never exchanged (BASELINE.json north_star).

The reference has no multi-process inference; the MES sweep it runs on one
device (src/pipelines/inference/inference_pipeline_ip.py:604-612, 646-661) is what gets sharded:
every image's trajectory is independent, so there is no data-path collective before the gather.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; a no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(n_items: int, rank: int, world: int, per_rank: int | None = None) -> Tuple[int, int, int]:
    """Contiguous shard [lo, hi) of ``n_items`` for ``rank`` plus the padded per-rank size.
    Every rank processes ``per_rank`` slots (static plans want one batch size); the tail ranks of a
    sweep that does not divide evenly repeat the last label and the padding is dropped after the
    gather (13 MES labels on 4 GPUs -> 4 slots each, 16 slots, 3 dropped)."""
    if per_rank is None:
        per_rank = (n_items + world - 1) // world
    lo = min(n_items, rank * per_rank)
    hi = min(n_items, lo + per_rank)
    return lo, hi, per_rank


def shard_labels(labels: torch.Tensor, rank: int, world: int, per_rank: int | None = None) -> Tuple[torch.Tensor, int]:
    """This rank's labels padded to ``per_rank`` by repeating the last valid one; returns
    (labels_local, n_valid)."""
    lo, hi, per_rank = shard_bounds(labels.shape[0], rank, world, per_rank)
    loc = labels[lo:hi]
    n_valid = loc.shape[0]
    if n_valid < per_rank:
        fill = loc[-1:] if n_valid > 0 else labels[-1:]
        loc = torch.cat([loc, fill.expand(per_rank - n_valid)])
    return loc.contiguous(), n_valid


def shared_initial_latent(seed: int, channels: int, side: int) -> torch.Tensor:
    """The ONE noise tensor every label starts from (inference_pipeline_ip.py:377-385), drawn from a
    CPU generator so that all ranks (and the CPU oracle) agree bit for bit without a broadcast."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    return torch.randn(1, channels, side, side, generator=g, dtype=torch.float32)


def all_gather_frames(frames: torch.Tensor, n_total: int | None = None) -> torch.Tensor:
    """frames (b,3,H,W) per rank -> (world*b,3,H,W) on every rank, rank-major (= label order);
    ``n_total`` drops the padding slots of an uneven sweep."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = frames
    else:
        world = dist.get_world_size()
        out = torch.empty((world * frames.shape[0], *frames.shape[1:]), dtype=frames.dtype,
                          device=frames.device)
        dist.all_gather_into_tensor(out, frames.contiguous())
    return out if n_total is None else out[:n_total]


def all_gather_frames_u8(be, frames: torch.Tensor, n_total: int | None = None) -> torch.Tensor:
    """The gather as the bench / sweep driver runs it: frames (b,3,H,W) fp32 in [0,1] are packed to uint8 NHWC on the
    GPU first (``dadd_frames_to_u8``: what the writers consume; 3.1 MB instead of 12.6 MB per rank at 512x512, b = 4),
    then ONE all-gather -> (world*b, H, W, 3) uint8 on every rank, rank-major."""
    b, _, h, w = frames.shape
    be.wait_current()
    u8 = be.zeros((b, h, w, 3), torch.uint8)
    be.frames_to_u8(frames.float().contiguous(), u8)
    be.release_to_current()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = u8
    else:
        out = torch.empty((dist.get_world_size() * b, h, w, 3), dtype=torch.uint8, device=u8.device)
        dist.all_gather_into_tensor(out, u8)
    return out if n_total is None else out[:n_total]


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def sweep_plan(n_labels: int, world: int, per_rank: int) -> List[Tuple[int, int]]:
    """[(lo, hi)] per rank — for logging and tests."""
    return [shard_bounds(n_labels, r, world, per_rank)[:2] for r in range(world)]
