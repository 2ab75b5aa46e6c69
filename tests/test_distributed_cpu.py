"""world_size-2 gloo checks of the N>1 path: label sharding, the shared initial latent, the single
all-gather of frames and the max-over-ranks timing reduction (CPU; the GPU run uses RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from progressive_stable_diffusion_amd import distributed as D


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_frames(labels, latent):
    """Stand-in for sample+decode: a deterministic function of (label, shared latent)."""
    base = latent.mean() + latent.flatten()[:12].reshape(1, 3, 2, 2)
    return base + labels.reshape(-1, 1, 1, 1) * torch.ones(1, 3, 2, 2)


def _worker(rank, world, port, n_labels, per_rank, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    labels = torch.linspace(0, 3, n_labels)
    loc, n_valid = D.shard_labels(labels, r, w, per_rank)
    assert loc.shape[0] == per_rank
    lat = D.shared_initial_latent(1234, 4, 8)
    frames = _fake_frames(loc, lat)
    D.barrier()
    allf = D.all_gather_frames(frames, n_total=n_labels)
    from tests.torch_backend import TorchRefBackend
    unit = (frames - frames.min()) / (frames.max() - frames.min() + 1e-6)        # [0,1] frames for the uint8 gather
    all8 = D.all_gather_frames_u8(TorchRefBackend(), unit.expand(-1, -1, 2, 2).contiguous(), n_total=n_labels)
    t = D.max_over_ranks(1.0 + rank)
    torch.save({"frames": allf, "u8": all8, "t": t, "n_valid": n_valid, "lat": lat}, f"{out_dir}/r{rank}.pt")
    dist.destroy_process_group()


@pytest.mark.parametrize("n_labels,per_rank", [(8, 4), (5, 4), (13, 8)])
def test_two_rank_sweep_equals_single_process(tmp_path, n_labels, per_rank):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n_labels, per_rank, str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(tmp_path / f"r{r}.pt", weights_only=True) for r in range(world)]
    lat = D.shared_initial_latent(1234, 4, 8)
    expect = _fake_frames(torch.linspace(0, 3, n_labels), lat)
    for o in outs:
        assert torch.equal(o["lat"], lat)                       # every rank starts from the same noise
        assert o["frames"].shape == (n_labels, 3, 2, 2)
        assert torch.equal(o["frames"], expect)                 # shard -> gather == single process
        assert o["u8"].shape == (n_labels, 2, 2, 3) and o["u8"].dtype == torch.uint8
        assert torch.equal(o["u8"], outs[0]["u8"])              # every rank holds the same gathered uint8 frames
        assert o["t"] == 2.0                                    # MAX over ranks
    assert sum(o["n_valid"] for o in outs) == n_labels


def test_shard_bounds_and_padding():
    assert D.sweep_plan(13, 4, 4) == [(0, 4), (4, 8), (8, 12), (12, 13)]
    assert D.sweep_plan(13, 8, 4) == [(0, 4), (4, 8), (8, 12), (12, 13)] + [(13, 13)] * 4
    lab = torch.linspace(0, 3, 13)
    loc, n = D.shard_labels(lab, 3, 4, 4)
    assert n == 1 and loc.tolist() == [3.0] * 4
    loc, n = D.shard_labels(lab, 6, 8, 4)                       # a rank with no valid label
    assert n == 0 and loc.tolist() == [3.0] * 4
    assert D.all_gather_frames(torch.ones(2, 3, 1, 1)).shape == (2, 3, 1, 1)   # single process
    assert D.max_over_ranks(3.5) == 3.5


def test_bench_launches_its_own_ranks():
    """``python bench.py --gpus 2`` with no torchrun environment (what the driver's multi-GPU tier runs) must start the
    two ranks itself — as a child ``python -m torch.distributed.run`` process — and relay rank 0's single JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    argv = bench.launcher_argv(8, ["--gpus", "8", "--steps", "3", "--warmup", "1"], 29511)
    assert argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in argv
    assert argv[argv.index("--master-addr") + 1] == "127.0.0.1" and argv[argv.index("--master-port") + 1] == "29511"
    assert argv[-6:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"] and argv[-7].endswith("bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-selftest"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["ranks_seen"] == [0.0, 1.0] and rec["max_over_ranks"] == 1.0
