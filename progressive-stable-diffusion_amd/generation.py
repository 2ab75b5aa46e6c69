"""Batched generation drivers and the asynchronous frame sink: the production callers of the
sampler — the evaluation pipeline's ``generate_all`` (src/pipelines/evaluation/evaluation_pipeline.py:842-975) and
the data-augmentation main loop (src/pipelines/inference/inference_pipeline_ip_data_augment.py:313-341,432-498).

Same job semantics as the reference (every source image -> the three OTHER MES classes, ``batch_images`` sources per
batch, one structure image and one noise draw per sample, resume by existing output file, results per target
class), re-designed around a static engine plan:
  * every batch is padded to ONE plan size (``batch_images * 3`` slots, the last batch repeats its last sample) so the
    captured step graph and the decoder plan are built once;
  * frames leave the GPU as uint8 NHWC (``dadd_frames_to_u8``: 4x fewer bytes than fp32) into a ring of pinned host
    buffers on a dedicated copy stream, so the D2H transfer and the BMP/PNG encoding of batch i overlap the denoising
    of batch i+1;
  * writer threads (PIL releases the GIL while encoding) take zero-copy numpy views of the pinned buffers.
"""
from __future__ import annotations

import threading
from collections import OrderedDict
from concurrent.futures import Future, ThreadPoolExecutor
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import inference_pipeline_ip as PIPE

ALL_MES_CLASSES = [0, 1, 2, 3]
IMAGE_EXTENSIONS = {".bmp", ".png", ".jpg", ".jpeg", ".tif", ".tiff"}


@dataclass
class GenerationJob:
    """One source image -> one target MES class (evaluation_pipeline.py:83-90)."""
    source_path: Path
    source_label: int
    target_label: int


def _collect_jobs(data_roots: Sequence[Path], max_per_class: int = 0) -> List[GenerationJob]:
    """(evaluation_pipeline.py:842-865) every image of every class directory -> the 3 other classes."""
    jobs: List[GenerationJob] = []
    for data_root in data_roots:
        for cls in ALL_MES_CLASSES:
            cls_dir = Path(data_root) / str(cls)
            if not cls_dir.is_dir():
                continue
            paths = sorted(p for p in cls_dir.iterdir() if p.suffix.lower() in IMAGE_EXTENSIONS)
            if max_per_class > 0:
                paths = paths[:max_per_class]
            for p in paths:
                jobs += [GenerationJob(p, cls, t) for t in ALL_MES_CLASSES if t != cls]
    return jobs


def _collect_pending_jobs(data_root: Path, train_dst: Path) -> List[dict]:
    """(inference_pipeline_ip_data_augment.py:313-341) {path, stem, source_mes, targets}; only targets whose
    ``<stem>_generated.bmp`` does not exist yet (resume-friendly)."""
    jobs: List[dict] = []
    for mes in ALL_MES_CLASSES:
        class_dir = Path(data_root) / "train" / str(mes)
        if not class_dir.exists():
            continue
        for img in sorted(class_dir.glob("*.bmp")):
            targets = [tc for tc in ALL_MES_CLASSES
                       if tc != mes and not (Path(train_dst) / str(tc) / f"{img.stem}_generated.bmp").exists()]
            if targets:
                jobs.append({"path": img, "stem": img.stem, "source_mes": mes, "targets": targets})
    return jobs


def _apply_gaussian_blur(images: Tensor, kernel_size: int = 15, sigma: float = 5.0) -> Tensor:
    """(evaluation_pipeline.py:339-352) separable reflect-padded Gaussian on the host copy of the structure image."""
    import torch.nn.functional as F
    x = torch.arange(kernel_size, dtype=images.dtype) - kernel_size // 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    pad = kernel_size // 2
    b = F.pad(images, (pad, pad, pad, pad), mode="reflect")
    b = F.conv2d(b, g.view(1, 1, 1, -1).expand(3, 1, 1, -1), groups=3)
    b = F.conv2d(b, g.view(1, 1, -1, 1).expand(3, 1, -1, 1), groups=3)
    return b.clamp(0, 1)


def _load_structure_image(image_path: Path, device: torch.device, target_size: int = 256, apply_blur: bool = False,
                          blur_kernel_size: int = 7, blur_sigma: float = 2.0) -> Tensor:
    """(evaluation_pipeline.py:355-371) one image -> CLIP-ready (1, 3, 224, 224) on ``device``."""
    import numpy as np
    from PIL import Image
    pil = Image.open(image_path).convert("RGB").resize((target_size, target_size), Image.BILINEAR)
    display = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).float() / 255.0
    if apply_blur:
        display = _apply_gaussian_blur(display[None], blur_kernel_size, blur_sigma)[0]
        # the reference hands the blurred tensor to CLIPImageProcessor as a PIL image: ToPILImage = mul(255).byte()
        # (evaluation_pipeline.py:364-369), i.e. truncation to 8 bits before the CLIP preprocessing
        display = display.mul(255).to(torch.uint8).float() / 255.0
    return PIPE._clip_preprocess(display).to(device)


def _write_image(arr, save_path: Path) -> Path:
    """HWC uint8 array -> file; the format follows the suffix (BMP for the augmentation set, PNG otherwise)."""
    from PIL import Image
    save_path.parent.mkdir(parents=True, exist_ok=True)
    Image.fromarray(arr).save(save_path, format="BMP" if save_path.suffix.lower() == ".bmp" else None)
    return save_path


class FrameSink:
    """GPU frames -> files without stalling the sampler.

    ``submit(frames, paths)`` packs fp32 NCHW [0,1] frames to uint8 NHWC on the backend stream, copies them to a
    pinned host slot on a COPY stream (ordered behind the pack by an event) and hands the slot to the writer pool once
    the copy's event has completed; the next batch's kernels run meanwhile.  ``slots`` bounds the batches in flight: a
    slot is reused only after its files are written.  Existing files are skipped when ``resume`` (the reference's
    resume-by-existing-file)."""

    def __init__(self, be, batch: int, height: int, width: int, *, workers: int = 8, slots: int = 3,
                 resume: bool = True):
        self.be, self.resume = be, resume
        self.shape = (batch, height, width, 3)
        on_gpu = be.device.type == "cuda"
        self.dev_u8 = [be.zeros(self.shape, torch.uint8) for _ in range(slots)]
        self.host = [torch.empty(self.shape, dtype=torch.uint8, pin_memory=on_gpu) for _ in range(slots)]
        self.copy_stream = torch.cuda.Stream(device=be.device) if on_gpu else None
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.busy: List[List[Future]] = [[] for _ in range(slots)]
        self.slot = 0
        self.written: List[Path] = []
        self.skipped = 0
        self._lock = threading.Lock()

    def _drain(self, i: int):
        for f in self.busy[i]:
            p = f.result()
            if p is not None:
                with self._lock:
                    self.written.append(p)
        self.busy[i] = []

    def submit(self, frames: Tensor, paths: Sequence[Optional[Path]]):
        """frames (b <= batch, 3, H, W) fp32 in [0,1] on the device; ``paths[k] is None`` drops frame k (padding)."""
        b = frames.shape[0]
        if b > self.shape[0] or tuple(frames.shape[2:]) != self.shape[1:3]:
            raise ValueError(f"frames {tuple(frames.shape)} do not fit the sink {self.shape}")
        i = self.slot
        self.slot = (self.slot + 1) % len(self.host)
        self._drain(i)                                  # the slot's previous files are on disk
        be = self.be
        be.wait_current()
        be.frames_to_u8(frames.float().contiguous(), self.dev_u8[i][:b])
        if self.copy_stream is not None:
            ev = torch.cuda.Event()
            ev.record(be.stream)
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(ev)
                self.host[i][:b].copy_(self.dev_u8[i][:b], non_blocking=True)
                done = torch.cuda.Event()
                done.record(self.copy_stream)
        else:
            self.host[i][:b].copy_(self.dev_u8[i][:b])
            done = None
        view = self.host[i].numpy()

        def job(k, path):
            if done is not None:
                done.synchronize()                      # the writer waits for the copy, not the sampler
            return _write_image(view[k], path)
        for k, path in enumerate(paths[:b]):
            if path is None:
                continue
            path = Path(path)
            if self.resume and path.exists():
                self.skipped += 1
                continue
            self.busy[i].append(self.pool.submit(job, k, path))

    def close(self) -> List[Path]:
        for i in range(len(self.busy)):
            self._drain(i)
        self.pool.shutdown(wait=True)
        return sorted(self.written)


def _pad_batch(t: Tensor, size: int) -> Tensor:
    """Repeat the last sample up to the plan's batch size (static plans; the padding slots are dropped later)."""
    if t.shape[0] == size:
        return t
    return torch.cat([t, t[-1:].expand(size - t.shape[0], *t.shape[1:])], dim=0)


def _sample_padded(module, targets: List[float], sources: List[float], structs: List[Tensor], plan_batch: int,
                   device, **kw) -> Tuple[Tensor, int]:
    """One padded ``_ddim_sample_batched`` + decode on the device -> (frames (plan_batch,3,H,W) in [0,1], n_valid)."""
    n = len(targets)
    tgt = _pad_batch(torch.tensor(targets, dtype=torch.float32, device=device), plan_batch)
    src = _pad_batch(torch.tensor(sources, dtype=torch.float32, device=device), plan_batch)
    pix = _pad_batch(torch.cat(structs, dim=0), plan_batch)
    # the reference draws randn(n, ...) for the n real samples of a batch (data_augment:239, evaluation_pipeline:506):
    # draw exactly that and pad the LATENTS, so a ragged last batch consumes the generator as the reference does
    side = kw.pop("latent_side", None) or module.cfg.dataset.image_size // 8
    noise = torch.randn(n, module.cfg.model.latent_channels, side, side, device=device, dtype=torch.float32)
    latents = PIPE._ddim_sample_batched(module, tgt, src, pix, device=device, latents=_pad_batch(noise, plan_batch), **kw)
    return PIPE._latents_to_images(module, latents), n


@torch.no_grad()
def generate_all(module, jobs: List[GenerationJob], cfg, device: torch.device, batch_images: int = 4,
                 sampling_steps: int = 50, image_scale: float = 1.0, steer_scale: float = 0.0,
                 guidance_scale: float = 1.0, eta: float = 0.0, apply_blur: bool = False, seed: int = 42,
                 use_fp16: bool = False, *, use_graph: bool = True) -> Dict[int, Tensor]:
    """(evaluation_pipeline.py:868-975) all jobs -> {target_class: (N,3,H,W) fp32 CPU frames in [0,1]}; jobs sorted by
    (source, target), ``batch_images`` sources per batch, the seed set ONCE.  ``use_fp16`` is accepted for
    signature parity: the engine always stores fp16 / accumulates fp32."""
    bk, bs_sigma = getattr(cfg.model, "blur_kernel_size", 7), getattr(cfg.model, "blur_sigma", 2.0)
    jobs_sorted = sorted(jobs, key=lambda j: (str(j.source_path), j.target_label))
    buckets: "OrderedDict[Path, List[GenerationJob]]" = OrderedDict()
    for j in jobs_sorted:
        buckets.setdefault(j.source_path, []).append(j)
    source_list = list(buckets.items())
    plan_batch = batch_images * (len(ALL_MES_CLASSES) - 1)
    size = cfg.dataset.image_size
    result: Dict[int, List[Tensor]] = {c: [] for c in ALL_MES_CLASSES}
    PIPE._set_seed(seed)
    for i in range(0, len(source_list), batch_images):
        targets, sources, structs = [], [], []
        for src_path, src_jobs in source_list[i:i + batch_images]:
            struct = _load_structure_image(src_path, device, size, apply_blur, bk, bs_sigma)
            for j in src_jobs:
                targets.append(float(j.target_label))
                sources.append(float(j.source_label))
                structs.append(struct)
        frames, n = _sample_padded(module, targets, sources, structs, plan_batch, device,
                                   sampling_steps=sampling_steps, eta=eta, image_scale=image_scale,
                                   steer_scale=steer_scale, guidance_scale=guidance_scale, use_graph=use_graph)
        frames = frames[:n].float().cpu()
        for k in range(n):
            result[int(targets[k])].append(frames[k:k + 1])
    return {c: (torch.cat(v, dim=0) if v else torch.zeros(0, 3, size, size)) for c, v in result.items()}


@torch.no_grad()
def augment_dataset(module, data_root: Path, train_dst: Path, device: torch.device, *, batch_images: int = 4,
                    sampling_steps: int = 50, eta: float = 0.0, image_scale: float = 1.0, steer_scale: float = 0.0,
                    guidance_scale: float = 1.0, save_workers: int = 8, image_size: Optional[int] = None,
                    use_graph: bool = True) -> Dict[int, int]:
    """(inference_pipeline_ip_data_augment.py:432-498) every pending (source, target) pair of ``data_root/train`` ->
    ``train_dst/<target>/<stem>_generated.bmp`` through the frame sink; returns the per-class counts."""
    jobs = _collect_pending_jobs(data_root, train_dst)
    size = image_size or module.cfg.dataset.image_size
    plan_batch = batch_images * (len(ALL_MES_CLASSES) - 1)
    counts = {m: 0 for m in ALL_MES_CLASSES}
    if not jobs:
        return counts
    sink = FrameSink(module.be, plan_batch, size, size, workers=save_workers)
    cls_of: Dict[Path, int] = {}
    try:
        for i in range(0, len(jobs), batch_images):
            targets, sources, structs, paths = [], [], [], []
            for job in jobs[i:i + batch_images]:
                struct = _load_structure_image(job["path"], device, size)
                for tc in job["targets"]:
                    targets.append(float(tc))
                    sources.append(float(job["source_mes"]))
                    structs.append(struct)
                    paths.append(Path(train_dst) / str(tc) / f"{job['stem']}_generated.bmp")
                    cls_of[paths[-1]] = tc
            frames, n = _sample_padded(module, targets, sources, structs, plan_batch, device,
                                       sampling_steps=sampling_steps, eta=eta, image_scale=image_scale,
                                       steer_scale=steer_scale, guidance_scale=guidance_scale, use_graph=use_graph,
                                       latent_side=size // 8)
            sink.submit(frames[:n], paths)
    finally:            # a sampler error must not leak the writer pool / pinned slots or swallow a writer's exception
        written = sink.close()
    for p in written:   # counts = files actually written (resume skips and failed writes are not counted)
        counts[cls_of[Path(p)]] += 1
    return counts
