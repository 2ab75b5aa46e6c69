"""CPU checks of the batched generation drivers and the frame sink through the TEST-ONLY torch
backend: job collection and bucketing as the reference does it, resume by existing file, padding of ragged batches
to the static plan size, the uint8 pack and the writer pool.  The same drivers run on the HIP backend in
test_gpu_parity.py."""
import numpy as np
import pytest
import torch
from PIL import Image

from progressive_stable_diffusion_amd import generation as G
from progressive_stable_diffusion_amd import weights as W
from progressive_stable_diffusion_amd.config import default_config
from progressive_stable_diffusion_amd.diffusion_module_ip import DiffusionModuleWithIP
from tests.torch_backend import TorchRefBackend

TINY_CLIP = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=1,
                 image_size=224, patch_size=14, projection_dim=32)


def _dataset(root, per_class=(2, 1, 0, 1), ext=".bmp"):
    rs = np.random.RandomState(0)
    for cls, n in enumerate(per_class):
        d = root / str(cls)
        d.mkdir(parents=True, exist_ok=True)
        for i in range(n):
            Image.fromarray((rs.rand(40, 48, 3) * 255).astype("uint8")).save(d / f"img{cls}_{i}{ext}")
    return root


def test_job_collection_and_resume(tmp_path):
    root = _dataset(tmp_path / "eval")
    (root / "1" / "notes.txt").write_text("x")
    jobs = G._collect_jobs([root])
    assert len(jobs) == 4 * 3 and all(j.target_label != j.source_label for j in jobs)
    assert [j.target_label for j in jobs[:3]] == [1, 2, 3] and jobs[0].source_path.name == "img0_0.bmp"
    assert len(G._collect_jobs([root], max_per_class=1)) == 3 * 3
    aug = _dataset(tmp_path / "aug" / "train").parent
    dst = tmp_path / "dst"
    pend = G._collect_pending_jobs(aug, dst)
    assert sum(len(j["targets"]) for j in pend) == 12
    (dst / "2").mkdir(parents=True)
    (dst / "2" / "img0_0_generated.bmp").write_bytes(b"x")          # already generated: skipped on resume
    pend = G._collect_pending_jobs(aug, dst)
    assert sum(len(j["targets"]) for j in pend) == 11 and pend[0]["targets"] == [1, 3]


def test_frame_sink_writes_exact_bytes_and_resumes(tmp_path):
    be = TorchRefBackend()
    sink = G.FrameSink(be, 4, 16, 20, workers=2, slots=2)
    g = torch.Generator().manual_seed(1)
    batches = [torch.rand(4, 3, 16, 20, generator=g), torch.rand(3, 3, 16, 20, generator=g), torch.rand(4, 3, 16, 20, generator=g)]
    paths = [[tmp_path / f"b{i}" / f"f{k}.bmp" for k in range(4)] for i in range(3)]
    paths[1] = paths[1][:3]
    paths[2][1] = None                                              # padding slot: dropped
    (tmp_path / "b0").mkdir()
    (tmp_path / "b0" / "f2.bmp").write_bytes(b"old")                # resume: left alone
    for fr, ps in zip(batches, paths):
        sink.submit(fr, ps)
    written = sink.close()
    assert len(written) == 4 + 3 + 3 - 1 and sink.skipped == 1
    assert (tmp_path / "b0" / "f2.bmp").read_bytes() == b"old" and not (tmp_path / "b2" / "f1.bmp").exists()
    for i, fr in enumerate(batches):
        for k in range(fr.shape[0]):
            p = tmp_path / f"b{i}" / f"f{k}.bmp"
            if p in written:
                got = torch.from_numpy(np.asarray(Image.open(p)).copy()).permute(2, 0, 1)
                assert torch.equal(got, fr[k].mul(255).to(torch.uint8)), p        # _tensor_to_bmp semantics
    with pytest.raises(ValueError):
        G.FrameSink(be, 2, 16, 20).submit(torch.rand(3, 3, 16, 20), [None] * 3)


def test_generate_all_and_augment_through_the_engine(tmp_path):
    """End to end on the torch backend at 64x64 / 2 steps: ragged last batch padded to the plan size, results per
    target class, augmentation files named and resumed as the reference does."""
    shapes = dict(W.unet_shapes())
    shapes.update(W.vae_shapes(encoder=False))
    shapes.update(W.conditioning_shapes(clip_hidden=64, clip_proj=32))
    shapes.update(W.clip_shapes(TINY_CLIP))
    sd = W.init_state_dict(shapes, 0)
    cfg = default_config(**{"dataset.image_size": 64})
    mod = DiffusionModuleWithIP(cfg, state_dict=sd, device="cpu", batch_size=6, clip_config=TINY_CLIP, backend=TorchRefBackend())
    root = _dataset(tmp_path / "eval", per_class=(1, 1, 0, 1))
    jobs = G._collect_jobs([root])
    assert len(jobs) == 9
    res = G.generate_all(mod, jobs, cfg, torch.device("cpu"), batch_images=2, sampling_steps=2, steer_scale=2.0, seed=3, use_graph=False)
    assert {c: tuple(v.shape) for c, v in res.items()} == {0: (2, 3, 64, 64), 1: (2, 3, 64, 64), 2: (3, 3, 64, 64), 3: (2, 3, 64, 64)}
    assert all(float(v.min()) >= 0.0 and float(v.max()) <= 1.0 for v in res.values())
    assert len(mod._unets) == 1 and (6, 8) in mod._unets            # ONE plan: the ragged batch was padded to 6 slots
    aug = _dataset(tmp_path / "aug" / "train", per_class=(1, 0, 1, 0)).parent
    dst = tmp_path / "dst"
    counts = G.augment_dataset(mod, aug, dst, torch.device("cpu"), batch_images=2, sampling_steps=1, steer_scale=1.0, save_workers=2, use_graph=False)
    assert counts == {0: 1, 1: 2, 2: 1, 3: 2}
    files = sorted(str(p.relative_to(dst)) for p in dst.rglob("*.bmp"))
    assert files == ["0/img2_0_generated.bmp", "1/img0_0_generated.bmp", "1/img2_0_generated.bmp", "2/img0_0_generated.bmp",
                     "3/img0_0_generated.bmp", "3/img2_0_generated.bmp"]
    assert np.asarray(Image.open(dst / "1" / "img0_0_generated.bmp")).shape == (64, 64, 3)
    assert G.augment_dataset(mod, aug, dst, torch.device("cpu"), batch_images=2, sampling_steps=1, use_graph=False) == {0: 0, 1: 0, 2: 0, 3: 0}


def test_blurred_structure_image_is_quantised_like_the_reference(tmp_path):
    """with ``apply_blur`` the reference converts the blurred float image to PIL (``ToPILImage`` =
    ``mul(255).byte()``) before CLIPImageProcessor (evaluation_pipeline.py:355-371): the CLIP input is the
    preprocessing of the 8-bit-truncated blur, not of the float blur."""
    from progressive_stable_diffusion_amd import inference_pipeline_ip as PIPE
    rs = np.random.RandomState(3)
    Image.fromarray((rs.rand(50, 60, 3) * 255).astype("uint8")).save(tmp_path / "a.bmp")
    got = G._load_structure_image(tmp_path / "a.bmp", torch.device("cpu"), 64, apply_blur=True, blur_kernel_size=7, blur_sigma=2.0)
    pil = Image.open(tmp_path / "a.bmp").convert("RGB").resize((64, 64), Image.BILINEAR)
    disp = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).float() / 255.0
    blur = G._apply_gaussian_blur(disp[None], 7, 2.0)[0]
    want = PIPE._clip_preprocess(blur.mul(255).to(torch.uint8).float() / 255.0)
    assert got.shape == (1, 3, 224, 224) and torch.equal(got, want)
    assert not torch.equal(got, PIPE._clip_preprocess(blur))         # the float blur is a different input
    plain = G._load_structure_image(tmp_path / "a.bmp", torch.device("cpu"), 64)
    assert torch.equal(plain, PIPE._clip_preprocess(disp))
