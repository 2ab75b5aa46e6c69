"""CPU oracle for the DADD / IP-Adapter DDIM sampler hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(``progressive-stable-diffusion_amd/``) may import this package; only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and there
only as the checker.

Everything here is a plain-PyTorch fp32 restatement written as pure functions over a
flat checkpoint-style ``state_dict`` (key layout: SURVEY.md Appendix D), so the product
and the oracle share nothing but tensors.

Pinning status
--------------
* ``conditioning.py`` (AOE, FeaturePurifier, ImageProjection[Plus]) and
  ``processors.py`` (triple-pathway and baseline cross-attention) restate
  reference-authored arithmetic and are PINNED: ``oracle/make_golden.py`` imported the
  reference modules in the build container and wrote ``tests/golden/*.npz``;
  ``tests/test_oracle_golden.py`` replays them.
* ``sd_unet.py`` / ``sd_vae.py`` restate the third-party ``diffusers``
  ``UNet2DConditionModel`` / ``AutoencoderKL`` (``diffusers>=0.31.0``,
  reference ``pyproject.toml:27``; not vendored, not installed here, no weights or
  config JSON on disk).  The reference holds no test or fixture at that boundary, so
  for those two files **parity is unpinned**: they follow the published SD-1.4
  architecture and are anchored only by the reference's call
  sites (``src/models/unet/unet.py:140-146``, ``src/models/vae/vae.py:71-112``) and
  by reproducing the known parameter counts (859.5 M / 83.7 M).
* ``sampler.py`` restates ``src/pipelines/inference/inference_pipeline_ip.py:232-486``;
  its known-answer constants (timestep grids, alphas_cumprod samples) are pinned by
  SURVEY.md Appendix C values recomputed in ``tests/test_oracle_golden.py``.
"""
