"""MI355X-native DDIM sampler for the DADD / IP-Adapter inference path.

Drop-in for ``src.pipelines.inference.inference_pipeline_ip`` +
``src.models.diffusion_module_ip`` of umutdundar99/progressive-stable-diffusion: Python host
code (this package) above a C-ABI library of hand-written gfx950 kernels
(``csrc/`` -> ``libdadd_hip.so``, declared in ``include/dadd_hip.h``).

The directory is named ``progressive-stable-diffusion_amd``; import it as
``progressive_stable_diffusion_amd`` (a two-line alias package at the repo root).
"""
__version__ = "0.1.0"
