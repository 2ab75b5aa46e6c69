"""Alias of progressive_stable_diffusion_amd.diffusion_module_ip under the reference's path."""
from progressive_stable_diffusion_amd.diffusion_module_ip import (  # noqa: F401
    DiffusionIPConfig, DiffusionModuleWithIP)
