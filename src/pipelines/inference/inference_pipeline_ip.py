"""Alias of progressive_stable_diffusion_amd.inference_pipeline_ip under the reference's path."""
from progressive_stable_diffusion_amd.inference_pipeline_ip import *  # noqa: F401,F403
from progressive_stable_diffusion_amd.inference_pipeline_ip import (  # noqa: F401
    _apply_leace, _build_labels, _create_progression_grid, _ddim_sample_ip, _latents_to_images,
    _load_and_preprocess_structure_image, _load_config, _load_leace_projection, _parse_args,
    _prepare_conditioning, _resolve_device, _save_sequence, _set_delta_scale_on_processors, _set_seed, main)

if __name__ == "__main__":
    main()
