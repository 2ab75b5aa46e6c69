"""Import-path aliases: ``src.pipelines.inference.inference_pipeline_ip`` and
``src.models.diffusion_module_ip`` (the reference's module paths) resolve to the MI355X engine."""
