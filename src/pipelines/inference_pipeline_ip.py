"""README.md:89 of the reference uses this older path; same alias."""
from src.pipelines.inference.inference_pipeline_ip import *  # noqa: F401,F403
from src.pipelines.inference.inference_pipeline_ip import main  # noqa: F401

if __name__ == "__main__":
    main()
