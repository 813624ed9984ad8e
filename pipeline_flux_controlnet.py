"""Drop-in module name of the reference (`from pipeline_flux_controlnet import FluxControlNetPipeline`, infer.py:3)."""
from reptext_amd.pipeline import (  # noqa: F401
    FluxControlNetPipeline,
    FluxPipelineOutput,
    calculate_shift,
    retrieve_latents,
    retrieve_timesteps,
)

__all__ = ["FluxControlNetPipeline", "FluxPipelineOutput", "calculate_shift", "retrieve_latents", "retrieve_timesteps"]
