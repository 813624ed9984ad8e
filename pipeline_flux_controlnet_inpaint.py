"""Drop-in module name of the reference (`from pipeline_flux_controlnet_inpaint import FluxControlNetPipeline`,
infer_inpaint.py:4). Same class name as the text-to-image pipeline, different module — as in the reference."""
from reptext_amd.pipeline import FluxPipelineOutput, calculate_shift, retrieve_latents, retrieve_timesteps  # noqa: F401
from reptext_amd.pipeline_inpaint import FluxControlNetPipeline  # noqa: F401

__all__ = ["FluxControlNetPipeline", "FluxPipelineOutput", "calculate_shift", "retrieve_latents", "retrieve_timesteps"]
