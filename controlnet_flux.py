"""Drop-in module name of the reference (`from controlnet_flux import FluxControlNetModel`, infer.py:2, PIPE:36).
The implementation lives in the MI355X package; this file only re-exports the public names."""
from reptext_amd.controlnet import FluxControlNetModel, FluxControlNetOutput, FluxMultiControlNetModel  # noqa: F401

__all__ = ["FluxControlNetModel", "FluxControlNetOutput", "FluxMultiControlNetModel"]
