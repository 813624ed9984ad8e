"""MI355X-native FLUX.1-dev + RepText-ControlNet denoising path (import as ``reptext_amd``).

Layout: csrc/ (HIP kernels + C ABI, built into librt_reptext_hip.so), native.py (ctypes binding),
ops.py (tensor-level calls), and the host-side mirror of the reference's Python interface
(controlnet.py, transformer.py, vae.py, scheduler.py, pipeline*.py).
"""
__version__ = "0.1.0"
