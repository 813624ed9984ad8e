"""Weights for the fp32 CPU oracle, fetched on demand (TEST INFRASTRUCTURE — see oracle/__init__.py).

The full-depth FLUX.1-dev + RepText stack (19+38 / 6+0 blocks, d = 3072) is 14.0 B parameters: 56 GB as fp32, more than the
hosts this runs on can spare. The oracle only ever indexes its parameter mapping (`p[name]`, `p.get(name)`), one tensor at a
time, so it can be handed a mapping that copies each tensor from where the weights already live — the state dict of the model
under test, bf16 in HBM — to a transient fp32 CPU tensor at the moment of use. Host memory then holds one weight matrix at a
time (151 MB for the 12288x3072 feed-forward) and both sides compute on identical, bf16-representable values.

Used by tests/test_configs_gpu.py (full-depth config-1 parity) and bench.py's cpu_baseline leg. Never by the product.
"""
from __future__ import annotations

from typing import Dict, Iterator, Mapping, Optional

import torch


class StreamedParams(Mapping):
    """Read-only mapping name -> fp32 CPU tensor over a state dict that lives elsewhere (any device, any float dtype).

    Nothing is cached: every lookup is a fresh device-to-host copy and upcast, freed when the caller drops it."""

    def __init__(self, state_dict: Dict[str, torch.Tensor]):
        self._sd = state_dict
        self.bytes_streamed = 0

    def __getitem__(self, name: str) -> torch.Tensor:
        t = self._sd[name]
        self.bytes_streamed += t.numel() * t.element_size()
        return t.detach().to(device="cpu", dtype=torch.float32)

    def get(self, name: str, default: Optional[torch.Tensor] = None):
        return self[name] if name in self._sd else default

    def __contains__(self, name) -> bool:
        return name in self._sd

    def __iter__(self) -> Iterator[str]:
        return iter(self._sd)

    def __len__(self) -> int:
        return len(self._sd)


def config1_case(seed: int = 5, T: int = 512, joint_dim: int = 4096, pooled_dim: int = 768, cond_ch: int = 128):
    """BASELINE.json configs[0] as data: 256x256, 2 steps, one Arabic-glyph-sized masked text line (`infer.py:27-33` at the
    plumbing size). bf16-representable fp32 tensors on the CPU; the same dict feeds oracle.denoise_loop and the GPU pipeline."""
    import numpy as np

    from . import flux_oracle as orc

    H = W = 256
    N = (H // 16) * (W // 16)
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    pe, pooled, hint = r(1, T, joint_dim), r(1, pooled_dim), r(1, N, cond_ch)
    lat0 = orc.pack_latents(r(1, 16, 2 * (H // 16), 2 * (W // 16)))
    mask_np = np.zeros([H, W], dtype=np.uint8)
    mask_np[60:140, 80:200] = 255
    rm = torch.nn.functional.interpolate(torch.from_numpy(mask_np)[None, None].float() / 255.0, scale_factor=1 / 16,
                                         mode="bilinear").reshape(1, -1, 1)
    sig = orc.flow_sigmas(2, orc.calculate_shift(N, 256, 4096, 0.5, 1.15))
    return dict(H=H, W=W, N=N, T=T, prompt_embeds=pe, pooled=pooled, hint=hint, latents=lat0, mask_u8=mask_np, region_mask=rm,
                sigmas=sig, img_ids=orc.latent_image_ids(2 * (H // 16), 2 * (W // 16)), txt_ids=torch.zeros(T, 3), guidance=3.5)


def config1_oracle(tp, tcfg, cp, ccfg, case, storage_dtype=None):
    """oracle.denoise_loop on a config1_case; `storage_dtype=torch.bfloat16` = the oracle at the GPU's storage precision."""
    from . import flux_oracle as orc

    c = case
    run = lambda: orc.denoise_loop(tp, tcfg, cp, ccfg, c["latents"], c["prompt_embeds"], c["pooled"], [c["hint"]], [c["region_mask"]],
                                   c["sigmas"], c["img_ids"], c["txt_ids"], c["guidance"])
    with torch.no_grad():
        if storage_dtype is None:
            return run()
        with orc.stored_as(storage_dtype):
            return run()


def config1_gpu(pipe, case, device):
    """The same case through FluxControlNetPipeline.__call__ at the parity tap (output_type='latent', PIPE:1132-1133)."""
    from PIL import Image

    c = case
    b16 = lambda t: t.to(device, torch.bfloat16)
    return pipe(prompt_embeds=b16(c["prompt_embeds"]), pooled_prompt_embeds=b16(c["pooled"]), height=c["H"], width=c["W"],
                num_inference_steps=2, guidance_scale=c["guidance"], control_image=[b16(c["hint"])],
                control_mask=[Image.fromarray(c["mask_u8"])], controlnet_conditioning_scale=1.0, controlnet_conditioning_step=30,
                latents=b16(c["latents"]), output_type="latent").images
