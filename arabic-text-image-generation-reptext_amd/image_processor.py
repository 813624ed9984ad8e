"""VaeImageProcessor — the pre/post-processing the reference takes from diffusers (PIPE:14,222; used at PIPE:680,694,
970,1140 and INP:228-234). Host-side (PIL / numpy / torch CPU) for inputs: it runs once per image before the loop.
Output post-processing on the hot path is done by the decoder's image_out kernel (vae.py); this class only wraps
the resulting uint8 array into PIL images.
"""
from __future__ import annotations

from typing import List, Optional, Union

import numpy as np
import torch
from PIL import Image

PipelineImageInput = Union[Image.Image, np.ndarray, torch.Tensor, List[Image.Image], List[np.ndarray], List[torch.Tensor]]


class VaeImageProcessor:
    def __init__(self, vae_scale_factor: int = 8, do_resize: bool = True, do_normalize: bool = True,
                 do_binarize: bool = False, do_convert_grayscale: bool = False, resample: str = "lanczos"):
        self.vae_scale_factor = vae_scale_factor
        self.do_resize, self.do_normalize = do_resize, do_normalize
        self.do_binarize, self.do_convert_grayscale = do_binarize, do_convert_grayscale
        self.resample = {"lanczos": Image.LANCZOS, "bilinear": Image.BILINEAR, "nearest": Image.NEAREST,
                         "bicubic": Image.BICUBIC}[resample]

    # -- inputs -------------------------------------------------------------------------------
    def _pil_to_tensor(self, images: List[Image.Image], height, width) -> torch.Tensor:
        arrs = []
        for im in images:
            if self.do_convert_grayscale:
                im = im.convert("L")
            if self.do_resize and height is not None and (im.height != height or im.width != width):
                im = im.resize((width, height), resample=self.resample)
            a = np.asarray(im).astype(np.float32) / 255.0
            if a.ndim == 2:
                a = a[..., None]
            arrs.append(a)
        x = torch.from_numpy(np.stack(arrs, axis=0)).permute(0, 3, 1, 2).contiguous()
        return x

    def preprocess(self, image: PipelineImageInput, height: Optional[int] = None, width: Optional[int] = None) -> torch.Tensor:
        """-> float32 [B,C,H,W]; PIL / uint8-like inputs are scaled to [0,1] then (do_normalize) to [-1,1]."""
        if isinstance(image, torch.Tensor):
            x = image if image.dim() == 4 else image.unsqueeze(0)
            x = x.to(torch.float32)
            if self.do_resize and height is not None and (x.shape[-2] != height or x.shape[-1] != width):
                x = torch.nn.functional.interpolate(x, size=(height, width))
            if self.do_normalize and float(x.min()) >= 0:
                x = 2.0 * x - 1.0
            return self._binarize(x)
        if isinstance(image, (Image.Image, np.ndarray)):
            image = [image]
        if isinstance(image[0], np.ndarray):
            pil = []
            for a in image:
                if a.dtype != np.uint8:
                    a = (np.clip(a, 0, 1) * 255).round().astype(np.uint8)
                pil.append(Image.fromarray(a.squeeze(-1) if (a.ndim == 3 and a.shape[-1] == 1) else a))
            image = pil
        if isinstance(image[0], torch.Tensor):
            return self.preprocess(torch.stack([t if t.dim() == 3 else t[0] for t in image]), height, width)
        x = self._pil_to_tensor(list(image), height, width)
        if self.do_normalize:
            x = 2.0 * x - 1.0
        return self._binarize(x)

    def _binarize(self, x):
        if self.do_binarize:
            x = (x >= 0.5).to(x.dtype)
        return x

    # -- outputs ------------------------------------------------------------------------------
    @staticmethod
    def numpy_to_pil(u8: np.ndarray) -> List[Image.Image]:
        return [Image.fromarray(a.squeeze(-1), mode="L") if a.shape[-1] == 1 else Image.fromarray(a) for a in u8]

    def postprocess_u8(self, u8_nhwc: torch.Tensor, output_type: str = "pil"):
        """u8_nhwc: uint8 [B,H,W,C] produced by the decoder's image_out kernel (round(clamp(x/2+0.5,0,1)*255))."""
        arr = u8_nhwc.cpu().numpy()
        if output_type == "pil":
            return self.numpy_to_pil(arr)
        if output_type == "np":
            return arr.astype(np.float32) / 255.0
        raise ValueError(f"unsupported output_type {output_type}")
