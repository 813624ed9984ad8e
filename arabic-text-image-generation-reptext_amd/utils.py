"""Small host utilities shared by the pipelines."""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import torch


def randn_tensor(shape: Sequence[int], generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 device=None, dtype=None) -> torch.Tensor:
    """Gaussian noise with diffusers' device rule (PIPE:26,599,643): draw on the generator's device (a CPU generator gives
    reproducible noise for any target device), then move. A list of generators draws one sample per batch entry."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    shape = tuple(shape)
    if isinstance(generator, (list, tuple)):
        if len(generator) != shape[0]:
            raise ValueError(f"got {len(generator)} generators for batch size {shape[0]}")
        parts = [randn_tensor((1,) + shape[1:], g, device, dtype) for g in generator]
        return torch.cat(parts, dim=0)
    gdev = generator.device if generator is not None else device
    out = torch.randn(shape, generator=generator, device=gdev, dtype=dtype)
    return out.to(device)
