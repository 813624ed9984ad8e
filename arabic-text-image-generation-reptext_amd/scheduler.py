"""FlowMatchEulerDiscreteScheduler with dynamic (exponential) time shifting — the scheduler the reference takes from
diffusers (PIPE:18) and drives at PIPE:948-967 (``set_timesteps(sigmas=..., mu=...)``) and PIPE:1109 (``step``).

Math per SURVEY.md Appendix A.6. The sigma schedule is host-side fp32 (n+1 scalars); ``step`` is one HIP axpy
(fp32 arithmetic, result rounded once to the sample dtype, as diffusers upcasts inside ``step``).
"""
from __future__ import annotations

import math
from typing import List, Optional, Union

import numpy as np
import torch

from . import ops
from .config import Config, flux_scheduler_config


def calculate_shift(image_seq_len, base_seq_len: int = 256, max_seq_len: int = 4096, base_shift: float = 0.5,
                    max_shift: float = 1.16):
    """Linear map seq_len -> mu (PIPE:78-88; note the 1.16 default that the pipeline overrides with config 1.15)."""
    slope = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    return image_seq_len * slope + (base_shift - slope * base_seq_len)


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, **config):
        self.config = flux_scheduler_config(**config)
        self.timesteps: Optional[torch.Tensor] = None
        self.sigmas: Optional[torch.Tensor] = None      # host fp32, n+1 entries (trailing 0)
        self._step_index: Optional[int] = None
        self._begin_index: Optional[int] = None
        self.num_inference_steps: Optional[int] = None

    @classmethod
    def from_config(cls, config):
        return cls(**{k: v for k, v in dict(config).items() if not k.startswith("_")})

    @property
    def step_index(self):
        return self._step_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    def _time_shift(self, mu: float, s: np.ndarray) -> np.ndarray:
        return math.exp(mu) / (math.exp(mu) + (1.0 / s - 1.0))

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None, sigmas: Optional[List[float]] = None,
                      mu: Optional[float] = None, timesteps: Optional[List[float]] = None):
        cfg = self.config
        if cfg.use_dynamic_shifting and mu is None:
            raise ValueError("`mu` must be passed when `use_dynamic_shifting` is set to be `True`")
        if sigmas is None:
            if timesteps is not None:
                s = np.asarray(timesteps, dtype=np.float32) / cfg.num_train_timesteps
            else:
                if num_inference_steps is None:
                    raise ValueError("pass num_inference_steps, sigmas or timesteps")
                # diffusers takes the grid's end points from the sigma table its constructor built — which, without dynamic
                # shifting, has ALREADY been passed through the static shift (sigma_max stays 1, sigma_min becomes
                # shift·s/(1+(shift-1)·s) at s = 1/num_train_timesteps) — and then applies the shift to the grid once more below.
                # Restated from knowledge of diffusers 0.36 (parity unpinned: the package is absent); FLUX.1's scheduler config
                # sets use_dynamic_shifting=True and never takes this branch.
                smax, smin = 1.0, 1.0 / cfg.num_train_timesteps
                if not cfg.use_dynamic_shifting:
                    smin = cfg.shift * smin / (1 + (cfg.shift - 1) * smin)
                t = np.linspace(smax * cfg.num_train_timesteps, smin * cfg.num_train_timesteps, num_inference_steps)
                s = (t / cfg.num_train_timesteps).astype(np.float32)
        else:
            s = np.asarray(sigmas, dtype=np.float32)
        if cfg.use_dynamic_shifting:
            s = self._time_shift(float(mu), s.astype(np.float32)).astype(np.float32)
        else:
            s = (cfg.shift * s / (1 + (cfg.shift - 1) * s)).astype(np.float32)
        self.num_inference_steps = len(s)
        sig = torch.from_numpy(np.ascontiguousarray(s)).to(torch.float32)
        self.timesteps = (sig * cfg.num_train_timesteps).to(device=device)
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        self._step_index = None
        self._begin_index = None

    def _init_step_index(self, timestep):
        if self._begin_index is not None:
            self._step_index = self._begin_index
            return
        t = float(timestep)
        idx = (self.timesteps.cpu() == t).nonzero()
        self._step_index = int(idx[1 if len(idx) > 1 else 0]) if len(idx) > 0 else 0

    def step_master_(self, model_output: torch.Tensor, sample32: torch.Tensor, sample_bf16: Optional[torch.Tensor] = None) -> None:
        """In-place Euler step on an fp32 master copy of the latents (used by this repo's pipelines between steps; ``step`` keeps
        the diffusers contract of returning the model dtype). Advances the step index exactly like ``step``."""
        if self._step_index is None:
            self._step_index = self._begin_index if self._begin_index is not None else 0
        i = self._step_index
        ops.euler_step_f32_(sample32, model_output.contiguous(), float(self.sigmas[i + 1] - self.sigmas[i]), sample_bf16)
        self._step_index = i + 1

    def step(self, model_output: torch.Tensor, timestep: Union[float, torch.Tensor], sample: torch.Tensor,
             return_dict: bool = True, **unused):
        if self._step_index is None:
            # the loop walks timesteps in order; avoid a device sync by starting at 0 unless a begin index was set
            self._step_index = self._begin_index if self._begin_index is not None else 0
        i = self._step_index
        dsigma = float(self.sigmas[i + 1] - self.sigmas[i])
        prev = sample.to(model_output.dtype).clone()
        ops.euler_step_(prev, model_output.contiguous(), dsigma)
        self._step_index = i + 1
        if not return_dict:
            return (prev,)
        return Config(prev_sample=prev)
