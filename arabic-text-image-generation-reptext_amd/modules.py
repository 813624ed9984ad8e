"""Parameter containers with diffusers state-dict key names.

These nn.Modules only HOLD weights (so ``state_dict()``/``load_state_dict()``/``.to()`` behave like the reference's
ModelMixin models); none of them has a torch ``forward``. Compute happens in mmdit.py through the HIP ops.
"""
from __future__ import annotations

import json
import os
from typing import Optional

import torch
import torch.nn as nn


class Lin(nn.Module):
    """Weights of an nn.Linear ([out,in] + bias), allocated uninitialised (no CPU init pass over 12 B params)."""

    def __init__(self, in_f: int, out_f: int, device=None, dtype=None):
        super().__init__()
        self.in_features, self.out_features = in_f, out_f
        self.weight = nn.Parameter(torch.empty(out_f, in_f, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_f, device=device, dtype=dtype), requires_grad=False)


class NormW(nn.Module):
    """RMSNorm weight [Dh]."""

    def __init__(self, dim: int, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(dim, device=device, dtype=dtype), requires_grad=False)


class _Holder(nn.Module):
    pass


def _ada(d: int, mult: int, **kw) -> nn.Module:
    m = _Holder()
    m.linear = Lin(d, mult * d, **kw)
    return m


def _ff(d: int, **kw) -> nn.Module:
    """ff.net.0.proj / ff.net.2 key layout of diffusers FeedForward."""
    m = _Holder()
    act = _Holder()
    act.proj = Lin(d, 4 * d, **kw)
    m.net = nn.ModuleList([act, _Holder(), Lin(4 * d, d, **kw)])
    return m


class DoubleBlockParams(nn.Module):
    """FluxTransformerBlock parameters (Appendix A.1)."""

    def __init__(self, d: int, Dh: int, **kw):
        super().__init__()
        self.norm1 = _ada(d, 6, **kw)
        self.norm1_context = _ada(d, 6, **kw)
        a = _Holder()
        for nm in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_add_out"):
            setattr(a, nm, Lin(d, d, **kw))
        a.to_out = nn.ModuleList([Lin(d, d, **kw)])
        for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            setattr(a, nm, NormW(Dh, **kw))
        self.attn = a
        self.ff = _ff(d, **kw)
        self.ff_context = _ff(d, **kw)


class SingleBlockParams(nn.Module):
    """FluxSingleTransformerBlock parameters (Appendix A.2)."""

    def __init__(self, d: int, Dh: int, **kw):
        super().__init__()
        self.norm = _ada(d, 3, **kw)
        self.proj_mlp = Lin(d, 4 * d, **kw)
        self.proj_out = Lin(5 * d, d, **kw)
        a = _Holder()
        for nm in ("to_q", "to_k", "to_v"):
            setattr(a, nm, Lin(d, d, **kw))
        a.norm_q = NormW(Dh, **kw)
        a.norm_k = NormW(Dh, **kw)
        self.attn = a


class TimeTextEmbedParams(nn.Module):
    """CombinedTimestep(Guidance)TextProjEmbeddings parameters (Appendix A.5)."""

    def __init__(self, d: int, pooled_dim: int, guidance: bool, **kw):
        super().__init__()

        def mlp(in_f):
            m = _Holder()
            m.linear_1 = Lin(in_f, d, **kw)
            m.linear_2 = Lin(d, d, **kw)
            return m

        self.timestep_embedder = mlp(256)
        if guidance:
            self.guidance_embedder = mlp(256)
        self.text_embedder = mlp(pooled_dim)


class WeightsIO:
    """Local-path checkpoint IO shared by the models: config.json + *.safetensors (sharded or not).

    Mirrors the subset of diffusers' ModelMixin.from_pretrained the reference uses (infer.py:30-33) for LOCAL
    directories only — hub ids cannot resolve offline and raise."""

    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.safetensors"

    @classmethod
    def _resolve_dir(cls, path: str, subfolder: Optional[str] = None) -> str:
        p = os.path.join(path, subfolder) if subfolder else path
        if not os.path.isdir(p):
            raise OSError(
                f"{cls.__name__}.from_pretrained: '{path}' is not a local directory. Hub ids (e.g. 'Shakker-Labs/RepText') "
                "need network access; download the snapshot and pass its path."
            )
        return p

    @staticmethod
    def _load_safetensors_dir(d: str) -> dict:
        from safetensors.torch import load_file

        idx = os.path.join(d, "diffusion_pytorch_model.safetensors.index.json")
        files = []
        if os.path.isfile(idx):
            with open(idx) as f:
                files = sorted(set(json.load(f)["weight_map"].values()))
        else:
            files = sorted(f for f in os.listdir(d) if f.endswith(".safetensors"))
        if not files:
            raise OSError(f"no .safetensors files in {d}")
        sd = {}
        for fn in files:
            sd.update(load_file(os.path.join(d, fn)))
        return sd

    def save_pretrained(self, d: str, max_shard_bytes: int = 10 << 30) -> None:
        from safetensors.torch import save_file

        os.makedirs(d, exist_ok=True)
        self.config.save_json(os.path.join(d, self.config_name), type(self).__name__)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        save_file(sd, os.path.join(d, self.weights_name))
