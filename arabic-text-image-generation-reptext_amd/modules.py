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


def hf_cache_dirs():
    """The hub cache root, by huggingface_hub's precedence: HF_HUB_CACHE, HUGGINGFACE_HUB_CACHE, $HF_HOME/hub,
    ~/.cache/huggingface/hub — the first one that is set is THE cache (no network is ever touched)."""
    for env in ("HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE"):
        if os.environ.get(env):
            return [os.environ[env]]
    if os.environ.get("HF_HOME"):
        return [os.path.join(os.environ["HF_HOME"], "hub")]
    return [os.path.join(os.path.expanduser("~"), ".cache", "huggingface", "hub")]


def resolve_model_path(name_or_path: str, revision: Optional[str] = None) -> str:
    """A local directory as it is; a hub id ("ORG/NAME", as infer.py:30-31 passes them) through the LOCAL hub cache layout
    ``<cache>/models--ORG--NAME/snapshots/<commit>/`` (``refs/<revision or main>`` names the commit; else the newest snapshot).
    Nothing is downloaded: a hub id without a cached snapshot raises with the paths that were tried."""
    if os.path.isdir(name_or_path):
        return name_or_path
    tried = []
    if "/" in name_or_path and not name_or_path.startswith((".", "/")) and name_or_path.count("/") == 1:
        folder = "models--" + name_or_path.replace("/", "--")
        for root in hf_cache_dirs():
            repo = os.path.join(root, folder)
            tried.append(repo)
            snaps = os.path.join(repo, "snapshots")
            if not os.path.isdir(snaps):
                continue
            ref = os.path.join(repo, "refs", revision or "main")
            if os.path.isfile(ref):
                with open(ref) as f:
                    cand = os.path.join(snaps, f.read().strip())
                if os.path.isdir(cand):
                    return cand
            if revision and os.path.isdir(os.path.join(snaps, revision)):
                return os.path.join(snaps, revision)
            cands = [os.path.join(snaps, d) for d in os.listdir(snaps) if os.path.isdir(os.path.join(snaps, d))]
            if cands:
                return max(cands, key=os.path.getmtime)
    raise OSError(
        f"'{name_or_path}' is neither a local directory nor a hub id with a snapshot in the local hub cache "
        f"(looked for {tried or 'no cache path: not an ORG/NAME id'}). There is no network access on this path: download the "
        "snapshot elsewhere and point HF_HOME / HF_HUB_CACHE at it, or pass the directory.")


class WeightsIO:
    """Checkpoint IO shared by the models: config.json + *.safetensors (single file or sharded with an index).

    Mirrors the subset of diffusers' ModelMixin.from_pretrained the reference uses (infer.py:30-33): local directories and
    hub ids that resolve through the local hub cache (resolve_model_path); nothing is ever downloaded."""

    config_name = "config.json"
    weights_name = "diffusion_pytorch_model.safetensors"

    @classmethod
    def _resolve_dir(cls, path: str, subfolder: Optional[str] = None) -> str:
        root = resolve_model_path(path)
        p = os.path.join(root, subfolder) if subfolder else root
        if not os.path.isdir(p):
            raise OSError(f"{cls.__name__}.from_pretrained: '{p}' does not exist (resolved from '{path}')")
        return p

    @classmethod
    def _load_safetensors_dir(cls, d: str) -> dict:
        """State dict of a model directory. With ``<weights_name>.index.json`` (diffusers' sharded layout, NB:2116-2118:
        three transformer shards) exactly the shards its weight_map names are read and every key must come from the shard
        the index assigns it to; otherwise every *.safetensors file of the directory."""
        from safetensors.torch import load_file

        idx = os.path.join(d, cls.weights_name + ".index.json")
        sd = {}
        if os.path.isfile(idx):
            with open(idx) as f:
                wm = json.load(f)["weight_map"]
            for fn in sorted(set(wm.values())):
                fp = os.path.join(d, fn)
                if not os.path.isfile(fp):
                    raise OSError(f"{idx} names shard '{fn}', which is missing from {d}")
                part = load_file(fp)
                for k, v in part.items():
                    if wm.get(k) == fn:
                        sd[k] = v
            missing = [k for k in wm if k not in sd]
            if missing:
                raise OSError(f"{len(missing)} tensors of {idx} were not found in their shards, e.g. {missing[:3]}")
            return sd
        files = sorted(f for f in os.listdir(d) if f.endswith(".safetensors"))
        if not files:
            raise OSError(f"no .safetensors files in {d}")
        for fn in files:
            sd.update(load_file(os.path.join(d, fn)))
        return sd

    def save_pretrained(self, d: str, max_shard_bytes: int = 10 << 30) -> None:
        """config.json + weights in diffusers' layout: one file, or ``…-0000k-of-0000n.safetensors`` shards plus the index when
        the state dict exceeds ``max_shard_bytes`` (diffusers' default is 10 GB; FLUX.1-dev's transformer ships as 3 shards)."""
        from safetensors.torch import save_file

        os.makedirs(d, exist_ok=True)
        self.config.save_json(os.path.join(d, self.config_name), type(self).__name__)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        total = sum(v.numel() * v.element_size() for v in sd.values())
        if total <= max_shard_bytes:
            save_file(sd, os.path.join(d, self.weights_name))
            return
        shards, cur, cur_b = [], {}, 0
        for k, v in sd.items():
            nb = v.numel() * v.element_size()
            if cur and cur_b + nb > max_shard_bytes:
                shards.append(cur)
                cur, cur_b = {}, 0
            cur[k] = v
            cur_b += nb
        shards.append(cur)
        stem = self.weights_name[: -len(".safetensors")]
        weight_map = {}
        for i, part in enumerate(shards):
            fn = f"{stem}-{i + 1:05d}-of-{len(shards):05d}.safetensors"
            save_file(part, os.path.join(d, fn))
            weight_map.update({k: fn for k in part})
        with open(os.path.join(d, self.weights_name + ".index.json"), "w") as f:
            json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=2)
