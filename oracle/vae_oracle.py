"""fp32 CPU oracle of the FLUX AutoencoderKL (TEST INFRASTRUCTURE — see oracle/__init__.py; parity unpinned).

Follows SURVEY.md Appendix A.7 (diffusers AutoencoderKL 0.36.0; the class is only imported by the reference,
PIPE:16, and called at PIPE:467,705,711 (encode) and PIPE:1139 (decode)). Weights: flat dict with diffusers keys.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from .flux_oracle import _s   # identity unless flux_oracle.stored_as(dtype) is active (the HIP path's storage precision)

Params = Dict[str, torch.Tensor]

FLUX_VAE_CFG = dict(in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 512, 512),
                    layers_per_block=2, norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159)


def _gn(p, name, x, groups, eps=1e-6):
    return F.group_norm(x, groups, p[name + ".weight"], p[name + ".bias"], eps)


def _conv(p, name, x, stride=1, padding=1):
    return F.conv2d(x, p[name + ".weight"], p[name + ".bias"], stride=stride, padding=padding)


def resnet(p: Params, pre: str, x, groups):
    h = _s(_conv(p, f"{pre}.conv1", _s(F.silu(_gn(p, f"{pre}.norm1", x, groups)))))
    h = _conv(p, f"{pre}.conv2", _s(F.silu(_gn(p, f"{pre}.norm2", h, groups))))
    if f"{pre}.conv_shortcut.weight" in p:
        x = _s(_conv(p, f"{pre}.conv_shortcut", x, padding=0))
    return _s(x + h)        # the HIP conv adds the skip in its epilogue: one rounding of the sum


def mid_attention(p: Params, pre: str, x, groups):
    """Attention(512, heads=1, GroupNorm, residual_connection=True) of the mid block."""
    B, C, H, W = x.shape
    t = _s(_gn(p, f"{pre}.group_norm", x, groups)).reshape(B, C, H * W).transpose(1, 2)
    q = _s(F.linear(t, p[f"{pre}.to_q.weight"], p[f"{pre}.to_q.bias"]))
    k = _s(F.linear(t, p[f"{pre}.to_k.weight"], p[f"{pre}.to_k.bias"]))
    v = _s(F.linear(t, p[f"{pre}.to_v.weight"], p[f"{pre}.to_v.bias"]))
    a = _s(_s(torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1)) @ v)
    o = F.linear(a, p[f"{pre}.to_out.0.weight"], p[f"{pre}.to_out.0.bias"])
    return _s(x + o.transpose(1, 2).reshape(B, C, H, W))


def _mid(p, pre, h, g):
    h = resnet(p, f"{pre}.resnets.0", h, g)
    h = mid_attention(p, f"{pre}.attentions.0", h, g)
    return resnet(p, f"{pre}.resnets.1", h, g)


def decode(p: Params, cfg: dict, z: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.decode (no post_quant_conv): z [B,16,h,w] (already /scaling + shift) -> [B,3,8h,8w]."""
    g = cfg["norm_num_groups"]
    chans = list(reversed(cfg["block_out_channels"]))
    h = _s(_conv(p, "decoder.conv_in", _s(z)))
    h = _mid(p, "decoder.mid_block", h, g)
    for i, _ in enumerate(chans):
        for j in range(cfg["layers_per_block"] + 1):
            h = resnet(p, f"decoder.up_blocks.{i}.resnets.{j}", h, g)
        if i < len(chans) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _s(_conv(p, f"decoder.up_blocks.{i}.upsamplers.0.conv", h))
    h = _s(F.silu(_gn(p, "decoder.conv_norm_out", h, g)))
    return _conv(p, "decoder.conv_out", h)


def encode_moments(p: Params, cfg: dict, x: torch.Tensor):
    """AutoencoderKL.encode -> (mean, logvar clamped to [-30, 20]); x [B,3,H,W] in [-1,1]."""
    g = cfg["norm_num_groups"]
    chans = list(cfg["block_out_channels"])
    h = _s(_conv(p, "encoder.conv_in", _s(x)))
    for i, _ in enumerate(chans):
        for j in range(cfg["layers_per_block"]):
            h = resnet(p, f"encoder.down_blocks.{i}.resnets.{j}", h, g)
        if i < len(chans) - 1:
            h = F.pad(h, (0, 1, 0, 1))
            h = _s(_conv(p, f"encoder.down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=0))
    h = _mid(p, "encoder.mid_block", h, g)
    h = _s(_conv(p, "encoder.conv_out", _s(F.silu(_gn(p, "encoder.conv_norm_out", h, g)))))
    mean, logvar = h.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def sample_latents(mean, logvar, noise):
    """DiagonalGaussianDistribution.sample: mean + exp(0.5 logvar) * noise."""
    return mean + torch.exp(0.5 * logvar) * noise


def init_vae_params(cfg: dict, seed: int, round_bf16: bool = True, decoder: bool = True, encoder: bool = True) -> Params:
    """Random VAE weights, diffusers key layout. Conv W ~ N(0, (1/sqrt(fan_in))²) keeps activations O(1) through
    ~30 layers so the parity test exercises every layer at a realistic dynamic range."""
    gen = torch.Generator().manual_seed(seed)
    p: Params = {}

    def conv(name, cout, cin, k):
        p[name + ".weight"] = torch.randn(cout, cin, k, k, generator=gen) / math.sqrt(cin * k * k)
        p[name + ".bias"] = 0.05 * torch.randn(cout, generator=gen)

    def lin(name, cout, cin):
        p[name + ".weight"] = torch.randn(cout, cin, generator=gen) / math.sqrt(cin)
        p[name + ".bias"] = 0.05 * torch.randn(cout, generator=gen)

    def norm(name, c):
        p[name + ".weight"] = 1.0 + 0.1 * torch.randn(c, generator=gen)
        p[name + ".bias"] = 0.1 * torch.randn(c, generator=gen)

    def res(pre, cin, cout):
        norm(f"{pre}.norm1", cin); conv(f"{pre}.conv1", cout, cin, 3)
        norm(f"{pre}.norm2", cout); conv(f"{pre}.conv2", cout, cout, 3)
        if cin != cout:
            conv(f"{pre}.conv_shortcut", cout, cin, 1)

    def mid(pre, c):
        res(f"{pre}.resnets.0", c, c)
        norm(f"{pre}.attentions.0.group_norm", c)
        for nm in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(f"{pre}.attentions.0.{nm}", c, c)
        res(f"{pre}.resnets.1", c, c)

    boc = list(cfg["block_out_channels"])
    L = cfg["layers_per_block"]
    if encoder:
        conv("encoder.conv_in", boc[0], cfg["in_channels"], 3)
        cin = boc[0]
        for i, c in enumerate(boc):
            for j in range(L):
                res(f"encoder.down_blocks.{i}.resnets.{j}", cin, c)
                cin = c
            if i < len(boc) - 1:
                conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
        mid("encoder.mid_block", boc[-1])
        norm("encoder.conv_norm_out", boc[-1])
        conv("encoder.conv_out", 2 * cfg["latent_channels"], boc[-1], 3)
    if decoder:
        rev = list(reversed(boc))
        conv("decoder.conv_in", rev[0], cfg["latent_channels"], 3)
        mid("decoder.mid_block", rev[0])
        cin = rev[0]
        for i, c in enumerate(rev):
            for j in range(L + 1):
                res(f"decoder.up_blocks.{i}.resnets.{j}", cin, c)
                cin = c
            if i < len(rev) - 1:
                conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
        norm("decoder.conv_norm_out", rev[-1])
        conv("decoder.conv_out", cfg["out_channels"], rev[-1], 3)
    if round_bf16:
        for k in p:
            p[k] = p[k].to(torch.bfloat16).to(torch.float32)
    return p
