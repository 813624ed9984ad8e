"""AutoencoderKL (FLUX VAE) on the HIP kernels of csrc/vae.hip — the class the reference imports from diffusers
(PIPE:16) and calls as ``vae.encode(x).latent_dist.sample()`` (PIPE:467,705,711) and ``vae.decode(z)`` (PIPE:1139).

Math per SURVEY.md Appendix A.7. State-dict keys follow diffusers (OIHW conv weights); at plan time the conv weights
are repacked to [Cout][ky][kx][Cin] with channels zero-padded to the kernel's granularity.
All activations live in zero-haloed NHWC bf16 buffers from a small pool (halo written once at allocation).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import native, ops
from .config import Config, flux_vae_config
from .modules import WeightsIO

BF16, F32 = torch.bfloat16, torch.float32


def _stream():
    return torch.cuda.current_stream().cuda_stream


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(cout, device=device, dtype=dtype), requires_grad=False)


class _Affine(nn.Module):
    def __init__(self, c, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(c, device=device, dtype=dtype), requires_grad=False)


class _Lin(nn.Module):
    def __init__(self, cin, cout, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(cout, device=device, dtype=dtype), requires_grad=False)


class _H(nn.Module):
    pass


def _resnet(cin, cout, **kw):
    m = _H()
    m.norm1 = _Affine(cin, **kw)
    m.conv1 = _Conv(cin, cout, 3, **kw)
    m.norm2 = _Affine(cout, **kw)
    m.conv2 = _Conv(cout, cout, 3, **kw)
    if cin != cout:
        m.conv_shortcut = _Conv(cin, cout, 1, **kw)
    return m


def _mid(c, **kw):
    m = _H()
    m.resnets = nn.ModuleList([_resnet(c, c, **kw), _resnet(c, c, **kw)])
    a = _H()
    a.group_norm = _Affine(c, **kw)
    a.to_q, a.to_k, a.to_v = _Lin(c, c, **kw), _Lin(c, c, **kw), _Lin(c, c, **kw)
    a.to_out = nn.ModuleList([_Lin(c, c, **kw)])
    m.attentions = nn.ModuleList([a])
    return m


class _Pool:
    """Zero-haloed NHWC buffers keyed by shape; kernels only ever write interiors, so halos stay zero."""

    def __init__(self, device):
        self.device = device
        self.free: Dict[tuple, List[torch.Tensor]] = {}

    def get(self, B, H, W, C, dtype=BF16) -> torch.Tensor:
        key = (B, H, W, C, dtype)
        lst = self.free.setdefault(key, [])
        if lst:
            return lst.pop()
        return torch.zeros(B, H + 2, W + 2, C, device=self.device, dtype=dtype)

    def put(self, t: torch.Tensor):
        B, Hp, Wp, C = t.shape
        self.free.setdefault((B, Hp - 2, Wp - 2, C, t.dtype), []).append(t)


def _pad64(c: int) -> int:
    return (c + 63) // 64 * 64


@dataclass
class DecoderOutput:
    sample: torch.Tensor


class DiagonalGaussianDistribution:
    """mean/logvar holder with ``sample(generator)`` / ``mode()`` like diffusers' (used at PIPE:95,705,711)."""

    def __init__(self, mean: torch.Tensor, logvar: torch.Tensor):
        self.mean, self.logvar = mean, logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        from .utils import randn_tensor

        noise = randn_tensor(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


@dataclass
class AutoencoderKLOutput:
    latent_dist: DiagonalGaussianDistribution


class AutoencoderKL(nn.Module, WeightsIO):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=16, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, norm_num_groups=32, act_fn="silu", scaling_factor=0.3611, shift_factor=0.1159,
                 use_quant_conv=False, use_post_quant_conv=False, mid_block_add_attention=True, force_upcast=True,
                 device=None, dtype=None, **unused):
        super().__init__()
        if use_quant_conv or use_post_quant_conv or not mid_block_add_attention:
            raise NotImplementedError("only the FLUX VAE topology (no quant convs, mid attention) is implemented")
        self.config = flux_vae_config(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                                      block_out_channels=list(block_out_channels), layers_per_block=layers_per_block,
                                      norm_num_groups=norm_num_groups, scaling_factor=scaling_factor, shift_factor=shift_factor)
        kw = dict(device=device, dtype=dtype)
        boc, L = list(block_out_channels), layers_per_block
        enc = _H()
        enc.conv_in = _Conv(in_channels, boc[0], 3, **kw)
        enc.down_blocks = nn.ModuleList()
        cin = boc[0]
        for i, c in enumerate(boc):
            blk = _H()
            blk.resnets = nn.ModuleList([_resnet(cin if j == 0 else c, c, **kw) for j in range(L)])
            cin = c
            if i < len(boc) - 1:
                ds = _H()
                ds.conv = _Conv(c, c, 3, **kw)
                blk.downsamplers = nn.ModuleList([ds])
            enc.down_blocks.append(blk)
        enc.mid_block = _mid(boc[-1], **kw)
        enc.conv_norm_out = _Affine(boc[-1], **kw)
        enc.conv_out = _Conv(boc[-1], 2 * latent_channels, 3, **kw)
        self.encoder = enc
        dec = _H()
        rev = list(reversed(boc))
        dec.conv_in = _Conv(latent_channels, rev[0], 3, **kw)
        dec.mid_block = _mid(rev[0], **kw)
        dec.up_blocks = nn.ModuleList()
        cin = rev[0]
        for i, c in enumerate(rev):
            blk = _H()
            blk.resnets = nn.ModuleList([_resnet(cin if j == 0 else c, c, **kw) for j in range(L + 1)])
            cin = c
            if i < len(rev) - 1:
                us = _H()
                us.conv = _Conv(c, c, 3, **kw)
                blk.upsamplers = nn.ModuleList([us])
            dec.up_blocks.append(blk)
        dec.conv_norm_out = _Affine(rev[-1], **kw)
        dec.conv_out = _Conv(rev[-1], out_channels, 3, **kw)
        self.decoder = dec
        self._packed: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._pool: Optional[_Pool] = None
        self._stats: Optional[torch.Tensor] = None

    # ------------------------------------------------------------------ plumbing
    @property
    def dtype(self):
        return self.decoder.conv_in.weight.dtype

    @property
    def device(self):
        return self.decoder.conv_in.weight.device

    def _invalidate_derived(self):
        """Drop every tensor derived from parameters (repacked conv weights, fused mid-attention q|k|v): they are rebuilt on
        next use. Called whenever parameters are rewritten IN PLACE — load_state_dict and random_init_ copy into the existing
        storage, so data_ptr() does not change and cannot serve as the cache key on its own."""
        self._packed = {}
        for m in self.modules():
            if hasattr(m, "_qkv"):
                del m._qkv

    def _apply(self, fn, *a, **k):
        self._invalidate_derived()
        self._pool, self._stats = None, None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, sd, strict: bool = True, **kw):
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._invalidate_derived()
        return out

    def random_init_(self, seed: int = 0):
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, p in self.named_parameters():
            r = torch.randn(p.shape, generator=g, device=p.device, dtype=F32)
            if p.dim() == 4:
                p.data.copy_(r / math.sqrt(p.shape[1] * p.shape[2] * p.shape[3]))
            elif p.dim() == 2:
                p.data.copy_(r / math.sqrt(p.shape[1]))
            elif name.endswith("weight"):
                p.data.copy_(1.0 + 0.1 * r)
            else:
                p.data.copy_(0.05 * r)
        self._invalidate_derived()
        return self

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, subfolder: Optional[str] = None, device=None, **unused):
        d = cls._resolve_dir(path, subfolder)
        cfg = Config.from_json_file(d + "/" + cls.config_name)
        m = cls(**cfg, device=device or "cpu", dtype=torch_dtype or BF16)
        m.load_state_dict({k: v.to(torch_dtype or BF16) for k, v in cls._load_safetensors_dir(d).items()}, strict=True)
        return m

    def _ready(self):
        if self.dtype != BF16:
            raise TypeError(f"AutoencoderKL: HIP path stores bf16; got {self.dtype}")
        if not self.decoder.conv_in.weight.is_cuda:
            raise RuntimeError(f"AutoencoderKL is on {self.device}; move it to the GPU. There is no CPU fallback.")
        if self._pool is None:
            self._pool = _Pool(self.device)
            self._stats = None

    def _w(self, conv: _Conv) -> Tuple[torch.Tensor, torch.Tensor]:
        """[Cout,Cin,k,k] -> bf16 [Cout_pad4][k][k][Cin_pad64] (zero padded), cached per parameter."""
        key = id(conv)
        hit = self._packed.get(key)
        if hit is not None and hit[2] == conv.weight.data_ptr():
            return hit[0], hit[1]
        cout, cin, k, _ = conv.weight.shape
        cp, op = _pad64(cin), (cout + 3) // 4 * 4
        w = torch.zeros(op, k, k, cp, device=self.device, dtype=BF16)
        w[:cout, :, :, :cin] = conv.weight.data.permute(0, 2, 3, 1)
        b = torch.zeros(op, device=self.device, dtype=BF16)
        b[:cout] = conv.bias.data
        self._packed[key] = (w.contiguous(), b, conv.weight.data_ptr())
        return w, b

    # ------------------------------------------------------------------ kernels
    def _conv(self, conv: _Conv, x: torch.Tensor, *, stride=1, up=False, res: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None, out_f32=False) -> torch.Tensor:
        w, b = self._w(conv)
        B, Hp, Wp, Cin = x.shape
        Hs, Ws = Hp - 2, Wp - 2
        if Cin != w.shape[3]:
            raise ValueError(f"conv input has {Cin} channels, packed weight expects {w.shape[3]}")
        Ho = Hs // 2 if stride == 2 else (2 * Hs if up else Hs)
        Wo = Ws // 2 if stride == 2 else (2 * Ws if up else Ws)
        cout = w.shape[0]
        if out is None:
            out = self._pool.get(B, Ho, Wo, cout, F32 if out_f32 else BF16)
        native.check("rt_conv2d_nhwc", native.load().rt_conv2d_nhwc(
            x.data_ptr(), w.data_ptr(), b.data_ptr(), None if res is None else res.data_ptr(), out.data_ptr(), B, Hs, Ws, Cin,
            cout, w.shape[1], stride, int(up), int(out_f32), _stream()))
        return out

    def _gn(self, norm: _Affine, x: torch.Tensor, silu: bool, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, Hp, Wp, C = x.shape
        if out is None:
            out = self._pool.get(B, Hp - 2, Wp - 2, C)
        lib = native.load()
        need = int(lib.rt_groupnorm_ws_bytes(B, Hp - 2, Wp - 2, self.config.norm_num_groups))
        if self._stats is None or self._stats.numel() * 8 < need:
            self._stats = torch.empty((need + 7) // 8, device=self.device, dtype=torch.float64)
        native.check("rt_groupnorm_silu_nhwc", lib.rt_groupnorm_silu_nhwc(
            x.data_ptr(), out.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), self._stats.data_ptr(), B, Hp - 2, Wp - 2, C,
            self.config.norm_num_groups, 1e-6, int(silu), _stream()))
        return out

    def _resnet(self, m, x: torch.Tensor) -> torch.Tensor:
        """x + conv2(silu(gn2(conv1(silu(gn1(x)))))) with optional 1x1 shortcut; consumes x (returned to the pool)."""
        t = self._gn(m.norm1, x, True)
        h = self._conv(m.conv1, t)
        self._pool.put(t)
        t = self._gn(m.norm2, h, True)
        if hasattr(m, "conv_shortcut"):
            sc = self._conv(m.conv_shortcut, x)
            self._pool.put(x)
            out = self._conv(m.conv2, t, res=sc, out=sc)          # in place over the shortcut
        else:
            out = self._conv(m.conv2, t, res=x, out=x)            # in place over x (pixel-wise dependence only)
        self._pool.put(t)
        self._pool.put(h)
        return out

    def _mid_attn(self, a, x: torch.Tensor) -> torch.Tensor:
        B, Hp, Wp, C = x.shape
        H, W = Hp - 2, Wp - 2
        HW = H * W
        if HW % 32 or C not in (128, 256, 512):
            raise ValueError("mid-block attention needs H*W % 32 == 0 and 128, 256 or 512 channels")
        t = self._gn(a.group_norm, x, False)
        tc = t[:, 1:-1, 1:-1, :].reshape(B, HW, C)               # compact copy of the interior (torch view+copy = plumbing)
        self._pool.put(t)
        if not hasattr(a, "_qkv"):
            a._qkv = (torch.cat([a.to_q.weight.data, a.to_k.weight.data, a.to_v.weight.data]).contiguous(),
                      torch.cat([a.to_q.bias.data, a.to_k.bias.data, a.to_v.bias.data]).contiguous())
        qkv = torch.empty(B, HW, 3 * C, device=x.device, dtype=BF16)
        ops.linear(tc, a._qkv[0], qkv, bias=a._qkv[1])
        # one head of C channels over HW positions, flash-style: no HW x HW scores (csrc/vae_attention.hip)
        o = torch.empty(B, HW, C, device=x.device, dtype=BF16)
        q, k, v = qkv[..., :C], qkv[..., C : 2 * C], qkv[..., 2 * C :]
        native.check("rt_vae_attention", native.load().rt_vae_attention(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), qkv.stride(1), qkv.stride(0), o.stride(1), o.stride(0), B, HW, C,
            1.0 / math.sqrt(C), _stream()))
        for b in range(B):
            # to_out + residual, written row by row into the haloed buffer: batch = image rows
            xin = x[b, 1:-1, 1:-1, :]                              # [H, W, C] strided view of the interior
            ops.linear(o[b].view(H, W, C), a.to_out[0].weight.data, xin, bias=a.to_out[0].bias.data, res=xin)
        return x

    def _mid_block(self, m, h):
        h = self._resnet(m.resnets[0], h)
        h = self._mid_attn(m.attentions[0], h)
        return self._resnet(m.resnets[1], h)

    # ------------------------------------------------------------------ public API
    @torch.no_grad()
    def _decode_haloed(self, z: torch.Tensor) -> torch.Tensor:
        """z: haloed NHWC bf16 [B,h+2,w+2,64] (16 real channels) -> haloed NHWC f32 [B,8h+2,8w+2,4] (3 real)."""
        d = self.decoder
        h = self._conv(d.conv_in, z)
        h = self._mid_block(d.mid_block, h)
        for blk in d.up_blocks:
            for r in blk.resnets:
                h = self._resnet(r, h)
            if hasattr(blk, "upsamplers"):
                up = self._conv(blk.upsamplers[0].conv, h, up=True)
                self._pool.put(h)
                h = up
        t = self._gn(d.conv_norm_out, h, True)
        self._pool.put(h)
        img = self._conv(d.conv_out, t, out_f32=True)
        self._pool.put(t)
        return img

    def _image_out(self, img: torch.Tensor, want_u8: bool):
        B, Hp, Wp, Cp = img.shape
        H, W, C = Hp - 2, Wp - 2, self.config.out_channels
        if want_u8:
            out = torch.empty(B, H, W, C, device=img.device, dtype=torch.uint8)
            native.check("rt_image_out", native.load().rt_image_out(img.data_ptr(), None, out.data_ptr(), B, H, W, Cp, C, _stream()))
        else:
            out = torch.empty(B, C, H, W, device=img.device, dtype=F32)
            native.check("rt_image_out", native.load().rt_image_out(img.data_ptr(), out.data_ptr(), None, B, H, W, Cp, C, _stream()))
        self._pool.put(img)
        return out

    @torch.no_grad()
    def decode_packed(self, packed: torch.Tensor, h2: int, w2: int, output_u8: bool = False) -> torch.Tensor:
        """Fast path used by the pipelines: packed latents [B,(h2/2)(w2/2),64] -> image. Fuses _unpack_latents and
        ``z / scaling_factor + shift_factor`` (PIPE:1136-1137) into the decoder's input write."""
        self._ready()
        B = packed.shape[0]
        C = self.config.latent_channels
        z = self._pool.get(B, h2, w2, _pad64(C))
        packed = packed.contiguous()          # named, so the buffer outlives the raw-pointer call
        native.check("rt_unpack_latents_haloed", native.load().rt_unpack_latents_haloed(
            packed.data_ptr(), z.data_ptr(), B, C, h2, w2, _pad64(C), 1.0 / self.config.scaling_factor,
            self.config.shift_factor, _stream()))
        img = self._decode_haloed(z)
        self._pool.put(z)
        return self._image_out(img, output_u8)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        """diffusers signature: z NCHW [B,16,h,w] (already un-scaled) -> sample NCHW [B,3,8h,8w] (fp32)."""
        self._ready()
        B, C, h, w = z.shape
        zin = self._pool.get(B, h, w, _pad64(C))
        z32 = z.to(F32).contiguous()
        native.check("rt_nchw_to_haloed_nhwc", native.load().rt_nchw_to_haloed_nhwc(
            z32.data_ptr(), zin.data_ptr(), B, C, h, w, _pad64(C), _stream()))
        img = self._decode_haloed(zin)
        self._pool.put(zin)
        out = self._image_out(img, False).to(z.dtype if z.dtype in (BF16, F32) else F32)
        return DecoderOutput(sample=out) if return_dict else (out,)

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x NCHW [B,3,H,W] in [-1,1] -> latent_dist (mean/logvar NCHW fp32->x.dtype)."""
        self._ready()
        e = self.encoder
        B, C, H, W = x.shape
        xin = self._pool.get(B, H, W, _pad64(C))
        lib = native.load()
        x32 = x.to(F32).contiguous()
        native.check("rt_nchw_to_haloed_nhwc", lib.rt_nchw_to_haloed_nhwc(x32.data_ptr(), xin.data_ptr(), B, C, H, W, _pad64(C), _stream()))
        h = self._conv(e.conv_in, xin)
        self._pool.put(xin)
        for blk in e.down_blocks:
            for r in blk.resnets:
                h = self._resnet(r, h)
            if hasattr(blk, "downsamplers"):
                dn = self._conv(blk.downsamplers[0].conv, h, stride=2)
                self._pool.put(h)
                h = dn
        h = self._mid_block(e.mid_block, h)
        t = self._gn(e.conv_norm_out, h, True)
        self._pool.put(h)
        mom = self._conv(e.conv_out, t)
        self._pool.put(t)
        Bm, Hp, Wp, Cm = mom.shape
        out = torch.empty(B, Cm, Hp - 2, Wp - 2, device=x.device, dtype=F32)
        native.check("rt_haloed_nhwc_to_nchw", lib.rt_haloed_nhwc_to_nchw(mom.data_ptr(), out.data_ptr(), B, Cm, Hp - 2, Wp - 2, Cm, _stream()))
        self._pool.put(mom)
        mean, logvar = out[:, : Cm // 2], out[:, Cm // 2 :]
        dt = x.dtype if x.dtype in (BF16, F32) else F32
        dist = DiagonalGaussianDistribution(mean.to(dt), logvar.to(dt))
        return AutoencoderKLOutput(latent_dist=dist) if return_dict else (dist,)
