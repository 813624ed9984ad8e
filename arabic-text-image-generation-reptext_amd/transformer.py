"""FluxTransformer2DModel — the denoiser the reference imports from diffusers (PIPE:31) and calls at PIPE:1092-1104.

Not present under /root/reference; behaviour per SURVEY.md Appendix A.3. Same constructor config keys, same
``forward`` keyword names and return convention (``(sample,)`` when ``return_dict=False``), HIP kernels underneath.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Sequence, Any, Dict, List, Optional

import torch
import torch.nn as nn

from . import mmdit, ops
from .config import Config, flux_dev_transformer_config
from .modules import DoubleBlockParams, Lin, SingleBlockParams, TimeTextEmbedParams, WeightsIO, _ada
from .ops import LinearProblem as P


@dataclass
class Transformer2DModelOutput:
    sample: torch.Tensor


@dataclass
class StaticEmbeds:
    """Loop-invariant embeddings of one model for one call of the pipeline (``_MMDiTBase.prepare_static``): the prompt never
    changes between denoising steps and neither do the hint latents, so ``context_embedder(prompt)`` (A.3 / CN:292) and
    ``controlnet_x_embedder(cond)`` (CN:280) are evaluated once per image instead of once per step. Both are kept in the
    residual stream's dtype; adding them back is exact (same operands as the per-step path, fp32 addition commutes)."""

    ctx: torch.Tensor                     # [Bc, T, d]  context_embedder(encoder_hidden_states) + bias
    hint: Optional[torch.Tensor] = None   # [Bc, N, d]  controlnet_x_embedder(controlnet_cond) + bias (towers only)


class _MMDiTBase(nn.Module, WeightsIO):
    """Construction, planning and embedding steps common to the transformer and the ControlNet tower."""

    def _build_trunk(self, cfg: Config, device, dtype):
        kw = dict(device=device, dtype=dtype)
        d = cfg.num_attention_heads * cfg.attention_head_dim
        if cfg.attention_head_dim != 128:
            raise ValueError("the HIP attention kernels are specialised for attention_head_dim == 128")
        if sum(cfg.axes_dims_rope) != cfg.attention_head_dim:
            raise ValueError("sum(axes_dims_rope) must equal attention_head_dim")
        self.inner_dim = d
        self.time_text_embed = TimeTextEmbedParams(d, cfg.pooled_projection_dim, bool(cfg.guidance_embeds), **kw)
        self.context_embedder = Lin(cfg.joint_attention_dim, d, **kw)
        self.x_embedder = Lin(cfg.in_channels, d, **kw)
        self.transformer_blocks = nn.ModuleList([DoubleBlockParams(d, cfg.attention_head_dim, **kw) for _ in range(cfg.num_layers)])
        self.single_transformer_blocks = nn.ModuleList([SingleBlockParams(d, cfg.attention_head_dim, **kw) for _ in range(cfg.num_single_layers)])
        self._plans = None
        self._plan_key = None
        self._rope_cache = {}

    # ---- properties the pipelines read (PIPE:906)
    @property
    def dtype(self):
        return self.x_embedder.weight.dtype

    @property
    def device(self):
        return self.x_embedder.weight.device

    def random_init_(self, seed: int = 0, std: float = 0.02, bias_std: float = 0.02):
        """Synthetic weights for benchmarks (no checkpoints offline): W ~ N(0, std²), small biases, norm weights ≈ 1."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith("norm_q.weight") or name.endswith("norm_k.weight") or name.endswith("norm_added_q.weight") or name.endswith("norm_added_k.weight"):
                p.data.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32))
            elif name.endswith(".bias"):
                p.data.copy_(bias_std * torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32))
            else:
                p.data.copy_(std * torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32))
        self._plans = None
        return self

    def _ensure_plans(self):
        """(Re)build fused weight views when parameters moved (``.to``) or were never planned."""
        key = (self.x_embedder.weight.data_ptr(), str(self.device), self.dtype)
        if self._plans is not None and self._plan_key == key:
            return self._plans
        if self.dtype != torch.bfloat16:
            raise TypeError(f"{type(self).__name__}: HIP path computes in bf16 storage; got {self.dtype}. Use torch_dtype=torch.bfloat16.")
        if not self.x_embedder.weight.is_cuda:
            raise RuntimeError(f"{type(self).__name__} is on {self.device}; move it to the GPU (.to('cuda')). There is no CPU fallback.")
        fp8, fa = getattr(self, "_fp8_linears", False), getattr(self, "_fp8_attention", False)
        self._plans = ([mmdit.plan_double(b, fp8, fa) for b in self.transformer_blocks], [mmdit.plan_single(b, fp8, fa) for b in self.single_transformer_blocks])
        self._plan_key = key
        self._rope_cache = {}
        return self._plans

    def _apply(self, fn, *a, **k):
        self._plans = None
        return super()._apply(fn, *a, **k)

    def enable_fp8_attention(self, on: bool = True):
        """BASELINE config 5 ("CDNA4 fp8 MFMA attention"): joint attention with e4m3 q, k, v and softmax numerators
        (rt_attention_fp8_prep / rt_attention_fp8_fwd; static quantisation, see csrc/attention_fp8.hip)."""
        self._fp8_attention = bool(on)
        self._plans = None
        return self

    def enable_fp8_linears(self, on=True):
        """BASELINE config 5 ("fp8 weights"): run the projections that read a LayerNorm output — to_q/k/v, add_q/k/v_proj,
        ff.net.0, ff_context.net.0, proj_mlp (58 % of the block's GEMM FLOPs) — on the e4m3 MFMA path. Weights are quantised
        per output channel when the plans are (re)built, activations per token inside the LayerNorm kernel; everything else
        (attention, out/down projections, residual stream, adaLN) is unchanged. Accuracy: e4m3 carries 3 mantissa bits —
        see DESIGN.md §4 for the measured floor. ``on="all"`` adds to_out / to_add_out, ff.net.2 and proj_out, whose bf16 inputs
        (attention output, GELU hidden) are quantised per token by one extra pass each. ``on="mx"``: the same projections, but
        those inputs carry one E8M0 scale per 32 elements (MX block scaling, applied by the MFMA's scale operand) and are written
        as e4m3 by the epilogue that computes them — the GELU hidden by the ff.net.0 / fused GEMM, the attention output by the
        e4m3 attention kernel (a block-quantise pass only when the bf16 attention kernel is in use): no separate passes."""
        if on not in (True, False, "ln", "all", "mx"):
            raise ValueError("enable_fp8_linears: True/'ln' (LayerNorm-fed projections), 'all' / 'mx' (every block projection), or False")
        self._fp8_linears = "ln" if on is True else on
        self._plans = None
        return self

    def load_state_dict(self, sd, strict: bool = True, **kw):
        out = super().load_state_dict(sd, strict=strict, **kw)
        if getattr(self, "_fp8_linears", False):
            self._plans = None          # the e4m3 copies of the weights are stale
        return out

    def _rope(self, txt_ids: torch.Tensor, img_ids: torch.Tensor):
        """cos/sin [S,128] fp32; constant over the denoising loop, so cached on the id tensors' identity."""
        if txt_ids.dim() == 3:
            txt_ids = txt_ids[0]
        if img_ids.dim() == 3:
            img_ids = img_ids[0]
        key = (txt_ids.data_ptr(), img_ids.data_ptr(), txt_ids.shape[0], img_ids.shape[0])
        hit = self._rope_cache.get(key)
        if hit is None:
            ids = torch.cat([txt_ids.to(torch.float32), img_ids.to(torch.float32)], dim=0)
            hit = ops.rope_table(ids, tuple(self.config.axes_dims_rope), 10000.0)
            if len(self._rope_cache) > 8:
                self._rope_cache.clear()
            self._rope_cache[key] = hit + (txt_ids, img_ids)   # keep the id tensors alive so data_ptr stays unique
        return hit[0], hit[1]

    def _temb(self, ws, timestep, guidance, pooled):
        ref16 = mmdit.REF_BF16_SCALARS
        t1000 = mmdit.bf16_round_trip_x1000(timestep.reshape(-1)) if ref16 else timestep.to(torch.float32).reshape(-1) * 1000.0
        if t1000.numel() == 1 and ws.B > 1:
            t1000 = t1000.expand(ws.B)
        g1000 = None
        if self.config.guidance_embeds:
            if guidance is None:
                raise ValueError("guidance_embeds=True requires `guidance`")
            g1000 = (mmdit.bf16_round_trip_x1000(guidance.reshape(-1)) if ref16 else guidance.to(torch.float32).reshape(-1) * 1000.0).contiguous()
            if g1000.numel() == 1 and ws.B > 1:
                g1000 = g1000.expand(ws.B).contiguous()
        return mmdit.time_text_embed(self.time_text_embed, ws, t1000.contiguous(), g1000, pooled)

    def prepare_static(self, encoder_hidden_states: torch.Tensor, controlnet_cond: Optional[torch.Tensor] = None) -> StaticEmbeds:
        """See StaticEmbeds. ``controlnet_cond`` [Bc,N,in+extra] only for models that own a ``controlnet_x_embedder``."""
        self._ensure_plans()
        d = self.inner_dim
        xdt = torch.float32 if mmdit.RESIDUAL_F32 else torch.bfloat16
        e = encoder_hidden_states.to(torch.bfloat16).contiguous()
        ctx = torch.empty(e.shape[0], e.shape[1], d, device=e.device, dtype=xdt)
        ops.linear(e, self.context_embedder.weight.data, ctx, bias=self.context_embedder.bias.data)
        hint = None
        if controlnet_cond is not None:
            cond, cxw = self._padded_hint(controlnet_cond.to(torch.bfloat16))
            if cond.shape[0] != e.shape[0]:
                cond = cond.expand(e.shape[0], -1, -1) if cond.shape[0] == 1 else cond.repeat(e.shape[0] // cond.shape[0], 1, 1)
            hint = torch.empty(e.shape[0], cond.shape[1], d, device=e.device, dtype=xdt)
            ops.linear(cond.contiguous(), cxw, hint, bias=self.controlnet_x_embedder.bias.data)
        return StaticEmbeds(ctx, hint)

    def build_modulation_table(self, timesteps, guidance, pooled) -> "mmdit.ModulationTable":
        """adaLN vectors of every block for every entry of ``timesteps`` (model-scale values, i.e. t/1000 as the pipeline
        passes them, PIPE:1048,1094) — see mmdit.ModulationTable. guidance [B] / pooled [B,P] as in ``forward``."""
        doubles, singles = self._ensure_plans()
        B, d = pooled.shape[0], self.inner_dim
        dev = pooled.device
        sc = mmdit.EmbedScratch(B, d, dev)
        temb_all = torch.empty(len(timesteps) * B, d, device=dev, dtype=torch.float32)
        for i, t in enumerate(timesteps):
            ts = torch.full((B,), float(t), device=dev, dtype=torch.float32)
            temb_all[i * B : (i + 1) * B].copy_(self._temb(sc, ts, guidance, pooled))
        return mmdit.ModulationTable(temb_all, len(timesteps), B, doubles, singles, getattr(self, "norm_out", None) and self.norm_out.linear)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, subfolder: Optional[str] = None, device=None, **unused):
        d = cls._resolve_dir(path, subfolder)
        cfg = Config.from_json_file(d + "/" + cls.config_name)
        model = cls(**cfg, device=device or "cpu", dtype=torch_dtype or torch.bfloat16)
        sd = cls._load_safetensors_dir(d)
        model.load_state_dict({k: v.to(torch_dtype or torch.bfloat16) for k, v in sd.items()}, strict=True)
        return model


class FluxTransformer2DModel(_MMDiTBase):
    def __init__(self, patch_size: int = 1, in_channels: int = 64, out_channels: Optional[int] = None, num_layers: int = 19,
                 num_single_layers: int = 38, attention_head_dim: int = 128, num_attention_heads: int = 24,
                 joint_attention_dim: int = 4096, pooled_projection_dim: int = 768, guidance_embeds: bool = False,
                 axes_dims_rope=(16, 56, 56), device=None, dtype=None):
        super().__init__()
        self.config = Config(patch_size=patch_size, in_channels=in_channels, out_channels=out_channels, num_layers=num_layers,
                             num_single_layers=num_single_layers, attention_head_dim=attention_head_dim,
                             num_attention_heads=num_attention_heads, joint_attention_dim=joint_attention_dim,
                             pooled_projection_dim=pooled_projection_dim, guidance_embeds=guidance_embeds,
                             axes_dims_rope=list(axes_dims_rope))
        self.out_channels = out_channels or in_channels
        self._build_trunk(self.config, device, dtype)
        d = self.inner_dim
        self.norm_out = _ada(d, 2, device=device, dtype=dtype)
        self.proj_out = Lin(d, patch_size * patch_size * self.out_channels, device=device, dtype=dtype)

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor = None,
                pooled_projections: torch.Tensor = None, timestep: torch.Tensor = None, img_ids: torch.Tensor = None,
                txt_ids: torch.Tensor = None, guidance: torch.Tensor = None,
                joint_attention_kwargs: Optional[Dict[str, Any]] = None, controlnet_block_samples=None,
                controlnet_single_block_samples=None, return_dict: bool = True, controlnet_blocks_repeat: bool = False,
                _mods: Optional["mmdit.StepMods"] = None, _sample_events: Optional[Sequence["torch.cuda.Event"]] = None,
                _static: Optional[StaticEmbeds] = None):
        """``_sample_events[k]`` (optional): event another stream records when controlnet_block_samples[k] is complete; the
        current stream waits for it right before the first block that consumes that sample. ``_static`` (optional): this
        model's loop-invariant embeddings (prepare_static) — the text rows are copied instead of recomputed."""
        doubles, singles = self._ensure_plans()
        cfg = self.config
        B, N, _ = hidden_states.shape
        Bc = encoder_hidden_states.shape[0]       # conditioning batch may be a multiple of B (inpaint CFG, Q6)
        T = encoder_hidden_states.shape[1]
        H, d = cfg.num_attention_heads, self.inner_dim
        ws = mmdit.workspace(Bc, T, N, d, hidden_states.device, need_single=len(singles) > 0)
        hs = hidden_states.to(torch.bfloat16)
        if Bc != B:
            if Bc % B:
                raise ValueError("conditioning batch must be a multiple of the latent batch")
            hs = hs.repeat(Bc // B, 1, 1)
        if _static is not None:
            ws.x[:, :T].copy_(_static.ctx)                                   # device copy of the loop-invariant text rows
            ops.linear(hs.contiguous(), self.x_embedder.weight.data, ws.x[:, T:], bias=self.x_embedder.bias.data)
        else:
            ops.linear_grouped([P(hs.contiguous(), self.x_embedder.weight.data, ws.x[:, T:], bias=self.x_embedder.bias.data),
                                P(encoder_hidden_states.to(torch.bfloat16).contiguous(), self.context_embedder.weight.data, ws.x[:, :T],
                                  bias=self.context_embedder.bias.data)])
        temb = None if _mods is not None else self._temb(ws, timestep, guidance, pooled_projections)
        cos, sin = self._rope(txt_ids, img_ids)
        nl, ns = len(doubles), len(singles)
        waited = set()
        for i, pl in enumerate(doubles):
            inj = None
            if controlnet_block_samples is not None:
                ns_c = len(controlnet_block_samples)
                k = i % ns_c if controlnet_blocks_repeat else i // int(math.ceil(nl / ns_c))
                inj = controlnet_block_samples[k]
                if _sample_events is not None and k not in waited:
                    torch.cuda.current_stream().wait_event(_sample_events[k])
                    waited.add(k)
            mmdit.run_double(pl, ws, temb, cos, sin, H, inject=inj, mods=None if _mods is None else _mods.double[i])
        for i, pl in enumerate(singles):
            inj = None
            if controlnet_single_block_samples is not None:
                inj = controlnet_single_block_samples[i // int(math.ceil(ns / len(controlnet_single_block_samples)))]
            mmdit.run_single(pl, ws, temb, cos, sin, H, inject=inj, mods=None if _mods is None else _mods.single[i])
        # AdaLayerNormContinuous: chunk order (scale, shift)  — A.3
        if _mods is not None:
            m = _mods.out
        else:
            m = ws.mod_a[:, : 2 * d]
            ops.gemv(temb, self.norm_out.linear.weight.data, self.norm_out.linear.bias.data, m, silu_in=True)
        ops.layernorm_modulate(ws.x[:, T:], ws.xn[:, T:], m[:, d : 2 * d], m[:, :d])
        out = torch.empty(Bc, N, self.proj_out.weight.shape[0], device=hs.device, dtype=torch.bfloat16)
        ops.linear(ws.xn[:, T:], self.proj_out.weight.data, out, bias=self.proj_out.bias.data)
        if not return_dict:
            return (out,)
        return Transformer2DModelOutput(sample=out)
