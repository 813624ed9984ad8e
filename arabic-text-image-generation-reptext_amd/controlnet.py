"""FluxControlNetModel — MI355X implementation of the reference's ControlNet tower.

Interface parity target: /root/reference/RepText/controlnet_flux.py
  * constructor config keys and defaults ............ CN:45-60
  * forward keyword names, defaults, return shapes ... CN:216-230, CN:398-413
  * FluxControlNetOutput ............................. CN:35-38
  * FluxMultiControlNetModel (residual sums) ......... CN:416-529
The tower itself (CN:277-396) is re-expressed as a sequence of HIP launches over the shared MMDiT workspace
(mmdit.py); there is no torch compute on the path and no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from . import mmdit, ops
from .config import Config
from .modules import Lin
from .ops import LinearProblem as P
from .transformer import StaticEmbeds, _MMDiTBase


@dataclass
class FluxControlNetOutput:
    controlnet_block_samples: Optional[List[torch.Tensor]]
    controlnet_single_block_samples: Optional[List[torch.Tensor]]

    def __iter__(self):          # lets callers unpack like the reference's tuple return
        yield self.controlnet_block_samples
        yield self.controlnet_single_block_samples


class FluxControlNetModel(_MMDiTBase):
    _supports_gradient_checkpointing = False   # inference-only build

    def __init__(self, patch_size: int = 1, in_channels: int = 64, num_layers: int = 19, num_single_layers: int = 38,
                 attention_head_dim: int = 128, num_attention_heads: int = 24, joint_attention_dim: int = 4096,
                 pooled_projection_dim: int = 768, guidance_embeds: bool = False, axes_dims_rope=(16, 56, 56),
                 num_mode: Optional[int] = None, extra_conditioning_channels: int = 0, extra_condition_channels: int = 0,
                 device=None, dtype=None):
        super().__init__()
        self.config = Config(patch_size=patch_size, in_channels=in_channels, num_layers=num_layers,
                             num_single_layers=num_single_layers, attention_head_dim=attention_head_dim,
                             num_attention_heads=num_attention_heads, joint_attention_dim=joint_attention_dim,
                             pooled_projection_dim=pooled_projection_dim, guidance_embeds=guidance_embeds,
                             axes_dims_rope=list(axes_dims_rope), num_mode=num_mode,
                             extra_conditioning_channels=extra_conditioning_channels,
                             extra_condition_channels=extra_condition_channels)
        self.out_channels = in_channels
        self._build_trunk(self.config, device, dtype)
        d = self.inner_dim
        kw = dict(device=device, dtype=dtype)
        self.controlnet_blocks = nn.ModuleList([Lin(d, d, **kw) for _ in range(num_layers)])
        self.controlnet_single_blocks = nn.ModuleList([Lin(d, d, **kw) for _ in range(num_single_layers)])
        self.union = num_mode is not None
        if self.union:
            self.controlnet_mode_embedder = nn.Embedding(num_mode, d, **kw)
        # only extra_condition_channels widens the hint embedder (CN:112-114); extra_conditioning_channels is inert
        self.controlnet_x_embedder = Lin(in_channels + extra_condition_channels, d, **kw)
        self.gradient_checkpointing = False

    def zero_init_controlnet_(self):
        """The reference constructs the zero-linears and the hint embedder zero-initialised (zero_module, CN:101,105,112)."""
        for m in list(self.controlnet_blocks) + list(self.controlnet_single_blocks) + [self.controlnet_x_embedder]:
            m.weight.data.zero_()
            m.bias.data.zero_()
        return self

    # -- attention-processor registry of the reference (CN:118-176). One fused HIP attention exists; the API is kept
    #    for signature parity and validates the count like the reference does.
    @property
    def attn_processors(self) -> Dict[str, Any]:
        names = [f"transformer_blocks.{i}.attn.processor" for i in range(len(self.transformer_blocks))]
        names += [f"single_transformer_blocks.{i}.attn.processor" for i in range(len(self.single_transformer_blocks))]
        return {n: "HipFlashAttention" for n in names}

    def set_attn_processor(self, processor):
        count = len(self.attn_processors)
        if isinstance(processor, dict) and len(processor) != count:
            raise ValueError(
                f"A dict of processors was passed, but the number of processors {len(processor)} does not match the"
                f" number of attention layers: {count}. Please make sure to pass {count} processor classes.")
        raise NotImplementedError("the MI355X path has a single fused attention kernel; custom processors are not supported")

    @classmethod
    def from_transformer(cls, transformer, num_layers: int = 4, num_single_layers: int = 10, attention_head_dim: int = 128,
                         num_attention_heads: int = 24, extra_condition_channels: int = 0,
                         load_weights_from_transformer: bool = True):
        """CN:182-214: build a tower from a transformer's config and copy its trunk weights."""
        cfg = dict(transformer.config)
        cfg.pop("out_channels", None)
        cfg.update(num_layers=num_layers, num_single_layers=num_single_layers, attention_head_dim=attention_head_dim,
                   num_attention_heads=num_attention_heads, extra_condition_channels=extra_condition_channels)
        cn = cls(**cfg, device=transformer.device, dtype=transformer.dtype)
        if load_weights_from_transformer:
            src = transformer.state_dict()
            own = cn.state_dict()
            for k in own:
                if k in src and own[k].shape == src[k].shape:
                    own[k].copy_(src[k])
        cn.zero_init_controlnet_()
        return cn

    def _padded_hint(self, cond: torch.Tensor):
        """(cond, weight) of controlnet_x_embedder with K padded to a multiple of 64 — e.g. the inpaint tower's 64 + 4 = 68
        hint channels (INP:807-813): the MFMA K-loop steps by 64, so both operands get zero columns. The padded weight is
        cached until the parameter is rewritten (load_state_dict / random_init_ / .to clear it)."""
        cxw = self.controlnet_x_embedder.weight.data
        kin = cxw.shape[1]
        if kin % 64 == 0:
            return cond, cxw
        kp = (kin + 63) // 64 * 64
        if getattr(self, "_cx_pad", None) is None or self._cx_pad.device != cxw.device:
            wpad = torch.zeros(cxw.shape[0], kp, device=cxw.device, dtype=cxw.dtype)
            wpad[:, :kin] = cxw
            self._cx_pad = wpad
        return torch.nn.functional.pad(cond, (0, kp - kin)), self._cx_pad

    def _invalidate_derived(self):
        self._cx_pad = None

    def load_state_dict(self, sd, strict: bool = True, **kw):
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._invalidate_derived()
        return out

    def random_init_(self, *a, **k):
        out = super().random_init_(*a, **k)
        self._invalidate_derived()
        return out

    def _apply(self, fn, *a, **k):
        self._invalidate_derived()
        return super()._apply(fn, *a, **k)

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, controlnet_cond: torch.Tensor, controlnet_mode: torch.Tensor = None,
                conditioning_scale: float = 1.0, encoder_hidden_states: torch.Tensor = None,
                pooled_projections: torch.Tensor = None, timestep: torch.Tensor = None, img_ids: torch.Tensor = None,
                txt_ids: torch.Tensor = None, guidance: torch.Tensor = None,
                joint_attention_kwargs: Optional[Dict[str, Any]] = None, return_dict: bool = True,
                _rowscale: Optional[torch.Tensor] = None, _accumulate_into: Optional[Sequence[torch.Tensor]] = None,
                _accumulate_single_into: Optional[Sequence[torch.Tensor]] = None, _mods: Optional["mmdit.StepMods"] = None,
                _overwrite: bool = False, _sample_events: Optional[Sequence["torch.cuda.Event"]] = None, _ws_tag: str = "",
                _static: Optional[StaticEmbeds] = None, _blocks_needed: Optional[Tuple[int, int]] = None):
        """Same contract as CN:216-413. ``joint_attention_kwargs`` is accepted and ignored (LoRA scale plumbing, no
        PEFT on this path). The private ``_rowscale`` / ``_accumulate_into`` arguments let the pipeline fuse its
        regional mask (PIPE:1062) and the sum over text lines (PIPE:1076-1080) into the zero-linear epilogues; with
        ``_overwrite`` the ``_accumulate_into`` buffers are written, not added to (first text line into preallocated
        buffers). ``_sample_events[i]`` is recorded on the current stream once double-block sample i is complete and
        ``_ws_tag`` selects a private workspace — both for running the tower on a side stream next to the transformer.
        ``_static``: this tower's loop-invariant embeddings for (prompt, this hint) from ``prepare_static`` — the per-step work
        is then x_embedder only. ``_blocks_needed`` = (double, single): evaluate only that many leading blocks; the samples of
        the rest come back as None (the pipeline knows which samples the transformer consumes: with 6 samples against 19
        blocks the sixth is never read, quirk Q5)."""
        doubles, singles = self._ensure_plans()
        cfg = self.config
        if self.union:
            # CN:294-301. RepText weights have num_mode=None; kept as an explicit error path rather than a silent skip.
            if controlnet_mode is None:
                raise ValueError("`controlnet_mode` cannot be `None` when applying ControlNet-Union")
            raise NotImplementedError("ControlNet-Union mode embedding is outside the RepText hot path (SURVEY.md §2 #11)")
        B, N, _ = hidden_states.shape
        Bc, T, _ = encoder_hidden_states.shape
        H, d = cfg.num_attention_heads, self.inner_dim
        ws = mmdit.workspace(Bc, T, N, d, hidden_states.device, need_single=len(singles) > 0, tag=_ws_tag)
        hs = hidden_states.to(torch.bfloat16)
        if hs.shape[0] != Bc:          # Q6: latents batch B against conditioning batch 2B broadcasts (B == 1 under CFG)
            hs = hs.expand(Bc, -1, -1) if hs.shape[0] == 1 else hs.repeat(Bc // hs.shape[0], 1, 1)
        x_i, x_t = ws.x[:, T:], ws.x[:, :T]
        # CN:277-292: h = x_embedder(latents) + controlnet_x_embedder(cond); e = context_embedder(prompt)
        if _static is not None:
            x_t.copy_(_static.ctx)                                           # device copy of the loop-invariant text rows
            ops.linear(hs.contiguous(), self.x_embedder.weight.data, x_i, bias=self.x_embedder.bias.data, res=_static.hint)
        else:
            cond = controlnet_cond.to(torch.bfloat16)
            if cond.shape[0] != Bc:
                cond = cond.expand(Bc, -1, -1) if cond.shape[0] == 1 else cond.repeat(Bc // cond.shape[0], 1, 1)
            ops.linear_grouped([P(hs.contiguous(), self.x_embedder.weight.data, x_i, bias=self.x_embedder.bias.data),
                                P(encoder_hidden_states.to(torch.bfloat16).contiguous(), self.context_embedder.weight.data, x_t,
                                  bias=self.context_embedder.bias.data)])
            cond, cxw = self._padded_hint(cond)
            ops.linear(cond.contiguous(), cxw, x_i, bias=self.controlnet_x_embedder.bias.data, res=x_i)
        temb = None if _mods is not None else self._temb(ws, timestep, guidance, pooled_projections)
        cos, sin = self._rope(txt_ids, img_ids)

        scale = float(conditioning_scale)

        def head(lin, dst_list, i):
            """zero-linear i on the current image rows: (W·h+b)·scale [·mask] [+ running sum] (CN:384-396)."""
            a = mmdit.image_rows_bf16(ws)
            if dst_list is not None:
                out = dst_list[i]
                ops.linear(a, lin.weight.data, out, bias=lin.bias.data, alpha=scale, rowscale=_rowscale, res=None if _overwrite else out)
            else:
                out = torch.empty(Bc, N, d, device=hs.device, dtype=torch.bfloat16)
                ops.linear(a, lin.weight.data, out, bias=lin.bias.data, alpha=scale, rowscale=_rowscale)
            return out

        nd = len(doubles) if _blocks_needed is None else min(len(doubles), int(_blocks_needed[0]))
        ns = len(singles) if _blocks_needed is None else min(len(singles), int(_blocks_needed[1]))
        if ns > 0:
            nd = len(doubles)          # the single blocks read the stream all double blocks have written
        block_samples: List[Optional[torch.Tensor]] = []
        for i, pl in enumerate(doubles):
            if i >= nd:
                block_samples.append(None)
                continue
            mmdit.run_double(pl, ws, temb, cos, sin, H, mods=None if _mods is None else _mods.double[i])
            block_samples.append(head(self.controlnet_blocks[i], _accumulate_into, i))
            if _sample_events is not None:
                _sample_events[i].record(torch.cuda.current_stream())
        single_samples: List[Optional[torch.Tensor]] = []
        for i, pl in enumerate(singles):
            if i >= ns:
                single_samples.append(None)
                continue
            mmdit.run_single(pl, ws, temb, cos, sin, H, mods=None if _mods is None else _mods.single[i])
            single_samples.append(head(self.controlnet_single_blocks[i], _accumulate_single_into, i))

        bs = block_samples if block_samples else None
        ss = single_samples if single_samples else None
        if not return_dict:
            return (bs, ss)
        return FluxControlNetOutput(controlnet_block_samples=bs, controlnet_single_block_samples=ss)


class FluxMultiControlNetModel(nn.Module):
    """Residual-summing wrapper over several towers (CN:416-529). Kept for import/signature parity; the RepText
    scripts never build one."""

    def __init__(self, controlnets, union: bool = False):
        super().__init__()
        self.nets = nn.ModuleList(controlnets)
        self.union = union

    @torch.no_grad()
    def forward(self, hidden_states, controlnet_cond, controlnet_mode, conditioning_scale, encoder_hidden_states=None,
                pooled_projections=None, timestep=None, img_ids=None, txt_ids=None, guidance=None,
                joint_attention_kwargs=None, return_dict: bool = True):
        nets = [self.nets[0]] * len(controlnet_cond) if len(self.nets) == 1 else list(self.nets)
        total_b = total_s = None
        for net, image, mode, scale in zip(nets, controlnet_cond, controlnet_mode, conditioning_scale):
            b, s = net(hidden_states=hidden_states, controlnet_cond=image, controlnet_mode=None if mode is None else mode[:, None],
                       conditioning_scale=scale, timestep=timestep, guidance=guidance, pooled_projections=pooled_projections,
                       encoder_hidden_states=encoder_hidden_states, txt_ids=txt_ids, img_ids=img_ids,
                       joint_attention_kwargs=joint_attention_kwargs, return_dict=False)
            if total_b is None and total_s is None:
                total_b, total_s = b, s
                continue
            if b is not None and total_b is not None:
                for acc, t in zip(total_b, b):
                    ops.masked_accumulate_(acc, t, None, 1.0, True)
            if s is not None and total_s is not None:
                for acc, t in zip(total_s, s):
                    ops.masked_accumulate_(acc, t, None, 1.0, True)
        return total_b, total_s
