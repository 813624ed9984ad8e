"""Shared execution engine of the MMDiT stacks (FLUX transformer and ControlNet tower).

Host side only: it sequences the HIP ops of ops.py over preallocated HBM buffers. Math per SURVEY.md
Appendix A.1 (double-stream block), A.2 (single-stream block), A.5 (embeddings).

HBM layout (B images, T text tokens, N image tokens, S = T+N, d = H·128), all bf16 unless noted:
  x    [B,S,d]    residual stream, TEXT ROWS FIRST (the order attention concatenates them, A.1 step 4)
  xn   [B,S,d]    LayerNorm+modulation output = A operand of the projections
  qkv  [B,S,3d]   double blocks: [q|k|v]; attention output overwrites q in place
  big  [B,S,7d]   single blocks: [k|v|q|mlp]; attention output overwrites q, so proj_out reads the
                  contiguous [attn|mlp] = big[..., 2d:7d] with K = 5d — no concat, no copy
  ffh  [B,S,4d]   double-block feed-forward hidden (image and text rows in one buffer)
  mod  f32        adaLN vectors: [B,6d] per stream (double), [B,3d] (single)
Image and text projections of a double block are issued as ONE grouped GEMM launch each.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import torch

from . import native, ops
from .ops import LinearProblem as P

import os

BF16, F32 = torch.bfloat16, torch.float32
# dtype of the residual stream x. The reference's bf16 run keeps it in bf16 (one rounding per residual add); fp32 (default)
# removes those roundings — measured on MI355X: rel-L2 vs the fp32 oracle 4.8e-3 -> 2.3e-3 on the 2+2-block transformer, for
# +1.9 % time per image (the GEMM epilogue and LayerNorm kernels take either dtype). RT_RESIDUAL_F32=0 selects bf16.
RESIDUAL_F32 = os.environ.get("RT_RESIDUAL_F32", "1") == "1"

# The reference's bf16 run rounds the SCALARS that feed the sinusoidal embeddings: PIPE:1025 casts t to the latents' dtype,
# PIPE:1048/1094 divide by 1000 in bf16 and CN:282-284 multiply by 1000 in bf16 again — t = 967.3 enters the embedding as 968,
# guidance 3.5 as 3504. The default here (and in the oracle) is the exact fp32 value, i.e. the fp32 CPU run the parity target
# names; `reference_bf16_scalars(True)` (pipelines: `pipe.reference_bf16_scalars = True`) reproduces the bf16 run's values.
REF_BF16_SCALARS = False


def reference_bf16_scalars(on: bool = True) -> None:
    global REF_BF16_SCALARS
    REF_BF16_SCALARS = bool(on)


def bf16_round_trip_x1000(v: torch.Tensor) -> torch.Tensor:
    """fp32 tensor holding what CN:282-284 computes from a model-scale scalar in a bf16 run: bf16(bf16(v) * 1000)."""
    return (v.to(torch.bfloat16) * 1000).to(torch.float32)


@dataclass
class DoublePlan:
    ada_img_w: torch.Tensor
    ada_img_b: torch.Tensor
    ada_txt_w: torch.Tensor
    ada_txt_b: torch.Tensor
    qkv_img_w: torch.Tensor
    qkv_img_b: torch.Tensor
    qkv_txt_w: torch.Tensor
    qkv_txt_b: torch.Tensor
    nq_img: torch.Tensor
    nk_img: torch.Tensor
    nq_txt: torch.Tensor
    nk_txt: torch.Tensor
    out_img_w: torch.Tensor
    out_img_b: torch.Tensor
    out_txt_w: torch.Tensor
    out_txt_b: torch.Tensor
    ff1_img_w: torch.Tensor
    ff1_img_b: torch.Tensor
    ff2_img_w: torch.Tensor
    ff2_img_b: torch.Tensor
    ff1_txt_w: torch.Tensor
    ff1_txt_b: torch.Tensor
    ff2_txt_w: torch.Tensor
    ff2_txt_b: torch.Tensor
    # e4m3 copies (+ per-output-channel scales) of the projections fed by a LayerNorm; None = bf16 path
    qkv_img_w8: Optional[torch.Tensor] = None
    qkv_img_ws: Optional[torch.Tensor] = None
    qkv_txt_w8: Optional[torch.Tensor] = None
    qkv_txt_ws: Optional[torch.Tensor] = None
    ff1_img_w8: Optional[torch.Tensor] = None
    ff1_img_ws: Optional[torch.Tensor] = None
    ff1_txt_w8: Optional[torch.Tensor] = None
    ff1_txt_ws: Optional[torch.Tensor] = None
    # level "all": also the projections whose input is a bf16 activation (attention output, GELU hidden), quantised by a pass
    out_img_w8: Optional[torch.Tensor] = None
    out_img_ws: Optional[torch.Tensor] = None
    out_txt_w8: Optional[torch.Tensor] = None
    out_txt_ws: Optional[torch.Tensor] = None
    ff2_img_w8: Optional[torch.Tensor] = None
    ff2_img_ws: Optional[torch.Tensor] = None
    ff2_txt_w8: Optional[torch.Tensor] = None
    ff2_txt_ws: Optional[torch.Tensor] = None
    fp8_attention: bool = False
    mx: bool = False        # level "mx": those projections read e4m3 + E8M0 block scales written by the producing epilogue


@dataclass
class SinglePlan:
    ada_w: torch.Tensor
    ada_b: torch.Tensor
    fused_w: torch.Tensor   # rows: [to_k | to_v | to_q | proj_mlp]
    fused_b: torch.Tensor
    nq: torch.Tensor
    nk: torch.Tensor
    out_w: torch.Tensor
    out_b: torch.Tensor
    fused_w8: Optional[torch.Tensor] = None
    fused_ws: Optional[torch.Tensor] = None
    out_w8: Optional[torch.Tensor] = None
    out_ws: Optional[torch.Tensor] = None
    fp8_attention: bool = False
    mx: bool = False


def _fuse(lins) -> tuple:
    """Concatenate Linear weights/biases along the output dim and re-point the originals at views of the fused
    storage, so the separate copies are freed and later in-place loads (load_state_dict) stay fused."""
    w = torch.cat([l.weight.data for l in lins], dim=0).contiguous()
    b = torch.cat([l.bias.data for l in lins], dim=0).contiguous()
    o = 0
    for l in lins:
        n = l.weight.shape[0]
        l.weight.data = w[o : o + n]
        l.bias.data = b[o : o + n]
        o += n
    return w, b


def plan_double(blk, fp8=False, fp8_attention: bool = False) -> DoublePlan:
    """fp8: False | 'ln' (LayerNorm-fed projections) | 'all' (every projection of the block, per-row scales, one quantise pass per
    bf16 input) | 'mx' (every projection; the non-LayerNorm inputs as e4m3 + block scales written by their producers)."""
    a = blk.attn
    qi_w, qi_b = _fuse([a.to_q, a.to_k, a.to_v])
    qt_w, qt_b = _fuse([a.add_q_proj, a.add_k_proj, a.add_v_proj])
    pl = DoublePlan(
        blk.norm1.linear.weight.data, blk.norm1.linear.bias.data, blk.norm1_context.linear.weight.data,
        blk.norm1_context.linear.bias.data, qi_w, qi_b, qt_w, qt_b, a.norm_q.weight.data, a.norm_k.weight.data,
        a.norm_added_q.weight.data, a.norm_added_k.weight.data, a.to_out[0].weight.data, a.to_out[0].bias.data,
        a.to_add_out.weight.data, a.to_add_out.bias.data, blk.ff.net[0].proj.weight.data, blk.ff.net[0].proj.bias.data,
        blk.ff.net[2].weight.data, blk.ff.net[2].bias.data, blk.ff_context.net[0].proj.weight.data,
        blk.ff_context.net[0].proj.bias.data, blk.ff_context.net[2].weight.data, blk.ff_context.net[2].bias.data)
    if fp8:
        pl.qkv_img_w8, pl.qkv_img_ws = ops.quantize_rows_fp8(pl.qkv_img_w)
        pl.qkv_txt_w8, pl.qkv_txt_ws = ops.quantize_rows_fp8(pl.qkv_txt_w)
        pl.ff1_img_w8, pl.ff1_img_ws = ops.quantize_rows_fp8(pl.ff1_img_w)
        pl.ff1_txt_w8, pl.ff1_txt_ws = ops.quantize_rows_fp8(pl.ff1_txt_w)
    if fp8 in ("all", "mx"):
        pl.mx = fp8 == "mx"
        if pl.mx and pl.out_img_w.shape[1] % 256:
            raise ValueError("the 'mx' level needs an inner width that is a multiple of 256")
        pl.out_img_w8, pl.out_img_ws = ops.quantize_rows_fp8(pl.out_img_w)
        pl.out_txt_w8, pl.out_txt_ws = ops.quantize_rows_fp8(pl.out_txt_w)
        pl.ff2_img_w8, pl.ff2_img_ws = ops.quantize_rows_fp8(pl.ff2_img_w)
        pl.ff2_txt_w8, pl.ff2_txt_ws = ops.quantize_rows_fp8(pl.ff2_txt_w)
    pl.fp8_attention = fp8_attention
    return pl


def plan_single(blk, fp8=False, fp8_attention: bool = False) -> SinglePlan:
    a = blk.attn
    fw, fb = _fuse([a.to_k, a.to_v, a.to_q, blk.proj_mlp])
    pl = SinglePlan(blk.norm.linear.weight.data, blk.norm.linear.bias.data, fw, fb, a.norm_q.weight.data,
                    a.norm_k.weight.data, blk.proj_out.weight.data, blk.proj_out.bias.data)
    if fp8:
        pl.fused_w8, pl.fused_ws = ops.quantize_rows_fp8(pl.fused_w)
    if fp8 in ("all", "mx"):
        pl.mx = fp8 == "mx"
        if pl.mx and pl.out_w.shape[0] % 256:
            raise ValueError("the 'mx' level needs an inner width that is a multiple of 256")
        pl.out_w8, pl.out_ws = ops.quantize_rows_fp8(pl.out_w)
    pl.fp8_attention = fp8_attention
    return pl


class Workspace:
    """Preallocated activations for one (B,T,N) shape; reused across blocks, steps and models on the same device."""

    def __init__(self, B: int, T: int, N: int, d: int, device, need_single: bool):
        S = T + N
        self.B, self.T, self.N, self.S, self.d = B, T, N, S, d
        e = lambda *shape, dt=BF16: torch.empty(*shape, device=device, dtype=dt)
        self.x = e(B, S, d, dt=F32 if RESIDUAL_F32 else BF16)
        self.xn = e(B, S, d)
        self.qkv = e(B, S, 3 * d)
        self.ffh = e(B, S, 4 * d)
        self.big = e(B, S, 7 * d) if need_single else None
        self.mod_a = e(B, 6 * d, dt=F32)
        self.mod_b = e(B, 6 * d, dt=F32)
        self.temb = e(B, d, dt=F32)
        self.tmp = e(B, d, dt=F32)
        self._device = device
        self.xn8 = None                            # fp8 path (allocated on first use): e4m3 LayerNorm output + row scales
        self.xs_t = self.xs_i = self.xs_all = None

    def fp8_buffers(self):
        if self.xn8 is None:
            B, T, N, S, d = self.B, self.T, self.N, self.S, self.d
            self.xn8 = torch.empty(B, S, d, device=self._device, dtype=ops.FP8)
            self.xs_t = torch.empty(B * T, device=self._device, dtype=F32)
            self.xs_i = torch.empty(B * N, device=self._device, dtype=F32)
            self.xs_all = torch.empty(B * S, device=self._device, dtype=F32)
        return self.xn8

    def fp8_attn_buffers(self, H: int):
        """(qk8 [B,S,2·H·128], vt8 flat) of rt_attention_fp8_prep."""
        if getattr(self, "_qk8", None) is None:
            self._qk8 = torch.empty(self.B, self.S, 2 * H * 128, device=self._device, dtype=ops.FP8)
            self._vt8 = torch.empty(int(native.load().rt_attention_fp8_vt_bytes(self.B, self.S, H)), device=self._device, dtype=ops.FP8)
        return self._qk8, self._vt8

    def fp8_wide(self, width: int) -> torch.Tensor:
        """e4m3 staging for quantised bf16 activations: [B,S,width] view of one buffer sized for the widest use (5d)."""
        if getattr(self, "_a8", None) is None or self._a8.shape[2] < width:
            self._a8 = torch.empty(self.B, self.S, max(width, 5 * self.d), device=self._device, dtype=ops.FP8)
        return self._a8[:, :, :width]


    def mx_scales(self) -> "ops.BlockScales":
        """E8M0 block scales of the fp8_wide buffer (one plane per 256 columns, sized for 5d)."""
        if getattr(self, "_a8s", None) is None:
            self._a8s = ops.BlockScales.empty(self.B, self.S, 5 * self.d, self._device)
        return self._a8s


_WS_CACHE = {}
# While a hipGraph of the loop is being captured (pipeline._denoise) every Workspace handed out is also appended here: the graph
# bakes in their device pointers, so its cache entry must OWN them — this cache may evict or replace an entry at any time.
CAPTURE_KEEP: Optional[list] = None


def workspace(B, T, N, d, device, need_single, tag: str = "") -> Workspace:
    """``tag`` separates the buffers of models that may run concurrently on two streams (pipeline: tower vs transformer)."""
    key = (B, T, N, d, str(device), RESIDUAL_F32, tag)
    ws = _WS_CACHE.get(key)
    if ws is None or (need_single and ws.big is None):
        if len(_WS_CACHE) > 6:
            _WS_CACHE.clear()
        ws = Workspace(B, T, N, d, device, need_single)
        _WS_CACHE[key] = ws
    if CAPTURE_KEEP is not None:
        CAPTURE_KEEP.append(ws)
    return ws


def time_text_embed(tte, ws: Workspace, t1000: torch.Tensor, g1000: Optional[torch.Tensor], pooled: torch.Tensor) -> torch.Tensor:
    """temb = MLP(sinusoid(t)) [+ MLP(sinusoid(g))] + MLP(pooled), fp32 [B,d] (A.5; CN:287-291)."""
    def mlp(m, x_f32, accumulate):
        ops.gemv(x_f32, m.linear_1.weight.data, m.linear_1.bias.data, ws.tmp, silu_out=True)
        ops.gemv(ws.tmp, m.linear_2.weight.data, m.linear_2.bias.data, ws.temb, accumulate=accumulate)

    mlp(tte.timestep_embedder, ops.timestep_embedding(t1000, 256), False)
    if g1000 is not None:
        mlp(tte.guidance_embedder, ops.timestep_embedding(g1000, 256), True)
    pooled_f32 = pooled if pooled.dtype == F32 else ops.to_f32(pooled)
    mlp(tte.text_embedder, pooled_f32.contiguous(), True)
    return ws.temb


class EmbedScratch:
    """The two [B,d] fp32 buffers time_text_embed needs, without a full Workspace."""

    def __init__(self, B: int, d: int, device):
        self.B = B
        self.temb = torch.empty(B, d, device=device, dtype=F32)
        self.tmp = torch.empty(B, d, device=device, dtype=F32)


@dataclass
class StepMods:
    """adaLN vectors of one denoising step, per block: fp32 row-slices of a ModulationTable."""

    double: List[tuple]                  # (image [B,6d], text [B,6d])
    single: List[torch.Tensor]           # [B,3d]
    out: Optional[torch.Tensor]          # [B,2d] (norm_out), transformer only


class ModulationTable:
    """All adaLN projections `linear(silu(temb))` (A.1 step 1, A.2, A.3) of a model for EVERY step of a schedule at once.

    temb depends only on (timestep, guidance, pooled text) — all known before the loop (PIPE:960-967,1025-1032) — so the
    63 per-step weight-streaming GEMVs (6.4 GB of adaLN weights per step) become one M = steps·B GEMM per block, issued
    once per image: the weights are streamed twice (hi + lo pass) instead of `steps` times. The two-term bf16 split of
    silu(temb) keeps fp32-activation accuracy on the bf16 MFMA path."""

    def __init__(self, temb_all: torch.Tensor, n_steps: int, B: int, doubles, singles, out_lin=None):
        self.n, self.B = n_steps, B
        hi, lo = ops.silu_split(temb_all, apply_silu=True)
        dev = temb_all.device
        M = temb_all.shape[0]

        def out_buf(w):
            return torch.empty(M, w.shape[0], device=dev, dtype=F32)

        self.double, self.single, self.out = [], [], None
        for pl in doubles:
            oi, ot = out_buf(pl.ada_img_w), out_buf(pl.ada_txt_w)
            ops.linear_grouped([P(hi, pl.ada_img_w, oi, bias=pl.ada_img_b), P(hi, pl.ada_txt_w, ot, bias=pl.ada_txt_b)])
            ops.linear_grouped([P(lo, pl.ada_img_w, oi, res=oi), P(lo, pl.ada_txt_w, ot, res=ot)])
            self.double.append((oi, ot))
        for pl in singles:
            o = out_buf(pl.ada_w)
            ops.linear(hi, pl.ada_w, o, bias=pl.ada_b)
            ops.linear(lo, pl.ada_w, o, res=o)
            self.single.append(o)
        if out_lin is not None:
            o = out_buf(out_lin.weight.data)
            ops.linear(hi, out_lin.weight.data, o, bias=out_lin.bias.data)
            ops.linear(lo, out_lin.weight.data, o, res=o)
            self.out = o

    def step(self, i: int) -> StepMods:
        r = slice(i * self.B, (i + 1) * self.B)
        return StepMods([(a[r], b[r]) for a, b in self.double], [s[r] for s in self.single], None if self.out is None else self.out[r])


def run_double(pl: DoublePlan, ws: Workspace, temb: torch.Tensor, cos, sin, H: int, inject: Optional[torch.Tensor] = None,
               mods: Optional[tuple] = None) -> None:
    """One FluxTransformerBlock on ws.x in place (A.1). ``inject`` [B,N,d] bf16 is added to the image rows after the
    block (A.3 ControlNet residual), fused into the last GEMM's epilogue. ``mods`` = precomputed (image, text) adaLN
    vectors for this step (ModulationTable); computed here from ``temb`` when absent."""
    T, d = ws.T, ws.d
    x_t, x_i = ws.x[:, :T], ws.x[:, T:]
    xn_t, xn_i = ws.xn[:, :T], ws.xn[:, T:]
    # 1. adaLN-Zero vectors: chunk order shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
    if mods is not None:
        mi, mt = mods
    else:
        mi, mt = ws.mod_a, ws.mod_b
        ops.gemv(temb, pl.ada_img_w, pl.ada_img_b, mi, silu_in=True)
        ops.gemv(temb, pl.ada_txt_w, pl.ada_txt_b, mt, silu_in=True)
    ch = lambda m, i: m[:, i * d : (i + 1) * d]
    fp8 = pl.qkv_img_w8 is not None
    if fp8:
        xn8 = ws.fp8_buffers()
        xn8_t, xn8_i = xn8[:, :T], xn8[:, T:]
    # 2. norm + modulate, 3. q,k,v projections of both streams, one launch
    if fp8:
        ops.layernorm_modulate_fp8(x_i, xn8_i, ws.xs_i, ch(mi, 0), ch(mi, 1))
        ops.layernorm_modulate_fp8(x_t, xn8_t, ws.xs_t, ch(mt, 0), ch(mt, 1))
        ops.linear_grouped([P(xn8_i, pl.qkv_img_w8, ws.qkv[:, T:], bias=pl.qkv_img_b, a_scale=ws.xs_i, w_scale=pl.qkv_img_ws),
                            P(xn8_t, pl.qkv_txt_w8, ws.qkv[:, :T], bias=pl.qkv_txt_b, a_scale=ws.xs_t, w_scale=pl.qkv_txt_ws)])
    else:
        ops.layernorm_modulate(x_i, xn_i, ch(mi, 0), ch(mi, 1))
        ops.layernorm_modulate(x_t, xn_t, ch(mt, 0), ch(mt, 1))
        ops.linear_grouped([P(xn_i, pl.qkv_img_w, ws.qkv[:, T:], bias=pl.qkv_img_b), P(xn_t, pl.qkv_txt_w, ws.qkv[:, :T], bias=pl.qkv_txt_b)])
    q, k, v = ws.qkv[..., :d], ws.qkv[..., d : 2 * d], ws.qkv[..., 2 * d :]
    if pl.fp8_attention:
        # 4.-6. RMSNorm(q,k) + RoPE -> e4m3 q|k and permuted Vᵀ, e4m3 joint attention; output over q ("mx": straight into the
        # out-projections' e4m3 operand with its block scales)
        qk8, vt8 = ws.fp8_attn_buffers(H)
        ops.attention_fp8_prep(ws.qkv, 0, d, 2 * d, H, T, pl.nq_txt, pl.nk_txt, pl.nq_img, pl.nk_img, cos, sin, qk8, vt8)
        if pl.mx:
            ops.attention_fp8_mx(qk8, vt8, ws.fp8_wide(d), ws.mx_scales(), H)
        else:
            ops.attention_fp8(qk8, vt8, q, H)
    else:
        # 4.-5. RMSNorm(q,k) + RoPE in place; 6. joint attention; output over q
        ops.qk_rmsnorm_rope(ws.qkv, 0, d, H, T, pl.nq_txt, pl.nk_txt, pl.nq_img, pl.nk_img, cos, sin)
        ops.attention(q, k, v, q, H)
    # 7./8. x += gate_msa * out_proj(attn)
    if pl.mx:
        a8, sc = ws.fp8_wide(d), ws.mx_scales()
        if not pl.fp8_attention:
            ops.quantize_mx_fp8_into(q, a8, sc)
        ops.linear_grouped([P(a8[:, T:], pl.out_img_w8, x_i, bias=pl.out_img_b, gate=ch(mi, 2), res=x_i, a_bscale=sc.rows(T), w_scale=pl.out_img_ws),
                            P(a8[:, :T], pl.out_txt_w8, x_t, bias=pl.out_txt_b, gate=ch(mt, 2), res=x_t, a_bscale=sc.rows(0), w_scale=pl.out_txt_ws)])
    elif pl.out_img_w8 is not None:
        a8 = ws.fp8_wide(d)
        ops.quantize_rows_fp8_into(q[:, T:], a8[:, T:], ws.xs_i)
        ops.quantize_rows_fp8_into(q[:, :T], a8[:, :T], ws.xs_t)
        ops.linear_grouped([P(a8[:, T:], pl.out_img_w8, x_i, bias=pl.out_img_b, gate=ch(mi, 2), res=x_i, a_scale=ws.xs_i, w_scale=pl.out_img_ws),
                            P(a8[:, :T], pl.out_txt_w8, x_t, bias=pl.out_txt_b, gate=ch(mt, 2), res=x_t, a_scale=ws.xs_t, w_scale=pl.out_txt_ws)])
    else:
        ops.linear_grouped([P(q[:, T:], pl.out_img_w, x_i, bias=pl.out_img_b, gate=ch(mi, 2), res=x_i),
                            P(q[:, :T], pl.out_txt_w, x_t, bias=pl.out_txt_b, gate=ch(mt, 2), res=x_t)])
    if fp8:
        ops.layernorm_modulate_fp8(x_i, xn8_i, ws.xs_i, ch(mi, 3), ch(mi, 4))
        ops.layernorm_modulate_fp8(x_t, xn8_t, ws.xs_t, ch(mt, 3), ch(mt, 4))
        # "mx": the GELU hidden leaves the epilogue as e4m3 + block scales (ffh is not written)
        h8, hs = (ws.fp8_wide(4 * d), ws.mx_scales()) if pl.mx else (None, None)
        o8 = lambda r0, r1: dict(out8=h8[:, r0:r1], out8_scales=hs.rows(r0), out8_from=0) if pl.mx else {}
        ops.linear_grouped([P(xn8_i, pl.ff1_img_w8, ws.ffh[:, T:], bias=pl.ff1_img_b, gelu_from=0, a_scale=ws.xs_i, w_scale=pl.ff1_img_ws, **o8(T, ws.S)),
                            P(xn8_t, pl.ff1_txt_w8, ws.ffh[:, :T], bias=pl.ff1_txt_b, gelu_from=0, a_scale=ws.xs_t, w_scale=pl.ff1_txt_ws, **o8(0, T))])
    else:
        ops.layernorm_modulate(x_i, xn_i, ch(mi, 3), ch(mi, 4))
        ops.layernorm_modulate(x_t, xn_t, ch(mt, 3), ch(mt, 4))
        ops.linear_grouped([P(xn_i, pl.ff1_img_w, ws.ffh[:, T:], bias=pl.ff1_img_b, gelu_from=0),
                            P(xn_t, pl.ff1_txt_w, ws.ffh[:, :T], bias=pl.ff1_txt_b, gelu_from=0)])
    if pl.mx:
        ops.linear_grouped([P(h8[:, T:], pl.ff2_img_w8, x_i, bias=pl.ff2_img_b, gate=ch(mi, 5), res=x_i, add2=inject, a_bscale=hs.rows(T), w_scale=pl.ff2_img_ws),
                            P(h8[:, :T], pl.ff2_txt_w8, x_t, bias=pl.ff2_txt_b, gate=ch(mt, 5), res=x_t, a_bscale=hs.rows(0), w_scale=pl.ff2_txt_ws)])
    elif pl.ff2_img_w8 is not None:
        a8 = ws.fp8_wide(4 * d)
        ops.quantize_rows_fp8_into(ws.ffh[:, T:], a8[:, T:], ws.xs_i)
        ops.quantize_rows_fp8_into(ws.ffh[:, :T], a8[:, :T], ws.xs_t)
        ops.linear_grouped([P(a8[:, T:], pl.ff2_img_w8, x_i, bias=pl.ff2_img_b, gate=ch(mi, 5), res=x_i, add2=inject, a_scale=ws.xs_i, w_scale=pl.ff2_img_ws),
                            P(a8[:, :T], pl.ff2_txt_w8, x_t, bias=pl.ff2_txt_b, gate=ch(mt, 5), res=x_t, a_scale=ws.xs_t, w_scale=pl.ff2_txt_ws)])
    else:
        ops.linear_grouped([P(ws.ffh[:, T:], pl.ff2_img_w, x_i, bias=pl.ff2_img_b, gate=ch(mi, 5), res=x_i, add2=inject),
                            P(ws.ffh[:, :T], pl.ff2_txt_w, x_t, bias=pl.ff2_txt_b, gate=ch(mt, 5), res=x_t)])


def run_single(pl: SinglePlan, ws: Workspace, temb: torch.Tensor, cos, sin, H: int, inject: Optional[torch.Tensor] = None,
               mods: Optional[torch.Tensor] = None) -> None:
    """One FluxSingleTransformerBlock on ws.x in place (A.2)."""
    T, d = ws.T, ws.d
    if mods is not None:
        m = mods
    else:
        m = ws.mod_a[:, : 3 * d]
        ops.gemv(temb, pl.ada_w, pl.ada_b, m, silu_in=True)                   # shift, scale, gate
    big = ws.big
    if pl.fused_w8 is not None:
        xn8 = ws.fp8_buffers()
        ops.layernorm_modulate_fp8(ws.x, xn8, ws.xs_all, m[:, :d], m[:, d : 2 * d])
        # "mx": the gelu(mlp) columns leave the epilogue as columns d.. of proj_out's e4m3 operand [attn|mlp], with their block scales
        o8 = dict(out8=ws.fp8_wide(5 * d)[..., d:], out8_scales=ws.mx_scales().cols(d), out8_from=3 * d) if pl.mx else {}
        ops.linear(xn8, pl.fused_w8, big, bias=pl.fused_b, gelu_from=3 * d, a_scale=ws.xs_all, w_scale=pl.fused_ws, **o8)
    else:
        ops.layernorm_modulate(ws.x, ws.xn, m[:, :d], m[:, d : 2 * d])
        ops.linear(ws.xn, pl.fused_w, big, bias=pl.fused_b, gelu_from=3 * d)    # [k|v|q|gelu(mlp)]
    q = big[..., 2 * d : 3 * d]
    if pl.fp8_attention:
        qk8, vt8 = ws.fp8_attn_buffers(H)
        ops.attention_fp8_prep(big, 2 * d, 0, d, H, 0, None, None, pl.nq, pl.nk, cos, sin, qk8, vt8)
        if pl.mx:
            ops.attention_fp8_mx(qk8, vt8, ws.fp8_wide(5 * d)[..., :d], ws.mx_scales(), H)
        else:
            ops.attention_fp8(qk8, vt8, q, H)
    else:
        ops.qk_rmsnorm_rope(big, 2 * d, 0, H, 0, None, None, pl.nq, pl.nk, cos, sin)
        ops.attention(q, big[..., :d], big[..., d : 2 * d], q, H)
    if pl.mx:
        a8, sc = ws.fp8_wide(5 * d), ws.mx_scales()
        if not pl.fp8_attention:
            ops.quantize_mx_fp8_into(q, a8[..., :d], sc)
        ops.linear(a8, pl.out_w8, ws.x, bias=pl.out_b, gate=m[:, 2 * d : 3 * d], res=ws.x, a_bscale=sc, w_scale=pl.out_ws)
    elif pl.out_w8 is not None:
        a8 = ws.fp8_wide(5 * d)
        ops.quantize_rows_fp8_into(big[..., 2 * d :], a8, ws.xs_all)
        ops.linear(a8, pl.out_w8, ws.x, bias=pl.out_b, gate=m[:, 2 * d : 3 * d], res=ws.x, a_scale=ws.xs_all, w_scale=pl.out_ws)
    else:
        ops.linear(big[..., 2 * d :], pl.out_w, ws.x, bias=pl.out_b, gate=m[:, 2 * d : 3 * d], res=ws.x)
    if inject is not None:                                                       # A.3: image tokens only
        for b in range(ws.B):   # image rows of one batch entry are contiguous; one call per image
            ops.masked_accumulate_(ws.x[b, T:].unsqueeze(0), inject[b : b + 1].contiguous(), None, 1.0, True)


def image_rows_bf16(ws: Workspace) -> torch.Tensor:
    """bf16 view/copy of the image rows of the residual stream, as a GEMM A operand (ControlNet zero-linears, CN:384-392).
    With a bf16 stream this is the stream itself; with an fp32 stream the rows are cast into the (free) xn buffer."""
    T = ws.T
    if ws.x.dtype == BF16:
        return ws.x[:, T:]
    for b in range(ws.B):
        src, dst = ws.x[b, T:], ws.xn[b, T:]
        native.check("rt_cast_f32_to_bf16", native.load().rt_cast_f32_to_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), ops._stream()))
    return ws.xn[:, T:]
