"""Tensor-level wrappers over the C ABI (native.py). PyTorch supplies device memory and the stream only.

Every function validates device/dtype/contiguity, then enqueues ONE native call on torch's current HIP
stream. Nothing here computes with torch ops; a CPU tensor is an error, not a fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import native

BF16 = torch.bfloat16
F32 = torch.float32
FP8 = torch.float8_e4m3fn        # OCP e4m3 (what gfx950's conversion and MFMA instructions use)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str, dtype=None) -> int:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tensor is on {t.device}; the HIP path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def _opt(t: Optional[torch.Tensor], name: str, dtype=None) -> Optional[int]:
    return None if t is None else _dev(t, name, dtype)


def _rowmajor2d(t: torch.Tensor, name: str):
    """Return (rows, cols, ld) for a 2-D view whose last dim is unit-stride."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: need a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.shape[0], t.shape[1], t.stride(0)


def _batched(t: torch.Tensor, name: str):
    """(batch, rows, cols, ld, batch_stride) of a [R,C] or [B,R,C] view with unit inner stride."""
    if t.dim() == 2:
        r, c, ld = _rowmajor2d(t, name)
        return 1, r, c, ld, 0
    if t.dim() != 3 or t.stride(2) != 1:
        raise ValueError(f"{name}: need [R,C] or [B,R,C] with unit inner stride, got {tuple(t.shape)} / {t.stride()}")
    return t.shape[0], t.shape[1], t.shape[2], t.stride(1), t.stride(0)


@dataclass
class BlockScales:
    """E8M0 block scales of an e4m3 activation tensor [B, R, D] (one byte per 32 consecutive elements of a row) in rt_gemm_group's
    layout: ``t`` uint8 [ceil(D/1024), B*Rp/64, 8, 16, 4, 4] = [plane of 1024 columns][64-row chunk][K-tile of 128 columns][row & 15]
    [32-block of the K-tile][row >> 4 & 3], Rp = R rounded up to 64 (rows between batch entries). ``rows(r0)`` / ``cols(k0)``
    address a sub-view of the e4m3 tensor (rows r0.. of every batch entry, r0 % 64 == 0; columns k0.., k0 % 128 == 0: a GEMM's A
    operand must start at column 0, producers may write at any k0)."""

    t: torch.Tensor
    R: int                  # rows between batch entries (multiple of 64)
    row0: int = 0
    k0: int = 0

    @staticmethod
    def empty(B: int, R: int, D: int, device) -> "BlockScales":
        if D % 128:
            raise ValueError("block-scaled rows need D % 128 == 0")
        Rp = (R + 63) // 64 * 64
        return BlockScales(torch.empty((D + 1023) // 1024, B * Rp // 64, 8, 16, 4, 4, device=device, dtype=torch.uint8), Rp)

    @property
    def plane(self) -> int:
        return self.t.stride(0)

    def rows(self, r0: int) -> "BlockScales":
        if r0 % 64:
            raise ValueError("row offset of block scales must be a multiple of 64 (text length T % 64 == 0)")
        return BlockScales(self.t, self.R, self.row0 + r0, self.k0)

    def cols(self, k0: int) -> "BlockScales":
        if k0 % 128:
            raise ValueError("column offset of block scales must be a multiple of 128")
        return BlockScales(self.t, self.R, self.row0, self.k0 + k0)

    def ptr(self) -> int:
        """Device address of the view's row origin (the column origin k0 travels separately)."""
        if not self.t.is_cuda or self.t.dtype != torch.uint8 or not self.t.is_contiguous():
            raise TypeError("block scales must be a contiguous uint8 tensor on the GPU")
        return self.t.data_ptr() + (self.row0 // 64) * 2048

    def columns_left(self) -> int:
        return self.t.shape[0] * 1024 - self.k0

    def _rm(self) -> torch.Tensor:
        """[rows_total, planes*32] row-major view-copy of the whole tensor (row = chunk*64 + i*16 + l15; col = plane*32 + ktile*4 + j)."""
        P = self.t.shape[0]
        return self.t.permute(1, 5, 3, 0, 2, 4).reshape(-1, P * 32)

    def rowmajor(self, B: int, R: int, D: int) -> torch.Tensor:
        """uint8 [B, R, D/32] copy of the view's scale bytes in plain row-major order (tests and tools)."""
        return self._rm().view(-1, self.R, self.t.shape[0] * 32)[:B, self.row0 : self.row0 + R, self.k0 // 32 : (self.k0 + D) // 32].contiguous()

    def set_rowmajor(self, sb: torch.Tensor) -> None:
        """Inverse of rowmajor(): write scale bytes given as uint8 [B, R, D/32] into the view (tests and tools)."""
        B, R, nb = sb.shape
        P = self.t.shape[0]
        full = self._rm().view(-1, self.R, P * 32).clone()
        full[:B, self.row0 : self.row0 + R, self.k0 // 32 : self.k0 // 32 + nb] = sb.to(full.device)
        C = self.t.shape[1]
        self.t.copy_(full.view(C, 4, 16, P, 8, 4).permute(3, 0, 4, 2, 5, 1))


@dataclass
class LinearProblem:
    """One group of rt_gemm_bf16: out = epilogue(a @ w.T + bias).

    a [M,K] or [B,M,K] bf16 views (unit inner stride); w [N,K] bf16; out [.., M, N] bf16|f32.
    gate f32: [B,N] view (one vector per batch) — or, for 2-D problems whose rows are batch-major,
    [M/rows_per_batch, N] with rows_per_batch set. res same dtype/shape as out (may alias it);
    add2 bf16 same shape; rowscale f32 [rows_per_batch or M] shared by the batch, or [B, rows] (one vector per batch entry)."""

    a: torch.Tensor
    w: torch.Tensor
    out: torch.Tensor
    bias: Optional[torch.Tensor] = None
    gate: Optional[torch.Tensor] = None
    res: Optional[torch.Tensor] = None
    add2: Optional[torch.Tensor] = None
    rowscale: Optional[torch.Tensor] = None
    rows_per_batch: int = 0
    gelu_from: Optional[int] = None
    alpha: float = 1.0
    # fp8 problems (rt_gemm_fp8): a and w are float8_e4m3fn, a_scale f32 [B*M] (one per activation row), w_scale f32 [N]
    a_scale: Optional[torch.Tensor] = None
    w_scale: Optional[torch.Tensor] = None
    # MX block scales (fp8 problems): a_bscale = the block scales of a; out8 / out8_scales = e4m3 + block-scale output for the
    # columns >= out8_from (those columns are then not written to out): the next projection's operand, no pass in between
    a_bscale: Optional[BlockScales] = None
    out8: Optional[torch.Tensor] = None
    out8_scales: Optional[BlockScales] = None
    out8_from: int = 0

    @property
    def is_fp8(self) -> bool:
        return self.a.dtype == FP8

    def to_group(self) -> native.GemmGroup:
        Bt, M, K, lda, sA = _batched(self.a, "a")
        N, Kw, ldw = _rowmajor2d(self.w, "w")
        Bo, Mo, No, ldc, sC = _batched(self.out, "out")
        if Kw != K or Mo != M or No != N or Bo != Bt:
            raise ValueError(f"linear shapes mismatch: a{tuple(self.a.shape)} w{tuple(self.w.shape)} out{tuple(self.out.shape)}")
        g = native.GemmGroup()
        op_dtype = FP8 if self.is_fp8 else BF16
        g.A = _dev(self.a, "a", op_dtype)
        g.W = _dev(self.w, "w", op_dtype)
        if self.is_fp8:
            if self.a_scale is not None:
                if self.a_scale.numel() != Bt * M or not self.a_scale.is_contiguous():
                    raise ValueError("a_scale must be contiguous with batch*M elements")
                g.a_scale = _dev(self.a_scale, "a_scale", F32)
            if self.w_scale is not None:
                if self.w_scale.numel() != N or not self.w_scale.is_contiguous():
                    raise ValueError("w_scale must be contiguous with N elements")
                g.w_scale = _dev(self.w_scale, "w_scale", F32)
            if self.a_bscale is not None:
                if self.a_bscale.k0 != 0 or self.a_bscale.columns_left() < K:
                    raise ValueError("block-scaled a: the operand starts at column 0 of its scale tensor, which must cover K")
                g.a_bscale, g.a_bscale_plane, g.a_bscale_rows = self.a_bscale.ptr(), self.a_bscale.plane, self.a_bscale.R
            if self.out8 is not None:
                B8, M8, N8, ld8, s8 = _batched(self.out8, "out8")
                if self.out8_scales is None or (B8, M8) != (Bt, M) or N8 != N - self.out8_from or self.out8_scales.columns_left() < N8:
                    raise ValueError("out8 must be [.., M, N - out8_from] e4m3 with its block scales")
                g.c8, g.c_bscale = _dev(self.out8, "out8", FP8), self.out8_scales.ptr()
                g.ldc8, g.stride_c8, g.c_bscale_plane, g.c_bscale_rows, g.c8_from = ld8, s8, self.out8_scales.plane, self.out8_scales.R, int(self.out8_from)
                g.c_bscale_k0 = self.out8_scales.k0
        elif self.a_scale is not None or self.w_scale is not None or self.a_bscale is not None or self.out8 is not None:
            raise TypeError("a_scale / w_scale / block scales belong to fp8 problems")
        if self.out.dtype not in (BF16, F32):
            raise TypeError("out must be bf16 or f32")
        g.C = _dev(self.out, "out")
        g.out_f32 = 1 if self.out.dtype == F32 else 0
        g.lda, g.ldw, g.ldc = lda, ldw, ldc
        g.strideA, g.strideC = sA, sC
        g.M, g.N, g.K, g.batch = M, N, K, Bt
        g.bias = _opt(self.bias, "bias", BF16)
        if self.bias is not None and self.bias.numel() != N:
            raise ValueError("bias length != N")
        rpb = self.rows_per_batch
        rows = rpb if rpb > 0 else M
        if self.gate is not None:
            gr, gc, gld = _rowmajor2d(self.gate, "gate")
            if gc < N or gr < Bt * (M // rows):
                raise ValueError("gate too small for this problem")
            g.gate = _dev(self.gate, "gate", F32)
            g.gate_ld = gld
        if self.res is not None:
            Br, rr, rc, ldr, sR = _batched(self.res, "res")
            if (Br, rr, rc) != (Bt, M, N) or self.res.dtype != self.out.dtype:
                raise ValueError("res must match out in shape and dtype")
            g.res = _dev(self.res, "res")
            g.ldr, g.strideR = ldr, sR
        if self.add2 is not None:
            B2, ar, ac, ld2, s2 = _batched(self.add2, "add2")
            if (B2, ar, ac) != (Bt, M, N):
                raise ValueError("add2 must match out in shape")
            g.add2 = _dev(self.add2, "add2", BF16)
            g.ld2, g.stride2 = ld2, s2
        if self.rowscale is not None:
            rsc = self.rowscale
            if rsc.dim() == 2 and rsc.shape == (Bt, rows) and rsc.stride(1) == 1 and Bt > 1:
                g.stride_rowscale = rsc.stride(0)                     # one mask per batch entry
            elif rsc.numel() != rows or not rsc.is_contiguous():
                raise ValueError("rowscale must be contiguous with rows_per_batch (or M) elements, or [batch, rows]")
            g.rowscale = _dev(rsc, "rowscale", F32)
        g.rows_per_batch = rpb
        g.gelu_from = N if self.gelu_from is None else int(self.gelu_from)
        g.alpha = float(self.alpha)
        return g


def linear_grouped(problems: Sequence[LinearProblem]) -> None:
    """Enqueue up to RT_GEMM_MAX_GROUPS independent linears as ONE launch (image + text stream of a block)."""
    n = len(problems)
    if not 1 <= n <= native.RT_GEMM_MAX_GROUPS:
        raise ValueError(f"1..{native.RT_GEMM_MAX_GROUPS} problems per launch")
    arr = (native.GemmGroup * n)(*[p.to_group() for p in problems])
    fp8 = problems[0].is_fp8
    if any(p.is_fp8 != fp8 for p in problems):
        raise TypeError("all problems of one launch must have the same operand dtype")
    if fp8:
        native.check("rt_gemm_fp8", native.load().rt_gemm_fp8(arr, n, _stream()))
    else:
        native.check("rt_gemm_bf16", native.load().rt_gemm_bf16(arr, n, _stream()))


def linear(a, w, out, **kw) -> torch.Tensor:
    linear_grouped([LinearProblem(a, w, out, **kw)])
    return out


def gemv(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *, silu_in=False,
         silu_out=False, accumulate=False) -> torch.Tensor:
    """out[b,n] (+)= post(sum_k pre(x[b,k]) w[n,k] + bias[n]); x,out f32, w,bias bf16."""
    B, K, ldx = _rowmajor2d(x, "x")
    N, Kw, ldw = _rowmajor2d(w, "w")
    Bo, No, ldy = _rowmajor2d(out, "out")
    if Kw != K or Bo != B or No != N:
        raise ValueError("gemv shapes mismatch")
    lib = native.load()
    step = 4
    for b0 in range(0, B, step):
        nb = min(step, B - b0)
        native.check("rt_gemv_bf16w", lib.rt_gemv_bf16w(
            _dev(x, "x", F32) + b0 * ldx * 4, ldx, _dev(w, "w", BF16), ldw, _opt(bias, "bias", BF16),
            _dev(out, "out", F32) + b0 * ldy * 4, ldy, nb, N, K, int(silu_in), int(silu_out), int(accumulate), _stream()))
    return out


def timestep_embedding(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    t = t.contiguous()
    out = torch.empty(t.numel(), dim, device=t.device, dtype=F32)
    native.check("rt_timestep_embedding", native.load().rt_timestep_embedding(_dev(t, "t", F32), out.data_ptr(), t.numel(), dim, _stream()))
    return out


def rope_table(ids: torch.Tensor, axes_dim=(16, 56, 56), theta: float = 10000.0):
    ids = ids.to(F32).contiguous()
    S = ids.shape[0]
    D = int(sum(axes_dim))
    cos = torch.empty(S, D, device=ids.device, dtype=F32)
    sin = torch.empty(S, D, device=ids.device, dtype=F32)
    ax = (C.c_int32 * 3)(*[int(a) for a in axes_dim])
    native.check("rt_rope_table", native.load().rt_rope_table(_dev(ids, "ids", F32), cos.data_ptr(), sin.data_ptr(), S, ax, float(theta), _stream()))
    return cos, sin


def layernorm_modulate(x: torch.Tensor, out: torch.Tensor, shift: Optional[torch.Tensor], scale: Optional[torch.Tensor],
                       eps: float = 1e-6) -> torch.Tensor:
    """x [B,R,D] (bf16|f32, unit inner stride) -> out [B,R,D] bf16 = LN(x)*(1+scale[b])+shift[b]; shift/scale f32 [B,D] views."""
    if x.dim() != 3 or out.dim() != 3 or x.shape != out.shape or x.stride(2) != 1 or out.stride(2) != 1:
        raise ValueError("layernorm_modulate: x/out must be [B,R,D] with unit inner stride")
    B, R, D = x.shape
    mod_ld = 0
    if scale is not None:
        if scale.shape != (B, D) or shift.shape != (B, D) or scale.stride(1) != 1 or shift.stride(1) != 1 or scale.stride(0) != shift.stride(0):
            raise ValueError("shift/scale must be [B,D] views with equal row stride")
        mod_ld = scale.stride(0)
    if x.dtype not in (BF16, F32):
        raise TypeError("x must be bf16 or f32")
    native.check("rt_layernorm_modulate", native.load().rt_layernorm_modulate(
        _dev(x, "x"), x.stride(1), x.stride(0), int(x.dtype == F32), _dev(out, "out", BF16), out.stride(1), out.stride(0),
        _opt(shift, "shift", F32), _opt(scale, "scale", F32), mod_ld, B, R, D, float(eps), _stream()))
    return out


def layernorm_modulate_fp8(x: torch.Tensor, out: torch.Tensor, row_scale: torch.Tensor, shift: Optional[torch.Tensor],
                           scale: Optional[torch.Tensor], eps: float = 1e-6) -> torch.Tensor:
    """As layernorm_modulate, quantising each modulated row to e4m3: out [B,R,D] float8_e4m3fn, row_scale f32 [B*R] contiguous
    (row b*R + r) — the A operand and a_scale of an fp8 LinearProblem."""
    if x.dim() != 3 or out.dim() != 3 or x.shape != out.shape or x.stride(2) != 1 or out.stride(2) != 1:
        raise ValueError("layernorm_modulate_fp8: x/out must be [B,R,D] with unit inner stride")
    B, R, D = x.shape
    if row_scale.numel() != B * R or not row_scale.is_contiguous():
        raise ValueError("row_scale must be contiguous with B*R elements")
    mod_ld = 0
    if scale is not None:
        if scale.shape != (B, D) or shift.shape != (B, D) or scale.stride(1) != 1 or shift.stride(1) != 1 or scale.stride(0) != shift.stride(0):
            raise ValueError("shift/scale must be [B,D] views with equal row stride")
        mod_ld = scale.stride(0)
    if x.dtype not in (BF16, F32):
        raise TypeError("x must be bf16 or f32")
    native.check("rt_layernorm_modulate_fp8", native.load().rt_layernorm_modulate_fp8(
        _dev(x, "x"), x.stride(1), x.stride(0), int(x.dtype == F32), _dev(out, "out", FP8), out.stride(1), out.stride(0),
        _dev(row_scale, "row_scale", F32), _opt(shift, "shift", F32), _opt(scale, "scale", F32), mod_ld, B, R, D, float(eps), _stream()))
    return out


def quantize_rows_fp8(x: torch.Tensor):
    """x [rows, D] bf16|f32 (unit inner stride) -> (e4m3 [rows, D], scale f32 [rows]) with x ≈ q * scale[:, None]."""
    rows, D, ldx = _rowmajor2d(x, "x")
    if x.dtype not in (BF16, F32):
        raise TypeError("x must be bf16 or f32")
    q = torch.empty(rows, D, device=x.device, dtype=FP8)
    sc = torch.empty(rows, device=x.device, dtype=F32)
    native.check("rt_quantize_rows_fp8", native.load().rt_quantize_rows_fp8(
        _dev(x, "x"), ldx, int(x.dtype == F32), q.data_ptr(), D, sc.data_ptr(), rows, D, _stream()))
    return q, sc


def quantize_rows_fp8_into(x: torch.Tensor, out: torch.Tensor, scale: torch.Tensor) -> None:
    """x [B,R,D] bf16|f32 view (unit inner stride, any row / batch stride) -> out [B,R,D] e4m3 view, scale f32 [B*R] contiguous
    (row b*R + r): the A operand and a_scale of an fp8 LinearProblem for activations that do not come out of a LayerNorm."""
    if x.dim() != 3 or out.dim() != 3 or x.shape != out.shape or x.stride(2) != 1 or out.stride(2) != 1:
        raise ValueError("quantize_rows_fp8_into: x/out must be [B,R,D] views with unit inner stride")
    B, R, D = x.shape
    if scale.numel() != B * R or not scale.is_contiguous():
        raise ValueError("scale must be contiguous with B*R elements")
    if x.dtype not in (BF16, F32):
        raise TypeError("x must be bf16 or f32")
    lib, st = native.load(), _stream()
    esz = x.element_size()
    _dev(out, "out", FP8), _dev(scale, "scale", F32), _dev(x, "x")
    for b in range(B):
        native.check("rt_quantize_rows_fp8", lib.rt_quantize_rows_fp8(
            x.data_ptr() + b * x.stride(0) * esz, x.stride(1), int(x.dtype == F32), out.data_ptr() + b * out.stride(0), out.stride(1),
            scale.data_ptr() + b * R * 4, R, D, st))


def quantize_mx_fp8_into(x: torch.Tensor, out: torch.Tensor, scales: BlockScales) -> None:
    """x [B,R,D] bf16|f32 view -> out [B,R,D] e4m3 view + E8M0 block scales (one per 32 elements, rt_quantize_mx_fp8): the A operand
    and a_bscale of an fp8 LinearProblem for tensors no fused producer writes."""
    if x.dim() != 3 or out.dim() != 3 or x.shape != out.shape or x.stride(2) != 1 or out.stride(2) != 1:
        raise ValueError("quantize_mx_fp8_into: x/out must be [B,R,D] views with unit inner stride")
    B, R, D = x.shape
    if x.dtype not in (BF16, F32):
        raise TypeError("x must be bf16 or f32")
    if D % 256 or scales.k0 != 0 or scales.columns_left() < D:
        raise ValueError("D % 256 == 0, written from column 0 of a scale tensor that covers D")
    lib, st = native.load(), _stream()
    esz = x.element_size()
    _dev(out, "out", FP8), _dev(x, "x")
    for b in range(B):
        native.check("rt_quantize_mx_fp8", lib.rt_quantize_mx_fp8(
            x.data_ptr() + b * x.stride(0) * esz, x.stride(1), int(x.dtype == F32), out.data_ptr() + b * out.stride(0), out.stride(1),
            scales.ptr() + b * (scales.R // 64) * 2048, scales.plane, R, D, st))


def dequantize_mx(q: torch.Tensor, scales: BlockScales) -> torch.Tensor:
    """fp32 [B,R,D] value of an e4m3 tensor with block scales (host-side helper for tests and tools; torch ops)."""
    B, R, D = q.shape
    e = scales.rowmajor(B, R, D).to(torch.float32) - 127.0
    return (q.to(torch.float32).view(B, R, D // 32, 32) * torch.exp2(e).unsqueeze(-1)).view(B, R, D)


def qk_rmsnorm_rope(buf: torch.Tensor, q_off: int, k_off: int, H: int, T: int, wq_txt, wk_txt, wq_img, wk_img,
                    cos: torch.Tensor, sin: torch.Tensor, eps: float = 1e-6) -> None:
    """In place on buf [B,S,ld] bf16: heads at columns q_off + h*128 / k_off + h*128."""
    if buf.dim() != 3 or buf.stride(2) != 1:
        raise ValueError("buf must be [B,S,ld] with unit inner stride")
    B, S, _ = buf.shape
    if cos.shape != (S, 128) or sin.shape != (S, 128) or not cos.is_contiguous() or not sin.is_contiguous():
        raise ValueError("cos/sin must be contiguous [S,128]")
    native.check("rt_qk_rmsnorm_rope", native.load().rt_qk_rmsnorm_rope(
        _dev(buf, "buf", BF16), buf.stride(1), buf.stride(0), q_off, k_off, _opt(wq_txt, "wq_txt", BF16), _opt(wk_txt, "wk_txt", BF16),
        _dev(wq_img, "wq_img", BF16), _dev(wk_img, "wk_img", BF16), _dev(cos, "cos", F32), _dev(sin, "sin", F32), B, S, T, H, float(eps), _stream()))


_ATTN_WS = {}
CAPTURE_KEEP: Optional[list] = None      # see mmdit.CAPTURE_KEEP: workspaces a graph under capture refers to


def _attention_workspace(B: int, S: int, H: int, device) -> Optional[torch.Tensor]:
    """Workspace of rt_attention_fwd's key-split tail (ticket counters + partial records), one per (shape, device, stream):
    two streams may run attention of the same shape at once (tower beside transformer). Zeroed once — the kernel leaves the
    counters zero. None when this shape splits nothing. A workspace first requested while a stream capture is running is
    zeroed by a memset NODE of that graph (replayed with it); should the capture be abandoned, pipeline._denoise drops the
    entries it created (drop_attention_workspaces), so a never-executed zero fill cannot be picked up later."""
    key = (B, S, H, str(device), _stream())
    ws = _ATTN_WS.get(key, False)
    if ws is False:
        n = int(native.load().rt_attention_ws_bytes(B, S, H))
        ws = torch.zeros(n, device=device, dtype=torch.uint8) if n > 0 else None
        if len(_ATTN_WS) > 16:
            _ATTN_WS.clear()
        _ATTN_WS[key] = ws
    if CAPTURE_KEEP is not None and ws is not None:
        CAPTURE_KEEP.append(ws)
    return ws


def drop_attention_workspaces(keys) -> None:
    for k in list(keys):
        _ATTN_WS.pop(k, None)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, H: int, scale: Optional[float] = None,
              split: bool = True) -> torch.Tensor:
    """q,k,v [B,S,H*128] views (common strides) of one buffer; out [B,S,H*128] view (may alias q). ``split=False`` runs every
    128-row block as one full-length workgroup (no key-split tail; A/B and tests)."""
    for name, t in (("q", q), ("k", k), ("v", v), ("out", out)):
        if t.dim() != 3 or t.stride(2) != 1 or t.shape[2] != H * 128:
            raise ValueError(f"{name}: need [B,S,{H*128}] with unit inner stride")
    B, S, _ = q.shape
    if not (q.stride() == k.stride() == v.stride()) or k.shape != q.shape or v.shape != q.shape or out.shape != q.shape:
        raise ValueError("q,k,v must share shape and strides")
    sc = (128 ** -0.5) if scale is None else float(scale)
    ws = _attention_workspace(B, S, H, q.device) if split else None
    native.check("rt_attention_fwd", native.load().rt_attention_fwd(
        _dev(q, "q", BF16), _dev(k, "k", BF16), _dev(v, "v", BF16), _dev(out, "out", BF16), q.stride(1), q.stride(0),
        out.stride(1), out.stride(0), B, S, H, sc, None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), _stream()))
    return out


def attention_fp8_prep(buf: torch.Tensor, q_off: int, k_off: int, v_off: int, H: int, T: int, wq_txt, wk_txt, wq_img, wk_img,
                       cos: torch.Tensor, sin: torch.Tensor, qk8: torch.Tensor, vt8: torch.Tensor, eps: float = 1e-6) -> None:
    """From the fused projection buffer buf [B,S,ld] bf16 (not modified): qk8 [B,S,2·H·128] e4m3 = 16 · RoPE(RMSNorm(q | k)),
    vt8 flat e4m3 of rt_attention_fp8_vt_bytes(B,S,H) bytes = Vᵀ per (batch, head), keys in MFMA operand order."""
    if buf.dim() != 3 or buf.stride(2) != 1:
        raise ValueError("buf must be [B,S,ld] with unit inner stride")
    B, S, _ = buf.shape
    if cos.shape != (S, 128) or sin.shape != (S, 128) or not cos.is_contiguous() or not sin.is_contiguous():
        raise ValueError("cos/sin must be contiguous [S,128]")
    lib = native.load()
    if qk8.shape != (B, S, 2 * H * 128) or not qk8.is_contiguous() or vt8.numel() < int(lib.rt_attention_fp8_vt_bytes(B, S, H)):
        raise ValueError("qk8 must be contiguous [B,S,2*H*128]; vt8 must hold rt_attention_fp8_vt_bytes(B,S,H) bytes")
    native.check("rt_attention_fp8_prep", lib.rt_attention_fp8_prep(
        _dev(buf, "buf", BF16), buf.stride(1), buf.stride(0), q_off, k_off, v_off, _opt(wq_txt, "wq_txt", BF16), _opt(wk_txt, "wk_txt", BF16),
        _dev(wq_img, "wq_img", BF16), _dev(wk_img, "wk_img", BF16), _dev(cos, "cos", F32), _dev(sin, "sin", F32),
        _dev(qk8, "qk8", FP8), _dev(vt8, "vt8", FP8), B, S, T, H, float(eps), _stream()))


def attention_fp8(qk8: torch.Tensor, vt8: torch.Tensor, out: torch.Tensor, H: int, scale: Optional[float] = None) -> torch.Tensor:
    """softmax(q kᵀ · scale) v from attention_fp8_prep's buffers -> out [B,S,>=H*128] bf16 view (unit inner stride)."""
    B, S, _ = qk8.shape
    if out.dim() != 3 or out.shape[0] != B or out.shape[1] != S or out.stride(2) != 1:
        raise ValueError("out must be [B,S,*] with unit inner stride")
    native.check("rt_attention_fp8_fwd", native.load().rt_attention_fp8_fwd(
        _dev(qk8, "qk8", FP8), _dev(vt8, "vt8", FP8), _dev(out, "out", BF16), out.stride(1), out.stride(0), B, S, H,
        float(scale if scale is not None else 128 ** -0.5), _stream()))
    return out


def attention_fp8_mx(qk8: torch.Tensor, vt8: torch.Tensor, out8: torch.Tensor, scales: BlockScales, H: int, scale: Optional[float] = None) -> torch.Tensor:
    """attention_fp8 with the output as e4m3 + E8M0 block scales (rt_attention_fp8_fwd_mx): out8 [B,S,>=H*128] e4m3 view, head h at
    columns h*128.. of the view; ``scales`` addresses the same view."""
    B, S, _ = qk8.shape
    if out8.dim() != 3 or out8.shape[0] != B or out8.shape[1] != S or out8.stride(2) != 1:
        raise ValueError("out8 must be [B,S,*] with unit inner stride")
    if scales.columns_left() < H * 128:
        raise ValueError("the scale tensor must cover the H*128 output columns")
    native.check("rt_attention_fp8_fwd_mx", native.load().rt_attention_fp8_fwd_mx(
        _dev(qk8, "qk8", FP8), _dev(vt8, "vt8", FP8), _dev(out8, "out8", FP8), out8.stride(1), out8.stride(0), scales.ptr(), scales.plane, scales.R, scales.k0,
        B, S, H, float(scale if scale is not None else 128 ** -0.5), _stream()))
    return out8


def euler_step_(x: torch.Tensor, v: torch.Tensor, dsigma: float) -> torch.Tensor:
    if not (x.is_contiguous() and v.is_contiguous()) or x.shape != v.shape:
        raise ValueError("euler_step_: contiguous tensors of equal shape")
    native.check("rt_euler_step", native.load().rt_euler_step(_dev(x, "x", BF16), _dev(v, "v", BF16), float(dsigma), x.numel(), _stream()))
    return x


def euler_step_f32_(x32: torch.Tensor, v: torch.Tensor, dsigma: float, x_bf16: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x32 += dsigma*v on an fp32 master state; optionally refreshes its bf16 copy."""
    if not (x32.is_contiguous() and v.is_contiguous()) or x32.shape != v.shape:
        raise ValueError("euler_step_f32_: contiguous tensors of equal shape")
    if x_bf16 is not None and (x_bf16.shape != x32.shape or not x_bf16.is_contiguous()):
        raise ValueError("x_bf16 must match x32")
    native.check("rt_euler_step_f32", native.load().rt_euler_step_f32(_dev(x32, "x32", F32), _dev(v, "v", BF16), _opt(x_bf16, "x_bf16", BF16),
                                                                      float(dsigma), x32.numel(), _stream()))
    return x32


def cfg_mix(v_uncond: torch.Tensor, v_text: torch.Tensor, s: float) -> torch.Tensor:
    u, t = v_uncond.contiguous(), v_text.contiguous()
    out = torch.empty_like(t)
    native.check("rt_cfg_mix", native.load().rt_cfg_mix(_dev(u, "u", BF16), _dev(t, "t", BF16), out.data_ptr(), float(s), out.numel(), _stream()))
    return out


def pack_latents(x: torch.Tensor) -> torch.Tensor:
    B, Cc, H2, W2 = x.shape
    x = x.contiguous()
    out = torch.empty(B, (H2 // 2) * (W2 // 2), Cc * 4, device=x.device, dtype=BF16)
    native.check("rt_pack_latents", native.load().rt_pack_latents(_dev(x, "x", BF16), out.data_ptr(), B, Cc, H2, W2, _stream()))
    return out


def unpack_latents_nhwc(packed: torch.Tensor, H2: int, W2: int, scaling: float, shift: float) -> torch.Tensor:
    """[B,(H2/2)(W2/2),4C] -> NHWC [B,H2,W2,C] bf16 holding packed/scaling + shift."""
    B, _, C4 = packed.shape
    Cc = C4 // 4
    packed = packed.contiguous()
    out = torch.empty(B, H2, W2, Cc, device=packed.device, dtype=BF16)
    native.check("rt_unpack_latents", native.load().rt_unpack_latents(_dev(packed, "packed", BF16), out.data_ptr(), B, Cc, H2, W2, 1.0 / float(scaling), float(shift), _stream()))
    return out


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=BF16)
    native.check("rt_cast_f32_to_bf16", native.load().rt_cast_f32_to_bf16(_dev(x, "x", F32), out.data_ptr(), x.numel(), _stream()))
    return out


def to_f32(x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=F32)
    native.check("rt_cast_bf16_to_f32", native.load().rt_cast_bf16_to_f32(_dev(x, "x", BF16), out.data_ptr(), x.numel(), _stream()))
    return out


def masked_accumulate_(y: torch.Tensor, x: torch.Tensor, rowscale: Optional[torch.Tensor], alpha: float = 1.0, accumulate: bool = True) -> torch.Tensor:
    """y[b,r,:] (+)= alpha*rowscale[r]*x[b,r,:]; contiguous [B,R,D]; x bf16, y bf16 or f32."""
    if x.shape != y.shape or x.dim() != 3 or not x.is_contiguous() or not y.is_contiguous() or y.dtype not in (BF16, F32):
        raise ValueError("masked_accumulate_: contiguous [B,R,D] tensors of equal shape (y bf16 or f32)")
    B, R, D = x.shape
    if rowscale is not None and (rowscale.numel() != R or not rowscale.is_contiguous()):
        raise ValueError("rowscale must be contiguous with R elements")
    native.check("rt_masked_accumulate", native.load().rt_masked_accumulate(
        _dev(x, "x", BF16), _dev(y, "y"), _opt(rowscale, "rowscale", F32), float(alpha), B, R, D, int(accumulate),
        int(y.dtype == F32), _stream()))
    return y


def silu_split(x: torch.Tensor, apply_silu: bool = True):
    """(hi, lo) bf16 with hi + lo ~= silu(x) to ~2^-17 relative: A operands for a two-pass bf16 GEMM on fp32 activations."""
    x = x.contiguous()
    hi = torch.empty(x.shape, device=x.device, dtype=BF16)
    lo = torch.empty(x.shape, device=x.device, dtype=BF16)
    native.check("rt_silu_split_bf16", native.load().rt_silu_split_bf16(_dev(x, "x", F32), hi.data_ptr(), lo.data_ptr(), x.numel(), int(apply_silu), _stream()))
    return hi, lo


def resize2d(x: torch.Tensor, size=None, scale_factor: Optional[float] = None, mode: str = "nearest", u8_scale: Optional[float] = None) -> torch.Tensor:
    """torch.nn.functional.interpolate(x, size=/scale_factor=, mode="nearest"|"bilinear", align_corners=False) on the device,
    bit-identical to ATen: x [..., H, W] fp32 — or uint8 with ``u8_scale`` (x / u8_scale is resized) — -> fp32 [..., OH, OW]."""
    if mode not in ("nearest", "bilinear"):
        raise ValueError("resize2d: mode must be 'nearest' or 'bilinear'")
    if (size is None) == (scale_factor is None):
        raise ValueError("resize2d: exactly one of size / scale_factor")
    x = x.contiguous()
    H, W = x.shape[-2], x.shape[-1]
    if size is not None:
        OH, OW = int(size[0]), int(size[1])
        sf = 0.0
    else:
        OH, OW = int(H * float(scale_factor)), int(W * float(scale_factor))   # floor(in * scale), as torch computes the size
        sf = float(scale_factor)
    is_u8 = x.dtype == torch.uint8
    if is_u8 != (u8_scale is not None) or (not is_u8 and x.dtype != F32):
        raise TypeError("resize2d: fp32 input, or uint8 input together with u8_scale")
    planes = x.numel() // (H * W)
    out = torch.empty(*x.shape[:-2], OH, OW, device=x.device, dtype=F32)
    native.check("rt_resize2d", native.load().rt_resize2d(_dev(x, "x"), int(is_u8), float(u8_scale or 1.0), out.data_ptr(), planes, H, W, OH, OW,
                                                      sf, sf, int(mode == "bilinear"), _stream()))
    return out


def glyph_blend(image: torch.Tensor, latents: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """where(bilinear-resized (image > 0).any(channel) > 0, 0.10 * latents + noise, noise) — PIPE:645-654; all fp32, on the device."""
    image, latents, noise = image.contiguous(), latents.contiguous(), noise.contiguous()
    if image.dim() != 4 or latents.shape != noise.shape or latents.dim() != 4 or image.shape[0] != latents.shape[0]:
        raise ValueError("glyph_blend: image [B,C,H,W], latents/noise [B,Cl,OH,OW]")
    B, Cimg, H, W = image.shape
    _, Cl, OH, OW = latents.shape
    out = torch.empty_like(noise)
    native.check("rt_glyph_blend", native.load().rt_glyph_blend(_dev(image, "image", F32), _dev(latents, "latents", F32), _dev(noise, "noise", F32),
                                                             out.data_ptr(), B, Cimg, H, W, Cl, OH, OW, _stream()))
    return out


def canny_u8(img: torch.Tensor, low: float = 50.0, high: float = 100.0, invert: bool = False, out_channels: int = 1) -> torch.Tensor:
    """cv2.Canny(img, low, high) of infer.py:16-22 on the device: img uint8 [H,W] or [H,W,C] -> uint8 [H,W,out_channels] edge map
    {0,255} (255 - edges with ``invert``), bit-identical to hints.canny_edges. One call = four kernels, no host sync."""
    if img.dtype != torch.uint8 or img.dim() not in (2, 3):
        raise TypeError("canny_u8: uint8 [H,W] or [H,W,C]")
    img = img.contiguous()
    H, W = img.shape[0], img.shape[1]
    Cc = 1 if img.dim() == 2 else img.shape[2]
    lib = native.load()
    ws = torch.empty(int(lib.rt_canny_ws_bytes(H, W)), device=img.device, dtype=torch.uint8)
    out = torch.empty(H, W, out_channels, device=img.device, dtype=torch.uint8)
    native.check("rt_canny_u8", lib.rt_canny_u8(_dev(img, "img"), H, W, Cc, float(low), float(high), out.data_ptr(), out_channels, int(invert),
                                               ws.data_ptr(), ws.numel(), _stream()))
    return out


def preprocess_u8(img: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """VaeImageProcessor.preprocess for uint8 images already at the target size: [B,H,W,C] (or [H,W,C] / [H,W]) -> f32 [B,C,H,W]
    = x/255 (then 2x-1), bit-identical to the host path."""
    if img.dtype != torch.uint8:
        raise TypeError("preprocess_u8: uint8 input")
    if img.dim() == 2:
        img = img[None, :, :, None]
    elif img.dim() == 3:
        img = img[None]
    img = img.contiguous()
    B, H, W, Cc = img.shape
    out = torch.empty(B, Cc, H, W, device=img.device, dtype=F32)
    native.check("rt_preprocess_u8", native.load().rt_preprocess_u8(_dev(img, "img"), out.data_ptr(), B, H, W, Cc, int(normalize), _stream()))
    return out
