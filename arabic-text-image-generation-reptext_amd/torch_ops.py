"""PyTorch custom-op registration of the HIP kernels (`torch.ops.reptext_amd.*`), SURVEY.md §8b.

BASELINE.json's north_star words the boundary as "Python host code calling hand-written HIP kernels through PyTorch-ROCm custom
ops". The compute boundary of this repository is the C ABI (include/reptext_hip.h) and `ops.py` reaches it through ctypes; this
module registers the same entry points with `torch.library` so that they are also addressable the way the reference's ecosystem
addresses operators: as dispatcher ops that take and return `torch.Tensor`s on a HIP device, run on the current stream, and show
up in the profiler / `torch.ops` namespace under their own names. Every op mutates a caller-provided output (declared through
`mutates_args`), allocates nothing and has no CPU implementation: a CPU tensor raises, as everywhere on this path.

`enable_dispatch(True)` (or RT_USE_TORCH_OPS=1) routes `ops.linear / attention / layernorm_modulate / qk_rmsnorm_rope` of the
model code through the dispatcher instead of calling ctypes directly. Off by default: the dispatcher adds ~10 us of host time to
each of the ~13 000 launches of an image, and the results are bit-identical either way (tests/test_kernels_gpu.py).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import ops

_LIB = "reptext_amd"
_direct = {}


def _register():
    if getattr(torch.ops, _LIB, None) is not None and hasattr(torch.ops.reptext_amd, "linear"):
        return
    co = torch.library.custom_op

    @co(f"{_LIB}::linear", mutates_args=("out",), device_types="cuda")
    def linear(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, bias: Optional[torch.Tensor] = None,
               gate: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None, add2: Optional[torch.Tensor] = None,
               rowscale: Optional[torch.Tensor] = None, gelu_from: int = -1, alpha: float = 1.0) -> None:
        """rt_gemm_bf16 with the fused epilogue of include/reptext_hip.h (one problem)."""
        _direct["linear"](a, w, out, bias=bias, gate=gate, res=res, add2=add2, rowscale=rowscale,
                          gelu_from=None if gelu_from < 0 else gelu_from, alpha=alpha)

    @co(f"{_LIB}::attention", mutates_args=("out",), device_types="cuda")
    def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, heads: int, scale: float = 0.0) -> None:
        """rt_attention_fwd (joint attention, head dim 128); scale 0 = 1/sqrt(128)."""
        _direct["attention"](q, k, v, out, heads, None if scale == 0.0 else scale)

    @co(f"{_LIB}::layernorm_modulate", mutates_args=("out",), device_types="cuda")
    def layernorm_modulate(x: torch.Tensor, out: torch.Tensor, shift: Optional[torch.Tensor] = None, scale: Optional[torch.Tensor] = None,
                           eps: float = 1e-6) -> None:
        _direct["layernorm_modulate"](x, out, shift, scale, eps)

    @co(f"{_LIB}::qk_rmsnorm_rope", mutates_args=("buf",), device_types="cuda")
    def qk_rmsnorm_rope(buf: torch.Tensor, q_off: int, k_off: int, heads: int, text_rows: int, wq_txt: Optional[torch.Tensor],
                        wk_txt: Optional[torch.Tensor], wq_img: torch.Tensor, wk_img: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                        eps: float = 1e-6) -> None:
        _direct["qk_rmsnorm_rope"](buf, q_off, k_off, heads, text_rows, wq_txt, wk_txt, wq_img, wk_img, cos, sin, eps)

    @co(f"{_LIB}::gemv", mutates_args=("out",), device_types="cuda")
    def gemv(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, silu_in: bool = False,
             silu_out: bool = False, accumulate: bool = False) -> None:
        _direct["gemv"](x, w, bias, out, silu_in=silu_in, silu_out=silu_out, accumulate=accumulate)

    @co(f"{_LIB}::euler_step_f32", mutates_args=("x32", "x_bf16"), device_types="cuda")
    def euler_step_f32(x32: torch.Tensor, v: torch.Tensor, dsigma: float, x_bf16: Optional[torch.Tensor] = None) -> None:
        _direct["euler_step_f32_"](x32, v, dsigma, x_bf16)


for _name in ("linear", "attention", "layernorm_modulate", "qk_rmsnorm_rope", "gemv", "euler_step_f32_"):
    _direct[_name] = getattr(ops, _name)
_register()


def enable_dispatch(on: bool = True) -> None:
    """Route the model code's calls of the registered kernels through torch.ops.reptext_amd.* (True) or straight to ctypes."""
    t = torch.ops.reptext_amd
    if on:
        ops.linear = lambda a, w, out, bias=None, gate=None, res=None, add2=None, rowscale=None, rows_per_batch=0, gelu_from=None, alpha=1.0, **kw: \
            (_direct["linear"](a, w, out, bias=bias, gate=gate, res=res, add2=add2, rowscale=rowscale, rows_per_batch=rows_per_batch,
                               gelu_from=gelu_from, alpha=alpha, **kw) if (rows_per_batch or kw) else
             t.linear(a, w, out, bias, gate, res, add2, rowscale, -1 if gelu_from is None else int(gelu_from), float(alpha))) or out
        ops.attention = lambda q, k, v, out, H, scale=None, split=True: (t.attention(q, k, v, out, H, 0.0 if scale is None else float(scale)) if split
                                                                          else _direct["attention"](q, k, v, out, H, scale, split=False)) or out
        ops.layernorm_modulate = lambda x, out, shift, scale, eps=1e-6: t.layernorm_modulate(x, out, shift, scale, eps) or out
        ops.qk_rmsnorm_rope = lambda buf, q_off, k_off, H, T, a, b, c, d, cos, sin, eps=1e-6: t.qk_rmsnorm_rope(buf, q_off, k_off, H, T, a, b, c, d, cos, sin, eps)
    else:
        for name in ("linear", "attention", "layernorm_modulate", "qk_rmsnorm_rope"):
            setattr(ops, name, _direct[name])


if os.environ.get("RT_USE_TORCH_OPS", "0") == "1":
    enable_dispatch(True)
