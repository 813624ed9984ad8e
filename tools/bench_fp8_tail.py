"""Is it worth quantising bf16 activations (attention output, GELU hidden) in a separate pass to run the remaining projections
(to_out / ff.net.2 / proj_out) on the e4m3 path? Times quantise pass + fp8 GEMM against the bf16 GEMM at the C2 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reptext_amd.ops as ops
from reptext_amd import native
from tools.bench_kernels import timeit, dev

FP8 = torch.float8_e4m3fn
for name, M, N, K, ld in [("single proj_out", 4608, 3072, 15360, 21504), ("ff.net.2", 4608, 3072, 12288, 12288), ("to_out", 4608, 3072, 3072, 9216)]:
    src = torch.randn(M, ld, device=dev).to(torch.bfloat16)
    a = src[:, ld - K:]
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    w8, sw = ops.quantize_rows_fp8(w)
    a8 = torch.empty(M, K, device=dev, dtype=FP8)
    sa = torch.empty(M, device=dev)
    res = torch.randn(M, N, device=dev)
    out = torch.empty(M, N, device=dev)
    gate = torch.randn(1, N, device=dev)
    lib = native.load()
    q = lambda: native.check("q", lib.rt_quantize_rows_fp8(a.data_ptr(), ld, 0, a8.data_ptr(), K, sa.data_ptr(), M, K, ops._stream()))
    t_q = timeit(q)
    t_8 = timeit(lambda: ops.linear(a8, w8, out, gate=gate, res=res, a_scale=sa, w_scale=sw))
    t_16 = timeit(lambda: ops.linear(a, w, out, gate=gate, res=res))
    print(f"{name:16s} M={M} N={N} K={K}: bf16 {t_16*1e6:7.1f} us | quantise {t_q*1e6:6.1f} + fp8 {t_8*1e6:7.1f} = {(t_q+t_8)*1e6:7.1f} us", flush=True)
