"""Epilogue share of the GEMM: same M,N at K=64 (one K-tile) vs K=3072, plain / gelu / gate+res."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reptext_amd.ops as ops
from bench_kernels import timeit
dev = torch.device("cuda:0")
for (M, N) in [(4608, 21504), (4608, 3072), (4096, 12288)]:
    for K in (64, 3072):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        gate = torch.randn(1, N, device=dev)
        res = torch.randn(M, N, device=dev).to(torch.bfloat16)
        t0 = timeit(lambda: ops.linear(a, w, out))
        t1 = timeit(lambda: ops.linear(a, w, out, bias=b, gelu_from=0))
        t2 = timeit(lambda: ops.linear(a, w, res, bias=b, gate=gate, res=res))
        print(f"M={M} N={N} K={K}: plain {t0*1e6:7.1f} us | bias+gelu {t1*1e6:7.1f} us | bias+gate+res(in place) {t2*1e6:7.1f} us", flush=True)
