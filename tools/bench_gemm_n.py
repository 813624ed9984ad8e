import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reptext_amd.ops as ops
from bench_kernels import timeit
dev = torch.device("cuda:0")
M, K = 4608, 3072
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
for N in (3072, 6144, 9216, 12288, 15360, 18432, 21504):
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.linear(a, w, out, bias=b))
    tiles = 18 * (N // 256)
    print(f"N={N:6d} tiles={tiles:5d} rounds={tiles/256:5.2f}: {t*1e6:7.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s  us/round(ceil)={t*1e6/-(-tiles//256):6.1f}", flush=True)
