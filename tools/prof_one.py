"""Run ONE kernel family a few times (for rocprofv3 --pmc passes): python tools/prof_one.py attn|gemm"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reptext_amd.ops as ops
dev = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "attn"
if what == "attn":
    B, S, H = 1, 4608, 24
    d = H * 128
    qkv = torch.randn(B, S, 3 * d, device=dev).to(torch.bfloat16)
    out = torch.empty(B, S, d, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.attention(qkv[..., :d], qkv[..., d:2*d], qkv[..., 2*d:], out, H)
else:
    M, N, K = 4608, 21504, 3072
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.linear(a, w, out)
torch.cuda.synchronize()
