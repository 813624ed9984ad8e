"""Run ONE kernel family a few times (for rocprofv3 --pmc passes): python tools/prof_one.py attn|gemm|attn8|gemm8"""
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
elif what == "gemm8":
    FP8 = torch.float8_e4m3fn
    M, N, K = 4608, 21504, 3072
    a = torch.randn(M, K, device=dev).to(FP8)
    w = torch.randn(N, K, device=dev).to(FP8)
    sa, sw = torch.rand(M, device=dev) + 0.5, torch.rand(N, device=dev) * 0.02
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.linear(a, w, out, a_scale=sa, w_scale=sw)
elif what == "attn8":
    from reptext_amd import native
    FP8 = torch.float8_e4m3fn
    B, S, H = 1, 4608, 24
    d = H * 128
    qkv = torch.randn(B, S, 3 * d, device=dev).to(torch.bfloat16)
    wn = torch.ones(128, device=dev, dtype=torch.bfloat16)
    cos, sin = torch.ones(S, 128, device=dev), torch.zeros(S, 128, device=dev)
    qk8 = torch.empty(B, S, 2 * d, device=dev, dtype=FP8)
    vt8 = torch.empty(int(native.load().rt_attention_fp8_vt_bytes(B, S, H)), device=dev, dtype=FP8)
    ops.attention_fp8_prep(qkv, 0, d, 2 * d, H, 512, wn, wn, wn, wn, cos, sin, qk8, vt8)
    out = torch.empty(B, S, d, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.attention_fp8(qk8, vt8, out, H)
else:
    M, N, K = 4608, 21504, 3072
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        ops.linear(a, w, out)
torch.cuda.synchronize()
