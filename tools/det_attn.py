import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reptext_amd.ops as ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for (B, S, H) in [(1, 4608, 24), (1, 512, 1), (1, 128, 1), (1, 64, 1)]:
    d = H * 128
    qkv = torch.randn(B, S, 3 * d, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(B, S, d, device=dev, dtype=torch.bfloat16)
    ops.attention(qkv[..., :d], qkv[..., d:2*d], qkv[..., 2*d:], out, H); torch.cuda.synchronize()
    ref = out.clone(); bad = 0; worst = 0; maxdiff = 0.0
    for _ in range(10):
        out.zero_(); ops.attention(qkv[..., :d], qkv[..., d:2*d], qkv[..., 2*d:], out, H); torch.cuda.synchronize()
        c = int((out != ref).sum()); bad += c > 0; worst = max(worst, c); maxdiff = max(maxdiff, float((out.float() - ref.float()).abs().max()))
    print(f"ABL={os.environ.get('RT_ATT_ABLATE','0')} B={B} S={S} H={H}: mismatching runs {bad}/10 worst {worst} maxdiff {maxdiff:.4f}", flush=True)
