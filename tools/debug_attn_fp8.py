import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reptext_amd.ops as ops
from reptext_amd import native
from oracle import flux_oracle as orc
gpu = torch.device("cuda:0"); FP8 = torch.float8_e4m3fn
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
B, S, H, T = 1, 256, 2, 64
d = H * 128
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B, S, 3 * d, generator=g).to(torch.bfloat16)
w1 = torch.ones(128, dtype=torch.bfloat16)
ids = torch.zeros(S, 3)
cos, sin = orc.rope_table(ids)           # identity rotation
qk8 = torch.empty(B, S, 2 * d, device=gpu, dtype=FP8)
S64 = (S + 63) // 64 * 64
vt8 = torch.empty(int(native.load().rt_attention_fp8_vt_bytes(B, S, H)), device=gpu, dtype=FP8)
ops.attention_fp8_prep(qkv.to(gpu), 0, d, 2 * d, H, T, w1.to(gpu), w1.to(gpu), w1.to(gpu), w1.to(gpu), cos.to(gpu), sin.to(gpu), qk8, vt8)
# ---- check vt8 against the documented permutation
v = qkv[..., 2 * d:].float().reshape(B, S, H, 128)
vt = vt8.float().cpu().reshape(B, H, 128, S64)
exp = torch.zeros(B, H, 128, S64)
for p in range(64):
    hh, kt, gq, e = p >> 5, (p >> 4) & 1, (p >> 2) & 3, p & 3
    key = 32 * kt + 8 * gq + 4 * hh + e
    for t in range(S64 // 64):
        if 64 * t + key < S:
            exp[:, :, :, 64 * t + p] = v[:, 64 * t + key].permute(0, 1, 2).transpose(1, 2).transpose(1, 2).permute(0, 1, 2).reshape(B, H, 128)
print("vt8 vs expected permutation (e4m3 rounding only):", rel(vt, exp.to(FP8).float()))
# ---- kernel vs torch on the SAME quantised buffers
q8 = qk8.float().cpu()[..., :d].reshape(B, S, H, 128).permute(0, 2, 1, 3)
k8 = qk8.float().cpu()[..., d:].reshape(B, S, H, 128).permute(0, 2, 1, 3)
v8 = v.to(FP8).float().permute(0, 2, 1, 3)
s = (q8 @ k8.transpose(-1, -2)) / (256.0 * math.sqrt(128))
e = torch.exp2((s - s.amax(-1, keepdim=True)) * math.log2(math.e) + 2.0)
o_ref = ((e.to(FP8).float() @ v8) / e.sum(-1, keepdim=True)).permute(0, 2, 1, 3).reshape(B, S, d)
o_ref_nop = ((e @ v8) / e.sum(-1, keepdim=True)).permute(0, 2, 1, 3).reshape(B, S, d)
out = torch.empty(B, S, d, device=gpu, dtype=torch.bfloat16)
ops.attention_fp8(qk8, vt8, out, H)
print("kernel vs torch on same q8/k8/v8 (P e4m3):", rel(out.float().cpu(), o_ref), " (P fp32):", rel(out.float().cpu(), o_ref_nop))
print("torch P-e4m3 vs P-fp32:", rel(o_ref, o_ref_nop))
# uniform attention: q = 0
qk8z = qk8.clone(); qk8z[..., :d] = 0
ops.attention_fp8(qk8z, vt8, out, H)
print("uniform attention vs mean(v8):", rel(out.float().cpu(), v8.mean(2, keepdim=True).expand(-1, -1, S, -1).permute(0, 2, 1, 3).reshape(B, S, d)))
