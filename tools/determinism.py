"""Run each kernel repeatedly on identical inputs and report bitwise mismatches (race screen)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reptext_amd.ops as ops
P = ops.LinearProblem
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
rb = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)

def screen(name, fn, out, n=12):
    fn(); torch.cuda.synchronize(); ref = out.clone(); bad = 0; worst = 0
    for _ in range(n):
        out.zero_(); fn(); torch.cuda.synchronize()
        d = (out != ref)
        c = int(d.sum())
        if c: bad += 1; worst = max(worst, c)
    print(f"{name:55s} mismatching runs {bad}/{n}  worst #elements {worst}", flush=True)

for (M, N, K) in [(4608, 21504, 3072), (4608, 3072, 15360), (4096, 3072, 12288), (4096, 12288, 3072), (512, 9216, 3072)]:
    a, w = rb(M, K), rb(N, K) * 0.02
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    screen(f"gemm {M}x{N}x{K}", lambda: ops.linear(a, w, out), out)
T, Ni, d = 512, 4096, 3072
x = rb(2, T + Ni, d); wi, wt = rb(3 * d, d) * 0.02, rb(3 * d, d) * 0.02
out = torch.empty(2, T + Ni, 3 * d, device=dev, dtype=torch.bfloat16)
screen("grouped qkv B=2", lambda: ops.linear_grouped([P(x[:, T:], wi, out[:, T:]), P(x[:, :T], wt, out[:, :T])]), out)
res = rb(2, T + Ni, d); gate = torch.randn(2, d, device=dev, generator=g)
h = rb(2, T + Ni, 4 * d); w2 = rb(d, 4 * d) * 0.02; wt2 = rb(d, 4 * d) * 0.02
o = res.clone()
def ff2():
    o.copy_(res)
    ops.linear_grouped([P(h[:, T:], w2, o[:, T:], gate=gate, res=o[:, T:]), P(h[:, :T], wt2, o[:, :T], gate=gate, res=o[:, :T])])
screen("grouped ff2 gate+res in place B=2", ff2, o)
qkv = rb(2, 4608, 3 * d); ao = torch.empty(2, 4608, d, device=dev, dtype=torch.bfloat16)
screen("attention B=2 S=4608 H=24", lambda: ops.attention(qkv[..., :d], qkv[..., d:2*d], qkv[..., 2*d:], ao, 24), ao)
q2 = qkv.clone()
def att_inplace():
    q2.copy_(qkv)
    ops.attention(q2[..., :d], q2[..., d:2*d], q2[..., 2*d:], q2[..., :d], 24)
screen("attention in place over q", att_inplace, q2)
print("batch rows equal (attention):", bool(torch.equal(ao[0], ao[0])))
# batch equality with identical samples
x1 = rb(1, T + Ni, d); x2 = x1.repeat(2, 1, 1).contiguous()
o2 = torch.empty(2, T + Ni, 3 * d, device=dev, dtype=torch.bfloat16)
ops.linear_grouped([P(x2[:, T:], wi, o2[:, T:]), P(x2[:, :T], wt, o2[:, :T])]); torch.cuda.synchronize()
print("grouped qkv identical samples equal:", bool(torch.equal(o2[0], o2[1])))
qq = rb(1, 4608, 3 * d).repeat(2, 1, 1).contiguous(); oo = torch.empty(2, 4608, d, device=dev, dtype=torch.bfloat16)
ops.attention(qq[..., :d], qq[..., d:2*d], qq[..., 2*d:], oo, 24); torch.cuda.synchronize()
print("attention identical samples equal:", bool(torch.equal(oo[0], oo[1])))
mod = torch.randn(2, 2 * d, device=dev, generator=g); mod[1] = mod[0]
ln = torch.empty(2, T + Ni, d, device=dev, dtype=torch.bfloat16)
ops.layernorm_modulate(x2, ln, mod[:, :d], mod[:, d:]); torch.cuda.synchronize()
print("layernorm identical samples equal:", bool(torch.equal(ln[0], ln[1])))
temb = torch.randn(1, d, device=dev, generator=g).repeat(2, 1).contiguous(); wa = rb(6 * d, d) * 0.02
y = torch.empty(2, 6 * d, device=dev)
ops.gemv(temb, wa, None, y, silu_in=True); torch.cuda.synchronize()
print("gemv identical samples equal:", bool(torch.equal(y[0], y[1])))
cos = torch.randn(4608, 128, device=dev, generator=g); sin = torch.randn(4608, 128, device=dev, generator=g)
wn = rb(128)
qr = qq.clone(); ops.qk_rmsnorm_rope(qr, 0, d, 24, 512, wn, wn, wn, wn, cos, sin); torch.cuda.synchronize()
print("qk_rmsnorm_rope identical samples equal:", bool(torch.equal(qr[0], qr[1])))
