"""Which lane's scale byte does v_mfma_scale_f32_16x16x128_f8f6f4 apply to which block? Structured scale patterns through rt_gemm_fp8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import reptext_amd.ops as ops

gpu = torch.device("cuda", 0)
FP8 = torch.float8_e4m3fn
g = torch.Generator().manual_seed(0)
M, N, K = 256, 256, 512
a8 = (torch.rand(1, M, K, generator=g) + 0.5).to(FP8)
w8 = (torch.rand(N, K, generator=g) + 0.5).to(FP8)


def run(sb):
    sc = ops.BlockScales.empty(1, M, K, gpu)
    sc.t.fill_(127)
    sc.set_rowmajor(sb.permute(1, 0, 2).reshape(1, M, K // 32))
    out = torch.empty(1, M, N, device=gpu, dtype=torch.float32)
    ops.linear(a8.to(gpu), w8.to(gpu), out, a_bscale=sc)
    ref = torch.einsum("bmk,nk->bmn", ops.dequantize_mx(a8.to(gpu), sc).cpu().double(), w8.float().double())
    return out.cpu().double(), ref


base = torch.full((K // 256, M, 8), 127, dtype=torch.uint8)
o, r = run(base)
print("all 127: rel", float((o - r).norm() / r.norm()))
o, r = run(base + 3)
print("all 130: rel", float((o - r).norm() / r.norm()), "ratio", float((o / r).mean()))
sb = base.clone(); sb += (torch.arange(M) % 4).to(torch.uint8)[None, :, None]
o, r = run(sb)
print("by row%4: rel", float((o - r).norm() / r.norm())); print((o / r)[0, :8, :8])
for blk in range(16):
    sb = base.clone(); sb.view(K // 256, M, 8)[blk // 8, :, blk % 8] = 131
    o, r = run(sb)
    o0, r0 = run(base)
    # which k-range got the x16? solve by comparing with candidates
    best = None
    for cand in range(16):
        sbc = base.clone(); sbc[cand // 8, :, cand % 8] = 131
        sc = ops.BlockScales.empty(1, M, K, gpu); sc.set_rowmajor(sbc.permute(1, 0, 2).reshape(1, M, K // 32))
        rc = torch.einsum("bmk,nk->bmn", ops.dequantize_mx(a8.to(gpu), sc).cpu().double(), w8.float().double())
        e = float((o - rc).norm() / rc.norm())
        if best is None or e < best[1]:
            best = (cand, e)
    print(f"scale at block {blk}: kernel behaves as block {best[0]} (rel {best[1]:.1e}); vs intended {float((o - r).norm() / r.norm()):.2e}")
