"""GPU parity of the MX block-scaled e4m3 path (BASELINE config 5, level "mx"): one E8M0 scale per 32 consecutive K-elements of an
activation row, consumed by the MFMA's scale operand (csrc/gemm_bf16.hip, MX = true) and produced inside the epilogues that compute the
activation (rt_gemm_fp8's c8 output, rt_attention_fp8_fwd_mx) or by rt_quantize_mx_fp8.

The quantisation rule is integer / byte work, so its bar is bit-exactness against oracle.flux_oracle.quant_mx_e4m3 on the same fp32
values; the contraction is fp32 accumulation of exact products, so its bar is accumulation-order noise (2e-5)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import flux_oracle as orc  # noqa: E402

FP8 = torch.float8_e4m3fn


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def ops(gpu):
    import reptext_amd.ops as ops

    return ops


def _scales_of(ops, sc, B, R, D):
    """[B, R, D/32] uint8 scale bytes of a BlockScales view in plain row-major order (host copy)."""
    return sc.rowmajor(B, R, D).cpu()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_quantize_mx_is_bit_exact(ops, gpu, dtype):
    g = torch.Generator().manual_seed(5)
    B, R, D = 2, 70, 768
    x = (torch.randn(B, R, D, generator=g) * torch.logspace(-4, 3, R)[None, :, None]).to(dtype)
    x[0, 3] = 0                                                    # all-zero blocks
    x[1, 7, 32:64] = 0
    x[1, 9, 5] = 448.0 * 2.0 ** 3                                  # exactly on a scale boundary
    x[1, 9, 40] = 448.0 * 2.0 ** 3 * (1 + 2.0 ** -7 if dtype == torch.bfloat16 else 1 + 2.0 ** -20)      # just above it
    big = torch.zeros(B, R + 5, D + 256, dtype=dtype)               # strided source view
    big[:, 2 : 2 + R, 256:] = x
    out = torch.zeros(B, R, D, device=gpu, dtype=FP8)
    sc = ops.BlockScales.empty(B, R, D, gpu)
    sc.t.fill_(255)
    ops.quantize_mx_fp8_into(big.to(gpu)[:, 2 : 2 + R, 256:], out, sc)
    q_ref, s_ref = orc.quant_mx_e4m3(x.float(), return_parts=True)
    assert torch.equal(_scales_of(ops, sc, B, R, D), s_ref)
    assert torch.equal(out.cpu().view(torch.uint8), q_ref.view(torch.uint8))
    assert float(out.float().abs().max()) <= 448.0
    # the dequantised value: e4m3 has 3 mantissa bits, rounding <= 2^-4 of the element (or half a subnormal step of the block)
    deq = ops.dequantize_mx(out, sc).cpu()
    step = torch.exp2(s_ref.float() - 127.0 - 10.0).repeat_interleave(32, dim=-1)
    assert bool(((deq - x.float()).abs() <= x.float().abs() * 2.0 ** -4 + step).all())


@pytest.mark.parametrize("M,N,K,batch", [(256, 256, 256, 1), (300, 520, 512, 1), (512, 768, 3072, 2), (4608, 3072, 15360, 1)])
def test_gemm_with_block_scaled_operand(ops, gpu, M, N, K, batch):
    """A carries E8M0 block scales spanning 2^-20 .. 2^20 per block; the products stay exact in fp32, so with f32 output the kernel must
    match an fp32 matmul of the de-quantised operands to accumulation-order noise — this pins which lane supplies which block's scale
    (lane group j of a fragment row = K-block j of the 128-element K-tile) and the plane layout."""
    g = torch.Generator().manual_seed(M + N + K)
    a8 = torch.randn(batch, M, K, generator=g).to(FP8)
    w8 = (torch.randn(N, K, generator=g) * 0.5).to(FP8)
    sw = torch.rand(N, generator=g) * 0.02 + 0.01
    bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
    sc = ops.BlockScales.empty(batch, M, K, gpu)
    sc.t.fill_(127)
    sc.set_rowmajor(torch.randint(107, 148, (batch, M, K // 32), generator=g, dtype=torch.uint8))
    a_deq = ops.dequantize_mx(a8.to(gpu), sc).cpu()
    ref = torch.einsum("bmk,nk->bmn", a_deq.double(), w8.float().double()) * sw.view(1, 1, N).double() + bias.double()
    out = torch.empty(batch, M, N, device=gpu, dtype=torch.float32)
    ops.linear(a8.to(gpu), w8.to(gpu), out, bias=bias.to(gpu), w_scale=sw.to(gpu), a_bscale=sc)
    err = rel_l2(out.cpu(), ref)
    print(f"gemm mx {M}x{N}x{K} b{batch} f32-out rel-L2 {err:.2e}")
    assert err < 2e-5
    out_b = torch.empty_like(out)
    ops.linear(a8.to(gpu), w8.to(gpu), out_b, bias=bias.to(gpu), w_scale=sw.to(gpu), a_bscale=sc)
    assert torch.equal(out, out_b)                                   # bitwise repeatable


def test_gemm_block_scaled_views_of_both_streams(ops, gpu):
    """The double blocks' shape: one e4m3 buffer [B, S, K] with text rows first, its two row ranges as the A operands of one grouped
    launch, scales addressed through BlockScales.rows()."""
    g = torch.Generator().manual_seed(11)
    B, T, Ni, K, N = 2, 192, 500, 768, 512
    S = T + Ni
    a8 = torch.randn(B, S, K, generator=g).to(FP8).to(gpu)
    sc = ops.BlockScales.empty(B, S, K, gpu)
    sc.set_rowmajor(torch.randint(117, 138, (B, S, K // 32), generator=g, dtype=torch.uint8))
    wi, wt = [(torch.randn(N, K, generator=g) * 0.5).to(FP8).to(gpu) for _ in range(2)]
    oi = torch.empty(B, Ni, N, device=gpu, dtype=torch.float32)
    ot = torch.empty(B, T, N, device=gpu, dtype=torch.float32)
    ops.linear_grouped([ops.LinearProblem(a8[:, T:], wi, oi, a_bscale=sc.rows(T)), ops.LinearProblem(a8[:, :T], wt, ot, a_bscale=sc.rows(0))])
    a_deq = ops.dequantize_mx(a8, sc).cpu().double()
    assert rel_l2(oi.cpu(), a_deq[:, T:] @ wi.float().cpu().double().T) < 2e-5
    assert rel_l2(ot.cpu(), a_deq[:, :T] @ wt.float().cpu().double().T) < 2e-5


@pytest.mark.parametrize("M,N,K,c8_from", [(256, 512, 256, 0), (300, 1024, 384, 256), (512, 1536, 512, 512)])
def test_gemm_epilogue_writes_block_scaled_e4m3(ops, gpu, M, N, K, c8_from):
    """Columns >= c8_from leave the epilogue as e4m3 + block scales. The f32-output launch of the same problem computes the same fp32
    epilogue values; quantising those with the oracle's rule must reproduce the kernel's bytes and scale bytes exactly, and the columns
    below c8_from must be the bf16 launch's."""
    g = torch.Generator().manual_seed(M + N)
    batch = 2
    a8 = torch.randn(batch, M, K, generator=g).to(FP8).to(gpu)
    w8 = (torch.randn(N, K, generator=g) * 0.5).to(FP8).to(gpu)
    sa = (torch.rand(batch * M, generator=g) + 0.5).to(gpu)
    sw = (torch.rand(N, generator=g) * 0.02 + 0.01).to(gpu)
    bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(gpu)
    kw = dict(bias=bias, a_scale=sa, w_scale=sw, gelu_from=c8_from)
    f32 = torch.empty(batch, M, N, device=gpu, dtype=torch.float32)
    ops.linear(a8, w8, f32, **kw)
    ref16 = torch.empty(batch, M, N, device=gpu, dtype=torch.bfloat16)
    ops.linear(a8, w8, ref16, **kw)
    N8 = N - c8_from
    big8 = torch.zeros(batch, M, N8 + 256, device=gpu, dtype=FP8)       # the e4m3 output lands in a column slice of a wider buffer
    sc = ops.BlockScales.empty(batch, M, N8 + 256, gpu)
    sc.t.fill_(0)
    out16 = torch.full((batch, M, N), 7.0, device=gpu, dtype=torch.bfloat16)
    ops.linear(a8, w8, out16, out8=big8[..., 256:], out8_scales=sc.cols(256), out8_from=c8_from, **kw)
    q_ref, s_ref = orc.quant_mx_e4m3(f32[..., c8_from:].cpu(), return_parts=True)
    got_s = _scales_of(ops, sc.cols(256), batch, M, N8)
    assert torch.equal(got_s, s_ref)
    assert torch.equal(big8[..., 256:].cpu().view(torch.uint8), q_ref.view(torch.uint8))
    assert torch.equal(out16[..., :c8_from], ref16[..., :c8_from])
    assert bool((out16[..., c8_from:] == 7.0).all())                   # those columns are not written to C
    assert bool((big8[..., :256].view(torch.uint8) == 0).all()) and bool((sc.rowmajor(batch, M, 256) == 0).all())


@pytest.mark.parametrize("B,S,H,T", [(1, 256, 2, 64), (2, 200, 4, 40), (1, 1100, 2, 0)])
def test_attention_fp8_block_scaled_output(ops, gpu, B, S, H, T):
    """rt_attention_fp8_fwd_mx against rt_attention_fp8_fwd on the same operands: the same kernel up to the epilogue, so the
    de-quantised e4m3 output may differ from the bf16 output by the two roundings only (e4m3: 2^-4 of the element or half a
    subnormal step of its block; bf16: 2^-9), and every scale byte must be the rule applied to its block (one step of slack for a
    block maximum that the bf16 rounding moved across a boundary)."""
    from reptext_amd import native

    g = torch.Generator().manual_seed(S + H)
    d = H * 128
    qkv = torch.randn(B, S, 3 * d, generator=g).to(torch.bfloat16).to(gpu)
    w = [(1.0 + 0.1 * torch.randn(128, generator=g)).to(torch.bfloat16).to(gpu) for _ in range(4)]
    ids = torch.cat([torch.zeros(T, 3), orc.latent_image_ids(2 * 10, 2 * ((S - T + 9) // 10))[: S - T]]) if S - T > 0 else torch.zeros(T, 3)
    cos, sin = orc.rope_table(ids)
    qk8 = torch.empty(B, S, 2 * d, device=gpu, dtype=FP8)
    vt8 = torch.empty(int(native.load().rt_attention_fp8_vt_bytes(B, S, H)), device=gpu, dtype=FP8)
    ops.attention_fp8_prep(qkv, 0, d, 2 * d, H, T, w[0], w[1], w[2], w[3], cos.to(gpu), sin.to(gpu), qk8, vt8)
    ref = torch.empty(B, S, d, device=gpu, dtype=torch.bfloat16)
    ops.attention_fp8(qk8, vt8, ref, H)
    wide = 256 + ((d + 255) // 256) * 256
    out8 = torch.zeros(B, S, wide, device=gpu, dtype=FP8)
    sc = ops.BlockScales.empty(B, S, wide, gpu)
    sc.t.fill_(0)
    ops.attention_fp8_mx(qk8, vt8, out8[..., 256:], sc.cols(256), H)
    deq = ops.dequantize_mx(out8[..., 256 : 256 + d], sc.cols(256)).cpu()
    r = ref.float().cpu()
    sb = _scales_of(ops, sc.cols(256), B, S, d).float()
    step = torch.exp2(sb - 127.0 - 10.0).repeat_interleave(32, dim=-1)
    bad = (deq - r).abs() > r.abs() * (2.0 ** -4 + 2.0 ** -8) + step
    assert not bool(bad.any()), int(bad.sum())
    rule = orc.mx_scale_byte(r.reshape(B, S, d // 32, 32).abs().amax(dim=-1)).float()
    assert bool(((sb - rule).abs() <= 1).all()) and float((sb != rule).float().mean()) < 0.02
    err = rel_l2(deq, r)
    print(f"attention fp8 -> mx output B={B} S={S} H={H}: {err:.3e} from the bf16 output")
    assert err < 3.5e-2
    assert bool((out8[..., :256].view(torch.uint8) == 0).all())
    out8b = torch.zeros_like(out8)
    ops.attention_fp8_mx(qk8, vt8, out8b[..., 256:], sc.cols(256), H)
    assert torch.equal(out8.view(torch.uint8), out8b.view(torch.uint8))
