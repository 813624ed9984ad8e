"""Kernel micro-benchmarks at the C2 shapes (1024², S=4608, d=3072). Run on the GPU box:
    python tools/bench_kernels.py [gemm|attn|all]
Prints TFLOP/s from torch.cuda.Event timing on the current stream (same stream the kernels are enqueued on)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reptext_amd.ops as ops

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench_gemm():
    shapes = [(4608, 21504, 3072), (4608, 3072, 15360), (4096, 9216, 3072), (4096, 12288, 3072), (4096, 3072, 12288),
              (4096, 3072, 3072), (512, 9216, 3072), (512, 12288, 3072), (8192, 8192, 8192)]
    for M, N, K in shapes:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.linear(a, w, out, bias=b))
        print(f"gemm M={M} N={N} K={K}: {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TF/s", flush=True)
    # grouped: image + text stream
    T, Ni, d = 512, 4096, 3072
    x = torch.randn(T + Ni, d, device=dev).to(torch.bfloat16)
    wi = (torch.randn(3 * d, d, device=dev) * 0.02).to(torch.bfloat16)
    wt = (torch.randn(3 * d, d, device=dev) * 0.02).to(torch.bfloat16)
    out = torch.empty(T + Ni, 3 * d, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.linear_grouped([ops.LinearProblem(x[T:], wi, out[T:]), ops.LinearProblem(x[:T], wt, out[:T])]))
    print(f"grouped qkv (4096+512)x9216x3072: {t*1e6:9.1f} us  {2*(T+Ni)*3*d*d/t/1e12:8.1f} TF/s", flush=True)


def bench_gemm_tiles():
    """Interleaved A/B of rt_gemm_bf16's tile modes (0 = 256x256 only, 3 = 288x192 + narrow tails) on the model's launches."""
    from reptext_amd import native
    lib = native.load()
    T, Ni, d = 512, 4096, 3072
    x = torch.randn(T + Ni, 4 * d, device=dev).to(torch.bfloat16)
    x32 = torch.randn(T + Ni, d, device=dev)
    gate = torch.randn(1, d, device=dev)
    launches = []
    w_out = (torch.randn(d, 5 * d, device=dev) * 0.02).to(torch.bfloat16)
    a_out = torch.randn(T + Ni, 5 * d, device=dev).to(torch.bfloat16)
    launches.append(("single out 4608x3072x15360 f32", 2 * (T + Ni) * d * 5 * d, lambda: ops.linear(a_out, w_out, x32, gate=gate, res=x32)))
    w_f = (torch.randn(7 * d, d, device=dev) * 0.02).to(torch.bfloat16)
    o_f = torch.empty(T + Ni, 7 * d, device=dev, dtype=torch.bfloat16)
    launches.append(("single fused 4608x21504x3072", 2 * (T + Ni) * 7 * d * d, lambda: ops.linear(x[:, :d], w_f, o_f, gelu_from=3 * d)))
    for name, N, K, f32 in (("qkv", 3 * d, d, False), ("ff1", 4 * d, d, False), ("ff2", d, 4 * d, True), ("out", d, d, True)):
        wi = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        wt = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
        o = x32 if f32 else torch.empty(T + Ni, N, device=dev, dtype=torch.bfloat16)
        kw = dict(res=None)
        def run(wi=wi, wt=wt, o=o, K=K, f32=f32):
            if f32:
                ops.linear_grouped([ops.LinearProblem(x[T:, :K], wi, o[T:], gate=gate, res=o[T:]), ops.LinearProblem(x[:T, :K], wt, o[:T], gate=gate, res=o[:T])])
            else:
                ops.linear_grouped([ops.LinearProblem(x[T:, :K], wi, o[T:]), ops.LinearProblem(x[:T, :K], wt, o[:T])])
        launches.append((f"double {name} (4096+512)x{N}x{K}", 2 * (T + Ni) * N * K, run))
    prev = lib.rt_gemm_tile_mode(-1)
    for name, fl, run in launches:
        res = {}
        for rnd in range(3):
            for mode in (0, 3):
                lib.rt_gemm_tile_mode(mode)
                res.setdefault(mode, []).append(timeit(run, iters=10, warm=2))
        t0, t3 = min(res[0]), min(res[3])
        print(f"{name:42s} mode0 {t0*1e6:7.1f} us {fl/t0/1e12:7.1f} TF/s | mode3 {t3*1e6:7.1f} us {fl/t3/1e12:7.1f} TF/s | {100*(t3/t0-1):+.1f} %", flush=True)
    lib.rt_gemm_tile_mode(prev)


def bench_gemm_fp8():
    FP8 = torch.float8_e4m3fn
    shapes = [(4608, 21504, 3072), (4096, 9216, 3072), (4096, 12288, 3072), (8192, 8192, 8192), (9728, 21504, 3072)]
    for M, N, K in shapes:
        a = torch.randn(M, K, device=dev).to(FP8)
        w = torch.randn(N, K, device=dev).to(FP8)
        sa = torch.rand(M, device=dev) + 0.5
        sw = torch.rand(N, device=dev) * 0.02
        b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda: ops.linear(a, w, out, bias=b, a_scale=sa, w_scale=sw))
        print(f"gemm fp8 M={M} N={N} K={K}: {t*1e6:9.1f} us  {2*M*N*K/t/1e12:8.1f} TF/s", flush=True)
    x = torch.randn(1, 4608, 3072, device=dev)
    o8 = torch.empty(1, 4608, 3072, device=dev, dtype=FP8)
    rs = torch.empty(4608, device=dev)
    mod = torch.randn(1, 6144, device=dev)
    t = timeit(lambda: ops.layernorm_modulate_fp8(x, o8, rs, mod[:, :3072], mod[:, 3072:]))
    print(f"layernorm_mod_fp8 4608x3072 (f32 in): {t*1e6:8.1f} us", flush=True)


def bench_attn():
    """Interleaved A/B of the two attention kernels (variant 0 = attention.hip, 1 = attention_v3.hip where it applies)."""
    from reptext_amd import native
    lib = native.load()
    prev = lib.rt_attention_variant(-1)
    for B, S, H in [(1, 4608, 24), (1, 4096, 32), (4, 4608, 24), (1, 768, 24), (1, 9728, 24)]:
        d = H * 128
        qkv = torch.randn(B, S, 3 * d, device=dev).to(torch.bfloat16)
        out = torch.empty(B, S, d, device=dev, dtype=torch.bfloat16)
        res = {}
        for rnd in range(3):
            for var in (0, 2):
                lib.rt_attention_variant(var)
                res.setdefault(var, []).append(timeit(lambda: ops.attention(qkv[..., :d], qkv[..., d:2*d], qkv[..., 2*d:], out, H), iters=10, warm=2))
        fl = 4 * B * H * S * S * 128
        t0, t1 = min(res[0]), min(res[2])
        print(f"attn B={B} S={S} H={H}: attention.hip {t0*1e6:8.1f} us {fl/t0/1e12:7.1f} TF/s | v3 {t1*1e6:8.1f} us {fl/t1/1e12:7.1f} TF/s | {100*(t1/t0-1):+.1f} %", flush=True)
    lib.rt_attention_variant(prev)


def bench_attn_fp8():
    from reptext_amd import native
    FP8 = torch.float8_e4m3fn
    for B, S, H in [(1, 4608, 24), (1, 9728, 24)]:
        d = H * 128
        qkv = torch.randn(B, S, 3 * d, device=dev).to(torch.bfloat16)
        w = torch.ones(128, device=dev, dtype=torch.bfloat16)
        cos = torch.ones(S, 128, device=dev); sin = torch.zeros(S, 128, device=dev)
        qk8 = torch.empty(B, S, 2 * d, device=dev, dtype=FP8)
        vt8 = torch.empty(int(native.load().rt_attention_fp8_vt_bytes(B, S, H)), device=dev, dtype=FP8)
        out = torch.empty(B, S, d, device=dev, dtype=torch.bfloat16)
        tp = timeit(lambda: ops.attention_fp8_prep(qkv, 0, d, 2 * d, H, 512, w, w, w, w, cos, sin, qk8, vt8))
        t = timeit(lambda: ops.attention_fp8(qk8, vt8, out, H))
        print(f"attn fp8 B={B} S={S} H={H}: prep {tp*1e6:7.1f} us, kernel {t*1e6:9.1f} us  {4*B*H*S*S*128/t/1e12:8.1f} TF/s", flush=True)


def bench_elem():
    S, d = 4608, 3072
    x = torch.randn(1, S, d, device=dev).to(torch.bfloat16)
    out = torch.empty_like(x)
    mod = torch.randn(1, 2 * d, device=dev)
    t = timeit(lambda: ops.layernorm_modulate(x, out, mod[:, :d], mod[:, d:]))
    print(f"layernorm_mod {S}x{d}: {t*1e6:8.1f} us  {2*S*d*2/t/1e9:8.1f} GB/s", flush=True)
    w = (torch.randn(6 * d, d, device=dev) * 0.02).to(torch.bfloat16)
    temb = torch.randn(1, d, device=dev)
    y = torch.empty(1, 6 * d, device=dev)
    t = timeit(lambda: ops.gemv(temb, w, None, y, silu_in=True))
    print(f"adaLN gemv {6*d}x{d}: {t*1e6:8.1f} us  {6*d*d*2/t/1e9:8.1f} GB/s", flush=True)
    qkv = torch.randn(1, S, 3 * d, device=dev).to(torch.bfloat16)
    wn = torch.ones(128, device=dev, dtype=torch.bfloat16)
    cos = torch.randn(S, 128, device=dev); sin = torch.randn(S, 128, device=dev)
    t = timeit(lambda: ops.qk_rmsnorm_rope(qkv, 0, d, 24, 512, wn, wn, wn, wn, cos, sin))
    print(f"qk_rmsnorm_rope {S}x{2*d}: {t*1e6:8.1f} us  {2*S*2*d*2/t/1e9:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("tiles", "all"):
        bench_gemm_tiles()
    if what in ("fp8", "all"):
        bench_gemm_fp8()
    if what in ("attn", "all"):
        bench_attn()
    if what in ("attn8", "all"):
        bench_attn_fp8()
    if what in ("elem", "all"):
        bench_elem()
