"""Timing of the block-scaled GEMM against the per-row-scaled one at the model's shapes (alternating, one process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import reptext_amd.native as native
if len(sys.argv) > 1:                     # a variant library of tools/mx_dev/make_ablations.py
    native.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib_" + sys.argv[1], "librt_reptext_hip.so")
    print("library:", native.LIB_PATH)
import reptext_amd.ops as ops

gpu = torch.device("cuda", 0)
FP8 = torch.float8_e4m3fn


def bench(M, N, K, mx, out_dtype=torch.float32, iters=30):
    g = torch.Generator().manual_seed(0)
    a8 = torch.randn(1, M, K, generator=g).to(FP8).to(gpu)
    w8 = (torch.randn(N, K, generator=g) * 0.5).to(FP8).to(gpu)
    sw = (torch.rand(N, generator=g) * 0.02 + 0.01).to(gpu)
    out = torch.empty(1, M, N, device=gpu, dtype=out_dtype)
    kw = dict(w_scale=sw)
    if mx:
        sc = ops.BlockScales.empty(1, M, K, gpu)
        sc.t.copy_(torch.randint(120, 134, tuple(sc.t.shape), generator=g, dtype=torch.uint8))           # any layout: timing only
        kw["a_bscale"] = sc
    else:
        kw["a_scale"] = (torch.rand(M, generator=g) + 0.5).to(gpu)
    for _ in range(5):
        ops.linear(a8, w8, out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        ops.linear(a8, w8, out, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, N, K) in [(4608, 3072, 15360), (4608, 3072, 12288), (4608, 3072, 3072), (9728, 3072, 15360)]:
    r = []
    for rep in range(3):
        r.append((bench(M, N, K, False), bench(M, N, K, True)))
    a, b = min(x[0] for x in r), min(x[1] for x in r)
    print(f"{M}x{N}x{K}: per-row {a:7.1f} us   block-scaled {b:7.1f} us   {100 * (b / a - 1):+.1f} %   ({2 * M * N * K / b / 1e6:.0f} TFLOP/s)", flush=True)


def bench_out8(M, N, K, c8_from, mx_out, iters=30):
    """The producing side: GELU columns >= c8_from as bf16 (mx_out False) or as e4m3 + block scales out of the epilogue."""
    g = torch.Generator().manual_seed(0)
    a8 = torch.randn(1, M, K, generator=g).to(FP8).to(gpu)
    w8 = (torch.randn(N, K, generator=g) * 0.5).to(FP8).to(gpu)
    sw = (torch.rand(N, generator=g) * 0.02 + 0.01).to(gpu)
    sa = (torch.rand(M, generator=g) + 0.5).to(gpu)
    bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16).to(gpu)
    out = torch.empty(1, M, N, device=gpu, dtype=torch.bfloat16)
    kw = dict(w_scale=sw, a_scale=sa, bias=bias, gelu_from=c8_from)
    if mx_out:
        kw.update(out8=torch.empty(1, M, N - c8_from, device=gpu, dtype=FP8), out8_scales=ops.BlockScales.empty(1, M, N - c8_from, gpu), out8_from=c8_from)
    for _ in range(5):
        ops.linear(a8, w8, out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        ops.linear(a8, w8, out, **kw)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, N, K, c0) in [(4608, 21504, 3072, 9216), (4608, 12288, 3072, 0)]:
    r = [(bench_out8(M, N, K, c0, False), bench_out8(M, N, K, c0, True)) for _ in range(3)]
    a, b = min(x[0] for x in r), min(x[1] for x in r)
    print(f"{M}x{N}x{K} gelu from {c0}: bf16 out {a:7.1f} us   e4m3+scales out {b:7.1f} us   {100 * (b / a - 1):+.1f} %", flush=True)
