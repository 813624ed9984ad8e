"""Does a GEMM whose weights come from HBM run slower than one whose weights sit in the 256 MB Infinity Cache?
Times single launches (events around each launch) (a) back to back on the same operands (warm), (b) after a 1 GiB fill that evicts the
memory-side cache (cold), (c) cold, then the weights touched by a streaming read (torch sum) just before the launch (prefetched)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reptext_amd.ops as ops

dev = torch.device("cuda:0")
junk = torch.empty(1 << 30, device=dev, dtype=torch.uint8)


def one(fn, pre=None, n=12):
    ts = []
    for _ in range(n):
        if pre is not None:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for M, N, K in [(4608, 21504, 3072), (4096, 12288, 3072), (4608, 3072, 15360), (4096, 9216, 3072), (4096, 3072, 12288)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.linear(a, w, out, bias=b)
    fl = 2 * M * N * K / 1e6
    evict = lambda: junk.fill_(1)
    def evict_prefetch():
        junk.fill_(1)
        w.view(torch.int32).sum(); a.view(torch.int32).sum()
    def evict_prefetch_w():
        junk.fill_(1)
        w.view(torch.int32).sum()
    tw, tc, tp, tpw = one(fn), one(fn, evict), one(fn, evict_prefetch), one(fn, evict_prefetch_w)
    print(f"{M}x{N}x{K}: warm {tw:7.1f} us {fl/tw:7.1f} TF/s | cold {tc:7.1f} us {fl/tc:7.1f} | cold+prefetch(W,A) {tp:7.1f} us {fl/tp:7.1f} | cold+prefetch(W) {tpw:7.1f} {fl/tpw:7.1f}", flush=True)
