"""Do two GEMM launches on two HIP streams share the chip? A: 192 tiles, B: 64 tiles, K=12288."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reptext_amd.ops as ops
dev = torch.device("cuda:0")
K = 12288
rb = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
a1, w1 = rb(4096, K), rb(3072, K) * 0.02      # 16 x 12 = 192 tiles
a2, w2 = rb(1024, K), rb(4096, K) * 0.02      # 4 x 16 = 64 tiles
o1 = torch.empty(4096, 3072, device=dev, dtype=torch.bfloat16)
o2 = torch.empty(1024, 4096, device=dev, dtype=torch.bfloat16)
s2 = torch.cuda.Stream()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def seq():
    ops.linear(a1, w1, o1); ops.linear(a2, w2, o2)
def par():
    cur = torch.cuda.current_stream()
    e = torch.cuda.Event(); e.record(cur); s2.wait_event(e)
    with torch.cuda.stream(s2):
        ops.linear(a2, w2, o2)
        j = torch.cuda.Event(); j.record(s2)
    ops.linear(a1, w1, o1)
    cur.wait_event(j)
print(f"A alone {t(lambda: ops.linear(a1, w1, o1)):.1f} us, B alone {t(lambda: ops.linear(a2, w2, o2)):.1f} us, sequential {t(seq):.1f} us, two streams {t(par):.1f} us")
