"""Does running the two row-halves of a GEMM chain on two HIP streams fill the tail rounds? FF1->FF2 chain at C2 shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import reptext_amd.ops as ops
dev = torch.device("cuda:0")
d, S = 3072, 4608
g = torch.Generator(device=dev).manual_seed(0)
rb = lambda *s: torch.randn(*s, device=dev, generator=g).to(torch.bfloat16)
xn, x = rb(S, d), rb(S, d)
w1, b1, w2, b2 = rb(4 * d, d) * 0.02, rb(4 * d), rb(d, 4 * d) * 0.02, rb(d)
wq, bq = rb(3 * d, d) * 0.02, rb(3 * d)
ffh = torch.empty(S, 4 * d, device=dev, dtype=torch.bfloat16)
qkv = torch.empty(S, 3 * d, device=dev, dtype=torch.bfloat16)
gate = torch.randn(1, d, device=dev, generator=g)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def chain(rows):
    ops.linear(xn[rows], w1, ffh[rows], bias=b1, gelu_from=0)
    ops.linear(ffh[rows], w2, x[rows], bias=b2, gate=gate, res=x[rows])
    ops.linear(xn[rows], wq, qkv[rows], bias=bq)

def one():
    chain(slice(0, S))

def two(split):
    cur = torch.cuda.current_stream()
    e0 = torch.cuda.Event(); e0.record(cur)
    evs = []
    for st, rows in ((s1, slice(0, split)), (s2, slice(split, S))):
        st.wait_event(e0)
        with torch.cuda.stream(st):
            chain(rows)
            e = torch.cuda.Event(); e.record(st); evs.append(e)
    for e in evs:
        cur.wait_event(e)

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

print(f"one stream (FF1+FF2+QKV, M=4608): {timeit(one):8.1f} us")
for split in (2304, 2048, 2560):
    print(f"two streams split at {split}:        {timeit(lambda: two(split)):8.1f} us")
