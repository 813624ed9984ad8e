"""Interleaved timing of the ablation builds of attention_v3 (tools/attn_v3_dev/make_ablations.py) in ONE process."""
import ctypes as C, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
d = H * 128
qkv = torch.randn(1, S, 3 * d, device=dev).to(torch.bfloat16)
out = torch.empty(1, S, d, device=dev, dtype=torch.bfloat16)
libs = {}
for n in sorted(os.listdir(here)):
    p = os.path.join(here, n, "librt_reptext_hip.so")
    if n.startswith("lib_") and os.path.isfile(p):
        lib = C.CDLL(p)
        lib.rt_attention_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int64] * 4 + [C.c_int32] * 3 + [C.c_float, C.c_void_p, C.c_int64, C.c_void_p]
        libs[n[4:]] = lib
use_ws = len(sys.argv) > 3 and sys.argv[3] == "ws"
st = torch.cuda.current_stream().cuda_stream
ws = None
if use_ws:
    any_lib = next(iter(libs.values()))
    any_lib.rt_attention_ws_bytes.restype = C.c_int64
    any_lib.rt_attention_ws_bytes.argtypes = [C.c_int32] * 3
    n = any_lib.rt_attention_ws_bytes(1, S, H)
    ws = torch.zeros(max(n, 256), device=dev, dtype=torch.uint8)
    print("workspace bytes", n)
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
def run(lib):
    r = lib.rt_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), qkv.stride(1), qkv.stride(0), out.stride(1), out.stride(0), 1, S, H, 128 ** -0.5, None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), st)
    assert r == 0, r
res = {n: [] for n in libs}
for lib in libs.values():
    lib.rt_attention_variant.argtypes = [C.c_int32]
    lib.rt_attention_variant(2)
for rnd in range(4):
    for n, lib in libs.items():
        if ws is not None:
            ws.zero_()          # ablated builds can leave tickets behind
        for _ in range(2): run(lib)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run(lib)
        e1.record(); torch.cuda.synchronize()
        res[n].append(e0.elapsed_time(e1) / 10 * 1e3)
base = min(res.get("BASE", [0]))
for n, ts in res.items():
    print(f"{n:8s} {min(ts):8.1f} us  ({100 * (min(ts) / base - 1):+.1f} %)", flush=True)
