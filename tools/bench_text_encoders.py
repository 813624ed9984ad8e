"""Per-prompt latency of the HIP prompt encoders at FLUX sizes (T5-XXL: 24 layers, d 4096, 64 heads, ff 10240, 512 tokens;
CLIP-L: 12 layers, d 768, 77 tokens), random weights."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from reptext_amd.text_encoders import CLIPTextModel, T5EncoderModel
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
t5 = T5EncoderModel(device=dev, dtype=torch.bfloat16)
clip = CLIPTextModel(eos_token_id=49407, device=dev, dtype=torch.bfloat16)
for m in (t5, clip):
    for p in m.parameters():
        p.data.copy_(0.02 * torch.randn(p.shape, device=dev, generator=g))
ids5 = torch.randint(0, 32000, (1, 512), device=dev)
idsc = torch.randint(0, 49000, (1, 77), device=dev); idsc[0, 20] = 49407
for name, m, ids in (("T5-XXL encoder, 512 tokens", t5, ids5), ("CLIP-L text, 77 tokens", clip, idsc)):
    m(ids); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        o = m(ids)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per prompt, finite={bool(torch.isfinite(o[0].float()).all())}", flush=True)
