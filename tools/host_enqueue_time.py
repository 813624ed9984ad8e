"""How far ahead of the GPU is the host? Time to ENQUEUE one image's kernels (no sync) vs the time the GPU needs to run them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from reptext_amd.config import flux_dev_transformer_config, flux_vae_config, reptext_controlnet_config
from reptext_amd.controlnet import FluxControlNetModel
from reptext_amd.pipeline import FluxControlNetPipeline
from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
from reptext_amd.transformer import FluxTransformer2DModel
from reptext_amd.vae import AutoencoderKL

dev, bf16 = torch.device("cuda:0"), torch.bfloat16
tr = FluxTransformer2DModel(**flux_dev_transformer_config(), device=dev, dtype=bf16).random_init_(seed=0)
cn = FluxControlNetModel(**reptext_controlnet_config(), device=dev, dtype=bf16).random_init_(seed=1)
vae = AutoencoderKL(**flux_vae_config(), device=dev, dtype=bf16).random_init_(seed=2)
pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), vae, None, None, None, None, tr, cn)
pipe.set_progress_bar_config(disable=True)
g = torch.Generator().manual_seed(1)
pe, pooled = torch.randn(1, 512, 4096, generator=g).to(dev, bf16), torch.randn(1, 768, generator=g).to(dev, bf16)
hints = [torch.randn(1, 4096, 128, generator=g).to(dev, bf16)]
masks = [torch.ones(4096, device=dev)]
for prec in ("bf16", "fp8"):
    if prec == "fp8":
        for m in (tr, cn):
            m.enable_fp8_linears("all").enable_fp8_attention(True)
    for it in range(2):
        lat = pipe._pack_latents(torch.randn(1, 16, 128, 128, generator=g).to(dev, bf16), 1, 16, 128, 128)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        B.run_image(pipe, lat, pe, pooled, hints, masks, 1024, 1024, 28)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{prec}: host enqueue {t1 - t0:.3f} s, GPU done after {t2 - t0:.3f} s", flush=True)

# ---- cost of the host alone: enqueue a short burst into an EMPTY queue (no back-pressure from a full HIP queue)
import reptext_amd.ops as ops
x = torch.randn(4608, 3072, device=dev).to(bf16)
w = (torch.randn(3072, 3072, device=dev) * 0.02).to(bf16)
o = torch.empty(4608, 3072, device=dev, dtype=bf16)
mod = torch.randn(1, 6144, device=dev)
xf = torch.randn(1, 4608, 3072, device=dev)
xo = torch.empty(1, 4608, 3072, device=dev, dtype=bf16)
for name, fn, n in (("linear (ctypes struct + launch)", lambda: ops.linear(x, w, o), 200),
                    ("layernorm_modulate", lambda: ops.layernorm_modulate(xf, xo, mod[:, :3072], mod[:, 3072:]), 200)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"host cost of one ops.{name}: {(t1 - t0) / n * 1e6:.1f} us", flush=True)
