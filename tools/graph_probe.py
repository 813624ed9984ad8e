"""Can the whole denoise loop be captured in ONE hipGraph (torch.cuda.graph around pipe._denoise) and what does replay cost?
Every kernel is enqueued through ctypes on torch's current stream, so capture sees them; host scalars (timesteps, dt) are baked.
    python tools/graph_probe.py [--depth-scale 1.0]"""
import sys, os, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from reptext_amd.config import flux_dev_transformer_config, reptext_controlnet_config
from reptext_amd.controlnet import FluxControlNetModel
from reptext_amd.pipeline import FluxControlNetPipeline, retrieve_timesteps
from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler, calculate_shift
from reptext_amd.transformer import FluxTransformer2DModel

ap = argparse.ArgumentParser()
ap.add_argument("--depth-scale", type=float, default=1.0)
ap.add_argument("--steps", type=int, default=28)
args = ap.parse_args()
dev = torch.device("cuda:0")
bf16 = torch.bfloat16
cfg_t, cfg_c = flux_dev_transformer_config(), reptext_controlnet_config()
if args.depth_scale != 1.0:
    cfg_t["num_layers"] = max(1, int(cfg_t["num_layers"] * args.depth_scale))
    cfg_t["num_single_layers"] = max(1, int(cfg_t["num_single_layers"] * args.depth_scale))
tr = FluxTransformer2DModel(**cfg_t, device=dev, dtype=bf16).random_init_(seed=0)
cn = FluxControlNetModel(**cfg_c, device=dev, dtype=bf16).random_init_(seed=1)
pipe = FluxControlNetPipeline(scheduler=FlowMatchEulerDiscreteScheduler(), vae=None, text_encoder=None, tokenizer=None, text_encoder_2=None,
                              tokenizer_2=None, transformer=tr, controlnet=cn)
pipe.set_progress_bar_config(disable=True)
pipe.capture_graphs = False          # this probe captures by hand; the product path (pipeline.GRAPH_CAPTURE) does the same per call signature
H = W = 1024
N = (H // 16) * (W // 16)
g = torch.Generator().manual_seed(1)
pe = torch.randn(1, 512, 4096, generator=g).to(dev, bf16)
pooled = torch.randn(1, 768, generator=g).to(dev, bf16)
hints = [torch.randn(1, N, 128, generator=g).to(dev, bf16)]
masks = [torch.rand(1, N, 1, generator=g).to(dev)]
lat0 = torch.randn(1, N, 64, generator=g).to(dev, bf16)
sc = pipe.scheduler.config
mu = calculate_shift(N, sc.base_image_seq_len, sc.max_image_seq_len, sc.base_shift, sc.max_shift)
text_ids = torch.zeros(512, 3, device=dev, dtype=bf16)
image_ids = pipe._prepare_latent_image_ids(1, 2 * (H // 16), 2 * (W // 16), dev, bf16)
pipe._guidance_scale, pipe._joint_attention_kwargs, pipe._interrupt = 3.5, None, False


def run(lat):
    timesteps, n = retrieve_timesteps(pipe.scheduler, args.steps, "cpu", None, np.linspace(1.0, 1 / args.steps, args.steps), mu=mu)   # host tensor: no sync in the loop
    return pipe._denoise(lat, pe, pooled, text_ids, image_ids, timesteps, hints, masks, 3.5, 1.0, args.steps, None, None, [], n)


def timed(fn, n=3):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


ref = run(lat0.clone()); run(lat0.clone())
print(f"memory reserved before capture: {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
t_eager = timed(lambda: run(lat0.clone()))
print(f"eager: {t_eager*1e3:.1f} ms per loop", flush=True)
static_in = lat0.clone()
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
t0 = time.perf_counter()
with torch.cuda.stream(s):
    run(static_in.clone())                                   # warm on the capture stream (workspaces keyed by stream)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=s):
        out = run(static_in)
torch.cuda.synchronize()
print(f"memory reserved after capture: {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
print(f"capture + instantiate: {time.perf_counter()-t0:.1f} s", flush=True)
static_in.copy_(lat0)
graph.replay(); torch.cuda.synchronize()
print("replay == eager bitwise:", bool(torch.equal(out, ref)), flush=True)
def rep():
    static_in.copy_(lat0); graph.replay()
t_graph = timed(rep)
t0 = time.perf_counter(); static_in.copy_(lat0); graph.replay(); t_host = time.perf_counter() - t0; torch.cuda.synchronize()
print(f"graph replay: {t_graph*1e3:.1f} ms per loop (host returns after {t_host*1e3:.1f} ms); eager {t_eager*1e3:.1f} ms", flush=True)
