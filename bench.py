#!/usr/bin/env python3
"""Headline benchmark: FLUX.1-dev + RepText ControlNet, 1024x1024, 28 steps, bf16, synthetic Arabic-glyph hints.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one full pass of the hot path over one batch: 28 denoising steps (ControlNet tower + transformer + Euler
step each) followed by the VAE decode to uint8, for `--batch-per-gpu` images on every rank (BASELINE.json configs[1];
SURVEY.md §8d timed region). Inputs (prompt embeddings, packed hint latents, regional mask, initial noise) are resident
in HBM before the timed region. Weights are random-init FLUX.1-dev / RepText shapes (no checkpoints offline).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed passes (images per GPU x batch)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=1, help="weak scaling (default): images per GPU per pass, fixed as N grows")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: TOTAL images per pass, fixed as N grows (BASELINE config 3: 32 over 8 GPUs); each rank runs its "
                         "shard_range share; overrides --batch-per-gpu")
    ap.add_argument("--shared-prompt", action="store_true",
                    help="one prompt / hint / mask for the whole batch (broadcast once, expanded per rank) instead of one per image")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--inference-steps", type=int, default=28)
    ap.add_argument("--text-lines", type=int, default=1)
    ap.add_argument("--precision", choices=["bf16", "fp8-ln", "fp8", "fp8-mx"], default="bf16",
                    help="bf16 = BASELINE config 2 (the headline); config 5's 'fp8 weights': fp8-ln = LayerNorm-fed projections on the e4m3 "
                         "MFMA path, fp8 = every projection of the blocks (per-row scales, quantise passes) + e4m3 attention, fp8-mx = the same with "
                         "MX block-scaled operands written by the attention / GELU epilogues (no quantise passes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline-pass", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every kernel of the loop from the host each pass instead of replaying the captured hipGraph")
    ap.add_argument("--depth-scale", type=float, default=1.0, help="DEBUG ONLY: scale layer counts (result flagged invalid)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------- work model
def tower_blocks_read(cfg_t, cfg_c):
    """Double blocks of the tower whose samples the transformer consumes: block i of the transformer reads sample
    i // ceil(n_t / n_c) — 5 of RepText's 6 against FLUX's 19 (quirk Q5). The pipeline does not evaluate the rest."""
    n_t, n_c = cfg_t["num_layers"], cfg_c["num_layers"]
    if n_c == 0 or cfg_c["num_single_layers"] > 0:
        return n_c
    return min(n_c, (n_t - 1) // math.ceil(n_t / n_c) + 1)


def flops_per_image(H, W, steps, lines, cfg_t, cfg_c, tower_blocks=None):
    """SURVEY §8d work model. ``tower_blocks``: double blocks of the tower actually evaluated (None = all, the reference's
    count); the roofline fractions use the EXECUTED count so that skipped dead work does not inflate them."""
    d = cfg_t["num_attention_heads"] * cfg_t["attention_head_dim"]
    T, N = 512, (H // 16) * (W // 16)
    S = T + N
    block = 24 * S * d * d + 4 * S * S * d
    tr = (cfg_t["num_layers"] + cfg_t["num_single_layers"]) * block + 2 * N * 64 * d + 2 * T * 4096 * d + 2 * N * d * 64
    L = (cfg_c["num_layers"] if tower_blocks is None else tower_blocks) + cfg_c["num_single_layers"]
    cn = L * block + L * 2 * N * d * d + 2 * N * 64 * d + 2 * N * (64 + cfg_c["extra_condition_channels"]) * d + 2 * T * 4096 * d
    return steps * (tr + lines * cn)


def synthetic_glyph_hint(height, width, text="مرحبا", shift=0):
    """PIL-rendered Arabic glyph on black + bbox position/region masks (host, once). Falls back to a plain box when no
    font with Arabic coverage is installed (the hint only sets mask geometry for the benchmark)."""
    import numpy as np
    from PIL import Image, ImageDraw, ImageFont

    img = Image.new("RGB", (width, height), (0, 0, 0))
    draw = ImageDraw.Draw(img)
    pos = (int(width * (0.10 + 0.08 * (shift % 7))), int(height * (0.10 + 0.11 * ((shift // 7) % 7))))
    try:
        font = ImageFont.truetype("DejaVuSans.ttf", max(height // 13, 12))
        draw.text(pos, text, font=font, fill=(255, 255, 255))
        bbox = draw.textbbox(pos, text, font=font)
    except Exception:
        bbox = (pos[0], pos[1], pos[0] + width // 4, pos[1] + height // 10)
        draw.rectangle(bbox, fill=(255, 255, 255))
    mask = np.zeros([height, width], dtype=np.uint8)
    mask[max(bbox[1] - 5, 0) : bbox[3] + 5, max(bbox[0] - 5, 0) : bbox[2] + 5] = 255
    return img, bbox, Image.fromarray(mask)


class GemmTimer:
    """HIP-event timing of every rt_gemm_bf16 launch on the stream it is enqueued on (roofline pass only)."""

    def __init__(self, ops_mod):
        self.ops = ops_mod
        self.events, self.flops, self.fp8, self.shapes = [], [], [], []
        self.att_events, self.att_flops = [], []
        self._orig = self._orig_att = self._orig_att8 = self._orig_att8mx = None

    def __enter__(self):
        ops = self.ops
        self._orig = ops.linear_grouped

        def timed(problems):
            fl, shp = 0, []
            for p in problems:
                a, w = p.a, p.w
                rows = a.shape[0] * a.shape[1] if a.dim() == 3 else a.shape[0]
                fl += 2 * rows * w.shape[0] * w.shape[1]
                shp.append(f"{rows}x{w.shape[0]}x{w.shape[1]}" + ("" if p.out.dtype == torch.bfloat16 else ":f32out"))
            self.shapes.append(" + ".join(shp))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()          # records on torch's current stream == the stream ops.py launches on
            self._orig(problems)
            e1.record()
            self.events.append((e0, e1))
            self.flops.append(fl)
            self.fp8.append(bool(problems[0].is_fp8))

        ops.linear_grouped = timed
        # the second MFMA kernel of the path, timed the same way (reported beside the roofline of the dominant one)
        self._orig_att, self._orig_att8, self._orig_att8mx = ops.attention, ops.attention_fp8, ops.attention_fp8_mx

        def timed_att(q, k, v, out, H, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self._orig_att(q, k, v, out, H, *a, **kw)
            e1.record()
            self.att_events.append((e0, e1))
            self.att_flops.append(4 * q.shape[0] * H * q.shape[1] * q.shape[1] * 128)
            return r

        def timed_att8(qk8, vt8, out, H, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self._orig_att8(qk8, vt8, out, H, *a, **kw)
            e1.record()
            self.att_events.append((e0, e1))
            self.att_flops.append(4 * qk8.shape[0] * H * qk8.shape[1] * qk8.shape[1] * 128)
            return r

        def timed_att8mx(qk8, vt8, out8, scales, H, *a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self._orig_att8mx(qk8, vt8, out8, scales, H, *a, **kw)
            e1.record()
            self.att_events.append((e0, e1))
            self.att_flops.append(4 * qk8.shape[0] * H * qk8.shape[1] * qk8.shape[1] * 128)
            return r

        ops.attention, ops.attention_fp8, ops.attention_fp8_mx = timed_att, timed_att8, timed_att8mx
        return self

    def __exit__(self, *a):
        self.ops.linear_grouped = self._orig
        self.ops.attention, self.ops.attention_fp8, self.ops.attention_fp8_mx = self._orig_att, self._orig_att8, self._orig_att8mx

    def attention_result(self):
        torch.cuda.synchronize()
        tot_ms = sum(e0.elapsed_time(e1) for e0, e1 in self.att_events)
        return len(self.att_events), sum(self.att_flops), tot_ms * 1e-3

    def per_shape(self, fp8=None, peak_tflops=2500.0):
        """One row per distinct launch shape (M x N x K of every problem of the launch): launches, mean HIP-event duration, TFLOP/s,
        share of all GEMM time — which launches pull the mean down (the evidence was collected, only the mean reported)."""
        torch.cuda.synchronize()
        rows, tot = {}, 0.0
        for i, (e0, e1) in enumerate(self.events):
            if fp8 is not None and self.fp8[i] != fp8:
                continue
            ms = e0.elapsed_time(e1)
            r = rows.setdefault(self.shapes[i], [0, 0.0, 0])
            r[0] += 1; r[1] += ms; r[2] += self.flops[i]
            tot += ms
        out = []
        for shp, (n, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            out.append({"shape_MxNxK": shp, "launches": n, "avg_us": round(ms / n * 1e3, 1), "tflops": round(tf, 1),
                        "frac_of_peak": round(tf / peak_tflops, 4), "share_of_gemm_time": round(ms / tot, 4) if tot else None})
        return out

    def result(self, fp8=None):
        """(launches, flops, seconds) of all GEMM launches, or only the e4m3 (fp8=True) / bf16 (fp8=False) ones."""
        torch.cuda.synchronize()
        sel = [i for i in range(len(self.events)) if fp8 is None or self.fp8[i] == fp8]
        tot_ms = sum(self.events[i][0].elapsed_time(self.events[i][1]) for i in sel)
        return len(sel), sum(self.flops[i] for i in sel), tot_ms * 1e-3


def pmc_traffic_bytes(kernel_key: str):
    """HBM-side bytes per launch of a kernel from the committed PMC passes (profiles/r03_traffic_pmc.json, else the last earlier round's: rocprofv3 --pmc
    FETCH_SIZE and --pmc WRITE_SIZE over this same command, FETCH_SIZE doubled as the MI355X guide prescribes for gfx950).
    Counters cannot be read from inside the timed process, so the value is the last profiled one; null when absent."""
    try:
        path = next(p for p in (os.path.join(ROOT, "profiles", f"r0{r}_traffic_pmc.json") for r in (3, 2, 1)) if os.path.isfile(p))
        with open(path) as f:
            d = json.load(f)
        if kernel_key.startswith("fp8:"):          # the passes over `bench.py --precision fp8`
            return int(d["fp8_run"]["kernels"][kernel_key[4:]]["hbm_bytes_per_launch_corrected"])
        if kernel_key.startswith("fp8mx:"):        # ... --precision fp8-mx
            return int(d["fp8mx_run"]["kernels"][kernel_key[6:]]["hbm_bytes_per_launch_corrected"])
        return int(d["kernels"][kernel_key]["hbm_bytes_per_launch_corrected"])
    except Exception:
        return None


def usable_cores() -> int:
    """Cores this process may actually run on: affinity mask capped by the cgroup CPU quota (the GPU box exposes 256
    logical CPUs but grants a 16-CPU share; running 256 threads there oversubscribes 16x)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("RT_CPU_BASELINE_THREADS", "16"))))


def cpu_model_name() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown CPU"


def cpu_baseline(cfg_t, H, W, steps, lines, cfg_c, budget_s=12.0, pipe=None):
    """The CPU path beside the GPU number: the fp32 oracle (stock torch CPU ops, kind "port") on the box's host
    cores, on a bounded sample:
      (1) BASELINE config 1 RUN IN FULL through oracle.denoise_loop — 256x256, 2 steps, T = 512, one masked text line, FLUX-dev
          shaped weights at FULL depth (19+38 transformer, 6+0 tower, d = 3072). The 14 B weights are the GPU models' own
          (random-init, bf16) and are streamed to the oracle one tensor at a time (oracle/streamed.py: 56 GB of fp32 never
          exist on the host). The SAME case then runs through the GPU pipeline and the latents are compared: the bench line
          carries a full-depth parity number next to the timing;
      (2) whole MMDiT blocks at the C2 sequence length (S = 4608), extrapolated by block count to one C2 image — this is
          `value`, in the metric's unit (images/sec)."""
    from oracle import flux_oracle as orc
    from oracle.streamed import StreamedParams, config1_case, config1_gpu, config1_oracle

    cores = usable_cores()
    torch.set_num_threads(cores)
    d = cfg_t["num_attention_heads"] * cfg_t["attention_head_dim"]
    g = torch.Generator().manual_seed(0)
    T = 512
    # ---- (1) config 1 end to end at full depth
    c1 = None
    if pipe is not None:
        case = config1_case()
        dev = next(pipe.transformer.parameters()).device
        gpu_lat = config1_gpu(pipe, case, dev).float().cpu()
        tp, cp = StreamedParams(pipe.transformer.state_dict()), StreamedParams(pipe.controlnet.state_dict())
        t0 = time.perf_counter()
        ref = config1_oracle(tp, cfg_t, cp, cfg_c, case)
        c1_s = time.perf_counter() - t0
        rel = float((gpu_lat.double() - ref.double()).norm() / ref.double().norm())
        c1 = {"oracle_s": round(c1_s, 1), "depth": f"{cfg_t['num_layers']}+{cfg_t['num_single_layers']} / tower {cfg_c['num_layers']}+{cfg_c['num_single_layers']}",
              "gpu_latents_rel_l2_vs_oracle": float(f"{rel:.3e}"), "weights_streamed_GB": round((tp.bytes_streamed + cp.bytes_streamed) / 1e9, 1),
              "note": "bf16-storage floor of this graph: tests/test_configs_gpu.py::test_c1_full_depth_19_38_tower_6_0_against_oracle"}
        del tp, cp, ref
    # ---- (2) C2 blocks
    N = (H // 16) * (W // 16)
    small = dict(cfg_t, num_layers=1, num_single_layers=1)
    p = orc.init_mmdit_params(small, seed=0, round_bf16=False)
    h, e = torch.randn(1, N, d, generator=g), torch.randn(1, T, d, generator=g)
    temb = torch.randn(1, d, generator=g)
    rope = orc.rope_table(torch.cat([torch.zeros(T, 3), orc.latent_image_ids(2 * (H // 16), 2 * (W // 16))]))
    x = torch.cat([e, h], dim=1)
    H_, Dh = cfg_t["num_attention_heads"], cfg_t["attention_head_dim"]
    td, ts, nd, ns = 0.0, 0.0, 0, 0
    t_start = time.perf_counter()
    with torch.no_grad():
        while time.perf_counter() - t_start < budget_s or nd == 0:
            t0 = time.perf_counter(); orc.double_block(p, "transformer_blocks.0", h, e, temb, rope, H_, Dh); td += time.perf_counter() - t0; nd += 1
            t0 = time.perf_counter(); orc.single_block(p, "single_transformer_blocks.0", x, temb, rope, H_, Dh); ts += time.perf_counter() - t0; ns += 1
    td, ts = td / nd, ts / ns
    n_double = cfg_t["num_layers"] + lines * cfg_c["num_layers"]
    n_single = cfg_t["num_single_layers"] + lines * cfg_c["num_single_layers"]
    sec_per_image = steps * (n_double * td + n_single * ts)
    c1_txt = "(1) skipped (no pipeline given). " if c1 is None else (
        f"(1) BASELINE config 1 run in full through oracle.denoise_loop: 256x256, 2 steps, S=768, d={d}, one masked text line, FULL depth "
        f"{c1['depth']}, weights streamed from the GPU models: {c1['oracle_s']:.1f} s; the same case on the GPU: latents rel-L2 "
        f"{c1['gpu_latents_rel_l2_vs_oracle']:.2e} vs this run. ")
    return {"value": 1.0 / sec_per_image, "unit": "images/sec", "cores": cores, "kind": "port", "cpu": cpu_model_name(),
            "config1_full_depth": c1,
            "sample": f"oracle fp32 torch-CPU on {cores} threads. " + c1_txt +
                      f"(2) value: {nd} double + {ns} single MMDiT blocks at the C2 sequence S={T+N} ({td:.2f}s / {ts:.2f}s each), "
                      f"extrapolated to {steps} steps x ({n_double} double + {n_single} single) blocks (the reference evaluates all "
                      f"{cfg_c['num_layers']} tower blocks); embedders, zero-linears and VAE decode not included"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("RT_DIST_BACKEND", "nccl")      # "gloo": rehearsal of the N>1 path on a box with fewer GPUs than ranks
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPUs: one process per GPU is required for RCCL")
    dev = torch.device("cuda", local_rank % ndev)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            # RCCL over xGMI is the path. If the communicator cannot be created the run FAILS (non-zero exit): a silent gloo
            # fallback would publish a number for a different transport. RT_DIST_BACKEND=gloo asks for the host transport
            # explicitly (rehearsals on a box with fewer GPUs than ranks, CPU tests).
            try:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                probe = torch.zeros(1, device=dev)
                dist.all_reduce(probe)                     # forces communicator creation now, not inside the timed region
                torch.cuda.synchronize()
            except Exception as e:
                print(f"[bench] RCCL init failed on rank {rank} ({type(e).__name__}: {e}). Not falling back: set RT_DIST_BACKEND=gloo "
                      "to run the broadcast over host memory on purpose.", file=sys.stderr, flush=True)
                raise SystemExit(3)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import reptext_amd.ops as ops
    from reptext_amd import dist as rdist
    from reptext_amd.config import flux_dev_transformer_config, flux_vae_config, reptext_controlnet_config
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel
    from reptext_amd.vae import AutoencoderKL

    cfg_t, cfg_c = flux_dev_transformer_config(), reptext_controlnet_config()
    if args.depth_scale != 1.0:
        cfg_t["num_layers"] = max(1, int(cfg_t["num_layers"] * args.depth_scale))
        cfg_t["num_single_layers"] = max(1, int(cfg_t["num_single_layers"] * args.depth_scale))
        cfg_c["num_layers"] = max(1, int(cfg_c["num_layers"] * args.depth_scale))
    bf16 = torch.bfloat16
    tkw = {k: v for k, v in cfg_t.items()}
    transformer = FluxTransformer2DModel(**tkw, device=dev, dtype=bf16).random_init_(seed=0)
    controlnet = FluxControlNetModel(**cfg_c, device=dev, dtype=bf16).random_init_(seed=1)   # zero-linears random too
    if args.precision != "bf16":
        level = {"fp8-ln": "ln", "fp8": "all", "fp8-mx": "mx"}[args.precision]
        transformer.enable_fp8_linears(level)
        controlnet.enable_fp8_linears(level)
        if args.precision in ("fp8", "fp8-mx"):
            transformer.enable_fp8_attention(True)
            controlnet.enable_fp8_attention(True)
    vae = AutoencoderKL(**flux_vae_config(), device=dev, dtype=bf16).random_init_(seed=2)
    pipe = FluxControlNetPipeline(scheduler=FlowMatchEulerDiscreteScheduler(), vae=vae, text_encoder=None, tokenizer=None,
                                  text_encoder_2=None, tokenizer_2=None, transformer=transformer, controlnet=controlnet)
    pipe.set_progress_bar_config(disable=True)

    H, W = args.height, args.width
    strong = args.global_batch > 0
    G = args.global_batch if strong else world * args.batch_per_gpu           # images per pass over all ranks
    if G < world:
        raise SystemExit(f"--global-batch {G} < {world} ranks: every rank needs at least one image")
    lo, hi = rdist.shard_range(G, rank, world)
    Bl = hi - lo                                                              # this rank's share
    N = (H // 16) * (W // 16)
    Gc = 1 if args.shared_prompt else G                                       # leading dim of the conditioning on the wire
    # ---- conditioning: rank 0 builds it for the WHOLE batch (one prompt, hint and regional mask per image unless
    #      --shared-prompt), ONE broadcast, every rank keeps its rows; text encoders are outside the timed region
    spec = [("prompt_embeds", (Gc, 512, 4096)), ("pooled", (Gc, 768))] + [(f"hint{i}", (Gc, N, 128)) for i in range(args.text_lines)] + \
           [(f"mask{i}", (Gc, N)) for i in range(args.text_lines)]
    cond = None
    if rank == 0:
        g = torch.Generator().manual_seed(1)
        pe = torch.randn(Gc, 512, 4096, generator=g)
        pooled = torch.randn(Gc, 768, generator=g)
        g2 = torch.Generator().manual_seed(2)
        hints = [torch.randn(Gc, N, 128, generator=g2) for _ in range(args.text_lines)]
        masks = []
        for _ in range(args.text_lines):
            per = []
            for s_i in range(Gc):                                               # a glyph box per image: same size, shifted position
                _, _, mask_img = synthetic_glyph_hint(H, W, shift=s_i)
                per.append(pipe._region_masks([mask_img], "cpu", torch.float32)[0].reshape(-1))
            masks.append(torch.stack(per))
        cond = rdist.Conditioning(pe.to(bf16).float(), pooled.to(bf16).float(), [h.to(bf16).float() for h in hints], masks)
    if world > 1:
        cond = rdist.broadcast_conditioning(cond, spec, dev, staging_device=("cpu" if backend == "gloo" else None))
    else:
        cond = rdist.Conditioning(cond.prompt_embeds.to(dev, bf16), cond.pooled.to(dev, bf16), [h.to(dev, bf16) for h in cond.hints],
                                  [m.to(dev) for m in cond.masks])
    mine = cond.shard(lo, hi)
    pe = mine.prompt_embeds.contiguous()
    pooled = mine.pooled.contiguous()
    hints = [h.contiguous() for h in mine.hints]
    masks_l = [m.contiguous() for m in mine.masks]                             # per line [Bl, N]

    marks = []                                   # per timed pass: (start, loop end, decode end) events

    def one_pass(pass_idx):
        ids = [pass_idx * G + s for s in range(lo, hi)]
        noise = rdist.sample_noise(ids, (16, 2 * (H // 16), 2 * (W // 16)), 42, bf16, dev)
        lat = pipe._pack_latents(noise, Bl, 16, 2 * (H // 16), 2 * (W // 16))
        return run_image(pipe, lat, pe, pooled, hints, masks_l, H, W, args.inference_steps, marks if pass_idx >= 0 else None)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    pipe.capture_graphs = not args.no_graph
    # hipGraph replay of the loop (pipeline.GRAPH_CAPTURE): a call signature runs eagerly the first time and is captured the second
    # time, so two untimed passes precede the timed region when fewer warm-up passes were asked for (reported in `config`).
    n_pre = args.warmup if args.no_graph else max(args.warmup, 2)
    for w in range(n_pre):
        one_pass(-1 - w)
    barrier()
    t0 = time.perf_counter()
    first_out = None
    for k in range(args.steps):
        out = one_pass(k)
        if k == 0:
            first_out = out                      # a reference only (every pass returns a fresh tensor): checked AFTER the timed region
    torch.cuda.synchronize()
    mine_s = time.perf_counter() - t0            # this rank's own shard, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [mine_s]
    if world > 1:
        cdev = "cpu" if backend == "gloo" else dev
        tt = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        allr = [torch.zeros(1, device=cdev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allr, torch.tensor([mine_s], device=cdev, dtype=torch.float64))
        per_rank = [float(t.item()) for t in allr]

    # ---- the bench looks at what it computed (outside the timed region): pass 0 is run again with the same sample ids
    check = None
    if first_out is not None:
        again = one_pass(0)
        lat = pipe._master_latents
        u8 = first_out.to(torch.float32)
        hist = torch.bincount(first_out.flatten().to(torch.int64), minlength=256).float()
        check = {"latents_finite": bool(torch.isfinite(lat).all()), "image_u8_std": round(float(u8.std()), 2), "image_u8_mean": round(float(u8.mean()), 2),
                 "image_distinct_levels": int((hist > 0).sum()), "saturated_frac": round(float((hist[0] + hist[255]) / hist.sum()), 4),
                 "bitwise_repeat_same_ids": bool(torch.equal(first_out, again)),
                 "differs_between_ids": bool(not torch.equal(first_out, out)) if args.steps > 1 else None,
                 "image_crc": int(first_out.to(torch.int64).flatten().mul(torch.arange(1, first_out.numel() + 1, device=dev) % 65521).sum().item() % (1 << 61))}
        check["ok"] = bool(check["latents_finite"] and check["image_u8_std"] > 1.0 and check["image_distinct_levels"] > 16 and check["saturated_frac"] < 0.98
                           and check["bitwise_repeat_same_ids"] and check["differs_between_ids"] is not False)
        if world > 1:
            okt = torch.tensor([1.0 if check["ok"] else 0.0], device=("cpu" if backend == "gloo" else dev), dtype=torch.float64)
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            check["ok_all_ranks"] = bool(okt.item() > 0.5)

    timed_marks = marks[: args.steps]
    loop_ms = sorted(e[0].elapsed_time(e[1]) for e in timed_marks)[len(timed_marks) // 2] if timed_marks else None
    dec_ms = sorted(e[1].elapsed_time(e[2]) for e in timed_marks)[len(timed_marks) // 2] if timed_marks else None
    images = G * args.steps                                                   # all ranks, all timed passes
    value = images / elapsed
    n_tower = tower_blocks_read(cfg_t, cfg_c)
    fl_img = flops_per_image(H, W, args.inference_steps, args.text_lines, cfg_t, cfg_c, tower_blocks=n_tower)   # EXECUTED work
    fl_img_ref = flops_per_image(H, W, args.inference_steps, args.text_lines, cfg_t, cfg_c)                    # the reference's count
    full8 = args.precision in ("fp8", "fp8-mx")
    e2e_peak = 5.0e15 if full8 else 2.5e15                    # dense MFMA peak of the dtype the projections run in

    roofline = None
    if rank == 0 and not args.no_roofline_pass:
        import reptext_amd.pipeline as _pl
        pipe.capture_graphs = False              # per-launch events need the eager loop ...
        _ov, _pl.OVERLAP_TOWER = _pl.OVERLAP_TOWER, False          # ... with every kernel alone on the chip (no tower beside the transformer)
        with GemmTimer(ops) as gt:
            one_pass(10 ** 6)
        pipe.capture_graphs, _pl.OVERLAP_TOWER = not args.no_graph, _ov
        if args.precision == "bf16":
            n_launch, fl, sec = gt.result()
            peak, kname, tkey = 2500.0, "gemm_pp_kernel<bf16> (rt_gemm_bf16)", "gemm"
        else:                                    # dominant kernel of the fp8 run: the e4m3 instantiation, priced at the dense fp8 peak
            n_launch, fl, sec = gt.result(fp8=True)
            peak, kname, tkey = 5000.0, "gemm_pp_kernel<e4m3> (rt_gemm_fp8)", ("fp8mx:" if args.precision == "fp8-mx" else "fp8:") + "gemm_pp_kernel"
        ach = fl / sec / 1e12
        roofline = {"bound": "mfma", "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": pmc_traffic_bytes(tkey) if tkey else None, "kernel": kname, "launches": n_launch,
                    "avg_launch_us": round(sec / n_launch * 1e6, 2), "avg_gflop_per_launch": round(fl / n_launch / 1e9, 2),
                    "e2e_tflops_per_gpu": round(fl_img * Bl * args.steps / elapsed / 1e12, 1),
                    "e2e_frac": round(fl_img * Bl * args.steps / elapsed / e2e_peak, 4), "e2e_peak_tflops": e2e_peak / 1e12,
                    "e2e_pflop_per_image_executed": round(fl_img / 1e15, 4), "e2e_pflop_per_image_reference": round(fl_img_ref / 1e15, 4)}
        roofline["per_shape"] = gt.per_shape(fp8=None if args.precision == "bf16" else True, peak_tflops=peak)
        na, fla, seca = gt.attention_result()
        if na:
            pk = 5000.0 if full8 else 2500.0
            roofline["attention_kernel"] = {"kernel": ("attention_v3_kernel (attention_fwd_kernel for S < 1536 or S % 256 != 0)" if not full8
                                                       else "attention_fp8_kernel"), "launches": na,
                                            "avg_launch_us": round(seca / na * 1e6, 2), "achieved": round(fla / seca / 1e12, 1), "peak": pk,
                                            "frac": round(fla / seca / (pk * 1e12), 4)}
        if args.precision != "bf16":
            nb, flb, secb = gt.result(fp8=False)
            roofline["bf16_gemm_launches"] = {"launches": nb, "achieved": round(flb / secb / 1e12, 1), "peak": 2500.0,
                                              "frac": round(flb / secb / 2.5e15, 4), "share_of_gemm_time": round(secb / (sec + secb), 3)}
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        pipe.capture_graphs = False
        cpu = cpu_baseline(cfg_t, H, W, args.inference_steps, args.text_lines, cfg_c, pipe=pipe if args.depth_scale == 1.0 and args.precision == "bf16" else None)

    if rank == 0:
        line = {
            "metric": f"images/sec (whole node), FLUX.1-dev+RepText CN, {H}^2, {args.inference_steps} steps",
            "value": round(value, 4), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "sec_per_image": round(elapsed / (args.steps * Bl), 4),
            "loop_only_ms_per_step": None if loop_ms is None else round(loop_ms, 2),          # rank 0, median over the timed passes
            "vae_decode_ms_per_step": None if dec_ms is None else round(dec_ms, 2),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp8-ln": "fp8 (e4m3 to_q/k/v, add_*_proj, ff.net.0, proj_mlp) + bf16",
                      "fp8": "fp8 (e4m3 block projections and attention; bf16 storage, fp32 residual stream)",
                      "fp8-mx": "fp8 (e4m3 block projections and attention, MX block-scaled activations out of the attention / GELU epilogues; "
                                "fp32 residual stream)"}[args.precision], "data": "synthetic",
            "config": {"workload": f"FLUX.1-dev (19+38 blocks) + RepText ControlNet (6+0), {H}x{W}, {args.inference_steps} steps, "
                                   f"{args.text_lines} text line(s), batch {Bl}/GPU on rank 0, denoise loop + VAE decode to uint8, random-init weights; "
                                   f"tower blocks evaluated {n_tower} of {cfg_c['num_layers']} (the last sample is never read, Q5)",
                       "global_batch": G, "conditioning": "shared prompt/hint/mask" if args.shared_prompt else "one prompt, hint and mask per image", "parallelism": f"batch-shard x{world}, one broadcast" + (f" ({'RCCL' if backend == 'nccl' else backend})" if world > 1 else ""),
                       "launch": ("eager (one ctypes launch per kernel)" if args.no_graph else f"denoise loop replayed from one hipGraph per call signature ({n_pre} untimed passes: eager, then capture)")
                                 + ("; ControlNet tower on a side stream beside the transformer" if os.environ.get("RT_OVERLAP_TOWER", "1") == "1" else "")},
            "per_rank_s": {"max": round(max(per_rank), 4), "min": round(min(per_rank), 4), "all": [round(x, 4) for x in per_rank]},
            "output_check": check, "roofline": roofline, "cpu_baseline": cpu,
        }
        if args.depth_scale != 1.0:
            line["INVALID_debug_depth_scale"] = args.depth_scale
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if check is not None and not (check.get("ok_all_ranks", check["ok"])):
        print(f"[bench] OUTPUT CHECK FAILED on rank {rank}: {check}", file=sys.stderr, flush=True)
        raise SystemExit(4)


def run_image(pipe, latents, pe, pooled, hints, rowscales, H, W, steps, marks=None):
    """The timed region: set up the schedule (host scalars), run the loop, decode to uint8.
    ``marks``: list receiving (start, loop_end, decode_end) events on the launch stream (loop-only / decode split)."""
    import numpy as np
    from reptext_amd.pipeline import retrieve_timesteps
    from reptext_amd.scheduler import calculate_shift

    dev = latents.device
    sc = pipe.scheduler.config
    mu = calculate_shift((H // 16) * (W // 16), sc.base_image_seq_len, sc.max_image_seq_len, sc.base_shift, sc.max_shift)
    timesteps, n = retrieve_timesteps(pipe.scheduler, steps, dev, None, np.linspace(1.0, 1 / steps, steps), mu=mu)
    h2, w2 = 2 * (H // 16), 2 * (W // 16)
    B = latents.shape[0]
    text_ids = torch.zeros(pe.shape[1], 3, device=dev, dtype=latents.dtype)
    image_ids = pipe._prepare_latent_image_ids(B, h2, w2, dev, latents.dtype)
    pipe._guidance_scale, pipe._joint_attention_kwargs, pipe._interrupt = 3.5, None, False
    masks = [m.reshape(m.shape[0], -1, 1) for m in rowscales]          # per line [B, N, 1] (one regional mask per image)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if marks is not None else None
    if ev:
        ev[0].record()
    lat = pipe._denoise(latents, pe, pooled, text_ids, image_ids, timesteps, hints, masks, 3.5, 1.0, steps, None, None, [], n)
    if ev:
        ev[1].record()
    out = pipe.vae.decode_packed(lat, h2, w2, output_u8=True)
    if ev:
        ev[2].record()
        marks.append(ev)
    return out


if __name__ == "__main__":
    main()
