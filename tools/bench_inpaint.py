"""BASELINE config 4 timed at its own size: `FluxControlNetPipeline` of pipeline_inpaint (infer_inpaint.py's flow, INP:1016-1300) at
1024x1024, 28 steps, FULL depth — FLUX.1-dev transformer 19+38, the RepText tower 6+0 (64 extra hint channels) and the inpaint tower 6+0
(4 extra channels), true CFG (internal batch 2), one masked text line, random-init weights, conditioning resident in HBM, latents out
(no VAE: the decode is the one bench.py times). Prints one JSON line; not the headline metric (that is bench.py at config 2)."""
import json, os, sys, time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image


def main():
    steps = int(os.environ.get("RT_INPAINT_STEPS", "28"))
    reps = int(os.environ.get("RT_INPAINT_REPS", "2"))
    gpu = torch.device("cuda", 0)
    from reptext_amd.config import flux_dev_transformer_config, reptext_controlnet_config
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline_inpaint import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    cfg_t, cfg_c = flux_dev_transformer_config(), reptext_controlnet_config()
    cfg_i = dict(cfg_c, extra_condition_channels=4)
    tr = FluxTransformer2DModel(**cfg_t, device=gpu, dtype=torch.bfloat16).random_init_(1)
    cn = FluxControlNetModel(**cfg_c, device=gpu, dtype=torch.bfloat16).random_init_(2)
    cni = FluxControlNetModel(**cfg_i, device=gpu, dtype=torch.bfloat16).random_init_(3)
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), None, None, None, None, None, tr, cn, cni)
    pipe.set_progress_bar_config(disable=True)
    H = W = 1024
    N, T = 4096, 512
    g = torch.Generator(device=gpu).manual_seed(3)
    r = lambda *s: torch.randn(*s, device=gpu, generator=g).to(torch.bfloat16)
    pe, pooled, npe, npooled = r(1, T, 4096), r(1, 768), r(1, T, 4096), r(1, 768)
    hint, hint_inp, lat0 = r(1, N, 64 + cfg_c["extra_condition_channels"]), r(1, N, 68), r(1, N, 64)
    m = np.zeros([H, W], dtype=np.uint8)
    m[200:420, 300:800] = 255
    mask = Image.fromarray(m)

    def call():
        return pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled, negative_prompt_embeds=npe, negative_pooled_prompt_embeds=npooled,
                    height=H, width=W, num_inference_steps=steps, guidance_scale=3.5, true_guidance_scale=2.0, control_image=[hint],
                    control_image_inpaint=hint_inp, control_mask=[mask], controlnet_conditioning_scale=1.0,
                    controlnet_conditioning_scale_inpaint=1.0, controlnet_conditioning_step=30, latents=lat0, output_type="latent").images

    first = call()
    torch.cuda.synchronize()
    second = call()                              # a second warm call (graph capture, if the pipeline captures)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = call()
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / reps
    ok = bool(torch.isfinite(out).all()) and torch.equal(out, second) and not torch.equal(out, lat0.float())
    print(json.dumps({"config": "C4: pipeline_inpaint, 1024x1024, %d steps, true CFG (internal batch 2), transformer 19+38, towers 6+0 (x2)" % steps,
                      "sec_per_image_loop_only": round(sec, 4), "images_per_sec": round(1 / sec, 4), "reps": reps, "finite_and_bitwise_repeat": ok,
                      "latents_moved_rel": round(float((out - lat0.float()).norm() / lat0.float().norm()), 3), "data": "synthetic, random-init weights"}))
    if not ok:
        raise SystemExit(4)


if __name__ == "__main__":
    main()
