"""The flow of the reference's infer.py (infer.py:25-134) on this build, end to end on one MI355X, without checkpoints or cv2.

Same steps: render each text line with PIL, bbox -> position / regional masks, Canny(50,100) inverted (reptext_amd.hints),
then FluxControlNetPipeline.__call__ with control_image / control_position / control_mask / control_glyph and a seeded generator.
Weights are random-init FLUX.1-dev / RepText shapes (no network); prompt embeddings are random tensors in place of the T5/CLIP
encoders, which sit outside the hot path. With real checkpoints on disk use `FluxControlNetModel.from_pretrained(path)` /
`FluxControlNetPipeline.from_pretrained(path, controlnet=...)` and pass `prompt=` instead.

    python tools/infer_demo.py [--size 1024] [--steps 30] [--depth-scale 1.0] [--out gpurun_out/result.jpg]
    python tools/infer_demo.py --inpaint      # infer_inpaint.py's flow (infer_inpaint.py:48-155): second 68-channel tower, masked
                                              # background image, position mask = bbox+-5, true CFG with negative embeddings
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import ImageFont

from controlnet_flux import FluxControlNetModel
from pipeline_flux_controlnet import FluxControlNetPipeline
from reptext_amd import hints
from reptext_amd.config import flux_dev_transformer_config, flux_vae_config, reptext_controlnet_config
from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
from reptext_amd.transformer import FluxTransformer2DModel
from reptext_amd.vae import AutoencoderKL

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--steps", type=int, default=30)            # infer.py:127
ap.add_argument("--depth-scale", type=float, default=1.0)
ap.add_argument("--out", default="gpurun_out/result.jpg")
ap.add_argument("--inpaint", action="store_true")
a = ap.parse_args()
dev, bf16 = torch.device("cuda:0"), torch.bfloat16
ct, cc = flux_dev_transformer_config(), reptext_controlnet_config()
if a.depth_scale != 1.0:
    ct["num_layers"] = max(1, int(ct["num_layers"] * a.depth_scale)); ct["num_single_layers"] = max(1, int(ct["num_single_layers"] * a.depth_scale))
    cc["num_layers"] = max(1, int(cc["num_layers"] * a.depth_scale))
controlnet = FluxControlNetModel(**cc, device=dev, dtype=bf16).random_init_(seed=1)
pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), AutoencoderKL(**flux_vae_config(), device=dev, dtype=bf16).random_init_(seed=2),
                              None, None, None, None, FluxTransformer2DModel(**ct, device=dev, dtype=bf16).random_init_(seed=0), controlnet)
pipe.set_progress_bar_config(disable=True)

width = height = a.size
font = ImageFont.truetype("DejaVuSans.ttf", max(a.size * 80 // 1024, 12))      # infer.py:39-41 uses Arial Unicode, 80 px
text_list = ["مرحبا", "RepText"]
text_position_list = [(a.size * 370 // 1024, a.size * 200 // 1024), (a.size * 370 // 1024, a.size * 330 // 1024)]
text_color_list = [(255, 255, 255), (255, 255, 255)]
control_image_list, control_position_list, control_mask_list, control_glyph_all = hints.build_text_hints(
    text_list, text_position_list, text_color_list, font, width, height)

g = torch.Generator().manual_seed(1)
prompt_embeds = torch.randn(1, 512, 4096, generator=g).to(dev, bf16)
pooled = torch.randn(1, 768, generator=g).to(dev, bf16)
generator = torch.Generator(device="cuda").manual_seed(42)                      # infer.py:113

if a.inpaint:
    import numpy as np
    from PIL import Image
    from pipeline_flux_controlnet_inpaint import FluxControlNetPipeline as InpaintPipeline

    ci = dict(cc, extra_condition_channels=4)                                    # 64 + 4 hint channels (INP:807-813)
    controlnet_inpaint = FluxControlNetModel(**ci, device=dev, dtype=bf16).random_init_(seed=3)
    ipipe = InpaintPipeline(pipe.scheduler, pipe.vae, None, None, None, None, pipe.transformer, controlnet, controlnet_inpaint=controlnet_inpaint)
    ipipe.set_progress_bar_config(disable=True)
    yy, xx = np.mgrid[0:height, 0:width]
    background = Image.fromarray(np.stack([(xx * 255 // width), (yy * 255 // height), np.full_like(xx, 96)], axis=-1).astype(np.uint8))
    background = hints.resize_img(background, max_side=a.size, min_side=a.size)
    width, height = background.size
    imgs, pos, masks, glyph = hints.build_text_hints(text_list[:1], text_position_list[:1], [(0, 255, 0)], font, width, height, position_margin=5)
    neg_e, neg_p = torch.randn(1, 512, 4096, generator=g).to(dev, bf16), torch.randn(1, 768, generator=g).to(dev, bf16)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        image = ipipe(prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled, negative_prompt_embeds=neg_e, negative_pooled_prompt_embeds=neg_p,
                      true_guidance_scale=3.5, control_image=imgs, control_position=pos, control_mask=masks, control_glyph=glyph,
                      controlnet_conditioning_scale=1.0, controlnet_conditioning_step=30, control_image_inpaint=background,
                      control_mask_inpaint=masks[-1], controlnet_conditioning_scale_inpaint=1.0, width=width, height=height,
                      num_inference_steps=a.steps, guidance_scale=3.5, generator=generator).images[0]
        torch.cuda.synchronize()
        print(f"inpaint call {it}: {time.perf_counter() - t0:.2f} s, {a.steps} steps, {width}x{height}, CFG (internal batch 2), two towers")
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    image.save(a.out)
    print("saved", a.out, image.size)
    sys.exit(0)

for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    image = pipe(prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled, control_image=control_image_list,
                 control_position=control_position_list, control_mask=control_mask_list, control_glyph=control_glyph_all,
                 controlnet_conditioning_scale=1.0, controlnet_conditioning_step=30, width=width, height=height,
                 num_inference_steps=a.steps, guidance_scale=3.5, generator=generator).images[0]
    torch.cuda.synchronize()
    print(f"call {it}: {time.perf_counter() - t0:.2f} s for {len(text_list)} text lines, {a.steps} steps, {width}x{height} (hint VAE-encodes + loop + decode)")
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
image.save(a.out)
print("saved", a.out, image.size)
