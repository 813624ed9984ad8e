"""Debug aid: GPU pipeline latents vs fp32 oracle and vs the bf16-storage oracle, for loop variants (tower off / on / masked)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from PIL import Image
from oracle import flux_oracle as orc
from reptext_amd.controlnet import FluxControlNetModel
from reptext_amd.pipeline import FluxControlNetPipeline
from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
from reptext_amd.transformer import FluxTransformer2DModel

SMALL_T = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=3, attention_head_dim=128, num_attention_heads=4,
               joint_attention_dim=256, pooled_projection_dim=64, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
SMALL_CN = dict(SMALL_T, num_layers=2, num_single_layers=0, extra_condition_channels=64)
gpu = torch.device("cuda:0")
tp = orc.init_mmdit_params(SMALL_T, seed=11)
cp = orc.init_mmdit_params(SMALL_CN, seed=12, controlnet=True)
tr = FluxTransformer2DModel(**SMALL_T, device=gpu, dtype=torch.bfloat16)
cn = FluxControlNetModel(**SMALL_CN, device=gpu, dtype=torch.bfloat16)
tr.load_state_dict(tp); cn.load_state_dict(cp)
pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), None, None, None, None, None, tr, cn)
pipe.set_progress_bar_config(disable=True)
H = W = 256
N, T = 256, 64
g = torch.Generator().manual_seed(5)
r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
pe, pooled, hint = r(1, T, 256), r(1, 64), r(1, N, 128)
lat0 = orc.pack_latents(r(1, 16, 32, 32))
mask_np = np.zeros([H, W], dtype=np.uint8); mask_np[60:140, 80:200] = 255
rm = torch.nn.functional.interpolate(torch.from_numpy(mask_np)[None, None].float() / 255.0, scale_factor=1 / 16, mode="bilinear").reshape(1, -1, 1)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
b16 = lambda t: t.to(gpu, torch.bfloat16)
ids, tids = orc.latent_image_ids(32, 32), torch.zeros(T, 3)
for steps in (1, 2, 4):
    sig = orc.flow_sigmas(steps, orc.calculate_shift(N, 256, 4096, 0.5, 1.15))
    for name, hints, masks_o, masks_g, cstep in (("tower off", [hint], [], None, 0), ("tower on", [hint], [], None, 99),
                                                  ("tower masked", [hint], [rm], [Image.fromarray(mask_np)], 99)):
        ref = orc.denoise_loop(tp, SMALL_T, cp, SMALL_CN, lat0, pe, pooled, hints, masks_o, sig, ids, tids, 3.5, conditioning_step=cstep)
        with orc.stored_as(torch.bfloat16):
            ref16 = orc.denoise_loop(tp, SMALL_T, cp, SMALL_CN, lat0, pe, pooled, hints, masks_o, sig, ids, tids, 3.5, conditioning_step=cstep)
        out = pipe(prompt_embeds=b16(pe), pooled_prompt_embeds=b16(pooled), height=H, width=W, num_inference_steps=steps, guidance_scale=3.5,
                   control_image=[b16(h) for h in hints], control_mask=masks_g, controlnet_conditioning_step=cstep, latents=b16(lat0),
                   output_type="latent").images.float().cpu()
        print(f"steps {steps} {name:13s}: gpu-fp32 {rel(out, ref):.3e}  gpu-stored {rel(out, ref16):.3e}  floor {rel(ref16, ref):.3e}"
              f"  | bf16(ref16) vs ref {rel(ref16.bfloat16().float(), ref):.3e}")
