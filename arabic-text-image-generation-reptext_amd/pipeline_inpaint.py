"""FluxControlNetPipeline, inpainting variant (RepText tower + a second, inpaint ControlNet + true CFG).

Interface parity target: /root/reference/RepText/pipeline_flux_controlnet_inpaint.py (same class name, other module)
  * constructor with ``controlnet_inpaint`` and a binarising mask processor ........ INP:195-239
  * encode_prompt with negative prompt ............................................. INP:333-448
  * prepare_latents_reptext where the glyph blend IS the initial noise ............. INP:598-653 (vs quirk Q1 of the base)
  * prepare_image_with_mask (masked image -> latents ‖ nearest-resized 1-mask) .... INP:761-826
  * __call__ keywords ............................................................... INP:846-883
  * loop: text towers (masked, summed) + inpaint tower (unmasked) + CFG mix with a zero first step (Q6,Q7,Q8) ... INP:1138-1285
Everything heavy is shared with pipeline.py; this file only adds the deltas.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .controlnet import FluxControlNetModel
from .image_processor import PipelineImageInput, VaeImageProcessor
from .pipeline import FluxControlNetPipeline as _BasePipeline
from .pipeline import FluxPipelineOutput, calculate_shift, retrieve_timesteps
from .utils import randn_tensor

DEFAULT_NEGATIVE_PROMPT = "bad quality, worst quality, text, signature, watermark, extra words"     # INP:416


class FluxControlNetPipeline(_BasePipeline):
    def __init__(self, scheduler, vae, text_encoder, tokenizer, text_encoder_2, tokenizer_2, transformer, controlnet,
                 controlnet_inpaint):
        super().__init__(scheduler, vae, text_encoder, tokenizer, text_encoder_2, tokenizer_2, transformer, controlnet)
        self.controlnet_inpaint = controlnet_inpaint
        self.mask_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor, do_resize=True, do_convert_grayscale=True,
                                                do_normalize=False, do_binarize=True)

    @property
    def components(self) -> Dict[str, Any]:
        c = super().components
        c["controlnet_inpaint"] = self.controlnet_inpaint
        return c

    # ------------------------------------------------------------------ INP:333-448
    def encode_prompt(self, prompt, prompt_2, device=None, num_images_per_prompt: int = 1, do_classifier_free_guidance: bool = True,
                      negative_prompt=None, negative_prompt_2=None, prompt_embeds=None, pooled_prompt_embeds=None,
                      max_sequence_length: int = 512, lora_scale=None, negative_prompt_embeds=None,
                      negative_pooled_prompt_embeds=None):
        """Returns (prompt_embeds, pooled, negative_prompt_embeds, negative_pooled, text_ids). Pre-computed negative
        embeddings may be passed (extension) when the pipeline has no text encoders."""
        pe, pooled, text_ids = super().encode_prompt(prompt, prompt_2, device=device, num_images_per_prompt=num_images_per_prompt,
                                                     prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
                                                     max_sequence_length=max_sequence_length)
        npe = npooled = None
        if do_classifier_free_guidance:
            if negative_prompt_embeds is not None:
                npe, npooled = negative_prompt_embeds, negative_pooled_prompt_embeds
            else:
                neg = negative_prompt or DEFAULT_NEGATIVE_PROMPT
                neg2 = negative_prompt_2 or neg
                npooled = self._get_clip_prompt_embeds(neg, num_images_per_prompt, device)
                npe = self._get_t5_prompt_embeds(neg2, num_images_per_prompt, max_sequence_length, device)
        return pe, pooled, npe, npooled, text_ids

    # ------------------------------------------------------------------ INP:598-653
    def prepare_latents_reptext(self, image, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        """As the base, but the glyph-blended tensor becomes the initial latents (INP:645-647): 0.10·glyph latent + noise where
        the down-sampled glyph mask is positive."""
        h2 = 2 * (int(height) // self.vae_scale_factor)
        w2 = 2 * (int(width) // self.vae_scale_factor)
        image = image.to(device=device, dtype=dtype)
        image_latents = self._encode_vae_image(image=image, generator=generator)
        n_img = image_latents.shape[0]
        if batch_size > n_img and batch_size % n_img == 0:
            image_latents = torch.cat([image_latents] * (batch_size // n_img), dim=0)
        elif batch_size > n_img:
            raise ValueError(f"Cannot duplicate `image` of batch size {n_img} to {batch_size} text prompts.")
        ids = self._prepare_latent_image_ids(batch_size, h2, w2, device, dtype)
        if latents is not None:
            return latents.to(device=device, dtype=dtype), ids
        noise = randn_tensor((batch_size, num_channels_latents, h2, w2), generator=generator, device=device, dtype=dtype)
        blended = self._glyph_blend(image, image_latents, noise)
        return self._pack_latents(blended.to(dtype), batch_size, num_channels_latents, h2, w2), ids

    # ------------------------------------------------------------------ INP:761-826
    def prepare_image_with_mask(self, image, mask, width, height, batch_size, num_images_per_prompt, device, dtype,
                                do_classifier_free_guidance=False):
        """Masked source image (masked pixels = -1) -> VAE latents (global-RNG posterior sample, Q2), concatenated with the
        nearest-resized inverted mask as a 17th channel, packed -> [B, N, 68]."""
        if not isinstance(image, torch.Tensor):
            image = self.image_processor.preprocess(image, height=height, width=width)
        repeat_by = batch_size if image.shape[0] == 1 else num_images_per_prompt
        image = image.repeat_interleave(repeat_by, dim=0).to(device=device, dtype=dtype)
        if not isinstance(mask, torch.Tensor):
            mask = self.mask_processor.preprocess(mask, height=height, width=width)
        mask = mask.repeat_interleave(repeat_by, dim=0).to(device=device, dtype=dtype)
        masked = torch.where((mask > 0.5).repeat(1, 3, 1, 1), torch.full_like(image, -1.0), image)
        lat = self.vae.encode(masked.to(self.vae.dtype)).latent_dist.sample()
        lat = ((lat - self.vae.config.shift_factor) * self.vae.config.scaling_factor).to(dtype)
        msize = (height // self.vae_scale_factor * 2, width // self.vae_scale_factor * 2)
        m = (ops.resize2d(mask.float(), size=msize, mode="nearest") if mask.is_cuda else F.interpolate(mask.float(), size=msize)).to(dtype)
        both = torch.cat([lat, 1 - m], dim=1)
        packed = self._pack_latents(both, batch_size * num_images_per_prompt, both.shape[1], both.shape[2], both.shape[3])
        if do_classifier_free_guidance:
            packed = torch.cat([packed] * 2)
        return packed, height, width

    def _is_packed_inpaint_hint(self, t) -> bool:
        cn = self.controlnet_inpaint
        return isinstance(t, torch.Tensor) and t.dim() == 3 and isinstance(cn, FluxControlNetModel) and \
            t.shape[-1] == cn.controlnet_x_embedder.weight.shape[1]

    # ------------------------------------------------------------------ INP:846-1313
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]] = None, prompt_2: Optional[Union[str, List[str]]] = None,
                 true_guidance_scale: float = 3.5, negative_prompt: Optional[Union[str, List[str]]] = None,
                 negative_prompt_2: Optional[Union[str, List[str]]] = None, height: Optional[int] = None, width: Optional[int] = None,
                 num_inference_steps: int = 28, timesteps: List[int] = None, guidance_scale: float = 7.0,
                 control_guidance_start: Union[float, List[float]] = 0.0, control_guidance_end: Union[float, List[float]] = 1.0,
                 control_image: PipelineImageInput = None, control_mode: Optional[Union[int, List[int]]] = None,
                 controlnet_conditioning_scale: Union[float, List[float]] = 1.0, controlnet_conditioning_step: int = 30,
                 num_images_per_prompt: Optional[int] = 1,
                 generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
                 latents: Optional[torch.FloatTensor] = None, prompt_embeds: Optional[torch.FloatTensor] = None,
                 pooled_prompt_embeds: Optional[torch.FloatTensor] = None, output_type: Optional[str] = "pil",
                 return_dict: bool = True, joint_attention_kwargs: Optional[Dict[str, Any]] = None,
                 callback_on_step_end: Optional[Callable[[int, int, Dict], None]] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], max_sequence_length: int = 512,
                 control_mask: Optional[torch.FloatTensor] = None, control_position: Optional[torch.FloatTensor] = None,
                 control_glyph: Optional[torch.FloatTensor] = None, control_image_inpaint: PipelineImageInput = None,
                 control_mask_inpaint: Optional[torch.FloatTensor] = None,
                 controlnet_conditioning_scale_inpaint: Union[float, List[float]] = 1.0,
                 negative_prompt_embeds: Optional[torch.FloatTensor] = None,
                 negative_pooled_prompt_embeds: Optional[torch.FloatTensor] = None):
        height = height or self.default_sample_size * self.vae_scale_factor
        width = width or self.default_sample_size * self.vae_scale_factor
        self.check_inputs(prompt, prompt_2, height, width, prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
                          callback_on_step_end_tensor_inputs=callback_on_step_end_tensor_inputs, max_sequence_length=max_sequence_length)
        self._guidance_scale, self._joint_attention_kwargs, self._interrupt = guidance_scale, joint_attention_kwargs, False
        cfg = self.do_classifier_free_guidance                          # enabled by guidance_scale > 1, scaled by true_guidance_scale (Q8)

        if isinstance(prompt, str):
            batch_size = 1
        elif isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        device, dtype = self._execution_device, self.transformer.dtype
        total = batch_size * num_images_per_prompt
        if cfg and total != 1:
            # INP:1033-1035,1145: latents keep batch B while the conditioning is 2B; the reference only broadcasts for B == 1 (Q6)
            raise ValueError("classifier-free guidance in this pipeline supports a single image per call (batch 1), as the reference does")

        pe, pooled, npe, npooled, text_ids = self.encode_prompt(
            prompt=prompt, prompt_2=prompt_2, prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
            do_classifier_free_guidance=cfg, negative_prompt=negative_prompt, negative_prompt_2=negative_prompt_2, device=device,
            num_images_per_prompt=num_images_per_prompt, max_sequence_length=max_sequence_length,
            negative_prompt_embeds=negative_prompt_embeds, negative_pooled_prompt_embeds=negative_pooled_prompt_embeds)
        pe, pooled = pe.to(device), pooled.to(device)
        if cfg:
            pe = torch.cat([npe.to(device), pe], dim=0)                 # negative first (INP:1034-1035)
            pooled = torch.cat([npooled.to(device), pooled], dim=0)

        hints: List[torch.Tensor] = []
        if isinstance(self.controlnet, FluxControlNetModel) and control_image is not None:
            positions = control_position if control_position is not None else [None] * len(control_image)
            for img, pos in zip(control_image, positions):
                if self._is_packed_hint(img):
                    h = img.to(device=device, dtype=dtype)
                    hints.append(torch.cat([h] * 2) if (cfg and h.shape[0] == total) else h)
                else:
                    h, height, width = self.prepare_image(image=img, image_position=pos, width=width, height=height, batch_size=total,
                                                          num_images_per_prompt=num_images_per_prompt, device=device, dtype=dtype,
                                                          do_classifier_free_guidance=cfg)
                    hints.append(h)
        if self._is_packed_inpaint_hint(control_image_inpaint):
            hint_inp = control_image_inpaint.to(device=device, dtype=dtype)
            if cfg and hint_inp.shape[0] == total:
                hint_inp = torch.cat([hint_inp] * 2)
        else:
            hint_inp, height, width = self.prepare_image_with_mask(image=control_image_inpaint, mask=control_mask_inpaint, width=width,
                                                                   height=height, batch_size=total, num_images_per_prompt=num_images_per_prompt,
                                                                   device=device, dtype=dtype, do_classifier_free_guidance=cfg)

        num_channels_latents = self.transformer.config.in_channels // 4
        sigmas = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)
        image_seq_len = (int(height) // self.vae_scale_factor) * (int(width) // self.vae_scale_factor)
        sc = self.scheduler.config
        mu = calculate_shift(image_seq_len, sc.base_image_seq_len, sc.max_image_seq_len, sc.base_shift, sc.max_shift)
        timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, timesteps, sigmas, mu=mu)

        if control_glyph is not None:
            init_image = self.image_processor.preprocess(control_glyph, height=height, width=width).to(dtype=torch.float32)
            latents, image_ids = self.prepare_latents_reptext(init_image, total, num_channels_latents, height, width, pe.dtype, device,
                                                              generator, None)
        else:
            latents, image_ids = self.prepare_latents(total, num_channels_latents, height, width, pe.dtype, device, generator, latents)
        self._num_timesteps = len(timesteps)
        masks = self._region_masks(control_mask, latents.device, latents.dtype)

        latents = self._denoise_inpaint(latents, pe, pooled, text_ids, image_ids, timesteps, hints, masks, hint_inp, guidance_scale,
                                        true_guidance_scale, cfg, controlnet_conditioning_scale, controlnet_conditioning_scale_inpaint,
                                        controlnet_conditioning_step, control_mode, callback_on_step_end,
                                        callback_on_step_end_tensor_inputs, num_inference_steps)

        if output_type == "latent":
            # The parity tap (PIPE:1132-1133). The loop's state is kept in fp32 (A.6: the scheduler steps in fp32), and that state
            # is what is returned: rounding it to bf16 here would by itself cost 1.8e-3 rel-L2, twice the whole loop's error.
            # `.to(torch.bfloat16)` gives the reference's bf16-run dtype.
            image = self._master_latents
        else:
            h2, w2 = 2 * (int(height) // self.vae_scale_factor), 2 * (int(width) // self.vae_scale_factor)
            if output_type in ("pil", "np"):
                image = self.image_processor.postprocess_u8(self.vae.decode_packed(latents, h2, w2, output_u8=True), output_type)
            elif output_type == "pt":
                image = (self.vae.decode_packed(latents, h2, w2) / 2 + 0.5).clamp(0, 1)
            else:
                raise ValueError(f"unsupported output_type {output_type}")
        self.maybe_free_model_hooks()
        if not return_dict:
            return (image,)
        return FluxPipelineOutput(images=image)

    def _denoise_inpaint(self, latents, pe, pooled, text_ids, image_ids, timesteps, hints, masks, hint_inp, guidance_scale, true_scale,
                         cfg, cn_scale, cn_scale_inp, cn_steps, control_mode, callback, callback_inputs, num_inference_steps):
        device = latents.device
        B = latents.shape[0]
        tvals = timesteps.to(torch.float32).cpu().tolist()
        guidance = torch.full((B,), float(guidance_scale), device=device, dtype=torch.float32) if self.transformer.config.guidance_embeds else None
        # one regional mask per text line, shared by the batch ([1,N,1], the reference's form) or one per image ([B,N,1])
        rowscales = [m.to(torch.float32).reshape(-1).contiguous() if m.shape[0] == 1 else m.to(torch.float32).reshape(m.shape[0], -1).contiguous() for m in masks]
        num_warmup = max(len(timesteps) - num_inference_steps * self.scheduler.order, 0)
        model_ts = [self._model_timestep(t) for t in tvals]
        g_tab = guidance if guidance is None or pooled.shape[0] == B else guidance.expand(pooled.shape[0]).contiguous()
        tab_t = self.transformer.build_modulation_table(model_ts, g_tab, pooled)
        n_cn = max(0, min(len(model_ts), cn_steps))
        tab_c = self.controlnet.build_modulation_table(model_ts[:n_cn], g_tab, pooled) if (hints and n_cn > 0) else None
        tab_i = self.controlnet_inpaint.build_modulation_table(model_ts[:n_cn], g_tab, pooled) if (hints and n_cn > 0) else None
        # loop-invariant embeddings, once per call (see FluxControlNetPipeline._denoise)
        static_t = self.transformer.prepare_static(pe)
        static_c = [self.controlnet.prepare_static(pe, h) for h in hints] if (hints and n_cn > 0) else []
        static_i = self.controlnet_inpaint.prepare_static(pe, hint_inp) if (hints and n_cn > 0) else None
        # preallocated sample buffers; the sixth sample of a 6-block tower is never read by the 19-block transformer (Q5)
        blocks_needed, sample_buf, single_buf = None, None, None
        if hints and n_cn > 0:
            c1, c2 = self.controlnet, self.controlnet_inpaint
            n_cd, n_cs = len(c1.transformer_blocks), len(c1.single_transformer_blocks)
            n_td, n_ts = len(self.transformer.transformer_blocks), len(self.transformer.single_transformer_blocks)
            if (len(c2.transformer_blocks), len(c2.single_transformer_blocks)) == (n_cd, n_cs):
                need_d = 0 if n_cd == 0 else (n_td - 1) // int(np.ceil(n_td / n_cd)) + 1
                need_s = 0 if n_cs == 0 or n_ts == 0 else (n_ts - 1) // int(np.ceil(n_ts / n_cs)) + 1
                blocks_needed = (min(need_d, n_cd), min(need_s, n_cs))
            Bc, N_, d_ = pe.shape[0], latents.shape[1], c1.inner_dim
            sample_buf = [torch.empty(Bc, N_, d_, device=device, dtype=torch.bfloat16) for _ in range(n_cd)]
            single_buf = [torch.empty(Bc, N_, d_, device=device, dtype=torch.bfloat16) for _ in range(n_cs)]
        # fp32 master copy of the latents between steps (the models read its bf16 copy): the scheduler computes in fp32 anyway
        # (A.6); not rounding the STATE 28 times keeps the loop close to the fp32 reference path. Callbacks see the bf16 copy.
        lat32 = latents.to(torch.float32).contiguous()
        latents = latents.to(torch.bfloat16).contiguous().clone()
        with self.progress_bar(total=num_inference_steps) as bar:
            for i, t in enumerate(tvals):
                if self.interrupt:
                    continue
                timestep = torch.full((B,), self._model_timestep(t), device=device, dtype=torch.float32)
                merged = merged_single = None
                for line, hint in enumerate(hints):
                    if i >= cn_steps:
                        samples = single_samples = None
                    else:
                        rs = rowscales[line] if rowscales else None
                        samples, single_samples = self.controlnet(
                            hidden_states=latents, controlnet_cond=hint, controlnet_mode=control_mode, conditioning_scale=cn_scale,
                            timestep=timestep, guidance=guidance, pooled_projections=pooled, encoder_hidden_states=pe, txt_ids=text_ids,
                            img_ids=image_ids, joint_attention_kwargs=self.joint_attention_kwargs, return_dict=False, _rowscale=rs,
                            _accumulate_into=sample_buf, _accumulate_single_into=single_buf, _overwrite=(line == 0),
                            _mods=None if tab_c is None else tab_c.step(i), _static=static_c[line] if static_c else None,
                            _blocks_needed=blocks_needed)
                    if line == 0:
                        merged, merged_single = samples, single_samples
                # The inpaint tower's residuals are ADDED to the text towers' — and dropped when those are absent (INP:1231-1245:
                # both sums are guarded by `control_block_samples is not None`). It is therefore only evaluated when it can matter.
                if merged is not None or merged_single is not None:
                    self.controlnet_inpaint(
                        hidden_states=latents, controlnet_cond=hint_inp, controlnet_mode=control_mode, conditioning_scale=cn_scale_inp,
                        timestep=timestep, guidance=guidance, pooled_projections=pooled, encoder_hidden_states=pe, txt_ids=text_ids,
                        img_ids=image_ids, joint_attention_kwargs=self.joint_attention_kwargs, return_dict=False,
                        _accumulate_into=sample_buf, _accumulate_single_into=single_buf, _mods=None if tab_i is None else tab_i.step(i),
                        _static=static_i, _blocks_needed=blocks_needed)
                noise_pred = self.transformer(
                    hidden_states=latents, timestep=timestep, guidance=guidance, pooled_projections=pooled, encoder_hidden_states=pe,
                    controlnet_block_samples=merged, controlnet_single_block_samples=merged_single, txt_ids=text_ids, img_ids=image_ids,
                    joint_attention_kwargs=self.joint_attention_kwargs, return_dict=False, _mods=tab_t.step(i), _static=static_t)[0]
                if cfg:
                    uncond, text = noise_pred[:B], noise_pred[B:]                          # chunk(2): negative first
                    if i > 0:
                        noise_pred = ops.cfg_mix(uncond, text, float(true_scale))
                    else:
                        noise_pred = torch.zeros_like(text)                                # first step: zero velocity (Q7)
                self.scheduler.step_master_(noise_pred, lat32, latents)
                if callback is not None:
                    env = {"latents": latents, "prompt_embeds": pe}
                    out = callback(self, i, timesteps[i], {k: env[k] for k in callback_inputs})
                    if "latents" in out:
                        latents = out.pop("latents").to(torch.bfloat16).contiguous()
                        lat32 = latents.to(torch.float32)
                    if "prompt_embeds" in out:                       # the loop-invariant embeddings are no longer valid
                        pe = out.pop("prompt_embeds")
                        static_t = self.transformer.prepare_static(pe)
                        static_c = [self.controlnet.prepare_static(pe, h) for h in hints] if static_c else []
                        static_i = self.controlnet_inpaint.prepare_static(pe, hint_inp) if static_i is not None else None
                if i == len(tvals) - 1 or ((i + 1) > num_warmup and (i + 1) % self.scheduler.order == 0):
                    bar.update()
        self._master_latents = lat32          # fp32 state of the loop; `latents` is its bf16 copy
        return latents
