"""FluxControlNetPipeline (text-to-image with RepText glyph conditions) on the MI355X kernels.

Interface parity target: /root/reference/RepText/pipeline_flux_controlnet.py
  * constructor components ........................ PIPE:194-226
  * __call__ keyword names / defaults / returns ... PIPE:751-781, 1132-1148
  * check_inputs errors ........................... PIPE:485-531
  * helper statics (ids, pack, unpack) ............ PIPE:533-570
  * latent / hint preparation ..................... PIPE:573-731
  * denoising loop semantics (quirks Q1-Q5) ....... PIPE:1016-1130, SURVEY.md §8a
The control flow is organised as plan -> prepare -> denoise -> decode stages over device-resident state; the
per-step work is a fixed sequence of HIP launches (mmdit.py) with the regional mask and the sum over text lines fused
into the ControlNet's zero-linear epilogues.
"""
from __future__ import annotations

import contextlib
import inspect
import json
import os
from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .controlnet import FluxControlNetModel, FluxMultiControlNetModel
from .image_processor import PipelineImageInput, VaeImageProcessor
from .scheduler import FlowMatchEulerDiscreteScheduler, calculate_shift
from .transformer import FluxTransformer2DModel
from .utils import randn_tensor
from .vae import AutoencoderKL


@dataclass
class FluxPipelineOutput:
    images: Any


def retrieve_latents(encoder_output, generator: Optional[torch.Generator] = None, sample_mode: str = "sample"):
    """PIPE:91-101."""
    if hasattr(encoder_output, "latent_dist") and sample_mode == "sample":
        return encoder_output.latent_dist.sample(generator)
    if hasattr(encoder_output, "latent_dist") and sample_mode == "argmax":
        return encoder_output.latent_dist.mode()
    if hasattr(encoder_output, "latents"):
        return encoder_output.latents
    raise AttributeError("Could not access latents of provided encoder_output")


def retrieve_timesteps(scheduler, num_inference_steps: Optional[int] = None, device=None, timesteps: Optional[List[int]] = None,
                       sigmas: Optional[List[float]] = None, **kwargs):
    """PIPE:104-160: drive ``scheduler.set_timesteps`` with a step count, custom timesteps or custom sigmas."""
    if timesteps is not None and sigmas is not None:
        raise ValueError("Only one of `timesteps` or `sigmas` can be passed. Please choose one to set custom values")
    accepted = set(inspect.signature(scheduler.set_timesteps).parameters.keys())
    if timesteps is not None:
        if "timesteps" not in accepted:
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support custom"
                             f" timestep schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(timesteps=timesteps, device=device, **kwargs)
    elif sigmas is not None:
        if "sigmas" not in accepted:
            raise ValueError(f"The current scheduler class {scheduler.__class__}'s `set_timesteps` does not support custom"
                             f" sigmas schedules. Please check whether you are using the correct scheduler.")
        scheduler.set_timesteps(sigmas=sigmas, device=device, **kwargs)
    else:
        scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
    return scheduler.timesteps, len(scheduler.timesteps)


class _Progress:
    def __init__(self, total, disable=False):
        self.bar = None
        if not disable:
            try:
                from tqdm.auto import tqdm

                self.bar = tqdm(total=total)
            except Exception:
                self.bar = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        if self.bar is not None:
            self.bar.close()

    def update(self, n=1):
        if self.bar is not None:
            self.bar.update(n)


# The ControlNet tower runs on a side stream next to the transformer (see _denoise_eager): within a step the two chains are
# independent except that transformer block i reads tower sample i // 4. Bitwise neutral (test_tower_stream_overlap_is_bitwise_
# neutral). Measured on MI355X: -0.5 % per image with the eager loop, -1.1 % when the loop is replayed from its hipGraph (2.105-2.110
# vs 2.125-2.137 s, alternating runs on one box: the fork/join is then part of the graph instead of host-side event calls). On by
# default (RT_OVERLAP_TOWER=0 turns it off); bench.py's roofline pass and the rocprof summaries run with it off so that every kernel
# is alone on the chip when it is timed.
OVERLAP_TOWER = os.environ.get("RT_OVERLAP_TOWER", "1") == "1"
# RT_GRAPH=1 (default): the denoising loop of a call signature seen before is captured ONCE into a hipGraph (every kernel of the
# 28 steps, ~7 000 nodes, host scalars baked in) and replayed for later calls with the same signature: bitwise the eager result,
# the host returns after ~5 ms instead of enqueueing ~13 000 launches, the GPU loses the launch bubbles (-0.7 % per image).
# The first call of a signature always runs eagerly (it also warms every lazily built buffer); capture failures fall back to
# eager for good. `pipe.capture_graphs = False` (or RT_GRAPH=0) turns it off.
GRAPH_CAPTURE = os.environ.get("RT_GRAPH", "1") == "1"
GRAPH_CACHE_MAX = 2


class FluxControlNetPipeline:
    model_cpu_offload_seq = "text_encoder->text_encoder_2->transformer->vae"
    _optional_components: List[str] = []
    _callback_tensor_inputs = ["latents", "prompt_embeds"]

    def __init__(self, scheduler: FlowMatchEulerDiscreteScheduler, vae: AutoencoderKL, text_encoder, tokenizer, text_encoder_2,
                 tokenizer_2, transformer: FluxTransformer2DModel,
                 controlnet: Union[FluxControlNetModel, List[FluxControlNetModel], Tuple[FluxControlNetModel], FluxMultiControlNetModel]):
        self.vae, self.text_encoder, self.text_encoder_2 = vae, text_encoder, text_encoder_2
        self.tokenizer, self.tokenizer_2 = tokenizer, tokenizer_2
        self.transformer, self.scheduler, self.controlnet = transformer, scheduler, controlnet
        self.vae_scale_factor = 2 ** len(self.vae.config.block_out_channels) if self.vae is not None else 16       # Q9
        self.image_processor = VaeImageProcessor(vae_scale_factor=self.vae_scale_factor)
        self.tokenizer_max_length = self.tokenizer.model_max_length if self.tokenizer is not None else 77
        self.default_sample_size = 64
        self._guidance_scale, self._joint_attention_kwargs, self._interrupt, self._num_timesteps = 1.0, None, False, 0
        self._progress_disabled = False

    # ------------------------------------------------------------------ component plumbing
    @property
    def components(self) -> Dict[str, Any]:
        return dict(scheduler=self.scheduler, vae=self.vae, text_encoder=self.text_encoder, tokenizer=self.tokenizer,
                    text_encoder_2=self.text_encoder_2, tokenizer_2=self.tokenizer_2, transformer=self.transformer,
                    controlnet=self.controlnet)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path: str, controlnet=None, torch_dtype=None, **kwargs):
        """Loader in the diffusers layout (``model_index.json`` + one sub-folder per component), as
        infer.py:31-33 calls it. ``pretrained_model_name_or_path`` is a local directory or a hub id that resolves through the
        LOCAL hub cache (modules.resolve_model_path: ``$HF_HOME/hub/models--ORG--NAME/snapshots/…``; nothing is downloaded).
        ``model_index.json`` decides which components exist: the text encoders load into reptext_amd.text_encoders (HIP
        kernels) and the tokenizers through `transformers` when listed and present; else they stay None and the caller
        passes ``prompt_embeds``/``pooled_prompt_embeds``. Components passed as keyword arguments are taken as they are."""
        from .modules import resolve_model_path

        root = resolve_model_path(pretrained_model_name_or_path, kwargs.pop("revision", None))
        dt = torch_dtype or torch.bfloat16
        index = {}
        ip = os.path.join(root, "model_index.json")
        if os.path.isfile(ip):
            with open(ip) as f:
                index = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        listed = lambda name: (not index) or (index.get(name) not in (None, [None, None]))
        for required in ("transformer", "vae"):
            if required not in kwargs and not (listed(required) and os.path.isdir(os.path.join(root, required))):
                raise OSError(f"{cls.__name__}.from_pretrained: component '{required}' is neither passed nor present under {root}")
        sched_cfg = {}
        sp = os.path.join(root, "scheduler", "scheduler_config.json")
        if os.path.isfile(sp):
            with open(sp) as f:
                sched_cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        known = set(FlowMatchEulerDiscreteScheduler().config.keys())
        scheduler = kwargs.pop("scheduler", None) or FlowMatchEulerDiscreteScheduler(**{k: v for k, v in sched_cfg.items() if k in known})
        transformer = kwargs.pop("transformer", None) or FluxTransformer2DModel.from_pretrained(root, torch_dtype=dt, subfolder="transformer")
        vae = kwargs.pop("vae", None) or AutoencoderKL.from_pretrained(root, torch_dtype=dt, subfolder="vae")
        te, te2 = kwargs.pop("text_encoder", None), kwargs.pop("text_encoder_2", None)
        tok, tok2 = kwargs.pop("tokenizer", None), kwargs.pop("tokenizer_2", None)
        try:
            if te is None and listed("text_encoder") and os.path.isdir(os.path.join(root, "text_encoder")):
                from .text_encoders import CLIPTextModel                 # the encoders themselves run on the HIP kernels

                te = CLIPTextModel.from_pretrained(root, subfolder="text_encoder", torch_dtype=dt)
            if te2 is None and listed("text_encoder_2") and os.path.isdir(os.path.join(root, "text_encoder_2")):
                from .text_encoders import T5EncoderModel

                te2 = T5EncoderModel.from_pretrained(root, subfolder="text_encoder_2", torch_dtype=dt)
            if tok is None and listed("tokenizer") and os.path.isdir(os.path.join(root, "tokenizer")):
                from transformers import CLIPTokenizer                   # tokenisation is host-side string work

                tok = CLIPTokenizer.from_pretrained(os.path.join(root, "tokenizer"))
            if tok2 is None and listed("tokenizer_2") and os.path.isdir(os.path.join(root, "tokenizer_2")):
                from transformers import T5TokenizerFast

                tok2 = T5TokenizerFast.from_pretrained(os.path.join(root, "tokenizer_2"))
        except Exception as e:  # pragma: no cover - depends on local files
            raise OSError(f"could not load text encoders from {root}: {e}") from e
        if isinstance(controlnet, str):
            controlnet = FluxControlNetModel.from_pretrained(controlnet, torch_dtype=dt)
        extra = {k: kwargs[k] for k in ("controlnet_inpaint",) if k in kwargs}
        return cls(scheduler=scheduler, vae=vae, text_encoder=te, tokenizer=tok, text_encoder_2=te2, tokenizer_2=tok2,
                   transformer=transformer, controlnet=controlnet, **extra)

    def save_pretrained(self, root: str, max_shard_bytes: int = 10 << 30) -> None:
        """Write the diffusers layout this class loads: model_index.json, scheduler/scheduler_config.json and one folder per
        model component that has weights here (the ControlNet is a separate repository in the reference and is not part of it)."""
        os.makedirs(root, exist_ok=True)
        index = {"_class_name": type(self).__name__, "_diffusers_version": "0.36.0"}
        for name in ("transformer", "vae", "text_encoder", "text_encoder_2"):
            m = getattr(self, name, None)
            if m is not None and hasattr(m, "save_pretrained"):
                m.save_pretrained(os.path.join(root, name), max_shard_bytes=max_shard_bytes) if name in ("transformer", "vae") else m.save_pretrained(os.path.join(root, name))
                index[name] = ["reptext_amd", type(m).__name__]
            else:
                index[name] = [None, None]
        for name in ("tokenizer", "tokenizer_2"):
            t = getattr(self, name, None)
            if t is not None and hasattr(t, "save_pretrained"):
                t.save_pretrained(os.path.join(root, name))
                index[name] = ["transformers", type(t).__name__]
            else:
                index[name] = [None, None]
        os.makedirs(os.path.join(root, "scheduler"), exist_ok=True)
        with open(os.path.join(root, "scheduler", "scheduler_config.json"), "w") as f:
            json.dump(dict(_class_name="FlowMatchEulerDiscreteScheduler", **dict(self.scheduler.config)), f, indent=2)
        index["scheduler"] = ["reptext_amd", "FlowMatchEulerDiscreteScheduler"]
        with open(os.path.join(root, "model_index.json"), "w") as f:
            json.dump(index, f, indent=2)

    def to(self, device=None, dtype=None):
        for m in self.components.values():
            if isinstance(m, torch.nn.Module):
                m.to(device=device, dtype=dtype) if dtype is not None else m.to(device)
        return self

    @property
    def _execution_device(self):
        return self.transformer.device

    @property
    def device(self):
        return self.transformer.device

    def maybe_free_model_hooks(self):
        """Offload hooks belong to accelerate-based CPU offload, which this single-device path does not use (no-op)."""

    def set_progress_bar_config(self, disable: bool = False, **kw):
        self._progress_disabled = disable

    def progress_bar(self, total=None):
        return _Progress(total, disable=self._progress_disabled)

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def joint_attention_kwargs(self):
        return self._joint_attention_kwargs

    @property
    def num_timesteps(self):
        return self._num_timesteps

    @property
    def interrupt(self):
        return self._interrupt

    # ------------------------------------------------------------------ prompt encoding (PIPE:232-456)
    def _require_text_stack(self):
        if self.text_encoder is None or self.text_encoder_2 is None or self.tokenizer is None or self.tokenizer_2 is None:
            raise ValueError("this pipeline was built without text encoders/tokenizers: pass `prompt_embeds` and "
                             "`pooled_prompt_embeds` instead of `prompt`")

    def _get_t5_prompt_embeds(self, prompt, num_images_per_prompt=1, max_sequence_length=512, device=None, dtype=None):
        self._require_text_stack()
        device = device or self._execution_device
        prompt = [prompt] if isinstance(prompt, str) else prompt
        tok = self.tokenizer_2(prompt, padding="max_length", max_length=max_sequence_length, truncation=True,
                               return_length=False, return_overflowing_tokens=False, return_tensors="pt")
        emb = self.text_encoder_2(tok.input_ids.to(device), output_hidden_states=False)[0]
        emb = emb.to(dtype=self.text_encoder_2.dtype, device=device)
        b, seq, _ = emb.shape
        return emb.repeat(1, num_images_per_prompt, 1).view(b * num_images_per_prompt, seq, -1)

    def _get_clip_prompt_embeds(self, prompt, num_images_per_prompt=1, device=None):
        self._require_text_stack()
        device = device or self._execution_device
        prompt = [prompt] if isinstance(prompt, str) else prompt
        tok = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer_max_length, truncation=True,
                             return_overflowing_tokens=False, return_length=False, return_tensors="pt")
        pooled = self.text_encoder(tok.input_ids.to(device), output_hidden_states=False).pooler_output
        pooled = pooled.to(dtype=self.text_encoder.dtype, device=device)
        return pooled.repeat(1, num_images_per_prompt).view(len(prompt) * num_images_per_prompt, -1)

    def encode_prompt(self, prompt, prompt_2, device=None, num_images_per_prompt: int = 1, prompt_embeds=None,
                      pooled_prompt_embeds=None, max_sequence_length: int = 512, lora_scale=None):
        """Returns (prompt_embeds [B,L,4096], pooled [B,768], text_ids zeros[L,3]) — PIPE:349-456 (LoRA scaling is inert:
        no PEFT on this path)."""
        device = device or self._execution_device
        if prompt_embeds is None:
            prompt = [prompt] if isinstance(prompt, str) else prompt
            prompt_2 = prompt_2 or prompt
            prompt_2 = [prompt_2] if isinstance(prompt_2, str) else prompt_2
            pooled_prompt_embeds = self._get_clip_prompt_embeds(prompt, num_images_per_prompt, device)
            prompt_embeds = self._get_t5_prompt_embeds(prompt_2, num_images_per_prompt, max_sequence_length, device)
        ids_dtype = self.text_encoder.dtype if self.text_encoder is not None else prompt_embeds.dtype
        text_ids = torch.zeros(prompt_embeds.shape[1], 3, device=device, dtype=ids_dtype)
        return prompt_embeds, pooled_prompt_embeds, text_ids

    def _encode_vae_image(self, image: torch.Tensor, generator):
        """PIPE:459-471: sample the posterior (per-sample generators allowed) and apply (z - shift) * scaling."""
        if isinstance(generator, list):
            lat = torch.cat([retrieve_latents(self.vae.encode(image[i : i + 1]), generator=generator[i]) for i in range(image.shape[0])], dim=0)
        else:
            lat = retrieve_latents(self.vae.encode(image), generator=generator)
        return (lat - self.vae.config.shift_factor) * self.vae.config.scaling_factor

    # ------------------------------------------------------------------ validation (PIPE:485-531)
    def check_inputs(self, prompt, prompt_2, height, width, prompt_embeds=None, pooled_prompt_embeds=None,
                     callback_on_step_end_tensor_inputs=None, max_sequence_length=None):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_on_step_end_tensor_inputs is not None:
            bad = [k for k in callback_on_step_end_tensor_inputs if k not in self._callback_tensor_inputs]
            if bad:
                raise ValueError(f"`callback_on_step_end_tensor_inputs` has to be in {self._callback_tensor_inputs}, but found {bad}")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `prompt`: {prompt} and `prompt_embeds`: {prompt_embeds}. Please make sure to"
                             " only forward one of the two.")
        if prompt_2 is not None and prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `prompt_2`: {prompt_2} and `prompt_embeds`: {prompt_embeds}. Please make sure to"
                             " only forward one of the two.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if prompt_2 is not None and not isinstance(prompt_2, (str, list)):
            raise ValueError(f"`prompt_2` has to be of type `str` or `list` but is {type(prompt_2)}")
        if prompt_embeds is not None and pooled_prompt_embeds is None:
            raise ValueError("If `prompt_embeds` are provided, `pooled_prompt_embeds` also have to be passed. Make sure to generate "
                             "`pooled_prompt_embeds` from the same text encoder that was used to generate `prompt_embeds`.")
        if max_sequence_length is not None and max_sequence_length > 512:
            raise ValueError(f"`max_sequence_length` cannot be greater than 512 but is {max_sequence_length}")

    # ------------------------------------------------------------------ latent helpers (PIPE:533-570)
    @staticmethod
    def _prepare_latent_image_ids(batch_size, height, width, device, dtype):
        """(0,row,col) per token of the (height/2)x(width/2) grid, row-major (height/width are LATENT sizes)."""
        rows = torch.arange(height // 2, dtype=torch.float32)
        cols = torch.arange(width // 2, dtype=torch.float32)
        ids = torch.stack([torch.zeros(height // 2, width // 2), rows[:, None].expand(-1, width // 2),
                           cols[None, :].expand(height // 2, -1)], dim=-1)
        return ids.reshape(-1, 3).to(device=device, dtype=dtype)

    @staticmethod
    def _pack_latents(latents, batch_size, num_channels_latents, height, width):
        """[B,C,H,W] -> [B,(H/2)(W/2),4C] with channel order (c,dy,dx). GPU bf16 tensors use the HIP kernel."""
        if latents.is_cuda and latents.dtype == torch.bfloat16:
            return ops.pack_latents(latents.reshape(batch_size, num_channels_latents, height, width))
        x = latents.reshape(batch_size, num_channels_latents, height // 2, 2, width // 2, 2)
        return x.permute(0, 2, 4, 1, 3, 5).reshape(batch_size, (height // 2) * (width // 2), num_channels_latents * 4)

    @staticmethod
    def _unpack_latents(latents, height, width, vae_scale_factor):
        b, _, ch = latents.shape
        h, w = height // vae_scale_factor, width // vae_scale_factor
        x = latents.reshape(b, h, w, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
        return x.reshape(b, ch // 4, h * 2, w * 2)

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        h2 = 2 * (int(height) // self.vae_scale_factor)
        w2 = 2 * (int(width) // self.vae_scale_factor)
        ids = self._prepare_latent_image_ids(batch_size, h2, w2, device, dtype)
        if latents is not None:
            return latents.to(device=device, dtype=dtype), ids
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        noise = randn_tensor((batch_size, num_channels_latents, h2, w2), generator=generator, device=device, dtype=dtype)
        return self._pack_latents(noise, batch_size, num_channels_latents, h2, w2), ids

    def _glyph_blend(self, image, image_latents, noise):
        """0.10·glyph latent + noise where the bilinearly down-sampled glyph mask is > 0 (PIPE:645-654). On the GPU one HIP
        kernel (mask, resize, threshold, blend: ops.glyph_blend, bit-identical resize to F.interpolate); torch ops on the CPU."""
        if image.is_cuda:
            return ops.glyph_blend(image.to(torch.float32), image_latents.to(torch.float32), noise.to(torch.float32)).to(noise.dtype)
        m = (image > 0).any(dim=1, keepdim=True).float()
        m = F.interpolate(m, size=noise.shape[-2:], mode="bilinear", align_corners=False) > 0
        return torch.where(m, 0.10 * image_latents + noise, noise)

    def prepare_latents_reptext(self, image, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        """PIPE:608-660 including quirk Q1: the glyph image IS VAE-encoded with `generator` (advancing its stream by one
        [B,16,h,w] normal draw) and the blended result is computed, but the returned latents are the plain noise."""
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
        _unused_blend = self._glyph_blend(image, image_latents, noise)   # computed and dropped, as in the reference (Q1)
        return self._pack_latents(noise, batch_size, num_channels_latents, h2, w2), ids

    def _prep_pixels(self, image, width, height, batch_size, num_images_per_prompt, device, dtype):
        if not isinstance(image, torch.Tensor):
            image = self.image_processor.preprocess(image, height=height, width=width)
        repeat_by = batch_size if image.shape[0] == 1 else num_images_per_prompt
        return image.repeat_interleave(repeat_by, dim=0).to(device=device, dtype=dtype)

    def prepare_image(self, image, width, height, batch_size, num_images_per_prompt, device, dtype, image_position=None,
                      do_classifier_free_guidance=False, guess_mode=False):
        """PIPE:663-731: VAE-encode the canny hint and the (3-channel repeated) position hint, concatenate on channels,
        pack -> [B, N, 128]. Both posteriors are sampled from the GLOBAL torch RNG (no generator; quirk Q2)."""
        image = self._prep_pixels(image, width, height, batch_size, num_images_per_prompt, device, dtype)
        pos = self._prep_pixels(image_position, width, height, batch_size, num_images_per_prompt, device, dtype).repeat(1, 3, 1, 1)
        sf, sc = self.vae.config.shift_factor, self.vae.config.scaling_factor
        lat = ((self.vae.encode(image.to(self.vae.dtype)).latent_dist.sample() - sf) * sc).to(dtype)
        plat = ((self.vae.encode(pos.to(self.vae.dtype)).latent_dist.sample() - sf) * sc).to(dtype)
        both = torch.cat([lat, plat], dim=1)
        packed = self._pack_latents(both, batch_size * num_images_per_prompt, both.shape[1], both.shape[2], both.shape[3])
        if do_classifier_free_guidance:
            packed = torch.cat([packed] * 2)
        return packed, height, width

    def _region_masks(self, control_mask, device, dtype) -> List[torch.Tensor]:
        """PIPE:1007-1013: mask/255 -> bilinear x1/16 -> [1, N, 1]. uint8 masks headed for the GPU are divided and resized
        there (ops.resize2d: same source-index rule and fp32 expression order as F.interpolate, bit-identical)."""
        out = []
        if control_mask is not None:
            for m in control_mask:
                arr = np.array(m)
                if torch.device(device).type == "cuda" and arr.dtype == np.uint8 and arr.ndim == 2:
                    t = ops.resize2d(torch.from_numpy(arr).to(device)[None, None], scale_factor=1 / 16, mode="bilinear", u8_scale=255.0)
                    out.append(t.reshape([1, -1, 1]).to(dtype))
                    continue
                rm = torch.from_numpy(arr) / 255.0
                t = F.interpolate(rm[None, None].float(), scale_factor=1 / 16, mode="bilinear").reshape([1, -1, 1])
                out.append(t.to(device=device, dtype=dtype))
        return out

    def _is_packed_hint(self, t) -> bool:
        cn = self.controlnet
        return isinstance(t, torch.Tensor) and t.dim() == 3 and isinstance(cn, FluxControlNetModel) and \
            t.shape[-1] == cn.controlnet_x_embedder.weight.shape[1]

    # ------------------------------------------------------------------ the call
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]] = None, prompt_2: Optional[Union[str, List[str]]] = None,
                 height: Optional[int] = None, width: Optional[int] = None, num_inference_steps: int = 28,
                 timesteps: List[int] = None, guidance_scale: float = 7.0,
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
                 control_glyph: Optional[torch.FloatTensor] = None):
        height = height or self.default_sample_size * self.vae_scale_factor
        width = width or self.default_sample_size * self.vae_scale_factor
        # control_guidance_start/end are normalised like the reference but have no effect on the result (inert: Q3)
        self.check_inputs(prompt, prompt_2, height, width, prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
                          callback_on_step_end_tensor_inputs=callback_on_step_end_tensor_inputs, max_sequence_length=max_sequence_length)
        self._guidance_scale, self._joint_attention_kwargs, self._interrupt = guidance_scale, joint_attention_kwargs, False

        if isinstance(prompt, str):
            batch_size = 1
        elif isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        device, dtype = self._execution_device, self.transformer.dtype
        total = batch_size * num_images_per_prompt

        prompt_embeds, pooled_prompt_embeds, text_ids = self.encode_prompt(
            prompt=prompt, prompt_2=prompt_2, prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds, device=device,
            num_images_per_prompt=num_images_per_prompt, max_sequence_length=max_sequence_length)
        prompt_embeds = prompt_embeds.to(device=device)
        pooled_prompt_embeds = pooled_prompt_embeds.to(device=device)

        # hints: one packed [B,N,128] tensor per text line (PIPE:928-942). Tensors that are already packed hint latents
        # ([B,N,in+extra]) are taken as they are — an extension used by the benchmarks and the multi-GPU broadcast.
        hints: List[torch.Tensor] = []
        if isinstance(self.controlnet, FluxControlNetModel) and control_image is not None:
            positions = control_position if control_position is not None else [None] * len(control_image)
            for img, pos in zip(control_image, positions):
                if self._is_packed_hint(img):
                    hints.append(img.to(device=device, dtype=dtype))
                else:
                    h, height, width = self.prepare_image(image=img, image_position=pos, width=width, height=height, batch_size=total,
                                                          num_images_per_prompt=num_images_per_prompt, device=device, dtype=dtype)
                    hints.append(h)

        num_channels_latents = self.transformer.config.in_channels // 4
        sigmas = np.linspace(1.0, 1 / num_inference_steps, num_inference_steps)
        image_seq_len = (int(height) // self.vae_scale_factor) * (int(width) // self.vae_scale_factor)
        sc = self.scheduler.config
        mu = calculate_shift(image_seq_len, sc.base_image_seq_len, sc.max_image_seq_len, sc.base_shift, sc.max_shift)
        timesteps, num_inference_steps = retrieve_timesteps(self.scheduler, num_inference_steps, device, timesteps, sigmas, mu=mu)

        if control_glyph is not None:
            init_image = self.image_processor.preprocess(control_glyph, height=height, width=width).to(dtype=torch.float32)
            latents, latent_image_ids = self.prepare_latents_reptext(init_image, total, num_channels_latents, height, width,
                                                                     prompt_embeds.dtype, device, generator, None)
        else:
            latents, latent_image_ids = self.prepare_latents(total, num_channels_latents, height, width, prompt_embeds.dtype, device,
                                                             generator, latents)
        self._num_timesteps = len(timesteps)
        masks = self._region_masks(control_mask, latents.device, latents.dtype)

        latents = self._denoise(latents, prompt_embeds, pooled_prompt_embeds, text_ids, latent_image_ids, timesteps, hints, masks,
                                guidance_scale, controlnet_conditioning_scale, controlnet_conditioning_step, control_mode,
                                callback_on_step_end, callback_on_step_end_tensor_inputs, num_inference_steps)

        if output_type == "latent":
            # The parity tap (PIPE:1132-1133). The loop's state is kept in fp32 (A.6: the scheduler steps in fp32), and that state
            # is what is returned: rounding it to bf16 here would by itself cost 1.8e-3 rel-L2, twice the whole loop's error.
            # `.to(torch.bfloat16)` gives the reference's bf16-run dtype.
            image = self._master_latents
        else:
            h2 = 2 * (int(height) // self.vae_scale_factor)
            w2 = 2 * (int(width) // self.vae_scale_factor)
            if output_type in ("pil", "np"):
                u8 = self.vae.decode_packed(latents, h2, w2, output_u8=True)
                image = self.image_processor.postprocess_u8(u8, output_type)
            elif output_type == "pt":
                image = (self.vae.decode_packed(latents, h2, w2) / 2 + 0.5).clamp(0, 1)
            else:
                raise ValueError(f"unsupported output_type {output_type}")
        self.maybe_free_model_hooks()
        if not return_dict:
            return (image,)
        return FluxPipelineOutput(images=image)

    reference_bf16_scalars = False     # True: round t, t/1000 and guidance to bf16 where the reference's bf16 run does (mmdit.py)

    def _model_timestep(self, t: float) -> float:
        """The value the models receive as `timestep` (PIPE:1025,1048: t.to(dtype) / 1000). fp32-exact by default; under
        `reference_bf16_scalars` the bf16 run's value bf16(bf16(t) / 1000)."""
        from . import mmdit

        mmdit.reference_bf16_scalars(self.reference_bf16_scalars)
        if not self.reference_bf16_scalars:
            return t / 1000.0
        return float((torch.tensor(t, dtype=torch.float32).to(torch.bfloat16) / 1000).to(torch.float32))

    # ------------------------------------------------------------------ hot loop (PIPE:1016-1130)
    def _denoise(self, latents, prompt_embeds, pooled, text_ids, image_ids, timesteps, hints, masks, guidance_scale,
                 cn_scale, cn_steps, control_mode, callback, callback_inputs, num_inference_steps):
        """Eager loop, or the replay of its captured hipGraph when this exact call signature has been seen before (GRAPH_CAPTURE)."""
        tvals = timesteps.to(torch.float32).cpu().tolist()                 # host copies: no per-step device sync
        use_graph = (GRAPH_CAPTURE and getattr(self, "capture_graphs", True) and callback is None and latents.is_cuda
                     and not self.interrupt and isinstance(self.controlnet, (FluxControlNetModel, type(None)))
                     and (control_mode is None) and self.joint_attention_kwargs is None)
        if not use_graph:
            return self._denoise_eager(latents, prompt_embeds, pooled, text_ids, image_ids, tvals, hints, masks, guidance_scale, cn_scale,
                                       cn_steps, control_mode, callback, callback_inputs, num_inference_steps, timesteps)
        from . import mmdit as _mm
        _mm.reference_bf16_scalars(self.reference_bf16_scalars)      # the module switch follows THIS pipeline before the key is built
        sig = lambda t: (tuple(t.shape), str(t.dtype), tuple(t.stride()))
        models = tuple((id(m), id(m._ensure_plans()), id(getattr(m, "_cx_pad", None)), bool(getattr(m, "_fp8_linears", False)),
                        bool(getattr(m, "_fp8_attention", False))) for m in (self.transformer, self.controlnet) if m is not None)
        key = (sig(latents), sig(prompt_embeds), sig(pooled), sig(text_ids), sig(image_ids), tuple(sig(h) for h in hints), tuple(sig(m) for m in masks),
               tuple(tvals), tuple(self.scheduler.sigmas.tolist()), float(guidance_scale), repr(cn_scale), int(cn_steps), int(num_inference_steps),
               str(latents.device), models, bool(self.reference_bf16_scalars), bool(_mm.RESIDUAL_F32), bool(OVERLAP_TOWER))
        cache = self.__dict__.setdefault("_graph_cache", {})
        ent = cache.get(key)
        if ent is None:                                      # first sight of this signature: eager (and warm), remember it
            if len(cache) >= GRAPH_CACHE_MAX:
                cache.pop(next(iter(cache)))
            cache[key] = "seen"
            return self._denoise_eager(latents, prompt_embeds, pooled, text_ids, image_ids, tvals, hints, masks, guidance_scale, cn_scale,
                                       cn_steps, control_mode, None, callback_inputs, num_inference_steps, timesteps)
        if ent == "failed":
            return self._denoise_eager(latents, prompt_embeds, pooled, text_ids, image_ids, tvals, hints, masks, guidance_scale, cn_scale,
                                       cn_steps, control_mode, None, callback_inputs, num_inference_steps, timesteps)
        ins = [latents, prompt_embeds, pooled, text_ids, image_ids] + list(hints) + list(masks)
        if ent == "seen":                                    # second call: capture
            static = [t.clone() for t in ins]
            nh = len(hints)
            keep = [m._ensure_plans() for m in (self.transformer, self.controlnet) if m is not None] + [getattr(self.controlnet, "_cx_pad", None)]
            graph = torch.cuda.CUDAGraph()
            step_index = self.scheduler._step_index
            # The graph bakes in the device pointers of every buffer the loop touches. Buffers that live in caches which may evict or
            # replace them later (activation workspaces, attention key-split workspaces, tower sample buffers, rope tables) are
            # recorded while the capture runs and OWNED by the graph's entry: a replay can never write into memory that was handed
            # back to the allocator (tests: test_graphs_of_two_shapes_keep_their_buffers).
            attn_keys = set(ops._ATTN_WS)
            _mm.CAPTURE_KEEP, ops.CAPTURE_KEEP = keep, keep
            try:
                # thread-local capture mode: calls made by OTHER threads (RCCL's watchdog polling its events) do not invalidate the capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    out = self._denoise_eager(static[0], static[1], static[2], static[3], static[4], tvals, static[5 : 5 + nh], static[5 + nh :],
                                              guidance_scale, cn_scale, cn_steps, control_mode, None, callback_inputs, num_inference_steps, timesteps,
                                              _quiet=True)
                    out32 = self._master_latents
            except Exception as e:                           # capture is an optimisation, never a requirement
                import sys
                print(f"[reptext_amd] hipGraph capture of the denoise loop failed ({type(e).__name__}: {e}); staying eager", file=sys.stderr, flush=True)
                cache[key] = "failed"
                _mm.CAPTURE_KEEP = ops.CAPTURE_KEEP = None
                ops.drop_attention_workspaces(set(ops._ATTN_WS) - attn_keys)     # their zero fill was recorded, never executed
                torch.cuda.synchronize()
                self.scheduler._step_index = step_index
                return self._denoise_eager(latents, prompt_embeds, pooled, text_ids, image_ids, tvals, hints, masks, guidance_scale, cn_scale,
                                           cn_steps, control_mode, None, callback_inputs, num_inference_steps, timesteps)
            _mm.CAPTURE_KEEP = ops.CAPTURE_KEEP = None
            keep.append(getattr(self, "_sample_cache", None))
            keep.extend(dict(m._rope_cache) for m in (self.transformer, self.controlnet) if m is not None and hasattr(m, "_rope_cache"))
            self.scheduler._step_index = step_index
            ent = cache[key] = {"graph": graph, "static": static, "out": out, "out32": out32, "keep": keep}
        for dst, src in zip(ent["static"], ins):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        ent["graph"].replay()
        self.scheduler._step_index = (self.scheduler._step_index or 0) + len(tvals)      # what the eager loop's step() calls leave behind
        self._master_latents = ent["out32"].clone()
        with self.progress_bar(total=num_inference_steps) as bar:
            bar.update(num_inference_steps)
        return ent["out"].clone()

    def _denoise_eager(self, latents, prompt_embeds, pooled, text_ids, image_ids, tvals, hints, masks, guidance_scale,
                       cn_scale, cn_steps, control_mode, callback, callback_inputs, num_inference_steps, timesteps, _quiet=False):
        device = latents.device
        B = latents.shape[0]
        guidance = torch.full((B,), float(guidance_scale), device=device, dtype=torch.float32) if self.transformer.config.guidance_embeds else None
        # one regional mask per text line, shared by the batch ([1,N,1], the reference's form) or one per image ([B,N,1])
        rowscales = [m.to(torch.float32).reshape(-1).contiguous() if m.shape[0] == 1 else m.to(torch.float32).reshape(m.shape[0], -1).contiguous() for m in masks]
        num_warmup = max(len(timesteps) - num_inference_steps * self.scheduler.order, 0)
        # adaLN vectors of every block for every step, once per image (timesteps/guidance/pooled are loop-invariant inputs)
        model_ts = [self._model_timestep(t) for t in tvals]
        tab_t = self.transformer.build_modulation_table(model_ts, guidance, pooled)
        fused_cn = isinstance(self.controlnet, FluxControlNetModel) and len(hints) > 0
        tab_c = self.controlnet.build_modulation_table(model_ts[: max(0, min(len(model_ts), cn_steps))], guidance, pooled) if fused_cn and cn_steps > 0 else None
        # Loop-invariant work, once per image instead of once per step (the prompt and the hint latents do not change inside the
        # loop): context_embedder(prompt) of both models, controlnet_x_embedder(hint) per text line. A callback that replaces
        # prompt_embeds invalidates them (recomputed below).
        static_t = self.transformer.prepare_static(prompt_embeds)
        static_c = [self.controlnet.prepare_static(prompt_embeds, h) for h in hints] if fused_cn and cn_steps > 0 else []
        # Which tower samples does the transformer read? Block i takes sample i // ceil(n_blocks / n_samples) (A.3): with 6
        # samples against 19 double blocks the sixth is never consumed (Q5), so its block and zero-linear are not evaluated.
        blocks_needed, sample_buf, single_buf = None, None, None
        if fused_cn:
            cnet = self.controlnet
            n_cd, n_cs = len(cnet.transformer_blocks), len(cnet.single_transformer_blocks)
            n_td, n_ts = len(self.transformer.transformer_blocks), len(self.transformer.single_transformer_blocks)
            need_d = 0 if n_cd == 0 else (n_td - 1) // int(np.ceil(n_td / n_cd)) + 1
            need_s = 0 if n_cs == 0 or n_ts == 0 else (n_ts - 1) // int(np.ceil(n_ts / n_cs)) + 1
            blocks_needed = (min(need_d, n_cd), min(need_s, n_cs))
            # sample buffers: allocated once per shape, written by the zero-linear epilogues every step (no per-step allocation)
            Bc, N_, d_ = prompt_embeds.shape[0], latents.shape[1], cnet.inner_dim
            key = (Bc, N_, d_, n_cd, n_cs, str(device))
            if getattr(self, "_sample_cache", None) is None or self._sample_cache[0] != key:
                mk = lambda n: [torch.empty(Bc, N_, d_, device=device, dtype=torch.bfloat16) for _ in range(n)]
                self._sample_cache = (key, mk(n_cd), mk(n_cs))
            sample_buf, single_buf = self._sample_cache[1], self._sample_cache[2]
        # fp32 master copy of the latents between steps (the models read its bf16 copy): the scheduler computes in fp32 anyway
        # (A.6); not rounding the STATE 28 times keeps the loop close to the fp32 reference path. Callbacks see the bf16 copy.
        lat32 = latents.to(torch.float32).contiguous()
        latents = latents.to(torch.bfloat16).contiguous().clone()
        # Tower ∥ transformer: within a step the tower's chain of kernels and the transformer's are independent except that
        # transformer block i consumes tower sample i // 4 (A.3). The tower therefore runs on a side stream into preallocated
        # sample buffers and the transformer waits, block by block, on the event of the sample it needs. At batch 1 most
        # launches fill only 27/32 of their last round of workgroups (216 GEMM tiles on 256 CUs, 864 attention workgroups on
        # 512 slots); two independent chains in flight fill some of those holes. Results are bitwise those of the serial order.
        # On by default (OVERLAP_TOWER; RT_OVERLAP_TOWER=0 turns it off): -0.5 % eager, -1.1 % inside the captured graph.
        overlap = (OVERLAP_TOWER and fused_cn and tab_c is not None and device.type == "cuda"
                   and len(self.controlnet.single_transformer_blocks) == 0)
        if overlap:
            if getattr(self, "_side_stream", None) is None or self._side_stream.device != device:
                self._side_stream = torch.cuda.Stream(device=device)
            side = self._side_stream
            sample_ev = [torch.cuda.Event() for _ in range(len(self.controlnet.transformer_blocks))]
        with (_Progress(num_inference_steps, disable=True) if _quiet else self.progress_bar(total=num_inference_steps)) as bar:
            for i, t in enumerate(tvals):
                if self.interrupt:
                    continue
                timestep = torch.full((B,), self._model_timestep(t), device=device, dtype=torch.float32)      # PIPE:1025,1048 (Q4)
                merged = merged_single = None
                events = None
                if overlap and i < cn_steps:
                    main = torch.cuda.current_stream()
                    side.wait_stream(main)                       # latents of this step (and, at i = 0, tables and hints) are ready
                    with torch.cuda.stream(side):
                        for line, hint in enumerate(hints):
                            self.controlnet(
                                hidden_states=latents, controlnet_cond=hint, controlnet_mode=control_mode, conditioning_scale=cn_scale,
                                timestep=timestep, guidance=guidance, pooled_projections=pooled, encoder_hidden_states=prompt_embeds,
                                txt_ids=text_ids, img_ids=image_ids, joint_attention_kwargs=self.joint_attention_kwargs,
                                return_dict=False, _rowscale=rowscales[line] if rowscales else None, _accumulate_into=sample_buf,
                                _overwrite=(line == 0), _sample_events=sample_ev if line == len(hints) - 1 else None,
                                _mods=tab_c.step(i), _ws_tag="tower", _static=static_c[line], _blocks_needed=blocks_needed)
                    merged, events = [b if k < blocks_needed[0] else None for k, b in enumerate(sample_buf)], sample_ev
                for line, hint in enumerate(hints if events is None else ()):
                    if i >= cn_steps:                                                             # Q3
                        samples = single_samples = None
                    elif fused_cn:
                        # zero-linear epilogues write (line 0) or add to (later lines: PIPE:1076-1087) the preallocated buffers
                        samples, single_samples = self.controlnet(
                            hidden_states=latents, controlnet_cond=hint, controlnet_mode=control_mode, conditioning_scale=cn_scale,
                            timestep=timestep, guidance=guidance, pooled_projections=pooled, encoder_hidden_states=prompt_embeds,
                            txt_ids=text_ids, img_ids=image_ids, joint_attention_kwargs=self.joint_attention_kwargs,
                            return_dict=False, _rowscale=rowscales[line] if rowscales else None, _accumulate_into=sample_buf,
                            _accumulate_single_into=single_buf, _overwrite=(line == 0), _mods=None if tab_c is None else tab_c.step(i),
                            _static=static_c[line] if static_c else None, _blocks_needed=blocks_needed)
                    else:
                        rs = rowscales[line] if rowscales else None
                        samples, single_samples = self.controlnet(
                            hidden_states=latents, controlnet_cond=hint, controlnet_mode=control_mode, conditioning_scale=cn_scale,
                            timestep=timestep, guidance=guidance, pooled_projections=pooled, encoder_hidden_states=prompt_embeds,
                            txt_ids=text_ids, img_ids=image_ids, joint_attention_kwargs=self.joint_attention_kwargs,
                            return_dict=False, _rowscale=rs, _accumulate_into=merged if line > 0 else None,
                            _accumulate_single_into=merged_single if line > 0 else None,
                            _mods=None if tab_c is None else tab_c.step(i))
                    if line == 0:
                        merged, merged_single = samples, single_samples
                    # line > 0: the zero-linear epilogues already summed into `merged` (PIPE:1076-1087)
                noise_pred = self.transformer(
                    hidden_states=latents, timestep=timestep, guidance=guidance, pooled_projections=pooled,
                    encoder_hidden_states=prompt_embeds, controlnet_block_samples=merged, controlnet_single_block_samples=merged_single,
                    txt_ids=text_ids, img_ids=image_ids, joint_attention_kwargs=self.joint_attention_kwargs, return_dict=False,
                    _mods=tab_t.step(i), _sample_events=events, _static=static_t)[0]
                if events is not None:
                    torch.cuda.current_stream().wait_stream(side)     # the tower has finished reading `latents` (its last sample is unused, Q5)
                self.scheduler.step_master_(noise_pred, lat32, latents)
                if callback is not None:
                    env = {"latents": latents, "prompt_embeds": prompt_embeds}
                    out = callback(self, i, timesteps[i], {k: env[k] for k in callback_inputs})
                    if "latents" in out:
                        latents = out.pop("latents").to(torch.bfloat16).contiguous()
                        lat32 = latents.to(torch.float32)
                    if "prompt_embeds" in out:                       # the loop-invariant embeddings are no longer valid
                        prompt_embeds = out.pop("prompt_embeds")
                        static_t = self.transformer.prepare_static(prompt_embeds)
                        static_c = [self.controlnet.prepare_static(prompt_embeds, h) for h in hints] if static_c else []
                if i == len(tvals) - 1 or ((i + 1) > num_warmup and (i + 1) % self.scheduler.order == 0):
                    bar.update()
        self._master_latents = lat32          # fp32 state of the loop; `latents` is its bf16 copy
        return latents
