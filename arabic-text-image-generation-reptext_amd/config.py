"""Model / scheduler configurations of the path (FLUX.1-dev, RepText ControlNet, FLUX VAE).

Values: SURVEY.md Appendix A.3/A.6/A.7 and the ctor defaults at controlnet_flux.py:45-60. A ``Config`` is a dict
with attribute access, matching how the reference reads ``transformer.config.guidance_embeds`` (PIPE:1028),
``vae.config.scaling_factor`` (PIPE:1137) and ``scheduler.config.base_image_seq_len`` (PIPE:954).
"""
from __future__ import annotations

import json
import os


class Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @classmethod
    def from_json_file(cls, path: str) -> "Config":
        with open(path) as f:
            return cls({k: v for k, v in json.load(f).items() if not k.startswith("_")})

    def save_json(self, path: str, class_name: str) -> None:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump({"_class_name": class_name, **self}, f, indent=2)


def flux_dev_transformer_config(**over) -> Config:
    c = Config(patch_size=1, in_channels=64, out_channels=None, num_layers=19, num_single_layers=38,
               attention_head_dim=128, num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768,
               guidance_embeds=True, axes_dims_rope=[16, 56, 56])
    c.update(over)
    return c


def reptext_controlnet_config(**over) -> Config:
    """Depth inferred from the 4.28 GB checkpoint (SURVEY.md header fact 3): 6 double, 0 single, guidance embeds."""
    c = Config(patch_size=1, in_channels=64, num_layers=6, num_single_layers=0, attention_head_dim=128,
               num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True,
               axes_dims_rope=[16, 56, 56], num_mode=None, extra_conditioning_channels=0, extra_condition_channels=64)
    c.update(over)
    return c


def flux_vae_config(**over) -> Config:
    c = Config(in_channels=3, out_channels=3, latent_channels=16, block_out_channels=[128, 256, 512, 512],
               layers_per_block=2, norm_num_groups=32, act_fn="silu", scaling_factor=0.3611, shift_factor=0.1159,
               use_quant_conv=False, use_post_quant_conv=False, mid_block_add_attention=True, force_upcast=True)
    c.update(over)
    return c


def flux_scheduler_config(**over) -> Config:
    c = Config(num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=1.15,
               base_image_seq_len=256, max_image_seq_len=4096)
    c.update(over)
    return c
