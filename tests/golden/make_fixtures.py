"""Regenerates the fixtures in this directory.  python tests/golden/make_fixtures.py

What they are — and are not:
  known_answers.json   closed-form values of SURVEY.md §8c (time-shift mu, the two sigma schedules, the 19-over-6 interval map,
                       parameter counts). They follow from formulas the reference states (PIPE:78-88, 948-967; CN:45-116), not
                       from running it.
  oracle_small.safetensors
                       inputs and fp32 outputs of THIS repository's CPU oracle (oracle/flux_oracle.py) on a seeded
                       reduced-depth model. NOT reference outputs: the reference cannot be imported here (no diffusers,
                       SURVEY.md §8c) and ships no tensors, so parity stays unpinned. The fixture is a regression pin: the
                       oracle's default (fp32) path must keep producing exactly these numbers while emulation switches
                       (stored_as, fp8_linears, fp8_attention) are added around it, and GPU tests can read expected values
                       without trusting a freshly edited oracle.
"""
import json
import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import flux_oracle as orc  # noqa: E402

SMALL_T = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=2,
               joint_attention_dim=128, pooled_projection_dim=32, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
SMALL_CN = dict(SMALL_T, num_layers=2, num_single_layers=0, extra_condition_channels=64)


def known_answers():
    return {
        "source": "SURVEY.md §8c (closed forms; fp32 evaluation of A.6 for the schedules)",
        "calculate_shift": {"4096": 1.15, "256": 0.5, "9216": 2.01667, "default_max_shift_4096": 1.16},
        "sigmas_c1_2steps_mu0.5": [1.0, 0.622459, 0.0],
        "sigmas_c2_28steps_mu1.15": [1.0, 0.988409, 0.976222, 0.963394, 0.949873, 0.935599, 0.920509, 0.904531, 0.887583, 0.869576,
                                     0.850406, 0.829956, 0.808096, 0.784672, 0.759511, 0.732413, 0.703145, 0.671435, 0.636964,
                                     0.599357, 0.558163, 0.512844, 0.462748, 0.407078, 0.344849, 0.274828, 0.195455, 0.104721, 0.0],
        "interval_map_19_over_6": [0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4],
        "parameter_counts": {"double_block": 339831296, "single_block": 141591808, "flux_dev_transformer": 11901408320,
                             "reptext_controlnet_billion": 2.1411},
    }


def oracle_small():
    g = torch.Generator().manual_seed(2024)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    tp = orc.init_mmdit_params(SMALL_T, seed=5)
    cp = orc.init_mmdit_params(SMALL_CN, seed=6, controlnet=True)
    B, T, h2, w2 = 1, 16, 8, 8
    N = (h2 // 2) * (w2 // 2)
    x = dict(latents=r(B, N, 64), cond=r(B, N, 128), prompt=r(B, T, 128), pooled=r(B, 32))
    ids, tids = orc.latent_image_ids(h2, w2), torch.zeros(T, 3)
    ts, gd = torch.full((B,), 0.622459), torch.full((B,), 3.5)
    samples, _ = orc.controlnet_forward(cp, SMALL_CN, x["latents"], x["cond"], x["prompt"], x["pooled"], ts, ids, tids, guidance=gd, conditioning_scale=0.8)
    vel = orc.transformer_forward(tp, SMALL_T, x["latents"], x["prompt"], x["pooled"], ts, ids, tids, guidance=gd, controlnet_block_samples=samples)
    sig = orc.flow_sigmas(3, orc.calculate_shift(N, 256, 4096, 0.5, 1.15))
    mask = (torch.arange(N) % 3 != 0).float().reshape(1, N, 1)
    lat = orc.denoise_loop(tp, SMALL_T, cp, SMALL_CN, x["latents"], x["prompt"], x["pooled"], [x["cond"]], [mask], sig, ids, tids, 3.5,
                           conditioning_scale=0.8, conditioning_step=2)
    out = {f"in.{k}": v for k, v in x.items()}
    out.update({"in.mask": mask, "in.sigmas": sig, "out.velocity": vel, "out.latents_3steps": lat})
    out.update({f"out.controlnet_sample.{i}": s for i, s in enumerate(samples)})
    return {k: v.contiguous() for k, v in out.items()}


if __name__ == "__main__":
    json.dump(known_answers(), open(os.path.join(HERE, "known_answers.json"), "w"), indent=1)
    save_file(oracle_small(), os.path.join(HERE, "oracle_small.safetensors"),
              metadata={"what": "outputs of this repository's fp32 CPU oracle on a seeded reduced model; NOT reference outputs (parity unpinned)"})
    print("written", os.listdir(HERE))
