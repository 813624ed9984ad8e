"""tests/golden/: closed-form known answers and a regression fixture of this repository's own fp32 oracle
(see tests/golden/make_fixtures.py for what the fixture is and is not — it is NOT a reference output; parity stays unpinned)."""
import json
import os
import sys

import pytest
import torch
from safetensors.torch import load_file

from oracle import flux_oracle as orc

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, HERE)
import make_fixtures as mk  # noqa: E402


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_known_answers_fixture_matches_oracle():
    ka = json.load(open(os.path.join(HERE, "known_answers.json")))
    assert abs(orc.calculate_shift(4096, 256, 4096, 0.5, 1.15) - ka["calculate_shift"]["4096"]) < 1e-12
    assert abs(orc.calculate_shift(256, 256, 4096, 0.5, 1.15) - ka["calculate_shift"]["256"]) < 1e-12
    assert abs(orc.calculate_shift(9216, 256, 4096, 0.5, 1.15) - ka["calculate_shift"]["9216"]) < 1e-5
    assert abs(orc.calculate_shift(4096) - ka["calculate_shift"]["default_max_shift_4096"]) < 1e-12
    assert torch.allclose(orc.flow_sigmas(2, 0.5), torch.tensor(ka["sigmas_c1_2steps_mu0.5"]), atol=1e-6)
    assert torch.allclose(orc.flow_sigmas(28, 1.15), torch.tensor(ka["sigmas_c2_28steps_mu1.15"]), atol=1.5e-6)
    assert orc.interval_map(19, 6) == ka["interval_map_19_over_6"]
    tp = orc.init_mmdit_params(dict(orc.FLUX_DEV_CFG, num_layers=1, num_single_layers=1), seed=0, round_bf16=False)
    n_double = sum(v.numel() for k, v in tp.items() if k.startswith("transformer_blocks.0."))
    n_single = sum(v.numel() for k, v in tp.items() if k.startswith("single_transformer_blocks.0."))
    assert n_double == ka["parameter_counts"]["double_block"] and n_single == ka["parameter_counts"]["single_block"]


def test_oracle_default_path_reproduces_its_fixture():
    """The fp32 path of the oracle must be untouched by the emulation switches living in the same functions."""
    fx = load_file(os.path.join(HERE, "oracle_small.safetensors"))
    now = mk.oracle_small()
    assert set(now) == set(fx)
    for k in fx:
        assert torch.allclose(now[k], fx[k], rtol=1e-5, atol=1e-6), k


@pytest.mark.gpu
def test_gpu_models_against_committed_fixture(gpu):
    """The HIP path on the fixture's inputs against the fixture's COMMITTED expected outputs (no oracle call at test time)."""
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    fx = load_file(os.path.join(HERE, "oracle_small.safetensors"))
    tp = orc.init_mmdit_params(mk.SMALL_T, seed=5)          # weights only; outputs come from the file
    cp = orc.init_mmdit_params(mk.SMALL_CN, seed=6, controlnet=True)
    tr = FluxTransformer2DModel(**mk.SMALL_T, device=gpu, dtype=torch.bfloat16)
    cn = FluxControlNetModel(**mk.SMALL_CN, device=gpu, dtype=torch.bfloat16)
    tr.load_state_dict(tp); cn.load_state_dict(cp)
    b16 = lambda t: t.to(gpu, torch.bfloat16)
    B, T, h2, w2 = 1, 16, 8, 8
    ids, tids = b16(orc.latent_image_ids(h2, w2)), b16(torch.zeros(T, 3))
    ts, gd = torch.full((B,), 0.622459, device=gpu), torch.full((B,), 3.5, device=gpu)
    kw = dict(encoder_hidden_states=b16(fx["in.prompt"]), pooled_projections=b16(fx["in.pooled"]), timestep=ts, img_ids=ids, txt_ids=tids, guidance=gd)
    samples, _ = cn(hidden_states=b16(fx["in.latents"]), controlnet_cond=b16(fx["in.cond"]), conditioning_scale=0.8, return_dict=False, **kw)
    for i, s in enumerate(samples):
        assert rel_l2(s.float().cpu(), fx[f"out.controlnet_sample.{i}"]) < 6e-3
    vel = tr(hidden_states=b16(fx["in.latents"]), controlnet_block_samples=samples, return_dict=False, **kw)[0]
    e_v = rel_l2(vel.float().cpu(), fx["out.velocity"])
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), None, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    # the per-token mask of the fixture as the pixel-resolution image the API takes: constant 16x16 blocks survive the
    # bilinear x1/16 of PIPE:1010 exactly
    import numpy as np
    tok = fx["in.mask"].reshape(h2 // 2, w2 // 2)
    mask_px = (tok.repeat_interleave(16, 0).repeat_interleave(16, 1) * 255).to(torch.uint8).numpy()
    lat = pipe(prompt_embeds=b16(fx["in.prompt"]), pooled_prompt_embeds=b16(fx["in.pooled"]), height=16 * h2 // 2, width=16 * w2 // 2,
               num_inference_steps=3, guidance_scale=3.5, control_image=[b16(fx["in.cond"])], control_mask=[np.ascontiguousarray(mask_px)],
               controlnet_conditioning_scale=0.8, controlnet_conditioning_step=2, latents=b16(fx["in.latents"]), output_type="latent").images
    e_l = rel_l2(lat.float().cpu(), fx["out.latents_3steps"])
    print(f"vs committed fixture: velocity {e_v:.3e}, 3-step latents {e_l:.3e}")
    assert e_v < 6e-3 and e_l < 2e-3
