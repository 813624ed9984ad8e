"""GPU checks of the BASELINE configurations at their own shapes ("configs not exercised on hardware"):

  C1  256x256, 2 steps, masked tower at the REAL width (d = 3072, H = 24, joint 4096; reduced depth so that the fp32 CPU
      oracle finishes in seconds): pipeline latents within the north-star 1e-3 of the CPU path, and on the bf16-storage floor;
  C3  the 4-images-per-GPU shard: batch invariance of a full-size double + single block at B = 4;
  C4  infer_inpaint.py's flow at 1024x1024 (internal batch 2 under CFG, 68-channel inpaint tower, S = 4608);
  C5  shape checks of the bf16 kernels live in test_fullsize_gpu.py; here the e4m3 attention at S = 9728.

plus the loop-invariant hoisting, the unused-tower-block skip, derived-weight cache invalidation and the on-disk formats.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import flux_oracle as orc  # noqa: E402
from oracle import vae_oracle as vorc  # noqa: E402
from test_models_gpu import assert_at_dtype_floor  # noqa: E402


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


WIDE_T = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=24,
              joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
WIDE_CN = dict(WIDE_T, num_layers=1, num_single_layers=0, extra_condition_channels=64)
VAE_SMALL = dict(vorc.FLUX_VAE_CFG, block_out_channels=(64, 128, 256, 256))


def _wide_models(gpu, seed_t=41, seed_c=42):
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.transformer import FluxTransformer2DModel

    tp = orc.init_mmdit_params(WIDE_T, seed=seed_t)
    cp = orc.init_mmdit_params(WIDE_CN, seed=seed_c, controlnet=True)
    tr = FluxTransformer2DModel(**WIDE_T, device=gpu, dtype=torch.bfloat16)
    cn = FluxControlNetModel(**WIDE_CN, device=gpu, dtype=torch.bfloat16)
    tr.load_state_dict(tp)
    cn.load_state_dict(cp)
    return tp, cp, tr, cn


def test_c1_pipeline_at_real_width(gpu):
    """BASELINE config 1 shape (256x256, 2 steps, T = 512 text tokens, one masked glyph line) with FLUX-dev-shaped weights at
    the real width — d = 3072, 24 heads, joint dim 4096, pooled 768; depth 2+2 (tower 1+0) so that the fp32 oracle runs in
    seconds. Latents at the parity tap (output_type='latent', PIPE:1132-1133) against oracle.denoise_loop.

    What the width does to the tolerance: at d = 3072 the oracle ITSELF, rounding to bf16 exactly where the HIP path stores
    bf16 (MFMA operands: LayerNorm output, q/k/v, softmax numerators, attention output, GELU hidden), sits 2.5e-3 from its
    own fp32 run — with N(0, 0.02^2) weights the four blocks move the latents by |dx|/|x| = 1.08 and a bf16 operand costs
    2^-9 per product chain. Keeping the velocity in fp32 as well would only move that floor to 2.3e-3 (measured with the
    oracle, tools/check_pipeline_floor.py). The north-star figure of 1e-3 is therefore reachable at the reduced width
    (test_pipeline_c1_latents_and_image: 8.4e-4) and not at this one with bf16 MFMA operands; what IS asserted here is that
    the GPU adds nothing to the floor (2.47e-3 against 2.48e-3 measured) — a logic error would, dtype noise cannot."""
    from PIL import Image

    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler

    tp, cp, tr, cn = _wide_models(gpu)
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), None, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    H = W = 256
    N, T = 256, 512
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    pe, pooled, hint = r(1, T, 4096), r(1, 768), r(1, N, 128)
    lat0 = orc.pack_latents(r(1, 16, 32, 32))
    mask_np = np.zeros([H, W], dtype=np.uint8)
    mask_np[60:140, 80:200] = 255
    rm = torch.nn.functional.interpolate(torch.from_numpy(mask_np)[None, None].float() / 255.0, scale_factor=1 / 16, mode="bilinear").reshape(1, -1, 1)
    sig = orc.flow_sigmas(2, orc.calculate_shift(N, 256, 4096, 0.5, 1.15))
    ids, tids = orc.latent_image_ids(32, 32), torch.zeros(T, 3)
    ref = orc.denoise_loop(tp, WIDE_T, cp, WIDE_CN, lat0, pe, pooled, [hint], [rm], sig, ids, tids, 3.5)
    out = pipe(prompt_embeds=pe.to(gpu, torch.bfloat16), pooled_prompt_embeds=pooled.to(gpu, torch.bfloat16), height=H, width=W,
               num_inference_steps=2, guidance_scale=3.5, control_image=[hint.to(gpu, torch.bfloat16)], control_mask=[Image.fromarray(mask_np)],
               controlnet_conditioning_scale=1.0, controlnet_conditioning_step=30, latents=lat0.to(gpu, torch.bfloat16),
               output_type="latent").images
    err = rel_l2(out.float().cpu(), ref)
    with orc.stored_as(torch.bfloat16):
        ref16 = orc.denoise_loop(tp, WIDE_T, cp, WIDE_CN, lat0, pe, pooled, [hint], [rm], sig, ids, tids, 3.5)
    err16, floor = rel_l2(out.float().cpu(), ref16), rel_l2(ref16, ref)
    print(f"C1 at real width: latents rel-L2 {err:.3e} vs fp32 oracle, {err16:.3e} vs bf16-storage oracle (floor {floor:.3e})")
    assert out.dtype == torch.float32
    assert err < 3.2e-3                            # measured 2.47e-3; the bf16-operand floor at this width is 2.48e-3 (see above)
    assert_at_dtype_floor(err, err16, floor)
    assert floor > 1.5e-3                          # the statement above stays checked: the floor itself is what exceeds 1e-3


def test_c1_full_depth_19_38_tower_6_0_against_oracle(gpu):
    """BASELINE config 1 with FLUX-dev-shaped weights AT FULL DEPTH — 19 double + 38 single transformer blocks, RepText tower
    6+0, d = 3072, 24 heads, joint 4096, T = 512 — 256x256, 2 steps, one masked text line, latents at the parity tap against
    oracle.denoise_loop (PIPE:1016-1130, infer.py:27-33). The only test in which every block index exists: the 19<->6 interval map
    at blocks 16-18 (sample 4; sample 5 never read, Q5), all 38 single-block plans, all 57+5 ModulationTable rows.

    Weights are random-init ON THE GPU (28 GB bf16, the bench's init) and streamed to the oracle one tensor at a time
    (oracle/streamed.py): host memory holds one fp32 matrix at once instead of 56 GB. Two oracle passes: fp32, and bf16-storage
    (the dtype floor of this graph). Asserted: the GPU is no further from the fp32 oracle than the CPU run at the same storage
    precision is, and as close to that run as two such runs get (assert_at_dtype_floor) — a wrong block index, plan or table row
    adds O(1), dtype noise cannot. The measured numbers are printed and recorded in DESIGN.md §4 / BASELINE.md §4."""
    import time

    from oracle.streamed import StreamedParams, config1_case, config1_gpu, config1_oracle
    from reptext_amd.config import flux_dev_transformer_config, reptext_controlnet_config
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    cfg_t, cfg_c = flux_dev_transformer_config(), reptext_controlnet_config()
    assert (cfg_t["num_layers"], cfg_t["num_single_layers"], cfg_c["num_layers"], cfg_c["num_single_layers"]) == (19, 38, 6, 0)
    tr = FluxTransformer2DModel(**cfg_t, device=gpu, dtype=torch.bfloat16).random_init_(seed=0)
    cn = FluxControlNetModel(**cfg_c, device=gpu, dtype=torch.bfloat16).random_init_(seed=1)
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), None, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    case = config1_case()
    out = config1_gpu(pipe, case, gpu).float().cpu()
    again = config1_gpu(pipe, case, gpu).float().cpu()
    assert out.dtype == torch.float32 and bool(torch.isfinite(out).all()) and torch.equal(out, again)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    tp, cp = StreamedParams(tr.state_dict()), StreamedParams(cn.state_dict())
    t0 = time.perf_counter()
    ref = config1_oracle(tp, cfg_t, cp, cfg_c, case)
    t1 = time.perf_counter()
    ref16 = config1_oracle(tp, cfg_t, cp, cfg_c, case, torch.bfloat16)
    t2 = time.perf_counter()
    err, err16, floor = rel_l2(out, ref), rel_l2(out, ref16), rel_l2(ref16, ref)
    moved = rel_l2(ref, case["latents"])
    print(f"C1 at FULL depth 19+38 / 6+0, d=3072: latents rel-L2 {err:.3e} vs fp32 oracle, {err16:.3e} vs bf16-storage oracle, floor {floor:.3e}; "
          f"loop moved the latents by {moved:.2f}; oracle {t1 - t0:.0f} s + {t2 - t1:.0f} s, {tp.bytes_streamed / 1e9:.0f}+{cp.bytes_streamed / 1e9:.0f} GB streamed")
    assert moved > 0.05                             # the stack does something
    assert_at_dtype_floor(err, err16, floor)
    del tr, cn, pipe
    torch.cuda.empty_cache()


def test_hoisted_embeddings_and_block_skip_are_exact(gpu):
    """The per-image hoisting (context_embedder, controlnet_x_embedder evaluated once: StaticEmbeds) and the skipped unused
    tower block give bit-identical samples / velocity to the per-step evaluation the reference does (CN:277-292, Q5)."""
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.transformer import FluxTransformer2DModel

    cfg_t = dict(WIDE_T, num_attention_heads=4, joint_attention_dim=256, pooled_projection_dim=64, num_layers=5, num_single_layers=1)
    cfg_c = dict(cfg_t, num_layers=3, num_single_layers=0, extra_condition_channels=4)          # 68 hint channels: padded K
    tr = FluxTransformer2DModel(**cfg_t, device=gpu, dtype=torch.bfloat16).random_init_(1)
    cn = FluxControlNetModel(**cfg_c, device=gpu, dtype=torch.bfloat16).random_init_(2)
    g = torch.Generator(device=gpu).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=gpu, generator=g).to(torch.bfloat16)
    B, T, N = 2, 64, 192
    lat, pe, pooled, cond = r(B, N, 64), r(B, T, 256), r(B, 64), r(B, N, 68)
    ids = orc.latent_image_ids(16, 48).to(gpu, torch.bfloat16)
    tids = torch.zeros(T, 3, device=gpu, dtype=torch.bfloat16)
    kw = dict(encoder_hidden_states=pe, pooled_projections=pooled, timestep=torch.full((B,), 0.6, device=gpu),
              guidance=torch.full((B,), 3.5, device=gpu), img_ids=ids, txt_ids=tids, return_dict=False)
    s_ref, _ = cn(hidden_states=lat, controlnet_cond=cond, **kw)
    s_ref = [s.clone() for s in s_ref]
    st = cn.prepare_static(pe, cond)
    s_hoist, _ = cn(hidden_states=lat, controlnet_cond=cond, _static=st, **kw)
    for a, b in zip(s_ref, s_hoist):
        assert torch.equal(a, b)
    # 5 transformer blocks against 3 samples: interval ceil(5/3) = 2 -> samples 0,1,2 all read; against 4 blocks only 0,1
    s_skip, _ = cn(hidden_states=lat, controlnet_cond=cond, _static=st, _blocks_needed=(2, 0), **kw)
    assert s_skip[2] is None and torch.equal(s_skip[0], s_ref[0]) and torch.equal(s_skip[1], s_ref[1])
    v_ref = tr(hidden_states=lat, controlnet_block_samples=s_ref, **kw)[0].clone()
    v_hoist = tr(hidden_states=lat, controlnet_block_samples=s_ref, _static=tr.prepare_static(pe), **kw)[0]
    assert torch.equal(v_ref, v_hoist)


def test_block_batch4_invariance_at_c2_shape(gpu):
    """BASELINE config 3's shard (4 images per GPU at S = 4608): entry b of a batch-4 pass through a full-size double + single
    block equals the batch-1 pass bit for bit — what makes a sample's result independent of how the batch is sharded."""
    from reptext_amd.transformer import FluxTransformer2DModel

    cfg = dict(WIDE_T, num_layers=1, num_single_layers=1)
    tr = FluxTransformer2DModel(**cfg, device=gpu, dtype=torch.bfloat16).random_init_(5)
    g = torch.Generator(device=gpu).manual_seed(2)
    r = lambda *s: torch.randn(*s, device=gpu, generator=g).to(torch.bfloat16)
    lat, pe, pooled = r(4, 4096, 64), r(4, 512, 4096), r(4, 768)
    ids = orc.latent_image_ids(128, 128).to(gpu, torch.bfloat16)
    tids = torch.zeros(512, 3, device=gpu, dtype=torch.bfloat16)
    kw = dict(img_ids=ids, txt_ids=tids, return_dict=False)
    o4 = tr(hidden_states=lat, encoder_hidden_states=pe, pooled_projections=pooled, timestep=torch.full((4,), 0.5, device=gpu),
            guidance=torch.full((4,), 3.5, device=gpu), **kw)[0].clone()
    assert torch.isfinite(o4.float()).all()
    for b in (0, 3):
        o1 = tr(hidden_states=lat[b : b + 1], encoder_hidden_states=pe[b : b + 1], pooled_projections=pooled[b : b + 1],
                timestep=torch.full((1,), 0.5, device=gpu), guidance=torch.full((1,), 3.5, device=gpu), **kw)[0]
        assert torch.equal(o4[b], o1[0])


def test_c4_inpaint_flow_at_1024(gpu):
    """BASELINE config 4 at its own shape: FluxControlNetInpaintPipeline at 1024x1024 — true CFG makes the internal batch 2 at
    S = 4608, the inpaint tower has 64 + 4 = 68 hint channels (K padded to 128 for the MFMA loop). Full width, depth 1+1
    (towers 1+0). Checked: step 0 leaves the latents exactly unchanged (zero velocity, INP:1264-1270, Q7); the unconditional
    branch does not depend on the positive prompt (no arithmetic across the two batch entries); two steps run finite and
    repeat bit for bit."""
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline_inpaint import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    cfg_t = dict(WIDE_T, num_layers=1, num_single_layers=1)
    cfg_c = dict(cfg_t, num_single_layers=0, extra_condition_channels=64)
    cfg_i = dict(cfg_t, num_single_layers=0, extra_condition_channels=4)
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
    hint, hint_inp, lat0 = r(1, N, 128), r(1, N, 68), r(1, N, 64)
    from PIL import Image

    mask_np = np.zeros([H, W], dtype=np.uint8)
    mask_np[200:420, 300:800] = 255
    mask = Image.fromarray(mask_np)

    def call(steps, pos=pe):
        return pipe(prompt_embeds=pos, pooled_prompt_embeds=pooled, negative_prompt_embeds=npe, negative_pooled_prompt_embeds=npooled,
                    height=H, width=W, num_inference_steps=steps, guidance_scale=3.5, true_guidance_scale=2.0,
                    control_image=[hint], control_image_inpaint=hint_inp, control_mask=[mask], controlnet_conditioning_scale=1.0,
                    controlnet_conditioning_scale_inpaint=1.0, controlnet_conditioning_step=30, latents=lat0, output_type="latent").images

    out1 = call(1)
    assert torch.equal(out1, lat0.float())                       # Q7: the first step's velocity is exactly zero
    out2 = call(2)
    assert torch.isfinite(out2).all() and not torch.equal(out2, lat0.float())
    assert torch.equal(out2, call(2))
    # entries of the internal batch are independent: the unconditional velocity is the same whatever the positive prompt is
    kw = dict(hidden_states=lat0, pooled_projections=torch.cat([npooled, pooled]), timestep=torch.full((1,), 0.7, device=gpu),
              guidance=torch.full((1,), 3.5, device=gpu), img_ids=orc.latent_image_ids(128, 128).to(gpu, torch.bfloat16),
              txt_ids=torch.zeros(T, 3, device=gpu, dtype=torch.bfloat16), return_dict=False)
    va = tr(encoder_hidden_states=torch.cat([npe, pe]), **kw)[0].clone()
    vb = tr(encoder_hidden_states=torch.cat([npe, r(1, T, 4096)]), **kw)[0]
    assert torch.equal(va[0], vb[0]) and not torch.equal(va[1], vb[1])
    sa, _ = cni(controlnet_cond=torch.cat([hint_inp] * 2), encoder_hidden_states=torch.cat([npe, pe]), **kw)
    kw_swapped = dict(kw, pooled_projections=torch.cat([pooled, npooled]))
    sb, _ = cni(controlnet_cond=torch.cat([hint_inp] * 2), encoder_hidden_states=torch.cat([pe, npe]), **kw_swapped)
    assert torch.equal(sa[0][0], sb[0][1]) and torch.equal(sa[0][1], sb[0][0])       # swapping the two entries swaps the results


def test_fp8_attention_at_c5_shape(gpu):
    """BASELINE config 5 (1536x1536: S = 9728) on the e4m3 attention: prep + kernel repeat bit for bit, constant V comes back
    as V (softmax weights sum to one), keys permuted with their values leave the output unchanged up to rounding."""
    import reptext_amd.ops as ops
    from reptext_amd import native

    FP8 = torch.float8_e4m3fn
    B, S, H = 1, 9728, 24
    d = H * 128
    g = torch.Generator(device=gpu).manual_seed(6)
    qkv = torch.randn(B, S, 3 * d, device=gpu, generator=g).to(torch.bfloat16)
    wn = torch.ones(128, device=gpu, dtype=torch.bfloat16)
    cos, sin = torch.ones(S, 128, device=gpu), torch.zeros(S, 128, device=gpu)
    qk8 = torch.empty(B, S, 2 * d, device=gpu, dtype=FP8)
    vt8 = torch.empty(int(native.load().rt_attention_fp8_vt_bytes(B, S, H)), device=gpu, dtype=FP8)

    def run(buf):
        ops.attention_fp8_prep(buf, 0, d, 2 * d, H, 512, wn, wn, wn, wn, cos, sin, qk8, vt8)
        o = torch.empty(B, S, d, device=gpu, dtype=torch.bfloat16)
        ops.attention_fp8(qk8, vt8, o, H)
        return o

    o_a = run(qkv)
    assert torch.isfinite(o_a.float()).all() and torch.equal(o_a, run(qkv))
    perm = torch.randperm(S, device=gpu, generator=g)
    qkv_p = qkv.clone()
    qkv_p[:, :, d:] = qkv[:, perm, d:]
    assert rel_l2(run(qkv_p).float(), o_a.float()) < 6e-2          # e4m3 numerators: 3 mantissa bits, summation order differs
    vconst = torch.randn(d, device=gpu, generator=g).to(FP8).to(torch.bfloat16)
    qkv_p[..., 2 * d :] = vconst
    assert rel_l2(run(qkv_p).float(), vconst.float().expand(B, S, d)) < 3e-3


def test_derived_weight_caches_follow_in_place_loads(gpu):
    """repacked conv weights / the fused mid-attention q|k|v of the VAE and the K-padded hint embedder of a
    68-channel tower are derived tensors; load_state_dict and random_init_ rewrite parameters IN PLACE (same data_ptr), so the
    caches must be dropped explicitly. A model that has run once and then receives new weights must equal a fresh model."""
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.vae import AutoencoderKL

    p_a, p_b = vorc.init_vae_params(VAE_SMALL, seed=3), vorc.init_vae_params(VAE_SMALL, seed=4)
    z = torch.randn(1, 16, 16, 16, generator=torch.Generator().manual_seed(1)).to(gpu, torch.bfloat16)
    vae = AutoencoderKL(**VAE_SMALL, device=gpu, dtype=torch.bfloat16)
    vae.load_state_dict(p_a)
    out_a = vae.decode(z, return_dict=False)[0].clone()
    vae.load_state_dict(p_b)                                         # in place: every data_ptr is unchanged
    out_b = vae.decode(z, return_dict=False)[0].clone()
    fresh = AutoencoderKL(**VAE_SMALL, device=gpu, dtype=torch.bfloat16)
    fresh.load_state_dict(p_b)
    assert torch.equal(out_b, fresh.decode(z, return_dict=False)[0]) and not torch.equal(out_a, out_b)
    vae.random_init_(9)
    fresh.random_init_(9)
    assert torch.equal(vae.decode(z, return_dict=False)[0], fresh.decode(z, return_dict=False)[0])

    cfg = dict(WIDE_T, num_attention_heads=2, joint_attention_dim=128, pooled_projection_dim=64, num_layers=1, num_single_layers=0,
               extra_condition_channels=4)
    g = torch.Generator(device=gpu).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=gpu, generator=g).to(torch.bfloat16)
    kw = dict(hidden_states=r(1, 64, 64), controlnet_cond=r(1, 64, 68), encoder_hidden_states=r(1, 32, 128), pooled_projections=r(1, 64),
              timestep=torch.full((1,), 0.5, device=gpu), guidance=torch.full((1,), 3.5, device=gpu),
              img_ids=orc.latent_image_ids(16, 16).to(gpu, torch.bfloat16), txt_ids=torch.zeros(32, 3, device=gpu, dtype=torch.bfloat16),
              return_dict=False)
    cn = FluxControlNetModel(**cfg, device=gpu, dtype=torch.bfloat16).random_init_(1)
    first = cn(**kw)[0][0].clone()
    cn.random_init_(2)                                               # in place
    second = cn(**kw)[0][0].clone()
    cn_fresh = FluxControlNetModel(**cfg, device=gpu, dtype=torch.bfloat16).random_init_(2)
    assert torch.equal(second, cn_fresh(**kw)[0][0]) and not torch.equal(first, second)
    sd = {k: v.clone() for k, v in FluxControlNetModel(**cfg, device=gpu, dtype=torch.bfloat16).random_init_(3).state_dict().items()}
    cn.load_state_dict(sd)
    cn_fresh.load_state_dict(sd)
    assert torch.equal(cn(**kw)[0][0], cn_fresh(**kw)[0][0])


def test_saved_pipeline_loads_through_hub_id_and_steps_identically(gpu, tmp_path, monkeypatch):
    """SURVEY §8f-2 / infer.py:27-33: a pipeline written in the diffusers layout (model_index.json, sharded transformer with
    its index) under the local hub cache loads back through its HUB ID and produces the same latents, bit for bit, as the
    in-memory models."""
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel
    from reptext_amd.vae import AutoencoderKL

    cfg_t = dict(WIDE_T, num_attention_heads=4, joint_attention_dim=256, pooled_projection_dim=64, num_layers=2, num_single_layers=2)
    cfg_c = dict(cfg_t, num_layers=2, num_single_layers=0, extra_condition_channels=64)
    tr = FluxTransformer2DModel(**cfg_t, device=gpu, dtype=torch.bfloat16).random_init_(1)
    cn = FluxControlNetModel(**cfg_c, device=gpu, dtype=torch.bfloat16).random_init_(2)
    vae = AutoencoderKL(**VAE_SMALL, device=gpu, dtype=torch.bfloat16).random_init_(3)
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), vae, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    hub = tmp_path / "hf_home" / "hub"
    commit = "f" * 40
    for repo, saver in (("black-forest-labs/FLUX.1-dev", lambda d: pipe.save_pretrained(d, max_shard_bytes=sum(v.numel() * 2 for v in tr.state_dict().values()) // 2)),
                        ("Shakker-Labs/RepText", cn.save_pretrained)):
        snap = hub / ("models--" + repo.replace("/", "--")) / "snapshots" / commit
        snap.mkdir(parents=True)
        (snap.parent.parent / "refs").mkdir()
        (snap.parent.parent / "refs" / "main").write_text(commit)
        saver(str(snap))
    assert len([f for f in os.listdir(hub / "models--black-forest-labs--FLUX.1-dev" / "snapshots" / commit / "transformer") if f.endswith(".safetensors")]) >= 2
    for env in ("HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE"):
        monkeypatch.delenv(env, raising=False)
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf_home"))
    # the three lines of infer.py:27-33, ids unchanged
    controlnet = FluxControlNetModel.from_pretrained("Shakker-Labs/RepText", torch_dtype=torch.bfloat16)
    pipe2 = FluxControlNetPipeline.from_pretrained("black-forest-labs/FLUX.1-dev", controlnet=controlnet, torch_dtype=torch.bfloat16).to("cuda")
    pipe2.set_progress_bar_config(disable=True)
    g = torch.Generator(device=gpu).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=gpu, generator=g).to(torch.bfloat16)
    kw = dict(prompt_embeds=r(1, 64, 256), pooled_prompt_embeds=r(1, 64), height=256, width=256, num_inference_steps=2, guidance_scale=3.5,
              control_image=[r(1, 256, 128)], latents=r(1, 256, 64), output_type="latent")
    assert torch.equal(pipe(**kw).images, pipe2(**kw).images)
    img_a, img_b = pipe(**dict(kw, output_type="np")).images, pipe2(**dict(kw, output_type="np")).images
    assert np.array_equal(img_a, img_b)


def test_reference_bf16_scalar_mode_matches_oracle(gpu):
    """`pipe.reference_bf16_scalars = True` rounds t, t/1000 and guidance*1000 to bf16 where the reference's bf16
    run does (PIPE:1025,1048; CN:282-284: t = 622.46 -> 624, guidance 3.5 -> 3504). The oracle has the same switch; the default of
    both is the exact fp32 scalar. Each mode agrees with its oracle; the two modes differ by more than the tolerance."""
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    cfg_t = dict(WIDE_T, num_attention_heads=4, joint_attention_dim=256, pooled_projection_dim=64, num_layers=2, num_single_layers=2)
    cfg_c = dict(cfg_t, num_layers=2, num_single_layers=0, extra_condition_channels=64)
    tp, cp = orc.init_mmdit_params(cfg_t, seed=51), orc.init_mmdit_params(cfg_c, seed=52, controlnet=True)
    tr = FluxTransformer2DModel(**cfg_t, device=gpu, dtype=torch.bfloat16)
    cn = FluxControlNetModel(**cfg_c, device=gpu, dtype=torch.bfloat16)
    tr.load_state_dict(tp)
    cn.load_state_dict(cp)
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), None, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    g = torch.Generator().manual_seed(8)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    N, T = 256, 64
    pe, pooled, hint, lat0 = r(1, T, 256), r(1, 64), r(1, N, 128), r(1, N, 64)
    sig = orc.flow_sigmas(3, orc.calculate_shift(N, 256, 4096, 0.5, 1.15))
    ids, tids = orc.latent_image_ids(32, 32), torch.zeros(T, 3)
    kw = dict(prompt_embeds=pe.to(gpu, torch.bfloat16), pooled_prompt_embeds=pooled.to(gpu, torch.bfloat16), height=256, width=256,
              num_inference_steps=3, guidance_scale=3.5, control_image=[hint.to(gpu, torch.bfloat16)], latents=lat0.to(gpu, torch.bfloat16),
              output_type="latent")
    try:
        out_exact = pipe(**kw).images.float().cpu()
        pipe.reference_bf16_scalars = True
        out_ref16 = pipe(**kw).images.float().cpu()
    finally:
        pipe.reference_bf16_scalars = False
        pipe._model_timestep(1.0)                    # resets the module-level switch
    ref_exact = orc.denoise_loop(tp, cfg_t, cp, cfg_c, lat0, pe, pooled, [hint], [None], sig, ids, tids, 3.5)
    with orc.reference_bf16_scalars():
        ref_ref16 = orc.denoise_loop(tp, cfg_t, cp, cfg_c, lat0, pe, pooled, [hint], [None], sig, ids, tids, 3.5)
    e_exact, e_ref16, gap = rel_l2(out_exact, ref_exact), rel_l2(out_ref16, ref_ref16), rel_l2(ref_ref16, ref_exact)
    print(f"scalar modes: exact {e_exact:.3e}, reference-bf16 {e_ref16:.3e}; the two oracles differ by {gap:.3e}")
    assert e_exact < 1.5e-3 and e_ref16 < 1.5e-3
    assert gap > 2 * max(e_exact, e_ref16)           # the rounding of the scalars is visible well above the parity error
    assert rel_l2(out_ref16, ref_exact) > 2 * e_ref16
