"""GPU parity of the AutoencoderKL kernels and of the end-to-end pipeline (BASELINE config C1 shape: 256x256, 2 steps)
against the fp32 CPU oracle. Reduced-depth / reduced-width random weights, bf16-rounded on both sides."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import flux_oracle as orc  # noqa: E402
from oracle import vae_oracle as vorc  # noqa: E402


from test_models_gpu import assert_at_dtype_floor  # noqa: E402  (GPU error must sit at the bf16-storage oracle's own floor)


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


VAE_SMALL = dict(vorc.FLUX_VAE_CFG, block_out_channels=(64, 128, 256, 256))   # same topology, 1/2 width: oracle in seconds
SMALL_T = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=3, attention_head_dim=128, num_attention_heads=4,
               joint_attention_dim=256, pooled_projection_dim=64, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
SMALL_CN = dict(SMALL_T, num_layers=2, num_single_layers=0, extra_condition_channels=64)


@pytest.fixture(scope="module")
def vae_pair(gpu):
    from reptext_amd.vae import AutoencoderKL

    p = vorc.init_vae_params(VAE_SMALL, seed=3)
    vae = AutoencoderKL(**VAE_SMALL, device=gpu, dtype=torch.bfloat16)
    vae.load_state_dict(p, strict=True)
    return p, vae


def test_conv_kernel_modes(gpu):
    """conv3x3 / 1x1 / fused nearest-2x / stride-2 pad(0,1,0,1) against torch conv2d."""
    import torch.nn.functional as F

    from reptext_amd import native

    lib = native.load()
    g = torch.Generator().manual_seed(0)
    B, H, W, Cin, Cout = 2, 12, 20, 64, 136
    x = torch.randn(B, Cin, H, W, generator=g).to(torch.bfloat16).float()
    st = torch.cuda.current_stream().cuda_stream

    def haloed(t):
        out = torch.zeros(t.shape[0], t.shape[2] + 2, t.shape[3] + 2, t.shape[1], device=gpu, dtype=torch.bfloat16)
        out[:, 1:-1, 1:-1, :] = t.permute(0, 2, 3, 1).to(gpu, torch.bfloat16)
        return out

    xh = haloed(x)
    for ks, stride, up in [(3, 1, 0), (1, 1, 0), (3, 1, 1), (3, 2, 0)]:
        w = (torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5).to(torch.bfloat16).float()
        b = torch.randn(Cout, generator=g).to(torch.bfloat16).float()
        if stride == 2:
            ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
        elif up:
            ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
        else:
            ref = F.conv2d(x, w, b, padding=ks // 2)
        Ho, Wo = ref.shape[2], ref.shape[3]
        res = torch.randn(B, Cout, Ho, Wo, generator=g).to(torch.bfloat16).float()
        y = torch.zeros(B, Ho + 2, Wo + 2, Cout, device=gpu, dtype=torch.bfloat16)
        wp = w.permute(0, 2, 3, 1).contiguous().to(gpu, torch.bfloat16)
        bd, rh = b.to(gpu, torch.bfloat16), haloed(res)      # keep the device tensors alive across the raw-pointer call
        native.check("conv", lib.rt_conv2d_nhwc(xh.data_ptr(), wp.data_ptr(), bd.data_ptr(), rh.data_ptr(),
                                                y.data_ptr(), B, H, W, Cin, Cout, ks, stride, up, 0, st))
        got = y[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2).float().cpu()
        assert rel_l2(got, ref + res) < 4e-3, (ks, stride, up)
        assert float(y[:, 0].abs().max()) == 0 and float(y[:, :, 0].abs().max()) == 0    # halo untouched
        assert float(y[:, -1].abs().max()) == 0 and float(y[:, :, -1].abs().max()) == 0
        if stride == 1 and not up:
            # these calls ran on the GEMM's convolution form (rt_conv2d_variant 1, the default); conv_nhwc_kernel accumulates every
            # output element in the same K order: the two must agree bit for bit, in place over the residual as the ResnetBlock uses it
            prev = lib.rt_conv2d_variant(0)
            try:
                y0 = rh.clone()
                native.check("conv", lib.rt_conv2d_nhwc(xh.data_ptr(), wp.data_ptr(), bd.data_ptr(), y0.data_ptr(), y0.data_ptr(), B, H, W, Cin, Cout, ks, 1, 0, 0, st))
            finally:
                lib.rt_conv2d_variant(prev)
            assert prev == 1
            y1 = rh.clone()
            native.check("conv", lib.rt_conv2d_nhwc(xh.data_ptr(), wp.data_ptr(), bd.data_ptr(), y1.data_ptr(), y1.data_ptr(), B, H, W, Cin, Cout, ks, 1, 0, 0, st))
            assert torch.equal(y0, y1) and torch.equal(y1, y)


def test_vae_decode(vae_pair, gpu):
    p, vae = vae_pair
    g = torch.Generator().manual_seed(1)
    z = torch.randn(1, 16, 16, 16, generator=g).to(torch.bfloat16).float()
    ref = vorc.decode(p, VAE_SMALL, z)
    out = vae.decode(z.to(gpu, torch.bfloat16), return_dict=False)[0]
    err = rel_l2(out.float().cpu(), ref)
    with orc.stored_as(torch.bfloat16):
        ref16 = vorc.decode(p, VAE_SMALL, z)
    err16 = rel_l2(out.float().cpu(), ref16)
    print(f"vae decode rel-L2 {err:.3e} vs fp32 oracle, {err16:.3e} vs bf16-storage oracle (floor {rel_l2(ref16, ref):.3e})")
    assert err < 3e-2          # ~40 bf16-stored layers incl. GroupNorms, bf16 skip connections
    assert_at_dtype_floor(err, err16, rel_l2(ref16, ref))
    # packed fast path: unpack + z/scaling + shift fused, uint8 out
    lat = ((z - 0.1159) * 0.3611).to(torch.bfloat16)
    packed = orc.pack_latents(lat.float()).to(gpu, torch.bfloat16)
    u8 = vae.decode_packed(packed, 16, 16, output_u8=True)
    ref_u8 = ((ref / 2 + 0.5).clamp(0, 1) * 255).round().permute(0, 2, 3, 1)
    diff = (u8.float().cpu() - ref_u8).abs()
    print(f"vae u8 mean abs diff {float(diff.mean()):.3f} max {float(diff.max()):.0f}")
    assert float(diff.mean()) < 2.0


def test_vae_encode(vae_pair, gpu):
    p, vae = vae_pair
    g = torch.Generator().manual_seed(2)
    x = (torch.rand(1, 3, 128, 128, generator=g) * 2 - 1).to(torch.bfloat16).float()
    mean, logvar = vorc.encode_moments(p, VAE_SMALL, x)
    dist = vae.encode(x.to(gpu, torch.bfloat16)).latent_dist
    e1, e2 = rel_l2(dist.mean.float().cpu(), mean), rel_l2(dist.logvar.float().cpu(), logvar)
    with orc.stored_as(torch.bfloat16):
        mean16, logvar16 = vorc.encode_moments(p, VAE_SMALL, x)
    print(f"vae encode rel-L2 mean {e1:.3e} logvar {e2:.3e} (floors {rel_l2(mean16, mean):.3e} {rel_l2(logvar16, logvar):.3e})")
    assert e1 < 3e-2 and e2 < 3e-2
    assert_at_dtype_floor(e1, rel_l2(dist.mean.float().cpu(), mean16), rel_l2(mean16, mean))
    assert_at_dtype_floor(e2, rel_l2(dist.logvar.float().cpu(), logvar16), rel_l2(logvar16, logvar))
    gen = torch.Generator().manual_seed(7)
    s = dist.sample(gen)
    noise = torch.randn(mean.shape, generator=torch.Generator().manual_seed(7), dtype=torch.bfloat16).float()
    assert rel_l2(s.float().cpu(), vorc.sample_latents(mean, logvar, noise)) < 3e-2


def test_pipeline_c1_latents_and_image(vae_pair, gpu):
    """BASELINE config C1 shape (256x256, 2 steps, one glyph line) through FluxControlNetPipeline.__call__ with
    output_type='latent' (the parity tap, PIPE:1132-1133) against the oracle's denoise loop."""
    from PIL import Image

    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    p_vae, vae = vae_pair
    tp = orc.init_mmdit_params(SMALL_T, seed=11)
    cp = orc.init_mmdit_params(SMALL_CN, seed=12, controlnet=True)
    tr = FluxTransformer2DModel(**SMALL_T, device=gpu, dtype=torch.bfloat16)
    cn = FluxControlNetModel(**SMALL_CN, device=gpu, dtype=torch.bfloat16)
    tr.load_state_dict(tp); cn.load_state_dict(cp)
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), vae, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    H = W = 256
    N, T = 256, 64
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    pe, pooled, hint = r(1, T, 256), r(1, 64), r(1, N, 128)
    noise = r(1, 16, 32, 32)
    lat0 = orc.pack_latents(noise)
    mask_np = np.zeros([H, W], dtype=np.uint8)
    mask_np[60:140, 80:200] = 255
    mask_img = Image.fromarray(mask_np)
    rm = torch.nn.functional.interpolate(torch.from_numpy(mask_np)[None, None].float() / 255.0, scale_factor=1 / 16, mode="bilinear").reshape(1, -1, 1)
    sig = orc.flow_sigmas(2, orc.calculate_shift(N, 256, 4096, 0.5, 1.15))
    assert abs(float(sig[1]) - 0.622459) < 1e-5
    ref = orc.denoise_loop(tp, SMALL_T, cp, SMALL_CN, lat0, pe, pooled, [hint], [rm], sig, orc.latent_image_ids(32, 32), torch.zeros(T, 3), 3.5)
    out = pipe(prompt_embeds=pe.to(gpu, torch.bfloat16), pooled_prompt_embeds=pooled.to(gpu, torch.bfloat16), height=H, width=W,
               num_inference_steps=2, guidance_scale=3.5, control_image=[hint.to(gpu, torch.bfloat16)], control_mask=[mask_img],
               controlnet_conditioning_scale=1.0, controlnet_conditioning_step=30, latents=lat0.to(gpu, torch.bfloat16),
               output_type="latent").images
    err = rel_l2(out.float().cpu(), ref)
    with orc.stored_as(torch.bfloat16):     # same oracle, bf16 where the HIP path stores bf16: logic error without dtype noise
        ref16 = orc.denoise_loop(tp, SMALL_T, cp, SMALL_CN, lat0, pe, pooled, [hint], [rm], sig, orc.latent_image_ids(32, 32), torch.zeros(T, 3), 3.5)
    err16 = rel_l2(out.float().cpu(), ref16)
    print(f"C1-shape pipeline latents rel-L2 {err:.3e} vs fp32 oracle, {err16:.3e} vs bf16-storage oracle")
    assert out.dtype == torch.float32          # the latent tap returns the loop's fp32 state
    assert err < 1e-3                          # north-star tolerance (BASELINE.json): latents within 1e-3 rel-L2 of the CPU path
    assert_at_dtype_floor(err, err16, rel_l2(ref16, ref))
    # zero-initialised tower == plain FLUX through the whole pipeline
    cn.zero_init_controlnet_()
    ref0 = orc.denoise_loop(tp, SMALL_T, None, None, lat0, pe, pooled, [], [], sig, orc.latent_image_ids(32, 32), torch.zeros(T, 3), 3.5)
    out0 = pipe(prompt_embeds=pe.to(gpu, torch.bfloat16), pooled_prompt_embeds=pooled.to(gpu, torch.bfloat16), height=H, width=W,
                num_inference_steps=2, guidance_scale=3.5, control_image=[hint.to(gpu, torch.bfloat16)], control_mask=[mask_img],
                latents=lat0.to(gpu, torch.bfloat16), output_type="latent").images
    assert rel_l2(out0.float().cpu(), ref0) < 2e-2
    # full call to PIL through the VAE decoder
    img = pipe(prompt_embeds=pe.to(gpu, torch.bfloat16), pooled_prompt_embeds=pooled.to(gpu, torch.bfloat16), height=H, width=W,
               num_inference_steps=2, guidance_scale=3.5, control_image=[hint.to(gpu, torch.bfloat16)], control_mask=[mask_img],
               latents=lat0.to(gpu, torch.bfloat16)).images[0]
    assert img.size == (256, 256)
    with pytest.raises(ValueError):
        pipe(prompt_embeds=pe.to(gpu), height=250, width=256)        # PIPE:496


def test_inpaint_pipeline_cfg_and_second_tower(vae_pair, gpu):
    """BASELINE config 4 shape family: text tower (masked) + inpaint tower (68 hint channels, unmasked) + true CFG with the
    zero-velocity first step, against the oracle restatement of INP:1138-1285."""
    from PIL import Image

    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline_inpaint import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    _, vae = vae_pair
    INP_CN = dict(SMALL_T, num_layers=2, num_single_layers=0, extra_condition_channels=4)
    tp = orc.init_mmdit_params(SMALL_T, seed=21)
    cp = orc.init_mmdit_params(SMALL_CN, seed=22, controlnet=True)
    ip = orc.init_mmdit_params(INP_CN, seed=23, controlnet=True)
    tr = FluxTransformer2DModel(**SMALL_T, device=gpu, dtype=torch.bfloat16)
    cn = FluxControlNetModel(**SMALL_CN, device=gpu, dtype=torch.bfloat16)
    cni = FluxControlNetModel(**INP_CN, device=gpu, dtype=torch.bfloat16)
    tr.load_state_dict(tp); cn.load_state_dict(cp); cni.load_state_dict(ip)
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), vae, None, None, None, None, tr, cn, cni)
    pipe.set_progress_bar_config(disable=True)
    H = W = 256
    N, T = 256, 64
    g = torch.Generator().manual_seed(8)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    pe, pooled, npe, npooled = r(1, T, 256), r(1, 64), r(1, T, 256), r(1, 64)
    hint, hint_inp = r(1, N, 128), r(1, N, 68)
    lat0 = orc.pack_latents(r(1, 16, 32, 32))
    mask_np = np.zeros([H, W], dtype=np.uint8); mask_np[40:200, 30:120] = 255
    rm = torch.nn.functional.interpolate(torch.from_numpy(mask_np)[None, None].float() / 255.0, scale_factor=1 / 16, mode="bilinear").reshape(1, -1, 1)
    sig = orc.flow_sigmas(3, orc.calculate_shift(N, 256, 4096, 0.5, 1.15))
    ids, tids = orc.latent_image_ids(32, 32), torch.zeros(T, 3)
    ref = orc.denoise_loop_inpaint(tp, SMALL_T, cp, SMALL_CN, ip, INP_CN, lat0, pe, pooled, npe, npooled, [hint], [rm], hint_inp, sig, ids, tids,
                                   guidance_scale=3.5, true_guidance_scale=2.0, conditioning_scale_inpaint=0.9)
    b16 = lambda t: t.to(gpu, torch.bfloat16)
    kw = dict(prompt_embeds=b16(pe), pooled_prompt_embeds=b16(pooled), negative_prompt_embeds=b16(npe), negative_pooled_prompt_embeds=b16(npooled),
              height=H, width=W, num_inference_steps=3, guidance_scale=3.5, true_guidance_scale=2.0, control_image=[b16(hint)],
              control_mask=[Image.fromarray(mask_np)], control_image_inpaint=b16(hint_inp), controlnet_conditioning_scale_inpaint=0.9,
              latents=b16(lat0), output_type="latent")
    out = pipe(**kw).images
    err = rel_l2(out.float().cpu(), ref)
    with orc.stored_as(torch.bfloat16):
        ref16 = orc.denoise_loop_inpaint(tp, SMALL_T, cp, SMALL_CN, ip, INP_CN, lat0, pe, pooled, npe, npooled, [hint], [rm], hint_inp, sig, ids,
                                         tids, guidance_scale=3.5, true_guidance_scale=2.0, conditioning_scale_inpaint=0.9)
    err16 = rel_l2(out.float().cpu(), ref16)
    print(f"inpaint pipeline (CFG, 2 towers, 3 steps) latents rel-L2 {err:.3e} vs fp32 oracle, {err16:.3e} vs bf16-storage oracle")
    assert err < 5e-3                          # CFG extrapolation (scale 2) doubles the velocity error; floor check below
    assert_at_dtype_floor(err, err16, rel_l2(ref16, ref))
    # step 0 is a zero-velocity step (Q7): a 1-step run returns the initial latents unchanged
    one = pipe(**dict(kw, num_inference_steps=1)).images
    assert torch.equal(one, b16(lat0).float())


def test_pipeline_many_steps_error_growth(vae_pair, gpu):
    """28 steps (the BASELINE step count) at the C1 resolution with reduced-depth weights: the latent error vs the fp32 oracle
    must not grow with the step count (fp32 master latents + fp32 residual stream)."""
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    _, vae = vae_pair
    tp = orc.init_mmdit_params(SMALL_T, seed=41)
    cp = orc.init_mmdit_params(SMALL_CN, seed=42, controlnet=True)
    tr = FluxTransformer2DModel(**SMALL_T, device=gpu, dtype=torch.bfloat16)
    cn = FluxControlNetModel(**SMALL_CN, device=gpu, dtype=torch.bfloat16)
    tr.load_state_dict(tp); cn.load_state_dict(cp)
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), vae, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    N, T, steps = 256, 64, 28
    g = torch.Generator().manual_seed(6)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    pe, pooled, hint = r(1, T, 256), r(1, 64), r(1, N, 128)
    lat0 = orc.pack_latents(r(1, 16, 32, 32))
    sig = orc.flow_sigmas(steps, orc.calculate_shift(N, 256, 4096, 0.5, 1.15))
    ref = orc.denoise_loop(tp, SMALL_T, cp, SMALL_CN, lat0, pe, pooled, [hint], [], sig, orc.latent_image_ids(32, 32), torch.zeros(T, 3), 3.5,
                           conditioning_step=20)
    b16 = lambda t: t.to(gpu, torch.bfloat16)
    out = pipe(prompt_embeds=b16(pe), pooled_prompt_embeds=b16(pooled), height=256, width=256, num_inference_steps=steps, guidance_scale=3.5,
               control_image=[b16(hint)], controlnet_conditioning_step=20, latents=b16(lat0), output_type="latent").images
    err = rel_l2(out.float().cpu(), ref)
    with orc.stored_as(torch.bfloat16):
        ref16 = orc.denoise_loop(tp, SMALL_T, cp, SMALL_CN, lat0, pe, pooled, [hint], [], sig, orc.latent_image_ids(32, 32), torch.zeros(T, 3),
                                 3.5, conditioning_step=20)
    err16 = rel_l2(out.float().cpu(), ref16)
    print(f"28-step pipeline latents rel-L2 {err:.3e} vs fp32 oracle, {err16:.3e} vs bf16-storage oracle")
    assert err < 1e-3                          # north-star tolerance; measured 2.5e-4
    assert_at_dtype_floor(err, err16, rel_l2(ref16, ref))


def test_call_with_pil_hints_matches_oracle_prelude(vae_pair, gpu):
    """The infer.py-shaped call: PIL canny / position / mask / glyph images go through preprocess -> VAE encode (posterior
    sampled from the GLOBAL RNG, quirk Q2) -> pack -> loop (PIPE:928-982). The global-RNG draws are replayed in the test
    (same device, same order, same shapes) so the oracle sees identical hint latents."""
    from PIL import Image, ImageDraw

    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    p_vae, vae = vae_pair
    tp = orc.init_mmdit_params(SMALL_T, seed=51)
    cp = orc.init_mmdit_params(SMALL_CN, seed=52, controlnet=True)
    tr = FluxTransformer2DModel(**SMALL_T, device=gpu, dtype=torch.bfloat16)
    cn = FluxControlNetModel(**SMALL_CN, device=gpu, dtype=torch.bfloat16)
    tr.load_state_dict(tp); cn.load_state_dict(cp)
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), vae, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    H = W = 256
    # synthetic glyph hint in the style of infer.py:71-100 (PIL only; cv2 is not needed for a synthetic edge image)
    glyph = Image.new("RGB", (W, H), (0, 0, 0))
    ImageDraw.Draw(glyph).rectangle((60, 90, 190, 150), fill=(255, 255, 255))
    edges = Image.new("RGB", (W, H), (255, 255, 255))
    ImageDraw.Draw(edges).rectangle((60, 90, 190, 150), outline=(0, 0, 0))
    pos_np = np.zeros([H, W], dtype=np.uint8); pos_np[90:150, 60:190] = 255
    mask_np = np.zeros([H, W], dtype=np.uint8); mask_np[85:155, 55:195] = 255
    position, mask = Image.fromarray(pos_np), Image.fromarray(mask_np)
    g = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    N, T = 256, 64
    pe, pooled = r(1, T, 256), r(1, 64)
    gen = torch.Generator().manual_seed(42)                     # CPU generator: reproducible for any device (utils.randn_tensor)

    torch.manual_seed(1234)
    out = pipe(prompt_embeds=pe.to(gpu, torch.bfloat16), pooled_prompt_embeds=pooled.to(gpu, torch.bfloat16), height=H, width=W,
               num_inference_steps=2, guidance_scale=3.5, control_image=[edges], control_position=[position], control_mask=[mask],
               control_glyph=glyph, generator=gen, output_type="latent").images

    # ---- oracle replay of the prelude
    torch.manual_seed(1234)
    n_img = torch.randn(1, 16, 32, 32, device=gpu, dtype=torch.bfloat16).float().cpu()      # latent_dist.sample() of the canny hint
    n_pos = torch.randn(1, 16, 32, 32, device=gpu, dtype=torch.bfloat16).float().cpu()      # ... of the position hint
    gen2 = torch.Generator().manual_seed(42)
    _glyph_noise = torch.randn(1, 16, 32, 32, generator=gen2, dtype=torch.float32)           # Q1: glyph posterior draw, result unused
    noise = torch.randn(1, 16, 32, 32, generator=gen2, dtype=torch.bfloat16).float()
    to_t = lambda im: torch.from_numpy(np.asarray(im).astype(np.float32) / 255.0)
    x_img = (to_t(edges).permute(2, 0, 1)[None] * 2 - 1).to(torch.bfloat16).float()
    x_pos = (to_t(position)[None, None] * 2 - 1).repeat(1, 3, 1, 1).to(torch.bfloat16).float()
    hint = []
    for x, n in ((x_img, n_img), (x_pos, n_pos)):
        mean, logvar = vorc.encode_moments(p_vae, VAE_SMALL, x)
        z = vorc.sample_latents(mean, logvar, n)
        hint.append(((z - 0.1159) * 0.3611).to(torch.bfloat16).float())
    packed_hint = orc.pack_latents(torch.cat(hint, dim=1))
    rm = torch.nn.functional.interpolate(torch.from_numpy(mask_np)[None, None].float() / 255.0, scale_factor=1 / 16, mode="bilinear").reshape(1, -1, 1)
    sig = orc.flow_sigmas(2, orc.calculate_shift(N, 256, 4096, 0.5, 1.15))
    ref = orc.denoise_loop(tp, SMALL_T, cp, SMALL_CN, orc.pack_latents(noise), pe, pooled, [packed_hint], [rm], sig, orc.latent_image_ids(32, 32),
                           torch.zeros(T, 3), 3.5)
    err = rel_l2(out.float().cpu(), ref)
    print(f"PIL-hint call (VAE-encoded hints, Q1/Q2 RNG order) latents rel-L2 {err:.3e}")
    assert err < 4e-3          # measured 1.3e-3 (hint latents come from the bf16 VAE encoder: 1e-2 on the hints, 1e-3 on the latents)


def test_vae_is_bitwise_reproducible(vae_pair, gpu):
    """GroupNorm statistics are reduced in a fixed order (no atomics), so encode and decode repeat bit for bit; a 1-ulp fp32
    wobble in one mean would otherwise be amplified to full bf16-level decorrelation a few layers later."""
    _, vae = vae_pair
    g = torch.Generator().manual_seed(11)
    z = torch.randn(2, 16, 16, 16, generator=g).to(gpu, torch.bfloat16)
    x = (torch.rand(1, 3, 128, 128, generator=g) * 2 - 1).to(gpu, torch.bfloat16)
    d0 = vae.decode(z, return_dict=False)[0].clone()
    e0 = vae.encode(x).latent_dist.mean.clone()
    for _ in range(3):
        assert torch.equal(vae.decode(z, return_dict=False)[0], d0)
        assert torch.equal(vae.encode(x).latent_dist.mean, e0)


def test_tower_stream_overlap_is_bitwise_neutral(vae_pair, gpu, monkeypatch):
    """The ControlNet tower runs on a side stream next to the transformer (events per sample). Two text lines with masks,
    tower active for 2 of 3 steps: latents must equal, bit for bit, those of the serial single-stream order."""
    import reptext_amd.pipeline as P
    from PIL import Image
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    _, vae = vae_pair
    tr = FluxTransformer2DModel(**SMALL_T, device=gpu, dtype=torch.bfloat16).random_init_(61)
    cn = FluxControlNetModel(**SMALL_CN, device=gpu, dtype=torch.bfloat16).random_init_(62)
    pipe = P.FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), vae, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    g = torch.Generator().manual_seed(8)
    r = lambda *s: torch.randn(*s, generator=g).to(gpu, torch.bfloat16)
    pe, pooled = r(1, 64, 256), r(1, 64)
    hints = [r(1, 256, 128), r(1, 256, 128)]
    lat0 = r(1, 256, 64)
    masks = []
    for box in ((40, 120, 30, 200), (140, 220, 60, 240)):
        m = np.zeros([256, 256], dtype=np.uint8); m[box[0]:box[1], box[2]:box[3]] = 255
        masks.append(Image.fromarray(m))
    kw = dict(prompt_embeds=pe, pooled_prompt_embeds=pooled, height=256, width=256, num_inference_steps=3, guidance_scale=3.5,
              control_image=hints, control_mask=masks, controlnet_conditioning_step=2, latents=lat0, output_type="latent")
    monkeypatch.setattr(P, "OVERLAP_TOWER", False)
    serial = pipe(**kw).images.clone()
    monkeypatch.setattr(P, "OVERLAP_TOWER", True)
    for _ in range(3):
        assert torch.equal(pipe(**kw).images, serial)


def test_denoise_loop_graph_replay_is_bitwise_the_eager_loop(vae_pair, gpu):
    """pipeline.GRAPH_CAPTURE: the first call of a signature runs eagerly, the second captures the whole loop (two text lines, tower
    off after step 2 of 3) into one hipGraph, later calls replay it with new inputs copied into its static buffers. Every result
    must equal the eager loop bit for bit — also for inputs the graph has never seen and after the weights were changed in place."""
    import reptext_amd.pipeline as P
    from PIL import Image
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    _, vae = vae_pair
    tr = FluxTransformer2DModel(**SMALL_T, device=gpu, dtype=torch.bfloat16).random_init_(71)
    cn = FluxControlNetModel(**SMALL_CN, device=gpu, dtype=torch.bfloat16).random_init_(72)
    pipe = P.FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), vae, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    g = torch.Generator().manual_seed(9)
    r = lambda *s: torch.randn(*s, generator=g).to(gpu, torch.bfloat16)
    masks = []
    for box in ((40, 120, 30, 200), (140, 220, 60, 240)):
        m = np.zeros([256, 256], dtype=np.uint8); m[box[0]:box[1], box[2]:box[3]] = 255
        masks.append(Image.fromarray(m))

    def inputs():
        return dict(prompt_embeds=r(1, 64, 256), pooled_prompt_embeds=r(1, 64), control_image=[r(1, 256, 128), r(1, 256, 128)], latents=r(1, 256, 64))

    fixed = dict(height=256, width=256, num_inference_steps=3, guidance_scale=3.5, control_mask=masks, controlnet_conditioning_step=2, output_type="latent")
    a, b = inputs(), inputs()
    pipe.capture_graphs = False
    ref_a, ref_b = pipe(**a, **fixed).images.clone(), pipe(**b, **fixed).images.clone()
    assert not torch.equal(ref_a, ref_b)
    pipe.capture_graphs = True
    assert torch.equal(pipe(**a, **fixed).images, ref_a)                       # eager, signature remembered
    assert torch.equal(pipe(**a, **fixed).images, ref_a)                       # captured + replayed
    ent = [v for v in pipe._graph_cache.values() if isinstance(v, dict)]
    assert len(ent) == 1                                                        # a graph exists for this signature
    assert torch.equal(pipe(**b, **fixed).images, ref_b)                       # replay on inputs the capture never saw
    assert torch.equal(pipe(**a, **fixed).images, ref_a)
    assert pipe.scheduler._step_index == 3                                      # what the eager loop leaves behind
    # weights changed in place (same storage): the graph reads the new values
    tr.random_init_(73)
    got = pipe(**a, **fixed).images.clone()
    pipe.capture_graphs = False
    assert torch.equal(got, pipe(**a, **fixed).images)
    assert not torch.equal(got, ref_a)
    # a different signature (2 steps) is not served by the 3-step graph
    pipe.capture_graphs = True
    two = dict(fixed, num_inference_steps=2)
    out2 = pipe(**a, **two).images.clone()
    pipe.capture_graphs = False
    assert torch.equal(out2, pipe(**a, **two).images)


def test_graphs_of_two_shapes_keep_their_buffers(vae_pair, gpu):
    """ADVICE round 2 (high + medium). A captured loop bakes in device pointers; the caches those buffers came from (tower sample
    buffers keyed by shape, mmdit._WS_CACHE, ops._ATTN_WS, rope tables) may evict them while the graph lives. Sequence
    A, A(capture), B, B(capture) replaces the sample cache; then every cache is cleared and the freed sizes are re-allocated as
    canaries before both graphs replay: outputs must equal the eager loop bit for bit and no canary byte may change.
    Also: toggling `reference_bf16_scalars` between calls of one shape must never replay the other mode's graph."""
    import reptext_amd.pipeline as P
    from reptext_amd import mmdit, ops
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    _, vae = vae_pair
    tr = FluxTransformer2DModel(**SMALL_T, device=gpu, dtype=torch.bfloat16).random_init_(81)
    cn = FluxControlNetModel(**SMALL_CN, device=gpu, dtype=torch.bfloat16).random_init_(82)
    pipe = P.FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), vae, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    g = torch.Generator().manual_seed(10)
    r = lambda *s: torch.randn(*s, generator=g).to(gpu, torch.bfloat16)

    def inputs(n):
        return dict(prompt_embeds=r(1, 64, 256), pooled_prompt_embeds=r(1, 64), control_image=[r(1, n, 128)], latents=r(1, n, 64))

    fa = dict(height=256, width=256, num_inference_steps=2, guidance_scale=3.5, controlnet_conditioning_step=2, output_type="latent")
    fb = dict(fa, width=384)
    a1, a2, b1, b2 = inputs(256), inputs(256), inputs(384), inputs(384)
    pipe.capture_graphs = False
    ref = {k: pipe(**i, **f).images.clone() for k, (i, f) in dict(a1=(a1, fa), a2=(a2, fa), b1=(b1, fb), b2=(b2, fb)).items()}
    pipe.reference_bf16_scalars = True
    ref16 = pipe(**a1, **fa).images.clone()
    pipe.reference_bf16_scalars = False
    assert not torch.equal(ref16, ref["a1"])
    pipe.capture_graphs = True
    assert torch.equal(pipe(**a1, **fa).images, ref["a1"])          # A: eager, remembered
    assert torch.equal(pipe(**a1, **fa).images, ref["a1"])          # A: captured
    assert torch.equal(pipe(**b1, **fb).images, ref["b1"])          # B: eager — replaces the shape-keyed sample cache
    assert torch.equal(pipe(**b1, **fb).images, ref["b1"])          # B: captured
    assert sum(isinstance(v, dict) for v in pipe._graph_cache.values()) == 2
    # evict everything the graphs' buffers were cached in, then grab the freed memory
    pipe._sample_cache = None
    mmdit._WS_CACHE.clear()
    ops._ATTN_WS.clear()
    tr._rope_cache.clear(); cn._rope_cache.clear()
    torch.cuda.synchronize()
    sizes = [256 * 64 * 2, 320 * 512 * 2, 320 * 512 * 4, 320 * 1536 * 2, 320 * 2048 * 2, 320 * 3584 * 2, 448 * 512 * 4, 448 * 3584 * 2, 1 << 20, 8 << 20]
    canaries = [torch.full((n,), 0x5A, device=gpu, dtype=torch.uint8) for n in sizes for _ in range(6)]
    for k, (i, f) in dict(a2=(a2, fa), b2=(b2, fb), a1=(a1, fa), b1=(b1, fb)).items():
        assert torch.equal(pipe(**i, **f).images, ref[k]), k         # replays on buffers the graph entries own
    torch.cuda.synchronize()
    assert all(bool((c == 0x5A).all()) for c in canaries)
    # mode toggle on a shape whose fp32-scalar graph exists: eager (new key), then its own graph; the old graph still serves the old mode
    pipe.reference_bf16_scalars = True
    assert torch.equal(pipe(**a1, **fa).images, ref16)
    assert torch.equal(pipe(**a1, **fa).images, ref16)
    pipe.reference_bf16_scalars = False
    assert torch.equal(pipe(**a1, **fa).images, ref["a1"])
    assert torch.equal(pipe(**a1, **fa).images, ref["a1"])


def test_pipeline_fp8_linears(vae_pair, gpu):
    """Config-5 precision through the whole loop (C1 shape, 4 steps, masked tower): latents vs the fp32 oracle and vs the oracle
    with the same e4m3 quantisation points; the GPU must sit on that run's floor."""
    from PIL import Image
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    _, vae = vae_pair
    tp = orc.init_mmdit_params(SMALL_T, seed=81)
    cp = orc.init_mmdit_params(SMALL_CN, seed=82, controlnet=True)
    tr = FluxTransformer2DModel(**SMALL_T, device=gpu, dtype=torch.bfloat16)
    cn = FluxControlNetModel(**SMALL_CN, device=gpu, dtype=torch.bfloat16)
    tr.load_state_dict(tp); cn.load_state_dict(cp)
    tr.enable_fp8_linears(); cn.enable_fp8_linears()
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), vae, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    N, T, steps = 256, 64, 4
    g = torch.Generator().manual_seed(15)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    pe, pooled, hint = r(1, T, 256), r(1, 64), r(1, N, 128)
    lat0 = orc.pack_latents(r(1, 16, 32, 32))
    mask_np = np.zeros([256, 256], dtype=np.uint8); mask_np[60:140, 80:200] = 255
    rm = torch.nn.functional.interpolate(torch.from_numpy(mask_np)[None, None].float() / 255.0, scale_factor=1 / 16, mode="bilinear").reshape(1, -1, 1)
    sig = orc.flow_sigmas(steps, orc.calculate_shift(N, 256, 4096, 0.5, 1.15))
    args = (tp, SMALL_T, cp, SMALL_CN, lat0, pe, pooled, [hint], [rm], sig, orc.latent_image_ids(32, 32), torch.zeros(T, 3), 3.5)
    ref = orc.denoise_loop(*args)
    with orc.stored_as(torch.bfloat16), orc.fp8_linears():
        ref8 = orc.denoise_loop(*args)
    b16 = lambda t: t.to(gpu, torch.bfloat16)
    out = pipe(prompt_embeds=b16(pe), pooled_prompt_embeds=b16(pooled), height=256, width=256, num_inference_steps=steps, guidance_scale=3.5,
               control_image=[b16(hint)], control_mask=[Image.fromarray(mask_np)], latents=b16(lat0), output_type="latent").images.float().cpu()
    err, err8, floor = rel_l2(out, ref), rel_l2(out, ref8), rel_l2(ref8, ref)
    print(f"fp8-linears pipeline latents rel-L2 {err:.3e} vs fp32 oracle, {err8:.3e} vs fp8 oracle (floor {floor:.3e})")
    assert_at_dtype_floor(err, err8, floor)
    assert err < 1e-2


def test_pipeline_non_square_ragged_shapes(vae_pair, gpu):
    """320 x 192 pixels -> 20 x 12 = 240 image tokens, 40 text tokens (S = 280: nothing is a multiple of a tile), two text lines,
    batch 2 through `num_images_per_prompt`-free batching: latents vs the fp32 oracle at the bf16 floor, image decodes."""
    from PIL import Image
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel

    _, vae = vae_pair
    tp = orc.init_mmdit_params(SMALL_T, seed=101)
    cp = orc.init_mmdit_params(SMALL_CN, seed=102, controlnet=True)
    tr = FluxTransformer2DModel(**SMALL_T, device=gpu, dtype=torch.bfloat16)
    cn = FluxControlNetModel(**SMALL_CN, device=gpu, dtype=torch.bfloat16)
    tr.load_state_dict(tp); cn.load_state_dict(cp)
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), vae, None, None, None, None, tr, cn)
    pipe.set_progress_bar_config(disable=True)
    H, W, T, B, steps = 192, 320, 40, 2, 3
    h2, w2 = 2 * (H // 16), 2 * (W // 16)
    N = (h2 // 2) * (w2 // 2)
    g = torch.Generator().manual_seed(23)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    pe, pooled = r(B, T, 256), r(B, 64)
    hints = [r(B, N, 128), r(B, N, 128)]
    lat0 = orc.pack_latents(r(B, 16, h2, w2))
    masks_np = []
    for box in ((16, 96, 32, 200), (100, 180, 120, 300)):
        m = np.zeros([H, W], dtype=np.uint8); m[box[0]:box[1], box[2]:box[3]] = 255
        masks_np.append(m)
    rms = [torch.nn.functional.interpolate(torch.from_numpy(m)[None, None].float() / 255.0, scale_factor=1 / 16, mode="bilinear").reshape(1, -1, 1) for m in masks_np]
    sig = orc.flow_sigmas(steps, orc.calculate_shift(N, 256, 4096, 0.5, 1.15))
    args = (tp, SMALL_T, cp, SMALL_CN, lat0, pe, pooled, hints, rms, sig, orc.latent_image_ids(h2, w2), torch.zeros(T, 3), 3.5)
    ref = orc.denoise_loop(*args)
    with orc.stored_as(torch.bfloat16):
        ref16 = orc.denoise_loop(*args)
    b16 = lambda t: t.to(gpu, torch.bfloat16)
    kw = dict(prompt_embeds=b16(pe), pooled_prompt_embeds=b16(pooled), height=H, width=W, num_inference_steps=steps, guidance_scale=3.5,
              control_image=[b16(h) for h in hints], control_mask=[Image.fromarray(m) for m in masks_np], latents=b16(lat0))
    out = pipe(**kw, output_type="latent").images.float().cpu()
    err, err16, floor = rel_l2(out, ref), rel_l2(out, ref16), rel_l2(ref16, ref)
    print(f"non-square ragged pipeline latents rel-L2 {err:.3e} vs fp32 oracle, {err16:.3e} vs bf16-storage oracle (floor {floor:.3e})")
    assert_at_dtype_floor(err, err16, floor)
    assert err < 1.5e-3
    imgs = pipe(**kw).images
    assert len(imgs) == B and imgs[0].size == (W, H)
