"""Pins the CPU oracle with the closed-form known answers of SURVEY.md §8c — the only pins that exist: the reference
ships no tests or golden tensors for this path (parity unpinned; see oracle/__init__.py)."""
import math

import torch

from oracle import flux_oracle as orc
from oracle import vae_oracle as vorc

C2_SIGMAS = [1.0, 0.988409, 0.976222, 0.963394, 0.949873, 0.935599, 0.920509, 0.904531, 0.887583, 0.869576, 0.850406, 0.829956,
             0.808096, 0.784672, 0.759511, 0.732413, 0.703145, 0.671435, 0.636964, 0.599357, 0.558163, 0.512844, 0.462748,
             0.407078, 0.344849, 0.274828, 0.195455, 0.104721, 0.0]


def test_calculate_shift():
    # scheduler config base/max shift 0.5/1.15, base/max seq 256/4096 (PIPE:952-958)
    assert abs(orc.calculate_shift(4096, 256, 4096, 0.5, 1.15) - 1.15) < 1e-12
    assert abs(orc.calculate_shift(256, 256, 4096, 0.5, 1.15) - 0.5) < 1e-12
    assert abs(orc.calculate_shift(9216, 256, 4096, 0.5, 1.15) - 2.01667) < 1e-5
    # function default max_shift is 1.16 (PIPE:83)
    assert abs(orc.calculate_shift(4096) - 1.16) < 1e-12


def test_sigma_schedules():
    s1 = orc.flow_sigmas(2, 0.5)
    assert torch.allclose(s1, torch.tensor([1.0, 0.622459, 0.0]), atol=1e-6)
    s2 = orc.flow_sigmas(28, 1.15)
    assert s2.dtype == torch.float32 and len(s2) == 29
    assert torch.allclose(s2, torch.tensor(C2_SIGMAS), atol=1.5e-6)


def test_pack_unpack_index_map_and_ids():
    x = torch.arange(2 * 3 * 4 * 6, dtype=torch.float32).reshape(2, 3, 4, 6)
    p = orc.pack_latents(x)
    assert p.shape == (2, 6, 12)
    for b in range(2):
        for i in range(2):
            for j in range(3):
                for c in range(3):
                    for dy in range(2):
                        for dx in range(2):
                            assert p[b, i * 3 + j, c * 4 + dy * 2 + dx] == x[b, c, 2 * i + dy, 2 * j + dx]
    assert torch.equal(orc.unpack_latents(p, 2 * 16, 3 * 16, 16), x)
    ids = orc.latent_image_ids(6, 8)          # latent 6x8 -> 3x4 tokens
    assert ids.shape == (12, 3)
    for r in range(3):
        for c in range(4):
            assert ids[r * 4 + c].tolist() == [0.0, float(r), float(c)]


def test_interval_map_19_over_6():
    assert orc.interval_map(19, 6) == [0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4]   # sample 5 never used (Q5)


def _count(cfg, controlnet):
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    lin = lambda o, i: o * i + o
    double = 2 * lin(6 * d, d) + 8 * lin(d, d) + 4 * 128 + 2 * (lin(4 * d, d) + lin(d, 4 * d))
    single = lin(3 * d, d) + lin(4 * d, d) + lin(d, 5 * d) + 3 * lin(d, d) + 2 * 128
    n = lin(d, 64) + lin(d, 4096) + (2 + (1 if cfg["guidance_embeds"] else 0) - 1) * 0
    emb = (2 if cfg["guidance_embeds"] else 1) * (lin(d, 256) + lin(d, d)) + lin(d, 768) + lin(d, d)
    n += emb + cfg["num_layers"] * double + cfg["num_single_layers"] * single
    if controlnet:
        n += (cfg["num_layers"] + cfg["num_single_layers"]) * lin(d, d) + lin(d, 64 + cfg["extra_condition_channels"])
    else:
        n += lin(2 * d, d) + lin(64, d)
    return double, single, n


def test_parameter_counts():
    double, single, total = _count(orc.FLUX_DEV_CFG, False)
    assert double == 339_831_296 and single == 141_591_808 and total == 11_901_408_320
    _, _, cn = _count(orc.REPTEXT_CN_CFG, True)
    assert abs(cn / 1e9 - 2.1411) < 5e-4 and abs(cn * 2 / 1e9 - 4.282) < 1e-3      # 4.28 GB checkpoint (NB:419)
    small = dict(orc.FLUX_DEV_CFG, num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=4096)
    p = orc.init_mmdit_params(dict(small), 0)
    d = 256
    assert p["transformer_blocks.0.norm1.linear.weight"].shape == (6 * d, d)
    assert p["single_transformer_blocks.0.proj_out.weight"].shape == (d, 5 * d)


def test_rope_and_timestep_conventions():
    ids = torch.tensor([[0.0, 0.0, 0.0], [0.0, 3.0, 5.0]])
    cos, sin = orc.rope_table(ids)
    assert cos.shape == (2, 128) and torch.all(cos[0] == 1) and torch.all(sin[0] == 0)      # text ids -> identity rotation
    assert torch.equal(cos[1, 0::2], cos[1, 1::2])                                             # each frequency repeated twice
    assert abs(float(cos[1, 16]) - math.cos(3.0)) < 1e-6 and abs(float(sin[1, 16 + 56]) - math.sin(5.0)) < 1e-6
    x = torch.randn(1, 2, 1, 128)
    y = orc.apply_rope(x, cos, sin)
    assert torch.allclose(y[0, 0], x[0, 0])
    assert abs(float(y[0, 1, 0, 16]) - float(x[0, 1, 0, 16] * math.cos(3.0) - x[0, 1, 0, 17] * math.sin(3.0))) < 1e-5
    e = orc.timestep_embedding(torch.tensor([0.0, 2.0]))
    assert e.shape == (2, 256) and torch.all(e[0, :128] == 1) and torch.all(e[0, 128:] == 0)     # [cos | sin]
    assert abs(float(e[1, 0]) - math.cos(2.0)) < 1e-6 and abs(float(e[1, 128]) - math.sin(2.0)) < 1e-6


def test_zero_controlnet_equals_plain_flux_and_masking():
    """§8c(6): zero-initialised zero-linears -> all-zero residuals -> identical transformer output (oracle level)."""
    cfg_t = dict(orc.FLUX_DEV_CFG, num_layers=1, num_single_layers=1, num_attention_heads=1, joint_attention_dim=64, pooled_projection_dim=32)
    cfg_c = dict(cfg_t, num_single_layers=0, extra_condition_channels=64)
    tp, cp = orc.init_mmdit_params(cfg_t, 1), orc.init_mmdit_params(cfg_c, 2, controlnet=True)
    for k in list(cp):
        if k.startswith("controlnet_blocks"):
            cp[k] = torch.zeros_like(cp[k])
    g = torch.Generator().manual_seed(0)
    N, T = 16, 8
    lat, cond, pe, pooled = torch.randn(1, N, 64, generator=g), torch.randn(1, N, 128, generator=g), torch.randn(1, T, 64, generator=g), torch.randn(1, 32, generator=g)
    ids, tids = orc.latent_image_ids(8, 8), torch.zeros(T, 3)
    sig = orc.flow_sigmas(2, 0.5)
    a = orc.denoise_loop(tp, cfg_t, cp, cfg_c, lat, pe, pooled, [cond], [None], sig, ids, tids, 3.5)
    b = orc.denoise_loop(tp, cfg_t, None, None, lat, pe, pooled, [], [], sig, ids, tids, 3.5)
    assert torch.allclose(a, b, atol=1e-6)
    # conditioning_step = 0 disables the tower too (Q3)
    cp2 = orc.init_mmdit_params(cfg_c, 2, controlnet=True)
    c = orc.denoise_loop(tp, cfg_t, cp2, cfg_c, lat, pe, pooled, [cond], [None], sig, ids, tids, 3.5, conditioning_step=0)
    assert torch.allclose(c, b, atol=1e-6)
    d = orc.denoise_loop(tp, cfg_t, cp2, cfg_c, lat, pe, pooled, [cond], [torch.zeros(1, N, 1)], sig, ids, tids, 3.5)
    assert torch.allclose(d, b, atol=1e-6)                                                     # an all-zero regional mask removes it


def test_vae_oracle_shapes():
    cfg = dict(vorc.FLUX_VAE_CFG, block_out_channels=(32, 32, 64, 64))
    p = vorc.init_vae_params(cfg, 0)
    img = vorc.decode(p, cfg, torch.randn(1, 16, 4, 4))
    assert img.shape == (1, 3, 32, 32)
    mean, logvar = vorc.encode_moments(p, cfg, torch.rand(1, 3, 32, 32) * 2 - 1)
    assert mean.shape == (1, 16, 4, 4) and float(logvar.max()) <= 20.0


def test_storage_precision_mode_is_the_bf16_floor():
    """`stored_as(bfloat16)` (the oracle at the HIP path's storage precision) must be off by default, deterministic, and sit
    one bf16 rounding floor away from the fp32 oracle — the figure the GPU-vs-fp32-oracle errors in DESIGN.md §4 reproduce."""
    cfg = dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128, num_attention_heads=4,
               joint_attention_dim=256, pooled_projection_dim=64, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
    tp = orc.init_mmdit_params(cfg, seed=1)
    g = torch.Generator().manual_seed(1)
    r = lambda *s: torch.randn(*s, generator=g).to(torch.bfloat16).float()
    B, T, h2, w2 = 1, 64, 16, 24
    N = (h2 // 2) * (w2 // 2)
    args = (tp, cfg, r(B, N, 64), r(B, T, 256), r(B, 64), torch.full((B,), 0.622459), orc.latent_image_ids(h2, w2), torch.zeros(T, 3))
    a = orc.transformer_forward(*args, guidance=torch.full((B,), 3.5))
    with orc.stored_as(torch.bfloat16):
        b = orc.transformer_forward(*args, guidance=torch.full((B,), 3.5))
        b2 = orc.transformer_forward(*args, guidance=torch.full((B,), 3.5))
    a2 = orc.transformer_forward(*args, guidance=torch.full((B,), 3.5))
    assert torch.equal(a, a2) and torch.equal(b, b2)
    floor = float((a - b).norm() / a.norm())
    assert 5e-4 < floor < 6e-3, floor


def test_oracle_primitives_match_torch_functional():
    """The torch ops the reference's path bottoms out in (via diffusers: SURVEY.md §2, last table) ARE present in this container.
    The oracle's hand-written primitives must agree with them — this pins the primitive level against the real third-party code:
    LayerNorm(eps 1e-6, no affine), RMSNorm(weight, eps 1e-6), GELU(tanh), SiLU, softmax attention with scale 1/sqrt(Dh),
    GroupNorm / conv2d / nearest-2x as the VAE oracle uses them."""
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 17, 256, generator=g, dtype=torch.float64)
    assert torch.allclose(orc.layer_norm(x), F.layer_norm(x, (256,), eps=1e-6), atol=1e-12)
    w = torch.randn(128, generator=g, dtype=torch.float64)
    h = torch.randn(2, 9, 4, 128, generator=g, dtype=torch.float64)
    assert torch.allclose(orc.rms_norm(h, w), F.rms_norm(h, (128,), w, eps=1e-6), atol=1e-12)
    assert torch.allclose(orc.gelu_tanh(x), F.gelu(x, approximate="tanh"), atol=1e-12)
    assert torch.allclose(orc.silu(x), F.silu(x), atol=1e-12)
    q, k, v = (torch.randn(2, 33, 4, 128, generator=g, dtype=torch.float64) for _ in range(3))
    sdpa = F.scaled_dot_product_attention(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3))   # default scale 1/sqrt(128)
    assert torch.allclose(orc.attention(q, k, v), sdpa.permute(0, 2, 1, 3).reshape(2, 33, 512), atol=1e-10)
    # RoPE as diffusers' apply_rotary_emb(use_real=True, use_real_unbind_dim=-1): x*cos + stack([-x_imag, x_real])*sin
    cos, sin = orc.rope_table(torch.tensor([[0.0, 3.0, 5.0], [0.0, 1.0, 2.0]]))
    xr = torch.randn(1, 2, 1, 128, generator=g)
    x_real, x_imag = xr.reshape(1, 2, 1, 64, 2).unbind(-1)
    rot = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    assert torch.allclose(orc.apply_rope(xr, cos, sin), xr * cos[None, :, None, :] + rot * sin[None, :, None, :], atol=1e-6)
    # each RoPE frequency appears twice, consecutively (repeat_interleave, CN:316-317 / A.5)
    assert torch.equal(cos[:, 0::2], cos[:, 1::2]) and torch.equal(sin[:, 0::2], sin[:, 1::2])


def test_streamed_params_feed_the_oracle_identically():
    """oracle/streamed.py: weights fetched one tensor at a time from another state dict (bf16, as the GPU models hold them)
    give the oracle bit-identical results to a resident fp32 dict of the same values; config1_case is BASELINE configs[0]."""
    import pytest

    from oracle.streamed import StreamedParams, config1_case, config1_oracle

    cfg_t = dict(patch_size=1, in_channels=64, num_layers=1, num_single_layers=2, attention_head_dim=128, num_attention_heads=2,
                 joint_attention_dim=64, pooled_projection_dim=32, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
    cfg_c = dict(cfg_t, num_single_layers=0, extra_condition_channels=64)
    tp, cp = orc.init_mmdit_params(cfg_t, 3), orc.init_mmdit_params(cfg_c, 4, controlnet=True)
    case = config1_case(T=32, joint_dim=64, pooled_dim=32)
    assert case["N"] == 256 and case["sigmas"].tolist() == pytest.approx([1.0, 0.622459, 0.0], abs=1e-6)
    assert 0 < float(case["region_mask"].sum()) < case["N"]
    ref = config1_oracle(tp, cfg_t, cp, cfg_c, case)
    sp_t = StreamedParams({k: v.to(torch.bfloat16) for k, v in tp.items()})
    sp_c = StreamedParams({k: v.to(torch.bfloat16) for k, v in cp.items()})
    got = config1_oracle(sp_t, cfg_t, sp_c, cfg_c, case)
    assert torch.equal(got, ref) and sp_t.bytes_streamed > 0 and len(sp_t) == len(tp) and "x_embedder.weight" in sp_t
    assert sp_t.get("no.such.bias") is None


def test_mx_block_quantisation_known_answers():
    """The E8M0 rule of the "mx" level (csrc/rt_common.h: rt_mx_scale_byte; oracle.flux_oracle.mx_scale_byte / quant_mx_e4m3): the
    smallest power of two 2^(s-127) with amax <= 448 * 2^(s-127), from closed-form cases; the quantised block never exceeds 448,
    de-quantises to within 2^-4 of every element (or half a subnormal step of the block), and re-quantising its own output is exact."""
    import torch
    from oracle import flux_oracle as orc

    amax = torch.tensor([0.0, 1e-45, 448.0, 448.0001, 449.0, 511.9, 512.0, 896.0, 896.1, 1.75, 1.7500001, 1.0, 3.5, 2.0 ** -9 * 1.75, 1e38])
    want = torch.tensor([1, 1, 127, 128, 128, 128, 128, 128, 129, 119, 120, 119, 120, 110, 245], dtype=torch.int32)
    assert torch.equal(orc.mx_scale_byte(amax), want)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(6, 512, generator=g) * torch.logspace(-6, 5, 6)[:, None]
    x[2, 32:64] = 0
    q, sb = orc.quant_mx_e4m3(x, return_parts=True)
    assert float(q.float().abs().max()) <= 448.0 and sb.shape == (6, 16)
    assert int(sb[2, 1]) == 1 and bool((q.float()[2, 32:64] == 0).all())
    d = orc.quant_mx_e4m3(x)
    step = torch.exp2(sb.float() - 127.0 - 10.0).repeat_interleave(32, dim=-1)
    assert bool(((d - x).abs() <= x.abs() * 2.0 ** -4 + step).all())
    assert torch.equal(orc.quant_mx_e4m3(d), d)                     # idempotent
    # every block uses the top binade of e4m3 it can: its largest element lands in [224, 448]
    top = q.float().abs().reshape(6, 16, 32).amax(dim=-1)
    nz = x.abs().reshape(6, 16, 32).amax(dim=-1) > 0
    assert bool(((top >= 224.0) & (top <= 448.0))[nz].all())
