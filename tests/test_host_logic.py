"""CPU tests of the host-side mirror of the reference interface: signatures, error behaviour, schedule, state-dict
layout, checkpoint IO, drop-in module names. No kernels run here."""
import inspect
import os

import numpy as np
import pytest
import torch

from oracle import flux_oracle as orc
from oracle import vae_oracle as vorc

SMALL_T = dict(patch_size=1, in_channels=64, num_layers=1, num_single_layers=1, attention_head_dim=128, num_attention_heads=1,
               joint_attention_dim=64, pooled_projection_dim=32, guidance_embeds=True, axes_dims_rope=(16, 56, 56))
SMALL_CN = dict(SMALL_T, num_single_layers=0, extra_condition_channels=64)


def test_dropin_module_names_and_call_signature():
    """infer.py:2-3 / infer_inpaint.py:3-4 import lines and the kwargs infer.py:117-130 passes."""
    from controlnet_flux import FluxControlNetModel, FluxControlNetOutput, FluxMultiControlNetModel  # noqa: F401
    from pipeline_flux_controlnet import FluxControlNetPipeline

    sig = inspect.signature(FluxControlNetPipeline.__call__)
    expected = ["prompt", "prompt_2", "height", "width", "num_inference_steps", "timesteps", "guidance_scale", "control_guidance_start",
                "control_guidance_end", "control_image", "control_mode", "controlnet_conditioning_scale", "controlnet_conditioning_step",
                "num_images_per_prompt", "generator", "latents", "prompt_embeds", "pooled_prompt_embeds", "output_type", "return_dict",
                "joint_attention_kwargs", "callback_on_step_end", "callback_on_step_end_tensor_inputs", "max_sequence_length",
                "control_mask", "control_position", "control_glyph"]
    assert list(sig.parameters)[1:] == expected                                      # PIPE:751-781 order
    d = {k: v.default for k, v in sig.parameters.items()}
    assert d["num_inference_steps"] == 28 and d["guidance_scale"] == 7.0 and d["controlnet_conditioning_step"] == 30   # Q10
    assert d["output_type"] == "pil" and d["max_sequence_length"] == 512
    fsig = inspect.signature(FluxControlNetModel.forward)
    assert list(fsig.parameters)[1:13] == ["hidden_states", "controlnet_cond", "controlnet_mode", "conditioning_scale", "encoder_hidden_states",
                                            "pooled_projections", "timestep", "img_ids", "txt_ids", "guidance", "joint_attention_kwargs",
                                            "return_dict"]                           # CN:216-230
    csig = inspect.signature(FluxControlNetModel.__init__)
    assert csig.parameters["num_layers"].default == 19 and csig.parameters["num_single_layers"].default == 38   # CN:49-50
    assert csig.parameters["guidance_embeds"].default is False and csig.parameters["extra_condition_channels"].default == 0


def test_state_dict_keys_match_diffusers_layout_and_counts():
    from reptext_amd.config import flux_dev_transformer_config, reptext_controlnet_config
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.transformer import FluxTransformer2DModel
    from reptext_amd.vae import AutoencoderKL

    tr = FluxTransformer2DModel(**SMALL_T, device="cpu", dtype=torch.bfloat16)
    cn = FluxControlNetModel(**SMALL_CN, device="cpu", dtype=torch.bfloat16)
    assert set(tr.state_dict()) == set(orc.init_mmdit_params(SMALL_T, 0))
    assert set(cn.state_dict()) == set(orc.init_mmdit_params(SMALL_CN, 0, controlnet=True))
    vcfg = dict(vorc.FLUX_VAE_CFG, block_out_channels=(32, 32, 64, 64))
    vae = AutoencoderKL(**vcfg, device="cpu", dtype=torch.bfloat16)
    assert set(vae.state_dict()) == set(vorc.init_vae_params(vcfg, 0))
    count = lambda cls, cfg: sum(p.numel() for p in cls(**cfg, device="meta", dtype=torch.bfloat16).parameters())
    assert count(FluxTransformer2DModel, flux_dev_transformer_config()) == 11_901_408_320
    assert abs(count(FluxControlNetModel, reptext_controlnet_config()) / 1e9 - 2.1411) < 5e-4


def test_checkpoint_roundtrip_local_dir(tmp_path):
    from reptext_amd.controlnet import FluxControlNetModel

    cn = FluxControlNetModel(**SMALL_CN, device="cpu", dtype=torch.bfloat16)
    cn.load_state_dict(orc.init_mmdit_params(SMALL_CN, 3, controlnet=True))
    cn.save_pretrained(str(tmp_path / "cn"))
    cn2 = FluxControlNetModel.from_pretrained(str(tmp_path / "cn"), torch_dtype=torch.bfloat16)
    assert cn2.config.extra_condition_channels == 64 and cn2.config.num_layers == 1
    for (k, a), (_, b) in zip(sorted(cn.state_dict().items()), sorted(cn2.state_dict().items())):
        assert torch.equal(a, b), k
    with pytest.raises(OSError):
        FluxControlNetModel.from_pretrained("Shakker-Labs/RepText", torch_dtype=torch.bfloat16)     # hub id: offline -> loud error


def _cpu_pipe():
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.transformer import FluxTransformer2DModel
    from reptext_amd.vae import AutoencoderKL

    tr = FluxTransformer2DModel(**SMALL_T, device="cpu", dtype=torch.bfloat16)
    cn = FluxControlNetModel(**SMALL_CN, device="cpu", dtype=torch.bfloat16)
    vae = AutoencoderKL(**dict(vorc.FLUX_VAE_CFG, block_out_channels=(32, 32, 64, 64)), device="cpu", dtype=torch.bfloat16)
    return FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), vae, None, None, None, None, tr, cn)


def test_pipeline_host_helpers_and_errors():
    pipe = _cpu_pipe()
    assert pipe.vae_scale_factor == 16 and pipe.default_sample_size == 64 and pipe.tokenizer_max_length == 77       # PIPE:219-226
    pe = torch.zeros(1, 8, 64)
    with pytest.raises(ValueError, match="divisible by 8"):
        pipe.check_inputs(None, None, 250, 256, prompt_embeds=pe, pooled_prompt_embeds=pe)
    with pytest.raises(ValueError, match="Cannot forward both"):
        pipe.check_inputs("a", None, 256, 256, prompt_embeds=pe, pooled_prompt_embeds=pe)
    with pytest.raises(ValueError, match="Provide either"):
        pipe.check_inputs(None, None, 256, 256)
    with pytest.raises(ValueError, match="pooled_prompt_embeds"):
        pipe.check_inputs(None, None, 256, 256, prompt_embeds=pe)
    with pytest.raises(ValueError, match="max_sequence_length"):
        pipe.check_inputs("a", None, 256, 256, max_sequence_length=513)
    with pytest.raises(ValueError, match="callback_on_step_end_tensor_inputs"):
        pipe.check_inputs("a", None, 256, 256, callback_on_step_end_tensor_inputs=["nope"])
    with pytest.raises(ValueError, match="without text encoders"):
        pipe.encode_prompt("a street sign", None)
    x = torch.randn(2, 16, 8, 12)
    packed = pipe._pack_latents(x, 2, 16, 8, 12)
    assert torch.equal(packed, orc.pack_latents(x))
    assert torch.equal(pipe._unpack_latents(packed, 64, 96, 16), x)
    assert torch.equal(pipe._prepare_latent_image_ids(1, 8, 12, "cpu", torch.float32), orc.latent_image_ids(8, 12))
    lat, ids = pipe.prepare_latents(2, 16, 128, 192, torch.float32, "cpu", torch.Generator().manual_seed(0))
    assert lat.shape == (2, 8 * 12, 64) and ids.shape == (96, 3)
    with pytest.raises(ValueError, match="list of generators"):
        pipe.prepare_latents(2, 16, 128, 192, torch.float32, "cpu", [torch.Generator()])
    from PIL import Image

    m = np.zeros([64, 64], dtype=np.uint8); m[16:48, 16:48] = 255
    masks = pipe._region_masks([Image.fromarray(m)], "cpu", torch.float32)
    ref = torch.nn.functional.interpolate(torch.from_numpy(m)[None, None].float() / 255.0, scale_factor=1 / 16, mode="bilinear").reshape(1, -1, 1)
    assert torch.equal(masks[0], ref) and masks[0].shape == (1, 16, 1)
    # the hot path itself refuses to run on CPU (no fallback)
    with pytest.raises(RuntimeError):
        pipe(prompt_embeds=torch.zeros(1, 8, 64, dtype=torch.bfloat16), pooled_prompt_embeds=torch.zeros(1, 32, dtype=torch.bfloat16),
             height=64, width=64, num_inference_steps=1, control_image=[torch.zeros(1, 16, 128, dtype=torch.bfloat16)], output_type="latent")


def test_scheduler_matches_known_schedule_and_reference_call_pattern():
    from reptext_amd.pipeline import retrieve_timesteps
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler, calculate_shift

    s = FlowMatchEulerDiscreteScheduler()
    mu = calculate_shift(4096, s.config.base_image_seq_len, s.config.max_image_seq_len, s.config.base_shift, s.config.max_shift)
    assert abs(mu - 1.15) < 1e-12 and abs(calculate_shift(4096) - 1.16) < 1e-12
    ts, n = retrieve_timesteps(s, 28, "cpu", None, np.linspace(1.0, 1 / 28, 28), mu=mu)
    assert n == 28 and s.order == 1 and torch.allclose(s.sigmas, orc.flow_sigmas(28, 1.15), atol=1e-7)
    assert torch.allclose(ts, s.sigmas[:-1] * 1000)
    with pytest.raises(ValueError, match="Only one of"):
        retrieve_timesteps(s, 28, "cpu", [1.0], [1.0], mu=mu)
    with pytest.raises(ValueError, match="mu"):
        s.set_timesteps(4)


def test_image_processor():
    from PIL import Image

    from reptext_amd.image_processor import VaeImageProcessor

    ip = VaeImageProcessor(vae_scale_factor=16)
    img = Image.fromarray((np.arange(64 * 64 * 3) % 256).astype(np.uint8).reshape(64, 64, 3))
    x = ip.preprocess(img, height=64, width=64)
    assert x.shape == (1, 3, 64, 64) and float(x.min()) >= -1 and float(x.max()) <= 1
    assert abs(float(x[0, 0, 0, 0]) - (2 * 0 / 255 - 1)) < 1e-6
    g = ip.preprocess(Image.fromarray(np.full((32, 32), 255, np.uint8)), height=64, width=64)
    assert g.shape == (1, 1, 64, 64) and float(g.min()) > 0.99
    u8 = torch.randint(0, 255, (2, 8, 8, 3), dtype=torch.uint8)
    pil = ip.postprocess_u8(u8, "pil")
    assert len(pil) == 2 and pil[0].size == (8, 8)
    assert np.allclose(ip.postprocess_u8(u8, "np"), u8.numpy() / 255.0)


def test_randn_tensor_device_rule():
    from reptext_amd.utils import randn_tensor

    a = randn_tensor((2, 4), torch.Generator().manual_seed(1), "cpu", torch.float32)
    b = torch.randn((2, 4), generator=torch.Generator().manual_seed(1))
    assert torch.equal(a, b)
    c = randn_tensor((2, 4), [torch.Generator().manual_seed(1), torch.Generator().manual_seed(2)], "cpu", torch.float32)
    assert torch.equal(c[1:], torch.randn((1, 4), generator=torch.Generator().manual_seed(2)))
    with pytest.raises(ValueError):
        randn_tensor((3, 4), [torch.Generator()], "cpu", torch.float32)


def test_inpaint_dropin_signature_and_mask_processor():
    """infer_inpaint.py:4 import + the kwargs infer_inpaint.py:132-151 passes; INP:846-883 order and defaults."""
    from pipeline_flux_controlnet_inpaint import FluxControlNetPipeline as Inpaint

    names = list(inspect.signature(Inpaint.__call__).parameters)[1:]
    assert names[:5] == ["prompt", "prompt_2", "true_guidance_scale", "negative_prompt", "negative_prompt_2"]
    for k in ("control_image_inpaint", "control_mask_inpaint", "controlnet_conditioning_scale_inpaint", "control_glyph", "control_mask"):
        assert k in names
    d = {k: v.default for k, v in inspect.signature(Inpaint.__call__).parameters.items()}
    assert d["true_guidance_scale"] == 3.5 and d["guidance_scale"] == 7.0 and d["num_inference_steps"] == 28
    assert list(inspect.signature(Inpaint.__init__).parameters)[1:] == ["scheduler", "vae", "text_encoder", "tokenizer", "text_encoder_2",
                                                                        "tokenizer_2", "transformer", "controlnet", "controlnet_inpaint"]
    from reptext_amd.image_processor import VaeImageProcessor
    from PIL import Image

    mp = VaeImageProcessor(vae_scale_factor=16, do_normalize=False, do_binarize=True, do_convert_grayscale=True)       # INP:228-234
    m = np.zeros((32, 32, 3), np.uint8); m[:16] = 200
    x = mp.preprocess(Image.fromarray(m), height=32, width=32)
    assert x.shape == (1, 1, 32, 32) and set(x.unique().tolist()) == {0.0, 1.0} and float(x[0, 0, 0, 0]) == 1.0


def test_bench_work_model_matches_survey_8d():
    """bench.py's algorithmic FLOPs per image = SURVEY.md §8d: block 24·S·d² + 4·S²·d, transformer 74.378 T, tower 8.309 T,
    28 steps = 2.3152 PFLOP (the VAE decode's 10.47 T is timed but not counted); C5 shape 5.968 PFLOP."""
    import bench
    from reptext_amd.config import flux_dev_transformer_config, reptext_controlnet_config

    ct, cc = flux_dev_transformer_config(), reptext_controlnet_config()
    f = bench.flops_per_image(1024, 1024, 28, 1, ct, cc)
    assert abs(f / 1e15 - 2.3152) < 1e-4
    f5 = bench.flops_per_image(1536, 1536, 28, 1, ct, cc)
    assert abs(f5 / 1e15 - 5.968) < 1e-3
    # usable_cores never exceeds the cap the CPU baseline is sized for
    assert 1 <= bench.usable_cores() <= 16


def test_text_encoder_state_dict_keys_match_transformers():
    """reptext_amd.text_encoders holds its weights under transformers' own state-dict keys (CPU, no kernels involved)."""
    from transformers import CLIPTextConfig, T5Config
    from transformers import CLIPTextModel as HFCLIP
    from transformers import T5EncoderModel as HFT5
    from reptext_amd.text_encoders import CLIPTextModel, T5EncoderModel

    cfg = T5Config(vocab_size=64, d_model=128, d_kv=64, d_ff=192, num_layers=2, num_heads=2, feed_forward_proj="gated-gelu", is_encoder_decoder=False)
    mine = T5EncoderModel(vocab_size=64, d_model=128, d_kv=64, d_ff=192, num_layers=2, num_heads=2, dtype=torch.bfloat16)
    hf_keys = {k for k in HFT5(cfg).state_dict() if k != "encoder.embed_tokens.weight"}
    assert set(mine.state_dict()) == hf_keys
    c = CLIPTextConfig(vocab_size=64, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=16)
    mc = CLIPTextModel(vocab_size=64, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=16,
                       dtype=torch.bfloat16)
    hf = {k if k.startswith("text_model.") else "text_model." + k for k in HFCLIP(c).state_dict() if not k.endswith("position_ids")}
    assert set(mc.state_dict()) == hf
    with pytest.raises(RuntimeError):            # CPU tensors: no fallback
        mine(torch.zeros(1, 64, dtype=torch.long))


def test_t5_relative_position_bucket_matches_transformers_cpu():
    """Host-side piece of the T5 encoder, pinned against the real implementation (transformers is importable here)."""
    from transformers.models.t5.modeling_t5 import T5Attention
    from reptext_amd.text_encoders import t5_relative_position_bucket

    rel = torch.arange(-700, 700)[None, :] - torch.arange(0, 5)[:, None]
    for nb, md in ((32, 128), (64, 256), (16, 64)):
        assert torch.equal(t5_relative_position_bucket(rel, nb, md), T5Attention._relative_position_bucket(rel, True, nb, md))


# ------------------------------------------------------------------------------------------- checkpoint formats
def _fake_hub_cache(tmp_path, monkeypatch, repo_id, populate):
    """<cache>/models--ORG--NAME/{refs/main, snapshots/<commit>/...}: the layout huggingface_hub leaves behind (offline)."""
    cache = tmp_path / "hf_home" / "hub"
    commit = "0123456789abcdef0123456789abcdef01234567"
    snap = cache / ("models--" + repo_id.replace("/", "--")) / "snapshots" / commit
    snap.mkdir(parents=True)
    refs = snap.parent.parent / "refs"
    refs.mkdir()
    (refs / "main").write_text(commit)
    populate(str(snap))
    for env in ("HF_HUB_CACHE", "HUGGINGFACE_HUB_CACHE"):
        monkeypatch.delenv(env, raising=False)
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf_home"))
    return str(snap)


def test_hub_ids_resolve_through_the_local_hub_cache(tmp_path, monkeypatch):
    """infer.py:30-31 passes hub ids. Offline they must resolve through $HF_HOME/hub/models--ORG--NAME/snapshots/<commit>,
    and fail loudly (naming the paths tried) when no snapshot is cached."""
    from reptext_amd.controlnet import FluxControlNetModel
    from reptext_amd.modules import resolve_model_path

    cn = FluxControlNetModel(**SMALL_CN, device="cpu", dtype=torch.bfloat16)
    cn.load_state_dict(orc.init_mmdit_params(SMALL_CN, 5, controlnet=True))
    snap = _fake_hub_cache(tmp_path, monkeypatch, "Shakker-Labs/RepText", cn.save_pretrained)
    assert resolve_model_path("Shakker-Labs/RepText") == snap
    cn2 = FluxControlNetModel.from_pretrained("Shakker-Labs/RepText", torch_dtype=torch.bfloat16)       # infer.py:30, unchanged
    for (k, a), (_, b) in zip(sorted(cn.state_dict().items()), sorted(cn2.state_dict().items())):
        assert torch.equal(a, b), k
    with pytest.raises(OSError, match="models--black-forest-labs--FLUX.1-dev"):
        resolve_model_path("black-forest-labs/FLUX.1-dev")
    assert resolve_model_path(snap) == snap                                                            # a directory stays a directory
    monkeypatch.setenv("HF_HUB_CACHE", str(tmp_path / "elsewhere"))                                     # takes precedence over HF_HOME
    with pytest.raises(OSError):
        resolve_model_path("Shakker-Labs/RepText")


def test_sharded_checkpoint_with_index_roundtrip(tmp_path):
    """FLUX.1-dev's transformer ships as three shards + diffusion_pytorch_model.safetensors.index.json (NB:2116-2118)."""
    import json

    from reptext_amd.transformer import FluxTransformer2DModel

    tr = FluxTransformer2DModel(**SMALL_T, device="cpu", dtype=torch.bfloat16)
    tr.load_state_dict(orc.init_mmdit_params(SMALL_T, 7))
    total = sum(v.numel() * 2 for v in tr.state_dict().values())
    d = tmp_path / "tr"
    tr.save_pretrained(str(d), max_shard_bytes=total // 2)                # greedy packing of whole tensors: 3 shards here
    files = sorted(os.listdir(d))
    shards = [f for f in files if f.endswith(".safetensors")]
    assert len(shards) >= 3 and shards[0] == f"diffusion_pytorch_model-00001-of-{len(shards):05d}.safetensors"
    idx = json.load(open(d / "diffusion_pytorch_model.safetensors.index.json"))
    assert set(idx["weight_map"]) == set(tr.state_dict()) and idx["metadata"]["total_size"] == total
    tr2 = FluxTransformer2DModel.from_pretrained(str(d), torch_dtype=torch.bfloat16)
    for (k, a), (_, b) in zip(sorted(tr.state_dict().items()), sorted(tr2.state_dict().items())):
        assert torch.equal(a, b), k
    os.remove(d / shards[1])                                              # a shard the index names is missing: loud error
    with pytest.raises(OSError, match="missing"):
        FluxTransformer2DModel.from_pretrained(str(d), torch_dtype=torch.bfloat16)


def test_pipeline_from_pretrained_reads_model_index(tmp_path, monkeypatch):
    """infer.py:31-33: FluxControlNetPipeline.from_pretrained(<hub id>, controlnet=..., torch_dtype=...). model_index.json
    decides which components exist; text encoders marked [null, null] stay None."""
    import json

    from reptext_amd.pipeline import FluxControlNetPipeline

    pipe = _cpu_pipe()
    pipe.transformer.load_state_dict(orc.init_mmdit_params(SMALL_T, 11))
    pipe.vae.load_state_dict({k: v.to(torch.bfloat16) for k, v in vorc.init_vae_params(dict(vorc.FLUX_VAE_CFG, block_out_channels=(32, 32, 64, 64)), 2).items()})
    snap = _fake_hub_cache(tmp_path, monkeypatch, "black-forest-labs/FLUX.1-dev", pipe.save_pretrained)
    idx = json.load(open(os.path.join(snap, "model_index.json")))
    assert idx["_class_name"] == "FluxControlNetPipeline" and idx["text_encoder"] == [None, None] and idx["transformer"][1] == "FluxTransformer2DModel"
    p2 = FluxControlNetPipeline.from_pretrained("black-forest-labs/FLUX.1-dev", controlnet=pipe.controlnet, torch_dtype=torch.bfloat16)
    assert p2.controlnet is pipe.controlnet and p2.text_encoder is None and p2.tokenizer_2 is None
    assert dict(p2.scheduler.config) == dict(pipe.scheduler.config)
    for (k, a), (_, b) in zip(sorted(pipe.transformer.state_dict().items()), sorted(p2.transformer.state_dict().items())):
        assert torch.equal(a, b), k
    for (k, a), (_, b) in zip(sorted(pipe.vae.state_dict().items()), sorted(p2.vae.state_dict().items())):
        assert torch.equal(a, b), k
    os.rename(os.path.join(snap, "vae"), os.path.join(snap, "vae_gone"))
    with pytest.raises(OSError, match="vae"):
        FluxControlNetPipeline.from_pretrained("black-forest-labs/FLUX.1-dev", controlnet=pipe.controlnet)


def test_reference_bf16_scalar_mode_and_static_shift_grid():
    """the reference's bf16 run rounds t, t/1000 and guidance·1000 to bf16 (PIPE:1025,1048; CN:282-284);
    the default path keeps them exact. Both modes exist in the pipeline and in the oracle and agree on the values."""
    from reptext_amd import mmdit
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler

    pipe = _cpu_pipe()
    assert pipe._model_timestep(967.3) == 967.3 / 1000.0
    pipe.reference_bf16_scalars = True
    mt = pipe._model_timestep(967.3)
    assert mt == float((torch.tensor(967.3).to(torch.bfloat16) / 1000).float()) and mt != 967.3 / 1000.0
    assert float(mmdit.bf16_round_trip_x1000(torch.tensor([mt]))[0]) == 968.0                       # 967.3 enters the sinusoid as 968
    assert float(mmdit.bf16_round_trip_x1000(torch.tensor([3.5]))[0]) == 3504.0                     # guidance 3.5 -> 3504
    with orc.reference_bf16_scalars():
        assert float(orc._x1000(orc._model_t(torch.tensor(967.3)))) == 968.0 and float(orc._x1000(torch.tensor(3.5))) == 3504.0
    assert float(orc._x1000(orc._model_t(torch.tensor(967.3)))) == pytest.approx(967.3, rel=1e-6)
    pipe.reference_bf16_scalars = False
    pipe._model_timestep(1.0)
    assert mmdit.REF_BF16_SCALARS is False
    # static-shift branch (diffusers' default scheduler config): explicit sigmas, and the default grid — linspace between the
    # ALREADY shifted end points of the constructor's table, shifted once more (closed form; parity unpinned, diffusers absent)
    s = FlowMatchEulerDiscreteScheduler(use_dynamic_shifting=False, shift=3.0)
    s.set_timesteps(sigmas=[1.0, 0.5])
    assert s.sigmas.tolist() == pytest.approx([1.0, 0.75, 0.0])
    s.set_timesteps(4)
    sh = lambda x: 3.0 * x / (1 + 2.0 * x)
    grid = np.linspace(1000.0, sh(1e-3) * 1000.0, 4) / 1000.0
    assert s.sigmas.tolist() == pytest.approx([sh(g) for g in grid] + [0.0], rel=1e-6)
    assert s.timesteps.tolist() == pytest.approx([1000 * sh(g) for g in grid], rel=1e-6) and s.num_inference_steps == 4
