"""Prompt encoders on the HIP kernels vs the REAL transformers classes (call sites PIPE:232-347).

Unlike the diffusers-side math, the third-party package holding this arithmetic IS importable in the build container
(transformers), so parity here is pinned: the reference's own dependency is instantiated on the CPU in fp32 with
random weights (rounded to bf16 so both sides see identical values), its state dict is loaded into reptext_amd.text_encoders, and
the outputs are compared. Tolerance: bf16 storage between stages against an fp32 run -> rel-L2 <= 1e-2, stated per assert."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def _round_bf16_(model):
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    return model


def test_relative_position_bucket_matches_transformers():
    from transformers.models.t5.modeling_t5 import T5Attention
    from reptext_amd.text_encoders import t5_relative_position_bucket

    rel = torch.arange(-600, 600)[None, :] - torch.arange(0, 3)[:, None]
    for nb, md in ((32, 128), (64, 256)):
        ref = T5Attention._relative_position_bucket(rel, bidirectional=True, num_buckets=nb, max_distance=md)
        assert torch.equal(t5_relative_position_bucket(rel, nb, md), ref)


@pytest.mark.parametrize("B,T", [(1, 64), (2, 128)])
def test_t5_encoder_vs_transformers(gpu, B, T):
    from transformers import T5Config
    from transformers import T5EncoderModel as HFT5
    from reptext_amd.text_encoders import T5EncoderModel

    torch.manual_seed(0)
    cfg = T5Config(vocab_size=512, d_model=256, d_kv=64, d_ff=640, num_layers=3, num_heads=4, relative_attention_num_buckets=32,
                   relative_attention_max_distance=128, feed_forward_proj="gated-gelu", dropout_rate=0.0, is_encoder_decoder=False, use_cache=False)
    hf = _round_bf16_(HFT5(cfg).eval())
    with torch.no_grad():          # default init leaves some tensors at tiny / unit scale: give every weight an exercised range
        for n, p in hf.named_parameters():
            if "layer_norm" in n:
                p.copy_((1.0 + 0.2 * torch.randn_like(p)).to(torch.bfloat16).float())
            elif "relative_attention_bias" in n:
                p.copy_((torch.randn_like(p)).to(torch.bfloat16).float())
    ids = torch.randint(0, 512, (B, T))
    with torch.no_grad():
        ref = hf(ids).last_hidden_state
    mine = T5EncoderModel(**{k: getattr(cfg, k) for k in ("vocab_size", "d_model", "d_kv", "d_ff", "num_layers", "num_heads",
                                                          "relative_attention_num_buckets", "relative_attention_max_distance",
                                                          "layer_norm_epsilon", "feed_forward_proj")}, device=gpu, dtype=torch.bfloat16)
    mine.load_state_dict(hf.state_dict(), strict=True)
    out = mine(ids.to(gpu), output_hidden_states=False)
    assert out[0].shape == (B, T, 256) and out[0].dtype == torch.bfloat16
    err = rel_l2(out[0].float().cpu(), ref)
    print(f"T5 encoder B={B} T={T}: rel-L2 {err:.3e} vs transformers fp32")
    assert err < 1e-2


@pytest.mark.parametrize("B", [1, 3])
def test_clip_text_model_vs_transformers(gpu, B):
    from transformers import CLIPTextConfig
    from transformers import CLIPTextModel as HFCLIP
    from reptext_amd.text_encoders import CLIPTextModel

    torch.manual_seed(1)
    cfg = CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                         max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=999, bos_token_id=998, pad_token_id=0)
    hf = _round_bf16_(HFCLIP(cfg).eval())
    with torch.no_grad():
        for n, p in hf.named_parameters():
            if "layer_norm" in n and n.endswith("weight"):
                p.copy_((1.0 + 0.2 * torch.randn_like(p)).to(torch.bfloat16).float())
            elif n.endswith("bias"):
                p.copy_((0.1 * torch.randn_like(p)).to(torch.bfloat16).float())
    ids = torch.randint(1, 990, (B, 77))
    eos_at = [20, 76, 5][:B]
    for b, e in enumerate(eos_at):                       # one EOS per row; everything after it is padding, as the tokenizer emits
        ids[b, e] = 999
        ids[b, e + 1:] = 0
    with torch.no_grad():
        r = hf(ids)
    mine = CLIPTextModel(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                         max_position_embeddings=77, hidden_act="quick_gelu", eos_token_id=999, layer_norm_eps=cfg.layer_norm_eps,
                         device=gpu, dtype=torch.bfloat16)
    mine.load_state_dict(hf.state_dict(), strict=True)
    o = mine(ids.to(gpu), output_hidden_states=False)
    assert o.pooler_output.shape == (B, 128)
    e_h, e_p = rel_l2(o.last_hidden_state.float().cpu(), r.last_hidden_state), rel_l2(o.pooler_output.float().cpu(), r.pooler_output)
    print(f"CLIP text B={B}: last_hidden_state {e_h:.3e}, pooler_output {e_p:.3e} vs transformers fp32")
    assert e_h < 1e-2 and e_p < 1e-2


def test_pipeline_encode_prompt_with_hip_encoders(gpu):
    """encode_prompt (PIPE:349-456) with the HIP encoders and stub tokenizers: shapes / dtypes / repeat semantics."""
    from reptext_amd.pipeline import FluxControlNetPipeline
    from reptext_amd.scheduler import FlowMatchEulerDiscreteScheduler
    from reptext_amd.text_encoders import CLIPTextModel, T5EncoderModel

    class Tok:
        def __init__(self, vocab, length, eos):
            self.vocab, self.model_max_length, self.eos = vocab, length, eos

        def __call__(self, prompt, padding=None, max_length=None, truncation=None, return_tensors=None, **kw):
            n = max_length or self.model_max_length
            ids = torch.zeros(len(prompt), n, dtype=torch.long)
            for i, p in enumerate(prompt):
                toks = [(ord(ch) % (self.vocab - 2)) + 1 for ch in p][: n - 1]
                ids[i, : len(toks)] = torch.tensor(toks)
                ids[i, len(toks)] = self.eos
            return type("Enc", (), {"input_ids": ids})()

    t5 = T5EncoderModel(vocab_size=512, d_model=256, d_kv=64, d_ff=640, num_layers=1, num_heads=4, device=gpu, dtype=torch.bfloat16)
    clip = CLIPTextModel(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2, eos_token_id=999,
                         device=gpu, dtype=torch.bfloat16)
    g = torch.Generator(device=gpu).manual_seed(0)
    for m in (t5, clip):
        for p in m.parameters():
            p.data.copy_(0.05 * torch.randn(p.shape, device=gpu, generator=g))
    pipe = FluxControlNetPipeline(FlowMatchEulerDiscreteScheduler(), None, clip, Tok(1000, 77, 999), t5, Tok(512, 512, 1), None, None)
    pe, pooled, tids = pipe.encode_prompt(["a street sign in city", "لافتة"], None, device=gpu, num_images_per_prompt=2, max_sequence_length=128)
    assert pe.shape == (4, 128, 256) and pooled.shape == (4, 128) and tids.shape == (128, 3)
    assert pe.dtype == torch.bfloat16 and torch.isfinite(pe.float()).all() and torch.isfinite(pooled.float()).all()
    assert torch.equal(pe[0], pe[1]) and not torch.equal(pe[0], pe[2])          # repeat per prompt, then the next prompt
