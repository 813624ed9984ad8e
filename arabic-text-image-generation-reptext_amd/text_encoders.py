"""Prompt encoders on the HIP kernels: T5-v1.1 encoder (FLUX's text_encoder_2: T5-XXL) and the CLIP text
model (text_encoder: CLIP-L), as `FluxControlNetPipeline` calls them (PIPE:232-347):

    prompt_embeds = text_encoder_2(input_ids, output_hidden_states=False)[0]              # [B, 512, 4096], no attention mask
    pooled        = text_encoder(input_ids, output_hidden_states=False).pooler_output     # [B, 768]

Same class names, constructor config keys and state-dict keys as `transformers`' T5EncoderModel / CLIPTextModel, so a local
snapshot's `config.json` + safetensors load unchanged; the arithmetic follows their published definitions (pre-norm T5 blocks with
T5LayerNorm, un-scaled dot products plus the shared bucketed relative-position bias, gated tanh-GELU feed-forward; pre-LN CLIP
blocks with a causal mask, learned positions, quick_gelu, final LayerNorm, pooled = hidden state at the EOS position).
`transformers` itself is importable in the build container, so — unlike the diffusers-side math — THIS part's parity is pinned:
tests/test_text_encoders_gpu.py compares against the real classes with shared random weights.

They run once per prompt, outside the denoising loop. Matrix work goes through rt_gemm_bf16 (fp32 residual stream); attention has
head dim 64 and a bias/mask, so it is assembled per (batch, head) from GEMM + rt_softmax_rows_bias + rt_transpose_bf16 + GEMM.
"""
from __future__ import annotations

import json
import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import native, ops
from .config import Config
from .modules import WeightsIO

BF16, F32 = torch.bfloat16, torch.float32


class _H(nn.Module):
    pass


class _W(nn.Module):
    """weight-only parameter holder (bias-free Linear / T5LayerNorm / Embedding)."""

    def __init__(self, *shape, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape, device=device, dtype=dtype), requires_grad=False)


class _WB(nn.Module):
    def __init__(self, out_f, in_f=None, device=None, dtype=None):
        super().__init__()
        shape = (out_f,) if in_f is None else (out_f, in_f)
        self.weight = nn.Parameter(torch.empty(*shape, device=device, dtype=dtype), requires_grad=False)
        self.bias = nn.Parameter(torch.empty(out_f, device=device, dtype=dtype), requires_grad=False)


class BaseModelOutput(tuple):
    """Tuple-like result with attribute access, the two ways the pipeline reads encoder outputs."""

    def __new__(cls, last_hidden_state, pooler_output=None):
        obj = super().__new__(cls, (last_hidden_state,) if pooler_output is None else (last_hidden_state, pooler_output))
        obj.last_hidden_state, obj.pooler_output = last_hidden_state, pooler_output
        return obj


def _stream():
    return ops._stream()


def _ids_i32(input_ids: torch.Tensor, device) -> torch.Tensor:
    return input_ids.to(device=device, dtype=torch.int32).contiguous()


def _attention_heads(q, k, v, out, H, bias, scale, T, Tp, scratch):
    """out[b, :, h*64:(h+1)*64] = softmax(scale * q_h k_hᵀ + bias[h or 0]) v_h for every (batch, head).

    q, k, v, out: [B, Tp, H*64] bf16 views (row stride = their own ld); bias f32 [Hb, T, T] with Hb in (1, H) or None.
    Tp = T rounded up to 64 (rows/keys >= T are padding: keys masked by writing zero probabilities, rows ignored)."""
    lib = native.load()
    B = q.shape[0]
    scores, probs, vt = scratch
    for b in range(B):
        for h in range(H):
            qh, kh, vh = q[b, :, h * 64 : (h + 1) * 64], k[b, :, h * 64 : (h + 1) * 64], v[b, :, h * 64 : (h + 1) * 64]
            ops.linear(qh, kh, scores)                                           # [Tp, Tp] f32 = q_h k_hᵀ
            bh = None if bias is None else bias[h if bias.shape[0] > 1 else 0]
            native.check("rt_softmax_rows_bias", lib.rt_softmax_rows_bias(
                scores.data_ptr(), Tp, None if bh is None else bh.data_ptr(), 0 if bh is None else bh.stride(0), probs.data_ptr(), Tp,
                T, T, Tp, float(scale), _stream()))
            native.check("rt_transpose_bf16", lib.rt_transpose_bf16(vh.data_ptr(), vt.data_ptr(), Tp, 64, vh.stride(0), Tp, _stream()))
            ops.linear(probs, vt, out[b, :, h * 64 : (h + 1) * 64])              # [Tp, 64] = P v_h


# =========================================================================================================== T5
def t5_relative_position_bucket(rel: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """Bidirectional bucketing of T5 (Raffel et al. 2020, mesh-tensorflow `_relative_position_bucket`): half the buckets per sign,
    exact up to num_buckets/4, logarithmic up to max_distance."""
    nb = num_buckets // 2
    ret = (rel > 0).to(torch.long) * nb
    n = rel.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).to(torch.long)
    large = torch.minimum(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


class T5EncoderModel(nn.Module, WeightsIO):
    config_name = "config.json"
    weights_name = "model.safetensors"

    def __init__(self, vocab_size: int = 32128, d_model: int = 4096, d_kv: int = 64, d_ff: int = 10240, num_layers: int = 24,
                 num_heads: int = 64, relative_attention_num_buckets: int = 32, relative_attention_max_distance: int = 128,
                 layer_norm_epsilon: float = 1e-6, feed_forward_proj: str = "gated-gelu", device=None, dtype=None, **unused):
        super().__init__()
        if d_kv != 64:
            raise ValueError("T5EncoderModel (HIP): d_kv must be 64 (T5 v1.1 / FLUX's T5-XXL)")
        if feed_forward_proj != "gated-gelu":
            raise ValueError("T5EncoderModel (HIP): only the v1.1 gated-gelu feed-forward is implemented")
        self.config = Config(vocab_size=vocab_size, d_model=d_model, d_kv=d_kv, d_ff=d_ff, num_layers=num_layers, num_heads=num_heads,
                             relative_attention_num_buckets=relative_attention_num_buckets,
                             relative_attention_max_distance=relative_attention_max_distance, layer_norm_epsilon=layer_norm_epsilon,
                             feed_forward_proj=feed_forward_proj)
        kw = dict(device=device, dtype=dtype)
        inner = num_heads * d_kv
        self.shared = _W(vocab_size, d_model, **kw)
        enc = _H()
        enc.block = nn.ModuleList()
        for i in range(num_layers):
            blk = _H()
            l0, l1 = _H(), _H()
            sa = _H()
            sa.q, sa.k, sa.v = _W(inner, d_model, **kw), _W(inner, d_model, **kw), _W(inner, d_model, **kw)
            sa.o = _W(d_model, inner, **kw)
            if i == 0:
                sa.relative_attention_bias = _W(relative_attention_num_buckets, num_heads, **kw)
            l0.SelfAttention, l0.layer_norm = sa, _W(d_model, **kw)
            ff = _H()
            ff.wi_0, ff.wi_1, ff.wo = _W(d_ff, d_model, **kw), _W(d_ff, d_model, **kw), _W(d_model, d_ff, **kw)
            l1.DenseReluDense, l1.layer_norm = ff, _W(d_model, **kw)
            blk.layer = nn.ModuleList([l0, l1])
            enc.block.append(blk)
        enc.final_layer_norm = _W(d_model, **kw)
        self.encoder = enc
        self._plans = None

    @property
    def dtype(self):
        return self.shared.weight.dtype

    @property
    def device(self):
        return self.shared.weight.device

    def _apply(self, fn, *a, **k):
        self._plans = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, sd, strict: bool = True, **kw):
        sd = {k: v for k, v in sd.items() if k != "encoder.embed_tokens.weight"}      # tied to `shared`
        self._plans = None
        return super().load_state_dict(sd, strict=strict, **kw)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, subfolder: Optional[str] = None, device=None, **unused):
        d = cls._resolve_dir(path, subfolder)
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = json.load(f)
        m = cls(**cfg, device=device or "cpu", dtype=torch_dtype or BF16)
        m.load_state_dict({k: v.to(torch_dtype or BF16) for k, v in cls._load_safetensors_dir(d).items()}, strict=True)
        return m

    def _ensure_plans(self):
        if self._plans is not None:
            return self._plans
        if self.dtype != BF16 or not self.shared.weight.is_cuda:
            raise RuntimeError("T5EncoderModel (HIP): bf16 on the GPU only; there is no CPU fallback")
        plans = []
        for blk in self.encoder.block:
            sa, ff = blk.layer[0].SelfAttention, blk.layer[1].DenseReluDense
            qkv = torch.cat([sa.q.weight.data, sa.k.weight.data, sa.v.weight.data], dim=0).contiguous()
            wi = torch.cat([ff.wi_1.weight.data, ff.wi_0.weight.data], dim=0).contiguous()      # [linear | gelu] halves
            plans.append((qkv, sa.o.weight.data, wi, ff.wo.weight.data, blk.layer[0].layer_norm.weight.data, blk.layer[1].layer_norm.weight.data))
        self._plans = plans
        self._bias_cache = {}
        return plans

    def _position_bias(self, T: int) -> torch.Tensor:
        """[H, T, T] f32: relative_attention_bias[bucket(key - query)], shared by all layers (it lives in block 0)."""
        hit = self._bias_cache.get(T)
        if hit is None:
            c = self.config
            ctx = torch.arange(T)[:, None]
            mem = torch.arange(T)[None, :]
            buckets = t5_relative_position_bucket(mem - ctx, c.relative_attention_num_buckets, c.relative_attention_max_distance)
            table = self.encoder.block[0].layer[0].SelfAttention.relative_attention_bias.weight.data.to(F32)     # [buckets, H]
            hit = table[buckets.to(table.device)].permute(2, 0, 1).contiguous()
            self._bias_cache = {T: hit}
        return hit

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask=None, output_hidden_states: bool = False, return_dict: bool = True, **unused):
        if attention_mask is not None:
            raise NotImplementedError("T5EncoderModel (HIP): the FLUX pipelines pass no attention mask (PIPE:287-289)")
        plans = self._ensure_plans()
        c = self.config
        dev = self.device
        lib = native.load()
        B, T = input_ids.shape
        if T % 64:
            raise ValueError("T5EncoderModel (HIP): sequence length must be a multiple of 64 (the pipelines pad to 512)")
        d, H, F_ = c.d_model, c.num_heads, c.d_ff
        inner = H * 64
        ids = _ids_i32(input_ids.reshape(-1), dev)
        emb = torch.empty(B * T, d, device=dev, dtype=BF16)
        native.check("rt_embedding_gather", lib.rt_embedding_gather(self.shared.weight.data_ptr(), d, ids.data_ptr(), emb.data_ptr(), d, B * T, d,
                                                                    c.vocab_size, _stream()))
        x = ops.to_f32(emb)                                                     # fp32 residual stream [B*T, d]
        xn = torch.empty(B * T, d, device=dev, dtype=BF16)
        qkv = torch.empty(B, T, 3 * inner, device=dev, dtype=BF16)
        att = torch.empty(B, T, inner, device=dev, dtype=BF16)
        hid = torch.empty(B * T, 2 * F_, device=dev, dtype=BF16)
        act = torch.empty(B * T, F_, device=dev, dtype=BF16)
        scratch = (torch.empty(T, T, device=dev, dtype=F32), torch.zeros(T, T, device=dev, dtype=BF16), torch.empty(64, T, device=dev, dtype=BF16))
        bias = self._position_bias(T)
        eps = float(c.layer_norm_epsilon)

        def rms(w, dst):
            native.check("rt_rmsnorm_rows", lib.rt_rmsnorm_rows(x.data_ptr(), d, 1, w.data_ptr(), dst.data_ptr(), d, B * T, d, eps, _stream()))

        for wqkv, wo, wi, wff, ln0, ln1 in plans:
            rms(ln0, xn)
            ops.linear(xn, wqkv, qkv.view(B * T, 3 * inner))
            _attention_heads(qkv[..., :inner], qkv[..., inner : 2 * inner], qkv[..., 2 * inner :], att, H, bias, 1.0, T, T, scratch)
            ops.linear(att.view(B * T, inner), wo, x, res=x)
            rms(ln1, xn)
            ops.linear(xn, wi, hid, gelu_from=F_)                                # [wi_1 x | gelu(wi_0 x)]
            native.check("rt_gated_mul", lib.rt_gated_mul(hid.data_ptr(), 2 * F_, act.data_ptr(), F_, B * T, F_, _stream()))
            ops.linear(act, wff, x, res=x)
        out = torch.empty(B * T, d, device=dev, dtype=BF16)
        rms(self.encoder.final_layer_norm.weight.data, out)
        out = out.view(B, T, d)
        return BaseModelOutput(out) if return_dict else (out,)

    __call__ = forward


# =========================================================================================================== CLIP
class CLIPTextModel(nn.Module, WeightsIO):
    config_name = "config.json"
    weights_name = "model.safetensors"

    def __init__(self, vocab_size: int = 49408, hidden_size: int = 768, intermediate_size: int = 3072, num_hidden_layers: int = 12,
                 num_attention_heads: int = 12, max_position_embeddings: int = 77, hidden_act: str = "quick_gelu",
                 layer_norm_eps: float = 1e-5, eos_token_id: int = 2, device=None, dtype=None, **unused):
        super().__init__()
        if hidden_size % num_attention_heads or hidden_size // num_attention_heads != 64:
            raise ValueError("CLIPTextModel (HIP): head dim must be 64 (CLIP-L: 768 / 12)")
        if hidden_act != "quick_gelu":
            raise ValueError("CLIPTextModel (HIP): only quick_gelu (openai/clip-vit-large-patch14) is implemented")
        self.config = Config(vocab_size=vocab_size, hidden_size=hidden_size, intermediate_size=intermediate_size,
                             num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                             max_position_embeddings=max_position_embeddings, hidden_act=hidden_act, layer_norm_eps=layer_norm_eps,
                             eos_token_id=eos_token_id)
        kw = dict(device=device, dtype=dtype)
        tm = _H()
        emb = _H()
        emb.token_embedding = _W(vocab_size, hidden_size, **kw)
        emb.position_embedding = _W(max_position_embeddings, hidden_size, **kw)
        tm.embeddings = emb
        enc = _H()
        enc.layers = nn.ModuleList()
        for _ in range(num_hidden_layers):
            l = _H()
            sa = _H()
            sa.q_proj, sa.k_proj, sa.v_proj, sa.out_proj = (_WB(hidden_size, hidden_size, **kw) for _ in range(4))
            l.self_attn = sa
            l.layer_norm1, l.layer_norm2 = _WB(hidden_size, **kw), _WB(hidden_size, **kw)
            mlp = _H()
            mlp.fc1, mlp.fc2 = _WB(intermediate_size, hidden_size, **kw), _WB(hidden_size, intermediate_size, **kw)
            l.mlp = mlp
            enc.layers.append(l)
        tm.encoder = enc
        tm.final_layer_norm = _WB(hidden_size, **kw)
        self.text_model = tm
        self._plans = None

    @property
    def dtype(self):
        return self.text_model.final_layer_norm.weight.dtype

    @property
    def device(self):
        return self.text_model.final_layer_norm.weight.device

    def _apply(self, fn, *a, **k):
        self._plans = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, sd, strict: bool = True, **kw):
        # transformers >= 5 saves the text model's keys without the `text_model.` prefix; accept both layouts
        if not any(k.startswith("text_model.") for k in sd):
            sd = {"text_model." + k: v for k, v in sd.items()}
        sd = {k: v for k, v in sd.items() if not k.endswith("position_ids")}
        self._plans = None
        return super().load_state_dict(sd, strict=strict, **kw)

    @classmethod
    def from_pretrained(cls, path: str, torch_dtype=None, subfolder: Optional[str] = None, device=None, **unused):
        d = cls._resolve_dir(path, subfolder)
        with open(os.path.join(d, cls.config_name)) as f:
            cfg = json.load(f)
        cfg = cfg.get("text_config", cfg)
        m = cls(**cfg, device=device or "cpu", dtype=torch_dtype or BF16)
        m.load_state_dict({k: v.to(torch_dtype or BF16) for k, v in cls._load_safetensors_dir(d).items()}, strict=True)
        return m

    def _ensure_plans(self):
        if self._plans is not None:
            return self._plans
        if self.dtype != BF16 or not self.text_model.final_layer_norm.weight.is_cuda:
            raise RuntimeError("CLIPTextModel (HIP): bf16 on the GPU only; there is no CPU fallback")

        def affine(ln):           # LayerNorm(x)·w + b == LN(x)·(1 + (w - 1)) + b: the adaLN kernel with constant vectors
            return (ln.weight.data.to(F32) - 1.0).reshape(1, -1).contiguous(), ln.bias.data.to(F32).reshape(1, -1).contiguous()

        plans = []
        for l in self.text_model.encoder.layers:
            sa = l.self_attn
            wqkv = torch.cat([sa.q_proj.weight.data, sa.k_proj.weight.data, sa.v_proj.weight.data], dim=0).contiguous()
            bqkv = torch.cat([sa.q_proj.bias.data, sa.k_proj.bias.data, sa.v_proj.bias.data], dim=0).contiguous()
            plans.append((wqkv, bqkv, sa.out_proj.weight.data, sa.out_proj.bias.data, l.mlp.fc1.weight.data, l.mlp.fc1.bias.data,
                          l.mlp.fc2.weight.data, l.mlp.fc2.bias.data, affine(l.layer_norm1), affine(l.layer_norm2)))
        self._plans = (plans, affine(self.text_model.final_layer_norm))
        return self._plans

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask=None, output_hidden_states: bool = False, return_dict: bool = True, **unused):
        if attention_mask is not None:
            raise NotImplementedError("CLIPTextModel (HIP): the FLUX pipelines pass no attention mask (PIPE:337-339)")
        plans, final_ln = self._ensure_plans()
        c = self.config
        dev = self.device
        lib = native.load()
        B, T = input_ids.shape
        if T > c.max_position_embeddings:
            raise ValueError("sequence longer than max_position_embeddings")
        Tp = (T + 63) // 64 * 64                                                 # K of the P·V GEMM must be a multiple of 64
        d, H, F_ = c.hidden_size, c.num_attention_heads, c.intermediate_size
        ids = torch.zeros(B, Tp, dtype=torch.int32, device=dev)
        ids[:, :T] = input_ids.to(dev, torch.int32)
        emb = torch.empty(B * Tp, d, device=dev, dtype=BF16)
        native.check("rt_embedding_gather", lib.rt_embedding_gather(self.text_model.embeddings.token_embedding.weight.data_ptr(), d, ids.data_ptr(),
                                                                    emb.data_ptr(), d, B * Tp, d, c.vocab_size, _stream()))
        pos = torch.zeros(Tp, d, device=dev, dtype=BF16)
        pos[:T] = self.text_model.embeddings.position_embedding.weight.data[:T]
        x = ops.to_f32(emb).view(B, Tp, d)
        for b in range(B):                                                        # x[b] += position embeddings
            ops.masked_accumulate_(x[b : b + 1], pos.unsqueeze(0), None, 1.0, True)
        xn = torch.empty(B, Tp, d, device=dev, dtype=BF16)
        qkv = torch.empty(B, Tp, 3 * d, device=dev, dtype=BF16)
        att = torch.empty(B, Tp, d, device=dev, dtype=BF16)
        hid = torch.empty(B * Tp, F_, device=dev, dtype=BF16)
        scratch = (torch.empty(Tp, Tp, device=dev, dtype=F32), torch.zeros(Tp, Tp, device=dev, dtype=BF16), torch.empty(64, Tp, device=dev, dtype=BF16))   # padded query rows keep zero probabilities
        causal = torch.full((1, T, T), float("-inf"), device=dev, dtype=F32).triu_(1).contiguous()     # -inf above the diagonal
        x2 = x.view(B * Tp, d)
        for wqkv, bqkv, wo, bo, w1, b1, w2, b2, ln1, ln2 in plans:
            ops.layernorm_modulate(x2.unsqueeze(0), xn.view(1, B * Tp, d), ln1[1], ln1[0], eps=float(c.layer_norm_eps))
            ops.linear(xn.view(B * Tp, d), wqkv, qkv.view(B * Tp, 3 * d), bias=bqkv)
            _attention_heads(qkv[..., :d], qkv[..., d : 2 * d], qkv[..., 2 * d :], att, H, causal, 64 ** -0.5, T, Tp, scratch)
            ops.linear(att.view(B * Tp, d), wo, x2, bias=bo, res=x2)
            ops.layernorm_modulate(x2.unsqueeze(0), xn.view(1, B * Tp, d), ln2[1], ln2[0], eps=float(c.layer_norm_eps))
            ops.linear(xn.view(B * Tp, d), w1, hid, bias=b1)
            native.check("rt_quick_gelu", lib.rt_quick_gelu(hid.data_ptr(), hid.numel(), _stream()))
            ops.linear(hid, w2, x2, bias=b2, res=x2)
        out = torch.empty(B, Tp, d, device=dev, dtype=BF16)
        ops.layernorm_modulate(x2.unsqueeze(0), out.view(1, B * Tp, d), final_ln[1], final_ln[0], eps=float(c.layer_norm_eps))
        last = out[:, :T]
        idc = input_ids.to(dev)
        if c.eos_token_id == 2:        # legacy configs: the EOS token has the highest id in the vocabulary
            pos_eos = idc.to(torch.int64).argmax(dim=-1)
        else:
            pos_eos = (idc == c.eos_token_id).to(torch.int64).argmax(dim=-1)
        pooled = last[torch.arange(B, device=dev), pos_eos]
        return BaseModelOutput(last, pooled) if return_dict else (last, pooled)

    __call__ = forward
