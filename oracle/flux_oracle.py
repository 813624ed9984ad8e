"""fp32 CPU oracle of the MMDiT math (TEST INFRASTRUCTURE — see oracle/__init__.py; parity unpinned).

Functional restatement, stock torch CPU ops, weights in a flat dict keyed like a diffusers state dict
(SURVEY.md Appendix B). Each function cites what it follows:
  CN   = /root/reference/RepText/controlnet_flux.py
  PIPE = /root/reference/RepText/pipeline_flux_controlnet.py
  A.x  = SURVEY.md Appendix A (diffusers 0.36.0 math, not present under /root/reference)
"""
from __future__ import annotations

import contextlib
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

# Storage-precision emulation (off by default = the fp32 oracle). Under `stored_as(torch.bfloat16)` every value the HIP
# path keeps in HBM as bf16 between fused stages (GEMM A operands, q/k/v, attention probabilities and output, GELU
# hidden, ControlNet samples, velocity) is rounded to bf16 at that point; everything else stays fp32, as on the GPU
# (residual stream, statistics, adaLN vectors, master latents). It separates dtype noise from logic error: GPU vs this
# mode must agree to accumulation-order noise, while GPU vs the plain fp32 oracle carries the bf16 rounding floor.
_STORE: Optional[torch.dtype] = None


_REF16_SCALARS = False


@contextlib.contextmanager
def reference_bf16_scalars(on: bool = True):
    """The scalar roundings of the reference's bf16 run (PIPE:1025 t.to(dtype); PIPE:1048/1094 t/1000 in bf16; CN:282-284
    timestep.to(dtype)*1000 and guidance.to(dtype)*1000 in bf16): t = 967.3 -> 968, guidance 3.5 -> 3504. Off = exact fp32."""
    global _REF16_SCALARS
    prev, _REF16_SCALARS = _REF16_SCALARS, bool(on)
    try:
        yield
    finally:
        _REF16_SCALARS = prev


def _x1000(v: torch.Tensor) -> torch.Tensor:
    return (v.to(torch.bfloat16) * 1000).float() if _REF16_SCALARS else v.float() * 1000


def _model_t(t: torch.Tensor) -> torch.Tensor:
    """PIPE:1025,1048: what the models get as `timestep` from scheduler value t (= sigma * 1000)."""
    return (t.to(torch.bfloat16) / 1000).float() if _REF16_SCALARS else t / 1000.0


def _s(x: torch.Tensor) -> torch.Tensor:
    return x if _STORE is None else x.to(_STORE).to(torch.float32)


@contextlib.contextmanager
def stored_as(dtype: Optional[torch.dtype]):
    global _STORE
    prev, _STORE = _STORE, dtype
    try:
        yield
    finally:
        _STORE = prev


# fp8 emulation (BASELINE config 5, off by default): under `fp8_linears()` the projections the HIP path runs on the e4m3 MFMA
# (those fed by a LayerNorm) see their input quantised per token and their weight per output channel, exactly as
# rt_layernorm_modulate_fp8 / rt_quantize_rows_fp8 do: scale = amax / 448, q = e4m3(x / scale), product de-quantised in fp32.
_FP8 = False
_FP8_SUFFIXES = (".attn.to_q", ".attn.to_k", ".attn.to_v", ".attn.add_q_proj", ".attn.add_k_proj", ".attn.add_v_proj",
                 ".ff.net.0.proj", ".ff_context.net.0.proj", ".proj_mlp")


_FP8_SUFFIXES_ALL = _FP8_SUFFIXES + (".attn.to_out.0", ".attn.to_add_out", ".ff.net.2", ".ff_context.net.2", ".proj_out")


@contextlib.contextmanager
def fp8_linears(on=True):
    """on: True / "ln" = the LayerNorm-fed projections, "all" = every projection inside the blocks (per-row scales), "mx" = every
    projection, the ones NOT fed by a LayerNorm (to_out, to_add_out, ff.net.2, ff_context.net.2, proj_out) with one E8M0 scale per
    32 consecutive input elements (`quant_mx_e4m3`) taken from the fp32 value their producer computed (GELU hidden, e4m3 attention
    output: no bf16 store in between — csrc/gemm_bf16.hip's c8 epilogue, rt_attention_fp8_fwd_mx)."""
    global _FP8
    prev, _FP8 = _FP8, ("ln" if on is True else on)
    try:
        yield
    finally:
        _FP8 = prev


def quant_rows_e4m3(x: torch.Tensor) -> torch.Tensor:
    """Quantise-dequantise along the last dim with one scale per row (restates csrc/norm_elem.hip's arithmetic)."""
    amax = x.abs().amax(dim=-1, keepdim=True)
    sc = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    q = (x * (1.0 / sc)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
    return q * sc


def mx_scale_byte(amax: torch.Tensor) -> torch.Tensor:
    """E8M0 byte of a block with maximum magnitude amax (csrc/rt_common.h: rt_mx_scale_byte): the smallest s with
    amax <= 448 * 2^(s-127), from the float's own exponent and mantissa; clamped to [1, 253]; 1 for an all-zero block."""
    m, e = torch.frexp(amax.to(torch.float32))                    # amax = m * 2^e, m in [0.5, 1)  ->  (2m) * 2^(e-1)
    s = e.to(torch.int32) - 1 + 127 - 8 + (2.0 * m > 1.75).to(torch.int32)
    s = torch.where(amax > 0, s, torch.ones_like(s))
    return s.clamp(1, 253)


def quant_mx_e4m3(x: torch.Tensor, return_parts: bool = False):
    """Quantise-dequantise along the last dim in blocks of 32 with a power-of-two scale per block (rt_quantize_mx_fp8)."""
    shp = x.shape
    xb = x.to(torch.float32).reshape(*shp[:-1], shp[-1] // 32, 32)
    sb = mx_scale_byte(xb.abs().amax(dim=-1))
    q = torch.ldexp(xb, (127 - sb).unsqueeze(-1)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    if return_parts:
        return q.reshape(shp), sb.to(torch.uint8)
    return torch.ldexp(q.to(torch.float32), (sb - 127).unsqueeze(-1)).reshape(shp)


def _act_out(x: torch.Tensor) -> torch.Tensor:
    """Output of a producer the "mx" level fuses its quantisation into: not stored as bf16 there."""
    return x if _FP8 == "mx" else _s(x)


def _ln_out(x: torch.Tensor) -> torch.Tensor:
    """LayerNorm-modulate output: stored as bf16 on the bf16 path, quantised straight from fp32 on the fp8 path."""
    return x if _FP8 else _s(x)


# --------------------------------------------------------------------------------------- primitives
def linear(p: Params, name: str, x: torch.Tensor) -> torch.Tensor:
    w = p[name + ".weight"]
    if _FP8 and "transformer_blocks." in name and name.endswith(_FP8_SUFFIXES_ALL if _FP8 in ("all", "mx") else _FP8_SUFFIXES):
        if _FP8 == "mx" and not name.endswith(_FP8_SUFFIXES):
            return F.linear(quant_mx_e4m3(x), quant_rows_e4m3(w), p.get(name + ".bias"))
        return F.linear(quant_rows_e4m3(x), quant_rows_e4m3(w), p.get(name + ".bias"))
    return F.linear(x, w, p.get(name + ".bias"))


def layer_norm(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """LayerNorm over the last dim, no affine (A.1 notation LN)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps)


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """A.1 step 3: x * rsqrt(mean(x^2) + eps) * w over the head dim."""
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def silu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(x)


def rope_table(ids: torch.Tensor, axes_dim: Sequence[int] = (16, 56, 56), theta: float = 10000.0):
    """FluxPosEmbed (CN:65,316-317; A.5): fp64 angles, each frequency repeated twice, -> fp32 [S, sum(axes)]."""
    cos_out, sin_out = [], []
    pos = ids.to(torch.float64)
    for a, d in enumerate(axes_dim):
        omega = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64) / d))
        ang = pos[:, a : a + 1] * omega[None, :]
        cos_out.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin_out.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_out, dim=1), torch.cat(sin_out, dim=1)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """A.1 step 5. x [B,S,H,Dh]; pairs (x[2j], x[2j+1]) rotated."""
    xr = x.reshape(*x.shape[:-1], -1, 2)
    x_real, x_imag = xr[..., 0], xr[..., 1]
    rot = torch.stack([-x_imag, x_real], dim=-1).flatten(-2)
    return x * cos[None, :, None, :] + rot * sin[None, :, None, :]


def timestep_embedding(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0) (A.5): [cos | sin]."""
    half = dim // 2
    f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    a = t.float()[:, None] * f[None, :]
    return torch.cat([a.cos(), a.sin()], dim=-1)


def time_text_embed(p: Params, prefix: str, t1000: torch.Tensor, g1000: Optional[torch.Tensor], pooled: torch.Tensor):
    """CombinedTimestep(Guidance)TextProjEmbeddings (CN:66-71,287-291; A.5)."""
    def mlp(name, x):
        return linear(p, f"{prefix}.{name}.linear_2", silu(linear(p, f"{prefix}.{name}.linear_1", x)))

    temb = mlp("timestep_embedder", timestep_embedding(t1000))
    if g1000 is not None:
        temb = temb + mlp("guidance_embedder", timestep_embedding(g1000))
    return temb + mlp("text_embedder", pooled)


_FP8_ATTN = False


@contextlib.contextmanager
def fp8_attention(on: bool = True):
    """Emulate csrc/attention_fp8.hip's static quantisation: q, k -> e4m3(16·x) (the 1/256 goes into the scale), v -> e4m3(v),
    numerators -> e4m3(exp2(s - max + 2)); row sums from the unrounded numerators; fp32 everywhere else."""
    global _FP8_ATTN
    prev, _FP8_ATTN = _FP8_ATTN, on
    try:
        yield
    finally:
        _FP8_ATTN = prev


def _e4m3(x: torch.Tensor) -> torch.Tensor:
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """softmax(q k^T / sqrt(Dh)) v, inputs [B,S,H,Dh] -> [B,S,H*Dh] (A.1 step 6)."""
    B, S, H, Dh = q.shape
    qh, kh, vh = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    if _FP8_ATTN:
        s = (_e4m3(qh * 16.0) @ _e4m3(kh * 16.0).transpose(-1, -2)) * (1.0 / (256.0 * math.sqrt(Dh)))
        e = torch.exp2((s - s.amax(dim=-1, keepdim=True)) * math.log2(math.e) + 2.0)
        o = (_e4m3(e) @ _e4m3(vh)) / e.sum(dim=-1, keepdim=True)
        return _act_out(o.permute(0, 2, 1, 3).reshape(B, S, H * Dh))
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(Dh)
    if _STORE is None:
        o = torch.softmax(s, dim=-1) @ vh
    else:   # row sum from the unrounded numerators, numerators rounded as the second product's operand
        e = torch.exp(s - s.amax(dim=-1, keepdim=True))
        o = (_s(e) @ vh) / e.sum(dim=-1, keepdim=True)
    return _s(o.permute(0, 2, 1, 3).reshape(B, S, H * Dh))


# --------------------------------------------------------------------------------------- blocks
def double_block(p: Params, pre: str, h, e, temb, rope, H: int = 24, Dh: int = 128):
    """FluxTransformerBlock (A.1), constructed at CN:78-83, called at CN:343-348. Returns (e, h)."""
    B, N, d = h.shape
    T = e.shape[1]
    cos, sin = rope
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = linear(p, f"{pre}.norm1.linear", silu(temb)).chunk(6, dim=-1)
    csh_a, csc_a, cg_a, csh_m, csc_m, cg_m = linear(p, f"{pre}.norm1_context.linear", silu(temb)).chunk(6, dim=-1)
    nh = _ln_out(layer_norm(h) * (1 + sc_a[:, None]) + sh_a[:, None])
    ne = _ln_out(layer_norm(e) * (1 + csc_a[:, None]) + csh_a[:, None])

    def heads(x):
        return _s(x).reshape(x.shape[0], x.shape[1], H, Dh)

    q = rms_norm(heads(linear(p, f"{pre}.attn.to_q", nh)), p[f"{pre}.attn.norm_q.weight"])
    k = rms_norm(heads(linear(p, f"{pre}.attn.to_k", nh)), p[f"{pre}.attn.norm_k.weight"])
    v = heads(linear(p, f"{pre}.attn.to_v", nh))
    eq = rms_norm(heads(linear(p, f"{pre}.attn.add_q_proj", ne)), p[f"{pre}.attn.norm_added_q.weight"])
    ek = rms_norm(heads(linear(p, f"{pre}.attn.add_k_proj", ne)), p[f"{pre}.attn.norm_added_k.weight"])
    ev = heads(linear(p, f"{pre}.attn.add_v_proj", ne))
    Q = _s(apply_rope(torch.cat([eq, q], dim=1), cos, sin))   # text first
    K = _s(apply_rope(torch.cat([ek, k], dim=1), cos, sin))
    V = torch.cat([ev, v], dim=1)
    A = attention(Q, K, V)
    a_e = linear(p, f"{pre}.attn.to_add_out", A[:, :T])
    a_h = linear(p, f"{pre}.attn.to_out.0", A[:, T:])

    def ff(name, x):
        return linear(p, f"{pre}.{name}.net.2", _act_out(gelu_tanh(linear(p, f"{pre}.{name}.net.0.proj", x))))

    h = h + g_a[:, None] * a_h
    h = h + g_m[:, None] * ff("ff", _ln_out(layer_norm(h) * (1 + sc_m[:, None]) + sh_m[:, None]))
    e = e + cg_a[:, None] * a_e
    e = e + cg_m[:, None] * ff("ff_context", _ln_out(layer_norm(e) * (1 + csc_m[:, None]) + csh_m[:, None]))
    return e, h


def single_block(p: Params, pre: str, x, temb, rope, H: int = 24, Dh: int = 128):
    """FluxSingleTransformerBlock (A.2), old call contract of CN:376-380 (x pre-concatenated, returns tensor)."""
    B, S, d = x.shape
    cos, sin = rope
    sh, sc, g = linear(p, f"{pre}.norm.linear", silu(temb)).chunk(3, dim=-1)
    nx = _ln_out(layer_norm(x) * (1 + sc[:, None]) + sh[:, None])
    m = _act_out(gelu_tanh(linear(p, f"{pre}.proj_mlp", nx)))
    q = rms_norm(_s(linear(p, f"{pre}.attn.to_q", nx)).reshape(B, S, H, Dh), p[f"{pre}.attn.norm_q.weight"])
    k = rms_norm(_s(linear(p, f"{pre}.attn.to_k", nx)).reshape(B, S, H, Dh), p[f"{pre}.attn.norm_k.weight"])
    v = _s(linear(p, f"{pre}.attn.to_v", nx)).reshape(B, S, H, Dh)
    A = attention(_s(apply_rope(q, cos, sin)), _s(apply_rope(k, cos, sin)), v)
    return x + g[:, None] * linear(p, f"{pre}.proj_out", torch.cat([A, m], dim=2))


# --------------------------------------------------------------------------------------- models
def controlnet_forward(p: Params, cfg: dict, hidden_states, controlnet_cond, encoder_hidden_states, pooled_projections,
                       timestep, img_ids, txt_ids, guidance=None, conditioning_scale: float = 1.0, _store_samples: bool = True):
    """FluxControlNetModel.forward, CN:277-408 (union mode excluded: CN:294-301 is out of scope).

    `_store_samples` only matters under stored_as(): the loops pass False and round after their mask/sum, where the HIP
    path (which fuses both into the zero-linear epilogue) rounds."""
    H, Dh = cfg["num_attention_heads"], cfg["attention_head_dim"]
    st = _s if _store_samples else (lambda t: t)
    h = linear(p, "x_embedder", _s(hidden_states)) + linear(p, "controlnet_x_embedder", _s(controlnet_cond))  # CN:277-280
    t1000 = _x1000(timestep)                                                                                # CN:282
    g1000 = _x1000(guidance) if (guidance is not None and cfg.get("guidance_embeds", False)) else None
    temb = time_text_embed(p, "time_text_embed", t1000, g1000, pooled_projections)                         # CN:287-291
    e = linear(p, "context_embedder", _s(encoder_hidden_states))                                            # CN:292
    rope = rope_table(torch.cat([txt_ids, img_ids], dim=0).float(), cfg.get("axes_dims_rope", (16, 56, 56)))  # CN:316-317
    block_samples = []
    for i in range(cfg["num_layers"]):                                                                       # CN:320-349
        e, h = double_block(p, f"transformer_blocks.{i}", h, e, temb, rope, H, Dh)
        block_samples.append(h)
    x = torch.cat([e, h], dim=1)                                                                             # CN:351
    T = e.shape[1]
    single_samples = []
    for i in range(cfg["num_single_layers"]):                                                                # CN:354-381
        x = single_block(p, f"single_transformer_blocks.{i}", x, temb, rope, H, Dh)
        single_samples.append(x[:, T:])
    outs = [st(linear(p, f"controlnet_blocks.{i}", _s(s)) * conditioning_scale) for i, s in enumerate(block_samples)]   # CN:384-396
    souts = [st(linear(p, f"controlnet_single_blocks.{i}", _s(s)) * conditioning_scale) for i, s in enumerate(single_samples)]
    return (outs or None), (souts or None)                                                                   # CN:398-408


def transformer_forward(p: Params, cfg: dict, hidden_states, encoder_hidden_states, pooled_projections, timestep,
                        img_ids, txt_ids, guidance=None, controlnet_block_samples=None,
                        controlnet_single_block_samples=None):
    """FluxTransformer2DModel.forward (A.3), call site PIPE:1092-1104."""
    H, Dh = cfg["num_attention_heads"], cfg["attention_head_dim"]
    h = linear(p, "x_embedder", _s(hidden_states))
    t1000 = _x1000(timestep)
    g1000 = _x1000(guidance) if (guidance is not None and cfg.get("guidance_embeds", False)) else None
    temb = time_text_embed(p, "time_text_embed", t1000, g1000, pooled_projections)
    e = linear(p, "context_embedder", _s(encoder_hidden_states))
    rope = rope_table(torch.cat([txt_ids, img_ids], dim=0).float(), cfg.get("axes_dims_rope", (16, 56, 56)))
    nl, ns = cfg["num_layers"], cfg["num_single_layers"]
    for i in range(nl):
        e, h = double_block(p, f"transformer_blocks.{i}", h, e, temb, rope, H, Dh)
        if controlnet_block_samples is not None:
            k = int(math.ceil(nl / len(controlnet_block_samples)))
            h = h + controlnet_block_samples[i // k]
    T = e.shape[1]
    x = torch.cat([e, h], dim=1)
    for i in range(ns):
        x = single_block(p, f"single_transformer_blocks.{i}", x, temb, rope, H, Dh)
        if controlnet_single_block_samples is not None:
            k = int(math.ceil(ns / len(controlnet_single_block_samples)))
            x = torch.cat([x[:, :T], x[:, T:] + controlnet_single_block_samples[i // k]], dim=1)
    h = x[:, T:]
    scale, shift = linear(p, "norm_out.linear", silu(temb)).chunk(2, dim=-1)      # SCALE first (A.3)
    h = _s(layer_norm(h) * (1 + scale[:, None]) + shift[:, None])
    return _s(linear(p, "proj_out", h))


# --------------------------------------------------------------------------------------- scheduler / latents
def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.16):
    """PIPE:78-88 (function default max_shift=1.16; the pipeline passes the scheduler config's 1.15, PIPE:952-958)."""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


def flow_sigmas(num_steps: int, mu: float) -> torch.Tensor:
    """FlowMatchEulerDiscreteScheduler.set_timesteps(sigmas=linspace(1,1/n,n), mu) (PIPE:948-967; A.6), + trailing 0."""
    s = torch.linspace(1.0, 1.0 / num_steps, num_steps, dtype=torch.float64).to(torch.float32)
    sig = math.exp(mu) / (math.exp(mu) + (1.0 / s - 1.0))
    return torch.cat([sig.to(torch.float32), torch.zeros(1)])


def euler_step(x: torch.Tensor, v: torch.Tensor, sigma: float, sigma_next: float) -> torch.Tensor:
    """scheduler.step (PIPE:1109; A.6)."""
    return (x.float() + (sigma_next - sigma) * v.float()).to(v.dtype)


def pack_latents(x: torch.Tensor) -> torch.Tensor:
    """PIPE:550-555."""
    B, C, Hh, Ww = x.shape
    x = x.view(B, C, Hh // 2, 2, Ww // 2, 2).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, (Hh // 2) * (Ww // 2), C * 4)


def unpack_latents(x: torch.Tensor, height: int, width: int, vae_scale_factor: int = 16) -> torch.Tensor:
    """PIPE:559-570."""
    B, _, ch = x.shape
    h, w = height // vae_scale_factor, width // vae_scale_factor
    x = x.view(B, h, w, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(B, ch // 4, h * 2, w * 2)


def latent_image_ids(h2: int, w2: int) -> torch.Tensor:
    """PIPE:535-546 with height=h2, width=w2 (latent sizes): rows (0, r, c)."""
    ids = torch.zeros(h2 // 2, w2 // 2, 3)
    ids[..., 1] += torch.arange(h2 // 2)[:, None]
    ids[..., 2] += torch.arange(w2 // 2)[None, :]
    return ids.reshape(-1, 3)


def interval_map(n_blocks: int, n_samples: int) -> List[int]:
    k = int(math.ceil(n_blocks / n_samples))
    return [i // k for i in range(n_blocks)]


# --------------------------------------------------------------------------------------- denoise loop
def denoise_loop(tp: Params, tcfg: dict, cp: Optional[Params], ccfg: Optional[dict], latents, prompt_embeds, pooled,
                 control_images: Sequence[torch.Tensor], control_masks: Sequence[Optional[torch.Tensor]], sigmas,
                 img_ids, txt_ids, guidance_scale: float, conditioning_scale: float = 1.0, conditioning_step: int = 10 ** 9,
                 on_step=None):
    """The hot loop of FluxControlNetPipeline.__call__, PIPE:1016-1130, text-to-image variant.

    latents [B,N,64]; control_images: per text line [B,N,128] packed hint latents; control_masks: per line
    [1,N,1] or None; sigmas = flow_sigmas(...) (timesteps = sigma*1000, passed to the models as t/1000, Q4)."""
    B = latents.shape[0]
    n = len(sigmas) - 1
    for i in range(n):
        t = sigmas[i] * 1000.0                                               # scheduler.timesteps[i]
        timestep = _model_t(t).expand(B)                                     # PIPE:1025,1048
        guidance = torch.full((B,), float(guidance_scale)) if tcfg.get("guidance_embeds", False) else None   # PIPE:1028-1032
        merged = None
        for line, cond in enumerate(control_images):                         # PIPE:1037-1087
            if i < conditioning_step and cp is not None:
                samples, _ = controlnet_forward(cp, ccfg, latents, cond, prompt_embeds, pooled, timestep, img_ids, txt_ids,
                                                guidance=guidance, conditioning_scale=conditioning_scale, _store_samples=False)
            else:
                samples = None
            if samples is not None:
                mask = control_masks[line] if len(control_masks) > 0 else None
                if mask is not None:
                    samples = [mask * s for s in samples]
            if line == 0:
                merged = None if samples is None else [_s(a) for a in samples]
            elif samples is not None and merged is not None:
                merged = [_s(a + b) for a, b in zip(merged, samples)]
        v = transformer_forward(tp, tcfg, latents, prompt_embeds, pooled, timestep, img_ids, txt_ids, guidance=guidance,
                                controlnet_block_samples=merged)
        latents = euler_step(latents, v, float(sigmas[i]), float(sigmas[i + 1]))   # PIPE:1109
        if on_step is not None:
            on_step(i, latents)
    return latents


def denoise_loop_inpaint(tp: Params, tcfg: dict, cp: Params, ccfg: dict, ip: Params, icfg: dict, latents, prompt_embeds, pooled,
                         neg_prompt_embeds, neg_pooled, control_images, control_masks, inpaint_cond, sigmas, img_ids, txt_ids,
                         guidance_scale: float, true_guidance_scale: float, conditioning_scale: float = 1.0,
                         conditioning_scale_inpaint: float = 1.0, conditioning_step: int = 10 ** 9):
    """Hot loop of the inpaint pipeline, pipeline_flux_controlnet_inpaint.py:1138-1285 (B = 1).

    CFG is on when guidance_scale > 1 (INP:241-242): conditioning = cat([negative, positive]) (INP:1033-1035), latents stay
    batch 1 and broadcast inside the models (Q6); the inpaint tower runs unmasked every step and its residuals are added only
    when the text towers produced some (INP:1231-1245); step 0 uses zero velocity, later steps uncond + s·(text − uncond) with
    s = true_guidance_scale (INP:1264-1270)."""
    cfg_on = guidance_scale > 1
    if cfg_on:
        pe = torch.cat([neg_prompt_embeds, prompt_embeds], dim=0)
        pl = torch.cat([neg_pooled, pooled], dim=0)
        rep = lambda t: torch.cat([t, t], dim=0)
    else:
        pe, pl = prompt_embeds, pooled
        rep = lambda t: t
    hints = [rep(c) for c in control_images]
    icond = rep(inpaint_cond)
    n = len(sigmas) - 1
    for i in range(n):
        B = latents.shape[0]
        timestep = _model_t(sigmas[i] * 1000.0).expand(B)
        guidance = torch.full((B,), float(guidance_scale)) if tcfg.get("guidance_embeds", False) else None
        lat_in = latents.expand(pe.shape[0], -1, -1) if cfg_on else latents        # what broadcasting inside the models amounts to
        ts_in = timestep.expand(pe.shape[0]) if cfg_on else timestep
        g_in = guidance.expand(pe.shape[0]) if (cfg_on and guidance is not None) else guidance
        merged = None
        for line, cond in enumerate(hints):
            samples = None
            if i < conditioning_step:
                samples, _ = controlnet_forward(cp, ccfg, lat_in, cond, pe, pl, ts_in, img_ids, txt_ids, guidance=g_in,
                                                conditioning_scale=conditioning_scale, _store_samples=False)
                mask = control_masks[line] if len(control_masks) > 0 else None
                if mask is not None:
                    samples = [mask * s for s in samples]
            if line == 0:
                merged = None if samples is None else [_s(a) for a in samples]
            elif samples is not None and merged is not None:
                merged = [_s(a + b) for a, b in zip(merged, samples)]
        isamples, _ = controlnet_forward(ip, icfg, lat_in, icond, pe, pl, ts_in, img_ids, txt_ids, guidance=g_in,
                                         conditioning_scale=conditioning_scale_inpaint, _store_samples=False)
        if isamples is not None and merged is not None:
            merged = [_s(a + b) for a, b in zip(merged, isamples)]
        v = transformer_forward(tp, tcfg, lat_in, pe, pl, ts_in, img_ids, txt_ids, guidance=g_in, controlnet_block_samples=merged)
        if cfg_on:
            v_u, v_t = v.chunk(2)
            v = _s(v_u + true_guidance_scale * (v_t - v_u)) if i > 0 else v_t * 0.0
        latents = euler_step(latents, v, float(sigmas[i]), float(sigmas[i + 1]))
    return latents


# --------------------------------------------------------------------------------------- synthetic weights
def _lin(p: Params, name: str, out_f: int, in_f: int, gen: torch.Generator, std: float = 0.02, bias_std: float = 0.0):
    p[name + ".weight"] = torch.randn(out_f, in_f, generator=gen) * std
    p[name + ".bias"] = torch.randn(out_f, generator=gen) * bias_std if bias_std > 0 else torch.zeros(out_f)


def init_mmdit_params(cfg: dict, seed: int, controlnet: bool = False, bias_std: float = 0.02,
                      round_bf16: bool = True) -> Params:
    """Random FLUX-shaped weights: W ~ N(0, 0.02²); key names per Appendix B.

    Biases and RMSNorm weights are given small random values as well (the survey's plan zeroes them, which would
    leave bias and norm-weight indexing untested). ControlNet zero-linears are NOT zero (§8d). With round_bf16 the
    values are rounded to bf16 and stored as fp32, so the GPU path (bf16 storage) sees identical weights."""
    g = torch.Generator().manual_seed(seed)
    d = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    Dh = cfg["attention_head_dim"]
    p: Params = {}
    _lin(p, "x_embedder", d, cfg["in_channels"], g, bias_std=bias_std)
    _lin(p, "context_embedder", d, cfg["joint_attention_dim"], g, bias_std=bias_std)
    emb = ["timestep_embedder"] + (["guidance_embedder"] if cfg.get("guidance_embeds", False) else [])
    for name in emb:
        _lin(p, f"time_text_embed.{name}.linear_1", d, 256, g, bias_std=bias_std)
        _lin(p, f"time_text_embed.{name}.linear_2", d, d, g, bias_std=bias_std)
    _lin(p, "time_text_embed.text_embedder.linear_1", d, cfg["pooled_projection_dim"], g, bias_std=bias_std)
    _lin(p, "time_text_embed.text_embedder.linear_2", d, d, g, bias_std=bias_std)
    for i in range(cfg["num_layers"]):
        pre = f"transformer_blocks.{i}"
        _lin(p, f"{pre}.norm1.linear", 6 * d, d, g, bias_std=bias_std)
        _lin(p, f"{pre}.norm1_context.linear", 6 * d, d, g, bias_std=bias_std)
        for nm in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            _lin(p, f"{pre}.attn.{nm}", d, d, g, bias_std=bias_std)
        for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            p[f"{pre}.attn.{nm}.weight"] = 1.0 + 0.1 * torch.randn(Dh, generator=g)
        for ffn in ("ff", "ff_context"):
            _lin(p, f"{pre}.{ffn}.net.0.proj", 4 * d, d, g, bias_std=bias_std)
            _lin(p, f"{pre}.{ffn}.net.2", d, 4 * d, g, bias_std=bias_std)
    for i in range(cfg["num_single_layers"]):
        pre = f"single_transformer_blocks.{i}"
        _lin(p, f"{pre}.norm.linear", 3 * d, d, g, bias_std=bias_std)
        _lin(p, f"{pre}.proj_mlp", 4 * d, d, g, bias_std=bias_std)
        _lin(p, f"{pre}.proj_out", d, 5 * d, g, bias_std=bias_std)
        for nm in ("to_q", "to_k", "to_v"):
            _lin(p, f"{pre}.attn.{nm}", d, d, g, bias_std=bias_std)
        for nm in ("norm_q", "norm_k"):
            p[f"{pre}.attn.{nm}.weight"] = 1.0 + 0.1 * torch.randn(Dh, generator=g)
    if controlnet:
        for i in range(cfg["num_layers"]):
            _lin(p, f"controlnet_blocks.{i}", d, d, g, bias_std=bias_std)
        for i in range(cfg["num_single_layers"]):
            _lin(p, f"controlnet_single_blocks.{i}", d, d, g, bias_std=bias_std)
        _lin(p, "controlnet_x_embedder", d, cfg["in_channels"] + cfg.get("extra_condition_channels", 0), g, bias_std=bias_std)
    else:
        _lin(p, "norm_out.linear", 2 * d, d, g, bias_std=bias_std)
        _lin(p, "proj_out", cfg.get("out_channels") or cfg["in_channels"], d, g, bias_std=bias_std)
    if round_bf16:
        for k in p:
            p[k] = p[k].to(torch.bfloat16).to(torch.float32)
    return p


FLUX_DEV_CFG = dict(patch_size=1, in_channels=64, num_layers=19, num_single_layers=38, attention_head_dim=128,
                    num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True,
                    axes_dims_rope=(16, 56, 56))
REPTEXT_CN_CFG = dict(patch_size=1, in_channels=64, num_layers=6, num_single_layers=0, attention_head_dim=128,
                      num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768, guidance_embeds=True,
                      axes_dims_rope=(16, 56, 56), extra_condition_channels=64)


def param_count(p: Params) -> int:
    return sum(int(v.numel()) for v in p.values())
