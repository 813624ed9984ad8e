/*
 * reptext_hip.h — C ABI of librt_reptext_hip.so, the MI355X (gfx950) kernels behind the
 * FLUX.1-dev + RepText-ControlNet denoising path and the AutoencoderKL decoder.
 *
 * Boundary rules:
 *   - extern "C", plain pointers and sizes only; no torch types cross this line.
 *   - every pointer is a DEVICE pointer (HBM) unless its comment says "host".
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and
 *     nothing here synchronises, allocates or frees (graph-capture safe).
 *   - return value: 0 = enqueued; negative = argument rejected (RT_E_*), positive = hipError_t.
 *   - bf16 tensors are raw uint16 bit patterns (torch.bfloat16 storage).
 *
 * Each entry replaces torch ops that the reference reaches through diffusers; the
 * reference call site it serves is cited as file:line relative to /root/reference/RepText
 * (CN = controlnet_flux.py, PIPE = pipeline_flux_controlnet.py, INP = ..._inpaint.py) and the
 * third-party math as SURVEY.md Appendix A.x.
 */
#ifndef REPTEXT_HIP_H
#define REPTEXT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_E_BADARG (-1)   /* null pointer / non-positive size */
#define RT_E_ALIGN (-2)    /* pointer or leading dimension not aligned as documented */
#define RT_E_SHAPE (-3)    /* shape outside what the kernel supports */

/* Library identity: returns a static string "reptext_hip <abi> gfx950". */
const char* rt_version(void);
/* ABI revision; bumped when any struct below changes. */
int rt_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Linear layers: C = epilogue(A · Wᵀ + bias), bf16 operands, fp32 accumulate on MFMA.
 * Replaces every nn.Linear on the path: CN:277,280,292,386,391; Appendix A.1 steps 3,6,7,8;
 * A.2 (proj_mlp, to_q/k/v fused, proj_out); A.3 (x_embedder, context_embedder, proj_out).
 *
 * Epilogue, applied per element (row m, column n) in this order:
 *   v = acc + bias[n]
 *   if n >= gelu_from:        v = gelu_tanh(v)                      (A.1 step 7 / A.2 act_mlp)
 *   if gate:                  v *= gate[(m / rows_per_batch) * gate_ld + n]   (adaLN-Zero gates)
 *   v *= alpha                                                      (CN:395 conditioning_scale)
 *   if rowscale:              v *= rowscale[batch_index * stride_rowscale + m % rows_per_batch]   (PIPE:1062 regional mask)
 *   if res:                   v += res[m*ldr + n]                   (residual stream)
 *   if add2:                  v += add2[m*ld2 + n]                  (A.3 ControlNet injection)
 *   C[m*ldc + n] = v   (bf16, or fp32 when out_f32)
 * `res`/`add2` may alias C. A group is one problem; up to RT_GEMM_MAX_GROUPS problems with the
 * same K-loop code are tiled into ONE launch (image stream + text stream of a double block).
 * Batched over `batch` with element strides (0 = shared).
 * ---------------------------------------------------------------------------------------- */
#define RT_GEMM_MAX_GROUPS 4

typedef struct rt_gemm_group {
  const void* A;        /* bf16 [batch][M][lda]          */
  const void* W;        /* bf16 [N][ldw]  (torch Linear weight layout) */
  void* C;              /* bf16|f32 [batch][M][ldc]       */
  const void* bias;     /* bf16 [N] or NULL               */
  const float* gate;    /* f32 [batch*..][gate_ld] or NULL */
  const void* res;      /* same dtype as C, or NULL       */
  const void* add2;     /* bf16 [batch][M][ld2] or NULL   */
  const float* rowscale;/* f32 [rows_per_batch] or NULL   */
  int64_t lda, ldw, ldc, ldr, ld2, gate_ld;
  int64_t strideA, strideC, strideR, stride2;  /* per-batch element strides */
  int32_t M, N, K;      /* K % 64 == 0, N % 4 == 0        */
  int32_t batch;
  int32_t rows_per_batch; /* rows sharing one gate vector; 0 => M */
  int32_t gelu_from;    /* first column that gets GELU-tanh; >= N => none */
  int32_t out_f32;      /* C/res dtype: 0 bf16, 1 f32     */
  float alpha;
  /* fp8 operands (rt_gemm_fp8 only; ignored by rt_gemm_bf16): A and W hold OCP e4m3 bytes, lda/ldw/strideA count
   * BYTES = elements, K % 128 == 0, and the product is de-quantised before the bias:
   *   acc[m][n] * a_scale[b*M + m] * w_scale[n]      (either pointer may be NULL = 1.0) */
  const float* a_scale; /* f32 [batch*M], one per activation row (rt_layernorm_modulate_fp8 / rt_quantize_rows_fp8) */
  const float* w_scale; /* f32 [N], one per output channel (rt_quantize_rows_fp8 on the weight rows)              */
  int64_t stride_rowscale; /* elements between the rowscale vectors of consecutive batch entries; 0 = one vector shared by the
                            * batch (the reference's one mask per text line); > 0: a mask per image of a sharded batch        */
  /* MX block scales (ABI 8; rt_gemm_fp8 only, ignored by rt_gemm_bf16). An activation row carries one E8M0 byte s (value
   * 2^(s-127)) per 32 consecutive K-elements. Scale tensors are laid out for the consuming GEMM wave: planes of 1024 K-elements
   * (8 K-tiles; a last partial plane is allocated whole); inside a plane 64-row chunks of 2 KiB in the order
   * [K-tile][row & 15][32-block of the K-tile][row >> 4 & 3]:
   *   byte(r, k) = base + (k/1024)*plane + (r/64)*2048 + ((k%1024)/128)*256 + (r%16)*16 + ((k%128)/32)*4 + (r/16)%4,  r = b*rows + m
   * (rt_common.h: rt_mx_scale_offset). base 16-byte aligned, plane % 2048 == 0, rows % 64 == 0 (rows = row count between batch
   * entries; a view that starts at row r0 of every entry passes base + (r0/64)*2048, r0 % 64 == 0).
   *  a_bscale != NULL: A's block scales (column 0 of A = k 0 of the scale tensor), applied by the MFMA itself
   *    (v_mfma_scale_f32_16x16x128_f8f6f4's scale operand) before the fp32 accumulation; a_scale may be NULL (usual) or given as well.
   *  c8 != NULL: columns n >= c8_from are NOT written to C; the epilogue value v (after every term) is quantised per 32 columns —
   *    s = the smallest exponent byte with max|v| <= 448 * 2^(s-127) (rt_quantize_mx_fp8's rule) — and stored as
   *    e4m3(v * 2^(127-s)) at c8[b*stride_c8 + m*ldc8 + (n - c8_from)] with s at byte(b*c_bscale_rows + m, c_bscale_k0 + n - c8_from)
   *    of (c_bscale, c_bscale_plane): the next GEMM's A and a_bscale, with no pass in between (c_bscale_k0 = the column of that A
   *    at which c8 starts, % 32 == 0).
   *    c8_from % 256 == 0, (N - c8_from) % 32 == 0, ldc8 % 8 == 0, c8 8-byte aligned. */
  const uint8_t* a_bscale;
  int64_t a_bscale_plane, a_bscale_rows;
  uint8_t* c8;
  uint8_t* c_bscale;
  int64_t ldc8, stride_c8, c_bscale_plane, c_bscale_rows;
  int32_t c8_from, c_bscale_k0;
  /* Convolution form (ABI 8; rt_gemm_bf16, one group, batch 1): conv_ks = 1 or 3 turns the problem into a stride-1, pad k/2 convolution
   * over a zero-haloed NHWC image. A = the image [B][conv_h2][conv_w2][conv_cin] (conv_h2 = H+2, conv_w2 = W+2) read as a matrix of
   * M = B*conv_h2*conv_w2 pixel rows, lda = conv_cin (% 64 == 0); W = [N][ks][ks][conv_cin], K = ks*ks*conv_cin; C / res / add2 = the
   * output image in the same haloed pixel order (ldc = its channel count). K-tile kt multiplies channels c0.. of tap (dy,dx) of
   * every pixel p with the rows of A shifted by (dy-1)*conv_w2 + (dx-1) pixels; rows shifted out of the image read as zero; halo
   * pixels are computed and NOT stored (the halo of C stays as it is). Replaces torch.nn.functional.conv2d of the AutoencoderKL
   * ResnetBlock2D / conv_in / shortcut convolutions (A.7); rt_conv2d_nhwc routes its stride-1, non-upsampling calls here.
   * conv_inv_w2 / conv_inv_h2 are filled in by the library. */
  int32_t conv_ks, conv_cin, conv_w2, conv_h2;
  float conv_inv_w2, conv_inv_h2;
} rt_gemm_group;

int rt_gemm_bf16(const rt_gemm_group* groups /* host */, int32_t ngroups, void* stream);

/* Tile selection of rt_gemm_bf16 (speed only: every output element is accumulated in the same K order by every tile shape, so
 * results are bit-identical in all modes). Bit 0: a single problem whose M x N is a whole number of 288x192 tiles and fills the
 * chip's CUs better that way (M = 4608, N = 3072: exactly 256 tiles) runs on them; bit 1: a launch of several rounds of
 * 256x256 tiles gives the columns of its poorly filled last round to 256x192 or 256x128 tiles. Default 0 (env RT_GEMM_TILES): on
 * MI355X neither pays (the chip is power-limited; measurements in DESIGN.md §5). mode >= 0 sets it, mode < 0 only queries;
 * returns the previous mode. For A/B measurements and tests. */
int rt_gemm_tile_mode(int32_t mode);

/* Same contraction and epilogue on v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3, fp32 accumulate, block scales fixed to
 * 1.0): BASELINE config 5 ("fp8 weights"), the projections fed by LayerNorm (to_q/k/v, add_*_proj, ff.net.0, proj_mlp).
 * Tiling, LDS image and schedule are those of rt_gemm_bf16 with a 128-element K-tile (the same 128-byte tile rows). */
int rt_gemm_fp8(const rt_gemm_group* groups /* host */, int32_t ngroups, void* stream);

/* Row-wise e4m3 quantisation: scale[r] = max|x[r][:]| / 448 (1.0 for an all-zero row), out[r][c] = e4m3(x[r][c] / scale[r]).
 * x bf16 (x_f32 = 0) or f32 [rows][ldx], out bytes [rows][ldo]; D % 8 == 0, D <= 65536. Used once per weight matrix (rows =
 * output channels) and for activation rows that do not come out of a LayerNorm. */
int rt_quantize_rows_fp8(const void* x, int64_t ldx, int32_t x_f32, void* out, int64_t ldo, float* scale,
                         int32_t rows, int32_t D, void* stream);

/* MX block quantisation of activation rows (config 5, "mx" level): per 32 consecutive elements one E8M0 scale byte
 * s = the smallest with max|x| <= 448 * 2^(s-127), clamped to [1, 253]; out = e4m3(x * 2^(127-s)), round to nearest even.
 * x bf16 (x_f32 = 0) or f32 [rows][ldx]; out bytes [rows][ldo]; scales in rt_gemm_group's layout (byte(row, k) there; bscale
 * 16-byte aligned, plane % 2048 == 0 and >= ceil(rows/64)*2048). D % 256 == 0. The fused producers (rt_gemm_fp8's c8 output,
 * rt_attention_fp8_fwd_mx) apply the same rule to their fp32 results; this pass serves tensors no fused producer writes. */
int rt_quantize_mx_fp8(const void* x, int64_t ldx, int32_t x_f32, void* out, int64_t ldo, uint8_t* bscale, int64_t plane,
                       int32_t rows, int32_t D, void* stream);

/* Small-M linear on fp32 activations, bf16 weights (adaLN modulation, time/guidance/pooled MLPs):
 *   y[b][n] (+)= post( Σ_k pre(x[b][k]) · W[n][k] + bias[n] ),  pre/post ∈ {identity, SiLU}
 * Replaces AdaLayerNormZero/ZeroSingle/Continuous `linear(silu(temb))` (A.1 step 1, A.2, A.3) and
 * CombinedTimestepGuidanceTextProjEmbeddings' MLPs (CN:287-291, A.5). HBM-bound on W. */
int rt_gemv_bf16w(const float* x, int64_t ldx, const void* W, int64_t ldw, const void* bias,
                  float* y, int64_t ldy, int32_t B, int32_t N, int32_t K,
                  int32_t silu_in, int32_t silu_out, int32_t accumulate, void* stream);

/* Timesteps(256, flip_sin_to_cos=True, shift 0): out[b] = [cos(t·f_j) | sin(t·f_j)], f_j = exp(-ln(1e4)·j/half).
 * A.5; feeds CN:287-291. t is already multiplied by 1000 by the caller (CN:282-284). */
int rt_timestep_embedding(const float* t, float* out, int32_t B, int32_t dim, void* stream);

/* FluxPosEmbed (CN:65,316-317; A.5): ids f32 [S][3] -> cos,sin f32 [S][sum(axes_dim)], fp64 angles,
 * each frequency repeated twice (interleaved). axes_dim host pointer, 3 ints. */
int rt_rope_table(const float* ids, float* cos_out, float* sin_out, int32_t S,
                  const int32_t* axes_dim /* host[3] */, float theta, void* stream);

/* LayerNorm(no affine, eps) over the last dim then (1+scale)·x + shift with per-batch vectors.
 * A.1 step 2/7/8, A.2, A.3 norm_out. x is bf16 or f32 (x_f32), out bf16.
 * rows = batch*rows_per_batch rows of length D (D % 8 == 0, D <= 8192); row r of batch b reads
 * x + b*stride_xb + r*ldx. shift/scale f32 with per-batch stride mod_ld (NULL => plain LN). */
int rt_layernorm_modulate(const void* x, int64_t ldx, int64_t stride_xb, int32_t x_f32,
                          void* out, int64_t ldo, int64_t stride_ob,
                          const float* shift, const float* scale, int64_t mod_ld,
                          int32_t batch, int32_t rows_per_batch, int32_t D, float eps, void* stream);
/* Same, quantising the modulated row to e4m3 on the way out: out bytes [rows][ldo] and row_scale f32 [batch*rows_per_batch]
 * (= max|y| / 448 of that row), the A operand and a_scale of rt_gemm_fp8. One HBM read of x, half the write bytes. */
int rt_layernorm_modulate_fp8(const void* x, int64_t ldx, int64_t stride_xb, int32_t x_f32,
                              void* out, int64_t ldo, int64_t stride_ob, float* row_scale,
                              const float* shift, const float* scale, int64_t mod_ld,
                              int32_t batch, int32_t rows_per_batch, int32_t D, float eps, void* stream);

/* RMSNorm(Dh, weight, eps) on q and k heads + interleaved-pair RoPE, in place on a fused
 * projection buffer (A.1 steps 3,5; A.2). Row (b,s) holds q at column q_off and k at k_off,
 * H heads of Dh=128 each. Rows s < T use the text weights (norm_added_q/k), others the image
 * weights; pass T = 0 for single-stream blocks. cos/sin f32 [S][128]. */
int rt_qk_rmsnorm_rope(void* buf, int64_t ld, int64_t stride_b, int64_t q_off, int64_t k_off,
                       const void* wq_txt, const void* wk_txt, const void* wq_img, const void* wk_img,
                       const float* cosv, const float* sinv,
                       int32_t B, int32_t S, int32_t T, int32_t H, float eps, void* stream);

/* Joint (non-causal, unmasked) attention, softmax(QKᵀ·scale)V, Dh = 128, flash-style on MFMA
 * (A.1 step 6; torch SDPA in the reference). q/k/v/o are bf16 with a common row stride ld
 * (elements) and per-batch stride; head h lives at column h*128. o may alias q (the query rows of
 * a 128-row block are overwritten only after every workgroup that reads them has finished).
 *
 * `ws` (optional, 256-byte aligned, rt_attention_ws_bytes(B,S,H) bytes): workspace of the key-split
 * tail. When (heads x query blocks) does not fill a whole number of rounds of the chip's workgroup
 * slots, the key tiles of the last partial round are dealt evenly over all slots and the partial
 * (max, sum, O) triples are combined in-kernel by the last arriver, in a fixed order (bitwise
 * reproducible; every batch entry is cut identically, so results do not depend on B). The first
 * B*H*ceil(S/128) int32 of ws are ticket counters: zero them ONCE after allocation; the kernel
 * leaves them zero. ws == NULL runs every block as one full-length workgroup.
 * rt_attention_ws_bytes returns 0 when nothing would be split for this shape on this device. */
int64_t rt_attention_ws_bytes(int32_t B, int32_t S, int32_t H);
int rt_attention_fwd(const void* q, const void* k, const void* v, void* o,
                     int64_t ld, int64_t stride_b, int64_t ldo, int64_t stride_ob,
                     int32_t B, int32_t S, int32_t H, float scale,
                     void* ws, int64_t ws_bytes, void* stream);

/* Which kernel serves rt_attention_fwd (speed only; both are tested against the same references): 1 (default, env RT_ATTN_V3) =
 * shapes with S % 256 == 0 and S >= 1536 run csrc/attention_v3.hip — one wave per SIMD, 64 query rows per wave, O / Q / row sums
 * in asm-owned accumulator registers, hand-placed MFMA gaps (S = 4608: 247 vs 264 us) —, everything else csrc/attention.hip;
 * 0 = attention.hip always; 2 = attention_v3 wherever S % 256 == 0. mode < 0 only queries; returns the previous mode. */
int rt_attention_variant(int32_t mode);

/* BASELINE config 5, "CDNA4 fp8 MFMA attention": the same joint attention with e4m3 q, k, v and softmax numerators on
 * v_mfma_scale_f32_32x32x64_f8f6f4 (fp32 scores, statistics and accumulators; bf16 output). Static quantisation: q and k
 * (RMS-normalised) are stored as e4m3(16·x), v is cast as is, no calibration pass.
 * rt_attention_fp8_prep replaces rt_qk_rmsnorm_rope on this path: from the fused projection buffer (row (b,s): q at q_off,
 * k at k_off, v at v_off, H heads of 128) it writes qk8 = [B][S][2·H·128] bytes (normalised, rotated q | k) and
 * vt8 = [B][H][128][S64] bytes (S64 = S rounded up to 64; Vᵀ with the keys of every 64-key tile permuted into MFMA operand
 * order, zero beyond S); rt_attention_fp8_vt_bytes gives vt8's size. buf is not modified. */
int64_t rt_attention_fp8_vt_bytes(int32_t B, int32_t S, int32_t H);
int rt_attention_fp8_prep(const void* buf, int64_t ld, int64_t stride_b, int64_t q_off, int64_t k_off, int64_t v_off,
                          const void* wq_txt, const void* wk_txt, const void* wq_img, const void* wk_img,
                          const float* cosv, const float* sinv, void* qk8, void* vt8,
                          int32_t B, int32_t S, int32_t T, int32_t H, float eps, void* stream);
int rt_attention_fp8_fwd(const void* qk8, const void* vt8, void* o, int64_t ldo, int64_t stride_ob,
                         int32_t B, int32_t S, int32_t H, float scale, void* stream);
/* The same attention with the output written as the NEXT projection's e4m3 A operand (config 5, "mx" level): o8[b][q][h*128 + d]
 * = e4m3(o * 2^(127-s)) with one E8M0 byte s per (row, 32 columns) in rt_gemm_group's scale layout (rt_quantize_mx_fp8's rule
 * applied to the fp32 output; no bf16 store and no quantisation pass in between): byte(b*bscale_rows + q, bscale_k0 + h*128 + d).
 * bscale_rows % 64 == 0, bscale_k0 % 128 == 0. Replaces `F.scaled_dot_product_attention` + the operand cast of to_out /
 * proj_out (A.1 steps 6-7, A.2) on the e4m3 path. */
int rt_attention_fp8_fwd_mx(const void* qk8, const void* vt8, void* o8, int64_t ldo8, int64_t stride_ob8, uint8_t* bscale,
                            int64_t plane, int64_t bscale_rows, int32_t bscale_k0, int32_t B, int32_t S, int32_t H, float scale,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * Prompt encoders (PIPE:232-347: T5-XXL encoder -> prompt_embeds [B,512,4096], CLIP-L text model ->
 * pooled_prompt_embeds [B,768]). Once per prompt, outside the loop. Matrix work is rt_gemm_bf16; attention (head dim 64)
 * is assembled per head from rt_gemm_bf16 / rt_softmax_rows_bias / rt_transpose_bf16 / rt_gemm_bf16.
 * ---------------------------------------------------------------------------------------- */
/* out[i][:] = table[clamp(ids[i])][:] — nn.Embedding lookup of bf16 rows (T5 `shared`, CLIP token_embedding). ids: device int32. */
int rt_embedding_gather(const void* table, int64_t ld, const int32_t* ids, void* out, int64_t ldo,
                        int32_t n, int32_t D, int32_t vocab, void* stream);
/* T5LayerNorm: out = x * rsqrt(mean(x^2) + eps) * w, x bf16 or f32 (x_f32), w/out bf16. No mean subtraction, no bias. */
int rt_rmsnorm_rows(const void* x, int64_t ldx, int32_t x_f32, const void* w, void* out, int64_t ldo,
                    int32_t rows, int32_t D, float eps, void* stream);
/* p = softmax(scale·s + bias) row-wise: s f32 [rows][lds], bias f32 [rows][ldb] or NULL (T5 relative-position bias, CLIP causal
 * mask as -inf), p bf16 [rows][ldp] with columns cols..cols_out-1 zero-filled (K padding of the P·V GEMM). A fully masked row
 * gives zeros. */
int rt_softmax_rows_bias(const float* s, int64_t lds, const float* bias, int64_t ldb, void* p, int64_t ldp,
                         int32_t rows, int32_t cols, int32_t cols_out, float scale, void* stream);
/* T5 v1.1 gated-GELU tail: out[r][c] = x[r][c] · x[r][F+c], bf16 (one half already activated by the GEMM epilogue). */
int rt_gated_mul(const void* x, int64_t ldx, void* out, int64_t ldo, int32_t rows, int32_t F, void* stream);
/* CLIP quick_gelu in place on n bf16 values: x · sigmoid(1.702 x). */
int rt_quick_gelu(void* x, int64_t n, void* stream);

/* FlowMatchEulerDiscreteScheduler.step (PIPE:1109; A.6): x = bf16(f32(x) + dsigma·f32(v)), in place. */
int rt_euler_step(void* x, const void* v, float dsigma, int64_t n, void* stream);

/* Same step on an fp32 master copy of the latents (x32 += dsigma·v, v bf16), optionally writing the bf16 copy the next model
 * call reads. diffusers computes the step in fp32 and rounds the STATE back to bf16 every step; keeping the state in fp32
 * removes 28 accumulated roundings (the fp32 CPU reference path keeps fp32 throughout). */
int rt_euler_step_f32(float* x, const void* v, void* x_bf16 /* nullable */, float dsigma, int64_t n, void* stream);

/* True-CFG mix of the inpaint pipeline (INP:1264-1270): out = uncond + s·(text − uncond); bf16. */
int rt_cfg_mix(const void* v_uncond, const void* v_text, void* out, float s, int64_t n, void* stream);

/* _pack_latents / _unpack_latents (PIPE:550-570): [B][C][2h][2w] <-> [B][h*w][4C], channel order (c,dy,dx).
 * unpack also applies z/scaling + shift (PIPE:1137) and writes NHWC bf16 [B][H2][W2][C]. */
int rt_pack_latents(const void* nchw, void* packed, int32_t B, int32_t C, int32_t H2, int32_t W2, void* stream);
int rt_unpack_latents(const void* packed, void* nhwc, int32_t B, int32_t C, int32_t H2, int32_t W2,
                      float inv_scale, float shift, void* stream);

/* Elementwise helpers on the path. */
int rt_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);
int rt_cast_bf16_to_f32(const void* x, float* y, int64_t n, void* stream);
/* hi = bf16(f(x)), lo = bf16(f(x) - hi), f = SiLU or identity: lets the adaLN projections of ALL denoising steps run as one
 * M = steps GEMM on MFMA (hi pass + lo pass accumulate) with fp32-activation accuracy (A.1 step 1, hoisted out of PIPE:1017). */
int rt_silu_split_bf16(const float* x, void* hi, void* lo, int64_t n, int32_t apply_silu, void* stream);
/* y[b][r][:] (+)= alpha · rowscale[r] · x[b][r][:]  — PIPE:1060-1087 masked sum over text lines; x bf16, y bf16 or f32 (y_f32). */
int rt_masked_accumulate(const void* x, void* y, const float* rowscale, float alpha,
                         int32_t batch, int32_t rows, int32_t D, int32_t accumulate, int32_t y_f32, void* stream);

/* ------------------------------------------------------------------------------------------
 * AutoencoderKL (PIPE:467,705,711 encode; PIPE:1139 decode; Appendix A.7).
 * Activations are zero-haloed NHWC bf16: [B][H+2][W+2][C]; the halo is allocated zeroed by the caller and never
 * written by these kernels, so 3x3 gathers need no bounds checks.
 * ---------------------------------------------------------------------------------------- */
/* GroupNorm(G groups, eps, affine gamma/beta bf16 [C]) + optional SiLU over the interior of x -> interior of y.
 * stats_ws: device scratch of rt_groupnorm_ws_bytes(B,H,W,G) bytes (8-byte aligned; need not be zeroed). All sums run in a
 * fixed order (no atomics): results are bitwise reproducible. C % 8 == 0, C % G == 0, 256 % (C/8) == 0. */
int64_t rt_groupnorm_ws_bytes(int32_t B, int32_t H, int32_t W, int32_t G);
int rt_groupnorm_silu_nhwc(const void* x, void* y, const void* gamma, const void* beta, void* stats_ws,
                           int32_t B, int32_t H, int32_t W, int32_t C, int32_t G, float eps, int32_t silu, void* stream);
/* 3x3 / 1x1 convolution as implicit GEMM on MFMA. x haloed [B][Hs+2][Ws+2][Cin], w bf16 [Cout][k][k][Cin] (repacked
 * from OIHW), y haloed [B][Ho+2][Wo+2][Cout] bf16|f32, res (optional) like y in bf16.
 *   stride 1: pad k/2; upsample2x fuses a nearest-2x Upsample2D in front (Ho = 2Hs)
 *   stride 2: k = 3, pad (0,1,0,1) = diffusers Downsample2D (Ho = Hs/2)
 * Cin % 64 == 0 (callers zero-pad channels), Cout % 4 == 0. */
int rt_conv2d_nhwc(const void* x, const void* w, const void* bias, const void* res, void* y,
                   int32_t B, int32_t Hs, int32_t Ws, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride,
                   int32_t upsample2x, int32_t out_f32, void* stream);
/* Which kernel serves rt_conv2d_nhwc's stride-1, non-upsampling, bf16-output calls with Cout >= 64 (speed only: both accumulate every
 * output element in the same K order, results are bit-identical): 1 (default, env RT_CONV_GEMM) = rt_gemm_bf16's convolution form,
 * 0 = conv_nhwc_kernel. mode >= 0 sets it, mode < 0 only queries; returns the previous mode. */
int rt_conv2d_variant(int32_t mode);
/* Row softmax for the VAE mid-block attention (1 head, Dh = 512, computed as GEMM -> softmax -> GEMM):
 * p[r][:] = softmax(scale * s[r][:]), f32 in, bf16 out. cols % 4 == 0. */
int rt_softmax_rows(const float* s, void* p, int32_t rows, int32_t cols, float scale, void* stream);
/* Mid-block attention of the AutoencoderKL, flash-style (A.7: Attention(C, 1 head) over the H*W positions of the 1/8-scale grid;
 * PIPE:1139 decode, PIPE:467,705,711 encode): o = softmax(q k^T * scale) v with ONE head of C channels, C in {128, 256, 512}.
 * q/k/v: bf16 [B][HW][>=C] views with a common row stride ld and batch stride (elements) — the fused q|k|v projection buffer;
 * o: bf16 [B][HW][ldo]. HW % 32 == 0. No (HW x HW) buffer exists: the channels are split over the waves of a workgroup, which
 * exchange partial scores through LDS (csrc/vae_attention.hip). Replaces GEMM -> rt_softmax_rows -> rt_transpose_bf16 -> GEMM. */
int rt_vae_attention(const void* q, const void* k, const void* v, void* o, int64_t ld, int64_t stride_b, int64_t ldo, int64_t stride_ob,
                     int32_t B, int32_t HW, int32_t C, float scale, void* stream);
/* out[c][r] = in[r][c], bf16 (V^T for the P·V GEMM). */
int rt_transpose_bf16(const void* in, void* out, int32_t R, int32_t C, int64_t ld_in, int64_t ld_out, void* stream);
/* Decoder tail: haloed NHWC f32 [B][H+2][W+2][Cp] -> NCHW f32 [B][C][H][W] (nchw, optional) and/or uint8 HWC
 * round(clamp(x/2+0.5,0,1)*255) (u8, optional) = VaeImageProcessor.postprocess (PIPE:1140). */
int rt_image_out(const float* x, float* nchw, uint8_t* u8, int32_t B, int32_t H, int32_t W, int32_t Cp, int32_t C, void* stream);
/* Encoder head / readout: NCHW f32 <-> haloed NHWC bf16 with channel padding to Cp. */
int rt_nchw_to_haloed_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Cp, void* stream);
int rt_haloed_nhwc_to_nchw(const void* x, float* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Cp, void* stream);
/* _unpack_latents + z/scaling + shift (PIPE:1136-1137) straight into the decoder's haloed NHWC input. */
int rt_unpack_latents_haloed(const void* packed, void* y, int32_t B, int32_t C, int32_t H2, int32_t W2, int32_t Cp,
                             float inv_scale, float shift, void* stream);


/* Hint-side resizes of the pipelines on the device, with torch.nn.functional.interpolate's rules
 * (align_corners=False, no antialias; the same fp32 expression order as ATen, so results are bit-identical):
 *   PIPE:1010-1012  regional mask / 255 -> bilinear x 1/16          (in_u8 = 1, in_scale = 255, scale = 1/16)
 *   INP:813         inpaint mask -> nearest, to the latent grid      (bilinear = 0, scale = 0: ratio = in/out)
 * `in`: uint8 (in_u8, divided by in_scale first) or f32 [planes][H][W]; out f32 [planes][OH][OW]; scale_h/scale_w > 0 are the
 * `scale_factor` torch was given (ratio = 1/scale), 0 = derive the ratio from the sizes. */
int rt_resize2d(const void* in, int32_t in_u8, float in_scale, float* out, int32_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW,
                float scale_h, float scale_w, int32_t bilinear, void* stream);
/* PIPE:645-654 / INP:640-647: out = 0.10 * latents + noise where the glyph mask (any channel of image > 0), bilinearly resized
 * to the latent grid, is > 0; noise elsewhere. image f32 [B][Cimg][H][W]; latents, noise, out f32 [B][Cl][OH][OW]. */
int rt_glyph_blend(const float* image, const float* latents, const float* noise, float* out, int32_t B, int32_t Cimg, int32_t H, int32_t W,
                   int32_t Cl, int32_t OH, int32_t OW, void* stream);

/* ------------------------------------------------------------------------------------------
 * Caller-side hint preparation on the device (csrc/hints.hip).
 * ---------------------------------------------------------------------------------------- */
/* cv2.Canny(image, low, high) of infer.py:16-22 (called at infer.py:98-100 on the rendered glyph) as cv::Canny documents it —
 * 3x3 Sobel with replicated borders taken per channel (the channel with the largest |dx|+|dy| supplies the gradient, first
 * channel on ties; no gray conversion), L1 magnitude, 4-sector non-maximum suppression with cv::Canny's tie rules, double
 * threshold, 8-connected hysteresis — bit-identical to reptext_amd/hints.py::canny_edges (parity with OpenCV itself is unpinned).
 * img: uint8 [H][W][C] (C = 1..4, interleaved); out: uint8 [H][W][out_channels], every channel = edge map {0,255}, or
 * 255 - edges when `invert` (infer.py:20-22: the hint is the inverted map repeated over 3 channels).
 * ws: rt_canny_ws_bytes(H, W) bytes of scratch, 256-byte aligned; need not be zeroed. Four kernels on `stream`, no host
 * round trip (hysteresis = one workgroup's breadth-first search over index frontiers in ws). */
int64_t rt_canny_ws_bytes(int32_t H, int32_t W);
int rt_canny_u8(const uint8_t* img, int32_t H, int32_t W, int32_t C, float low, float high, uint8_t* out, int32_t out_channels,
                int32_t invert, void* ws, int64_t ws_bytes, void* stream);
/* VaeImageProcessor.preprocess (PIPE:680,694,970) for uint8 images that already have the target size: img [B][H][W][C] ->
 * out f32 [B][C][H][W] = x / 255, then 2x - 1 when `normalize` (the same fp32 roundings numpy/torch take: bit-identical). */
int rt_preprocess_u8(const uint8_t* img, float* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t normalize, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* REPTEXT_HIP_H */
