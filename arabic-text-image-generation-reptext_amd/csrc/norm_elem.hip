// HBM-bound kernels of the denoising path: LayerNorm+modulation, q/k RMSNorm+RoPE, adaLN GEMV,
// timestep/RoPE tables, Euler step, latent pack/unpack, casts. All loads/stores are 8–16 B per lane,
// statistics in fp32. Math per SURVEY.md Appendix A.1/A.2/A.5/A.6 and PIPE:550-570,1109.
#include "rt_common.h"
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------------------------------
// LayerNorm (no affine) + (1+scale)·x + shift. One wave per row, row kept in registers (single HBM read).
// ---------------------------------------------------------------------------------------------------
// 8 floats -> 8 e4m3 bytes (OCP e4m3fn on gfx950), inputs already divided by the row scale and clamped to +-448
__device__ __forceinline__ u32x2 pack_e4m3x8(const float (&y)[8]) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(y[4], y[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(y[6], y[7], hi, true);
  return u32x2{(uint32_t)lo, (uint32_t)hi};
}

constexpr float E4M3_MAX = 448.f;

template <bool X_F32, int NCH, bool OUT_FP8 = false>   // NCH = chunks of 8 elements per lane  (D <= 64*8*NCH)
__global__ __launch_bounds__(256) void layernorm_mod_kernel(
    const void* __restrict__ x, int64_t ldx, int64_t stride_xb, bf16_t* __restrict__ out, int64_t ldo,
    int64_t stride_ob, const float* __restrict__ shift, const float* __restrict__ scale, int64_t mod_ld,
    int batch, int rows_per_batch, int D, float eps, float* __restrict__ row_scale = nullptr) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= batch * rows_per_batch) return;
  const int b = row / rows_per_batch, r = row - b * rows_per_batch;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) {
      if (X_F32) {
        const float* p = reinterpret_cast<const float*>(x) + b * stride_xb + (int64_t)r * ldx + col;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), bb = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[c][i] = a[i]; v[c][4 + i] = bb[i]; }
      } else {
        const bf16_t* p = reinterpret_cast<const bf16_t*>(x) + b * stride_xb + (int64_t)r * ldx + col;
        const u32x4 u = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[c][2 * i] = bf16lo(u[i]); v[c][2 * i + 1] = bf16hi(u[i]); }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) s += v[c][i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[c][i] = 0.f;
    }
  }
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) {
      float y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = (v[c][i] - mean) * rstd;
      if (scale) {
        const float* sc = scale + b * mod_ld + col;
        const float* sh = shift + b * mod_ld + col;
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh), h1 = *reinterpret_cast<const f32x4*>(sh + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          y[i] = y[i] * (1.f + s0[i]) + h0[i];
          y[4 + i] = y[4 + i] * (1.f + s1[i]) + h1[i];
        }
      }
      if constexpr (OUT_FP8) {                       // keep the modulated row in registers; quantise after the row max is known
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[c][i] = y[i]; amax = fmaxf(amax, fabsf(y[i])); }
      } else {
        u32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(y[2 * i], y[2 * i + 1]);
        *reinterpret_cast<u32x4*>(out + b * stride_ob + (int64_t)r * ldo + col) = o;
      }
    }
  }
  if constexpr (OUT_FP8) {
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.f / E4M3_MAX) : 1.f;
    const float inv = 1.f / sc;
    if (lane == 0) row_scale[row] = sc;
    uint8_t* o8 = reinterpret_cast<uint8_t*>(out) + b * stride_ob + (int64_t)r * ldo;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * 8;
      if (col < D) {
        float y[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = fminf(fmaxf(v[c][i] * inv, -E4M3_MAX), E4M3_MAX);
        *reinterpret_cast<u32x2*>(o8 + col) = pack_e4m3x8(y);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Row-wise e4m3 quantisation (weights once at load; activation rows that do not come out of a LayerNorm).
// One wave per row. bf16 rows of up to 64·8·NCH elements are read ONCE and held packed in registers between the max pass and
// the convert pass (quantize_rows_fp8_reg_kernel); longer or f32 rows are read twice (quantize_rows_fp8_kernel).
// ---------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void quantize_rows_fp8_reg_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ out,
                                                                    int64_t ldo, float* __restrict__ scale, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  u32x4 v[NCH];
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    v[c] = u32x4{0u, 0u, 0u, 0u};
    if (col < D) v[c] = *reinterpret_cast<const u32x4*>(x + (int64_t)row * ldx + col);
#pragma unroll
    for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fmaxf(fabsf(bf16lo(v[c][i])), fabsf(bf16hi(v[c][i]))));
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax * (1.f / E4M3_MAX) : 1.f;
  const float inv = 1.f / sc;
  if (lane == 0) scale[row] = sc;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = (c * 64 + lane) * 8;
    if (col < D) {
      float y[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        y[2 * i] = fminf(fmaxf(bf16lo(v[c][i]) * inv, -E4M3_MAX), E4M3_MAX);
        y[2 * i + 1] = fminf(fmaxf(bf16hi(v[c][i]) * inv, -E4M3_MAX), E4M3_MAX);
      }
      *reinterpret_cast<u32x2*>(out + (int64_t)row * ldo + col) = pack_e4m3x8(y);
    }
  }
}

template <bool X_F32>
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const void* __restrict__ x, int64_t ldx, uint8_t* __restrict__ out,
                                                                int64_t ldo, float* __restrict__ scale, int rows, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  auto load8 = [&](int col, float (&y)[8]) {
    if (X_F32) {
      const float* p = reinterpret_cast<const float*>(x) + (int64_t)row * ldx + col;
      const f32x4 a = *reinterpret_cast<const f32x4*>(p), bb = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { y[i] = a[i]; y[4 + i] = bb[i]; }
    } else {
      const u32x4 u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(x) + (int64_t)row * ldx + col);
#pragma unroll
      for (int i = 0; i < 4; ++i) { y[2 * i] = bf16lo(u[i]); y[2 * i + 1] = bf16hi(u[i]); }
    }
  };
  float amax = 0.f;
  for (int col = lane * 8; col < D; col += 512) {
    float y[8];
    load8(col, y);
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(y[i]));
  }
  amax = wave_max(amax);
  const float sc = amax > 0.f ? amax * (1.f / E4M3_MAX) : 1.f;
  const float inv = 1.f / sc;
  if (lane == 0) scale[row] = sc;
  for (int col = lane * 8; col < D; col += 512) {
    float y[8];
    load8(col, y);
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = fminf(fmaxf(y[i] * inv, -E4M3_MAX), E4M3_MAX);
    *reinterpret_cast<u32x2*>(out + (int64_t)row * ldo + col) = pack_e4m3x8(y);
  }
}

// MX block quantisation (reptext_hip.h: rt_quantize_mx_fp8): 8 elements per thread, 4 threads per 32-element block.
template <bool X_F32>
__global__ __launch_bounds__(256) void quantize_mx_fp8_kernel(const void* __restrict__ x, int64_t ldx, uint8_t* __restrict__ out, int64_t ldo,
                                                              uint8_t* __restrict__ bscale, int64_t plane, int rows, int D) {
  const int per_row = D >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int row = (int)(idx / per_row);
  const int col = (int)(idx - (int64_t)row * per_row) * 8;
  const bool ok = row < rows;                             // whole 4-lane groups are in or out together (per_row % 32 == 0)
  float y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) y[i] = 0.f;
  if (ok) {
    if (X_F32) {
      const float* p = reinterpret_cast<const float*>(x) + (int64_t)row * ldx + col;
      const f32x4 a = *reinterpret_cast<const f32x4*>(p), bb = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { y[i] = a[i]; y[4 + i] = bb[i]; }
    } else {
      const u32x4 u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(x) + (int64_t)row * ldx + col);
#pragma unroll
      for (int i = 0; i < 4; ++i) { y[2 * i] = bf16lo(u[i]); y[2 * i + 1] = bf16hi(u[i]); }
    }
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(y[i]));
  amax = fmaxf(amax, __shfl_xor(amax, 1));
  amax = fmaxf(amax, __shfl_xor(amax, 2));
  const int sb = rt_mx_scale_byte(amax);
  const float inv = rt_mx_inv_scale(sb);
  if (!ok) return;
#pragma unroll
  for (int i = 0; i < 8; ++i) y[i] = fminf(fmaxf(y[i] * inv, -E4M3_MAX), E4M3_MAX);
  *reinterpret_cast<u32x2*>(out + (int64_t)row * ldo + col) = pack_e4m3x8(y);
  if ((threadIdx.x & 3) == 0) bscale[rt_mx_scale_offset(row, col, plane)] = (uint8_t)sb;
}

// ---------------------------------------------------------------------------------------------------
// q/k RMSNorm(128) + RoPE in place. 16 lanes per 128-vector (8 elements = 4 rotation pairs each),
// 4 vectors per wave. Grid covers B*S*H*2 vectors (q and k).
// ---------------------------------------------------------------------------------------------------
// A 16-lane group takes one token row and QK_HC heads of it (q and k: 2·QK_HC vectors of 128), so the row's cos/sin entries and
// the norm weights are loaded once per 2·QK_HC vectors. Measured at 4608×24 heads: one vector per group 21.6 µs, QK_HC = 1 19.7 µs,
// 2: 22.9, 4: 33 — the kernel lives on wave-level parallelism, longer per-lane chains lose more than the table bytes they save.
constexpr int QK_HC = 1;
__global__ __launch_bounds__(256) void qk_rmsnorm_rope_kernel(
    bf16_t* __restrict__ buf, int64_t ld, int64_t stride_b, int64_t q_off, int64_t k_off,
    const bf16_t* __restrict__ wq_txt, const bf16_t* __restrict__ wk_txt, const bf16_t* __restrict__ wq_img,
    const bf16_t* __restrict__ wk_img, const float* __restrict__ cosv, const float* __restrict__ sinv, int B, int S,
    int T, int H, float eps) {
  const int nchunk = (H + QK_HC - 1) / QK_HC;
  const int64_t grp = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);   // (row, head chunk)
  const int sub = threadIdx.x & 15;
  if (grp >= (int64_t)B * S * nchunk) return;   // whole 16-lane groups exit together; shuffles below stay within the group
  const int hc = (int)(grp % nchunk);
  const int64_t row = grp / nchunk;
  const int s = (int)(row % S);
  const int b = (int)(row / S);
  bf16_t* base = buf + b * stride_b + (int64_t)s * ld + sub * 8;
  const bf16_t* wq = (s < T) ? wq_txt : wq_img;
  const bf16_t* wk = (s < T) ? wk_txt : wk_img;
  const u32x4 wuq = *reinterpret_cast<const u32x4*>(wq + sub * 8), wuk = *reinterpret_cast<const u32x4*>(wk + sub * 8);
  const float* cp = cosv + (int64_t)s * 128 + sub * 8;
  const float* sp = sinv + (int64_t)s * 128 + sub * 8;
  const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp), c1 = *reinterpret_cast<const f32x4*>(cp + 4);
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
  const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
  const float sn[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
  u32x4 u[2 * QK_HC];
#pragma unroll
  for (int v = 0; v < 2 * QK_HC; ++v) {
    const int h = hc * QK_HC + (v >> 1);
    u[v] = u32x4{0u, 0u, 0u, 0u};
    if (h < H) u[v] = *reinterpret_cast<const u32x4*>(base + ((v & 1) ? k_off : q_off) + h * 128);
  }
#pragma unroll
  for (int v = 0; v < 2 * QK_HC; ++v) {
    const int h = hc * QK_HC + (v >> 1);
    const u32x4 wu = (v & 1) ? wuk : wuq;
    float x[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[2 * i] = bf16lo(u[v][i]); x[2 * i + 1] = bf16hi(u[v][i]); }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = x[2 * i] * r * bf16lo(wu[i]);
      const float bq = x[2 * i + 1] * r * bf16hi(wu[i]);
      o[i] = pack_bf16x2(a * cs[2 * i] - bq * sn[2 * i], bq * cs[2 * i + 1] + a * sn[2 * i + 1]);
    }
    if (h < H) *reinterpret_cast<u32x4*>(base + ((v & 1) ? k_off : q_off) + h * 128) = o;
  }
}

// ---------------------------------------------------------------------------------------------------
// GEMV on fp32 activations, bf16 weights: y[b][n] (+)= post(Σ_k pre(x[b][k]) W[n][k] + bias[n]).
// One wave per output row n; x staged (pre-activated) in LDS as fp32; B <= 8 per launch.
// ---------------------------------------------------------------------------------------------------
constexpr int GEMV_MAXB = 8;
__global__ __launch_bounds__(256) void gemv_bf16w_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const bf16_t* __restrict__ W, int64_t ldw,
                                                          const bf16_t* __restrict__ bias, float* __restrict__ y,
                                                          int64_t ldy, int B, int N, int K, int silu_in, int silu_out,
                                                          int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xs = reinterpret_cast<float*>(smem);   // [B][K]
  for (int i = threadIdx.x; i < B * K; i += blockDim.x) {
    const int b = i / K, k = i - b * K;
    float v = x[b * ldx + k];
    xs[i] = silu_in ? silu_f(v) : v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  constexpr int ROWS_PER_WAVE = 4;
  for (int rr = 0; rr < ROWS_PER_WAVE; ++rr) {
    const int n = (blockIdx.x * 4 + wave) * ROWS_PER_WAVE + rr;
    if (n >= N) break;
    float acc[GEMV_MAXB];
#pragma unroll
    for (int b = 0; b < GEMV_MAXB; ++b) acc[b] = 0.f;
    const bf16_t* wr = W + (int64_t)n * ldw;
    for (int k = lane * 8; k < K; k += 512) {
      const u32x4 u = *reinterpret_cast<const u32x4*>(wr + k);
      float wv[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { wv[2 * i] = bf16lo(u[i]); wv[2 * i + 1] = bf16hi(u[i]); }
#pragma unroll
      for (int b = 0; b < GEMV_MAXB; ++b) {
        if (b < B) {
          const f32x4 a0 = *reinterpret_cast<const f32x4*>(xs + b * K + k);
          const f32x4 a1 = *reinterpret_cast<const f32x4*>(xs + b * K + k + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[b] += a0[i] * wv[i] + a1[i] * wv[4 + i];
        }
      }
    }
#pragma unroll
    for (int b = 0; b < GEMV_MAXB; ++b) {
      if (b < B) {
        float v = wave_sum(acc[b]);
        if (lane == 0) {
          if (bias) v += bf16_to_f32(bias[n]);
          if (silu_out) v = silu_f(v);
          float* yp = y + b * ldy + n;
          *yp = accumulate ? (*yp + v) : v;
        }
      }
    }
  }
}

// Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]
__global__ void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  const float f = expf(-logf(10000.0f) * (float)j / (float)half);
  const float a = t[b] * f;
  out[b * dim + j] = cosf(a);
  out[b * dim + half + j] = sinf(a);
}

// FluxPosEmbed: fp64 angles, interleaved repeat.
__global__ void rope_table_kernel(const float* __restrict__ ids, float* __restrict__ cosv, float* __restrict__ sinv,
                                  int S, int d0, int d1, int d2, float theta) {
  const int D = d0 + d1 + d2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * (D / 2)) return;
  const int s = i / (D / 2);
  int j = i - s * (D / 2);        // frequency index over concatenated axes
  int axis = 0, da = d0, off = 0;
  if (j >= d0 / 2) { j -= d0 / 2; axis = 1; da = d1; off = d0; }
  if (axis == 1 && j >= d1 / 2) { j -= d1 / 2; axis = 2; da = d2; off = d0 + d1; }
  const double omega = 1.0 / pow((double)theta, (double)(2 * j) / (double)da);
  const double ang = (double)ids[s * 3 + axis] * omega;
  const float c = (float)cos(ang), sn = (float)sin(ang);
  float* cp = cosv + (int64_t)s * D + off + 2 * j;
  float* sp = sinv + (int64_t)s * D + off + 2 * j;
  cp[0] = c; cp[1] = c; sp[0] = sn; sp[1] = sn;
}

__global__ void euler_step_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ v, float ds, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    u32x4 a = reinterpret_cast<u32x4*>(x)[i];
    const u32x4 b = reinterpret_cast<const u32x4*>(v)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      a[j] = pack_bf16x2(bf16lo(a[j]) + ds * bf16lo(b[j]), bf16hi(a[j]) + ds * bf16hi(b[j]));
    reinterpret_cast<u32x4*>(x)[i] = a;
  }
}
__global__ void euler_step_tail_kernel(bf16_t* x, const bf16_t* v, float ds, int64_t start, int64_t n) {
  const int64_t i = start + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = f32_to_bf16(bf16_to_f32(x[i]) + ds * bf16_to_f32(v[i]));
}

// fp32 master latents: x32 += ds * v (v bf16); also emits the bf16 copy the next step's x_embedder reads
__global__ void euler_step_f32_kernel(float* __restrict__ x, const bf16_t* __restrict__ v, bf16_t* __restrict__ xb, float ds,
                                      int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float r = x[i] + ds * bf16_to_f32(v[i]);
    x[i] = r;
    if (xb) xb[i] = f32_to_bf16(r);
  }
}

__global__ void cfg_mix_kernel(const bf16_t* __restrict__ u, const bf16_t* __restrict__ t, bf16_t* __restrict__ o,
                               float s, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = bf16_to_f32(u[i]), b = bf16_to_f32(t[i]);
    o[i] = f32_to_bf16(a + s * (b - a));
  }
}

// packed[b][(i*W+j)][c*4+dy*2+dx] = x[b][c][2i+dy][2j+dx]
__global__ void pack_latents_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ p, int B, int C, int H2, int W2) {
  const int64_t n = (int64_t)B * C * H2 * W2;
  const int h = H2 / 2, w = W2 / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = i;
    const int ch = (int)(t % (C * 4)); t /= (C * 4);
    const int j = (int)(t % w); t /= w;
    const int ii = (int)(t % h);
    const int b = (int)(t / h);
    const int c = ch >> 2, dy = (ch >> 1) & 1, dx = ch & 1;
    p[i] = x[(((int64_t)b * C + c) * H2 + (2 * ii + dy)) * W2 + 2 * j + dx];
  }
}
// inverse, with z/scaling + shift, output NHWC bf16 [B][H2][W2][C]
__global__ void unpack_latents_kernel(const bf16_t* __restrict__ p, bf16_t* __restrict__ y, int B, int C, int H2, int W2,
                                      float inv_scale, float shift) {
  const int64_t n = (int64_t)B * C * H2 * W2;
  const int h = H2 / 2, w = W2 / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = i;
    const int c = (int)(t % C); t /= C;
    const int xx = (int)(t % W2); t /= W2;
    const int yy = (int)(t % H2);
    const int b = (int)(t / H2);
    const int tok = (yy >> 1) * w + (xx >> 1);
    const int ch = c * 4 + (yy & 1) * 2 + (xx & 1);
    const float v = bf16_to_f32(p[((int64_t)b * h * w + tok) * (C * 4) + ch]);
    y[i] = f32_to_bf16(v * inv_scale + shift);
  }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = f32_to_bf16(x[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = bf16_to_f32(x[i]);
}

// fp32 destination variant (fp32 residual stream): y[b][r][:] (+)= alpha * rowscale[r] * x[b][r][:]
__global__ void masked_accumulate_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y,
                                             const float* __restrict__ rowscale, float alpha, int batch, int rows, int D8,
                                             int accumulate) {
  const int64_t n = (int64_t)batch * rows * D8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)((i / D8) % rows);
    const float m = alpha * (rowscale ? rowscale[r] : 1.0f);
    const u32x4 a = reinterpret_cast<const u32x4*>(x)[i];
    f32x4 lo = f32x4{bf16lo(a[0]), bf16hi(a[0]), bf16lo(a[1]), bf16hi(a[1])} * m;
    f32x4 hi = f32x4{bf16lo(a[2]), bf16hi(a[2]), bf16lo(a[3]), bf16hi(a[3])} * m;
    f32x4* yp = reinterpret_cast<f32x4*>(y) + 2 * i;
    if (accumulate) { lo += yp[0]; hi += yp[1]; }
    yp[0] = lo; yp[1] = hi;
  }
}

// y[b][r][:] (+)= alpha * rowscale[r] * x[b][r][:], 8 elements per lane
__global__ void masked_accumulate_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                         const float* __restrict__ rowscale, float alpha, int batch, int rows, int D8,
                                         int accumulate) {
  const int64_t n = (int64_t)batch * rows * D8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)((i / D8) % rows);
    const float m = alpha * (rowscale ? rowscale[r] : 1.0f);
    const u32x4 a = reinterpret_cast<const u32x4*>(x)[i];
    u32x4 o;
    if (accumulate) {
      const u32x4 c = reinterpret_cast<const u32x4*>(y)[i];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = pack_bf16x2(bf16lo(c[j]) + m * bf16lo(a[j]), bf16hi(c[j]) + m * bf16hi(a[j]));
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(m * bf16lo(a[j]), m * bf16hi(a[j]));
    }
    reinterpret_cast<u32x4*>(y)[i] = o;
  }
}

// hi = bf16(silu(x)), lo = bf16(silu(x) - hi): two-term bf16 split so a bf16 MFMA GEMM reproduces an fp32-activation product
__global__ void silu_split_kernel(const float* __restrict__ x, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int64_t n,
                                  int apply_silu) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = apply_silu ? silu_f(x[i]) : x[i];
    const bf16_t h = f32_to_bf16(v);
    hi[i] = h;
    lo[i] = f32_to_bf16(v - bf16_to_f32(h));
  }
}

inline int grid_for(int64_t n, int block) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" {

int rt_layernorm_modulate(const void* x, int64_t ldx, int64_t stride_xb, int32_t x_f32, void* out, int64_t ldo,
                          int64_t stride_ob, const float* shift, const float* scale, int64_t mod_ld, int32_t batch,
                          int32_t rows_per_batch, int32_t D, float eps, void* stream) {
  if (!x || !out || batch < 1 || rows_per_batch < 1 || D < 8) return RT_E_BADARG;
  if ((shift == nullptr) != (scale == nullptr)) return RT_E_BADARG;
  if (D % 8 || D > 8192) return RT_E_SHAPE;
  if (!RT_ALIGNED(x, 16) || !RT_ALIGNED(out, 16) || ldx % 8 || ldo % 8 || stride_xb % 8 || stride_ob % 8)
    return RT_E_ALIGN;
  if (scale && (!RT_ALIGNED(scale, 16) || !RT_ALIGNED(shift, 16) || mod_ld % 4)) return RT_E_ALIGN;
  const int rows = batch * rows_per_batch;
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int nch = (D + 511) / 512;
#define LN_LAUNCH(F32, N)                                                                                         \
  hipLaunchKernelGGL((layernorm_mod_kernel<F32, N>), grid, block, 0, st, x, ldx, stride_xb, (bf16_t*)out, ldo,   \
                     stride_ob, shift, scale, mod_ld, batch, rows_per_batch, D, eps)
  if (x_f32) {
    if (nch <= 1) LN_LAUNCH(true, 1); else if (nch <= 2) LN_LAUNCH(true, 2); else if (nch <= 4) LN_LAUNCH(true, 4);
    else if (nch <= 6) LN_LAUNCH(true, 6); else if (nch <= 8) LN_LAUNCH(true, 8); else LN_LAUNCH(true, 16);
  } else {
    if (nch <= 1) LN_LAUNCH(false, 1); else if (nch <= 2) LN_LAUNCH(false, 2); else if (nch <= 4) LN_LAUNCH(false, 4);
    else if (nch <= 6) LN_LAUNCH(false, 6); else if (nch <= 8) LN_LAUNCH(false, 8); else LN_LAUNCH(false, 16);
  }
#undef LN_LAUNCH
  return rt_hip_status();
}

int rt_layernorm_modulate_fp8(const void* x, int64_t ldx, int64_t stride_xb, int32_t x_f32, void* out, int64_t ldo,
                              int64_t stride_ob, float* row_scale, const float* shift, const float* scale, int64_t mod_ld,
                              int32_t batch, int32_t rows_per_batch, int32_t D, float eps, void* stream) {
  if (!x || !out || !row_scale || batch < 1 || rows_per_batch < 1 || D < 8) return RT_E_BADARG;
  if ((shift == nullptr) != (scale == nullptr)) return RT_E_BADARG;
  if (D % 8 || D > 8192) return RT_E_SHAPE;
  if (!RT_ALIGNED(x, 16) || !RT_ALIGNED(out, 8) || ldx % 8 || ldo % 8 || stride_xb % 8 || stride_ob % 8) return RT_E_ALIGN;
  if (scale && (!RT_ALIGNED(scale, 16) || !RT_ALIGNED(shift, 16) || mod_ld % 4)) return RT_E_ALIGN;
  const int rows = batch * rows_per_batch;
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int nch = (D + 511) / 512;
#define LN8_LAUNCH(F32, N)                                                                                        \
  hipLaunchKernelGGL((layernorm_mod_kernel<F32, N, true>), grid, block, 0, st, x, ldx, stride_xb, (bf16_t*)out, ldo, \
                     stride_ob, shift, scale, mod_ld, batch, rows_per_batch, D, eps, row_scale)
  if (x_f32) {
    if (nch <= 2) LN8_LAUNCH(true, 2); else if (nch <= 6) LN8_LAUNCH(true, 6); else LN8_LAUNCH(true, 16);
  } else {
    if (nch <= 2) LN8_LAUNCH(false, 2); else if (nch <= 6) LN8_LAUNCH(false, 6); else LN8_LAUNCH(false, 16);
  }
#undef LN8_LAUNCH
  return rt_hip_status();
}

int rt_quantize_rows_fp8(const void* x, int64_t ldx, int32_t x_f32, void* out, int64_t ldo, float* scale, int32_t rows,
                         int32_t D, void* stream) {
  if (!x || !out || !scale || rows < 1 || D < 8) return RT_E_BADARG;
  if (D % 8 || D > 65536) return RT_E_SHAPE;
  if (!RT_ALIGNED(x, 16) || !RT_ALIGNED(out, 8) || ldx % 8 || ldo % 8) return RT_E_ALIGN;
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  const int nch = (D + 511) / 512;
#define QR_LAUNCH(N) hipLaunchKernelGGL(quantize_rows_fp8_reg_kernel<N>, grid, block, 0, st, (const bf16_t*)x, ldx, (uint8_t*)out, ldo, scale, rows, D)
  if (x_f32) hipLaunchKernelGGL(quantize_rows_fp8_kernel<true>, grid, block, 0, st, x, ldx, (uint8_t*)out, ldo, scale, rows, D);
  else if (nch <= 6) QR_LAUNCH(6);
  else if (nch <= 12) QR_LAUNCH(12);
  else if (nch <= 24) QR_LAUNCH(24);
  else if (nch <= 30) QR_LAUNCH(30);
  else hipLaunchKernelGGL(quantize_rows_fp8_kernel<false>, grid, block, 0, st, x, ldx, (uint8_t*)out, ldo, scale, rows, D);
#undef QR_LAUNCH
  return rt_hip_status();
}

int rt_quantize_mx_fp8(const void* x, int64_t ldx, int32_t x_f32, void* out, int64_t ldo, uint8_t* bscale, int64_t plane,
                       int32_t rows, int32_t D, void* stream) {
  if (!x || !out || !bscale || rows < 1 || D < 256) return RT_E_BADARG;
  if (D % 256 || plane % 2048 || plane < (((int64_t)rows + 63) / 64) * 2048) return RT_E_SHAPE;
  if (!RT_ALIGNED(x, 16) || !RT_ALIGNED(out, 8) || ldx % 8 || ldo % 8 || !RT_ALIGNED(bscale, 16)) return RT_E_ALIGN;
  const int64_t n = (int64_t)rows * (D / 8);
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (x_f32) hipLaunchKernelGGL(quantize_mx_fp8_kernel<true>, grid, block, 0, st, x, ldx, (uint8_t*)out, ldo, bscale, plane, rows, D);
  else hipLaunchKernelGGL(quantize_mx_fp8_kernel<false>, grid, block, 0, st, x, ldx, (uint8_t*)out, ldo, bscale, plane, rows, D);
  return rt_hip_status();
}

int rt_qk_rmsnorm_rope(void* buf, int64_t ld, int64_t stride_b, int64_t q_off, int64_t k_off, const void* wq_txt,
                       const void* wk_txt, const void* wq_img, const void* wk_img, const float* cosv,
                       const float* sinv, int32_t B, int32_t S, int32_t T, int32_t H, float eps, void* stream) {
  if (!buf || !wq_img || !wk_img || !cosv || !sinv || B < 1 || S < 1 || H < 1 || T < 0 || T > S) return RT_E_BADARG;
  if (T > 0 && (!wq_txt || !wk_txt)) return RT_E_BADARG;
  if (!RT_ALIGNED(buf, 16) || ld % 8 || stride_b % 8 || q_off % 8 || k_off % 8 || !RT_ALIGNED(cosv, 16) ||
      !RT_ALIGNED(sinv, 16) || !RT_ALIGNED(wq_img, 16) || !RT_ALIGNED(wk_img, 16))
    return RT_E_ALIGN;
  const int64_t ngrp = (int64_t)B * S * ((H + QK_HC - 1) / QK_HC);
  const int64_t blocks = (ngrp + 15) / 16;
  if (blocks > 0x7fffffff) return RT_E_SHAPE;
  hipLaunchKernelGGL(qk_rmsnorm_rope_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (bf16_t*)buf, ld, stride_b, q_off, k_off, (const bf16_t*)wq_txt, (const bf16_t*)wk_txt,
                     (const bf16_t*)wq_img, (const bf16_t*)wk_img, cosv, sinv, B, S, T, H, eps);
  return rt_hip_status();
}

int rt_gemv_bf16w(const float* x, int64_t ldx, const void* W, int64_t ldw, const void* bias, float* y, int64_t ldy,
                  int32_t B, int32_t N, int32_t K, int32_t silu_in, int32_t silu_out, int32_t accumulate,
                  void* stream) {
  if (!x || !W || !y || B < 1 || N < 1 || K < 1) return RT_E_BADARG;
  if (B > GEMV_MAXB || K % 8 || (int64_t)B * K * 4 > 64 * 1024) return RT_E_SHAPE;
  if (!RT_ALIGNED(W, 16) || ldw % 8) return RT_E_ALIGN;
  const int rows_per_block = 16;
  hipLaunchKernelGGL(gemv_bf16w_kernel, dim3((N + rows_per_block - 1) / rows_per_block), dim3(256),
                     (size_t)B * K * 4, (hipStream_t)stream, x, ldx, (const bf16_t*)W, ldw, (const bf16_t*)bias, y,
                     ldy, B, N, K, silu_in, silu_out, accumulate);
  return rt_hip_status();
}

int rt_timestep_embedding(const float* t, float* out, int32_t B, int32_t dim, void* stream) {
  if (!t || !out || B < 1 || dim < 2 || dim % 2) return RT_E_BADARG;
  const int n = B * dim / 2;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, out, B, dim);
  return rt_hip_status();
}

int rt_rope_table(const float* ids, float* cos_out, float* sin_out, int32_t S, const int32_t* axes_dim, float theta,
                  void* stream) {
  if (!ids || !cos_out || !sin_out || !axes_dim || S < 1) return RT_E_BADARG;
  const int d0 = axes_dim[0], d1 = axes_dim[1], d2 = axes_dim[2];
  if (d0 < 2 || d1 < 2 || d2 < 2 || d0 % 2 || d1 % 2 || d2 % 2) return RT_E_SHAPE;
  const int n = S * ((d0 + d1 + d2) / 2);
  hipLaunchKernelGGL(rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ids, cos_out, sin_out,
                     S, d0, d1, d2, theta);
  return rt_hip_status();
}

int rt_euler_step(void* x, const void* v, float dsigma, int64_t n, void* stream) {
  if (!x || !v || n < 1) return RT_E_BADARG;
  if (!RT_ALIGNED(x, 16) || !RT_ALIGNED(v, 16)) return RT_E_ALIGN;
  const int64_t n8 = n / 8;
  if (n8 > 0)
    hipLaunchKernelGGL(euler_step_kernel, dim3(grid_for(n8, 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x,
                       (const bf16_t*)v, dsigma, n8);
  if (n8 * 8 < n)
    hipLaunchKernelGGL(euler_step_tail_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (bf16_t*)x, (const bf16_t*)v,
                       dsigma, n8 * 8, n);
  return rt_hip_status();
}

int rt_euler_step_f32(float* x, const void* v, void* x_bf16, float dsigma, int64_t n, void* stream) {
  if (!x || !v || n < 1) return RT_E_BADARG;
  hipLaunchKernelGGL(euler_step_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)v,
                     (bf16_t*)x_bf16, dsigma, n);
  return rt_hip_status();
}

int rt_cfg_mix(const void* v_uncond, const void* v_text, void* out, float s, int64_t n, void* stream) {
  if (!v_uncond || !v_text || !out || n < 1) return RT_E_BADARG;
  hipLaunchKernelGGL(cfg_mix_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)v_uncond,
                     (const bf16_t*)v_text, (bf16_t*)out, s, n);
  return rt_hip_status();
}

int rt_pack_latents(const void* nchw, void* packed, int32_t B, int32_t C, int32_t H2, int32_t W2, void* stream) {
  if (!nchw || !packed || B < 1 || C < 1 || H2 < 2 || W2 < 2 || H2 % 2 || W2 % 2) return RT_E_BADARG;
  const int64_t n = (int64_t)B * C * H2 * W2;
  hipLaunchKernelGGL(pack_latents_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)nchw, (bf16_t*)packed, B, C, H2, W2);
  return rt_hip_status();
}

int rt_unpack_latents(const void* packed, void* nhwc, int32_t B, int32_t C, int32_t H2, int32_t W2, float inv_scale,
                      float shift, void* stream) {
  if (!packed || !nhwc || B < 1 || C < 1 || H2 < 2 || W2 < 2 || H2 % 2 || W2 % 2) return RT_E_BADARG;
  const int64_t n = (int64_t)B * C * H2 * W2;
  hipLaunchKernelGGL(unpack_latents_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)packed, (bf16_t*)nhwc, B, C, H2, W2, inv_scale, shift);
  return rt_hip_status();
}

int rt_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream) {
  if (!x || !y || n < 1) return RT_E_BADARG;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)y, n);
  return rt_hip_status();
}
int rt_cast_bf16_to_f32(const void* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n < 1) return RT_E_BADARG;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, y, n);
  return rt_hip_status();
}

int rt_silu_split_bf16(const float* x, void* hi, void* lo, int64_t n, int32_t apply_silu, void* stream) {
  if (!x || !hi || !lo || n < 1) return RT_E_BADARG;
  hipLaunchKernelGGL(silu_split_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)hi, (bf16_t*)lo, n,
                     apply_silu);
  return rt_hip_status();
}

int rt_masked_accumulate(const void* x, void* y, const float* rowscale, float alpha, int32_t batch, int32_t rows,
                         int32_t D, int32_t accumulate, int32_t y_f32, void* stream) {
  if (!x || !y || batch < 1 || rows < 1 || D < 8) return RT_E_BADARG;
  if (D % 8) return RT_E_SHAPE;
  if (!RT_ALIGNED(x, 16) || !RT_ALIGNED(y, 16)) return RT_E_ALIGN;
  const int64_t n = (int64_t)batch * rows * (D / 8);
  if (y_f32) {
    hipLaunchKernelGGL(masked_accumulate_f32_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (float*)y, rowscale, alpha, batch, rows, D / 8, accumulate);
    return rt_hip_status();
  }
  hipLaunchKernelGGL(masked_accumulate_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)y, rowscale, alpha, batch, rows, D / 8, accumulate);
  return rt_hip_status();
}

}  // extern "C"
