// Small kernels of the prompt encoders (PIPE:232-347: T5-XXL encoder -> prompt_embeds, CLIP-L text model
// -> pooled_prompt_embeds). They run once per prompt, outside the denoising loop; the matrix work goes through rt_gemm_bf16,
// the attention (head dim 64, with T5's relative-position bias or CLIP's causal mask) is assembled per head from
// rt_gemm_bf16 + rt_softmax_rows_bias + rt_transpose_bf16 + rt_gemm_bf16, as the VAE's mid-block attention is.
#include "rt_common.h"

namespace {

// out[i][:] = table[ids[i]][:]  (bf16 rows, 16 bytes per lane)
__global__ __launch_bounds__(256) void embedding_gather_kernel(const bf16_t* __restrict__ table, int64_t ld, const int32_t* __restrict__ ids,
                                                               bf16_t* __restrict__ out, int64_t ldo, int n, int D, int vocab) {
  const int c8 = D / 8;
  const int64_t total = (int64_t)n * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(i / c8), c = (int)(i - (int64_t)row * c8);
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    *reinterpret_cast<u32x4*>(out + (int64_t)row * ldo + c * 8) = *reinterpret_cast<const u32x4*>(table + (int64_t)id * ld + c * 8);
  }
}

// T5LayerNorm: y = x * rsqrt(mean(x^2) + eps) * w  (no mean subtraction, no bias). One wave per row; x bf16 or f32.
template <bool X_F32>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const void* __restrict__ x, int64_t ldx, const bf16_t* __restrict__ w,
                                                           bf16_t* __restrict__ out, int64_t ldo, int rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  auto load8 = [&](int col, float (&y)[8]) {
    if (X_F32) {
      const float* p = reinterpret_cast<const float*>(x) + (int64_t)row * ldx + col;
      const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) { y[i] = a[i]; y[4 + i] = b[i]; }
    } else {
      const u32x4 u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(x) + (int64_t)row * ldx + col);
#pragma unroll
      for (int i = 0; i < 4; ++i) { y[2 * i] = bf16lo(u[i]); y[2 * i + 1] = bf16hi(u[i]); }
    }
  };
  float ss = 0.f;
  for (int col = lane * 8; col < D; col += 512) {
    float y[8];
    load8(col, y);
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += y[i] * y[i];
  }
  const float r = rsqrtf(wave_sum(ss) / (float)D + eps);
  for (int col = lane * 8; col < D; col += 512) {
    float y[8];
    load8(col, y);
    const u32x4 wu = *reinterpret_cast<const u32x4*>(w + col);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = pack_bf16x2(y[2 * i] * r * bf16lo(wu[i]), y[2 * i + 1] * r * bf16hi(wu[i]));
    *reinterpret_cast<u32x4*>(out + (int64_t)row * ldo + col) = o;
  }
}

// p = softmax(scale * s + bias) over `cols`; s f32 [rows][lds], bias f32 [rows][ldb] (may hold -inf), p bf16 [rows][ldp];
// columns cols..cols_out-1 of p are written as zeros (K padding of the following P·V GEMM). One workgroup per row.
__global__ __launch_bounds__(256) void softmax_rows_bias_kernel(const float* __restrict__ s, int64_t lds_, const float* __restrict__ bias,
                                                                int64_t ldb, bf16_t* __restrict__ p, int64_t ldp, int cols,
                                                                int cols_out, float scale) {
  __shared__ float red[8];
  const float* row = s + (int64_t)blockIdx.x * lds_;
  const float* brow = bias ? bias + (int64_t)blockIdx.x * ldb : nullptr;
  bf16_t* out = p + (int64_t)blockIdx.x * ldp;
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < cols; i += blockDim.x) mx = fmaxf(mx, row[i] * scale + (brow ? brow[i] : 0.f));
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  if (mx == -INFINITY) mx = 0.f;                     // fully masked row: all-zero probabilities instead of NaN
  float sum = 0.f;
  for (int i = threadIdx.x; i < cols; i += blockDim.x) sum += exp2f((row[i] * scale + (brow ? brow[i] : 0.f) - mx) * 1.4426950408889634f);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sum;
  __syncthreads();
  const float tot = red[4] + red[5] + red[6] + red[7];
  const float inv = tot > 0.f ? 1.0f / tot : 0.f;
  for (int i = threadIdx.x; i < cols_out; i += blockDim.x) {
    float v = 0.f;
    if (i < cols) v = exp2f((row[i] * scale + (brow ? brow[i] : 0.f) - mx) * 1.4426950408889634f) * inv;
    out[i] = (bf16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
  }
}

// gated activation of T5 v1.1: out[r][c] = x[r][c] * x[r][F + c]  (the GEMM epilogue has already applied GELU to one half)
__global__ __launch_bounds__(256) void gated_mul_kernel(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ out, int64_t ldo,
                                                        int rows, int F) {
  const int c8 = F / 8;
  const int64_t total = (int64_t)rows * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / c8), c = (int)(i - (int64_t)r * c8);
    const u32x4 a = *reinterpret_cast<const u32x4*>(x + (int64_t)r * ldx + c * 8);
    const u32x4 b = *reinterpret_cast<const u32x4*>(x + (int64_t)r * ldx + F + c * 8);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf16x2(bf16lo(a[j]) * bf16lo(b[j]), bf16hi(a[j]) * bf16hi(b[j]));
    *reinterpret_cast<u32x4*>(out + (int64_t)r * ldo + c * 8) = o;
  }
}

// CLIP's quick_gelu, in place: x * sigmoid(1.702 x)
__global__ __launch_bounds__(256) void quick_gelu_kernel(bf16_t* __restrict__ x, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    u32x4 u = *reinterpret_cast<u32x4*>(x + i * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = bf16lo(u[j]), b = bf16hi(u[j]);
      u[j] = pack_bf16x2(a / (1.f + __expf(-1.702f * a)), b / (1.f + __expf(-1.702f * b)));
    }
    *reinterpret_cast<u32x4*>(x + i * 8) = u;
  }
}

static inline unsigned grid_for(int64_t n, int block) {
  int64_t g = (n + block - 1) / block;
  return (unsigned)(g > 65536 ? 65536 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int rt_embedding_gather(const void* table, int64_t ld, const int32_t* ids, void* out, int64_t ldo, int32_t n, int32_t D,
                                   int32_t vocab, void* stream) {
  if (!table || !ids || !out || n < 1 || D < 8 || vocab < 1) return RT_E_BADARG;
  if (D % 8 || ld % 8 || ldo % 8 || ld < D || ldo < D) return RT_E_SHAPE;
  if (!RT_ALIGNED(table, 16) || !RT_ALIGNED(out, 16)) return RT_E_ALIGN;
  hipLaunchKernelGGL(embedding_gather_kernel, dim3(grid_for((int64_t)n * (D / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)table, ld, ids, (bf16_t*)out, ldo, n, D, vocab);
  return rt_hip_status();
}

extern "C" int rt_rmsnorm_rows(const void* x, int64_t ldx, int32_t x_f32, const void* w, void* out, int64_t ldo, int32_t rows, int32_t D,
                               float eps, void* stream) {
  if (!x || !w || !out || rows < 1 || D < 8) return RT_E_BADARG;
  if (D % 8 || ldx % 8 || ldo % 8) return RT_E_SHAPE;
  if (!RT_ALIGNED(x, 16) || !RT_ALIGNED(w, 16) || !RT_ALIGNED(out, 16)) return RT_E_ALIGN;
  const dim3 grid((rows + 3) / 4), block(256);
  if (x_f32) hipLaunchKernelGGL(rmsnorm_rows_kernel<true>, grid, block, 0, (hipStream_t)stream, x, ldx, (const bf16_t*)w, (bf16_t*)out, ldo, rows, D, eps);
  else hipLaunchKernelGGL(rmsnorm_rows_kernel<false>, grid, block, 0, (hipStream_t)stream, x, ldx, (const bf16_t*)w, (bf16_t*)out, ldo, rows, D, eps);
  return rt_hip_status();
}

extern "C" int rt_softmax_rows_bias(const float* s, int64_t lds_, const float* bias, int64_t ldb, void* p, int64_t ldp, int32_t rows,
                                    int32_t cols, int32_t cols_out, float scale, void* stream) {
  if (!s || !p || rows < 1 || cols < 1 || cols_out < cols) return RT_E_BADARG;
  if (lds_ < cols || ldp < cols_out || (bias && ldb < cols)) return RT_E_SHAPE;
  hipLaunchKernelGGL(softmax_rows_bias_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, s, lds_, bias, ldb, (bf16_t*)p, ldp, cols,
                     cols_out, scale);
  return rt_hip_status();
}

extern "C" int rt_gated_mul(const void* x, int64_t ldx, void* out, int64_t ldo, int32_t rows, int32_t F, void* stream) {
  if (!x || !out || rows < 1 || F < 8) return RT_E_BADARG;
  if (F % 8 || ldx % 8 || ldo % 8 || ldx < 2 * (int64_t)F || ldo < F) return RT_E_SHAPE;
  if (!RT_ALIGNED(x, 16) || !RT_ALIGNED(out, 16)) return RT_E_ALIGN;
  hipLaunchKernelGGL(gated_mul_kernel, dim3(grid_for((int64_t)rows * (F / 8), 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, ldx,
                     (bf16_t*)out, ldo, rows, F);
  return rt_hip_status();
}

extern "C" int rt_quick_gelu(void* x, int64_t n, void* stream) {
  if (!x || n < 8) return RT_E_BADARG;
  if (n % 8) return RT_E_SHAPE;
  if (!RT_ALIGNED(x, 16)) return RT_E_ALIGN;
  hipLaunchKernelGGL(quick_gelu_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, n / 8);
  return rt_hip_status();
}
