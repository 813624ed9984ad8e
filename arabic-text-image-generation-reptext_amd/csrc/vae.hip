// AutoencoderKL kernels (FLUX VAE): implicit-GEMM convolution on MFMA, GroupNorm(+SiLU), row softmax and
// transpose for the mid-block attention, image post-processing. Math: SURVEY.md Appendix A.7 (diffusers
// AutoencoderKL, not present under /root/reference); call sites PIPE:467,705,711 (encode) and PIPE:1139 (decode).
//
// Activation layout in HBM: zero-haloed NHWC bf16, [B][H+2][W+2][C]. The 1-pixel halo is allocated zeroed and
// never written, so the 3×3 gather needs no bounds checks and LDS-DMA can fetch every tap directly:
//   stride 1 : source (y+dy-1, x+dx-1)            -> halo index +1
//   nearest-2x upsample fused: source floor((y+dy-1)/2)  (so Up = gather, no 4x tensor is ever written)
//   stride 2, pad (0,1,0,1) (encoder Downsample)  : source (2y+dy, 2x+dx)
// The GEMM is M = B·Ho·Wo output pixels, N = Cout, K = taps·Cin with the same 256×256×64 tiling, LDS image and
// swizzle as gemm_bf16.hip; a K-tile is 64 channels of one tap, so its A rows are 128-B runs of the input.
#include "rt_common.h"
#include <initializer_list>
#include <stdlib.h>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int THREADS = 512;
constexpr int TILE_BYTES = BM * BK * 2;
constexpr int BUF_BYTES = 2 * TILE_BYTES;
constexpr int LDS_BYTES = 2 * BUF_BYTES;

struct ConvArgs {
  const bf16_t* x;
  const bf16_t* w;
  const bf16_t* bias;
  const bf16_t* res;
  void* y;
  int B, Hs, Ws, Ho, Wo, Cin, Cout, ks, stride, ups, out_f32;
};

// WNC = wave columns of the 8-wave workgroup: 4 -> 256x256 tile, waves 2(M) x 4(N), 128x64 each (the GEMM's arrangement);
// 2 -> 256x128 tile, waves 4(M) x 2(N), 64x64 each — for layers with Cout <= 128 (the decoder's 1024x1024 stages), where the wide
// tile left half of the waves without a single live output column.
template <int WNC>
__global__ __launch_bounds__(THREADS, 2) void conv_nhwc_kernel(const ConvArgs a) {
  constexpr int BN = 64 * WNC;                       // shadows the file-level 256
  constexpr int MI = BM / (8 / WNC) / 16;            // 16-row fragments per wave: 8 or 4
  constexpr int WP = BN / 64;                        // weight pieces (8 rows each) per wave: 4 or 2
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_m = (a.B * a.Ho * a.Wo + BM - 1) / BM;
  const int tn = blockIdx.x / tiles_m;
  const int tm = blockIdx.x - tn * tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;
  const int M = a.B * a.Ho * a.Wo;
  const int K = a.ks * a.ks * a.Cin;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WNC, wn = wave % WNC;
  const int pad = a.ks >> 1;
  const int Wp = a.Ws + 2;

  // staging: this lane's 4 output pixels (one per piece) and weight rows
  int pb[4], py[4], px[4], plc[4];
  const bf16_t* srcW[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = wave * 32 + p * 8 + (lane >> 3);
    plc[p] = ((lane & 7) ^ ((row >> 1) & 7)) * 8;
    const int m = min(m0 + row, M - 1);
    const int b = m / (a.Ho * a.Wo);
    const int r = m - b * a.Ho * a.Wo;
    pb[p] = b; py[p] = r / a.Wo; px[p] = r - py[p] * a.Wo;
    // weight rows: WP pieces of 8 rows per wave (8 waves x WP x 8 = BN rows), same chunk permutation (row & 15 pattern repeats)
    const int wrow = wave * (8 * WP) + (p % WP) * 8 + (lane >> 3);
    const int wr = min(n0 + wrow, a.Cout - 1);
    srcW[p] = a.w + (int64_t)wr * K + (((lane & 7) ^ ((wrow >> 1) & 7)) * 8);
  }
  const int stage_off = wave * 32 * 128;
  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * BUF_BYTES + stage_off;
    const int k0 = kt * BK;
    const int tap = k0 / a.Cin;                 // wave-uniform
    const int c0 = k0 - tap * a.Cin;
    const int dy = tap / a.ks, dx = tap - dy * a.ks;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      int sy, sx;
      if (a.stride == 2) { sy = 2 * py[p] + dy + 1; sx = 2 * px[p] + dx + 1; }
      else { sy = ((py[p] + dy - pad) >> a.ups) + 1; sx = ((px[p] + dx - pad) >> a.ups) + 1; }
      const bf16_t* src = a.x + (((int64_t)pb[p] * (a.Hs + 2) + sy) * Wp + sx) * a.Cin + c0 + plc[p];
      __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(base + p * 1024), 16, 0, 0);
      if (p < WP) __builtin_amdgcn_global_load_lds(GLB_PTR(srcW[p] + k0), LDS_PTR(smem + buf * BUF_BYTES + TILE_BYTES + (wave * WP + p) * 1024), 16, 0, 0);
    }
  };

  const int l15 = lane & 15;
  const int sw = (lane >> 1) & 7;
  const int rd0 = l15 * 128 + (((0 + (lane >> 4)) ^ sw) << 4);
  const int rd1 = l15 * 128 + (((4 + (lane >> 4)) ^ sw) << 4);
  const int a_base = wm * (MI * 16) * 128;
  const int w_base = TILE_BYTES + wn * 64 * 128;

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // waves whose 64 output columns are all >= Cout still stage (they share the tile) but skip the MFMAs
  const bool wave_live = (n0 + wn * 64) < a.Cout;
  const int nk = K / BK;
  stage(0, 0);
  rt_dma_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* tb = smem + cur * BUF_BYTES;
    if (wave_live) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int rd = kk ? rd1 : rd0;
        bf16x8 wf[4], af[MI];
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(tb + w_base + j * 2048 + rd);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8*>(tb + a_base + i * 2048 + rd);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
    rt_dma_barrier();     // next tile landed in every wave's rows; this tile's reads are done
  }

  if (!wave_live) return;
  const int mrow = m0 + wm * (MI * 16) + l15;
  const int ncol = n0 + wn * 64 + 4 * (lane >> 4);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = mrow + i * 16;
    if (m >= M) continue;
    const int b = m / (a.Ho * a.Wo);
    const int r = m - b * a.Ho * a.Wo;
    const int oy = r / a.Wo, ox = r - oy * a.Wo;
    const int64_t pix = (((int64_t)b * (a.Ho + 2) + oy + 1) * (a.Wo + 2) + ox + 1) * a.Cout;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = ncol + j * 16;
      if (n >= a.Cout) continue;
      f32x4 v = acc[i][j];
      if (a.bias) {
        const u32x2 bb = *reinterpret_cast<const u32x2*>(a.bias + n);
        v[0] += bf16lo(bb[0]); v[1] += bf16hi(bb[0]); v[2] += bf16lo(bb[1]); v[3] += bf16hi(bb[1]);
      }
      if (a.res) {
        const u32x2 rr = *reinterpret_cast<const u32x2*>(a.res + pix + n);
        v[0] += bf16lo(rr[0]); v[1] += bf16hi(rr[0]); v[2] += bf16lo(rr[1]); v[3] += bf16hi(rr[1]);
      }
      if (a.out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + pix + n) = v;
      } else {
        u32x2 o;
        o[0] = pack_bf16x2(v[0], v[1]);
        o[1] = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(a.y) + pix + n) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm over haloed NHWC, bitwise reproducible (no atomics: every sum has a fixed order).
// Pass 1: per-thread fp32 partials per channel -> per-workgroup per-(group,stat) fp64 partial, summed in thread order.
// Pass 2: per-(b,group,stat) sum of the workgroup partials in workgroup order. Pass 3: normalise, affine, optional SiLU,
// write the interior of a haloed buffer.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_kernel(const bf16_t* __restrict__ x, double* __restrict__ part, int B,
                                                        int H, int W, int C, int G, int pix_per_block) {
  // thread t handles channel chunk (t % (C/8)) of pixels t / (C/8) + k*(256/(C/8))
  __shared__ float lane_part[256][17];                     // [thread][8 sums | 8 sums of squares], +1 pad
  const int c8 = C / 8;
  const int b = blockIdx.y;
  const int cpg = C / G;                                  // channels per group
  const int p0 = blockIdx.x * pix_per_block;
  const int pend = min(p0 + pix_per_block, H * W);
  const int chunk = threadIdx.x % c8;
  const int pstep = blockDim.x / c8;
  float se[8], qe[8];                                      // one partial per element of this thread's 8-channel chunk
#pragma unroll
  for (int i = 0; i < 8; ++i) { se[i] = 0.f; qe[i] = 0.f; }
  // four pixels per trip: the four 16-byte loads are in flight together (same per-thread summation order as one at a time)
  int p = p0 + threadIdx.x / c8;
  auto ld = [&](int pp) {
    const int yy = pp / W, xx = pp - yy * W;
    return *reinterpret_cast<const u32x4*>(x + (((int64_t)b * (H + 2) + yy + 1) * (W + 2) + xx + 1) * C + chunk * 8);
  };
  auto acc = [&](const u32x4 u) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a0 = bf16lo(u[i]), a1 = bf16hi(u[i]);
      se[2 * i] += a0; qe[2 * i] += a0 * a0;
      se[2 * i + 1] += a1; qe[2 * i + 1] += a1 * a1;
    }
  };
  for (; p + 3 * pstep < pend; p += 4 * pstep) {
    const u32x4 u0 = ld(p), u1 = ld(p + pstep), u2 = ld(p + 2 * pstep), u3 = ld(p + 3 * pstep);
    acc(u0); acc(u1); acc(u2); acc(u3);
  }
  for (; p < pend; p += pstep) acc(ld(p));
#pragma unroll
  for (int i = 0; i < 8; ++i) { lane_part[threadIdx.x][i] = se[i]; lane_part[threadIdx.x][8 + i] = qe[i]; }
  __syncthreads();
  // thread j < 2G owns (group j/2, stat j%2): channels [g*cpg, (g+1)*cpg) over all pixel slots, in a fixed order
  for (int j = threadIdx.x; j < 2 * G; j += blockDim.x) {
    const int g = j >> 1, stat = j & 1;
    double acc = 0.0;
    for (int ch = g * cpg; ch < (g + 1) * cpg; ++ch) {
      const int ck = ch >> 3, el = (ch & 7) + 8 * stat;
      for (int k = 0; k < pstep; ++k) acc += (double)lane_part[ck + k * c8][el];
    }
    part[((int64_t)b * gridDim.x + blockIdx.x) * 2 * G + j] = acc;
  }
}

// One workgroup of 1024 threads per batch entry: thread (seg, j) sums the partials k = seg, seg + nseg, ... of statistic j (independent
// loads, up to nblk / nseg each), the nseg segment sums of a statistic are then added in segment order — a fixed order, run to run.
// (The first version walked all nblk <= 1024 partials of a statistic in one thread: 75-250 us per call for a few KiB of data.)
__global__ __launch_bounds__(1024) void gn_reduce_kernel(const double* __restrict__ part, int nblk, int G2, float* __restrict__ mr,
                                                          double cnt, float eps) {
  __shared__ double seg_sum[1024];
  __shared__ double tot_s[1024];
  const int b = blockIdx.x;
  const int nseg = blockDim.x / G2;              // host launches with G2 <= 1024 dividing... any G2 <= blockDim.x works (tail threads idle)
  const int j = threadIdx.x % G2, seg = threadIdx.x / G2;
  double acc = 0.0;
  if (seg < nseg)
    for (int k = seg; k < nblk; k += nseg) acc += part[((int64_t)b * nblk + k) * G2 + j];
  seg_sum[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < G2) {
    double tot = 0.0;
    for (int sgm = 0; sgm < nseg; ++sgm) tot += seg_sum[sgm * G2 + threadIdx.x];
    tot_s[threadIdx.x] = tot;
  }
  __syncthreads();
  // mean and 1/std of every group, once (gn_apply used to redo this fp64 arithmetic for every element)
  if (threadIdx.x < G2 / 2) {
    const double sm = tot_s[2 * threadIdx.x], sq = tot_s[2 * threadIdx.x + 1];
    const double mean = sm / cnt;
    const double var = sq / cnt - mean * mean;
    mr[((int64_t)b * (G2 / 2) + threadIdx.x) * 2] = (float)mean;
    mr[((int64_t)b * (G2 / 2) + threadIdx.x) * 2 + 1] = rsqrtf((float)(var > 0.0 ? var : 0.0) + eps);
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                        const float* __restrict__ mr, const bf16_t* __restrict__ gamma,
                                                        const bf16_t* __restrict__ beta, int B, int H, int W, int C, int G,
                                                        int silu) {
  // grid: x = 256-thread slices of one image row (W * C/8 chunks), y = (batch, row): one 32-bit division per thread instead of the
  // three 64-bit ones of a flat index, and the row/batch indices are scalar
  const int c8 = C / 8;
  const int cpg = C / G;
  const int yy = (int)blockIdx.y % H, b = (int)blockIdx.y / H;
  {
    const int ir = (int)blockIdx.x * 256 + (int)threadIdx.x;
    if (ir >= W * c8) return;
    const int xx = ir / c8, chunk = ir - xx * c8;
    const int64_t off = (((int64_t)b * (H + 2) + yy + 1) * (W + 2) + xx + 1) * C + chunk * 8;
    const u32x4 u = *reinterpret_cast<const u32x4*>(x + off);
    const u32x4 gu = *reinterpret_cast<const u32x4*>(gamma + chunk * 8);
    const u32x4 bu = *reinterpret_cast<const u32x4*>(beta + chunk * 8);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[2] = {bf16lo(u[j]), bf16hi(u[j])};
      const float gm[2] = {bf16lo(gu[j]), bf16hi(gu[j])};
      const float bt[2] = {bf16lo(bu[j]), bf16hi(bu[j])};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int g = (chunk * 8 + 2 * j + e) / cpg;
        const float mean = mr[((int64_t)b * G + g) * 2], rstd = mr[((int64_t)b * G + g) * 2 + 1];
        float t = (v[e] - mean) * rstd * gm[e] + bt[e];
        if (silu) t = silu_fast_f(t);
        v[e] = t;
      }
      o[j] = pack_bf16x2(v[0], v[1]);
    }
    *reinterpret_cast<u32x4*>(y + off) = o;
  }
}

// row softmax: p = softmax(scale * s) over `cols`, f32 in -> bf16 out. One workgroup per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16_t* __restrict__ p, int cols,
                                                            float scale) {
  __shared__ float red[8];
  const float* row = s + (int64_t)blockIdx.x * cols;
  bf16_t* out = p + (int64_t)blockIdx.x * cols;
  float mx = -INFINITY;
  for (int i = threadIdx.x * 4; i < cols; i += blockDim.x * 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  const float c = scale * 1.4426950408889634f;
  for (int i = threadIdx.x * 4; i < cols; i += blockDim.x * 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += exp2f((v[j] - mx) * c);
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int i = threadIdx.x * 4; i < cols; i += blockDim.x * 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
    u32x2 o;
    o[0] = pack_bf16x2(exp2f((v[0] - mx) * c) * inv, exp2f((v[1] - mx) * c) * inv);
    o[1] = pack_bf16x2(exp2f((v[2] - mx) * c) * inv, exp2f((v[3] - mx) * c) * inv);
    *reinterpret_cast<u32x2*>(out + i) = o;
  }
}

// out[c][r] = in[r][c] through a 64×64 LDS tile (bf16)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                              int R, int C, int64_t ld_in, int64_t ld_out) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? in[(int64_t)(r0 + r) * ld_in + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < C && r0 + r < R) out[(int64_t)(c0 + c) * ld_out + r0 + r] = tile[r][c];
  }
}

// decoder tail: haloed NHWC f32 [B][H+2][W+2][Cp] -> NCHW f32 [B][C][H][W] and uint8 HWC = round(clamp(x/2+0.5,0,1)*255)
__global__ void image_out_kernel(const float* __restrict__ x, float* __restrict__ nchw, uint8_t* __restrict__ u8, int B,
                                 int H, int W, int Cp, int C) {
  const int64_t n = (int64_t)B * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = i;
    const int xx = (int)(p % W); p /= W;
    const int yy = (int)(p % H);
    const int b = (int)(p / H);
    const float* src = x + (((int64_t)b * (H + 2) + yy + 1) * (W + 2) + xx + 1) * Cp;
    for (int c = 0; c < C; ++c) {
      const float v = src[c];
      if (nchw) nchw[(((int64_t)b * C + c) * H + yy) * W + xx] = v;
      if (u8) {
        const float t = fminf(fmaxf(v * 0.5f + 0.5f, 0.f), 1.f) * 255.f;
        u8[i * C + c] = (uint8_t)rintf(t);
      }
    }
  }
}

// image pre-processing + halo: NCHW f32 [B][C][H][W] -> haloed NHWC bf16 [B][H+2][W+2][Cp] (channels >= C zero)
__global__ void nchw_to_haloed_nhwc_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int B, int C, int H, int W,
                                           int Cp) {
  const int64_t n = (int64_t)B * H * W * Cp;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = i;
    const int c = (int)(p % Cp); p /= Cp;
    const int xx = (int)(p % W); p /= W;
    const int yy = (int)(p % H);
    const int b = (int)(p / H);
    const float v = c < C ? x[(((int64_t)b * C + c) * H + yy) * W + xx] : 0.f;
    y[(((int64_t)b * (H + 2) + yy + 1) * (W + 2) + xx + 1) * Cp + c] = f32_to_bf16(v);
  }
}

// haloed NHWC bf16 -> NCHW f32 (encoder moments / generic readout)
__global__ void haloed_nhwc_to_nchw_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int B, int C, int H, int W,
                                           int Cp) {
  const int64_t n = (int64_t)B * C * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t p = i;
    const int xx = (int)(p % W); p /= W;
    const int yy = (int)(p % H); p /= H;
    const int c = (int)(p % C);
    const int b = (int)(p / C);
    y[i] = bf16_to_f32(x[(((int64_t)b * (H + 2) + yy + 1) * (W + 2) + xx + 1) * Cp + c]);
  }
}

// packed latents [B][(H2/2)(W2/2)][4C] bf16 -> haloed NHWC bf16 [B][H2+2][W2+2][Cp], value/scaling + shift (PIPE:1136-1137)
__global__ void unpack_to_haloed_kernel(const bf16_t* __restrict__ p, bf16_t* __restrict__ y, int B, int C, int H2, int W2,
                                        int Cp, float inv_scale, float shift) {
  const int64_t n = (int64_t)B * H2 * W2 * Cp;
  const int w = W2 / 2, h = H2 / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t t = i;
    const int c = (int)(t % Cp); t /= Cp;
    const int xx = (int)(t % W2); t /= W2;
    const int yy = (int)(t % H2);
    const int b = (int)(t / H2);
    float v = 0.f;
    if (c < C) {
      const int tok = (yy >> 1) * w + (xx >> 1);
      const int ch = c * 4 + (yy & 1) * 2 + (xx & 1);
      v = bf16_to_f32(p[((int64_t)b * h * w + tok) * (C * 4) + ch]) * inv_scale + shift;
    }
    y[(((int64_t)b * (H2 + 2) + yy + 1) * (W2 + 2) + xx + 1) * Cp + c] = f32_to_bf16(v);
  }
}

inline int grid_for(int64_t n, int block) {
  int64_t g = (n + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace

// ---- torch.nn.functional.interpolate on [planes][H][W] fp32 (or uint8 * in_scale) -> fp32, modes "nearest" and "bilinear"
// with align_corners=False, antialias=False: the pipelines' regional-mask /16 (PIPE:1010-1012), glyph-mask (PIPE:645-650) and
// inpaint-mask (INP:813) resizes. Same source-index rule and the same fp32 expression order as ATen's upsample kernels:
//   ratio = scales given ? 1/scale : in/out;  nearest: src = min(floor(dst * ratio), in-1)
//   bilinear: real = max(ratio*(dst+0.5)-0.5, 0); i0 = (int)real; i1 = i0 + (i0 < in-1); l1 = real - i0; l0 = 1 - l1
//             out = fma(h1*w1, p11, fma(h1*w0, p10, fma(h0*w1, p01, (h0*w0)*p00)))
// The four tap weights are formed first and the taps accumulated in that order — the order ATen's CPU kernel (torch 2.10)
// was found to use: bit-identical for the power-of-two ratios the pipelines have (16, 8; every product is then exact),
// within one fp32 ulp otherwise (ATen's own result there depends on its vector ISA path). Roundings are explicit (__f*_rn).
__global__ void resize2d_kernel(const void* __restrict__ in, int in_u8, float in_scale, float* __restrict__ out, int planes, int H, int W,
                                int OH, int OW, float rh, float rw, int bilinear) {
#pragma clang fp contract(off)          // every rounding step below is the one ATen takes; no silent mul+add fusion
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)planes * OH * OW) return;
  const int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH), p = (int)(idx / ((int64_t)OW * OH));
  auto px = [&](int y, int x) -> float {
    const int64_t o = ((int64_t)p * H + y) * W + x;
    return in_u8 ? (float)reinterpret_cast<const uint8_t*>(in)[o] / in_scale : reinterpret_cast<const float*>(in)[o];
  };
  if (!bilinear) {
    const int sy = min((int)floorf((float)oy * rh), H - 1), sx = min((int)floorf((float)ox * rw), W - 1);
    out[idx] = px(sy, sx);
    return;
  }
  const float ry = fmaxf(__fsub_rn(__fmul_rn(rh, (float)oy + 0.5f), 0.5f), 0.f), rx = fmaxf(__fsub_rn(__fmul_rn(rw, (float)ox + 0.5f), 0.5f), 0.f);
  const int y0 = (int)ry, x0 = (int)rx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float h1 = fminf(fmaxf(ry - (float)y0, 0.f), 1.f), w1 = fminf(fmaxf(rx - (float)x0, 0.f), 1.f);
  const float h0 = 1.f - h1, w0 = 1.f - w1;
  float acc = __fmul_rn(__fmul_rn(h0, w0), px(y0, x0));
  acc = __fmaf_rn(__fmul_rn(h0, w1), px(y0, x1), acc);
  acc = __fmaf_rn(__fmul_rn(h1, w0), px(y1, x0), acc);
  out[idx] = __fmaf_rn(__fmul_rn(h1, w1), px(y1, x1), acc);
}

// 0.10 * glyph latent + noise where the bilinearly down-sampled glyph mask is > 0, else noise (PIPE:645-654 / INP:640-647):
// mask = any channel of the [-1,1]-normalised glyph image > 0, resized to the latent grid as above, thresholded at > 0.
__global__ void glyph_blend_kernel(const float* __restrict__ img, const float* __restrict__ lat, const float* __restrict__ noise,
                                   float* __restrict__ out, int B, int Cimg, int H, int W, int Cl, int OH, int OW, float rh, float rw) {
#pragma clang fp contract(off)          // torch computes 0.10 * lat and the sum as two roundings
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * OH * OW) return;
  const int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH), b = (int)(idx / ((int64_t)OW * OH));
  auto m = [&](int y, int x) -> float {
    bool any = false;
    for (int c = 0; c < Cimg; ++c) any = any || img[(((int64_t)b * Cimg + c) * H + y) * W + x] > 0.f;
    return any ? 1.f : 0.f;
  };
  const float ry = fmaxf(__fsub_rn(__fmul_rn(rh, (float)oy + 0.5f), 0.5f), 0.f), rx = fmaxf(__fsub_rn(__fmul_rn(rw, (float)ox + 0.5f), 0.5f), 0.f);
  const int y0 = (int)ry, x0 = (int)rx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float h1 = fminf(fmaxf(ry - (float)y0, 0.f), 1.f), w1 = fminf(fmaxf(rx - (float)x0, 0.f), 1.f);
  const float h0 = 1.f - h1, w0 = 1.f - w1;
  float acc = __fmul_rn(__fmul_rn(h0, w0), m(y0, x0));                       // the mask is 0/1 and the weights are >= 0: any order gives
  acc = __fmaf_rn(__fmul_rn(h0, w1), m(y0, x1), acc);                        // the same sign, which is all that is used
  acc = __fmaf_rn(__fmul_rn(h1, w0), m(y1, x0), acc);
  const bool on = __fmaf_rn(__fmul_rn(h1, w1), m(y1, x1), acc) > 0.f;
  for (int c = 0; c < Cl; ++c) {
    const int64_t o = (((int64_t)b * Cl + c) * OH + oy) * OW + ox;
    const float scaled = 0.10f * lat[o];        // plain operators: the pragma above applies to them (not to header intrinsics)
    out[o] = on ? scaled + noise[o] : noise[o];
  }
}

namespace {
int g_conv_variant = -1;        // 1: stride-1 convolutions on the GEMM's convolution form (default), 0: conv_nhwc_kernel for everything
int conv_variant_now() {
  if (g_conv_variant < 0) {
    const char* e = getenv("RT_CONV_GEMM");
    g_conv_variant = e ? (atoi(e) != 0) : 1;
  }
  return g_conv_variant;
}
}  // namespace

extern "C" {

int rt_conv2d_variant(int32_t mode) {
  const int prev = conv_variant_now();
  if (mode >= 0) g_conv_variant = mode != 0;
  return prev;
}

int rt_conv2d_nhwc(const void* x, const void* w, const void* bias, const void* res, void* y, int32_t B, int32_t Hs,
                   int32_t Ws, int32_t Cin, int32_t Cout, int32_t ksize, int32_t stride, int32_t upsample2x,
                   int32_t out_f32, void* stream) {
  if (!x || !w || !y || B < 1 || Hs < 1 || Ws < 1 || Cin < 1 || Cout < 1) return RT_E_BADARG;
  if ((ksize != 1 && ksize != 3) || (stride != 1 && stride != 2) || (stride == 2 && (upsample2x || ksize != 3)))
    return RT_E_SHAPE;
  if (Cin % 64 || Cout % 4) return RT_E_SHAPE;
  if (stride == 2 && (Hs % 2 || Ws % 2)) return RT_E_SHAPE;
  if (!RT_ALIGNED(x, 16) || !RT_ALIGNED(w, 16) || !RT_ALIGNED(y, 16)) return RT_E_ALIGN;
  ConvArgs a;
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = (const bf16_t*)bias; a.res = (const bf16_t*)res; a.y = y;
  a.B = B; a.Hs = Hs; a.Ws = Ws; a.Cin = Cin; a.Cout = Cout; a.ks = ksize; a.stride = stride; a.ups = upsample2x ? 1 : 0;
  a.Ho = stride == 2 ? Hs / 2 : (upsample2x ? 2 * Hs : Hs);
  a.Wo = stride == 2 ? Ws / 2 : (upsample2x ? 2 * Ws : Ws);
  a.out_f32 = out_f32;
  // stride 1, no fused upsample, a real channel count, bf16 out: the ping-pong GEMM in its convolution form (gemm_bf16.hip, CONV) —
  // same K order per output element, so the same bits as conv_nhwc_kernel below (tests/test_vae_real_gpu.py), at GEMM speed.
  if (conv_variant_now() && stride == 1 && !upsample2x && !out_f32 && Cout >= 64 && (int64_t)B * (Hs + 2) * (Ws + 2) < (1 << 24)) {
    rt_gemm_group g{};
    g.A = x; g.W = w; g.C = y; g.bias = bias; g.res = res;
    g.M = B * (Hs + 2) * (Ws + 2); g.N = Cout; g.K = ksize * ksize * Cin; g.batch = 1;
    g.lda = Cin; g.ldw = g.K; g.ldc = Cout; g.ldr = Cout;
    g.gelu_from = Cout; g.alpha = 1.f;
    g.conv_ks = ksize; g.conv_cin = Cin; g.conv_w2 = Ws + 2; g.conv_h2 = Hs + 2;
    return rt_gemm_bf16(&g, 1, stream);
  }
  const int64_t M = (int64_t)B * a.Ho * a.Wo;
  if (M > 0x7fffffff) return RT_E_SHAPE;
  const bool narrow = Cout <= 128;                     // 256x128 tile: every wave has live columns
  const int bn = narrow ? 128 : BN;
  const int tiles = (int)((M + BM - 1) / BM) * ((Cout + bn - 1) / bn);
  static bool attr_set = false;
  if (!attr_set) {
    for (const void* f : {reinterpret_cast<const void*>(conv_nhwc_kernel<4>), reinterpret_cast<const void*>(conv_nhwc_kernel<2>)}) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) return (int)e;
    }
    attr_set = true;
  }
  if (narrow) hipLaunchKernelGGL(conv_nhwc_kernel<2>, dim3(tiles), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(conv_nhwc_kernel<4>, dim3(tiles), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, a);
  return rt_hip_status();
}

static inline int gn_pix_per_block(int HW) {   // at most 1024 workgroups per image, and as close to that as 32-pixel blocks allow: the
  // decoder's 128x128 ... 512x512 stages ran on 16 ... 256 workgroups of 1024 pixels (45 us for 17 MB at 128x128: latency, not bytes)
  int ppb = 32;
  while ((HW + ppb - 1) / ppb > 1024) ppb *= 2;
  return ppb;
}

int64_t rt_groupnorm_ws_bytes(int32_t B, int32_t H, int32_t W, int32_t G) {
  if (B < 1 || H < 1 || W < 1 || G < 1) return 0;
  const int HW = H * W;
  const int ppb = gn_pix_per_block(HW);
  const int64_t nblk = (HW + ppb - 1) / ppb;
  return (int64_t)B * 2 * G * (nblk + 1) * (int64_t)sizeof(double);
}

int rt_groupnorm_silu_nhwc(const void* x, void* y, const void* gamma, const void* beta, void* stats_ws, int32_t B,
                           int32_t H, int32_t W, int32_t C, int32_t G, float eps, int32_t silu, void* stream) {
  if (!x || !y || !gamma || !beta || !stats_ws || B < 1 || H < 1 || W < 1 || C < 8 || G < 1) return RT_E_BADARG;
  if (C % 8 || C % G || 256 % (C / 8) || C / 8 > 256) return RT_E_SHAPE;
  if (!RT_ALIGNED(x, 16) || !RT_ALIGNED(y, 16) || !RT_ALIGNED(gamma, 16) || !RT_ALIGNED(beta, 16) || !RT_ALIGNED(stats_ws, 8)) return RT_E_ALIGN;
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W;
  const int ppb = gn_pix_per_block(HW);
  const int nblk = (HW + ppb - 1) / ppb;
  float* mr = (float*)stats_ws;                            // [B][G][mean, 1/std] (in the first B*2G doubles of the workspace)
  double* part = (double*)stats_ws + (int64_t)B * 2 * G;   // [B][nblk][2G] workgroup partials
  hipLaunchKernelGGL(gn_stats_kernel, dim3(nblk, B), dim3(256), 0, st, (const bf16_t*)x, part, B, H, W, C, G, ppb);
  if (2 * G > 1024) return RT_E_SHAPE;
  hipLaunchKernelGGL(gn_reduce_kernel, dim3(B), dim3(1024), 0, st, (const double*)part, nblk, 2 * G, mr, (double)HW * (C / G), eps);
  if ((int64_t)B * H > 65535) return RT_E_SHAPE;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((W * (C / 8) + 255) / 256, B * H), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y,
                     (const float*)mr, (const bf16_t*)gamma, (const bf16_t*)beta, B, H, W, C, G, silu);
  return rt_hip_status();
}

int rt_softmax_rows(const float* s, void* p, int32_t rows, int32_t cols, float scale, void* stream) {
  if (!s || !p || rows < 1 || cols < 4) return RT_E_BADARG;
  if (cols % 4) return RT_E_SHAPE;
  if (!RT_ALIGNED(s, 16) || !RT_ALIGNED(p, 8)) return RT_E_ALIGN;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, s, (bf16_t*)p, cols, scale);
  return rt_hip_status();
}

int rt_transpose_bf16(const void* in, void* out, int32_t R, int32_t C, int64_t ld_in, int64_t ld_out, void* stream) {
  if (!in || !out || R < 1 || C < 1 || ld_in < C || ld_out < R) return RT_E_BADARG;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)in, (bf16_t*)out, R, C, ld_in, ld_out);
  return rt_hip_status();
}

int rt_image_out(const float* x, float* nchw, uint8_t* u8, int32_t B, int32_t H, int32_t W, int32_t Cp, int32_t C,
                 void* stream) {
  if (!x || (!nchw && !u8) || B < 1 || H < 1 || W < 1 || C < 1 || Cp < C) return RT_E_BADARG;
  hipLaunchKernelGGL(image_out_kernel, dim3(grid_for((int64_t)B * H * W, 256)), dim3(256), 0, (hipStream_t)stream, x, nchw,
                     u8, B, H, W, Cp, C);
  return rt_hip_status();
}

int rt_nchw_to_haloed_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Cp, void* stream) {
  if (!x || !y || B < 1 || C < 1 || H < 1 || W < 1 || Cp < C) return RT_E_BADARG;
  hipLaunchKernelGGL(nchw_to_haloed_nhwc_kernel, dim3(grid_for((int64_t)B * H * W * Cp, 256)), dim3(256), 0,
                     (hipStream_t)stream, x, (bf16_t*)y, B, C, H, W, Cp);
  return rt_hip_status();
}

int rt_haloed_nhwc_to_nchw(const void* x, float* y, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Cp, void* stream) {
  if (!x || !y || B < 1 || C < 1 || H < 1 || W < 1 || Cp < C) return RT_E_BADARG;
  hipLaunchKernelGGL(haloed_nhwc_to_nchw_kernel, dim3(grid_for((int64_t)B * C * H * W, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)x, y, B, C, H, W, Cp);
  return rt_hip_status();
}

int rt_unpack_latents_haloed(const void* packed, void* y, int32_t B, int32_t C, int32_t H2, int32_t W2, int32_t Cp,
                             float inv_scale, float shift, void* stream) {
  if (!packed || !y || B < 1 || C < 1 || H2 < 2 || W2 < 2 || H2 % 2 || W2 % 2 || Cp < C) return RT_E_BADARG;
  hipLaunchKernelGGL(unpack_to_haloed_kernel, dim3(grid_for((int64_t)B * H2 * W2 * Cp, 256)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)packed, (bf16_t*)y, B, C, H2, W2, Cp, inv_scale, shift);
  return rt_hip_status();
}

int rt_resize2d(const void* in, int32_t in_u8, float in_scale, float* out, int32_t planes, int32_t H, int32_t W, int32_t OH, int32_t OW,
                float scale_h, float scale_w, int32_t bilinear, void* stream) {
  if (!in || !out || planes < 1 || H < 1 || W < 1 || OH < 1 || OW < 1 || (in_u8 && in_scale == 0.f)) return RT_E_BADARG;
  // ATen: ratio = (float)(1.0 / (double)scale) when a scale factor was given, else (float)in / out
  const float rh = scale_h > 0.f ? (float)(1.0 / (double)scale_h) : (float)H / (float)OH, rw = scale_w > 0.f ? (float)(1.0 / (double)scale_w) : (float)W / (float)OW;
  hipLaunchKernelGGL(resize2d_kernel, dim3(grid_for((int64_t)planes * OH * OW, 256)), dim3(256), 0, (hipStream_t)stream, in, in_u8, in_scale,
                     out, planes, H, W, OH, OW, rh, rw, bilinear);
  return rt_hip_status();
}

int rt_glyph_blend(const float* image, const float* latents, const float* noise, float* out, int32_t B, int32_t Cimg, int32_t H, int32_t W,
                   int32_t Cl, int32_t OH, int32_t OW, void* stream) {
  if (!image || !latents || !noise || !out || B < 1 || Cimg < 1 || H < 1 || W < 1 || Cl < 1 || OH < 1 || OW < 1) return RT_E_BADARG;
  hipLaunchKernelGGL(glyph_blend_kernel, dim3(grid_for((int64_t)B * OH * OW, 256)), dim3(256), 0, (hipStream_t)stream, image, latents, noise,
                     out, B, Cimg, H, W, Cl, OH, OW, (float)H / (float)OH, (float)W / (float)OW);
  return rt_hip_status();
}

}  // extern "C"
