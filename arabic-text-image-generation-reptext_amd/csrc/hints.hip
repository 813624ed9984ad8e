// Caller-side hint preparation on the device: what infer.py:16-22,98-100 does with cv2.Canny(image, 50, 100)
// and what PIPE:680,694,970 do with VaeImageProcessor.preprocess, for uint8 images that already have the target size.
//
// rt_canny_u8 restates cv::Canny's documented algorithm exactly as reptext_amd/hints.py::canny_edges does on the host (the bar is
// bit-exactness against that function; parity with OpenCV itself is unpinned — cv2 is absent):
//   1. 3x3 Sobel with replicated borders, per channel; per pixel the channel with the largest |dx|+|dy| supplies (dx, dy), the
//      first channel winning ties (cv::Canny for cn > 1; no gray conversion);
//   2. L1 magnitude; non-maximum suppression over four direction sectors split at tan 22.5° / tan 67.5° with cv::Canny's tie
//      rules (strict '>' towards left/up, '>=' towards right/down, strict on both diagonals); magnitudes outside the image are 0;
//   3. double threshold (mag > low: candidate, mag > high: strong);
//   4. hysteresis: every candidate 8-connected to a strong pixel through candidates becomes an edge. The edge SET is unique, so
//      the traversal order is free: one workgroup runs a breadth-first search from the strong pixels over index frontiers in the
//      workspace (all byte/integer work; the image is ~1 M pixels, the frontiers a few ten thousand). No host round trip, no
//      grid-wide barrier, graph-capturable.
// All byte / integer arithmetic apart from the two tangent comparisons, which hints.py makes in float64 — as here.
#include "rt_common.h"

namespace {

constexpr int HT = 1024;   // threads of the hysteresis workgroup

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// dx, dy (int16) and magnitude (int16, <= 2040) per pixel
__global__ void canny_grad_kernel(const uint8_t* __restrict__ img, int H, int W, int C, int16_t* __restrict__ gx, int16_t* __restrict__ gy,
                                  int16_t* __restrict__ mag) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int xl = clampi(x - 1, 0, W - 1), xr = clampi(x + 1, 0, W - 1), yu = clampi(y - 1, 0, H - 1), yd = clampi(y + 1, 0, H - 1);
  int bx = 0, by = 0, best = -1;
  for (int c = 0; c < C; ++c) {
    auto px = [&](int yy, int xx) -> int { return (int)img[((int64_t)yy * W + xx) * C + c]; };
    const int tl = px(yu, xl), tc = px(yu, x), tr = px(yu, xr), ml = px(y, xl), mr = px(y, xr), bl = px(yd, xl), bc = px(yd, x), br = px(yd, xr);
    const int dx = (tr + 2 * mr + br) - (tl + 2 * ml + bl);
    const int dy = (bl + 2 * bc + br) - (tl + 2 * tc + tr);
    const int m = abs(dx) + abs(dy);
    if (m > best) { best = m; bx = dx; by = dy; }
  }
  const int64_t o = (int64_t)y * W + x;
  gx[o] = (int16_t)bx;
  gy[o] = (int16_t)by;
  mag[o] = (int16_t)best;
}

// state: 0 = nothing, 1 = candidate (mag > low, survived NMS), 2 = strong (mag > high)
__global__ void canny_nms_kernel(const int16_t* __restrict__ gx, const int16_t* __restrict__ gy, const int16_t* __restrict__ mag, int H, int W,
                                 float low, float high, uint8_t* __restrict__ state) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  auto m = [&](int yy, int xx) -> int { return (yy < 0 || yy >= H || xx < 0 || xx >= W) ? 0 : (int)mag[(int64_t)yy * W + xx]; };
  const int64_t o = (int64_t)y * W + x;
  const int dx = gx[o], dy = gy[o], c = mag[o];
  const int ax = abs(dx), ay = abs(dy);
  const bool horiz = (double)ay < (double)ax * 0.41421356237;
  const bool vert = (double)ay > (double)ax * 2.41421356237;
  bool keep;
  if (horiz) keep = c > m(y, x - 1) && c >= m(y, x + 1);
  else if (vert) keep = c > m(y - 1, x) && c >= m(y + 1, x);
  else if ((dx ^ dy) >= 0) keep = c > m(y - 1, x - 1) && c > m(y + 1, x + 1);
  else keep = c > m(y - 1, x + 1) && c > m(y + 1, x - 1);
  uint8_t s = 0;
  if (keep && (float)c > low) s = ((float)c > high) ? 2 : 1;
  state[o] = s;
}

// One workgroup: breadth-first growth of the strong set through candidates. frontier A/B: int32 pixel indices in the workspace.
__global__ __launch_bounds__(HT) void canny_hysteresis_kernel(uint8_t* state, int H, int W, int* fa, int* fb) {
  __shared__ int n_cur, n_next;
  const int tid = threadIdx.x;
  const int64_t npix = (int64_t)H * W;
  if (tid == 0) { n_cur = 0; n_next = 0; }
  __syncthreads();
  // level 0: every strong pixel (its candidates neighbours are visited in the loop below)
  for (int64_t p = tid; p < npix; p += HT)
    if (state[p] == 2) fa[atomicAdd(&n_cur, 1)] = (int)p;
  __syncthreads();
  int* cur = fa;
  int* nxt = fb;
  while (true) {
    const int n = n_cur;
    if (n == 0) break;
    for (int i = tid; i < n; i += HT) {
      const int p = cur[i];
      const int y = p / W, x = p - y * W;
#pragma unroll
      for (int dyy = -1; dyy <= 1; ++dyy)
#pragma unroll
        for (int dxx = -1; dxx <= 1; ++dxx) {
          const int yy = y + dyy, xx = x + dxx;
          if ((dyy | dxx) == 0 || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
          const int q = yy * W + xx;
          // claim a candidate exactly once: byte-wide compare-and-swap through its aligned 32-bit word
          unsigned* wp = reinterpret_cast<unsigned*>(state + (q & ~3));
          const unsigned sh = (unsigned)(q & 3) * 8u;
          unsigned old = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          while (((old >> sh) & 0xffu) == 1u) {
            const unsigned want = (old & ~(0xffu << sh)) | (2u << sh);
            if (__hip_atomic_compare_exchange_strong(wp, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
              nxt[atomicAdd(&n_next, 1)] = q;
              break;
            }
          }
        }
    }
    __syncthreads();
    if (tid == 0) { n_cur = n_next; n_next = 0; }
    __syncthreads();
    int* t = cur; cur = nxt; nxt = t;
  }
}

// edges: 255 where state == 2; optionally inverted (255 - e) and replicated into `oc` interleaved channels (infer.py:16-22)
__global__ void canny_out_kernel(const uint8_t* __restrict__ state, int64_t npix, int oc, int invert, uint8_t* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  uint8_t e = state[p] == 2 ? 255 : 0;
  if (invert) e = 255 - e;
  for (int c = 0; c < oc; ++c) out[p * oc + c] = e;
}

// VaeImageProcessor.preprocess of uint8 HWC images at their own size: float32 NCHW, x/255 then (normalize) 2x - 1.
// Rounding steps as numpy/torch take them: one fp32 division, then 2·q (exact) minus 1 (one rounding).
__global__ void preprocess_u8_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int B, int H, int W, int C, int normalize) {
#pragma clang fp contract(off)
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)B * C * H * W;
  if (idx >= n) return;
  const int x = (int)(idx % W), y = (int)((idx / W) % H), c = (int)((idx / ((int64_t)W * H)) % C), b = (int)(idx / ((int64_t)W * H * C));
  float v = __fdiv_rn((float)img[(((int64_t)b * H + y) * W + x) * C + c], 255.0f);
  if (normalize) v = __fsub_rn(__fmul_rn(2.0f, v), 1.0f);
  out[idx] = v;
}

inline int64_t align256(int64_t n) { return (n + 255) / 256 * 256; }

}  // namespace

extern "C" {

int64_t rt_canny_ws_bytes(int32_t H, int32_t W) {
  if (H < 1 || W < 1) return 0;
  const int64_t n = (int64_t)H * W;
  // gx, gy, mag (int16) | state (u8, padded to a multiple of 4: claimed through 32-bit words) | two frontiers (int32)
  return 3 * align256(n * 2) + align256(n + 4) + 2 * align256(n * 4);
}

int rt_canny_u8(const uint8_t* img, int32_t H, int32_t W, int32_t C, float low, float high, uint8_t* out, int32_t out_channels,
                int32_t invert, void* ws, int64_t ws_bytes, void* stream) {
  if (!img || !out || !ws || H < 1 || W < 1 || C < 1 || C > 4 || out_channels < 1 || out_channels > 4) return RT_E_BADARG;
  if ((int64_t)H * W >= ((int64_t)1 << 31)) return RT_E_SHAPE;
  if (ws_bytes < rt_canny_ws_bytes(H, W) || !RT_ALIGNED(ws, 256)) return RT_E_BADARG;
  if (low > high) { const float t = low; low = high; high = t; }
  const int64_t n = (int64_t)H * W;
  char* p = static_cast<char*>(ws);
  int16_t* gx = reinterpret_cast<int16_t*>(p); p += align256(n * 2);
  int16_t* gy = reinterpret_cast<int16_t*>(p); p += align256(n * 2);
  int16_t* mag = reinterpret_cast<int16_t*>(p); p += align256(n * 2);
  uint8_t* state = reinterpret_cast<uint8_t*>(p); p += align256(n + 4);
  int* fa = reinterpret_cast<int*>(p); p += align256(n * 4);
  int* fb = reinterpret_cast<int*>(p);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((W + 255) / 256, H);
  hipLaunchKernelGGL(canny_grad_kernel, grid, dim3(256), 0, st, img, H, W, C, gx, gy, mag);
  hipLaunchKernelGGL(canny_nms_kernel, grid, dim3(256), 0, st, (const int16_t*)gx, (const int16_t*)gy, (const int16_t*)mag, H, W, low, high, state);
  hipLaunchKernelGGL(canny_hysteresis_kernel, dim3(1), dim3(HT), 0, st, state, H, W, fa, fb);
  hipLaunchKernelGGL(canny_out_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const uint8_t*)state, n, out_channels, invert, out);
  return rt_hip_status();
}

int rt_preprocess_u8(const uint8_t* img, float* out, int32_t B, int32_t H, int32_t W, int32_t C, int32_t normalize, void* stream) {
  if (!img || !out || B < 1 || H < 1 || W < 1 || C < 1) return RT_E_BADARG;
  const int64_t n = (int64_t)B * C * H * W;
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, out, B, H, W, C, normalize);
  return rt_hip_status();
}

}  // extern "C"
