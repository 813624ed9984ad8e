// Bare MFMA-rate probe: register-only loops on random data, 2 waves per SIMD, all CUs. bf16 16x16x32 vs block-scaled fp8
// 16x16x128 (scales = 1.0). Prints TFLOP/s. Used to decide whether an fp8 GEMM path can pay on a clock-limited chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE> __global__ __launch_bounds__(512) void probe(const uint32_t* in, float* out, int iters) {
  const int lane = threadIdx.x;
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) {
    bf16x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(in + (lane * 16 + i * 4) % 4096);
    for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const bf16x8*>(in + (lane * 8 + i * 4 + 64) % 4096);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
    }
  } else {
    i32x8 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const i32x8*>(in + (lane * 32 + i * 8) % 4096);
    for (int i = 0; i < 2; ++i) b[i] = *reinterpret_cast<const i32x8*>(in + (lane * 16 + i * 8 + 64) % 4096);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i * 2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i * 2 + j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  std::vector<uint32_t> h(4096 + 64);
  uint32_t x = 99;
  for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (x & 0x7F7F7F7F) | 0x30303030 & 0x3F3F3F3F; v = x & 0xBFBFBFBF & 0x7B7B7B7B | 0x30303030; }   // moderate exponents, no NaN patterns
  uint32_t* in; float* out;
  hipMalloc(&in, h.size() * 4); hipMalloc(&out, 256 * 2 * 512 * 4);
  hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, grid = 256;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, 0);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 0, 0, in, out, iters);
      else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 0, 0, in, out, iters);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)grid * 8 /*waves*/ * iters * 8 /*mfma*/ * 2.0 * 16 * 16 * (mode == 0 ? 32 : 128);
      printf("%s: %.2f ms  %.1f TFLOP/s\n", mode == 0 ? "bf16 16x16x32" : "fp8 scaled 16x16x128", ms, flops / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
