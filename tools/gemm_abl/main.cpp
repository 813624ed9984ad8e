#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>
#include "reptext_hip.h"
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4608, N = argc > 2 ? atoi(argv[2]) : 21504, K = argc > 3 ? atoi(argv[3]) : 3072;
  std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
  uint32_t x = 777;
  auto rnd = [&](float sc) { x = x * 1664525u + 1013904223u; float f = (((x >> 8) & 0xFFFF) / 65536.f * 2.f - 1.f) * sc; uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
  for (auto& v : ha) v = rnd(1.f);
  for (auto& v : hw) v = rnd(0.03f);
  uint16_t *a, *w, *c;
  hipMalloc(&a, ha.size() * 2); hipMalloc(&w, hw.size() * 2); hipMalloc(&c, (size_t)M * N * 2);
  hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  rt_gemm_group g{};
  g.A = a; g.W = w; g.C = c; g.lda = K; g.ldw = K; g.ldc = N; g.M = M; g.N = N; g.K = K; g.batch = 1; g.gelu_from = N; g.alpha = 1.f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) if (rt_gemm_bf16(&g, 1, 0)) { printf("launch failed\n"); return 1; }
  hipDeviceSynchronize();
  const int it = 30;
  hipEventRecord(e0, 0);
  for (int i = 0; i < it; ++i) rt_gemm_bf16(&g, 1, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double t = ms / it * 1e-3;
  printf("%-44s %dx%dx%d: %.1f us  %.1f TF/s\n", argv[0], M, N, K, t * 1e6, 2.0 * M * N * (double)K / t / 1e12);
  return 0;
}
