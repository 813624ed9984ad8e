// The compute boundary without Python or torch: a plain C++ host links librt_reptext_hip.so through include/reptext_hip.h,
// allocates with hipMalloc, calls rt_gemm_bf16 (bias + GELU on the upper columns) and rt_attention_fwd, and checks sampled
// outputs against a scalar CPU computation.  Build + run (GPU box):
//   hipcc -O2 -Iinclude tools/capi_smoke/capi_smoke.cpp -Larabic-text-image-generation-reptext_amd -lrt_reptext_hip \
//         -Wl,-rpath,$PWD/arabic-text-image-generation-reptext_amd -o tools/capi_smoke/capi_smoke && tools/capi_smoke/capi_smoke
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "reptext_hip.h"

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng = 2463534242u;
static float rnd() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return ((rng >> 8) & 0xFFFF) / 32768.f - 1.f; }
static float gelu_tanh(float x) { return 0.5f * x * (1.f + tanhf(0.7978845608f * (x + 0.044715f * x * x * x))); }

int main() {
  printf("%s (abi %d)\n", rt_version(), rt_abi_version());
  int bad = 0;
  {  // ---- linear: C[M][N] = gelu_from(A[M][K] W[N][K]^T + bias)
    const int M = 300, N = 520, K = 192;
    std::vector<uint16_t> a((size_t)M * K), w((size_t)N * K), b(N), c((size_t)M * N);
    for (auto& v : a) v = f2bf(rnd());
    for (auto& v : w) v = f2bf(rnd() * 0.1f);
    for (auto& v : b) v = f2bf(rnd());
    void *da, *dw, *db, *dc;
    hipMalloc(&da, a.size() * 2); hipMalloc(&dw, w.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dc, c.size() * 2);
    hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
    rt_gemm_group g; memset(&g, 0, sizeof g);
    g.A = da; g.W = dw; g.C = dc; g.bias = db; g.lda = K; g.ldw = K; g.ldc = N; g.M = M; g.N = N; g.K = K; g.batch = 1; g.gelu_from = N / 2; g.alpha = 1.f;
    const int rc = rt_gemm_bf16(&g, 1, nullptr);
    hipDeviceSynchronize();
    hipMemcpy(c.data(), dc, c.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int t = 0; t < 400; ++t) {
      const int m = (t * 37) % M, n = (t * 101) % N;
      double acc = 0;
      for (int k = 0; k < K; ++k) acc += (double)bf2f(a[(size_t)m * K + k]) * bf2f(w[(size_t)n * K + k]);
      float ref = (float)acc + bf2f(b[n]);
      if (n >= N / 2) ref = gelu_tanh(ref);
      worst = fmax(worst, fabs(bf2f(c[(size_t)m * N + n]) - ref) / (fabs(ref) + 0.05));
    }
    printf("rt_gemm_bf16 rc=%d worst sampled relative error %.3e\n", rc, worst);
    bad += (rc != 0) || !(worst < 2e-2);
    g.K = 100;                                    // error convention: bad shape -> RT_E_SHAPE, nothing launched
    bad += rt_gemm_bf16(&g, 1, nullptr) != RT_E_SHAPE;
  }
  {  // ---- attention: one head of 128, S = 200 (ragged last tile)
    const int S = 200, H = 1, D = 128;
    std::vector<uint16_t> qkv((size_t)S * 3 * D), o((size_t)S * D);
    for (auto& v : qkv) v = f2bf(rnd());
    void *dq, *dout;
    hipMalloc(&dq, qkv.size() * 2); hipMalloc(&dout, o.size() * 2);
    hipMemcpy(dq, qkv.data(), qkv.size() * 2, hipMemcpyHostToDevice);
    const uint16_t* base = (const uint16_t*)dq;
    const int rc = rt_attention_fwd(base, base + D, base + 2 * D, dout, 3 * D, (int64_t)S * 3 * D, D, (int64_t)S * D, 1, S, H, 1.f / sqrtf((float)D), nullptr, 0, nullptr);
    hipDeviceSynchronize();
    hipMemcpy(o.data(), dout, o.size() * 2, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int qi = 0; qi < S; qi += 7) {
      std::vector<double> p(S);
      double mx = -1e30, sum = 0;
      for (int j = 0; j < S; ++j) {
        double s = 0;
        for (int d = 0; d < D; ++d) s += (double)bf2f(qkv[(size_t)qi * 3 * D + d]) * bf2f(qkv[(size_t)j * 3 * D + D + d]);
        p[j] = s / sqrt((double)D); mx = fmax(mx, p[j]);
      }
      for (int j = 0; j < S; ++j) { p[j] = exp(p[j] - mx); sum += p[j]; }
      for (int d = 0; d < D; d += 5) {
        double acc = 0;
        for (int j = 0; j < S; ++j) acc += p[j] * bf2f(qkv[(size_t)j * 3 * D + 2 * D + d]);
        worst = fmax(worst, fabs(bf2f(o[(size_t)qi * D + d]) - acc / sum) / (fabs(acc / sum) + 0.05));
      }
    }
    printf("rt_attention_fwd rc=%d worst sampled relative error %.3e\n", rc, worst);
    bad += (rc != 0) || !(worst < 3e-2);
  }
  printf(bad ? "FAILED\n" : "OK\n");
  return bad ? 1 : 0;
}
