#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstdint>
#include <cstring>
extern "C" int rt_abl_read_stamps(unsigned long long*, int);
extern "C" int rt_attention_fwd(const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, int64_t, int32_t, int32_t, int32_t, float, void*);
int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 4608, H = 24, d = H * 128;
  size_t n = (size_t)S * 3 * d;
  std::vector<uint16_t> h(n);
  uint32_t x = 12345;
  for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; float f = ((x >> 8) & 0xFFFF) / 65536.f * 2.f - 1.f; uint32_t u; memcpy(&u, &f, 4); h[i] = u >> 16; }
  uint16_t *qkv, *o;
  hipMalloc(&qkv, n * 2); hipMalloc(&o, (size_t)S * d * 2);
  hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) rt_attention_fwd(qkv, qkv + d, qkv + 2 * d, o, 3 * d, (int64_t)S * 3 * d, d, (int64_t)S * d, 1, S, H, 0.0883883f, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int it = 30;
  for (int i = 0; i < it; ++i) rt_attention_fwd(qkv, qkv + d, qkv + 2 * d, o, 3 * d, (int64_t)S * 3 * d, d, (int64_t)S * d, 1, S, H, 0.0883883f, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double t = ms / it * 1e-3;
  printf("%s S=%d: %.1f us  %.1f TF/s\n", argv[0], S, t * 1e6, 4.0 * H * S * (double)S * 128 / t / 1e12);
  {
    const int nwg = ((S + 127) / 128) * H;
    std::vector<unsigned long long> st(4 * 8192);
    if (rt_abl_read_stamps(st.data(), 4 * 8192) == 0) {
      std::vector<double> clk, cyc;
      unsigned long long rmin = ~0ull, rmax = 0;
      for (int i = 0; i < nwg && i < 8192; ++i) if (st[4 * i + 1]) { clk.push_back((double)st[4 * i] / st[4 * i + 1] * 0.1); cyc.push_back((double)st[4 * i]); rmin = std::min(rmin, st[4*i+2]); rmax = std::max(rmax, st[4*i+3]); }
      std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
      if (!clk.empty()) printf("   in-kernel clock median %.3f GHz; loop cycles per WG median %.0f (min %.0f max %.0f); first-start..last-end %.1f us\n", clk[clk.size()/2], cyc[cyc.size()/2], cyc.front(), cyc.back(), (rmax - rmin) * 0.01);
    }
  }
  return 0;
}
