// Development harness for csrc/attention.hip (no torch): correctness against a naive fp32 GPU reference on sampled rows,
// bitwise repeatability, batch invariance, and interleaved timing of the split / unsplit paths.
//   hipcc --offload-arch=gfx950 -O3 -fno-honor-nans -Iinclude -Icsrc csrc/attention.hip tools/attn_dev/main.cpp -o tools/attn_dev/attn_dev
//   tools/attn_dev/attn_dev [S] [B] [iters]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "reptext_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static inline float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

// one thread per (row, head): naive softmax(q k^T) v in fp32 for the sampled rows
__global__ void ref_rows(const uint16_t* qkv, int S, int H, int ld, const int* rows, int nrows, float scale, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows * H) return;
  const int r = rows[i / H], h = i % H;
  const uint16_t* q = qkv + (size_t)r * ld + h * 128;
  float qf[128];
  for (int d = 0; d < 128; ++d) qf[d] = __uint_as_float((uint32_t)q[d] << 16);
  float m = -INFINITY, l = 0.f, acc[128];
  for (int d = 0; d < 128; ++d) acc[d] = 0.f;
  const int dm = H * 128;
  for (int k = 0; k < S; ++k) {
    const uint16_t* kp = qkv + (size_t)k * ld + dm + h * 128;
    float s = 0.f;
    for (int d = 0; d < 128; ++d) s += qf[d] * __uint_as_float((uint32_t)kp[d] << 16);
    s *= scale;
    const float mn = fmaxf(m, s), a = __expf(m - mn), p = __expf(s - mn);
    const uint16_t* vp = qkv + (size_t)k * ld + 2 * dm + h * 128;
    for (int d = 0; d < 128; ++d) acc[d] = acc[d] * a + p * __uint_as_float((uint32_t)vp[d] << 16);
    l = l * a + p;
    m = mn;
  }
  for (int d = 0; d < 128; ++d) out[(size_t)i * 128 + d] = acc[d] / l;
}

int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 4608, B = argc > 2 ? atoi(argv[2]) : 1, iters = argc > 3 ? atoi(argv[3]) : 20;
  const int H = 24, d = H * 128, ld = 3 * d;
  const float scale = 0.08838834764831845f;
  const size_t n1 = (size_t)S * ld, n = n1 * B;
  std::vector<uint16_t> h(n);
  uint32_t x = 12345;
  for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; float f = (((x >> 8) & 0xFFFF) / 65536.f * 2.f - 1.f) * 1.7f; uint32_t u; memcpy(&u, &f, 4); h[i] = (uint16_t)(u >> 16); }
  // spike: one key row far larger than the rest against a few query rows of head 3 (forces the rare rescale branch late in the sweep)
  const int spike = getenv("ATTN_DEV_SPIKE") ? atoi(getenv("ATTN_DEV_SPIKE")) : S - 70;
  if (S > 200) for (int dd = 0; dd < 128; ++dd) { h[(size_t)spike * ld + d + 3 * 128 + dd] = 0x4120; /* 10.0 */ h[(size_t)77 * ld + 3 * 128 + dd] = 0x3f80; /* 1.0 */ }
  if (getenv("ATTN_DEV_PATTERN")) {   // q = k = 0 (uniform attention), v[key][*] = key / S: an item's output is the mean key position of the keys it saw
    for (size_t r = 0; r < (size_t)B * S; ++r)
      for (int c = 0; c < ld; ++c) {
        float f = c < 2 * d ? 0.f : (float)(r % S) / S;
        uint32_t u; memcpy(&u, &f, 4); h[r * ld + c] = (uint16_t)(u >> 16);
      }
  }
  uint16_t *qkv, *o0, *o1, *o2;
  CK(hipMalloc(&qkv, n * 2));
  const size_t on = (size_t)B * S * d;
  CK(hipMalloc(&o0, on * 2)); CK(hipMalloc(&o1, on * 2)); CK(hipMalloc(&o2, on * 2));
  CK(hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice));
  const int64_t wsb = rt_attention_ws_bytes(B, S, H);
  void* ws = nullptr;
  if (wsb) { CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb)); }
  printf("S=%d B=%d H=%d ws=%.1f MB\n", S, B, H, wsb / 1e6);
  auto run = [&](uint16_t* o, bool split) {
    int rc = rt_attention_fwd(qkv, qkv + d, qkv + 2 * d, o, ld, (int64_t)S * ld, d, (int64_t)S * d, B, S, H, scale, split ? ws : nullptr, split ? wsb : 0, 0);
    if (rc) { printf("rt_attention_fwd rc=%d\n", rc); exit(3); }
  };
  CK(hipMemset(o0, 0xff, on * 2)); CK(hipMemset(o1, 0xff, on * 2)); CK(hipMemset(o2, 0xff, on * 2));
  run(o0, false); run(o1, true); run(o2, true);
  CK(hipDeviceSynchronize());
  // ---- reference on sampled rows of batch entry 0 (and the last entry)
  std::vector<int> rows;
  for (int r = 0; r < S; r += std::max(1, S / 97)) rows.push_back(r);
  rows.push_back(S - 1); rows.push_back(std::min(S - 1, 77)); rows.push_back(std::min(S - 1, 127)); rows.push_back(std::min(S - 1, 128));
  int* drows; float* dref;
  CK(hipMalloc(&drows, rows.size() * 4)); CK(hipMalloc(&dref, rows.size() * H * 128 * 4));
  CK(hipMemcpy(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
  std::vector<uint16_t> ho0(on), ho1(on), ho2(on);
  CK(hipMemcpy(ho0.data(), o0, on * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho1.data(), o1, on * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho2.data(), o2, on * 2, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int bb : {0, B - 1}) {
    hipLaunchKernelGGL(ref_rows, dim3((rows.size() * H + 63) / 64), dim3(64), 0, 0, qkv + (size_t)bb * n1, S, H, ld, drows, (int)rows.size(), scale, dref);
    std::vector<float> ref(rows.size() * H * 128);
    CK(hipMemcpy(ref.data(), dref, ref.size() * 4, hipMemcpyDeviceToHost));
    for (int which = 0; which < 2; ++which) {
      const std::vector<uint16_t>& ho = which ? ho1 : ho0;
      double num = 0, den = 0, mx = 0;
      for (size_t i = 0; i < rows.size(); ++i)
        for (int c = 0; c < d; ++c) {
          const float g = bf2f(ho[((size_t)bb * S + rows[i]) * d + c]), r = ref[(i * H + c / 128) * 128 + c % 128];
          num += (double)(g - r) * (g - r); den += (double)r * r; mx = std::max(mx, (double)fabsf(g - r));
        }
      const double rel = sqrt(num / den);
      printf("  batch %d %s: rel-L2 vs fp32 reference on %zu rows = %.3e  max abs %.3e %s\n", bb, which ? "split  " : "unsplit", rows.size(), rel, mx, rel < 6e-3 ? "ok" : "FAIL");
      if (!(rel < 6e-3)) bad = 1;
    }
    if (B == 1) break;
  }
  size_t diff12 = 0, diff01 = 0, nan0 = 0;
  for (size_t i = 0; i < on; ++i) { diff12 += ho1[i] != ho2[i]; diff01 += ho0[i] != ho1[i]; nan0 += (ho1[i] & 0x7f80) == 0x7f80; }
  printf("  split run 1 vs run 2: %zu differing values (must be 0); unsplit vs split: %zu of %zu differ; non-finite %zu\n", diff12, diff01, on, nan0);
  if (diff12 || nan0) bad = 1;
  { int shown = 0; for (size_t i = 0; i < on && shown < 4; ++i) if ((ho1[i] & 0x7f80) == 0x7f80) { printf("    non-finite at row %zu head %zu col %zu (unsplit there: %04x)\n", (i / d) % S, (i % d) / 128, i % 128, ho0[i]); i = (i / 128 + 1) * 128; ++shown; } }
  if (getenv("ATTN_DEV_PATTERN")) {
    const int nqb = (S + 127) / 128;
    for (int hd = 0; hd < H; ++hd) {
      printf("  head %2d:", hd);
      for (int qb = 0; qb < nqb; ++qb) printf(" %.3f", bf2f(ho1[(size_t)(qb * 128) * d + hd * 128]));
      printf("\n");
    }
  }
  if (getenv("ATTN_DEV_RECORDS") && ws) {   // recombine one row of the first split item of XCD group 0 on the host
    std::vector<char> hw(wsb);
    CK(hipMemcpy(hw.data(), ws, wsb, hipMemcpyDeviceToHost));
    const int NI = H * ((S + 127) / 128), nqb = (S + 127) / 128, spx = 64, nfull = (NI / 8 / spx) * spx;
    const size_t cntb = ((size_t)B * NI * 4 + 255) / 256 * 256, REC = 67584, RW = 16896;
    const int item = nfull, hd = item / nqb, qb = item % nqb;
    printf("  first split item of group 0: item %d = head %d qblock %d\n", item, hd, qb);
    for (int lane : {0, 5}) {
      double M = -1e30, m[2], l[2][2], o[2][4];
      for (int j = 0; j < 2; ++j) {
        const char* rj = hw.data() + cntb + ((size_t)(0 * spx + j) * 2 + 0) * REC + 0 * RW;
        const float* ml0 = (const float*)(rj + 16384 + lane * 8); const float* ml1 = (const float*)(rj + 16384 + (lane + 32) * 8);
        m[j] = ml0[0]; l[j][0] = ml0[1]; l[j][1] = ml1[1];
        for (int e = 0; e < 4; ++e) o[j][e] = ((const float*)(rj + lane * 16))[e];
        printf("    lane %d part %d: m %.4f (lane+32: %.4f)  l %.5f + %.5f  O[0..3] %.5f %.5f %.5f %.5f\n", lane, j, m[j], ml1[0], l[j][0], l[j][1], o[j][0], o[j][1], o[j][2], o[j][3]);
        M = std::max(M, m[j]);
      }
      double L = 0, oo[4] = {0, 0, 0, 0};
      for (int j = 0; j < 2; ++j) { const double w = exp2(m[j] - M); L += (l[j][0] + l[j][1]) * w; for (int e = 0; e < 4; ++e) oo[e] += o[j][e] * w; }
      const size_t base = (size_t)(qb * 128 + lane) * d + hd * 128;
      printf("    host recombine: %.5f %.5f %.5f %.5f | split kernel: %.5f %.5f %.5f %.5f | unsplit kernel: %.5f %.5f %.5f %.5f\n", oo[0] / L, oo[1] / L, oo[2] / L, oo[3] / L,
             bf2f(ho1[base]), bf2f(ho1[base + 1]), bf2f(ho1[base + 2]), bf2f(ho1[base + 3]), bf2f(ho0[base]), bf2f(ho0[base + 1]), bf2f(ho0[base + 2]), bf2f(ho0[base + 3]));
    }
  }
  if (getenv("ATTN_DEV_VERBOSE")) {   // per (head, query block): how far is the split result from the unsplit one
    const int nqb = (S + 127) / 128;
    for (int hd = 0; hd < H; ++hd) {
      printf("  head %2d:", hd);
      for (int qb = 0; qb < nqb; ++qb) {
        double mx = 0;
        for (int r = qb * 128; r < std::min(S, qb * 128 + 128); ++r)
          for (int c = 0; c < 128; ++c) mx = std::max(mx, (double)fabsf(bf2f(ho0[(size_t)r * d + hd * 128 + c]) - bf2f(ho1[(size_t)r * d + hd * 128 + c])));
        printf(" %.0e", mx);
      }
      printf("\n");
    }
  }
  if (B > 1) {   // batch invariance: entry 0 of this batch against a B = 1 launch on the same data
    uint16_t* o3; CK(hipMalloc(&o3, (size_t)S * d * 2));
    const int64_t w1 = rt_attention_ws_bytes(1, S, H);
    void* ws1 = nullptr; if (w1) { CK(hipMalloc(&ws1, w1)); CK(hipMemset(ws1, 0, w1)); }
    int rc = rt_attention_fwd(qkv, qkv + d, qkv + 2 * d, o3, ld, (int64_t)S * ld, d, (int64_t)S * d, 1, S, H, scale, ws1, w1, 0);
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> h3((size_t)S * d);
    CK(hipMemcpy(h3.data(), o3, h3.size() * 2, hipMemcpyDeviceToHost));
    size_t df = 0; for (size_t i = 0; i < h3.size(); ++i) df += h3[i] != ho1[i];
    printf("  batch invariance (entry 0 at B=%d vs B=1): %zu differing values (must be 0) rc=%d\n", B, df, rc);
    if (df) bad = 1;
  }
  // ---- timing: interleaved rounds in one process
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> tu, ts;
  for (int round = 0; round < 5; ++round)
    for (int which = 0; which < 2; ++which) {
      for (int i = 0; i < 3; ++i) run(o0, which);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) run(o0, which);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      (which ? ts : tu).push_back(ms / iters * 1e3f);
    }
  std::sort(tu.begin(), tu.end()); std::sort(ts.begin(), ts.end());
  const double fl = 4.0 * B * H * (double)S * S * 128;
  printf("  unsplit: median %.1f us (min %.1f)  %.0f TF/s | split: median %.1f us (min %.1f)  %.0f TF/s\n", tu[2], tu[0], fl / tu[2] / 1e6, ts[2], ts[0], fl / ts[2] / 1e6);
  printf(bad ? "RESULT: FAIL\n" : "RESULT: OK\n");
  return bad;
}
