// rt_attention_fp8_prep / rt_attention_fp8_fwd — BASELINE config 5's "CDNA4 fp8 MFMA attention": the joint attention of the
// MMDiT blocks (A.1 step 6 / A.2) with e4m3 Q, K, V and softmax numerators on v_mfma_scale_f32_32x32x64_f8f6f4 (block scales
// fixed to 1.0), fp32 scores / statistics / accumulators, bf16 output. Same flash-style structure as csrc/attention.hip
// (4 waves = 128 query rows of one (batch, head); 64-key tiles by LDS-DMA into a 2-deep ring; the query index on the lane for
// both products, so the numerators feed the second product from registers) at half the MFMA cycles and half the LDS bytes.
//
// Quantisation is static, no calibration pass: q and k are RMS-normalised (|element| <= sqrt(128)·|w|), so they are stored as
// e4m3(16·x) and the 1/256 is folded into the softmax scale; v is cast as is (saturating at +-448); the numerators
// exp2(s - m) lie in (0, 2^6] by the deferred-rescale rule.
//
// Layouts produced by rt_attention_fp8_prep (which also applies the q/k RMSNorm + RoPE of rt_qk_rmsnorm_rope):
//   qk8 [B][S][2·H·128] bytes: q heads at column h·128, k heads at H·128 + h·128
//   vt8 [B][H][128][S64] bytes, S64 = 64·ceil(S/64): Vᵀ, keys permuted inside every 64-key tile so that the 32 bytes a lane
//       needs for the second product are contiguous: position p = 32·hh + 16·kt + 4·g + e holds key 32·kt + 8·g + 4·hh + e
//       — the key order in which a lane of half hh holds the Sᵀ accumulators of the two 32-key halves kt. Keys >= S are zero.
#include "rt_common.h"
#include <type_traits>

namespace {

constexpr int DH = 128;
constexpr int BQ = 128;
constexpr int BKV = 64;
constexpr int KT_B = BKV * DH;          // K tile bytes (64 rows x 128 B)
constexpr int VT_B = DH * BKV;          // Vᵀ tile bytes (128 rows x 64 B)
constexpr int SLOT_B = KT_B + VT_B;     // 16 KiB
constexpr int THREADS = 256;
constexpr float RESCALE_THR = 6.0f;
constexpr float QK_PRESCALE = 16.0f;
constexpr float P_BIAS = 2.0f;         // numerators are stored as exp2(s - m + 2) <= 2^8 < 448: two more binades above e4m3's floor
constexpr float E4M3_MAX = 448.f;

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef const __attribute__((address_space(3))) char* lds_cptr;

__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }   // see attention.hip

__device__ __forceinline__ i32x8 lds_read32(lds_cptr p0, lds_cptr p1) {
  const i32x4 a = *(const __attribute__((address_space(3))) i32x4*)p0;
  const i32x4 b = *(const __attribute__((address_space(3))) i32x4*)p1;
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ __forceinline__ float sat448(float v) { return fminf(fmaxf(v, -E4M3_MAX), E4M3_MAX); }

// ---------------------------------------------------------------------------------------------------
// prep, part 1: q/k RMSNorm(128) + RoPE (math of qk_rmsnorm_rope_kernel) -> e4m3(16·x). 16 lanes per 128-vector.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_qk_kernel(const bf16_t* __restrict__ buf, int64_t ld, int64_t stride_b, int64_t q_off,
                                                      int64_t k_off, const bf16_t* __restrict__ wq_txt,
                                                      const bf16_t* __restrict__ wk_txt, const bf16_t* __restrict__ wq_img,
                                                      const bf16_t* __restrict__ wk_img, const float* __restrict__ cosv,
                                                      const float* __restrict__ sinv, uint8_t* __restrict__ qk8, int B, int S, int T,
                                                      int H, float eps) {
  // one 16-lane group per (row, head): q then k, so the row's cos/sin entries are loaded once for both (see qk_rmsnorm_rope_kernel)
  const int64_t grp = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  if (grp >= (int64_t)B * S * H) return;
  int64_t t = grp;
  const int h = (int)(t % H); t /= H;
  const int s = (int)(t % S);
  const int b = (int)(t / S);
  const float* cp = cosv + (int64_t)s * 128 + sub * 8;
  const float* sp = sinv + (int64_t)s * 128 + sub * 8;
  const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp), c1 = *reinterpret_cast<const f32x4*>(cp + 4);
  const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
  const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
  const float sn[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
  const bf16_t* row = buf + b * stride_b + (int64_t)s * ld + h * 128 + sub * 8;
  u32x4 uu[2] = {*reinterpret_cast<const u32x4*>(row + q_off), *reinterpret_cast<const u32x4*>(row + k_off)};
#pragma unroll
  for (int isk = 0; isk < 2; ++isk) {
    const bf16_t* w = (s < T) ? (isk ? wk_txt : wq_txt) : (isk ? wk_img : wq_img);
    const u32x4 u = uu[isk];
    const u32x4 wu = *reinterpret_cast<const u32x4*>(w + sub * 8);
    float x[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { x[2 * i] = bf16lo(u[i]); x[2 * i + 1] = bf16hi(u[i]); }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
    float y[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a = x[2 * i] * r * bf16lo(wu[i]);
      const float bq = x[2 * i + 1] * r * bf16hi(wu[i]);
      y[2 * i] = sat448((a * cs[2 * i] - bq * sn[2 * i]) * QK_PRESCALE);
      y[2 * i + 1] = sat448((bq * cs[2 * i + 1] + a * sn[2 * i + 1]) * QK_PRESCALE);
    }
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(y[4], y[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(y[6], y[7], hi, true);
    uint8_t* o = qk8 + ((int64_t)b * S + s) * (2 * H * 128) + (isk ? H * 128 : 0) + h * 128 + sub * 8;
    *reinterpret_cast<u32x2*>(o) = u32x2{(uint32_t)lo, (uint32_t)hi};
  }
}

// ---------------------------------------------------------------------------------------------------
// prep, part 2: V [S][128] bf16 of one (batch, head) -> Vᵀ e4m3 [128][S64], keys permuted per 64-key tile. One workgroup
// per (tile, head, batch): thread = (d row 0..127, half hf 0..1) writes the 32 positions p = 32·hf .. 32·hf+31 of its row.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_vt_kernel(const bf16_t* __restrict__ buf, int64_t ld, int64_t stride_b, int64_t v_off,
                                                      uint8_t* __restrict__ vt8, int S, int S64, int H) {
  __shared__ uint16_t tile[BKV][DH + 2];          // +2: de-phase the column reads
  const int t = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const bf16_t* src = buf + b * stride_b + v_off + h * DH;
  for (int i = threadIdx.x; i < BKV * (DH / 8); i += blockDim.x) {
    const int row = i / (DH / 8), c8 = i - row * (DH / 8);
    const int key = t * BKV + row;
    u32x4 u = u32x4{0u, 0u, 0u, 0u};
    if (key < S) u = *reinterpret_cast<const u32x4*>(src + (int64_t)key * ld + c8 * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) { tile[row][c8 * 8 + 2 * j] = (uint16_t)(u[j] & 0xffffu); tile[row][c8 * 8 + 2 * j + 1] = (uint16_t)(u[j] >> 16); }
  }
  __syncthreads();
  const int d = threadIdx.x >> 1, hf = threadIdx.x & 1;
  uint32_t w[8];
#pragma unroll
  for (int q4 = 0; q4 < 8; ++q4) {                 // dword q4 of the 32 bytes: j = 4·q4 .. +3 ; j = 16·kt + 4·g + e
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = 4 * q4 + e;
      const int kt = j >> 4, g = (j >> 2) & 3;
      const int key = 32 * kt + 8 * g + 4 * hf + e;
      y[e] = sat448(__uint_as_float((uint32_t)tile[key][d] << 16));
    }
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(y[0], y[1], v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(y[2], y[3], v, true);
    w[q4] = (uint32_t)v;
  }
  uint8_t* dst = vt8 + (((int64_t)b * H + h) * DH + d) * S64 + (int64_t)t * BKV + 32 * hf;
  *reinterpret_cast<u32x4*>(dst) = u32x4{w[0], w[1], w[2], w[3]};
  *reinterpret_cast<u32x4*>(dst + 16) = u32x4{w[4], w[5], w[6], w[7]};
}

// ---------------------------------------------------------------------------------------------------
// attention
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS, 2) void attention_fp8_kernel(const uint8_t* __restrict__ qk8, const uint8_t* __restrict__ vt8,
                                                                   bf16_t* O, int64_t ldo, int64_t stride_ob, int S, int S64, int H,
                                                                   float scale_log2, uint8_t* O8, int64_t ldo8, int64_t stride_ob8,
                                                                   uint8_t* bsc, int64_t bsc_plane, int64_t bsc_rows, int bsc_k0) {
  __shared__ __attribute__((aligned(16))) char smem[2 * SLOT_B];   // [slot][K 8 KiB | Vᵀ 8 KiB]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  // XCD-aware placement (speed only): workgroups are dealt round-robin over the 8 XCDs, so blockIdx.x & 7 labels the XCD group;
  // each group owns a CONTIGUOUS run of the (head, query block) items — whole heads — so a head's K/Vᵀ stay in one XCD's L2.
  const int nqb = (S + BQ - 1) / BQ, NI = H * nqb;
  const int xcd = (int)blockIdx.x & 7, xslot = (int)blockIdx.x >> 3, b = (int)blockIdx.y;
  const int ibase = NI >> 3, iextra = NI & 7;
  if (xslot >= ibase + (xcd < iextra ? 1 : 0)) return;                       // padding workgroup of a group with one item fewer
  const int item = xcd * ibase + (xcd < iextra ? xcd : iextra) + xslot;
  const int head = item / nqb;
  const int q0 = (item - head * nqb) * BQ;
  const int ldqk = 2 * H * DH;

  const uint8_t* Qb = qk8 + (int64_t)b * S * ldqk + head * DH;
  const uint8_t* Kb = Qb + H * DH;
  const uint8_t* Vb = vt8 + ((int64_t)b * H + head) * DH * S64;

  // ---- Q fragments (B operand of Sᵀ = K·Qᵀ): lane holds Q[q][64ks + 32hh .. +31] for k-step ks
  i32x8 qf[2];
  {
    const int qrow = min(q0 + wave * 32 + l31, S - 1);
    const uint8_t* qp = Qb + (int64_t)qrow * ldqk + 32 * hh;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const i32x4 a = *reinterpret_cast<const i32x4*>(qp + 64 * ks), c = *reinterpret_cast<const i32x4*>(qp + 64 * ks + 16);
      qf[ks] = __builtin_shufflevector(a, c, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  }

  // ---- staging by buffer_load ... lds (descriptor + invariant VGPR offset + per-tile SGPR offset).
  // K tile: 8 pieces of 8 rows x 128 B; wave w stages pieces 2w, 2w+1. Lane -> row lane>>3, physical chunk lane&7, which
  // receives logical chunk (lane&7) ^ ((row>>1)&7).   Vᵀ tile: 8 pieces of 16 rows x 64 B; lane -> row lane>>2, physical
  // chunk lane&3 <- logical (lane&3) ^ ((row>>2)&3).
  const __amdgpu_buffer_rsrc_t rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(Kb), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcV = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(Vb), 0, -1, 0x00020000);
  int krow[2];
  uint32_t koff[2], voff[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    krow[p] = (wave * 2 + p) * 8 + (lane >> 3);
    koff[p] = (uint32_t)krow[p] * (uint32_t)ldqk + (uint32_t)((((lane & 7) ^ ((krow[p] >> 1) & 7))) << 4);
    const int vrow = (wave * 2 + p) * 16 + (lane >> 2);
    voff[p] = (uint32_t)vrow * (uint32_t)S64 + (uint32_t)((((lane & 3) ^ ((vrow >> 2) & 3))) << 4);
  }
  const int ktile_stride = BKV * ldqk;            // bytes between K tiles (host checks (S+64)·ldqk < 2^31)
  auto stage = [&](int slot, int tix, bool clamp) {
    const int kv0 = tix * BKV;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      uint32_t ko = koff[p];
      if (clamp) ko = (uint32_t)(min(kv0 + krow[p], S - 1) - kv0) * (uint32_t)ldqk + (uint32_t)((((lane & 7) ^ ((krow[p] >> 1) & 7))) << 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcK, LDS_PTR(smem + slot * SLOT_B + (wave * 2 + p) * 1024), 16, (int)ko, tix * ktile_stride, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcV, LDS_PTR(smem + slot * SLOT_B + KT_B + (wave * 2 + p) * 1024), 16, (int)voff[p], tix * BKV, 0, 0);
    }
  };

  // ---- per-lane LDS read addresses (bytes inside a slot)
  // K: key row 32kt + l31, logical chunks 4ks + 2hh and +1                    (+ kt * 4096)
  lds_cptr kp[2][2];
  {
    const int f = (l31 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int c = 0; c < 2; ++c) kp[ks][c] = (lds_cptr)smem + l31 * 128 + (((4 * ks + 2 * hh + c) ^ f) << 4);
  }
  // Vᵀ: d row 32dt + l31, logical chunks 2hh and 2hh+1                        (+ dt * 2048)
  lds_cptr vp[2];
  {
    const int f = (l31 >> 2) & 3;
#pragma unroll
    for (int c = 0; c < 2; ++c) vp[c] = (lds_cptr)smem + KT_B + l31 * 64 + (((2 * hh + c) ^ f) << 4);
  }

  f32x16 o_acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int ntiles = (S + BKV - 1) / BKV;

  auto tile = [&](auto slot_c, auto tail_c, int t) {
    constexpr int SLOT = decltype(slot_c)::value;
    constexpr bool TAIL = decltype(tail_c)::value;      // last <= 3 tiles: guarded / clamped staging, masked keys
    constexpr int SB = SLOT * SLOT_B;
    rt_dma_barrier();                          // tile t landed in every wave's rows; the other slot is free
    if constexpr (!TAIL) stage(SLOT ^ 1, t + 1, false);
    else if (t + 1 < ntiles) stage(SLOT ^ 1, t + 1, (t + 2) * BKV > S);

    // ---- Sᵀ = K·Qᵀ (x256: both operands carry the x16 prescale)
    f32x16 s_acc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s_acc[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const i32x8 kf = lds_read32(kp[ks][0] + SB + kt * 4096, kp[ks][1] + SB + kt * 4096);
        s_acc[kt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(kf, qf[ks], s_acc[kt], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
      }
    }
    if constexpr (TAIL) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * BKV + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= S) s_acc[kt][r] = -INFINITY;
        }
    }
    // ---- online softmax, log2 domain (see attention.hip)
    float mx = max3f(s_acc[0][0], s_acc[1][0], s_acc[0][1]);
    mx = max3f(mx, s_acc[1][1], s_acc[0][2]);
#pragma unroll
    for (int r = 3; r < 16; ++r) mx = max3f(mx, s_acc[0][r], s_acc[1][r - 1]);
    mx = max3f(mx, s_acc[1][15], s_acc[1][14]);
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * scale_log2;
    if (__any(mx - m_run > RESCALE_THR)) {
      asm volatile("" ::: "memory");
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
    }
    float psum = 0.f;
    const float m_sub = m_run - P_BIAS;
    i32x8 pf;                                   // byte j = 16kt + r of the lane's 32-byte operand
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        float p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          p[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kt][4 * r4 + e], scale_log2, -m_sub));
          psum += p[e];
        }
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(p[0], p[1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(p[2], p[3], w, true);
        pf[4 * kt + r4] = w;
      }
    l_run += psum;

    // ---- Oᵀ += Vᵀ·Pᵀ : one k-step of 64 keys per d block
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const i32x8 vf = lds_read32(vp[0] + SB + dt * 2048, vp[1] + SB + dt * 2048);
      o_acc[dt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vf, pf, o_acc[dt], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  const int nfull = S / BKV;
  stage(0, 0, BKV > S);
  int t = 0;
  for (; t + 2 < nfull; t += 2) {
    tile(S0{}, std::false_type{}, t);
    tile(S1{}, std::false_type{}, t + 1);
  }
  for (; t < ntiles; ++t) {
    if (t & 1) tile(S1{}, std::true_type{}, t);
    else tile(S0{}, std::true_type{}, t);
  }

  // ---- epilogue: O[q][d] = Oᵀ / l ; lane holds q = l31, d = 32dt + (r&3) + 8(r>>2) + 4hh
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + wave * 32 + l31;
  if (O8) {
    // e4m3 + E8M0 block scales (rt_attention_fp8_fwd_mx): a 32-column block of the output row = one dt; the lane holds 16 of its
    // columns, lane ^ 32 the other 16. Same rule as rt_quantize_mx_fp8 on the fp32 value o * inv; the next GEMM reads it as A.
    const bool valid = qrow < S;
    const int m = min(qrow, S - 1);
    uint8_t* orow = O8 + b * stride_ob8 + (int64_t)m * ldo8 + head * DH;
    uint32_t sbytes = 0;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      float am = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) am = fmaxf(am, fabsf(o_acc[dt][r]));
      am = rt_max_over_halves(am * inv);
      const int sb = rt_mx_scale_byte(am);
      const float f = inv * rt_mx_inv_scale(sb);
      sbytes |= (uint32_t)sb << (8 * dt);
      int w[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        w[g] = 0;
        w[g] = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(o_acc[dt][4 * g + 0] * f), sat448(o_acc[dt][4 * g + 1] * f), w[g], false);
        w[g] = __builtin_amdgcn_cvt_pk_fp8_f32(sat448(o_acc[dt][4 * g + 2] * f), sat448(o_acc[dt][4 * g + 3] * f), w[g], true);
      }
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        const auto x = __builtin_amdgcn_permlane32_swap((unsigned)w[g], (unsigned)w[g + 1], false, false);
        if (valid) *reinterpret_cast<u32x2*>(orow + dt * 32 + 8 * (g + hh)) = u32x2{x[0], x[1]};
      }
    }
    if (valid && hh == 0) {
      const int64_t o0 = rt_mx_scale_offset((int64_t)b * bsc_rows + m, bsc_k0 + head * DH, bsc_plane);      // block dt of the head: 4 bytes further each
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) bsc[o0 + 4 * dt] = (uint8_t)(sbytes >> (8 * dt));
    }
    return;
  }
  const bool wide = (ldo % 8 == 0) && (stride_ob % 8 == 0) && ((reinterpret_cast<uintptr_t>(O) & 15) == 0);
  if (wide) {
    rt_store_o_rows(O + b * stride_ob + (int64_t)min(qrow, S - 1) * ldo + head * DH, qrow < S, hh, o_acc, inv);
  } else if (qrow < S) {
    bf16_t* op = O + b * stride_ob + (int64_t)qrow * ldo + head * DH + 4 * hh;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        u32x2 w;
        w[0] = pack_bf16x2(o_acc[dt][4 * g + 0] * inv, o_acc[dt][4 * g + 1] * inv);
        w[1] = pack_bf16x2(o_acc[dt][4 * g + 2] * inv, o_acc[dt][4 * g + 3] * inv);
        *reinterpret_cast<u32x2*>(op + dt * 32 + 8 * g) = w;
      }
  }
}

}  // namespace

extern "C" int64_t rt_attention_fp8_vt_bytes(int32_t B, int32_t S, int32_t H) {
  if (B < 1 || S < 1 || H < 1) return 0;
  return (int64_t)B * H * DH * (((int64_t)S + BKV - 1) / BKV * BKV);
}

extern "C" int rt_attention_fp8_prep(const void* buf, int64_t ld, int64_t stride_b, int64_t q_off, int64_t k_off, int64_t v_off,
                                     const void* wq_txt, const void* wk_txt, const void* wq_img, const void* wk_img,
                                     const float* cosv, const float* sinv, void* qk8, void* vt8, int32_t B, int32_t S, int32_t T,
                                     int32_t H, float eps, void* stream) {
  if (!buf || !wq_img || !wk_img || !cosv || !sinv || !qk8 || !vt8 || B < 1 || S < 1 || H < 1 || T < 0 || T > S) return RT_E_BADARG;
  if (T > 0 && (!wq_txt || !wk_txt)) return RT_E_BADARG;
  if (!RT_ALIGNED(buf, 16) || ld % 8 || stride_b % 8 || q_off % 8 || k_off % 8 || v_off % 8 || !RT_ALIGNED(cosv, 16) ||
      !RT_ALIGNED(sinv, 16) || !RT_ALIGNED(wq_img, 16) || !RT_ALIGNED(wk_img, 16) || !RT_ALIGNED(qk8, 16) || !RT_ALIGNED(vt8, 16))
    return RT_E_ALIGN;
  const int64_t ngrp = (int64_t)B * S * H;
  const int64_t blocks = (ngrp + 15) / 16;
  if (blocks > 0x7fffffff) return RT_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(prep_qk_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)buf, ld, stride_b, q_off, k_off,
                     (const bf16_t*)wq_txt, (const bf16_t*)wk_txt, (const bf16_t*)wq_img, (const bf16_t*)wk_img, cosv, sinv,
                     (uint8_t*)qk8, B, S, T, H, eps);
  const int S64 = (S + BKV - 1) / BKV * BKV;
  hipLaunchKernelGGL(prep_vt_kernel, dim3(S64 / BKV, H, B), dim3(256), 0, st, (const bf16_t*)buf, ld, stride_b, v_off, (uint8_t*)vt8, S, S64, H);
  return rt_hip_status();
}

extern "C" int rt_attention_fp8_fwd(const void* qk8, const void* vt8, void* o, int64_t ldo, int64_t stride_ob, int32_t B, int32_t S,
                                    int32_t H, float scale, void* stream) {
  if (!qk8 || !vt8 || !o || B < 1 || S < 1 || H < 1) return RT_E_BADARG;
  if (!RT_ALIGNED(qk8, 16) || !RT_ALIGNED(vt8, 16) || !RT_ALIGNED(o, 8) || ldo % 4 || stride_ob % 4) return RT_E_ALIGN;
  if (ldo < (int64_t)H * DH) return RT_E_SHAPE;
  if ((int64_t)(S + BKV) * 2 * H * DH >= (int64_t)1 << 31) return RT_E_SHAPE;      // per-tile byte offsets are 32-bit
  const int S64 = (S + BKV - 1) / BKV * BKV;
  const int NI = H * ((S + BQ - 1) / BQ);
  const dim3 grid(8 * ((NI + 7) / 8), B);
  hipLaunchKernelGGL(attention_fp8_kernel, grid, dim3(THREADS), 0, (hipStream_t)stream, (const uint8_t*)qk8, (const uint8_t*)vt8,
                     (bf16_t*)o, ldo, stride_ob, S, S64, H, scale * 1.4426950408889634f / (QK_PRESCALE * QK_PRESCALE), (uint8_t*)nullptr,
                     (int64_t)0, (int64_t)0, (uint8_t*)nullptr, (int64_t)0, (int64_t)0, 0);
  return rt_hip_status();
}

extern "C" int rt_attention_fp8_fwd_mx(const void* qk8, const void* vt8, void* o8, int64_t ldo8, int64_t stride_ob8, uint8_t* bscale,
                                       int64_t plane, int64_t bscale_rows, int32_t bscale_k0, int32_t B, int32_t S, int32_t H, float scale,
                                       void* stream) {
  if (!qk8 || !vt8 || !o8 || !bscale || B < 1 || S < 1 || H < 1) return RT_E_BADARG;
  if (!RT_ALIGNED(qk8, 16) || !RT_ALIGNED(vt8, 16) || !RT_ALIGNED(o8, 8) || ldo8 % 8 || stride_ob8 % 8 || !RT_ALIGNED(bscale, 16) || plane % 2048)
    return RT_E_ALIGN;
  if (ldo8 < (int64_t)H * DH || bscale_k0 < 0 || bscale_k0 % 128 != 0 || bscale_rows % 64 != 0 ||
      plane < ((((int64_t)(B - 1) * bscale_rows + S) + 63) / 64) * 2048)
    return RT_E_SHAPE;
  if ((int64_t)(S + BKV) * 2 * H * DH >= (int64_t)1 << 31) return RT_E_SHAPE;
  const int S64 = (S + BKV - 1) / BKV * BKV;
  const int NI = H * ((S + BQ - 1) / BQ);
  const dim3 grid(8 * ((NI + 7) / 8), B);
  hipLaunchKernelGGL(attention_fp8_kernel, grid, dim3(THREADS), 0, (hipStream_t)stream, (const uint8_t*)qk8, (const uint8_t*)vt8,
                     (bf16_t*)nullptr, (int64_t)0, (int64_t)0, S, S64, H, scale * 1.4426950408889634f / (QK_PRESCALE * QK_PRESCALE), (uint8_t*)o8,
                     ldo8, stride_ob8, bscale, plane, bscale_rows, (int)bscale_k0);
  return rt_hip_status();
}
