// rt_attention_fwd — joint text+image attention of the MMDiT blocks: softmax(Q·Kᵀ·scale)·V, non-causal,
// no mask, head dim 128, bf16 in/out, fp32 scores/statistics/accumulators. Flash-style: the S×S score
// matrix never leaves the CU. Reference: torch SDPA as called by diffusers' FluxAttnProcessor (SURVEY.md
// Appendix A.1 step 6 / A.2), reached from controlnet_flux.py:343-348,376-380 and PIPE:1092.
//
// Workgroup = 4 waves = 128 query rows of one (batch, head); wave w owns query rows 32w..32w+31 for the whole
// key sweep. Keys/values arrive in 64-row tiles by LDS-DMA into a 2-deep ring (K 16 KiB + V 16 KiB per slot).
//
// Both products run on v_mfma_f32_32x32x16_bf16 with the QUERY index on the lane:
//   Sᵀ[key][q]  = K · Qᵀ      A = K rows from LDS (ds_read_b128), B = Q rows held in registers
//   Oᵀ[d][q]   += Vᵀ · Pᵀ     A = Vᵀ via ds_read_b64_tr_b16 (hardware transpose), B = the Sᵀ accumulator
//                             registers themselves, exponentiated and packed to bf16 (no LDS round trip)
// so the softmax max/sum of a query row are lane-local apart from one lane<->lane+32 exchange, and the
// per-row rescale of O is a per-lane scalar multiply.
//
// LDS image of a K or V tile: 64 rows × 256 B; 16-B chunk c of row r lives at 256·r + 16·(c ^ f(r)),
// f(r) = ((r&3)<<2) | ((r>>2)&3): conflict-free for the row reads (K) and the transposed reads (V).
// LDS-DMA writes lane-linear, so the XOR is applied to each lane's source address.
#include "rt_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int DH = 128;
constexpr int BQ = 128;       // query rows per workgroup
constexpr int BKV = 64;       // keys per tile
constexpr int TILE_B = BKV * DH * 2;   // 16 KiB
constexpr int ATT_THREADS = 256;
constexpr float RESCALE_THR = 6.0f;   // log2 units: P <= 64 between rescales

__device__ __forceinline__ int swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// 3-input max. Plain fmaxf so the compiler inserts the MFMA->VALU wait states itself: an inline-asm v_max3 here read the
// accumulators before the MFMA had retired them (run-to-run differences in the running max; csrc/Makefile builds this file
// with -fno-honor-nans so no canonicalising v_max is emitted in front and the pair folds to one v_max3_f32).
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

typedef const __attribute__((address_space(3))) char* lds_cptr;
__device__ __forceinline__ s16x4 tr_read(lds_cptr p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

__global__ __launch_bounds__(ATT_THREADS, 2) void attention_fwd_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, bf16_t* O,
    int64_t ld, int64_t stride_b, int64_t ldo, int64_t stride_ob, int S, int H, float scale_log2) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];   // [slot][K|V]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * BQ;

  const bf16_t* Qb = Q + b * stride_b + head * DH;
  const bf16_t* Kb = K + b * stride_b + head * DH;
  const bf16_t* Vb = V + b * stride_b + head * DH;

  // ---- Q fragments (B operand of Sᵀ = K·Qᵀ): lane holds Q[q][16ks + 8hh .. +7]
  bf16x8 qf[8];
  {
    const int qrow = min(q0 + wave * 32 + l31, S - 1);
    const bf16_t* qp = Qb + (int64_t)qrow * ld + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }

  // ---- staging: wave w stages pieces 4w..4w+3 (4 rows each) of K and of V; K and V share row/chunk offsets.
  const int srow = lane >> 4;                 // row inside a piece
  const int spc = lane & 15;                  // physical chunk written by this lane
  int srow_t[4];                              // row inside the tile per piece
  uint32_t soff[4];                           // BYTE offset of (row, logical chunk) from the tile's first row: unsigned 32-bit,
                                              // so the DMA address is (uniform 64-bit base in SGPRs) + (VGPR offset) with no
                                              // per-tile 64-bit vector address arithmetic
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    srow_t[p] = (wave * 4 + p) * 4 + srow;
    soff[p] = ((uint32_t)srow_t[p] * (uint32_t)ld + (uint32_t)((spc ^ swz(srow_t[p])) << 3)) * 2u;
  }
  auto stage = [&](int slot, int kv0, bool clamp) {
    char* kb = smem + slot * 2 * TILE_B;
    const char* kt0 = reinterpret_cast<const char*>(Kb + (int64_t)kv0 * ld);      // wave-uniform tile bases
    const char* vt0 = reinterpret_cast<const char*>(Vb + (int64_t)kv0 * ld);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      uint32_t off = soff[p];
      if (clamp)   // ragged last tile: rows past the end re-read row S-1 (masked later)
        off = ((uint32_t)(min(kv0 + srow_t[p], S - 1) - kv0) * (uint32_t)ld + (uint32_t)((spc ^ swz(srow_t[p])) << 3)) * 2u;
      __builtin_amdgcn_global_load_lds(GLB_PTR(kt0 + off), LDS_PTR(kb + (wave * 4 + p) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(vt0 + off), LDS_PTR(kb + TILE_B + (wave * 4 + p) * 1024), 16, 0, 0);
    }
  };

  // ---- per-lane LDS read offsets (bytes inside a tile); everything that varies with (kt, ks, s, dt) statically is an
  //      immediate, so the loop carries no address arithmetic.
  // K row read: row = 32kt + l31, chunk = 2ks + hh  ->  l31*256 + ((2ks+hh) ^ swz(l31))*16  (+ kt*8192)
  lds_cptr kp[8];                             // 32-bit LDS addresses; slot / kt offsets are immediates at the use site
  {
    const int ksw = swz(l31);                 // swz depends on row & 15 only
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kp[ks] = (lds_cptr)smem + (l31 * 256 + (((2 * ks + hh) ^ ksw) << 4));
  }
  // V transposed read: 16-lane group g = lane>>4 (hh = g>>1, d half g&1); lane 4q+p of the group supplies row r0+q,
  // chunk c0+(p>>1), 8-byte half p&1. Rows: first block 4hh+tq, second block +8 (per k-step: + 32kt + 16s).
  const int tq = (lane >> 2) & 3, tp = lane & 3;
  const int tg1 = (lane >> 4) & 1;
  lds_cptr vp[2][4];
  {
    const int cl = tg1 * 2 + (tp >> 1);
    const int rl0 = 4 * hh + tq, rl1 = rl0 + 8;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      vp[0][dt] = (lds_cptr)smem + TILE_B + rl0 * 256 + (((dt * 4 + cl) ^ swz(rl0)) << 4) + 8 * (tp & 1);
      vp[1][dt] = (lds_cptr)smem + TILE_B + rl1 * 256 + (((dt * 4 + cl) ^ swz(rl1)) << 4) + 8 * (tp & 1);
    }
  }

  f32x16 o_acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;     // running max (log2 domain, may lag the true max by <= RESCALE_THR) and row sum

  const int ntiles = (S + BKV - 1) / BKV;

  auto tile = [&](auto slot_c, auto tail_c, int t) {
    constexpr int SLOT = decltype(slot_c)::value;
    // TAIL = false: steady state — this tile and the one it stages are full, nothing is checked, the body is straight-line
    // code (no address arithmetic, no branches but the rare rescale). TAIL = true: the last <= 3 tiles of a row — staging is
    // guarded and clamped, keys >= S are masked.
    constexpr bool TAIL = decltype(tail_c)::value;
    constexpr bool RAGGED = TAIL;
    constexpr int SB = SLOT * 2 * TILE_B;
    rt_dma_barrier();                          // tile t landed (every wave's vmcnt(0), then the barrier); the other slot is free
    if constexpr (!TAIL) stage(SLOT ^ 1, (t + 1) * BKV, false);
    else if (t + 1 < ntiles) stage(SLOT ^ 1, (t + 1) * BKV, (t + 2) * BKV > S);

    // ---- Sᵀ = K·Qᵀ : two 32-key tiles
    f32x16 s_acc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s_acc[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const bf16x8 kf = *(const __attribute__((address_space(3))) bf16x8*)(kp[ks] + SB + kt * 8192);
        s_acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s_acc[kt], 0, 0, 0);
      }
    }
    // key of s_acc[kt][r]: 64t + 32kt + (r&3) + 8(r>>2) + 4hh
    if constexpr (RAGGED) {                    // mask keys >= S
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * BKV + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
          if (key >= S) s_acc[kt][r] = -INFINITY;
        }
    }

    // ---- online softmax in the log2 domain; the row's other 32 keys live in lane ^ 32.
    float mx = max3f(s_acc[0][0], s_acc[1][0], s_acc[0][1]);
    mx = max3f(mx, s_acc[1][1], s_acc[0][2]);
#pragma unroll
    for (int r = 3; r < 16; ++r) mx = max3f(mx, s_acc[0][r], s_acc[1][r - 1]);
    mx = max3f(mx, s_acc[1][15], s_acc[1][14]);
    mx = fmaxf(mx, __shfl_xor(mx, 32)) * scale_log2;
    // Deferred rescale: O and l are only rescaled when some row's max grew by more than RESCALE_THR (log2 units) over
    // the max it is currently normalised with; otherwise P = exp2(s - m_run) <= 2^THR, harmless in fp32 accumulators
    // and (being a relative format) in the bf16 P operand. The decision precedes every use of this tile's P.
    if (__any(mx - m_run > RESCALE_THR)) {
      asm volatile("" ::: "memory");           // keep this rare block a real branch (not if-converted into 64 multiplies)
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: exp2(-inf) = 0 on zeroed state
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
    }
    float psum = 0.f;
    bf16x8 pf[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kt][8 * s2 + j], scale_log2, -m_run));
          psum += p;
          pf[kt][s2][j] = (__bf16)p;
        }
    l_run += psum;

    // ---- Oᵀ += Vᵀ·Pᵀ : element j of lane-half hh of k-step (kt,s) is key 32kt + 16s + 8(j>>2) + 4hh + (j&3)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const s16x4 lo = tr_read(vp[0][dt] + SB + kt * 8192 + s2 * 4096);
          const s16x4 hi = tr_read(vp[1][dt] + SB + kt * 8192 + s2 * 4096);
          const bf16x8 vf = __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi),
                                                    0, 1, 2, 3, 4, 5, 6, 7);
          o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[kt][s2], o_acc[dt], 0, 0, 0);
        }
      }
  };

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  const int nfull = S / BKV;                   // tiles with all 64 keys valid
  stage(0, 0, BKV > S);
  int t = 0;
  for (; t + 2 < nfull; t += 2) {              // tiles t, t+1 and the tiles they stage (t+1, t+2) are all full
    tile(S0{}, std::false_type{}, t);
    tile(S1{}, std::false_type{}, t + 1);
  }
  for (; t < ntiles; ++t) {                    // at most 3 tiles
    if (t & 1) tile(S1{}, std::true_type{}, t);
    else tile(S0{}, std::true_type{}, t);
  }

  // ---- epilogue: O[q][d] = Oᵀ / l ; lane holds q = l31, d = 32dt + (r&3) + 8(r>>2) + 4hh
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + wave * 32 + l31;
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


// ---------------------------------------------------------------------------------------------------
// Software-pipelined variant (opt-in: RT_ATTN_VARIANT=pipe). Same tiling, layouts and arithmetic as attention_fwd_kernel; what changes is the
// order of issue inside a wave, so that the MFMA pipe and the vector ALU work at the same time instead of in turns:
//   iteration t:   Sᵀ(t+1) = K(t+1)·Qᵀ   (16 MFMAs)   ∥  exp2 / row-sum / bf16 pack of three quarters of tile t
//                  Oᵀ += Vᵀ(t)·Pᵀ(t)     (8 MFMAs)    ∥  the last quarter of tile t
//                                         (8 MFMAs)    ∥  row max of tile t+1, rescale decision
// Each MFMA is followed by its slice of vector work and a scheduling fence, so the emitted order is the written order.
// K is staged two tiles ahead and V one tile ahead (2 + 2 ring slots, one barrier per tile); LDS fragments are read two
// MFMAs ahead of their use.
// ---------------------------------------------------------------------------------------------------
constexpr int K_SLOT_B = TILE_B;              // K slots at 0 and TILE_B, V slots at 2·TILE_B and 3·TILE_B
constexpr int V_BASE_B = 2 * TILE_B;
#ifndef RT_ATTN_PF
#define RT_ATTN_PF 2
#endif
constexpr int PF = RT_ATTN_PF;                 // LDS fragments are read PF MFMAs ahead of their use

__global__ __launch_bounds__(ATT_THREADS, 2) void attention_fwd_pipe_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, bf16_t* O,
    int64_t ld, int64_t stride_b, int64_t ldo, int64_t stride_ob, int S, int H, float scale_log2) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];   // [K0 | K1 | V0 | V1]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * BQ;

  const bf16_t* Qb = Q + b * stride_b + head * DH;
  const bf16_t* Kb = K + b * stride_b + head * DH;
  const bf16_t* Vb = V + b * stride_b + head * DH;

  bf16x8 qf[8];
  {
    const int qrow = min(q0 + wave * 32 + l31, S - 1);
    const bf16_t* qp = Qb + (int64_t)qrow * ld + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }

  // ---- staging (see attention_fwd_kernel): wave w stages pieces 4w..4w+3 (4 rows each) of a tile
  const int srow = lane >> 4, spc = lane & 15;
  int srow_t[4];
  uint32_t soff[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    srow_t[p] = (wave * 4 + p) * 4 + srow;
    soff[p] = ((uint32_t)srow_t[p] * (uint32_t)ld + (uint32_t)((spc ^ swz(srow_t[p])) << 3)) * 2u;
  }
  // DMA by buffer_load ... lds: descriptor (4 SGPRs) + loop-invariant VGPR offset + per-tile SGPR offset, so staging costs no
  // vector ALU work and no 64-bit address registers. num_records = 2^32-1: rows are clamped here, not by the range check.
  const __amdgpu_buffer_rsrc_t rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Kb), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Vb), 0, -1, 0x00020000);
  const int tile_stride_b = BKV * (int)ld * 2;           // bytes between tiles (host checks S*ld*2 < 2^31)
  auto stage_one = [&](const __amdgpu_buffer_rsrc_t& rs, int lds_base, int tix, bool clamp) {
    const int kv0 = tix * BKV;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      uint32_t off = soff[p];
      if (clamp)
        off = ((uint32_t)(min(kv0 + srow_t[p], S - 1) - kv0) * (uint32_t)ld + (uint32_t)((spc ^ swz(srow_t[p])) << 3)) * 2u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(smem + lds_base + (wave * 4 + p) * 1024), 16, (int)off, tix * tile_stride_b, 0, 0);
    }
  };

  lds_cptr kp[8];
  {
    const int ksw = swz(l31);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kp[ks] = (lds_cptr)smem + (l31 * 256 + (((2 * ks + hh) ^ ksw) << 4));
  }
  const int tq = (lane >> 2) & 3, tp = lane & 3;
  const int tg1 = (lane >> 4) & 1;
  lds_cptr vp[2][4];
  {
    const int cl = tg1 * 2 + (tp >> 1);
    const int rl0 = 4 * hh + tq, rl1 = rl0 + 8;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      vp[0][dt] = (lds_cptr)smem + V_BASE_B + rl0 * 256 + (((dt * 4 + cl) ^ swz(rl0)) << 4) + 8 * (tp & 1);
      vp[1][dt] = (lds_cptr)smem + V_BASE_B + rl1 * 256 + (((dt * 4 + cl) ^ swz(rl1)) << 4) + 8 * (tp & 1);
    }
  }

  f32x16 o_acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 sc[2][2];                             // score tiles by tile parity: [t & 1][32-key half]

  const int ntiles = (S + BKV - 1) / BKV;
  const int nfull = S / BKV;

  // K fragment i (= 32-key half i>>3, k-step i&7) of the K slot at byte offset kb
  auto k_read = [&](int i, int kb) -> bf16x8 {
    return *(const __attribute__((address_space(3))) bf16x8*)(kp[i & 7] + kb + (i >> 3) * 8192);
  };
  // Vᵀ fragment n (= k-step n>>2 of the tile (16 keys each), d block n&3) of the V slot at byte offset vb
  auto v_read = [&](int n, int vb) -> bf16x8 {
    const int dt = n & 3, step = n >> 2;
    const s16x4 lo = tr_read(vp[0][dt] + vb + step * 4096);
    const s16x4 hi = tr_read(vp[1][dt] + vb + step * 4096);
    return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto mask_tile = [&](f32x16 (&sx)[2], int t) {   // keys >= S of tile t
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t * BKV + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (key >= S) sx[kt][r] = -INFINITY;
      }
  };
  // row max of a score tile (two chains), both 32-key halves of the row combined across lane <-> lane+32
  auto row_max_finish = [&](float ma, float mb) -> float {
    float mx = fmaxf(ma, mb);
    const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mx), __builtin_bit_cast(unsigned, mx), false, false);
    return fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1])) * scale_log2;
  };
  // deferred rescale (see attention_fwd_kernel): only when some row's max outgrew the one it is normalised with
  auto decide = [&](float mx) {
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
  };

  bf16x8 pf[2][2];
  float psum = 0.f;
  // softmax numerator of element e (0..31) of the current tile: half e>>4, register e&15 -> operand pf[e>>4][(e>>3)&1][e&7]
  auto soft = [&](const f32x16 (&cur)[2], int e) {
    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(cur[e >> 4][e & 15], scale_log2, -m_run));
    psum += p;
    pf[e >> 4][(e >> 3) & 1][e & 7] = (__bf16)p;
  };

  auto iteration = [&](auto par_c, auto tail_c, int t) {
    constexpr int PAR = decltype(par_c)::value;          // t & 1
    constexpr bool TAIL = decltype(tail_c)::value;       // last iterations: guarded staging, masked keys
    constexpr int KB_NEXT = (PAR ^ 1) * K_SLOT_B;        // K(t+1)
    constexpr int VB_CUR = PAR * TILE_B;                 // V(t), relative to V_BASE_B (folded into vp)
    f32x16 (&cur)[2] = sc[PAR];
    f32x16 (&nxt)[2] = sc[PAR ^ 1];
    const bool has_next = TAIL ? (t + 1 < ntiles) : true;

    rt_dma_barrier();    // K(t+1), V(t) landed (every wave's vmcnt(0), then the barrier); K slot PAR and V slot PAR^1 are free
    if constexpr (!TAIL) {
      stage_one(rsrcK, PAR * K_SLOT_B, t + 2, false);
      stage_one(rsrcV, V_BASE_B + (PAR ^ 1) * TILE_B, t + 1, false);
    } else {
      if (t + 2 < ntiles) stage_one(rsrcK, PAR * K_SLOT_B, t + 2, (t + 3) * BKV > S);
      if (t + 1 < ntiles) stage_one(rsrcV, V_BASE_B + (PAR ^ 1) * TILE_B, t + 1, (t + 2) * BKV > S);
    }
    psum = 0.f;

    // ---- phase 1: Sᵀ(t+1) ∥ softmax numerators 0..23 of tile t
    if (has_next) {
      bf16x8 kf[PF + 1];
#pragma unroll
      for (int i = 0; i < PF; ++i) kf[i] = k_read(i, KB_NEXT);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i + PF < 16) kf[(i + PF) % (PF + 1)] = k_read(i + PF, KB_NEXT);
        const int kt = i >> 3, ks = i & 7;
        if (ks == 0) {
          f32x16 z;
#pragma unroll
          for (int r = 0; r < 16; ++r) z[r] = 0.f;
          nxt[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i % (PF + 1)], qf[ks], z, 0, 0, 0);
        } else {
          nxt[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i % (PF + 1)], qf[ks], nxt[kt], 0, 0, 0);
        }
#pragma unroll
        for (int e = (3 * i) / 2; e < (3 * (i + 1)) / 2; ++e) soft(cur, e);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 24; ++e) soft(cur, e);
    }

    // ---- phase 2: Oᵀ += Vᵀ(t)·Pᵀ(t) ∥ numerators 24..31, then ∥ row max of tile t+1
    if (TAIL && has_next) mask_tile(nxt, t + 1);
    float ma = -INFINITY, mb = -INFINITY;
    {
      bf16x8 vf[PF + 1];
#pragma unroll
      for (int n = 0; n < PF; ++n) vf[n] = v_read(n, VB_CUR);
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        if (n + PF < 16) vf[(n + PF) % (PF + 1)] = v_read(n + PF, VB_CUR);
        const int kt = n >> 3, s2 = (n >> 2) & 1, dt = n & 3;
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[n % (PF + 1)], pf[kt][s2], o_acc[dt], 0, 0, 0);
        if (n < 8) {
          soft(cur, 24 + n);
        } else if (has_next) {
          const int g = n - 8;           // values 4g..4g+3 of the 32: two per chain
          ma = max3f(ma, nxt[0][2 * g], nxt[0][2 * g + 1]);
          mb = max3f(mb, nxt[1][2 * g], nxt[1][2 * g + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    l_run += psum;
    if (has_next) decide(row_max_finish(ma, mb));
  };

  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  // ---- prologue: K(0), V(0), K(1) in flight; Sᵀ(0) and its row max computed alone
  stage_one(rsrcK, 0, 0, BKV > S);
  stage_one(rsrcV, V_BASE_B, 0, BKV > S);
  if (ntiles > 1) stage_one(rsrcK, K_SLOT_B, 1, 2 * BKV > S);
  rt_dma_barrier();
  {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[0][kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) sc[0][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_read(kt * 8 + ks, 0), qf[ks], sc[0][kt], 0, 0, 0);
    }
    if (BKV > S) mask_tile(sc[0], 0);
    float ma = -INFINITY, mb = -INFINITY;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      ma = max3f(ma, sc[0][0][2 * g], sc[0][0][2 * g + 1]);
      mb = max3f(mb, sc[0][1][2 * g], sc[0][1][2 * g + 1]);
    }
    decide(row_max_finish(ma, mb));
  }
  int t = 0;
  for (; t + 3 < nfull; t += 2) {              // iterations t, t+1: tiles up to t+3 are full
    iteration(P0{}, std::false_type{}, t);
    iteration(P1{}, std::false_type{}, t + 1);
  }
  for (; t < ntiles; ++t) {                    // at most 4 iterations
    if (t & 1) iteration(P1{}, std::true_type{}, t);
    else iteration(P0{}, std::true_type{}, t);
  }

  // ---- epilogue: O[q][d] = Oᵀ / l ; lane holds q = l31, d = 32dt + (r&3) + 8(r>>2) + 4hh
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + wave * 32 + l31;
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

extern "C" int rt_attention_fwd(const void* q, const void* k, const void* v, void* o, int64_t ld, int64_t stride_b,
                                int64_t ldo, int64_t stride_ob, int32_t B, int32_t S, int32_t H, float scale,
                                void* stream) {
  if (!q || !k || !v || !o || B < 1 || S < 1 || H < 1) return RT_E_BADARG;
  if (!RT_ALIGNED(q, 16) || !RT_ALIGNED(k, 16) || !RT_ALIGNED(v, 16) || !RT_ALIGNED(o, 8) || ld % 8 || stride_b % 8 ||
      ldo % 4 || stride_ob % 4)
    return RT_E_ALIGN;
  if (ld < (int64_t)H * DH || ldo < (int64_t)H * DH) return RT_E_SHAPE;
  if ((int64_t)(S + BKV) * ld * 2 >= (int64_t)1 << 31) return RT_E_SHAPE;      // per-tile byte offsets are 32-bit
  const dim3 grid((S + BQ - 1) / BQ, H, B);
  // RT_ATTN_VARIANT=pipe selects the software-pipelined schedule: +4 % in isolation at S >= 4608, equal inside the model
  // (2.203 vs 2.206 s/image) and slower at S = 768, so the plain kernel stays the default (attention anatomy).
  static int variant = -1;
  if (variant < 0) {
    const char* e = getenv("RT_ATTN_VARIANT");
    variant = (e && e[0] == 'p') ? 1 : 0;
  }
  if (variant == 0)
    hipLaunchKernelGGL(attention_fwd_kernel, grid, dim3(ATT_THREADS), 0, (hipStream_t)stream, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ld, stride_b, ldo, stride_ob, S, H,
                       scale * 1.4426950408889634f);
  else
    hipLaunchKernelGGL(attention_fwd_pipe_kernel, grid, dim3(ATT_THREADS), 0, (hipStream_t)stream, (const bf16_t*)q,
                       (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ld, stride_b, ldo, stride_ob, S, H,
                       scale * 1.4426950408889634f);
  return rt_hip_status();
}
