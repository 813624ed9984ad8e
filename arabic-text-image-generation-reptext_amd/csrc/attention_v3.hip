// attention_v3 — the joint attention of the MMDiT blocks (same math and interface as csrc/attention.hip: softmax(Q·Kᵀ·scale)·V,
// head dim 128, bf16 in/out; torch SDPA as reached from controlnet_flux.py:343-348,376-380 and PIPE:1092)
// restructured for ONE wave per SIMD with the whole 512-entry register file:
//
//   * a workgroup = 4 waves = 256 query rows of one (batch, head); a wave owns 64 rows = two 32-row query blocks (a, b) that
//     SHARE every K and V fragment it reads from LDS: half the LDS bytes per MFMA of the 32-rows-per-wave kernel;
//   * Oᵀ (2 x 4 accumulators of 32x32: 128 registers) and the Q fragments (64 registers) live in the ACCUMULATOR half of the
//     register file for the whole item, the scores, numerators and K / Vᵀ fragments in the architectural half. hipcc decides
//     per FUNCTION whether MFMA results go to AGPRs or VGPRs (at a 512-register budget: all to AGPRs, and it then copies every
//     score back with v_accvgpr_read — 576 copies per tile in a trial build), so the MFMAs here are inline asm with the register
//     class of each operand spelled out: scores "+v" (they feed the VALU), Oᵀ "+a", Q "a";
//   * the tile loop is software-pipelined by 32-key halves over four groups of 16 MFMAs, each MFMA followed by its share of the
//     vector work (one softmax element = fma, exp2, add, half a cvt_pk; one running-max step; an LDS read; now and then an
//     LDS-DMA piece), in source order with a scheduling fence per MFMA gap — with a single wave per SIMD nothing else hides
//     vector work behind the matrix pipe:
//         G1  Sᵀ(k0,t)  = K·Qᵀ      ∥ numerators 16..31 of k1(t-1), Vᵀ(k1,t-1) fragments, 4 DMA pieces of tile t+1
//         G2  Oᵀ += Vᵀ·Pᵀ (k1,t-1)   ∥ numerators 0..15 of k0(t) (speculative, against the running max), row max of k0(t), K(k1,t)
//         G3  Sᵀ(k1,t)              ∥ max exchange + decision for k0(t), numerators 16..31 of k0(t), Vᵀ(k0,t) fragments
//         G4  Oᵀ += Vᵀ·Pᵀ (k0,t)     ∥ numerators 0..15 of k1(t) (speculative), row max of k1(t), K(k0,t+1), 4 DMA pieces of t+2
//     The numerators of a half are started against the running max m before that half's own maximum is known; the decision
//     (two gaps into the next group) only acts when some row's maximum outgrew m by more than 2^6: it then rescales O and l and
//     recomputes the 16 speculative numerators (rare after the first tiles). Every numerator that reaches an MFMA was formed
//     against the maximum O is normalised with.
//   * K/V tiles of 64 keys arrive by LDS-DMA (asm-issued: see rt_dma16_asm) into a 3-deep ring, one barrier per tile: the barrier
//     in front of G4(t) publishes tile t+1 (copied a whole tile earlier) and frees the stage of tile t-1 for tile t+2.
//
// Applies when S is a multiple of 256 (every shape of the pipelines at 1024² / 1536² / 256²); other shapes run attention.hip.
// Default for S >= 1536 (rt_attention_variant / RT_ATTN_V3 = 0 turns it off, 2 forces it wherever S % 256 == 0). Measured on MI355X
// against attention.hip, interleaved in one process: S = 4608 x 24 heads 247 vs 264 us (-6.6 %), 4096 x 32 heads 244 vs 265,
// batch 4 -5 %, S = 9728 986 vs 1065 us (1179 TFLOP/s); S = 768 25 vs 19 us (left to attention.hip). DESIGN.md §5 has the
// ablation table (MFMA-only floor of this structure at S = 4096 x 32: 183 us = the chip holding ~1.6 GHz under back-to-back MFMAs).
#include "rt_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int DH = 128;
constexpr int BQ3 = 256;             // query rows per work item
constexpr int BKV = 64;
constexpr int TILE_B = BKV * DH * 2; // 16 KiB (one K or V tile)
constexpr int STAGE_B = 2 * TILE_B;  // K | V
constexpr int NSTAGE = 3;
constexpr int V3_THREADS = 256;
constexpr float RESCALE_THR = 6.0f;

__device__ __forceinline__ int swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// ---- the accumulator half of the register file is OWNED by the asm statements of this file -------------------------------------
// O fragment f = d block * 2 + query block lives in a[16f : 16f+15], Q fragment i = query block * 8 + k-step in a[128+4i : 131+4i]
// (attention_v3_regs.h). The statements name those registers literally and list them as clobbers — which is also what makes the
// kernel descriptor allocate them. hipcc must never put a value of its own there. A statement that only READS them (the score
// MFMAs read Q) has no way to say so, and hipcc did park two long-lived values in a128 / a130 across the whole tile loop in one
// build: EVERY statement below therefore lists ALL of a0..a191 as clobbered, so no value of the compiler's can be live in them
// across any MFMA. This file is built with -mllvm -amdgpu-spill-vgpr-to-agpr=0 (no VGPR spills into AGPRs) and contains no
// "a"-constrained operand; `make audit_v3` (tools/audit_v3_asm.py, also run by tests/test_capi_and_layout.py) rejects a build
// with a VGPR spill, scratch, or any compiler-generated v_accvgpr_* naming a0..a191.
#include "attention_v3_regs.h"
#define V3_EACH8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define V3_EACH10(M) V3_EACH8(M) M(8) M(9)
#define V3_EACH16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
// Oᵀ fragment f += Vᵀ fragment · numerators
__device__ __forceinline__ void mfma_o(int f, const bf16x8& v, const bf16x8& p) {
  switch (f) {
#define V3_C(F) case F: asm volatile("v_mfma_f32_32x32x16_bf16 " V3_OREG_##F ", %0, %1, " V3_OREG_##F ::"v"(v), "v"(p) : V3_ALL_CL); break;
    V3_EACH8(V3_C)
#undef V3_C
  }
}
// row sums on the matrix pipe: L fragment (8 + query block) += ones(32 x 16) · numerators — every row of the result is the sum of
// the k-step's 16 keys per query, over BOTH lane halves (no cross-lane add later), from the same bf16 numerators O is built from.
// One MFMA per k-step and query block (8 per tile, +12.5 % matrix work) instead of 64 v_add_f32 per lane and tile: with one wave
// per SIMD the loop is bound by instruction ISSUE, and an MFMA costs 8 issue cycles against 4 per add.
__device__ __forceinline__ void mfma_l(int qb, const bf16x8& p) {
  if (qb) asm volatile("v_mfma_f32_32x32x16_bf16 " V3_OREG_9 ", " V3_ONES_REG ", %0, " V3_OREG_9 ::"v"(p) : V3_ALL_CL);
  else asm volatile("v_mfma_f32_32x32x16_bf16 " V3_OREG_8 ", " V3_ONES_REG ", %0, " V3_OREG_8 ::"v"(p) : V3_ALL_CL);
}
__device__ __forceinline__ void ones_write() {     // eight bf16 1.0 in a[224:227]
  asm volatile("v_accvgpr_write_b32 a224, %0\n\tv_accvgpr_write_b32 a225, %0\n\tv_accvgpr_write_b32 a226, %0\n\tv_accvgpr_write_b32 a227, %0" ::"v"(0x3F803F80u) : V3_ALL_CL);
}
// scores (architectural registers: they feed the VALU) = / += K fragment · Q fragment i
__device__ __forceinline__ void mfma_s0(f32x16& c, const bf16x8& k, int i) {
  switch (i) {
#define V3_C(I) case I: asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, " V3_QREG_##I ", 0" : "=&v"(c) : "v"(k) : V3_ALL_CL); break;
    V3_EACH16(V3_C)
#undef V3_C
  }
}
__device__ __forceinline__ void mfma_s(f32x16& c, const bf16x8& k, int i) {
  switch (i) {
#define V3_C(I) case I: asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, " V3_QREG_##I ", %0" : "+v"(c) : "v"(k) : V3_ALL_CL); break;
    V3_EACH16(V3_C)
#undef V3_C
  }
}
__device__ __forceinline__ void q_write(int i, const bf16x8& q) {
  const u32x4 w = __builtin_bit_cast(u32x4, q);
  switch (i) {
#define V3_C(I) case I: asm volatile(V3_QWRITE_##I ::"v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]) : V3_ALL_CL); break;
    V3_EACH16(V3_C)
#undef V3_C
  }
}
__device__ __forceinline__ void o_zero(int f) {
  switch (f) {
#define V3_C(F) case F: asm volatile(V3_OZERO_##F ::: V3_ALL_CL); break;
    V3_EACH10(V3_C)
#undef V3_C
  }
}
__device__ __forceinline__ void o_scale(int f, float alpha) {
  float tmp;
  switch (f) {
#define V3_C(F) case F: asm volatile(V3_OSCALE_##F : "=&v"(tmp) : "v"(alpha) : V3_ALL_CL); break;
    V3_EACH10(V3_C)
#undef V3_C
  }
}
__device__ __forceinline__ f32x16 o_read(int f) {
  float x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15;
  switch (f) {
#define V3_C(F) case F: asm volatile(V3_OREAD_##F : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4), "=v"(x5), "=v"(x6), "=v"(x7), "=v"(x8), "=v"(x9), "=v"(x10), "=v"(x11), "=v"(x12), "=v"(x13), "=v"(x14), "=v"(x15) : : V3_ALL_CL); break;
    V3_EACH10(V3_C)
#undef V3_C
  }
  return f32x16{x0, x1, x2, x3, x4, x5, x6, x7, x8, x9, x10, x11, x12, x13, x14, x15};
}
#define V3_SB() __builtin_amdgcn_sched_barrier(0)
// an opaque use+def: nothing that reads x can be scheduled above this point (MFMA results need 12 wait states before a VALU read;
// the asm MFMAs are invisible to hipcc's hazard recogniser, so the distance is kept by ORDER: two MFMAs always sit in between)
#define V3_PIN(x) asm volatile("" : "+v"(x))

typedef const __attribute__((address_space(3))) char* lds_cptr;
__device__ __forceinline__ s16x4 tr_read(lds_cptr p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

constexpr int REC3_WAVE_B = 8 * 16 * 64 * 4 + 64 * 16;   // one wave's partial: 8 O fragments (16 registers x 64 lanes, fp32) + (mA, lA, mB, lB) per lane
constexpr int REC3_B = 4 * REC3_WAVE_B;                 // 135 168 B per (workgroup, segment)
constexpr int CNT_ALIGN = 256;
constexpr int SPLIT_MIN_TILES = 8;

struct V3Geom {
  int S, H, nqb, ntiles, NI;   // NI = H * nqb work items per batch entry
  int spx;                     // workgroups per XCD group (one per CU)
  int split;                   // key-split tail enabled (workspace present)
};
// How the items of one XCD group are cut (identical on host and device): `nfull` items run whole, one per workgroup and round;
// the key tiles of the `rem` items of the last, partly filled round are dealt to all `spx` workgroups in equal contiguous runs
// (csrc/attention.hip: the same scheme at 256-row items).
struct V3Cut { int start, cnt, nfull, rem; };
__host__ __device__ inline V3Cut v3_cut(const V3Geom& G, int xcd) {
  V3Cut c;
  const int base = G.NI >> 3, extra = G.NI & 7;
  c.cnt = base + (xcd < extra ? 1 : 0);
  c.start = xcd * base + (xcd < extra ? xcd : extra);
  c.nfull = c.cnt;
  c.rem = 0;
  if (G.split) {
    const int nf = (c.cnt / G.spx) * G.spx, rem = c.cnt - nf;
    if (rem > 0 && rem * 16 < G.spx * 15 && (int64_t)rem * G.ntiles >= (int64_t)G.spx * SPLIT_MIN_TILES) {
      c.nfull = nf;
      c.rem = rem;
    }
  }
  return c;
}

__global__ __launch_bounds__(V3_THREADS, 1) void attention_v3_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                     const bf16_t* __restrict__ V, bf16_t* O, int64_t ld, int64_t stride_b,
                                                                     int64_t ldo, int64_t stride_ob, float scale_log2, const V3Geom G,
                                                                     int* counters, char* records) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3 stages][K|V]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int ntiles = G.ntiles;
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3, b = (int)blockIdx.y;
  const V3Cut cut = v3_cut(G, xcd);
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);

  // ---- per-lane constants
  const int srow = lane >> 4, spc = lane & 15;          // staging: wave w stages pieces 4w..4w+3 (4 rows each) of K and of V
  uint32_t soff[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row_t = (wave * 4 + p) * 4 + srow;
    soff[p] = ((uint32_t)row_t * (uint32_t)ld + (uint32_t)((spc ^ swz(row_t)) << 3)) * 2u;
  }
  const int tile_stride_b = BKV * (int)ld * 2;

  // ---- this workgroup's segments: its whole items (one per round), then the one or two pieces of its run of the split items
  const int nwhole = slot < cut.nfull ? (cut.nfull - slot + G.spx - 1) / G.spx : 0;
  int run_lo = 0, run_hi = 0, run_i0 = 0;
  if (cut.rem && slot < G.spx) {
    const unsigned U = (unsigned)cut.rem * (unsigned)ntiles;
    run_lo = (int)((unsigned)slot * U / (unsigned)G.spx);
    run_hi = (int)((unsigned)(slot + 1) * U / (unsigned)G.spx);
    run_i0 = run_lo / ntiles;
  }
  const int nseg = nwhole + (cut.rem && slot < G.spx ? 2 : 0);

  for (int si = 0; si < nseg; ++si) {
    int item, tb, te, seg = 0;
    if (si < nwhole) {
      item = cut.start + slot + si * G.spx;
      tb = 0;
      te = ntiles;
    } else {
      seg = si - nwhole;
      item = cut.start + cut.nfull + run_i0 + seg;
      tb = seg == 0 ? run_lo - run_i0 * ntiles : 0;
      te = seg == 0 ? min(run_hi - run_i0 * ntiles, ntiles) : run_hi - (run_i0 + 1) * ntiles;
      if (te <= tb) continue;
    }
    const int head = item / G.nqb;
    const int q0 = (item - head * G.nqb) * BQ3;
    const bf16_t* Qb = Q + b * stride_b + head * DH;
    const rt_srd_t rsrcK = rt_make_srd(K + b * stride_b + head * DH), rsrcV = rt_make_srd(V + b * stride_b + head * DH);
    const uint32_t wdst = lds0 + wave * 4096;           // this wave's four pieces inside a K (or V) tile
    auto dma_piece = [&](int sbase, uint32_t soffs, int i) __attribute__((always_inline)) {   // piece i of 8: K pieces 0..3, V pieces 4..7 of this wave
      rt_dma16_asm((i & 4) ? rsrcV : rsrcK, wdst + sbase + ((i & 4) ? TILE_B : 0) + (i & 3) * 1024, soff[i & 3], soffs);
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // nothing of the previous segment's (clamped) copies is still on its way
    __syncthreads();                                     // the previous segment's LDS reads (tiles, ticket word) are done
#pragma unroll
    for (int i = 0; i < 8; ++i) dma_piece(0, (uint32_t)(tb * tile_stride_b), i);
    if (tb + 1 < te) {
#pragma unroll
      for (int i = 0; i < 8; ++i) dma_piece(STAGE_B, (uint32_t)((tb + 1) * tile_stride_b), i);
    }
    // ---- Q fragments (B operand of Sᵀ = K·Qᵀ): lane holds Q[q][16ks + 8hh .. +7] for its row of block a and of block b
    {
      const bf16_t* qp = Qb + (int64_t)(q0 + wave * 64 + l31) * ld + 8 * hh;
      bf16x8 qa[8], qb[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        qa[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
        qb[ks] = *reinterpret_cast<const bf16x8*>(qp + 32 * ld + ks * 16);
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) { q_write(ks, qa[ks]); q_write(8 + ks, qb[ks]); }     // parked in a[128:191] for the whole segment
#pragma unroll
      for (int f = 0; f < 10; ++f) o_zero(f);
      ones_write();
      asm volatile("s_nop 7" ::: "memory");               // accvgpr writes settle before the first MFMA reads them
    }
    float mA = -INFINITY, mB = -INFINITY;                // running max (log2 domain) per query block; the row sums live in L fragments


    f32x16 sAa, sAb, sBa, sBb;          // scores: key half k0 (A) / k1 (B) x query block a / b
    bf16x8 pA[2][2], pB[2][2];          // numerators [query block][k-step of 16 keys]
    bf16x8 kf[8], vf[8];                // K fragments of one 32-key half [ks]; Vᵀ fragments of one half [s2*4 + dt]
    float mxa0, mxa1, mxb0, mxb1;       // running-max chains
    float tf[2], pp[2];                 // a softmax element in flight: exponent (after the fma), numerator (after the exp2)
    int pend1 = -1, pend2 = -1;         // elements whose exp2 / whose pack is still to come (compile-time constants after unrolling)

    // One softmax element o of a half (order: k-step, query block, j), against the CURRENT running max, as a three-stage pipeline a
    // gap apart — F: fma (score·scale − m), X: exp2, C: bf16 pack — so that no instruction of a gap waits for the result of the one
    // issued just before it (a lone wave has nobody else's instructions to fill those slots with). The opaque uses pin every step
    // where it is written: left alone, LLVM sinks the speculative numerators of a half below the decision branch that may redo
    // them — all 16 in one lump, no MFMA beside them.
    auto elem_f = [&](int o, const f32x16& Sa, const f32x16& Sb) __attribute__((always_inline)) {
      const int ks = o >> 4, qb = (o >> 3) & 1, j = o & 7, r = 8 * ks + j;
      float t = __builtin_fmaf(qb ? Sb[r] : Sa[r], scale_log2, qb ? -mB : -mA);
      asm volatile("" : "+v"(t));
      tf[o & 1] = t;
    };
    auto elem_x = [&](int o) __attribute__((always_inline)) {
      float p = __builtin_amdgcn_exp2f(tf[o & 1]);
      asm volatile("" : "+v"(p));
      pp[o & 1] = p;
    };
    auto elem_c = [&](int o, bf16x8 (&P)[2][2]) __attribute__((always_inline)) {
      const int ks = o >> 4, qb = (o >> 3) & 1, j = o & 7;
      P[qb][ks][j] = (__bf16)pp[o & 1];
      if (j & 1) asm volatile("" : "+v"(P[qb][ks]));
    };
    auto elem = [&](int o, const f32x16& Sa, const f32x16& Sb, bf16x8 (&P)[2][2]) __attribute__((always_inline)) {
      if (pend2 >= 0) elem_c(pend2, P);
      if (pend1 >= 0) elem_x(pend1);
      elem_f(o, Sa, Sb);
      pend2 = pend1;
      pend1 = o;
    };
    auto elem_flush = [&](bf16x8 (&P)[2][2]) __attribute__((always_inline)) {
      if (pend2 >= 0) elem_c(pend2, P);
      if (pend1 >= 0) { elem_x(pend1); elem_c(pend1, P); }
      pend1 = pend2 = -1;
    };
    // step k (0..7) of the max chain over the 16 scores of one fragment
    auto maxstep = [&](int k, const f32x16& s, float& m0, float& m1) __attribute__((always_inline)) {
      if (k == 0) m0 = max3f(s[0], s[1], s[2]);
      else if (k == 1) m1 = max3f(s[3], s[4], s[5]);
      else if (k == 2) m0 = max3f(m0, s[6], s[7]);
      else if (k == 3) m1 = max3f(m1, s[8], s[9]);
      else if (k == 4) m0 = max3f(m0, s[10], s[11]);
      else if (k == 5) m1 = max3f(m1, s[12], s[13]);
      else if (k == 6) m0 = max3f(m0, s[14], s[15]);
      else m0 = fmaxf(m0, m1);
    };
    // both halves of a row's maximum (lanes l and l+32) through one v_permlane32_swap (asm: see attention.hip), scaled to log2 units
    auto exchange = [&](float mx) __attribute__((always_inline)) -> float {
      unsigned xa = __builtin_bit_cast(unsigned, mx), xb = xa;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xa), "+v"(xb));
      return fmaxf(__builtin_bit_cast(float, xa), __builtin_bit_cast(float, xb)) * scale_log2;
    };
    // Decision for one half whose first 16 numerators (P[.][0]) were formed speculatively against (mA, mB): when a row's maximum
    // outgrew the running one by more than the threshold, move the maxima, rescale l (from its snapshot) and O, and redo those 16.
    auto decide = [&](float mxra, float mxrb, const f32x16& Sa, const f32x16& Sb, bf16x8 (&P)[2][2]) __attribute__((always_inline)) {
      if (__builtin_expect(!!__any((mxra - mA > RESCALE_THR) | (mxrb - mB > RESCALE_THR)), 0)) {
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // O's last MFMA has retired before the copies below read it
        const float nA = fmaxf(mA, mxra), nB = fmaxf(mB, mxrb);
        const float alA = __builtin_amdgcn_exp2f(mA - nA), alB = __builtin_amdgcn_exp2f(mB - nB);   // first tile: exp2(-inf) = 0
        mA = nA; mB = nB;
#pragma unroll
        for (int f = 0; f < 10; ++f) o_scale(f, (f & 1) ? alB : alA);
#pragma unroll
        for (int o = 0; o < 16; ++o) { elem_f(o, Sa, Sb); elem_x(o); elem_c(o, P); }
      }
    };
    // LDS read addresses of the stage in use: ka = K rows of one stage, va = Vᵀ blocks of one stage. They are moved to the next
    // stage once per tile (16 integer adds, as fillers) and made opaque, so every read is base register + immediate.
    // (recomputed per segment from the lane index: 16 registers that need not stay live across the partial-record / combine code)
    int ka[8];                                             // K row read: row 32kb + l31, chunk 2ks + hh (LDS byte address in stage 0)
    {
      const int ksw = swz(l31);
  #pragma unroll
      for (int ks = 0; ks < 8; ++ks) ka[ks] = (int)lds0 + l31 * 256 + (((2 * ks + hh) ^ ksw) << 4);
    }
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg1 = (lane >> 4) & 1;
    int va[8];                                             // Vᵀ transposed reads (see attention.hip): [row block 0/1][dt]
    {
      const int cl = tg1 * 2 + (tp >> 1);
      const int rl0 = 4 * hh + tq, rl1 = rl0 + 8;
  #pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        va[dt] = (int)lds0 + TILE_B + rl0 * 256 + (((dt * 4 + cl) ^ swz(rl0)) << 4) + 8 * (tp & 1);
        va[4 + dt] = (int)lds0 + TILE_B + rl1 * 256 + (((dt * 4 + cl) ^ swz(rl1)) << 4) + 8 * (tp & 1);
      }
    }
    auto mov_ka = [&](int i, int delta) __attribute__((always_inline)) { ka[i] += delta; asm volatile("" : "+v"(ka[i])); };
    auto mov_va = [&](int i, int delta) __attribute__((always_inline)) { va[i] += delta; asm volatile("" : "+v"(va[i])); };
    auto kread = [&](int kb, int ks) __attribute__((always_inline)) -> bf16x8 {
      return *(const __attribute__((address_space(3))) bf16x8*)((lds_cptr)(uintptr_t)(uint32_t)(ka[ks] + kb * 8192));
    };
    auto vread = [&](int kb, int s2, int dt) __attribute__((always_inline)) -> bf16x8 {
      const s16x4 lo = tr_read((lds_cptr)(uintptr_t)(uint32_t)(va[dt] + kb * 8192 + s2 * 4096));
      const s16x4 hi = tr_read((lds_cptr)(uintptr_t)(uint32_t)(va[4 + dt] + kb * 8192 + s2 * 4096));
      return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // Sᵀ of one 32-key half (16 MFMAs: k-step x query block), one filler slot per gap
    auto qk_group = [&](f32x16& Sa, f32x16& Sb, auto&& filler) __attribute__((always_inline)) {
      // ONE wait for the group's eight fragments (an opaque use of all of them) instead of a counted s_waitcnt in front of every MFMA
      asm volatile("" ::"v"(kf[0]), "v"(kf[1]), "v"(kf[2]), "v"(kf[3]), "v"(kf[4]), "v"(kf[5]), "v"(kf[6]), "v"(kf[7]));
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int ks = g >> 1;
        if (g == 0) mfma_s0(Sa, kf[0], 0);
        else if (g == 1) mfma_s0(Sb, kf[0], 8);
        else if (g & 1) mfma_s(Sb, kf[ks], 8 + ks);
        else mfma_s(Sa, kf[ks], ks);
        filler(g);
        V3_SB();
      }
    };
    // Oᵀ += Vᵀ·Pᵀ for one 32-key half (16 MFMAs: k-step x d block x query block)
    // 20 MFMAs: per k-step 8 x (d block, query block) + 2 row-sum MFMAs; filler(g) is called for g = 0..19
    auto pv_group = [&](bf16x8 (&P)[2][2], auto&& filler) __attribute__((always_inline)) {
      asm volatile("" ::"v"(vf[0]), "v"(vf[1]), "v"(vf[2]), "v"(vf[3]), "v"(vf[4]), "v"(vf[5]), "v"(vf[6]), "v"(vf[7]));
#pragma unroll
      for (int g = 0; g < 20; ++g) {
        const int s2 = g / 10, h = g % 10;
        if (h < 8) mfma_o((h >> 1) * 2 + (h & 1), vf[s2 * 4 + (h >> 1)], P[h & 1][s2]);
        else mfma_l(h & 1, P[h & 1][s2]);
        filler(g);
        V3_SB();
      }
    };
    // fillers of a group that finishes a half: max exchange (gaps 0, 1), decision (gap 2), numerators 16..31 (gaps 3..15)
    auto second_half = [&](int g, const f32x16& Sa, const f32x16& Sb, bf16x8 (&P)[2][2], float ra, float rb, float& ea, float& eb)
        __attribute__((always_inline)) {
      if (g == 0) ea = exchange(ra);
      if (g == 1) eb = exchange(rb);
      if (g == 2) decide(ea, eb, Sa, Sb, P);
      if (g >= 3 && g <= 12) elem(13 + g, Sa, Sb, P);                                  // 16..25
      if (g >= 13) { elem(26 + 2 * (g - 13), Sa, Sb, P); elem(27 + 2 * (g - 13), Sa, Sb, P); }   // 26..31
      if (g == 15) elem_flush(P);
    };
    // fillers of a PV group that starts the next half: speculative numerators 0..15, its max chains
    auto first_half = [&](int g, f32x16& Sa, f32x16& Sb, bf16x8 (&P)[2][2]) __attribute__((always_inline)) {
      if (g == 0) V3_PIN(Sa);                      // two MFMAs behind its last accumulation: safe to read from here on
      if (g == 1) V3_PIN(Sb);
      if (g >= 2 && g <= 17) elem(g - 2, Sa, Sb, P);     // elements 0..15, one per gap
      if (g == 18) elem_flush(P);
      if (g >= 1 && g <= 8) maxstep(g - 1, Sa, mxa0, mxa1);
      if (g >= 10 && g <= 17) maxstep(g - 10, Sb, mxb0, mxb1);
    };

    // ---------------------------------------------------------------------------------------------- the tile loop
    // stage of tile t = (t - tb) % 3; the segment's first two tiles are in flight (the waits of the Q loads covered them: in-order counter)
    float rka = 0.f, rkb = 0.f;        // row maxima (this lane's 16 keys) of the half whose decision is pending
    int st_cur = 0, st_nxt = STAGE_B, st_prv = 2 * STAGE_B;
    auto tile = [&](auto first_c, auto last_c, int t) __attribute__((always_inline)) {
      constexpr bool first = decltype(first_c)::value, last = decltype(last_c)::value;
      const int sb = st_cur, sbn = st_nxt, sbp = st_prv;             // LDS byte offsets of the stages of tiles t, t+1, t-1 (= t+2)
      const uint32_t so2 = (uint32_t)(min(t + 2, te - 1) * tile_stride_b);   // tile t+2 (clamped: a harmless re-copy at the end)
      float ea = 0.f, eb = 0.f;
      // ---- G1: Sᵀ(k0,t) ∥ decision + numerators 16..31 of k1(t-1), Vᵀ(k1,t-1) fragments
      qk_group(sAa, sAb, [&](int g) __attribute__((always_inline)) {
        if constexpr (!first) {
          second_half(g, sBa, sBb, pB, rka, rkb, ea, eb);
          if ((g & 1) == 0) { const int i = g >> 1; vf[(i >> 2) * 4 + (i & 3)] = vread(1, i >> 2, i & 3); }
        }
      });
      // ---- G2: PV(k1,t-1) ∥ speculative numerators 0..15 of k0(t), max chains of k0(t), K(k1,t) fragments
      if constexpr (!first) {
        pv_group(pB, [&](int g) __attribute__((always_inline)) {
          first_half(g, sAa, sAb, pA);
          if (g < 16) { if (g & 1) kf[g >> 1] = kread(1, g >> 1); else mov_va(g >> 1, sb - sbp); }
        });
      } else {
        // first tile: no PV to overlap with; the scores of k0 need their 12 wait states before the VALU reads them
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        V3_PIN(sAa); V3_PIN(sAb);
#pragma unroll
        for (int g = 0; g < 8; ++g) { maxstep(g, sAa, mxa0, mxa1); maxstep(g, sAb, mxb0, mxb1); }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) kf[ks] = kread(1, ks);      // va already points at the segment's first stage
      }
      rka = mxa0; rkb = mxb0;                          // maxstep 7 leaves a chain's result in its first variable
      V3_SB();
      // ---- G3: Sᵀ(k1,t) ∥ exchange + decision for k0(t), numerators 16..31 (first tile: all 32) of k0(t), Vᵀ(k0,t) fragments
      qk_group(sBa, sBb, [&](int g) __attribute__((always_inline)) {
        if constexpr (first) {
          if (g == 0) ea = exchange(rka);
          if (g == 1) { eb = exchange(rkb); mA = fmaxf(mA, ea); mB = fmaxf(mB, eb); }      // nothing accumulated yet: just fix the maxima
          if (g >= 2) { elem(2 * (g - 2), sAa, sAb, pA); elem(2 * (g - 2) + 1, sAa, sAb, pA); }
          if (g == 14) { elem(28, sAa, sAb, pA); elem(29, sAa, sAb, pA); }
          if (g == 15) { elem(30, sAa, sAb, pA); elem(31, sAa, sAb, pA); elem_flush(pA); }
        } else {
          second_half(g, sAa, sAb, pA, rka, rkb, ea, eb);
        }
        if ((g & 1) == 0) { const int i = g >> 1; vf[(i >> 2) * 4 + (i & 3)] = vread(0, i >> 2, i & 3); } else { mov_ka(g >> 1, sbn - sb); }
      });
      // ---- barrier: tile t+1 has landed everywhere, the stage of tile t-1 is free
      if constexpr (!last) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        V3_SB();
      }
      // ---- G4: PV(k0,t) ∥ speculative numerators 0..15 of k1(t), max chains of k1(t), K(k0,t+1) fragments, DMA of tile t+2
      pv_group(pA, [&](int g) __attribute__((always_inline)) {
        first_half(g, sBa, sBb, pB);
        if constexpr (!last) {
          if (g < 16 && (g & 1)) kf[g >> 1] = kread(0, g >> 1);
          if (g >= 2 && g < 18 && (g & 1) == 0) dma_piece(sbp, so2, (g - 2) >> 1);
        }
      });
      { const int r_ = st_cur; st_cur = st_nxt; st_nxt = st_prv; st_prv = r_; }        // rotate the ring
      rka = mxa0; rkb = mxb0;
      if constexpr (last) {
        // ---- drain: decision + numerators 16..31 of k1(t), then PV(k1,t)
        V3_SB();
        const float ea2 = exchange(rka), eb2 = exchange(rkb);
        decide(ea2, eb2, sBa, sBb, pB);
#pragma unroll
        for (int o = 16; o < 32; ++o) elem(o, sBa, sBb, pB);
        elem_flush(pB);
#pragma unroll
        for (int i = 0; i < 8; ++i) vf[(i >> 2) * 4 + (i & 3)] = vread(1, i >> 2, i & 3);
        V3_SB();
        pv_group(pB, [&](int) __attribute__((always_inline)) {});
      }
    };
    using TT = std::true_type;
    using FF = std::false_type;
    __syncthreads();                                     // the segment's first tile has landed everywhere (every wave's Q-load waits covered its DMA)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kf[ks] = kread(0, ks);
    if (te - tb == 1) {
      tile(TT{}, TT{}, tb);
    } else {
      tile(TT{}, FF{}, tb);
      for (int t = tb + 1; t + 1 < te; ++t) tile(FF{}, FF{}, t);
      tile(FF{}, TT{}, te - 1);
    }

    // ---- the segment's result: Oᵀ in a[0:127] (unnormalised), row sums in a[192:223], maxima in (mA, mB)
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");            // the last MFMA has retired before its accumulators are read
    const bool whole = (tb == 0 && te == ntiles);
    const float lA = o_read(8)[0], lB = o_read(9)[0];              // every register of an L fragment holds its query's row sum
    bf16_t* orow = O + b * stride_ob + (int64_t)(q0 + wave * 64 + l31) * ldo + head * DH;
    if (whole) {
      // ---- epilogue: O[q][d] = Oᵀ / l ; lane holds q = l31 of each block, d = 32dt + (r&3) + 8(r>>2) + 4hh
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        f32x16 t4[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) t4[dt] = o_read(dt * 2 + qb);
        rt_store_o_rows(orow + qb * 32 * ldo, true, hh, t4, 1.0f / (qb ? lB : lA));
      }
      continue;
    }
    // ---- partial: write (Oᵀ, m, l) through to memory, take a ticket on the item; the last ticket combines the item's records
    const int jpart = slot;
    const int ritem = item - (cut.start + cut.nfull);            // index among the split items of this group
    {
      char* rec = records + ((((int64_t)b * 8 + xcd) * G.spx + jpart) * 2 + seg) * (int64_t)REC3_B + wave * REC3_WAVE_B;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rec, 0, REC3_WAVE_B, 0x00020000);
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        const f32x16 x = o_read(f);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 vv = {x[4 * g + 0], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, vv), rs, (f * 4 + g) * 1024 + lane * 16, 0, 16 /* sc1: write-through */);
        }
      }
      const f32x4 ml = {mA, lA, mB, lB};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ml), rs, 32768 + lane * 16, 0, 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains its write-through stores
    __syncthreads();
    int* cnt = counters + (int64_t)b * G.NI + item;
    const unsigned U = (unsigned)cut.rem * (unsigned)ntiles, spx = (unsigned)G.spx;
    const unsigned a = (unsigned)ritem * (unsigned)ntiles, bnd = a + (unsigned)ntiles;
    const int j_first = (int)(((a + 1) * spx + U - 1) / U) - 1;
    const int j_last = min(G.spx - 1, (int)((bnd * spx + U - 1) / U) - 1);
    const int nparts = j_last - j_first + 1;
    volatile int* flag = reinterpret_cast<volatile int*>(smem);
    if (tid == 0) {
      const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int lastp = (old == nparts - 1) ? 1 : 0;
      if (lastp) {
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                         // drop this CU's stale lines
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      *flag = lastp;
    }
    __syncthreads();
    if (!__builtin_amdgcn_readfirstlane(*flag)) continue;
    // record of splitting workgroup j for this item: its second segment when the item starts after the run does
    auto rec_of = [&](int j) -> const char* {
      const unsigned lo_j = (unsigned)j * U / spx;
      const int sj = (a > lo_j) ? 1 : 0;
      return records + ((((int64_t)b * 8 + xcd) * G.spx + j) * 2 + sj) * (int64_t)REC3_B + wave * REC3_WAVE_B;
    };
    // pass 1: common maxima and the combined row sums; pass 2: per query block, weighted sums of the records IN RUN ORDER
    // (bitwise reproducible whatever the arrival order was). One query block's four fragments at a time: 64 accumulators.
    float MA = -INFINITY, MB = -INFINITY;
    for (int j = j_first; j <= j_last; ++j) {
      const f32x4 ml = *reinterpret_cast<const f32x4*>(rec_of(__builtin_amdgcn_readfirstlane(j)) + 32768 + lane * 16);
      MA = fmaxf(MA, ml[0]);
      MB = fmaxf(MB, ml[2]);
    }
    float LA = 0.f, LB = 0.f;
    for (int j = j_first; j <= j_last; ++j) {
      const f32x4 ml = *reinterpret_cast<const f32x4*>(rec_of(__builtin_amdgcn_readfirstlane(j)) + 32768 + lane * 16);
      LA = __builtin_fmaf(ml[1], __builtin_amdgcn_exp2f(ml[0] - MA), LA);
      LB = __builtin_fmaf(ml[3], __builtin_amdgcn_exp2f(ml[2] - MB), LB);
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      f32x16 t4[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) t4[dt][r] = 0.f;
      for (int j = j_first; j <= j_last; ++j) {
        const char* rj = rec_of(__builtin_amdgcn_readfirstlane(j));
        const f32x4 ml = *reinterpret_cast<const f32x4*>(rj + 32768 + lane * 16);
        const float w = __builtin_amdgcn_exp2f(qb ? ml[2] - MB : ml[0] - MA);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rj + ((dt * 2 + qb) * 4 + g) * 1024 + lane * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) t4[dt][4 * g + e] = __builtin_fmaf(v[e], w, t4[dt][4 * g + e]);
          }
      }
      rt_store_o_rows(orow + qb * 32 * ldo, true, hh, t4, 1.0f / (qb ? LB : LA));
    }
  }
}

int g_v3_mode = -1;       // 1 (default, RT_ATTN_V3) = use attention_v3 where it applies and pays; 2 = wherever it applies (tests)
int v3_mode_now() {
  if (g_v3_mode < 0) {
    const char* e = getenv("RT_ATTN_V3");
    g_v3_mode = e ? atoi(e) : 1;
  }
  return g_v3_mode;
}
V3Geom v3_geom(int S, int H, bool split) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int x = 0;
    if (hipDeviceGetAttribute(&x, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && x > 0) cus = x;
  }
  V3Geom G;
  G.S = S; G.H = H; G.nqb = S / BQ3; G.ntiles = S / BKV; G.NI = H * G.nqb;
  G.spx = cus / 8 > 0 ? cus / 8 : 1;
  G.split = split ? 1 : 0;
  return G;
}

}  // namespace

// The counter region is laid out for attention.hip's item count (128-row items: twice ours), so one workspace serves either
// kernel: both keep their counters inside it (zero at rest) and their records behind it.
static int64_t v3_cnt_bytes(int32_t B, int32_t S, int32_t H) {
  return (((int64_t)B * H * ((S + 127) / 128) * 4 + CNT_ALIGN - 1) / CNT_ALIGN) * CNT_ALIGN;
}
// Workspace attention_v3 wants for (B, S, H): ticket counters + partial records of the key-split tail; 0 when nothing would be split
// or the shape is not taken. rt_attention_ws_bytes (attention.hip) returns the larger of the two kernels' needs.
int64_t rt_attention_v3_ws_bytes(int32_t B, int32_t S, int32_t H) {
  if (S % BQ3 != 0) return 0;
  const V3Geom G = v3_geom(S, H, true);
  bool any = false;
  for (int x = 0; x < 8; ++x) any = any || v3_cut(G, x).rem > 0;
  if (!any) return 0;
  return v3_cnt_bytes(B, S, H) + (int64_t)B * 8 * G.spx * 2 * REC3_B;
}

// Called by rt_attention_fwd: returns 1 when the launch was taken over, 0 when the shape is left to attention.hip, < 0 / hipError on failure.
int rt_attention_v3_try(const void* q, const void* k, const void* v, void* o, int64_t ld, int64_t stride_b, int64_t ldo, int64_t stride_ob,
                        int32_t B, int32_t S, int32_t H, float scale, void* ws, int64_t ws_bytes, void* stream) {
  if (!v3_mode_now() || S % BQ3 != 0 || (ldo % 8) || (stride_ob % 8) || !RT_ALIGNED(o, 16)) return 0;
  // short sequences (config 1's S = 768: 25 vs 19 us) stay with attention.hip, whose two workgroups per CU hide each other's
  // prologues; from ~1.5 k keys on the 64-rows-per-wave loop wins (S = 4608: 247 vs 264 us, S = 9728: 986 vs 1065 us)
  if (v3_mode_now() == 1 && S < 1536) return 0;
  const int64_t need = rt_attention_v3_ws_bytes(B, S, H);
  const bool split = ws != nullptr && need > 0 && ws_bytes >= need && RT_ALIGNED(ws, 256);
  const V3Geom G = v3_geom(S, H, split);
  if ((int64_t)G.NI * (G.ntiles + 1) * G.spx >= ((int64_t)1 << 31)) return 0;
  const int lds = NSTAGE * STAGE_B;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_v3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  int slots = 0;
  for (int x = 0; x < 8; ++x) {
    const V3Cut c = v3_cut(G, x);
    const int s = c.rem ? G.spx : (c.nfull < G.spx ? c.nfull : G.spx);
    slots = s > slots ? s : slots;
  }
  const int64_t cnt_b = v3_cnt_bytes(B, S, H);
  hipLaunchKernelGGL(attention_v3_kernel, dim3(8 * slots, B), dim3(V3_THREADS), lds, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k,
                     (const bf16_t*)v, (bf16_t*)o, ld, stride_b, ldo, stride_ob, scale * 1.4426950408889634f, G, split ? (int*)ws : nullptr,
                     split ? (char*)ws + cnt_b : nullptr);
  const int st = rt_hip_status();
  return st == RT_OK ? 1 : st;
}

int rt_attention_v3_mode(int mode) {
  const int prev = v3_mode_now();
  if (mode >= 0) g_v3_mode = mode;
  return prev;
}
