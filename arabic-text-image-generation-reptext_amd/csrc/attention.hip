// rt_attention_fwd — joint text+image attention of the MMDiT blocks: softmax(Q·Kᵀ·scale)·V, non-causal,
// no mask, head dim 128, bf16 in/out, fp32 scores/statistics/accumulators. Flash-style: the S×S score
// matrix never leaves the CU. Reference: torch SDPA as called by diffusers' FluxAttnProcessor (SURVEY.md
// Appendix A.1 step 6 / A.2), reached from controlnet_flux.py:343-348,376-380 and PIPE:1092.
//
// Work item = 128 query rows of one (batch, head); a workgroup = 4 waves, wave w owns query rows 32w..32w+31.
// Keys/values arrive in 64-row tiles by LDS-DMA into a 2-deep ring (K 16 KiB + V 16 KiB per slot).
//
// Both products run on v_mfma_f32_32x32x16_bf16 with the QUERY index on the lane:
//   Sᵀ[key][q]  = K · Qᵀ      A = K rows from LDS (ds_read_b128), B = Q rows held in registers
//   Oᵀ[d][q]   += Vᵀ · Pᵀ     A = Vᵀ via ds_read_b64_tr_b16 (hardware transpose), B = the Sᵀ accumulator
//                             registers themselves, exponentiated and packed to bf16 (no LDS round trip)
// so the softmax max/sum of a query row are lane-local apart from one lane<->lane+32 exchange, and the
// per-row rescale of O is a per-lane scalar multiply.
//
// Schedule inside a wave (per 64-key tile, two 32-key halves h0, h1):
//   Sᵀ(h0) | Sᵀ(h1) ∥ numerators of h0 | check h0 | Oᵀ += Vᵀ(h0)·Pᵀ(h0) ∥ numerators of h1 | check h1 | Oᵀ += Vᵀ(h1)·Pᵀ(h1)
// The numerators of a half are computed SPECULATIVELY against the running max m_run while the matrix pipe works on the
// next product; the row max of the half is taken beside them and checked afterwards. Only when some row's max outgrew
// m_run by more than RESCALE_THR (rare after the first tiles) a fix-up block rescales O and l and recomputes that half's
// numerators against the new max — so the common path has no branch between an MFMA chain and the vector work that
// overlaps it, and every numerator that reaches an MFMA was formed against the max O is normalised with.
//
// Placement (speed only, never correctness): the grid is one-dimensional per batch entry. Workgroups are dealt
// round-robin over the 8 XCDs, so blockIdx.x & 7 labels the XCD group; each group owns a CONTIGUOUS run of the
// (head, query block) items, i.e. whole heads (3 of 24 per XCD): K/V of a head are fetched into one XCD's L2 only.
//
// Key-split tail: with NI items per XCD group on `spx` workgroup slots (2 per CU), the last NI mod spx items would
// occupy a partial round of full-length workgroups (864 items on 512 slots: 1.69 rounds run as 2). Instead those
// `rem` items × ntiles key tiles are dealt to `spx` workgroups in equal contiguous runs of tiles; a run covers the
// tail of one item and/or the head of the next. A workgroup that covers only part of an item writes its unnormalised
// (m, l, Oᵀ) to a workspace record (write-through stores) and takes a ticket on the item's counter; the workgroup that
// draws the last ticket combines the item's records IN RUN ORDER (not arrival order: bitwise reproducible) and
// stores the output. Nobody waits on anybody, so no dispatch-order assumption exists. The decomposition depends on
// (S, H, CU count) only — every batch entry is cut identically, so results do not depend on the batch size.
//
// LDS image of a K or V tile: 64 rows × 256 B; 16-B chunk c of row r lives at 256·r + 16·(c ^ f(r)),
// f(r) = ((r&3)<<2) | ((r>>2)&3): conflict-free for the row reads (K) and the transposed reads (V).
// LDS-DMA writes lane-linear, so the XOR is applied to each lane's source address.
#include "rt_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int DH = 128;
constexpr int BQ = 128;       // query rows per work item
constexpr int BKV = 64;       // keys per tile
constexpr int TILE_B = BKV * DH * 2;   // 16 KiB
constexpr int ATT_THREADS = 256;
constexpr float RESCALE_THR = 6.0f;   // log2 units: P <= 64 between rescales
constexpr int REC_WAVE_B = 64 * 64 * 4 + 64 * 8;   // one wave's partial: Oᵀ (64 registers × 64 lanes, fp32) + (m, l) per lane
constexpr int REC_B = 4 * REC_WAVE_B;              // 67 584 B per (workgroup, segment)
constexpr int CNT_ALIGN = 256;
constexpr int SPLIT_MIN_TILES = 8;                 // do not split runs shorter than this many key tiles

struct AttnGeom {
  int S, H, nqb, ntiles, NI;   // NI = H * nqb work items per batch entry
  int spx;                     // workgroup slots per XCD group (2 per CU)
  int split;                   // key-split tail enabled (workspace present and it pays)
};

// Wave-uniform conditions that are almost never true (ragged last tile, a row maximum that outgrew the running one): tell the
// block placement so, so that the rare code sits out of line and the common path falls through.
#define RT_RARE(c) __builtin_expect(!!(c), 0)
#define RT_USUAL(c) __builtin_expect(!!(c), 1)

__device__ __forceinline__ int swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// 3-input max. Plain fmaxf so the compiler inserts the MFMA->VALU wait states itself: an inline-asm v_max3 here read the
// accumulators before the MFMA had retired them (run-to-run differences in the running max; csrc/Makefile builds this file
// with -fno-honor-nans so no canonicalising v_max is emitted in front and the pair folds to one v_max3_f32).
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

typedef const __attribute__((address_space(3))) char* lds_cptr;
__device__ __forceinline__ s16x4 tr_read(lds_cptr p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// How the items of one XCD group are cut (identical on host and device; scalar arithmetic only).
struct GroupCut {
  int start, cnt;   // first item and number of items of this group
  int nfull;        // items run by one full-length workgroup each
  int rem;          // items whose key tiles are dealt to `spx` workgroups (0: no split)
};
__host__ __device__ inline GroupCut group_cut(const AttnGeom& G, int xcd) {
  GroupCut c;
  const int base = G.NI >> 3, extra = G.NI & 7;
  c.cnt = base + (xcd < extra ? 1 : 0);
  c.start = xcd * base + (xcd < extra ? xcd : extra);
  c.nfull = c.cnt;
  c.rem = 0;
  if (G.split) {
    const int nf = (c.cnt / G.spx) * G.spx, rem = c.cnt - nf;
    // split when the partial round is less than 15/16 full and every run keeps >= SPLIT_MIN_TILES tiles
    if (rem > 0 && rem * 16 < G.spx * 15 && (int64_t)rem * G.ntiles >= (int64_t)G.spx * SPLIT_MIN_TILES) {
      c.nfull = nf;
      c.rem = rem;
    }
  }
  return c;
}
__host__ __device__ inline int group_slots(const AttnGeom& G, int xcd) {
  const GroupCut c = group_cut(G, xcd);
  return c.nfull + (c.rem ? G.spx : 0);
}

__global__ __launch_bounds__(ATT_THREADS, 2) void attention_fwd_kernel(
    const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, bf16_t* O,
    int64_t ld, int64_t stride_b, int64_t ldo, int64_t stride_ob, float scale_log2, const AttnGeom G,
    int* counters, char* records) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];   // [slot][K|V]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int S = G.S;
  const int ntiles = G.ntiles;

  // ---- which run of (item, key tile) pairs is this workgroup's? (scalar)
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3, b = (int)blockIdx.y;
  const GroupCut cut = group_cut(G, xcd);
  int run_lo, run_hi;            // in units of key tiles, relative to the first item this workgroup touches
  int item0;                     // that first item
  int jpart = -1;                // index among the `spx` splitting workgroups, -1 for a full-length workgroup
  if (slot < cut.nfull) {
    item0 = cut.start + slot;
    run_lo = 0;
    run_hi = ntiles;
  } else {
    jpart = slot - cut.nfull;
    if (cut.rem == 0 || jpart >= G.spx) return;
    const unsigned U = (unsigned)cut.rem * (unsigned)ntiles;      // host checks rem * ntiles * spx < 2^31
    const int lo = (int)((unsigned)jpart * U / (unsigned)G.spx), hi = (int)((unsigned)(jpart + 1) * U / (unsigned)G.spx);
    const int i0 = lo / ntiles;
    item0 = cut.start + cut.nfull + i0;
    run_lo = lo - i0 * ntiles;
    run_hi = hi - i0 * ntiles;   // may exceed ntiles: the run continues into item0 + 1
  }

  // ---- per-lane constants that do not depend on the item
  // staging: wave w stages pieces 4w..4w+3 (4 rows each) of K and of V; K and V share row/chunk offsets.
  const int srow = lane >> 4;                 // row inside a piece
  const int spc = lane & 15;                  // physical chunk written by this lane
  uint32_t soff[4];                           // BYTE offset of (row, logical chunk) from the tile's first row
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row_t = (wave * 4 + p) * 4 + srow;          // row inside the tile
    soff[p] = ((uint32_t)row_t * (uint32_t)ld + (uint32_t)((spc ^ swz(row_t)) << 3)) * 2u;
  }
  // K row read: row = 32kt + l31, chunk = 2ks + hh  ->  l31*256 + ((2ks+hh) ^ swz(l31))*16  (+ kt*8192)
  lds_cptr kp[8];
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

  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  for (int seg = 0; seg < 2; ++seg) {
    // ---- segment = the part of this workgroup's run that lies inside one item
    const int item = item0 + seg;
    const int tb = seg == 0 ? run_lo : 0;
    const int te = seg == 0 ? min(run_hi, ntiles) : run_hi - ntiles;
    if (te <= tb) break;
    const int head = item / G.nqb;
    const int q0 = (item - head * G.nqb) * BQ;
    const bf16_t* Qb = Q + b * stride_b + head * DH;
    const bf16_t* Kb = K + b * stride_b + head * DH;
    const bf16_t* Vb = V + b * stride_b + head * DH;

    // LDS-DMA by buffer_load ... lds: 4-SGPR descriptor per operand + the lane's loop-invariant 32-bit byte offset + the tile's
    // byte offset in an SGPR — no per-tile vector address arithmetic. num_records = 2^32-1: rows are clamped here, not by the
    // range check (the host checks (S + 64) * ld * 2 < 2^31).
    const rt_srd_t rsrcK = rt_make_srd(Kb), rsrcV = rt_make_srd(Vb);
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);          // LDS byte address of the ring
    const int tile_stride_b = BKV * (int)ld * 2;
    // One branch for the whole tile, the ragged case out of line: a taken branch costs the wave an instruction-buffer refill, and the
    // common path of the tile loop should not contain any (see RT_RARE).
    auto stage = [&](int sl, int tix, bool clamp) {
      const uint32_t kb = lds0 + sl * 2 * TILE_B;
      if (RT_RARE(clamp)) {   // ragged last tile: rows past the end re-read row S-1 (masked later)
        const int kv0 = tix * BKV;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int row_t = (wave * 4 + p) * 4 + srow;
          const uint32_t off = ((uint32_t)(min(kv0 + row_t, S - 1) - kv0) * (uint32_t)ld + (uint32_t)((spc ^ swz(row_t)) << 3)) * 2u;
          rt_dma16_asm(rsrcK, kb + (wave * 4 + p) * 1024, off, (uint32_t)(tix * tile_stride_b));
          rt_dma16_asm(rsrcV, kb + TILE_B + (wave * 4 + p) * 1024, off, (uint32_t)(tix * tile_stride_b));
        }
      } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          rt_dma16_asm(rsrcK, kb + (wave * 4 + p) * 1024, soff[p], (uint32_t)(tix * tile_stride_b));
          rt_dma16_asm(rsrcV, kb + TILE_B + (wave * 4 + p) * 1024, soff[p], (uint32_t)(tix * tile_stride_b));
        }
      }
    };

    __syncthreads();                            // previous segment's LDS reads (tiles, ticket word) are done
    stage(0, tb, (tb + 1) * BKV > S);     // first tile in flight before the Q loads are waited for

    // ---- Q fragments (B operand of Sᵀ = K·Qᵀ): lane holds Q[q][16ks + 8hh .. +7]
    bf16x8 qf[8];
    {
      const int qrow = min(q0 + wave * 32 + l31, S - 1);
      const bf16_t* qp = Qb + (int64_t)qrow * ld + 8 * hh;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
      // The loads are waited for HERE (an opaque use), once per item: left to the first use, hipcc keeps the counted vmcnt waits
      // inside the tile loop, where they would also wait for the (uncounted, asm-issued) LDS-DMA of the next tile.
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+v"(qf[ks]));
    }

    f32x16 o_acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;     // running max (log2 domain, may lag the true max by <= RESCALE_THR) and row sum

    // ---- softmax pieces (all per lane; the row's other keys live in lane ^ 32)
    // row max of one 32-key half (this lane's 16 keys)
    auto half_max = [&](const f32x16& s) -> float {
      float ma = max3f(s[0], s[1], s[2]), mb = max3f(s[3], s[4], s[5]);
      ma = max3f(ma, s[6], s[7]);
      mb = max3f(mb, s[8], s[9]);
      ma = max3f(ma, s[10], s[11]);
      mb = max3f(mb, s[12], s[13]);
      return fmaxf(max3f(ma, s[14], s[15]), mb);
    };
    // Deferred rescale: O and l are only rescaled when some row's max grew by more than RESCALE_THR (log2 units) over the
    // max it is currently normalised with; otherwise P = exp2(s - m_run) <= 2^THR, harmless in fp32 accumulators and
    // (being a relative format) in the bf16 P operand. Returns the factor O must be multiplied with (1 = nothing to do);
    // m_run and l_run are updated here, O by the caller once every MFMA fed with old-scale numerators has been issued.
    auto decide = [&](float mx, bool& pend) -> float {
      // Both partial maxima of a row (lanes l and l+32) through one v_permlane32_swap: it exchanges the upper half of its first
      // operand with the lower half of its second, so after it each lane holds its own value in one register and its
      // partner's in the other. Inline asm: with the builtin, hipcc (ROCm 7.2) folds max(result[0], result[1]) to result[0]
      // (the swap is emitted, the v_max is not), so each half-wave saw only one of the two maxima — found by the
      // spiked-key test: a row whose largest score sat in lanes 32-63 was never rescaled. s_nop 1 = the two wait states
      // a VALU write needs before v_permlane*_swap reads it.
      unsigned xa = __builtin_bit_cast(unsigned, mx), xb = xa;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xa), "+v"(xb));
      const float mxr = fmaxf(__builtin_bit_cast(float, xa), __builtin_bit_cast(float, xb)) * scale_log2;
      float alpha = 1.f;
      pend = false;
      if (RT_RARE(__any(mxr - m_run > RESCALE_THR))) {
        asm volatile("" ::: "memory");           // keep this rare block a real branch (not if-converted)
        const float m_new = fmaxf(m_run, mxr);
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // first tile: exp2(-inf) = 0 on zeroed state
        m_run = m_new;
        l_run *= alpha;
        pend = true;
      }
      return alpha;
    };
    auto rescale_o = [&](float alpha) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
    };
    // numerators of 8 keys (one k-step of Oᵀ += Vᵀ·Pᵀ): registers 8·s2 .. 8·s2+7 of a half
    auto numer8 = [&](const f32x16& s, int s2, bf16x8& pf) {
      float ps = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[8 * s2 + j], scale_log2, -m_run));
        ps += p;
        pf[j] = (__bf16)p;
      }
      l_run += ps;
    };

#define RT_SB() __builtin_amdgcn_sched_barrier(0)
// Opaque use+def: the value must exist HERE (keeps the optimiser from sinking a step's vector work below a later branch)
#define RT_PIN(v) asm volatile("" : "+v"(v))
// n groups of { 1 MFMA, nds LDS reads, nva vector/transcendental ops }
#define RT_WEAVE(n, nds, nva)                                          \
  _Pragma("unroll") for (int w_ = 0; w_ < (n); ++w_) {                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 \
    if ((nds) > 0) __builtin_amdgcn_sched_group_barrier(0x100, (nds), 0); \
    if ((nva) > 0) __builtin_amdgcn_sched_group_barrier(0x402, (nva), 0); \
  }

    auto tile = [&](auto slot_c, int t) {
      constexpr int SLOT = decltype(slot_c)::value;
      constexpr int SB = SLOT * 2 * TILE_B;
      // wave-uniform, almost always false: the tile is the ragged last one of the row (keys >= S are masked) / the tile it
      // stages is (its rows are clamped) / there is no next tile in this run
      const bool ragged = (t + 1) * BKV > S;
      auto kread = [&](int h, int ks) -> bf16x8 {
        return *(const __attribute__((address_space(3))) bf16x8*)(kp[ks] + SB + h * 8192); };
      auto vread = [&](int h, int s2, int dt) -> bf16x8 {
        const s16x4 lo = tr_read(vp[0][dt] + SB + h * 8192 + s2 * 4096);
        const s16x4 hi = tr_read(vp[1][dt] + SB + h * 8192 + s2 * 4096);
        return __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
      };
      auto mask_half = [&](f32x16& s, int h) {   // key of s{h}[r]: 64t + 32h + (r&3) + 8(r>>2) + 4hh
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t * BKV + 32 * h + (r & 3) + 8 * (r >> 2) + 4 * hh >= S) s[r] = -INFINITY;
      };
      rt_dma_barrier();                          // tile t landed (every wave's vmcnt(0), then the barrier); the other slot is free
      f32x16 s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
      bf16x8 kA[8], kB[8], vA[4], vB[4];         // operand fragments, read one step ahead of the MFMAs that use them
      bf16x8 p00, p01, p10, p11;                 // numerators [half][k-step]
      bool pend;
      // ---- 1: Sᵀ(h0) = K[0:32]·Qᵀ (8 MFMAs) beside the LDS-DMA issue of tile t+1 (one 1-KiB piece behind each MFMA: the asm
      //         statements are placed by hand, sched_group_barrier cannot see inside them) and the first reads of h1
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) kA[ks] = kread(0, ks);
      const bool next_ragged = (t + 2) * BKV > S;
      if (RT_RARE(t + 1 < te && next_ragged)) stage(SLOT ^ 1, t + 1, true);       // clamped rows: all eight pieces at once, out of line
      const bool weave = t + 1 < te && !next_ragged;
      {
        const uint32_t nb = lds0 + (SLOT ^ 1) * 2 * TILE_B, so = (uint32_t)((t + 1) * tile_stride_b);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kA[ks], qf[ks], s0, 0, 0, 0);
          if (RT_USUAL(weave)) rt_dma16_asm((ks & 1) ? rsrcV : rsrcK, nb + ((ks & 1) ? TILE_B : 0) + (wave * 4 + (ks >> 1)) * 1024, soff[ks >> 1], so);
          if (ks >= 4 && ks < 7) kB[ks - 4] = kread(1, ks - 4);
          RT_SB();
        }
      }
      // ---- 2: first 3 MFMAs of Sᵀ(h1) cover the retirement of Sᵀ(h0); row max of h0, decision (no numerators pending)
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kB[ks], qf[ks], s1, 0, 0, 0);
#pragma unroll
      for (int ks = 3; ks < 8; ++ks) kB[ks] = kread(1, ks);
      if (RT_RARE(ragged)) mask_half(s0, 0);
      float mx0 = half_max(s0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      RT_SB();
      {
        const float alpha = decide(mx0, pend);
        if (RT_RARE(pend)) rescale_o(alpha);
      }
      RT_SB();
      // ---- 3: rest of Sᵀ(h1) ∥ numerators of h0, k-step 0; Vᵀ fragments of (h0, k-step 0) come in
#pragma unroll
      for (int ks = 3; ks < 8; ++ks) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kB[ks], qf[ks], s1, 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vA[dt] = vread(0, 0, dt);
      numer8(s0, 0, p00);
      RT_WEAVE(5, 2, 6)
      RT_PIN(p00); RT_PIN(l_run);
      RT_SB();
      // ---- 4: Oᵀ += Vᵀ(h0, k-step 0)·p00 ∥ numerators of h0 k-step 1, row max of h1
      // element j of lane-half hh of k-step (h, s2) is key 32h + 16s2 + 8(j>>2) + 4hh + (j&3)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vA[dt], p00, o_acc[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vB[dt] = vread(0, 1, dt);
      numer8(s0, 1, p01);
      if (RT_RARE(ragged)) mask_half(s1, 1);
      float mx1 = half_max(s1);
      RT_WEAVE(4, 2, 9)
      RT_PIN(p01); RT_PIN(l_run); RT_PIN(mx1);
      RT_SB();
      const float alpha1 = decide(mx1, pend);    // p01 (old scale) is still to be multiplied into O: O is rescaled after step 5
      RT_SB();
      // ---- 5: Oᵀ += Vᵀ(h0, k-step 1)·p01 ∥ numerators of h1 k-step 0 (against the new max)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vB[dt], p01, o_acc[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vA[dt] = vread(1, 0, dt);
      numer8(s1, 0, p10);
      RT_WEAVE(4, 2, 7)
      RT_PIN(p10); RT_PIN(l_run);
      RT_SB();
      if (RT_RARE(pend)) rescale_o(alpha1);
      RT_SB();
      // ---- 6: Oᵀ += Vᵀ(h1, k-step 0)·p10 ∥ numerators of h1 k-step 1
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vA[dt], p10, o_acc[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vB[dt] = vread(1, 1, dt);
      numer8(s1, 1, p11);
      RT_WEAVE(4, 2, 7)
      RT_PIN(p11); RT_PIN(l_run);
      RT_SB();
      // ---- 7: Oᵀ += Vᵀ(h1, k-step 1)·p11
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vB[dt], p11, o_acc[dt], 0, 0, 0);
    };

    // LDS slot = parity of (t - tb)
    int t = tb;
    for (; t + 1 < te; t += 2) {
      tile(S0{}, t);
      tile(S1{}, t + 1);
    }
    if (t < te) tile(S0{}, t);

    const int qrow = q0 + wave * 32 + l31;
    const bool whole = (tb == 0 && te == ntiles);
    if (!whole) {
      // ---- partial: write (Oᵀ, m, l) of this run through to memory, take a ticket, last ticket combines the item
      const int ritem = item - (cut.start + cut.nfull);            // index among the split items of this group
      char* rec = records + ((((int64_t)b * 8 + xcd) * G.spx + jpart) * 2 + seg) * (int64_t)REC_B + wave * REC_WAVE_B;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rec, 0, REC_WAVE_B, 0x00020000);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 vv = {o_acc[i][4 * g + 0], o_acc[i][4 * g + 1], o_acc[i][4 * g + 2], o_acc[i][4 * g + 3]};
          const u32x4 w = __builtin_bit_cast(u32x4, vv);
          __builtin_amdgcn_raw_buffer_store_b128(w, rs, (i * 4 + g) * 1024 + lane * 16, 0, 16 /* sc1: write-through */);
        }
      {
        const float2 mlf = {m_run, l_run};
        const u32x2 ml = __builtin_bit_cast(u32x2, mlf);
        __builtin_amdgcn_raw_buffer_store_b64(ml, rs, 16384 + lane * 8, 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave drains its write-through stores
      __syncthreads();
      int* cnt = counters + (int64_t)b * G.NI + item;
      // which splitting workgroups cover this item (same closed forms on every workgroup; 32-bit scalar arithmetic)
      const unsigned U = (unsigned)cut.rem * (unsigned)ntiles, spx = (unsigned)G.spx;
      const unsigned a = (unsigned)ritem * (unsigned)ntiles, bnd = a + (unsigned)ntiles;
      const int j_first = (int)(((a + 1) * spx + U - 1) / U) - 1;
      const int j_last = min(G.spx - 1, (int)((bnd * spx + U - 1) / U) - 1);
      const int nparts = j_last - j_first + 1;
      volatile int* flag = reinterpret_cast<volatile int*>(smem);
      if (tid == 0) {
        const int old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old == nparts - 1) ? 1 : 0;
        if (last) {
          __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                         // drop this CU's stale lines
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        *flag = last;
      }
      __syncthreads();
      const int last = __builtin_amdgcn_readfirstlane(*flag);
      if (last) {
        // record of splitting workgroup j for this item: its second segment when the item starts after the run does
        auto rec_of = [&](int j) -> const char* {
          const unsigned lo_j = (unsigned)j * U / spx;
          const int sj = (a > lo_j) ? 1 : 0;
          return records + ((((int64_t)b * 8 + xcd) * G.spx + j) * 2 + sj) * (int64_t)REC_B + wave * REC_WAVE_B;
        };
        // pass 1: common max; pass 2: weighted sum in run order
        float M = -INFINITY;
        for (int j = j_first; j <= j_last; ++j) {
          const char* rj = rec_of(__builtin_amdgcn_readfirstlane(j));
          M = fmaxf(M, *reinterpret_cast<const float*>(rj + 16384 + lane * 8));
        }
        float L = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
        for (int j = j_first; j <= j_last; ++j) {
          const char* rj = rec_of(__builtin_amdgcn_readfirstlane(j));
          const float2 ml = *reinterpret_cast<const float2*>(rj + 16384 + lane * 8);
          const float wgt = __builtin_amdgcn_exp2f(ml.x - M);
          L = __builtin_fmaf(ml.y, wgt, L);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(rj + (i * 4 + g) * 1024 + lane * 16);
#pragma unroll
              for (int e = 0; e < 4; ++e) o_acc[i][4 * g + e] = __builtin_fmaf(v[e], wgt, o_acc[i][4 * g + e]);
            }
        }
        l_run = L;
      }
      if (!last) continue;
    }

    // ---- epilogue: O[q][d] = Oᵀ / l ; lane holds q = l31, d = 32dt + (r&3) + 8(r>>2) + 4hh
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
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
}

int g_slots_per_xcd = 0;   // 2 workgroups per CU, CUs / 8 per XCD group (queried once)

int slots_per_xcd() {
  if (g_slots_per_xcd == 0) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      int v = 0;
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const char* e = getenv("RT_ATTN_SLOTS_PER_XCD");   // A/B and tests only
    g_slots_per_xcd = e ? atoi(e) : (cus / 8) * 2;
    if (g_slots_per_xcd < 1) g_slots_per_xcd = 1;
  }
  return g_slots_per_xcd;
}

AttnGeom make_geom(int S, int H, bool want_split) {
  AttnGeom G;
  G.S = S;
  G.H = H;
  G.nqb = (S + BQ - 1) / BQ;
  G.ntiles = (S + BKV - 1) / BKV;
  G.NI = H * G.nqb;
  G.spx = slots_per_xcd();
  G.split = want_split ? 1 : 0;
  return G;
}

}  // namespace

// Workspace of the key-split tail for (B, S, H) on the current device: item counters (zero before first use; the kernel
// leaves them zero) followed by the partial records. 0 when no item of this shape would be split.
int64_t rt_attention_v3_ws_bytes(int32_t B, int32_t S, int32_t H);
extern "C" int64_t rt_attention_ws_bytes(int32_t B, int32_t S, int32_t H) {
  if (B < 1 || S < 1 || H < 1) return 0;
  static const bool off = getenv("RT_ATTN_SPLIT") && getenv("RT_ATTN_SPLIT")[0] == '0';
  if (off) return 0;
  const int64_t need3 = rt_attention_v3_ws_bytes(B, S, H);     // the larger of the two kernels' needs: either may serve the call
  const AttnGeom G = make_geom(S, H, true);
  bool any = false;
  for (int x = 0; x < 8; ++x) any = any || group_cut(G, x).rem > 0;
  if (!any) return need3;
  const int64_t cnt_b = (((int64_t)B * G.NI * 4 + CNT_ALIGN - 1) / CNT_ALIGN) * CNT_ALIGN;
  const int64_t need = cnt_b + (int64_t)B * 8 * G.spx * 2 * REC_B;
  return need > need3 ? need : need3;
}

// csrc/attention_v3.hip: the one-wave-per-SIMD kernel (64 query rows per wave); takes the launch when S % 256 == 0
int rt_attention_v3_try(const void* q, const void* k, const void* v, void* o, int64_t ld, int64_t stride_b, int64_t ldo, int64_t stride_ob,
                        int32_t B, int32_t S, int32_t H, float scale, void* ws, int64_t ws_bytes, void* stream);
int rt_attention_v3_mode(int mode);
extern "C" int rt_attention_variant(int32_t mode) { return rt_attention_v3_mode(mode); }

extern "C" int rt_attention_fwd(const void* q, const void* k, const void* v, void* o, int64_t ld, int64_t stride_b,
                                int64_t ldo, int64_t stride_ob, int32_t B, int32_t S, int32_t H, float scale,
                                void* ws, int64_t ws_bytes, void* stream) {
  if (!q || !k || !v || !o || B < 1 || S < 1 || H < 1) return RT_E_BADARG;
  if (!RT_ALIGNED(q, 16) || !RT_ALIGNED(k, 16) || !RT_ALIGNED(v, 16) || !RT_ALIGNED(o, 8) || ld % 8 || stride_b % 8 ||
      ldo % 4 || stride_ob % 4)
    return RT_E_ALIGN;
  if (ld < (int64_t)H * DH || ldo < (int64_t)H * DH) return RT_E_SHAPE;
  if ((int64_t)(S + BKV) * ld * 2 >= (int64_t)1 << 31) return RT_E_SHAPE;      // per-tile byte offsets are 32-bit
  if ((int64_t)((S + BQ - 1) / BQ) * H * ((S + BKV - 1) / BKV + 1) * slots_per_xcd() >= ((int64_t)1 << 31)) return RT_E_SHAPE;   // 32-bit run arithmetic
  {
    const int r = rt_attention_v3_try(q, k, v, o, ld, stride_b, ldo, stride_ob, B, S, H, scale, ws, ws_bytes, stream);
    if (r != 0) return r == 1 ? RT_OK : r;
  }
  const int64_t need = rt_attention_ws_bytes(B, S, H);
  const bool split = ws != nullptr && need > 0;
  if (split && (ws_bytes < need || !RT_ALIGNED(ws, 256))) return RT_E_BADARG;
  const AttnGeom G = make_geom(S, H, split);
  int wmax = 0;
  for (int x = 0; x < 8; ++x) wmax = group_slots(G, x) > wmax ? group_slots(G, x) : wmax;
  const int64_t cnt_b = (((int64_t)B * G.NI * 4 + CNT_ALIGN - 1) / CNT_ALIGN) * CNT_ALIGN;
  const dim3 grid(8 * wmax, B);
  hipLaunchKernelGGL(attention_fwd_kernel, grid, dim3(ATT_THREADS), 0, (hipStream_t)stream, (const bf16_t*)q,
                     (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, ld, stride_b, ldo, stride_ob,
                     scale * 1.4426950408889634f, G, split ? (int*)ws : nullptr, split ? (char*)ws + cnt_b : nullptr);
  return rt_hip_status();
}
