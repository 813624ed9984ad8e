// rt_gemm_bf16 — C = epilogue(A·Wᵀ + bias): bf16 operands, fp32 accumulate on v_mfma_f32_16x16x32_bf16.
//
// Tiling (CDNA4, wave64): 256×256 output tile per 512-thread workgroup (8 waves as 2(M)×4(N), each wave
// 128×64 = 8×4 MFMA fragments), BK = 64. Both operands are K-contiguous ([M][K] and [N][K]), so both are
// staged with LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction = 8 rows × 128 B) into a
// double-buffered 2×(32+32) KiB LDS image and read back with ds_read_b128.
//
// LDS image: row r of a tile is 128 B = eight 16-B chunks; logical chunk c is stored at physical chunk
// c ^ ((r>>1)&7). LDS-DMA writes lane-linear, so the permutation is applied to each lane's SOURCE address
// and again on the read (both sides or neither). With it every ds_read_b128 lane group of the 16x16x32
// fragment read (16 rows × 4 chunks) hits 16 distinct 16-B slots of the 256-B bank row: conflict-free.
//
// The MFMA is issued "swapped": W rows are the A operand, activation rows the B operand, so D[i][j] has
// i = output column n and j = output row m. Each lane then owns 4 CONSECUTIVE output columns of one row:
// bias/gate/residual loads and the store are 8-byte (bf16) or 16-byte (f32) vectors.
//
// Reference math: torch.nn.functional.linear as reached from controlnet_flux.py:277,280,292,386,391 and the
// diffusers blocks; epilogue terms documented in include/reptext_hip.h.
#include "rt_common.h"
#include <stdlib.h>
#include <initializer_list>
#include <type_traits>

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int THREADS = 512;
constexpr int TILE_BYTES = BM * BK * 2;           // 32 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;         // A + W
constexpr int LDS_BYTES = 2 * BUF_BYTES;          // double buffered: 128 KiB

struct GroupDev {
  rt_gemm_group g;
  int tiles_m, tiles_n, tile_begin;
  int wide_store;      // bf16 output rows are 16-byte addressable in 8-column steps (C 16-B aligned, ldc/strideC/N % 8 == 0)
};
struct Launch {
  GroupDev grp[RT_GEMM_MAX_GROUPS];
  int ngroups;
};

// Epilogue for one wave: 8x4 accumulator fragments -> C. Lane owns rows mrow + 16i (i<8) and, per fragment column j,
// the 4 consecutive columns ncol + 16j .. +3. Column-only terms (bias, gate) are loaded ONCE for the 4 fragment columns;
// per-row terms (residual, add2) are fetched one row ahead of the row being finished, so no store waits on a load.
template <bool OUT_F32, bool FP8 = false>
__device__ __forceinline__ void epilogue_tile(const rt_gemm_group& g, int bidx, int mrow, int ncol, f32x4 (&acc)[8][4], bool wide = false) {
  const int rpb = g.rows_per_batch > 0 ? g.rows_per_batch : g.M;
  bool nok[4];
  f32x4 bias4[4], gate4[4];
  f32x4 wsc4[4];                                        // fp8 only: de-quantisation scale of the 4 output channels
  if constexpr (FP8) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = ncol + 16 * j;
      wsc4[j] = g.w_scale ? *reinterpret_cast<const f32x4*>(g.w_scale + (n < g.N ? n : 0)) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = ncol + 16 * j;
    nok[j] = n < g.N;
    const int nc = nok[j] ? n : 0;                    // clamp: masked columns read column 0, never stored
    bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (g.bias) {
      const u32x2 b = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(g.bias) + nc);
      bias4[j] = f32x4{bf16lo(b[0]), bf16hi(b[0]), bf16lo(b[1]), bf16hi(b[1])};
    }
  }
  // gate depends on (batch, row / rpb): constant over the tile unless the tile straddles batch entries (rpb < M)
  // e4m3 instantiation: 16 more registers hold the de-quantisation scales, so the tile-constant gate vectors are re-read per
  // row there (L1 hits) instead of being kept — that was the 4-VGPR spill of gemm_pp_kernel<true>.
  const bool gate_per_row = g.gate && ((rpb < g.M) || FP8);
  if (g.gate && !gate_per_row) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      gate4[j] = *reinterpret_cast<const f32x4*>(g.gate + (int64_t)bidx * g.gate_ld + (nok[j] ? ncol + 16 * j : 0));
  }
  typedef typename std::conditional<OUT_F32, f32x4, u32x2>::type res_t;
  res_t rnext[4];
  u32x2 anext[4];
  auto fetch_row = [&](int i) {
    const int m = min(mrow + 16 * i, g.M - 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nc = nok[j] ? ncol + 16 * j : 0;
      if (g.res) {
        const int64_t roff = (int64_t)bidx * g.strideR + (int64_t)m * g.ldr + nc;
        if (OUT_F32) rnext[j] = *reinterpret_cast<const res_t*>(reinterpret_cast<const float*>(g.res) + roff);
        else rnext[j] = *reinterpret_cast<const res_t*>(reinterpret_cast<const bf16_t*>(g.res) + roff);
      }
      if (g.add2) anext[j] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(g.add2) + (int64_t)bidx * g.stride2 + (int64_t)m * g.ld2 + nc);
    }
  };
  if (g.res || g.add2) fetch_row(0);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = mrow + 16 * i;
    res_t rcur[4];
    u32x2 acur[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { rcur[j] = rnext[j]; acur[j] = anext[j]; }
    if ((g.res || g.add2) && i + 1 < 8) fetch_row(i + 1);
    const int mc = min(m, g.M - 1);
    const float rs = g.rowscale ? g.rowscale[(int64_t)bidx * g.stride_rowscale + mc % rpb] * g.alpha : g.alpha;
    u32x2 packed[4];                                   // bf16 results of the 4 fragment columns (wide-store path)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = ncol + 16 * j;
      f32x4 v;
      if constexpr (FP8) v = acc[i][j] * (wsc4[j] * (g.a_scale ? g.a_scale[(int64_t)bidx * g.M + mc] : 1.f)) + bias4[j];
      else v = acc[i][j] + bias4[j];
      if (n >= g.gelu_from) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(v[e]);
      }
      if (g.gate) {
        if (gate_per_row) {
          const int gb = bidx * (g.M / rpb) + mc / rpb;
          v *= *reinterpret_cast<const f32x4*>(g.gate + (int64_t)gb * g.gate_ld + (nok[j] ? n : 0));
        } else {
          v *= gate4[j];
        }
      }
      v *= rs;
      if (g.res) {
        if constexpr (OUT_F32) v += rcur[j];
        else { v[0] += bf16lo(rcur[j][0]); v[1] += bf16hi(rcur[j][0]); v[2] += bf16lo(rcur[j][1]); v[3] += bf16hi(rcur[j][1]); }
      }
      if (g.add2) { v[0] += bf16lo(acur[j][0]); v[1] += bf16hi(acur[j][0]); v[2] += bf16lo(acur[j][1]); v[3] += bf16hi(acur[j][1]); }
      if (m < g.M && nok[j]) {
        const int64_t coff = (int64_t)bidx * g.strideC + (int64_t)m * g.ldc + n;
        if constexpr (OUT_F32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + coff) = v;
        } else {
          u32x2 o;
          o[0] = pack_bf16x2(v[0], v[1]);
          o[1] = pack_bf16x2(v[2], v[3]);
          if (!wide) *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(g.C) + coff) = o;
        }
      }
      if constexpr (!OUT_F32) {
        packed[j][0] = pack_bf16x2(v[0], v[1]);
        packed[j][1] = pack_bf16x2(v[2], v[3]);
      }
    }
    if constexpr (!OUT_F32) {
      // Wide store: the four 16-lane groups c of a wave hold columns 4c..4c+3 of each 16-column fragment — 8 bytes per lane,
      // 32 stores per tile. v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of another;
      // applied to the registers of fragments (j, j+1) it leaves every lane with 8 CONSECUTIVE columns (its own 4 plus its
      // neighbour group's): groups 0/2 take fragment j, groups 1/3 fragment j+1. Same bytes, same addresses, half the store
      // instructions (the store tail of a tile is issue-bound).
      if (wide) {
        const int c = (ncol >> 2) & 3;                // this lane's 16-lane group
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
          const auto x = __builtin_amdgcn_permlane16_swap(packed[jp][0], packed[jp + 1][0], false, false);
          const auto y = __builtin_amdgcn_permlane16_swap(packed[jp][1], packed[jp + 1][1], false, false);
          const int n8 = (ncol - 4 * c) + 16 * (jp + (c & 1)) + 8 * (c >> 1);       // first of this lane's 8 columns
          if (m < g.M && n8 < g.N) {
            const int64_t coff = (int64_t)bidx * g.strideC + (int64_t)m * g.ldc + n8;
            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(g.C) + coff) = u32x4{x[0], y[0], x[1], y[1]};
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// gemm_pp_kernel: the ping-pong schedule.
//   * a K-tile is consumed in 4 phases of 16 MFMAs, one quadrant of the wave's 128x64 output each, in the order
//     (a0,b0) (a0,b1) (a1,b1) (a1,b0) so only 12+4+8+0 fragment reads are needed per K-tile;
//   * waves 4-7 run one barrier behind waves 0-3, so of the two waves that share a SIMD one is in its MFMA cluster
//     while the other reads fragments / issues LDS-DMA: the matrix pipe always has a wave feeding it;
//   * the operand parts a0,b0,b1,a1 of tile t+1 are issued (2 LDS-DMA per wave) in phases 1..4 of tile t and
//     consumed in the same order one tile later; they stay in flight ACROSS the barriers behind a counted
//     s_waitcnt vmcnt(4) (never 0 in steady state). Each wait sits before the barrier that precedes the MFMA
//     cluster, i.e. one barrier earlier than the first read of that data by EITHER wave group (the staggered group
//     reads one barrier later), which is what orders LDS-DMA writes for other waves' ds_reads.
// ---------------------------------------------------------------------------------------------------
#define RT_BAR()                              \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    __builtin_amdgcn_s_barrier();             \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)
#define RT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// FP8 = false: bf16 operands, two v_mfma_f32_16x16x32_bf16 k-steps per 64-element K-tile.
// FP8 = true : e4m3 operands, ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 128-element K-tile (block scales 1.0). A tile row is
//              128 bytes either way, so staging, LDS image, swizzle, fragment addresses and the barrier schedule are shared.
//              The MFMA's lane group j (lane>>4) is given the two 16-byte chunks j and j+4 of the row as its 32 k-values —
//              the chunks the bf16 form reads for its k-steps 0 and 1 — for both operands alike, so the contraction still
//              covers every k exactly once and the reads keep their conflict-free pattern.
template <bool FP8>
__global__ __launch_bounds__(THREADS, 2) void gemm_pp_kernel(const Launch L) {
  constexpr int ESZ = FP8 ? 1 : 2;                   // bytes per operand element
  constexpr int BKE = 128 / ESZ;                     // elements per K-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef const __attribute__((address_space(4))) Launch* LaunchPtr;
  LaunchPtr Lp = (LaunchPtr)__builtin_amdgcn_kernarg_segment_ptr();
  (void)L;
  // XCD-aware placement (speed only, never correctness): workgroups are dealt round-robin over the 8 XCDs, so
  // blocks b and b+8 share an L2. Remap so each XCD owns a CONTIGUOUS run of the tile order (bijective for any grid
  // size), then walk each problem in panels of 8 tile-columns, row by row: the 32 tiles an XCD runs at a time form
  // a ~4x8 patch that shares 4 A row-panels and 8 W row-panels in that XCD's L2 instead of ~9 and ~14.
  const int nwg = (int)gridDim.x;
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int q8 = nwg >> 3, r8 = nwg & 7;
  const int lin = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  int gi = 0;
#pragma unroll
  for (int i = 1; i < RT_GEMM_MAX_GROUPS; ++i)
    if (i < Lp->ngroups && lin >= Lp->grp[i].tile_begin) gi = i;
#if defined(__HIP_DEVICE_COMPILE__)
  const GroupDev G = Lp->grp[gi];
#else
  const GroupDev G = L.grp[0];
#endif
  const rt_gemm_group& g = G.g;
  int t = lin - G.tile_begin;
  const int tiles_per_batch = G.tiles_m * G.tiles_n;
  const int bidx = t / tiles_per_batch;
  t -= bidx * tiles_per_batch;
  constexpr int PANEL = 8;
  const int panel = t / (PANEL * G.tiles_m);
  const int pw = min(PANEL, G.tiles_n - panel * PANEL);          // width of this (possibly last, narrower) panel
  const int tp = t - panel * PANEL * G.tiles_m;
  const int tm = tp / pw;
  const int tn = panel * PANEL + (tp - tm * pw);
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- staging: part 0 = a0, 1 = b0, 2 = b1, 3 = a1; each wave stages 16 rows (2 pieces of 8) of every part.
  //      a-part rows r' in [0,128): tile row = r' + 64*ah + (r' >= 64 ? 64 : 0)      (rows of both wave rows' half ah)
  //      b-part rows r' in [0,128): tile row = (r'/32)*64 + 32*bh + r'%32            (rows of all four wave cols' half bh)
  const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)bidx * g.strideA * ESZ;
  const char* Wb = reinterpret_cast<const char*>(g.W);
  uint32_t src[4][2];     // byte offsets from Ab / Wb (both tensors are < 2^32 bytes; checked on the host)
  int lds_off[4][2];
#pragma unroll
  for (int part = 0; part < 4; ++part) {
    const bool is_a = (part == 0 || part == 3);
    const int half = (part == 2 || part == 3) ? 1 : 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int rowbase;   // wave-uniform first row of this 8-row piece inside its operand tile
      if (is_a) rowbase = wave * 16 + q * 8 + 64 * half + (wave >= 4 ? 64 : 0);
      else rowbase = (wave >> 1) * 64 + 32 * half + (wave & 1) * 16 + q * 8;
      const int row = rowbase + (lane >> 3);
      const int lc = (lane & 7) ^ ((row >> 1) & 7);
      if (is_a) {
        const int am = min(m0 + row, g.M - 1);
        src[part][q] = (uint32_t)(((int64_t)am * g.lda) * ESZ + lc * 16);
        lds_off[part][q] = rowbase * 128;
      } else {
        const int wr = min(n0 + row, g.N - 1);
        src[part][q] = (uint32_t)(((int64_t)wr * g.ldw) * ESZ + lc * 16);
        lds_off[part][q] = TILE_BYTES + rowbase * 128;
      }
    }
  }
  // LDS-DMA by buffer_load ... lds: 4-SGPR descriptor per operand + the lane's invariant 32-bit byte offset + the K offset in an
  // SGPR — no per-K-tile vector address arithmetic and half the address registers of the global_load form.
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Ab), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wb), 0, -1, 0x00020000);
  auto issue = [&](int part, int buf, int koff) {                        // koff in BYTES along the row
#pragma unroll
    for (int q = 0; q < 2; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds((part == 0 || part == 3) ? rsrcA : rsrcW, LDS_PTR(smem + buf * BUF_BYTES + lds_off[part][q]), 16,
                                               (int)src[part][q], koff, 0, 0);
  };

  const int l15 = lane & 15;
  const int sw = (lane >> 1) & 7;
  const int rd0 = l15 * 128 + (((0 + (lane >> 4)) ^ sw) << 4);
  const int rd1 = l15 * 128 + (((4 + (lane >> 4)) ^ sw) << 4);
  const int a_base = wm * 128 * 128;
  const int w_base = TILE_BYTES + wn * 64 * 128;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2];        // current a-half: [frag][kk]
  bf16x8 wf[2][2][2];     // both b-halves: [half][frag][kk]

#define RT_READ_A(ah)                                                                                             \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                 \
    af[i][0] = *reinterpret_cast<const bf16x8*>(tb + a_base + ((ah)*4 + i) * 2048 + rd0);                          \
    af[i][1] = *reinterpret_cast<const bf16x8*>(tb + a_base + ((ah)*4 + i) * 2048 + rd1);                          \
  }
#define RT_READ_B(bh)                                                                                             \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                 \
    wf[bh][j][0] = *reinterpret_cast<const bf16x8*>(tb + w_base + ((bh)*2 + j) * 2048 + rd0);                      \
    wf[bh][j][1] = *reinterpret_cast<const bf16x8*>(tb + w_base + ((bh)*2 + j) * 2048 + rd1);                      \
  }
#define RT_CAT8(lo, hi) __builtin_shufflevector(__builtin_bit_cast(i32x4, lo), __builtin_bit_cast(i32x4, hi), 0, 1, 2, 3, 4, 5, 6, 7)
#define RT_MFMA(ah, bh)                                                                                           \
  do {                                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    if constexpr (FP8) {                                                                                          \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                               \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
          acc[(ah)*4 + i][(bh)*2 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                         \
              RT_CAT8(wf[bh][j][0], wf[bh][j][1]), RT_CAT8(af[i][0], af[i][1]), acc[(ah)*4 + i][(bh)*2 + j],      \
              0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 0x7F7F7F7F /* 2^0 */, 0, 0x7F7F7F7F);                            \
    } else {                                                                                                      \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                            \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
          _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
            acc[(ah)*4 + i][(bh)*2 + j] =                                                                         \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[bh][j][kk], af[i][kk], acc[(ah)*4 + i][(bh)*2 + j], 0, 0, 0); \
    }                                                                                                             \
    __builtin_amdgcn_s_setprio(0);                                                                                \
  } while (0)

  const int nk = g.K / BKE;
  issue(0, 0, 0); issue(1, 0, 0); issue(2, 0, 0); issue(3, 0, 0);
  RT_VMCNT(4);
  RT_BAR();
  if (wm == 1) RT_BAR();                       // stagger waves 4-7 by one barrier

  for (int kt = 0; kt + 1 < nk; ++kt) {
    const char* tb = smem + (kt & 1) * BUF_BYTES;
    const int nb = (kt & 1) ^ 1;
    const int koff = (kt + 1) * 128;           // bytes
    // ---- phase 1: (a0,b0)
    RT_READ_A(0); RT_READ_B(0);
    issue(0, nb, koff);
    RT_VMCNT(4);                               // b1(kt) landed (younger: a1(kt), a0(kt+1))
    RT_BAR(); RT_MFMA(0, 0); RT_BAR();
    // ---- phase 2: (a0,b1)
    RT_READ_B(1);
    issue(1, nb, koff);
    RT_VMCNT(4);                               // a1(kt) landed (younger: a0(kt+1), b0(kt+1))
    RT_BAR(); RT_MFMA(0, 1); RT_BAR();
    // ---- phase 3: (a1,b1)
    RT_READ_A(1);
    issue(2, nb, koff);
    RT_BAR(); RT_MFMA(1, 1); RT_BAR();
    // ---- phase 4: (a1,b0)  (b0 still in registers)
    issue(3, nb, koff);
    RT_VMCNT(4);                               // a0(kt+1), b0(kt+1) landed (younger: b1(kt+1), a1(kt+1))
    RT_BAR(); RT_MFMA(1, 0); RT_BAR();
  }
  {                                            // last K-tile: nothing left to issue, drain
    const char* tb = smem + ((nk - 1) & 1) * BUF_BYTES;
    RT_READ_A(0); RT_READ_B(0);
    RT_VMCNT(0);
    RT_BAR(); RT_MFMA(0, 0); RT_BAR();
    RT_READ_B(1);
    RT_BAR(); RT_MFMA(0, 1); RT_BAR();
    RT_READ_A(1);
    RT_BAR(); RT_MFMA(1, 1); RT_BAR();
    RT_BAR(); RT_MFMA(1, 0); RT_BAR();
  }
  if (wm == 0) RT_BAR();                       // balance the stagger barrier
#undef RT_READ_A
#undef RT_READ_B
#undef RT_MFMA
#undef RT_CAT8

  const int mrow = m0 + wm * 128 + l15;
  const int ncol = n0 + wn * 64 + 4 * (lane >> 4);
  if (g.out_f32) epilogue_tile<true, FP8>(g, bidx, mrow, ncol, acc);
  else epilogue_tile<false, FP8>(g, bidx, mrow, ncol, acc, G.wide_store != 0);
}

}  // namespace

static int launch_gemm(const rt_gemm_group* groups, int32_t ngroups, void* stream, bool fp8) {
  if (!groups || ngroups < 1 || ngroups > RT_GEMM_MAX_GROUPS) return RT_E_BADARG;
  Launch L{};
  L.ngroups = ngroups;
  int total = 0;
  const int bke = fp8 ? 128 : BK;                     // elements per K-tile
  const int al = fp8 ? 16 : 8;                        // elements per 16 bytes
  const int64_t esz = fp8 ? 1 : 2;
  for (int i = 0; i < ngroups; ++i) {
    const rt_gemm_group& g = groups[i];
    if (!g.A || !g.W || !g.C || g.M < 1 || g.N < 1 || g.K < 1 || g.batch < 1) return RT_E_BADARG;
    if (g.K % bke != 0 || g.N % 4 != 0) return RT_E_SHAPE;
    if (g.rows_per_batch > 0 && g.M % g.rows_per_batch != 0) return RT_E_SHAPE;
    if (!RT_ALIGNED(g.A, 16) || !RT_ALIGNED(g.W, 16) || g.lda % al || g.ldw % al || g.strideA % al) return RT_E_ALIGN;
    if (g.lda < g.K || g.ldw < g.K || g.ldc < g.N) return RT_E_SHAPE;
    // staging offsets are 32-bit byte offsets from the (per-batch) operand base
    if (((int64_t)g.M * g.lda) * esz >= ((int64_t)1 << 32) || ((int64_t)g.N * g.ldw) * esz >= ((int64_t)1 << 32)) return RT_E_SHAPE;
    const int cal = g.out_f32 ? 16 : 8;
    if (!RT_ALIGNED(g.C, cal) || g.ldc % 4 || g.strideC % 4) return RT_E_ALIGN;
    if (g.res && (!RT_ALIGNED(g.res, cal) || g.ldr % 4 || g.strideR % 4)) return RT_E_ALIGN;
    if (g.add2 && (!RT_ALIGNED(g.add2, 8) || g.ld2 % 4 || g.stride2 % 4)) return RT_E_ALIGN;
    if (g.bias && !RT_ALIGNED(g.bias, 8)) return RT_E_ALIGN;
    if (g.gate && (!RT_ALIGNED(g.gate, 16) || g.gate_ld % 4)) return RT_E_ALIGN;
    if (fp8 && g.w_scale && !RT_ALIGNED(g.w_scale, 16)) return RT_E_ALIGN;
    L.grp[i].g = g;
    L.grp[i].tiles_m = (g.M + BM - 1) / BM;
    L.grp[i].tiles_n = (g.N + BN - 1) / BN;
    static const bool wide_on = !(getenv("RT_GEMM_WIDE_STORE") && getenv("RT_GEMM_WIDE_STORE")[0] == '0');   // A/B switch
    L.grp[i].wide_store = (wide_on && !g.out_f32 && RT_ALIGNED(g.C, 16) && g.ldc % 8 == 0 && g.strideC % 8 == 0 && g.N % 8 == 0) ? 1 : 0;
    L.grp[i].tile_begin = total;
    total += L.grp[i].tiles_m * L.grp[i].tiles_n * g.batch;
  }
  static bool attr_done = false;
  if (!attr_done) {
    for (const void* f : {reinterpret_cast<const void*>(gemm_pp_kernel<false>), reinterpret_cast<const void*>(gemm_pp_kernel<true>)}) {
      hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
      if (e != hipSuccess) return (int)e;
    }
    attr_done = true;
  }
  if (fp8) hipLaunchKernelGGL(gemm_pp_kernel<true>, dim3(total), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, L);
  else hipLaunchKernelGGL(gemm_pp_kernel<false>, dim3(total), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, L);
  return rt_hip_status();
}

extern "C" int rt_gemm_bf16(const rt_gemm_group* groups, int32_t ngroups, void* stream) {
  return launch_gemm(groups, ngroups, stream, false);
}

extern "C" int rt_gemm_fp8(const rt_gemm_group* groups, int32_t ngroups, void* stream) {
  return launch_gemm(groups, ngroups, stream, true);
}
