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

constexpr int BK = 64;
constexpr int THREADS = 512;

// Tile geometry. 8 waves as 2(M) x 4(N); a wave's output is (NI0+NI1) x (NJ0+NJ1) fragments of 16x16, consumed per K-tile in the
// four quadrants (a0,b0) (a0,b1) (a1,b1) (a1,b0) of its "halves" (a0 = first NI0 row fragments, b0 = first NJ0 column fragments).
//   Geo<4,4,2,2>  256x256  the general tile (128 accumulator registers per lane, 128 KiB of LDS)
//   Geo<5,4,2,1>  288x192  M = 4608 x N = 3072 is EXACTLY 256 of these (16 x 16): the single blocks' out-projection fills all 256
//                          CUs in one round instead of 216 tiles of 256x256 on 256 CUs (84 %); 108 accumulator registers, 136 KiB
//   Geo<4,4,2,1>  256x192  } narrower tiles for the TAIL of a multi-round launch (gemm_mix_kernel): the columns that would form a
//   Geo<4,4,1,1>  256x128  } partly filled last round of 256x256 tiles are cut into 3/4- or 1/2-width tiles instead
// The K order of every output element is the same in all of them, so results are bit-identical whatever tile computed them.
template <int NI0_, int NI1_, int NJ0_, int NJ1_>
struct Geo {
  static constexpr int NI0 = NI0_, NI1 = NI1_, NJ0 = NJ0_, NJ1 = NJ1_;
  static constexpr int NI = NI0 + NI1, NJ = NJ0 + NJ1;
  static constexpr int NIH = NI0 > NI1 ? NI0 : NI1, NJH = NJ0 > NJ1 ? NJ0 : NJ1;
  static constexpr int WMR = 16 * NI, WNC = 16 * NJ;          // rows / columns per wave
  static constexpr int BM = 2 * WMR, BN = 4 * WNC;
  static constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128;
  // LDS-DMA pieces (8 rows x 128 B = 1 KiB, one wave-instruction) per wave and operand part; an a-half of 4*NIh pieces that does
  // not divide by 8 waves is padded with dummy pieces (landing in a scratch region) so that every wave issues the same count
  static constexpr int PA0 = (4 * NI0 + 7) / 8, PA1 = (4 * NI1 + 7) / 8, PB0 = NJ0, PB1 = NJ1;
  static constexpr bool DUMMY = (4 * NI0) % 8 != 0 || (4 * NI1) % 8 != 0;
  static constexpr int DUMMY_BYTES = DUMMY ? 8 * 1024 : 0;
  static constexpr int BUF_BYTES = A_BYTES + W_BYTES + DUMMY_BYTES;
  static constexpr int LDS_BYTES = 2 * BUF_BYTES;             // double buffered
};
using Geo256 = Geo<4, 4, 2, 2>;
using Geo288 = Geo<5, 4, 2, 1>;
using Geo192 = Geo<4, 4, 2, 1>;
using Geo128 = Geo<4, 4, 1, 1>;
constexpr int LDS_MAX = Geo288::LDS_BYTES > Geo256::LDS_BYTES ? Geo288::LDS_BYTES : Geo256::LDS_BYTES;

struct GroupDev {
  rt_gemm_group g;
  int tiles_m, tiles_n, tile_begin;
  int wide_store;      // bf16 output rows are 16-byte addressable in 8-column steps (C 16-B aligned, ldc/strideC/N % 8 == 0)
  // gemm_mix_kernel only: columns [0, n_split) are cut into 256-wide tiles, [n_split, N) into narrow ones
  int n_split, tiles_n_narrow, narrow_begin;
};
struct Launch {
  GroupDev grp[RT_GEMM_MAX_GROUPS];
  int ngroups;
  int wide_total, narrow_total;      // gemm_mix_kernel: tiles of each kind over all groups
};

// Epilogue for one wave: NI x NJ accumulator fragments -> C. Lane owns rows mrow + 16i and, per fragment column j,
// the 4 consecutive columns ncol + 16j .. +3. Column-only terms (bias, gate) are loaded ONCE for the fragment columns;
// per-row terms (residual, add2) are fetched one row ahead of the row being finished, so no store waits on a load.
// CONV (rt_gemm_group::conv_ks > 0): rows are the pixels of a zero-haloed NHWC image in memory order; only interior pixels are stored
// (the halo must stay zero) — rt_conv_interior decides per row.
__device__ __forceinline__ bool rt_conv_interior(const rt_gemm_group& g, int p) {
  // p < 2^24 (host check): the float quotient is within one of the exact one
  int q = (int)((float)p * g.conv_inv_w2);
  int x = p - q * g.conv_w2;
  if (x < 0) { x += g.conv_w2; --q; } else if (x >= g.conv_w2) { x -= g.conv_w2; ++q; }
  int b = (int)((float)q * g.conv_inv_h2);
  int y = q - b * g.conv_h2;
  if (y < 0) y += g.conv_h2; else if (y >= g.conv_h2) y -= g.conv_h2;
  return x >= 1 && x <= g.conv_w2 - 2 && y >= 1 && y <= g.conv_h2 - 2;
}

template <bool OUT_F32, bool FP8, int NI, int NJ, bool CONV = false>
__device__ __forceinline__ void epilogue_tile(const rt_gemm_group& g, int bidx, int mrow, int ncol, f32x4 (&acc)[NI][NJ], bool wide = false) {
  const int rpb = g.rows_per_batch > 0 ? g.rows_per_batch : g.M;
  bool nok[NJ];
  f32x4 bias4[NJ], gate4[NJ];
  f32x4 wsc4[NJ];                                       // fp8 only: de-quantisation scale of the 4 output channels
  if constexpr (FP8) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int n = ncol + 16 * j;
      wsc4[j] = g.w_scale ? *reinterpret_cast<const f32x4*>(g.w_scale + (n < g.N ? n : 0)) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
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
    for (int j = 0; j < NJ; ++j)
      gate4[j] = *reinterpret_cast<const f32x4*>(g.gate + (int64_t)bidx * g.gate_ld + (nok[j] ? ncol + 16 * j : 0));
  }
  typedef typename std::conditional<OUT_F32, f32x4, u32x2>::type res_t;
  res_t rnext[NJ];
  u32x2 anext[NJ];
  auto fetch_row = [&](int i) {
    const int m = min(mrow + 16 * i, g.M - 1);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
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
  uint32_t sbw[(NJ + 1) / 2] = {};                     // MX output: scale bytes of four consecutive fragment rows per 32-column block
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    // CONV: a halo pixel's row is computed like any other and dropped here (m = M disables every store of the row)
    const int m = (CONV && !rt_conv_interior(g, mrow + 16 * i)) ? g.M : mrow + 16 * i;
    res_t rcur[NJ];
    u32x2 acur[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { rcur[j] = rnext[j]; acur[j] = anext[j]; }
    if ((g.res || g.add2) && i + 1 < NI) fetch_row(i + 1);
    const int mc = min(m, g.M - 1);
    const float rs = g.rowscale ? g.rowscale[(int64_t)bidx * g.stride_rowscale + mc % rpb] * g.alpha : g.alpha;
    u32x2 packed[NJ];                                  // bf16 results of the fragment columns (wide-store path)
    f32x4 vrow[NJ];                                    // e4m3 + block-scale output: the row's values before quantisation
    // MX output (rt_gemm_group::c8): tile-uniform, c8_from is a multiple of the tile width
    const bool to8 = FP8 && g.c8 != nullptr && ncol >= g.c8_from;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
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
      vrow[j] = v;
      if (m < g.M && nok[j] && !to8) {
        const int64_t coff = (int64_t)bidx * g.strideC + (int64_t)m * g.ldc + n;
        if constexpr (OUT_F32) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + coff) = v;
        } else {
          u32x2 o;
          o[0] = pack_bf16x2(v[0], v[1]);
          o[1] = pack_bf16x2(v[2], v[3]);
          if (!wide || (NJ % 2 == 1 && j == NJ - 1)) *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(g.C) + coff) = o;
        }
      }
      if constexpr (!OUT_F32) {
        packed[j][0] = pack_bf16x2(v[0], v[1]);
        packed[j][1] = pack_bf16x2(v[2], v[3]);
      }
    }
    if constexpr (FP8 && !OUT_F32 && NJ % 2 == 0) {
      // e4m3 + E8M0 block scales: a 32-column block = the fragment pair (jp, jp+1), spread over the four 16-lane rows of the wave
      // (4 columns of each fragment per lane). Block maximum = 8 values in the lane, then across the lane rows; every lane derives
      // the same scale byte, scales its 8 values by the exact power of two, converts (RNE) and — after the same v_permlane16_swap
      // as the wide bf16 store — owns 8 consecutive bytes. One lane per row writes the wave's scale bytes (2 per fragment pair).
      if (to8) {
        const int c = (ncol >> 2) & 3;
#pragma unroll
        for (int jp = 0; jp + 1 < NJ; jp += 2) {
          float am = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) am = fmaxf(am, fmaxf(fabsf(vrow[jp][e]), fabsf(vrow[jp + 1][e])));
          am = rt_max_over_lane_rows(am);
          const int sb = rt_mx_scale_byte(am);
          const float inv = rt_mx_inv_scale(sb);
          int w0 = 0, w1 = 0;
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(vrow[jp][0] * inv, vrow[jp][1] * inv, w0, false);
          w0 = __builtin_amdgcn_cvt_pk_fp8_f32(vrow[jp][2] * inv, vrow[jp][3] * inv, w0, true);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(vrow[jp + 1][0] * inv, vrow[jp + 1][1] * inv, w1, false);
          w1 = __builtin_amdgcn_cvt_pk_fp8_f32(vrow[jp + 1][2] * inv, vrow[jp + 1][3] * inv, w1, true);
          const auto x = __builtin_amdgcn_permlane16_swap((unsigned)w0, (unsigned)w1, false, false);
          const int n8 = (ncol - 4 * c) + 16 * (jp + (c & 1)) + 8 * (c >> 1);       // first of this lane's 8 columns
          const int k8 = n8 - g.c8_from;                                             // column of the consumer's A
          if (m < g.M && n8 < g.N)
            *reinterpret_cast<u32x2*>(g.c8 + (int64_t)bidx * g.stride_c8 + (int64_t)m * g.ldc8 + k8) = u32x2{x[0], x[1]};
          // scale bytes: in rt_mx_scale_offset's order the rows m, m+16, m+32, m+48 of a 64-row chunk (the four fragment rows of an
          // a-half of this lane) are consecutive bytes: gathered over four passes of the row loop and stored as ONE dword. Rows past M
          // inside the chunk get whatever their accumulators hold — padding rows of the scale tensor, never read for a stored result.
          sbw[jp >> 1] |= (uint32_t)sb << (8 * (i & 3));
          if ((i & 3) == 3) {
            const int kb = (ncol - 4 * c) + 16 * jp - g.c8_from;                     // first column of the block
            const int m4 = m - 48;                                                   // first of the four rows
            if (c == 0 && m4 < g.M && kb + g.c8_from < g.N)
              *reinterpret_cast<uint32_t*>(g.c_bscale + rt_mx_scale_offset((int64_t)bidx * g.c_bscale_rows + m4, kb + g.c_bscale_k0, g.c_bscale_plane)) = sbw[jp >> 1];
            sbw[jp >> 1] = 0;
          }
        }
      }
    }
    if constexpr (!OUT_F32) {
      // Wide store: the four 16-lane groups c of a wave hold columns 4c..4c+3 of each 16-column fragment — 8 bytes per lane.
      // v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of another; applied to the registers
      // of fragments (j, j+1) it leaves every lane with 8 CONSECUTIVE columns (its own 4 plus its neighbour group's): groups
      // 0/2 take fragment j, groups 1/3 fragment j+1. Same bytes, same addresses, half the store instructions (the store tail of
      // a tile is issue-bound). An odd last fragment column is stored narrow (above).
      if (wide && !to8) {
        const int c = (ncol >> 2) & 3;                // this lane's 16-lane group
#pragma unroll
        for (int jp = 0; jp + 1 < NJ; jp += 2) {
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
// gemm_tile: one output tile, the ping-pong schedule.
//   * a K-tile is consumed in 4 phases, one quadrant of the wave's output each, in the order (a0,b0) (a0,b1) (a1,b1) (a1,b0)
//     so only (NI0+NJ0) + NJ1 + NI1 + 0 fragment reads (x2 k-steps) are needed per K-tile;
//   * waves 4-7 run one barrier behind waves 0-3, so of the two waves that share a SIMD one is in its MFMA cluster
//     while the other reads fragments / issues LDS-DMA: the matrix pipe always has a wave feeding it;
//   * the operand parts a0,b0,b1,a1 of tile t+1 are issued (PA0, PB0, PB1, PA1 LDS-DMA pieces per wave) in phases 1..4 of tile t
//     and consumed in the same order one tile later; they stay in flight ACROSS the barriers behind counted s_waitcnt vmcnt
//     (never 0 in steady state). Each wait sits before the barrier that precedes the MFMA cluster, i.e. one barrier earlier
//     than the first read of that data by EITHER wave group (the staggered group reads one barrier later), which is what orders
//     LDS-DMA writes for other waves' ds_reads. The counts: a wait that must cover part X leaves exactly the pieces issued after
//     X outstanding (vmcnt counts in issue order).
// ---------------------------------------------------------------------------------------------------
#define RT_BAR()                              \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    __builtin_amdgcn_s_barrier();             \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)
template <int N>
__device__ __forceinline__ void rt_vmcnt() {
  static_assert(N >= 0 && N <= 8, "vmcnt immediate");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

// FP8 = false: bf16 operands, two v_mfma_f32_16x16x32_bf16 k-steps per 64-element K-tile.
// FP8 = true : e4m3 operands, ONE v_mfma_scale_f32_16x16x128_f8f6f4 per 128-element K-tile (block scales 1.0). A tile row is
//              128 bytes either way, so staging, LDS image, swizzle, fragment addresses and the barrier schedule are shared.
//              The MFMA's lane group j (lane>>4) is given the two 16-byte chunks j and j+4 of the row as its 32 k-values —
//              the chunks the bf16 form reads for its k-steps 0 and 1 — for both operands alike, so the contraction still
//              covers every k exactly once and the reads keep their conflict-free pattern.
//
// MX = true (e4m3 only): A carries one E8M0 block scale per 32 K-elements (rt_gemm_group::a_bscale). The 256 rows x 4 bytes a tile
//              needs per K-tile make 8 KiB per EIGHT K-tiles: every wave fetches one contiguous 1-KiB run of the scale plane (four
//              K-tiles of a 64-row chunk, rt_mx_scale_offset's order) with one 16-byte LDS-DMA piece, issued together with part a0 of
//              every eighth K-tile (so it is retired by the waits that retire a0) into a 2 x 8 KiB ring behind the operand buffers
//              (a 4-byte piece per pair of K-tiles cost 4.5 % at K = 15360: an LDS-DMA holds its wave's issue whatever its size; one
//              per eight K-tiles is a quarter of that). The chunk order
//              puts the scales of the four fragment rows of an a-half that a lane feeds with its K-block j into ONE dword: a lane
//              reads it with one ds_read_b32 per a-half and K-tile and the MFMA picks the fragment's byte by its op_sel immediate
//              (8 ds_read_u8 per K-tile instead cost +6 % at K = 15360, tools/mx_dev). The scale goes to the MFMA's second operand
//              (the activation rows; the weights keep scale 2^0). Measured (tools/mx_dev/probe_scales.py): in the instruction's own
//              K order lane group g holds k = 16g..16g+15 in its first four registers and 64+16g.. in the other four — the chunks
//              j and j+4 read above, so the hardware's K order IS the memory order — and the byte supplied by lane group b scales
//              K-block b (k = 32b..32b+31) of that lane's row, whichever lanes hold those values: lane (row, j) supplies block j.
//
// CONV = true (bf16 only): a stride-1 k x k convolution over a zero-haloed NHWC image as THIS GEMM (rt_gemm_group::conv_*): A = the
//              image as a [pixels incl. halo][Cin] matrix in memory order, W = [Cout][k*k][Cin], K = k*k*Cin. For output pixel p
//              (also a haloed position) tap (dy,dx) reads pixel p + (dy-1)(W+2) + (dx-1): the SAME row shift for every p, so K-tile kt
//              (64 channels c0.. of one tap) is the ordinary A tile with its rows shifted by a wave-uniform number of pixels — added
//              to the lanes' byte offsets (4 VALU adds per K-tile; the buffer descriptor's range check turns reads in front of or
//              behind the image, which only halo rows make, into zeros) — and c0 in the scalar offset. Halo rows are computed and
//              dropped in the epilogue (0.4-3 % of the rows). Everything else — tile, LDS image, ping-pong schedule, counted waits —
//              is the GEMM's; the K order of an output element is that of csrc/vae.hip's conv_nhwc_kernel (bit-identical results).
template <bool FP8, class G_, bool MX = false, bool CONV = false>
__device__ __forceinline__ void gemm_tile(const rt_gemm_group& g, int bidx, int m0, int n0, bool wide_store, char* smem) {
  using T = G_;
  static_assert(!MX || FP8, "block scales belong to the e4m3 form");
  static_assert(!CONV || !FP8, "the convolution form is bf16");
  constexpr int ESZ = FP8 ? 1 : 2;                   // bytes per operand element
  constexpr int BKE = 128 / ESZ;                     // elements per K-tile
  constexpr int NPC = T::PA0 + T::PB0 + T::PB1 + T::PA1;          // pieces per wave and K-tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- staging. LDS image: A tile rows in tile order (wave row wm at wm*WMR, its half 1 at +16*NI0), W tile rows likewise
  //      (wave column wn at wn*WNC, half 1 at +16*NJ0), 128 B per row, 16-byte chunk c of row r at c ^ ((r>>1)&7).
  //      Part a-half h = rows of BOTH wave rows' half h (4*NIh pieces of 8 rows), part b-half h = rows of all four wave columns'
  //      half h (8*NJh pieces); wave w issues pieces w, w+8, ... of every part.
  const char* Ab = reinterpret_cast<const char*>(g.A) + (int64_t)bidx * g.strideA * ESZ;
  const char* Wb = reinterpret_cast<const char*>(g.W);
  uint32_t src[NPC];      // byte offsets from Ab / Wb (both tensors are < 2^32 bytes; checked on the host)
  int lds_off[NPC];
  {
    int k = 0;
    auto a_piece = [&](int half, int q) {
      constexpr int dummy_base = T::A_BYTES + T::W_BYTES;
      const int nih = half ? T::NI1 : T::NI0;
      const int p = wave + 8 * q;
      const bool real = p < 4 * nih;
      const int pp = real ? p : p - 8 * ((p - 4 * nih) / 8 + 1);          // a dummy piece re-reads one of this wave's earlier pieces
      const int wmp = pp / (2 * nih), within = pp - wmp * 2 * nih;
      const int rowbase = wmp * T::WMR + (half ? 16 * T::NI0 : 0) + within * 8;
      const int row = rowbase + (lane >> 3);
      const int lc = (lane & 7) ^ ((row >> 1) & 7);
      const int am = min(m0 + row, g.M - 1);
      src[k] = (uint32_t)(((int64_t)am * g.lda) * ESZ + lc * 16);
      lds_off[k] = real ? rowbase * 128 : dummy_base + wave * 1024;
      ++k;
    };
    auto b_piece = [&](int half, int q) {
      const int njh = half ? T::NJ1 : T::NJ0;
      const int p = wave + 8 * q;                                           // < 8*njh always
      const int wnp = p / (2 * njh), within = p - wnp * 2 * njh;
      const int rowbase = wnp * T::WNC + (half ? 16 * T::NJ0 : 0) + within * 8;
      const int row = rowbase + (lane >> 3);
      const int lc = (lane & 7) ^ ((row >> 1) & 7);
      const int wr = min(n0 + row, g.N - 1);
      src[k] = (uint32_t)(((int64_t)wr * g.ldw) * ESZ + lc * 16);
      lds_off[k] = T::A_BYTES + rowbase * 128;
      ++k;
    };
#pragma unroll
    for (int q = 0; q < T::PA0; ++q) a_piece(0, q);
#pragma unroll
    for (int q = 0; q < T::PB0; ++q) b_piece(0, q);
#pragma unroll
    for (int q = 0; q < T::PB1; ++q) b_piece(1, q);
#pragma unroll
    for (int q = 0; q < T::PA1; ++q) a_piece(1, q);
  }
  // LDS-DMA by buffer_load ... lds: 4-SGPR descriptor per operand + the lane's invariant 32-bit byte offset + the K offset in an
  // SGPR — no per-K-tile vector address arithmetic and half the address registers of the global_load form.
  // CONV: num_records = the image's bytes, so a shifted row in front of / behind the image reads zeros instead of foreign memory
  const __amdgpu_buffer_rsrc_t rsrcA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Ab), 0, CONV ? (int)((int64_t)g.M * g.lda * ESZ) : -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wb), 0, -1, 0x00020000);
  // part 0 = a0, 1 = b0, 2 = b1, 3 = a1 (issue order inside a K-tile)
  // CONV: K-tile kt = channels c0.. of tap; the A parts take (row shift of the tap in bytes -> lane offsets, c0 -> scalar offset)
  int cv_shift = 0, cv_c0 = 0, cv_kx = 0;                               // of the K-tile being ISSUED
  auto conv_at = [&](int kt) {                                           // position the issue state on K-tile kt (kt = 0 or the next one)
    if constexpr (CONV) {
      if (kt == 0) {
        cv_c0 = 0;
        cv_shift = g.conv_ks == 3 ? -(g.conv_w2 + 1) * (int)g.lda * 2 : 0;
      } else {
        cv_c0 += 128;                                                    // bytes
        if (cv_c0 == g.conv_cin * 2) {                                   // next tap: +1 pixel, or from the row's third tap to the next row's first
          cv_c0 = 0;
          const bool wrap = ++cv_kx == 3;
          if (wrap) cv_kx = 0;
          cv_shift += (wrap ? g.conv_w2 - 2 : 1) * (int)g.lda * 2;
        }
      }
    }
  };
  auto issue = [&](auto part_c, int buf, int koff) {                     // koff in BYTES along the row
    constexpr int part = decltype(part_c)::value;
    constexpr int first = part == 0 ? 0 : part == 1 ? T::PA0 : part == 2 ? T::PA0 + T::PB0 : T::PA0 + T::PB0 + T::PB1;
    constexpr int cnt = part == 0 ? T::PA0 : part == 1 ? T::PB0 : part == 2 ? T::PB1 : T::PA1;
    constexpr bool isA = part == 0 || part == 3;
#pragma unroll
    for (int q = 0; q < cnt; ++q)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(isA ? rsrcA : rsrcW, LDS_PTR(smem + buf * T::BUF_BYTES + lds_off[first + q]), 16,
                                               (CONV && isA) ? (int)src[first + q] + cv_shift : (int)src[first + q], (CONV && isA) ? cv_c0 : koff, 0, 0);
  };
  // MX: the scale piece of K-tile octet `oct` -> ring slot oct & 1. Wave w copies the 1 KiB of K-tile w of the octet: the 256 bytes of
  // each of the tile's four 64-row chunks (lane group l >> 4 = chunk; 2 KiB apart in the plane), so that the LDS image is K-tile-major —
  // [K-tile & 15][chunk][256 B] over both slots — and a K-tile's offset is ONE shift-and-mask of kt (the chunk-major image needed nine
  // scalar operations per read: +3 % at K = 15360). Chunks past the last row are clamped (their rows are never stored).
  constexpr int SC_OFF = 2 * T::BUF_BYTES;
  // The base pointer and the plane size are detached from the kernel-argument struct by an opaque asm: otherwise hipcc, short of
  // SGPRs in this loop, re-materialises them from the spilled struct — 16 v_readlane_b32 in every odd K-tile (+9 % at K = 15360).
  uint32_t sc_lo = 0, sc_hi = 0;
  int sc_plane = 0, sc_src = 0;
  if constexpr (MX) {
    static_assert(T::BM == 256 && T::NI0 == 4 && T::NI1 == 4, "the scale chunk order is that of the 256-row tile");
    sc_lo = (uint32_t)reinterpret_cast<uintptr_t>(g.a_bscale);
    sc_hi = (uint32_t)(reinterpret_cast<uintptr_t>(g.a_bscale) >> 32);
    sc_plane = (int)g.a_bscale_plane;
    asm volatile("" : "+s"(sc_lo), "+s"(sc_hi), "+s"(sc_plane));
    const int64_t row_b = (int64_t)bidx * g.a_bscale_rows;             // multiple of 64 (host check), m0 is a multiple of 256
    const int chunk = (int)min((row_b + m0) / 64 + (lane >> 4), (row_b + g.M - 1) / 64);
    sc_src = chunk * 2048 + wave * 256 + (lane & 15) * 16;
  }
  auto issue_scales = [&](int oct) {
    if constexpr (MX) {
      const __amdgpu_buffer_rsrc_t rsrcS =
          __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint8_t*>(((uint64_t)sc_hi << 32) | sc_lo), 0, -1, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcS, LDS_PTR(smem + SC_OFF + (oct & 1) * 8192 + wave * 1024), 16, sc_src, oct * sc_plane, 0, 0);
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;

  const int l15 = lane & 15;
  const int sw = (lane >> 1) & 7;
  const int rd0 = l15 * 128 + (((0 + (lane >> 4)) ^ sw) << 4);
  const int rd1 = l15 * 128 + (((4 + (lane >> 4)) ^ sw) << 4);
  const int sc_rd = SC_OFF + wm * 512 + l15 * 16 + (lane >> 4) * 4;          // MX: + (K-tile & 15)*1024 + a-half*256
  const int a_base = wm * T::WMR * 128;
  const int w_base = T::A_BYTES + wn * T::WNC * 128;

  f32x4 acc[T::NI][T::NJ];
#pragma unroll
  for (int i = 0; i < T::NI; ++i)
#pragma unroll
    for (int j = 0; j < T::NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[T::NIH][2];        // current a-half: [frag][kk]
  bf16x8 wf[2][T::NJH][2];     // both b-halves: [half][frag][kk]
  int sc = 0x7F7F7F7F;         // MX: block scale bytes of the current a-half's four fragments (byte i = fragment i) for this K-tile

#define RT_NIH(ah) ((ah) ? T::NI1 : T::NI0)
#define RT_NJH(bh) ((bh) ? T::NJ1 : T::NJ0)
#define RT_READ_A(ah)                                                                                             \
  _Pragma("unroll") for (int i = 0; i < RT_NIH(ah); ++i) {                                                        \
    af[i][0] = *reinterpret_cast<const bf16x8*>(tb + a_base + ((ah)*T::NI0 + i) * 2048 + rd0);                     \
    af[i][1] = *reinterpret_cast<const bf16x8*>(tb + a_base + ((ah)*T::NI0 + i) * 2048 + rd1);                     \
  }
#define RT_READ_S(ah, kt_)                                                                                        \
  if constexpr (MX) sc = *reinterpret_cast<const int*>(smem + sc_rd + (((kt_) & 15) << 10) + (ah)*256);
#define RT_READ_B(bh)                                                                                             \
  _Pragma("unroll") for (int j = 0; j < RT_NJH(bh); ++j) {                                                        \
    wf[bh][j][0] = *reinterpret_cast<const bf16x8*>(tb + w_base + ((bh)*T::NJ0 + j) * 2048 + rd0);                 \
    wf[bh][j][1] = *reinterpret_cast<const bf16x8*>(tb + w_base + ((bh)*T::NJ0 + j) * 2048 + rd1);                 \
  }
#define RT_CAT8(lo, hi) __builtin_shufflevector(__builtin_bit_cast(i32x4, lo), __builtin_bit_cast(i32x4, hi), 0, 1, 2, 3, 4, 5, 6, 7)
/* one fragment row i of the quadrant; i is a literal: the scale byte of fragment i is picked by the op_sel IMMEDIATE */ \
#define RT_MFMA8_ROW(ah, bh, i)                                                                                   \
  if constexpr ((i) < RT_NIH(ah)) {                                                                               \
    _Pragma("unroll") for (int j = 0; j < RT_NJH(bh); ++j)                                                        \
      acc[(ah)*T::NI0 + (i)][(bh)*T::NJ0 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                 \
          RT_CAT8(wf[bh][j][0], wf[bh][j][1]), RT_CAT8(af[i][0], af[i][1]), acc[(ah)*T::NI0 + (i)][(bh)*T::NJ0 + j], \
          0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 0x7F7F7F7F /* weights: 2^0 */, MX ? ((i) & 3) : 0, MX ? sc : 0x7F7F7F7F); \
  }
#define RT_MFMA(ah, bh)                                                                                           \
  do {                                                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                                \
    if constexpr (FP8) {                                                                                          \
      RT_MFMA8_ROW(ah, bh, 0); RT_MFMA8_ROW(ah, bh, 1); RT_MFMA8_ROW(ah, bh, 2); RT_MFMA8_ROW(ah, bh, 3); RT_MFMA8_ROW(ah, bh, 4); \
    } else {                                                                                                      \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                            \
        _Pragma("unroll") for (int i = 0; i < RT_NIH(ah); ++i)                                                    \
          _Pragma("unroll") for (int j = 0; j < RT_NJH(bh); ++j)                                                  \
            acc[(ah)*T::NI0 + i][(bh)*T::NJ0 + j] =                                                               \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[bh][j][kk], af[i][kk], acc[(ah)*T::NI0 + i][(bh)*T::NJ0 + j], 0, 0, 0); \
    }                                                                                                             \
    __builtin_amdgcn_s_setprio(0);                                                                                \
  } while (0)

  const int nk = g.K / BKE;
  conv_at(0);
  issue(P0{}, 0, 0); issue_scales(0); issue(P1{}, 0, 0); issue(P2{}, 0, 0); issue(P3{}, 0, 0);
  rt_vmcnt<T::PB1 + T::PA1>();                 // a0 (+ scales), b0 of tile 0 landed (younger: b1, a1)
  RT_BAR();
  if (wm == 1) RT_BAR();                       // stagger waves 4-7 by one barrier

  // MX: the scale piece of the NEXT eight K-tiles goes out with part a0 of every eighth tile, i.e. in iterations kt % 8 == 7. The counted
  // waits keep their immediates: with the extra piece among the younger ones they ask for one more of the older pieces than strictly
  // needed — conservative, never too weak. Measured alternatives (tools/mx_dev, K = 15360): exact immediates behind a scalar branch
  // on kt & 1 are 4 % SLOWER (the branch splits the block between the DMA issue and the barrier); unrolling the loop by two to make
  // the parity a compile-time fact costs 18 spilled VGPRs (both buffers' address sets stay live).
  for (int kt = 0; kt + 1 < nk; ++kt) {
    const char* tb = smem + (kt & 1) * T::BUF_BYTES;
    const int nb = (kt & 1) ^ 1;
    const int koff = (kt + 1) * 128;           // bytes
    conv_at(kt + 1);
    // ---- phase 1: (a0,b0)
    RT_READ_A(0); RT_READ_S(0, kt); RT_READ_B(0);
    issue(P0{}, nb, koff);
    if constexpr (MX) { if ((kt & 7) == 7) issue_scales((kt + 1) >> 3); }
    rt_vmcnt<T::PA1 + T::PA0>();               // b1(kt) landed (younger: a1(kt), a0(kt+1))
    RT_BAR(); RT_MFMA(0, 0); RT_BAR();
    // ---- phase 2: (a0,b1)
    RT_READ_B(1);
    issue(P1{}, nb, koff);
    rt_vmcnt<T::PA0 + T::PB0>();               // a1(kt) landed (younger: a0(kt+1), b0(kt+1))
    RT_BAR(); RT_MFMA(0, 1); RT_BAR();
    // ---- phase 3: (a1,b1)
    RT_READ_A(1); RT_READ_S(1, kt);
    issue(P2{}, nb, koff);
    RT_BAR(); RT_MFMA(1, 1); RT_BAR();
    // ---- phase 4: (a1,b0)  (b0 still in registers)
    issue(P3{}, nb, koff);
    rt_vmcnt<T::PB1 + T::PA1>();               // a0(kt+1) (+ scales), b0(kt+1) landed (younger: b1(kt+1), a1(kt+1))
    RT_BAR(); RT_MFMA(1, 0); RT_BAR();
  }
  {                                            // last K-tile: nothing left to issue, drain
    const int kt = nk - 1;
    const char* tb = smem + (kt & 1) * T::BUF_BYTES;
    RT_READ_A(0); RT_READ_S(0, kt); RT_READ_B(0);
    rt_vmcnt<0>();
    RT_BAR(); RT_MFMA(0, 0); RT_BAR();
    RT_READ_B(1);
    RT_BAR(); RT_MFMA(0, 1); RT_BAR();
    RT_READ_A(1); RT_READ_S(1, kt);
    RT_BAR(); RT_MFMA(1, 1); RT_BAR();
    RT_BAR(); RT_MFMA(1, 0); RT_BAR();
  }
  if (wm == 0) RT_BAR();                       // balance the stagger barrier
#undef RT_READ_A
#undef RT_READ_B
#undef RT_READ_S
#undef RT_MFMA
#undef RT_MFMA8_ROW
#undef RT_CAT8
#undef RT_NIH
#undef RT_NJH

  const int mrow = m0 + wm * T::WMR + l15;
  const int ncol = n0 + wn * T::WNC + 4 * (lane >> 4);
  if (g.out_f32) epilogue_tile<true, FP8, T::NI, T::NJ, CONV>(g, bidx, mrow, ncol, acc);
  else epilogue_tile<false, FP8, T::NI, T::NJ, CONV>(g, bidx, mrow, ncol, acc, wide_store);
}

// XCD-aware placement (speed only, never correctness): workgroups are dealt round-robin over the 8 XCDs, so blocks b and b+8
// share an L2. `xcd_run` gives XCD group x a CONTIGUOUS run of an n-item order (bijective for any n): item index of slot s.
__device__ __forceinline__ int xcd_run_start(int n, int xcd) {
  const int q8 = n >> 3, r8 = n & 7;
  return xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
}
__device__ __forceinline__ int xcd_run_count(int n, int xcd) { return (n >> 3) + (xcd < (n & 7) ? 1 : 0); }

// Walk one problem in panels of 8 tile-columns, row by row: the 32 tiles an XCD runs at a time form a ~4x8 patch that shares
// 4 A row-panels and 8 W row-panels in that XCD's L2 instead of ~9 and ~14.
__device__ __forceinline__ void panel_walk(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
  constexpr int PANEL = 8;
  const int panel = t / (PANEL * tiles_m);
  const int pw = min(PANEL, tiles_n - panel * PANEL);          // width of this (possibly last, narrower) panel
  const int tp = t - panel * PANEL * tiles_m;
  tm = tp / pw;
  tn = panel * PANEL + (tp - tm * pw);
}

typedef const __attribute__((address_space(4))) Launch* LaunchPtr;

// One geometry for the whole launch (G_ = Geo256: every ordinary launch; Geo288: M x N a whole number of 288x192 tiles).
template <bool FP8, class G_, bool MX = false, bool CONV = false>
__global__ __launch_bounds__(THREADS, 2) void gemm_pp_kernel(const Launch L) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LaunchPtr Lp = (LaunchPtr)__builtin_amdgcn_kernarg_segment_ptr();
  (void)L;
  const int nwg = (int)gridDim.x;
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int lin = xcd_run_start(nwg, xcd) + slot;
  int gi = 0;
#pragma unroll
  for (int i = 1; i < RT_GEMM_MAX_GROUPS; ++i)
    if (i < Lp->ngroups && lin >= Lp->grp[i].tile_begin) gi = i;
#if defined(__HIP_DEVICE_COMPILE__)
  const GroupDev G = Lp->grp[gi];
#else
  const GroupDev G = L.grp[0];
#endif
  int t = lin - G.tile_begin;
  const int tiles_per_batch = G.tiles_m * G.tiles_n;
  const int bidx = t / tiles_per_batch;
  t -= bidx * tiles_per_batch;
  int tm, tn;
  panel_walk(t, G.tiles_m, G.tiles_n, tm, tn);
  gemm_tile<FP8, G_, MX, CONV>(G.g, bidx, tm * G_::BM, tn * G_::BN, G.wide_store != 0, smem);
}

// Two geometries in one launch: the 256-wide tiles of every group first, then the narrow tiles (GN_) of the columns the host
// left over (GroupDev::n_split). Every XCD group owns a contiguous run of the wide order AND a contiguous run of the narrow
// order and walks the wide ones first, so all eight switch to the short tiles together and the last round is (nearly) full.
template <class GN_>
__global__ __launch_bounds__(THREADS, 2) void gemm_mix_kernel(const Launch L) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  LaunchPtr Lp = (LaunchPtr)__builtin_amdgcn_kernarg_segment_ptr();
  (void)L;
  const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
  const int nW = Lp->wide_total, nN = Lp->narrow_total;
  const int cW = xcd_run_count(nW, xcd), cN = xcd_run_count(nN, xcd);
  if (slot >= cW + cN) return;
  const bool narrow = slot >= cW;
  const int lin = narrow ? xcd_run_start(nN, xcd) + (slot - cW) : xcd_run_start(nW, xcd) + slot;
  int gi = 0;
#pragma unroll
  for (int i = 1; i < RT_GEMM_MAX_GROUPS; ++i)
    if (i < Lp->ngroups && lin >= (narrow ? Lp->grp[i].narrow_begin : Lp->grp[i].tile_begin)) gi = i;
#if defined(__HIP_DEVICE_COMPILE__)
  const GroupDev G = Lp->grp[gi];
#else
  const GroupDev G = L.grp[0];
#endif
  int t = lin - (narrow ? G.narrow_begin : G.tile_begin);
  const int tn_cnt = narrow ? G.tiles_n_narrow : G.tiles_n;
  const int tiles_per_batch = G.tiles_m * tn_cnt;
  const int bidx = t / tiles_per_batch;
  t -= bidx * tiles_per_batch;
  int tm, tn;
  panel_walk(t, G.tiles_m, tn_cnt, tm, tn);
  if (narrow) gemm_tile<false, GN_>(G.g, bidx, tm * GN_::BM, G.n_split + tn * GN_::BN, G.wide_store != 0, smem);
  else gemm_tile<false, Geo256>(G.g, bidx, tm * Geo256::BM, tn * Geo256::BN, G.wide_store != 0, smem);
}

// ---- host side: which geometry? --------------------------------------------------------------------------------
int g_num_cus = 0;
int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0, v = 0;
    g_num_cus = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) g_num_cus = v;
  }
  return g_num_cus;
}
int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
// bit 0: 288x192 launches, bit 1: narrow tail tiles; -1 = not read yet (RT_GEMM_TILES). Default 0: measured on MI355X the fuller last
// round buys nothing — interleaved A/B at the model's shapes: 4608x3072x15360 on 256 tiles of 288x192 348 us vs 342 us on 216 tiles of
// 256x256; q|k|v -1.6 %, ff1 -0.8 %; in the model 1210 vs 1213 TFLOP/s. The chip is power-limited: CUs left idle in a last round give
// their share of the clock to the busy ones, and the narrower tiles re-read more operand bytes per FLOP.
int g_tile_mode = -1;
int tile_mode_now() {
  if (g_tile_mode < 0) g_tile_mode = env_int("RT_GEMM_TILES", 0) & 3;
  return g_tile_mode;
}

template <class K>
int set_lds(K kern, int bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

static int launch_gemm(const rt_gemm_group* groups, int32_t ngroups, void* stream, bool fp8) {
  if (!groups || ngroups < 1 || ngroups > RT_GEMM_MAX_GROUPS) return RT_E_BADARG;
  Launch L{};
  L.ngroups = ngroups;
  int total = 0;
  bool mx = false, conv = false;
  const int bke = fp8 ? 128 : BK;                     // elements per K-tile
  const int al = fp8 ? 16 : 8;                        // elements per 16 bytes
  const int64_t esz = fp8 ? 1 : 2;
  static const bool wide_on = env_int("RT_GEMM_WIDE_STORE", 1) != 0;   // A/B switches (tests and tools)
  const int tile_mode = tile_mode_now();
  for (int i = 0; i < ngroups; ++i) {
    const rt_gemm_group& g = groups[i];
    if (!g.A || !g.W || !g.C || g.M < 1 || g.N < 1 || g.K < 1 || g.batch < 1) return RT_E_BADARG;
    if (g.K % bke != 0 || g.N % 4 != 0) return RT_E_SHAPE;
    if (g.rows_per_batch > 0 && g.M % g.rows_per_batch != 0) return RT_E_SHAPE;
    if (!RT_ALIGNED(g.A, 16) || !RT_ALIGNED(g.W, 16) || g.lda % al || g.ldw % al || g.strideA % al) return RT_E_ALIGN;
    if ((g.conv_ks == 0 && g.lda < g.K) || g.ldw < g.K || g.ldc < g.N) return RT_E_SHAPE;
    // staging offsets are 32-bit byte offsets from the (per-batch) operand base
    if (((int64_t)g.M * g.lda) * esz >= ((int64_t)1 << 32) || ((int64_t)g.N * g.ldw) * esz >= ((int64_t)1 << 32)) return RT_E_SHAPE;
    const int cal = g.out_f32 ? 16 : 8;
    if (!RT_ALIGNED(g.C, cal) || g.ldc % 4 || g.strideC % 4) return RT_E_ALIGN;
    if (g.res && (!RT_ALIGNED(g.res, cal) || g.ldr % 4 || g.strideR % 4)) return RT_E_ALIGN;
    if (g.add2 && (!RT_ALIGNED(g.add2, 8) || g.ld2 % 4 || g.stride2 % 4)) return RT_E_ALIGN;
    if (g.bias && !RT_ALIGNED(g.bias, 8)) return RT_E_ALIGN;
    if (g.gate && (!RT_ALIGNED(g.gate, 16) || g.gate_ld % 4)) return RT_E_ALIGN;
    if (fp8 && g.w_scale && !RT_ALIGNED(g.w_scale, 16)) return RT_E_ALIGN;
    if (fp8) {                                          // MX block scales: all groups of a launch with A scales, or none
      if ((g.a_bscale != nullptr) != (groups[0].a_bscale != nullptr)) return RT_E_BADARG;
      if (g.a_bscale) {
        mx = true;
        if (g.a_bscale_plane % 2048 != 0 || g.a_bscale_rows % 64 != 0) return RT_E_SHAPE;
        if (g.a_bscale_plane < ((((int64_t)(g.batch - 1) * g.a_bscale_rows + g.M) + 63) / 64) * 2048) return RT_E_SHAPE;
        if ((int64_t)((g.K + 1023) / 1024) * g.a_bscale_plane >= ((int64_t)1 << 31)) return RT_E_SHAPE;
        if (!RT_ALIGNED(g.a_bscale, 16)) return RT_E_ALIGN;
      }
      if (g.c8) {
        if (g.out_f32 || !g.c_bscale) return RT_E_BADARG;
        if (g.c8_from < 0 || g.c8_from % 256 != 0 || g.c8_from >= g.N || (g.N - g.c8_from) % 32 != 0 || g.c_bscale_plane % 2048 != 0 || g.c_bscale_rows % 64 != 0 ||
            g.c_bscale_k0 < 0 || g.c_bscale_k0 % 32 != 0) return RT_E_SHAPE;
        if (!RT_ALIGNED(g.c8, 8) || g.ldc8 % 8 || g.stride_c8 % 8 || !RT_ALIGNED(g.c_bscale, 16)) return RT_E_ALIGN;
      }
    }
    if (g.conv_ks != 0) {                               // convolution form (see gemm_tile): one bf16 problem, rows = haloed pixels
      if (fp8 || ngroups != 1 || g.batch != 1 || (g.conv_ks != 1 && g.conv_ks != 3)) return RT_E_BADARG;
      if (g.conv_cin < 64 || g.conv_cin % 64 != 0 || g.lda != g.conv_cin || g.K != g.conv_ks * g.conv_ks * g.conv_cin) return RT_E_SHAPE;
      if (g.conv_w2 < 3 || g.conv_h2 < 3 || g.M % (g.conv_w2 * g.conv_h2) != 0 || g.M >= (1 << 24)) return RT_E_SHAPE;
      conv = true;
    }
    L.grp[i].g = g;
    L.grp[i].tiles_m = (g.M + Geo256::BM - 1) / Geo256::BM;
    L.grp[i].tiles_n = (g.N + Geo256::BN - 1) / Geo256::BN;
    L.grp[i].wide_store = (wide_on && !g.out_f32 && RT_ALIGNED(g.C, 16) && g.ldc % 8 == 0 && g.strideC % 8 == 0 && g.N % 8 == 0) ? 1 : 0;
    L.grp[i].tile_begin = total;
    total += L.grp[i].tiles_m * L.grp[i].tiles_n * g.batch;
  }
  static bool attr_done = false;
  if (!attr_done) {
    int e = set_lds(gemm_pp_kernel<false, Geo256>, Geo256::LDS_BYTES);
    if (!e) e = set_lds(gemm_pp_kernel<true, Geo256>, Geo256::LDS_BYTES);
    if (!e) e = set_lds(gemm_pp_kernel<true, Geo256, true>, Geo256::LDS_BYTES + 16384);
    if (!e) e = set_lds(gemm_pp_kernel<false, Geo288>, Geo288::LDS_BYTES);
    if (!e) e = set_lds(gemm_pp_kernel<false, Geo256, false, true>, Geo256::LDS_BYTES);
    if (!e) e = set_lds(gemm_pp_kernel<false, Geo128, false, true>, Geo128::LDS_BYTES);
    if (!e) e = set_lds(gemm_mix_kernel<Geo192>, Geo256::LDS_BYTES);
    if (!e) e = set_lds(gemm_mix_kernel<Geo128>, Geo256::LDS_BYTES);
    if (e) return e;
    attr_done = true;
  }
  hipStream_t st = (hipStream_t)stream;
  if (fp8) {
    if (mx) hipLaunchKernelGGL((gemm_pp_kernel<true, Geo256, true>), dim3(total), dim3(THREADS), Geo256::LDS_BYTES + 16384, st, L);
    else hipLaunchKernelGGL((gemm_pp_kernel<true, Geo256>), dim3(total), dim3(THREADS), Geo256::LDS_BYTES, st, L);
    return rt_hip_status();
  }
  if (conv) {                                           // N <= 128 (the decoder's 1024x1024 stages): the 256x128 tile, no dead columns
    L.grp[0].g.conv_inv_w2 = 1.0f / (float)groups[0].conv_w2;
    L.grp[0].g.conv_inv_h2 = 1.0f / (float)groups[0].conv_h2;
    if (groups[0].N <= 128) {
      L.grp[0].tiles_n = (groups[0].N + Geo128::BN - 1) / Geo128::BN;
      hipLaunchKernelGGL((gemm_pp_kernel<false, Geo128, false, true>), dim3(L.grp[0].tiles_m * L.grp[0].tiles_n), dim3(THREADS), Geo128::LDS_BYTES, st, L);
    } else {
      hipLaunchKernelGGL((gemm_pp_kernel<false, Geo256, false, true>), dim3(total), dim3(THREADS), Geo256::LDS_BYTES, st, L);
    }
    return rt_hip_status();
  }
  const int cus = num_cus();
  // (1) 288x192 tiles: one problem whose M x N is a whole number of them and whose 256x256 tiling leaves the last round of
  //     workgroups emptier. Cost model: rounds x tile area, the 288x192 tile charged 4 % for its lower operand reuse.
  if ((tile_mode & 1) && ngroups == 1 && groups[0].M % Geo288::BM == 0 && groups[0].N % Geo288::BN == 0) {
    const rt_gemm_group& g = groups[0];
    const int t288 = (g.M / Geo288::BM) * (g.N / Geo288::BN) * g.batch;
    const double c256 = (double)((total + cus - 1) / cus) * Geo256::BM * Geo256::BN;
    const double c288 = (double)((t288 + cus - 1) / cus) * Geo288::BM * Geo288::BN * 1.04;
    if (c288 < c256) {
      L.grp[0].tiles_m = g.M / Geo288::BM;
      L.grp[0].tiles_n = g.N / Geo288::BN;
      hipLaunchKernelGGL((gemm_pp_kernel<false, Geo288>), dim3(t288), dim3(THREADS), Geo288::LDS_BYTES, st, L);
      return rt_hip_status();
    }
  }
  // (2) narrow tail: a launch of several rounds whose last round is poorly filled gives up the columns of that round to 3/4- or
  //     1/2-width tiles (same row panels). Every group is cut at the same fraction of its columns (they share N in practice).
  if ((tile_mode & 2) && total > cus) {
    const int rounds = (total + cus - 1) / cus;
    const double base = (double)rounds;                                    // time in units of one 256x256 round
    int best_w = 0, best_cols = 0;
    double best = base * 0.985;                                            // must beat the plain tiling by 1.5 %
    for (int w : {192, 128}) {
      const double unit = w / 256.0 * (w == 192 ? 1.03 : 1.08);            // a narrow tile's time in 256x256 tiles (less operand reuse)
      // give up c column-tiles (of 256) of every group to the narrow geometry
      int min_tn = 1 << 30;
      for (int i = 0; i < ngroups; ++i) min_tn = L.grp[i].tiles_n < min_tn ? L.grp[i].tiles_n : min_tn;
      for (int c = 1; c < min_tn; ++c) {
        bool ok = true;
        int wide = 0, narrow = 0;
        for (int i = 0; i < ngroups; ++i) {
          const rt_gemm_group& g = groups[i];
          const int n_split = (L.grp[i].tiles_n - c) * 256;
          if (g.N % 256 != 0 || (g.N - n_split) % w != 0) { ok = false; break; }
          wide += L.grp[i].tiles_m * (L.grp[i].tiles_n - c) * g.batch;
          narrow += L.grp[i].tiles_m * ((g.N - n_split) / w) * g.batch;
        }
        if (!ok) continue;
        // wide tiles run first; narrow tiles fill the CUs as they free up: list scheduling bound
        const double tw = (double)wide / cus, tn_ = (double)narrow / cus * unit;
        const double t = ((wide + cus - 1) / cus) + (double)((narrow + cus - 1) / cus) * unit;
        const double lb = tw + tn_;
        const double est = lb + (t - lb) * 0.5;                             // between the perfect and the round-by-round bound
        if (est < best) { best = est; best_w = w; best_cols = c; }
      }
    }
    if (best_w) {
      int wide = 0, narrow = 0;
      for (int i = 0; i < ngroups; ++i) {
        const rt_gemm_group& g = groups[i];
        GroupDev& G = L.grp[i];
        G.tiles_n -= best_cols;
        G.n_split = G.tiles_n * 256;
        G.tiles_n_narrow = (g.N - G.n_split) / best_w;
        G.tile_begin = wide;
        G.narrow_begin = narrow;
        wide += G.tiles_m * G.tiles_n * g.batch;
        narrow += G.tiles_m * G.tiles_n_narrow * g.batch;
      }
      L.wide_total = wide;
      L.narrow_total = narrow;
      int per_xcd = 0;
      for (int x = 0; x < 8; ++x) {
        const int c = (wide >> 3) + (x < (wide & 7)) + (narrow >> 3) + (x < (narrow & 7));
        per_xcd = c > per_xcd ? c : per_xcd;
      }
      if (best_w == 192) hipLaunchKernelGGL((gemm_mix_kernel<Geo192>), dim3(8 * per_xcd), dim3(THREADS), Geo256::LDS_BYTES, st, L);
      else hipLaunchKernelGGL((gemm_mix_kernel<Geo128>), dim3(8 * per_xcd), dim3(THREADS), Geo256::LDS_BYTES, st, L);
      return rt_hip_status();
    }
  }
  hipLaunchKernelGGL((gemm_pp_kernel<false, Geo256>), dim3(total), dim3(THREADS), Geo256::LDS_BYTES, st, L);
  return rt_hip_status();
}

extern "C" int rt_gemm_tile_mode(int32_t mode) {
  const int prev = tile_mode_now();
  if (mode >= 0) g_tile_mode = mode & 3;
  return prev;
}

extern "C" int rt_gemm_bf16(const rt_gemm_group* groups, int32_t ngroups, void* stream) {
  return launch_gemm(groups, ngroups, stream, false);
}

extern "C" int rt_gemm_fp8(const rt_gemm_group* groups, int32_t ngroups, void* stream) {
  return launch_gemm(groups, ngroups, stream, true);
}
