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

namespace {

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int THREADS = 512;
constexpr int TILE_BYTES = BM * BK * 2;           // 32 KiB per operand tile
constexpr int BUF_BYTES = 2 * TILE_BYTES;         // A + W
constexpr int LDS_BYTES = 2 * BUF_BYTES;          // double buffered: 128 KiB

struct GroupDev {
  rt_gemm_group g;
  int tiles_m, tiles_n, tile_begin, pad;
};
struct Launch {
  GroupDev grp[RT_GEMM_MAX_GROUPS];
  int ngroups;
};

template <bool OUT_F32, typename GT>
__device__ __forceinline__ void epilogue_store(const GT& g, int bidx, int m, int n, f32x4 v) {
  // v holds columns n..n+3 of row m. All optional terms are wave-uniform branches.
  if (g.bias) {
    const u32x2 b = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(g.bias) + n);
    v[0] += bf16lo(b[0]); v[1] += bf16hi(b[0]); v[2] += bf16lo(b[1]); v[3] += bf16hi(b[1]);
  }
  if (n >= g.gelu_from) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = gelu_tanh_f(v[i]);
  }
  const int rpb = g.rows_per_batch > 0 ? g.rows_per_batch : g.M;
  if (g.gate) {
    const int gb = bidx * (g.M / rpb) + m / rpb;
    const f32x4 gt = *reinterpret_cast<const f32x4*>(g.gate + (int64_t)gb * g.gate_ld + n);
    v *= gt;
  }
  v *= g.alpha;
  if (g.rowscale) v *= g.rowscale[m % rpb];
  const int64_t coff = (int64_t)bidx * g.strideC + (int64_t)m * g.ldc + n;
  if (g.res) {
    const int64_t roff = (int64_t)bidx * g.strideR + (int64_t)m * g.ldr + n;
    if (OUT_F32) {
      v += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.res) + roff);
    } else {
      const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(g.res) + roff);
      v[0] += bf16lo(r[0]); v[1] += bf16hi(r[0]); v[2] += bf16lo(r[1]); v[3] += bf16hi(r[1]);
    }
  }
  if (g.add2) {
    const int64_t aoff = (int64_t)bidx * g.stride2 + (int64_t)m * g.ld2 + n;
    const u32x2 r = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(g.add2) + aoff);
    v[0] += bf16lo(r[0]); v[1] += bf16hi(r[0]); v[2] += bf16lo(r[1]); v[3] += bf16hi(r[1]);
  }
  if (OUT_F32) {
    *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + coff) = v;
  } else {
    u32x2 o;
    o[0] = pack_bf16x2(v[0], v[1]);
    o[1] = pack_bf16x2(v[2], v[3]);
    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(g.C) + coff) = o;
  }
}

__global__ __launch_bounds__(THREADS, 2) void gemm_bf16_kernel(const Launch L) {
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- which problem / tile is this workgroup? (scalar)
  // The by-value Launch is indexed dynamically; read it through the kernarg segment pointer (scalar loads)
  // so the compiler does not copy the struct to scratch.
  typedef const __attribute__((address_space(4))) Launch* LaunchPtr;
  LaunchPtr Lp = (LaunchPtr)__builtin_amdgcn_kernarg_segment_ptr();
  (void)L;
  int gi = 0;
#pragma unroll
  for (int i = 1; i < RT_GEMM_MAX_GROUPS; ++i)
    if (i < Lp->ngroups && (int)blockIdx.x >= Lp->grp[i].tile_begin) gi = i;
#if defined(__HIP_DEVICE_COMPILE__)
  const GroupDev G = Lp->grp[gi];      // scalar loads into SGPRs
#else
  const GroupDev G = L.grp[0];         // host pass only parses this body
#endif
  const rt_gemm_group& g = G.g;
  int t = (int)blockIdx.x - G.tile_begin;
  const int tiles_per_batch = G.tiles_m * G.tiles_n;
  const int bidx = t / tiles_per_batch;
  t -= bidx * tiles_per_batch;
  const int tn = t / G.tiles_m;
  const int tm = t - tn * G.tiles_m;
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- staging addresses: wave w stages rows [32w, 32w+32) of each operand, 4 pieces of 8 rows.
  const bf16_t* Ab = reinterpret_cast<const bf16_t*>(g.A) + (int64_t)bidx * g.strideA;
  const bf16_t* Wb = reinterpret_cast<const bf16_t*>(g.W);
  const bf16_t* srcA[4];
  const bf16_t* srcW[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = wave * 32 + p * 8 + (lane >> 3);
    const int lc = (lane & 7) ^ ((row >> 1) & 7);      // logical chunk this lane must fetch
    const int am = min(m0 + row, g.M - 1);             // clamp: tail rows read valid memory, never stored
    const int wr = min(n0 + row, g.N - 1);
    srcA[p] = Ab + (int64_t)am * g.lda + lc * 8;
    srcW[p] = Wb + (int64_t)wr * g.ldw + lc * 8;
  }
  const int stage_off = wave * 32 * 128;               // byte offset of this wave's rows inside a tile

  auto stage = [&](int buf, int kt) {
    char* base = smem + buf * BUF_BYTES + stage_off;
    const int koff = kt * BK;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      __builtin_amdgcn_global_load_lds(GLB_PTR(srcA[p] + koff), LDS_PTR(base + p * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(srcW[p] + koff), LDS_PTR(base + TILE_BYTES + p * 1024), 16, 0, 0);
    }
  };

  // ---- fragment read offsets (bytes inside a tile); identical swizzle for both operands.
  const int l15 = lane & 15;
  const int sw = (lane >> 1) & 7;                       // ((row>>1)&7) with row ≡ l15 (mod 16)
  const int rd0 = l15 * 128 + (((0 + (lane >> 4)) ^ sw) << 4);
  const int rd1 = l15 * 128 + (((4 + (lane >> 4)) ^ sw) << 4);
  const int a_base = wm * 128 * 128;                    // activation rows of this wave (bytes)
  const int w_base = TILE_BYTES + wn * 64 * 128;        // weight rows of this wave

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = g.K / BK;
  stage(0, 0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const char* tb = smem + cur * BUF_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int rd = kk ? rd1 : rd0;
      bf16x8 wf[4], af[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(tb + w_base + j * 2048 + rd);
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(tb + a_base + i * 2048 + rd);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();   // drains the LDS-DMA (vmcnt(0)) and orders buffer reuse
  }

  // ---- epilogue: lane owns row m = ... + l15 and columns n = ... + 4*(lane>>4) .. +3 of each fragment
  const int mrow = m0 + wm * 128 + l15;
  const int ncol = n0 + wn * 64 + 4 * (lane >> 4);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = mrow + i * 16;
    if (m < g.M) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = ncol + j * 16;
        if (n < g.N) {
          if (g.out_f32) epilogue_store<true>(g, bidx, m, n, acc[i][j]);
          else epilogue_store<false>(g, bidx, m, n, acc[i][j]);
        }
      }
    }
  }
}

}  // namespace

extern "C" int rt_gemm_bf16(const rt_gemm_group* groups, int32_t ngroups, void* stream) {
  if (!groups || ngroups < 1 || ngroups > RT_GEMM_MAX_GROUPS) return RT_E_BADARG;
  Launch L{};
  L.ngroups = ngroups;
  int total = 0;
  for (int i = 0; i < ngroups; ++i) {
    const rt_gemm_group& g = groups[i];
    if (!g.A || !g.W || !g.C || g.M < 1 || g.N < 1 || g.K < 1 || g.batch < 1) return RT_E_BADARG;
    if (g.K % BK != 0 || g.N % 4 != 0) return RT_E_SHAPE;
    if (g.rows_per_batch > 0 && g.M % g.rows_per_batch != 0) return RT_E_SHAPE;
    if (!RT_ALIGNED(g.A, 16) || !RT_ALIGNED(g.W, 16) || g.lda % 8 || g.ldw % 8 || g.strideA % 8) return RT_E_ALIGN;
    if (g.lda < g.K || g.ldw < g.K || g.ldc < g.N) return RT_E_SHAPE;
    const int cal = g.out_f32 ? 16 : 8;
    if (!RT_ALIGNED(g.C, cal) || g.ldc % 4 || g.strideC % 4) return RT_E_ALIGN;
    if (g.res && (!RT_ALIGNED(g.res, cal) || g.ldr % 4 || g.strideR % 4)) return RT_E_ALIGN;
    if (g.add2 && (!RT_ALIGNED(g.add2, 8) || g.ld2 % 4 || g.stride2 % 4)) return RT_E_ALIGN;
    if (g.bias && !RT_ALIGNED(g.bias, 8)) return RT_E_ALIGN;
    if (g.gate && (!RT_ALIGNED(g.gate, 16) || g.gate_ld % 4)) return RT_E_ALIGN;
    L.grp[i].g = g;
    L.grp[i].tiles_m = (g.M + BM - 1) / BM;
    L.grp[i].tiles_n = (g.N + BN - 1) / BN;
    L.grp[i].tile_begin = total;
    total += L.grp[i].tiles_m * L.grp[i].tiles_n * g.batch;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_bf16_kernel, dim3(total), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, L);
  return rt_hip_status();
}
