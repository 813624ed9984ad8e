// rt_vae_attention — the AutoencoderKL mid-block attention (reached from PIPE:1139 decode and
// PIPE:467,705,711 encode): ONE head of C = 128·NW channels (512 in the FLUX VAE) over H·W positions, softmax(q kᵀ / sqrt(C)) v,
// flash-style — the (H·W)² score matrix (1 GiB fp32 at 1024², 5.4 GB at 1536²) never exists.
//
// A 512-wide head does not fit one wave's registers the way the 128-wide MMDiT heads do (Oᵀ alone would be 256 accumulators per
// lane), so the CHANNELS are split over the waves of a workgroup: a workgroup = NW waves = 32 query rows; wave w owns channels
// 128w .. 128w+127 of q, k, v and of the output. Per 32-key tile:
//   1. every wave forms its PARTIAL scores Sᵀ_w[key][q] = K[:, slice w] · Qᵀ[slice w, :]  (8 MFMAs 32x32x16, the key on the rows,
//      the query on the lane — the orientation of csrc/attention.hip) and writes them to an LDS exchange buffer;
//   2. after a barrier every wave sums the NW partials in the fixed order w = 0..NW-1 — all waves now hold the SAME bits of S,
//      so the softmax statistics they each keep (running max, row sum; lane-local apart from one lane<->lane+32 exchange) never
//      diverge and need no further communication;
//   3. Oᵀ[slice w] += Vᵀ[slice w] · Pᵀ with the exponentiated score registers as the B operand (no LDS round trip) and Vᵀ by
//      ds_read_b64_tr_b16 (8 MFMAs).
// K and V tiles arrive by LDS-DMA (each wave stages its own 256-byte column slice of the 32 rows) into a 2-deep ring; two
// barriers per tile. The LDS image of a slice is attention.hip's: 256-byte rows, 16-byte chunk c of row r at c ^ f(r).
// Bitwise reproducible (no atomics, fixed summation order); batch entries are independent grid rows.
#include "rt_common.h"

namespace {

constexpr int BKV = 32;                 // keys per tile
constexpr int BQ = 32;                  // query rows per workgroup
constexpr int SLICE_B = BKV * 256;      // one wave's K (or V) slice of a tile: 32 rows x 128 channels x 2 B = 8 KiB
constexpr int XCH_WAVE_B = 64 * 64;     // one wave's partial scores: 16 fp32 per lane = 4 KiB
constexpr float RESCALE_THR = 6.0f;     // log2 units, as in attention.hip

__device__ __forceinline__ int swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

typedef const __attribute__((address_space(3))) char* lds_cptr;
__device__ __forceinline__ s16x4 tr_read(lds_cptr p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void vae_attention_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int HW, int64_t ld,
                                                                int64_t stride_b, int64_t ldo, int64_t stride_ob, float scale_log2) {
  constexpr int STAGE_B = 2 * NW * SLICE_B;                      // K slices | V slices
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages] | exchange [NW][4 KiB]
  char* xch = smem + 2 * STAGE_B;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int q0 = (int)blockIdx.x * BQ, b = (int)blockIdx.y;
  const int ntiles = HW / BKV;
  const bf16_t* Qb = Q + b * stride_b + wave * 128;
  const bf16_t* Kb = K + b * stride_b + wave * 128;
  const bf16_t* Vb = V + b * stride_b + wave * 128;

  // staging: this wave's slice, 8 pieces of 4 rows; lane -> (row in piece, physical chunk)
  const int srow = lane >> 4, spc = lane & 15;
  uint32_t soff[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row_t = p * 4 + srow;
    soff[p] = ((uint32_t)row_t * (uint32_t)ld + (uint32_t)((spc ^ swz(row_t)) << 3)) * 2u;
  }
  const __amdgpu_buffer_rsrc_t rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Kb), 0, -1, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcV = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Vb), 0, -1, 0x00020000);
  const int tile_stride_b = BKV * (int)ld * 2;
  auto stage = [&](int sl, int tix) {
    char* kb = smem + sl * STAGE_B + wave * SLICE_B;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcK, LDS_PTR(kb + p * 1024), 16, (int)soff[p], tix * tile_stride_b, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcV, LDS_PTR(kb + NW * SLICE_B + p * 1024), 16, (int)soff[p], tix * tile_stride_b, 0, 0);
    }
  };
  stage(0, 0);

  // read addresses inside a slice (see attention.hip): K rows for the A operand of Sᵀ, Vᵀ by transposed reads
  int kp[8];
  {
    const int ksw = swz(l31);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) kp[ks] = l31 * 256 + (((2 * ks + hh) ^ ksw) << 4);
  }
  const int tq = (lane >> 2) & 3, tp = lane & 3, tg1 = (lane >> 4) & 1;
  int vp[2][4];
  {
    const int cl = tg1 * 2 + (tp >> 1);
    const int rl0 = 4 * hh + tq, rl1 = rl0 + 8;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      vp[0][dt] = rl0 * 256 + (((dt * 4 + cl) ^ swz(rl0)) << 4) + 8 * (tp & 1);
      vp[1][dt] = rl1 * 256 + (((dt * 4 + cl) ^ swz(rl1)) << 4) + 8 * (tp & 1);
    }
  }

  // Q fragments of this wave's channel slice (B operand): lane holds Q[q][128w + 16ks + 8hh .. +7]
  bf16x8 qf[8];
  {
    const bf16_t* qp = Qb + (int64_t)(q0 + l31) * ld + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + ks * 16);
  }
  f32x16 o_acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int t = 0; t < ntiles; ++t) {
    const int sl = t & 1;
    lds_cptr kb = (lds_cptr)smem + sl * STAGE_B + wave * SLICE_B;
    lds_cptr vb = kb + NW * SLICE_B;
    rt_dma_barrier();                                  // tile t landed; every wave is done with tile t-1 (its stage and the exchange)
    if (t + 1 < ntiles) stage(sl ^ 1, t + 1);
    // 1. partial scores over this wave's 128 channels
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 kf = *(const __attribute__((address_space(3))) bf16x8*)(kb + kp[ks]);
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s, 0, 0, 0);
    }
    {
      f32x4* xw = reinterpret_cast<f32x4*>(xch + wave * XCH_WAVE_B);
#pragma unroll
      for (int g = 0; g < 4; ++g) xw[g * 64 + lane] = f32x4{s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]};
    }
    rt_lds_barrier();                                  // partials visible; the LDS-DMA of tile t+1 stays in flight
    // 2. full scores: the NW partials in fixed order (identical bits on every wave)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a = reinterpret_cast<const f32x4*>(xch)[g * 64 + lane];
#pragma unroll
      for (int w = 1; w < NW; ++w) a += reinterpret_cast<const f32x4*>(xch + w * XCH_WAVE_B)[g * 64 + lane];
      s[4 * g] = a[0]; s[4 * g + 1] = a[1]; s[4 * g + 2] = a[2]; s[4 * g + 3] = a[3];
    }
    // softmax update (per lane: its query row, 16 of the tile's 32 keys; the other 16 live in lane ^ 32)
    float ma = max3f(s[0], s[1], s[2]), mb = max3f(s[3], s[4], s[5]);
    ma = max3f(ma, s[6], s[7]);
    mb = max3f(mb, s[8], s[9]);
    ma = max3f(ma, s[10], s[11]);
    mb = max3f(mb, s[12], s[13]);
    const float mx = fmaxf(max3f(ma, s[14], s[15]), mb);
    unsigned xa = __builtin_bit_cast(unsigned, mx), xb = xa;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(xa), "+v"(xb));     // see attention.hip: the builtin's pair is mis-folded
    const float mxr = fmaxf(__builtin_bit_cast(float, xa), __builtin_bit_cast(float, xb)) * scale_log2;
    if (__any(mxr - m_run > RESCALE_THR)) {
      const float m_new = fmaxf(m_run, mxr);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);                       // first tile: exp2(-inf) = 0 on zeroed state
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[i][r] *= alpha;
    }
    bf16x8 pf[2];
    float ps = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[8 * s2 + j], scale_log2, -m_run));
        ps += p;
        pf[s2][j] = (__bf16)p;
      }
    l_run += ps;
    // 3. Oᵀ[slice] += Vᵀ[slice] · Pᵀ
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const s16x4 lo = tr_read(vb + vp[0][dt] + s2 * 4096);
        const s16x4 hi = tr_read(vb + vp[1][dt] + s2 * 4096);
        const bf16x8 vf = __builtin_shufflevector(__builtin_bit_cast(bf16x4, lo), __builtin_bit_cast(bf16x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
        o_acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[s2], o_acc[dt], 0, 0, 0);
      }
  }
  // epilogue: O[q][128w + d] = Oᵀ / l
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  rt_store_o_rows(O + b * stride_ob + (int64_t)(q0 + l31) * ldo + wave * 128, true, hh, o_acc, inv);
}

template <int NW>
int launch(const void* q, const void* k, const void* v, void* o, int64_t ld, int64_t stride_b, int64_t ldo, int64_t stride_ob, int B, int HW,
           float scale, hipStream_t st) {
  constexpr int LDS = 2 * 2 * NW * SLICE_B + NW * XCH_WAVE_B;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vae_attention_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_done = true;
  }
  hipLaunchKernelGGL(vae_attention_kernel<NW>, dim3(HW / BQ, B), dim3(NW * 64), LDS, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                     (bf16_t*)o, HW, ld, stride_b, ldo, stride_ob, scale * 1.4426950408889634f);
  return rt_hip_status();
}

}  // namespace

extern "C" int rt_vae_attention(const void* q, const void* k, const void* v, void* o, int64_t ld, int64_t stride_b, int64_t ldo,
                                int64_t stride_ob, int32_t B, int32_t HW, int32_t C, float scale, void* stream) {
  if (!q || !k || !v || !o || B < 1 || HW < 1 || C < 1) return RT_E_BADARG;
  if (C != 128 && C != 256 && C != 512) return RT_E_SHAPE;               // NW = C / 128 waves per workgroup
  if (HW % 32 != 0) return RT_E_SHAPE;
  if (!RT_ALIGNED(q, 16) || !RT_ALIGNED(k, 16) || !RT_ALIGNED(v, 16) || !RT_ALIGNED(o, 16) || ld % 8 || stride_b % 8 || ldo % 8 || stride_ob % 8)
    return RT_E_ALIGN;
  if (ld < C || ldo < C || (int64_t)(HW + BKV) * ld * 2 >= ((int64_t)1 << 31)) return RT_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (C == 512) return launch<4>(q, k, v, o, ld, stride_b, ldo, stride_ob, B, HW, scale, st);
  if (C == 256) return launch<2>(q, k, v, o, ld, stride_b, ldo, stride_ob, B, HW, scale, st);
  return launch<1>(q, k, v, o, ld, stride_b, ldo, stride_ob, B, HW, scale, st);
}
