// Shared device helpers for the gfx950 kernels. CDNA4 only: wave = 64 lanes, MFMA, LDS-DMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "reptext_hip.h"

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define RT_WAVE 64
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// round-to-nearest-even; plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 at -O3).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return *reinterpret_cast<bf16_t*>(&h);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ float bf16lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// x·sigmoid(x) on the transcendental unit (v_exp_f32, v_rcp_f32): 4 instructions instead of ~25; |rel err| < 3e-7 (1 ulp of rcp)
__device__ __forceinline__ float silu_fast_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// GELU tanh approximation: 0.5 x (1 + tanh(sqrt(2/pi)(x + 0.044715 x^3)))  == x * sigmoid(2u)
__device__ __forceinline__ float gelu_tanh_f(float x) {
  // x·sigmoid(2u), u = sqrt(2/pi)(x + 0.044715x³); exp and reciprocal on the transcendental unit (v_exp_f32, v_rcp_f32):
  // ~8 VALU ops per element instead of ~25 for __expf + IEEE divide. |rel err| < 4e-7.
  const float x2 = x * x;
  const float t = x * (-2.302208198f - 0.1029432397f * x2);     // -2u·log2(e)
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

static inline int rt_hip_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? RT_OK : (int)e;
}
#define RT_ALIGNED(p, a) ((((uintptr_t)(p)) & ((a)-1)) == 0)

// Row-per-lane epilogue of the attention kernels: lane (q = lane&31, hh = lane>>5) holds, per (dt, g), the four consecutive
// output columns 32dt + 8g + 4hh .. +3 of its query row. v_permlane32_swap on the registers of groups (g, g+1) hands the lower
// half-wave [own g | partner's g] and the upper half-wave [partner's g+1 | own g+1]: 16 contiguous bytes per lane, so the tile
// is written with 8 dwordx4 stores per lane instead of 16 dwordx2 (same bytes, same addresses).
__device__ __forceinline__ void rt_store_o_rows(bf16_t* orow /* row base + head*128, no hh offset */, bool valid, int hh,
                                                const f32x16 (&o_acc)[4], float inv) {
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      u32x2 a, b;
      a[0] = pack_bf16x2(o_acc[dt][4 * g + 0] * inv, o_acc[dt][4 * g + 1] * inv);
      a[1] = pack_bf16x2(o_acc[dt][4 * g + 2] * inv, o_acc[dt][4 * g + 3] * inv);
      b[0] = pack_bf16x2(o_acc[dt][4 * g + 4] * inv, o_acc[dt][4 * g + 5] * inv);
      b[1] = pack_bf16x2(o_acc[dt][4 * g + 6] * inv, o_acc[dt][4 * g + 7] * inv);
      const auto x = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
      const auto y = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
      if (valid) *reinterpret_cast<u32x4*>(orow + dt * 32 + 8 * (g + hh)) = u32x4{x[0], y[0], x[1], y[1]};
    }
}

// Barrier that PUBLISHES LDS-DMA data: every wave first drains its own global->LDS copies (s_waitcnt vmcnt(0)), then the
// workgroup meets. __syncthreads() alone is not enough: its workgroup-scope fence does not wait on vector-memory operations,
// and the compiler's own vmcnt wait is only placed before THIS wave's next LDS read (it was found missing on a loop
// back-edge of the e4m3 attention kernel — other waves' rows were then read before they had landed, visibly so on a cold
// first launch). Use this at every barrier after which DMA-staged rows of OTHER waves are read.
__device__ __forceinline__ void rt_dma_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// Barrier that orders LDS traffic only (ds_write -> barrier -> ds_read by other waves) and leaves LDS-DMA / global loads in
// flight: __syncthreads()'s fence also waits for every pending vector-memory operation — it drains an LDS-DMA issued for a LATER
// tile — while these address-space-scoped fences lower to s_waitcnt lgkmcnt(0) only (checked in the .s).
__device__ __forceinline__ void rt_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// LDS-DMA issued from inline asm: one 1-KiB piece, global (buffer descriptor + per-lane byte offset + scalar byte offset) -> LDS at
// the wave-uniform byte address `lds_addr` + lane*16. Why asm: hipcc tracks a builtin LDS-DMA as a pending LDS WRITE and puts
// `s_waitcnt vmcnt(0)` in front of the wave's next LDS read it cannot prove disjoint (seen in front of the first
// ds_read_b64_tr_b16 of every attention tile, a third of the way into the tile: the copy for the NEXT tile was being waited for
// there). An asm statement is invisible to that bookkeeping, so the copy stays in flight until OUR wait: every barrier that
// publishes DMA-staged rows must be rt_dma_barrier() (s_waitcnt vmcnt(0) + barrier) or a counted wait. M0 (the LDS base of the
// copy) is written in the same statement that uses it; nothing else in these kernels depends on M0.
typedef __attribute__((ext_vector_type(4))) uint32_t rt_srd_t;
__device__ __forceinline__ rt_srd_t rt_make_srd(const void* base) {
  const uint64_t a = reinterpret_cast<uint64_t>(base);
  return rt_srd_t{(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)a), (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32)) & 0xffffu, 0xffffffffu,
                  0x00020000u};
}
__device__ __forceinline__ void rt_dma16_asm(rt_srd_t srd, uint32_t lds_addr, uint32_t voff, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(srd), "s"(soff) : "memory");
}

// ---- MX block scales (config 5): one E8M0 byte per 32 consecutive K-elements of an activation row -------------------------------
// Scale of a block with maximum magnitude amax: the smallest power of two 2^e with amax <= 448 * 2^e (448 = 1.75 * 2^8 is e4m3's
// largest finite value), so no element saturates. Returned as the biased byte s = e + 127, clamped to [1, 253] (an all-zero block
// gets s = 1; its elements are zero whatever the scale). From the float's own fields: e = E - 127 - 8, one more when the mantissa
// exceeds 1.75's. oracle/: mx_scale_byte.
__device__ __forceinline__ int rt_mx_scale_byte(float amax) {
  const uint32_t u = __float_as_uint(amax);
  int s = (int)(u >> 23) - 8 + ((u & 0x7fffffu) > 0x600000u ? 1 : 0);
  return min(max(s, 1), 253);
}
// 2^-(s-127) as a float, exact for s in [1, 253]
__device__ __forceinline__ float rt_mx_inv_scale(int s) { return __uint_as_float((uint32_t)(254 - s) << 23); }

// Byte offset of the scale of (row, k) inside a scale tensor (reptext_hip.h, rt_gemm_group): planes of 1024 K-elements (8 K-tiles of
// 128); inside a plane 64-row chunks of 2 KiB ordered [K-tile][row & 15][32-block of the K-tile][row >> 4 & 3] — the order in which a
// GEMM wave wants them: 1 KiB = what one wave copies per eight K-tiles (four K-tiles of one chunk), one dword = the four fragment
// rows (16 apart) a lane feeds with the same K-block.
__device__ __forceinline__ int64_t rt_mx_scale_offset(int64_t row, int k, int64_t plane) {
  return (int64_t)(k >> 10) * plane + (row >> 6) * 2048 + ((k & 1023) >> 7) * 256 + (row & 15) * 16 + ((k & 127) >> 5) * 4 + ((row >> 4) & 3);
}

// max over the four lanes {l&15 + 16c, c = 0..3} (the 16-lane rows of a wave), result in all four: v_permlane16_swap exchanges the
// odd rows of its first operand with the even rows of its second, v_permlane32_swap the upper half with the lower half. Inline asm
// for the reason given at the row-max exchange of attention.hip (hipcc folds max(result[0], result[1]) of the builtins).
__device__ __forceinline__ float rt_max_over_lane_rows(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  float m = fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
  a = __builtin_bit_cast(unsigned, m);
  b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}
// max over the lane pair (l, l ^ 32)
__device__ __forceinline__ float rt_max_over_halves(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}

