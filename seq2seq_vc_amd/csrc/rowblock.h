// Shared pieces of the row-block kernels (attn_block.hip, gemm_rowpro.hip): a workgroup of 4 wavefronts owns 64 rows of a
// (rows, D) bf16 matrix (D <= 512), transforms them in registers (a wave owns rows wave, wave + 4, ...; a lane 8 consecutive
// channels), keeps the result in LDS as MFMA A operand and multiplies it with K-contiguous weight rows that stream from
// global memory straight into B fragments.
//
// These kernels run ONE wavefront per SIMD (the LDS tiles allow one workgroup per CU) on cold caches (a kernel boundary
// invalidates the L2 for data other XCDs wrote): nothing hides a memory round trip (~1-2 us) but the loads the same wave has
// in flight.  So every phase issues ALL its global loads before it touches the first result -- the row loads of a prologue
// as one batch of 16-byte loads, the weight fragments PF k-steps ahead of the MFMAs that consume them (registers renamed by
// full unrolling) and the first PF steps before the prologue even starts.
//
// ... and it runs COLD CODE: the instruction cache is invalidated per dispatch and one workgroup per CU executes every
// instruction once, so straight-line code costs ~0.45 us per KB fetched (profiles/r03_*: the first versions of these kernels
// were 70-100 KB of fully unrolled row loops and inlined epilogues and took 45-70 us, whatever the loads did).  Hence: row
// loops are ROLLED and work on an LDS image of the tile (a cooperative 16-byte copy in, the per-row arithmetic in place),
// bf16 packing uses the hardware converter (one instruction per pair instead of ~12), only the short k loops are unrolled.
#pragma once
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace rowblock {

// barrier for LDS hand-offs only: __syncthreads() also drains the vector-memory counter, i.e. it would wait for every global
// store issued so far to be acknowledged (~a memory round trip per barrier with one wave per SIMD)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// ---- weight fragments: wave-owned 16-column tiles, tile j of this wave = rows wrow[j] + lane part, k step ks = + ks * 32 ----
template <int MAXT, int PF>
__device__ __forceinline__ void preload_b(const bf16_t* const (&wrow)[MAXT], const bool (&live)[MAXT], bf16x8_t (&pre)[PF][MAXT]) {
#pragma unroll
  for (int s = 0; s < PF; ++s)
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
      if (live[j]) pre[s][j] = *reinterpret_cast<const bf16x8_t*>(wrow[j] + s * 32);
}

// acc[j][mt] += A[mt-th 16-row tile][D] . B_j^T over all D / 32 k steps; A from the LDS image `As` (pitch AP elements)
template <int MAXT, int KS, int PF>
__device__ __forceinline__ void mma_rows64(const bf16_t* As, int AP, const bf16_t* const (&wrow)[MAXT], const bool (&live)[MAXT],
                                           bf16x8_t (&pre)[PF][MAXT], f32x4_t (&acc)[MAXT][4]) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8_t cur[MAXT];
#pragma unroll
    for (int j = 0; j < MAXT; ++j) cur[j] = pre[ks % PF][j];
    if (ks + PF < KS) {
#pragma unroll
      for (int j = 0; j < MAXT; ++j)
        if (live[j]) pre[ks % PF][j] = *reinterpret_cast<const bf16x8_t*>(wrow[j] + (ks + PF) * 32);
    }
    bf16x8_t af[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const bf16x8_t*>(As + (mt * 16 + lr) * AP + ks * 32 + lg * 8);
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
      if (live[j]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[j][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], cur[j], acc[j][mt], 0, 0, 0);
      }
  }
}

// ---- bf16 packing (common.h: the hardware converter) ----
__device__ __forceinline__ uint32_t pack2(float a, float b) { return f2bf2(a, b); }
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) { return pack_bf16x8(f); }
__device__ __forceinline__ bf16_t cvt1(float f) { return f2bf(f); }

// ---- a 64 x D tile of dense rows <-> LDS image [64][D + 8]: all 256 threads, piece p = t + 256 * i (row = p / (D/8)), every
// load issued before the first store (the loads of several tiles are in flight together) ----
template <int D>
__device__ __forceinline__ void tile_load(const bf16_t* row0, int rows_valid, uint4 (&r)[D / 32]) {
  constexpr int VPR = D / 8;
#pragma unroll
  for (int i = 0; i < D / 32; ++i) {
    const int p = threadIdx.x + 256 * i;
    const int row = p / VPR, c = p - row * VPR;
    r[i] = make_uint4(0, 0, 0, 0);
    if (row < rows_valid) r[i] = *reinterpret_cast<const uint4*>(row0 + (int64_t)row * D + c * 8);
  }
}
template <int D>
__device__ __forceinline__ void tile_store(const uint4 (&r)[D / 32], bf16_t* lds) {
  constexpr int VPR = D / 8;
#pragma unroll
  for (int i = 0; i < D / 32; ++i) {
    const int p = threadIdx.x + 256 * i;
    const int row = p / VPR, c = p - row * VPR;
    *reinterpret_cast<uint4*>(lds + row * (D + 8) + c * 8) = r[i];
  }
}

}  // namespace rowblock
