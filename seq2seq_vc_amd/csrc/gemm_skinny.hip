// Skinny MFMA GEMM: C[M <= 64, N] = A[M, K] . B[N, K]^T with both operands dense and K-contiguous -- the shape
// of every projection in an autoregressive decode step (M = utterances in flight; reference:
// modules/transformer/decoder.py:239-273 runs these as 1-row torch Linear calls per step and per layer).
//
// The 64x64-tile kernel would put such a problem on N/64 workgroups and stage operands through LDS; here the
// weight matrix is spread over N/16 workgroups, each wavefront streams its K-quarter of a 16-row weight panel
// straight from global memory into MFMA fragments (16-byte loads, 4 k-steps in flight), the four K-quarters
// are summed through LDS, and the epilogue (alpha, bias, activation, residual, dtype) is the common one.
// Pure weight streaming: bytes = N*K*s per launch, each read exactly once.
#include <cstdlib>
#include "gemm_common.h"

namespace {

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  static constexpr int VEC = 8, KSTEP = 32;      // one mfma_f32_16x16x32_bf16 per step
  typedef bf16x8_t type;
  static __device__ __forceinline__ type zero() { return (type){0, 0, 0, 0, 0, 0, 0, 0}; }
  static __device__ __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Frag<float> {
  static constexpr int VEC = 4, KSTEP = 16;      // four mfma_f32_16x16x4f32 per step (lane element e <-> k = 4*(lane>>4)+e)
  typedef f32x4_t type;
  static __device__ __forceinline__ type zero() { return (type){0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
    return c;
  }
};

// LEAN: the common epilogue only (epilogue_store_lean; MT <= 2 -- the decode batches): 7-10 KB of the general epilogue's
// activation expansions per inlined copy are code a launch-bound kernel pays for without running it
template <typename T, int MT, bool LEAN = false>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const s2svc_gemm_desc d) {
  typedef Frag<T> F;
  typedef typename F::type frag_t;
  constexpr int UNROLL = 4;
  __shared__ float red[3][MT][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const T* A = (const T*)d.A.ptr;
  const T* B = (const T*)d.B.ptr;
  const int ksteps = (d.K + F::KSTEP - 1) / F::KSTEP;
  const int per = (ksteps + 3) / 4;
  const int ks0 = wave * per;
  const int ks1 = (ks0 + per < ksteps) ? ks0 + per : ksteps;
  const bool brow = (n0 + lr) < d.N;
  const T* bp = B + (int64_t)(n0 + lr) * d.B.ld + lg * F::VEC;
  const T* ap[MT];
  bool arow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    arow[i] = (i * 16 + lr) < d.M;
    ap[i] = A + (int64_t)(i * 16 + lr) * d.A.ld + lg * F::VEC;
  }
  f32x4_t acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  for (int ks = ks0; ks < ks1; ks += UNROLL) {
    frag_t a[UNROLL][MT], b[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const int k = (ks + u) * F::KSTEP + lg * F::VEC;
      const bool kin = (ks + u) < ks1 && k < d.K;        // K is a multiple of VEC: whole vectors only
      b[u] = (kin && brow) ? *reinterpret_cast<const frag_t*>(bp + (int64_t)(ks + u) * F::KSTEP) : F::zero();
#pragma unroll
      for (int i = 0; i < MT; ++i)
        a[u][i] = (kin && arow[i]) ? *reinterpret_cast<const frag_t*>(ap[i] + (int64_t)(ks + u) * F::KSTEP) : F::zero();
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i] = F::mma(a[u][i], b[u], acc[i]);
  }

  // sum the four K-quarters: waves 1..3 park their 16x16 partials in LDS, wave 0 adds them in a fixed order
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 1][i][r * 64 + lane] = acc[i][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = ((acc[i][r] + red[0][i][r * 64 + lane]) + red[1][i][r * 64 + lane]) + red[2][i][r * 64 + lane];
        const int m = i * 16 + lg * 4 + r, n = n0 + lr;
        if (m < d.M && n < d.N) {
          if (LEAN) epilogue_store_lean<false>(d, m, n, v);
          else epilogue_store_f(d, 0, 0, m, n, v);
        }
      }
  }
}

template <typename T>
void launch_skinny(const s2svc_gemm_desc& d, hipStream_t st) {
  dim3 grid((d.N + 15) / 16), block(256);
  const int mt = (d.M + 15) / 16;
  static const bool lean_on = true;
  if (lean_on && mt <= 2 && epilogue_lean_ok(d)) {
    if (mt == 1) hipLaunchKernelGGL((gemm_skinny_kernel<T, 1, true>), grid, block, 0, st, d);
    else hipLaunchKernelGGL((gemm_skinny_kernel<T, 2, true>), grid, block, 0, st, d);
    return;
  }
  if (mt == 1) hipLaunchKernelGGL((gemm_skinny_kernel<T, 1>), grid, block, 0, st, d);
  else if (mt == 2) hipLaunchKernelGGL((gemm_skinny_kernel<T, 2>), grid, block, 0, st, d);
  else if (mt == 3) hipLaunchKernelGGL((gemm_skinny_kernel<T, 3>), grid, block, 0, st, d);
  else hipLaunchKernelGGL((gemm_skinny_kernel<T, 4>), grid, block, 0, st, d);
}

}  // namespace

// returns 1 if the GEMM was launched here, 0 if the shape is not a skinny one (caller falls through)
extern "C" int s2svc_gemm_try_skinny(const s2svc_gemm_desc* desc, void* stream) {
  const s2svc_gemm_desc& d = *desc;
  if (d.M > 64 || d.nb0 * d.nb1 != 1 || d.splitk > 1 || d.a_rowsum) return 0;
  if (d.A.mode != S2SVC_OP_DENSE || d.B.mode != S2SVC_OP_DENSE) return 0;
  if (d.A.layout != S2SVC_LAYOUT_KC || d.B.layout != S2SVC_LAYOUT_KC) return 0;
  const int vec = d.dtype == S2S_F32 ? 4 : 8;
  if (d.K % vec || d.A.ld % vec || d.B.ld % vec) return 0;
  if (((uintptr_t)d.A.ptr) % 16 || ((uintptr_t)d.B.ptr) % 16) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (d.dtype == S2S_F32) launch_skinny<float>(d, st);
  else launch_skinny<bf16_t>(d, st);
  S2S_CHECK_LAUNCH("gemm_skinny_kernel");
  return 1;
}
