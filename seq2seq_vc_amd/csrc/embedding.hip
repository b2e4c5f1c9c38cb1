// Token embedding lookup of Transformer-TTS and its weight gradient.
// reference: models/transformer_tts.py:63-77 (torch.nn.Embedding(idim, adim, padding_idx=0) as encoder input layer).
//   forward : y[i, :] = W[idx[i], :]                       (the row of padding_idx is whatever W holds: torch semantics)
//   backward: dW[v, :] = sum_{i : idx[i] == v} dy[i, :]    with dW[padding_idx, :] = 0
// The gradient is a deterministic gather-reduce (one workgroup per vocabulary row x 256 columns adds the rows of its tokens
// in token order) rather than an atomic scatter: vocabularies here are tens of symbols, token counts ~10^3 .. 10^4.
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

template <typename T>
__global__ void embedding_fwd_kernel(int64_t n, int D, int V, const int64_t* __restrict__ idx, const float* __restrict__ w,
                                     T* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * D; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    const int d = (int)(i - r * D);
    const int64_t v = idx[r];
    stf(y + i, (v >= 0 && v < V) ? w[v * D + d] : 0.f);
  }
}

// One workgroup per (vocabulary row v, 256 columns).  Round 6: the token list is no longer walked one token per dependent load (1200
// tokens = 100 us on the tail of the Transformer-TTS backward pass): 2048 tokens at a time are tested by the whole workgroup (8 loads
// per thread in flight), the positions that hold v are compacted IN ORDER into LDS (wave ballots + prefix counts), and only those rows
// of dy are added -- in increasing token order, so the sums keep their bits.
template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(int64_t n, int D, const int64_t* __restrict__ idx,
                                                            const T* __restrict__ dy, int64_t padding_idx,
                                                            float* __restrict__ dw, int accumulate) {
  __shared__ int list[2048];
  __shared__ int wave_cnt[4];
  __shared__ int total;
  const int v = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int d = blockIdx.y * 256 + t;
  float acc = 0.f;
  if (v != padding_idx) {
    for (int64_t base = 0; base < n; base += 2048) {
      int64_t tok[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int64_t i = base + c * 256 + t;
        tok[c] = i < n ? idx[i] : -1;
      }
      if (t == 0) total = 0;
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const bool hit = tok[c] == v;
        const uint64_t m = __ballot(hit);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = total;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = c * 256 + t;
        __syncthreads();
        if (t == 0) total += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
      }
      const int cnt = total;
      if (d < D) {
        for (int k0 = 0; k0 < cnt; k0 += 8) {
          float r[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) r[k] = (k0 + k < cnt) ? ldf(dy + (base + list[k0 + k]) * D + d) : 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (k0 + k < cnt) acc += r[k];
        }
      }
      __syncthreads();
    }
  }
  if (d >= D) return;
  float* o = dw + (int64_t)v * D + d;
  *o = (accumulate ? *o : 0.f) + acc;
}

}  // namespace

extern "C" int s2svc_embedding_fwd(int dtype, int64_t n, int D, int V, const int64_t* idx, const float* w, void* y, void* stream) {
  S2S_REQUIRE(n >= 0 && D > 0 && V > 0 && idx && w && y, "embedding_fwd: bad arguments");
  if (n == 0) return 0;
  int64_t nb = (n * D + 255) / 256;
  if (nb > 65535) nb = 65535;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32) hipLaunchKernelGGL(embedding_fwd_kernel<float>, dim3((unsigned)nb), dim3(256), 0, st, n, D, V, idx, w, (float*)y);
  else hipLaunchKernelGGL(embedding_fwd_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, n, D, V, idx, w, (bf16_t*)y);
  S2S_CHECK_LAUNCH("embedding_fwd_kernel");
  return 0;
}

extern "C" int s2svc_embedding_bwd(int dtype, int64_t n, int D, int V, const int64_t* idx, const void* dy, int64_t padding_idx,
                                   float* dw, int accumulate, void* stream) {
  S2S_REQUIRE(n >= 0 && D > 0 && V > 0 && idx && dw, "embedding_bwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(V, (D + 255) / 256);
  if (dtype == S2S_F32) hipLaunchKernelGGL(embedding_bwd_kernel<float>, grid, dim3(256), 0, st, n, D, idx, (const float*)dy, padding_idx, dw, accumulate);
  else hipLaunchKernelGGL(embedding_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, n, D, idx, (const bf16_t*)dy, padding_idx, dw, accumulate);
  S2S_CHECK_LAUNCH("embedding_bwd_kernel");
  return 0;
}
