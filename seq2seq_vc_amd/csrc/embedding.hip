// Token embedding lookup of Transformer-TTS and its weight gradient.
// reference: models/transformer_tts.py:63-77 (torch.nn.Embedding(idim, adim, padding_idx=0) as encoder input layer).
//   forward : y[i, :] = W[idx[i], :]                       (the row of padding_idx is whatever W holds: torch semantics)
//   backward: dW[v, :] = sum_{i : idx[i] == v} dy[i, :]    with dW[padding_idx, :] = 0
// The gradient is a deterministic gather-reduce (one workgroup per vocabulary row x 256 columns walks the token list
// in order) rather than an atomic scatter: vocabularies here are tens of symbols, token counts ~10^4.
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

template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(int64_t n, int D, const int64_t* __restrict__ idx,
                                                            const T* __restrict__ dy, int64_t padding_idx,
                                                            float* __restrict__ dw, int accumulate) {
  const int v = blockIdx.x;
  const int d = blockIdx.y * 256 + threadIdx.x;
  if (d >= D) return;
  float acc = 0.f;
  if (v != padding_idx) {
    for (int64_t i = 0; i < n; ++i)
      if (idx[i] == v) acc += ldf(dy + i * D + d);          // uniform branch: every thread of the block tests the same token
  }
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
