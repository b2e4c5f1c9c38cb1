// Depthwise (groups = channels) 1-D convolution on channel-last activations, with dilation.
//   y[b,t,c] = bias[c] + sum_j w[c,j] * x[b, t + (j - (k-1)/2)*dil, c]        ('same' zero padding)
// reference: modules/conformer/convolution.py:42-51,70 (depthwise_conv, k = 7/15/31) and the dilated
// depth-separable convs of the VITS flows (modules/vits/flow.py:137-146).  HBM-bound: lanes run along the
// contiguous channel axis (coalesced), the k taps of a (t, c) output re-read neighbouring rows from L1/L2.
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

template <typename T>
__global__ void dwconv_fwd_kernel(int B, int Tn, int C, int ks, int dil, const T* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ bias, T* __restrict__ y, int flip) {
  const int64_t n = (int64_t)B * Tn * C;
  const int pad = (ks - 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t bt = i / C;
    const int t = (int)(bt % Tn);
    float acc = (bias && !flip) ? bias[c] : 0.f;
    for (int j = 0; j < ks; ++j) {
      const int tt = t + (j - pad) * dil;
      if (tt >= 0 && tt < Tn) {
        const float wv = flip ? w[c * ks + (ks - 1 - j)] : w[c * ks + j];
        acc += wv * ldf(x + i + (int64_t)(tt - t) * C);
      }
    }
    stf(y + i, acc);
  }
}

// partial dw[chunk][c][j] = sum over the chunk's rows of dy[b,t,c] * x[b, t+(j-pad)*dil, c]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(int B, int Tn, int C, int ks, int dil, const T* __restrict__ x,
                                                           const T* __restrict__ dy, float* __restrict__ ws,
                                                           int rows_per_chunk) {
  __shared__ float sh[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int chunk = blockIdx.y, j = blockIdx.z;
  const int pad = (ks - 1) / 2;
  const int rows = B * Tn;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = (r0 + rows_per_chunk < rows) ? r0 + rows_per_chunk : rows;
  float acc = 0.f;
  if (c < C) {
    for (int r = r0 + rl; r < r1; r += 4) {
      const int t = r % Tn;
      const int tt = t + (j - pad) * dil;
      if (tt >= 0 && tt < Tn) acc += ldf(dy + (int64_t)r * C + c) * ldf(x + (int64_t)(r + tt - t) * C + c);
    }
  }
  sh[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < C) ws[((int64_t)chunk * C + c) * ks + j] = sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl];
}

__global__ void dwconv_wgrad_final_kernel(int n, int chunks, const float* __restrict__ ws, float* __restrict__ dw, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float t = 0.f;
  for (int k = 0; k < chunks; ++k) t += ws[(int64_t)k * n + i];
  dw[i] = (accumulate ? dw[i] : 0.f) + t;
}

inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

// flip = 0: forward.  flip = 1: data gradient (same kernel on dy with reversed taps, no bias).
extern "C" int s2svc_dwconv(int dtype, int B, int Tn, int C, int ks, int dil, const void* x, const float* w,
                            const float* bias, void* y, int flip, void* stream) {
  const int64_t n = (int64_t)B * Tn * C;
  if (n == 0) return 0;
  S2S_REQUIRE(ks >= 1 && (ks & 1) && dil >= 1, "dwconv: kernel size must be odd, dilation >= 1");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(dwconv_fwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, B, Tn, C, ks, dil, (const float*)x, w, bias, (float*)y, flip);
  else
    hipLaunchKernelGGL(dwconv_fwd_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, B, Tn, C, ks, dil, (const bf16_t*)x, w, bias, (bf16_t*)y, flip);
  S2S_CHECK_LAUNCH("dwconv_fwd_kernel");
  return 0;
}

// ws: >= ws_chunks * C * ks floats
extern "C" int s2svc_dwconv_wgrad(int dtype, int B, int Tn, int C, int ks, int dil, const void* x, const void* dy,
                                  float* dw, int accumulate, float* ws, int ws_chunks, void* stream) {
  const int rows = B * Tn;
  if (rows == 0) return 0;
  S2S_REQUIRE(ws && ws_chunks > 0, "dwconv_wgrad: workspace required");
  hipStream_t st = (hipStream_t)stream;
  int chunks = (rows + 127) / 128;
  if (chunks > ws_chunks) chunks = ws_chunks;
  const int rpc = (rows + chunks - 1) / chunks;
  dim3 grid((C + 63) / 64, chunks, ks);
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(dwconv_wgrad_kernel<float>, grid, dim3(256), 0, st, B, Tn, C, ks, dil, (const float*)x, (const float*)dy, ws, rpc);
  else
    hipLaunchKernelGGL(dwconv_wgrad_kernel<bf16_t>, grid, dim3(256), 0, st, B, Tn, C, ks, dil, (const bf16_t*)x, (const bf16_t*)dy, ws, rpc);
  S2S_CHECK_LAUNCH("dwconv_wgrad_kernel");
  const int n = C * ks;
  hipLaunchKernelGGL(dwconv_wgrad_final_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, chunks, ws, dw, accumulate);
  S2S_CHECK_LAUNCH("dwconv_wgrad_final_kernel");
  return 0;
}
