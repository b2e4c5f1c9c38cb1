// Duration-predictor loss: MSE in the log domain over non-padded tokens.
// reference: losses/duration_predictor_loss.py:38-57
//   loss = mean_{t < ilens[b]} (d_outs[b,t] - log(ds[b,t] + offset))^2        (masked_select + MSELoss(mean))
// (B, T_text) is a few thousand values: one workgroup, fixed-order block reduction (deterministic).
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return t;
}

// stats[0] = number of valid tokens, out[0] = loss
__global__ __launch_bounds__(256) void dur_loss_fwd_kernel(int B, int Tn, const float* __restrict__ d, const float* __restrict__ ds,
                                                           const int32_t* __restrict__ lens, float offset, int mean,
                                                           float* __restrict__ stats, float* __restrict__ out) {
  __shared__ float sh[4];
  float acc = 0.f, cnt = 0.f;
  for (int i = threadIdx.x; i < B * Tn; i += 256) {
    const int b = i / Tn, t = i - b * Tn;
    if (!lens || t < lens[b]) {
      const float e = d[i] - logf(ds[i] + offset);
      acc += e * e;
      cnt += 1.f;
    }
  }
  acc = block_sum(acc, sh);
  cnt = block_sum(cnt, sh);
  if (threadIdx.x == 0) {
    stats[0] = cnt;
    out[0] = mean ? acc / cnt : acc;
  }
}

__global__ void dur_loss_bwd_kernel(int B, int Tn, const float* __restrict__ d, const float* __restrict__ ds,
                                    const int32_t* __restrict__ lens, float offset, int mean, const float* __restrict__ stats,
                                    const float* __restrict__ g, float* __restrict__ dd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Tn) return;
  const int b = i / Tn, t = i - b * Tn;
  float v = 0.f;
  if (!lens || t < lens[b]) v = 2.f * (d[i] - logf(ds[i] + offset)) * g[0] * (mean ? 1.f / stats[0] : 1.f);
  dd[i] = v;
}

}  // namespace

extern "C" int s2svc_duration_loss_fwd(int B, int Tn, const float* d_outs, const float* ds, const int32_t* lens, float offset,
                                       int mean, float* stats, float* out, void* stream) {
  S2S_REQUIRE(B > 0 && Tn > 0 && d_outs && ds && stats && out, "duration_loss_fwd: bad arguments");
  hipLaunchKernelGGL(dur_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, B, Tn, d_outs, ds, lens, offset, mean, stats, out);
  S2S_CHECK_LAUNCH("dur_loss_fwd_kernel");
  return 0;
}

extern "C" int s2svc_duration_loss_bwd(int B, int Tn, const float* d_outs, const float* ds, const int32_t* lens, float offset,
                                       int mean, const float* stats, const float* g, float* dd, void* stream) {
  S2S_REQUIRE(B > 0 && Tn > 0 && d_outs && ds && stats && g && dd, "duration_loss_bwd: bad arguments");
  hipLaunchKernelGGL(dur_loss_bwd_kernel, dim3((B * Tn + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, Tn, d_outs, ds, lens,
                     offset, mean, stats, g, dd);
  S2S_CHECK_LAUNCH("dur_loss_bwd_kernel");
  return 0;
}
