// Masked sequence losses, fused forward (deterministic two-stage reduction) and backward.
//
// reference: losses/seq2seq_loss.py:30-59 (masked L1(after)+L1(before), BCEWithLogits pos_weight),
//            losses/l1_loss.py:22-49, losses/guided_attention_loss.py:142-165,
//            losses/duration_predictor_loss.py:38-57.
// The reference materialises five masked_select copies; here the length mask is evaluated from
// olens inside the kernel and nothing but the inputs is read.
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  __syncthreads();
  return t;  // valid on thread 0
}

// partial[blk][0] = sum |after-ys| ; [1] = sum |before-ys| ; [2] = sum bce terms     over valid frames
template <typename T>
__global__ __launch_bounds__(256) void seq_loss_fwd_kernel(int B, int Tm, int D, const T* __restrict__ after,
                                                           const T* __restrict__ before, const T* __restrict__ logits,
                                                           const float* __restrict__ ys, const float* __restrict__ labels,
                                                           const int32_t* __restrict__ olens, float pos_weight,
                                                           float* __restrict__ partial) {
  __shared__ float sh[4];
  const int64_t n = (int64_t)B * Tm * D;
  float a = 0.f, b = 0.f, c = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bt = i / D;
    const int t = (int)(bt % Tm), bb = (int)(bt / Tm);
    if (t < olens[bb]) {
      const float y = ys[i];
      if (after) a += fabsf(ldf(after + i) - y);
      b += fabsf(ldf(before + i) - y);
      if (logits && (i % D) == 0) {
        const float x = ldf(logits + bt), lab = labels[bt];
        const float lw = 1.f + (pos_weight - 1.f) * lab;
        c += (1.f - lab) * x + lw * (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.f));
      }
    }
  }
  float t0 = block_sum(a, sh), t1 = block_sum(b, sh), t2 = block_sum(c, sh);
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 3 + 0] = t0;
    partial[blockIdx.x * 3 + 1] = t1;
    partial[blockIdx.x * 3 + 2] = t2;
  }
}

// out[0] = l1 = mean|after-ys| + mean|before-ys| ; out[1] = bce ; out[2] = #valid frames
__global__ void seq_loss_final_kernel(int nblk, int B, int Tm, int D, const float* __restrict__ partial,
                                      const int32_t* __restrict__ olens, float* __restrict__ out) {
  float a = 0.f, b = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 64) {
    a += partial[i * 3 + 0]; b += partial[i * 3 + 1]; c += partial[i * 3 + 2];
  }
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  float cnt = 0.f;
  for (int i = threadIdx.x; i < B; i += 64) { int l = olens[i]; cnt += (float)(l < Tm ? l : Tm); }
  cnt = wave_sum(cnt);
  if (threadIdx.x == 0) {
    out[0] = a / (cnt * D) + b / (cnt * D);
    out[1] = c / cnt;
    out[2] = cnt;
  }
}

template <typename T>
__global__ void seq_loss_bwd_kernel(int B, int Tm, int D, const T* __restrict__ after, const T* __restrict__ before,
                                    const T* __restrict__ logits, const float* __restrict__ ys,
                                    const float* __restrict__ labels, const int32_t* __restrict__ olens, float pos_weight,
                                    const float* __restrict__ stats, const float* __restrict__ g_l1,
                                    const float* __restrict__ g_bce, T* __restrict__ d_after, T* __restrict__ d_before,
                                    T* __restrict__ d_logits) {
  const int64_t n = (int64_t)B * Tm * D;
  const float cnt = stats[2];
  const float s1 = (g_l1 ? *g_l1 : 1.f) / (cnt * D);
  const float s2 = (g_bce ? *g_bce : 1.f) / cnt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bt = i / D;
    const int t = (int)(bt % Tm), bb = (int)(bt / Tm);
    const bool ok = t < olens[bb];
    const float y = ys[i];
    if (d_after) {
      const float d = ldf(after + i) - y;
      stf(d_after + i, ok ? s1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 0.f);
    }
    {
      const float d = ldf(before + i) - y;
      stf(d_before + i, ok ? s1 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 0.f);
    }
    if (d_logits && (i % D) == 0) {
      float gl = 0.f;
      if (ok) {
        const float x = ldf(logits + bt), lab = labels[bt];
        const float lw = 1.f + (pos_weight - 1.f) * lab;
        const float sg = 1.f / (1.f + expf(-x));
        gl = s2 * ((1.f - lab) - lw * (1.f - sg));
      }
      stf(d_logits + bt, gl);
    }
  }
}

// guided attention: sum over valid (b,h,to,ti) of (1-exp(-((ti/il - to/ol)^2)/(2 sigma^2))) * att
template <typename T>
__global__ __launch_bounds__(256) void ga_loss_fwd_kernel(int B, int H, int To, int Ti, const T* __restrict__ att,
                                                          const int32_t* __restrict__ ilens,
                                                          const int32_t* __restrict__ olens, float sigma,
                                                          float* __restrict__ partial) {
  __shared__ float sh[4];
  const int64_t n = (int64_t)B * H * To * Ti;
  float a = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ti = (int)(i % Ti);
    const int64_t r = i / Ti;
    const int to = (int)(r % To);
    const int b = (int)(r / To / H);
    const int il = ilens[b], ol = olens[b];
    if (ti < il && to < ol) {
      const float d = (float)ti / (float)il - (float)to / (float)ol;
      a += (1.f - expf(-(d * d) / (2.f * sigma * sigma))) * ldf(att + i);
    }
  }
  float t0 = block_sum(a, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t0;
}
__global__ void ga_loss_final_kernel(int nblk, int B, int H, int To, int Ti, const float* __restrict__ partial,
                                     const int32_t* __restrict__ ilens, const int32_t* __restrict__ olens, float alpha,
                                     float* __restrict__ out) {
  float a = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 64) a += partial[i];
  a = wave_sum(a);
  float cnt = 0.f;
  for (int i = threadIdx.x; i < B; i += 64) {
    int il = ilens[i] < Ti ? ilens[i] : Ti, ol = olens[i] < To ? olens[i] : To;
    cnt += (float)il * (float)ol;
  }
  cnt = wave_sum(cnt) * H;
  if (threadIdx.x == 0) { out[0] = alpha * a / cnt; out[1] = cnt; }
}
template <typename T>
__global__ void ga_loss_bwd_kernel(int B, int H, int To, int Ti, const int32_t* __restrict__ ilens,
                                   const int32_t* __restrict__ olens, float sigma, float alpha,
                                   const float* __restrict__ stats, const float* __restrict__ gout, T* __restrict__ datt) {
  const int64_t n = (int64_t)B * H * To * Ti;
  const float s = (gout ? *gout : 1.f) * alpha / stats[1];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int ti = (int)(i % Ti);
    const int64_t r = i / Ti;
    const int to = (int)(r % To);
    const int b = (int)(r / To / H);
    const int il = ilens[b], ol = olens[b];
    float g = 0.f;
    if (ti < il && to < ol) {
      const float d = (float)ti / (float)il - (float)to / (float)ol;
      g = s * (1.f - expf(-(d * d) / (2.f * sigma * sigma)));
    }
    stf(datt + i, g);
  }
}

inline int red_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

}  // namespace

// partial: >= 3*1024 floats.  out: 3 floats {l1, bce, valid_frames}.  after/logits may be NULL.
extern "C" int s2svc_seq_loss_fwd(int dtype, int B, int Tm, int D, const void* after, const void* before,
                                  const void* logits, const float* ys, const float* labels, const int32_t* olens,
                                  float pos_weight, float* partial, float* out, void* stream) {
  S2S_REQUIRE(B > 0 && Tm > 0 && D > 0 && before && ys && olens && partial && out, "seq_loss_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int nb = red_blocks((int64_t)B * Tm * D);
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(seq_loss_fwd_kernel<float>, dim3(nb), dim3(256), 0, st, B, Tm, D, (const float*)after,
                       (const float*)before, (const float*)logits, ys, labels, olens, pos_weight, partial);
  else
    hipLaunchKernelGGL(seq_loss_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, B, Tm, D, (const bf16_t*)after,
                       (const bf16_t*)before, (const bf16_t*)logits, ys, labels, olens, pos_weight, partial);
  S2S_CHECK_LAUNCH("seq_loss_fwd_kernel");
  hipLaunchKernelGGL(seq_loss_final_kernel, dim3(1), dim3(64), 0, st, nb, B, Tm, D, partial, olens, out);
  S2S_CHECK_LAUNCH("seq_loss_final_kernel");
  return 0;
}

extern "C" int s2svc_seq_loss_bwd(int dtype, int B, int Tm, int D, const void* after, const void* before,
                                  const void* logits, const float* ys, const float* labels, const int32_t* olens,
                                  float pos_weight, const float* stats, const float* g_l1, const float* g_bce,
                                  void* d_after, void* d_before, void* d_logits, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)B * Tm * D;
  int nb = (int)((n + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(seq_loss_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, B, Tm, D, (const float*)after,
                       (const float*)before, (const float*)logits, ys, labels, olens, pos_weight, stats, g_l1, g_bce,
                       (float*)d_after, (float*)d_before, (float*)d_logits);
  else
    hipLaunchKernelGGL(seq_loss_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, B, Tm, D, (const bf16_t*)after,
                       (const bf16_t*)before, (const bf16_t*)logits, ys, labels, olens, pos_weight, stats, g_l1, g_bce,
                       (bf16_t*)d_after, (bf16_t*)d_before, (bf16_t*)d_logits);
  S2S_CHECK_LAUNCH("seq_loss_bwd_kernel");
  return 0;
}

// partial >= 1024 floats; out: 2 floats {loss, valid_count}
extern "C" int s2svc_guided_attn_loss_fwd(int dtype, int B, int H, int To, int Ti, const void* att, const int32_t* ilens,
                                          const int32_t* olens, float sigma, float alpha, float* partial, float* out,
                                          void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int nb = red_blocks((int64_t)B * H * To * Ti);
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(ga_loss_fwd_kernel<float>, dim3(nb), dim3(256), 0, st, B, H, To, Ti, (const float*)att, ilens, olens, sigma, partial);
  else
    hipLaunchKernelGGL(ga_loss_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, B, H, To, Ti, (const bf16_t*)att, ilens, olens, sigma, partial);
  S2S_CHECK_LAUNCH("ga_loss_fwd_kernel");
  hipLaunchKernelGGL(ga_loss_final_kernel, dim3(1), dim3(64), 0, st, nb, B, H, To, Ti, partial, ilens, olens, alpha, out);
  S2S_CHECK_LAUNCH("ga_loss_final_kernel");
  return 0;
}

extern "C" int s2svc_guided_attn_loss_bwd(int dtype, int B, int H, int To, int Ti, const int32_t* ilens,
                                          const int32_t* olens, float sigma, float alpha, const float* stats,
                                          const float* gout, void* datt, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)B * H * To * Ti;
  int nb = (int)((n + 255) / 256);
  if (nb > 2048) nb = 2048;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(ga_loss_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, B, H, To, Ti, ilens, olens, sigma, alpha, stats, gout, (float*)datt);
  else
    hipLaunchKernelGGL(ga_loss_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, B, H, To, Ti, ilens, olens, sigma, alpha, stats, gout, (bf16_t*)datt);
  S2S_CHECK_LAUNCH("ga_loss_bwd_kernel");
  return 0;
}
