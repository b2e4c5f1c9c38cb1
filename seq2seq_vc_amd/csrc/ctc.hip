// Forward-sum alignment loss (CTC over the attention matrix) and its beta-binomial prior, batched on
// the GPU: one workgroup per utterance runs the alpha recursion, the beta recursion and the gradient in a
// single launch (fp32, like torch's CPU kernel).
//
// reference: losses/forward_sum_loss.py:26-76 -- per-utterance Python loop over
//   F.ctc_loss(log_probs = pad_blank(log_p_attn + prior)[:T_b, :N_b+1], targets = 1..N_b,
//              zero_infinity=True, reduction='mean')  ->  nll_b / N_b ;  loss = sum_b / B
// and :78-116 (_generate_prior: scipy.stats.betabinom.logpmf(k, N, t, T-t+1), t = 1..T).
// The gradient reproduces torch's ctc_loss backward exactly (Graves eq. 16 form):
//   d/d lp[t,c] = exp(lp[t,c]) - exp(logsumexp_{s: l_s = c}(alpha_t(s) + beta_t(s)) + nll - lp[t,c])
// (the blank column is a constant pad, so only the label columns receive gradient).
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

constexpr int CTC_THREADS = 256;
constexpr int CTC_KMAX = 8;  // extended length S = 2N+1 <= 2048
constexpr int CTC_PD = 4;    // steps of prefetch distance in the one-position-per-thread path

__device__ __forceinline__ float lse2(float a, float b) {
  const float m = fmaxf(a, b);
  if (m == -__builtin_huge_valf()) return m;
  return logf(expf(a - m) + expf(b - m)) + m;
}
__device__ __forceinline__ float lse3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == -__builtin_huge_valf()) return m;
  return logf(expf(a - m) + expf(b - m) + expf(c - m)) + m;
}

__global__ __launch_bounds__(CTC_THREADS) void forward_sum_kernel(int B, int Tf, int Tx, const float* __restrict__ logp,
                                                                  const float* __restrict__ prior,
                                                                  const int32_t* __restrict__ text_lens,
                                                                  const int32_t* __restrict__ feat_lens, float log_blank,
                                                                  float* __restrict__ alpha_ws, float* __restrict__ loss_b,
                                                                  float* __restrict__ grad) {
  extern __shared__ float sh[];  // 2 columns of S_pad floats (+ 2 guard cells each side)
  const int b = blockIdx.x, tid = threadIdx.x;
  int N = text_lens[b], Tn = feat_lens[b];
  if (N > Tx) N = Tx;
  if (Tn > Tf) Tn = Tf;
  const int S = 2 * N + 1;
  const int Spad = 2 * Tx + 1;
  const float NINF = -__builtin_huge_valf();
  const float* lpb = logp + (int64_t)b * Tf * Tx;
  const float* prb = prior + (int64_t)b * Tf * Tx;
  float* gb = grad + (int64_t)b * Tf * Tx;
  float* aw = alpha_ws + (int64_t)b * Tf * Spad;
  for (int i = tid; i < Tf * Tx; i += CTC_THREADS) gb[i] = 0.f;
  if (N <= 0 || Tn <= 0) {
    if (tid == 0) loss_b[b] = 0.f;
    return;
  }
  float* col0 = sh + 2;
  float* col1 = sh + 2 + (Spad + 4);
  if (tid < 2) { sh[tid] = NINF; sh[(Spad + 4) + tid] = NINF; }

  // lp(t, s): log-prob of the symbol at extended position s
  auto lp = [&](int t, int s) -> float {
    if ((s & 1) == 0) return log_blank;
    const int j = (s - 1) >> 1;
    return lpb[(int64_t)t * Tx + j] + prb[(int64_t)t * Tx + j];
  };

  // ---- alpha ----
  for (int k = 0; k < CTC_KMAX; ++k) {
    const int s = tid + k * CTC_THREADS;
    if (s < S) {
      const float v = (s < 2) ? lp(0, s) : NINF;
      col0[s] = v;
      aw[s] = v;
    }
  }
  __syncthreads();
  float* prev = col0;
  float* cur = col1;
  // S <= CTC_THREADS (one extended position per thread, the usual case): the two global loads behind lp(t, s) are issued
  // CTC_PD steps ahead, so a step of the recursion is an LDS round trip + one lse3, not a trip to L2 / HBM
  const bool one_per_thread = S <= CTC_THREADS;
  if (one_per_thread) {
    const int s = tid;
    const bool act = s < S;
    float lq[CTC_PD];
#pragma unroll
    for (int q = 0; q < CTC_PD; ++q) lq[q] = (act && 1 + q < Tn) ? lp(1 + q, s) : 0.f;
    for (int t = 1; t < Tn; t += CTC_PD) {
#pragma unroll
      for (int q = 0; q < CTC_PD; ++q) {
        const int tt = t + q;
        if (tt >= Tn) break;
        const float l = lq[q];
        if (act && tt + CTC_PD < Tn) lq[q] = lp(tt + CTC_PD, s);
        if (act) {
          const float a1 = prev[s], a2 = prev[s - 1];
          const float a3 = ((s & 1) && s >= 3) ? prev[s - 2] : NINF;
          const float v = lse3(a1, a2, a3) + l;
          cur[s] = v;
          aw[(int64_t)tt * Spad + s] = v;
        }
        __syncthreads();
        float* tmp = prev; prev = cur; cur = tmp;
      }
    }
  }
  for (int t = 1; t < Tn && !one_per_thread; ++t) {
    for (int k = 0; k < CTC_KMAX; ++k) {
      const int s = tid + k * CTC_THREADS;
      if (s < S) {
        const float a1 = prev[s], a2 = prev[s - 1];
        const float a3 = ((s & 1) && s >= 3) ? prev[s - 2] : NINF;
        const float v = lse3(a1, a2, a3) + lp(t, s);
        cur[s] = v;
        aw[(int64_t)t * Spad + s] = v;
      }
    }
    __syncthreads();
    float* tmp = prev; prev = cur; cur = tmp;
  }
  const float nll = -lse2(prev[S - 1], S >= 2 ? prev[S - 2] : NINF);
  const bool inf_loss = !(nll < __builtin_huge_valf());  // zero_infinity=True
  if (tid == 0) loss_b[b] = inf_loss ? 0.f : nll / (float)(N < 1 ? 1 : N);
  if (inf_loss) return;
  __syncthreads();

  // ---- beta + gradient ----
  // guard cells above S: reads of s+1, s+2 beyond the end must see -inf
  float* bprev = col0;
  float* bcur = col1;
  for (int k = 0; k < CTC_KMAX; ++k) {
    const int s = tid + k * CTC_THREADS;
    if (s < S + 2 && s < Spad + 2) { bprev[s] = NINF; bcur[s] = NINF; }
  }
  __syncthreads();
  for (int k = 0; k < CTC_KMAX; ++k) {
    const int s = tid + k * CTC_THREADS;
    if (s < S) {
      const float v = (s >= S - 2) ? lp(Tn - 1, s) : NINF;
      bprev[s] = v;
      if (s & 1) {
        const int j = (s - 1) >> 1;
        const float l = lp(Tn - 1, s);
        const float ab = aw[(int64_t)(Tn - 1) * Spad + s] + v;
        gb[(int64_t)(Tn - 1) * Tx + j] = expf(l) - expf(ab + nll - l);
      }
    }
  }
  __syncthreads();
  if (one_per_thread) {
    const int s = tid;
    const bool act = s < S, lab = act && (s & 1);
    const int j = (s - 1) >> 1;
    float lq[CTC_PD], aq[CTC_PD];
#pragma unroll
    for (int q = 0; q < CTC_PD; ++q) {
      const int tt = Tn - 2 - q;
      lq[q] = (act && tt >= 0) ? lp(tt, s) : 0.f;
      aq[q] = (lab && tt >= 0) ? aw[(int64_t)tt * Spad + s] : 0.f;
    }
    for (int t = Tn - 2; t >= 0; t -= CTC_PD) {
#pragma unroll
      for (int q = 0; q < CTC_PD; ++q) {
        const int tt = t - q;
        if (tt < 0) break;
        const float l = lq[q], a = aq[q];
        if (act && tt - CTC_PD >= 0) lq[q] = lp(tt - CTC_PD, s);
        if (lab && tt - CTC_PD >= 0) aq[q] = aw[(int64_t)(tt - CTC_PD) * Spad + s];
        if (act) {
          const float b1 = bprev[s], b2 = bprev[s + 1];
          const float b3 = ((s & 1) && s + 2 < S) ? bprev[s + 2] : NINF;
          const float v = lse3(b1, b2, b3) + l;
          bcur[s] = v;
          if (lab) gb[(int64_t)tt * Tx + j] = expf(l) - expf(a + v + nll - l);
        }
        __syncthreads();
        float* tmp = bprev; bprev = bcur; bcur = tmp;
      }
    }
  }
  for (int t = Tn - 2; t >= 0 && !one_per_thread; --t) {
    for (int k = 0; k < CTC_KMAX; ++k) {
      const int s = tid + k * CTC_THREADS;
      if (s < S) {
        const float b1 = bprev[s], b2 = bprev[s + 1];
        const float b3 = ((s & 1) && s + 2 < S) ? bprev[s + 2] : NINF;
        const float l = lp(t, s);
        const float v = lse3(b1, b2, b3) + l;
        bcur[s] = v;
        if (s & 1) {
          const int j = (s - 1) >> 1;
          const float ab = aw[(int64_t)t * Spad + s] + v;
          gb[(int64_t)t * Tx + j] = expf(l) - expf(ab + nll - l);
        }
      }
    }
    __syncthreads();
    float* tmp = bprev; bprev = bcur; bcur = tmp;
  }
  // scale by 1 / (N * B): reduction='mean' over the single-item batch divides by target length
  __syncthreads();
  const float sc = 1.f / ((float)(N < 1 ? 1 : N) * (float)B);
  for (int i = tid; i < Tn * Tx; i += CTC_THREADS) {
    const int j = i % Tx;
    gb[i] = (j < N) ? gb[i] * sc : 0.f;
  }
}

// prior[b,t,j] = log BetaBinomial(k=j | n=N_b, a=t+1, b=T_b-t), -inf outside the valid region (fp64 math)
__global__ void betabinom_prior_kernel(int B, int Tf, int Tx, const int32_t* __restrict__ text_lens,
                                       const int32_t* __restrict__ feat_lens, float* __restrict__ prior) {
  const int64_t n = (int64_t)B * Tf * Tx;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % Tx);
    const int64_t r = i / Tx;
    const int t = (int)(r % Tf);
    const int b = (int)(r / Tf);
    const int N = text_lens[b], Tn = feat_lens[b];
    float out = -__builtin_huge_valf();
    if (t < Tn && j < N) {
      const double nn = (double)N, k = (double)j, a = (double)(t + 1), bb = (double)(Tn - t);
      const double comb = lgamma(nn + 1.0) - lgamma(k + 1.0) - lgamma(nn - k + 1.0);
      const double num = lgamma(k + a) + lgamma(nn - k + bb) - lgamma(nn + a + bb);
      const double den = lgamma(a) + lgamma(bb) - lgamma(a + bb);
      out = (float)(comb + num - den);
    }
    prior[i] = out;
  }
}

}  // namespace

extern "C" int64_t s2svc_forward_sum_ws_bytes(int B, int Tf, int Tx) {
  return (int64_t)B * Tf * (2 * Tx + 1) * 4;
}

// loss_b: (B,) per-utterance nll/N (sum/B is the reference loss); grad: (B,Tf,Tx) d(sum_b loss_b / B)/d(log_p_attn)
extern "C" int s2svc_forward_sum(int B, int Tf, int Tx, const float* log_p_attn, const float* prior,
                                 const int32_t* text_lens, const int32_t* feat_lens, float log_blank, void* ws,
                                 float* loss_b, float* grad, void* stream) {
  S2S_REQUIRE(B >= 0 && Tf > 0 && Tx > 0, "forward_sum: bad shape");
  S2S_REQUIRE(2 * Tx + 1 <= CTC_THREADS * CTC_KMAX, "forward_sum: T_text too large");
  if (B == 0) return 0;
  const size_t shm = (size_t)2 * (2 * Tx + 1 + 4) * sizeof(float);
  hipLaunchKernelGGL(forward_sum_kernel, dim3(B), dim3(CTC_THREADS), shm, (hipStream_t)stream, B, Tf, Tx, log_p_attn, prior,
                     text_lens, feat_lens, log_blank, (float*)ws, loss_b, grad);
  S2S_CHECK_LAUNCH("forward_sum_kernel");
  return 0;
}

extern "C" int s2svc_betabinom_prior(int B, int Tf, int Tx, const int32_t* text_lens, const int32_t* feat_lens,
                                     float* prior, void* stream) {
  const int64_t n = (int64_t)B * Tf * Tx;
  if (n == 0) return 0;
  int nb = (int)((n + 255) / 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(betabinom_prior_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, B, Tf, Tx, text_lens, feat_lens, prior);
  S2S_CHECK_LAUNCH("betabinom_prior_kernel");
  return 0;
}
