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
constexpr int CTC_PD = 4;    // steps of prefetch distance in the one-position-per-thread path (12 measured the same: the step is the LDS round trip + lse3 chain)

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

// barrier for LDS hand-offs only: __syncthreads() also drains the vector-memory counter, i.e. it would wait every step for
// the prefetched loads of four steps ahead and for the workspace store to be acknowledged by L2
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The alpha recursion (blockIdx.y == 0, forward in t) and the beta recursion (blockIdx.y == 1, backward in t) of one
// utterance do not depend on each other: they run as two workgroups side by side and leave alpha_t(s) / beta_t(s) in the
// workspace; forward_sum_grad_kernel combines them.  The sequential chain is T steps instead of 2 T.
__global__ __launch_bounds__(CTC_THREADS) void forward_sum_pass_kernel(int B, int Tf, int Tx, const float* __restrict__ logp,
                                                                       const float* __restrict__ prior,
                                                                       const int32_t* __restrict__ text_lens,
                                                                       const int32_t* __restrict__ feat_lens, float log_blank,
                                                                       float* __restrict__ ws, float* __restrict__ loss_b) {
  extern __shared__ float sh[];  // 2 columns of S_pad floats (+ 2 guard cells each side)
  const int b = blockIdx.x, tid = threadIdx.x;
  const bool beta = blockIdx.y == 1;
  int N = text_lens[b], Tn = feat_lens[b];
  if (N > Tx) N = Tx;
  if (Tn > Tf) Tn = Tf;
  const int S = 2 * N + 1;
  const int Spad = 2 * Tx + 1;
  const float NINF = -__builtin_huge_valf();
  const float* lpb = logp + (int64_t)b * Tf * Tx;
  const float* prb = prior + (int64_t)b * Tf * Tx;
  float* w = ws + ((int64_t)b * 2 + (beta ? 1 : 0)) * Tf * Spad;       // alpha_t(s) or beta_t(s), row t
  float* nll_slot = ws + (int64_t)B * 2 * Tf * Spad + b;
  if (N <= 0 || Tn <= 0) {
    if (tid == 0 && !beta) { loss_b[b] = 0.f; *nll_slot = __builtin_huge_valf(); }
    return;
  }
  float* col0 = sh + 2;
  float* col1 = sh + 2 + (Spad + 4);
  if (tid < 2) { sh[tid] = NINF; sh[(Spad + 4) + tid] = NINF; }
  for (int k = 0; k < CTC_KMAX; ++k) {       // guard cells above S: reads of s+1, s+2 beyond the end must see -inf
    const int s = tid + k * CTC_THREADS;
    if (s >= S && s < Spad + 2) { col0[s] = NINF; col1[s] = NINF; }
  }
  __syncthreads();

  // lp(t, s): log-prob of the symbol at extended position s
  auto lp = [&](int t, int s) -> float {
    if ((s & 1) == 0) return log_blank;
    const int j = (s - 1) >> 1;
    return lpb[(int64_t)t * Tx + j] + prb[(int64_t)t * Tx + j];
  };
  // one step of either recursion at position s from the previous column
  auto step = [&](const float* prev, int s, float l) -> float {
    if (!beta) {
      const float a1 = prev[s], a2 = prev[s - 1];
      const float a3 = ((s & 1) && s >= 3) ? prev[s - 2] : NINF;
      return lse3(a1, a2, a3) + l;
    }
    const float b1 = prev[s], b2 = prev[s + 1];
    const float b3 = ((s & 1) && s + 2 < S) ? prev[s + 2] : NINF;
    return lse3(b1, b2, b3) + l;
  };
  const int t_first = beta ? Tn - 1 : 0, dt = beta ? -1 : 1;       // row t_first + dt * n is the n-th row of the pass
  for (int k = 0; k < CTC_KMAX; ++k) {
    const int s = tid + k * CTC_THREADS;
    if (s < S) {
      const float v = (beta ? (s >= S - 2) : (s < 2)) ? lp(t_first, s) : NINF;
      col0[s] = v;
      w[(int64_t)t_first * Spad + s] = v;
    }
  }
  __syncthreads();
  float* prev = col0;
  float* cur = col1;
  if (S <= CTC_THREADS) {
    // one extended position per thread (the usual case): the two global loads behind lp(t, s) are issued CTC_PD steps ahead,
    // so a step is an LDS round trip + one lse3, not a trip to L2 / HBM
    const int s = tid;
    const bool act = s < S;
    float lq[CTC_PD];
#pragma unroll
    for (int q = 0; q < CTC_PD; ++q) lq[q] = (act && 1 + q < Tn) ? lp(t_first + dt * (1 + q), s) : 0.f;
    for (int n = 1; n < Tn; n += CTC_PD) {
#pragma unroll
      for (int q = 0; q < CTC_PD; ++q) {
        const int nn = n + q;
        if (nn < Tn) {                          // workgroup-uniform (no `break`: the loop must unroll, lq[] lives in registers)
          const float l = lq[q];
          if (act && nn + CTC_PD < Tn) lq[q] = lp(t_first + dt * (nn + CTC_PD), s);
          if (act) {
            const float v = step(prev, s, l);
            cur[s] = v;
            w[(int64_t)(t_first + dt * nn) * Spad + s] = v;
          }
          lds_barrier();
          float* tmp = prev; prev = cur; cur = tmp;
        }
      }
    }
  } else {
    for (int n = 1; n < Tn; ++n) {
      const int t = t_first + dt * n;
      for (int k = 0; k < CTC_KMAX; ++k) {
        const int s = tid + k * CTC_THREADS;
        if (s < S) {
          const float v = step(prev, s, lp(t, s));
          cur[s] = v;
          w[(int64_t)t * Spad + s] = v;
        }
      }
      __syncthreads();
      float* tmp = prev; prev = cur; cur = tmp;
    }
  }
  if (!beta && tid == 0) {
    const float nll = -lse2(prev[S - 1], S >= 2 ? prev[S - 2] : NINF);
    const bool inf_loss = !(nll < __builtin_huge_valf());  // zero_infinity=True
    loss_b[b] = inf_loss ? 0.f : nll / (float)(N < 1 ? 1 : N);
    *nll_slot = inf_loss ? __builtin_huge_valf() : nll;
  }
}

// d/d lp[t,c] = exp(lp[t,c]) - exp(alpha_t(s) + beta_t(s) + nll - lp[t,c]), s = 2c + 1, scaled by 1 / (N * B) (reduction='mean'
// over the single-item batch divides by the target length); 0 outside the utterance and for an infinite loss
__global__ void forward_sum_grad_kernel(int B, int Tf, int Tx, const float* __restrict__ logp, const float* __restrict__ prior,
                                        const int32_t* __restrict__ text_lens, const int32_t* __restrict__ feat_lens,
                                        const float* __restrict__ ws, float* __restrict__ grad) {
  const int64_t n = (int64_t)B * Tf * Tx;
  const int Spad = 2 * Tx + 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int j = (int)(i % Tx);
    const int64_t r = i / Tx;
    const int t = (int)(r % Tf), b = (int)(r / Tf);
    int N = text_lens[b], Tn = feat_lens[b];
    if (N > Tx) N = Tx;
    if (Tn > Tf) Tn = Tf;
    const float nll = ws[(int64_t)B * 2 * Tf * Spad + b];
    float g = 0.f;
    if (j < N && t < Tn && nll < __builtin_huge_valf()) {
      const int s = 2 * j + 1;
      const float l = logp[i] + prior[i];
      const float a = ws[((int64_t)b * 2 * Tf + t) * Spad + s];
      const float v = ws[(((int64_t)b * 2 + 1) * Tf + t) * Spad + s];
      g = (expf(l) - expf(a + v + nll - l)) * (1.f / ((float)(N < 1 ? 1 : N) * (float)B));
    }
    grad[i] = g;
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
  return ((int64_t)B * 2 * Tf * (2 * Tx + 1) + B) * 4;      // alpha and beta tables + the per-utterance nll
}

// loss_b: (B,) per-utterance nll/N (sum/B is the reference loss); grad: (B,Tf,Tx) d(sum_b loss_b / B)/d(log_p_attn)
extern "C" int s2svc_forward_sum(int B, int Tf, int Tx, const float* log_p_attn, const float* prior,
                                 const int32_t* text_lens, const int32_t* feat_lens, float log_blank, void* ws,
                                 float* loss_b, float* grad, void* stream) {
  S2S_REQUIRE(B >= 0 && Tf > 0 && Tx > 0, "forward_sum: bad shape");
  S2S_REQUIRE(2 * Tx + 1 <= CTC_THREADS * CTC_KMAX, "forward_sum: T_text too large");
  if (B == 0) return 0;
  const size_t shm = (size_t)2 * (2 * Tx + 1 + 4) * sizeof(float);
  hipLaunchKernelGGL(forward_sum_pass_kernel, dim3(B, 2), dim3(CTC_THREADS), shm, (hipStream_t)stream, B, Tf, Tx, log_p_attn, prior,
                     text_lens, feat_lens, log_blank, (float*)ws, loss_b);
  S2S_CHECK_LAUNCH("forward_sum_pass_kernel");
  const int64_t n = (int64_t)B * Tf * Tx;
  int nb = (int)((n + 255) / 256);
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(forward_sum_grad_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, B, Tf, Tx, log_p_attn, prior, text_lens,
                     feat_lens, (const float*)ws, grad);
  S2S_CHECK_LAUNCH("forward_sum_grad_kernel");
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
