// Monotonic alignment search (AAS-VC / Glow-TTS Viterbi) on the GPU: one wavefront per utterance.
//
// reference: seq2seq_vc/modules/alignments.py:63-93 (_monotonic_alignment_search, numba-JIT on the
// host) and :281-310 (viterbi_decode: per-utterance D2H copy -> MAS -> bincount -> H2D).  This kernel
// removes the host round trip: log_p_attn stays in HBM, the DP runs in fp64 registers with the
// neighbour value Q[i-1, j-1] fetched by a wave shuffle, one decision bit per cell is kept in LDS,
// and the backtrack, the duration bincount and the binarisation-loss gather happen in the same launch.
//
// Semantics kept bit-for-bit:  lane = text index i.  Q[0,j] = prefix sum of log_p[0..j, 0];
// Q[i,j] = max(Q[i-1,j-1], Q[i,j-1]) + log_p[j,i] for 1 <= i <= min(j, T_inp-1); all other cells
// stay -inf.  Backtrack from A[T_mel-1] = T_inp-1 with `Q[i-1,j] >= Q[i,j]` preferring i-1
// (-inf >= -inf is true), i == 0 stays 0.  The decision the backtrack needs at column j is exactly
// the comparison the forward step makes when it builds column j+1, so it is recorded there.
// Deviation (documented in DESIGN.md section 5): row 0 is an fp64 running prefix sum, O(T) instead of the
// reference's O(T^2) re-summation of an fp32 slice.
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

// q of the lane below, for the dependent chain of the search: two 32-bit DPP moves instead of two ds_bpermute (~10 instead of
// ~100 cycles per column of the recursion).  Lane 0 keeps its own value (the caller overwrites it).  Like the reductions of common.h
// this is a DPP move: every lane of the wave must be active where it is called (the kernel is one full wave under uniform control flow).
__device__ __forceinline__ double wave_shr1_d(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int slo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);      // wave_shr:1
  const int shi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(shi, slo);
}


template <int S, bool DEC_LDS>  // S >= ceil(T_inp / 64) text slots per lane
__global__ __launch_bounds__(64) void mas_kernel(int B, int Tf, int Tx, const float* __restrict__ logp,
                                                 const int32_t* __restrict__ text_lens,
                                                 const int32_t* __restrict__ feat_lens, int32_t* __restrict__ path,
                                                 float* __restrict__ ds, float* __restrict__ binmean,
                                                 uint64_t* __restrict__ dec_global) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  int T_inp = text_lens[b], T_mel = feat_lens[b];
  if (T_inp > Tx) T_inp = Tx;
  if (T_mel > Tf) T_mel = Tf;
  const float* lp = logp + (int64_t)b * Tf * Tx;
  int32_t* pth = path + (int64_t)b * Tf;
  float* dsb = ds + (int64_t)b * Tx;

  for (int i = lane; i < Tx; i += 64) dsb[i] = 0.f;
  for (int j = lane; j < Tf; j += 64) pth[j] = -1;
  if (T_inp <= 0 || T_mel <= 0) {
    if (lane == 0) binmean[b] = 0.f;
    return;
  }
  // decision words: dec[j*S + s] bit l  <=>  Q[i-1, j] >= Q[i, j] for i = s*64 + l
  uint64_t* dec_l = reinterpret_cast<uint64_t*>(smem);
  int32_t* pth_l = reinterpret_cast<int32_t*>(smem + (size_t)Tf * S * 8);
  uint64_t* dec_g = dec_global + (int64_t)b * Tf * S;

  const double NINF = -__builtin_huge_val();
  double q[S];
#pragma unroll
  for (int s = 0; s < S; ++s) q[s] = NINF;
  if (lane == 0) q[0] = (double)lp[0];

  // The log-probs of U columns are fetched while the U columns before them run through the dependent chain (a group's loads used
  // to be issued only after the previous group's last step: one exposed trip to L2 / HBM per U = 4 columns, ~0.4 us per column).
  constexpr int U = 8;
  float nxt[U][S];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int i = s * 64 + lane, j = j0 + u;
        nxt[u][s] = (j < T_mel && i < T_inp) ? lp[(int64_t)j * Tx + i] : 0.f;
      }
  };
  fetch(1);
  for (int j0 = 1; j0 < T_mel; j0 += U) {
    float val[U][S];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int s = 0; s < S; ++s) val[u][s] = nxt[u][s];
    if (j0 + U < T_mel) fetch(j0 + U);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 + u;
      if (j < T_mel) {
        double up[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
          double t = wave_shr1_d(q[s]);                      // lane l <- lane l - 1 (DPP wave_shr:1: VALU moves, no LDS crossbar)
          if (lane == 0) t = NINF;
          if (s > 0) {
            const double carry = __shfl(q[s - 1], 63, 64);
            if (lane == 0) t = carry;
          }
          up[s] = t;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
          const int i = s * 64 + lane;
          const bool ge = (i >= 1) && (up[s] >= q[s]);
          const uint64_t word = __ballot(ge);
          if (lane == 0) {
            if (DEC_LDS) dec_l[(j - 1) * S + s] = word; else dec_g[(int64_t)(j - 1) * S + s] = word;
          }
          if (i < T_inp) {
            const double v = (double)val[u][s];
            if (i == 0) q[s] = q[s] + v;                        // running prefix sum of row 0
            else if (i <= j) q[s] = (ge ? up[s] : q[s]) + v;
          }
        }
      }
    }
  }
  __syncthreads();

  if (DEC_LDS) {
    // backtrack: the sequential chain only walks the decision bits in LDS (one dependent LDS read per column); the path goes to
    // HBM and the binarisation-loss gather lp[j, path[j]] runs afterwards on all lanes -- with the gather inside the chain every
    // column paid a dependent global load (~0.4 us: 100 of the kernel's 133 us at T_mel = 256).  The sum's order changes
    // (per-lane fp64 partial sums + a butterfly instead of one fp64 chain); the path and the durations are bit-identical.
    if (lane == 0) {
      int a = T_inp - 1;
      pth_l[T_mel - 1] = a;
      for (int j = T_mel - 2; j >= 0; --j) {
        if (a != 0) {
          const uint64_t w = dec_l[j * S + (a >> 6)];
          if ((w >> (a & 63)) & 1ull) a = a - 1;
        }
        pth_l[j] = a;
      }
    }
    __syncthreads();
    double acc = 0.0;
    for (int j = lane; j < T_mel; j += 64) {
      const int a = pth_l[j];
      pth[j] = a;
      acc += (double)lp[(int64_t)j * Tx + a];
    }
    acc = wave_sum_d(acc);
    if (lane == 0) binmean[b] = (float)(acc / (double)T_mel);
  } else if (lane == 0) {
    int a = T_inp - 1;
    double acc = (double)lp[(int64_t)(T_mel - 1) * Tx + a];
    pth[T_mel - 1] = a;
    if (DEC_LDS) pth_l[T_mel - 1] = a;
    for (int j = T_mel - 2; j >= 0; --j) {
      if (a != 0) {
        const uint64_t w = DEC_LDS ? dec_l[j * S + (a >> 6)] : dec_g[(int64_t)j * S + (a >> 6)];
        if ((w >> (a & 63)) & 1ull) a = a - 1;
      }
      pth[j] = a;
      if (DEC_LDS) pth_l[j] = a;
      acc += (double)lp[(int64_t)j * Tx + a];
    }
    binmean[b] = (float)(acc / (double)T_mel);
  }
  __syncthreads();
  // durations = bincount(path)
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int i = s * 64 + lane;
    if (i < T_inp) {
      int cnt = 0;
      if (DEC_LDS) { for (int j = 0; j < T_mel; ++j) cnt += (pth_l[j] == i); }
      else { for (int j = 0; j < T_mel; ++j) cnt += (pth[j] == i); }
      dsb[i] = (float)cnt;
    }
  }
}

// d(bin_loss)/d(log_p_attn)[b, t, path[b,t]] = -gscale / (B * feat_len[b])
__global__ void mas_binloss_bwd_kernel(int B, int Tf, int Tx, const int32_t* __restrict__ path,
                                       const int32_t* __restrict__ feat_lens, const float* __restrict__ gout,
                                       float* __restrict__ dlogp) {
  const int64_t n = (int64_t)B * Tf;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / Tf);
    const int a = path[i];
    if (a >= 0) {
      int fl = feat_lens[b];
      if (fl > Tf) fl = Tf;
      dlogp[i * Tx + a] += -(*gout) / ((float)B * (float)fl);
    }
  }
}

}  // namespace

static inline int mas_slots(int Tx) {
  int need = (Tx + 63) / 64, S = 1;
  while (S < need) S <<= 1;
  return S;
}

extern "C" int64_t s2svc_mas_ws_bytes(int B, int Tf, int Tx) {
  return (int64_t)B * Tf * mas_slots(Tx) * 8;
}

extern "C" int s2svc_mas(int B, int Tf, int Tx, const float* log_p_attn, const int32_t* text_lens,
                         const int32_t* feat_lens, int32_t* path, float* ds, float* binmean, void* ws, void* stream) {
  S2S_REQUIRE(B >= 0 && Tf > 0 && Tx > 0, "mas: bad shape");
  S2S_REQUIRE(Tx <= 1024, "mas: T_text > 1024 not supported");
  if (B == 0) return 0;
  const int S = mas_slots(Tx);
  const size_t lds_need = (size_t)Tf * S * 8 + (size_t)Tf * 4;
  const bool in_lds = lds_need <= 60 * 1024;
  S2S_REQUIRE(in_lds || ws, "mas: workspace required for this size");
  hipStream_t st = (hipStream_t)stream;
  uint64_t* w = (uint64_t*)ws;
#define MAS_LAUNCH(SS)                                                                                              \
  do {                                                                                                              \
    if (in_lds)                                                                                                     \
      hipLaunchKernelGGL((mas_kernel<SS, true>), dim3(B), dim3(64), lds_need, st, B, Tf, Tx, log_p_attn, text_lens, \
                         feat_lens, path, ds, binmean, w);                                                          \
    else                                                                                                            \
      hipLaunchKernelGGL((mas_kernel<SS, false>), dim3(B), dim3(64), 0, st, B, Tf, Tx, log_p_attn, text_lens,       \
                         feat_lens, path, ds, binmean, w);                                                          \
  } while (0)
  switch (S) {
    case 1: MAS_LAUNCH(1); break;
    case 2: MAS_LAUNCH(2); break;
    case 4: MAS_LAUNCH(4); break;
    case 8: MAS_LAUNCH(8); break;
    default: MAS_LAUNCH(16); break;
  }
#undef MAS_LAUNCH
  S2S_CHECK_LAUNCH("mas_kernel");
  return 0;
}

extern "C" int s2svc_mas_binloss_bwd(int B, int Tf, int Tx, const int32_t* path, const int32_t* feat_lens,
                                     const float* gout, float* dlogp, void* stream) {
  if (B == 0) return 0;
  const int64_t n = (int64_t)B * Tf;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(mas_binloss_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, B, Tf, Tx, path, feat_lens,
                     gout, dlogp);
  S2S_CHECK_LAUNCH("mas_binloss_bwd_kernel");
  return 0;
}
