// Autoregressive decode step, round 6: the tail of a step as ONE launch.
//
// reference: models/vtn.py:344-389 (generation loop: the stop test and the step counter), embedding.py:115-125 (positional encoding of
// the next position).
//
// A dependent launch of the captured step costs ~1.9 us before it does anything and every dependent trip to memory ~1 us more: the
// one-thread `decode_advance` of rounds 1-5 took 4.3 us.  decode_emit_advance_kernel = decode_emit (frames / stop probabilities / stop
// test of this position) on ONE packed feat_out | prob_out projection + the step counter and dropout seed advanced by the last workgroup
// to finish + the positional-encoding row of the NEXT position written as a (1, D) residual row for the input Linear of the next step
// (its GEMM epilogue adds it with row stride 0: the `decode_posenc` launch is gone).  56 -> 53 launches per VTN vc1 step.
//
// What round 6 also built, measured and REMOVED (profiles/AB_LOG.md "decode step"): the attention kernel carrying its head's share of the
// output projection (all operands staged by LDS-DMA in one trip, 8.2-9.2 us) with the H fp32 partials + residual + bias summed as the
// input stage of the next LayerNorm + projection kernel (8.7 us against 5.4 without) -- 17.4 us per attention sublayer against 17.2 for
// attention + out-projection + LayerNorm-projection as three launches: both fusions REPLICATE traffic (every (utterance, head) workgroup
// re-reads its weight slice, every consumer workgroup re-reads all partials), which costs what the launch saved; and a one-launch prenet
// (a workgroup per utterance streaming 369 KB of weights: 22-27 us against 19 for its four launches).
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void decode_emit_advance_kernel(int r, int odim, const T* __restrict__ feat, const T* __restrict__ logit,
                                                                  int64_t ldf_, float threshold, const int32_t* __restrict__ minlen,
                                                                  const int32_t* __restrict__ maxlen, int32_t* __restrict__ pos,
                                                                  float* __restrict__ outs, int64_t outs_bs, float* __restrict__ probs,
                                                                  int64_t probs_bs, T* __restrict__ prev, int32_t* __restrict__ stop_at,
                                                                  uint64_t* __restrict__ seed_base, uint64_t seed_stride,
                                                                  uint32_t* __restrict__ ticket, const float* __restrict__ pe,
                                                                  const float* __restrict__ alpha, int D, int pe_rows, T* __restrict__ pe_next) {
  const int b = blockIdx.x;
  const int p = *pos;
  for (int i = threadIdx.x; i < r * odim; i += 256) {
    const float v = ldf(feat + (int64_t)b * ldf_ + i);
    outs[(int64_t)b * outs_bs + (int64_t)p * r * odim + i] = v;
    if (i >= (r - 1) * odim) stf(prev + (int64_t)b * odim + (i - (r - 1) * odim), v);
  }
  if (pe_next && p + 1 < pe_rows) {                  // alpha * pe[p + 1]: the residual row of the next step's input Linear
    const float al = alpha ? *alpha : 1.f;
    for (int d = b * 256 + threadIdx.x; d < D; d += gridDim.x * 256) stf(pe_next + d, al * pe[(int64_t)(p + 1) * D + d]);
  }
  if (threadIdx.x == 0) {
    bool fire = false;
    for (int i = 0; i < r; ++i) {
      const float pr = 1.f / (1.f + expf(-ldf(logit + (int64_t)b * ldf_ + i)));
      probs[(int64_t)b * probs_bs + (int64_t)p * r + i] = pr;
      fire = fire || (pr >= threshold);
    }
    if ((fire || p + 1 >= maxlen[b]) && p + 1 >= minlen[b] && stop_at[b] == 0) stop_at[b] = p + 1;
  }
  __syncthreads();                                   // every thread of this workgroup has read *pos
  if (threadIdx.x == 0) {
    const uint32_t drawn = atomicAdd(ticket, 1u);    // (integer atomic: no order of summation to keep)
    if (drawn == gridDim.x - 1) {                    // every workgroup read *pos before it drew its ticket
      *ticket = 0u;
      *pos = p + 1;
      if (seed_base) *seed_base += seed_stride;
    }
  }
}

}  // namespace

// s2svc_decode_emit + s2svc_decode_advance (+ the next position's positional-encoding row) in one launch: feat / logit rows have stride
// ldf (columns of one packed projection); `ticket`: one zero-initialised device uint32 owned by the caller (zero again when the kernel
// ends); pe_next != NULL: pe_next[0 .. D) = alpha * pe[(*pos + 1) * D + .] in `dtype` (pe: (pe_rows, D) fp32, alpha: device scalar or NULL).
extern "C" int s2svc_decode_emit_advance(int dtype, int B, int r, int odim, const void* feat, const void* logit, int64_t ldf_, float threshold,
                                         const int32_t* minlen, const int32_t* maxlen, int32_t* pos, float* outs, int64_t outs_bs, float* probs,
                                         int64_t probs_bs, void* prev, int32_t* stop_at, uint64_t* seed_base, uint64_t seed_stride,
                                         uint32_t* ticket, const float* pe, const float* alpha, int D, int pe_rows, void* pe_next,
                                         void* stream) {
  S2S_REQUIRE(B > 0 && r > 0 && odim > 0 && feat && logit && pos && outs && probs && prev && stop_at && minlen && maxlen && ticket,
              "decode_emit_advance: bad arguments");
  S2S_REQUIRE(!pe_next || (pe && D > 0 && pe_rows > 0), "decode_emit_advance: pe_next needs the table");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(decode_emit_advance_kernel<float>, dim3(B), dim3(256), 0, st, r, odim, (const float*)feat, (const float*)logit, ldf_, threshold,
                       minlen, maxlen, pos, outs, outs_bs, probs, probs_bs, (float*)prev, stop_at, seed_base, seed_stride, ticket, pe, alpha, D,
                       pe_rows, (float*)pe_next);
  else
    hipLaunchKernelGGL(decode_emit_advance_kernel<bf16_t>, dim3(B), dim3(256), 0, st, r, odim, (const bf16_t*)feat, (const bf16_t*)logit, ldf_,
                       threshold, minlen, maxlen, pos, outs, outs_bs, probs, probs_bs, (bf16_t*)prev, stop_at, seed_base, seed_stride, ticket, pe,
                       alpha, D, pe_rows, (bf16_t*)pe_next);
  S2S_CHECK_LAUNCH("decode_emit_advance_kernel");
  return 0;
}
