// One layer of the VITS DilatedDepthSeparableConv stack (the body of every ConvFlow and of the two conditioning stacks of the
// stochastic duration predictor) as ONE launch per direction, fp32:
//
//   forward   y1 = dwconv_dilated(x) ; y2 = gelu(LN1(y1)) ; y3 = y2 . W^T + b ; out = mask * (x + dropout(gelu(LN2(y3))))
//   backward  (g = d out)  dres = mask * g ; du2 = dres * dropmask * gelu'(.) ; dy3 = LN2'(du2) ; dy2 = dy3 . W ;
//             du1 = dy2 * gelu'(.) ; dy1 = LN1'(du1)            (the depthwise convolution's data / weight gradients, the 1x1
//             weight gradient and the LayerNorm gamma / beta reductions stay with the kernels that had them: they need whole columns)
//
// reference: modules/vits/flow.py:110-190 (DilatedDepthSeparableConv.forward: per layer conv -> LN -> GELU -> 1x1 -> LN -> GELU ->
// dropout, x = x + y, masked), called from flow.py:250-310 (ConvFlow) and modules/duration_predictor.py:211-304.
//
// Why: the tensors are (B * T_text <= 64, 384 channels) -- 1024 rows x 384 in the vc2 recipe -- so each of the four kernels of a
// layer (depthwise conv, LayerNorm + GELU, 1x1 GEMM, LayerNorm + GELU + dropout + residual) is a launch-floor-sized launch, and a
// step runs 30 such layers forward and backward on the auxiliary stream: ~450 of the 1 080 launches of an AAS-VC step, 4.4 ms of
// dependent launches beside the decoder.  Here a workgroup (4 waves) owns 16 consecutive rows: the depthwise taps are
// three coalesced row reads, a LayerNorm is a wave-level reduction over a row held in registers (a lane owns channels lane + 64 i),
// the 1x1 convolution is a 16 x C x C product on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) with the activation rows in LDS
// (row pitch C + 4 floats: conflict-free 16-byte fragment reads) and the weight rows streamed from L2 straight into B fragments
// (a lane's float4 = 4 consecutive k of one output column, the same k positions in the A fragment -- 4 MFMAs per load pair).
// Every intermediate the backward pass and the weight-gradient kernels need is written once (y1, y2, y3, the row statistics).
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float dds_f32x4;

__device__ __forceinline__ float dds_gelu(float u) { return 0.5f * u * (1.f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float dds_gelu_d(float u) {
  return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * expf(-0.5f * u * u);
}

// acc[j] (16 rows x 16 columns n0 + 16 j .. + 15) = sum_k As[m][k] * W[n][k]; As: LDS, row pitch C + 4; W: global, row pitch C
template <int C>
__device__ __forceinline__ void dds_rows16_gemm(const float* As, const float* __restrict__ W, int n0, int lane, dds_f32x4 (&acc)[C / 64]) {
  constexpr int NT = C / 64, KS = C / 16, PITCH = C + 4;
  const int lr = lane & 15, lg = lane >> 4;
  const float* wp = W + (int64_t)(n0 + lr) * C + 4 * lg;
  const float* ap = As + lr * PITCH + 4 * lg;
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (dds_f32x4){0.f, 0.f, 0.f, 0.f};
  float4 bn[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bn[j] = *reinterpret_cast<const float4*>(wp + (int64_t)j * 16 * C);
#pragma unroll 2
  for (int ks = 0; ks < KS; ++ks) {
    float4 bc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bc[j] = bn[j];
    if (ks + 1 < KS) {
#pragma unroll
      for (int j = 0; j < NT; ++j) bn[j] = *reinterpret_cast<const float4*>(wp + (int64_t)j * 16 * C + (ks + 1) * 16);
    }
    const float4 a = *reinterpret_cast<const float4*>(ap + ks * 16);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bc[j].x, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bc[j].y, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bc[j].z, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bc[j].w, acc[j], 0, 0, 0);
    }
  }
}

// the accumulators of a wave (rows lg * 4 + i, column n0 + 16 j + lr) + bias -> the LDS row block
template <int C>
__device__ __forceinline__ void dds_acc_to_lds(const dds_f32x4 (&acc)[C / 64], const float* __restrict__ bias, int n0, int lane, float* Ys) {
  constexpr int NT = C / 64, PITCH = C + 4;
  const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j * 16 + lr;
    const float b = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) Ys[(lg * 4 + i) * PITCH + n] = acc[j][i] + b;
  }
}

struct dds_fwd_args {
  int rows, Tn, dil;
  const float* x;
  const int32_t* lens;
  const float *dw_w, *dw_b, *g1, *b1, *W, *bias, *g2, *b2;
  float eps, drop_p;
  const uint64_t* seed_base;
  uint64_t seed_off;
  float *y1, *mean1, *rstd1, *y2, *y3, *mean2, *rstd2, *out;
};

template <int C>
__global__ __launch_bounds__(256) void dds_layer_fwd_kernel(const dds_fwd_args a) {
  constexpr int NV = C / 64, PITCH = C + 4;
  __shared__ __attribute__((aligned(16))) float As[16 * PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 16;
  // ---- phase A: depthwise dilated convolution (3 taps) -> LayerNorm 1 -> GELU, a wave per row
#pragma unroll 1
  for (int rr = 0; rr < 4; ++rr) {
    const int lrow = wave * 4 + rr, r = r0 + lrow;
    if (r >= a.rows) {                                   // (uniform) rows past the batch: defined zeros for the product
#pragma unroll
      for (int i = 0; i < NV; ++i) As[lrow * PITCH + lane + 64 * i] = 0.f;
      continue;
    }
    const int t = r % a.Tn;
    const int64_t base = (int64_t)r * C;
    float v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      float acc = a.dw_b ? a.dw_b[c] : 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int tt = t + (j - 1) * a.dil;
        if (tt >= 0 && tt < a.Tn) acc += a.dw_w[c * 3 + j] * a.x[base + (int64_t)(tt - t) * C + c];
      }
      v[i] = acc;
      a.y1[base + c] = acc;
      sum += acc;
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const float d = v[i] - mean; sq += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)C + a.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      const float o = dds_gelu((v[i] - mean) * rstd * a.g1[c] + a.b1[c]);
      a.y2[base + c] = o;
      As[lrow * PITCH + c] = o;
    }
    if (lane == 0) { a.mean1[r] = mean; a.rstd1[r] = rstd; }
  }
  __syncthreads();
  // ---- phase B: y3 = y2 . W^T + b   (16 x C x C on the exact-fp32 MFMA; a wave owns C / 4 output columns)
  dds_f32x4 acc[NV];
  dds_rows16_gemm<C>(As, a.W, wave * (C / 4), lane, acc);
  __syncthreads();                                       // every wave is past its A reads: the block takes y3
  dds_acc_to_lds<C>(acc, a.bias, wave * (C / 4), lane, As);
  __syncthreads();
  // ---- phase C: LayerNorm 2 -> GELU -> dropout -> + x -> mask
  const uint64_t seed = (a.seed_base ? *a.seed_base : 0ull) + a.seed_off;
  const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
#pragma unroll 1
  for (int rr = 0; rr < 4; ++rr) {
    const int lrow = wave * 4 + rr, r = r0 + lrow;
    if (r >= a.rows) continue;
    const int64_t base = (int64_t)r * C;
    float v[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i] = As[lrow * PITCH + lane + 64 * i];
      a.y3[base + lane + 64 * i] = v[i];
      sum += v[i];
    }
    const float mean = wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) { const float d = v[i] - mean; sq += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)C + a.eps);
    const bool ok = !a.lens || (r % a.Tn) < a.lens[r / a.Tn];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      float o = dds_gelu((v[i] - mean) * rstd * a.g2[c] + a.b2[c]);
      if (a.drop_p > 0.f) o *= dropout_scale(seed, (uint64_t)(base + c), a.drop_p, inv_keep);
      o += a.x[base + c];
      a.out[base + c] = ok ? o : 0.f;
    }
    if (lane == 0) { a.mean2[r] = mean; a.rstd2[r] = rstd; }
  }
}

struct dds_bwd_args {
  int rows, Tn;
  const float* g;
  const int32_t* lens;
  const float *y3, *mean2, *rstd2, *g2, *b2;
  float drop_p;
  const uint64_t* seed_base;
  uint64_t seed_off;
  const float* Wt;                                        // W^T: [in][out] row-major (reduction over the output channels contiguous)
  const float *y1, *mean1, *rstd1, *g1, *b1;
  float *dres, *du2, *dy3, *du1, *dy1;
};

template <int C>
__global__ __launch_bounds__(256) void dds_layer_bwd_kernel(const dds_bwd_args a) {
  constexpr int NV = C / 64, PITCH = C + 4;
  __shared__ __attribute__((aligned(16))) float As[16 * PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 16;
  const uint64_t seed = (a.seed_base ? *a.seed_base : 0ull) + a.seed_off;
  const float inv_keep = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
  // ---- phase A: residual gradient, dropout mask, GELU', LayerNorm 2 backward -> dy3
#pragma unroll 1
  for (int rr = 0; rr < 4; ++rr) {
    const int lrow = wave * 4 + rr, r = r0 + lrow;
    if (r >= a.rows) {
#pragma unroll
      for (int i = 0; i < NV; ++i) As[lrow * PITCH + lane + 64 * i] = 0.f;
      continue;
    }
    const int64_t base = (int64_t)r * C;
    const bool ok = !a.lens || (r % a.Tn) < a.lens[r / a.Tn];
    const float mu = a.mean2[r], rs = a.rstd2[r];
    float gg[NV], xh[NV];
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      const float d = ok ? a.g[base + c] : 0.f;
      a.dres[base + c] = d;
      xh[i] = (a.y3[base + c] - mu) * rs;
      const float u = xh[i] * a.g2[c] + a.b2[c];
      float du = d * dds_gelu_d(u);
      if (a.drop_p > 0.f) du *= dropout_scale(seed, (uint64_t)(base + c), a.drop_p, inv_keep);
      a.du2[base + c] = du;
      gg[i] = du * a.g2[c];
      sa += gg[i];
      sb += gg[i] * xh[i];
    }
    sa = wave_sum(sa) / (float)C;
    sb = wave_sum(sb) / (float)C;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      const float o = rs * (gg[i] - sa - xh[i] * sb);
      a.dy3[base + c] = o;
      As[lrow * PITCH + c] = o;
    }
  }
  __syncthreads();
  // ---- phase B: dy2 = dy3 . W   (the product of the forward pass with W^T in place of W, no bias)
  dds_f32x4 acc[NV];
  dds_rows16_gemm<C>(As, a.Wt, wave * (C / 4), lane, acc);
  __syncthreads();
  dds_acc_to_lds<C>(acc, nullptr, wave * (C / 4), lane, As);
  __syncthreads();
  // ---- phase C: GELU', LayerNorm 1 backward -> dy1 (the gradient at the depthwise convolution's output)
#pragma unroll 1
  for (int rr = 0; rr < 4; ++rr) {
    const int lrow = wave * 4 + rr, r = r0 + lrow;
    if (r >= a.rows) continue;
    const int64_t base = (int64_t)r * C;
    const float mu = a.mean1[r], rs = a.rstd1[r];
    float gg[NV], xh[NV];
    float sa = 0.f, sb = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane + 64 * i;
      xh[i] = (a.y1[base + c] - mu) * rs;
      const float u = xh[i] * a.g1[c] + a.b1[c];
      const float du = As[lrow * PITCH + c] * dds_gelu_d(u);
      a.du1[base + c] = du;
      gg[i] = du * a.g1[c];
      sa += gg[i];
      sb += gg[i] * xh[i];
    }
    sa = wave_sum(sa) / (float)C;
    sb = wave_sum(sb) / (float)C;
#pragma unroll
    for (int i = 0; i < NV; ++i) a.dy1[base + lane + 64 * i] = rs * (gg[i] - sa - xh[i] * sb);
  }
}

bool al16(const void* p) { return ((uintptr_t)p) % 16 == 0; }

}  // namespace

extern "C" int s2svc_dds_layer_supported(int C, int ks) { return (ks == 3 && (C == 192 || C == 256 || C == 384 || C == 512)) ? 1 : 0; }

extern "C" int s2svc_dds_layer_fwd(int B, int Tn, int C, int ks, int dil, const float* x, const int32_t* lens, const float* dw_w,
                                   const float* dw_b, const float* g1, const float* b1, const float* W, const float* bias,
                                   const float* g2, const float* b2, float eps, float drop_p, const uint64_t* seed_base,
                                   uint64_t seed_off, float* y1, float* mean1, float* rstd1, float* y2, float* y3, float* mean2,
                                   float* rstd2, float* out, void* stream) {
  S2S_REQUIRE(s2svc_dds_layer_supported(C, ks), "dds_layer_fwd: C in {192, 256, 384, 512}, kernel size 3");
  S2S_REQUIRE(B >= 0 && Tn > 0 && dil >= 1 && x && dw_w && g1 && b1 && W && g2 && b2 && y1 && mean1 && rstd1 && y2 && y3 && mean2 && rstd2 && out,
              "dds_layer_fwd: missing operand");
  S2S_REQUIRE(al16(W), "dds_layer_fwd: the 1x1 weight must be 16-byte aligned");
  S2S_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "dds_layer_fwd: dropout probability in [0, 1)");
  const int rows = B * Tn;
  if (rows == 0) return 0;
  dds_fwd_args a = {rows, Tn, dil, x, lens, dw_w, dw_b, g1, b1, W, bias, g2, b2, eps, drop_p, seed_base, seed_off,
                    y1, mean1, rstd1, y2, y3, mean2, rstd2, out};
  const dim3 grid((rows + 15) / 16), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (C) {
    case 192: hipLaunchKernelGGL(dds_layer_fwd_kernel<192>, grid, block, 0, st, a); break;
    case 256: hipLaunchKernelGGL(dds_layer_fwd_kernel<256>, grid, block, 0, st, a); break;
    case 384: hipLaunchKernelGGL(dds_layer_fwd_kernel<384>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(dds_layer_fwd_kernel<512>, grid, block, 0, st, a); break;
  }
  S2S_CHECK_LAUNCH("dds_layer_fwd_kernel");
  return 0;
}

extern "C" int s2svc_dds_layer_bwd(int B, int Tn, int C, const float* g, const int32_t* lens, const float* y3, const float* mean2,
                                   const float* rstd2, const float* g2, const float* b2, float drop_p, const uint64_t* seed_base,
                                   uint64_t seed_off, const float* Wt, const float* y1, const float* mean1, const float* rstd1,
                                   const float* g1, const float* b1, float* dres, float* du2, float* dy3, float* du1, float* dy1,
                                   void* stream) {
  S2S_REQUIRE(s2svc_dds_layer_supported(C, 3), "dds_layer_bwd: C in {192, 256, 384, 512}");
  S2S_REQUIRE(B >= 0 && Tn > 0 && g && y3 && mean2 && rstd2 && g2 && b2 && Wt && y1 && mean1 && rstd1 && g1 && b1 && dres && du2 && dy3 && du1 && dy1,
              "dds_layer_bwd: missing operand");
  S2S_REQUIRE(al16(Wt), "dds_layer_bwd: the transposed 1x1 weight must be 16-byte aligned");
  const int rows = B * Tn;
  if (rows == 0) return 0;
  dds_bwd_args a = {rows, Tn, g, lens, y3, mean2, rstd2, g2, b2, drop_p, seed_base, seed_off, Wt, y1, mean1, rstd1, g1, b1,
                    dres, du2, dy3, du1, dy1};
  const dim3 grid((rows + 15) / 16), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (C) {
    case 192: hipLaunchKernelGGL(dds_layer_bwd_kernel<192>, grid, block, 0, st, a); break;
    case 256: hipLaunchKernelGGL(dds_layer_bwd_kernel<256>, grid, block, 0, st, a); break;
    case 384: hipLaunchKernelGGL(dds_layer_bwd_kernel<384>, grid, block, 0, st, a); break;
    default: hipLaunchKernelGGL(dds_layer_bwd_kernel<512>, grid, block, 0, st, a); break;
  }
  S2S_CHECK_LAUNCH("dds_layer_bwd_kernel");
  return 0;
}
