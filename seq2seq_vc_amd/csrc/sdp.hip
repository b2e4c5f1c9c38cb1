// Stochastic duration predictor (VITS flows) kernels: fused LayerNorm+activation(+dropout+residual+mask),
// rank-1 channel expansion, row masking, the rational-quadratic spline coupling (forward, inverse, backward)
// and the three fused "glue" stages of the variational NLL.
//
// reference: modules/duration_predictor.py:211-304 (StochasticDurationPredictor.forward),
//            modules/vits/flow.py:18-310 (Flip/Log/ElementwiseAffine/DilatedDepthSeparableConv/ConvFlow),
//            modules/vits/transform.py:17-216 (piecewise rational-quadratic transform, linear tails).
// Layout: channel-last rows r = (b, t); x_mask of the reference is (t < lens[b]).  Every tensor handed between
// stages is kept masked (zero rows past the utterance), which is value-identical to the reference's
// `x * x_mask` at every consumer and lets the mask multiply ride inside the producing kernel.
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

constexpr float LOG_2PI = 1.8378770664093453f;

__device__ __forceinline__ float act_deriv(float u, int act) {   // derivative at the pre-activation u
  switch (act) {
    case S2S_ACT_RELU: return u > 0.f ? 1.f : 0.f;
    case S2S_ACT_TANH: { const float t = tanhf(u); return 1.f - t * t; }
    case S2S_ACT_SWISH: { const float sg = 1.f / (1.f + expf(-u)); return sg * (1.f + u * (1.f - sg)); }
    case S2S_ACT_SIGMOID: { const float sg = 1.f / (1.f + expf(-u)); return sg * (1.f - sg); }
    case S2S_ACT_GELU: return 0.5f * (1.f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * expf(-0.5f * u * u);
    default: return 1.f;
  }
}

__device__ __forceinline__ bool row_valid(const int32_t* lens, int T, int64_t row) {
  return !lens || (int)(row % T) < lens[row / T];
}

// ------------------------------------------------------------------------------------------------
// y[r,:] = valid(r) ? x[r,:] : 0
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void mask_rows_kernel(int64_t n, int Tn, int C, const T* __restrict__ x, const int32_t* __restrict__ lens,
                                 T* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    stf(y + i, row_valid(lens, Tn, r) ? ldf(x + i) : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------
// Conv1d(1 -> C, k=1) of a scalar sequence, plus conditioning, masked:  y[r,c] = valid(r)*(a[r]*w[c] + b[c] + g[r,c])
// backward: dg = valid*dy (also the masked gradient the weight reductions read); da[r] = sum_c dg[r,c]*w[c]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void expand_fwd_kernel(int64_t n, int Tn, int C, const float* __restrict__ a, const float* __restrict__ w,
                                  const float* __restrict__ bias, const T* __restrict__ g, const int32_t* __restrict__ lens,
                                  T* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    float v = 0.f;
    if (row_valid(lens, Tn, r)) v = a[r] * w[c] + (bias ? bias[c] : 0.f) + (g ? ldf(g + i) : 0.f);
    stf(y + i, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void expand_bwd_kernel(int64_t rows, int Tn, int C, const T* __restrict__ dy,
                                                         const float* __restrict__ w, const int32_t* __restrict__ lens,
                                                         T* __restrict__ dg, float* __restrict__ da) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const bool ok = row_valid(lens, Tn, r);
  float s = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float d = ok ? ldf(dy + r * C + c) : 0.f;
    stf(dg + r * C + c, d);
    s += d * w[c];
  }
  s = wave_sum(s);
  if (lane == 0) da[r] = s;
}

// ------------------------------------------------------------------------------------------------
// y[r,:] = valid(r) * (res[r,:] + dropout(act(LayerNorm(x[r,:]))))      (res / lens optional; rows in registers)
// flow.py:148-190: each DDS layer is conv -> LN -> GELU -> 1x1 -> LN -> GELU -> dropout, then x + y, then * mask
// ------------------------------------------------------------------------------------------------
// ACT >= 0: the activation is a compile-time constant (GELU, NV = 4: the duration predictor's 192 / 256 channels -- the
// run-time switch expands tanh / erf / exp NV times: 11.9 -> ~3 KB of code for a kernel that runs ~120 times per AAS-VC step)
template <typename T, int NV, int ACT = -1>
__global__ __launch_bounds__(256) void ln_act_fwd_kernel(int rows, int D, int Tn, const T* __restrict__ x,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         int act_rt, const T* __restrict__ res, const int32_t* __restrict__ lens,
                                                         float p, const uint64_t* seed_base, uint64_t seed_off, T* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int act = ACT >= 0 ? ACT : act_rt;
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t base = (int64_t)row * D;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float v[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < D ? ldf(x + base + c) : 0.f;
    sum += v[i];
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float d = (lane + 64 * i) < D ? v[i] - mean : 0.f;
    sq += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
  const bool ok = row_valid(lens, Tn, row);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < D) {
      float o = act_apply((v[i] - mean) * rstd * gamma[c] + beta[c], act);
      if (p > 0.f) o *= dropout_scale(seed, (uint64_t)(base + c), p, inv_keep);
      if (res) o += ldf(res + base + c);
      stf(y + base + c, ok ? o : 0.f);
    }
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// The first half of a DDS layer in one launch: u = depthwise Conv1d(x) (kernel ks, dilation dil, zero padding inside the
// utterance's T frames; written out: the backward pass normalises it again), y = act(LayerNorm(u)).  A wave owns a frame: its
// lanes read the ks frames t + (j - pad) dil of their channels (coalesced rows), accumulate bias + sum_j w[c][j] x in tap order
// (the arithmetic of dwconv_fwd_vec_kernel), then the row statistics exactly as ln_act_fwd_kernel.
template <typename T, int NV, int ACT = -1>
__global__ __launch_bounds__(256) void dw_ln_act_fwd_kernel(int rows, int D, int Tn, int ks, int dil, const T* __restrict__ x,
                                                            const float* __restrict__ dw_w, const float* __restrict__ dw_b,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            int act_rt, T* __restrict__ u_out, T* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int act = ACT >= 0 ? ACT : act_rt;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int b = row / Tn, t = row - b * Tn;
  const int pad = (ks - 1) / 2;
  const int64_t base = (int64_t)row * D, bbase = (int64_t)b * Tn * D;
  float v[NV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = 0.f;
    if (c < D) {
      float a = dw_b ? dw_b[c] : 0.f;
      for (int j = 0; j < ks; ++j) {
        const int tt = t + (j - pad) * dil;
        if (tt >= 0 && tt < Tn) a += dw_w[c * ks + j] * ldf(x + bbase + (int64_t)tt * D + c);
      }
      stf(u_out + base + c, a);
      v[i] = a;
    }
    sum += v[i];
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float d = (lane + 64 * i) < D ? v[i] - mean : 0.f;
    sq += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < D) stf(y + base + c, act_apply((v[i] - mean) * rstd * gamma[c] + beta[c], act));
  }
  if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// du = valid * dy * dropmask * act'(u)  (gradient at the LN output u; dgamma / dbeta are column reductions of it),
// dx = rstd*(du*gamma - mean(du*gamma) - xhat*mean(du*gamma*xhat)),  dres = valid * dy
template <typename T, int NV, int ACT = -1>
__global__ __launch_bounds__(256) void ln_act_bwd_kernel(int rows, int D, int Tn, const T* __restrict__ dy, const T* __restrict__ x,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, int act_rt,
                                                         const int32_t* __restrict__ lens, float p, const uint64_t* seed_base,
                                                         uint64_t seed_off, T* __restrict__ du_out, T* __restrict__ dx,
                                                         T* __restrict__ dres) {
  const int act = ACT >= 0 ? ACT : act_rt;
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t base = (int64_t)row * D;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const float mu = mean[row], rs = rstd[row];
  const bool ok = row_valid(lens, Tn, row);
  float g[NV], xh[NV];
  float a = 0.f, b = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    g[i] = 0.f; xh[i] = 0.f;
    if (c < D) {
      const float d = ok ? ldf(dy + base + c) : 0.f;
      if (dres) stf(dres + base + c, d);
      xh[i] = (ldf(x + base + c) - mu) * rs;
      const float u = xh[i] * gamma[c] + beta[c];
      float du = d * act_deriv(u, act);
      if (p > 0.f) du *= dropout_scale(seed, (uint64_t)(base + c), p, inv_keep);
      stf(du_out + base + c, du);
      g[i] = du * gamma[c];
      a += g[i];
      b += g[i] * xh[i];
    }
  }
  a = wave_sum(a) / (float)D;
  b = wave_sum(b) / (float)D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < D) stf(dx + base + c, rs * (g[i] - a - xh[i] * b));
  }
}

// ------------------------------------------------------------------------------------------------
// Rational-quadratic spline coupling with linear tails, NB bins, one thread per row.
// h[r, 0:NB] / h[r, NB:2NB] scaled by hscale are the unnormalised widths / heights, h[r, 2NB:3NB-1] the unnormalised
// inner derivatives (flow.py:296-305); transform.py:96-216 for the arithmetic.  Rows past the utterance give 0.
// ------------------------------------------------------------------------------------------------
template <int NB>
struct Spline {
  float cw[NB + 1], ch[NB + 1], d[NB + 1];      // knot positions and derivatives
  float sw[NB], sh[NB];                         // softmax(widths), softmax(heights) (kept for the backward)
  __device__ __forceinline__ void knots(const float* u, float scale, float bound, float mn, float* cs, float* sm) {
    float mx = -3.4e38f;
#pragma unroll
    for (int i = 0; i < NB; ++i) mx = fmaxf(mx, u[i] * scale);
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i) { sm[i] = expf(u[i] * scale - mx); tot += sm[i]; }
    const float inv = 1.f / tot;
    float cum = 0.f;
    cs[0] = -bound;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      sm[i] *= inv;
      cum += mn + (1.f - mn * NB) * sm[i];
      cs[i + 1] = 2.f * bound * cum - bound;
    }
    cs[NB] = bound;
  }
  __device__ __forceinline__ void build(const float* h, float hscale, float bound, float min_w, float min_h, float min_d) {
    knots(h, hscale, bound, min_w, cw, sw);
    knots(h + NB, hscale, bound, min_h, ch, sh);
    const float cst = logf(expf(1.f - min_d) - 1.f);
#pragma unroll
    for (int k = 0; k <= NB; ++k) {
      const float ud = (k == 0 || k == NB) ? cst : h[2 * NB + k - 1];
      d[k] = min_d + (ud > 20.f ? ud : log1pf(expf(ud)));     // F.softplus (threshold 20)
    }
  }
  // bin of v among the knots `loc` (last knot nudged by 1e-6, transform.py:33-41)
  __device__ __forceinline__ int bin(const float* loc, float v) const {
    int n = 0;
#pragma unroll
    for (int k = 0; k <= NB; ++k) n += (v >= (k == NB ? loc[k] + 1e-6f : loc[k])) ? 1 : 0;
    n -= 1;
    return n < 0 ? 0 : (n > NB - 1 ? NB - 1 : n);
  }
};

template <int NB>
__global__ void rq_spline_fwd_kernel(int64_t rows, int Tn, const float* __restrict__ x, const float* __restrict__ h, float hscale,
                                     float bound, const int32_t* __restrict__ lens, int inverse, float* __restrict__ out,
                                     float* __restrict__ lad, int lad_accumulate) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float o = 0.f, l = 0.f;
  if (row_valid(lens, Tn, r)) {
    const float xv = x[r];
    o = xv;
    if (xv >= -bound && xv <= bound) {
      float hv[3 * NB - 1];
#pragma unroll
      for (int i = 0; i < 3 * NB - 1; ++i) hv[i] = h[r * (3 * NB - 1) + i];
      Spline<NB> s;
      s.build(hv, hscale, bound, 1e-3f, 1e-3f, 1e-3f);
      const int k = s.bin(inverse ? s.ch : s.cw, xv);
      float in_cw = 0.f, in_w = 0.f, in_ch = 0.f, in_h = 0.f, da = 0.f, db = 0.f;
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (i == k) { in_cw = s.cw[i]; in_w = s.cw[i + 1] - s.cw[i]; in_ch = s.ch[i]; in_h = s.ch[i + 1] - s.ch[i]; da = s.d[i]; db = s.d[i + 1]; }
      const float dl = in_h / in_w;
      if (inverse) {
        const float e = xv - in_ch, q = da + db - 2.f * dl;
        const float A = e * q + in_h * (dl - da), Bq = in_h * da - e * q, Cq = -dl * e;
        const float root = (2.f * Cq) / (-Bq - sqrtf(Bq * Bq - 4.f * A * Cq));
        o = root * in_w + in_cw;
        const float tt = root * (1.f - root);
        const float den = dl + q * tt;
        const float num = dl * dl * (db * root * root + 2.f * dl * tt + da * (1.f - root) * (1.f - root));
        l = -(logf(num) - 2.f * logf(den));
      } else {
        const float th = (xv - in_cw) / in_w, tt = th * (1.f - th);
        const float den = dl + (da + db - 2.f * dl) * tt;
        o = in_ch + in_h * (dl * th * th + da * tt) / den;
        const float num = dl * dl * (db * th * th + 2.f * dl * tt + da * (1.f - th) * (1.f - th));
        l = logf(num) - 2.f * logf(den);
      }
    }
  }
  out[r] = o;
  lad[r] = (lad_accumulate ? lad[r] : 0.f) + l;
}

// reverse mode of the forward (non-inverse) branch; g_lad[b] is the gradient of every row's log|det| of utterance b
template <int NB>
__global__ void rq_spline_bwd_kernel(int64_t rows, int Tn, const float* __restrict__ x, const float* __restrict__ h, float hscale,
                                     float bound, const int32_t* __restrict__ lens, const float* __restrict__ g_out,
                                     const float* __restrict__ g_lad, float* __restrict__ dx, float* __restrict__ dh) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float gh[3 * NB - 1];
#pragma unroll
  for (int i = 0; i < 3 * NB - 1; ++i) gh[i] = 0.f;
  float gx = 0.f;
  if (row_valid(lens, Tn, r)) {
    const float xv = x[r];
    const float go = g_out ? g_out[r] : 0.f;
    const float gl = g_lad ? g_lad[r / Tn] : 0.f;
    gx = go;                                       // identity tails
    if (xv >= -bound && xv <= bound) {
      float hv[3 * NB - 1];
#pragma unroll
      for (int i = 0; i < 3 * NB - 1; ++i) hv[i] = h[r * (3 * NB - 1) + i];
      Spline<NB> s;
      s.build(hv, hscale, bound, 1e-3f, 1e-3f, 1e-3f);
      const int k = s.bin(s.cw, xv);
      float in_cw = 0.f, wid = 0.f, in_ch = 0.f, hgt = 0.f, a = 0.f, b = 0.f;
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if (i == k) { in_cw = s.cw[i]; wid = s.cw[i + 1] - s.cw[i]; in_ch = s.ch[i]; hgt = s.ch[i + 1] - s.ch[i]; a = s.d[i]; b = s.d[i + 1]; }
      const float sl = hgt / wid;
      const float th = (xv - in_cw) / wid, tt = th * (1.f - th), omt = 1.f - th;
      const float q = a + b - 2.f * sl;
      const float den = sl + q * tt;
      const float N1 = sl * th * th + a * tt;
      const float M = b * th * th + 2.f * sl * tt + a * omt * omt;
      const float num = sl * sl * M;
      // out = in_ch + hgt*N1/den ; lad = log(num) - 2 log(den)
      float g_inch = go, g_hgt = go * N1 / den;
      const float g_N1 = go * hgt / den;
      const float g_den = -go * hgt * N1 / (den * den) - 2.f * gl / den;
      const float g_num = gl / num;
      float g_s = g_num * (2.f * sl * M + sl * sl * 2.f * tt);
      const float g_M = g_num * sl * sl;
      float g_b = g_M * th * th, g_a = g_M * omt * omt, g_tt = g_M * 2.f * sl, g_th = g_M * (2.f * b * th - 2.f * a * omt);
      g_s += g_N1 * th * th; g_a += g_N1 * tt; g_th += g_N1 * 2.f * sl * th; g_tt += g_N1 * a;
      g_s += g_den * (1.f - 2.f * tt); g_a += g_den * tt; g_b += g_den * tt; g_tt += g_den * q;
      g_th += g_tt * (1.f - 2.f * th);
      gx = g_th / wid;
      float g_incw = -g_th / wid;
      float g_wid = -g_th * th / wid;
      g_hgt += g_s / wid;
      g_wid += -g_s * sl / wid;
      // scatter to the knot arrays: in_cw = cw[k], wid = cw[k+1]-cw[k], in_ch = ch[k], hgt = ch[k+1]-ch[k], a = d[k], b = d[k+1]
      float gcw_lo = g_incw - g_wid, gcw_hi = g_wid;        // d/d cw[k], d/d cw[k+1]
      float gch_lo = g_inch - g_hgt, gch_hi = g_hgt;
      // cw[j] (1 <= j <= NB-1) = 2*bound*sum_{i<j} w_i - bound  ->  g_w_i = 2*bound * sum_{j>i, j<=NB-1} g_cw[j]
      float gw[NB], ghh[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        float accw = 0.f, acch = 0.f;
        // g_cw[j] is gcw_lo at j == k, gcw_hi at j == k+1 (interior knots only)
        if (k > i && k >= 1 && k <= NB - 1) { accw += gcw_lo; acch += gch_lo; }
        if (k + 1 > i && k + 1 <= NB - 1) { accw += gcw_hi; acch += gch_hi; }
        gw[i] = 2.f * bound * accw * (1.f - 1e-3f * NB);
        ghh[i] = 2.f * bound * acch * (1.f - 1e-3f * NB);
      }
      float dotw = 0.f, doth = 0.f;
#pragma unroll
      for (int i = 0; i < NB; ++i) { dotw += s.sw[i] * gw[i]; doth += s.sh[i] * ghh[i]; }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        gh[i] = s.sw[i] * (gw[i] - dotw) * hscale;
        gh[NB + i] = s.sh[i] * (ghh[i] - doth) * hscale;
      }
      // d[j] = min_d + softplus(ud_j) for interior j; sigmoid(ud) = 1 - exp(-softplus(ud))
#pragma unroll
      for (int j = 1; j <= NB - 1; ++j) {
        const float gd = (j == k ? g_a : 0.f) + (j == k + 1 ? g_b : 0.f);
        const float ud = hv[2 * NB + j - 1];
        gh[2 * NB + j - 1] = gd * (ud > 20.f ? 1.f : 1.f / (1.f + expf(-ud)));
      }
    }
  }
  dx[r] = gx;
#pragma unroll
  for (int i = 0; i < 3 * NB - 1; ++i) dh[r * (3 * NB - 1) + i] = gh[i];
}

// ------------------------------------------------------------------------------------------------
// Glue stage 1 (duration_predictor.py:239-245 + ElementwiseAffine flow.py:66-93):
//   e = noise*mask ; z_c = (m_c + exp(logs_c)*e_c)*mask, c = 0, 1         noise: (B, 2, T) as the reference draws it
// backward: dm_c[b] = sum_t mask*dz_c ; dlogs_c[b] = sum_t mask*dz_c*exp(logs_c)*e_c   (per-utterance partials)
// ------------------------------------------------------------------------------------------------
__global__ void sdp_head_fwd_kernel(int B, int Tn, const float* __restrict__ noise, const int32_t* __restrict__ lens,
                                    const float* __restrict__ m, const float* __restrict__ logs, float* __restrict__ z0,
                                    float* __restrict__ z1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Tn) return;
  const int b = i / Tn, t = i % Tn;
  const bool ok = t < lens[b];
  z0[i] = ok ? m[0] + expf(logs[0]) * noise[((int64_t)b * 2 + 0) * Tn + t] : 0.f;
  z1[i] = ok ? m[1] + expf(logs[1]) * noise[((int64_t)b * 2 + 1) * Tn + t] : 0.f;
}

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const float t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return t;
}

// part[b, 0..3] = dm0, dm1, dlogs0, dlogs1
__global__ __launch_bounds__(256) void sdp_head_bwd_kernel(int Tn, const float* __restrict__ noise, const int32_t* __restrict__ lens,
                                                           const float* __restrict__ logs, const float* __restrict__ dz0,
                                                           const float* __restrict__ dz1, float* __restrict__ part) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const int n = lens[b] < Tn ? lens[b] : Tn;
  float a0 = 0.f, a1 = 0.f, l0 = 0.f, l1 = 0.f;
  const float e0s = expf(logs[0]), e1s = expf(logs[1]);
  for (int t = threadIdx.x; t < n; t += 256) {
    const float g0 = dz0[(int64_t)b * Tn + t], g1 = dz1[(int64_t)b * Tn + t];
    a0 += g0; a1 += g1;
    l0 += g0 * e0s * noise[((int64_t)b * 2 + 0) * Tn + t];
    l1 += g1 * e1s * noise[((int64_t)b * 2 + 1) * Tn + t];
  }
  a0 = block_sum_256(a0, sh); a1 = block_sum_256(a1, sh); l0 = block_sum_256(l0, sh); l1 = block_sum_256(l1, sh);
  if (threadIdx.x == 0) { part[b * 4 + 0] = a0; part[b * 4 + 1] = a1; part[b * 4 + 2] = l0; part[b * 4 + 3] = l1; }
}

// ------------------------------------------------------------------------------------------------
// Glue stage 2 (duration_predictor.py:249-262: dequantisation + LogFlow + ElementwiseAffine):
//   u = sigmoid(z_u)*mask ; v = (w - u)*mask ; lz = log(max(v, 1e-5))*mask ; y0 = (m0 + exp(logs0)*lz)*mask ;
//   y1 = (m1 + exp(logs1)*z1)*mask
// backward: g_lz = dy0*exp(logs0) + dlz ; dz_u = -g_lz * [v > 1e-5]/v * sigmoid'(z_u) ; dz1 = dy1*exp(logs1)
// ------------------------------------------------------------------------------------------------
__global__ void sdp_mid_fwd_kernel(int B, int Tn, const float* __restrict__ zu, const float* __restrict__ z1,
                                   const float* __restrict__ w, const int32_t* __restrict__ lens, const float* __restrict__ m,
                                   const float* __restrict__ logs, float* __restrict__ y0, float* __restrict__ y1,
                                   float* __restrict__ lz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Tn) return;
  const int b = i / Tn, t = i % Tn;
  float o0 = 0.f, o1 = 0.f, l = 0.f;
  if (t < lens[b]) {
    const float u = 1.f / (1.f + expf(-zu[i]));
    const float v = w[i] - u;
    l = logf(fmaxf(v, 1e-5f));
    o0 = m[0] + expf(logs[0]) * l;
    o1 = m[1] + expf(logs[1]) * z1[i];
  }
  y0[i] = o0; y1[i] = o1; lz[i] = l;
}

// dzu_extra: gradient reaching z_u from the tail stage (added here so z_u gets ONE gradient tensor)
__global__ __launch_bounds__(256) void sdp_mid_bwd_kernel(int Tn, const float* __restrict__ zu, const float* __restrict__ z1,
                                                          const float* __restrict__ w, const int32_t* __restrict__ lens,
                                                          const float* __restrict__ logs, const float* __restrict__ dy0,
                                                          const float* __restrict__ dy1, const float* __restrict__ dlz,
                                                          float* __restrict__ dzu, float* __restrict__ dz1, float* __restrict__ part) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const int n = lens[b] < Tn ? lens[b] : Tn;
  const float e0s = expf(logs[0]), e1s = expf(logs[1]);
  float a0 = 0.f, a1 = 0.f, l0 = 0.f, l1 = 0.f;
  for (int t = threadIdx.x; t < Tn; t += 256) {
    const int64_t i = (int64_t)b * Tn + t;
    float gzu = 0.f, gz1 = 0.f;
    if (t < n) {
      const float g0 = dy0 ? dy0[i] : 0.f, g1 = dy1 ? dy1[i] : 0.f;
      const float u = 1.f / (1.f + expf(-zu[i]));
      const float v = w[i] - u;
      const float l = logf(fmaxf(v, 1e-5f));
      const float g_lz = g0 * e0s + (dlz ? dlz[i] : 0.f);
      gzu = v > 1e-5f ? -g_lz / v * u * (1.f - u) : 0.f;
      gz1 = g1 * e1s;
      a0 += g0; a1 += g1;
      l0 += g0 * e0s * l;
      l1 += g1 * e1s * z1[i];
    }
    dzu[i] = gzu; dz1[i] = gz1;
  }
  a0 = block_sum_256(a0, sh); a1 = block_sum_256(a1, sh); l0 = block_sum_256(l0, sh); l1 = block_sum_256(l1, sh);
  if (threadIdx.x == 0) { part[b * 4 + 0] = a0; part[b * 4 + 1] = a1; part[b * 4 + 2] = l0; part[b * 4 + 3] = l1; }
}

// ------------------------------------------------------------------------------------------------
// Glue stage 3: the per-utterance NLL (duration_predictor.py:246-280)
//   logdet_q = sum_t mask*(logs_q0 + logs_q1) + sum_t lad_q + sum_t mask*(logsigmoid(z_u) + logsigmoid(-z_u))
//   logq     = sum_t -0.5*mask*(2*log(2pi) + e0^2 + e1^2) - logdet_q
//   logdet_p = sum_t -lz + sum_t mask*(logs_p0 + logs_p1) + sum_t lad_p
//   nll      = sum_t 0.5*mask*(2*log(2pi) + a^2 + b^2) - logdet_p ;   out[b] = nll + logq
// ------------------------------------------------------------------------------------------------
// number of non-padded positions of the batch: sum_b min(lens[b], Tn)  (uniform loads, B is a batch size)
__device__ __forceinline__ int sdp_total_frames(int B, int Tn, const int32_t* __restrict__ lens) {
  int n = 0;
  for (int b = 0; b < B; ++b) n += lens[b] < Tn ? (lens[b] > 0 ? lens[b] : 0) : Tn;
  return n > 0 ? n : 1;
}

__global__ __launch_bounds__(256) void sdp_tail_fwd_kernel(int Tn, const float* __restrict__ noise, const int32_t* __restrict__ lens,
                                                           const float* __restrict__ zu, const float* __restrict__ lz,
                                                           const float* __restrict__ lad_q, const float* __restrict__ lad_p,
                                                           const float* __restrict__ af, const float* __restrict__ bf,
                                                           const float* __restrict__ logs_q, const float* __restrict__ logs_p,
                                                           float* __restrict__ out, int normalize) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const int n = lens[b] < Tn ? lens[b] : Tn;
  float acc = 0.f;
  for (int t = threadIdx.x; t < n; t += 256) {
    const int64_t i = (int64_t)b * Tn + t;
    const float e0 = noise[((int64_t)b * 2 + 0) * Tn + t], e1 = noise[((int64_t)b * 2 + 1) * Tn + t];
    const float z = zu[i];
    // logsigmoid(z) + logsigmoid(-z) = -|z| - 2*log1p(exp(-|z|))
    const float ls = -fabsf(z) - 2.f * log1pf(expf(-fabsf(z)));
    const float logdet_q = logs_q[0] + logs_q[1] + lad_q[i] + ls;
    const float logq = -0.5f * (2.f * LOG_2PI + e0 * e0 + e1 * e1) - logdet_q;
    const float logdet_p = -lz[i] + logs_p[0] + logs_p[1] + lad_p[i];
    const float nll = 0.5f * (2.f * LOG_2PI + af[i] * af[i] + bf[i] * bf[i]) - logdet_p;
    acc += nll + logq;
  }
  acc = block_sum_256(acc, sh);
  if (threadIdx.x == 0) {
    // normalize: / the number of non-padded text positions of the whole batch (models/aas_vc.py:403: `/ torch.sum(h_masks)` applied here)
    if (normalize) acc /= (float)sdp_total_frames((int)gridDim.x, Tn, lens);
    out[b] = acc;
  }
}

// (sdp_total_frames: defined above the forward kernel)
// g[b] = d loss / d out[b].  d_af = g*a, d_bf = g*b, d_lz = +g, d_zu = -g*(1 - 2*sigmoid(z_u)) on valid rows;
// the gradient of every lad row is -g[b] (handed to the spline backward as a per-utterance scalar);
// the direct gradient of each of logs_q0, logs_q1, logs_p0, logs_p1 is sum_b -g[b]*frames[b]: part[b, 0..1] = -g*frames
__global__ void sdp_tail_bwd_kernel(int B, int Tn, const float* __restrict__ g, const int32_t* __restrict__ lens,
                                    const float* __restrict__ zu, const float* __restrict__ af, const float* __restrict__ bf,
                                    float* __restrict__ d_af, float* __restrict__ d_bf, float* __restrict__ d_lz,
                                    float* __restrict__ d_zu, float* __restrict__ neg_g, float* __restrict__ part, int normalize) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Tn) return;
  const int b = i / Tn, t = i % Tn;
  const float gb = normalize ? g[b] / (float)sdp_total_frames(B, Tn, lens) : g[b];
  const bool ok = t < lens[b];
  d_af[i] = ok ? gb * af[i] : 0.f;
  d_bf[i] = ok ? gb * bf[i] : 0.f;
  d_lz[i] = ok ? gb : 0.f;
  d_zu[i] = ok ? -gb * (1.f - 2.f / (1.f + expf(-zu[i]))) : 0.f;
  if (t == 0) {
    neg_g[b] = -gb;
    const int n = lens[b] < Tn ? lens[b] : Tn;
    part[b * 2 + 0] = -gb * (float)n;
    part[b * 2 + 1] = -gb * (float)n;
  }
}

// inference read-out (duration_predictor.py:300-304 after the inverse affine flow, flow.py:90-93):
//   dur = ceil(exp((a - m0)*exp(-logs0)) * mask)
__global__ void sdp_inverse_out_kernel(int B, int Tn, const float* __restrict__ a, const int32_t* __restrict__ lens,
                                       const float* __restrict__ m, const float* __restrict__ logs, float* __restrict__ dur) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Tn) return;
  const int b = i / Tn, t = i % Tn;
  dur[i] = t < lens[b] ? ceilf(expf((a[i] - m[0]) * expf(-logs[0]))) : 0.f;
}

inline int blocks_for(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b > 65535 ? 65535 : (b < 1 ? 1 : b)); }

}  // namespace

extern "C" int s2svc_mask_rows(int dtype, int B, int Tn, int C, const void* x, const int32_t* lens, void* y, void* stream) {
  const int64_t n = (int64_t)B * Tn * C;
  if (n == 0) return 0;
  S2S_REQUIRE(lens, "mask_rows: lens required");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32) hipLaunchKernelGGL(mask_rows_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, st, n, Tn, C, (const float*)x, lens, (float*)y);
  else hipLaunchKernelGGL(mask_rows_kernel<bf16_t>, dim3(blocks_for(n)), dim3(256), 0, st, n, Tn, C, (const bf16_t*)x, lens, (bf16_t*)y);
  S2S_CHECK_LAUNCH("mask_rows_kernel");
  return 0;
}

extern "C" int s2svc_expand_fwd(int dtype, int B, int Tn, int C, const float* a, const float* w, const float* bias, const void* g,
                                const int32_t* lens, void* y, void* stream) {
  const int64_t n = (int64_t)B * Tn * C;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32) hipLaunchKernelGGL(expand_fwd_kernel<float>, dim3(blocks_for(n)), dim3(256), 0, st, n, Tn, C, a, w, bias, (const float*)g, lens, (float*)y);
  else hipLaunchKernelGGL(expand_fwd_kernel<bf16_t>, dim3(blocks_for(n)), dim3(256), 0, st, n, Tn, C, a, w, bias, (const bf16_t*)g, lens, (bf16_t*)y);
  S2S_CHECK_LAUNCH("expand_fwd_kernel");
  return 0;
}

extern "C" int s2svc_expand_bwd(int dtype, int B, int Tn, int C, const void* dy, const float* w, const int32_t* lens, void* dg,
                                float* da, void* stream) {
  const int64_t rows = (int64_t)B * Tn;
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (dtype == S2S_F32) hipLaunchKernelGGL(expand_bwd_kernel<float>, grid, dim3(256), 0, st, rows, Tn, C, (const float*)dy, w, lens, (float*)dg, da);
  else hipLaunchKernelGGL(expand_bwd_kernel<bf16_t>, grid, dim3(256), 0, st, rows, Tn, C, (const bf16_t*)dy, w, lens, (bf16_t*)dg, da);
  S2S_CHECK_LAUNCH("expand_bwd_kernel");
  return 0;
}

extern "C" int s2svc_ln_act_fwd(int dtype, int rows, int D, int Tn, const void* x, const float* gamma, const float* beta, float eps,
                                int act, const void* res, const int32_t* lens, float drop_p, const uint64_t* seed_base,
                                uint64_t seed_off, void* y, float* mean, float* rstd, void* stream) {
  S2S_REQUIRE(rows >= 0 && D > 0 && D <= 1024 && mean && rstd, "ln_act_fwd: bad arguments (D <= 1024)");
  S2S_REQUIRE(!lens || Tn > 0, "ln_act_fwd: lens needs T");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((rows + 3) / 4), block(256);
#define S2S_LNACT(T, ...)                                                                                                    \
  hipLaunchKernelGGL((ln_act_fwd_kernel<T, __VA_ARGS__>), grid, block, 0, st, rows, D, Tn, (const T*)x, gamma, beta, eps, act, (const T*)res, \
                     lens, drop_p, seed_base, seed_off, (T*)y, mean, rstd)
  if (dtype == S2S_F32) {
    if (D <= 256 && act == S2S_ACT_GELU) S2S_LNACT(float, 4, S2S_ACT_GELU);
    else if (D <= 512 && act == S2S_ACT_GELU) S2S_LNACT(float, 8, S2S_ACT_GELU);
    else if (D <= 512) S2S_LNACT(float, 8);
    else S2S_LNACT(float, 16);
  } else {
    if (D <= 256 && act == S2S_ACT_GELU) S2S_LNACT(bf16_t, 4, S2S_ACT_GELU);
    else if (D <= 512 && act == S2S_ACT_GELU) S2S_LNACT(bf16_t, 8, S2S_ACT_GELU);
    else if (D <= 512) S2S_LNACT(bf16_t, 8);
    else S2S_LNACT(bf16_t, 16);
  }
#undef S2S_LNACT
  S2S_CHECK_LAUNCH("ln_act_fwd_kernel");
  return 0;
}

extern "C" int s2svc_dw_ln_act_fwd(int B, int Tn, int D, int ks, int dil, const float* x, const float* dw_w, const float* dw_b,
                                   const float* gamma, const float* beta, float eps, int act, float* u, float* y, float* mean,
                                   float* rstd, void* stream) {
  S2S_REQUIRE(B >= 0 && Tn > 0 && D > 0 && D <= 512 && ks >= 1 && ks % 2 == 1 && dil >= 1 && x && dw_w && u && y && mean && rstd,
              "dw_ln_act_fwd: bad arguments (fp32, D <= 512, odd kernel size)");
  const int rows = B * Tn;
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((rows + 3) / 4), block(256);
#define S2S_DWLN(...)                                                                                                         \
  hipLaunchKernelGGL((dw_ln_act_fwd_kernel<float, __VA_ARGS__>), grid, block, 0, st, rows, D, Tn, ks, dil, x, dw_w, dw_b, gamma, beta, eps, \
                     act, u, y, mean, rstd)
  if (D <= 256 && act == S2S_ACT_GELU) S2S_DWLN(4, S2S_ACT_GELU);
  else if (act == S2S_ACT_GELU) S2S_DWLN(8, S2S_ACT_GELU);
  else S2S_DWLN(8);
#undef S2S_DWLN
  S2S_CHECK_LAUNCH("dw_ln_act_fwd_kernel");
  return 0;
}

extern "C" int s2svc_ln_act_bwd(int dtype, int rows, int D, int Tn, const void* dy, const void* x, const float* mean,
                                const float* rstd, const float* gamma, const float* beta, int act, const int32_t* lens,
                                float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* du, void* dx, void* dres,
                                void* stream) {
  S2S_REQUIRE(rows >= 0 && D > 0 && D <= 1024 && du && dx, "ln_act_bwd: bad arguments (D <= 1024)");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((rows + 3) / 4), block(256);
#define S2S_LNACTB(T, ...)                                                                                                   \
  hipLaunchKernelGGL((ln_act_bwd_kernel<T, __VA_ARGS__>), grid, block, 0, st, rows, D, Tn, (const T*)dy, (const T*)x, mean, rstd, gamma, beta, \
                     act, lens, drop_p, seed_base, seed_off, (T*)du, (T*)dx, (T*)dres)
  if (dtype == S2S_F32) {
    if (D <= 256 && act == S2S_ACT_GELU) S2S_LNACTB(float, 4, S2S_ACT_GELU);
    else if (D <= 512 && act == S2S_ACT_GELU) S2S_LNACTB(float, 8, S2S_ACT_GELU);
    else if (D <= 512) S2S_LNACTB(float, 8);
    else S2S_LNACTB(float, 16);
  } else {
    if (D <= 256 && act == S2S_ACT_GELU) S2S_LNACTB(bf16_t, 4, S2S_ACT_GELU);
    else if (D <= 512 && act == S2S_ACT_GELU) S2S_LNACTB(bf16_t, 8, S2S_ACT_GELU);
    else if (D <= 512) S2S_LNACTB(bf16_t, 8);
    else S2S_LNACTB(bf16_t, 16);
  }
#undef S2S_LNACTB
  S2S_CHECK_LAUNCH("ln_act_bwd_kernel");
  return 0;
}

extern "C" int s2svc_rq_spline_fwd(int B, int Tn, int bins, const float* x, const float* h, float hscale, float bound,
                                   const int32_t* lens, int inverse, float* out, float* lad, int lad_accumulate, void* stream) {
  S2S_REQUIRE(bins == 10, "rq_spline: only 10 bins are built (ConvFlow default, flow.py:257)");
  const int64_t rows = (int64_t)B * Tn;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(rq_spline_fwd_kernel<10>, dim3(blocks_for(rows)), dim3(256), 0, (hipStream_t)stream, rows, Tn, x, h, hscale, bound,
                     lens, inverse, out, lad, lad_accumulate);
  S2S_CHECK_LAUNCH("rq_spline_fwd_kernel");
  return 0;
}

extern "C" int s2svc_rq_spline_bwd(int B, int Tn, int bins, const float* x, const float* h, float hscale, float bound,
                                   const int32_t* lens, const float* g_out, const float* g_lad, float* dx, float* dh, void* stream) {
  S2S_REQUIRE(bins == 10, "rq_spline: only 10 bins are built (ConvFlow default, flow.py:257)");
  const int64_t rows = (int64_t)B * Tn;
  if (rows == 0) return 0;
  hipLaunchKernelGGL(rq_spline_bwd_kernel<10>, dim3(blocks_for(rows)), dim3(256), 0, (hipStream_t)stream, rows, Tn, x, h, hscale, bound,
                     lens, g_out, g_lad, dx, dh);
  S2S_CHECK_LAUNCH("rq_spline_bwd_kernel");
  return 0;
}

extern "C" int s2svc_sdp_head_fwd(int B, int Tn, const float* noise, const int32_t* lens, const float* m, const float* logs,
                                  float* z0, float* z1, void* stream) {
  if (B * Tn == 0) return 0;
  hipLaunchKernelGGL(sdp_head_fwd_kernel, dim3(blocks_for((int64_t)B * Tn)), dim3(256), 0, (hipStream_t)stream, B, Tn, noise, lens, m, logs, z0, z1);
  S2S_CHECK_LAUNCH("sdp_head_fwd_kernel");
  return 0;
}

extern "C" int s2svc_sdp_head_bwd(int B, int Tn, const float* noise, const int32_t* lens, const float* logs, const float* dz0,
                                  const float* dz1, float* part, void* stream) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(sdp_head_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, Tn, noise, lens, logs, dz0, dz1, part);
  S2S_CHECK_LAUNCH("sdp_head_bwd_kernel");
  return 0;
}

extern "C" int s2svc_sdp_mid_fwd(int B, int Tn, const float* zu, const float* z1, const float* w, const int32_t* lens,
                                 const float* m, const float* logs, float* y0, float* y1, float* lz, void* stream) {
  if (B * Tn == 0) return 0;
  hipLaunchKernelGGL(sdp_mid_fwd_kernel, dim3(blocks_for((int64_t)B * Tn)), dim3(256), 0, (hipStream_t)stream, B, Tn, zu, z1, w, lens, m, logs, y0, y1, lz);
  S2S_CHECK_LAUNCH("sdp_mid_fwd_kernel");
  return 0;
}

extern "C" int s2svc_sdp_mid_bwd(int B, int Tn, const float* zu, const float* z1, const float* w, const int32_t* lens,
                                 const float* logs, const float* dy0, const float* dy1, const float* dlz, float* dzu, float* dz1,
                                 float* part, void* stream) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(sdp_mid_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, Tn, zu, z1, w, lens, logs, dy0, dy1, dlz, dzu, dz1, part);
  S2S_CHECK_LAUNCH("sdp_mid_bwd_kernel");
  return 0;
}

extern "C" int s2svc_sdp_tail_fwd(int B, int Tn, const float* noise, const int32_t* lens, const float* zu, const float* lz,
                                  const float* lad_q, const float* lad_p, const float* af, const float* bf, const float* logs_q,
                                  const float* logs_p, float* out, int normalize, void* stream) {
  if (B == 0) return 0;
  hipLaunchKernelGGL(sdp_tail_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, Tn, noise, lens, zu, lz, lad_q, lad_p, af, bf, logs_q, logs_p, out,
                     normalize);
  S2S_CHECK_LAUNCH("sdp_tail_fwd_kernel");
  return 0;
}

extern "C" int s2svc_sdp_inverse_out(int B, int Tn, const float* a, const int32_t* lens, const float* m, const float* logs,
                                     float* dur, void* stream) {
  if (B * Tn == 0) return 0;
  hipLaunchKernelGGL(sdp_inverse_out_kernel, dim3(blocks_for((int64_t)B * Tn)), dim3(256), 0, (hipStream_t)stream, B, Tn, a, lens, m, logs, dur);
  S2S_CHECK_LAUNCH("sdp_inverse_out_kernel");
  return 0;
}

extern "C" int s2svc_sdp_tail_bwd(int B, int Tn, const float* g, const int32_t* lens, const float* zu, const float* af,
                                  const float* bf, float* d_af, float* d_bf, float* d_lz, float* d_zu, float* neg_g, float* part,
                                  int normalize, void* stream) {
  if (B * Tn == 0) return 0;
  hipLaunchKernelGGL(sdp_tail_bwd_kernel, dim3(blocks_for((int64_t)B * Tn)), dim3(256), 0, (hipStream_t)stream, B, Tn, g, lens, zu, af, bf,
                     d_af, d_bf, d_lz, d_zu, neg_g, part, normalize);
  S2S_CHECK_LAUNCH("sdp_tail_bwd_kernel");
  return 0;
}
