// Core of the Conformer convolution module in training mode, bf16, channel-last:
//     (pointwise conv 1) -> GLU -> depthwise conv k -> BatchNorm1d (batch statistics) -> Swish -> (pointwise conv 2)
// reference: modules/conformer/convolution.py:56-79.  The two pointwise convolutions stay GEMMs; what lies between them was five
// launches forward (GLU, depthwise conv, two for the statistics, BatchNorm-apply + Swish) and eight backward.  Here:
//   forward   convmod_fwd_kernel      y2 (B,T,2C) -> z (B,T,C) = dwconv(glu(y2)) and the per-tile sums of z and z^2   [+ finalize]
//             bn_swish_apply_kernel   z -> swish(BN(z))  (16-byte accesses; nothing but z is kept for the backward pass)
//   backward  convmod_bwd_stats       da, z -> per-chunk sums of dpre and dpre * xhat (dpre = da * swish'(BN(z)), recomputed)  [+ sum]
//             convmod_bwd_kernel      da, z, y2 -> dy2: BatchNorm' -> depthwise data gradient -> GLU', and the per-tile partial sums
//                                     of the depthwise weight / bias gradients (g = glu(y2) is recomputed, never stored)
//             convmod_wgrad_final     partials -> dw (C,1,k), db (C)   (off the data-gradient chain)
// A workgroup owns 64 frames x 64 channels of one utterance (+ (k-1)/2 halo frames each side, zero outside the utterance as the
// convolution's 'same' padding; padded frames inside T are data, exactly like the reference).  HBM-bound: every tensor is read
// once per tile (+ 22 % halo at k = 15) in 16-byte accesses, the taps run on an LDS image [frame][channel] that a lane reads
// down its own column (conflict-free).  All sums have a fixed order (deterministic).
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

constexpr int TT = 64, CT = 64, FR = TT / 4;

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float round_bf(float v) { return bf2f(f2bf(v)); }

// per-channel parameters (fp32 slices of a flat parameter buffer: 4-byte alignment is all that is guaranteed)
__device__ __forceinline__ void ldp8(const float* __restrict__ p, float (&f)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = p[e];
}

__device__ __forceinline__ void st8(float* p, const float (&f)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}

template <int KS>
__global__ __launch_bounds__(256) void convmod_fwd_kernel(int Tn, int C, const bf16_t* __restrict__ y2, const float* __restrict__ w,
                                                          const float* __restrict__ bias, bf16_t* __restrict__ z,
                                                          float* __restrict__ ws, int tchunks, const int32_t* __restrict__ vlens) {
  constexpr int PAD = (KS - 1) / 2, ROWS = TT + 2 * PAD, WIN = FR + KS - 1, NP = (ROWS + 31) / 32;
  __shared__ __attribute__((aligned(16))) float G[ROWS * CT];
  __shared__ float red[2][4][CT];
  const int c0 = blockIdx.x * CT;
  const int b = blockIdx.y / tchunks, t0 = (blockIdx.y % tchunks) * TT;
  const int Te = vlens ? (vlens[b] < Tn ? vlens[b] : Tn) : Tn;      // frames >= Te are absent (common.h): zero padding, not in the statistics
  const int v = threadIdx.x & 7, r8 = threadIdx.x >> 3;
  const int cl = threadIdx.x & 63, rq = threadIdx.x >> 6;
  const bf16_t* yb = y2 + (int64_t)b * Tn * 2 * C + c0 + v * 8;
  // ---- phase 1: g = a * sigmoid(gate) for the tile's frames and its halo -> LDS (all loads first)
  uint4 av[NP], gv[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = r8 + 32 * i, t = t0 - PAD + p;
    av[i] = make_uint4(0, 0, 0, 0);
    gv[i] = make_uint4(0, 0, 0, 0);
    if (p < ROWS && t >= 0 && t < Te) {
      av[i] = *reinterpret_cast<const uint4*>(yb + (int64_t)t * 2 * C);
      gv[i] = *reinterpret_cast<const uint4*>(yb + (int64_t)t * 2 * C + C);
    }
  }
  float wr[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) wr[j] = w[(int64_t)(c0 + cl) * KS + j];
  const float bs = bias ? bias[c0 + cl] : 0.f;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int p = r8 + 32 * i;
    if (p < ROWS) {
      float a[8], gt[8], g[8];
      unpack_bf16x8(av[i], a);
      unpack_bf16x8(gv[i], gt);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = a[e] * sigm(gt[e]);        // rows outside the utterance: a = 0 -> g = 0
      st8(&G[p * CT + v * 8], g);
    }
  }
  __syncthreads();
  // ---- phase 2: a lane owns one channel and FR consecutive frames; sliding window in registers
  float win[WIN], out[FR];
#pragma unroll
  for (int i = 0; i < WIN; ++i) win[i] = G[(rq * FR + i) * CT + cl];
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int o = 0; o < FR; ++o) {
    float acc = bs;
#pragma unroll
    for (int j = 0; j < KS; ++j) acc += wr[j] * win[o + j];
    out[o] = round_bf(acc);
    if (t0 + rq * FR + o < Te) { s0 += out[o]; s1 += out[o] * out[o]; }
  }
  __syncthreads();
#pragma unroll
  for (int o = 0; o < FR; ++o) G[(rq * FR + o) * CT + cl] = out[o];
  red[0][rq][cl] = s0;
  red[1][rq][cl] = s1;
  __syncthreads();
  if (rq == 0) {
    const int chunk = blockIdx.y;
    ws[((int64_t)chunk * 2 + 0) * C + c0 + cl] = ((red[0][0][cl] + red[0][1][cl]) + red[0][2][cl]) + red[0][3][cl];
    ws[((int64_t)chunk * 2 + 1) * C + c0 + cl] = ((red[1][0][cl] + red[1][1][cl]) + red[1][2][cl]) + red[1][3][cl];
  }
  // ---- phase 3: the z tile, 16 bytes per lane
#pragma unroll
  for (int i = 0; i < TT / 32; ++i) {
    const int r = r8 + 32 * i, t = t0 + r;
    if (t < Tn) {
      float f[8];
      load_f32x8(&G[r * CT + v * 8], f);
      *reinterpret_cast<uint4*>(z + ((int64_t)b * Tn + t) * C + c0 + v * 8) = pack_bf16x8(f);
    }
  }
}

// sum of the chunk partials (sum z, sum z^2) -> mean, rstd, running statistics (the arithmetic of bn_stage2_finalize_kernel)
__global__ __launch_bounds__(256) void convmod_bn_finalize_kernel(int C, int chunks, const float* __restrict__ ws, int rows, float eps,
                                                                  float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                                  float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                  int64_t* __restrict__ num_batches, int Tn,
                                                                  const int32_t* __restrict__ vlens) {
  const int n = rows_present(rows, Tn, vlens);
  __shared__ float sh[2][4][64];
  const int cl = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float t0 = 0.f, t1 = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int k = kg; k < chunks; k += 4) {        // same order of additions, 16 loads in flight
      t0 += ws[((int64_t)k * 2 + 0) * C + c];
      t1 += ws[((int64_t)k * 2 + 1) * C + c];
    }
  }
  sh[0][kg][cl] = t0;
  sh[1][kg][cl] = t1;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches) *num_batches += 1;
  if (kg != 0 || c >= C) return;
  const float scale = 1.0f / (float)n;
  const float m = (((sh[0][0][cl] + sh[0][1][cl]) + sh[0][2][cl]) + sh[0][3][cl]) * scale;
  const float ex2 = (((sh[1][0][cl] + sh[1][1][cl]) + sh[1][2][cl]) + sh[1][3][cl]) * scale;
  float vc = ex2 - m * m;
  vc = vc > 0.f ? vc : 0.f;
  mean[c] = m;
  rstd[c] = 1.0f / sqrtf(vc + eps);
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * m;
    const float unb = n > 1 ? vc * ((float)n / (float)(n - 1)) : vc;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * unb;
  }
}

// out = swish((z - mean) * rstd * gamma + beta), a workgroup = 64 channels x rows_per_wg rows
__global__ __launch_bounds__(256) void bn_swish_apply_kernel(int rows, int C, const bf16_t* __restrict__ z, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, bf16_t* __restrict__ out, int rows_per_wg,
                                                             int Tn, const int32_t* __restrict__ vlens) {
  const int v = threadIdx.x & 7, r8 = threadIdx.x >> 3;
  const int c = blockIdx.x * CT + v * 8;
  const int r0 = blockIdx.y * rows_per_wg;
  const int r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
  float m[8], rs[8], ga[8], be[8];
  ldp8(mean + c, m);
  ldp8(rstd + c, rs);
  ldp8(gamma + c, ga);
  ldp8(beta + c, be);
#pragma unroll
  for (int e = 0; e < 8; ++e) { ga[e] *= rs[e]; be[e] -= m[e] * ga[e]; }
#pragma unroll 4
  for (int r = r0 + r8; r < r1; r += 32) {
    float f[8];
    if (!row_present(r, Tn, vlens)) {               // absent frame: zero for whatever reads it as a convolution's padding
      *reinterpret_cast<uint4*>(out + (int64_t)r * C + c) = make_uint4(0, 0, 0, 0);
      continue;
    }
    unpack_bf16x8(*reinterpret_cast<const uint4*>(z + (int64_t)r * C + c), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float pre = f[e] * ga[e] + be[e];
      f[e] = pre * sigm(pre);
    }
    *reinterpret_cast<uint4*>(out + (int64_t)r * C + c) = pack_bf16x8(f);
  }
}

// dpre = da * swish'(pre), pre = xhat * gamma + beta, xhat = (z - mean) * rstd
__device__ __forceinline__ float dswish(float pre) {
  const float s = sigm(pre);
  return s * (1.f + pre * (1.f - s));
}

// per-chunk (64 rows) column sums of dpre and dpre * xhat: ws[chunk][2][C]
__global__ __launch_bounds__(256) void convmod_bwd_stats_kernel(int rows, int C, const bf16_t* __restrict__ da, const bf16_t* __restrict__ z,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ ws, int Tn, const int32_t* __restrict__ vlens) {
  __shared__ float sh[2][4][CT];
  const int v = threadIdx.x & 7, r8 = threadIdx.x >> 3, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c = blockIdx.x * CT + v * 8;
  const int r0 = blockIdx.y * 64;
  uint4 dv[2], zv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r0 + r8 + 32 * i;
    dv[i] = make_uint4(0, 0, 0, 0);
    zv[i] = make_uint4(0, 0, 0, 0);
    if (r < rows && row_present(r, Tn, vlens)) {
      dv[i] = *reinterpret_cast<const uint4*>(da + (int64_t)r * C + c);
      zv[i] = *reinterpret_cast<const uint4*>(z + (int64_t)r * C + c);
    }
  }
  float m[8], rs[8], ga[8], be[8];
  ldp8(mean + c, m);
  ldp8(rstd + c, rs);
  ldp8(gamma + c, ga);
  ldp8(beta + c, be);
  float a0[8], a1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float g[8], zz[8];
    unpack_bf16x8(dv[i], g);               // rows past the end: da = 0 -> no contribution
    unpack_bf16x8(zv[i], zz);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (zz[e] - m[e]) * rs[e];
      const float d = g[e] * dswish(xh * ga[e] + be[e]);
      a0[e] += d;
      a1[e] += d * xh;
    }
  }
  // lanes with the same vector lane v hold the wave's 8 rows: butterfly over lane bits 3..5
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a0[e] = xor32_sum(xor16_sum(xor8_sum(a0[e])));      // (DPP / permlane swaps instead of ds_bpermute, common.h)
    a1[e] = xor32_sum(xor16_sum(xor8_sum(a1[e])));
  }
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[0][wave][v * 8 + e] = a0[e]; sh[1][wave][v * 8 + e] = a1[e]; }
  }
  __syncthreads();
  if (threadIdx.x < CT) {
    const int cl = threadIdx.x;
    ws[((int64_t)blockIdx.y * 2 + 0) * C + blockIdx.x * CT + cl] = ((sh[0][0][cl] + sh[0][1][cl]) + sh[0][2][cl]) + sh[0][3][cl];
    ws[((int64_t)blockIdx.y * 2 + 1) * C + blockIdx.x * CT + cl] = ((sh[1][0][cl] + sh[1][1][cl]) + sh[1][2][cl]) + sh[1][3][cl];
  }
}

// sum of the chunk partials -> sdy, sdyx; optionally accumulated into the gradient slots of beta / gamma
__global__ __launch_bounds__(256) void convmod_sum2_kernel(int C, int chunks, const float* __restrict__ ws, float* __restrict__ sdy,
                                                           float* __restrict__ sdyx, float* __restrict__ dbeta_acc,
                                                           float* __restrict__ dgamma_acc) {
  __shared__ float sh[2][4][64];
  const int cl = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float t0 = 0.f, t1 = 0.f;
  if (c < C) {
#pragma unroll 8
    for (int k = kg; k < chunks; k += 4) {        // same order of additions, 16 loads in flight
      t0 += ws[((int64_t)k * 2 + 0) * C + c];
      t1 += ws[((int64_t)k * 2 + 1) * C + c];
    }
  }
  sh[0][kg][cl] = t0;
  sh[1][kg][cl] = t1;
  __syncthreads();
  if (kg != 0 || c >= C) return;
  t0 = ((sh[0][0][cl] + sh[0][1][cl]) + sh[0][2][cl]) + sh[0][3][cl];
  t1 = ((sh[1][0][cl] + sh[1][1][cl]) + sh[1][2][cl]) + sh[1][3][cl];
  sdy[c] = t0;
  sdyx[c] = t1;
  if (dbeta_acc) dbeta_acc[c] += t0;
  if (dgamma_acc) dgamma_acc[c] += t1;
}

template <int KS>
__global__ __launch_bounds__(256) void convmod_bwd_kernel(int Tn, int C, const bf16_t* __restrict__ da,
                                                          const bf16_t* __restrict__ z, const bf16_t* __restrict__ y2,
                                                          const float* __restrict__ w, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ sdy,
                                                          const float* __restrict__ sdyx, bf16_t* __restrict__ dy2,
                                                          float* __restrict__ wsw, int tchunks, const int32_t* __restrict__ vlens) {
  constexpr int PAD = (KS - 1) / 2, ROWS = TT + 2 * PAD, WIN = FR + KS - 1, NP = (ROWS + 31) / 32;
  constexpr int RED_ROWS = 4 * (KS + 1);
  constexpr int SM_ROWS = (2 * ROWS > TT + RED_ROWS) ? 2 * ROWS : TT + RED_ROWS;
  __shared__ __attribute__((aligned(16))) float S[SM_ROWS * CT];
  float* DZ = S;                        // [ROWS][CT] dz with halo; later rows 0..TT-1 = dg
  float* G = S + ROWS * CT;             // [ROWS][CT] g with halo
  float* RED = S + TT * CT;             // later: [(KS+1)][4][CT] partial weight / bias gradients
  const int c0 = blockIdx.x * CT;
  const int b = blockIdx.y / tchunks, t0 = (blockIdx.y % tchunks) * TT;
  const int v = threadIdx.x & 7, r8 = threadIdx.x >> 3;
  const int cl = threadIdx.x & 63, rq = threadIdx.x >> 6;
  const int cv = c0 + v * 8;
  const bf16_t* yb = y2 + (int64_t)b * Tn * 2 * C + cv;
  const bf16_t* dab = da + (int64_t)b * Tn * C + cv;
  const bf16_t* zb = z + (int64_t)b * Tn * C + cv;
  const int Te = vlens ? (vlens[b] < Tn ? vlens[b] : Tn) : Tn;      // frames >= Te are absent (common.h)
  const float inv_n = 1.0f / (float)rows_present((int)(gridDim.y / tchunks) * Tn, Tn, vlens);
  // ---- phase 1
  {
    float k_s[8], k_b[8], k_m[8], k_r[8], k_1[8], k_2[8], k_3[8];
    {
      float ga[8], sy[8], sx[8];
      ldp8(mean + cv, k_m);
      ldp8(rstd + cv, k_r);
      ldp8(gamma + cv, ga);
      ldp8(beta + cv, k_b);
      ldp8(sdy + cv, sy);
      ldp8(sdyx + cv, sx);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        k_s[e] = ga[e];
        k_1[e] = ga[e] * k_r[e];                 // dz = k_1 * (d - k_2 - xhat * k_3)
        k_2[e] = sy[e] * inv_n;
        k_3[e] = sx[e] * inv_n;
      }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int p = r8 + 32 * i, t = t0 - PAD + p;
      if (p < ROWS) {
        float dz[8], g[8];
        if (t >= 0 && t < Te) {
          float d[8], zz[8], a[8], gt[8];
          unpack_bf16x8(*reinterpret_cast<const uint4*>(dab + (int64_t)t * C), d);
          unpack_bf16x8(*reinterpret_cast<const uint4*>(zb + (int64_t)t * C), zz);
          unpack_bf16x8(*reinterpret_cast<const uint4*>(yb + (int64_t)t * 2 * C), a);
          unpack_bf16x8(*reinterpret_cast<const uint4*>(yb + (int64_t)t * 2 * C + C), gt);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = (zz[e] - k_m[e]) * k_r[e];
            const float dp = d[e] * dswish(xh * k_s[e] + k_b[e]);
            dz[e] = k_1[e] * (dp - k_2[e] - xh * k_3[e]);
            g[e] = a[e] * sigm(gt[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) dz[e] = g[e] = 0.f;
        }
        st8(&DZ[p * CT + v * 8], dz);
        st8(&G[p * CT + v * 8], g);
      }
    }
  }
  float wr[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) wr[j] = w[(int64_t)(c0 + cl) * KS + j];
  __syncthreads();
  // ---- phase 2: depthwise data gradient, weight / bias gradient partials; a lane = one channel, FR frames
  float dg[FR], dwp[KS], dbp = 0.f;
  {
    float wd[WIN], wg[WIN];
#pragma unroll
    for (int i = 0; i < WIN; ++i) { wd[i] = DZ[(rq * FR + i) * CT + cl]; wg[i] = G[(rq * FR + i) * CT + cl]; }
#pragma unroll
    for (int o = 0; o < FR; ++o) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < KS; ++j) acc += wr[j] * wd[o + KS - 1 - j];
      dg[o] = acc;
      dbp += wd[o + PAD];
    }
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int o = 0; o < FR; ++o) acc += wd[o + PAD] * wg[o + j];
      dwp[j] = acc;
    }
  }
  __syncthreads();
#pragma unroll
  for (int o = 0; o < FR; ++o) DZ[(rq * FR + o) * CT + cl] = dg[o];
#pragma unroll
  for (int j = 0; j < KS; ++j) RED[(j * 4 + rq) * CT + cl] = dwp[j];
  RED[(KS * 4 + rq) * CT + cl] = dbp;
  __syncthreads();
  // ---- phase 3a: this tile's partial weight / bias gradients  wsw[chunk][C][KS + 1]
  for (int e = threadIdx.x; e < (KS + 1) * CT; e += 256) {
    const int j = e >> 6, ch = e & 63;
    const float* q = RED + j * 4 * CT + ch;
    wsw[((int64_t)blockIdx.y * C + c0 + ch) * (KS + 1) + j] = ((q[0] + q[CT]) + q[2 * CT]) + q[3 * CT];
  }
  // ---- phase 3b: GLU backward, 16 bytes per lane
  bf16_t* ob = dy2 + (int64_t)b * Tn * 2 * C + cv;
#pragma unroll
  for (int i = 0; i < TT / 32; ++i) {
    const int r = r8 + 32 * i, t = t0 + r;
    if (t >= Te && t < Tn) {                       // absent frame: no gradient leaves it
      *reinterpret_cast<uint4*>(ob + (int64_t)t * 2 * C) = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(ob + (int64_t)t * 2 * C + C) = make_uint4(0, 0, 0, 0);
    } else if (t < Tn) {
      float a[8], gt[8], d[8], oa[8], og[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(yb + (int64_t)t * 2 * C), a);
      unpack_bf16x8(*reinterpret_cast<const uint4*>(yb + (int64_t)t * 2 * C + C), gt);
      load_f32x8(&DZ[r * CT + v * 8], d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float s = sigm(gt[e]);
        oa[e] = d[e] * s;
        og[e] = d[e] * a[e] * s * (1.f - s);
      }
      *reinterpret_cast<uint4*>(ob + (int64_t)t * 2 * C) = pack_bf16x8(oa);
      *reinterpret_cast<uint4*>(ob + (int64_t)t * 2 * C + C) = pack_bf16x8(og);
    }
  }
}

__global__ void convmod_wgrad_final_kernel(int C, int ks, int chunks, const float* __restrict__ wsw, float* __restrict__ dw,
                                           float* __restrict__ db, int accumulate) {
  const int n = C * (ks + 1);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float t = 0.f;
#pragma unroll 16
  for (int k = 0; k < chunks; ++k) t += wsw[(int64_t)k * n + i];      // same order of additions, 16 loads in flight
  const int c = i / (ks + 1), j = i - c * (ks + 1);
  if (j < ks) {
    if (dw) dw[c * ks + j] = (accumulate ? dw[c * ks + j] : 0.f) + t;
  } else if (db) {
    db[c] = (accumulate ? db[c] : 0.f) + t;
  }
}

// ---------------------------------------------------------------------------------------------------------
// BatchNorm1d (training) + activation + dropout on channel-last rows, bf16, C % 8 == 0, in 16-byte accesses -- the Postnet layers
// (pre_postnets.py:108-165: Conv1d -> BatchNorm1d -> Tanh -> Dropout).  Same launch count forward as the scalar kernels of norm.hip
// (statistics, finalize, apply) but vector loads; backward 3 launches instead of 4 + 2: the activation / dropout derivative is
// recomputed inside the statistics pass and inside the data-gradient pass instead of being written by a kernel of its own, and the
// BatchNorm parameter gradients are added to their slots by the reduction's second stage.
// A workgroup = 64 channels (8 vector lanes) x 64 rows; sums in a fixed order.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_grad_saved(float s, float m, int act) {      // elementwise.hip: act_grad_from_saved
  if (act == S2S_ACT_RELU) return s > 0.f ? 1.f : 0.f;
  if (act == S2S_ACT_TANH) { const float yv = m > 0.f ? s / m : 0.f; return 1.f - yv * yv; }
  if (act == S2S_ACT_SIGMOID) { const float yv = m > 0.f ? s / m : 0.f; return yv * (1.f - yv); }
  if (act == S2S_ACT_SWISH) { const float sg = 1.f / (1.f + expf(-s)); return sg * (1.f + s * (1.f - sg)); }
  if (act == S2S_ACT_GELU) {
    const float cdf = 0.5f * (1.f + erff(s * 0.70710678118654752f));
    return cdf + s * 0.3989422804014327f * expf(-0.5f * s * s);
  }
  return 1.f;
}

// reduce a0 / a1 (8 channels per lane, 8 rows per wave) over the workgroup's rows -> ws[chunk][2][C]
__device__ __forceinline__ void bn_chunk_reduce(float (&a0)[8], float (&a1)[8], float (&sh)[2][4][CT], int C, int c, float* __restrict__ ws) {
  const int v = threadIdx.x & 7, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a0[e] = xor32_sum(xor16_sum(xor8_sum(a0[e])));      // (DPP / permlane swaps instead of ds_bpermute, common.h)
    a1[e] = xor32_sum(xor16_sum(xor8_sum(a1[e])));
  }
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { sh[0][wave][v * 8 + e] = a0[e]; sh[1][wave][v * 8 + e] = a1[e]; }
  }
  __syncthreads();
  if (threadIdx.x < CT) {
    const int cl = threadIdx.x, cc = blockIdx.x * CT + cl;
    if (cc < C) {
      ws[((int64_t)blockIdx.y * 2 + 0) * C + cc] = ((sh[0][0][cl] + sh[0][1][cl]) + sh[0][2][cl]) + sh[0][3][cl];
      ws[((int64_t)blockIdx.y * 2 + 1) * C + cc] = ((sh[1][0][cl] + sh[1][1][cl]) + sh[1][2][cl]) + sh[1][3][cl];
    }
  }
  (void)c;
}

__global__ __launch_bounds__(256) void bn_stats_vec_kernel(int rows, int C, const bf16_t* __restrict__ x, float* __restrict__ ws, int rpc,
                                                           int Tn, const int32_t* __restrict__ vlens) {
  __shared__ float sh[2][4][CT];
  const int v = threadIdx.x & 7, r8 = threadIdx.x >> 3;
  const int c = blockIdx.x * CT + v * 8;
  const int r0 = blockIdx.y * rpc;
  const int r1 = r0 + rpc < rows ? r0 + rpc : rows;
  float a0[8], a1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int r = r0 + r8; r < r1; r += 32) {
      if (!row_present(r, Tn, vlens)) continue;
      float f[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(x + (int64_t)r * C + c), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { a0[e] += f[e]; a1[e] += f[e] * f[e]; }
    }
  }
  bn_chunk_reduce(a0, a1, sh, C, c, ws);
}

// y = dropout(act((x - mean) * rstd * gamma + beta)); pre_act (optional) = the BatchNorm output before the activation
__global__ __launch_bounds__(256) void bn_act_apply_vec_kernel(int rows, int C, const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int act, float p,
                                                               const uint64_t* seed_base, uint64_t seed_off, bf16_t* __restrict__ y,
                                                               bf16_t* __restrict__ pre_act, int rows_per_wg, int Tn,
                                                               const int32_t* __restrict__ vlens) {
  const int v = threadIdx.x & 7, r8 = threadIdx.x >> 3;
  const int c = blockIdx.x * CT + v * 8;
  if (c >= C) return;
  const int r0 = blockIdx.y * rows_per_wg;
  const int r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float m[8], rs[8], ga[8], be[8];
  ldp8(mean + c, m);
  ldp8(rstd + c, rs);
  ldp8(gamma + c, ga);
  ldp8(beta + c, be);
#pragma unroll 2
  for (int r = r0 + r8; r < r1; r += 32) {
    const int64_t o = (int64_t)r * C + c;
    float f[8], mk[8];
    if (!row_present(r, Tn, vlens)) {               // absent frame: the next convolution's zero padding
      if (pre_act) *reinterpret_cast<uint4*>(pre_act + o) = make_uint4(0, 0, 0, 0);
      *reinterpret_cast<uint4*>(y + o) = make_uint4(0, 0, 0, 0);
      continue;
    }
    unpack_bf16x8(*reinterpret_cast<const uint4*>(x + o), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = (f[e] - m[e]) * rs[e] * ga[e] + be[e];
    if (pre_act) *reinterpret_cast<uint4*>(pre_act + o) = pack_bf16x8(f);
    if (p > 0.f) dropout_scale8(seed, (uint64_t)o, p, inv_keep, mk);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      f[e] = act_apply(f[e], act);
      if (p > 0.f) f[e] *= mk[e];
    }
    *reinterpret_cast<uint4*>(y + o) = pack_bf16x8(f);
  }
}

// g = dz * mask * act'(saved);  per-chunk sums of g and g * xhat
__global__ __launch_bounds__(256) void bn_act_bwd_stats_kernel(int rows, int C, const bf16_t* __restrict__ dz, const bf16_t* __restrict__ saved,
                                                               const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, int act, float p,
                                                               const uint64_t* seed_base, uint64_t seed_off, float* __restrict__ ws, int rpc,
                                                               int Tn, const int32_t* __restrict__ vlens) {
  __shared__ float sh[2][4][CT];
  const int v = threadIdx.x & 7, r8 = threadIdx.x >> 3;
  const int c = blockIdx.x * CT + v * 8;
  const int r0 = blockIdx.y * rpc;
  const int r1 = r0 + rpc < rows ? r0 + rpc : rows;
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float a0[8], a1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) a0[e] = a1[e] = 0.f;
  if (c < C) {
    float m[8], rs[8];
    ldp8(mean + c, m);
    ldp8(rstd + c, rs);
#pragma unroll 2
    for (int r = r0 + r8; r < r1; r += 32) {
      if (!row_present(r, Tn, vlens)) continue;
      const int64_t o = (int64_t)r * C + c;
      float g[8], sd[8], xx[8], mk[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(dz + o), g);
      if (saved) unpack_bf16x8(*reinterpret_cast<const uint4*>(saved + o), sd);
      unpack_bf16x8(*reinterpret_cast<const uint4*>(x + o), xx);
      if (p > 0.f) dropout_scale8(seed, (uint64_t)o, p, inv_keep, mk);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float mm = p > 0.f ? mk[e] : 1.f;
        const float gg = g[e] * mm * (saved ? act_grad_saved(sd[e], mm, act) : 1.f);
        a0[e] += gg;
        a1[e] += gg * (xx[e] - m[e]) * rs[e];
      }
    }
  }
  bn_chunk_reduce(a0, a1, sh, C, c, ws);
}

// dx = gamma * rstd * (g - sum_g / N - xhat * sum_g_xhat / N), g recomputed as above
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(int rows, int C, const bf16_t* __restrict__ dz,
                                                               const bf16_t* __restrict__ saved, const bf16_t* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ sdy,
                                                               const float* __restrict__ sdyx, int act, float p,
                                                               const uint64_t* seed_base, uint64_t seed_off, bf16_t* __restrict__ dx,
                                                               int rows_per_wg, int Tn, const int32_t* __restrict__ vlens) {
  const int v = threadIdx.x & 7, r8 = threadIdx.x >> 3;
  const int c = blockIdx.x * CT + v * 8;
  if (c >= C) return;
  const float inv_n = 1.0f / (float)rows_present(rows, Tn, vlens);
  const int r0 = blockIdx.y * rows_per_wg;
  const int r1 = r0 + rows_per_wg < rows ? r0 + rows_per_wg : rows;
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float m[8], rs[8], k1[8], k2[8], k3[8];
  ldp8(mean + c, m);
  ldp8(rstd + c, rs);
  ldp8(gamma + c, k1);
  ldp8(sdy + c, k2);
  ldp8(sdyx + c, k3);
#pragma unroll
  for (int e = 0; e < 8; ++e) { k1[e] *= rs[e]; k2[e] *= inv_n; k3[e] *= inv_n; }
#pragma unroll 2
  for (int r = r0 + r8; r < r1; r += 32) {
    const int64_t o = (int64_t)r * C + c;
    float g[8], sd[8], xx[8], mk[8];
    if (!row_present(r, Tn, vlens)) {               // absent frame: no gradient leaves it
      *reinterpret_cast<uint4*>(dx + o) = make_uint4(0, 0, 0, 0);
      continue;
    }
    unpack_bf16x8(*reinterpret_cast<const uint4*>(dz + o), g);
    if (saved) unpack_bf16x8(*reinterpret_cast<const uint4*>(saved + o), sd);
    unpack_bf16x8(*reinterpret_cast<const uint4*>(x + o), xx);
    if (p > 0.f) dropout_scale8(seed, (uint64_t)o, p, inv_keep, mk);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float mm = p > 0.f ? mk[e] : 1.f;
      const float gg = g[e] * mm * (saved ? act_grad_saved(sd[e], mm, act) : 1.f);
      g[e] = k1[e] * (gg - k2[e] - (xx[e] - m[e]) * rs[e] * k3[e]);
    }
    *reinterpret_cast<uint4*>(dx + o) = pack_bf16x8(g);
  }
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
// rows per statistics chunk: 64 for short inputs, more (a multiple of 32) so that there are at most 64 chunk partials to sum
int bn_rows_per_chunk(int rows) {
  int rpc = ((rows + 63) / 64 + 31) / 32 * 32;
  return rpc < 64 ? 64 : rpc;
}

}  // namespace

extern "C" int s2svc_convmod_supported(int C, int ks) { return (C > 0 && C % 64 == 0 && (ks == 7 || ks == 15 || ks == 31)) ? 1 : 0; }

extern "C" int s2svc_convmod_fwd(int B, int Tn, int C, int ks, const void* y2, const float* w, const float* bias, void* z, float eps,
                                 float momentum, float* mean, float* rstd, float* run_mean, float* run_var, int64_t* num_batches,
                                 float* ws, const int32_t* vlens, void* stream) {
  S2S_REQUIRE(B > 0 && Tn > 0 && s2svc_convmod_supported(C, ks), "convmod_fwd: C must be a multiple of 64, kernel size 7 / 15 / 31");
  S2S_REQUIRE(y2 && w && z && mean && rstd && ws && aligned16(y2) && aligned16(z), "convmod_fwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int tchunks = (Tn + TT - 1) / TT;
  dim3 grid(C / CT, B * tchunks);
#define S2S_CM_FWD(K)                                                                                                           \
  hipLaunchKernelGGL(convmod_fwd_kernel<K>, grid, dim3(256), 0, st, Tn, C, (const bf16_t*)y2, w, bias, (bf16_t*)z, ws, tchunks, vlens)
  if (ks == 7) S2S_CM_FWD(7);
  else if (ks == 15) S2S_CM_FWD(15);
  else S2S_CM_FWD(31);
#undef S2S_CM_FWD
  S2S_CHECK_LAUNCH("convmod_fwd_kernel");
  hipLaunchKernelGGL(convmod_bn_finalize_kernel, dim3(C / 64), dim3(256), 0, st, C, B * tchunks, ws, B * Tn, eps, momentum, mean, rstd,
                     run_mean, run_var, num_batches, Tn, vlens);
  S2S_CHECK_LAUNCH("convmod_bn_finalize_kernel");
  return 0;
}

extern "C" int s2svc_bn_swish_apply(int64_t rows, int C, const void* z, const float* mean, const float* rstd, const float* gamma,
                                    const float* beta, void* out, int Tn, const int32_t* vlens, void* stream) {
  S2S_REQUIRE(rows > 0 && C > 0 && C % 64 == 0 && z && mean && rstd && gamma && beta && out && aligned16(z) && aligned16(out),
              "bn_swish_apply: bad args (bf16, C % 64 == 0)");
  S2S_REQUIRE(!vlens || (Tn > 0 && rows % Tn == 0), "bn_swish_apply: vlens needs rows = B * Tn");
  const int rpw = 128;
  hipLaunchKernelGGL(bn_swish_apply_kernel, dim3(C / CT, (int)((rows + rpw - 1) / rpw)), dim3(256), 0, (hipStream_t)stream, (int)rows, C,
                     (const bf16_t*)z, mean, rstd, gamma, beta, (bf16_t*)out, rpw, Tn, vlens);
  S2S_CHECK_LAUNCH("bn_swish_apply_kernel");
  return 0;
}

// ws_stats >= ceil(B*Tn / 64) * 2 * C floats, ws_w >= B * ceil(Tn / 64) * C * (ks + 1) floats
extern "C" int s2svc_convmod_bwd(int B, int Tn, int C, int ks, const void* da, const void* z, const void* y2, const float* w,
                                 const float* mean, const float* rstd, const float* gamma, const float* beta, void* dy2, float* sdy,
                                 float* sdyx, float* dgamma_acc, float* dbeta_acc, float* ws_stats, float* ws_w, const int32_t* vlens,
                                 void* stream) {
  S2S_REQUIRE(B > 0 && Tn > 0 && s2svc_convmod_supported(C, ks), "convmod_bwd: C must be a multiple of 64, kernel size 7 / 15 / 31");
  S2S_REQUIRE(da && z && y2 && w && mean && rstd && gamma && beta && dy2 && sdy && sdyx && ws_stats && ws_w && aligned16(da) &&
              aligned16(z) && aligned16(y2) && aligned16(dy2), "convmod_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int rows = B * Tn, chunks = (rows + 63) / 64;
  hipLaunchKernelGGL(convmod_bwd_stats_kernel, dim3(C / CT, chunks), dim3(256), 0, st, rows, C, (const bf16_t*)da, (const bf16_t*)z, mean,
                     rstd, gamma, beta, ws_stats, Tn, vlens);
  S2S_CHECK_LAUNCH("convmod_bwd_stats_kernel");
  hipLaunchKernelGGL(convmod_sum2_kernel, dim3(C / 64), dim3(256), 0, st, C, chunks, ws_stats, sdy, sdyx, dbeta_acc, dgamma_acc);
  S2S_CHECK_LAUNCH("convmod_sum2_kernel");
  const int tchunks = (Tn + TT - 1) / TT;
  dim3 grid(C / CT, B * tchunks);
#define S2S_CM_BWD(K)                                                                                                              \
  hipLaunchKernelGGL(convmod_bwd_kernel<K>, grid, dim3(256), 0, st, Tn, C, (const bf16_t*)da, (const bf16_t*)z,                    \
                     (const bf16_t*)y2, w, mean, rstd, gamma, beta, sdy, sdyx, (bf16_t*)dy2, ws_w, tchunks, vlens)
  if (ks == 7) S2S_CM_BWD(7);
  else if (ks == 15) S2S_CM_BWD(15);
  else S2S_CM_BWD(31);
#undef S2S_CM_BWD
  S2S_CHECK_LAUNCH("convmod_bwd_kernel");
  return 0;
}

extern "C" int s2svc_convmod_wgrad_final(int C, int ks, int chunks, const float* ws_w, float* dw, float* db, int accumulate, void* stream) {
  S2S_REQUIRE(C > 0 && ks > 0 && chunks > 0 && ws_w, "convmod_wgrad_final: bad args");
  const int n = C * (ks + 1);
  hipLaunchKernelGGL(convmod_wgrad_final_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, ks, chunks, ws_w, dw, db,
                     accumulate);
  S2S_CHECK_LAUNCH("convmod_wgrad_final_kernel");
  return 0;
}

// ---- BatchNorm1d (training) + activation + dropout, bf16, C % 8 == 0 (see above) ----
// mean / rstd of x (rows, C) + running statistics: two launches.  ws >= ceil(rows / 64) * 2 * C floats.
extern "C" int s2svc_bn_stats_vec(int rows, int C, const void* x, float eps, float momentum, float* mean, float* rstd, float* run_mean,
                                  float* run_var, int64_t* num_batches, float* ws, int Tn, const int32_t* vlens, void* stream) {
  S2S_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && x && mean && rstd && ws && aligned16(x), "bn_stats_vec: bad args (bf16, C % 8 == 0)");
  S2S_REQUIRE(!vlens || (Tn > 0 && rows % Tn == 0), "bn_stats_vec: vlens needs rows = B * Tn");
  hipStream_t st = (hipStream_t)stream;
  const int rpc = bn_rows_per_chunk(rows), chunks = (rows + rpc - 1) / rpc, ct = (C + CT - 1) / CT;
  hipLaunchKernelGGL(bn_stats_vec_kernel, dim3(ct, chunks), dim3(256), 0, st, rows, C, (const bf16_t*)x, ws, rpc, Tn, vlens);
  S2S_CHECK_LAUNCH("bn_stats_vec_kernel");
  hipLaunchKernelGGL(convmod_bn_finalize_kernel, dim3(ct), dim3(256), 0, st, C, chunks, ws, rows, eps, momentum, mean, rstd, run_mean,
                     run_var, num_batches, Tn, vlens);
  S2S_CHECK_LAUNCH("convmod_bn_finalize_kernel");
  return 0;
}

extern "C" int s2svc_bn_act_apply_vec(int rows, int C, const void* x, const float* mean, const float* rstd, const float* gamma,
                                      const float* beta, int act, float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* y,
                                      void* pre_act, int Tn, const int32_t* vlens, void* stream) {
  S2S_REQUIRE(!vlens || (Tn > 0 && rows % Tn == 0), "bn_act_apply_vec: vlens needs rows = B * Tn");
  S2S_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && x && mean && rstd && gamma && beta && y && aligned16(x) && aligned16(y) &&
              aligned16(pre_act), "bn_act_apply_vec: bad args (bf16, C % 8 == 0)");
  const int rpw = 128;
  hipLaunchKernelGGL(bn_act_apply_vec_kernel, dim3((C + CT - 1) / CT, (rows + rpw - 1) / rpw), dim3(256), 0, (hipStream_t)stream, rows, C,
                     (const bf16_t*)x, mean, rstd, gamma, beta, act, drop_p, seed_base, seed_off, (bf16_t*)y, (bf16_t*)pre_act, rpw, Tn,
                     vlens);
  S2S_CHECK_LAUNCH("bn_act_apply_vec_kernel");
  return 0;
}

// dz = gradient of y = dropout(act(BN(x))); saved = y (relu / tanh / sigmoid) or the pre-activation (swish / gelu), NULL for act none
// without dropout.  -> dx, sdy = sum g, sdyx = sum g * xhat (the gradients of beta / gamma, also ADDED to dbeta_acc / dgamma_acc when
// given).  Three launches.  ws >= ceil(rows / 64) * 2 * C floats.
extern "C" int s2svc_bn_act_bwd_vec(int rows, int C, const void* dz, const void* saved, const void* x, const float* mean, const float* rstd,
                                    const float* gamma, int act, float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* dx,
                                    float* sdy, float* sdyx, float* dgamma_acc, float* dbeta_acc, float* ws, int Tn, const int32_t* vlens,
                                    void* stream) {
  S2S_REQUIRE(!vlens || (Tn > 0 && rows % Tn == 0), "bn_act_bwd_vec: vlens needs rows = B * Tn");
  S2S_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && dz && x && mean && rstd && gamma && dx && sdy && sdyx && ws && aligned16(dz) &&
              aligned16(saved) && aligned16(x) && aligned16(dx) && (saved || act == S2S_ACT_NONE), "bn_act_bwd_vec: bad args (bf16, C % 8 == 0)");
  hipStream_t st = (hipStream_t)stream;
  const int rpc = bn_rows_per_chunk(rows), chunks = (rows + rpc - 1) / rpc, ct = (C + CT - 1) / CT;
  hipLaunchKernelGGL(bn_act_bwd_stats_kernel, dim3(ct, chunks), dim3(256), 0, st, rows, C, (const bf16_t*)dz, (const bf16_t*)saved,
                     (const bf16_t*)x, mean, rstd, act, drop_p, seed_base, seed_off, ws, rpc, Tn, vlens);
  S2S_CHECK_LAUNCH("bn_act_bwd_stats_kernel");
  hipLaunchKernelGGL(convmod_sum2_kernel, dim3(ct), dim3(256), 0, st, C, chunks, ws, sdy, sdyx, dbeta_acc, dgamma_acc);
  S2S_CHECK_LAUNCH("convmod_sum2_kernel");
  const int rpw = 128;
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(ct, (rows + rpw - 1) / rpw), dim3(256), 0, st, rows, C,
                     (const bf16_t*)dz, (const bf16_t*)saved, (const bf16_t*)x, mean, rstd, gamma, sdy, sdyx, act, drop_p, seed_base,
                     seed_off, (bf16_t*)dx, rpw, Tn, vlens);
  S2S_CHECK_LAUNCH("bn_act_bwd_apply_kernel");
  return 0;
}
