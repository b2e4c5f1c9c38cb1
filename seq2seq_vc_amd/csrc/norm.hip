// LayerNorm (fused with residual-add + dropout), BatchNorm1d over (B*T) rows, and the chunked
// deterministic column reductions they and the bias gradients share.  All HBM-bound: one read of
// each input, one write of each output, statistics in fp32 registers / wave shuffles.
#include <cstring>
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// LayerNorm forward: one wavefront per row.
//   s = res ? res + dropout(x) : x ;  y = (s - mean) * rstd * gamma + beta
// reference: modules/transformer/layer_norm.py:12-42 (eps 1e-12) and the residual/dropout lines of
// encoder_layer.py:96-113, decoder_layer.py:104-127, conformer/encoder_layer.py:118-170.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(int rows, int D, const T* __restrict__ x, const T* __restrict__ res,
                                                     float p, float hscale, const uint64_t* seed_base, uint64_t seed_off, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float eps, T* __restrict__ y,
                                                     T* __restrict__ s_out, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t base = (int64_t)row * D;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float sum = 0.f;
  for (int c = lane; c < D; c += 64) {
    float v = ldf(x + base + c);
    if (res) {
      if (p > 0.f) v *= dropout_scale(seed, (uint64_t)(base + c), p, inv_keep);
      v *= hscale;
      v += ldf(res + base + c);
      stf(s_out + base + c, v);
      v = ldf(s_out + base + c);  // statistics on the stored (rounded) value
    }
    sum += v;
  }
  const float mean = wave_sum(sum) / (float)D;
  const T* s = res ? s_out : x;
  float sq = 0.f;
  for (int c = lane; c < D; c += 64) {
    float d = ldf(s + base + c) - mean;
    sq += d * d;
  }
  const float var = wave_sum(sq) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
  for (int c = lane; c < D; c += 64) {
    float v = (ldf(s + base + c) - mean) * rstd;
    stf(y + base + c, v * gamma[c] + beta[c]);
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// Same contract, rows held in registers (D <= 64*NV): one global read of x / res, one write of s and y -- no
// store -> reload round trips, which dominate the latency of this kernel on short problems.
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16_t>(float v) { return bf2f(f2bf(v)); }

template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_reg_kernel(int rows, int D, const T* __restrict__ x, const T* __restrict__ res,
                                                         float p, float hscale, const uint64_t* seed_base, uint64_t seed_off,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         T* __restrict__ y, T* __restrict__ s_out, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t base = (int64_t)row * D;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float v[NV], r[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < D ? ldf(x + base + c) : 0.f;
    r[i] = (res && c < D) ? ldf(res + base + c) : 0.f;
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (res && c < D) {
      float t = v[i];
      if (p > 0.f) t *= dropout_scale(seed, (uint64_t)(base + c), p, inv_keep);
      t = t * hscale + r[i];
      stf(s_out + base + c, t);
      v[i] = round_to<T>(t);     // statistics on the stored (rounded) value
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
  const float var = wave_sum(sq) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + 64 * i;
    if (c < D) stf(y + base + c, (v[i] - mean) * rstd * gamma[c] + beta[c]);
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// LayerNorm backward wrt the normalised input s (+ fused residual/dropout split):
//   g = dy*gamma ; ds = rstd*(g - mean(g) - xhat*mean(g*xhat)) + ds_extra
//   dres = ds ; dh = ds * dropmask
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(int rows, int D, const T* __restrict__ dy, const T* __restrict__ s,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const T* __restrict__ ds_extra,
                                                     float p, float hscale, const uint64_t* seed_base, uint64_t seed_off, T* __restrict__ ds, T* __restrict__ dh) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t base = (int64_t)row * D;
  const float mu = mean[row], rs = rstd[row];
  float a = 0.f, b = 0.f;
  for (int c = lane; c < D; c += 64) {
    float g = ldf(dy + base + c) * gamma[c];
    float xh = (ldf(s + base + c) - mu) * rs;
    a += g;
    b += g * xh;
  }
  a = wave_sum(a) / (float)D;
  b = wave_sum(b) / (float)D;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int c = lane; c < D; c += 64) {
    float g = ldf(dy + base + c) * gamma[c];
    float xh = (ldf(s + base + c) - mu) * rs;
    float v = rs * (g - a - xh * b);
    if (ds_extra) v += ldf(ds_extra + base + c);
    stf(ds + base + c, v);
    if (dh) {
      float m = p > 0.f ? dropout_scale(seed, (uint64_t)(base + c), p, inv_keep) : 1.f;
      stf(dh + base + c, v * m * hscale);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// bf16, D % 8 == 0, D <= 512*NV (NV = 1, 2, 4): a lane owns NV groups of 8 CONSECUTIVE channels -- every tensor moves in 16-byte
// loads / stores (the lane + 64*i mapping above issues one 2-byte access per element: 6 per lane and tensor at
// D = 384).  Same arithmetic, same summation order inside a lane group; the wave reduction is the same butterfly.
// ------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_vec_kernel(int rows, int D, const bf16_t* __restrict__ x, const bf16_t* __restrict__ res,
                                                         float p, float hscale, const uint64_t* seed_base, uint64_t seed_off,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         bf16_t* __restrict__ y, bf16_t* __restrict__ s_out, float* __restrict__ mean_out,
                                                         float* __restrict__ rstd_out) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int64_t base = (int64_t)row * D;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float v[NV][8];
  float sum = 0.f;
  // gamma / beta are needed after the two row reductions only, but their loads are issued with the row's: nothing hides a second
  // memory round trip in a kernel whose whole life is one
  float gm[NV][8], bt[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < D) { load_f32x8(gamma + c, gm[i]); load_f32x8(beta + c, bt[i]); }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < D) {
      unpack_bf16x8(*reinterpret_cast<const uint4*>(x + base + c), v[i]);
      if (res) {
        float r[8], m[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(res + base + c), r);
        if (p > 0.f) dropout_scale8(seed, (uint64_t)(base + c), p, inv_keep, m);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float t = v[i][e];
          if (p > 0.f) t *= m[e];
          v[i][e] = t * hscale + r[e];
        }
        const uint4 sv = pack_bf16x8(v[i]);
        *reinterpret_cast<uint4*>(s_out + base + c) = sv;
        unpack_bf16x8(sv, v[i]);               // statistics on the stored (rounded) value
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if ((lane + 64 * i) * 8 < D) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    }
  const float var = wave_sum(sq) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < D) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * gm[i][e] + bt[i][e];
      *reinterpret_cast<uint4*>(y + base + c) = pack_bf16x8(o);
    }
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// one row of the vectorised LayerNorm backward pass (a wave per row, a lane owns 8 consecutive channels per 512-channel group);
// PG: also adds the row's contribution to the parameter gradients (d beta += dy, d gamma += dy * xhat) into the lane's accumulators
template <int NV, bool PG>
__device__ __forceinline__ void ln_bwd_vec_row(int row, int D, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ s,
                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                               const float* __restrict__ gamma, const bf16_t* __restrict__ ds_extra, float p, float hscale,
                                               uint64_t seed, bf16_t* __restrict__ ds, bf16_t* __restrict__ dh, float (&accb)[NV][8],
                                               float (&accg)[NV][8]) {
  const int lane = threadIdx.x & 63;
  const int64_t base = (int64_t)row * D;
  const float mu = mean[row], rs = rstd[row];
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float g[NV][8], xh[NV][8];
  float a = 0.f, b = 0.f;
  // the pending residual-stream gradient is only needed after the row statistics, but its load is issued HERE, with the others:
  // behind the two wave reductions it was a second exposed memory round trip (~1.5 us of a 7 us kernel on cold caches)
  uint4 exv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 8;
    exv[i] = make_uint4(0, 0, 0, 0);
    if (ds_extra && c < D) exv[i] = *reinterpret_cast<const uint4*>(ds_extra + base + c);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < D) {
      float gm[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + base + c), g[i]);
      unpack_bf16x8(*reinterpret_cast<const uint4*>(s + base + c), xh[i]);
      load_f32x8(gamma + c, gm);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        xh[i][e] = (xh[i][e] - mu) * rs;
        if (PG) {
          accb[i][e] += g[i][e];
          accg[i][e] += g[i][e] * xh[i][e];
        }
        g[i][e] *= gm[e];
        a += g[i][e];
        b += g[i][e] * xh[i][e];
      }
    }
  }
  a = wave_sum(a) / (float)D;
  b = wave_sum(b) / (float)D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 8;
    if (c < D) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = rs * (g[i][e] - a - xh[i][e] * b);
      if (ds_extra) {
        float ex[8];
        unpack_bf16x8(exv[i], ex);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += ex[e];
      }
      *reinterpret_cast<uint4*>(ds + base + c) = pack_bf16x8(v);
      if (dh) {
        float m[8];
        if (p > 0.f) dropout_scale8(seed, (uint64_t)(base + c), p, inv_keep, m);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * (p > 0.f ? m[e] : 1.f) * hscale;
        *reinterpret_cast<uint4*>(dh + base + c) = pack_bf16x8(v);
      }
    }
  }
}

template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_vec_kernel(int rows, int D, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ s,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const bf16_t* __restrict__ ds_extra,
                                                         float p, float hscale, const uint64_t* seed_base, uint64_t seed_off,
                                                         bf16_t* __restrict__ ds, bf16_t* __restrict__ dh) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float nb[NV][8], ng[NV][8];               // (unused without PG)
  ln_bwd_vec_row<NV, false>(row, D, dy, s, mean, rstd, gamma, ds_extra, p, hscale, seed, ds, dh, nb, ng);
}

// The same backward pass WITH the parameter gradients' first reduction stage (round 4): a block of 8 waves owns 8 * rpw consecutive rows
// (a wave rpw of them, one after the other), every lane adds dy and dy * xhat of its channels over its rows in registers, the eight
// waves' sums are combined through LDS in wave order and the block writes ONE partial row pair ws[block][2][D] -- the layout of
// colreduce_stage1's chunk partials, so the (grouped) second stage finishes the job (colreduce mode 7 = "partials ready").  The separate
// first stage re-read dy and the LayerNorm input: 25 MB per 4096 x 1536 site, 0.29 ms of the AAS-VC step (timing-only build).
template <int NV>
__global__ __launch_bounds__(512) void ln_bwd_vec_pg_kernel(int rows, int D, int rpw, const bf16_t* __restrict__ dy, const bf16_t* __restrict__ s,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, const bf16_t* __restrict__ ds_extra,
                                                            float p, float hscale, const uint64_t* seed_base, uint64_t seed_off,
                                                            bf16_t* __restrict__ ds, bf16_t* __restrict__ dh, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) float pg_sh[];      // [8 waves][2][Dp], Dp = NV * 512
  constexpr int Dp = NV * 512;
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float accb[NV][8], accg[NV][8];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) accb[i][e] = accg[i][e] = 0.f;
  const int row0 = ((int)blockIdx.x * 8 + wave) * rpw;
  for (int j = 0; j < rpw; ++j)
    if (row0 + j < rows) ln_bwd_vec_row<NV, true>(row0 + j, D, dy, s, mean, rstd, gamma, ds_extra, p, hscale, seed, ds, dh, accb, accg);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (lane + 64 * i) * 8;
    float* qb = pg_sh + (wave * 2 + 0) * Dp + c;
    float* qg = pg_sh + (wave * 2 + 1) * Dp + c;
    *reinterpret_cast<float4*>(qb) = make_float4(accb[i][0], accb[i][1], accb[i][2], accb[i][3]);
    *reinterpret_cast<float4*>(qb + 4) = make_float4(accb[i][4], accb[i][5], accb[i][6], accb[i][7]);
    *reinterpret_cast<float4*>(qg) = make_float4(accg[i][0], accg[i][1], accg[i][2], accg[i][3]);
    *reinterpret_cast<float4*>(qg + 4) = make_float4(accg[i][4], accg[i][5], accg[i][6], accg[i][7]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 512) {
    float tb = pg_sh[c], tg = pg_sh[Dp + c];
#pragma unroll
    for (int w = 1; w < 8; ++w) {            // wave order: fixed
      tb += pg_sh[(w * 2 + 0) * Dp + c];
      tg += pg_sh[(w * 2 + 1) * Dp + c];
    }
    ws[((int64_t)blockIdx.x * 2 + 0) * D + c] = tb;
    ws[((int64_t)blockIdx.x * 2 + 1) * D + c] = tg;
  }
}

// ------------------------------------------------------------------------------------------------
// Chunked column reduction over rows of a (rows, D) matrix, deterministic (fixed summation order):
//   mode 0: sum[c] = S dy[r,c]
//   mode 1: sum[c] = S dy[r,c];  dot[c] = S dy[r,c]*(x[r,c]-mean[r])*rstd[r]     (LayerNorm dgamma/dbeta)
//   mode 2: sum[c] = S dy[r,c];  dot[c] = S dy[r,c]*(x[r,c]-mean[c])*rstd[c]     (BatchNorm dgamma/dbeta)
//   mode 3: sum[c] = S (x[r,c]-mean[c])^2                                        (BatchNorm variance)
//   mode 4: sum[c] = S dy[r,c]*x[r,c]                                            (posenc alpha etc.)
//   mode 6: sum[c] = S x[r,c];  dot[c] = S x[r,c]^2                              (BatchNorm mean and E[x^2] in ONE pass)
//   mode 7 (grouped launch only): the chunk partials are already in ws[ws_chunks][2][D] (ln_bwd_vec_pg_kernel): stage 2 only
// stage 1 writes ws[chunk][2][D]; stage 2 sums the chunks and multiplies by `scale`.
// ------------------------------------------------------------------------------------------------
// MASKED: rows r = b * Tn + t with t >= vlens[b] are absent (common.h) and skipped
template <typename T, bool MASKED = false>
__device__ __forceinline__ void colreduce_stage1_body(int rows, int D, int mode, const T* __restrict__ dy,
                                                      const T* __restrict__ x, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, float* __restrict__ ws,
                                                      int rows_per_chunk, int bx, int chunk, float (&sh)[2][4][64], int Tn = 0,
                                                      const int32_t* __restrict__ vlens = nullptr) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = bx * 64 + cl;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = (r0 + rows_per_chunk < rows) ? r0 + rows_per_chunk : rows;
  float s0 = 0.f, s1 = 0.f;
  if (c < D) {
    float mc = 0.f, rc = 1.f;
    if (mode == 2 || mode == 3) { mc = mean[c]; rc = rstd ? rstd[c] : 1.f; }
    for (int r = r0 + rl; r < r1; r += 4) {
      if (MASKED && !row_present(r, Tn, vlens)) continue;
      const int64_t o = (int64_t)r * D + c;
      if (mode == 0) {
        s0 += ldf(dy + o);
      } else if (mode == 1) {
        float g = ldf(dy + o);
        s0 += g;
        s1 += g * (ldf(x + o) - mean[r]) * rstd[r];
      } else if (mode == 2) {
        float g = ldf(dy + o);
        s0 += g;
        s1 += g * (ldf(x + o) - mc) * rc;
      } else if (mode == 3) {
        float dv = ldf(x + o) - mc;
        s0 += dv * dv;
      } else if (mode == 6) {          // one-pass moments: s0 = sum x ; s1 = sum x^2
        const float v = ldf(x + o);
        s0 += v;
        s1 += v * v;
      } else if (mode == 5) {          // s0 = sum dy ; s1 = sum dy * v[r]  (v handed in through `mean`)
        float g = ldf(dy + o);
        s0 += g;
        s1 += g * mean[r];
      } else {
        s0 += ldf(dy + o) * ldf(x + o);
      }
    }
  }
  sh[0][rl][cl] = s0;
  sh[1][rl][cl] = s1;
  __syncthreads();
  if (rl == 0 && c < D) {
    float t0 = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
    float t1 = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
    ws[((int64_t)chunk * 2 + 0) * D + c] = t0;
    ws[((int64_t)chunk * 2 + 1) * D + c] = t1;
  }
}

// 16-byte variant (bf16, D % 8 == 0, 16-byte aligned tensors): a lane owns 8 consecutive columns, a wave covers 512 columns
// of a row with one load per tensor, four rows of a row group in flight.  The scalar body moves 2 bytes per lane and load
// (2.7 TB/s over the LayerNorm / bias reductions of an AAS-VC decoder layer: 150 MB in 55 us).  Same rows per wave, same order
// of additions per column, same combination of the four waves as the scalar body.
__device__ __forceinline__ void colreduce_stage1_vec_body(int rows, int D, int mode, const bf16_t* __restrict__ dy,
                                                          const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ ws,
                                                          int rows_per_chunk, int bx, int chunk, float (&sh)[2][4][512]) {
  const int lane = threadIdx.x & 63;
  const int rl = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int c = bx * 512 + lane * 8;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = (r0 + rows_per_chunk < rows) ? r0 + rows_per_chunk : rows;
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
  if (c < D) {
    float mc[8], rc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { mc[e] = 0.f; rc[e] = 1.f; }
    if (mode == 2 || mode == 3) {
      load_f32x8(mean + c, mc);
      if (rstd) load_f32x8(rstd + c, rc);
    }
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 4) {
      const int64_t o = (int64_t)r * D + c;
      float g[8], v[8];
      if (mode != 3 && mode != 6) unpack_bf16x8(*reinterpret_cast<const uint4*>(dy + o), g);
      if (mode != 0 && mode != 5) unpack_bf16x8(*reinterpret_cast<const uint4*>(x + o), v);
      if (mode == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s0[e] += g[e];
      } else if (mode == 1) {
        const float mr = mean[r], rr = rstd[r];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[e] += g[e]; s1[e] += g[e] * (v[e] - mr) * rr; }
      } else if (mode == 2) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[e] += g[e]; s1[e] += g[e] * (v[e] - mc[e]) * rc[e]; }
      } else if (mode == 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float dv = v[e] - mc[e]; s0[e] += dv * dv; }
      } else if (mode == 6) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[e] += v[e]; s1[e] += v[e] * v[e]; }
      } else if (mode == 5) {
        const float mr = mean[r];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[e] += g[e]; s1[e] += g[e] * mr; }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) s0[e] += g[e] * v[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sh[0][rl][lane * 8 + e] = s0[e];
    sh[1][rl][lane * 8 + e] = s1[e];
  }
  __syncthreads();
  // 512 columns x 2 sums: thread t combines column t (sum 0) and column t (sum 1) of its half
  for (int i = threadIdx.x; i < 1024; i += 256) {
    const int which = i >> 9, col = i & 511;
    if (bx * 512 + col < D)
      ws[((int64_t)chunk * 2 + which) * D + bx * 512 + col] =
          sh[which][0][col] + sh[which][1][col] + sh[which][2][col] + sh[which][3][col];
  }
}

// wide rows only: at D = 384 / 512 (VTN) one workgroup per 512 columns leaves too few workgroups (4.00 -> 4.02 ms per step),
// at D = 1536 / 3072 (AAS-VC decoder) the 16-byte loads win (12.39 -> 12.30 ms)
__device__ __forceinline__ bool cr_vec_ok(int dtype, int D, const void* dy, const void* x) {
  return dtype == S2S_BF16 && D >= 768 && (D & 7) == 0 && ((((uintptr_t)dy) | ((uintptr_t)x)) & 15) == 0;
}
static bool cr_vec_ok_host(int dtype, int D, const void* dy, const void* x) {
  return dtype == S2S_BF16 && D >= 768 && (D & 7) == 0 && ((((uintptr_t)dy) | ((uintptr_t)x)) & 15) == 0;
}

template <typename T>
__global__ __launch_bounds__(256) void colreduce_stage1(int rows, int D, int mode, const T* __restrict__ dy,
                                                        const T* __restrict__ x, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, float* __restrict__ ws,
                                                        int rows_per_chunk) {
  __shared__ float sh[2][4][64];
  colreduce_stage1_body<T>(rows, D, mode, dy, x, mean, rstd, ws, rows_per_chunk, blockIdx.x, blockIdx.y, sh);
}

template <typename T>
__global__ __launch_bounds__(256) void colreduce_stage1_masked(int rows, int D, int mode, const T* __restrict__ dy,
                                                               const T* __restrict__ x, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ ws,
                                                               int rows_per_chunk, int Tn, const int32_t* __restrict__ vlens) {
  __shared__ float sh[2][4][64];
  colreduce_stage1_body<T, true>(rows, D, mode, dy, x, mean, rstd, ws, rows_per_chunk, blockIdx.x, blockIdx.y, sh, Tn, vlens);
}

__global__ __launch_bounds__(256) void colreduce_stage1_vec(int rows, int D, int mode, const bf16_t* __restrict__ dy,
                                                            const bf16_t* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ ws,
                                                            int rows_per_chunk) {
  __shared__ float shv[2][4][512];
  colreduce_stage1_vec_body(rows, D, mode, dy, x, mean, rstd, ws, rows_per_chunk, blockIdx.x, blockIdx.y, shv);
}

// 64 columns x 4 chunk-groups per workgroup: each thread sums every 4th chunk partial (independent loads), the four
// groups are combined through LDS in a fixed order
__device__ __forceinline__ void colreduce_stage2_body(int D, int chunks, const float* __restrict__ ws, float scale,
                                                      float* __restrict__ out_sum, float* __restrict__ out_dot, int accumulate, int bx,
                                                      float (&sh)[2][4][64]) {
  const int cl = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const int c = bx * 64 + cl;
  float t0 = 0.f, t1 = 0.f;
  if (c < D) {
#pragma unroll 8
    for (int k = kg; k < chunks; k += 4) {        // same order of additions, 16 loads in flight
      t0 += ws[((int64_t)k * 2 + 0) * D + c];
      t1 += ws[((int64_t)k * 2 + 1) * D + c];
    }
  }
  sh[0][kg][cl] = t0;
  sh[1][kg][cl] = t1;
  __syncthreads();
  if (kg == 0 && c < D) {
    t0 = ((sh[0][0][cl] + sh[0][1][cl]) + sh[0][2][cl]) + sh[0][3][cl];
    t1 = ((sh[1][0][cl] + sh[1][1][cl]) + sh[1][2][cl]) + sh[1][3][cl];
    if (out_sum) out_sum[c] = (accumulate ? out_sum[c] : 0.f) + t0 * scale;
    if (out_dot) out_dot[c] = (accumulate ? out_dot[c] : 0.f) + t1 * scale;
  }
}

// scale < 0: the mean over the PRESENT rows (scale = 1 / their number, from rows / Tn / vlens)
__global__ __launch_bounds__(256) void colreduce_stage2(int D, int chunks, const float* __restrict__ ws, float scale,
                                                        float* __restrict__ out_sum, float* __restrict__ out_dot, int accumulate,
                                                        int rows, int Tn, const int32_t* __restrict__ vlens) {
  __shared__ float sh[2][4][64];
  if (scale < 0.f) scale = 1.0f / (float)rows_present(rows, Tn, vlens);
  colreduce_stage2_body(D, chunks, ws, scale, out_sum, out_dot, accumulate, blockIdx.x, sh);
}

// Grouped column reductions: up to S2S_CR_MAX independent reductions (the LayerNorm / bias / BatchNorm parameter
// gradients of a few consecutive layers) in TWO launches (stage 1, stage 2) instead of two per reduction.  Items travel by
// value in the kernel arguments; blockIdx.z selects the item, blocks outside its (column tiles, chunks) extent exit.
#define S2S_CR_MAX 24
struct cr_args {
  s2svc_colreduce_item it[S2S_CR_MAX];
  int32_t chunks[S2S_CR_MAX], rpc[S2S_CR_MAX];
  int32_t n;
};
static_assert(sizeof(cr_args) <= 4096, "kernel arguments are limited to 4 KB");

__global__ __launch_bounds__(256) void colreduce_grouped_stage1(const cr_args a) {
  __shared__ float sh[2][4][64];
  __shared__ float shv[2][4][512];
  const s2svc_colreduce_item& it = a.it[blockIdx.z];
  if (it.mode == 7 || (int)blockIdx.y >= a.chunks[blockIdx.z]) return;
  if (cr_vec_ok(it.dtype, it.D, it.dy, it.x)) {          // (uniform) 16-byte variant: 512 columns per workgroup
    if ((int)blockIdx.x * 512 >= it.D) return;
    colreduce_stage1_vec_body(it.rows, it.D, it.mode, (const bf16_t*)it.dy, (const bf16_t*)it.x, it.mean, it.rstd, it.ws,
                              a.rpc[blockIdx.z], blockIdx.x, blockIdx.y, shv);
    return;
  }
  if ((int)blockIdx.x * 64 >= it.D) return;
  if (it.dtype == S2S_F32)
    colreduce_stage1_body<float>(it.rows, it.D, it.mode, (const float*)it.dy, (const float*)it.x, it.mean, it.rstd, it.ws,
                                 a.rpc[blockIdx.z], blockIdx.x, blockIdx.y, sh);
  else
    colreduce_stage1_body<bf16_t>(it.rows, it.D, it.mode, (const bf16_t*)it.dy, (const bf16_t*)it.x, it.mean, it.rstd, it.ws,
                                  a.rpc[blockIdx.z], blockIdx.x, blockIdx.y, sh);
}
__global__ __launch_bounds__(256) void colreduce_grouped_stage2(const cr_args a) {
  __shared__ float sh[2][4][64];
  const s2svc_colreduce_item& it = a.it[blockIdx.y];
  if ((int)blockIdx.x * 64 >= it.D) return;
  colreduce_stage2_body(it.D, a.chunks[blockIdx.y], it.ws, it.scale, it.out_sum, it.out_dot, it.accumulate, blockIdx.x, sh);
}

// ------------------------------------------------------------------------------------------------
// BatchNorm1d apply (channel-last rows): y = dropout(act((x-mean[c])*rstd[c]*gamma[c]+beta[c]))
// reference: modules/pre_postnets.py:108-165 (Conv1d->BatchNorm1d->Tanh->Dropout) and
// modules/conformer/convolution.py:73-75 (BatchNorm1d -> Swish).  Padded frames are part of the
// statistics exactly as in the reference (no masking).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void bn_apply_kernel(int64_t total, int C, const T* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, int act, float p, const uint64_t* seed_base, uint64_t seed_off, T* __restrict__ y,
                                T* __restrict__ pre_act, int Tn, const int32_t* __restrict__ vlens) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    if (vlens && !row_present((int)(i / C), Tn, vlens)) {      // absent frame (common.h): the next convolution's zero padding
      if (pre_act) stf(pre_act + i, 0.f);
      stf(y + i, 0.f);
      continue;
    }
    float v = (ldf(x + i) - mean[c]) * rstd[c] * gamma[c] + beta[c];
    if (pre_act) stf(pre_act + i, v);
    v = act_apply(v, act);
    if (p > 0.f) v *= dropout_scale(seed, (uint64_t)i, p, inv_keep);
    stf(y + i, v);
  }
}

// dx = gamma*rstd*(dy - sum_dy/N - xhat*sum_dy_xhat/N)   (training-mode BatchNorm backward)
// eval mode (use_batch_stats = 0): dx = gamma*rstd*dy
template <typename T>
__global__ void bn_bwd_kernel(int64_t total, int C, float inv_n, const T* __restrict__ dy, const T* __restrict__ x,
                              const float* __restrict__ mean, const float* __restrict__ rstd,
                              const float* __restrict__ gamma, const float* __restrict__ sum_dy,
                              const float* __restrict__ sum_dy_xhat, int use_batch_stats, T* __restrict__ dx, int Tn,
                              const int32_t* __restrict__ vlens) {
  if (vlens) inv_n = 1.0f / (float)rows_present((int)(total / C), Tn, vlens);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    if (vlens && !row_present((int)(i / C), Tn, vlens)) {      // absent frame: no gradient leaves it
      stf(dx + i, 0.f);
      continue;
    }
    float g = ldf(dy + i);
    float v;
    if (use_batch_stats) {
      float xh = (ldf(x + i) - mean[c]) * rstd[c];
      v = gamma[c] * rstd[c] * (g - sum_dy[c] * inv_n - xh * sum_dy_xhat[c] * inv_n);
    } else {
      v = gamma[c] * rstd[c] * g;
    }
    stf(dx + i, v);
  }
}

// var (biased) -> rstd, and the running-statistics update of torch.nn.BatchNorm1d (momentum 0.1,
// unbiased variance in the running buffer).
__global__ void bn_finalize_kernel(int C, int n, float eps, float momentum, const float* __restrict__ mean,
                                   const float* __restrict__ var, float* __restrict__ rstd, float* __restrict__ run_mean,
                                   float* __restrict__ run_var, int64_t* __restrict__ num_batches, int var_is_ex2, int Tn,
                                   const int32_t* __restrict__ vlens) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  n = rows_present(n, Tn, vlens);
  if (c == 0 && num_batches) *num_batches += 1;
  if (c >= C) return;
  float vc = var[c];
  if (var_is_ex2) {                   // `var` holds E[x^2] (one-pass statistics): biased variance = E[x^2] - mean^2
    vc -= mean[c] * mean[c];
    vc = vc > 0.f ? vc : 0.f;
  }
  rstd[c] = 1.0f / sqrtf(vc + eps);
  if (run_mean) {
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean[c];
    const float unb = n > 1 ? vc * ((float)n / (float)(n - 1)) : vc;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * unb;
  }
}

// One-pass training statistics, second half: the stage-2 sum of the chunk partials (colreduce mode 6: sum x, sum x^2 -- same
// order of additions as colreduce_stage2) and bn_finalize_kernel's arithmetic in ONE launch.
__global__ __launch_bounds__(256) void bn_stage2_finalize_kernel(int C, int chunks, const float* __restrict__ ws, int n, float eps,
                                                                 float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                                 float* __restrict__ run_mean, float* __restrict__ run_var,
                                                                 int64_t* __restrict__ num_batches, int Tn,
                                                                 const int32_t* __restrict__ vlens) {
  __shared__ float sh[2][4][64];
  const int cl = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  n = rows_present(n, Tn, vlens);
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

__global__ void rstd_from_var_kernel(int C, float eps, const float* __restrict__ var, float* __restrict__ rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) rstd[c] = 1.0f / sqrtf(var[c] + eps);
}

inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

// the 16-byte LayerNorm kernels: bf16, D a multiple of 8 and at most 2048, every tensor 16-byte aligned
static bool ln_vec_ok(int D, const void* a, const void* b, const void* c, const void* d, const void* e) {
  if (D % 8 != 0 || D > 2048) return false;
  const void* ps[5] = {a, b, c, d, e};
  for (const void* q : ps)
    if (q && ((uintptr_t)q) % 16 != 0) return false;
  return true;
}

extern "C" int s2svc_layernorm_fwd(int dtype, int rows, int D, const void* x, const void* res, float drop_p, float hscale,
                                   const uint64_t* seed_base, uint64_t seed_off, const float* gamma, const float* beta, float eps, void* y,
                                   void* s_out, float* mean, float* rstd, void* stream) {
  S2S_REQUIRE(rows >= 0 && D > 0, "layernorm_fwd: bad shape");
  S2S_REQUIRE(!res || s_out, "layernorm_fwd: s_out required with residual");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((rows + 3) / 4), block(256);
#define S2S_LN_FWD(KERNEL, T)                                                                                            \
  hipLaunchKernelGGL(KERNEL, grid, block, 0, st, rows, D, (const T*)x, (const T*)res, drop_p, hscale, seed_base, seed_off, \
                     gamma, beta, eps, (T*)y, (T*)s_out, mean, rstd)
  if (dtype == S2S_F32) {
    if (D <= 512) S2S_LN_FWD((ln_fwd_reg_kernel<float, 8>), float);
    else if (D <= 1024) S2S_LN_FWD((ln_fwd_reg_kernel<float, 16>), float);
    else S2S_LN_FWD(ln_fwd_kernel<float>, float);
  } else if (ln_vec_ok(D, x, res, y, s_out, gamma) && ((uintptr_t)beta) % 16 == 0) {
    if (D <= 512) S2S_LN_FWD(ln_fwd_vec_kernel<1>, bf16_t);
    else if (D <= 1024) S2S_LN_FWD(ln_fwd_vec_kernel<2>, bf16_t);
    else S2S_LN_FWD(ln_fwd_vec_kernel<4>, bf16_t);
  } else {
    if (D <= 512) S2S_LN_FWD((ln_fwd_reg_kernel<bf16_t, 8>), bf16_t);
    else if (D <= 1024) S2S_LN_FWD((ln_fwd_reg_kernel<bf16_t, 16>), bf16_t);
    else S2S_LN_FWD(ln_fwd_kernel<bf16_t>, bf16_t);
  }
#undef S2S_LN_FWD
  S2S_CHECK_LAUNCH("ln_fwd_kernel");
  return 0;
}

extern "C" int s2svc_layernorm_bwd(int dtype, int rows, int D, const void* dy, const void* s, const float* mean,
                                   const float* rstd, const float* gamma, const void* ds_extra, float drop_p, float hscale,
                                   const uint64_t* seed_base, uint64_t seed_off, void* ds, void* dh, void* stream) {
  S2S_REQUIRE(rows >= 0 && D > 0, "layernorm_bwd: bad shape");
  if (rows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((rows + 3) / 4), block(256);
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(ln_bwd_kernel<float>, grid, block, 0, st, rows, D, (const float*)dy, (const float*)s, mean, rstd,
                       gamma, (const float*)ds_extra, drop_p, hscale, seed_base, seed_off, (float*)ds, (float*)dh);
  else if (ln_vec_ok(D, dy, s, ds, dh, ds_extra) && ((uintptr_t)gamma) % 16 == 0) {
    if (D <= 512)
      hipLaunchKernelGGL(ln_bwd_vec_kernel<1>, grid, block, 0, st, rows, D, (const bf16_t*)dy, (const bf16_t*)s, mean, rstd, gamma,
                         (const bf16_t*)ds_extra, drop_p, hscale, seed_base, seed_off, (bf16_t*)ds, (bf16_t*)dh);
    else if (D <= 1024)
      hipLaunchKernelGGL(ln_bwd_vec_kernel<2>, grid, block, 0, st, rows, D, (const bf16_t*)dy, (const bf16_t*)s, mean, rstd, gamma,
                         (const bf16_t*)ds_extra, drop_p, hscale, seed_base, seed_off, (bf16_t*)ds, (bf16_t*)dh);
    else
      hipLaunchKernelGGL(ln_bwd_vec_kernel<4>, grid, block, 0, st, rows, D, (const bf16_t*)dy, (const bf16_t*)s, mean, rstd, gamma,
                         (const bf16_t*)ds_extra, drop_p, hscale, seed_base, seed_off, (bf16_t*)ds, (bf16_t*)dh);
  } else
    hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, grid, block, 0, st, rows, D, (const bf16_t*)dy, (const bf16_t*)s, mean,
                       rstd, gamma, (const bf16_t*)ds_extra, drop_p, hscale, seed_base, seed_off, (bf16_t*)ds, (bf16_t*)dh);
  S2S_CHECK_LAUNCH("ln_bwd_kernel");
  return 0;
}

// LayerNorm backward + the first reduction stage of its parameter gradients in one launch (ln_bwd_vec_pg_kernel).
// _pg_chunks: 0 if the shape / pointers are not eligible (the caller uses s2svc_layernorm_bwd + a mode-1 column reduction), else the
// number of partial row pairs the launch writes to ws[chunks][2][D]; they enter s2svc_colreduce_grouped as a mode-7 item.
static int ln_pg_rpw(int rows) {
  static const int env = 0;
  if (env > 0) return env;
  int rpw = (rows + 2047) / 2048;                 // ~256 blocks of 8 waves
  return rpw < 1 ? 1 : rpw;
}
// > 64 KB of dynamic LDS needs the function attribute, per DEVICE (a second GPU in one process is a second function handle) and the
// call can fail: the eligibility test asks here, so a failure sends the site to the plain backward kernel + separate reduction.
template <int NV>
static bool ln_pg_lds_ok() {
  const size_t lds = (size_t)8 * 2 * NV * 512 * sizeof(float);
  if (lds <= 64 * 1024) return true;
  static int state[64] = {};                 // per device: 0 = not tried, 1 = set, -1 = refused
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (state[dev] == 0) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ln_bwd_vec_pg_kernel<NV>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) (void)hipGetLastError();
    state[dev] = e == hipSuccess ? 1 : -1;
  }
  return state[dev] == 1;
}
static bool ln_pg_lds_ok_for(int D) { return D <= 512 ? ln_pg_lds_ok<1>() : D <= 1024 ? ln_pg_lds_ok<2>() : ln_pg_lds_ok<4>(); }

extern "C" int s2svc_layernorm_bwd_pg_chunks(int dtype, int rows, int D, const void* dy, const void* s, const float* gamma,
                                             const void* ds_extra, const void* ds, const void* dh) {
  static const bool on = true;
  if (!on || dtype != S2S_BF16 || D > 2048 || rows < 2048 || (int64_t)rows * D < 1500000) return 0;
  if (!ln_vec_ok(D, dy, s, ds, dh, ds_extra) || ((uintptr_t)gamma) % 16) return 0;
  if (!ln_pg_lds_ok_for(D)) return 0;
  const int rpw = ln_pg_rpw(rows);
  return (rows + 8 * rpw - 1) / (8 * rpw);
}

extern "C" int s2svc_layernorm_bwd_pg(int dtype, int rows, int D, const void* dy, const void* s, const float* mean,
                                      const float* rstd, const float* gamma, const void* ds_extra, float drop_p, float hscale,
                                      const uint64_t* seed_base, uint64_t seed_off, void* ds, void* dh, float* ws, void* stream) {
  const int chunks = s2svc_layernorm_bwd_pg_chunks(dtype, rows, D, dy, s, gamma, ds_extra, ds, dh);
  S2S_REQUIRE(chunks > 0 && ws && ((uintptr_t)ws) % 16 == 0, "layernorm_bwd_pg: not eligible (check s2svc_layernorm_bwd_pg_chunks)");
  hipStream_t st = (hipStream_t)stream;
  const int rpw = ln_pg_rpw(rows);
#define S2S_LN_PG(NV_)                                                                                                               \
  do {                                                                                                                                \
    const size_t lds = (size_t)8 * 2 * (NV_) * 512 * sizeof(float);                                                                   \
    hipLaunchKernelGGL(ln_bwd_vec_pg_kernel<NV_>, dim3((unsigned)chunks), dim3(512), lds, st, rows, D, rpw, (const bf16_t*)dy,         \
                       (const bf16_t*)s, mean, rstd, gamma, (const bf16_t*)ds_extra, drop_p, hscale, seed_base, seed_off, (bf16_t*)ds, \
                       (bf16_t*)dh, ws);                                                                                             \
  } while (0)
  if (D <= 512) S2S_LN_PG(1);
  else if (D <= 1024) S2S_LN_PG(2);
  else S2S_LN_PG(4);
#undef S2S_LN_PG
  S2S_CHECK_LAUNCH("ln_bwd_vec_pg_kernel");
  return 0;
}

extern "C" int s2svc_colreduce(int dtype, int rows, int D, int mode, const void* dy, const void* x, const float* mean,
                               const float* rstd, float scale, float* out_sum, float* out_dot, int accumulate,
                               float* ws, int ws_chunks, int Tn, const int32_t* vlens, void* stream) {
  S2S_REQUIRE(rows >= 0 && D > 0 && ws && ws_chunks > 0, "colreduce: bad args");
  S2S_REQUIRE(mode >= 0 && mode <= 6, "colreduce: bad mode");
  S2S_REQUIRE(!vlens || (Tn > 0 && rows % Tn == 0 && mode != 1 && mode != 5), "colreduce: vlens needs rows = B * Tn (modes 0, 2, 3, 4, 6)");
  S2S_REQUIRE(scale >= 0.f || rows > 0, "colreduce: scale < 0 (mean over the present rows) needs rows");
  hipStream_t st = (hipStream_t)stream;
  int chunks = (rows + 63) / 64;
  if (chunks > ws_chunks) chunks = ws_chunks;
  if (chunks < 1) chunks = 1;
  const int rpc = (rows + chunks - 1) / chunks;
  dim3 grid((D + 63) / 64, chunks), block(256);
  if (vlens && dtype == S2S_F32)
    hipLaunchKernelGGL(colreduce_stage1_masked<float>, grid, block, 0, st, rows, D, mode, (const float*)dy, (const float*)x, mean,
                       rstd, ws, rpc, Tn, vlens);
  else if (vlens)
    hipLaunchKernelGGL(colreduce_stage1_masked<bf16_t>, grid, block, 0, st, rows, D, mode, (const bf16_t*)dy, (const bf16_t*)x,
                       mean, rstd, ws, rpc, Tn, vlens);
  else if (cr_vec_ok_host(dtype, D, dy, x))
    hipLaunchKernelGGL(colreduce_stage1_vec, dim3((D + 511) / 512, chunks), block, 0, st, rows, D, mode, (const bf16_t*)dy,
                       (const bf16_t*)x, mean, rstd, ws, rpc);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(colreduce_stage1<float>, grid, block, 0, st, rows, D, mode, (const float*)dy, (const float*)x, mean,
                       rstd, ws, rpc);
  else
    hipLaunchKernelGGL(colreduce_stage1<bf16_t>, grid, block, 0, st, rows, D, mode, (const bf16_t*)dy, (const bf16_t*)x,
                       mean, rstd, ws, rpc);
  S2S_CHECK_LAUNCH("colreduce_stage1");
  hipLaunchKernelGGL(colreduce_stage2, dim3((D + 63) / 64), dim3(256), 0, st, D, chunks, ws, scale, out_sum, out_dot,
                     accumulate, rows, Tn, vlens);
  S2S_CHECK_LAUNCH("colreduce_stage2");
  return 0;
}

extern "C" int s2svc_colreduce_grouped(const s2svc_colreduce_item* items, int n, void* stream) {
  S2S_REQUIRE(items && n > 0, "colreduce_grouped: bad args");
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += S2S_CR_MAX) {
    cr_args a;
    std::memset(&a, 0, sizeof(a));
    a.n = (n - i0 < S2S_CR_MAX) ? n - i0 : S2S_CR_MAX;
    int max_tiles = 1, max_tiles1 = 1, max_chunks = 1, n_stage1 = 0;
    for (int i = 0; i < a.n; ++i) {
      const s2svc_colreduce_item& it = items[i0 + i];
      S2S_REQUIRE(it.rows >= 0 && it.D > 0 && it.ws && it.ws_chunks > 0, "colreduce_grouped: bad item");
      S2S_REQUIRE(it.mode >= 0 && it.mode <= 7, "colreduce_grouped: bad mode");
      S2S_REQUIRE(it.dtype == S2S_F32 || it.dtype == S2S_BF16, "colreduce_grouped: bad dtype");
      int chunks = (it.rows + 63) / 64;
      if (chunks > it.ws_chunks) chunks = it.ws_chunks;
      if (chunks < 1) chunks = 1;
      if (it.mode == 7) chunks = it.ws_chunks;           // partials ready: every one of them
      a.it[i] = it;
      a.chunks[i] = chunks;
      a.rpc[i] = (it.rows + chunks - 1) / chunks;
      const int tiles = (it.D + 63) / 64;
      const int tiles1 = cr_vec_ok_host(it.dtype, it.D, it.dy, it.x) ? (it.D + 511) / 512 : tiles;     // stage 1 (see its kernel)
      if (tiles > max_tiles) max_tiles = tiles;
      if (it.mode != 7 && tiles1 > max_tiles1) max_tiles1 = tiles1;
      if (it.mode != 7 && chunks > max_chunks) max_chunks = chunks;
      if (it.mode != 7) ++n_stage1;
    }
    if (n_stage1 > 0) {
      hipLaunchKernelGGL(colreduce_grouped_stage1, dim3(max_tiles1, max_chunks, a.n), dim3(256), 0, st, a);
      S2S_CHECK_LAUNCH("colreduce_grouped_stage1");
    }
    hipLaunchKernelGGL(colreduce_grouped_stage2, dim3(max_tiles, a.n), dim3(256), 0, st, a);
    S2S_CHECK_LAUNCH("colreduce_grouped_stage2");
  }
  return 0;
}

extern "C" int s2svc_bn_finalize(int C, int n, float eps, float momentum, const float* mean, const float* var,
                                 float* rstd, float* run_mean, float* run_var, int64_t* num_batches, int var_is_ex2, int Tn,
                                 const int32_t* vlens, void* stream) {
  S2S_REQUIRE(!vlens || (Tn > 0 && n % Tn == 0), "bn_finalize: vlens needs n = B * Tn");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, n, eps, momentum,
                     mean, var, rstd, run_mean, run_var, num_batches, var_is_ex2, Tn, vlens);
  S2S_CHECK_LAUNCH("bn_finalize_kernel");
  return 0;
}

// mean / rstd of a training-mode BatchNorm1d over (rows, C) in TWO launches (one pass over x: fp32 sums of x and x^2 per chunk,
// then sum of the partials + variance + rstd + running statistics); ws >= ws_chunks*2*C floats.  Same values as
// s2svc_colreduce(mode 6, scale 1/rows) + s2svc_bn_finalize(var_is_ex2 = 1), one launch less.
extern "C" int s2svc_bn_stats(int dtype, int rows, int C, const void* x, float eps, float momentum, float* mean, float* rstd,
                              float* run_mean, float* run_var, int64_t* num_batches, float* ws, int ws_chunks, int Tn,
                              const int32_t* vlens, void* stream) {
  S2S_REQUIRE(rows > 0 && C > 0 && x && mean && rstd && ws && ws_chunks > 0, "bn_stats: bad args");
  S2S_REQUIRE(!vlens || (Tn > 0 && rows % Tn == 0), "bn_stats: vlens needs rows = B * Tn");
  S2S_REQUIRE(dtype == S2S_F32 || dtype == S2S_BF16, "bn_stats: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  int chunks = (rows + 63) / 64;
  if (chunks > ws_chunks) chunks = ws_chunks;
  const int rpc = (rows + chunks - 1) / chunks;
  dim3 grid((C + 63) / 64, chunks), block(256);
  if (vlens && dtype == S2S_F32)
    hipLaunchKernelGGL(colreduce_stage1_masked<float>, grid, block, 0, st, rows, C, 6, (const float*)nullptr, (const float*)x,
                       (const float*)nullptr, (const float*)nullptr, ws, rpc, Tn, vlens);
  else if (vlens)
    hipLaunchKernelGGL(colreduce_stage1_masked<bf16_t>, grid, block, 0, st, rows, C, 6, (const bf16_t*)nullptr, (const bf16_t*)x,
                       (const float*)nullptr, (const float*)nullptr, ws, rpc, Tn, vlens);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(colreduce_stage1<float>, grid, block, 0, st, rows, C, 6, (const float*)nullptr, (const float*)x,
                       (const float*)nullptr, (const float*)nullptr, ws, rpc);
  else
    hipLaunchKernelGGL(colreduce_stage1<bf16_t>, grid, block, 0, st, rows, C, 6, (const bf16_t*)nullptr, (const bf16_t*)x,
                       (const float*)nullptr, (const float*)nullptr, ws, rpc);
  S2S_CHECK_LAUNCH("colreduce_stage1");
  hipLaunchKernelGGL(bn_stage2_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, st, C, chunks, ws, rows, eps, momentum, mean,
                     rstd, run_mean, run_var, num_batches, Tn, vlens);
  S2S_CHECK_LAUNCH("bn_stage2_finalize_kernel");
  return 0;
}

extern "C" int s2svc_rstd_from_var(int C, float eps, const float* var, float* rstd, void* stream) {
  hipLaunchKernelGGL(rstd_from_var_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, eps, var, rstd);
  S2S_CHECK_LAUNCH("rstd_from_var_kernel");
  return 0;
}

extern "C" int s2svc_bn_apply(int dtype, int64_t rows, int C, const void* x, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, int act, float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* y,
                              void* pre_act, int Tn, const int32_t* vlens, void* stream) {
  const int64_t total = rows * C;
  if (total == 0) return 0;
  S2S_REQUIRE(!vlens || (Tn > 0 && rows % Tn == 0 && rows < (1ll << 31)), "bn_apply: vlens needs rows = B * Tn");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, st, total, C, (const float*)x, mean,
                       rstd, gamma, beta, act, drop_p, seed_base, seed_off, (float*)y, (float*)pre_act, Tn, vlens);
  else
    hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, st, total, C, (const bf16_t*)x, mean,
                       rstd, gamma, beta, act, drop_p, seed_base, seed_off, (bf16_t*)y, (bf16_t*)pre_act, Tn, vlens);
  S2S_CHECK_LAUNCH("bn_apply_kernel");
  return 0;
}

extern "C" int s2svc_bn_bwd(int dtype, int64_t rows, int C, const void* dy, const void* x, const float* mean,
                            const float* rstd, const float* gamma, const float* sum_dy, const float* sum_dy_xhat,
                            int use_batch_stats, void* dx, int Tn, const int32_t* vlens, void* stream) {
  const int64_t total = rows * C;
  if (total == 0) return 0;
  S2S_REQUIRE(!vlens || (Tn > 0 && rows % Tn == 0 && rows < (1ll << 31)), "bn_bwd: vlens needs rows = B * Tn");
  hipStream_t st = (hipStream_t)stream;
  const float inv_n = 1.0f / (float)rows;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(bn_bwd_kernel<float>, dim3(ew_blocks(total)), dim3(256), 0, st, total, C, inv_n, (const float*)dy,
                       (const float*)x, mean, rstd, gamma, sum_dy, sum_dy_xhat, use_batch_stats, (float*)dx, Tn, vlens);
  else
    hipLaunchKernelGGL(bn_bwd_kernel<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, st, total, C, inv_n, (const bf16_t*)dy,
                       (const bf16_t*)x, mean, rstd, gamma, sum_dy, sum_dy_xhat, use_batch_stats, (bf16_t*)dx, Tn, vlens);
  S2S_CHECK_LAUNCH("bn_bwd_kernel");
  return 0;
}
