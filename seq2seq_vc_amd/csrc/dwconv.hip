// Depthwise (groups = channels) 1-D convolution on channel-last activations, with dilation.
//   y[b,t,c] = bias[c] + sum_j w[c,j] * x[b, t + (j - (k-1)/2)*dil, c]        ('same' zero padding)
// reference: modules/conformer/convolution.py:42-51,70 (depthwise_conv, k = 7/15/31) and the dilated
// depth-separable convs of the VITS flows (modules/vits/flow.py:137-146).  HBM-bound: lanes run along the
// contiguous channel axis (coalesced), the k taps of a (t, c) output re-read neighbouring rows from L1/L2.
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

template <typename T>
__global__ void dwconv_fwd_kernel(int B, int Tn, int C, int ks, int dil, const T* __restrict__ x, const float* __restrict__ w,
                                  const float* __restrict__ bias, T* __restrict__ y, int flip) {
  const int64_t n = (int64_t)B * Tn * C;
  const int pad = (ks - 1) / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t bt = i / C;
    const int t = (int)(bt % Tn);
    float acc = (bias && !flip) ? bias[c] : 0.f;
    for (int j = 0; j < ks; ++j) {
      const int tt = t + (j - pad) * dil;
      if (tt >= 0 && tt < Tn) {
        const float wv = flip ? w[c * ks + (ks - 1 - j)] : w[c * ks + j];
        acc += wv * ldf(x + i + (int64_t)(tt - t) * C);
      }
    }
    stf(y + i, acc);
  }
}

// Vectorised forward: a thread owns VEC consecutive channels and TT consecutive frames of one utterance; the
// workgroup's weights sit in LDS tap-major.  With dilation 1 the TT + ks - 1 input rows are each loaded once
// (16 bytes) and scattered into the outputs they touch; other dilations load per (output, tap).
template <typename T> struct DwVec;
template <> struct DwVec<bf16_t> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&f)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&f)[8]) {
    uint4 v;
    v.x = f2bf2(f[0], f[1]);
    v.y = f2bf2(f[2], f[3]);
    v.z = f2bf2(f[4], f[5]);
    v.w = f2bf2(f[6], f[7]);
    *reinterpret_cast<uint4*>(p) = v;
  }
};
template <> struct DwVec<float> {
  static constexpr int VEC = 4;
  static __device__ __forceinline__ void ld(const float* p, float (&f)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&f)[4]) { *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]); }
};

template <typename T, int TT>
__global__ __launch_bounds__(256) void dwconv_fwd_vec_kernel(int Tn, int C, int ks, int dil, const T* __restrict__ x,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             T* __restrict__ y, int flip, int tchunks, const T* __restrict__ add) {
  constexpr int VEC = DwVec<T>::VEC;
  constexpr int CT = 8 * VEC;                 // channels per workgroup: 8 vector lanes
  extern __shared__ float sw[];               // [ks][CT] weights (tap-major), then [CT] bias
  const int c0 = blockIdx.x * CT;
  const int b = blockIdx.y / tchunks, tc = blockIdx.y % tchunks;
  const int pad = (ks - 1) / 2;
  for (int i = threadIdx.x; i < ks * CT; i += 256) {
    const int j = i / CT, cl = i % CT;
    const int c = c0 + cl;
    sw[i] = c < C ? (flip ? w[c * ks + (ks - 1 - j)] : w[c * ks + j]) : 0.f;
  }
  for (int i = threadIdx.x; i < CT; i += 256) sw[ks * CT + i] = (bias && !flip && c0 + i < C) ? bias[c0 + i] : 0.f;
  __syncthreads();
  const int vl = threadIdx.x & 7;              // vector lane within the channel tile
  const int c = c0 + vl * VEC;
  const int t0 = (tc * 32 + (threadIdx.x >> 3)) * TT;
  if (c >= C || t0 >= Tn) return;
  const T* xb = x + (int64_t)b * Tn * C + c;
  T* yb = y + (int64_t)b * Tn * C + c;
  float acc[TT][VEC];
#pragma unroll
  for (int o = 0; o < TT; ++o)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[o][e] = sw[ks * CT + vl * VEC + e];
  if (dil == 1) {
    for (int rr = 0; rr < TT + ks - 1; ++rr) {
      const int tt = t0 - pad + rr;
      if (tt < 0 || tt >= Tn) continue;
      float f[VEC];
      DwVec<T>::ld(xb + (int64_t)tt * C, f);
#pragma unroll
      for (int o = 0; o < TT; ++o) {
        const int j = rr - o;                  // tap index for output t0+o
        if (j >= 0 && j < ks) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[o][e] += sw[j * CT + vl * VEC + e] * f[e];
        }
      }
    }
  } else {
#pragma unroll
    for (int o = 0; o < TT; ++o) {
      for (int j = 0; j < ks; ++j) {
        const int tt = t0 + o + (j - pad) * dil;
        if (tt < 0 || tt >= Tn || t0 + o >= Tn) continue;
        float f[VEC];
        DwVec<T>::ld(xb + (int64_t)tt * C, f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[o][e] += sw[j * CT + vl * VEC + e] * f[e];
      }
    }
  }
#pragma unroll
  for (int o = 0; o < TT; ++o)
    if (t0 + o < Tn) {
      if (add) {                                   // y = conv + add (a second gradient of the same tensor rides along)
        float f[VEC];
        DwVec<T>::ld(add + (int64_t)b * Tn * C + c + (int64_t)(t0 + o) * C, f);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[o][e] += f[e];
      }
      DwVec<T>::st(yb + (int64_t)(t0 + o) * C, acc[o]);
    }
}

// partial dw[chunk][c][j] = sum over the chunk's rows of dy[b,t,c] * x[b, t+(j-pad)*dil, c]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(int B, int Tn, int C, int ks, int dil, const T* __restrict__ x,
                                                           const T* __restrict__ dy, float* __restrict__ ws,
                                                           int rows_per_chunk) {
  __shared__ float sh[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int chunk = blockIdx.y, j = blockIdx.z;
  const int pad = (ks - 1) / 2;
  const int rows = B * Tn;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = (r0 + rows_per_chunk < rows) ? r0 + rows_per_chunk : rows;
  float acc = 0.f;
  if (c < C) {
    for (int r = r0 + rl; r < r1; r += 4) {
      const int t = r % Tn;
      const int tt = t + (j - pad) * dil;
      if (tt >= 0 && tt < Tn) acc += ldf(dy + (int64_t)r * C + c) * ldf(x + (int64_t)(r + tt - t) * C + c);
    }
  }
  sh[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && c < C) ws[((int64_t)chunk * C + c) * ks + j] = sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl];
}

// Tiled variant (ks <= KSMAX, halo (ks-1)/2 * dil <= DW_HALO): a workgroup stages DW_ROWS rows of dy and the same rows of x
// plus the halo for 64 channels in LDS (fp32, one coalesced pass over each tensor instead of one pass per tap), a thread
// owns one channel and a quarter of the rows and keeps all ks tap sums in registers.
constexpr int DW_ROWS = 64, DW_HALO = 32;
template <typename T, int KSMAX>
__global__ __launch_bounds__(256) void dwconv_wgrad_tiled_kernel(int B, int Tn, int C, int ks, int dil, const T* __restrict__ x,
                                                                 const T* __restrict__ dy, float* __restrict__ ws,
                                                                 int rows_per_chunk) {
  extern __shared__ float dsh[];                    // x tile (DW_ROWS + 2 halo) x 64, dy tile DW_ROWS x 64
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int chunk = blockIdx.y;
  const int pad = (ks - 1) / 2, halo = pad * dil;
  float* xt = dsh;
  float* yt = dsh + (DW_ROWS + 2 * halo) * 64;
  const int rows = B * Tn;
  const int r0 = chunk * rows_per_chunk;
  const int r1 = (r0 + rows_per_chunk < rows) ? r0 + rows_per_chunk : rows;
  float acc[KSMAX];
#pragma unroll
  for (int j = 0; j < KSMAX; ++j) acc[j] = 0.f;
  for (int rb = r0; rb < r1; rb += DW_ROWS) {
    __syncthreads();
    // batches of 8 independent loads per thread, then the LDS stores (a load-store loop would wait for every load)
    for (int pb = 0; pb < DW_ROWS + 2 * halo; pb += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = pb + u * 4 + rl, r = rb - halo + p;
        v[u] = (c < C && p < DW_ROWS + 2 * halo && r >= 0 && r < rows) ? ldf(x + (int64_t)r * C + c) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = pb + u * 4 + rl;
        if (p < DW_ROWS + 2 * halo) xt[p * 64 + cl] = v[u];
      }
    }
    for (int pb = 0; pb < DW_ROWS; pb += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = pb + u * 4 + rl, r = rb + p;
        v[u] = (c < C && r < r1) ? ldf(dy + (int64_t)r * C + c) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) yt[(pb + u * 4 + rl) * 64 + cl] = v[u];
    }
    __syncthreads();
    for (int p = rl * (DW_ROWS / 4); p < (rl + 1) * (DW_ROWS / 4); ++p) {
      const int r = rb + p;
      if (r >= r1) break;
      const int t = r % Tn;
      const float g = yt[p * 64 + cl];
#pragma unroll
      for (int j = 0; j < KSMAX; ++j) {
        if (j < ks) {
          const int off = (j - pad) * dil, tt = t + off;
          if (tt >= 0 && tt < Tn) acc[j] += g * xt[(p + halo + off) * 64 + cl];
        }
      }
    }
  }
  __syncthreads();
  float* red = dsh;                                 // [tap][row group][channel] partial sums (KSMAX x 4 x 64 <= the x tile)
#pragma unroll
  for (int j = 0; j < KSMAX; ++j)
    if (j < ks) red[(j * 4 + rl) * 64 + cl] = acc[j];
  __syncthreads();
  for (int e = threadIdx.x; e < ks * 64; e += 256) {
    const int j = e >> 6, ch = e & 63;
    const int cc = blockIdx.x * 64 + ch;
    if (cc < C) {
      const float* q = red + j * 256 + ch;
      ws[((int64_t)chunk * C + cc) * ks + j] = q[0] + q[64] + q[128] + q[192];
    }
  }
}

__global__ void dwconv_wgrad_final_kernel(int n, int chunks, const float* __restrict__ ws, float* __restrict__ dw, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float t = 0.f;
#pragma unroll 16
  for (int k = 0; k < chunks; ++k) t += ws[(int64_t)k * n + i];        // same order of additions, 16 loads in flight
  dw[i] = (accumulate ? dw[i] : 0.f) + t;
}

inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int s2svc_dwconv_add(int dtype, int B, int Tn, int C, int ks, int dil, const void* x, const float* w, const float* bias,
                                const void* add, void* y, int flip, void* stream);
// flip = 0: forward.  flip = 1: data gradient (same kernel on dy with reversed taps, no bias).
extern "C" int s2svc_dwconv(int dtype, int B, int Tn, int C, int ks, int dil, const void* x, const float* w,
                            const float* bias, void* y, int flip, void* stream) {
  return s2svc_dwconv_add(dtype, B, Tn, C, ks, dil, x, w, bias, nullptr, y, flip, stream);
}

// y = dwconv(x) + add (add: same shape as y, or NULL).  Needs the vectorised kernel when add is given (C % 4 (fp32) / 8 (bf16) == 0,
// 16-byte aligned tensors, ks <= 63).
extern "C" int s2svc_dwconv_add(int dtype, int B, int Tn, int C, int ks, int dil, const void* x, const float* w, const float* bias,
                                const void* add, void* y, int flip, void* stream) {
  const int64_t n = (int64_t)B * Tn * C;
  if (n == 0) return 0;
  S2S_REQUIRE(ks >= 1 && (ks & 1) && dil >= 1, "dwconv: kernel size must be odd, dilation >= 1");
  hipStream_t st = (hipStream_t)stream;
  const int vec = dtype == S2S_F32 ? 4 : 8;
  const bool vec_ok = C % vec == 0 && ((uintptr_t)x) % 16 == 0 && ((uintptr_t)y) % 16 == 0 && ((uintptr_t)add) % 16 == 0 && ks <= 63;
  S2S_REQUIRE(!add || vec_ok, "dwconv_add: the additive input needs the vectorised kernel (aligned tensors, C % 4 / 8 == 0)");
  if (vec_ok) {
    constexpr int TT = 4;
    const int tchunks = (Tn + 32 * TT - 1) / (32 * TT);
    dim3 grid((C + 8 * vec - 1) / (8 * vec), B * tchunks);
    const size_t shm = (size_t)(ks + 1) * 8 * vec * sizeof(float);
    if (dtype == S2S_F32)
      hipLaunchKernelGGL((dwconv_fwd_vec_kernel<float, TT>), grid, dim3(256), shm, st, Tn, C, ks, dil, (const float*)x, w, bias, (float*)y, flip, tchunks, (const float*)add);
    else
      hipLaunchKernelGGL((dwconv_fwd_vec_kernel<bf16_t, TT>), grid, dim3(256), shm, st, Tn, C, ks, dil, (const bf16_t*)x, w, bias, (bf16_t*)y, flip, tchunks, (const bf16_t*)add);
    S2S_CHECK_LAUNCH("dwconv_fwd_vec_kernel");
    return 0;
  }
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(dwconv_fwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, B, Tn, C, ks, dil, (const float*)x, w, bias, (float*)y, flip);
  else
    hipLaunchKernelGGL(dwconv_fwd_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, B, Tn, C, ks, dil, (const bf16_t*)x, w, bias, (bf16_t*)y, flip);
  S2S_CHECK_LAUNCH("dwconv_fwd_kernel");
  return 0;
}

// ws: >= ws_chunks * C * ks floats
extern "C" int s2svc_dwconv_wgrad(int dtype, int B, int Tn, int C, int ks, int dil, const void* x, const void* dy,
                                  float* dw, int accumulate, float* ws, int ws_chunks, void* stream) {
  const int rows = B * Tn;
  if (rows == 0) return 0;
  S2S_REQUIRE(ws && ws_chunks > 0, "dwconv_wgrad: workspace required");
  hipStream_t st = (hipStream_t)stream;
  int chunks = (rows + 127) / 128;
  if (chunks > ws_chunks) chunks = ws_chunks;
  const int rpc = (rows + chunks - 1) / chunks;
  const int halo = (ks - 1) / 2 * dil;
  if (ks <= 16 && halo <= DW_HALO) {
    chunks = (rows + DW_ROWS - 1) / DW_ROWS;
    if (chunks > ws_chunks) chunks = ws_chunks;
    const int rpc = ((rows + chunks - 1) / chunks + DW_ROWS - 1) / DW_ROWS * DW_ROWS;     // whole tiles per workgroup
    chunks = (rows + rpc - 1) / rpc;
    dim3 tgrid((C + 63) / 64, chunks);
    const size_t shm = (size_t)(2 * DW_ROWS + 2 * halo) * 64 * sizeof(float);
    if (dtype == S2S_F32)
      hipLaunchKernelGGL((dwconv_wgrad_tiled_kernel<float, 16>), tgrid, dim3(256), shm, st, B, Tn, C, ks, dil, (const float*)x, (const float*)dy, ws, rpc);
    else
      hipLaunchKernelGGL((dwconv_wgrad_tiled_kernel<bf16_t, 16>), tgrid, dim3(256), shm, st, B, Tn, C, ks, dil, (const bf16_t*)x, (const bf16_t*)dy, ws, rpc);
    S2S_CHECK_LAUNCH("dwconv_wgrad_tiled_kernel");
    const int n = C * ks;
    hipLaunchKernelGGL(dwconv_wgrad_final_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, chunks, ws, dw, accumulate);
    S2S_CHECK_LAUNCH("dwconv_wgrad_final_kernel");
    return 0;
  }
  dim3 grid((C + 63) / 64, chunks, ks);
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(dwconv_wgrad_kernel<float>, grid, dim3(256), 0, st, B, Tn, C, ks, dil, (const float*)x, (const float*)dy, ws, rpc);
  else
    hipLaunchKernelGGL(dwconv_wgrad_kernel<bf16_t>, grid, dim3(256), 0, st, B, Tn, C, ks, dil, (const bf16_t*)x, (const bf16_t*)dy, ws, rpc);
  S2S_CHECK_LAUNCH("dwconv_wgrad_kernel");
  const int n = C * ks;
  hipLaunchKernelGGL(dwconv_wgrad_final_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, chunks, ws, dw, accumulate);
  S2S_CHECK_LAUNCH("dwconv_wgrad_final_kernel");
  return 0;
}
