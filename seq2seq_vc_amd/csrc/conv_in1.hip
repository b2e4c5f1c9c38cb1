// First layer of the Conv2d subsampling front-end: Conv2d(1 -> O, 3x3, stride 2) + ReLU on the (B, T, F) mel
// batch itself (NHWC with C = 1), output NHWC (B, T1, F1, O).
// reference: modules/transformer/subsampling.py:58-60.  K = 9 is far too thin for the MFMA GEMM; this layer is
// pure HBM streaming: 4 B/input sample read (re-reads served by L1/L2) + O*s bytes written per output pixel.
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

__device__ __forceinline__ void store8(float* p, const float (&r)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(r[4], r[5], r[6], r[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&r)[8]) {
  uint4 v;
  v.x = f2bf2(r[0], r[1]);
  v.y = f2bf2(r[2], r[3]);
  v.z = f2bf2(r[4], r[5]);
  v.w = f2bf2(r[6], r[7]);
  // non-temporal: the 122 MB output stream of the forward layer is read next by a GEMM that streams it once (37.5 -> 32.7 us for VTN's
  // layer, step 3.625 -> 3.616 ms in two interleaved pairs; bit-identical)
  typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
  u32x4_nt t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<u32x4_nt*>(p));
}

// A thread owns 8 consecutive channels for the whole launch -- its 9x8 weights + 8 biases live in 80 registers -- and walks
// every (256 / (O/8))-th pixel of the block's contiguous pixel range: per pixel 9 broadcast input loads, 72 FMAs and one
// 16-byte (bf16) / two 16-byte (fp32) coalesced stores; nothing but the store stream touches memory in the loop.
// (The first version re-read the weights from LDS for every pixel -- 320 B of LDS traffic per 16 B stored -- and ran at
// 0.67 TB/s.)
template <typename T>
__global__ __launch_bounds__(256) void conv_in1_fwd_kernel(int B, int Tn, int Fn, int T1, int F1, int O, const T* __restrict__ x,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           T* __restrict__ y, int pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [10][O]: 9 taps + bias (coalesced staging of the weights)
  for (int i = threadIdx.x; i < O * 9; i += 256) sw[(i % 9) * O + i / 9] = w[i];
  for (int i = threadIdx.x; i < O; i += 256) sw[9 * O + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int og = O / 8, PL = 256 / og;
  const int g = threadIdx.x % og, pl = threadIdx.x / og;
  if (pl >= PL) return;
  float wr[10][8];
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const float4 w0 = *reinterpret_cast<const float4*>(sw + k * O + g * 8), w1 = *reinterpret_cast<const float4*>(sw + k * O + g * 8 + 4);
    wr[k][0] = w0.x; wr[k][1] = w0.y; wr[k][2] = w0.z; wr[k][3] = w0.w; wr[k][4] = w1.x; wr[k][5] = w1.y; wr[k][6] = w1.z; wr[k][7] = w1.w;
  }
  const uint32_t npix = (uint32_t)B * T1 * F1;
  const uint32_t p0 = blockIdx.x * (uint32_t)pix_per_block;
  const uint32_t p1 = (p0 + pix_per_block < npix) ? p0 + pix_per_block : npix;
#pragma unroll 2
  for (uint32_t p = p0 + pl; p < p1; p += PL) {
    const uint32_t q = p / (uint32_t)F1;
    const int f1 = (int)(p - q * (uint32_t)F1);
    const int b = (int)(q / (uint32_t)T1);
    const int t1 = (int)(q - (uint32_t)b * (uint32_t)T1);
    const T* xb = x + ((int64_t)b * Tn + 2 * t1) * Fn + 2 * f1;
    float xv[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xv[kh * 3 + kw] = ldf(xb + kh * Fn + kw);
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = wr[9][e];
#pragma unroll
      for (int k = 0; k < 9; ++k) a += wr[k][e] * xv[k];
      r[e] = a > 0.f ? a : 0.f;
    }
    store8(y + (int64_t)p * O + g * 8, r);
  }
}

// partial[chunk][o][0..8] = sum_p dy[p,o]*x[p,tap] ; partial[chunk][o][9] = sum_p dy[p,o]   (dy already ReLU-masked)
template <typename T>
__global__ __launch_bounds__(256) void conv_in1_wgrad_kernel(int B, int Tn, int Fn, int T1, int F1, int O, const T* __restrict__ x,
                                                             const T* __restrict__ dy, const T* __restrict__ y,
                                                             float* __restrict__ partial, int pix_per_chunk) {
  const int64_t npix = (int64_t)B * T1 * F1;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_chunk;
  const int64_t p1 = (p0 + pix_per_chunk < npix) ? p0 + pix_per_chunk : npix;
  for (int o = threadIdx.x; o < O; o += 256) {
    float acc[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.f;
    for (int64_t p = p0; p < p1; ++p) {
      const int f1 = (int)(p % F1);
      const int64_t q = p / F1;
      const int t1 = (int)(q % T1);
      const int b = (int)(q / T1);
      float g = ldf(dy + p * O + o);
      if (y && !(ldf(y + p * O + o) > 0.f)) g = 0.f;      // relu' of the forward output, fused (no mask pass over dy)
      const T* xb = x + ((int64_t)b * Tn + 2 * t1) * Fn + 2 * f1;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] += g * ldf(xb + kh * Fn + kw);
      acc[9] += g;
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) partial[((int64_t)blockIdx.x * O + o) * 10 + k] = acc[k];
  }
}

// bf16 streaming variant: a thread owns 8 channels (one 16-byte load of dy and of y per pixel) and every
// (256 / (O/8))-th pixel of the chunk; 80 fp32 accumulators per thread, pixel lanes folded through LDS in a fixed
// order at the end.  HBM-bound: 2 (dy) + 2 (y) bytes per (pixel, channel); the 9 input samples are L1 broadcasts.
struct __attribute__((aligned(4))) u32x2_a4 { uint32_t a, b; };
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(w[e] << 16); f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}

__global__ __launch_bounds__(256) void conv_in1_wgrad_vec_kernel(int B, int Tn, int Fn, int T1, int F1, int O,
                                                                 const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                 const bf16_t* __restrict__ y, float* __restrict__ partial,
                                                                 int pix_per_chunk) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [O/8][80]
  const int og = O / 8, PL = 256 / og;
  const int g = threadIdx.x % og, pl = threadIdx.x / og;
  const int64_t npix = (int64_t)B * T1 * F1;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_chunk;
  const int64_t p1 = (p0 + pix_per_chunk < npix) ? p0 + pix_per_chunk : npix;
  float acc[8][10];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[e][k] = 0.f;
  // a pixel's window starts at an even column: with an even row pitch every row of it is 4-byte aligned and its 4th value
  // (read, not used) is still inside the row (2 f1 + 3 <= Fn - 1)
  const bool x_pairs = (Fn % 2 == 0) && ((reinterpret_cast<uintptr_t>(x) & 3) == 0);
  const uint32_t mgF = 0xFFFFFFFFu / (uint32_t)F1 + 1u, mgT = 0xFFFFFFFFu / (uint32_t)T1 + 1u;      // (F1, T1 >= 2: see the launcher)
  if (pl < PL) {
#pragma unroll 2
    for (int64_t p = p0 + pl; p < p1; p += PL) {
      const uint4 gv = *reinterpret_cast<const uint4*>(dy + p * O + g * 8);
      uint4 yv = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
      if (y) yv = *reinterpret_cast<const uint4*>(y + p * O + g * 8);
      // pixel -> (b, t1, f1) by multiply-high (exact: npix * F1 < 2^32 is checked by the launcher): the two integer divisions
      // were ~80 of the ~200 instructions of an iteration of this VALU-bound loop
      const uint32_t pi = (uint32_t)p, q = __umulhi(pi, mgF);
      const int f1 = (int)(pi - q * (uint32_t)F1);
      const int b = (int)__umulhi(q, mgT);
      const int t1 = (int)(q - (uint32_t)b * (uint32_t)T1);
      const bf16_t* xb = x + ((int64_t)b * Tn + 2 * t1) * Fn + 2 * f1;
      float xv[9];
      if (x_pairs) {                 // (uniform) even row pitch, 4-byte aligned base: the 3 taps of a row are one 8-byte load
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const u32x2_a4 v = *reinterpret_cast<const u32x2_a4*>(xb + kh * Fn);
          xv[kh * 3 + 0] = __uint_as_float(v.a << 16);
          xv[kh * 3 + 1] = __uint_as_float(v.a & 0xffff0000u);
          xv[kh * 3 + 2] = __uint_as_float(v.b << 16);
        }
      } else {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) xv[kh * 3 + kw] = ldf(xb + kh * Fn + kw);
      }
      float gf[8], yf[8];
      unpack8(gv, gf);
      unpack8(yv, yf);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float gg = yf[e] > 0.f ? gf[e] : 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[e][k] += gg * xv[k];
        acc[e][9] += gg;
      }
    }
  }
  for (int r = 1; r < PL; ++r) {           // fixed-order fold of the pixel lanes
    __syncthreads();
    if (pl == r) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < 10; ++k) red[(e * 10 + k) * og + g] = acc[e][k];
    }
    __syncthreads();
    if (pl == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[e][k] += red[(e * 10 + k) * og + g];
    }
  }
  if (pl == 0) {
    float* out = partial + ((int64_t)blockIdx.x * O + g * 8) * 10;
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int k = 0; k < 10; k += 2) *reinterpret_cast<float2*>(out + e * 10 + k) = make_float2(acc[e][k], acc[e][k + 1]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same partial sums on the matrix cores (bf16, no ReLU mask: the consumer's data-gradient GEMM already applied it).
// conv_in1_wgrad_vec_kernel is VALU-bound: 80 FMAs + unpacking + index arithmetic per 16 bytes of dy -- 64 us for the 122 MB of
// VTN's dy (1.9 TB/s) as the very LAST kernel of the backward pass.  As a product it is  P[tap, o] = sum_pix xcol[pix, tap] *
// dy[pix, o]  with 9 taps (+ a column of ones: the bias gradient) padded to the 16 rows of v_mfma_f32_16x16x32_bf16:
//   * a workgroup walks its pixel chunk in blocks of 32 pixels (the K of one MFMA); the block of dy (32 x O bf16) goes through LDS
//     (16-byte global loads, padded rows) because a B fragment needs 8 CONSECUTIVE PIXELS of one channel per lane (2-byte LDS reads);
//   * A fragment: lane (tap = lane & 15, pixel group lane >> 4) gathers its 8 pixels' window values straight from x (L2-resident);
//   * the four waves split the O / 16 channel groups; accumulators 16 taps x 16 channels per group.
// Any lane -> k assignment is fine as long as A and B agree (both: pixel 8 (lane >> 4) + j), rows / columns are lane & 15.
// Partials leave in the layout of the VALU kernel ([chunk][O][10]); conv_in1_wgrad_final_kernel is shared.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 ci_bf16x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short ci_u16x8_t;
typedef __attribute__((ext_vector_type(4))) float ci_f32x4_t;

// blockIdx.y = slice of 64 GPW channels: the LDS block of a slice of 192 channels is 12.5 KB, which still fits beside a 144 KB
// workgroup of the weight-gradient GEMM that runs next to this kernel at the end of the backward pass (with all 384 channels in
// one 25 KB block the two could not share a CU: VTN step +0.06 ms although the kernel alone was 15 us faster).
template <int GPW>        // channel groups (of 16) per wave = 16-byte dy vectors per thread and block: a slice is 64 GPW channels
__global__ __launch_bounds__(256) void conv_in1_wgrad_mfma_kernel(int B, int Tn, int Fn, int T1, int F1, int Ot,
                                                                  const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                  float* __restrict__ partial, int pix_per_chunk) {
  constexpr int O = 64 * GPW, PITCH = O + 8, VPR = O / 8;
  __shared__ __attribute__((aligned(16))) bf16_t tile[32 * PITCH];
  const int cbase = blockIdx.y * O;
  dy += cbase;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kg = lane >> 4;
  const int64_t npix = (int64_t)B * T1 * F1;
  const int64_t p0 = (int64_t)blockIdx.x * pix_per_chunk;
  const int64_t p1 = (p0 + pix_per_chunk < npix) ? p0 + pix_per_chunk : npix;
  ci_f32x4_t acc[GPW];
#pragma unroll
  for (int g = 0; g < GPW; ++g) acc[g] = (ci_f32x4_t){0.f, 0.f, 0.f, 0.f};
  const uint32_t mgF = 0xFFFFFFFFu / (uint32_t)F1 + 1u, mgT = 0xFFFFFFFFu / (uint32_t)T1 + 1u;
  const int kh = n / 3, kw = n - kh * 3;                     // (taps 0 .. 8)
  // this thread's GPW vectors of a dy block: vector v = tid + 256 i -> (row, 8-channel piece)
  int vrow[GPW], vcol[GPW];
#pragma unroll
  for (int i = 0; i < GPW; ++i) {
    const int v = tid + 256 * i;
    vrow[i] = v / VPR;
    vcol[i] = (v - vrow[i] * VPR) * 8;
  }
  uint4 vals[GPW];
  ci_u16x8_t au;
  // requests of one block: dy vectors into registers, this lane's tap over its 8 pixels (A fragment)
#define CI_FETCH(P)                                                                                                          \
  {                                                                                                                          \
    const int64_t pb_ = (P);                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < GPW; ++i) {                                                                        \
      vals[i] = make_uint4(0u, 0u, 0u, 0u);                                                                                  \
      if (pb_ + vrow[i] < p1) vals[i] = *reinterpret_cast<const uint4*>(dy + (pb_ + vrow[i]) * Ot + vcol[i]);                \
    }                                                                                                                        \
    au = (ci_u16x8_t){0, 0, 0, 0, 0, 0, 0, 0};                                                                               \
    if (n < 10) {                                                                                                            \
      const int64_t pf_ = pb_ + 8 * kg;                                                                                      \
      const uint32_t pi_ = (uint32_t)pf_, q_ = __umulhi(pi_, mgF);                                                           \
      int f1_ = (int)(pi_ - q_ * (uint32_t)F1);                                                                              \
      int b_ = (int)__umulhi(q_, mgT);                                                                                       \
      int t1_ = (int)(q_ - (uint32_t)b_ * (uint32_t)T1);                                                                     \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                        \
        if (pf_ + j < p1) au[j] = n == 9 ? (unsigned short)0x3f80 : x[((int64_t)b_ * Tn + 2 * t1_ + kh) * Fn + 2 * f1_ + kw]; \
        if (++f1_ == F1) { f1_ = 0; if (++t1_ == T1) { t1_ = 0; ++b_; } }                                                    \
      }                                                                                                                      \
    }                                                                                                                        \
  }
  CI_FETCH(p0);
  for (int64_t p = p0; p < p1; p += 32) {
#pragma unroll
    for (int i = 0; i < GPW; ++i) *reinterpret_cast<uint4*>(tile + vrow[i] * PITCH + vcol[i]) = vals[i];
    const ci_bf16x8_t a = __builtin_bit_cast(ci_bf16x8_t, au);
    __syncthreads();
    if (p + 32 < p1) CI_FETCH(p + 32);                       // in flight behind this block's fragment reads and MFMAs
#pragma unroll
    for (int g = 0; g < GPW; ++g) {
      const int ch = (wave * GPW + g) * 16 + n;
      ci_u16x8_t bu;
#pragma unroll
      for (int j = 0; j < 8; ++j) bu[j] = tile[(8 * kg + j) * PITCH + ch];
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(ci_bf16x8_t, bu), acc[g], 0, 0, 0);
    }
    __syncthreads();
  }
#undef CI_FETCH
  // D[row = 4 (lane >> 4) + r (tap), col = lane & 15 (channel)]
#pragma unroll
  for (int g = 0; g < GPW; ++g) {
    const int ch = (wave * GPW + g) * 16 + n;
    float* out = partial + ((int64_t)blockIdx.x * Ot + cbase + ch) * 10;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tap = 4 * kg + r;
      if (tap < 10) out[tap] = acc[g][r];
    }
  }
}

// one wavefront per output element: lanes stride over the chunk partials (independent loads), fixed-order wave sum
__global__ __launch_bounds__(256) void conv_in1_wgrad_final_kernel(int O, int chunks, const float* __restrict__ partial,
                                                                   float* __restrict__ dw, float* __restrict__ db, int accumulate) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= O * 10) return;
  float t = 0.f;
  for (int c = lane; c < chunks; c += 64) t += partial[(int64_t)c * O * 10 + i];
  t = wave_sum(t);
  if (lane == 0) {
    const int o = i / 10, k = i % 10;
    if (k < 9) dw[o * 9 + k] = (accumulate ? dw[o * 9 + k] : 0.f) + t;
    else if (db) db[o] = (accumulate ? db[o] : 0.f) + t;
  }
}

}  // namespace

extern "C" int s2svc_conv_in1_fwd(int dtype, int B, int Tn, int Fn, int O, const void* x, const float* w, const float* bias,
                                  void* y, void* stream) {
  S2S_REQUIRE(O % 8 == 0 && Tn >= 3 && Fn >= 3, "conv_in1_fwd: need O % 8 == 0 and T,F >= 3");
  const int T1 = (Tn - 3) / 2 + 1, F1 = (Fn - 3) / 2 + 1;
  const int64_t npix = (int64_t)B * T1 * F1;
  if (npix == 0) return 0;
  S2S_REQUIRE(O <= 2048 && npix < ((int64_t)1 << 31), "conv_in1_fwd: need O <= 2048 and fewer than 2^31 output pixels");
  const int lanes = 256 / (O / 8);
  int nb = (int)((npix + lanes * 8 - 1) / (lanes * 8));        // >= 8 pixels per lane
  if (nb > 1024) nb = 1024;
  const int ppb = (int)((npix + nb - 1) / nb);
  nb = (int)((npix + ppb - 1) / ppb);
  const size_t shm = (size_t)O * 10 * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(conv_in1_fwd_kernel<float>, dim3(nb), dim3(256), shm, st, B, Tn, Fn, T1, F1, O, (const float*)x, w, bias, (float*)y, ppb);
  else
    hipLaunchKernelGGL(conv_in1_fwd_kernel<bf16_t>, dim3(nb), dim3(256), shm, st, B, Tn, Fn, T1, F1, O, (const bf16_t*)x, w, bias, (bf16_t*)y, ppb);
  S2S_CHECK_LAUNCH("conv_in1_fwd_kernel");
  return 0;
}

// partial: >= chunks*O*10 floats with chunks = min(1024, ceil(npix/64)); dw (O,9), db (O) fp32.
// y (nullable): the forward output; when given, dy is masked by relu'(y) on the fly (dy then is the gradient of the
// ReLU OUTPUT, as autograd hands it over).
extern "C" int s2svc_conv_in1_wgrad(int dtype, int B, int Tn, int Fn, int O, const void* x, const void* dy, const void* y,
                                    float* dw, float* db, int accumulate, float* partial, int max_chunks, void* stream) {
  const int T1 = (Tn - 3) / 2 + 1, F1 = (Fn - 3) / 2 + 1;
  const int64_t npix = (int64_t)B * T1 * F1;
  if (npix == 0) return 0;
  S2S_REQUIRE(partial && max_chunks > 0, "conv_in1_wgrad: workspace required");
  int chunks = (int)((npix + 63) / 64);
  if (chunks > max_chunks) chunks = max_chunks;
  const int ppc = (int)((npix + chunks - 1) / chunks);
  chunks = (int)((npix + ppc - 1) / ppc);
  hipStream_t st = (hipStream_t)stream;
  const int og = O / 8;
  static const bool mfma_on = true;
  if (mfma_on && dtype == S2S_BF16 && !y && O % 64 == 0 && npix * (F1 > T1 ? F1 : T1) < ((int64_t)1 << 32) && F1 >= 2 && T1 >= 2 &&
      (uintptr_t)dy % 16 == 0) {
    const int gq = O / 64, gpw = gq % 3 == 0 ? 3 : gq % 2 == 0 ? 2 : 1;
    const dim3 grid((unsigned)chunks, (unsigned)(gq / gpw));
#define CI_LAUNCH(G) hipLaunchKernelGGL(conv_in1_wgrad_mfma_kernel<G>, grid, dim3(256), 0, st, B, Tn, Fn, T1, F1, O, \
                                        (const bf16_t*)x, (const bf16_t*)dy, partial, ppc)
    if (gpw == 3) CI_LAUNCH(3);
    else if (gpw == 2) CI_LAUNCH(2);
    else CI_LAUNCH(1);
#undef CI_LAUNCH
  } else if (dtype != S2S_F32 && O % 8 == 0 && og <= 256 && npix * (F1 > T1 ? F1 : T1) < ((int64_t)1 << 32) && F1 >= 2 && T1 >= 2 &&
      (uintptr_t)dy % 16 == 0 && (!y || (uintptr_t)y % 16 == 0))
    hipLaunchKernelGGL(conv_in1_wgrad_vec_kernel, dim3(chunks), dim3(256), (size_t)og * 80 * sizeof(float), st, B, Tn, Fn, T1, F1, O,
                       (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)y, partial, ppc);
  else if (dtype == S2S_F32)
    hipLaunchKernelGGL(conv_in1_wgrad_kernel<float>, dim3(chunks), dim3(256), 0, st, B, Tn, Fn, T1, F1, O, (const float*)x, (const float*)dy, (const float*)y, partial, ppc);
  else
    hipLaunchKernelGGL(conv_in1_wgrad_kernel<bf16_t>, dim3(chunks), dim3(256), 0, st, B, Tn, Fn, T1, F1, O, (const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)y, partial, ppc);
  S2S_CHECK_LAUNCH("conv_in1_wgrad_kernel");
  hipLaunchKernelGGL(conv_in1_wgrad_final_kernel, dim3((O * 10 + 3) / 4), dim3(256), 0, st, O, chunks, partial, dw, db, accumulate);
  S2S_CHECK_LAUNCH("conv_in1_wgrad_final_kernel");
  return 0;
}
