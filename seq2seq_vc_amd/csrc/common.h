// Shared device helpers for the gfx950 (CDNA4) kernels of the seq2seq-vc hot path.
// Wavefront = 64 lanes everywhere; no 32-wide assumptions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define S2S_WAVE 64

enum { S2S_F32 = 0, S2S_BF16 = 1 };
enum { S2S_ACT_NONE = 0, S2S_ACT_RELU = 1, S2S_ACT_TANH = 2, S2S_ACT_SWISH = 3, S2S_ACT_SIGMOID = 4,
       S2S_ACT_GELU = 5 };

typedef uint16_t bf16_t;  // raw bf16 storage

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserved (matches torch's float->bfloat16 cast)
__device__ __forceinline__ bf16_t f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

template <typename T> struct Cvt;
template <> struct Cvt<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Cvt<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};
template <typename T> __device__ __forceinline__ float ldf(const T* p) { return Cvt<T>::ld(p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { Cvt<T>::st(p, v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case S2S_ACT_RELU: return x > 0.f ? x : 0.f;
    case S2S_ACT_TANH: return tanhf(x);
    case S2S_ACT_SWISH: return x / (1.f + __expf(-x));
    case S2S_ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
    case S2S_ACT_GELU: return 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
    default: return x;
  }
}

// ---------------------------------------------------------------------------------------------
// Counter-based RNG (Philox-4x32-10) for dropout masks: a mask is a pure function of
// (seed, element index), so backward kernels regenerate it instead of reading it from HBM.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                             uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
  uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
  uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ uint4 philox4(uint64_t seed, uint64_t ctr) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0x243F6A88u, c3 = 0x85A308D3u;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}
// keep-scale for element idx: 0 (dropped) or 1/(1-p). One Philox call covers 4 consecutive idx.
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
  uint4 r = philox4(seed, idx >> 2);
  uint32_t w = (idx & 3) == 0 ? r.x : (idx & 3) == 1 ? r.y : (idx & 3) == 2 ? r.z : r.w;
  // uniform in [0,1): drop when u < p
  float u = (float)(w >> 8) * (1.0f / 16777216.0f);
  return u < p ? 0.f : inv_keep;
}

// error plumbing shared by all translation units (defined in api.hip)
extern "C" void s2svc_set_error(const char* msg);
#define S2S_CHECK_LAUNCH(name)                                         \
  do {                                                                 \
    hipError_t _e = hipGetLastError();                                 \
    if (_e != hipSuccess) {                                            \
      s2svc_set_error(name);                                           \
      s2svc_set_error(hipGetErrorString(_e));                          \
      return -2;                                                       \
    }                                                                  \
  } while (0)
#define S2S_REQUIRE(cond, msg)                                         \
  do {                                                                 \
    if (!(cond)) { s2svc_set_error(msg); return -1; }                  \
  } while (0)
