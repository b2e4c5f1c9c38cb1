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
// float -> bf16 with the hardware converter (v_cvt_pk_bf16_f32, gfx950): round-to-nearest-even, NaN stays NaN -- the values
// torch's float->bfloat16 cast produces.  One instruction per PAIR; the integer formulation it replaces (~12 VALU
// instructions per value with its NaN branch) was a third of the code of the small bf16 kernels, and these run cold code
// (see tools/kernel_code_sizes.py).
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {          // two values -> one dword (lo in bits 0..15)
  typedef __attribute__((ext_vector_type(2))) __bf16 s2s_bf2_t;
  typedef __attribute__((ext_vector_type(2))) float s2s_f2_t;
  const s2s_f2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, s2s_bf2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(f2bf2(f, 0.f) & 0xffffu); }

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
// Counter-based RNG for dropout masks: a mask is a pure function of (seed, element index), so backward kernels
// regenerate it instead of reading it from HBM.  One 64-bit draw = SplitMix64's output function (Stafford "Mix13")
// applied to seed + (group+1)*golden-ratio, i.e. element `group` of the SplitMix64 stream started at `seed`; it
// covers the 4 consecutive element indices 4*group..4*group+3 with 16 bits each (drop when field < p * 2^16).
// (The first version used Philox-4x32-10 and spent one full 10-round call PER ELEMENT at the scalar call sites:
// ~180 us of the 6.5 ms VTN step went to mask generation; this mixer is ~6x cheaper per element.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t dropout_draw(uint64_t seed, uint64_t group) {
  uint64_t z = seed + (group + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t dropout_threshold(float p) { return (uint32_t)(p * 65536.0f); }
// keep-scale for element idx: 0 (dropped) or 1/(1-p)
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t idx, float p, float inv_keep) {
  const uint32_t w = (uint32_t)(dropout_draw(seed, idx >> 2) >> ((idx & 3) * 16)) & 0xffffu;
  return w < dropout_threshold(p) ? 0.f : inv_keep;
}

// ---- 16-byte (8 x bf16 / 2 x 4 x fp32) register <-> memory helpers of the vectorised element-wise kernels ----
__device__ __forceinline__ void unpack_bf16x8(const uint4& v, float (&f)[8]) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { f[2 * e] = __uint_as_float(w[e] << 16); f[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 pack_bf16x8(const float (&f)[8]) {
  uint4 v;
  v.x = f2bf2(f[0], f[1]);
  v.y = f2bf2(f[2], f[3]);
  v.z = f2bf2(f[4], f[5]);
  v.w = f2bf2(f[6], f[7]);
  return v;
}
__device__ __forceinline__ void load_f32x8(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
// keep-scales of 8 consecutive elements starting at idx (idx % 8 == 0): two draws, the fields dropout_scale() reads
__device__ __forceinline__ void dropout_scale8(uint64_t seed, uint64_t idx, float p, float inv_keep, float (&m)[8]) {
  const uint32_t thr = dropout_threshold(p);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const uint64_t r = dropout_draw(seed, (idx >> 2) + q);
#pragma unroll
    for (int e = 0; e < 4; ++e) m[4 * q + e] = ((uint32_t)(r >> (16 * e)) & 0xffffu) < thr ? 0.f : inv_keep;
  }
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
