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

// Cross-lane reductions WITHOUT the LDS crossbar (round 5).  hipcc lowers every __shfl_xor to ds_bpermute_b32 (~100 cycles of
// dependent latency each: 12 of them in a LayerNorm row, 4 per softmax row of the fused attention kernels -- a visible part of
// kernels that run 3-5 us).  The partner at lane ^ 1 / ^ 2 is a DPP quad permute, at ^ 4 two bank-masked DPP row shifts, at ^ 8 a
// DPP row rotation, at ^ 16 / ^ 32 gfx950's v_permlane16_swap / v_permlane32_swap (with both operands = v the two results hold
// (even rows, even rows) / (odd rows, odd rows), resp. (low half, low half) / (high half, high half)): VALU instructions, and the
// SAME partners in the SAME order as the butterflies they replace, so every sum keeps its bits.  ALL 64 LANES MUST BE ACTIVE (they
// are: every caller reduces under wave-uniform control flow): with an inactive partner a swap returns the lane's own value twice
// (ds_bpermute returned 0), so a partial-wave sum would double-count.  A build with -DS2SVC_DEBUG_EXEC (_lib.build_library(debug_exec=True)
// -> libs2svc_hip_dbgexec.so, made by __graft_entry__.build(); tests/gpu_kernel_check.py: debug_exec_build_runs_clean) traps in swap16 / swap32 / the DPP moves
// when EXEC is not all ones.
#ifdef S2SVC_DEBUG_EXEC
#define S2S_ASSERT_FULL_EXEC()                                              \
  do {                                                                      \
    if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();            \
  } while (0)
#else
#define S2S_ASSERT_FULL_EXEC() ((void)0)
#endif
template <int CTRL, int BANKS = 0xf>
__device__ __forceinline__ float dpp_mov(float v, float old = 0.f) {
  S2S_ASSERT_FULL_EXEC();
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, BANKS, false));
}
__device__ __forceinline__ float xor1_of(float v) { return dpp_mov<0xB1>(v); }             // quad_perm:[1,0,3,2]
__device__ __forceinline__ float xor2_of(float v) { return dpp_mov<0x4E>(v); }             // quad_perm:[2,3,0,1]
__device__ __forceinline__ float xor4_of(float v) {                                        // quads 0 <-> 1, 2 <-> 3 of every row
  const float t = dpp_mov<0x104, 0x5>(v);                                                  // row_shl:4 into banks 0, 2
  return dpp_mov<0x114, 0xa>(v, t);                                                        // row_shr:4 into banks 1, 3
}
__device__ __forceinline__ float xor8_of(float v) { return dpp_mov<0x128>(v); }            // row_ror:8
struct lane_pair { float a, b; };
__device__ __forceinline__ lane_pair swap16(float v) {
  S2S_ASSERT_FULL_EXEC();
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return {__uint_as_float(r[0]), __uint_as_float(r[1])};
}
__device__ __forceinline__ lane_pair swap32(float v) {
  S2S_ASSERT_FULL_EXEC();
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return {__uint_as_float(r[0]), __uint_as_float(r[1])};
}
// v + partner: of the two swap results one is the lane's own value and the other its partner's, so a + b is the butterfly step
__device__ __forceinline__ float xor1_sum(float v) { return v + xor1_of(v); }
__device__ __forceinline__ float xor2_sum(float v) { return v + xor2_of(v); }
__device__ __forceinline__ float xor4_sum(float v) { return v + xor4_of(v); }
__device__ __forceinline__ float xor8_sum(float v) { return v + xor8_of(v); }
__device__ __forceinline__ float xor16_sum(float v) { const lane_pair p = swap16(v); return p.a + p.b; }
__device__ __forceinline__ float xor32_sum(float v) { const lane_pair p = swap32(v); return p.a + p.b; }
__device__ __forceinline__ float row16_sum(float v) {        // every lane of a 16-lane row gets the row's sum (partners 1, 2, 4, 8)
  return xor8_sum(xor4_sum(xor2_sum(xor1_sum(v))));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, xor1_of(v));
  v = fmaxf(v, xor2_of(v));
  v = fmaxf(v, xor4_of(v));
  return fmaxf(v, xor8_of(v));
}
__device__ __forceinline__ float wave_sum(float v) {         // partners 32, 16, 8, 4, 2, 1: the order of the old butterfly
  return xor1_sum(xor2_sum(xor4_sum(xor8_sum(xor16_sum(xor32_sum(v))))));
}
__device__ __forceinline__ float wave_max(float v) {
  lane_pair p = swap32(v);
  v = fmaxf(p.a, p.b);
  p = swap16(v);
  v = fmaxf(p.a, p.b);
  v = fmaxf(v, xor8_of(v));
  v = fmaxf(v, xor4_of(v));
  v = fmaxf(v, xor2_of(v));
  return fmaxf(v, xor1_of(v));
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

// ---- absent rows (captured training steps on batches that do not fill their padded shape) ----
// A (B, Tn, C) activation of a captured step is allocated at the PADDED length Tn, while the reference computes on the batch cropped to
// its longest utterance (models/vtn.py:208-214, models/aas_vc.py:523-524).  Kernels that mix along time or over the batch (BatchNorm
// statistics, the depthwise convolution of the Conformer module, nearest-neighbour resampling) take `vlens` (B int32, device memory,
// graph DATA: modules.LensBank): row r = b * Tn + t is ABSENT when t >= vlens[b] -- it is excluded from every sum and count, read as
// the convolution's zero padding, and written as zero on the way out (forward and backward).  vlens == nullptr: every row is present.
__device__ __forceinline__ bool row_present(int r, int Tn, const int32_t* __restrict__ vlens) {
  if (!vlens) return true;
  const int b = r / Tn;
  return r - b * Tn < vlens[b];
}
// number of present rows (every thread computes it: B is a batch size, the loads are uniform and cached)
__device__ __forceinline__ int rows_present(int rows, int Tn, const int32_t* __restrict__ vlens) {
  if (!vlens) return rows;
  int n = 0;
  for (int b = 0; b < rows / Tn; ++b) n += vlens[b] < Tn ? (vlens[b] > 0 ? vlens[b] : 0) : Tn;
  return n;
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
