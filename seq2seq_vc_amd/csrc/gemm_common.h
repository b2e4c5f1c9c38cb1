// Shared by the MFMA GEMM kernels (gemm_fast.hip, gemm_skinny.hip): fragment types and the output epilogue.
#pragma once
#include "common.h"
#include "../../include/s2svc_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// C[m, n] (batch z0, z1) = act(alpha * v + bias[n]) + res[m, n]  (+ C when accumulating), stored in c_dtype
__device__ __forceinline__ void epilogue_store_f(const s2svc_gemm_desc& d, int z0, int z1, int m, int n, float v) {
  v *= d.alpha;
  if (d.bias) v += d.bias[n];
  v = act_apply(v, d.act);
  const int64_t co = (int64_t)z0 * d.cbs0 + (int64_t)z1 * d.cbs1 + (int64_t)m * d.ldc + n;
  if (d.res) {
    const int64_t ro = (int64_t)z0 * d.rbs0 + (int64_t)z1 * d.rbs1 + (int64_t)m * d.ldr + n;
    v += d.c_dtype == S2S_F32 ? ((const float*)d.res)[ro] : bf2f(((const bf16_t*)d.res)[ro]);
  }
  if (d.c_dtype == S2S_F32) {
    float* c = (float*)d.C + co;
    *c = d.accumulate ? *c + v : v;
  } else {
    bf16_t* c = (bf16_t*)d.C + co;
    *c = f2bf(d.accumulate ? bf2f(*c) + v : v);
  }
}
