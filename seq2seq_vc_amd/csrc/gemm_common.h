// Shared by the MFMA GEMM kernels (gemm_fast.hip, gemm_skinny.hip): fragment types and the output epilogue.
#pragma once
#include "common.h"
#include "../../include/s2svc_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// d swish(x) / dx from the pre-activation (the formula of act_dropout_bwd, elementwise.hip)
__device__ __forceinline__ float swish_grad(float x) {
  const float sg = 1.f / (1.f + expf(-x));
  return sg * (1.f + x * (1.f - sg));
}

// the optional stage between activation and residual: v * dropmask(m*N + n) * (emask[m, n] > 0  |  swish'(emask[m, n]))
__device__ __forceinline__ float epilogue_stage_f(const s2svc_gemm_desc& d, int m, int n, float v) {
  if (d.drop_p > 0.f) {
    const uint64_t seed = (d.seed_base ? *d.seed_base : 0ull) + d.seed_off;
    v *= dropout_scale(seed, (uint64_t)((int64_t)m * d.N + n), d.drop_p, 1.f / (1.f - d.drop_p));
  }
  if (d.emask) {
    int64_t er = m;                      // emask has C's row layout: with a c_map its rows are the mapped rows (see c_row_of)
    if (d.c_map) {
      const int per_b = d.cm_Tc * d.cm_Fc;
      const int b = m / per_b, rem = m - b * per_b;
      const int i = rem / d.cm_Fc, j = rem - i * d.cm_Fc;
      er = ((int64_t)b * d.cm_T1 + 2 * i + d.cm_pt) * d.cm_F1 + 2 * j + d.cm_pf;
    }
    const int64_t mo = er * d.ldm + n;
    const float e = d.c_dtype == S2S_F32 ? ((const float*)d.emask)[mo] : bf2f(((const bf16_t*)d.emask)[mo]);
    v = d.emask_mode == 1 ? v * swish_grad(e) : (e > 0.f ? v : 0.f);
  }
  return v;
}

// C[m, n] (batch z0, z1) = act(alpha * v + bias[n]) [stage] + res[m, n]  (+ C when accumulating), stored in c_dtype.
// STAGED = false is what the register-staged kernels (gemm.hip, gemm_fast.hip, gemm_skinny.hip) inline 64x into
// their unrolled accumulator loops: it must stay small, or hipcc stops unrolling, indexes the accumulators at run time
// and demotes them to scratch memory.  Those kernels get the stage from gemm_stage_kernel (gemm.hip) as a second pass.
// row of C that GEMM row m is stored at (identity unless the descriptor carries a c_map, see s2svc_hip.h)
__device__ __forceinline__ int64_t c_row_of(const s2svc_gemm_desc& d, int m) {
  if (!d.c_map) return m;
  const int per_b = d.cm_Tc * d.cm_Fc;
  const int b = m / per_b, rem = m - b * per_b;
  const int i = rem / d.cm_Fc, j = rem - i * d.cm_Fc;
  return ((int64_t)b * d.cm_T1 + 2 * i + d.cm_pt) * d.cm_F1 + 2 * j + d.cm_pf;
}

// Division of row indices by image extents without the ~25-instruction integer-division sequence: q = mulhi(n, magic) is
// exact for n * d < 2^32 (the launchers check that bound on the host, see rows_fit_fastdiv in gemm_glds.hip).
struct fastdiv_t { uint32_t d, mg; };
__device__ __forceinline__ fastdiv_t fastdiv_make(int d) {
  fastdiv_t f;
  f.d = (uint32_t)d;
  f.mg = d > 1 ? 0xFFFFFFFFu / (uint32_t)d + 1u : 0u;
  return f;
}
__device__ __forceinline__ int fastdiv(int n, const fastdiv_t& f) { return f.d > 1 ? (int)__umulhi((uint32_t)n, f.mg) : n; }

struct c_map_t { fastdiv_t per_b, fc; };
__device__ __forceinline__ c_map_t c_map_make(const s2svc_gemm_desc& d) {
  c_map_t c;
  c.per_b = fastdiv_make(d.c_map ? d.cm_Tc * d.cm_Fc : 1);
  c.fc = fastdiv_make(d.c_map ? d.cm_Fc : 1);
  return c;
}
__device__ __forceinline__ int64_t c_row_fast(const s2svc_gemm_desc& d, const c_map_t& c, int m) {
  if (!d.c_map) return m;
  const int b = fastdiv(m, c.per_b), rem = m - b * (int)c.per_b.d;
  const int i = fastdiv(rem, c.fc), j = rem - i * (int)c.fc.d;
  return ((int64_t)b * d.cm_T1 + 2 * i + d.cm_pt) * d.cm_F1 + 2 * j + d.cm_pf;
}

template <bool STAGED = false>
__device__ __forceinline__ void epilogue_store_f(const s2svc_gemm_desc& d, int z0, int z1, int m, int n, float v) {
  v *= d.alpha;
  if (d.bias) v += d.bias[n];
  if (d.c_pre) {
    const int64_t po = (int64_t)z0 * d.cbs0 + (int64_t)z1 * d.cbs1 + (int64_t)m * d.ldc + n;
    if (d.c_dtype == S2S_F32) ((float*)d.c_pre)[po] = v;
    else ((bf16_t*)d.c_pre)[po] = f2bf(v);
  }
  v = act_apply(v, d.act);
  if (STAGED) v = epilogue_stage_f(d, m, n, v);
  const int64_t co = (int64_t)z0 * d.cbs0 + (int64_t)z1 * d.cbs1 + (STAGED ? c_row_of(d, m) : (int64_t)m) * d.ldc + n;
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

// ---------------------------------------------------------------------------------------------------------
// Vectorised epilogue: a wavefront's WTM x WTN accumulator tile goes through a wave-private fp32 LDS tile so that
// every lane ends up with 8 consecutive columns of one row -> bias / activation / residual on 8 values and ONE
// 16-byte (bf16) or two 16-byte (fp32) global stores per lane, 128+ contiguous bytes per row segment.  (The MFMA
// accumulator layout gives a lane 4 rows x 1 column per fragment: stored directly that is 2-byte scattered stores,
// which dominates short-K GEMMs.)  Same arithmetic as epilogue_store_f (fp32, one rounding).
// Requires (checked by the caller, uniform): N % 8 == 0, ldc / batch strides % 8 == 0, 16-byte aligned C (and res).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool epilogue_vec_ok(const s2svc_gemm_desc& d) {
  bool ok = (d.N % 8 == 0) && (d.ldc % 8 == 0) && (d.cbs0 % 8 == 0) && (d.cbs1 % 8 == 0) && (((uintptr_t)d.C) % 16 == 0);
  if (d.res) ok = ok && (d.ldr % 8 == 0) && (d.rbs0 % 8 == 0) && (d.rbs1 % 8 == 0) && (((uintptr_t)d.res) % 16 == 0);
  if (d.emask) ok = ok && (d.ldm % 8 == 0) && (((uintptr_t)d.emask) % 16 == 0);
  if (d.c_pre) ok = ok && (((uintptr_t)d.c_pre) % 16 == 0);
  return ok;
}

// The two halves of epilogue_tile: `stage` copies a wave's accumulator tile into its fp32 LDS tile (register indices are
// compile-time constants: it must be expanded per sub-tile), `flush` does everything else from LDS.  A kernel with several
// sub-tiles per wave (gemm_8ph.hip) stages them inside a `switch` of a ROLLED loop and shares ONE copy of the flush code: the
// epilogue is cold code (the instruction cache is invalidated per dispatch) and four inlined copies of it were most of the
// "fixed" ~11 us of every launch of those kernels.
template <int WTM, int WTN>
__device__ __forceinline__ void epilogue_stage(const f32x4_t (&acc)[WTM / 16][WTN / 16], float* cs) {
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int i = 0; i < WTM / 16; ++i)
#pragma unroll
    for (int j = 0; j < WTN / 16; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i * 16 + lg * 4 + r, col = j * 16 + lr;
        cs[row * WTN + (col ^ (((row >> 2) & (WTN / 16 - 1)) << 4))] = acc[i][j][r];
      }
}

template <int WTM, int WTN>
__device__ __forceinline__ void epilogue_flush(const s2svc_gemm_desc& d, int z0, int z1, int m_base, int n_base, const float* cs,
                                               int splitk, int zs, int zb);

template <int WTM, int WTN>
__device__ __forceinline__ void epilogue_tile(const s2svc_gemm_desc& d, int z0, int z1, int m_base, int n_base,
                                              const f32x4_t (&acc)[WTM / 16][WTN / 16], float* cs, int splitk, int zs, int zb) {
  epilogue_stage<WTM, WTN>(acc, cs);
  epilogue_flush<WTM, WTN>(d, z0, z1, m_base, n_base, cs, splitk, zs, zb);
}

template <int WTM, int WTN>
__device__ __forceinline__ void epilogue_flush(const s2svc_gemm_desc& d, int z0, int z1, int m_base, int n_base, const float* cs,
                                               int splitk, int zs, int zb) {
  const int lane = threadIdx.x & 63;
  if (splitk > 1 || !epilogue_vec_ok(d)) {
    // element-wise path (split-K partials, unaligned C): still read back from LDS, so that the accumulator registers
    // are only ever indexed by compile-time constants (a runtime-indexed acc[][] is demoted to scratch memory and
    // spilled inside the K loop) and consecutive lanes touch consecutive columns
#pragma unroll 1
    for (int e = lane; e < WTM * WTN; e += 64) {
      const int row = e / WTN, col = e - row * WTN;
      const int m = m_base + row, n = n_base + col;
      if (m >= d.M || n >= d.N) continue;
      const float v = cs[row * WTN + (col ^ (((row >> 2) & (WTN / 16 - 1)) << 4))];
      if (splitk > 1) d.ws[(((int64_t)zs * (d.nb0 * d.nb1) + zb) * d.M + m) * d.N + n] = v;
      else epilogue_store_f<true>(d, z0, z1, m, n, v);
    }
    return;
  }
  constexpr int LPR = WTN / 8;          // lanes per row
  constexpr int RPP = 64 / LPR;         // rows per pass
  const c_map_t cm = c_map_make(d);
  // NOT unrolled: the body is load/store bound, and unrolling it (8 passes x 8 values x RNG state) raised the
  // register demand of the whole kernel until hipcc spilled the MFMA accumulators inside the K loop (5x slower)
#pragma unroll 1
  for (int p = 0; p < (WTM + RPP - 1) / RPP; ++p) {       // (a 16 x 16 wave tile is half a pass: rows >= WTM idle)
    const int row = p * RPP + lane / LPR, col = (lane % LPR) * 8;
    const int m = m_base + row, n = n_base + col;
    if ((WTM % RPP != 0 && row >= WTM) || m >= d.M || n >= d.N) continue;
    const float* src = cs + row * WTN + (col ^ (((row >> 2) & (WTN / 16 - 1)) << 4));
    const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if (d.bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(d.bias + n), b1 = *reinterpret_cast<const float4*>(d.bias + n + 4);
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * d.alpha + bb[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= d.alpha;
    }
    if (d.c_pre) {
      const int64_t po = (int64_t)z0 * d.cbs0 + (int64_t)z1 * d.cbs1 + (int64_t)m * d.ldc + n;
      if (d.c_dtype == S2S_F32) {
        float* q = (float*)d.c_pre + po;
        *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(q + 4) = make_float4(v[4], v[5], v[6], v[7]);
      } else {
        uint4 o;
        o.x = f2bf2(v[0], v[1]);
        o.y = f2bf2(v[2], v[3]);
        o.z = f2bf2(v[4], v[5]);
        o.w = f2bf2(v[6], v[7]);
        *reinterpret_cast<uint4*>((bf16_t*)d.c_pre + po) = o;
      }
    }
    if (d.act != S2S_ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = act_apply(v[e], d.act);
    }
    if (d.drop_p > 0.f) {
      const uint64_t seed = (d.seed_base ? *d.seed_base : 0ull) + d.seed_off;
      const float inv_keep = 1.f / (1.f - d.drop_p);
      const uint64_t idx = (uint64_t)((int64_t)m * d.N + n);
      // idx is a multiple of 8 (N % 8 == 0, col % 8 == 0): two 64-bit draws cover the 8 values, exactly the fields
      // dropout_scale(seed, idx + e) reads for them
      const uint32_t thr = dropout_threshold(d.drop_p);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint64_t r = dropout_draw(seed, (idx >> 2) + q);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] *= ((uint32_t)(r >> (16 * e)) & 0xffffu) < thr ? 0.f : inv_keep;
      }
    }
    const int64_t crow = c_row_fast(d, cm, m);
    if (d.emask) {
      const int64_t mo = crow * d.ldm + n;            // emask has C's row layout (mapped rows under a c_map)
      if (d.c_dtype == S2S_F32) {
        const float4 e0 = *reinterpret_cast<const float4*>((const float*)d.emask + mo), e1 = *reinterpret_cast<const float4*>((const float*)d.emask + mo + 4);
        const float ee[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = d.emask_mode == 1 ? v[e] * swish_grad(ee[e]) : (ee[e] > 0.f ? v[e] : 0.f);
      } else {
        const uint4 ev = *reinterpret_cast<const uint4*>((const bf16_t*)d.emask + mo);
        const uint32_t w[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float e0 = __uint_as_float(w[e] << 16), e1 = __uint_as_float(w[e] & 0xffff0000u);
          v[2 * e] = d.emask_mode == 1 ? v[2 * e] * swish_grad(e0) : (e0 > 0.f ? v[2 * e] : 0.f);
          v[2 * e + 1] = d.emask_mode == 1 ? v[2 * e + 1] * swish_grad(e1) : (e1 > 0.f ? v[2 * e + 1] : 0.f);
        }
      }
    }
    const int64_t co = (int64_t)z0 * d.cbs0 + (int64_t)z1 * d.cbs1 + crow * d.ldc + n;
    if (d.res) {
      const int64_t ro = (int64_t)z0 * d.rbs0 + (int64_t)z1 * d.rbs1 + (int64_t)m * d.ldr + n;
      if (d.c_dtype == S2S_F32) {
        const float4 r0 = *reinterpret_cast<const float4*>((const float*)d.res + ro), r1 = *reinterpret_cast<const float4*>((const float*)d.res + ro + 4);
        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
      } else {
        const uint4 rv = *reinterpret_cast<const uint4*>((const bf16_t*)d.res + ro);
        const uint32_t w[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
      }
    }
    if (d.c_dtype == S2S_F32) {
      float* c = (float*)d.C + co;
      if (d.accumulate) {
        const float4 c0 = *reinterpret_cast<const float4*>(c), c1 = *reinterpret_cast<const float4*>(c + 4);
        v[0] += c0.x; v[1] += c0.y; v[2] += c0.z; v[3] += c0.w; v[4] += c1.x; v[5] += c1.y; v[6] += c1.z; v[7] += c1.w;
      }
      *reinterpret_cast<float4*>(c) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      bf16_t* c = (bf16_t*)d.C + co;
      if (d.accumulate) {
        const uint4 cv = *reinterpret_cast<const uint4*>(c);
        const uint32_t w[4] = {cv.x, cv.y, cv.z, cv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
      }
      uint4 o;
      o.x = f2bf2(v[0], v[1]);
      o.y = f2bf2(v[2], v[3]);
      o.z = f2bf2(v[4], v[5]);
      o.w = f2bf2(v[6], v[7]);
      *reinterpret_cast<uint4*>(c) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The ROW-MAPPED epilogue of the transposed-convolution classes (Conv2d data gradient: gemm_8ph.hip P8_TCONV2D): c_map rows, an
// optional ReLU mask with C's row layout, bf16 C, nothing else.  The general flush loads the mask inside its rolled row loop --
// a dependent trip to HBM per pass, 16 per wave of the 512 x 128 tile; its "fixed" cost was 30 us per class launch (6 K tiles: 38.7 us
// stand-alone; the four classes 223 us with the mask against 164 without).  Here the passes of a sub-tile are unrolled and all of
// their mask vectors are requested BEFORE the accumulators are read back from LDS: one exposed trip per sub-tile.
// ---------------------------------------------------------------------------------------------------------
__host__ __device__ inline bool epilogue_cmap_ok(const s2svc_gemm_desc& d) {
  if (!d.c_map || d.c_dtype != S2S_BF16 || d.nb0 * d.nb1 != 1 || d.splitk > 1 || d.alpha != 1.0f || d.c_pre || d.accumulate) return false;
  if (d.bias || d.res || d.act != S2S_ACT_NONE || d.drop_p > 0.f) return false;
  if (d.emask && (d.emask_mode != 0 || d.ldm != d.ldc)) return false;
  if (d.N % 8 || d.ldc % 8 || ((uintptr_t)d.C) % 16 || (d.emask && ((uintptr_t)d.emask) % 16)) return false;
  return true;
}

template <int WTM, int WTN>
__device__ __forceinline__ void epilogue_flush_cmap(const s2svc_gemm_desc& d, const c_map_t& cm, int m_base, int n_base, const float* cs) {
  const int lane = threadIdx.x & 63;
  constexpr int LPR = WTN / 8, RPP = 64 / LPR, NP = WTM / RPP;
  const int col = (lane % LPR) * 8, n = n_base + col;
  if (n >= d.N) return;
  int64_t off[NP];
  uint4 ev[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int m = m_base + p * RPP + lane / LPR;
    off[p] = m < d.M ? c_row_fast(d, cm, m) * d.ldc + n : -1;
    ev[p] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    if (d.emask && off[p] >= 0) ev[p] = *reinterpret_cast<const uint4*>((const bf16_t*)d.emask + off[p]);
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    if (off[p] < 0) continue;
    const int row = p * RPP + lane / LPR;
    const float* src = cs + row * WTN + (col ^ (((row >> 2) & (WTN / 16 - 1)) << 4));
    const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const uint32_t w[4] = {ev[p].x, ev[p].y, ev[p].z, ev[p].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = __uint_as_float(w[e] << 16) > 0.f ? v[2 * e] : 0.f;
      v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u) > 0.f ? v[2 * e + 1] : 0.f;
    }
    uint4 o;
    o.x = f2bf2(v[0], v[1]);
    o.y = f2bf2(v[2], v[3]);
    o.z = f2bf2(v[4], v[5]);
    o.w = f2bf2(v[6], v[7]);
    // non-temporal: each class writes its quarter of a 122 MB gradient once; nobody re-reads it before it has left the L2 (the four classes:
    // 195 -> 172 us with these stores; bit-identical)
    typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
    u32x4_nt t = {o.x, o.y, o.z, o.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4_nt*>((bf16_t*)d.C + off[p]));
  }
}

// ---------------------------------------------------------------------------------------------------------
// The COMMON epilogue: what the Linear layers of the training chains use and nothing else -- bias, ReLU, dropout, a ReLU mask
// (emask mode 0), a residual, bf16 C, one problem (no batch, split-K, alpha, pre-activation output, row map, accumulation,
// fp32 C, unaligned tails).  Same arithmetic and the same masks as epilogue_flush.  Why a second copy: the general flush carries
// the tanh / GELU / sigmoid / Swish expansions (8-fold unrolled), Swish' masks, the fp32 and accumulate paths and the
// element-wise fallback -- 17 of the 20.8 KB of gemm_dma_kernel<32, 64> -- and a launch-bound kernel pays for code it never
// runs (2016 x 384 x 384: 5.6 -> 5.1 us with a kernel of 3.2 KB; tools/kernel_code_sizes.py).  epilogue_common_ok() is the
// host-side test.
// ---------------------------------------------------------------------------------------------------------
inline bool epilogue_common_ok(const s2svc_gemm_desc& d) {
  if (d.c_dtype != S2S_BF16 || d.nb0 * d.nb1 != 1 || d.splitk > 1 || d.alpha != 1.0f || d.c_pre || d.c_map || d.accumulate) return false;
  if (d.act != S2S_ACT_NONE && d.act != S2S_ACT_RELU) return false;
  if (d.emask && d.emask_mode != 0) return false;
  if (d.N % 8 || d.ldc % 8 || ((uintptr_t)d.C) % 16) return false;
  if (d.res && (d.ldr % 8 || ((uintptr_t)d.res) % 16)) return false;
  if (d.emask && (d.ldm % 8 || ((uintptr_t)d.emask) % 16)) return false;
  return true;
}

template <int WTM, int WTN>
__device__ __forceinline__ void epilogue_flush_common(const s2svc_gemm_desc& d, int m_base, int n_base, const float* cs) {
  const int lane = threadIdx.x & 63;
  constexpr int LPR = WTN / 8, RPP = 64 / LPR;
  // a lane keeps its 8 columns through the row loop: the bias is loaded ONCE, ahead of the loop (the rolled loop otherwise pays a
  // dependent global load per iteration for the same eight values)
  const int col = (lane % LPR) * 8, n = n_base + col;
  if (n >= d.N) return;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (d.bias) {
    b0 = *reinterpret_cast<const float4*>(d.bias + n);
    b1 = *reinterpret_cast<const float4*>(d.bias + n + 4);
  }
#pragma unroll 1
  for (int p = 0; p < WTM / RPP; ++p) {
    const int row = p * RPP + lane / LPR;
    const int m = m_base + row;
    if (m >= d.M) continue;
    const float* src = cs + row * WTN + (col ^ (((row >> 2) & (WTN / 16 - 1)) << 4));
    const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if (d.bias) {
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (d.act == S2S_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    if (d.drop_p > 0.f) {
      const uint64_t seed = (d.seed_base ? *d.seed_base : 0ull) + d.seed_off;
      const float inv_keep = 1.f / (1.f - d.drop_p);
      const uint64_t idx = (uint64_t)((int64_t)m * d.N + n);
      const uint32_t thr = dropout_threshold(d.drop_p);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint64_t r = dropout_draw(seed, (idx >> 2) + q);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] *= ((uint32_t)(r >> (16 * e)) & 0xffffu) < thr ? 0.f : inv_keep;
      }
    }
    if (d.emask) {
      const uint4 ev = *reinterpret_cast<const uint4*>((const bf16_t*)d.emask + (int64_t)m * d.ldm + n);
      const uint32_t w[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] = __uint_as_float(w[e] << 16) > 0.f ? v[2 * e] : 0.f;
        v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u) > 0.f ? v[2 * e + 1] : 0.f;
      }
    }
    if (d.res) {
      const uint4 rv = *reinterpret_cast<const uint4*>((const bf16_t*)d.res + (int64_t)m * d.ldr + n);
      const uint32_t w[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
    }
    uint4 o;
    o.x = f2bf2(v[0], v[1]);
    o.y = f2bf2(v[2], v[3]);
    o.z = f2bf2(v[4], v[5]);
    o.w = f2bf2(v[6], v[7]);
    *reinterpret_cast<uint4*>((bf16_t*)d.C + (int64_t)m * d.ldc + n) = o;
  }
}

// ---------------------------------------------------------------------------------------------------------
// The SWISH epilogues of the Conformer feed-forward blocks (bf16, one problem): forward = bias, pre-activation as a second
// output (c_pre), Swish, dropout; data gradient through w_2 = dropout mask x swish'(pre-activation) (emask mode 1).  The general
// flush spends ~9 us per 4096 x 1536 output on them (IEEE divisions, expf, its 19 KB of code) -- 32 launches per AAS-VC step;
// here the sigmoid is v_exp_f32 + v_rcp_f32 (1 ulp; the results are rounded to bf16) and nothing else is carried.
// ---------------------------------------------------------------------------------------------------------
inline bool epilogue_swish_ok(const s2svc_gemm_desc& d) {
  if (d.c_dtype != S2S_BF16 || d.nb0 * d.nb1 != 1 || d.splitk > 1 || d.alpha != 1.0f || d.c_map || d.accumulate || d.res) return false;
  if (d.N % 8 || d.ldc % 8 || ((uintptr_t)d.C) % 16) return false;
  const bool fwd = d.act == S2S_ACT_SWISH && d.c_pre && !d.emask && ((uintptr_t)d.c_pre) % 16 == 0;
  const bool bwd = d.act == S2S_ACT_NONE && !d.c_pre && d.emask && d.emask_mode == 1 && d.ldm % 8 == 0 && ((uintptr_t)d.emask) % 16 == 0;
  return fwd || bwd;
}

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

template <int WTM, int WTN>
__device__ __forceinline__ void epilogue_flush_swish(const s2svc_gemm_desc& d, int m_base, int n_base, const float* cs) {
  const int lane = threadIdx.x & 63;
  constexpr int LPR = WTN / 8, RPP = 64 / LPR;
  // a lane keeps its 8 columns through the row loop: the bias is loaded ONCE, ahead of the loop (the rolled loop otherwise pays a
  // dependent global load per iteration for the same eight values)
  const int col = (lane % LPR) * 8, n = n_base + col;
  if (n >= d.N) return;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (d.bias) {
    b0 = *reinterpret_cast<const float4*>(d.bias + n);
    b1 = *reinterpret_cast<const float4*>(d.bias + n + 4);
  }
#pragma unroll 1
  for (int p = 0; p < WTM / RPP; ++p) {
    const int row = p * RPP + lane / LPR;
    const int m = m_base + row;
    if (m >= d.M) continue;
    const float* src = cs + row * WTN + (col ^ (((row >> 2) & (WTN / 16 - 1)) << 4));
    const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if (d.bias) {
      v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    }
    if (d.c_pre) {                       // forward: the pre-activation leaves as it is, then Swish
      uint4 o;
      o.x = f2bf2(v[0], v[1]);
      o.y = f2bf2(v[2], v[3]);
      o.z = f2bf2(v[4], v[5]);
      o.w = f2bf2(v[6], v[7]);
      *reinterpret_cast<uint4*>((bf16_t*)d.c_pre + (int64_t)m * d.ldc + n) = o;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= fast_sigmoid(v[e]);
    }
    if (d.drop_p > 0.f) {
      const uint64_t seed = (d.seed_base ? *d.seed_base : 0ull) + d.seed_off;
      const float inv_keep = 1.f / (1.f - d.drop_p);
      const uint64_t idx = (uint64_t)((int64_t)m * d.N + n);
      const uint32_t thr = dropout_threshold(d.drop_p);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint64_t r = dropout_draw(seed, (idx >> 2) + q);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] *= ((uint32_t)(r >> (16 * e)) & 0xffffu) < thr ? 0.f : inv_keep;
      }
    }
    if (d.emask) {                       // data gradient: x swish'(pre-activation)
      const uint4 ev = *reinterpret_cast<const uint4*>((const bf16_t*)d.emask + (int64_t)m * d.ldm + n);
      const uint32_t w[4] = {ev.x, ev.y, ev.z, ev.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float u0 = __uint_as_float(w[e] << 16), u1 = __uint_as_float(w[e] & 0xffff0000u);
        const float s0 = fast_sigmoid(u0), s1 = fast_sigmoid(u1);
        v[2 * e] *= s0 * (1.f + u0 * (1.f - s0));
        v[2 * e + 1] *= s1 * (1.f + u1 * (1.f - s1));
      }
    }
    uint4 o;
    o.x = f2bf2(v[0], v[1]);
    o.y = f2bf2(v[2], v[3]);
    o.z = f2bf2(v[4], v[5]);
    o.w = f2bf2(v[6], v[7]);
    *reinterpret_cast<uint4*>((bf16_t*)d.C + (int64_t)m * d.ldc + n) = o;
  }
}

// the fp32 counterpart (the duration predictor's Linear layers and their gradients: bias, ReLU, fp32 residual, accumulation into
// a gradient slot; no dropout / mask)
inline bool epilogue_common32_ok(const s2svc_gemm_desc& d) {       // (split-K: the kernel stores raw partials, see epilogue_partials)
  if (d.c_dtype != S2S_F32 || d.nb0 * d.nb1 != 1 || d.alpha != 1.0f || d.c_pre || d.c_map) return false;
  if (d.act != S2S_ACT_NONE && d.act != S2S_ACT_RELU) return false;
  if (d.emask || d.drop_p > 0.f) return false;
  if (d.N % 8 || d.ldc % 4 || ((uintptr_t)d.C) % 16) return false;
  if (d.res && (d.ldr % 4 || ((uintptr_t)d.res) % 16)) return false;
  if (d.bias && ((uintptr_t)d.bias) % 16) return false;
  return true;
}

template <int WTM, int WTN>
__device__ __forceinline__ void epilogue_flush_common32(const s2svc_gemm_desc& d, int m_base, int n_base, const float* cs) {
  const int lane = threadIdx.x & 63;
  constexpr int LPR = WTN / 8, RPP = 64 / LPR;
  const int col = (lane % LPR) * 8, n = n_base + col;        // (bias: once per lane, ahead of the rolled row loop -- see above)
  if (n >= d.N) return;
  float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
  if (d.bias) {
    b0 = *reinterpret_cast<const float4*>(d.bias + n);
    b1 = *reinterpret_cast<const float4*>(d.bias + n + 4);
  }
#pragma unroll 1
  for (int p = 0; p < (WTM + RPP - 1) / RPP; ++p) {
    const int row = p * RPP + lane / LPR;
    const int m = m_base + row;
    if ((WTM % RPP != 0 && row >= WTM) || m >= d.M) continue;
    const float* src = cs + row * WTN + (col ^ (((row >> 2) & (WTN / 16 - 1)) << 4));
    float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    if (d.bias) {
      lo.x += b0.x; lo.y += b0.y; lo.z += b0.z; lo.w += b0.w; hi.x += b1.x; hi.y += b1.y; hi.z += b1.z; hi.w += b1.w;
    }
    if (d.act == S2S_ACT_RELU) {
      lo.x = fmaxf(lo.x, 0.f); lo.y = fmaxf(lo.y, 0.f); lo.z = fmaxf(lo.z, 0.f); lo.w = fmaxf(lo.w, 0.f);
      hi.x = fmaxf(hi.x, 0.f); hi.y = fmaxf(hi.y, 0.f); hi.z = fmaxf(hi.z, 0.f); hi.w = fmaxf(hi.w, 0.f);
    }
    if (d.res) {
      const float* r = (const float*)d.res + (int64_t)m * d.ldr + n;
      const float4 r0 = *reinterpret_cast<const float4*>(r), r1 = *reinterpret_cast<const float4*>(r + 4);
      lo.x += r0.x; lo.y += r0.y; lo.z += r0.z; lo.w += r0.w; hi.x += r1.x; hi.y += r1.y; hi.z += r1.z; hi.w += r1.w;
    }
    float* c = (float*)d.C + (int64_t)m * d.ldc + n;
    if (d.accumulate) {
      const float4 c0 = *reinterpret_cast<const float4*>(c), c1 = *reinterpret_cast<const float4*>(c + 4);
      lo.x += c0.x; lo.y += c0.y; lo.z += c0.z; lo.w += c0.w; hi.x += c1.x; hi.y += c1.y; hi.z += c1.z; hi.w += c1.w;
    }
    *reinterpret_cast<float4*>(c) = lo;
    *reinterpret_cast<float4*>(c + 4) = hi;
  }
}

// split-K partial sums of a wave's tile -> workspace slice zs (the reduction kernel applies the epilogue), N % 8 == 0
template <int WTM, int WTN>
__device__ __forceinline__ void epilogue_partials(const s2svc_gemm_desc& d, int zs, int m_base, int n_base, const float* cs) {
  const int lane = threadIdx.x & 63;
  constexpr int LPR = WTN / 8, RPP = 64 / LPR;
#pragma unroll 1
  for (int p = 0; p < (WTM + RPP - 1) / RPP; ++p) {
    const int row = p * RPP + lane / LPR, col = (lane % LPR) * 8;
    const int m = m_base + row, n = n_base + col;
    if ((WTM % RPP != 0 && row >= WTM) || m >= d.M || n >= d.N) continue;
    const float* src = cs + row * WTN + (col ^ (((row >> 2) & (WTN / 16 - 1)) << 4));
    float* w = d.ws + ((int64_t)zs * d.M + m) * d.N + n;
    *reinterpret_cast<float4*>(w) = *reinterpret_cast<const float4*>(src);
    *reinterpret_cast<float4*>(w + 4) = *reinterpret_cast<const float4*>(src + 4);
  }
}

// element-wise form of the common epilogue for the skinny (decode) kernels, which inline their epilogue once per accumulator
// element: bias, ReLU, [STAGED: dropout], residual, store in C's dtype -- the arithmetic of epilogue_store_f in that order
inline bool epilogue_lean_ok(const s2svc_gemm_desc& d) {
  return d.nb0 * d.nb1 == 1 && d.splitk <= 1 && d.alpha == 1.0f && !d.c_pre && !d.c_map && !d.accumulate && !d.emask &&
         (d.act == S2S_ACT_NONE || d.act == S2S_ACT_RELU);
}
template <bool STAGED>
__device__ __forceinline__ void epilogue_store_lean(const s2svc_gemm_desc& d, int m, int n, float v) {
  if (d.bias) v += d.bias[n];
  if (d.act == S2S_ACT_RELU) v = v > 0.f ? v : 0.f;
  if (STAGED && d.drop_p > 0.f) {
    const uint64_t seed = (d.seed_base ? *d.seed_base : 0ull) + d.seed_off;
    v *= dropout_scale(seed, (uint64_t)((int64_t)m * d.N + n), d.drop_p, 1.f / (1.f - d.drop_p));
  }
  const int64_t co = (int64_t)m * d.ldc + n;
  if (d.res) {
    const int64_t ro = (int64_t)m * d.ldr + n;
    v += d.c_dtype == S2S_F32 ? ((const float*)d.res)[ro] : bf2f(((const bf16_t*)d.res)[ro]);
  }
  if (d.c_dtype == S2S_F32) ((float*)d.C)[co] = v;
  else ((bf16_t*)d.C)[co] = f2bf(v);
}
