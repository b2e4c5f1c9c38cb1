// Autoregressive decode step kernels (one new decoder position per step, all utterances of the batch in lockstep).
//
// reference: models/vtn.py:344-389 (the generation loop), modules/transformer/decoder.py:239-273
// (forward_one_step), decoder_layer.py:85-132 (per-layer cache).  The reference re-projects K/V over the whole
// prefix every step and grows `ys` by concatenation; here K/V live in a static cache of capacity Lmax, the step
// index is a device-resident scalar (so one captured hipGraph replays for every step), and the stop test runs
// on the device.
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

// y[b, :] = x[b, :] * xscale + alpha * pe[pos, :]
template <typename T>
__global__ __launch_bounds__(256) void decode_posenc_kernel(int B, int D, const T* __restrict__ x, float xscale,
                                                            const float* __restrict__ alpha, const float* __restrict__ pe,
                                                            const int32_t* __restrict__ pos, T* __restrict__ y) {
  const int p = *pos;
  const float a = alpha ? *alpha : 1.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * D; i += gridDim.x * blockDim.x) {
    const int d = i % D;
    stf(y + i, ldf(x + i) * xscale + a * pe[(int64_t)p * D + d]);
  }
}

// vector of VEC consecutive elements -> floats
template <typename T, int VEC> struct VLoad;
template <> struct VLoad<bf16_t, 8> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&f)[8]) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
  }
};
template <> struct VLoad<float, 4> {
  static __device__ __forceinline__ void ld(const float* p, float (&f)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
};
template <typename T> struct VLoad<T, 1> {
  static __device__ __forceinline__ void ld(const T* p, float (&f)[1]) { f[0] = ldf(p); }
};

// One workgroup per (utterance, head).  Optional append of the new key/value row at *pos; scores: 4 lanes per
// key, 64 keys per pass, 16-byte row loads all in flight at once; softmax statistics by one wavefront; context:
// (d_k / VEC) column chunks x S key-splits of threads accumulate P.V partials, summed through LDS in fixed order.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void decode_attn_kernel(int H, int dk, const T* __restrict__ q, int64_t ldq, T* __restrict__ kc,
                                                          T* __restrict__ vc, int64_t ldt, int64_t cbs, const T* __restrict__ knew,
                                                          const T* __restrict__ vnew, int64_t ldn, const int32_t* __restrict__ pos,
                                                          const int32_t* __restrict__ klen, int Tk, float scale, T* __restrict__ ctx,
                                                          int64_t ldo, float* __restrict__ att, int64_t att_bs, int64_t att_hs,
                                                          int64_t att_ps) {
  extern __shared__ float sm[];   // Tk scores | dk query values | S x dk context partials
  float* sc = sm;
  float* sq = sm + Tk;
  float* part = sq + dk;
  __shared__ float red[2];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int p = *pos;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* kb = kc + (int64_t)b * cbs + h * dk;
  T* vb = vc + (int64_t)b * cbs + h * dk;
  int n;
  if (knew) {                      // self-attention: append this step's key/value, attend to positions 0..pos
    n = p + 1;
    if (n > Tk) n = Tk;            // capacity guard (the host never replays past the cache capacity)
    for (int d = threadIdx.x; d < dk; d += 256) {
      kb[(int64_t)p * ldt + d] = knew[(int64_t)b * ldn + h * dk + d];
      vb[(int64_t)p * ldt + d] = vnew[(int64_t)b * ldn + h * dk + d];
    }
  } else {
    n = klen ? klen[b] : Tk;
    if (n > Tk) n = Tk;
  }
  for (int d = threadIdx.x; d < dk; d += 256) sq[d] = ldf(q + (int64_t)b * ldq + h * dk + d);
  __syncthreads();                 // also orders the cache append before the reads below (same workgroup)
  const int nvec = dk / VEC;
  {
    const int c = threadIdx.x & 3;
    for (int j0 = 0; j0 < n; j0 += 64) {
      const int j = j0 + (threadIdx.x >> 2);
      float s = 0.f;
      if (j < n) {
        for (int v = c; v < nvec; v += 4) {
          float f[VEC];
          VLoad<T, VEC>::ld(kb + (int64_t)j * ldt + v * VEC, f);
#pragma unroll
          for (int e = 0; e < VEC; ++e) s += sq[v * VEC + e] * f[e];
        }
      }
      s = xor2_sum(xor1_sum(s));           // the 4 lanes of a key (DPP quad permutes, common.h)
      if (c == 0 && j < n) sc[j] = s * scale;
    }
  }
  __syncthreads();
  if (wave == 0) {
    float mx = -3.4028234663852886e38f;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) sum += expf(sc[j] - mx);
    sum = wave_sum(sum);
    if (lane == 0) { red[0] = mx; red[1] = 1.f / sum; }
  }
  __syncthreads();
  const float mx = red[0], inv = red[1];
  float* arow = att ? att + (int64_t)b * att_bs + (int64_t)h * att_hs + (int64_t)p * att_ps : nullptr;
  for (int j = threadIdx.x; j < Tk; j += 256) {
    const float pr = j < n ? expf(sc[j] - mx) * inv : 0.f;
    sc[j] = pr;
    if (arow) arow[j] = pr;
  }
  __syncthreads();
  const int S = 256 / nvec < 1 ? 1 : 256 / nvec;     // key splits (host sizes `part` for this)
  {
    const int c = threadIdx.x % nvec, sidx = threadIdx.x / nvec;
    if (sidx < S) {
      float acc[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
      for (int j = sidx; j < n; j += S) {
        float f[VEC];
        VLoad<T, VEC>::ld(vb + (int64_t)j * ldt + c * VEC, f);
        const float pj = sc[j];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += pj * f[e];
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) part[sidx * dk + c * VEC + e] = acc[e];
    }
  }
  __syncthreads();
  for (int d = threadIdx.x; d < dk; d += 256) {
    float acc = 0.f;
    for (int s2 = 0; s2 < S; ++s2) acc += part[s2 * dk + d];
    stf(ctx + (int64_t)b * ldo + h * dk + d, acc);
  }
}

// outs[b, pos*r + i, :] = feat[b, i*odim : (i+1)*odim] ; probs[b, pos*r + i] = sigmoid(logit[b, i]) ;
// prev[b, :] = last of the r frames ; with idx = pos+1 (the reference's 1-based step counter, vtn.py:345),
// stop_at[b] = idx the first time (any prob >= threshold || idx >= maxlen[b]) && idx >= minlen[b]  (vtn.py:378-381)
template <typename T>
__global__ __launch_bounds__(256) void decode_emit_kernel(int r, int odim, const T* __restrict__ feat, const T* __restrict__ logit,
                                                          float threshold, const int32_t* __restrict__ minlen,
                                                          const int32_t* __restrict__ maxlen, const int32_t* __restrict__ pos,
                                                          float* __restrict__ outs, int64_t outs_bs, float* __restrict__ probs,
                                                          int64_t probs_bs, T* __restrict__ prev, int32_t* __restrict__ stop_at) {
  const int b = blockIdx.x;
  const int p = *pos;
  for (int i = threadIdx.x; i < r * odim; i += 256) {
    const float v = ldf(feat + (int64_t)b * r * odim + i);
    outs[(int64_t)b * outs_bs + (int64_t)p * r * odim + i] = v;
    if (i >= (r - 1) * odim) stf(prev + (int64_t)b * odim + (i - (r - 1) * odim), v);
  }
  if (threadIdx.x == 0) {
    bool fire = false;
    for (int i = 0; i < r; ++i) {
      const float pr = 1.f / (1.f + expf(-ldf(logit + (int64_t)b * r + i)));
      probs[(int64_t)b * probs_bs + (int64_t)p * r + i] = pr;
      fire = fire || (pr >= threshold);
    }
    if ((fire || p + 1 >= maxlen[b]) && p + 1 >= minlen[b] && stop_at[b] == 0) stop_at[b] = p + 1;
  }
}

__global__ void decode_advance_kernel(int32_t* pos, uint64_t* seed_base, uint64_t seed_stride) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    *pos += 1;
    if (seed_base) *seed_base += seed_stride;
  }
}

}  // namespace

extern "C" int s2svc_decode_posenc(int dtype, int B, int D, const void* x, float xscale, const float* alpha, const float* pe,
                                   const int32_t* pos, void* y, void* stream) {
  S2S_REQUIRE(B >= 0 && D > 0 && pos && pe, "decode_posenc: bad arguments");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int nb = (B * D + 255) / 256;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(decode_posenc_kernel<float>, dim3(nb), dim3(256), 0, st, B, D, (const float*)x, xscale, alpha, pe, pos, (float*)y);
  else
    hipLaunchKernelGGL(decode_posenc_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, B, D, (const bf16_t*)x, xscale, alpha, pe, pos, (bf16_t*)y);
  S2S_CHECK_LAUNCH("decode_posenc_kernel");
  return 0;
}

extern "C" int s2svc_decode_attn(int dtype, int B, int H, int dk, const void* q, int64_t ldq, void* kcache, void* vcache,
                                 int64_t ldt, int64_t cbs, const void* knew, const void* vnew, int64_t ldn, const int32_t* pos,
                                 const int32_t* klen, int Tk, float scale, void* ctx, int64_t ldo, float* att, int64_t att_bs,
                                 int64_t att_hs, int64_t att_ps, void* stream) {
  S2S_REQUIRE(B >= 0 && H > 0 && dk > 0 && dk <= 256 && Tk > 0 && pos, "decode_attn: bad arguments");
  S2S_REQUIRE((knew == nullptr) == (vnew == nullptr), "decode_attn: knew and vnew go together");
  if (B == 0) return 0;
  const int vec = dtype == S2S_F32 ? 4 : 8;
  const size_t esz = dtype == S2S_F32 ? 4 : 2;
  // 16-byte row loads need aligned rows; otherwise the element-wise instantiation runs
  const bool vector_ok = dk % vec == 0 && ldt % vec == 0 && cbs % vec == 0 && ((uintptr_t)kcache) % 16 == 0 &&
                         ((uintptr_t)vcache) % 16 == 0;
  const int nvec = vector_ok ? dk / vec : dk;
  const int S = 256 / nvec < 1 ? 1 : 256 / nvec;
  const size_t shm = ((size_t)Tk + dk + (size_t)S * dk) * sizeof(float);
  S2S_REQUIRE(shm <= 60 * 1024, "decode_attn: key capacity too large for the LDS score buffer");
  (void)esz;
  hipStream_t st = (hipStream_t)stream;
#define S2S_DECODE_ATTN(T, VEC)                                                                                              \
  hipLaunchKernelGGL((decode_attn_kernel<T, VEC>), dim3(B * H), dim3(256), shm, st, H, dk, (const T*)q, ldq, (T*)kcache,       \
                     (T*)vcache, ldt, cbs, (const T*)knew, (const T*)vnew, ldn, pos, klen, Tk, scale, (T*)ctx, ldo, att, att_bs, \
                     att_hs, att_ps)
  if (dtype == S2S_F32) {
    if (vector_ok) S2S_DECODE_ATTN(float, 4); else S2S_DECODE_ATTN(float, 1);
  } else {
    if (vector_ok) S2S_DECODE_ATTN(bf16_t, 8); else S2S_DECODE_ATTN(bf16_t, 1);
  }
#undef S2S_DECODE_ATTN
  S2S_CHECK_LAUNCH("decode_attn_kernel");
  return 0;
}

extern "C" int s2svc_decode_emit(int dtype, int B, int r, int odim, const void* feat, const void* logit, float threshold,
                                 const int32_t* minlen, const int32_t* maxlen, const int32_t* pos, float* outs, int64_t outs_bs, float* probs, int64_t probs_bs, void* prev,
                                 int32_t* stop_at, void* stream) {
  S2S_REQUIRE(B >= 0 && r > 0 && odim > 0 && pos && outs && probs && prev && stop_at, "decode_emit: bad arguments");
  S2S_REQUIRE(minlen && maxlen, "decode_emit: per-utterance minlen/maxlen required");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(decode_emit_kernel<float>, dim3(B), dim3(256), 0, st, r, odim, (const float*)feat, (const float*)logit, threshold,
                       minlen, maxlen, pos, outs, outs_bs, probs, probs_bs, (float*)prev, stop_at);
  else
    hipLaunchKernelGGL(decode_emit_kernel<bf16_t>, dim3(B), dim3(256), 0, st, r, odim, (const bf16_t*)feat, (const bf16_t*)logit,
                       threshold, minlen, maxlen, pos, outs, outs_bs, probs, probs_bs, (bf16_t*)prev, stop_at);
  S2S_CHECK_LAUNCH("decode_emit_kernel");
  return 0;
}

extern "C" int s2svc_decode_advance(int32_t* pos, uint64_t* seed_base, uint64_t seed_stride, void* stream) {
  S2S_REQUIRE(pos, "decode_advance: pos required");
  hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pos, seed_base, seed_stride);
  S2S_CHECK_LAUNCH("decode_advance_kernel");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm -> Linear for a decode step:  C[M <= 64, N] = act(LN(S)[M, K] . W[N, K]^T + bias) (+ res), optionally also
// Y = LN(S) written out (by workgroup 0) for the residual path of a post-LN layer.  Replaces the LayerNorm launch in front
// of every projection of decoder_layer.py:85-132 for one position: the 16..64 rows of a step are normalised by every
// workgroup on its own (M * K <= 64 x 1536 elements, far cheaper than a launch), then stream through the same MFMA
// fragment scheme as gemm_skinny_kernel (weights straight from global memory, 4 waves split K, LDS reduction).
// With gamma == NULL the kernel is a plain skinny linear with the dropout stage of the prenet in its epilogue.
// ------------------------------------------------------------------------------------------------
#include "gemm_common.h"

namespace {

template <typename T> struct DFrag;
template <> struct DFrag<bf16_t> {
  static constexpr int VEC = 8, KSTEP = 32;
  typedef bf16x8_t type;
  static __device__ __forceinline__ type zero() { return (type){0, 0, 0, 0, 0, 0, 0, 0}; }
  static __device__ __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void unpack(type v, float (&f)[8]) {
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
    const u16x8 u = __builtin_bit_cast(u16x8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = bf2f(u[e]);
  }
  static __device__ __forceinline__ type pack(const float (&f)[8]) {
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
    u16x8 u;
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] = f2bf(f[e]);
    return __builtin_bit_cast(type, u);
  }
};
template <> struct DFrag<float> {
  static constexpr int VEC = 4, KSTEP = 16;
  typedef f32x4_t type;
  static __device__ __forceinline__ type zero() { return (type){0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ f32x4_t mma(type a, type b, f32x4_t c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], c, 0, 0, 0);
    return c;
  }
  static __device__ __forceinline__ void unpack(type v, float (&f)[4]) { f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3]; }
  static __device__ __forceinline__ type pack(const float (&f)[4]) { return (type){f[0], f[1], f[2], f[3]}; }
};

// NS = k-steps per wave held in registers (the whole K range of a wave is loaded up front: one exposed memory latency per
// launch; the LayerNorm statistics come from those registers -- two-pass, partial sums through LDS across the four waves)
template <typename T, int MT, int NS, bool LEAN = false>       // LEAN: see gemm_skinny_kernel
__global__ __launch_bounds__(256) void ln_linear_skinny_kernel(const s2svc_gemm_desc d, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, T* __restrict__ y_out,
                                                               int64_t ldy) {
  typedef DFrag<T> F;
  typedef typename F::type frag_t;
  __shared__ float red[3][MT][256];
  __shared__ float st_part[4][MT * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const T* A = (const T*)d.A.ptr;
  const T* B = (const T*)d.B.ptr;
  const int ksteps = (d.K + F::KSTEP - 1) / F::KSTEP;
  const int per = (ksteps + 3) / 4;                      // <= NS (checked by the launcher)
  const int ks0 = wave * per;
  const bool brow = (n0 + lr) < d.N;
  const T* bp = B + (int64_t)(n0 + lr) * d.B.ld + lg * F::VEC;
  frag_t a[NS][MT], b[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int k = (ks0 + u) * F::KSTEP + lg * F::VEC;
    const bool kin = u < per && (ks0 + u) < ksteps && k < d.K;
    b[u] = (kin && brow) ? *reinterpret_cast<const frag_t*>(bp + (int64_t)(ks0 + u) * F::KSTEP) : F::zero();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = i * 16 + lr;
      a[u][i] = (kin && row < d.M) ? *reinterpret_cast<const frag_t*>(A + (int64_t)row * d.A.ld + k) : F::zero();
    }
  }
  if (gamma) {
    float mean[MT], rstd[MT];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
          const int k = (ks0 + u) * F::KSTEP + lg * F::VEC;
          if (u < per && (ks0 + u) < ksteps && k < d.K) {
            float f[F::VEC];
            F::unpack(a[u][i], f);
#pragma unroll
            for (int e = 0; e < F::VEC; ++e) { const float t = pass ? f[e] - mean[i] : f[e]; s += pass ? t * t : t; }
          }
        }
        s = xor32_sum(xor16_sum(s));       // the four 16-lane rows (v_permlane16/32_swap, common.h)
        if (lg == 0) st_part[wave][i * 16 + lr] = s;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int row = i * 16 + lr;
        const float tot = ((st_part[0][row] + st_part[1][row]) + st_part[2][row]) + st_part[3][row];
        if (pass == 0) mean[i] = tot / (float)d.K;
        else rstd[i] = rsqrtf(tot / (float)d.K + eps);
      }
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int k = (ks0 + u) * F::KSTEP + lg * F::VEC;
      if (u < per && (ks0 + u) < ksteps && k < d.K) {
        float g[F::VEC], bt[F::VEC];
#pragma unroll
        for (int e = 0; e < F::VEC; ++e) { g[e] = gamma[k + e]; bt[e] = beta[k + e]; }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int row = i * 16 + lr;
          if (row < d.M) {
            float f[F::VEC];
            F::unpack(a[u][i], f);
#pragma unroll
            for (int e = 0; e < F::VEC; ++e) f[e] = (f[e] - mean[i]) * rstd[i] * g[e] + bt[e];
            a[u][i] = F::pack(f);
            if (y_out && blockIdx.x == 0) *reinterpret_cast<frag_t*>(y_out + (int64_t)row * ldy + k) = a[u][i];
          }
        }
      }
    }
  }
  f32x4_t acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NS; ++u)
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = F::mma(a[u][i], b[u], acc[i]);
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave - 1][i][r * 64 + lane] = acc[i][r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = ((acc[i][r] + red[0][i][r * 64 + lane]) + red[1][i][r * 64 + lane]) + red[2][i][r * 64 + lane];
        const int m = i * 16 + lg * 4 + r, n = n0 + lr;
        if (m < d.M && n < d.N) {
          if (LEAN) epilogue_store_lean<true>(d, m, n, v);
          else epilogue_store_f<true>(d, 0, 0, m, n, v);
        }
      }
  }
}

template <typename T, int NS>
void launch_ln_linear_ns(const s2svc_gemm_desc& d, const float* gamma, const float* beta, float eps, void* y_out, int64_t ldy, hipStream_t st) {
  dim3 grid((d.N + 15) / 16), block(256);
  const int mt = (d.M + 15) / 16;
  static const bool lean_on = true;
  if (lean_on && mt <= 2 && epilogue_lean_ok(d)) {
    if (mt == 1) hipLaunchKernelGGL((ln_linear_skinny_kernel<T, 1, NS, true>), grid, block, 0, st, d, gamma, beta, eps, (T*)y_out, ldy);
    else hipLaunchKernelGGL((ln_linear_skinny_kernel<T, 2, NS, true>), grid, block, 0, st, d, gamma, beta, eps, (T*)y_out, ldy);
    return;
  }
  if (mt == 1) hipLaunchKernelGGL((ln_linear_skinny_kernel<T, 1, NS>), grid, block, 0, st, d, gamma, beta, eps, (T*)y_out, ldy);
  else if (mt == 2) hipLaunchKernelGGL((ln_linear_skinny_kernel<T, 2, NS>), grid, block, 0, st, d, gamma, beta, eps, (T*)y_out, ldy);
  else if (mt == 3) hipLaunchKernelGGL((ln_linear_skinny_kernel<T, 3, NS>), grid, block, 0, st, d, gamma, beta, eps, (T*)y_out, ldy);
  else hipLaunchKernelGGL((ln_linear_skinny_kernel<T, 4, NS>), grid, block, 0, st, d, gamma, beta, eps, (T*)y_out, ldy);
}

// k-steps per wave: K = 384 -> 3 (bf16) / 6 (fp32); up to K = 512 (bf16) / 256.. handled by the larger instantiations
template <typename T>
bool launch_ln_linear(const s2svc_gemm_desc& d, const float* gamma, const float* beta, float eps, void* y_out, int64_t ldy, hipStream_t st) {
  const int kstep = DFrag<T>::KSTEP;
  const int per = ((d.K + kstep - 1) / kstep + 3) / 4;
  if (per <= 3) launch_ln_linear_ns<T, 3>(d, gamma, beta, eps, y_out, ldy, st);
  else if (per <= 6) launch_ln_linear_ns<T, 6>(d, gamma, beta, eps, y_out, ldy, st);
  else if (per <= 12 && d.M <= 32) launch_ln_linear_ns<T, 12>(d, gamma, beta, eps, y_out, ldy, st);
  else return false;
  return true;
}

}  // namespace

// 1 if s2svc_decode_ln_linear takes an (M x K) input of this dtype (the register-resident form: M <= 64, whole 16-byte vectors
// along K, at most 6 k steps per wave -- 12 up to M = 32); callers fall back to s2svc_layernorm_fwd + s2svc_gemm otherwise
extern "C" int s2svc_decode_ln_linear_supported(int dtype, int M, int K) {
  if ((dtype != S2S_F32 && dtype != S2S_BF16) || M <= 0 || M > 64 || K <= 0) return 0;
  const int vec = dtype == S2S_F32 ? 4 : 8;
  if (K % vec != 0) return 0;
  const int kstep = dtype == S2S_F32 ? DFrag<float>::KSTEP : DFrag<bf16_t>::KSTEP;
  const int per = ((K + kstep - 1) / kstep + 3) / 4;
  return per <= 6 || (per <= 12 && M <= 32);
}

extern "C" int s2svc_decode_ln_linear(const s2svc_gemm_desc* desc, const float* gamma, const float* beta, float eps, void* y_out,
                                      int64_t ldy, void* stream) {
  S2S_REQUIRE(desc != nullptr, "decode_ln_linear: null desc");
  const s2svc_gemm_desc& d = *desc;
  S2S_REQUIRE(d.M > 0 && d.M <= 64 && d.N > 0 && d.K > 0 && d.nb0 * d.nb1 <= 1 && d.splitk <= 1 && !d.a_rowsum && !d.c_map,
              "decode_ln_linear: M <= 64, unbatched, unsplit");
  S2S_REQUIRE(d.A.mode == S2SVC_OP_DENSE && d.B.mode == S2SVC_OP_DENSE && d.A.layout == S2SVC_LAYOUT_KC && d.B.layout == S2SVC_LAYOUT_KC,
              "decode_ln_linear: dense K-contiguous operands");
  S2S_REQUIRE((gamma == nullptr) == (beta == nullptr), "decode_ln_linear: gamma and beta come together");
  const int vec = d.dtype == S2S_F32 ? 4 : 8;
  S2S_REQUIRE(d.K % vec == 0 && d.A.ld % vec == 0 && d.B.ld % vec == 0 && (!y_out || ldy % vec == 0), "decode_ln_linear: K / strides must be whole 16-byte vectors");
  S2S_REQUIRE(((uintptr_t)d.A.ptr) % 16 == 0 && ((uintptr_t)d.B.ptr) % 16 == 0 && ((uintptr_t)y_out) % 16 == 0, "decode_ln_linear: 16-byte aligned operands");
  S2S_REQUIRE(d.drop_p < 1.f && (!(d.drop_p > 0.f || d.emask) || d.ldc == d.N), "decode_ln_linear: the dropout stage needs a contiguous C");
  hipStream_t st = (hipStream_t)stream;
  const bool ok = d.dtype == S2S_F32 ? launch_ln_linear<float>(d, gamma, beta, eps, y_out, ldy, st)
                                     : launch_ln_linear<bf16_t>(d, gamma, beta, eps, y_out, ldy, st);
  S2S_REQUIRE(ok, "decode_ln_linear: K too large for the register-resident form (use s2svc_layernorm_fwd + s2svc_gemm)");
  S2S_CHECK_LAUNCH("ln_linear_skinny_kernel");
  return 0;
}
