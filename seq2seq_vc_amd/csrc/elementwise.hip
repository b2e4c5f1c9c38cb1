// HBM-bound elementwise kernels of the hot path: dropout, activation backward, positional encodings,
// residual adds, head-bias adds of the relative-position attention, GLU, casts and small permutes.
// Grid-stride loops, 256-thread blocks, capped at 2048 blocks (8 per CU).
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}
#define EW_LOOP(i, total) \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (int64_t)gridDim.x * blockDim.x)

// y = dropout(act(x))           reference: torch.nn.Dropout / F.dropout call sites
// (modules/pre_postnets.py:63-66 -- the always-on prenet dropout)
template <typename T>
__global__ void act_dropout_fwd_kernel(int64_t n, const T* __restrict__ x, int act, float p, const uint64_t* seed_base, uint64_t seed_off, T* __restrict__ y) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  EW_LOOP(i, n) {
    float v = act_apply(ldf(x + i), act);
    if (p > 0.f) v *= dropout_scale(seed, (uint64_t)i, p, inv_keep);
    stf(y + i, v);
  }
}

// dx = dz * mask * act'(.) where z = dropout(act(x)) is the saved forward OUTPUT (relu / tanh /
// sigmoid derivatives are functions of the output), or the saved INPUT for swish / gelu.
template <typename T>
__global__ void act_dropout_bwd_kernel(int64_t n, const T* __restrict__ dz, const T* __restrict__ saved, int act, float p,
                                       const uint64_t* seed_base, uint64_t seed_off, T* __restrict__ dx) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  EW_LOOP(i, n) {
    const float m = p > 0.f ? dropout_scale(seed, (uint64_t)i, p, inv_keep) : 1.f;
    const float s = ldf(saved + i);
    float d;
    if (act == S2S_ACT_RELU) {
      d = s > 0.f ? 1.f : 0.f;
    } else if (act == S2S_ACT_TANH) {
      const float yv = m > 0.f ? s / m : 0.f;
      d = 1.f - yv * yv;
    } else if (act == S2S_ACT_SIGMOID) {
      const float yv = m > 0.f ? s / m : 0.f;
      d = yv * (1.f - yv);
    } else if (act == S2S_ACT_SWISH) {  // saved = pre-activation x
      const float sg = 1.f / (1.f + expf(-s));
      d = sg * (1.f + s * (1.f - sg));
    } else if (act == S2S_ACT_GELU) {   // saved = pre-activation x (erf form)
      const float cdf = 0.5f * (1.f + erff(s * 0.70710678118654752f));
      d = cdf + s * 0.3989422804014327f * expf(-0.5f * s * s);
    } else {
      d = 1.f;
    }
    stf(dx + i, ldf(dz + i) * m * d);
  }
}

// bf16, n % 8 == 0, 16-byte aligned: 8 consecutive elements per thread and iteration (one 16-byte load per tensor, one
// 16-byte store, two RNG draws) -- same values as the scalar kernels above
__device__ __forceinline__ float act_grad_from_saved(float s, float m, int act) {
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
__global__ void act_dropout_fwd_vec_kernel(int64_t n8, const bf16_t* __restrict__ x, int act, float p, const uint64_t* seed_base,
                                           uint64_t seed_off, bf16_t* __restrict__ y) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  EW_LOOP(i, n8) {
    float v[8], m[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(x + i * 8), v);
    if (p > 0.f) dropout_scale8(seed, (uint64_t)(i * 8), p, inv_keep, m);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[e] = act_apply(v[e], act);
      if (p > 0.f) v[e] *= m[e];
    }
    *reinterpret_cast<uint4*>(y + i * 8) = pack_bf16x8(v);
  }
}
__global__ void act_dropout_bwd_vec_kernel(int64_t n8, const bf16_t* __restrict__ dz, const bf16_t* __restrict__ saved, int act, float p,
                                           const uint64_t* seed_base, uint64_t seed_off, bf16_t* __restrict__ dx) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  EW_LOOP(i, n8) {
    float g[8], s[8], m[8];
    unpack_bf16x8(*reinterpret_cast<const uint4*>(dz + i * 8), g);
    unpack_bf16x8(*reinterpret_cast<const uint4*>(saved + i * 8), s);
    if (p > 0.f) dropout_scale8(seed, (uint64_t)(i * 8), p, inv_keep, m);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float mm = p > 0.f ? m[e] : 1.f;
      g[e] = g[e] * mm * act_grad_from_saved(s[e], mm, act);
    }
    *reinterpret_cast<uint4*>(dx + i * 8) = pack_bf16x8(g);
  }
}

// y = dropout(x*xscale + alpha*pe[t, :])    x: (B, T, D), pe: fp32 (>=T, D) table rows t0..t0+T-1
// reference: layers/positional_encoding.py:57-70 (x*sqrt(d)+pe), :94-106 (x + alpha*pe),
//            :226-235 / :293-309 (relative variants: x*sqrt(d) only, pe handed to the attention)
template <typename T>
__global__ void posenc_fwd_kernel(int64_t n, int Tlen, int D, const T* __restrict__ x, float xscale,
                                  const float* __restrict__ alpha, const float* __restrict__ pe, float p, const uint64_t* seed_base, uint64_t seed_off,
                                  T* __restrict__ y) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const float a = alpha ? *alpha : 1.f;
  EW_LOOP(i, n) {
    const int d = (int)(i % D);
    const int t = (int)((i / D) % Tlen);
    float v = ldf(x + i) * xscale;
    if (pe) v += a * pe[(int64_t)t * D + d];
    if (p > 0.f) v *= dropout_scale(seed, (uint64_t)i, p, inv_keep);
    stf(y + i, v);
  }
}
// dx = dy*mask*xscale ; g_alpha_elem = dy*mask*pe (written to `dpe_prod` for a later column/total reduce)
template <typename T>
__global__ void posenc_bwd_kernel(int64_t n, int Tlen, int D, const T* __restrict__ dy, float xscale,
                                  const float* __restrict__ pe, float p, const uint64_t* seed_base, uint64_t seed_off, T* __restrict__ dx,
                                  float* __restrict__ dalpha_partial) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float acc = 0.f;
  EW_LOOP(i, n) {
    const int d = (int)(i % D);
    const int t = (int)((i / D) % Tlen);
    const float m = p > 0.f ? dropout_scale(seed, (uint64_t)i, p, inv_keep) : 1.f;
    const float g = ldf(dy + i) * m;
    stf(dx + i, g * xscale);
    if (dalpha_partial) acc += g * pe[(int64_t)t * D + d];
  }
  if (dalpha_partial) {
    __shared__ float sh[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dalpha_partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
  }
}
__global__ void sum_partials_kernel(int n, const float* __restrict__ part, float* __restrict__ out, int accumulate) {
  // single wavefront, fixed order => deterministic
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) acc += part[i];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) *out = (accumulate ? *out : 0.f) + acc;
}

// out = a*x + b*y
template <typename T>
__global__ void axpby_kernel(int64_t n, float a, const T* __restrict__ x, float b, const T* __restrict__ y, T* __restrict__ out) {
  EW_LOOP(i, n) {
    float v = a * ldf(x + i);
    if (y) v += b * ldf(y + i);
    stf(out + i, v);
  }
}

// out = x0 + x1 (+ x2 (+ x3)), summed in that order in fp32: the gradients that meet at a tensor with several consumers
// (ops.functional.fan_out) in ONE launch, instead of the k - 1 element-wise adds autograd's accumulation issues
template <typename T>
__global__ void add_n_kernel(int64_t n, const T* __restrict__ x0, const T* __restrict__ x1, const T* __restrict__ x2,
                             const T* __restrict__ x3, T* __restrict__ out) {
  EW_LOOP(i, n) {
    float v = ldf(x0 + i) + ldf(x1 + i);
    if (x2) v += ldf(x2 + i);
    if (x3) v += ldf(x3 + i);
    stf(out + i, v);
  }
}
// the same on 8 bf16 / 4 fp32 values per lane (n a multiple of 8, 16-byte aligned tensors)
__global__ void add_n_vec_bf16_kernel(int64_t n8, const uint4* __restrict__ x0, const uint4* __restrict__ x1, const uint4* __restrict__ x2,
                                      const uint4* __restrict__ x3, uint4* __restrict__ out) {
  EW_LOOP(i, n8) {
    float a[8], b[8];
    unpack_bf16x8(x0[i], a);
    unpack_bf16x8(x1[i], b);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] += b[e];
    if (x2) {
      unpack_bf16x8(x2[i], b);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += b[e];
    }
    if (x3) {
      unpack_bf16x8(x3[i], b);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += b[e];
    }
    out[i] = pack_bf16x8(a);
  }
}

// qu = q + u[h,:], qv = q + v[h,:]     q: (rows, D = H*dk), u/v: fp32 (D)
// reference: modules/transformer/attention.py:283-286
template <typename T>
__global__ void add_head_bias_kernel(int64_t n, int D, const T* __restrict__ q, const float* __restrict__ u,
                                     const float* __restrict__ v, T* __restrict__ qu, T* __restrict__ qv) {
  EW_LOOP(i, n) {
    const int d = (int)(i % D);
    const float x = ldf(q + i);
    stf(qu + i, x + u[d]);
    stf(qv + i, x + v[d]);
  }
}

// the same with q a column block of a packed projection (row stride ldq): qu / qv dense (rows, D)
template <typename T>
__global__ void add_head_bias_ld_kernel(int64_t n, int D, const T* __restrict__ q, int64_t ldq, const float* __restrict__ u,
                                        const float* __restrict__ v, T* __restrict__ qu, T* __restrict__ qv) {
  EW_LOOP(i, n) {
    const int64_t r = i / D;
    const int d = (int)(i - r * D);
    const float x = ldf(q + r * ldq + d);
    stf(qu + i, x + u[d]);
    stf(qv + i, x + v[d]);
  }
}

// out[r, d] = a[r, d] + b[r, d] over row-strided (rows, D) views (a column block of a packed gradient as destination)
template <typename T>
__global__ void add_rows_kernel(int64_t n, int D, const T* __restrict__ a, int64_t lda, const T* __restrict__ b, int64_t ldb,
                                T* __restrict__ out, int64_t ldo) {
  EW_LOOP(i, n) {
    const int64_t r = i / D;
    const int d = (int)(i - r * D);
    stf(out + r * ldo + d, ldf(a + r * lda + d) + ldf(b + r * ldb + d));
  }
}

// GLU over the channel halves of channel-last rows: y[r,c] = x[r,c] * sigmoid(x[r,C+c])
// reference: modules/conformer/convolution.py:68 (nn.functional.glu(x, dim=1) on (B,2C,T))
template <typename T>
__global__ void glu_fwd_kernel(int64_t rows, int C, const T* __restrict__ x, T* __restrict__ y) {
  const int64_t n = rows * C;
  EW_LOOP(i, n) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    const float a = ldf(x + r * 2 * C + c), g = ldf(x + r * 2 * C + C + c);
    stf(y + i, a / (1.f + expf(-g)));
  }
}
template <typename T>
__global__ void glu_bwd_kernel(int64_t rows, int C, const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx) {
  const int64_t n = rows * C;
  EW_LOOP(i, n) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    const float a = ldf(x + r * 2 * C + c), g = ldf(x + r * 2 * C + C + c);
    const float sg = 1.f / (1.f + expf(-g));
    const float d = ldf(dy + i);
    stf(dx + r * 2 * C + c, d * sg);
    stf(dx + r * 2 * C + C + c, d * a * sg * (1.f - sg));
  }
}

template <typename TI, typename TO>
__global__ void cast_kernel(int64_t n, const TI* __restrict__ x, TO* __restrict__ y) {
  EW_LOOP(i, n) stf(y + i, ldf(x + i));
}

// out[a][c][b] = in[a][b][c]   (optionally flipping b: out[a][c][b] = in[a][B-1-b][c] is NOT this;
// flip applies to the index that becomes innermost-but-one, see conv weight packing in ops/conv.py)
template <typename TI, typename TO>
__global__ void permute021_kernel(int A, int Bd, int Cd, const TI* __restrict__ in, TO* __restrict__ out) {
  const int64_t n = (int64_t)A * Bd * Cd;
  EW_LOOP(i, n) {  // i indexes OUT (a, c, b)
    const int b = (int)(i % Bd);
    const int64_t t = i / Bd;
    const int c = (int)(t % Cd);
    const int a = (int)(t / Cd);
    stf(out + i, ldf(in + ((int64_t)a * Bd + b) * Cd + c));
  }
}
// generic 3-d strided gather: out[i0][i1][i2] (contiguous) = in[i0*s0 + i1*s1 + i2*s2]
template <typename TI, typename TO>
__global__ void gather3_kernel(int n0, int n1, int n2, int64_t s0, int64_t s1, int64_t s2, int64_t off,
                               const TI* __restrict__ in, TO* __restrict__ out) {
  const int64_t n = (int64_t)n0 * n1 * n2;
  EW_LOOP(i, n) {
    const int i2 = (int)(i % n2);
    const int64_t t = i / n2;
    const int i1 = (int)(t % n1);
    const int i0 = (int)(t / n1);
    stf(out + i, ldf(in + off + i0 * s0 + i1 * s1 + i2 * s2));
  }
}

// several gathers of fp32 sources in ONE launch (the permuted weight copies of every convolution, refreshed after the
// optimiser step): blockIdx.y = job; jobs travel by value in the kernel arguments
#define S2S_GATHER_MAX 24
struct gather_jobs {
  s2svc_gather3_job j[S2S_GATHER_MAX];
  int32_t n;
};
static_assert(sizeof(gather_jobs) <= 4096, "kernel arguments are limited to 4 KB");

__global__ void gather3_grouped_kernel(const gather_jobs g) {
  const s2svc_gather3_job& jb = g.j[blockIdx.y];
  const int64_t n = (int64_t)jb.n0 * jb.n1 * jb.n2;
  const float* in = (const float*)jb.in;
  // Column gathers (the innermost output index has the LARGEST source stride: the data-gradient layout (C_in, k, C_out) of a
  // Conv1d weight (C_out, C_in, k), s2 = C_in * k) through 64 x 64 LDS tiles: read with the lanes along (i0, i1) -- a run of
  // source addresses -- and written with the lanes along i2.  The element-wise loop below reads 4 bytes per 18 KB-apart
  // address there (the two 1536 x 1536 x 3 aligner weights of AAS-VC: most of a 135 us launch).
  const int64_t a0 = jb.s0 < 0 ? -jb.s0 : jb.s0, a1 = jb.s1 < 0 ? -jb.s1 : jb.s1;
  if (jb.s2 >= 64 && jb.s2 > a0 && jb.s2 > a1 && jb.n2 >= 32) {           // (uniform per job)
    __shared__ float tile[64][65];
    const int R = jb.n0 * jb.n1;
    const int tc = (jb.n2 + 63) / 64, nt = ((R + 63) / 64) * tc;
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
      const int r0 = (t / tc) * 64, c0 = (t - (t / tc) * tc) * 64;
      const int r = r0 + x;
      int64_t a = 0;
      if (r < R) {
        const int i0 = r / jb.n1, i1 = r - i0 * jb.n1;
        a = jb.off + i0 * jb.s0 + i1 * jb.s1;
      }
#pragma unroll 4
      for (int cc = y; cc < 64; cc += 4) tile[cc][x] = (r < R && c0 + cc < jb.n2) ? in[a + (int64_t)(c0 + cc) * jb.s2] : 0.f;
      __syncthreads();
#pragma unroll 4
      for (int rr = y; rr < 64; rr += 4) {
        const int ro = r0 + rr, c = c0 + x;
        if (ro < R && c < jb.n2) {
          const float v = tile[x][rr];
          const int64_t o = (int64_t)ro * jb.n2 + c;
          if (jb.out_dtype == S2S_F32) ((float*)jb.out)[o] = v;
          else ((bf16_t*)jb.out)[o] = f2bf(v);
        }
      }
      __syncthreads();
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int i2 = (int)(i % jb.n2);
    const int64_t t = i / jb.n2;
    const int i1 = (int)(t % jb.n1);
    const int i0 = (int)(t / jb.n1);
    const float v = in[jb.off + i0 * jb.s0 + i1 * jb.s1 + i2 * jb.s2];
    if (jb.out_dtype == S2S_F32) ((float*)jb.out)[i] = v;
    else ((bf16_t*)jb.out)[i] = f2bf(v);
  }
}

template <typename T>
int launch_typed1(int dtype);

}  // namespace

#define DISPATCH(dtype, KERNEL, total, ...)                                                                  \
  do {                                                                                                       \
    if ((total) == 0) return 0;                                                                              \
    hipStream_t st_ = (hipStream_t)stream;                                                                   \
    if ((dtype) == S2S_F32) hipLaunchKernelGGL(KERNEL<float>, dim3(ew_blocks(total)), dim3(256), 0, st_, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<bf16_t>, dim3(ew_blocks(total)), dim3(256), 0, st_, __VA_ARGS__);           \
    S2S_CHECK_LAUNCH(#KERNEL);                                                                               \
  } while (0)

#define P(T_, p) ((T_*)(p))

extern "C" int s2svc_act_dropout_fwd(int dtype, int64_t n, const void* x, int act, float p, const uint64_t* seed_base, uint64_t seed_off, void* y,
                                     void* stream) {
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(act_dropout_fwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, n, (const float*)x, act, p, seed_base, seed_off, (float*)y);
  else if (n % 8 == 0 && ((uintptr_t)x) % 16 == 0 && ((uintptr_t)y) % 16 == 0)
    hipLaunchKernelGGL(act_dropout_fwd_vec_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, st, n / 8, (const bf16_t*)x, act, p, seed_base, seed_off, (bf16_t*)y);
  else
    hipLaunchKernelGGL(act_dropout_fwd_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, n, (const bf16_t*)x, act, p, seed_base, seed_off, (bf16_t*)y);
  S2S_CHECK_LAUNCH("act_dropout_fwd_kernel");
  return 0;
}

extern "C" int s2svc_act_dropout_bwd(int dtype, int64_t n, const void* dz, const void* saved, int act, float p,
                                     const uint64_t* seed_base, uint64_t seed_off, void* dx, void* stream) {
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(act_dropout_bwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, n, (const float*)dz, (const float*)saved, act, p, seed_base, seed_off, (float*)dx);
  else if (n % 8 == 0 && ((uintptr_t)dz) % 16 == 0 && ((uintptr_t)saved) % 16 == 0 && ((uintptr_t)dx) % 16 == 0)
    hipLaunchKernelGGL(act_dropout_bwd_vec_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, st, n / 8, (const bf16_t*)dz, (const bf16_t*)saved, act, p, seed_base, seed_off, (bf16_t*)dx);
  else
    hipLaunchKernelGGL(act_dropout_bwd_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, n, (const bf16_t*)dz, (const bf16_t*)saved, act, p, seed_base, seed_off, (bf16_t*)dx);
  S2S_CHECK_LAUNCH("act_dropout_bwd_kernel");
  return 0;
}

extern "C" int s2svc_posenc_fwd(int dtype, int64_t B, int T, int D, const void* x, float xscale, const float* alpha,
                                const float* pe, float p, const uint64_t* seed_base, uint64_t seed_off, void* y, void* stream) {
  const int64_t n = B * T * D;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(posenc_fwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, n, T, D, (const float*)x, xscale, alpha, pe, p, seed_base, seed_off, (float*)y);
  else
    hipLaunchKernelGGL(posenc_fwd_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, n, T, D, (const bf16_t*)x, xscale, alpha, pe, p, seed_base, seed_off, (bf16_t*)y);
  S2S_CHECK_LAUNCH("posenc_fwd_kernel");
  return 0;
}

// dalpha (nullable) gets sum(dy*mask*pe); `partials` must hold >= 2048 floats when dalpha != NULL
extern "C" int s2svc_posenc_bwd(int dtype, int64_t B, int T, int D, const void* dy, float xscale, const float* pe,
                                float p, const uint64_t* seed_base, uint64_t seed_off, void* dx, float* dalpha, float* partials, void* stream) {
  const int64_t n = B * T * D;
  if (n == 0) return 0;
  S2S_REQUIRE(!dalpha || (partials && pe), "posenc_bwd: dalpha needs partials and pe");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = ew_blocks(n);
  float* part = dalpha ? partials : nullptr;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(posenc_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, n, T, D, (const float*)dy, xscale, pe, p, seed_base, seed_off, (float*)dx, part);
  else
    hipLaunchKernelGGL(posenc_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, n, T, D, (const bf16_t*)dy, xscale, pe, p, seed_base, seed_off, (bf16_t*)dx, part);
  S2S_CHECK_LAUNCH("posenc_bwd_kernel");
  if (dalpha) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, blocks, partials, dalpha, 0);
    S2S_CHECK_LAUNCH("sum_partials_kernel");
  }
  return 0;
}

extern "C" int s2svc_axpby(int dtype, int64_t n, float a, const void* x, float b, const void* y, void* out, void* stream) {
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(axpby_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, n, a, (const float*)x, b, (const float*)y, (float*)out);
  else
    hipLaunchKernelGGL(axpby_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, n, a, (const bf16_t*)x, b, (const bf16_t*)y, (bf16_t*)out);
  S2S_CHECK_LAUNCH("axpby_kernel");
  return 0;
}

extern "C" int s2svc_add_n(int dtype, int64_t n, int k, const void* x0, const void* x1, const void* x2, const void* x3, void* out,
                           void* stream) {
  if (n == 0) return 0;
  S2S_REQUIRE(k >= 2 && k <= 4 && x0 && x1 && out && (k < 3 || x2) && (k < 4 || x3), "add_n: 2 to 4 inputs");
  hipStream_t st = (hipStream_t)stream;
  if (k < 3) x2 = nullptr;
  if (k < 4) x3 = nullptr;
  const uintptr_t al = (uintptr_t)x0 | (uintptr_t)x1 | (uintptr_t)x2 | (uintptr_t)x3 | (uintptr_t)out;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(add_n_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, n, (const float*)x0, (const float*)x1, (const float*)x2,
                       (const float*)x3, (float*)out);
  else if (n % 8 == 0 && al % 16 == 0)
    hipLaunchKernelGGL(add_n_vec_bf16_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, st, n / 8, (const uint4*)x0, (const uint4*)x1,
                       (const uint4*)x2, (const uint4*)x3, (uint4*)out);
  else
    hipLaunchKernelGGL(add_n_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, n, (const bf16_t*)x0, (const bf16_t*)x1, (const bf16_t*)x2,
                       (const bf16_t*)x3, (bf16_t*)out);
  S2S_CHECK_LAUNCH("add_n_kernel");
  return 0;
}

extern "C" int s2svc_add_head_bias(int dtype, int64_t rows, int D, const void* q, const float* u, const float* v,
                                   void* qu, void* qv, void* stream) {
  const int64_t n = rows * D;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(add_head_bias_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, n, D, (const float*)q, u, v, (float*)qu, (float*)qv);
  else
    hipLaunchKernelGGL(add_head_bias_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, n, D, (const bf16_t*)q, u, v, (bf16_t*)qu, (bf16_t*)qv);
  S2S_CHECK_LAUNCH("add_head_bias_kernel");
  return 0;
}

extern "C" int s2svc_add_head_bias_ld(int dtype, int64_t rows, int D, const void* q, int64_t ldq, const float* u, const float* v,
                                      void* qu, void* qv, void* stream) {
  S2S_REQUIRE(rows >= 0 && D > 0 && ldq >= D, "add_head_bias_ld: bad args");
  const int64_t n = rows * D;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(add_head_bias_ld_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, n, D, (const float*)q, ldq, u, v, (float*)qu, (float*)qv);
  else
    hipLaunchKernelGGL(add_head_bias_ld_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, n, D, (const bf16_t*)q, ldq, u, v, (bf16_t*)qu, (bf16_t*)qv);
  S2S_CHECK_LAUNCH("add_head_bias_ld_kernel");
  return 0;
}

extern "C" int s2svc_add_rows(int dtype, int64_t rows, int D, const void* a, int64_t lda, const void* b, int64_t ldb, void* out,
                              int64_t ldo, void* stream) {
  S2S_REQUIRE(rows >= 0 && D > 0 && lda >= D && ldb >= D && ldo >= D, "add_rows: bad args");
  const int64_t n = rows * D;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(add_rows_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, n, D, (const float*)a, lda, (const float*)b, ldb, (float*)out, ldo);
  else
    hipLaunchKernelGGL(add_rows_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, n, D, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo);
  S2S_CHECK_LAUNCH("add_rows_kernel");
  return 0;
}

extern "C" int s2svc_glu_fwd(int dtype, int64_t rows, int C, const void* x, void* y, void* stream) {
  const int64_t n = rows * C;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(glu_fwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, rows, C, (const float*)x, (float*)y);
  else
    hipLaunchKernelGGL(glu_fwd_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, rows, C, (const bf16_t*)x, (bf16_t*)y);
  S2S_CHECK_LAUNCH("glu_fwd_kernel");
  return 0;
}

extern "C" int s2svc_glu_bwd(int dtype, int64_t rows, int C, const void* x, const void* dy, void* dx, void* stream) {
  const int64_t n = rows * C;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(glu_bwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, rows, C, (const float*)x, (const float*)dy, (float*)dx);
  else
    hipLaunchKernelGGL(glu_bwd_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, rows, C, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx);
  S2S_CHECK_LAUNCH("glu_bwd_kernel");
  return 0;
}

// dtype pairs: (in_dtype, out_dtype)
extern "C" int s2svc_cast(int in_dtype, int out_dtype, int64_t n, const void* x, void* y, void* stream) {
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 g(ew_blocks(n)), b(256);
  if (in_dtype == S2S_F32 && out_dtype == S2S_BF16)
    hipLaunchKernelGGL((cast_kernel<float, bf16_t>), g, b, 0, st, n, (const float*)x, (bf16_t*)y);
  else if (in_dtype == S2S_BF16 && out_dtype == S2S_F32)
    hipLaunchKernelGGL((cast_kernel<bf16_t, float>), g, b, 0, st, n, (const bf16_t*)x, (float*)y);
  else if (in_dtype == S2S_F32 && out_dtype == S2S_F32)
    hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, st, n, (const float*)x, (float*)y);
  else
    hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), g, b, 0, st, n, (const bf16_t*)x, (bf16_t*)y);
  S2S_CHECK_LAUNCH("cast_kernel");
  return 0;
}

extern "C" int s2svc_gather3_grouped(const s2svc_gather3_job* jobs, int n, void* stream) {
  S2S_REQUIRE(n >= 0 && (n == 0 || jobs), "gather3_grouped: bad args");
  hipStream_t st = (hipStream_t)stream;
  for (int i0 = 0; i0 < n; i0 += S2S_GATHER_MAX) {
    gather_jobs g;
    g.n = (n - i0 < S2S_GATHER_MAX) ? n - i0 : S2S_GATHER_MAX;
    int64_t most = 0;
    for (int i = 0; i < g.n; ++i) {
      g.j[i] = jobs[i0 + i];
      S2S_REQUIRE(g.j[i].in && g.j[i].out && g.j[i].n0 > 0 && g.j[i].n1 > 0 && g.j[i].n2 > 0, "gather3_grouped: bad job");
      const int64_t e = (int64_t)g.j[i].n0 * g.j[i].n1 * g.j[i].n2;
      most = e > most ? e : most;
    }
    int bx = (int)((most + 255) / 256);
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL(gather3_grouped_kernel, dim3(bx, g.n), dim3(256), 0, st, g);
    S2S_CHECK_LAUNCH("gather3_grouped_kernel");
  }
  return 0;
}

// dst[o][b][a] (+)= src[o][a][b]  (fp32; o < n, a < A, b < B): a convolution weight gradient leaves its GEMM as (C_out, taps, C_in) and
// belongs in the parameter's (C_out, C_in, taps) gradient slot.  One workgroup per o: the A x B matrix is read row-contiguous into LDS
// and written row-contiguous from it (gather3 + axpby read 4 bytes per 1.5 KB-apart address: 17-39 us + 17-19 us for 1.3-2.8 M
// elements at the end of VTN's backward pass, where every kernel's time is the step's time).
namespace {
__global__ __launch_bounds__(256) void permute_inner_kernel(int A, int B, const float* __restrict__ src, float* __restrict__ dst,
                                                            int accumulate) {
  extern __shared__ float pi_tile[];          // [A][B + 1]
  const int64_t base = (int64_t)blockIdx.x * A * B;
  const int n = A * B, pitch = B + 1;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int a = i / B, b = i - a * B;
    pi_tile[a * pitch + b] = src[base + i];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {
    const int b = i / A, a = i - b * A;
    const float v = pi_tile[a * pitch + b];
    dst[base + i] = accumulate ? dst[base + i] + v : v;
  }
}
}  // namespace

extern "C" int s2svc_permute_inner(int n, int A, int B, const float* src, float* dst, int accumulate, void* stream) {
  S2S_REQUIRE(n >= 0 && A > 0 && B > 0 && (n == 0 || (src && dst)), "permute_inner: bad args");
  S2S_REQUIRE((int64_t)A * (B + 1) * 4 <= 64 * 1024, "permute_inner: one A x B matrix must fit 64 KB of LDS");
  if (n == 0) return 0;
  hipLaunchKernelGGL(permute_inner_kernel, dim3((unsigned)n), dim3(256), (size_t)A * (B + 1) * sizeof(float), (hipStream_t)stream, A, B, src, dst,
                     accumulate);
  S2S_CHECK_LAUNCH("permute_inner_kernel");
  return 0;
}

// out (contiguous, n0 x n1 x n2, out_dtype) = in[off + i0*s0 + i1*s1 + i2*s2] (in_dtype)
extern "C" int s2svc_gather3(int in_dtype, int out_dtype, int n0, int n1, int n2, int64_t s0, int64_t s1, int64_t s2,
                             int64_t off, const void* in, void* out, void* stream) {
  const int64_t n = (int64_t)n0 * n1 * n2;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  dim3 g(ew_blocks(n)), b(256);
  if (in_dtype == S2S_F32 && out_dtype == S2S_BF16)
    hipLaunchKernelGGL((gather3_kernel<float, bf16_t>), g, b, 0, st, n0, n1, n2, s0, s1, s2, off, (const float*)in, (bf16_t*)out);
  else if (in_dtype == S2S_BF16 && out_dtype == S2S_F32)
    hipLaunchKernelGGL((gather3_kernel<bf16_t, float>), g, b, 0, st, n0, n1, n2, s0, s1, s2, off, (const bf16_t*)in, (float*)out);
  else if (in_dtype == S2S_F32 && out_dtype == S2S_F32)
    hipLaunchKernelGGL((gather3_kernel<float, float>), g, b, 0, st, n0, n1, n2, s0, s1, s2, off, (const float*)in, (float*)out);
  else
    hipLaunchKernelGGL((gather3_kernel<bf16_t, bf16_t>), g, b, 0, st, n0, n1, n2, s0, s1, s2, off, (const bf16_t*)in, (bf16_t*)out);
  S2S_CHECK_LAUNCH("gather3_kernel");
  return 0;
}
