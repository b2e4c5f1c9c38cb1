// Attention probability kernels: scale + (relative-position shift) + length/causal masking + softmax
// + re-mask + dropout, one wavefront per score row, reductions by wave shuffles; and the matching
// backward.  Masks are never materialised: kernels take per-batch key lengths and a causal flag.
//
// reference: modules/transformer/attention.py:63-93 (forward_attention), :237-260 (rel_shift, new),
//            :142-160 (rel_shift, legacy), :278-303 (matrix_ac + matrix_bd) / sqrt(d_k).
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

// source element of the shifted relative-position term for score (i, j); returns false => 0
__device__ __forceinline__ bool rel_src(int mode, int T1, int i, int j, int& si, int& sc) {
  if (mode == 1) {            // new implementation: bd[i, T1-1-i+j]
    si = i; sc = T1 - 1 - i + j; return true;
  }
  // legacy: pad-left, view (T+1, T), drop first row, view (T, T)
  const int f = T1 + i * T1 + j;
  si = f / (T1 + 1);
  const int c = f - si * (T1 + 1);
  sc = c - 1;
  return c >= 1;
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(int B, int H, int T1, int T2, int ld, const float* __restrict__ scores,
                                                          const float* __restrict__ bd, int Lp, int ldb, int rel_mode, float scale,
                                                          const int32_t* __restrict__ klen, int causal, float p,
                                                          const uint64_t* seed_base, uint64_t seed_off, T* __restrict__ attn, T* __restrict__ pdrop) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nrows = (int64_t)B * H * T1;
  if (row >= nrows) return;
  const int i = (int)(row % T1);
  const int64_t bh = row / T1;
  const int b = (int)(bh / H);
  const int kl = klen ? klen[b] : T2;
  const float* srow = scores + row * ld;   // rows are padded to `ld` >= T2 columns (pad columns are written as 0)
  const float* bdb = bd ? bd + bh * (int64_t)T1 * ldb : nullptr;
  const float NEG = -3.4028234663852886e38f;  // torch.finfo(float32).min

  float mx = NEG;
  for (int j = lane; j < T2; j += 64) {
    const bool ok = (j < kl) && (!causal || j <= i);
    float v = srow[j];
    if (bdb) {
      int si, sc;
      if (rel_src(rel_mode, T1, i, j, si, sc)) v += bdb[(int64_t)si * ldb + sc];
    }
    v = ok ? v * scale : NEG;
    mx = fmaxf(mx, v);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < T2; j += 64) {
    const bool ok = (j < kl) && (!causal || j <= i);
    float v = srow[j];
    if (bdb) {
      int si, sc;
      if (rel_src(rel_mode, T1, i, j, si, sc)) v += bdb[(int64_t)si * ldb + sc];
    }
    v = ok ? v * scale : NEG;
    sum += expf(v - mx);
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int j = lane; j < T2; j += 64) {
    const bool ok = (j < kl) && (!causal || j <= i);
    float v = srow[j];
    if (bdb) {
      int si, sc;
      if (rel_src(rel_mode, T1, i, j, si, sc)) v += bdb[(int64_t)si * ldb + sc];
    }
    v = ok ? v * scale : NEG;
    float pr = ok ? expf(v - mx) * inv : 0.f;  // masked_fill(mask, 0.0) after softmax
    const int64_t o = row * ld + j;
    stf(attn + o, pr);
    if (pdrop) {
      float m = p > 0.f ? dropout_scale(seed, (uint64_t)o, p, inv_keep) : 1.f;
      stf(pdrop + o, pr * m);
    }
  }
  for (int j = T2 + lane; j < ld; j += 64) {
    stf(attn + row * ld + j, 0.f);
    if (pdrop) stf(pdrop + row * ld + j, 0.f);
  }
}

// zero `bytes` bytes at p (16-byte aligned head assumed: torch allocations; byte tail handled)
__global__ void zero_bytes_kernel(char* __restrict__ p, size_t bytes) {
  const size_t n16 = bytes / 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    reinterpret_cast<uint4*>(p)[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0)
    for (size_t i = n16 * 16 + threadIdx.x; i < bytes; i += blockDim.x) p[i] = 0;
}

// dS = P * (dP*mask - sum_j P*dP*mask) ; dscores = dS*scale ; scatter of the same value into dbd
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(int B, int H, int T1, int T2, int ld, const T* __restrict__ attn,
                                                          const float* __restrict__ dp, const T* __restrict__ dattn, float scale, float p, const uint64_t* seed_base, uint64_t seed_off,
                                                          T* __restrict__ dscores, T* __restrict__ dbd, int Lp, int ldb,
                                                          int rel_mode) {
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nrows = (int64_t)B * H * T1;
  if (row >= nrows) return;
  const int i = (int)(row % T1);
  const int64_t bh = row / T1;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  float dot = 0.f;
  for (int j = lane; j < T2; j += 64) {
    const int64_t o = row * ld + j;
    float m = p > 0.f ? dropout_scale(seed, (uint64_t)o, p, inv_keep) : 1.f;
    dot += ldf(attn + o) * (dp[o] * m + (dattn ? ldf(dattn + o) : 0.f));
  }
  dot = wave_sum(dot);
  T* dbdb = dbd ? dbd + bh * (int64_t)T1 * ldb : nullptr;
  for (int j = T2 + lane; j < ld; j += 64) stf(dscores + row * ld + j, 0.f);
  for (int j = lane; j < T2; j += 64) {
    const int64_t o = row * ld + j;
    float m = p > 0.f ? dropout_scale(seed, (uint64_t)o, p, inv_keep) : 1.f;
    float pr = ldf(attn + o);
    float ds = pr * (dp[o] * m + (dattn ? ldf(dattn + o) : 0.f) - dot) * scale;
    stf(dscores + o, ds);
    if (dbdb) {
      int si, sc;
      if (rel_src(rel_mode, T1, i, j, si, sc)) stf(dbdb + (int64_t)si * ldb + sc, ds);
    }
  }
}

}  // namespace

extern "C" int s2svc_attn_softmax_fwd(int dtype, int B, int H, int T1, int T2, int ld, const float* scores, const float* bd,
                                      int Lp, int ldb, int rel_mode, float scale, const int32_t* klen, int causal, float drop_p,
                                      const uint64_t* seed_base, uint64_t seed_off, void* attn, void* pdrop, void* stream) {
  S2S_REQUIRE(B >= 0 && H > 0 && T1 >= 0 && T2 > 0 && ld >= T2, "attn_softmax_fwd: bad shape");
  S2S_REQUIRE(!bd || (rel_mode == 1 && Lp == 2 * T1 - 1 && T1 == T2) || (rel_mode == 2 && Lp == T1 && T1 == T2),
              "attn_softmax_fwd: bad relative-position shape");
  S2S_REQUIRE(!bd || ldb >= Lp, "attn_softmax_fwd: bd row stride smaller than its length");
  const int64_t nrows = (int64_t)B * H * T1;
  if (nrows == 0) return 0;
  dim3 grid((unsigned)((nrows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(softmax_fwd_kernel<float>, grid, block, 0, st, B, H, T1, T2, ld, scores, bd, Lp, ldb, rel_mode, scale, klen,
                       causal, drop_p, seed_base, seed_off, (float*)attn, (float*)pdrop);
  else
    hipLaunchKernelGGL(softmax_fwd_kernel<bf16_t>, grid, block, 0, st, B, H, T1, T2, ld, scores, bd, Lp, ldb, rel_mode, scale, klen,
                       causal, drop_p, seed_base, seed_off, (bf16_t*)attn, (bf16_t*)pdrop);
  S2S_CHECK_LAUNCH("softmax_fwd_kernel");
  return 0;
}

extern "C" int s2svc_attn_softmax_bwd(int dtype, int B, int H, int T1, int T2, int ld, const void* attn, const float* dp, const void* dattn,
                                      float scale, float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* dscores, void* dbd, int Lp,
                                      int ldb, int rel_mode, void* stream) {
  S2S_REQUIRE(B >= 0 && H > 0 && T1 >= 0 && T2 > 0 && (!dbd || ldb >= Lp), "attn_softmax_bwd: bad shape");
  const int64_t nrows = (int64_t)B * H * T1;
  if (nrows == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const size_t esz = dtype == S2S_F32 ? 4 : 2;
  if (dbd) {
    // dbd is a scatter target: zero it with a KERNEL.  (hipMemsetAsync here became a memset node under stream capture, and a
    // captured step replayed more than once then returned garbage in dbd on this ROCm build -- the first replay was right, eager
    // launches were right; every gradient below the relative-position attention inherited it.)
    const size_t bytes = (size_t)B * H * T1 * ldb * esz;
    const size_t n16 = bytes / 16;
    int zb = (int)((n16 + 255) / 256);
    if (zb > 4096) zb = 4096;
    if (zb < 1) zb = 1;
    hipLaunchKernelGGL(zero_bytes_kernel, dim3(zb), dim3(256), 0, st, (char*)dbd, bytes);
    S2S_CHECK_LAUNCH("zero_bytes_kernel");
  }
  dim3 grid((unsigned)((nrows + 3) / 4)), block(256);
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(softmax_bwd_kernel<float>, grid, block, 0, st, B, H, T1, T2, ld, (const float*)attn, dp, (const float*)dattn, scale, drop_p,
                       seed_base, seed_off, (float*)dscores, (float*)dbd, Lp, ldb, rel_mode);
  else
    hipLaunchKernelGGL(softmax_bwd_kernel<bf16_t>, grid, block, 0, st, B, H, T1, T2, ld, (const bf16_t*)attn, dp, (const bf16_t*)dattn, scale,
                       drop_p, seed_base, seed_off, (bf16_t*)dscores, (bf16_t*)dbd, Lp, ldb, rel_mode);
  S2S_CHECK_LAUNCH("softmax_bwd_kernel");
  return 0;
}
