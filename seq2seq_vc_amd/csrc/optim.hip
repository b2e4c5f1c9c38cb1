// Flat-buffer optimiser step: global grad-norm (deterministic two-stage) -> clip coefficient ->
// WarmupLR -> Adam, plus the bf16 shadow copy of the updated weights, in three launches over ONE
// contiguous fp32 parameter buffer (all parameters of the model are views into it).
//
// reference: trainers/ar_vc.py:99-107 (zero_grad / backward / clip_grad_norm_ / optimizer.step /
// scheduler.step), torch.optim.Adam (betas 0.9/0.999, eps 1e-8, no weight decay),
// schedulers/warmup_lr.py:54-61.  The step counter and the learning rate live in device memory so the
// whole training step can be replayed from a hipGraph.
//
// HBM traffic per step: 16 B/param read (p, g, m, v) + 12 B/param written (p, m, v) [+2 B bf16 shadow].
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(int64_t n, const float* __restrict__ g, double* __restrict__ partial) {
  __shared__ double sh[4];
  // 16-byte loads, four independent fp64 accumulators per thread (the scalar version ran at 2.2 TB/s)
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const int64_t n4 = (((uintptr_t)g) % 16 == 0) ? n / 4 : 0;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = g4[i];
    a0 += (double)v.x * (double)v.x;
    a1 += (double)v.y * (double)v.y;
    a2 += (double)v.z * (double)v.z;
    a3 += (double)v.w * (double)v.w;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = (double)g[i];
    a0 += v * v;
  }
  double acc = (a0 + a1) + (a2 + a3);
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// state[0] = step (as float, incremented here), state[1] = lr used this step, state[2] = grad norm,
// state[3] = clip coefficient
__global__ void adam_prepare_kernel(int nblk, const double* __restrict__ partial, float max_norm, float base_lr,
                                    float warmup_steps, float* __restrict__ state) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 64) acc += partial[i];
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(acc);
    float coef = 1.f;
    if (max_norm > 0.f) {
      coef = max_norm / (norm + 1e-6f);
      if (coef > 1.f) coef = 1.f;
    }
    const float step = state[0] + 1.f;
    float lr = base_lr;
    if (warmup_steps > 0.f) {
      // lr_k used at optimiser step k (1-based) is the scheduler's value after k-1 scheduler.step()
      // calls: base * w^0.5 * min(k^-0.5, k * w^-1.5)
      const float a = rsqrtf(step), b = step * powf(warmup_steps, -1.5f);
      lr = base_lr * sqrtf(warmup_steps) * (a < b ? a : b);
    }
    state[0] = step;
    state[1] = lr;
    state[2] = norm;
    state[3] = coef;
  }
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float coef, float beta1, float beta2, float eps,
                                         float step_size, float inv_sqrt_bc2) {
  const float gi = g * coef;
  m = beta1 * m + (1.f - beta1) * gi;
  v = beta2 * v + (1.f - beta2) * gi * gi;
  const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
  p = p - step_size * (m / denom);
}

// 28 (+2) bytes of traffic per parameter and nothing else: four parameters per thread and iteration, every stream in 16-byte
// accesses (n is a multiple of 4: the flat buffers are padded to 64 elements)
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* q) {
  const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(q));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nt_store4(float* q, float4 a) {
  f32x4_nt t = {a.x, a.y, a.z, a.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4_nt*>(q));
}

template <bool NT>
__global__ void adam_update_kernel(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                   float* __restrict__ v, bf16_t* __restrict__ shadow, float beta1, float beta2, float eps,
                                   const float* __restrict__ state) {
  const float step = state[0], lr = state[1], coef = state[3];
  const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
  const float step_size = lr / bc1, inv_sqrt_bc2 = 1.f / sqrtf(bc2);
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 pp, gg, mm, vv;
    if (NT) {            // streamed once per step: no reuse to keep in L2 / MALL
      pp = nt_load4(p + 4 * i); gg = nt_load4(g + 4 * i); mm = nt_load4(m + 4 * i); vv = nt_load4(v + 4 * i);
    } else {
      pp = reinterpret_cast<const float4*>(p)[i];
      gg = reinterpret_cast<const float4*>(g)[i];
      mm = reinterpret_cast<const float4*>(m)[i];
      vv = reinterpret_cast<const float4*>(v)[i];
    }
    adam_one(pp.x, gg.x, mm.x, vv.x, coef, beta1, beta2, eps, step_size, inv_sqrt_bc2);
    adam_one(pp.y, gg.y, mm.y, vv.y, coef, beta1, beta2, eps, step_size, inv_sqrt_bc2);
    adam_one(pp.z, gg.z, mm.z, vv.z, coef, beta1, beta2, eps, step_size, inv_sqrt_bc2);
    adam_one(pp.w, gg.w, mm.w, vv.w, coef, beta1, beta2, eps, step_size, inv_sqrt_bc2);
    if (NT) {
      nt_store4(m + 4 * i, mm); nt_store4(v + 4 * i, vv);
    } else {
      reinterpret_cast<float4*>(m)[i] = mm;
      reinterpret_cast<float4*>(v)[i] = vv;
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    if (shadow) {
      uint2 o;
      o.x = f2bf2(pp.x, pp.y);
      o.y = f2bf2(pp.z, pp.w);
      reinterpret_cast<uint2*>(shadow)[i] = o;
    }
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    adam_one(pi, g[i], mi, vi, coef, beta1, beta2, eps, step_size, inv_sqrt_bc2);
    m[i] = mi;
    v[i] = vi;
    p[i] = pi;
    if (shadow) shadow[i] = f2bf(pi);
  }
}

}  // namespace

// partial: >= 1024 doubles.  state: 4 floats in device memory (see adam_prepare_kernel).
extern "C" int s2svc_adam_step(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                               void* bf16_shadow, float beta1, float beta2, float eps, float max_norm, float base_lr,
                               float warmup_steps, double* partial, float* state, void* stream) {
  S2S_REQUIRE(n > 0 && params && grads && exp_avg && exp_avg_sq && partial && state, "adam_step: bad args");
  hipStream_t st = (hipStream_t)stream;
  int nb = (int)((n + 255) / 256);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, n, grads, partial);
  S2S_CHECK_LAUNCH("sumsq_kernel");
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, st, nb, partial, max_norm, base_lr, warmup_steps, state);
  S2S_CHECK_LAUNCH("adam_prepare_kernel");
  S2S_REQUIRE(((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16 == 0 &&
                  (!bf16_shadow || ((uintptr_t)bf16_shadow) % 8 == 0), "adam_step: 16-byte aligned buffers");
  // one thread per four parameters, no grid-stride cap: measured on 157.5 M / 30.5 M parameters (sumsq + prepare + update,
  // one box) 1017 / 183 us against 1150 / 190 with 4096 blocks striding and 1085 / 199 for the one-parameter-per-thread form
  const int64_t ub = (n / 4 + 255) / 256 > 0 ? (n / 4 + 255) / 256 : 1;
  // non-temporal loads / stores of the moments when a buffer is larger than the memory-side cache (256 MB): nothing of this pass is
  // read again before it is evicted (AAS-VC, 630 MB per buffer: 11.61 -> 11.58 ms per step; VTN, 122 MB: neutral, left cached)
  const bool nt = n * 4 > (256ll << 20);
  if (nt) hipLaunchKernelGGL(adam_update_kernel<true>, dim3((unsigned)ub), dim3(256), 0, st, n, params, grads, exp_avg, exp_avg_sq,
                             (bf16_t*)bf16_shadow, beta1, beta2, eps, state);
  else hipLaunchKernelGGL(adam_update_kernel<false>, dim3((unsigned)ub), dim3(256), 0, st, n, params, grads, exp_avg, exp_avg_sq,
                          (bf16_t*)bf16_shadow, beta1, beta2, eps, state);
  S2S_CHECK_LAUNCH("adam_update_kernel");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Transposed bf16 shadow of the 2-D weights: dst[c, r] = src[r, c] for every matrix of a table, ONE launch.
// The data-gradient GEMMs dX = dY . W read W with the reduction index (out-features) strided; with W^T kept beside
// the bf16 shadow they become plain K-contiguous x K-contiguous products and take the all-DMA kernel.
// tiles: one int4-sized entry per 64x64 tile, {src offset of the matrix, dst offset, rows << 32 | cols, tile index}
// (element offsets into src / dst).  Refreshed once per optimiser step (2 B read + 2 B written per parameter).
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void transpose_tiles_kernel(const int64_t* __restrict__ tiles, const bf16_t* __restrict__ src,
                                                              bf16_t* __restrict__ dst) {
  // 64 x 64 tile through LDS (row pitch 66: the column reads of phase 2 step 33 dwords -> conflict-free); 16-byte global
  // loads and stores when the matrix allows it (rows, cols, offsets multiples of 8), element-wise otherwise
  __shared__ bf16_t t[64][66];
  const int64_t* e = tiles + (int64_t)blockIdx.x * 4;
  const int64_t so = e[0], dof = e[1];
  const int rows = (int)(e[2] >> 32), cols = (int)(e[2] & 0xffffffff);
  const int tcols = (cols + 63) / 64;
  const int tr = (int)(e[3] / tcols), tc = (int)(e[3] % tcols);
  const bool vec = (rows % 8 == 0) && (cols % 8 == 0) && (so % 8 == 0) && (dof % 8 == 0);
  if (vec) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {                      // 64 rows x 8 vectors = 512 vector loads
      const int v = threadIdx.x + i * 256;
      const int r = tr * 64 + (v >> 3), c = tc * 64 + (v & 7) * 8;
      if (r < rows && c < cols) {
        const uint4 q = *reinterpret_cast<const uint4*>(src + so + (int64_t)r * cols + c);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          t[v >> 3][(v & 7) * 8 + 2 * k] = (bf16_t)(w[k] & 0xffffu);
          t[v >> 3][(v & 7) * 8 + 2 * k + 1] = (bf16_t)(w[k] >> 16);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {                      // output row = source column; 8 consecutive source rows per vector
      const int v = threadIdx.x + i * 256;
      const int cl = v >> 3, rl = (v & 7) * 8;
      const int c = tc * 64 + cl, r = tr * 64 + rl;
      if (c < cols && r < rows) {
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = (uint32_t)t[rl + 2 * k][cl] | ((uint32_t)t[rl + 2 * k + 1][cl] << 16);
        *reinterpret_cast<uint4*>(dst + dof + (int64_t)c * rows + r) = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    return;
  }
  for (int v = threadIdx.x; v < 64 * 64; v += 256) {
    const int r = tr * 64 + (v >> 6), c = tc * 64 + (v & 63);
    if (r < rows && c < cols) t[v >> 6][v & 63] = src[so + (int64_t)r * cols + c];
  }
  __syncthreads();
  for (int v = threadIdx.x; v < 64 * 64; v += 256) {
    const int c = tc * 64 + (v >> 6), r = tr * 64 + (v & 63);
    if (r < rows && c < cols) dst[dof + (int64_t)c * rows + r] = t[v & 63][v >> 6];
  }
}
}  // namespace

extern "C" int s2svc_transpose_tiles(int64_t ntiles, const int64_t* tiles, const void* src, void* dst, void* stream) {
  S2S_REQUIRE(ntiles >= 0 && (ntiles == 0 || (tiles && src && dst)), "transpose_tiles: bad args");
  if (ntiles == 0) return 0;
  hipLaunchKernelGGL(transpose_tiles_kernel, dim3((unsigned)ntiles), dim3(256), 0, (hipStream_t)stream, tiles, (const bf16_t*)src,
                     (bf16_t*)dst);
  S2S_CHECK_LAUNCH("transpose_tiles_kernel");
  return 0;
}
