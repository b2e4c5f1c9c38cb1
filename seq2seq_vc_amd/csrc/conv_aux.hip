// Helpers around the implicit-GEMM convolutions.
//
// col2im for the 3x3 stride-2 Conv2d of the subsampling front-end (modules/transformer/subsampling.py:
// 58-63): the data gradient is computed as a dense GEMM dCols[M2, 9*C] = dY[M2, O] . Wp[O, 9*C] followed
// by this gather, which sums the (at most 4) taps that touch each input pixel.  Layouts are NHWC.
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

template <typename T>
__global__ void col2im_s2_kernel(int B, int T1, int F1, int C, int T2, int F2, const T* __restrict__ dcols,
                                 T* __restrict__ dx) {
  const int64_t n = (int64_t)B * T1 * F1 * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    int64_t r = i / C;
    const int f1 = (int)(r % F1); r /= F1;
    const int t1 = (int)(r % T1);
    const int b = (int)(r / T1);
    float acc = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int tt = t1 - kh;
      if (tt < 0 || (tt & 1)) continue;
      const int t2 = tt >> 1;
      if (t2 >= T2) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ff = f1 - kw;
        if (ff < 0 || (ff & 1)) continue;
        const int f2 = ff >> 1;
        if (f2 >= F2) continue;
        const int64_t m = ((int64_t)b * T2 + t2) * F2 + f2;
        acc += ldf(dcols + (m * 9 + (kh * 3 + kw)) * C + c);
      }
    }
    stf(dx + i, acc);
  }
}

// 16-byte channel vectors (8 bf16): one thread = one input pixel x 8 channels, <= 4 contributing taps
__global__ void col2im_s2_vec_kernel(int B, int T1, int F1, int C, int T2, int F2, const bf16_t* __restrict__ dcols,
                                     bf16_t* __restrict__ dx) {
  const int cg = C / 8;
  const int64_t n = (int64_t)B * T1 * F1 * cg;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    int64_t r = i / cg;
    const int f1 = (int)(r % F1); r /= F1;
    const int t1 = (int)(r % T1);
    const int b = (int)(r / T1);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int tt = t1 - kh;
      const int t2 = tt >> 1;
      const bool okt = tt >= 0 && !(tt & 1) && t2 < T2;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ff = f1 - kw;
        const int f2 = ff >> 1;
        if (!(okt && ff >= 0 && !(ff & 1) && f2 < F2)) continue;
        const int64_t m = ((int64_t)b * T2 + t2) * F2 + f2;
        const uint4 v = *reinterpret_cast<const uint4*>(dcols + (m * 9 + (kh * 3 + kw)) * C + g * 8);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[2 * e] += __uint_as_float(w[e] << 16); acc[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
      }
    }
    uint4 o;
    o.x = f2bf2(acc[0], acc[1]);
    o.y = f2bf2(acc[2], acc[3]);
    o.z = f2bf2(acc[4], acc[5]);
    o.w = f2bf2(acc[6], acc[7]);
    *reinterpret_cast<uint4*>(dx + (i / cg) * C + g * 8) = o;
  }
}

// nearest-neighbour resampling along time of channel-last rows: y[b, t, :] = x[b, floor(t*Tin/Tout), :]
// (F.interpolate(mode="nearest") as used at models/aas_vc.py:340-349)
// ext_in / ext_out (captured steps on batches that do not fill their padded shape, common.h "absent rows"): the lengths the reference's
// cropped tensors have -- they set the scale; output frames >= *ext_out are written as zero, input frames >= *ext_in never read.
template <typename T>
__global__ void interp_nearest_kernel(int B, int Tin_cap, int Tout_cap, int C, const T* __restrict__ x, T* __restrict__ y,
                                      const int32_t* __restrict__ ext_in, const int32_t* __restrict__ ext_out) {
  const int64_t n = (int64_t)B * Tout_cap * C;
  const int Tin = ext_in ? (ext_in[0] < Tin_cap ? (ext_in[0] > 0 ? ext_in[0] : 1) : Tin_cap) : Tin_cap;
  const int Tout = ext_out ? (ext_out[0] < Tout_cap ? (ext_out[0] > 0 ? ext_out[0] : 1) : Tout_cap) : Tout_cap;
  const float scale = (float)Tin / (float)Tout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int t = (int)(r % Tout_cap);
    const int b = (int)(r / Tout_cap);
    int src = (int)floorf((float)t * scale);
    if (src > Tin - 1) src = Tin - 1;
    if (t < Tout) y[i] = x[((int64_t)b * Tin_cap + src) * C + c];
    else stf(y + i, 0.f);
  }
}
// backward: dx[b, s, :] = sum over t with src(t) == s of dy[b, t, :]
template <typename T>
__global__ void interp_nearest_bwd_kernel(int B, int Tin_cap, int Tout_cap, int C, const T* __restrict__ dy, T* __restrict__ dx,
                                          const int32_t* __restrict__ ext_in, const int32_t* __restrict__ ext_out) {
  const int64_t n = (int64_t)B * Tin_cap * C;
  const int Tin = ext_in ? (ext_in[0] < Tin_cap ? (ext_in[0] > 0 ? ext_in[0] : 1) : Tin_cap) : Tin_cap;
  const int Tout = ext_out ? (ext_out[0] < Tout_cap ? (ext_out[0] > 0 ? ext_out[0] : 1) : Tout_cap) : Tout_cap;
  const float scale = (float)Tin / (float)Tout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int s = (int)(r % Tin_cap);
    const int b = (int)(r / Tin_cap);
    float acc = 0.f;
    if (s >= Tin) { stf(dx + i, 0.f); continue; }
    // candidate output frames: t in [ceil(s/scale) - 1, ceil((s+1)/scale) + 1]
    int lo = (int)floorf((float)s / scale) - 1, hi = (int)ceilf((float)(s + 1) / scale) + 1;
    if (lo < 0) lo = 0;
    if (hi > Tout - 1) hi = Tout - 1;
    for (int t = lo; t <= hi; ++t) {
      int src = (int)floorf((float)t * scale);
      if (src > Tin - 1) src = Tin - 1;
      if (src == s) acc += ldf(dy + ((int64_t)b * Tout_cap + t) * C + c);
    }
    stf(dx + i, acc);
  }
}

inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int s2svc_col2im_s2(int dtype, int B, int T1, int F1, int C, int T2, int F2, const void* dcols, void* dx,
                               void* stream) {
  const int64_t n = (int64_t)B * T1 * F1 * C;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(col2im_s2_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, B, T1, F1, C, T2, F2, (const float*)dcols, (float*)dx);
  else if (C % 8 == 0 && ((uintptr_t)dcols) % 16 == 0 && ((uintptr_t)dx) % 16 == 0)
    hipLaunchKernelGGL(col2im_s2_vec_kernel, dim3(ew_blocks(n / 8)), dim3(256), 0, st, B, T1, F1, C, T2, F2, (const bf16_t*)dcols, (bf16_t*)dx);
  else
    hipLaunchKernelGGL(col2im_s2_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, B, T1, F1, C, T2, F2, (const bf16_t*)dcols, (bf16_t*)dx);
  S2S_CHECK_LAUNCH("col2im_s2_kernel");
  return 0;
}

extern "C" int s2svc_interp_nearest(int dtype, int B, int Tin, int Tout, int C, const void* x, void* y, const int32_t* ext_in,
                                    const int32_t* ext_out, void* stream) {
  const int64_t n = (int64_t)B * Tout * C;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(interp_nearest_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, B, Tin, Tout, C, (const float*)x, (float*)y, ext_in, ext_out);
  else
    hipLaunchKernelGGL(interp_nearest_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, B, Tin, Tout, C, (const bf16_t*)x, (bf16_t*)y, ext_in, ext_out);
  S2S_CHECK_LAUNCH("interp_nearest_kernel");
  return 0;
}

extern "C" int s2svc_interp_nearest_bwd(int dtype, int B, int Tin, int Tout, int C, const void* dy, void* dx, const int32_t* ext_in,
                                        const int32_t* ext_out, void* stream) {
  const int64_t n = (int64_t)B * Tin * C;
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(interp_nearest_bwd_kernel<float>, dim3(ew_blocks(n)), dim3(256), 0, st, B, Tin, Tout, C, (const float*)dy, (float*)dx, ext_in, ext_out);
  else
    hipLaunchKernelGGL(interp_nearest_bwd_kernel<bf16_t>, dim3(ew_blocks(n)), dim3(256), 0, st, B, Tin, Tout, C, (const bf16_t*)dy, (bf16_t*)dx, ext_in, ext_out);
  S2S_CHECK_LAUNCH("interp_nearest_bwd_kernel");
  return 0;
}
