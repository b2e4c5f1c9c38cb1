// Length regulator of FastSpeech-style models and the duration extraction that feeds it.
//   * s2svc_length_regulate_{index,fwd,bwd}: replaces modules/length_regulator.py:46-97 (a Python list of
//     torch.repeat_interleave calls + pad_list): frame i of utterance b is repeated ds[b, i] times along time.
//       index : exclusive prefix sums of the durations (start[b, i]) and the source frame of every output frame
//               (idx[b, t] = i, or -1 past the utterance's total duration) -- one workgroup per utterance;
//       fwd   : y[b, t, :] = idx >= 0 ? x[b, idx, :] : pad  (16-byte channel vectors);
//       bwd   : dx[b, i, :] = sum of dy over the frame's run [start, start + d): a fixed-order segment sum (no atomics).
//   * s2svc_attn_durations: replaces utils/duration_calculator.py:13-65 for the Transformer case: picks the most diagonal
//     attention head (largest mean over output frames of the row maximum), counts for every input position how many output
//     frames have their arg-max there, and reports the focus rate.
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

__global__ __launch_bounds__(256) void lr_index_kernel(int Tx, int Tout, const int32_t* __restrict__ ds, int32_t* __restrict__ start,
                                                       int32_t* __restrict__ idx, int32_t* __restrict__ total) {
  __shared__ int scan[256];
  __shared__ int carry_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) carry_s = 0;
  for (int t = tid; t < Tout; t += 256) idx[(int64_t)b * Tout + t] = -1;
  __syncthreads();
  for (int base = 0; base < Tx; base += 256) {
    const int i = base + tid;
    int dur = i < Tx ? ds[(int64_t)b * Tx + i] : 0;
    if (dur < 0) dur = 0;
    scan[tid] = dur;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {                 // Hillis-Steele inclusive scan
      const int v = tid >= o ? scan[tid - o] : 0;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    const int carry = carry_s;
    const int st = carry + scan[tid] - dur;
    if (i < Tx) {
      start[(int64_t)b * Tx + i] = st;
      for (int t = st; t < st + dur && t < Tout; ++t) idx[(int64_t)b * Tout + t] = i;
    }
    __syncthreads();
    if (tid == 255) carry_s = carry + scan[255];
    __syncthreads();
  }
  if (tid == 0 && total) total[b] = carry_s;
}

// VEC channels per thread (16 bytes): bf16 x 8 / fp32 x 4; scalar fallback with VEC = 1
template <typename T, int VEC>
__global__ void lr_fwd_kernel(int64_t n_vec, int Tx, int Tout, int D, const T* __restrict__ x, const int32_t* __restrict__ idx, float pad,
                              T* __restrict__ y) {
  const int dv = D / VEC;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_vec; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % dv);
    const int64_t bt = e / dv;
    const int b = (int)(bt / Tout);
    const int i = idx[bt];
    T* dst = y + bt * D + (int64_t)c * VEC;
    if (i >= 0) {
      const T* src = x + ((int64_t)b * Tx + i) * D + (int64_t)c * VEC;
      if (VEC > 1) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
      else *dst = *src;
    } else {
#pragma unroll
      for (int v = 0; v < VEC; ++v) stf(dst + v, pad);
    }
  }
}

template <typename T>
__global__ void lr_bwd_kernel(int64_t n, int Tx, int Tout, int D, const T* __restrict__ dy, const int32_t* __restrict__ start,
                              const int32_t* __restrict__ ds, T* __restrict__ dx) {
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % D);
    const int64_t bi = e / D;
    const int b = (int)(bi / Tx);
    const int st = start[bi];
    int dur = ds[bi];
    if (dur < 0) dur = 0;
    float acc = 0.f;
    for (int t = st; t < st + dur && t < Tout; ++t) acc += ldf(dy + ((int64_t)b * Tout + t) * D + c);
    stf(dx + e, acc);
  }
}

// one workgroup: att (NH, Tf, Tx) fp32
__global__ __launch_bounds__(256) void attn_durations_kernel(int NH, int Tf, int Tx, const float* __restrict__ att, int64_t* __restrict__ dur,
                                                             float* __restrict__ focus, int32_t* __restrict__ head) {
  __shared__ float red[256];
  __shared__ float best_s;
  __shared__ int best_h;
  const int tid = threadIdx.x;
  if (tid == 0) { best_s = -1.f; best_h = 0; }
  __syncthreads();
  for (int h = 0; h < NH; ++h) {              // diagonal score of head h: mean over frames of the row maximum
    float part = 0.f;
    for (int t = tid; t < Tf; t += 256) {
      const float* row = att + ((int64_t)h * Tf + t) * Tx;
      float m = row[0];
      for (int j = 1; j < Tx; ++j) m = fmaxf(m, row[j]);
      part += m;
    }
    red[tid] = part;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    if (tid == 0) {
      const float s = red[0] / (float)Tf;
      if (s > best_s) { best_s = s; best_h = h; }       // first maximum, as torch.argmax
    }
    __syncthreads();
  }
  const int h = best_h;
  for (int j = tid; j < Tx; j += 256) {
    int64_t n = 0;
    for (int t = 0; t < Tf; ++t) {            // frames whose (first) arg-max is j
      const float* row = att + ((int64_t)h * Tf + t) * Tx;
      const float v = row[j];
      bool is = true;
      for (int k = 0; k < Tx && is; ++k) {
        const float u = row[k];
        if (u > v || (u == v && k < j)) is = false;
      }
      n += is ? 1 : 0;
    }
    dur[j] = n;
  }
  if (tid == 0) {
    if (focus) *focus = best_s;
    if (head) *head = h;
  }
}

inline int lr_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int s2svc_length_regulate_index(int B, int Tx, int Tout, const int32_t* ds, int32_t* start, int32_t* idx, int32_t* total,
                                           void* stream) {
  S2S_REQUIRE(B > 0 && Tx > 0 && Tout >= 0 && ds && start && (idx || Tout == 0), "length_regulate_index: bad args");
  hipLaunchKernelGGL(lr_index_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, Tx, Tout, ds, start, idx, total);
  S2S_CHECK_LAUNCH("lr_index_kernel");
  return 0;
}

extern "C" int s2svc_length_regulate_fwd(int dtype, int B, int Tx, int Tout, int D, const void* x, const int32_t* idx, float pad_value,
                                         void* y, void* stream) {
  S2S_REQUIRE(B > 0 && Tx > 0 && D > 0 && x && idx && y, "length_regulate_fwd: bad args");
  if (Tout == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)B * Tout * D;
  const bool al = ((uintptr_t)x) % 16 == 0 && ((uintptr_t)y) % 16 == 0;
  if (dtype == S2S_F32) {
    if (D % 4 == 0 && al)
      hipLaunchKernelGGL((lr_fwd_kernel<float, 4>), dim3(lr_blocks(n / 4)), dim3(256), 0, st, n / 4, Tx, Tout, D, (const float*)x, idx, pad_value, (float*)y);
    else
      hipLaunchKernelGGL((lr_fwd_kernel<float, 1>), dim3(lr_blocks(n)), dim3(256), 0, st, n, Tx, Tout, D, (const float*)x, idx, pad_value, (float*)y);
  } else {
    if (D % 8 == 0 && al)
      hipLaunchKernelGGL((lr_fwd_kernel<bf16_t, 8>), dim3(lr_blocks(n / 8)), dim3(256), 0, st, n / 8, Tx, Tout, D, (const bf16_t*)x, idx, pad_value, (bf16_t*)y);
    else
      hipLaunchKernelGGL((lr_fwd_kernel<bf16_t, 1>), dim3(lr_blocks(n)), dim3(256), 0, st, n, Tx, Tout, D, (const bf16_t*)x, idx, pad_value, (bf16_t*)y);
  }
  S2S_CHECK_LAUNCH("lr_fwd_kernel");
  return 0;
}

extern "C" int s2svc_length_regulate_bwd(int dtype, int B, int Tx, int Tout, int D, const void* dy, const int32_t* start, const int32_t* ds,
                                         void* dx, void* stream) {
  S2S_REQUIRE(B > 0 && Tx > 0 && D > 0 && dy && start && ds && dx, "length_regulate_bwd: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n = (int64_t)B * Tx * D;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(lr_bwd_kernel<float>, dim3(lr_blocks(n)), dim3(256), 0, st, n, Tx, Tout, D, (const float*)dy, start, ds, (float*)dx);
  else
    hipLaunchKernelGGL(lr_bwd_kernel<bf16_t>, dim3(lr_blocks(n)), dim3(256), 0, st, n, Tx, Tout, D, (const bf16_t*)dy, start, ds, (bf16_t*)dx);
  S2S_CHECK_LAUNCH("lr_bwd_kernel");
  return 0;
}

extern "C" int s2svc_attn_durations(int NH, int Tf, int Tx, const float* att, int64_t* durations, float* focus_rate, int32_t* head,
                                    void* stream) {
  S2S_REQUIRE(NH > 0 && Tf > 0 && Tx > 0 && att && durations, "attn_durations: bad args");
  hipLaunchKernelGGL(attn_durations_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, NH, Tf, Tx, att, durations, focus_rate, head);
  S2S_CHECK_LAUNCH("attn_durations_kernel");
  return 0;
}
