// STFT -> log-mel front-end helpers.  The two contractions (framing x windowed DFT basis, |spectrum| x
// mel basis) run on the fp32 MFMA GEMM (csrc/gemm.hip; the frame matrix is never materialised: rows of the
// A operand overlap in memory with leading dimension = hop); these kernels do the HBM-bound glue.
// reference: bin/preprocess.py:30-92 (librosa.stft(center=True, pad_mode="reflect"), abs, mel basis,
// np.maximum(eps, .), np.log10).
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

// y[i] = x[reflect(i - pad)], i in [0, n + 2*pad)   (numpy "reflect": edge sample not repeated)
__global__ void reflect_pad_kernel(int64_t n, int pad, const float* __restrict__ x, float* __restrict__ y) {
  const int64_t m = n + 2 * (int64_t)pad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t j = i - pad;
    if (n > 1) {
      const int64_t period = 2 * (n - 1);
      j %= period;
      if (j < 0) j += period;
      if (j >= n) j = period - j;
    } else {
      j = 0;
    }
    y[i] = x[j];
  }
}

// spc[m, k] = sqrt(re^2 + im^2) with re = z[m, k], im = z[m, nb + k]
__global__ void magnitude_kernel(int64_t frames, int nb, const float* __restrict__ z, float* __restrict__ spc) {
  const int64_t n = frames * nb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / nb;
    const int k = (int)(i - m * nb);
    const float re = z[m * 2 * nb + k], im = z[m * 2 * nb + nb + k];
    spc[i] = sqrtf(re * re + im * im);
  }
}

// y = log_b(max(eps, x)) [* scale + shift per column: optional fused mean/variance normalisation]
__global__ void log_clamp_kernel(int64_t n, int D, const float* __restrict__ x, float eps, float inv_log_base,
                                 const float* __restrict__ mean, const float* __restrict__ inv_scale, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = logf(fmaxf(eps, x[i])) * inv_log_base;
    if (mean) {
      const int c = (int)(i % D);
      v = (v - mean[c]) * inv_scale[c];
    }
    y[i] = v;
  }
}

inline int ew_blocks(int64_t total) {
  int64_t b = (total + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int s2svc_reflect_pad(int64_t n, int pad, const float* x, float* y, void* stream) {
  S2S_REQUIRE(n > 0 && pad >= 0, "reflect_pad: bad args");
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(ew_blocks(n + 2 * pad)), dim3(256), 0, (hipStream_t)stream, n, pad, x, y);
  S2S_CHECK_LAUNCH("reflect_pad_kernel");
  return 0;
}

extern "C" int s2svc_magnitude(int64_t frames, int nb, const float* z, float* spc, void* stream) {
  if (frames == 0) return 0;
  hipLaunchKernelGGL(magnitude_kernel, dim3(ew_blocks(frames * nb)), dim3(256), 0, (hipStream_t)stream, frames, nb, z, spc);
  S2S_CHECK_LAUNCH("magnitude_kernel");
  return 0;
}

extern "C" int s2svc_log_clamp(int64_t n, int D, const float* x, float eps, float inv_log_base, const float* mean,
                               const float* inv_scale, float* y, void* stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(log_clamp_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, n, D, x, eps, inv_log_base, mean,
                     inv_scale, y);
  S2S_CHECK_LAUNCH("log_clamp_kernel");
  return 0;
}
