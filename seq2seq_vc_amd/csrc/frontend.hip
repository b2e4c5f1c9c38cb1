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

// ------------------------------------------------------------------------------------------------
// Batched front-end (B utterances per launch): wav batch -> normalised, zero-padded (B, Tmax, n_mels) log-mel batch +
// frame counts, i.e. what bin/preprocess.py (per utterance), bin/normalize.py and the collater's padding produce in three
// passes over HDF5 files.  Launches per batch: reflect_pad_batch, ONE fp32-MFMA GEMM (framing x windowed DFT basis, all
// utterances as batch planes), mel_log_batch (magnitude + mel projection + clamp + log + mean/variance normalisation +
// padding zeros).  The mel projection uses the filterbank's sparsity: a triangular filter touches bins [lo, hi) only
// (~1000 non-zeros of 80 x 513).
// ------------------------------------------------------------------------------------------------
namespace {

// y (B, ld): y[b, i] = x[b, reflect(i - pad)] for i < n_b + 2*pad, 0 behind (so that frames past the utterance read zeros)
__global__ void reflect_pad_batch_kernel(int B, int64_t Nmax, int pad, int64_t ld, const float* __restrict__ x,
                                         const int32_t* __restrict__ nlen, float* __restrict__ y) {
  const int64_t total = (int64_t)B * ld;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(e / ld);
    const int64_t i = e - (int64_t)b * ld;
    const int64_t n = nlen[b];
    float v = 0.f;
    if (n > 0 && i < n + 2 * (int64_t)pad) {
      int64_t j = i - pad;
      if (n > 1) {
        const int64_t period = 2 * (n - 1);
        j %= period;
        if (j < 0) j += period;
        if (j >= n) j = period - j;
      } else {
        j = 0;
      }
      v = x[(int64_t)b * Nmax + j];
    }
    y[e] = v;
  }
}

// one workgroup per (utterance, frame): z row = [re(0..nb) | im(0..nb)]
__global__ __launch_bounds__(128) void mel_log_batch_kernel(int Tmax, int nb, int nmel, const float* __restrict__ z,
                                                            const int32_t* __restrict__ frames, const float* __restrict__ melb,
                                                            const int32_t* __restrict__ lo, const int32_t* __restrict__ hi, float eps,
                                                            float inv_log_base, const float* __restrict__ mean,
                                                            const float* __restrict__ inv_scale, float* __restrict__ out) {
  extern __shared__ float mag[];
  const int t = blockIdx.x, b = blockIdx.y;
  float* o = out + ((int64_t)b * Tmax + t) * nmel;
  if (t >= frames[b]) {                      // padding frame of the batch
    for (int m = threadIdx.x; m < nmel; m += blockDim.x) o[m] = 0.f;
    return;
  }
  const float* zr = z + ((int64_t)b * Tmax + t) * (2 * (int64_t)nb);
  for (int k = threadIdx.x; k < nb; k += blockDim.x) {
    const float re = zr[k], im = zr[nb + k];
    mag[k] = sqrtf(re * re + im * im);
  }
  __syncthreads();
  for (int m = threadIdx.x; m < nmel; m += blockDim.x) {
    const float* w = melb + (int64_t)m * nb;
    float acc = 0.f;
    for (int k = lo[m]; k < hi[m]; ++k) acc += w[k] * mag[k];
    float v = logf(fmaxf(eps, acc)) * inv_log_base;
    if (mean) v = (v - mean[m]) * inv_scale[m];
    o[m] = v;
  }
}

// ragged feature rows -> zero-padded batch, optional per-column normalisation and stop labels (the collater's work:
// collaters/ar_vc.py:20-73).  offsets: B + 1 row offsets into `ragged` (rows x D).
__global__ void ragged_to_padded_kernel(int B, int Tmax, int D, const float* __restrict__ ragged, const int64_t* __restrict__ offsets,
                                        const float* __restrict__ mean, const float* __restrict__ inv_scale, float* __restrict__ out,
                                        float* __restrict__ labels) {
  const int64_t total = (int64_t)B * Tmax * D;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(e % D);
    const int64_t bt = e / D;
    const int t = (int)(bt % Tmax), b = (int)(bt / Tmax);
    const int64_t r0 = offsets[b], len = offsets[b + 1] - r0;
    float v = 0.f;
    if (t < len) {
      v = ragged[(r0 + t) * D + c];
      if (mean) v = (v - mean[c]) * inv_scale[c];
    }
    out[e] = v;
    if (labels && c == 0) labels[bt] = t >= len - 1 ? 1.f : 0.f;
  }
}

}  // namespace

extern "C" int s2svc_reflect_pad_batch(int B, int64_t Nmax, int pad, int64_t ld, const float* x, const int32_t* nlen, float* y,
                                       void* stream) {
  S2S_REQUIRE(B > 0 && Nmax > 0 && pad >= 0 && ld >= Nmax + 2 * pad && x && nlen && y, "reflect_pad_batch: bad args");
  hipLaunchKernelGGL(reflect_pad_batch_kernel, dim3(ew_blocks((int64_t)B * ld)), dim3(256), 0, (hipStream_t)stream, B, Nmax, pad, ld, x, nlen, y);
  S2S_CHECK_LAUNCH("reflect_pad_batch_kernel");
  return 0;
}

extern "C" int s2svc_mel_log_batch(int B, int Tmax, int nb, int nmel, const float* z, const int32_t* frames, const float* melb,
                                   const int32_t* lo, const int32_t* hi, float eps, float inv_log_base, const float* mean,
                                   const float* inv_scale, float* out, void* stream) {
  S2S_REQUIRE(B > 0 && Tmax > 0 && nb > 0 && nmel > 0 && z && frames && melb && lo && hi && out, "mel_log_batch: bad args");
  hipLaunchKernelGGL(mel_log_batch_kernel, dim3(Tmax, B), dim3(128), nb * sizeof(float), (hipStream_t)stream, Tmax, nb, nmel, z, frames,
                     melb, lo, hi, eps, inv_log_base, mean, inv_scale, out);
  S2S_CHECK_LAUNCH("mel_log_batch_kernel");
  return 0;
}

extern "C" int s2svc_ragged_to_padded(int B, int Tmax, int D, const float* ragged, const int64_t* offsets, const float* mean,
                                      const float* inv_scale, float* out, float* labels, void* stream) {
  S2S_REQUIRE(B > 0 && Tmax > 0 && D > 0 && ragged && offsets && out, "ragged_to_padded: bad args");
  hipLaunchKernelGGL(ragged_to_padded_kernel, dim3(ew_blocks((int64_t)B * Tmax * D)), dim3(256), 0, (hipStream_t)stream, B, Tmax, D, ragged,
                     offsets, mean, inv_scale, out, labels);
  S2S_CHECK_LAUNCH("ragged_to_padded_kernel");
  return 0;
}
