// Scalar and index glue of a training step: the handful of tiny operations between the fused kernels (loss composition, logging sums,
// seed bump, zero fills, teacher-forcing shift, stop labels) that used to be ATen calls -- each a dependent launch of a generic kernel
// inside the captured step (76 per AAS-VC step, profiles/r05_aasvc_train_bf16_timeline.txt).  One small kernel per job here.
// reference lines these stand in for: trainers/aas_vc.py:100-139 (loss sums), models/vtn.py:236-260 (shifted decoder input, stop
// labels), losses/forward_sum_loss.py:70-76 and modules/alignments.py:303-309 (means over the batch).
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

// 16-byte stores over the aligned middle, bytes at the ragged ends
__global__ __launch_bounds__(256) void fill_zero_kernel(char* __restrict__ p, int64_t nbytes) {
  const uintptr_t a = (uintptr_t)p;
  const int64_t head = (int64_t)((16 - (a & 15)) & 15) < nbytes ? (int64_t)((16 - (a & 15)) & 15) : nbytes;
  const int64_t nvec = (nbytes - head) >> 4;
  uint4* v = reinterpret_cast<uint4*>(p + head);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) v[i] = make_uint4(0, 0, 0, 0);
  if (blockIdx.x == 0) {
    for (int64_t i = threadIdx.x; i < head; i += blockDim.x) p[i] = 0;
    const int64_t tail0 = head + (nvec << 4);
    for (int64_t i = tail0 + threadIdx.x; i < nbytes; i += blockDim.x) p[i] = 0;
  }
}

__global__ void seed_advance_kernel(uint64_t* seed, uint64_t inc) { *seed += inc; }

// one wavefront, fixed order: s_i = sum_j x_i[j] ; out = sum_i w_i s_i
__global__ __launch_bounds__(64) void weighted_sum_kernel(const s2svc_scalar_terms t, float* __restrict__ out) {
  float total = 0.f;
  for (int i = 0; i < t.k; ++i) {
    float acc = 0.f;
    for (int j = threadIdx.x; j < t.n[i]; j += 64) acc += t.x[i][j];
    total += t.w[i] * wave_sum(acc);
  }
  if (threadIdx.x == 0) *out = total;
}
// dx_i[j] = w_i * g
__global__ __launch_bounds__(64) void weighted_sum_bwd_kernel(const s2svc_scalar_terms t, const float* __restrict__ g) {
  const float gv = *g;
  for (int i = 0; i < t.k; ++i)
    for (int j = threadIdx.x; j < t.n[i]; j += 64) const_cast<float*>(t.x[i])[j] = t.w[i] * gv;
}
// acc[i] = beta * acc[i] + w_i * sum_j x_i[j]     (running sums of the logged losses)
__global__ __launch_bounds__(64) void scalars_axpy_kernel(const s2svc_scalar_terms t, float beta, float* __restrict__ acc) {
  for (int i = 0; i < t.k; ++i) {
    float a = 0.f;
    for (int j = threadIdx.x; j < t.n[i]; j += 64) a += t.x[i][j];
    a = wave_sum(a);
    if (threadIdx.x == 0) acc[i] = (beta != 0.f ? beta * acc[i] : 0.f) + t.w[i] * a;
  }
}

// out (rows, ldo) = [in (rows, N) | zeros]
template <typename T>
__global__ void pad_cols_kernel(int64_t n, int N, int ldo, const T* __restrict__ in, T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ldo;
    const int c = (int)(i - r * ldo);
    stf(out + i, c < N ? ldf(in + r * N + c) : 0.f);
  }
}

// teacher forcing: out[b, t, :] = t == 0 ? 0 : ys[b, t * r - 1, :]   (= cat(zeros, ys[:, r-1::r][:, :-1]), vtn.py:236-243), cast on the way
template <typename T>
__global__ void decoder_input_kernel(int64_t n, int Tin, int r, int D, int64_t bstride, const float* __restrict__ ys, T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const int64_t bt = i / D;
    const int t = (int)(bt % Tin);
    const int64_t b = bt / Tin;
    stf(out + i, t == 0 ? 0.f : ys[b * bstride + (int64_t)(t * r - 1) * D + d]);
  }
}

// out = labels with a 1 at frame lens[b] - 1 (vtn.py:253-260: torch.scatter(labels, 1, (olens - 1).unsqueeze(1), 1.0))
__global__ void stop_labels_kernel(int B, int T, int64_t ldl, const float* __restrict__ labels, const int32_t* __restrict__ lens,
                                   float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  const int b = i / T, t = i - b * T;
  out[i] = (t == lens[b] - 1) ? 1.f : labels[(int64_t)b * ldl + t];
}

// text batch with <eos> behind every sequence (models/transformer_tts.py:139-142: F.pad(xs, [0, 1], value = pad) + in-place eos):
// out (B, T + 1) = [xs[b, :len] | eos | pad ...]
__global__ void append_eos_kernel(int B, int T, int64_t ldx, const int64_t* __restrict__ xs, const int32_t* __restrict__ lens, int64_t eos,
                                  int64_t pad, int64_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * (T + 1)) return;
  const int b = i / (T + 1), t = i - b * (T + 1);
  out[i] = t == lens[b] ? eos : (t < T ? xs[(int64_t)b * ldx + t] : pad);
}

// out (rows, D) = src (rows, D) with row stride lds (a column block of a wider matrix made dense)
template <typename T>
__global__ void copy_rows_kernel(int64_t n, int D, int64_t lds, const T* __restrict__ src, T* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / D;
    out[i] = src[r * lds + (i - r * D)];
  }
}

inline int gl_blocks(int64_t n) { int64_t b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b)); }

bool terms_ok(const s2svc_scalar_terms* t) {
  if (!t || t->k < 1 || t->k > S2SVC_SCALAR_TERMS_MAX) return false;
  for (int i = 0; i < t->k; ++i)
    if (!t->x[i] || t->n[i] < 0) return false;
  return true;
}

}  // namespace

extern "C" int s2svc_fill_zero(void* p, int64_t nbytes, void* stream) {
  if (nbytes == 0) return 0;
  S2S_REQUIRE(p && nbytes > 0, "fill_zero: bad args");
  hipLaunchKernelGGL(fill_zero_kernel, dim3(gl_blocks(nbytes >> 4)), dim3(256), 0, (hipStream_t)stream, (char*)p, nbytes);
  S2S_CHECK_LAUNCH("fill_zero_kernel");
  return 0;
}

extern "C" int s2svc_seed_advance(uint64_t* seed, uint64_t inc, void* stream) {
  S2S_REQUIRE(seed, "seed_advance: bad args");
  hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, seed, inc);
  S2S_CHECK_LAUNCH("seed_advance_kernel");
  return 0;
}

extern "C" int s2svc_weighted_sum(const s2svc_scalar_terms* terms, float* out, void* stream) {
  S2S_REQUIRE(terms_ok(terms) && out, "weighted_sum: 1..8 terms, every pointer set");
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *terms, out);
  S2S_CHECK_LAUNCH("weighted_sum_kernel");
  return 0;
}

extern "C" int s2svc_weighted_sum_bwd(const s2svc_scalar_terms* grads, const float* g, void* stream) {
  S2S_REQUIRE(terms_ok(grads) && g, "weighted_sum_bwd: 1..8 terms, every pointer set");
  hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *grads, g);
  S2S_CHECK_LAUNCH("weighted_sum_bwd_kernel");
  return 0;
}

extern "C" int s2svc_scalars_axpy(const s2svc_scalar_terms* terms, float beta, float* acc, void* stream) {
  S2S_REQUIRE(terms_ok(terms) && acc, "scalars_axpy: 1..8 terms, every pointer set");
  hipLaunchKernelGGL(scalars_axpy_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *terms, beta, acc);
  S2S_CHECK_LAUNCH("scalars_axpy_kernel");
  return 0;
}

extern "C" int s2svc_pad_cols(int dtype, int64_t rows, int N, int ldo, const void* in, void* out, void* stream) {
  const int64_t n = rows * ldo;
  if (n == 0) return 0;
  S2S_REQUIRE(in && out && N > 0 && ldo >= N, "pad_cols: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32) hipLaunchKernelGGL(pad_cols_kernel<float>, dim3(gl_blocks(n)), dim3(256), 0, st, n, N, ldo, (const float*)in, (float*)out);
  else hipLaunchKernelGGL(pad_cols_kernel<bf16_t>, dim3(gl_blocks(n)), dim3(256), 0, st, n, N, ldo, (const bf16_t*)in, (bf16_t*)out);
  S2S_CHECK_LAUNCH("pad_cols_kernel");
  return 0;
}

extern "C" int s2svc_decoder_input(int out_dtype, int B, int Tin, int r, int D, int64_t ys_batch_stride, const float* ys, void* out,
                                   void* stream) {
  const int64_t n = (int64_t)B * Tin * D;
  if (n == 0) return 0;
  S2S_REQUIRE(ys && out && r >= 1 && ys_batch_stride >= (int64_t)((Tin - 1) * r) * D, "decoder_input: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == S2S_F32)
    hipLaunchKernelGGL(decoder_input_kernel<float>, dim3(gl_blocks(n)), dim3(256), 0, st, n, Tin, r, D, ys_batch_stride, ys, (float*)out);
  else
    hipLaunchKernelGGL(decoder_input_kernel<bf16_t>, dim3(gl_blocks(n)), dim3(256), 0, st, n, Tin, r, D, ys_batch_stride, ys, (bf16_t*)out);
  S2S_CHECK_LAUNCH("decoder_input_kernel");
  return 0;
}

extern "C" int s2svc_append_eos(int B, int T, int64_t ldx, const int64_t* xs, const int32_t* lens, int64_t eos, int64_t pad, int64_t* out,
                                void* stream) {
  if (B == 0) return 0;
  S2S_REQUIRE(xs && lens && out && ldx >= T, "append_eos: bad args");
  hipLaunchKernelGGL(append_eos_kernel, dim3((B * (T + 1) + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, T, ldx, xs, lens, eos, pad, out);
  S2S_CHECK_LAUNCH("append_eos_kernel");
  return 0;
}

extern "C" int s2svc_copy_rows(int dtype, int64_t rows, int D, int64_t lds, const void* src, void* out, void* stream) {
  const int64_t n = rows * D;
  if (n == 0) return 0;
  S2S_REQUIRE(src && out && lds >= D, "copy_rows: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32) hipLaunchKernelGGL(copy_rows_kernel<float>, dim3(gl_blocks(n)), dim3(256), 0, st, n, D, lds, (const float*)src, (float*)out);
  else hipLaunchKernelGGL(copy_rows_kernel<bf16_t>, dim3(gl_blocks(n)), dim3(256), 0, st, n, D, lds, (const bf16_t*)src, (bf16_t*)out);
  S2S_CHECK_LAUNCH("copy_rows_kernel");
  return 0;
}

extern "C" int s2svc_stop_labels(int B, int T, int64_t ld_labels, const float* labels, const int32_t* lens, float* out, void* stream) {
  if (B * T == 0) return 0;
  S2S_REQUIRE(labels && lens && out && ld_labels >= T, "stop_labels: bad args");
  hipLaunchKernelGGL(stop_labels_kernel, dim3((B * T + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, T, ld_labels, labels, lens, out);
  S2S_CHECK_LAUNCH("stop_labels_kernel");
  return 0;
}
