// STFT -> log-mel in ONE launch with the FFT in LDS: wav batch (B, Nmax) -> normalised, zero-padded (B, Tmax, n_mels) log-mel.
//
// reference: bin/preprocess.py:30-92 (librosa.stft(center=True, pad_mode="reflect", window="hann") -> abs -> mel basis ->
// max(eps, .) -> log10), bin/normalize.py:172-193 ((x - mean) / scale), the collater's zero padding.
//
// One wavefront per frame.  A real frame of N samples is the complex sequence z[n] = x[2n] + i x[2n+1] of N/2 points
// (windowed while loading; the reflect padding of librosa's centre mode is an index computation on the raw waveform, no
// padded copy exists), transformed by a radix-4 Stockham autosort FFT between two wave-private LDS buffers (a lane owns
// N/128 butterflies of every stage; LDS operations of one wavefront execute in order, so no barrier is needed between the
// stages), unpacked to the one-sided spectrum X[0 .. N/2], reduced to magnitudes in LDS, projected on the mel filters (a lane
// owns a filter and walks only its non-zero bins: the compact weight list, ~2 N/2 values, sits in LDS), clamped, logged,
// normalised and stored as one coalesced row of n_mels floats.  HBM traffic per frame = hop samples in (the overlap is served
// by the L2) + n_mels floats out: the algorithmic minimum (4 B / sample + 320 B / frame); the DFT-as-GEMM formulation it
// replaces (frontend.hip) did 40x the flops and wrote / re-read a (frames, N + 2) fp32 spectrum.
// Twiddle factors, window and mel weights are tables built on the host in float64 and rounded once.
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

struct c32 { float x, y; };
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ c32 mul_mi(c32 a) { return {a.y, -a.x}; }          // a * (-i)
// (padding the FFT buffers against the bank conflicts of the first passes' strided stores -- one element per 32 -- was measured:
// 80 -> 82.5 KB of LDS per workgroup drops the occupancy from two workgroups per CU to one and costs more than it saves)

struct fft_args {
  int B, Tmax, hop, nmel;
  int64_t Nmax;
  const float* x;              // (B, Nmax) raw waveforms
  const int32_t* nlen;         // (B) samples
  const int32_t* frames;       // (B) frames = 1 + n // hop
  const float* tables;         // packed fp32, 16-byte aligned, zero-filled to whole 16-byte vectors:
                               //   w_half [N/2] complex exp(-2 pi i m / (N/2)) | w_full [N/2 + 1] complex exp(-2 pi i k / N) |
                               //   win [N] (zero-padded window) | melw [melw_n rounded up to even] (compact mel weights: filter m =
                               //   melw[off[m] .. off[m] + hi[m] - lo[m]))
  const int32_t* mel_lo;
  const int32_t* mel_hi;
  const int32_t* mel_off;
  int melw_n, mel_maxw;        // melw_n includes a zero tail of >= mel_maxw + 1 values
  float eps, inv_log_base;
  const float* mean;
  const float* inv_scale;
  float* out;                  // (B, Tmax, nmel)
};

constexpr int WAVES = 8;       // frames per workgroup

// LOG2N = log2(fft size): 9, 10, 11
template <int LOG2N>
__global__ __launch_bounds__(64 * WAVES) void stft_logmel_fft_kernel(fft_args a) {
  constexpr int N = 1 << LOG2N, H = N / 2, PPL = H / 64;      // points of the complex FFT, points per lane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  c32* tw = reinterpret_cast<c32*>(smem_raw);                 // [H] twiddles of the H-point FFT
  c32* tf = tw + H;                                           // [H + 1] unpack twiddles
  float* win = reinterpret_cast<float*>(tf + H + 1);          // [N]
  float* mw = win + N;                                        // [melw_n]
  const int tab_vecs = (int)((sizeof(c32) * (2 * H + 1) + sizeof(float) * (N + ((a.melw_n + 1) & ~1)) + 15) / 16);
  c32* bufs = reinterpret_cast<c32*>(smem_raw + (size_t)tab_vecs * 16);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  c32* bx = bufs + wave * 2 * H;
  c32* by = bx + H;
  // the four tables are ONE packed array in global memory, laid out as in LDS: every thread issues all its 16-byte loads before
  // its first store (one memory round trip instead of one per table), and a workgroup keeps them for all its frame groups
  {
    const uint4* tg = reinterpret_cast<const uint4*>(a.tables);
    uint4* tl = reinterpret_cast<uint4*>(smem_raw);
    constexpr int NV = 4;
    for (int i0 = 0; i0 < tab_vecs; i0 += NV * 64 * WAVES) {
      uint4 r[NV];
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int i = i0 + u * 64 * WAVES + threadIdx.x;
        r[u] = i < tab_vecs ? tg[i] : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < NV; ++u) {
        const int i = i0 + u * 64 * WAVES + threadIdx.x;
        if (i < tab_vecs) tl[i] = r[u];
      }
    }
  }
  __syncthreads();
  const int gpb = (a.Tmax + WAVES - 1) / WAVES;               // frame groups per utterance
#pragma unroll 1
  for (int g = blockIdx.x; g < a.B * gpb; g += gridDim.x) {   // persistent: no workgroup-level synchronisation inside
  const int b = g / gpb, t = (g - b * gpb) * WAVES + wave;
  if (t >= a.Tmax) continue;
  float* orow = a.out + ((int64_t)b * a.Tmax + t) * a.nmel;
  if (t >= a.frames[b]) {                                     // padding frame of the batch
    for (int m = lane; m < a.nmel; m += 64) orow[m] = 0.f;
    continue;
  }
  // ---- frame t: samples [t * hop - N/2, t * hop + N/2) of the reflect-padded utterance, windowed, even / odd packed ----
  {
    const float* xb = a.x + (int64_t)b * a.Nmax;
    const int64_t n = a.nlen[b];
    const int64_t s0 = (int64_t)t * a.hop - H;
    const bool inner = s0 >= 0 && s0 + N <= n;               // uniform: no reflection needed
    const int64_t period = n > 1 ? 2 * (n - 1) : 1;
    if (inner && ((reinterpret_cast<uintptr_t>(xb + s0) & 7) == 0)) {      // interior frame, 8-byte aligned: one load per point
#pragma unroll
      for (int q = 0; q < PPL; ++q) {
        const int p = lane + 64 * q;
        const float2 xv = *reinterpret_cast<const float2*>(xb + s0 + 2 * p);
        const float2 wv = *reinterpret_cast<const float2*>(win + 2 * p);
        bx[p] = {xv.x * wv.x, xv.y * wv.y};
      }
    } else
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
      const int p = lane + 64 * q;                            // complex point: samples 2p, 2p + 1 of the frame
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        int64_t j = s0 + 2 * p + e;
        if (!inner) {
          if (n > 1) {
            j %= period;
            if (j < 0) j += period;
            if (j >= n) j = period - j;
          } else {
            j = 0;
          }
        }
        v[e] = xb[j] * win[2 * p + e];
      }
      bx[p] = {v[0], v[1]};
    }
  }
  // ---- H-point complex FFT, Stockham autosort: radix-4 passes (p = 1, 4, 16, ...) and one radix-2 pass if log2(H) is odd ----
  c32* src = bx;
  c32* dst = by;
  int p = 1;
#pragma unroll 1
  for (; p * 4 <= H; p *= 4) {
    __builtin_amdgcn_wave_barrier();
    constexpr int T = H / 4;                                  // butterflies per pass
#pragma unroll
    for (int q = 0; q < (T + 63) / 64; ++q) {
      const int i = lane + 64 * q;
      if (T % 64 == 0 || i < T) {
        const int k = i & (p - 1);
        const int j = ((i - k) << 2) + k;
        // twiddles exp(-2 pi i k r / (4 p)), r = 1, 2, 3 = table entries k r (H / (4 p))
        const int st = (H / 4) / p;
        const c32 u0 = src[i];
        c32 u1 = src[i + T], u2 = src[i + 2 * T], u3 = src[i + 3 * T];
        if (p > 1) {                                         // (uniform) the first pass has k = 0: all twiddles are 1
          u1 = cmul(u1, tw[k * st]);
          u2 = cmul(u2, tw[2 * k * st]);
          u3 = cmul(u3, tw[3 * k * st]);
        }
        const c32 v0 = cadd(u0, u2), v1 = csub(u0, u2), v2 = cadd(u1, u3), v3 = mul_mi(csub(u1, u3));
        dst[j] = cadd(v0, v2);
        dst[j + p] = cadd(v1, v3);
        dst[j + 2 * p] = csub(v0, v2);
        dst[j + 3 * p] = csub(v1, v3);
      }
    }
    c32* tmp = src; src = dst; dst = tmp;
  }
  if (p < H) {                                                // one radix-2 pass left (p = H / 2)
    __builtin_amdgcn_wave_barrier();
    constexpr int T = H / 2;
#pragma unroll
    for (int q = 0; q < T / 64; ++q) {
      const int i = lane + 64 * q;
      const int k = i & (p - 1);
      const int j = ((i - k) << 1) + k;
      const c32 u0 = src[i], u1 = cmul(src[i + T], tw[k * ((H / 2) / p)]);
      dst[j] = cadd(u0, u1);
      dst[j + p] = csub(u0, u1);
    }
    c32* tmp = src; src = dst; dst = tmp;
  }
  __builtin_amdgcn_wave_barrier();
  // ---- unpack to the one-sided spectrum of the real frame, magnitudes into the free buffer ----
  float* mag = reinterpret_cast<float*>(dst);                 // [H + 1] <= 2 H floats
  // X[k] = xe + w^k xo and X[H - k] = conj(xe - w^k xo) with xe = (Z[k] + conj Z[H-k]) / 2, xo = -i (Z[k] - conj Z[H-k]) / 2:
  // one twiddle product gives the magnitudes of two bins
  for (int k = lane; k <= H / 2; k += 64) {
    const c32 zk = src[k];
    c32 zc = src[(H - k) & (H - 1)];
    zc.y = -zc.y;
    const c32 xe = {0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y)};
    const c32 xo = mul_mi({0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y)});
    const c32 wx = cmul(tf[k], xo);
    const c32 a0 = cadd(xe, wx), a1 = csub(xe, wx);
    mag[k] = sqrtf(a0.x * a0.x + a0.y * a0.y);
    mag[H - k] = sqrtf(a1.x * a1.x + a1.y * a1.y);
  }
  __builtin_amdgcn_wave_barrier();
  // ---- mel projection over each filter's bins, clamp, log, normalisation ----
  for (int m0 = 0; m0 < a.nmel; m0 += 64) {
    const int m = m0 + lane;
    const bool have = m < a.nmel;
    const int lo = have ? a.mel_lo[m] : 0, hi = have ? a.mel_hi[m] : 0;
    const float* w = mw + (have ? a.mel_off[m] : 0);
    float acc0 = 0.f, acc1 = 0.f;
    // wave-uniform trip count (the widest filter of the table): no divergent loop control; lanes past their filter add zeros
    // (weights read past a filter's end belong to the next one or to the zero tail of the list, bins are clamped to H)
    for (int it = 0; it < a.mel_maxw; it += 2) {
      const int k0 = lo + it, k1 = k0 + 1;
      const float w0 = k0 < hi ? w[it] : 0.f, w1 = k1 < hi ? w[it + 1] : 0.f;
      acc0 += w0 * mag[k0 < H ? k0 : H];
      acc1 += w1 * mag[k1 < H ? k1 : H];
    }
    if (have) {
      float v = logf(fmaxf(a.eps, acc0 + acc1)) * a.inv_log_base;
      if (a.mean) v = (v - a.mean[m]) * a.inv_scale[m];
      orow[m] = v;
    }
  }
  __builtin_amdgcn_wave_barrier();                            // the next frame of this wave reuses its two buffers
  }
}

template <int LOG2N>
int launch(const fft_args& a, hipStream_t st) {
  constexpr int N = 1 << LOG2N, H = N / 2;
  const size_t tab = (sizeof(c32) * (2 * H + 1) + sizeof(float) * (N + ((a.melw_n + 1) & ~1)) + 15) / 16 * 16;
  const size_t lds = tab + sizeof(c32) * WAVES * 2 * H;
  static size_t attr_set = 0;
  if (lds > 64 * 1024 && attr_set < lds) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(stft_logmel_fft_kernel<LOG2N>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) { s2svc_set_error("stft_logmel_fft: cannot raise the dynamic LDS limit"); return -2; }
    attr_set = lds;
  }
  const int groups = a.B * ((a.Tmax + WAVES - 1) / WAVES);
  const int per_cu = (int)(160 * 1024 / lds) > 0 ? (int)(160 * 1024 / lds) : 1;
  const int resident = 256 * (per_cu < 4 ? per_cu : 4);       // workgroups the chip holds at once
  hipLaunchKernelGGL(stft_logmel_fft_kernel<LOG2N>, dim3(groups < resident ? groups : resident), dim3(64 * WAVES), lds, st, a);
  S2S_CHECK_LAUNCH("stft_logmel_fft_kernel");
  return 0;
}

}  // namespace

extern "C" int s2svc_stft_logmel_fft_supported(int n_fft, int nmel, int melw_n) {
  return (n_fft == 512 || n_fft == 1024 || n_fft == 2048) && nmel >= 1 && nmel <= 4096 && melw_n >= 0 && melw_n <= 8192;
}

extern "C" int s2svc_stft_logmel_fft(int B, int64_t Nmax, int Tmax, int n_fft, int hop, int nmel, const float* x, const int32_t* nlen,
                                     const int32_t* frames, const float* tables, const int32_t* mel_lo, const int32_t* mel_hi, const int32_t* mel_off, int melw_n,
                                     int mel_maxw, float eps, float inv_log_base, const float* mean, const float* inv_scale, float* out, void* stream) {
  S2S_REQUIRE(s2svc_stft_logmel_fft_supported(n_fft, nmel, melw_n), "stft_logmel_fft: n_fft must be 512 / 1024 / 2048");
  S2S_REQUIRE(mel_maxw >= 0 && mel_maxw < melw_n, "stft_logmel_fft: the weight list needs a zero tail of mel_maxw + 1 values");
  S2S_REQUIRE(B > 0 && Nmax > 0 && Tmax > 0 && hop > 0 && x && nlen && frames && tables && ((uintptr_t)tables) % 16 == 0 && mel_lo && mel_hi &&
              mel_off && out, "stft_logmel_fft: bad args");
  fft_args a;
  a.B = B; a.Tmax = Tmax; a.hop = hop; a.nmel = nmel; a.Nmax = Nmax; a.x = x; a.nlen = nlen; a.frames = frames; a.tables = tables;
  a.mel_lo = mel_lo; a.mel_hi = mel_hi; a.mel_off = mel_off; a.melw_n = melw_n; a.mel_maxw = mel_maxw;
  a.eps = eps; a.inv_log_base = inv_log_base; a.mean = mean; a.inv_scale = inv_scale; a.out = out;
  hipStream_t st = (hipStream_t)stream;
  if (n_fft == 512) return launch<9>(a, st);
  if (n_fft == 1024) return launch<10>(a, st);
  return launch<11>(a, st);
}
