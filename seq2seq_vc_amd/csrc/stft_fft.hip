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

// ---------------------------------------------------------------------------------------------------------
// n_fft = 1024, radix-8 (round 3, second version): the 512-point complex FFT as THREE passes of in-register 8-point DFTs
// (512 = 8^3: a lane owns one butterfly per pass), so a frame makes two LDS round trips for the FFT instead of five, the first
// pass reads nothing (its inputs are the lane's own 8 loaded points) and the last one writes nothing (its outputs Z[lane + 64 r]
// stay in registers).  Every table a lane needs is the same for every frame it processes -- the window values of its points, the
// twiddles of passes 2 and 3 (k = lane & 7 resp. lane), the unpack twiddles of its 8 bins -- and lives in registers (60), loaded
// once per kernel.  Unpack: the partner Z[(512 - k) mod 512] of bin k = lane + 64 r sits in lane (64 - lane) mod 64, register
// 7 - r: one exchange through LDS.  Mel projection by SEGMENTS: with triangular filters a bin has at most two non-zero weights, in
// neighbouring filters; the bins between the peaks of filters s - 1 and s form segment s (n_mels + 1 segments), lane s walks its
// range once and forms up[s] = sum w[s][k] mag[k] and dn[s] = sum w[s-1][k] mag[k]; mel[m] = up[m] + dn[m+1].  Trip counts are
// half the filter widths (and the host checks the structure: anything else takes the generic kernel above).
// LDS per workgroup: 8 waves x 2 buffers x 576 points (one pad point group per 64: pass 2 stores with a stride of 64 points) +
// 4.1 KB of bin weights = 78 KB, two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------
struct fft8_args {
  int B, Tmax, hop, nmel;
  int64_t Nmax;
  const float* x;
  const int32_t* nlen;
  const int32_t* frames;
  const float* tables;         // the packed table of the generic kernel: w_half [512] complex | w_full [513] complex | win [1024] | ...
  const int32_t* seg_lo;       // [nmel + 1] first bin of segment s: the bins between the peaks of filters s - 1 and s
  const int32_t* seg_len;      // [nmel + 1]
  const float* wud;            // [513][2]: weight of filter s(k) (rising side) and of filter s(k) - 1 (falling side) at bin k
  float eps, inv_log_base;
  const float* mean;
  const float* inv_scale;
  float* out;
};

__device__ __forceinline__ int pad8(int d) { return d + ((d >> 6) << 3); }

// y[m] = sum_r u[r] exp(-2 pi i r m / 8), in place
__device__ __forceinline__ void dft8(c32 (&u)[8]) {
  const float h = 0.70710678118654752f;
  const c32 a0 = cadd(u[0], u[4]), a1 = csub(u[0], u[4]), a2 = cadd(u[2], u[6]), a3 = mul_mi(csub(u[2], u[6]));
  const c32 b0 = cadd(u[1], u[5]), b1 = csub(u[1], u[5]), b2 = cadd(u[3], u[7]), b3 = mul_mi(csub(u[3], u[7]));
  const c32 e0 = cadd(a0, a2), e2 = csub(a0, a2), e1 = cadd(a1, a3), e3 = csub(a1, a3);
  const c32 o0 = cadd(b0, b2), o2 = mul_mi(csub(b0, b2));
  c32 o1 = cadd(b1, b3), o3 = csub(b1, b3);
  o1 = {h * (o1.x + o1.y), h * (o1.y - o1.x)};           // * (1 - i) / sqrt 2
  o3 = {h * (o3.y - o3.x), -h * (o3.x + o3.y)};          // * (-1 - i) / sqrt 2
  u[0] = cadd(e0, o0); u[4] = csub(e0, o0);
  u[1] = cadd(e1, o1); u[5] = csub(e1, o1);
  u[2] = cadd(e2, o2); u[6] = csub(e2, o2);
  u[3] = cadd(e3, o3); u[7] = csub(e3, o3);
}

// 4 frames per workgroup: ~170 VGPRs allow three waves per SIMD, i.e. three of these workgroups (41 KB of LDS each) per CU
constexpr int WAVES8 = 4;

__global__ __launch_bounds__(64 * WAVES8) __attribute__((amdgpu_waves_per_eu(3, 3))) void stft_logmel_fft8_kernel(fft8_args a) {
  constexpr int N = 1024, H = 512, BUF = 576;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2* wud = reinterpret_cast<float2*>(smem_raw);                          // [H + 1] (+ pad to 520)
  const int nseg = a.nmel + 1;
  int32_t* seg_lo = reinterpret_cast<int32_t*>(wud + 520);                    // [nmel + 1]
  int32_t* seg_len = seg_lo + nseg;
  float* updn = reinterpret_cast<float*>(seg_len + nseg);                    // [WAVES8][2][nmel + 1]
  const int updn_n = (2 * (a.nmel + 1) + 3) & ~3;
  c32* bufs = reinterpret_cast<c32*>(updn + WAVES8 * updn_n);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  c32* bx = bufs + wave * 2 * BUF;
  c32* by = bx + BUF;
  float* up = updn + wave * updn_n;
  float* dn = up + a.nmel + 1;
  for (int i = threadIdx.x; i < H + 1; i += 64 * WAVES8) wud[i] = reinterpret_cast<const float2*>(a.wud)[i];
  for (int i = threadIdx.x; i < nseg; i += 64 * WAVES8) { seg_lo[i] = a.seg_lo[i]; seg_len[i] = a.seg_len[i]; }
  // per-lane constants
  const c32* w_half = reinterpret_cast<const c32*>(a.tables);
  const c32* w_full = w_half + H;
  const float* wing = reinterpret_cast<const float*>(w_full + H + 1);
  float2 wn[8];
  c32 tw2[7], tw3[7], tf[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    wn[q] = *reinterpret_cast<const float2*>(wing + 2 * (lane + 64 * q));
    tf[q] = w_full[lane + 64 * q];
  }
#pragma unroll
  for (int r = 1; r < 8; ++r) {
    tw2[r - 1] = w_half[((lane & 7) * r * 8) & (H - 1)];
    tw3[r - 1] = w_half[(lane * r) & (H - 1)];
  }
  __syncthreads();
  // segment rounds: trip count of a round = its longest segment (wave-uniform), found once
  int seg_l[3] = {0, 0, 0}, seg_n[3] = {0, 0, 0}, maxlen[3] = {0, 0, 0};
#pragma unroll
  for (int rd = 0; rd < 3; ++rd) {
    const int m = rd * 64 + lane;
    if (m < nseg) { seg_l[rd] = seg_lo[m]; seg_n[rd] = seg_len[m]; }
    int mx = seg_n[rd];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
    maxlen[rd] = mx;
  }
  const int gpb = (a.Tmax + WAVES8 - 1) / WAVES8, total = a.B * gpb;
  // raw samples (before the window) of this wave's frame of group g: two per lane and radix-8 leg; false = nothing to transform
  auto load = [&](int g, float2 (&xv)[8]) -> bool {
    if (g >= total) return false;
    const int b = g / gpb, t = (g - b * gpb) * WAVES8 + wave;
    if (t >= a.Tmax || t >= a.frames[b]) return false;
    const float* xb = a.x + (int64_t)b * a.Nmax;
    const int64_t n = a.nlen[b];
    const int64_t s0 = (int64_t)t * a.hop - H;
    const bool inner = s0 >= 0 && s0 + N <= n;
    if (inner && ((reinterpret_cast<uintptr_t>(xb + s0) & 7) == 0)) {
#pragma unroll
      for (int q = 0; q < 8; ++q) xv[q] = *reinterpret_cast<const float2*>(xb + s0 + 2 * (lane + 64 * q));
    } else {
      const int64_t period = n > 1 ? 2 * (n - 1) : 1;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          int64_t j = s0 + 2 * (lane + 64 * q) + e;
          if (!inner) {
            if (n > 1) {
              j %= period;
              if (j < 0) j += period;
              if (j >= n) j = period - j;
            } else {
              j = 0;
            }
          }
          v[e] = xb[j];
        }
        xv[q] = make_float2(v[0], v[1]);
      }
    }
    return true;
  };
  float2 cur[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) cur[q] = make_float2(0.f, 0.f);
  bool have = load(blockIdx.x, cur);
#pragma unroll 1
  for (int g = blockIdx.x; g < total; g += gridDim.x) {
    // the next frame's samples travel while this one is transformed (a frame used to start with an exposed trip to L2 / HBM)
    float2 nxt[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) nxt[q] = make_float2(0.f, 0.f);
    const bool have_next = load(g + gridDim.x, nxt);
    const int b = g / gpb, t = (g - b * gpb) * WAVES8 + wave;
    float* orow = a.out + ((int64_t)b * a.Tmax + t) * a.nmel;
    if (!have) {
      if (t < a.Tmax)
        for (int m = lane; m < a.nmel; m += 64) orow[m] = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) cur[q] = nxt[q];
      have = have_next;
      continue;
    }
    c32 u[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) u[q] = {cur[q].x * wn[q].x, cur[q].y * wn[q].y};
#pragma unroll
    for (int q = 0; q < 8; ++q) cur[q] = nxt[q];
    have = have_next;
    // ---- pass 1 (p = 1): butterfly i = lane on the loaded points (i + 64 r); outputs dst[8 i + r]
    dft8(u);
    {
      c32* d = bx + pad8(8 * lane);
#pragma unroll
      for (int r = 0; r < 8; r += 2) *reinterpret_cast<float4*>(d + r) = make_float4(u[r].x, u[r].y, u[r + 1].x, u[r + 1].y);
    }
    __builtin_amdgcn_wave_barrier();
    // ---- pass 2 (p = 8): inputs src[lane + 64 r] * w^(k r), k = lane & 7; outputs dst[(lane - k) * 8 + k + 8 r]
#pragma unroll
    for (int r = 0; r < 8; ++r) u[r] = bx[pad8(lane + 64 * r)];
#pragma unroll
    for (int r = 1; r < 8; ++r) u[r] = cmul(u[r], tw2[r - 1]);
    dft8(u);
    {
      const int j = ((lane >> 3) << 6) + (lane & 7);
#pragma unroll
      for (int r = 0; r < 8; ++r) by[pad8(j + 8 * r)] = u[r];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- pass 3 (p = 64): inputs src[lane + 64 r] * w^(lane r); outputs Z[lane + 64 r] in registers
#pragma unroll
    for (int r = 0; r < 8; ++r) u[r] = by[pad8(lane + 64 * r)];
#pragma unroll
    for (int r = 1; r < 8; ++r) u[r] = cmul(u[r], tw3[r - 1]);
    dft8(u);
    // ---- exchange: Z[(512 - k) mod 512] for k = lane + 64 r
#pragma unroll
    for (int r = 0; r < 8; ++r) bx[pad8(lane + 64 * r)] = u[r];
    __builtin_amdgcn_wave_barrier();
    float* mag = reinterpret_cast<float*>(by);               // [H + 1]
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = lane + 64 * r;
      c32 zc = bx[pad8((H - k) & (H - 1))];
      zc.y = -zc.y;
      const c32 zk = u[r];
      const c32 xe = {0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y)};
      const c32 xo = mul_mi({0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y)});
      const c32 wx = cmul(tf[r], xo);
      const c32 a0 = cadd(xe, wx);
      mag[k] = sqrtf(a0.x * a0.x + a0.y * a0.y);
      if (k == 0) mag[H] = fabsf(zk.x - zk.y);              // X[N/2] = Re Z[0] - Im Z[0]
    }
    __builtin_amdgcn_wave_barrier();
    // ---- mel segments
#pragma unroll
    for (int rd = 0; rd < 3; ++rd) {
      if (rd * 64 >= nseg) break;
      float su = 0.f, sd = 0.f;
      for (int it = 0; it < maxlen[rd]; ++it) {
        const int k = seg_l[rd] + (it < seg_n[rd] ? it : 0);
        const float2 w = wud[k];
        const float mg = it < seg_n[rd] ? mag[k] : 0.f;
        su += w.x * mg;
        sd += w.y * mg;
      }
      const int m = rd * 64 + lane;
      if (m < nseg) { up[m] = su; dn[m] = sd; }
    }
    __builtin_amdgcn_wave_barrier();
    for (int m = lane; m < a.nmel; m += 64) {
      float v = logf(fmaxf(a.eps, up[m] + dn[m + 1])) * a.inv_log_base;
      if (a.mean) v = (v - a.mean[m]) * a.inv_scale[m];
      orow[m] = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
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

// n_fft = 1024 with triangular (two-filters-per-bin) mel weights: the radix-8 kernel.  seg_lo / seg_len (nmel) and wud (513 x 2)
// come from the host (frontend._fft8_tables), `tables` is the packed table of s2svc_stft_logmel_fft.
extern "C" int s2svc_stft_logmel_fft8(int B, int64_t Nmax, int Tmax, int hop, int nmel, const float* x, const int32_t* nlen,
                                      const int32_t* frames, const float* tables, const int32_t* seg_lo, const int32_t* seg_len,
                                      const float* wud, float eps, float inv_log_base, const float* mean, const float* inv_scale,
                                      float* out, void* stream) {
  S2S_REQUIRE(B > 0 && Nmax > 0 && Tmax > 0 && hop > 0 && nmel >= 1 && nmel <= 128 && x && nlen && frames && tables && seg_lo && seg_len && wud && out &&
              ((uintptr_t)tables) % 16 == 0 && ((uintptr_t)wud) % 8 == 0, "stft_logmel_fft8: bad args (n_mels <= 128)");
  fft8_args a;
  a.B = B; a.Tmax = Tmax; a.hop = hop; a.nmel = nmel; a.Nmax = Nmax; a.x = x; a.nlen = nlen; a.frames = frames; a.tables = tables;
  a.seg_lo = seg_lo; a.seg_len = seg_len; a.wud = wud; a.eps = eps; a.inv_log_base = inv_log_base; a.mean = mean; a.inv_scale = inv_scale;
  a.out = out;
  const int updn_n = (2 * (nmel + 1) + 3) & ~3;
  const size_t lds = 520 * 8 + (size_t)(nmel + 1) * 8 + (size_t)WAVES8 * updn_n * 4 + (size_t)WAVES8 * 2 * 576 * 8;
  static size_t attr_set = 0;
  if (lds > 64 * 1024 && attr_set < lds) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(stft_logmel_fft8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) { s2svc_set_error("stft_logmel_fft8: cannot raise the dynamic LDS limit"); return -2; }
    attr_set = lds;
  }
  const int groups = B * ((Tmax + WAVES8 - 1) / WAVES8);
  const int per_cu = (int)(160 * 1024 / lds) > 0 ? (int)(160 * 1024 / lds) : 1;
  const int resident = 256 * (per_cu < 3 ? per_cu : 3);
  hipLaunchKernelGGL(stft_logmel_fft8_kernel, dim3(groups < resident ? groups : resident), dim3(64 * WAVES8), lds, (hipStream_t)stream, a);
  S2S_CHECK_LAUNCH("stft_logmel_fft8_kernel");
  return 0;
}
