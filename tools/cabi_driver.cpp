// C++ driver for the extern "C" entry points of libs2svc_hip.so WITHOUT Python ("export plain extern "C" entry
// points for the MAS / CTC / STFT kernels so they can be driven from a C++ test binary"): device buffers from hipMalloc, the
// library's functions called exactly as include/s2svc_hip.h declares them, results checked against known answers computed here.
//
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/cabi_driver.cpp -o tools/cabi_driver.bin \
//         -Lseq2seq_vc_amd/csrc -ls2svc_hip -Wl,-rpath,'$ORIGIN/../seq2seq_vc_amd/csrc'
//   tools/cabi_driver.bin            # prints one line per check, exit code = number of failed checks
//
// Checks: (1) monotonic alignment search on SURVEY 8(c)'s known-answer vectors KAT1 / KAT2 (modules/alignments.py:63-93);
// (2) the forward-sum (CTC) loss of losses/forward_sum_loss.py:58-76 against a log-space alpha recursion in double precision;
// (3) the one-launch STFT -> log-mel kernel (bin/preprocess.py:30-92) on a bin-centred cosine against the closed form
//     |X[k0]| = A N / 4, |X[k0 +- 1]| = A N / 8 through a Slaney mel basis built here in double precision.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/s2svc_hip.h"

#define HIP_OK(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(99); } \
  } while (0)
#define LIB_OK(x)                                                                      \
  do {                                                                                 \
    int r_ = (x);                                                                      \
    if (r_ != 0) { std::printf("library error %d (%s) at %s:%d\n", r_, s2svc_last_error(), __FILE__, __LINE__); std::exit(98); } \
  } while (0)

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  explicit DevBuf(size_t n_) : n(n_) { HIP_OK(hipMalloc(&p, (n_ ? n_ : 1) * sizeof(T))); }
  explicit DevBuf(const std::vector<T>& h) : DevBuf(h.size()) { HIP_OK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
  ~DevBuf() { (void)hipFree(p); }
  std::vector<T> host() const {
    std::vector<T> h(n);
    HIP_OK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }
};

static int g_fail = 0;
static void report(bool ok, const char* what) {
  std::printf("%s  %s\n", ok ? "PASS" : "FAIL", what);
  if (!ok) ++g_fail;
}

// ---- (1) alignment search ------------------------------------------------------------------------------------------
static void check_mas() {
  const int Tf = 6, Tx = 3;
  const double p1[6][3] = {{.7, .2, .1}, {.6, .3, .1}, {.2, .6, .2}, {.1, .6, .3}, {.1, .2, .7}, {.05, .15, .8}};
  std::vector<float> lp(2 * Tf * Tx);
  for (int t = 0; t < Tf; ++t)
    for (int j = 0; j < Tx; ++j) {
      lp[t * Tx + j] = (float)std::log(p1[t][j]);               // KAT1
      lp[Tf * Tx + t * Tx + j] = (float)std::log(1.0 / 3.0);    // KAT2: all ties
    }
  DevBuf<float> d_lp(lp);
  DevBuf<int32_t> d_tl(std::vector<int32_t>{Tx, Tx}), d_fl(std::vector<int32_t>{Tf, Tf}), d_path((size_t)2 * Tf);
  DevBuf<float> d_ds((size_t)2 * Tx), d_bin(2);
  DevBuf<unsigned char> d_ws((size_t)s2svc_mas_ws_bytes(2, Tf, Tx) + 16);
  LIB_OK(s2svc_mas(2, Tf, Tx, d_lp.p, d_tl.p, d_fl.p, d_path.p, d_ds.p, d_bin.p, d_ws.p, nullptr));
  HIP_OK(hipDeviceSynchronize());
  const std::vector<int32_t> path = d_path.host();
  const std::vector<float> ds = d_ds.host();
  const int want1[6] = {0, 0, 1, 1, 2, 2}, want2[6] = {0, 0, 0, 0, 1, 2};
  bool ok1 = true, ok2 = true;
  for (int t = 0; t < Tf; ++t) { ok1 = ok1 && path[t] == want1[t]; ok2 = ok2 && path[Tf + t] == want2[t]; }
  report(ok1 && ds[0] == 2.f && ds[1] == 2.f && ds[2] == 2.f, "s2svc_mas: KAT1 path [0,0,1,1,2,2], durations [2,2,2]");
  report(ok2 && ds[3] == 4.f && ds[4] == 1.f && ds[5] == 1.f, "s2svc_mas: KAT2 (all ties) path [0,0,0,0,1,2], durations [4,1,1]");
}

// ---- (2) forward-sum loss ---------------------------------------------------------------------------------------------
static double logaddexp(double a, double b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  return a > b ? a + std::log1p(std::exp(b - a)) : b + std::log1p(std::exp(a - b));
}
// CTC negative log-likelihood of the target 1..N given log "probabilities" lp[t][c] (c = 0 blank), divided by N (reduction
// 'mean' of torch.nn.functional.ctc_loss on one utterance)
static double ctc_ref(const std::vector<double>& lp, int T, int N) {
  const int S = 2 * N + 1, C = N + 1;
  std::vector<double> a(S, -INFINITY), b(S);
  a[0] = lp[0];
  a[1] = lp[1];
  for (int t = 1; t < T; ++t) {
    for (int s = 0; s < S; ++s) {
      double v = a[s];
      if (s >= 1) v = logaddexp(v, a[s - 1]);
      if (s >= 2 && (s & 1)) v = logaddexp(v, a[s - 2]);        // distinct labels: the skip over a blank is always allowed
      const int c = (s & 1) ? (s + 1) / 2 : 0;
      b[s] = v + lp[(size_t)t * C + c];
    }
    a.swap(b);
  }
  return -logaddexp(a[S - 1], a[S - 2]) / N;
}

static void check_forward_sum() {
  const int B = 2, Tf = 12, Tx = 5;
  const int tl[2] = {5, 3}, fl[2] = {12, 9};
  std::vector<float> lp((size_t)B * Tf * Tx), prior((size_t)B * Tf * Tx, 0.f);
  unsigned s = 12345u;
  for (auto& v : lp) { s = s * 1664525u + 1013904223u; v = -0.2f - 3.0f * (float)((s >> 8) & 0xffff) / 65536.0f; }
  DevBuf<float> d_lp(lp), d_prior(prior), d_loss(B), d_grad((size_t)B * Tf * Tx);
  DevBuf<int32_t> d_tl(std::vector<int32_t>{tl[0], tl[1]}), d_fl(std::vector<int32_t>{fl[0], fl[1]});
  DevBuf<unsigned char> d_ws((size_t)s2svc_forward_sum_ws_bytes(B, Tf, Tx) + 16);
  LIB_OK(s2svc_forward_sum(B, Tf, Tx, d_lp.p, d_prior.p, d_tl.p, d_fl.p, -1.0f, d_ws.p, d_loss.p, d_grad.p, nullptr));
  HIP_OK(hipDeviceSynchronize());
  const std::vector<float> loss = d_loss.host();
  bool ok = true;
  char msg[256];
  double worst = 0;
  for (int b = 0; b < B; ++b) {
    const int T = fl[b], N = tl[b];
    std::vector<double> q((size_t)T * (N + 1));
    for (int t = 0; t < T; ++t) {
      q[(size_t)t * (N + 1)] = -1.0;                                     // the blank column: log(e^-1)
      for (int j = 0; j < N; ++j) q[(size_t)t * (N + 1) + 1 + j] = lp[((size_t)b * Tf + t) * Tx + j];
    }
    const double want = ctc_ref(q, T, N);
    worst = std::fmax(worst, std::fabs(want - loss[b]));
    ok = ok && std::fabs(want - loss[b]) < 1e-4 * std::fmax(1.0, std::fabs(want));
  }
  std::snprintf(msg, sizeof msg, "s2svc_forward_sum: per-utterance CTC loss vs a double-precision alpha recursion (max diff %.2e)", worst);
  report(ok, msg);
}

// ---- (3) STFT -> log-mel ---------------------------------------------------------------------------------------------
static double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m;
}

static void check_stft() {
  const int sr = 16000, N = 1024, hop = 256, nmel = 80, H = N / 2, nb = H + 1;
  const double fmin = 80, fmax = 7600, PI = 3.14159265358979323846;
  // Slaney mel basis, float64 (librosa.filters.mel defaults)
  std::vector<double> pts(nmel + 2), fb((size_t)nmel * nb, 0.0);
  for (int i = 0; i < nmel + 2; ++i) pts[i] = mel_to_hz(hz_to_mel(fmin) + (hz_to_mel(fmax) - hz_to_mel(fmin)) * i / (nmel + 1));
  for (int m = 0; m < nmel; ++m)
    for (int k = 0; k < nb; ++k) {
      const double f = (double)k * sr / N, up = (f - pts[m]) / (pts[m + 1] - pts[m]), down = (pts[m + 2] - f) / (pts[m + 2] - pts[m + 1]);
      const double w = std::fmin(up, down);
      fb[(size_t)m * nb + k] = w > 0 ? w * 2.0 / (pts[m + 2] - pts[m]) : 0.0;
    }
  std::vector<int32_t> lo(nmel), hi(nmel), off(nmel);
  std::vector<float> melw;
  int maxw = 0;
  for (int m = 0; m < nmel; ++m) {
    int a = 0, b = 0;
    for (int k = 0; k < nb; ++k) if (fb[(size_t)m * nb + k] > 0) { if (b == 0) a = k; b = k + 1; }
    lo[m] = a; hi[m] = b; off[m] = (int32_t)melw.size();
    for (int k = a; k < b; ++k) melw.push_back((float)fb[(size_t)m * nb + k]);
    if (b - a > maxw) maxw = b - a;
  }
  maxw += 1;
  melw.resize(melw.size() + maxw + 2, 0.f);
  const int melw_n = (int)melw.size();
  // packed tables: w_half | w_full | win | melw (+ zero fill)
  std::vector<float> tab;
  for (int m = 0; m < H; ++m) { tab.push_back((float)std::cos(-2 * PI * m / H)); tab.push_back((float)std::sin(-2 * PI * m / H)); }
  for (int k = 0; k <= H; ++k) { tab.push_back((float)std::cos(-2 * PI * k / N)); tab.push_back((float)std::sin(-2 * PI * k / N)); }
  for (int i = 0; i < N; ++i) tab.push_back((float)(0.5 - 0.5 * std::cos(2 * PI * i / N)));
  for (float v : melw) tab.push_back(v);
  if (melw_n % 2) tab.push_back(0.f);
  while (tab.size() % 4) tab.push_back(0.f);
  // a bin-centred cosine
  const int n = hop * 40, k0 = 100, frames = 1 + n / hop;
  const double A = 0.37;
  std::vector<float> x(n);
  for (int i = 0; i < n; ++i) x[i] = (float)(A * std::cos(2 * PI * k0 * i / N + 0.3));
  DevBuf<float> d_x(x), d_tab(tab), d_out((size_t)frames * nmel);
  DevBuf<int32_t> d_n(std::vector<int32_t>{n}), d_fr(std::vector<int32_t>{frames}), d_lo(lo), d_hi(hi), d_off(off);
  LIB_OK(s2svc_stft_logmel_fft(1, n, frames, N, hop, nmel, d_x.p, d_n.p, d_fr.p, d_tab.p, d_lo.p, d_hi.p, d_off.p, melw_n, maxw, 1e-10f,
                               (float)(1.0 / std::log(10.0)), nullptr, nullptr, d_out.p, nullptr));
  HIP_OK(hipDeviceSynchronize());
  const std::vector<float> out = d_out.host();
  double worst = 0;
  int hit = 0;
  for (int m = 0; m < nmel; ++m) {
    const double v = fb[(size_t)m * nb + k0] * A * N / 4 + (fb[(size_t)m * nb + k0 - 1] + fb[(size_t)m * nb + k0 + 1]) * A * N / 8;
    if (v < 1e-5) continue;
    ++hit;
    for (int t = 8; t < frames - 8; ++t) worst = std::fmax(worst, std::fabs(out[(size_t)t * nmel + m] - std::log10(v)));
  }
  char msg[256];
  std::snprintf(msg, sizeof msg, "s2svc_stft_logmel_fft: bin-centred cosine, %d filters hit, max |log10 diff| on interior frames %.2e", hit, worst);
  report(hit >= 2 && worst < 1e-4, msg);
}

int main() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { std::printf("no GPU\n"); return 97; }
  std::printf("libs2svc_hip ABI version %d\n", s2svc_abi_version());
  check_mas();
  check_forward_sum();
  check_stft();
  std::printf("%d check(s) failed\n", g_fail);
  return g_fail;
}
