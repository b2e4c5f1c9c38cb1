/* C restatement of the AAS monotonic alignment search -- TEST INFRASTRUCTURE ONLY (the fast twin of
 * oracle/mas.py; also the single-thread CPU baseline for the MAS kernel).
 *
 * follows seq2seq_vc/modules/alignments.py:63-93 (_monotonic_alignment_search, numba nopython) and
 * :281-310 (viterbi_decode).  Q is float64; row 0 is a float64 running prefix sum (see DESIGN.md
 * section 5, documented deviations).  Pinned by tests/golden/mas_kats.npz (paths produced by the reference itself).
 *
 *   gcc -O3 -shared -fPIC -o libmas_oracle.so mas.c
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* log_p: (T_mel, T_inp) row-major float32 with row stride `ld`; path: (T_mel,) int64.  returns 0 / -1 */
int mas_oracle(const float* log_p, int T_mel, int T_inp, int ld, int64_t* path) {
  if (T_mel <= 0 || T_inp <= 0) return -1;
  double* Q = (double*)malloc(sizeof(double) * (size_t)T_inp * T_mel);
  if (!Q) return -1;
  for (size_t i = 0; i < (size_t)T_inp * T_mel; ++i) Q[i] = -INFINITY;
  double acc = 0.0;
  for (int j = 0; j < T_mel; ++j) { /* alignments.py:72-73 */
    acc += (double)log_p[(size_t)j * ld];
    Q[j] = acc;
  }
  for (int j = 1; j < T_mel; ++j) { /* alignments.py:76-78 */
    int imax = j + 1 < T_inp ? j + 1 : T_inp;
    for (int i = 1; i < imax; ++i) {
      double a = Q[(size_t)(i - 1) * T_mel + j - 1], b = Q[(size_t)i * T_mel + j - 1];
      Q[(size_t)i * T_mel + j] = (a > b ? a : b) + (double)log_p[(size_t)j * ld + i];
    }
  }
  path[T_mel - 1] = T_inp - 1;
  for (int j = T_mel - 2; j >= 0; --j) { /* alignments.py:81-92 */
    int64_t ib = path[j + 1], ia = ib - 1;
    if (ib == 0) path[j] = 0;
    else path[j] = (Q[(size_t)ia * T_mel + j] >= Q[(size_t)ib * T_mel + j]) ? ia : ib;
  }
  free(Q);
  return 0;
}

/* viterbi_decode over a padded batch: log_p (B, Tf, Tx); ds (B, Tx) float32 (zeroed here); bin_loss out */
int viterbi_oracle(const float* log_p, int B, int Tf, int Tx, const int64_t* text_lens, const int64_t* feat_lens,
                   float* ds, double* bin_loss) {
  int64_t* path = (int64_t*)malloc(sizeof(int64_t) * (size_t)(Tf > 0 ? Tf : 1));
  if (!path) return -1;
  double total = 0.0;
  for (size_t i = 0; i < (size_t)B * Tx; ++i) ds[i] = 0.f;
  for (int b = 0; b < B; ++b) {
    int T = (int)feat_lens[b], N = (int)text_lens[b];
    const float* lp = log_p + (size_t)b * Tf * Tx;
    if (mas_oracle(lp, T, N, Tx, path)) { free(path); return -1; }
    float s = 0.f; /* fp32 mean like torch's .mean() on an fp32 vector (summation order differs: <= 1 ulp-level) */
    for (int t = 0; t < T; ++t) {
      ds[(size_t)b * Tx + path[t]] += 1.f;
      s += lp[(size_t)t * Tx + path[t]];
    }
    total -= (double)s / (double)T;
  }
  *bin_loss = total / (double)B;
  free(path);
  return 0;
}
