// GEMM with a ROW PROLOGUE for reductions over the model width (K = D <= 512), bf16:
//     C[M, N] = epilogue( prologue(X)[M, D] . W[N, D]^T )
//   prologue 1: LayerNorm of the rows (the normalised rows and the row statistics are also written, by column block 0)
//   prologue 2: rows * dropmask(p, seed) * hscale (the gradient of `res + hscale * dropout(h)` w.r.t. h; optionally written)
//   prologue 0: the rows as they are
// The whole K extent of a 64-row tile is 48 KB at D = 384: it is loaded once, transformed in registers (a wave owns rows
// wave, wave + 4, ...; a lane 8 consecutive channels -- the arithmetic of norm.hip's ln_fwd_vec / ln_bwd_vec) and kept in LDS as
// the A operand; a wave owns 64 of the tile's 256 output columns and streams their K-contiguous weight rows straight into
// MFMA B fragments (16 bytes per lane per k step, each used by the four 16-row tiles).  The epilogue is the shared one
// (gemm_common.h: bias, activation, dropout, activation-derivative mask, residual; 16-byte stores).
//
// reference: the LayerNorm in front of a feed-forward block (modules/transformer/encoder_layer.py:108-113,
// decoder_layer.py:122-127 -> positionwise_feed_forward.py:30-32) and, backward, the dropout mask in front of the block's
// second Linear.  Unfused these are a 5 us LayerNorm / element-wise launch in front of a 12 us GEMM on the dependent chain of
// the training step; the prologue is recomputed by the N / 256 column blocks of a row tile instead (64 x D values: trivial).
#include "gemm_common.h"
#include "rowblock.h"

namespace {

struct rp_args {
  int mode;
  const bf16_t* x;            // (M, D) rows, dense
  const float* gamma;         // mode 1
  const float* beta;
  float eps;
  bf16_t* y;                  // mode 1: normalised rows;  mode 2: masked rows (or NULL)
  float* mean;                // mode 1
  float* rstd;
  float p_a, hscale;          // mode 2
  const uint64_t* seed_a_base;
  uint64_t seed_a_off;
  const bf16_t* w;            // (N, D) K-contiguous
  s2svc_gemm_desc e;          // M, N, C, ldc, bias, act, drop_p / seed, emask, res, c_pre: the epilogue's fields
};

template <int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemm_rowpro_kernel(rp_args a) {
  constexpr int YP = D + 8, KS = D / 32, VPR = D / 8;
  constexpr int PF = KS <= 12 ? KS : 12;                 // k steps of weight fragments in flight (all of them up to D = 384)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Ys = reinterpret_cast<bf16_t*>(smem_raw);              // [64][YP]; after the k loop: 4 x (64 x 64) fp32 epilogue tiles
  const int M = a.e.M, N = a.e.N;
  const int m_base = blockIdx.x * 64, n_base = blockIdx.y * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  // weight rows of this wave's four 16-column tiles (clamped: columns >= N are computed on a valid row and dropped by the epilogue)
  const bf16_t* wrow[4];
  const bool live[4] = {true, true, true, true};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int n = n_base + wave * 64 + j * 16 + lr;
    n = n < N ? n : N - 1;
    wrow[j] = a.w + (int64_t)n * D + lg * 8;
  }
  bf16x8_t pre[PF][4];
  rowblock::preload_b<4, PF>(wrow, live, pre);           // in flight across the whole prologue
  {
    uint4 xr[D / 32];
    rowblock::tile_load<D>(a.x + (int64_t)m_base * D, M - m_base, xr);
    rowblock::tile_store<D>(xr, Ys);
  }
  rowblock::lds_barrier();
  if (a.mode != 0) {                                     // rows wave*16 .. +15 of the LDS image, in place (rolled: cold code is what costs)
    const bool act = lane < VPR;
    const bool writer = blockIdx.y == 0;
    float g8[8], b8[8];
    if (a.mode == 1 && act) { load_f32x8(a.gamma + lane * 8, g8); load_f32x8(a.beta + lane * 8, b8); }
    const uint64_t aseed = (a.seed_a_base ? *a.seed_a_base : 0ull) + a.seed_a_off;
    const float akeep = a.p_a > 0.f ? 1.f / (1.f - a.p_a) : 1.f;
#pragma unroll 2
    for (int rr = 0; rr < 16; ++rr) {
      const int row = wave * 16 + rr;
      const int m = m_base + row;
      const int64_t base = (int64_t)m * D;
      bf16_t* yrow = Ys + row * YP + lane * 8;
      float vv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) vv[e] = 0.f;
      if (act) unpack_bf16x8(*reinterpret_cast<const uint4*>(yrow), vv);
      if (a.mode == 1) {
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += vv[e];
        const float mean = wave_sum(sum) / (float)D;
        float sq = 0.f;
        if (act) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float d = vv[e] - mean; sq += d * d; }
        }
        const float var = wave_sum(sq) / (float)D;
        const float rstd = 1.0f / sqrtf(var + a.eps);
        if (act) {
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] = (vv[e] - mean) * rstd * g8[e] + b8[e];
        }
        if (writer && m < M && lane == 0) { a.mean[m] = mean; a.rstd[m] = rstd; }
      } else {
        if (a.p_a > 0.f && act && m < M) {
          float mk[8];
          dropout_scale8(aseed, (uint64_t)(base + lane * 8), a.p_a, akeep, mk);
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] *= mk[e];
        }
        if (a.hscale != 1.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] *= a.hscale;
        }
      }
      if (act) {
        const uint4 o = rowblock::pack8(vv);
        *reinterpret_cast<uint4*>(yrow) = o;
        if (writer && a.y && m < M) *reinterpret_cast<uint4*>(a.y + base + lane * 8) = o;
      }
    }
    rowblock::lds_barrier();
  }
  f32x4_t acc[4][4];                                     // [16-column tile][16-row tile]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[j][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  rowblock::mma_rows64<4, KS, PF>(Ys, YP, wrow, live, pre, acc);
  rowblock::lds_barrier();                                       // every wave is past its A reads: the region becomes epilogue staging
  // ---- epilogue (the arithmetic of gemm_common.h's epilogue_tile: bias, activation, dropout, relu' mask, residual) in two
  // phases: every global load of the wave's 64 x 64 tile (mask and residual rows, 8 passes of 8 rows) is issued first and parked
  // in LDS slots the same lane reads back, then a ROLLED loop does the arithmetic and the 16-byte stores -- with one wave per SIMD a
  // load inside the pass loop would cost a memory round trip per pass ----
  float* cs = reinterpret_cast<float*>(smem_raw) + wave * (64 * 64);
  uint4* es = reinterpret_cast<uint4*>(smem_raw + 4 * 64 * 64 * sizeof(float)) + wave * 512;
  uint4* rs = es + 4 * 512;
  const int nb = n_base + wave * 64;
  const s2svc_gemm_desc& d = a.e;
  if (nb < N) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = mt * 16 + lg * 4 + r, col = j * 16 + lr;
          cs[row * 64 + (col ^ (((row >> 2) & 3) << 4))] = acc[j][mt][r];
        }
    const int prow = lane >> 3, col = (lane & 7) * 8;
    const int n = nb + col;
    const bool ncol = n < N;
    if (d.emask || d.res) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int m = m_base + p * 8 + prow;
        if (m < M && ncol) {
          if (d.emask) es[p * 64 + lane] = *reinterpret_cast<const uint4*>((const bf16_t*)d.emask + (int64_t)m * d.ldm + n);
          if (d.res) rs[p * 64 + lane] = *reinterpret_cast<const uint4*>((const bf16_t*)d.res + (int64_t)m * d.ldr + n);
        }
      }
    }
    float bb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bb[e] = 0.f;
    if (d.bias && ncol) load_f32x8(d.bias + n, bb);
    const uint64_t seed = (d.seed_base ? *d.seed_base : 0ull) + d.seed_off;
    const float inv_keep = d.drop_p > 0.f ? 1.f / (1.f - d.drop_p) : 1.f;
#pragma unroll 1
    for (int p = 0; p < 8; ++p) {
      const int row = p * 8 + prow;
      const int m = m_base + row;
      if (m >= M || !ncol) continue;
      const float* src = cs + row * 64 + (col ^ (((row >> 2) & 3) << 4));
      const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = act_apply(v[e] + bb[e], d.act);
      if (d.drop_p > 0.f) {
        float mk[8];
        dropout_scale8(seed, (uint64_t)((int64_t)m * N + n), d.drop_p, inv_keep, mk);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= mk[e];
      }
      if (d.emask) {
        float ee[8];
        unpack_bf16x8(es[p * 64 + lane], ee);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ee[e] > 0.f ? v[e] : 0.f;
      }
      if (d.res) {
        float rr8[8];
        unpack_bf16x8(rs[p * 64 + lane], rr8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rr8[e];
      }
      *reinterpret_cast<uint4*>((bf16_t*)d.C + (int64_t)m * d.ldc + n) = rowblock::pack8(v);
    }
  }
}

template <int D>
int launch(const rp_args& a, hipStream_t st) {
  constexpr size_t lds_a = sizeof(bf16_t) * 64 * (D + 8), lds_e = sizeof(float) * 4 * 64 * 64 + 2 * 4 * 512 * sizeof(uint4);
  constexpr size_t lds = lds_a > lds_e ? lds_a : lds_e;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rowpro_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) { s2svc_set_error("gemm_rowpro: cannot raise the dynamic LDS limit"); return -2; }
    attr_set = true;
  }
  dim3 grid((a.e.M + 63) / 64, (a.e.N + 255) / 256);
  hipLaunchKernelGGL(gemm_rowpro_kernel<D>, grid, dim3(256), lds, st, a);
  S2S_CHECK_LAUNCH("gemm_rowpro_kernel");
  return 0;
}

}  // namespace

extern "C" int s2svc_gemm_rowpro_supported(int dtype, int D) { return dtype == S2S_BF16 && (D == 256 || D == 384 || D == 512); }

extern "C" int s2svc_gemm_rowpro(int mode, int D, const void* x, const float* gamma, const float* beta, float eps, void* y, float* mean,
                                 float* rstd, float p_a, float hscale, const uint64_t* seed_a_base, uint64_t seed_a_off, const void* w,
                                 const s2svc_gemm_desc* epi, void* stream) {
  S2S_REQUIRE(epi && s2svc_gemm_rowpro_supported(epi->dtype, D) && epi->c_dtype == S2S_BF16, "gemm_rowpro: bf16 with D in {256, 384, 512}");
  S2S_REQUIRE(mode >= 0 && mode <= 2, "gemm_rowpro: mode 0 (rows), 1 (LayerNorm) or 2 (dropout mask)");
  S2S_REQUIRE(x && w && epi->C && ((uintptr_t)x) % 16 == 0 && ((uintptr_t)w) % 16 == 0, "gemm_rowpro: missing / unaligned operand");
  S2S_REQUIRE(mode != 1 || (gamma && beta && y && mean && rstd && ((uintptr_t)gamma) % 16 == 0 && ((uintptr_t)beta) % 16 == 0 &&
                            ((uintptr_t)y) % 16 == 0), "gemm_rowpro: LayerNorm prologue needs gamma / beta / y / mean / rstd (16-byte aligned)");
  S2S_REQUIRE(!y || ((uintptr_t)y) % 16 == 0, "gemm_rowpro: y must be 16-byte aligned");
  S2S_REQUIRE(epi->K == D && epi->nb0 * epi->nb1 <= 1 && epi->splitk <= 1 && !epi->c_map && !epi->a_rowsum && !epi->c_pre &&
              !epi->accumulate && epi->alpha == 1.0f && (!epi->emask || epi->emask_mode == 0),
              "gemm_rowpro: K must equal D; no batches / split-K / c_map / row sums / c_pre / accumulate / alpha / swish mask");
  {
    const s2svc_gemm_desc& d = *epi;
    bool ok = (d.N % 8 == 0) && (d.ldc % 8 == 0) && (((uintptr_t)d.C) % 16 == 0);
    if (d.res) ok = ok && (d.ldr % 8 == 0) && (((uintptr_t)d.res) % 16 == 0);
    if (d.emask) ok = ok && (d.ldm % 8 == 0) && (((uintptr_t)d.emask) % 16 == 0);
    if (d.c_pre) ok = ok && (((uintptr_t)d.c_pre) % 16 == 0);
    if (d.bias) ok = ok && (((uintptr_t)d.bias) % 16 == 0);
    S2S_REQUIRE(ok, "gemm_rowpro: the output (and bias / res / emask / c_pre) must allow 16-byte accesses");
  }
  if (epi->M == 0 || epi->N == 0) return 0;
  rp_args a;
  a.mode = mode; a.x = (const bf16_t*)x; a.gamma = gamma; a.beta = beta; a.eps = eps; a.y = (bf16_t*)y; a.mean = mean; a.rstd = rstd;
  a.p_a = p_a; a.hscale = hscale; a.seed_a_base = seed_a_base; a.seed_a_off = seed_a_off; a.w = (const bf16_t*)w; a.e = *epi;
  hipStream_t st = (hipStream_t)stream;
  if (D == 256) return launch<256>(a, st);
  if (D == 384) return launch<384>(a, st);
  return launch<512>(a, st);
}
