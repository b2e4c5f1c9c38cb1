// Fused attention SUB-LAYERS for short sequences (T1, T2 <= 64), bf16: one workgroup per (utterance, head), ONE launch for
//   forward : [LayerNorm of the 64 x D input rows] -> this head's Q (K, V) projection -> scores, mask, softmax, dropout, P.V
//   backward: [dropout mask | LayerNorm backward of the residual stream] -> dCtx_h = dA . Wo[:, head] -> dQ, dK, dV
//
// reference: modules/transformer/encoder_layer.py:61-119 / decoder_layer.py:63-134 (norm -> self_attn / src_attn -> dropout ->
// residual) around modules/transformer/attention.py:39-111.  The unfused path runs LayerNorm, the packed projection GEMM and
// the attention kernel as three launch-bound kernels (5 + 10 + 11 us at VTN's 2016 x 384 rows) and the backward as
// LayerNorm-backward, output-projection data gradient and attention backward (5.5 + 7 + 18 us); a dependent kernel costs
// ~4.5 us on this stack whatever it does, so the chain is what the step time is made of.
//
// Workgroups do not communicate: the head's workgroup recomputes the row prologue (LayerNorm of 64 x D values: trivial) and
// reads the whole input tile (48 KB at D = 384); side outputs every head would produce identically (normalised rows, row
// statistics, the residual-stream gradient) are written by head (row % H).  The projection weights never touch LDS: a wave
// owns 16-column tiles of the head's output and streams their K-contiguous weight rows straight into MFMA B fragments
// (16 bytes per lane per k step, each fragment used by the four 16-row tiles of the input), A fragments come from the LDS
// image of the normalised rows.  The attention part is csrc/attn_fused.hip with its operands already in LDS.
// Dropout masks are functions of (seed, element index) -- (seed of the residual dropout, row * D + column) and
// (seed of the attention dropout, index in the attention map) -- exactly the masks the unfused kernels draw.
#include "rowblock.h"
#include "../../include/s2svc_hip.h"

namespace {

constexpr int TP = 72;                       // pitch (elements) of the [*][64] tiles: 144-byte rows, conflict-free b128 reads
constexpr float NEG = -3.4028234663852886e38f;

__device__ __forceinline__ float grp16_max(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float grp16_sum(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ bf16x8_t zero8() { return (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0}; }

// rows [0, R) x DK of a (rows, ld) matrix -> LDS, in two halves so that the loads of several tiles (and of the row prologue)
// are in flight together: stage_load puts this thread's DK / 32 16-byte pieces into registers, stage_store* writes them.
//   row-major image: lds[row * (DK + 8) + d]     (piece p = t + 256 * i: row = p / (DK/8), c = p % (DK/8))
//   transposed image: lds[d * TP + row]          (piece p: row = p % 64, c = p / 64: conflict-free 2-byte LDS writes)
template <int DK, bool TRANSPOSED>
__device__ __forceinline__ void stage_load(const bf16_t* g, int64_t ld, int R, uint4 (&v)[DK / 32]) {
  constexpr int ppr = DK / 8;
#pragma unroll
  for (int i = 0; i < DK / 32; ++i) {
    const int p = threadIdx.x + 256 * i;
    const int row = TRANSPOSED ? p % 64 : p / ppr, c = TRANSPOSED ? p / 64 : p % ppr;
    v[i] = make_uint4(0, 0, 0, 0);
    if (row < R) v[i] = *reinterpret_cast<const uint4*>(g + (int64_t)row * ld + c * 8);
  }
}
template <int DK>
__device__ __forceinline__ void stage_store(const uint4 (&v)[DK / 32], bf16_t* lds) {
  constexpr int ppr = DK / 8;
#pragma unroll
  for (int i = 0; i < DK / 32; ++i) {
    const int p = threadIdx.x + 256 * i;
    const int row = p / ppr, c = p % ppr;
    *reinterpret_cast<uint4*>(lds + row * (DK + 8) + c * 8) = v[i];
  }
}
template <int DK>
__device__ __forceinline__ void stage_store_t(const uint4 (&v)[DK / 32], bf16_t* lds) {
#pragma unroll
  for (int i = 0; i < DK / 32; ++i) {
    const int p = threadIdx.x + 256 * i;
    const int row = p % 64, c = p / 64;
    const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lds[(c * 8 + 2 * e) * TP + row] = (bf16_t)(w[e] & 0xffffu);
      lds[(c * 8 + 2 * e + 1) * TP + row] = (bf16_t)(w[e] >> 16);
    }
  }
}

// A wave's 16 x DK accumulator tile -> global rows with 16-byte stores through a wave-private LDS tile (pitch DK + 8).
template <int DK>
__device__ __forceinline__ void store_tile_rows(const f32x4_t (&acc)[DK / 16], bf16_t* stage, bf16_t* dst, int64_t ld, int rows_valid) {
  constexpr int KP = DK + 8;
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
#pragma unroll
  for (int dn = 0; dn < DK / 16; ++dn)
#pragma unroll
    for (int r = 0; r < 4; ++r) stage[(lg * 4 + r) * KP + dn * 16 + lr] = f2bf(acc[dn][r]);
  constexpr int VPR = DK / 8;                         // 16-byte vectors per row
#pragma unroll
  for (int v = lane; v < 16 * VPR; v += 64) {
    const int row = v / VPR, c8 = v - row * VPR;
    if (row < rows_valid)
      *reinterpret_cast<uint4*>(dst + (int64_t)row * ld + c8 * 8) = *reinterpret_cast<const uint4*>(stage + row * KP + c8 * 8);
  }
}

// 64 rows x DK columns of an LDS tile (pitch DK + 8) -> global rows (all 256 threads, 16-byte stores)
template <int DK>
__device__ __forceinline__ void store_lds_rows(const bf16_t* lds, bf16_t* dst, int64_t ld, int rows_valid) {
  constexpr int KP = DK + 8, VPR = DK / 8;
  for (int v = threadIdx.x; v < 64 * VPR; v += 256) {
    const int row = v / VPR, c8 = v - row * VPR;
    if (row < rows_valid)
      *reinterpret_cast<uint4*>(dst + (int64_t)row * ld + c8 * 8) = *reinterpret_cast<const uint4*>(lds + row * KP + c8 * 8);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
struct ab_fwd_args {
  int H, T1, T2;
  const bf16_t* x;            // (B, T1, D) input rows, dense
  const float* gamma;         // LayerNorm over D in front of the projection (NULL: the rows are used as they are)
  const float* beta;
  float eps;
  bf16_t* y;                  // (B, T1, D) normalised rows (gamma != NULL), written by head (row % H)
  float* mean;                // (B * T1) row statistics
  float* rstd;
  const bf16_t* w;            // packed projection weight rows [NPROJ * D][D] (K-contiguous): [Wq ; Wk ; Wv] or Wq
  const float* bias;          // [NPROJ * D] or NULL
  bf16_t* proj;               // (B, T1, NPROJ * D): the packed projection (saved for the backward pass)
  const bf16_t* k;            // NPROJ == 1: keys / values of the memory, (B, T2, .) views
  int64_t ldk, kbs;
  const bf16_t* v;
  int64_t ldv, vbs;
  const int32_t* klen;
  int causal;
  float scale, p;
  const uint64_t* seed_base;
  uint64_t seed_off;
  bf16_t* attn;               // (B, H, T1, ld) pre-dropout probabilities, ld % 8 == 0
  int ld;
  bf16_t* out;                // (B, T1, D) context vectors
};

template <int DK, int D, int NPROJ>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void attn_block_fwd_kernel(ab_fwd_args a) {
  constexpr int KP = DK + 8, YP = D + 8, KS = D / 32, VPR = D / 8;
  constexpr int NT = NPROJ * DK / 16;            // 16-column tiles of this head's projection
  constexpr int MAXT = (NT + 3) / 4;             // per wave (tile t belongs to wave t % 4)
  constexpr int PF = MAXT * KS <= 32 ? KS : 6;   // k steps of weight fragments in flight (4 VGPRs per tile and step)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Ys = reinterpret_cast<bf16_t*>(smem_raw);      // [64][YP] (normalised) input rows
  bf16_t* Qs = Ys + 64 * YP;                             // [64][KP]
  bf16_t* Ks = Qs + 64 * KP;                             // [64][KP]
  bf16_t* Vs = Ks + 64 * KP;                             // [64][KP]  (row-major V: only for the store of the packed projection)
  bf16_t* Vt = Vs + 64 * KP;                             // [DK][TP]
  bf16_t* Pw = Vt + DK * TP;                             // [4][16][TP]  dropped probabilities (A operand of P.V)
  bf16_t* Pa = Pw + 4 * 16 * TP;                         // [4][16][TP]  probabilities before dropout (the attention map's rows)
  const int H = a.H, T1 = a.T1, T2 = a.T2;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const uint64_t seed = (a.seed_base ? *a.seed_base : 0ull) + a.seed_off;
  const float inv_keep = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  const int kl = a.klen ? (a.klen[b] < T2 ? a.klen[b] : T2) : T2;
  // weight fragments of this wave's tiles: the first PF k steps are in flight across the whole prologue
  const bf16_t* wrow[MAXT];
  bool live[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int t = j * 4 + wave;                      // tile index (may be >= NT for the last j)
    live[j] = t < NT;
    const int n0 = (t < NT ? t : 0) * 16;
    const int which = n0 / DK, c0 = n0 - which * DK;
    wrow[j] = a.w + ((int64_t)which * D + h * DK + c0 + lr) * D + lg * 8;
  }
  bf16x8_t pre[PF][MAXT];
  rowblock::preload_b<MAXT, PF>(wrow, live, pre);
  {
    uint4 kreg[DK / 32], vreg[DK / 32], xr[D / 32];
    if (NPROJ == 1) {                                    // memory keys / values
      stage_load<DK, false>(a.k + (int64_t)b * a.kbs + h * DK, a.ldk, T2, kreg);
      stage_load<DK, true>(a.v + (int64_t)b * a.vbs + h * DK, a.ldv, T2, vreg);
    }
    rowblock::tile_load<D>(a.x + (int64_t)b * T1 * D, T1, xr);
    rowblock::tile_store<D>(xr, Ys);
    if (NPROJ == 1) {
      stage_store<DK>(kreg, Ks);
      stage_store_t<DK>(vreg, Vt);
    }
  }
  rowblock::lds_barrier();
  // ---- prologue: LayerNorm of rows wave*16 .. +15 of the LDS image, in place (rolled loop: cold code is what costs) ----
  if (a.gamma) {
    const bool act = lane < VPR;
    float g8[8], b8[8];
    if (act) { load_f32x8(a.gamma + lane * 8, g8); load_f32x8(a.beta + lane * 8, b8); }
#pragma unroll 2
    for (int rr = 0; rr < 16; ++rr) {
      const int row = wave * 16 + rr;
      bf16_t* yrow = Ys + row * YP + lane * 8;
      float vv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) vv[e] = 0.f;
      if (act) unpack_bf16x8(*reinterpret_cast<const uint4*>(yrow), vv);
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
        const uint4 o = row < T1 ? rowblock::pack8(vv) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(yrow) = o;
        if (row < T1 && (row % H) == h) *reinterpret_cast<uint4*>(a.y + ((int64_t)b * T1 + row) * D + lane * 8) = o;
      }
      if (lane == 0 && row < T1 && (row % H) == h) { a.mean[(int64_t)b * T1 + row] = mean; a.rstd[(int64_t)b * T1 + row] = rstd; }
    }
    rowblock::lds_barrier();
  }
  // ---- projection: [64 x D] . W_h^T -> Q (K, V) of this head; wave w owns tiles w, w + 4, ... ----
  {
    f32x4_t acc[MAXT][4];
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[j][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    rowblock::mma_rows64<MAXT, KS, PF>(Ys, YP, wrow, live, pre, acc);
    // bias, then into the LDS images the attention part reads (accumulator layout: row mt*16 + lg*4 + r, column c0 + lr)
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int t = j * 4 + wave;
      if (t < NT) {
        const int n0 = t * 16;
        const int which = n0 / DK, c0 = n0 - which * DK;
        const float bs = a.bias ? a.bias[which * D + h * DK + c0 + lr] : 0.f;
        bf16_t* dst = (which == 0 ? Qs : (which == 1 ? Ks : Vs)) + (lg * 4) * KP + c0 + lr;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const uint32_t p01 = rowblock::pack2(acc[j][mt][0] + bs, acc[j][mt][1] + bs);
          const uint32_t p23 = rowblock::pack2(acc[j][mt][2] + bs, acc[j][mt][3] + bs);
          dst[(mt * 16 + 0) * KP] = (bf16_t)(p01 & 0xffffu);
          dst[(mt * 16 + 1) * KP] = (bf16_t)(p01 >> 16);
          dst[(mt * 16 + 2) * KP] = (bf16_t)(p23 & 0xffffu);
          dst[(mt * 16 + 3) * KP] = (bf16_t)(p23 >> 16);
          if (which == 2) *reinterpret_cast<uint2*>(Vt + (c0 + lr) * TP + mt * 16 + lg * 4) = make_uint2(p01, p23);
        }
      }
    }
  }
  rowblock::lds_barrier();
  // the packed projection goes to HBM for the backward pass (rows < T1)
  {
    bf16_t* pb = a.proj + (int64_t)b * T1 * (NPROJ * D) + h * DK;
    store_lds_rows<DK>(Qs, pb, NPROJ * D, T1);
    if (NPROJ == 3) {
      store_lds_rows<DK>(Ks, pb + D, NPROJ * D, T1);
      store_lds_rows<DK>(Vs, pb + 2 * D, NPROJ * D, T1);
    }
  }
  // ---- attention (csrc/attn_fused.hip with the operands in LDS) ----
  bf16x8_t qa[DK / 32];
#pragma unroll
  for (int ks = 0; ks < DK / 32; ++ks) qa[ks] = *reinterpret_cast<const bf16x8_t*>(Qs + (wave * 16 + lr) * KP + ks * 32 + lg * 8);
  f32x4_t s[4];
#pragma unroll
  for (int jn = 0; jn < 4; ++jn) {
    s[jn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < DK / 32; ++ks) {
      const bf16x8_t kb = *reinterpret_cast<const bf16x8_t*>(Ks + (jn * 16 + lr) * KP + ks * 32 + lg * 8);
      s[jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[ks], kb, s[jn], 0, 0, 0);
    }
  }
  bf16_t* pw = Pw + wave * 16 * TP;
  bf16_t* pa = Pa + wave * 16 * TP;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = wave * 16 + lg * 4 + r;
    float val[4];
    float mx = NEG;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int j = jn * 16 + lr;
      const bool ok = j < kl && (!a.causal || j <= i);
      val[jn] = ok ? s[jn][r] * a.scale : NEG;
      mx = fmaxf(mx, val[jn]);
    }
    mx = grp16_max(mx);
    float ex[4];
    float sum = 0.f;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) { ex[jn] = expf(val[jn] - mx); sum += ex[jn]; }
    sum = grp16_sum(sum);
    const float inv = 1.f / sum;
    const int64_t arow = ((int64_t)(b * H + h) * T1 + i) * a.ld;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int j = jn * 16 + lr;
      const bool ok = j < kl && (!a.causal || j <= i);
      const bf16_t pb = rowblock::cvt1(ok ? ex[jn] * inv : 0.f);     // masked_fill(mask, 0.0) after the softmax
      pa[(lg * 4 + r) * TP + j] = pb;
      float pd = bf2f(pb);                                           // P.V consumes the stored (rounded) probabilities
      if (a.p > 0.f) pd *= dropout_scale(seed, (uint64_t)(arow + j), a.p, inv_keep);
      pw[(lg * 4 + r) * TP + j] = rowblock::cvt1(pd);
    }
  }
  // the attention map's 16 rows of this wave: 16-byte stores from the wave-private tile (LDS ops of a wave execute in order)
  {
    const int vpr = a.ld / 8;                          // 16-byte vectors per row (<= 8)
    for (int v = lane; v < 16 * vpr; v += 64) {
      const int row = v / vpr, c8 = v - row * vpr;
      const int i = wave * 16 + row;
      if (i < T1)
        *reinterpret_cast<uint4*>(a.attn + ((int64_t)(b * H + h) * T1 + i) * a.ld + c8 * 8) = *reinterpret_cast<const uint4*>(pa + row * TP + c8 * 8);
    }
  }
  f32x4_t o[DK / 16];
#pragma unroll
  for (int dn = 0; dn < DK / 16; ++dn) o[dn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const bf16x8_t pfr = *reinterpret_cast<const bf16x8_t*>(pw + lr * TP + ks * 32 + lg * 8);
#pragma unroll
    for (int dn = 0; dn < DK / 16; ++dn) {
      const bf16x8_t vb = *reinterpret_cast<const bf16x8_t*>(Vt + (dn * 16 + lr) * TP + ks * 32 + lg * 8);
      o[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pfr, vb, o[dn], 0, 0, 0);
    }
  }
  rowblock::lds_barrier();       // the cooperative store of Qs above (all threads read all rows) is done: Qs rows of this wave become its staging area
  store_tile_rows<DK>(o, Qs + wave * 16 * KP, a.out + ((int64_t)b * T1 + wave * 16) * D + h * DK, D, T1 - wave * 16);
}

// ---------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------
struct ab_bwd_args {
  int H, T1, T2;
  // prologue: the gradient dA (B, T1, D) of the output projection's result, from the gradient of the residual stream
  //   mode 0: dA = g * dropmask(p_res) * hscale                       (g = gradient of s = res + hscale * dropout(A))
  //   mode 1: dS = ds_extra + LayerNorm'(g; s, mean, rstd, gamma);  dA = dS * dropmask(p_res) * hscale;  dS is written
  int mode;
  const bf16_t* g;            // (B, T1, D)
  const bf16_t* s;            // mode 1: LayerNorm input rows
  const float* mean;
  const float* rstd;
  const float* gamma;
  const bf16_t* ds_extra;     // mode 1: gradient reaching s directly (or NULL)
  bf16_t* ds;                 // mode 1: (B, T1, D), written by head (row % H)
  bf16_t* da;                 // (B, T1, D) or NULL: written by head (row % H) (operand of the output projection's weight gradient)
  float p_res, hscale;
  const uint64_t* seed_res_base;
  uint64_t seed_res_off;
  const bf16_t* wo_t;         // Wo^T (D_in, D_out) row-major = rows [h*DK + n] hold Wo[:, h*DK + n]  (K-contiguous B operand)
  // attention backward
  const bf16_t* q; int64_t ldq, qbs;
  const bf16_t* k; int64_t ldk, kbs;
  const bf16_t* v; int64_t ldv, vbs;
  const bf16_t* attn; const bf16_t* dattn; int ld;
  float scale, p;
  const uint64_t* seed_base;
  uint64_t seed_off;
  bf16_t* dq; int64_t lddq, dqbs;
  bf16_t* dk; int64_t lddk, dkbs;
  bf16_t* dv; int64_t lddv, dvbs;
};

template <int DK, int D>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void attn_block_bwd_kernel(ab_bwd_args a) {
  constexpr int KP = DK + 8, YP = D + 8, KS = D / 32, VPR = D / 8;
  constexpr int NT = DK / 16;
  constexpr int MAXT = (NT + 3) / 4;
  constexpr int PF = KS <= 12 ? KS : 12;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* DAs = reinterpret_cast<bf16_t*>(smem_raw);     // [64][YP] g rows -> dA rows; dead after the projection: dSt | Pt | dSw live there
  bf16_t* dSt = DAs;                                     // [64][TP]
  bf16_t* Pt = dSt + 64 * TP;                            // [64][TP]
  bf16_t* dSw = Pt + 64 * TP;                            // [4][16][TP]
  bf16_t* Vs = DAs + 64 * YP;                            // [64][KP] V rows (B operand of dP = dO V^T); later output staging
  bf16_t* Kt = Vs + 64 * KP;                             // [DK][TP]
  bf16_t* Qt = Kt + DK * TP;                             // [DK][TP]
  bf16_t* dOt = Qt + DK * TP;                            // [DK][TP]
  bf16_t* dOs = dOt + DK * TP;                           // [64][KP] dO rows (A operand of dP)
  bf16_t* Ss = DAs + 64 * YP;                            // prologue only (mode 1): [64][YP] LayerNorm input rows ...
  bf16_t* Es = Ss + 64 * YP;                             // ... and [64][YP] directly arriving gradient rows (over Vs .. dOs and beyond)
  static_assert(64 * YP >= 2 * 64 * TP + 4 * 16 * TP, "the dS / P tiles must fit into the dA region");
  const int H = a.H, T1 = a.T1, T2 = a.T2;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const uint64_t seed = (a.seed_base ? *a.seed_base : 0ull) + a.seed_off;
  const float inv_keep = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  // everything this kernel reads from global memory before its first barrier is issued up front
  const bf16_t* wrow[MAXT];
  bool live[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int t = j * 4 + wave;
    live[j] = t < NT;
    wrow[j] = a.wo_t + ((int64_t)h * DK + (t < NT ? t : 0) * 16 + lr) * D + lg * 8;
  }
  bf16x8_t pre[PF][MAXT];
  rowblock::preload_b<MAXT, PF>(wrow, live, pre);
  uint4 vreg[DK / 32], kreg[DK / 32], qreg[DK / 32];
  stage_load<DK, false>(a.v + (int64_t)b * a.vbs + h * DK, a.ldv, T2, vreg);
  stage_load<DK, true>(a.k + (int64_t)b * a.kbs + h * DK, a.ldk, T2, kreg);
  stage_load<DK, true>(a.q + (int64_t)b * a.qbs + h * DK, a.ldq, T1, qreg);
  {
    uint4 gr[D / 32];
    rowblock::tile_load<D>(a.g + (int64_t)b * T1 * D, T1, gr);
    if (a.mode == 1) {
      uint4 sr[D / 32];
      rowblock::tile_load<D>(a.s + (int64_t)b * T1 * D, T1, sr);
      if (a.ds_extra) {
        uint4 er[D / 32];
        rowblock::tile_load<D>(a.ds_extra + (int64_t)b * T1 * D, T1, er);
        rowblock::tile_store<D>(er, Es);
      }
      rowblock::tile_store<D>(sr, Ss);
    }
    rowblock::tile_store<D>(gr, DAs);
  }
  rowblock::lds_barrier();
  // ---- prologue: dA rows wave*16 .. +15, in place (rolled) ----
  if (a.mode == 1 || a.p_res > 0.f || a.hscale != 1.f) {
    const bool act = lane < VPR;
    const uint64_t rseed = (a.seed_res_base ? *a.seed_res_base : 0ull) + a.seed_res_off;
    const float rkeep = a.p_res > 0.f ? 1.f / (1.f - a.p_res) : 1.f;
    float gm[8];
    if (a.mode == 1 && act) load_f32x8(a.gamma + lane * 8, gm);
    float mu_l = 0.f, rs_l = 0.f;                        // statistics of this wave's 16 rows: one load, lane rr holds row rr's
    if (a.mode == 1 && lane < 16 && wave * 16 + lane < T1) {
      mu_l = a.mean[(int64_t)b * T1 + wave * 16 + lane];
      rs_l = a.rstd[(int64_t)b * T1 + wave * 16 + lane];
    }
#pragma unroll 2
    for (int rr = 0; rr < 16; ++rr) {
      const int row = wave * 16 + rr;
      const int64_t grow = (int64_t)b * T1 + row;
      const int64_t base = grow * D;
      const bool live_r = row < T1 && act;
      bf16_t* drow = DAs + row * YP + lane * 8;
      float vv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) vv[e] = 0.f;
      if (act) unpack_bf16x8(*reinterpret_cast<const uint4*>(drow), vv);
      if (a.mode == 1) {
        float xh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) xh[e] = 0.f;
        if (act) unpack_bf16x8(*reinterpret_cast<const uint4*>(Ss + row * YP + lane * 8), xh);
        const float mu = rowblock::readlane_f(mu_l, rr), rs = rowblock::readlane_f(rs_l, rr);
        float sa = 0.f, sb = 0.f;
        if (live_r) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            vv[e] *= gm[e];
            xh[e] = (xh[e] - mu) * rs;
            sa += vv[e];
            sb += vv[e] * xh[e];
          }
        }
        sa = wave_sum(sa) / (float)D;
        sb = wave_sum(sb) / (float)D;
        if (live_r) {
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] = rs * (vv[e] - sa - xh[e] * sb);
          if (a.ds_extra) {
            float ex[8];
            unpack_bf16x8(*reinterpret_cast<const uint4*>(Es + row * YP + lane * 8), ex);
#pragma unroll
            for (int e = 0; e < 8; ++e) vv[e] += ex[e];
          }
          if ((row % H) == h) *reinterpret_cast<uint4*>(a.ds + base + lane * 8) = rowblock::pack8(vv);
          // (the unfused LayerNorm backward derives dh from the unrounded values as well)
        }
      }
      if (live_r) {
        if (a.p_res > 0.f) {
          float m[8];
          dropout_scale8(rseed, (uint64_t)(base + lane * 8), a.p_res, rkeep, m);
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] *= m[e];
        }
        if (a.hscale != 1.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) vv[e] *= a.hscale;
        }
      }
      if (act) {
        const uint4 o = live_r ? rowblock::pack8(vv) : make_uint4(0, 0, 0, 0);
        *reinterpret_cast<uint4*>(drow) = o;
        if (live_r && a.da && (row % H) == h) *reinterpret_cast<uint4*>(a.da + base + lane * 8) = o;
      }
    }
    rowblock::lds_barrier();                 // the s / extra tiles are dead: their region takes the attention operands
  }
  stage_store<DK>(vreg, Vs);
  stage_store_t<DK>(kreg, Kt);
  stage_store_t<DK>(qreg, Qt);
  // ---- dO_h = dA . Wo[:, head columns]  (64 x DK, reduction over D); wave w owns tiles w, w + 4, ... ----
  {
    f32x4_t acc[MAXT][4];
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[j][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    rowblock::mma_rows64<MAXT, KS, PF>(DAs, YP, wrow, live, pre, acc);
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const int t = j * 4 + wave;
      if (t < NT) {
        const int c0 = t * 16;
        bf16_t* dst = dOs + (lg * 4) * KP + c0 + lr;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const uint32_t p01 = rowblock::pack2(acc[j][mt][0], acc[j][mt][1]);
          const uint32_t p23 = rowblock::pack2(acc[j][mt][2], acc[j][mt][3]);
          dst[(mt * 16 + 0) * KP] = (bf16_t)(p01 & 0xffffu);
          dst[(mt * 16 + 1) * KP] = (bf16_t)(p01 >> 16);
          dst[(mt * 16 + 2) * KP] = (bf16_t)(p23 & 0xffffu);
          dst[(mt * 16 + 3) * KP] = (bf16_t)(p23 >> 16);
          *reinterpret_cast<uint2*>(dOt + (c0 + lr) * TP + mt * 16 + lg * 4) = make_uint2(p01, p23);
        }
      }
    }
  }
  rowblock::lds_barrier();                 // dO / V / K^T / Q^T complete; every wave is past its DAs reads: the region now holds dSt / Pt / dSw
  // ---- attention backward (csrc/attn_fused.hip) ----
  bf16x8_t da[DK / 32];
#pragma unroll
  for (int ks = 0; ks < DK / 32; ++ks) da[ks] = *reinterpret_cast<const bf16x8_t*>(dOs + (wave * 16 + lr) * KP + ks * 32 + lg * 8);
  f32x4_t dp[4];
#pragma unroll
  for (int jn = 0; jn < 4; ++jn) {
    dp[jn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < DK / 32; ++ks) {
      const bf16x8_t vb = *reinterpret_cast<const bf16x8_t*>(Vs + (jn * 16 + lr) * KP + ks * 32 + lg * 8);
      dp[jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[ks], vb, dp[jn], 0, 0, 0);
    }
  }
  bf16_t* dsw = dSw + wave * 16 * TP;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int il = lg * 4 + r, i = wave * 16 + il;
    const int64_t arow = ((int64_t)(b * H + h) * T1 + i) * a.ld;
    float pv[4], t[4], m[4];
    float dot = 0.f;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int j = jn * 16 + lr;
      const bool in = i < T1 && j < T2;
      pv[jn] = in ? bf2f(a.attn[arow + j]) : 0.f;
      m[jn] = (a.p > 0.f && in) ? dropout_scale(seed, (uint64_t)(arow + j), a.p, inv_keep) : 1.f;
      t[jn] = dp[jn][r] * m[jn] + ((a.dattn && in) ? bf2f(a.dattn[arow + j]) : 0.f);
      dot += pv[jn] * t[jn];
    }
    dot = grp16_sum(dot);
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int j = jn * 16 + lr;
      const bf16_t ds = rowblock::cvt1(pv[jn] * (t[jn] - dot) * a.scale);
      const bf16_t pd = rowblock::cvt1(pv[jn] * m[jn]);
      dsw[il * TP + j] = ds;
      dSt[j * TP + i] = ds;
      Pt[j * TP + i] = pd;
    }
  }
  rowblock::lds_barrier();
  {
    f32x4_t acc[DK / 16];
#pragma unroll
    for (int dn = 0; dn < DK / 16; ++dn) acc[dn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(dsw + lr * TP + ks * 32 + lg * 8);
#pragma unroll
      for (int dn = 0; dn < DK / 16; ++dn) {
        const bf16x8_t bb = *reinterpret_cast<const bf16x8_t*>(Kt + (dn * 16 + lr) * TP + ks * 32 + lg * 8);
        acc[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bb, acc[dn], 0, 0, 0);
      }
    }
    // (Vs was last read before the barrier above: its rows wave*16.. are this wave's output staging area from here on)
    store_tile_rows<DK>(acc, Vs + wave * 16 * KP, a.dq + (int64_t)b * a.dqbs + (int64_t)(wave * 16) * a.lddq + h * DK, a.lddq, T1 - wave * 16);
  }
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const bf16_t* At = which == 0 ? dSt : Pt;
    const bf16_t* Bt = which == 0 ? Qt : dOt;
    f32x4_t acc[DK / 16];
#pragma unroll
    for (int dn = 0; dn < DK / 16; ++dn) acc[dn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t af = *reinterpret_cast<const bf16x8_t*>(At + (wave * 16 + lr) * TP + ks * 32 + lg * 8);
#pragma unroll
      for (int dn = 0; dn < DK / 16; ++dn) {
        const bf16x8_t bb = *reinterpret_cast<const bf16x8_t*>(Bt + (dn * 16 + lr) * TP + ks * 32 + lg * 8);
        acc[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bb, acc[dn], 0, 0, 0);
      }
    }
    bf16_t* dst = which == 0 ? a.dk + (int64_t)b * a.dkbs : a.dv + (int64_t)b * a.dvbs;
    const int64_t ldd = which == 0 ? a.lddk : a.lddv;
    store_tile_rows<DK>(acc, Vs + wave * 16 * KP, dst + (int64_t)(wave * 16) * ldd + h * DK, ldd, T2 - wave * 16);
  }
}

template <int DK, int D, int NPROJ>
constexpr size_t fwd_lds_bytes() { return sizeof(bf16_t) * (64 * (D + 8) + 3 * 64 * (DK + 8) + DK * TP + 2 * 4 * 16 * TP); }
template <int DK, int D>
constexpr size_t bwd_lds_bytes() {
  constexpr size_t main_b = sizeof(bf16_t) * (64 * (D + 8) + 2 * 64 * (DK + 8) + 3 * DK * TP), pro_b = sizeof(bf16_t) * 3 * 64 * (D + 8);
  return main_b > pro_b ? main_b : pro_b;          // the prologue holds three row tiles (g, s, extra) where the attention operands go later
}

template <int DK, int D, int NPROJ>
int launch_fwd(int B, const ab_fwd_args& a, hipStream_t st) {
  constexpr size_t lds = fwd_lds_bytes<DK, D, NPROJ>();
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_fwd_kernel<DK, D, NPROJ>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) { s2svc_set_error("attn_block_fwd: cannot raise the dynamic LDS limit"); return -2; }
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_block_fwd_kernel<DK, D, NPROJ>), dim3(B * a.H), dim3(256), lds, st, a);
  S2S_CHECK_LAUNCH("attn_block_fwd_kernel");
  return 0;
}
template <int DK, int D>
int launch_bwd(int B, const ab_bwd_args& a, hipStream_t st) {
  constexpr size_t lds = bwd_lds_bytes<DK, D>();
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_block_bwd_kernel<DK, D>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) { s2svc_set_error("attn_block_bwd: cannot raise the dynamic LDS limit"); return -2; }
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_block_bwd_kernel<DK, D>), dim3(B * a.H), dim3(256), lds, st, a);
  S2S_CHECK_LAUNCH("attn_block_bwd_kernel");
  return 0;
}

bool al16(const void* p) { return ((uintptr_t)p) % 16 == 0; }

}  // namespace

extern "C" int s2svc_attn_block_supported(int dtype, int T1, int T2, int D, int H) {
  if (dtype != S2S_BF16 || H < 1 || D % H != 0) return 0;
  const int dk = D / H;
  return T1 >= 1 && T1 <= 64 && T2 >= 1 && T2 <= 64 && ((D == 256 && dk == 64) || (D == 384 && dk == 96));
}

extern "C" int s2svc_attn_block_fwd(int B, int H, int T1, int T2, int D, int nproj, const void* x, const float* gamma, const float* beta,
                                    float eps, void* y, float* mean, float* rstd, const void* w, const float* bias, void* proj,
                                    const void* k, int64_t ldk, int64_t kbs, const void* v, int64_t ldv, int64_t vbs,
                                    const int32_t* klen, int causal, float scale, float drop_p, const uint64_t* seed_base,
                                    uint64_t seed_off, void* attn, int ld, void* out, void* stream) {
  S2S_REQUIRE(s2svc_attn_block_supported(S2S_BF16, T1, T2, D, H), "attn_block_fwd: unsupported shape (bf16, T <= 64, (D, d_k) in {(256,64),(384,96)})");
  S2S_REQUIRE(nproj == 3 || nproj == 1, "attn_block_fwd: nproj must be 3 (self-attention) or 1 (source attention)");
  S2S_REQUIRE(nproj == 1 || T1 == T2, "attn_block_fwd: self-attention needs T1 == T2");
  S2S_REQUIRE(x && w && proj && attn && out && ld >= T2 && ld % 8 == 0 && ld <= 64 && al16(attn), "attn_block_fwd: missing operand (attn rows: ld % 8 == 0, 16-byte aligned)");
  S2S_REQUIRE(al16(x) && al16(w) && al16(proj) && al16(out) && (!gamma || (al16(gamma) && al16(beta) && al16(y) && mean && rstd)),
              "attn_block_fwd: 16-byte aligned operands (and y / mean / rstd with a LayerNorm)");
  if (nproj == 1)
    S2S_REQUIRE(k && v && al16(k) && al16(v) && ldk % 8 == 0 && kbs % 8 == 0 && ldv % 8 == 0 && vbs % 8 == 0,
                "attn_block_fwd: memory keys / values must be 16-byte aligned views with strides that are multiples of 8");
  if (B == 0) return 0;
  ab_fwd_args a;
  a.H = H; a.T1 = T1; a.T2 = T2;
  a.x = (const bf16_t*)x; a.gamma = gamma; a.beta = beta; a.eps = eps; a.y = (bf16_t*)y; a.mean = mean; a.rstd = rstd;
  a.w = (const bf16_t*)w; a.bias = bias; a.proj = (bf16_t*)proj;
  a.k = (const bf16_t*)k; a.ldk = ldk; a.kbs = kbs; a.v = (const bf16_t*)v; a.ldv = ldv; a.vbs = vbs;
  a.klen = klen; a.causal = causal; a.scale = scale; a.p = drop_p; a.seed_base = seed_base; a.seed_off = seed_off;
  a.attn = (bf16_t*)attn; a.ld = ld; a.out = (bf16_t*)out;
  hipStream_t st = (hipStream_t)stream;
  if (nproj == 3) {
    if (D == 256) return launch_fwd<64, 256, 3>(B, a, st);
    return launch_fwd<96, 384, 3>(B, a, st);
  }
  if (D == 256) return launch_fwd<64, 256, 1>(B, a, st);
  return launch_fwd<96, 384, 1>(B, a, st);
}

extern "C" int s2svc_attn_block_bwd(int B, int H, int T1, int T2, int D, int mode, const void* g, const void* s, const float* mean,
                                    const float* rstd, const float* gamma, const void* ds_extra, void* ds, void* da, float p_res,
                                    float hscale, const uint64_t* seed_res_base, uint64_t seed_res_off, const void* wo_t,
                                    const void* q, int64_t ldq, int64_t qbs, const void* k, int64_t ldk, int64_t kbs, const void* v,
                                    int64_t ldv, int64_t vbs, const void* attn, const void* dattn, int ld, float scale, float drop_p,
                                    const uint64_t* seed_base, uint64_t seed_off, void* dq, int64_t lddq, int64_t dqbs, void* dk_out,
                                    int64_t lddk, int64_t dkbs, void* dv, int64_t lddv, int64_t dvbs, void* stream) {
  S2S_REQUIRE(s2svc_attn_block_supported(S2S_BF16, T1, T2, D, H), "attn_block_bwd: unsupported shape (bf16, T <= 64, (D, d_k) in {(256,64),(384,96)})");
  S2S_REQUIRE(mode == 0 || mode == 1, "attn_block_bwd: mode 0 (dropout mask) or 1 (LayerNorm backward)");
  S2S_REQUIRE(g && wo_t && q && k && v && attn && dq && dk_out && dv && ld >= T2, "attn_block_bwd: missing operand");
  S2S_REQUIRE(mode == 0 || (s && mean && rstd && gamma && ds && al16(s) && al16(gamma) && al16(ds) && (!ds_extra || al16(ds_extra))),
              "attn_block_bwd: LayerNorm-backward prologue needs s / mean / rstd / gamma / ds (16-byte aligned)");
  S2S_REQUIRE(al16(g) && al16(wo_t) && al16(q) && al16(k) && al16(v) && al16(dq) && al16(dk_out) && al16(dv) && (!da || al16(da)),
              "attn_block_bwd: 16-byte aligned operands");
  S2S_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && qbs % 8 == 0 && kbs % 8 == 0 && vbs % 8 == 0 && lddq % 8 == 0 &&
              lddk % 8 == 0 && lddv % 8 == 0 && dqbs % 8 == 0 && dkbs % 8 == 0 && dvbs % 8 == 0,
              "attn_block_bwd: strides must be multiples of 8 elements");
  if (B == 0) return 0;
  ab_bwd_args a;
  a.H = H; a.T1 = T1; a.T2 = T2; a.mode = mode;
  a.g = (const bf16_t*)g; a.s = (const bf16_t*)s; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.ds_extra = (const bf16_t*)ds_extra;
  a.ds = (bf16_t*)ds; a.da = (bf16_t*)da; a.p_res = p_res; a.hscale = hscale; a.seed_res_base = seed_res_base; a.seed_res_off = seed_res_off;
  a.wo_t = (const bf16_t*)wo_t;
  a.q = (const bf16_t*)q; a.ldq = ldq; a.qbs = qbs; a.k = (const bf16_t*)k; a.ldk = ldk; a.kbs = kbs; a.v = (const bf16_t*)v; a.ldv = ldv; a.vbs = vbs;
  a.attn = (const bf16_t*)attn; a.dattn = (const bf16_t*)dattn; a.ld = ld; a.scale = scale; a.p = drop_p; a.seed_base = seed_base; a.seed_off = seed_off;
  a.dq = (bf16_t*)dq; a.lddq = lddq; a.dqbs = dqbs; a.dk = (bf16_t*)dk_out; a.lddk = lddk; a.dkbs = dkbs; a.dv = (bf16_t*)dv; a.lddv = lddv; a.dvbs = dvbs;
  hipStream_t st = (hipStream_t)stream;
  if (D == 256) return launch_bwd<64, 256>(B, a, st);
  return launch_bwd<96, 384>(B, a, st);
}
