// Fused multi-head attention for short sequences (T1, T2 <= 64), bf16: one workgroup per (utterance, head).
//
// reference: modules/transformer/attention.py:63-111 -- scores = Q K^T / sqrt(d_k); masked_fill(min); softmax;
// masked_fill(0) (stored as self.attn); dropout; @ V.  The unfused path runs this as 3 batched GEMMs + a softmax
// kernel forward and 4 GEMMs + a softmax kernel backward per attention block; at VTN's shapes (T = 63/64, d_k = 96,
// 128 (b, h) pairs) every one of those is a ~7 us launch-bound kernel.  Here forward and backward are ONE launch
// each: K / V (and their transposes, where an MFMA operand needs the other index contiguous) are staged in LDS
// once per workgroup, each wavefront owns 16 query rows, softmax runs on the MFMA accumulator layout with 16-lane
// shuffles, P goes to HBM only as the API-visible attention map (pre-dropout, bf16, rows padded to `ld`).
// Dropout masks are functions of (seed, element index in the attention map) -- the backward regenerates them.
#include "common.h"
#include "../../include/s2svc_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace {

constexpr int TP = 72;                       // pitch (elements) of the [*][64] tiles: 144-byte rows, conflict-free b128 reads
constexpr float NEG = -3.4028234663852886e38f;

__device__ __forceinline__ float grp16_max(float v) {
  return row16_max(v);          // DPP rotations, no LDS crossbar (common.h)
}
__device__ __forceinline__ float grp16_sum(float v) {
  return row16_sum(v);
}
__device__ __forceinline__ bf16x8_t zero8() { return (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0}; }
// component r (wave-uniform, from a ROLLED loop) of an accumulator: these kernels run once per launch on a cold instruction
// cache with one workgroup per CU, so code size is time (~0.45 us per KB executed, tools/kernel_code_sizes.py) -- the row
// loops of the softmax and the three output GEMMs of the backward are rolled, not unrolled
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
__device__ __forceinline__ uint32_t comp4(const u32x4_t& v, int r) { return r == 0 ? v[0] : r == 1 ? v[1] : r == 2 ? v[2] : v[3]; }
__device__ __forceinline__ float comp(const f32x4_t& v, int r) { return r == 0 ? v[0] : r == 1 ? v[1] : r == 2 ? v[2] : v[3]; }

// Operand staging, split into "issue every global load" / "write LDS".  load_rows / put_rows: rows [0, R) x DK of a (rows, ld)
// matrix -> LDS row-major with pitch DK + 8 (zero rows past R); load_rows_t / put_rows_t: the same rows transposed,
// lds[d * TP + row] (zero columns past R).  A workgroup of these kernels is alone on its CU
// with cold caches, and a rolled load -> LDS-store loop pays one memory round trip (~1-2 us) per iteration and per matrix
// (12 in the backward).  With the pieces of ALL matrices in registers before the first LDS store the prologue is one round trip.
template <int DK>
__device__ __forceinline__ void load_rows(const bf16_t* g, int64_t ld, int R, uint4 (&r)[DK / 32]) {
  constexpr int ppr = DK / 8;
#pragma unroll
  for (int i = 0; i < DK / 32; ++i) {
    const int p = threadIdx.x + 256 * i;
    const int row = p / ppr, c = p - row * ppr;
    r[i] = make_uint4(0, 0, 0, 0);
    if (row < R) r[i] = *reinterpret_cast<const uint4*>(g + (int64_t)row * ld + c * 8);
  }
}
template <int DK>
__device__ __forceinline__ void load_rows_t(const bf16_t* g, int64_t ld, int R, uint4 (&r)[DK / 32]) {
#pragma unroll
  for (int i = 0; i < DK / 32; ++i) {
    const int p = threadIdx.x + 256 * i;
    const int row = p & 63, c = p >> 6;
    r[i] = make_uint4(0, 0, 0, 0);
    if (row < R) r[i] = *reinterpret_cast<const uint4*>(g + (int64_t)row * ld + c * 8);
  }
}
template <int DK>
__device__ __forceinline__ void put_rows(const uint4 (&r)[DK / 32], bf16_t* lds) {
  constexpr int ppr = DK / 8;
#pragma unroll
  for (int i = 0; i < DK / 32; ++i) {
    const int p = threadIdx.x + 256 * i;
    const int row = p / ppr, c = p - row * ppr;
    *reinterpret_cast<uint4*>(lds + row * (DK + 8) + c * 8) = r[i];
  }
}
template <int DK>
__device__ __forceinline__ void put_rows_t(const uint4 (&r)[DK / 32], bf16_t* lds) {
#pragma unroll
  for (int i = 0; i < DK / 32; ++i) {
    const int p = threadIdx.x + 256 * i;
    const int row = p & 63, c = p >> 6;          // consecutive threads -> consecutive rows: conflict-free 2-byte LDS writes
    const uint32_t w[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      lds[(c * 8 + 2 * e) * TP + row] = (bf16_t)(w[e] & 0xffffu);
      lds[(c * 8 + 2 * e + 1) * TP + row] = (bf16_t)(w[e] >> 16);
    }
  }
}

// A wave's 16 x DK accumulator tile -> global rows with 16-byte stores: the MFMA layout gives a lane one column of 4
// rows per fragment (2-byte scattered stores, 24 per lane at DK = 96); staged through a wave-private LDS tile (pitch KP)
// every lane writes 8 consecutive columns at once.  LDS ops of one wavefront execute in order: no barrier inside.
template <int DK>
__device__ __forceinline__ void store_tile_rows(const f32x4_t (&acc)[DK / 16], bf16_t* stage, bf16_t* dst, int64_t ld, int rows_valid,
                                                bool vec_ok) {
  constexpr int KP = DK + 8;
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
  if (!vec_ok) {
#pragma unroll
    for (int dn = 0; dn < DK / 16; ++dn)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (lg * 4 + r < rows_valid) dst[(int64_t)(lg * 4 + r) * ld + dn * 16 + lr] = f2bf(acc[dn][r]);
    return;
  }
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

template <int DK>
__global__ __launch_bounds__(256) void attn_fused_fwd_kernel(int H, int T1, int T2, const bf16_t* __restrict__ q, int64_t ldq, int64_t qbs,
                                                             const bf16_t* __restrict__ k, int64_t ldk, int64_t kbs,
                                                             const bf16_t* __restrict__ v, int64_t ldv, int64_t vbs,
                                                             const int32_t* __restrict__ klen, int causal, float scale, float p,
                                                             const uint64_t* seed_base, uint64_t seed_off, bf16_t* __restrict__ attn,
                                                             int ld, bf16_t* __restrict__ out, int64_t ldo, int64_t obs) {
  constexpr int KP = DK + 8;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * KP];
  __shared__ __attribute__((aligned(16))) bf16_t Vt[DK * TP];
  __shared__ __attribute__((aligned(16))) bf16_t Pw[4][16 * TP];
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const int kl = klen ? (klen[b] < T2 ? klen[b] : T2) : T2;
  uint4 rk[DK / 32], rv[DK / 32];
  load_rows<DK>(k + (int64_t)b * kbs + h * DK, ldk, T2, rk);
  load_rows_t<DK>(v + (int64_t)b * vbs + h * DK, ldv, T2, rv);
  // Q fragments of this wave's 16 rows, straight from global
  const int qi = wave * 16 + lr;
  bf16x8_t qa[DK / 32];
#pragma unroll
  for (int ks = 0; ks < DK / 32; ++ks)
    qa[ks] = qi < T1 ? *reinterpret_cast<const bf16x8_t*>(q + (int64_t)b * qbs + (int64_t)qi * ldq + h * DK + ks * 32 + lg * 8) : zero8();
  put_rows<DK>(rk, Ks);
  put_rows_t<DK>(rv, Vt);
  __syncthreads();
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
  // softmax over the 64 keys of each of this lane's 4 rows (row = wave*16 + lg*4 + r ; column of s[jn][r] = jn*16 + lr)
  bf16_t* pw = Pw[wave];
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
    const int i = wave * 16 + lg * 4 + r;
    float val[4];
    float mx = NEG;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int j = jn * 16 + lr;
      const bool ok = j < kl && (!causal || j <= i);
      val[jn] = ok ? comp(s[jn], r) * scale : NEG;
      mx = fmaxf(mx, val[jn]);
    }
    mx = grp16_max(mx);
    float sum = 0.f, ex[4];
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {       // (each exponential once: it is needed for the sum and for the probability)
      ex[jn] = expf(val[jn] - mx);
      sum += ex[jn];
    }
    sum = grp16_sum(sum);
    const float inv = 1.f / sum;
    const int64_t arow = ((int64_t)(b * H + h) * T1 + i) * ld;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int j = jn * 16 + lr;
      const bool ok = j < kl && (!causal || j <= i);
      const float pr = ok ? ex[jn] * inv : 0.f;                      // masked_fill(mask, 0.0) after the softmax
      const bf16_t pb = f2bf(pr);
      if (i < T1 && j < ld) attn[arow + j] = pb;
      float pd = bf2f(pb);                                           // P.V consumes the stored (rounded) probabilities
      if (p > 0.f) pd *= dropout_scale(seed, (uint64_t)(arow + j), p, inv_keep);
      pw[(lg * 4 + r) * TP + j] = f2bf(pd);
    }
  }
  // O = Pdrop . V   (wave-private P tile: LDS ops of one wavefront execute in order, no barrier needed)
  f32x4_t o[DK / 16];
#pragma unroll
  for (int dn = 0; dn < DK / 16; ++dn) o[dn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    const bf16x8_t pa = *reinterpret_cast<const bf16x8_t*>(pw + lr * TP + ks * 32 + lg * 8);
#pragma unroll
    for (int dn = 0; dn < DK / 16; ++dn) {
      const bf16x8_t vb = *reinterpret_cast<const bf16x8_t*>(Vt + (dn * 16 + lr) * TP + ks * 32 + lg * 8);
      o[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, vb, o[dn], 0, 0, 0);
    }
  }
  // every wave is past its K reads: Ks becomes the output staging area (LDS-only barrier: __syncthreads() would also wait for
  // the attention-map stores above to be acknowledged)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  {
    bf16_t* base = out + (int64_t)b * obs + (int64_t)(wave * 16) * ldo + h * DK;
    const bool vec_ok = (ldo % 8 == 0) && (obs % 8 == 0) && (((uintptr_t)out) % 16 == 0);
    store_tile_rows<DK>(o, Ks + wave * 16 * KP, base, ldo, T1 - wave * 16, vec_ok);
  }
}

template <int DK>
__global__ __launch_bounds__(256) void attn_fused_bwd_kernel(int H, int T1, int T2, const bf16_t* __restrict__ q, int64_t ldq, int64_t qbs,
                                                             const bf16_t* __restrict__ k, int64_t ldk, int64_t kbs,
                                                             const bf16_t* __restrict__ v, int64_t ldv, int64_t vbs,
                                                             const bf16_t* __restrict__ dout, int64_t ldo, int64_t obs,
                                                             const bf16_t* __restrict__ attn, const bf16_t* __restrict__ dattn, int ld,
                                                             float scale, float p, const uint64_t* seed_base, uint64_t seed_off,
                                                             bf16_t* __restrict__ dq, int64_t lddq, int64_t dqbs,
                                                             bf16_t* __restrict__ dkk, int64_t lddk, int64_t dkbs,
                                                             bf16_t* __restrict__ dv, int64_t lddv, int64_t dvbs) {
  constexpr int KP = DK + 8;
  __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * KP];     // V rows      (B operand of dP = dO V^T)
  __shared__ __attribute__((aligned(16))) bf16_t Kt[DK * TP];     // K^T         (B operand of dQ = dS K)
  __shared__ __attribute__((aligned(16))) bf16_t Qt[DK * TP];     // Q^T         (B operand of dK = dS^T Q)
  __shared__ __attribute__((aligned(16))) bf16_t dOt[DK * TP];    // dO^T        (B operand of dV = P^T dO)
  __shared__ __attribute__((aligned(16))) bf16_t dSt[64 * TP];    // dS^T        (A operand of dK)
  __shared__ __attribute__((aligned(16))) bf16_t Pt[64 * TP];     // Pdrop^T     (A operand of dV)
  __shared__ __attribute__((aligned(16))) bf16_t dSw[4][16 * TP]; // dS rows of each wave (A operand of dQ)
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  const uint64_t seed = (seed_base ? *seed_base : 0ull) + seed_off;
  const float inv_keep = p > 0.f ? 1.f / (1.f - p) : 1.f;
  // every global read of the kernel is issued here, before the first LDS store (see load_rows): the four operand tiles, this
  // wave's dO fragments, and the lane's 16 attention-map / map-gradient elements of the softmax backward
  uint4 rv[DK / 32], rk[DK / 32], rq[DK / 32], rdo[DK / 32];
  load_rows<DK>(v + (int64_t)b * vbs + h * DK, ldv, T2, rv);
  load_rows_t<DK>(k + (int64_t)b * kbs + h * DK, ldk, T2, rk);
  load_rows_t<DK>(q + (int64_t)b * qbs + h * DK, ldq, T1, rq);
  load_rows_t<DK>(dout + (int64_t)b * obs + h * DK, ldo, T1, rdo);
  const int qi = wave * 16 + lr;
  bf16x8_t da[DK / 32];
#pragma unroll
  for (int ks = 0; ks < DK / 32; ++ks)
    da[ks] = qi < T1 ? *reinterpret_cast<const bf16x8_t*>(dout + (int64_t)b * obs + (int64_t)qi * ldo + h * DK + ks * 32 + lg * 8) : zero8();
  u32x4_t praw[4], graw[4];            // [jn][r]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = wave * 16 + lg * 4 + r;
    const int64_t arow = ((int64_t)(b * H + h) * T1 + i) * ld;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int j = jn * 16 + lr;
      const bool in = i < T1 && j < T2;
      praw[jn][r] = in ? attn[arow + j] : 0u;
      graw[jn][r] = (dattn && in) ? dattn[arow + j] : 0u;
    }
  }
  put_rows<DK>(rv, Vs);
  put_rows_t<DK>(rk, Kt);
  put_rows_t<DK>(rq, Qt);
  put_rows_t<DK>(rdo, dOt);
  __syncthreads();
  // dP = dO . V^T
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
  // dS = P * (dP*mask + dattn - rowdot) * scale ; Pdrop = P * mask      (accumulator layout: row lg*4+r, column jn*16+lr)
  bf16_t* dsw = dSw[wave];
#pragma unroll 1
  for (int r = 0; r < 4; ++r) {
    const int il = lg * 4 + r, i = wave * 16 + il;
    const int64_t arow = ((int64_t)(b * H + h) * T1 + i) * ld;
    float pv[4], t[4], m[4];
    float dot = 0.f;
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int j = jn * 16 + lr;
      const bool in = i < T1 && j < T2;
      pv[jn] = bf2f((bf16_t)(comp4(praw[jn], r)));
      m[jn] = (p > 0.f && in) ? dropout_scale(seed, (uint64_t)(arow + j), p, inv_keep) : 1.f;
      t[jn] = comp(dp[jn], r) * m[jn] + bf2f((bf16_t)(comp4(graw[jn], r)));
      dot += pv[jn] * t[jn];
    }
    dot = grp16_sum(dot);
#pragma unroll
    for (int jn = 0; jn < 4; ++jn) {
      const int j = jn * 16 + lr;
      const bf16_t ds = f2bf(pv[jn] * (t[jn] - dot) * scale);
      const bf16_t pd = f2bf(pv[jn] * m[jn]);
      dsw[il * TP + j] = ds;
      dSt[j * TP + i] = ds;
      Pt[j * TP + i] = pd;
    }
  }
  __syncthreads();
  // dQ = dS . K (rows: this wave's queries) ; dK = dS^T . Q ; dV = Pdrop^T . dO (rows: keys wave*16 .. wave*16+15): the same
  // 16 x 64 by 64 x DK product on three operand pairs -- ONE copy of the code, rolled
  // (Vs was last read before the barrier above: its rows wave*16.. are this wave's output staging area from here on)
#pragma unroll 1
  for (int which = 0; which < 3; ++which) {
    const bf16_t* At = which == 0 ? dsw : (which == 1 ? dSt : Pt) + wave * 16 * TP;
    const bf16_t* Bt = which == 0 ? Kt : which == 1 ? Qt : dOt;
    f32x4_t acc[DK / 16];
#pragma unroll
    for (int dn = 0; dn < DK / 16; ++dn) acc[dn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(At + lr * TP + ks * 32 + lg * 8);
#pragma unroll
      for (int dn = 0; dn < DK / 16; ++dn) {
        const bf16x8_t bb = *reinterpret_cast<const bf16x8_t*>(Bt + (dn * 16 + lr) * TP + ks * 32 + lg * 8);
        acc[dn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, acc[dn], 0, 0, 0);
      }
    }
    bf16_t* dst = which == 0 ? dq + (int64_t)b * dqbs : which == 1 ? dkk + (int64_t)b * dkbs : dv + (int64_t)b * dvbs;
    const int64_t ldd = which == 0 ? lddq : which == 1 ? lddk : lddv, dbs = which == 0 ? dqbs : which == 1 ? dkbs : dvbs;
    store_tile_rows<DK>(acc, Vs + wave * 16 * KP, dst + (int64_t)(wave * 16) * ldd + h * DK, ldd, (which == 0 ? T1 : T2) - wave * 16,
                        (ldd % 8 == 0) && (dbs % 8 == 0) && (((uintptr_t)dst) % 16 == 0));
  }
}

}  // namespace

extern "C" int s2svc_attn_fused_supported(int dtype, int T1, int T2, int dk) {
  return dtype == S2S_BF16 && T1 >= 1 && T1 <= 64 && T2 >= 1 && T2 <= 64 && (dk == 32 || dk == 64 || dk == 96 || dk == 128);
}

extern "C" int s2svc_attn_fused_fwd(int B, int H, int T1, int T2, int dk, const void* q, int64_t ldq, int64_t qbs, const void* k,
                                    int64_t ldk, int64_t kbs, const void* v, int64_t ldv, int64_t vbs, const int32_t* klen, int causal,
                                    float scale, float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* attn, int ld,
                                    void* out, int64_t ldo, int64_t obs, void* stream) {
  S2S_REQUIRE(s2svc_attn_fused_supported(S2S_BF16, T1, T2, dk), "attn_fused_fwd: unsupported shape (bf16, T <= 64, d_k in {32,64,96,128})");
  S2S_REQUIRE(ld >= T2 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && qbs % 8 == 0 && kbs % 8 == 0 && vbs % 8 == 0,
              "attn_fused_fwd: strides must be multiples of 8 elements");
  S2S_REQUIRE(((uintptr_t)q) % 16 == 0 && ((uintptr_t)k) % 16 == 0 && ((uintptr_t)v) % 16 == 0, "attn_fused_fwd: 16-byte aligned q/k/v");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
#define S2S_AF_FWD(DK)                                                                                                        \
  hipLaunchKernelGGL((attn_fused_fwd_kernel<DK>), dim3(B * H), dim3(256), 0, st, H, T1, T2, (const bf16_t*)q, ldq, qbs,        \
                     (const bf16_t*)k, ldk, kbs, (const bf16_t*)v, ldv, vbs, klen, causal, scale, drop_p, seed_base, seed_off, \
                     (bf16_t*)attn, ld, (bf16_t*)out, ldo, obs)
  if (dk == 32) S2S_AF_FWD(32); else if (dk == 64) S2S_AF_FWD(64); else if (dk == 96) S2S_AF_FWD(96); else S2S_AF_FWD(128);
#undef S2S_AF_FWD
  S2S_CHECK_LAUNCH("attn_fused_fwd_kernel");
  return 0;
}

extern "C" int s2svc_attn_fused_bwd(int B, int H, int T1, int T2, int dk, const void* q, int64_t ldq, int64_t qbs, const void* k,
                                    int64_t ldk, int64_t kbs, const void* v, int64_t ldv, int64_t vbs, const void* dout, int64_t ldo,
                                    int64_t obs, const void* attn, const void* dattn, int ld, float scale, float drop_p,
                                    const uint64_t* seed_base, uint64_t seed_off, void* dq, int64_t lddq, int64_t dqbs, void* dk_out,
                                    int64_t lddk, int64_t dkbs, void* dv, int64_t lddv, int64_t dvbs, void* stream) {
  S2S_REQUIRE(s2svc_attn_fused_supported(S2S_BF16, T1, T2, dk), "attn_fused_bwd: unsupported shape (bf16, T <= 64, d_k in {32,64,96,128})");
  S2S_REQUIRE(ld >= T2 && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && qbs % 8 == 0 && kbs % 8 == 0 && vbs % 8 == 0 &&
              obs % 8 == 0, "attn_fused_bwd: strides must be multiples of 8 elements");
  S2S_REQUIRE(((uintptr_t)q) % 16 == 0 && ((uintptr_t)k) % 16 == 0 && ((uintptr_t)v) % 16 == 0 && ((uintptr_t)dout) % 16 == 0,
              "attn_fused_bwd: 16-byte aligned q/k/v/dout");
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
#define S2S_AF_BWD(DK)                                                                                                          \
  hipLaunchKernelGGL((attn_fused_bwd_kernel<DK>), dim3(B * H), dim3(256), 0, st, H, T1, T2, (const bf16_t*)q, ldq, qbs,          \
                     (const bf16_t*)k, ldk, kbs, (const bf16_t*)v, ldv, vbs, (const bf16_t*)dout, ldo, obs, (const bf16_t*)attn, \
                     (const bf16_t*)dattn, ld, scale, drop_p, seed_base, seed_off, (bf16_t*)dq, lddq, dqbs, (bf16_t*)dk_out, lddk,  \
                     dkbs, (bf16_t*)dv, lddv, dvbs)
  if (dk == 32) S2S_AF_BWD(32); else if (dk == 64) S2S_AF_BWD(64); else if (dk == 96) S2S_AF_BWD(96); else S2S_AF_BWD(128);
#undef S2S_AF_BWD
  S2S_CHECK_LAUNCH("attn_fused_bwd_kernel");
  return 0;
}
