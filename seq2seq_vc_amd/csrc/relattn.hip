// Relative-position self-attention (Transformer-XL style, the "new" rel_shift of ESPnet) without the (B, H, T, 2T-1) position
// term in memory -- bf16, T <= 256, d_k a multiple of 32.
//     score[i, j] = ((q_i + pos_bias_u) . k_j + (q_i + pos_bias_v) . pos_{T-1-i+j}) / sqrt(d_k)
// reference: modules/transformer/attention.py:237-260 (rel_shift), :262-305 (RelPositionMultiHeadedAttention.forward).
// The separate kernels run this as: add_head_bias -> GEMM (ac, fp32 (B,H,T,T)) -> GEMM (bd, fp32 (B,H,T,2T-1)) -> softmax
// kernel with the shift as index arithmetic (forward), and GEMM (dP) -> zero fill -> softmax backward with a scatter into dbd
// (backward).  Here:
//   relattn_fwd_kernel   q, k, pos, u, v -> attention map (+ its dropped copy) and qu = q + u, qv = q + v for the backward GEMMs
// (A fused backward kernel -- dP, softmax backward and the un-shifted dbd rows in one launch per 64-row block -- existed through round 5
// as an opt-in: correct, and slower than the batched GEMM + softmax-backward kernels it replaced, 30 / 46 vs 26 / 33 us at d_k = 192 / 768;
// removed in round 6, profiles/AB_LOG.md.)
// One workgroup per (utterance, head, block of 64 query rows); wave w owns the key columns 64 w .. 64 w + 63 of all 64 rows
// (16 accumulator tiles of 16 x 16).  The position term of a row block needs the 319 position rows c0 .. c0 + 318,
// c0 = T - 64 - i0; wave w multiplies its rows with 8 of the 20 tiles of that window and the shift
//     bd[il][63 - il + j]      (il = row inside the block)
// is, in the MFMA accumulator layout (row = 4 lg + r, column = lr of a 16 x 16 tile), a ROTATION of the 16 columns inside a
// lane group by 15 - 4 lg - r, with the wrapped lanes taking the next tile: one ds_bpermute per accumulator register, no trip
// through memory.  Operands (k, the position window, qu, qv) are streamed in 32-wide slices of d_k by LDS-DMA
// (`global_load_lds_dwordx4`, swizzled lane-linear image as in gemm_glds.hip) through three LDS stages with counted
// `s_waitcnt vmcnt(n)` + one raw barrier per slice: two slices are in flight while one is multiplied.  (A first version staged through
// registers: the compiler waits for ALL outstanding loads at the top of the loop, so every slice paid a trip to L2 -- 2 us per slice.)
// The probabilities leave through an LDS image of the tile in 16-byte stores.
// Dropout masks are functions of (seed, element index of the attention map), the same function the separate kernels use.
#include "common.h"
#include "../../include/s2svc_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace {

constexpr int SP = 264;       // pitch of the [64][256] bf16 staging tiles
constexpr float NEG = -3.4028234663852886e38f;

__device__ __forceinline__ float grp16_max(float v) {
  return row16_max(v);          // DPP rotations, no LDS crossbar (common.h)
}
__device__ __forceinline__ float grp16_sum(float v) {
  return row16_sum(v);
}

struct ra_fwd_args {
  int H, T, dk, L, ld;
  const bf16_t* q; int64_t ldq, qbs;
  const bf16_t* k; int64_t ldk, kbs;
  const bf16_t* pos; int64_t ldp;
  const float* u; const float* v;
  const int32_t* klen;
  float scale, p;
  const uint64_t* seed_base; uint64_t seed_off;
  bf16_t* attn; bf16_t* pdrop;
  bf16_t* qu; bf16_t* qv;       // (B, T, H * dk) contiguous
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS image of a 32-wide operand slice, as the LDS-DMA instruction writes it (lane-linear, 64-byte rows) with the XOR swizzle of
// gemm_glds.hip applied to the per-lane SOURCE address and to the fragment read: piece c of row r lives at r*64 + ((c ^ swz(r)) << 4).
__device__ __forceinline__ int swz32(int r) { return (-(r >> 2)) & 3; }
__device__ __forceinline__ int frag_off(int r, int c) { return r * 64 + ((c ^ swz32(r)) << 4); }

// Workgroups are handed to the 8 XCDs round-robin in linear id order, and each XCD has its own L2.  The row blocks of one
// (utterance, head) stream the SAME k / v rows (and position window): map them to one XCD so that its L2 fetches them once
// (identity mapping: the 4 row blocks of a pair land on 4 XCDs and every one of them pulls the pair's operands over the fabric).
__device__ __forceinline__ void block_of(int& rb, int& pair) {
  const int nrb = gridDim.x, npairs = gridDim.y, total = nrb * npairs;
  const int n = blockIdx.x + nrb * blockIdx.y;
  rb = blockIdx.x;
  pair = blockIdx.y;
  if ((total & 7) == 0 && ((total >> 3) % nrb) == 0) {
    const int xcd = n & 7, slot = n >> 3;
    pair = (slot / nrb) * 8 + xcd;
    rb = slot - (slot / nrb) * nrb;
  }
}

constexpr int ST_K = 0, ST_P = 256 * 64, ST_QU = ST_P + 320 * 64, ST_QV = ST_QU + 64 * 64, ST_BYTES = ST_QV + 64 * 64;   // 45,056 B
constexpr int FWD_STAGES = 3;

__global__ __launch_bounds__(256) void relattn_fwd_kernel(const ra_fwd_args a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int T = a.T, dk = a.dk, H = a.H;
  int rb_, bh;
  block_of(rb_, bh);
  const int i0 = rb_ * 64;
  const int b = bh / H, h = bh % H;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, lr = lane & 15, lg = lane >> 4;
  const int c0 = T - 64 - i0;
  const int D = H * dk;
  const bf16_t* kg = a.k + (int64_t)b * a.kbs + h * dk;
  const bf16_t* qg = a.q + (int64_t)b * a.qbs + h * dk;
  const bf16_t* pg = a.pos + h * dk;
  const int nsteps = dk / 32;
  // ---- qu = q + pos_bias_u, qv = q + pos_bias_v for this block's rows: written once (the backward GEMMs read them too) and then
  //      streamed back as MFMA operands like k and pos
  {
    const int vpr = dk >> 3;                          // 16-byte vectors per row
#pragma unroll 2
    for (int idx = t; idx < 64 * vpr; idx += 256) {
      const int row = idx / vpr, c8 = idx - row * vpr;
      const int i = i0 + row;
      if (i < T) {
        float f[8], fu[8], fv[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(qg + (int64_t)i * a.ldq + c8 * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { fu[e] = f[e] + a.u[h * dk + c8 * 8 + e]; fv[e] = f[e] + a.v[h * dk + c8 * 8 + e]; }
        const int64_t o = ((int64_t)b * T + i) * D + h * dk + c8 * 8;
        *reinterpret_cast<uint4*>(a.qu + o) = pack_bf16x8(fu);
        *reinterpret_cast<uint4*>(a.qv + o) = pack_bf16x8(fv);
      }
    }
  }
  __syncthreads();                                    // the stores are acknowledged (vmcnt) before any wave streams them back
  // ---- LDS-DMA pipeline: three stages, two slices in flight, counted waits.  Rows outside the utterance / the position table
  //      are CLAMPED, not zeroed: keys j >= T are masked, position rows outside [0, L) only meet scores with j < 0 or j >= T,
  //      query rows >= T are never stored.  Every wave issues exactly 11 DMA instructions per slice (4 k, 5 pos, qu, qv).
  const int wu = __builtin_amdgcn_readfirstlane(w);   // scalar: the LDS destinations (M0) stay on the SALU
  const char* kbase = reinterpret_cast<const char*>(kg);
  const char* pbase = reinterpret_cast<const char*>(pg);
  const char* ubase = reinterpret_cast<const char*>(a.qu + (int64_t)b * T * D + h * dk);
  const char* vbase = reinterpret_cast<const char*>(a.qv + (int64_t)b * T * D + h * dk);
  int64_t koff[4], poff[5], qoff;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (w * 4 + i) * 16 + (lane >> 2);
    koff[i] = ((int64_t)min(r, T - 1) * a.ldk + (((lane & 3) ^ swz32(r)) << 3)) * 2;
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int r = (w * 5 + i) * 16 + (lane >> 2);
    const int c = min(max(c0 + r, 0), a.L - 1);
    poff[i] = ((int64_t)c * a.ldp + (((lane & 3) ^ swz32(r)) << 3)) * 2;
  }
  {
    const int r = w * 16 + (lane >> 2);
    qoff = ((int64_t)min(i0 + r, T - 1) * D + (((lane & 3) ^ swz32(r)) << 3)) * 2;
  }
  auto issue = [&](int s, int stage) {
    unsigned char* st = smem + stage * ST_BYTES;
    const int64_t dB = (int64_t)s * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void*)(kbase + koff[i] + dB), (lds_void*)(st + ST_K + (wu * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 5; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void*)(pbase + poff[i] + dB), (lds_void*)(st + ST_P + (wu * 5 + i) * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(ubase + qoff + dB), (lds_void*)(st + ST_QU + wu * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gbl_void*)(vbase + qoff + dB), (lds_void*)(st + ST_QV + wu * 1024), 16, 0, 0);
  };

  f32x4_t ac[4][4], bd[4][5];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) ac[rt][jt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tn = 0; tn < 5; ++tn) bd[rt][tn] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  }
  issue(0, 0);
  issue(nsteps > 1 ? 1 : 0, 1);                       // slices past the end are re-issued copies: the counted waits stay exact
  int cur = 0;
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    wait_vmcnt<11>();                                 // slice s has landed (this wave's part); slice s + 1 may still be moving
    __builtin_amdgcn_s_barrier();                     // ... for every wave; and everyone is done reading slice s - 1
    {
      int nxt = cur + 2;
      if (nxt >= FWD_STAGES) nxt -= FWD_STAGES;
      issue(s + 2 < nsteps ? s + 2 : nsteps - 1, nxt);
    }
    const unsigned char* st = smem + cur * ST_BYTES;
    bf16x8_t qa[4], qb[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      qa[rt] = *reinterpret_cast<const bf16x8_t*>(st + ST_QU + frag_off(rt * 16 + lr, lg));
      qb[rt] = *reinterpret_cast<const bf16x8_t*>(st + ST_QV + frag_off(rt * 16 + lr, lg));
    }
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const bf16x8_t kb = *reinterpret_cast<const bf16x8_t*>(st + ST_K + frag_off(64 * w + jt * 16 + lr, lg));
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) ac[rt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[rt], kb, ac[rt][jt], 0, 0, 0);
    }
#pragma unroll
    for (int uu = 0; uu < 8; ++uu) {
      const bf16x8_t pb = *reinterpret_cast<const bf16x8_t*>(st + ST_P + frag_off((4 * w + uu) * 16 + lr, lg));
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const int tn = uu - 3 + rt;
        if (tn >= 0 && tn < 5) bd[rt][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb[rt], pb, bd[rt][tn], 0, 0, 0);
      }
    }
    cur = cur + 1 == FWD_STAGES ? 0 : cur + 1;
  }
  wait_vmcnt<0>();                                   // the copies issued past the end
  __syncthreads();                                   // operand slices are dead: LDS becomes the score tile
  // ---- shift, scale, mask -> fp32 score tile S[64][SF] in LDS.  Row il of S is later overwritten, by the wave that owns it,
  //      with the bf16 rows of the map (first half of the row's bytes) and of its dropped copy (second half): SF * 4 = 2 * SP * 2.
  constexpr int SF = SP;
  float* S = reinterpret_cast<float*>(smem);
  const int kl0 = a.klen ? a.klen[b] : T;
  const int kl = kl0 < T ? kl0 : T;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int off = 15 - 4 * lg - r;
      const int srcl = (lane & 48) | ((lr + off) & 15);
      const bool wrap = lr + off >= 16;
      float sh[5];
#pragma unroll
      for (int tn = 0; tn < 5; ++tn) sh[tn] = __shfl(bd[rt][tn][r], srcl, 64);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        const int j = 64 * w + 16 * jt + lr;
        const float val = (ac[rt][jt][r] + (wrap ? sh[jt + 1] : sh[jt])) * a.scale;
        S[(rt * 16 + 4 * lg + r) * SF + j] = j < kl ? val : NEG;
      }
    }
  }
  __syncthreads();
  // ---- softmax, one wave per row (16 rows each, ROLLED: this kernel runs cold code), a lane owns columns lane + 64 c
  const uint64_t seed = (a.seed_base ? *a.seed_base : 0ull) + a.seed_off;
  const float inv_keep = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
#pragma unroll 1
  for (int rr = 0; rr < 16; ++rr) {
    const int il = w * 16 + rr;
    float val[4];
    float mx = NEG;
#pragma unroll
    for (int c = 0; c < 4; ++c) { val[c] = S[il * SF + lane + 64 * c]; mx = fmaxf(mx, val[c]); }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) { val[c] = expf(val[c] - mx); sum += val[c]; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    const int64_t arow = ((int64_t)bh * T + i0 + il) * a.ld;
    bf16_t* rowA = reinterpret_cast<bf16_t*>(S + il * SF);     // LDS operations of one wave execute in order: the row was read above
    bf16_t* rowD = rowA + SP;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = lane + 64 * c;
      const float pr = j < kl ? val[c] * inv : 0.f;               // masked_fill(mask, 0.0) after the softmax
      const bf16_t pb = f2bf(pr);
      rowA[j] = pb;
      if (a.pdrop) {
        float pd = bf2f(pb);
        if (a.p > 0.f && j < a.ld) pd *= dropout_scale(seed, (uint64_t)(arow + j), a.p, inv_keep);
        rowD[j] = f2bf(pd);
      }
    }
  }
  __syncthreads();
  const int nv = a.ld >> 3;                           // 16-byte vectors per row of the map
  for (int n = t; n < 64 * 32; n += 256) {
    const int row = n >> 5, c8 = n & 31;
    if (i0 + row < T && c8 < nv) {
      const int64_t o = ((int64_t)bh * T + i0 + row) * a.ld + c8 * 8;
      const bf16_t* rowA = reinterpret_cast<const bf16_t*>(S + row * SF);
      *reinterpret_cast<uint4*>(a.attn + o) = *reinterpret_cast<const uint4*>(rowA + c8 * 8);
      if (a.pdrop) *reinterpret_cast<uint4*>(a.pdrop + o) = *reinterpret_cast<const uint4*>(rowA + SP + c8 * 8);
    }
  }
}

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

extern "C" int s2svc_relattn_supported(int dtype, int T, int dk, int rel_mode) {
  return (dtype == S2S_BF16 && rel_mode == 1 && T >= 1 && T <= 256 && dk >= 32 && dk % 32 == 0) ? 1 : 0;
}

// q, k: (B, T, .) views with row stride ldq / ldk and batch stride qbs / kbs (elements), head h at columns h * dk; pos (L = 2T - 1, .)
// with row stride ldp; u, v (H * dk) fp32; klen (B) int32 or NULL; attn / pdrop (B, H, T, ld) bf16, ld = T rounded up to 8 (pdrop
// NULL when drop_p == 0); qu, qv (B, T, H * dk) bf16 contiguous.
extern "C" int s2svc_relattn_fwd(int B, int H, int T, int dk, const void* q, int64_t ldq, int64_t qbs, const void* k, int64_t ldk,
                                 int64_t kbs, const void* pos, int64_t ldp, int L, const float* u, const float* v, const int32_t* klen,
                                 float scale, float drop_p, const uint64_t* seed_base, uint64_t seed_off, void* attn, void* pdrop, int ld,
                                 void* qu, void* qv, void* stream) {
  S2S_REQUIRE(s2svc_relattn_supported(S2S_BF16, T, dk, 1) && L == 2 * T - 1, "relattn_fwd: bf16, T <= 256, d_k % 32 == 0, L = 2T - 1");
  S2S_REQUIRE(q && k && pos && u && v && attn && qu && qv && (drop_p <= 0.f || pdrop) && ld >= T && ld % 8 == 0 && ld <= 256,
              "relattn_fwd: bad args");
  S2S_REQUIRE(ldq % 8 == 0 && qbs % 8 == 0 && ldk % 8 == 0 && kbs % 8 == 0 && ldp % 8 == 0 && al16(q) && al16(k) && al16(pos) &&
              al16(attn) && al16(pdrop) && al16(qu) && al16(qv), "relattn_fwd: 16-byte aligned operands, strides multiples of 8");
  if (B == 0) return 0;
  ra_fwd_args a;
  a.H = H; a.T = T; a.dk = dk; a.L = L; a.ld = ld;
  a.q = (const bf16_t*)q; a.ldq = ldq; a.qbs = qbs; a.k = (const bf16_t*)k; a.ldk = ldk; a.kbs = kbs;
  a.pos = (const bf16_t*)pos; a.ldp = ldp; a.u = u; a.v = v; a.klen = klen; a.scale = scale; a.p = drop_p;
  a.seed_base = seed_base; a.seed_off = seed_off; a.attn = (bf16_t*)attn; a.pdrop = (bf16_t*)pdrop; a.qu = (bf16_t*)qu; a.qv = (bf16_t*)qv;
  const size_t lds = (size_t)FWD_STAGES * ST_BYTES;                 // three operand stages (135,168 B) >= the score tile (67,584 B)
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(relattn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      s2svc_set_error("relattn_fwd: cannot raise the dynamic LDS limit");
      return -2;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(relattn_fwd_kernel, dim3((T + 63) / 64, B * H), dim3(256), lds, (hipStream_t)stream, a);
  S2S_CHECK_LAUNCH("relattn_fwd_kernel");
  return 0;
}

