// Autoregressive decode step, round 6: fewer and shorter launches (56 -> 39 per step for VTN vc1).
//
// reference: models/vtn.py:344-389 (generation loop), modules/transformer/decoder.py:239-273 (forward_one_step),
// decoder_layer.py:85-132 (one position of a layer), pre_postnets.py:53-66 (Prenet), embedding.py:115-125 (positional encoding).
//
// A dependent launch of the captured step costs ~1.9 us before it does anything, and every DEPENDENT trip to memory inside a kernel
// ~1 us more (profiles/r05_decode_step_timeline.txt: the one-thread `decode_advance` takes 4.3 us: kernel arguments -> pos -> store).
// So the step is shortened two ways:
//   * fewer launches:  the attention kernel also multiplies its head's context with its COLUMN SLICE of the output projection and
//     leaves a (B, H, D) fp32 partial; the LayerNorm + projection kernel that consumes the sublayer (s2svc_decode_ln_linear_parts)
//     adds residual + bias + the H partials while it loads its rows -- the out-projection launch is gone (2 per layer);
//     Prenet (Linear-ReLU-dropout x 2) + input Linear + positional encoding are ONE launch (a workgroup per utterance);
//     feat_out | prob_out are one projection; the step counter advances in the emit kernel (last workgroup to finish).
//   * one trip per kernel:  decode_attn_proj_kernel requests EVERYTHING it will read -- query, this step's key / value, every cache
//     row up to the capacity, its slice of the projection weight, the step index -- before it looks at any of it (rows past the valid
//     length are masked afterwards instead of not being loaded: the loads then do not depend on `pos`).
// bf16 only (the timed path); fp32 parity runs keep the kernels of decode.hip.
#include "common.h"
#include "../../include/s2svc_hip.h"

namespace {

struct df_attn_args {
  int H, D, Tk;
  const bf16_t* q; int64_t ldq;
  bf16_t* kc; bf16_t* vc; int64_t ldt, cbs;
  const bf16_t* knew; const bf16_t* vnew; int64_t ldn;
  const int32_t* pos; const int32_t* klen;
  float scale;
  const bf16_t* wo; int64_t ldw;
  float* part;
  float* att; int64_t att_bs, att_hs, att_ps;
};

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

// rows of `npc` 16-byte pieces -> LDS, row-major and contiguous, by LDS-DMA: piece idx = it * 256 + thread goes to byte idx * 16 (one
// instruction of a wave fills 1 KB: lane-linear); pieces past `np` re-read the last one into the slack the region ends with.
__device__ __forceinline__ void dma_rows(const bf16_t* base, int64_t ld, int npc, int np, unsigned char* lds, int t) {
  const int iters = (np + 255) >> 8;
  const int wave0 = __builtin_amdgcn_readfirstlane(t & ~63);
  for (int it = 0; it < iters; ++it) {
    int idx = it * 256 + t;
    if (idx >= np) idx = np - 1;
    const int j = idx / npc, c = idx - j * npc;
    __builtin_amdgcn_global_load_lds((gbl_void*)(base + (int64_t)j * ld + c * 8), (lds_void*)(lds + ((it * 256 + wave0) << 4)), 16, 0, 0);
  }
}
__device__ __forceinline__ int round256(int n) { return (n + 255) & ~255; }

// One workgroup per (utterance, head).  Everything the kernel reads is requested before anything is looked at: q / this step's key and
// value / pos / klen by ordinary loads, the cache rows 0 .. Tk - 1 of K and V and the head's slice of the projection weight (D rows of
// dk) by LDS-DMA in rolled loops (no registers, ~2 KB of code: this kernel runs cold, see tools/kernel_code_sizes.py), one
// `s_waitcnt vmcnt(0)` -- ONE trip to memory.  Then, all out of LDS: scores (a thread per key), softmax (one wavefront), context
// (256 / DK threads per column over interleaved key ranges, summed in order), projection partial (a thread per output feature).
// Dynamic LDS: K | V | W images (pieces rounded up to 256) then floats: q | knew | vnew | scores | ctx | split partials | red.
template <int DK>
__global__ __launch_bounds__(256) void decode_attn_proj_kernel(const df_attn_args a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  constexpr int NPC = DK / 8;                    // 16-byte pieces per row
  const int Tk = a.Tk, D = a.D, H = a.H;
  const int Tkp = (Tk + 63) & ~63;
  const int NP = Tk * NPC, NW = D * NPC;
  unsigned char* Kl = smem;
  unsigned char* Vl = Kl + (round256(NP) << 4);
  unsigned char* Wl = Vl + (round256(NP) << 4);
  float* sq = reinterpret_cast<float*>(Wl + (round256(NW) << 4));
  float* sknew = sq + DK;
  float* svnew = sknew + DK;
  float* sc = svnew + DK;
  float* ctx = sc + Tkp;
  float* part2 = ctx + DK;
  float* red = part2 + 256;
  const int b = blockIdx.x / H, h = blockIdx.x % H, t = threadIdx.x;
  const bool self = a.knew != nullptr;
  // ---- every load of this kernel; none depends on another
  uint4 qraw = {0u, 0u, 0u, 0u}, knraw = {0u, 0u, 0u, 0u}, vnraw = {0u, 0u, 0u, 0u};
  if (t < NPC) {
    qraw = *reinterpret_cast<const uint4*>(a.q + (int64_t)b * a.ldq + h * DK + t * 8);
    if (self) {
      knraw = *reinterpret_cast<const uint4*>(a.knew + (int64_t)b * a.ldn + h * DK + t * 8);
      vnraw = *reinterpret_cast<const uint4*>(a.vnew + (int64_t)b * a.ldn + h * DK + t * 8);
    }
  }
  const int p = *a.pos;
  const int kl = a.klen ? a.klen[b] : Tk;
  dma_rows(a.kc + (int64_t)b * a.cbs + h * DK, a.ldt, NPC, NP, Kl, t);
  dma_rows(a.vc + (int64_t)b * a.cbs + h * DK, a.ldt, NPC, NP, Vl, t);
  dma_rows(a.wo + h * DK, a.ldw, NPC, NW, Wl, t);
  int n_keys = self ? p + 1 : kl;
  if (n_keys > Tk) n_keys = Tk;                   // capacity guard (the host never replays past the cache capacity)
  if (t < NPC) {
    float f[8];
    unpack_bf16x8(qraw, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) sq[t * 8 + e] = f[e];
    if (self) {
      unpack_bf16x8(knraw, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) sknew[t * 8 + e] = f[e];
      unpack_bf16x8(vnraw, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) svnew[t * 8 + e] = f[e];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (self && t < NPC && p < Tk) {                // this step's key / value: into the LDS images (and, at the end, into the caches)
    *reinterpret_cast<uint4*>(Kl + (((int64_t)p * NPC + t) << 4)) = knraw;
    *reinterpret_cast<uint4*>(Vl + (((int64_t)p * NPC + t) << 4)) = vnraw;
  }
  __syncthreads();
  // ---- scores: a thread per key
  for (int j = t; j < Tkp; j += 256) {
    float s = 0.f;
    if (j < n_keys) {
#pragma unroll 2
      for (int c = 0; c < NPC; ++c) {
        float f[8];
        unpack_bf16x8(*reinterpret_cast<const uint4*>(Kl + ((j * NPC + c) << 4)), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += sq[c * 8 + e] * f[e];
      }
      s *= a.scale;
    }
    sc[j] = s;
  }
  __syncthreads();
  if (t < 64) {                                    // softmax statistics by one wavefront (all 64 lanes active)
    float mx = -3.4028234663852886e38f;
    for (int j = t; j < n_keys; j += 64) mx = fmaxf(mx, sc[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = t; j < n_keys; j += 64) sum += __expf(sc[j] - mx);
    sum = wave_sum(sum);
    if (t == 0) { red[0] = mx; red[1] = 1.f / sum; }
  }
  __syncthreads();
  {
    const float mx = red[0], inv = red[1];
    float* arow = a.att ? a.att + (int64_t)b * a.att_bs + (int64_t)h * a.att_hs + (int64_t)p * a.att_ps : nullptr;
    for (int j = t; j < Tk; j += 256) {
      const float pr = j < n_keys ? __expf(sc[j] - mx) * inv : 0.f;
      sc[j] = pr;
      if (arow) arow[j] = pr;
    }
  }
  __syncthreads();
  // ---- context: S = 256 / DK threads per column, interleaved key ranges, summed in order
  constexpr int S = 256 / DK;
  {
    const int s = t / DK, d = t - s * DK;
    if (s < S) {
      const bf16_t* vcol = reinterpret_cast<const bf16_t*>(Vl) + d;
      float acc = 0.f;
      for (int j = s; j < n_keys; j += S) acc += sc[j] * bf2f(vcol[j * DK]);
      part2[s * DK + d] = acc;
    }
  }
  __syncthreads();
  if (t < DK) {
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) acc += part2[s * DK + t];
    ctx[t] = acc;
  }
  __syncthreads();
  // ---- this head's share of the output projection: part[b][h][n] = sum_d Wo[n][h DK + d] ctx[d]
  float* po = a.part + ((int64_t)b * H + h) * D;
  for (int n = t; n < D; n += 256) {
    float s = 0.f;
#pragma unroll 2
    for (int c = 0; c < NPC; ++c) {
      float f[8];
      unpack_bf16x8(*reinterpret_cast<const uint4*>(Wl + ((n * NPC + c) << 4)), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += ctx[c * 8 + e] * f[e];
    }
    po[n] = s;
  }
  // ---- self-attention: this step's key / value join the cache (read by the NEXT step's launch)
  if (self && t < NPC && p < Tk) {
    bf16_t* kw = a.kc + (int64_t)b * a.cbs + h * DK + (int64_t)p * a.ldt + t * 8;
    bf16_t* vw = a.vc + (int64_t)b * a.cbs + h * DK + (int64_t)p * a.ldt + t * 8;
    *reinterpret_cast<uint4*>(kw) = knraw;
    *reinterpret_cast<uint4*>(vw) = vnraw;
  }
}

// ---- Prenet + input Linear + positional encoding of one position: a workgroup per utterance ------------------------------------------
struct df_prenet_args {
  int nl;                                   // layers: nl - 1 x (Linear, ReLU, dropout) + the input Linear (no activation, no dropout)
  const void* w[4]; const float* bias[4];
  int N[4], K[4];
  float drop_p; uint64_t seed_off[4]; const uint64_t* seed_base;
  const void* x; int64_t ldx;               // (B, K[0]) previous output frame
  float xscale; const float* alpha; const float* pe; const int32_t* pos;
  void* y; int64_t ldy;                     // (B, N[nl-1])
};

template <typename T>
__global__ __launch_bounds__(256) void decode_prenet_kernel(const df_prenet_args a) {
  extern __shared__ float sm[];             // two row buffers of max(N, K) floats
  constexpr int VEC = sizeof(T) == 2 ? 8 : 4;
  const int b = blockIdx.x, t = threadIdx.x;
  int maxd = a.K[0];
  for (int l = 0; l < a.nl; ++l) maxd = a.N[l] > maxd ? a.N[l] : maxd;
  float* cur = sm;
  float* nxt = sm + maxd;
  const int p = *a.pos;
  const float al = a.alpha ? *a.alpha : 1.f;
  const uint64_t sb = a.seed_base ? *a.seed_base : 0ull;
  for (int k = t; k < a.K[0]; k += 256) cur[k] = ldf((const T*)a.x + (int64_t)b * a.ldx + k);
  __syncthreads();
  for (int l = 0; l < a.nl; ++l) {
    const int N = a.N[l], K = a.K[l];
    const T* W = (const T*)a.w[l];
    const bool last = l == a.nl - 1;
    for (int n = t; n < N; n += 256) {
      const T* wr = W + (int64_t)n * K;
      float acc = 0.f;
      for (int k0 = 0; k0 < K; k0 += VEC) {
        float f[VEC];
        if (VEC == 8) {
          const uint4 v = *reinterpret_cast<const uint4*>(wr + k0);
          float g[8];
          unpack_bf16x8(v, g);
#pragma unroll
          for (int e = 0; e < VEC; ++e) f[e] = g[e];
        } else {
          const float4 v = *reinterpret_cast<const float4*>(wr + k0);
          f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc += cur[k0 + e] * f[e];
      }
      if (a.bias[l]) acc += a.bias[l][n];
      if (!last) {
        acc = acc > 0.f ? acc : 0.f;
        if (a.drop_p > 0.f) acc *= dropout_scale(sb + a.seed_off[l], (uint64_t)((int64_t)b * N + n), a.drop_p, 1.f / (1.f - a.drop_p));
        if (sizeof(T) == 2) acc = bf2f(f2bf(acc));            // the storage rounding between two layers of the unfused path
        nxt[n] = acc;
      } else {
        if (sizeof(T) == 2) acc = bf2f(f2bf(acc));
        stf((T*)a.y + (int64_t)b * a.ldy + n, acc * a.xscale + al * a.pe[(int64_t)p * N + n]);
      }
    }
    __syncthreads();
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
}

// ---- emit + advance ------------------------------------------------------------------------------------------------------------------------
// decode_emit of decode.hip with a row stride for feat / logit (one packed feat_out | prob_out projection); the LAST workgroup to
// finish (a ticket counter in device memory, reset by its drawer) advances the step index and the dropout seed.
template <typename T>
__global__ __launch_bounds__(256) void decode_emit_advance_kernel(int r, int odim, const T* __restrict__ feat, const T* __restrict__ logit,
                                                                  int64_t ldf_, float threshold, const int32_t* __restrict__ minlen,
                                                                  const int32_t* __restrict__ maxlen, int32_t* __restrict__ pos,
                                                                  float* __restrict__ outs, int64_t outs_bs, float* __restrict__ probs,
                                                                  int64_t probs_bs, T* __restrict__ prev, int32_t* __restrict__ stop_at,
                                                                  uint64_t* __restrict__ seed_base, uint64_t seed_stride,
                                                                  uint32_t* __restrict__ ticket) {
  const int b = blockIdx.x;
  const int p = *pos;
  for (int i = threadIdx.x; i < r * odim; i += 256) {
    const float v = ldf(feat + (int64_t)b * ldf_ + i);
    outs[(int64_t)b * outs_bs + (int64_t)p * r * odim + i] = v;
    if (i >= (r - 1) * odim) stf(prev + (int64_t)b * odim + (i - (r - 1) * odim), v);
  }
  if (threadIdx.x == 0) {
    bool fire = false;
    for (int i = 0; i < r; ++i) {
      const float pr = 1.f / (1.f + expf(-ldf(logit + (int64_t)b * ldf_ + i)));
      probs[(int64_t)b * probs_bs + (int64_t)p * r + i] = pr;
      fire = fire || (pr >= threshold);
    }
    if ((fire || p + 1 >= maxlen[b]) && p + 1 >= minlen[b] && stop_at[b] == 0) stop_at[b] = p + 1;
  }
  __syncthreads();                                   // every thread of this workgroup has read *pos
  if (threadIdx.x == 0) {
    const uint32_t drawn = atomicAdd(ticket, 1u);    // (integer atomic: no order of summation to keep)
    if (drawn == gridDim.x - 1) {                    // every workgroup read *pos before it drew its ticket
      *ticket = 0u;
      *pos = p + 1;
      if (seed_base) *seed_base += seed_stride;
    }
  }
}

bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

namespace {
size_t df_attn_lds_bytes(int dk, int D, int Tk) {
  const int64_t np = (((int64_t)Tk * (dk / 8)) + 255) & ~255ll, nw = (((int64_t)D * (dk / 8)) + 255) & ~255ll;
  return (size_t)(16 * (2 * np + nw) + 4 * (4 * dk + ((Tk + 63) & ~63) + 256 + 2));
}
template <int DK>
int df_launch_attn(const df_attn_args& a, int B, size_t shm, hipStream_t st) {
  static size_t attr_set = 0;
  if (shm > 64 * 1024 && attr_set < shm) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(decode_attn_proj_kernel<DK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) {
      s2svc_set_error("decode_attn_proj: cannot raise the dynamic LDS limit");
      return -2;
    }
    attr_set = shm;
  }
  hipLaunchKernelGGL(decode_attn_proj_kernel<DK>, dim3(a.H * B), dim3(256), shm, st, a);
  return 0;
}
}  // namespace

extern "C" int s2svc_decode_attn_proj_supported(int dtype, int H, int dk, int D, int Tk) {
  if (dtype != S2S_BF16 || H <= 0 || D != H * dk || Tk <= 0 || Tk > 4096) return 0;
  if (dk != 32 && dk != 64 && dk != 96 && dk != 128) return 0;
  return df_attn_lds_bytes(dk, D, Tk) <= 156 * 1024 ? 1 : 0;       // K, V (all Tk rows) and the head's weight slice live in LDS
}

// One decoder position of one attention sublayer INCLUDING its share of the output projection:
//   part[b, h, :] = Wo[:, h dk : (h + 1) dk] . softmax(q_h . K_h^T * scale) V_h        (fp32, (B, H, D))
// q (B, .) rows (stride ldq), caches / projected memory (B, Tk, .) (time stride ldt, batch stride cbs); knew / vnew != NULL: self-attention,
// this step's key / value rows (stride ldn) are used at position *pos and appended to the caches; else keys 0 .. klen[b] - 1.
// The consumer adds residual + bias + the H partials (s2svc_decode_ln_linear_parts).  att: as s2svc_decode_attn.
extern "C" int s2svc_decode_attn_proj(int B, int H, int dk, const void* q, int64_t ldq, void* kcache, void* vcache, int64_t ldt, int64_t cbs,
                                      const void* knew, const void* vnew, int64_t ldn, const int32_t* pos, const int32_t* klen, int Tk,
                                      float scale, const void* wo, int64_t ldw, float* part, float* att, int64_t att_bs, int64_t att_hs,
                                      int64_t att_ps, void* stream) {
  const int D = H * dk;
  S2S_REQUIRE(B >= 0 && s2svc_decode_attn_proj_supported(S2S_BF16, H, dk, D, Tk), "decode_attn_proj: unsupported shape (see _supported)");
  S2S_REQUIRE(q && kcache && vcache && pos && wo && part, "decode_attn_proj: null argument");
  S2S_REQUIRE((knew == nullptr) == (vnew == nullptr), "decode_attn_proj: knew and vnew go together");
  S2S_REQUIRE(al16(q) && al16(kcache) && al16(vcache) && al16(wo) && al16(knew) && al16(vnew) && ldq % 8 == 0 && ldt % 8 == 0 && cbs % 8 == 0 &&
              ldn % 8 == 0 && ldw % 8 == 0, "decode_attn_proj: 16-byte aligned operands, strides multiples of 8");
  if (B == 0) return 0;
  df_attn_args a;
  a.H = H; a.D = D; a.Tk = Tk;
  a.q = (const bf16_t*)q; a.ldq = ldq; a.kc = (bf16_t*)kcache; a.vc = (bf16_t*)vcache; a.ldt = ldt; a.cbs = cbs;
  a.knew = (const bf16_t*)knew; a.vnew = (const bf16_t*)vnew; a.ldn = ldn; a.pos = pos; a.klen = klen; a.scale = scale;
  a.wo = (const bf16_t*)wo; a.ldw = ldw; a.part = part; a.att = att; a.att_bs = att_bs; a.att_hs = att_hs; a.att_ps = att_ps;
  const size_t shm = df_attn_lds_bytes(dk, D, Tk);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (dk == 32) rc = df_launch_attn<32>(a, B, shm, st);
  else if (dk == 64) rc = df_launch_attn<64>(a, B, shm, st);
  else if (dk == 96) rc = df_launch_attn<96>(a, B, shm, st);
  else rc = df_launch_attn<128>(a, B, shm, st);
  if (rc) return rc;
  S2S_CHECK_LAUNCH("decode_attn_proj_kernel");
  return 0;
}

// nl <= 4 layers: w[l] (N[l], K[l]) row-major in `dtype`, bias[l] fp32 or NULL, K[l + 1] == N[l]; layers 0 .. nl - 2 are
// Linear -> ReLU -> dropout(drop_p, seed *seed_base + seed_off[l], element index b N + n: the masks of s2svc_gemm's epilogue stage),
// the last one is the plain input Linear followed by y = v xscale + alpha pe[*pos]  (pre_postnets.py:53-66, embedding.py:115-125).
extern "C" int s2svc_decode_prenet(int dtype, int B, int nl, const void* const* w, const float* const* bias, const int32_t* N, const int32_t* K,
                                   float drop_p, const uint64_t* seed_base, const uint64_t* seed_off, const void* x, int64_t ldx,
                                   float xscale, const float* alpha, const float* pe, const int32_t* pos, void* y, int64_t ldy,
                                   void* stream) {
  S2S_REQUIRE(dtype == S2S_F32 || dtype == S2S_BF16, "decode_prenet: bad dtype");
  S2S_REQUIRE(B >= 0 && nl >= 1 && nl <= 4 && w && bias && N && K && x && pe && pos && y && drop_p >= 0.f && drop_p < 1.f, "decode_prenet: bad arguments");
  if (B == 0) return 0;
  const int vec = dtype == S2S_F32 ? 4 : 8;
  df_prenet_args a;
  a.nl = nl;
  int maxd = K[0];
  for (int l = 0; l < 4; ++l) { a.w[l] = nullptr; a.bias[l] = nullptr; a.N[l] = 0; a.K[l] = 0; a.seed_off[l] = 0; }
  for (int l = 0; l < nl; ++l) {
    S2S_REQUIRE(w[l] && N[l] > 0 && K[l] > 0 && K[l] % vec == 0 && al16(w[l]) && (l == 0 || K[l] == N[l - 1]), "decode_prenet: layer shapes (K % 16 bytes, chained)");
    a.w[l] = w[l]; a.bias[l] = bias[l]; a.N[l] = N[l]; a.K[l] = K[l]; a.seed_off[l] = seed_off ? seed_off[l] : 0;
    maxd = N[l] > maxd ? N[l] : maxd;
  }
  S2S_REQUIRE(maxd <= 4096, "decode_prenet: layer width above 4096");
  a.drop_p = drop_p; a.seed_base = seed_base; a.x = x; a.ldx = ldx; a.xscale = xscale; a.alpha = alpha; a.pe = pe; a.pos = pos; a.y = y; a.ldy = ldy;
  const size_t shm = (size_t)2 * maxd * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32) hipLaunchKernelGGL(decode_prenet_kernel<float>, dim3(B), dim3(256), shm, st, a);
  else hipLaunchKernelGGL(decode_prenet_kernel<bf16_t>, dim3(B), dim3(256), shm, st, a);
  S2S_CHECK_LAUNCH("decode_prenet_kernel");
  return 0;
}

// s2svc_decode_emit + s2svc_decode_advance in one launch: feat / logit rows have stride ldf (columns of one packed projection);
// `ticket`: one zero-initialised device uint32 owned by the caller (it is zero again when the kernel ends).
extern "C" int s2svc_decode_emit_advance(int dtype, int B, int r, int odim, const void* feat, const void* logit, int64_t ldf_, float threshold,
                                         const int32_t* minlen, const int32_t* maxlen, int32_t* pos, float* outs, int64_t outs_bs, float* probs,
                                         int64_t probs_bs, void* prev, int32_t* stop_at, uint64_t* seed_base, uint64_t seed_stride,
                                         uint32_t* ticket, void* stream) {
  S2S_REQUIRE(B > 0 && r > 0 && odim > 0 && feat && logit && pos && outs && probs && prev && stop_at && minlen && maxlen && ticket,
              "decode_emit_advance: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == S2S_F32)
    hipLaunchKernelGGL(decode_emit_advance_kernel<float>, dim3(B), dim3(256), 0, st, r, odim, (const float*)feat, (const float*)logit, ldf_, threshold,
                       minlen, maxlen, pos, outs, outs_bs, probs, probs_bs, (float*)prev, stop_at, seed_base, seed_stride, ticket);
  else
    hipLaunchKernelGGL(decode_emit_advance_kernel<bf16_t>, dim3(B), dim3(256), 0, st, r, odim, (const bf16_t*)feat, (const bf16_t*)logit, ldf_,
                       threshold, minlen, maxlen, pos, outs, outs_bs, probs, probs_bs, (bf16_t*)prev, stop_at, seed_base, seed_stride, ticket);
  S2S_CHECK_LAUNCH("decode_emit_advance_kernel");
  return 0;
}
