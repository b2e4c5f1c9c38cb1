// Attention of medium-length sequences around ONE score tile per launch -- bf16, T2 <= 512 keys, any T1, d_k a multiple of 32.
// reference: modules/transformer/attention.py:63-93 (scores, masked_fill with the dtype's minimum, softmax, masked_fill with 0,
// dropout), :95-111 (the context), and their backward; the backward of :237-303 (relative-position attention) up to dS / dbd.
//   MODE 0   attn[b, h, i, j] = softmax_j(q_i . k_j / sqrt(d_k) masked to j < klen[b] (and j <= i when causal)), pdrop = dropout(attn)
//            [+ DK2: ctx = pdrop . v]
//   MODE 1   dP = dctx . v^T on chip;  dS = P (dP mask + dattn - rowsum(P (dP mask + dattn))) scale
//            [+ dbd: dS un-shifted into the gradient of the position term]  [+ DK2: dq = dS . k]
// Round 6: the Transformer-TTS blocks (T up to 320 x 151: above the 64 frames of attn_fused.hip, no position term for relattn.hip) ran
// the forward as scores GEMM (fp32 (B, H, T1, T2) to HBM) + softmax kernel + context GEMM and the backward as dP GEMM + softmax
// backward + three products: dependent launches and 8-byte-per-element round trips on a chain that is bound by its launch count
// (C4 step 3.96 -> 3.40 ms with this file, profiles/AB_LOG.md round 6 item 12).  The kernel is relattn.hip's forward without the position
// term and with NW wavefronts (2: T2 <= 128, 4: T2 <= 256, 8: T2 <= 512): one workgroup per (utterance, head, block of 64 query rows),
// wave w owns the key columns 64 w .. 64 w + 63 (16 accumulator tiles); the two operands of the score product (q, k forward; dctx, v
// backward) are streamed in 32-wide slices of d_k by LDS-DMA through three stages with counted `s_waitcnt vmcnt` + one raw barrier per
// slice; the scores meet in an fp32 LDS tile, one wave per row does the softmax (or its backward), the bf16 rows leave in 16-byte stores.
// DK2 in {64, 96, 128}: the finished bf16 tile is then the A operand of a second product with the head's (keys x d_k) matrix, which was
// fetched into registers under the row pass and is laid over the dead tile for `ds_read_b64_tr_b16` (see below).
// Same output layout ((B, H, T1, ld), ld = T2 rounded up to 8, pad columns zero) and the same dropout masks (a function of the seed and
// the element index in that layout) as softmax.hip / attn_fused.hip: forward and backward may come from different kernel families.
#include "common.h"
#include "../../include/s2svc_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace {

constexpr float AM_NEG = -3.4028234663852886e38f;

struct am_args {
  int H, T1, T2, dk, ld;
  const bf16_t* q; int64_t ldq, qbs;
  const bf16_t* k; int64_t ldk, kbs;
  const int32_t* klen;
  int causal;
  float scale, p;
  const uint64_t* seed_base; uint64_t seed_off;
  bf16_t* attn; bf16_t* pdrop;
  // backward (MODE 1): q = d context, k = v; p_in the stored map, dattn the gradient that reached the map itself (or NULL), ds the output
  const bf16_t* p_in; const bf16_t* dattn; bf16_t* ds;
  // ... and, for relative-position attention (T1 == T2, the "new" rel_shift), the same values where the shift took them from:
  // dbd[b, h, i, T1 - 1 - i + j] = dS[b, h, i, j], every other element of the (B, H, T1, ldb) tensor zero
  bf16_t* dbd; int ldb;
  // second product (DK2 > 0): out2[i, h d_k + .] = tile[i, :] . m2[:, h d_k + .] with the tile this workgroup just finished -- forward:
  // the (dropped) map times v = the context; backward: dS times k = dq.  m2 (B, T2, .) view, out2 (B, T1, .) view
  const bf16_t* m2; int64_t ldm2, m2bs;
  bf16_t* out2; int64_t ldo2, o2bs;
};

typedef __attribute__((ext_vector_type(4))) uint32_t am_u32x4_t;
typedef __attribute__((ext_vector_type(4))) short am_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short am_s16x8_t;
typedef __attribute__((address_space(3))) am_s16x4_t am_lds_s16x4_t;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;
template <int N> __device__ __forceinline__ void am_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// LDS image of a 32-wide operand slice (64-byte rows, lane-linear as the LDS-DMA instruction writes it) with the XOR swizzle of
// gemm_glds.hip / relattn.hip on the per-lane SOURCE address and on the fragment read
__device__ __forceinline__ int am_swz32(int r) { return (-(r >> 2)) & 3; }
__device__ __forceinline__ int am_frag_off(int r, int c) { return r * 64 + ((c ^ am_swz32(r)) << 4); }

// the row blocks of one (utterance, head) stream the same k rows: keep them on one XCD (see relattn.hip)
__device__ __forceinline__ void am_block_of(int& rb, int& pair) {
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

template <int NW, int MODE, int DK2>
__global__ __launch_bounds__(64 * NW) void attn_map_kernel(const am_args a) {
  constexpr int NT = 64 * NW;                        // threads = key columns a workgroup covers
  constexpr int ST_K = 0, ST_Q = NT * 64, ST_BYTES = ST_Q + 64 * 64;
  // fp32 score rows of SP = NT + 4 floats: rows 4 apart (the lane groups of an accumulator store) sit 16 banks apart -- with NT + 8 they met
  // on the same banks (SQ_LDS_BANK_CONFLICT 27-36 % of the LDS cycles, profiles/r06_attn_map_pmc.txt).  The same bytes later hold the
  // row's two bf16 rows: the map at byte 0, its dropped copy at element RD = NT + 8 (16-byte aligned): 2 * (NT + 8) + 2 * NT = 4 * SP bytes
  constexpr int SP = NT + 4, RD = NT + 8;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int T1 = a.T1, T2 = a.T2, dk = a.dk, H = a.H;
  int rb_, bh;
  am_block_of(rb_, bh);
  const int i0 = rb_ * 64;
  const int b = bh / H, h = bh % H;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, lr = lane & 15, lg = lane >> 4;
  const char* kbase = reinterpret_cast<const char*>(a.k + (int64_t)b * a.kbs + h * dk);
  const char* qbase = reinterpret_cast<const char*>(a.q + (int64_t)b * a.qbs + h * dk);
  const int nsteps = dk / 32;
  const int wu = __builtin_amdgcn_readfirstlane(w);
  // rows outside the tensors are CLAMPED, not zeroed: keys j >= T2 are masked, query rows >= T1 are never stored.  Every wave issues
  // exactly 4 + QI DMA instructions per slice (4 x 16 key rows of its 64 columns, QI x 16 query rows: with 8 waves, waves w and
  // w + 4 fetch the same query rows into the same place, so that the counted waits are the same in every wave)
  constexpr int QI = NW >= 4 ? 1 : 4 / NW;            // query-row instructions per wave and slice
  int64_t koff[4], qoff[QI];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (w * 4 + i) * 16 + (lane >> 2);
    koff[i] = ((int64_t)min(r, T2 - 1) * a.ldk + (((lane & 3) ^ am_swz32(r)) << 3)) * 2;
  }
#pragma unroll
  for (int i = 0; i < QI; ++i) {
    const int r = ((w * QI + i) & 3) * 16 + (lane >> 2);
    qoff[i] = ((int64_t)min(i0 + r, T1 - 1) * a.ldq + (((lane & 3) ^ am_swz32(r)) << 3)) * 2;
  }
  auto issue = [&](int s, int stage) {
    unsigned char* st = smem + stage * ST_BYTES;
    const int64_t dB = (int64_t)s * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void*)(kbase + koff[i] + dB), (lds_void*)(st + ST_K + (wu * 4 + i) * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < QI; ++i)
      __builtin_amdgcn_global_load_lds((gbl_void*)(qbase + qoff[i] + dB), (lds_void*)(st + ST_Q + ((wu * QI + i) & 3) * 1024), 16, 0, 0);
  };
  // backward: the stored map (and the gradient on it) of the rows this wave finishes -- RW = 64 / NW rows x NW columns per lane, pairs of
  // adjacent bf16 -- are fetched a group of G rows AHEAD of their use: the first group NOW, so that its latency runs under the product,
  // group g + 1 while group g is computed (loading per row inside the row loop serialised RW memory latencies behind the product and
  // cost more than the launch this kernel saves; all rows in registers needs an unrolled row loop, 27 KB of cold code).  These loads
  // are older than every LDS-DMA below: the counted waits stay valid.
  constexpr int RW = 64 / NW, CP = NW / 2, G = 4, NG = RW / G;
  uint32_t pa[G][CP], da[G][CP];
  auto fetch = [&](int g, uint32_t (&xp)[G][CP], uint32_t (&xd)[G][CP]) {
#pragma unroll
    for (int rr = 0; rr < G; ++rr) {
      const int64_t arow = ((int64_t)bh * T1 + min(i0 + w * RW + g * G + rr, T1 - 1)) * a.ld;
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        const int j = min(2 * lane + 128 * c, a.ld - 2);          // (clamped, not branched: columns >= ld are zeroed at the use)
        xp[rr][c] = *reinterpret_cast<const uint32_t*>(a.p_in + arow + j);
        xd[rr][c] = a.dattn ? *reinterpret_cast<const uint32_t*>(a.dattn + arow + j) : 0u;
      }
    }
  };
  if constexpr (MODE == 1) fetch(0, pa, da);
  f32x4_t ac[4][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) ac[rt][jt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  issue(0, 0);
  issue(nsteps > 1 ? 1 : 0, 1);                       // slices past the end are re-issued copies: the counted waits stay exact
  int cur = 0;
#pragma unroll 1
  for (int s = 0; s < nsteps; ++s) {
    am_wait_vmcnt<4 + QI>();                          // slice s has landed (this wave's part); slice s + 1 may still be moving
    __builtin_amdgcn_s_barrier();                     // ... for every wave; and everyone is done reading slice s - 1
    {
      int nxt = cur + 2;
      if (nxt >= 3) nxt -= 3;
      issue(s + 2 < nsteps ? s + 2 : nsteps - 1, nxt);
    }
    const unsigned char* st = smem + cur * ST_BYTES;
    bf16x8_t qa[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) qa[rt] = *reinterpret_cast<const bf16x8_t*>(st + ST_Q + am_frag_off(rt * 16 + lr, lg));
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
      const bf16x8_t kb = *reinterpret_cast<const bf16x8_t*>(st + ST_K + am_frag_off(64 * w + jt * 16 + lr, lg));
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) ac[rt][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[rt], kb, ac[rt][jt], 0, 0, 0);
    }
    cur = cur + 1 == 3 ? 0 : cur + 1;
  }
  am_wait_vmcnt<0>();                                // the copies issued past the end
  __syncthreads();                                   // operand slices are dead: LDS becomes the score tile
  // second product: its matrix (NT key rows x DK2, this head's columns) is fetched into registers NOW -- in the order of the LDS image it
  // will be written to once the tile is dead (1 KiB sub-tiles [32 keys][16 columns], the layout ds_read_b64_tr_b16 reads: gemm_glds.hip)
  // -- so that the trip runs under the row pass.  Keys past T2 are clamped: their tile columns are exactly zero.
  constexpr int NP = DK2 > 0 ? DK2 / 8 : 1;
  am_u32x4_t m2r[NP];
  if constexpr (DK2 > 0) {
    const char* mb = reinterpret_cast<const char*>(a.m2 + (int64_t)b * a.m2bs + h * DK2);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int pidx = t + NT * i;
      const int sidx = pidx >> 6, l = pidx & 63;
      const int ks = sidx / (DK2 / 16), mt = sidx - ks * (DK2 / 16);
      const int r = min(ks * 32 + (l >> 1), T2 - 1);
      m2r[i] = *reinterpret_cast<const am_u32x4_t*>(mb + ((int64_t)r * a.ldm2 + mt * 16 + (l & 1) * 8) * 2);
    }
  }
  float* S = reinterpret_cast<float*>(smem);
  const uint64_t seed = (a.seed_base ? *a.seed_base : 0ull) + a.seed_off;
  const float inv_keep = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  bf16_t* const gout = MODE == 0 ? a.attn : a.ds;
  if constexpr (MODE == 0) {
    // ---- scale, mask -> fp32 score tile S[64][SP].  Row il is later overwritten, by the wave that owns it, with the bf16 rows of the
    //      map (from byte 0) and of its dropped copy (from element RD)
    const int kl0 = a.klen ? a.klen[b] : T2;
    const int kl = kl0 < T2 ? kl0 : T2;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = rt * 16 + 4 * lg + r;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
          const int j = 64 * w + 16 * jt + lr;
          const bool ok = j < kl && (!a.causal || j <= i0 + il);
          S[il * SP + j] = ok ? ac[rt][jt][r] * a.scale : AM_NEG;
        }
      }
    __syncthreads();
    // ---- softmax, one wave per row (64 / NW rows each, ROLLED: this kernel runs cold code), a lane owns columns lane + 64 c
#pragma unroll 1
    for (int rr = 0; rr < 64 / NW; ++rr) {
      const int il = w * (64 / NW) + rr;
      const int klr = a.causal ? (kl < i0 + il + 1 ? kl : i0 + il + 1) : kl;      // valid keys of this row
      float val[NW];
      float mx = AM_NEG;
#pragma unroll
      for (int c = 0; c < NW; ++c) { val[c] = S[il * SP + lane + 64 * c]; mx = fmaxf(mx, val[c]); }
      mx = wave_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < NW; ++c) { val[c] = expf(val[c] - mx); sum += val[c]; }
      sum = wave_sum(sum);
      const float inv = 1.f / sum;
      const int64_t arow = ((int64_t)bh * T1 + i0 + il) * a.ld;
      bf16_t* rowA = reinterpret_cast<bf16_t*>(S + il * SP);      // LDS operations of one wave execute in order: the row was read above
      bf16_t* rowD = rowA + RD;
#pragma unroll
      for (int c = 0; c < NW; ++c) {
        const int j = lane + 64 * c;
        const float pr = j < klr ? val[c] * inv : 0.f;              // masked_fill(mask, 0.0) after the softmax
        rowA[j] = f2bf(pr);
        if (a.pdrop) {
          float pd = pr;
          if (a.p > 0.f && j < a.ld) pd *= dropout_scale(seed, (uint64_t)(arow + j), a.p, inv_keep);
          rowD[j] = f2bf(pd);
        }
      }
    }
  } else {
    // ---- backward: the tile is dP = d context . v^T;  dS = P (dP mask + d map - sum_j P (dP mask + d map)) scale  (softmax.hip's
    //      backward kernel with the product in front of it: the same lane <-> column assignment, the same order of the row sum)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = rt * 16 + 4 * lg + r;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) S[il * SP + 64 * w + 16 * jt + lr] = ac[rt][jt][r];
      }
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
      uint32_t pn[G][CP], dn[G][CP];
      if (g + 1 < NG) fetch(g + 1, pn, dn);
#pragma unroll
      for (int rr = 0; rr < G; ++rr) {
        const int il = w * RW + g * G + rr;
        const int64_t arow = ((int64_t)bh * T1 + i0 + il) * a.ld;
        float pr[NW], tt[NW];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          const int j = 2 * lane + 128 * c;
          const float2 dp2 = *reinterpret_cast<const float2*>(S + il * SP + j);
          float m0 = 1.f, m1 = 1.f;
          if (a.p > 0.f) {
            m0 = dropout_scale(seed, (uint64_t)(arow + j), a.p, inv_keep);
            m1 = dropout_scale(seed, (uint64_t)(arow + j + 1), a.p, inv_keep);
          }
          const uint32_t pw = j < a.ld ? pa[rr][c] : 0u, dw = j < a.ld ? da[rr][c] : 0u;
          pr[2 * c] = __uint_as_float(pw << 16);                     // (pad columns of the stored map are zero: dS = 0 there)
          pr[2 * c + 1] = __uint_as_float(pw & 0xffff0000u);
          tt[2 * c] = dp2.x * m0 + __uint_as_float(dw << 16);
          tt[2 * c + 1] = dp2.y * m1 + __uint_as_float(dw & 0xffff0000u);
          dot += pr[2 * c] * tt[2 * c] + pr[2 * c + 1] * tt[2 * c + 1];
        }
        dot = wave_sum(dot);
        uint32_t* rowA = reinterpret_cast<uint32_t*>(S + il * SP);   // (LDS operations of one wave execute in order: the row was read above)
#pragma unroll
        for (int c = 0; c < CP; ++c)
          rowA[lane + 64 * c] = f2bf2(pr[2 * c] * (tt[2 * c] - dot) * a.scale, pr[2 * c + 1] * (tt[2 * c + 1] - dot) * a.scale);
      }
      if (g + 1 < NG) {
#pragma unroll
        for (int rr = 0; rr < G; ++rr)
#pragma unroll
          for (int c = 0; c < CP; ++c) { pa[rr][c] = pn[rr][c]; da[rr][c] = dn[rr][c]; }
      }
    }
  }
  __syncthreads();
  const int nv = a.ld >> 3;                           // 16-byte vectors per row of the map
  constexpr int VPR = NT / 8;
  for (int n = t; n < 64 * VPR; n += NT) {
    const int row = n / VPR, c8 = n - row * VPR;
    if (i0 + row < T1 && c8 < nv) {
      const int64_t o = ((int64_t)bh * T1 + i0 + row) * a.ld + c8 * 8;
      const bf16_t* rowA = reinterpret_cast<const bf16_t*>(S + row * SP);
      *reinterpret_cast<uint4*>(gout + o) = *reinterpret_cast<const uint4*>(rowA + c8 * 8);
      if (MODE == 0 && a.pdrop) *reinterpret_cast<uint4*>(a.pdrop + o) = *reinterpret_cast<const uint4*>(rowA + RD + c8 * 8);
    }
  }
  if (MODE == 1 && a.dbd) {                           // whole rows, zeros included: no fill launch in front of this kernel
    const int nvb = a.ldb >> 3;
    for (int n = t; n < 64 * nvb; n += NT) {
      const int row = n / nvb, c8 = n - row * nvb;
      const int i = i0 + row;
      if (i >= T1) break;
      const unsigned short* rowA = reinterpret_cast<const unsigned short*>(S + row * SP);
      const int j0 = c8 * 8 - (T1 - 1 - i);
      uint32_t o4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ja = j0 + 2 * e, jb = ja + 1;
        const uint32_t lo = (ja >= 0 && ja < T2) ? rowA[ja] : 0u, hi = (jb >= 0 && jb < T2) ? rowA[jb] : 0u;
        o4[e] = lo | (hi << 16);
      }
      *reinterpret_cast<uint4*>(a.dbd + ((int64_t)bh * T1 + i) * a.ldb + c8 * 8) = make_uint4(o4[0], o4[1], o4[2], o4[3]);
    }
  }
  if constexpr (DK2 > 0) {
    // ---- out2 = tile . m2.  A wave takes RT row tiles of 16 rows and 1 / CS of the 16-column tiles; its A fragments come from the
    //      bf16 rows in LDS (in the k order the transposing read delivers: 4 lg .. 4 lg + 3 and 16 + 4 lg .. of every 32 keys), then
    //      the tile is dead and LDS takes the image of m2 out of the registers
    constexpr int RT = NW >= 4 ? 1 : 4 / NW, CS = NW > 4 ? NW / 4 : 1, KH = NT / 32, MT = DK2 / 16 / CS;
    const int rt0 = NW >= 4 ? w / CS : w * RT, c0 = (w % CS) * MT;
    const int sel = (MODE == 0 && a.pdrop) ? RD * 2 : 0;            // forward with dropout: the dropped copy is what meets v
    bf16x8_t af[RT][KH];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const unsigned char* rowp = reinterpret_cast<const unsigned char*>(S + ((rt0 + rt) * 16 + lr) * SP) + sel + 8 * lg;
#pragma unroll
      for (int ks = 0; ks < KH; ++ks) {
        const am_s16x4_t lo = *reinterpret_cast<const am_s16x4_t*>(rowp + ks * 64);
        const am_s16x4_t hi = *reinterpret_cast<const am_s16x4_t*>(rowp + ks * 64 + 32);
        const am_s16x8_t v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[rt][ks] = __builtin_bit_cast(bf16x8_t, v8);
      }
    }
    __syncthreads();                                   // fragments taken, the stores above have read their rows
#pragma unroll
    for (int i = 0; i < NP; ++i) *reinterpret_cast<am_u32x4_t*>(smem + (size_t)(t + NT * i) * 16) = m2r[i];
    __syncthreads();
    f32x4_t o[RT][MT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) o[rt][mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // (the MT fragments of the next 32 keys are read while this step's MFMAs run: one LDS latency per step, not one per fragment)
    bf16x8_t bq[2][MT];
    auto load_b = [&](int ks, bf16x8_t (&dst)[MT]) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const unsigned char* pp = smem + (ks * (DK2 / 16) + c0 + mt) * 1024 + (lg * 4 + (lr >> 2)) * 32 + (lr & 3) * 8;
        const am_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((am_lds_s16x4_t*)pp);
        const am_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((am_lds_s16x4_t*)(pp + 512));
        const am_s16x8_t v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        dst[mt] = __builtin_bit_cast(bf16x8_t, v8);
      }
    };
    load_b(0, bq[0]);
#pragma unroll
    for (int ks = 0; ks < KH; ++ks) {
      if (ks + 1 < KH) load_b(ks + 1, bq[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);               // (the scheduler otherwise folds the reads back in front of their MFMAs)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) o[rt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt][ks], bq[ks & 1][mt], o[rt][mt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    bf16_t* ob = a.out2 + (int64_t)b * a.o2bs + h * DK2;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + (rt0 + rt) * 16 + 4 * lg + r;
        if (i < T1) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) ob[(int64_t)i * a.ldo2 + (c0 + mt) * 16 + lr] = f2bf(o[rt][mt][r]);
        }
      }
  }
}

bool am_al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

template <int NW, int MODE, int DK2>
int am_launch(const am_args& a, int B, hipStream_t st) {
  constexpr int NT = 64 * NW;
  constexpr size_t stages = (size_t)3 * (NT * 64 + 64 * 64), tile = (size_t)64 * (NT + 4) * 4;
  static_assert((size_t)NT * DK2 * 2 <= tile, "the second product's matrix must fit the dead tile");
  const size_t lds = stages > tile ? stages : tile;
  static size_t attr_set = 0;
  if (lds > 64 * 1024 && attr_set < lds) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_map_kernel<NW, MODE, DK2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      s2svc_set_error("attn_map: cannot raise the dynamic LDS limit");
      return -2;
    }
    attr_set = lds;
  }
  hipLaunchKernelGGL((attn_map_kernel<NW, MODE, DK2>), dim3((a.T1 + 63) / 64, B * a.H), dim3(NT), lds, st, a);
  return 0;
}

template <int MODE, int DK2>
int am_launch_nw(const am_args& a, int B, hipStream_t st) {
  return a.T2 <= 128 ? am_launch<2, MODE, DK2>(a, B, st) : a.T2 <= 256 ? am_launch<4, MODE, DK2>(a, B, st) : am_launch<8, MODE, DK2>(a, B, st);
}

template <int MODE>
int am_dispatch(const am_args& a, int B, hipStream_t st) {
  if (!a.m2) return am_launch_nw<MODE, 0>(a, B, st);
  switch (a.dk) {
    case 64: return am_launch_nw<MODE, 64>(a, B, st);
    case 96: return am_launch_nw<MODE, 96>(a, B, st);
    case 128: return am_launch_nw<MODE, 128>(a, B, st);
  }
  s2svc_set_error("attn_map: the second product needs d_k in {64, 96, 128}");
  return -1;
}

bool am_view_ok(const void* p, int64_t ld, int64_t bs) { return p && am_al16(p) && ld % 8 == 0 && bs % 8 == 0; }

}  // namespace

extern "C" int s2svc_attn_map_supported(int dtype, int T1, int T2, int dk) {
  return (dtype == S2S_BF16 && T1 >= 1 && T2 >= 1 && T2 <= 512 && dk >= 32 && dk % 32 == 0) ? 1 : 0;
}

extern "C" int s2svc_attn_map_product_supported(int dk) { return (dk == 64 || dk == 96 || dk == 128) ? 1 : 0; }

// q (B, T1, .) / k (B, T2, .) views with row strides ldq / ldk and batch strides qbs / kbs (elements), head h at columns h * dk;
// klen (B) int32 or NULL; attn / pdrop (B, H, T1, ld) bf16, ld = T2 rounded up to 8 (pdrop NULL when drop_p == 0).
// v != NULL (s2svc_attn_map_product_supported(dk)): ctx (B, T1, .) view receives the context (dropped map) . v in the same launch.
extern "C" int s2svc_attn_map_fwd(int B, int H, int T1, int T2, int dk, const void* q, int64_t ldq, int64_t qbs, const void* k, int64_t ldk,
                                  int64_t kbs, const int32_t* klen, int causal, float scale, float drop_p, const uint64_t* seed_base,
                                  uint64_t seed_off, void* attn, void* pdrop, int ld, const void* v, int64_t ldv, int64_t vbs, void* ctx,
                                  int64_t ldc, int64_t cbs, void* stream) {
  S2S_REQUIRE(s2svc_attn_map_supported(S2S_BF16, T1, T2, dk), "attn_map_fwd: bf16, T2 <= 512, d_k % 32 == 0");
  S2S_REQUIRE(B >= 0 && H > 0 && q && k && attn && ld >= T2 && ld % 8 == 0 && ld <= ((T2 + 63) / 64) * 64 && (drop_p == 0.f || pdrop) && drop_p < 1.f,
              "attn_map_fwd: bad args");
  S2S_REQUIRE(ldq % 8 == 0 && qbs % 8 == 0 && ldk % 8 == 0 && kbs % 8 == 0 && am_al16(q) && am_al16(k) && am_al16(attn) && am_al16(pdrop),
              "attn_map_fwd: 16-byte aligned operands, strides multiples of 8");
  S2S_REQUIRE(!v || (am_view_ok(v, ldv, vbs) && ctx && s2svc_attn_map_product_supported(dk)), "attn_map_fwd: context product needs v, ctx, d_k in {64, 96, 128}");
  if (B == 0) return 0;
  am_args a;
  a.H = H; a.T1 = T1; a.T2 = T2; a.dk = dk; a.ld = ld;
  a.q = (const bf16_t*)q; a.ldq = ldq; a.qbs = qbs; a.k = (const bf16_t*)k; a.ldk = ldk; a.kbs = kbs;
  a.klen = klen; a.causal = causal; a.scale = scale; a.p = drop_p; a.seed_base = seed_base; a.seed_off = seed_off;
  a.attn = (bf16_t*)attn; a.pdrop = drop_p > 0.f ? (bf16_t*)pdrop : nullptr;
  a.p_in = nullptr; a.dattn = nullptr; a.ds = nullptr; a.dbd = nullptr; a.ldb = 0;
  a.m2 = (const bf16_t*)v; a.ldm2 = ldv; a.m2bs = vbs; a.out2 = (bf16_t*)ctx; a.ldo2 = ldc; a.o2bs = cbs;
  const int rc = am_dispatch<0>(a, B, (hipStream_t)stream);
  if (rc) return rc;
  S2S_CHECK_LAUNCH("attn_map_kernel<fwd>");
  return 0;
}

// The gradient of the scaled scores in one launch: dS = P (dP mask + dattn - rowsum(P (dP mask + dattn))) scale with dP = dctx . v^T
// never leaving the chip.  dctx (B, T1, .) / v (B, T2, .) views as q / k above; attn (B, H, T1, ld) the stored map; dattn the gradient
// that reached the map itself (same layout) or NULL; ds (B, H, T1, ld) bf16 out (pad columns zero); masks regenerated from the seed.
// dbd != NULL (relative-position self-attention, T1 == T2, "new" rel_shift): dbd (B, H, T1, ldb) bf16, ldb >= 2 T1 - 1 a multiple of 8,
// receives dS at the positions the shift read (dbd[b, h, i, T1 - 1 - i + j] = dS[b, h, i, j]) and zeros elsewhere.
// k != NULL (s2svc_attn_map_product_supported(dk)): dq (B, T1, .) view receives dS . k in the same launch.
extern "C" int s2svc_attn_map_bwd(int B, int H, int T1, int T2, int dk, const void* dctx, int64_t ldo, int64_t obs, const void* v, int64_t ldv,
                                  int64_t vbs, const void* attn, const void* dattn, float scale, float drop_p, const uint64_t* seed_base,
                                  uint64_t seed_off, void* ds, int ld, void* dbd, int ldb, const void* k, int64_t ldk, int64_t kbs, void* dq,
                                  int64_t lddq, int64_t dqbs, void* stream) {
  S2S_REQUIRE(!dbd || (T1 == T2 && ldb >= 2 * T1 - 1 && ldb % 8 == 0 && am_al16(dbd)), "attn_map_bwd: dbd needs T1 == T2, ldb >= 2 T1 - 1, ldb % 8 == 0");
  S2S_REQUIRE(s2svc_attn_map_supported(S2S_BF16, T1, T2, dk), "attn_map_bwd: bf16, T2 <= 512, d_k % 32 == 0");
  S2S_REQUIRE(B >= 0 && H > 0 && dctx && v && attn && ds && ld >= T2 && ld % 8 == 0 && ld <= ((T2 + 63) / 64) * 64 && drop_p < 1.f, "attn_map_bwd: bad args");
  S2S_REQUIRE(ldo % 8 == 0 && obs % 8 == 0 && ldv % 8 == 0 && vbs % 8 == 0 && am_al16(dctx) && am_al16(v) && am_al16(ds),
              "attn_map_bwd: 16-byte aligned operands, strides multiples of 8");
  S2S_REQUIRE(!k || (am_view_ok(k, ldk, kbs) && dq && s2svc_attn_map_product_supported(dk)), "attn_map_bwd: dq product needs k, dq, d_k in {64, 96, 128}");
  if (B == 0) return 0;
  am_args a;
  a.H = H; a.T1 = T1; a.T2 = T2; a.dk = dk; a.ld = ld;
  a.q = (const bf16_t*)dctx; a.ldq = ldo; a.qbs = obs; a.k = (const bf16_t*)v; a.ldk = ldv; a.kbs = vbs;
  a.klen = nullptr; a.causal = 0; a.scale = scale; a.p = drop_p; a.seed_base = seed_base; a.seed_off = seed_off;
  a.attn = nullptr; a.pdrop = nullptr;
  a.p_in = (const bf16_t*)attn; a.dattn = (const bf16_t*)dattn; a.ds = (bf16_t*)ds; a.dbd = (bf16_t*)dbd; a.ldb = ldb;
  a.m2 = (const bf16_t*)k; a.ldm2 = ldk; a.m2bs = kbs; a.out2 = (bf16_t*)dq; a.ldo2 = lddq; a.o2bs = dqbs;
  const int rc = am_dispatch<1>(a, B, (hipStream_t)stream);
  if (rc) return rc;
  S2S_CHECK_LAUNCH("attn_map_kernel<bwd>");
  return 0;
}
