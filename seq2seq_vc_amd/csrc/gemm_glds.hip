// bf16 MFMA GEMM, LDS-DMA staged and double buffered (same contract as gemm.hip / gemm_fast.hip).
//
//   * K-contiguous operands (activations, weights, implicit-im2col conv inputs) go HBM -> LDS with
//     `global_load_lds_dwordx4` (16 B per lane, no staging registers, no ds_write pass).  The LDS image of a tile is
//     lane-linear, as that instruction requires; bank conflicts of the ds_read_b128 fragment reads are removed by an
//     XOR swizzle applied to the per-lane SOURCE address and, identically, to the read address: the 16-byte piece
//     c of row r lives at r*128 + ((c ^ ((r>>1)&7))<<4).  With that map every 16-lane group of a fragment read
//     (rows r..r+15 at pieces c / c^1) hits 16 distinct 16-byte bank slots.
//   * Row-contiguous operands (activations in wgrad, weights in dgrad) are DMA-ed k-major into LDS and their fragments
//     come from ds_read_b64_tr_b16 transpose reads (TrStage / tr_fragment below); a K-contiguous partner then reads the
//     same permuted k positions with two ds_read_b64 (kc_fragment_perm).  The register-transpose path (RcStage: 8x8
//     blocks, v_perm_b32, ds_write_b128 into the swizzled image) remains in the file for operands the transpose read cannot take.
//   * Two LDS buffers: tile t+1 is in flight (DMA + global loads) while the MFMAs consume tile t; one barrier per
//     K tile.  Pieces outside the matrix / the conv input are fetched from a 16-byte zero block.
//   * 128x128 or 64x64 output tile per 4-wave workgroup (2x2 waves, 4x4 / 2x2 MFMA 16x16x32 fragments per wave),
//     BK = 64, optional split-K (fp32 partials + splitk_reduce_kernel) and the fused A-row-sum of the wgrad GEMMs.
#include <cstring>
#include "gemm_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __attribute__((aligned(16))) uint4 g_zero16 = {0u, 0u, 0u, 0u};

// operand kinds (compile-time: the K loop has no mode branches): layout x addressing
enum { G_KC_DENSE = 0, G_KC_CONV1D = 1, G_KC_CONV2D = 2, G_RC_DENSE = 3, G_RC_CONV1D = 4, G_RC_CONV2D = 5, G_KC_TCONV2D = 6,
       G_TR_DENSE = 7, G_TR_CONV1D = 8, G_TR_CONV2D = 9 };   // row-contiguous operands, LDS-DMA staged + transpose reads
#define S2S_IS_TR(KIND) ((KIND) >= G_TR_DENSE)

// LDS image of an operand tile with BKT bf16 per row (16-byte pieces): piece c of row r lives at
//   BKT = 64 (128-B rows, 8 pieces): r*128 + ((c ^ ((r>>1)&7)) << 4)
//   BKT = 32 ( 64-B rows, 4 pieces): r*64  + ((c ^ ((-(r>>2))&3)) << 4)
// both maps put the 16 lanes of every ds_read_b128 lane group (rows r..r+15 at pieces c / c^1) on 16 distinct
// 16-byte bank slots of the 256-byte LDS row.
template <int BKT> __device__ __forceinline__ int swz_of(int r) { return BKT == 64 ? ((r >> 1) & 7) : ((-(r >> 2)) & 3); }
template <int BKT> __device__ __forceinline__ int lds_off_t(int r, int c) { return r * (BKT * 2) + ((c ^ swz_of<BKT>(r)) << 4); }
__device__ __forceinline__ int lds_off(int r, int c) { return lds_off_t<64>(r, c); }

// address of the zero block as an opaque per-lane value: keeps `ok ? src : zero` a plain 64-bit select, so ONE DMA /
// load instruction serves valid and padding lanes alike (a visible global address makes hipcc split every load in two
// EXEC-masked halves with a branch around each)
__device__ __forceinline__ uint64_t zero_addr() {
  uint64_t z = reinterpret_cast<uint64_t>(&g_zero16);
  asm volatile("" : "+v"(z));
  return z;
}

// ---------------------------------------------------------------------------------------------------------
// K-contiguous operand: ROWS/32 LDS-DMA instructions per wave and K tile (64 lanes x 16 B = 8 rows each).
// All address arithmetic is select-based (no branches around the DMA); conv taps advance by compare-and-wrap
// from the tile's uniform (tap, channel) origin instead of a per-lane division (needs C >= 64).
// ---------------------------------------------------------------------------------------------------------
template <int ROWS, int KIND, int BKT = 64>
struct KcStage {
  static constexpr int PIECES = BKT / 8;               // 16-byte pieces per row
  static constexpr int RPI = 64 / PIECES;              // rows per DMA instruction (64 lanes x 16 B)
  static constexpr int NI = ROWS / RPI / 4;            // DMA instructions per wave and tile
  static_assert(NI >= 1, "tile too small for 4 waves");
  int64_t rowbase[NI];            // element offset of the row's first tap ; < 0: row outside the matrix
  int trow[NI];                   // conv1d: frame index of the row ; tconv2d: class-grid row i
  int tcol[NI];                   // tconv2d: class-grid column j
  int piece8;                     // 8 * source piece of this lane (the same for every row group, see init)
  // Scalar-base path (dense and conv2d rows; full k tiles; C % BKT == 0; operand below 4 GiB): the lane-dependent part of
  // every source address is a 32-bit byte offset fixed at init (rows past the matrix clamp to its last row -- their
  // products only reach C rows / columns that are never stored), the k-dependent part is ONE scalar added to the uniform
  // base, so the DMA instructions take the SGPR-base + VGPR-offset form and the k loop spends no VALU work on addresses.
  static constexpr bool SCALAR_KIND = (KIND == G_KC_DENSE || KIND == G_KC_CONV2D);
  // The transposed-convolution rows (tconv2d) take the same scalar walk over (tap, channel); which of the <= 2 x 2 taps of
  // the parity class fall inside the output-gradient image is a 4-bit mask per row fixed at init, so a DMA instruction
  // costs one 64-bit add, one mask test and the select of the zero block.
  uint32_t rowoff[NI];
  int vmask[NI];                  // tconv2d: bit (2 * ta + fb) = tap (ta, fb) of this row reads inside the image
  bool scalar_ok;                 // uniform
  int st_k0, st_c0, st_kh, st_kw; // conv2d / tconv2d: the (tap, channel) position of k tile st_k0, walked without divisions
  __device__ __forceinline__ void init(const s2svc_operand& o, int r0, int R) {
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;      // (& 3: the 8-wave split-K kernel stages per 4-wave group)
    scalar_ok = false;
    st_k0 = -1; st_c0 = 0; st_kh = 0; st_kw = 0;
    if (SCALAR_KIND) {
      const int64_t elems = KIND == G_KC_DENSE ? (int64_t)R * o.ld
                                               : (int64_t)(R / (o.T2 * o.F2) + 1) * o.T1 * o.F1 * o.ld;
      scalar_ok = elems * 2 < (1ll << 32) && R > 0 && (KIND == G_KC_DENSE || o.C % BKT == 0);
    }
    if (KIND == G_KC_TCONV2D) {
      const int64_t elems = (int64_t)(R / (o.T1 * o.F1) + 1) * o.T2 * o.F2 * o.ld;
      scalar_ok = elems * 2 < (1ll << 32) && o.C % BKT == 0;
    }
    // rows of group g = 4i + wave: rl = g*RPI + lane/PIECES ; the swizzle term of rl does not depend on i
    // (BKT 64: ((g&1)*4 + (lane>>4)) & 7 with g&1 == wave&1 ; BKT 32: (-(lane>>4)) & 3)
    piece8 = ((lane % PIECES) ^ swz_of<BKT>(wave * RPI + lane / PIECES)) << 3;
    // row index -> image position by multiply-high (exact: the host checked rows * extent < 2^32, rows_fit_fastdiv)
    const fastdiv_t fd0 = fastdiv_make(KIND == G_KC_TCONV2D ? o.T1 * o.F1 : (KIND == G_KC_CONV2D ? o.F2 : 1));
    const fastdiv_t fd1 = fastdiv_make(KIND == G_KC_TCONV2D ? o.F1 : (KIND == G_KC_CONV2D ? o.T2 : 1));
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int rl = (i * 4 + wave) * RPI + lane / PIECES;
      const int rr = r0 + rl;
      const int r = (SCALAR_KIND && rr >= R) ? R - 1 : rr;     // (the other kinds mark the row invalid below)
      trow[i] = 0;
      tcol[i] = 0;
      vmask[i] = 0;
      if (KIND == G_KC_TCONV2D) {       // class grid (o.T1 x o.F1) -> output-gradient pixel (i, j) of the (B, T2, F2, C) tensor
        const int per_b = o.T1 * o.F1;
        const int b = fastdiv(r, fd0), rem = r - b * per_b;
        trow[i] = fastdiv(rem, fd1);
        tcol[i] = rem - trow[i] * o.F1;
        rowbase[i] = ((int64_t)(b * o.T2 + trow[i]) * o.F2 + tcol[i]) * o.ld;
        int m = 0;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
          for (int fb = 0; fb < 2; ++fb) {
            const int ti = trow[i] - ta, tj = tcol[i] - fb;
            if (rr < R && ti >= 0 && ti < o.T2 && tj >= 0 && tj < o.F2) m |= 1 << (2 * ta + fb);
          }
        vmask[i] = m;
      } else if (KIND == G_KC_DENSE) {
        rowbase[i] = (int64_t)r * o.ld;
      } else if (KIND == G_KC_CONV1D) {
        rowbase[i] = (int64_t)r * o.ld;
        trow[i] = r % o.T;
      } else {
        const int bt = fastdiv(r, fd0), f2 = r - bt * o.F2;
        const int b = fastdiv(bt, fd1), t2 = bt - b * o.T2;
        rowbase[i] = ((int64_t)(b * o.T1 + 2 * t2) * o.F1 + 2 * f2) * o.ld;
      }
      rowoff[i] = (uint32_t)((rowbase[i] + piece8) * 2);
      if (rr >= R) rowbase[i] = -1;
    }
  }
  __device__ __forceinline__ void issue(const s2svc_operand& o, const bf16_t* base, int k0, int K, char* lds) {
    const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);      // scalar: the LDS destinations (M0) stay on the SALU
    if (SCALAR_KIND && scalar_ok && k0 + BKT <= K) {
      int64_t koff = k0;
      if (KIND == G_KC_CONV2D) {
        if (k0 != st_k0) {                      // (first tile of this workgroup's k range)
          const int tap = k0 / o.C;
          st_c0 = k0 - tap * o.C;
          st_kh = tap / 3;
          st_kw = tap - st_kh * 3;
        }
        koff = (int64_t)(st_kh * o.F1 + st_kw) * o.ld + st_c0;
        st_k0 = k0 + BKT;
        st_c0 += BKT;
        if (st_c0 >= o.C) {
          st_c0 = 0;
          if (++st_kw == 3) { st_kw = 0; ++st_kh; }
        }
      }
      const char* sbase = reinterpret_cast<const char*>(base + koff);
#pragma unroll
      for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_global_load_lds((gbl_void*)(sbase + rowoff[i]), (lds_void*)(lds + (i * 4 + wave) * 1024), 16, 0, 0);
      return;
    }
    const uint64_t zaddr = zero_addr();
    if (KIND == G_KC_TCONV2D && scalar_ok && k0 + BKT <= K) {
      const int nf = 2 - (o.pad & 1);                       // taps along f of this parity class
      if (k0 != st_k0) {
        const int tap = k0 / o.C;
        st_c0 = k0 - tap * o.C;
        st_kh = tap / nf;
        st_kw = tap - st_kh * nf;
      }
      const int64_t koff = -(int64_t)(st_kh * o.F2 + st_kw) * o.ld + st_c0;
      const int bit = 1 << (2 * st_kh + st_kw);
      st_k0 = k0 + BKT;
      st_c0 += BKT;
      if (st_c0 >= o.C) {
        st_c0 = 0;
        if (++st_kw == nf) { st_kw = 0; ++st_kh; }
      }
      const char* sbase = reinterpret_cast<const char*>(base + koff);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const uint64_t src = (vmask[i] & bit) ? reinterpret_cast<uint64_t>(sbase + rowoff[i]) : zaddr;
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(lds + (i * 4 + wave) * 1024), 16, 0, 0);
      }
      return;
    }
    const int k = k0 + piece8;
    const bool kin = k < K;
    int64_t koff = k;             // element offset added to the row base
    int tap = 0, ta = 0, fb = 0;
    if (KIND != G_KC_DENSE) {
      const int tap0 = k0 / o.C, c0 = k0 - tap0 * o.C;      // uniform over the workgroup
      int c = c0 + piece8;
      tap = tap0;
      if (c >= o.C) { c -= o.C; tap += 1; }
      if (KIND == G_KC_CONV1D) {
        koff = (int64_t)(tap - o.pad) * o.ld + c;
      } else if (KIND == G_KC_TCONV2D) {
        const int nf = 2 - (o.pad & 1);                     // taps along f of this parity class
        ta = tap / nf;
        fb = tap - ta * nf;
        koff = -(int64_t)(ta * o.F2 + fb) * o.ld + c;
      } else {
        const int kh = tap / 3, kw = tap - kh * 3;
        koff = (int64_t)(kh * o.F1 + kw) * o.ld + c;
      }
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      bool ok = kin && rowbase[i] >= 0;
      if (KIND == G_KC_CONV1D) {
        const int tt = trow[i] + tap - o.pad;
        ok = ok && tt >= 0 && tt < o.T;
      }
      if (KIND == G_KC_TCONV2D) {
        const int ti = trow[i] - ta, tj = tcol[i] - fb;
        ok = ok && ti >= 0 && ti < o.T2 && tj >= 0 && tj < o.F2;
      }
      const uint64_t src = ok ? reinterpret_cast<uint64_t>(base + rowbase[i] + koff) : zaddr;
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(lds + (i * 4 + wave) * 1024), 16, 0, 0);
    }
  }
};

// ---------------------------------------------------------------------------------------------------------
// Row-contiguous operand: 8x8 register-block transpose (one block per thread; ROWS blocks per tile)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t perm_lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ uint32_t perm_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

template <int ROWS, int KIND>
struct RcStage {
  // The tile is ROWS/8 row blocks x 8 k blocks of 8x8 elements.  All 256 threads take part: TPB threads share a
  // block, each loading KS = 8/TPB consecutive k (16 B = 8 rows per load) and writing, per row, its KS k-values
  // (8 or 4 bytes) into the row's 16-byte piece of the swizzled image.
  // (Measured: for 128-row tiles one thread per block -- 8 loads, 8 ds_write_b128, half the threads idle -- is the
  // faster arrangement, for 64-row tiles four threads per block.)
  static constexpr int RB = ROWS / 8;
  static constexpr int TPB = ROWS == 64 ? 4 : 1;
  static constexpr int KS = 8 / TPB;         // k per thread
  static constexpr int ACTIVE = ROWS * TPB;  // threads that carry a (partial) block
  uint4 reg[KS];
  __device__ __forceinline__ void load(const s2svc_operand& o, const bf16_t* base, int r0, int R, int k0, int K) {
    if (ACTIVE < 256 && threadIdx.x >= ACTIVE) return;
    const int rb = threadIdx.x % RB, rest = threadIdx.x / RB;
    const int kb = rest / TPB, sub = rest % TPB;
    const int r = r0 + rb * 8;
    int tap = 0, c = r;
    if (KIND != G_RC_DENSE) { tap = r / o.C; c = r - tap * o.C; }
    const uint64_t zaddr = zero_addr();
    // the KS consecutive k of a thread walk (b, t2, f2) / (b, t) incrementally: one division per tile, not per load
    const int kfirst = k0 + kb * 8 + sub * KS;
    int f2 = 0, t2 = 0, bb = 0, t = 0;
    if (KIND == G_RC_CONV2D) {
      const int bt = kfirst / o.F2;
      f2 = kfirst - bt * o.F2;
      bb = bt / o.T2;
      t2 = bt - bb * o.T2;
    } else if (KIND == G_RC_CONV1D) {
      t = kfirst % o.T;
    }
    const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
      const int k = kfirst + j;
      bool ok = r < R && k < K;
      int64_t off;
      if (KIND == G_RC_DENSE) {
        off = (int64_t)k * o.ld + r;
      } else if (KIND == G_RC_CONV1D) {
        const int tt = t + tap - o.pad;
        ok = ok && tt >= 0 && tt < o.T;
        off = (int64_t)(k + tap - o.pad) * o.ld + c;
        if (++t == o.T) t = 0;
      } else {
        off = ((int64_t)(bb * o.T1 + 2 * t2 + kh) * o.F1 + (2 * f2 + kw)) * o.ld + c;
        if (++f2 == o.F2) { f2 = 0; if (++t2 == o.T2) { t2 = 0; ++bb; } }
      }
      reg[j] = *reinterpret_cast<const uint4*>(ok ? reinterpret_cast<uint64_t>(base + off) : zaddr);
    }
  }
  __device__ __forceinline__ void store(char* lds) const {
    if (ACTIVE < 256 && threadIdx.x >= ACTIVE) return;
    const int rb = threadIdx.x % RB, rest = threadIdx.x / RB;
    const int kb = rest / TPB, sub = rest % TPB;
    // element e (row rb*8+e) of reg[j] sits in dword e/2, half e%2
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      char* dst = lds + lds_off(rb * 8 + e, kb) + sub * (KS * 2);
      if (KS == 8) {
        uint32_t a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = (&reg[j % KS].x)[e >> 1];
        uint4 v;
        v.x = (e & 1) ? perm_hi(a[0], a[1]) : perm_lo(a[0], a[1]);
        v.y = (e & 1) ? perm_hi(a[2], a[3]) : perm_lo(a[2], a[3]);
        v.z = (e & 1) ? perm_hi(a[4], a[5]) : perm_lo(a[4], a[5]);
        v.w = (e & 1) ? perm_hi(a[6], a[7]) : perm_lo(a[6], a[7]);
        *reinterpret_cast<uint4*>(dst) = v;
      } else if (KS == 4) {
        const uint32_t a0 = (&reg[0].x)[e >> 1], a1 = (&reg[1 % KS].x)[e >> 1], a2 = (&reg[2 % KS].x)[e >> 1], a3 = (&reg[3 % KS].x)[e >> 1];
        uint2 v;
        v.x = (e & 1) ? perm_hi(a0, a1) : perm_lo(a0, a1);
        v.y = (e & 1) ? perm_hi(a2, a3) : perm_lo(a2, a3);
        *reinterpret_cast<uint2*>(dst) = v;
      } else {
        const uint32_t a0 = (&reg[0].x)[e >> 1], a1 = (&reg[1 % KS].x)[e >> 1];
        *reinterpret_cast<uint32_t*>(dst) = (e & 1) ? perm_hi(a0, a1) : perm_lo(a0, a1);
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------
// Row-contiguous operand WITHOUT the register transposes: the tile is DMA-ed k-major into LDS and the MFMA fragments
// are read with ds_read_b64_tr_b16 (gfx950's transposing LDS read: a 16-lane group reads a [4 k][16 rows] block, lane
// c of the group receives the 4 k-values of row c).  LDS image of a BK = 64 tile: 2 k-halves x ROWS/16 subtiles of
// [32 k][16 rows] bf16 (32-byte k-rows, 1 KiB = exactly one wave DMA instruction, lane l -> k-row l/2, rows 8*(l&1)..+7).
// A half-wave (2 lane groups) of a transpose read covers 8 consecutive 32-byte k-rows = one 256-byte bank row: no
// conflicts.  A fragment (8 k per lane) takes two reads; lane group g gets k = 4g..4g+3 and 16+4g..16+4g+3 of the
// k-half -- a PERMUTATION of the MFMA's k positions, harmless because both operands of the product use it
// (a K-contiguous partner reads the same positions: kc_fragment_perm).
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

template <int ROWS, int KIND>
struct TrStage {
  static constexpr int SUB = ROWS / 16;             // subtiles per k-half
  static constexpr int NI = 2 * SUB / 4;            // DMA instructions per wave and tile
  static_assert(NI >= 1, "tile too small for 4 waves");
  // Scalar-base path (dense and conv2d; full k tiles; operand below 4 GiB), as in KcStage: the lane's part of every
  // source address is a 32-bit byte offset -- fixed at init for a dense operand; for conv2d the (tap, channel) part is
  // fixed and the lane's two output pixels (one per k half) are walked from k tile to k tile with two compares instead of
  // two divisions per DMA instruction.  Rows past the matrix clamp to its last 16-byte piece (their products only reach
  // C rows / columns that are never stored).
  static constexpr bool SCALAR_KIND = (KIND == G_TR_DENSE || KIND == G_TR_CONV2D);
  uint32_t laneoff[NI];
  int st_k0;
  int pf2[2], pt2[2];
  uint32_t poff[2];
  __device__ __forceinline__ void init(const s2svc_operand& o, int r0, int R) {
    st_k0 = -1;
    if (!SCALAR_KIND) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int sidx = i * 4 + wave;
      const int kh = sidx / SUB, mt = sidx - kh * SUB;
      int r = r0 + mt * 16 + (lane & 1) * 8;
      if (r > R - 8) r = R - 8 > 0 ? R - 8 : 0;
      if (KIND == G_TR_DENSE) {
        laneoff[i] = (uint32_t)(((int64_t)(kh * 32 + (lane >> 1)) * o.ld + r) * 2);
      } else {
        const int tap = r / o.C, c = r - tap * o.C;
        const int kh3 = tap / 3, kw = tap - kh3 * 3;
        laneoff[i] = (uint32_t)(((int64_t)(kh3 * o.F1 + kw) * o.ld + c) * 2);
      }
    }
    pf2[0] = pf2[1] = pt2[0] = pt2[1] = 0;
    poff[0] = poff[1] = 0u;
  }
  __device__ __forceinline__ bool scalar_ok(const s2svc_operand& o, int R, int K) const {      // uniform
    if (!SCALAR_KIND || R < 8 || (R & 7)) return false;        // (the clamp of init moves whole 16-byte pieces)
    if (KIND == G_TR_DENSE) return (int64_t)K * o.ld * 2 < (1ll << 32);
    const int64_t elems = (int64_t)(K / (o.T2 * o.F2) + 1) * o.T1 * o.F1 * o.ld;
    return elems * 2 < (1ll << 32) && 64 / o.F2 + 1 <= o.T2;
  }
  __device__ __forceinline__ void issue(const s2svc_operand& o, const bf16_t* base, int r0, int R, int k0, int K, char* lds) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (SCALAR_KIND && k0 + 64 <= K && scalar_ok(o, R, K)) {
      if (KIND == G_TR_DENSE) {
        const char* sbase = reinterpret_cast<const char*>(base + (int64_t)k0 * o.ld);
#pragma unroll
        for (int i = 0; i < NI; ++i)
          __builtin_amdgcn_global_load_lds((gbl_void*)(sbase + laneoff[i]), (lds_void*)(lds + (i * 4 + wave) * 1024), 16, 0, 0);
        return;
      }
      if (k0 != st_k0) {                        // first tile of this workgroup's k range: place the two pixels
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int k = k0 + h * 32 + (lane >> 1);
          const int bt = k / o.F2;
          pf2[h] = k - bt * o.F2;
          const int b = bt / o.T2;
          pt2[h] = bt - b * o.T2;
          poff[h] = (uint32_t)(((int64_t)(b * o.T1 + 2 * pt2[h]) * o.F1 + 2 * pf2[h]) * o.ld * 2);
        }
      }
      const char* sbase = reinterpret_cast<const char*>(base);
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const int kh = (i * 4 + wave) / SUB;     // (uniform; a compile-time constant unless the tile has < 64 rows)
        __builtin_amdgcn_global_load_lds((gbl_void*)(sbase + (uint32_t)((kh ? poff[1] : poff[0]) + laneoff[i])),
                                         (lds_void*)(lds + (i * 4 + wave) * 1024), 16, 0, 0);
      }
      // next k tile: every pixel index grows by 64 = q * F2 + rem
      const int q = 64 / o.F2, rem = 64 - q * o.F2;
      const uint32_t step = (uint32_t)((int64_t)(2 * rem + 2 * q * o.F1) * o.ld * 2);
      const uint32_t wrapf = (uint32_t)((int64_t)(2 * o.F1 - 2 * o.F2) * o.ld * 2);
      const uint32_t wrapt = (uint32_t)((int64_t)(o.T1 - 2 * o.T2) * o.F1 * o.ld * 2);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        pf2[h] += rem;
        pt2[h] += q;
        poff[h] += step;
        if (pf2[h] >= o.F2) { pf2[h] -= o.F2; pt2[h] += 1; poff[h] += wrapf; }
        if (pt2[h] >= o.T2) { pt2[h] -= o.T2; poff[h] += wrapt; }
      }
      st_k0 = k0 + 64;
      return;
    }
    const uint64_t zaddr = zero_addr();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int sidx = i * 4 + wave;                // subtile: k-half kh, row block mt
      const int kh = sidx / SUB, mt = sidx - kh * SUB;
      const int k = k0 + kh * 32 + (lane >> 1);
      const int r = r0 + mt * 16 + (lane & 1) * 8;
      bool ok = r < R && k < K;
      int64_t off;
      if (KIND == G_TR_DENSE) {
        off = (int64_t)k * o.ld + r;
      } else if (KIND == G_TR_CONV1D) {
        const int tap = r / o.C, c = r - tap * o.C;
        const int t = k % o.T, tt = t + tap - o.pad;
        ok = ok && tt >= 0 && tt < o.T;
        off = (int64_t)(k + tap - o.pad) * o.ld + c;
      } else {
        const int tap = r / o.C, c = r - tap * o.C;
        const int f2 = k % o.F2, bt = k / o.F2;
        const int t2 = bt % o.T2, b = bt / o.T2;
        const int kh3 = tap / 3, kw = tap - kh3 * 3;
        off = ((int64_t)(b * o.T1 + 2 * t2 + kh3) * o.F1 + (2 * f2 + kw)) * o.ld + c;
      }
      const uint64_t src = ok ? reinterpret_cast<uint64_t>(base + off) : zaddr;
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(lds + sidx * 1024), 16, 0, 0);
    }
  }
};

// fragment of a K-contiguous (swizzled-piece) tile whose PARTNER is TR-staged: the same k positions as tr_fragment,
// k = 4g..4g+3 and 16+4g..16+4g+3 of the k-half = half a 16-byte piece each (two ds_read_b64)
__device__ __forceinline__ bf16x8_t kc_fragment_perm(const char* stage, int row, int ks, int g) {
  typedef __attribute__((ext_vector_type(4))) short s16x4;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x4 lo = *reinterpret_cast<const s16x4*>(stage + lds_off(row, ks * 4 + (g >> 1)) + (g & 1) * 8);
  const s16x4 hi = *reinterpret_cast<const s16x4*>(stage + lds_off(row, ks * 4 + 2 + (g >> 1)) + (g & 1) * 8);
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// MFMA fragment of row block `mt` (16 rows), k-half `ks` of a TR-staged tile
template <int ROWS>
__device__ __forceinline__ bf16x8_t tr_fragment(const char* stage, int mt, int ks, int lane) {
  const char* p = stage + (ks * (ROWS / 16) + mt) * 1024 + ((lane >> 4) * 4 + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p + 512));
  typedef __attribute__((ext_vector_type(8))) short s16x8_t;
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

// XCD-aware tile order.  The dispatcher deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own
// L2); handing XCD x the x-th CONTIGUOUS run of output tiles (row-major, n fastest) makes the workgroups that run
// side by side on one XCD share A row panels and B column panels through that L2 instead of each pulling them over
// the fabric.  Bijective for any grid size; z-planes (batch / split-K) keep the plain order unless the plane is a
// multiple of 8 workgroups (the XCD phase of a plane would otherwise depend on z).
__device__ __forceinline__ void tile_of_block(int& bm, int& bn) {
  const int gx = gridDim.x, nwg = gridDim.x * gridDim.y;
  int id = blockIdx.y * gx + blockIdx.x;
  if (gridDim.z == 1 || (nwg & 7) == 0) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = id & 7, j = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  bm = id / gx;
  bn = id - bm * gx;
}

template <int ROWS, int KIND, int CLS = (S2S_IS_TR(KIND) ? 2 : ((KIND >= G_RC_DENSE && KIND <= G_RC_CONV2D) ? 1 : 0))> struct Stage;
template <int ROWS, int KIND> struct Stage<ROWS, KIND, 2> {
  TrStage<ROWS, KIND> s;
  __device__ __forceinline__ void init(const s2svc_operand& o, int r0, int R) { s.init(o, r0, R); }
  __device__ __forceinline__ void issue(const s2svc_operand& o, const bf16_t* base, int r0, int R, int k0, int K, char* lds) { s.issue(o, base, r0, R, k0, K, lds); }
  __device__ __forceinline__ void finish(char*) const {}
};
template <int ROWS, int KIND> struct Stage<ROWS, KIND, 0> {
  KcStage<ROWS, KIND> s;
  __device__ __forceinline__ void init(const s2svc_operand& o, int r0, int R) { s.init(o, r0, R); }
  __device__ __forceinline__ void issue(const s2svc_operand& o, const bf16_t* base, int, int, int k0, int K, char* lds) { s.issue(o, base, k0, K, lds); }
  __device__ __forceinline__ void finish(char*) const {}
};
template <int ROWS, int KIND> struct Stage<ROWS, KIND, 1> {
  RcStage<ROWS, KIND> s;
  __device__ __forceinline__ void init(const s2svc_operand&, int, int) {}
  __device__ __forceinline__ void issue(const s2svc_operand& o, const bf16_t* base, int r0, int R, int k0, int K, char*) { s.load(o, base, r0, R, k0, K); }
  __device__ __forceinline__ void finish(char* lds) const { s.store(lds); }
};

// One output tile of one GEMM: the body shared by the plain launch (one problem per grid) and the grouped launch (several
// independent problems in one grid, see gemm_grouped_kernel).  smem: 2 * (BM + BN) * 128 bytes, 1024-aligned.
template <int BM, int BN, int AMODE, int BMODE>
__device__ __forceinline__ void gemm_glds_tile(const s2svc_gemm_desc& d, int tile_m, int tile_n, int zb, int zs, char* smem) {
  constexpr int BK = 64;                               // bf16 per K tile: 128-B rows, 8 pieces of 16 B
  constexpr int FM = BM / 32, FN = BN / 32;            // 16x16 fragments per wave along m / n
  constexpr int ABYTES = BM * 128, BBYTES = BN * 128, STAGE_BYTES = ABYTES + BBYTES;

  const int splitk = d.splitk > 1 ? d.splitk : 1;
  const int z0 = zb / d.nb1, z1 = zb - z0 * d.nb1;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const bf16_t* Ab = (const bf16_t*)d.A.ptr + (int64_t)z0 * d.A.bs0 + (int64_t)z1 * d.A.bs1;
  const bf16_t* Bb = (const bf16_t*)d.B.ptr + (int64_t)z0 * d.B.bs0 + (int64_t)z1 * d.B.bs1;
  const int ktiles = (d.K + BK - 1) / BK;
  const int per = (ktiles + splitk - 1) / splitk;
  const int kt_begin = zs * per;
  const int kt_end = (kt_begin + per < ktiles) ? kt_begin + per : ktiles;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
  const int lr = lane & 15, lg = lane >> 4;

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  Stage<BM, AMODE> sa;
  Stage<BN, BMODE> sb;
  sa.init(d.A, m0, d.M);
  sb.init(d.B, n0, d.N);
  if (kt_begin < kt_end) {
    sa.issue(d.A, Ab, m0, d.M, kt_begin * BK, d.K, smem);
    sb.issue(d.B, Bb, n0, d.N, kt_begin * BK, d.K, smem + ABYTES);
    sa.finish(smem);
    sb.finish(smem + ABYTES);
  }
  __syncthreads();                 // (drains the DMA: the compiler places vmcnt(0) ahead of the barrier)

  constexpr int TPR = 256 / BM;    // threads per A row for the fused row sums
  const bool do_rowsum = (d.a_rowsum != nullptr) && (tile_n == 0);
  float rowsum = 0.f;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    char* As = smem + cur * STAGE_BYTES;
    char* Bs = As + ABYTES;
    char* An = smem + (cur ^ 1) * STAGE_BYTES;
    const bool more = (kt + 1 < kt_end);
    if (more) {                    // tile kt+1 starts moving before the MFMAs of tile kt
      sa.issue(d.A, Ab, m0, d.M, (kt + 1) * BK, d.K, An);
      sb.issue(d.B, Bb, n0, d.N, (kt + 1) * BK, d.K, An + ABYTES);
    }
    if (do_rowsum && S2S_IS_TR(AMODE)) {
      // k-major image: row r is column r%16 of the subtiles (kh, r/16); `part` takes every TPR-th k-row
      const int r = threadIdx.x / TPR, part = threadIdx.x % TPR;
#pragma unroll
      for (int kk = 0; kk < 64 / TPR; ++kk) {
        const int k = kk * TPR + part;
        rowsum += bf2f(*reinterpret_cast<const bf16_t*>(As + ((k >> 5) * (BM / 16) + (r >> 4)) * 1024 + (k & 31) * 32 + (r & 15) * 2));
      }
    } else if (do_rowsum) {
      const int r = threadIdx.x / TPR, part = threadIdx.x % TPR;       // part covers 8/TPR pieces of the row
#pragma unroll
      for (int c = 0; c < 8 / TPR; ++c) {
        const uint4 v = *reinterpret_cast<const uint4*>(As + lds_off(r, part * (8 / TPR) + c));
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) rowsum += __uint_as_float(w[e] << 16) + __uint_as_float(w[e] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8_t a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        a[i] = S2S_IS_TR(AMODE) ? tr_fragment<BM>(As, wm / 16 + i, ks, lane)
               : S2S_IS_TR(BMODE) ? kc_fragment_perm(As, wm + i * 16 + lr, ks, lg)
                                  : *reinterpret_cast<const bf16x8_t*>(As + lds_off(wm + i * 16 + lr, ks * 4 + lg));
#pragma unroll
      for (int j = 0; j < FN; ++j)
        b[j] = S2S_IS_TR(BMODE) ? tr_fragment<BN>(Bs, wn / 16 + j, ks, lane)
               : S2S_IS_TR(AMODE) ? kc_fragment_perm(Bs, wn + j * 16 + lr, ks, lg)
                                  : *reinterpret_cast<const bf16x8_t*>(Bs + lds_off(wn + j * 16 + lr, ks * 4 + lg));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      sa.finish(An);
      sb.finish(An + ABYTES);
    }
    __syncthreads();
  }

  if (do_rowsum) {
#pragma unroll
    for (int o = 1; o < TPR; o <<= 1) rowsum += __shfl_xor(rowsum, o, 64);
    const int m = m0 + threadIdx.x / TPR;
    if ((threadIdx.x % TPR) == 0 && m < d.M) {
      if (splitk > 1) d.a_rowsum_ws[(int64_t)zs * d.M + m] = rowsum;
      else d.a_rowsum[m] = (d.a_rowsum_accumulate ? d.a_rowsum[m] : 0.f) + rowsum;
    }
  }
  static_assert(2 * STAGE_BYTES >= BM * BN * 4, "the operand stages double as the fp32 C tiles of the epilogue");
  __syncthreads();               // every wave is done with the operand stages: reuse them as fp32 C tiles
  epilogue_tile<BM / 2, BN / 2>(d, z0, z1, m0 + wm, n0 + wn, acc, reinterpret_cast<float*>(smem) + wave * (BM / 2) * (BN / 2),
                                splitk, zs, zb);
}

template <int BM, int BN, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void gemm_glds_kernel(const s2svc_gemm_desc d) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * (BM + BN) * 128];
  const int splitk = d.splitk > 1 ? d.splitk : 1;
  const int zb = blockIdx.z / splitk, zs = blockIdx.z - zb * splitk;
  int tile_m, tile_n;
  tile_of_block(tile_m, tile_n);
  gemm_glds_tile<BM, BN, AMODE, BMODE>(d, tile_m, tile_n, zb, zs, smem);
}

// Grouped launch: up to S2S_GROUP_MAX independent GEMMs (the weight-gradient GEMMs of a few consecutive layers, queued
// during backward) as ONE grid.  tile_start[p] .. tile_start[p+1] are the workgroups of problem p (row-major tiles).
// The descriptors travel BY VALUE in the kernel-argument segment (nothing to upload; hipGraph capture records them with
// the node).  Each workgroup runs the full K loop of its tile: with several problems' tiles in flight the grid fills
// the chip without split-K, so there is no partial-sum workspace and no reduction pass.
#define S2S_GROUP_MAX 10                                   /* 10 * 384 B + prefix sums < the 4 KB kernarg limit */
struct group_args {
  s2svc_gemm_desc d[S2S_GROUP_MAX];
  int32_t tile_start[S2S_GROUP_MAX + 1];
  int32_t n;
};
static_assert(sizeof(group_args) <= 4096, "kernel arguments are limited to 4 KB");

template <int BM, int BN, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void gemm_grouped_kernel(const group_args g) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * (BM + BN) * 128];
  int p = 0;                                              // tile_start[p] <= blockIdx.x < tile_start[p + 1]
#pragma unroll
  for (int i = 1; i < S2S_GROUP_MAX; ++i) p += (i < g.n && g.tile_start[i] <= (int)blockIdx.x) ? 1 : 0;
  const s2svc_gemm_desc& d = g.d[p];
  const int t = (int)blockIdx.x - g.tile_start[p];
  const int tiles_n = (d.N + BN - 1) / BN;
  const int tile_m = t / tiles_n;
  gemm_glds_tile<BM, BN, AMODE, BMODE>(d, tile_m, t - tile_m * tiles_n, 0, 0, smem);
}

// Grouped launch of BATCHED problems of one operand-kind pair: the batched products of an attention backward pass
// (dV = P^T dctx, dK = dS^T Q, d pos = dbd^T qv: row-contiguous x row-contiguous; dQ = dS K, d qv = dbd pos: K-contiguous x
// row-contiguous) are 3-8 GFLOP each with a reduction of 256-511 -- a launch of 128-768 short tiles whose time is the launch
// floor, the pipeline fill and a ragged last round.  tile_start[] counts tiles x batches; a workgroup decodes (problem, batch,
// tile) and runs the same tile code as the plain launch.
template <int BM, int BN, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void gemm_grouped_batched_kernel(const group_args g) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * (BM + BN) * 128];
  int p = 0;
#pragma unroll
  for (int i = 1; i < S2S_GROUP_MAX; ++i) p += (i < g.n && g.tile_start[i] <= (int)blockIdx.x) ? 1 : 0;
  const s2svc_gemm_desc& d = g.d[p];
  const int t = (int)blockIdx.x - g.tile_start[p];
  const int tiles_n = (d.N + BN - 1) / BN;
  const int per = ((d.M + BM - 1) / BM) * tiles_n;
  const int zb = t / per;
  const int r = t - zb * per;
  const int tile_m = r / tiles_n;
  gemm_glds_tile<BM, BN, AMODE, BMODE>(d, tile_m, r - tile_m * tiles_n, zb, 0, smem);
}

// The same with a CAPPED grid: gridDim.x workgroups walk all tiles.  Weight-gradient launches run on side
// streams beside the data-gradient chain; one workgroup per tile (500-1300 of them) takes every CU slot for the length of the
// launch and the chain's small kernels queue behind them, a capped grid leaves slots free.  Same tile code: same bits.
template <int BM, int BN, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void gemm_grouped_capped_kernel(const group_args g) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * (BM + BN) * 128];
  const int total = g.tile_start[S2S_GROUP_MAX];
#pragma unroll 1
  for (int bt = (int)blockIdx.x; bt < total; bt += (int)gridDim.x) {
    int p = 0;
#pragma unroll
    for (int i = 1; i < S2S_GROUP_MAX; ++i) p += (i < g.n && g.tile_start[i] <= bt) ? 1 : 0;
    const s2svc_gemm_desc& d = g.d[p];
    const int t = bt - g.tile_start[p];
    const int tiles_n = (d.N + BN - 1) / BN;
    const int tile_m = t / tiles_n;
    gemm_glds_tile<BM, BN, AMODE, BMODE>(d, tile_m, t - tile_m * tiles_n, 0, 0, smem);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// All-DMA variant (both operands K-contiguous): NS LDS stages, NS-1 tiles in flight.  The wait for tile t is a
// COUNTED s_waitcnt (the DMAs of the later tiles stay in flight across the barrier) followed by a raw s_barrier;
// the stage freed by that barrier is refilled immediately.  One barrier per K tile, no drain in the main loop.
// Every wave issues exactly NI_A + NI_B DMA instructions per tile, so "tile t landed" == vmcnt <= (tiles issued
// after t) * (NI_A + NI_B).
// ---------------------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BM, int BN, int KA, int KB, int NS, int BK, bool LEAN = false>
__global__ __launch_bounds__(256) void gemm_dma_kernel(const s2svc_gemm_desc d) {
  constexpr int FM = BM / 32, FN = BN / 32;
  constexpr int ABYTES = BM * BK * 2, BBYTES = BN * BK * 2, STAGE_BYTES = ABYTES + BBYTES;
  constexpr int PER_TILE = KcStage<BM, KA, BK>::NI + KcStage<BN, KB, BK>::NI;   // DMA instructions per wave and tile
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE_BYTES];

  const int splitk = d.splitk > 1 ? d.splitk : 1;
  const int zb = blockIdx.z / splitk, zs = blockIdx.z - zb * splitk;
  const int z0 = zb / d.nb1, z1 = zb - z0 * d.nb1;
  int tile_m, tile_n;
  tile_of_block(tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const bf16_t* Ab = (const bf16_t*)d.A.ptr + (int64_t)z0 * d.A.bs0 + (int64_t)z1 * d.A.bs1;
  const bf16_t* Bb = (const bf16_t*)d.B.ptr + (int64_t)z0 * d.B.bs0 + (int64_t)z1 * d.B.bs1;
  const int ktiles = (d.K + BK - 1) / BK;
  const int per = (ktiles + splitk - 1) / splitk;
  const int kt_begin = zs * per;
  const int kt_end = (kt_begin + per < ktiles) ? kt_begin + per : ktiles;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
  const int lr = lane & 15, lg = lane >> 4;

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  KcStage<BM, KA, BK> sa;
  KcStage<BN, KB, BK> sb;
  sa.init(d.A, m0, d.M);
  sb.init(d.B, n0, d.N);
  // prologue: NS-1 tiles in flight (tiles past the end are issued too -- they read the zero block -- so that the
  // per-wave DMA count per stage is always PER_TILE and the counted waits below stay exact)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
    sa.issue(d.A, Ab, (kt_begin + s) * BK, (kt_begin + s) < kt_end ? d.K : 0, smem + s * STAGE_BYTES);
    sb.issue(d.B, Bb, (kt_begin + s) * BK, (kt_begin + s) < kt_end ? d.K : 0, smem + s * STAGE_BYTES + ABYTES);
  }
  int cur = 0;                                          // stage of tile kt
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    wait_vmcnt<(NS - 2) * PER_TILE>();                  // tile kt has landed; NS-2 later tiles may still be moving
    __builtin_amdgcn_s_barrier();                       // ... for every wave; and everyone is done reading tile kt-1
    {                                                   // refill the stage tile kt-1 lived in with tile kt+NS-1
      int nxt = cur + NS - 1;
      if (nxt >= NS) nxt -= NS;
      const int kn = kt + NS - 1;
      sa.issue(d.A, Ab, kn * BK, kn < kt_end ? d.K : 0, smem + nxt * STAGE_BYTES);
      sb.issue(d.B, Bb, kn * BK, kn < kt_end ? d.K : 0, smem + nxt * STAGE_BYTES + ABYTES);
    }
    const char* As = smem + cur * STAGE_BYTES;
    const char* Bs = As + ABYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8_t a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(As + lds_off_t<BK>(wm + i * 16 + lr, ks * 4 + lg));
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + lds_off_t<BK>(wn + j * 16 + lr, ks * 4 + lg));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
  }
  wait_vmcnt<0>();                                      // the zero-block DMAs issued past the end

  static_assert(sizeof(smem) >= (size_t)BM * BN * 4, "the operand stages double as the fp32 C tiles of the epilogue");
  __syncthreads();               // every wave is done with the operand stages: reuse them as fp32 C tiles
  if (LEAN) {       // the common epilogue only (see epilogue_flush_common): a fraction of the code
    float* cs = reinterpret_cast<float*>(smem) + wave * (BM / 2) * (BN / 2);
    epilogue_stage<BM / 2, BN / 2>(acc, cs);
    epilogue_flush_common<BM / 2, BN / 2>(d, m0 + wm, n0 + wn, cs);
    return;
  }
  epilogue_tile<BM / 2, BN / 2>(d, z0, z1, m0 + wm, n0 + wn, acc, reinterpret_cast<float*>(smem) + wave * (BM / 2) * (BN / 2),
                                splitk, zs, zb);
}

// ---------------------------------------------------------------------------------------------------------
// Split-K INSIDE the workgroup (round 5): the all-DMA kernel for long reductions over few tiles -- the feed-forward and packed-
// projection products of the VTN / TTS stacks with K = 1152 / 1536 (2016 x 384 x 1536: 378 tiles of 32 x 64, 24 K tiles).  Their
// launch is a serial loop of K tiles at ~0.33 us each (one barrier, six fragment reads, four MFMAs per wave: all latency), 8 us of an
// 11 us launch, with two thirds of every CU idle.  Here a workgroup is 8 waves = two groups of four; each group walks HALF of the K
// range through its own LDS ring (same staging, same counted waits; the barriers are the workgroup's, both groups make the same number
// of trips), then the second group's accumulators go through LDS and the first group adds them -- first half + second half, a fixed
// order -- and runs the epilogue.  No second launch, no workspace, no atomics.  The result differs from the unsplit kernel's in fp32
// summation order only; WHICH kernel a problem gets is a function of its shape.
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN, int NS>
__global__ __launch_bounds__(512) void gemm_dma_k2_kernel(const s2svc_gemm_desc d) {
  constexpr int BK = 64;
  constexpr int FM = BM / 32, FN = BN / 32;
  constexpr int ABYTES = BM * BK * 2, BBYTES = BN * BK * 2, STAGE_BYTES = ABYTES + BBYTES, RING = NS * STAGE_BYTES;
  constexpr int PER_TILE = KcStage<BM, G_KC_DENSE, BK>::NI + KcStage<BN, G_KC_DENSE, BK>::NI;
  __shared__ __attribute__((aligned(1024))) char smem[2 * RING];
  static_assert(2 * RING >= 2 * BM * BN * 4, "the rings double as the partial-sum buffer and the fp32 C tiles");
  int tile_m, tile_n;
  tile_of_block(tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const bf16_t* Ab = (const bf16_t*)d.A.ptr;
  const bf16_t* Bb = (const bf16_t*)d.B.ptr;
  const int ktiles = (d.K + BK - 1) / BK;
  const int half = (ktiles + 1) / 2;
  const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave8 >> 2, wave = wave8 & 3, lane = threadIdx.x & 63;
  const int kt_begin = grp * half;
  const int kt_end = (kt_begin + half < ktiles) ? kt_begin + half : ktiles;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
  const int lr = lane & 15, lg = lane >> 4;
  char* ring = smem + grp * RING;

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  KcStage<BM, G_KC_DENSE, BK> sa;
  KcStage<BN, G_KC_DENSE, BK> sb;
  sa.init(d.A, m0, d.M);
  sb.init(d.B, n0, d.N);
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
    sa.issue(d.A, Ab, (kt_begin + s) * BK, (kt_begin + s) < kt_end ? d.K : 0, ring + s * STAGE_BYTES);
    sb.issue(d.B, Bb, (kt_begin + s) * BK, (kt_begin + s) < kt_end ? d.K : 0, ring + s * STAGE_BYTES + ABYTES);
  }
  int cur = 0;
  for (int it = 0; it < half; ++it) {                     // both groups make `half` trips (the second one's last may be empty)
    const int kt = kt_begin + it;
    wait_vmcnt<(NS - 2) * PER_TILE>();
    __builtin_amdgcn_s_barrier();
    {
      int nxt = cur + NS - 1;
      if (nxt >= NS) nxt -= NS;
      const int kn = kt + NS - 1;
      sa.issue(d.A, Ab, kn * BK, kn < kt_end ? d.K : 0, ring + nxt * STAGE_BYTES);
      sb.issue(d.B, Bb, kn * BK, kn < kt_end ? d.K : 0, ring + nxt * STAGE_BYTES + ABYTES);
    }
    const char* As = ring + cur * STAGE_BYTES;
    const char* Bs = As + ABYTES;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8_t a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(As + lds_off_t<BK>(wm + i * 16 + lr, ks * 4 + lg));
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + lds_off_t<BK>(wn + j * 16 + lr, ks * 4 + lg));
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    cur = cur + 1 == NS ? 0 : cur + 1;
  }
  wait_vmcnt<0>();
  __syncthreads();               // every wave is done with both rings
  // second half -> LDS in register order (wave w of group 1 and wave w of group 0 hold the same output elements in the same
  // registers: no layout arithmetic, conflict-free 4-byte accesses), first half adds and finishes the tile
  float* part = reinterpret_cast<float*>(smem) + BM * BN;       // [wave][i][j][r][lane]
  if (grp == 1) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(((wave * FM + i) * FN + j) * 4 + r) * 64 + lane] = acc[i][j][r];
  }
  __syncthreads();
  if (grp == 1) return;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] += part[(((wave * FM + i) * FN + j) * 4 + r) * 64 + lane];
  float* cs = reinterpret_cast<float*>(smem) + wave * (BM / 2) * (BN / 2);
  epilogue_stage<BM / 2, BN / 2>(acc, cs);
  epilogue_flush_common<BM / 2, BN / 2>(d, m0 + wm, n0 + wn, cs);
}

constexpr bool k2_enabled() { return true; }      // long reductions over few tiles: the 8-wave split-K kernel
constexpr int k2_min_tiles() { return 12; }       // K tiles from which the split is taken

constexpr bool lean_enabled() { return true; }    // launches with the common epilogue carry the lean flush code

constexpr int dma_stages() { return 0; }          // 0 = the built-in stage policy, see launch_kinds

constexpr bool deep_stages() { return true; }     // the 5-stage variant of the 32x64 kernel for long reductions

constexpr int deep_min_tiles() { return 16; }     // K tiles from which the 5-stage variant is taken

constexpr bool tr_enabled() { return true; }      // row-contiguous operands through ds_read_b64_tr_b16

int kind_of(const s2svc_operand& o) {
  if (o.mode == S2SVC_OP_TCONV2D_S2) return o.layout == S2SVC_LAYOUT_KC ? G_KC_TCONV2D : -1;
  const int base = o.layout == S2SVC_LAYOUT_RC ? G_RC_DENSE : G_KC_DENSE;
  return base + (o.mode == S2SVC_OP_DENSE ? 0 : (o.mode == S2SVC_OP_CONV1D ? 1 : 2));
}

// the operand-kind pairs the autograd code issues (ops/functional.py); anything else stays on the older kernels
template <int BM, int BN>
bool launch_kinds(const s2svc_gemm_desc& d, dim3 grid, hipStream_t st) {
  int ka = kind_of(d.A), kb = kind_of(d.B);
  // row-contiguous operands: k-major LDS-DMA staging + transpose reads
  if (tr_enabled()) {
    if (ka >= G_RC_DENSE && ka <= G_RC_CONV2D) ka += G_TR_DENSE - G_RC_DENSE;
    if (kb >= G_RC_DENSE && kb <= G_RC_CONV2D) kb += G_TR_DENSE - G_RC_DENSE;
  }
#define S2S_GLDS_CASE(KA, KB)                                                                       \
  if (ka == KA && kb == KB) {                                                                      \
    hipLaunchKernelGGL((gemm_glds_kernel<BM, BN, KA, KB>), grid, dim3(256), 0, st, d);              \
    return true;                                                                                   \
  }
  // all-DMA operands: the 64x64 tile runs the 3-stage counted-wait pipeline (48 KB LDS, 3 workgroups per CU); the
  // 128x128 tile keeps two stages (64 KB, 2 workgroups per CU -- a third stage leaves one workgroup per CU and
  // measured ~2x slower; BK = 32 with 3-4 stages measured 10-15 % slower).
#define S2S_DMA_CASE(KA, KB)                                                                        \
  if (ka == KA && kb == KB && !d.a_rowsum && BM == 64 && dma_stages() != 2) {                      \
    hipLaunchKernelGGL((gemm_dma_kernel<64, 64, KA, KB, 3, 64>), grid, dim3(256), 0, st, d);       \
    return true;                                                                                   \
  }
  if (ka == G_KC_DENSE && kb == G_KC_DENSE && !d.a_rowsum && BM == 64 && dma_stages() != 2 && lean_enabled() && epilogue_common_ok(d)) {
    hipLaunchKernelGGL((gemm_dma_kernel<64, 64, G_KC_DENSE, G_KC_DENSE, 3, 64, true>), grid, dim3(256), 0, st, d);
    return true;
  }
  S2S_DMA_CASE(G_KC_DENSE, G_KC_DENSE)
  S2S_DMA_CASE(G_KC_CONV1D, G_KC_DENSE)
  S2S_DMA_CASE(G_KC_CONV2D, G_KC_DENSE)
  S2S_DMA_CASE(G_KC_TCONV2D, G_KC_DENSE)
#undef S2S_DMA_CASE
  S2S_GLDS_CASE(G_KC_DENSE, G_KC_DENSE)      // linear forward, attention scores
  S2S_GLDS_CASE(G_KC_DENSE, G_RC_DENSE)      // linear dgrad, P.V
  S2S_GLDS_CASE(G_RC_DENSE, G_RC_DENSE)      // linear wgrad
  S2S_GLDS_CASE(G_RC_DENSE, G_KC_DENSE)
  S2S_GLDS_CASE(G_KC_CONV1D, G_KC_DENSE)     // Conv1d forward (implicit im2col)
  S2S_GLDS_CASE(G_KC_CONV1D, G_RC_DENSE)     // Conv1d dgrad
  S2S_GLDS_CASE(G_RC_DENSE, G_RC_CONV1D)     // Conv1d wgrad
  S2S_GLDS_CASE(G_KC_CONV2D, G_KC_DENSE)     // Conv2d 3x3 s2 forward
  S2S_GLDS_CASE(G_RC_DENSE, G_RC_CONV2D)     // Conv2d wgrad
  S2S_GLDS_CASE(G_KC_TCONV2D, G_KC_DENSE)    // Conv2d dgrad, one parity class (transposed convolution, c_map store)
  S2S_GLDS_CASE(G_TR_DENSE, G_TR_DENSE)      // linear wgrad, transpose-read path
  S2S_GLDS_CASE(G_KC_DENSE, G_TR_DENSE)      // linear dgrad (no transposed weight copy), P.V
  S2S_GLDS_CASE(G_TR_DENSE, G_KC_DENSE)
  S2S_GLDS_CASE(G_KC_CONV1D, G_TR_DENSE)     // Conv1d dgrad
  S2S_GLDS_CASE(G_TR_DENSE, G_TR_CONV1D)     // Conv1d wgrad
  S2S_GLDS_CASE(G_TR_DENSE, G_TR_CONV2D)     // Conv2d wgrad
#undef S2S_GLDS_CASE
  return false;
}

bool operand_ok(const s2svc_operand& o) {
  if (((uintptr_t)o.ptr) % 16) return false;
  if (o.ld % 8 || o.bs0 % 8 || o.bs1 % 8) return false;
  if (o.mode != S2SVC_OP_DENSE && (o.C % 8 || o.C < 64)) return false;   // taps advance by one compare-and-wrap per piece
  return true;
}

constexpr int big_min_tiles() { return 192; }     // 128x128 tiles from this many of them on

constexpr bool bm32_enabled() { return true; }

constexpr bool disabled() { return false; }

// the kernels split row indices into image positions with multiply-high divisions, exact while rows * extent < 2^32
bool rows_fit_fastdiv(const s2svc_gemm_desc& d) {
  const int64_t lim = 1ll << 32;
  const int64_t rows = (int64_t)(d.M > d.N ? d.M : d.N) + 256;
  if (d.c_map && rows * ((int64_t)d.cm_Tc * d.cm_Fc) >= lim) return false;
  const s2svc_operand* ops[2] = {&d.A, &d.B};
  for (const s2svc_operand* o : ops) {
    if (o->mode == S2SVC_OP_TCONV2D_S2 && rows * ((int64_t)o->T1 * o->F1) >= lim) return false;
    if (o->mode == S2SVC_OP_CONV2D_S2 && rows * (int64_t)(o->F2 > o->T2 ? o->F2 : o->T2) >= lim) return false;
  }
  return true;
}

bool extent_ok(const s2svc_operand& o, int extent) {
  // 16-byte pieces run along the row index (RC) or along k (KC): the extent must be a multiple of 8, or the caller
  // declares the rows zero-padded up to one (whole pieces are then fetched unmasked)
  if (extent % 8 == 0) return true;
  return o.mode == S2SVC_OP_DENSE && o.zero_padded && o.ld >= (int64_t)((extent + 7) / 8 * 8);
}

}  // namespace

// ---- class weight matrices of the stride-2 transposed convolution (see S2SVC_OP_TCONV2D_S2) ----------------------
namespace {
__global__ void tconv2d_weights_kernel(int O, int C, const float* __restrict__ w, bf16_t* __restrict__ out) {
  const int64_t n = (int64_t)9 * C * O;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    // e walks the output: class block, then [c][tap*O + o]
    const int64_t co = (int64_t)C * O;
    int cls, ntaps;
    int64_t rel;
    if (e < 4 * co) { cls = 0; ntaps = 4; rel = e; }
    else if (e < 6 * co) { cls = 1; ntaps = 2; rel = e - 4 * co; }
    else if (e < 8 * co) { cls = 2; ntaps = 2; rel = e - 6 * co; }
    else { cls = 3; ntaps = 1; rel = e - 8 * co; }
    const int pt = cls >> 1, pf = cls & 1, nf = 2 - pf;
    const int c = (int)(rel / ((int64_t)ntaps * O));
    const int ko = (int)(rel - (int64_t)c * ntaps * O);
    const int tap = ko / O, o = ko - tap * O;
    const int ta = tap / nf, fb = tap - ta * nf;
    const int kh = pt + 2 * ta, kw = pf + 2 * fb;
    out[e] = f2bf(w[(((int64_t)o * C + c) * 3 + kh) * 3 + kw]);
  }
}
}  // namespace

// the same through LDS: a workgroup takes 32 output channels x 32 input channels (x 9 taps): reads runs of 288 contiguous floats,
// writes runs of 32 contiguous bf16 values (the element-wise kernel reads 4 bytes per 13.8 KB-apart address: 13-27 us for 1.3 M
// weights, on the backward chain right in front of the data-gradient GEMMs).  O % 32 == 0 and C % 32 == 0.
__global__ __launch_bounds__(256) void tconv2d_weights_tiled_kernel(int O, int C, const float* __restrict__ w, bf16_t* __restrict__ out) {
  __shared__ float t[32][32 * 9 + 1];                 // [o][c * 9 + tap]
  const int o0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  for (int i = threadIdx.x; i < 32 * 288; i += 256) {
    const int o = i / 288, r = i - o * 288;
    t[o][r] = w[((int64_t)(o0 + o) * C + c0) * 9 + r];
  }
  __syncthreads();
  const int64_t co = (int64_t)C * O;
  // outputs: per class, [c][tap * O + o]; lanes along o (32), then (c, tap-in-class)
  const int o = threadIdx.x & 31;
  for (int j = threadIdx.x >> 5; j < 32 * 9; j += 8) {
    const int c = j / 9, k = j - c * 9;               // k enumerates (class, tap): 4 + 2 + 2 + 1
    int cls, tap, ntaps;
    int64_t cbase;
    if (k < 4) { cls = 0; tap = k; ntaps = 4; cbase = 0; }
    else if (k < 6) { cls = 1; tap = k - 4; ntaps = 2; cbase = 4 * co; }
    else if (k < 8) { cls = 2; tap = k - 6; ntaps = 2; cbase = 6 * co; }
    else { cls = 3; tap = 0; ntaps = 1; cbase = 8 * co; }
    const int pt = cls >> 1, pf = cls & 1, nf = 2 - pf;
    const int ta = tap / nf, fb = tap - ta * nf;
    const int kh = pt + 2 * ta, kw = pf + 2 * fb;
    out[cbase + ((int64_t)(c0 + c) * ntaps + tap) * O + o0 + o] = f2bf(t[o][c * 9 + kh * 3 + kw]);
  }
}

extern "C" int s2svc_tconv2d_weights(int O, int C, const float* w, void* out_bf16, void* stream) {
  S2S_REQUIRE(O > 0 && C > 0 && w && out_bf16, "tconv2d_weights: bad args");
  if (O % 32 == 0 && C % 32 == 0) {
    hipLaunchKernelGGL(tconv2d_weights_tiled_kernel, dim3(O / 32, C / 32), dim3(256), 0, (hipStream_t)stream, O, C, w, (bf16_t*)out_bf16);
    S2S_CHECK_LAUNCH("tconv2d_weights_tiled_kernel");
    return 0;
  }
  const int64_t n = (int64_t)9 * C * O;
  int nb = (int)((n + 255) / 256);
  if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(tconv2d_weights_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, O, C, w, (bf16_t*)out_bf16);
  S2S_CHECK_LAUNCH("tconv2d_weights_kernel");
  return 0;
}

// ---- grouped weight-gradient GEMMs (C[N_out, N_in] (+)= dY^T . X, both operands row-contiguous) -------------------
extern "C" int s2svc_gemm_grouped_ok(const s2svc_gemm_desc* desc) {
  const s2svc_gemm_desc& d = *desc;
  if (disabled() || d.dtype != S2S_BF16 || d.nb0 * d.nb1 != 1) return 0;
  if (kind_of(d.A) != G_RC_DENSE || kind_of(d.B) != G_RC_DENSE) return 0;
  if (!operand_ok(d.A) || !operand_ok(d.B) || !extent_ok(d.A, d.M) || !extent_ok(d.B, d.N)) return 0;
  if (d.emask || d.drop_p > 0.f || d.M <= 0 || d.N <= 0 || d.K <= 0) return 0;
  return 1;
}

// the parity-class GEMMs of one transposed convolution (Conv2d data gradient) may share a launch as well: same operand
// kinds, disjoint output pixels (c_map), reductions of 1 / 2 / 2 / 4 taps -- one grid instead of four short ones
static bool tconv_group_ok(const s2svc_gemm_desc& d) {
  if (disabled() || d.dtype != S2S_BF16 || d.nb0 * d.nb1 != 1 || !d.c_map) return false;
  if (kind_of(d.A) != G_KC_TCONV2D || kind_of(d.B) != G_KC_DENSE) return false;
  if (!operand_ok(d.A) || !operand_ok(d.B) || !extent_ok(d.A, d.K) || !extent_ok(d.B, d.K) || !rows_fit_fastdiv(d)) return false;
  if (d.emask || d.drop_p > 0.f || d.a_rowsum || d.M <= 0 || d.N <= 0 || d.K <= 0) return false;
  return true;
}

extern "C" int s2svc_gemm_grouped_try_8ph(const s2svc_gemm_desc* descs, int n, void* stream);

extern "C" int s2svc_gemm_grouped(const s2svc_gemm_desc* descs, int n, int tile, void* stream) {
  S2S_REQUIRE(descs && n > 0 && (tile == 64 || tile == 128), "gemm_grouped: bad args");
  hipStream_t st = (hipStream_t)stream;
  if (tconv_group_ok(descs[0])) {
    S2S_REQUIRE(tile == 128 && n <= S2S_GROUP_MAX, "gemm_grouped: transposed-convolution groups take 128x128 tiles, <= 10 problems");
    group_args g;
    std::memset(&g, 0, sizeof(g));
    g.n = n;
    int64_t total = 0;
    for (int i = 0; i < n; ++i) {
      S2S_REQUIRE(tconv_group_ok(descs[i]) && descs[i].splitk <= 1, "gemm_grouped: mixed or ineligible transposed-convolution descriptors");
      g.d[i] = descs[i];
      g.tile_start[i] = (int32_t)total;
      total += (int64_t)((descs[i].M + 127) / 128) * ((descs[i].N + 127) / 128);
      S2S_REQUIRE(total < (1ll << 30), "gemm_grouped: too many tiles");
    }
    for (int i = n; i <= S2S_GROUP_MAX; ++i) g.tile_start[i] = (int32_t)total;
    hipLaunchKernelGGL((gemm_grouped_kernel<128, 128, G_KC_TCONV2D, G_KC_DENSE>), dim3((unsigned)total), dim3(256), 0, st, g);
    S2S_CHECK_LAUNCH("gemm_grouped_kernel (tconv2d)");
    return 0;
  }
  for (int i0 = 0; i0 < n; i0 += S2S_GROUP_MAX) {
    const int cnt = (n - i0 < S2S_GROUP_MAX) ? n - i0 : S2S_GROUP_MAX;
    for (int i = 0; i < cnt; ++i) {
      const s2svc_gemm_desc& d = descs[i0 + i];
      S2S_REQUIRE(s2svc_gemm_grouped_ok(&d), "gemm_grouped: a descriptor is not eligible (check s2svc_gemm_grouped_ok first)");
      S2S_REQUIRE(d.splitk <= 1, "gemm_grouped: grouped problems run unsplit (splitk must be <= 1)");
    }
    int taken = 0;                                  // problems of exact 256 x 128 tiles: the 8-wave kernel (gemm_8ph.hip)
    if (tr_enabled()) {
      taken = s2svc_gemm_grouped_try_8ph(descs + i0, cnt, stream);
      if (taken < 0) return taken;
    }
    group_args g;
    std::memset(&g, 0, sizeof(g));
    int64_t total = 0;
    for (int i = 0; i < cnt; ++i) {
      if (taken & (1 << i)) continue;
      const s2svc_gemm_desc& d = descs[i0 + i];
      g.d[g.n] = d;
      g.tile_start[g.n++] = (int32_t)total;
      total += (int64_t)((d.M + tile - 1) / tile) * ((d.N + tile - 1) / tile);
      S2S_REQUIRE(total < (1ll << 30), "gemm_grouped: too many tiles");
    }
    if (g.n == 0) continue;
    for (int i = g.n; i <= S2S_GROUP_MAX; ++i) g.tile_start[i] = (int32_t)total;
    static const int cap = 0;
    if (tr_enabled() && cap > 0 && total > cap) {
      if (tile == 128)
        hipLaunchKernelGGL((gemm_grouped_capped_kernel<128, 128, G_TR_DENSE, G_TR_DENSE>), dim3((unsigned)cap), dim3(256), 0, st, g);
      else
        hipLaunchKernelGGL((gemm_grouped_capped_kernel<64, 64, G_TR_DENSE, G_TR_DENSE>), dim3((unsigned)cap), dim3(256), 0, st, g);
    } else if (tr_enabled()) {
      if (tile == 128)
        hipLaunchKernelGGL((gemm_grouped_kernel<128, 128, G_TR_DENSE, G_TR_DENSE>), dim3((unsigned)total), dim3(256), 0, st, g);
      else
        hipLaunchKernelGGL((gemm_grouped_kernel<64, 64, G_TR_DENSE, G_TR_DENSE>), dim3((unsigned)total), dim3(256), 0, st, g);
    } else if (tile == 128)
      hipLaunchKernelGGL((gemm_grouped_kernel<128, 128, G_RC_DENSE, G_RC_DENSE>), dim3((unsigned)total), dim3(256), 0, st, g);
    else
      hipLaunchKernelGGL((gemm_grouped_kernel<64, 64, G_RC_DENSE, G_RC_DENSE>), dim3((unsigned)total), dim3(256), 0, st, g);
    S2S_CHECK_LAUNCH("gemm_grouped_kernel");
  }
  return 0;
}

// One grid for batched problems of ONE operand-kind pair (see gemm_grouped_batched_kernel).  Returns 0 = launched, 1 = not
// eligible as a group (nothing was launched: the caller runs the problems one by one with s2svc_gemm), < 0 = error.
extern "C" int s2svc_gemm_grouped_batched(const s2svc_gemm_desc* descs, int n, void* stream) {
  S2S_REQUIRE(descs && n > 0, "gemm_grouped_batched: bad args");
  if (disabled() || !tr_enabled() || n > S2S_GROUP_MAX) return 1;
  const int ka = kind_of(descs[0].A), kb = kind_of(descs[0].B);
  if (!((ka == G_KC_DENSE || ka == G_RC_DENSE) && kb == G_RC_DENSE)) return 1;
  group_args g;
  std::memset(&g, 0, sizeof(g));
  int64_t total128 = 0;
  for (int i = 0; i < n; ++i) {
    const s2svc_gemm_desc& d = descs[i];
    if (d.dtype != S2S_BF16 || d.splitk > 1 || d.a_rowsum || kind_of(d.A) != ka || kind_of(d.B) != kb) return 1;
    if (!operand_ok(d.A) || !operand_ok(d.B)) return 1;
    if (!extent_ok(d.A, d.A.layout == S2SVC_LAYOUT_RC ? d.M : d.K) || !extent_ok(d.B, d.B.layout == S2SVC_LAYOUT_RC ? d.N : d.K)) return 1;
    if (!rows_fit_fastdiv(d)) return 1;
    total128 += (int64_t)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.nb0 * d.nb1;
  }
  // 128 x 128 tiles once they fill the chip (the rule of the plain launch, over the group), else 64 x 64
  static const int force_tile = 0;      // A/B aid
  const int tile = force_tile == 64 || force_tile == 128 ? force_tile : (total128 >= 256 ? 128 : 64);
  int64_t total = 0;
  for (int i = 0; i < n; ++i) {
    const s2svc_gemm_desc& d = descs[i];
    g.d[i] = d;
    g.tile_start[i] = (int32_t)total;
    total += (int64_t)((d.M + tile - 1) / tile) * ((d.N + tile - 1) / tile) * d.nb0 * d.nb1;
    S2S_REQUIRE(total < (1ll << 30), "gemm_grouped_batched: too many tiles");
  }
  g.n = n;
  for (int i = n; i <= S2S_GROUP_MAX; ++i) g.tile_start[i] = (int32_t)total;
  hipStream_t st = (hipStream_t)stream;
  if (ka == G_KC_DENSE) {
    if (tile == 128) hipLaunchKernelGGL((gemm_grouped_batched_kernel<128, 128, G_KC_DENSE, G_TR_DENSE>), dim3((unsigned)total), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_grouped_batched_kernel<64, 64, G_KC_DENSE, G_TR_DENSE>), dim3((unsigned)total), dim3(256), 0, st, g);
  } else {
    if (tile == 128) hipLaunchKernelGGL((gemm_grouped_batched_kernel<128, 128, G_TR_DENSE, G_TR_DENSE>), dim3((unsigned)total), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_grouped_batched_kernel<64, 64, G_TR_DENSE, G_TR_DENSE>), dim3((unsigned)total), dim3(256), 0, st, g);
  }
  S2S_CHECK_LAUNCH("gemm_grouped_batched_kernel");
  return 0;
}

// returns 1 if launched here (the caller still runs the split-K reduction), 0 if the problem is not eligible
extern "C" int s2svc_gemm_try_glds(const s2svc_gemm_desc* desc, void* stream) {
  const s2svc_gemm_desc& d = *desc;
  if (disabled() || d.dtype != S2S_BF16) return 0;
  if (kind_of(d.A) < 0 || kind_of(d.B) < 0 || kind_of(d.B) == G_KC_TCONV2D) return 0;
  if (!operand_ok(d.A) || !operand_ok(d.B)) return 0;
  if (!extent_ok(d.A, d.A.layout == S2SVC_LAYOUT_RC ? d.M : d.K)) return 0;
  if (!extent_ok(d.B, d.B.layout == S2SVC_LAYOUT_RC ? d.N : d.K)) return 0;
  if (!rows_fit_fastdiv(d)) return 0;
  if (d.tile_hint != 0 && d.tile_hint != 64 && d.tile_hint != 128) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int splitk = d.splitk > 1 ? d.splitk : 1;
  const int64_t tiles128 = (int64_t)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.nb0 * d.nb1 * splitk;
  bool big = tiles128 >= big_min_tiles() && d.M >= 128 && d.N >= 128;
  // under one wave of 128x128 tiles AND a short reduction: the 3-stage 64x64 kernel (3 workgroups per CU) hides the
  // few K steps better (VTN FFN 2016 x 1536 x 384: step -0.13 ms)
  if (big && tiles128 < 256 && d.K <= 384) big = false;
  if (d.tile_hint == 128) big = true;
  if (d.tile_hint == 64) big = false;
  bool launched;
  // short problems (fewer 64x64 tiles than CUs): halve the tile height, so that ~2 workgroups share a CU and cover each
  // other's load latency -- with one resident workgroup per CU the 6-step K loop of a 2016 x 384 x 384 linear is a
  // chain of exposed DMA latencies
  const int64_t tiles64 = (int64_t)((d.M + 63) / 64) * ((d.N + 63) / 64) * d.nb0 * d.nb1 * splitk;
  if (!big && bm32_enabled() && tiles64 < 256 && d.M > 64 && !d.a_rowsum && kind_of(d.A) == G_KC_DENSE && kind_of(d.B) == G_KC_DENSE) {
    dim3 grid((d.N + 63) / 64, (d.M + 31) / 32, d.nb0 * d.nb1 * splitk);
    // long reductions (K >= 1024: the feed-forward / packed-projection data gradients of VTN, 18-24 K tiles) with few workgroups:
    // five stages in flight (60 KB) instead of three -- with 2 tiles of lookahead (~0.7 us of MFMA work) every K tile waits for
    // its own DMA round trip
    const bool lean = lean_enabled() && epilogue_common_ok(d);
    if (lean && k2_enabled() && splitk == 1 && d.nb0 * d.nb1 == 1 && (d.K + 63) / 64 >= k2_min_tiles()) {
      hipLaunchKernelGGL((gemm_dma_k2_kernel<32, 64, 3>), dim3(grid.x, grid.y, 1), dim3(512), 0, st, d);
      S2S_CHECK_LAUNCH("gemm_dma_k2_kernel");
      return 1;
    }
    if (deep_stages() && (d.K + 63) / 64 / splitk >= deep_min_tiles()) {
      if (lean) hipLaunchKernelGGL((gemm_dma_kernel<32, 64, G_KC_DENSE, G_KC_DENSE, 5, 64, true>), grid, dim3(256), 0, st, d);
      else hipLaunchKernelGGL((gemm_dma_kernel<32, 64, G_KC_DENSE, G_KC_DENSE, 5, 64>), grid, dim3(256), 0, st, d);
    }
    else if (lean)
      hipLaunchKernelGGL((gemm_dma_kernel<32, 64, G_KC_DENSE, G_KC_DENSE, 3, 64, true>), grid, dim3(256), 0, st, d);
    else
      hipLaunchKernelGGL((gemm_dma_kernel<32, 64, G_KC_DENSE, G_KC_DENSE, 3, 64>), grid, dim3(256), 0, st, d);
    S2S_CHECK_LAUNCH("gemm_dma_kernel");
    return 1;
  }
  if (big) {
    dim3 grid((d.N + 127) / 128, (d.M + 127) / 128, d.nb0 * d.nb1 * splitk);
    launched = launch_kinds<128, 128>(d, grid, st);
  } else {
    dim3 grid((d.N + 63) / 64, (d.M + 63) / 64, d.nb0 * d.nb1 * splitk);
    launched = launch_kinds<64, 64>(d, grid, st);
  }
  if (!launched) return 0;
  S2S_CHECK_LAUNCH("gemm_glds_kernel");
  return 1;
}
