// bf16 MFMA GEMM, 256-row tiles, 8 wavefronts, phase-interleaved (the guide's "256^2 8-phase" structure, written for this
// library's operand descriptors).  Same contract as gemm_glds.hip for the operand kinds it takes:
//   A: K-contiguous, dense or implicit-im2col Conv2d 3x3 stride 2 (activations / output gradients),
//   B: K-contiguous dense (weights, or their transposed bf16 shadow for data gradients),  K % 64 == 0, no split-K.
//
// Why another kernel: the 128x128 / 4-wave kernel of gemm_glds.hip issues [DMA next tile | fragment reads | 32 MFMAs | drain |
// barrier] once per K tile and every wave of the workgroup is in the same phase at the same time -- its SQ counters read
// MFMA pipe busy 38 %, waves parked at s_waitcnt / barrier 36 % (profiles/r01_roofline_pmc_sq.txt).  Here
//   * a workgroup is 8 waves (2 along M x 4 along N) on a 256 x BN tile, BN = 256 or 128: two waves per SIMD, one from
//     each M half.  The halves run half a phase apart (one extra s_barrier for the second half before the loop), so while
//     one wave of a SIMD issues its 16 MFMAs the other one issues LDS reads and LDS-DMA instructions;
//   * a K tile (BK = 64) is consumed in phases of 16 MFMAs: one 64 x 32 quadrant of the wave's 128 x 64 output (BN = 256,
//     4 phases) or one 64 x 32 half of its 128 x 32 output (BN = 128, 2 phases).  A phase reads only the fragments it
//     newly needs (A rows of one M half: 8 ds_read_b128, B rows of one N half: 4) -- 0.375 LDS fragment reads per MFMA
//     instead of 0.5;
//   * staging is LDS-DMA (global_load_lds_dwordx4) in UNITS of 128 rows x 64 k (16 KB, 2 instructions per wave).  A unit
//     holds the rows that die together: A.m0 = the first 64 rows of both wave rows, A.m1 = the second, B.n0 / B.n1 = the
//     first / second 32 columns of all four wave columns.  A unit is re-filled, for the K tile two ahead, in the phase
//     after its last read; the wait is a COUNTED s_waitcnt vmcnt(6) -- three units stay in flight across every
//     barrier, nothing drains in the loop.  LDS: 2 K tiles x (256 + BN) rows x 128 B = 128 / 96 KB, one workgroup per CU.
//   Ordering rules (see MI355X_MICROARCH.md, "Two waves per SIMD", item 7): a unit is read one phase AFTER the phase whose
//   vmcnt wait (placed before that phase's first barrier) retired it; a unit is re-filled in the phase AFTER the one whose
//   reads of it were retired by lgkmcnt(0) before that phase's first barrier.  Both hold for either wave half at half a
//   phase of skew.
//   * LDS image of a unit: lane-linear per DMA instruction (8 rows x 128 B), piece c of row r at r*128 + ((c ^ ((r>>1)&7))<<4)
//     -- the XOR is applied to the per-lane SOURCE address and to the fragment read address (conflict-free ds_read_b128).
//   * epilogue: the accumulators leave through wave-private fp32 LDS tiles, 64 x 32 at a time (epilogue_tile: bias /
//     activation / dropout / residual / c_map, 16-byte stores).
#include <cstring>
#include "gemm_common.h"

namespace {

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__device__ __attribute__((aligned(16))) uint4 g_zero16_8ph = {0u, 0u, 0u, 0u};

enum { P8_DENSE = 0, P8_CONV2D = 1, P8_TCONV2D = 2, P8_CONV1D = 3 };

template <int N> __device__ __forceinline__ void p8_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void p8_wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// XCD-aware tile order (see gemm_glds.hip: XCD x gets the x-th contiguous run of tile ids) -- and, within the run, tiles walk DOWN
// groups of GH tile rows before they move to the next tile column, so that the ~32 workgroups an XCD runs side by side cover a
// compact GH x (32 / GH) block of tiles instead of a strip of one or two tile rows: the rows of A and B that block pulls through
// the XCD's L2 are GH * BM + (32 / GH) * BN instead of 2 * BM + 16 * BN (packed Q|K|V on 256 x 288 tiles: 3200 instead of 5120).
// GH = the power of two nearest sqrt(32 * BN / BM) from below in the sense GH^2 <= 2 * 32 * BN / BM; narrow grids (< 4 tile columns:
// the Conv2d front-end) keep the row-major order.  Bijective for any grid.  Measured: 8192^3 on 256 x 256 tiles 802 -> 772 us
// (54.8 -> 56.9 % of the bf16 peak), 4096^3 and everything of the two training steps unchanged (their operands fit the L2s either
// way) -- GROUPED is set for the 256 x 256 / 512 x 128 kernel only.
template <int BM, int BN, bool GROUPED = false>
__device__ __forceinline__ void p8_tile_of_block(int& bm, int& bn) {
  const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy;
  int id = blockIdx.y * gx + blockIdx.x;
  if (gridDim.z == 1 || (nwg & 7) == 0) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = id & 7, j = id >> 3;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  if (GROUPED && gx >= 4) {
    const int qe = (nwg >> 3) < 32 ? (nwg >> 3) : 32;              // workgroups of one XCD that run at the same time
    const int t2 = 2 * qe * BN / BM;
    int gh = 1;
    while (gh * 2 <= gy && 4 * gh * gh <= t2) gh *= 2;
    const int per = gh * gx, grp = id / per, within = id - grp * per;
    const int rows = (gy - grp * gh) < gh ? (gy - grp * gh) : gh;   // the last group may be short
    const int c = within / rows;
    bm = grp * gh + (within - c * rows);
    bn = c;
    return;
  }
  bm = id / gx;
  bn = id - bm * gx;
}

// byte offset of (row r, piece slot for k piece c) inside a unit
__device__ __forceinline__ int p8_off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// The scalar part of a source address: byte offset of K tile kt from the operand's base (added to the uniform base pointer).
template <int KIND>
struct P8Walk {
  // conv2d: K tile kt covers channels c0 .. c0+63 of tap (kh, kw); walked, no division in the loop.  The walk runs over the
  // 9 TAPS of one 64-channel slice before it moves to the next slice (k = tap * C + c is only the storage order of the weight
  // rows): the taps of a slice re-read the same input pixels (each is under 9/4 windows), 9 K tiles apart instead of C/64 * 9
  // -- a 512-row tile then re-touches 150 KB, not 1.6 MB, and the re-reads stay in the XCD's L2 instead of going to the
  // fabric (HBM-side traffic of the VTN front-end GEMM: 2.2 x -> see profiles/roofline_pmc.json).
  // tconv2d (one parity class of the stride-2 transposed convolution, S2SVC_OP_TCONV2D_S2): tap (kh, kw) = (ta, fb) of the
  // nt x nf taps of the class reads the output-gradient pixel (i - ta, j - fb); same order (taps innermost), the weight
  // rows are k = tap * C + c with tap = ta * nf + fb.
  // conv1d (stride 1, 'same' padding, S2SVC_OP_CONV1D: round 5): tap kh of the 2 pad + 1 taps reads row r + kh - pad of the (B T, C)
  // activations -- the row shift is part of this scalar offset, rows whose shifted frame leaves its utterance read the zero block
  // (p8_rowmask1d); taps innermost over a 64-channel slice, weight rows k = tap * C + c.
  int kt, c0, kh, kw;
  __device__ __forceinline__ void init(const s2svc_operand& o, int kt0) {
    kt = kt0;
    if (KIND == P8_CONV1D) {
      const int ntap = 2 * o.pad + 1;
      const int cs = kt0 / ntap;
      c0 = cs * 64;
      kh = kt0 - cs * ntap;
      kw = 0;
    }
    if (KIND == P8_CONV2D) {
      const int cs = kt0 / 9, tap = kt0 - cs * 9;
      c0 = cs * 64;
      kh = tap / 3;
      kw = tap - kh * 3;
    }
    if (KIND == P8_TCONV2D) {
      const int nf = 2 - (o.pad & 1), ntap = (2 - (o.pad >> 1)) * nf;
      const int cs = kt0 / ntap, tap = kt0 - cs * ntap;
      c0 = cs * 64;
      kh = tap / nf;
      kw = tap - kh * nf;
    }
  }
  __device__ __forceinline__ int64_t off_bytes(const s2svc_operand& o) const {
    if (KIND == P8_CONV2D) return ((int64_t)(kh * o.F1 + kw) * o.ld + c0) * 2;
    if (KIND == P8_TCONV2D) return (-(int64_t)(kh * o.F2 + kw) * o.ld + c0) * 2;
    if (KIND == P8_CONV1D) return ((int64_t)(kh - o.pad) * o.ld + c0) * 2;
    return (int64_t)kt * 128;
  }
  // the same K tile of the dense B operand (rows of K = taps * C elements, k = tap * C + c)
  __device__ __forceinline__ int64_t off_bytes_b(const s2svc_operand& o) const {
    if (KIND == P8_CONV2D) return ((int64_t)(kh * 3 + kw) * o.C + c0) * 2;
    if (KIND == P8_TCONV2D) return ((int64_t)(kh * (2 - (o.pad & 1)) + kw) * o.C + c0) * 2;
    if (KIND == P8_CONV1D) return ((int64_t)kh * o.C + c0) * 2;
    return (int64_t)kt * 128;
  }
  // tconv2d: the bit of this tap in the per-row validity masks (p8_rowmask)
  __device__ __forceinline__ int bit() const { return KIND == P8_CONV1D ? 1 << kh : 1 << (2 * kh + kw); }
  __device__ __forceinline__ void next(const s2svc_operand& o) {
    ++kt;
    if (KIND == P8_CONV2D) {
      if (++kw == 3) {
        kw = 0;
        if (++kh == 3) {
          kh = 0;
          c0 += 64;
        }
      }
    }
    if (KIND == P8_CONV1D) {
      if (++kh == 2 * o.pad + 1) {
        kh = 0;
        c0 += 64;
      }
    }
    if (KIND == P8_TCONV2D) {
      if (++kw == 2 - (o.pad & 1)) {
        kw = 0;
        if (++kh == 2 - (o.pad >> 1)) {
          kh = 0;
          c0 += 64;
        }
      }
    }
  }
};

// per-lane byte offset of tile row `r` (clamped into the matrix), source piece `c`
template <int KIND>
__device__ __forceinline__ uint32_t p8_rowoff(const s2svc_operand& o, int r, int R, int c) {
  if (r >= R) r = R - 1;            // rows past the matrix: their products only reach C rows / columns that are never stored
  int64_t e;
  if (KIND == P8_CONV2D) {
    const int bt = r / o.F2, f2 = r - bt * o.F2;
    const int b = bt / o.T2, t2 = bt - b * o.T2;
    e = ((int64_t)(b * o.T1 + 2 * t2) * o.F1 + 2 * f2) * o.ld;
  } else if (KIND == P8_TCONV2D) {     // class grid (o.T1 x o.F1) -> output-gradient pixel (i, j) of the (B, T2, F2, C) tensor
    const int per_b = o.T1 * o.F1;
    const int b = r / per_b, rem = r - b * per_b;
    const int i = rem / o.F1, j = rem - i * o.F1;
    e = ((int64_t)(b * o.T2 + i) * o.F2 + j) * o.ld;
  } else {
    e = (int64_t)r * o.ld;
  }
  return (uint32_t)((e + c * 8) * 2);
}

// tconv2d: which of the <= 2 x 2 taps of the parity class read inside the output-gradient image for tile row r
// (bit 2 * ta + fb); rows past the matrix read nothing
__device__ __forceinline__ int p8_rowmask(const s2svc_operand& o, int r, int R) {
  if (r >= R) return 0;
  const int per_b = o.T1 * o.F1;
  const int rem = r - (r / per_b) * per_b;
  const int i = rem / o.F1, j = rem - i * o.F1;
  int m = 0;
#pragma unroll
  for (int ta = 0; ta < 2; ++ta)
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
      if (i - ta >= 0 && i - ta < o.T2 && j - fb >= 0 && j - fb < o.F2) m |= 1 << (2 * ta + fb);
  return m;
}

// conv1d: which of the 2 pad + 1 taps stay inside the utterance of tile row r (bit = tap)
__device__ __forceinline__ int p8_rowmask1d(const s2svc_operand& o, int r, int R) {
  if (r >= R) return 0;
  const int t = r - (r / o.T) * o.T;
  int m = 0;
  for (int tap = 0; tap <= 2 * o.pad; ++tap)
    if (t + tap - o.pad >= 0 && t + tap - o.pad < o.T) m |= 1 << tap;
  return m;
}
template <int KA>
__device__ __forceinline__ int p8_rowmask_of(const s2svc_operand& o, int r, int R) {
  return KA == P8_TCONV2D ? p8_rowmask(o, r, R) : KA == P8_CONV1D ? p8_rowmask1d(o, r, R) : 0;
}

// one unit = NI DMA instructions of this wave; base == nullptr: the unit lies past the last K tile (the instructions are
// still issued, from a zero block, so that the counted waits stay exact)
template <int NI>
__device__ __forceinline__ void p8_issue(const char* base, const uint32_t (&off)[NI], char* lds_unit, int wave_s) {
  const char* b = base ? base : reinterpret_cast<const char*>(&g_zero16_8ph);
#pragma unroll
  for (int e = 0; e < NI; ++e) {
    const uint32_t o = base ? off[e] : 0u;
    __builtin_amdgcn_global_load_lds((gbl_void*)(b + o), (lds_void*)(lds_unit + (wave_s * NI + e) * 1024), 16, 0, 0);
  }
}

// the same with a per-lane validity test (tconv2d: the tap of this K tile falls outside the image for some rows): those
// lanes read the zero block
template <int NI>
__device__ __forceinline__ void p8_issue_masked(const char* base, const uint32_t (&off)[NI], const int (&mask)[NI], int bit,
                                                char* lds_unit, int wave_s) {
  const char* z = reinterpret_cast<const char*>(&g_zero16_8ph);
#pragma unroll
  for (int e = 0; e < NI; ++e) {
    // (the select costs: the aligner's Conv1d 4096 x 1536 x 4608 takes 85.7 us with it and 68.6 us with no masking at all; a wave-uniform
    // "every row of this instruction is valid" fast path was built and measured -- no gain for the Conv1d, the transposed-convolution classes
    // 192 -> 214 us: the loop's scalar registers spill and the per-instruction branches break its schedule -- removed)
    const char* src = (base && (mask[e] & bit)) ? base + off[e] : z;
    __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(lds_unit + (wave_s * NI + e) * 1024), 16, 0, 0);
  }
}

// A units go through this: masked for the transposed-convolution operand only
template <int KA, int NI>
__device__ __forceinline__ void p8_issue_a(const char* base, const uint32_t (&off)[NI], const int (&mask)[NI], int bit, char* lds_unit,
                                           int wave_s) {
  if (KA == P8_TCONV2D || KA == P8_CONV1D) p8_issue_masked<NI>(base, off, mask, bit, lds_unit, wave_s);
  else p8_issue<NI>(base, off, lds_unit, wave_s);
}

// fragment reads of one unit: NF row blocks of 16 rows starting at unit row r0, both k halves of the tile
template <int NF>
__device__ __forceinline__ void p8_read(const char* unit, int rd_base, int p0, int p1, bf16x8_t (&f)[NF][2]) {
#pragma unroll
  for (int i = 0; i < NF; ++i) {
    f[i][0] = *reinterpret_cast<const bf16x8_t*>(unit + rd_base + i * 2048 + p0);
    f[i][1] = *reinterpret_cast<const bf16x8_t*>(unit + rd_base + i * 2048 + p1);
  }
}

template <int NA, int NB>
__device__ __forceinline__ void p8_mfma(const bf16x8_t (&a)[NA][2], const bf16x8_t (&b)[NB][2], f32x4_t (&acc)[NA][NB]) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][ks], b[j][ks], acc[i][j], 0, 0, 0);
  __builtin_amdgcn_s_setprio(0);
}

#define P8_PHASE_SYNC_IN()                 \
  p8_wait_lgkm0();                         \
  __builtin_amdgcn_sched_barrier(0);       \
  __builtin_amdgcn_s_barrier();            \
  __builtin_amdgcn_sched_barrier(0)
#define P8_PHASE_SYNC_OUT()                \
  __builtin_amdgcn_sched_barrier(0);       \
  __builtin_amdgcn_s_barrier();            \
  __builtin_amdgcn_sched_barrier(0)

// ---------------------------------------------------------------------------------------------------------
// 4 phases per K tile, wave tile 128 x 64.  (WR, WC) = wave rows x wave columns: (2, 4) -> 256 x 256 tile, 128 KB LDS;
// (4, 2) -> 512 x 128 tile for outputs with few columns (N = 384: 3 column tiles, 225 workgroups on 256 CUs), 160 KB LDS.
// Units: A.m0 / A.m1 = WR * 64 rows (WR DMA instructions per wave), B.n0 / B.n1 = WC * 32 rows (WC / 2 per wave).
// ---------------------------------------------------------------------------------------------------------
template <int KA, int WR, int WC, bool STAGGER, bool LEAN = false>
__global__ __launch_bounds__(512) void gemm_8ph_kernel_q(const s2svc_gemm_desc d) {
  static_assert(WR * WC == 8 && (WC == 2 || WC == 4), "8 waves");
  constexpr int BM = WR * 128, BN = WC * 64;
  constexpr int NIA = WR, NIB = WC / 2;                      // DMA instructions per wave and unit
  constexpr int UA = WR * 64 * 128, UB = WC * 32 * 128;      // unit bytes
  constexpr int BUF = 2 * UA + 2 * UB;                       // A.m0 | A.m1 | B.n0 | B.n1
  constexpr int OA0 = 0, OA1 = UA, OB0 = 2 * UA, OB1 = 2 * UA + UB;
  __shared__ __attribute__((aligned(1024))) char smem[2 * BUF];
  const int zb = blockIdx.z;
  const int z0 = zb / d.nb1, z1 = zb - z0 * d.nb1;
  int tile_m, tile_n;
  p8_tile_of_block<BM, BN, true>(tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const char* Ab = reinterpret_cast<const char*>((const bf16_t*)d.A.ptr + (int64_t)z0 * d.A.bs0 + (int64_t)z1 * d.A.bs1);
  const char* Bb = reinterpret_cast<const char*>((const bf16_t*)d.B.ptr + (int64_t)z0 * d.B.bs0 + (int64_t)z1 * d.B.bs1);
  const int nt = d.K / 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int half = wave >> 2;                                 // waves w and w + 4 share a SIMD: the two halves run skewed
  const int lr = lane & 15, lg = lane >> 4;

  // source offsets of this wave's DMA instructions per unit
  uint32_t offA[2][NIA], offB[2][NIB];
  int mA[2][NIA];                                                 // tconv2d: taps inside the image, per row
#pragma unroll
  for (int e = 0; e < NIA; ++e) {
    const int ru = (wave * NIA + e) * 8 + (lane >> 3);            // unit row of this lane
    const int c = (lane & 7) ^ ((ru >> 1) & 7);                   // source piece that belongs in this lane's slot
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = m0 + (ru >> 6) * 128 + h * 64 + (ru & 63);
      offA[h][e] = p8_rowoff<KA>(d.A, row, d.M, c);
      mA[h][e] = p8_rowmask_of<KA>(d.A, row, d.M);
    }
  }
#pragma unroll
  for (int e = 0; e < NIB; ++e) {
    const int ru = (wave * NIB + e) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((ru >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) offB[h][e] = p8_rowoff<P8_DENSE>(d.B, n0 + (ru >> 5) * 64 + h * 32 + (ru & 31), d.N, c);
  }
  // fragment read addresses
  const int sw = (lr >> 1) & 7;
  const int p0 = (lg ^ sw) << 4, p1 = ((4 + lg) ^ sw) << 4;
  const int rdA = (wr * 64 + lr) * 128, rdB = (wc * 32 + lr) * 128;

  f32x4_t acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  P8Walk<KA> wa, wb;      // K tile of the next A unit to issue (t + 2 in the loop); wb: one behind (B.n0 of tile t + 1)
  wa.init(d.A, 0);
  wb.init(d.A, 0);
  // prologue: tile 0 complete, tile 1 without B.n0 (issued in phase 1 of tile 0)
  {
    const char* a0 = Ab + wa.off_bytes(d.A);
    const int bit0 = wa.bit();
    p8_issue_a<KA, NIA>(a0, offA[0], mA[0], bit0, smem + OA0, wave);
    p8_issue<NIB>(Bb, offB[0], smem + OB0, wave);
    p8_issue<NIB>(Bb, offB[1], smem + OB1, wave);
    p8_issue_a<KA, NIA>(a0, offA[1], mA[1], bit0, smem + OA1, wave);
    wa.next(d.A);
    wb.next(d.A);
    const bool has1 = nt > 1;
    const char* a1 = has1 ? Ab + wa.off_bytes(d.A) : nullptr;
    const char* b1 = has1 ? Bb + wa.off_bytes_b(d.A) : nullptr;
    const int bit1 = wa.bit();
    p8_issue_a<KA, NIA>(a1, offA[0], mA[0], bit1, smem + BUF + OA0, wave);
    p8_issue<NIB>(b1, offB[1], smem + BUF + OB1, wave);
    p8_issue_a<KA, NIA>(a1, offA[1], mA[1], bit1, smem + BUF + OA1, wave);
    wa.next(d.A);
  }
  p8_wait_vmcnt<2 * NIA + NIB>();
  __builtin_amdgcn_s_barrier();
  if (STAGGER && half == 1) __builtin_amdgcn_s_barrier();

  bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * BUF;
    char* oth = smem + ((t & 1) ^ 1) * BUF;
    const char* a2 = (t + 2 < nt) ? Ab + wa.off_bytes(d.A) : nullptr;        // A units of tile t + 2
    const int bit2 = wa.bit();
    const char* b2 = (t + 2 < nt) ? Bb + wa.off_bytes_b(d.A) : nullptr;
    const char* b1 = (t + 1 < nt) ? Bb + wb.off_bytes_b(d.A) : nullptr;
    // ---- phase 1: quadrant (m0, n0)
    p8_read<2>(cur + OB0, rdB, p0, p1, fb0);
    p8_read<4>(cur + OA0, rdA, p0, p1, fa);
    p8_issue<NIB>(b1, offB[0], oth + OB0, wave);                              // B.n0 of tile t + 1
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 2>(fa, fb0, acc[0][0]);
    P8_PHASE_SYNC_OUT();
    // ---- phase 2: quadrant (m0, n1)
    p8_read<2>(cur + OB1, rdB, p0, p1, fb1);
    p8_issue_a<KA, NIA>(a2, offA[0], mA[0], bit2, cur + OA0, wave);            // A.m0 of tile t + 2
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 2>(fa, fb1, acc[0][1]);
    P8_PHASE_SYNC_OUT();
    // ---- phase 3: quadrant (m1, n1)
    p8_read<4>(cur + OA1, rdA, p0, p1, fa);
    p8_issue<NIB>(b2, offB[1], cur + OB1, wave);                              // B.n1 of tile t + 2
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 2>(fa, fb1, acc[1][1]);
    P8_PHASE_SYNC_OUT();
    // ---- phase 4: quadrant (m1, n0)   (fb0 is still live: B.n0 needs no second read, its unit died after phase 1 --
    //      it is nevertheless re-filled only in phase 1 of the next tile, keeping one unit per phase)
    p8_issue_a<KA, NIA>(a2, offA[1], mA[1], bit2, cur + OA1, wave);            // A.m1 of tile t + 2
    p8_wait_vmcnt<2 * NIA + NIB>();                                           // everything of tile t + 1 has landed
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 2>(fa, fb0, acc[1][0]);
    P8_PHASE_SYNC_OUT();
    wa.next(d.A);
    wb.next(d.A);
  }
  if (STAGGER && half == 0) __builtin_amdgcn_s_barrier();
  p8_wait_vmcnt<0>();                  // the zero-block DMAs issued past the end
  __syncthreads();                     // every wave is done with the operand stages: reuse them as fp32 C tiles
  float* cs = reinterpret_cast<float*>(smem) + wave * (64 * 32);
  const bool cmap_lean = KA == P8_TCONV2D && d.tile_hint != 65 && epilogue_cmap_ok(d);      // uniform (tile_hint 65: A/B aid, the general flush)
  const c_map_t cmq = c_map_make(d);
  // the four 64 x 32 sub-tiles in a ROLLED loop: staging is per sub-tile (constant register indices), the flush code exists once
#pragma unroll 1
  for (int q = 0; q < 4; ++q) {
    switch (q) {
      case 0: epilogue_stage<64, 32>(acc[0][0], cs); break;
      case 1: epilogue_stage<64, 32>(acc[0][1], cs); break;
      case 2: epilogue_stage<64, 32>(acc[1][0], cs); break;
      default: epilogue_stage<64, 32>(acc[1][1], cs); break;
    }
    if (LEAN) epilogue_flush_common<64, 32>(d, m0 + wr * 128 + (q >> 1) * 64, n0 + wc * 64 + (q & 1) * 32, cs);
    else if (KA == P8_TCONV2D && cmap_lean) epilogue_flush_cmap<64, 32>(d, cmq, m0 + wr * 128 + (q >> 1) * 64, n0 + wc * 64 + (q & 1) * 32, cs);
    else epilogue_flush<64, 32>(d, z0, z1, m0 + wr * 128 + (q >> 1) * 64, n0 + wc * 64 + (q & 1) * 32, cs, 1, 0, zb);
  }
}

// ---------------------------------------------------------------------------------------------------------
// BN = 128: 2 phases per K tile (wave tile 128 x 32: a phase = 64 rows x 32 columns x K 64)
// ---------------------------------------------------------------------------------------------------------
template <int KA, bool STAGGER, int LEAN = 0>            // LEAN: 0 = the general epilogue, 1 = the common one, 2 = the Swish pair
__global__ __launch_bounds__(512) void gemm_8ph_kernel_128(const s2svc_gemm_desc d) {
  constexpr int UNIT = 16384, BUF = 3 * UNIT;            // A.m0 | A.m1 | B
  __shared__ __attribute__((aligned(1024))) char smem[2 * BUF];
  const int zb = blockIdx.z;
  const int z0 = zb / d.nb1, z1 = zb - z0 * d.nb1;
  int tile_m, tile_n;
  p8_tile_of_block<256, 128>(tile_m, tile_n);
  const int m0 = tile_m * 256, n0 = tile_n * 128;
  const char* Ab = reinterpret_cast<const char*>((const bf16_t*)d.A.ptr + (int64_t)z0 * d.A.bs0 + (int64_t)z1 * d.A.bs1);
  const char* Bb = reinterpret_cast<const char*>((const bf16_t*)d.B.ptr + (int64_t)z0 * d.B.bs0 + (int64_t)z1 * d.B.bs1);
  const int nt = d.K / 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;

  uint32_t offA[2][2], offB[2];
  int mA[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int ru = (wave * 2 + e) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((ru >> 1) & 7);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = m0 + (ru >> 6) * 128 + h * 64 + (ru & 63);
      offA[h][e] = p8_rowoff<KA>(d.A, row, d.M, c);
      mA[h][e] = p8_rowmask_of<KA>(d.A, row, d.M);
    }
    offB[e] = p8_rowoff<P8_DENSE>(d.B, n0 + ru, d.N, c);
  }
  const int sw = (lr >> 1) & 7;
  const int p0 = (lg ^ sw) << 4, p1 = ((4 + lg) ^ sw) << 4;
  const int rdA = (wr * 64 + lr) * 128, rdB = (wc * 32 + lr) * 128;

  f32x4_t acc[2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[a][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  P8Walk<KA> wa0, wa1;    // K tile of the next A.m0 / A.m1 unit to issue (A.m1 runs one tile behind A.m0 in issue order)
  wa0.init(d.A, 0);
  wa1.init(d.A, 0);
  {
    const char* a0 = Ab + wa0.off_bytes(d.A);
    const int bit0 = wa0.bit();
    p8_issue_a<KA, 2>(a0, offA[0], mA[0], bit0, smem + 0 * UNIT, wave);
    p8_issue<2>(Bb, offB, smem + 2 * UNIT, wave);
    p8_issue_a<KA, 2>(a0, offA[1], mA[1], bit0, smem + 1 * UNIT, wave);
    wa0.next(d.A);
    wa1.next(d.A);
    const bool has1 = nt > 1;
    p8_issue_a<KA, 2>(has1 ? Ab + wa0.off_bytes(d.A) : nullptr, offA[0], mA[0], wa0.bit(), smem + BUF + 0 * UNIT, wave);
    p8_issue<2>(has1 ? Bb + wa1.off_bytes_b(d.A) : nullptr, offB, smem + BUF + 2 * UNIT, wave);
    wa0.next(d.A);
  }
  p8_wait_vmcnt<4>();                  // tile 0 has landed (A.m0, B of tile 1 may still be moving)
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();

  bf16x8_t fa[4][2], fb[2][2];
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * BUF;
    char* oth = smem + ((t & 1) ^ 1) * BUF;
    // ---- phase 1: rows m0
    p8_read<2>(cur + 2 * UNIT, rdB, p0, p1, fb);
    p8_read<4>(cur + 0 * UNIT, rdA, p0, p1, fa);
    p8_issue_a<KA, 2>((t + 1 < nt) ? Ab + wa1.off_bytes(d.A) : nullptr, offA[1], mA[1], wa1.bit(), oth + 1 * UNIT, wave);     // A.m1 of tile t + 1
    p8_wait_vmcnt<6>();                                                                             // A.m1 of tile t has landed
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 2>(fa, fb, acc[0]);
    P8_PHASE_SYNC_OUT();
    // ---- phase 2: rows m1
    p8_read<4>(cur + 1 * UNIT, rdA, p0, p1, fa);
    p8_issue_a<KA, 2>((t + 2 < nt) ? Ab + wa0.off_bytes(d.A) : nullptr, offA[0], mA[0], wa0.bit(), cur + 0 * UNIT, wave);     // A.m0 of tile t + 2
    p8_issue<2>((t + 2 < nt) ? Bb + wa0.off_bytes_b(d.A) : nullptr, offB, cur + 2 * UNIT, wave);      // B of tile t + 2
    p8_wait_vmcnt<6>();                                                                             // A.m0, B of tile t + 1 have landed
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 2>(fa, fb, acc[1]);
    P8_PHASE_SYNC_OUT();
    wa0.next(d.A);
    wa1.next(d.A);
  }
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();
  p8_wait_vmcnt<0>();
  __syncthreads();
  float* cs = reinterpret_cast<float*>(smem) + wave * (64 * 32);
  const bool cmap_lean = KA == P8_TCONV2D && d.tile_hint != 65 && epilogue_cmap_ok(d);      // uniform
  const c_map_t cmq = c_map_make(d);
#pragma unroll 1
  for (int a = 0; a < 2; ++a) {          // rolled: one copy of the flush code (see epilogue_stage)
    if (a == 0) epilogue_stage<64, 32>(acc[0], cs);
    else epilogue_stage<64, 32>(acc[1], cs);
    if (LEAN == 2) epilogue_flush_swish<64, 32>(d, m0 + wr * 128 + a * 64, n0 + wc * 32, cs);
    else if (LEAN == 1) epilogue_flush_common<64, 32>(d, m0 + wr * 128 + a * 64, n0 + wc * 32, cs);       // (gemm_common.h: a fraction of the code)
    else if (KA == P8_TCONV2D && cmap_lean) epilogue_flush_cmap<64, 32>(d, cmq, m0 + wr * 128 + a * 64, n0 + wc * 32, cs);
    else epilogue_flush<64, 32>(d, z0, z1, m0 + wr * 128 + a * 64, n0 + wc * 32, cs, 1, 0, zb);
  }
}

// ---------------------------------------------------------------------------------------------------------
// 256 x (96 * PH) tiles, PH = 2 / 3: the geometry that gives M = 4096, N = 1536 * PH exactly 256 workgroups (one per CU, one round) --
// the packed Q|K|V projection of the AAS-VC decoder (N = 4608) is 2.25 rounds of 256 x 128 tiles otherwise and pays for 3.
// 8 waves as 4 (rows) x 2 (columns), wave tile 64 x (48 * PH); a K tile is PH phases of 4 x 3 fragments (24 MFMAs): phase 0
// reads the wave's 4 A fragments (kept for the whole K tile) and the B fragments of its first 48 columns, phase p those of
// columns 48 p .. 48 p + 47: (4 + 3 PH) / (12 PH) fragment reads per MFMA pair (0.36 at PH = 3).  Units: A = the 256 rows
// (4 DMA instructions per wave), B.p = the 96 rows of phase p (rows 0..47: wave column 0, 48..95: wave column 1) = 12 DMA
// instructions: waves 0..5 issue two each, waves 6 and 7 issue two from the zero block into 32 pad rows of the unit so that
// the counted waits are the same in every wave (letting them issue nothing and wait for their A units only measured the same:
// 52.7 vs 53.5, 51.3 vs 51.8 us).  Two stages; a unit is re-filled in the phase after the one
// that read it, for the K tile two ahead (the rules of the kernels above); every phase waits for what the NEXT phase reads.
// Dense operands, one problem, the common epilogue.
// ---------------------------------------------------------------------------------------------------------
template <int PH, bool STAGGER>
__global__ __launch_bounds__(512) void gemm_8ph_kernel_n96(const s2svc_gemm_desc d) {
  static_assert(PH == 2 || PH == 3, "PH");   // (PH = 1, 256 x 96 on three stages, was built and measured: 28.8 vs 25.3 us at 4096 x 1536 x 1536)
  constexpr int WN = 48 * PH, BN = 96 * PH;
  constexpr int UA = 256 * 128, UB = 128 * 128;            // B unit: 96 rows + 32 pad rows
  constexpr int BUF = UA + PH * UB;
  __shared__ __attribute__((aligned(1024))) char smem[2 * BUF];
  int tile_m, tile_n;
  p8_tile_of_block<256, BN>(tile_m, tile_n);
  const int m0 = tile_m * 256, n0 = tile_n * BN;
  const char* Ab = reinterpret_cast<const char*>(d.A.ptr);
  const char* Bb = reinterpret_cast<const char*>(d.B.ptr);
  const int nt = d.K / 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int half = wave >> 2;
  const int lr = lane & 15, lg = lane >> 4;
  const bool bwave = wave < 6;                              // this wave's B instructions move rows (the others: pad rows)

  uint32_t offA[4], offB[PH][2];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int ru = (wave * 4 + e) * 8 + (lane >> 3);
    offA[e] = p8_rowoff<P8_DENSE>(d.A, m0 + ru, d.M, (lane & 7) ^ ((ru >> 1) & 7));
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int ru = (wave * 2 + e) * 8 + (lane >> 3);        // unit row; 96 .. 127: pad
    const int c = (lane & 7) ^ ((ru >> 1) & 7);
    const int w = ru >= 48 ? 1 : 0, j = ru - 48 * w;
#pragma unroll
    for (int p = 0; p < PH; ++p) offB[p][e] = bwave ? p8_rowoff<P8_DENSE>(d.B, n0 + w * WN + p * 48 + j, d.N, c) : 0u;
  }
  const int sw = (lr >> 1) & 7;
  const int p0 = (lg ^ sw) << 4, p1 = ((4 + lg) ^ sw) << 4;
  const int rdA = (wr * 64 + lr) * 128, rdB = (wc * 48 + lr) * 128;

  f32x4_t acc[PH][4][3];
#pragma unroll
  for (int p = 0; p < PH; ++p)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[p][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

#define N96_A(T, BUFP) p8_issue<4>((T) < nt ? Ab + (int64_t)(T) * 128 : nullptr, offA, (BUFP), wave)
#define N96_B(T, P, BUFP) p8_issue<2>(((T) < nt && bwave) ? Bb + (int64_t)(T) * 128 : nullptr, offB[P], (BUFP) + UA + (P) * UB, wave)
#define N96_WAIT(N, A_DUE) p8_wait_vmcnt<(N)>()
  // prologue, in the order of the steady state: tile 0, then of tile 1 everything but its last B unit
  N96_A(0, smem);
#pragma unroll
  for (int p = 0; p < PH; ++p) N96_B(0, p, smem);
  N96_A(1, smem + BUF);
#pragma unroll
  for (int p = 0; p + 1 < PH; ++p) N96_B(1, p, smem + BUF);
  // A + B.0 of tile 0 have landed: what was issued behind them may still be moving
  N96_WAIT(PH == 2 ? 8 : 12, true);
  __builtin_amdgcn_s_barrier();
  if (STAGGER && half == 1) __builtin_amdgcn_s_barrier();

  bf16x8_t fa[4][2], fb[3][2];
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * BUF;
    char* oth = smem + ((t & 1) ^ 1) * BUF;
    // ---- phase 0: columns 0 .. 47 of the wave
    p8_read<3>(cur + UA, rdB, p0, p1, fb);
    p8_read<4>(cur, rdA, p0, p1, fa);
    N96_B(t + 1, PH - 1, oth);                              // the last B unit of tile t + 1 (read in the last phase of tile t - 1)
    N96_WAIT(PH == 2 ? 8 : 12, false);                      // B.1 of tile t has landed
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 3>(fa, fb, acc[0]);
    P8_PHASE_SYNC_OUT();
    // ---- phase 1
    p8_read<3>(cur + UA + UB, rdB, p0, p1, fb);
    N96_A(t + 2, cur);                                      // A, B.0 of tile t + 2
    N96_B(t + 2, 0, cur);
    N96_WAIT(PH == 2 ? 8 : 16, PH == 2);                    // PH = 2: A, B.0 of tile t + 1; PH = 3: B.2 of tile t
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 3>(fa, fb, acc[1]);
    P8_PHASE_SYNC_OUT();
    if (PH == 3) {
      // ---- phase 2
      p8_read<3>(cur + UA + 2 * UB, rdB, p0, p1, fb);
      N96_B(t + 2, 1, cur);                                 // B.1 of tile t + 2
      N96_WAIT(12, true);                                   // A, B.0 of tile t + 1
      P8_PHASE_SYNC_IN();
      p8_mfma<4, 3>(fa, fb, acc[PH - 1]);
      P8_PHASE_SYNC_OUT();
    }
  }
#undef N96_WAIT
#undef N96_A
#undef N96_B
  if (STAGGER && half == 0) __builtin_amdgcn_s_barrier();
  p8_wait_vmcnt<0>();
  __syncthreads();
  float* cs = reinterpret_cast<float*>(smem) + wave * (64 * 32);
  const int mw = m0 + wr * 64, nw = n0 + wc * WN;
  // the wave's 3 PH column fragments leave in pairs (64 x 32) and, for an odd count, one single (64 x 16); rolled, one copy of
  // each flush
  constexpr int NJ = 3 * PH;
#pragma unroll 1
  for (int q = 0; q < NJ / 2; ++q) {
    f32x4_t t2[4][2];
#pragma unroll
    for (int qq = 0; qq < NJ / 2; ++qq)
      if (qq == q) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          t2[i][0] = acc[(2 * qq) / 3][i][(2 * qq) % 3];
          t2[i][1] = acc[(2 * qq + 1) / 3][i][(2 * qq + 1) % 3];
        }
      }
    epilogue_stage<64, 32>(t2, cs);
    epilogue_flush_common<64, 32>(d, mw, nw + q * 32, cs);
  }
  if (NJ & 1) {
    f32x4_t t1[4][1];
#pragma unroll
    for (int i = 0; i < 4; ++i) t1[i][0] = acc[PH - 1][i][2];
    epilogue_stage<64, 16>(t1, cs);
    epilogue_flush_common<64, 16>(d, mw, nw + (NJ - 1) * 16, cs);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Row-contiguous operands ("TR"): the weight-gradient GEMMs C[n_out, n_in] (+)= dY^T . X of the Linear layers, reduction over
// the B*T rows of the batch, both operands stored [k][row].  Same 256 x 128 tile, units, phases and counted waits as
// gemm_8ph_kernel_128; what changes is the image of a unit and how fragments leave it (as in gemm_glds.hip's TrStage):
//   * a unit (128 rows x 64 k) is 16 pieces of 1 KiB, one DMA instruction each: piece s = (64-row half s >> 3, k block s & 7)
//     holds 8 k rows x 64 rows, a k row = 128 contiguous bytes of the operand = 8 slots of 16 bytes (round 4; until round 3 a
//     piece was [32 k][16 rows]: 32 bytes per k row, i.e. 32 cache lines per DMA instruction instead of 8 -- the K loop of the
//     transposed kernels ran at 1.05 us per K tile against 0.72 us for the K-contiguous kernel of the same schedule).  Slot j
//     of k row kin holds the 16-row block (j >> 1) ^ ((kin >> 1) & 3), rows 8 (j & 1) .. + 7 of it: the XOR (applied to the
//     per-lane SOURCE address and to the fragment address) spreads the four k rows a 16-lane group reads over all banks;
//   * a fragment is two ds_read_b64_tr_b16 (lane group g gets k = 4g .. 4g+3 and 16+4g .. 16+4g+3 of the k half -- a
//     permutation of the MFMA k positions, the same for both operands);
//   * the fused bias gradient (a_rowsum: sum over k of every A row) is MFMA work too -- an extra product with an all-ones B
//     fragment, the (K tile, k half) pairs dealt round-robin to the four waves of a wave row (+12.5 % MFMAs in the workgroups of
//     the first column tile only), partial sums combined through LDS in a fixed order.
// Exact tiles only (M % 256 == N % 128 == K % 64 == 0): AAS-VC's decoder (1536 / 3072 / 4608 features, 4096 rows).
// The 4-wave 128 x 128 kernel it replaces there runs these at 17-25 % of the MFMA peak (one barrier per K tile, every wave in
// the same phase); VTN's 384-feature layers keep it (grouped, 64 x 64 tiles).
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short p8_s16x4_t;
typedef __attribute__((address_space(3))) p8_s16x4_t p8_lds_s16x4_t;

// source of DMA piece s (0 .. 15) of a unit for this lane: its k row inside the K tile and the unit row of its 8 values
__device__ __forceinline__ void p8_tr_src(int s, int lane, int& k, int& ur) {
  const int kin = lane >> 3, j = lane & 7;
  k = (s & 7) * 8 + kin;
  ur = (s >> 3) * 64 + (((j >> 1) ^ ((kin >> 1) & 3)) << 4) + (j & 1) * 8;
}

// per-lane byte offset of the fragment reads of 16-row block b (0 .. 7) inside a unit, first k half, first 16 k
__device__ __forceinline__ int p8_tr_frag_off(int b, int lane) {
  const int g = lane >> 4, r = lane & 15;
  const int kin = (g & 1) * 4 + (r >> 2);               // k row inside its block of 8 (k = 4 g + (r >> 2) of the 16)
  return (b >> 2) * 8192 + (g >> 1) * 1024 + kin * 128 + ((((b & 3) ^ ((kin >> 1) & 3))) << 5) + (r & 3) * 8;
}

// fragments of NF 16-row blocks (per-lane offsets foff[], see p8_tr_frag_off), both k halves
template <int NF>
__device__ __forceinline__ void p8_read_tr(const char* unit, const int (&foff)[NF], bf16x8_t (&f)[NF][2]) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const char* p = unit + foff[i] + ks * 4096;
      const p8_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((p8_lds_s16x4_t*)p);
      const p8_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((p8_lds_s16x4_t*)(p + 2048));
      const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      f[i][ks] = __builtin_bit_cast(bf16x8_t, v);
    }
}

template <bool STAGGER>
__device__ __forceinline__ void p8_tr_tile(const s2svc_gemm_desc& d, int tile_m, int tile_n, char* smem) {
  constexpr int UNIT = 16384, BUF = 3 * UNIT;            // A.m0 | A.m1 | B
  const int m0 = tile_m * 256, n0 = tile_n * 128;
  const char* Ab = reinterpret_cast<const char*>(d.A.ptr);
  const char* Bb = reinterpret_cast<const char*>(d.B.ptr);
  const int nt = d.K / 64;
  const int64_t stepA = (int64_t)d.A.ld * 128, stepB = (int64_t)d.B.ld * 128;      // bytes per K tile (64 k rows)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // source offsets: DMA instruction s = wave * 2 + e of a unit = piece s (p8_tr_src)
  uint32_t offA[2][2], offB[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    int k, ur;                                            // k row inside the K tile, unit row of the lane's 8 values
    p8_tr_src(wave * 2 + e, lane, k, ur);
#pragma unroll
    for (int h = 0; h < 2; ++h) offA[h][e] = (uint32_t)(((int64_t)k * d.A.ld + m0 + (ur >> 6) * 128 + h * 64 + (ur & 63)) * 2);
    offB[e] = (uint32_t)(((int64_t)k * d.B.ld + n0 + ur) * 2);
  }
  int foA[4], foB[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) foA[i] = p8_tr_frag_off(wr * 4 + i, lane);
#pragma unroll
  for (int j = 0; j < 2; ++j) foB[j] = p8_tr_frag_off(wc * 2 + j, lane);

  f32x4_t acc[2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[a][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bool do_rowsum = d.a_rowsum != nullptr && tile_n == 0;      // uniform
  f32x4_t rs[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) rs[a][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  p8_issue<2>(Ab, offA[0], smem + 0 * UNIT, wave);
  p8_issue<2>(Bb, offB, smem + 2 * UNIT, wave);
  p8_issue<2>(Ab, offA[1], smem + 1 * UNIT, wave);
  {
    const bool has1 = nt > 1;
    p8_issue<2>(has1 ? Ab + stepA : nullptr, offA[0], smem + BUF + 0 * UNIT, wave);
    p8_issue<2>(has1 ? Bb + stepB : nullptr, offB, smem + BUF + 2 * UNIT, wave);
  }
  p8_wait_vmcnt<4>();                  // tile 0 has landed (A.m0, B of tile 1 may still be moving)
  __builtin_amdgcn_s_barrier();
  if (STAGGER && wr == 1) __builtin_amdgcn_s_barrier();

  bf16x8_t fa[4][2], fb[2][2];
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * BUF;
    char* oth = smem + ((t & 1) ^ 1) * BUF;
    const int mine = (wc - 2 * t) & 3;                   // pair (t, ks) belongs to wave column (2 t + ks) % 4: ks == mine (0 / 1 / none)
    // ---- phase 1: rows m0
    p8_read_tr<2>(cur + 2 * UNIT, foB, fb);
    p8_read_tr<4>(cur + 0 * UNIT, foA, fa);
    p8_issue<2>((t + 1 < nt) ? Ab + (int64_t)(t + 1) * stepA : nullptr, offA[1], oth + 1 * UNIT, wave);     // A.m1 of tile t + 1
    p8_wait_vmcnt<6>();                                                                              // A.m1 of tile t has landed
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 2>(fa, fb, acc[0]);
    if (do_rowsum && mine < 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rs[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mine ? fa[i][1] : fa[i][0], ones, rs[0][i], 0, 0, 0);
    }
    P8_PHASE_SYNC_OUT();
    // ---- phase 2: rows m1
    p8_read_tr<4>(cur + 1 * UNIT, foA, fa);
    p8_issue<2>((t + 2 < nt) ? Ab + (int64_t)(t + 2) * stepA : nullptr, offA[0], cur + 0 * UNIT, wave);     // A.m0 of tile t + 2
    p8_issue<2>((t + 2 < nt) ? Bb + (int64_t)(t + 2) * stepB : nullptr, offB, cur + 2 * UNIT, wave);        // B of tile t + 2
    p8_wait_vmcnt<6>();                                                                              // A.m0, B of tile t + 1 have landed
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 2>(fa, fb, acc[1]);
    if (do_rowsum && mine < 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rs[1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mine ? fa[i][1] : fa[i][0], ones, rs[1][i], 0, 0, 0);
    }
    P8_PHASE_SYNC_OUT();
  }
  if (STAGGER && wr == 0) __builtin_amdgcn_s_barrier();
  p8_wait_vmcnt<0>();
  __syncthreads();
  if (do_rowsum) {
    // every column of rs[a][i] holds the partial sums of rows (a, i, lg * 4 + r): column lr == 0 writes them, then one thread
    // per row adds the four wave columns in a fixed order
    float* part = reinterpret_cast<float*>(smem + 2 * BUF - 4096);         // [4 wc][256 rows], above the C tiles of the epilogue
    const int lr = lane & 15, lg = lane >> 4;
    if (lr == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) part[wc * 256 + wr * 128 + a * 64 + i * 16 + lg * 4 + r] = rs[a][i][r];
    }
    __syncthreads();
    if (threadIdx.x < 256) {
      const int m = m0 + (int)threadIdx.x;
      const float v = ((part[threadIdx.x] + part[256 + threadIdx.x]) + part[512 + threadIdx.x]) + part[768 + threadIdx.x];
      d.a_rowsum[m] = (d.a_rowsum_accumulate ? d.a_rowsum[m] : 0.f) + v;
    }
  }
  float* cs = reinterpret_cast<float*>(smem) + wave * (64 * 32);
#pragma unroll 1
  for (int a = 0; a < 2; ++a) {
    if (a == 0) epilogue_stage<64, 32>(acc[0], cs);
    else epilogue_stage<64, 32>(acc[1], cs);
    epilogue_flush<64, 32>(d, 0, 0, m0 + wr * 128 + a * 64, n0 + wc * 32, cs, 1, 0, 0);
  }
}

// the same on a 256 x 256 tile: the four-phase schedule of gemm_8ph_kernel_q<., 2, 4> (wave tile 128 x 64, units A.m0 / A.m1 /
// B.n0 / B.n1 of 128 rows each, 128 KB of LDS).  ~15 % faster per flop than the two-phase tile; taken when its tiles fill the chip
// (grouped launches of several layers' gradients: five AAS-VC decoder problems are 252 tiles).
template <bool STAGGER>
__device__ __forceinline__ void p8_tr_tile_q(const s2svc_gemm_desc& d, int tile_m, int tile_n, char* smem) {
  constexpr int UNIT = 16384, BUF = 4 * UNIT;            // A.m0 | A.m1 | B.n0 | B.n1
  constexpr int OA0 = 0, OA1 = UNIT, OB0 = 2 * UNIT, OB1 = 3 * UNIT;
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const char* Ab = reinterpret_cast<const char*>(d.A.ptr);
  const char* Bb = reinterpret_cast<const char*>(d.B.ptr);
  const int nt = d.K / 64;
  const int64_t stepA = (int64_t)d.A.ld * 128, stepB = (int64_t)d.B.ld * 128;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int half = wave >> 2;

  uint32_t offA[2][2], offB[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    int k, ur;
    p8_tr_src(wave * 2 + e, lane, k, ur);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      offA[h][e] = (uint32_t)(((int64_t)k * d.A.ld + m0 + (ur >> 6) * 128 + h * 64 + (ur & 63)) * 2);
      offB[h][e] = (uint32_t)(((int64_t)k * d.B.ld + n0 + (ur >> 5) * 64 + h * 32 + (ur & 31)) * 2);
    }
  }
  int foA[4], foB[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) foA[i] = p8_tr_frag_off(wr * 4 + i, lane);
#pragma unroll
  for (int j = 0; j < 2; ++j) foB[j] = p8_tr_frag_off(wc * 2 + j, lane);

  f32x4_t acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[a][b][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const bool do_rowsum = d.a_rowsum != nullptr && tile_n == 0;
  f32x4_t rs[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) rs[a][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);

  // prologue: tile 0 complete, tile 1 without B.n0 (issued in phase 1 of tile 0)
  p8_issue<2>(Ab, offA[0], smem + OA0, wave);
  p8_issue<2>(Bb, offB[0], smem + OB0, wave);
  p8_issue<2>(Bb, offB[1], smem + OB1, wave);
  p8_issue<2>(Ab, offA[1], smem + OA1, wave);
  {
    const bool has1 = nt > 1;
    p8_issue<2>(has1 ? Ab + stepA : nullptr, offA[0], smem + BUF + OA0, wave);
    p8_issue<2>(has1 ? Bb + stepB : nullptr, offB[1], smem + BUF + OB1, wave);
    p8_issue<2>(has1 ? Ab + stepA : nullptr, offA[1], smem + BUF + OA1, wave);
  }
  p8_wait_vmcnt<6>();
  __builtin_amdgcn_s_barrier();
  if (STAGGER && half == 1) __builtin_amdgcn_s_barrier();

  bf16x8_t fa[4][2], fb0[2][2], fb1[2][2];
  for (int t = 0; t < nt; ++t) {
    char* cur = smem + (t & 1) * BUF;
    char* oth = smem + ((t & 1) ^ 1) * BUF;
    const char* a2 = (t + 2 < nt) ? Ab + (int64_t)(t + 2) * stepA : nullptr;
    const char* b2 = (t + 2 < nt) ? Bb + (int64_t)(t + 2) * stepB : nullptr;
    const char* b1 = (t + 1 < nt) ? Bb + (int64_t)(t + 1) * stepB : nullptr;
    const int mine = (wc - 2 * t) & 3;
    // ---- phase 1: quadrant (m0, n0)
    p8_read_tr<2>(cur + OB0, foB, fb0);
    p8_read_tr<4>(cur + OA0, foA, fa);
    p8_issue<2>(b1, offB[0], oth + OB0, wave);                                // B.n0 of tile t + 1
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 2>(fa, fb0, acc[0][0]);
    P8_PHASE_SYNC_OUT();
    // ---- phase 2: quadrant (m0, n1)
    p8_read_tr<2>(cur + OB1, foB, fb1);
    p8_issue<2>(a2, offA[0], cur + OA0, wave);                                // A.m0 of tile t + 2
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 2>(fa, fb1, acc[0][1]);
    if (do_rowsum && mine < 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rs[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mine ? fa[i][1] : fa[i][0], ones, rs[0][i], 0, 0, 0);
    }
    P8_PHASE_SYNC_OUT();
    // ---- phase 3: quadrant (m1, n1)
    p8_read_tr<4>(cur + OA1, foA, fa);
    p8_issue<2>(b2, offB[1], cur + OB1, wave);                                // B.n1 of tile t + 2
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 2>(fa, fb1, acc[1][1]);
    P8_PHASE_SYNC_OUT();
    // ---- phase 4: quadrant (m1, n0)
    p8_issue<2>(a2, offA[1], cur + OA1, wave);                                // A.m1 of tile t + 2
    p8_wait_vmcnt<6>();                                                       // everything of tile t + 1 has landed
    P8_PHASE_SYNC_IN();
    p8_mfma<4, 2>(fa, fb0, acc[1][0]);
    if (do_rowsum && mine < 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) rs[1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mine ? fa[i][1] : fa[i][0], ones, rs[1][i], 0, 0, 0);
    }
    P8_PHASE_SYNC_OUT();
  }
  if (STAGGER && half == 0) __builtin_amdgcn_s_barrier();
  p8_wait_vmcnt<0>();
  __syncthreads();
  if (do_rowsum) {
    float* part = reinterpret_cast<float*>(smem + 2 * BUF - 4096);
    const int lr = lane & 15, lg = lane >> 4;
    if (lr == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) part[wc * 256 + wr * 128 + a * 64 + i * 16 + lg * 4 + r] = rs[a][i][r];
    }
    __syncthreads();
    if (threadIdx.x < 256) {
      const int m = m0 + (int)threadIdx.x;
      const float v = ((part[threadIdx.x] + part[256 + threadIdx.x]) + part[512 + threadIdx.x]) + part[768 + threadIdx.x];
      d.a_rowsum[m] = (d.a_rowsum_accumulate ? d.a_rowsum[m] : 0.f) + v;
    }
  }
  float* cs = reinterpret_cast<float*>(smem) + wave * (64 * 32);
#pragma unroll 1
  for (int q = 0; q < 4; ++q) {
    switch (q) {
      case 0: epilogue_stage<64, 32>(acc[0][0], cs); break;
      case 1: epilogue_stage<64, 32>(acc[0][1], cs); break;
      case 2: epilogue_stage<64, 32>(acc[1][0], cs); break;
      default: epilogue_stage<64, 32>(acc[1][1], cs); break;
    }
    epilogue_flush<64, 32>(d, 0, 0, m0 + wr * 128 + (q >> 1) * 64, n0 + wc * 64 + (q & 1) * 32, cs, 1, 0, 0);
  }
}

template <bool STAGGER>
__global__ __launch_bounds__(512) void gemm_8ph_tr_kernel(const s2svc_gemm_desc d) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * 3 * 16384];
  int tile_m, tile_n;
  p8_tile_of_block<256, 128>(tile_m, tile_n);
  p8_tr_tile<STAGGER>(d, tile_m, tile_n, smem);
}

// Grouped launch (see gemm_grouped_kernel, gemm_glds.hip): the weight gradients of a few consecutive layers as ONE grid of
// 256 x 128 tiles, descriptors by value in the kernel arguments.
#define P8_GROUP_MAX 10
struct p8_group_args {
  s2svc_gemm_desc d[P8_GROUP_MAX];
  int32_t tile_start[P8_GROUP_MAX + 1];
  int32_t n;
};
static_assert(sizeof(p8_group_args) <= 4096, "kernel arguments are limited to 4 KB");

template <bool STAGGER, int BN>
__global__ __launch_bounds__(512) void gemm_8ph_tr_grouped_kernel(const p8_group_args g) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * (BN == 256 ? 4 : 3) * 16384];
  int p = 0;
#pragma unroll
  for (int i = 1; i < P8_GROUP_MAX; ++i) p += (i < g.n && g.tile_start[i] <= (int)blockIdx.x) ? 1 : 0;
  const s2svc_gemm_desc& d = g.d[p];
  const int t = (int)blockIdx.x - g.tile_start[p];
  const int tiles_n = d.N / BN;
  const int tile_m = t / tiles_n;
  if (BN == 256) p8_tr_tile_q<STAGGER>(d, tile_m, t - tile_m * tiles_n, smem);
  else p8_tr_tile<STAGGER>(d, tile_m, t - tile_m * tiles_n, smem);
}

// ---------------------------------------------------------------------------------------------------------
// RAGGED weight-gradient tiles ("W8", round 4): the same 256 x 128 two-phase schedule for ANY dense row-contiguous weight
// gradient C[M, N] (+)= A^T . B (M, N multiples of 8; any K) -- VTN's 384 / 1152 / 1536 / 4608 / 7296-feature layers with
// reductions of 2016 / 2048 rows, which used to run on gemm_grouped_kernel<64, 64> (4 waves, one barrier + a drained DMA per
// K tile, 32 flop per staged byte: MFMA pipe 10 % busy, waves parked 69 % of their cycles, profiles/r03_step_mfma_busy.txt).
// What it adds to p8_tr_tile:
//   * raggedness is a property of the DMA SOURCE only: a lane whose 8 rows lie past M / N, or whose k row lies past K (or past
//     the end of its K chunk), reads the 16-byte zero block instead -- the LDS image, fragments, phases and counted waits are
//     those of the exact kernel; rows past the matrix are never stored.  A wave whose 64-row half (or 32-column slice) is
//     entirely outside the matrix skips its fragment reads and MFMAs (it still issues its DMA share and meets the barriers),
//     so the half-empty second row tile of a 384-row output leaves the SIMD's matrix pipe to the partner wave;
//   * WORK UNITS (problem, K chunk, tile): a problem's reduction is cut into chunks of kt_chunk K tiles -- a function of K
//     ONLY, so a staged backward pass (other groups) sums in the order of the uncut one.  One chunk: the unit accumulates
//     straight into C (the flat-gradient slot).  Several: every unit stores its fp32 partial tile (and partial bias row sums)
//     to a workspace slice and w8_reduce_kernel adds the slices in chunk order -- deterministic, no atomics;
//   * compact problem records (96 bytes instead of the 384-byte descriptor): 40 problems per launch.
// Unit order inside a problem: chunk-major, then row tile, column tile innermost -- neighbours share the A panel of their
// (chunk, row tile); the launch deals unit ids to XCDs in contiguous runs (as p8_tile_of_block does for one problem).
// ---------------------------------------------------------------------------------------------------------
#define W8_MAX 40
struct w8_prob {
  const void* A;                 // [K][lda]  (dY: row-contiguous, element (k, m) at A[k * lda + m])
  const void* B;                 // [K][ldb]  (X)
  float* C;                      // [M][ldc] fp32
  float* rowsum;                 // [M] bias gradient (sum over k of A[k][m]) or null
  float* ws;                     // nchunks > 1: [nchunks][M][N] partial tiles
  float* rs_ws;                  // nchunks > 1 and rowsum: [nchunks][M]
  int32_t lda, ldb, ldc, M, N, K;
  int32_t tiles_m, tiles_n, nchunks, kt_chunk;
  int32_t flags;                 // bit 0: accumulate into C, bit 1: accumulate into rowsum, bit 2: B is the implicit im2col operand `cv`
  int32_t reserved_;
};
// B as the implicit im2col matrix of a Conv2d 3x3 stride 2 (S2SVC_OP_CONV2D_S2, row-contiguous): reduction row k = output pixel
// (b, t2, f2), column n = tap * C + c  ->  x[b, 2 t2 + tap / 3, 2 f2 + tap % 3, c] (ld = q.ldb elements per input pixel).  C % 128 == 0, so
// the 128 columns of a tile lie inside ONE tap: a loader lane's four source rows are four pixels, decomposed per K tile by two
// multiply-high divisions each.  One geometry per launch (a Conv2d weight gradient is launched on its own).
// kind 1: the implicit im2col matrix of a Conv1d (stride 1, 'same' padding, S2SVC_OP_CONV1D): reduction row k = frame (b, t), column
// n = tap * C + c  ->  x[(k + tap - pad) * ldb + c] if 0 <= t + tap - pad < T, else 0  (T1 = T, F1 = pad).
struct w8_conv {
  int32_t T1, F1, T2, F2, C, kind;
};
struct w8_args {
  w8_prob p[W8_MAX];
  int32_t unit_start[W8_MAX + 1];
  int32_t n, total;
  w8_conv cv;
};
static_assert(sizeof(w8_prob) == 96, "w8_prob layout");
static_assert(sizeof(w8_args) <= 4096, "kernel arguments are limited to 4 KB");

// fp32 C tile of one wave (64 x 32, staged in `cs` by epilogue_stage): accumulate into C, or store the chunk's partial
__device__ __forceinline__ void w8_flush(const w8_prob& q, int chunk, int m_base, int n_base, const float* cs) {
  const int lane = threadIdx.x & 63;
  const bool partial = q.nchunks > 1;
#pragma unroll 1
  for (int p = 0; p < 4; ++p) {
    const int row = p * 16 + (lane >> 2), col = (lane & 3) * 8;
    const int m = m_base + row, n = n_base + col;
    if (m >= q.M || n >= q.N) continue;
    const float* src = cs + row * 32 + (col ^ (((row >> 2) & 1) << 4));
    float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
    float* c = partial ? q.ws + ((int64_t)chunk * q.M + m) * q.N + n : q.C + (int64_t)m * q.ldc + n;
    if (!partial && (q.flags & 1)) {
      const float4 c0 = *reinterpret_cast<const float4*>(c), c1 = *reinterpret_cast<const float4*>(c + 4);
      lo.x += c0.x; lo.y += c0.y; lo.z += c0.z; lo.w += c0.w; hi.x += c1.x; hi.y += c1.y; hi.z += c1.z; hi.w += c1.w;
    }
    *reinterpret_cast<float4*>(c) = lo;
    *reinterpret_cast<float4*>(c + 4) = hi;
  }
}

// ---------------------------------------------------------------------------------------------------------
// LOADER-SPECIALISED W8 tile (round 4, "LS"): 12 waves -- 8 CONSUMER waves (the 2 x 4 wave grid of the 256 x 128 tile: fragment
// reads + MFMAs, nothing else) and 4 LOADER waves (one per SIMD: all LDS-DMA instructions, the address masks, the counted vmcnt
// waits) over a ring of THREE K-tile stages (144 KB) with ONE s_barrier per K tile.
// Why: timing builds of the 8-wave kernel with one of {MFMA, fragment reads, DMA} removed (profiles/AB_LOG.md, round 4) put the three at
// 0.44 / 0.27 / 0.35 us per K tile above a 0.18 us loop floor, and all three together at 1.04 us -- nearly their sum: a wave that issues
// its share of the DMAs (6 instructions of 60-185 cycles each: MI355X_MICROARCH.md, "LDS-DMA piece issue cost") and waits for
// its reads is not issuing MFMAs, and the two barriers per phase keep all eight waves in that lock step.  With the DMA issue on
// waves of their own the consumers' loop is [20 fragment reads, 32 MFMAs, barrier]; the loaders run a full K tile ahead.
//   barrier B(t) (t = 0 .. nt): loaders arrive when THEIR share of tile t has landed (counted vmcnt: the 12 instructions of the tile
//   issued last may still be in flight) -- so tile t is complete for every reader behind B(t); consumers arrive at B(t + 1) with
//   every read of tile t retired (lgkmcnt(0)) -- so behind B(t + 1) the loaders may overwrite stage t % 3 with tile t + 3.
// Same LDS image, fragment offsets, masks, chunking, row sums and epilogue as the 8-wave tile it replaced (removed at the end of round 4
// together with its fragment-prefetch variant: bit-identical results, 1.05 - 1.07 us per K tile).
// Measured (tools/_scratch sweep, one problem of 72-96 tiles): 0.83 us per K tile against 1.07 (8 waves, fragment prefetch) and 1.05
// (p8_tr_tile); a variant with four sub-phases and the next sub-phase's fragments requested ahead of the MFMAs needs 12 registers
// more than the 168 that three waves per SIMD allow (spills: slower with the row sums, +3 % without) -- not kept.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void w8ls_tile(const w8_prob& q, const w8_conv& cv, int tile_m, int tile_n, int chunk, char* smem) {
  constexpr int UNIT = 16384, BUF = 3 * UNIT, NS = 3;    // A.m0 | A.m1 | B per stage
  const int m0 = tile_m * 256, n0 = tile_n * 128;
  const int ktiles = (q.K + 63) >> 6;
  const int kt0 = chunk * q.kt_chunk;
  const int nt = (ktiles - kt0 < q.kt_chunk) ? ktiles - kt0 : q.kt_chunk;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_rowsum = q.rowsum != nullptr && tile_n == 0;      // uniform
  if (wave >= 8) {
    // ================= loader =================
    const int lw = wave - 8;
    const int64_t stepA = (int64_t)q.lda * 128, stepB = (int64_t)q.ldb * 128;
    const char* Ab = reinterpret_cast<const char*>(q.A) + (int64_t)kt0 * stepA;
    const char* Bb = reinterpret_cast<const char*>(q.B) + (int64_t)kt0 * stepB;
    const int krem = q.K - kt0 * 64;
    int kin[4];
    uint32_t offA[2][4], offB[4];
    bool okA[2][4], okB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                        // pieces lw * 4 + i of every unit
      int ur;
      p8_tr_src(lw * 4 + i, lane, kin[i], ur);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = m0 + (ur >> 6) * 128 + h * 64 + (ur & 63);
        okA[h][i] = row < q.M;
        offA[h][i] = (uint32_t)(((int64_t)kin[i] * q.lda + row) * 2);
      }
      okB[i] = n0 + ur < q.N;
      offB[i] = (uint32_t)(((int64_t)kin[i] * q.ldb + n0 + ur) * 2);
    }
    // implicit im2col B (flags bit 2): the tile's tap and first channel; source pixel of reduction row p decomposed per K tile
    const bool convB = (q.flags & 4) != 0;               // uniform
    const bool conv1 = convB && cv.kind == 1;            // uniform
    const fastdiv_t dv_pb = fastdiv_make(!convB ? 1 : conv1 ? cv.T1 : cv.T2 * cv.F2), dv_f = fastdiv_make(convB && !conv1 ? cv.F2 : 1);
    uint32_t colB[4] = {0u, 0u, 0u, 0u};
    int tshift = 0;                                      // conv1d: tap - pad
    if (convB) {
      const int tap = n0 / cv.C, c0 = n0 - tap * cv.C, kh = tap / 3, kw = tap - kh * 3;
      tshift = tap - cv.F1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int kk, ur;
        p8_tr_src(lw * 4 + i, lane, kk, ur);
        colB[i] = conv1 ? (uint32_t)((c0 + ur) * 2) : (uint32_t)((((int64_t)kh * cv.F1 + kw) * q.ldb + c0 + ur) * 2);
      }
    }
    const char* Bx = reinterpret_cast<const char*>(q.B);
    const char* z = reinterpret_cast<const char*>(&g_zero16_8ph);
    // the 12 DMA instructions of this loader for chunk-relative K tile T into stage T % 3
#define W8LS_ISSUE(T)                                                                                                         \
    {                                                                                                                         \
      const int tt_ = (T);                                                                                                    \
      const int klim_ = (tt_ < nt) ? krem - tt_ * 64 : 0;                                                                     \
      char* st_ = smem + (tt_ % NS) * BUF;                                                                                    \
      const char* ab_ = Ab + (int64_t)tt_ * stepA;                                                                            \
      const char* bb_ = Bb + (int64_t)tt_ * stepB;                                                                            \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                         \
        const bool kok_ = kin[i] < klim_;                                                                                     \
        const char* s0_ = (okA[0][i] && kok_) ? ab_ + offA[0][i] : z;                                                         \
        const char* s1_ = (okA[1][i] && kok_) ? ab_ + offA[1][i] : z;                                                         \
        const char* s2_ = (okB[i] && kok_) ? bb_ + offB[i] : z;                                                               \
        if (convB && okB[i] && kok_) {                                                                                       \
          const int p_ = (kt0 + tt_) * 64 + kin[i];                                                                           \
          const int b_ = fastdiv(p_, dv_pb), r_ = p_ - b_ * (int)dv_pb.d;                                                     \
          if (conv1) {                                                                                                        \
            const int ts_ = r_ + tshift;                                                                                      \
            s2_ = (ts_ >= 0 && ts_ < cv.T1) ? Bx + (int64_t)(p_ + tshift) * q.ldb * 2 + colB[i] : z;                          \
          } else {                                                                                                            \
            const int t2_ = fastdiv(r_, dv_f), f2_ = r_ - t2_ * (int)dv_f.d;                                                  \
            s2_ = Bx + ((int64_t)(b_ * cv.T1 + 2 * t2_) * cv.F1 + 2 * f2_) * q.ldb * 2 + colB[i];                             \
          }                                                                                                                   \
        }                                                                                                                     \
        __builtin_amdgcn_global_load_lds((gbl_void*)s0_, (lds_void*)(st_ + 0 * UNIT + (lw * 4 + i) * 1024), 16, 0, 0);       \
        __builtin_amdgcn_global_load_lds((gbl_void*)s1_, (lds_void*)(st_ + 1 * UNIT + (lw * 4 + i) * 1024), 16, 0, 0);       \
        __builtin_amdgcn_global_load_lds((gbl_void*)s2_, (lds_void*)(st_ + 2 * UNIT + (lw * 4 + i) * 1024), 16, 0, 0);       \
      }                                                                                                                       \
    }
    W8LS_ISSUE(0);
    W8LS_ISSUE(1);
    p8_wait_vmcnt<12>();               // this loader's share of tile 0 has landed
    __builtin_amdgcn_s_barrier();      // B(0)
#pragma unroll 1
    for (int t = 0; t < nt; ++t) {
      W8LS_ISSUE(t + 2);               // stage (t + 2) % 3 = (t - 1) % 3: every consumer retired its reads of tile t - 1 before B(t)
      p8_wait_vmcnt<12>();             // tile t + 1 has landed
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();    // B(t + 1)
      __builtin_amdgcn_sched_barrier(0);
    }
#undef W8LS_ISSUE
    p8_wait_vmcnt<0>();                // (the zero-block DMAs issued past the end)
    __builtin_amdgcn_s_barrier();      // the consumers' barrier in front of the epilogue: the stages become C tiles
    if (do_rowsum) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();      // end of the unit (the consumers' epilogue has left the stages)
    return;
  }
  // ================= consumer =================
  const int wr = wave >> 2, wc = wave & 3;
  int foA[4], foB[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) foA[i] = p8_tr_frag_off(wr * 4 + i, lane);
#pragma unroll
  for (int j = 0; j < 2; ++j) foB[j] = p8_tr_frag_off(wc * 2 + j, lane);
  const bool liveN = (n0 + wc * 32 < q.N) || do_rowsum;
  const bool live0 = liveN && (m0 + wr * 128 < q.M), live1 = liveN && (m0 + wr * 128 + 64 < q.M);
  f32x4_t acc[2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[a][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  f32x4_t rs[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) rs[a][i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 ones_s = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);
  bf16x8_t fa[4][2], fb[2][2];
  __builtin_amdgcn_s_barrier();        // B(0): tile 0 is complete
#pragma unroll 1
  for (int t = 0; t < nt; ++t) {
    const char* st = smem + (t % NS) * BUF;
    const int mine = (wc - 2 * t) & 3;                   // row-sum pair (t, ks) belongs to wave column (2 t + ks) % 4
    if (live0) {
      p8_read_tr<2>(st + 2 * UNIT, foB, fb);
      p8_read_tr<4>(st + 0 * UNIT, foA, fa);
      p8_wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
      p8_mfma<4, 2>(fa, fb, acc[0]);
      if (do_rowsum && mine < 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rs[0][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mine ? fa[i][1] : fa[i][0], ones, rs[0][i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (live1) {
      p8_read_tr<4>(st + 1 * UNIT, foA, fa);
      p8_wait_lgkm0();
      __builtin_amdgcn_sched_barrier(0);
      p8_mfma<4, 2>(fa, fb, acc[1]);
      if (do_rowsum && mine < 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) rs[1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mine ? fa[i][1] : fa[i][0], ones, rs[1][i], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();      // B(t + 1): every read of tile t is retired (lgkmcnt(0) above); tile t + 1 is complete
    __builtin_amdgcn_sched_barrier(0);
  }
  __builtin_amdgcn_s_barrier();        // (with the loaders' vmcnt(0)) the operand stages are dead: reuse them as fp32 C tiles
  if (do_rowsum) {
    float* part = reinterpret_cast<float*>(smem + 2 * BUF - 4096);         // [4 wc][256 rows], above the C tiles of the epilogue
    const int lr = lane & 15, lg = lane >> 4;
    if (lr == 0) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) part[wc * 256 + wr * 128 + a * 64 + i * 16 + lg * 4 + r] = rs[a][i][r];
    }
    p8_wait_lgkm0();                                     // the partial sums are in LDS
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x < 256) {
      const int m = m0 + (int)threadIdx.x;
      if (m < q.M) {
        const float v = ((part[threadIdx.x] + part[256 + threadIdx.x]) + part[512 + threadIdx.x]) + part[768 + threadIdx.x];
        if (q.nchunks > 1) q.rs_ws[(int64_t)chunk * q.M + m] = v;
        else q.rowsum[m] = ((q.flags & 2) ? q.rowsum[m] : 0.f) + v;
      }
    }
  }
  float* cs = reinterpret_cast<float*>(smem) + wave * (64 * 32);
#pragma unroll 1
  for (int a = 0; a < 2; ++a) {
    if (a == 0) epilogue_stage<64, 32>(acc[0], cs);
    else epilogue_stage<64, 32>(acc[1], cs);
    w8_flush(q, chunk, m0 + wr * 128 + a * 64, n0 + wc * 32, cs);
  }
  p8_wait_lgkm0();
  __builtin_amdgcn_s_barrier();        // end of the unit: a further unit's DMAs may overwrite the C tiles
}

// one workgroup per unit, or -- gridDim.x < total: a CAPPED grid (ops.kernels.set_wgrad_cap: forked gradient batches) -- gridDim.x workgroups that
// walk the units and leave the other CUs to the stream beside them
__global__ __launch_bounds__(768) void gemm_w8ls_kernel(const w8_args g) {
  __shared__ __attribute__((aligned(1024))) char smem[3 * 3 * 16384];
#pragma unroll 1
  for (int id = (int)blockIdx.x; id < g.total; id += (int)gridDim.x) {
    int u = id;
    if (gridDim.x == (unsigned)g.total) {                // XCD x gets the x-th contiguous run of units
      const int q8 = g.total >> 3, r8 = g.total & 7;
      const int xcd = id & 7, j = id >> 3;
      u = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
    }
    int p = 0;
#pragma unroll 1
    for (int i = 1; i < g.n; ++i) p += (g.unit_start[i] <= u) ? 1 : 0;
    const w8_prob& q = g.p[p];
    int r = u - g.unit_start[p];
    const int per_chunk = q.tiles_m * q.tiles_n;
    const int chunk = r / per_chunk;
    r -= chunk * per_chunk;
    const int tile_m = r / q.tiles_n;
    w8ls_tile(q, g.cv, tile_m, r - tile_m * q.tiles_n, chunk, smem);
  }
}

// C (+)= sum over chunks of the partial tiles, in chunk order; the same for the bias row sums.  blockIdx.y = problem.
__global__ __launch_bounds__(256) void w8_reduce_kernel(const w8_args g) {
  const w8_prob& q = g.p[blockIdx.y];
  if (q.nchunks <= 1) return;
  const int n4 = q.N >> 2;
  const int64_t total4 = (int64_t)q.M * n4, plane = (int64_t)q.M * q.N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
    const int m = (int)(i / n4), n = (int)(i - (int64_t)m * n4) * 4;
    const float* w = q.ws + (int64_t)m * q.N + n;
    float4 s = *reinterpret_cast<const float4*>(w);
#pragma unroll 1
    for (int c = 1; c < q.nchunks; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(w + c * plane);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* cp = q.C + (int64_t)m * q.ldc + n;
    if (q.flags & 1) {
      const float4 c0 = *reinterpret_cast<const float4*>(cp);
      s.x += c0.x; s.y += c0.y; s.z += c0.z; s.w += c0.w;
    }
    *reinterpret_cast<float4*>(cp) = s;
  }
  if (q.rowsum && blockIdx.x == 0) {
    for (int m = (int)threadIdx.x; m < q.M; m += 256) {
      float s = q.rs_ws[m];
#pragma unroll 1
      for (int c = 1; c < q.nchunks; ++c) s += q.rs_ws[(int64_t)c * q.M + m];
      q.rowsum[m] = ((q.flags & 2) ? q.rowsum[m] : 0.f) + s;
    }
  }
}

template <bool STAGGER>
__global__ __launch_bounds__(512) void gemm_8ph_tr_kernel_q(const s2svc_gemm_desc d) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * 4 * 16384];
  int tile_m, tile_n;
  p8_tile_of_block<256, 256>(tile_m, tile_n);
  p8_tr_tile_q<STAGGER>(d, tile_m, tile_n, smem);
}

int p8_force_bn();
// 256 x 256 or 256 x 128 tiles for `t256` / `t128` tiles in total: rounds of workgroups on the 256 CUs in units of one
// 256 x 128 tile, the big tile ~15 % faster per flop (the cost model of s2svc_gemm_try_8ph)
bool p8_tr_take_q(bool all_n256, int64_t t256, int64_t t128) {
  const int force = p8_force_bn();
  if (force == 1) return all_n256;
  if (force == 3) return false;
  if (!all_n256) return false;
  const double cost256 = 2.0 * 0.85 * (double)((t256 + 255) / 256), cost128 = (double)((t128 + 255) / 256);
  return t256 >= 160 && cost256 <= cost128;
}

bool p8_tr_ok(const s2svc_gemm_desc& d) {      // exact tiles of dense row-contiguous operands, fp32 or bf16 C, no batch / split-K
  if (d.dtype != S2S_BF16 || d.nb0 * d.nb1 != 1 || d.splitk > 1) return false;
  if (d.A.layout != S2SVC_LAYOUT_RC || d.B.layout != S2SVC_LAYOUT_RC || d.A.mode != S2SVC_OP_DENSE || d.B.mode != S2SVC_OP_DENSE) return false;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.M % 256 || d.N % 128 || d.K % 64) return false;
  if (((uintptr_t)d.A.ptr) % 16 || ((uintptr_t)d.B.ptr) % 16 || d.A.ld % 8 || d.B.ld % 8 || d.A.ld < d.M || d.B.ld < d.N) return false;
  if ((int64_t)64 * d.A.ld * 2 + (int64_t)d.M * 2 >= (1ll << 32) || (int64_t)64 * d.B.ld * 2 + (int64_t)d.N * 2 >= (1ll << 32)) return false;
  if (d.emask || d.drop_p > 0.f || d.c_map || d.c_pre) return false;
  return true;
}

constexpr int p8_tr_mode() { return 1; }     // weight gradients on the 8-wave kernels (the 4-wave path was the A/B alternative through round 5)

int g_p8_mode = -1;
int p8_mode() {       // s2svc_gemm_set_8ph: 0 = off, 1 = on (default), 2 = on without the half-phase skew of the wave halves
  if (g_p8_mode < 0) g_p8_mode = 1;
  return g_p8_mode;
}

constexpr int p8_min_tiles() { return 128; }  // take the kernel from this many 256-row tiles on

int g_p8_n96 = -1;
int p8_n96_mode() {     // s2svc_gemm_set_8ph: 0 = off, 1 = by policy (default), 2 = wherever N is a multiple of 96 PH, 4 / 5 = PH = 2 / 3 only (tests / benchmarks)
  if (g_p8_n96 < 0) g_p8_n96 = 1;
  return g_p8_n96;
}
// phases (tile width / 96) of the one-round geometry for this problem, 0 = keep the geometry chosen so far (`tiles` workgroups)
int p8_n96_phases(const s2svc_gemm_desc& d, int geo, int64_t tiles) {
  const int md = p8_n96_mode();
  if (md == 0 || d.K < 128 || d.N % 96) return 0;
  const int64_t tm = (d.M + 255) / 256;
  for (int ph = 3; ph >= 2; --ph) {
    if (d.N % (96 * ph)) continue;
    const int64_t t = tm * (d.N / (96 * ph));
    if (md >= 2) { if (md == 2 || md == 2 + ph) return ph; continue; }
    // by policy: one full round (>= 7/8 of the CUs) where the other geometry is short of one or runs over into the next
    if (t <= 256 && t >= 224 && (geo == 0 || tiles < 224 || tiles > 256)) return ph;
  }
  return 0;
}

int g_p8_geo = -1;
int p8_force_bn() {   // s2svc_gemm_set_8ph (1|2|3): force the tile geometry 256x256 / 512x128 / 256x128
  if (g_p8_geo < 0) g_p8_geo = 0;
  return g_p8_geo;
}

bool p8_operand_ok(const s2svc_operand& o, int rows, int K) {
  if (o.layout != S2SVC_LAYOUT_KC || ((uintptr_t)o.ptr) % 16 || o.ld % 8 || o.bs0 % 8 || o.bs1 % 8) return false;
  int64_t elems;
  if (o.mode == S2SVC_OP_DENSE) {
    elems = (int64_t)rows * o.ld;
  } else if (o.mode == S2SVC_OP_CONV2D_S2) {
    if (o.C % 64 || o.C < 64 || K != 9 * o.C) return false;
    elems = (int64_t)(rows / (o.T2 * o.F2) + 1) * o.T1 * o.F1 * o.ld;
  } else if (o.mode == S2SVC_OP_CONV1D) {           // stride 1, 'same' padding, odd kernel width 2 pad + 1 (an A-operand mode)
    if (o.pad < 0 || o.pad > 7 || o.C % 64 || o.C < 64 || K != (2 * o.pad + 1) * o.C || o.T <= 0 || rows % o.T || o.ld < o.C) return false;
    elems = (int64_t)(rows + o.pad) * o.ld;
  } else if (o.mode == S2SVC_OP_TCONV2D_S2) {       // one parity class of the transposed convolution (an A-operand mode)
    const int ntap = (2 - (o.pad >> 1)) * (2 - (o.pad & 1));
    if (o.pad < 0 || o.pad > 3 || o.C % 64 || o.C < 64 || K != ntap * o.C || o.T1 <= 0 || o.F1 <= 0) return false;
    elems = (int64_t)(rows / (o.T1 * o.F1) + 1) * o.T2 * o.F2 * o.ld;
  } else {
    return false;
  }
  return elems * 2 < (1ll << 32);            // 32-bit per-lane byte offsets
}

}  // namespace

// A/B switch for tests and benchmarks: returns the previous mode (see p8_mode)
extern "C" int s2svc_gemm_set_8ph(int mode) {
  const int prev = p8_mode() | (p8_force_bn() << 4) | ((p8_n96_mode() + 1) << 8);
  if (mode >= 0) {
    const int n96 = (mode >> 8) & 15;            // 0: leave; 1 + the n96 mode otherwise (p8_n96_mode)
    mode &= 255;
    if ((mode & 15) <= 2 && (mode >> 4) <= 3) {
      g_p8_mode = mode & 15;
      g_p8_geo = mode >> 4;
    }
    if (n96 >= 1 && n96 <= 6) g_p8_n96 = n96 - 1;
  }
  return prev;
}

// returns 1 if launched here, 0 if the problem is not eligible (the caller falls through to gemm_glds.hip)
extern "C" int s2svc_gemm_try_8ph(const s2svc_gemm_desc* desc, void* stream) {
  const s2svc_gemm_desc& d = *desc;
  const int mode = p8_mode();
  if (mode != 0 && p8_tr_mode() && d.tile_hint != 64 && p8_tr_ok(d) && (int64_t)(d.M / 256) * (d.N / 128) >= p8_min_tiles()) {
    const int64_t t128 = (int64_t)(d.M / 256) * (d.N / 128);
    if (p8_tr_take_q(d.N % 256 == 0, t128 / 2, t128)) {
      dim3 grid((unsigned)(d.N / 256), (unsigned)(d.M / 256), 1);
      if (mode == 2) hipLaunchKernelGGL((gemm_8ph_tr_kernel_q<false>), grid, dim3(512), 0, (hipStream_t)stream, d);
      else hipLaunchKernelGGL((gemm_8ph_tr_kernel_q<true>), grid, dim3(512), 0, (hipStream_t)stream, d);
    } else {
      dim3 grid((unsigned)(d.N / 128), (unsigned)(d.M / 256), 1);
      if (mode == 2) hipLaunchKernelGGL((gemm_8ph_tr_kernel<false>), grid, dim3(512), 0, (hipStream_t)stream, d);
      else hipLaunchKernelGGL((gemm_8ph_tr_kernel<true>), grid, dim3(512), 0, (hipStream_t)stream, d);
    }
    S2S_CHECK_LAUNCH("gemm_8ph_tr_kernel");
    return 1;
  }
  if (mode == 0 || d.dtype != S2S_BF16 || d.splitk > 1 || d.a_rowsum || d.tile_hint == 64) return 0;
  if (d.K < 128 || d.K % 64 || d.M < 256 || d.N < 64) return 0;
  if (d.B.mode != S2SVC_OP_DENSE || !p8_operand_ok(d.A, d.M, d.K) || !p8_operand_ok(d.B, d.N, d.K)) return 0;
  if (d.A.mode == S2SVC_OP_TCONV2D_S2 && d.nb0 * d.nb1 != 1) return 0;
  const int64_t nb = (int64_t)d.nb0 * d.nb1;
  const int64_t t256 = (int64_t)((d.M + 255) / 256) * ((d.N + 255) / 256) * nb;       // 256 x 256 tiles
  const int64_t t512 = (int64_t)((d.M + 511) / 512) * ((d.N + 127) / 128) * nb;       // 512 x 128 tiles
  const int64_t t128 = (int64_t)((d.M + 255) / 256) * ((d.N + 127) / 128) * nb;       // 256 x 128 tiles (2-phase kernel)
  // geometry: square tiles when the columns fill them and the grid fills the chip; tall tiles for outputs with few
  // columns (fewer wasted columns, one round of workgroups); the 2-phase 256 x 128 kernel for what is left
  int geo = 0;                                   // 1: 256 x 256, 2: 512 x 128, 3: 256 x 128
  const int waste256 = (int)(((d.N + 255) / 256) * 256 - d.N), waste128 = (int)(((d.N + 127) / 128) * 128 - d.N);
  // rounds of workgroups on the 256 CUs, in units of one 256 x 128 tile: the big tiles run ~15 % faster per flop
  // (gemm8_bench: 4096^3 1285 vs 1099 TFLOP/s) but a second, nearly empty round of them costs two units
  // (4096 x 4608 x 1536, the packed Q|K|V projection of the AAS-VC decoder: 288 tiles of 256 x 256 = 2 rounds = 3.4 units,
  // 576 tiles of 256 x 128 = 3 units)
  const double cost256 = 2.0 * 0.85 * (double)((t256 + 255) / 256), cost128 = (double)((t128 + 255) / 256);
  if (waste256 <= waste128 && t256 >= 192 && cost256 <= cost128) geo = 1;
  else if (d.M >= 2048 && t512 >= 160 && t512 <= 256) geo = 2;
  else if (t128 >= p8_min_tiles()) geo = 3;
  if (p8_force_bn() >= 1 && p8_force_bn() <= 3) geo = p8_force_bn();
  hipStream_t st = (hipStream_t)stream;
  // 256 x 96 PH tiles where they fit the chip in ONE round and the geometry above does not (gemm_8ph_kernel_n96)
  if (d.A.mode == S2SVC_OP_DENSE && nb == 1 && mode == 1 && (p8_force_bn() == 0 || p8_n96_mode() >= 2) && epilogue_common_ok(d)) {
    const int ph = p8_n96_phases(d, geo, geo == 1 ? t256 : geo == 2 ? t512 : t128);
    if (ph) {
      dim3 grid((unsigned)((d.N + 96 * ph - 1) / (96 * ph)), (unsigned)((d.M + 255) / 256), 1);
      if (ph == 3) hipLaunchKernelGGL((gemm_8ph_kernel_n96<3, true>), grid, dim3(512), 0, st, d);
      else hipLaunchKernelGGL((gemm_8ph_kernel_n96<2, true>), grid, dim3(512), 0, st, d);
      S2S_CHECK_LAUNCH("gemm_8ph_kernel_n96");
      return 1;
    }
  }
  if (geo == 0) return 0;
  if (d.A.mode == S2SVC_OP_CONV1D) {
    // Conv1d as an implicit GEMM (the aligner's 1536 -> 1536 k3 layer over 4096 frames: 58 GF forward and data gradient, 0.24-0.28 of the
    // peak on the 4-wave kernel): the common epilogue only, one problem
    static const bool c1d_on = true;
    if (!c1d_on || mode != 1 || nb != 1 || !epilogue_common_ok(d)) return 0;
    const int bm1 = geo == 2 ? 512 : 256, bn1 = geo == 1 ? 256 : 128;
    dim3 grid1((unsigned)((d.N + bn1 - 1) / bn1), (unsigned)((d.M + bm1 - 1) / bm1), 1);
    if (geo == 1) hipLaunchKernelGGL((gemm_8ph_kernel_q<P8_CONV1D, 2, 4, true, true>), grid1, dim3(512), 0, st, d);
    else if (geo == 2) hipLaunchKernelGGL((gemm_8ph_kernel_q<P8_CONV1D, 4, 2, true, true>), grid1, dim3(512), 0, st, d);
    else hipLaunchKernelGGL((gemm_8ph_kernel_128<P8_CONV1D, true, 1>), grid1, dim3(512), 0, st, d);
    S2S_CHECK_LAUNCH("gemm_8ph_kernel (conv1d)");
    return 1;
  }
  const bool conv = d.A.mode == S2SVC_OP_CONV2D_S2, tconv = d.A.mode == S2SVC_OP_TCONV2D_S2;
  const int bm = geo == 2 ? 512 : 256, bn = geo == 1 ? 256 : 128;
  dim3 grid((unsigned)((d.N + bn - 1) / bn), (unsigned)((d.M + bm - 1) / bm), (unsigned)nb);
#define P8_LAUNCH(KERNEL, ...)                                                                            \
  do {                                                                                                    \
    if (mode == 2) {                                                                                      \
      if (conv) hipLaunchKernelGGL((KERNEL<P8_CONV2D, ##__VA_ARGS__, false>), grid, dim3(512), 0, st, d); \
      else if (tconv) hipLaunchKernelGGL((KERNEL<P8_TCONV2D, ##__VA_ARGS__, false>), grid, dim3(512), 0, st, d); \
      else hipLaunchKernelGGL((KERNEL<P8_DENSE, ##__VA_ARGS__, false>), grid, dim3(512), 0, st, d);       \
    } else {                                                                                              \
      if (conv) hipLaunchKernelGGL((KERNEL<P8_CONV2D, ##__VA_ARGS__, true>), grid, dim3(512), 0, st, d);  \
      else if (tconv) hipLaunchKernelGGL((KERNEL<P8_TCONV2D, ##__VA_ARGS__, true>), grid, dim3(512), 0, st, d); \
      else hipLaunchKernelGGL((KERNEL<P8_DENSE, ##__VA_ARGS__, true>), grid, dim3(512), 0, st, d);        \
    }                                                                                                     \
  } while (0)
  static const bool lean_on = true;
  if (lean_on && !conv && !tconv && mode != 2 && epilogue_common_ok(d)) {       // dense operands + the common epilogue: lean variants
    if (geo == 1) hipLaunchKernelGGL((gemm_8ph_kernel_q<P8_DENSE, 2, 4, true, true>), grid, dim3(512), 0, st, d);
    else if (geo == 2) hipLaunchKernelGGL((gemm_8ph_kernel_q<P8_DENSE, 4, 2, true, true>), grid, dim3(512), 0, st, d);
    else hipLaunchKernelGGL((gemm_8ph_kernel_128<P8_DENSE, true, 1>), grid, dim3(512), 0, st, d);
  } else if (lean_on && geo == 3 && !conv && !tconv && mode != 2 && epilogue_swish_ok(d)) {        // the Conformer feed-forward pair
    hipLaunchKernelGGL((gemm_8ph_kernel_128<P8_DENSE, true, 2>), grid, dim3(512), 0, st, d);
  } else if (geo == 1) P8_LAUNCH(gemm_8ph_kernel_q, 2, 4);
  else if (geo == 2) P8_LAUNCH(gemm_8ph_kernel_q, 4, 2);
  else P8_LAUNCH(gemm_8ph_kernel_128);
#undef P8_LAUNCH
  S2S_CHECK_LAUNCH("gemm_8ph_kernel");
  return 1;
}

// the grouped weight-gradient launch on the 8-wave kernel: runs every descriptor that has exact 256 x 128 tiles (p8_tr_ok) and
// returns the bit mask of those (0 = none: switched off, or nothing eligible); the caller runs the others on gemm_grouped_kernel.
// WHICH kernel a problem gets depends on its own shape only, never on what shares the launch: the two kernels sum the bias
// row-sums in different orders, and a staged backward pass (other flush points, other groups) must give the bits of the
// uncut one (tests/gpu_model_check.py: stage_graphs_replay_equals_eager_full_size).  The tile width (256 / 128) does depend
// on the group -- it changes the schedule, not the order of any sum.
extern "C" int s2svc_gemm_grouped_try_8ph(const s2svc_gemm_desc* descs, int n, void* stream) {
  const int mode = p8_mode();
  if (mode == 0 || !p8_tr_mode() || n <= 0 || n > P8_GROUP_MAX) return 0;
  p8_group_args g;
  std::memset(&g, 0, sizeof(g));
  int64_t t128 = 0, t256 = 0;
  bool all256 = true;
  int mask = 0, m = 0;
  for (int i = 0; i < n; ++i) {
    // (and at least 64 tiles of 128 x 128 of its own -- a shape-only rule as well: a 256 x 1536 output is 12 of these tiles)
    if (!p8_tr_ok(descs[i]) || (int64_t)(descs[i].M / 128) * (descs[i].N / 128) < 64) continue;
    mask |= 1 << i;
    g.d[m++] = descs[i];
    t128 += (int64_t)(descs[i].M / 256) * (descs[i].N / 128);
    all256 = all256 && descs[i].N % 256 == 0;
    t256 += (int64_t)(descs[i].M / 256) * (descs[i].N / 256);
  }
  if (m == 0) return 0;
  S2S_REQUIRE(t128 < (1ll << 30), "gemm_grouped: too many tiles");
  g.n = m;
  const bool q = p8_tr_take_q(all256, t256, t128);
  int64_t total = 0;
  for (int i = 0; i < m; ++i) {
    g.tile_start[i] = (int32_t)total;
    total += (int64_t)(g.d[i].M / 256) * (g.d[i].N / (q ? 256 : 128));
  }
  for (int i = m; i <= P8_GROUP_MAX; ++i) g.tile_start[i] = (int32_t)total;
  hipStream_t st = (hipStream_t)stream;
  if (q) {
    if (mode == 2) hipLaunchKernelGGL((gemm_8ph_tr_grouped_kernel<false, 256>), dim3((unsigned)total), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((gemm_8ph_tr_grouped_kernel<true, 256>), dim3((unsigned)total), dim3(512), 0, st, g);
  } else {
    if (mode == 2) hipLaunchKernelGGL((gemm_8ph_tr_grouped_kernel<false, 128>), dim3((unsigned)total), dim3(512), 0, st, g);
    else hipLaunchKernelGGL((gemm_8ph_tr_grouped_kernel<true, 128>), dim3((unsigned)total), dim3(512), 0, st, g);
  }
  S2S_CHECK_LAUNCH("gemm_8ph_tr_grouped_kernel");
  return mask;
}

// ---- ragged weight gradients on the 8-wave kernel (W8) -----------------------------------------------------------
namespace {
int g_w8_mode = -1, g_w8_kt = -1;
int w8_mode() {       // s2svc_gemm_set_w8(0, .): these problems stay on gemm_grouped_kernel<64, 64> (A/B switch)
  if (g_w8_mode < 0) g_w8_mode = 1;
  return g_w8_mode;
}
int w8_kt_chunk_env() {   // s2svc_gemm_set_w8(., kt): K tiles (of 64 rows) per chunk; reductions up to this long run unsplit
  if (g_w8_kt < 0) g_w8_kt = 64;
  return g_w8_kt;
}
// the chunking of a reduction: a function of the problem's OWN shape only (see the kernel's header) -- outputs of >= 64 tiles fill the
// chip unsplit (and their partial tiles would be hundreds of MB of workspace traffic: AAS-VC's 4608 x 1536 gradients), smaller ones
// are cut by K.  Chunk length 64 K tiles (4096 rows): measured in the steps -- VTN's 2016 / 2048-row reductions are one chunk at 32 or 64
// (3.80 ms either way; 16 / 8 / 4: 3.87 / 3.94 / 4.10), AAS-VC's 4096-row reductions run unsplit at 64 (11.47 vs 11.68 ms at 32: no
// partial tiles, no reduction launch; 128: the same)
void w8_chunks(int M, int N, int K, int& nchunks, int& kt_chunk, bool conv = false) {
  const int ktiles = (K + 63) / 64;
  if (conv) {       // the Conv2d weight gradient: a long reduction (tens of thousands of pixels) over ~50 tiles.  VTN's 384 x 3456 over 38304
    // pixels, stand-alone: chunks of 64 / 75 / 86 / 100 / 120 / 150 / 200 K tiles = 182 / 145 / 165 / 184 / 201 / 153 / 196 us (the 4-wave split-K
    // kernel: 152); the VTN step is the same for all of them (3.65-3.67 ms) -- 75
    static const int ck = 75;
    kt_chunk = ck < 1 ? 1 : ck;
    nchunks = (ktiles + kt_chunk - 1) / kt_chunk;
    return;
  }
  if ((int64_t)((M + 255) / 256) * ((N + 127) / 128) >= 64) {
    nchunks = 1;
    kt_chunk = ktiles;
    return;
  }
  kt_chunk = w8_kt_chunk_env();
  nchunks = (ktiles + kt_chunk - 1) / kt_chunk;
  if (nchunks > 16) {
    kt_chunk = (ktiles + 15) / 16;
    nchunks = (ktiles + kt_chunk - 1) / kt_chunk;
  }
}
bool w8_ok(const s2svc_gemm_desc& d) {
  if (p8_mode() == 0 || !p8_tr_mode() || !w8_mode()) return false;
  if (d.dtype != S2S_BF16 || d.c_dtype != S2S_F32 || d.nb0 * d.nb1 != 1 || d.splitk > 1) return false;
  if (d.A.layout != S2SVC_LAYOUT_RC || d.B.layout != S2SVC_LAYOUT_RC || d.A.mode != S2SVC_OP_DENSE) return false;
  const bool convB = d.B.mode == S2SVC_OP_CONV2D_S2;         // the Conv2d 3x3 stride 2 weight gradient: B = implicit im2col of the layer's input
  const bool conv1B = d.B.mode == S2SVC_OP_CONV1D;           // the Conv1d weight gradient (big outputs only: the aligner's 1536 x 4608)
  if (d.B.mode != S2SVC_OP_DENSE && !convB && !conv1B) return false;
  if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.M % 8 || d.N % 8) return false;
  if (((uintptr_t)d.A.ptr) % 16 || ((uintptr_t)d.B.ptr) % 16 || d.A.ld % 8 || d.B.ld % 8 || d.A.ld < d.M) return false;
  if ((int64_t)64 * d.A.ld * 2 + (int64_t)d.M * 2 >= (1ll << 32)) return false;
  if (convB) {
    static const bool conv_on = true;
    if (!conv_on || d.B.C < 128 || d.B.C % 128 || d.N != 9 * d.B.C || d.B.ld < d.B.C) return false;
    if (d.B.T1 <= 0 || d.B.F1 <= 0 || d.B.T2 <= 0 || d.B.F2 <= 0 || 2 * (d.B.T2 - 1) + 3 > d.B.T1 || 2 * (d.B.F2 - 1) + 3 > d.B.F1) return false;
    if (d.K % (d.B.T2 * d.B.F2)) return false;               // whole images
    if ((int64_t)(d.K + 64) * (d.B.T2 * d.B.F2) >= (1ll << 32)) return false;       // the multiply-high divisions are exact below that
  } else if (conv1B) {
    static const bool conv1_on = true;
    if (!conv1_on || d.B.C < 128 || d.B.C % 128 || d.B.pad < 0 || d.N != (2 * d.B.pad + 1) * d.B.C || d.B.ld < d.B.C) return false;
    if (d.B.T <= 0 || d.K % d.B.T) return false;             // whole utterances
    if ((int64_t)(d.K + 64) * d.B.T >= (1ll << 32)) return false;
    if ((int64_t)((d.M + 255) / 256) * ((d.N + 127) / 128) < 64) return false;       // few tiles (the Postnet's 256 x 1280): the split-K 4-wave kernel
  } else {
    if (d.B.ld < d.N || (int64_t)64 * d.B.ld * 2 + (int64_t)d.N * 2 >= (1ll << 32)) return false;
  }
  if (((uintptr_t)d.C) % 16 || d.ldc % 4 || d.ldc < d.N) return false;
  if (d.bias || d.res || d.act != S2S_ACT_NONE || d.alpha != 1.0f || d.emask || d.drop_p > 0.f || d.c_map || d.c_pre) return false;
  // the exact-256 problems with >= 64 tiles of 128 x 128 (AAS-VC's decoder) ran on p8_tr_tile / p8_tr_tile_q until the loader-
  // specialised tile beat both per flop (0.83 us per 256 x 128 K tile against 1.05)
  static const bool exact_too = true;
  if (!exact_too && w8_mode() != 2 && p8_tr_ok(d) && (int64_t)(d.M / 128) * (d.N / 128) >= 64) return false;
  return true;
}
int64_t w8_ws_floats(const s2svc_gemm_desc& d) {
  int nc, kc;
  w8_chunks(d.M, d.N, d.K, nc, kc, d.B.mode == S2SVC_OP_CONV2D_S2);
  if (nc <= 1) return 0;
  int64_t f = (int64_t)nc * d.M * d.N;
  if (d.a_rowsum) f += (((int64_t)nc * d.M + 3) / 4) * 4;
  return f;
}
}  // namespace

// A/B switch for tests and benchmarks: on = 0 / 1 (< 0: unchanged), kt_chunk = K tiles per chunk (<= 0: unchanged).  Returns the
// previous (on | kt_chunk << 8).  A process that trains keeps both fixed: the chunking decides the order of summation.
extern "C" int s2svc_gemm_set_w8(int on, int kt_chunk) {
  const int prev = w8_mode() | (w8_kt_chunk_env() << 8);
  if (on >= 0) g_w8_mode = on > 2 ? 1 : on;
  if (kt_chunk > 0) g_w8_kt = kt_chunk;
  return prev;
}

// 1 if the 8-wave ragged weight-gradient kernel takes this problem (a function of the descriptor only)
extern "C" int s2svc_gemm_wgrad_ok(const s2svc_gemm_desc* desc) { return desc && w8_ok(*desc) ? 1 : 0; }

// fp32 elements of workspace s2svc_gemm_wgrad_grouped needs for these problems (0: every reduction runs unsplit)
extern "C" int64_t s2svc_gemm_wgrad_ws_floats(const s2svc_gemm_desc* descs, int n) {
  int64_t f = 0;
  for (int i = 0; i < n; ++i) f += w8_ws_floats(descs[i]);
  return f;
}

// every descriptor must satisfy s2svc_gemm_wgrad_ok; no two of them may write the same C / a_rowsum; `ws` (device, 16-byte
// aligned, s2svc_gemm_wgrad_ws_floats(...) floats, may be NULL if that is 0) must stay untouched until the launches have run
extern "C" int s2svc_gemm_wgrad_grouped_bg(const s2svc_gemm_desc* descs, int n, float* ws, void* stream, int wgs_cap);

extern "C" int s2svc_gemm_wgrad_grouped(const s2svc_gemm_desc* descs, int n, float* ws, void* stream) {
  return s2svc_gemm_wgrad_grouped_bg(descs, n, ws, stream, 0);
}

// wgs_cap > 0: a grid of at most that many workgroups (they walk the units in order)
extern "C" int s2svc_gemm_wgrad_grouped_bg(const s2svc_gemm_desc* descs, int n, float* ws, void* stream, int wgs_cap) {
  S2S_REQUIRE(descs && n > 0, "gemm_wgrad_grouped: bad args");
  hipStream_t st = (hipStream_t)stream;
  const int mode = p8_mode();
  static const int cap_env = 0;
  const int cap = wgs_cap > 0 ? wgs_cap : cap_env;
  int64_t ws_off = 0;
  for (int i0 = 0; i0 < n; i0 += W8_MAX) {
    const int cnt = (n - i0 < W8_MAX) ? n - i0 : W8_MAX;
    w8_args g;
    std::memset(&g, 0, sizeof(g));
    int64_t total = 0;
    bool any_split = false, have_cv = false;
    for (int i = 0; i < cnt; ++i) {
      const s2svc_gemm_desc& d = descs[i0 + i];
      S2S_REQUIRE(w8_ok(d), "gemm_wgrad_grouped: a descriptor is not eligible (check s2svc_gemm_wgrad_ok first)");
      w8_prob& q = g.p[i];
      q.A = d.A.ptr; q.B = d.B.ptr; q.C = (float*)d.C; q.rowsum = d.a_rowsum;
      q.lda = (int32_t)d.A.ld; q.ldb = (int32_t)d.B.ld; q.ldc = (int32_t)d.ldc;
      q.M = d.M; q.N = d.N; q.K = d.K;
      q.tiles_m = (d.M + 255) / 256; q.tiles_n = (d.N + 127) / 128;
      int nc, kc;
      w8_chunks(d.M, d.N, d.K, nc, kc, d.B.mode == S2SVC_OP_CONV2D_S2);
      q.nchunks = nc; q.kt_chunk = kc;
      q.flags = (d.accumulate ? 1 : 0) | (d.a_rowsum_accumulate ? 2 : 0);
      if (d.B.mode == S2SVC_OP_CONV2D_S2 || d.B.mode == S2SVC_OP_CONV1D) {
        const bool c1 = d.B.mode == S2SVC_OP_CONV1D;
        const w8_conv cv = {c1 ? d.B.T : d.B.T1, c1 ? d.B.pad : d.B.F1, c1 ? 1 : d.B.T2, c1 ? 1 : d.B.F2, d.B.C, c1 ? 1 : 0};
        S2S_REQUIRE(!have_cv || std::memcmp(&cv, &g.cv, sizeof(cv)) == 0, "gemm_wgrad_grouped: one Conv2d geometry per launch");
        g.cv = cv;
        have_cv = true;
        q.flags |= 4;
      }
      if (nc > 1) {
        S2S_REQUIRE(ws != nullptr && ((uintptr_t)ws) % 16 == 0, "gemm_wgrad_grouped: split reductions need a 16-byte aligned workspace");
        any_split = true;
        q.ws = ws + ws_off;
        ws_off += (int64_t)nc * d.M * d.N;
        if (d.a_rowsum) { q.rs_ws = ws + ws_off; ws_off += (((int64_t)nc * d.M + 3) / 4) * 4; }
      }
      g.unit_start[i] = (int32_t)total;
      total += (int64_t)q.tiles_m * q.tiles_n * nc;
      S2S_REQUIRE(total < (1ll << 30), "gemm_wgrad_grouped: too many units");
    }
    for (int i = cnt; i <= W8_MAX; ++i) g.unit_start[i] = (int32_t)total;
    g.n = cnt;
    g.total = (int32_t)total;
    const unsigned wgs = (unsigned)((cap > 0 && total > cap) ? cap : total);
    hipLaunchKernelGGL(gemm_w8ls_kernel, dim3(wgs), dim3(768), 0, st, g);
    S2S_CHECK_LAUNCH("gemm_w8ls_kernel");
    if (any_split) {
      // (grid.x only schedules: every output is summed by one thread in chunk order) -- more workgroups for big split outputs
      // (the Conv2d weight gradient: 1.3 M outputs x 10 chunks)
      int64_t big4 = 0;
      for (int i = 0; i < cnt; ++i)
        if (g.p[i].nchunks > 1 && (int64_t)g.p[i].M * g.p[i].N / 4 > big4) big4 = (int64_t)g.p[i].M * g.p[i].N / 4;
      unsigned gx = big4 >= (1 << 18) ? (unsigned)(big4 / (256 * 4)) : 48u;
      if (gx > 512) gx = 512;
      hipLaunchKernelGGL(w8_reduce_kernel, dim3(gx, (unsigned)cnt), dim3(256), 0, st, g);
      S2S_CHECK_LAUNCH("w8_reduce_kernel");
    }
  }
  return 0;
}

