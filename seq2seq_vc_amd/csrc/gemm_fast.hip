// Fast paths of the MFMA GEMM (same contract as gemm.hip::gemm_kernel, which stays as the fully general
// fallback).  What is different here:
//   * operand addressing is specialised at compile time (dense / implicit Conv1d / implicit Conv2d-s2, for
//     K-contiguous "KC" and row-contiguous "RC" operands): per-thread row bases are computed once, the K loop
//     only adds 32-bit tile offsets;
//   * deep K tiles (BK = 128 for the 64x64 tile, 64 for the 128x128 tile in bf16) so that the many small GEMMs
//     of this workload (K = 80..1536) finish in 3-12 global round trips, each with 4 x 16 B loads in flight
//     per thread and operand;
//   * RC operands (dgrad weights, wgrad activations, P^T / V in attention) are transposed in REGISTERS
//     (VECxVEC blocks, v_perm_b32 for bf16) and written to LDS as full 16-byte rows -- no scalar LDS stores.
// Requirements checked on the host (else the generic kernel runs): 16-byte aligned bases, leading dimensions
// and batch strides multiples of the vector length, channel counts multiples of the vector length.
#include "common.h"
#include "../../include/s2svc_hip.h"

#include "gemm_common.h"

namespace {

enum { AM_KC = 0, AM_RC = 1 };

template <typename T> struct VecCfg;
template <> struct VecCfg<float>  { static constexpr int VEC = 4; static constexpr int PAD = 4; };
template <> struct VecCfg<bf16_t> { static constexpr int VEC = 8; static constexpr int PAD = 8; };

// ---------------------------------------------------------------------------------------------
// KC loader: NV vectors per thread, vector = (row, kk..kk+VEC) ; element address = rowbase + f(k)
// ---------------------------------------------------------------------------------------------
template <typename T, int ROWS, int BK, bool DENSE = false>
struct KcLoader {
  static constexpr int VEC = VecCfg<T>::VEC;
  static constexpr int VPR = BK / VEC;             // vectors per row
  static constexpr int NV = ROWS * VPR / 256;
  static_assert(NV >= 1, "tile too small");
  uint4 reg[NV];
  int64_t rowbase[NV];
  int trow[NV];      // conv1d: frame index of the row ; <0: row out of range
  __device__ __forceinline__ void init(const s2svc_operand& o, int r0, int R) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = threadIdx.x + i * 256;
      const int r = r0 + v / VPR;
      trow[i] = 0;
      if (r >= R) { trow[i] = -1; rowbase[i] = 0; continue; }
      if (DENSE || o.mode == S2SVC_OP_DENSE) {
        rowbase[i] = (int64_t)r * o.ld;
      } else if (o.mode == S2SVC_OP_CONV1D) {
        rowbase[i] = (int64_t)r * o.ld;
        trow[i] = r % o.T;
      } else {
        const int f2 = r % o.F2, bt = r / o.F2;
        const int t2 = bt % o.T2, b = bt / o.T2;
        rowbase[i] = ((int64_t)(b * o.T1 + 2 * t2) * o.F1 + 2 * f2) * o.ld;
      }
    }
  }
  __device__ __forceinline__ void load(const s2svc_operand& o, const T* base, int k0, int K) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = threadIdx.x + i * 256;
      const int k = k0 + (v % VPR) * VEC;
      uint4 val = make_uint4(0, 0, 0, 0);
      if (trow[i] >= 0 && k < K) {
        if (DENSE || o.mode == S2SVC_OP_DENSE) {
          val = *reinterpret_cast<const uint4*>(base + rowbase[i] + k);
        } else {
          const int tap = k / o.C, c = k - tap * o.C;
          if (o.mode == S2SVC_OP_CONV1D) {
            const int tt = trow[i] + tap - o.pad;
            if (tt >= 0 && tt < o.T) val = *reinterpret_cast<const uint4*>(base + rowbase[i] + (int64_t)(tap - o.pad) * o.ld + c);
          } else {
            const int kh = tap / 3, kw = tap - kh * 3;
            val = *reinterpret_cast<const uint4*>(base + rowbase[i] + (int64_t)(kh * o.F1 + kw) * o.ld + c);
          }
        }
      }
      reg[i] = val;
    }
  }
  __device__ __forceinline__ void store(T* lds) const {
    constexpr int PITCH = BK + VecCfg<T>::PAD;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = threadIdx.x + i * 256;
      *reinterpret_cast<uint4*>(lds + (v / VPR) * PITCH + (v % VPR) * VEC) = reg[i];
    }
  }
};

// ---------------------------------------------------------------------------------------------
// RC loader: VEC x VEC register-block transpose.  Block (kb, rb): VEC loads of 16 B at k = k0+kb*VEC+j,
// rows r0+rb*VEC.. ; stored as VEC rows of 16 B.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Transposer;
template <> struct Transposer<float> {
  static __device__ __forceinline__ void run(const uint4 (&in)[4], uint4 (&out)[4]) {
    out[0] = make_uint4(in[0].x, in[1].x, in[2].x, in[3].x);
    out[1] = make_uint4(in[0].y, in[1].y, in[2].y, in[3].y);
    out[2] = make_uint4(in[0].z, in[1].z, in[2].z, in[3].z);
    out[3] = make_uint4(in[0].w, in[1].w, in[2].w, in[3].w);
  }
};
template <> struct Transposer<bf16_t> {
  static __device__ __forceinline__ uint32_t lo(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
  static __device__ __forceinline__ uint32_t hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }
  static __device__ __forceinline__ void run(const uint4 (&in)[8], uint4 (&out)[8]) {
    // out[e] = (in[0][e], in[1][e], ..., in[7][e]) ; element e of in[j] sits in dword e/2, half e%2
    out[0] = make_uint4(lo(in[0].x, in[1].x), lo(in[2].x, in[3].x), lo(in[4].x, in[5].x), lo(in[6].x, in[7].x));
    out[1] = make_uint4(hi(in[0].x, in[1].x), hi(in[2].x, in[3].x), hi(in[4].x, in[5].x), hi(in[6].x, in[7].x));
    out[2] = make_uint4(lo(in[0].y, in[1].y), lo(in[2].y, in[3].y), lo(in[4].y, in[5].y), lo(in[6].y, in[7].y));
    out[3] = make_uint4(hi(in[0].y, in[1].y), hi(in[2].y, in[3].y), hi(in[4].y, in[5].y), hi(in[6].y, in[7].y));
    out[4] = make_uint4(lo(in[0].z, in[1].z), lo(in[2].z, in[3].z), lo(in[4].z, in[5].z), lo(in[6].z, in[7].z));
    out[5] = make_uint4(hi(in[0].z, in[1].z), hi(in[2].z, in[3].z), hi(in[4].z, in[5].z), hi(in[6].z, in[7].z));
    out[6] = make_uint4(lo(in[0].w, in[1].w), lo(in[2].w, in[3].w), lo(in[4].w, in[5].w), lo(in[6].w, in[7].w));
    out[7] = make_uint4(hi(in[0].w, in[1].w), hi(in[2].w, in[3].w), hi(in[4].w, in[5].w), hi(in[6].w, in[7].w));
  }
};

template <typename T, int ROWS, int BK, bool DENSE = false>
struct RcLoader {
  static constexpr int VEC = VecCfg<T>::VEC;
  static constexpr int RB = ROWS / VEC;                 // row blocks
  static constexpr int NBLK = (BK / VEC) * RB;          // blocks per tile
  static constexpr int NB = (NBLK + 255) / 256;         // blocks per thread (some threads idle when NBLK < 256)
  uint4 reg[NB][VEC];
  __device__ __forceinline__ void init(const s2svc_operand&, int, int) {}
  __device__ __forceinline__ void load(const s2svc_operand& o, const T* base, int r0, int R, int k0, int K) {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int blk = threadIdx.x + n * 256;
      const bool active = blk < NBLK;
      const int rb = blk % RB, kb = blk / RB;
      const int r = r0 + rb * VEC;
      int tap = 0, c = r;
      if (!DENSE && o.mode != S2SVC_OP_DENSE) { tap = r / o.C; c = r - tap * o.C; }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int k = k0 + kb * VEC + j;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (active && r < R && k < K) {
          if (DENSE || o.mode == S2SVC_OP_DENSE) {
            val = *reinterpret_cast<const uint4*>(base + (int64_t)k * o.ld + r);
          } else if (o.mode == S2SVC_OP_CONV1D) {
            const int t = k % o.T, tt = t + tap - o.pad;
            if (tt >= 0 && tt < o.T) val = *reinterpret_cast<const uint4*>(base + (int64_t)(k + tap - o.pad) * o.ld + c);
          } else {
            const int f2 = k % o.F2, bt = k / o.F2;
            const int t2 = bt % o.T2, b = bt / o.T2;
            const int kh = tap / 3, kw = tap - kh * 3;
            val = *reinterpret_cast<const uint4*>(base + ((int64_t)(b * o.T1 + 2 * t2 + kh) * o.F1 + (2 * f2 + kw)) * o.ld + c);
          }
        }
        reg[n][j] = val;
      }
    }
  }
  __device__ __forceinline__ void store(T* lds) const {
    constexpr int PITCH = BK + VecCfg<T>::PAD;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const int blk = threadIdx.x + n * 256;
      if (blk < NBLK) {
        const int rb = blk % RB, kb = blk / RB;
        uint4 out[VEC];
        Transposer<T>::run(reg[n], out);
#pragma unroll
        for (int e = 0; e < VEC; ++e) *reinterpret_cast<uint4*>(lds + (rb * VEC + e) * PITCH + kb * VEC) = out[e];
      }
    }
  }
};

template <typename T, int ROWS, int BK, int MODE, bool DENSE = false> struct Loader;
template <typename T, int ROWS, int BK, bool DENSE> struct Loader<T, ROWS, BK, AM_KC, DENSE> {
  KcLoader<T, ROWS, BK, DENSE> l;
  __device__ __forceinline__ void init(const s2svc_operand& o, int r0, int R) { l.init(o, r0, R); }
  __device__ __forceinline__ void load(const s2svc_operand& o, const T* base, int r0, int R, int k0, int K) { l.load(o, base, k0, K); }
  __device__ __forceinline__ void store(T* lds) const { l.store(lds); }
};
template <typename T, int ROWS, int BK, bool DENSE> struct Loader<T, ROWS, BK, AM_RC, DENSE> {
  RcLoader<T, ROWS, BK, DENSE> l;
  __device__ __forceinline__ void init(const s2svc_operand& o, int r0, int R) { l.init(o, r0, R); }
  __device__ __forceinline__ void load(const s2svc_operand& o, const T* base, int r0, int R, int k0, int K) { l.load(o, base, r0, R, k0, K); }
  __device__ __forceinline__ void store(T* lds) const { l.store(lds); }
};

// ---------------------------------------------------------------------------------------------
template <typename T, int FM, int FN, int BK> struct MmaF;
template <int FM, int FN, int BK> struct MmaF<bf16_t, FM, FN, BK> {
  static __device__ __forceinline__ void run(const bf16_t* As, const bf16_t* Bs, int wm, int wn, f32x4_t (&acc)[FM][FN]) {
    constexpr int P = BK + VecCfg<bf16_t>::PAD;
    const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      bf16x8_t a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(As + (wm + i * 16 + lr) * P + ks * 32 + lg * 8);
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn + j * 16 + lr) * P + ks * 32 + lg * 8);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
};
template <int FM, int FN, int BK> struct MmaF<float, FM, FN, BK> {
  static __device__ __forceinline__ void run(const float* As, const float* Bs, int wm, int wn, f32x4_t (&acc)[FM][FN]) {
    constexpr int P = BK + VecCfg<float>::PAD;
    const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      f32x4_t a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const f32x4_t*>(As + (wm + i * 16 + lr) * P + ks * 16 + lg * 4);
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const f32x4_t*>(Bs + (wn + j * 16 + lr) * P + ks * 16 + lg * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
  }
};

// LEAN: dense operands and the common fp32 epilogue only (epilogue_flush_common32) -- the duration predictor's Linear layers
template <typename T, int BM, int BN, int BK, int AMODE, int BMODE, bool LEAN = false>
__global__ __launch_bounds__(256) void gemm_fast_kernel(const s2svc_gemm_desc d) {
  constexpr int PITCH = BK + VecCfg<T>::PAD;
  constexpr int FM = BM / 32, FN = BN / 32;
  // one array: the epilogue reuses it as the four waves' fp32 C tiles (BM * BN * 4 bytes: more than the operand tiles of the
  // 128 x 128 kernel need)
  constexpr size_t AB_BYTES = sizeof(T) * (BM + BN) * PITCH, C_BYTES = (size_t)BM * BN * 4;
  __shared__ __attribute__((aligned(16))) T smem_ab[(AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES) / sizeof(T)];
  T* As = smem_ab;
  T* Bs = smem_ab + BM * PITCH;

  const int splitk = d.splitk > 1 ? d.splitk : 1;
  const int zb = blockIdx.z / splitk, zs = blockIdx.z - zb * splitk;
  const int z0 = zb / d.nb1, z1 = zb - z0 * d.nb1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const T* Ab = (const T*)d.A.ptr + (int64_t)z0 * d.A.bs0 + (int64_t)z1 * d.A.bs1;
  const T* Bb = (const T*)d.B.ptr + (int64_t)z0 * d.B.bs0 + (int64_t)z1 * d.B.bs1;
  const int ktiles = (d.K + BK - 1) / BK;
  const int per = (ktiles + splitk - 1) / splitk;
  const int kt_begin = zs * per;
  const int kt_end = (kt_begin + per < ktiles) ? kt_begin + per : ktiles;
  const int wave = threadIdx.x >> 6;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  Loader<T, BM, BK, AMODE, LEAN> la;
  Loader<T, BN, BK, BMODE, LEAN> lb;
  la.init(d.A, m0, d.M);
  lb.init(d.B, n0, d.N);
  if (kt_begin < kt_end) {
    la.load(d.A, Ab, m0, d.M, kt_begin * BK, d.K);
    lb.load(d.B, Bb, n0, d.N, kt_begin * BK, d.K);
    la.store(As);
    lb.store(Bs);
  }
  __syncthreads();
  constexpr int TPR = 256 / BM;                  // threads per A row for the fused row sums
  const bool do_rowsum = (d.a_rowsum != nullptr) && (blockIdx.x == 0);
  float rowsum = 0.f;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = (kt + 1 < kt_end);
    if (more) {
      la.load(d.A, Ab, m0, d.M, (kt + 1) * BK, d.K);
      lb.load(d.B, Bb, n0, d.N, (kt + 1) * BK, d.K);
    }
    if (do_rowsum) {
      const T* rp = As + (threadIdx.x / TPR) * PITCH + (threadIdx.x % TPR) * (BK / TPR);
#pragma unroll 8
      for (int e = 0; e < BK / TPR; ++e) rowsum += ldf(rp + e);
    }
    MmaF<T, FM, FN, BK>::run(As, Bs, wm, wn, acc);
    __syncthreads();
    if (more) {
      la.store(As);
      lb.store(Bs);
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
  // the accumulators leave through wave-private fp32 LDS tiles (the operand tiles are dead): ONE rolled copy of the epilogue
  // instead of FM * FN * 4 inlined copies of the element-wise one (each with the activation switch: 40 KB of code in the fp32
  // 64 x 64 kernel, which the duration predictor of AAS-VC launches ~100 times per step)
  __syncthreads();
  float* cs = reinterpret_cast<float*>(smem_ab) + wave * (BM / 2) * (BN / 2);
  if (LEAN) {
    epilogue_stage<BM / 2, BN / 2>(acc, cs);
    if (splitk > 1) epilogue_partials<BM / 2, BN / 2>(d, zs, m0 + wm, n0 + wn, cs);
    else epilogue_flush_common32<BM / 2, BN / 2>(d, m0 + wm, n0 + wn, cs);
  } else {
    epilogue_tile<BM / 2, BN / 2>(d, z0, z1, m0 + wm, n0 + wn, acc, cs, splitk, zs, zb);
  }
}

template <typename T, int BM, int BN, int BK>
void launch_modes(const s2svc_gemm_desc& d, dim3 grid, hipStream_t st) {
  const bool arc = d.A.layout == S2SVC_LAYOUT_RC, brc = d.B.layout == S2SVC_LAYOUT_RC;
  static const bool lean_on = true;
  if constexpr (sizeof(T) == 4 && BM <= 64) {
    if (lean_on && d.A.mode == S2SVC_OP_DENSE && d.B.mode == S2SVC_OP_DENSE && epilogue_common32_ok(d)) {
      if (!arc && !brc) hipLaunchKernelGGL((gemm_fast_kernel<float, BM, BN, BK, AM_KC, AM_KC, true>), grid, dim3(256), 0, st, d);
      else if (!arc && brc) hipLaunchKernelGGL((gemm_fast_kernel<float, BM, BN, BK, AM_KC, AM_RC, true>), grid, dim3(256), 0, st, d);
      else if (arc && !brc) hipLaunchKernelGGL((gemm_fast_kernel<float, BM, BN, BK, AM_RC, AM_KC, true>), grid, dim3(256), 0, st, d);
      else hipLaunchKernelGGL((gemm_fast_kernel<float, BM, BN, BK, AM_RC, AM_RC, true>), grid, dim3(256), 0, st, d);
      return;
    }
  }
  if (!arc && !brc) hipLaunchKernelGGL((gemm_fast_kernel<T, BM, BN, BK, AM_KC, AM_KC>), grid, dim3(256), 0, st, d);
  else if (!arc && brc) hipLaunchKernelGGL((gemm_fast_kernel<T, BM, BN, BK, AM_KC, AM_RC>), grid, dim3(256), 0, st, d);
  else if (arc && !brc) hipLaunchKernelGGL((gemm_fast_kernel<T, BM, BN, BK, AM_RC, AM_KC>), grid, dim3(256), 0, st, d);
  else hipLaunchKernelGGL((gemm_fast_kernel<T, BM, BN, BK, AM_RC, AM_RC>), grid, dim3(256), 0, st, d);
}

bool operand_ok(const s2svc_operand& o, int vec, size_t esz) {
  if (((uintptr_t)o.ptr) % 16) return false;
  if (o.ld % vec || o.bs0 % vec || o.bs1 % vec) return false;
  if (o.mode != S2SVC_OP_DENSE && (o.C % vec)) return false;
  (void)esz;
  return true;
}

}  // namespace

// returns 1 if the fast path was launched, 0 if the caller must use the generic kernel, <0 on error
extern "C" int s2svc_gemm_try_fast(const s2svc_gemm_desc* desc, void* stream) {
  const s2svc_gemm_desc& d = *desc;
  const int vec = d.dtype == S2S_F32 ? 4 : 8;
  const size_t esz = d.dtype == S2S_F32 ? 4 : 2;
  if (!operand_ok(d.A, vec, esz) || !operand_ok(d.B, vec, esz)) return 0;
  // 16-byte vectors run along the row index (RC) or along k (KC): the extent must be a vector multiple, or -- for
  // dense operands -- the caller declares the rows zero-padded up to one (whole vectors are read unmasked: the
  // zero tail contributes nothing to the reduction and the extra rows are never stored)
  auto extent_ok = [&](const s2svc_operand& o, int extent) {
    if (extent % vec == 0) return true;
    return o.mode == S2SVC_OP_DENSE && o.zero_padded && o.ld >= (int64_t)((extent + vec - 1) / vec * vec);
  };
  if (!extent_ok(d.A, d.A.layout == S2SVC_LAYOUT_RC ? d.M : d.K)) return 0;
  if (!extent_ok(d.B, d.B.layout == S2SVC_LAYOUT_RC ? d.N : d.K)) return 0;
  if (d.tile_hint != 0 && d.tile_hint != 32 && d.tile_hint != 64 && d.tile_hint != 128) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int splitk = d.splitk > 1 ? d.splitk : 1;
  const int64_t tiles128 = (int64_t)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.nb0 * d.nb1 * splitk;
  bool big = tiles128 >= 256 && d.M >= 128 && d.N >= 128;
  if (d.tile_hint == 128) big = true;
  if (d.tile_hint == 64) big = false;
  if (big) {
    dim3 grid((d.N + 127) / 128, (d.M + 127) / 128, d.nb0 * d.nb1 * splitk);
    if (d.dtype == S2S_F32) launch_modes<float, 128, 128, 32>(d, grid, st);
    else launch_modes<bf16_t, 128, 128, 64>(d, grid, st);
  } else {
    // fp32 problems that leave most of the chip idle on 64 x 64 tiles (the duration predictor's 1024 x 384 x 384 Linear layers:
    // 96 workgroups; its 29-column spline projection: 16) take 32 x 32 tiles with K tiles of 128: the fp32 MFMA
    // (v_mfma_f32_16x16x4f32, 256 flop / clk / CU) makes a 64 x 64 x 384 workgroup 5 us of matrix-pipe time on its own;
    // a quarter of it per workgroup, on four times the workgroups.  Same order of every sum (k ascending), same bits.
    static const bool t32_on = true;
    const int64_t tiles64 = (int64_t)((d.M + 63) / 64) * ((d.N + 63) / 64) * d.nb0 * d.nb1 * splitk;
    const bool t32_ok = d.dtype == S2S_F32 && d.K >= 128 && d.A.mode == S2SVC_OP_DENSE && d.B.mode == S2SVC_OP_DENSE;
    if (t32_ok && ((t32_on && d.tile_hint == 0 && tiles64 < 128 && !d.a_rowsum) || d.tile_hint == 32)) {      // (hint 32: ops.kernels.plan_gemm)
      dim3 grid((d.N + 31) / 32, (d.M + 31) / 32, d.nb0 * d.nb1 * splitk);
      launch_modes<float, 32, 32, 128>(d, grid, st);
    } else {
      dim3 grid((d.N + 63) / 64, (d.M + 63) / 64, d.nb0 * d.nb1 * splitk);
      if (d.dtype == S2S_F32) launch_modes<float, 64, 64, 64>(d, grid, st);
      else launch_modes<bf16_t, 64, 64, 128>(d, grid, st);
    }
  }
  S2S_CHECK_LAUNCH("gemm_fast_kernel");
  return 1;
}
