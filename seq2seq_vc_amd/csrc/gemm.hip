// Tiled MFMA GEMM for gfx950 with implicit-convolution operand addressing.
//
//   C[z][m,n] = act(alpha * sum_k A_z(m,k) * B_z(n,k) + bias[n]) + res_z[m,n]
//
// One kernel serves every dense contraction of the hot path (Linear / Conv1d / Conv2d-stride-2 /
// QK^T / PV and all their dgrad / wgrad forms): operands are described by (layout, mode) pairs
// instead of being materialised (no im2col, no transposed copies).
//
//   * 256 threads = 4 wavefronts (2x2); each wavefront owns a (BM/2)x(BN/2) block of 16x16 MFMA tiles.
//   * bf16: v_mfma_f32_16x16x32_bf16 (fp32 accumulate).  fp32: v_mfma_f32_16x16x4_f32, which is an
//     exact k-ordered fmaf chain -- this is the parity path against the fp32 CPU reference.
//   * global -> registers (16-byte vectors along the contiguous index) -> LDS [row][k] with padding;
//     the next K-tile's global loads are in flight while the current tile is multiplied.
//   * K-strided ("RC") operands are transposed on their way into LDS, so fragments are always read
//     as one 16-byte ds_read per 16x(8|4) sub-block.
#include <stdlib.h>
#include <cstdio>
#include <cstdlib>
#include "gemm_common.h"
#include "../../include/s2svc_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace {

constexpr int BK = 32;

template <typename T> struct TileCfg;
template <> struct TileCfg<float>  { static constexpr int VEC = 4; static constexpr int PITCH = 36; };
template <> struct TileCfg<bf16_t> { static constexpr int VEC = 8; static constexpr int PITCH = 40; };

// element offset of operand element (r, k); `valid` false => contributes zero
__device__ __forceinline__ int64_t op_addr(const s2svc_operand& o, int r, int k, bool& valid) {
  valid = true;
  if (o.mode == S2SVC_OP_DENSE)
    return o.layout == S2SVC_LAYOUT_KC ? (int64_t)r * o.ld + k : (int64_t)k * o.ld + r;
  int m, q;  // m: spatial (b,t[,f]) index, q: implicit (tap, channel) index
  if (o.layout == S2SVC_LAYOUT_KC) { m = r; q = k; } else { m = k; q = r; }
  int tap = q / o.C, c = q - tap * o.C;
  if (o.mode == S2SVC_OP_CONV1D) {
    int t = m % o.T, tt = t + tap - o.pad;
    valid = (tt >= 0) && (tt < o.T);
    return (int64_t)(m + tap - o.pad) * o.ld + c;
  }
  int f2 = m % o.F2, bt = m / o.F2;
  int t2 = bt % o.T2, b = bt / o.T2;
  int kh = tap / 3, kw = tap - kh * 3;
  return ((int64_t)(b * o.T1 + 2 * t2 + kh) * o.F1 + (2 * f2 + kw)) * o.ld + c;
}

template <typename T>
__device__ __forceinline__ void put_elem(uint4& v, int e, T x);
template <> __device__ __forceinline__ void put_elem<float>(uint4& v, int e, float x) {
  uint32_t u = __float_as_uint(x);
  if (e == 0) v.x = u; else if (e == 1) v.y = u; else if (e == 2) v.z = u; else v.w = u;
}
template <> __device__ __forceinline__ void put_elem<bf16_t>(uint4& v, int e, bf16_t x) {
  uint32_t sh = (e & 1) * 16, msk = ~(0xffffu << sh), u = ((uint32_t)x) << sh;
  int w = e >> 1;
  if (w == 0) v.x = (v.x & msk) | u; else if (w == 1) v.y = (v.y & msk) | u;
  else if (w == 2) v.z = (v.z & msk) | u; else v.w = (v.w & msk) | u;
}

// Load one 16-byte vector of an operand tile.  (r, k) is the first element; the vector runs along k
// for KC layouts and along r for RC layouts.  R/K are the logical extents (zero fill outside).
template <typename T>
__device__ __forceinline__ uint4 load_vec(const s2svc_operand& o, const T* base, int r, int k, int R, int K) {
  constexpr int VEC = TileCfg<T>::VEC;
  uint4 out = make_uint4(0, 0, 0, 0);
  const bool kc = (o.layout == S2SVC_LAYOUT_KC);
  if (r >= R || k >= K) return out;
  const int nr = kc ? 1 : (R - r < VEC ? R - r : VEC);
  const int nk = kc ? (K - k < VEC ? K - k : VEC) : 1;
  bool v0;
  int64_t a0 = op_addr(o, r, k, v0);
  const int n = kc ? nk : nr;
  bool fast = (n == VEC) && ((((uintptr_t)(base + a0)) & 15) == 0);
  if (fast && o.mode != S2SVC_OP_DENSE) fast = (o.C % VEC == 0) && (((kc ? k : r) % VEC) == 0);
  if (fast) {
    if (v0) out = *reinterpret_cast<const uint4*>(base + a0);
    return out;
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    if (e < n) {
      bool v;
      int64_t a = kc ? op_addr(o, r, k + e, v) : op_addr(o, r + e, k, v);
      if (v) put_elem<T>(out, e, base[a]);
    }
  }
  return out;
}

template <typename T>
__device__ __forceinline__ void store_tile_vec(T* lds, int layout, int row, int kk, uint4 v) {
  constexpr int VEC = TileCfg<T>::VEC, PITCH = TileCfg<T>::PITCH;
  if (layout == S2SVC_LAYOUT_KC) {
    *reinterpret_cast<uint4*>(lds + row * PITCH + kk) = v;
  } else {
    const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) lds[(row + i) * PITCH + kk] = e[i];
  }
}

template <typename T, int ROWS>
struct Stager {
  static constexpr int VEC = TileCfg<T>::VEC;
  static constexpr int NV = ROWS * BK / VEC / 256;
  static_assert(NV >= 1, "tile too small");
  uint4 reg[NV];
  __device__ __forceinline__ void vec_coord(int layout, int i, int& row, int& kk) const {
    int v = threadIdx.x + i * 256;
    if (layout == S2SVC_LAYOUT_KC) { row = v / (BK / VEC); kk = (v % (BK / VEC)) * VEC; }
    else { kk = v / (ROWS / VEC); row = (v % (ROWS / VEC)) * VEC; }
  }
  __device__ __forceinline__ void load(const s2svc_operand& o, const T* base, int r0, int k0, int R, int K) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int row, kk; vec_coord(o.layout, i, row, kk);
      reg[i] = load_vec<T>(o, base, r0 + row, k0 + kk, R, K);
    }
  }
  __device__ __forceinline__ void store(T* lds, int layout) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int row, kk; vec_coord(layout, i, row, kk);
      store_tile_vec<T>(lds, layout, row, kk, reg[i]);
    }
  }
};

template <typename T, int FM, int FN> struct Mma;
template <int FM, int FN> struct Mma<bf16_t, FM, FN> {
  static __device__ __forceinline__ void run(const bf16_t* As, const bf16_t* Bs, int wm, int wn, f32x4_t (&acc)[FM][FN]) {
    constexpr int P = TileCfg<bf16_t>::PITCH;
    const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
    bf16x8_t a[FM], b[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(As + (wm + i * 16 + lr) * P + lg * 8);
#pragma unroll
    for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn + j * 16 + lr) * P + lg * 8);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
  }
};
template <int FM, int FN> struct Mma<float, FM, FN> {
  static __device__ __forceinline__ void run(const float* As, const float* Bs, int wm, int wn, f32x4_t (&acc)[FM][FN]) {
    constexpr int P = TileCfg<float>::PITCH;
    const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f32x4_t a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const f32x4_t*>(As + (wm + i * 16 + lr) * P + ks * 16 + lg * 4);
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const f32x4_t*>(Bs + (wn + j * 16 + lr) * P + ks * 16 + lg * 4);
      // the 4 MFMAs of a 16-wide k block each take one element of the 4-vectors: MFMA e covers
      // k = {4g + e}; any consistent assignment is a valid partition of the reduction.
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
  }
};

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_kernel(const s2svc_gemm_desc d) {
  constexpr int PITCH = TileCfg<T>::PITCH;
  constexpr int FM = BM / 32, FN = BN / 32;
  __shared__ __attribute__((aligned(16))) T As[BM * PITCH];
  __shared__ __attribute__((aligned(16))) T Bs[BN * PITCH];

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

  Stager<T, BM> sa;
  Stager<T, BN> sb;
  if (kt_begin < kt_end) {
    sa.load(d.A, Ab, m0, kt_begin * BK, d.M, d.K);
    sb.load(d.B, Bb, n0, kt_begin * BK, d.N, d.K);
    sa.store(As, d.A.layout);
    sb.store(Bs, d.B.layout);
  }
  __syncthreads();
  constexpr int TPR = 256 / BM;  // threads per A row for the fused row sums (bias gradient of a wgrad GEMM)
  const bool do_rowsum = (d.a_rowsum != nullptr) && (blockIdx.x == 0);
  float rowsum = 0.f;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = (kt + 1 < kt_end);
    if (more) {
      sa.load(d.A, Ab, m0, (kt + 1) * BK, d.M, d.K);
      sb.load(d.B, Bb, n0, (kt + 1) * BK, d.N, d.K);
    }
    if (do_rowsum) {
      const T* rp = As + (threadIdx.x / TPR) * PITCH + (threadIdx.x % TPR) * (BK / TPR);
#pragma unroll
      for (int e = 0; e < BK / TPR; ++e) rowsum += ldf(rp + e);
    }
    Mma<T, FM, FN>::run(As, Bs, wm, wn, acc);
    __syncthreads();
    if (more) {
      sa.store(As, d.A.layout);
      sb.store(Bs, d.B.layout);
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

  const int lane = threadIdx.x & 63, lc = lane & 15, lq = lane >> 4;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + lq * 4 + r;
        const int n = n0 + wn + j * 16 + lc;
        if (m < d.M && n < d.N) {
          if (splitk > 1) {
            const int nbatch = d.nb0 * d.nb1;
            d.ws[(((int64_t)zs * nbatch + zb) * d.M + m) * d.N + n] = acc[i][j][r];
          } else {
            epilogue_store_f(d, z0, z1, m, n, acc[i][j][r]);
          }
        }
      }
}

__global__ void splitk_reduce_kernel(const s2svc_gemm_desc d) {
  const int nbatch = d.nb0 * d.nb1;
  const int64_t total = (int64_t)nbatch * d.M * d.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
#pragma unroll 8
    for (int z = 0; z < d.splitk; ++z) s += d.ws[(int64_t)z * total + i];      // the partials' loads in flight together, same order of additions
    const int n = (int)(i % d.N);
    const int64_t t = i / d.N;
    const int m = (int)(t % d.M);
    const int zb = (int)(t / d.M);
    epilogue_store_f(d, zb / d.nb1, zb % d.nb1, m, n, s);
  }
  if (d.a_rowsum) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.M; i += (int64_t)gridDim.x * blockDim.x) {
      float s = 0.f;
      for (int z = 0; z < d.splitk; ++z) s += d.a_rowsum_ws[(int64_t)z * d.M + i];
      d.a_rowsum[i] = (d.a_rowsum_accumulate ? d.a_rowsum[i] : 0.f) + s;
    }
  }
}

// second pass for the kernels whose epilogue does not carry the dropout / mask stage: C = stage(C) in place
__global__ void gemm_stage_kernel(const s2svc_gemm_desc d) {
  const int64_t total = (int64_t)d.M * d.N;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / d.N), n = (int)(i - (int64_t)m * d.N);
    const int64_t co = (int64_t)m * d.ldc + n;
    if (d.c_dtype == S2S_F32) {
      float* c = (float*)d.C + co;
      *c = epilogue_stage_f(d, m, n, *c);
    } else {
      bf16_t* c = (bf16_t*)d.C + co;
      *c = f2bf(epilogue_stage_f(d, m, n, bf2f(*c)));
    }
  }
}

int launch_stage_pass(const s2svc_gemm_desc& d, hipStream_t st) {
  const int64_t total = (int64_t)d.M * d.N;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gemm_stage_kernel, dim3(blocks), dim3(256), 0, st, d);
  S2S_CHECK_LAUNCH("gemm_stage_kernel");
  return 0;
}

template <typename T, int BM, int BN>
int launch(const s2svc_gemm_desc& d, hipStream_t st) {
  const int splitk = d.splitk > 1 ? d.splitk : 1;
  dim3 grid((d.N + BN - 1) / BN, (d.M + BM - 1) / BM, d.nb0 * d.nb1 * splitk);
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN>), grid, dim3(256), 0, st, d);
  S2S_CHECK_LAUNCH("gemm_kernel");
  if (splitk > 1) {
    const int64_t total = (int64_t)d.nb0 * d.nb1 * d.M * d.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, d);
    S2S_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return 0;
}

}  // namespace

extern "C" int s2svc_gemm_try_fast(const s2svc_gemm_desc* desc, void* stream);    // gemm_fast.hip
extern "C" int s2svc_gemm_try_skinny(const s2svc_gemm_desc* desc, void* stream);  // gemm_skinny.hip (M <= 64)
extern "C" int s2svc_gemm_try_glds(const s2svc_gemm_desc* desc, void* stream);    // gemm_glds.hip (bf16, LDS-DMA)
extern "C" int s2svc_gemm_try_8ph(const s2svc_gemm_desc* desc, void* stream);     // gemm_8ph.hip (bf16, 256-row tiles, 8 waves)

static bool generic_forced() {
  static int v = -1;
  if (v < 0) v = 0;
  return v == 1;
}

extern "C" int s2svc_gemm(const s2svc_gemm_desc* desc, void* stream) {
  S2S_REQUIRE(desc != nullptr, "s2svc_gemm: null desc");
  s2svc_gemm_desc d = *desc;
  if (d.nb0 < 1) d.nb0 = 1;
  if (d.nb1 < 1) d.nb1 = 1;
  S2S_REQUIRE(d.M >= 0 && d.N >= 0 && d.K >= 0, "s2svc_gemm: negative dims");
  if (d.M == 0 || d.N == 0) return 0;
  S2S_REQUIRE(d.A.ptr && d.B.ptr && d.C, "s2svc_gemm: null operand");
  S2S_REQUIRE(d.dtype == S2S_F32 || d.dtype == S2S_BF16, "s2svc_gemm: bad dtype");
  S2S_REQUIRE(d.c_dtype == S2S_F32 || d.c_dtype == S2S_BF16, "s2svc_gemm: bad c_dtype");
  S2S_REQUIRE(d.splitk <= 1 || d.ws != nullptr, "s2svc_gemm: splitk needs workspace");
  S2S_REQUIRE(!d.a_rowsum || (d.nb0 * d.nb1 == 1 && (d.splitk <= 1 || d.a_rowsum_ws)), "s2svc_gemm: a_rowsum needs nb == 1 (and a workspace with splitk)");
  S2S_REQUIRE(d.A.mode == S2SVC_OP_DENSE || d.A.C > 0, "s2svc_gemm: conv operand A needs C");
  S2S_REQUIRE(d.B.mode == S2SVC_OP_DENSE || d.B.C > 0, "s2svc_gemm: conv operand B needs C");
  hipStream_t st = (hipStream_t)stream;
  if (d.A.mode == S2SVC_OP_TCONV2D_S2 || d.B.mode == S2SVC_OP_TCONV2D_S2 || d.c_map) {
    // transposed-convolution operand / mapped C rows: only the bf16 LDS-DMA kernels implement them
    S2S_REQUIRE(d.B.mode != S2SVC_OP_TCONV2D_S2, "s2svc_gemm: S2SVC_OP_TCONV2D_S2 is an A-operand mode");
    S2S_REQUIRE(d.nb0 * d.nb1 == 1 && d.splitk <= 1 && !d.res && d.drop_p == 0.f && !d.a_rowsum,
                "s2svc_gemm: tconv2d / c_map GEMMs are unbatched, unsplit and have no residual / dropout stage");
    S2S_REQUIRE(!d.emask || (d.emask_mode == 0 && d.ldm == d.ldc), "s2svc_gemm: a c_map GEMM takes a relu mask with C's row layout only");
    S2S_REQUIRE(!d.c_map || (d.cm_Tc > 0 && d.cm_Fc > 0 && d.M % (d.cm_Tc * d.cm_Fc) == 0), "s2svc_gemm: bad c_map grid");
    int rc = s2svc_gemm_try_8ph(&d, stream);         // the 8-wave kernel takes the transposed-convolution operand and mapped C rows
    if (rc == 0) rc = s2svc_gemm_try_glds(&d, stream);
    S2S_REQUIRE(rc != 0, "s2svc_gemm: tconv2d / c_map need bf16 operands the LDS-DMA kernel accepts (16-byte aligned, C % 8 == 0, C >= 64)");
    return rc < 0 ? rc : 0;
  }
  // dropout / mask stage: native in the LDS-DMA kernels' epilogue; every other kernel family gets it as a second pass
  // over C (which is only the same thing when nothing is added to C after the stage)
  S2S_REQUIRE(!d.c_pre || !d.c_map, "s2svc_gemm: c_pre (pre-activation output) is not available with c_map");
  S2S_REQUIRE(d.emask_mode == 0 || (d.emask_mode == 1 && d.emask), "s2svc_gemm: emask_mode 1 needs emask (the pre-activation)");
  const bool staged = d.drop_p > 0.f || d.emask != nullptr;
  S2S_REQUIRE(!staged || (d.nb0 * d.nb1 == 1 && d.ldc == d.N), "s2svc_gemm: the dropout/mask stage needs an unbatched contiguous C");
  S2S_REQUIRE(!staged || d.drop_p < 1.f, "s2svc_gemm: drop_p must be < 1");
  auto finish = [&](bool stage_done) -> int {
    if (!staged || stage_done) return 0;
    S2S_REQUIRE(!d.res && !d.accumulate, "s2svc_gemm: dropout/mask stage with residual/accumulate needs the LDS-DMA kernel");
    return launch_stage_pass(d, st);
  };
  if (!generic_forced()) {
    const int rs = s2svc_gemm_try_skinny(&d, stream);
    if (rs != 0) return rs < 0 ? rs : finish(false);
    int rc = d.c_map ? 0 : s2svc_gemm_try_8ph(&d, stream);
    if (rc == 0) rc = s2svc_gemm_try_glds(&d, stream);
    if (rc == 0) rc = s2svc_gemm_try_fast(&d, stream);      // (its epilogue is the staged one as well since round 3)
    const bool native_stage = rc == 1 && d.splitk <= 1;
    if (rc < 0) return rc;
    if (rc == 1) {
      if (d.splitk > 1) {
        const int64_t total = (int64_t)d.nb0 * d.nb1 * d.M * d.N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, d);
        S2S_CHECK_LAUNCH("splitk_reduce_kernel");
      }
      return finish(native_stage);
    }
  }
  {  // S2SVC_GEMM_LOG=1: report every problem the specialised kernels declined (tuning aid)
    static int log = -1;
    if (log < 0) { const char* e = getenv("S2SVC_GEMM_LOG"); log = (e && e[0] == '1') ? 1 : 0; }
    if (log == 1)
      fprintf(stderr, "[s2svc_gemm generic] M=%d N=%d K=%d nb=%dx%d splitk=%d dtype=%d A(layout=%d mode=%d ld=%lld C=%d) B(layout=%d mode=%d ld=%lld C=%d)\n",
              d.M, d.N, d.K, d.nb0, d.nb1, d.splitk, d.dtype, d.A.layout, d.A.mode, (long long)d.A.ld, d.A.C, d.B.layout, d.B.mode,
              (long long)d.B.ld, d.B.C);
  }
  const int64_t tiles128 = (int64_t)((d.M + 127) / 128) * ((d.N + 127) / 128) * d.nb0 * d.nb1 * (d.splitk > 1 ? d.splitk : 1);
  const bool big = tiles128 >= 384 && d.M >= 128 && d.N >= 128;
  int rg;
  if (d.dtype == S2S_F32) rg = big ? launch<float, 128, 128>(d, st) : launch<float, 64, 64>(d, st);
  else rg = big ? launch<bf16_t, 128, 128>(d, st) : launch<bf16_t, 64, 64>(d, st);
  return rg < 0 ? rg : finish(false);
}
