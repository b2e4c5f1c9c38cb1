// Micro-benchmark: what does an in-kernel exchange between workgroups cost on MI355X, compared with a kernel boundary?
//   hipcc --offload-arch=gfx950 -O3 -o xcd_barrier_bench tools/xcd_barrier_bench.hip && ./xcd_barrier_bench
// 256 workgroups (one per CU; team membership from HW_REG_XCC_ID), 500 x [write, barrier, read the neighbour's data, barrier].
// Result (profiles/r02_xcd_barrier_bench.txt): 3.7-4.8 us per barrier + exchange even inside one XCD and with L2-local
// atomics -- the same as a dependent kernel launch inside a hipGraph (~4.5 us).  A persistent decode "megakernel" with
// in-kernel barriers therefore buys nothing over a chain of launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf; }

struct Ctl {
  unsigned reg_total;          // WGs registered
  unsigned pad0[31];
  unsigned xcc_count[16];      // WGs per XCC
  unsigned pad1[16];
  unsigned err;
  unsigned pad2[31];
  unsigned cnt[260 * 32];       // barrier counters, one 128-byte line per team
};

__device__ bool spin_until(unsigned* p, unsigned target, unsigned* err) {
  for (unsigned it = 0; it < (1u << 16); ++it) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
  }
  *err = 1;
  return false;
}

__device__ __forceinline__ unsigned ld_l2(unsigned* p) {   // returning atomic add of 0: executes in the L2
  unsigned v, z = 0;
  asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(z) : "memory");
  return v;
}
__device__ __forceinline__ void add_l2(unsigned* p, unsigned v) {
  asm volatile("global_atomic_add %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ bool spin_l2(unsigned* p, unsigned target, unsigned* err) {
  for (unsigned it = 0; it < (1u << 14); ++it) {
    if (ld_l2(p) >= target) return true;
  }
  *err = 1;
  return false;
}

// mode 0: team barrier (C WGs of one XCD), mode 1: device-wide barrier
__global__ __launch_bounds__(256) void bench(Ctl* ctl, int C, int nbar, int mode, float* data, unsigned long long* out_cycles, int* out_info) {
  extern __shared__ char smem[];
  __shared__ int s_team, s_rank, s_xcc, s_ok;
  const int G = gridDim.x;
  if (threadIdx.x == 0) {
    const int x = xcc_id();
    const unsigned t = atomicAdd(&ctl->xcc_count[x], 1u);
    s_xcc = x; s_team = x * 32 + t / C; s_rank = t % C;
    __threadfence();
    atomicAdd(&ctl->reg_total, 1u);
    s_ok = spin_until(&ctl->reg_total, G, &ctl->err) ? 1 : 0;
    out_info[blockIdx.x * 4 + 0] = x; out_info[blockIdx.x * 4 + 1] = t;
  }
  __syncthreads();
  if (!s_ok) return;
  const int team = s_team, rank = s_rank;
  unsigned* cnt = mode != 1 ? &ctl->cnt[team * 32] : &ctl->cnt[258 * 32];
  const int members = mode != 1 ? C : G;
  unsigned epoch = 0;
  float* tdata = data + (size_t)(mode != 1 ? team : 0) * 4096;
  const int me = mode != 1 ? rank : blockIdx.x;
  unsigned long long t0 = wall_clock64();
  int bad = 0;
  for (int b = 0; b < nbar; ++b) {
    if (__hip_atomic_load(&ctl->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) break;
    // write my slot, barrier, read neighbour's slot
    if (threadIdx.x < 64) tdata[me * 64 + threadIdx.x] = (float)(b * 1000 + me);
    __syncthreads();                                  // s_waitcnt vmcnt(0) + s_barrier: the stores of every wave are acknowledged
    if (threadIdx.x == 0) {
      epoch += members;
      if (mode == 2) { add_l2(cnt, 1u); spin_l2(cnt, epoch, &ctl->err); }
      else { __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); spin_until(cnt, epoch, &ctl->err); }
    }
    __syncthreads();
    if (mode == 2) asm volatile("buffer_inv sc1" ::: "memory");
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop stale L1 lines
    const int nb = (me + 1) % members;
    if (threadIdx.x < 64) {
      const float v = tdata[nb * 64 + threadIdx.x];
      if (v != (float)(b * 1000 + nb)) bad++;
    }
    // second barrier so that nobody overwrites before the neighbour has read
    __syncthreads();
    if (threadIdx.x == 0) {
      epoch += members;
      if (mode == 2) { add_l2(cnt, 1u); spin_l2(cnt, epoch, &ctl->err); }
      else { __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); spin_until(cnt, epoch, &ctl->err); }
    }
    __syncthreads();
  }
  unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { out_cycles[blockIdx.x] = t1 - t0; out_info[blockIdx.x * 4 + 2] = team; out_info[blockIdx.x * 4 + 3] = rank; }
  if (bad) atomicAdd(&ctl->err, 16u);
}

int main(int argc, char** argv) {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
  const int G = p.multiProcessorCount;
  printf("CUs %d, wall clock rate %d kHz\n", G, p.clockRate);
  Ctl* ctl; CK(hipMalloc(&ctl, sizeof(Ctl)));
  float* data; CK(hipMalloc(&data, 64 * 64 * 4096 * sizeof(float)));
  unsigned long long* cyc; CK(hipMalloc(&cyc, G * 8));
  int* info; CK(hipMalloc(&info, G * 16));
  int wc_khz = 0; CK(hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, dev));
  printf("wall clock %d kHz\n", wc_khz);
  CK(hipFuncSetAttribute((const void*)bench, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  for (int mode = 0; mode < 3; ++mode)
    for (int C : {1, 4, 16, 32}) {
      if (mode == 1 && C != 32) continue;
      CK(hipMemset(ctl, 0, sizeof(Ctl)));
      const int nbar = 500;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(bench, dim3(G), dim3(256), 100 * 1024, 0, ctl, C, nbar, mode, data, cyc, info);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      Ctl h; CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
      std::vector<unsigned long long> hc(G); CK(hipMemcpy(hc.data(), cyc, G * 8, hipMemcpyDeviceToHost));
      std::vector<int> hi(G * 4); CK(hipMemcpy(hi.data(), info, G * 16, hipMemcpyDeviceToHost));
      unsigned long long mx = 0; for (auto c : hc) mx = c > mx ? c : mx;
      printf("mode %d C %2d: kernel %.3f ms, %d x 2 barriers: %.3f us per barrier (event), max cycles %llu -> %.3f us per barrier; err %u; xcc counts:",
             mode, C, ms, nbar, ms * 1e3 / (2 * nbar), mx, (double)mx / (wc_khz * 1e-3) / (2 * nbar), h.err);
      for (int x = 0; x < 16; ++x) printf(" %u", h.xcc_count[x]);
      printf("\n"); fflush(stdout);
      if (mode == 0 && C == 16) { printf("first WGs (xcc,ticket): "); for (int i = 0; i < 20; ++i) printf("(%d,%d) ", hi[i * 4], hi[i * 4 + 1]); printf("\n"); }
    }
  return 0;
}
