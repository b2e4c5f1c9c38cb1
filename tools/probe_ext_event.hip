// Does hipEventRecordWithFlags(hipEventRecordExternal) inside a stream capture give an event-record node that orders a second
// stream's hipStreamWaitEvent (issued after hipGraphLaunch) behind that point of THIS launch?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void add1(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
__global__ void spin(float* p, int n, int iters) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { float v = p[i]; for (int k = 0; k < iters; ++k) v = v * 1.0000001f + 1e-9f; p[i] = v; } }
__global__ void copy8(const float* a, float* c) { if (threadIdx.x < 8) c[threadIdx.x] = a[threadIdx.x]; }
int main() {
  const int n = 1 << 22;
  float *a, *junk, *c;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&junk, n * 4)); CK(hipMalloc(&c, 64));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(junk, 0, n * 4));
  hipStream_t mainS, side;
  CK(hipStreamCreate(&mainS)); CK(hipStreamCreate(&side));
  hipEvent_t ev, t0, t1, t2;
  CK(hipEventCreate(&ev)); CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1)); CK(hipEventCreate(&t2));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(mainS, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 10; ++i) spin<<<n / 256, 256, 0, mainS>>>(junk, n, 200);
  add1<<<n / 256, 256, 0, mainS>>>(a, n);
  hipError_t er = hipEventRecordWithFlags(ev, mainS, hipEventRecordExternal);
  printf("hipEventRecordWithFlags(external) in capture: %s\n", hipGetErrorString(er));
  if (er != hipSuccess) return 2;
  for (int i = 0; i < 100; ++i) spin<<<n / 256, 256, 0, mainS>>>(junk, n, 200);
  CK(hipStreamEndCapture(mainS, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  bool ok = true, overlapped = true;
  for (int it = 0; it < 5; ++it) {
    CK(hipEventRecord(t0, mainS));
    CK(hipGraphLaunch(ge, mainS));
    CK(hipEventRecord(t2, mainS));
    CK(hipStreamWaitEvent(side, ev, 0));
    copy8<<<1, 64, 0, side>>>(a, c);
    CK(hipEventRecord(t1, side));
    CK(hipDeviceSynchronize());
    float h[8]; CK(hipMemcpy(h, c, 32, hipMemcpyDeviceToHost));
    float ms1, ms2; CK(hipEventElapsedTime(&ms1, t0, t1)); CK(hipEventElapsedTime(&ms2, t0, t2));
    printf("launch %d: side stream saw a = %.0f (expect %d); side done at %.3f ms, graph done at %.3f ms\n", it, h[0], it + 1, ms1, ms2);
    ok = ok && h[0] == (float)(it + 1);
    overlapped = overlapped && ms1 < 0.7f * ms2;
  }
  printf("%s %s\n", ok ? "ORDERED" : "STALE", overlapped ? "OVERLAPPED" : "NOT-OVERLAPPED");
  return 0;
}
