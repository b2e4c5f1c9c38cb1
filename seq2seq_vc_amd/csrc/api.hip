// Error plumbing + ABI version of libs2svc_hip.so.
#include <string.h>
#include "common.h"
#include "../../include/s2svc_hip.h"

static thread_local char g_err[512] = "";

extern "C" void s2svc_set_error(const char* msg) {
  size_t n = strlen(g_err);
  if (n && n + 2 < sizeof(g_err)) { g_err[n++] = ':'; g_err[n++] = ' '; g_err[n] = 0; }
  strncat(g_err, msg, sizeof(g_err) - n - 1);
}
extern "C" const char* s2svc_last_error(void) {
  static thread_local char out[512];
  strncpy(out, g_err, sizeof(out));
  g_err[0] = 0;
  return out;
}
extern "C" int s2svc_abi_version(void) { return 2; }

// Hand-off points INSIDE a captured graph (round 6; distributed.OverlappedBackward "marks"): an event recorded on a capturing stream
// with hipEventRecordExternal becomes an event-record NODE of the graph -- every launch of the graph records it when the node's
// dependencies have run -- and a hipStreamWaitEvent issued on another stream AFTER hipGraphLaunch waits for that point of THAT launch
// (tools/probe_ext_event.hip: ordered and overlapped on this stack; torch's own Event(external=True) is refused on ROCm builds).
// The gradient exchange of a finished backward stage can then start while the SAME graph keeps running the next stage: the
// granularity of the exchange no longer costs a graph boundary per bucket.  Outside a capture _record is a plain hipEventRecord.
extern "C" int s2svc_event_create(void** out) {
  S2S_REQUIRE(out != nullptr, "event_create: null out");
  hipEvent_t ev;
  // (default flags: an event created with hipEventDisableTiming is refused by hipEventRecordWithFlags(.., hipEventRecordExternal) on ROCm 7)
  if (hipEventCreate(&ev) != hipSuccess) { s2svc_set_error("event_create: hipEventCreate failed"); return -2; }
  *out = (void*)ev;
  return 0;
}
extern "C" int s2svc_event_destroy(void* ev) {
  if (ev && hipEventDestroy((hipEvent_t)ev) != hipSuccess) { s2svc_set_error("event_destroy failed"); return -2; }
  return 0;
}
extern "C" int s2svc_event_record(void* ev, void* stream) {
  S2S_REQUIRE(ev != nullptr, "event_record: null event");
  hipStream_t st = (hipStream_t)stream;
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &status) != hipSuccess) { s2svc_set_error("event_record: hipStreamIsCapturing failed"); return -2; }
  if (status != hipStreamCaptureStatusActive) {
    const hipError_t e = hipEventRecord((hipEvent_t)ev, st);
    if (e != hipSuccess) { s2svc_set_error("event_record"); s2svc_set_error(hipGetErrorString(e)); return -2; }
    return 0;
  }
  // capturing: hipEventRecordWithFlags(.., hipEventRecordExternal) is the one-call form (it works on the ROCm 7.2 runtime,
  // tools/probe_ext_event.hip), but the HIP runtime bundled with the torch wheel (7.0) answers "invalid argument" -- so the node is
  // added by hand: the capture's graph and its current frontier from hipStreamGetCaptureInfo_v2, an event-record node behind that
  // frontier, and the frontier moved onto the node so that it stays inside the chain
  hipError_t e = hipEventRecordWithFlags((hipEvent_t)ev, st, hipEventRecordExternal);
  if (e == hipSuccess) return 1;
  (void)hipGetLastError();
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  e = hipStreamGetCaptureInfo_v2(st, &status, &id, &graph, &deps, &ndeps);
  if (e != hipSuccess || !graph) { s2svc_set_error("event_record: hipStreamGetCaptureInfo_v2"); s2svc_set_error(hipGetErrorString(e)); return -2; }
  hipGraphNode_t node = nullptr;
  e = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, (hipEvent_t)ev);
  if (e != hipSuccess) { s2svc_set_error("event_record: hipGraphAddEventRecordNode"); s2svc_set_error(hipGetErrorString(e)); return -2; }
  e = hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies);
  if (e != hipSuccess) { s2svc_set_error("event_record: hipStreamUpdateCaptureDependencies"); s2svc_set_error(hipGetErrorString(e)); return -2; }
  return 1;
}
extern "C" int s2svc_stream_wait_event(void* stream, void* ev) {
  S2S_REQUIRE(ev != nullptr, "stream_wait_event: null event");
  const hipError_t e = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0);
  if (e != hipSuccess) { s2svc_set_error("stream_wait_event"); s2svc_set_error(hipGetErrorString(e)); return -2; }
  return 0;
}

// The launch floor of this stack, measured rather than assumed: a kernel whose workgroups do nothing but store one word each.
// bench.py times it inside the same graph loops as the memory-bound kernels (their bytes / s are rated against the HBM peak with
// and without this floor), tools/gemm8_bench.py subtracts it from the fixed cost of a GEMM launch.
namespace {
__global__ void launch_floor_kernel(unsigned* __restrict__ sink) {
  if (threadIdx.x == 0) sink[blockIdx.x & 1023] = blockIdx.x;
}
}  // namespace
extern "C" int s2svc_launch_floor(int workgroups, int threads, void* sink_1024_words, void* stream) {
  S2S_REQUIRE(workgroups > 0 && threads > 0 && threads <= 1024 && sink_1024_words, "launch_floor: bad args");
  hipLaunchKernelGGL(launch_floor_kernel, dim3((unsigned)workgroups), dim3((unsigned)threads), 0, (hipStream_t)stream, (unsigned*)sink_1024_words);
  S2S_CHECK_LAUNCH("launch_floor_kernel");
  return 0;
}
