// rt.h -- minimal runtime shim: device memory / stream calls for the HIP build,
// plain host memory for the host-emulator build (tools/hostemu, tests only).
#pragma once
#include "prim.h"
#include <cstdlib>
#include <cstring>

namespace star { namespace rt {

#ifdef STAR_HOSTEMU
inline int dev_malloc(void** p, size_t bytes) { return posix_memalign(p, 256, bytes ? bytes : 256) ? 1 : 0; }
inline void dev_free(void* p) { free(p); }
inline int memset_async(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
inline int memcpy_h2d(void* d, const void* s, size_t n, hipStream_t) { memcpy(d, s, n); return 0; }
inline int memcpy_d2h(void* d, const void* s, size_t n, hipStream_t) { memcpy(d, s, n); return 0; }
inline int memcpy_d2d(void* d, const void* s, size_t n, hipStream_t) { memcpy(d, s, n); return 0; }
inline int stream_sync(hipStream_t) { return 0; }
inline int set_device(int) { return 0; }
inline int device_count() { return 1; }
inline bool device_is_gfx950(int) { return true; }
inline int device_cu_count(int) { return 256; }
inline const char* last_error_string() { return "hostemu"; }
inline int peek_error() { return 0; }
inline void* event_record(hipStream_t) { return nullptr; }
inline float event_elapsed_ms(void*, void*) { return 0.f; }
inline void event_destroy(void*) {}
// hipGraph capture of a launch sequence: not available on the emulator (callers fall back to eager launches)
struct GraphExec { void* g = nullptr; void* e = nullptr; };
constexpr bool graphs_available = false;
inline int stream_create(hipStream_t*) { return 1; }
inline void stream_destroy(hipStream_t) {}
inline int stream_wait_stream(hipStream_t, hipStream_t) { return 1; }
inline int capture_begin(hipStream_t) { return 1; }
inline int capture_end(hipStream_t, GraphExec*) { return 1; }
inline int graph_launch(const GraphExec&, hipStream_t) { return 1; }
inline void graph_destroy(GraphExec&) {}
#else
inline int dev_malloc(void** p, size_t bytes) {
  if (hipMalloc(p, bytes ? bytes : 256) == hipSuccess) return 0;
  (void)hipGetLastError();   // reported through the return value; do not leave it as HIP's sticky last error
  return 1;
}
inline void dev_free(void* p) { (void)hipFree(p); }
inline int memset_async(void* p, int v, size_t n, hipStream_t s) { return hipMemsetAsync(p, v, n, s) != hipSuccess; }
inline int memcpy_h2d(void* d, const void* s, size_t n, hipStream_t st) { return hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, st) != hipSuccess; }
inline int memcpy_d2h(void* d, const void* s, size_t n, hipStream_t st) { return hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, st) != hipSuccess; }
inline int memcpy_d2d(void* d, const void* s, size_t n, hipStream_t st) { return hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, st) != hipSuccess; }
inline int stream_sync(hipStream_t s) { return hipStreamSynchronize(s) != hipSuccess; }
inline int set_device(int d) { return hipSetDevice(d) != hipSuccess; }
inline int device_count() { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }
inline const char* last_error_string() { return hipGetErrorString(hipGetLastError()); }
inline bool device_is_gfx950(int d) {
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, d) != hipSuccess) return false;
  return strncmp(pr.gcnArchName, "gfx950", 6) == 0 && pr.sharedMemPerBlockOptin >= 160 * 1024;
}
inline int device_cu_count(int d) {
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, d) != hipSuccess || pr.multiProcessorCount <= 0) return 256;
  return pr.multiProcessorCount;
}
inline int peek_error() { return hipPeekAtLastError() != hipSuccess; }
inline void* event_record(hipStream_t s) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; (void)hipEventRecord(e, s); return (void*)e; }
inline float event_elapsed_ms(void* a, void* b) { float ms = 0.f; if (a && b) { (void)hipEventSynchronize((hipEvent_t)b); (void)hipEventElapsedTime(&ms, (hipEvent_t)a, (hipEvent_t)b); } return ms; }
inline void event_destroy(void* e) { if (e) (void)hipEventDestroy((hipEvent_t)e); }
// hipGraph capture of a launch sequence (unet.cpp: one whole forward).  Capture needs a real stream (not the legacy default
// stream torch hands out by default), so the executor owns one and orders it against the caller's stream with events.
struct GraphExec { hipGraph_t g = nullptr; hipGraphExec_t e = nullptr; };
constexpr bool graphs_available = true;
inline int stream_create(hipStream_t* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking) != hipSuccess; }
inline void stream_destroy(hipStream_t s) { if (s) (void)hipStreamDestroy(s); }
inline int stream_wait_stream(hipStream_t waiter, hipStream_t on) {   // everything queued on `on` so far happens before what `waiter` gets next
  hipEvent_t ev;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return 1;
  int rc = hipEventRecord(ev, on) != hipSuccess;
  if (!rc) rc = hipStreamWaitEvent(waiter, ev, 0) != hipSuccess;
  (void)hipEventDestroy(ev);   // released once the recorded work has completed
  return rc;
}
inline int capture_begin(hipStream_t s) { return hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess; }
inline int capture_end(hipStream_t s, GraphExec* out) {
  hipGraph_t g = nullptr;
  if (hipStreamEndCapture(s, &g) != hipSuccess || !g) return 1;
  hipGraphExec_t e = nullptr;
  if (hipGraphInstantiate(&e, g, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(g); return 1; }
  out->g = g; out->e = e;
  return 0;
}
inline int graph_launch(const GraphExec& x, hipStream_t s) { return hipGraphLaunch(x.e, s) != hipSuccess; }
inline void graph_destroy(GraphExec& x) {
  if (x.e) (void)hipGraphExecDestroy(x.e);
  if (x.g) (void)hipGraphDestroy(x.g);
  x.e = nullptr; x.g = nullptr;
}
#endif

}}  // namespace star::rt
