// A stand-in for the HIP runtime, for the HOST-SIDE sanitizer builds of the library (tests/test_host_sanitizers.py): every source under
// torch-assimilate_amd/csrc is compiled host-only (`hipcc --cuda-host-only`: kernels become launch stubs) with -fsanitize=address,undefined
// or -fsanitize=thread and linked against this file instead of libamdhip64.  "Device" memory is host memory, kernels do nothing (their
// launches are counted), events complete at once.  What runs for real is what the sanitizers are here for: the step driver's two launch
// threads, job queues, slot rotation, per-workspace tables, option snapshots, the custom-communicator callbacks.
// Test infrastructure only: nothing in the product links it.
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>

namespace {
std::atomic<long long> g_launches{0}, g_events{0};
std::mutex g_mu;
std::set<void*> g_allocs;
thread_local struct { dim3 g, b; size_t shm; hipStream_t s; } t_cfg;
struct StubEvent { std::atomic<long long> recorded{0}; };
void* dev_alloc(size_t n) {
  void* p = nullptr;
  if (posix_memalign(&p, 256, n ? n : 1) != 0) return nullptr;
  memset(p, 0, n);
  std::lock_guard<std::mutex> lk(g_mu);
  g_allocs.insert(p);
  return p;
}
}  // namespace

extern "C" long long mia_stub_launch_count() { return g_launches.load(); }
extern "C" long long mia_stub_live_allocations() { std::lock_guard<std::mutex> lk(g_mu); return (long long)g_allocs.size(); }

extern "C" {
hipError_t hipMalloc(void** p, size_t n) { *p = dev_alloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { *p = dev_alloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) {
  if (!p) return hipSuccess;
  { std::lock_guard<std::mutex> lk(g_mu); if (!g_allocs.erase(p)) return hipErrorInvalidValue; }
  free(p);
  return hipSuccess;
}
hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(new StubEvent()); ++g_events; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { if (!a || !b || !ms) return hipErrorInvalidHandle; *ms = 0.01f; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<StubEvent*>(e); --g_events; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { if (!e) return hipErrorInvalidHandle; ++reinterpret_cast<StubEvent*>(e)->recorded; return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { if (!e) return hipErrorInvalidHandle; (void)reinterpret_cast<StubEvent*>(e)->recorded.load(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { if (!e) return hipErrorInvalidHandle; (void)reinterpret_cast<StubEvent*>(e)->recorded.load(); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) { return (reinterpret_cast<uintptr_t>(s) & 4) ? hipErrorNotReady : hipSuccess; }      // (some streams busy, some idle)
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { if (!e) return hipErrorInvalidHandle; (void)reinterpret_cast<StubEvent*>(e)->recorded.load(); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) { *st = hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*) { return hipErrorNotSupported; }
hipError_t hipIpcOpenMemHandle(void**, hipIpcMemHandle_t, unsigned) { return hipErrorNotSupported; }
hipError_t hipIpcCloseMemHandle(void*) { return hipErrorNotSupported; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { ++g_launches; return hipSuccess; }
hipError_t hipExtLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t, hipEvent_t start, hipEvent_t stop, int) {
  ++g_launches;
  if (start) ++reinterpret_cast<StubEvent*>(start)->recorded;
  if (stop) ++reinterpret_cast<StubEvent*>(stop)->recorded;
  return hipSuccess;
}
// what clang's kernel-launch stubs and fat-binary registration call
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t shm, hipStream_t s) { t_cfg.g = g; t_cfg.b = b; t_cfg.shm = shm; t_cfg.s = s; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* shm, hipStream_t* s) { *g = t_cfg.g; *b = t_cfg.b; *shm = t_cfg.shm; *s = t_cfg.s; return hipSuccess; }
void** __hipRegisterFatBinary(const void*) { static void* handle = nullptr; return &handle; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, char*, int, size_t, int, int) {}
void __hipRegisterManagedVar(void**, void**, void*, const char*, size_t, unsigned) {}
void __hipUnregisterFatBinary(void**) {}
}
