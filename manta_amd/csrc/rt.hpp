// Host-side runtime shim: the product build talks to the HIP runtime; the test-only emulator build
// (-DMANTA_WAVE_EMU, tests/emu/) maps "device" memory to host memory and runs kernels on the lock-step
// wave emulator.  Nothing in the product library selects between them at run time.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>

#ifdef MANTA_WAVE_EMU
#include <cstdlib>
#include <cstring>
#include "wave.hpp"
namespace rt {
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
inline void  init(int) {}
inline std::string deviceName() { return "wave-emulator (test infrastructure)"; }
inline int   cuCount() { return 2; }
inline size_t freeBytes() { return size_t(1) << 30; }
inline int   poisonByte() { const char* e = std::getenv("MANTA_EMU_POISON"); return e ? std::atoi(e) : 0xab; }  // "uninitialised" memory pattern
inline void* dmalloc(size_t n) { void* p = std::malloc(n ? n : 1); if (!p) throw Error("emu malloc failed"); std::memset(p, poisonByte(), n); return p; }
inline void  dfree(void* p) { std::free(p); }
inline void* dmallocFine(size_t n) { return dmalloc(n); }
inline void  h2d(void* d, const void* h, size_t n) { if (n) std::memcpy(d, h, n); }
inline void  d2h(void* h, const void* d, size_t n) { if (n) std::memcpy(h, d, n); }
inline void  d2hAsync(void* h, const void* d, size_t n) { if (n) std::memcpy(h, d, n); }
inline void  d2d(void* to, const void* from, size_t n) { if (n) std::memcpy(to, from, n); }
inline bool  streamWrite32(void* d, uint32_t v) { *static_cast<uint32_t*>(d) = v; return true; }
inline void  setDevice(int) {}
inline int   currentDevice() { return 0; }
inline void  dzero(void* d, size_t n) { if (n) std::memset(d, 0, n); }
inline void  dfill(void* d, int byte, size_t n) { if (n) std::memset(d, byte, n); }
inline void  sync() {}
template <typename K, typename P>
inline void launch(K kernel, int grid, size_t ldsBytes, const P& params) { wv_emu::launch(grid, ldsBytes, [&]() { kernel(params); }); }
template <typename K, typename P>
inline void launchSingle(K kernel, int grid, size_t ldsBytes, const P& params) { wv_emu::launch(grid, ldsBytes, [&]() { kernel(params); }); }
template <typename K, typename P>
inline void launchWG(K kernel, int grid, int nWaves, size_t ldsBytes, const P& params) { wv_emu::launchWG(grid, nWaves, ldsBytes, [&]() { kernel(params); }); }
inline int roundGrid(int waves) { return waves; }
template <typename K>
inline int blocksPerCu(K, int, size_t, int fallback) { return fallback; }
struct Stream {
  Stream() {}
  Stream(int, int) {}
  bool masked() const { return false; }
};
inline void useStream(Stream*) {}
struct ScopedStream { explicit ScopedStream(Stream&) {} };
struct Event { void record() {} void recordOn(Stream&) {} void sync() {} };
inline void curStreamWaits(Event&) {}
inline void streamWaits(Stream&, Event&) {}
inline float elapsedMs(const Event&, const Event&) { return 0.f; }
inline void* hostAlloc(size_t n) { void* p = std::malloc(n ? n : 1); if (!p) throw Error("emu malloc failed"); return p; }
inline void  hostFree(void* p) { std::free(p); }
}  // namespace rt
#else
#include <hip/hip_runtime.h>
namespace rt {
struct Error : std::runtime_error { using std::runtime_error::runtime_error; };
inline void check(hipError_t e, const char* what)
{
  if (e != hipSuccess) throw Error(std::string(what) + ": " + hipGetErrorString(e));
}
inline void init(int dev)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) throw Error("no HIP device available (this library has no CPU path)");
  if (dev >= 0) check(hipSetDevice(dev), "hipSetDevice");
}
inline void setDevice(int dev) { check(hipSetDevice(dev), "hipSetDevice"); }
inline int  currentDevice()
{
  int dev = 0;
  check(hipGetDevice(&dev), "hipGetDevice");
  return dev;
}
inline std::string deviceName()
{
  int dev = 0;
  check(hipGetDevice(&dev), "hipGetDevice");
  hipDeviceProp_t p;
  check(hipGetDeviceProperties(&p, dev), "hipGetDeviceProperties");
  return std::string(p.gcnArchName) + " / " + std::to_string(p.multiProcessorCount) + " CUs / " + std::to_string(p.totalGlobalMem) + " B";
}
inline int cuCount()
{
  int dev = 0;
  check(hipGetDevice(&dev), "hipGetDevice");
  hipDeviceProp_t p;
  check(hipGetDeviceProperties(&p, dev), "hipGetDeviceProperties");
  return p.multiProcessorCount;
}
inline size_t freeBytes()
{
  size_t f = 0, t = 0;
  check(hipMemGetInfo(&f, &t), "hipMemGetInfo");
  return f;
}
inline void* dmalloc(size_t n)
{
  void* p = nullptr;
  check(hipMalloc(&p, n ? n : 1), "hipMalloc");
  return p;
}
inline void dfree(void* p) { (void)hipFree(p); }
/// fine-grained device memory: not cached in the (per-XCD, mutually incoherent) L2s, so a kernel that polls it sees a value the
/// copy engine writes while the kernel runs
inline void* dmallocFine(size_t n)
{
  void* p = nullptr;
  check(hipExtMallocWithFlags(&p, n ? n : 1, hipDeviceMallocFinegrained), "hipExtMallocWithFlags(finegrained)");
  return p;
}
/// a non-blocking stream.  Every pipeline object / context owns one and makes it the calling thread's CURRENT stream for
/// the duration of an API call (ScopedStream): copies, memsets, launches and events below all go to the current stream, so
/// pipelines driven by different host threads overlap on the device (nothing here touches the null stream or
/// hipDeviceSynchronize).
struct Stream {
  hipStream_t s = nullptr;
  bool        cuMasked = false;
  Stream() { check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate"); }
  /// a stream whose kernels run on `cuCount - reservedCus` of the device's CUs only: the lowest `reservedCus` bits of the queue's CU mask
  /// are cleared.  The mask's bits interleave the XCDs (bit i belongs to XCD i mod 8), so a multiple of 8 takes the same number of CUs out
  /// of every XCD.  What is left out stays free for the kernels of the other streams -- a persistent launch on this stream cannot starve
  /// them.  Falls back to an ordinary stream if the runtime refuses the mask (masked() says which).
  Stream(int cuCount, int reservedCus)
  {
    if (reservedCus > 0 && reservedCus < cuCount) {
      uint32_t mask[32];
      const unsigned words = unsigned(cuCount + 31) / 32;
      for (unsigned w = 0; w < 32; ++w) mask[w] = 0;
      for (int cu = reservedCus; cu < cuCount; ++cu) mask[cu >> 5] |= 1u << (cu & 31);
      if (words <= 32 && hipExtStreamCreateWithCUMask(&s, words, mask) == hipSuccess) {
        cuMasked = true;
        return;
      }
      (void)hipGetLastError();
      s = nullptr;
    }
    check(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate");
  }
  bool masked() const { return cuMasked; }
  ~Stream() { if (s) (void)hipStreamDestroy(s); }
  Stream(const Stream&) = delete;
  Stream& operator=(const Stream&) = delete;
};
inline hipStream_t& launchStream()
{
  static thread_local hipStream_t cur = nullptr;  // null stream unless a ScopedStream / useStream() says otherwise
  return cur;
}
inline void useStream(Stream* st) { launchStream() = st ? st->s : nullptr; }
struct ScopedStream {
  hipStream_t prev;
  explicit ScopedStream(Stream& st) : prev(launchStream()) { launchStream() = st.s; }
  ~ScopedStream() { launchStream() = prev; }
  ScopedStream(const ScopedStream&) = delete;
  ScopedStream& operator=(const ScopedStream&) = delete;
};
inline void sync() { check(hipStreamSynchronize(launchStream()), "hipStreamSynchronize"); }
/// host -> device on the current stream.  Pageable sources are staged by the runtime before the call returns; pinned
/// sources (hostAlloc) are DMA'd asynchronously -- callers sync() before they let the source go.
inline void h2d(void* d, const void* h, size_t n) { if (n) check(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, launchStream()), "hipMemcpy H2D"); }
/// device -> host on the current stream; returns when the bytes are on the host
inline void d2h(void* h, const void* d, size_t n)
{
  if (!n) return;
  check(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, launchStream()), "hipMemcpy D2H");
  sync();
}
/// device -> device on the current stream (asynchronous)
inline void d2d(void* to, const void* from, size_t n) { if (n) check(hipMemcpyAsync(to, from, n, hipMemcpyDeviceToDevice, launchStream()), "hipMemcpy D2D"); }
/// same without the wait: several copies, then one sync()
inline void d2hAsync(void* h, const void* d, size_t n) { if (n) check(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, launchStream()), "hipMemcpy D2H"); }
/// a 32-bit write performed by the command processor once everything queued on the current stream before it has completed: no
/// copy kernel, no DMA engine (a tiny hipMemcpyAsync is a shader copy that needs a free workgroup slot).  False if the runtime
/// refuses (beta API): the caller then falls back to a 4-byte copy.
inline bool streamWrite32(void* d, uint32_t v)
{
  static const bool disabled = std::getenv("MANTA_AMD_NO_STREAM_WRITE") != nullptr;
  if (disabled) return false;
  const hipError_t e = hipStreamWriteValue32(launchStream(), d, v, 0);
  if (e != hipSuccess) (void)hipGetLastError();
  return e == hipSuccess;
}
inline void dzero(void* d, size_t n) { if (n) check(hipMemsetAsync(d, 0, n, launchStream()), "hipMemset"); }
inline void dfill(void* d, int byte, size_t n) { if (n) check(hipMemsetAsync(d, byte, n, launchStream()), "hipMemset"); }
/// page-locked host memory (staging buffers of the pipelines; manta_host_alloc)
inline void* hostAlloc(size_t n)
{
  void* p = nullptr;
  check(hipHostMalloc(&p, n ? n : 1, hipHostMallocDefault), "hipHostMalloc");
  return p;
}
inline void hostFree(void* p) { (void)hipHostFree(p); }
/// kernels launched with more than the default 64 KB of dynamic LDS: the limit is raised to the whole CU's 160 KB once per
/// kernel and thread (the same kernel is launched with different sizes: contig_kernel's size classes)
inline void allowFullLds(const void* kernel)
{
  static thread_local const void* prepared[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  static thread_local int         devOf[8]    = {-1, -1, -1, -1, -1, -1, -1, -1};
  const int dev = currentDevice();
  for (int i = 0; i < 8; ++i)
    if (prepared[i] == kernel && devOf[i] == dev) return;
  check(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840), "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
  for (int i = 0; i < 8; ++i)
    if (prepared[i] == nullptr || i == 7) {
      prepared[i] = kernel;
      devOf[i]    = dev;
      break;
    }
}
template <typename K, typename P>
inline void launch(K kernel, int grid, size_t ldsBytes, const P& params)
{
  // `grid` counts WAVEFRONTS and must be a multiple of WV_WAVES_PER_WG (see roundGrid); ldsBytes is per wave
  hipLaunchKernelGGL(kernel, dim3(grid / WV_WAVES_PER_WG), dim3(64 * WV_WAVES_PER_WG), ldsBytes * WV_WAVES_PER_WG, launchStream(), params);
  check(hipGetLastError(), "kernel launch");
}
/// `grid` single-wave workgroups with `ldsBytes` of dynamic LDS each (may exceed the 64 KB default limit)
template <typename K, typename P>
inline void launchSingle(K kernel, int grid, size_t ldsBytes, const P& params)
{
  allowFullLds(reinterpret_cast<const void*>(kernel));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(64), ldsBytes, launchStream(), params);
  check(hipGetLastError(), "kernel launch");
}
/// `grid` cooperative workgroups of `nWaves` wavefronts with `ldsBytes` of dynamic LDS each
template <typename K, typename P>
inline void launchWG(K kernel, int grid, int nWaves, size_t ldsBytes, const P& params)
{
  allowFullLds(reinterpret_cast<const void*>(kernel));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * nWaves), ldsBytes, launchStream(), params);
  check(hipGetLastError(), "kernel launch");
}
inline int roundGrid(int waves) { return ((waves + WV_WAVES_PER_WG - 1) / WV_WAVES_PER_WG) * WV_WAVES_PER_WG; }
/// workgroups of `threads` threads and `ldsBytes` of dynamic LDS the runtime places on one CU (`fallback` if it does not say)
template <typename K>
inline int blocksPerCu(K kernel, int threads, size_t ldsBytes, int fallback)
{
  allowFullLds(reinterpret_cast<const void*>(kernel));
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kernel), threads, ldsBytes) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fallback;
  }
  return n;
}
/// HIP event, recorded on the current stream
struct Event {
  hipEvent_t e = nullptr;
  Event() { check(hipEventCreate(&e), "hipEventCreate"); }
  ~Event() { if (e) (void)hipEventDestroy(e); }
  Event(const Event&) = delete;
  Event& operator=(const Event&) = delete;
  void record() { check(hipEventRecord(e, launchStream()), "hipEventRecord"); }
  void recordOn(Stream& st) { check(hipEventRecord(e, st.s), "hipEventRecord"); }
  void sync() { check(hipEventSynchronize(e), "hipEventSynchronize"); }  // the host waits
};
inline void curStreamWaits(Event& ev) { check(hipStreamWaitEvent(launchStream(), ev.e, 0), "hipStreamWaitEvent"); }
inline void streamWaits(Stream& st, Event& ev) { check(hipStreamWaitEvent(st.s, ev.e, 0), "hipStreamWaitEvent"); }
inline float elapsedMs(const Event& a, const Event& b)
{
  float ms = 0.f;
  check(hipEventSynchronize(b.e), "hipEventSynchronize");
  check(hipEventElapsedTime(&ms, a.e, b.e), "hipEventElapsedTime");
  return ms;
}
}  // namespace rt
#endif
