// Host <-> device staging used by the C++ adapters (ORBmatcher.h, Optimizer.h): the matcher / tracking-optimiser entry points of
// morb_hip.h take DEVICE pointers (their natural callers keep frames resident in HBM); an adapter that is handed the reference's
// host-side containers uploads them per call.  Plain HIP runtime C API — compiles with g++ (-I/opt/rocm/include, -lamdhip64).
#pragma once
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

namespace morb_adapter {

inline void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// The stream the calling thread's copies are queued on: an adapter method sets it to its handle's stream for the duration of the call
// (StreamScope), so that uploads, kernels and downloads of one reference call share one stream and nothing touches the null stream —
// a copy there would wait for every blocking stream of the device, i.e. for whatever LocalBundleAdjustment trial the mapping thread
// has in flight (System.cc:209 runs Tracking and LocalMapping side by side).  NULL (no scope) = the null stream, as before.
inline hipStream_t& current_stream() { static thread_local hipStream_t s = nullptr; return s; }
struct StreamScope {
  hipStream_t saved;
  explicit StreamScope(void* stream) : saved(current_stream()) { current_stream() = reinterpret_cast<hipStream_t>(stream); }
  ~StreamScope() { current_stream() = saved; }
  StreamScope(const StreamScope&) = delete;
  StreamScope& operator=(const StreamScope&) = delete;
};
inline void sync_current_stream() { hip_check(hipStreamSynchronize(current_stream()), "hipStreamSynchronize"); }

template <typename T>
class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  explicit DeviceBuffer(size_t n) { resize(n); }
  DeviceBuffer(const T* host, size_t n) { resize(n); upload(host, n); }
  explicit DeviceBuffer(const std::vector<T>& v) : DeviceBuffer(v.data(), v.size()) {}
  ~DeviceBuffer() { if (p_) (void)hipFree(p_); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  void resize(size_t n) {
    if (n <= cap_) { n_ = n; return; }
    if (p_) (void)hipFree(p_);
    hip_check(hipMalloc(reinterpret_cast<void**>(&p_), (n ? n : 1) * sizeof(T)), "hipMalloc");
    cap_ = n_ = n;
  }
  void assign(const T* host, size_t n) { resize(n); upload(host, n); }   // grow-only: a static / member buffer re-used from call to call
  // `host` may be a temporary of the caller: the copy is complete when upload() returns (hipMemcpy's contract, on the scope's stream)
  void upload(const T* host, size_t n) {
    if (!n) return;
    hip_check(hipMemcpyAsync(p_, host, n * sizeof(T), hipMemcpyHostToDevice, current_stream()), "hipMemcpy H2D");
    sync_current_stream();
  }
  void fill_bytes(int byte) { if (n_) hip_check(hipMemsetAsync(p_, byte, n_ * sizeof(T), current_stream()), "hipMemset"); }
  void download(T* host, size_t n) const {
    if (!n) return;
    hip_check(hipMemcpyAsync(host, p_, n * sizeof(T), hipMemcpyDeviceToHost, current_stream()), "hipMemcpy D2H");
    sync_current_stream();
  }
  std::vector<T> to_host() const { std::vector<T> v(n_); download(v.data(), n_); return v; }
  T* get() { return p_; }
  const T* get() const { return p_; }
  size_t size() const { return n_; }

 private:
  T* p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

}  // namespace morb_adapter
