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
  void upload(const T* host, size_t n) { if (n) hip_check(hipMemcpy(p_, host, n * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D"); }
  void fill_bytes(int byte) { if (n_) hip_check(hipMemset(p_, byte, n_ * sizeof(T)), "hipMemset"); }
  void download(T* host, size_t n) const { if (n) hip_check(hipMemcpy(host, p_, n * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H"); }
  std::vector<T> to_host() const { std::vector<T> v(n_); download(v.data(), n_); return v; }
  T* get() { return p_; }
  const T* get() const { return p_; }
  size_t size() const { return n_; }

 private:
  T* p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

}  // namespace morb_adapter
