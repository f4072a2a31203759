// Host <-> device staging used by the C++ adapters (ORBmatcher.h, Optimizer.h): the matcher / tracking-optimiser entry points of
// morb_hip.h take DEVICE pointers (their natural callers keep frames resident in HBM); an adapter that is handed the reference's
// host-side containers uploads them per call.  Plain HIP runtime C API — compiles with g++ (-I/opt/rocm/include, -lamdhip64).
#pragma once
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstring>
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
inline void sync_current_stream() { hip_check(hipStreamSynchronize(current_stream()), "hipStreamSynchronize"); }

// Pinned staging of the calling thread (round 5).  A reference call hands the adapter ~15 host arrays (keypoints, descriptors, map-point fields, ...);
// uploaded one by one from pageable memory, each copy was a staged transfer plus a stream synchronisation — 0.25 ms of a 0.33 ms
// SearchByProjection call.  Inside a StreamScope an upload now copies the array into this arena (so the caller's temporary may die at once) and queues
// an asynchronous copy from there: no wait until the call's results are downloaded.  The arena is rewound when the outermost scope of the thread
// opens (after the stream has drained) and grows by doubling; outgrown blocks live until that rewind.  (Never freed at thread exit: the HIP
// runtime may be gone by then.)
struct PinnedArena {
  char* base = nullptr;
  size_t cap = 0, off = 0;
  int depth = 0;
  hipStream_t lastStream = nullptr;   // the stream the most recent upload from the arena was queued on (the one to drain before a rewind)
  bool pending = false;               // an upload has been queued from the arena since the last rewind
  std::vector<char*> outgrown;
  void* take(size_t n) {
    n = (n + 63) & ~(size_t)63;
    if (off + n > cap) {
      size_t want = cap ? cap * 2 : (size_t)1 << 20;
      while (want < n) want *= 2;
      char* fresh = nullptr;
      hip_check(hipHostMalloc(reinterpret_cast<void**>(&fresh), want, hipHostMallocDefault), "hipHostMalloc");
      if (base) outgrown.push_back(base);   // copies queued from it are still in flight
      base = fresh; cap = want; off = 0;
    }
    void* p = base + off;
    off += n;
    return p;
  }
  void rewind() {
    for (char* o : outgrown) (void)hipHostFree(o);
    outgrown.clear();
    off = 0;
  }
};
inline PinnedArena& arena() { static thread_local PinnedArena* a = new PinnedArena(); return *a; }

struct StreamScope {
  hipStream_t saved;
  explicit StreamScope(void* stream) : saved(current_stream()) {
    current_stream() = reinterpret_cast<hipStream_t>(stream);
    PinnedArena& a = arena();
    if (a.depth == 0 && a.off) {
      // whatever an earlier call queued from the arena must have landed before its bytes are reused — on the stream THAT call used (the tracking thread
      // alternates between the matcher handle's stream and the optimizer handle's; a call that threw after its uploads left copies in flight there)
      if (a.pending) {
        const hipError_t e = hipStreamSynchronize(a.lastStream);
        if (e != hipSuccess) { current_stream() = saved; hip_check(e, "hipStreamSynchronize"); }   // (depth untouched: the destructor will not run)
      }
      a.pending = false;
      a.rewind();
    }
    ++a.depth;   // only once nothing above can throw
  }
  ~StreamScope() { --arena().depth; current_stream() = saved; }
  StreamScope(const StreamScope&) = delete;
  StreamScope& operator=(const StreamScope&) = delete;
};

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
    if (arena().depth > 0) {   // inside an adapter call: through the thread's pinned arena, no wait (see PinnedArena)
      void* st = arena().take(n * sizeof(T));
      std::memcpy(st, host, n * sizeof(T));
      arena().lastStream = current_stream(); arena().pending = true;
      hip_check(hipMemcpyAsync(p_, st, n * sizeof(T), hipMemcpyHostToDevice, current_stream()), "hipMemcpy H2D");
      return;
    }
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
