// Host instantiation of morb_slam_amd/csrc/quadtree.h for CPU tests only (the product uses the device
// instantiation inside the HIP kernels).  Also exposes libstdc++ std::sort with the compareNodes ordering so
// the introsort emulation can be checked element for element.
#include <algorithm>
#include <cstdint>
#include <vector>

#define QT_HOST_FAST_FORWARD 1   // also build the serial restatement of round 6's fast forward (quadtree.h)
#include "../../morb_slam_amd/csrc/quadtree.h"

extern "C" {

static int distribute(const uint32_t* keys_in, int n, int width, int height, int N, uint32_t* out, int outCap, bool ff) {
  using namespace morbqt;
  const int nIni = (int)roundf((float)width / (float)height);
  const int nodeCap = qt_node_cap(N, nIni), listCap = qt_list_cap(nodeCap);
  std::vector<uint32_t> keys(keys_in, keys_in + n), tmp(n + 1);
  std::vector<Node> nodes(nodeCap);
  std::vector<uint16_t> freeIds(nodeCap), list(listCap);
  std::vector<uint64_t> vA(nodeCap), vB(nodeCap);
  Work w{keys.data(), tmp.data(), nodes.data(), freeIds.data(), list.data(), vA.data(), vB.data(), nullptr, nullptr, nullptr, nodeCap, listCap};
  w.hostFastForward = ff;
  return qt_distribute(w, (uint32_t)n, width, height, N, out, outCap);
}
// the sweeps one by one (the reference's own order of events) / the full sweeps built at once, then the same largest-first phase
int qt_host_distribute(const uint32_t* keys_in, int n, int width, int height, int N, uint32_t* out, int outCap) { return distribute(keys_in, n, width, height, N, out, outCap, false); }
int qt_host_distribute_ff(const uint32_t* keys_in, int n, int width, int height, int N, uint32_t* out, int outCap) { return distribute(keys_in, n, width, height, N, out, outCap, true); }

void qt_host_sort(uint64_t* v, int n) { morbqt::qt_std_sort(v, n); }

void qt_ref_std_sort(uint64_t* v, int n) {
  std::sort(v, v + n, [](uint64_t& a, uint64_t& b) { return (a >> 16) < (b >> 16); });
}
}
