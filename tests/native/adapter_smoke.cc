// Compiles the drop-in C++ adapter (include/morb/ORBextractor.h) against libmorb_hip.so and runs one
// extraction: the C++ side of the boundary a reference maintainer would link.
#include <cstdio>
#include <vector>

#include "../../include/morb/ORBextractor.h"

int main() {
  using namespace ORB_SLAM3;
  podcv::Mat8u img;
  img.rows = 240; img.cols = 320; img.step = 320;
  img.data.resize(320 * 240);
  unsigned s = 12345;
  for (auto& p : img.data) { s = s * 1664525u + 1013904223u; p = (uint8_t)(s >> 24); }
  for (int y = 60; y < 180; ++y) for (int x = 80; x < 240; ++x) img.data[y * 320 + x] = (uint8_t)(((x / 16 + y / 16) & 1) * 200 + 20);
  ORBextractor ext(300, 1.2f, 4, 20, 7);
  std::vector<podcv::KeyPoint> k;
  std::vector<uint8_t> d;
  std::vector<int> lap = {0, 0};
  int mono = ext(img, k, d, lap);
  std::printf("adapter smoke: monoIndex=%d keypoints=%zu levels=%d\n", mono, k.size(), ext.GetLevels());
  podcv::Mat8u empty;
  if (ext(empty, k, d, lap) != -1) return 2;
  return (mono >= 0 && mono == (int)k.size() && !d.empty()) ? 0 : 1;
}
