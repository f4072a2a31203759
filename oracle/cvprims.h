// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it.
//
// cvprims.h: CPU restatement of the OpenCV primitives the reference's hot path calls.  OpenCV (>=4.4,
// CI pin 4.5.2 — /root/reference/CMakeLists.txt:27, .github/workflows/cmake.yml:40) is a third-party
// dependency that is NOT vendored under /root/reference and is absent from this image, so these follow
// the *published generic C++ algorithms* of OpenCV 4.x (imgproc/resize.cpp, features2d/fast.cpp +
// fast_score.cpp, imgproc/smooth.dispatch.cpp fixed-point Gaussian, core/mathfuncs_core fastAtan2).
// PARITY UNPINNED at this boundary: the reference ships no test or golden vector for this path and no
// OpenCV binary exists here to generate one.  Each primitive sits behind one function so
// an OpenCV-backed check can be swapped in on a machine that has it.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace orc {

// cv::KeyPoint POD (28 bytes): {Point2f pt; float size; float angle; float response; int octave; int class_id;}
struct KeyPoint {
  float x, y, size, angle, response;
  int octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

// A view onto 8-bit single-channel pixels (the cv::Mat ROI idiom: data points at the ROI origin,
// step is the parent's row pitch).
struct Img {
  uint8_t* data = nullptr;
  int cols = 0, rows = 0, step = 0;
  uint8_t* ptr(int y) const { return data + (size_t)y * step; }
  uint8_t& at(int y, int x) const { return data[(size_t)y * step + x]; }
  Img roi(int x0, int y0, int w, int h) const { return Img{data + (size_t)y0 * step + x0, w, h, step}; }
};

// cvRound: round-half-to-even (SSE cvtss2si / lrint under the default rounding mode).
static inline int cvRound(float v) { return (int)lrintf(v); }
static inline int cvRound(double v) { return (int)lrint(v); }
static inline int cvFloor(float v) { int i = (int)v; return i - (i > v); }
static inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
static inline int cvCeil(float v) { int i = (int)v; return i + (i < v); }
static inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }

// cv::borderInterpolate(p, len, BORDER_REFLECT_101)
static inline int reflect101(int p, int len) {
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0) p = -p; else p = 2 * (len - 1) - p;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

// cv::copyMakeBorder(src, dst, b, b, b, b, BORDER_REFLECT_101 [+ISOLATED]); dst is (rows+2b)x(cols+2b);
// src may be the interior ROI of dst (the in-place use at ORBextractor.cc:1104).
void copyMakeBorder101(const Img& src, const Img& dst, int border);

// cv::resize(src, dst, dst.size(), 0, 0, INTER_LINEAR) for CV_8UC1 (generic fixed-point path,
// INTER_RESIZE_COEF_BITS = 11).
void resizeLinear8u(const Img& src, const Img& dst);

// cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) on a continuous CV_8U image:
// OpenCV's bit-exact fixed-point path, 8.8 kernel with error diffusion = {18,34,48,56,48,34,18}/256.
void gaussianBlur7x7s2(const Img& src, const Img& dst);
extern const int kGauss7[7];

// cv::FAST(img, kps, threshold, true) (TYPE_9_16); keypoints row-major, response = cornerScore.
void fast9_16(const Img& img, std::vector<KeyPoint>& kps, int threshold, bool nms);
// max(A,-B): the largest t for which the pixel is still a 9/16 corner at threshold t-1... see .cc
int fastCornerStrength(const uint8_t* p, int step);

// cv::fastAtan2(y, x) in degrees [0,360)
float fastAtan2(float y, float x);

// glibc 2.35 cosf/sinf (sysdeps/ieee754/flt-32/s_{cos,sin}f.c, sincosf.h) restated; verified bit-equal
// to this container's libm for every float in [0, 7] (tools/check_sincosf.c).  The reference calls
// cos(float)/sin(float) (ORBextractor.cc:104-105); restating keeps the oracle independent of the host
// libm and lets the HIP kernel run the identical double-precision sequence.
float cosf_glibc(float y);
float sinf_glibc(float y);

}  // namespace orc
