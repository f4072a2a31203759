// ORACLE — TEST INFRASTRUCTURE ONLY (see cvprims.h).  CPU restatement of OpenCV 4.x primitives.
#include "cvprims.h"

#include <algorithm>
#include <cfloat>

namespace orc {

void copyMakeBorder101(const Img& src, const Img& dst, int b) {
  const int w = src.cols, h = src.rows;
  // interior first (no-op when src is already dst's interior), then left/right of interior rows,
  // then whole top/bottom rows copied from the (already bordered) interior rows — the same result as
  // cv::copyMakeBorder's per-row tab lookup.
  for (int y = 0; y < h; ++y) {
    uint8_t* d = dst.ptr(y + b) + b;
    const uint8_t* s = src.ptr(y);
    if (d != s) memmove(d, s, w);
  }
  for (int y = 0; y < h; ++y) {
    uint8_t* row = dst.ptr(y + b);
    for (int x = 0; x < b; ++x) row[x] = row[b + reflect101(x - b, w)];
    for (int x = 0; x < b; ++x) row[b + w + x] = row[b + reflect101(w + x, w)];
  }
  const int W = w + 2 * b;
  for (int y = 0; y < b; ++y) memcpy(dst.ptr(y), dst.ptr(b + reflect101(y - b, h)), W);
  for (int y = 0; y < b; ++y) memcpy(dst.ptr(b + h + y), dst.ptr(b + reflect101(h + y, h)), W);
}

// imgproc/resize.cpp: resize() coefficient set-up for INTER_LINEAR + HResizeLinear<uchar,int,short,2048>
// + VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>.
void resizeLinear8u(const Img& src, const Img& dst) {
  const int sw = src.cols, sh = src.rows, dw = dst.cols, dh = dst.rows;
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  const int ONE = 2048;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
  for (int dx = 0; dx < dw; ++dx) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cvFloor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx + 1 >= sw) {
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    }
    xofs[dx] = sx;
    float c0 = 1.f - fx, c1 = fx;
    ialpha[dx * 2] = (short)std::min(std::max(cvRound(c0 * ONE), -32768), 32767);
    ialpha[dx * 2 + 1] = (short)std::min(std::max(cvRound(c1 * ONE), -32768), 32767);
  }
  for (int dy = 0; dy < dh; ++dy) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cvFloor(fy);
    fy -= sy;
    yofs[dy] = sy;
    float c0 = 1.f - fy, c1 = fy;
    ibeta[dy * 2] = (short)std::min(std::max(cvRound(c0 * ONE), -32768), 32767);
    ibeta[dy * 2 + 1] = (short)std::min(std::max(cvRound(c1 * ONE), -32768), 32767);
  }
  std::vector<int> r0(dw), r1(dw);
  auto hresize = [&](int sy, std::vector<int>& out) {
    sy = sy < 0 ? 0 : (sy < sh ? sy : sh - 1);  // clip(sy, 0, ssize.height)
    const uint8_t* S = src.ptr(sy);
    for (int dx = 0; dx < dw; ++dx) {
      int sx = xofs[dx];
      int sx1 = sx + 1 < sw ? sx + 1 : sx;  // alpha[1]==0 there (the "dx >= xmax" branch: S[sx]*ONE)
      out[dx] = S[sx] * ialpha[dx * 2] + S[sx1] * ialpha[dx * 2 + 1];
    }
  };
  for (int dy = 0; dy < dh; ++dy) {
    hresize(yofs[dy], r0);
    hresize(yofs[dy] + 1, r1);
    const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    uint8_t* D = dst.ptr(dy);
    for (int x = 0; x < dw; ++x)
      D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// getGaussianKernelFixedPoint_ED(n=7, sigma=2, 8 fractional bits): 17.96->18, 33.52->34, 48.34->48,
// centre = 256 - 2*(18+34+48) = 56.
const int kGauss7[7] = {18, 34, 48, 56, 48, 34, 18};

// fixedSmoothInvoker<uint8_t, ufixedpoint16>: row pass u8 x 8.8 -> 8.8 (exact, <= 65280), column pass
// 8.8 x 8.8 -> 16.16, rounded to u8 with +0.5 (ufixedpoint32 -> uint8_t cast adds 1<<15, shifts 16).
void gaussianBlur7x7s2(const Img& src, const Img& dst) {
  const int w = src.cols, h = src.rows;
  std::vector<uint16_t> rows((size_t)w * h);
  for (int y = 0; y < h; ++y) {
    const uint8_t* S = src.ptr(y);
    for (int x = 0; x < w; ++x) {
      uint32_t acc = 0;
      for (int k = -3; k <= 3; ++k) acc += (uint32_t)kGauss7[k + 3] * S[reflect101(x + k, w)];
      rows[(size_t)y * w + x] = (uint16_t)acc;
    }
  }
  for (int y = 0; y < h; ++y) {
    uint8_t* D = dst.ptr(y);
    for (int x = 0; x < w; ++x) {
      uint32_t acc = 0;
      for (int k = -3; k <= 3; ++k) acc += (uint32_t)kGauss7[k + 3] * rows[(size_t)reflect101(y + k, h) * w + x];
      D[x] = (uint8_t)((acc + 32768u) >> 16);
    }
  }
}

static const int kFastOfs[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                                    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// fast_score.cpp cornerScore<16>(ptr, pixel, threshold) returns max(threshold, A, C) - 1 with
// A = max over the 16 circular 9-arcs of min(v - p_k), C = the same for (p_k - v).  A pixel is a corner
// at threshold t iff max(A, C) > t, so for every detected corner the score is max(A, C) - 1, independent
// of t.  This returns max(A, C) (may be <= 0 for flat pixels).
int fastCornerStrength(const uint8_t* p, int step) {
  int d[25];
  const int v = p[0];
  for (int k = 0; k < 25; ++k) d[k] = v - p[kFastOfs[k & 15][0] + kFastOfs[k & 15][1] * step];
  int best = -256;
  for (int s = 0; s < 16; ++s) {
    int mn = d[s], mx = d[s];
    for (int j = 1; j < 9; ++j) { mn = std::min(mn, d[s + j]); mx = std::max(mx, d[s + j]); }
    best = std::max(best, std::max(mn, -mx));
  }
  return best;
}

// features2d/fast.cpp FAST_t<16>: rows 3..rows-3, cols 3..cols-3; score buffer of three rows, zero where
// not a corner; a corner survives NMS iff its score is strictly greater than all 8 neighbours.
void fast9_16(const Img& img, std::vector<KeyPoint>& kps, int threshold, bool nms) {
  kps.clear();
  const int W = img.cols, H = img.rows;
  if (W < 7 || H < 7) return;
  threshold = std::min(std::max(threshold, 0), 255);
  std::vector<uint8_t> score((size_t)W * H, 0);
  std::vector<uint8_t> corner((size_t)W * H, 0);
  for (int y = 3; y < H - 3; ++y)
    for (int x = 3; x < W - 3; ++x) {
      const uint8_t* p = img.ptr(y) + x;
      // definition used by the detector: >= 9 contiguous ring pixels all < v - t or all > v + t
      const int v = p[0];
      {  // cv::FAST's high-speed test (same result, keeps the CPU baseline honest): any 9-arc contains ring pixel
         // 0 or 8, so both must not be "similar"
        const int q0 = p[kFastOfs[0][0] + kFastOfs[0][1] * img.step], q8 = p[kFastOfs[8][0] + kFastOfs[8][1] * img.step];
        const bool d = (q0 < v - threshold) || (q8 < v - threshold), b = (q0 > v + threshold) || (q8 > v + threshold);
        if (!d && !b) continue;
        const int q4 = p[kFastOfs[4][0] + kFastOfs[4][1] * img.step], q12 = p[kFastOfs[12][0] + kFastOfs[12][1] * img.step];
        const bool d2 = d && ((q4 < v - threshold) || (q12 < v - threshold)), b2 = b && ((q4 > v + threshold) || (q12 > v + threshold));
        if (!d2 && !b2) continue;
      }
      int runD = 0, runB = 0;
      bool is = false;
      for (int k = 0; k < 25 && !is; ++k) {
        int q = p[kFastOfs[k & 15][0] + kFastOfs[k & 15][1] * img.step];
        runD = (q < v - threshold) ? runD + 1 : 0;
        runB = (q > v + threshold) ? runB + 1 : 0;
        is = runD > 8 || runB > 8;
      }
      if (is) {
        corner[(size_t)y * W + x] = 1;
        int s = std::max(threshold, fastCornerStrength(p, img.step)) - 1;  // cornerScore<16>
        score[(size_t)y * W + x] = (uint8_t)s;
      }
    }
  for (int y = 3; y < H - 3; ++y)
    for (int x = 3; x < W - 3; ++x) {
      if (!corner[(size_t)y * W + x]) continue;
      const int s = score[(size_t)y * W + x];
      bool keep = true;
      if (nms) {
        for (int dy = -1; dy <= 1 && keep; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            if (!dx && !dy) continue;
            if (!(s > score[(size_t)(y + dy) * W + x + dx])) { keep = false; break; }
          }
      }
      if (keep) kps.push_back(KeyPoint{(float)x, (float)y, 7.f, -1.f, (float)s, 0, -1});
    }
}

// core/mathfuncs_core.simd.hpp atan_f32 (single precision throughout).
float fastAtan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  static const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  static const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  static const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ---- glibc 2.35 sincosf ----------------------------------------------------------------------------
namespace {
struct sincos_t { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
const sincos_t kSC[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2,
     0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2,
     -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
inline uint32_t asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline uint32_t abstop12(float x) { return (asuint(x) >> 20) & 0x7ff; }
inline double sinf_poly(double x, double x2, const sincos_t* p, int n) {
  if ((n & 1) == 0) {
    double x3 = x * x2, s1 = p->s2 + x2 * p->s3, x7 = x3 * x2, s = x + x3 * p->s1;
    return s + x7 * s1;
  }
  double x4 = x2 * x2, c2 = p->c3 + x2 * p->c4, c1 = p->c0 + x2 * p->c1, x6 = x4 * x2, c = c1 + x4 * p->c2;
  return c + x6 * c2;
}
inline double reduce_fast(double x, const sincos_t* p, int* np) {
  double r = x * p->hpi_inv;
  int n = ((int32_t)r + 0x800000) >> 24;
  *np = n;
  return x - n * p->hpi;
}
}  // namespace

// Valid for |y| < 120 (the extractor only passes [0, 2*pi]).
float cosf_glibc(float y) {
  double x = y;
  int n;
  const sincos_t* p = &kSC[0];
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    double x2 = x * x;
    if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
    return (float)sinf_poly(x, x2, p, 1);
  }
  x = reduce_fast(x, p, &n);
  double s = p->sign[n & 3];
  if (n & 2) p = &kSC[1];
  return (float)sinf_poly(x * s, x * x, p, n ^ 1);
}
float sinf_glibc(float y) {
  double x = y;
  int n;
  const sincos_t* p = &kSC[0];
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
    double s = x * x;
    if (abstop12(y) < abstop12(0x1p-12f)) return y;
    return (float)sinf_poly(x, s, p, 0);
  }
  x = reduce_fast(x, p, &n);
  double s = p->sign[n & 3];
  if (n & 2) p = &kSC[1];
  return (float)sinf_poly(x * s, x * x, p, n);
}

}  // namespace orc

// ---- cv::undistortPoints(src, K, distCoeffs, R = I, P) — OpenCV 4.x calib3d cvUndistortPointsInternal, the 6-argument overload's
// TermCriteria(MAX_ITER, 5, 0.01): FP64, five fixed-point iterations, k = (k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4) with everything
// past k3 zero for the reference's 4- or 5-entry mDistCoef.  Used by Frame::UndistortKeyPoints / ComputeImageBounds
// (reference src/Frame.cc:829-887).  PARITY UNPINNED (OpenCV is not vendored).
extern "C" void orc_undistort_points(int n, const float* xy, float fx, float fy, float cx, float cy, const float* dist5,
                                     float pfx, float pfy, float pcx, float pcy, float* out) {
  double k[12] = {0};
  k[0] = dist5[0]; k[1] = dist5[1]; k[2] = dist5[2]; k[3] = dist5[3]; k[4] = dist5[4];
  const double dfx = fx, dfy = fy, dcx = cx, dcy = cy, ifx = 1. / dfx, ify = 1. / dfy;
  for (int i = 0; i < n; ++i) {
    const double u = xy[2 * i], v = xy[2 * i + 1];
    double x = (u - dcx) * ifx, y = (v - dcy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; ++j) {
      const double r2 = x * x + y * y;
      const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      if (icdist < 0) { x = (u - dcx) * ifx; y = (v - dcy) * ify; break; }
      const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
      const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    const double RR[9] = {(double)pfx, 0, (double)pcx, 0, (double)pfy, (double)pcy, 0, 0, 1};
    const double xx = RR[0] * x + RR[1] * y + RR[2], yy = RR[3] * x + RR[4] * y + RR[5], ww = 1. / (RR[6] * x + RR[7] * y + RR[8]);
    out[2 * i] = (float)(xx * ww);
    out[2 * i + 1] = (float)(yy * ww);
  }
}

// Frame::ComputeStereoFromRGBD (reference src/Frame.cc:1049-1067); kp / kpUn = (x, y) pairs
extern "C" void orc_stereo_from_rgbd(int n, const float* kp, const float* kpUn, const float* depth, int W, int H, float bf,
                                     float* uRight, float* depthOut) {
  for (int i = 0; i < n; ++i) {
    uRight[i] = -1; depthOut[i] = -1;
    const int v = (int)kp[2 * i + 1], u = (int)kp[2 * i];   // imDepth.at<float>(float v, float u)
    if (u < 0 || u >= W || v < 0 || v >= H) continue;        // (out-of-image reads are undefined in the reference)
    const float d = depth[(size_t)v * W + u];
    if (d > 0) { depthOut[i] = d; uRight[i] = kpUn[2 * i] - bf / d; }
  }
}
