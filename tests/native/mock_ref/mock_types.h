// MOCKS — NOT the reference, NOT OpenCV / Eigen / Sophus / DBoW2.  Declarations, with just enough canned behaviour to carry test data, of
// exactly the names the reference-typed members of include/morb/ORBmatcher.h / Optimizer.h touch (member names and types read off
// /root/reference/include/*.h).  Two uses: tests/native/call_sites_check.cc (the call expressions of src/Tracking.cc / src/LocalMapping.cc
// pasted verbatim must compile, CPU suite) and tests/native/reference_members_check.cc (mock objects filled from test files drive the
// members on the GPU box; the results must equal those of the view-taking adapters, which are compared with the oracle).
// "Canned": an SE3f carries the arrays the test wrote (rotation, translation, camera centre, quaternion) and returns them; no real algebra.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
#include <vector>

namespace Eigen {
struct Vector3f {
  float v[3];
  Vector3f() : v{0, 0, 0} {}
  Vector3f(float a, float b, float c) : v{a, b, c} {}
  float operator()(int i) const { return v[i]; }
  float& operator()(int i) { return v[i]; }
  Vector3f operator/(float s) const { return Vector3f(v[0] / s, v[1] / s, v[2] / s); }
};
struct Vector2f { float v[2] = {0, 0}; float operator()(int i) const { return v[i]; } };
struct Matrix3f {
  float m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  const void* canned = nullptr;   // (mock: the SE3f this rotation was taken from)
  float operator()(int r, int c) const { return m[3 * r + c]; }
  float& operator()(int r, int c) { return m[3 * r + c]; }
};
struct Quaternionf {
  float q[4];
  Quaternionf() : q{0, 0, 0, 1} {}
  Quaternionf(float w, float x, float y, float z) : q{x, y, z, w} {}
  float x() const { return q[0]; } float y() const { return q[1]; } float z() const { return q[2]; } float w() const { return q[3]; }
};
template <int R, int C> struct MatD { double m[R * C] = {0}; double operator()(int r, int c) const { return m[C * r + c]; } double& operator()(int r, int c) { return m[C * r + c]; } };
template <int R> struct VecD { double v[R] = {0}; double operator()(int i) const { return v[i]; } double& operator()(int i) { return v[i]; } };
typedef MatD<3, 3> Matrix3d;
typedef VecD<3> Vector3d;
template <int N> struct VecF { float v[N] = {0}; float operator()(int i) const { return v[i]; } float& operator()(int i) { return v[i]; } };
template <int R, int C> struct MatF { float m[R * C] = {0}; float operator()(int r, int c) const { return m[C * r + c]; } float& operator()(int r, int c) { return m[C * r + c]; } };
template <int N> struct DiagF { VecF<N> d; const VecF<N>& diagonal() const { return d; } };
}  // namespace Eigen
typedef Eigen::MatD<15, 15> Matrix15d;

namespace Sophus {
struct SE3f {
  using QuaternionType = Eigen::Quaternionf;
  using Point = Eigen::Vector3f;
  float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0}, Ow[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1};
  SE3f() {}
  SE3f(const Eigen::Quaternionf& qq, const Eigen::Vector3f& tt) { for (int k = 0; k < 4; ++k) q[k] = qq.q[k]; for (int k = 0; k < 3; ++k) t[k] = tt.v[k]; }
  SE3f(const Eigen::Matrix3f& RR, const Eigen::Vector3f& tt) {
    if (RR.canned) { *this = *static_cast<const SE3f*>(RR.canned); return; }
    for (int k = 0; k < 9; ++k) R[k] = RR.m[k];
    for (int k = 0; k < 3; ++k) t[k] = tt.v[k];
  }
  Eigen::Matrix3f rotationMatrix() const { Eigen::Matrix3f M; for (int k = 0; k < 9; ++k) M.m[k] = R[k]; M.canned = this; return M; }
  Eigen::Vector3f translation() const { return Eigen::Vector3f(t[0], t[1], t[2]); }
  Eigen::Quaternionf unit_quaternion() const { return Eigen::Quaternionf(q[3], q[0], q[1], q[2]); }
  // (the translation of the inverse is carried, not computed: Ow as the test wrote it)
  SE3f inverse() const { SE3f o; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) o.R[3 * r + c] = R[3 * c + r]; for (int k = 0; k < 3; ++k) { o.t[k] = Ow[k]; o.Ow[k] = t[k]; } return o; }
  SE3f operator*(const SE3f& b) const {   // R, t composed in float, left to right (q / Ow stay the left factor's)
    SE3f o = *this;
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) o.R[3 * r + c] = R[3 * r] * b.R[c] + R[3 * r + 1] * b.R[3 + c] + R[3 * r + 2] * b.R[6 + c];
      o.t[r] = R[3 * r] * b.t[0] + R[3 * r + 1] * b.t[1] + R[3 * r + 2] * b.t[2] + t[r];
    }
    return o;
  }
  Eigen::Vector3f operator*(const Eigen::Vector3f& x) const { return x; }
};
struct RxSO3f { float q[4] = {0, 0, 0, 1}; Eigen::Quaternionf quaternion() const { return Eigen::Quaternionf(q[3], q[0], q[1], q[2]); } };
struct Sim3f {
  SE3f T;                                   // (canned) SE3(rotationMatrix(), translation() / scale())
  float raw[7] = {0, 0, 0, 1, 0, 0, 0};     // RxSO3 quaternion xyzw, translation
  const Sim3f* inv = nullptr;
  Eigen::Matrix3f rotationMatrix() const { return T.rotationMatrix(); }
  Eigen::Vector3f translation() const { return Eigen::Vector3f(raw[4], raw[5], raw[6]); }
  float scale() const { return 1.f; }
  RxSO3f rxso3() const { RxSO3f r; for (int k = 0; k < 4; ++k) r.q[k] = raw[k]; return r; }
  Sim3f inverse() const { return inv ? *inv : *this; }
};
template <class S> using Sim3 = Sim3f;
template <class S> using SE3 = SE3f;
}  // namespace Sophus

namespace cv {
struct Point2f { float x = 0, y = 0; };
struct KeyPoint { Point2f pt; float size = 0, angle = 0, response = 0; int octave = 0, class_id = -1; };
struct Mat {
  std::vector<uint8_t> data; int rows = 0, cols = 32;
  template <typename T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data.data() + (size_t)r * cols); }
  template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data.data() + (size_t)r * cols); }
};
}  // namespace cv
namespace DBoW2 { typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector; }

namespace ORB_SLAM3 {
struct GeometricCamera {
  std::vector<float> mvParameters; Eigen::Vector2f ep;
  float getParameter(const int i) { return mvParameters[i]; }
  size_t size() { return mvParameters.size(); }
  Eigen::Vector2f project(const Eigen::Vector3f&) { return ep; }   // (canned)
};
namespace IMU {
struct Bias {
  float bax = 0, bay = 0, baz = 0, bwx = 0, bwy = 0, bwz = 0;
  Bias() {}
  Bias(float ax, float ay, float az, float wx, float wy, float wz) : bax(ax), bay(ay), baz(az), bwx(wx), bwy(wy), bwz(wz) {}
};
struct Calib { Sophus::SE3f mTbc; };
struct Preintegrated {
  float dT = 0; Eigen::MatF<15, 15> C; Eigen::DiagF<6> Nga, NgaWalk; Bias b; Eigen::Matrix3f dR; Eigen::Vector3f dV, dP;
  Eigen::Matrix3f JRg, JVg, JVa, JPg, JPa; Eigen::Vector3f avgA, avgW;
  void SetNewBias(const Bias&) {}
};
}  // namespace IMU
struct ConstraintPoseImu {
  ConstraintPoseImu(const Eigen::Matrix3d& R, const Eigen::Vector3d& t, const Eigen::Vector3d& v, const Eigen::Vector3d& g, const Eigen::Vector3d& a, const Matrix15d& h)
      : Rwb(R), twb(t), vwb(v), bg(g), ba(a), H(h) {}
  Eigen::Matrix3d Rwb; Eigen::Vector3d twb, vwb, bg, ba; Matrix15d H;
};
}  // namespace ORB_SLAM3
