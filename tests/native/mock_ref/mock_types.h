// SYNTAX-CHECK MOCKS — NOT the reference, NOT OpenCV / Eigen / Sophus / DBoW2.  Declarations (no behaviour) of exactly the
// names include/morb/reference_glue.h uses, so that the glue — which can only be compiled for real inside the reference tree —
// is at least parsed and type-checked by tests/test_oracle_cpu.py::test_reference_glue_parses.  Nothing links against this.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace Eigen {
struct Vector3f { float v[3]; Vector3f() : v{0, 0, 0} {} Vector3f(float a, float b, float c) : v{a, b, c} {} float operator()(int i) const { return v[i]; } };
struct Vector2f { float v[2]; float operator()(int i) const { return v[i]; } };
struct Matrix3f { float m[9]; float operator()(int r, int c) const { return m[3 * r + c]; } };
struct Quaternionf { float q[4]; Quaternionf() : q{0, 0, 0, 1} {} Quaternionf(float w, float x, float y, float z) : q{x, y, z, w} {} float x() const { return q[0]; } float y() const { return q[1]; } float z() const { return q[2]; } float w() const { return q[3]; } };
}  // namespace Eigen
namespace Sophus {
struct SE3f {
  SE3f() {}
  SE3f(const Eigen::Quaternionf&, const Eigen::Vector3f&) {}
  Eigen::Matrix3f rotationMatrix() const { return {}; }
  Eigen::Vector3f translation() const { return {}; }
  Eigen::Quaternionf unit_quaternion() const { return {}; }
  SE3f inverse() const { return {}; }
  SE3f operator*(const SE3f&) const { return {}; }
  Eigen::Vector3f operator*(const Eigen::Vector3f&) const { return {}; }
};
}  // namespace Sophus
namespace cv {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct Mat { template <typename T> T* ptr(int) const { return nullptr; } };
}  // namespace cv
namespace DBoW2 { typedef std::map<unsigned int, std::vector<unsigned int>> FeatureVector; }
