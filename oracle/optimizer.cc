// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path.
//
// CPU restatement of Optimizer::PoseOptimization (/root/reference/src/Optimizer.cc:762-1051) and
// Optimizer::LocalBundleAdjustment (:1053-1441) together with the parts of g2o they run through:
//   LM loop + ORB-SLAM stop rule   Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-194
//   outer loop                     core/sparse_optimizer.cpp:354-420
//   buildSystem / setLambda / Schur solve / back-substitution   core/block_solver.hpp:354-590
//   quadratic forms                core/base_unary_edge.hpp:43-72, base_binary_edge.hpp:55-113, base_edge.h:58-102
//   Huber kernel                   core/robust_kernel_impl.cpp:65-91
//   SE3Quat exp / product / map    types/se3quat.h:99-281, VertexSE3Expmap::oplusImpl types_six_dof_expmap.h:73-76
//   edges                          src/OptimizableTypes.cpp:49-209, include/OptimizableTypes.h:34-164,
//                                  types/types_six_dof_expmap.cpp:190-197,228-270,339-403, Pinhole.cpp:38-83
// Eigen (not vendored, absent here) supplies quaternion <-> matrix conversions, 3x3 inverse and the LDLT
// factorisations; they are restated with plain double arithmetic (published Eigen formulas).  All of this is
// FP64 like g2o; the reference's float leaks (stereo cam_project's `const float invz`, float camera
// parameters, float Huber deltas, float chi2 comparisons) are kept.  Pinhole mono and rectified-stereo edges
// are covered; the fisheye "ToBody" edges (KannalaBrandt8) are not yet.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "matcher.h"
#include "orb_oracle.h"

double* g_orcTrace = nullptr; int g_orcTraceCap = 0, g_orcTraceN = 0;   // developer hook: per LM trial {currentChi, tempChi, lambda, rho, scale, ok}
namespace orc {
namespace ba {

struct SE3Quat {
  double q[4];  // x y z w
  double t[3];
};

static void normalizeRotation(SE3Quat& s) {  // se3quat.h:276-281
  if (s.q[3] < 0) for (double& c : s.q) c *= -1;
  const double n = std::sqrt(s.q[0] * s.q[0] + s.q[1] * s.q[1] + s.q[2] * s.q[2] + s.q[3] * s.q[3]);
  for (double& c : s.q) c /= n;
}
static void rotate(const double q[4], const double v[3], double out[3]) {  // Eigen QuaternionBase::_transformVector
  const double ux = q[0], uy = q[1], uz = q[2], w = q[3];
  double uv[3] = {uy * v[2] - uz * v[1], uz * v[0] - ux * v[2], ux * v[1] - uy * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  out[0] = v[0] + w * uv[0] + (uy * uv[2] - uz * uv[1]);
  out[1] = v[1] + w * uv[1] + (uz * uv[0] - ux * uv[2]);
  out[2] = v[2] + w * uv[2] + (ux * uv[1] - uy * uv[0]);
}
static void toRotationMatrix(const double q[4], double R[9]) {  // Eigen QuaternionBase::toRotationMatrix
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void fromRotationMatrix(const double m[9], double q[4]) {  // Eigen quaternionbase_assign_impl<.,3,3>
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
    q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
  }
}
static SE3Quat mul(const SE3Quat& a, const SE3Quat& b) {  // se3quat.h:99-105
  SE3Quat r = a;
  double rt[3];
  rotate(a.q, b.t, rt);
  for (int i = 0; i < 3; ++i) r.t[i] += rt[i];
  const double* p = a.q; const double* o = b.q;
  r.q[3] = p[3] * o[3] - p[0] * o[0] - p[1] * o[1] - p[2] * o[2];
  r.q[0] = p[3] * o[0] + p[0] * o[3] + p[1] * o[2] - p[2] * o[1];
  r.q[1] = p[3] * o[1] + p[1] * o[3] + p[2] * o[0] - p[0] * o[2];
  r.q[2] = p[3] * o[2] + p[2] * o[3] + p[0] * o[1] - p[1] * o[0];
  normalizeRotation(r);
  return r;
}
static void mapPoint(const SE3Quat& T, const double x[3], double out[3]) {  // se3quat.h:212-215
  rotate(T.q, x, out);
  for (int i = 0; i < 3; ++i) out[i] += T.t[i];
}
static void mat3mul(const double A[9], const double B[9], double C[9]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
#ifdef XP_TRIG
static inline double xp_sin(double x) { const double xx = x * x; return x * (1.0 + xx * (-1.0 / 6 + xx * (1.0 / 120 + xx * (-1.0 / 5040 + xx * (1.0 / 362880 + xx * (-1.0 / 39916800)))))); }
static inline double xp_cos(double x) { const double xx = x * x; return 1.0 + xx * (-0.5 + xx * (1.0 / 24 + xx * (-1.0 / 720 + xx * (1.0 / 40320 + xx * (-1.0 / 3628800 + xx * (1.0 / 479001600)))))); }
#define XSIN xp_sin
#define XCOS xp_cos
#define XPOW3(t) ((t) * (t) * (t))
#else
#define XSIN std::sin
#define XCOS std::cos
#define XPOW3(t) std::pow(t, 3)
#endif
static SE3Quat expSE3(const double u[6]) {  // se3quat.h:219-250
  const double omega[3] = {u[0], u[1], u[2]}, upsilon[3] = {u[3], u[4], u[5]};
  const double theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  const double Om[9] = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};
  double Om2[9];
  mat3mul(Om, Om, Om2);
  double R[9], V[9];
  if (theta < 0.00001) {
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + Om[i] + Om2[i];
    memcpy(V, R, sizeof R);
  } else {
    const double a = XSIN(theta) / theta, b = (1 - XCOS(theta)) / (theta * theta);
    const double c = (theta - XSIN(theta)) / XPOW3(theta);
    for (int i = 0; i < 9; ++i) {
      R[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * Om[i] + b * Om2[i];
      V[i] = (i % 4 == 0 ? 1.0 : 0.0) + b * Om[i] + c * Om2[i];
    }
  }
  SE3Quat r;
  fromRotationMatrix(R, r.q);
  for (int i = 0; i < 3; ++i) r.t[i] = V[i * 3] * upsilon[0] + V[i * 3 + 1] * upsilon[1] + V[i * 3 + 2] * upsilon[2];
  normalizeRotation(r);  // SE3Quat(const Quaterniond&, const Vector3d&) normalises
  return r;
}
static SE3Quat fromFloatPose(const float p[7]) {  // g2o::SE3Quat(Tcw.unit_quaternion().cast<double>(), t.cast<double>())
  SE3Quat s;
  for (int i = 0; i < 4; ++i) s.q[i] = (double)p[i];
  for (int i = 0; i < 3; ++i) s.t[i] = (double)p[4 + i];
  normalizeRotation(s);
  return s;
}

enum { KIND_MONO = 0, KIND_STEREO = 1, KIND_KB8_LEFT = 2, KIND_KB8_RIGHT = 3 };  // 2/3: fisheye rig (left camera / right camera 'ToBody')

struct Edge {
  int kind, pose, point;
  double obs[3];
  double info;        // invSigma2 (information = info * I)
  double delta;       // Huber delta; <= 0: no robust kernel
  int level;
  double err[3];      // _error as last computed
};

struct Camera { float fx, fy, cx, cy, bf; };

struct Graph {
  std::vector<SE3Quat> poses, poseBackup;
  std::vector<char> poseFixed;
  std::vector<std::vector<double>> dummy;
  std::vector<double> points, pointBackup;  // 3 per point
  std::vector<char> pointFixed;             // unary edges = binary edges to a fixed point
  std::vector<Edge> edges;
  Camera cam;
  orc::kb8::Cam kbL, kbR;   // fisheye rig cameras (KIND_KB8_*)
  SE3Quat Trl;              // mTrl: left-camera frame -> right-camera frame
  bool unaryForm = false;  // use the "...OnlyPose" Jacobian formulas (types_six_dof_expmap.cpp:375-403)
  // solver state
  std::vector<int> poseCol, pointCol, active;
  int nFreePoses = 0, nFreePoints = 0;
  std::vector<double> Hpp, Hll, Hpl, b, x;  // Hpp: dense (6P)^2 reduced-size storage for the pose part
  std::vector<int> hplPose, hplPoint;       // per active edge: block owner
  double lambda = 0, ni = 2;
  int nBad = 0, levenbergIterations = 0;
  double userLambdaInit = 0;
  volatile const int* forceStop = nullptr;

  int dim(const Edge& e) const { return e.kind == KIND_STEREO ? 3 : 2; }

  void project(const Edge& e, const double xc[3], double out[3]) const {
    if (e.kind == KIND_KB8_LEFT || e.kind == KIND_KB8_RIGHT) {  // KannalaBrandt8::project(Vector3d)
      orc::kb8::projectD(e.kind == KIND_KB8_LEFT ? kbL : kbR, xc, out);
      out[2] = 0;
    } else if (e.kind == KIND_MONO) {  // Pinhole::project(Vector3d), Pinhole.cpp:38-44 (float parameters in double math)
      out[0] = (double)cam.fx * xc[0] / xc[2] + (double)cam.cx;
      out[1] = (double)cam.fy * xc[1] / xc[2] + (double)cam.cy;
      out[2] = 0;
    } else {  // cam_project, types_six_dof_expmap.cpp:190-197 / :339-346: const float invz
      const float invz = (float)(1.0f / xc[2]);
      out[0] = xc[0] * invz * (double)cam.fx + (double)cam.cx;
      out[1] = xc[1] * invz * (double)cam.fy + (double)cam.cy;
      out[2] = out[0] - (double)cam.bf * invz;
    }
  }
  void computeError(Edge& e) const {
    double xc[3], pr[3];
    if (e.kind == KIND_KB8_RIGHT) mapPoint(mul(Trl, poses[e.pose]), &points[3 * e.point], xc);  // (mTrl * T).map(Xw)
    else mapPoint(poses[e.pose], &points[3 * e.point], xc);
    project(e, xc, pr);
    for (int i = 0; i < dim(e); ++i) e.err[i] = e.obs[i] - pr[i];
  }
  double chi2(const Edge& e) const {  // base_edge.h:58-61
    double s = 0;
    for (int i = 0; i < dim(e); ++i) s += e.err[i] * (e.info * e.err[i]);
    return s;
  }
  static void robustify(double delta, double e, double rho[3]) {  // robust_kernel_impl.cpp:78-91
    const double dsqr = delta * delta;
    if (e <= dsqr) { rho[0] = e; rho[1] = 1.; rho[2] = 0.; }
    else {
      const double sqrte = std::sqrt(e);
      rho[0] = 2 * sqrte * delta - dsqr;
      rho[1] = delta / sqrte;
      rho[2] = -0.5 * rho[1] / e;
    }
  }
  void computeActiveErrors() { for (int k : active) computeError(edges[k]); }
  double activeRobustChi2() const {
    double s = 0;
    for (int k : active) {
      const Edge& e = edges[k];
      double c = chi2(e);
      if (e.delta > 0) { double rho[3]; robustify(e.delta, c, rho); c = rho[0]; }
      s += c;
    }
    return s;
  }
  // Jacobians: Jp (d x 6, pose), Jl (d x 3, point)
  void linearize(const Edge& e, double Jp[18], double Jl[9]) const {
    double xc[3];
    const SE3Quat& T = poses[e.pose];
    mapPoint(T, &points[3 * e.point], xc);
    const double x = xc[0], y = xc[1], z = xc[2];
    double R[9];
    toRotationMatrix(T.q, R);
    if (e.kind == KIND_KB8_LEFT || e.kind == KIND_KB8_RIGHT) {
      // OptimizableTypes.cpp:49-62 (left) / :88-104 (right: -projectJac(X_r) * R_rl * SE3deriv(X_l))
      const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
      double pj[6], M[9];
      if (e.kind == KIND_KB8_LEFT) {
        orc::kb8::projectJac(kbL, xc, pj);
        for (int i = 0; i < 9; ++i) M[i] = (i % 4 == 0) ? 1.0 : 0.0;
      } else {
        double xr[3];
        mapPoint(Trl, xc, xr);
        orc::kb8::projectJac(kbR, xr, pj);
        toRotationMatrix(Trl.q, M);
      }
      double pjM[6];
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c) pjM[r * 3 + c] = pj[r * 3] * M[c] + pj[r * 3 + 1] * M[3 + c] + pj[r * 3 + 2] * M[6 + c];
      for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 6; ++c) Jp[r * 6 + c] = -(pjM[r * 3] * D[c] + pjM[r * 3 + 1] * D[6 + c] + pjM[r * 3 + 2] * D[12 + c]);
        for (int c = 0; c < 3; ++c) Jl[r * 3 + c] = -(pjM[r * 3] * R[c] + pjM[r * 3 + 1] * R[3 + c] + pjM[r * 3 + 2] * R[6 + c]);
      }
    } else if (e.kind == KIND_MONO) {
      // -projectJac * SE3deriv and -projectJac * R  (OptimizableTypes.cpp:49-62, :134-156; Pinhole.cpp:73-83)
      const double fx = cam.fx, fy = cam.fy;
      const double pj[6] = {fx / z, 0.0, -fx * x / (z * z), 0.0, fy / z, -fy * y / (z * z)};
      const double D[18] = {0, z, -y, 1, 0, 0, -z, 0, x, 0, 1, 0, y, -x, 0, 0, 0, 1};
      for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 6; ++c) Jp[r * 6 + c] = -(pj[r * 3] * D[c] + pj[r * 3 + 1] * D[6 + c] + pj[r * 3 + 2] * D[12 + c]);
        for (int c = 0; c < 3; ++c) Jl[r * 3 + c] = -(pj[r * 3] * R[c] + pj[r * 3 + 1] * R[3 + c] + pj[r * 3 + 2] * R[6 + c]);
      }
    } else {
      const double fx = cam.fx, fy = cam.fy, bf = cam.bf;
      if (unaryForm) {  // EdgeStereoSE3ProjectXYZOnlyPose::linearizeOplus (:375-403)
        const double invz = 1.0 / z, invz_2 = invz * invz;
        Jp[0] = x * y * invz_2 * fx; Jp[1] = -(1 + (x * x * invz_2)) * fx; Jp[2] = y * invz * fx;
        Jp[3] = -invz * fx; Jp[4] = 0; Jp[5] = x * invz_2 * fx;
        Jp[6] = (1 + y * y * invz_2) * fy; Jp[7] = -x * y * invz_2 * fy; Jp[8] = -x * invz * fy;
        Jp[9] = 0; Jp[10] = -invz * fy; Jp[11] = y * invz_2 * fy;
        Jp[12] = Jp[0] - bf * y * invz_2; Jp[13] = Jp[1] + bf * x * invz_2; Jp[14] = Jp[2];
        Jp[15] = Jp[3]; Jp[16] = 0; Jp[17] = Jp[5] - bf * invz_2;
        for (int i = 0; i < 9; ++i) Jl[i] = 0;
      } else {  // EdgeStereoSE3ProjectXYZ::linearizeOplus (:228-270)
        const double z_2 = z * z;
        Jl[0] = -fx * R[0] / z + fx * x * R[6] / z_2; Jl[1] = -fx * R[1] / z + fx * x * R[7] / z_2;
        Jl[2] = -fx * R[2] / z + fx * x * R[8] / z_2;
        Jl[3] = -fy * R[3] / z + fy * y * R[6] / z_2; Jl[4] = -fy * R[4] / z + fy * y * R[7] / z_2;
        Jl[5] = -fy * R[5] / z + fy * y * R[8] / z_2;
        Jl[6] = Jl[0] - bf * R[6] / z_2; Jl[7] = Jl[1] - bf * R[7] / z_2; Jl[8] = Jl[2] - bf * R[8] / z_2;
        Jp[0] = x * y / z_2 * fx; Jp[1] = -(1 + (x * x / z_2)) * fx; Jp[2] = y / z * fx;
        Jp[3] = -1. / z * fx; Jp[4] = 0; Jp[5] = x / z_2 * fx;
        Jp[6] = (1 + y * y / z_2) * fy; Jp[7] = -x * y / z_2 * fy; Jp[8] = -x / z * fy;
        Jp[9] = 0; Jp[10] = -1. / z * fy; Jp[11] = y / z_2 * fy;
        Jp[12] = Jp[0] - bf * y / z_2; Jp[13] = Jp[1] + bf * x / z_2; Jp[14] = Jp[2];
        Jp[15] = Jp[3]; Jp[16] = 0; Jp[17] = Jp[5] - bf / z_2;
      }
    }
  }

  // initializeOptimization(level): active edges = edges at that level (sparse_optimizer.cpp:166-290); free
  // vertices indexed poses first, then (marginalised) points
  void initializeOptimization(int level) {
    active.clear();
    for (int k = 0; k < (int)edges.size(); ++k) if (edges[k].level == level) active.push_back(k);
    poseCol.assign(poses.size(), -1);
    pointCol.assign(points.size() / 3, -1);
    std::vector<char> poseUsed(poses.size(), 0), pointUsed(points.size() / 3, 0);
    for (int k : active) { poseUsed[edges[k].pose] = 1; pointUsed[edges[k].point] = 1; }
    nFreePoses = nFreePoints = 0;
    for (size_t i = 0; i < poses.size(); ++i) if (poseUsed[i] && !poseFixed[i]) poseCol[i] = nFreePoses++;
    for (size_t i = 0; i < pointUsed.size(); ++i) if (pointUsed[i] && !pointFixed[i]) pointCol[i] = nFreePoints++;
  }

  void buildSystem() {  // block_solver.hpp:502-560 + the quadratic forms
    const int P = 6 * nFreePoses;
    Hpp.assign((size_t)P * P, 0.0);
    Hll.assign((size_t)nFreePoints * 9, 0.0);
    Hpl.assign(active.size() * 18, 0.0);
    hplPose.assign(active.size(), -1);
    hplPoint.assign(active.size(), -1);
    b.assign(P + 3 * nFreePoints, 0.0);
    for (size_t a = 0; a < active.size(); ++a) {
      const Edge& e = edges[active[a]];
      const int d = dim(e);
      double Jp[18], Jl[9];
      linearize(e, Jp, Jl);
      double w = 1.0;
      if (e.delta > 0) { double rho[3]; robustify(e.delta, chi2(e), rho); w = rho[1]; }
      const int pc = poseCol[e.pose], lc = pointCol[e.point];
      // omega_r = -omega * error (* rho'), weightedOmega = rho' * omega
      double wr[3];
      for (int i = 0; i < d; ++i) wr[i] = -e.info * e.err[i] * w;
      const double wo = w * e.info;
      if (pc >= 0) {
        for (int r = 0; r < 6; ++r) {
          double s = 0;
          for (int i = 0; i < d; ++i) s += Jp[i * 6 + r] * wr[i];
          b[6 * pc + r] += s;
          for (int c = 0; c < 6; ++c) {
            double h = 0;
            for (int i = 0; i < d; ++i) h += Jp[i * 6 + r] * wo * Jp[i * 6 + c];
            Hpp[(size_t)(6 * pc + r) * P + 6 * pc + c] += h;
          }
        }
      }
      if (lc >= 0) {
        for (int r = 0; r < 3; ++r) {
          double s = 0;
          for (int i = 0; i < d; ++i) s += Jl[i * 3 + r] * wr[i];
          b[P + 3 * lc + r] += s;
          for (int c = 0; c < 3; ++c) {
            double h = 0;
            for (int i = 0; i < d; ++i) h += Jl[i * 3 + r] * wo * Jl[i * 3 + c];
            Hll[(size_t)lc * 9 + r * 3 + c] += h;
          }
        }
      }
      if (pc >= 0 && lc >= 0) {
        hplPose[a] = pc; hplPoint[a] = lc;
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 3; ++c) {
            double h = 0;
            for (int i = 0; i < d; ++i) h += Jp[i * 6 + r] * wo * Jl[i * 3 + c];
            Hpl[a * 18 + r * 3 + c] = h;
          }
      }
    }
  }

  double computeLambdaInit() const {  // optimization_algorithm_levenberg.cpp:171-185
    if (userLambdaInit > 0) return userLambdaInit;
    const int P = 6 * nFreePoses;
    double m = 0;
    for (int i = 0; i < P; ++i) m = std::max(std::fabs(Hpp[(size_t)i * P + i]), m);
    for (int l = 0; l < nFreePoints; ++l)
      for (int j = 0; j < 3; ++j) m = std::max(std::fabs(Hll[(size_t)l * 9 + j * 4]), m);
    return 1e-5 * m;
  }

  // dense LDLT (Eigen::LDLT / SimplicialLDLT stand-in); requirePositive = LinearSolverDense's isPositive() test
  static bool ldltSolve(std::vector<double> A, int n, const double* rhs, double* out, bool requirePositive) {
    std::vector<double> D(n);
    for (int j = 0; j < n; ++j) {
      double d = A[(size_t)j * n + j];
      for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * D[k];
      if (requirePositive ? !(d > 0) : (d == 0 || d != d)) return false;
      D[j] = d;
      for (int i = j + 1; i < n; ++i) {
        double s = A[(size_t)i * n + j];
        for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * D[k];
        A[(size_t)i * n + j] = s / d;
      }
    }
    std::vector<double> y(rhs, rhs + n);
    for (int i = 0; i < n; ++i) for (int k = 0; k < i; ++k) y[i] -= A[(size_t)i * n + k] * y[k];
    for (int i = 0; i < n; ++i) y[i] /= D[i];
    for (int i = n - 1; i >= 0; --i) for (int k = i + 1; k < n; ++k) y[i] -= A[(size_t)k * n + i] * y[k];
    for (int i = 0; i < n; ++i) out[i] = y[i];
    return true;
  }
  static void inv3(const double m[9], double o[9]) {  // Eigen 3x3 inverse: cofactors / determinant
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  }

  // BlockSolver::solve with lambda already added to the diagonals (block_solver.hpp:354-480)
  bool solveSystem(double lam) {
    const int P = 6 * nFreePoses;
    x.assign(b.size(), 0.0);
    std::vector<double> H = Hpp;
    for (int i = 0; i < P; ++i) H[(size_t)i * P + i] += lam;
    if (nFreePoints == 0) return ldltSolve(H, P, b.data(), x.data(), true);  // LinearSolverDense
    std::vector<double> Dinv((size_t)nFreePoints * 9), bs(b.begin(), b.begin() + P);
    for (int l = 0; l < nFreePoints; ++l) {
      double D[9];
      memcpy(D, &Hll[(size_t)l * 9], sizeof D);
      D[0] += lam; D[4] += lam; D[8] += lam;
      inv3(D, &Dinv[(size_t)l * 9]);
    }
    // per landmark: edges (blocks) that touch it
    std::vector<std::vector<int>> byPoint(nFreePoints);
    for (size_t a = 0; a < active.size(); ++a) if (hplPoint[a] >= 0) byPoint[hplPoint[a]].push_back((int)a);
    for (int l = 0; l < nFreePoints; ++l) {
      const double* Di = &Dinv[(size_t)l * 9];
      double db[3];
      for (int r = 0; r < 3; ++r) db[r] = Di[r * 3] * b[P + 3 * l] + Di[r * 3 + 1] * b[P + 3 * l + 1] + Di[r * 3 + 2] * b[P + 3 * l + 2];
      for (int a1 : byPoint[l]) {
        const double* B1 = &Hpl[(size_t)a1 * 18];
        const int i1 = hplPose[a1];
        double BD[18];
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 3; ++c) BD[r * 3 + c] = B1[r * 3] * Di[c] + B1[r * 3 + 1] * Di[3 + c] + B1[r * 3 + 2] * Di[6 + c];
        for (int r = 0; r < 6; ++r) bs[6 * i1 + r] -= B1[r * 3] * db[0] + B1[r * 3 + 1] * db[1] + B1[r * 3 + 2] * db[2];
        for (int a2 : byPoint[l]) {
          const double* B2 = &Hpl[(size_t)a2 * 18];
          const int i2 = hplPose[a2];
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c)
              H[(size_t)(6 * i1 + r) * P + 6 * i2 + c] -= BD[r * 3] * B2[c * 3] + BD[r * 3 + 1] * B2[c * 3 + 1] + BD[r * 3 + 2] * B2[c * 3 + 2];
        }
      }
    }
    if (!ldltSolve(H, P, bs.data(), x.data(), false)) return false;  // LinearSolverEigen (SimplicialLDLT)
    for (int l = 0; l < nFreePoints; ++l) {  // xl = Dinv * (bl - Hpl^T xp)
      double cl[3] = {b[P + 3 * l], b[P + 3 * l + 1], b[P + 3 * l + 2]};
      for (int a : byPoint[l]) {
        const double* B = &Hpl[(size_t)a * 18];
        const int i1 = hplPose[a];
        for (int c = 0; c < 3; ++c)
          for (int r = 0; r < 6; ++r) cl[c] -= B[r * 3 + c] * x[6 * i1 + r];
      }
      const double* Di = &Dinv[(size_t)l * 9];
      for (int r = 0; r < 3; ++r) x[P + 3 * l + r] = Di[r * 3] * cl[0] + Di[r * 3 + 1] * cl[1] + Di[r * 3 + 2] * cl[2];
    }
    return true;
  }

  void push() { poseBackup = poses; pointBackup = points; }
  void pop() { poses = poseBackup; points = pointBackup; }
  void update() {  // SparseOptimizer::update -> oplus
    const int P = 6 * nFreePoses;
    for (size_t i = 0; i < poses.size(); ++i)
      if (poseCol[i] >= 0) poses[i] = mul(expSE3(&x[6 * poseCol[i]]), poses[i]);
    for (size_t i = 0; i < pointCol.size(); ++i)
      if (pointCol[i] >= 0) for (int r = 0; r < 3; ++r) points[3 * i + r] += x[P + 3 * pointCol[i] + r];
  }
  bool terminate() const { return forceStop && *forceStop; }

  enum Result { OK, Terminate };
  Result solve(int iteration) {  // optimization_algorithm_levenberg.cpp:61-169
    computeActiveErrors();
    double currentChi = activeRobustChi2();
    double tempChi = currentChi;
    const double iniChi = currentChi;
    buildSystem();
    if (g_orcTrace && g_orcTraceN == 0 && g_orcTraceCap >= 512 && nFreePoints == 0) { double* t = g_orcTrace + 6 * 504; for (int k = 0; k < 36; ++k) t[k] = Hpp[k]; for (int k = 0; k < 6; ++k) t[36 + k] = b[k]; }
    if (iteration == 0) { lambda = computeLambdaInit(); ni = 2; nBad = 0; }
    double rho = 0;
    int qmax = 0;
    do {
      push();
      const bool ok2 = solveSystem(lambda);
      update();
      computeActiveErrors();
      tempChi = activeRobustChi2();
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = (currentChi - tempChi);
      double scale = 0;
      for (size_t j = 0; j < x.size(); ++j) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (g_orcTrace && g_orcTraceN == 0 && g_orcTraceCap >= 512) { double* t = g_orcTrace + 6 * 500; for (int k = 0; k < 6; ++k) t[k] = x[k]; for (size_t i = 0; i < poses.size(); ++i) if (poseCol[i] >= 0) { for (int k = 0; k < 4; ++k) t[6 + k] = poses[i].q[k]; for (int k = 0; k < 3; ++k) t[10 + k] = poses[i].t[k]; } }
      if (g_orcTrace && g_orcTraceN < 500) { double* t = g_orcTrace + 6 * g_orcTraceN++; t[0] = currentChi; t[1] = tempChi; t[2] = lambda; t[3] = rho; t[4] = scale; t[5] = ok2; }
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        const double scaleFactor = std::max(1. / 3., alpha);
        lambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        pop();
      }
      qmax++;
    } while (rho < 0 && qmax < 10 && !terminate());
    levenbergIterations += qmax;
    if (qmax == 10 || rho == 0) return Terminate;
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) return Terminate;
    return OK;
  }
  int optimize(int iterations) {  // sparse_optimizer.cpp:354-420
    int cj = 0;
    bool ok = true;
    for (int i = 0; i < iterations && !terminate() && ok; i++) {
      ok = solve(i) == OK;
      ++cj;
    }
    return cj;
  }
};

}  // namespace ba
}  // namespace orc

using namespace orc::ba;
extern "C" {

// Optimizer::PoseOptimization on flattened inputs.  n candidate features; hasMP[i] != 0 <=> mvpMapPoints[i];
// obs[i] = (kpUn.x, kpUn.y, mvuRight[i]) (mvuRight < 0 => monocular edge); pose = unit quaternion (x,y,z,w) +
// translation as floats, in/out; outlier[i] = mvbOutlier[i].  Returns nInitialCorrespondences - nBad
// (0 when fewer than 3 correspondences).  stats (optional, 2 ints): outer LM iterations, LM trials.
void orc_set_trace(double* buf, int cap) { g_orcTrace = buf; g_orcTraceCap = cap; g_orcTraceN = 0; }
int orc_trace_count() { return g_orcTraceN; }
int orc_pose_optimization(int n, const uint8_t* hasMP, const float* obs, const float* invSigma2, const float* Xw,
                          float fx, float fy, float cx, float cy, float bf, float* pose, uint8_t* outlier, int* stats) {
  Graph g;
  g.cam = Camera{fx, fy, cx, cy, bf};
  g.unaryForm = true;
  g.poses.push_back(fromFloatPose(pose));
  g.poseFixed.push_back(0);
  const float deltaMono = (float)std::sqrt(5.991), deltaStereo = (float)std::sqrt(7.815);  // Optimizer.cc:802-803
  std::vector<int> featOfEdge;
  int nInitialCorrespondences = 0;
  for (int i = 0; i < n; ++i) {
    if (!hasMP[i]) continue;
    nInitialCorrespondences++;
    outlier[i] = 0;
    Edge e;
    memset(&e, 0, sizeof e);
    e.kind = obs[3 * i + 2] < 0 ? KIND_MONO : KIND_STEREO;
    e.pose = 0;
    e.point = (int)g.points.size() / 3;
    for (int k = 0; k < 3; ++k) { e.obs[k] = (double)obs[3 * i + k]; g.points.push_back((double)Xw[3 * i + k]); }
    g.pointFixed.push_back(1);
    e.info = (double)invSigma2[i];
    e.delta = e.kind == KIND_MONO ? (double)deltaMono : (double)deltaStereo;
    e.level = 0;
    g.edges.push_back(e);
    featOfEdge.push_back(i);
  }
  if (stats) stats[0] = stats[1] = 0;
  if (nInitialCorrespondences < 3) return 0;
  const float chi2Mono[4] = {5.991f, 5.991f, 5.991f, 5.991f}, chi2Stereo[4] = {7.815f, 7.815f, 7.815f, 7.815f};
  const SE3Quat initial = g.poses[0];
  int nBad = 0;
  for (size_t it = 0; it < 4; it++) {
    g.poses[0] = initial;  // vSE3->setEstimate(pFrame->GetPose()) — the frame pose is only written at the end
    g.initializeOptimization(0);
    const int its = g.optimize(10);
    if (stats) stats[0] += its;
    nBad = 0;
    for (size_t k = 0; k < g.edges.size(); ++k) {
      Edge& e = g.edges[k];
      const int idx = featOfEdge[k];
      if (outlier[idx]) g.computeError(e);
      const float chi2 = (float)g.chi2(e);
      const float th = e.kind == KIND_MONO ? chi2Mono[it] : chi2Stereo[it];
      if (chi2 > th) { outlier[idx] = 1; e.level = 1; nBad++; }
      else { outlier[idx] = 0; e.level = 0; }
      if (it == 2) e.delta = 0;
    }
    if (g.edges.size() < 10) break;
  }
  if (stats) stats[1] = g.levenbergIterations;
  for (int i = 0; i < 4; ++i) pose[i] = (float)g.poses[0].q[i];
  for (int i = 0; i < 3; ++i) pose[4 + i] = (float)g.poses[0].t[i];
  return nInitialCorrespondences - nBad;
}

// Optimizer::PoseOptimization for a fisheye rig (pFrame->mpCamera2 != NULL, Optimizer.cc:880-946): feature i < Nleft is a
// left-camera observation (EdgeSE3ProjectXYZOnlyPose, KB8 left), i >= Nleft a right-camera one
// (EdgeSE3ProjectXYZOnlyPoseToBody, KB8 right, mTrl).  obs = (x, y) per feature.
int orc_pose_optimization_fisheye(int n, int Nleft, const uint8_t* hasMP, const float* obs, const float* invSigma2,
                                  const float* Xw, const float* camL8, const float* camR8, const float* Trl7, float* pose,
                                  uint8_t* outlier, int* stats) {
  Graph g;
  g.cam = Camera{0, 0, 0, 0, 0};
  memcpy(g.kbL.p, camL8, 32); memcpy(g.kbR.p, camR8, 32);
  g.Trl = fromFloatPose(Trl7);
  g.unaryForm = true;
  g.poses.push_back(fromFloatPose(pose));
  g.poseFixed.push_back(0);
  const float deltaMono = (float)std::sqrt(5.991);
  std::vector<int> featOfEdge;
  int nInitialCorrespondences = 0;
  for (int i = 0; i < n; ++i) {
    if (!hasMP[i]) continue;
    nInitialCorrespondences++;
    outlier[i] = 0;
    Edge e;
    memset(&e, 0, sizeof e);
    e.kind = i < Nleft ? KIND_KB8_LEFT : KIND_KB8_RIGHT;
    e.pose = 0;
    e.point = (int)g.points.size() / 3;
    e.obs[0] = (double)obs[2 * i]; e.obs[1] = (double)obs[2 * i + 1];
    for (int k = 0; k < 3; ++k) g.points.push_back((double)Xw[3 * i + k]);
    g.pointFixed.push_back(1);
    e.info = (double)invSigma2[i];
    e.delta = (double)deltaMono;
    g.edges.push_back(e);
    featOfEdge.push_back(i);
  }
  if (stats) stats[0] = stats[1] = 0;
  if (nInitialCorrespondences < 3) return 0;
  const SE3Quat initial = g.poses[0];
  int nBad = 0;
  for (size_t it = 0; it < 4; it++) {
    g.poses[0] = initial;
    g.initializeOptimization(0);
    const int its = g.optimize(10);
    if (stats) stats[0] += its;
    nBad = 0;
    for (size_t k = 0; k < g.edges.size(); ++k) {
      Edge& e = g.edges[k];
      const int idx = featOfEdge[k];
      if (outlier[idx]) g.computeError(e);
      const float chi2 = (float)g.chi2(e);
      if (chi2 > 5.991f) { outlier[idx] = 1; e.level = 1; nBad++; }
      else { outlier[idx] = 0; e.level = 0; }
      if (it == 2) e.delta = 0;
    }
    if (g.edges.size() < 10) break;
  }
  if (stats) stats[1] = g.levenbergIterations;
  for (int i = 0; i < 4; ++i) pose[i] = (float)g.poses[0].q[i];
  for (int i = 0; i < 3; ++i) pose[4 + i] = (float)g.poses[0].t[i];
  return nInitialCorrespondences - nBad;
}

// Optimizer::LocalBundleAdjustment on a flattened problem: nKF keyframe poses (float quaternion xyzw +
// translation, in/out; kfFixed[i] != 0 for lFixedCameras and the map's initial keyframe), nMP points (float xyz,
// in/out), nE observations (kf index, mp index, obs (x, y, uRight) with uRight < 0 => monocular, invSigma2).
// lambdaInit100 != 0 <=> pMap->IsInertial() (:1137).  eraseFlag[e] = 1 when the reference would erase the
// observation (:1366-1401).  Returns the number of outer LM iterations run by optimize(10).
int orc_local_ba(int nKF, float* kfPose, const uint8_t* kfFixed, int nMP, float* mpPos, int nE, const int* eKF,
                 const int* eMP, const float* eObs, const float* eInvSigma2, float fx, float fy, float cx, float cy,
                 float bf, int lambdaInit100, const int* stopFlag, uint8_t* eraseFlag, int* stats) {
  Graph g;
  g.cam = Camera{fx, fy, cx, cy, bf};
  for (int i = 0; i < nKF; ++i) { g.poses.push_back(fromFloatPose(kfPose + 7 * i)); g.poseFixed.push_back(kfFixed[i] ? 1 : 0); }
  for (int i = 0; i < nMP; ++i) { for (int k = 0; k < 3; ++k) g.points.push_back((double)mpPos[3 * i + k]); g.pointFixed.push_back(0); }
  const float thHuberMono = (float)std::sqrt(5.991), thHuberStereo = (float)std::sqrt(7.815);
  for (int k = 0; k < nE; ++k) {
    Edge e;
    memset(&e, 0, sizeof e);
    e.kind = eObs[3 * k + 2] < 0 ? KIND_MONO : KIND_STEREO;
    e.pose = eKF[k]; e.point = eMP[k];
    for (int j = 0; j < 3; ++j) e.obs[j] = (double)eObs[3 * k + j];
    e.info = (double)eInvSigma2[k];
    e.delta = e.kind == KIND_MONO ? (double)thHuberMono : (double)thHuberStereo;
    g.edges.push_back(e);
  }
  if (lambdaInit100) g.userLambdaInit = 100.0;
  g.forceStop = stopFlag;
  if (stats) stats[0] = stats[1] = 0;
  if (stopFlag && *stopFlag) return 0;  // :1355-1356
  g.initializeOptimization(0);
  const int its = g.optimize(10);
  if (stats) { stats[0] = its; stats[1] = g.levenbergIterations; }
  for (int k = 0; k < nE; ++k) {
    const Edge& e = g.edges[k];
    double xc[3];
    mapPoint(g.poses[e.pose], &g.points[3 * e.point], xc);
    const bool depthPositive = xc[2] > 0.0;
    const double th = e.kind == KIND_MONO ? 5.991 : 7.815;
    eraseFlag[k] = (g.chi2(e) > th || !depthPositive) ? 1 : 0;
  }
  for (int i = 0; i < nKF; ++i) {
    // fixed keyframes of lFixedCameras are not written back by the reference; local ones are (:1418-1428);
    // writing back an unchanged fixed pose is the identity up to the float round trip, so only free ones here
    if (kfFixed[i]) continue;
    for (int k = 0; k < 4; ++k) kfPose[7 * i + k] = (float)g.poses[i].q[k];
    for (int k = 0; k < 3; ++k) kfPose[7 * i + 4 + k] = (float)g.poses[i].t[k];
  }
  for (int i = 0; i < nMP; ++i) for (int k = 0; k < 3; ++k) mpPos[3 * i + k] = (float)g.points[3 * i + k];
  return its;
}

// LocalBundleAdjustment on a KannalaBrandt8 rig (pKFi->mpCamera2, Optimizer.cc:1244-1351): every observation is an
// EdgeSE3ProjectXYZ with the left camera (eRight == 0) or an EdgeSE3ProjectXYZToBody with the right camera (eRight != 0).
int orc_local_ba_fisheye(int nKF, float* kfPose, const uint8_t* kfFixed, int nMP, float* mpPos, int nE, const int* eKF,
                         const int* eMP, const float* eObs2, const uint8_t* eRight, const float* eInvSigma2, const float* camL8,
                         const float* camR8, const float* Trl7, int lambdaInit100, const int* stopFlag, uint8_t* eraseFlag,
                         int* stats) {
  Graph g;
  g.cam = Camera{0, 0, 0, 0, 0};
  memcpy(g.kbL.p, camL8, 32); memcpy(g.kbR.p, camR8, 32);
  g.Trl = fromFloatPose(Trl7);
  for (int i = 0; i < nKF; ++i) { g.poses.push_back(fromFloatPose(kfPose + 7 * i)); g.poseFixed.push_back(kfFixed[i] ? 1 : 0); }
  for (int i = 0; i < nMP; ++i) { for (int k = 0; k < 3; ++k) g.points.push_back((double)mpPos[3 * i + k]); g.pointFixed.push_back(0); }
  const float thHuberMono = (float)std::sqrt(5.991);
  for (int k = 0; k < nE; ++k) {
    Edge e;
    memset(&e, 0, sizeof e);
    e.kind = eRight[k] ? KIND_KB8_RIGHT : KIND_KB8_LEFT;
    e.pose = eKF[k]; e.point = eMP[k];
    e.obs[0] = (double)eObs2[2 * k]; e.obs[1] = (double)eObs2[2 * k + 1];
    e.info = (double)eInvSigma2[k];
    e.delta = (double)thHuberMono;
    g.edges.push_back(e);
  }
  if (lambdaInit100) g.userLambdaInit = 100.0;
  g.forceStop = stopFlag;
  if (stats) stats[0] = stats[1] = 0;
  if (stopFlag && *stopFlag) return 0;
  g.initializeOptimization(0);
  const int its = g.optimize(10);
  if (stats) { stats[0] = its; stats[1] = g.levenbergIterations; }
  for (int k = 0; k < nE; ++k) {
    const Edge& e = g.edges[k];
    double xc[3];
    if (e.kind == KIND_KB8_RIGHT) mapPoint(mul(g.Trl, g.poses[e.pose]), &g.points[3 * e.point], xc);   // OptimizableTypes.h:152-158
    else mapPoint(g.poses[e.pose], &g.points[3 * e.point], xc);
    eraseFlag[k] = (g.chi2(e) > 5.991 || !(xc[2] > 0.0)) ? 1 : 0;   // :1366-1390
  }
  for (int i = 0; i < nKF; ++i) {
    if (kfFixed[i]) continue;
    for (int k = 0; k < 4; ++k) kfPose[7 * i + k] = (float)g.poses[i].q[k];
    for (int k = 0; k < 3; ++k) kfPose[7 * i + 4 + k] = (float)g.poses[i].t[k];
  }
  for (int i = 0; i < nMP; ++i) for (int k = 0; k < 3; ++k) mpPos[3 * i + k] = (float)g.points[3 * i + k];
  return its;
}
}
