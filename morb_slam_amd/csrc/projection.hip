// Projection-guided matchers and the frustum test for MI355X (gfx950), batched over frames, non-fisheye
// branches (Frame::Nleft == -1):
//   F3  Frame::isInFrustum + MapPoint::PredictScale          (reference src/Frame.cc:611-678, MapPoint.cc:536-566)
//   F4  Frame::PosInGrid / GetFeaturesInArea semantics       (Frame.cc:742-820; 64 x 48 grid, Frame.h:44-45)
//   M1  ORBmatcher::SearchByProjection(Frame&, MapPoints..)  (src/ORBmatcher.cc:42-209)
//   M2  ORBmatcher::SearchByProjection(Cur, Last, th, bMono) (:1521-1733)
//   M4  ORBmatcher::SearchForTriangulation                   (:821-1042)
// The reference walks a 64 x 48 grid of index vectors and keeps "first best" in the walk order (cell column,
// cell row, insertion order).  Here no grid is built: a wave tests every feature of the frame against the query
// window (same cell-range + level + |dx|,|dy| < r predicate, PosInGrid's round() included) and packs
// (distance, cell, index, octave) into one 64-bit key whose ordering IS the reference's walk order, so best /
// second best are plain wave64 min-reductions.  The greedy dependency ("skip features that already hold a map
// point with observations", which includes the ones assigned earlier in the same call) is resolved by a second
// kernel: one wave per frame replays the queries in reference order over the pre-computed candidate keys.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"
#include "internal_abi.h"
#include "libm_f32.h"
#include "wave.h"
#include "kb8.h"

using namespace morb;

struct morb_matcher;  // defined in matcher.hip
extern "C" {
void* morb_matcher_stream(const morb_matcher*);
}

namespace {

constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;
constexpr int GRID_ROWS = 48, GRID_COLS = 64;
constexpr int CAND_CAP = 128;  // candidate keys kept per query; longer lists are re-derived in the resolve pass

struct Desc { uint32_t w[8]; };
__device__ __forceinline__ Desc load_desc(const uint8_t* p) {
  Desc d;
  const uint4* q = reinterpret_cast<const uint4*>(p);
  const uint4 a = q[0], b = q[1];
  d.w[0] = a.x; d.w[1] = a.y; d.w[2] = a.z; d.w[3] = a.w; d.w[4] = b.x; d.w[5] = b.y; d.w[6] = b.z; d.w[7] = b.w;
  return d;
}
__device__ __forceinline__ int hamming(const Desc& a, const Desc& b) {
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += __popc(a.w[i] ^ b.w[i]);
  return s;
}
// (selects on the values: as `if (k < k1) { k2 = k1; k1 = k; } else if (k < k2) k2 = k;` the compiler stored k through a run-time-selected
// address of k1 / k2, i.e. a private array in scratch memory inside the candidate loops)
__device__ __forceinline__ void top2_insert(unsigned long long& k1, unsigned long long& k2, unsigned long long k) {
  const bool lt1 = k < k1, lt2 = k < k2;
  k2 = lt1 ? k1 : (lt2 ? k : k2);
  k1 = lt1 ? k : k1;
}
// the two smallest of the lanes' (k1 <= k2) pairs, in every lane.  Keys carry the candidate index, so they are unique (apart
// from the ~0 sentinel): the runner-up is the smallest of "k2 of the lane that owns the winner, k1 of every other lane".  Two
// DPP reductions (wave.h) instead of a 6-step butterfly of four ds_bpermute each: this sits on the critical path of the
// sequential resolve loops.  All 64 lanes must be active.
__device__ __forceinline__ void wave_top2(unsigned long long& k1, unsigned long long& k2) {
  const unsigned long long g1 = morbwave::min_u64(k1);
  const unsigned long long g2 = morbwave::min_u64(k1 == g1 ? k2 : k1);
  k1 = g1; k2 = g2;
}

struct Query {       // one window search (a map point / a last-frame feature)
  float x, y, r;     // window centre and half size
  int minLevel, maxLevel;
  float xr, erMax;   // stereo gate: |xr - uRight[j]| <= erMax for features with uRight > 0
  float angle;       // query keypoint angle (rotation histogram, M2)
  int valid;
  int jLo, jHi;      // feature index range searched; jHi == 0 means "all features" (fisheye rig: left | right halves)
  int gate;          // 1: Fuse's reprojection gate (ORBmatcher.cc:1160-1181) on (x, y, xr) against the candidate keypoint
};

// key = dist << 32 | cell << 20 | j << 4 | octave   (cell = posX * 48 + posY: the GetFeaturesInArea walk order)
__device__ __forceinline__ unsigned long long make_key(const morb_frame_params& P, const Query& q, const Desc& qd,
                                                       const morb_keypoint& kp, int j, const uint8_t* descRow,
                                                       const float* uRight, int cx0, int cx1, int cy0, int cy1) {
  const int posX = (int)roundf((kp.x - P.minX) * P.gridInvW), posY = (int)roundf((kp.y - P.minY) * P.gridInvH);
  if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) return ~0ull;  // never entered the grid
  if (posX < cx0 || posX > cx1 || posY < cy0 || posY > cy1) return ~0ull;
  const bool bCheckLevels = (q.minLevel > 0) || (q.maxLevel >= 0);
  if (bCheckLevels) {
    if (kp.octave < q.minLevel) return ~0ull;
    if (q.maxLevel >= 0 && kp.octave > q.maxLevel) return ~0ull;
  }
  const float distx = kp.x - q.x, disty = kp.y - q.y;
  if (!(fabsf(distx) < q.r && fabsf(disty) < q.r)) return ~0ull;
  if (uRight) {
    const float ur = uRight[j];
    if (ur > 0) {
      const float er = fabsf(q.xr - ur);
      if (er > q.erMax) return ~0ull;
    }
  }
  if (q.gate == 1) {   // mvInvLevelSigma2[l] = 1.0f / mvLevelSigma2[l] (ORBextractor.cc:421)
    const float invS = 1.0f / P.levelSigma2[kp.octave & 15];
    const float ex = q.x - kp.x, ey = q.y - kp.y;
    if (uRight && uRight[j] >= 0) {
      const float er = q.xr - uRight[j];
      const float e2 = ex * ex + ey * ey + er * er;
      if ((double)(e2 * invS) > 7.8) return ~0ull;
    } else {
      const float e2 = ex * ex + ey * ey;
      if ((double)(e2 * invS) > 5.99) return ~0ull;
    }
  }
  const int d = hamming(qd, load_desc(descRow));
  return ((unsigned long long)d << 32) | ((unsigned long long)(posX * GRID_ROWS + posY) << 20) |
         ((unsigned long long)j << 4) | (unsigned long long)(kp.octave & 15);
}
__device__ __forceinline__ bool cell_range(const morb_frame_params& P, const Query& q, int& cx0, int& cx1, int& cy0, int& cy1) {
  cx0 = max(0, (int)floorf((q.x - P.minX - q.r) * P.gridInvW));
  if (cx0 >= GRID_COLS) return false;
  cx1 = min(GRID_COLS - 1, (int)ceilf((q.x - P.minX + q.r) * P.gridInvW));
  if (cx1 < 0) return false;
  cy0 = max(0, (int)floorf((q.y - P.minY - q.r) * P.gridInvH));
  if (cy0 >= GRID_ROWS) return false;
  cy1 = min(GRID_ROWS - 1, (int)ceilf((q.y - P.minY + q.r) * P.gridInvH));
  if (cy1 < 0) return false;
  return true;
}

// ---------------------------------------------------------------------------------------------------
// F3: isInFrustum, one thread per map point
__global__ __launch_bounds__(256) void k_frustum(morb_frame_params P, const float* __restrict__ Rcw, const float* __restrict__ tcw,
                                                 const float* __restrict__ Ow, int mpCap, const int* __restrict__ nMPv,
                                                 const float* __restrict__ Pw, const float* __restrict__ normal,
                                                 const float* __restrict__ maxDist, const float* __restrict__ minDist,
                                                 float viewingCosLimit, const float* __restrict__ ratioThr,
                                                 const float* __restrict__ kb8,   // NULL: pinhole of P; else fx fy cx cy k0..k3
                                                 uint8_t* __restrict__ inView, float* __restrict__ projX,
                                                 float* __restrict__ projY, float* __restrict__ projXR,
                                                 float* __restrict__ depth, int* __restrict__ level, float* __restrict__ viewCosOut) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nMPv[f]) return;
  const size_t o = (size_t)f * mpCap + i;
  uint8_t in = 0;
  float pX = -1, pY = -1, pXR = -1, dep = -1, vc = -1;
  int lvl = -1;
  const float* R = Rcw + 9 * f; const float* t = tcw + 3 * f; const float* O = Ow + 3 * f;
  const float* X = Pw + o * 3;
  float Pc[3];
  for (int r = 0; r < 3; ++r) Pc[r] = (R[r * 3] * X[0] + R[r * 3 + 1] * X[1]) + R[r * 3 + 2] * X[2] + t[r];
  const float Pc_dist = sqrtf(Pc[0] * Pc[0] + Pc[1] * Pc[1] + Pc[2] * Pc[2]);
  const float invz = 1.0f / Pc[2];
  do {
    if (Pc[2] < 0.0f) break;
    float u, v;
    if (kb8) {   // KannalaBrandt8::project(Vector3f) (KannalaBrandt8.cpp:49-67): isInFrustumChecks, Frame.cc:1304-1309
      const float x2_plus_y2 = Pc[0] * Pc[0] + Pc[1] * Pc[1];
      const float theta = morbm::atan2f_glibc(sqrtf(x2_plus_y2), Pc[2]);
      const float psi = morbm::atan2f_glibc(Pc[1], Pc[0]);
      const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
                  theta9 = theta7 * theta2;
      const float r = theta + kb8[4] * theta3 + kb8[5] * theta5 + kb8[6] * theta7 + kb8[7] * theta9;
      u = kb8[0] * r * morbm::cosf_glibc(psi) + kb8[2];
      v = kb8[1] * r * morbm::sinf_glibc(psi) + kb8[3];
    } else {
      u = P.fx * Pc[0] / Pc[2] + P.cx; v = P.fy * Pc[1] / Pc[2] + P.cy;
    }
    if (u < P.minX || u > P.maxX) break;
    if (v < P.minY || v > P.maxY) break;
    if (!kb8) { pX = u; pY = v; }   // the pinhole path stores the projection before the remaining checks (Frame.cc:633-634)
    const float maxDistance = 1.2f * maxDist[o], minDistance = 0.8f * minDist[o];
    const float PO[3] = {X[0] - O[0], X[1] - O[1], X[2] - O[2]};
    const float dist = sqrtf(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    if (dist < minDistance || dist > maxDistance) break;
    const float* Pn = normal + o * 3;
    const float viewCos = (PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) / dist;
    if (viewCos < viewingCosLimit) break;
    // PredictScale: ceil(logf(ratio) / logScaleFactor) clamped to [0, nlevels-1].  ratioThr[n] = the largest float
    // ratio whose predicted level is <= n, tabulated on the host with the host libm's logf (the reference's logf),
    // so the device needs no logf of its own and reproduces the reference's rounding exactly.
    const float ratio = maxDist[o] / dist;
    int n = 0;
    while (n < P.nlevels - 1 && ratio > ratioThr[n]) ++n;
    in = 1; pX = u; pY = v; pXR = kb8 ? -1.0f : u - P.mbf * invz; dep = Pc_dist; lvl = n; vc = viewCos;
  } while (0);
  inView[o] = in; projX[o] = pX; projY[o] = pY; projXR[o] = pXR; depth[o] = dep; level[o] = lvl; viewCosOut[o] = vc;
}

// ---------------------------------------------------------------------------------------------------
// query preparation
__global__ __launch_bounds__(256) void k_prep_mps(morb_frame_params P, int mpCap, const int* __restrict__ nMPv,
                                                  const uint8_t* __restrict__ inView, const uint8_t* __restrict__ isBad,
                                                  const float* __restrict__ depth, const float* __restrict__ projX,
                                                  const float* __restrict__ projY, const float* __restrict__ projXR,
                                                  const int* __restrict__ level, const float* __restrict__ viewCos, float th,
                                                  int bFarPoints, float thFarPoints, Query* __restrict__ qs) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= mpCap) return;
  const size_t o = (size_t)f * mpCap + i;
  Query q;
  memset(&q, 0, sizeof q);
  if (i < nMPv[f] && inView[o] && !(bFarPoints && depth[o] > thFarPoints) && !isBad[o]) {
    const int lv = level[o];
    float r = ((double)viewCos[o] > 0.998) ? 2.5f : 4.0f;  // RadiusByViewingCos (:211-216)
    if ((double)th != 1.0) r *= th;
    q.valid = 1; q.x = projX[o]; q.y = projY[o]; q.r = r * P.scaleFactors[lv];
    q.minLevel = lv - 1; q.maxLevel = lv; q.xr = projXR[o]; q.erMax = r * P.scaleFactors[lv];
  }
  qs[o] = q;
}

// Fisheye rig: two queries per map point — 2i: left camera (features [0, Nleft), radius x th), 2i + 1: right camera
// (features [Nleft, N), no th factor, only if mnTrackScaleLevelR != -1); ORBmatcher.cc:50-66 / :140-151.
__global__ __launch_bounds__(256) void k_prep_mps_fisheye(morb_frame_params P, int mpCap, const int* __restrict__ nMPv,
                                                          const int* __restrict__ fImg, const int* __restrict__ count,
                                                          const int* __restrict__ nLeftv, const uint8_t* __restrict__ inViewL,
                                                          const uint8_t* __restrict__ inViewR, const uint8_t* __restrict__ isBad,
                                                          const float* __restrict__ depthL, const float* __restrict__ projXL,
                                                          const float* __restrict__ projYL, const int* __restrict__ levelL,
                                                          const float* __restrict__ viewCosL, const float* __restrict__ projXR,
                                                          const float* __restrict__ projYR, const int* __restrict__ levelR,
                                                          const float* __restrict__ viewCosR, float th, int bFarPoints,
                                                          float thFarPoints, Query* __restrict__ qs) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= mpCap) return;
  const size_t o = (size_t)f * mpCap + i;
  Query ql, qr;
  memset(&ql, 0, sizeof ql);
  memset(&qr, 0, sizeof qr);
  const int nLeft = nLeftv[f], N = count[fImg[f]];
  if (i < nMPv[f] && (inViewL[o] || inViewR[o]) && !(bFarPoints && depthL[o] > thFarPoints) && !isBad[o]) {
    if (inViewL[o] && nLeft > 0) {
      const int lv = levelL[o];
      float r = ((double)viewCosL[o] > 0.998) ? 2.5f : 4.0f;
      if ((double)th != 1.0) r *= th;
      ql.valid = 1; ql.x = projXL[o]; ql.y = projYL[o]; ql.r = r * P.scaleFactors[lv];
      ql.minLevel = lv - 1; ql.maxLevel = lv; ql.jLo = 0; ql.jHi = nLeft;
    }
    if (inViewR[o] && levelR[o] != -1 && N > nLeft) {
      const int lv = levelR[o];
      const float r = ((double)viewCosR[o] > 0.998) ? 2.5f : 4.0f;
      qr.valid = 1; qr.x = projXR[o]; qr.y = projYR[o]; qr.r = r * P.scaleFactors[lv];
      qr.minLevel = lv - 1; qr.maxLevel = lv; qr.jLo = nLeft; qr.jHi = N;
    }
  }
  qs[2 * o] = ql;
  qs[2 * o + 1] = qr;
}

__device__ __forceinline__ void q_rotate_f(const float* q, const float* v, float* out) {
  const float ux = q[0], uy = q[1], uz = q[2], w = q[3];
  float a = uy * v[2] - uz * v[1], b = uz * v[0] - ux * v[2], c = ux * v[1] - uy * v[0];
  a += a; b += b; c += c;
  out[0] = v[0] + w * a + (uy * c - uz * b);
  out[1] = v[1] + w * b + (uz * a - ux * c);
  out[2] = v[2] + w * c + (ux * b - uy * a);
}

__global__ __launch_bounds__(256) void k_prep_last(morb_frame_params P, int cap, const int* __restrict__ count,
                                                   const int* __restrict__ lastImg, const morb_keypoint* __restrict__ kps,
                                                   const uint8_t* __restrict__ lastValid, const float* __restrict__ lastXw,
                                                   const float* __restrict__ Tcw, float th, const uint8_t* __restrict__ fwd,
                                                   const uint8_t* __restrict__ bwd, Query* __restrict__ qs, int* __restrict__ nqOut) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (nqOut && i == 0) nqOut[f] = count[lastImg[f]];   // the frame's query count = the LAST image's features (was a launch of its own)
  if (i >= cap) return;
  const size_t o = (size_t)f * cap + i;
  Query q;
  memset(&q, 0, sizeof q);
  const int img = lastImg[f];
  if (i < count[img] && lastValid[o]) {
    const float* T = Tcw + 7 * f;
    float x3Dc[3];
    q_rotate_f(T, lastXw + o * 3, x3Dc);
    x3Dc[0] += T[4]; x3Dc[1] += T[5]; x3Dc[2] += T[6];
    const float invzc = (float)(1.0 / (double)x3Dc[2]);
    const float u = P.fx * x3Dc[0] / x3Dc[2] + P.cx, v = P.fy * x3Dc[1] / x3Dc[2] + P.cy;
    if (!(invzc < 0) && !(u < P.minX || u > P.maxX) && !(v < P.minY || v > P.maxY)) {
      const morb_keypoint kp = kps[(size_t)img * cap + i];
      const int oct = kp.octave;
      q.valid = 1; q.x = u; q.y = v; q.r = th * P.scaleFactors[oct];
      if (fwd[f]) { q.minLevel = oct; q.maxLevel = -1; }
      else if (bwd[f]) { q.minLevel = 0; q.maxLevel = oct; }
      else { q.minLevel = oct - 1; q.maxLevel = oct + 1; }
      q.xr = u - P.mbf * invzc; q.erMax = q.r; q.angle = kp.angle;
    }
  }
  qs[o] = q;
}

// Fisheye current frame (CurrentFrame.Nleft != -1, ORBmatcher.cc:1521-1733): query 2i = left pass, 2i + 1 = right pass
// of last-frame feature i.  Both project with the LEFT camera model (mpCamera; the right pass after
// GetRelativePoseTrl() — the reference's quirk); the right pass has no bounds / depth test of its own.
__device__ __forceinline__ void kb8_project_dev(const float* c, const float* v3, float& u, float& v) {   // KannalaBrandt8.cpp:49-67
  const float x2_plus_y2 = v3[0] * v3[0] + v3[1] * v3[1];
  const float theta = morbm::atan2f_glibc(sqrtf(x2_plus_y2), v3[2]);
  const float psi = morbm::atan2f_glibc(v3[1], v3[0]);
  const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2,
              theta9 = theta7 * theta2;
  const float r = theta + c[4] * theta3 + c[5] * theta5 + c[6] * theta7 + c[7] * theta9;
  u = c[0] * r * morbm::cosf_glibc(psi) + c[2];
  v = c[1] * r * morbm::sinf_glibc(psi) + c[3];
}
__global__ __launch_bounds__(256) void k_prep_last_fisheye(morb_frame_params P, int cap, const int* __restrict__ count,
                                                           const int* __restrict__ lastImg, const int* __restrict__ curImg,
                                                           const int* __restrict__ nLeftCur, const morb_keypoint* __restrict__ kps,
                                                           const uint8_t* __restrict__ lastValid, const float* __restrict__ lastXw,
                                                           const float* __restrict__ Tcw, const float* __restrict__ camTrl,   // [8 + 7]
                                                           float th, const uint8_t* __restrict__ fwd,
                                                           const uint8_t* __restrict__ bwd, Query* __restrict__ qs) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  const size_t o = (size_t)f * cap + i;
  Query ql, qr;
  memset(&ql, 0, sizeof ql);
  memset(&qr, 0, sizeof qr);
  const int img = lastImg[f];
  if (i < count[img] && lastValid[o]) {
    const float* T = Tcw + 7 * f;
    float x3Dc[3];
    q_rotate_f(T, lastXw + o * 3, x3Dc);
    x3Dc[0] += T[4]; x3Dc[1] += T[5]; x3Dc[2] += T[6];
    const float invzc = (float)(1.0 / (double)x3Dc[2]);
    float u, v;
    kb8_project_dev(camTrl, x3Dc, u, v);
    if (!(invzc < 0) && !(u < P.minX || u > P.maxX) && !(v < P.minY || v > P.maxY)) {
      const morb_keypoint kp = kps[(size_t)img * cap + i];
      const int oct = kp.octave;
      const int nLeft = nLeftCur[f], N = count[curImg[f]];
      ql.valid = 1; ql.x = u; ql.y = v; ql.r = th * P.scaleFactors[oct];
      if (fwd[f]) { ql.minLevel = oct; ql.maxLevel = -1; }
      else if (bwd[f]) { ql.minLevel = 0; ql.maxLevel = oct; }
      else { ql.minLevel = oct - 1; ql.maxLevel = oct + 1; }
      ql.angle = kp.angle; ql.jLo = 0; ql.jHi = nLeft;
      if (nLeft <= 0) ql.valid = 0;
      qr = ql;
      const float* Trl = camTrl + 8;
      float x3Dr[3];
      q_rotate_f(Trl, x3Dc, x3Dr);
      x3Dr[0] += Trl[4]; x3Dr[1] += Trl[5]; x3Dr[2] += Trl[6];
      kb8_project_dev(camTrl, x3Dr, qr.x, qr.y);
      qr.jLo = nLeft; qr.jHi = N; qr.valid = (ql.valid && N > nLeft) ? 1 : 0;
    }
  }
  qs[2 * o] = ql;
  qs[2 * o + 1] = qr;
}

// ---------------------------------------------------------------------------------------------------
// M7: queries of the keyframe-projection searches — Fuse x2 (ORBmatcher.cc:1044-1321), SearchByProjection(KF, Sim3) x2
// (:397-601), one direction of SearchBySim3 (:1354-1499).  One thread per map point.
//   projMode 0: pCamera->project(p3Dc) = fx * X / Z + cx            (Fuse, SearchByProjection v1; KB8 when cam8 != NULL)
//            1: invz = 1 / Z (float), u = fx * (X * invz) + cx       (SearchByProjection with vpPointsKFs, :531-536)
//            2: invz = (float)(1.0 / Z), same form                    (SearchBySim3, :1373-1378)
//   sim (8 floats per problem, RxSO3 quaternion + translation, or NULL): applied after T; then dist3D = |p| in the
//   target camera and there is no viewing-angle test (SearchBySim3); otherwise dist3D = |Pw - Ow| and PO . Pn >= dist / 2.
__device__ __forceinline__ void sim3_map_f(const float* S, const float* p, float* out) {   // Sophus rxso3.hpp:265-273
  const float qx = S[0], qy = S[1], qz = S[2], qw = S[3];
  const float scale = ((qx * qx + qy * qy) + qz * qz) + qw * qw;
  float a = qy * p[2] - qz * p[1], b = qz * p[0] - qx * p[2], c = qx * p[1] - qy * p[0];
  a += a; b += b; c += c;
  out[0] = scale * p[0] + (qw * a + (qy * c - qz * b)) + S[4];
  out[1] = scale * p[1] + (qw * b + (qz * a - qx * c)) + S[5];
  out[2] = scale * p[2] + (qw * c + (qx * b - qy * a)) + S[6];
}
__global__ __launch_bounds__(256) void k_prep_kfproj(morb_frame_params P, int mpCap, const int* __restrict__ nMPv,
                                                     const uint8_t* __restrict__ valid, const float* __restrict__ Pw,
                                                     const float* __restrict__ normal, const float* __restrict__ maxDist,
                                                     const float* __restrict__ minDist, const float* __restrict__ T,
                                                     const float* __restrict__ sim, const float* __restrict__ Ow,
                                                     const float* __restrict__ ratioThr, const float* __restrict__ kb8,
                                                     const int* __restrict__ jLo, const int* __restrict__ jHi, float th,
                                                     int projMode, int gate, Query* __restrict__ qs) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= mpCap) return;
  const size_t o = (size_t)f * mpCap + i;
  Query q;
  memset(&q, 0, sizeof q);
  do {
    if (i >= nMPv[f] || !valid[o]) break;
    const float* X = Pw + o * 3;
    float p[3];
    q_rotate_f(T + 7 * f, X, p);
    p[0] += T[7 * f + 4]; p[1] += T[7 * f + 5]; p[2] += T[7 * f + 6];
    if (sim) { float p2[3]; sim3_map_f(sim + 8 * f, p, p2); p[0] = p2[0]; p[1] = p2[1]; p[2] = p2[2]; }
    if (p[2] < 0.0f) break;
    float u, v;
    if (kb8) kb8_project_dev(kb8, p, u, v);
    else if (projMode == 0) { u = P.fx * p[0] / p[2] + P.cx; v = P.fy * p[1] / p[2] + P.cy; }
    else {
      const float invz = projMode == 1 ? 1 / p[2] : (float)(1.0 / (double)p[2]);
      const float x = p[0] * invz, y = p[1] * invz;
      u = P.fx * x + P.cx; v = P.fy * y + P.cy;
    }
    if (!(u >= P.minX && u < P.maxX && v >= P.minY && v < P.maxY)) break;          // KeyFrame::IsInImage
    const float maxDistance = 1.2f * maxDist[o], minDistance = 0.8f * minDist[o];
    float dist3D;
    if (sim) dist3D = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    else {
      const float* O = Ow + 3 * f;
      const float PO[3] = {X[0] - O[0], X[1] - O[1], X[2] - O[2]};
      dist3D = sqrtf(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
      if (dist3D < minDistance || dist3D > maxDistance) break;
      const float* Pn = normal + o * 3;
      if ((double)(PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) < 0.5 * (double)dist3D) break;
    }
    if (dist3D < minDistance || dist3D > maxDistance) break;
    const float ratio = maxDist[o] / dist3D;       // MapPoint::PredictScale through the host-built threshold table
    int lvl = 0;
    while (lvl < P.nlevels - 1 && ratio > ratioThr[lvl]) ++lvl;
    q.valid = 1; q.x = u; q.y = v; q.r = th * P.scaleFactors[lvl];
    q.minLevel = lvl - 1; q.maxLevel = lvl;
    q.xr = u - P.mbf * (1 / p[2]); q.erMax = 3.4e38f;   // ur of Fuse's gate; the M1 / M2 stereo-window gate never fires
    q.gate = gate;
    if (jHi) { q.jLo = jLo ? jLo[f] : 0; q.jHi = jHi[f]; }
  } while (0);
  qs[o] = q;
}

// Best candidate of every query, queries independent of each other (no "already taken" state): one wave per query.
__global__ __launch_bounds__(256) void k_best_per_query(morb_frame_params P, int qCap, const Query* __restrict__ qs,
                                                        const uint8_t* __restrict__ qDesc, const int* __restrict__ fImg, int cap,
                                                        const int* __restrict__ count, const morb_keypoint* __restrict__ kps,
                                                        const uint8_t* __restrict__ desc, const float* __restrict__ uRight,
                                                        const unsigned long long* __restrict__ cand,
                                                        const int* __restrict__ candCnt, int thAccept, int* __restrict__ bestIdx,
                                                        int* __restrict__ bestDist) {
  const int f = blockIdx.y, lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= qCap) return;
  const size_t qo = (size_t)f * qCap + qi;
  const int cnt = candCnt[qo];
  unsigned long long best = ~0ull;
  if (cnt > 0 && cnt <= CAND_CAP) {
    for (int c = lane; c < cnt; c += 64) { const unsigned long long k = cand[qo * CAND_CAP + c]; best = k < best ? k : best; }
  } else if (cnt > CAND_CAP) {   // dense window: derive the keys again from the features
    const Query q = qs[qo];
    int cx0, cx1, cy0, cy1;
    cell_range(P, q, cx0, cx1, cy0, cy1);
    const int img = fImg[f], N = count[img];
    const Desc qd = load_desc(qDesc + qo * 32);
    const float* ur = uRight ? uRight + (size_t)f * cap : nullptr;
    const int jLo = q.jHi > 0 ? q.jLo : 0, jHi = q.jHi > 0 ? min(q.jHi, N) : N;
    for (int j = jLo + lane; j < jHi; j += 64) {
      const unsigned long long k = make_key(P, q, qd, kps[(size_t)img * cap + j], j, desc + ((size_t)img * cap + j) * 32, ur, cx0, cx1, cy0, cy1);
      best = k < best ? k : best;
    }
  }
  best = morbwave::min_u64(best);   // DPP reduction, all lanes active
  if (lane == 0) {
    const bool ok = best != ~0ull && (int)(best >> 32) <= thAccept;
    bestIdx[qo] = ok ? (int)((best >> 4) & 0xFFFF) : -1;
    if (bestDist) bestDist[qo] = ok ? (int)(best >> 32) : -1;
  }
}

// SearchBySim3's agreement check (:1501-1516): match12[i1] = idx2 iff vnMatch1[i1] == idx2 and vnMatch2[idx2] == i1
__global__ __launch_bounds__(256) void k_sim3_agree(int cap, const int* __restrict__ vn1, const int* __restrict__ vn2,
                                                    int* __restrict__ match12, int* __restrict__ nFound) {
  const int f = blockIdx.y, i1 = blockIdx.x * 256 + threadIdx.x;
  int hit = 0;
  if (i1 < cap) {
    const int idx2 = vn1[(size_t)f * cap + i1];
    int r = -1;
    if (idx2 >= 0 && vn2[(size_t)f * cap + idx2] == i1) { r = idx2; hit = 1; }
    match12[(size_t)f * cap + i1] = r;
  }
  const int n = __syncthreads_count(hit);
  if (threadIdx.x == 0 && n) atomicAdd(&nFound[f], n);
}

// ---------------------------------------------------------------------------------------------------
// phase A: candidate keys per query (one wave per query, lanes over the frame's features)
__global__ __launch_bounds__(256) void k_candidates(morb_frame_params P, int qCap, const Query* __restrict__ qs,
                                                    const uint8_t* __restrict__ qDesc, const int* __restrict__ fImg, int cap,
                                                    const int* __restrict__ count, const morb_keypoint* __restrict__ kps,
                                                    const uint8_t* __restrict__ desc, const float* __restrict__ uRight,
                                                    unsigned long long* __restrict__ cand, int* __restrict__ candCnt,
                                                    int qShift) {   // query descriptor row = query index >> qShift (fisheye: 2 queries per map point)
  const int f = blockIdx.y, lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= qCap) return;
  const size_t qo = (size_t)f * qCap + qi;
  const Query q = qs[qo];
  int n = 0;
  int cx0, cx1, cy0, cy1;
  if (q.valid && cell_range(P, q, cx0, cx1, cy0, cy1)) {
    const int img = fImg[f];
    const int N = count[img];
    const Desc qd = load_desc(qDesc + (qo >> qShift) * 32);
    const float* ur = uRight ? uRight + (size_t)f * cap : nullptr;
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const int jLo = q.jHi > 0 ? q.jLo : 0, jHi = q.jHi > 0 ? min(q.jHi, N) : N;
    for (int j0 = jLo; j0 < jHi; j0 += 64) {
      const int j = j0 + lane;
      unsigned long long k = ~0ull;
      if (j < jHi) k = make_key(P, q, qd, kps[(size_t)img * cap + j], j, desc + ((size_t)img * cap + j) * 32, ur, cx0, cx1, cy0, cy1);
      const uint64_t m = __ballot(k != ~0ull);
      if (k != ~0ull) {
        const int slot = n + __popcll(m & lt);
        if (slot < CAND_CAP) cand[qo * CAND_CAP + slot] = k;
      }
      n += __popcll(m);
    }
  }
  if (lane == 0) candCnt[qo] = n;
}

// phase B: replay the queries in reference order, one wave per frame.
// MODE 0: best only, accept dist <= thAccept (SearchByProjection with the last frame :1617 / with a keyframe :1805)
// MODE 1: best + second best with the level rule and the ratio test (SearchByProjection with map points :118-137)
// MODE 2: SearchForInitialization (:603-700): candidates already matched with a smaller-or-equal distance are
//         skipped, a better match steals the feature from its previous owner
// MODE 3: MODE 1 on a fisheye rig (F.Nleft != -1, :42-209): queries 2i / 2i + 1 are map point i in the left / right
//         camera; a match also claims the stereo partner (mvLeftToRightMatch / mvRightToLeftMatch), and a left match
//         rejected by the ratio test skips the right pass of that map point (the `continue` at :122)
// MODE 4: MODE 0 on a fisheye current frame (:1521-1733): queries 2i / 2i + 1 = left / right pass of last-frame feature
//         i; an empty left window skips the right pass (the `continue` at :1583)
template <int MODE>
__global__ __launch_bounds__(64) void k_resolve(morb_frame_params P, int qCap, const int* __restrict__ nQv,
                                                const Query* __restrict__ qs, const uint8_t* __restrict__ qDesc,
                                                const uint8_t* __restrict__ qHasObs, const int* __restrict__ fImg, int cap,
                                                const int* __restrict__ count, const morb_keypoint* __restrict__ kps,
                                                const uint8_t* __restrict__ desc, const float* __restrict__ uRight,
                                                const uint8_t* __restrict__ blockedIn,
                                                const unsigned long long* __restrict__ cand, const int* __restrict__ candCnt,
                                                float nnratio, int thAccept, int checkOri, int* __restrict__ match,
                                                int* __restrict__ nmatches, int* __restrict__ entryJ, int* __restrict__ entryBin,
                                                float* __restrict__ prevMatched, const int* __restrict__ l2r,
                                                const int* __restrict__ r2l, const int* __restrict__ nLeftv) {
  constexpr int QS = (MODE == 3 || MODE == 4) ? 1 : 0;   // query index -> descriptor / hasObs row
  extern __shared__ __align__(8) uint8_t smemRaw[];
  uint8_t* blocked = smemRaw;                                   // MODE 0/1: [cap]
  int* matchedDist = reinterpret_cast<int*>(smemRaw);           // MODE 2:   [cap]
  int* match21 = reinterpret_cast<int*>(smemRaw) + cap;         // MODE 2:   [cap]
  __shared__ int hist[HISTO_LENGTH];
  const int f = blockIdx.x, lane = threadIdx.x;
  const int img = fImg[f];
  const int N = count[img];
  const int nQ = nQv ? nQv[f] : count[fImg[f]];
  if (MODE == 2) {
    for (int j = lane; j < N; j += 64) { matchedDist[j] = 0x7fffffff; match21[j] = -1; }
  } else {
    for (int j = lane; j < N; j += 64) blocked[j] = blockedIn ? blockedIn[(size_t)f * cap + j] : 0;
  }
  if (lane < HISTO_LENGTH) hist[lane] = 0;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const float* ur = uRight ? uRight + (size_t)f * cap : nullptr;
  int* mF = match + (size_t)f * (MODE == 2 ? qCap : cap);
  int nm = 0, nEntries = 0;
  const float factor = 1.0f / HISTO_LENGTH;
  const int nLeft = MODE == 3 ? nLeftv[f] : 0;
  int skipRightOf = -1;
  const int nQtot = QS ? 2 * nQ : nQ;
  for (int qi = 0; qi < nQtot && qi < qCap; ++qi) {
    const size_t qo = (size_t)f * qCap + qi;
    if (MODE == 3 && (qi & 1) && (qi >> 1) == skipRightOf) continue;
    if (MODE == 4 && (qi & 1) && candCnt[qo - 1] == 0) continue;   // the left pass hit `if (vIndices2.empty()) continue;`
    const int cnt = candCnt[qo];
    if (cnt == 0) continue;
    unsigned long long k1 = ~0ull, k2 = ~0ull;
    if (cnt <= CAND_CAP) {
      for (int c = lane; c < cnt; c += 64) {
        const unsigned long long k = cand[qo * CAND_CAP + c];
        const int j = (int)((k >> 4) & 0xFFFF);
        const bool skip = MODE == 2 ? (matchedDist[j] <= (int)(k >> 32)) : (blocked[j] != 0);
        if (!skip) top2_insert(k1, k2, k);
      }
    } else {  // dense window: derive the keys again from the features
      const Query q = qs[qo];
      int cx0, cx1, cy0, cy1;
      cell_range(P, q, cx0, cx1, cy0, cy1);
      const Desc qd = load_desc(qDesc + (qo >> QS) * 32);
      const int jLo = q.jHi > 0 ? q.jLo : 0, jHi = q.jHi > 0 ? min(q.jHi, N) : N;
      for (int j = jLo + lane; j < jHi; j += 64) {
        if (MODE != 2 && blocked[j]) continue;
        const unsigned long long k = make_key(P, q, qd, kps[(size_t)img * cap + j], j, desc + ((size_t)img * cap + j) * 32, ur, cx0, cx1, cy0, cy1);
        if (k == ~0ull) continue;
        if (MODE == 2 && matchedDist[j] <= (int)(k >> 32)) continue;
        top2_insert(k1, k2, k);
      }
    }
    wave_top2(k1, k2);
    if (k1 == ~0ull) continue;
    const int bestDist = (int)(k1 >> 32), bestIdx = (int)((k1 >> 4) & 0xFFFF);
    bool accept;
    if (MODE == 1 || MODE == 3) {  // ORBmatcher.cc:118-137 / :182-186
      const int bestLevel = (int)(k1 & 15);
      const int bestDist2 = k2 == ~0ull ? 256 : (int)(k2 >> 32), bestLevel2 = k2 == ~0ull ? -1 : (int)(k2 & 15);
      accept = bestDist <= TH_HIGH && !(bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2);
      if (MODE == 3 && !(qi & 1) && bestDist <= TH_HIGH && !accept) skipRightOf = qi >> 1;
    } else if (MODE == 2) {  // :646-647
      const int bestDist2 = k2 == ~0ull ? 0x7fffffff : (int)(k2 >> 32);
      accept = bestDist <= TH_LOW && (float)bestDist < (float)bestDist2 * nnratio;
    } else {
      accept = bestDist <= thAccept;
    }
    if (accept) {
      int stolen = 0;
      if (MODE == 2) stolen = match21[bestIdx] >= 0 ? 1 : 0;
      if (lane == 0) {
        if (MODE == 2) {
          if (match21[bestIdx] >= 0) mF[match21[bestIdx]] = -1;
          mF[qi] = bestIdx;
          match21[bestIdx] = qi;
          matchedDist[bestIdx] = bestDist;
        } else if (MODE == 3) {
          const int mp = qi >> 1;
          const uint8_t ho = qHasObs ? qHasObs[qo >> 1] : 1;
          mF[bestIdx] = mp; blocked[bestIdx] = ho;
          const int partner = (qi & 1) ? r2l[(size_t)f * cap + (bestIdx - nLeft)] : l2r[(size_t)f * cap + bestIdx];
          if (partner != -1) {
            const int pj = (qi & 1) ? partner : partner + nLeft;
            mF[pj] = mp; blocked[pj] = ho;
          }
        } else if (MODE == 4) {
          mF[bestIdx] = qi >> 1;
          blocked[bestIdx] = qHasObs ? qHasObs[qo >> 1] : 1;
        } else {
          mF[bestIdx] = qi;
          blocked[bestIdx] = qHasObs ? qHasObs[qo] : 1;
        }
        if (MODE != 1 && MODE != 3 && checkOri) {
          float rot = qs[qo].angle - kps[(size_t)img * cap + bestIdx].angle;
          if (rot < 0.0f) rot += 360.0f;
          int bin = (int)roundf(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          entryJ[(size_t)f * qCap + nEntries] = MODE == 2 ? qi : bestIdx;
          entryBin[(size_t)f * qCap + nEntries] = bin;
          hist[bin] += 1;
        }
      }
      nm += 1 - stolen;
      if (MODE == 3) {   // the stereo partner counts as a match of its own (:127-131, :189-193)
        const int partner = (qi & 1) ? r2l[(size_t)f * cap + (bestIdx - nLeft)] : l2r[(size_t)f * cap + bestIdx];
        if (partner != -1) ++nm;
      }
      ++nEntries;
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (MODE != 1 && MODE != 3 && checkOri) {  // ComputeThreeMaxima + un-assign
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < HISTO_LENGTH; i++) {
      const int s = hist[i];
      if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
      else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
      else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
    int removed = 0;
    if (MODE == 2) {
      // entries of one query are unique, but a stolen match may already be -1: decrement only live ones (:683-686)
      for (int e = lane; e < nEntries; e += 64) {
        const int b = entryBin[(size_t)f * qCap + e];
        const int i1 = entryJ[(size_t)f * qCap + e];
        if (b != ind1 && b != ind2 && b != ind3 && mF[i1] >= 0) { mF[i1] = -1; ++removed; }
      }
    } else {
      for (int e = lane; e < nEntries; e += 64) {
        const int b = entryBin[(size_t)f * qCap + e];
        if (b != ind1 && b != ind2 && b != ind3) { mF[entryJ[(size_t)f * qCap + e]] = -1; ++removed; }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) removed += __shfl_xor(removed, off, 64);
    nm -= removed;
  }
  if (MODE == 2 && prevMatched) {  // :692-695
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int i1 = lane; i1 < nQ && i1 < qCap; i1 += 64) {
      const int j = mF[i1];
      if (j >= 0) {
        const morb_keypoint kp = kps[(size_t)img * cap + j];
        prevMatched[((size_t)f * qCap + i1) * 2] = kp.x;
        prevMatched[((size_t)f * qCap + i1) * 2 + 1] = kp.y;
      }
    }
  }
  if (lane == 0) nmatches[f] = nm;
}

// ---------------------------------------------------------------------------------------------------
// Round 5: MODE 0 / MODE 1 in ONE launch, one workgroup per frame (replaces k_candidates + k_resolve for the two searches
// Tracking runs on every frame).  Three ideas:
//  (1) the frame's 64 x 48 grid (Frame::AssignFeaturesToGrid, Frame.cc:501-528) is built in LDS as a CSR whose rows are sorted
//      by (cell column, cell row, feature index) — the order GetFeaturesInArea (:742-807) hands features out in — so a query
//      looks at the handful of features in its cell range instead of at every feature of the frame, and a feature's position p
//      in that order replaces (cell, index) in the key: key = dist << 20 | p << 4 | octave fits 32 bits;
//  (2) a thread per query keeps its (short) candidate list, c-major in global memory so that the threads of a wave read
//      neighbouring words; lists longer than SEARCH_CAP are not stored, their query walks the grid again when asked;
//  (3) the order dependence of the reference's loop lives only in "skip features that an EARLIER query gave a map point with
//      observations".  Let blk[p] = 1 + the smallest query index that accepts p and blocks it (0: blocked on entry).  Given blk,
//      every query's result is independent: best / second best among its candidates with blk[p] > q.  Iterate: results from
//      blk, blk from results, until nothing changes.  Query q only reads results of queries < q, so by induction the fixed point
//      is unique and equals the sequential replay; queries below the first one that changed are final and are not recomputed.
//      Typical frames converge in 3 - 5 passes of a few microseconds (the serial replay took 0.7 ms per frame).
constexpr int SEARCH_THREADS = 1024, SEARCH_CAP = 32, GRID_CELLS = GRID_ROWS * GRID_COLS;

__device__ __forceinline__ void top2_insert32(uint32_t& k1, uint32_t& k2, uint32_t k) {
  const bool lt1 = k < k1, lt2 = k < k2;
  k2 = lt1 ? k1 : (lt2 ? k : k2);
  k1 = lt1 ? k : k1;
}

// one candidate of query q: the feature at position p of the grid order; ~0u when a test of the reference's inner loop rejects it.
// The frame's features are staged in LDS in GRID ORDER (x, y, octave | uRight | descriptor), so the walk over a cell run reads
// consecutive LDS rows instead of making a dependent global round trip per candidate (one frame alone: 75 -> ~25 us).
struct SearchFeat { float x, y; int oct; float ur; };   // 16 B
// LDSDESC is a compile-time choice: with a run-time one the compiler merges the LDS and the global descriptor pointer into a generic one and
// every distance becomes two FLAT loads behind an `s_waitcnt vmcnt(0)` — which also waits for the candidate store of the visit before.
// FULL: the tests only the keyframe searches need — a feature-index range (rig keyframes: left | right halves) and Fuse's reprojection gate
template <bool LDSDESC, bool FULL = false>
__device__ __forceinline__ uint32_t search_key(const Query& q, const Desc& qd, int p, const SearchFeat* __restrict__ sf,
                                               const uint16_t* __restrict__ featOf, const uint32_t* __restrict__ sDesc,
                                               const uint8_t* __restrict__ descRow, bool useUr, const float* levelSigma2 = nullptr) {
  // (the walk is bound by instruction issue — sixteen waves share the CU — so the tests are one predicate, not a ladder of branches: ~55
  // instead of ~120 instructions per visit.  The feature-index range of a query (jLo / jHi: fisheye searches only) does not occur here.)
  const SearchFeat ft = sf[p];
  const bool bCheckLevels = (q.minLevel > 0) || (q.maxLevel >= 0);
  bool ok = !bCheckLevels || (ft.oct >= q.minLevel && (q.maxLevel < 0 || ft.oct <= q.maxLevel));
  ok = ok && (fabsf(ft.x - q.x) < q.r) && (fabsf(ft.y - q.y) < q.r);
  ok = ok && !(useUr && ft.ur > 0 && fabsf(q.xr - ft.ur) > q.erMax);
  if (FULL) {
    if (q.jHi > 0) { const int j = featOf[p]; ok = ok && j >= q.jLo && j < q.jHi; }
    if (q.gate == 1) {   // ORBmatcher.cc:1160-1181; mvInvLevelSigma2[l] = 1.0f / mvLevelSigma2[l] (ORBextractor.cc:421)
      const float invS = 1.0f / levelSigma2[ft.oct & 15];
      const float ex = q.x - ft.x, ey = q.y - ft.y;
      if (useUr && ft.ur >= 0) {
        const float er = q.xr - ft.ur;
        const float e2 = ex * ex + ey * ey + er * er;
        ok = ok && !((double)(e2 * invS) > 7.8);
      } else {
        const float e2 = ex * ex + ey * ey;
        ok = ok && !((double)(e2 * invS) > 5.99);
      }
    }
  }
  if (!ok) return ~0u;
  int d;
  if (LDSDESC) {
    const uint4 a = *reinterpret_cast<const uint4*>(sDesc + (size_t)p * 8), b = *reinterpret_cast<const uint4*>(sDesc + (size_t)p * 8 + 4);
    d = __popc(qd.w[0] ^ a.x) + __popc(qd.w[1] ^ a.y) + __popc(qd.w[2] ^ a.z) + __popc(qd.w[3] ^ a.w) + __popc(qd.w[4] ^ b.x) +
        __popc(qd.w[5] ^ b.y) + __popc(qd.w[6] ^ b.z) + __popc(qd.w[7] ^ b.w);
  } else {
    d = hamming(qd, load_desc(descRow + (size_t)featOf[p] * 32));
  }
  return ((uint32_t)d << 20) | ((uint32_t)p << 4) | (uint32_t)(ft.oct & 15);
}

struct SearchLds {   // carved from dynamic LDS
  uint32_t* start;   // [GRID_CELLS + 1] CSR row starts
  uint32_t* cur;     // [GRID_CELLS]     counts, then fill cursors
  uint32_t* blk;     // [capR]           1 + smallest blocking query (0 = blocked on entry, ~0u = free); later: owner
  uint32_t* res;     // [qCapR]          per query: ~0u none, else p | hasObs << 16 (| bin << 17 at the end)
  SearchFeat* feat;  // [capR]           features in grid order
  uint32_t* sDesc;   // [capR][8]        their descriptors (NULL when they do not fit: read from global memory)
  uint16_t* cellOf;  // [capR]           feature -> cell (0xFFFF: outside the grid)
  uint16_t* tmp;     // [capR]           unsorted CSR
  uint16_t* featOf;  // [capR]           position -> feature
  uint32_t* qcnt;    // [qCapR]          candidates of the query (> SEARCH_CAP: walk again)
  uint32_t* qOff;    // [qCapR + 4]      exclusive prefix of the queries' visit counts
  uint32_t* qRange;  // [qCapR]          cx0 | cx1 << 6 | cy0 << 12 | cy1 << 18 | valid << 31
  uint8_t* blk0;     // [capR]           blocked on entry; later: "an entry of this feature fell to the rotation filter"
  uint32_t* blk2;    // [capR]           the fixed point's second copy of blk (the passes alternate)
};
__host__ __device__ inline size_t search_lds_bytes(int cap, int qCap, bool withDesc) {
  const size_t capR = (size_t)(cap + 3) & ~(size_t)3, qCapR = (size_t)(qCap + 3) & ~(size_t)3;
  return 4 * ((size_t)GRID_CELLS + 4) + 4 * (size_t)GRID_CELLS + 4 * capR + 4 * qCapR + 16 * capR + (withDesc ? 32 * capR : 0) + 2 * capR * 3 +
         4 * qCapR + 4 * (qCapR + 4) + 4 * qCapR + capR + 4 * capR;
}

#ifdef MORB_SEARCH_CYCLES   // developer build (tools/ab_build.py): thread 0 of frame 0 adds up where its cycles go
__device__ unsigned long long g_searchCyc[8];
extern "C" int morb_search_cycles(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_searchCyc), sizeof(unsigned long long) * 8); const unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_searchCyc), z, sizeof z); return 0; }
#define SRCH_MARK(slot) do { __syncthreads(); if (f == 0 && tid == 0) { const long long now_ = clock64(); g_searchCyc[slot] += (unsigned long long)(now_ - tMark); tMark = now_; } } while (0)
#define SRCH_CNT(slot) do { if (f == 0 && tid == 0) g_searchCyc[slot] += 1; } while (0)
#else
#define SRCH_MARK(slot)
#define SRCH_CNT(slot)
#endif
template <int MODE, bool LDSDESC>
__global__ __launch_bounds__(SEARCH_THREADS) void k_search(morb_frame_params P, int qCap, const int* __restrict__ nQv,
                                                          const Query* __restrict__ qs, const uint8_t* __restrict__ qDesc,
                                                          const uint8_t* __restrict__ qHasObs, const int* __restrict__ fImg, int cap,
                                                          const int* __restrict__ count, const morb_keypoint* __restrict__ kps,
                                                          const uint8_t* __restrict__ desc, const float* __restrict__ uRight,
                                                          const uint8_t* __restrict__ blockedIn, uint32_t* __restrict__ cand,
                                                          float nnratio, int thAccept, int checkOri, int* __restrict__ match,
                                                          int* __restrict__ nmatches) {
  constexpr int withDesc = LDSDESC ? 1 : 0;
  // MODE 5: every query on its own (Fuse, SearchBySim3: no "already matched" state) — the best key per query by an LDS atomic minimum during the
  // walk, `match` = bestIdx [frames][qCap], `nmatches` = bestDist [frames][qCap] or NULL; no lists, no fixed point
  static_assert(MODE == 0 || MODE == 1 || MODE == 5, "the fisheye / initialisation searches keep the serial replay");
  constexpr bool FULLKEY = MODE == 5;
  extern __shared__ __align__(16) uint8_t smemRaw[];
  __shared__ int hist[HISTO_LENGTH];
  __shared__ int keep3[3];
  __shared__ uint32_t waveTot[SEARCH_THREADS / 64];
  __shared__ int sLo2[2], sAcc, sRem;
  const int f = blockIdx.x, tid = threadIdx.x;
  const int img = fImg[f];
  const int N = min(count[img], cap);
  const int nQ = min(nQv ? nQv[f] : N, qCap);
  const size_t capR = (size_t)(cap + 3) & ~(size_t)3, qCapR = (size_t)(qCap + 3) & ~(size_t)3;
  SearchLds L;
  {
    uint8_t* p = smemRaw;
    L.start = (uint32_t*)p; p += 4 * ((size_t)GRID_CELLS + 4);
    L.cur = (uint32_t*)p; p += 4 * (size_t)GRID_CELLS;
    L.blk = (uint32_t*)p; p += 4 * capR;
    L.res = (uint32_t*)p; p += 4 * qCapR;
    L.feat = (SearchFeat*)p; p += 16 * capR;
    L.sDesc = withDesc ? (uint32_t*)p : nullptr; p += withDesc ? 32 * capR : 0;
    L.cellOf = (uint16_t*)p; p += 2 * capR;
    L.tmp = (uint16_t*)p; p += 2 * capR;
    L.featOf = (uint16_t*)p; p += 2 * capR;
    L.qcnt = (uint32_t*)p; p += 4 * qCapR;
    L.qOff = (uint32_t*)p; p += 4 * (qCapR + 4);
    L.qRange = (uint32_t*)p; p += 4 * qCapR;
    L.blk0 = p; p += capR;
    L.blk2 = (uint32_t*)p;
  }
  const morb_keypoint* kpRow = kps + (size_t)img * cap;
  const uint8_t* descRow = desc + (size_t)img * cap * 32;
  const float* ur = uRight ? uRight + (size_t)f * cap : nullptr;
#ifdef MORB_SEARCH_CYCLES
  long long tMark = clock64();
#endif
  // ---- (1) the grid
  for (int c = tid; c < GRID_CELLS; c += SEARCH_THREADS) L.cur[c] = 0;
  if (tid < HISTO_LENGTH) hist[tid] = 0;
  if (tid == 0) { sAcc = 0; sRem = 0; }
  __syncthreads();
  for (int j = tid; j < N; j += SEARCH_THREADS) {
    const float kx = kpRow[j].x, ky = kpRow[j].y;
    const int posX = (int)roundf((kx - P.minX) * P.gridInvW), posY = (int)roundf((ky - P.minY) * P.gridInvH);
    int c = 0xFFFF;
    if (!(posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS)) { c = posX * GRID_ROWS + posY; atomicAdd(&L.cur[c], 1u); }
    L.cellOf[j] = (uint16_t)c;
  }
  __syncthreads();
  {   // exclusive scan of the 3072 cell counts: three cells per thread, wave scan, wave totals
    const int c0 = tid * 3;
    uint32_t a = 0, b = 0, c = 0;
    if (c0 < GRID_CELLS) { a = L.cur[c0]; b = L.cur[c0 + 1]; c = L.cur[c0 + 2]; }
    const uint32_t mine = a + b + c;
    uint32_t inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off, 64); if ((tid & 63) >= off) inc += o; }
    if ((tid & 63) == 63) waveTot[tid >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += waveTot[w];
    const uint32_t ex = base + inc - mine;
    if (c0 < GRID_CELLS) {
      L.start[c0] = ex; L.start[c0 + 1] = ex + a; L.start[c0 + 2] = ex + a + b;
      L.cur[c0] = ex; L.cur[c0 + 1] = ex + a; L.cur[c0 + 2] = ex + a + b;
      if (c0 + 3 == GRID_CELLS) L.start[GRID_CELLS] = ex + mine;
    }
  }
  __syncthreads();
  const int M = (int)L.start[GRID_CELLS];   // features inside the grid
  for (int j = tid; j < N; j += SEARCH_THREADS) {
    const int c = L.cellOf[j];
    if (c != 0xFFFF) L.tmp[atomicAdd(&L.cur[c], 1u)] = (uint16_t)j;
  }
  __syncthreads();
  for (int p = tid; p < M; p += SEARCH_THREADS) {   // sort each (short) row by feature index: position = start + #smaller
    const int j = L.tmp[p], c = L.cellOf[j];
    const int s = (int)L.start[c], e = (int)L.start[c + 1];
    int r = s;
    for (int t = s; t < e; ++t) r += (int)L.tmp[t] < j ? 1 : 0;
    L.featOf[r] = (uint16_t)j;
    L.blk0[r] = blockedIn ? (blockedIn[(size_t)f * cap + j] != 0) : 0;
    SearchFeat ft;
    ft.x = kpRow[j].x; ft.y = kpRow[j].y; ft.oct = kpRow[j].octave; ft.ur = ur ? ur[j] : -1.0f;
    L.feat[r] = ft;
  }
  __syncthreads();
  if (LDSDESC) {   // descriptors in grid order: eight lanes per feature, a dword each
    for (int t = tid; t < M * 8; t += SEARCH_THREADS) {
      const int p = t >> 3, wd = t & 7;
      L.sDesc[t] = reinterpret_cast<const uint32_t*>(descRow + (size_t)L.featOf[p] * 32)[wd];
    }
    __syncthreads();
  }
  SRCH_MARK(0);
  // ---- (2) candidate lists.  A query visits the features of its cell range — 3 on average, 20+ for a wide window at a coarse level — so with a
  // thread per query a wave ran as long as its widest window (a quarter of the lanes' slots did work).  Instead: every query's visit count from the
  // CSR, a prefix sum over the queries, and the VISITS dealt evenly to the threads; a thread walks its share (usually inside one or two queries) and
  // appends what passes the tests to the query's list (LDS counter, c-major global rows).
  uint32_t* candF = cand + (size_t)f * SEARCH_CAP * qCap;
  auto for_each_candidate = [&](const Query& q, const Desc& qd, auto&& fn) {   // (the whole walk of one query: lists longer than SEARCH_CAP)
    int cx0, cx1, cy0, cy1;
    if (!cell_range(P, q, cx0, cx1, cy0, cy1)) return;
    for (int cx = cx0; cx <= cx1; ++cx) {
      const int p0 = (int)L.start[cx * GRID_ROWS + cy0], p1 = (int)L.start[cx * GRID_ROWS + cy1 + 1];
      for (int p = p0; p < p1; ++p) {
        const uint32_t k = search_key<LDSDESC, FULLKEY>(q, qd, p, L.feat, L.featOf, L.sDesc, descRow, ur != nullptr, P.levelSigma2);
        if (k != ~0u) fn(k);
      }
    }
  };
  for (int qi = tid; qi < nQ; qi += SEARCH_THREADS) {
    const Query q = qs[(size_t)f * qCap + qi];
    uint32_t rg = 0, w = 0;
    int cx0, cx1, cy0, cy1;
    if (q.valid && cell_range(P, q, cx0, cx1, cy0, cy1)) {
      // GetFeaturesInArea's cell range (floor / ceil of the window's ends) holds about twice the cells a feature with |dx| < r, |dy| < r can lie
      // in: a feature's cell is round((k - min) * inv), monotone in k, so only cells round((c -+ (r + margin) - min) * inv) can pass the window
      // test that follows anyway.  The margin (0.01 px + 1e-4 r) is far above the rounding of the float expressions involved.
      const float mr = q.r * 1.0001f + 0.01f;
      cx0 = max(cx0, (int)roundf((q.x - mr - P.minX) * P.gridInvW)); cx1 = min(cx1, (int)roundf((q.x + mr - P.minX) * P.gridInvW));
      cy0 = max(cy0, (int)roundf((q.y - mr - P.minY) * P.gridInvH)); cy1 = min(cy1, (int)roundf((q.y + mr - P.minY) * P.gridInvH));
      if (cx0 <= cx1 && cy0 <= cy1) {
        rg = (uint32_t)cx0 | ((uint32_t)cx1 << 6) | ((uint32_t)cy0 << 12) | ((uint32_t)cy1 << 18) | 0x80000000u;
        for (int cx = cx0; cx <= cx1; ++cx) w += L.start[cx * GRID_ROWS + cy1 + 1] - L.start[cx * GRID_ROWS + cy0];
      }
    }
    L.qRange[qi] = rg;
    L.qOff[qi] = w;
    L.qcnt[qi] = 0;
    L.res[qi] = ~0u;
  }
  __syncthreads();
  {   // exclusive scan of the visit counts (in place): thread t owns queries [t K, (t + 1) K)
    const int K = (nQ + SEARCH_THREADS - 1) / SEARCH_THREADS;
    const int b = min(tid * K, nQ), e = min(b + K, nQ);
    uint32_t mine = 0;
    for (int i = b; i < e; ++i) mine += L.qOff[i];
    uint32_t inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off, 64); if ((tid & 63) >= off) inc += o; }
    __syncthreads();                       // (waveTot was read by the grid's scan)
    if ((tid & 63) == 63) waveTot[tid >> 6] = inc;
    __syncthreads();
    uint32_t run = inc - mine;
    for (int w = 0; w < (tid >> 6); ++w) run += waveTot[w];
    for (int i = b; i < e; ++i) { const uint32_t w = L.qOff[i]; L.qOff[i] = run; run += w; }
    if (tid == SEARCH_THREADS - 1) L.qOff[nQ] = run;
  }
  __syncthreads();
  {
    const int W = (int)L.qOff[nQ];
#ifdef MORB_SEARCH_CYCLES
#endif
    const int per = (W + SEARCH_THREADS - 1) / SEARCH_THREADS;
    int item = min(tid * per, W);
    const int end = min(item + per, W);
#ifdef MORB_SEARCH_CYCLES
    long long tW = clock64(); unsigned long long cBs = 0, cLd = 0, cIn = 0;
#define WK(acc) do { const long long n_ = clock64(); acc += (unsigned long long)(n_ - tW); tW = n_; } while (0)
#else
#define WK(acc)
#endif
    if (item < end) {
      int lo_ = 0, hi_ = nQ;               // the query that holds visit `item`: the last one with qOff <= item
      while (hi_ - lo_ > 1) { const int mid = (lo_ + hi_) >> 1; if ((int)L.qOff[mid] <= item) lo_ = mid; else hi_ = mid; }
      int qi = lo_;
      WK(cBs);
      while (item < end) {
        while ((int)L.qOff[qi + 1] <= item) ++qi;            // (queries without visits)
        const size_t qo = (size_t)f * qCap + qi;
        const Query q = qs[qo];
        const Desc qd = load_desc(qDesc + qo * 32);
        const uint32_t rg = L.qRange[qi];
        const int cx1 = (int)((rg >> 6) & 63), cy0 = (int)((rg >> 12) & 63), cy1 = (int)((rg >> 18) & 63);
        int cx = (int)(rg & 63);
        int local = item - (int)L.qOff[qi];
        const int qEnd = min(end, (int)L.qOff[qi + 1]);
        int p0 = (int)L.start[cx * GRID_ROWS + cy0], p1 = (int)L.start[cx * GRID_ROWS + cy1 + 1];
        while (local >= p1 - p0) { local -= p1 - p0; ++cx; p0 = (int)L.start[cx * GRID_ROWS + cy0]; p1 = (int)L.start[cx * GRID_ROWS + cy1 + 1]; }
        int p = p0 + local;
#ifdef MORB_SEARCH_CYCLES
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        WK(cLd);
        while (item < qEnd) {
          const uint32_t k = search_key<LDSDESC, FULLKEY>(q, qd, p, L.feat, L.featOf, L.sDesc, descRow, ur != nullptr, P.levelSigma2);
          if (k != ~0u) {
            if (MODE == 5) atomicMin(&L.res[qi], k);
            else {
              const uint32_t slot = atomicAdd(&L.qcnt[qi], 1u);
              if (slot < (uint32_t)SEARCH_CAP) candF[(size_t)slot * qCap + qi] = k;
            }
          }
          ++item; ++p;
          while (p == p1 && cx < cx1) { ++cx; p = (int)L.start[cx * GRID_ROWS + cy0]; p1 = (int)L.start[cx * GRID_ROWS + cy1 + 1]; }
        }
        ++qi;
        WK(cIn);
      }
    }
#ifdef MORB_SEARCH_CYCLES
    if (f == 0 && tid == 0) { g_searchCyc[5] += cBs; g_searchCyc[6] += cLd; g_searchCyc[7] += cIn; }
#endif
  }
  __threadfence_block();
  __syncthreads();
  if (MODE == 5) {
    int* bestIdx = match + (size_t)f * qCap;
    int* bestDist = nmatches ? nmatches + (size_t)f * qCap : nullptr;
    for (int qi = tid; qi < qCap; qi += SEARCH_THREADS) {
      const uint32_t r = qi < nQ ? L.res[qi] : ~0u;
      const bool ok = r != ~0u && (int)(r >> 20) <= thAccept;
      bestIdx[qi] = ok ? (int)L.featOf[(r >> 4) & 0xFFFFu] : -1;
      if (bestDist) bestDist[qi] = ok ? (int)(r >> 20) : -1;
    }
    return;
  }
  SRCH_MARK(1);
  // ---- (3) results <-> blk until nothing changes.  The owner of a query (thread qi % 1024, slot qi / 1024) keeps the first SEARCH_REG keys of its
  // list in registers: a pass then touches LDS only (the lists were re-read from L2 in every pass: a dependent round trip per key).
  constexpr int SEARCH_QPT = 2, SEARCH_REG = 4;
  uint32_t kreg[SEARCH_QPT][SEARCH_REG];
  uint32_t ownHo = 0;    // bit s: hasObs of slot s's query
#pragma unroll
  for (int sl = 0; sl < SEARCH_QPT; ++sl) {
    const int qi = tid + sl * SEARCH_THREADS;
#pragma unroll
    for (int c = 0; c < SEARCH_REG; ++c) kreg[sl][c] = ~0u;
    if (qi < nQ) {
      const int n = (int)min(L.qcnt[qi], (uint32_t)SEARCH_CAP);
#pragma unroll
      for (int c = 0; c < SEARCH_REG; ++c) if (c < n) kreg[sl][c] = candF[(size_t)c * qCap + qi];
      if (qHasObs ? (qHasObs[(size_t)f * qCap + qi] != 0) : true) ownHo |= 1u << sl;
    }
  }
  auto decide = [&](int qi, uint32_t k1, uint32_t k2, bool ho) -> uint32_t {
    if (k1 == ~0u) return ~0u;
    const int bestDist = (int)(k1 >> 20);
    bool accept;
    if (MODE == 1) {   // ORBmatcher.cc:118-137
      const int bestLevel = (int)(k1 & 15);
      const int bestDist2 = k2 == ~0u ? 256 : (int)(k2 >> 20), bestLevel2 = k2 == ~0u ? -1 : (int)(k2 & 15);
      accept = bestDist <= TH_HIGH && !(bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2);
    } else {
      accept = bestDist <= thAccept;
    }
    return accept ? (((k1 >> 4) & 0xFFFFu) | ((ho ? 1u : 0u) << 16)) : ~0u;
  };
  // Two barriers per pass: blk and the "first changed query" word exist twice and the passes alternate — while a pass recomputes the results from its
  // copy it also resets the other one for the next pass (four barriers per pass before: reset | scatter | recompute | read the word).
  int lo = 0, pb = 0;
  for (int p = tid; p < M; p += SEARCH_THREADS) L.blk[p] = L.blk0[p] ? 0u : ~0u;
  if (tid == 0) { sLo2[0] = 0x7fffffff; sLo2[1] = 0x7fffffff; }
  __syncthreads();
  for (;;) {
    SRCH_CNT(4);
    uint32_t* const B = pb ? L.blk2 : L.blk;
    uint32_t* const Bn = pb ? L.blk : L.blk2;
    for (int qi = tid; qi < nQ; qi += SEARCH_THREADS) {
      const uint32_t r = L.res[qi];
      if (r != ~0u && (r & 0x10000u)) atomicMin(&B[r & 0xFFFFu], (uint32_t)qi + 1u);
    }
    __syncthreads();
    int changed = 0x7fffffff;
#pragma unroll
    for (int sl = 0; sl < SEARCH_QPT; ++sl) {
      const int qi = tid + sl * SEARCH_THREADS;
      if (qi < lo || qi >= nQ) continue;
      const int n = (int)L.qcnt[qi];
      if (n == 0) continue;
      uint32_t k1 = ~0u, k2 = ~0u;
      if (n <= SEARCH_CAP) {
#pragma unroll
        for (int c = 0; c < SEARCH_REG; ++c) {
          const uint32_t k = kreg[sl][c];
          if (k != ~0u && B[(k >> 4) & 0xFFFFu] > (uint32_t)qi) top2_insert32(k1, k2, k);
        }
        for (int c = SEARCH_REG; c < n; ++c) {
          const uint32_t k = candF[(size_t)c * qCap + qi];
          if (B[(k >> 4) & 0xFFFFu] > (uint32_t)qi) top2_insert32(k1, k2, k);
        }
      } else {
        const size_t qo = (size_t)f * qCap + qi;
        const Query q = qs[qo];
        const Desc qd = load_desc(qDesc + qo * 32);
        for_each_candidate(q, qd, [&](uint32_t k) { if (B[(k >> 4) & 0xFFFFu] > (uint32_t)qi) top2_insert32(k1, k2, k); });
      }
      const uint32_t r = decide(qi, k1, k2, (ownHo >> sl) & 1u);
      if (r != L.res[qi]) { L.res[qi] = r; changed = min(changed, qi); }
    }
    for (int qi = max(lo, SEARCH_QPT * SEARCH_THREADS) + tid; qi < nQ; qi += SEARCH_THREADS) {   // (more than 2048 queries: no register slot)
      const int n = (int)L.qcnt[qi];
      if (n == 0) continue;
      uint32_t k1 = ~0u, k2 = ~0u;
      if (n <= SEARCH_CAP) {
        for (int c = 0; c < n; ++c) {
          const uint32_t k = candF[(size_t)c * qCap + qi];
          if (B[(k >> 4) & 0xFFFFu] > (uint32_t)qi) top2_insert32(k1, k2, k);
        }
      } else {
        const size_t qo = (size_t)f * qCap + qi;
        const Query q = qs[qo];
        const Desc qd = load_desc(qDesc + qo * 32);
        for_each_candidate(q, qd, [&](uint32_t k) { if (B[(k >> 4) & 0xFFFFu] > (uint32_t)qi) top2_insert32(k1, k2, k); });
      }
      const uint32_t r = decide(qi, k1, k2, qHasObs ? (qHasObs[(size_t)f * qCap + qi] != 0) : true);
      if (r != L.res[qi]) { L.res[qi] = r; changed = min(changed, qi); }
    }
    if (changed != 0x7fffffff) atomicMin(&sLo2[pb], changed);
    for (int p = tid; p < M; p += SEARCH_THREADS) Bn[p] = L.blk0[p] ? 0u : ~0u;   // (the next pass's copy: nobody reads it in this pass)
    if (tid == 0) sLo2[pb ^ 1] = 0x7fffffff;                                        // (read last behind the barrier before this pass's scatter)
    __syncthreads();
    const int first = sLo2[pb];
    if (first == 0x7fffffff) break;
    lo = first + 1;      // queries up to the first change are final (their inputs are results of queries below them)
    pb ^= 1;
  }
  SRCH_MARK(2);
  // ---- outputs: last accepted query per feature, rotation filter (MODE 0), counts
  for (int p = tid; p < M; p += SEARCH_THREADS) { L.blk[p] = 0u; L.blk0[p] = 0; }
  __syncthreads();
  int acc = 0;
  for (int qi = tid; qi < nQ; qi += SEARCH_THREADS) {
    uint32_t r = L.res[qi];
    if (r == ~0u) continue;
    ++acc;
    const int p = (int)(r & 0xFFFFu);
    atomicMax(&L.blk[p], (uint32_t)qi + 1u);
    if (MODE == 0 && checkOri) {
      float rot = qs[(size_t)f * qCap + qi].angle - kpRow[L.featOf[p]].angle;
      if (rot < 0.0f) rot += 360.0f;
      int bin = (int)roundf(rot * (1.0f / HISTO_LENGTH));
      if (bin == HISTO_LENGTH) bin = 0;
      atomicAdd(&hist[bin], 1);
      L.res[qi] = r | ((uint32_t)bin << 17);
    }
  }
  if (acc) atomicAdd(&sAcc, acc);
  __syncthreads();
  if (MODE == 0 && checkOri) {
    if (tid == 0) {   // ComputeThreeMaxima (:1844-1876)
      int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
      for (int i = 0; i < HISTO_LENGTH; i++) {
        const int s = hist[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
      }
      if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
      else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
      keep3[0] = ind1; keep3[1] = ind2; keep3[2] = ind3;
    }
    __syncthreads();
    int rem = 0;
    for (int qi = tid; qi < nQ; qi += SEARCH_THREADS) {
      const uint32_t r = L.res[qi];
      if (r == ~0u) continue;
      const int b = (int)((r >> 17) & 31);
      if (b != keep3[0] && b != keep3[1] && b != keep3[2]) { L.blk0[r & 0xFFFFu] = 1; ++rem; }   // every entry counts (:1722-1728)
    }
    if (rem) atomicAdd(&sRem, rem);
    __syncthreads();
  }
  int* mF = match + (size_t)f * cap;
  for (int p = tid; p < M; p += SEARCH_THREADS) {
    const uint32_t o = L.blk[p];
    if (o) mF[L.featOf[p]] = L.blk0[p] ? -1 : (int)o - 1;
  }
  if (tid == 0) nmatches[f] = sAcc - sRem;
  SRCH_MARK(3);
}

// query preparation for SearchByProjection(CurrentFrame, KeyFrame*, ...) (:1735-1790) and SearchForInitialization
__global__ __launch_bounds__(256) void k_prep_kf(morb_frame_params P, int cap, const int* __restrict__ count,
                                                 const int* __restrict__ kfImg, const morb_keypoint* __restrict__ kps,
                                                 const uint8_t* __restrict__ kfValid, const float* __restrict__ Xw,
                                                 const float* __restrict__ maxDist, const float* __restrict__ minDist,
                                                 const float* __restrict__ Tcw, const float* __restrict__ Ow, float th,
                                                 const float* __restrict__ ratioThr, Query* __restrict__ qs,
                                                 // KannalaBrandt8 rig frame (NULL: pinhole): mpCamera's fx fy cx cy k0..k3, and the search looks among the
                                                 // current frame's LEFT features [0, nLeftCur) only (GetFeaturesInArea's bRight = false)
                                                 const float* __restrict__ kb8, const int* __restrict__ nLeftCur) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  const size_t o = (size_t)f * cap + i;
  Query q;
  memset(&q, 0, sizeof q);
  const int img = kfImg[f];
  if (i < count[img] && kfValid[o]) {
    const float* T = Tcw + 7 * f;
    const float* X = Xw + o * 3;
    float x3Dc[3];
    q_rotate_f(T, X, x3Dc);
    x3Dc[0] += T[4]; x3Dc[1] += T[5]; x3Dc[2] += T[6];
    float u, v;
    if (kb8) kb8_project_dev(kb8, x3Dc, u, v);
    else { u = P.fx * x3Dc[0] / x3Dc[2] + P.cx; v = P.fy * x3Dc[1] / x3Dc[2] + P.cy; }
    const float PO[3] = {X[0] - Ow[3 * f], X[1] - Ow[3 * f + 1], X[2] - Ow[3 * f + 2]};
    const float dist3D = sqrtf(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    const float maxDistance = 1.2f * maxDist[o], minDistance = 0.8f * minDist[o];
    if (!(u < P.minX || u > P.maxX) && !(v < P.minY || v > P.maxY) && !(dist3D < minDistance || dist3D > maxDistance)) {
      const float ratio = maxDist[o] / dist3D;
      int n = 0;
      while (n < P.nlevels - 1 && ratio > ratioThr[n]) ++n;
      q.valid = 1; q.x = u; q.y = v; q.r = th * P.scaleFactors[n];
      q.minLevel = n - 1; q.maxLevel = n + 1; q.angle = kps[(size_t)img * cap + i].angle;
      if (nLeftCur) { q.jLo = 0; q.jHi = nLeftCur[f]; q.valid = q.jHi > 0 ? 1 : 0; }
    }
  }
  qs[o] = q;
}
__global__ __launch_bounds__(256) void k_prep_init(int cap, const int* __restrict__ count, const int* __restrict__ img1v,
                                                   const morb_keypoint* __restrict__ kps, const float* __restrict__ prevMatched,
                                                   int windowSize, Query* __restrict__ qs) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cap) return;
  const size_t o = (size_t)f * cap + i;
  Query q;
  memset(&q, 0, sizeof q);
  const int img = img1v[f];
  if (i < count[img]) {
    const morb_keypoint kp = kps[(size_t)img * cap + i];
    if (!(kp.octave > 0)) {
      q.valid = 1; q.x = prevMatched[o * 2]; q.y = prevMatched[o * 2 + 1]; q.r = (float)windowSize;
      q.minLevel = kp.octave; q.maxLevel = kp.octave; q.angle = kp.angle;
    }
  }
  qs[o] = q;
}

__global__ void k_gather_desc(const uint8_t* __restrict__ desc, const int* __restrict__ img, int cap, uint8_t* __restrict__ out) {
  const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;   // one uint4 (16 B) per thread
  if (i >= cap * 2) return;
  reinterpret_cast<uint4*>(out)[(size_t)f * cap * 2 + i] = reinterpret_cast<const uint4*>(desc)[(size_t)img[f] * cap * 2 + i];
}
__global__ void k_fill_m1(int* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = -1;
}
__global__ void k_gather_counts(const int* __restrict__ count, const int* __restrict__ img, int n, int* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = count[img[i]];
}

// ---------------------------------------------------------------------------------------------------
// M4: SearchForTriangulation.  vbMatched2 is never set in this fork, so every KF1 feature is independent:
// one wave per KF1 feature, lanes over the KF2 features of the same BoW node (sorted lists from k_bow_sort),
// minimum of (dist, -position): the reference's "dist <= bestDist replaces" keeps the LAST equal candidate.
__global__ __launch_bounds__(256) void k_triangulation(const unsigned long long* __restrict__ sorted, int cap,
                                                       const int* __restrict__ count, const int* __restrict__ img1v,
                                                       const int* __restrict__ img2v, const morb_keypoint* __restrict__ kps,
                                                       const uint8_t* __restrict__ desc, const int* __restrict__ node,
                                                       const uint8_t* __restrict__ hasMP, const float* __restrict__ uRight,
                                                       morb_frame_params P, const float* __restrict__ F12v,
                                                       const float* __restrict__ epv, int bOnlyStereo, int bCoarse,
                                                       int* __restrict__ match12, int* __restrict__ bin12,
                                                       // KannalaBrandt8 rig (NULL = pinhole): [camL8 | camR8 | per pair Tll Tlr Trl Trr as R, t]
                                                       const float* __restrict__ rig, const int* __restrict__ nLeft1v,
                                                       const int* __restrict__ nLeft2v) {
  const int pair = blockIdx.y, lane = threadIdx.x & 63;
  const int idx1 = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int i1 = img1v[pair], i2 = img2v[pair];
  if (idx1 >= cap) return;
  int result = -1, bin = -1;
  const size_t o1 = (size_t)i1 * cap + idx1;
  const int nd = idx1 < count[i1] ? node[o1] : -1;
  const bool fish = rig != nullptr;
  const bool bStereo1 = !fish && uRight && uRight[o1] >= 0;   // (!pKF1->mpCamera2 && mvuRight >= 0), :884
  if (nd >= 0 && !hasMP[o1] && !(bOnlyStereo && !bStereo1)) {
    const unsigned long long* s2 = sorted + (size_t)i2 * cap;
    const int n2 = count[i2];
    int lo = 0, hi = n2;
    const unsigned long long target = (unsigned long long)(unsigned)nd << 32;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s2[mid] < target) lo = mid + 1; else hi = mid; }
    const morb_keypoint kp1 = kps[o1];
    const Desc d1 = load_desc(desc + o1 * 32);
    const float* F12 = F12v + 9 * pair;
    const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
    const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
    const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
    const float den = a * a + b * b;
    unsigned long long best = ~0ull;
    for (int s0 = lo; s0 < n2; s0 += 64) {
      const unsigned long long first = s2[s0];
      if ((unsigned)(first >> 32) != (unsigned)nd) break;  // wave-uniform: s0 is uniform
      const int s = s0 + lane;
      if (s < n2) {
        const unsigned long long k2 = s2[s];
        if ((unsigned)(k2 >> 32) == (unsigned)nd) {
          const int idx2 = (int)(k2 & 0xFFFFFFFFu);
          const size_t o2 = (size_t)i2 * cap + idx2;
          const bool bStereo2 = !fish && uRight && uRight[o2] >= 0;
          if (!hasMP[o2] && !(bOnlyStereo && !bStereo2)) {
            const int dist = hamming(d1, load_desc(desc + o2 * 32));
            if (dist <= TH_LOW) {
              const morb_keypoint kp2 = kps[o2];
              bool ok = true;
              if (fish) {
                // camera pair and relative pose follow the sides of the two features (:934-966); the constraint is
                // KannalaBrandt8::epipolarConstrain = TriangulateMatches(...) > 0.0001 (KannalaBrandt8.cpp:307-321)
                if (!bCoarse) {
                  const bool bRight1 = !(idx1 < nLeft1v[pair]), bRight2 = !(idx2 < nLeft2v[pair]);
                  const float* T = rig + 16 + (size_t)pair * 48 + 12 * ((bRight1 ? 2 : 0) + (bRight2 ? 1 : 0));
                  morbkb8::KB8 c1, c2;
                  for (int q = 0; q < 8; ++q) { c1.p[q] = rig[(bRight1 ? 8 : 0) + q]; c2.p[q] = rig[(bRight2 ? 8 : 0) + q]; }
                  float p3D[3];
                  ok = morbkb8::triangulate_matches(c1, c2, T, T + 9, kp1.x, kp1.y, kp2.x, kp2.y, P.levelSigma2[kp1.octave],
                                                    P.levelSigma2[kp2.octave], p3D) > 0.0001f;
                }
              } else if (!bStereo1 && !bStereo2) {
                const float distex = epv[2 * pair] - kp2.x, distey = epv[2 * pair + 1] - kp2.y;
                if (distex * distex + distey * distey < 100 * P.scaleFactors[kp2.octave]) ok = false;
              }
              if (ok && !bCoarse && !fish) {  // Pinhole::epipolarConstrain (Pinhole.cpp:111-139)
                const float num = a * kp2.x + b * kp2.y + c;
                if (den == 0) ok = false;
                else { const float dsqr = num * num / den; ok = (double)dsqr < 3.84 * (double)P.levelSigma2[kp2.octave]; }
              }
              if (ok) {
                const unsigned long long k = ((unsigned long long)dist << 32) | (unsigned)(0x7FFFFFFF - (s - lo));
                best = k < best ? k : best;
              }
            }
          }
        }
      }
    }
    best = morbwave::min_u64(best);   // DPP reduction, all lanes active
    if (best != ~0ull) {
      const int pos = lo + (0x7FFFFFFF - (int)(best & 0xFFFFFFFFu));
      result = (int)(s2[pos] & 0xFFFFFFFFu);
      float rot = kp1.angle - kps[(size_t)i2 * cap + result].angle;
      if (rot < 0.0f) rot += 360.0f;
      bin = (int)roundf(rot * (1.0f / HISTO_LENGTH));
      if (bin == HISTO_LENGTH) bin = 0;
    }
  }
  if (lane == 0) { match12[(size_t)pair * cap + idx1] = result; bin12[(size_t)pair * cap + idx1] = bin; }
}

}  // namespace

// =====================================================================================================
extern "C" {

// largest float ratio whose PredictScale level is <= n, using the host libm logf (monotone): bisection on the
// float bit pattern
static float ratio_threshold(int n, float logScaleFactor) {
  auto lvl = [&](float r) { return (int)std::ceil(std::log(r) / logScaleFactor); };
  uint32_t lo = 0x00800000u, hi = 0x7F7FFFFFu;  // smallest normal .. FLT_MAX
  auto asf = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
  if (lvl(asf(lo)) > n) return 0.f;
  while (hi - lo > 1) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (lvl(asf(mid)) <= n) lo = mid; else hi = mid;
  }
  return asf(lo);
}

// thresholds of all levels of a pyramid (~500 logf calls): kept per thread for the pyramid last asked about
static void level_thresholds(const morb_frame_params* P, float* thr16) {
  thread_local int cn = -1; thread_local float cl = 0.f; thread_local float ct[16];
  if (cn != P->nlevels || cl != P->logScaleFactor) {
    for (int n = 0; n < 16; ++n) ct[n] = n < P->nlevels - 1 ? ratio_threshold(n, P->logScaleFactor) : 3.4e38f;
    cn = P->nlevels; cl = P->logScaleFactor;
  }
  memcpy(thr16, ct, sizeof ct);
}

static int frustum_impl(morb_matcher* m, const morb_frame_params* P, const float* cam8, int nframes, const float* d_Rcw,
                        const float* d_tcw, const float* d_Ow, int mpCap, const int* d_nMP, const float* d_Pw,
                        const float* d_normal, const float* d_maxDist, const float* d_minDist, float viewingCosLimit,
                        uint8_t* d_inView, float* d_projX, float* d_projY, float* d_projXR, float* d_depth, int* d_level,
                        float* d_viewCos, void* stream) {
  MORB_REQUIRE(m && P && d_Rcw && d_tcw && d_Ow && d_nMP && d_Pw && d_normal && d_maxDist && d_minDist && d_inView && d_projX &&
                   d_projY && d_projXR && d_depth && d_level && d_viewCos, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && mpCap > 0 && P->nlevels >= 1 && P->nlevels <= 16, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  float thr[24];   // 16 level thresholds + the 8 camera parameters of the KB8 variant
  level_thresholds(P, thr);
  for (int n = 0; n < 8; ++n) thr[16 + n] = cam8 ? cam8[n] : 0.f;
  void* d_thr = nullptr;
  int rc = morb_matcher_const(m, cam8 ? 1 : 0, thr, sizeof thr, &d_thr, st);   // uploaded once per camera, not per call
  if (rc != MORB_OK) return rc;
  const float* d_kb8 = cam8 ? (const float*)d_thr + 16 : nullptr;
  hipLaunchKernelGGL(k_frustum, dim3(div_up(mpCap, 256), nframes), dim3(256), 0, st, *P, d_Rcw, d_tcw, d_Ow, mpCap, d_nMP, d_Pw,
                     d_normal, d_maxDist, d_minDist, viewingCosLimit, (const float*)d_thr, (const float*)d_kb8, d_inView, d_projX,
                     d_projY, d_projXR, d_depth, d_level, d_viewCos);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_is_in_frustum_batch(morb_matcher* m, const morb_frame_params* P, int nframes, const float* d_Rcw, const float* d_tcw,
                             const float* d_Ow, int mpCap, const int* d_nMP, const float* d_Pw, const float* d_normal,
                             const float* d_maxDist, const float* d_minDist, float viewingCosLimit, uint8_t* d_inView,
                             float* d_projX, float* d_projY, float* d_projXR, float* d_depth, int* d_level,
                             float* d_viewCos, void* stream) {
  return frustum_impl(m, P, nullptr, nframes, d_Rcw, d_tcw, d_Ow, mpCap, d_nMP, d_Pw, d_normal, d_maxDist, d_minDist,
                      viewingCosLimit, d_inView, d_projX, d_projY, d_projXR, d_depth, d_level, d_viewCos, stream);
}

int morb_is_in_frustum_kb8_batch(morb_matcher* m, const morb_frame_params* P, const float* cam8, int nframes, const float* d_R,
                                 const float* d_t, const float* d_twc, int mpCap, const int* d_nMP, const float* d_Pw,
                                 const float* d_normal, const float* d_maxDist, const float* d_minDist, float viewingCosLimit,
                                 uint8_t* d_inView, float* d_projX, float* d_projY, float* d_depth, int* d_level,
                                 float* d_viewCos, void* stream) {
  MORB_REQUIRE(cam8, MORB_ERR_INVALID, "NULL camera");
  void* xr = nullptr;   // the pinhole-only mTrackProjXR slot of the shared kernel
  int rc = morb_matcher_workspace(m, 7, sizeof(float) * (size_t)nframes * mpCap, &xr);
  if (rc != MORB_OK) return rc;
  return frustum_impl(m, P, cam8, nframes, d_R, d_t, d_twc, mpCap, d_nMP, d_Pw, d_normal, d_maxDist, d_minDist, viewingCosLimit,
                      d_inView, d_projX, d_projY, (float*)xr, d_depth, d_level, d_viewCos, stream);
}

static int window_search(morb_matcher* m, const morb_frame_params* P, int mode, int nframes, int qCap, const int* d_nQ,
                         const Query* d_qs, const uint8_t* d_qDesc, const uint8_t* d_qHasObs, const int* d_fImg, int cap,
                         const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc, const float* d_uRight,
                         const uint8_t* d_blocked, float nnratio, int thAccept, int checkOri, int* d_match, int* d_nmatches,
                         float* d_prevMatched, hipStream_t st, const int* d_l2r = nullptr, const int* d_r2l = nullptr,
                         const int* d_nLeft = nullptr, bool ranged = false) {   // ranged: queries carry a feature-index range (jLo / jHi)
  void *cand = nullptr, *cnt = nullptr, *ej = nullptr, *eb = nullptr;
  if ((mode == 0 || mode == 1) && !ranged && cap <= 65535 && qCap <= 65535 && search_lds_bytes(cap, qCap, false) <= 150 * 1024 && !getenv("MORB_SERIAL_RESOLVE")) {
    // one launch, one workgroup per frame: grid in LDS, candidate lists, blocked-feature fixed point (k_search)
    const int withDesc = search_lds_bytes(cap, qCap, true) <= 150 * 1024 ? 1 : 0;
    const size_t lds = search_lds_bytes(cap, qCap, withDesc != 0);
    int rc = morb_matcher_workspace(m, 0, sizeof(uint32_t) * (size_t)nframes * qCap * SEARCH_CAP, &cand);
    if (rc != MORB_OK) return rc;
#define MORB_SEARCH(MODE, WD)                                                                                                              \
  do {                                                                                                                                     \
    MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_search<MODE, WD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_search<MODE, WD>), dim3(nframes), dim3(SEARCH_THREADS), lds, st, *P, qCap, d_nQ, d_qs, d_qDesc, d_qHasObs, d_fImg, cap, \
                       d_count, d_kps, d_desc, d_uRight, d_blocked, (uint32_t*)cand, nnratio, thAccept, checkOri, d_match, d_nmatches);         \
  } while (0)
    if (mode == 0) { if (withDesc) MORB_SEARCH(0, true); else MORB_SEARCH(0, false); }
    else { if (withDesc) MORB_SEARCH(1, true); else MORB_SEARCH(1, false); }
#undef MORB_SEARCH
    MORB_HIP_CHECK(hipGetLastError());
    return MORB_OK;
  }
  int rc = morb_matcher_workspace(m, 0, sizeof(unsigned long long) * (size_t)nframes * qCap * CAND_CAP, &cand);
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 1, sizeof(int) * (size_t)nframes * qCap, &cnt);
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 2, sizeof(int) * (size_t)nframes * qCap, &ej);
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 3, sizeof(int) * (size_t)nframes * qCap, &eb);
  if (rc != MORB_OK) return rc;
  hipLaunchKernelGGL(k_candidates, dim3(div_up(qCap, 4), nframes), dim3(256), 0, st, *P, qCap, d_qs, d_qDesc, d_fImg, cap, d_count,
                     d_kps, d_desc, d_uRight, (unsigned long long*)cand, (int*)cnt, (mode == 3 || mode == 4) ? 1 : 0);
#define MORB_RESOLVE(MODE, SMEM)                                                                                             \
  hipLaunchKernelGGL(k_resolve<MODE>, dim3(nframes), dim3(64), (SMEM), st, *P, qCap, d_nQ, d_qs, d_qDesc, d_qHasObs, d_fImg, cap, \
                     d_count, d_kps, d_desc, d_uRight, d_blocked, (const unsigned long long*)cand, (const int*)cnt, nnratio,   \
                     thAccept, checkOri, d_match, d_nmatches, (int*)ej, (int*)eb, d_prevMatched, d_l2r, d_r2l, d_nLeft)
  if (mode == 1) MORB_RESOLVE(1, (size_t)cap);
  else if (mode == 3) MORB_RESOLVE(3, (size_t)cap);
  else if (mode == 4) MORB_RESOLVE(4, (size_t)cap);
  else if (mode == 2) {
    MORB_REQUIRE((size_t)cap * 8 <= 64 * 1024, MORB_ERR_UNSUPPORTED, "too many features for SearchForInitialization's LDS state");
    MORB_RESOLVE(2, (size_t)cap * 8);
  } else MORB_RESOLVE(0, (size_t)cap);
#undef MORB_RESOLVE
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

int morb_search_by_projection_mps_batch(morb_matcher* m, const morb_frame_params* P, int nframes, const int* d_fImg, int cap,
                                        const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc,
                                        const float* d_uRight, const uint8_t* d_blocked, int mpCap, const int* d_nMP,
                                        const uint8_t* d_inView, const uint8_t* d_isBad, const float* d_depth,
                                        const float* d_projX, const float* d_projY, const float* d_projXR, const int* d_level,
                                        const float* d_viewCos, const uint8_t* d_mpDesc, const uint8_t* d_mpHasObs, float th,
                                        int bFarPoints, float thFarPoints, float nnratio, int* d_matchF, int* d_nmatches,
                                        void* stream) {
  MORB_REQUIRE(m && P && d_fImg && d_count && d_kps && d_desc && d_nMP && d_inView && d_isBad && d_depth && d_projX && d_projY &&
                   d_projXR && d_level && d_viewCos && d_mpDesc && d_mpHasObs && d_matchF && d_nmatches, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && cap > 0 && cap <= 65535 && mpCap > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  void* qs = nullptr;
  int rc = morb_matcher_workspace(m, 5, sizeof(Query) * (size_t)nframes * mpCap, &qs);
  if (rc != MORB_OK) return rc;
  hipLaunchKernelGGL(k_prep_mps, dim3(div_up(mpCap, 256), nframes), dim3(256), 0, st, *P, mpCap, d_nMP, d_inView, d_isBad, d_depth,
                     d_projX, d_projY, d_projXR, d_level, d_viewCos, th, bFarPoints, thFarPoints, (Query*)qs);
  return window_search(m, P, 1, nframes, mpCap, d_nMP, (const Query*)qs, d_mpDesc, d_mpHasObs, d_fImg, cap, d_count, d_kps,
                       d_desc, d_uRight, d_blocked, nnratio, TH_HIGH, 0, d_matchF, d_nmatches, nullptr, st);
}

int morb_search_by_projection_mps_fisheye_batch(morb_matcher* m, const morb_frame_params* P, int nframes, const int* d_fImg,
                                                int cap, const int* d_count, const int* d_nLeft, const morb_keypoint* d_kps,
                                                const uint8_t* d_desc, const int* d_l2r, const int* d_r2l,
                                                const uint8_t* d_blocked, int mpCap, const int* d_nMP,
                                                const uint8_t* d_inViewL, const uint8_t* d_inViewR, const uint8_t* d_isBad,
                                                const float* d_depthL, const float* d_projXL, const float* d_projYL,
                                                const int* d_levelL, const float* d_viewCosL, const float* d_projXR,
                                                const float* d_projYR, const int* d_levelR, const float* d_viewCosR,
                                                const uint8_t* d_mpDesc, const uint8_t* d_mpHasObs, float th, int bFarPoints,
                                                float thFarPoints, float nnratio, int* d_matchF, int* d_nmatches, void* stream) {
  MORB_REQUIRE(m && P && d_fImg && d_count && d_nLeft && d_kps && d_desc && d_l2r && d_r2l && d_nMP && d_inViewL && d_inViewR &&
                   d_isBad && d_depthL && d_projXL && d_projYL && d_levelL && d_viewCosL && d_projXR && d_projYR && d_levelR &&
                   d_viewCosR && d_mpDesc && d_mpHasObs && d_matchF && d_nmatches, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && cap > 0 && cap <= 65535 && mpCap > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  void* qs = nullptr;
  int rc = morb_matcher_workspace(m, 5, sizeof(Query) * (size_t)nframes * mpCap * 2, &qs);
  if (rc != MORB_OK) return rc;
  hipLaunchKernelGGL(k_prep_mps_fisheye, dim3(div_up(mpCap, 256), nframes), dim3(256), 0, st, *P, mpCap, d_nMP, d_fImg, d_count,
                     d_nLeft, d_inViewL, d_inViewR, d_isBad, d_depthL, d_projXL, d_projYL, d_levelL, d_viewCosL, d_projXR, d_projYR,
                     d_levelR, d_viewCosR, th, bFarPoints, thFarPoints, (Query*)qs);
  // 2 queries per map point; the resolve pass (mode 3) walks them as (left, right) pairs in map-point order
  return window_search(m, P, 3, nframes, 2 * mpCap, d_nMP, (const Query*)qs, d_mpDesc, d_mpHasObs, d_fImg, cap, d_count, d_kps,
                       d_desc, nullptr, d_blocked, nnratio, TH_HIGH, 0, d_matchF, d_nmatches, nullptr, st, d_l2r, d_r2l, d_nLeft);
}

int morb_search_by_projection_last_batch(morb_matcher* m, const morb_frame_params* P, int nframes, const int* d_curImg,
                                         const int* d_lastImg, int cap, const int* d_count, const morb_keypoint* d_kps,
                                         const uint8_t* d_desc, const float* d_curURight, const uint8_t* d_curBlocked,
                                         const float* d_Tcw, const uint8_t* d_lastValid, const float* d_lastXw,
                                         const uint8_t* d_lastMPdesc, const uint8_t* d_lastMPhasObs, float th,
                                         const uint8_t* d_bForward, const uint8_t* d_bBackward, int checkOri, int* d_matchCur,
                                         int* d_nmatches, void* stream) {
  MORB_REQUIRE(m && P && d_curImg && d_lastImg && d_count && d_kps && d_desc && d_Tcw && d_lastValid && d_lastXw && d_lastMPdesc &&
                   d_lastMPhasObs && d_bForward && d_bBackward && d_matchCur && d_nmatches, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && cap > 0 && cap <= 65535, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  void* qs = nullptr;
  int rc = morb_matcher_workspace(m, 5, sizeof(Query) * (size_t)nframes * cap, &qs);
  if (rc != MORB_OK) return rc;
  // the number of queries of frame f is the feature count of its LAST image (written by the same launch)
  void* nq = nullptr;
  rc = morb_matcher_workspace(m, 6, sizeof(int) * (size_t)nframes, &nq);
  if (rc != MORB_OK) return rc;
  hipLaunchKernelGGL(k_prep_last, dim3(div_up(cap, 256), nframes), dim3(256), 0, st, *P, cap, d_count, d_lastImg, d_kps, d_lastValid,
                     d_lastXw, d_Tcw, th, d_bForward, d_bBackward, (Query*)qs, (int*)nq);
  return window_search(m, P, 0, nframes, cap, (const int*)nq, (const Query*)qs, d_lastMPdesc, d_lastMPhasObs, d_curImg, cap,
                       d_count, d_kps, d_desc, d_curURight, d_curBlocked, 0.f, TH_HIGH, checkOri, d_matchCur, d_nmatches, nullptr, st);
}

int morb_search_by_projection_last_fisheye_batch(morb_matcher* m, const morb_frame_params* P, const float* cam8, const float* Trl7,
                                                 int nframes, const int* d_curImg, const int* d_lastImg, const int* d_nLeftCur,
                                                 int cap, const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc,
                                                 const uint8_t* d_curBlocked, const float* d_Tcw, const uint8_t* d_lastValid,
                                                 const float* d_lastXw, const uint8_t* d_lastMPdesc,
                                                 const uint8_t* d_lastMPhasObs, float th, const uint8_t* d_bForward,
                                                 const uint8_t* d_bBackward, int checkOri, int* d_matchCur, int* d_nmatches,
                                                 void* stream) {
  MORB_REQUIRE(m && P && cam8 && Trl7 && d_curImg && d_lastImg && d_nLeftCur && d_count && d_kps && d_desc && d_Tcw && d_lastValid &&
                   d_lastXw && d_lastMPdesc && d_lastMPhasObs && d_bForward && d_bBackward && d_matchCur && d_nmatches,
               MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && cap > 0 && cap <= 65535, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  float ct[15];
  for (int i = 0; i < 8; ++i) ct[i] = cam8[i];
  for (int i = 0; i < 7; ++i) ct[8 + i] = Trl7[i];
  void* d_ct = nullptr;
  int rc = morb_matcher_const(m, 2, ct, sizeof ct, &d_ct, st);
  if (rc != MORB_OK) return rc;
  void* qs = nullptr;
  rc = morb_matcher_workspace(m, 5, sizeof(Query) * (size_t)nframes * cap * 2, &qs);
  if (rc != MORB_OK) return rc;
  hipLaunchKernelGGL(k_prep_last_fisheye, dim3(div_up(cap, 256), nframes), dim3(256), 0, st, *P, cap, d_count, d_lastImg, d_curImg,
                     d_nLeftCur, d_kps, d_lastValid, d_lastXw, d_Tcw, (const float*)d_ct, th, d_bForward, d_bBackward, (Query*)qs);
  void* nq = nullptr;
  rc = morb_matcher_workspace(m, 6, sizeof(int) * (size_t)nframes, &nq);
  if (rc != MORB_OK) return rc;
  hipLaunchKernelGGL(k_gather_counts, dim3(div_up(nframes, 256)), dim3(256), 0, st, d_count, d_lastImg, nframes, (int*)nq);
  return window_search(m, P, 4, nframes, 2 * cap, (const int*)nq, (const Query*)qs, d_lastMPdesc, d_lastMPhasObs, d_curImg, cap,
                       d_count, d_kps, d_desc, nullptr, d_curBlocked, 0.f, TH_HIGH, checkOri, d_matchCur, d_nmatches, nullptr, st);
}

static int search_by_projection_kf_impl(morb_matcher* m, const morb_frame_params* P, const float* cam8, const int* d_nLeftCur, int nframes,
                                        const int* d_curImg, const int* d_kfImg, int cap, const int* d_count, const morb_keypoint* d_kps,
                                        const uint8_t* d_desc, const uint8_t* d_curHasMP, const float* d_Tcw, const float* d_Ow,
                                        const uint8_t* d_kfValid, const float* d_Xw, const float* d_maxDist,
                                        const float* d_minDist, const uint8_t* d_mpDesc, float th, int ORBdist, int checkOri,
                                        int* d_matchCur, int* d_nmatches, void* stream) {
  MORB_REQUIRE(m && P && d_curImg && d_kfImg && d_count && d_kps && d_desc && d_Tcw && d_Ow && d_kfValid && d_Xw && d_maxDist &&
                   d_minDist && d_mpDesc && d_matchCur && d_nmatches, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nframes > 0 && cap > 0 && cap <= 65535, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  float thr[24] = {0};   // (same table as isInFrustum's, so the two share constant slots 0 / 1)
  level_thresholds(P, thr);
  for (int n = 0; n < 8; ++n) thr[16 + n] = cam8 ? cam8[n] : 0.f;
  void *d_thr = nullptr, *qs = nullptr, *nq = nullptr;
  int rc = morb_matcher_const(m, cam8 ? 1 : 0, thr, sizeof thr, &d_thr, st);
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 5, sizeof(Query) * (size_t)nframes * cap, &qs);
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 6, sizeof(int) * (size_t)nframes, &nq);
  if (rc != MORB_OK) return rc;
  hipLaunchKernelGGL(k_prep_kf, dim3(div_up(cap, 256), nframes), dim3(256), 0, st, *P, cap, d_count, d_kfImg, d_kps, d_kfValid, d_Xw,
                     d_maxDist, d_minDist, d_Tcw, d_Ow, th, (const float*)d_thr, (Query*)qs, cam8 ? (const float*)d_thr + 16 : (const float*)nullptr,
                     d_nLeftCur);
  hipLaunchKernelGGL(k_gather_counts, dim3(div_up(nframes, 256)), dim3(256), 0, st, d_count, d_kfImg, nframes, (int*)nq);
  return window_search(m, P, 0, nframes, cap, (const int*)nq, (const Query*)qs, d_mpDesc, nullptr, d_curImg, cap, d_count, d_kps,
                       d_desc, nullptr, d_curHasMP, 0.f, ORBdist, checkOri, d_matchCur, d_nmatches, nullptr, st, nullptr, nullptr, nullptr,
                       d_nLeftCur != nullptr);
}

int morb_search_by_projection_kf_batch(morb_matcher* m, const morb_frame_params* P, int nframes, const int* d_curImg,
                                       const int* d_kfImg, int cap, const int* d_count, const morb_keypoint* d_kps,
                                       const uint8_t* d_desc, const uint8_t* d_curHasMP, const float* d_Tcw, const float* d_Ow,
                                       const uint8_t* d_kfValid, const float* d_Xw, const float* d_maxDist,
                                       const float* d_minDist, const uint8_t* d_mpDesc, float th, int ORBdist, int checkOri,
                                       int* d_matchCur, int* d_nmatches, void* stream) {
  return search_by_projection_kf_impl(m, P, nullptr, nullptr, nframes, d_curImg, d_kfImg, cap, d_count, d_kps, d_desc, d_curHasMP, d_Tcw, d_Ow,
                                      d_kfValid, d_Xw, d_maxDist, d_minDist, d_mpDesc, th, ORBdist, checkOri, d_matchCur, d_nmatches, stream);
}

int morb_search_by_projection_kf_rig_batch(morb_matcher* m, const morb_frame_params* P, const float* cam8, int nframes, const int* d_curImg,
                                           const int* d_kfImg, const int* d_nLeftCur, int cap, const int* d_count,
                                           const morb_keypoint* d_kps, const uint8_t* d_desc, const uint8_t* d_curHasMP, const float* d_Tcw,
                                           const float* d_Ow, const uint8_t* d_kfValid, const float* d_Xw, const float* d_maxDist,
                                           const float* d_minDist, const uint8_t* d_mpDesc, float th, int ORBdist, int checkOri,
                                           int* d_matchCur, int* d_nmatches, void* stream) {
  MORB_REQUIRE(cam8 && d_nLeftCur, MORB_ERR_INVALID, "NULL rig argument");
  return search_by_projection_kf_impl(m, P, cam8, d_nLeftCur, nframes, d_curImg, d_kfImg, cap, d_count, d_kps, d_desc, d_curHasMP, d_Tcw, d_Ow,
                                      d_kfValid, d_Xw, d_maxDist, d_minDist, d_mpDesc, th, ORBdist, checkOri, d_matchCur, d_nmatches, stream);
}

int morb_search_for_initialization_batch(morb_matcher* m, const morb_frame_params* P, int npairs, const int* d_img1,
                                         const int* d_img2, int cap, const int* d_count, const morb_keypoint* d_kps,
                                         const uint8_t* d_desc, float* d_prevMatched, int windowSize, float nnratio,
                                         int checkOri, int* d_matches12, int* d_nmatches, void* stream) {
  MORB_REQUIRE(m && P && d_img1 && d_img2 && d_count && d_kps && d_desc && d_prevMatched && d_matches12 && d_nmatches,
               MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(npairs > 0 && cap > 0 && cap <= 65535, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  void *qs = nullptr, *nq = nullptr, *qd = nullptr;
  int rc = morb_matcher_workspace(m, 5, sizeof(Query) * (size_t)npairs * cap, &qs);
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 6, sizeof(int) * (size_t)npairs, &nq);
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 7, (size_t)npairs * cap * 32, &qd);
  if (rc != MORB_OK) return rc;
  hipLaunchKernelGGL(k_prep_init, dim3(div_up(cap, 256), npairs), dim3(256), 0, st, cap, d_count, d_img1, d_kps, d_prevMatched,
                     windowSize, (Query*)qs);
  hipLaunchKernelGGL(k_gather_counts, dim3(div_up(npairs, 256)), dim3(256), 0, st, d_count, d_img1, npairs, (int*)nq);
  // query descriptors = F1's descriptor rows, gathered per pair
  hipLaunchKernelGGL(k_gather_desc, dim3(div_up(cap * 2, 256), npairs), dim3(256), 0, st, d_desc, d_img1, cap, (uint8_t*)qd);
  hipLaunchKernelGGL(k_fill_m1, dim3(div_up(npairs * cap, 256)), dim3(256), 0, st, d_matches12, npairs * cap);
  return window_search(m, P, 2, npairs, cap, (const int*)nq, (const Query*)qs, (const uint8_t*)qd, nullptr, d_img2, cap, d_count,
                       d_kps, d_desc, nullptr, nullptr, nnratio, TH_LOW, checkOri, d_matches12, d_nmatches, d_prevMatched, st);
}

}  // extern "C"


extern "C" {

// F12 = K1^-T [t12]x R12 K2^-1 in float, products left to right (Pinhole.cpp:118-122)
static void fundamental_f12(const float* K1, const float* K2, const float* R12, const float* t12, float* F12) {
  const float k1it[9] = {1.f / K1[0], 0, 0, 0, 1.f / K1[1], 0, -K1[2] / K1[0], -K1[3] / K1[1], 1.f};
  const float k2i[9] = {1.f / K2[0], 0, -K2[2] / K2[0], 0, 1.f / K2[1], -K2[3] / K2[1], 0, 0, 1.f};
  const float tx[9] = {0, -t12[2], t12[1], t12[2], 0, -t12[0], -t12[1], t12[0], 0};
  auto mul = [](const float* A, const float* B, float* C) {
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) C[i * 3 + j] = (A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j]) + A[i * 3 + 2] * B[6 + j];
  };
  float m1[9], m2[9];
  mul(k1it, tx, m1);
  mul(m1, R12, m2);
  mul(m2, k2i, F12);
}

}

namespace {
__global__ __launch_bounds__(256) void k_rot_filter12(const int* __restrict__ count, const int* __restrict__ img1v, int cap,
                                                      int checkOri, int* __restrict__ match12, const int* __restrict__ bin12,
                                                      int* __restrict__ nmatches) {
  __shared__ int hist[HISTO_LENGTH];
  __shared__ int keep[3];
  __shared__ int total;
  const int pair = blockIdx.x, tid = threadIdx.x;
  const int n1 = count[img1v[pair]];
  int* mm = match12 + (size_t)pair * cap;
  const int* bb = bin12 + (size_t)pair * cap;
  if (tid < HISTO_LENGTH) hist[tid] = 0;
  if (tid == 0) total = 0;
  __syncthreads();
  for (int i = tid; i < n1; i += 256)
    if (mm[i] >= 0) { atomicAdd(&total, 1); if (checkOri) atomicAdd(&hist[bb[i]], 1); }
  __syncthreads();
  if (checkOri) {
    if (tid == 0) {
      int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
      for (int i = 0; i < HISTO_LENGTH; i++) {
        const int s = hist[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
      }
      if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
      else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
      keep[0] = ind1; keep[1] = ind2; keep[2] = ind3;
    }
    __syncthreads();
    for (int i = tid; i < n1; i += 256)
      if (mm[i] >= 0) { const int b = bb[i]; if (b != keep[0] && b != keep[1] && b != keep[2]) { mm[i] = -1; atomicSub(&total, 1); } }
    __syncthreads();
  }
  if (tid == 0) nmatches[pair] = total;
}
}  // namespace

// ---- M7 host side ---------------------------------------------------------------------------------------------------
static int upload_ratio_thresholds(morb_matcher* m, const morb_frame_params* P, const float* cam8, hipStream_t st, const float** d_thr,
                                   const float** d_kb8) {
  float thr[24];
  level_thresholds(P, thr);
  for (int n = 0; n < 8; ++n) thr[16 + n] = cam8 ? cam8[n] : 0.f;
  void* d = nullptr;
  int rc = morb_matcher_const(m, cam8 ? 1 : 0, thr, sizeof thr, &d, st);
  if (rc != MORB_OK) return rc;
  *d_thr = (const float*)d;
  *d_kb8 = cam8 ? (const float*)d + 16 : nullptr;
  return MORB_OK;
}

// queries -> candidate keys (workspaces 0 / 1 / 5)
static int kfproj_candidates(morb_matcher* m, const morb_frame_params* P, int nprob, const int* d_kfImg, int cap, const int* d_count,
                             const morb_keypoint* d_kps, const uint8_t* d_desc, const float* d_uRight, int mpCap, const int* d_nMP,
                             const uint8_t* d_valid, const float* d_Pw, const float* d_normal, const float* d_maxDist,
                             const float* d_minDist, const uint8_t* d_mpDesc, const float* d_T, const float* d_sim, const float* d_Ow,
                             const float* cam8, const int* d_jLo, const int* d_jHi, float th, int projMode, int gate, hipStream_t st,
                             const Query** qsOut, const unsigned long long** candOut, const int** cntOut) {   // candOut == NULL: the queries only
  const float *d_thr = nullptr, *d_kb8 = nullptr;
  int rc = upload_ratio_thresholds(m, P, cam8, st, &d_thr, &d_kb8);
  if (rc != MORB_OK) return rc;
  void *qs = nullptr, *cand = nullptr, *cnt = nullptr;
  rc = morb_matcher_workspace(m, 5, sizeof(Query) * (size_t)nprob * mpCap, &qs);
  if (rc == MORB_OK && candOut) rc = morb_matcher_workspace(m, 0, sizeof(unsigned long long) * (size_t)nprob * mpCap * CAND_CAP, &cand);
  if (rc == MORB_OK && candOut) rc = morb_matcher_workspace(m, 1, sizeof(int) * (size_t)nprob * mpCap, &cnt);
  if (rc != MORB_OK) return rc;
  hipLaunchKernelGGL(k_prep_kfproj, dim3(div_up(mpCap, 256), nprob), dim3(256), 0, st, *P, mpCap, d_nMP, d_valid, d_Pw, d_normal,
                     d_maxDist, d_minDist, d_T, d_sim, d_Ow, d_thr, d_kb8, d_jLo, d_jHi, th, projMode, gate, (Query*)qs);
  *qsOut = (const Query*)qs;
  if (!candOut) return MORB_OK;
  hipLaunchKernelGGL(k_candidates, dim3(div_up(mpCap, 4), nprob), dim3(256), 0, st, *P, mpCap, (const Query*)qs, d_mpDesc, d_kfImg, cap,
                     d_count, d_kps, d_desc, d_uRight, (unsigned long long*)cand, (int*)cnt, 0);
  *candOut = (const unsigned long long*)cand; *cntOut = (const int*)cnt;
  return MORB_OK;
}

// best feature of every (independent) query: one launch of k_search<5> when the frame's tables fit LDS, else k_candidates + k_best_per_query
static int best_per_query(morb_matcher* m, const morb_frame_params* P, int nprob, int qCap, const int* d_nQ, const Query* qs, const uint8_t* d_qDesc,
                          const int* d_kfImg, int cap, const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc, const float* d_uRight,
                          int thAccept, int* d_bestIdx, int* d_bestDist, hipStream_t st) {
  if (cap <= 65535 && qCap <= 65535 && search_lds_bytes(cap, qCap, false) <= 150 * 1024 && !getenv("MORB_SERIAL_RESOLVE")) {
    const bool withDesc = search_lds_bytes(cap, qCap, true) <= 150 * 1024;
    const size_t lds = search_lds_bytes(cap, qCap, withDesc);
    if (withDesc) {
      MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_search<5, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((k_search<5, true>), dim3(nprob), dim3(SEARCH_THREADS), lds, st, *P, qCap, d_nQ, qs, d_qDesc, (const uint8_t*)nullptr, d_kfImg, cap,
                         d_count, d_kps, d_desc, d_uRight, (const uint8_t*)nullptr, (uint32_t*)nullptr, 0.f, thAccept, 0, d_bestIdx, d_bestDist);
    } else {
      MORB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_search<5, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((k_search<5, false>), dim3(nprob), dim3(SEARCH_THREADS), lds, st, *P, qCap, d_nQ, qs, d_qDesc, (const uint8_t*)nullptr, d_kfImg, cap,
                         d_count, d_kps, d_desc, d_uRight, (const uint8_t*)nullptr, (uint32_t*)nullptr, 0.f, thAccept, 0, d_bestIdx, d_bestDist);
    }
    MORB_HIP_CHECK(hipGetLastError());
    return MORB_OK;
  }
  void *cand = nullptr, *cnt = nullptr;
  int rc = morb_matcher_workspace(m, 0, sizeof(unsigned long long) * (size_t)nprob * qCap * CAND_CAP, &cand);
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 1, sizeof(int) * (size_t)nprob * qCap, &cnt);
  if (rc != MORB_OK) return rc;
  hipLaunchKernelGGL(k_candidates, dim3(div_up(qCap, 4), nprob), dim3(256), 0, st, *P, qCap, qs, d_qDesc, d_kfImg, cap, d_count, d_kps, d_desc, d_uRight,
                     (unsigned long long*)cand, (int*)cnt, 0);
  hipLaunchKernelGGL(k_best_per_query, dim3(div_up(qCap, 4), nprob), dim3(256), 0, st, *P, qCap, qs, d_qDesc, d_kfImg, cap, d_count, d_kps, d_desc, d_uRight,
                     (const unsigned long long*)cand, (const int*)cnt, thAccept, d_bestIdx, d_bestDist);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

extern "C" int morb_fuse_batch(morb_matcher* m, const morb_frame_params* P, int nprob, const int* d_kfImg, int cap, const int* d_count,
                               const morb_keypoint* d_kps, const uint8_t* d_desc, const float* d_uRight, const float* d_Tcw,
                               const float* d_Ow, const float* cam8, const int* d_jLo, const int* d_jHi, int mpCap, const int* d_nMP,
                               const uint8_t* d_valid, const float* d_Pw, const float* d_normal, const float* d_maxDist,
                               const float* d_minDist, const uint8_t* d_mpDesc, float th, int sim3Form, int* d_bestIdx,
                               int* d_bestDist, void* stream) {
  MORB_REQUIRE(m && P && d_kfImg && d_count && d_kps && d_desc && d_Tcw && d_Ow && d_nMP && d_valid && d_Pw && d_normal && d_maxDist &&
                   d_minDist && d_mpDesc && d_bestIdx, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nprob > 0 && cap > 0 && cap <= 65535 && mpCap > 0 && P->nlevels >= 1 && P->nlevels <= 16, MORB_ERR_INVALID, "bad sizes");
  MORB_REQUIRE((d_jLo == nullptr) == (d_jHi == nullptr), MORB_ERR_INVALID, "feature range needs both ends");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  const Query* qs; const unsigned long long* cand; const int* cnt;
  int rc = kfproj_candidates(m, P, nprob, d_kfImg, cap, d_count, d_kps, d_desc, d_uRight, mpCap, d_nMP, d_valid, d_Pw, d_normal,
                             d_maxDist, d_minDist, d_mpDesc, d_Tcw, nullptr, d_Ow, cam8, d_jLo, d_jHi, th, 0, sim3Form ? 0 : 1, st, &qs,
                             nullptr, nullptr);
  if (rc != MORB_OK) return rc;
  return best_per_query(m, P, nprob, mpCap, d_nMP, qs, d_mpDesc, d_kfImg, cap, d_count, d_kps, d_desc, d_uRight, TH_LOW, d_bestIdx, d_bestDist, st);
}

// cam8 / d_nLeft: a KannalaBrandt8 rig keyframe — the reference then searches the LEFT camera's features only (GetFeaturesInArea with bRight = false,
// mvKeysUn = mvKeys) and projects with mpCamera->project (the first form, cam8) or with the pinhole formula on pKF->fx ... (the twin, cam8 == NULL)
static int search_by_projection_sim3_impl(morb_matcher* m, const morb_frame_params* P, int nprob, const int* d_kfImg, int cap,
                                          const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc,
                                          const float* d_Tcw, const float* d_Ow, int mpCap, const int* d_nMP,
                                          const uint8_t* d_valid, const float* d_Pw, const float* d_normal,
                                          const float* d_maxDist, const float* d_minDist, const uint8_t* d_mpDesc,
                                          const uint8_t* d_matched, int th, float ratioHamming, int manualProjection,
                                          const float* cam8, const int* d_nLeft, int* d_matchF, int* d_nmatches, void* stream) {
  MORB_REQUIRE(m && P && d_kfImg && d_count && d_kps && d_desc && d_Tcw && d_Ow && d_nMP && d_valid && d_Pw && d_normal && d_maxDist &&
                   d_minDist && d_mpDesc && d_matched && d_matchF && d_nmatches, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(nprob > 0 && cap > 0 && cap <= 65535 && mpCap > 0 && P->nlevels >= 1 && P->nlevels <= 16, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  const Query* qs; const unsigned long long* cand; const int* cnt;
  int rc = kfproj_candidates(m, P, nprob, d_kfImg, cap, d_count, d_kps, d_desc, nullptr, mpCap, d_nMP, d_valid, d_Pw, d_normal, d_maxDist,
                             d_minDist, d_mpDesc, d_Tcw, nullptr, d_Ow, cam8, nullptr, d_nLeft, (float)th, manualProjection ? 1 : 0, 0,
                             st, &qs, &cand, &cnt);
  if (rc != MORB_OK) return rc;
  void *ej = nullptr, *eb = nullptr;
  rc = morb_matcher_workspace(m, 2, sizeof(int) * (size_t)nprob * mpCap, &ej);
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 3, sizeof(int) * (size_t)nprob * mpCap, &eb);
  if (rc != MORB_OK) return rc;
  // bestDist <= TH_LOW * ratioHamming (int vs float product, :486 / :593) == bestDist <= floor(TH_LOW * ratioHamming)
  const int thAccept = (int)floorf((float)TH_LOW * ratioHamming);
  MORB_HIP_CHECK(hipMemsetAsync(d_matchF, 0xFF, sizeof(int) * (size_t)nprob * cap, st));   // -1 everywhere
  // the sequential pass: features already holding a match are skipped and a new match blocks its feature (:466, :487)
  hipLaunchKernelGGL(k_resolve<0>, dim3(nprob), dim3(64), (size_t)cap, st, *P, mpCap, d_nMP, qs, d_mpDesc, (const uint8_t*)nullptr,
                     d_kfImg, cap, d_count, d_kps, d_desc, (const float*)nullptr, d_matched, cand, cnt, 0.f, thAccept, 0, d_matchF,
                     d_nmatches, (int*)ej, (int*)eb, (float*)nullptr, (const int*)nullptr, (const int*)nullptr, (const int*)nullptr);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

extern "C" int morb_search_by_projection_sim3_batch(morb_matcher* m, const morb_frame_params* P, int nprob, const int* d_kfImg, int cap,
                                                    const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc,
                                                    const float* d_Tcw, const float* d_Ow, int mpCap, const int* d_nMP,
                                                    const uint8_t* d_valid, const float* d_Pw, const float* d_normal,
                                                    const float* d_maxDist, const float* d_minDist, const uint8_t* d_mpDesc,
                                                    const uint8_t* d_matched, int th, float ratioHamming, int manualProjection,
                                                    int* d_matchF, int* d_nmatches, void* stream) {
  return search_by_projection_sim3_impl(m, P, nprob, d_kfImg, cap, d_count, d_kps, d_desc, d_Tcw, d_Ow, mpCap, d_nMP, d_valid, d_Pw, d_normal, d_maxDist,
                                        d_minDist, d_mpDesc, d_matched, th, ratioHamming, manualProjection, nullptr, nullptr, d_matchF, d_nmatches, stream);
}
extern "C" int morb_search_by_projection_sim3_rig_batch(morb_matcher* m, const morb_frame_params* P, int nprob, const int* d_kfImg, int cap,
                                                        const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc,
                                                        const float* d_Tcw, const float* d_Ow, int mpCap, const int* d_nMP,
                                                        const uint8_t* d_valid, const float* d_Pw, const float* d_normal,
                                                        const float* d_maxDist, const float* d_minDist, const uint8_t* d_mpDesc,
                                                        const uint8_t* d_matched, int th, float ratioHamming, int manualProjection,
                                                        const float* cam8, const int* d_nLeft, int* d_matchF, int* d_nmatches, void* stream) {
  MORB_REQUIRE(d_nLeft && (cam8 || manualProjection), MORB_ERR_INVALID, "rig form: NLeft per keyframe, and the left camera's parameters unless the projection is the manual one");
  return search_by_projection_sim3_impl(m, P, nprob, d_kfImg, cap, d_count, d_kps, d_desc, d_Tcw, d_Ow, mpCap, d_nMP, d_valid, d_Pw, d_normal, d_maxDist,
                                        d_minDist, d_mpDesc, d_matched, th, ratioHamming, manualProjection, manualProjection ? nullptr : cam8, d_nLeft,
                                        d_matchF, d_nmatches, stream);
}

static int search_by_sim3_impl(morb_matcher* m, const morb_frame_params* P, int npairs, const int* d_kf1Img,
                               const int* d_kf2Img, int cap, const int* d_count, const morb_keypoint* d_kps,
                               const uint8_t* d_desc, const float* d_T1w, const float* d_T2w, const float* d_S12,
                               const float* d_S21, const uint8_t* d_valid1, const float* d_Pw1, const float* d_maxDist1,
                               const float* d_minDist1, const uint8_t* d_mpDesc1, const uint8_t* d_valid2,
                               const float* d_Pw2, const float* d_maxDist2, const float* d_minDist2,
                               const uint8_t* d_mpDesc2, float th, const int* d_nLeft1, const int* d_nLeft2, int* d_vnMatch1, int* d_vnMatch2,
                               int* d_match12, int* d_nFound, void* stream) {
  MORB_REQUIRE(m && P && d_kf1Img && d_kf2Img && d_count && d_kps && d_desc && d_T1w && d_T2w && d_S12 && d_S21 && d_valid1 && d_Pw1 &&
                   d_maxDist1 && d_minDist1 && d_mpDesc1 && d_valid2 && d_Pw2 && d_maxDist2 && d_minDist2 && d_mpDesc2 && d_vnMatch1 &&
                   d_vnMatch2 && d_match12 && d_nFound, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(npairs > 0 && cap > 0 && cap <= 65535 && P->nlevels >= 1 && P->nlevels <= 16, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  void* nq = nullptr;
  int rc = morb_matcher_workspace(m, 6, sizeof(int) * (size_t)npairs * 2, &nq);
  if (rc != MORB_OK) return rc;
  int* n1 = (int*)nq; int* n2 = n1 + npairs;
  hipLaunchKernelGGL(k_gather_counts, dim3(div_up(npairs, 256)), dim3(256), 0, st, d_count, d_kf1Img, npairs, n1);
  hipLaunchKernelGGL(k_gather_counts, dim3(div_up(npairs, 256)), dim3(256), 0, st, d_count, d_kf2Img, npairs, n2);
  const Query* qs; const unsigned long long* cand; const int* cnt;
  // map points of keyframe 1 -> camera 1 -> camera 2 (S21) -> keyframe 2's features (:1354-1425)
  rc = kfproj_candidates(m, P, npairs, d_kf2Img, cap, d_count, d_kps, d_desc, nullptr, cap, n1, d_valid1, d_Pw1, nullptr, d_maxDist1,
                         d_minDist1, d_mpDesc1, d_T1w, d_S21, nullptr, nullptr, nullptr, d_nLeft2, th, 2, 0, st, &qs, nullptr, nullptr);
  if (rc != MORB_OK) return rc;
  rc = best_per_query(m, P, npairs, cap, n1, qs, d_mpDesc1, d_kf2Img, cap, d_count, d_kps, d_desc, nullptr, TH_HIGH, d_vnMatch1, nullptr, st);
  if (rc != MORB_OK) return rc;
  // and the other way round (:1428-1499)
  rc = kfproj_candidates(m, P, npairs, d_kf1Img, cap, d_count, d_kps, d_desc, nullptr, cap, n2, d_valid2, d_Pw2, nullptr, d_maxDist2,
                         d_minDist2, d_mpDesc2, d_T2w, d_S12, nullptr, nullptr, nullptr, d_nLeft1, th, 2, 0, st, &qs, nullptr, nullptr);
  if (rc != MORB_OK) return rc;
  rc = best_per_query(m, P, npairs, cap, n2, qs, d_mpDesc2, d_kf1Img, cap, d_count, d_kps, d_desc, nullptr, TH_HIGH, d_vnMatch2, nullptr, st);
  if (rc != MORB_OK) return rc;
  MORB_HIP_CHECK(hipMemsetAsync(d_nFound, 0, sizeof(int) * npairs, st));
  hipLaunchKernelGGL(k_sim3_agree, dim3(div_up(cap, 256), npairs), dim3(256), 0, st, cap, d_vnMatch1, d_vnMatch2, d_match12, d_nFound);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}
extern "C" int morb_search_by_sim3_batch(morb_matcher* m, const morb_frame_params* P, int npairs, const int* d_kf1Img,
                                         const int* d_kf2Img, int cap, const int* d_count, const morb_keypoint* d_kps,
                                         const uint8_t* d_desc, const float* d_T1w, const float* d_T2w, const float* d_S12,
                                         const float* d_S21, const uint8_t* d_valid1, const float* d_Pw1, const float* d_maxDist1,
                                         const float* d_minDist1, const uint8_t* d_mpDesc1, const uint8_t* d_valid2,
                                         const float* d_Pw2, const float* d_maxDist2, const float* d_minDist2,
                                         const uint8_t* d_mpDesc2, float th, int* d_vnMatch1, int* d_vnMatch2, int* d_match12,
                                         int* d_nFound, void* stream) {
  return search_by_sim3_impl(m, P, npairs, d_kf1Img, d_kf2Img, cap, d_count, d_kps, d_desc, d_T1w, d_T2w, d_S12, d_S21, d_valid1, d_Pw1, d_maxDist1, d_minDist1,
                             d_mpDesc1, d_valid2, d_Pw2, d_maxDist2, d_minDist2, d_mpDesc2, th, nullptr, nullptr, d_vnMatch1, d_vnMatch2, d_match12, d_nFound, stream);
}
extern "C" int morb_search_by_sim3_rig_batch(morb_matcher* m, const morb_frame_params* P, int npairs, const int* d_kf1Img,
                                             const int* d_kf2Img, int cap, const int* d_count, const morb_keypoint* d_kps,
                                             const uint8_t* d_desc, const float* d_T1w, const float* d_T2w, const float* d_S12,
                                             const float* d_S21, const uint8_t* d_valid1, const float* d_Pw1, const float* d_maxDist1,
                                             const float* d_minDist1, const uint8_t* d_mpDesc1, const uint8_t* d_valid2,
                                             const float* d_Pw2, const float* d_maxDist2, const float* d_minDist2,
                                             const uint8_t* d_mpDesc2, float th, const int* d_nLeft1, const int* d_nLeft2, int* d_vnMatch1,
                                             int* d_vnMatch2, int* d_match12, int* d_nFound, void* stream) {
  MORB_REQUIRE(d_nLeft1 && d_nLeft2, MORB_ERR_INVALID, "rig form: NLeft of both keyframes");
  return search_by_sim3_impl(m, P, npairs, d_kf1Img, d_kf2Img, cap, d_count, d_kps, d_desc, d_T1w, d_T2w, d_S12, d_S21, d_valid1, d_Pw1, d_maxDist1, d_minDist1,
                             d_mpDesc1, d_valid2, d_Pw2, d_maxDist2, d_minDist2, d_mpDesc2, th, d_nLeft1, d_nLeft2, d_vnMatch1, d_vnMatch2, d_match12, d_nFound, stream);
}

extern "C" int morb_search_for_triangulation_batch(morb_matcher* m, const morb_frame_params* P, int npairs, const int* d_img1,
                                                   const int* d_img2, int nimg, int cap, const int* d_count,
                                                   const morb_keypoint* d_kps, const uint8_t* d_desc, const int* d_node,
                                                   const uint8_t* d_hasMP, const float* d_uRight, const float* R12,
                                                   const float* t12, const float* ep, int bOnlyStereo, int bCoarse,
                                                   int checkOri, int* d_match12, int* d_nmatches, void* stream) {
  MORB_REQUIRE(m && P && d_img1 && d_img2 && d_count && d_kps && d_desc && d_node && d_hasMP && R12 && t12 && ep && d_match12 &&
                   d_nmatches, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(npairs > 0 && nimg > 0 && cap > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  unsigned long long* sorted = nullptr;
  int rc = morb_bow_sort_images(m, nimg, d_node, d_count, cap, &sorted, st);
  if (rc != MORB_OK) return rc;
  std::vector<float> hf((size_t)npairs * 11);
  const float K[4] = {P->fx, P->fy, P->cx, P->cy};
  for (int p = 0; p < npairs; ++p) {
    fundamental_f12(K, K, R12 + 9 * p, t12 + 3 * p, &hf[(size_t)p * 9]);
    hf[(size_t)npairs * 9 + 2 * p] = ep[2 * p];
    hf[(size_t)npairs * 9 + 2 * p + 1] = ep[2 * p + 1];
  }
  void *dF = nullptr, *dBin = nullptr;
  rc = morb_matcher_workspace(m, 4, sizeof(float) * hf.size(), &dF);
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 2, sizeof(int) * (size_t)npairs * cap, &dBin);
  if (rc != MORB_OK) return rc;
  MORB_HIP_CHECK(hipMemcpyAsync(dF, hf.data(), sizeof(float) * hf.size(), hipMemcpyHostToDevice, st));
  MORB_HIP_CHECK(hipStreamSynchronize(st));  // hf is a local
  hipLaunchKernelGGL(k_triangulation, dim3(div_up(cap, 4), npairs), dim3(256), 0, st, sorted, cap, d_count, d_img1, d_img2, d_kps,
                     d_desc, d_node, d_hasMP, d_uRight, *P, (const float*)dF, (const float*)dF + (size_t)npairs * 9, bOnlyStereo,
                     bCoarse, d_match12, (int*)dBin, nullptr, nullptr, nullptr);
  hipLaunchKernelGGL(k_rot_filter12, dim3(npairs), dim3(256), 0, st, d_count, d_img1, cap, checkOri, d_match12, (const int*)dBin, d_nmatches);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}

extern "C" int morb_search_for_triangulation_fisheye_batch(morb_matcher* m, const morb_frame_params* P, int npairs, const int* d_img1,
                                                           const int* d_img2, const int* d_nLeft1, const int* d_nLeft2, int nimg,
                                                           int cap, const int* d_count, const morb_keypoint* d_kps,
                                                           const uint8_t* d_desc, const int* d_node, const uint8_t* d_hasMP,
                                                           const float* camL8, const float* camR8, const float* T4, int bOnlyStereo,
                                                           int bCoarse, int checkOri, int* d_match12, int* d_nmatches, void* stream) {
  MORB_REQUIRE(m && P && d_img1 && d_img2 && d_nLeft1 && d_nLeft2 && d_count && d_kps && d_desc && d_node && d_hasMP && camL8 && camR8 &&
                   T4 && d_match12 && d_nmatches, MORB_ERR_INVALID, "NULL argument");
  MORB_REQUIRE(npairs > 0 && nimg > 0 && cap > 0, MORB_ERR_INVALID, "bad sizes");
  MORB_HIP_CHECK(hipSetDevice(morb_matcher_device(m)));
  hipStream_t st = stream ? (hipStream_t)stream : (hipStream_t)morb_matcher_stream(m);
  unsigned long long* sorted = nullptr;
  int rc = morb_bow_sort_images(m, nimg, d_node, d_count, cap, &sorted, st);
  if (rc != MORB_OK) return rc;
  std::vector<float> hf(16 + (size_t)npairs * 48);
  for (int i = 0; i < 8; ++i) { hf[i] = camL8[i]; hf[8 + i] = camR8[i]; }
  for (size_t i = 0; i < (size_t)npairs * 48; ++i) hf[16 + i] = T4[i];
  void *dF = nullptr, *dBin = nullptr;
  rc = morb_matcher_workspace(m, 4, sizeof(float) * hf.size(), &dF);
  if (rc == MORB_OK) rc = morb_matcher_workspace(m, 2, sizeof(int) * (size_t)npairs * cap, &dBin);
  if (rc != MORB_OK) return rc;
  MORB_HIP_CHECK(hipMemcpyAsync(dF, hf.data(), sizeof(float) * hf.size(), hipMemcpyHostToDevice, st));
  MORB_HIP_CHECK(hipStreamSynchronize(st));  // hf is a local
  hipLaunchKernelGGL(k_triangulation, dim3(div_up(cap, 4), npairs), dim3(256), 0, st, sorted, cap, d_count, d_img1, d_img2, d_kps,
                     d_desc, d_node, d_hasMP, (const float*)nullptr, *P, (const float*)dF, (const float*)dF, bOnlyStereo, bCoarse,
                     d_match12, (int*)dBin, (const float*)dF, d_nLeft1, d_nLeft2);
  hipLaunchKernelGGL(k_rot_filter12, dim3(npairs), dim3(256), 0, st, d_count, d_img1, cap, checkOri, d_match12, (const int*)dBin, d_nmatches);
  MORB_HIP_CHECK(hipGetLastError());
  return MORB_OK;
}
