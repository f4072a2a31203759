// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path.
//
// CPU restatement (flattened inputs) of the projection-guided matchers and the frustum test, non-fisheye
// branches (Frame::Nleft == -1):
//   Frame::isInFrustum                    /root/reference/src/Frame.cc:611-678, MapPoint::PredictScale MapPoint.cc:536-566
//   Frame::PosInGrid / GetFeaturesInArea  Frame.cc:809-820, :742-807 (grid 64 x 48, Frame.h:44-45)
//   ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th, bFarPoints, thFarPoints)   ORBmatcher.cc:42-209
//   ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono)                 ORBmatcher.cc:1521-1733
//   ORBmatcher::SearchForTriangulation                                                        ORBmatcher.cc:821-1042
//   Pinhole::project / epipolarConstrain                                                      Pinhole.cpp:46-52, :111-139
// Eigen / Sophus float expressions are restated with explicit left-to-right float arithmetic (the oracle is
// built -ffp-contract=off); vs the real binary they are equal up to float rounding (PARITY UNPINNED there).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "cvprims.h"
#include <climits>
#include "matcher.h"
#include "orb_oracle.h"

namespace orc {

static const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;
static const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;

void ComputeThreeMaxima(const std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3);

struct Grid {
  std::vector<size_t> cell[FRAME_GRID_COLS][FRAME_GRID_ROWS];
};

// Frame::PosInGrid (:809-820)
static bool PosInGrid(const orc_frame& F, const KeyPoint& kp, int& posX, int& posY) {
  posX = (int)std::round((kp.x - F.minX) * F.gridInvW);
  posY = (int)std::round((kp.y - F.minY) * F.gridInvH);
  if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) return false;
  return true;
}
// Frame::AssignFeaturesToGrid (:501-528)
static void AssignFeaturesToGrid(const orc_frame& F, Grid& g) {
  const KeyPoint* k = (const KeyPoint*)F.kpsUn;
  for (int i = 0; i < F.N; i++) {
    int x, y;
    if (PosInGrid(F, k[i], x, y)) g.cell[x][y].push_back(i);
  }
}
// Frame::GetFeaturesInArea (:742-807)
static std::vector<size_t> GetFeaturesInArea(const orc_frame& F, const Grid& g, float x, float y, float r, int minLevel,
                                             int maxLevel) {
  std::vector<size_t> vIndices;
  const KeyPoint* k = (const KeyPoint*)F.kpsUn;
  const float factorX = r, factorY = r;
  const int nMinCellX = std::max(0, (int)std::floor((x - F.minX - factorX) * F.gridInvW));
  if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
  const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)std::ceil((x - F.minX + factorX) * F.gridInvW));
  if (nMaxCellX < 0) return vIndices;
  const int nMinCellY = std::max(0, (int)std::floor((y - F.minY - factorY) * F.gridInvH));
  if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
  const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)std::ceil((y - F.minY + factorY) * F.gridInvH));
  if (nMaxCellY < 0) return vIndices;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
    for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
      const std::vector<size_t>& vCell = g.cell[ix][iy];
      for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
        const KeyPoint& kpUn = k[vCell[j]];
        if (bCheckLevels) {
          if (kpUn.octave < minLevel) continue;
          if (maxLevel >= 0)
            if (kpUn.octave > maxLevel) continue;
        }
        const float distx = kpUn.x - x;
        const float disty = kpUn.y - y;
        if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) vIndices.push_back(vCell[j]);
      }
    }
  return vIndices;
}

// Eigen quaternion * vector (float), Sophus::SE3f * Vector3f
static void rotateF(const float* q, const float* v, float* out) {
  const float ux = q[0], uy = q[1], uz = q[2], w = q[3];
  float a = uy * v[2] - uz * v[1], b = uz * v[0] - ux * v[2], c = ux * v[1] - uy * v[0];
  a += a; b += b; c += c;
  out[0] = v[0] + w * a + (uy * c - uz * b);
  out[1] = v[1] + w * b + (uz * a - ux * c);
  out[2] = v[2] + w * c + (ux * b - uy * a);
}

// ---- Frame::isInFrustum (:611-678) ----------------------------------------------------------------------------
void isInFrustum(const orc_frame& F, const float* mRcw, const float* mtcw, const float* mOw, int nMP, const float* Pw,
                 const float* normal, const float* mfMaxDistance, const float* mfMinDistance, float viewingCosLimit,
                 uint8_t* mbTrackInView, float* mTrackProjX, float* mTrackProjY, float* mTrackProjXR, float* mTrackDepth,
                 int* mnTrackScaleLevel, float* mTrackViewCos) {
  for (int i = 0; i < nMP; ++i) {
    mbTrackInView[i] = 0;
    mTrackProjX[i] = -1; mTrackProjY[i] = -1; mTrackProjXR[i] = -1; mTrackDepth[i] = -1; mnTrackScaleLevel[i] = -1;
    mTrackViewCos[i] = -1;
    const float* P = Pw + 3 * i;
    float Pc[3];
    for (int r = 0; r < 3; ++r) Pc[r] = (mRcw[r * 3] * P[0] + mRcw[r * 3 + 1] * P[1]) + mRcw[r * 3 + 2] * P[2] + mtcw[r];
    const float Pc_dist = std::sqrt(Pc[0] * Pc[0] + Pc[1] * Pc[1] + Pc[2] * Pc[2]);
    const float PcZ = Pc[2];
    const float invz = 1.0f / PcZ;
    if (PcZ < 0.0f) continue;
    const float u = F.fx * Pc[0] / Pc[2] + F.cx;  // Pinhole::project(Vector3f)
    const float v = F.fy * Pc[1] / Pc[2] + F.cy;
    if (u < F.minX || u > F.maxX) continue;
    if (v < F.minY || v > F.maxY) continue;
    mTrackProjX[i] = u;
    mTrackProjY[i] = v;
    const float maxDistance = 1.2f * mfMaxDistance[i];  // MapPoint::GetMaxDistanceInvariance
    const float minDistance = 0.8f * mfMinDistance[i];
    const float PO[3] = {P[0] - mOw[0], P[1] - mOw[1], P[2] - mOw[2]};
    const float dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    if (dist < minDistance || dist > maxDistance) continue;
    const float* Pn = normal + 3 * i;
    const float viewCos = (PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) / dist;
    if (viewCos < viewingCosLimit) continue;
    // MapPoint::PredictScale (MapPoint.cc:552-566): log(float) = logf
    const float ratio = mfMaxDistance[i] / dist;
    int nScale = (int)std::ceil(std::log(ratio) / F.logScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= F.nlevels) nScale = F.nlevels - 1;
    mbTrackInView[i] = 1;
    mTrackProjXR[i] = u - F.mbf * invz;
    mTrackDepth[i] = Pc_dist;
    mnTrackScaleLevel[i] = nScale;
    mTrackViewCos[i] = viewCos;
  }
}

// ---- ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints) (:42-209) ----
int SearchByProjectionMPs(const orc_frame& F, const uint8_t* fBlocked, int nMP, const uint8_t* mbTrackInView,
                          const uint8_t* isBad, const float* mTrackDepth, const float* mTrackProjX,
                          const float* mTrackProjY, const float* mTrackProjXR, const int* mnTrackScaleLevel,
                          const float* mTrackViewCos, const uint8_t* mpDesc, const uint8_t* mpHasObs, float th,
                          bool bFarPoints, float thFarPoints, float mfNNratio, int* matchF) {
  Grid g;
  AssignFeaturesToGrid(F, g);
  const KeyPoint* k = (const KeyPoint*)F.kpsUn;
  std::vector<char> blocked(fBlocked, fBlocked + F.N);
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  for (int iMP = 0; iMP < nMP; iMP++) {
    if (!mbTrackInView[iMP]) continue;
    if (bFarPoints && mTrackDepth[iMP] > thFarPoints) continue;
    if (isBad[iMP]) continue;
    const int nPredictedLevel = mnTrackScaleLevel[iMP];
    float r = mTrackViewCos[iMP] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos (:211-216)
    if (bFactor) r *= th;
    const std::vector<size_t> vIndices = GetFeaturesInArea(F, g, mTrackProjX[iMP], mTrackProjY[iMP],
                                                           r * F.scaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
    if (vIndices.empty()) continue;
    const uint8_t* MPdescriptor = mpDesc + (size_t)iMP * 32;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (size_t q = 0; q < vIndices.size(); ++q) {
      const size_t idx = vIndices[q];
      if (blocked[idx]) continue;  // F.mvpMapPoints[idx] && Observations() > 0
      if (F.uRight && F.uRight[idx] > 0) {
        const float er = std::fabs(mTrackProjXR[iMP] - F.uRight[idx]);
        if (er > r * F.scaleFactors[nPredictedLevel]) continue;
      }
      const int dist = DescriptorDistance(MPdescriptor, F.desc + idx * 32);
      if (dist < bestDist) {
        bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = k[idx].octave; bestIdx = (int)idx;
      } else if (dist < bestDist2) {
        bestLevel2 = k[idx].octave; bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
      if (bestLevel != bestLevel2 || bestDist <= mfNNratio * bestDist2) {
        matchF[bestIdx] = iMP;
        blocked[bestIdx] = mpHasObs[iMP] ? 1 : 0;
        nmatches++;
      }
    }
  }
  return nmatches;
}

// ---- Frame::isInFrustumChecks (Frame.cc:1276-1346), one camera of a KannalaBrandt8 rig -------------------------
// The caller passes mR, mt, twc exactly as :1283-1293 builds them (left: mRcw, mtcw, mOw; right: Rrl * mRcw,
// Rrl * mtcw + trl, mRwc * mTlr.translation() + mOw) and the camera's parameters (fx fy cx cy k0..k3).  Fields of a
// point that fails a check are reported as -1 (the reference leaves them stale; Frame::isInFrustum :668-671 resets the
// two flags and the two levels to false / -1 before the checks).
extern "C" void orc_kb8_project_f(const float* cam8, const float* v3, float* uv);
void isInFrustumChecks(const orc_frame& F, const float* cam8, const float* mR, const float* mt, const float* twc, int nMP,
                       const float* Pw, const float* normal, const float* mfMaxDistance, const float* mfMinDistance,
                       float viewingCosLimit, uint8_t* inView, float* projX, float* projY, float* depth, int* level,
                       float* viewCosOut) {
  for (int i = 0; i < nMP; ++i) {
    inView[i] = 0; projX[i] = -1; projY[i] = -1; depth[i] = -1; level[i] = -1; viewCosOut[i] = -1;
    const float* P = Pw + 3 * i;
    float Pc[3];
    for (int r = 0; r < 3; ++r) Pc[r] = (mR[r * 3] * P[0] + mR[r * 3 + 1] * P[1]) + mR[r * 3 + 2] * P[2] + mt[r];
    const float Pc_dist = std::sqrt(Pc[0] * Pc[0] + Pc[1] * Pc[1] + Pc[2] * Pc[2]);
    if (Pc[2] < 0.0f) continue;
    float uv[2];
    orc_kb8_project_f(cam8, Pc, uv);
    if (uv[0] < F.minX || uv[0] > F.maxX) continue;
    if (uv[1] < F.minY || uv[1] > F.maxY) continue;
    const float maxDistance = 1.2f * mfMaxDistance[i];
    const float minDistance = 0.8f * mfMinDistance[i];
    const float PO[3] = {P[0] - twc[0], P[1] - twc[1], P[2] - twc[2]};
    const float dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    if (dist < minDistance || dist > maxDistance) continue;
    const float* Pn = normal + 3 * i;
    const float viewCos = (PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) / dist;
    if (viewCos < viewingCosLimit) continue;
    const float ratio = mfMaxDistance[i] / dist;
    int nScale = (int)std::ceil(std::log(ratio) / F.logScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= F.nlevels) nScale = F.nlevels - 1;
    inView[i] = 1; projX[i] = uv[0]; projY[i] = uv[1]; depth[i] = Pc_dist; level[i] = nScale; viewCosOut[i] = viewCos;
  }
}

// ---- SearchByProjection(Frame&, MapPoints) on a fisheye rig: the F.Nleft != -1 branches of :42-209 -------------------
// F holds the Nleft left features followed by the right ones (mvKeys | mvKeysRight, mDescriptors = vconcat, Frame.cc:211).
// l2r / r2l = mvLeftToRightMatch / mvRightToLeftMatch.
int SearchByProjectionMPsFisheye(const orc_frame& F, int Nleft, const int* l2r, const int* r2l, const uint8_t* fBlocked, int nMP,
                                 const uint8_t* inViewL, const uint8_t* inViewR, const uint8_t* isBad, const float* depthL,
                                 const float* projXL, const float* projYL, const int* levelL, const float* viewCosL,
                                 const float* projXR, const float* projYR, const int* levelR, const float* viewCosR,
                                 const uint8_t* mpDesc, const uint8_t* mpHasObs, float th, bool bFarPoints, float thFarPoints,
                                 float mfNNratio, int* matchF) {
  orc_frame FL = F, FR = F;
  FL.N = Nleft;
  FR.N = F.N - Nleft; FR.kpsUn = F.kpsUn + Nleft; FR.desc = F.desc + (size_t)Nleft * 32;
  Grid gl, gr;                                  // mGrid, mGridRight (Frame.cc:501-528)
  AssignFeaturesToGrid(FL, gl);
  AssignFeaturesToGrid(FR, gr);
  const KeyPoint* k = (const KeyPoint*)F.kpsUn;
  std::vector<char> blocked(fBlocked, fBlocked + F.N);
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  for (int iMP = 0; iMP < nMP; iMP++) {
    if (!inViewL[iMP] && !inViewR[iMP]) continue;
    if (bFarPoints && depthL[iMP] > thFarPoints) continue;
    if (isBad[iMP]) continue;
    const uint8_t* MPdescriptor = mpDesc + (size_t)iMP * 32;
    if (inViewL[iMP]) {
      const int nPredictedLevel = levelL[iMP];
      float r = viewCosL[iMP] > 0.998 ? 2.5f : 4.0f;
      if (bFactor) r *= th;
      const std::vector<size_t> vIndices = GetFeaturesInArea(FL, gl, projXL[iMP], projYL[iMP], r * F.scaleFactors[nPredictedLevel],
                                                             nPredictedLevel - 1, nPredictedLevel);
      if (!vIndices.empty()) {
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t q = 0; q < vIndices.size(); ++q) {
          const size_t idx = vIndices[q];
          if (blocked[idx]) continue;
          const int dist = DescriptorDistance(MPdescriptor, F.desc + idx * 32);
          if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = k[idx].octave; bestIdx = (int)idx;
          } else if (dist < bestDist2) {
            bestLevel2 = k[idx].octave; bestDist2 = dist;
          }
        }
        if (bestDist <= TH_HIGH) {
          if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;   // skips the right pass too
          if (bestLevel != bestLevel2 || bestDist <= mfNNratio * bestDist2) {
            matchF[bestIdx] = iMP; blocked[bestIdx] = mpHasObs[iMP] ? 1 : 0;
            if (l2r[bestIdx] != -1) {
              matchF[l2r[bestIdx] + Nleft] = iMP; blocked[l2r[bestIdx] + Nleft] = mpHasObs[iMP] ? 1 : 0;
              nmatches++;
            }
            nmatches++;
          }
        }
      }
    }
    if (inViewR[iMP]) {
      const int nPredictedLevel = levelR[iMP];
      if (nPredictedLevel != -1) {
        const float r = viewCosR[iMP] > 0.998 ? 2.5f : 4.0f;   // no th factor in the right pass (:145)
        const std::vector<size_t> vIndices = GetFeaturesInArea(FR, gr, projXR[iMP], projYR[iMP], r * F.scaleFactors[nPredictedLevel],
                                                               nPredictedLevel - 1, nPredictedLevel);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t q = 0; q < vIndices.size(); ++q) {
          const size_t idx = vIndices[q];
          if (blocked[idx + Nleft]) continue;
          const int dist = DescriptorDistance(MPdescriptor, F.desc + (idx + Nleft) * 32);
          if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = k[idx + Nleft].octave; bestIdx = (int)idx;
          } else if (dist < bestDist2) {
            bestLevel2 = k[idx + Nleft].octave; bestDist2 = dist;
          }
        }
        if (bestDist <= TH_HIGH) {
          if (bestLevel == bestLevel2 && bestDist > mfNNratio * bestDist2) continue;
          if (r2l[bestIdx] != -1) {
            matchF[r2l[bestIdx]] = iMP; blocked[r2l[bestIdx]] = mpHasObs[iMP] ? 1 : 0;
            nmatches++;
          }
          matchF[bestIdx + Nleft] = iMP; blocked[bestIdx + Nleft] = mpHasObs[iMP] ? 1 : 0;
          nmatches++;
        }
      }
    }
  }
  return nmatches;
}

// ---- ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono) (:1521-1733) ----
// bForward / bBackward are computed by the caller exactly as :1530-1539 does (two flags per call).
int SearchByProjectionLast(const orc_frame& Cur, const uint8_t* curBlocked, const float* Tcw7, int nLast,
                           const KeyPoint* lastKpsUn, const uint8_t* lastValid, const float* lastXw,
                           const uint8_t* lastMPdesc, const uint8_t* lastMPhasObs, float th, bool bForward, bool bBackward,
                           bool mbCheckOrientation, int* matchCur) {
  Grid g;
  AssignFeaturesToGrid(Cur, g);
  const KeyPoint* kc = (const KeyPoint*)Cur.kpsUn;
  std::vector<char> blocked(curBlocked, curBlocked + Cur.N);
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  for (int i = 0; i < nLast; i++) {
    if (!lastValid[i]) continue;  // pMP && !LastFrame.mvbOutlier[i]
    float x3Dc[3];
    rotateF(Tcw7, lastXw + 3 * i, x3Dc);
    x3Dc[0] += Tcw7[4]; x3Dc[1] += Tcw7[5]; x3Dc[2] += Tcw7[6];
    const float invzc = (float)(1.0 / x3Dc[2]);
    if (invzc < 0) continue;
    const float u = Cur.fx * x3Dc[0] / x3Dc[2] + Cur.cx;
    const float v = Cur.fy * x3Dc[1] / x3Dc[2] + Cur.cy;
    if (u < Cur.minX || u > Cur.maxX) continue;
    if (v < Cur.minY || v > Cur.maxY) continue;
    const int nLastOctave = lastKpsUn[i].octave;
    const float radius = th * Cur.scaleFactors[nLastOctave];
    std::vector<size_t> vIndices2;
    if (bForward) vIndices2 = GetFeaturesInArea(Cur, g, u, v, radius, nLastOctave, -1);
    else if (bBackward) vIndices2 = GetFeaturesInArea(Cur, g, u, v, radius, 0, nLastOctave);
    else vIndices2 = GetFeaturesInArea(Cur, g, u, v, radius, nLastOctave - 1, nLastOctave + 1);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = lastMPdesc + (size_t)i * 32;
    int bestDist = 256, bestIdx2 = -1;
    for (size_t q = 0; q < vIndices2.size(); ++q) {
      const size_t i2 = vIndices2[q];
      if (blocked[i2]) continue;
      if (Cur.uRight && Cur.uRight[i2] > 0) {
        const float ur = u - Cur.mbf * invzc;
        const float er = std::fabs(ur - Cur.uRight[i2]);
        if (er > radius) continue;
      }
      const int dist = DescriptorDistance(dMP, Cur.desc + i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
    }
    if (bestDist <= TH_HIGH) {
      matchCur[bestIdx2] = i;
      blocked[bestIdx2] = lastMPhasObs[i] ? 1 : 0;
      nmatches++;
      if (mbCheckOrientation) {
        float rot = lastKpsUn[i].angle - kc[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
          matchCur[rotHist[i][j]] = -1;
          nmatches--;
        }
      }
    }
  }
  return nmatches;
}

// ---- SearchByProjection(CurrentFrame, LastFrame, th, bMono) with CurrentFrame.Nleft != -1 (:1521-1733) --------------
// Cur holds the left features followed by the right ones.  Quirks kept: the right pass projects with mpCamera (the
// LEFT camera's model) after GetRelativePoseTrl(), has no bounds / depth test, and is skipped together with the left
// pass by every `continue` of the left pass (behind the camera, out of bounds, empty left window).
int SearchByProjectionLastFisheye(const orc_frame& Cur, int NleftCur, const float* cam8, const float* Trl7,
                                  const uint8_t* curBlocked, const float* Tcw7, int nLast, const KeyPoint* lastKps,
                                  const uint8_t* lastValid, const float* lastXw, const uint8_t* lastMPdesc,
                                  const uint8_t* lastMPhasObs, float th, bool bForward, bool bBackward,
                                  bool mbCheckOrientation, int* matchCur) {
  orc_frame FL = Cur, FR = Cur;
  FL.N = NleftCur;
  FR.N = Cur.N - NleftCur; FR.kpsUn = Cur.kpsUn + NleftCur; FR.desc = Cur.desc + (size_t)NleftCur * 32;
  Grid gl, gr;
  AssignFeaturesToGrid(FL, gl);
  AssignFeaturesToGrid(FR, gr);
  const KeyPoint* kc = (const KeyPoint*)Cur.kpsUn;
  std::vector<char> blocked(curBlocked, curBlocked + Cur.N);
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  for (int i = 0; i < nLast; i++) {
    if (!lastValid[i]) continue;
    float x3Dc[3];
    rotateF(Tcw7, lastXw + 3 * i, x3Dc);
    x3Dc[0] += Tcw7[4]; x3Dc[1] += Tcw7[5]; x3Dc[2] += Tcw7[6];
    const float invzc = (float)(1.0 / x3Dc[2]);
    if (invzc < 0) continue;
    float uv[2];
    orc_kb8_project_f(cam8, x3Dc, uv);
    if (uv[0] < Cur.minX || uv[0] > Cur.maxX) continue;
    if (uv[1] < Cur.minY || uv[1] > Cur.maxY) continue;
    const int nLastOctave = lastKps[i].octave;
    const float radius = th * Cur.scaleFactors[nLastOctave];
    const uint8_t* dMP = lastMPdesc + (size_t)i * 32;
    for (int pass = 0; pass < 2; ++pass) {
      const orc_frame& Fs = pass ? FR : FL;
      const Grid& g = pass ? gr : gl;
      const int base = pass ? NleftCur : 0;
      float u = uv[0], v = uv[1];
      if (pass) {
        float x3Dr[3], uvr[2];
        rotateF(Trl7, x3Dc, x3Dr);
        x3Dr[0] += Trl7[4]; x3Dr[1] += Trl7[5]; x3Dr[2] += Trl7[6];
        orc_kb8_project_f(cam8, x3Dr, uvr);
        u = uvr[0]; v = uvr[1];
      }
      std::vector<size_t> vIndices2;
      if (bForward) vIndices2 = GetFeaturesInArea(Fs, g, u, v, radius, nLastOctave, -1);
      else if (bBackward) vIndices2 = GetFeaturesInArea(Fs, g, u, v, radius, 0, nLastOctave);
      else vIndices2 = GetFeaturesInArea(Fs, g, u, v, radius, nLastOctave - 1, nLastOctave + 1);
      if (!pass && vIndices2.empty()) break;   // :1583: `continue` of the outer loop, the right pass is skipped too
      int bestDist = 256, bestIdx2 = -1;
      for (size_t q = 0; q < vIndices2.size(); ++q) {
        const size_t i2 = vIndices2[q];
        if (blocked[i2 + base]) continue;
        const int dist = DescriptorDistance(dMP, Cur.desc + (i2 + base) * 32);
        if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
      }
      if (bestDist <= TH_HIGH) {
        matchCur[bestIdx2 + base] = i;
        blocked[bestIdx2 + base] = lastMPhasObs[i] ? 1 : 0;
        nmatches++;
        if (mbCheckOrientation) {
          float rot = lastKps[i].angle - kc[bestIdx2 + base].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(bestIdx2 + base);
        }
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
          matchCur[rotHist[i][j]] = -1;
          nmatches--;
        }
      }
    }
  }
  return nmatches;
}

// ---- ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist) (:1735-1842) ----
// kfValid[i] != 0 <=> vpMPs[i] && !isBad() && !sAlreadyFound.count(pMP); curHasMP[i2] != 0 <=> CurrentFrame.mvpMapPoints[i2].
// cam8 != NULL: CurrentFrame is a KannalaBrandt8 rig frame — Cur holds its LEFT features only (what mGrid holds and GetFeaturesInArea's
// default bRight = false searches), the projection is mpCamera's; kfKpsUn[i] for a right keyframe feature: see include/morb_hip.h
int SearchByProjectionKF(const orc_frame& Cur, const uint8_t* curHasMP, const float* Tcw7, const float* Ow, int nKF,
                         const KeyPoint* kfKpsUn, const uint8_t* kfValid, const float* Xw, const float* mfMaxDistance,
                         const float* mfMinDistance, const uint8_t* mpDesc, float th, int ORBdist, bool mbCheckOrientation,
                         int* matchCur, const float* cam8 = nullptr) {
  Grid g;
  AssignFeaturesToGrid(Cur, g);
  const KeyPoint* kc = (const KeyPoint*)Cur.kpsUn;
  std::vector<char> has(curHasMP, curHasMP + Cur.N);
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  for (int i = 0; i < nKF; i++) {
    if (!kfValid[i]) continue;
    const float* x3Dw = Xw + 3 * i;
    float x3Dc[3];
    rotateF(Tcw7, x3Dw, x3Dc);
    x3Dc[0] += Tcw7[4]; x3Dc[1] += Tcw7[5]; x3Dc[2] += Tcw7[6];
    float u = Cur.fx * x3Dc[0] / x3Dc[2] + Cur.cx;
    float v = Cur.fy * x3Dc[1] / x3Dc[2] + Cur.cy;
    if (cam8) { float uv[2]; orc_kb8_project_f(cam8, x3Dc, uv); u = uv[0]; v = uv[1]; }
    if (u < Cur.minX || u > Cur.maxX) continue;
    if (v < Cur.minY || v > Cur.maxY) continue;
    const float PO[3] = {x3Dw[0] - Ow[0], x3Dw[1] - Ow[1], x3Dw[2] - Ow[2]};
    const float dist3D = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    const float maxDistance = 1.2f * mfMaxDistance[i], minDistance = 0.8f * mfMinDistance[i];
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const float ratio = mfMaxDistance[i] / dist3D;
    int nPredictedLevel = (int)std::ceil(std::log(ratio) / Cur.logScaleFactor);
    if (nPredictedLevel < 0) nPredictedLevel = 0;
    else if (nPredictedLevel >= Cur.nlevels) nPredictedLevel = Cur.nlevels - 1;
    const float radius = th * Cur.scaleFactors[nPredictedLevel];
    const std::vector<size_t> vIndices2 = GetFeaturesInArea(Cur, g, u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = mpDesc + (size_t)i * 32;
    int bestDist = 256, bestIdx2 = -1;
    for (size_t q = 0; q < vIndices2.size(); ++q) {
      const size_t i2 = vIndices2[q];
      if (has[i2]) continue;
      const int dist = DescriptorDistance(dMP, Cur.desc + i2 * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
    }
    if (bestDist <= ORBdist) {
      matchCur[bestIdx2] = i;
      has[bestIdx2] = 1;
      nmatches++;
      if (mbCheckOrientation) {
        float rot = kfKpsUn[i].angle - kc[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { matchCur[rotHist[i][j]] = -1; nmatches--; }
  }
  return nmatches;
}

// ---- ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) (:603-700) ----
int SearchForInitialization(int n1, const KeyPoint* kps1, const uint8_t* desc1, const orc_frame& F2, float* vbPrevMatched,
                            int windowSize, float mfNNratio, bool mbCheckOrientation, int* vnMatches12) {
  Grid g;
  AssignFeaturesToGrid(F2, g);
  const KeyPoint* k2 = (const KeyPoint*)F2.kpsUn;
  int nmatches = 0;
  for (int i = 0; i < n1; ++i) vnMatches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  std::vector<int> vMatchedDistance(F2.N, 2147483647), vnMatches21(F2.N, -1);
  for (int i1 = 0; i1 < n1; i1++) {
    const KeyPoint kp1 = kps1[i1];
    const int level1 = kp1.octave;
    if (level1 > 0) continue;
    std::vector<size_t> vIndices2 = GetFeaturesInArea(F2, g, vbPrevMatched[2 * i1], vbPrevMatched[2 * i1 + 1], (float)windowSize, level1, level1);
    if (vIndices2.empty()) continue;
    const uint8_t* d1 = desc1 + (size_t)i1 * 32;
    int bestDist = 2147483647, bestDist2 = 2147483647, bestIdx2 = -1;
    for (size_t q = 0; q < vIndices2.size(); ++q) {
      const size_t i2 = vIndices2[q];
      const int dist = DescriptorDistance(d1, F2.desc + i2 * 32);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
      else if (dist < bestDist2) { bestDist2 = dist; }
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * mfNNratio) {
        if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        vnMatches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (mbCheckOrientation) {
          float rot = kps1[i1].angle - k2[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(i1);
        }
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        const int idx1 = rotHist[i][j];
        if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
      }
    }
  }
  for (int i1 = 0; i1 < n1; i1++)
    if (vnMatches12[i1] >= 0) { vbPrevMatched[2 * i1] = k2[vnMatches12[i1]].x; vbPrevMatched[2 * i1 + 1] = k2[vnMatches12[i1]].y; }
  return nmatches;
}

// ---- Pinhole::epipolarConstrain (Pinhole.cpp:111-139) with F12 given --------------------------------------
static bool epipolarConstrain(const float* F12, const KeyPoint& kp1, const KeyPoint& kp2, float unc) {
  const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
  const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
  const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
  const float num = a * kp2.x + b * kp2.y + c;
  const float den = a * a + b * b;
  if (den == 0) return false;
  const float dsqr = num * num / den;
  return dsqr < 3.84 * unc;
}

// F12 = K1^-T * [t12]x * R12 * K2^-1 (Pinhole.cpp:118-122), float, products left to right
void FundamentalF12(const float* K1 /*fx fy cx cy*/, const float* K2, const float* R12, const float* t12, float* F12) {
  // K^-1 = [[1/fx, 0, -cx/fx], [0, 1/fy, -cy/fy], [0, 0, 1]];  K^-T is its transpose
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

// ---- ORBmatcher::SearchForTriangulation (:821-1042), pinhole / no second camera ---------------------------
int SearchForTriangulation(int n1, const KeyPoint* kps1, const uint8_t* desc1, const int* node1, const uint8_t* hasMP1,
                           const float* uRight1, const float* levelSigma2_1, int n2, const KeyPoint* kps2,
                           const uint8_t* desc2, const int* node2, const uint8_t* hasMP2, const float* uRight2,
                           const float* levelSigma2_2, const float* scaleFactors2, const float* F12, const float* ep,
                           bool bOnlyStereo, bool bCoarse, bool mbCheckOrientation, int* vMatches12) {
  std::map<int, std::vector<unsigned>> vFeatVec1, vFeatVec2;
  for (int i = 0; i < n1; ++i) if (node1[i] >= 0) vFeatVec1[node1[i]].push_back(i);
  for (int i = 0; i < n2; ++i) if (node2[i] >= 0) vFeatVec2[node2[i]].push_back(i);
  int nmatches = 0;
  for (int i = 0; i < n1; ++i) vMatches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  auto f1it = vFeatVec1.begin(), f1end = vFeatVec1.end();
  auto f2it = vFeatVec2.begin(), f2end = vFeatVec2.end();
  while (f1it != f1end && f2it != f2end) {
    if (f1it->first == f2it->first) {
      for (size_t i1 = 0, iend1 = f1it->second.size(); i1 < iend1; i1++) {
        const size_t idx1 = f1it->second[i1];
        if (hasMP1[idx1]) continue;
        const bool bStereo1 = uRight1 && uRight1[idx1] >= 0;
        if (bOnlyStereo) if (!bStereo1) continue;
        const KeyPoint& kp1 = kps1[idx1];
        const uint8_t* d1 = desc1 + idx1 * 32;
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (size_t i2 = 0, iend2 = f2it->second.size(); i2 < iend2; i2++) {
          const size_t idx2 = f2it->second[i2];
          if (hasMP2[idx2]) continue;  // vbMatched2 is never set in this fork
          const bool bStereo2 = uRight2 && uRight2[idx2] >= 0;
          if (bOnlyStereo) if (!bStereo2) continue;
          const int dist = DescriptorDistance(d1, desc2 + idx2 * 32);
          if (dist > TH_LOW || dist > bestDist) continue;
          const KeyPoint& kp2 = kps2[idx2];
          if (!bStereo1 && !bStereo2) {
            const float distex = ep[0] - kp2.x;
            const float distey = ep[1] - kp2.y;
            if (distex * distex + distey * distey < 100 * scaleFactors2[kp2.octave]) continue;
          }
          if (bCoarse || epipolarConstrain(F12, kp1, kp2, levelSigma2_2[kp2.octave])) {
            bestIdx2 = (int)idx2;
            bestDist = dist;
          }
        }
        if (bestIdx2 >= 0) {
          const KeyPoint& kp2 = kps2[bestIdx2];
          vMatches12[idx1] = bestIdx2;
          nmatches++;
          if (mbCheckOrientation) {
            float rot = kp1.angle - kp2.angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back((int)idx1);
          }
        }
      }
      f1it++;
      f2it++;
    } else if (f1it->first < f2it->first) {
      f1it = vFeatVec1.lower_bound(f2it->first);
    } else {
      f2it = vFeatVec2.lower_bound(f1it->first);
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        vMatches12[rotHist[i][j]] = -1;
        nmatches--;
      }
    }
  }
  (void)levelSigma2_1;
  return nmatches;
}

// ---- SearchForTriangulation between two keyframes of a KannalaBrandt8 rig (pKF->mpCamera2 != NULL, :821-1042) --------
// Features of each keyframe: left camera's first (NLeft), then the right camera's.  T4 = Tll, Tlr, Trl, Trr (:845-852),
// each as R (row-major 3x3) then t.  bStereo1 / bStereo2 are false on such rigs (so bOnlyStereo rejects everything),
// the epipole-distance gate is skipped (:925), the camera pair and (R12, t12) follow the sides of the two features
// (:934-966) and the constraint is KannalaBrandt8::epipolarConstrain = TriangulateMatches(...) > 0.0001 (:307-321).
extern "C" float orc_kb8_triangulate_matches(const float* cam1_8, const float* cam2_8, float x1, float y1, float x2, float y2,
                                             const float* R12, const float* t12, float sigmaLevel, float unc, float* p3D);
int SearchForTriangulationFisheye(int n1, int NLeft1, const KeyPoint* kps1, const uint8_t* desc1, const int* node1,
                                  const uint8_t* hasMP1, int n2, int NLeft2, const KeyPoint* kps2, const uint8_t* desc2,
                                  const int* node2, const uint8_t* hasMP2, const float* levelSigma2, const float* camL8,
                                  const float* camR8, const float* T4, bool bOnlyStereo, bool bCoarse, bool mbCheckOrientation,
                                  int* vMatches12) {
  std::map<int, std::vector<unsigned>> vFeatVec1, vFeatVec2;
  for (int i = 0; i < n1; ++i) if (node1[i] >= 0) vFeatVec1[node1[i]].push_back(i);
  for (int i = 0; i < n2; ++i) if (node2[i] >= 0) vFeatVec2[node2[i]].push_back(i);
  int nmatches = 0;
  for (int i = 0; i < n1; ++i) vMatches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  auto f1it = vFeatVec1.begin(), f1end = vFeatVec1.end();
  auto f2it = vFeatVec2.begin(), f2end = vFeatVec2.end();
  while (f1it != f1end && f2it != f2end) {
    if (f1it->first == f2it->first) {
      for (size_t i1 = 0, iend1 = f1it->second.size(); i1 < iend1; i1++) {
        const size_t idx1 = f1it->second[i1];
        if (hasMP1[idx1]) continue;
        if (bOnlyStereo) continue;                 // bStereo1 == false
        const KeyPoint& kp1 = kps1[idx1];
        const bool bRight1 = !((int)idx1 < NLeft1);
        const uint8_t* d1 = desc1 + idx1 * 32;
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (size_t i2 = 0, iend2 = f2it->second.size(); i2 < iend2; i2++) {
          const size_t idx2 = f2it->second[i2];
          if (hasMP2[idx2]) continue;
          const int dist = DescriptorDistance(d1, desc2 + idx2 * 32);
          if (dist > TH_LOW || dist > bestDist) continue;
          const KeyPoint& kp2 = kps2[idx2];
          const bool bRight2 = !((int)idx2 < NLeft2);
          const float* T = T4 + 12 * ((bRight1 ? 2 : 0) + (bRight2 ? 1 : 0));   // ll, lr, rl, rr
          float p3D[3];
          if (bCoarse || orc_kb8_triangulate_matches(bRight1 ? camR8 : camL8, bRight2 ? camR8 : camL8, kp1.x, kp1.y, kp2.x, kp2.y, T,
                                                     T + 9, levelSigma2[kp1.octave], levelSigma2[kp2.octave], p3D) > 0.0001f) {
            bestIdx2 = (int)idx2;
            bestDist = dist;
          }
        }
        if (bestIdx2 >= 0) {
          const KeyPoint& kp2 = kps2[bestIdx2];
          vMatches12[idx1] = bestIdx2;
          nmatches++;
          if (mbCheckOrientation) {
            float rot = kp1.angle - kp2.angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back((int)idx1);
          }
        }
      }
      f1it++;
      f2it++;
    } else if (f1it->first < f2it->first) {
      f1it = vFeatVec1.lower_bound(f2it->first);
    } else {
      f2it = vFeatVec2.lower_bound(f1it->first);
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        vMatches12[rotHist[i][j]] = -1;
        nmatches--;
      }
    }
  }
  return nmatches;
}

// =====================================================================================================================
// M7 ("next" row N2): the loop-closing / local-mapping variants of project -> window -> best Hamming.
// What is restated here is each method's search; the map-graph bookkeeping that follows a hit (Replace / AddObservation
// / AddMapPoint, ORBmatcher.cc:1196-1208, :1303-1310) stays with the caller and is replayed from the per-point results.
// =====================================================================================================================
// KeyFrame::GetFeaturesInArea (KeyFrame.cc:729-774): the Frame version without the level filter.
static std::vector<size_t> KFGetFeaturesInArea(const orc_frame& F, const Grid& g, float x, float y, float r) {
  return GetFeaturesInArea(F, g, x, y, r, -1, -1);   // bCheckLevels = (minLevel > 0) || (maxLevel >= 0) = false
}
static bool KFIsInImage(const orc_frame& F, float x, float y) { return x >= F.minX && x < F.maxX && y >= F.minY && y < F.maxY; }
static int PredictScaleKF(const orc_frame& F, float mfMaxDistance, float dist) {   // MapPoint.cc:536-550
  const float ratio = mfMaxDistance / dist;
  int nScale = (int)std::ceil(std::log(ratio) / F.logScaleFactor);
  if (nScale < 0) nScale = 0;
  else if (nScale >= F.nlevels) nScale = F.nlevels - 1;
  return nScale;
}

// ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th, bRight) (:1044-1213) with bRight == false on a pinhole
// keyframe, and Fuse(KeyFrame*, Sim3f& Scw, vpPoints, th, vpReplacePoint) (:1215-1321) with sim3Form (no chi2 gate;
// the caller passes Tcw = SE3(Scw.rotationMatrix(), Scw.translation() / Scw.scale()) and Ow = Tcw.inverse().translation()).
// valid[i] = the point passes the reference's state checks (non-null, !isBad(), !IsInKeyFrame / !spAlreadyFound).
// bestIdx[i] = feature chosen for map point i (bestDist <= TH_LOW) or -1.
void FuseSearch(const orc_frame& KF, const float* invLevelSigma2, const float* Tcw7, const float* Ow, int nMP,
                const uint8_t* valid, const float* Pw, const float* normal, const float* mfMaxDistance,
                const float* mfMinDistance, const uint8_t* mpDesc, float th, bool sim3Form, int* bestIdxOut, int* bestDistOut) {
  Grid g;
  AssignFeaturesToGrid(KF, g);
  const KeyPoint* k = (const KeyPoint*)KF.kpsUn;
  for (int i = 0; i < nMP; i++) {
    bestIdxOut[i] = -1; bestDistOut[i] = -1;
    if (!valid[i]) continue;
    const float* p3Dw = Pw + 3 * i;
    float p3Dc[3];
    rotateF(Tcw7, p3Dw, p3Dc);
    p3Dc[0] += Tcw7[4]; p3Dc[1] += Tcw7[5]; p3Dc[2] += Tcw7[6];
    if (p3Dc[2] < 0.0f) continue;
    const float invz = 1 / p3Dc[2];
    const float u = KF.fx * p3Dc[0] / p3Dc[2] + KF.cx, v = KF.fy * p3Dc[1] / p3Dc[2] + KF.cy;   // pCamera->project
    if (!KFIsInImage(KF, u, v)) continue;
    const float ur = u - KF.mbf * invz;
    const float maxDistance = 1.2f * mfMaxDistance[i], minDistance = 0.8f * mfMinDistance[i];
    const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
    const float dist3D = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const float* Pn = normal + 3 * i;
    if ((PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) < 0.5 * dist3D) continue;
    const int nPredictedLevel = PredictScaleKF(KF, mfMaxDistance[i], dist3D);
    const float radius = th * KF.scaleFactors[nPredictedLevel];
    const std::vector<size_t> vIndices = KFGetFeaturesInArea(KF, g, u, v, radius);
    if (vIndices.empty()) continue;
    const uint8_t* dMP = mpDesc + (size_t)i * 32;
    int bestDist = sim3Form ? INT_MAX : 256, bestIdx = -1;
    for (size_t q = 0; q < vIndices.size(); ++q) {
      const size_t idx = vIndices[q];
      const KeyPoint& kp = k[idx];
      const int kpLevel = kp.octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      if (!sim3Form) {
        if (KF.uRight && KF.uRight[idx] >= 0) {
          const float ex = u - kp.x, ey = v - kp.y, er = ur - KF.uRight[idx];
          const float e2 = ex * ex + ey * ey + er * er;
          if (e2 * invLevelSigma2[kpLevel] > 7.8) continue;
        } else {
          const float ex = u - kp.x, ey = v - kp.y;
          const float e2 = ex * ex + ey * ey;
          if (e2 * invLevelSigma2[kpLevel] > 5.99) continue;
        }
      }
      const int dist = DescriptorDistance(dMP, KF.desc + idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
    }
    if (bestDist <= TH_LOW) { bestIdxOut[i] = bestIdx; bestDistOut[i] = bestDist; }
  }
}

// ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th, bRight) (:1044-1213) on a KannalaBrandt8 rig (KeyFrame::NLeft != -1):
// KF holds the NLeft left features followed by the right ones (mvKeys | mvKeysRight, one descriptor matrix).  bRight picks the side:
// Tcw7 / Ow = GetRightPose() / GetRightCameraCenter() and cam8 = mpCamera2 (:1050-1054), else GetPose() / GetCameraCenter() /
// mpCamera; the window is looked up in that side's grid with that side's keypoints (KeyFrame.cc:758-764, side-local indices),
// mvuRight[idx] — read with the side-local index (:1154), all -1 on a rig — sends every candidate through the 5.99 gate, and the
// descriptor row / returned index is idx + NLeft on the right (:1177).
// sim3Form: Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (:1215-1321) on a rig keyframe — no rig branch in the reference: pCamera = pKF->mpCamera, the
// left features, no reprojection gate, bestDist starts at INT_MAX
void FuseSearchRig(const orc_frame& KF, int NLeft, bool bRight, const float* cam8, const float* invLevelSigma2, const float* Tcw7,
                   const float* Ow, int nMP, const uint8_t* valid, const float* Pw, const float* normal, const float* mfMaxDistance,
                   const float* mfMinDistance, const uint8_t* mpDesc, float th, int* bestIdxOut, int* bestDistOut, bool sim3Form = false) {
  orc_frame S = KF;   // the side's features as a frame of their own
  const int base = bRight ? NLeft : 0;
  S.N = bRight ? KF.N - NLeft : NLeft;
  S.kpsUn = KF.kpsUn + base; S.desc = KF.desc + (size_t)base * 32; S.uRight = nullptr;
  Grid g;
  AssignFeaturesToGrid(S, g);
  const KeyPoint* k = (const KeyPoint*)S.kpsUn;
  for (int i = 0; i < nMP; i++) {
    bestIdxOut[i] = -1; bestDistOut[i] = -1;
    if (!valid[i]) continue;
    const float* p3Dw = Pw + 3 * i;
    float p3Dc[3], uv[2];
    rotateF(Tcw7, p3Dw, p3Dc);
    p3Dc[0] += Tcw7[4]; p3Dc[1] += Tcw7[5]; p3Dc[2] += Tcw7[6];
    if (p3Dc[2] < 0.0f) continue;
    orc_kb8_project_f(cam8, p3Dc, uv);
    if (!KFIsInImage(KF, uv[0], uv[1])) continue;
    const float maxDistance = 1.2f * mfMaxDistance[i], minDistance = 0.8f * mfMinDistance[i];
    const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
    const float dist3D = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const float* Pn = normal + 3 * i;
    if ((PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) < 0.5 * dist3D) continue;
    const int nPredictedLevel = PredictScaleKF(KF, mfMaxDistance[i], dist3D);
    const float radius = th * KF.scaleFactors[nPredictedLevel];
    const std::vector<size_t> vIndices = KFGetFeaturesInArea(S, g, uv[0], uv[1], radius);
    if (vIndices.empty()) continue;
    const uint8_t* dMP = mpDesc + (size_t)i * 32;
    int bestDist = sim3Form ? INT_MAX : 256, bestIdx = -1;
    for (size_t q = 0; q < vIndices.size(); ++q) {
      const size_t idx = vIndices[q];
      const KeyPoint& kp = k[idx];
      const int kpLevel = kp.octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const float ex = uv[0] - kp.x, ey = uv[1] - kp.y;
      const float e2 = ex * ex + ey * ey;
      if (!sim3Form && e2 * invLevelSigma2[kpLevel] > 5.99) continue;
      const int dist = DescriptorDistance(dMP, S.desc + idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx + base; }
    }
    if (bestDist <= TH_LOW) { bestIdxOut[i] = bestIdx; bestDistOut[i] = bestDist; }
  }
}

// ORBmatcher::SearchByProjection(KeyFrame*, Sim3f& Scw, vpPoints, vpMatched, th, ratioHamming) (:397-494) and its twin
// with vpPointsKFs / vpMatchedKF (:496-601; manualProjection: u = fx * (X * invz) + cx instead of mpCamera->project).
// matched[idx] != 0 <=> vpMatched[idx] on entry; matchF[idx] = index of the map point newly assigned to feature idx.
// NLeft >= 0: a KannalaBrandt8 rig keyframe (KF holds mvKeys | mvKeysRight).  The reference has no rig branch here: GetFeaturesInArea(u, v, r) looks up
// the LEFT grid (bRight defaults to false, KeyFrame.cc:729-774), mvKeysUn is the copy of mvKeys, and the first form projects with
// pKF->mpCamera->project (:433) = the left KB8 camera (cam8); the twin keeps its pinhole formula on pKF->fx ... (:536-543).  vpMatched spans all N features.
int SearchByProjectionSim3(const orc_frame& KFall, const float* Tcw7, const float* Ow, int nMP, const uint8_t* valid,
                           const float* Pw, const float* normal, const float* mfMaxDistance, const float* mfMinDistance,
                           const uint8_t* mpDesc, const uint8_t* matchedIn, int th, float ratioHamming, bool manualProjection,
                           int* matchF, const float* cam8 = nullptr, int NLeft = -1) {
  orc_frame KF = KFall;
  if (NLeft >= 0) KF.N = NLeft;
  Grid g;
  AssignFeaturesToGrid(KF, g);
  KF.N = KFall.N;
  const KeyPoint* k = (const KeyPoint*)KF.kpsUn;
  std::vector<char> vpMatched(matchedIn, matchedIn + KF.N);
  for (int i = 0; i < KF.N; ++i) matchF[i] = -1;
  int nmatches = 0;
  for (int iMP = 0; iMP < nMP; iMP++) {
    if (!valid[iMP]) continue;
    const float* p3Dw = Pw + 3 * iMP;
    float p3Dc[3];
    rotateF(Tcw7, p3Dw, p3Dc);
    p3Dc[0] += Tcw7[4]; p3Dc[1] += Tcw7[5]; p3Dc[2] += Tcw7[6];
    if (p3Dc[2] < 0.0) continue;
    float u, v;
    if (manualProjection) {
      const float invz = 1 / p3Dc[2];
      const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
      u = KF.fx * x + KF.cx; v = KF.fy * y + KF.cy;
    } else if (cam8) {
      float uv[2];
      orc_kb8_project_f(cam8, p3Dc, uv);
      u = uv[0]; v = uv[1];
    } else {
      u = KF.fx * p3Dc[0] / p3Dc[2] + KF.cx; v = KF.fy * p3Dc[1] / p3Dc[2] + KF.cy;
    }
    if (!KFIsInImage(KF, u, v)) continue;
    const float maxDistance = 1.2f * mfMaxDistance[iMP], minDistance = 0.8f * mfMinDistance[iMP];
    const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
    const float dist = std::sqrt(PO[0] * PO[0] + PO[1] * PO[1] + PO[2] * PO[2]);
    if (dist < minDistance || dist > maxDistance) continue;
    const float* Pn = normal + 3 * iMP;
    if ((PO[0] * Pn[0] + PO[1] * Pn[1] + PO[2] * Pn[2]) < 0.5 * dist) continue;
    const int nPredictedLevel = PredictScaleKF(KF, mfMaxDistance[iMP], dist);
    const float radius = th * KF.scaleFactors[nPredictedLevel];
    const std::vector<size_t> vIndices = KFGetFeaturesInArea(KF, g, u, v, radius);
    if (vIndices.empty()) continue;
    const uint8_t* dMP = mpDesc + (size_t)iMP * 32;
    int bestDist = 256, bestIdx = -1;
    for (size_t q = 0; q < vIndices.size(); ++q) {
      const size_t idx = vIndices[q];
      if (vpMatched[idx]) continue;
      const int kpLevel = k[idx].octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const int dist2 = DescriptorDistance(dMP, KF.desc + idx * 32);
      if (dist2 < bestDist) { bestDist = dist2; bestIdx = (int)idx; }
    }
    if (bestDist <= TH_LOW * ratioHamming) {
      vpMatched[bestIdx] = 1;
      matchF[bestIdx] = iMP;
      nmatches++;
    }
  }
  return nmatches;
}

// One direction of ORBmatcher::SearchBySim3 (:1354-1425 / :1428-1499): map points of keyframe A (valid = has a MapPoint,
// !isBad(), !vbAlreadyMatched) -> camera A (TAw) -> camera B through the similarity S_BA -> window in keyframe B.
// Sim3 = RxSO3 quaternion (x y z w, squared norm = scale) + translation; Sophus rxso3.hpp:265-273.
static void sim3Map(const float* S8, const float* p, float* out) {
  const float qx = S8[0], qy = S8[1], qz = S8[2], qw = S8[3];
  const float scale = ((qx * qx + qy * qy) + qz * qz) + qw * qw;
  float a = qy * p[2] - qz * p[1], b = qz * p[0] - qx * p[2], c = qx * p[1] - qy * p[0];
  a += a; b += b; c += c;
  out[0] = scale * p[0] + (qw * a + (qy * c - qz * b)) + S8[4];
  out[1] = scale * p[1] + (qw * b + (qz * a - qx * c)) + S8[5];
  out[2] = scale * p[2] + (qw * c + (qx * b - qy * a)) + S8[6];
}
// NLeftB >= 0: B is a rig keyframe — its left features only (GetFeaturesInArea with bRight = false); the projection stays the pinhole formula (:1375-1379)
void SearchBySim3Dir(const orc_frame& Ball, const float* TAw7, const float* SBA8, int nA, const uint8_t* valid, const float* Pw,
                     const float* mfMaxDistance, const float* mfMinDistance, const uint8_t* mpDesc, float th, int* vnMatch, int NLeftB = -1) {
  orc_frame B = Ball;
  if (NLeftB >= 0) B.N = NLeftB;
  Grid g;
  AssignFeaturesToGrid(B, g);
  const KeyPoint* k = (const KeyPoint*)B.kpsUn;
  for (int i = 0; i < nA; i++) {
    vnMatch[i] = -1;
    if (!valid[i]) continue;
    float pA[3], pB[3];
    rotateF(TAw7, Pw + 3 * i, pA);
    pA[0] += TAw7[4]; pA[1] += TAw7[5]; pA[2] += TAw7[6];
    sim3Map(SBA8, pA, pB);
    if (pB[2] < 0.0) continue;
    const float invz = (float)(1.0 / pB[2]);
    const float x = pB[0] * invz, y = pB[1] * invz;
    const float u = B.fx * x + B.cx, v = B.fy * y + B.cy;
    if (!KFIsInImage(B, u, v)) continue;
    const float maxDistance = 1.2f * mfMaxDistance[i], minDistance = 0.8f * mfMinDistance[i];
    const float dist3D = std::sqrt(pB[0] * pB[0] + pB[1] * pB[1] + pB[2] * pB[2]);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const int nPredictedLevel = PredictScaleKF(B, mfMaxDistance[i], dist3D);
    const float radius = th * B.scaleFactors[nPredictedLevel];
    const std::vector<size_t> vIndices = KFGetFeaturesInArea(B, g, u, v, radius);
    if (vIndices.empty()) continue;
    const uint8_t* dMP = mpDesc + (size_t)i * 32;
    int bestDist = INT_MAX, bestIdx = -1;
    for (size_t q = 0; q < vIndices.size(); ++q) {
      const size_t idx = vIndices[q];
      if (k[idx].octave < nPredictedLevel - 1 || k[idx].octave > nPredictedLevel) continue;
      const int dist = DescriptorDistance(dMP, B.desc + idx * 32);
      if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
    }
    if (bestDist <= TH_HIGH) vnMatch[i] = bestIdx;
  }
}

// ---- Tracking.cc's host loops between the searches and PoseOptimization (checker for csrc/tracking.hip) -----------------------
// Frame::SetPose -> UpdatePoseMatrices (Frame.cc:541-585): mRcw = Tcw.rotationMatrix() (Eigen::Quaternionf::toRotationMatrix),
// mtcw, mOw = Tcw.inverse().translation() (Sophus: so3().inverse() * (translation() * -1))
void FrameSetPose(const float* Tcw7, float* mRcw, float* mtcw, float* mOw) {
  const float x = Tcw7[0], y = Tcw7[1], z = Tcw7[2], w = Tcw7[3];
  const float tx = 2.0f * x, ty = 2.0f * y, tz = 2.0f * z;
  const float twx = tx * w, twy = ty * w, twz = tz * w;
  const float txx = tx * x, txy = ty * x, txz = tz * x;
  const float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  mRcw[0] = 1.0f - (tyy + tzz); mRcw[1] = txy - twz; mRcw[2] = txz + twy;
  mRcw[3] = txy + twz; mRcw[4] = 1.0f - (txx + tzz); mRcw[5] = tyz - twx;
  mRcw[6] = txz - twy; mRcw[7] = tyz + twx; mRcw[8] = 1.0f - (txx + tyy);
  for (int i = 0; i < 3; ++i) mtcw[i] = Tcw7[4 + i];
  const float qc[4] = {-x, -y, -z, w};
  const float mt[3] = {Tcw7[4] * -1.0f, Tcw7[5] * -1.0f, Tcw7[6] * -1.0f};
  rotateF(qc, mt, mOw);
}
// Optimizer::PoseOptimization's edge fill (Optimizer.cc:803-905) over mvpMapPoints given as rows of a map-point table
void PoseEdges(const orc_frame& F, const float* invLevelSigma2, const int* frameMP, const float* mpXw, uint8_t* hasMP, float* obs,
               float* invSigma2, float* Xw) {
  const KeyPoint* kps = (const KeyPoint*)F.kpsUn;
  for (int i = 0; i < F.N; ++i) {
    const int mp = frameMP[i];
    hasMP[i] = mp >= 0;
    obs[3 * i] = obs[3 * i + 1] = 0.f; obs[3 * i + 2] = -1.f; invSigma2[i] = 0.f;
    Xw[3 * i] = Xw[3 * i + 1] = Xw[3 * i + 2] = 0.f;
    if (mp < 0) continue;
    obs[3 * i] = kps[i].x; obs[3 * i + 1] = kps[i].y; obs[3 * i + 2] = F.uRight ? F.uRight[i] : -1.f;
    invSigma2[i] = invLevelSigma2[kps[i].octave];
    for (int k = 0; k < 3; ++k) Xw[3 * i + k] = mpXw[3 * mp + k];
  }
}
// TrackWithMotionModel's discard loop (Tracking.cc:2716-2740) + SearchLocalPoints' first loop (:3117-3133); returns nmatches
int DiscardOutliers(int N, int* frameMP, uint8_t* outlier, int nMP, const uint8_t* mpHasObs, uint8_t* blocked, uint8_t* mpSeen,
                    int* nmatchesMap) {
  int nmatches = 0, nmap = 0;
  if (mpSeen) for (int i = 0; i < nMP; ++i) mpSeen[i] = 0;
  for (int i = 0; i < N; ++i) {
    if (blocked) blocked[i] = 0;
    const int mp = frameMP[i];
    if (mp < 0) continue;
    if (mpSeen) mpSeen[mp] = 1;                     // pMP->mnLastFrameSeen = mCurrentFrame.mnId (both branches)
    if (outlier[i]) { frameMP[i] = -1; outlier[i] = 0; }
    else {
      ++nmatches;
      if (mpHasObs[mp]) { ++nmap; if (blocked) blocked[i] = 1; }
    }
  }
  *nmatchesMap = nmap;
  return nmatches;
}

}  // namespace orc

using namespace orc;
extern "C" {
void orc_frame_set_pose(const float* Tcw7, float* Rcw, float* tcw, float* Ow) { FrameSetPose(Tcw7, Rcw, tcw, Ow); }
void orc_pose_edges(const orc_frame* F, const float* invLevelSigma2, const int* frameMP, const float* mpXw, uint8_t* hasMP, float* obs,
                    float* invSigma2, float* Xw) {
  PoseEdges(*F, invLevelSigma2, frameMP, mpXw, hasMP, obs, invSigma2, Xw);
}
int orc_discard_outliers(int N, int* frameMP, uint8_t* outlier, int nMP, const uint8_t* mpHasObs, uint8_t* blocked, uint8_t* mpSeen,
                         int* nmatchesMap) {
  return DiscardOutliers(N, frameMP, outlier, nMP, mpHasObs, blocked, mpSeen, nmatchesMap);
}

void orc_is_in_frustum(const orc_frame* F, const float* Rcw, const float* tcw, const float* Ow, int nMP, const float* Pw,
                       const float* normal, const float* maxDist, const float* minDist, float viewingCosLimit,
                       uint8_t* inView, float* projX, float* projY, float* projXR, float* depth, int* level,
                       float* viewCos) {
  isInFrustum(*F, Rcw, tcw, Ow, nMP, Pw, normal, maxDist, minDist, viewingCosLimit, inView, projX, projY, projXR, depth, level,
              viewCos);
}
int orc_search_by_projection_mps(const orc_frame* F, const uint8_t* fBlocked, int nMP, const uint8_t* inView,
                                 const uint8_t* isBad, const float* depth, const float* projX, const float* projY,
                                 const float* projXR, const int* level, const float* viewCos, const uint8_t* mpDesc,
                                 const uint8_t* mpHasObs, float th, int bFarPoints, float thFarPoints, float nnratio,
                                 int* matchF) {
  return SearchByProjectionMPs(*F, fBlocked, nMP, inView, isBad, depth, projX, projY, projXR, level, viewCos, mpDesc, mpHasObs,
                               th, bFarPoints != 0, thFarPoints, nnratio, matchF);
}
void orc_is_in_frustum_kb8(const orc_frame* F, const float* cam8, const float* mR, const float* mt, const float* twc, int nMP,
                           const float* Pw, const float* normal, const float* maxDist, const float* minDist,
                           float viewingCosLimit, uint8_t* inView, float* projX, float* projY, float* depth, int* level,
                           float* viewCos) {
  isInFrustumChecks(*F, cam8, mR, mt, twc, nMP, Pw, normal, maxDist, minDist, viewingCosLimit, inView, projX, projY, depth, level,
                    viewCos);
}
int orc_search_by_projection_mps_fisheye(const orc_frame* F, int Nleft, const int* l2r, const int* r2l, const uint8_t* fBlocked,
                                         int nMP, const uint8_t* inViewL, const uint8_t* inViewR, const uint8_t* isBad,
                                         const float* depthL, const float* projXL, const float* projYL, const int* levelL,
                                         const float* viewCosL, const float* projXR, const float* projYR, const int* levelR,
                                         const float* viewCosR, const uint8_t* mpDesc, const uint8_t* mpHasObs, float th,
                                         int bFarPoints, float thFarPoints, float nnratio, int* matchF) {
  return SearchByProjectionMPsFisheye(*F, Nleft, l2r, r2l, fBlocked, nMP, inViewL, inViewR, isBad, depthL, projXL, projYL, levelL,
                                      viewCosL, projXR, projYR, levelR, viewCosR, mpDesc, mpHasObs, th, bFarPoints != 0,
                                      thFarPoints, nnratio, matchF);
}
int orc_search_by_projection_last(const orc_frame* Cur, const uint8_t* curBlocked, const float* Tcw7, int nLast,
                                  const orc_keypoint* lastKpsUn, const uint8_t* lastValid, const float* lastXw,
                                  const uint8_t* lastMPdesc, const uint8_t* lastMPhasObs, float th, int bForward,
                                  int bBackward, int checkOri, int* matchCur) {
  return SearchByProjectionLast(*Cur, curBlocked, Tcw7, nLast, (const KeyPoint*)lastKpsUn, lastValid, lastXw, lastMPdesc,
                                lastMPhasObs, th, bForward != 0, bBackward != 0, checkOri != 0, matchCur);
}
int orc_search_by_projection_last_fisheye(const orc_frame* Cur, int NleftCur, const float* cam8, const float* Trl7,
                                          const uint8_t* curBlocked, const float* Tcw7, int nLast, const orc_keypoint* lastKps,
                                          const uint8_t* lastValid, const float* lastXw, const uint8_t* lastMPdesc,
                                          const uint8_t* lastMPhasObs, float th, int bForward, int bBackward, int checkOri,
                                          int* matchCur) {
  return SearchByProjectionLastFisheye(*Cur, NleftCur, cam8, Trl7, curBlocked, Tcw7, nLast, (const KeyPoint*)lastKps, lastValid,
                                       lastXw, lastMPdesc, lastMPhasObs, th, bForward != 0, bBackward != 0, checkOri != 0, matchCur);
}
int orc_search_by_projection_kf(const orc_frame* Cur, const uint8_t* curHasMP, const float* Tcw7, const float* Ow, int nKF,
                                const orc_keypoint* kfKpsUn, const uint8_t* kfValid, const float* Xw, const float* maxDist,
                                const float* minDist, const uint8_t* mpDesc, float th, int ORBdist, int checkOri, int* matchCur) {
  return SearchByProjectionKF(*Cur, curHasMP, Tcw7, Ow, nKF, (const KeyPoint*)kfKpsUn, kfValid, Xw, maxDist, minDist, mpDesc, th,
                              ORBdist, checkOri != 0, matchCur);
}
int orc_search_by_projection_kf_rig(const orc_frame* CurLeft, const float* cam8, const uint8_t* curHasMP, const float* Tcw7, const float* Ow, int nKF,
                                    const orc_keypoint* kfKps, const uint8_t* kfValid, const float* Xw, const float* maxDist,
                                    const float* minDist, const uint8_t* mpDesc, float th, int ORBdist, int checkOri, int* matchCur) {
  return SearchByProjectionKF(*CurLeft, curHasMP, Tcw7, Ow, nKF, (const KeyPoint*)kfKps, kfValid, Xw, maxDist, minDist, mpDesc, th,
                              ORBdist, checkOri != 0, matchCur, cam8);
}
int orc_search_for_initialization(int n1, const orc_keypoint* kps1, const uint8_t* desc1, const orc_frame* F2, float* prevMatched,
                                  int windowSize, float nnratio, int checkOri, int* matches12) {
  return SearchForInitialization(n1, (const KeyPoint*)kps1, desc1, *F2, prevMatched, windowSize, nnratio, checkOri != 0, matches12);
}
int orc_search_for_triangulation_fisheye(int n1, int NLeft1, const orc_keypoint* kps1, const uint8_t* desc1, const int* node1,
                                         const uint8_t* hasMP1, int n2, int NLeft2, const orc_keypoint* kps2, const uint8_t* desc2,
                                         const int* node2, const uint8_t* hasMP2, const float* levelSigma2, const float* camL8,
                                         const float* camR8, const float* T4, int bOnlyStereo, int bCoarse, int checkOri,
                                         int* matches12) {
  return SearchForTriangulationFisheye(n1, NLeft1, (const KeyPoint*)kps1, desc1, node1, hasMP1, n2, NLeft2, (const KeyPoint*)kps2,
                                       desc2, node2, hasMP2, levelSigma2, camL8, camR8, T4, bOnlyStereo != 0, bCoarse != 0,
                                       checkOri != 0, matches12);
}
void orc_fuse_search(const orc_frame* KF, const float* invLevelSigma2, const float* Tcw7, const float* Ow, int nMP, const uint8_t* valid,
                     const float* Pw, const float* normal, const float* maxDist, const float* minDist, const uint8_t* mpDesc, float th,
                     int sim3Form, int* bestIdx, int* bestDist) {
  FuseSearch(*KF, invLevelSigma2, Tcw7, Ow, nMP, valid, Pw, normal, maxDist, minDist, mpDesc, th, sim3Form != 0, bestIdx, bestDist);
}
void orc_fuse_search_rig(const orc_frame* KF, int NLeft, int bRight, const float* cam8, const float* invLevelSigma2, const float* Tcw7,
                         const float* Ow, int nMP, const uint8_t* valid, const float* Pw, const float* normal, const float* maxDist,
                         const float* minDist, const uint8_t* mpDesc, float th, int* bestIdx, int* bestDist) {
  FuseSearchRig(*KF, NLeft, bRight != 0, cam8, invLevelSigma2, Tcw7, Ow, nMP, valid, Pw, normal, maxDist, minDist, mpDesc, th, bestIdx, bestDist);
}
int orc_search_by_projection_sim3_rig(const orc_frame* KF, int NLeft, const float* cam8, const float* Tcw7, const float* Ow, int nMP, const uint8_t* valid,
                                      const float* Pw, const float* normal, const float* maxDist, const float* minDist, const uint8_t* mpDesc,
                                      const uint8_t* matchedIn, int th, float ratioHamming, int manualProjection, int* matchF) {
  return SearchByProjectionSim3(*KF, Tcw7, Ow, nMP, valid, Pw, normal, maxDist, minDist, mpDesc, matchedIn, th, ratioHamming, manualProjection != 0, matchF,
                                manualProjection ? nullptr : cam8, NLeft);
}
void orc_search_by_sim3_dir_rig(const orc_frame* B, int NLeftB, const float* TAw7, const float* SBA8, int nA, const uint8_t* valid, const float* Pw,
                                const float* maxDist, const float* minDist, const uint8_t* mpDesc, float th, int* vnMatch) {
  SearchBySim3Dir(*B, TAw7, SBA8, nA, valid, Pw, maxDist, minDist, mpDesc, th, vnMatch, NLeftB);
}
void orc_fuse_search_rig_sim3(const orc_frame* KF, int NLeft, const float* cam8, const float* invLevelSigma2, const float* Tcw7, const float* Ow, int nMP,
                              const uint8_t* valid, const float* Pw, const float* normal, const float* maxDist, const float* minDist,
                              const uint8_t* mpDesc, float th, int* bestIdx, int* bestDist) {
  FuseSearchRig(*KF, NLeft, false, cam8, invLevelSigma2, Tcw7, Ow, nMP, valid, Pw, normal, maxDist, minDist, mpDesc, th, bestIdx, bestDist, true);
}
int orc_search_by_projection_sim3(const orc_frame* KF, const float* Tcw7, const float* Ow, int nMP, const uint8_t* valid,
                                  const float* Pw, const float* normal, const float* maxDist, const float* minDist,
                                  const uint8_t* mpDesc, const uint8_t* matchedIn, int th, float ratioHamming, int manualProjection,
                                  int* matchF) {
  return SearchByProjectionSim3(*KF, Tcw7, Ow, nMP, valid, Pw, normal, maxDist, minDist, mpDesc, matchedIn, th, ratioHamming,
                                manualProjection != 0, matchF);
}
void orc_search_by_sim3_dir(const orc_frame* B, const float* TAw7, const float* SBA8, int nA, const uint8_t* valid, const float* Pw,
                            const float* maxDist, const float* minDist, const uint8_t* mpDesc, float th, int* vnMatch) {
  SearchBySim3Dir(*B, TAw7, SBA8, nA, valid, Pw, maxDist, minDist, mpDesc, th, vnMatch);
}
void orc_fundamental_f12(const float* K1, const float* K2, const float* R12, const float* t12, float* F12) {
  FundamentalF12(K1, K2, R12, t12, F12);
}
int orc_search_for_triangulation(int n1, const orc_keypoint* kps1, const uint8_t* desc1, const int* node1,
                                 const uint8_t* hasMP1, const float* uRight1, const float* sigma2_1, int n2,
                                 const orc_keypoint* kps2, const uint8_t* desc2, const int* node2, const uint8_t* hasMP2,
                                 const float* uRight2, const float* sigma2_2, const float* scaleFactors2, const float* F12,
                                 const float* ep, int bOnlyStereo, int bCoarse, int checkOri, int* matches12) {
  return SearchForTriangulation(n1, (const KeyPoint*)kps1, desc1, node1, hasMP1, uRight1, sigma2_1, n2, (const KeyPoint*)kps2,
                                desc2, node2, hasMP2, uRight2, sigma2_2, scaleFactors2, F12, ep, bOnlyStereo != 0,
                                bCoarse != 0, checkOri != 0, matches12);
}
}
