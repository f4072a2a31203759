// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path.
//
// CPU restatement of the reference's matchers on flattened (POD) inputs: ORBmatcher
// (/root/reference/src/ORBmatcher.cc) and the two per-frame stereo matchers in Frame.cc.  The reference
// operates on Frame/KeyFrame/MapPoint objects; the C-ABI boundary flattens exactly the fields each function
// reads, and these functions take the same flattened form so that HIP and oracle consume
// identical bytes.  PARITY UNPINNED where OpenCV defines the arithmetic (BFMatcher tie order, cv::norm).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#include "cvprims.h"
#include "matcher.h"
#include "orb_oracle.h"

namespace orc {

static const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;  // ORBmatcher.cc:35-37

// ORBmatcher.cc:1880-1894 (SWAR popcount on 8 x int32)
int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
  const int32_t* pa = reinterpret_cast<const int32_t*>(a);
  const int32_t* pb = reinterpret_cast<const int32_t*>(b);
  int dist = 0;
  for (int i = 0; i < 8; i++, pa++, pb++) {
    unsigned int v = *pa ^ *pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

// ORBmatcher.cc:1844-1876
void ComputeThreeMaxima(const std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) {
      max3 = max2; max2 = max1; max1 = s;
      ind3 = ind2; ind2 = ind1; ind1 = i;
    } else if (s > max2) {
      max3 = max2; max2 = s;
      ind3 = ind2; ind2 = i;
    } else if (s > max3) {
      max3 = s; ind3 = i;
    }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

// Frame::ComputeStereoMatches, Frame.cc:889-1047.  pyrL/pyrR: mvImagePyramid of the two extractors.
// Guards added where the reference has UB: row indices clamped to [0, nRows), empty vDistIdx.
void ComputeStereoMatches(int N, const KeyPoint* mvKeys, const uint8_t* mDescriptors, int Nr,
                          const KeyPoint* mvKeysRight, const uint8_t* mDescriptorsRight, const float* mvScaleFactors,
                          const float* mvInvScaleFactors, const std::vector<Img>& pyrL, const std::vector<Img>& pyrR,
                          float mbf, float mb, float* mvuRight, float* mvDepth) {
  for (int i = 0; i < N; ++i) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; }
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  const int nRows = pyrL[0].rows;
  std::vector<std::vector<size_t>> vRowIndices(nRows);
  for (int iR = 0; iR < Nr; iR++) {
    const KeyPoint& kp = mvKeysRight[iR];
    const float kpY = kp.y;
    const float r = 2.0f * mvScaleFactors[kp.octave];
    const int maxr = (int)std::ceil(kpY + r);
    const int minr = (int)std::floor(kpY - r);
    for (int yi = std::max(minr, 0); yi <= std::min(maxr, nRows - 1); yi++) vRowIndices[yi].push_back(iR);
  }
  const float minZ = mb;
  const float minD = 0;
  const float maxD = mbf / minZ;
  std::vector<std::pair<int, int>> vDistIdx;
  for (int iL = 0; iL < N; iL++) {
    const KeyPoint& kpL = mvKeys[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const int row = (int)vL;
    if (row < 0 || row >= nRows) continue;
    const std::vector<size_t>& vCandidates = vRowIndices[row];
    if (vCandidates.empty()) continue;
    const float minU = uL - maxD;
    const float maxU = uL - minD;
    if (maxU < 0) continue;
    int bestDist = TH_HIGH;
    size_t bestIdxR = 0;
    const uint8_t* dL = mDescriptors + (size_t)iL * 32;
    for (size_t iC = 0; iC < vCandidates.size(); iC++) {
      const size_t iR = vCandidates[iC];
      const KeyPoint& kpR = mvKeysRight[iR];
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      const float uR = kpR.x;
      if (uR >= minU && uR <= maxU) {
        const int dist = DescriptorDistance(dL, mDescriptorsRight + iR * 32);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist < thOrbDist) {
      const float uR0 = mvKeysRight[bestIdxR].x;
      const float scaleFactor = mvInvScaleFactors[kpL.octave];
      const float scaleduL = std::round(kpL.x * scaleFactor);
      const float scaledvL = std::round(kpL.y * scaleFactor);
      const float scaleduR0 = std::round(uR0 * scaleFactor);
      const int w = 5;
      const Img& IL = pyrL[kpL.octave];
      const Img& IRimg = pyrR[kpL.octave];
      int bestDistS = INT_MAX;
      int bestincR = 0;
      const int L = 5;
      float vDists[2 * 5 + 1];
      const float iniu = scaleduR0 + L - w;
      const float endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= IRimg.cols) continue;
      for (int incR = -L; incR <= +L; incR++) {
        // cv::norm(IL, IR, NORM_L1) over the 11x11 patches
        int sad = 0;
        for (int dy = -w; dy <= w; ++dy)
          for (int dx = -w; dx <= w; ++dx) {
            int a = IL.at((int)scaledvL + dy, (int)scaleduL + dx);
            int b = IRimg.at((int)scaledvL + dy, (int)scaleduR0 + incR + dx);
            sad += std::abs(a - b);
          }
        float dist = (float)(double)sad;
        if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;
      const float dist1 = vDists[L + bestincR - 1];
      const float dist2 = vDists[L + bestincR];
      const float dist3 = vDists[L + bestincR + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = mvScaleFactors[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) {
          disparity = 0.01;
          bestuR = uL - 0.01;
        }
        mvDepth[iL] = mbf / disparity;
        mvuRight[iL] = bestuR;
        vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
      }
    }
  }
  if (vDistIdx.empty()) return;  // reference: UB (vDistIdx[0] on an empty vector)
  std::sort(vDistIdx.begin(), vDistIdx.end());
  const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
    if (vDistIdx[i].first < thDist) break;
    mvuRight[vDistIdx[i].second] = -1;
    mvDepth[vDistIdx[i].second] = -1;
  }
}

// cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, k=2)  (Frame.cc:46, :1242): exhaustive, ascending
// distance, equal distances keep the lower train index first.
void knnMatch2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx /*[nq][2]*/, int* dist /*[nq][2]*/) {
  for (int i = 0; i < nq; ++i) {
    int b0 = INT_MAX, b1 = INT_MAX, i0 = -1, i1 = -1;
    for (int j = 0; j < nt; ++j) {
      const int d = DescriptorDistance(q + (size_t)i * 32, t + (size_t)j * 32);
      if (d < b0) { b1 = b0; i1 = i0; b0 = d; i0 = j; }
      else if (d < b1) { b1 = d; i1 = j; }
    }
    idx[2 * i] = i0; idx[2 * i + 1] = i1;
    dist[2 * i] = i0 < 0 ? -1 : b0; dist[2 * i + 1] = i1 < 0 ? -1 : b1;
  }
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches), ORBmatcher.cc:218-395, non-fisheye branch
// (F.Nleft == -1).  FeatureVectors are given as one node id per feature (the DBoW2 contract: map nodeId ->
// ascending feature indices; a negative node id = feature absent from the FeatureVector).  kfHasMP[i] != 0 <=>
// vpMapPointsKF[i] != NULL && !isBad().  matchF[j] = KF feature index matched to frame feature j, or -1.
// FNleft = F.Nleft (-1: pinhole frame; otherwise features [0, FNleft) are the left camera's, the rest the right
// camera's, and the second set of best / second-best variables of :262-299 and the nested right assignment of
// :333-365 apply — including its `|| true` and its dependence on the LEFT best distance).
int SearchByBoW(int nKF, const uint8_t* descKF, const float* angleKF, const uint8_t* kfHasMP, const int* nodeKF,
                int nF, const uint8_t* descF, const float* angleF, const int* nodeF, float mfNNratio,
                bool mbCheckOrientation, int* matchF, int FNleft = -1) {
  std::map<int, std::vector<unsigned>> vFeatVecKF, FFeatVec;
  for (int i = 0; i < nKF; ++i) if (nodeKF[i] >= 0) vFeatVecKF[nodeKF[i]].push_back(i);
  for (int i = 0; i < nF; ++i) if (nodeF[i] >= 0) FFeatVec[nodeF[i]].push_back(i);
  for (int i = 0; i < nF; ++i) matchF[i] = -1;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  auto KFit = vFeatVecKF.begin(), KFend = vFeatVecKF.end();
  auto Fit = FFeatVec.begin(), Fend = FFeatVec.end();
  while (KFit != KFend && Fit != Fend) {
    if (KFit->first == Fit->first) {
      const std::vector<unsigned>& vIndicesKF = KFit->second;
      const std::vector<unsigned>& vIndicesF = Fit->second;
      for (size_t iKF = 0; iKF < vIndicesKF.size(); iKF++) {
        const unsigned realIdxKF = vIndicesKF[iKF];
        if (!kfHasMP[realIdxKF]) continue;
        const uint8_t* dKF = descKF + (size_t)realIdxKF * 32;
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        int bestDist1R = 256, bestIdxFR = -1, bestDist2R = 256;
        for (size_t iF = 0; iF < vIndicesF.size(); iF++) {
          const unsigned realIdxF = vIndicesF[iF];
          if (matchF[realIdxF] >= 0) continue;
          const int dist = DescriptorDistance(dKF, descF + (size_t)realIdxF * 32);
          if (FNleft == -1) {
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
            else if (dist < bestDist2) { bestDist2 = dist; }
          } else {
            if ((int)realIdxF < FNleft && dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
            else if ((int)realIdxF < FNleft && dist < bestDist2) { bestDist2 = dist; }
            if ((int)realIdxF >= FNleft && dist < bestDist1R) { bestDist2R = bestDist1R; bestDist1R = dist; bestIdxFR = realIdxF; }
            else if ((int)realIdxF >= FNleft && dist < bestDist2R) { bestDist2R = dist; }
          }
        }
        if (bestDist1 <= TH_LOW) {
          if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
            matchF[bestIdxF] = (int)realIdxKF;
            if (mbCheckOrientation) {
              float rot = angleKF[realIdxKF] - angleF[bestIdxF];
              if (rot < 0.0) rot += 360.0f;
              int bin = (int)std::round(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              rotHist[bin].push_back(bestIdxF);
            }
            nmatches++;
          }
          if (bestDist1R <= TH_LOW) {
            if (static_cast<float>(bestDist1R) < mfNNratio * static_cast<float>(bestDist2R) || true) {
              matchF[bestIdxFR] = (int)realIdxKF;
              if (mbCheckOrientation) {
                float rot = angleKF[realIdxKF] - angleF[bestIdxFR];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdxFR);
              }
              nmatches++;
            }
          }
        }
      }
      KFit++;
      Fit++;
    } else if (KFit->first < Fit->first) {
      KFit = vFeatVecKF.lower_bound(Fit->first);
    } else {
      Fit = FFeatVec.lower_bound(KFit->first);
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        matchF[rotHist[i][j]] = -1;
        nmatches--;
      }
    }
  }
  return nmatches;
}

// ---- ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (:702-819) ------------
// hasMP[i] != 0 <=> vpMapPoints[i] && !isBad().  nValid = mvKeysUn.size() (features at or beyond it are skipped on
// fisheye keyframes, :734, :751).  matches12[idx1] = idx2 of the matched pKF2 feature, or -1.
int SearchByBoWKFKF(int n1, int nValid1, const uint8_t* desc1, const float* angle1, const uint8_t* hasMP1, const int* node1, int n2,
                    int nValid2, const uint8_t* desc2, const float* angle2, const uint8_t* hasMP2, const int* node2,
                    float mfNNratio, bool mbCheckOrientation, int* matches12) {
  std::map<int, std::vector<unsigned>> vFeatVec1, vFeatVec2;
  for (int i = 0; i < n1; ++i) if (node1[i] >= 0) vFeatVec1[node1[i]].push_back(i);
  for (int i = 0; i < n2; ++i) if (node2[i] >= 0) vFeatVec2[node2[i]].push_back(i);
  for (int i = 0; i < n1; ++i) matches12[i] = -1;
  std::vector<char> vbMatched2(n2, 0);
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  int nmatches = 0;
  auto f1it = vFeatVec1.begin(), f1end = vFeatVec1.end();
  auto f2it = vFeatVec2.begin(), f2end = vFeatVec2.end();
  while (f1it != f1end && f2it != f2end) {
    if (f1it->first == f2it->first) {
      for (size_t i1 = 0, iend1 = f1it->second.size(); i1 < iend1; i1++) {
        const size_t idx1 = f1it->second[i1];
        if ((int)idx1 >= nValid1) continue;
        if (!hasMP1[idx1]) continue;
        const uint8_t* d1 = desc1 + idx1 * 32;
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (size_t i2 = 0, iend2 = f2it->second.size(); i2 < iend2; i2++) {
          const size_t idx2 = f2it->second[i2];
          if ((int)idx2 >= nValid2) continue;
          if (vbMatched2[idx2] || !hasMP2[idx2]) continue;
          const int dist = DescriptorDistance(d1, desc2 + idx2 * 32);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }
          else if (dist < bestDist2) { bestDist2 = dist; }
        }
        if (bestDist1 < TH_LOW) {
          if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
            matches12[idx1] = bestIdx2;
            vbMatched2[bestIdx2] = 1;
            if (mbCheckOrientation) {
              float rot = angle1[idx1] - angle2[bestIdx2];
              if (rot < 0.0) rot += 360.0f;
              int bin = (int)std::round(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              rotHist[bin].push_back((int)idx1);
            }
            nmatches++;
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
        matches12[rotHist[i][j]] = -1;
        nmatches--;
      }
    }
  }
  return nmatches;
}

// ---- MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:367-435), the search: index of the observed descriptor with
// the least median Hamming distance to the others (first minimum).  vDescriptors = N rows of 32 bytes in observation order.
int DistinctiveDescriptor(const uint8_t* vDescriptors, int N) {
  if (N <= 0) return -1;
  std::vector<std::vector<int>> Distances(N, std::vector<int>(N, 0));
  for (int i = 0; i < N; i++) {
    Distances[i][i] = 0;
    for (int j = i + 1; j < N; j++) {
      const int distij = DescriptorDistance(vDescriptors + (size_t)i * 32, vDescriptors + (size_t)j * 32);
      Distances[i][j] = distij;
      Distances[j][i] = distij;
    }
  }
  int BestMedian = INT_MAX, BestIdx = 0;
  for (int i = 0; i < N; i++) {
    std::vector<int> vDists(Distances[i].begin(), Distances[i].end());
    std::sort(vDists.begin(), vDists.end());
    const int median = vDists[(size_t)(0.5 * (N - 1))];
    if (median < BestMedian) { BestMedian = median; BestIdx = i; }
  }
  return BestIdx;
}

// DBoW2 TemplatedVocabulary::transform(feature, word id, weight, nid, levelsup)
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259): greedy descent, at each level the child with the
// smallest Hamming distance (first minimum), nid = the ancestor at level (L - levelsup).  The tree is given as
// a complete k-ary array: node 0 = root, children of node n are firstChild[n] .. firstChild[n]+k-1
// (firstChild < 0 for leaves), one 32-byte descriptor per node.
void BowTransform(const uint8_t* feat, int n, const uint8_t* nodeDesc, const int* firstChild, int k, int L, int levelsup,
                  int* wordId, int* nodeId, const int* childCount = nullptr) {
  const int nid_level = L - levelsup;
  for (int f = 0; f < n; ++f) {
    int final_id = 0, current_level = 0, nid = 0;
    if (nid_level <= 0) nid = 0;
    do {
      ++current_level;
      const int c0 = firstChild[final_id];
      int best_d = DescriptorDistance(feat + (size_t)f * 32, nodeDesc + (size_t)c0 * 32);
      int best = c0;
      const int nc = childCount ? childCount[final_id] : k;
      for (int c = c0 + 1; c < c0 + nc; ++c) {
        int d = DescriptorDistance(feat + (size_t)f * 32, nodeDesc + (size_t)c * 32);
        if (d < best_d) { best_d = d; best = c; }
      }
      final_id = best;
      if (current_level == nid_level) nid = final_id;
    } while (firstChild[final_id] >= 0);
    wordId[f] = final_id;
    nodeId[f] = nid;
  }
}

}  // namespace orc

using namespace orc;
extern "C" {

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return DescriptorDistance(a, b); }

void orc_three_maxima(const int* counts, int L, int* ind) {
  std::vector<std::vector<int>> h(L);
  for (int i = 0; i < L; ++i) h[i].resize(counts[i]);
  ind[0] = ind[1] = ind[2] = -1;
  ComputeThreeMaxima(h.data(), L, ind[0], ind[1], ind[2]);
}

void orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx, int* dist) { knnMatch2(q, nq, t, nt, idx, dist); }

int orc_search_by_bow(int nKF, const uint8_t* descKF, const float* angleKF, const uint8_t* kfHasMP, const int* nodeKF,
                      int nF, const uint8_t* descF, const float* angleF, const int* nodeF, float nnratio, int checkOri,
                      int* matchF) {
  return SearchByBoW(nKF, descKF, angleKF, kfHasMP, nodeKF, nF, descF, angleF, nodeF, nnratio, checkOri != 0, matchF);
}

void orc_distinctive_descriptors(int nMP, const int* start, const uint8_t* desc, int* bestIdx) {
  for (int m = 0; m < nMP; ++m) bestIdx[m] = DistinctiveDescriptor(desc + (size_t)start[m] * 32, start[m + 1] - start[m]);
}

int orc_search_by_bow_kfkf(int n1, int nValid1, const uint8_t* desc1, const float* angle1, const uint8_t* hasMP1, const int* node1,
                           int n2, int nValid2, const uint8_t* desc2, const float* angle2, const uint8_t* hasMP2, const int* node2,
                           float nnratio, int checkOri, int* matches12) {
  return SearchByBoWKFKF(n1, nValid1, desc1, angle1, hasMP1, node1, n2, nValid2, desc2, angle2, hasMP2, node2, nnratio, checkOri != 0,
                         matches12);
}

int orc_search_by_bow_fisheye(int nKF, const uint8_t* descKF, const float* angleKF, const uint8_t* kfHasMP, const int* nodeKF,
                              int nF, int FNleft, const uint8_t* descF, const float* angleF, const int* nodeF, float nnratio,
                              int checkOri, int* matchF) {
  return SearchByBoW(nKF, descKF, angleKF, kfHasMP, nodeKF, nF, descF, angleF, nodeF, nnratio, checkOri != 0, matchF, FNleft);
}

void orc_bow_transform_tree(const uint8_t* feat, int n, const uint8_t* nodeDesc, const int* firstChild, const int* childCount, int L,
                            int levelsup, int* wordId, int* nodeId) {
  BowTransform(feat, n, nodeDesc, firstChild, 0, L, levelsup, wordId, nodeId, childCount);
}
void orc_bow_transform(const uint8_t* feat, int n, const uint8_t* nodeDesc, const int* firstChild, int k, int L,
                       int levelsup, int* wordId, int* nodeId) {
  BowTransform(feat, n, nodeDesc, firstChild, k, L, levelsup, wordId, nodeId);
}
}

// TemplatedVocabulary::transform(features, BowVector& v, FeatureVector& fv, levelsup): the BowVector
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1190, BowVector.cpp:34-84, ScoringObject.h:60-92).  leaf[i] = leaf node of
// feature i (from orc_bow_transform*), nodeWordId (optional) maps it to the WordId, nodeWeight = WordValue (double).
// weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; scoring: 0 L1_NORM, 1 L2_NORM, 2 CHI_SQUARE, 3 KL, 4 BHATTACHARYYA, 5 DOT_PRODUCT.
#include <map>
extern "C" int orc_bow_vector(int n, const int* leaf, const int* nodeWordId, const double* nodeWeight, int weighting, int scoring,
                              int* outWord, double* outValue) {
  std::map<unsigned, double> v;
  const bool must = scoring != 5;            // __SCORING_CLASS(DotProductScoring, false, L1)
  const bool l2 = scoring == 1;              // __SCORING_CLASS(L2Scoring, true, L2); every other object normalises with L1
  for (int i = 0; i < n; ++i) {
    if (leaf[i] < 0) continue;
    const double w = nodeWeight[leaf[i]];
    if (!(w > 0)) continue;                  // "not stopped"
    const unsigned id = (unsigned)(nodeWordId ? nodeWordId[leaf[i]] : leaf[i]);
    auto it = v.lower_bound(id);
    if (weighting == 0 || weighting == 1) {  // addWeight
      if (it != v.end() && !(id < it->first)) it->second += w; else v.insert(it, {id, w});
    } else {                                 // addIfNotExist
      if (it == v.end() || id < it->first) v.insert(it, {id, w});
    }
  }
  if ((weighting == 0 || weighting == 1) && !v.empty() && !must) {
    const double nd = (double)v.size();
    for (auto& kv : v) kv.second /= nd;
  }
  if (must) {                                // BowVector::normalize
    double norm = 0.0;
    if (!l2) { for (auto& kv : v) norm += std::fabs(kv.second); }
    else { for (auto& kv : v) norm += kv.second * kv.second; norm = std::sqrt(norm); }
    if (norm > 0.0) for (auto& kv : v) kv.second /= norm;
  }
  int k = 0;
  for (auto& kv : v) { outWord[k] = (int)kv.first; outValue[k] = kv.second; ++k; }
  return k;
}
