// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path.
//
// CPU restatement of ORB_SLAM3::ORBextractor (reference: /root/reference/src/ORBextractor.cc,
// include/ORBextractor.h).  Each function cites the lines it follows.  OpenCV primitives come from
// cvprims.cc (published OpenCV 4.x algorithms, PARITY UNPINNED — see cvprims.h): the reference cannot be
// built here (OpenCV/Eigen absent) and ships no tests or golden vectors for this path, so the fixtures in
// tests/golden are oracle_* fixtures: they pin GPU == oracle, not oracle == OpenCV.
#include <algorithm>
#include <list>
#include <utility>
#include <vector>

#include "cvprims.h"
#include "matcher.h"
#include "orb_oracle.h"

namespace orc {

static const int PATCH_SIZE = 31;       // ORBextractor.cc:71
static const int HALF_PATCH_SIZE = 15;  // :72
static const int EDGE_THRESHOLD = 19;   // :73

static const int bit_pattern_31_[256 * 4] = {
#include "../morb_slam_amd/csrc/orb_pattern.inc"
};

struct Pt { int x, y; };

// ORBextractor.cc:75-99
static float IC_Angle(const Img& image, float ptx, float pty, const std::vector<int>& u_max) {
  int m_01 = 0, m_10 = 0;
  const uint8_t* center = &image.at(cvRound(pty), cvRound(ptx));
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  int step = image.step;
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    int d = u_max[v];
    for (int u = -d; u <= d; ++u) {
      int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fastAtan2((float)m_01, (float)m_10);
}

// ORBextractor.cc:101-145.  cos/sin are the float overloads (std::cos(float) -> cosf).
static const float factorPI = (float)(3.14159265358979323846 / 180.f);
static void computeOrbDescriptor(const KeyPoint& kpt, const Img& img, const Pt* pattern, uint8_t* desc) {
  float angle = (float)kpt.angle * factorPI;
  float a = cosf_glibc(angle), b = sinf_glibc(angle);
  const uint8_t* center = &img.at(cvRound(kpt.y), cvRound(kpt.x));
  const int step = img.step;
  auto get = [&](int idx) -> int {
    // cvRound(pattern.x*b + pattern.y*a)*step + cvRound(pattern.x*a - pattern.y*b): float products and
    // sum, no contraction (oracle is built -ffp-contract=off; see DESIGN.md "FP conventions").
    float fy = pattern[idx].x * b + pattern[idx].y * a;
    float fx = pattern[idx].x * a - pattern[idx].y * b;
    return center[cvRound(fy) * step + cvRound(fx)];
  };
  for (int i = 0; i < 32; ++i, pattern += 16) {
    int val = 0;
    for (int k = 0; k < 8; ++k) {
      int t0 = get(2 * k), t1 = get(2 * k + 1);
      val |= (t0 < t1) << k;
    }
    desc[i] = (uint8_t)val;
  }
}

// include/ORBextractor.h:30-42
struct ExtractorNode {
  std::vector<KeyPoint> vKeys;
  Pt UL, UR, BL, BR;
  std::list<ExtractorNode>::iterator lit;
  bool bNoMore = false;
  void DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4);
};

// ORBextractor.cc:475-523
void ExtractorNode::DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4) {
  const int halfX = (int)std::ceil(static_cast<float>(UR.x - UL.x) / 2);
  const int halfY = (int)std::ceil(static_cast<float>(BR.y - UL.y) / 2);
  n1.UL = UL;
  n1.UR = Pt{UL.x + halfX, UL.y};
  n1.BL = Pt{UL.x, UL.y + halfY};
  n1.BR = Pt{UL.x + halfX, UL.y + halfY};
  n2.UL = n1.UR;
  n2.UR = UR;
  n2.BL = n1.BR;
  n2.BR = Pt{UR.x, UL.y + halfY};
  n3.UL = n1.BL;
  n3.UR = n1.BR;
  n3.BL = BL;
  n3.BR = Pt{n1.BR.x, BL.y};
  n4.UL = n3.UR;
  n4.UR = n2.BR;
  n4.BL = n3.BR;
  n4.BR = BR;
  for (size_t i = 0; i < vKeys.size(); i++) {
    const KeyPoint& kp = vKeys[i];
    if (kp.x < n1.UR.x) {
      if (kp.y < n1.BR.y) n1.vKeys.push_back(kp); else n3.vKeys.push_back(kp);
    } else if (kp.y < n1.BR.y)
      n2.vKeys.push_back(kp);
    else
      n4.vKeys.push_back(kp);
  }
  if (n1.vKeys.size() == 1) n1.bNoMore = true;
  if (n2.vKeys.size() == 1) n2.bNoMore = true;
  if (n3.vKeys.size() == 1) n3.bNoMore = true;
  if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

// ORBextractor.cc:525-538
static bool compareNodes(std::pair<int, ExtractorNode*>& e1, std::pair<int, ExtractorNode*>& e2) {
  if (e1.first < e2.first) return true;
  if (e1.first > e2.first) return false;
  return e1.second->UL.x < e2.second->UL.x;
}

// ORBextractor.cc:540-738.  std::list / std::sort (libstdc++) on purpose: the final list order and the
// tie behaviour of std::sort define the output order.
std::vector<KeyPoint> DistributeOctTree(const std::vector<KeyPoint>& vToDistributeKeys, int minX, int maxX,
                                        int minY, int maxY, int N) {
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<ExtractorNode> lNodes;
  std::vector<ExtractorNode*> vpIniNodes(nIni);
  for (int i = 0; i < nIni; i++) {
    ExtractorNode ni;
    ni.UL = Pt{(int)(hX * static_cast<float>(i)), 0};
    ni.UR = Pt{(int)(hX * static_cast<float>(i + 1)), 0};
    ni.BL = Pt{ni.UL.x, maxY - minY};
    ni.BR = Pt{ni.UR.x, maxY - minY};
    lNodes.push_back(ni);
    vpIniNodes[i] = &lNodes.back();
  }
  for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
    const KeyPoint& kp = vToDistributeKeys[i];
    vpIniNodes[(size_t)(kp.x / hX)]->vKeys.push_back(kp);
  }
  auto lit = lNodes.begin();
  while (lit != lNodes.end()) {
    if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
    else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
    else lit++;
  }
  bool bFinish = false;
  std::vector<std::pair<int, ExtractorNode*>> vSizeAndPointerToNode;
  auto pushChild = [&](ExtractorNode& n, int* nToExpand) {
    if (n.vKeys.size() > 0) {
      lNodes.push_front(n);
      if (n.vKeys.size() > 1) {
        if (nToExpand) (*nToExpand)++;
        vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
        lNodes.front().lit = lNodes.begin();
      }
    }
  };
  while (!bFinish) {
    int prevSize = (int)lNodes.size();
    lit = lNodes.begin();
    int nToExpand = 0;
    vSizeAndPointerToNode.clear();
    while (lit != lNodes.end()) {
      if (lit->bNoMore) { lit++; continue; }
      ExtractorNode n1, n2, n3, n4;
      lit->DivideNode(n1, n2, n3, n4);
      pushChild(n1, &nToExpand);
      pushChild(n2, &nToExpand);
      pushChild(n3, &nToExpand);
      pushChild(n4, &nToExpand);
      lit = lNodes.erase(lit);
    }
    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
      bFinish = true;
    } else if (((int)lNodes.size() + nToExpand * 3) > N) {
      while (!bFinish) {
        prevSize = (int)lNodes.size();
        std::vector<std::pair<int, ExtractorNode*>> vPrev = vSizeAndPointerToNode;
        vSizeAndPointerToNode.clear();
        std::sort(vPrev.begin(), vPrev.end(), compareNodes);
        for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
          ExtractorNode n1, n2, n3, n4;
          vPrev[j].second->DivideNode(n1, n2, n3, n4);
          pushChild(n1, nullptr);
          pushChild(n2, nullptr);
          pushChild(n3, nullptr);
          pushChild(n4, nullptr);
          lNodes.erase(vPrev[j].second->lit);
          if ((int)lNodes.size() >= N) break;
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
      }
    }
  }
  std::vector<KeyPoint> vResultKeys;
  for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
    std::vector<KeyPoint>& vNodeKeys = it->vKeys;
    KeyPoint* pKP = &vNodeKeys[0];
    float maxResponse = pKP->response;
    for (size_t k = 1; k < vNodeKeys.size(); k++)
      if (vNodeKeys[k].response > maxResponse) { pKP = &vNodeKeys[k]; maxResponse = vNodeKeys[k].response; }
    vResultKeys.push_back(*pKP);
  }
  return vResultKeys;
}

struct Extractor {
  int nfeatures, nlevels, iniThFAST, minThFAST;
  float scaleFactor;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<int> mnFeaturesPerLevel, umax;
  std::vector<Pt> pattern;
  // per-call state (mvImagePyramid is public state in the reference: ORBextractor.h:76)
  std::vector<std::vector<uint8_t>> pyrStore;  // padded buffers
  std::vector<Img> mvImagePyramid;             // interior ROIs
  std::vector<std::vector<uint8_t>> blurStore;
  std::vector<std::vector<KeyPoint>> candidates, levelKeys;

  // ORBextractor.cc:406-464
  Extractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
      : nfeatures(_nfeatures), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST),
        scaleFactor(_scaleFactor) {
    mvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f;
    mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
      mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor;
      mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
      mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
      mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mnFeaturesPerLevel.resize(nlevels);
    float factor = 1.0f / scaleFactor;
    float nDesiredFeaturesPerScale =
        nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
      mnFeaturesPerLevel[level] = cvRound(nDesiredFeaturesPerScale);
      sumFeatures += mnFeaturesPerLevel[level];
      nDesiredFeaturesPerScale *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);
    const Pt* pattern0 = (const Pt*)bit_pattern_31_;
    pattern.assign(pattern0, pattern0 + 512);
    umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
    int vmin = cvCeil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRound(std::sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  // ORBextractor.cc:1088-1112
  void ComputePyramid(const Img& image) {
    pyrStore.resize(nlevels);
    mvImagePyramid.resize(nlevels);
    for (int level = 0; level < nlevels; ++level) {
      float scale = mvInvScaleFactor[level];
      int sw = cvRound((float)image.cols * scale), sh = cvRound((float)image.rows * scale);
      int ww = sw + EDGE_THRESHOLD * 2, wh = sh + EDGE_THRESHOLD * 2;
      pyrStore[level].assign((size_t)ww * wh, 0);
      Img temp{pyrStore[level].data(), ww, wh, ww};
      mvImagePyramid[level] = temp.roi(EDGE_THRESHOLD, EDGE_THRESHOLD, sw, sh);
      if (level != 0) {
        resizeLinear8u(mvImagePyramid[level - 1], mvImagePyramid[level]);
        copyMakeBorder101(mvImagePyramid[level], temp, EDGE_THRESHOLD);
      } else {
        copyMakeBorder101(image, temp, EDGE_THRESHOLD);
      }
    }
  }

  // ORBextractor.cc:740-844
  void ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& allKeypoints) {
    allKeypoints.assign(nlevels, {});
    candidates.assign(nlevels, {});
    const float W = 35;
    for (int level = 0; level < nlevels; ++level) {
      const int minBorderX = EDGE_THRESHOLD - 3;
      const int minBorderY = minBorderX;
      const int maxBorderX = mvImagePyramid[level].cols - EDGE_THRESHOLD + 3;
      const int maxBorderY = mvImagePyramid[level].rows - EDGE_THRESHOLD + 3;
      std::vector<KeyPoint> vToDistributeKeys;
      const float width = (float)(maxBorderX - minBorderX);
      const float height = (float)(maxBorderY - minBorderY);
      const int nCols = (int)(width / W);
      const int nRows = (int)(height / W);
      const int wCell = (int)std::ceil(width / nCols);
      const int hCell = (int)std::ceil(height / nRows);
      for (int i = 0; i < nRows; i++) {
        const float iniY = (float)(minBorderY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBorderY - 3) continue;
        if (maxY > maxBorderY) maxY = (float)maxBorderY;
        for (int j = 0; j < nCols; j++) {
          const float iniX = (float)(minBorderX + j * wCell);
          float maxX = iniX + wCell + 6;
          if (iniX >= maxBorderX - 6) continue;
          if (maxX > maxBorderX) maxX = (float)maxBorderX;
          std::vector<KeyPoint> vKeysCell;
          Img cell = mvImagePyramid[level].roi((int)iniX, (int)iniY, (int)maxX - (int)iniX, (int)maxY - (int)iniY);
          fast9_16(cell, vKeysCell, iniThFAST, true);
          if (vKeysCell.empty()) fast9_16(cell, vKeysCell, minThFAST, true);
          for (auto& kp : vKeysCell) {
            kp.x += j * wCell;
            kp.y += i * hCell;
            vToDistributeKeys.push_back(kp);
          }
        }
      }
      candidates[level] = vToDistributeKeys;
      std::vector<KeyPoint>& keypoints = allKeypoints[level];
      keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                    mnFeaturesPerLevel[level]);
      const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
      for (auto& kp : keypoints) {
        kp.x += minBorderX;
        kp.y += minBorderY;
        kp.octave = level;
        kp.size = (float)scaledPatchSize;
      }
    }
    for (int level = 0; level < nlevels; ++level)
      for (auto& kp : allKeypoints[level]) kp.angle = IC_Angle(mvImagePyramid[level], kp.x, kp.y, umax);
  }

  // ORBextractor.cc:1006-1086
  int operator()(const Img& image, std::vector<KeyPoint>& _keypoints, std::vector<uint8_t>& _descriptors,
                 int lap0, int lap1) {
    if (image.data == nullptr || image.cols == 0 || image.rows == 0) return -1;
    ComputePyramid(image);
    std::vector<std::vector<KeyPoint>> allKeypoints;
    ComputeKeyPointsOctTree(allKeypoints);
    levelKeys = allKeypoints;
    int nkeypoints = 0;
    for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
    _descriptors.assign((size_t)nkeypoints * 32, 0);
    _keypoints.assign(nkeypoints, KeyPoint{0, 0, 0, -1, 0, 0, -1});
    blurStore.assign(nlevels, {});
    int monoIndex = 0, stereoIndex = nkeypoints - 1;
    for (int level = 0; level < nlevels; ++level) {
      std::vector<KeyPoint>& keypoints = allKeypoints[level];
      int nkeypointsLevel = (int)keypoints.size();
      if (nkeypointsLevel == 0) continue;
      // clone() => continuous, the pad is not visible to the blur (:1049-1050)
      const Img& lv = mvImagePyramid[level];
      std::vector<uint8_t> clone((size_t)lv.cols * lv.rows);
      for (int y = 0; y < lv.rows; ++y) memcpy(&clone[(size_t)y * lv.cols], lv.ptr(y), lv.cols);
      blurStore[level].resize(clone.size());
      Img work{clone.data(), lv.cols, lv.rows, lv.cols};
      Img blurred{blurStore[level].data(), lv.cols, lv.rows, lv.cols};
      gaussianBlur7x7s2(work, blurred);
      std::vector<uint8_t> desc((size_t)nkeypointsLevel * 32);
      for (int i = 0; i < nkeypointsLevel; ++i)
        computeOrbDescriptor(keypoints[i], blurred, pattern.data(), &desc[(size_t)i * 32]);
      float scale = mvScaleFactor[level];
      int i = 0;
      for (auto& kp : keypoints) {
        if (level != 0) { kp.x *= scale; kp.y *= scale; }
        if (kp.x >= lap0 && kp.x <= lap1) {
          _keypoints.at(stereoIndex) = kp;
          memcpy(&_descriptors[(size_t)stereoIndex * 32], &desc[(size_t)i * 32], 32);
          stereoIndex--;
        } else {
          _keypoints.at(monoIndex) = kp;
          memcpy(&_descriptors[(size_t)monoIndex * 32], &desc[(size_t)i * 32], 32);
          monoIndex++;
        }
        i++;
      }
    }
    return monoIndex;
  }
};

}  // namespace orc

// ---- C API (ctypes-friendly) ------------------------------------------------------------------------
using namespace orc;
extern "C" {

orc_extractor* orc_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST) {
  return reinterpret_cast<orc_extractor*>(new Extractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST));
}
void orc_extractor_destroy(orc_extractor* e) { delete reinterpret_cast<Extractor*>(e); }

int orc_extract(orc_extractor* h, const uint8_t* img, int w, int h_, int stride, int lap0, int lap1,
                orc_keypoint* kps, uint8_t* desc, int cap, int* n_out) {
  Extractor* e = reinterpret_cast<Extractor*>(h);
  Img image{const_cast<uint8_t*>(img), w, h_, stride};
  std::vector<KeyPoint> k;
  std::vector<uint8_t> d;
  int mono = (*e)(image, k, d, lap0, lap1);
  if (mono < 0) { *n_out = 0; return -1; }
  int n = (int)k.size();
  *n_out = n;
  if (n > cap) return -2;
  if (n) {
    memcpy(kps, k.data(), (size_t)n * sizeof(KeyPoint));
    memcpy(desc, d.data(), (size_t)n * 32);
  }
  return mono;
}

void orc_extractor_tables(orc_extractor* h, float* scale, float* invScale, float* sigma2, float* invSigma2,
                          int* featPerLevel, int* umax16) {
  Extractor* e = reinterpret_cast<Extractor*>(h);
  for (int i = 0; i < e->nlevels; ++i) {
    scale[i] = e->mvScaleFactor[i];
    invScale[i] = e->mvInvScaleFactor[i];
    sigma2[i] = e->mvLevelSigma2[i];
    invSigma2[i] = e->mvInvLevelSigma2[i];
    featPerLevel[i] = e->mnFeaturesPerLevel[i];
  }
  for (int i = 0; i < 16; ++i) umax16[i] = e->umax[i];
}

int orc_level_size(orc_extractor* h, int lvl, int* w, int* hh) {
  Extractor* e = reinterpret_cast<Extractor*>(h);
  if (lvl < 0 || lvl >= (int)e->mvImagePyramid.size()) return -1;
  *w = e->mvImagePyramid[lvl].cols;
  *hh = e->mvImagePyramid[lvl].rows;
  return 0;
}
// padded level image, (h+38) x (w+38) contiguous
int orc_level_image(orc_extractor* h, int lvl, uint8_t* out) {
  Extractor* e = reinterpret_cast<Extractor*>(h);
  if (lvl < 0 || lvl >= (int)e->pyrStore.size()) return -1;
  memcpy(out, e->pyrStore[lvl].data(), e->pyrStore[lvl].size());
  return 0;
}
// blurred level image h x w contiguous; returns 0 when the level had no keypoints (no blur computed)
int orc_level_blurred(orc_extractor* h, int lvl, uint8_t* out) {
  Extractor* e = reinterpret_cast<Extractor*>(h);
  if (lvl < 0 || lvl >= (int)e->blurStore.size()) return -1;
  if (e->blurStore[lvl].empty()) return 0;
  memcpy(out, e->blurStore[lvl].data(), e->blurStore[lvl].size());
  return 1;
}
int orc_level_candidates(orc_extractor* h, int lvl, orc_keypoint* out, int cap) {
  Extractor* e = reinterpret_cast<Extractor*>(h);
  int n = (int)e->candidates[lvl].size();
  if (out) memcpy(out, e->candidates[lvl].data(), (size_t)std::min(n, cap) * sizeof(KeyPoint));
  return n;
}
int orc_level_keypoints(orc_extractor* h, int lvl, orc_keypoint* out, int cap) {
  Extractor* e = reinterpret_cast<Extractor*>(h);
  int n = (int)e->levelKeys[lvl].size();
  if (out) memcpy(out, e->levelKeys[lvl].data(), (size_t)std::min(n, cap) * sizeof(KeyPoint));
  return n;
}

// stand-alone primitives for unit tests
void orc_resize_linear(const uint8_t* src, int sw, int sh, int sstep, uint8_t* dst, int dw, int dh, int dstep) {
  resizeLinear8u(Img{const_cast<uint8_t*>(src), sw, sh, sstep}, Img{dst, dw, dh, dstep});
}
void orc_gaussian7(const uint8_t* src, int w, int h, int sstep, uint8_t* dst, int dstep) {
  gaussianBlur7x7s2(Img{const_cast<uint8_t*>(src), w, h, sstep}, Img{dst, w, h, dstep});
}
void orc_border101(uint8_t* buf, int w, int h, int step, int border) {
  Img whole{buf, w + 2 * border, h + 2 * border, step};
  copyMakeBorder101(whole.roi(border, border, w, h), whole, border);
}
int orc_fast(const uint8_t* img, int w, int h, int step, int threshold, int nms, orc_keypoint* out, int cap) {
  std::vector<KeyPoint> k;
  fast9_16(Img{const_cast<uint8_t*>(img), w, h, step}, k, threshold, nms != 0);
  if (out) memcpy(out, k.data(), (size_t)std::min((int)k.size(), cap) * sizeof(KeyPoint));
  return (int)k.size();
}
int orc_distribute(const orc_keypoint* in, int n, int minX, int maxX, int minY, int maxY, int N, orc_keypoint* out,
                   int cap) {
  std::vector<KeyPoint> v((const KeyPoint*)in, (const KeyPoint*)in + n);
  std::vector<KeyPoint> r = DistributeOctTree(v, minX, maxX, minY, maxY, N);
  if (out) memcpy(out, r.data(), (size_t)std::min((int)r.size(), cap) * sizeof(KeyPoint));
  return (int)r.size();
}
// Frame::ComputeStereoMatches on the keypoints/descriptors of two extractors that have just processed the
// left and right image (their mvImagePyramid members are read, Frame.cc:895,974,987)
void orc_stereo_matches(orc_extractor* left, orc_extractor* right, int N, const orc_keypoint* kpsL, const uint8_t* descL,
                        int Nr, const orc_keypoint* kpsR, const uint8_t* descR, float mbf, float mb, float* uRight,
                        float* depth) {
  Extractor* l = reinterpret_cast<Extractor*>(left);
  Extractor* r = reinterpret_cast<Extractor*>(right);
  ComputeStereoMatches(N, (const KeyPoint*)kpsL, descL, Nr, (const KeyPoint*)kpsR, descR, l->mvScaleFactor.data(),
                       l->mvInvScaleFactor.data(), l->mvImagePyramid, r->mvImagePyramid, mbf, mb, uRight, depth);
}
float orc_fast_atan2(float y, float x) { return fastAtan2(y, x); }
float orc_cosf(float x) { return cosf_glibc(x); }
float orc_sinf(float x) { return sinf_glibc(x); }
}
