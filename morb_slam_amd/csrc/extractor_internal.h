// Internal layout of the extractor handle and its per-level geometry, shared by the translation units of
// libmorb_hip.so (the stereo matcher reads the pyramids the extractor owns, like Frame::ComputeStereoMatches
// reads mpORBextractorLeft->mvImagePyramid, Frame.cc:895).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "morb_hip.h"

namespace morb {
constexpr int kMaxLevels = 16;
constexpr int EDGE = 19;        // EDGE_THRESHOLD  ORBextractor.cc:73
constexpr int HALF_PATCH = 15;  // HALF_PATCH_SIZE :72
constexpr int PATCH = 31;       // PATCH_SIZE :71
constexpr int MINB = 16;        // minBorderX/Y = EDGE_THRESHOLD - 3 (:746-747)
constexpr int kLdsKeys = 3072;  // distribute: the LDS key arrays hold at least this many candidates per level (more for larger images, see configure())

struct LevelGeom {
  int w, h;
  int pstride, bstride;
  unsigned long long pyrOff, pyrImg;    // bytes: slab offset, per-image size
  unsigned long long blurOff, blurImg;
  int nCols, nRows, wCell, hCell, cellBase;
  int maxBorderX, maxBorderY;
  int quota, nIni;
  int nodeCap, listCap;
  int selBase, selCap;                  // per-image offsets into sel[]
  unsigned long long qtOff, qtImg;      // element offsets into the global key scratch
  int blurTileBase, blurTilesX, blurTilesY;
  int xtabOff, ytabOff;                 // offsets (entries) into the resize tables
  float scale;                          // mvScaleFactor[l]
  float kpSize;                         // (float)(int)(PATCH_SIZE * mvScaleFactor[l])  (:831)
  // k_distribute packs several levels of an image into one workgroup, a wave per level: the workgroup (group) and wave of this
  // level, the byte offset of its LDS region inside the workgroup's allocation, and the candidate keys its LDS arrays hold
  int distGroup, distWave, distLdsOff, distKeyCap;
  int distTeam;                         // all waves of the workgroup work this level as a team (quadtree.h: Team)
};

// What k_fastw needs of every level, passed by value: the kernarg segment is read with scalar loads, so looking a
// cell's level up costs no dependent global-memory round trip.
struct FastGeom {
  int cellBase[kMaxLevels], nCols[kMaxLevels], wCell[kMaxLevels], hCell[kMaxLevels];
  int pstride[kMaxLevels], maxBorderX[kMaxLevels], maxBorderY[kMaxLevels];
  unsigned long long pyrOff[kMaxLevels], pyrImg[kMaxLevels];
  unsigned wCellMagic[kMaxLevels];      // ceil(2^16 / wCell): x / wCell == (x * magic) >> 16 for every x < 128
};

// k_fastw works cell by cell.  One descriptor per FAST cell of an image (all levels), built on the host, read with one scalar load
// (the "segment" of round 2's kernel, which ran up to three cells per workgroup; the cell count field stays at 1).
struct FastSeg {
  unsigned winOff;   // byte offset of the window's top-left pixel inside the level's (padded) image
  int cell0;         // flat index of the segment's first cell within the image (cellBase + ci * nCols + c0)
  int geo;           // level | cells << 8 | window width << 16 | window height << 24 (width 0: nothing to evaluate)
  int key0;          // added to window coordinates to form candidate keys: c0 * wCell | (ci * hCell) << 16
};

// What k_describe needs of every level, by value (see FastGeom).
struct DescGeom {
  unsigned long long pyrOff[kMaxLevels], pyrImg[kMaxLevels], blurOff[kMaxLevels], blurImg[kMaxLevels];
  int pstride[kMaxLevels], bstride[kMaxLevels];
  float scale[kMaxLevels], kpSize[kMaxLevels];
};

struct alignas(8) ResizeTab {  // one entry per padded destination column / row (8-byte aligned: scalar loads need dword alignment)
  short s0, s1, c0, c1;  // source index (clipped), source index + 1 (clipped), fixed-point weights (2048 = 1)
};

// k_resize / k_level01 work on linearly numbered items (extractor.hip): per destination dword of a level (PyrCol), per destination
// row (PyrRow), per pad dword of level 0 (PyrEdge)
struct alignas(16) PyrCol {   // 48 bytes: three 16-byte loads
  int base;            // the chunk's 8-byte source window starts at byte base + sh of the source's padded row; base is a multiple of 4
  uint32_t sel[4];     // v_perm_b32 selector of pixel k: left source byte -> byte 0, right source byte -> byte 2, bytes 1 and 3 zero
  uint32_t coef[4];    // c0 | c1 << 16 (2048 = 1)
  int sh, pad_[2];
};
struct alignas(16) PyrRow { int s0, s1; uint32_t c0s, c1s; };   // source rows (clipped), vertical weights << 12
struct alignas(16) PyrEdge { int dword, base; uint32_t sel; int pad_; };   // destination dword, source byte offset of its 4-byte window, byte selector
struct PyrLevel { int nC, nG, nItems, colOff, rowOff, ldsRows; uint32_t magicC; };
struct PyrLevel0 { int nInt, j0, nEdge, nG, nIntItems, nItems; uint32_t magicInt, magicEdge; };

}  // namespace morb

struct morb_extractor {
  morb::FastGeom fastGeom = {};
  morb::DescGeom descGeom = {};
  int2* d_kref = nullptr;   // per selected keypoint: output slot | level << 24 (or -1), packed key (k_layout -> k_describe)
  int nfeatures = 0, nlevels = 0, iniTh = 0, minTh = 0, device = 0;
  float scaleFactor = 1.2f;
  std::vector<float> scale, invScale, sigma2, invSigma2;
  std::vector<int> quota;
  int umax[16];

  int W = 0, H = 0, nimgCap = 0, nimgLast = 0;
  morb::LevelGeom geom[morb::kMaxLevels];
  int totalCells = 0, cellCap = 0, maxCells = 0, maxNodeCap = 0, maxListCap = 0;
  int fastSegs[2] = {0, 0}, fastRows[2] = {0, 0};   // k_fastw launch groups (cells, LDS window rows)
  int fastP = 128;                                  // k_fastw: LDS pitch of the segment windows
  int selPerImg = 0, blurTiles = 0, outCap = 0;
  size_t pyrBytes = 0, blurBytes = 0, qtElems = 0, distSmem = 0, fastSmem[2] = {0, 0};
  int distKeyCap = 0;                        // candidate keys of level 0 that fit the quadtree's LDS arrays (smaller levels: scaled by area)
  int distGroups = 0, distWaves = 1;         // k_distribute grid: workgroups per image, waves per workgroup
  // the same for calls with few images (latency): the big levels are worked by a team of QT_MAX_WAVES waves each
  int distGroupsTeam = 0; size_t distSmemTeam = 0;
  morb::LevelGeom* d_geomTeam = nullptr;

  hipStream_t stream = nullptr;
  hipStream_t sideStream = nullptr;          // the blur runs here, underneath the quadtree (fork after FAST, join before describe)
  hipEvent_t evFork = nullptr, evJoin = nullptr;
  hipEvent_t evPyr = nullptr;      // recorded on the launch stream behind the last pyramid launch (morb_extractor_event_after_pyramid)
  bool wantPyrEvent = false;       // ... once a caller has asked for it
  morb::LevelGeom* d_geom = nullptr;
  morb::ResizeTab* d_tabs = nullptr;
  morb::PyrCol* d_pcol = nullptr; morb::PyrRow* d_prow = nullptr; morb::PyrEdge* d_pedge = nullptr;
  morb::PyrLevel pyrLv[morb::kMaxLevels] = {}; morb::PyrLevel0 pyrL0 = {};
  bool pyrPacked = true;   // every level's four-pixel chunks fit k_resize's 8-byte source windows (scale factors up to ~1.75)
  morb::FastSeg* d_segTab = nullptr;
  uint8_t *d_pyr = nullptr, *d_blur = nullptr;
  uint32_t *d_cand = nullptr, *d_qt = nullptr, *d_sel = nullptr;
  int *d_candCnt = nullptr, *d_selCnt = nullptr,  *d_lap = nullptr;
  int *h_status = nullptr, *d_status = nullptr;   // pinned, device-mapped flags the kernels can raise (bit 0: a level with > 65535 FAST candidates)
  // staging for the single-image host API
  uint8_t* d_img = nullptr; size_t imgBytes = 0;
  void* d_out1 = nullptr;   // morb_extract's result block: the four pointers below point into it
  morb_keypoint* d_kps1 = nullptr; uint8_t* d_desc1 = nullptr; int *d_cnt1 = nullptr, *d_mono1 = nullptr;
  uint8_t* h_io1 = nullptr; size_t ioBytes1 = 0;   // pinned host staging of morb_extract: the image on the way in, [cnt, mono | keypoints | descriptors] on the way out
  std::vector<int> lapLast;  // host mirror of d_lap
  // profiling: a ring of event sets, one per morb_extract_batch call, read back (and averaged) on demand so the
  // timed region never synchronises with the host
  bool profiling = false;
  static constexpr int kProfRing = 64;
  std::vector<hipEvent_t> ev;  // [kProfRing][8]
  int profCalls = 0;
  float stageMs[7] = {0};
};

