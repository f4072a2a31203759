// Drop-in adapter: ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:44-105) over libmorb_hip.so.
// Same constructor, same operator(), same accessors, and a public mvImagePyramid member that Frame::ComputeStereoMatches can keep
// indexing (Frame.cc:895, :974, :987, :992): the levels are downloaded from the device on the first access after an extraction, not
// by operator() itself — a front end that uses this library's stereo matcher (which reads the pyramids where they are, in HBM) never
// pays for eight padded host copies per call, an unchanged Frame.cc still finds them.  With OpenCV headers present the cv:: types are
// used directly (that branch has never been compiled here: this build container has no OpenCV); without them the POD stand-ins below
// keep the header compilable, and tests/native/adapters_check.cc drives the same lazy path through them.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../morb_hip.h"

#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#define MORB_HAVE_OPENCV 1
#else
#define MORB_HAVE_OPENCV 0
#endif

namespace ORB_SLAM3 {

#if !MORB_HAVE_OPENCV
namespace podcv {  // minimal stand-ins with the layouts the ABI uses
struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };
struct Mat8u { std::vector<uint8_t> data; int rows = 0, cols = 0, step = 0; bool empty() const { return rows == 0 || cols == 0; } };
}  // namespace podcv
#endif

// mvImagePyramid: the vector<Mat> of the reference, filled on first use.  operator[] / size() / iteration / conversion to the vector
// trigger one download of all levels per extraction (the reference's readers run after both extractor threads have joined, Frame.cc:197-198).
template <typename MatT>
class LazyPyramid {
 public:
  using Fetch = void (*)(void* owner, std::vector<MatT>& levels);
  LazyPyramid(void* owner, Fetch fetch) : owner_(owner), fetch_(fetch) {}
  MatT& operator[](size_t l) { ensure(); return levels_[l]; }
  size_t size() { ensure(); return levels_.size(); }
  bool empty() { return size() == 0; }
  typename std::vector<MatT>::iterator begin() { ensure(); return levels_.begin(); }
  typename std::vector<MatT>::iterator end() { ensure(); return levels_.end(); }
  operator std::vector<MatT>&() { ensure(); return levels_; }
  void invalidate() { fresh_ = false; }          // called by operator(): the device pyramid changed
  bool downloaded() const { return fresh_; }
 private:
  void ensure() { if (!fresh_) { fetch_(owner_, levels_); fresh_ = true; } }
  void* owner_; Fetch fetch_;
  std::vector<MatT> levels_;
  bool fresh_ = false;
};

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0)
      : mvImagePyramid(this, &ORBextractor::fetchPyramid), nlevels_(nlevels), scaleFactor_(scaleFactor) {
    if (morb_extractor_create(&h_, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device) != MORB_OK)
      throw std::runtime_error(std::string("morb_extractor_create: ") + morb_last_error());
    mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    morb_extractor_tables(h_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(), nullptr);
    cap_ = morb_extractor_max_keypoints(h_);
  }
  ~ORBextractor() { morb_extractor_destroy(h_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

#if MORB_HAVE_OPENCV
  // int operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>&, cv::OutputArray, std::vector<int>&)
  int operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& _keypoints,
                 cv::OutputArray _descriptors, std::vector<int>& vLappingArea) {
    if (_image.empty()) return -1;
    cv::Mat image = _image.getMat();
    CV_Assert(image.type() == CV_8UC1);
    static_assert(sizeof(cv::KeyPoint) == sizeof(morb_keypoint), "cv::KeyPoint layout");
    std::vector<morb_keypoint> k(cap_);
    std::vector<uint8_t> d((size_t)cap_ * 32);
    int n = 0;
    const int mono = morb_extract(h_, image.data, image.cols, image.rows, (int)image.step, vLappingArea[0], vLappingArea[1],
                                  k.data(), d.data(), cap_, &n);
    if (mono < 0) { if (mono == MORB_ERR_EMPTY) return -1; throw std::runtime_error(morb_last_error()); }
    mvImagePyramid.invalidate();
    _keypoints.resize(n);
    if (n) std::memcpy(static_cast<void*>(_keypoints.data()), k.data(), sizeof(morb_keypoint) * n);
    if (n == 0) _descriptors.release();
    else { _descriptors.create(n, 32, CV_8U); std::memcpy(_descriptors.getMat().data, d.data(), (size_t)n * 32); }
    return mono;
  }
  // reference: public member read by Frame::ComputeStereoMatches (Frame.cc:895); levels are ROIs into padded host copies, like the reference's
  LazyPyramid<cv::Mat> mvImagePyramid;
#else
  LazyPyramid<podcv::Mat8u> mvImagePyramid;   // (interior of each level, step = width)
  int operator()(const podcv::Mat8u& image, std::vector<podcv::KeyPoint>& keypoints, std::vector<uint8_t>& descriptors,
                 const std::vector<int>& vLappingArea) {
    if (image.empty()) return -1;
    static_assert(sizeof(podcv::KeyPoint) == sizeof(morb_keypoint), "KeyPoint layout");
    keypoints.resize(cap_);
    descriptors.resize((size_t)cap_ * 32);
    int n = 0;
    const int mono = morb_extract(h_, image.data.data(), image.cols, image.rows, image.step, vLappingArea[0], vLappingArea[1],
                                  reinterpret_cast<morb_keypoint*>(keypoints.data()), descriptors.data(), cap_, &n);
    if (mono < 0) { if (mono == MORB_ERR_EMPTY) return -1; throw std::runtime_error(morb_last_error()); }
    mvImagePyramid.invalidate();
    keypoints.resize(n);
    descriptors.resize((size_t)n * 32);
    return mono;
  }
#endif

  int inline GetLevels() { return nlevels_; }
  float inline GetScaleFactor() { return scaleFactor_; }
  std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
  std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
  std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
  std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

  morb_extractor* handle() { return h_; }  // for the batched / device-resident entry points

 private:
#if MORB_HAVE_OPENCV
  static void fetchPyramid(void* self, std::vector<cv::Mat>& levels) {  // host copies of the levels with their 19-px pad, exposed as ROIs like the reference
    ORBextractor* e = static_cast<ORBextractor*>(self);
    levels.resize(e->nlevels_);
    for (int l = 0; l < e->nlevels_; ++l) {
      int w, h, s; const uint8_t* p;
      morb_extractor_pyramid_level(e->h_, 0, l, &p, &w, &h, &s);
      cv::Mat padded(h + 38, w + 38, CV_8UC1);
      morb_extractor_pyramid_level_host(e->h_, 0, l, padded.data);
      levels[l] = padded(cv::Rect(19, 19, w, h));
    }
  }
#else
  static void fetchPyramid(void* self, std::vector<podcv::Mat8u>& levels) {
    ORBextractor* e = static_cast<ORBextractor*>(self);
    levels.resize(e->nlevels_);
    for (int l = 0; l < e->nlevels_; ++l) {
      int w, h, s; const uint8_t* p;
      morb_extractor_pyramid_level(e->h_, 0, l, &p, &w, &h, &s);
      std::vector<uint8_t> padded((size_t)(h + 38) * (w + 38));
      morb_extractor_pyramid_level_host(e->h_, 0, l, padded.data());
      podcv::Mat8u& m = levels[l];
      m.rows = h; m.cols = w; m.step = w; m.data.resize((size_t)w * h);
      for (int y = 0; y < h; ++y) std::memcpy(&m.data[(size_t)y * w], &padded[(size_t)(y + 19) * (w + 38) + 19], w);
    }
  }
#endif
  morb_extractor* h_ = nullptr;
  int nlevels_, cap_ = 0;
  float scaleFactor_;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

}  // namespace ORB_SLAM3
