#pragma once
#include "mock_types.h"
namespace ORB_SLAM3 {
class MapPoint;
class Frame {   // mock: the members of include/Frame.h the glue touches
 public:
  Sophus::SE3f GetPose() const; bool HasPose() const; void SetPose(const Sophus::SE3f&);
  int N; std::vector<cv::KeyPoint> mvKeysUn; std::vector<float> mvuRight; cv::Mat mDescriptors; DBoW2::FeatureVector mFeatVec;
  std::vector<MapPoint*> mvpMapPoints; std::vector<bool> mvbOutlier; std::vector<float> mvInvLevelSigma2, mvScaleFactors, mvLevelSigma2;
  static float fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
  float mbf, mb, mfLogScaleFactor; int mnScaleLevels; long unsigned int mnId;
};
}  // namespace ORB_SLAM3
