#pragma once
#include "mock_types.h"
namespace ORB_SLAM3 {
class MapPoint; class Map;
struct GeometricCamera { Eigen::Vector2f project(const Eigen::Vector3f&); };
class KeyFrame {   // mock: the members of include/KeyFrame.h the glue touches
 public:
  Sophus::SE3f GetPose(); Sophus::SE3f GetPoseInverse(); Eigen::Vector3f GetCameraCenter(); void SetPose(const Sophus::SE3f&);
  std::vector<MapPoint*> GetMapPointMatches(); std::vector<KeyFrame*> GetVectorCovisibleKeyFrames(); bool isBad(); Map* GetMap();
  void EraseMapPointMatch(MapPoint*);
  const int N = 0; const std::vector<cv::KeyPoint> mvKeysUn; const std::vector<float> mvuRight; const cv::Mat mDescriptors; DBoW2::FeatureVector mFeatVec;
  const float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0, mfLogScaleFactor = 0;
  const int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0, mnScaleLevels = 0;
  const std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  long unsigned int mnId, mnBALocalForKF, mnBAFixedForKF;
  GeometricCamera* mpCamera;
};
}  // namespace ORB_SLAM3
