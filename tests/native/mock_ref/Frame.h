#pragma once
#include "KeyFrame.h"
namespace ORB_SLAM3 {
class Frame {   // mock: the members of include/Frame.h the reference-typed members touch
 public:
  Sophus::SE3f GetPose() const { return mTcw; } bool HasPose() const { return mbHasPose; } void SetPose(const Sophus::SE3f& T) { mTcw = T; mbHasPose = true; }
  Sophus::SE3f GetRelativePoseTrl() { return mTrl; } void ComputeBoW() {}
  Eigen::Matrix3f GetImuRotation() { return mRwb; } Eigen::Vector3f GetImuPosition() const { return mOwb; } Eigen::Vector3f GetVelocity() const { return mVw; }
  void SetImuPoseVelocity(const Eigen::Matrix3f& R, const Eigen::Vector3f& t, const Eigen::Vector3f& v) { mRwb = R; mOwb = t; mVw = v; }
  int N = 0, Nleft = -1, Nright = -1;
  std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn; std::vector<float> mvuRight; cv::Mat mDescriptors; DBoW2::FeatureVector mFeatVec;
  std::vector<MapPoint*> mvpMapPoints; std::vector<bool> mvbOutlier; std::vector<float> mvInvLevelSigma2, mvScaleFactors, mvLevelSigma2;
  std::vector<int> mvLeftToRightMatch, mvRightToLeftMatch;
  static float fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
  float mbf = 0, mb = 0, mfLogScaleFactor = 0; int mnScaleLevels = 0; long unsigned int mnId = 0;
  GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
  IMU::Bias mImuBias; IMU::Calib mImuCalib; IMU::Preintegrated *mpImuPreintegrated = nullptr, *mpImuPreintegratedFrame = nullptr;
  KeyFrame* mpLastKeyFrame = nullptr; Frame* mpPrevFrame = nullptr; ConstraintPoseImu* mpcpi = nullptr;
  // (mock state)
  Sophus::SE3f mTcw, mTrl; bool mbHasPose = false; Eigen::Matrix3f mRwb; Eigen::Vector3f mOwb, mVw;
};
}  // namespace ORB_SLAM3
