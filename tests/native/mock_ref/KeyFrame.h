#pragma once
#include "MapPoint.h"
#include "Map.h"
namespace ORB_SLAM3 {
class KeyFrame {   // mock: the members of include/KeyFrame.h the reference-typed members touch
 public:
  Sophus::SE3f GetPose() { return mTcw; } Sophus::SE3f GetPoseInverse() { return mTcw.inverse(); }
  Eigen::Vector3f GetCameraCenter() { return Eigen::Vector3f(mTcw.Ow[0], mTcw.Ow[1], mTcw.Ow[2]); } void SetPose(const Sophus::SE3f& T) { mTcw = T; }
  std::vector<MapPoint*> GetMapPointMatches() { return mvpMapPoints; } std::set<MapPoint*> GetMapPoints() { std::set<MapPoint*> s; for (auto* p : mvpMapPoints) if (p && !p->isBad()) s.insert(p); return s; }
  MapPoint* GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; } void AddMapPoint(MapPoint* p, const size_t& idx) { mvpMapPoints[idx] = p; }
  void EraseMapPointMatch(MapPoint* p) { for (auto& q : mvpMapPoints) if (q == p) q = nullptr; }
  std::vector<KeyFrame*> GetVectorCovisibleKeyFrames() { return mvpCov; } bool isBad() { return mbBad; } Map* GetMap() { return mpMap; }
  Sophus::SE3f GetRelativePoseTrl() { return mTrl; } Sophus::SE3f GetRightPose() { return mTrw; } Sophus::SE3f GetRightPoseInverse() { return mTwr; }
  Eigen::Vector3f GetRightCameraCenter() { return Eigen::Vector3f(mTrw.Ow[0], mTrw.Ow[1], mTrw.Ow[2]); }
  Eigen::Matrix3f GetImuRotation() { return mRwb; } Eigen::Vector3f GetImuPosition() { return mOwb; } Eigen::Vector3f GetVelocity() { return mVw; }
  IMU::Bias GetImuBias() { return mImuBias; } void SetVelocity(const Eigen::Vector3f& v) { mVw = v; } void SetNewBias(const IMU::Bias& b) { mImuBias = b; }
  int N = 0, NLeft = -1, NRight = -1;
  std::vector<cv::KeyPoint> mvKeys, mvKeysRight, mvKeysUn; std::vector<float> mvuRight; cv::Mat mDescriptors; DBoW2::FeatureVector mFeatVec;
  float fx = 0, fy = 0, cx = 0, cy = 0, mbf = 0, mb = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0, mfLogScaleFactor = 0, mfScaleFactor = 1.2f;
  int mnMinX = 0, mnMinY = 0, mnMaxX = 0, mnMaxY = 0, mnScaleLevels = 0;
  std::vector<float> mvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
  long unsigned int mnId = 0, mnBALocalForKF = 0, mnBAFixedForKF = 0;
  GeometricCamera *mpCamera = nullptr, *mpCamera2 = nullptr;
  bool bImu = false; KeyFrame* mPrevKF = nullptr; IMU::Preintegrated* mpImuPreintegrated = nullptr; IMU::Calib mImuCalib;
  // (mock state)
  Sophus::SE3f mTcw, mTrl, mTrw, mTwr; std::vector<MapPoint*> mvpMapPoints; std::vector<KeyFrame*> mvpCov; bool mbBad = false; Map* mpMap = nullptr;
  Eigen::Matrix3f mRwb; Eigen::Vector3f mOwb, mVw; IMU::Bias mImuBias;
};
}  // namespace ORB_SLAM3
