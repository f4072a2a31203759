#pragma once
#include "mock_types.h"
namespace ORB_SLAM3 {
class KeyFrame; class Map;
class MapPoint {   // mock: the members of include/MapPoint.h the reference-typed members touch
 public:
  Eigen::Vector3f GetWorldPos() { return mWorldPos; } Eigen::Vector3f GetNormal() { return mNormalVector; }
  void SetWorldPos(const Eigen::Vector3f& p) { mWorldPos = p; } void UpdateNormalAndDepth() { ++nUpdates; }
  std::map<KeyFrame*, std::tuple<int, int>> GetObservations() { return mObservations; } int Observations() { return nObs; }
  bool isBad() { return mbBad; } cv::Mat GetDescriptor() { return mDescriptor; }
  float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; } float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }
  void EraseObservation(KeyFrame* k) { mObservations.erase(k); } void AddObservation(KeyFrame* k, int idx) { mObservations[k] = std::make_tuple(idx, -1); ++nObs; }
  void Replace(MapPoint* p) { mpReplaced = p; mbBad = true; } bool IsInKeyFrame(KeyFrame* k) { return mObservations.count(k) != 0; }
  std::tuple<int, int> GetIndexInKeyFrame(KeyFrame* k) { return mObservations.count(k) ? mObservations[k] : std::make_tuple(-1, -1); }
  Map* GetMap() { return mpMap; }
  long unsigned int mnLastFrameSeen = 0, mnBALocalForKF = 0, mnId = 0, mnFuseCandidateForKF = 0;
  float mTrackProjX = 0, mTrackProjY = 0, mTrackDepth = 0, mTrackDepthR = 0, mTrackProjXR = 0, mTrackProjYR = 0;
  bool mbTrackInView = false, mbTrackInViewR = false;
  int mnTrackScaleLevel = 0, mnTrackScaleLevelR = 0;
  float mTrackViewCos = 0, mTrackViewCosR = 0;
  static std::mutex mGlobalMutex;
  // (mock state)
  Eigen::Vector3f mWorldPos, mNormalVector; cv::Mat mDescriptor; std::map<KeyFrame*, std::tuple<int, int>> mObservations; int nObs = 0, nUpdates = 0;
  bool mbBad = false; Map* mpMap = nullptr; MapPoint* mpReplaced = nullptr;
  void mock_set_distances(float mx, float mn) { mfMaxDistance = mx; mfMinDistance = mn; }
 protected:
  float mfMinDistance = 0, mfMaxDistance = 0;   // include/MapPoint.h:242-243
  std::mutex mMutexPos;
};
}  // namespace ORB_SLAM3
