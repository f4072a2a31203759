#pragma once
#include "mock_types.h"
namespace ORB_SLAM3 {
class KeyFrame; class Map;
class MapPoint {   // mock: the members of include/MapPoint.h the glue touches
 public:
  Eigen::Vector3f GetWorldPos(); Eigen::Vector3f GetNormal(); void SetWorldPos(const Eigen::Vector3f&); void UpdateNormalAndDepth();
  std::map<KeyFrame*, std::tuple<int, int>> GetObservations(); int Observations(); bool isBad(); cv::Mat GetDescriptor();
  float GetMinDistanceInvariance(); float GetMaxDistanceInvariance(); void EraseObservation(KeyFrame*); Map* GetMap();
  long unsigned int mnLastFrameSeen, mnBALocalForKF, mnId;
  static std::mutex mGlobalMutex;
};
}  // namespace ORB_SLAM3
