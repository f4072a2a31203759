#pragma once
#include "mock_types.h"
namespace ORB_SLAM3 {
class Map {
 public:
  long unsigned int GetInitKFid() { return initKF; } bool IsInertial() { return inertial; } void IncreaseChangeIndex() { ++changes; }
  long unsigned int KeyFramesInMap() { return nKFs; } bool GetIniertialBA2() { return ba2; }
  std::mutex mMutexMapUpdate;
  long unsigned int initKF = 0, nKFs = 0; bool inertial = false, ba2 = false; int changes = 0;
};
}  // namespace ORB_SLAM3
