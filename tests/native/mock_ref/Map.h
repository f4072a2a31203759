#pragma once
#include "mock_types.h"
namespace ORB_SLAM3 {
class Map { public: long unsigned int GetInitKFid(); bool IsInertial(); void IncreaseChangeIndex(); std::mutex mMutexMapUpdate; };
}  // namespace ORB_SLAM3
