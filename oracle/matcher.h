// ORACLE — TEST INFRASTRUCTURE ONLY.  Internal prototypes shared between oracle translation units.
#pragma once
#include <vector>

#include "cvprims.h"
namespace orc {
int DescriptorDistance(const uint8_t* a, const uint8_t* b);
void ComputeStereoMatches(int N, const KeyPoint* mvKeys, const uint8_t* mDescriptors, int Nr, const KeyPoint* mvKeysRight,
                          const uint8_t* mDescriptorsRight, const float* mvScaleFactors, const float* mvInvScaleFactors,
                          const std::vector<Img>& pyrL, const std::vector<Img>& pyrR, float mbf, float mb,
                          float* mvuRight, float* mvDepth);
}  // namespace orc
