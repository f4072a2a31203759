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
namespace kb8 {
struct Cam { float p[8]; };  // fx fy cx cy k0 k1 k2 k3
void projectF(const Cam& c, const float* v3D, float* uv);
void projectD(const Cam& c, const double* v3D, double* uv);
void unproject(const Cam& c, float px, float py, float* ray);
void projectJac(const Cam& c, const double* v, double* J);
}  // namespace kb8
}  // namespace orc
