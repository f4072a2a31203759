// Parses and type-checks include/morb/reference_glue.h against MOCK declarations of the reference classes (tests/native/mock_ref:
// names only, not the reference and not OpenCV / Eigen / Sophus).  -fsyntax-only; instantiates every inline function by taking its address.
#include "morb/reference_glue.h"
namespace g = ORB_SLAM3::morb_glue;
void* use[] = {(void*)(int (*)(ORB_SLAM3::ORBmatcher&, ORB_SLAM3::Frame&, const std::vector<ORB_SLAM3::MapPoint*>&, float, bool, float))&g::SearchByProjection,
               (void*)(int (*)(ORB_SLAM3::ORBmatcher&, ORB_SLAM3::Frame&, ORB_SLAM3::Frame&, float, bool))&g::SearchByProjection,
               (void*)&g::SearchByBoW, (void*)&g::SearchForTriangulation, (void*)&g::PoseOptimization, (void*)&g::LocalBundleAdjustment};
