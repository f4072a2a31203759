// The call sites of the reference, pasted VERBATIM (expression text of /root/reference/src/Tracking.cc, src/LocalMapping.cc and
// src/LoopClosing.cc; only the surrounding declarations are written here), compiled against include/morb/ORBmatcher.h / Optimizer.h and
// the mock declarations of tests/native/mock_ref (names and types read off the reference headers: this image has no OpenCV / Eigen).
// It compiles <=> the adapters' member templates accept the reference's call expressions with ZERO edits.  -fsyntax-only in the CPU suite
// (tests/test_oracle_cpu.py::test_reference_call_sites_compile_unchanged); every enclosing function is marked `used`, so it is emitted and the
// members it calls are instantiated.
#include <string>

#include "Frame.h"      // tests/native/mock_ref
#include "KeyFrame.h"
#include "Map.h"
#include "MapPoint.h"
#include "ORBmatcher.h"   // include/morb (first on the include path, as in an integrated tree)
#include "Optimizer.h"

using namespace std;
namespace ORB_SLAM3 {
float Frame::fx, Frame::fy, Frame::cx, Frame::cy, Frame::mnMinX, Frame::mnMaxX, Frame::mnMinY, Frame::mnMaxY, Frame::mfGridElementWidthInv,
    Frame::mfGridElementHeightInv;
std::mutex MapPoint::mGlobalMutex;
enum class CameraType { MONOCULAR, STEREO, RGBD, IMU_MONOCULAR, IMU_STEREO, IMU_RGBD };
struct LocalMapperStub { bool mbFarPoints = false; float mThFarPoints = 0; };

struct Tracking {
  Frame mCurrentFrame, mLastFrame;
  KeyFrame* mpReferenceKF = nullptr;
  vector<MapPoint*> mvpLocalMapPoints;
  LocalMapperStub* mpLocalMapper = nullptr;
  CameraType mSensor = CameraType::STEREO;

  __attribute__((used)) bool TrackReferenceKeyFrame() {   // Tracking.cc:2535-2559
    mCurrentFrame.ComputeBoW();
    ORBmatcher matcher(0.7, true);
    vector<MapPoint*> vpMapPointMatches;

    int nmatches =
        matcher.SearchByBoW(mpReferenceKF, mCurrentFrame, vpMapPointMatches);

    if (nmatches < 15) return false;
    mCurrentFrame.mvpMapPoints = vpMapPointMatches;
    mCurrentFrame.SetPose(mLastFrame.GetPose());
    Optimizer::PoseOptimization(&mCurrentFrame);
    return true;
  }
  __attribute__((used)) bool TrackWithMotionModel() {   // Tracking.cc:2660-2710
    ORBmatcher matcher(0.9, true);
    int th;

    if (mSensor == CameraType::STEREO)
      th = 7;
    else
      th = 15;

    int nmatches = matcher.SearchByProjection(
        mCurrentFrame, mLastFrame, th,
        mSensor == CameraType::MONOCULAR || mSensor == CameraType::IMU_MONOCULAR);

    // If few matches, uses a wider window search
    if (nmatches < 20) {
      fill(mCurrentFrame.mvpMapPoints.begin(), mCurrentFrame.mvpMapPoints.end(),
           static_cast<MapPoint*>(NULL));

      nmatches = matcher.SearchByProjection(
          mCurrentFrame, mLastFrame, 2 * th,
          mSensor == CameraType::MONOCULAR || mSensor == CameraType::IMU_MONOCULAR);
    }
    if (nmatches < 20) return false;

    // Optimize frame pose with all matches
    Optimizer::PoseOptimization(&mCurrentFrame);
    return true;
  }
  __attribute__((used)) void TrackLocalMap(bool imuInitialized, bool mapUpdated) {   // Tracking.cc:2755-2790
    int inliers = 0;
    if (!imuInitialized)
      Optimizer::PoseOptimization(&mCurrentFrame);
    else {
      if (!mapUpdated) {
        inliers = Optimizer::PoseInertialOptimizationLastFrame(
            &mCurrentFrame);  // , !mpLastKeyFrame->GetMap()->GetIniertialBA1());
      } else {
        inliers = Optimizer::PoseInertialOptimizationLastKeyFrame(
            &mCurrentFrame);  // , !mpLastKeyFrame->GetMap()->GetIniertialBA1());
      }
    }
    (void)inliers;
  }
  __attribute__((used)) void SearchLocalPoints() {   // Tracking.cc:3160-3180
    ORBmatcher matcher(0.8);
    int th = 1;
    /*int matches = */matcher.SearchByProjection(mCurrentFrame, mvpLocalMapPoints,
                                             th, mpLocalMapper->mbFarPoints,
                                             mpLocalMapper->mThFarPoints);
  }
  __attribute__((used)) int Relocalization(KeyFrame* pKF, vector<vector<MapPoint*>>& vvpMapPointMatches, int i, set<MapPoint*>& sFound) {   // Tracking.cc:3460-3560
    ORBmatcher matcher(0.75, true);
    int nmatches =
        matcher.SearchByBoW(pKF, mCurrentFrame, vvpMapPointMatches[i]);
    ORBmatcher matcher2(0.9, true);
    int nadditional = matcher2.SearchByProjection(mCurrentFrame, pKF, sFound, 10, 100);
    nadditional = matcher2.SearchByProjection(mCurrentFrame, pKF, sFound, 3, 64);
    return nmatches + nadditional;
  }
  __attribute__((used)) int Initialization(Frame& mInitialFrame, vector<cv::Point2f>& mvbPrevMatched, vector<int>& mvIniMatches) {   // Tracking.cc:2150-2155
    ORBmatcher matcher(0.9, true);
    int nmatches = matcher.SearchForInitialization(
        mInitialFrame, mCurrentFrame, mvbPrevMatched, mvIniMatches, 100);
    return nmatches;
  }
};

struct TrackerStub { int GetMatchesInliers() { return 0; } };
struct LocalMapping {
  KeyFrame* mpCurrentKeyFrame = nullptr;
  bool mbAbortBA = false, mbMonocular = false, mbInertial = false;
  TrackerStub* mpTracker = nullptr;

  __attribute__((used)) void Run() {   // LocalMapping.cc:160-185
    int num_FixedKF_BA = 0;
    int num_OptKF_BA = 0;
    int num_MPs_BA = 0;
    int num_edges_BA = 0;
    if (mbInertial && mpCurrentKeyFrame->GetMap()->IsInertial()) {
            bool bLarge =
                ((mpTracker->GetMatchesInliers() > 75) && mbMonocular) ||
                ((mpTracker->GetMatchesInliers() > 100) && !mbMonocular);
            Optimizer::LocalInertialBA(
                mpCurrentKeyFrame, &mbAbortBA, mpCurrentKeyFrame->GetMap(),
                num_FixedKF_BA, num_OptKF_BA, num_MPs_BA, num_edges_BA, bLarge,
                !mpCurrentKeyFrame->GetMap()->GetIniertialBA2());
    } else {
            Optimizer::LocalBundleAdjustment(
                mpCurrentKeyFrame, &mbAbortBA, mpCurrentKeyFrame->GetMap(),
                num_FixedKF_BA, num_OptKF_BA, num_MPs_BA, num_edges_BA);
    }
  }
  __attribute__((used)) void CreateNewMapPoints(const vector<KeyFrame*>& vpNeighKFs) {   // LocalMapping.cc:440-473
    ORBmatcher matcher(0.6, false);
    for (size_t i = 0; i < vpNeighKFs.size(); i++) {
      KeyFrame* pKF2 = vpNeighKFs[i];
      // Search matches that fullfil epipolar constraint
      vector<pair<size_t, size_t>> vMatchedIndices;
      bool bCoarse = mbInertial;

      matcher.SearchForTriangulation(mpCurrentKeyFrame, pKF2, vMatchedIndices,
                                     false, bCoarse);
    }
  }
  __attribute__((used)) void SearchInNeighbors(vector<KeyFrame*>& vpTargetKFs) {   // LocalMapping.cc:765-802
    ORBmatcher matcher;
    vector<MapPoint*> vpMapPointMatches = mpCurrentKeyFrame->GetMapPointMatches();
    for (vector<KeyFrame*>::iterator vit = vpTargetKFs.begin(),
                                     vend = vpTargetKFs.end();
         vit != vend; vit++) {
      KeyFrame* pKFi = *vit;

      matcher.Fuse(pKFi, vpMapPointMatches);
      if (pKFi->NLeft != -1) matcher.Fuse(pKFi, vpMapPointMatches, true);
    }
    vector<MapPoint*> vpFuseCandidates;
    matcher.Fuse(mpCurrentKeyFrame, vpFuseCandidates);
    if (mpCurrentKeyFrame->NLeft != -1)
      matcher.Fuse(mpCurrentKeyFrame, vpFuseCandidates, true);
  }
};

struct LoopClosing {   // LoopClosing.cc: the Sim3 searches and the loop fusion
  KeyFrame* mpCurrentKF = nullptr;
  __attribute__((used)) int calls(KeyFrame* pKFi, KeyFrame* pMostBoWMatchesKF, Sophus::Sim3f& mScw, vector<MapPoint*>& vpMapPoints, vector<KeyFrame*>& vpKeyFrames,
            vector<MapPoint*>& vpMatchedMPs, vector<KeyFrame*>& vpMatchedKFs, vector<MapPoint*>& vpMatchedPoints, Sophus::Sim3f& gScm) {
    ORBmatcher matcherBoW(0.9, true);
    ORBmatcher matcher(0.75, true);
    int num = matcherBoW.SearchByBoW(mpCurrentKF, pKFi, vpMatchedPoints);                                  // LoopClosing.cc:633
    int numProjMatches = matcher.SearchByProjection(mpCurrentKF, mScw, vpMapPoints, vpKeyFrames, vpMatchedMPs, vpMatchedKFs, 8, 1.5);   // :728
    int numProjOptMatches = matcher.SearchByProjection(mpCurrentKF, mScw, vpMapPoints, vpMatchedMPs, 5, 1.0);   // :766
    num += matcher.SearchBySim3(mpCurrentKF, pMostBoWMatchesKF, vpMatchedMPs, gScm, 7.5);                    // (ORBmatcher.h:95)
    vector<MapPoint*> vpReplacePoints(vpMapPoints.size(), static_cast<MapPoint*>(NULL));
    int numFused = matcher.Fuse(pKFi, mScw, vpMapPoints, 4, vpReplacePoints);                                 // :1993
    return num + numProjMatches + numProjOptMatches + numFused;
  }
};
int DescriptorDistanceCall(const cv::Mat& a, const cv::Mat& b) { return ORBmatcher::DescriptorDistance(a, b); }   // ORBmatcher.h:41
}  // namespace ORB_SLAM3
