/* ORACLE — TEST INFRASTRUCTURE ONLY.  C API of the CPU restatement (ctypes-loadable liboracle.so).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library. */
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, size, angle, response; int octave, class_id; } orc_keypoint; /* cv::KeyPoint, 28 B */
typedef struct orc_extractor orc_extractor;
/* the Frame fields the projection matchers read (Frame.h), flattened */
typedef struct {
  int N;
  const orc_keypoint* kpsUn;   /* mvKeysUn */
  const uint8_t* desc;         /* mDescriptors */
  const float* uRight;         /* mvuRight or NULL */
  float minX, minY, maxX, maxY, gridInvW, gridInvH;
  const float* scaleFactors;
  int nlevels;
  float fx, fy, cx, cy, mbf, mb, logScaleFactor;
} orc_frame;

orc_extractor* orc_extractor_create(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
void orc_extractor_destroy(orc_extractor*);
/* ORBextractor::operator() — returns monoIndex, -1 for an empty image, -2 if cap is too small */
int orc_extract(orc_extractor*, const uint8_t* img, int w, int h, int stride, int lap0, int lap1,
                orc_keypoint* kps, uint8_t* desc, int cap, int* n_out);
void orc_extractor_tables(orc_extractor*, float* scale, float* invScale, float* sigma2, float* invSigma2,
                          int* featPerLevel, int* umax16);
int orc_level_size(orc_extractor*, int lvl, int* w, int* h);
int orc_level_image(orc_extractor*, int lvl, uint8_t* out_padded);
int orc_level_blurred(orc_extractor*, int lvl, uint8_t* out);
int orc_level_candidates(orc_extractor*, int lvl, orc_keypoint* out, int cap);
int orc_level_keypoints(orc_extractor*, int lvl, orc_keypoint* out, int cap);

void orc_resize_linear(const uint8_t* src, int sw, int sh, int sstep, uint8_t* dst, int dw, int dh, int dstep);
void orc_gaussian7(const uint8_t* src, int w, int h, int sstep, uint8_t* dst, int dstep);
void orc_border101(uint8_t* buf, int w, int h, int step, int border);
int orc_fast(const uint8_t* img, int w, int h, int step, int threshold, int nms, orc_keypoint* out, int cap);
int orc_distribute(const orc_keypoint* in, int n, int minX, int maxX, int minY, int maxY, int N,
                   orc_keypoint* out, int cap);
void orc_stereo_matches(orc_extractor* left, orc_extractor* right, int N, const orc_keypoint* kpsL, const uint8_t* descL,
                        int Nr, const orc_keypoint* kpsR, const uint8_t* descR, float mbf, float mb, float* uRight,
                        float* depth);
int orc_descriptor_distance(const uint8_t* a, const uint8_t* b);
void orc_three_maxima(const int* counts, int L, int* ind3);
void orc_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int* idx, int* dist);
int orc_search_by_bow(int nKF, const uint8_t* descKF, const float* angleKF, const uint8_t* kfHasMP, const int* nodeKF,
                      int nF, const uint8_t* descF, const float* angleF, const int* nodeF, float nnratio, int checkOri,
                      int* matchF);
/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:367-435) for nMP map points: descriptors of point m = rows [start[m], start[m+1]) */
void orc_distinctive_descriptors(int nMP, const int* start, const uint8_t* desc, int* bestIdx);
/* SearchByBoW(pKF1, pKF2, vpMatches12) (ORBmatcher.cc:702-819) */
int orc_search_by_bow_kfkf(int n1, int nValid1, const uint8_t* desc1, const float* angle1, const uint8_t* hasMP1, const int* node1,
                           int n2, int nValid2, const uint8_t* desc2, const float* angle2, const uint8_t* hasMP2, const int* node2,
                           float nnratio, int checkOri, int* matches12);
/* SearchByBoW(pKF, F, ...) with F.Nleft = FNleft != -1 (ORBmatcher.cc:262-299, :333-365) */
int orc_search_by_bow_fisheye(int nKF, const uint8_t* descKF, const float* angleKF, const uint8_t* kfHasMP, const int* nodeKF,
                              int nF, int FNleft, const uint8_t* descF, const float* angleF, const int* nodeF, float nnratio,
                              int checkOri, int* matchF);
void orc_bow_transform_tree(const uint8_t* feat, int n, const uint8_t* nodeDesc, const int* firstChild, const int* childCount, int L,
                            int levelsup, int* wordId, int* nodeId);
void orc_bow_transform(const uint8_t* feat, int n, const uint8_t* nodeDesc, const int* firstChild, int k, int L,
                       int levelsup, int* wordId, int* nodeId);
int orc_pose_optimization(int n, const uint8_t* hasMP, const float* obs, const float* invSigma2, const float* Xw,
                          float fx, float fy, float cx, float cy, float bf, float* pose, uint8_t* outlier, int* stats);
int orc_local_ba(int nKF, float* kfPose, const uint8_t* kfFixed, int nMP, float* mpPos, int nE, const int* eKF,
                 const int* eMP, const float* eObs, const float* eInvSigma2, float fx, float fy, float cx, float cy,
                 float bf, int lambdaInit100, const int* stopFlag, uint8_t* eraseFlag, int* stats);
void orc_is_in_frustum(const orc_frame* F, const float* Rcw, const float* tcw, const float* Ow, int nMP, const float* Pw,
                       const float* normal, const float* maxDist, const float* minDist, float viewingCosLimit,
                       uint8_t* inView, float* projX, float* projY, float* projXR, float* depth, int* level,
                       float* viewCos);
int orc_search_by_projection_mps(const orc_frame* F, const uint8_t* fBlocked, int nMP, const uint8_t* inView,
                                 const uint8_t* isBad, const float* depth, const float* projX, const float* projY,
                                 const float* projXR, const int* level, const float* viewCos, const uint8_t* mpDesc,
                                 const uint8_t* mpHasObs, float th, int bFarPoints, float thFarPoints, float nnratio,
                                 int* matchF);
/* Frame::isInFrustumChecks for one camera of a KB8 rig; SearchByProjection(F, MapPoints) with F.Nleft != -1 */
void orc_is_in_frustum_kb8(const orc_frame* F, const float* cam8, const float* mR, const float* mt, const float* twc, int nMP,
                           const float* Pw, const float* normal, const float* maxDist, const float* minDist,
                           float viewingCosLimit, uint8_t* inView, float* projX, float* projY, float* depth, int* level,
                           float* viewCos);
int orc_search_by_projection_mps_fisheye(const orc_frame* F, int Nleft, const int* l2r, const int* r2l, const uint8_t* fBlocked,
                                         int nMP, const uint8_t* inViewL, const uint8_t* inViewR, const uint8_t* isBad,
                                         const float* depthL, const float* projXL, const float* projYL, const int* levelL,
                                         const float* viewCosL, const float* projXR, const float* projYR, const int* levelR,
                                         const float* viewCosR, const uint8_t* mpDesc, const uint8_t* mpHasObs, float th,
                                         int bFarPoints, float thFarPoints, float nnratio, int* matchF);
/* Tracking.cc's host loops between the searches and PoseOptimization (Frame.cc:541-585, Optimizer.cc:803-905, Tracking.cc:2716-2740, :3117-3133) */
void orc_frame_set_pose(const float* Tcw7, float* Rcw, float* tcw, float* Ow);
void orc_pose_edges(const orc_frame* F, const float* invLevelSigma2, const int* frameMP, const float* mpXw, uint8_t* hasMP, float* obs,
                    float* invSigma2, float* Xw);
int orc_discard_outliers(int N, int* frameMP, uint8_t* outlier, int nMP, const uint8_t* mpHasObs, uint8_t* blocked, uint8_t* mpSeen,
                         int* nmatchesMap);
int orc_search_by_projection_last(const orc_frame* Cur, const uint8_t* curBlocked, const float* Tcw7, int nLast,
                                  const orc_keypoint* lastKpsUn, const uint8_t* lastValid, const float* lastXw,
                                  const uint8_t* lastMPdesc, const uint8_t* lastMPhasObs, float th, int bForward,
                                  int bBackward, int checkOri, int* matchCur);
int orc_search_by_projection_last_fisheye(const orc_frame* Cur, int NleftCur, const float* cam8, const float* Trl7,
                                          const uint8_t* curBlocked, const float* Tcw7, int nLast, const orc_keypoint* lastKps,
                                          const uint8_t* lastValid, const float* lastXw, const uint8_t* lastMPdesc,
                                          const uint8_t* lastMPhasObs, float th, int bForward, int bBackward, int checkOri,
                                          int* matchCur);
int orc_search_by_projection_kf(const orc_frame* Cur, const uint8_t* curHasMP, const float* Tcw7, const float* Ow, int nKF,
                                const orc_keypoint* kfKpsUn, const uint8_t* kfValid, const float* Xw, const float* maxDist,
                                const float* minDist, const uint8_t* mpDesc, float th, int ORBdist, int checkOri, int* matchCur);
int orc_search_by_projection_kf_rig(const orc_frame* CurLeft, const float* cam8, const uint8_t* curHasMP, const float* Tcw7, const float* Ow, int nKF,
                                    const orc_keypoint* kfKps, const uint8_t* kfValid, const float* Xw, const float* maxDist,
                                    const float* minDist, const uint8_t* mpDesc, float th, int ORBdist, int checkOri, int* matchCur);
int orc_search_for_initialization(int n1, const orc_keypoint* kps1, const uint8_t* desc1, const orc_frame* F2, float* prevMatched,
                                  int windowSize, float nnratio, int checkOri, int* matches12);
void orc_fundamental_f12(const float* K1, const float* K2, const float* R12, const float* t12, float* F12);
int orc_search_for_triangulation(int n1, const orc_keypoint* kps1, const uint8_t* desc1, const int* node1,
                                 const uint8_t* hasMP1, const float* uRight1, const float* sigma2_1, int n2,
                                 const orc_keypoint* kps2, const uint8_t* desc2, const int* node2, const uint8_t* hasMP2,
                                 const float* uRight2, const float* sigma2_2, const float* scaleFactors2, const float* F12,
                                 const float* ep, int bOnlyStereo, int bCoarse, int checkOri, int* matches12);
int orc_search_for_triangulation_fisheye(int n1, int NLeft1, const orc_keypoint* kps1, const uint8_t* desc1, const int* node1,
                                         const uint8_t* hasMP1, int n2, int NLeft2, const orc_keypoint* kps2, const uint8_t* desc2,
                                         const int* node2, const uint8_t* hasMP2, const float* levelSigma2, const float* camL8,
                                         const float* camR8, const float* T4, int bOnlyStereo, int bCoarse, int checkOri,
                                         int* matches12);
float orc_kb8_triangulate_matches(const float* cam1_8, const float* cam2_8, float x1, float y1, float x2, float y2, const float* R12,
                                  const float* t12, float sigmaLevel, float unc, float* p3D);
/* M7 (N2): Fuse x2, SearchByProjection(KF, Sim3) x2, one direction of SearchBySim3 (ORBmatcher.cc:397-601, :1044-1519) */
void orc_fuse_search(const orc_frame* KF, const float* invLevelSigma2, const float* Tcw7, const float* Ow, int nMP, const uint8_t* valid,
                     const float* Pw, const float* normal, const float* maxDist, const float* minDist, const uint8_t* mpDesc, float th,
                     int sim3Form, int* bestIdx, int* bestDist);
void orc_fuse_search_rig(const orc_frame* KF, int NLeft, int bRight, const float* cam8, const float* invLevelSigma2, const float* Tcw7,
                         const float* Ow, int nMP, const uint8_t* valid, const float* Pw, const float* normal, const float* maxDist,
                         const float* minDist, const uint8_t* mpDesc, float th, int* bestIdx, int* bestDist);
int orc_search_by_projection_sim3_rig(const orc_frame* KF, int NLeft, const float* cam8, const float* Tcw7, const float* Ow, int nMP, const uint8_t* valid,
                                      const float* Pw, const float* normal, const float* maxDist, const float* minDist, const uint8_t* mpDesc,
                                      const uint8_t* matchedIn, int th, float ratioHamming, int manualProjection, int* matchF);
void orc_search_by_sim3_dir_rig(const orc_frame* B, int NLeftB, const float* TAw7, const float* SBA8, int nA, const uint8_t* valid, const float* Pw,
                                const float* maxDist, const float* minDist, const uint8_t* mpDesc, float th, int* vnMatch);
void orc_fuse_search_rig_sim3(const orc_frame* KF, int NLeft, const float* cam8, const float* invLevelSigma2, const float* Tcw7, const float* Ow, int nMP,
                              const uint8_t* valid, const float* Pw, const float* normal, const float* maxDist, const float* minDist,
                              const uint8_t* mpDesc, float th, int* bestIdx, int* bestDist);
int orc_search_by_projection_sim3(const orc_frame* KF, const float* Tcw7, const float* Ow, int nMP, const uint8_t* valid,
                                  const float* Pw, const float* normal, const float* maxDist, const float* minDist,
                                  const uint8_t* mpDesc, const uint8_t* matchedIn, int th, float ratioHamming, int manualProjection,
                                  int* matchF);
void orc_search_by_sim3_dir(const orc_frame* B, const float* TAw7, const float* SBA8, int nA, const uint8_t* valid, const float* Pw,
                            const float* maxDist, const float* minDist, const uint8_t* mpDesc, float th, int* vnMatch);
void orc_kb8_project_f(const float* cam8, const float* v3, float* uv);
void orc_kb8_project_d(const float* cam8, const double* v3, double* uv);
void orc_kb8_unproject(const float* cam8, float x, float y, float* ray);
void orc_kb8_project_jac(const float* cam8, const double* v3, double* J6);
int orc_stereo_fisheye_matches(int Nleft, int monoLeft, const orc_keypoint* kpsL, const uint8_t* descL, int Nright,
                               int monoRight, const orc_keypoint* kpsR, const uint8_t* descR, const float* camL8,
                               const float* camR8, const float* Rlr, const float* tlr, const float* levelSigma2,
                               int* leftToRight, int* rightToLeft, float* depth, float* p3D);
int orc_pose_optimization_fisheye(int n, int Nleft, const uint8_t* hasMP, const float* obs, const float* invSigma2,
                                  const float* Xw, const float* camL8, const float* camR8, const float* Trl7, float* pose,
                                  uint8_t* outlier, int* stats);
int orc_local_ba_fisheye(int nKF, float* kfPose, const uint8_t* kfFixed, int nMP, float* mpPos, int nE, const int* eKF,
                         const int* eMP, const float* eObs2, const uint8_t* eRight, const float* eInvSigma2, const float* camL8,
                         const float* camR8, const float* Trl7, int lambdaInit100, const int* stopFlag, uint8_t* eraseFlag,
                         int* stats);
/* ---- N1 slice: IMU preintegration + PoseInertialOptimizationLastKeyFrame (oracle/inertial.cc) ---- */
typedef struct {
  float dT;
  float dR[9], dV[3], dP[3];                 /* row-major 3x3 */
  float JRg[9], JVg[9], JVa[9], JPg[9], JPa[9];
  float C[225];                              /* 15 x 15 row-major */
  float b[6];                                /* bax bay baz bwx bwy bwz (IMU::Bias order) */
  float nga[6], ngaWalk[6];                  /* diagonals of Calib::Cov / CovWalk: gyro x3, acc x3 */
  float avgA[3], avgW[3];
} orc_imu_preintegrated;
void orc_imu_preintegrate(const float* bias6, const float* ngaDiag6, const float* walkDiag6, int n, const float* acc,
                          const float* gyro, const float* dt, orc_imu_preintegrated* out);
void orc_imu_delta(const orc_imu_preintegrated* P, const float* b1, float* dR, float* dV, float* dP);
int orc_pose_inertial_optimization_last_keyframe(int n, const uint8_t* hasMP, const float* obs, const float* invSigma2,
                                                 const float* Xw, const uint8_t* closeFlag, float fx, float fy, float cx,
                                                 float cy, float bf, const float* Tbc12, const float* kfState21,
                                                 const orc_imu_preintegrated* pre, int bRecInit, float* state21,
                                                 uint8_t* outlier, double* prior246);
int orc_pose_inertial_optimization_last_frame(int n, const uint8_t* hasMP, const float* obs, const float* invSigma2,
                                              const float* Xw, const uint8_t* closeFlag, float fx, float fy, float cx, float cy,
                                              float bf, const float* Tbc12, const float* prevState21,
                                              const orc_imu_preintegrated* preFrame, const orc_imu_preintegrated* preKF,
                                              const double* prevPrior246, int bRecInit, float* state21, uint8_t* outlier,
                                              double* prior246);
int orc_local_inertial_ba(int nKF, float* kfState21, const uint8_t* kfKind, int nMP, float* mpPos, const uint8_t* mpClose, int nE,
                          const int* eKF, const int* eMP, const float* eObs, const float* eInvSigma2, int nI, const int* iKF1,
                          const int* iKF2, const orc_imu_preintegrated* iPre, const uint8_t* iRobust, const float* iInfoScale,
                          float fx, float fy, float cx, float cy, float bf, const float* Tbc12, int bLarge, uint8_t* eraseFlag,
                          int* stats2);
/* fisheye-rig forms: rig28 = left KB8 (8), right KB8 (8), Trl rotation (9, row-major) + translation (3); features / edges are
 * monocular on the left (index < Nleft, eRight == 0) or the right camera */
int orc_pose_inertial_optimization_last_keyframe_fisheye(int n, int Nleft, const uint8_t* hasMP, const float* obs,
                                                         const float* invSigma2, const float* Xw, const uint8_t* closeFlag,
                                                         const float* rig28, const float* Tbc12, const float* kfState21,
                                                         const orc_imu_preintegrated* pre, int bRecInit, float* state21,
                                                         uint8_t* outlier, double* prior246);
int orc_pose_inertial_optimization_last_frame_fisheye(int n, int Nleft, const uint8_t* hasMP, const float* obs, const float* invSigma2,
                                                      const float* Xw, const uint8_t* closeFlag, const float* rig28, const float* Tbc12,
                                                      const float* prevState21, const orc_imu_preintegrated* preFrame,
                                                      const orc_imu_preintegrated* preKF, const double* prevPrior246, int bRecInit,
                                                      float* state21, uint8_t* outlier, double* prior246);
int orc_local_inertial_ba_fisheye(int nKF, float* kfState21, const uint8_t* kfKind, int nMP, float* mpPos, const uint8_t* mpClose,
                                  int nE, const int* eKF, const int* eMP, const float* eObs, const uint8_t* eRight,
                                  const float* eInvSigma2, int nI, const int* iKF1, const int* iKF2, const orc_imu_preintegrated* iPre,
                                  const uint8_t* iRobust, const float* iInfoScale, const float* rig28, const float* Tbc12, int bLarge,
                                  uint8_t* eraseFlag, int* stats2);
void orc_undistort_points(int n, const float* xy, float fx, float fy, float cx, float cy, const float* dist5, float pfx, float pfy,
                          float pcx, float pcy, float* out);
void orc_stereo_from_rgbd(int n, const float* kp, const float* kpUn, const float* depth, int W, int H, float bf, float* uRight,
                          float* depthOut);
int orc_bow_vector(int n, const int* leaf, const int* nodeWordId, const double* nodeWeight, int weighting, int scoring, int* outWord,
                   double* outValue);
float orc_fast_atan2(float y, float x);
float orc_cosf(float x);
float orc_sinf(float x);

#ifdef __cplusplus
}
#endif
