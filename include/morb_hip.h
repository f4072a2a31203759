/* morb_hip.h — C ABI of libmorb_hip.so: the MI355X (gfx950) implementation of MORB_SLAM's per-frame front end.
 *
 * The reference has no FFI layer: its boundary for this path is three C++ class surfaces called in-process
 * (SURVEY.md §8b).  Each entry point below names the reference interface it replaces (paths relative to the
 * reference root).  Plain pointers, sizes and PODs only; every function returns an int status
 * (MORB_OK = 0, negative = error) unless documented otherwise; no exceptions cross this boundary.
 * Pointers named d_* are DEVICE pointers (HIP), everything else is host memory.  `stream` is a hipStream_t
 * passed as void* (NULL = the handle's own stream).
 */
#ifndef MORB_HIP_H
#define MORB_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MORB_OK 0
#define MORB_ERR_INVALID (-1)   /* bad argument */
#define MORB_ERR_HIP (-2)       /* HIP runtime error; see morb_last_error() */
#define MORB_ERR_CAPACITY (-3)  /* caller buffer too small */
#define MORB_ERR_UNSUPPORTED (-4)
#define MORB_ERR_EMPTY (-5)     /* empty image: ORBextractor::operator() returns -1 (src/ORBextractor.cc:1011) */

/* cv::KeyPoint as the reference stores it (28 bytes): pt.x, pt.y, size, angle, response, octave, class_id */
typedef struct morb_keypoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} morb_keypoint;

const char* morb_last_error(void);
int morb_device_count(void);

/* ------------------------------------------------------------------------------------------------------
 * ORBextractor  (include/ORBextractor.h:44-105, src/ORBextractor.cc)
 * ---------------------------------------------------------------------------------------------------- */
typedef struct morb_extractor morb_extractor;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)  ORBextractor.h:49-50,
 * ORBextractor.cc:406-464.  `device` = HIP device ordinal. */
int morb_extractor_create(morb_extractor** out, int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                          int minThFAST, int device);
void morb_extractor_destroy(morb_extractor*);

/* GetLevels / GetScaleFactor / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares  (ORBextractor.h:59-74); arrays hold nlevels entries; any pointer may be NULL. */
int morb_extractor_levels(const morb_extractor*);
float morb_extractor_scale_factor(const morb_extractor*);
int morb_extractor_tables(const morb_extractor*, float* scaleFactors, float* invScaleFactors, float* sigma2,
                          float* invSigma2, int* featuresPerLevel);

/* Upper bound of keypoints one image can yield (size the kps/desc buffers with it). */
int morb_extractor_max_keypoints(const morb_extractor*);

/* int ORBextractor::operator()(image, mask (ignored), keypoints, descriptors, vLappingArea)
 * ORBextractor.h:55-57, ORBextractor.cc:1006-1086.  Host image in (CV_8UC1, `stride` bytes per row), host
 * keypoints (28 B each) and descriptors (32 B each) out, *n = number of keypoints.
 * Returns monoIndex (>= 0) exactly as the reference does, MORB_ERR_EMPTY for an empty image (reference: -1),
 * or another negative status. */
int morb_extract(morb_extractor*, const uint8_t* image, int width, int height, int stride, int lap0, int lap1,
                 morb_keypoint* kps, uint8_t* desc, int cap, int* n);

/* Batched, device-resident form of operator(): nimg images of identical size, image i at
 * d_images + i*image_pitch.  lap = host array [nimg][2] of lapping areas, or NULL for {0,0} (the rectified
 * stereo call, Frame.cc:194-197).  Outputs are device arrays: d_kps [nimg][cap], d_desc [nimg][cap][32],
 * d_count [nimg], d_mono [nimg] (monoIndex); cap >= morb_extractor_max_keypoints() (else MORB_ERR_CAPACITY: the lapping-area
 * keypoints fill every image's range from the back).  Asynchronous on `stream`. */
int morb_extract_batch(morb_extractor*, const uint8_t* d_images, int nimg, int width, int height, int stride,
                       size_t image_pitch, const int* lap, morb_keypoint* d_kps, uint8_t* d_desc, int cap,
                       int* d_count, int* d_mono, void* stream);

/* std::vector<cv::Mat> mvImagePyramid (public member, ORBextractor.h:76; read by
 * Frame::ComputeStereoMatches, Frame.cc:895,974,987).  Level `lvl` of image `img` of the last batch:
 * *d_ptr points at the interior origin (the cv::Mat ROI), the 19-px BORDER_REFLECT_101 pad surrounds it in
 * the same allocation, *stride is the row pitch. */
int morb_extractor_pyramid_level(const morb_extractor*, int img, int lvl, const uint8_t** d_ptr, int* width,
                                 int* height, int* stride);
/* Same, copied to host, including the pad: out is (height+38) x (width+38) contiguous. */
int morb_extractor_pyramid_level_host(const morb_extractor*, int img, int lvl, uint8_t* out_padded);
/* Debug/parity taps on the last batch (host copies): blurred level (height x width contiguous), FAST
 * candidates of a level in vToDistributeKeys order, keypoints of a level after DistributeOctTree. */
int morb_extractor_blurred_level_host(const morb_extractor*, int img, int lvl, uint8_t* out);
int morb_extractor_level_candidates_host(const morb_extractor*, int img, int lvl, morb_keypoint* out, int cap,
                                         int* n);
int morb_extractor_level_keypoints_host(const morb_extractor*, int img, int lvl, morb_keypoint* out, int cap,
                                        int* n);

/* Per-stage device time: with profiling enabled every morb_extract_batch records HIP events on its stream at
 * the stage boundaries (no host synchronisation in the call).  morb_extractor_stage_ms waits for the recorded
 * calls (at most the last 64), writes the per-call AVERAGE in ms for stages 0..6 = pyramid, blur, fast,
 * distribute, layout, describe, total, resets the window and returns the number of calls averaged.
 * Mirrors the reference's REGISTER_TIMES spans (src/Frame.cc:190-206). */
int morb_extractor_set_profiling(morb_extractor*, int enable);
int morb_extractor_stage_ms(morb_extractor*, float* ms7);
/* The event (hipEvent_t) the last morb_extract_batch recorded on its stream after the FAST stage — where the memory-bound half of the
 * extraction (pyramid, FAST's window loads) ends and the quadtree's mostly idle chip begins.  A caller that pipelines frames can make
 * another stream wait for it (hipStreamWaitEvent) so that other work — the previous frame's matchers — lands underneath the quadtree
 * instead of beside the next pyramid.  Owned by the handle; valid until the handle is destroyed. */
int morb_extractor_event_after_fast(morb_extractor*, void** event);
/* Likewise the event recorded behind the last pyramid launch (ComputePyramid, src/ORBextractor.cc:1088-1112), i.e. where the FAST stage starts:
 * k_fastw is bound by vector-instruction issue and leaves the memory pipe idle, so a pipelining caller may prefer to put the previous frame's
 * matchers (BoW descent, SAD rows: memory-pipe work) underneath IT rather than beside the pyramid, which needs the same pipe.
 * The event is recorded by the extractions queued AFTER the first call of this function on the handle (call it once before the first batch):
 * callers that never ask do not pay for an event between the pyramid and FAST. */
int morb_extractor_event_after_pyramid(morb_extractor*, void** event);
/* Failure flags raised on the device by the extractions since the last query (cleared by the call).  To be read AFTER the stream of a
 * morb_extract_batch call has been synchronised; morb_extract checks it itself.  Bit 0: a pyramid level held more than 65535 FAST
 * candidates (noise-like images; the quadtree counts children in 16 bits) -> returns MORB_ERR_UNSUPPORTED, that call's keypoints are not valid. */
int morb_extractor_status(morb_extractor*, int* flags);
/* hipStreamWaitEvent(stream, event) through the HIP runtime THIS library is linked against: a host language that holds raw stream /
 * event handles (ctypes, cgo) must not open a second copy of libamdhip64 to make `stream` wait for the event above. */
int morb_stream_wait_event(void* stream, void* event);

/* ------------------------------------------------------------------------------------------------------
 * Matchers: ORBmatcher (include/ORBmatcher.h:36-129, src/ORBmatcher.cc) and the per-frame stereo matchers
 * of Frame.cc.  The reference walks Frame/KeyFrame/MapPoint objects; the ABI takes the fields each function
 * reads, flattened: features of image i live at [i*cap, i*cap + count[i]) of the keypoint / descriptor arrays
 * (the layout morb_extract_batch writes).  All array arguments are DEVICE pointers; calls are asynchronous
 * on `stream` (NULL = the matcher's own stream).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct morb_matcher morb_matcher;
int morb_matcher_create(morb_matcher** out, int device);
void morb_matcher_destroy(morb_matcher*);
/* Wait for the work queued on the matcher's OWN stream (calls made with stream = NULL) — not for the device: another handle's work
 * (a LocalBundleAdjustment trial on the mapping thread's optimizer, System.cc:209) keeps running. */
int morb_matcher_sync(morb_matcher*);
/* The matcher's own stream (hipStream_t): a host adapter queues its uploads / downloads there instead of on the null stream. */
void* morb_matcher_stream(const morb_matcher*);

/* Feature slabs — the unit one GPU ships to another (no counterpart in the single-process reference; north_star's frame sharding).
 * A frame matched against its predecessor (SearchByBoW / SearchByProjection(Cur, Last)) needs the predecessor's keypoints, descriptors and
 * BoW node ids; with frames dealt round-robin over GPUs they live on the previous GPU.  pack gathers S rows of the [nimg][cap] feature arrays
 * (row d_rows[f], or row f when d_rows is NULL — e.g. the left images 0, 2, 4, ...) into ONE contiguous buffer of morb_feature_slab_bytes(S, cap)
 * bytes: [S][cap] keypoints | [S][cap][32] descriptors | [S][cap] node ids (-1 when d_node is NULL) | [S] counts.  The caller moves that one
 * buffer (hipMemcpyPeerAsync over xGMI, or an RCCL send / recv) and unpack scatters it into rows d_rows[f] of the receiver's pool.  All
 * pointers are DEVICE pointers; asynchronous on `stream`.  INTEGRATION.md section 5 shows the sharding loop of a C++ Tracking.
 * PRECONDITIONS (not checked: the sizes live in device memory): the slab holds morb_feature_slab_bytes(S, cap) bytes; every d_rows[f]
 * (or f itself when d_rows is NULL) is a row of arrays allocated with the SAME cap as the packer's; the counts inside the slab are
 * <= cap (pack writes min(count, cap)).  A slab packed with another cap must not be unpacked: the row pitch is part of the format. */
size_t morb_feature_slab_bytes(int S, int cap);
int morb_feature_slab_pack(morb_matcher*, int S, int cap, const int* d_rows, const morb_keypoint* d_kps, const uint8_t* d_desc, const int* d_node,
                           const int* d_count, void* d_slab, void* stream);
int morb_feature_slab_unpack(morb_matcher*, int S, int cap, const void* d_slab, const int* d_rows, morb_keypoint* d_kps, uint8_t* d_desc, int* d_node,
                             int* d_count, void* stream);

/* static int ORBmatcher::DescriptorDistance(a, b)  ORBmatcher.h:43, ORBmatcher.cc:1880-1894; n pairs of 32-byte
 * descriptors -> n distances. */
int morb_hamming_pairs(morb_matcher*, const uint8_t* d_a, const uint8_t* d_b, int n, int* d_out, void* stream);

/* cv::BFMatcher(NORM_HAMMING).knnMatch(query, train, matches, 2) as Frame::ComputeStereoFishEyeMatches uses
 * it (Frame.cc:46, :1242).  Problem p: query rows [qOff[p], nq[p]) of d_query + p*qPitch*32, train rows
 * [tOff[p], nt[p]) likewise (qOff/tOff NULL = 0: monoLeft / monoRight in the reference).  Output for query row
 * r (relative to qOff): d_idx[(p*qPitch + r)*2 + {0,1}] = train index relative to tOff (-1 if absent),
 * d_dist likewise; ties: lower train index first. */
int morb_hamming_knn2_batch(morb_matcher*, int nprob, const uint8_t* d_query, const int* d_nq, int qPitch,
                            const int* d_qOff, const uint8_t* d_train, const int* d_nt, int tPitch, const int* d_tOff,
                            int* d_idx, int* d_dist, void* stream);

/* void Frame::ComputeStereoMatches()  Frame.cc:889-1047 for nframes rectified stereo frames whose images the
 * extractor has just processed in ONE batch with left = image 2f, right = image 2f+1 (it reads the
 * extractor's mvImagePyramid like the reference does, Frame.cc:895,974,987).  mbf, mb as Frame::mbf / mb.
 * Outputs mvuRight / mvDepth: d_uRight[f*cap + i], d_depth[f*cap + i] for left keypoint i (-1 = no match). */
int morb_stereo_match_batch(morb_matcher*, const morb_extractor*, int nframes, const morb_keypoint* d_kps,
                            const uint8_t* d_desc, const int* d_count, int cap, float mbf, float mb, float* d_uRight,
                            float* d_depth, void* stream);

/* void Frame::ComputeStereoFishEyeMatches()  Frame.cc:1222-1274 for nframes fisheye stereo frames (left = image 2f,
 * right = image 2f+1 of one extract batch): brute-force 2-NN over the lapping-area features (from index
 * d_mono[image] = the monoIndex ORBextractor::operator() returned), ratio 0.7, KannalaBrandt8::TriangulateMatches.
 * camL8 / camR8 (HOST) = fx fy cx cy k0..k3; Rlr9 / tlr3 (HOST) = Frame::mRlr / mtlr; levelSigma2 = mvLevelSigma2.
 * Outputs [nframes][cap]: mvLeftToRightMatch, mvRightToLeftMatch, mvDepth, mvStereo3Dpoints ([..][3]); d_nMatches[f]. */
int morb_stereo_fisheye_match_batch(morb_matcher*, int nframes, const morb_keypoint* d_kps, const uint8_t* d_desc,
                                    const int* d_count, const int* d_mono, int cap, const float* camL8, const float* camR8,
                                    const float* Rlr9, const float* tlr3, const float* levelSigma2, int nlevels,
                                    int* d_leftToRight, int* d_rightToLeft, float* d_depth, float* d_p3D, int* d_nMatches,
                                    void* stream);

/* DBoW2 TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
 * (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1218-1259) for every feature of nimg images: the vocabulary is
 * a k-ary tree in arrays (node 0 = root; children of n = firstChild[n] .. firstChild[n]+k-1; firstChild < 0 =
 * leaf; one 32-byte descriptor per node).  d_wordId = leaf node reached, d_nodeId = ancestor at level
 * L - levelsup: the FeatureVector key that SearchByBoW / SearchForTriangulation bucket on (Frame::ComputeBoW,
 * Frame.cc:822-827, levelsup = 4). */
int morb_bow_transform_batch(morb_matcher*, int nimg, const uint8_t* d_desc, const int* d_count, int cap,
                             const uint8_t* d_nodeDesc, const int* d_firstChild, int k, int L, int levelsup,
                             int* d_wordId, int* d_nodeId, void* stream);

/* The same descent on a trained vocabulary whose nodes may have fewer than k children (ORBvoc.txt: 1 082 073 nodes, not the
 * 1 111 111 of a complete 10-ary tree): children of node n are [d_firstChild[n], d_firstChild[n] + d_childCount[n]). */
int morb_bow_transform_tree_batch(morb_matcher* m, int nimg, const uint8_t* d_desc, const int* d_count, int cap,
                                  const uint8_t* d_nodeDesc, const int* d_firstChild, const int* d_childCount, int L, int levelsup,
                                  int* d_wordId, int* d_nodeId, void* stream);

/* DBoW2 text vocabulary (TemplatedVocabulary::loadFromTextFile, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1420;
 * SURVEY 8f N3): header "k L scoring weighting", then one node per line "parent isLeaf d0 .. d31 weight"; node ids in file
 * order from 1 (0 = root), word ids = order of the leaves.  Host-side loader; morb_vocabulary_arrays copies the flattened
 * tree (caller-allocated arrays of nNodes entries; any may be NULL) for upload to the device.  wordId[leaf node] maps the
 * transform's leaf node id to DBoW2's WordId. */
typedef struct morb_vocabulary morb_vocabulary;
int morb_vocabulary_load_text(const char* path, morb_vocabulary** out);
void morb_vocabulary_destroy(morb_vocabulary* v);
int morb_vocabulary_info(const morb_vocabulary* v, int* k, int* L, int* nNodes, int* nWords);
int morb_vocabulary_arrays(const morb_vocabulary* v, uint8_t* nodeDesc, int* firstChild, int* childCount, int* wordId, float* weight);
/* the node weights as DBoW2 holds them (WordValue = double) and the header's scoring / weighting types (BowVector.h:39-56:
 * weighting 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; scoring 0 L1_NORM .. 5 DOT_PRODUCT); any pointer may be NULL */
int morb_vocabulary_weights(const morb_vocabulary* v, double* weight, int* scoring, int* weighting);

/* The BowVector half of TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
 * (TemplatedVocabulary.h:1127-1190, BowVector.cpp:34-84), for nimg images: d_leaf = the leaf NODE ids of the descent
 * (morb_bow_transform*'s d_wordId output), d_nodeWordId = node -> WordId (NULL: the node id itself), d_nodeWeight = node weights
 * (double).  Words with weight 0 are "stopped".  Outputs: the map's entries in ascending word order, d_bowWord / d_bowValue
 * [nimg][cap] and d_bowCount[nimg]. */
int morb_bow_vector_batch(morb_matcher*, int nimg, const int* d_leaf, const int* d_count, int cap, const int* d_nodeWordId,
                          const double* d_nodeWeight, int weighting, int scoring, int* d_bowWord, double* d_bowValue, int* d_bowCount,
                          void* stream);

/* int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)
 * ORBmatcher.h:68, ORBmatcher.cc:218-395 (non-fisheye branch) for npairs (keyframe, frame) pairs drawn from a
 * pool of nimg images: pair p matches image d_kfImg[p] (as pKF) against image d_fImg[p] (as F).
 * d_node[i*cap + j] = FeatureVector node id of feature j (negative = not in the FeatureVector);
 * d_hasMP[i*cap + j] != 0 <=> vpMapPointsKF[j] is non-NULL and not bad.  nnratio / checkOri = the ORBmatcher
 * constructor arguments.  d_matchF[p*cap + j] = keyframe feature matched to frame feature j (-1 = none; the
 * adapter maps it to vpMapPointsKF[...]); d_nmatches[p] = the return value. */
int morb_search_by_bow_batch(morb_matcher*, int npairs, const int* d_kfImg, const int* d_fImg, int nimg,
                             const morb_keypoint* d_kps, const uint8_t* d_desc, const int* d_node, const int* d_count,
                             const uint8_t* d_hasMP, int cap, float nnratio, int checkOri, int* d_matchF,
                             int* d_nmatches, void* stream);

/* MapPoint::ComputeDistinctiveDescriptors (MapPoint.cc:367-435) for nMP map points at once: the observed
 * descriptors of point m (the rows the reference pushes into vDescriptors, in observation order) are rows
 * [d_start[m], d_start[m + 1]) of d_desc.  d_bestIdx[m] = index (within the point's rows) of the descriptor with the least
 * median Hamming distance to the others (first minimum), -1 for a point without descriptors; the caller clones that row
 * into mDescriptor.  At most 65535 descriptors per point. */
int morb_distinctive_descriptors_batch(morb_matcher* m, int nMP, const int* d_start, const uint8_t* d_desc, int* d_bestIdx,
                                       void* stream);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) (ORBmatcher.cc:702-819; loop
 * closing / merging): both sides are keyframes of the pool; hasMP[img][i] != 0 <=> GetMapPointMatches()[i] && !isBad();
 * d_nValid[img] = mvKeysUn.size() (features at or beyond it are skipped on fisheye keyframes, :734 / :751; NULL = all).
 * match12[p][idx1] = index of the matched pKF2 feature or -1 (vpMatches12[idx1] = vpMapPoints2[match12]). */
int morb_search_by_bow_kfkf_batch(morb_matcher* m, int npairs, const int* d_kf1Img, const int* d_kf2Img, const int* d_nValid, int nimg,
                                  const morb_keypoint* d_kps, const uint8_t* d_desc, const int* d_node, const int* d_count,
                                  const uint8_t* d_hasMP, int cap, float nnratio, int checkOri, int* d_match12, int* d_nmatches,
                                  void* stream);

/* The same with a fisheye frame F (F.Nleft != -1, ORBmatcher.cc:262-299 and :333-365): image fImg[p] holds the
 * Nleft = d_nLeft[p] left features followed by the right ones; left and right candidates of a node are ranked
 * separately, the right winner is taken whenever the LEFT best distance passes TH_LOW (the reference's nesting and
 * its `|| true`).  d_nLeft[p] = -1 marks a pinhole frame. */
int morb_search_by_bow_fisheye_batch(morb_matcher* m, int npairs, const int* d_kfImg, const int* d_fImg, const int* d_nLeft,
                                     int nimg, const morb_keypoint* d_kps, const uint8_t* d_desc, const int* d_node,
                                     const int* d_count, const uint8_t* d_hasMP, int cap, float nnratio, int checkOri,
                                     int* d_matchF, int* d_nmatches, void* stream);

/* The Frame members the projection-guided searches read, as one POD (all frames of a batch share one camera):
 * mnMinX/Y, mnMaxX/Y, mfGridElementWidthInv/HeightInv, fx, fy, cx, cy, mbf, mb, mfLogScaleFactor, mnScaleLevels,
 * mvScaleFactors, mvLevelSigma2 (include/Frame.h). */
typedef struct morb_frame_params {
  float minX, minY, maxX, maxY, gridInvW, gridInvH;
  float fx, fy, cx, cy, mbf, mb, logScaleFactor;
  int32_t nlevels;
  float scaleFactors[16];
  float levelSigma2[16];
} morb_frame_params;

/* bool Frame::isInFrustum(MapPoint*, viewingCosLimit)  Frame.h / Frame.cc:611-678 (+ MapPoint::PredictScale,
 * MapPoint.cc:536-566) for d_nMP[f] map points of each of nframes frames (arrays [nframes][mpCap]).  Inputs:
 * d_Rcw [f][9] row-major, d_tcw [f][3], d_Ow [f][3] (Frame::mRcw, mtcw, mOw), world position, normal,
 * mfMaxDistance, mfMinDistance per point.  Outputs = the MapPoint tracking fields the function fills:
 * mbTrackInView, mTrackProjX, mTrackProjY, mTrackProjXR, mTrackDepth, mnTrackScaleLevel, mTrackViewCos. */
int morb_is_in_frustum_batch(morb_matcher*, const morb_frame_params*, int nframes, const float* d_Rcw, const float* d_tcw,
                             const float* d_Ow, int mpCap, const int* d_nMP, const float* d_Pw, const float* d_normal,
                             const float* d_maxDist, const float* d_minDist, float viewingCosLimit, uint8_t* d_inView,
                             float* d_projX, float* d_projY, float* d_projXR, float* d_depth, int* d_level,
                             float* d_viewCos, void* stream);

/* int ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th, bFarPoints, thFarPoints)
 * ORBmatcher.h:49-51, ORBmatcher.cc:42-209.  Frame f uses the features of image d_fImg[f]; d_uRight / d_blocked
 * are [nframes][cap] (mvuRight or NULL; blocked != 0 <=> mvpMapPoints[i] && Observations() > 0).  Map points
 * [nframes][mpCap] carry the fields isInFrustum wrote plus isBad, the representative descriptor and
 * hasObs (Observations() > 0).  d_matchF [nframes][cap] in/out: index of the map point assigned to feature i
 * (caller initialises to -1 or keeps earlier assignments); d_nmatches[f] = return value. */
int morb_search_by_projection_mps_batch(morb_matcher*, const morb_frame_params*, int nframes, const int* d_fImg, int cap,
                                        const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc,
                                        const float* d_uRight, const uint8_t* d_blocked, int mpCap, const int* d_nMP,
                                        const uint8_t* d_inView, const uint8_t* d_isBad, const float* d_depth,
                                        const float* d_projX, const float* d_projY, const float* d_projXR, const int* d_level,
                                        const float* d_viewCos, const uint8_t* d_mpDesc, const uint8_t* d_mpHasObs, float th,
                                        int bFarPoints, float thFarPoints, float nnratio, int* d_matchF, int* d_nmatches,
                                        void* stream);

/* Fisheye (KannalaBrandt8) rig, Frame::isInFrustum with Nleft != -1 (Frame.cc:665-677) = Frame::isInFrustumChecks
 * (Frame.cc:1276-1346) once per camera.  One call = one camera: the caller passes mR, mt, twc exactly as :1283-1293
 * builds them (left: mRcw, mtcw, mOw; right: Rrl * mRcw, Rrl * mtcw + trl, mRwc * mTlr.translation() + mOw; row-major
 * 3x3 / 3 floats per frame) and that camera's parameters cam8 = fx fy cx cy k0 k1 k2 k3 (host pointer).  Outputs as
 * morb_is_in_frustum_batch minus mTrackProjXR: left call -> mbTrackInView, mTrackProjX/Y, mTrackDepth,
 * mnTrackScaleLevel, mTrackViewCos; right call -> mbTrackInViewR, mTrackProjXR/YR, mTrackDepthR, mnTrackScaleLevelR,
 * mTrackViewCosR.  Fields of a rejected point are -1 (the reference leaves them stale). */
int morb_is_in_frustum_kb8_batch(morb_matcher* m, const morb_frame_params* P, const float* cam8, int nframes, const float* d_R,
                                 const float* d_t, const float* d_twc, int mpCap, const int* d_nMP, const float* d_Pw,
                                 const float* d_normal, const float* d_maxDist, const float* d_minDist, float viewingCosLimit,
                                 uint8_t* d_inView, float* d_projX, float* d_projY, float* d_depth, int* d_level,
                                 float* d_viewCos, void* stream);

/* ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, bFarPoints, thFarPoints) on a fisheye rig: the
 * F.Nleft != -1 branches of ORBmatcher.cc:42-209.  Image fImg[f] of the pool holds the Nleft left features followed
 * by the right ones (mvKeys | mvKeysRight, mDescriptors = vconcat, Frame.cc:211-214): count = N, d_nLeft[f] = Nleft.
 * d_l2r / d_r2l [nframes][cap] = mvLeftToRightMatch / mvRightToLeftMatch (local indices, -1 = none).  Left-camera
 * fields (*L) and right-camera fields (*R) come from the two morb_is_in_frustum_kb8_batch calls.  matchF[f][j] =
 * map point index claimed by feature j (left and right halves of the same table), -1 otherwise; nmatches as the
 * reference counts them (a stereo partner counts as a second match). */
int morb_search_by_projection_mps_fisheye_batch(morb_matcher* m, const morb_frame_params* P, int nframes, const int* d_fImg,
                                                int cap, const int* d_count, const int* d_nLeft, const morb_keypoint* d_kps,
                                                const uint8_t* d_desc, const int* d_l2r, const int* d_r2l,
                                                const uint8_t* d_blocked, int mpCap, const int* d_nMP,
                                                const uint8_t* d_inViewL, const uint8_t* d_inViewR, const uint8_t* d_isBad,
                                                const float* d_depthL, const float* d_projXL, const float* d_projYL,
                                                const int* d_levelL, const float* d_viewCosL, const float* d_projXR,
                                                const float* d_projYR, const int* d_levelR, const float* d_viewCosR,
                                                const uint8_t* d_mpDesc, const uint8_t* d_mpHasObs, float th, int bFarPoints,
                                                float thFarPoints, float nnratio, int* d_matchF, int* d_nmatches, void* stream);

/* ---- marshalling between the tracking-side searches and PoseOptimization for device-resident batches of frames ----
 * The host loops Tracking.cc runs between its calls (one frame, std::vector<MapPoint*>), restated for [nframes][cap]
 * tables in HBM so that SearchByProjection -> PoseOptimization -> isInFrustum -> SearchByProjection -> PoseOptimization
 * needs no host round trip (morb_slam_amd/tracking.py, bench.py's tracking chain).  The class adapters of include/morb/
 * do these steps on the host where the reference does.  A frame's map points are rows of a per-frame table
 * [nframes][mpCap]; "mvpMapPoints" is an index into it, -1 = NULL.
 *
 * void Frame::SetPose(Tcw) -> UpdatePoseMatrices()  src/Frame.cc:541-585.  d_pose7 [f][7] (unit quaternion xyzw +
 * translation, world -> camera) -> mRcw (row-major 9), mtcw, mOw = Tcw.inverse().translation(). */
int morb_frame_set_pose_batch(morb_matcher*, int nframes, const float* d_pose7, float* d_Rcw, float* d_tcw, float* d_Ow,
                              void* stream);
/* The unary edges Optimizer::PoseOptimization(Frame*) builds (src/Optimizer.cc:803-905): one per feature that holds a map
 * point; obs = (mvKeysUn[i].pt, mvuRight[i]) (mono edge when mvuRight < 0), invSigma2 = mvInvLevelSigma2[octave], Xw = the
 * map point's world position.  d_match == NULL: d_frameMP [f][cap] is read.  d_match != NULL: d_frameMP is WRITTEN from it,
 * through d_remap [f][remapCap] when given (SearchByProjection(Cur, Last) returns indices of last-frame features;
 * LastFrame.mvpMapPoints as indices is the remap).  Outputs are morb_pose_optimization_batch's inputs. */
int morb_pose_edges_batch(morb_matcher*, const morb_frame_params*, int nframes, const int* d_fImg, int cap, const int* d_count,
                          const morb_keypoint* d_kps, const float* d_uRight, const int* d_match, const int* d_remap, int remapCap,
                          int mpCap, const float* d_mpXw, int* d_frameMP, uint8_t* d_hasMP, float* d_obs, float* d_invSigma2,
                          float* d_Xw, void* stream);
/* Tracking::TrackWithMotionModel's outlier discard (src/Tracking.cc:2716-2740), SearchLocalPoints' first loop (:3117-3133)
 * and TrackLocalMap's inlier count (:2779-2806): features flagged by PoseOptimization lose their map point (d_outlier is
 * cleared), d_nmatches[f] = features that keep one, d_nmatchesMap[f] = those whose point has observations
 * (d_mpHasObs [f][mpCap], NULL = all), d_blocked [f][cap] (nullable) = the `mvpMapPoints[i] && Observations() > 0` flag
 * SearchByProjection(F, MapPoints) skips on, d_mpSeen [f][mpCap] (nullable) = mnLastFrameSeen == this frame (kept or just
 * lost): the points SearchLocalPoints must not project again. */
int morb_track_discard_outliers_batch(morb_matcher*, int nframes, const int* d_fImg, int cap, const int* d_count, int* d_frameMP,
                                      uint8_t* d_outlier, int mpCap, const uint8_t* d_mpHasObs, uint8_t* d_blocked,
                                      uint8_t* d_mpSeen, int* d_nmatches, int* d_nmatchesMap, void* stream);

/* int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono)
 * ORBmatcher.h:55-56, ORBmatcher.cc:1521-1733.  Frame pair f = (image d_curImg[f], image d_lastImg[f]); d_Tcw
 * [f][7] = CurrentFrame pose (quaternion xyzw + translation); per last-frame feature [nframes][cap]:
 * lastValid (map point present and not outlier), its world position, representative descriptor, hasObs.
 * d_bForward / d_bBackward [nframes]: the two flags of :1538-1539 (computed by the caller from the two poses).
 * d_matchCur [nframes][cap] in/out: index of the last-frame feature whose map point is assigned to current
 * feature i. */
int morb_search_by_projection_last_batch(morb_matcher*, const morb_frame_params*, int nframes, const int* d_curImg,
                                         const int* d_lastImg, int cap, const int* d_count, const morb_keypoint* d_kps,
                                         const uint8_t* d_desc, const float* d_curURight, const uint8_t* d_curBlocked,
                                         const float* d_Tcw, const uint8_t* d_lastValid, const float* d_lastXw,
                                         const uint8_t* d_lastMPdesc, const uint8_t* d_lastMPhasObs, float th,
                                         const uint8_t* d_bForward, const uint8_t* d_bBackward, int checkOri, int* d_matchCur,
                                         int* d_nmatches, void* stream);

/* The same with a fisheye current frame (CurrentFrame.Nleft != -1): left pass + right pass of ORBmatcher.cc:1521-1733.
 * Image curImg[f] holds the d_nLeftCur[f] left features followed by the right ones; cam8 = the LEFT camera's KB8
 * parameters (the reference projects both passes with mpCamera), Trl7 = GetRelativePoseTrl() as quaternion xyzw +
 * translation (host pointers).  Projections go through the device's atan2f / cosf / sinf (ulp-level differences from
 * the host libm can move a candidate that sits exactly on a window edge). */
int morb_search_by_projection_last_fisheye_batch(morb_matcher* m, const morb_frame_params* P, const float* cam8, const float* Trl7,
                                                 int nframes, const int* d_curImg, const int* d_lastImg, const int* d_nLeftCur,
                                                 int cap, const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc,
                                                 const uint8_t* d_curBlocked, const float* d_Tcw, const uint8_t* d_lastValid,
                                                 const float* d_lastXw, const uint8_t* d_lastMPdesc,
                                                 const uint8_t* d_lastMPhasObs, float th, const uint8_t* d_bForward,
                                                 const uint8_t* d_bBackward, int checkOri, int* d_matchCur, int* d_nmatches,
                                                 void* stream);

/* int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, th, ORBdist)
 * ORBmatcher.h:60-62, ORBmatcher.cc:1735-1842 (relocalisation refinement).  Pair f = (current image d_curImg[f],
 * keyframe image d_kfImg[f]); d_Tcw [f][7], d_Ow [f][3] = Tcw.inverse().translation(); per keyframe feature
 * [nframes][cap]: kfValid (map point present, not bad, not in sAlreadyFound), world position, mfMaxDistance,
 * mfMinDistance, representative descriptor.  d_curHasMP [nframes][cap] != 0 <=> CurrentFrame.mvpMapPoints[i].
 * d_matchCur in/out as in the last-frame variant. */
int morb_search_by_projection_kf_batch(morb_matcher*, const morb_frame_params*, int nframes, const int* d_curImg,
                                       const int* d_kfImg, int cap, const int* d_count, const morb_keypoint* d_kps,
                                       const uint8_t* d_desc, const uint8_t* d_curHasMP, const float* d_Tcw, const float* d_Ow,
                                       const uint8_t* d_kfValid, const float* d_Xw, const float* d_maxDist,
                                       const float* d_minDist, const uint8_t* d_mpDesc, float th, int ORBdist, int checkOri,
                                       int* d_matchCur, int* d_nmatches, void* stream);

/* The same member when CurrentFrame is a KannalaBrandt8 rig frame (CurrentFrame.Nleft != -1).  The reference has no rig branch here: it
 * projects with CurrentFrame.mpCamera (cam8 = the LEFT camera's fx fy cx cy k0..k3, host pointer) and GetFeaturesInArea's bRight
 * defaults to false, so only the current frame's left features [0, d_nLeftCur[f]) are searched; rows hold left | right features.
 * Rotation check: the reference reads pKF->mvKeysUn[i] for every keyframe map point i, which is out of bounds for the keyframe's RIGHT
 * features (mvKeysUn holds the NLeft left keypoints); here feature i's own keypoint is used — mvKeysRight[i - NLeft] for a right
 * feature — i.e. row entry i of the keyframe image. */
int morb_search_by_projection_kf_rig_batch(morb_matcher*, const morb_frame_params*, const float* cam8, int nframes, const int* d_curImg,
                                           const int* d_kfImg, const int* d_nLeftCur, int cap, const int* d_count,
                                           const morb_keypoint* d_kps, const uint8_t* d_desc, const uint8_t* d_curHasMP, const float* d_Tcw,
                                           const float* d_Ow, const uint8_t* d_kfValid, const float* d_Xw, const float* d_maxDist,
                                           const float* d_minDist, const uint8_t* d_mpDesc, float th, int ORBdist, int checkOri,
                                           int* d_matchCur, int* d_nmatches, void* stream);


/* int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vbPrevMatched, vnMatches12, windowSize)
 * ORBmatcher.h:80-82, ORBmatcher.cc:603-700 (monocular initialisation).  Pair p = (image d_img1[p], image d_img2[p]);
 * d_prevMatched [npairs][cap][2] in/out (vbPrevMatched); d_matches12 [npairs][cap] = vnMatches12. */
int morb_search_for_initialization_batch(morb_matcher*, const morb_frame_params*, int npairs, const int* d_img1,
                                         const int* d_img2, int cap, const int* d_count, const morb_keypoint* d_kps,
                                         const uint8_t* d_desc, float* d_prevMatched, int windowSize, float nnratio,
                                         int checkOri, int* d_matches12, int* d_nmatches, void* stream);

/* int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, vMatchedPairs, bOnlyStereo, bCoarse)
 * ORBmatcher.h:84-87, ORBmatcher.cc:821-1042 (pinhole keyframes, no second camera) for npairs keyframe pairs of
 * a pool of nimg images (arrays [nimg][cap]; d_uRight may be NULL).  R12, t12 (HOST, [npairs][9], [npairs][3]) =
 * T1w * Tw2; ep (HOST [npairs][2]) = projection of KF1's centre into KF2 (:832-835).  d_match12 [npairs][cap] =
 * vMatches12 (feature of KF2 matched to feature i of KF1, -1 = none; the pair list is its non-negative entries in
 * ascending i), d_nmatches = return value. */
int morb_search_for_triangulation_batch(morb_matcher*, const morb_frame_params*, int npairs, const int* d_img1,
                                        const int* d_img2, int nimg, int cap, const int* d_count, const morb_keypoint* d_kps,
                                        const uint8_t* d_desc, const int* d_node, const uint8_t* d_hasMP, const float* d_uRight,
                                        const float* R12, const float* t12, const float* ep, int bOnlyStereo, int bCoarse,
                                        int checkOri, int* d_match12, int* d_nmatches, void* stream);

/* SearchForTriangulation between two keyframes of a KannalaBrandt8 rig (pKF->mpCamera2 != NULL, ORBmatcher.cc:845-852,
 * :884, :925, :934-975): each image holds its left features (d_nLeft1[p] / d_nLeft2[p] of them) followed by the right
 * ones; T4 = [npairs][4][12] host floats: Tll, Tlr, Trl, Trr (T1w * Tw2, T1w * Twr2, Tr1w * Tw2, Tr1w * Twr2), each
 * as rotation matrix (row-major) then translation.  bStereo is false on such rigs (bOnlyStereo rejects everything),
 * there is no epipole-distance gate, and the constraint is KannalaBrandt8::epipolarConstrain (TriangulateMatches >
 * 1e-4) with the cameras / relative pose of the two features' sides.  P->levelSigma2 = mvLevelSigma2. */
int morb_search_for_triangulation_fisheye_batch(morb_matcher* m, const morb_frame_params* P, int npairs, const int* d_img1,
                                                const int* d_img2, const int* d_nLeft1, const int* d_nLeft2, int nimg, int cap,
                                                const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc,
                                                const int* d_node, const uint8_t* d_hasMP, const float* camL8, const float* camR8,
                                                const float* T4, int bOnlyStereo, int bCoarse, int checkOri, int* d_match12,
                                                int* d_nmatches, void* stream);

/* ---- M7: loop-closing / local-mapping searches ---------------------------------------------------------
 * Common inputs: problem f searches keyframe image d_kfImg[f] of the pool; map points are [nprob][mpCap] arrays;
 * d_valid[f][i] != 0 <=> the point passes the reference's state checks at the top of its loop (non-null, !isBad(),
 * !IsInKeyFrame(pKF) / !spAlreadyFound.count(pMP)); d_maxDist / d_minDist = mfMaxDistance / mfMinDistance (the 1.2 /
 * 0.8 factors of Get{Max,Min}DistanceInvariance are applied inside); poses are quaternion xyzw + translation.
 *
 * ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th, bRight) (ORBmatcher.cc:1044-1213) and
 * Fuse(KeyFrame*, Sim3f& Scw, vpPoints, th, vpReplacePoint) (:1215-1321; sim3Form != 0: no reprojection gate, the caller
 * passes Tcw = SE3(Scw.rotationMatrix(), Scw.translation() / Scw.scale()) and Ow = Tcw.inverse().translation()).
 * Output: d_bestIdx[f][i] = feature chosen for map point i (bestDist <= TH_LOW) or -1, d_bestDist likewise (may be NULL).
 * The map-graph bookkeeping that follows a hit (Replace / AddObservation / AddMapPoint, :1196-1208, :1303-1310) depends on
 * live map state and stays with the caller, which replays the reference loop over these per-point results.
 * bRight on a fisheye rig: pass the right camera's pose / centre, cam8 = mpCamera2's KB8 parameters (host pointer, NULL =
 * the pinhole of P), d_jLo / d_jHi = [NLeft, N) per problem (NULL = all features) and d_uRight = NULL.  The Sim3 form on a rig keyframe (no rig
 * branch in the reference: pCamera = pKF->mpCamera, left features): cam8 = mpCamera's parameters, d_jLo / d_jHi = [0, NLeft). */
int morb_fuse_batch(morb_matcher* m, const morb_frame_params* P, int nprob, const int* d_kfImg, int cap, const int* d_count,
                    const morb_keypoint* d_kps, const uint8_t* d_desc, const float* d_uRight, const float* d_Tcw, const float* d_Ow,
                    const float* cam8, const int* d_jLo, const int* d_jHi, int mpCap, const int* d_nMP, const uint8_t* d_valid,
                    const float* d_Pw, const float* d_normal, const float* d_maxDist, const float* d_minDist,
                    const uint8_t* d_mpDesc, float th, int sim3Form, int* d_bestIdx, int* d_bestDist, void* stream);

/* ORBmatcher::SearchByProjection(KeyFrame*, Sim3f& Scw, vpPoints, vpMatched, th, ratioHamming) (:397-494) and its twin
 * with vpPointsKFs / vpMatchedKF (:496-601, manualProjection != 0: u = fx * (X * (1 / Z)) + cx instead of
 * mpCamera->project).  Tcw / Ow from Scw as above.  d_matched[f][idx] != 0 <=> vpMatched[idx] on entry.  Map points are
 * visited in order; a match blocks its feature.  d_matchF[f][idx] = index of the map point newly assigned to feature idx
 * or -1 (the caller writes vpMatched / vpMatchedKF from it); d_nmatches as returned by the reference. */
int morb_search_by_projection_sim3_batch(morb_matcher* m, const morb_frame_params* P, int nprob, const int* d_kfImg, int cap,
                                         const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc, const float* d_Tcw,
                                         const float* d_Ow, int mpCap, const int* d_nMP, const uint8_t* d_valid, const float* d_Pw,
                                         const float* d_normal, const float* d_maxDist, const float* d_minDist,
                                         const uint8_t* d_mpDesc, const uint8_t* d_matched, int th, float ratioHamming,
                                         int manualProjection, int* d_matchF, int* d_nmatches, void* stream);

/* ORBmatcher::SearchBySim3(pKF1, pKF2, vpMatches12, S12, th) (:1323-1519).  Per pair: T1w / T2w = GetPose(); S12 and S21 =
 * S12.inverse() as 7 floats each (RxSO3 quaternion xyzw whose squared norm is the scale, then the translation).  d_valid1[p][i1]
 * != 0 <=> vpMapPoints1[i1] && !vbAlreadyMatched1[i1] && !isBad() (likewise 2); per-feature map-point arrays are [npairs][cap].
 * Outputs: d_vnMatch1 / d_vnMatch2 (the two one-way tables), d_match12[p][i1] = idx2 for mutually consistent pairs else -1,
 * d_nFound. */
int morb_search_by_sim3_batch(morb_matcher* m, const morb_frame_params* P, int npairs, const int* d_kf1Img, const int* d_kf2Img, int cap,
                              const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc, const float* d_T1w,
                              const float* d_T2w, const float* d_S12, const float* d_S21, const uint8_t* d_valid1, const float* d_Pw1,
                              const float* d_maxDist1, const float* d_minDist1, const uint8_t* d_mpDesc1, const uint8_t* d_valid2,
                              const float* d_Pw2, const float* d_maxDist2, const float* d_minDist2, const uint8_t* d_mpDesc2,
                              float th, int* d_vnMatch1, int* d_vnMatch2, int* d_match12, int* d_nFound, void* stream);

/* The loop-closing searches on keyframes of a KannalaBrandt8 rig (KeyFrame::NLeft != -1; the feature row holds mvKeys | mvKeysRight).  The reference has no
 * rig branch in them: pKF->GetFeaturesInArea(u, v, r) defaults to bRight = false and mvKeysUn is the copy of mvKeys, so only the LEFT camera's features
 * [0, NLeft) are candidates (d_nLeft*[p] = NLeft), while every map point of the keyframe (left and right indices) is projected.
 * SearchByProjection(pKF, Scw, ...) (:397-494) projects with pKF->mpCamera->project (:433) = the left KB8 camera: cam8 (HOST, fx fy cx cy k0..k3); its twin
 * with vpPointsKFs (:496-601, manualProjection != 0) and SearchBySim3 (:1323-1519) keep the pinhole formula on pKF->fx, fy, cx, cy (:536-543, :1375-1379),
 * i.e. P's — cam8 is ignored there.  Everything else as in the two entry points above. */
int morb_search_by_projection_sim3_rig_batch(morb_matcher* m, const morb_frame_params* P, int nprob, const int* d_kfImg, int cap,
                                             const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc, const float* d_Tcw,
                                             const float* d_Ow, int mpCap, const int* d_nMP, const uint8_t* d_valid, const float* d_Pw,
                                             const float* d_normal, const float* d_maxDist, const float* d_minDist,
                                             const uint8_t* d_mpDesc, const uint8_t* d_matched, int th, float ratioHamming,
                                             int manualProjection, const float* cam8, const int* d_nLeft, int* d_matchF, int* d_nmatches,
                                             void* stream);
int morb_search_by_sim3_rig_batch(morb_matcher* m, const morb_frame_params* P, int npairs, const int* d_kf1Img, const int* d_kf2Img, int cap,
                                  const int* d_count, const morb_keypoint* d_kps, const uint8_t* d_desc, const float* d_T1w,
                                  const float* d_T2w, const float* d_S12, const float* d_S21, const uint8_t* d_valid1, const float* d_Pw1,
                                  const float* d_maxDist1, const float* d_minDist1, const uint8_t* d_mpDesc1, const uint8_t* d_valid2,
                                  const float* d_Pw2, const float* d_maxDist2, const float* d_minDist2, const uint8_t* d_mpDesc2,
                                  float th, const int* d_nLeft1, const int* d_nLeft2, int* d_vnMatch1, int* d_vnMatch2, int* d_match12,
                                  int* d_nFound, void* stream);


/* ---- Frame-side helpers of the non-rectified / RGB-D input paths ----
 * void Frame::UndistortKeyPoints()  Frame.h, Frame.cc:829-857: cv::undistortPoints(mvKeys, K, mDistCoef, I, mK) for nimg images
 * (DEVICE pointers, image i at offset i*cap; d_count NULL = cap).  dist5 (HOST) = k1 k2 p1 p2 k3 (mDistCoef, k3 = 0 when it has
 * four entries); dist5[0] == 0 copies the records (:830-833).  Only pt.x / pt.y differ between d_kps and d_kpsUn. */
int morb_undistort_keypoints_batch(morb_matcher*, int nimg, int cap, const int* d_count, const morb_keypoint* d_kps, float fx,
                                   float fy, float cx, float cy, const float* dist5, morb_keypoint* d_kpsUn, void* stream);
/* void Frame::ComputeStereoFromRGBD(const cv::Mat& imDepth)  Frame.cc:1049-1067: d_depth = CV_32F depth images (already scaled
 * by mDepthMapFactor), row / image pitches in floats; outputs mvuRight / mvDepth ([nimg][cap], -1 where the depth is <= 0). */
int morb_stereo_from_rgbd_batch(morb_matcher*, int nimg, int cap, const int* d_count, const morb_keypoint* d_kps,
                                const morb_keypoint* d_kpsUn, const float* d_depth, int width, int height, size_t rowPitchFloats,
                                size_t imagePitchFloats, float bf, float* d_uRight, float* d_depthOut, void* stream);
/* void Frame::ComputeImageBounds(const cv::Mat& imLeft)  Frame.cc:859-887 (host): bounds4 = mnMinX, mnMaxX, mnMinY, mnMaxY. */
int morb_image_bounds(int width, int height, float fx, float fy, float cx, float cy, const float* dist5, float* bounds4);

/* ------------------------------------------------------------------------------------------------------
 * Optimizer  (include/Optimizer.h:46-139, src/Optimizer.cc; g2o Levenberg-Marquardt semantics)
 * Poses cross the boundary the way the reference hands them to g2o: unit quaternion (x, y, z, w) followed by
 * the translation, 7 floats (Sophus::SE3f::unit_quaternion() / translation(), Optimizer.cc:781-783, :1046-1048).
 * Pinhole mono (uRight < 0) and rectified-stereo (uRight >= 0) observations; PoseOptimization also for the
 * KannalaBrandt8 fisheye rig ("ToBody" edges), and so has LocalBundleAdjustment (morb_local_bundle_adjustment_fisheye).
 * ---------------------------------------------------------------------------------------------------- */
typedef struct morb_optimizer morb_optimizer;
int morb_optimizer_create(morb_optimizer** out, int device);
void morb_optimizer_destroy(morb_optimizer*);
/* Wait for the work queued on the optimizer's OWN stream (calls made with stream = NULL), not for the device. */
int morb_optimizer_sync(morb_optimizer*);
void* morb_optimizer_stream(const morb_optimizer*);   /* the optimizer's own stream (hipStream_t) */
/* PoseOptimization's deterministic mode.  on (the default since round 5): the sums over the edges (H, b, the robustified chi2) are taken in
 * edge order, one addition after the other, as g2o's sequential loop over its id-sorted active edges does (sparse_optimizer.cpp:482-487):
 * iterations AND trials are then g2o's, decision for decision (tests/test_optimizer_gpu.py).  The ordered sums run on the FP64 matrix core
 * (v_mfma_f64_4x4x4_4b_f64 adds its four products one after the other, each rounded: four edges per instruction) when the device passes the
 * self-test of morb_optimizer_create, on dependent v_add_f64 otherwise — the same bits either way.
 * off: per-thread partial sums and a tree — the reference's poses to ~1e-9, identical outlier flags, inlier counts and outer iterations; near
 * convergence rho = dChi2 / scale is ~0 and its sign follows the last bits of those sums, so the number of LM TRIALS can differ by one
 * (observed: 50 vs 49 in one of nine problems).  ~3 % faster per 256-frame batch (round 4: 1.7 x; the edge-order mode has closed the gap). */
int morb_optimizer_set_exact_order(morb_optimizer*, int on);
/* Which PoseOptimization path this handle runs (any out pointer may be NULL): *exact_order = the mode above; *mfma_chain = 1 when the edge-order sums
 * are carried by v_mfma_f64_4x4x4 (four edges per instruction), 0 when by dependent v_add_f64 — the order being reproduced is g2o's accumulation over
 * its active edges (Thirdparty/g2o/g2o/core/base_unary_edge.hpp:43-72 constructQuadraticForm, called edge by edge from
 * sparse_optimizer.cpp:482-487); *mfma_selftest = what the create-time self-test said on this device: 1 passed, 0 REJECTED the device (the vector chain
 * runs, same bits, slower), -1 not run (MORB_PO2_CHAIN=valu|mfma forced the choice) or could not run.  bench.py prints all three. */
int morb_optimizer_info(const morb_optimizer*, int* mfma_chain, int* exact_order, int* mfma_selftest);

/* static int Optimizer::PoseOptimization(Frame* pFrame)  Optimizer.h:86, Optimizer.cc:762-1051, for nframes
 * frames at once (DEVICE pointers, frame f at offset f*cap): d_count[f] = Frame::N (NULL = cap),
 * d_hasMP[i] != 0 <=> mvpMapPoints[i], d_obs[i] = (mvKeysUn[i].pt.x, .pt.y, mvuRight[i]),
 * d_invSigma2[i] = mvInvLevelSigma2[octave], d_Xw[i] = MapPoint world position; fx..bf = Frame::fx, fy, cx, cy,
 * mbf.  d_pose [nframes][7] in/out (Frame::GetPose / SetPose), d_outlier[i] = mvbOutlier[i],
 * d_nInliers[f] = the return value (0 when fewer than 3 correspondences; the pose is then left untouched).
 * d_stats (optional) [nframes][2] = outer LM iterations, LM trials. */
int morb_pose_optimization_batch(morb_optimizer*, int nframes, int cap, const int* d_count, const uint8_t* d_hasMP,
                                 const float* d_obs, const float* d_invSigma2, const float* d_Xw, float fx, float fy,
                                 float cx, float cy, float bf, float* d_pose, uint8_t* d_outlier, int* d_nInliers,
                                 int* d_stats, void* stream);

/* PoseOptimization for a fisheye stereo rig (pFrame->mpCamera2 != NULL: Optimizer.cc:880-946): features
 * [0, d_nLeft[f]) are left-camera observations (EdgeSE3ProjectXYZOnlyPose on the left KannalaBrandt8 camera), the rest
 * right-camera ones (EdgeSE3ProjectXYZOnlyPoseToBody with mTrl).  d_obs[i] = (x, y, unused).  camL8 / camR8 (HOST) =
 * fx fy cx cy k0 k1 k2 k3; Trl7 (HOST) = Frame::GetRelativePoseTrl() as quaternion xyzw + translation. */
int morb_pose_optimization_fisheye_batch(morb_optimizer*, int nframes, int cap, const int* d_count, const int* d_nLeft,
                                         const uint8_t* d_hasMP, const float* d_obs, const float* d_invSigma2,
                                         const float* d_Xw, const float* camL8, const float* camR8, const float* Trl7,
                                         float* d_pose, uint8_t* d_outlier, int* d_nInliers, int* d_stats, void* stream);

/* ---- visual-inertial tracking and mapping ----
 * IMU::Preintegrated as plain data (include/ImuTypes.h:154-263): 3 x 3 blocks row-major, C = the 15 x 15 covariance
 * row-major, b = the bias the measurements were integrated with in IMU::Bias order (bax bay baz bwx bwy bwz),
 * nga / ngaWalk = the diagonals of IMU::Calib::Cov / CovWalk (gyro x3, acc x3; ImuTypes.cc:375-388). */
typedef struct {
  float dT;
  float dR[9], dV[3], dP[3];
  float JRg[9], JVg[9], JVa[9], JPg[9], JPa[9];
  float C[225];
  float b[6];
  float nga[6], ngaWalk[6];
  float avgA[3], avgW[3];
} morb_imu_preintegrated;

/* IMU::Preintegrated::Initialize + IntegrateNewMeasurement over each measurement sequence (ImuTypes.cc:152-170, :191-247;
 * what Tracking::PreintegrateIMU feeds it, Tracking.cc:1705-1790) for nseq sequences at once.  DEVICE pointers:
 * sequence s owns the measurements [d_start[s], d_start[s+1]) of d_acc / d_gyro ([.][3]) and d_dt; d_bias [nseq][6].
 * ngaDiag6 / walkDiag6 are HOST pointers. */
int morb_imu_preintegrate_batch(morb_optimizer*, int nseq, const int* d_start, const float* d_acc, const float* d_gyro,
                                const float* d_dt, const float* d_bias, const float* ngaDiag6, const float* walkDiag6,
                                morb_imu_preintegrated* d_out, void* stream);

/* static int Optimizer::PoseInertialOptimizationLastKeyFrame(Frame* pFrame, bool bRecInit)  Optimizer.h:127-128,
 * Optimizer.cc:4391-4757, for nframes frames at once (DEVICE pointers, frame f at offset f*cap; same per-feature arrays
 * as morb_pose_optimization_batch plus d_close[i] != 0 <=> mvpMapPoints[i]->mTrackDepth < 10).  States are Rwb (9,
 * row-major), twb, velocity, gyro bias, acc bias = 21 floats: d_kfState = pFrame->mpLastKeyFrame (fixed vertices),
 * d_state in/out = the frame (GetImuRotation / GetImuPosition / GetVelocity / mImuBias -> SetImuPoseVelocity, mImuBias).
 * Tbc12 (HOST) = mImuCalib.mTbc rotation (9) + translation (3); d_pre[f] = pFrame->mpImuPreintegrated.
 * d_nInliers[f] = the return value; d_prior (optional) [nframes][246] doubles = pFrame->mpcpi (ConstraintPoseImu): the
 * 21 state values in FP64 followed by the 15 x 15 H (after the constructor's eigenvalue clamp), row-major.  One pinhole camera; the _fisheye forms below take a KannalaBrandt8 rig. */
int morb_pose_inertial_optimization_last_keyframe_batch(morb_optimizer*, int nframes, int cap, const int* d_count,
                                                        const uint8_t* d_hasMP, const float* d_obs, const float* d_invSigma2,
                                                        const float* d_Xw, const uint8_t* d_close, float fx, float fy, float cx,
                                                        float cy, float bf, const float* Tbc12, const float* d_kfState,
                                                        const morb_imu_preintegrated* d_pre, int bRecInit, float* d_state,
                                                        uint8_t* d_outlier, int* d_nInliers, double* d_prior, void* stream);

/* static int Optimizer::PoseInertialOptimizationLastFrame(Frame* pFrame, bool bRecInit)  Optimizer.h:125-126,
 * Optimizer.cc:4761-5161: the frame and the previous frame (d_prevState, free) are optimised together, tied by
 * EdgeInertial(d_preFrame = pFrame->mpImuPreintegratedFrame), the bias random walks (information from d_preKF =
 * pFrame->mpImuPreintegrated) and EdgePriorPoseImu(d_prevPrior = pFp->mpcpi, [nframes][246] doubles as produced by either
 * function).  d_prior (optional) = the frame's new mpcpi after Optimizer::Marginalize(H, 0, 14).  Other arguments as above. */
int morb_pose_inertial_optimization_last_frame_batch(morb_optimizer*, int nframes, int cap, const int* d_count,
                                                     const uint8_t* d_hasMP, const float* d_obs, const float* d_invSigma2,
                                                     const float* d_Xw, const uint8_t* d_close, float fx, float fy, float cx, float cy,
                                                     float bf, const float* Tbc12, const float* d_prevState,
                                                     const morb_imu_preintegrated* d_preFrame, const morb_imu_preintegrated* d_preKF,
                                                     const double* d_prevPrior, int bRecInit, float* d_state, uint8_t* d_outlier,
                                                     int* d_nInliers, double* d_prior, void* stream);

/* static void Optimizer::LocalInertialBA(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int&, int&, int&, int&, bool bLarge,
 * bool bRecInit)  Optimizer.h:71-74, Optimizer.cc:2324-2897, on the graph the reference assembles at :2337-2768, flattened
 * (HOST pointers).  nKF keyframes with states of 21 floats (Rwb row-major, twb, velocity, gyro bias, acc bias);
 * kfKind[k]: 0 = temporal optimizable keyframe (vpOptimizableKFs), 1 = the fixed keyframe before the window (its IMU
 * state enters the last inertial link), 2 = fixed keyframe that only observes points (lFixedKeyFrames).  nMP points
 * (mpClose[j] != 0 <=> mTrackDepth < 10), nE observations (eObs = x, y, uRight; uRight < 0 = EdgeMono, else EdgeStereo).
 * nI inertial links: iKF1 = mPrevKF, iKF2 = the keyframe owning iPre[i] (mpImuPreintegrated); iRobust[i] != 0 and
 * iInfoScale[i] = 1e-2 on the link to the fixed keyframe (and iRobust on all links when bRecInit) as at :2553-2563; each
 * link also carries EdgeGyroRW / EdgeAccRW.  bLarge selects 4 iterations / lambda 1e-2 instead of 10 / 1.
 * Outputs: kfState21 (optimizable keyframes) and mpPos in place, eraseFlag[e] = 1 where the reference erases the
 * observation (:2773-2826), stats3 = {outer LM iterations, LM trials, ok} with ok = 0 for "FAIL LOCAL-INERTIAL BA"
 * (nothing written back).  One pinhole camera; morb_local_inertial_ba_fisheye for a KB8 rig. */
int morb_local_inertial_ba(morb_optimizer*, int nKF, float* kfState21, const uint8_t* kfKind, int nMP, float* mpPos,
                           const uint8_t* mpClose, int nE, const int* eKF, const int* eMP, const float* eObs, const float* eInvSigma2,
                           int nI, const int* iKF1, const int* iKF2, const morb_imu_preintegrated* iPre, const uint8_t* iRobust,
                           const float* iInfoScale, float fx, float fy, float cx, float cy, float bf, const float* Tbc12, int bLarge,
                           uint8_t* eraseFlag, int* stats3);

/* The three inertial optimisers on a fisheye rig (pFrame->mpCamera2 / pKFi->mpCamera2: Optimizer.cc:4453-4528, :2722-2754;
 * ImuCamPose with two cameras, G2oTypes.cc:96-113).  rig28 (HOST) = left KannalaBrandt8 parameters (8), right ones (8), the
 * rotation (9, row-major) and translation (3) of Frame::GetRelativePoseTrl().  Every observation is monocular: features
 * [0, d_nLeft[f]) / edges with eRight[e] == 0 on the left camera (EdgeMonoOnlyPose(Xw, 0) / EdgeMono(0), mvKeys), the rest on
 * the right camera (cam_idx 1, mvKeysRight); d_obs / eObs = (x, y, unused).  Other arguments as the pinhole forms. */
int morb_pose_inertial_optimization_last_keyframe_fisheye_batch(morb_optimizer*, int nframes, int cap, const int* d_count, const int* d_nLeft,
                                                                const uint8_t* d_hasMP, const float* d_obs, const float* d_invSigma2,
                                                                const float* d_Xw, const uint8_t* d_close, const float* rig28,
                                                                const float* Tbc12, const float* d_kfState,
                                                                const morb_imu_preintegrated* d_pre, int bRecInit, float* d_state,
                                                                uint8_t* d_outlier, int* d_nInliers, double* d_prior, void* stream);
int morb_pose_inertial_optimization_last_frame_fisheye_batch(morb_optimizer*, int nframes, int cap, const int* d_count, const int* d_nLeft,
                                                             const uint8_t* d_hasMP, const float* d_obs, const float* d_invSigma2,
                                                             const float* d_Xw, const uint8_t* d_close, const float* rig28, const float* Tbc12,
                                                             const float* d_prevState, const morb_imu_preintegrated* d_preFrame,
                                                             const morb_imu_preintegrated* d_preKF, const double* d_prevPrior, int bRecInit,
                                                             float* d_state, uint8_t* d_outlier, int* d_nInliers, double* d_prior,
                                                             void* stream);
int morb_local_inertial_ba_fisheye(morb_optimizer*, int nKF, float* kfState21, const uint8_t* kfKind, int nMP, float* mpPos,
                                   const uint8_t* mpClose, int nE, const int* eKF, const int* eMP, const float* eObs, const uint8_t* eRight,
                                   const float* eInvSigma2, int nI, const int* iKF1, const int* iKF2, const morb_imu_preintegrated* iPre,
                                   const uint8_t* iRobust, const float* iInfoScale, const float* rig28, const float* Tbc12, int bLarge,
                                   uint8_t* eraseFlag, int* stats3);

/* static void Optimizer::LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap, int& num_fixedKF,
 * int& num_OptKF, int& num_MPs, int& num_edges)  Optimizer.h:67-69, Optimizer.cc:1053-1441, on the graph the
 * reference assembles at :1058-1351, flattened (HOST pointers): nKF keyframes (local ones first or in any
 * order; kfFixed[i] != 0 for lFixedCameras and the map's initial keyframe), nMP local map points, nE
 * observations (eKF, eMP indices; eObs = (x, y, uRight); eInvSigma2).  lambdaInit100 != 0 <=>
 * pMap->IsInertial() (:1137).  stopFlag = pbStopFlag itself (a bool is one byte; NULL = none): read at entry (the graph is
 * not optimised when set, :1355) and polled at every LM iteration and trial like optimizer.setForceStopFlag does (:1142).
 * Outputs: optimised kfPose (free keyframes) and mpPos in place, eraseFlag[e] = 1 where the reference erases the
 * observation (:1366-1401), stats2 = {outer LM iterations, LM trials}. */
int morb_local_bundle_adjustment(morb_optimizer*, int nKF, float* kfPose, const uint8_t* kfFixed, int nMP, float* mpPos,
                                 int nE, const int* eKF, const int* eMP, const float* eObs, const float* eInvSigma2,
                                 float fx, float fy, float cx, float cy, float bf, int lambdaInit100,
                                 const unsigned char* stopFlag, uint8_t* eraseFlag, int* stats2);

/* The same in three steps, so that a problem can stay resident in HBM and be solved repeatedly (benchmarks) or
 * aborted from another thread: create (upload + CSR build), solve (device only, asynchronous on `stream`,
 * restarts from the uploaded initial values), results (synchronises, downloads).  morb_ba_set_stop mirrors
 * LocalMapping::InterruptBA -> mbAbortBA (LocalMapping.cc:884): the flag lives in pinned host memory mapped into the
 * device, so setting it makes no HIP call, may come from any thread while a solve runs, and is seen at the top of the
 * next outer iteration / LM trial like in g2o (sparse_optimizer.cpp:376, optimization_algorithm_levenberg.cpp:149). */
typedef struct morb_ba_problem morb_ba_problem;
int morb_ba_problem_create(morb_optimizer*, morb_ba_problem** out, int nKF, const float* kfPose, const uint8_t* kfFixed,
                           int nMP, const float* mpPos, int nE, const int* eKF, const int* eMP, const float* eObs,
                           const float* eInvSigma2, float fx, float fy, float cx, float cy, float bf,
                           int lambdaInit100);

/* The same for a KannalaBrandt8 stereo rig (pKFi->mpCamera2 != NULL, Optimizer.cc:1244-1351): every observation is an
 * EdgeSE3ProjectXYZ with the left camera (eRight[e] == 0; mvuRight < 0 on such rigs, so there are no stereo edges) or
 * an EdgeSE3ProjectXYZToBody with the right camera behind mTrl (eRight[e] != 0, observation = mvKeysRight[rightIndex
 * - NLeft].pt).  eObs2 = [nE][2]; camL8 / camR8 = fx fy cx cy k0..k3; Trl7 = GetRelativePoseTrl() as quaternion xyzw +
 * translation.  Erase rule: chi2 > 5.991 || !isDepthPositive() for both kinds (:1366-1390). */
int morb_ba_problem_create_fisheye(morb_optimizer* o, morb_ba_problem** out, int nKF, const float* kfPose, const uint8_t* kfFixed,
                                   int nMP, const float* mpPos, int nE, const int* eKF, const int* eMP, const float* eObs2,
                                   const uint8_t* eRight, const float* eInvSigma2, const float* camL8, const float* camR8,
                                   const float* Trl7, int lambdaInit100);
int morb_local_bundle_adjustment_fisheye(morb_optimizer* o, int nKF, float* kfPose, const uint8_t* kfFixed, int nMP, float* mpPos,
                                         int nE, const int* eKF, const int* eMP, const float* eObs2, const uint8_t* eRight,
                                         const float* eInvSigma2, const float* camL8, const float* camR8, const float* Trl7,
                                         int lambdaInit100, const unsigned char* stopFlag, uint8_t* eraseFlag, int* stats2);
void morb_ba_problem_destroy(morb_ba_problem*);
int morb_ba_set_stop(morb_ba_problem*, int stop);
/* mode 0 (default): one launch per LM phase over the whole GPU, accept / reject decided on the device; mode 1: the whole LM loop
 * inside ONE persistent workgroup (for many small problems). */
int morb_ba_set_mode(morb_ba_problem*, int mode);
/* Mode 0 BLOCKS the calling host thread until the last LM decision: it queues trial after trial on `stream`, one trial ahead of the
 * decisions, and watches two words in mapped host memory (forwarding the caller's stop flag meanwhile); MORB_ERR_HIP if the stream
 * reports a fault while it waits.  The kernels queued behind the last decision are empty; results are read with morb_ba_results
 * (which synchronises).  Mode 1 only enqueues one launch. */
int morb_ba_solve(morb_ba_problem*, void* stream);
int morb_ba_results(morb_ba_problem*, float* kfPose, float* mpPos, uint8_t* eraseFlag, int* stats2);
/* Measurement hook (bench.py's LocalBA roofline entry; not a reference method): times the FP64-MFMA Schur product of block_solver.hpp:
 * 354-480 on the operands the last morb_ba_solve left behind.  flops = MFMA flops issued per launch, usefulFlops = the sparse form's. */
int morb_ba_schur_profile(morb_ba_problem*, int iters, float* msPerLaunch, double* flops, double* usefulFlops);

#ifdef __cplusplus
}
#endif
#endif /* MORB_HIP_H */
