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
 * d_count [nimg], d_mono [nimg] (monoIndex).  Asynchronous on `stream`. */
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

#ifdef __cplusplus
}
#endif
#endif /* MORB_HIP_H */
