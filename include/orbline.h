/* orbline.h -- C ABI of liborbline_hip.so: the MI355X (gfx950) implementation of the per-frame
 * feature path of ORB_Line_SLAM.  Plain pointers and sizes only (no C++/torch types); every entry
 * point names the reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - an olf_ctx serves one image size, one parameter block and up to max_images images per call;
 *     a stereo pair is two images: image index = 2*pair + side (0 = left, 1 = right);
 *   - *_dev entry points take DEVICE pointers and enqueue on `stream` (a hipStream_t, NULL = the
 *     context's own stream) without synchronising; the others take HOST pointers, copy and block;
 *   - per-image outputs are fixed-stride records: image i's key points start at
 *     kps[i * olf_orb_capacity(ctx)], descriptors at desc[i * capacity * 32]; counts[i] says how
 *     many are valid;
 *   - return value: OLF_OK or a negative OLF_ERR_* (orbline_types.h); olf_last_error() gives text.
 *   - a context is not re-entrant; use one per host thread (the reference runs one extractor
 *     object per std::thread, src/Frame.cc:164-171).
 */
#ifndef ORBLINE_H
#define ORBLINE_H

#include "orbline_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct olf_ctx olf_ctx;

/* library / device */
const char* olf_last_error(void);
int olf_device_count(void);                 /* number of visible HIP devices (0 on a CPU-only box) */
/* parameters of Examples/PL/PL_KITTI00-02.yaml:42-55,95-128 + src/Config.cpp:26-160 defaults */
int olf_default_params(olf_params* p);

/* context: replaces constructing ORBextractor x2 + Lineextractor x2 (src/Tracking.cc:131-142) */
int  olf_ctx_create(const olf_params* p, int width, int height, int max_images, olf_ctx** out);
void olf_ctx_destroy(olf_ctx* ctx);
int  olf_ctx_synchronize(olf_ctx* ctx);

/* ---- ORBextractor (include/ORBextractor.h:52-118, src/ORBextractor.cc) -------------------- */
/* GetLevels / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (include/ORBextractor.h:68-91) + mnFeaturesPerLevel; arrays of nlevels */
int olf_orb_scale_tables(const olf_ctx* ctx, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                         int32_t* features_per_level);
/* level sizes of mvImagePyramid (src/ORBextractor.cc:1113-1114) */
int olf_orb_level_sizes(const olf_ctx* ctx, int32_t* widths, int32_t* heights);
/* per-image capacity of the key point / descriptor records */
int olf_orb_capacity(const olf_ctx* ctx);
/* ORBextractor::operator()(image, mask [ignored], keypoints, descriptors), src/ORBextractor.cc:1045-1107,
 * for n_images images of width x height, row stride = width. */
int olf_orb_extract_dev(olf_ctx* ctx, const uint8_t* d_images, int n_images, olf_keypoint* d_kps, uint8_t* d_desc,
                        int32_t* d_counts, void* stream);
int olf_orb_extract(olf_ctx* ctx, const uint8_t* images, int n_images, olf_keypoint* kps, uint8_t* desc, int32_t* counts);
/* read back mvImagePyramid[level] of image `image` of the last extract call (public member of the
 * reference class, read by Frame::ComputeStereoMatches src/Frame.cc:799-816).  blurred != 0 returns
 * the GaussianBlur'ed working image of src/ORBextractor.cc:1087-1088 instead.  dst: w*h bytes. */
int olf_orb_pyramid_level(olf_ctx* ctx, int image, int level, int blurred, uint8_t* dst);
/* debug/test: per-level FAST candidates handed to DistributeOctTree (vToDistributeKeys,
 * src/ORBextractor.cc:821-827) as int32 triples (x,y,score) relative to minBorder. */
int olf_orb_debug_candidates(olf_ctx* ctx, int image, int level, int32_t* xys, int cap, int32_t* count);

#ifdef __cplusplus
}
#endif
#endif
