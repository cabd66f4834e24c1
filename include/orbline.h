/* orbline.h -- C ABI of liborbline_hip.so: the MI355X (gfx950) implementation of the per-frame
 * feature path of ORB_Line_SLAM.  Plain pointers and sizes only (no C++/torch types); every entry
 * point names the reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - an olf_ctx serves one image size, one parameter block and up to max_images images per call;
 *     a stereo pair is two images: image index = 2*pair + side (0 = left, 1 = right);
 *   - *_dev entry points take DEVICE pointers and enqueue on `stream` (a hipStream_t, NULL = the
 *     context's own stream, a non-blocking one) without synchronising; the others take HOST pointers, copy and
 *     block.  The legacy default stream cannot be named -- its handle IS NULL -- and is not ordered with the
 *     context's streams: a caller that works on it (torch.cuda.current_stream() outside a stream context)
 *     has to synchronise, or work on a stream of its own and pass that;
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
/* A context belongs to the device that is current when it is created (calls made with another device current return OLF_ERR_INVALID) and
 * to ONE host thread.  Its scratch buffers are shared by every call, so at most one stream may have work of a context in flight at a time:
 * the *_dev entry points accept any stream, but work enqueued on a second stream must be ordered after the first (event / synchronise) by
 * the caller. */
int  olf_ctx_create(const olf_params* p, int width, int height, int max_images, olf_ctx** out);
void olf_ctx_destroy(olf_ctx* ctx);
/* Capacity overflows (more corners / key points / segments than a fixed-size device buffer holds) are never silent: the blocking
 * host-pointer entry points return OLF_ERR_CAPACITY themselves; for the asynchronous *_dev entry points the flag is collected by
 * olf_ctx_synchronize (waits for the context's two streams, then reports and clears it) or, without waiting for anything,
 * by olf_ctx_poll_status (call it once the caller's own stream has passed the work in question).  The reference has no such limits
 * (std::vector growth); a refused frame is the equivalent of its std::bad_alloc. */
int  olf_ctx_synchronize(olf_ctx* ctx);
/* Batch pipelining for olf_stereo_frames_dev.  By default the line stream of a call forks from `stream` at the call (the images may have been
 * produced on it), i.e. behind the ORB / stereo / matching tail of the PREVIOUS call on that stream.  With an input event (a hipEvent_t the
 * caller records on whichever stream produces d_images, before the call; NULL = default) the line stream waits for that event only, and the
 * LSD front of batch k + 1 runs beside the tail of batch k.  The context's buffers allow it: the LSD front writes line-path scratch only, and
 * everything of batch k + 1 that touches the output buffers or the shared LBD planes is ordered behind batch k's work on `stream` (api.cpp,
 * olf_stereo_frames_dev).  The caller still orders `stream` itself behind the production of d_images.  The event is ONE-SHOT: the next
 * olf_stereo_frames_dev consumes it (set it again before every call that should use one); the host entry olf_stereo_frames ignores and clears it. */
int  olf_ctx_set_input_event(olf_ctx* ctx, void* hip_event);
/* Deferred join for olf_stereo_frames_dev.  By default the call ends with `stream` waiting for the line stream: every output is complete on `stream`.  The
 * reference's tracker consumes the point features first (TrackReferenceKeyFrame: ComputeBoW + SearchByBoW, src/Tracking.cc:963-970) and the line features after
 * them (:1296-1308); with the deferred join on, the call returns with the ORB-side outputs (key points, descriptors, counts, mvuRight, mvDepth) complete on
 * `stream` and the line-side outputs (key lines, LBD descriptors, line matches) still being produced on the context's line stream --
 * olf_stereo_frames_join_dev(ctx, stream) makes `stream` wait for them (the next olf_stereo_frames_dev call does it itself if the caller did not).
 * A join on the frame call's own stream discharges the obligation for the context; a join on any OTHER stream (explicit, or the implicit one of
 * olf_match_bf_dev / olf_stereo_lines_dev / olf_frames_pack_dev when they are handed a line-side output of the pending call) orders that stream only,
 * and later entries on the frame call's stream still wait.  A kernel of the caller's own that reads line-side outputs must be ordered by the caller
 * (olf_stereo_frames_join_dev on its stream). */
int  olf_ctx_set_deferred_join(olf_ctx* ctx, int on);
int  olf_stereo_frames_join_dev(olf_ctx* ctx, void* stream);
int  olf_ctx_poll_status(olf_ctx* ctx);

/* Stage timing with HIP events recorded on the stream each stage is launched on (the reference's only
 * instrumentation is std::chrono around TrackStereo, Examples/PL/PL_stereo_kitti.cc:80-97).  Accumulates
 * per-stage total milliseconds and call counts while enabled; read/reset synchronise the device. */
int olf_profile_enable(olf_ctx* ctx, int on);
int olf_profile_reset(olf_ctx* ctx);
int olf_profile_stage_count(void);
const char* olf_profile_stage_name(int stage);
int olf_profile_read(olf_ctx* ctx, double* total_ms, int32_t* calls);   /* arrays of olf_profile_stage_count() */

/* ---- input conditioning ahead of the path ------------------------------------------------- */
enum { OLF_RGB2GRAY = 0, OLF_BGR2GRAY = 1, OLF_RGBA2GRAY = 2, OLF_BGRA2GRAY = 3 };
/* cv::cvtColor(img, gray, CV_*2GRAY) of Tracking::GrabImageStereo (src/Tracking.cc:193-218): n_images interleaved 8-bit
 * images (3 or 4 channels) of the context's size -> n_images gray images.  Device pointers. */
int olf_cvt_gray_dev(olf_ctx* ctx, const uint8_t* d_src, int code, int n_images, uint8_t* d_gray, void* stream);
/* cv::remap(src, dst, M1, M2, INTER_LINEAR) with two CV_32FC1 maps and BORDER_CONSTANT 0, the EuRoC rectification of
 * Examples/PL/PL_stereo_euroc.cc:136-137.  src: n_images images src_w x src_h (stride src_w); maps: dst_w*dst_h floats each
 * (shared by all images, e.g. one call per camera); dst: n_images images dst_w x dst_h.  Device pointers. */
int olf_remap_linear_dev(olf_ctx* ctx, const uint8_t* d_src, int src_w, int src_h, const float* d_mapx, const float* d_mapy, int dst_w,
                         int dst_h, int n_images, uint8_t* d_dst, void* stream);
/* cv::initUndistortRectifyMap(K, D, R, P(3x3), Size(w, h), CV_32F, M1, M2), Examples/PL/PL_stereo_euroc.cc:97-98: the two CV_32FC1 maps
 * olf_remap_linear takes.  K, R, P: 3 x 3 row-major doubles (P = the left 3 x 3 block of the projection matrix); D: n_dist <= 8 distortion
 * coefficients in OpenCV's order k1 k2 p1 p2 k3 k4 k5 k6.  Double arithmetic in the order of OpenCV 3.4's generic code (convention C.13). */
int olf_init_undistort_rectify_map_dev(olf_ctx* ctx, const double* K, const double* D, int n_dist, const double* R, const double* P, int w, int h,
                                       float* d_map1, float* d_map2, void* stream);
int olf_init_undistort_rectify_map(olf_ctx* ctx, const double* K, const double* D, int n_dist, const double* R, const double* P, int w, int h,
                                   float* map1, float* map2);
/* host-buffer forms (copy, run, copy back, block) */
int olf_cvt_gray(olf_ctx* ctx, const uint8_t* src, int code, int n_images, uint8_t* gray);
int olf_remap_linear(olf_ctx* ctx, const uint8_t* src, int src_w, int src_h, const float* mapx, const float* mapy, int dst_w, int dst_h,
                     int n_images, uint8_t* dst);

/* ---- MapPoint / MapLine::ComputeDistinctiveDescriptors (src/MapPoint.cc:254-318, src/MapLine.cc:257-322) --
 * n_points landmarks; landmark p is observed by descriptors desc[offs[p] .. offs[p+1]) (32 bytes each, the rows the reference gathers
 * from the non-bad key frames, in its std::map iteration order).  best[p] = index inside that list of the descriptor with the least
 * median distance to the others (first minimum), -1 for an empty list.  Host buffers; at most 1024 observations per landmark. */
int olf_distinctive_descriptors(olf_ctx* ctx, const uint8_t* desc, const int32_t* offs, int n_points, int32_t* best);

/* ---- key-frame feature record of the binary map file -----------------------------------------
 * void Map::SaveKeyFrame(ofstream &f, KeyFrame* kf), src/Map.cc:283-373 / KeyFrame* Map::LoadKeyFrame(ifstream &f, ...), :376-531,
 * for the members the feature path produces: byte-exact what f.write((char*)&member, sizeof(member)) writes member by member.
 * Host code (no device needed), like the reference's.  *_ids: NULL = no MapPoint / MapLine anywhere (ULONG_MAX is written). */
size_t olf_kf_record_bytes(int n_keys, int n_lines);
int olf_kf_record_pack(uint64_t frame_id, uint64_t kf_id, double timestamp, const float* t3, const float* quat4, int n_keys,
                       const olf_keypoint* keys, const float* uright, const float* depth, const uint8_t* desc, const uint64_t* mappoint_ids,
                       int n_lines, const olf_keyline* lines, const float* disparity2, const double* le3, const uint8_t* ldesc,
                       const uint64_t* mapline_ids, uint8_t* out, size_t capacity, size_t* written);
int olf_kf_record_counts(const uint8_t* buf, size_t len, int32_t* n_keys, int32_t* n_lines, size_t* record_bytes);
int olf_kf_record_unpack(const uint8_t* buf, size_t len, uint64_t* frame_id, uint64_t* kf_id, double* timestamp, float* t3, float* quat4,
                         olf_keypoint* keys, float* uright, float* depth, uint8_t* desc, uint64_t* mappoint_ids, olf_keyline* lines,
                         float* disparity2, double* le3, uint8_t* ldesc, uint64_t* mapline_ids);

/* ---- BoW transform: ORBVocabulary / LineVocabulary (include/ORBVocabulary.h:30-34) ----------
 * = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>; transform() is called by Frame::ComputeBoW (src/Frame.cc:585-597) and
 * KeyFrame::ComputeBoW (src/KeyFrame.cc:96-112) with levelsup = 4. */
typedef struct olf_voc olf_voc;
/* scoring: 0 L1_NORM 1 L2_NORM 2 CHI_SQUARE 3 KL 4 BHATTACHARYYA 5 DOT_PRODUCT; weighting: 0 TF_IDF 1 TF 2 IDF 3 BINARY
 * (Thirdparty/DBoW2/DBoW2/BowVector.h:36-53).  Nodes in id order (node 0 = root, entry ignored): parent[i] < i, is_leaf[i] marks a
 * word (word ids are assigned in node order), desc 32 bytes per node, weight per node.  The tree is uploaded to the current device. */
int olf_voc_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t* parent, const uint8_t* is_leaf, const uint8_t* desc,
                   const double* weight, olf_voc** out);
/* TemplatedVocabulary::loadFromTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1425): "k L scoring weighting" then one
 * "parent isLeaf d0..d31 weight" line per node. */
int olf_voc_load_text(const char* path, olf_voc** out);
void olf_voc_destroy(olf_voc* voc);
int olf_voc_info(const olf_voc* voc, int* k, int* L, int* scoring, int* weighting, int* n_nodes, int* n_words);
/* transform(feature, word_id, weight, nid, levelsup) (:1217-1261) for n descriptors; device pointers. */
int olf_bow_words_dev(olf_ctx* ctx, const olf_voc* voc, const uint8_t* d_desc, int n, int levelsup, int32_t* d_word, double* d_weight,
                      int32_t* d_node, void* stream);
/* host: per-feature (word, weight, node) -> BowVector (ascending word id) + FeatureVector (CSR over ascending node id), with the
 * vocabulary's weighting / normalisation (:1127-1195, BowVector.cpp:34-84, FeatureVector.cpp:31-45).  Capacities: n (fv_offs n+1). */
int olf_bow_assemble(const olf_voc* voc, const int32_t* word, const double* weight, const int32_t* node, int n, int32_t* bow_ids,
                     double* bow_vals, int* n_bow, int32_t* fv_nodes, int32_t* fv_offs, int32_t* fv_idx, int* n_fv);
/* transform(features, v, fv, levelsup) for one image's descriptors (host buffers): descent on the GPU, assembly on the host. */
int olf_bow_transform(olf_ctx* ctx, const olf_voc* voc, const uint8_t* desc, int n, int levelsup, int32_t* bow_ids, double* bow_vals,
                      int* n_bow, int32_t* fv_nodes, int32_t* fv_offs, int32_t* fv_idx, int* n_fv);
/* Batched, device-resident ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (src/ORBmatcher.cc:161-290,
 * Tracking::TrackReferenceKeyFrame, src/Tracking.cc:963-970) over consecutive frames, Frame::ComputeBoW (src/Frame.cc:585-597) of every frame
 * included -- no host step between a frame's descriptors and its matches.  Frame j = image j * img_stride of the key point / descriptor / count
 * buffers of olf_orb_extract_dev (stride olf_orb_capacity(); img_stride 2 = the left images of a stereo batch); pair j: pKF = frame j, F = frame
 * j + 1, j < n_frames - 1.  d_mp_valid / d_mp_bad [n_frames][capacity] bytes: vpMapPointsKF[i] != NULL / isBad() of a frame in its key-frame
 * role (NULL: every feature holds a good map point).  levelsup: 4 in the reference.  d_matches [n_frames - 1][capacity]: per feature of F the
 * index of the pKF feature whose map point it received (-1 = NULL); d_nmatches [n_frames - 1] = the reference's return values.  The greedy
 * state of the reference (a feature of F that holds a match is skipped, :214) is kept: inside a vocabulary node the key frame's features are
 * walked in order by one wave; different nodes share no feature and run side by side. */
int olf_search_by_bow_batch_dev(olf_ctx* ctx, const olf_voc* voc, int n_frames, int img_stride, const olf_keypoint* d_kps, const uint8_t* d_desc,
                                const int32_t* d_counts, const uint8_t* d_mp_valid, const uint8_t* d_mp_bad, float nnratio, int check_orientation,
                                int levelsup, int32_t* d_matches, int32_t* d_nmatches, void* stream);

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
/* one image whose rows are row_stride bytes apart (a cv::Mat ROI: data, step) -- no host-side repacking */
int olf_orb_extract_strided(olf_ctx* ctx, const uint8_t* image, size_t row_stride, olf_keypoint* kps, uint8_t* desc, int32_t* count);
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
/* debug: the context's 64-int device status block (overflow flags in [0]; instrumented builds put cycle counters at [16..31]). */
int olf_debug_status(olf_ctx* ctx, int32_t* out64);
int olf_debug_status_n(olf_ctx* ctx, int32_t* out, int n);
/* debug: the regions the last LSD growth logged for `image`, in detection order: (list start, pixel count) pairs and final region angles */
int olf_debug_lsd_owner(olf_ctx* ctx, int image, uint32_t* out);      /* debug: owner words of the last growth, Ws*Hs */
int olf_debug_lsd_regions(olf_ctx* ctx, int image, int32_t* start_n, double* angle, int cap, int32_t* count);     /* the first n (<= 256) words */
/* debug/test: waves per image of the LSD region-growing kernel (1..16; 0 = the one-wave sequential agent; -1 = automatic from the batch
 * size) and entries of its reorder buffer (128, 256, 512, or 1024 with several workgroups per image; 0 = automatic).  Results do not depend on either. */
int olf_debug_lsd_waves(olf_ctx* ctx, int waves_per_image, int rob_entries);
/* debug/test: workgroups (CUs) that grow ONE image together when the kernel runs 16 waves per image (1, 2 or 4; 0 = automatic from the batch
 * size: calls of up to 64 images -- the one-pair-per-call shape of Frame::Frame, src/Frame.cc:164-171 -- take 2).  Results do not depend on it. */
int olf_debug_lsd_groups(olf_ctx* ctx, int groups);
/* debug/test: where the groups of an image run.  0 (default): on workgroups 8 apart, which the hardware's round-robin placement puts on ONE XCD (they
 * meet in its L2); 1: on consecutive workgroups, i.e. on DIFFERENT XCDs.  What crosses between groups is agent-scope traffic either way: results do
 * not depend on it (tests/test_lsd_grow_gpu.py::test_growth_groups_scattered_over_xcds), only the time does. */
int olf_debug_lsd_scatter(olf_ctx* ctx, int on);
/* debug/test: cap the primary pixel log of the one-wave region growing at `entries` pixels per image (0 = the context's own capacity: every pixel in
 * contexts of up to 2048 images, half of the pixels in larger -- batch -- contexts).  Images that log more are grown again on a block of the context's spill
 * arena inside the same call; when the arena is exhausted the call reports OLF_ERR_CAPACITY.  Results do not depend on it. */
int olf_debug_lsd_log_cap(olf_ctx* ctx, int entries);
/* debug / tests: the kernel that replays libstdc++'s std::sort for the LSD seed order (convention C.9 variant 1, csrc/lsd_seedsort.hip) on a
 * caller-supplied array of n <= Ws*Hs keys, (field << 22) | payload with a 10-bit field: out receives the keys whose field is <= kthr in the
 * order std::sort(keys, keys + n, field ascending) leaves them; depth_limit < 0 = introsort's own 2 * floor(log2 n), a small value forces
 * its heap-sort branch. */
int olf_debug_seed_sort(olf_ctx* ctx, const uint32_t* keys, int n, int kthr, int depth_limit, uint32_t* out, int32_t* out_n);
/* debug / tests: the kernel variant of that replay -- 0: one wave per image (batches), 1 / 2: 4 / 8 cooperating waves per image (few images: the
 * drop-in's one-pair-per-call shape), 3 / 4: groups of 4 / 8 images per workgroup whose waves take over each other's ranges (large batches), 5: 2 waves per image,
 * -1: chosen from the batch size.  Results do not depend on it. */
int olf_debug_seed_sort_mode(olf_ctx* ctx, int mode);
/* debug / tests: the seed-order kernel of the capacity path (csrc/lsd_wide.hip: lsd_n_bins > 1024 or an LSD working image of 2^22 pixels and more -- free YAML
 * keys of the reference, src/Config.cpp:268,274) on a caller-supplied array of 64-bit keys (field << 32 | payload).  full = 0: the order libstdc++'s
 * std::sort(begin, end, field ascending) leaves (convention C.9 variant 1, OpenCV lsd.cpp ll_angle); full = 1: ascending whole words (variant 0).  Keys whose
 * field is <= kthr are listed: out receives their payloads in order, *out_n their number.  depth_limit: introsort's depth limit (-1: 2 * floor(log2 n)).  The
 * context must be a wide one. */
int olf_debug_seed_sort_wide(olf_ctx* ctx, const uint64_t* keys, int n, int64_t kthr, int depth_limit, int full, uint32_t* out, int32_t* out_n);
/* debug / tests: cap the 32-pixel chunk pool the multi-wave growth may use per image (0: all of it).  An image that exhausts the pool is grown
 * again by the one-wave agent inside the same call -- the result does not change, only the time. */
int olf_debug_lsd_pool(olf_ctx* ctx, int pool_chunks);
/* debug/test: the LSD agent's unscaled exact float division against IEEE division on blocks*256*per_thread pseudo-random operand pairs
 * from its operand range; *mismatches = number of quotients that differ in any bit (must be 0). */
int olf_debug_fdiv_sweep(olf_ctx* ctx, uint64_t seed, int blocks, int per_thread, uint64_t* mismatches);
/* debug/test: the lean sqrt(n / 4.0) of the LSD key kernel (ll_angle's gradient norm; lsd_device.hpp sqrt_quarter) against the compiler's IEEE
 * sqrt on every integer n in [0, count); *mismatches = number of results that differ in any bit (must be 0). */
int olf_debug_sqrtq_sweep(olf_ctx* ctx, int count, uint64_t* mismatches);
/* debug/test: the growth agent's cheap alignment test (dot / cross products of the region's float sums with a candidate's tabulated direction, decided under a
 * margin that bounds cv::fastAtan2's error -- csrc/lsd.hip, PF bit 16) against the reference's expression |fastAtan2(sums) * DEG2RAD - angle| <= prec
 * (OpenCV lsd.cpp isAligned as called by region_grow; LSDDetector_custom.cpp:246,262) on blocks*256*per_thread pseudo-random (sums, candidate) pairs, three
 * quarters of them within 3 mrad of the tolerance: out3[0] = certain decisions that contradict the reference (must be 0), out3[1] = decisions left to the
 * reference's expression, out3[2] = all. */
int olf_debug_align_sweep(olf_ctx* ctx, uint64_t seed, int blocks, int per_thread, uint64_t* out3);

/* ---- Frame::ComputeStereoMatches (src/Frame.cc:702-876) ------------------------------------- */
/* Stereo point matching for n_pairs pairs whose ORB features (images 2p = left, 2p+1 = right) came from
 * the immediately preceding olf_orb_extract*_dev call on this context (its device-resident
 * mvImagePyramid of both extractors is read for the 11x11 SAD refinement, src/Frame.cc:799-816).
 * Outputs per pair, stride olf_orb_capacity(): mvuRight / mvDepth (-1 = no match), src/Frame.cc:704-705. */
int olf_stereo_points_dev(olf_ctx* ctx, int n_pairs, const olf_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_counts,
                          float* d_uright, float* d_depth, void* stream);
/* host convenience: ExtractORB x2 + ComputeStereoMatches for n_pairs pairs (images: 2*n_pairs) */
int olf_stereo_points(olf_ctx* ctx, const uint8_t* images, int n_pairs, olf_keypoint* kps, uint8_t* desc, int32_t* counts,
                      float* uright, float* depth);

/* ---- LineMatcher / ORBmatcher brute force (src/LineMatcher.cpp:42-62,104-132; App. A.10) ------- */
/* match(desc1, desc2, nnr, matches_12) for n_sets independent sets: set s has d_nA[s*a_step] rows at
 * d_descA + s*strideA*32 and d_nB[s*b_step] rows at d_descB + s*strideB*32.  kNN(2) both ways, ratio
 * test d0 < d1*nnr, mutual check when best_lr (Config::bestLRMatches()).  d_m12: n_sets*strideA ints,
 * -1 = no match. */
int olf_match_bf_dev(olf_ctx* ctx, const uint8_t* d_descA, const int32_t* d_nA, int strideA, int a_step, const uint8_t* d_descB,
                     const int32_t* d_nB, int strideB, int b_step, int n_sets, float nnr, int best_lr, int32_t* d_m12, void* stream);
int olf_match_bf(olf_ctx* ctx, const uint8_t* descA, int nA, const uint8_t* descB, int nB, float nnr, int best_lr, int32_t* m12);
/* cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) (host buffers): per query best index, best and second distance
 * (-1 / INT_MAX where the train set is too small) */
int olf_knn2(olf_ctx* ctx, const uint8_t* descQ, int nQ, const uint8_t* descT, int nT, int32_t* idx0, int32_t* dist0, int32_t* dist1);
/* Candidate-list distances for ORBmatcher::SearchByProjection (src/ORBmatcher.cc:1330-1472) / SearchByBoW (:161-290):
 * query i is compared with train rows cand_idx[cand_offsets[i] .. cand_offsets[i+1]) (CSR, built by the caller from
 * Frame::GetFeaturesInArea / the BoW feature vectors); dist[k] receives the Hamming distance of pair k (0xffff for an
 * out-of-range index).  The order-dependent greedy resolution stays with the caller. */
int olf_match_candidates_dev(olf_ctx* ctx, const uint8_t* d_descQ, int nQ, const uint8_t* d_descT, int nT, const int32_t* d_cand_offsets,
                             const int32_t* d_cand_idx, uint16_t* d_dist, void* stream);
int olf_match_candidates(olf_ctx* ctx, const uint8_t* descQ, int nQ, const uint8_t* descT, int nT, const int32_t* cand_offsets,
                         const int32_t* cand_idx, uint16_t* dist);
/* ---- the per-frame ORBmatcher searches, complete (host candidate generation + GPU distances + the reference's resolution) ----
 * Plain view of the Frame / KeyFrame members these searches read (include/Frame.h:49-260, include/KeyFrame.h).  Pointers a search
 * does not read may be NULL.  mp_valid / mp_obs stand for `mvpMapPoints[i] != NULL` and `mvpMapPoints[i]->Observations() > 0`; the
 * searches that assign map points to the current frame update them in place, as the reference updates mvpMapPoints.
 * Constness: the entry points take `const olf_frame_view*` -- the VIEW (its pointers and counts) is never modified; mp_valid and mp_obs are
 * deliberately pointers to non-const bytes, because for the current frame they are outputs (each function's comment names what it updates).
 * Every other array is read-only. */
typedef struct olf_frame_view {
    const olf_keypoint* keys;     /* mvKeysUn (= mvKeys for a rectified camera, src/Frame.cc:601-605)            */
    const uint8_t* desc;          /* mDescriptors [n][32]                                                        */
    const float*   uright;        /* mvuRight [n] (negative = monocular point)                                   */
    int32_t        n;             /* N                                                                           */
    uint8_t*       mp_valid;      /* [n]                                                                         */
    uint8_t*       mp_obs;        /* [n]                                                                         */
    const uint8_t* mp_bad;        /* [n] pMP->isBad()                                                            */
    const float*   mp_world;      /* [n][3] pMP->GetWorldPos()                                                   */
    const uint8_t* mp_desc;       /* [n][32] pMP->GetDescriptor()                                                */
    const uint8_t* outlier;       /* [n] mvbOutlier                                                              */
    const float*   Tcw;           /* mTcw, 4x4 row-major                                                         */
    float fx, fy, cx, cy, mbf, minX, maxX, minY, maxY;     /* calibration, mnMinX .. mnMaxY                      */
    const float*   scale_factors; /* mvScaleFactors [n_levels]                                                   */
    int32_t        n_levels;
    const float*   mp_maxd;       /* [n] pMP->mfMaxDistance (key-frame searches that predict a scale level)      */
    const float*   mp_mind;       /* [n] pMP->mfMinDistance                                                      */
    const int32_t* fv_nodes;      /* mFeatVec (DBoW2::FeatureVector) as CSR: ascending node ids [fv_n],          */
    const int32_t* fv_offsets;    /*   offsets [fv_n + 1],                                                       */
    const int32_t* fv_features;   /*   feature indices                                                           */
    int32_t        fv_n;
} olf_frame_view;
/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono), src/ORBmatcher.cc:1330-1472
 * (Tracking::TrackWithMotionModel, every frame).  matches[i2] = index of the LastFrame feature whose map point CurrentFrame feature i2
 * received (-1 = none); cur->mp_valid / mp_obs are updated; *nmatches = the reference's return value. */
int olf_search_by_projection(olf_ctx* ctx, const olf_frame_view* cur, const olf_frame_view* last, float th, int bMono, int check_orientation,
                             int32_t* matches, int32_t* nmatches);
/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono, map<int,int>& match12),
 * src/ORBmatcher.cc:1474-1618 -- the overload Tracking::TrackWithMotionModelWithLine calls (src/Tracking.cc:1296,1302).  Same search; match12[i2]
 * = the value the reference's map holds under key i2 (-1: no such key): match12.insert keeps the FIRST LastFrame index CurrentFrame feature i2
 * was matched with (:1577) while matches[i2] / mvpMapPoints[i2] keep the last one (:1575); match12.erase on a rotation rejection (:1612).
 * Walking i2 upwards over match12[i2] >= 0 reproduces the map's iteration order. */
int olf_search_by_projection_match12(olf_ctx* ctx, const olf_frame_view* cur, const olf_frame_view* last, float th, int bMono, int check_orientation,
                                     int32_t* matches, int32_t* match12, int32_t* nmatches);
/* int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize),
 * src/ORBmatcher.cc:407-522 (monocular map initialisation).  Only keys / desc / n / the image bounds of the views are read.
 * prev_matched: f1->n (x, y) pairs, updated in place with the matched F2 positions; matches12[i1] = F2 index or -1. */
int olf_search_for_initialization(olf_ctx* ctx, const olf_frame_view* f1, const olf_frame_view* f2, float* prev_matched, int window_size,
                                  float nnratio, int check_orientation, int32_t* matches12, int32_t* nmatches);
/* int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches), src/ORBmatcher.cc:161-290
 * (Tracking::TrackReferenceKeyFrame / Relocalization).  matched[iF] = index of the key-frame feature whose map point feature iF of F
 * received (-1 = none). */
int olf_search_by_bow(olf_ctx* ctx, const olf_frame_view* kf, const olf_frame_view* f, float nnratio, int check_orientation, int32_t* matched,
                      int32_t* nmatches);
/* int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th), src/ORBmatcher.cc:47-131
 * (Tracking::SearchLocalPoints, every frame).  Map point members as arrays of n_mp: mbTrackInView, isBad(), mnTrackScaleLevel,
 * mTrackViewCos, (mTrackProjX, mTrackProjY, mTrackProjXR) interleaved, GetDescriptor(), Observations() > 0.
 * matches[idx] = index into vpMapPoints given to feature idx of F (-1 = none); f->mp_valid / mp_obs are updated. */
int olf_search_local_map(olf_ctx* ctx, const olf_frame_view* f, int n_mp, const uint8_t* track_in_view, const uint8_t* bad,
                         const int32_t* track_scale_level, const float* track_view_cos, const float* track_proj3, const uint8_t* mp_desc,
                         const uint8_t* mp_obs, float th, float nnratio, int32_t* matches, int32_t* nmatches);

/* bool Frame::isInFrustum(MapPoint *pMP, float viewingCosLimit), src/Frame.cc:388-444, for n_mp map points at once (host arithmetic, no
 * device work): the producer of olf_search_local_map's inputs in Tracking::SearchLocalPoints (src/Tracking.cc:1900-1945).  f supplies mTcw,
 * the calibration, the image bounds and mvScaleFactors.  Outputs per point: mbTrackInView, mnTrackScaleLevel, mTrackViewCos and
 * (mTrackProjX, mTrackProjY, mTrackProjXR); a point that fails a gate only gets track_in_view = 0. */
int olf_is_in_frustum(const olf_frame_view* f, int n_mp, const float* world, const float* normal, const float* maxd, const float* mind,
                      float viewing_cos_limit, uint8_t* track_in_view, int32_t* track_scale_level, float* track_view_cos, float* track_proj3);

/* ---- the LocalMapping / LoopClosing / relocalisation searches, same split (host candidates, GPU distances, host resolution) ----
 * int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th,
 * const int ORBdist), src/ORBmatcher.cc:1620-1747.  already_found[i] = sAlreadyFound.count(pKF's i-th map point) (may be NULL);
 * matches[i2] = key-frame feature whose map point CurrentFrame feature i2 received; cur->mp_valid is updated. */
int olf_search_by_projection_kf(olf_ctx* ctx, const olf_frame_view* cur, const olf_frame_view* kf, const uint8_t* already_found, float th,
                                int orb_dist, int check_orientation, int32_t* matches, int32_t* nmatches);
/* int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12), src/ORBmatcher.cc:524-657.
 * matches12[idx1] = feature of pKF2 whose map point is taken (-1 = NULL). */
int olf_search_by_bow_kf(olf_ctx* ctx, const olf_frame_view* kf1, const olf_frame_view* kf2, float nnratio, int check_orientation,
                         int32_t* matches12, int32_t* nmatches);
/* int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t,size_t>> &vMatchedPairs,
 * const bool bOnlyStereo), src/ORBmatcher.cc:659-825.  F12 3x3 row-major; Cw = pKF1->GetCameraCenter() (NULL: derived from kf1->Tcw).
 * matches12[idx1] = idx2 (-1 = none): vMatchedPairs is the list of (idx1, matches12[idx1]) in idx1 order. */
int olf_search_for_triangulation(olf_ctx* ctx, const olf_frame_view* kf1, const olf_frame_view* kf2, const float* F12, const float* Cw,
                                 int only_stereo, int check_orientation, int32_t* matches12, int32_t* nmatches);
/* The search part of int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th), src/ORBmatcher.cc:827-948:
 * per map point (skip = !pMP || isBad() || IsInKeyFrame(pKF); world, normal, mfMaxDistance, mfMinDistance, descriptor) the most similar key
 * point inside the projection window: best_idx / best_dist (-1 / 256 where a gate rejects the point).  Ow = pKF->GetCameraCenter()
 * (NULL: derived from kf->Tcw).  The reference then fuses when best_dist <= TH_LOW (:950-972, map mutation, host code). */
int olf_fuse_search(olf_ctx* ctx, const olf_frame_view* kf, int n_mp, const uint8_t* skip, const float* world, const float* normal,
                    const float* maxd, const float* mind, const uint8_t* desc, float th, const float* Ow, int32_t* best_idx, int32_t* best_dist);
/* The search part of int ORBmatcher::Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, vector<MapPoint*>
 * &vpReplacePoint), src/ORBmatcher.cc:977-1102 (Scw 4x4 row-major; skip = isBad() || spAlreadyFound.count(pMP)); best_dist is INT_MAX
 * where nothing was found. */
int olf_fuse_search_sim3(olf_ctx* ctx, const olf_frame_view* kf, const float* Scw, int n_mp, const uint8_t* skip, const float* world,
                         const float* normal, const float* maxd, const float* mind, const uint8_t* desc, float th, int32_t* best_idx,
                         int32_t* best_dist);
/* int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th),
 * src/ORBmatcher.cc:292-405 (LoopClosing::ComputeSim3, src/LoopClosing.cc:381): the loop candidate's map points projected into the key frame
 * under the Sim3 pose Scw (4x4, row-major).  skip[i] = vpPoints[i]->isBad() || spAlreadyFound.count(vpPoints[i]) (:311-312, :321); the point
 * arrays as in olf_fuse_search_sim3.  matched[idx] (in / out, kf->n bytes) = vpMatched[idx] != NULL: a key point that holds a match is passed
 * over (:378) and a point whose best distance is <= TH_LOW takes its key point at once (:397-401) -- matches[idx] = the index into vpPoints
 * key point idx received in this call (-1 = none); *nmatches = the reference's return value. */
int olf_search_by_projection_sim3(olf_ctx* ctx, const olf_frame_view* kf, const float* Scw, int n_mp, const uint8_t* skip, const float* world,
                                  const float* normal, const float* maxd, const float* mind, const uint8_t* desc, float th, uint8_t* matched,
                                  int32_t* matches, int32_t* nmatches);
/* int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12,
 * const cv::Mat &t12, const float th), src/ORBmatcher.cc:1104-1328.  matches12[i1]: in -- -1 = NULL, >= 0 = pMP->GetIndexInKeyFrame(pKF2),
 * -2 = a map point pKF2 does not observe; out -- additionally the agreed matches.  vn_match1 / vn_match2 = vnMatch1 / vnMatch2. */
int olf_search_by_sim3(olf_ctx* ctx, const olf_frame_view* kf1, const olf_frame_view* kf2, int32_t* matches12, float s12, const float* R12,
                       const float* t12, float th, int32_t* vn_match1, int32_t* vn_match2, int32_t* nfound);

/* ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1795-1811) over all pairs: out[nA][nB] uint16 (host buffers) */
int olf_hamming_matrix(olf_ctx* ctx, const uint8_t* descA, int nA, const uint8_t* descB, int nB, uint16_t* out);

/* ---- Lineextractor (include/LineExtractor.h:40-72, src/LineExtractor.cc:31-67) ------------------ */
/* per-image capacity of the key line / LBD descriptor records (lsd_nfeatures, or the detector's own
 * limit when lsd_nfeatures == 0) */
int olf_line_capacity(const olf_ctx* ctx);
/* Lineextractor::operator()(image, mask [ignored], keylines, descriptors_line): LSDDetectorC::detect with
 * the context's LSD options, top-N by response, BinaryDescriptor::compute (LBD). */
int olf_line_extract_strided(olf_ctx* ctx, const uint8_t* image, size_t row_stride, olf_keyline* kls, uint8_t* ldesc, int32_t* lcount);
int olf_line_extract_dev(olf_ctx* ctx, const uint8_t* d_images, int n_images, olf_keyline* d_kls, uint8_t* d_ldesc, int32_t* d_lcounts,
                         void* stream);
int olf_line_extract(olf_ctx* ctx, const uint8_t* images, int n_images, olf_keyline* kls, uint8_t* ldesc, int32_t* lcounts);
/* BinaryDescriptor::compute(image, keylines, descriptors) on caller-supplied key lines (host buffers;
 * Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp:524-687): kls [n_images][capacity], counts[n_images] */
int olf_lbd_compute(olf_ctx* ctx, const uint8_t* images, int n_images, const olf_keyline* kls, const int32_t* lcounts, uint8_t* ldesc);
/* debug/test: the sigma-0.6 blurred, x1.2 upsampled LSD working image of `image` (dst >= ws*hs bytes) */
int olf_lsd_debug_scaled(olf_ctx* ctx, int image, uint8_t* dst, int32_t* ws, int32_t* hs);

/* ---- Frame::ComputeStereoMatches_Lines (src/Frame.cc:878-1000) + matchGrid (src/LineMatcher.cpp:220-299) */
/* key lines / LBD descriptors of images 2p (left) and 2p+1 (right), stride olf_line_capacity().
 * Outputs per pair, stride capacity: matches_12 (-1 = none), mvDisparity_l (2 floats, -1 = mono),
 * mvle_l (3 doubles, 0 = mono). */
int olf_stereo_lines_dev(olf_ctx* ctx, int n_pairs, const olf_keyline* d_kls, const uint8_t* d_ldesc, const int32_t* d_lcounts,
                         int32_t* d_matches12, float* d_disp, double* d_le, void* stream);
int olf_stereo_lines(olf_ctx* ctx, int n_pairs, const olf_keyline* kls, const uint8_t* ldesc, const int32_t* lcounts, int32_t* matches12,
                     float* disp, double* le);

/* ---- fused entry: the feature part of Frame::Frame (stereo + lines), src/Frame.cc:136-221 -------- */
typedef struct olf_frame_buffers {
    olf_keypoint* kps;      /* [2*n_pairs][orb capacity]      mvKeys / mvKeysRight                   */
    uint8_t*  desc;         /* [2*n_pairs][orb capacity][32]  mDescriptors / mDescriptorsRight       */
    int32_t*  counts;       /* [2*n_pairs]                    N, Nr                                  */
    float*    uright;       /* [n_pairs][orb capacity]        mvuRight                               */
    float*    depth;        /* [n_pairs][orb capacity]        mvDepth                                */
    olf_keyline* kls;       /* [2*n_pairs][line capacity]     mvKeys_Line / mvKeysRight_Line         */
    uint8_t*  ldesc;        /* [2*n_pairs][line capacity][32] mDescriptors_Line / mDescriptorsRight_Line */
    int32_t*  lcounts;      /* [2*n_pairs]                                                           */
    int32_t*  lmatches12;   /* [n_pairs][line capacity]       stereo line matches (left -> right)    */
    float*    ldisp;        /* [n_pairs][line capacity][2]    mvDisparity_l                          */
    double*   lle;          /* [n_pairs][line capacity][3]    mvle_l                                 */
} olf_frame_buffers;
/* ExtractORB x2 + ExtractLine x2 (the reference's 4 threads, src/Frame.cc:164-171, here two HIP streams),
 * ComputeStereoMatches, ComputeStereoMatches_Lines, for n_pairs stereo pairs.  All pointers in `out` are
 * device pointers for the _dev form, host pointers otherwise. */
int olf_stereo_frames_dev(olf_ctx* ctx, const uint8_t* d_images, int n_pairs, const olf_frame_buffers* out, void* stream);
int olf_stereo_frames(olf_ctx* ctx, const uint8_t* images, int n_pairs, const olf_frame_buffers* out);

/* getLineCoords(x1, y1, x2, y2, line_coords), src/gridStructure.cpp:33-41: cells of the reference's Bresenham walk (src/LineIterator.cpp), host
 * arithmetic.  xy receives up to cap (x, y) pairs, *n the number of cells. */
int olf_line_coords(double x1, double y1, double x2, double y2, int32_t* xy, int cap, int32_t* n);

/* ---- multi-GPU: the trimmed wire record of a batch ------------------------------------------------------------------
 * What a rank sends to rank 0 after a batch: header, counts, then only the rows in use of every array of olf_frame_buffers (layout in
 * csrc/records.hip; host mirror and parser in orb_line_slam_amd/records.py).  `out` holds device pointers; d_dst >= olf_frames_pack_bound
 * bytes is always enough; *d_bytes (device or pinned host memory) receives the record size.  A record larger than dst_capacity sets the
 * capacity flag (olf_ctx_synchronize) and writes only the header. */
size_t olf_frames_pack_bound(const olf_ctx* ctx, int n_pairs);
int olf_frames_pack_dev(olf_ctx* ctx, const olf_frame_buffers* out, int n_pairs, uint8_t* d_dst, size_t dst_capacity, uint64_t* d_bytes, void* stream);
/* The map points a frame owns right after stereo matching: d_mask[i] = d_depth[i] > 0, the test of Tracking::StereoInitialization /
 * UpdateLastFrame on mvDepth (src/Tracking.cc:584-588, 1096-1099: `float z = mvDepth[i]; if(z>0)`), as the byte mask d_mp_valid of
 * olf_search_by_bow_batch_dev.  n = number of floats (n_pairs * olf_orb_capacity() for the depth plane of olf_stereo_points_dev);
 * d_depth 16-byte aligned, d_mask 4-byte aligned. */
int olf_stereo_points_mask_dev(olf_ctx* ctx, const float* d_depth, size_t n, uint8_t* d_mask, void* stream);

/* measurement: rate of a plain 16-byte-per-thread device copy kernel over `bytes` (read + written bytes per second): the practical HBM
 * ceiling bench.py reports next to the specification's 8 TB/s */
int olf_debug_copy_bandwidth(olf_ctx* ctx, size_t bytes, int reps, double* gbytes_per_s);

#ifdef __cplusplus
}
#endif
#endif
