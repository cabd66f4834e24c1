/* orbline_types.h -- POD records and parameter blocks shared by the HIP library
 * (liborbline_hip.so), the C++ adaptor and the CPU oracle.
 *
 * Every record mirrors, field for field, a type that crosses the reference's
 * ORBextractor / Lineextractor / ORBmatcher / LineMatcher / Frame surfaces
 * (paths relative to /root/reference):
 *   olf_keypoint  == cv::KeyPoint (28 B) as filled by src/ORBextractor.cc:839-849,1097-1103
 *   olf_keyline   == cv::line_descriptor::KeyLine (68 B),
 *                    Thirdparty/line_descriptor/include/line_descriptor/descriptor_custom.hpp:105-144
 *   descriptors   == rows of a continuous cv::Mat N x 32 CV_8U (src/ORBextractor.cc:1069,
 *                    Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp:634)
 */
#ifndef ORBLINE_TYPES_H
#define ORBLINE_TYPES_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OLF_DESC_BYTES 32      /* 256-bit ORB and LBD descriptors                     */
#define OLF_MAX_LEVELS 16
#define OLF_GRID_COLS 64       /* include/Frame.h:51 FRAME_GRID_COLS                  */
#define OLF_GRID_ROWS 48       /* include/Frame.h:52 FRAME_GRID_ROWS                  */

typedef struct olf_keypoint {
    float x, y;        /* pt, level-0 pixel coordinates                                 */
    float size;        /* (int)(31*scale[octave])                                       */
    float angle;       /* degrees [0,360)                                               */
    float response;    /* FAST score                                                    */
    int32_t octave;
    int32_t class_id;  /* -1, as cv::KeyPoint's default                                 */
} olf_keypoint;

typedef struct olf_keyline {
    float angle;
    int32_t class_id;
    int32_t octave;
    float pt_x, pt_y;
    float response;
    float size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength;
    int32_t numOfPixels;
} olf_keyline;

/* ORBextractor ctor arguments, src/ORBextractor.cc:412-416; values from
 * Examples/PL/PL_KITTI00-02.yaml:42-55 are the defaults of olf_default_params(). */
typedef struct olf_orb_params {
    int32_t nfeatures;
    float   scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    /* convention C.11: fixed-point Gaussian taps of the 7x7 sigma-2 blur.  0: every tap rounded on its own, [18,34,49,55,49,34,18],
     * sum 257 (OpenCV 3.4.0-3.4.5, the range the reference most plausibly linked); 1: error-diffused taps that sum to 256 exactly
     * (later releases' "bit-exact" kernel), [18,34,48,56,48,34,18] */
    int32_t conv_gauss_sum256;
} olf_orb_params;

/* Lineextractor 11-argument ctor (include/LineExtractor.h:44-45) -- the LSD options
 * forwarded at src/LineExtractor.cc:44-53. */
typedef struct olf_line_params {
    int32_t lsd_nfeatures;      /* 0 = keep all                                        */
    double  min_line_length;    /* relative to min(w,h); src/LineExtractor.cc:53       */
    int32_t lsd_refine;         /* 0 LSD_REFINE_NONE, 1 LSD_REFINE_STD, 2 LSD_REFINE_ADV */
    double  lsd_scale;
    double  lsd_sigma_scale;
    double  lsd_quant;
    double  lsd_ang_th;         /* degrees, (0, 180).  Up to 80 the growth agent decides alignment by dot / cross products under a proven margin */
    double  lsd_log_eps;
    double  lsd_density_th;
    int32_t lsd_n_bins;         /* 2 .. 2^24.  Up to 1024 bins and LSD working images below 2^22 pixels (round(lsd_scale w) x round(lsd_scale h)) take the
                                 * fast path; beyond either the 64-bit-key capacity path (csrc/lsd_wide.hip): same results, about 6 x the time */
    /* conventions where the un-vendored OpenCV decides the result and its version is not pinned by the reference:
     * C.11 conv_gauss_sum256: as in olf_orb_params, for LSD's sigma-0.6 blur and LBD's 5x5 sigma-1 blur.
     * C.10 conv_resize_exact: LSD's x1.2 upsampling, 0: cv::resize INTER_LINEAR (11-bit coefficients); 1: INTER_LINEAR_EXACT (8-bit
     *      coefficients, round-to-nearest at the end), which later 3.4.x releases of lsd.cpp call.
     * C.9  conv_seed_order: order of the seeds INSIDE a gradient bin.  1 (default): whatever std::sort(begin, end, norm descending) of
     *      libstdc++ leaves -- OpenCV >= 3.3 pushes every pixel as {point, bin} and calls the unstable std::sort; the reference's CMakeLists.txt
     *      asks for OpenCV 3.4 first.  The order is a property of libstdc++'s introsort run over the whole pixel sequence; csrc/lsd_seedsort.hip
     *      replays it bit-exactly (oracle: the real std::sort).  0: raster order (the linked-list pseudo-ordering of the original LSD and of
     *      OpenCV <= 3.2; a stable radix sort on the device). */
    int32_t conv_gauss_sum256;
    int32_t conv_resize_exact;
    int32_t conv_seed_order;
    /* C.6  conv_libm_float: the unqualified cos / sin / atan2 / sqrt calls on FLOAT arguments inside cv::LineSegmentDetector's region_grow
     *      (sumdx += cos(float(angle))), LSDDetectorC::detectImpl (kl.angle = atan2(dy, dx), LSDDetector_custom.cpp:298) and BinaryDescriptor::computeLBD
     *      (dL = cos / sin(direction), binary_descriptor_custom.cpp:1130-1131; 1 / sqrt(tempM), :1283-1341).  Which overload they resolve to depends on the
     *      headers of the build: 0 (default) = ::cos(double) etc. -- the C functions, result rounded to float by the assignment; 1 = the float overloads
     *      (cosf / sinf / atan2f / sqrtf and a float division), which libstdc++ >= 6 makes visible when <math.h> is included.  Both variants are restated
     *      in the oracle and on the device (glibc 2.35's cosf / sinf / atan2f bit for bit). */
    int32_t conv_libm_float;
} olf_line_params;

/* Camera / matching scalars read on the path:
 * src/Tracking.cc:54-79,156 ; src/Config.cpp:26-160 */
typedef struct olf_stereo_params {
    float  fx;                  /* Camera.fx                                           */
    float  bf;                  /* Camera.bf                                           */
    int32_t matching_s_ws;      /* Config::matchingSWs()                               */
    double line_sim_th;         /* Config::lineSimTh()                                 */
    double min_ratio_12_l;      /* Config::minRatio12L()                               */
    double min_disp;            /* Config::minDisp()                                   */
    double line_horiz_th;       /* Config::lineHorizTh()                               */
    double stereo_overlap_th;   /* Config::stereoOverlapTh()                           */
    double ls_min_disp_ratio;   /* Config::lsMinDispRatio()                            */
    int32_t best_lr_matches;    /* Config::bestLRMatches()                             */
    /* conv_eigen_recip: `le_l = le_l / std::sqrt(...)` on an Eigen::Vector3d (src/Frame.cc:939).  Eigen 3.2+ divides every coefficient (0, default);
     * the 3.0 / 3.1 line -- CMakeLists.txt:45 accepts 3.1.0 -- implements vector / scalar as a multiplication by the reciprocal computed once
     * (scalar_quotient1_impl for non-integer scalars): 1.  mvle_l differs in the last bit between the two. */
    int32_t conv_eigen_recip;
} olf_stereo_params;

/* The parameter block of olf_ctx_create.  It must be initialised by olf_default_params(), which stamps abi_version and struct_size;
 * olf_ctx_create refuses a block whose stamp differs from the library's (a caller built against an older header would otherwise hand over a
 * shorter struct, and the conv_* fields would be read from whatever follows it). */
#define OLF_ABI_VERSION 4u
typedef struct olf_params {
    uint32_t          abi_version;   /* OLF_ABI_VERSION of the header the caller was built with */
    uint32_t          struct_size;   /* sizeof(olf_params) as the caller sees it               */
    olf_orb_params    orb;
    olf_line_params   line;
    olf_stereo_params stereo;
} olf_params;

/* status codes returned by every entry point */
enum {
    OLF_OK = 0,
    OLF_ERR_INVALID = -1,     /* bad argument (null, size mismatch, non-8UC1 ...)      */
    OLF_ERR_CAPACITY = -2,    /* caller-supplied capacity too small                    */
    OLF_ERR_HIP = -3,         /* HIP runtime error, see olf_last_error()               */
    OLF_ERR_NODEVICE = -4     /* no gfx950 device visible                              */
};

#ifdef __cplusplus
}
#endif
#endif
