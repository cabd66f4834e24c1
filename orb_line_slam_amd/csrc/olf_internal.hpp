// olf_internal.hpp -- internal layout of the device context behind the C ABI (include/orbline.h).
//
// One olf_ctx serves a fixed image size (w x h), a fixed parameter block and up to max_images
// images per call (a stereo pair = 2 images: index 2*pair + side).  All buffers are allocated once
// at creation and sized for the worst case, so the hot path never allocates; with 288 GB of HBM3E
// a context for hundreds of KITTI pairs is a few GB.
//
// HBM layout (per image, all level images packed with a 64-byte aligned pitch):
//   pyr   : u8  levels 0..L-1   (level 0 is an aligned copy of the input)      mvImagePyramid
//   blur  : u8  levels 0..L-1   GaussianBlur(7x7, sigma 2) of pyr               workingMat
//   cells : u32 per FAST cell a slot of cell_cap packed candidates (x:12 | y:12 | score:8, cell-local
//           coordinates are converted to level coordinates relative to minBorder) + one count
//   cand  : u32 per level, candidates gathered in reference order (vToDistributeKeys)
//   lvl_kp: u32 per level, octree survivors in lNodes order
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/orbline_types.h"

#define OLF_HIP_CHECK(expr)                                                                 \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            olf::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));              \
            return OLF_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

namespace olf {

void set_error(const std::string& s);

#ifdef __HIPCC__
// Wave-wide vote.  hip's __ballot() converts the predicate to an int first and compares that with 0 (a v_cndmask + v_cmp_ne pair per vote);
// the wave64 builtin takes the compare's lane mask as it is.
// Wave priority experiments (profiles/r4t_wave_priority_ab.txt): -DOLF_GUEST_PRIO=n raises the ORB-stream kernels that run beside the growth agents,
// -DOLF_AGENT_PRIO=n the agents themselves (s_setprio 0 .. 3; 0 = the hardware default, no instruction emitted)
#ifndef OLF_GUEST_PRIO
#define OLF_GUEST_PRIO 0
#endif
#ifndef OLF_AGENT_PRIO
#define OLF_AGENT_PRIO 0
#endif
#define OLF_SET_GUEST_PRIO() do { if (OLF_GUEST_PRIO) __builtin_amdgcn_s_setprio(OLF_GUEST_PRIO); } while (0)
#define OLF_SET_AGENT_PRIO() do { if (OLF_AGENT_PRIO) __builtin_amdgcn_s_setprio(OLF_AGENT_PRIO); } while (0)
// sum over the 64 lanes of a wave, wave-uniform result: two quad permutes, the two row mirrors (DPP, no LDS), then the four rows' sums through scalar registers
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true);       // quad_perm [1, 0, 3, 2]
    v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);       // quad_perm [2, 3, 0, 1]
    v += __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);      // row_half_mirror
    v += __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true);      // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// minimum over the 64 lanes of a wave (signed), wave-uniform result: the same four DPP steps, then the four rows' minima through scalar registers
__device__ __forceinline__ int wave_min_i32(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ unsigned long long wave_vote(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// set bits of a lane mask below this lane (v_mbcnt pair; "__popcll(m & ((1ull << lane) - 1))" compiles to a 64-bit shift, two bit-field inserts and two counts)
__device__ __forceinline__ int wave_rank_below(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }
// this lane's bit of a (wave-uniform) lane mask as a predicate: the mask becomes the EXEC / select operand as it is, no per-lane shift and compare
__device__ __forceinline__ bool wave_bit(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
#endif

constexpr int kEdge = 19;          // EDGE_THRESHOLD   src/ORBextractor.cc:76
constexpr int kMinBorder = 16;     // EDGE_THRESHOLD-3 src/ORBextractor.cc:775
constexpr int kHalfPatch = 15;     // HALF_PATCH_SIZE  src/ORBextractor.cc:75

// Geometry of one pyramid level; identical for every image of the context.
struct LevelGeom {
    int w, h, pitch;        // level size; pitch is a multiple of 64
    int offset;             // byte offset of the level inside a per-image pyramid block
    int maxBorderX, maxBorderY;
    int nCols, nRows, wCell, hCell;   // FAST cell grid, src/ORBextractor.cc:783-789
    uint32_t wCellM, hCellM;          // ceil(2^20 / wCell), ceil(2^20 / hCell): v / cell == (v * M) >> 20 for 0 <= v < 2^14 (k_fast_score's cell of a corner without a division)
    int cellBase;           // index of this level's first cell in the per-image cell arrays
    int quota;              // mnFeaturesPerLevel[level]
    int nIni;               // DistributeOctTree root count
    float hX;               // DistributeOctTree root width
    float scale;            // mvScaleFactor[level]
    float inv_scale;        // mvInvScaleFactor[level]
    int patch_size;         // (int)(31*scale)
    int candBase;           // offset of this level in the per-image candidate array
    int candCap;            // capacity of this level's candidate array
    int kpBase, kpCap;      // per-level survivor array (kpCap = quota + 8)
    int resizeTabX, resizeTabY;  // offsets into the resize coefficient tables (level >= 1)
    int resizeTiled;             // 1: the LDS-tiled resize kernel's staging area covers every tile of this level
};

struct ResizeCoef { int16_t ofs; int16_t a0; int16_t a1; int16_t pad; };

struct OrbGeom {
    int nlevels;
    int W, H, in_pitch;
    int pyrBytes;           // per image
    int totalCells;         // per image
    int cellCap;            // candidates per cell slot
    int candTotal;          // per image
    int kpTotal;            // per image (sum of kpCap)
    int outCap;             // max key points per image returned (nfeatures + 4*nlevels)
    int iniTh, minTh;
    int maxNodes;           // octree LDS node capacity (power of two)
    LevelGeom lv[OLF_MAX_LEVELS];
    int umax[16];
    int blurTaps[7];
};

// Host-side derivation of every table of ORBextractor::ORBextractor (src/ORBextractor.cc:412-472),
// of the level sizes (:1113-1114), of the cell grid (:775-789) and of the cv::resize
// coefficient tables.  Pure host code, shared by api.cpp.
struct OrbHostTables {
    std::vector<float> sf, inv_sf, sigma2, inv_sigma2;
    std::vector<int> nPerLevel;
    OrbGeom geom;
    std::vector<ResizeCoef> rx, ry;    // concatenated over levels
    int build(const olf_orb_params& p, int W, int H);
};

std::vector<int> gaussian_taps_q8(int n, double sigma, int sum256 = 0);
// fills coef[0..dn) for a 1-D cv::resize(INTER_LINEAR) axis: sn source samples, step `scale`
void resize_axis_coefs(int sn, int dn, double scale, bool clamp_like_x, ResizeCoef* coef);

// ----------------------------------------------------------------------------------------------
struct OrbDeviceBufs {
    uint8_t* pyr = nullptr;       // [max_images][pyrBytes]
    uint8_t* blur = nullptr;
    uint32_t* cells = nullptr;    // [max_images][totalCells][cellCap]
    int* cellCount = nullptr;     // [max_images][totalCells]
    uint32_t* cand = nullptr;     // [max_images][candTotal]
    uint16_t* candNode = nullptr; // [max_images][candTotal]
    int* candCount = nullptr;     // [max_images][nlevels]
    uint32_t* lvlKp = nullptr;    // [max_images][kpTotal]
    int* lvlCount = nullptr;      // [max_images][nlevels]
    float* lvlAngle = nullptr;    // [max_images][kpTotal]
    ResizeCoef* rx = nullptr;
    ResizeCoef* ry = nullptr;
    OrbGeom* geom = nullptr;      // device copy
    int* status = nullptr;        // device error flags (capacity overflow etc.)
    unsigned char* octSpill = nullptr;   // [max_images][nlevels][octree_lds_bytes(maxNodes)] only when maxNodes > 2048 (orb_octree.hip)
};

// kernel launchers (orb_*.hip)
int launch_orb_pyramid(const OrbGeom& g, const OrbDeviceBufs& b, const uint8_t* d_in, int n_images, hipStream_t s);
int launch_orb_fast(const OrbGeom& g, const OrbDeviceBufs& b, int n_images, hipStream_t s);
int launch_orb_octree(const OrbGeom& g, const OrbDeviceBufs& b, int n_images, hipStream_t s);
size_t octree_lds_bytes(int M);
int launch_orb_blur(const OrbGeom& g, const OrbDeviceBufs& b, int n_images, hipStream_t s);
int launch_orb_describe(const OrbGeom& g, const OrbDeviceBufs& b, int n_images, olf_keypoint* d_kps, uint8_t* d_desc,
                        int* d_counts, int out_cap, hipStream_t s);

// filters.hip
struct Taps7 { int t[7]; };
int launch_sep7(const uint8_t* src, size_t srcImgStride, int srcPitch, uint8_t* dst, size_t dstImgStride, int dstPitch, int W, int H,
                const int* taps7, int n_images, hipStream_t s);
int launch_sep_wide(const uint8_t* src, size_t srcImgStride, int srcPitch, uint8_t* dst, size_t dstImgStride, int dstPitch, int W, int H,
                    const int* taps, int r, int n_images, hipStream_t s);

bool resize_tiled_fits(const ResizeCoef* rx, const ResizeCoef* ry, int sw, int sh, int dw, int dh);
bool resize_strip_fits(const ResizeCoef* rx, int sw, int dw, int ncols = 4);      // the register-only kernel (bit 1 of resizeTiled; ncols = 5: the fused LSD kernel, bit 2)
int launch_resize_tiled(const uint8_t* src, size_t srcImgStride, int srcPitch, int sw, int sh, uint8_t* dst, size_t dstImgStride, int dstPitch,
                        int dw, int dh, const ResizeCoef* d_rx, const ResizeCoef* d_ry, int n_images, hipStream_t s, bool strip = false);

// precond.hip
int launch_cvt_gray(const uint8_t* src, uint8_t* dst, int w, int h, int code, int n_images, hipStream_t s);
int launch_init_rectify_map(const double* ir9, const double* k8, double fx, double fy, double u0, double v0, int w, int h, float* d_map1, float* d_map2,
                            hipStream_t s);
int launch_remap_linear(const uint8_t* src, int sw, int sh, const float* mapx, const float* mapy, int dw, int dh, uint8_t* dst, int n_images,
                        hipStream_t s);

// match.hip
int launch_stereo_points(const OrbGeom& g, const OrbDeviceBufs& b, int n_pairs, const olf_keypoint* d_kps, const uint8_t* d_desc,
                         const int* d_counts, int cap, float mbf, float fx, float* d_uRight, float* d_depth, int* d_sad, int* d_bestKey,
                         unsigned* d_perm, hipStream_t s);
int launch_match_bf(const uint8_t* dA, const int* nA, int strideA, int aStep, const uint8_t* dB, const int* nB, int strideB, int bStep,
                    int n_sets, float nnr, int best_lr, int* ws, int* m12, hipStream_t s);
int launch_knn2(const uint8_t* dA, const int* nA, int strideA, const uint8_t* dB, const int* nB, int strideB, int n_sets, int* idx0,
                int* dist0, int* dist1, hipStream_t s);
int launch_match_candidates(const uint8_t* q, int nQ, const uint8_t* t, int nT, const int* offs, const int* cand, uint16_t* out, hipStream_t s);
int launch_distinctive(const uint8_t* desc, const int* offs, int n_points, int* best, hipStream_t s);
int launch_hamming_matrix(const uint8_t* a, int nA, const uint8_t* b, int nB, uint16_t* out, hipStream_t s);

}  // namespace olf
