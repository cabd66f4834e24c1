// line_internal.hpp -- device layout of the line half of the path (Lineextractor: LSD + LBD) and
// of the stereo line matcher.  See lsd.hip / lbd.hip / linematch.hip.
//
// HBM layout per image (W x H input, Ws x Hs = round(lsd_scale W) x round(lsd_scale H) LSD working size, Ps = Ws * Hs):
//   lsdBlur  : u8  H  x pitchW     GaussianBlur(sigma_scale) of the input                  (LSD step 1)
//   scaled   : u8  Hs x pitchS     bilinear rescale                                        (LSD step 1)
//   grad     : u32 Ps              gx:11 | gy:11 | ISO (bit 22) | NOTDEF (bit 30) | USED (bit 31)   (LSD ll_angle + region growing state)
//   keysA/B  : u32 Ps each         ((n_bins-1-bin) << 22 | address) of every defined pixel, raster order / sorted; once consumed, keysA holds
//                                  the 16-byte records of the logged regions and keysB the 24-byte segment candidates
//   region   : u32 Ps              level-line angles (degrees) for k_lsd_iso, then the pixel log of the grown regions (x | y << 16)
//   rawLines : olf_keyline maxDetect  key lines in detection order (after the length filter)
//   lbdBlur  : u8  H x pitchW      GaussianBlur(5x5, sigma 1)                              (LBD)
//   dxdy     : u32 H x W           packed Sobel (dx, dy) int16 pair                        (LBD)
//   rowSums  : float4 nLines x 63  per support-region row: (pgdL, ngdL, pgdO, ngdO)        (LBD)
// Per context (image independent): angDeg (4 B) and angEnt (32 B), 2^22 entries each (see LineDeviceBufs).
#pragma once
#include "olf_internal.hpp"

namespace olf {

struct LineGeom {
    int W, H, pitchW;          // input size, 64-byte aligned pitch of the u8 work images
    int Ws, Hs, pitchS, Ps;    // LSD working size, Ps = Ws*Hs
    int nThr;                  // smallest gx^2+gy^2 whose norm sqrt(n/4.0) exceeds rho
    int nBins;
    int wide;                  // lsd_n_bins > 1024 or Ps >= 2^22: the seed order takes the 64-bit keys of lsd_wide.hip (keysA / keysB hold 8 bytes per pixel), the growth the one-wave agent
    int minRegSize;
    double prec;               // pi * ang_th / 180
    double precWrap;           // smallest double >= 2*pi - prec (exact): n >= precWrap <=> |n - 2*pi| <= prec for n in (3*pi/2, 2*pi + prec]
    double scale;              // lsd_scale
    double minLength;          // min_line_length * min(W, H)
    int maxDetect;             // capacity of the raw key line list
    int rectGrid;              // regions per image covered by a k_lsd_rect launch (blocks past regCount exit at once)
    int maxRegions;            // capacity of the per-image region log (a logged region owns >= minRegSize pixels of its own)
    int nFeatures;             // lsd_nfeatures (0 = keep all)
    int outCap;                // key lines returned per image
    int pitchD;                // row stride of dxdy in pixels: W rounded up to a multiple of 4 (16-byte stores)
    int lsdTaps[7];            // sigma 0.6 (7x7)
    int lsdWideR;              // > 3: LSD's blur is wider than 7 taps (lsd_scale < 0.74 or a large lsd_sigma_scale): radius, taps in lsdWide
    int lsdWide[15];
    int lbdTaps[7];            // sigma 1 (5x5, zero padded to 7)
    float gaussCoefL[21];      // (float) of the reference's double weights
    float gaussCoefG[63];
    int resizeTabX, resizeTabY;
    int resizeTiled;
    int seedOrder;             // convention C.9: 0 raster order inside a gradient bin (stable radix sort), 1 libstdc++'s std::sort order (lsd_seedsort.hip)
    int refine;                // lsd_refine: 0 LSD_REFINE_NONE, 1 LSD_REFINE_STD (density check, second growth, reduce_region_radius inside the agent)
    double densityTh;          // lsd_density_th
    double logNT, logEps, pProb;   // LSD_REFINE_ADV: 5 (log10 Ws + log10 Hs) / 2 + log10 11 (host libm), lsd_log_eps, ang_th / 180
    int libmFloat;             // convention C.6 (conv_libm_float): 1 = the float overloads of cos / sin / atan2 / sqrt inside LSD / KeyLine / LBD
    uint32_t divWsM; int divWsS; // idx / Ws for 0 <= idx < 2^22 without a division: __umulhi(idx, divWsM) >> divWsS (exact: host_tables.cpp)
    // the growth agent's cheap alignment test (lsd.hip, PF bit 16): a pixel whose level-line direction d (unit vector, AngEnt::seed) makes the angle D with the
    // region's float sums S is aligned for certain if |S x d| <= tan(prec - m) S.d - delta, not aligned for certain if |S x d| >= tan(prec + m) S.d + delta; m
    // covers the error of cv::fastAtan2 (the reference's region angle) against the true angle of S.  alignTanLo < 0: the tolerance is too wide for the folded
    // form (ang_th > 80 degrees) -- every decision takes the reference's expression
    float alignTanLo, alignTanHi;
    float alignDeg;            // 180 - lsd_ang_th as a float: two level-line angles (degrees) a, b are aligned <=> | |a - b| - 180 | >= alignDeg (decided exactly in double near the boundary)
    int regionStride;          // 32-bit words per image of LineDeviceBufs::region: ONE stride for both pixel-list formats (chunk chains of the multi-wave growth: regionStride / 32
                               // chunks; contiguous (pixel, gradient word) log of the one-wave agent: 2 * Ps words), so a fallen-back image never lands in a neighbour's chunks
    int resizeExact;           // convention C.10: the upsampling is cv::resize INTER_LINEAR_EXACT (8-bit coefficients in rx / ry)
    // the one-wave agent's pixel log (filled in by olf_ctx_create / olf_debug_lsd_log_cap): entries of the image's own log, and the spill arena -- blocks of Ps
    // entries for the images that outgrow it; spillCtl[0] counts the blocks handed out in a call, spillOf[img] is the image's block or -1.  Read from here (scalar
    // loads on the rare path) instead of travelling as kernel arguments: the agent has no scalar registers to spare
    int logCap;
    int spillBlocks;
    uint32_t* spillArena;
    int* spillCtl;
    int* spillOf;
};

struct LineDeviceBufs {
    uint8_t* lsdBlur = nullptr;
    uint8_t* scaled = nullptr;
    uint32_t* grad = nullptr;
    uint32_t* keysA = nullptr;
    uint32_t* keysB = nullptr;
    int* keyCount = nullptr;       // [n] defined-pixel count
    int* maxN = nullptr;           // [n] max gx^2+gy^2 over defined pixels
    int* chunkCnt = nullptr;       // [n][ceil(Ps / 4096)] defined pixels per gradient chunk (raster-ordered key emission)
    uint32_t* region = nullptr;
    olf_keyline* rawLines = nullptr;
    int* rawCount = nullptr;
    int* regCount = nullptr;       // [n] regions logged by the agent (records alias keysA, free once the keys are sorted)
    uint8_t* lbdBlur = nullptr;
    uint32_t* dxdy = nullptr;
    float* rowSums = nullptr;      // [n][outCap][63][4]
    float* lbdStarts = nullptr;    // [n][outCap][64][2] start of every support-region row + (dL0, dL1) in slot 63 (k_lbd_prep)
    ResizeCoef* rx = nullptr;
    ResizeCoef* ry = nullptr;
    LineGeom* geom = nullptr;
    uint32_t* sortHist = nullptr;  // [n][ceil(Ps / 8192)][32] radix sort of the keys (lsd_sort.hip): per-chunk digit histograms -> prefixes
    uint32_t* sortBase = nullptr;  // [n][32] where each digit value's bucket starts
    int* status = nullptr;
    float* angDeg = nullptr;       // [2^22] level-line angle (degrees) of the packed gradient pair (gx:11 | gy:11), image independent
    void* angEnt = nullptr;        // [2^22] AngEnt (lsd_device.hpp): angle in radians, cos / sin as an added pixel, the sums a seed starts with -- 32 B
    int ownerImages = 0;           // images `owner` is sized for: the multi-wave growth runs on at most 3072 images per call (lsd_grow_waves), larger calls take the one-wave agent, which has no owner words
    uint32_t* owner = nullptr;     // [ownerImages][Ps] region growing: FREE or (seed rank << 10 | ROB slot) of the region that claimed the pixel (lsd_grow.hip)
    int* links = nullptr;          // [n][nChunks] next chunk of a region's pixel list (-1: last)
    int nChunks = 0;               // 32-pixel chunks per image in `region` (ids < 1024: the ROB slots' own chunks, then the pool)
    bool skipScaled = false;         // fused stereo entry: the enlarged working image is consumed inside k_lsd_upgrad and not written (olf_lsd_debug_scaled needs the stand-alone entry)
    hipEvent_t sortEvent = nullptr;  // when set, launch_lsd_front records it in front of the seed ordering (the dense, bandwidth-bound part of the front is through)
    int* growFmt = nullptr;        // [n] after the multi-wave growth: 0 chunk chains, -1 given up (pool exhausted), 1 grown again by the one-wave agent (contiguous log)
    int poolChunks = 0;            // olf_debug_lsd_pool: > 0 caps the chunk pool the multi-wave kernel may use (tests of the fall-back)
    int* topBuf = nullptr;         // [n][SS_TOP_WORDS] job lists / counters / final ranges of the seed sort's grid-wide top levels (lsd_seedsort.hip)
    int forceSortMode = -1;        // olf_debug_seed_sort_mode: 0 one wave per image, 1 / 2 the 4- / 8-wave kernel of lsd_seedsort.hip; -1: by batch size
    int forceNW = -1, forceE = 0;  // olf_debug_lsd_waves: waves per image (0: the one-wave agent) and ROB entries of the growth kernel; -1 / 0: automatic
    unsigned char* mg = nullptr;   // [mgImages][mgStride] several workgroups per image (lsd_grow.hip, MG): control words, steal-notice words, the groups' staging lists of logged regions
    size_t mgStride = 0;
    int mgImages = 0;              // images `mg` is sized for (small batches only: the latency path)
    int forceG = -1;               // olf_debug_lsd_groups: workgroups per image of the multi-wave growth (1, 2, 4); -1: by batch size
    int scatter = 0;               // olf_debug_lsd_scatter: the groups of an image on consecutive blocks (different XCDs) instead of on one XCD
    bool chained = false;          // the last growth wrote chunk chains (multi-wave kernel), not the contiguous log of the one-wave agent
    // the one-wave agent's pixel log is sized by a measured bound in batch contexts (LineGeom::regionStride): an image whose logged regions outgrow it moves on
    // to a block of the spill arena -- a full-size log (Ps entries); spillCtl[0] counts the blocks handed out in a call, spillOf[img] is the image's block or -1
    uint32_t* spill = nullptr; int* spillCtl = nullptr; int* spillOf = nullptr; int spillBlocks = 0;
    int logCapOverride = 0;        // olf_debug_lsd_log_cap: > 0 caps the primary log (entries) -- tests of the spill path
};

#ifdef OLF_NO_BATCH_CTX
constexpr int kBatchCtxImages = 1 << 30;      // (A/B builds)
#else
constexpr int kBatchCtxImages = 2048;
#endif
//        // contexts for more images than this are batch contexts: the one-wave agent only (no owner words), pixel log sized by a bound + spill arena

struct LineHostTables {
    LineGeom geom;
    std::vector<ResizeCoef> rx, ry;
    int build(const olf_line_params& p, int W, int H, int max_images = 2);
};

int launch_lsd_front(const LineGeom& g, LineDeviceBufs& b, const uint8_t* d_in, int in_pitch, int n_images, hipStream_t s);
int launch_lsd_grow(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s);
int launch_lsd_rect(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s);
int launch_lbd_dense(const LineGeom& g, const LineDeviceBufs& b, const uint8_t* d_in, int in_pitch, int n_images, hipStream_t s);
int launch_line_select_lbd(const LineGeom& g, const LineDeviceBufs& b, const uint8_t* d_in, int in_pitch, int n_images,
                           olf_keyline* d_kls, uint8_t* d_desc, int* d_counts, hipStream_t s, bool denseDone = false);
int launch_lbd_only(const LineGeom& g, const LineDeviceBufs& b, const uint8_t* d_in, int in_pitch, int n_images, const olf_keyline* d_kls,
                    uint8_t* d_desc, const int* d_counts, hipStream_t s);
int launch_gauss7_img(const uint8_t* src, int srcPitch, size_t srcStride, uint8_t* dst, int dstPitch, size_t dstStride, int W, int H,
                      const LineGeom& g, int which, int n_images, hipStream_t s);
int launch_lsd_angle_table(LineDeviceBufs& b, int libmFloat, hipStream_t s);
int launch_fdiv_sweep(unsigned long long seed, int blocks, int per_thread, unsigned long long* d_mismatches, hipStream_t s);
int launch_sqrtq_sweep(int count, unsigned long long* d_mismatches, hipStream_t s);
int launch_align_sweep(const LineDeviceBufs& b, unsigned long long seed, int blocks, int per_thread, unsigned long long* d_out, hipStream_t s);
int lsd_sort_max_chunks(int Ps);
size_t lsd_grow_mg_stride(int maxRegions);   // bytes per image of LineDeviceBufs::mg
constexpr int kMwMaxImages = 3072;           // images per call up to which the multi-wave growth is chosen (lsd_grow_waves)
constexpr int kMgMaxImages = 64;             // images grown by several workgroups each in one call, at most
int lsd_seedsort_top_words();      // ints per image of LineDeviceBufs::topBuf
int launch_lsd_seedsort(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s, int nOverride, int kthrOverride, int depthOverride);
int launch_lsd_sort_wide(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s, int nOverride, long long kthrOverride, int depthOverride, int fullOverride);

size_t stereo_lines_prep_bytes(int n_images, int cap);
int launch_stereo_lines(int W, int H, const olf_stereo_params& P, int n_pairs, const olf_keyline* d_kls, const uint8_t* d_desc,
                        const int* d_counts, int cap, void* d_prep, uint16_t* d_dist /* [pairs][cap][cap] */, int* d_m21, int* d_m12,
                        float* d_disp, double* d_le, hipStream_t s);

}  // namespace olf
