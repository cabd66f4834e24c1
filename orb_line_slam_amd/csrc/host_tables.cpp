// host_tables.cpp -- host-side tables of the ORB extractor (product code, no oracle dependency).
//
// Mirrors ORBextractor::ORBextractor (reference src/ORBextractor.cc:412-472): float scale chain,
// per-level quotas, umax; ComputePyramid's level sizes (:1113-1114); the FAST cell grid of
// ComputeKeyPointsOctTree (:775-789); the root layout of DistributeOctTree (:545-547); and the
// coefficient tables cv::resize(INTER_LINEAR) builds for an 8-bit image and
// cv::getGaussianKernel + 8-bit fixed point conversion (App. A.3).  These are evaluated once per
// context on the host, in the same float/double expressions as the reference, and uploaded.
#include "olf_internal.hpp"
#include "line_internal.hpp"
#include <algorithm>
#include <cmath>
#include <cfenv>

namespace olf {

static inline int cv_round_f(float v) { return (int)lrintf(v); }   // cvRound: half-to-even
static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor(double v) { int i = (int)v; return i - (i > v); }
static inline int cv_ceil(double v) { int i = (int)v; return i + (i < v); }

// sum256: convention C.11, the error-diffused "bit-exact" taps of later OpenCV releases (see include/orbline_types.h)
std::vector<int> gaussian_taps_q8(int n, double sigma, int sum256)
{
    if (sum256) {
        std::vector<double> k(n);
        double s = 0;
        const double scale2X = -0.5 / (sigma * sigma);
        for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; k[i] = std::exp(scale2X * x * x); s += k[i]; }
        std::vector<int> q(n);
        double err = 0;
        int sum = 0;
        for (int i = 0; i < n / 2; ++i) {
            const double adj = k[i] / s * 256.0 + err;
            const int v = cv_round_d(adj);
            err = adj - v;
            q[i] = q[n - 1 - i] = v;
            sum += v;
        }
        q[n / 2] = 256 - 2 * sum;
        return q;
    }
    std::vector<float> cf(n);
    double scale2X = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < n; ++i) {
        double x = i - (n - 1) * 0.5;
        cf[i] = (float)std::exp(scale2X * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    std::vector<int> q(n);
    for (int i = 0; i < n; ++i) {
        cf[i] = (float)(cf[i] * sum);
        q[i] = cv_round_d((double)cf[i] * 256.0);
    }
    return q;
}

void resize_axis_coefs(int sn, int dn, double scale, bool clamp_like_x, ResizeCoef* coef)
{
    for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = cv_floor(f);
        f -= s;
        if (clamp_like_x) {   // horizontal: cv::resize forces the weight when the tap leaves the row
            if (s < 0) { f = 0; s = 0; }
            if (s >= sn - 1) { f = 0; s = sn - 1; }
        }
        int a0 = cv_round_f((1.f - f) * 2048.f), a1 = cv_round_f(f * 2048.f);
        coef[d].ofs = (int16_t)s;   // vertical: rows s and s+1 are clipped to [0, sn) by the kernel
        coef[d].a0 = (int16_t)std::min(std::max(a0, -32768), 32767);
        coef[d].a1 = (int16_t)std::min(std::max(a1, -32768), 32767);
        coef[d].pad = 0;
    }
}

// cv::resize INTER_LINEAR_EXACT (convention C.10): 8-bit coefficients, both taps clamped into the image on either axis
void resize_axis_coefs_exact(int sn, int dn, double scale, ResizeCoef* coef)
{
    for (int d = 0; d < dn; ++d) {
        const double f = scale * (d + 0.5) - 0.5;
        int i = cv_floor(f);
        int a = cv_round_d((f - i) * 256.0);
        if (i < 0) { i = 0; a = 0; }
        if (i >= sn - 1) { i = sn - 1; a = 0; }
        coef[d].ofs = (int16_t)i; coef[d].a0 = (int16_t)(256 - a); coef[d].a1 = (int16_t)a; coef[d].pad = 0;
    }
}

int OrbHostTables::build(const olf_orb_params& p, int W, int H)
{
    const int nlevels = p.nlevels;
    if (nlevels < 1 || nlevels > OLF_MAX_LEVELS || p.nfeatures < 1 || !(p.scale_factor > 1.0f)) return OLF_ERR_INVALID;
    const double scaleFactor = p.scale_factor;   // the member is a double (include/ORBextractor.h:103)
    sf.assign(nlevels, 1.f); sigma2.assign(nlevels, 1.f); inv_sf.resize(nlevels); inv_sigma2.resize(nlevels);
    for (int i = 1; i < nlevels; ++i) {
        sf[i] = (float)(sf[i - 1] * scaleFactor);
        sigma2[i] = sf[i] * sf[i];
    }
    for (int i = 0; i < nlevels; ++i) {
        inv_sf[i] = 1.0f / sf[i];
        inv_sigma2[i] = 1.0f / sigma2[i];
    }
    nPerLevel.resize(nlevels);
    float factor = (float)(1.0f / scaleFactor);
    float nDesired = p.nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; ++l) {
        nPerLevel[l] = cv_round_f(nDesired);
        sum += nPerLevel[l];
        nDesired *= factor;
    }
    nPerLevel[nlevels - 1] = std::max(p.nfeatures - sum, 0);

    OrbGeom& g = geom;
    g = OrbGeom();
    g.nlevels = nlevels; g.W = W; g.H = H; g.in_pitch = W;
    g.iniTh = std::min(std::max(p.ini_th_fast, 1), 255);
    g.minTh = std::min(std::max(p.min_th_fast, 1), 255);
    {
        int v, v0, vmax = cv_floor(kHalfPatch * std::sqrt(2.f) / 2 + 1);
        int vmin = cv_ceil(kHalfPatch * std::sqrt(2.f) / 2);
        const double hp2 = kHalfPatch * kHalfPatch;
        for (v = 0; v <= vmax; ++v) g.umax[v] = cv_round_d(std::sqrt(hp2 - v * v));
        for (v = kHalfPatch, v0 = 0; v >= vmin; --v) {
            while (g.umax[v0] == g.umax[v0 + 1]) ++v0;
            g.umax[v] = v0;
            ++v0;
        }
    }
    std::vector<int> taps = gaussian_taps_q8(7, 2.0, p.conv_gauss_sum256);
    for (int i = 0; i < 7; ++i) g.blurTaps[i] = taps[i];

    rx.clear(); ry.clear();
    int off = 0, cells = 0, cand = 0, kps = 0, cellCap = 1, maxQuota = 1, maxKp = 1;
    for (int l = 0; l < nlevels; ++l) {
        LevelGeom& L = g.lv[l];
        L.w = cv_round_f((float)W * inv_sf[l]);
        L.h = cv_round_f((float)H * inv_sf[l]);
        // every level must hold at least one 30 px FAST cell and a landscape octree root
        if (L.w - 2 * kMinBorder < 30 || L.h - 2 * kMinBorder < 30) return OLF_ERR_INVALID;
        L.pitch = (L.w + 63) & ~63;
        L.offset = off;
        off += L.pitch * L.h;
        off = (off + 255) & ~255;
        L.maxBorderX = L.w - kMinBorder; L.maxBorderY = L.h - kMinBorder;
        const float width = (float)(L.maxBorderX - kMinBorder), height = (float)(L.maxBorderY - kMinBorder);
        L.nCols = (int)(width / 30.f); L.nRows = (int)(height / 30.f);
        L.wCell = (int)std::ceil(width / L.nCols); L.hCell = (int)std::ceil(height / L.nRows);
        // (M = ceil(2^20 / d): e = M d - 2^20 < d, and floor(v M / 2^20) == v / d as long as v e < 2^20 -- v < 2^20 / d; levels are at most 2^14 wide, cells at most 64)
        L.wCellM = (uint32_t)(((1u << 20) + L.wCell - 1) / L.wCell); L.hCellM = (uint32_t)(((1u << 20) + L.hCell - 1) / L.hCell);
        if (L.w >= (1 << 14) || L.h >= (1 << 14) || L.wCell > 64 || L.hCell > 64) { L.wCellM = 0; L.hCellM = 0; }      // (0: the kernel divides)
        L.cellBase = cells;
        cells += L.nCols * L.nRows;
        cellCap = std::max(cellCap, ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2));
        L.quota = nPerLevel[l];
        maxQuota = std::max(maxQuota, L.quota);
        L.nIni = (int)std::round(width / height);
        if (L.nIni < 1) return OLF_ERR_INVALID;   // portrait images divide by zero in the reference
        L.hX = width / L.nIni;
        L.scale = sf[l]; L.inv_scale = inv_sf[l];
        L.patch_size = (int)(31 * sf[l]);
        // DistributeOctTree returns at most quota + 3 nodes once its size checks run, but the first sweep splits all nIni root nodes
        // unconditionally (src/ORBextractor.cc:556-640): a small quota can come back as 4 * nIni key points
        L.kpBase = kps; L.kpCap = std::max(L.quota + 8, 4 * L.nIni + 4); kps += L.kpCap;
        maxKp = std::max(maxKp, L.kpCap);
        if (l > 0) {
            const LevelGeom& P = g.lv[l - 1];
            L.resizeTabX = (int)rx.size(); L.resizeTabY = (int)ry.size();
            rx.resize(rx.size() + L.w); ry.resize(ry.size() + L.h);
            // cv::resize with an explicit dsize: inv_scale = dsize/ssize, scale = 1/inv_scale
            double inv_x = (double)L.w / P.w, inv_y = (double)L.h / P.h;
            resize_axis_coefs(P.w, L.w, 1. / inv_x, true, &rx[L.resizeTabX]);
            resize_axis_coefs(P.h, L.h, 1. / inv_y, false, &ry[L.resizeTabY]);
            L.resizeTiled = resize_tiled_fits(&rx[L.resizeTabX], &ry[L.resizeTabY], P.w, P.h, L.w, L.h) ? 1 : 0;
            if (L.resizeTiled && resize_strip_fits(&rx[L.resizeTabX], P.w, L.w)) L.resizeTiled |= 2;      // the register-only kernel applies too
        }
    }
    g.pyrBytes = off;
    g.totalCells = cells;
    g.cellCap = cellCap;
    if (cellCap > 1024) return OLF_ERR_CAPACITY;      // k_cells_sort orders a cell's candidates in a 1024-entry LDS array
    for (int l = 0; l < nlevels; ++l) {
        LevelGeom& L = g.lv[l];
        long cap = (long)L.nCols * L.nRows * cellCap;
        L.candCap = (int)std::min<long>(cap, 65535);
        L.candBase = cand;
        cand += (L.candCap + 63) & ~63;
    }
    g.candTotal = cand;
    g.kpTotal = kps;
    g.outCap = kps;
    int mn = 64;
    while (mn < maxKp) mn <<= 1;
    g.maxNodes = mn;
    // k_octree keeps a level's node list in LDS (orb_octree.hip, octree_lds_bytes: 66 bytes per node + 16 KB): 2048 nodes fit the 160 KB,
    // i.e. up to 2040 key points on one level; larger levels spill the lists to global memory (node ids are 16-bit: at most 32768 nodes)
    if (mn > 32768) return OLF_ERR_CAPACITY;
    return OLF_OK;
}

// ---------------------------------------------------------------------------------------------
// Line side: constants of cv::LineSegmentDetector (OpenCV 3.4 lsd.cpp) as set up
// by LSDDetectorC::detectImpl (Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:246-253) and the
// LBD weights of BinaryDescriptor (binary_descriptor_custom.cpp:217-259).
int LineHostTables::build(const olf_line_params& p, int W, int H, int max_images)
{
    LineGeom& g = geom;
    g = LineGeom();
    if (p.lsd_refine < 0 || p.lsd_refine > 2) return OLF_ERR_INVALID;        // LSD_REFINE_NONE / STD / ADV
    if (p.conv_seed_order != 0 && p.conv_seed_order != 1) return OLF_ERR_INVALID;
    if (p.conv_libm_float != 0 && p.conv_libm_float != 1) return OLF_ERR_INVALID;
    if (!(p.lsd_scale > 0) || p.lsd_n_bins < 2 || p.lsd_n_bins > (1 << 24) || !(p.lsd_ang_th > 0 && p.lsd_ang_th < 180)) return OLF_ERR_INVALID;
    const double kPI = 3.1415926535897932384626433832795;
    g.W = W; g.H = H; g.pitchW = (W + 63) & ~63; g.pitchD = (W + 3) & ~3;
    g.scale = p.lsd_scale;
    g.Ws = cv_round_d(W * p.lsd_scale); g.Hs = cv_round_d(H * p.lsd_scale);
    g.pitchS = (g.Ws + 63) & ~63;
    g.Ps = g.Ws * g.Hs;
    // chunk pool of the multi-wave growth: every pixel in a list once (Ps / 32) plus one partly filled chunk per logged region and ROB slot; at least the
    // 2 * Ps words the one-wave agent's (pixel, gradient word) log needs -- one stride for both formats (see LineGeom::regionStride)
    {   // division of a pixel index (< 2^22) by Ws as a multiply-high: p = max(32, 22 + ceil(log2 Ws)), M = ceil(2^p / Ws); the error term e = M Ws - 2^p < Ws
        // satisfies idx * e < 2^22 * 2^ceil(log2 Ws) <= 2^p, which is the condition for floor(idx M / 2^p) == idx / Ws
        int k = 0; while ((1 << k) < g.Ws) ++k;
        const int p = std::max(32, 22 + k);
        const unsigned long long M = ((1ull << p) + (unsigned long long)g.Ws - 1) / (unsigned long long)g.Ws;
        g.divWsM = (uint32_t)M; g.divWsS = p - 32;
    }
    g.libmFloat = p.conv_libm_float ? 1 : 0;
    g.alignDeg = p.lsd_ang_th < 90 ? (float)(180.0 - p.lsd_ang_th) : -1.f;      // (tolerances of 90 degrees and more: the folded form does not hold, k_lsd_keys decides in double)
    g.regionStride = std::max(1024 + g.Ps / 32 + 64, (2 * g.Ps + 31) / 32) * 32;
    // batch contexts (more than kBatchCtxImages images; only the one-wave agent runs there): the log is sized for HALF of the pixels in logged regions -- the
    // bench scene logs 105 k of 670 k, its long-line scene 231 k -- and an image that needs more is grown again on a full-size block of the spill arena by a
    // second launch (k_lsd_grow `retry`).  The seed sort's grid-wide top levels keep their 2 * (Ps / 64 + 2 * SS_TOP_JOBS + 4) words of scratch.
    if (max_images > kBatchCtxImages) g.regionStride = ((std::max(g.Ps, 8192) + 31) / 32) * 32;
    // (more than 1024 bins or 2^22 pixels and more: the 64-bit keys of lsd_wide.hip -- the capacity path; 2^27 pixels bound the 32-bit index arithmetic)
    if ((long)g.Ws * g.Hs >= (1L << 27) || g.Ws < 8 || g.Hs < 8 || g.Ws > 32767 || g.Hs > 32767) return OLF_ERR_INVALID;
    g.wide = (p.lsd_n_bins > 1024 || (long)g.Ws * g.Hs >= (1L << 22)) ? 1 : 0;
    g.prec = kPI * p.lsd_ang_th / 180;
    {   // 2*pi - prec in 64-bit-mantissa arithmetic is exact (two doubles three binades apart), then rounded up to a double
        const long double w = (long double)(2 * kPI) - (long double)g.prec;
        double t = (double)w;
        if ((long double)t < w) t = std::nextafter(t, 1e300);
        g.precWrap = t;
    }
    {   // the cheap alignment test of the growth agent.  m = 6e-4 rad: cv::fastAtan2's polynomial is within 0.00955 degrees of the true angle over every float
        // quotient (exhaustive: tools/micro/fastatan2_bound.c), its three reflections add 3 half-ulps of 360 (4.6e-5 degrees), together 1.7e-4 rad; the
        // direction a table entry stands for (cos / sin of the float-rounded angle, rounded to float) is within 4e-7 rad of the entry's angle and the float dot /
        // cross products within 4e-7 rad of the exact ones -- the margin is three times the sum
        const double m = 6e-4;
        if (p.lsd_ang_th <= 80.0) {
            g.alignTanLo = std::nextafter((float)std::tan(g.prec - m), 0.f);
            g.alignTanHi = std::nextafter((float)std::tan(g.prec + m), 1e30f);
        } else { g.alignTanLo = -1.f; g.alignTanHi = -1.f; }
    }
    const double pp = p.lsd_ang_th / 180;
    const double rho = p.lsd_quant / std::sin(g.prec);
    int n = 0;
    while (!(std::sqrt(n / 4.0) > rho)) ++n;
    g.nThr = n;
    g.nBins = p.lsd_n_bins;
    const double LOG_NT = 5 * (std::log10(double(g.Ws)) + std::log10(double(g.Hs))) / 2 + std::log10(11.0);
    g.minRegSize = int(-LOG_NT / std::log10(pp));
    g.logNT = LOG_NT; g.logEps = p.lsd_log_eps; g.pProb = pp;
    g.minLength = p.min_line_length * std::min(W, H);
    // raw segments kept per image before the top-N: a region owns at least minRegSize pixels, and only noise comes near one segment per 48 pixels
    // (round 3's 4096 / 8192 refused pure noise at lsd_scale 2; k_line_select sorts lists beyond 8192 in global memory)
    g.maxDetect = std::min(32768, std::max(4096, g.Ps / 48));
    // lsd_nfeatures = 0 keeps every segment: the LBD row sums (63 x 16 bytes per kept line) then scale with maxDetect -- a big batch keeps round 3's capacity
    if (p.lsd_nfeatures == 0) g.maxDetect = std::min(g.maxDetect, std::max(4096, (int)(16e9 / ((double)std::max(max_images, 1) * 63 * 16))));
    // 16-byte region records alias the unsorted key buffer, 24-byte segment candidates the sorted one (4 bytes per pixel each)
    g.maxRegions = std::min(g.Ps / std::max(g.minRegSize, 1) + 1, g.Ps / 6 - 1);
    g.rectGrid = g.maxRegions;
    g.nFeatures = p.lsd_nfeatures;
    g.outCap = p.lsd_nfeatures > 0 ? p.lsd_nfeatures : g.maxDetect;
    for (int i = 0; i < 7; ++i) { g.lsdTaps[i] = 0; g.lbdTaps[i] = 0; }
    g.lsdWideR = 0;
    for (int i = 0; i < 15; ++i) g.lsdWide[i] = 0;
    if (p.lsd_scale != 1) {
        const double sigma = (p.lsd_scale < 1) ? (p.lsd_sigma_scale / p.lsd_scale) : p.lsd_sigma_scale;
        const unsigned hk = (unsigned)std::ceil(sigma * std::sqrt(2 * 3.0 * std::log(10.0)));
        if (hk > 7 || (int)(2 * hk) >= std::min(W, H)) return OLF_ERR_INVALID;   // kernels wider than 15 taps (lsd_scale < 0.32 at the default sigma) are not implemented
        std::vector<int> t = gaussian_taps_q8(1 + 2 * hk, sigma, p.conv_gauss_sum256);
        if (hk > 3) { g.lsdWideR = (int)hk; for (unsigned i = 0; i < t.size(); ++i) g.lsdWide[i] = t[i]; }      // the general kernel (k_sep_wide)
        else for (unsigned i = 0; i < t.size(); ++i) g.lsdTaps[3 - hk + i] = t[i];
    } else g.lsdTaps[3] = 256;               // identity: cv::LineSegmentDetector skips blur+resize at scale 1
    {
        std::vector<int> t = gaussian_taps_q8(5, 1.0, p.conv_gauss_sum256);
        for (int i = 0; i < 5; ++i) g.lbdTaps[1 + i] = t[i];
    }
    {   // integer divisions are the reference's (binary_descriptor_custom.cpp:224-257)
        const int widthOfBand = 7, numBands = 9;
        double u = (widthOfBand * 3 - 1) / 2;
        double sigma = (widthOfBand * 2 + 1) / 2;
        double invsigma2 = -1 / (2 * sigma * sigma);
        for (int i = 0; i < widthOfBand * 3; ++i) { double dis = i - u; g.gaussCoefL[i] = (float)std::exp(dis * dis * invsigma2); }
        u = (numBands * widthOfBand - 1) / 2;
        sigma = u;
        invsigma2 = -1 / (2 * sigma * sigma);
        for (int i = 0; i < numBands * widthOfBand; ++i) { double dis = i - u; g.gaussCoefG[i] = (float)std::exp(dis * dis * invsigma2); }
    }
    rx.resize(g.Ws); ry.resize(g.Hs);
    g.resizeTabX = 0; g.resizeTabY = 0;
    // resize(gaussian_img, scaled_image, Size(), SCALE, SCALE, INTER_LINEAR): scale_x = 1/SCALE
    g.resizeExact = p.conv_resize_exact ? 1 : 0;
    g.seedOrder = p.conv_seed_order;
    g.refine = p.lsd_refine;
    g.densityTh = p.lsd_density_th;
    if (g.resizeExact) {
        resize_axis_coefs_exact(W, g.Ws, 1. / p.lsd_scale, rx.data());
        resize_axis_coefs_exact(H, g.Hs, 1. / p.lsd_scale, ry.data());
        g.resizeTiled = 0;                                 // the tiled kernel implements INTER_LINEAR's arithmetic only
    } else {
        resize_axis_coefs(W, g.Ws, 1. / p.lsd_scale, true, rx.data());
        resize_axis_coefs(H, g.Hs, 1. / p.lsd_scale, false, ry.data());
        g.resizeTiled = resize_tiled_fits(rx.data(), ry.data(), W, H, g.Ws, g.Hs) ? 1 : 0;
        if (g.resizeTiled && resize_strip_fits(rx.data(), W, g.Ws)) g.resizeTiled |= 2;
        if ((g.resizeTiled & 2) && resize_strip_fits(rx.data(), W, g.Ws, 5) && g.pitchW >= 8 && (g.pitchS & 3) == 0) g.resizeTiled |= 4;      // k_lsd_upgrad
    }
    return OLF_OK;
}

}  // namespace olf
