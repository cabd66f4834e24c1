// line_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// Restates the line half of the path (paths relative to /root/reference):
//   Lineextractor::operator()            src/LineExtractor.cc:31-67
//   LSDDetectorC::detect/detectImpl      Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:56-102, :218-324
//   cv::createLineSegmentDetector->detect  (un-vendored OpenCV 3.4 lsd.cpp)
//   cv::LineIterator count               (App. A.8)
//   BinaryDescriptor::compute (LBD)      Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp:217-259 (weights),
//                                        :350-412 (blur, Sobel, binaryConversion), :539-687 (computeImpl), :1026-1372 (computeLBD)
// PARITY UNPINNED (see oracle_common.hpp).  Conventions:
//   C.3  the top-N selection by response uses a stable order (response desc, detection index asc);
//        orc_line_extract can also run the reference's std::sort so tests can check that they agree.
//   C.6  unqualified libm calls on float arguments inside cv::line_descriptor / cv::LineSegmentDetector
//        (cos, sin, atan2, sqrt) are evaluated in double and rounded to float.
#include "oracle_common.hpp"
#include <cfloat>
#include <chrono>

namespace orc {

// wall time of the last line_extract of this thread: [0] LSD, [1] key lines + top-N + LBD (bench.py's per-stage CPU breakdown)
thread_local double g_line_stage_ms[2] = {0, 0};

static const double kPI = 3.1415926535897932384626433832795;
static const double NOTDEF = -1024.0, M_3_2_PI = (3 * kPI) / 2, M_2__PI = 2 * kPI, DEG_TO_RADS = kPI / 180;

struct RegionPoint { int x, y; double angle, modgrad; };
struct Vec4f { float v[4]; };

struct LsdState {
    int w = 0, h = 0;
    std::vector<double> angles, modgrad;
    std::vector<uint8_t> used;
    std::vector<int> order;   // pixel addresses, bins high->low, raster inside a bin
};

static inline bool is_aligned(const LsdState& S, int addr, double theta, double prec)
{
    const double a = S.angles[addr];
    if (a == NOTDEF) return false;
    double n_theta = theta - a;
    if (n_theta < 0) n_theta = -n_theta;
    if (n_theta > M_3_2_PI) {
        n_theta -= M_2__PI;
        if (n_theta < 0) n_theta = -n_theta;
    }
    return n_theta <= prec;
}

static inline double angle_diff(double a, double b)
{
    double diff = a - b;
    while (diff <= -kPI) diff += M_2__PI;
    while (diff > kPI) diff -= M_2__PI;
    return std::fabs(diff);
}

static inline double angle_diff_signed(double a, double b)
{
    double diff = a - b;
    while (diff <= -kPI) diff += M_2__PI;
    while (diff > kPI) diff -= M_2__PI;
    return diff;
}
static inline double distSq(double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); }
static inline double dist(double x1, double y1, double x2, double y2) { return std::sqrt(distSq(x1, y1, x2, y2)); }

struct Rect { double x1, y1, x2, y2, width, x, y, theta, dx, dy, prec, p; };

// cv LineSegmentDetectorImpl::region_grow (lsd.cpp): 8-neighbourhood growth from `addr0`, running region angle from float sums
// convention C.6 (include/orbline_types.h conv_libm_float): which overload the unqualified libm calls on float arguments resolve to.  Set per thread by
// the entry points that carry the parameter block (lsd_detect, line_extract) or take the flag (orc_lbd_compute_conv); 0 = the C functions on doubles.
static thread_local int t_libm_float = 0;
struct LibmScope { int prev; explicit LibmScope(int v) : prev(t_libm_float) { t_libm_float = v; } ~LibmScope() { t_libm_float = prev; } };

// (t_grow_iters / t_grow_first: statistics for tools/ only -- the number of batches of <= 8 FIFO entries the device agent's schedule needs for this
// region, and the region's size after its first FIFO entry)
static thread_local int t_grow_iters = 0, t_grow_first = 0, t_grow_itersW[3] = {0, 0, 0}, t_grow_prefW[3] = {0, 0, 0};
static thread_local std::vector<int> t_size_at;
static void region_grow(LsdState& S, int addr0, std::vector<RegionPoint>& reg, double& reg_angle, double prec)
{
    const int W = S.w, H = S.h;
    reg.clear();
    t_grow_iters = 0; t_grow_first = 0;
    size_t batch_end = 0;
    t_size_at.clear();
    reg_angle = S.angles[addr0];
    reg.push_back({addr0 % W, addr0 / W, reg_angle, S.modgrad[addr0]});
    float sumdx = float(std::cos(reg_angle));
    float sumdy = float(std::sin(reg_angle));
    S.used[addr0] = 1;
    for (size_t i = 0; i < reg.size(); ++i) {
        if (i == batch_end) { ++t_grow_iters; batch_end = i + std::min<size_t>(8, reg.size() - i); }
        if (i == 1) t_grow_first = (int)reg.size();
        t_size_at.push_back((int)reg.size());
        const int rx = reg[i].x, ry = reg[i].y;
        const int xx_min = std::max(rx - 1, 0), xx_max = std::min(rx + 1, W - 1);
        const int yy_min = std::max(ry - 1, 0), yy_max = std::min(ry + 1, H - 1);
        for (int yy = yy_min; yy <= yy_max; ++yy) {
            int c_addr = xx_min + yy * W;
            for (int xx = xx_min; xx <= xx_max; ++xx, ++c_addr) {
                if (!S.used[c_addr] && is_aligned(S, c_addr, reg_angle, prec)) {
                    S.used[c_addr] = 1;
                    const double angle = S.angles[c_addr];
                    reg.push_back({xx, yy, angle, S.modgrad[c_addr]});
                    if (t_libm_float) { sumdx += cosf(float(angle)); sumdy += sinf(float(angle)); }      // convention C.6, float overloads
                    else {
                        sumdx += std::cos((double)float(angle));   // convention C.6: double libm, rounded by the float +=
                        sumdy += std::sin((double)float(angle));
                    }
                    reg_angle = fastAtan2(sumdy, sumdx) * DEG_TO_RADS;
                }
            }
        }
    }
    // a window phase that handles the leading FIFO entries within Chebyshev distance D of the seed without a gather: the batches left to the general loop
    for (int D = 1; D <= 3; ++D) {
        size_t i0 = 0;
        while (i0 < reg.size() && std::max(std::abs(reg[i0].x - reg[0].x), std::abs(reg[i0].y - reg[0].y)) <= D) ++i0;
        int it = 0;
        for (size_t i = i0; i < reg.size(); ) { ++it; i += std::min<size_t>(8, t_size_at[i] - i); }
        t_grow_itersW[D - 1] = it; t_grow_prefW[D - 1] = (int)i0;
    }
}

// cv LineSegmentDetectorImpl::region2rect + get_theta
static void region2rect(const std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec)
{
    double x = 0, y = 0, sum = 0;
    for (const RegionPoint& q : reg) {
        const double weight = q.modgrad;
        x += double(q.x) * weight;
        y += double(q.y) * weight;
        sum += weight;
    }
    x /= sum; y /= sum;
    double Ixx = 0, Iyy = 0, Ixy = 0;
    for (const RegionPoint& q : reg) {
        const double dx = double(q.x) - x, dy = double(q.y) - y, weight = q.modgrad;
        Ixx += dy * dy * weight;
        Iyy += dx * dx * weight;
        Ixy -= dx * dy * weight;
    }
    const double lambda = 0.5 * (Ixx + Iyy - std::sqrt((Ixx - Iyy) * (Ixx - Iyy) + 4.0 * Ixy * Ixy));
    double theta = (std::fabs(Ixx) > std::fabs(Iyy)) ? double(fastAtan2(float(lambda - Ixx), float(Ixy)))
                                                     : double(fastAtan2(float(Ixy), float(lambda - Iyy)));
    theta *= DEG_TO_RADS;
    if (angle_diff(theta, reg_angle) > prec) theta += kPI;
    const double dx = std::cos(theta), dy = std::sin(theta);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (const RegionPoint& q : reg) {
        const double regdx = double(q.x) - x, regdy = double(q.y) - y;
        const double l = regdx * dx + regdy * dy;
        const double w = -regdx * dy + regdy * dx;
        if (l > l_max) l_max = l;
        else if (l < l_min) l_min = l;
        if (w > w_max) w_max = w;
        else if (w < w_min) w_min = w;
    }
    rec.x1 = x + l_min * dx; rec.y1 = y + l_min * dy; rec.x2 = x + l_max * dx; rec.y2 = y + l_max * dy;
    rec.width = w_max - w_min;
    rec.x = x; rec.y = y; rec.theta = theta; rec.dx = dx; rec.dy = dy; rec.prec = prec; rec.p = p;
    if (rec.width < 1.0) rec.width = 1.0;
}

// ---- LSD_REFINE_STD / LSD_REFINE_ADV (convention C.14: restated from OpenCV 3.4's modules/imgproc/src/lsd.cpp as recalled -- the file is not in
// /root/reference, like the rest of OpenCV; unpinned).  reduce_region_radius, refine, rect_nfa (with its integer edge steps and the
// tailp->p.x comparisons the original has), nfa, log_gamma, rect_improve.
static bool reduce_region_radius(LsdState& S, std::vector<RegionPoint>& reg, double reg_angle, double prec, double p, Rect& rec, double density, double density_th)
{
    const double xc = double(reg[0].x), yc = double(reg[0].y);
    const double radSq1 = distSq(xc, yc, rec.x1, rec.y1), radSq2 = distSq(xc, yc, rec.x2, rec.y2);
    double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
    while (density < density_th) {
        radSq *= 0.75 * 0.75;
        for (size_t i = 0; i < reg.size(); ++i) {
            if (distSq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) {
                S.used[(size_t)reg[i].y * S.w + reg[i].x] = 0;
                std::swap(reg[i], reg[reg.size() - 1]);
                reg.pop_back();
                --i;
            }
        }
        if (reg.size() < 2) return false;
        region2rect(reg, reg_angle, prec, p, rec);
        density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    }
    return true;
}

static bool refine(LsdState& S, std::vector<RegionPoint>& reg, double& reg_angle, double prec, double p, Rect& rec, double density_th)
{
    double density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density >= density_th) return true;
    const double xc = double(reg[0].x), yc = double(reg[0].y);
    const double ang_c = reg[0].angle;
    double sum = 0, s_sum = 0;
    int n = 0;
    for (size_t i = 0; i < reg.size(); ++i) {
        S.used[(size_t)reg[i].y * S.w + reg[i].x] = 0;
        if (dist(xc, yc, reg[i].x, reg[i].y) < rec.width) {
            const double ang_d = angle_diff_signed(reg[i].angle, ang_c);
            sum += ang_d;
            s_sum += ang_d * ang_d;
            ++n;
        }
    }
    const double mean_angle = sum / double(n);
    const double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
    region_grow(S, reg[0].y * S.w + reg[0].x, reg, reg_angle, tau);
    if (reg.size() < 2) return false;
    region2rect(reg, reg_angle, prec, p, rec);
    density = double(reg.size()) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
    if (density < density_th) return reduce_region_radius(S, reg, reg_angle, prec, p, rec, density, density_th);
    return true;
}

static inline bool double_equal(double a, double b)
{
    if (a == b) return true;
    const double abs_diff = std::fabs(a - b), aa = std::fabs(a), bb = std::fabs(b);
    double abs_max = (aa > bb) ? aa : bb;
    if (abs_max < DBL_MIN) abs_max = DBL_MIN;
    return (abs_diff / abs_max) <= (100.0 * DBL_EPSILON);
}
static double log_gamma_lanczos(double x)
{
    static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
    double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5);
    double b = 0;
    for (int n = 0; n < 7; ++n) {
        a -= std::log(x + double(n));
        b += q[n] * std::pow(x, double(n));
    }
    return a + std::log(b);
}
static double log_gamma_windschitl(double x)
{
    return 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0)));
}
static inline double log_gamma(double x) { return x > 15.0 ? log_gamma_windschitl(x) : log_gamma_lanczos(x); }

static double nfa(int n, int k, double p, double LOG_NT)
{
    if (n == 0 || k == 0) return -LOG_NT;
    if (n == k) return -LOG_NT - double(n) * std::log10(p);
    const double p_term = p / (1 - p);
    const double log1term = log_gamma(double(n) + 1) - log_gamma(double(k) + 1) - log_gamma(double(n - k) + 1) + double(k) * std::log(p) +
                            double(n - k) * std::log(1.0 - p);
    double term = std::exp(log1term);
    if (double_equal(term, 0)) {
        if (k > n * p) return -log1term / 2.30258509299404568402 - LOG_NT;      // M_LN10
        return -LOG_NT;
    }
    double bin_tail = term;
    const double tolerance = 0.1;
    for (int i = k + 1; i <= n; ++i) {
        const double bin_term = double(n - i + 1) / double(i);
        const double mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1) {
            const double err = term * ((1 - std::pow(mult_term, double(n - i + 1))) / (1 - mult_term) - 1);
            if (err < tolerance * std::fabs(-std::log10(bin_tail) - LOG_NT) * bin_tail) break;
        }
    }
    return -std::log10(bin_tail) - LOG_NT;
}

static bool is_aligned_xy(const LsdState& S, int x, int y, double theta, double prec) { return is_aligned(S, y * S.w + x, theta, prec); }

static double rect_nfa(const LsdState& S, const Rect& rec, double LOG_NT)
{
    struct Edge { int x, y; bool taken; };
    int total_pts = 0, alg_pts = 0;
    const double half_width = rec.width / 2.0;
    const double dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
    Edge ordered_x[4];
    Edge* min_y = &ordered_x[0];
    Edge* max_y = &ordered_x[0];
    ordered_x[0] = {int(rec.x1 - dyhw), int(rec.y1 + dxhw), false};
    ordered_x[1] = {int(rec.x2 - dyhw), int(rec.y2 + dxhw), false};
    ordered_x[2] = {int(rec.x2 + dyhw), int(rec.y2 - dxhw), false};
    ordered_x[3] = {int(rec.x1 + dyhw), int(rec.y1 - dxhw), false};
    std::sort(ordered_x, ordered_x + 4, [](const Edge& a, const Edge& b) { return a.x == b.x ? a.y < b.y : a.x < b.x; });
    for (unsigned i = 1; i < 4; ++i) {
        if (min_y->y > ordered_x[i].y) min_y = &ordered_x[i];
        if (max_y->y < ordered_x[i].y) max_y = &ordered_x[i];
    }
    min_y->taken = true;
    Edge* leftmost = nullptr;
    for (unsigned i = 0; i < 4; ++i)
        if (!ordered_x[i].taken) {
            if (!leftmost) leftmost = &ordered_x[i];
            else if (leftmost->x > ordered_x[i].x) leftmost = &ordered_x[i];
        }
    leftmost->taken = true;
    Edge* rightmost = nullptr;
    for (unsigned i = 0; i < 4; ++i)
        if (!ordered_x[i].taken) {
            if (!rightmost) rightmost = &ordered_x[i];
            else if (rightmost->x < ordered_x[i].x) rightmost = &ordered_x[i];
        }
    rightmost->taken = true;
    Edge* tailp = nullptr;
    for (unsigned i = 0; i < 4; ++i)
        if (!ordered_x[i].taken) {
            if (!tailp) tailp = &ordered_x[i];
            else if (tailp->x > ordered_x[i].x) tailp = &ordered_x[i];
        }
    tailp->taken = true;
    // (integer divisions and the tailp->x comparisons: as in the original)
    const double flstep = (min_y->y != leftmost->y) ? (min_y->x - leftmost->x) / (min_y->y - leftmost->y) : 0;
    const double slstep = (leftmost->y != tailp->x) ? (leftmost->x - tailp->x) / (leftmost->y - tailp->x) : 0;
    const double frstep = (min_y->y != rightmost->y) ? (min_y->x - rightmost->x) / (min_y->y - rightmost->y) : 0;
    const double srstep = (rightmost->y != tailp->x) ? (rightmost->x - tailp->x) / (rightmost->y - tailp->x) : 0;
    double lstep = flstep, rstep = frstep;
    double left_x = min_y->x, right_x = min_y->x;
    const int min_iter = min_y->y, max_iter = max_y->y;
    for (int y = min_iter; y <= max_iter; ++y) {
        if (y < 0 || y >= S.h) continue;
        for (int x = int(left_x); x <= int(right_x); ++x) {
            if (x < 0 || x >= S.w) continue;
            ++total_pts;
            if (is_aligned_xy(S, x, y, rec.theta, rec.prec)) ++alg_pts;
        }
        if (y >= leftmost->y) lstep = slstep;
        if (y >= rightmost->y) rstep = srstep;
        left_x += lstep;
        right_x += rstep;
    }
    return nfa(total_pts, alg_pts, rec.p, LOG_NT);
}

static double rect_improve(const LsdState& S, Rect& rec, double LOG_NT, double LOG_EPS)
{
    const double delta = 0.5, delta_2 = delta / 2.0;
    double log_nfa = rect_nfa(S, rec, LOG_NT);
    if (log_nfa > LOG_EPS) return log_nfa;
    Rect r = rec;
    for (int n = 0; n < 5; ++n) {
        r.p /= 2;
        r.prec = r.p * kPI;
        const double log_nfa_new = rect_nfa(S, r, LOG_NT);
        if (log_nfa_new > log_nfa) { log_nfa = log_nfa_new; rec = r; }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (unsigned n = 0; n < 5; ++n)
        if ((r.width - delta) >= 0.5) {
            r.width -= delta;
            const double log_nfa_new = rect_nfa(S, r, LOG_NT);
            if (log_nfa_new > log_nfa) { rec = r; log_nfa = log_nfa_new; }
        }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (unsigned n = 0; n < 5; ++n)
        if ((r.width - delta) >= 0.5) {
            r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2;
            r.width -= delta;
            const double log_nfa_new = rect_nfa(S, r, LOG_NT);
            if (log_nfa_new > log_nfa) { rec = r; log_nfa = log_nfa_new; }
        }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (unsigned n = 0; n < 5; ++n)
        if ((r.width - delta) >= 0.5) {
            r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2;
            r.width -= delta;
            const double log_nfa_new = rect_nfa(S, r, LOG_NT);
            if (log_nfa_new > log_nfa) { rec = r; log_nfa = log_nfa_new; }
        }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (unsigned n = 0; n < 5; ++n)
        if ((r.width - delta) >= 0.5) {
            r.p /= 2;
            r.prec = r.p * kPI;
            const double log_nfa_new = rect_nfa(S, r, LOG_NT);
            if (log_nfa_new > log_nfa) { rec = r; log_nfa = log_nfa_new; }
        }
    return log_nfa;
}

// cv::LineSegmentDetector::detect (refine = P.lsd_refine: 0 NONE, 1 STD, 2 ADV).  scaled_out (optional) receives the
// blurred + upsampled image, for stage-wise comparison.
void lsd_detect(const Image& image, const olf_line_params& P, std::vector<Vec4f>& lines, Image* scaled_out, std::vector<int>* region_sizes)
{
    LibmScope libm_scope(P.conv_libm_float);
    lines.clear();
    const double SCALE = P.lsd_scale, SIGMA_SCALE = P.lsd_sigma_scale, QUANT = P.lsd_quant, ANG_TH = P.lsd_ang_th;
    const int N_BINS = P.lsd_n_bins;
    const double prec = kPI * ANG_TH / 180, p = ANG_TH / 180;
    const double rho = QUANT / std::sin(prec);
    Image scaled;
    if (SCALE != 1) {
        const double sigma = (SCALE < 1) ? (SIGMA_SCALE / SCALE) : SIGMA_SCALE;
        const double sprec = 3;
        const unsigned int hk = (unsigned int)std::ceil(sigma * std::sqrt(2 * sprec * std::log(10.0)));
        Image g = gaussian_blur_u8(image, gaussian_taps_q8(1 + 2 * hk, sigma, P.conv_gauss_sum256));
        // resize(gaussian_img, scaled_image, Size(), SCALE, SCALE): dsize = round(size*SCALE), scale = 1/SCALE
        const int dw = cvRound(image.w * SCALE), dh = cvRound(image.h * SCALE);
        scaled = P.conv_resize_exact ? resize_linear_exact_u8(g, dw, dh, 1. / SCALE, 1. / SCALE) : resize_linear_u8(g, dw, dh, 1. / SCALE, 1. / SCALE);
    } else scaled = image;
    if (scaled_out) *scaled_out = scaled;
    LsdState S;
    const int W = S.w = scaled.w, H = S.h = scaled.h;
    S.angles.assign((size_t)W * H, NOTDEF);
    S.modgrad.assign((size_t)W * H, 0.0);
    // ---- ll_angle
    double max_grad = -1;
    for (int y = 0; y < H - 1; ++y) {
        const uint8_t* r0 = scaled.row(y);
        const uint8_t* r1 = scaled.row(y + 1);
        for (int x = 0; x < W - 1; ++x) {
            int DA = r1[x + 1] - r0[x];
            int BC = r0[x + 1] - r1[x];
            int gx = DA + BC, gy = DA - BC;
            double norm = std::sqrt((gx * gx + gy * gy) / 4.0);
            S.modgrad[(size_t)y * W + x] = norm;
            if (norm <= rho) S.angles[(size_t)y * W + x] = NOTDEF;
            else {
                S.angles[(size_t)y * W + x] = fastAtan2(float(gx), float(-gy)) * DEG_TO_RADS;
                if (norm > max_grad) max_grad = norm;
            }
        }
    }
    // ---- pseudo-ordering: bins of the gradient norm, high to low, raster order inside a bin
    {
        const double bin_coef = (max_grad > 0) ? double(N_BINS - 1) / max_grad : 0;
        std::vector<int> cnt(N_BINS + 1, 0);
        std::vector<int> bin((size_t)W * H, -1);
        for (int y = 0; y < H - 1; ++y)
            for (int x = 0; x < W - 1; ++x) {
                int i = int(S.modgrad[(size_t)y * W + x] * bin_coef);
                bin[(size_t)y * W + x] = i;
                ++cnt[i];
            }
        std::vector<int> start(N_BINS + 1, 0);
        int acc = 0;
        for (int b = N_BINS - 1; b >= 0; --b) { start[b] = acc; acc += cnt[b]; }
        S.order.assign(acc, 0);
        if (P.conv_seed_order == 1) {
            // convention C.9, variant 1: OpenCV >= 3.3 pushes every pixel (x < w-1, y < h-1) as {point, bin} in raster order and calls
            // std::sort(ordered_points.begin(), ordered_points.end(), compare_norm) with compare_norm(a, b) = a.norm > b.norm -- an unstable
            // sort: the order inside a bin is libstdc++'s introsort order.  Same library, same sequence, same comparator here.
            struct NormPoint { int addr, norm; };
            std::vector<NormPoint> pts;
            pts.reserve(acc);
            for (int y = 0; y < H - 1; ++y)
                for (int x = 0; x < W - 1; ++x) pts.push_back({y * W + x, bin[(size_t)y * W + x]});
            std::sort(pts.begin(), pts.end(), [](const NormPoint& a, const NormPoint& b) { return a.norm > b.norm; });
            for (size_t i = 0; i < pts.size(); ++i) S.order[i] = pts[i].addr;
        } else {
            for (int y = 0; y < H - 1; ++y)
                for (int x = 0; x < W - 1; ++x) {
                    int b = bin[(size_t)y * W + x];
                    S.order[start[b]++] = y * W + x;
                }
        }
    }
    const double LOG_NT = 5 * (std::log10(double(W)) + std::log10(double(H))) / 2 + std::log10(11.0);
    const int min_reg_size = int(-LOG_NT / std::log10(p));
    S.used.assign((size_t)W * H, 0);
    std::vector<RegionPoint> reg;
    for (size_t oi = 0; oi < S.order.size(); ++oi) {
        const int addr0 = S.order[oi];
        if (S.used[addr0] || S.angles[addr0] == NOTDEF) continue;
        double reg_angle;
        region_grow(S, addr0, reg, reg_angle, prec);
        if (region_sizes) { region_sizes->push_back((int)reg.size()); region_sizes->push_back(t_grow_iters); region_sizes->push_back(reg.size() > 1 ? t_grow_first : 1);
                            for (int D = 0; D < 3; ++D) { region_sizes->push_back(t_grow_itersW[D]); region_sizes->push_back(t_grow_prefW[D]); } }
        if ((int)reg.size() < min_reg_size) continue;
        Rect rec;
        region2rect(reg, reg_angle, prec, p, rec);
        if (P.lsd_refine > 0) {
            if (!refine(S, reg, reg_angle, prec, p, rec, P.lsd_density_th)) continue;
            if (P.lsd_refine >= 2) {
                const double log_nfa = rect_improve(S, rec, LOG_NT, P.lsd_log_eps);
                if (log_nfa <= P.lsd_log_eps) continue;
            }
        }
        double x1 = rec.x1, y1 = rec.y1, x2 = rec.x2, y2 = rec.y2;
        x1 += 0.5; y1 += 0.5; x2 += 0.5; y2 += 0.5;
        if (SCALE != 1) { x1 /= SCALE; y1 /= SCALE; x2 /= SCALE; y2 /= SCALE; }
        lines.push_back({{float(x1), float(y1), float(x2), float(y2)}});
    }
}

// LSDDetectorC::detectImpl (opts overload), 1 octave: Vec4f -> KeyLine
void make_keylines(const std::vector<Vec4f>& segs, int cols, int rows, double min_length, std::vector<olf_keyline>& out)
{
    out.clear();
    int class_counter = -1;
    for (const Vec4f& s : segs) {
        float e[4] = {s.v[0], s.v[1], s.v[2], s.v[3]};
        // checkLineExtremes
        if (e[0] < 0) e[0] = 0;
        if (e[0] >= cols) e[0] = (float)cols - 1.0f;
        if (e[2] < 0) e[2] = 0;
        if (e[2] >= cols) e[2] = (float)cols - 1.0f;
        if (e[1] < 0) e[1] = 0;
        if (e[1] >= rows) e[1] = (float)rows - 1.0f;
        if (e[3] < 0) e[3] = 0;
        if (e[3] >= rows) e[3] = (float)rows - 1.0f;
        const double length = (float)std::sqrt(std::pow((double)(e[0] - e[2]), 2) + std::pow((double)(e[1] - e[3]), 2));
        if (!(length > min_length)) continue;
        olf_keyline kl;
        const float octaveScale = 1.0f;   // pow((float)scale, 0)
        kl.startPointX = e[0] * octaveScale; kl.startPointY = e[1] * octaveScale;
        kl.endPointX = e[2] * octaveScale; kl.endPointY = e[3] * octaveScale;
        kl.sPointInOctaveX = e[0]; kl.sPointInOctaveY = e[1]; kl.ePointInOctaveX = e[2]; kl.ePointInOctaveY = e[3];
        kl.lineLength = (float)length;
        // cv::LineIterator(img, Point2f, Point2f).count : end points rounded (cvRound), 8-connected
        const int x1 = cvRoundf(e[0]), y1 = cvRoundf(e[1]), x2 = cvRoundf(e[2]), y2 = cvRoundf(e[3]);
        kl.numOfPixels = std::max(std::abs(x2 - x1), std::abs(y2 - y1)) + 1;
        kl.angle = t_libm_float ? atan2f(kl.endPointY - kl.startPointY, kl.endPointX - kl.startPointX)      // convention C.6
                                : (float)std::atan2((double)(kl.endPointY - kl.startPointY), (double)(kl.endPointX - kl.startPointX));
        kl.class_id = ++class_counter;
        kl.octave = 0;
        kl.size = (kl.endPointX - kl.startPointX) * (kl.endPointY - kl.startPointY);
        kl.response = kl.lineLength / std::max(cols, rows);
        kl.pt_x = (kl.endPointX + kl.startPointX) / 2; kl.pt_y = (kl.endPointY + kl.startPointY) / 2;
        out.push_back(kl);
    }
}

// ---- LBD ----------------------------------------------------------------------------------------
static const int8_t kBandPairs[64] = {
#include "lbd_band_pairs.inc"
};
static const int NUM_OF_BANDS = 9, WIDTH_OF_BAND = 7;

// cv::Sobel(src, dst, CV_16S, dx, dy, 3) with BORDER_REFLECT_101 (App. A.9)
static void sobel3(const Image& src, std::vector<int16_t>& dxI, std::vector<int16_t>& dyI)
{
    const int w = src.w, h = src.h;
    dxI.assign((size_t)w * h, 0); dyI.assign((size_t)w * h, 0);
    for (int y = 0; y < h; ++y) {
        const uint8_t* r0 = src.row(reflect101(y - 1, h));
        const uint8_t* r1 = src.row(y);
        const uint8_t* r2 = src.row(reflect101(y + 1, h));
        for (int x = 0; x < w; ++x) {
            const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            dxI[(size_t)y * w + x] = (int16_t)((r0[xp] - r0[xm]) + 2 * (r1[xp] - r1[xm]) + (r2[xp] - r2[xm]));
            dyI[(size_t)y * w + x] = (int16_t)((r2[xm] + 2 * r2[x] + r2[xp]) - (r0[xm] + 2 * r0[x] + r0[xp]));
        }
    }
}

void lbd_compute(const Image& image, const std::vector<olf_keyline>& keylines, std::vector<uint8_t>& desc,
                 std::vector<float>* float_desc, int conv_gauss_sum256 = 0)
{
    const int n = (int)keylines.size();
    desc.assign((size_t)n * 32, 0);
    if (float_desc) float_desc->assign((size_t)n * 72, 0.f);
    if (n == 0) return;   // "Error: keypoint list is empty" + return (binary_descriptor_custom.cpp:556-560)
    // weights, binary_descriptor_custom.cpp:217-259 (integer divisions are the reference's)
    std::vector<double> gaussCoefL(WIDTH_OF_BAND * 3), gaussCoefG(NUM_OF_BANDS * WIDTH_OF_BAND);
    {
        double u = (WIDTH_OF_BAND * 3 - 1) / 2;
        double sigma = (WIDTH_OF_BAND * 2 + 1) / 2;
        double invsigma2 = -1 / (2 * sigma * sigma);
        for (int i = 0; i < WIDTH_OF_BAND * 3; ++i) { double dis = i - u; gaussCoefL[i] = std::exp(dis * dis * invsigma2); }
        u = (NUM_OF_BANDS * WIDTH_OF_BAND - 1) / 2;
        sigma = u;
        invsigma2 = -1 / (2 * sigma * sigma);
        for (int i = 0; i < NUM_OF_BANDS * WIDTH_OF_BAND; ++i) { double dis = i - u; gaussCoefG[i] = std::exp(dis * dis * invsigma2); }
    }
    Image blurred = gaussian_blur_u8(image, gaussian_taps_q8(5, 1.0, conv_gauss_sum256));
    std::vector<int16_t> dxImg, dyImg;
    sobel3(blurred, dxImg, dyImg);
    const short heightOfLSP = (short)(WIDTH_OF_BAND * NUM_OF_BANDS);
    const short realWidth = (short)image.w, imageWidth = (short)(realWidth - 1), imageHeight = (short)(image.h - 1);
    for (int li = 0; li < n; ++li) {
        const olf_keyline& kl = keylines[li];
        float pgdLBandSum[9] = {0}, ngdLBandSum[9] = {0}, pgdL2BandSum[9] = {0}, ngdL2BandSum[9] = {0};
        float pgdOBandSum[9] = {0}, ngdOBandSum[9] = {0}, pgdO2BandSum[9] = {0}, ngdO2BandSum[9] = {0};
        const short lengthOfLSP = (short)kl.numOfPixels;
        const short halfWidth = (short)((lengthOfLSP - 1) / 2);
        const short halfHeight = (short)((heightOfLSP - 1) / 2);
        const float lineMiddlePointX = (float)(0.5 * (kl.sPointInOctaveX + kl.ePointInOctaveX));
        const float lineMiddlePointY = (float)(0.5 * (kl.sPointInOctaveY + kl.ePointInOctaveY));
        float dL[2], dO[2];
        if (t_libm_float) { dL[0] = cosf(kl.angle); dL[1] = sinf(kl.angle); }      // osl.direction = kl.angle; convention C.6
        else { dL[0] = (float)std::cos((double)kl.angle); dL[1] = (float)std::sin((double)kl.angle); }
        dO[0] = -dL[1]; dO[1] = dL[0];
        float sCorX0 = -dL[0] * halfWidth + dL[1] * halfHeight + lineMiddlePointX;
        float sCorY0 = -dL[1] * halfWidth - dL[0] * halfHeight + lineMiddlePointY;
        for (short hID = 0; hID < heightOfLSP; ++hID) {
            float sCorX = sCorX0, sCorY = sCorY0;
            float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
            for (short wID = 0; wID < lengthOfLSP; ++wID) {
                short tempCor = (short)std::round(sCorX);
                const short xCor = (tempCor < 0) ? 0 : (tempCor > imageWidth) ? imageWidth : tempCor;
                tempCor = (short)std::round(sCorY);
                const short yCor = (tempCor < 0) ? 0 : (tempCor > imageHeight) ? imageHeight : tempCor;
                const short dx = dxImg[yCor * realWidth + xCor], dy = dyImg[yCor * realWidth + xCor];
                const float gDL = dx * dL[0] + dy * dL[1];
                const float gDO = dx * dO[0] + dy * dO[1];
                if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
                if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
                sCorX += dL[0];
                sCorY += dL[1];
            }
            sCorX0 -= dL[1];
            sCorY0 += dL[0];
            float coefInGaussion = (float)gaussCoefG[hID];
            pgdLRowSum = coefInGaussion * pgdLRowSum;
            ngdLRowSum = coefInGaussion * ngdLRowSum;
            const float pgdL2RowSum = pgdLRowSum * pgdLRowSum, ngdL2RowSum = ngdLRowSum * ngdLRowSum;
            pgdORowSum = coefInGaussion * pgdORowSum;
            ngdORowSum = coefInGaussion * ngdORowSum;
            const float pgdO2RowSum = pgdORowSum * pgdORowSum, ngdO2RowSum = ngdORowSum * ngdORowSum;
            auto add = [&](short bandID, float c) {
                pgdLBandSum[bandID] += c * pgdLRowSum;
                ngdLBandSum[bandID] += c * ngdLRowSum;
                pgdL2BandSum[bandID] += c * c * pgdL2RowSum;
                ngdL2BandSum[bandID] += c * c * ngdL2RowSum;
                pgdOBandSum[bandID] += c * pgdORowSum;
                ngdOBandSum[bandID] += c * ngdORowSum;
                pgdO2BandSum[bandID] += c * c * pgdO2RowSum;
                ngdO2BandSum[bandID] += c * c * ngdO2RowSum;
            };
            short bandID = (short)(hID / WIDTH_OF_BAND);
            add(bandID, (float)gaussCoefL[hID % WIDTH_OF_BAND + WIDTH_OF_BAND]);
            bandID--;
            if (bandID >= 0) add(bandID, (float)gaussCoefL[hID % WIDTH_OF_BAND + 2 * WIDTH_OF_BAND]);
            bandID = (short)(bandID + 2);
            if (bandID < NUM_OF_BANDS) add(bandID, (float)gaussCoefL[hID % WIDTH_OF_BAND]);
        }
        float desVec[72];
        const float invN2 = (float)(1.0 / (WIDTH_OF_BAND * 2.0)), invN3 = (float)(1.0 / (WIDTH_OF_BAND * 3.0));
        for (short bandID = 0; bandID < NUM_OF_BANDS; ++bandID) {
            const float invN = (bandID == 0 || bandID == NUM_OF_BANDS - 1) ? invN2 : invN3;
            const short desID = (short)(bandID * 8);
            float temp = pgdLBandSum[bandID] * invN;
            desVec[desID] = temp;
            desVec[desID + 4] = (float)std::sqrt((double)(pgdL2BandSum[bandID] * invN - temp * temp));
            temp = ngdLBandSum[bandID] * invN;
            desVec[desID + 1] = temp;
            desVec[desID + 5] = (float)std::sqrt((double)(ngdL2BandSum[bandID] * invN - temp * temp));
            temp = pgdOBandSum[bandID] * invN;
            desVec[desID + 2] = temp;
            desVec[desID + 6] = (float)std::sqrt((double)(pgdO2BandSum[bandID] * invN - temp * temp));
            temp = ngdOBandSum[bandID] * invN;
            desVec[desID + 3] = temp;
            desVec[desID + 7] = (float)std::sqrt((double)(ngdO2BandSum[bandID] * invN - temp * temp));
        }
        float tempM = 0, tempS = 0;
        for (int b = 0; b < NUM_OF_BANDS; ++b) {
            const float* d = desVec + 8 * b;
            tempM += d[0] * d[0]; tempM += d[1] * d[1]; tempM += d[2] * d[2]; tempM += d[3] * d[3];
            tempS += d[4] * d[4]; tempS += d[5] * d[5]; tempS += d[6] * d[6]; tempS += d[7] * d[7];
        }
        if (t_libm_float) { tempM = 1 / sqrtf(tempM); tempS = 1 / sqrtf(tempS); }      // convention C.6: float sqrt, float divide
        else {
            tempM = (float)(1 / std::sqrt((double)tempM));   // convention C.6: double sqrt, double divide
            tempS = (float)(1 / std::sqrt((double)tempS));
        }
        for (int b = 0; b < NUM_OF_BANDS; ++b) {
            float* d = desVec + 8 * b;
            d[0] = d[0] * tempM; d[1] = d[1] * tempM; d[2] = d[2] * tempM; d[3] = d[3] * tempM;
            d[4] = d[4] * tempS; d[5] = d[5] * tempS; d[6] = d[6] * tempS; d[7] = d[7] * tempS;
        }
        for (int i = 0; i < 72; ++i)
            if (desVec[i] > 0.4) desVec[i] = (float)0.4;
        float temp = 0;
        for (int i = 0; i < 72; ++i) temp += desVec[i] * desVec[i];
        temp = t_libm_float ? 1 / sqrtf(temp) : (float)(1 / std::sqrt((double)temp));
        for (int i = 0; i < 72; ++i) desVec[i] = desVec[i] * temp;
        if (float_desc) std::memcpy(&(*float_desc)[(size_t)li * 72], desVec, sizeof(desVec));
        // binaryConversion over the 32 band pairs
        for (int c = 0; c < 32; ++c) {
            const float* f1 = &desVec[8 * kBandPairs[2 * c]];
            const float* f2 = &desVec[8 * kBandPairs[2 * c + 1]];
            uint8_t r = 0;
            for (int i = 0; i < 8; ++i)
                if (f1[i] > f2[i]) r = (uint8_t)(r + (1 << i));
            desc[(size_t)li * 32 + c] = r;
        }
    }
}

struct sort_lines_by_response {
    bool operator()(const olf_keyline& a, const olf_keyline& b) const { return a.response > b.response; }
};

// Lineextractor::operator(): detect, top-N by response, LBD.  use_std_sort: the reference's unstable
// std::sort (C.3) instead of the stable convention.
void line_extract(const Image& img, const olf_line_params& P, bool use_std_sort, std::vector<olf_keyline>& kls, std::vector<uint8_t>& desc,
                  std::vector<olf_keyline>* all_detected)
{
    LibmScope libm_scope(P.conv_libm_float);
    std::vector<Vec4f> segs;
    const auto t0 = std::chrono::steady_clock::now();
    lsd_detect(img, P, segs, nullptr, nullptr);
    g_line_stage_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();      // LSD
    const auto t1 = std::chrono::steady_clock::now();
    const double min_length = P.min_line_length * std::min(img.w, img.h);
    make_keylines(segs, img.w, img.h, min_length, kls);
    if (all_detected) *all_detected = kls;
    if ((int)kls.size() > P.lsd_nfeatures && P.lsd_nfeatures != 0) {
        if (use_std_sort) std::sort(kls.begin(), kls.end(), sort_lines_by_response());
        else std::stable_sort(kls.begin(), kls.end(), sort_lines_by_response());
        kls.resize(P.lsd_nfeatures);
        for (int i = 0; i < P.lsd_nfeatures; ++i) kls[i].class_id = i;
    }
    lbd_compute(img, kls, desc, nullptr, P.conv_gauss_sum256);
    g_line_stage_ms[1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();      // key lines, top-N, LBD
}

}  // namespace orc

using namespace orc;
// test hook for the OpenCV fixtures (opencv34_lineiterator.npz): cv::LineIterator(img, Point2f, Point2f).count as make_keylines computes it
extern "C" int orc_line_iterator_count(float x1f, float y1f, float x2f, float y2f, int /*w*/, int /*h*/)
{
    const int x1 = orc::cvRoundf(x1f), y1 = orc::cvRoundf(y1f), x2 = orc::cvRoundf(x2f), y2 = orc::cvRoundf(y2f);
    return std::max(std::abs(x2 - x1), std::abs(y2 - y1)) + 1;
}

extern "C" {

// raw LSD segments (x1,y1,x2,y2 floats) + optional scaled image (dims via sw/sh) + region sizes
// std::sort itself on an array of keys ((field << 22) | payload), field ascending: what convention C.9 variant 1 means for a bin sequence.
// The element type does not enter libstdc++'s algorithm (only comparisons and moves do), so a plain uint32_t array stands for the vector of
// {Point, norm} that ll_angle sorts with compare_norm (norm descending <=> field ascending).
void orc_std_sort_keys(const uint32_t* keys, int n, uint32_t* out)
{
    std::vector<uint32_t> v(keys, keys + n);
    std::sort(v.begin(), v.end(), [](uint32_t a, uint32_t b) { return (a >> 22) < (b >> 22); });
    std::memcpy(out, v.data(), (size_t)n * 4);
}

// The same sort with an explicit depth limit (libstdc++ bits/stl_algo.h restated: __introsort_loop with __move_median_to_first +
// __unguarded_partition, std::partial_sort = the library's own heap sort when the limit is reached, __final_insertion_sort = a stable sort
// of what the loop leaves).  With depth_limit = 2 * floor(log2 n) it must equal orc_std_sort_keys (tests/test_oracle_cpu.py checks that);
// smaller limits reach the heap-sort branch that real images never reach.
void orc_introsort_keys(const uint32_t* keys, int n, int depth_limit, uint32_t* out)
{
    std::vector<uint32_t> A(keys, keys + n);
    auto K = [](uint32_t e) { return e >> 22; };
    auto lt = [&](uint32_t a, uint32_t b) { return K(a) < K(b); };
    struct R { int f, l, d; };
    std::vector<R> stack;
    if (n > 0) stack.push_back({0, n, depth_limit});
    while (!stack.empty()) {
        R r = stack.back(); stack.pop_back();
        int first = r.f, last = r.l, depth = r.d;
        while (last - first > 16) {
            if (depth == 0) { std::partial_sort(A.begin() + first, A.begin() + last, A.begin() + last, lt); break; }
            --depth;
            const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
            if (lt(A[a], A[b])) { if (lt(A[b], A[c])) std::swap(A[first], A[b]); else if (lt(A[a], A[c])) std::swap(A[first], A[c]); else std::swap(A[first], A[a]); }
            else if (lt(A[a], A[c])) std::swap(A[first], A[a]);
            else if (lt(A[b], A[c])) std::swap(A[first], A[c]);
            else std::swap(A[first], A[b]);
            const uint32_t piv = A[first];
            int i = first + 1, j = last;
            for (;;) {
                while (lt(A[i], piv)) ++i;
                --j;
                while (lt(piv, A[j])) --j;
                if (!(i < j)) break;
                std::swap(A[i], A[j]);
                ++i;
            }
            stack.push_back({i, last, depth});
            last = i;
        }
    }
    std::stable_sort(A.begin(), A.end(), lt);
    std::memcpy(out, A.data(), (size_t)n * 4);
}

// The two sorts above on 64-bit keys (field << 32 | payload) -- the keys of the capacity path (csrc/lsd_wide.hip: lsd_n_bins > 1024 or a working image of 2^22
// pixels and more).  full = 0: compared by the field alone (convention C.9 variant 1: what std::sort leaves for ll_angle's {point, norm} vector); full = 1: whole
// words (variant 0: the defined pixels' keys are distinct and ascend with the address inside a field, so any sort gives the stable order).
void orc_std_sort_keys64(const uint64_t* keys, int n, int full, uint64_t* out)
{
    std::vector<uint64_t> v(keys, keys + n);
    if (full) std::sort(v.begin(), v.end());
    else std::sort(v.begin(), v.end(), [](uint64_t a, uint64_t b) { return (a >> 32) < (b >> 32); });
    std::memcpy(out, v.data(), (size_t)n * 8);
}

void orc_introsort_keys64(const uint64_t* keys, int n, int depth_limit, int full, uint64_t* out)
{
    std::vector<uint64_t> A(keys, keys + n);
    auto K = [full](uint64_t e) { return full ? e : (e >> 32); };
    auto lt = [&](uint64_t a, uint64_t b) { return K(a) < K(b); };
    struct R { int f, l, d; };
    std::vector<R> stack;
    if (n > 0) stack.push_back({0, n, depth_limit});
    while (!stack.empty()) {
        R r = stack.back(); stack.pop_back();
        int first = r.f, last = r.l, depth = r.d;
        while (last - first > 16) {
            if (depth == 0) { std::partial_sort(A.begin() + first, A.begin() + last, A.begin() + last, lt); break; }
            --depth;
            const int mid = first + (last - first) / 2, a = first + 1, b = mid, c = last - 1;
            if (lt(A[a], A[b])) { if (lt(A[b], A[c])) std::swap(A[first], A[b]); else if (lt(A[a], A[c])) std::swap(A[first], A[c]); else std::swap(A[first], A[a]); }
            else if (lt(A[a], A[c])) std::swap(A[first], A[a]);
            else if (lt(A[b], A[c])) std::swap(A[first], A[c]);
            else std::swap(A[first], A[b]);
            const uint64_t piv = A[first];
            int i = first + 1, j = last;
            for (;;) {
                while (lt(A[i], piv)) ++i;
                --j;
                while (lt(piv, A[j])) --j;
                if (!(i < j)) break;
                std::swap(A[i], A[j]);
                ++i;
            }
            stack.push_back({i, last, depth});
            last = i;
        }
    }
    std::stable_sort(A.begin(), A.end(), lt);
    std::memcpy(out, A.data(), (size_t)n * 8);
}

int orc_lsd_detect(const uint8_t* img, int w, int h, const olf_line_params* P, float* segs, int cap, int* n, uint8_t* scaled, int* sw, int* sh)
{
    Image im(w, h);
    std::memcpy(im.d.data(), img, (size_t)w * h);
    std::vector<Vec4f> lines;
    Image sc;
    lsd_detect(im, *P, lines, &sc, nullptr);
    *n = (int)lines.size();
    for (int i = 0; i < std::min(*n, cap); ++i) std::memcpy(segs + 4 * i, lines[i].v, 16);
    if (sw) *sw = sc.w;
    if (sh) *sh = sc.h;
    if (scaled) std::memcpy(scaled, sc.d.data(), sc.d.size());
    return *n > cap ? OLF_ERR_CAPACITY : OLF_OK;
}

int orc_line_extract(const uint8_t* img, int w, int h, const olf_line_params* P, int use_std_sort, olf_keyline* kls, uint8_t* desc, int cap, int* n,
                     olf_keyline* all_kls, int all_cap, int* n_all)
{
    Image im(w, h);
    std::memcpy(im.d.data(), img, (size_t)w * h);
    std::vector<olf_keyline> k, all;
    std::vector<uint8_t> d;
    line_extract(im, *P, use_std_sort != 0, k, d, &all);
    *n = (int)k.size();
    if (n_all) *n_all = (int)all.size();
    if (*n > cap) return OLF_ERR_CAPACITY;
    if (*n) { std::memcpy(kls, k.data(), k.size() * sizeof(olf_keyline)); std::memcpy(desc, d.data(), d.size()); }
    if (all_kls) std::memcpy(all_kls, all.data(), std::min<size_t>(all.size(), all_cap) * sizeof(olf_keyline));
    return OLF_OK;
}

// LBD only, on caller-supplied key lines (float_desc optional: n*72 floats)
int orc_lbd_compute(const uint8_t* img, int w, int h, const olf_keyline* kls, int n, uint8_t* desc, float* float_desc)
{
    Image im(w, h);
    std::memcpy(im.d.data(), img, (size_t)w * h);
    std::vector<olf_keyline> k(kls, kls + n);
    std::vector<uint8_t> d;
    std::vector<float> fd;
    lbd_compute(im, k, d, float_desc ? &fd : nullptr);
    std::memcpy(desc, d.data(), d.size());
    if (float_desc) std::memcpy(float_desc, fd.data(), fd.size() * sizeof(float));
    return OLF_OK;
}

int orc_sobel3(const uint8_t* img, int w, int h, int16_t* dx, int16_t* dy)
{
    Image im(w, h);
    std::memcpy(im.d.data(), img, (size_t)w * h);
    std::vector<int16_t> a, b;
    sobel3(im, a, b);
    std::memcpy(dx, a.data(), a.size() * 2);
    std::memcpy(dy, b.data(), b.size() * 2);
    return 0;
}

}  // extern "C"

extern "C" int orc_lsd_region_stats(const uint8_t* img, int w, int h, const olf_line_params* P, int* sizes, int cap)
{
    orc::Image im(w, h);
    std::memcpy(im.d.data(), img, (size_t)w * h);
    std::vector<orc::Vec4f> lines;
    std::vector<int> rs;
    orc::lsd_detect(im, *P, lines, nullptr, &rs);
    for (int i = 0; i < std::min((int)rs.size(), cap); ++i) sizes[i] = rs[i];
    return (int)rs.size();
}
