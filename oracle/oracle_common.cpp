// oracle_common.cpp -- CPU ORACLE (test infrastructure): restated OpenCV 3.4 primitives.
// See oracle_common.hpp for scope; every function cites what it restates.
#include "oracle_common.hpp"
#include <cfloat>

namespace orc {

// cv::getGaussianKernel(n, sigma>0, CV_32F): taps exp(-x^2/(2 sigma^2)) stored as float,
// normalised by the double sum of the float taps; createSeparableLinearFilter then converts
// them with convertTo(CV_32S, 256) (cvRound).  SURVEY App. A.3.
// sum256 (convention C.11): the later "bit-exact" kernel instead -- taps computed in double, converted to 8 fractional bits with the
// rounding error carried from tap to tap (outside in), the centre tap taking what is left of 256 (getGaussianKernelFixedPoint_ED).
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
            const int v = cvRound(adj);
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
        q[i] = cvRound((double)cf[i] * 256.0);
    }
    return q;
}

// Separable fixed-point filter: row pass exact int32, column pass (sum + 2^15) >> 16, saturated.
Image gaussian_blur_u8(const Image& src, const std::vector<int>& taps)
{
    // (integer arithmetic throughout: the interior runs without the border index and in loops a compiler vectorises -- the sums are the same
    // numbers in any order; the CPU baseline of bench.py times this function)
    const int n = (int)taps.size(), r = n / 2, w = src.w, h = src.h;
    std::vector<int> tmp((size_t)w * h);
    const int* tp = taps.data();
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = src.row(y);
        int* t = &tmp[(size_t)y * w];
        const int xa = std::min(r, w), xb = std::max(xa, w - r);
        for (int x = 0; x < w; ++x) {
            if (x == xa && xb > xa) {
                for (int xi = xa; xi < xb; ++xi) {
                    int acc = 0;
                    for (int k = 0; k < n; ++k) acc += tp[k] * s[xi + k - r];
                    t[xi] = acc;
                }
                x = xb - 1;
                continue;
            }
            int acc = 0;
            for (int k = 0; k < n; ++k) acc += tp[k] * s[reflect101(x + k - r, w)];
            t[x] = acc;
        }
    }
    Image dst(w, h);
    std::vector<int> accRow(w);
    for (int y = 0; y < h; ++y) {
        std::fill(accRow.begin(), accRow.end(), 32768);
        for (int k = 0; k < n; ++k) {
            const int* t = &tmp[(size_t)reflect101(y + k - r, h) * w];
            const int c = tp[k];
            int* a = accRow.data();
            for (int x = 0; x < w; ++x) a[x] += c * t[x];
        }
        uint8_t* d = dst.row(y);
        for (int x = 0; x < w; ++x) d[x] = sat_u8(accRow[x] >> 16);
    }
    return dst;
}

// cv::resize INTER_LINEAR_EXACT, 8UC1 (resize.cpp resize_bitExact<uchar, ufixedpoint16>, recalled -- convention C.10): source position
// scale * (d + 0.5) - 0.5 in (soft) double, offset = floor, the fractional part converted to 8 fractional bits (round to nearest), clamped at
// both image edges; horizontal pass in 8.8 fixed point (exact), vertical pass 8.8 x 0.8 -> 8.16, one rounding (+ 2^15 >> 16).
Image resize_linear_exact_u8(const Image& src, int dw, int dh, double scale_x, double scale_y)
{
    const int sw = src.w, sh = src.h;
    auto coefs = [](int dn, int sn, double scale, std::vector<int>& ofs, std::vector<int>& c1) {
        ofs.resize(dn); c1.resize(dn);
        for (int d = 0; d < dn; ++d) {
            const double f = scale * (d + 0.5) - 0.5;
            int i = cvFloor(f);
            int a = cvRound((f - i) * 256.0);
            if (i < 0) { i = 0; a = 0; }
            if (i >= sn - 1) { i = sn - 1; a = 0; }
            ofs[d] = i; c1[d] = a;
        }
    };
    std::vector<int> xo, xa, yo, ya;
    coefs(dw, sw, scale_x, xo, xa);
    coefs(dh, sh, scale_y, yo, ya);
    Image dst(dw, dh);
    for (int dy = 0; dy < dh; ++dy) {
        const uint8_t* s0 = src.row(yo[dy]);
        const uint8_t* s1 = src.row(std::min(yo[dy] + 1, sh - 1));
        const int b1 = ya[dy], b0 = 256 - b1;
        for (int dx = 0; dx < dw; ++dx) {
            const int x0 = xo[dx], x1 = std::min(x0 + 1, sw - 1), a1 = xa[dx], a0 = 256 - a1;
            const int h0 = s0[x0] * a0 + s0[x1] * a1, h1 = s1[x0] * a0 + s1[x1] * a1;      // 8.8
            dst.at(dx, dy) = sat_u8((h0 * b0 + h1 * b1 + 32768) >> 16);
        }
    }
    return dst;
}

// cv::resize INTER_LINEAR, 8UC1: 11-bit coefficients (INTER_RESIZE_COEF_SCALE = 2048),
// HResizeLinear<uchar,int,short> then VResizeLinear's 8-bit specialisation.  SURVEY App. A.2.
Image resize_linear_u8(const Image& src, int dw, int dh, double scale_x, double scale_y)
{
    const int sw = src.w, sh = src.h;
    std::vector<int> xofs(dw), a0(dw), a1(dw);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        a0[dx] = (int16_t)std::min(std::max(cvRoundf((1.f - fx) * 2048.f), -32768), 32767);
        a1[dx] = (int16_t)std::min(std::max(cvRoundf(fx * 2048.f), -32768), 32767);
    }
    Image dst(dw, dh);
    std::vector<int> h0(dw), h1(dw);
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        int b0 = (int16_t)cvRoundf((1.f - fy) * 2048.f), b1 = (int16_t)cvRoundf(fy * 2048.f);
        int y0 = std::min(std::max(sy, 0), sh - 1), y1 = std::min(std::max(sy + 1, 0), sh - 1);
        const uint8_t *S0 = src.row(y0), *S1 = src.row(y1);
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx], sx1 = std::min(sx + 1, sw - 1);
            h0[dx] = S0[sx] * a0[dx] + S0[sx1] * a1[dx];
            h1[dx] = S1[sx] * a0[dx] + S1[sx1] * a1[dx];
        }
        uint8_t* D = dst.row(dy);
        for (int dx = 0; dx < dw; ++dx)
            D[dx] = (uint8_t)((((b0 * (h0[dx] >> 4)) >> 16) + ((b1 * (h1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    return dst;
}

// cv::fastAtan2 -> atanImpl<float>; SURVEY App. A.5.  fp32, no FMA.
float fastAtan2(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// The reference's SWAR bit count over 8 x 32-bit words.
int hamming256(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        uint32_t pa, pb;
        std::memcpy(&pa, a + 4 * i, 4);
        std::memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

}  // namespace orc
