// oracle_common.hpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// Plain C++ restatement of the reference's per-frame feature path, used only by tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / CPU baseline.
// Nothing under orb_line_slam_amd/ may include, link or call it.
//
// PARITY UNPINNED: the reference has no tests, golden vectors or fixtures and
// cannot be built here (needs OpenCV 3.4 / Eigen / Pangolin); only its two
// STL-only files build (oracle/_ref, see oracle/Makefile) and pin gridStructure/LineIterator.
// The OpenCV primitives restated here follow SURVEY.md Appendix A (OpenCV 3.4 generic paths);
// the non-deterministic sites of the reference follow the conventions of Appendix C.
//
// Compile with -ffp-contract=off (convention C.4: no FMA contraction anywhere).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../include/orbline_types.h"

namespace orc {

// cvRound: round-half-to-even (SSE cvtss2si / cvtsd2si), SURVEY App. A.6
static inline int cvRound(double v) { return (int)std::lrint(v); }
static inline int cvRoundf(float v) { return (int)std::lrintf(v); }
static inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
static inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }
static inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> d;
    Image() {}
    Image(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_) {}
    uint8_t& at(int x, int y) { return d[(size_t)y * w + x]; }
    uint8_t at(int x, int y) const { return d[(size_t)y * w + x]; }
    const uint8_t* row(int y) const { return &d[(size_t)y * w]; }
    uint8_t* row(int y) { return &d[(size_t)y * w]; }
};

// BORDER_REFLECT_101 index (App. A.1): gfedcb|abcdefgh|gfedcba
static inline int reflect101(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

// cv::getGaussianKernel(n, sigma, CV_32F) followed by the 8-bit fixed-point conversion
// (App. A.3): integer taps round(256*k).
std::vector<int> gaussian_taps_q8(int n, double sigma, int sum256 = 0);
// cv::resize(..., INTER_LINEAR_EXACT) for 8UC1 (convention C.10): 8-bit coefficients, exact 8.8 x 0.8 fixed point, one rounding at the end
Image resize_linear_exact_u8(const Image& src, int dw, int dh, double scale_x, double scale_y);
// cv::GaussianBlur on 8-bit single channel, BORDER_REFLECT_101, fixed point (App. A.3).
Image gaussian_blur_u8(const Image& src, const std::vector<int>& taps);
// cv::resize(..., INTER_LINEAR) for 8UC1 (App. A.2). scale_x/scale_y are the *source step
// per destination pixel* (1/inv_scale) exactly as cv::resize computes them.
Image resize_linear_u8(const Image& src, int dw, int dh, double scale_x, double scale_y);
// cv::fastAtan2 (App. A.5), degrees in [0,360)
float fastAtan2(float y, float x);

// 256-bit Hamming distance, src/ORBmatcher.cc:1795-1811 == src/LineMatcher.cpp:134-150
int hamming256(const uint8_t* a, const uint8_t* b);

}  // namespace orc
