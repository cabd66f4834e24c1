// precond_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// Input conditioning that precedes the path, restated from OpenCV 3.4 (un-vendored; PARITY UNPINNED):
//   cv::cvtColor(img, gray, CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY)   reference src/Tracking.cc:193-218
//       8-bit fixed point: (R*4899 + G*9617 + B*1868 + 2^13) >> 14                      (RGB2Gray<uchar>, yuv_shift = 14)
//   cv::remap(src, dst, M1, M2, INTER_LINEAR) with two CV_32FC1 maps, BORDER_CONSTANT 0  Examples/PL/PL_stereo_euroc.cc:136-137
//       sx = cvRound(mapx*32), sy = cvRound(mapy*32); integer part >> 5, 5-bit fractions; BilinearTab_i weights
//       32*{(32-fy)(32-fx), (32-fy)fx, fy(32-fx), fy*fx} (their sum is exactly 2^15, so initInterTab2D's fix-up never fires);
//       D = (sum + 2^14) >> 15; taps outside the source read the border value 0.
#include "oracle_common.hpp"

using namespace orc;
extern "C" {

// code: 0 RGB, 1 BGR, 2 RGBA, 3 BGRA (channel order of src)
int orc_cvt_gray(const uint8_t* src, int w, int h, int code, uint8_t* dst)
{
    const int cn = code >= 2 ? 4 : 3;
    const bool bgr = code & 1;
    for (int i = 0; i < w * h; ++i) {
        const uint8_t* p = src + (size_t)i * cn;
        const int r = bgr ? p[2] : p[0], g = p[1], b = bgr ? p[0] : p[2];
        dst[i] = (uint8_t)((r * 4899 + g * 9617 + b * 1868 + (1 << 13)) >> 14);
    }
    return 0;
}

int orc_remap_linear(const uint8_t* src, int sw, int sh, const float* mapx, const float* mapy, int dw, int dh, uint8_t* dst)
{
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            const int sxf = cvRoundf(mapx[(size_t)y * dw + x] * 32), syf = cvRoundf(mapy[(size_t)y * dw + x] * 32);
            const int fx = sxf & 31, fy = syf & 31;
            int sx = sxf >> 5, sy = syf >> 5;
            sx = std::min(std::max(sx, -32768), 32767); sy = std::min(std::max(sy, -32768), 32767);   // saturate_cast<short>
            const int w0 = 32 * (32 - fy) * (32 - fx), w1 = 32 * (32 - fy) * fx, w2 = 32 * fy * (32 - fx), w3 = 32 * fy * fx;
            auto px = [&](int xx, int yy) -> int { return (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src[(size_t)yy * sw + xx] : 0; };
            const int v = px(sx, sy) * w0 + px(sx + 1, sy) * w1 + px(sx, sy + 1) * w2 + px(sx + 1, sy + 1) * w3;
            dst[(size_t)y * dw + x] = sat_u8((v + (1 << 14)) >> 15);
        }
    return 0;
}

// cv::initUndistortRectifyMap(K, D, R, P(3x3), size, CV_32FC1, map1, map2) -- Examples/PL/PL_stereo_euroc.cc:97-98 -- as OpenCV 3.4's
// generic C++ path computes it (imgproc/undistort.cpp; recalled, not verified: convention C.13): everything in double;
// iR = (P * R)^-1 with cv::invert's closed-form 3 x 3 adjugate (n <= 3 never reaches the LU code); per row the homogeneous source coordinates
// start at (i*ir[1] + ir[2], i*ir[4] + ir[5], i*ir[7] + ir[8]) and ADVANCE by (ir[0], ir[3], ir[6]) per column -- a running sum, not a
// product; radial (k1 k2 k3 / k4 k5 k6) and tangential (p1 p2) distortion, no thin-prism / tilt terms; one rounding to float at the end.
// D: k1 k2 p1 p2 k3 k4 k5 k6 (nd of them, the rest zero).
int orc_init_undistort_rectify_map(const double* K, const double* D, int nd, const double* R, const double* P, int w, int h, float* map1, float* map2)
{
    double Ar_R[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Ar_R[3 * r + c] = P[3 * r] * R[c] + P[3 * r + 1] * R[3 + c] + P[3 * r + 2] * R[6 + c];
    const double* m = Ar_R;
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (d == 0.) return -1;
    d = 1. / d;
    double ir[9];
    ir[0] = (m[4] * m[8] - m[5] * m[7]) * d; ir[1] = (m[2] * m[7] - m[1] * m[8]) * d; ir[2] = (m[1] * m[5] - m[2] * m[4]) * d;
    ir[3] = (m[5] * m[6] - m[3] * m[8]) * d; ir[4] = (m[0] * m[8] - m[2] * m[6]) * d; ir[5] = (m[2] * m[3] - m[0] * m[5]) * d;
    ir[6] = (m[3] * m[7] - m[4] * m[6]) * d; ir[7] = (m[1] * m[6] - m[0] * m[7]) * d; ir[8] = (m[0] * m[4] - m[1] * m[3]) * d;
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < nd && i < 8; ++i) k[i] = D[i];
    const double k1 = k[0], k2 = k[1], p1 = k[2], p2 = k[3], k3 = k[4], k4 = k[5], k5 = k[6], k6 = k[7];
    const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
    for (int i = 0; i < h; ++i) {
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < w; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
            const double ww = 1. / _w, x = _x * ww, y = _y * ww;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, _2xy = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
            const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
            map1[(size_t)i * w + j] = (float)(fx * xd + u0);
            map2[(size_t)i * w + j] = (float)(fy * yd + v0);
        }
    }
    return 0;
}

}  // extern "C"
