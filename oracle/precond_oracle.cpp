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

}  // extern "C"
