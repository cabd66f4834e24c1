// tests/opencv_decl -- DECLARATION-ONLY stand-in for the few OpenCV 3.4 types the adaptor's ORBLINE_WITH_OPENCV blocks name.
// TEST INFRASTRUCTURE, and of the weakest kind: it pins NOTHING.  It exists so that `g++ -fsyntax-only -DORBLINE_WITH_OPENCV` can type-check
// include/orbline_adaptor.hpp / orbline_reference_api.hpp against the call shapes of the reference (src/Frame.cc:350-364,
// include/ORBextractor.h:66-68, include/LineExtractor.h:49-50) in an image that has no OpenCV.  The signatures are written from the public OpenCV
// 3.4 API as documented (cv::Mat, cv::_InputArray, cv::_OutputArray, cv::KeyPoint); no member is defined, nothing links against it, no result of the
// path depends on it, and it is never on an include path of the product or of the oracle.
#pragma once
#include <cstddef>
#include <vector>

typedef unsigned char uchar;
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_Assert(expr) do { if (!(expr)) ::cv::error_stub(#expr); } while (0)

namespace cv {
void error_stub(const char*);
template <typename T> struct Point_ { T x, y; };
typedef Point_<float> Point2f;
struct MatStep { operator std::size_t() const; std::size_t operator[](int) const; };
class Mat {
public:
    Mat();
    Mat(int rows, int cols, int type);
    Mat(const Mat&);
    ~Mat();
    Mat& operator=(const Mat&);
    void create(int rows, int cols, int type);
    void release();
    Mat clone() const;
    Mat row(int y) const;
    bool isContinuous() const;
    bool empty() const;
    int type() const;
    int depth() const;
    template <typename T> T* ptr(int y = 0);
    template <typename T> const T* ptr(int y = 0) const;
    template <typename T> T& at(int y, int x);
    int flags, dims, rows, cols;
    uchar* data;
    MatStep step;
};
class _InputArray {
public:
    _InputArray();
    _InputArray(const Mat&);
    template <typename T> _InputArray(const std::vector<T>&);
    Mat getMat(int idx = -1) const;
    bool empty() const;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray();
    _OutputArray(Mat&);
    void create(int rows, int cols, int type, int i = -1, bool allowTransposed = false, int fixedDepthMask = 0) const;
    void release() const;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
