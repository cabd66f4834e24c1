// tests/opencv_decl -- declaration-only stand-in for cv::line_descriptor::KeyLine (the reference vendors the real one:
// Thirdparty/line_descriptor/include/line_descriptor/descriptor_custom.hpp:105-186).  Test infrastructure that pins nothing; field order as olf_keyline
// (include/orbline_types.h), which the GPU tests compare with the oracle's records.
#pragma once
#include <opencv2/core/core.hpp>
namespace cv { namespace line_descriptor {
struct KeyLine {
    float angle;
    int class_id, octave;
    Point2f pt;
    float response, size;
    float startPointX, startPointY, endPointX, endPointY;
    float sPointInOctaveX, sPointInOctaveY, ePointInOctaveX, ePointInOctaveY;
    float lineLength;
    int numOfPixels;
};
} }  // namespace cv::line_descriptor
