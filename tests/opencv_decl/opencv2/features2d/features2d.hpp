// tests/opencv_decl -- declaration-only stand-in (see opencv2/core/core.hpp in this directory: test infrastructure that pins nothing).
#pragma once
#include <opencv2/core/core.hpp>
namespace cv {
class KeyPoint {          // the seven fields of cv::KeyPoint, 28 bytes
public:
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};
}  // namespace cv
