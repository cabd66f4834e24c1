// Compile-only check (g++ -fsyntax-only -DORBLINE_WITH_OPENCV -Itests/opencv_decl -Iinclude): the cv::InputArray / cv::OutputArray / cv::Mat / KeyLine
// overloads of include/orbline_adaptor.hpp -- the ones the reference's Frame::ExtractORB / Frame::ExtractLine call (src/Frame.cc:350-364, signatures
// include/ORBextractor.h:66-68, include/LineExtractor.h:49-50) -- and the reference-signature templates instantiated on cv::Mat.  The OpenCV side is
// tests/opencv_decl, a declaration-only stand-in: this test proves that the blocks parse and type-check, nothing about OpenCV's behaviour.
#ifndef ORBLINE_WITH_OPENCV
#define ORBLINE_WITH_OPENCV
#endif
#include "orbline_adaptor.hpp"

namespace {
struct FrameLike {        // the members Frame::ExtractORB / ExtractLine touch (include/Frame.h)
    ORB_SLAM2::ORBextractor *mpORBextractorLeft, *mpORBextractorRight;
    ORB_SLAM2::Lineextractor *mpLineextractorLeft, *mpLineextractorRight;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight;
    cv::Mat mDescriptors, mDescriptorsRight;
    std::vector<cv::line_descriptor::KeyLine> mvKeys_Line, mvKeysRight_Line;
    cv::Mat mDescriptors_Line, mDescriptorsRight_Line;
    void ExtractORB(int flag, const cv::Mat& im)
    {
        if (flag == 0) (*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors);                  // src/Frame.cc:352-355, verbatim shape
        else (*mpORBextractorRight)(im, cv::Mat(), mvKeysRight, mDescriptorsRight);
    }
    void ExtractLine(int flag, const cv::Mat& im)
    {
        if (flag == 0) (*mpLineextractorLeft)(im, cv::Mat(), mvKeys_Line, mDescriptors_Line);       // src/Frame.cc:360-363
        else (*mpLineextractorRight)(im, cv::Mat(), mvKeysRight_Line, mDescriptorsRight_Line);
    }
};
}  // namespace

int typecheck_only(FrameLike& f, const cv::Mat& imLeft, const cv::Mat& imRight)
{
    f.ExtractORB(0, imLeft); f.ExtractORB(1, imRight);
    f.ExtractLine(0, imLeft); f.ExtractLine(1, imRight);
    const cv::Mat& pyr = f.mpORBextractorLeft->mvImagePyramid[0];                                       // public member read by Frame::ComputeStereoMatches (src/Frame.cc:799-816)
    std::vector<int> matches_12;
    int n = ORB_SLAM2::match(f.mDescriptors_Line, f.mDescriptorsRight_Line, 0.75f, matches_12);          // src/Tracking.cc:1308 shape (cv::Mat descriptors)
    n += ORB_SLAM2::matchNNR(f.mDescriptors_Line, f.mDescriptorsRight_Line, 0.75f, matches_12);
    n += ORB_SLAM2::distance(f.mDescriptors.row(0), f.mDescriptorsRight.row(0));
    n += ORB_SLAM2::ORBmatcher::DescriptorDistance(f.mDescriptors.row(0), f.mDescriptorsRight.row(0));  // src/ORBmatcher.cc:1795
    return n + pyr.rows + f.mpORBextractorLeft->GetLevels();
}
