// compile + link check of include/orbline_adaptor.hpp without OpenCV (CPU suite; no GPU call is made)
#include "../include/orbline_adaptor.hpp"
#include <cstdio>
int main()
{
    ORB_SLAM2::ORBextractor orb(1000, 1.2f, 8, 20, 7);
    ORB_SLAM2::Lineextractor line(200, 0.025);
    uint8_t a[32] = {0}, b[32];
    for (int i = 0; i < 32; ++i) b[i] = 0xff;
    if (ORB_SLAM2::ORBmatcher::DescriptorDistance(a, b) != 256) return 1;
    if (orb.GetLevels() != 8 || ORB_SLAM2::ORBmatcher::TH_HIGH != 100) return 2;
    if (olf_device_count() == 0) {   // no GPU: the extractor must throw, not fall back
        std::vector<olf_keypoint> k; std::vector<uint8_t> d; std::vector<uint8_t> img(640 * 480, 7);
        try { orb(img.data(), 640, 480, k, d); return 3; } catch (const std::runtime_error& e) { std::printf("expected: %s\n", e.what()); }
    }
    {   // the vocabulary adaptor instantiates with DBoW2's container types (std::map based)
        ORB_SLAM2::ORBVocabulary voc;
        std::map<unsigned, double> bow; std::map<unsigned, std::vector<unsigned>> fv;
        if (!voc.empty() || voc.loadFromTextFile("/nonexistent/voc.txt")) return 4;
        voc.transform(nullptr, a, 1, bow, fv, 4);            // empty vocabulary: empty vectors, no device call
        if (!bow.empty() || !fv.empty()) return 5;
    }
    std::printf("ADAPTOR_OK\n");
    return 0;
}
