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
    {   // the per-frame searches instantiate; with a null context the library reports the bad argument and the adaptor throws
        ORB_SLAM2::ORBmatcher m(0.9f, true);
        olf_frame_view f = {};
        std::vector<int32_t> matches;
        ORB_SLAM2::ORBmatcher::TrackedMapPoints mps;
        int thrown = 0;
        try { m.SearchByProjection(nullptr, f, f, 7.f, false, matches); } catch (const std::runtime_error&) { ++thrown; }
        try { m.SearchByProjection(nullptr, f, mps, 1.f, matches); } catch (const std::runtime_error&) { ++thrown; }
        try { m.SearchByBoW(nullptr, f, f, matches); } catch (const std::runtime_error&) { ++thrown; }
        ORB_SLAM2::ORBmatcher::FuseMapPoints fm;
        std::vector<std::pair<size_t, size_t>> pairs;
        std::vector<int32_t> bi, bd;
        const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0}, I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        try { m.SearchByProjection(nullptr, f, f, nullptr, 10.f, 100, matches); } catch (const std::runtime_error&) { ++thrown; }
        try { m.SearchByBoW(nullptr, ORB_SLAM2::ORBmatcher::KeyFramePair(), f, f, matches); } catch (const std::runtime_error&) { ++thrown; }
        try { m.SearchForTriangulation(nullptr, f, f, I9, false, pairs); } catch (const std::runtime_error&) { ++thrown; }
        try { m.FuseSearch(nullptr, f, fm, 3.f, bi, bd); } catch (const std::runtime_error&) { ++thrown; }
        try { m.FuseSearch(nullptr, f, I16, fm, 4.f, bi, bd); } catch (const std::runtime_error&) { ++thrown; }
        try { m.SearchBySim3(nullptr, f, f, matches, 1.f, I9, z3, 7.5f); } catch (const std::runtime_error&) { ++thrown; }
        if (thrown != 9) return 6;
    }
    std::printf("ADAPTOR_OK\n");
    return 0;
}
