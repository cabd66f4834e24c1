// Instantiates every template of include/orbline_reference_api.hpp -- the reference's own call signatures -- with minimal stand-ins that
// carry the member names the reference's Frame / KeyFrame / MapPoint / MapLine / cv::Mat expose (include/Frame.h:137-260,
// include/KeyFrame.h, include/MapPoint.h, include/MapLine.h).  No OpenCV: the stand-in Mat has the handful of members the templates use.
//   * without a GPU (the CPU test): everything must compile and link; the calls that need a device must throw, not fall back.
//   * with a GPU (run by tests/test_search_gpu.py): the reference-signature calls must give exactly what the view-based calls give, and
//     StereoFrameFeatures must fill the frame like the fused entry does.
#include "../include/orbline_adaptor.hpp"
#include <array>
#include <cstdio>
#include <cstdlib>
#include <map>

namespace standin {
// the members of cv::Mat the adaptor touches
struct Mat {
    int rows = 0, cols = 0;
    std::vector<uint8_t> store;
    uint8_t* data = nullptr;
    size_t step = 0;
    int elem = 1;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(const Mat& o) : rows(o.rows), cols(o.cols), store(o.store), step(o.step), elem(o.elem) { data = store.empty() ? nullptr : store.data(); }
    Mat& operator=(const Mat& o) { rows = o.rows; cols = o.cols; store = o.store; step = o.step; elem = o.elem; data = store.empty() ? nullptr : store.data(); return *this; }
    void create(int r, int c, int type) { rows = r; cols = c; elem = type == 5 ? 4 : 1; step = (size_t)c * elem; store.assign((size_t)r * step, 0); data = store.empty() ? nullptr : store.data(); }
    bool empty() const { return rows == 0 || cols == 0; }
    bool isContinuous() const { return true; }
    Mat clone() const { return *this; }
    template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * step); }
    template <class T> T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <class T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
    template <class T> const T& at(int i) const { return reinterpret_cast<const T*>(data)[i]; }      // 3 x 1 vectors
    template <class T> T& at(int i) { return reinterpret_cast<T*>(data)[i]; }
};
typedef olf_keypoint KeyPoint;          // layout of cv::KeyPoint
typedef olf_keyline KeyLine;            // layout of cv::line_descriptor::KeyLine

struct MapPoint {
    Mat world, desc;
    bool bad = false;
    int nobs = 1;
    bool mbTrackInView = false;
    int mnTrackScaleLevel = 0;
    float mTrackViewCos = 1.f, mTrackProjX = 0.f, mTrackProjY = 0.f, mTrackProjXR = -1.f;
    Mat GetWorldPos() const { return world; }
    Mat GetDescriptor() const { return desc; }
    bool isBad() const { return bad; }
    int Observations() const { return nobs; }
};
struct MapLine {
    Mat desc;
    Mat GetDescriptor() const { return desc; }
};
typedef std::map<unsigned, std::vector<unsigned>> FeatureVector;   // DBoW2::FeatureVector

struct Frame {
    ORB_SLAM2::ORBextractor *mpORBextractorLeft = nullptr, *mpORBextractorRight = nullptr;
    ORB_SLAM2::Lineextractor *mpLineextractorLeft = nullptr, *mpLineextractorRight = nullptr;
    static float fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY;
    float mbf = 386.1448f, mb = 0.537f;
    int N = 0, N_l = 0;
    std::vector<KeyPoint> mvKeys, mvKeysRight, mvKeysUn;
    std::vector<KeyLine> mvKeys_Line, mvKeysRight_Line;
    std::vector<float> mvuRight, mvDepth;
    std::vector<std::pair<float, float>> mvDisparity_l;
    std::vector<std::array<double, 3>> mvle_l;      // std::vector<Vector3d> in the reference (Eigen): operator[] is all the adaptor uses
    FeatureVector mFeatVec;
    Mat mDescriptors, mDescriptorsRight, mDescriptors_Line, mDescriptorsRight_Line;
    std::vector<MapPoint*> mvpMapPoints;
    std::vector<bool> mvbOutlier;
    Mat mTcw;
    int mnScaleLevels = 8;
    std::vector<float> mvScaleFactors;
};
float Frame::fx = 718.856f, Frame::fy = 718.856f, Frame::cx = 607.19f, Frame::cy = 185.2f;
float Frame::mnMinX = 0.f, Frame::mnMaxX = 1242.f, Frame::mnMinY = 0.f, Frame::mnMaxY = 375.f;

struct KeyFrame {
    std::vector<KeyPoint> mvKeysUn;
    Mat mDescriptors;
    FeatureVector mFeatVec;
    std::vector<MapPoint*> mps;
    std::vector<MapPoint*> GetMapPointMatches() const { return mps; }
};
}  // namespace standin

using namespace standin;

static unsigned long long rng_state = 88172645463325252ull;
static unsigned rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 11); }

static Mat desc_rows(int n) { Mat m(n, 32, 0); for (size_t i = 0; i < m.store.size(); ++i) m.store[i] = (uint8_t)rnd(); return m; }

static void fill_frame(Frame& F, int n)
{
    F.N = n;
    F.mvKeysUn.resize(n);
    for (int i = 0; i < n; ++i) {
        olf_keypoint k = {};
        k.x = 20.f + (float)(rnd() % 1200); k.y = 20.f + (float)(rnd() % 330); k.octave = (int)(rnd() % 8); k.angle = (float)(rnd() % 360); k.size = 31.f; k.class_id = -1;
        F.mvKeysUn[i] = k;
    }
    F.mvKeys = F.mvKeysUn;
    F.mDescriptors = desc_rows(n);
    F.mvuRight.assign(n, -1.f); F.mvDepth.assign(n, -1.f);
    F.mvpMapPoints.assign(n, nullptr); F.mvbOutlier.assign(n, false);
    F.mvScaleFactors.resize(8); F.mvScaleFactors[0] = 1.f;
    for (int i = 1; i < 8; ++i) F.mvScaleFactors[i] = F.mvScaleFactors[i - 1] * 1.2f;
    F.mTcw.create(4, 4, 5);
    for (int i = 0; i < 4; ++i) F.mTcw.at<float>(i, i) = 1.f;
}

int main()
{
    const bool gpu = olf_device_count() > 0;
    int thrown = 0;
    ORB_SLAM2::ORBmatcher matcher(0.9f, true);

    // ---- frames with map points: last frame's points project into the current frame (identity poses, points in front of the camera)
    Frame last, cur;
    const int n = 300;
    fill_frame(last, n); fill_frame(cur, n);
    std::vector<MapPoint> pool(n);
    for (int i = 0; i < n; ++i) {
        MapPoint& p = pool[i];
        const float z = 5.f + (float)(rnd() % 20);
        const olf_keypoint& k = cur.mvKeysUn[i];                      // the point projects onto key point i of the current frame
        p.world.create(3, 1, 5);
        p.world.at<float>(0) = (k.x - Frame::cx) * z / Frame::fx; p.world.at<float>(1) = (k.y - Frame::cy) * z / Frame::fy; p.world.at<float>(2) = z;
        p.desc.create(1, 32, 0);
        for (int b = 0; b < 32; ++b) p.desc.data[b] = cur.mDescriptors.ptr<uint8_t>(i)[b];      // and looks like it
        last.mvpMapPoints[i] = (i % 3) ? &p : nullptr;
        last.mvKeysUn[i].octave = k.octave; last.mvKeysUn[i].angle = k.angle;
        p.mbTrackInView = (i % 4) != 0; p.mnTrackScaleLevel = k.octave; p.mTrackProjX = k.x; p.mTrackProjY = k.y;
    }

    // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)     src/Tracking.cc:1296
    int nm = -1;
    try { nm = matcher.SearchByProjection(cur, last, 7.f, false); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
    if (gpu) {
        int assigned = 0;
        for (int i = 0; i < n; ++i) assigned += cur.mvpMapPoints[i] != nullptr;
        if (nm <= 0 || assigned != nm) { std::printf("SearchByProjection(Frame, Frame): %d matches, %d assigned\n", nm, assigned); return 10; }
        for (int i = 0; i < n; ++i) if (cur.mvpMapPoints[i] && cur.mvpMapPoints[i] != &pool[i]) return 11;     // every point found its own key point
    }
    // int SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th = 3)            src/Tracking.cc:1941
    Frame f2; fill_frame(f2, n);
    f2.mvKeysUn = cur.mvKeysUn; f2.mDescriptors = cur.mDescriptors;
    std::vector<MapPoint*> local;
    for (int i = 0; i < n; ++i) local.push_back(&pool[i]);
    int nl = -1;
    try { nl = matcher.SearchByProjection(f2, local, 3.f); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
    if (gpu) {
        if (nl <= 0) return 12;
        for (int i = 0; i < n; ++i) if (f2.mvpMapPoints[i] && f2.mvpMapPoints[i] != &pool[i]) return 13;
    }
    // int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches)                        src/Tracking.cc:970
    KeyFrame kf;
    kf.mvKeysUn = cur.mvKeysUn; kf.mDescriptors = cur.mDescriptors;
    kf.mps.assign(n, nullptr);
    for (int i = 0; i < n; ++i) if (i % 2) kf.mps[i] = &pool[i];
    for (int i = 0; i < n; ++i) { kf.mFeatVec[(unsigned)(i % 10)].push_back((unsigned)i); cur.mFeatVec[(unsigned)(i % 10)].push_back((unsigned)i); }
    std::vector<MapPoint*> bowMatches;
    int nb = -1;
    try { nb = matcher.SearchByBoW(&kf, cur, bowMatches); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
    if (gpu) {
        if (nb <= 0 || (int)bowMatches.size() != n) return 14;
        for (int i = 0; i < n; ++i) if (bowMatches[i] && bowMatches[i] != &pool[i]) return 15;
    }
    // int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize) src/Tracking.cc:628
    {
        struct Point2f { float x, y; };
        Frame ini; fill_frame(ini, n);
        ini.mvKeysUn = cur.mvKeysUn; ini.mDescriptors = cur.mDescriptors;
        std::vector<Point2f> prev(n);
        for (int i = 0; i < n; ++i) { prev[i].x = cur.mvKeysUn[i].x + 1.f; prev[i].y = cur.mvKeysUn[i].y - 1.f; }
        std::vector<int> ini12;
        int ni = -1;
        try { ni = matcher.SearchForInitialization(ini, cur, prev, ini12, 100); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
        if (gpu) {
            if (ni <= 0 || (int)ini12.size() != n) return 30;
            for (int i = 0; i < n; ++i) if (ini12[i] >= 0 && (ini12[i] != i || prev[i].x != cur.mvKeysUn[i].x || cur.mvKeysUn[i].octave > 0)) return 31;
        }
    }
    // static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b); int distance(const cv::Mat&, const cv::Mat&)     (host, no device)
    Mat zero(1, 32, 0), ones(1, 32, 0);
    for (int b = 0; b < 32; ++b) ones.data[b] = 0xff;
    if (ORB_SLAM2::ORBmatcher::DescriptorDistance(zero, ones) != 256 || ORB_SLAM2::distance(zero, ones) != 256) return 16;

    // int match(const cv::Mat&, const cv::Mat&, float, std::vector<int>&) / matchNNR / match(std::vector<MapLine*>, Frame&, ...)   src/Tracking.cc:1308,979,1970
    Mat d1 = desc_rows(40), d2 = d1;
    std::vector<int> m12;
    int r1 = -1, r2 = -1, r3 = -1;
    try { r1 = ORB_SLAM2::match(d1, d2, 0.9f, m12); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
    if (gpu) { if (r1 != 40) return 17; for (int i = 0; i < 40; ++i) if (m12[i] != i) return 18; }
    try { r2 = ORB_SLAM2::matchNNR(d1, d2, 0.9f, m12); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
    std::vector<MapLine> mls(40);
    std::vector<MapLine*> pml;
    for (int i = 0; i < 40; ++i) { mls[i].desc.create(1, 32, 0); for (int b = 0; b < 32; ++b) mls[i].desc.data[b] = d1.ptr<uint8_t>(i)[b]; pml.push_back(&mls[i]); }
    Frame lf; lf.mDescriptors_Line = d2;
    try { r3 = ORB_SLAM2::match(pml, lf, 0.9f, m12); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
    if (gpu && (r2 != 40 || r3 != 40)) return 19;

    // GridStructure / GridWindow / getLineCoords / matchGrid x2 (host)                                             src/Frame.cc:910-926
    {
        ORB_SLAM2::GridStructure grid(48, 64);
        std::list<std::pair<int, int>> cells;
        ORB_SLAM2::getLineCoords(1.5, 2.5, 20.5, 9.25, cells);
        if (cells.empty() || cells.front() != std::make_pair(1, 2)) return 20;
        for (const auto& c : cells) grid.at(c.first, c.second).push_back(0);
        ORB_SLAM2::GridWindow w; w.width = std::make_pair(10, 0); w.height = std::make_pair(0, 0);
        std::vector<ORB_SLAM2::line_2d> lines1(1, std::make_pair(std::make_pair(3, 2), std::make_pair(21, 9)));
        std::vector<std::pair<double, double>> dir2(1, std::make_pair(0.94, 0.34));
        Mat a = desc_rows(1), b = a;
        std::vector<int> g12;
        if (ORB_SLAM2::matchGrid(lines1, a, grid, b, dir2, w, g12) != 1 || g12[0] != 0) return 21;
        std::vector<ORB_SLAM2::point_2d> pts(1, std::make_pair(3, 2));
        std::vector<int> p12;
        if (ORB_SLAM2::matchGrid(pts, a, grid, b, w, p12) != 1 || p12[0] != 0) return 22;
    }

    // StereoFrameFeatures(frame, imLeft, imRight): the feature part of Frame::Frame                                src/Frame.cc:164-171,199-207
    {
        ORB_SLAM2::ORBextractor orbL(1000, 1.2f, 8, 20, 7), orbR(1000, 1.2f, 8, 20, 7);
        ORB_SLAM2::Lineextractor lineL(200, 0.025), lineR(200, 0.025);
        Frame F;
        F.mpORBextractorLeft = &orbL; F.mpORBextractorRight = &orbR; F.mpLineextractorLeft = &lineL; F.mpLineextractorRight = &lineR;
        Mat L(480, 640, 0), R(480, 640, 0);
        for (int y = 0; y < 480; ++y)
            for (int x = 0; x < 640; ++x) { const uint8_t v = (uint8_t)(((x / 40 + y / 40) & 1) ? 200 : 40); L.at<uint8_t>(y, x) = v; R.at<uint8_t>(y, (x + 632) % 640) = v; }
        try { ORB_SLAM2::StereoFrameFeatures(F, L, R); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
        if (gpu) {
            if (F.N <= 0 || F.N != (int)F.mvKeys.size() || F.mDescriptors.rows != F.N || (int)F.mvuRight.size() != F.N) return 23;
            if (F.N_l <= 0 || (int)F.mvKeys_Line.size() != F.N_l || F.mDescriptors_Line.rows != F.N_l || (int)F.mvle_l.size() != F.N_l) return 24;
            // the same images through the per-class drop-in (strided entry) give the same left key points
            std::vector<olf_keypoint> k; std::vector<uint8_t> d;
            orbR.extract(L.data, 640, 480, 640, k, d);
            if ((int)k.size() != F.N || std::memcmp(k.data(), F.mvKeys.data(), k.size() * sizeof(olf_keypoint)) != 0) return 25;
        }
        Mat small(100, 100, 0);
        try { ORB_SLAM2::StereoFrameFeatures(F, L, small); return 26; } catch (const std::runtime_error&) {}      // size mismatch throws, src/Frame.cc:145-146
    }
    if (!gpu && thrown != 8) { std::printf("no device: %d of 8 device calls threw\n", thrown); return 40; }
    if (gpu && thrown != 0) return 41;
    std::printf(gpu ? "REFERENCE_API_OK %d %d %d\n" : "REFERENCE_API_COMPILED %d %d %d\n", nm, nl, nb);
    return 0;
}
