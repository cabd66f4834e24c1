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
#include <cmath>
#include <map>
#include <set>

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

struct KeyFrame;
struct MapPoint {
    Mat world, desc, normal;
    bool bad = false;
    int nobs = 1;
    std::map<KeyFrame*, size_t> observers;
    MapPoint* replaced = nullptr;
    Mat GetNormal() const { return normal; }
    bool IsInKeyFrame(KeyFrame* k) const { return observers.count(k) != 0; }
    int GetIndexInKeyFrame(KeyFrame* k) const { auto it = observers.find(k); return it == observers.end() ? -1 : (int)it->second; }
    void AddObservation(KeyFrame* k, size_t idx) { if (!observers.count(k)) { observers[k] = idx; ++nobs; } }
    void Replace(MapPoint* p) { bad = true; replaced = p; }
    // only the public accessors of the reference (mfMaxDistance / mfMinDistance are protected there): the templates have to invert them
    float GetMaxDistanceInvariance() const { return 1.2f * maxDistance; }
    float GetMinDistanceInvariance() const { return 0.8f * minDistance; }
    float rawMax() const { return maxDistance; }
    float rawMin() const { return minDistance; }
    void setDistances(float mx, float mn) { maxDistance = mx; minDistance = mn; }
private:
    float maxDistance = 0.f, minDistance = 0.f;
public:
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
    int N = 0;
    std::vector<KeyPoint> mvKeysUn;
    std::vector<float> mvuRight;
    Mat mDescriptors;
    FeatureVector mFeatVec;
    std::vector<MapPoint*> mps;
    float fx = 718.856f, fy = 718.856f, cx = 607.19f, cy = 185.2f, mbf = 386.1448f;
    int mnMinX = 0, mnMinY = 0, mnMaxX = 1242, mnMaxY = 375;
    int mnScaleLevels = 8;
    std::vector<float> mvScaleFactors;
    Mat Tcw;
    std::vector<MapPoint*> GetMapPointMatches() { return mps; }
    MapPoint* GetMapPoint(size_t idx) { return mps[idx]; }
    std::set<MapPoint*> GetMapPoints() { std::set<MapPoint*> s; for (MapPoint* p : mps) if (p && !p->bad) s.insert(p); return s; }
    void AddMapPoint(MapPoint* p, size_t idx) { mps[idx] = p; }
    Mat GetPose() { return Tcw; }
    Mat GetCameraCenter() { Mat c; c.create(3, 1, 5); for (int r = 0; r < 3; ++r) { float a = 0; for (int k = 0; k < 3; ++k) a -= Tcw.at<float>(k, r) * Tcw.at<float>(k, 3); c.at<float>(r) = a; } return c; }
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
    // ---- the rest of the ORBmatcher surface (include/ORBmatcher.h:37-103) through the reference's own signatures ----------------------------
    {
        // every pool point gets a normal, distances and a second descriptor-identical twin (for the two-frame overload with match12)
        for (int i = 0; i < n; ++i) {
            MapPoint& p = pool[i];
            const float x = p.world.at<float>(0), y = p.world.at<float>(1), z = p.world.at<float>(2), d = std::sqrt(x * x + y * y + z * z);
            p.normal.create(3, 1, 5);
            p.normal.at<float>(0) = x / d; p.normal.at<float>(1) = y / d; p.normal.at<float>(2) = z / d;
            const float mx = d * std::pow(1.2f, (float)cur.mvKeysUn[i].octave + 0.5f);         // PredictScale -> octave + 1: accepts the key point's octave
            p.setDistances(mx, mx / std::pow(1.2f, 8.f));
            p.observers.clear(); p.bad = false; p.nobs = 1;
        }
        auto make_kf = [&](KeyFrame& k, const Frame& F) {
            k.N = F.N; k.mvKeysUn = F.mvKeysUn; k.mvuRight = F.mvuRight; k.mDescriptors = F.mDescriptors; k.mFeatVec = FeatureVector();
            for (int i = 0; i < F.N; ++i) k.mFeatVec[(unsigned)(i % 10)].push_back((unsigned)i);
            k.mps.assign(F.N, nullptr); k.mvScaleFactors = F.mvScaleFactors;
            k.Tcw.create(4, 4, 5); for (int i = 0; i < 4; ++i) k.Tcw.at<float>(i, i) = 1.f;
        };
        // int SearchByProjection(Frame&, const Frame&, th, bMono, map<int,int>& match12)                          src/Tracking.cc:1296,1302
        {
            Frame c2; fill_frame(c2, n); c2.mvKeysUn = cur.mvKeysUn; c2.mDescriptors = cur.mDescriptors;
            std::map<int, int> match12; match12[-5] = 1;
            int n12 = -1;
            try { n12 = matcher.SearchByProjection(c2, last, 7.f, false, match12); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
            if (gpu) {
                if (n12 != nm || (int)match12.size() != nm || match12.count(-5)) { std::printf("match12: %d matches, %d keys, want %d\n", n12, (int)match12.size(), nm); return 50; }
                for (const auto& kv : match12) if (c2.mvpMapPoints[kv.first] != last.mvpMapPoints[kv.second]) return 51;
            }
        }
        KeyFrame kA; make_kf(kA, cur);
        for (int i = 0; i < n; ++i) if (i % 3) kA.mps[i] = &pool[i];
        // int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, th, ORBdist)   src/Tracking.cc:2322,2336
        {
            Frame c3; fill_frame(c3, n); c3.mvKeysUn = cur.mvKeysUn; c3.mDescriptors = cur.mDescriptors;
            std::set<MapPoint*> sFound; sFound.insert(&pool[1]); sFound.insert(&pool[2]);
            int nr = -1;
            try { nr = matcher.SearchByProjection(c3, &kA, sFound, 10.f, 100); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
            if (gpu) {
                if (nr <= 0 || c3.mvpMapPoints[1] || c3.mvpMapPoints[2]) { std::printf("SearchByProjection(Frame, KeyFrame): %d\n", nr); return 52; }
                for (int i = 0; i < n; ++i) if (c3.mvpMapPoints[i] && c3.mvpMapPoints[i] != &pool[i]) return 53;
            }
        }
        // int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, vpPoints, vpMatched, th)                               src/LoopClosing.cc:381
        Mat Scw; Scw.create(4, 4, 5); for (int i = 0; i < 4; ++i) Scw.at<float>(i, i) = 1.f;
        {
            KeyFrame kB; make_kf(kB, cur);
            std::vector<MapPoint*> pts, matched(n, nullptr);
            for (int i = 0; i < n; ++i) pts.push_back(&pool[i]);
            matched[4] = &pool[4]; matched[5] = &pool[7];
            int ns = -1;
            try { ns = matcher.SearchByProjection(&kB, Scw, pts, matched, 10); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
            if (gpu) {
                if (ns <= 0 || matched[5] != &pool[7]) { std::printf("SearchByProjection(KeyFrame, Scw): %d\n", ns); return 54; }
                int cnt = 0;
                for (int i = 0; i < n; ++i) { if (i != 5 && matched[i] && matched[i] != &pool[i]) return 55; cnt += matched[i] != nullptr; }
                if (cnt != ns + 2) return 56;
            }
        }
        // int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12)                            src/LoopClosing.cc:271
        KeyFrame kC; make_kf(kC, cur);
        for (int i = 0; i < n; ++i) if (i % 2) kC.mps[i] = &pool[i];
        {
            std::vector<MapPoint*> v12;
            int nk = -1;
            try { nk = matcher.SearchByBoW(&kA, &kC, v12); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
            if (gpu) {
                if (nk <= 0 || (int)v12.size() != n) { std::printf("SearchByBoW(KF, KF): %d\n", nk); return 57; }
                for (int i = 0; i < n; ++i) if (v12[i] && (v12[i] != &pool[i] || !kA.mps[i] || !kC.mps[i])) return 58;
            }
        }
        // int SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo)                                     src/LocalMapping.cc:268
        {
            KeyFrame k1, k2; make_kf(k1, cur); make_kf(k2, cur);
            k2.Tcw.at<float>(0, 3) = -0.5f;                                   // a baseline along x: F12 = [t]x for identity rotations, rows of equal y match
            Mat F12; F12.create(3, 3, 5);
            F12.at<float>(1, 2) = -1.f; F12.at<float>(2, 1) = 1.f;
            std::vector<std::pair<size_t, size_t>> pairs(3);
            int nt = -1;
            try { nt = matcher.SearchForTriangulation(&k1, &k2, F12, pairs, false); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
            if (gpu) {
                if (nt < 0 || (int)pairs.size() != nt) { std::printf("SearchForTriangulation: %d, %d pairs\n", nt, (int)pairs.size()); return 59; }
                for (size_t k = 1; k < pairs.size(); ++k) if (pairs[k].first <= pairs[k - 1].first) return 60;      // idx1 ascending, as the reference builds the list
            }
        }
        // int Fuse(pKF, vpMapPoints, th) and int Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)                          src/LocalMapping.cc:489, LoopClosing.cc:605
        {
            KeyFrame kF; make_kf(kF, cur);
            MapPoint other; other.nobs = 5;
            kF.mps[9] = &other;                                                // key point 9 already holds a point with more observations
            std::vector<MapPoint*> pts;
            for (int i = 0; i < n; ++i) pts.push_back(&pool[i]);
            pts.push_back(nullptr);
            int nf = -1;
            try { nf = matcher.Fuse(&kF, pts, 3.f); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
            if (gpu) {
                if (nf <= 0) { std::printf("Fuse: %d\n", nf); return 61; }
                int added = 0;
                for (int i = 0; i < n; ++i) { if (i != 9 && kF.mps[i]) { if (kF.mps[i] != &pool[i] || !pool[i].IsInKeyFrame(&kF)) return 62; ++added; } }
                if (!pool[9].bad || pool[9].replaced != &other || added + 1 != nf) { std::printf("Fuse: %d fused, %d added\n", nf, added); return 63; }
                pool[9].bad = false; pool[9].replaced = nullptr;
            }
            for (int i = 0; i < n; ++i) { pool[i].observers.clear(); pool[i].nobs = 1; }
            KeyFrame kG; make_kf(kG, cur);
            kG.mps[11] = &other;
            std::vector<MapPoint*> repl(pts.size(), nullptr);
            int ng = -1;
            pts.pop_back();
            try { ng = matcher.Fuse(&kG, Scw, pts, 4.f, repl); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
            if (gpu) {
                if (ng <= 0 || repl[11] != &other) { std::printf("Fuse(Scw): %d\n", ng); return 64; }
                for (int i = 0; i < n; ++i) if (i != 11 && kG.mps[i] && kG.mps[i] != &pool[i]) return 65;
            }
            for (int i = 0; i < n; ++i) { pool[i].observers.clear(); pool[i].nobs = 1; }
        }
        // int SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th)                                                  src/LoopClosing.cc:329
        {
            KeyFrame k1, k2; make_kf(k1, cur); make_kf(k2, cur);
            for (int i = 0; i < n; ++i) { k1.mps[i] = (i % 4) ? &pool[i] : nullptr; k2.mps[i] = (i % 5) ? &pool[i] : nullptr; if (k2.mps[i]) pool[i].observers[&k2] = (size_t)i; }
            std::vector<MapPoint*> v12(n, nullptr);
            v12[1] = &pool[1];                                                 // already matched (to key point 1 of k2)
            Mat R12, t12; R12.create(3, 3, 5); t12.create(3, 1, 5);
            for (int i = 0; i < 3; ++i) R12.at<float>(i, i) = 1.f;
            int n3 = -1;
            const float s12 = 1.f;
            try { n3 = matcher.SearchBySim3(&k1, &k2, v12, s12, R12, t12, 7.5f); } catch (const std::runtime_error& e) { ++thrown; if (gpu) std::printf("threw: %s\n", e.what()); }
            if (gpu) {
                if (n3 <= 0) { std::printf("SearchBySim3: %d\n", n3); return 66; }
                int cnt = 0;
                for (int i = 0; i < n; ++i) if (v12[i]) { if (v12[i] != &pool[i] || !k1.mps[i] || !k2.mps[i]) return 67; ++cnt; }
                if (cnt != n3 + 1) return 68;
            }
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
        if (gpu) {
            // Config::hasLines() == false (src/LineExtractor.cc:37, src/Frame.cc:203): the same points, no line member touched by the stereo step
            ORB_SLAM2::AdaptorConfig::hasLines() = false;
            Frame G;
            G.mpORBextractorLeft = &orbL; G.mpORBextractorRight = &orbR; G.mpLineextractorLeft = &lineL; G.mpLineextractorRight = &lineR;
            ORB_SLAM2::StereoFrameFeatures(G, L, R);
            std::vector<olf_keyline> kl0; std::vector<uint8_t> ld0;
            lineL.extract(L.data, 640, 480, 640, kl0, ld0);
            ORB_SLAM2::AdaptorConfig::hasLines() = true;
            if (G.N != F.N || std::memcmp(G.mvKeys.data(), F.mvKeys.data(), (size_t)F.N * sizeof(olf_keypoint)) != 0 || G.mvuRight != F.mvuRight) return 27;
            if (G.N_l != 0 || !G.mvKeys_Line.empty() || !G.mvDisparity_l.empty() || !kl0.empty()) return 28;
            // a frame without key points: the constructor returns before the stereo step (src/Frame.cc:176-177)
            Mat flatL(480, 640, 0), flatR(480, 640, 0);
            for (int y = 0; y < 480; ++y) for (int x = 0; x < 640; ++x) { flatL.at<uint8_t>(y, x) = 90; flatR.at<uint8_t>(y, x) = 90; }
            Frame Z;
            Z.mpORBextractorLeft = &orbL; Z.mpORBextractorRight = &orbR; Z.mpLineextractorLeft = &lineL; Z.mpLineextractorRight = &lineR;
            Z.mvuRight.assign(3, 7.f);
            ORB_SLAM2::StereoFrameFeatures(Z, flatL, flatR);
            if (Z.N != 0 || !Z.mvKeys.empty() || Z.mvuRight.size() != 3 || Z.N_l != (int)Z.mvKeys_Line.size()) return 29;
        }
        Mat small(100, 100, 0);
        try { ORB_SLAM2::StereoFrameFeatures(F, L, small); return 26; } catch (const std::runtime_error&) {}      // size mismatch throws, src/Frame.cc:145-146
    }
    if (!gpu && thrown != 16) { std::printf("no device: %d of 16 device calls threw\n", thrown); return 40; }
    if (gpu && thrown != 0) return 41;
    std::printf(gpu ? "REFERENCE_API_OK %d %d %d\n" : "REFERENCE_API_COMPILED %d %d %d\n", nm, nl, nb);
    return 0;
}
