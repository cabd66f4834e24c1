// orbline_adaptor.hpp -- header-only C++ adaptor that re-creates the reference's class surfaces on top of the
// C ABI (include/orbline.h), so that Frame.cc / Tracking.cc of ORB_Line_SLAM compile against it unchanged:
//   ORB_SLAM2::ORBextractor   include/ORBextractor.h:52-118
//   ORB_SLAM2::Lineextractor  include/LineExtractor.h:40-72
//   ORB_SLAM2::ORBmatcher::DescriptorDistance, ORB_SLAM2::match / matchNNR / distance  (include/ORBmatcher.h,
//   include/LineMatcher.h:57-69)
// With OpenCV present define ORBLINE_WITH_OPENCV before including: the classes then take/return cv::Mat,
// cv::KeyPoint and cv::line_descriptor::KeyLine exactly like the reference (olf_keypoint / olf_keyline are
// layout-identical, so the conversion is a memcpy).  Without it (this repository's own build and tests, which
// have no OpenCV) the same classes work on the POD records.
#pragma once
#include <cstring>
#include <stdexcept>
#include <type_traits>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>
#include "orbline.h"

#ifdef ORBLINE_WITH_OPENCV
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <line_descriptor_custom.hpp>
static_assert(sizeof(cv::KeyPoint) == sizeof(olf_keypoint), "cv::KeyPoint layout");
static_assert(sizeof(cv::line_descriptor::KeyLine) == sizeof(olf_keyline), "KeyLine layout");
#endif

namespace ORB_SLAM2 {

// Parameters the reference reads from its Config singleton inside the functions mirrored here (src/Config.cpp:45,26-160,215-216).  Inside the reference
// tree define ORBLINE_CONFIG to the reference's class before including this header (it has the same static accessors): #define ORBLINE_CONFIG Config
struct AdaptorConfig {
    static bool& hasLines() { static bool v = true; return v; }            // src/LineExtractor.cc:37, src/Frame.cc:203
    static bool& bestLRMatches() { static bool v = true; return v; }
    static double& minRatio12P() { static double v = 0.75; return v; }
    static double& lineSimTh() { static double v = 0.75; return v; }
};
#ifndef ORBLINE_CONFIG
#define ORBLINE_CONFIG AdaptorConfig
#endif

namespace olf_detail {
inline void check(int rc, const char* where)
{
    if (rc != OLF_OK) throw std::runtime_error(std::string(where) + ": " + olf_last_error());
}
// one context per (object, image size); created lazily like the reference's lazily sized cv::Mat buffers
struct Ctx {
    olf_ctx* h = nullptr;
    olf_params p;
    int w = 0, hgt = 0;
    Ctx() { olf_default_params(&p); }
    ~Ctx() { if (h) olf_ctx_destroy(h); }
    olf_ctx* get(int width, int height, const olf_params* with = nullptr)
    {
        // `with`: the full parameter block of a fused call (this object's own part plus the line / stereo parameters of its siblings)
        if (with && std::memcmp(with, &p, sizeof(p)) != 0) { p = *with; if (h) olf_ctx_destroy(h); h = nullptr; }
        if (!h || w != width || hgt != height) {
            if (h) olf_ctx_destroy(h);
            h = nullptr;
            check(olf_ctx_create(&p, width, height, 2, &h), "olf_ctx_create");
            w = width; hgt = height;
        }
        return h;
    }
    Ctx(const Ctx&) = delete;
    Ctx& operator=(const Ctx&) = delete;
};
}  // namespace olf_detail

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
    {
        c.p.orb.nfeatures = nfeatures; c.p.orb.scale_factor = scaleFactor; c.p.orb.nlevels = nlevels;
        c.p.orb.ini_th_fast = iniThFAST; c.p.orb.min_th_fast = minThFAST;
        scale_.assign(nlevels, 1.f); inv_.assign(nlevels, 1.f); s2_.assign(nlevels, 1.f); is2_.assign(nlevels, 1.f);
    }
    // POD form: image w x h, row stride = w.  Mask is ignored, as in the reference.
    void operator()(const uint8_t* image, int w, int h, std::vector<olf_keypoint>& keypoints, std::vector<uint8_t>& descriptors)
    {
        extract(image, w, h, (size_t)w, keypoints, descriptors);
    }
    // rows `stride` bytes apart (cv::Mat::step)
    void extract(const uint8_t* image, int w, int h, size_t stride, std::vector<olf_keypoint>& keypoints, std::vector<uint8_t>& descriptors)
    {
        keypoints.clear(); descriptors.clear();
        if (!image || w <= 0 || h <= 0) return;   // src/ORBextractor.cc:1048-1049
        olf_ctx* x = c.get(w, h);
        const int cap = olf_orb_capacity(x);
        keypoints.resize(cap); descriptors.resize((size_t)cap * OLF_DESC_BYTES);
        int32_t n = 0;
        olf_detail::check(olf_orb_extract_strided(x, image, stride, keypoints.data(), descriptors.data(), &n), "olf_orb_extract");
        keypoints.resize(n); descriptors.resize((size_t)n * OLF_DESC_BYTES);
        olf_orb_scale_tables(x, scale_.data(), inv_.data(), s2_.data(), is2_.data(), nullptr);
    }
#ifdef ORBLINE_WITH_OPENCV
    void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray _descriptors)
    {
        if (_image.empty()) return;
        cv::Mat image = _image.getMat();
        CV_Assert(image.type() == CV_8UC1);
        std::vector<olf_keypoint> k; std::vector<uint8_t> d;
        extract(image.data, image.cols, image.rows, image.step, k, d);       // a ROI keeps its stride: no host-side clone
        keypoints.resize(k.size());
        if (!k.empty()) std::memcpy(keypoints.data(), k.data(), k.size() * sizeof(olf_keypoint));
        if (k.empty()) _descriptors.release();
        else { _descriptors.create((int)k.size(), 32, CV_8U); std::memcpy(_descriptors.getMat().data, d.data(), d.size()); }
        mvImagePyramid.resize(GetLevels());
        std::vector<int32_t> lw(GetLevels()), lh(GetLevels());
        olf_orb_level_sizes(c.h, lw.data(), lh.data());
        for (int l = 0; l < GetLevels(); ++l) {   // the reference exposes the pyramid as a public member (src/Frame.cc:799-816 reads it)
            mvImagePyramid[l].create(lh[l], lw[l], CV_8U);
            olf_orb_pyramid_level(c.h, 0, l, 0, mvImagePyramid[l].data);
        }
    }
    std::vector<cv::Mat> mvImagePyramid;
#endif
    int GetLevels() { return c.p.orb.nlevels; }
    float GetScaleFactor() { return c.p.orb.scale_factor; }
    std::vector<float> GetScaleFactors() { ensure(); return scale_; }
    std::vector<float> GetInverseScaleFactors() { ensure(); return inv_; }
    std::vector<float> GetScaleSigmaSquares() { ensure(); return s2_; }
    std::vector<float> GetInverseScaleSigmaSquares() { ensure(); return is2_; }
    // mvImagePyramid[level] of the last call (w*h bytes)
    std::vector<uint8_t> PyramidLevel(int level, int* w = nullptr, int* h = nullptr)
    {
        std::vector<int32_t> lw(GetLevels()), lh(GetLevels());
        olf_detail::check(olf_orb_level_sizes(c.h, lw.data(), lh.data()), "olf_orb_level_sizes");
        std::vector<uint8_t> out((size_t)lw[level] * lh[level]);
        olf_detail::check(olf_orb_pyramid_level(c.h, 0, level, 0, out.data()), "olf_orb_pyramid_level");
        if (w) *w = lw[level];
        if (h) *h = lh[level];
        return out;
    }
    olf_ctx* context(int w, int h, const olf_params* with = nullptr) { return c.get(w, h, with); }
    const olf_params& params() const { return c.p; }

private:
    void ensure()
    {   // the scale chain does not depend on the image size; any context computes it
        olf_ctx* x = c.h ? c.h : c.get(640, 480);
        olf_orb_scale_tables(x, scale_.data(), inv_.data(), s2_.data(), is2_.data(), nullptr);
    }
    olf_detail::Ctx c;
    std::vector<float> scale_, inv_, s2_, is2_;
};

class Lineextractor {
public:
    Lineextractor(int lsd_nfeatures, double llength_th, bool bFLD_ = false) : bFLD(bFLD_)
    {
        c.p.line.lsd_nfeatures = lsd_nfeatures; c.p.line.min_line_length = llength_th;
    }
    Lineextractor(int lsd_nfeatures, double llength_th, int lsd_refine, double lsd_scale, double lsd_sigma_scale, double lsd_quant,
                  double lsd_ang_th, double lsd_log_eps, double lsd_density_th, int lsd_n_bins, bool bFLD_ = false)
        : bFLD(bFLD_)
    {
        olf_line_params& l = c.p.line;
        l.lsd_nfeatures = lsd_nfeatures; l.min_line_length = llength_th; l.lsd_refine = lsd_refine; l.lsd_scale = lsd_scale;
        l.lsd_sigma_scale = lsd_sigma_scale; l.lsd_quant = lsd_quant; l.lsd_ang_th = lsd_ang_th; l.lsd_log_eps = lsd_log_eps;
        l.lsd_density_th = lsd_density_th; l.lsd_n_bins = lsd_n_bins;
    }
    void operator()(const uint8_t* image, int w, int h, std::vector<olf_keyline>& keylines, std::vector<uint8_t>& descriptors)
    {
        extract(image, w, h, (size_t)w, keylines, descriptors);
    }
    void extract(const uint8_t* image, int w, int h, size_t stride, std::vector<olf_keyline>& keylines, std::vector<uint8_t>& descriptors)
    {
        keylines.clear(); descriptors.clear();
        if (!ORBLINE_CONFIG::hasLines() || bFLD || !image) return;      // src/LineExtractor.cc:37 (Config::hasLines()), :68 (bFLD)
        olf_ctx* x = c.get(w, h);
        const int cap = olf_line_capacity(x);
        keylines.resize(cap); descriptors.resize((size_t)cap * OLF_DESC_BYTES);
        int32_t n = 0;
        olf_detail::check(olf_line_extract_strided(x, image, stride, keylines.data(), descriptors.data(), &n), "olf_line_extract");
        keylines.resize(n); descriptors.resize((size_t)n * OLF_DESC_BYTES);
    }
#ifdef ORBLINE_WITH_OPENCV
    void operator()(const cv::Mat& image, const cv::Mat& /*mask*/, std::vector<cv::line_descriptor::KeyLine>& keylines, cv::Mat& descriptors_line)
    {
        keylines.clear();
        if (!ORBLINE_CONFIG::hasLines() || bFLD) return;      // the reference returns with descriptors_line untouched (src/LineExtractor.cc:35-37,68)
        if (image.depth() != 0) throw std::runtime_error("Error, depth image!= 0");   // LSDDetector_custom.cpp:236-237
        std::vector<olf_keyline> k; std::vector<uint8_t> d;
        extract(image.data, image.cols, image.rows, image.step, k, d);
        keylines.resize(k.size());
        if (!k.empty()) std::memcpy((void*)keylines.data(), k.data(), k.size() * sizeof(olf_keyline));
        descriptors_line = cv::Mat((int)k.size(), 32, CV_8UC1);
        if (!k.empty()) std::memcpy(descriptors_line.data, d.data(), d.size());
    }
#endif
    const olf_params& params() const { return c.p; }
private:
    olf_detail::Ctx c;
    bool bFLD;
};

// ---- matchers ---------------------------------------------------------------------------------------------------
// int distance(const cv::Mat&, const cv::Mat&) / ORBmatcher::DescriptorDistance: the scalar op stays inline on the host,
// exactly like the reference's bit hack (src/ORBmatcher.cc:1795-1811); the batched searches go through the C ABI.
inline int distance(const uint8_t* a, const uint8_t* b)
{
    int d = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t x, y;
        std::memcpy(&x, a + 8 * i, 8); std::memcpy(&y, b + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;   // src/ORBmatcher.cc:39-41
    ORBmatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
    static int DescriptorDistance(const uint8_t* a, const uint8_t* b) { return distance(a, b); }

    // ---- the reference's own signatures (include/ORBmatcher.h:44-66), templates over the reference's Frame / KeyFrame / MapPoint / cv::Mat;
    // defined in orbline_reference_api.hpp (included at the end of this header): the call sites of Tracking.cc compile unchanged
    template <class MatT, class = typename std::enable_if<std::is_class<MatT>::value>::type> static int DescriptorDistance(const MatT& a, const MatT& b);
    template <class FrameT> int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono);
    template <class FrameT, class MapPointT> int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th = 3);
    template <class KeyFrameT, class FrameT, class MapPointT> int SearchByBoW(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches);
    template <class FrameT, class Point2fT> int SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<Point2fT>& vbPrevMatched, std::vector<int>& vnMatches12,
                                                                        int windowSize = 10);
    // the rest of include/ORBmatcher.h:37-103 (Tracking.cc:1296,1302,2322,2336; LocalMapping.cc:268,489,514; LoopClosing.cc:271,329,381,605)
    template <class FrameT> int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono, std::map<int, int>& match12);
    template <class FrameT, class KeyFrameT, class MapPointT>
    int SearchByProjection(FrameT& CurrentFrame, KeyFrameT* pKF, const std::set<MapPointT*>& sAlreadyFound, const float th, const int ORBdist);
    template <class KeyFrameT, class MatT, class MapPointT>
    int SearchByProjection(KeyFrameT* pKF, MatT Scw, const std::vector<MapPointT*>& vpPoints, std::vector<MapPointT*>& vpMatched, int th);
    template <class KeyFrameT, class MapPointT> int SearchByBoW(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12);
    template <class KeyFrameT, class MatT>
    int SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, MatT F12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo);
    template <class KeyFrameT, class MapPointT> int Fuse(KeyFrameT* pKF, const std::vector<MapPointT*>& vpMapPoints, const float th = 3.0);
    template <class KeyFrameT, class MatT, class MapPointT>
    int Fuse(KeyFrameT* pKF, MatT Scw, const std::vector<MapPointT*>& vpPoints, float th, std::vector<MapPointT*>& vpReplacePoint);
    template <class KeyFrameT, class MatT, class MapPointT>
    int SearchBySim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12, const float& s12, const MatT& R12, const MatT& t12, const float th);

    // The members of the reference's MapPoints read by SearchByProjection(Frame&, const vector<MapPoint*>&, th), gathered into arrays
    struct TrackedMapPoints {
        int n = 0;
        const uint8_t* mbTrackInView = nullptr; const uint8_t* isBad = nullptr; const int32_t* mnTrackScaleLevel = nullptr;
        const float* mTrackViewCos = nullptr; const float* mTrackProjXYR = nullptr;     // (mTrackProjX, mTrackProjY, mTrackProjXR) per point
        const uint8_t* descriptor = nullptr; const uint8_t* observed = nullptr;        // GetDescriptor(), Observations() > 0
    };
    // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono), src/ORBmatcher.cc:1330-1472.
    // The frames are olf_frame_view gathers of the Frame members; matches[i2] = LastFrame feature whose map point
    // CurrentFrame feature i2 received.
    int SearchByProjection(olf_ctx* ctx, const olf_frame_view& CurrentFrame, const olf_frame_view& LastFrame, float th, bool bMono,
                           std::vector<int32_t>& matches) const
    {
        matches.assign(CurrentFrame.n, -1);
        int32_t n = 0;
        olf_detail::check(olf_search_by_projection(ctx, &CurrentFrame, &LastFrame, th, bMono ? 1 : 0, mbCheckOrientation ? 1 : 0, matches.data(), &n),
                          "olf_search_by_projection");
        return n;
    }
    // int SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th = 3), src/ORBmatcher.cc:47-131
    int SearchByProjection(olf_ctx* ctx, const olf_frame_view& F, const TrackedMapPoints& vpMapPoints, float th, std::vector<int32_t>& matches) const
    {
        matches.assign(F.n, -1);
        int32_t n = 0;
        const TrackedMapPoints& m = vpMapPoints;
        olf_detail::check(olf_search_local_map(ctx, &F, m.n, m.mbTrackInView, m.isBad, m.mnTrackScaleLevel, m.mTrackViewCos, m.mTrackProjXYR,
                                               m.descriptor, m.observed, th, mfNNratio, matches.data(), &n), "olf_search_local_map");
        return n;
    }
    // int SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12, int windowSize = 10),
    // src/ORBmatcher.cc:407-522; vbPrevMatched as F1.n (x, y) pairs, updated in place
    int SearchForInitialization(olf_ctx* ctx, const olf_frame_view& F1, const olf_frame_view& F2, float* vbPrevMatched, std::vector<int>& vnMatches12,
                                int windowSize = 10) const
    {
        vnMatches12.assign(F1.n, -1);
        static_assert(sizeof(int) == sizeof(int32_t), "vnMatches12 is written as int32_t");
        int32_t n = 0;
        olf_detail::check(olf_search_for_initialization(ctx, &F1, &F2, vbPrevMatched, windowSize, mfNNratio, mbCheckOrientation ? 1 : 0, vnMatches12.data(), &n),
                          "olf_search_for_initialization");
        return n;
    }
    // int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches), src/ORBmatcher.cc:161-290
    int SearchByBoW(olf_ctx* ctx, const olf_frame_view& KF, const olf_frame_view& F, std::vector<int32_t>& vpMapPointMatches) const
    {
        vpMapPointMatches.assign(F.n, -1);
        int32_t n = 0;
        olf_detail::check(olf_search_by_bow(ctx, &KF, &F, mfNNratio, mbCheckOrientation ? 1 : 0, vpMapPointMatches.data(), &n), "olf_search_by_bow");
        return n;
    }
    // The members of the reference's MapPoints read by the two Fuse searches, gathered into arrays
    struct FuseMapPoints {
        int n = 0;
        const uint8_t* skip = nullptr;             // !pMP || isBad() || IsInKeyFrame(pKF)   (Sim3 form: isBad() || spAlreadyFound.count(pMP))
        const float* world = nullptr; const float* normal = nullptr; const float* mfMaxDistance = nullptr; const float* mfMinDistance = nullptr;
        const uint8_t* descriptor = nullptr;
    };
    // int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist), :1620-1747
    int SearchByProjection(olf_ctx* ctx, const olf_frame_view& CurrentFrame, const olf_frame_view& KF, const uint8_t* sAlreadyFound, float th,
                           int ORBdist, std::vector<int32_t>& matches) const
    {
        matches.assign(CurrentFrame.n, -1);
        int32_t n = 0;
        olf_detail::check(olf_search_by_projection_kf(ctx, &CurrentFrame, &KF, sAlreadyFound, th, ORBdist, mbCheckOrientation ? 1 : 0, matches.data(), &n),
                          "olf_search_by_projection_kf");
        return n;
    }
    // int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12), :524-657 (distinguished from the Frame form by the tag)
    struct KeyFramePair {};
    int SearchByBoW(olf_ctx* ctx, KeyFramePair, const olf_frame_view& KF1, const olf_frame_view& KF2, std::vector<int32_t>& vpMatches12) const
    {
        vpMatches12.assign(KF1.n, -1);
        int32_t n = 0;
        olf_detail::check(olf_search_by_bow_kf(ctx, &KF1, &KF2, mfNNratio, mbCheckOrientation ? 1 : 0, vpMatches12.data(), &n), "olf_search_by_bow_kf");
        return n;
    }
    // int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo), :659-825
    int SearchForTriangulation(olf_ctx* ctx, const olf_frame_view& KF1, const olf_frame_view& KF2, const float* F12, bool bOnlyStereo,
                               std::vector<std::pair<size_t, size_t>>& vMatchedPairs) const
    {
        std::vector<int32_t> m12(KF1.n, -1);
        int32_t n = 0;
        olf_detail::check(olf_search_for_triangulation(ctx, &KF1, &KF2, F12, nullptr, bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, m12.data(), &n),
                          "olf_search_for_triangulation");
        vMatchedPairs.clear();
        for (int i = 0; i < KF1.n; ++i) if (m12[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m12[i]);
        return n;
    }
    // the search of int Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th = 3.0), :827-948: best key point per map
    // point; the caller fuses where bestDist <= TH_LOW (:950-972)
    void FuseSearch(olf_ctx* ctx, const olf_frame_view& KF, const FuseMapPoints& m, float th, std::vector<int32_t>& bestIdx, std::vector<int32_t>& bestDist) const
    {
        bestIdx.assign(m.n, -1); bestDist.assign(m.n, 256);
        olf_detail::check(olf_fuse_search(ctx, &KF, m.n, m.skip, m.world, m.normal, m.mfMaxDistance, m.mfMinDistance, m.descriptor, th, nullptr,
                                          bestIdx.data(), bestDist.data()), "olf_fuse_search");
    }
    // the search of int Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, vector<MapPoint*> &vpReplacePoint), :977-1102
    void FuseSearch(olf_ctx* ctx, const olf_frame_view& KF, const float* Scw, const FuseMapPoints& m, float th, std::vector<int32_t>& bestIdx,
                    std::vector<int32_t>& bestDist) const
    {
        bestIdx.assign(m.n, -1); bestDist.assign(m.n, 2147483647);
        olf_detail::check(olf_fuse_search_sim3(ctx, &KF, Scw, m.n, m.skip, m.world, m.normal, m.mfMaxDistance, m.mfMinDistance, m.descriptor, th,
                                               bestIdx.data(), bestDist.data()), "olf_fuse_search_sim3");
    }
    // int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th), :1104-1328
    int SearchBySim3(olf_ctx* ctx, const olf_frame_view& KF1, const olf_frame_view& KF2, std::vector<int32_t>& vpMatches12, float s12, const float* R12,
                     const float* t12, float th) const
    {
        std::vector<int32_t> v1(KF1.n), v2(KF2.n);
        int32_t n = 0;
        olf_detail::check(olf_search_by_sim3(ctx, &KF1, &KF2, vpMatches12.data(), s12, R12, t12, th, v1.data(), v2.data(), &n), "olf_search_by_sim3");
        return n;
    }
    float mfNNratio;
    bool mbCheckOrientation;
};

// int match(desc1, desc2, nnr, matches_12) with Config::bestLRMatches() passed explicitly (src/LineMatcher.cpp:104-132)
inline int match(olf_ctx* ctx, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, float nnr, bool bestLRMatches, std::vector<int>& matches_12)
{
    matches_12.assign(n1, -1);
    olf_detail::check(olf_match_bf(ctx, desc1, n1, desc2, n2, nnr, bestLRMatches ? 1 : 0, matches_12.data()), "olf_match_bf");
    int m = 0;
    for (int v : matches_12) m += v >= 0;
    return m;
}
inline int matchNNR(olf_ctx* ctx, const uint8_t* desc1, int n1, const uint8_t* desc2, int n2, float nnr, std::vector<int>& matches_12)
{
    return match(ctx, desc1, n1, desc2, n2, nnr, false, matches_12);
}

// ORBVocabulary / LineVocabulary (include/ORBVocabulary.h:30-34 = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>): the tree lives on the
// GPU; BowVec / FeatVec are the caller's DBoW2::BowVector (std::map<WordId, WordValue>) and DBoW2::FeatureVector
// (std::map<NodeId, std::vector<unsigned>>) -- any ordered map with those value types works, entries arrive in ascending key order.
class ORBVocabulary {
public:
    ORBVocabulary() : voc_(nullptr) {}
    ~ORBVocabulary() { olf_voc_destroy(voc_); }
    ORBVocabulary(const ORBVocabulary&) = delete;
    ORBVocabulary& operator=(const ORBVocabulary&) = delete;
    // bool loadFromTextFile(const std::string&), Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1425
    bool loadFromTextFile(const std::string& filename)
    {
        olf_voc_destroy(voc_); voc_ = nullptr;
        return olf_voc_load_text(filename.c_str(), &voc_) == OLF_OK;
    }
    bool empty() const { return size() == 0; }
    unsigned size() const { int nw = 0; return voc_ && olf_voc_info(voc_, nullptr, nullptr, nullptr, nullptr, nullptr, &nw) == OLF_OK ? (unsigned)nw : 0u; }
    // void transform(const std::vector<TDescriptor>& features, BowVector& v, FeatureVector& fv, int levelsup), :1127-1195
    // (Frame::ComputeBoW, src/Frame.cc:585-597: descriptors = the rows of mDescriptors, levelsup = 4)
    template <class BowVec, class FeatVec>
    void transform(olf_ctx* ctx, const uint8_t* descriptors, int n, BowVec& v, FeatVec& fv, int levelsup) const
    {
        v.clear(); fv.clear();
        if (!voc_ || n <= 0) return;
        std::vector<int32_t> ids(n), nodes(n), offs(n + 1), idx(n);
        std::vector<double> vals(n);
        int nb = 0, nf = 0;
        olf_detail::check(olf_bow_transform(ctx, voc_, descriptors, n, levelsup, ids.data(), vals.data(), &nb, nodes.data(), offs.data(), idx.data(), &nf),
                          "olf_bow_transform");
        for (int i = 0; i < nb; ++i) v.insert(v.end(), typename BowVec::value_type(ids[i], vals[i]));
        for (int a = 0; a < nf; ++a)
            fv.insert(fv.end(), typename FeatVec::value_type(nodes[a], typename FeatVec::mapped_type(idx.begin() + offs[a], idx.begin() + offs[a + 1])));
    }
    olf_voc* handle() const { return voc_; }
private:
    olf_voc* voc_;
};
typedef ORBVocabulary LineVocabulary;

}  // namespace ORB_SLAM2

#include "orbline_reference_api.hpp"
