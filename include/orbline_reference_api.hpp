// orbline_reference_api.hpp -- the reference's OWN call signatures on top of the C ABI, so that the call sites of Tracking.cc need no edit:
//
//   matcher.SearchByProjection(mCurrentFrame, mLastFrame, th, bMono, match12)        src/Tracking.cc:1296,1302   (include/ORBmatcher.h:53; the
//                                                                                    four-argument form :52 as well)
//   matcher.SearchByBoW(mpReferenceKF, mCurrentFrame, vpMapPointMatches)             src/Tracking.cc:970         (include/ORBmatcher.h:66)
//   matcher.SearchByProjection(mCurrentFrame, mvpLocalMapPoints, th)                 src/Tracking.cc:1941        (include/ORBmatcher.h:48)
//   match(mLastFrame.mDescriptors_Line, mCurrentFrame.mDescriptors_Line, nnr, m12)   src/Tracking.cc:1308, :979  (include/LineMatcher.h:61)
//   match(mvpLocalMapLines, mCurrentFrame, nnr, m12) / matchNNR / distance           src/Tracking.cc:1970        (include/LineMatcher.h:57-63)
//   matchGrid(points1 | lines1, desc1, grid, desc2, [directions2,] w, matches_12)    src/Frame.cc:926            (include/LineMatcher.h:66-69)
//   GridStructure / GridWindow / getLineCoords                                       include/gridStructure.h:33-58
//   StereoFrameFeatures(frame, imLeft, imRight)  = the feature part of Frame::Frame  src/Frame.cc:164-171,199-207
//   and every other ORBmatcher call of the reference: SearchByProjection(cur, pKF, sFound, th, ORBdist) Tracking.cc:2322,2336;
//   SearchForTriangulation LocalMapping.cc:268; Fuse(pKF, vpMapPoints[, th]) LocalMapping.cc:489,514; SearchByBoW(pKF1, pKF2, vpMatches12)
//   LoopClosing.cc:271; SearchBySim3 LoopClosing.cc:329; SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) LoopClosing.cc:381;
//   Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) LoopClosing.cc:605; SearchForInitialization Tracking.cc:628
//
// Everything is a template over the reference's types (Frame, KeyFrame, MapPoint, MapLine, cv::Mat), used only through the public members
// the reference functions themselves read (include/Frame.h:137-260, include/KeyFrame.h, include/MapPoint.h): the header compiles without
// OpenCV -- tests/adaptor_reference_api.cpp instantiates every template with minimal stand-in structs -- and, inside the reference tree,
// binds to the real classes.  A function gathers the members into an olf_frame_view, makes one C-ABI call and scatters the assignments the
// reference function makes (mvpMapPoints[...] = pMP, NULL on a rotation-histogram rejection).
//
// Contexts: the matcher functions run on a small per-thread context (olf_detail::thread_ctx()) -- ORBmatcher objects are stack locals used
// from the Tracking, LocalMapping and LoopClosing threads concurrently (SURVEY 8(b) "Threading"), and a context serves one thread.
#pragma once
#include <cmath>
#include <list>
#include <map>
#include <set>
#include <unordered_set>
#include <limits>
#include <type_traits>
#include "orbline_adaptor.hpp"

namespace ORB_SLAM2 {

// (AdaptorConfig / ORBLINE_CONFIG -- the Config singleton's accessors read here -- are defined in orbline_adaptor.hpp)

namespace olf_detail {

// one small context per host thread for the matcher calls (their kernels do not depend on the image size; 320 x 240 is about the smallest
// image whose eight pyramid levels all hold a FAST cell)
inline olf_ctx* thread_ctx()
{
    struct Holder { olf_ctx* h = nullptr; ~Holder() { if (h) olf_ctx_destroy(h); } };
    static thread_local Holder t;
    if (!t.h) {
        olf_params p;
        olf_default_params(&p);
        check(olf_ctx_create(&p, 320, 240, 1, &t.h), "olf_ctx_create (matcher context)");
    }
    return t.h;
}

// rows of a continuous N x 32 CV_8U matrix (cv::Mat or anything with rows / cols / data / isContinuous / clone)
template <class MatT> struct DescRows {
    MatT keep;
    const uint8_t* p;
    int n;
    explicit DescRows(const MatT& m) : keep(m.isContinuous() ? m : m.clone()), p(keep.data), n(keep.rows) {}
};

template <class KeyPointVec> const olf_keypoint* keypoints(const KeyPointVec& v)
{
    static_assert(sizeof(typename KeyPointVec::value_type) == sizeof(olf_keypoint), "cv::KeyPoint layout");
    return reinterpret_cast<const olf_keypoint*>(v.data());
}

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>) -> CSR
struct FeatVecCSR {
    std::vector<int32_t> nodes, offs, feats;
    template <class FV> explicit FeatVecCSR(const FV& fv)
    {
        offs.push_back(0);
        for (typename FV::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            nodes.push_back((int32_t)it->first);
            for (size_t k = 0; k < it->second.size(); ++k) feats.push_back((int32_t)it->second[k]);
            offs.push_back((int32_t)feats.size());
        }
    }
    void attach(olf_frame_view& v) const { v.fv_nodes = nodes.data(); v.fv_offsets = offs.data(); v.fv_features = feats.data(); v.fv_n = (int32_t)nodes.size(); }
};

// The members of a reference Frame read by the per-frame searches, gathered once (the arrays live as long as the object)
template <class FrameT> struct FrameGather {
    olf_frame_view v;
    std::vector<uint8_t> valid, obs, bad, outlier, mpdesc;
    std::vector<float> world;
    float Tcw[16];
    DescRows<decltype(FrameT::mDescriptors)> desc;
    // with_points: also gather GetWorldPos() / GetDescriptor() of the frame's map points (what SearchByProjection(cur, last) reads of `last`)
    FrameGather(const FrameT& F, bool with_points) : desc(F.mDescriptors)
    {
        v = olf_frame_view();
        const int n = F.N;
        v.keys = keypoints(F.mvKeysUn); v.desc = desc.p; v.uright = F.mvuRight.empty() ? nullptr : F.mvuRight.data(); v.n = n;
        valid.assign(n, 0); obs.assign(n, 0); bad.assign(n, 0); outlier.assign(n, 0);
        if (with_points) { world.assign((size_t)3 * n, 0.f); mpdesc.assign((size_t)32 * n, 0); }
        for (int i = 0; i < n; ++i) {
            auto* pMP = F.mvpMapPoints[i];
            outlier[i] = i < (int)F.mvbOutlier.size() && F.mvbOutlier[i] ? 1 : 0;
            if (!pMP) continue;
            valid[i] = 1; obs[i] = pMP->Observations() > 0 ? 1 : 0; bad[i] = pMP->isBad() ? 1 : 0;
            if (with_points) {
                const auto wp = pMP->GetWorldPos();
                for (int k = 0; k < 3; ++k) world[3 * i + k] = wp.template at<float>(k);
                const auto d = pMP->GetDescriptor();
                std::memcpy(&mpdesc[(size_t)32 * i], d.data, 32);
            }
        }
        v.mp_valid = valid.data(); v.mp_obs = obs.data(); v.mp_bad = bad.data(); v.outlier = outlier.data();
        if (with_points) { v.mp_world = world.data(); v.mp_desc = mpdesc.data(); }
        if (!F.mTcw.empty()) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Tcw[4 * r + c] = F.mTcw.template at<float>(r, c); v.Tcw = Tcw; }
        v.fx = F.fx; v.fy = F.fy; v.cx = F.cx; v.cy = F.cy; v.mbf = F.mbf;
        v.minX = F.mnMinX; v.maxX = F.mnMaxX; v.minY = F.mnMinY; v.maxY = F.mnMaxY;
        v.scale_factors = F.mvScaleFactors.data(); v.n_levels = F.mnScaleLevels;
    }
};

}  // namespace olf_detail

// ---- ORBmatcher: the member templates declared in orbline_adaptor.hpp -------------------------------------------------------------------------------------
// static int DescriptorDistance(const cv::Mat &a, const cv::Mat &b), include/ORBmatcher.h:44
template <class MatT, class> int ORBmatcher::DescriptorDistance(const MatT& a, const MatT& b)
{
    return distance(static_cast<const uint8_t*>(a.data), static_cast<const uint8_t*>(b.data));
}

// int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono), src/ORBmatcher.cc:1330-1472
template <class FrameT> int ORBmatcher::SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono)
{
    {
        olf_detail::FrameGather<FrameT> cur(CurrentFrame, false), last(LastFrame, true);
        const std::vector<uint8_t> before = cur.valid;
        std::vector<int32_t> m;
        const int n = SearchByProjection(olf_detail::thread_ctx(), cur.v, last.v, th, bMono, m);
        for (int i2 = 0; i2 < CurrentFrame.N; ++i2) {
            if (m[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[m[i2]];
            else if (before[i2] && !cur.valid[i2]) CurrentFrame.mvpMapPoints[i2] = nullptr;      // assigned, then dropped by the rotation histogram (:1459)
        }
        return n;
    }
}

// int SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th = 3), src/ORBmatcher.cc:47-131
template <class FrameT, class MapPointT> int ORBmatcher::SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, const float th)
{
    {
        olf_detail::FrameGather<FrameT> f(F, false);
        const int n_mp = (int)vpMapPoints.size();
        std::vector<uint8_t> inView(n_mp, 0), bad(n_mp, 0), observed(n_mp, 0), desc((size_t)32 * n_mp, 0);
        std::vector<int32_t> level(n_mp, 0);
        std::vector<float> viewCos(n_mp, 0.f), proj((size_t)3 * n_mp, 0.f);
        for (int i = 0; i < n_mp; ++i) {
            MapPointT* pMP = vpMapPoints[i];
            if (!pMP) { bad[i] = 1; continue; }            // (the reference dereferences every entry; a null entry can only be skipped)
            inView[i] = pMP->mbTrackInView ? 1 : 0; bad[i] = pMP->isBad() ? 1 : 0; level[i] = pMP->mnTrackScaleLevel; viewCos[i] = pMP->mTrackViewCos;
            proj[3 * i] = pMP->mTrackProjX; proj[3 * i + 1] = pMP->mTrackProjY; proj[3 * i + 2] = pMP->mTrackProjXR;
            observed[i] = pMP->Observations() > 0 ? 1 : 0;
            const auto d = pMP->GetDescriptor();
            std::memcpy(&desc[(size_t)32 * i], d.data, 32);
        }
        TrackedMapPoints t;
        t.n = n_mp; t.mbTrackInView = inView.data(); t.isBad = bad.data(); t.mnTrackScaleLevel = level.data(); t.mTrackViewCos = viewCos.data();
        t.mTrackProjXYR = proj.data(); t.descriptor = desc.data(); t.observed = observed.data();
        std::vector<int32_t> m;
        const int n = SearchByProjection(olf_detail::thread_ctx(), f.v, t, th, m);
        for (int idx = 0; idx < F.N; ++idx) if (m[idx] >= 0) F.mvpMapPoints[idx] = vpMapPoints[m[idx]];
        return n;
    }
}

// int SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize = 10),
// src/ORBmatcher.cc:407-522 (Tracking::MonocularInitialization, src/Tracking.cc:628)
template <class FrameT, class Point2fT> int ORBmatcher::SearchForInitialization(FrameT& F1, FrameT& F2, std::vector<Point2fT>& vbPrevMatched,
                                                                                  std::vector<int>& vnMatches12, int windowSize)
{
    static_assert(sizeof(Point2fT) == 2 * sizeof(float), "cv::Point2f is two floats");
    olf_detail::FrameGather<FrameT> f1(F1, false), f2(F2, false);
    if ((int)vbPrevMatched.size() < F1.N) throw std::runtime_error("SearchForInitialization: vbPrevMatched shorter than F1.mvKeysUn");
    return SearchForInitialization(olf_detail::thread_ctx(), f1.v, f2.v, reinterpret_cast<float*>(vbPrevMatched.data()), vnMatches12, windowSize);
}

// int SearchByBoW(KeyFrame* pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches), src/ORBmatcher.cc:161-290
template <class KeyFrameT, class FrameT, class MapPointT> int ORBmatcher::SearchByBoW(KeyFrameT* pKF, FrameT& F, std::vector<MapPointT*>& vpMapPointMatches)
{
    {
        const std::vector<MapPointT*> vpMapPointsKF = pKF->GetMapPointMatches();
        vpMapPointMatches = std::vector<MapPointT*>(F.N, static_cast<MapPointT*>(nullptr));
        olf_frame_view kf = olf_frame_view(), f = olf_frame_view();
        olf_detail::DescRows<decltype(KeyFrameT::mDescriptors)> dk(pKF->mDescriptors);
        olf_detail::DescRows<decltype(FrameT::mDescriptors)> df(F.mDescriptors);
        const int nk = (int)vpMapPointsKF.size();
        std::vector<uint8_t> valid(nk, 0), bad(nk, 0), fvalid(F.N, 0), fobs(F.N, 0);
        for (int i = 0; i < nk; ++i) if (vpMapPointsKF[i]) { valid[i] = 1; bad[i] = vpMapPointsKF[i]->isBad() ? 1 : 0; }
        kf.keys = olf_detail::keypoints(pKF->mvKeysUn); kf.desc = dk.p; kf.n = nk; kf.mp_valid = valid.data(); kf.mp_bad = bad.data();
        f.keys = olf_detail::keypoints(F.mvKeysUn); f.desc = df.p; f.n = F.N; f.mp_valid = fvalid.data(); f.mp_obs = fobs.data();
        const olf_detail::FeatVecCSR ck(pKF->mFeatVec), cf(F.mFeatVec);
        ck.attach(kf); cf.attach(f);
        std::vector<int32_t> m;
        const int n = SearchByBoW(olf_detail::thread_ctx(), kf, f, m);
        for (int iF = 0; iF < F.N; ++iF) if (m[iF] >= 0) vpMapPointMatches[iF] = vpMapPointsKF[m[iF]];
        return n;
    }
}

// ---- the rest of the ORBmatcher surface (include/ORBmatcher.h:37-103): the searches of the relocaliser, LocalMapping and LoopClosing ----------
namespace olf_detail {

// MapPoint::mfMaxDistance / mfMinDistance are protected in the reference (include/MapPoint.h:145-146); the searches need the raw values
// (PredictScale divides mfMaxDistance by the distance, src/MapPoint.cc:397-412).  Inside the reference tree either declare
//     friend struct ORB_SLAM2::olf_detail::MapPointDistances;
// in class MapPoint (one line) -- then the members are read directly -- or nothing: the fall-back inverts the public
// GetMaxDistanceInvariance() = 1.2f * mfMaxDistance / GetMinDistanceInvariance() = 0.8f * mfMinDistance by searching the floats around the
// quotient for one whose product reproduces the returned value (exact except where two neighbouring floats share a product).
struct MapPointDistances {
    template <class MP> static auto maxd(MP* p, int) -> decltype((float)p->mfMaxDistance) { return p->mfMaxDistance; }
    template <class MP> static float maxd(MP* p, long) { return invert(p->GetMaxDistanceInvariance(), 1.2f); }
    template <class MP> static auto mind(MP* p, int) -> decltype((float)p->mfMinDistance) { return p->mfMinDistance; }
    template <class MP> static float mind(MP* p, long) { return invert(p->GetMinDistanceInvariance(), 0.8f); }
    static float invert(float y, float k)
    {
        float x = y / k;
        for (int pass = 0; pass < 2; ++pass) {
            float c = x;
            for (int s = 0; s < 3 && !(k * c == y); ++s) c = std::nextafter(c, pass ? 3.4e38f : -3.4e38f);
            if (k * c == y) return c;
        }
        return x;
    }
};

template <class MatT> void mat4(const MatT& m, float* out16) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out16[4 * r + c] = m.template at<float>(r, c); }

// The members of a reference KeyFrame read by the key-frame searches (include/KeyFrame.h:57-62,92-116,185-231)
template <class KeyFrameT> struct KeyFrameGather {
    typedef typename std::remove_pointer<typename decltype(std::declval<KeyFrameT&>().GetMapPointMatches())::value_type>::type MapPointT;
    olf_frame_view v;
    std::vector<MapPointT*> mps;
    std::vector<uint8_t> valid, obs, bad, mpdesc;
    std::vector<float> world, maxd, mind;
    float Tcw[16];
    DescRows<decltype(KeyFrameT::mDescriptors)> desc;
    FeatVecCSR fv;
    KeyFrameGather(KeyFrameT* pKF, bool with_points) : mps(pKF->GetMapPointMatches()), desc(pKF->mDescriptors), fv(pKF->mFeatVec)
    {
        v = olf_frame_view();
        const int n = (int)mps.size();
        v.keys = keypoints(pKF->mvKeysUn); v.desc = desc.p; v.uright = pKF->mvuRight.empty() ? nullptr : pKF->mvuRight.data(); v.n = n;
        valid.assign(n, 0); obs.assign(n, 0); bad.assign(n, 0);
        if (with_points) { world.assign((size_t)3 * n, 0.f); mpdesc.assign((size_t)32 * n, 0); maxd.assign(n, 0.f); mind.assign(n, 0.f); }
        for (int i = 0; i < n; ++i) {
            MapPointT* pMP = mps[i];
            if (!pMP) continue;
            valid[i] = 1; obs[i] = pMP->Observations() > 0 ? 1 : 0; bad[i] = pMP->isBad() ? 1 : 0;
            if (with_points) {
                const auto wp = pMP->GetWorldPos();
                for (int k = 0; k < 3; ++k) world[3 * i + k] = wp.template at<float>(k);
                const auto d = pMP->GetDescriptor();
                std::memcpy(&mpdesc[(size_t)32 * i], d.data, 32);
                maxd[i] = MapPointDistances::maxd(pMP, 0); mind[i] = MapPointDistances::mind(pMP, 0);
            }
        }
        v.mp_valid = valid.data(); v.mp_obs = obs.data(); v.mp_bad = bad.data();
        if (with_points) { v.mp_world = world.data(); v.mp_desc = mpdesc.data(); v.mp_maxd = maxd.data(); v.mp_mind = mind.data(); }
        mat4(pKF->GetPose(), Tcw); v.Tcw = Tcw;
        v.fx = pKF->fx; v.fy = pKF->fy; v.cx = pKF->cx; v.cy = pKF->cy; v.mbf = pKF->mbf;
        v.minX = (float)pKF->mnMinX; v.maxX = (float)pKF->mnMaxX; v.minY = (float)pKF->mnMinY; v.maxY = (float)pKF->mnMaxY;
        v.scale_factors = pKF->mvScaleFactors.data(); v.n_levels = pKF->mnScaleLevels;
        fv.attach(v);
    }
};

// the members of a list of reference MapPoints read by the two Fuse searches and SearchByProjection(pKF, Scw, ...)
template <class MapPointT> struct PointGather {
    std::vector<uint8_t> skip, desc;
    std::vector<float> world, normal, maxd, mind;
    ORBmatcher::FuseMapPoints m;
    template <class SkipFn> PointGather(const std::vector<MapPointT*>& pts, SkipFn skip_if)
    {
        const size_t n = pts.size();
        skip.assign(n, 1); desc.assign(32 * n, 0); world.assign(3 * n, 0.f); normal.assign(3 * n, 0.f); maxd.assign(n, 0.f); mind.assign(n, 0.f);
        for (size_t i = 0; i < n; ++i) {
            MapPointT* p = pts[i];
            if (!p || skip_if(p)) continue;
            skip[i] = 0;
            const auto w = p->GetWorldPos(); const auto nv = p->GetNormal(); const auto d = p->GetDescriptor();
            for (int k = 0; k < 3; ++k) { world[3 * i + k] = w.template at<float>(k); normal[3 * i + k] = nv.template at<float>(k); }
            std::memcpy(&desc[32 * i], d.data, 32);
            maxd[i] = MapPointDistances::maxd(p, 0); mind[i] = MapPointDistances::mind(p, 0);
        }
        m.n = (int)n; m.skip = skip.data(); m.world = world.data(); m.normal = normal.data(); m.mfMaxDistance = maxd.data(); m.mfMinDistance = mind.data();
        m.descriptor = desc.data();
    }
};
}  // namespace olf_detail

// int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono, map<int,int>& match12),
// src/ORBmatcher.cc:1474-1618 (Tracking::TrackWithMotionModelWithLine, src/Tracking.cc:1296,1302)
template <class FrameT> int ORBmatcher::SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, const float th, const bool bMono, std::map<int, int>& match12)
{
    olf_detail::FrameGather<FrameT> cur(CurrentFrame, false), last(LastFrame, true);
    const std::vector<uint8_t> before = cur.valid;
    std::vector<int32_t> m(CurrentFrame.N, -1), first(CurrentFrame.N, -1);
    int32_t n = 0;
    olf_detail::check(olf_search_by_projection_match12(olf_detail::thread_ctx(), &cur.v, &last.v, th, bMono ? 1 : 0, mbCheckOrientation ? 1 : 0, m.data(),
                                                       first.data(), &n), "olf_search_by_projection_match12");
    match12.clear();
    for (int i2 = 0; i2 < CurrentFrame.N; ++i2) {
        if (m[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = LastFrame.mvpMapPoints[m[i2]];              // the LAST point assigned (:1575)
        else if (before[i2] && !cur.valid[i2]) CurrentFrame.mvpMapPoints[i2] = nullptr;             // dropped by the rotation histogram (:1610)
        if (first[i2] >= 0) match12.insert(match12.end(), std::pair<int, int>(i2, first[i2]));      // the FIRST one inserted (:1577)
    }
    return n;
}

// int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist),
// src/ORBmatcher.cc:1620-1747 (Tracking::Relocalization, src/Tracking.cc:2322,2336)
template <class FrameT, class KeyFrameT, class MapPointT>
int ORBmatcher::SearchByProjection(FrameT& CurrentFrame, KeyFrameT* pKF, const std::set<MapPointT*>& sAlreadyFound, const float th, const int ORBdist)
{
    olf_detail::FrameGather<FrameT> cur(CurrentFrame, false);
    olf_detail::KeyFrameGather<KeyFrameT> kf(pKF, true);
    const std::vector<uint8_t> before = cur.valid;
    std::vector<uint8_t> found(kf.mps.size(), 0);
    for (size_t i = 0; i < kf.mps.size(); ++i) found[i] = kf.mps[i] && sAlreadyFound.count(kf.mps[i]) ? 1 : 0;
    std::vector<int32_t> m;
    const int n = SearchByProjection(olf_detail::thread_ctx(), cur.v, kf.v, found.data(), th, ORBdist, m);
    for (int i2 = 0; i2 < CurrentFrame.N; ++i2) {
        if (m[i2] >= 0) CurrentFrame.mvpMapPoints[i2] = kf.mps[m[i2]];
        else if (before[i2] && !cur.valid[i2]) CurrentFrame.mvpMapPoints[i2] = nullptr;
    }
    return n;
}

// int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th),
// src/ORBmatcher.cc:292-405 (LoopClosing::ComputeSim3, src/LoopClosing.cc:381)
template <class KeyFrameT, class MatT, class MapPointT>
int ORBmatcher::SearchByProjection(KeyFrameT* pKF, MatT Scw, const std::vector<MapPointT*>& vpPoints, std::vector<MapPointT*>& vpMatched, int th)
{
    olf_detail::KeyFrameGather<KeyFrameT> kf(pKF, false);
    std::set<MapPointT*> spAlreadyFound(vpMatched.begin(), vpMatched.end());
    spAlreadyFound.erase(static_cast<MapPointT*>(nullptr));
    const olf_detail::PointGather<MapPointT> pts(vpPoints, [&](MapPointT* p) { return p->isBad() || spAlreadyFound.count(p) != 0; });
    float S[16];
    olf_detail::mat4(Scw, S);
    std::vector<uint8_t> matched(vpMatched.size(), 0);
    for (size_t i = 0; i < vpMatched.size(); ++i) matched[i] = vpMatched[i] ? 1 : 0;
    if ((int)matched.size() != kf.v.n) throw std::runtime_error("SearchByProjection(pKF, Scw, ...): vpMatched must have one entry per key point of pKF");
    std::vector<int32_t> m(kf.v.n, -1);
    int32_t n = 0;
    olf_detail::check(olf_search_by_projection_sim3(olf_detail::thread_ctx(), &kf.v, S, pts.m.n, pts.m.skip, pts.m.world, pts.m.normal, pts.m.mfMaxDistance,
                                                    pts.m.mfMinDistance, pts.m.descriptor, (float)th, matched.data(), m.data(), &n),
                      "olf_search_by_projection_sim3");
    for (int idx = 0; idx < kf.v.n; ++idx) if (m[idx] >= 0) vpMatched[idx] = vpPoints[m[idx]];
    return n;
}

// int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12), src/ORBmatcher.cc:524-657 (LoopClosing.cc:271)
template <class KeyFrameT, class MapPointT> int ORBmatcher::SearchByBoW(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12)
{
    olf_detail::KeyFrameGather<KeyFrameT> k1(pKF1, false), k2(pKF2, false);
    std::vector<int32_t> m;
    const int n = SearchByBoW(olf_detail::thread_ctx(), KeyFramePair(), k1.v, k2.v, m);
    vpMatches12 = std::vector<MapPointT*>(k1.mps.size(), static_cast<MapPointT*>(nullptr));
    for (size_t i = 0; i < k1.mps.size(); ++i) if (m[i] >= 0) vpMatches12[i] = k2.mps[m[i]];
    return n;
}

// int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, cv::Mat F12, vector<pair<size_t, size_t>> &vMatchedPairs, const bool bOnlyStereo),
// src/ORBmatcher.cc:659-825 (LocalMapping::CreateNewMapPoints, src/LocalMapping.cc:268)
template <class KeyFrameT, class MatT>
int ORBmatcher::SearchForTriangulation(KeyFrameT* pKF1, KeyFrameT* pKF2, MatT F12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo)
{
    olf_detail::KeyFrameGather<KeyFrameT> k1(pKF1, false), k2(pKF2, false);
    float F[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) F[3 * r + c] = F12.template at<float>(r, c);
    const auto Cw = pKF1->GetCameraCenter();
    const float cw[3] = {Cw.template at<float>(0), Cw.template at<float>(1), Cw.template at<float>(2)};
    std::vector<int32_t> m12(k1.v.n, -1);
    int32_t n = 0;
    olf_detail::check(olf_search_for_triangulation(olf_detail::thread_ctx(), &k1.v, &k2.v, F, cw, bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0, m12.data(), &n),
                      "olf_search_for_triangulation");
    vMatchedPairs.clear();
    vMatchedPairs.reserve(n > 0 ? n : 0);
    for (int i = 0; i < k1.v.n; ++i) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)m12[i]));
    return n;
}

// int Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th = 3.0), src/ORBmatcher.cc:827-975 (LocalMapping.cc:489,514).
// The search of every point runs first (one call); what the reference does with a hit (:950-972: Replace / AddObservation / AddMapPoint) is
// replayed on the reference's own objects in list order.  A hit's (bestIdx, bestDist) depends on nothing that loop changes; whether a point
// is looked at does (isBad() / IsInKeyFrame(pKF) after an earlier Replace or AddObservation), so that gate is evaluated again at its turn.
template <class KeyFrameT, class MapPointT> int ORBmatcher::Fuse(KeyFrameT* pKF, const std::vector<MapPointT*>& vpMapPoints, const float th)
{
    olf_detail::KeyFrameGather<KeyFrameT> kf(pKF, false);
    const olf_detail::PointGather<MapPointT> pts(vpMapPoints, [&](MapPointT* p) { return p->isBad() || p->IsInKeyFrame(pKF); });
    const auto Ow = pKF->GetCameraCenter();
    const float ow[3] = {Ow.template at<float>(0), Ow.template at<float>(1), Ow.template at<float>(2)};
    std::vector<int32_t> bestIdx(pts.m.n, -1), bestDist(pts.m.n, 256);
    olf_detail::check(olf_fuse_search(olf_detail::thread_ctx(), &kf.v, pts.m.n, pts.m.skip, pts.m.world, pts.m.normal, pts.m.mfMaxDistance, pts.m.mfMinDistance,
                                      pts.m.descriptor, th, ow, bestIdx.data(), bestDist.data()), "olf_fuse_search");
    int nFused = 0;
    for (size_t i = 0; i < vpMapPoints.size(); ++i) {
        MapPointT* pMP = vpMapPoints[i];
        if (!pMP || pts.skip[i] || bestDist[i] > TH_LOW) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;
        MapPointT* pMPinKF = pKF->GetMapPoint(bestIdx[i]);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, bestIdx[i]);
            pKF->AddMapPoint(pMP, bestIdx[i]);
        }
        nFused++;
    }
    return nFused;
}

// int Fuse(KeyFrame *pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, float th, vector<MapPoint*> &vpReplacePoint),
// src/ORBmatcher.cc:977-1102 (LoopClosing::SearchAndFuse, src/LoopClosing.cc:605)
template <class KeyFrameT, class MatT, class MapPointT>
int ORBmatcher::Fuse(KeyFrameT* pKF, MatT Scw, const std::vector<MapPointT*>& vpPoints, float th, std::vector<MapPointT*>& vpReplacePoint)
{
    olf_detail::KeyFrameGather<KeyFrameT> kf(pKF, false);
    const std::set<MapPointT*> spAlreadyFound = pKF->GetMapPoints();
    const olf_detail::PointGather<MapPointT> pts(vpPoints, [&](MapPointT* p) { return p->isBad() || spAlreadyFound.count(p) != 0; });
    float S[16];
    olf_detail::mat4(Scw, S);
    std::vector<int32_t> bestIdx, bestDist;
    FuseSearch(olf_detail::thread_ctx(), kf.v, S, pts.m, th, bestIdx, bestDist);
    int nFused = 0;
    for (size_t iMP = 0; iMP < vpPoints.size(); ++iMP) {
        MapPointT* pMP = vpPoints[iMP];
        if (!pMP || pts.skip[iMP] || bestDist[iMP] > TH_LOW) continue;
        MapPointT* pMPinKF = pKF->GetMapPoint(bestIdx[iMP]);
        if (pMPinKF) { if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF; }
        else { pMP->AddObservation(pKF, bestIdx[iMP]); pKF->AddMapPoint(pMP, bestIdx[iMP]); }
        nFused++;
    }
    return nFused;
}

// int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12,
// const float th), src/ORBmatcher.cc:1104-1328 (LoopClosing::ComputeSim3, src/LoopClosing.cc:329)
template <class KeyFrameT, class MatT, class MapPointT>
int ORBmatcher::SearchBySim3(KeyFrameT* pKF1, KeyFrameT* pKF2, std::vector<MapPointT*>& vpMatches12, const float& s12, const MatT& R12, const MatT& t12, const float th)
{
    olf_detail::KeyFrameGather<KeyFrameT> k1(pKF1, true), k2(pKF2, true);
    const int N1 = k1.v.n;
    std::vector<int32_t> m12(N1, -1);
    for (int i = 0; i < N1 && i < (int)vpMatches12.size(); ++i) {
        MapPointT* pMP = vpMatches12[i];
        if (!pMP) continue;
        const int idx2 = pMP->GetIndexInKeyFrame(pKF2);          // (:1139-1151)
        m12[i] = idx2 >= 0 ? idx2 : -2;
    }
    float R[9], t[3];
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[3 * r + c] = R12.template at<float>(r, c); t[r] = t12.template at<float>(r); }
    const std::vector<int32_t> before = m12;
    const int n = SearchBySim3(olf_detail::thread_ctx(), k1.v, k2.v, m12, s12, R, t, th);
    for (int i1 = 0; i1 < N1; ++i1) if (m12[i1] >= 0 && m12[i1] != before[i1]) vpMatches12[i1] = k2.mps[m12[i1]];      // (:1319)
    return n;
}

// ---- LineMatcher free functions (include/LineMatcher.h:57-69) --------------------------------------------------------------------------
// int distance(const cv::Mat &a, const cv::Mat &b)
template <class MatT, class = typename std::enable_if<std::is_class<MatT>::value>::type> int distance(const MatT& a, const MatT& b) { return distance(static_cast<const uint8_t*>(a.data), static_cast<const uint8_t*>(b.data)); }

// int matchNNR(const cv::Mat &desc1, const cv::Mat &desc2, float nnr, std::vector<int> &matches_12), src/LineMatcher.cpp:42-62
template <class MatT> int matchNNR(const MatT& desc1, const MatT& desc2, float nnr, std::vector<int>& matches_12)
{
    const olf_detail::DescRows<MatT> a(desc1), b(desc2);
    return matchNNR(olf_detail::thread_ctx(), a.p, a.n, b.p, b.n, nnr, matches_12);
}
// int match(const cv::Mat &desc1, const cv::Mat &desc2, float nnr, std::vector<int> &matches_12), src/LineMatcher.cpp:104-132
template <class MatT> int match(const MatT& desc1, const MatT& desc2, float nnr, std::vector<int>& matches_12)
{
    const olf_detail::DescRows<MatT> a(desc1), b(desc2);
    return match(olf_detail::thread_ctx(), a.p, a.n, b.p, b.n, nnr, ORBLINE_CONFIG::bestLRMatches(), matches_12);
}
// int match(const std::vector<MapLine*> &mvpLocalMapLines, Frame &CurrentFrame, float nnr, std::vector<int> &matches_12), src/LineMatcher.cpp:64-73
// (everything after the reference's early `return matchNNR(...)` is dead code)
template <class MapLineT, class FrameT> int match(const std::vector<MapLineT*>& vpLocalMapLines, FrameT& CurrentFrame, float nnr, std::vector<int>& matches_12)
{
    std::vector<uint8_t> d1((size_t)32 * vpLocalMapLines.size());
    for (size_t i = 0; i < vpLocalMapLines.size(); ++i) { const auto d = vpLocalMapLines[i]->GetDescriptor(); std::memcpy(&d1[32 * i], d.data, 32); }
    const olf_detail::DescRows<decltype(FrameT::mDescriptors_Line)> b(CurrentFrame.mDescriptors_Line);
    return matchNNR(olf_detail::thread_ctx(), d1.data(), (int)vpLocalMapLines.size(), b.p, b.n, nnr, matches_12);
}

// ---- include/gridStructure.h:33-58 + src/LineIterator.cpp: the bucket grid of the stereo line matcher, host side ------------------------
// (Frame::ComputeStereoMatches_Lines itself runs on the GPU through olf_stereo_lines / StereoFrameFeatures; these are for callers that
// use the grid matcher directly, with candidate sets that are far too small for a kernel launch.)
typedef std::pair<int, int> point_2d;
typedef std::pair<point_2d, point_2d> line_2d;
struct GridWindow { std::pair<int, int> width, height; };

class GridStructure {
public:
    int rows, cols;
    GridStructure(int rows_, int cols_) : rows(rows_), cols(cols_)
    {
        if (rows <= 0 || cols <= 0) throw std::runtime_error("[GridStructure] invalid dimension");
        grid.assign((size_t)cols * rows, std::list<int>());
    }
    std::list<int>& at(int x, int y) { return (x >= 0 && x < cols && y >= 0 && y < rows) ? grid[(size_t)x * rows + y] : out_of_bounds; }
    void get(int x, int y, const GridWindow& w, std::unordered_set<int>& indices) const
    {
        const int x0 = std::max(0, x - w.width.first), x1 = std::min(cols, x + w.width.second + 1);
        const int y0 = std::max(0, y - w.height.first), y1 = std::min(rows, y + w.height.second + 1);
        for (int cx = x0; cx < x1; ++cx)
            for (int cy = y0; cy < y1; ++cy) { const std::list<int>& c = grid[(size_t)cx * rows + cy]; indices.insert(c.begin(), c.end()); }
    }
    void clear() { for (size_t i = 0; i < grid.size(); ++i) grid[i].clear(); }
private:
    std::vector<std::list<int>> grid;
    std::list<int> out_of_bounds;
};

// void getLineCoords(double x1, double y1, double x2, double y2, std::list<std::pair<int, int>> &line_coords): the cells of the reference's
// double-precision Bresenham walk (src/LineIterator.cpp:34-77), through the library's implementation of it
inline void getLineCoords(double x1, double y1, double x2, double y2, std::list<std::pair<int, int>>& line_coords)
{
    line_coords.clear();
    int32_t n = 0;
    std::vector<int32_t> xy(2 * 4096);
    olf_detail::check(olf_line_coords(x1, y1, x2, y2, xy.data(), 4096, &n), "olf_line_coords");
    for (int i = 0; i < n; ++i) line_coords.push_back(std::make_pair((int)xy[2 * i], (int)xy[2 * i + 1]));
}

namespace olf_detail {
// the candidate loop both matchGrid overloads share (src/LineMatcher.cpp:152-299); keep(i2) is the per-candidate gate of the line overload
template <class MatT, class CandFn, class KeepFn>
int match_grid_core(size_t n1, const MatT& desc1, const MatT& desc2, double ratio, CandFn candidates_of, KeepFn keep, std::vector<int>& matches_12)
{
    const DescRows<MatT> d1(desc1), d2(desc2);
    const bool lr = ORBLINE_CONFIG::bestLRMatches();
    int matches = 0;
    matches_12.resize(d1.n, -1);
    std::vector<int> matches_21, distances;
    if (lr) { matches_21.resize(d2.n, -1); distances.resize(d2.n, std::numeric_limits<int>::max()); }
    for (int i1 = 0; i1 < (int)n1; ++i1) {
        int best_d = std::numeric_limits<int>::max(), best_d2 = std::numeric_limits<int>::max(), best_idx = -1;
        std::unordered_set<int> cand;
        candidates_of(i1, cand);
        for (std::unordered_set<int>::const_iterator it = cand.begin(); it != cand.end(); ++it) {
            const int i2 = *it;
            if (i2 < 0 || i2 >= d2.n || !keep(i1, i2)) continue;
            const int d = distance(d1.p + (size_t)32 * i1, d2.p + (size_t)32 * i2);
            if (lr) { if (d < distances[i2]) { distances[i2] = d; matches_21[i2] = i1; } else continue; }
            if (d < best_d) { best_d2 = best_d; best_d = d; best_idx = i2; }
            else if (d < best_d2) best_d2 = d;
        }
        if (best_d < best_d2 * ratio) { matches_12[i1] = best_idx; ++matches; }
    }
    if (lr)
        for (size_t i1 = 0; i1 < matches_12.size(); ++i1) { int& i2 = matches_12[i1]; if (i2 >= 0 && matches_21[i2] != (int)i1) { i2 = -1; --matches; } }
    return matches;
}
}  // namespace olf_detail

// int matchGrid(const std::vector<point_2d> &points1, const cv::Mat &desc1, const GridStructure &grid, const cv::Mat &desc2, const GridWindow &w, ...)
template <class MatT>
int matchGrid(const std::vector<point_2d>& points1, const MatT& desc1, const GridStructure& grid, const MatT& desc2, const GridWindow& w, std::vector<int>& matches_12)
{
    if ((int)points1.size() != desc1.rows) throw std::runtime_error("[matchGrid] Each point needs a corresponding descriptor!");
    return olf_detail::match_grid_core(points1.size(), desc1, desc2, ORBLINE_CONFIG::minRatio12P(),
                                       [&](int i1, std::unordered_set<int>& c) { grid.get(points1[i1].first, points1[i1].second, w, c); },
                                       [](int, int) { return true; }, matches_12);
}
// int matchGrid(const std::vector<line_2d> &lines1, const cv::Mat &desc1, const GridStructure &grid, const cv::Mat &desc2,
//               const std::vector<std::pair<double, double>> &directions2, const GridWindow &w, std::vector<int> &matches_12), src/LineMatcher.cpp:220-299
template <class MatT>
int matchGrid(const std::vector<line_2d>& lines1, const MatT& desc1, const GridStructure& grid, const MatT& desc2,
              const std::vector<std::pair<double, double>>& directions2, const GridWindow& w, std::vector<int>& matches_12, double min_ratio_12_l = 0.9)
{
    if ((int)lines1.size() != desc1.rows) throw std::runtime_error("[matchGrid] Each line needs a corresponding descriptor!");
    std::vector<std::pair<double, double>> v1(lines1.size());
    for (size_t i = 0; i < lines1.size(); ++i) {
        const point_2d sp = lines1[i].first, ep = lines1[i].second;
        std::pair<double, double> v = std::make_pair((double)(ep.first - sp.first), (double)(ep.second - sp.second));
        const double mag = std::sqrt(v.first * v.first + v.second * v.second);
        v.first /= mag; v.second /= mag;
        v1[i] = v;
    }
    const double sim = ORBLINE_CONFIG::lineSimTh();
    return olf_detail::match_grid_core(lines1.size(), desc1, desc2, min_ratio_12_l,
                                       [&](int i1, std::unordered_set<int>& c) {
                                           grid.get(lines1[i1].first.first, lines1[i1].first.second, w, c);
                                           grid.get(lines1[i1].second.first, lines1[i1].second.second, w, c);
                                       },
                                       [&](int i1, int i2) { return !(std::abs(v1[i1].first * directions2[i2].first + v1[i1].second * directions2[i2].second) < sim); },
                                       matches_12);
}

// ---- the feature part of Frame::Frame(imLeft, imRight, ...) (src/Frame.cc:136-221) as one call -------------------------------------------
// Replaces the four extraction threads (:164-171), ComputeStereoMatches (:202) and ComputeStereoMatches_Lines (:205): fills mvKeys,
// mvKeysRight, mDescriptors, mDescriptorsRight, mvuRight, mvDepth, mvKeys_Line, mvKeysRight_Line, mDescriptors_Line, mDescriptorsRight_Line,
// mvDisparity_l, mvle_l, N, N_l of `F`.  The parameters come from the frame's own extractor objects and camera members; the context is
// the left ORB extractor's (one per image size).  Throws like the reference when the two images differ in size (:145-146).
// Config::hasLines() == false (:37 of LineExtractor.cc, :203): no line is extracted or matched -- the key line members come back empty, N_l = 0, and the
// call runs the point half only.  A frame without key points returns like the reference's constructor does (:176-177): N = 0, N_l = the left key lines
// as extracted, and none of the stereo members (mvuRight, mvDepth, mvDisparity_l, mvle_l) is touched.
template <class FrameT, class MatT>
void StereoFrameFeatures(FrameT& F, const MatT& imLeft, const MatT& imRight, const olf_stereo_params* stereo = nullptr)
{
    if (imLeft.rows != imRight.rows || imLeft.cols != imRight.cols) throw std::runtime_error("[StereoFrame] Left and right images have different sizes");
    const int w = imLeft.cols, h = imLeft.rows;
    olf_params p = F.mpORBextractorLeft->params();
    p.line = F.mpLineextractorLeft->params().line;
    if (stereo) p.stereo = *stereo;
    p.stereo.fx = F.fx; p.stereo.bf = F.mbf;
    olf_ctx* ctx = F.mpORBextractorLeft->context(w, h, &p);
    const int cap = olf_orb_capacity(ctx), lcap = olf_line_capacity(ctx);
    const size_t npx = (size_t)w * h;
    std::vector<uint8_t> pair(2 * npx);
    for (int y = 0; y < h; ++y) {                                           // rows may be strided (a cv::Mat ROI)
        std::memcpy(&pair[(size_t)y * w], imLeft.template ptr<uint8_t>(y), w);
        std::memcpy(&pair[npx + (size_t)y * w], imRight.template ptr<uint8_t>(y), w);
    }
    std::vector<olf_keypoint> k((size_t)2 * cap);
    std::vector<uint8_t> d((size_t)2 * cap * 32), ld((size_t)2 * lcap * 32);
    std::vector<float> ur(cap), dep(cap), ldisp((size_t)2 * lcap);
    std::vector<olf_keyline> kl((size_t)2 * lcap);
    std::vector<int32_t> lm(lcap);
    std::vector<double> lle((size_t)3 * lcap);
    int32_t n[2] = {0, 0}, nl[2] = {0, 0};
    const bool lines = ORBLINE_CONFIG::hasLines();
    if (lines) {
        olf_frame_buffers o = {k.data(), d.data(), n, ur.data(), dep.data(), kl.data(), ld.data(), nl, lm.data(), ldisp.data(), lle.data()};
        olf_detail::check(olf_stereo_frames(ctx, pair.data(), 1, &o), "olf_stereo_frames");
    } else
        olf_detail::check(olf_stereo_points(ctx, pair.data(), 1, k.data(), d.data(), n, ur.data(), dep.data()), "olf_stereo_points");
    typedef typename std::remove_reference<decltype(F.mvKeys[0])>::type KP;
    typedef typename std::remove_reference<decltype(F.mvKeys_Line[0])>::type KL;
    static_assert(sizeof(KP) == sizeof(olf_keypoint) && sizeof(KL) == sizeof(olf_keyline), "cv::KeyPoint / KeyLine layout");
    auto fill_kp = [&](std::vector<KP>& dst, const olf_keypoint* src, int cnt) { dst.resize(cnt); if (cnt) std::memcpy((void*)dst.data(), src, (size_t)cnt * sizeof(KP)); };
    auto fill_kl = [&](std::vector<KL>& dst, const olf_keyline* src, int cnt) { dst.resize(cnt); if (cnt) std::memcpy((void*)dst.data(), src, (size_t)cnt * sizeof(KL)); };
    auto fill_mat = [&](MatT& m, const uint8_t* src, int cnt) { m.create(cnt, 32, 0 /* CV_8U */); if (cnt) std::memcpy(m.data, src, (size_t)cnt * 32); };
    fill_kp(F.mvKeys, k.data(), n[0]); fill_kp(F.mvKeysRight, k.data() + cap, n[1]);
    fill_mat(F.mDescriptors, d.data(), n[0]); fill_mat(F.mDescriptorsRight, d.data() + (size_t)cap * 32, n[1]);
    fill_kl(F.mvKeys_Line, kl.data(), nl[0]); fill_kl(F.mvKeysRight_Line, kl.data() + lcap, nl[1]);
    if (lines) { fill_mat(F.mDescriptors_Line, ld.data(), nl[0]); fill_mat(F.mDescriptorsRight_Line, ld.data() + (size_t)lcap * 32, nl[1]); }
    F.N = n[0]; F.N_l = nl[0];
    if (n[0] == 0) return;                                                  // "if(mvKeys.empty()) return;" (src/Frame.cc:176-177)
    F.mvuRight.assign(ur.begin(), ur.begin() + n[0]); F.mvDepth.assign(dep.begin(), dep.begin() + n[0]);
    if (!lines) return;                                                     // (src/Frame.cc:203)
    F.mvDisparity_l.resize(nl[0]); F.mvle_l.resize(nl[0]);
    for (int i = 0; i < nl[0]; ++i) {
        F.mvDisparity_l[i] = std::make_pair(ldisp[2 * i], ldisp[2 * i + 1]);
        for (int c = 0; c < 3; ++c) F.mvle_l[i][c] = lle[3 * i + c];
    }
}

}  // namespace ORB_SLAM2
