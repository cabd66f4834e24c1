// search_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// Restates (paths relative to /root/reference), on plain arrays instead of Frame / KeyFrame / MapPoint objects:
//   ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)   src/ORBmatcher.cc:1330-1472
//   ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches)     src/ORBmatcher.cc:161-290
//   ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vpMatches12)        src/ORBmatcher.cc:524-657
//   ORBmatcher::SearchByProjection(Frame&, KeyFrame*, sAlreadyFound, th, ORBdist)   src/ORBmatcher.cc:1620-1747 (+ MapPoint::PredictScale, src/MapPoint.cc:414-429)
//   ORBmatcher::SearchForTriangulation(KF1, KF2, F12, vMatchedPairs, bOnlyStereo)   src/ORBmatcher.cc:659-825 (+ CheckDistEpipolarLine :142-161)
//   ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>&, th), the search part             src/ORBmatcher.cc:827-975
//   ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, vpPoints, th, vpReplacePoint), the search part   src/ORBmatcher.cc:977-1102
//   ORBmatcher::SearchBySim3(KF1, KF2, vpMatches12, s12, R12, t12, th)               src/ORBmatcher.cc:1104-1328
//   ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)    src/ORBmatcher.cc:47-131 (+ RadiusByViewingCos :133-139)
//   ORBmatcher::ComputeThreeMaxima                                    src/ORBmatcher.cc:1749-1790
//   Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea        src/Frame.cc:334-349, :572-582, :517-570
// cv::Mat products of CV_32F operands (convention C.12): a plain product of inner length 3 (Rcw*x3Dw+tcw, Rlw*twc+tlw, sR21*p+t21, -sR21*t12)
// takes cv::gemm's small-matrix path -- the three products summed in float, alpha / the C term applied in double, one rounding; a product
// with a transposed operand (-Rcw.t()*tcw) takes the generic path -- double accumulation, one rounding.  Recalled from OpenCV 3.4
// matmul.cpp; OpenCV is not in the image, so this is a stated convention, not a verified fact.
// PARITY UNPINNED (see oracle_common.hpp).
#include "oracle_common.hpp"
#include <map>
#include <climits>

namespace orc {
static const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30, GC = 64, GR = 48;

struct Cam { float fx, fy, cx, cy, mbf, minX, maxX, minY, maxY; };

struct GridFrame {
    const olf_keypoint* keys; int N; Cam c;
    float wInv, hInv;
    std::vector<size_t> grid[GC][GR];
    void build()
    {
        wInv = static_cast<float>(GC) / (c.maxX - c.minX);
        hInv = static_cast<float>(GR) / (c.maxY - c.minY);
        for (int i = 0; i < N; ++i) {
            int posX = (int)std::round((keys[i].x - c.minX) * wInv), posY = (int)std::round((keys[i].y - c.minY) * hInv);
            if (posX < 0 || posX >= GC || posY < 0 || posY >= GR) continue;
            grid[posX][posY].push_back(i);
        }
    }
    std::vector<size_t> area(float x, float y, float r, int minLevel = -1, int maxLevel = -1) const
    {
        std::vector<size_t> v;
        const int nMinCellX = std::max(0, (int)std::floor((x - c.minX - r) * wInv));
        if (nMinCellX >= GC) return v;
        const int nMaxCellX = std::min(GC - 1, (int)std::ceil((x - c.minX + r) * wInv));
        if (nMaxCellX < 0) return v;
        const int nMinCellY = std::max(0, (int)std::floor((y - c.minY - r) * hInv));
        if (nMinCellY >= GR) return v;
        const int nMaxCellY = std::min(GR - 1, (int)std::ceil((y - c.minY + r) * hInv));
        if (nMaxCellY < 0) return v;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++)
                for (size_t j : grid[ix][iy]) {
                    const olf_keypoint& kp = keys[j];
                    if (bCheckLevels) {
                        if (kp.octave < minLevel) continue;
                        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                    }
                    const float distx = kp.x - x, disty = kp.y - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) v.push_back(j);
                }
        return v;
    }
};

static void three_maxima(std::vector<int>* histo, int L, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) ind3 = -1;
}

static void mat3_mul_add(const float* T /*4x4 row-major*/, const float v[3], float out[3])   // R*v + t
{
    for (int r = 0; r < 3; ++r) {
        const float t = T[4 * r] * v[0] + T[4 * r + 1] * v[1] + T[4 * r + 2] * v[2];      // float sum of the three products
        out[r] = (float)((double)t + (double)T[4 * r + 3]);
    }
}
}  // namespace orc

using namespace orc;
extern "C" {

// (match12: the reference's map<int, int> of the overload src/ORBmatcher.cc:1474-1618, nullptr for the four-argument overload :1330-1472)
static int search_by_projection_impl(const olf_keypoint* curKeys, const uint8_t* curDesc, const float* curURight, int curN, uint8_t* cur_mp_valid,
                             uint8_t* cur_mp_obs, const float* curTcw, const olf_keypoint* lastKeys, int lastN, const uint8_t* last_mp_valid,
                             const float* last_mp_world, const uint8_t* last_mp_desc, const uint8_t* last_mp_obs, const uint8_t* last_outlier,
                             const float* lastTcw, const float* cam9, const float* scaleFactors, float th, int bMono, int checkOri, int* matches,
                             std::map<int, int>* match12)
{
    if (match12) match12->clear();
    Cam c = {cam9[0], cam9[1], cam9[2], cam9[3], cam9[4], cam9[5], cam9[6], cam9[7], cam9[8]};
    const float mb = c.mbf / c.fx;
    GridFrame G; G.keys = curKeys; G.N = curN; G.c = c; G.build();
    for (int i = 0; i < curN; ++i) matches[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    // twc = -Rcw^T * tcw ; tlc = Rlw*twc + tlw
    float twc[3], tlc[3];
    for (int r = 0; r < 3; ++r) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += (double)curTcw[4 * k + r] * curTcw[4 * k + 3];
        twc[r] = (float)(-acc);
    }
    mat3_mul_add(lastTcw, twc, tlc);
    const bool bForward = tlc[2] > mb && !bMono, bBackward = -tlc[2] > mb && !bMono;
    for (int i = 0; i < lastN; i++) {
        if (!last_mp_valid[i] || last_outlier[i]) continue;
        float x3Dc[3];
        mat3_mul_add(curTcw, last_mp_world + 3 * i, x3Dc);
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = 1.0 / x3Dc[2];
        if (invzc < 0) continue;
        float u = c.fx * xc * invzc + c.cx, v = c.fy * yc * invzc + c.cy;
        if (u < c.minX || u > c.maxX) continue;
        if (v < c.minY || v > c.maxY) continue;
        const int nLastOctave = lastKeys[i].octave;
        const float radius = th * scaleFactors[nLastOctave];
        std::vector<size_t> vIndices2;
        if (bForward) vIndices2 = G.area(u, v, radius, nLastOctave);
        else if (bBackward) vIndices2 = G.area(u, v, radius, 0, nLastOctave);
        else vIndices2 = G.area(u, v, radius, nLastOctave - 1, nLastOctave + 1);
        if (vIndices2.empty()) continue;
        const uint8_t* dMP = last_mp_desc + 32 * (size_t)i;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            if (cur_mp_valid[i2] && cur_mp_obs[i2]) continue;
            if (curURight[i2] > 0) {
                const float ur = u - c.mbf * invzc;
                const float er = std::fabs(ur - curURight[i2]);
                if (er > radius) continue;
            }
            const int dist = hamming256(dMP, curDesc + 32 * i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
        }
        if (bestDist <= TH_HIGH) {
            cur_mp_valid[bestIdx2] = 1; cur_mp_obs[bestIdx2] = last_mp_obs[i];
            matches[bestIdx2] = i;
            nmatches++;
            if (match12) match12->insert(std::pair<int, int>(bestIdx2, i));      // :1577 -- insert() keeps an existing key
            if (checkOri) {
                float rot = lastKeys[i].angle - curKeys[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j : rotHist[i]) { cur_mp_valid[j] = 0; matches[j] = -1; nmatches--; if (match12) match12->erase(j); }
    }
    return nmatches;
}

int orc_search_by_projection(const olf_keypoint* curKeys, const uint8_t* curDesc, const float* curURight, int curN, uint8_t* cur_mp_valid,
                             uint8_t* cur_mp_obs, const float* curTcw, const olf_keypoint* lastKeys, int lastN, const uint8_t* last_mp_valid,
                             const float* last_mp_world, const uint8_t* last_mp_desc, const uint8_t* last_mp_obs, const uint8_t* last_outlier,
                             const float* lastTcw, const float* cam9, const float* scaleFactors, float th, int bMono, int checkOri, int* matches)
{
    return search_by_projection_impl(curKeys, curDesc, curURight, curN, cur_mp_valid, cur_mp_obs, curTcw, lastKeys, lastN, last_mp_valid, last_mp_world,
                                     last_mp_desc, last_mp_obs, last_outlier, lastTcw, cam9, scaleFactors, th, bMono, checkOri, matches, nullptr);
}

// ... with the map<int, int>& match12 of src/ORBmatcher.cc:1474-1618: pairs (key, value) in the map's iteration order, *n_pairs of them
int orc_search_by_projection_match12(const olf_keypoint* curKeys, const uint8_t* curDesc, const float* curURight, int curN, uint8_t* cur_mp_valid,
                             uint8_t* cur_mp_obs, const float* curTcw, const olf_keypoint* lastKeys, int lastN, const uint8_t* last_mp_valid,
                             const float* last_mp_world, const uint8_t* last_mp_desc, const uint8_t* last_mp_obs, const uint8_t* last_outlier,
                             const float* lastTcw, const float* cam9, const float* scaleFactors, float th, int bMono, int checkOri, int* matches,
                             int* pairs, int* n_pairs)
{
    std::map<int, int> m12;
    const int n = search_by_projection_impl(curKeys, curDesc, curURight, curN, cur_mp_valid, cur_mp_obs, curTcw, lastKeys, lastN, last_mp_valid,
                                            last_mp_world, last_mp_desc, last_mp_obs, last_outlier, lastTcw, cam9, scaleFactors, th, bMono, checkOri,
                                            matches, &m12);
    int k = 0;
    for (const auto& kv : m12) { pairs[2 * k] = kv.first; pairs[2 * k + 1] = kv.second; ++k; }
    *n_pairs = k;
    return n;
}

// ORBmatcher::SearchForInitialization, src/ORBmatcher.cc:407-522; prevMatched: (x, y) per F1 key point, updated in place
int orc_search_for_initialization(const olf_keypoint* keys1, const uint8_t* desc1, int n1, const olf_keypoint* keys2, const uint8_t* desc2, int n2,
                                  const float* cam9, float* prevMatched, int windowSize, float nnratio, int checkOri, int* vnMatches12)
{
    Cam c = {cam9[0], cam9[1], cam9[2], cam9[3], cam9[4], cam9[5], cam9[6], cam9[7], cam9[8]};
    GridFrame G; G.keys = keys2; G.N = n2; G.c = c; G.build();
    int nmatches = 0;
    for (int i = 0; i < n1; ++i) vnMatches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(n2, INT_MAX), vnMatches21(n2, -1);
    for (int i1 = 0; i1 < n1; i1++) {
        const olf_keypoint kp1 = keys1[i1];
        const int level1 = kp1.octave;
        if (level1 > 0) continue;
        std::vector<size_t> vIndices2 = G.area(prevMatched[2 * i1], prevMatched[2 * i1 + 1], windowSize, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t* d1 = desc1 + 32 * (size_t)i1;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            const int dist = hamming256(d1, desc2 + 32 * i2);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (checkOri) {
                    float rot = keys1[i1].angle - keys2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int idx1 : rotHist[i]) if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (vnMatches12[i1] >= 0) { prevMatched[2 * i1] = keys2[vnMatches12[i1]].x; prevMatched[2 * i1 + 1] = keys2[vnMatches12[i1]].y; }
    return nmatches;
}

// feature vectors as CSR: node ids ascending (std::map order), offsets, indices
int orc_search_by_bow(const olf_keypoint* kfKeys, const uint8_t* kfDesc, const uint8_t* kf_mp_valid, const uint8_t* kf_mp_bad, const int* kfNodes,
                      const int* kfOffs, const int* kfIdx, int kfNNodes, const olf_keypoint* fKeys, const uint8_t* fDesc, int fN, const int* fNodes,
                      const int* fOffs, const int* fIdx, int fNNodes, float nnratio, int checkOri, int* matched)
{
    for (int i = 0; i < fN; ++i) matched[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int a = 0, b = 0;
    while (a < kfNNodes && b < fNNodes) {
        if (kfNodes[a] == fNodes[b]) {
            for (int p = kfOffs[a]; p < kfOffs[a + 1]; ++p) {
                const int realIdxKF = kfIdx[p];
                if (!kf_mp_valid[realIdxKF] || kf_mp_bad[realIdxKF]) continue;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int q = fOffs[b]; q < fOffs[b + 1]; ++q) {
                    const int realIdxF = fIdx[q];
                    if (matched[realIdxF] >= 0) continue;
                    const int dist = hamming256(kfDesc + 32 * (size_t)realIdxKF, fDesc + 32 * (size_t)realIdxF);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW && static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                    matched[bestIdxF] = realIdxKF;
                    if (checkOri) {
                        float rot = kfKeys[realIdxKF].angle - fKeys[bestIdxF].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(bestIdxF);
                    }
                    nmatches++;
                }
            }
            ++a; ++b;
        } else if (kfNodes[a] < fNodes[b]) { while (a < kfNNodes && kfNodes[a] < fNodes[b]) ++a; }   // lower_bound
        else { while (b < fNNodes && fNodes[b] < kfNodes[a]) ++b; }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { matched[j] = -1; nmatches--; }
        }
    }
    return nmatches;
}


// Local-map tracking search (Tracking::SearchLocalPoints, src/Tracking.cc:1941): map points as SoA
//   in_view = mbTrackInView, bad = isBad(), level = mnTrackScaleLevel, view_cos = mTrackViewCos, proj = (mTrackProjX, mTrackProjY, mTrackProjXR),
//   desc = GetDescriptor(), obs = Observations() > 0.   matches[idx] = index of the map point assigned to feature idx.
int orc_search_local_map(const olf_keypoint* curKeys, const uint8_t* curDesc, const float* curURight, int curN, uint8_t* cur_mp_valid,
                         uint8_t* cur_mp_obs, const float* cam9, const float* scaleFactors, int nMP, const uint8_t* in_view, const uint8_t* bad,
                         const int* level, const float* view_cos, const float* proj3, const uint8_t* mp_desc, const uint8_t* mp_obs, float th,
                         float nnratio, int* matches)
{
    Cam c = {cam9[0], cam9[1], cam9[2], cam9[3], cam9[4], cam9[5], cam9[6], cam9[7], cam9[8]};
    GridFrame G; G.keys = curKeys; G.N = curN; G.c = c; G.build();
    for (int i = 0; i < curN; ++i) matches[i] = -1;
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (int iMP = 0; iMP < nMP; iMP++) {
        if (!in_view[iMP]) continue;
        if (bad[iMP]) continue;
        const int nPredictedLevel = level[iMP];
        float r = view_cos[iMP] > 0.998 ? 2.5f : 4.0f;      // RadiusByViewingCos: float compared with the double literal
        if (bFactor) r *= th;
        const std::vector<size_t> vIndices = G.area(proj3[3 * iMP], proj3[3 * iMP + 1], r * scaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t idx : vIndices) {
            if (cur_mp_valid[idx] && cur_mp_obs[idx]) continue;
            if (curURight[idx] > 0) {
                const float er = std::fabs(proj3[3 * iMP + 2] - curURight[idx]);
                if (er > r * scaleFactors[nPredictedLevel]) continue;
            }
            const int dist = hamming256(mp_desc + 32 * (size_t)iMP, curDesc + 32 * idx);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = curKeys[idx].octave; bestIdx = (int)idx; }
            else if (dist < bestDist2) { bestLevel2 = curKeys[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            cur_mp_valid[bestIdx] = 1; cur_mp_obs[bestIdx] = mp_obs[iMP];
            matches[bestIdx] = iMP;
            nmatches++;
        }
    }
    return nmatches;
}
}  // extern "C"

// SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, sAlreadyFound, th, ORBdist) (relocalisation, src/Tracking.cc:1786,1801):
// kf_* = the key frame's map point matches as SoA (valid, bad, already found, world position, descriptor, mfMaxDistance, mfMinDistance);
// matches[i2] = key frame feature index whose map point was assigned to current feature i2.
extern "C" int orc_search_by_projection_kf(const olf_keypoint* curKeys, const uint8_t* curDesc, int curN, uint8_t* cur_mp_valid, const float* curTcw,
                                           const float* cam9, const float* scaleFactors, int nLevels, float logScaleFactor,
                                           const olf_keypoint* kfKeys, int kfN, const uint8_t* kf_valid, const uint8_t* kf_bad,
                                           const uint8_t* kf_found, const float* kf_world, const uint8_t* kf_desc, const float* kf_maxd,
                                           const float* kf_mind, float th, int ORBdist, int checkOri, int* matches)
{
    using namespace orc;
    Cam c = {cam9[0], cam9[1], cam9[2], cam9[3], cam9[4], cam9[5], cam9[6], cam9[7], cam9[8]};
    GridFrame G; G.keys = curKeys; G.N = curN; G.c = c; G.build();
    for (int i = 0; i < curN; ++i) matches[i] = -1;
    int nmatches = 0;
    float Ow[3];
    for (int r = 0; r < 3; ++r) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += (double)curTcw[4 * k + r] * curTcw[4 * k + 3];
        Ow[r] = (float)(-acc);
    }
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < kfN; i++) {
        if (!kf_valid[i]) continue;
        if (kf_bad[i] || kf_found[i]) continue;
        const float* x3Dw = kf_world + 3 * i;
        float x3Dc[3];
        mat3_mul_add(curTcw, x3Dw, x3Dc);
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = 1.0 / x3Dc[2];
        const float u = c.fx * xc * invzc + c.cx, v = c.fy * yc * invzc + c.cy;
        if (u < c.minX || u > c.maxX) continue;
        if (v < c.minY || v > c.maxY) continue;
        double nrm = 0;
        for (int k = 0; k < 3; ++k) { const float po = x3Dw[k] - Ow[k]; nrm += (double)po * (double)po; }     // cv::norm(CV_32F, NORM_L2)
        const float dist3D = (float)std::sqrt(nrm);
        const float maxDistance = 1.2f * kf_maxd[i], minDistance = 0.8f * kf_mind[i];
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const float ratio = kf_maxd[i] / dist3D;
        int nPredictedLevel = (int)std::ceil(std::log(ratio) / logScaleFactor);
        if (nPredictedLevel < 0) nPredictedLevel = 0;
        else if (nPredictedLevel >= nLevels) nPredictedLevel = nLevels - 1;
        const float radius = th * scaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices2 = G.area(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1);
        if (vIndices2.empty()) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t i2 : vIndices2) {
            if (cur_mp_valid[i2]) continue;
            const int dist = hamming256(kf_desc + 32 * (size_t)i, curDesc + 32 * i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
        }
        if (bestDist <= ORBdist) {
            cur_mp_valid[bestIdx2] = 1;
            matches[bestIdx2] = i;
            nmatches++;
            if (checkOri) {
                float rot = kfKeys[i].angle - curKeys[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (int j : rotHist[i]) { cur_mp_valid[j] = 0; matches[j] = -1; nmatches--; }
    }
    return nmatches;
}

// SearchForTriangulation (LocalMapping::CreateNewMapPoints): features WITHOUT a map point, same vocabulary node, epipolar constraint.
// (vbMatched2 is never set in the reference, so several idx1 may take the same idx2 -- reproduced.)  matches12[idx1] = idx2 or -1.
extern "C" int orc_search_for_triangulation(const olf_keypoint* keys1, const uint8_t* desc1, int n1, const uint8_t* mp1_valid, const float* uRight1,
                                            const int* nodes1, const int* offs1, const int* idx1v, int nNodes1, const float* Cw3,
                                            const olf_keypoint* keys2, const uint8_t* desc2, int n2, const uint8_t* mp2_valid, const float* uRight2,
                                            const int* nodes2, const int* offs2, const int* idx2v, int nNodes2, const float* T2w, const float* cam4,
                                            const float* scaleFactors2, const float* levelSigma2, const float* F12, int bOnlyStereo, int checkOri,
                                            int* matches12)
{
    using namespace orc;
    float C2[3];
    mat3_mul_add(T2w, Cw3, C2);
    const float fx = cam4[0], fy = cam4[1], cx = cam4[2], cy = cam4[3];
    const float invz = 1.0f / C2[2];
    const float ex = fx * C2[0] * invz + cx, ey = fy * C2[1] * invz + cy;
    int nmatches = 0;
    std::vector<bool> vbMatched2(n2, false);
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    auto epi_ok = [&](const olf_keypoint& kp1, const olf_keypoint& kp2) -> bool {      // CheckDistEpipolarLine, F12 row-major 3x3
        const float a = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
        const float b = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
        const float c = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
        const float num = a * kp2.x + b * kp2.y + c;
        const float den = a * a + b * b;
        if (den == 0) return false;
        const float dsqr = num * num / den;
        return dsqr < 3.84 * levelSigma2[kp2.octave];
    };
    int a = 0, b = 0;
    while (a < nNodes1 && b < nNodes2) {
        if (nodes1[a] == nodes2[b]) {
            for (int p = offs1[a]; p < offs1[a + 1]; ++p) {
                const int idx1 = idx1v[p];
                if (mp1_valid[idx1]) continue;
                const bool bStereo1 = uRight1[idx1] >= 0;
                if (bOnlyStereo && !bStereo1) continue;
                const olf_keypoint& kp1 = keys1[idx1];
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int q = offs2[b]; q < offs2[b + 1]; ++q) {
                    const int idx2 = idx2v[q];
                    if (vbMatched2[idx2] || mp2_valid[idx2]) continue;
                    const bool bStereo2 = uRight2[idx2] >= 0;
                    if (bOnlyStereo && !bStereo2) continue;
                    const int dist = hamming256(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (dist > TH_LOW || dist > bestDist) continue;
                    const olf_keypoint& kp2 = keys2[idx2];
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - kp2.x, distey = ey - kp2.y;
                        if (distex * distex + distey * distey < 100 * scaleFactors2[kp2.octave]) continue;
                    }
                    if (epi_ok(kp1, kp2)) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    const olf_keypoint& kp2 = keys2[bestIdx2];
                    matches12[idx1] = bestIdx2;
                    nmatches++;
                    if (checkOri) {
                        float rot = kp1.angle - kp2.angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)std::round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(idx1);
                    }
                }
            }
            ++a; ++b;
        } else if (nodes1[a] < nodes2[b]) { while (a < nNodes1 && nodes1[a] < nodes2[b]) ++a; }
        else { while (b < nNodes2 && nodes2[b] < nodes1[a]) ++b; }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { matches12[j] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th), the search part: per map point the most similar key point inside the
// projection window (bestIdx, bestDist), or -1 when a geometric gate rejects the point / no candidate passes.  The map mutation that
// follows (Replace / AddObservation, :950-972) only depends on (bestIdx, bestDist <= TH_LOW) and on map state, and never feeds back
// into another point's search, so the reference's loop body splits exactly at that line.
extern "C" void orc_fuse_search(const olf_keypoint* keys, const uint8_t* desc, const float* uRight, int N, const float* Tcw, const float* Ow3,
                                const float* cam9, const float* scaleFactors, const float* invLevelSigma2, int nLevels, float logScaleFactor,
                                int nMP, const uint8_t* mp_skip, const float* mp_world, const float* mp_normal, const float* mp_maxd,
                                const float* mp_mind, const uint8_t* mp_desc, float th, int* bestIdxOut, int* bestDistOut)
{
    using namespace orc;
    Cam c = {cam9[0], cam9[1], cam9[2], cam9[3], cam9[4], cam9[5], cam9[6], cam9[7], cam9[8]};
    GridFrame G; G.keys = keys; G.N = N; G.c = c; G.build();
    for (int i = 0; i < nMP; i++) {
        bestIdxOut[i] = -1; bestDistOut[i] = 256;
        if (mp_skip[i]) continue;                    // !pMP, isBad(), IsInKeyFrame(pKF)
        const float* p3Dw = mp_world + 3 * i;
        float p3Dc[3];
        mat3_mul_add(Tcw, p3Dw, p3Dc);
        if (p3Dc[2] < 0.0f) continue;
        const float invz = 1 / p3Dc[2];
        const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
        const float u = c.fx * x + c.cx, v = c.fy * y + c.cy;
        if (!(u >= c.minX && u < c.maxX && v >= c.minY && v < c.maxY)) continue;
        const float ur = u - c.mbf * invz;
        const float maxDistance = 1.2f * mp_maxd[i], minDistance = 0.8f * mp_mind[i];
        float PO[3]; double nrm = 0, dot = 0;
        for (int k = 0; k < 3; ++k) { PO[k] = p3Dw[k] - Ow3[k]; nrm += (double)PO[k] * (double)PO[k]; dot += (double)PO[k] * (double)mp_normal[3 * i + k]; }
        const float dist3D = (float)std::sqrt(nrm);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        if (dot < 0.5 * dist3D) continue;
        const float ratio = mp_maxd[i] / dist3D;
        int nPredictedLevel = (int)std::ceil(std::log(ratio) / logScaleFactor);
        if (nPredictedLevel < 0) nPredictedLevel = 0;
        else if (nPredictedLevel >= nLevels) nPredictedLevel = nLevels - 1;
        const float radius = th * scaleFactors[nPredictedLevel];
        const std::vector<size_t> vIndices = G.area(u, v, radius);
        if (vIndices.empty()) continue;
        int bestDist = 256, bestIdx = -1;
        for (size_t idx : vIndices) {
            const olf_keypoint& kp = keys[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (uRight[idx] >= 0) {
                const float ex = u - kp.x, ey = v - kp.y, er = ur - uRight[idx];
                const float e2 = ex * ex + ey * ey + er * er;
                if (e2 * invLevelSigma2[kpLevel] > 7.8) continue;
            } else {
                const float ex = u - kp.x, ey = v - kp.y;
                const float e2 = ex * ex + ey * ey;
                if (e2 * invLevelSigma2[kpLevel] > 5.99) continue;
            }
            const int dist = hamming256(mp_desc + 32 * (size_t)i, desc + 32 * idx);
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        bestIdxOut[i] = bestIdx; bestDistOut[i] = bestDist;
    }
}

// Frame::isInFrustum, src/Frame.cc:388-444, over arrays of map points (GetMaxDistanceInvariance = 1.2f * mfMaxDistance,
// GetMinDistanceInvariance = 0.8f * mfMinDistance, src/MapPoint.cc:402-412; PredictScale :414-429)
extern "C" void orc_is_in_frustum(const float* Tcw, const float* cam9, const float* scaleFactors, int nLevels, float logScaleFactor, int nMP,
                                  const float* world, const float* normal, const float* maxd, const float* mind, float viewingCosLimit,
                                  uint8_t* inView, int* level, float* viewCosOut, float* proj3)
{
    using namespace orc;
    const Cam c = {cam9[0], cam9[1], cam9[2], cam9[3], cam9[4], cam9[5], cam9[6], cam9[7], cam9[8]};
    float mOw[3];
    for (int r = 0; r < 3; ++r) {                                  // mOw = -mRcw.t() * mtcw
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += (double)Tcw[4 * k + r] * (double)Tcw[4 * k + 3];
        mOw[r] = (float)(-acc);
    }
    for (int i = 0; i < nMP; ++i) {
        inView[i] = 0;
        const float* P = world + 3 * i;
        float Pc[3];
        mat3_mul_add(Tcw, P, Pc);
        if (Pc[2] < 0.0f) continue;
        const float invz = 1.0f / Pc[2];
        const float u = c.fx * Pc[0] * invz + c.cx;
        const float v = c.fy * Pc[1] * invz + c.cy;
        if (u < c.minX || u > c.maxX) continue;
        if (v < c.minY || v > c.maxY) continue;
        const float maxDistance = 1.2f * maxd[i], minDistance = 0.8f * mind[i];
        double nrm = 0, dot = 0;
        for (int k = 0; k < 3; ++k) { const float po = P[k] - mOw[k]; nrm += (double)po * po; dot += (double)po * (double)normal[3 * i + k]; }
        const float dist = (float)std::sqrt(nrm);
        if (dist < minDistance || dist > maxDistance) continue;
        const float viewCos = (float)(dot / dist);
        if (viewCos < viewingCosLimit) continue;
        const float ratio = maxd[i] / dist;
        int nScale = (int)std::ceil(std::log(ratio) / logScaleFactor);
        if (nScale < 0) nScale = 0; else if (nScale >= nLevels) nScale = nLevels - 1;
        inView[i] = 1; level[i] = nScale; viewCosOut[i] = viewCos;
        proj3[3 * i] = u; proj3[3 * i + 1] = v; proj3[3 * i + 2] = u - c.mbf * invz;
    }
}

// ---- Sim3 searches of the loop closer -------------------------------------------------------------------------------------------
// cv::Mat scalar algebra used below (CV_32F): `M / s` and `s * M` are MatExpr scalings evaluated by convertTo, i.e. every element times
// the double factor rounded to float ((float)(1.0 / s), (float)s); Mat::dot accumulates float products in double; `-A*b` is a gemm with
// alpha = -1 (small-matrix path unless an operand is transposed, see the header).
static void r3_mul_add(const float* R9, const float* v, const float* t3, float out[3], double alpha = 1.0, bool transposed = false)
{
    for (int r = 0; r < 3; ++r) {
        double acc = 0;
        if (transposed) for (int k = 0; k < 3; ++k) acc += (double)R9[3 * r + k] * v[k];
        else { const float t = R9[3 * r] * v[0] + R9[3 * r + 1] * v[1] + R9[3 * r + 2] * v[2]; acc = (double)t; }
        out[r] = (float)(alpha * acc + (t3 ? (double)t3[r] : 0.0));
    }
}

// test hook for the OpenCV fixtures (tests/test_oracle_cpu.py, opencv34_gemm.npz): the two CV_32F product forms of convention C.12 on one 3x3 matrix --
// out1 = R * x + t (small-matrix path), out2 = -R.t() * t (transposed operand, generic path)
extern "C" void orc_gemm3_check(const float* R9, const float* x3, const float* t3, float* out1, float* out2)
{
    r3_mul_add(R9, x3, t3, out1);
    float Rt[9];
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Rt[3 * r + k] = R9[3 * k + r];
    r3_mul_add(Rt, t3, nullptr, out2, -1.0, true);
}

// src/ORBmatcher.cc:985-989: Scw -> Rcw, tcw, Ow
extern "C" void orc_sim3_decompose(const float* Scw /*4x4 row-major*/, float* Rcw9, float* tcw3, float* Ow3)
{
    double d = 0;
    for (int k = 0; k < 3; ++k) d += (double)Scw[k] * (double)Scw[k];
    const float scw = (float)std::sqrt(d);
    const float inv = (float)(1.0 / (double)scw);
    for (int r = 0; r < 3; ++r) {
        for (int k = 0; k < 3; ++k) Rcw9[3 * r + k] = Scw[4 * r + k] * inv;
        tcw3[r] = Scw[4 * r + 3] * inv;
    }
    float Rt[9];
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) Rt[3 * r + k] = Rcw9[3 * k + r];
    r3_mul_add(Rt, tcw3, nullptr, Ow3, -1.0, true);      // Ow = -Rcw.t() * tcw: transposed operand, generic path
}

static int predict_scale(float maxd, float dist, float logScaleFactor, int nLevels)     // MapPoint::PredictScale, src/MapPoint.cc:414-429
{
    const float ratio = maxd / dist;
    int n = (int)std::ceil(std::log(ratio) / logScaleFactor);
    if (n < 0) n = 0; else if (n >= nLevels) n = nLevels - 1;
    return n;
}

// best key point of a key frame for a map point projected at (u, v): octave in [level-1, level], smallest Hamming distance, first wins
static void best_in_area(const GridFrame& G, const uint8_t* desc, float u, float v, float radius, int level, const uint8_t* dMP, int& bestIdx, int& bestDist)
{
    bestIdx = -1; bestDist = INT_MAX;
    for (size_t idx : G.area(u, v, radius)) {
        const int kpLevel = G.keys[idx].octave;
        if (kpLevel < level - 1 || kpLevel > level) continue;
        const int dist = hamming256(dMP, desc + 32 * idx);
        if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
    }
}

// The search part of Fuse(pKF, Scw, vpPoints, th, vpReplacePoint): per point (bestIdx, bestDist) or (-1, INT_MAX) when a gate rejects it.
// mp_skip = isBad() || spAlreadyFound.count(pMP).  What follows in the reference (:1086-1099) only reads (bestIdx, bestDist <= TH_LOW).
extern "C" void orc_fuse_search_sim3(const olf_keypoint* keys, const uint8_t* desc, int N, const float* Scw, const float* cam9, const float* scaleFactors,
                          int nLevels, float logScaleFactor, int nMP, const uint8_t* mp_skip, const float* mp_world, const float* mp_normal,
                          const float* mp_maxd, const float* mp_mind, const uint8_t* mp_desc, float th, int* bestIdxOut, int* bestDistOut)
{
    Cam c = {cam9[0], cam9[1], cam9[2], cam9[3], cam9[4], cam9[5], cam9[6], cam9[7], cam9[8]};
    GridFrame G; G.keys = keys; G.N = N; G.c = c; G.build();
    float Rcw[9], tcw[3], Ow[3];
    orc_sim3_decompose(Scw, Rcw, tcw, Ow);
    for (int i = 0; i < nMP; i++) {
        bestIdxOut[i] = -1; bestDistOut[i] = INT_MAX;
        if (mp_skip[i]) continue;
        const float* p3Dw = mp_world + 3 * i;
        float p3Dc[3];
        r3_mul_add(Rcw, p3Dw, tcw, p3Dc);
        if (p3Dc[2] < 0.0f) continue;
        const float invz = (float)(1.0 / p3Dc[2]);
        const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
        const float u = c.fx * x + c.cx, v = c.fy * y + c.cy;
        if (!(u >= c.minX && u < c.maxX && v >= c.minY && v < c.maxY)) continue;
        const float maxDistance = 1.2f * mp_maxd[i], minDistance = 0.8f * mp_mind[i];
        float PO[3]; double nrm = 0, dot = 0;
        for (int k = 0; k < 3; ++k) { PO[k] = p3Dw[k] - Ow[k]; nrm += (double)PO[k] * (double)PO[k]; dot += (double)PO[k] * (double)mp_normal[3 * i + k]; }
        const float dist3D = (float)std::sqrt(nrm);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        if (dot < 0.5 * dist3D) continue;
        const int nPredictedLevel = predict_scale(mp_maxd[i], dist3D, logScaleFactor, nLevels);
        const float radius = th * scaleFactors[nPredictedLevel];
        best_in_area(G, desc, u, v, radius, nPredictedLevel, mp_desc + 32 * (size_t)i, bestIdxOut[i], bestDistOut[i]);
    }
}

// int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th),
// src/ORBmatcher.cc:292-405 (LoopClosing.cc:381).  mp_skip = isBad() || spAlreadyFound.count(pMP); matched[idx] = vpMatched[idx] != NULL (in / out);
// kfMatch[idx] = index into vpPoints assigned to key point idx by this call.  Returns nmatches.
extern "C" int orc_search_by_projection_sim3(const olf_keypoint* keys, const uint8_t* desc, int N, const float* Scw, const float* cam9, const float* scaleFactors,
                          int nLevels, float logScaleFactor, int nMP, const uint8_t* mp_skip, const float* mp_world, const float* mp_normal,
                          const float* mp_maxd, const float* mp_mind, const uint8_t* mp_desc, int th, uint8_t* matched, int* kfMatch)
{
    Cam c = {cam9[0], cam9[1], cam9[2], cam9[3], cam9[4], cam9[5], cam9[6], cam9[7], cam9[8]};
    GridFrame G; G.keys = keys; G.N = N; G.c = c; G.build();
    float Rcw[9], tcw[3], Ow[3];
    orc_sim3_decompose(Scw, Rcw, tcw, Ow);                 // :301-305
    for (int i = 0; i < N; ++i) kfMatch[i] = -1;
    int nmatches = 0;
    for (int iMP = 0; iMP < nMP; iMP++) {
        if (mp_skip[iMP]) continue;                        // :320-322
        const float* p3Dw = mp_world + 3 * iMP;
        float p3Dc[3];
        r3_mul_add(Rcw, p3Dw, tcw, p3Dc);                  // :328
        if (p3Dc[2] < 0.0) continue;                       // :331
        const float invz = 1 / p3Dc[2];                    // :335 (an int over a float: float division)
        const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
        const float u = c.fx * x + c.cx, v = c.fy * y + c.cy;
        if (!(u >= c.minX && u < c.maxX && v >= c.minY && v < c.maxY)) continue;      // KeyFrame::IsInImage
        const float maxDistance = 1.2f * mp_maxd[iMP], minDistance = 0.8f * mp_mind[iMP];      // Get{Max,Min}DistanceInvariance, src/MapPoint.cc:385-396
        float PO[3]; double nrm = 0, dot = 0;
        for (int k = 0; k < 3; ++k) { PO[k] = p3Dw[k] - Ow[k]; nrm += (double)PO[k] * (double)PO[k]; dot += (double)PO[k] * (double)mp_normal[3 * iMP + k]; }
        const float dist = (float)std::sqrt(nrm);          // cv::norm(PO)
        if (dist < minDistance || dist > maxDistance) continue;
        if (dot < 0.5 * dist) continue;                    // :358
        const int nPredictedLevel = predict_scale(mp_maxd[iMP], dist, logScaleFactor, nLevels);
        const float radius = th * scaleFactors[nPredictedLevel];
        const uint8_t* dMP = mp_desc + 32 * (size_t)iMP;
        int bestDist = 256, bestIdx = -1;
        for (size_t idx : G.area(u, v, radius)) {
            if (matched[idx]) continue;                    // :378
            const int kpLevel = keys[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int d = hamming256(dMP, desc + 32 * idx);
            if (d < bestDist) { bestDist = d; bestIdx = (int)idx; }
        }
        if (bestDist <= TH_LOW) { matched[bestIdx] = 1; kfMatch[bestIdx] = iMP; nmatches++; }
    }
    return nmatches;
}

// SearchBySim3.  Per key frame: keys / desc / n, Tcw (4x4), and for every feature its map point (valid, bad, world, maxd, mind, descriptor);
// already1 / already2 = vbAlreadyMatched1 / 2 (:1139-1151, derived by the caller from vpMatches12).  Outputs vnMatch1 / vnMatch2 (:1153-1154) and
// matches12[i1] = idx2 where the two directions agree (:1312-1325; the caller stores vpMapPoints2[idx2]).  Returns nFound.
extern "C" int orc_search_by_sim3(const olf_keypoint* keys1, const uint8_t* desc1, int n1, const float* T1w, const uint8_t* mp1_valid, const uint8_t* mp1_bad,
                       const float* mp1_world, const float* mp1_maxd, const float* mp1_mind, const uint8_t* mp1_desc, const uint8_t* already1,
                       const olf_keypoint* keys2, const uint8_t* desc2, int n2, const float* T2w, const uint8_t* mp2_valid, const uint8_t* mp2_bad,
                       const float* mp2_world, const float* mp2_maxd, const float* mp2_mind, const uint8_t* mp2_desc, const uint8_t* already2,
                       const float* cam9, const float* scaleFactors, int nLevels, float logScaleFactor, float s12, const float* R12, const float* t12,
                       float th, int* vnMatch1, int* vnMatch2, int* matches12)
{
    Cam c = {cam9[0], cam9[1], cam9[2], cam9[3], cam9[4], cam9[5], cam9[6], cam9[7], cam9[8]};
    GridFrame G1; G1.keys = keys1; G1.N = n1; G1.c = c; G1.build();
    GridFrame G2; G2.keys = keys2; G2.N = n2; G2.c = c; G2.build();
    float sR12[9], sR21[9], t21[3];
    const float inv = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) { sR12[3 * r + k] = s12 * R12[3 * r + k]; sR21[3 * r + k] = inv * R12[3 * k + r]; }
    r3_mul_add(sR21, t12, nullptr, t21, -1.0);
    for (int i = 0; i < n1; ++i) { vnMatch1[i] = -1; matches12[i] = -1; }
    for (int i = 0; i < n2; ++i) vnMatch2[i] = -1;
    for (int dir = 0; dir < 2; ++dir) {
        // dir 0: points of KF1 into KF2 (:1157-1232); dir 1: points of KF2 into KF1 (:1235-1309)
        const int n = dir ? n2 : n1;
        const uint8_t *valid = dir ? mp2_valid : mp1_valid, *bad = dir ? mp2_bad : mp1_bad, *already = dir ? already2 : already1;
        const float *world = dir ? mp2_world : mp1_world, *maxd = dir ? mp2_maxd : mp1_maxd, *mind = dir ? mp2_mind : mp1_mind;
        const uint8_t* dMPs = dir ? mp2_desc : mp1_desc;
        const float* Tsrc = dir ? T2w : T1w;
        const float *sR = dir ? sR12 : sR21, *t = dir ? t12 : t21;
        const GridFrame& G = dir ? G1 : G2;
        const uint8_t* descT = dir ? desc1 : desc2;
        int* out = dir ? vnMatch2 : vnMatch1;
        for (int i = 0; i < n; ++i) {
            if (!valid[i] || already[i]) continue;
            if (bad[i]) continue;
            float pa[3], pb[3];
            mat3_mul_add(Tsrc, world + 3 * i, pa);
            r3_mul_add(sR, pa, t, pb);
            if (pb[2] < 0.0) continue;
            const float invz = (float)(1.0 / pb[2]);
            const float x = pb[0] * invz, y = pb[1] * invz;
            const float u = c.fx * x + c.cx, v = c.fy * y + c.cy;
            if (!(u >= c.minX && u < c.maxX && v >= c.minY && v < c.maxY)) continue;
            const float maxDistance = 1.2f * maxd[i], minDistance = 0.8f * mind[i];
            double nrm = 0;
            for (int k = 0; k < 3; ++k) nrm += (double)pb[k] * (double)pb[k];
            const float dist3D = (float)std::sqrt(nrm);
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            const int nPredictedLevel = predict_scale(maxd[i], dist3D, logScaleFactor, nLevels);
            const float radius = th * scaleFactors[nPredictedLevel];
            int bestIdx, bestDist;
            best_in_area(G, descT, u, v, radius, nPredictedLevel, dMPs + 32 * (size_t)i, bestIdx, bestDist);
            if (bestDist <= TH_HIGH) out[i] = bestIdx;
        }
    }
    int nFound = 0;
    for (int i1 = 0; i1 < n1; ++i1) {
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0 && vnMatch2[idx2] == i1) { matches12[i1] = idx2; nFound++; }
    }
    return nFound;
}

// SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vpMatches12) (loop closing): matches12[idx1] = idx2 (the feature of KF2 whose map point is taken)
extern "C" int orc_search_by_bow_kf(const olf_keypoint* keys1, const uint8_t* desc1, int n1, const uint8_t* mp1_valid, const uint8_t* mp1_bad,
                                    const int* nodes1, const int* offs1, const int* idx1v, int nNodes1, const olf_keypoint* keys2,
                                    const uint8_t* desc2, int n2, const uint8_t* mp2_valid, const uint8_t* mp2_bad, const int* nodes2,
                                    const int* offs2, const int* idx2v, int nNodes2, float nnratio, int checkOri, int* matches12)
{
    using namespace orc;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    std::vector<bool> vbMatched2(n2, false);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0, a = 0, b = 0;
    while (a < nNodes1 && b < nNodes2) {
        if (nodes1[a] == nodes2[b]) {
            for (int p = offs1[a]; p < offs1[a + 1]; ++p) {
                const int idx1 = idx1v[p];
                if (!mp1_valid[idx1]) continue;
                if (mp1_bad[idx1]) continue;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int q = offs2[b]; q < offs2[b + 1]; ++q) {
                    const int idx2 = idx2v[q];
                    if (vbMatched2[idx2] || !mp2_valid[idx2]) continue;
                    if (mp2_bad[idx2]) continue;
                    const int dist = hamming256(desc1 + 32 * (size_t)idx1, desc2 + 32 * (size_t)idx2);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        matches12[idx1] = bestIdx2;
                        vbMatched2[bestIdx2] = true;
                        if (checkOri) {
                            float rot = keys1[idx1].angle - keys2[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)std::round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(idx1);
                        }
                        nmatches++;
                    }
                }
            }
            ++a; ++b;
        } else if (nodes1[a] < nodes2[b]) { while (a < nNodes1 && nodes1[a] < nodes2[b]) ++a; }      // lower_bound on the ordered map
        else { while (b < nNodes2 && nodes2[b] < nodes1[a]) ++b; }
    }
    if (checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        three_maxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (int j : rotHist[i]) { matches12[j] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:254-318) == MapLine::ComputeDistinctiveDescriptors (src/MapLine.cc:257-322)
// on a batch of landmarks given as CSR lists of observing descriptors.
extern "C" void orc_distinctive_descriptors(const uint8_t* desc, const int* offs, int n_points, int* best)
{
    for (int p = 0; p < n_points; ++p) {
        const size_t N = (size_t)(offs[p + 1] - offs[p]);
        if (N == 0) { best[p] = -1; continue; }
        const uint8_t* d = desc + 32 * (size_t)offs[p];
        std::vector<float> Distances(N * N);
        for (size_t i = 0; i < N; i++) {
            Distances[i * N + i] = 0;
            for (size_t j = i + 1; j < N; j++) {
                const int distij = orc::hamming256(d + 32 * i, d + 32 * j);
                Distances[i * N + j] = (float)distij; Distances[j * N + i] = (float)distij;
            }
        }
        int BestMedian = INT_MAX, BestIdx = 0;
        for (size_t i = 0; i < N; i++) {
            std::vector<int> vDists(Distances.begin() + i * N, Distances.begin() + (i + 1) * N);
            std::sort(vDists.begin(), vDists.end());
            const int median = vDists[(size_t)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = (int)i; }
        }
        best[p] = BestIdx;
    }
}
