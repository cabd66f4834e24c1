// stereo_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// Restates (paths relative to /root/reference):
//   Frame::ComputeStereoMatches      src/Frame.cc:702-876
//   matchNNR / match(desc1,desc2)    src/LineMatcher.cpp:42-62, :104-132 (cv::BFMatcher knnMatch k=2, App. A.10)
// PARITY UNPINNED (see oracle_common.hpp).  Convention C.5: an empty vDistIdx (no stereo match at
// all) skips the median filter instead of the reference's out-of-range read (src/Frame.cc:863).
#include "oracle_common.hpp"
#include <climits>
#include <utility>

namespace orc {
struct OrbResult;  // orb_oracle.cpp
}

namespace orc {

static const int TH_HIGH = 100, TH_LOW = 50;   // src/ORBmatcher.cc:39-40

// pyrL/pyrR: level images (ROI part of mvImagePyramid) of the left / right extractor
void compute_stereo_matches(const std::vector<olf_keypoint>& keysL, const uint8_t* descL,
                            const std::vector<olf_keypoint>& keysR, const uint8_t* descR,
                            const std::vector<Image>& pyrL, const std::vector<Image>& pyrR,
                            const std::vector<float>& sf, const std::vector<float>& inv_sf, float mbf, float fx,
                            std::vector<float>& uRight, std::vector<float>& depth, std::vector<int>* sad_out)
{
    const int N = (int)keysL.size();
    uRight.assign(N, -1.0f);
    depth.assign(N, -1.0f);
    if (sad_out) sad_out->assign(N, -1);
    const float mb = mbf / fx;                       // src/Frame.cc:197
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = pyrL[0].h;
    std::vector<std::vector<size_t>> vRowIndices(nRows);
    const int Nr = (int)keysR.size();
    for (int iR = 0; iR < Nr; ++iR) {
        const olf_keypoint& kp = keysR[iR];
        const float kpY = kp.y;
        const float r = 2.0f * sf[kp.octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; ++yi)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);   // the reference does not bound-check; rows are in range for real key points
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < N; ++iL) {
        const olf_keypoint& kpL = keysL[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        const std::vector<size_t>& cand = vRowIndices[(size_t)vL];
        if (cand.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t* dL = descL + (size_t)iL * 32;
        for (size_t iC = 0; iC < cand.size(); ++iC) {
            const size_t iR = cand[iC];
            const olf_keypoint& kpR = keysR[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = hamming256(dL, descR + iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = keysR[bestIdxR].x;
            const float scaleFactor = inv_sf[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5;
            const Image& IL = pyrL[kpL.octave];
            const Image& IRm = pyrR[kpL.octave];
            const int cyL = (int)scaledvL, cxL = (int)scaleduL;
            int bestDistS = INT_MAX, bestincR = 0;
            const int L = 5;
            std::vector<float> vDists(2 * L + 1);
            const float iniu = scaleduR0 + L - w;
            const float endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= IRm.w) continue;
            for (int incR = -L; incR <= +L; ++incR) {
                const int cxR = (int)scaleduR0 + incR;
                // cv::norm(IL - centreL, IR - centreR, NORM_L1): integer-valued floats, exact in double (App. A.11)
                double acc = 0;
                const float cL = (float)IL.at(cxL, cyL), cR = (float)IRm.at(cxR, cyL);
                for (int dy = -w; dy <= w; ++dy)
                    for (int dx = -w; dx <= w; ++dx) {
                        float a = (float)IL.at(cxL + dx, cyL + dy) - cL;
                        float b = (float)IRm.at(cxR + dx, cyL + dy) - cR;
                        acc += std::fabs(a - b);
                    }
                float dist = (float)acc;
                if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = sf[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) {
                    disparity = 0.01;
                    bestuR = uL - 0.01;
                }
                depth[iL] = mbf / disparity;
                uRight[iL] = bestuR;
                vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
                if (sad_out) (*sad_out)[iL] = bestDistS;
            }
        }
    }
    if (vDistIdx.empty()) return;   // convention C.5
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; --i) {
        if (vDistIdx[i].first < thDist) break;
        uRight[vDistIdx[i].second] = -1;
        depth[vDistIdx[i].second] = -1;
    }
}

// cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, 2) + ratio test of matchNNR.  Fewer than 2 train rows:
// the reference indexes out of range (src/LineMatcher.cpp:55); convention App. A.10: no match.
int match_nnr(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, std::vector<int>& m12)
{
    int matches = 0;
    m12.assign(n1, -1);
    if (n2 < 2) return 0;
    for (int i = 0; i < n1; ++i) {
        int b0 = INT_MAX, b1 = INT_MAX, i0 = -1;
        for (int j = 0; j < n2; ++j) {
            int d = hamming256(d1 + (size_t)i * 32, d2 + (size_t)j * 32);
            if (d < b0) { b1 = b0; b0 = d; i0 = j; }
            else if (d < b1) b1 = d;
        }
        if ((float)b0 < (float)b1 * nnr) { m12[i] = i0; ++matches; }
    }
    return matches;
}

int match_lr(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, bool best_lr, std::vector<int>& m12)
{
    int matches = match_nnr(d1, n1, d2, n2, nnr, m12);
    if (best_lr) {
        std::vector<int> m21;
        match_nnr(d2, n2, d1, n1, nnr, m21);
        for (int i1 = 0; i1 < n1; ++i1) {
            int& i2 = m12[i1];
            if (i2 >= 0 && m21[i2] != i1) { i2 = -1; --matches; }
        }
    }
    return matches;
}

}  // namespace orc

using namespace orc;
extern "C" {

// kNN-2 brute force with ratio + (optional) mutual check: match(desc1, desc2, nnr, matches_12)
int orc_match_bf(const uint8_t* d1, int n1, const uint8_t* d2, int n2, float nnr, int best_lr, int* m12)
{
    std::vector<int> m;
    int r = match_lr(d1, n1, d2, n2, nnr, best_lr != 0, m);
    for (int i = 0; i < n1; ++i) m12[i] = m[i];
    return r;
}

// raw kNN-2 (knnMatch): idx0, d0, d1 per query (INT_MAX / -1 when unavailable)
int orc_knn2(const uint8_t* d1, int n1, const uint8_t* d2, int n2, int* idx0, int* dist0, int* dist1)
{
    for (int i = 0; i < n1; ++i) {
        int b0 = INT_MAX, b1 = INT_MAX, i0 = -1;
        for (int j = 0; j < n2; ++j) {
            int d = hamming256(d1 + (size_t)i * 32, d2 + (size_t)j * 32);
            if (d < b0) { b1 = b0; b0 = d; i0 = j; }
            else if (d < b1) b1 = d;
        }
        idx0[i] = i0; dist0[i] = b0; dist1[i] = b1;
    }
    return 0;
}

}  // extern "C"
