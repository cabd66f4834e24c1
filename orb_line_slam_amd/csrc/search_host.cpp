// search_host.cpp -- host side of the ORBmatcher searches behind the C ABI: the three every-frame ones (olf_search_by_projection,
// olf_search_by_bow, olf_search_local_map, fed by olf_is_in_frustum) and the relocalisation / LocalMapping / LoopClosing ones
// (olf_search_by_projection_kf, olf_search_by_bow_kf, olf_search_for_triangulation, olf_fuse_search, olf_fuse_search_sim3,
// olf_search_by_sim3).  Split of the reference's loops, as SURVEY 8(b) / App. C.7 prescribe:
//   1. candidate generation on the host, exactly as the reference walks them (Frame::GetFeaturesInArea over the 64 x 48 grid,
//      src/Frame.cc:517-570; the merge of two DBoW2 feature vectors, src/ORBmatcher.cc:176-203),
//   2. every DescriptorDistance of the search in one k_match_candidates launch (match.hip),
//   3. the reference's greedy resolution, which depends on the map points assigned so far, on the host in the reference's order.
// Map mutations (MapPoint::Replace / AddObservation in the Fuse functions) stay with the caller: they never feed back into a search.
// Float expressions are written as the reference writes them (src/ORBmatcher.cc, float unless a double literal promotes them); cv::Mat
// products of CV_32F operands accumulate in double and round once; the library is built with -ffp-contract=off.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/orbline.h"
#include "olf_internal.hpp"

#define OLF_TRY(expr) do { const int _rc = (expr); if (_rc != OLF_OK) return _rc; } while (0)

namespace {
using namespace olf;
constexpr int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;          // src/ORBmatcher.cc:39-41
constexpr int GRID_COLS = 64, GRID_ROWS = 48;                          // FRAME_GRID_COLS / FRAME_GRID_ROWS, include/Frame.h:43-44

// Frame::mGrid (src/Frame.cc:334-349) as one index array: the features of cell (ix, iy) are cell[ix * GRID_ROWS + iy] .. [+1), in
// feature order -- the order push_back gives them in the reference.
struct Grid {
    const olf_frame_view& f;
    float wInv, hInv;
    std::vector<int> cell, item;
    explicit Grid(const olf_frame_view& fr) : f(fr)
    {
        wInv = static_cast<float>(GRID_COLS) / (f.maxX - f.minX);      // mfGridElementWidthInv, src/Frame.cc:186-187
        hInv = static_cast<float>(GRID_ROWS) / (f.maxY - f.minY);
        std::vector<int> where((size_t)std::max(f.n, 0));
        cell.assign(GRID_COLS * GRID_ROWS + 1, 0);
        for (int i = 0; i < f.n; ++i) {                                 // PosInGrid, src/Frame.cc:572-582
            const int posX = (int)std::round((f.keys[i].x - f.minX) * wInv), posY = (int)std::round((f.keys[i].y - f.minY) * hInv);
            where[i] = (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) ? -1 : posX * GRID_ROWS + posY;
            if (where[i] >= 0) ++cell[where[i] + 1];
        }
        for (int c = 0; c < GRID_COLS * GRID_ROWS; ++c) cell[c + 1] += cell[c];
        item.resize(cell.back());
        std::vector<int> fill(cell.begin(), cell.end() - 1);
        for (int i = 0; i < f.n; ++i) if (where[i] >= 0) item[fill[where[i]]++] = i;
    }
    // Frame::GetFeaturesInArea: appends the indices to `out`, returns how many
    int area(float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) const
    {
        const int nMinCellX = std::max(0, (int)std::floor((x - f.minX - r) * wInv));
        if (nMinCellX >= GRID_COLS) return 0;
        const int nMaxCellX = std::min(GRID_COLS - 1, (int)std::ceil((x - f.minX + r) * wInv));
        if (nMaxCellX < 0) return 0;
        const int nMinCellY = std::max(0, (int)std::floor((y - f.minY - r) * hInv));
        if (nMinCellY >= GRID_ROWS) return 0;
        const int nMaxCellY = std::min(GRID_ROWS - 1, (int)std::ceil((y - f.minY + r) * hInv));
        if (nMaxCellY < 0) return 0;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        const size_t before = out.size();
        for (int ix = nMinCellX; ix <= nMaxCellX; ++ix) {
            // the cells (ix, nMinCellY .. nMaxCellY) are adjacent in `cell`
            for (int p = cell[ix * GRID_ROWS + nMinCellY]; p < cell[ix * GRID_ROWS + nMaxCellY + 1]; ++p) {
                const int j = item[p];
                const olf_keypoint& kp = f.keys[j];
                if (bCheckLevels) {
                    if (kp.octave < minLevel) continue;
                    if (maxLevel >= 0 && kp.octave > maxLevel) continue;
                }
                const float distx = kp.x - x, disty = kp.y - y;
                if (std::fabs(distx) < r && std::fabs(disty) < r) out.push_back(j);
            }
        }
        return (int)(out.size() - before);
    }
};

// the queries of one search: descriptor rows gathered contiguously, CSR candidate lists, distances from the GPU
struct Batch {
    std::vector<uint8_t> descQ;
    std::vector<int> offs{0}, cand, owner;
    std::vector<uint16_t> dist;
    void add(int who, const uint8_t* d) { owner.push_back(who); descQ.insert(descQ.end(), d, d + 32); offs.push_back((int)cand.size()); }
    int run(olf_ctx* c, const uint8_t* descT, int nT)
    {
        dist.assign(cand.size(), 0);
        if (owner.empty() || cand.empty()) return OLF_OK;
        return olf_match_candidates(c, descQ.data(), (int)owner.size(), descT, nT, offs.data(), cand.data(), dist.data());
    }
};

// ORBmatcher::ComputeThreeMaxima, src/ORBmatcher.cc:1749-1790
void three_maxima(const std::vector<int>* histo, int& ind1, int& ind2, int& ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    ind1 = ind2 = ind3 = -1;
    for (int i = 0; i < HISTO_LENGTH; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) ind3 = -1;
}

int rot_bin(float angle1, float angle2)
{
    float rot = angle1 - angle2;
    if (rot < 0.0) rot += 360.0f;
    int bin = (int)std::round(rot * (1.0f / HISTO_LENGTH));
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

// cv::Mat products of CV_32F operands (convention C.12): a plain product A*b (+ c) of inner length 3 takes cv::gemm's
// small-matrix path (flags == 0, 2 <= len <= 4): the three products are summed in float, alpha and the C term are applied in double and the
// result is rounded once.  Products with a transposed operand (A.t()*b) take the generic path: double accumulation, one rounding.
inline float dot3_small(const float* a, const float* b)
{
    float t = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];        // ((a0*b0 + a1*b1) + a2*b2) in float, no contraction (-ffp-contract=off)
    return t;
}
void rot_apply(const float* T, const float* v, float alpha_t, float* out)     // R * v + alpha_t * t, T = 4x4 row-major
{
    for (int r = 0; r < 3; ++r) out[r] = (float)((double)dot3_small(T + 4 * r, v) + (double)alpha_t * (double)T[4 * r + 3]);
}

bool bad_view(const olf_frame_view* f, bool needs_pose)
{
    return !f || f->n < 0 || (f->n && (!f->keys || !f->desc)) || (needs_pose && !f->Tcw);
}
// views whose scale tables are indexed by a predicted pyramid level (MapPoint::PredictScale) or by a key point's octave
bool bad_levels(const olf_frame_view* f) { return f->n_levels < 1 || f->n_levels > OLF_MAX_LEVELS || !f->scale_factors; }

// Both SearchByProjection(Frame, Frame) overloads: src/ORBmatcher.cc:1330-1472 and, with match12, :1474-1618.  match12 models the reference's
// map<int, int>: match12.insert(pair(bestIdx2, i)) keeps the FIRST last-frame index a current feature was matched with (:1577; a feature whose
// new map point has no observations can be matched again, and mvpMapPoints[bestIdx2] = pMP then keeps the last one), match12.erase(idx) on a
// rotation-histogram rejection (:1612).
int search_by_projection_frames(olf_ctx* c, const olf_frame_view* cur, const olf_frame_view* last, float th, int bMono, int check_orientation,
                                int32_t* matches, int32_t* match12, int32_t* nmatches)
{
    if (!c || bad_view(cur, true) || bad_view(last, true) || !matches || !nmatches || !cur->scale_factors || !cur->mp_valid || !cur->mp_obs ||
        (cur->n && !cur->uright) || (last->n && (!last->mp_valid || !last->mp_world || !last->mp_desc || !last->mp_obs))) {
        set_error("olf_search_by_projection: bad argument"); return OLF_ERR_INVALID;
    }
    for (int i = 0; i < cur->n; ++i) matches[i] = -1;
    if (match12) for (int i = 0; i < cur->n; ++i) match12[i] = -1;
    *nmatches = 0;
    const float mb = cur->mbf / cur->fx;
    // twc = -Rcw.t() * tcw;  tlc = Rlw * twc + tlw                                   (:1341-1349)
    float twc[3], tlc[3];
    for (int r = 0; r < 3; ++r) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += (double)cur->Tcw[4 * k + r] * (double)cur->Tcw[4 * k + 3];
        twc[r] = (float)(-acc);
    }
    rot_apply(last->Tcw, twc, 1.0f, tlc);
    const bool bForward = tlc[2] > mb && !bMono, bBackward = -tlc[2] > mb && !bMono;

    const Grid grid(*cur);
    Batch q;
    struct Meta { float u, invzc, radius; };
    std::vector<Meta> meta;
    for (int i = 0; i < last->n; ++i) {
        if (!last->mp_valid[i]) continue;
        if (last->outlier && last->outlier[i]) continue;
        float x3Dc[3];
        rot_apply(cur->Tcw, last->mp_world + 3 * (size_t)i, 1.0f, x3Dc);
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        if (invzc < 0) continue;
        const float u = cur->fx * xc * invzc + cur->cx, v = cur->fy * yc * invzc + cur->cy;
        if (u < cur->minX || u > cur->maxX) continue;
        if (v < cur->minY || v > cur->maxY) continue;
        const int nLastOctave = last->keys[i].octave;
        if (nLastOctave < 0 || nLastOctave >= cur->n_levels) { set_error("olf_search_by_projection: octave outside mvScaleFactors"); return OLF_ERR_INVALID; }
        const float radius = th * cur->scale_factors[nLastOctave];
        int got;
        if (bForward) got = grid.area(u, v, radius, nLastOctave, -1, q.cand);
        else if (bBackward) got = grid.area(u, v, radius, 0, nLastOctave, q.cand);
        else got = grid.area(u, v, radius, nLastOctave - 1, nLastOctave + 1, q.cand);
        if (!got) continue;
        q.add(i, last->mp_desc + 32 * (size_t)i);
        meta.push_back({u, invzc, radius});
    }
    OLF_TRY(q.run(c, cur->desc, cur->n));

    std::vector<int> rotHist[HISTO_LENGTH];
    int n = 0;
    for (size_t k = 0; k < q.owner.size(); ++k) {
        const int i = q.owner[k];
        const Meta& m = meta[k];
        int bestDist = 256, bestIdx2 = -1;
        for (int p = q.offs[k]; p < q.offs[k + 1]; ++p) {
            const int i2 = q.cand[p];
            if (cur->mp_valid[i2] && cur->mp_obs[i2]) continue;
            if (cur->uright[i2] > 0) {
                const float ur = m.u - cur->mbf * m.invzc;
                const float er = std::fabs(ur - cur->uright[i2]);
                if (er > m.radius) continue;
            }
            const int dist = q.dist[p];
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= TH_HIGH) {
            cur->mp_valid[bestIdx2] = 1;
            cur->mp_obs[bestIdx2] = last->mp_obs[i];
            matches[bestIdx2] = i;
            if (match12 && match12[bestIdx2] < 0) match12[bestIdx2] = i;
            n++;
            if (check_orientation) rotHist[rot_bin(last->keys[i].angle, cur->keys[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        three_maxima(rotHist, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b == ind1 || b == ind2 || b == ind3) continue;
            for (int j : rotHist[b]) { cur->mp_valid[j] = 0; matches[j] = -1; if (match12) match12[j] = -1; n--; }
        }
    }
    *nmatches = n;
    return OLF_OK;
}
}  // namespace

extern "C" {

int olf_search_by_projection(olf_ctx* c, const olf_frame_view* cur, const olf_frame_view* last, float th, int bMono, int check_orientation,
                             int32_t* matches, int32_t* nmatches)
{
    return search_by_projection_frames(c, cur, last, th, bMono, check_orientation, matches, nullptr, nmatches);
}

int olf_search_by_projection_match12(olf_ctx* c, const olf_frame_view* cur, const olf_frame_view* last, float th, int bMono, int check_orientation,
                                     int32_t* matches, int32_t* match12, int32_t* nmatches)
{
    if (!match12) { set_error("olf_search_by_projection_match12: bad argument"); return OLF_ERR_INVALID; }
    return search_by_projection_frames(c, cur, last, th, bMono, check_orientation, matches, match12, nmatches);
}

int olf_search_for_initialization(olf_ctx* c, const olf_frame_view* f1, const olf_frame_view* f2, float* prev_matched, int window_size, float nnratio,
                                  int check_orientation, int32_t* matches12, int32_t* nmatches)
{
    if (!c || bad_view(f1, false) || bad_view(f2, false) || !matches12 || !nmatches || (f1->n && !prev_matched)) {
        set_error("olf_search_for_initialization: bad argument"); return OLF_ERR_INVALID;
    }
    for (int i = 0; i < f1->n; ++i) matches12[i] = -1;
    *nmatches = 0;
    // candidate lists: GetFeaturesInArea(vbPrevMatched[i1], windowSize, level1, level1) for the level-0 key points of F1   (:421-429)
    const Grid grid(*f2);
    Batch q;
    for (int i1 = 0; i1 < f1->n; ++i1) {
        const int level1 = f1->keys[i1].octave;
        if (level1 > 0) continue;
        if (!grid.area(prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window_size, level1, level1, q.cand)) continue;
        q.add(i1, f1->desc + 32 * (size_t)i1);
    }
    OLF_TRY(q.run(c, f2->desc, f2->n));
    // the resolution depends on the matches made so far (vMatchedDistance): replayed in F1 order                             (:431-486)
    std::vector<int> matchedDistance((size_t)f2->n, INT_MAX), matches21((size_t)f2->n, -1);
    std::vector<int> rotHist[HISTO_LENGTH];
    int n = 0;
    for (size_t k = 0; k < q.owner.size(); ++k) {
        const int i1 = q.owner[k];
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int p = q.offs[k]; p < q.offs[k + 1]; ++p) {
            const int i2 = q.cand[p], dist = q.dist[p];
            if (matchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW && (float)bestDist < (float)bestDist2 * nnratio) {
            if (matches21[bestIdx2] >= 0) { matches12[matches21[bestIdx2]] = -1; n--; }
            matches12[i1] = bestIdx2;
            matches21[bestIdx2] = i1;
            matchedDistance[bestIdx2] = bestDist;
            n++;
            if (check_orientation) rotHist[rot_bin(f1->keys[i1].angle, f2->keys[bestIdx2].angle)].push_back(i1);
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        three_maxima(rotHist, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b == ind1 || b == ind2 || b == ind3) continue;
            for (int idx1 : rotHist[b]) if (matches12[idx1] >= 0) { matches12[idx1] = -1; n--; }
        }
    }
    for (int i1 = 0; i1 < f1->n; ++i1)                                                       // "Update prev matched"          (:516-519)
        if (matches12[i1] >= 0) { prev_matched[2 * i1] = f2->keys[matches12[i1]].x; prev_matched[2 * i1 + 1] = f2->keys[matches12[i1]].y; }
    *nmatches = n;
    return OLF_OK;
}

int olf_search_by_bow(olf_ctx* c, const olf_frame_view* kf, const olf_frame_view* f, float nnratio, int check_orientation, int32_t* matched,
                      int32_t* nmatches)
{
    if (!c || bad_view(kf, false) || bad_view(f, false) || !matched || !nmatches || (kf->n && (!kf->mp_valid || !kf->mp_bad)) ||
        (kf->fv_n && (!kf->fv_nodes || !kf->fv_offsets || !kf->fv_features)) || (f->fv_n && (!f->fv_nodes || !f->fv_offsets || !f->fv_features))) {
        set_error("olf_search_by_bow: bad argument"); return OLF_ERR_INVALID;
    }
    for (int i = 0; i < f->n; ++i) matched[i] = -1;
    *nmatches = 0;
    // the two ordered maps are walked in step; only nodes present in both contribute (:176-203, :273-281)
    Batch q;
    int a = 0, b = 0;
    while (a < kf->fv_n && b < f->fv_n) {
        if (kf->fv_nodes[a] == f->fv_nodes[b]) {
            for (int p = kf->fv_offsets[a]; p < kf->fv_offsets[a + 1]; ++p) {
                const int realIdxKF = kf->fv_features[p];
                if (realIdxKF < 0 || realIdxKF >= kf->n) { set_error("olf_search_by_bow: feature index outside the key frame"); return OLF_ERR_INVALID; }
                if (!kf->mp_valid[realIdxKF]) continue;
                if (kf->mp_bad[realIdxKF]) continue;
                q.cand.insert(q.cand.end(), f->fv_features + f->fv_offsets[b], f->fv_features + f->fv_offsets[b + 1]);
                q.add(realIdxKF, kf->desc + 32 * (size_t)realIdxKF);
            }
            ++a; ++b;
        } else if (kf->fv_nodes[a] < f->fv_nodes[b]) ++a;
        else ++b;
    }
    OLF_TRY(q.run(c, f->desc, f->n));

    std::vector<int> rotHist[HISTO_LENGTH];
    int n = 0;
    for (size_t k = 0; k < q.owner.size(); ++k) {
        const int realIdxKF = q.owner[k];
        int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
        for (int p = q.offs[k]; p < q.offs[k + 1]; ++p) {
            const int realIdxF = q.cand[p];
            if (realIdxF < 0 || realIdxF >= f->n) { set_error("olf_search_by_bow: feature index outside the frame"); return OLF_ERR_INVALID; }
            if (matched[realIdxF] >= 0) continue;
            const int dist = q.dist[p];
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 <= TH_LOW) {
            if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                matched[bestIdxF] = realIdxKF;
                if (check_orientation) rotHist[rot_bin(kf->keys[realIdxKF].angle, f->keys[bestIdxF].angle)].push_back(bestIdxF);
                n++;
            }
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        three_maxima(rotHist, ind1, ind2, ind3);
        for (int bb = 0; bb < HISTO_LENGTH; bb++) {
            if (bb == ind1 || bb == ind2 || bb == ind3) continue;
            for (int j : rotHist[bb]) { matched[j] = -1; n--; }
        }
    }
    *nmatches = n;
    return OLF_OK;
}

int olf_search_local_map(olf_ctx* c, const olf_frame_view* f, int n_mp, const uint8_t* track_in_view, const uint8_t* bad,
                         const int32_t* track_scale_level, const float* track_view_cos, const float* track_proj3, const uint8_t* mp_desc,
                         const uint8_t* mp_obs, float th, float nnratio, int32_t* matches, int32_t* nmatches)
{
    if (!c || bad_view(f, false) || n_mp < 0 || !matches || !nmatches || !f->scale_factors || !f->mp_valid || !f->mp_obs || (f->n && !f->uright) ||
        (n_mp && (!track_in_view || !bad || !track_scale_level || !track_view_cos || !track_proj3 || !mp_desc || !mp_obs))) {
        set_error("olf_search_local_map: bad argument"); return OLF_ERR_INVALID;
    }
    for (int i = 0; i < f->n; ++i) matches[i] = -1;
    *nmatches = 0;
    const bool bFactor = th != 1.0;
    const Grid grid(*f);
    Batch q;
    std::vector<float> radii;
    for (int iMP = 0; iMP < n_mp; ++iMP) {
        if (!track_in_view[iMP]) continue;
        if (bad[iMP]) continue;
        const int nPredictedLevel = track_scale_level[iMP];
        if (nPredictedLevel < 0 || nPredictedLevel >= f->n_levels) { set_error("olf_search_local_map: scale level outside mvScaleFactors"); return OLF_ERR_INVALID; }
        float r = track_view_cos[iMP] > 0.998 ? 2.5f : 4.0f;            // RadiusByViewingCos, :133-139
        if (bFactor) r *= th;
        const float rs = r * f->scale_factors[nPredictedLevel];
        if (!grid.area(track_proj3[3 * (size_t)iMP], track_proj3[3 * (size_t)iMP + 1], rs, nPredictedLevel - 1, nPredictedLevel, q.cand)) continue;
        q.add(iMP, mp_desc + 32 * (size_t)iMP);
        radii.push_back(rs);
    }
    OLF_TRY(q.run(c, f->desc, f->n));

    int n = 0;
    for (size_t k = 0; k < q.owner.size(); ++k) {
        const int iMP = q.owner[k];
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int p = q.offs[k]; p < q.offs[k + 1]; ++p) {
            const int idx = q.cand[p];
            if (f->mp_valid[idx] && f->mp_obs[idx]) continue;
            if (f->uright[idx] > 0) {
                const float er = std::fabs(track_proj3[3 * (size_t)iMP + 2] - f->uright[idx]);
                if (er > radii[k]) continue;
            }
            const int dist = q.dist[p];
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = f->keys[idx].octave; bestIdx = idx; }
            else if (dist < bestDist2) { bestLevel2 = f->keys[idx].octave; bestDist2 = dist; }
        }
        // Apply ratio to second match (only if best and second are in the same scale level)
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            f->mp_valid[bestIdx] = 1;
            f->mp_obs[bestIdx] = mp_obs[iMP];
            matches[bestIdx] = iMP;
            n++;
        }
    }
    *nmatches = n;
    return OLF_OK;
}

}  // extern "C"

// ---- LocalMapping / LoopClosing / relocalisation searches ------------------------------------------------------------------------

namespace {
// merge of two DBoW2 feature vectors: calls f(a, b) for every node present in both (:537-543, :683-689)
template <class F>
void for_common_nodes(const olf_frame_view& A, const olf_frame_view& B, F&& f)
{
    int a = 0, b = 0;
    while (a < A.fv_n && b < B.fv_n) {
        if (A.fv_nodes[a] == B.fv_nodes[b]) { f(a, b); ++a; ++b; }
        else if (A.fv_nodes[a] < B.fv_nodes[b]) ++a;
        else ++b;
    }
}

bool bad_fv(const olf_frame_view* v) { return v->fv_n < 0 || (v->fv_n && (!v->fv_nodes || !v->fv_offsets || !v->fv_features)); }

void camera_centre(const float* Tcw, float* Ow)                 // -Rcw.t() * tcw
{
    for (int r = 0; r < 3; ++r) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += (double)Tcw[4 * k + r] * (double)Tcw[4 * k + 3];
        Ow[r] = (float)(-acc);
    }
}

// alpha * R * v (+ t), cv::gemm.  transposed: R9 holds the transpose of the matrix the reference multiplies with .t() -> generic path
void r3_apply(const float* R9, const float* v, const float* t3, float* out, double alpha = 1.0, bool transposed = false)
{
    for (int r = 0; r < 3; ++r) {
        double acc = 0;
        if (transposed) for (int k = 0; k < 3; ++k) acc += (double)R9[3 * r + k] * (double)v[k];
        else acc = (double)dot3_small(R9 + 3 * r, v);
        out[r] = (float)(alpha * acc + (t3 ? (double)t3[r] : 0.0));
    }
}

int predict_scale(float maxd, float dist, float logScaleFactor, int nLevels)       // MapPoint::PredictScale, src/MapPoint.cc:414-429
{
    const float ratio = maxd / dist;
    int n = (int)std::ceil(std::log(ratio) / logScaleFactor);
    if (n < 0) n = 0; else if (n >= nLevels) n = nLevels - 1;
    return n;
}

float log_scale_factor(const olf_frame_view& f) { return f.n_levels > 1 ? std::log(f.scale_factors[1]) : 1.0f; }    // mfLogScaleFactor

// pinhole projection + IsInImage (half-open, KeyFrame::IsInImage)
bool project_in_image(const olf_frame_view& K, const float* p3Dc, float& u, float& v, float& invz)
{
    if (p3Dc[2] < 0.0f) return false;
    invz = (float)(1.0 / p3Dc[2]);
    const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
    u = K.fx * x + K.cx; v = K.fy * y + K.cy;
    return u >= K.minX && u < K.maxX && v >= K.minY && v < K.maxY;
}
}  // namespace

extern "C" {

int olf_is_in_frustum(const olf_frame_view* f, int n_mp, const float* world, const float* normal, const float* maxd, const float* mind,
                      float viewing_cos_limit, uint8_t* track_in_view, int32_t* track_scale_level, float* track_view_cos, float* track_proj3)
{
    if (!f || !f->Tcw || !f->scale_factors || f->n_levels < 1 || n_mp < 0 ||
        (n_mp && (!world || !normal || !maxd || !mind || !track_in_view || !track_scale_level || !track_view_cos || !track_proj3))) {
        set_error("olf_is_in_frustum: bad argument"); return OLF_ERR_INVALID;
    }
    float Ow[3];
    camera_centre(f->Tcw, Ow);                                   // mOw, Frame::UpdatePoseMatrices (src/Frame.cc:380-386)
    const float logSF = log_scale_factor(*f);
    for (int i = 0; i < n_mp; ++i) {
        track_in_view[i] = 0;
        const float* P = world + 3 * (size_t)i;
        // 3D in camera coordinates
        float Pc[3];
        rot_apply(f->Tcw, P, 1.0f, Pc);
        const float PcX = Pc[0], PcY = Pc[1], PcZ = Pc[2];
        // Check positive depth
        if (PcZ < 0.0f) continue;
        // Project in image and check it is not outside
        const float invz = 1.0f / PcZ;
        const float u = f->fx * PcX * invz + f->cx, v = f->fy * PcY * invz + f->cy;
        if (u < f->minX || u > f->maxX) continue;
        if (v < f->minY || v > f->maxY) continue;
        // Check distance is in the scale invariance region of the MapPoint
        const float maxDistance = 1.2f * maxd[i], minDistance = 0.8f * mind[i];
        float PO[3]; double nrm = 0, dot = 0;
        for (int k = 0; k < 3; ++k) { PO[k] = P[k] - Ow[k]; nrm += (double)PO[k] * (double)PO[k]; dot += (double)PO[k] * (double)normal[3 * (size_t)i + k]; }
        const float dist = (float)std::sqrt(nrm);
        if (dist < minDistance || dist > maxDistance) continue;
        // Check viewing angle
        const float viewCos = (float)(dot / dist);
        if (viewCos < viewing_cos_limit) continue;
        // Predict scale in the image; data used by the tracking
        track_scale_level[i] = predict_scale(maxd[i], dist, logSF, f->n_levels);
        track_in_view[i] = 1;
        track_proj3[3 * (size_t)i] = u; track_proj3[3 * (size_t)i + 1] = v; track_proj3[3 * (size_t)i + 2] = u - f->mbf * invz;
        track_view_cos[i] = viewCos;
    }
    return OLF_OK;
}

int olf_search_by_projection_kf(olf_ctx* c, const olf_frame_view* cur, const olf_frame_view* kf, const uint8_t* already_found, float th,
                                int orb_dist, int check_orientation, int32_t* matches, int32_t* nmatches)
{
    if (!c || bad_view(cur, true) || bad_view(kf, false) || !matches || !nmatches || !cur->scale_factors || !cur->mp_valid ||
        (kf->n && (!kf->mp_valid || !kf->mp_bad || !kf->mp_world || !kf->mp_desc || !kf->mp_maxd || !kf->mp_mind))) {
        set_error("olf_search_by_projection_kf: bad argument"); return OLF_ERR_INVALID;
    }
    if (bad_levels(cur)) { set_error("olf_search_by_projection_kf: n_levels / scale_factors missing"); return OLF_ERR_INVALID; }
    for (int i = 0; i < cur->n; ++i) matches[i] = -1;
    *nmatches = 0;
    float Ow[3];
    camera_centre(cur->Tcw, Ow);
    const float logSF = log_scale_factor(*cur);
    const Grid grid(*cur);
    Batch q;
    for (int i = 0; i < kf->n; ++i) {
        if (!kf->mp_valid[i]) continue;
        if (kf->mp_bad[i] || (already_found && already_found[i])) continue;
        const float* x3Dw = kf->mp_world + 3 * (size_t)i;
        float x3Dc[3];
        rot_apply(cur->Tcw, x3Dw, 1.0f, x3Dc);
        const float xc = x3Dc[0], yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        const float u = cur->fx * xc * invzc + cur->cx, v = cur->fy * yc * invzc + cur->cy;
        if (u < cur->minX || u > cur->maxX) continue;
        if (v < cur->minY || v > cur->maxY) continue;
        // Compute predicted scale level
        double nrm = 0;
        for (int k = 0; k < 3; ++k) { const float po = x3Dw[k] - Ow[k]; nrm += (double)po * (double)po; }
        const float dist3D = (float)std::sqrt(nrm);
        const float maxDistance = 1.2f * kf->mp_maxd[i], minDistance = 0.8f * kf->mp_mind[i];
        // Depth must be inside the scale pyramid of the image
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = predict_scale(kf->mp_maxd[i], dist3D, logSF, cur->n_levels);
        // Search in a window
        const float radius = th * cur->scale_factors[nPredictedLevel];
        if (!grid.area(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1, q.cand)) continue;
        q.add(i, kf->mp_desc + 32 * (size_t)i);
    }
    OLF_TRY(q.run(c, cur->desc, cur->n));
    std::vector<int> rotHist[HISTO_LENGTH];
    int n = 0;
    for (size_t k = 0; k < q.owner.size(); ++k) {
        const int i = q.owner[k];
        int bestDist = 256, bestIdx2 = -1;
        for (int p = q.offs[k]; p < q.offs[k + 1]; ++p) {
            const int i2 = q.cand[p];
            if (cur->mp_valid[i2]) continue;
            const int dist = q.dist[p];
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestDist <= orb_dist) {
            cur->mp_valid[bestIdx2] = 1;
            matches[bestIdx2] = i;
            n++;
            if (check_orientation) rotHist[rot_bin(kf->keys[i].angle, cur->keys[bestIdx2].angle)].push_back(bestIdx2);
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        three_maxima(rotHist, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b == ind1 || b == ind2 || b == ind3) continue;
            for (int j : rotHist[b]) { cur->mp_valid[j] = 0; matches[j] = -1; n--; }
        }
    }
    *nmatches = n;
    return OLF_OK;
}

int olf_search_by_bow_kf(olf_ctx* c, const olf_frame_view* kf1, const olf_frame_view* kf2, float nnratio, int check_orientation,
                         int32_t* matches12, int32_t* nmatches)
{
    if (!c || bad_view(kf1, false) || bad_view(kf2, false) || !matches12 || !nmatches || bad_fv(kf1) || bad_fv(kf2) ||
        (kf1->n && (!kf1->mp_valid || !kf1->mp_bad)) || (kf2->n && (!kf2->mp_valid || !kf2->mp_bad))) {
        set_error("olf_search_by_bow_kf: bad argument"); return OLF_ERR_INVALID;
    }
    for (int i = 0; i < kf1->n; ++i) matches12[i] = -1;
    *nmatches = 0;
    Batch q;
    bool oob = false;
    for_common_nodes(*kf1, *kf2, [&](int a, int b) {
        for (int p = kf1->fv_offsets[a]; p < kf1->fv_offsets[a + 1]; ++p) {
            const int idx1 = kf1->fv_features[p];
            if (idx1 < 0 || idx1 >= kf1->n) { oob = true; continue; }
            if (!kf1->mp_valid[idx1]) continue;
            if (kf1->mp_bad[idx1]) continue;
            q.cand.insert(q.cand.end(), kf2->fv_features + kf2->fv_offsets[b], kf2->fv_features + kf2->fv_offsets[b + 1]);
            q.add(idx1, kf1->desc + 32 * (size_t)idx1);
        }
    });
    for (int idx2 : q.cand) if (idx2 < 0 || idx2 >= kf2->n) oob = true;
    if (oob) { set_error("olf_search_by_bow_kf: feature index outside its key frame"); return OLF_ERR_INVALID; }
    OLF_TRY(q.run(c, kf2->desc, kf2->n));
    std::vector<uint8_t> vbMatched2((size_t)std::max(kf2->n, 0), 0);
    std::vector<int> rotHist[HISTO_LENGTH];
    int n = 0;
    for (size_t k = 0; k < q.owner.size(); ++k) {
        const int idx1 = q.owner[k];
        int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
        for (int p = q.offs[k]; p < q.offs[k + 1]; ++p) {
            const int idx2 = q.cand[p];
            if (vbMatched2[idx2] || !kf2->mp_valid[idx2]) continue;
            if (kf2->mp_bad[idx2]) continue;
            const int dist = q.dist[p];
            if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 < TH_LOW) {
            if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                matches12[idx1] = bestIdx2;
                vbMatched2[bestIdx2] = 1;
                if (check_orientation) rotHist[rot_bin(kf1->keys[idx1].angle, kf2->keys[bestIdx2].angle)].push_back(idx1);
                n++;
            }
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        three_maxima(rotHist, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b == ind1 || b == ind2 || b == ind3) continue;
            for (int j : rotHist[b]) { matches12[j] = -1; n--; }
        }
    }
    *nmatches = n;
    return OLF_OK;
}

int olf_search_for_triangulation(olf_ctx* c, const olf_frame_view* kf1, const olf_frame_view* kf2, const float* F12, const float* Cw,
                                 int only_stereo, int check_orientation, int32_t* matches12, int32_t* nmatches)
{
    if (!c || bad_view(kf1, !Cw) || bad_view(kf2, true) || !F12 || !matches12 || !nmatches || bad_fv(kf1) || bad_fv(kf2) || !kf2->scale_factors ||
        (kf1->n && (!kf1->mp_valid || !kf1->uright)) || (kf2->n && (!kf2->mp_valid || !kf2->uright))) {
        set_error("olf_search_for_triangulation: bad argument"); return OLF_ERR_INVALID;
    }
    for (int i = 0; i < kf1->n; ++i) matches12[i] = -1;
    *nmatches = 0;
    // Compute epipole in second image (:666-676)
    float cw[3], C2[3];
    if (Cw) std::memcpy(cw, Cw, sizeof(cw)); else camera_centre(kf1->Tcw, cw);
    rot_apply(kf2->Tcw, cw, 1.0f, C2);
    const float invz = 1.0f / C2[2];
    const float ex = kf2->fx * C2[0] * invz + kf2->cx, ey = kf2->fy * C2[1] * invz + kf2->cy;
    Batch q;
    bool oob = false;
    for_common_nodes(*kf1, *kf2, [&](int a, int b) {
        for (int p = kf1->fv_offsets[a]; p < kf1->fv_offsets[a + 1]; ++p) {
            const int idx1 = kf1->fv_features[p];
            if (idx1 < 0 || idx1 >= kf1->n) { oob = true; continue; }
            // If there is already a MapPoint skip
            if (kf1->mp_valid[idx1]) continue;
            const bool bStereo1 = kf1->uright[idx1] >= 0;
            if (only_stereo) if (!bStereo1) continue;
            q.cand.insert(q.cand.end(), kf2->fv_features + kf2->fv_offsets[b], kf2->fv_features + kf2->fv_offsets[b + 1]);
            q.add(idx1, kf1->desc + 32 * (size_t)idx1);
        }
    });
    for (int idx2 : q.cand) if (idx2 < 0 || idx2 >= kf2->n) oob = true;
    if (oob) { set_error("olf_search_for_triangulation: feature index outside its key frame"); return OLF_ERR_INVALID; }
    OLF_TRY(q.run(c, kf2->desc, kf2->n));
    std::vector<int> rotHist[HISTO_LENGTH];
    int n = 0;
    for (size_t k = 0; k < q.owner.size(); ++k) {
        const int idx1 = q.owner[k];
        const bool bStereo1 = kf1->uright[idx1] >= 0;
        const olf_keypoint& kp1 = kf1->keys[idx1];
        int bestDist = TH_LOW, bestIdx2 = -1;
        for (int p = q.offs[k]; p < q.offs[k + 1]; ++p) {
            const int idx2 = q.cand[p];
            // If we have already matched or there is a MapPoint skip (vbMatched2 is never set in the reference)
            if (kf2->mp_valid[idx2]) continue;
            const bool bStereo2 = kf2->uright[idx2] >= 0;
            if (only_stereo) if (!bStereo2) continue;
            const int dist = q.dist[p];
            if (dist > TH_LOW || dist > bestDist) continue;
            const olf_keypoint& kp2 = kf2->keys[idx2];
            if (kp2.octave < 0 || kp2.octave >= kf2->n_levels) { set_error("olf_search_for_triangulation: octave outside mvScaleFactors"); return OLF_ERR_INVALID; }
            if (!bStereo1 && !bStereo2) {
                const float distex = ex - kp2.x, distey = ey - kp2.y;
                if (distex * distex + distey * distey < 100 * kf2->scale_factors[kp2.octave]) continue;
            }
            // CheckDistEpipolarLine (:142-161), mvLevelSigma2[l] = mvScaleFactor[l]^2
            const float ea = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
            const float eb = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
            const float ec = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
            const float num = ea * kp2.x + eb * kp2.y + ec;
            const float den = ea * ea + eb * eb;
            if (den == 0) continue;
            const float dsqr = num * num / den;
            const float sigma2 = kf2->scale_factors[kp2.octave] * kf2->scale_factors[kp2.octave];
            if (dsqr < 3.84 * sigma2) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
            matches12[idx1] = bestIdx2;
            n++;
            if (check_orientation) rotHist[rot_bin(kp1.angle, kf2->keys[bestIdx2].angle)].push_back(idx1);
        }
    }
    if (check_orientation) {
        int ind1, ind2, ind3;
        three_maxima(rotHist, ind1, ind2, ind3);
        for (int b = 0; b < HISTO_LENGTH; b++) {
            if (b == ind1 || b == ind2 || b == ind3) continue;
            for (int j : rotHist[b]) { matches12[j] = -1; n--; }
        }
    }
    *nmatches = n;
    return OLF_OK;
}

}  // extern "C"

namespace {
// per map point: gates, window, candidates.  Rcw9 / tcw3 / Ow3: camera pose; stereo_gate: the chi-square test of the plain Fuse.
int fuse_core(olf_ctx* c, const char* who, const olf_frame_view* kf, const float* Rcw9, const float* tcw3, const float* Ow3, int n_mp,
              const uint8_t* skip, const float* world, const float* normal, const float* maxd, const float* mind, const uint8_t* desc, float th,
              bool stereo_gate, int none_dist, int32_t* best_idx, int32_t* best_dist, uint8_t* matched = nullptr, int32_t* kf_match = nullptr,
              int32_t* nmatches = nullptr)
{
    // matched != nullptr: SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (src/ORBmatcher.cc:292-405) -- key points that hold a match are
    // skipped (:378), a point whose best distance is <= TH_LOW takes its key point at once (:397-401), so later points see it taken
    if (bad_levels(kf)) { set_error(std::string(who) + ": n_levels / scale_factors missing"); return OLF_ERR_INVALID; }
    const float logSF = log_scale_factor(*kf);
    const Grid grid(*kf);
    Batch q;
    struct Meta { float u, v, ur; int level; };
    std::vector<Meta> meta;
    int taken = 0;
    for (int i = 0; i < n_mp; ++i) {
        if (best_idx) { best_idx[i] = -1; best_dist[i] = none_dist; }
        if (skip && skip[i]) continue;
        const float* p3Dw = world + 3 * (size_t)i;
        float p3Dc[3], u, v, invz;
        r3_apply(Rcw9, p3Dw, tcw3, p3Dc);
        if (!project_in_image(*kf, p3Dc, u, v, invz)) continue;
        const float ur = u - kf->mbf * invz;
        const float maxDistance = 1.2f * maxd[i], minDistance = 0.8f * mind[i];
        float PO[3]; double nrm = 0, dot = 0;
        for (int k = 0; k < 3; ++k) { PO[k] = p3Dw[k] - Ow3[k]; nrm += (double)PO[k] * (double)PO[k]; dot += (double)PO[k] * (double)normal[3 * (size_t)i + k]; }
        const float dist3D = (float)std::sqrt(nrm);
        // Depth must be inside the scale pyramid of the image
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        // Viewing angle must be less than 60 deg
        if (dot < 0.5 * dist3D) continue;
        const int nPredictedLevel = predict_scale(maxd[i], dist3D, logSF, kf->n_levels);
        const float radius = th * kf->scale_factors[nPredictedLevel];
        if (!grid.area(u, v, radius, -1, -1, q.cand)) continue;
        q.add(i, desc + 32 * (size_t)i);
        meta.push_back({u, v, ur, nPredictedLevel});
    }
    const int rc = q.run(c, kf->desc, kf->n);
    if (rc != OLF_OK) return rc;
    for (size_t k = 0; k < q.owner.size(); ++k) {
        const Meta& m = meta[k];
        int bestDist = none_dist, bestIdx = -1;
        for (int p = q.offs[k]; p < q.offs[k + 1]; ++p) {
            const int idx = q.cand[p];
            if (matched && matched[idx]) continue;
            const olf_keypoint& kp = kf->keys[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < m.level - 1 || kpLevel > m.level) continue;
            if (stereo_gate) {
                if (kpLevel < 0 || kpLevel >= kf->n_levels) { set_error(std::string(who) + ": octave outside mvScaleFactors"); return OLF_ERR_INVALID; }
                const float sigma2 = kf->scale_factors[kpLevel] * kf->scale_factors[kpLevel];
                const float invSigma2 = 1.0f / sigma2;                        // mvInvLevelSigma2, src/ORBextractor.cc:434-436
                const float ex = m.u - kp.x, ey = m.v - kp.y;
                if (kf->uright[idx] >= 0) {
                    // Check reprojection error in stereo
                    const float er = m.ur - kf->uright[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * invSigma2 > 7.8) continue;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * invSigma2 > 5.99) continue;
                }
            }
            const int dist = q.dist[p];
            if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
        }
        if (best_idx) { best_idx[q.owner[k]] = bestIdx; best_dist[q.owner[k]] = bestDist; }
        if (matched && bestDist <= TH_LOW) { matched[bestIdx] = 1; kf_match[bestIdx] = q.owner[k]; ++taken; }
    }
    if (nmatches) *nmatches = taken;
    return OLF_OK;
}

// Decompose Scw (src/ORBmatcher.cc:301-305, :985-989): scw = sqrt(row0 . row0); Rcw = sRcw / scw, tcw = Scw.col(3) / scw (a cv::Mat divided by a
// scalar is a scaling by the double 1/scw rounded to float); Ow = -Rcw.t() * tcw
void sim3_decompose(const float* Scw, float* R, float* t, float* ow)
{
    double d = 0;
    for (int k = 0; k < 3; ++k) d += (double)Scw[k] * (double)Scw[k];
    const float scw = (float)std::sqrt(d);
    const float inv = (float)(1.0 / (double)scw);
    float Rt[9];
    for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) { R[3 * r + k] = Scw[4 * r + k] * inv; Rt[3 * k + r] = R[3 * r + k]; } t[r] = Scw[4 * r + 3] * inv; }
    r3_apply(Rt, t, nullptr, ow, -1.0, true);
}
}  // namespace

extern "C" {

int olf_fuse_search(olf_ctx* c, const olf_frame_view* kf, int n_mp, const uint8_t* skip, const float* world, const float* normal,
                    const float* maxd, const float* mind, const uint8_t* desc, float th, const float* Ow, int32_t* best_idx, int32_t* best_dist)
{
    if (!c || bad_view(kf, true) || n_mp < 0 || !best_idx || !best_dist || !kf->scale_factors || (kf->n && !kf->uright) ||
        (n_mp && (!world || !normal || !maxd || !mind || !desc))) { set_error("olf_fuse_search: bad argument"); return OLF_ERR_INVALID; }
    float R[9], t[3], ow[3];
    for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) R[3 * r + k] = kf->Tcw[4 * r + k]; t[r] = kf->Tcw[4 * r + 3]; }
    if (Ow) std::memcpy(ow, Ow, sizeof(ow)); else camera_centre(kf->Tcw, ow);
    return fuse_core(c, "olf_fuse_search", kf, R, t, ow, n_mp, skip, world, normal, maxd, mind, desc, th, true, 256, best_idx, best_dist);
}

int olf_fuse_search_sim3(olf_ctx* c, const olf_frame_view* kf, const float* Scw, int n_mp, const uint8_t* skip, const float* world,
                         const float* normal, const float* maxd, const float* mind, const uint8_t* desc, float th, int32_t* best_idx,
                         int32_t* best_dist)
{
    if (!c || bad_view(kf, false) || !Scw || n_mp < 0 || !best_idx || !best_dist || !kf->scale_factors ||
        (n_mp && (!world || !normal || !maxd || !mind || !desc))) { set_error("olf_fuse_search_sim3: bad argument"); return OLF_ERR_INVALID; }
    float R[9], t[3], ow[3];
    sim3_decompose(Scw, R, t, ow);
    return fuse_core(c, "olf_fuse_search_sim3", kf, R, t, ow, n_mp, skip, world, normal, maxd, mind, desc, th, false, 2147483647, best_idx,
                     best_dist);
}

int olf_search_by_projection_sim3(olf_ctx* c, const olf_frame_view* kf, const float* Scw, int n_mp, const uint8_t* skip, const float* world,
                                  const float* normal, const float* maxd, const float* mind, const uint8_t* desc, float th, uint8_t* matched,
                                  int32_t* matches, int32_t* nmatches)
{
    if (!c || bad_view(kf, false) || !Scw || n_mp < 0 || !matched || !matches || !nmatches || !kf->scale_factors ||
        (n_mp && (!world || !normal || !maxd || !mind || !desc))) { set_error("olf_search_by_projection_sim3: bad argument"); return OLF_ERR_INVALID; }
    float R[9], t[3], ow[3];
    sim3_decompose(Scw, R, t, ow);
    for (int i = 0; i < kf->n; ++i) matches[i] = -1;
    return fuse_core(c, "olf_search_by_projection_sim3", kf, R, t, ow, n_mp, skip, world, normal, maxd, mind, desc, th, false, 256, nullptr, nullptr,
                     matched, matches, nmatches);
}

int olf_search_by_sim3(olf_ctx* c, const olf_frame_view* kf1, const olf_frame_view* kf2, int32_t* matches12, float s12, const float* R12,
                       const float* t12, float th, int32_t* vn_match1, int32_t* vn_match2, int32_t* nfound)
{
    auto incomplete = [](const olf_frame_view* k) {
        return k->n && (!k->mp_valid || !k->mp_bad || !k->mp_world || !k->mp_desc || !k->mp_maxd || !k->mp_mind);
    };
    if (!c || bad_view(kf1, true) || bad_view(kf2, true) || !matches12 || !R12 || !t12 || !vn_match1 || !vn_match2 || !nfound ||
        !kf1->scale_factors || !kf2->scale_factors || incomplete(kf1) || incomplete(kf2)) {
        set_error("olf_search_by_sim3: bad argument"); return OLF_ERR_INVALID;
    }
    if (bad_levels(kf1) || bad_levels(kf2)) { set_error("olf_search_by_sim3: n_levels / scale_factors missing"); return OLF_ERR_INVALID; }
    const int N1 = kf1->n, N2 = kf2->n;
    // Transformation between cameras (:1123-1125)
    float sR12[9], sR21[9], t21[3];
    const float inv = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) { sR12[3 * r + k] = s12 * R12[3 * r + k]; sR21[3 * r + k] = inv * R12[3 * k + r]; }
    r3_apply(sR21, t12, nullptr, t21, -1.0);
    std::vector<uint8_t> already1((size_t)N1, 0), already2((size_t)N2, 0);
    for (int i = 0; i < N1; i++) {
        if (matches12[i] != -1) {
            already1[i] = 1;
            const int idx2 = matches12[i];
            if (idx2 >= 0 && idx2 < N2) already2[idx2] = 1;
        }
    }
    const float logSF = log_scale_factor(*kf1);
    for (int dir = 0; dir < 2; ++dir) {
        // dir 0: the map points of KF1 into KF2 (:1157-1232); dir 1: those of KF2 into KF1 (:1235-1309)
        const olf_frame_view &src = dir ? *kf2 : *kf1, &dst = dir ? *kf1 : *kf2;
        const std::vector<uint8_t>& already = dir ? already2 : already1;
        const float *sR = dir ? sR12 : sR21, *t = dir ? t12 : t21;
        int32_t* out = dir ? vn_match2 : vn_match1;
        for (int i = 0; i < src.n; ++i) out[i] = -1;
        const Grid grid(dst);
        Batch q;
        std::vector<int> levels;
        for (int i = 0; i < src.n; ++i) {
            if (!src.mp_valid[i] || already[i]) continue;
            if (src.mp_bad[i]) continue;
            float pa[3], pb[3], u, v, invz;
            rot_apply(src.Tcw, src.mp_world + 3 * (size_t)i, 1.0f, pa);
            r3_apply(sR, pa, t, pb);
            if (!project_in_image(dst, pb, u, v, invz)) continue;
            const float maxDistance = 1.2f * src.mp_maxd[i], minDistance = 0.8f * src.mp_mind[i];
            double nrm = 0;
            for (int k = 0; k < 3; ++k) nrm += (double)pb[k] * (double)pb[k];
            const float dist3D = (float)std::sqrt(nrm);
            if (dist3D < minDistance || dist3D > maxDistance) continue;
            const int nPredictedLevel = predict_scale(src.mp_maxd[i], dist3D, logSF, dst.n_levels);
            const float radius = th * dst.scale_factors[nPredictedLevel];
            if (!grid.area(u, v, radius, -1, -1, q.cand)) continue;
            q.add(i, src.mp_desc + 32 * (size_t)i);
            levels.push_back(nPredictedLevel);
        }
        OLF_TRY(q.run(c, dst.desc, dst.n));
        for (size_t k = 0; k < q.owner.size(); ++k) {
            int bestDist = 2147483647, bestIdx = -1;
            for (int p = q.offs[k]; p < q.offs[k + 1]; ++p) {
                const int idx = q.cand[p];
                const int oct = dst.keys[idx].octave;
                if (oct < levels[k] - 1 || oct > levels[k]) continue;
                const int dist = q.dist[p];
                if (dist < bestDist) { bestDist = dist; bestIdx = idx; }
            }
            if (bestDist <= TH_HIGH) out[q.owner[k]] = bestIdx;
        }
    }
    // Check agreement
    int found = 0;
    for (int i1 = 0; i1 < N1; i1++) {
        const int idx2 = vn_match1[i1];
        if (idx2 >= 0 && vn_match2[idx2] == i1) { matches12[i1] = idx2; found++; }
    }
    *nfound = found;
    return OLF_OK;
}

// getLineCoords (src/gridStructure.cpp:33-41): the cells visited by the reference's double-precision Bresenham walk (src/LineIterator.cpp:34-77),
// host arithmetic -- the same walk csrc/linematch.hip makes on the device for the stereo line matcher.  xy: (x, y) pairs; *n receives the
// number of cells (which may exceed cap; only cap pairs are written).
int olf_line_coords(double x1, double y1, double x2, double y2, int32_t* xy, int cap, int32_t* n)
{
    if (!xy || !n || cap < 0) { set_error("olf_line_coords: bad argument"); return OLF_ERR_INVALID; }
    const bool steep = std::fabs(y2 - y1) > std::fabs(x2 - x1);
    if (steep) { std::swap(x1, y1); std::swap(x2, y2); }
    if (x1 > x2) { std::swap(x1, x2); std::swap(y1, y2); }
    const double dx = x2 - x1, dy = std::fabs(y2 - y1);
    double error = dx / 2.0;
    const int ystep = (y1 < y2) ? 1 : -1;
    int y = (int)y1, count = 0;
    const int maxX = (int)x2;
    for (int x = (int)x1; x <= maxX; ++x) {
        if (count < cap) { xy[2 * count] = steep ? y : x; xy[2 * count + 1] = steep ? x : y; }
        ++count;
        error -= dy;
        if (error < 0) { y += ystep; error += dx; }
        if (x == 2147483647) break;
    }
    *n = count;
    return OLF_OK;
}

}  // extern "C"
