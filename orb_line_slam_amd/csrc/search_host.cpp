// search_host.cpp -- host side of the per-frame ORBmatcher searches behind the C ABI (olf_search_by_projection, olf_search_by_bow,
// olf_search_local_map).  Split of the reference's loops, as SURVEY 8(b) / App. C.7 prescribe:
//   1. candidate generation on the host, exactly as the reference walks them (Frame::GetFeaturesInArea over the 64 x 48 grid,
//      src/Frame.cc:517-570; the merge of two DBoW2 feature vectors, src/ORBmatcher.cc:176-203),
//   2. every DescriptorDistance of the search in one k_match_candidates launch (match.hip),
//   3. the reference's greedy resolution, which depends on the map points assigned so far, on the host in the reference's order.
// Float expressions are written as the reference writes them (src/ORBmatcher.cc, float unless a double literal promotes them); the
// library is built with -ffp-contract=off.
#include <algorithm>
#include <cmath>
#include <cstring>
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

// cv::Mat products of CV_32F operands (cv::gemm): double accumulation, one rounding (convention of DESIGN.md App. C)
void rot_apply(const float* T, const float* v, float alpha_t, float* out)     // R * v + alpha_t * t, T = 4x4 row-major
{
    for (int r = 0; r < 3; ++r) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += (double)T[4 * r + k] * (double)v[k];
        out[r] = (float)(acc + (double)alpha_t * (double)T[4 * r + 3]);
    }
}

bool bad_view(const olf_frame_view* f, bool needs_pose)
{
    return !f || f->n < 0 || (f->n && (!f->keys || !f->desc)) || (needs_pose && !f->Tcw);
}
}  // namespace

extern "C" {

int olf_search_by_projection(olf_ctx* c, const olf_frame_view* cur, const olf_frame_view* last, float th, int bMono, int check_orientation,
                             int32_t* matches, int32_t* nmatches)
{
    if (!c || bad_view(cur, true) || bad_view(last, true) || !matches || !nmatches || !cur->scale_factors || !cur->mp_valid || !cur->mp_obs ||
        (cur->n && !cur->uright) || (last->n && (!last->mp_valid || !last->mp_world || !last->mp_desc || !last->mp_obs))) {
        set_error("olf_search_by_projection: bad argument"); return OLF_ERR_INVALID;
    }
    for (int i = 0; i < cur->n; ++i) matches[i] = -1;
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
            n++;
            if (check_orientation) rotHist[rot_bin(last->keys[i].angle, cur->keys[bestIdx2].angle)].push_back(bestIdx2);
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
