// linematch_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// Restates (paths relative to /root/reference):
//   Frame::ComputeStereoMatches_Lines      src/Frame.cc:878-1000 (+ lineSegmentOverlapStereo :1002-1037,
//                                          filterLineSegmentDisparity :1039-1048)
//   matchGrid (lines)                      src/LineMatcher.cpp:220-299
//   GridStructure / getLineCoords          src/gridStructure.cpp:33-83
//   LineIterator (double Bresenham)        src/LineIterator.cpp:34-77
// The two STL-only files are also compiled from the reference itself into oracle/_ref/libref_grid.so
// and tests/test_oracle_cpu.py checks this restatement against them (the only part of the path the
// reference can pin here).  Candidate iteration uses std::unordered_set<int> with the reference's
// insertion sequence (convention C.2); with min_ratio_12_l < 1 the order provably cannot change the
// result (a tie for the best distance always fails the ratio test).
#include "oracle_common.hpp"
#include <limits>
#include <list>
#include <unordered_set>
#include <utility>

namespace orc {

// ---- LineIterator / getLineCoords ---------------------------------------------------------------
void line_coords(double x1_, double y1_, double x2_, double y2_, std::vector<std::pair<int, int>>& out)
{
    out.clear();
    double x1 = x1_, y1 = y1_, x2 = x2_, y2 = y2_;
    const bool steep = std::abs(y2_ - y1_) > std::abs(x2_ - x1_);
    if (steep) { std::swap(x1, y1); std::swap(x2, y2); }
    if (x1 > x2) { std::swap(x1, x2); std::swap(y1, y2); }
    const double dx = x2 - x1, dy = std::abs(y2 - y1);
    double error = dx / 2.0;
    const int ystep = (y1 < y2) ? 1 : -1;
    int x = static_cast<int>(x1), y = static_cast<int>(y1);
    const int maxX = static_cast<int>(x2);
    while (x <= maxX) {
        out.push_back(steep ? std::make_pair(y, x) : std::make_pair(x, y));
        error -= dy;
        if (error < 0) { y += ystep; error += dx; }
        ++x;
    }
}

struct Grid {
    int rows, cols;
    std::vector<std::vector<std::list<int>>> g;   // [x][y]
    Grid(int r, int c) : rows(r), cols(c), g(c, std::vector<std::list<int>>(r)) {}
    void push(int x, int y, int idx) { if (x >= 0 && x < cols && y >= 0 && y < rows) g[x][y].push_back(idx); }
    void get(int x, int y, int wl, int wr, int hu, int hd, std::unordered_set<int>& out) const
    {
        const int min_x = std::max(0, x - wl), max_x = std::min(cols, x + wr + 1);
        const int min_y = std::max(0, y - hu), max_y = std::min(rows, y + hd + 1);
        for (int x_ = min_x; x_ < max_x; ++x_)
            for (int y_ = min_y; y_ < max_y; ++y_) out.insert(g[x_][y_].begin(), g[x_][y_].end());
    }
};

static inline double dot2(const std::pair<double, double>& a, const std::pair<double, double>& b) { return a.first * b.first + a.second * b.second; }
static inline void normalize2(std::pair<double, double>& v)
{
    const double m = std::sqrt(dot2(v, v));
    v.first /= m; v.second /= m;
}

static double overlap_stereo(double spl_obs, double epl_obs, double spl_proj, double epl_proj, double line_horiz_th)
{
    double overlap = 1.f;
    if (std::fabs(epl_obs - spl_obs) > line_horiz_th) {
        const double sln = std::min(spl_obs, epl_obs), eln = std::max(spl_obs, epl_obs);
        const double spn = std::min(spl_proj, epl_proj), epn = std::max(spl_proj, epl_proj);
        const double length = eln - spn;
        if ((epn < sln) || (spn > eln)) overlap = 0.f;
        else {
            if ((epn > eln) && (spn < sln)) overlap = eln - sln;
            else overlap = std::min(eln, epn) - std::max(sln, spn);
        }
        if (length > 0.01f) overlap = overlap / length;
        else overlap = 0.f;
        if (overlap > 1.f) overlap = 1.f;
    }
    return overlap;
}

void stereo_lines(const std::vector<olf_keyline>& klL, const uint8_t* descL, const std::vector<olf_keyline>& klR, const uint8_t* descR,
                  int img_w, int img_h, const olf_stereo_params& P, std::vector<int>& matches_12, std::vector<float>& disp,
                  std::vector<double>& le)
{
    const int NL = (int)klL.size(), NR = (int)klR.size();
    matches_12.assign(NL, -1);
    disp.assign((size_t)NL * 2, -1.f);
    le.assign((size_t)NL * 3, 0.0);
    if (NL == 0 || NR == 0) return;
    const double inv_width = OLF_GRID_COLS / static_cast<double>(img_w);
    const double inv_height = OLF_GRID_ROWS / static_cast<double>(img_h);
    typedef std::pair<int, int> point_2d;
    std::vector<std::pair<point_2d, point_2d>> coords;
    for (const olf_keyline& kl : klL)
        coords.push_back(std::make_pair(point_2d((int)(kl.startPointX * inv_width), (int)(kl.startPointY * inv_height)),
                                        point_2d((int)(kl.endPointX * inv_width), (int)(kl.endPointY * inv_height))));
    Grid grid(OLF_GRID_ROWS, OLF_GRID_COLS);
    std::vector<std::pair<double, double>> directions(NR);
    std::vector<std::pair<int, int>> lc;
    for (int idx = 0; idx < NR; ++idx) {
        const olf_keyline& kl = klR[idx];
        std::pair<double, double>& v = directions[idx];
        v = std::make_pair((kl.endPointX - kl.startPointX) * inv_width, (kl.endPointY - kl.startPointY) * inv_height);
        normalize2(v);
        line_coords(kl.startPointX * inv_width, kl.startPointY * inv_height, kl.endPointX * inv_width, kl.endPointY * inv_height, lc);
        for (const auto& p : lc) grid.push(p.first, p.second, idx);
    }
    // ---- matchGrid
    std::vector<int> matches_21, distances;
    if (P.best_lr_matches) { matches_21.assign(NR, -1); distances.assign(NR, std::numeric_limits<int>::max()); }
    for (int i1 = 0; i1 < NL; ++i1) {
        int best_d = std::numeric_limits<int>::max(), best_d2 = std::numeric_limits<int>::max(), best_idx = -1;
        const point_2d sp = coords[i1].first, ep = coords[i1].second;
        std::pair<double, double> v = std::make_pair(ep.first - sp.first, ep.second - sp.second);
        normalize2(v);
        std::unordered_set<int> candidates;
        grid.get(sp.first, sp.second, P.matching_s_ws, 0, 0, 0, candidates);
        grid.get(ep.first, ep.second, P.matching_s_ws, 0, 0, 0, candidates);
        if (candidates.empty()) continue;
        for (const int& i2 : candidates) {
            if (i2 < 0 || i2 >= NR) continue;
            if (std::abs(dot2(v, directions[i2])) < P.line_sim_th) continue;
            const int d = hamming256(descL + (size_t)i1 * 32, descR + (size_t)i2 * 32);
            if (P.best_lr_matches) {
                if (d < distances[i2]) { distances[i2] = d; matches_21[i2] = i1; }
                else continue;
            }
            if (d < best_d) { best_d2 = best_d; best_d = d; best_idx = i2; }
            else if (d < best_d2) best_d2 = d;
        }
        if (best_d < best_d2 * P.min_ratio_12_l) matches_12[i1] = best_idx;
    }
    if (P.best_lr_matches)
        for (int i1 = 0; i1 < NL; ++i1) {
            int& i2 = matches_12[i1];
            if (i2 >= 0 && matches_21[i2] != i1) i2 = -1;
        }
    // ---- end-point disparities (src/Frame.cc:930-960)
    for (int i1 = 0; i1 < NL; ++i1) {
        const int i2 = matches_12[i1];
        if (i2 < 0) continue;
        const double sp_l[3] = {klL[i1].startPointX, klL[i1].startPointY, 1.0};
        const double ep_l[3] = {klL[i1].endPointX, klL[i1].endPointY, 1.0};
        double le_l[3] = {sp_l[1] * ep_l[2] - sp_l[2] * ep_l[1], sp_l[2] * ep_l[0] - sp_l[0] * ep_l[2], sp_l[0] * ep_l[1] - sp_l[1] * ep_l[0]};
        const double nrm = std::sqrt(le_l[0] * le_l[0] + le_l[1] * le_l[1]);
        if (P.conv_eigen_recip) { const double inv = 1.0 / nrm; le_l[0] = le_l[0] * inv; le_l[1] = le_l[1] * inv; le_l[2] = le_l[2] * inv; }      // Eigen 3.0 / 3.1: v / s = v * (1 / s)
        else { le_l[0] = le_l[0] / nrm; le_l[1] = le_l[1] / nrm; le_l[2] = le_l[2] / nrm; }
        double sp_r[3] = {klR[i2].startPointX, klR[i2].startPointY, 1.0};
        double ep_r[3] = {klR[i2].endPointX, klR[i2].endPointY, 1.0};
        const double overlap = overlap_stereo(sp_l[1], ep_l[1], sp_r[1], ep_r[1], P.line_horiz_th);
        // the reference overwrites sp_r first and then evaluates ep_r with the NEW sp_r (comma initialiser)
        const double nsx = (sp_r[0] * (sp_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - sp_l[1])) / (sp_r[1] - ep_r[1]);
        sp_r[0] = nsx; sp_r[1] = sp_l[1]; sp_r[2] = 1.0;
        const double nex = (sp_r[0] * (ep_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - ep_l[1])) / (sp_r[1] - ep_r[1]);
        ep_r[0] = nex; ep_r[1] = ep_l[1]; ep_r[2] = 1.0;
        double disp_s = sp_l[0] - sp_r[0], disp_e = ep_l[0] - ep_r[0];
        if (std::min(disp_s, disp_e) / std::max(disp_s, disp_e) < P.ls_min_disp_ratio) { disp_s = -1.0; disp_e = -1.0; }
        if (disp_s >= P.min_disp && disp_e >= P.min_disp && std::abs(sp_l[1] - ep_l[1]) > P.line_horiz_th &&
            std::abs(sp_r[1] - ep_r[1]) > P.line_horiz_th && overlap > P.stereo_overlap_th) {
            disp[2 * i1] = (float)disp_s; disp[2 * i1 + 1] = (float)disp_e;
            le[3 * i1] = le_l[0]; le[3 * i1 + 1] = le_l[1]; le[3 * i1 + 2] = le_l[2];
        }
    }
}

}  // namespace orc

using namespace orc;
extern "C" {

int orc_line_coords(double x1, double y1, double x2, double y2, int* xy, int cap)
{
    std::vector<std::pair<int, int>> lc;
    line_coords(x1, y1, x2, y2, lc);
    for (int i = 0; i < std::min((int)lc.size(), cap); ++i) { xy[2 * i] = lc[i].first; xy[2 * i + 1] = lc[i].second; }
    return (int)lc.size();
}

// same contract as ref_grid_query in oracle/ref_shim.cpp
int orc_grid_query(int rows, int cols, const double* segs, int n, int qx, int qy, int wl, int wr, int hu, int hd, int qx2, int qy2,
                   int use_second, int* out, int cap)
{
    Grid grid(rows, cols);
    std::vector<std::pair<int, int>> lc;
    for (int i = 0; i < n; ++i) {
        line_coords(segs[4 * i], segs[4 * i + 1], segs[4 * i + 2], segs[4 * i + 3], lc);
        for (auto& p : lc) grid.push(p.first, p.second, i);
    }
    std::unordered_set<int> cand;
    grid.get(qx, qy, wl, wr, hu, hd, cand);
    if (use_second) grid.get(qx2, qy2, wl, wr, hu, hd, cand);
    int k = 0;
    for (int v : cand) { if (k < cap) out[k] = v; ++k; }
    return k;
}

int orc_stereo_lines(const olf_keyline* klL, const uint8_t* descL, int nL, const olf_keyline* klR, const uint8_t* descR, int nR, int w, int h,
                     const olf_stereo_params* P, int* matches12, float* disp, double* le)
{
    std::vector<olf_keyline> a(klL, klL + nL), b(klR, klR + nR);
    std::vector<int> m;
    std::vector<float> d;
    std::vector<double> l;
    stereo_lines(a, descL, b, descR, w, h, *P, m, d, l);
    for (int i = 0; i < nL; ++i) {
        matches12[i] = m[i];
        disp[2 * i] = d[2 * i]; disp[2 * i + 1] = d[2 * i + 1];
        le[3 * i] = l[3 * i]; le[3 * i + 1] = l[3 * i + 1]; le[3 * i + 2] = l[3 * i + 2];
    }
    return OLF_OK;
}

}  // extern "C"
