// orb_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// Restates ORB_SLAM2::ORBextractor of the reference (paths relative to /root/reference):
//   ctor / tables            src/ORBextractor.cc:412-472
//   ComputePyramid           src/ORBextractor.cc:1109-1134      (+ cv::resize, App. A.2)
//   ComputeKeyPointsOctTree  src/ORBextractor.cc:767-855        (+ cv::FAST, App. A.4)
//   DistributeOctTree        src/ORBextractor.cc:541-765, DivideNode :483-539
//   IC_Angle                 src/ORBextractor.cc:79-106         (+ cv::fastAtan2, App. A.5)
//   blur + descriptors       src/ORBextractor.cc:1036-1107, computeOrbDescriptor :110-149
// PARITY UNPINNED (see oracle_common.hpp).  Convention C.1 (octree tie-break): nodes with equal
// key-point counts are expanded most-recently-created first.
#include "oracle_common.hpp"
#include "orb_oracle.hpp"
#include <chrono>
#include <list>
#include <utility>

namespace orc {

static const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;

static const int8_t kPattern[1024] = {
#include "orb_pattern_31.inc"
};


struct OrbTables {
    int nfeatures, nlevels, iniTh, minTh;
    double scaleFactor;  // the header declares the member as double (include/ORBextractor.h:103)
    std::vector<float> sf, inv_sf, sigma2, inv_sigma2;
    std::vector<int> nPerLevel, umax;
    explicit OrbTables(const olf_orb_params& p)
        : nfeatures(p.nfeatures), nlevels(p.nlevels), iniTh(p.ini_th_fast), minTh(p.min_th_fast), scaleFactor(p.scale_factor)
    {
        sf.resize(nlevels); sigma2.resize(nlevels); inv_sf.resize(nlevels); inv_sigma2.resize(nlevels);
        sf[0] = 1.0f; sigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; ++i) {
            sf[i] = (float)(sf[i - 1] * scaleFactor);
            sigma2[i] = sf[i] * sf[i];
        }
        for (int i = 0; i < nlevels; ++i) {
            inv_sf[i] = 1.0f / sf[i];
            inv_sigma2[i] = 1.0f / sigma2[i];
        }
        nPerLevel.resize(nlevels);
        float factor = (float)(1.0f / scaleFactor);
        float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int l = 0; l < nlevels - 1; ++l) {
            nPerLevel[l] = cvRoundf(nDesired);
            sum += nPerLevel[l];
            nDesired *= factor;
        }
        nPerLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
        umax.resize(HALF_PATCH_SIZE + 1);
        int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
        int vmin = cvCeil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cvRound(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }
};

// ---- cv::FAST(img, kps, threshold, true) on a sub-image (TYPE_9_16) ------------------------
static const int kRing[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                 {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// cornerScore<16>: threshold-seeded max over the sixteen 9-arcs of min |difference|, minus 1
static int corner_score16(const int d[25], int threshold)
{
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]); a = std::min(a, d[k + 5]); a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]); a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]); b = std::max(b, d[k + 4]); b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]); b = std::max(b, d[k + 7]); b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

struct FastKP { int x, y, score; };

// Restates FAST_t<16>: sub-image = rows [y0,y1), cols [x0,x1) of img.  Key points are
// returned in sub-image coordinates, row-major, after 3x3 non-max suppression on the score
// (non-corners and the 3-pixel frame count as 0; strict '>').
static void fast9_16(const Image& img, int x0, int y0, int x1, int y1, int threshold, std::vector<FastKP>& out)
{
    out.clear();
    const int w = x1 - x0, h = y1 - y0;
    if (w < 7 || h < 7) return;
    threshold = std::min(std::max(threshold, 0), 255);
    std::vector<uint8_t> score((size_t)w * h, 0);
    // ring offsets in this image's stride, and the darker / brighter class of every difference (FAST_t<16>'s threshold_tab: 1 = darker than v - t,
    // 2 = brighter than v + t)
    int off[25];
    for (int k = 0; k < 25; ++k) off[k] = kRing[k % 16][1] * img.w + kRing[k % 16][0];
    uint8_t tab[512];
    for (int i = -255; i <= 255; ++i) tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);
    for (int y = 3; y < h - 3; ++y) {
        const uint8_t* rowp = img.row(y0 + y) + x0;
        for (int x = 3; x < w - 3; ++x) {
            const uint8_t* ptr = rowp + x;
            const int v = ptr[0];
            const uint8_t* t = tab + 255 - v;      // t[ring value] = class of (ring value - v)
            // the result-neutral pre-test every FAST implementation has (FAST_t<16> does it in this order): 9 contiguous ring pixels of one class contain
            // one pixel of every opposite pair (k, k + 8), so the AND over the eight pairs of the OR of the pair's classes keeps that class's bit
            int d = t[ptr[off[0]]] | t[ptr[off[8]]];
            if (d == 0) continue;
            d &= t[ptr[off[2]]] | t[ptr[off[10]]];
            d &= t[ptr[off[4]]] | t[ptr[off[12]]];
            d &= t[ptr[off[6]]] | t[ptr[off[14]]];
            if (d == 0) continue;
            d &= t[ptr[off[1]]] | t[ptr[off[9]]];
            d &= t[ptr[off[3]]] | t[ptr[off[11]]];
            d &= t[ptr[off[5]]] | t[ptr[off[13]]];
            d &= t[ptr[off[7]]] | t[ptr[off[15]]];
            if (d == 0) continue;
            int ring[25];
            for (int k = 0; k < 25; ++k) ring[k] = ptr[off[k]];
            bool corner = false;
            if (d & 1) {   // >= 9 contiguous darker
                int vt = v - threshold, count = 0;
                for (int k = 0; k < 25; ++k) {
                    if (ring[k] < vt) { if (++count > 8) { corner = true; break; } }
                    else count = 0;
                }
            }
            if (!corner && (d & 2)) {  // >= 9 contiguous brighter
                int vt = v + threshold, count = 0;
                for (int k = 0; k < 25; ++k) {
                    if (ring[k] > vt) { if (++count > 8) { corner = true; break; } }
                    else count = 0;
                }
            }
            if (corner) {
                int dd[25];
                for (int k = 0; k < 25; ++k) dd[k] = v - ring[k];
                score[(size_t)y * w + x] = (uint8_t)corner_score16(dd, threshold);
            }
        }
    }
    for (int y = 3; y < h - 3; ++y)
        for (int x = 3; x < w - 3; ++x) {
            int s = score[(size_t)y * w + x];
            if (!s) continue;  // not a corner (a corner's score is >= threshold; threshold 0 corners score >= 0:
                               // cv::FAST keeps those too, but the path never uses threshold 0)
            const uint8_t* p = &score[(size_t)y * w + x];
            if (s > p[-1] && s > p[1] && s > p[-w - 1] && s > p[-w] && s > p[-w + 1] && s > p[w - 1] && s > p[w] && s > p[w + 1])
                out.push_back({x, y, s});
        }
}

// ---- DistributeOctTree ------------------------------------------------------------------
struct Node {
    std::vector<KP> keys;
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    std::list<Node>::iterator lit;
    bool noMore = false;
    long seq = 0;  // creation order (convention C.1)
};

static void divide_node(const Node& n, Node& n1, Node& n2, Node& n3, Node& n4)
{
    const int halfX = (int)std::ceil(static_cast<float>(n.URx - n.ULx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(n.BRy - n.ULy) / 2);
    n1.ULx = n.ULx; n1.ULy = n.ULy; n1.URx = n.ULx + halfX; n1.URy = n.ULy;
    n1.BLx = n.ULx; n1.BLy = n.ULy + halfY; n1.BRx = n.ULx + halfX; n1.BRy = n.ULy + halfY;
    n2.ULx = n1.URx; n2.ULy = n1.URy; n2.URx = n.URx; n2.URy = n.URy;
    n2.BLx = n1.BRx; n2.BLy = n1.BRy; n2.BRx = n.URx; n2.BRy = n.ULy + halfY;
    n3.ULx = n1.BLx; n3.ULy = n1.BLy; n3.URx = n1.BRx; n3.URy = n1.BRy;
    n3.BLx = n.BLx; n3.BLy = n.BLy; n3.BRx = n1.BRx; n3.BRy = n.BLy;
    n4.ULx = n3.URx; n4.ULy = n3.URy; n4.URx = n2.BRx; n4.URy = n2.BRy;
    n4.BLx = n3.BRx; n4.BLy = n3.BRy; n4.BRx = n.BRx; n4.BRy = n.BRy;
    for (const KP& kp : n.keys) {
        if (kp.x < n1.URx) {
            if (kp.y < n1.BRy) n1.keys.push_back(kp);
            else n3.keys.push_back(kp);
        } else if (kp.y < n1.BRy) n2.keys.push_back(kp);
        else n4.keys.push_back(kp);
    }
    if (n1.keys.size() == 1) n1.noMore = true;
    if (n2.keys.size() == 1) n2.noMore = true;
    if (n3.keys.size() == 1) n3.noMore = true;
    if (n4.keys.size() == 1) n4.noMore = true;
}

typedef std::pair<int, Node*> SizeNode;
static bool size_seq_less(const SizeNode& a, const SizeNode& b)
{
    if (a.first != b.first) return a.first < b.first;
    return a.second->seq < b.second->seq;  // stands in for the pointer compare (C.1)
}

static std::vector<KP> distribute_octree(const std::vector<KP>& toDistribute, int minX, int maxX, int minY, int maxY, int N)
{
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<Node> lNodes;
    std::vector<Node*> ini(nIni);
    long seq = 0;
    for (int i = 0; i < nIni; ++i) {
        Node ni;
        ni.ULx = (int)(hX * static_cast<float>(i)); ni.ULy = 0;
        ni.URx = (int)(hX * static_cast<float>(i + 1)); ni.URy = 0;
        ni.BLx = ni.ULx; ni.BLy = maxY - minY;
        ni.BRx = ni.URx; ni.BRy = maxY - minY;
        ni.seq = seq++;
        lNodes.push_back(ni);
        ini[i] = &lNodes.back();
    }
    for (const KP& kp : toDistribute) ini[(int)(kp.x / hX)]->keys.push_back(kp);
    for (auto lit = lNodes.begin(); lit != lNodes.end();) {
        if (lit->keys.size() == 1) { lit->noMore = true; ++lit; }
        else if (lit->keys.empty()) lit = lNodes.erase(lit);
        else ++lit;
    }
    bool finish = false;
    std::vector<SizeNode> vSize;
    auto push_child = [&](Node& c, int* nToExpand) {
        if (c.keys.size() > 0) {
            c.seq = seq++;
            lNodes.push_front(c);
            if (c.keys.size() > 1) {
                if (nToExpand) ++*nToExpand;
                vSize.push_back(std::make_pair((int)c.keys.size(), &lNodes.front()));
                lNodes.front().lit = lNodes.begin();
            }
        }
    };
    while (!finish) {
        int prevSize = (int)lNodes.size();
        auto lit = lNodes.begin();
        int nToExpand = 0;
        vSize.clear();
        while (lit != lNodes.end()) {
            if (lit->noMore) { ++lit; continue; }
            Node n1, n2, n3, n4;
            divide_node(*lit, n1, n2, n3, n4);
            push_child(n1, &nToExpand); push_child(n2, &nToExpand);
            push_child(n3, &nToExpand); push_child(n4, &nToExpand);
            lit = lNodes.erase(lit);
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) finish = true;
        else if (((int)lNodes.size() + nToExpand * 3) > N) {
            while (!finish) {
                prevSize = (int)lNodes.size();
                std::vector<SizeNode> prev = vSize;
                vSize.clear();
                std::sort(prev.begin(), prev.end(), size_seq_less);
                for (int j = (int)prev.size() - 1; j >= 0; --j) {
                    Node n1, n2, n3, n4;
                    divide_node(*prev[j].second, n1, n2, n3, n4);
                    push_child(n1, nullptr); push_child(n2, nullptr);
                    push_child(n3, nullptr); push_child(n4, nullptr);
                    lNodes.erase(prev[j].second->lit);
                    if ((int)lNodes.size() >= N) break;
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) finish = true;
            }
        }
    }
    std::vector<KP> result;
    for (auto& nd : lNodes) {
        const KP* best = &nd.keys[0];
        float maxResponse = best->response;
        for (size_t k = 1; k < nd.keys.size(); ++k)
            if (nd.keys[k].response > maxResponse) { best = &nd.keys[k]; maxResponse = nd.keys[k].response; }
        result.push_back(*best);
    }
    return result;
}

// ---- pyramid / key points / descriptors ---------------------------------------------------

// wall time (ms) of the calling thread's last orb_extract by stage: pyramid, FAST (both thresholds, incl. the 3x3 suppression), DistributeOctTree, IC_Angle,
// Gaussian blur, rBRIEF + record assembly (bench.py's cpu_baseline.stages_ms.orb_*)
thread_local double g_orb_stage_ms[6] = {0, 0, 0, 0, 0, 0};
namespace { struct StageClock { std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    double lap() { const auto n = std::chrono::steady_clock::now(); const double ms = std::chrono::duration<double, std::milli>(n - t).count(); t = n; return ms; } }; }

static void compute_pyramid(const Image& img, const OrbTables& T, std::vector<Image>& pyr)
{
    pyr.resize(T.nlevels);
    for (int l = 0; l < T.nlevels; ++l) {
        float scale = T.inv_sf[l];
        int sw = cvRoundf((float)img.w * scale), sh = cvRoundf((float)img.h * scale);
        if (l == 0) pyr[0] = img;
        else {
            const Image& p = pyr[l - 1];
            // cv::resize with explicit dsize: inv_scale = dsize/ssize, scale = 1/inv_scale
            double inv_x = (double)sw / p.w, inv_y = (double)sh / p.h;
            pyr[l] = resize_linear_u8(p, sw, sh, 1. / inv_x, 1. / inv_y);
        }
    }
}

static void compute_keypoints_octree(const OrbTables& T, OrbResult& R)
{
    const float W = 30;
    R.candidates.assign(T.nlevels, {});
    R.level_kps.assign(T.nlevels, {});
    for (int level = 0; level < T.nlevels; ++level) {
        const Image& im = R.pyramid[level];
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = im.w - EDGE_THRESHOLD + 3, maxBorderY = im.h - EDGE_THRESHOLD + 3;
        std::vector<KP>& cand = R.candidates[level];
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W), nRows = (int)(height / W);
        const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
        std::vector<FastKP> cell;
        for (int i = 0; i < nRows; ++i) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; ++j) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                StageClock fc;
                fast9_16(im, (int)iniX, (int)iniY, (int)maxX, (int)maxY, T.iniTh, cell);
                if (cell.empty()) fast9_16(im, (int)iniX, (int)iniY, (int)maxX, (int)maxY, T.minTh, cell);
                g_orb_stage_ms[1] += fc.lap();
                for (const FastKP& f : cell) {
                    KP kp;
                    kp.x = (float)f.x + j * wCell; kp.y = (float)f.y + i * hCell;
                    kp.size = 7.f; kp.angle = -1.f; kp.response = (float)f.score; kp.octave = 0;
                    cand.push_back(kp);
                }
            }
        }
        std::vector<KP>& kps = R.level_kps[level];
        StageClock oc;
        if (!cand.empty()) kps = distribute_octree(cand, minBorderX, maxBorderX, minBorderY, maxBorderY, T.nPerLevel[level]);
        g_orb_stage_ms[2] += oc.lap();
        const int scaledPatchSize = (int)(PATCH_SIZE * T.sf[level]);
        for (KP& kp : kps) {
            kp.x += minBorderX; kp.y += minBorderY;
            kp.octave = level; kp.size = (float)scaledPatchSize;
        }
    }
    // IC_Angle on the un-blurred level images
    StageClock ac;
    for (int level = 0; level < T.nlevels; ++level) {
        const Image& im = R.pyramid[level];
        for (KP& kp : R.level_kps[level]) {
            int m_01 = 0, m_10 = 0;
            const int cx = cvRoundf(kp.x), cy = cvRoundf(kp.y);
            for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * im.at(cx + u, cy);
            for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
                int v_sum = 0, d = T.umax[v];
                for (int u = -d; u <= d; ++u) {
                    int val_plus = im.at(cx + u, cy + v), val_minus = im.at(cx + u, cy - v);
                    v_sum += (val_plus - val_minus);
                    m_10 += u * (val_plus + val_minus);
                }
                m_01 += v * v_sum;
            }
            kp.angle = fastAtan2((float)m_01, (float)m_10);
        }
    }
    g_orb_stage_ms[3] = ac.lap();
}

static void compute_orb_descriptor(const KP& kpt, const Image& img, uint8_t* desc)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float angle = (float)kpt.angle * factorPI;
    float a = cosf(angle), b = sinf(angle);  // 'using namespace std' in the reference => cosf/sinf (C.6)
    const int cx = cvRoundf(kpt.x), cy = cvRoundf(kpt.y);
    const int8_t* pat = kPattern;
    auto get = [&](int idx) -> int {
        float px = (float)pat[2 * idx], py = (float)pat[2 * idx + 1];
        int yy = cvRoundf(px * b + py * a), xx = cvRoundf(px * a - py * b);
        return img.at(cx + xx, cy + yy);
    };
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int k = 0; k < 8; ++k) {
            int t0 = get(2 * k), t1 = get(2 * k + 1);
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

void orb_extract(const Image& img, const olf_orb_params& p, OrbResult& R)
{
    OrbTables T(p);
    for (double& v : g_orb_stage_ms) v = 0;
    StageClock sc;
    compute_pyramid(img, T, R.pyramid);
    g_orb_stage_ms[0] = sc.lap();
    compute_keypoints_octree(T, R);
    R.kps.clear(); R.desc.clear();
    R.blurred.assign(T.nlevels, Image());
    const std::vector<int> taps = gaussian_taps_q8(7, 2.0, p.conv_gauss_sum256);
    for (int level = 0; level < T.nlevels; ++level) {
        std::vector<KP>& kps = R.level_kps[level];
        if (kps.empty()) continue;
        sc.lap();
        R.blurred[level] = gaussian_blur_u8(R.pyramid[level], taps);
        g_orb_stage_ms[4] += sc.lap();
        size_t off = R.desc.size();
        R.desc.resize(off + kps.size() * 32);
        for (size_t i = 0; i < kps.size(); ++i) compute_orb_descriptor(kps[i], R.blurred[level], &R.desc[off + i * 32]);
        const float scale = T.sf[level];
        for (const KP& k : kps) {
            olf_keypoint o;
            o.x = k.x; o.y = k.y;
            if (level != 0) { o.x *= scale; o.y *= scale; }
            o.size = k.size; o.angle = k.angle; o.response = k.response; o.octave = k.octave; o.class_id = -1;
            R.kps.push_back(o);
        }
        g_orb_stage_ms[5] += sc.lap();
    }
}

void orb_scale_tables(const olf_orb_params& p, std::vector<float>& sf, std::vector<float>& inv_sf)
{
    OrbTables T(p);
    sf = T.sf; inv_sf = T.inv_sf;
}

}  // namespace orc

// ---- C entry points (ctypes) ----------------------------------------------------------------
using namespace orc;
extern "C" {

// scale / quota / umax tables (known-answer tests of SURVEY App. B)
int orc_orb_tables(const olf_orb_params* p, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2, int* n_per_level, int* umax16)
{
    OrbTables T(*p);
    for (int i = 0; i < T.nlevels; ++i) {
        if (sf) sf[i] = T.sf[i];
        if (inv_sf) inv_sf[i] = T.inv_sf[i];
        if (sigma2) sigma2[i] = T.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = T.inv_sigma2[i];
        if (n_per_level) n_per_level[i] = T.nPerLevel[i];
    }
    if (umax16) for (int i = 0; i < 16; ++i) umax16[i] = T.umax[i];
    return 0;
}

// level sizes of the pyramid for a w x h input
int orc_orb_level_sizes(const olf_orb_params* p, int w, int h, int* lw, int* lh)
{
    OrbTables T(*p);
    for (int l = 0; l < T.nlevels; ++l) {
        lw[l] = cvRoundf((float)w * T.inv_sf[l]);
        lh[l] = cvRoundf((float)h * T.inv_sf[l]);
    }
    return 0;
}

// Full extraction.  Optional debug outputs (may be null):
//   pyr_out / blur_out : concatenated level images (sum of lw*lh bytes)
//   cand_out           : per level candidates as int32 triples (x,y,score), cand_counts[nlevels], capacity cand_cap triples/level
int orc_orb_extract(const uint8_t* img, int w, int h, int stride, const olf_orb_params* p,
                    olf_keypoint* kps, uint8_t* desc, int cap, int* n,
                    uint8_t* pyr_out, uint8_t* blur_out, int32_t* cand_out, int* cand_counts, int cand_cap)
{
    if (!img || w < 64 || h < 64) return OLF_ERR_INVALID;
    Image im(w, h);
    for (int y = 0; y < h; ++y) std::memcpy(im.row(y), img + (size_t)y * stride, w);
    OrbResult R;
    orb_extract(im, *p, R);
    *n = (int)R.kps.size();
    if (*n > cap) return OLF_ERR_CAPACITY;
    if (*n) {
        std::memcpy(kps, R.kps.data(), R.kps.size() * sizeof(olf_keypoint));
        std::memcpy(desc, R.desc.data(), R.desc.size());
    }
    size_t off = 0;
    for (int l = 0; l < p->nlevels; ++l) {
        size_t sz = R.pyramid[l].d.size();
        if (pyr_out) std::memcpy(pyr_out + off, R.pyramid[l].d.data(), sz);
        if (blur_out && !R.blurred[l].d.empty()) std::memcpy(blur_out + off, R.blurred[l].d.data(), sz);
        off += sz;
        if (cand_out) {
            int c = (int)R.candidates[l].size();
            cand_counts[l] = c;
            for (int i = 0; i < std::min(c, cand_cap); ++i) {
                cand_out[((size_t)l * cand_cap + i) * 3 + 0] = (int)R.candidates[l][i].x;
                cand_out[((size_t)l * cand_cap + i) * 3 + 1] = (int)R.candidates[l][i].y;
                cand_out[((size_t)l * cand_cap + i) * 3 + 2] = (int)R.candidates[l][i].response;
            }
        }
    }
    return OLF_OK;
}

int orc_hamming256(const uint8_t* a, const uint8_t* b) { return hamming256(a, b); }

int orc_resize_linear(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, double scale_x, double scale_y)
{
    Image s(sw, sh);
    std::memcpy(s.d.data(), src, (size_t)sw * sh);
    Image d = resize_linear_u8(s, dw, dh, scale_x, scale_y);
    std::memcpy(dst, d.d.data(), (size_t)dw * dh);
    return 0;
}

int orc_gaussian_blur(const uint8_t* src, int w, int h, uint8_t* dst, int ksize, double sigma, int* taps_out)
{
    Image s(w, h);
    std::memcpy(s.d.data(), src, (size_t)w * h);
    std::vector<int> taps = gaussian_taps_q8(ksize, sigma);
    if (taps_out) for (int i = 0; i < ksize; ++i) taps_out[i] = taps[i];
    Image d = gaussian_blur_u8(s, taps);
    std::memcpy(dst, d.d.data(), (size_t)w * h);
    return 0;
}

float orc_fast_atan2(float y, float x) { return fastAtan2(y, x); }

// the 8-bit fixed-point Gaussian taps under either convention C.11 variant
int orc_gaussian_taps(int ksize, double sigma, int sum256, int* taps_out)
{
    std::vector<int> t = gaussian_taps_q8(ksize, sigma, sum256);
    for (int i = 0; i < ksize; ++i) taps_out[i] = t[i];
    return 0;
}

}  // extern "C"
