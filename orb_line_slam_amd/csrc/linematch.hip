// linematch.hip -- Frame::ComputeStereoMatches_Lines (reference src/Frame.cc:878-1000) with
// matchGrid(lines) (src/LineMatcher.cpp:220-299), GridStructure::get (src/gridStructure.cpp:65-76) and the
// double-precision Bresenham of src/LineIterator.cpp:34-77, gfx950.
//
// The reference rasterises every right line into a 64x48 grid of std::list<int> and collects, per left
// line, the ids found in a 1-row window left of both end points.  A digital line covers a contiguous
// span of cells in each grid row, so "right line i2 is in the window" == "its span in that row overlaps
// [x-ws, x]": we keep, per right line, the (first, last) cell of each of the 48 rows and never build
// lists.  The running per-right-line best distance (`distances[i2]`, Config::bestLRMatches()) is a
// strict prefix minimum over left lines in index order: one thread per right line walks the left
// lines.  Candidate order inside the reference's unordered_set cannot change the result while
// min_ratio_12_l < 1 (a tie for the best distance always fails the ratio test), see DESIGN.md.
#include "line_internal.hpp"
#include "device_math.hpp"

namespace olf {

constexpr int GC = OLF_GRID_COLS, GR = OLF_GRID_ROWS;

struct LinePrep {
    int spx, spy, epx, epy;     // truncated grid cells of the end points
    double vx, vy;              // normalised direction (grid units)
    unsigned char lo[GR], hi[GR];   // per grid row: first/last covered cell, lo > hi = row not covered
};

__device__ __forceinline__ int ham256_u4(const uint4* a, const uint4* b)
{
    const uint4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
           __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// one thread per line (left and right): grid coordinates, direction, per-row cell spans
__global__ __launch_bounds__(64) void k_lines_prep(const olf_keyline* __restrict__ kls, const int* __restrict__ counts, int cap, int W, int H,
                                                   LinePrep* __restrict__ prep)
{
    const int image = blockIdx.y, i = blockIdx.x * 64 + threadIdx.x;
    if (i >= counts[image]) return;
    const olf_keyline kl = kls[(size_t)image * cap + i];
    const double inv_width = (double)GC / (double)W, inv_height = (double)GR / (double)H;
    LinePrep P;
    const double sx = d_mul((double)kl.startPointX, inv_width), sy = d_mul((double)kl.startPointY, inv_height);
    const double ex = d_mul((double)kl.endPointX, inv_width), ey = d_mul((double)kl.endPointY, inv_height);
    P.spx = (int)sx; P.spy = (int)sy; P.epx = (int)ex; P.epy = (int)ey;
    if ((image & 1) == 0) {
        // left line: direction of the truncated end points (src/LineMatcher.cpp:249-250)
        double vx = (double)(P.epx - P.spx), vy = (double)(P.epy - P.spy);
        const double m = sqrt(d_add(d_mul(vx, vx), d_mul(vy, vy)));
        P.vx = vx / m; P.vy = vy / m;
    } else {
        // right line: direction in grid units (src/Frame.cc:913-915)
        double vx = d_mul((double)f_sub(kl.endPointX, kl.startPointX), inv_width), vy = d_mul((double)f_sub(kl.endPointY, kl.startPointY), inv_height);
        const double m = sqrt(d_add(d_mul(vx, vx), d_mul(vy, vy)));
        P.vx = vx / m; P.vy = vy / m;
    }
#pragma unroll
    for (int r = 0; r < GR; ++r) { P.lo[r] = 255; P.hi[r] = 0; }
    if (image & 1) {
        // LineIterator(x1,y1,x2,y2) + getNext (src/LineIterator.cpp)
        double x1 = sx, y1 = sy, x2 = ex, y2 = ey;
        const bool steep = fabs(d_sub(y2, y1)) > fabs(d_sub(x2, x1));
        if (steep) { double t = x1; x1 = y1; y1 = t; t = x2; x2 = y2; y2 = t; }
        if (x1 > x2) { double t = x1; x1 = x2; x2 = t; t = y1; y1 = y2; y2 = t; }
        const double dx = d_sub(x2, x1), dy = fabs(d_sub(y2, y1));
        double error = dx / 2.0;
        const int ystep = (y1 < y2) ? 1 : -1;
        int x = (int)x1, y = (int)y1;
        const int maxX = (int)x2;
        for (int it = 0; it < 4096 && x <= maxX; ++it) {
            const int cx = steep ? y : x, cy = steep ? x : y;
            if (cx >= 0 && cx < GC && cy >= 0 && cy < GR) {
                P.lo[cy] = (unsigned char)min((int)P.lo[cy], cx);
                P.hi[cy] = (unsigned char)max((int)P.hi[cy], cx);
            }
            error = d_sub(error, dy);
            if (error < 0) { y += ystep; error = d_add(error, dx); }
            ++x;
        }
    }
    prep[(size_t)image * cap + i] = P;
}

__device__ __forceinline__ bool in_window(const LinePrep& R, int cx, int cy, int ws)
{
    if (cy < 0 || cy >= GR) return false;
    const int lo = max(0, cx - ws), hi = min(GC, cx + 1) - 1;   // GridStructure::get: [x-ws, x+0+1)
    if (lo > hi) return false;
    return (int)R.lo[cy] <= hi && (int)R.hi[cy] >= lo;
}

// one thread per right line i2: walk left lines in order, keep the strict running minimum
__global__ __launch_bounds__(64) void k_lines_dist(const LinePrep* __restrict__ prep, const uint8_t* __restrict__ desc,
                                                   const int* __restrict__ counts, int cap, int ws, double sim_th, int best_lr,
                                                   uint16_t* __restrict__ dist, int* __restrict__ m21)
{
    const int pair = blockIdx.y, i2 = blockIdx.x * 64 + threadIdx.x;
    const int nL = counts[2 * pair], nR = counts[2 * pair + 1];
    if (i2 >= nR) return;
    // (a local copy whose row spans are indexed by a run-time row: 136 bytes of scratch, L1 resident.  Measured against it, round 5: reading the spans through the
    // pointer instead takes the stage from 4.7 to 25 ms per 3072 pairs, k_lines_prep's spans in LDS from 4.7 to 5.6 -- the scratch stays)
    const LinePrep R = prep[(size_t)(2 * pair + 1) * cap + i2];
    const uint4* dR = reinterpret_cast<const uint4*>(desc + ((size_t)(2 * pair + 1) * cap + i2) * OLF_DESC_BYTES);
    int running = 0x7fffffff, who = -1;
    uint16_t* col = dist + (size_t)pair * cap * cap + i2;
    for (int i1 = 0; i1 < nL; ++i1) {
        const LinePrep& Lp = prep[(size_t)(2 * pair) * cap + i1];
        uint16_t outv = 0xffff;
        if (in_window(R, Lp.spx, Lp.spy, ws) || in_window(R, Lp.epx, Lp.epy, ws)) {
            const double dt = d_add(d_mul(Lp.vx, R.vx), d_mul(Lp.vy, R.vy));
            if (!(fabs(dt) < sim_th)) {
                const int d = ham256_u4(reinterpret_cast<const uint4*>(desc + ((size_t)(2 * pair) * cap + i1) * OLF_DESC_BYTES), dR);
                if (best_lr) {
                    if (d < running) { running = d; who = i1; outv = (uint16_t)d; }
                } else outv = (uint16_t)d;
            }
        }
        col[(size_t)i1 * cap] = outv;
    }
    m21[(size_t)pair * cap + i2] = who;
}

// The same matrix with ONE WAVE per right line (few pairs per call -- the drop-in's online shape, where one thread per right line leaves a pair to eight
// waves walking 500 left lines each): the lanes take 64 consecutive left lines, the strict running minimum becomes an exclusive prefix minimum over the lanes
// seeded with the carry of the earlier steps -- left line i1 is a new record iff its distance is below everything before it, exactly the sequential scan.
__device__ __forceinline__ bool in_window_g(const LinePrep* __restrict__ R, int cx, int cy, int ws)
{
    if (cy < 0 || cy >= GR) return false;
    const int lo = max(0, cx - ws), hi = min(GC, cx + 1) - 1;
    if (lo > hi) return false;
    return (int)R->lo[cy] <= hi && (int)R->hi[cy] >= lo;
}
__global__ __launch_bounds__(256) void k_lines_dist_w(const LinePrep* __restrict__ prep, const uint8_t* __restrict__ desc,
                                                     const int* __restrict__ counts, int cap, int ws, double sim_th, int best_lr,
                                                     uint16_t* __restrict__ dist, int* __restrict__ m21)
{
    const int pair = blockIdx.y, i2 = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nL = counts[2 * pair], nR = counts[2 * pair + 1];
    if (i2 >= nR) return;
    const LinePrep* Rp = prep + (size_t)(2 * pair + 1) * cap + i2;
    const double Rvx = Rp->vx, Rvy = Rp->vy;
    const uint4* dR = reinterpret_cast<const uint4*>(desc + ((size_t)(2 * pair + 1) * cap + i2) * OLF_DESC_BYTES);
    const int INF = 0x7fffffff;
    int running = INF, who = -1;
    uint16_t* col = dist + (size_t)pair * cap * cap + i2;
    for (int base = 0; base < nL; base += 64) {
        const int i1 = base + lane;
        int d = INF;
        if (i1 < nL) {
            const LinePrep* Lp = prep + (size_t)(2 * pair) * cap + i1;
            const int spx = Lp->spx, spy = Lp->spy, epx = Lp->epx, epy = Lp->epy;
            if (in_window_g(Rp, spx, spy, ws) || in_window_g(Rp, epx, epy, ws)) {
                const double dt = d_add(d_mul(Lp->vx, Rvx), d_mul(Lp->vy, Rvy));
                if (!(fabs(dt) < sim_th)) d = ham256_u4(reinterpret_cast<const uint4*>(desc + ((size_t)(2 * pair) * cap + i1) * OLF_DESC_BYTES), dR);
            }
        }
        uint16_t outv = 0xffff;
        if (best_lr) {
            int e = d;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(e, o); if (lane >= o) e = min(e, t); }
            int excl = __shfl_up(e, 1);
            if (lane == 0) excl = INF;
            excl = min(excl, running);
            const bool rec = d != INF && d < excl;
            if (rec) outv = (uint16_t)d;
            const unsigned long long rm = __ballot(rec);
            if (rm) { const int l = 63 - __builtin_clzll(rm); running = __shfl(d, l); who = base + l; }
        } else if (d != INF) outv = (uint16_t)d;
        if (i1 < nL) col[(size_t)i1 * cap] = outv;
    }
    if (lane == 0) m21[(size_t)pair * cap + i2] = who;
}

// one wave per left line: best / second best over the considered candidates (the row of the distance matrix is read coalesced, lanes
// stride over the right lines; first index wins a tie, the second best is the second smallest value of the multiset -- what the
// reference's sequential scan produces), ratio + mutual test, then the end-point disparities of src/Frame.cc:930-960 on lane 0
__global__ __launch_bounds__(256) void k_lines_resolve(const olf_keyline* __restrict__ kls, const int* __restrict__ counts, int cap,
                                                       const uint16_t* __restrict__ dist, const int* __restrict__ m21, olf_stereo_params P,
                                                       int* __restrict__ m12, float* __restrict__ disp, double* __restrict__ le)
{
    const int pair = blockIdx.y, i1 = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nL = counts[2 * pair], nR = counts[2 * pair + 1];
    if (i1 >= nL) return;
    const uint16_t* row = dist + (size_t)pair * cap * cap + (size_t)i1 * cap;
    unsigned key = 0xffffffffu;            // (distance << 16 | index) of the lane's best
    int second = 0x7fffffff;               // the lane's second smallest distance
    for (int i2 = lane; i2 < nR; i2 += 64) {
        const int d = row[i2];
        if (d == 0xffff) continue;
        const unsigned k = ((unsigned)d << 16) | (unsigned)i2;
        if (k < key) { if (key != 0xffffffffu) second = (int)(key >> 16); key = k; }      // lane indices ascend: k < key <=> d < best
        else if (d < second) second = d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned ok = (unsigned)__shfl_xor((int)key, o);
        const int os = __shfl_xor(second, o);
        const unsigned lo = min(key, ok), hi = max(key, ok);
        const int hv = hi == 0xffffffffu ? 0x7fffffff : (int)(hi >> 16);
        second = min(min(second, os), hv);
        key = lo;
    }
    if (lane != 0) return;
    const int best_d = key == 0xffffffffu ? 0x7fffffff : (int)(key >> 16), best_d2 = second, best_idx = key == 0xffffffffu ? -1 : (int)(key & 0xffffu);
    int match = -1;
    if ((double)best_d < d_mul((double)best_d2, P.min_ratio_12_l)) match = best_idx;
    if (match >= 0 && P.best_lr_matches && m21[(size_t)pair * cap + match] != i1) match = -1;
    const size_t o = (size_t)pair * cap + i1;
    m12[o] = match;
    float ds = -1.f, de = -1.f;
    double l0 = 0, l1 = 0, l2 = 0;
    if (match >= 0) {
        const olf_keyline a = kls[(size_t)(2 * pair) * cap + i1], b = kls[(size_t)(2 * pair + 1) * cap + match];
        const double spl0 = a.startPointX, spl1 = a.startPointY, epl0 = a.endPointX, epl1 = a.endPointY;
        // le_l = sp_l x ep_l, normalised by its first two components
        double c0 = d_sub(spl1, epl1), c1 = d_sub(epl0, spl0), c2 = d_sub(d_mul(spl0, epl1), d_mul(spl1, epl0));
        const double nrm = sqrt(d_add(d_mul(c0, c0), d_mul(c1, c1)));
        if (P.conv_eigen_recip) { const double inv = 1.0 / nrm; c0 = d_mul(c0, inv); c1 = d_mul(c1, inv); c2 = d_mul(c2, inv); }      // Eigen 3.0 / 3.1: v / s = v * (1 / s)
        else { c0 = c0 / nrm; c1 = c1 / nrm; c2 = c2 / nrm; }
        double spr0 = b.startPointX, spr1 = b.startPointY, epr0 = b.endPointX, epr1 = b.endPointY;
        // lineSegmentOverlapStereo(sp_l(1), ep_l(1), sp_r(1), ep_r(1))
        double overlap = 1.0;
        if (fabs(d_sub(epl1, spl1)) > P.line_horiz_th) {
            const double sln = fmin(spl1, epl1), eln = fmax(spl1, epl1), spn = fmin(spr1, epr1), epn = fmax(spr1, epr1);
            const double length = d_sub(eln, spn);
            if ((epn < sln) || (spn > eln)) overlap = 0.0;
            else {
                if ((epn > eln) && (spn < sln)) overlap = d_sub(eln, sln);
                else overlap = d_sub(fmin(eln, epn), fmax(sln, spn));
            }
            if (length > (double)0.01f) overlap = overlap / length;
            else overlap = 0.0;
            if (overlap > 1.0) overlap = 1.0;
        }
        // sp_r is overwritten first; ep_r is then computed from the NEW sp_r (comma initialiser order)
        const double den1 = d_sub(spr1, epr1);
        const double nsx = d_add(d_mul(spr0, d_sub(spl1, epr1)), d_mul(epr0, d_sub(spr1, spl1))) / den1;
        spr0 = nsx; spr1 = spl1;
        const double den2 = d_sub(spr1, epr1);
        const double nex = d_add(d_mul(spr0, d_sub(epl1, epr1)), d_mul(epr0, d_sub(spr1, epl1))) / den2;
        epr0 = nex; epr1 = epl1;
        double disp_s = d_sub(spl0, spr0), disp_e = d_sub(epl0, epr0);
        if (fmin(disp_s, disp_e) / fmax(disp_s, disp_e) < P.ls_min_disp_ratio) { disp_s = -1.0; disp_e = -1.0; }
        if (disp_s >= P.min_disp && disp_e >= P.min_disp && fabs(d_sub(spl1, epl1)) > P.line_horiz_th &&
            fabs(d_sub(spr1, epr1)) > P.line_horiz_th && overlap > P.stereo_overlap_th) {
            ds = (float)disp_s; de = (float)disp_e;
            l0 = c0; l1 = c1; l2 = c2;
        }
    }
    disp[2 * o] = ds; disp[2 * o + 1] = de;
    le[3 * o] = l0; le[3 * o + 1] = l1; le[3 * o + 2] = l2;
}

size_t stereo_lines_prep_bytes(int n_images, int cap) { return (size_t)n_images * cap * sizeof(LinePrep); }

int launch_stereo_lines(int W, int H, const olf_stereo_params& P, int n_pairs, const olf_keyline* d_kls, const uint8_t* d_desc,
                        const int* d_counts, int cap, void* d_prep, uint16_t* d_dist, int* d_m21, int* d_m12, float* d_disp, double* d_le,
                        hipStream_t s)
{
    LinePrep* prep = reinterpret_cast<LinePrep*>(d_prep);
    hipLaunchKernelGGL(k_lines_prep, dim3((cap + 63) / 64, 2 * n_pairs), dim3(64), 0, s, d_kls, d_counts, cap, W, H, prep);
    // OLF_LINES_DIST_W: pairs per call up to which a wave (not a thread) takes a right line
    static const int wMax = [] { const char* e = getenv("OLF_LINES_DIST_W"); return e ? atoi(e) : 256; }();
    if (n_pairs <= wMax)
        hipLaunchKernelGGL(k_lines_dist_w, dim3((cap + 3) / 4, n_pairs), dim3(256), 0, s, prep, d_desc, d_counts, cap, P.matching_s_ws, P.line_sim_th,
                           P.best_lr_matches, d_dist, d_m21);
    else
        hipLaunchKernelGGL(k_lines_dist, dim3((cap + 63) / 64, n_pairs), dim3(64), 0, s, prep, d_desc, d_counts, cap, P.matching_s_ws, P.line_sim_th,
                           P.best_lr_matches, d_dist, d_m21);
    hipLaunchKernelGGL(k_lines_resolve, dim3((cap + 3) / 4, n_pairs), dim3(256), 0, s, d_kls, d_counts, cap, d_dist, d_m21, P, d_m12, d_disp, d_le);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
