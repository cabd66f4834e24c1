// ref_shim.cpp -- C entry points over the REAL reference classes GridStructure / LineIterator
// (src/gridStructure.cpp, src/LineIterator.cpp, compiled from /root/reference where they lie).
// Test infrastructure: validates the oracle's restatement of those two files.
#include "gridStructure.h"
#include "LineIterator.h"
#include <cstring>
using namespace ORB_SLAM2;
extern "C" {
// rasterise (x1,y1)-(x2,y2) with the reference's Bresenham; returns count, writes up to cap (x,y) pairs
int ref_line_coords(double x1, double y1, double x2, double y2, int* xy, int cap)
{
    std::list<std::pair<int, int>> lc;
    getLineCoords(x1, y1, x2, y2, lc);
    int n = 0;
    for (auto& p : lc) { if (n < cap) { xy[2 * n] = p.first; xy[2 * n + 1] = p.second; } ++n; }
    return n;
}
// Fill a rows x cols grid with n segments (each rasterised as above, segment idx pushed to each cell,
// exactly like src/Frame.cc:917-919), then query window [x-wl, x+wr] x [y-hu, y+hd] for (qx,qy) and
// return the candidate set in std::unordered_set iteration order.
int ref_grid_query(int rows, int cols, const double* segs, int n, int qx, int qy, int wl, int wr, int hu, int hd,
                   int qx2, int qy2, int use_second, int* out, int cap)
{
    GridStructure grid(rows, cols);
    std::list<std::pair<int, int>> lc;
    for (int i = 0; i < n; ++i) {
        getLineCoords(segs[4 * i], segs[4 * i + 1], segs[4 * i + 2], segs[4 * i + 3], lc);
        for (auto& p : lc) grid.at(p.first, p.second).push_back(i);
    }
    GridWindow w; w.width = std::make_pair(wl, wr); w.height = std::make_pair(hu, hd);
    std::unordered_set<int> cand;
    grid.get(qx, qy, w, cand);
    if (use_second) grid.get(qx2, qy2, w, cand);
    int k = 0;
    for (int v : cand) { if (k < cap) out[k] = v; ++k; }
    return k;
}
}
