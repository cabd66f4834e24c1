// lsd_sim.cpp -- design tool (not product code, not a test): discrete-event simulation of ORDERED SPECULATIVE region growing.
//
// The sequential seed loop of cv::LineSegmentDetector (oracle/line_oracle.cpp:121-148) is replayed by NW concurrent workers that take
// seeds in rank order, claim pixels in a rank-valued owner map (lower rank wins, a younger region that needs a pixel claimed by an older,
// not yet final region yields and is re-run later, a region that loses a pixel is re-run) and commit in rank order.  The tool checks
// that the committed regions equal the sequential ones and reports the makespan in agent iterations for several NW, i.e. how much
// in-image parallelism the policy exposes on the bench images before any kernel is written.
//
// build: g++ -O2 -std=c++17 -ffp-contract=off tools/lsd_sim.cpp -o /tmp/lsd_sim oracle/liboracle.so orb_line_slam_amd/csrc/libolf_synth.so -Wl,-rpath,$PWD/oracle -Wl,-rpath,$PWD/orb_line_slam_amd/csrc
#include "../oracle/oracle_common.hpp"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <queue>
#include <deque>
#include <algorithm>
#include <cassert>

extern "C" int olf_synth_stereo(uint64_t seed, int W, int H, uint8_t* left, uint8_t* right);
using namespace orc;

static const double kPI = 3.1415926535897932384626433832795;
static const double NOTDEF = -1024.0, M_3_2_PI = (3 * kPI) / 2, M_2__PI = 2 * kPI, DEG_TO_RADS = kPI / 180;

struct Field {
    int W, H;
    std::vector<double> ang;
    std::vector<int> order;     // rank -> address
    std::vector<uint8_t> iso;   // no neighbour aligned with the pixel's own angle: a one-pixel region whenever it seeds
    double prec;
    int minReg;
};

static bool aligned(const Field& F, int addr, double theta)
{
    const double a = F.ang[addr];
    if (a == NOTDEF) return false;
    double n = theta - a;
    if (n < 0) n = -n;
    if (n > M_3_2_PI) { n -= M_2__PI; if (n < 0) n = -n; }
    return n <= F.prec;
}

static Field make_field(const Image& image)
{
    Field F;
    const double SCALE = 1.2, SIGMA = 0.6;
    const int N_BINS = 1024;
    F.prec = kPI * 22.5 / 180;
    const double p = 22.5 / 180, rho = 2.0 / std::sin(F.prec);
    const unsigned hk = (unsigned)std::ceil(SIGMA * std::sqrt(2 * 3 * std::log(10.0)));
    Image g = gaussian_blur_u8(image, gaussian_taps_q8(1 + 2 * hk, SIGMA));
    Image sc = resize_linear_u8(g, cvRound(image.w * SCALE), cvRound(image.h * SCALE), 1. / SCALE, 1. / SCALE);
    const int W = F.W = sc.w, H = F.H = sc.h;
    F.ang.assign((size_t)W * H, NOTDEF);
    std::vector<double> mod((size_t)W * H, 0.0);
    double maxg = -1;
    for (int y = 0; y < H - 1; ++y)
        for (int x = 0; x < W - 1; ++x) {
            const int DA = sc.at(x + 1, y + 1) - sc.at(x, y), BC = sc.at(x + 1, y) - sc.at(x, y + 1);
            const int gx = DA + BC, gy = DA - BC;
            const double norm = std::sqrt((gx * gx + gy * gy) / 4.0);
            mod[(size_t)y * W + x] = norm;
            if (norm > rho) { F.ang[(size_t)y * W + x] = fastAtan2(float(gx), float(-gy)) * DEG_TO_RADS; if (norm > maxg) maxg = norm; }
        }
    const double bc = maxg > 0 ? double(N_BINS - 1) / maxg : 0;
    std::vector<std::vector<int>> bins(N_BINS);
    for (int y = 0; y < H - 1; ++y)
        for (int x = 0; x < W - 1; ++x) {
            const int a = y * W + x;
            if (F.ang[a] == NOTDEF) continue;       // undefined pixels never seed
            bins[int(mod[a] * bc)].push_back(a);
        }
    for (int b = N_BINS - 1; b >= 0; --b) for (int a : bins[b]) F.order.push_back(a);
    F.iso.assign((size_t)W * H, 0);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const int a = y * W + x;
            if (F.ang[a] == NOTDEF) continue;
            bool iso = true;
            for (int yy = std::max(y - 1, 0); yy <= std::min(y + 1, H - 1); ++yy)
                for (int xx = std::max(x - 1, 0); xx <= std::min(x + 1, W - 1); ++xx)
                    if ((yy != y || xx != x) && aligned(F, yy * W + xx, F.ang[a])) iso = false;
            F.iso[a] = iso;
        }
    const double LOG_NT = 5 * (std::log10(double(W)) + std::log10(double(H))) / 2 + std::log10(11.0);
    F.minReg = int(-LOG_NT / std::log10(p));
    return F;
}

struct Region { int rank; std::vector<int> px; int iters; };

// sequential reference: regions in seed order, with the agent's iteration count (up to 7 FIFO entries per iteration)
static std::vector<Region> run_seq(const Field& F, long long* total_iters)
{
    const int W = F.W, H = F.H;
    std::vector<uint8_t> used((size_t)W * H, 0);
    std::vector<Region> out;
    long long ti = 0;
    for (size_t r = 0; r < F.order.size(); ++r) {
        const int a0 = F.order[r];
        if (used[a0]) continue;
        Region R; R.rank = (int)r; R.px.push_back(a0);
        double reg_angle = F.ang[a0];
        float sdx = float(std::cos(reg_angle)), sdy = float(std::sin(reg_angle));
        used[a0] = 1;
        for (size_t i = 0; i < R.px.size(); ++i) {
            const int rx = R.px[i] % W, ry = R.px[i] / W;
            for (int yy = std::max(ry - 1, 0); yy <= std::min(ry + 1, H - 1); ++yy)
                for (int xx = std::max(rx - 1, 0); xx <= std::min(rx + 1, W - 1); ++xx) {
                    const int c = yy * W + xx;
                    if (!used[c] && aligned(F, c, reg_angle)) {
                        used[c] = 1; R.px.push_back(c);
                        sdx += std::cos((double)float(F.ang[c])); sdy += std::sin((double)float(F.ang[c]));
                        reg_angle = fastAtan2(sdy, sdx) * DEG_TO_RADS;
                    }
                }
        }
        R.iters = R.px.size() == 1 ? 0 : (int)((R.px.size() + 6) / 7);   // lower bound; the real count depends on FIFO depth per step
        ti += R.iters;
        out.push_back(std::move(R));
    }
    *total_iters = ti;
    return out;
}

// ---- ordered speculation -------------------------------------------------------------------------
enum St { GROWING, DONE, PARKED };
struct Entry {
    St st; std::vector<int> px; size_t i = 0; double reg_angle = 0; float sdx = 0, sdy = 0; int blocker = -1; bool invalid = false; int worker = -1;
    int runs = 0;
    int delay = 0;      // per-run overhead still to be served (dispatch / pick / prologue / finish of the kernel, in iterations)
};

struct Sim {
    const Field& F;
    int NW, K;                      // workers, FIFO entries per iteration
    std::vector<int> owner;         // rank or -1
    std::map<int, Entry> rob;       // unresolved live seeds
    int next = 0;                   // next rank to dispatch
    int watermark = 0;
    std::vector<int> wrank;         // worker -> rank or -1
    long long ticks = 0, work = 0, wasted = 0, aborts = 0, steals = 0, parks = 0;
    std::vector<Region> committed;
    int robCap;
    bool alignedOnly;
    int OV = 0;                     // iterations of overhead charged to every (re-)run of a region
    int OVp = 0, B = 1;             // ... and to every pick; a pick claims up to B runnable seeds, which the worker then grows one after the other
    std::vector<std::deque<int>> wq;    // worker -> claimed ranks (front = the one being grown)
    Sim(const Field& f, int nw, int k, int cap, bool ao) : F(f), NW(nw), K(k), owner((size_t)f.W * f.H, -1), wrank(nw, -1), robCap(cap), alignedOnly(ao), wq(nw) {}

    bool is_final(int o) const { return o >= 0 && o < watermark; }

    void release(Entry& e, int rank) { for (int p : e.px) if (owner[p] == rank) owner[p] = -1; }

    void start(int rank, Entry& e, int w)
    {
        const int a0 = F.order[rank];
        e.st = GROWING; e.px.clear(); e.px.push_back(a0); e.i = 0; e.invalid = false; e.worker = w; ++e.runs; e.delay = OV + (wq[w].empty() ? OVp : 0);
        e.reg_angle = F.ang[a0]; e.sdx = float(std::cos(e.reg_angle)); e.sdy = float(std::sin(e.reg_angle));
        owner[a0] = rank;
        wq[w].push_back(rank); wrank[w] = wq[w].front();
    }
    void drop(int w, int rank)
    {
        auto& q = wq[w];
        q.erase(std::remove(q.begin(), q.end(), rank), q.end());
        wrank[w] = q.empty() ? -1 : q.front();
    }
    void park(int rank, Entry& e, int blocker)
    {
        release(e, rank); wasted += e.px.size(); e.px.clear();
        e.st = PARKED; e.blocker = blocker; e.invalid = false;
        if (e.worker >= 0) { drop(e.worker, rank); e.worker = -1; }
        ++parks;
    }
    // one agent iteration of the region of `rank`; returns false when the region is complete
    void step(int rank, Entry& e)
    {
        const int W = F.W, H = F.H;
        ++work;
        if (e.delay > 0) { --e.delay; return; }
        const size_t end = std::min(e.px.size(), e.i + (size_t)K);
        for (; e.i < end; ++e.i) {
            const int rx = e.px[e.i] % W, ry = e.px[e.i] / W;
            for (int yy = std::max(ry - 1, 0); yy <= std::min(ry + 1, H - 1); ++yy)
                for (int xx = std::max(rx - 1, 0); xx <= std::min(rx + 1, W - 1); ++xx) {
                    const int c = yy * W + xx;
                    if (F.ang[c] == NOTDEF) continue;
                    const int o = owner[c];
                    if (o == rank) continue;
                    if (o >= 0 && o < rank) {
                        if (is_final(o)) continue;                 // used by a final region
                        if (!alignedOnly || aligned(F, c, e.reg_angle)) { ++aborts; park(rank, e, o); return; }   // younger yields
                        continue;
                    }
                    if (!aligned(F, c, e.reg_angle)) continue;
                    if (o > rank) {                                 // steal from a younger region
                        ++steals;
                        auto it = rob.find(o);
                        assert(it != rob.end());
                        Entry& v = it->second;
                        if (v.st == GROWING) v.invalid = true; else if (v.st == DONE) { park(o, v, rank); }
                    }
                    owner[c] = rank; e.px.push_back(c);
                    e.sdx += std::cos((double)float(F.ang[c])); e.sdy += std::sin((double)float(F.ang[c]));
                    e.reg_angle = fastAtan2(e.sdy, e.sdx) * DEG_TO_RADS;
                }
        }
        if (e.i >= e.px.size()) { e.st = DONE; drop(e.worker, rank); e.worker = -1; }
    }

    void advance_watermark()
    {
        for (;;) {
            if (rob.empty()) { watermark = next; return; }
            auto it = rob.begin();
            watermark = it->first;
            Entry& e = it->second;
            if (e.st != DONE) return;
            Region R; R.rank = it->first; R.px = std::move(e.px); R.iters = 0;
            // committed: every rank below watermark+1 is final
            committed.push_back(std::move(R));
            rob.erase(it);
        }
    }

    void run()
    {
        const int nk = (int)F.order.size();
        for (;;) {
            advance_watermark();
            // hand work to idle workers: first re-runnable parked seeds (lowest rank first), then new seeds
            for (int w = 0; w < NW; ++w) {
                if (wrank[w] >= 0) continue;
                for (int b = 0; b < B; ++b) {
                bool got = false;
                for (auto& kv : rob) {
                    Entry& e = kv.second;
                    if (e.st != PARKED) continue;
                    const int rank = kv.first;
                    const int a0 = F.order[rank];
                    const int o = owner[a0];
                    if (o >= 0 && o < rank) {
                        if (is_final(o)) { e.st = DONE; e.px.clear(); e.px.push_back(-1); continue; }   // consumed: dead (resolved at commit)
                        continue;                                        // still claimed by an unfinished older region
                    }
                    // re-run only when the blocker is resolved (final or dead) -- or when this is the head
                    auto bi = rob.find(e.blocker);
                    if (bi != rob.end() && rank != rob.begin()->first) continue;
                    if (o > rank) { auto it = rob.find(o); Entry& v = it->second; ++steals; if (v.st == GROWING) v.invalid = true; else if (v.st == DONE) park(o, v, rank); }
                    if (F.iso[a0]) { e.st = DONE; e.px.assign(1, a0); owner[a0] = rank; continue; }
                    start(rank, e, w); got = true; break;
                }
                if (got) continue;      // (next seed of this worker's batch)
                while (next < nk && (int)rob.size() < robCap) {
                    const int rank = next++;
                    const int a0 = F.order[rank];
                    const int o = owner[a0];
                    if (o >= 0) {
                        if (is_final(o)) continue;                       // dead seed
                        Entry& e = rob[rank]; e.st = PARKED; e.blocker = o; ++parks; continue;   // claimed by an unfinished older region
                    }
                    Entry& e = rob[rank];
                    if (F.iso[a0]) { e.st = DONE; e.px.assign(1, a0); owner[a0] = rank; continue; }
                    start(rank, e, w); got = true; break;
                }
                if (!got) break;
                }
            }
            bool any = false;
            for (int w = 0; w < NW; ++w) {
                const int rank = wrank[w];
                if (rank < 0) continue;
                any = true;
                Entry& e = rob[rank];
                if (e.invalid) { park(rank, e, e.blocker); continue; }
                step(rank, e);
            }
            if (!any && next >= nk && rob.empty()) break;
            if (!any) {
                // nothing runnable: only happens transiently (parked entries waiting for the head to commit)
                bool prog = false;
                for (auto& kv : rob) if (kv.second.st == DONE) { prog = true; break; }
                if (!prog && next >= nk) { fprintf(stderr, "deadlock\n"); exit(1); }
            }
            ++ticks;
        }
    }
};

// ---- the same policy on G workgroups --------------------------------------------------------------------------------------------------------
// Seeds are dealt to the groups by rank-interleaved windows of WS seeds; every group has its OWN reorder buffer (robCap entries, what fits a workgroup's LDS),
// its own workers and a local watermark (rank of its oldest unresolved seed).  The global watermark is the minimum of the local ones, and a group sees the
// other groups' watermarks D ticks late (they travel through memory).  A group's head entry commits when it is older than everything the group can see
// unresolved elsewhere.  Pixel claims (owner words) and steal notices are immediate (L2 atomics).  Same check as above: committed regions == sequential ones.
struct Sim2 {
    const Field& F;
    int NW, K, G, WS, D, robCap, OV;
    std::vector<int> owner;
    std::vector<std::map<int, Entry>> rob;      // per group
    std::vector<int> gnext;                     // per group: next rank of its windows to dispatch
    std::vector<std::deque<int>> hist;          // per group: published local watermarks, newest at the back
    std::vector<int> wrank;
    long long ticks = 0, work = 0, wasted = 0, aborts = 0, steals = 0, parks = 0, idle = 0;
    std::vector<Region> committed;
    int nk;
    Sim2(const Field& f, int nw, int k, int g, int ws, int d, int cap, int ov)
        : F(f), NW(nw), K(k), G(g), WS(ws), D(d), robCap(cap), OV(ov), owner((size_t)f.W * f.H, -1), rob(g), gnext(g), hist(g), wrank(nw, -1), nk((int)f.order.size())
    {
        for (int q = 0; q < G; ++q) gnext[q] = std::min(q * WS, nk);
    }
    int grp(int rank) const { return (rank / WS) % G; }
    int wgrp(int w) const { return w % G; }
    int advance(int r) const { ++r; if (r % WS == 0) r += (G - 1) * WS; return std::min(r, nk); }
    int local_wm(int g) const { return rob[g].empty() ? gnext[g] : rob[g].begin()->first; }
    int seen_wm(int g, int h) const                 // group h's watermark as group g sees it
    {
        if (h == g || D == 0 || hist[h].empty()) return local_wm(h);
        const int n = (int)hist[h].size();
        return hist[h][std::max(0, n - 1 - D)];
    }
    int global_seen(int g) const { int m = nk; for (int h = 0; h < G; ++h) m = std::min(m, seen_wm(g, h)); return m; }
    int others_seen(int g) const { int m = nk; for (int h = 0; h < G; ++h) if (h != g) m = std::min(m, seen_wm(g, h)); return m; }
    Entry* find(int rank) { auto& r = rob[grp(rank)]; auto it = r.find(rank); return it == r.end() ? nullptr : &it->second; }

    void release(Entry& e, int rank) { for (int p : e.px) if (owner[p] == rank) owner[p] = -1; }
    void start(int rank, Entry& e, int w)
    {
        const int a0 = F.order[rank];
        e.st = GROWING; e.px.clear(); e.px.push_back(a0); e.i = 0; e.invalid = false; e.worker = w; ++e.runs; e.delay = OV;
        e.reg_angle = F.ang[a0]; e.sdx = float(std::cos(e.reg_angle)); e.sdy = float(std::sin(e.reg_angle));
        owner[a0] = rank; wrank[w] = rank;
    }
    void park(int rank, Entry& e, int blocker)
    {
        release(e, rank); wasted += e.px.size(); e.px.clear();
        e.st = PARKED; e.blocker = blocker; e.invalid = false;
        if (e.worker >= 0) { wrank[e.worker] = -1; e.worker = -1; }
        ++parks;
    }
    struct Notice { long long due; int victim, thief; };
    std::vector<Notice> notices;                // steal notices on their way to another group (that group's entries live in ITS LDS: the notice goes through memory)
    void deliver(int victim, int thief)
    {
        Entry* v = find(victim);
        if (!v) { fprintf(stderr, "notice for a committed region\n"); exit(1); }      // (cannot happen: the thief is older and its group's watermark travels behind the notice)
        if (v->st == GROWING) v->invalid = true; else if (v->st == DONE) park(victim, *v, thief);
    }
    void take_from(int victim, int thief)
    {
        ++steals;
        if (D > 0 && grp(victim) != grp(thief)) { notices.push_back({ticks + D, victim, thief}); return; }
        deliver(victim, thief);
    }
    void step(int rank, Entry& e)
    {
        const int W = F.W, H = F.H, g = grp(rank);
        ++work;
        if (e.delay > 0) { --e.delay; return; }
        const int wm = global_seen(g);
        const size_t end = std::min(e.px.size(), e.i + (size_t)K);
        for (; e.i < end; ++e.i) {
            const int rx = e.px[e.i] % W, ry = e.px[e.i] / W;
            for (int yy = std::max(ry - 1, 0); yy <= std::min(ry + 1, H - 1); ++yy)
                for (int xx = std::max(rx - 1, 0); xx <= std::min(rx + 1, W - 1); ++xx) {
                    const int c = yy * W + xx;
                    if (F.ang[c] == NOTDEF) continue;
                    const int o = owner[c];
                    if (o == rank) continue;
                    if (o >= 0 && o < rank) {
                        if (o < wm) continue;                                        // used by a region this group knows to be final
                        if (aligned(F, c, e.reg_angle)) { ++aborts; park(rank, e, o); return; }
                        continue;
                    }
                    if (!aligned(F, c, e.reg_angle)) continue;
                    if (o > rank) take_from(o, rank);
                    owner[c] = rank; e.px.push_back(c);
                    e.sdx += std::cos((double)float(F.ang[c])); e.sdy += std::sin((double)float(F.ang[c]));
                    e.reg_angle = fastAtan2(e.sdy, e.sdx) * DEG_TO_RADS;
                }
        }
        if (e.i >= e.px.size()) { e.st = DONE; wrank[e.worker] = -1; e.worker = -1; }
    }
    void commit(int g)
    {
        const int lim = others_seen(g);
        while (!rob[g].empty()) {
            auto it = rob[g].begin();
            if (it->first >= lim || it->second.st != DONE) return;
            Region R; R.rank = it->first; R.px = std::move(it->second.px); R.iters = 0;
            committed.push_back(std::move(R));
            rob[g].erase(it);
        }
    }
    // a runnable seed for worker w of group g: parked entries of the group whose blocker is resolved (or that are the globally oldest), then new seeds
    bool hand(int w)
    {
        const int g = wgrp(w);
        const int wm = global_seen(g);
        for (auto& kv : rob[g]) {
            Entry& e = kv.second;
            if (e.st != PARKED) continue;
            const int rank = kv.first, a0 = F.order[rank], o = owner[a0];
            if (o >= 0 && o < rank) {
                if (o < wm) { e.st = DONE; e.px.clear(); e.px.push_back(-1); }       // consumed by a final region: dead
                continue;
            }
            if (!(e.blocker < wm || rank == wm)) continue;
            if (o > rank) take_from(o, rank);
            if (F.iso[a0]) { e.st = DONE; e.px.assign(1, a0); owner[a0] = rank; continue; }
            start(rank, e, w);
            return true;
        }
        while (gnext[g] < nk && (int)rob[g].size() < robCap) {
            const int rank = gnext[g]; gnext[g] = advance(rank);
            const int a0 = F.order[rank], o = owner[a0];
            if (o >= 0 && o < rank) {
                if (o < wm) continue;                                                // dead seed
                Entry& e = rob[g][rank]; e.st = PARKED; e.blocker = o; ++parks; continue;
            }
            Entry& e = rob[g][rank];
            if (o > rank) take_from(o, rank);                                        // a younger region of another group's later window was quicker
            if (F.iso[a0]) { e.st = DONE; e.px.assign(1, a0); owner[a0] = rank; continue; }
            start(rank, e, w);
            return true;
        }
        return false;
    }
    void run()
    {
        for (;;) {
            for (size_t q = 0; q < notices.size();) {
                if (notices[q].due <= ticks) { deliver(notices[q].victim, notices[q].thief); notices[q] = notices.back(); notices.pop_back(); } else ++q;
            }
            for (int g = 0; g < G; ++g) commit(g);
            for (int g = 0; g < G; ++g) { hist[g].push_back(local_wm(g)); if ((int)hist[g].size() > D + 2) hist[g].pop_front(); }
            for (int w = 0; w < NW; ++w) if (wrank[w] < 0) hand(w);
            bool any = false;
            for (int w = 0; w < NW; ++w) {
                const int rank = wrank[w];
                if (rank < 0) { ++idle; continue; }
                any = true;
                Entry& e = *find(rank);
                if (e.invalid) { park(rank, e, e.blocker); continue; }
                step(rank, e);
            }
            bool left = false;
            for (int g = 0; g < G; ++g) if (gnext[g] < nk || !rob[g].empty()) left = true;
            if (!any && !left) break;
            if (++ticks > 4000000) { fprintf(stderr, "no progress\n"); exit(1); }
        }
    }
};

int main(int argc, char** argv)
{
    const int W = argc > 1 ? atoi(argv[1]) : 1242, H = argc > 2 ? atoi(argv[2]) : 375;
    const uint64_t seed = argc > 3 ? atoll(argv[3]) : 1000;
    Image L(W, H), R(W, H);
    olf_synth_stereo(seed, W, H, L.d.data(), R.d.data());
    Field F = make_field(L);
    long long seqIters = 0;
    std::vector<Region> seq = run_seq(F, &seqIters);
    size_t big = 0, grow = 0, tot = 0, maxn = 0;
    std::map<int, int> hist;
    for (auto& r : seq) { if ((int)r.px.size() >= F.minReg) ++big; if (r.px.size() > 1) { ++grow; tot += r.px.size(); } maxn = std::max(maxn, r.px.size());
        int b = 0; size_t s = r.px.size(); while (s > 1) { s >>= 1; ++b; } hist[b]++; }
    printf("image %dx%d seed %llu: scaled %dx%d, keys %zu, regions %zu (grown %zu, >=min %zu, max %zu), grown pixels %zu, seq iterations(K=7) %lld\n", W, H,
           (unsigned long long)seed, F.W, F.H, F.order.size(), seq.size(), grow, big, maxn, tot, seqIters);
    printf("size histogram (log2 bucket: count):"); for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second); printf("\n");
    long long pxw = 0; for (auto& r : seq) if (r.px.size() > 1) pxw += r.px.size();
    const int capArg = argc > 4 ? atoi(argv[4]) : 256;
    const int Karg = argc > 5 ? atoi(argv[5]) : 7;
    const int nwArg = argc > 6 ? atoi(argv[6]) : 0;
    const int ovArg = argc > 7 ? atoi(argv[7]) : 0;
    const int ovpArg = argc > 8 ? atoi(argv[8]) : 0, bArg = argc > 9 ? atoi(argv[9]) : 1;
    const int gArg = argc > 10 ? atoi(argv[10]) : 0, dArg = argc > 11 ? atoi(argv[11]) : 1, wsArg = argc > 12 ? atoi(argv[12]) : 64;
    if (gArg > 0) {
        // lsd_sim W H seed robCapPerGroup K NW OV - - G D WS
        Sim2 S(F, nwArg ? nwArg : 32, Karg, gArg, wsArg, dArg, capArg, ovArg);
        S.run();
        std::sort(S.committed.begin(), S.committed.end(), [](const Region& a, const Region& b) { return a.rank < b.rank; });
        bool ok = true;
        size_t ci = 0;
        for (auto& r : seq) {
            while (ci < S.committed.size() && S.committed[ci].px.size() == 1 && S.committed[ci].px[0] == -1) ++ci;
            if (ci >= S.committed.size() || S.committed[ci].rank != r.rank || S.committed[ci].px != r.px) { ok = false; break; }
            ++ci;
        }
        printf("G=%d groups x %d workers, %d entries per group, windows of %d seeds, watermark delay %d, OV=%d: ticks %8lld  work %8lld  idle %8lld aborts %6lld steals %6lld parks %6lld  %s\n",
               gArg, S.NW / gArg, capArg, wsArg, dArg, ovArg, S.ticks, S.work, S.idle, S.aborts, S.steals, S.parks, ok ? "EXACT" : "MISMATCH");
        return 0;
    }
    for (int K : {Karg}) {
        for (int nw : (nwArg ? std::vector<int>{nwArg} : std::vector<int>{1, 4, 16, 64})) {
            for (int cap : {capArg}) {
                Sim S(F, nw, K, cap, true);
                S.OV = ovArg; S.OVp = ovpArg; S.B = bArg;
                S.run();
                bool ok = true;
                size_t ci = 0;
                for (auto& r : seq) {
                    while (ci < S.committed.size() && S.committed[ci].px.size() == 1 && S.committed[ci].px[0] == -1) ++ci;
                    if (ci >= S.committed.size() || S.committed[ci].rank != r.rank || S.committed[ci].px != r.px) { ok = false; break; }
                    ++ci;
                }
                printf("K=%d NW=%3d robCap=%d: ticks %8lld  work %8lld  aborts %6lld steals %6lld parks %6lld wastedpx %7lld  %s\n", K, nw, cap, S.ticks,
                       S.work, S.aborts, S.steals, S.parks, S.wasted, ok ? "EXACT" : "MISMATCH");
            }
        }
    }
    return 0;
}
