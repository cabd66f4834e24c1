// frame_oracle.cpp -- CPU ORACLE (test infrastructure, NOT product code).
// Restates the feature part of Frame::Frame (stereo + lines), reference src/Frame.cc:136-221:
// ExtractORB x2 (:351-357), ComputeStereoMatches (:702-876).  Line stages are added by
// line_oracle.cpp.  PARITY UNPINNED (see oracle_common.hpp).
#include "orb_oracle.hpp"

namespace orc {
void compute_stereo_matches(const std::vector<olf_keypoint>& keysL, const uint8_t* descL,
                            const std::vector<olf_keypoint>& keysR, const uint8_t* descR,
                            const std::vector<Image>& pyrL, const std::vector<Image>& pyrR,
                            const std::vector<float>& sf, const std::vector<float>& inv_sf, float mbf, float fx,
                            std::vector<float>& uRight, std::vector<float>& depth, std::vector<int>* sad_out);
}
namespace orc {
struct Vec4f;
void line_extract(const Image& img, const olf_line_params& P, bool use_std_sort, std::vector<olf_keyline>& kls, std::vector<uint8_t>& desc,
                  std::vector<olf_keyline>* all_detected);
void stereo_lines(const std::vector<olf_keyline>& klL, const uint8_t* descL, const std::vector<olf_keyline>& klR, const uint8_t* descR,
                  int img_w, int img_h, const olf_stereo_params& P, std::vector<int>& matches_12, std::vector<float>& disp,
                  std::vector<double>& le);
}
#include <malloc.h>
#include <thread>
#include <atomic>
#include <chrono>
thread_local double g_frame_stage_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
thread_local double g_frame_orb_stage_ms[12] = {0};      // orb_extract's six stages (orb_oracle.cpp g_orb_stage_ms), left then right image
namespace orc { extern thread_local double g_orb_stage_ms[6]; }
namespace orc { extern thread_local double g_line_stage_ms[2]; }
using namespace orc;

extern "C" {

// ORB on both images + ComputeStereoMatches.  Outputs: kpsL/descL/nL, kpsR/descR/nR, uRight[cap], depth[cap]
int orc_stereo_points(const uint8_t* imgL, const uint8_t* imgR, int w, int h, const olf_params* p,
                      olf_keypoint* kpsL, uint8_t* descL, int* nL, olf_keypoint* kpsR, uint8_t* descR, int* nR, int cap,
                      float* uRight, float* depth, int* sad)
{
    Image L(w, h), R(w, h);
    std::memcpy(L.d.data(), imgL, (size_t)w * h);
    std::memcpy(R.d.data(), imgR, (size_t)w * h);
    OrbResult rl, rr;
    orb_extract(L, p->orb, rl);
    orb_extract(R, p->orb, rr);
    *nL = (int)rl.kps.size(); *nR = (int)rr.kps.size();
    if (*nL > cap || *nR > cap) return OLF_ERR_CAPACITY;
    std::memcpy(kpsL, rl.kps.data(), rl.kps.size() * sizeof(olf_keypoint));
    std::memcpy(descL, rl.desc.data(), rl.desc.size());
    std::memcpy(kpsR, rr.kps.data(), rr.kps.size() * sizeof(olf_keypoint));
    std::memcpy(descR, rr.desc.data(), rr.desc.size());
    std::vector<float> sf, inv_sf, u, d;
    std::vector<int> s;
    orb_scale_tables(p->orb, sf, inv_sf);
    if (rl.kps.empty()) return OLF_OK;   // Frame ctor returns early (src/Frame.cc:176-177)
    compute_stereo_matches(rl.kps, rl.desc.data(), rr.kps, rr.desc.data(), rl.pyramid, rr.pyramid, sf, inv_sf, p->stereo.bf, p->stereo.fx, u, d, &s);
    for (int i = 0; i < *nL; ++i) { uRight[i] = u[i]; depth[i] = d[i]; if (sad) sad[i] = s[i]; }
    return OLF_OK;
}

// The whole feature part of Frame::Frame (stereo + lines), src/Frame.cc:136-221.  threads = 4 runs the four
// extractions on std::threads exactly like src/Frame.cc:164-171 (the cpu_baseline "reference-shaped" mode);
// threads = 1 runs them back to back.  Any output pointer may be null.
int orc_stereo_frame(const uint8_t* imgL, const uint8_t* imgR, int w, int h, const olf_params* p, int threads,
                     olf_keypoint* kpsL, uint8_t* descL, int* nL, olf_keypoint* kpsR, uint8_t* descR, int* nR, int cap, float* uRight, float* depth,
                     olf_keyline* klL, uint8_t* ldescL, int* nlL, olf_keyline* klR, uint8_t* ldescR, int* nlR, int lcap, int* lm12, float* ldisp,
                     double* lle)
{
    Image L(w, h), R(w, h);
    std::memcpy(L.d.data(), imgL, (size_t)w * h);
    std::memcpy(R.d.data(), imgR, (size_t)w * h);
    OrbResult rl, rr;
    std::vector<olf_keyline> kl, kr;
    std::vector<uint8_t> dl, dr;
    // wall time per stage (ms) of this frame: ORB L, ORB R, LSD L, LBD L, LSD R, LBD R, stereo points, stereo lines (g_frame_stage_ms)
    double* st = g_frame_stage_ms;
    auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    double* ost = g_frame_orb_stage_ms;      // (the extractions may run on their own threads: their thread-local stage times are copied out there)
    auto f0 = [&, ost] { const auto t = std::chrono::steady_clock::now(); orb_extract(L, p->orb, rl); st[0] = ms_since(t); for (int i = 0; i < 6; ++i) ost[i] = g_orb_stage_ms[i]; };
    auto f1 = [&, ost] { const auto t = std::chrono::steady_clock::now(); orb_extract(R, p->orb, rr); st[1] = ms_since(t); for (int i = 0; i < 6; ++i) ost[6 + i] = g_orb_stage_ms[i]; };
    auto f2 = [&] { line_extract(L, p->line, false, kl, dl, nullptr); st[2] = g_line_stage_ms[0]; st[3] = g_line_stage_ms[1]; };
    auto f3 = [&] { line_extract(R, p->line, false, kr, dr, nullptr); st[4] = g_line_stage_ms[0]; st[5] = g_line_stage_ms[1]; };
    if (threads >= 4) {
        std::thread t0(f0), t1(f1), t2(f2), t3(f3);
        t0.join(); t1.join(); t2.join(); t3.join();
    } else { f0(); f1(); f2(); f3(); }
    std::vector<float> sf, inv_sf, u, d;
    orb_scale_tables(p->orb, sf, inv_sf);
    const auto t6 = std::chrono::steady_clock::now();
    if (!rl.kps.empty())
        compute_stereo_matches(rl.kps, rl.desc.data(), rr.kps, rr.desc.data(), rl.pyramid, rr.pyramid, sf, inv_sf, p->stereo.bf, p->stereo.fx, u, d, nullptr);
    st[6] = ms_since(t6);
    std::vector<int> m;
    std::vector<float> dsp;
    std::vector<double> le;
    const auto t7 = std::chrono::steady_clock::now();
    stereo_lines(kl, dl.data(), kr, dr.data(), w, h, p->stereo, m, dsp, le);
    st[7] = ms_since(t7);
    if (nL) *nL = (int)rl.kps.size();
    if (nR) *nR = (int)rr.kps.size();
    if (nlL) *nlL = (int)kl.size();
    if (nlR) *nlR = (int)kr.size();
    if ((int)rl.kps.size() > cap || (int)rr.kps.size() > cap || (int)kl.size() > lcap || (int)kr.size() > lcap) return OLF_ERR_CAPACITY;
    if (kpsL) std::memcpy(kpsL, rl.kps.data(), rl.kps.size() * sizeof(olf_keypoint));
    if (descL) std::memcpy(descL, rl.desc.data(), rl.desc.size());
    if (kpsR) std::memcpy(kpsR, rr.kps.data(), rr.kps.size() * sizeof(olf_keypoint));
    if (descR) std::memcpy(descR, rr.desc.data(), rr.desc.size());
    if (uRight) for (size_t i = 0; i < u.size(); ++i) { uRight[i] = u[i]; depth[i] = d[i]; }
    if (klL) std::memcpy(klL, kl.data(), kl.size() * sizeof(olf_keyline));
    if (ldescL) std::memcpy(ldescL, dl.data(), dl.size());
    if (klR) std::memcpy(klR, kr.data(), kr.size() * sizeof(olf_keyline));
    if (ldescR) std::memcpy(ldescR, dr.data(), dr.size());
    if (lm12) for (size_t i = 0; i < m.size(); ++i) { lm12[i] = m[i]; ldisp[2 * i] = dsp[2 * i]; ldisp[2 * i + 1] = dsp[2 * i + 1]; lle[3 * i] = le[3 * i]; lle[3 * i + 1] = le[3 * i + 1]; lle[3 * i + 2] = le[3 * i + 2]; }
    return OLF_OK;
}

// the stage times of the calling thread's last orc_stereo_frame (8 doubles, see there)
void orc_frame_stage_ms(double* out8) { for (int i = 0; i < 8; ++i) out8[i] = g_frame_stage_ms[i]; }
// orb_extract's stages of the same frame: pyramid, FAST, octree, IC_Angle, blur, rBRIEF -- left image, then right (12 doubles)
void orc_frame_orb_stage_ms(double* out12) { for (int i = 0; i < 12; ++i) out12[i] = g_frame_orb_stage_ms[i]; }

// cpu_baseline "mode B" of bench.py: n_threads workers, each running whole frames (the four extractions back to back) taken from a shared
// counter, over n_frames frames cycling through n_distinct stereo pairs (imgs: [2 * n_distinct][h][w]); returns stereo frames per second
double orc_stereo_frames_throughput(const uint8_t* imgs, int n_distinct, int w, int h, const olf_params* p, int n_threads, int n_frames)
{
    // every frame allocates (and frees) tens of MB of std::vector; with glibc's default thresholds each of them is an mmap / munmap of fresh,
    // zero-filled pages, and hundreds of threads faulting pages of ONE address space serialise in the kernel (round 2 measured 0.19 frames/s
    // per thread against 4 for a single frame on four threads).  Keep freed memory in the per-thread arenas instead: after its first frame a
    // worker reuses its own pages.
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, -1);
    mallopt(M_ARENA_MAX, 2 * n_threads + 8);
    std::atomic<int> next(0), failed(0);
    const size_t npx = (size_t)w * h;
    auto worker = [&] {
        for (;;) {
            const int f = next.fetch_add(1);
            if (f >= n_frames) return;
            const int d = f % n_distinct;
            int a, b, c, e;
            if (orc_stereo_frame(imgs + (size_t)(2 * d) * npx, imgs + (size_t)(2 * d + 1) * npx, w, h, p, 1, nullptr, nullptr, &a, nullptr, nullptr, &b, 1 << 20,
                                 nullptr, nullptr, nullptr, nullptr, &c, nullptr, nullptr, &e, 1 << 20, nullptr, nullptr, nullptr) != OLF_OK) failed.fetch_add(1);
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int i = 0; i < n_threads; ++i) th.emplace_back(worker);
    for (auto& t : th) t.join();
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return failed.load() ? -1.0 : n_frames / sec;
}

}  // extern "C"
