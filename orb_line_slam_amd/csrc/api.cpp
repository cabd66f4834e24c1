// api.cpp -- the extern "C" boundary of liborbline_hip.so (include/orbline.h).
// Owns the device context; every compute entry point launches HIP kernels -- there is no CPU
// fallback: without a gfx950 device olf_ctx_create fails with OLF_ERR_NODEVICE.
#include "olf_internal.hpp"
#include "line_internal.hpp"
#include "../../include/orbline.h"
#include <map>
#include <mutex>
#include <cstring>
#include <mutex>

namespace olf {
static thread_local std::string g_err;
void set_error(const std::string& s) { g_err = s; }
}  // namespace olf

using namespace olf;

struct olf_ctx {
    olf_params params;
    int W = 0, H = 0, max_images = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    OrbHostTables orb;
    OrbDeviceBufs ob;
    // staging for the host-pointer entry points
    uint8_t* d_images = nullptr;
    olf_keypoint* d_kps = nullptr;
    uint8_t* d_desc = nullptr;
    int* d_counts = nullptr;
    int last_n_images = 0;
    float* d_uright = nullptr;     // [max_pairs][outCap]
    float* d_depth = nullptr;
    int* d_sad = nullptr;
    int* d_bestkey = nullptr;
    unsigned* d_rowperm = nullptr; // [max_images][outCap] key point lists ordered by row (stereo candidate search)
    // line side
    LineHostTables line;
    LineDeviceBufs lb;
    olf_keyline* d_kls = nullptr;      // [max_images][lineCap]   (host-pointer staging / fused path)
    uint8_t* d_ldesc = nullptr;
    int* d_lcounts = nullptr;
    void* d_lprep = nullptr;
    uint16_t* d_ldist = nullptr;
    int* d_lm21 = nullptr;
    int* d_lm12 = nullptr;
    float* d_ldisp = nullptr;
    double* d_lle = nullptr;
    hipStream_t stream2 = nullptr;
    bool has_tables = false;       // holds a reference on the device's shared LSD angle tables
    bool mark_front = false;
    int orb_wait_after = 0;        // fused entry: the ORB stream waits for the LSD front behind this many of its own dense stages (pyramid, blur, FAST)
    hipEvent_t ev_front = nullptr;
    hipEvent_t input_event = nullptr;  // olf_ctx_set_input_event: not owned; the fused entry's line stream waits for it instead of forking from the caller's stream
    hipEvent_t ev_lbd = nullptr;       // fused entry: the LBD gradient images are ready (computed on the ORB stream in the seed ordering's shadow)
    bool deferred_join = false;        // olf_ctx_set_deferred_join: olf_stereo_frames_dev returns with the line path still running on the line stream
    bool scaled_aliased = false;       // batch context: the LSD working images live in the key buffers (dead before those are written)
    bool join_pending = false;         // ... and this call's line path has not been joined yet (olf_stereo_frames_join_dev, or the next call)
    // small contexts (the drop-in's one-pair-per-call shape): the eleven output arrays of olf_stereo_frames sit in ONE device slab, so that the host entry brings them
    // back with one copy into pinned memory instead of eleven (each a launch and a gap of its own: 0.44 ms of a 9 ms call, profiles/r5b_pair_timeline.txt)
    uint8_t* d_outslab = nullptr; uint8_t* h_outslab = nullptr; size_t outslab_bytes = 0; size_t outslab_off[12] = {0};
    // every line-side output that call is still writing (key lines, LBD descriptors, counts, line matches, disparities, line equations): an entry that is handed
    // one of them (matcher, line stereo, packer) joins first.  The obligation is discharged for the whole context only when the frame call's OWN stream has
    // waited (pend_stream); a join on another stream orders that stream and leaves the obligation standing
    struct { const uint8_t* p; size_t bytes; } pend[6] = {};
    hipStream_t pend_stream = nullptr;
    bool defer_lbd = false;            // fused entry: olf_line_extract_dev stops behind the rectangles; selection + LBD are enqueued by the caller
    bool lbd_pre = false;              // fused entry: olf_orb_extract_dev computes the LBD gradient images behind its blur and records ev_lbd
    hipEvent_t ev_sort = nullptr;      // recorded in front of the seed ordering (the dense, bandwidth-bound half of the LSD front is through)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // stage profiling (olf_profile_*): HIP events recorded on the stream each stage is launched on
    bool prof_on = false;
    struct ProfRec { int stage; hipEvent_t a, b; };
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[16] = {0};
    int prof_calls[16] = {0};
    // grow-on-demand scratch (matcher k-NN tables, host-pointer staging)
    void* scratch[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t scratch_bytes[4] = {0, 0, 0, 0};
    std::vector<void*> allocs;
};

// A context (and everything it owns) lives on the device that was current when it was created; calling into it with another device
// current would send its launches and allocations to the wrong GPU.
static int check_device(const olf_ctx* c, const char* who)
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess || d != c->device) { set_error(std::string(who) + ": the context belongs to another device than the current one"); return OLF_ERR_INVALID; }
    return OLF_OK;
}

static int scratch_get(olf_ctx* c, int slot, size_t bytes, void** out)
{
    if (c->scratch_bytes[slot] < bytes) {
        if (c->scratch[slot]) { OLF_HIP_CHECK(hipDeviceSynchronize()); (void)hipFree(c->scratch[slot]); c->scratch[slot] = nullptr; c->scratch_bytes[slot] = 0; }
        size_t want = std::max<size_t>(bytes * 3 / 2, 4096);
        OLF_HIP_CHECK(hipMalloc(&c->scratch[slot], want));
        c->scratch_bytes[slot] = want;
    }
    *out = c->scratch[slot];
    return OLF_OK;
}

// The LSD angle / cos-sin tables depend on nothing but the packed gradient pair: one copy per device, shared by every context of the process
// (a reference Frame is served by four extractor objects, each with a context of its own; include/orbline_adaptor.hpp).
namespace {
struct AngleTables { float* angDeg = nullptr; void* angEnt = nullptr; int refs = 0; };
std::mutex g_tab_mu;
std::map<int, AngleTables> g_tabs;

// (one table per device AND per convention C.6 variant: the entries' cos / sin differ between them)
int angle_tables_acquire(int device, int libmFloat, LineDeviceBufs& l, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_tab_mu);
    AngleTables& t = g_tabs[device * 2 + (libmFloat ? 1 : 0)];
    if (t.refs == 0) {
        if (hipMalloc(&t.angDeg, sizeof(float) << 22) != hipSuccess || hipMalloc(&t.angEnt, (size_t)32 << 22) != hipSuccess) {
            (void)hipFree(t.angDeg); (void)hipFree(t.angEnt);
            t = AngleTables();
            set_error("hipMalloc failed (LSD angle tables)");
            return OLF_ERR_HIP;
        }
        l.angDeg = t.angDeg; l.angEnt = t.angEnt;
        if (launch_lsd_angle_table(l, libmFloat, s) != OLF_OK || hipStreamSynchronize(s) != hipSuccess) {
            (void)hipFree(t.angDeg); (void)hipFree(t.angEnt);
            t = AngleTables();
            set_error("LSD angle table build failed");
            return OLF_ERR_HIP;
        }
    }
    ++t.refs;
    l.angDeg = t.angDeg; l.angEnt = t.angEnt;
    return OLF_OK;
}

void angle_tables_release(int device, int libmFloat)
{
    std::lock_guard<std::mutex> lk(g_tab_mu);
    auto it = g_tabs.find(device * 2 + (libmFloat ? 1 : 0));
    if (it == g_tabs.end() || it->second.refs <= 0) return;
    if (--it->second.refs == 0) {
        (void)hipFree(it->second.angDeg); (void)hipFree(it->second.angEnt);
        g_tabs.erase(it);
    }
}
}  // namespace

namespace olf {
int launch_pack_records(const olf_frame_buffers& fb, int n_pairs, int cap, int lcap, uint8_t* d_dst, size_t dst_capacity, int* d_rowOfs,
                        unsigned long long* d_bytes, int* d_status, hipStream_t s);
size_t pack_records_bound(int n_pairs, int cap, int lcap);
int launch_copy16(const void* src, void* dst, size_t bytes, hipStream_t s);
int launch_depth_mask(const float* depth, uint8_t* mask, size_t n, hipStream_t s);
}

namespace olf {
hipStream_t ctx_stream(olf_ctx* c) { return c->stream; }
int ctx_scratch(olf_ctx* c, int slot, size_t bytes, void** out) { return scratch_get(c, slot, bytes, out); }
}

enum { ST_ORB_PYRAMID, ST_ORB_FAST, ST_ORB_OCTREE, ST_ORB_BLUR, ST_ORB_DESCRIBE, ST_STEREO_POINTS, ST_LSD_FRONT, ST_LSD_GROW, ST_LSD_RECT, ST_LINE_LBD,
       ST_STEREO_LINES, ST_MATCH_BF, ST_COUNT };
static const char* kStageNames[ST_COUNT] = {"orb_pyramid", "orb_fast_cells", "orb_octree", "orb_blur", "orb_describe", "stereo_points",
                                            "lsd_front", "lsd_grow", "lsd_rect", "line_select_lbd", "stereo_lines", "match_bf"};

static hipEvent_t prof_event(olf_ctx* c)
{
    if (!c->prof_pool.empty()) { hipEvent_t e = c->prof_pool.back(); c->prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
struct StageScope {
    olf_ctx* c; hipStream_t s; int stage; hipEvent_t a = nullptr;
    StageScope(olf_ctx* c_, hipStream_t s_, int st) : c(c_), s(s_), stage(st) { if (c->prof_on) { a = prof_event(c); (void)hipEventRecord(a, s); } }
    ~StageScope() { if (a) { hipEvent_t b = prof_event(c); (void)hipEventRecord(b, s); c->prof_recs.push_back({stage, a, b}); } }
};

template <typename T>
static int dev_alloc(olf_ctx* c, T** p, size_t n)
{
    void* q = nullptr;
    OLF_HIP_CHECK(hipMalloc(&q, std::max<size_t>(n * sizeof(T), 256)));
    c->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return OLF_OK;
}

#define OLF_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != OLF_OK) return _rc; \
    } while (0)

extern "C" {

const char* olf_last_error(void) { return g_err.c_str(); }

int olf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int olf_default_params(olf_params* p)
{
    if (!p) return OLF_ERR_INVALID;
    std::memset(p, 0, sizeof(*p));
    p->abi_version = OLF_ABI_VERSION; p->struct_size = (uint32_t)sizeof(olf_params);
    p->orb.nfeatures = 2000; p->orb.scale_factor = 1.2f; p->orb.nlevels = 8; p->orb.ini_th_fast = 20; p->orb.min_th_fast = 7;
    p->line.lsd_nfeatures = 500; p->line.min_line_length = 0.025; p->line.lsd_refine = 0; p->line.lsd_scale = 1.2;
    p->line.lsd_sigma_scale = 0.6; p->line.lsd_quant = 2.0; p->line.lsd_ang_th = 22.5; p->line.lsd_log_eps = 1.0;
    p->line.lsd_density_th = 0.6; p->line.lsd_n_bins = 1024;
    p->line.conv_seed_order = 1;      // OpenCV >= 3.3 (the reference asks for 3.4 first, CMakeLists.txt:37-43): std::sort seed order
    p->stereo.fx = 718.856f; p->stereo.bf = 386.1448f; p->stereo.matching_s_ws = 10; p->stereo.line_sim_th = 0.75;
    p->stereo.min_ratio_12_l = 0.9; p->stereo.min_disp = 1.0; p->stereo.line_horiz_th = 0.1; p->stereo.stereo_overlap_th = 0.75;
    p->stereo.ls_min_disp_ratio = 0.7; p->stereo.best_lr_matches = 1;
    return OLF_OK;
}

void olf_ctx_destroy(olf_ctx* c)
{
    if (!c) return;
    if (c->has_tables) angle_tables_release(c->device, c->params.line.conv_libm_float);
    if (c->h_outslab) (void)hipHostFree(c->h_outslab);
    for (void* p : c->allocs) (void)hipFree(p);
    for (void* p : c->scratch) if (p) (void)hipFree(p);
    for (auto& r : c->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (hipEvent_t e : c->prof_pool) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_front) (void)hipEventDestroy(c->ev_front);
    if (c->ev_sort) (void)hipEventDestroy(c->ev_sort);
    if (c->ev_lbd) (void)hipEventDestroy(c->ev_lbd);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int olf_ctx_create(const olf_params* p, int width, int height, int max_images, olf_ctx** out)
{
    if (!p || !out || width < 64 || height < 64 || max_images < 1) { set_error("olf_ctx_create: bad argument"); return OLF_ERR_INVALID; }
    *out = nullptr;
    if (p->abi_version != OLF_ABI_VERSION || p->struct_size != (uint32_t)sizeof(olf_params)) {
        set_error("olf_ctx_create: the parameter block was not initialised by this library's olf_default_params() (abi_version / struct_size differ: "
                  "caller built against another include/orbline_types.h)");
        return OLF_ERR_INVALID;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("olf_ctx_create: no HIP device visible (this library has no CPU path)");
        return OLF_ERR_NODEVICE;
    }
    olf_ctx* c = new olf_ctx();
    c->params = *p; c->W = width; c->H = height; c->max_images = max_images;
    int rc = c->orb.build(p->orb, width, height);
    if (rc != OLF_OK) {
        set_error(rc == OLF_ERR_CAPACITY ? "olf_ctx_create: more than 32760 key points on one pyramid level (nfeatures too large for this scale factor / level count)"
                                         : "olf_ctx_create: image size / ORB parameters not supported (every pyramid level needs at least 62 x 62 pixels and a landscape cell grid)");
        delete c; return rc;
    }
    auto fail = [&](int code) { olf_ctx_destroy(c); return code; };
    if (hipGetDevice(&c->device) != hipSuccess) return fail(OLF_ERR_HIP);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(OLF_ERR_HIP); }
    const OrbGeom& g = c->orb.geom;
    const size_t n = (size_t)max_images;
    OrbDeviceBufs& b = c->ob;
    const bool allocTrace = getenv("OLF_ALLOC_TRACE") != nullptr;      // (tools/mem_per_pair.py: every buffer of the context, per image)
#define A(ptr, count) do { if ((rc = dev_alloc(c, &(ptr), (count))) != OLF_OK) return fail(rc); \
                           if (allocTrace) fprintf(stderr, "alloc %-18s %8.3f MB per image\n", #ptr, (double)(count) * sizeof(*(ptr)) / 1e6 / (double)n); } while (0)
    A(b.pyr, n * g.pyrBytes); A(b.blur, n * g.pyrBytes);
    A(b.cells, n * g.totalCells * g.cellCap); A(b.cellCount, n * g.totalCells);
    A(b.cand, n * g.candTotal); A(b.candNode, n * g.candTotal); A(b.candCount, n * g.nlevels);
    if (g.maxNodes > 2048) A(b.octSpill, n * g.nlevels * octree_lds_bytes(g.maxNodes));
    A(b.lvlKp, n * g.kpTotal); A(b.lvlCount, n * g.nlevels); A(b.lvlAngle, n * g.kpTotal);
    A(b.rx, c->orb.rx.size() + 1); A(b.ry, c->orb.ry.size() + 1); A(b.geom, 1); A(b.status, 256);
    A(c->d_uright, ((n + 1) / 2) * g.outCap); A(c->d_depth, ((n + 1) / 2) * g.outCap); A(c->d_sad, ((n + 1) / 2) * g.outCap); A(c->d_bestkey, ((n + 1) / 2) * g.outCap); A(c->d_rowperm, n * g.outCap);
    A(c->d_images, n * width * height); A(c->d_kps, n * g.outCap); A(c->d_desc, n * g.outCap * OLF_DESC_BYTES); A(c->d_counts, n);
#undef A
    if (hipMemcpy(b.rx, c->orb.rx.data(), c->orb.rx.size() * sizeof(ResizeCoef), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b.ry, c->orb.ry.data(), c->orb.ry.size() * sizeof(ResizeCoef), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b.geom, &g, sizeof(OrbGeom), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(b.status, 0, 1024) != hipSuccess) {
        set_error("olf_ctx_create: table upload failed");
        return fail(OLF_ERR_HIP);
    }
    // ---- line side ----
    rc = c->line.build(p->line, width, height, max_images);
    if (rc != OLF_OK) { set_error("olf_ctx_create: LSD parameters not supported"); return fail(rc); }
    const LineGeom& lg = c->line.geom;
    // The line stream gets a priority of its own.  The runtime multiplexes streams onto a few hardware queues, and two streams of the SAME priority can land on
    // one queue: the line path and the caller's stream then run back to back (measured: the host-to-host pipeline at 330 ms per 3072-pair batch instead of 220
    // whenever the caller's compute stream and this one aliased, profiles/r4aq_stream_queue_aliasing.txt).  Streams of different priorities never share a queue.
    // OLF_S2_PRIO: 1 low (default: it leaves the high level to a caller's copy streams -- pipeline.py), -1 high, 0 the caller's level (the hazard, for A/B);
    // the step time is the same for all three (226-231 ms on one box).
    int s2prio = 1;
    if (const char* e = getenv("OLF_S2_PRIO")) s2prio = atoi(e);
    int prLeast = 0, prGreatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest);
    const int prio = s2prio < 0 ? prGreatest : s2prio > 0 ? prLeast : 0;
    if (hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_front, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_sort, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_lbd, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) { set_error("stream/event creation failed"); return fail(OLF_ERR_HIP); }
    LineDeviceBufs& l = c->lb;
#define A(ptr, count) do { if ((rc = dev_alloc(c, &(ptr), (count))) != OLF_OK) return fail(rc); \
                           if (allocTrace) fprintf(stderr, "alloc %-18s %8.3f MB per image\n", #ptr, (double)(count) * sizeof(*(ptr)) / 1e6 / (double)n); } while (0)
    // batch contexts (> kBatchCtxImages images): the LSD blur and the enlarged working image are dead before the key kernel / the seed sort write the key
    // buffers (one stream, kernels in order): they live there
    const bool batchCtx = n > (size_t)kBatchCtxImages;
    const size_t keyWords = lg.wide ? 2 : 1;      // (64-bit sort keys: lsd_wide.hip)
    A(l.grad, n * lg.Ps); A(l.keysA, n * lg.Ps * keyWords); A(l.keysB, n * lg.Ps * keyWords);
    if (batchCtx && lg.Ps * 4 >= lg.pitchW * lg.H && lg.Ps * 4 >= lg.pitchS * lg.Hs) { l.lsdBlur = reinterpret_cast<uint8_t*>(l.keysA); l.scaled = reinterpret_cast<uint8_t*>(l.keysB); c->scaled_aliased = true; }
    else { A(l.lsdBlur, n * lg.pitchW * lg.H); A(l.scaled, n * lg.pitchS * lg.Hs); }
    A(l.topBuf, n * (size_t)lsd_seedsort_top_words()); A(l.keyCount, n * 32); A(l.maxN, n * 32); A(l.chunkCnt, n * ((lg.Ps + 4095) / 4096)); A(l.sortHist, n * (size_t)lsd_sort_max_chunks(lg.Ps) * 32); A(l.sortBase, n * 32);
    // chunk pool of the multi-wave growth: every pixel in a list once (Ps / 32) plus one partly filled chunk per logged region and ROB slot;
    // at least the 2 * Ps words the one-wave agent's log needs (LineGeom::regionStride, one stride for both formats).  An image that still runs out is
    // grown again by the one-wave agent (launch_lsd_grow)
    l.nChunks = lg.regionStride / 32;
    // region: chunk pool of the multi-wave growth / 8-byte (pixel, gradient word) log of the one-wave agent
    // (batch contexts: the one-wave agent only -- no owner words, a log sized by a bound (host_tables.cpp) and a spill arena of full-size logs for the images that outgrow it)
    A(l.region, n * (size_t)lg.regionStride); l.ownerImages = (batchCtx || lg.wide) ? 0 : (int)std::min<size_t>(n, kMwMaxImages); A(l.owner, (size_t)l.ownerImages * lg.Ps); A(l.links, n * (size_t)l.nChunks);
    // (a full-size log never spills: other contexts get an arena only when olf_debug_lsd_log_cap asks for one)
    l.spillBlocks = lg.regionStride / 2 >= lg.Ps ? 0 : (int)std::max<size_t>(4, n / 16);      // (one image in 16 may have more than half of its pixels in logged regions)
    if (l.spillBlocks) A(l.spill, (size_t)l.spillBlocks * 2 * lg.Ps);
    A(l.spillCtl, 64); A(l.spillOf, n);
    c->line.geom.logCap = lg.regionStride / 2; c->line.geom.spillBlocks = l.spillBlocks; c->line.geom.spillArena = l.spill; c->line.geom.spillCtl = l.spillCtl; c->line.geom.spillOf = l.spillOf;
    l.mgImages = (int)std::min<size_t>(n, kMgMaxImages); l.mgStride = lsd_grow_mg_stride(lg.maxRegions);
    A(l.mg, (size_t)l.mgImages * l.mgStride);
    A(l.rawLines, n * lg.maxDetect); A(l.rawCount, n); A(l.regCount, n); A(l.growFmt, n); A(l.lbdBlur, n * lg.pitchW * lg.H); A(l.dxdy, n * lg.pitchD * lg.H);
    A(l.rowSums, n * lg.outCap * 63 * 4); A(l.lbdStarts, n * lg.outCap * 64 * 2); A(l.rx, c->line.rx.size()); A(l.ry, c->line.ry.size()); A(l.geom, 1);
    A(c->d_kls, n * lg.outCap); A(c->d_ldesc, n * lg.outCap * OLF_DESC_BYTES); A(c->d_lcounts, n);
    {
        const size_t np = (n + 1) / 2;
        A(c->d_ldist, np * lg.outCap * lg.outCap); A(c->d_lm21, np * lg.outCap); A(c->d_lm12, np * lg.outCap);
        A(c->d_ldisp, np * lg.outCap * 2); A(c->d_lle, np * lg.outCap * 3);
        void* q = nullptr;
        if (hipMalloc(&q, stereo_lines_prep_bytes((int)n, lg.outCap)) != hipSuccess) { set_error("hipMalloc failed"); return fail(OLF_ERR_HIP); }
        c->allocs.push_back(q); c->d_lprep = q;
    }
    if (n <= 16) {
        const size_t np = (n + 1) / 2, ocap = g.outCap, lcap = lg.outCap;
        const size_t sz[11] = {n * ocap * sizeof(olf_keypoint), n * ocap * OLF_DESC_BYTES, n * sizeof(int), np * ocap * sizeof(float), np * ocap * sizeof(float),
                               n * lcap * sizeof(olf_keyline), n * lcap * OLF_DESC_BYTES, n * sizeof(int), np * lcap * sizeof(int), np * lcap * 2 * sizeof(float), np * lcap * 3 * sizeof(double)};
        size_t off = 0;
        for (int k = 0; k < 11; ++k) { c->outslab_off[k] = off; off += (sz[k] + 255) & ~(size_t)255; }
        c->outslab_off[11] = off; c->outslab_bytes = off;
        A(c->d_outslab, off);
        if (hipHostMalloc(reinterpret_cast<void**>(&c->h_outslab), off, hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc failed"); return fail(OLF_ERR_HIP); }
        uint8_t* q = c->d_outslab;
        c->d_kps = reinterpret_cast<olf_keypoint*>(q + c->outslab_off[0]); c->d_desc = q + c->outslab_off[1]; c->d_counts = reinterpret_cast<int*>(q + c->outslab_off[2]);
        c->d_uright = reinterpret_cast<float*>(q + c->outslab_off[3]); c->d_depth = reinterpret_cast<float*>(q + c->outslab_off[4]);
        c->d_kls = reinterpret_cast<olf_keyline*>(q + c->outslab_off[5]); c->d_ldesc = q + c->outslab_off[6]; c->d_lcounts = reinterpret_cast<int*>(q + c->outslab_off[7]);
        c->d_lm12 = reinterpret_cast<int*>(q + c->outslab_off[8]); c->d_ldisp = reinterpret_cast<float*>(q + c->outslab_off[9]); c->d_lle = reinterpret_cast<double*>(q + c->outslab_off[10]);
    }
#undef A
    l.status = b.status;
    if (hipMemcpy(l.rx, c->line.rx.data(), c->line.rx.size() * sizeof(ResizeCoef), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(l.ry, c->line.ry.data(), c->line.ry.size() * sizeof(ResizeCoef), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(l.geom, &lg, sizeof(LineGeom), hipMemcpyHostToDevice) != hipSuccess) { set_error("line table upload failed"); return fail(OLF_ERR_HIP); }
    if ((rc = angle_tables_acquire(c->device, p->line.conv_libm_float, l, c->stream)) != OLF_OK) return fail(rc);
    c->has_tables = true;
    *out = c;
    return OLF_OK;
}

static int check_status(olf_ctx* c);

// a deferred join that the caller has not asked for yet, in front of an entry that reads what the line stream is still writing
static int join_if_pending(olf_ctx* c, hipStream_t s)
{
    if (!c->join_pending) return OLF_OK;
    OLF_HIP_CHECK(hipStreamWaitEvent(s, c->ev_join, 0));
    if (s == c->pend_stream) c->join_pending = false;      // (a side stream's wait orders the side stream only)
    return OLF_OK;
}
static bool in_pending_line_outputs(const olf_ctx* c, const void* p)
{
    if (!c->join_pending || !p) return false;
    const uint8_t* q = static_cast<const uint8_t*>(p);
    for (const auto& r : c->pend) if (r.p && q >= r.p && q < r.p + r.bytes) return true;
    return false;
}

int olf_ctx_synchronize(olf_ctx* c)
{
    if (!c) return OLF_ERR_INVALID;
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream2));
    c->join_pending = false;            // (both streams are idle: nothing is left to wait for)
    return check_status(c);
}

int olf_ctx_set_input_event(olf_ctx* c, void* hip_event)
{
    if (!c) return OLF_ERR_INVALID;
    c->input_event = static_cast<hipEvent_t>(hip_event);
    return OLF_OK;
}

int olf_ctx_poll_status(olf_ctx* c)
{
    if (!c) return OLF_ERR_INVALID;
    return check_status(c);
}

// ---- stage profiling: HIP events on the launching streams -------------------------------------
int olf_profile_enable(olf_ctx* c, int on)
{
    if (!c) return OLF_ERR_INVALID;
    c->prof_on = on != 0;
    return OLF_OK;
}

static int prof_collect(olf_ctx* c)
{
    OLF_HIP_CHECK(hipDeviceSynchronize());
    for (auto& r : c->prof_recs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c->prof_ms[r.stage] += ms; c->prof_calls[r.stage] += 1; }
        c->prof_pool.push_back(r.a); c->prof_pool.push_back(r.b);
    }
    c->prof_recs.clear();
    return OLF_OK;
}

int olf_profile_reset(olf_ctx* c)
{
    if (!c) return OLF_ERR_INVALID;
    OLF_TRY(prof_collect(c));
    for (int i = 0; i < 16; ++i) { c->prof_ms[i] = 0; c->prof_calls[i] = 0; }
    return OLF_OK;
}

int olf_profile_stage_count(void) { return ST_COUNT; }
const char* olf_profile_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? kStageNames[i] : ""; }

int olf_profile_read(olf_ctx* c, double* total_ms, int32_t* calls)
{
    if (!c || !total_ms || !calls) return OLF_ERR_INVALID;
    OLF_TRY(prof_collect(c));
    for (int i = 0; i < ST_COUNT; ++i) { total_ms[i] = c->prof_ms[i]; calls[i] = c->prof_calls[i]; }
    return OLF_OK;
}

int olf_orb_scale_tables(const olf_ctx* c, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* fpl)
{
    if (!c) return OLF_ERR_INVALID;
    for (int i = 0; i < c->params.orb.nlevels; ++i) {
        if (scale) scale[i] = c->orb.sf[i];
        if (inv_scale) inv_scale[i] = c->orb.inv_sf[i];
        if (sigma2) sigma2[i] = c->orb.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = c->orb.inv_sigma2[i];
        if (fpl) fpl[i] = c->orb.nPerLevel[i];
    }
    return OLF_OK;
}

int olf_orb_level_sizes(const olf_ctx* c, int32_t* widths, int32_t* heights)
{
    if (!c) return OLF_ERR_INVALID;
    for (int i = 0; i < c->orb.geom.nlevels; ++i) { widths[i] = c->orb.geom.lv[i].w; heights[i] = c->orb.geom.lv[i].h; }
    return OLF_OK;
}

int olf_orb_capacity(const olf_ctx* c) { return c ? c->orb.geom.outCap : OLF_ERR_INVALID; }

int olf_orb_extract_dev(olf_ctx* c, const uint8_t* d_images, int n_images, olf_keypoint* d_kps, uint8_t* d_desc, int32_t* d_counts,
                        void* stream)
{
    if (!c || !d_images || !d_kps || !d_desc || !d_counts) { set_error("olf_orb_extract_dev: null argument"); return OLF_ERR_INVALID; }
    OLF_TRY(check_device(c, "olf_orb_extract_dev"));
    if (n_images < 0 || n_images > c->max_images) { set_error("olf_orb_extract_dev: n_images exceeds the context capacity"); return OLF_ERR_CAPACITY; }
    if (n_images == 0) return OLF_OK;   // ORBextractor::operator() returns silently on an empty image
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const OrbGeom& g = c->orb.geom;
    c->last_n_images = n_images;
    // (the blur only needs the pyramid: with orb_wait_after >= 2 it runs ahead of FAST, beside the dense half of the LSD front)
    const int wa = c->orb_wait_after;
    { StageScope t(c, s, ST_ORB_PYRAMID); OLF_TRY(launch_orb_pyramid(g, c->ob, d_images, n_images, s)); }
    if (wa == 1) OLF_HIP_CHECK(hipStreamWaitEvent(s, c->ev_front, 0));
    // wa == 4 (OLF_SCHED=5): the blur in the shadow of the seed sort -- it waits for the dense half of the LSD front only (ev_sort), FAST for all of it
    if (wa == 4) OLF_HIP_CHECK(hipStreamWaitEvent(s, c->ev_sort, 0));
    if (wa >= 2) { StageScope t(c, s, ST_ORB_BLUR); OLF_TRY(launch_orb_blur(g, c->ob, n_images, s)); }
    if (c->lbd_pre) {      // (also beside the seed sort: GaussianBlur + Sobel of the LBD octave need the input images only)
        OLF_TRY(launch_lbd_dense(c->line.geom, c->lb, d_images, c->W, n_images, s));
        OLF_HIP_CHECK(hipEventRecord(c->ev_lbd, s));
    }
    if (wa == 2 || wa == 4) OLF_HIP_CHECK(hipStreamWaitEvent(s, c->ev_front, 0));
    { StageScope t(c, s, ST_ORB_FAST); OLF_TRY(launch_orb_fast(g, c->ob, n_images, s)); }
    if (wa == 3) OLF_HIP_CHECK(hipStreamWaitEvent(s, c->ev_front, 0));
    { StageScope t(c, s, ST_ORB_OCTREE); OLF_TRY(launch_orb_octree(g, c->ob, n_images, s)); }
    if (wa < 2) { StageScope t(c, s, ST_ORB_BLUR); OLF_TRY(launch_orb_blur(g, c->ob, n_images, s)); }
    { StageScope t(c, s, ST_ORB_DESCRIBE); OLF_TRY(launch_orb_describe(g, c->ob, n_images, d_kps, d_desc, d_counts, g.outCap, s)); }
    return OLF_OK;
}

static int check_status(olf_ctx* c)
{
    int st[4] = {0, 0, 0, 0};
    OLF_HIP_CHECK(hipMemcpy(st, c->ob.status, sizeof(st), hipMemcpyDeviceToHost));
    if (st[0]) {
        (void)hipMemset(c->ob.status, 0, 16);
        set_error("device capacity overflow, flags=" + std::to_string(st[0]) +
                  " (1/2/4: ORB corner / candidate / key point buffers, 8: LSD regions, segments or pixel-list pool, 16: LSD growth watchdog, 32: frame record buffer, 64: LSD seed sort, final-range list of the grid-wide top levels)");
        return OLF_ERR_CAPACITY;
    }
    return OLF_OK;
}

int olf_orb_extract(olf_ctx* c, const uint8_t* images, int n_images, olf_keypoint* kps, uint8_t* desc, int32_t* counts)
{
    if (!c || !images || !kps || !desc || !counts) { set_error("olf_orb_extract: null argument"); return OLF_ERR_INVALID; }
    if (n_images < 0 || n_images > c->max_images) return OLF_ERR_CAPACITY;
    if (n_images == 0) return OLF_OK;
    const size_t npx = (size_t)c->W * c->H, cap = c->orb.geom.outCap;
    OLF_HIP_CHECK(hipMemcpyAsync(c->d_images, images, npx * n_images, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(olf_orb_extract_dev(c, c->d_images, n_images, c->d_kps, c->d_desc, c->d_counts, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(kps, c->d_kps, cap * n_images * sizeof(olf_keypoint), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(desc, c->d_desc, cap * n_images * OLF_DESC_BYTES, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(counts, c->d_counts, n_images * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return check_status(c);
}

int olf_orb_extract_strided(olf_ctx* c, const uint8_t* image, size_t row_stride, olf_keypoint* kps, uint8_t* desc, int32_t* count)
{
    if (!c || !image || !kps || !desc || !count || row_stride < (size_t)c->W) { set_error("olf_orb_extract_strided: bad argument"); return OLF_ERR_INVALID; }
    const size_t cap = c->orb.geom.outCap;
    OLF_HIP_CHECK(hipMemcpy2DAsync(c->d_images, c->W, image, row_stride, c->W, c->H, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(olf_orb_extract_dev(c, c->d_images, 1, c->d_kps, c->d_desc, c->d_counts, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(count, c->d_counts, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(kps, c->d_kps, cap * sizeof(olf_keypoint), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(desc, c->d_desc, cap * OLF_DESC_BYTES, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return check_status(c);
}

int olf_orb_pyramid_level(olf_ctx* c, int image, int level, int blurred, uint8_t* dst)
{
    if (!c || !dst || image < 0 || image >= c->max_images || level < 0 || level >= c->orb.geom.nlevels) return OLF_ERR_INVALID;
    const LevelGeom& L = c->orb.geom.lv[level];
    const uint8_t* src = (blurred ? c->ob.blur : c->ob.pyr) + (size_t)image * c->orb.geom.pyrBytes + L.offset;
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    OLF_HIP_CHECK(hipMemcpy2D(dst, L.w, src, L.pitch, L.w, L.h, hipMemcpyDeviceToHost));
    return OLF_OK;
}

int olf_debug_status(olf_ctx* c, int32_t* out64)
{
    if (!c || !out64) return OLF_ERR_INVALID;
    OLF_HIP_CHECK(hipDeviceSynchronize());
    OLF_HIP_CHECK(hipMemcpy(out64, c->ob.status, 256, hipMemcpyDeviceToHost));
    return OLF_OK;
}

size_t olf_frames_pack_bound(const olf_ctx* c, int n_pairs)
{
    return c && n_pairs >= 0 ? olf::pack_records_bound(n_pairs, c->orb.geom.outCap, c->line.geom.outCap) : 0;
}

int olf_stereo_points_mask_dev(olf_ctx* c, const float* d_depth, size_t n, uint8_t* d_mask, void* stream)
{
    if (!c || !d_depth || !d_mask) { set_error("olf_stereo_points_mask_dev: null argument"); return OLF_ERR_INVALID; }
    OLF_TRY(check_device(c, "olf_stereo_points_mask_dev"));
    return olf::launch_depth_mask(d_depth, d_mask, n, stream ? (hipStream_t)stream : c->stream);
}

int olf_frames_pack_dev(olf_ctx* c, const olf_frame_buffers* fb, int n_pairs, uint8_t* d_dst, size_t dst_capacity, uint64_t* d_bytes, void* stream)
{
    if (!c || !fb || !d_dst || !d_bytes || n_pairs < 0 || 2 * n_pairs > c->max_images || dst_capacity < 64 + (size_t)16 * n_pairs + 16) {
        set_error("olf_frames_pack_dev: bad argument"); return OLF_ERR_INVALID;
    }
    OLF_TRY(check_device(c, "olf_frames_pack_dev"));
    OLF_TRY(join_if_pending(c, stream ? (hipStream_t)stream : c->stream));      // (the packer reads every output of the frame call, the line path's too)
    void* ofs = nullptr;
    OLF_TRY(scratch_get(c, 3, (size_t)8 * n_pairs * sizeof(int) + 64, &ofs));
    return olf::launch_pack_records(*fb, n_pairs, c->orb.geom.outCap, c->line.geom.outCap, d_dst, dst_capacity, static_cast<int*>(ofs),
                                    reinterpret_cast<unsigned long long*>(d_bytes), c->ob.status, stream ? (hipStream_t)stream : c->stream);
}

int olf_debug_copy_bandwidth(olf_ctx* c, size_t bytes, int reps, double* gbytes_per_s)
{
    if (!c || !gbytes_per_s || bytes < 16 || reps < 1) { set_error("olf_debug_copy_bandwidth: bad argument"); return OLF_ERR_INVALID; }
    OLF_TRY(check_device(c, "olf_debug_copy_bandwidth"));
    void *a = nullptr, *b = nullptr;
    OLF_HIP_CHECK(hipMalloc(&a, bytes));
    if (hipMalloc(&b, bytes) != hipSuccess) { (void)hipFree(a); set_error("olf_debug_copy_bandwidth: hipMalloc failed"); return OLF_ERR_HIP; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipMemsetAsync(a, 1, bytes, c->stream);
    int rc = olf::launch_copy16(a, b, bytes, c->stream);                    // warm-up
    (void)hipEventRecord(e0, c->stream);
    for (int i = 0; i < reps && rc == OLF_OK; ++i) rc = olf::launch_copy16(i & 1 ? b : a, i & 1 ? a : b, bytes, c->stream);
    (void)hipEventRecord(e1, c->stream);
    const hipError_t se = hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(a); (void)hipFree(b);
    if (rc != OLF_OK || se != hipSuccess || !(ms > 0)) { set_error("olf_debug_copy_bandwidth: copy failed"); return OLF_ERR_HIP; }
    *gbytes_per_s = 2.0 * (double)(bytes / 16 * 16) * reps / (ms * 1e-3) / 1e9;      // bytes read + bytes written
    return OLF_OK;
}

int olf_debug_lsd_waves(olf_ctx* c, int waves_per_image, int rob_entries)
{
    const bool pow2 = rob_entries > 0 && (rob_entries & (rob_entries - 1)) == 0;
    if (!c || waves_per_image > 16 || waves_per_image < -1 || (rob_entries != 0 && (!pow2 || rob_entries < 128 || rob_entries > 1024))) {
        set_error("olf_debug_lsd_waves: bad argument"); return OLF_ERR_INVALID;
    }
    c->lb.forceNW = waves_per_image;
    c->lb.forceE = rob_entries;
    return OLF_OK;
}

// debug / tests: workgroups per image of the multi-wave growth (1, 2 or 4; 0: chosen from the batch size)
int olf_debug_lsd_groups(olf_ctx* c, int groups)
{
    if (!c || !(groups == 0 || groups == 1 || groups == 2 || groups == 4)) { set_error("olf_debug_lsd_groups: bad argument"); return OLF_ERR_INVALID; }
    c->lb.forceG = groups > 0 ? groups : -1;
    return OLF_OK;
}

// debug / tests: cap the one-wave agent's primary pixel log at `entries` (0: the context's own size): images whose logged regions need more move to the spill arena
int olf_debug_lsd_log_cap(olf_ctx* c, int entries)
{
    if (!c || entries < 0) { set_error("olf_debug_lsd_log_cap: bad argument"); return OLF_ERR_INVALID; }
    OLF_TRY(check_device(c, "olf_debug_lsd_log_cap"));
    if (entries > 0 && !c->lb.spill) {      // a context whose log holds every pixel has no arena of its own
        const size_t blocks = std::max<size_t>(2, (size_t)c->max_images - (size_t)c->max_images / 4);      // (tests: three quarters of the images may spill)
        void* q = nullptr;
        OLF_HIP_CHECK(hipMalloc(&q, blocks * 8 * (size_t)c->line.geom.Ps));
        c->allocs.push_back(q);
        c->lb.spill = static_cast<uint32_t*>(q); c->lb.spillBlocks = (int)blocks;
    }
    c->lb.logCapOverride = entries;
    LineGeom& g = c->line.geom;
    g.spillArena = c->lb.spill; g.spillBlocks = c->lb.spillBlocks;
    g.logCap = entries > 0 ? std::min(entries, g.regionStride / 2) : g.regionStride / 2;
    OLF_HIP_CHECK(hipDeviceSynchronize());
    OLF_HIP_CHECK(hipMemcpy(c->lb.geom, &g, sizeof(LineGeom), hipMemcpyHostToDevice));
    return OLF_OK;
}

// debug / tests: deal the growth groups of an image to consecutive workgroups (different XCDs under round-robin placement) instead of to one XCD
int olf_debug_lsd_scatter(olf_ctx* c, int on)
{
    if (!c) { set_error("olf_debug_lsd_scatter: bad argument"); return OLF_ERR_INVALID; }
    c->lb.scatter = on ? 1 : 0;
    return OLF_OK;
}

// debug / tests: the std::sort seed-order kernel (lsd_seedsort.hip) on a caller-supplied key array ((field << 22) | payload, sorted by the
// 10-bit field ascending exactly as libstdc++'s std::sort would leave it); kthr: only keys whose field is <= kthr are listed (-1: from the
// image statistics -- not meaningful here, pass n_bins - 1 to list everything); depth_limit: introsort's depth limit (-1: 2 * floor(log2 n))
int olf_debug_seed_sort(olf_ctx* c, const uint32_t* keys, int n, int kthr, int depth_limit, uint32_t* out, int32_t* out_n)
{
    if (!c || !keys || !out || !out_n || n < 0 || n > c->line.geom.Ps || kthr < 0 || kthr > 1023) { set_error("olf_debug_seed_sort: bad argument"); return OLF_ERR_INVALID; }
    OLF_HIP_CHECK(hipMemcpyAsync(c->lb.keysA, keys, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(launch_lsd_seedsort(c->line.geom, c->lb, 1, c->stream, n, kthr, depth_limit));
    int cnt = 0;
    OLF_HIP_CHECK(hipMemcpyAsync(&cnt, c->lb.keyCount, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    *out_n = cnt;
    if (cnt < 0 || cnt > n) { set_error("olf_debug_seed_sort: count out of range"); return OLF_ERR_HIP; }
    if (cnt) OLF_HIP_CHECK(hipMemcpy(out, c->lb.keysB, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    return OLF_OK;
}

// debug / tests: the 64-bit seed-order kernel (lsd_wide.hip) on a caller-supplied key array (field << 32 | payload); full = 0: compared by the field alone, the
// order libstdc++'s std::sort leaves (convention C.9 variant 1); full = 1: compared as whole words (variant 0); kthr: the keys whose field is <= kthr are listed;
// depth_limit: introsort's depth limit (-1: 2 * floor(log2 n)).  out receives the listed keys' payloads (the pixel addresses) in order.
int olf_debug_seed_sort_wide(olf_ctx* c, const uint64_t* keys, int n, int64_t kthr, int depth_limit, int full, uint32_t* out, int32_t* out_n)
{
    if (!c || !keys || !out || !out_n || n < 0 || n > c->line.geom.Ps || kthr < 0 || kthr > 0xffffffffll || !c->line.geom.wide) {
        set_error("olf_debug_seed_sort_wide: bad argument (the context must be a wide one: lsd_n_bins > 1024 or 2^22 pixels and more)"); return OLF_ERR_INVALID; }
    OLF_HIP_CHECK(hipMemcpyAsync(c->lb.keysA, keys, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(launch_lsd_sort_wide(c->line.geom, c->lb, 1, c->stream, n, (long long)kthr, depth_limit, full ? 1 : 0));
    int cnt = 0;
    OLF_HIP_CHECK(hipMemcpyAsync(&cnt, c->lb.keyCount, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    *out_n = cnt;
    if (cnt < 0 || cnt > n) { set_error("olf_debug_seed_sort_wide: count out of range"); return OLF_ERR_HIP; }
    if (cnt) OLF_HIP_CHECK(hipMemcpy(out, c->lb.keysB, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    return check_status(c);
}

// debug / tests: which seed-sort kernel runs (-1: chosen from the batch size; 0: one wave per image; 1 / 2: 4 / 8 waves per image)
int olf_debug_seed_sort_mode(olf_ctx* c, int mode)
{
    if (!c || mode < -1 || mode > 5) { set_error("olf_debug_seed_sort_mode: bad argument"); return OLF_ERR_INVALID; }
    c->lb.forceSortMode = mode;
    return OLF_OK;
}

// debug / tests: cap the chunk pool of the multi-wave growth (0: the whole pool) so that the fall-back to the one-wave agent can be exercised
int olf_debug_lsd_pool(olf_ctx* c, int pool_chunks)
{
    if (!c || pool_chunks < 0) { set_error("olf_debug_lsd_pool: bad argument"); return OLF_ERR_INVALID; }
    c->lb.poolChunks = pool_chunks;
    return OLF_OK;
}

// debug: the regions logged by the last growth for `image`, in detection order: (first chunk or list start, pixels, final region angle) triples
int olf_debug_lsd_regions(olf_ctx* c, int image, int32_t* start_n /* [cap][2] */, double* angle, int cap, int32_t* count)
{
    if (!c || !start_n || !angle || !count || image < 0 || image >= c->max_images) return OLF_ERR_INVALID;
    OLF_HIP_CHECK(hipDeviceSynchronize());
    int nr = 0;
    OLF_HIP_CHECK(hipMemcpy(&nr, c->lb.regCount + image, sizeof(int), hipMemcpyDeviceToHost));
    *count = nr;
    struct Rec { int start, n; double angle; };
    std::vector<Rec> r((size_t)std::min(nr, cap));
    if (!r.empty())
        OLF_HIP_CHECK(hipMemcpy(r.data(), reinterpret_cast<const Rec*>(c->lb.keysA) + (size_t)image * c->line.geom.maxRegions, r.size() * sizeof(Rec), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < r.size(); ++i) { start_n[2 * i] = r[i].start; start_n[2 * i + 1] = r[i].n; angle[i] = r[i].angle; }
    return OLF_OK;
}

// debug: the owner words (seed rank << 10 | ROB slot, 0xffffffff = never claimed) the last multi-region growth left for `image` (Ws*Hs words)
int olf_debug_lsd_owner(olf_ctx* c, int image, uint32_t* out)
{
    if (!c || !out || image < 0 || image >= c->lb.ownerImages) return OLF_ERR_INVALID;
    OLF_HIP_CHECK(hipDeviceSynchronize());
    OLF_HIP_CHECK(hipMemcpy(out, c->lb.owner + (size_t)image * c->line.geom.Ps, (size_t)c->line.geom.Ps * 4, hipMemcpyDeviceToHost));
    return OLF_OK;
}

int olf_debug_status_n(olf_ctx* c, int32_t* out, int n)
{
    if (!c || !out || n < 1 || n > 256) return OLF_ERR_INVALID;
    OLF_HIP_CHECK(hipDeviceSynchronize());
    OLF_HIP_CHECK(hipMemcpy(out, c->ob.status, (size_t)n * 4, hipMemcpyDeviceToHost));
    return OLF_OK;
}

int olf_debug_fdiv_sweep(olf_ctx* c, uint64_t seed, int blocks, int per_thread, uint64_t* mismatches)
{
    if (!c || !mismatches || blocks < 1 || per_thread < 1) { set_error("olf_debug_fdiv_sweep: bad argument"); return OLF_ERR_INVALID; }
    void* st = nullptr;
    OLF_TRY(scratch_get(c, 1, 64, &st));
    OLF_HIP_CHECK(hipMemsetAsync(st, 0, 8, c->stream));
    OLF_TRY(launch_fdiv_sweep((unsigned long long)seed, blocks, per_thread, (unsigned long long*)st, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(mismatches, st, 8, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

int olf_debug_sqrtq_sweep(olf_ctx* c, int count, uint64_t* mismatches)
{
    if (!c || !mismatches || count < 1) { set_error("olf_debug_sqrtq_sweep: bad argument"); return OLF_ERR_INVALID; }
    void* st = nullptr;
    OLF_TRY(scratch_get(c, 1, 64, &st));
    OLF_HIP_CHECK(hipMemsetAsync(st, 0, 8, c->stream));
    OLF_TRY(launch_sqrtq_sweep(count, (unsigned long long*)st, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(mismatches, st, 8, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

int olf_debug_align_sweep(olf_ctx* c, uint64_t seed, int blocks, int per_thread, uint64_t* out3)
{
    if (!c || !out3 || blocks < 1 || per_thread < 1) { set_error("olf_debug_align_sweep: bad argument"); return OLF_ERR_INVALID; }
    if (c->line.geom.alignTanLo < 0.f) { set_error("olf_debug_align_sweep: the cheap alignment test is off for lsd_ang_th > 80 degrees"); return OLF_ERR_INVALID; }
    void* st = nullptr;
    OLF_TRY(scratch_get(c, 1, 64, &st));
    OLF_HIP_CHECK(hipMemsetAsync(st, 0, 24, c->stream));
    OLF_TRY(launch_align_sweep(c->lb, (unsigned long long)seed, blocks, per_thread, (unsigned long long*)st, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(out3, st, 24, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

int olf_orb_debug_candidates(olf_ctx* c, int image, int level, int32_t* xys, int cap, int32_t* count)
{
    if (!c || !xys || !count || image < 0 || image >= c->max_images || level < 0 || level >= c->orb.geom.nlevels) return OLF_ERR_INVALID;
    const OrbGeom& g = c->orb.geom;
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    int n = 0;
    OLF_HIP_CHECK(hipMemcpy(&n, c->ob.candCount + image * g.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    *count = n;
    std::vector<uint32_t> tmp(std::max(n, 1));
    OLF_HIP_CHECK(hipMemcpy(tmp.data(), c->ob.cand + (size_t)image * g.candTotal + g.lv[level].candBase, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < std::min(n, cap); ++i) {
        xys[3 * i] = tmp[i] >> 20; xys[3 * i + 1] = (tmp[i] >> 8) & 0xfff; xys[3 * i + 2] = tmp[i] & 0xff;
    }
    return OLF_OK;
}

// ---------------------------------------------------------------------------------------------
int olf_stereo_points_dev(olf_ctx* c, int n_pairs, const olf_keypoint* d_kps, const uint8_t* d_desc, const int32_t* d_counts,
                          float* d_uright, float* d_depth, void* stream)
{
    if (!c || !d_kps || !d_desc || !d_counts || !d_uright || !d_depth) { set_error("olf_stereo_points_dev: null argument"); return OLF_ERR_INVALID; }
    OLF_TRY(check_device(c, "olf_stereo_points_dev"));
    if (n_pairs < 0 || 2 * n_pairs > c->max_images) return OLF_ERR_CAPACITY;
    if (n_pairs == 0) return OLF_OK;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    StageScope t(c, s, ST_STEREO_POINTS);
    return launch_stereo_points(c->orb.geom, c->ob, n_pairs, d_kps, d_desc, d_counts, c->orb.geom.outCap, c->params.stereo.bf,
                                c->params.stereo.fx, d_uright, d_depth, c->d_sad, c->d_bestkey, c->d_rowperm, s);
}

int olf_stereo_points(olf_ctx* c, const uint8_t* images, int n_pairs, olf_keypoint* kps, uint8_t* desc, int32_t* counts, float* uright,
                      float* depth)
{
    if (!c || !images || !kps || !desc || !counts || !uright || !depth) { set_error("olf_stereo_points: null argument"); return OLF_ERR_INVALID; }
    const int n_images = 2 * n_pairs;
    if (n_pairs < 0 || n_images > c->max_images) return OLF_ERR_CAPACITY;
    if (n_pairs == 0) return OLF_OK;
    const size_t npx = (size_t)c->W * c->H, cap = c->orb.geom.outCap;
    OLF_HIP_CHECK(hipMemcpyAsync(c->d_images, images, npx * n_images, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(olf_orb_extract_dev(c, c->d_images, n_images, c->d_kps, c->d_desc, c->d_counts, c->stream));
    OLF_TRY(olf_stereo_points_dev(c, n_pairs, c->d_kps, c->d_desc, c->d_counts, c->d_uright, c->d_depth, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(kps, c->d_kps, cap * n_images * sizeof(olf_keypoint), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(desc, c->d_desc, cap * n_images * OLF_DESC_BYTES, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(counts, c->d_counts, n_images * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(uright, c->d_uright, cap * n_pairs * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(depth, c->d_depth, cap * n_pairs * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return check_status(c);
}

int olf_match_bf_dev(olf_ctx* c, const uint8_t* dA, const int32_t* nA, int strideA, int a_step, const uint8_t* dB, const int32_t* nB,
                     int strideB, int b_step, int n_sets, float nnr, int best_lr, int32_t* d_m12, void* stream)
{
    if (!c || !dA || !dB || !nA || !nB || !d_m12 || strideA < 0 || strideB < 0 || n_sets < 0) { set_error("olf_match_bf_dev: bad argument"); return OLF_ERR_INVALID; }
    OLF_TRY(check_device(c, "olf_match_bf_dev"));
    if (n_sets == 0 || strideA == 0) return OLF_OK;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    // line descriptors / counts of a frame call whose line path is still running (deferred join): wait for it; ORB descriptors go ahead
    if (in_pending_line_outputs(c, dA) || in_pending_line_outputs(c, dB) || in_pending_line_outputs(c, nA) || in_pending_line_outputs(c, nB)) OLF_TRY(join_if_pending(c, s));
    void* ws = nullptr;
    OLF_TRY(scratch_get(c, 0, (size_t)3 * (strideA + strideB) * n_sets * sizeof(int), &ws));
    StageScope t(c, s, ST_MATCH_BF);
    return launch_match_bf(dA, nA, strideA, a_step, dB, nB, strideB, b_step, n_sets, nnr, best_lr, (int*)ws, d_m12, s);
}

// host staging helper: [descA | descB | nA nB | out...]
int olf_match_bf(olf_ctx* c, const uint8_t* descA, int nA, const uint8_t* descB, int nB, float nnr, int best_lr, int32_t* m12)
{
    if (!c || !m12 || nA < 0 || nB < 0 || (nA && !descA) || (nB && !descB)) { set_error("olf_match_bf: bad argument"); return OLF_ERR_INVALID; }
    if (nA == 0) return OLF_OK;
    void* st = nullptr;
    const size_t bA = (size_t)nA * 32, bB = (size_t)std::max(nB, 1) * 32;
    OLF_TRY(scratch_get(c, 1, bA + bB + 64 + (size_t)nA * 4, &st));
    uint8_t* dA = (uint8_t*)st; uint8_t* dB = dA + bA; int* dn = (int*)(dB + bB); int* dm = dn + 16;
    int n[2] = {nA, nB};
    OLF_HIP_CHECK(hipMemcpyAsync(dA, descA, bA, hipMemcpyHostToDevice, c->stream));
    if (nB) OLF_HIP_CHECK(hipMemcpyAsync(dB, descB, (size_t)nB * 32, hipMemcpyHostToDevice, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(dn, n, sizeof(n), hipMemcpyHostToDevice, c->stream));
    OLF_TRY(olf_match_bf_dev(c, dA, dn, nA, 1, dB, dn + 1, std::max(nB, 1), 1, 1, nnr, best_lr, dm, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(m12, dm, (size_t)nA * 4, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

int olf_knn2(olf_ctx* c, const uint8_t* descQ, int nQ, const uint8_t* descT, int nT, int32_t* idx0, int32_t* dist0, int32_t* dist1)
{
    if (!c || !idx0 || !dist0 || !dist1 || nQ < 0 || nT < 0 || (nQ && !descQ) || (nT && !descT)) { set_error("olf_knn2: bad argument"); return OLF_ERR_INVALID; }
    if (nQ == 0) return OLF_OK;
    void* st = nullptr;
    const size_t bQ = (size_t)nQ * 32, bT = (size_t)std::max(nT, 1) * 32;
    OLF_TRY(scratch_get(c, 1, bQ + bT + 64 + (size_t)nQ * 12, &st));
    uint8_t* dQ = (uint8_t*)st; uint8_t* dT = dQ + bQ; int* dn = (int*)(dT + bT); int* o = dn + 16;
    int n[2] = {nQ, nT};
    OLF_HIP_CHECK(hipMemcpyAsync(dQ, descQ, bQ, hipMemcpyHostToDevice, c->stream));
    if (nT) OLF_HIP_CHECK(hipMemcpyAsync(dT, descT, (size_t)nT * 32, hipMemcpyHostToDevice, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(dn, n, sizeof(n), hipMemcpyHostToDevice, c->stream));
    OLF_TRY(launch_knn2(dQ, dn, nQ, dT, dn + 1, std::max(nT, 1), 1, o, o + nQ, o + 2 * nQ, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(idx0, o, (size_t)nQ * 4, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(dist0, o + nQ, (size_t)nQ * 4, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(dist1, o + 2 * nQ, (size_t)nQ * 4, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

int olf_match_candidates_dev(olf_ctx* c, const uint8_t* dQ, int nQ, const uint8_t* dT, int nT, const int32_t* d_offs, const int32_t* d_cand,
                             uint16_t* d_dist, void* stream)
{
    if (!c || nQ < 0 || nT < 0 || (nQ && (!dQ || !d_offs || !d_dist))) { set_error("olf_match_candidates_dev: bad argument"); return OLF_ERR_INVALID; }
    OLF_TRY(check_device(c, "olf_match_candidates_dev"));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    StageScope t(c, s, ST_MATCH_BF);
    return launch_match_candidates(dQ, nQ, dT, nT, d_offs, d_cand, d_dist, s);
}

int olf_match_candidates(olf_ctx* c, const uint8_t* descQ, int nQ, const uint8_t* descT, int nT, const int32_t* offs, const int32_t* cand,
                         uint16_t* dist)
{
    if (!c || nQ < 0 || nT < 0 || (nQ && (!descQ || !offs)) || (nT && !descT)) { set_error("olf_match_candidates: bad argument"); return OLF_ERR_INVALID; }
    if (nQ == 0) return OLF_OK;
    const int nnz = offs[nQ];
    if (nnz < 0 || offs[0] != 0 || (nnz && (!cand || !dist))) { set_error("olf_match_candidates: bad CSR"); return OLF_ERR_INVALID; }
    if (nnz == 0) return OLF_OK;
    void* st = nullptr;
    const size_t bQ = (size_t)nQ * 32, bT = (size_t)std::max(nT, 1) * 32, bO = ((size_t)(nQ + 1) * 4 + 15) & ~(size_t)15, bC = ((size_t)nnz * 4 + 15) & ~(size_t)15;
    OLF_TRY(scratch_get(c, 1, bQ + bT + bO + bC + (size_t)nnz * 2 + 64, &st));
    uint8_t* dQ = (uint8_t*)st; uint8_t* dT = dQ + bQ; int* dO = (int*)(dT + bT); int* dC = (int*)((uint8_t*)dO + bO); uint16_t* dD = (uint16_t*)((uint8_t*)dC + bC);
    OLF_HIP_CHECK(hipMemcpyAsync(dQ, descQ, bQ, hipMemcpyHostToDevice, c->stream));
    if (nT) OLF_HIP_CHECK(hipMemcpyAsync(dT, descT, (size_t)nT * 32, hipMemcpyHostToDevice, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(dO, offs, (size_t)(nQ + 1) * 4, hipMemcpyHostToDevice, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(dC, cand, (size_t)nnz * 4, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(olf_match_candidates_dev(c, dQ, nQ, dT, nT, dO, dC, dD, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(dist, dD, (size_t)nnz * 2, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

int olf_cvt_gray_dev(olf_ctx* c, const uint8_t* d_src, int code, int n_images, uint8_t* d_gray, void* stream)
{
    if (!c || !d_src || !d_gray || code < 0 || code > 3 || n_images < 0) { set_error("olf_cvt_gray_dev: bad argument"); return OLF_ERR_INVALID; }
    OLF_TRY(check_device(c, "olf_cvt_gray_dev"));
    if (n_images == 0) return OLF_OK;
    return launch_cvt_gray(d_src, d_gray, c->W, c->H, code, n_images, stream ? (hipStream_t)stream : c->stream);
}

int olf_init_undistort_rectify_map_dev(olf_ctx* c, const double* K, const double* D, int n_dist, const double* R, const double* P, int w, int h,
                                       float* d_map1, float* d_map2, void* stream)
{
    if (!c || !K || !R || !P || !d_map1 || !d_map2 || w < 1 || h < 1 || n_dist < 0 || n_dist > 8 || (n_dist && !D)) {
        set_error("olf_init_undistort_rectify_map_dev: bad argument (up to 8 distortion coefficients k1 k2 p1 p2 k3 k4 k5 k6)"); return OLF_ERR_INVALID;
    }
    OLF_TRY(check_device(c, "olf_init_undistort_rectify_map_dev"));
    // iR = (P * R)^-1 in double: cv::invert's closed form for 3 x 3 (convention C.13)
    double m[9];
    for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) m[3 * r + q] = P[3 * r] * R[q] + P[3 * r + 1] * R[3 + q] + P[3 * r + 2] * R[6 + q];
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (d == 0.) { set_error("olf_init_undistort_rectify_map_dev: P * R is singular"); return OLF_ERR_INVALID; }
    d = 1. / d;
    double ir[9];
    ir[0] = (m[4] * m[8] - m[5] * m[7]) * d; ir[1] = (m[2] * m[7] - m[1] * m[8]) * d; ir[2] = (m[1] * m[5] - m[2] * m[4]) * d;
    ir[3] = (m[5] * m[6] - m[3] * m[8]) * d; ir[4] = (m[0] * m[8] - m[2] * m[6]) * d; ir[5] = (m[2] * m[3] - m[0] * m[5]) * d;
    ir[6] = (m[3] * m[7] - m[4] * m[6]) * d; ir[7] = (m[1] * m[6] - m[0] * m[7]) * d; ir[8] = (m[0] * m[4] - m[1] * m[3]) * d;
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n_dist; ++i) k[i] = D[i];
    return launch_init_rectify_map(ir, k, K[0], K[4], K[2], K[5], w, h, d_map1, d_map2, stream ? (hipStream_t)stream : c->stream);
}

int olf_init_undistort_rectify_map(olf_ctx* c, const double* K, const double* D, int n_dist, const double* R, const double* P, int w, int h, float* map1, float* map2)
{
    if (!c || !map1 || !map2 || w < 1 || h < 1) { set_error("olf_init_undistort_rectify_map: bad argument"); return OLF_ERR_INVALID; }
    void* st = nullptr;
    const size_t bm = (size_t)w * h * sizeof(float);
    OLF_TRY(scratch_get(c, 2, 2 * bm + 64, &st));
    float* d1 = static_cast<float*>(st);
    float* d2 = d1 + (size_t)w * h;
    OLF_TRY(olf_init_undistort_rectify_map_dev(c, K, D, n_dist, R, P, w, h, d1, d2, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(map1, d1, bm, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(map2, d2, bm, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

int olf_remap_linear_dev(olf_ctx* c, const uint8_t* d_src, int sw, int sh, const float* d_mapx, const float* d_mapy, int dw, int dh, int n_images,
                         uint8_t* d_dst, void* stream)
{
    if (!c || !d_src || !d_mapx || !d_mapy || !d_dst || sw < 1 || sh < 1 || dw < 1 || dh < 1 || n_images < 0) { set_error("olf_remap_linear_dev: bad argument"); return OLF_ERR_INVALID; }
    OLF_TRY(check_device(c, "olf_remap_linear_dev"));
    if (n_images == 0) return OLF_OK;
    return launch_remap_linear(d_src, sw, sh, d_mapx, d_mapy, dw, dh, d_dst, n_images, stream ? (hipStream_t)stream : c->stream);
}

int olf_cvt_gray(olf_ctx* c, const uint8_t* src, int code, int n_images, uint8_t* gray)
{
    if (!c || !src || !gray || code < 0 || code > 3 || n_images < 0) { set_error("olf_cvt_gray: bad argument"); return OLF_ERR_INVALID; }
    if (n_images == 0) return OLF_OK;
    const size_t npx = (size_t)c->W * c->H * n_images, cn = code >= 2 ? 4 : 3;
    void* st = nullptr;
    OLF_TRY(scratch_get(c, 2, npx * (cn + 1) + 64, &st));
    uint8_t* ds = (uint8_t*)st; uint8_t* dd = ds + ((npx * cn + 15) & ~(size_t)15);
    OLF_HIP_CHECK(hipMemcpyAsync(ds, src, npx * cn, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(olf_cvt_gray_dev(c, ds, code, n_images, dd, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(gray, dd, npx, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

int olf_remap_linear(olf_ctx* c, const uint8_t* src, int sw, int sh, const float* mapx, const float* mapy, int dw, int dh, int n_images, uint8_t* dst)
{
    if (!c || !src || !mapx || !mapy || !dst || sw < 1 || sh < 1 || dw < 1 || dh < 1 || n_images < 0) { set_error("olf_remap_linear: bad argument"); return OLF_ERR_INVALID; }
    if (n_images == 0) return OLF_OK;
    const size_t bs = ((size_t)sw * sh * n_images + 15) & ~(size_t)15, bm = (size_t)dw * dh * 4, bd = (size_t)dw * dh * n_images;
    void* st = nullptr;
    OLF_TRY(scratch_get(c, 2, bs + 2 * bm + bd + 64, &st));
    uint8_t* ds = (uint8_t*)st; float* mx = (float*)(ds + bs); float* my = mx + (size_t)dw * dh; uint8_t* dd = (uint8_t*)(my + (size_t)dw * dh);
    OLF_HIP_CHECK(hipMemcpyAsync(ds, src, (size_t)sw * sh * n_images, hipMemcpyHostToDevice, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(mx, mapx, bm, hipMemcpyHostToDevice, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(my, mapy, bm, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(olf_remap_linear_dev(c, ds, sw, sh, mx, my, dw, dh, n_images, dd, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(dst, dd, bd, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

int olf_hamming_matrix(olf_ctx* c, const uint8_t* descA, int nA, const uint8_t* descB, int nB, uint16_t* out)
{
    if (!c || !out || nA < 0 || nB < 0 || (nA && !descA) || (nB && !descB)) { set_error("olf_hamming_matrix: bad argument"); return OLF_ERR_INVALID; }
    if (nA == 0 || nB == 0) return OLF_OK;
    void* st = nullptr;
    const size_t bA = (size_t)nA * 32, bB = (size_t)nB * 32, bO = (size_t)nA * nB * 2;
    OLF_TRY(scratch_get(c, 1, bA + bB + bO, &st));
    uint8_t* dA = (uint8_t*)st; uint8_t* dB = dA + bA; uint16_t* dO = (uint16_t*)(dB + bB);
    OLF_HIP_CHECK(hipMemcpyAsync(dA, descA, bA, hipMemcpyHostToDevice, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(dB, descB, bB, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(launch_hamming_matrix(dA, nA, dB, nB, dO, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(out, dO, bO, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

int olf_distinctive_descriptors(olf_ctx* c, const uint8_t* desc, const int32_t* offs, int n_points, int32_t* best)
{
    if (!c || !offs || !best || n_points < 0) { set_error("olf_distinctive_descriptors: bad argument"); return OLF_ERR_INVALID; }
    if (n_points == 0) return OLF_OK;
    const int total = offs[n_points];
    if (offs[0] != 0 || total < 0 || (total > 0 && !desc)) { set_error("olf_distinctive_descriptors: bad offsets"); return OLF_ERR_INVALID; }
    for (int i = 0; i < n_points; ++i) {
        if (offs[i + 1] < offs[i]) { set_error("olf_distinctive_descriptors: offsets must not decrease"); return OLF_ERR_INVALID; }
        if (offs[i + 1] - offs[i] > 1024) { set_error("olf_distinctive_descriptors: more than 1024 observations of one landmark"); return OLF_ERR_CAPACITY; }
    }
    void* st = nullptr;
    const size_t bD = ((size_t)total * 32 + 15) & ~(size_t)15, bO = (size_t)(n_points + 1) * 4;
    OLF_TRY(scratch_get(c, 1, bD + bO + (size_t)n_points * 4 + 64, &st));
    uint8_t* dD = (uint8_t*)st; int* dO = (int*)(dD + bD); int* dB = dO + n_points + 1;
    if (total) OLF_HIP_CHECK(hipMemcpyAsync(dD, desc, (size_t)total * 32, hipMemcpyHostToDevice, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(dO, offs, bO, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(launch_distinctive(dD, dO, n_points, dB, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(best, dB, (size_t)n_points * 4, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

// ---------------------------------------------------------------------------------------------
int olf_line_capacity(const olf_ctx* c) { return c ? c->line.geom.outCap : OLF_ERR_INVALID; }

int olf_line_extract_dev(olf_ctx* c, const uint8_t* d_images, int n_images, olf_keyline* d_kls, uint8_t* d_ldesc, int32_t* d_lcounts, void* stream)
{
    if (!c || !d_images || !d_kls || !d_ldesc || !d_lcounts) { set_error("olf_line_extract_dev: null argument"); return OLF_ERR_INVALID; }
    OLF_TRY(check_device(c, "olf_line_extract_dev"));
    if (n_images < 0 || n_images > c->max_images) return OLF_ERR_CAPACITY;
    if (n_images == 0) return OLF_OK;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    // mark_front: the ORB stream waits for ev_front, recorded behind the whole front including the seed ordering.  OLF_SCHED + 16 records it in
    // front of the std::sort replay instead -- measured worse (297.7 against 282.7 ms per 3072-pair step: the ORB kernels then slow the sort down and
    // still meet the growth agents afterwards), kept as a switch for experiments.
    static const bool front_before_sort = getenv("OLF_SCHED") && (atoi(getenv("OLF_SCHED")) & 16);
    const bool early = c->mark_front && c->line.geom.seedOrder == 1 && front_before_sort;
    c->lb.sortEvent = early ? c->ev_front : (c->mark_front && c->line.geom.seedOrder == 1) ? c->ev_sort : nullptr;
    if (c->mark_front && c->line.geom.seedOrder != 1) OLF_HIP_CHECK(hipEventRecord(c->ev_sort, s));      // (no sort kernel to hide behind: the event is already true)
    { StageScope t(c, s, ST_LSD_FRONT); const int rc = launch_lsd_front(c->line.geom, c->lb, d_images, c->W, n_images, s); c->lb.sortEvent = nullptr; OLF_TRY(rc); }
    if (c->mark_front && !early) OLF_HIP_CHECK(hipEventRecord(c->ev_front, s));
    { StageScope t(c, s, ST_LSD_GROW); OLF_TRY(launch_lsd_grow(c->line.geom, c->lb, n_images, s)); }
    { StageScope t(c, s, ST_LSD_RECT); OLF_TRY(launch_lsd_rect(c->line.geom, c->lb, n_images, s)); }
    if (c->defer_lbd) return OLF_OK;
    { StageScope t(c, s, ST_LINE_LBD); OLF_TRY(launch_line_select_lbd(c->line.geom, c->lb, d_images, c->W, n_images, d_kls, d_ldesc, d_lcounts, s)); }
    return OLF_OK;
}

int olf_line_extract(olf_ctx* c, const uint8_t* images, int n_images, olf_keyline* kls, uint8_t* ldesc, int32_t* lcounts)
{
    if (!c || !images || !kls || !ldesc || !lcounts) { set_error("olf_line_extract: null argument"); return OLF_ERR_INVALID; }
    if (n_images < 0 || n_images > c->max_images) return OLF_ERR_CAPACITY;
    if (n_images == 0) return OLF_OK;
    const size_t npx = (size_t)c->W * c->H, cap = c->line.geom.outCap;
    OLF_HIP_CHECK(hipMemcpyAsync(c->d_images, images, npx * n_images, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(olf_line_extract_dev(c, c->d_images, n_images, c->d_kls, c->d_ldesc, c->d_lcounts, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(kls, c->d_kls, cap * n_images * sizeof(olf_keyline), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(ldesc, c->d_ldesc, cap * n_images * OLF_DESC_BYTES, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(lcounts, c->d_lcounts, n_images * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return check_status(c);
}

int olf_line_extract_strided(olf_ctx* c, const uint8_t* image, size_t row_stride, olf_keyline* kls, uint8_t* ldesc, int32_t* lcount)
{
    if (!c || !image || !kls || !ldesc || !lcount || row_stride < (size_t)c->W) { set_error("olf_line_extract_strided: bad argument"); return OLF_ERR_INVALID; }
    const size_t cap = c->line.geom.outCap;
    OLF_HIP_CHECK(hipMemcpy2DAsync(c->d_images, c->W, image, row_stride, c->W, c->H, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(olf_line_extract_dev(c, c->d_images, 1, c->d_kls, c->d_ldesc, c->d_lcounts, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(kls, c->d_kls, cap * sizeof(olf_keyline), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(ldesc, c->d_ldesc, cap * OLF_DESC_BYTES, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(lcount, c->d_lcounts, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return check_status(c);
}

int olf_lbd_compute(olf_ctx* c, const uint8_t* images, int n_images, const olf_keyline* kls, const int32_t* lcounts, uint8_t* ldesc)
{
    if (!c || !images || !kls || !lcounts || !ldesc) { set_error("olf_lbd_compute: null argument"); return OLF_ERR_INVALID; }
    if (n_images < 0 || n_images > c->max_images) return OLF_ERR_CAPACITY;
    if (n_images == 0) return OLF_OK;
    const size_t npx = (size_t)c->W * c->H, cap = c->line.geom.outCap;
    for (int i = 0; i < n_images; ++i)
        if (lcounts[i] < 0 || lcounts[i] > (int)cap) { set_error("olf_lbd_compute: count exceeds capacity"); return OLF_ERR_CAPACITY; }
    OLF_HIP_CHECK(hipMemcpyAsync(c->d_images, images, npx * n_images, hipMemcpyHostToDevice, c->stream));
    // stage the caller's key lines as the "raw" list, with selection disabled by pretending they are final
    OLF_HIP_CHECK(hipMemcpyAsync(c->d_kls, kls, cap * n_images * sizeof(olf_keyline), hipMemcpyHostToDevice, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(c->d_lcounts, lcounts, n_images * sizeof(int), hipMemcpyHostToDevice, c->stream));
    OLF_TRY(launch_lbd_only(c->line.geom, c->lb, c->d_images, c->W, n_images, c->d_kls, c->d_ldesc, c->d_lcounts, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(ldesc, c->d_ldesc, cap * n_images * OLF_DESC_BYTES, hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

int olf_lsd_debug_scaled(olf_ctx* c, int image, uint8_t* dst, int32_t* ws, int32_t* hs)
{
    if (!c || !dst || image < 0 || image >= c->max_images) return OLF_ERR_INVALID;
    if (c->scaled_aliased) { set_error("olf_lsd_debug_scaled: a batch context does not keep the enlarged image (use a context of at most 2048 images)"); return OLF_ERR_INVALID; }
    const LineGeom& g = c->line.geom;
    OLF_HIP_CHECK(hipDeviceSynchronize());
    OLF_HIP_CHECK(hipMemcpy2D(dst, g.Ws, c->lb.scaled + (size_t)image * g.pitchS * g.Hs, g.pitchS, g.Ws, g.Hs, hipMemcpyDeviceToHost));
    if (ws) *ws = g.Ws;
    if (hs) *hs = g.Hs;
    return OLF_OK;
}

int olf_stereo_lines_dev(olf_ctx* c, int n_pairs, const olf_keyline* d_kls, const uint8_t* d_ldesc, const int32_t* d_lcounts, int32_t* d_m12,
                         float* d_disp, double* d_le, void* stream)
{
    if (!c || !d_kls || !d_ldesc || !d_lcounts || !d_m12 || !d_disp || !d_le) { set_error("olf_stereo_lines_dev: null argument"); return OLF_ERR_INVALID; }
    OLF_TRY(check_device(c, "olf_stereo_lines_dev"));
    if (n_pairs < 0 || 2 * n_pairs > c->max_images) return OLF_ERR_CAPACITY;
    if (n_pairs == 0) return OLF_OK;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    // key lines / descriptors of a frame call whose line path is still running (deferred join): this stream waits for it first
    if (s != c->stream2 && (in_pending_line_outputs(c, d_kls) || in_pending_line_outputs(c, d_ldesc) || in_pending_line_outputs(c, d_lcounts))) OLF_TRY(join_if_pending(c, s));
    StageScope t(c, s, ST_STEREO_LINES);
    return launch_stereo_lines(c->W, c->H, c->params.stereo, n_pairs, d_kls, d_ldesc, d_lcounts, c->line.geom.outCap, c->d_lprep, c->d_ldist,
                               c->d_lm21, d_m12, d_disp, d_le, s);
}

int olf_stereo_lines(olf_ctx* c, int n_pairs, const olf_keyline* kls, const uint8_t* ldesc, const int32_t* lcounts, int32_t* m12, float* disp,
                     double* le)
{
    if (!c || !kls || !ldesc || !lcounts || !m12 || !disp || !le) { set_error("olf_stereo_lines: null argument"); return OLF_ERR_INVALID; }
    if (n_pairs < 0 || 2 * n_pairs > c->max_images) return OLF_ERR_CAPACITY;
    if (n_pairs == 0) return OLF_OK;
    const size_t cap = c->line.geom.outCap, ni = 2 * (size_t)n_pairs;
    for (size_t i = 0; i < ni; ++i)
        if (lcounts[i] < 0 || lcounts[i] > (int)cap) { set_error("olf_stereo_lines: count exceeds capacity"); return OLF_ERR_CAPACITY; }
    OLF_HIP_CHECK(hipMemcpyAsync(c->d_kls, kls, cap * ni * sizeof(olf_keyline), hipMemcpyHostToDevice, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(c->d_ldesc, ldesc, cap * ni * OLF_DESC_BYTES, hipMemcpyHostToDevice, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(c->d_lcounts, lcounts, ni * sizeof(int), hipMemcpyHostToDevice, c->stream));
    OLF_TRY(olf_stereo_lines_dev(c, n_pairs, c->d_kls, c->d_ldesc, c->d_lcounts, c->d_lm12, c->d_ldisp, c->d_lle, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(m12, c->d_lm12, cap * n_pairs * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(disp, c->d_ldisp, cap * n_pairs * 2 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipMemcpyAsync(le, c->d_lle, cap * n_pairs * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    return OLF_OK;
}

int olf_stereo_frames_dev(olf_ctx* c, const uint8_t* d_images, int n_pairs, const olf_frame_buffers* o, void* stream)
{
    if (!c || !d_images || !o || !o->kps || !o->desc || !o->counts || !o->uright || !o->depth || !o->kls || !o->ldesc || !o->lcounts ||
        !o->lmatches12 || !o->ldisp || !o->lle) { set_error("olf_stereo_frames_dev: null argument"); return OLF_ERR_INVALID; }
    OLF_TRY(check_device(c, "olf_stereo_frames_dev"));
    if (n_pairs < 0 || 2 * n_pairs > c->max_images) return OLF_ERR_CAPACITY;
    if (n_pairs == 0) return OLF_OK;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const int n_images = 2 * n_pairs;
    if (c->join_pending) { OLF_HIP_CHECK(hipStreamWaitEvent(s, c->ev_join, 0)); c->join_pending = false; }      // (a deferred join the caller never asked for)
    // The line path runs beside the ORB path on the second stream (the reference's 4 extraction threads, src/Frame.cc:164-171).  The ORB
    // stream builds its pyramid beside the dense half of the LSD front and then waits for the front's end: two dense pipelines at once only
    // slow each other down (OLF_SCHED=0: 290 ms per 3072-pair step), whereas FAST beside the latency-bound growth agents is hidden almost
    // completely (105 ms beside them, 24 alone, and the agents keep their stand-alone speed).  Round 3, ms per step: wait before the pyramid
    // 266.3, behind it 260.3, behind pyramid + blur 262.5, behind FAST 279.  Round 4 (profiles/r4o_*): behind the pyramid 241.6, and with the blur moved
    // into the shadow of the one-wave seed sort (schedule 5, the default) 235.6 .. 239.5.  OLF_ONE_STREAM serialises the two paths.
    static const bool one_stream = getenv("OLF_ONE_STREAM") != nullptr;
    if (one_stream) {
        OLF_TRY(olf_line_extract_dev(c, d_images, n_images, o->kls, o->ldesc, o->lcounts, s));
        OLF_TRY(olf_stereo_lines_dev(c, n_pairs, o->kls, o->ldesc, o->lcounts, o->lmatches12, o->ldisp, o->lle, s));
        OLF_TRY(olf_orb_extract_dev(c, d_images, n_images, o->kps, o->desc, o->counts, s));
        OLF_TRY(olf_stereo_points_dev(c, n_pairs, o->kps, o->desc, o->counts, o->uright, o->depth, s));
        return OLF_OK;
    }
    // OLF_SCHED: 0 both paths at once; 1 .. 4: the ORB stream waits for the LSD front before its pyramid / behind the pyramid / behind pyramid + blur /
    // behind pyramid + blur + FAST (+ 16: the front ends before the seed sort); 5: pyramid beside the dense front, blur beside the seed sort, FAST behind it
    static const int sched = (getenv("OLF_SCHED") ? atoi(getenv("OLF_SCHED")) : 5) & 15;
    // schedule 5 also moves the LBD gradient images (GaussianBlur + Sobel of the input, 6 ms of dense work) from the tail of the line stream into the
    // seed ordering's shadow on the ORB stream (OLF_LBD_PRE=0: off); the line stream's selection + LBD then wait for ev_lbd, so they are enqueued
    // behind the ORB extraction here (an event has to be recorded before the wait for it is enqueued)
    static const bool lbd_pre = sched == 5 && !(getenv("OLF_LBD_PRE") && atoi(getenv("OLF_LBD_PRE")) == 0);
    // Fork.  With the caller's input event (olf_ctx_set_input_event) the line stream does not wait for what is still queued on `s` -- the ORB / stereo /
    // matching tail of the previous batch -- and the LSD front of this batch runs beside it.  Safe with lbd_pre only: the LSD front, growth and rectangles
    // write line-path scratch that nothing on `s` reads, and selection / LBD / line stereo (which write the output buffers the previous batch's matchers
    // and packer on `s` may still read) sit behind ev_lbd, recorded on `s` behind all of the previous batch's work there.
    // (the event is consumed by the call: one-shot -- a stale handle must never order a later call)
    hipEvent_t in_ev = c->input_event;
    c->input_event = nullptr;
    if (in_ev && lbd_pre) OLF_HIP_CHECK(hipStreamWaitEvent(c->stream2, in_ev, 0));
    else {
        OLF_HIP_CHECK(hipEventRecord(c->ev_fork, s));
        OLF_HIP_CHECK(hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    }
    c->mark_front = sched != 0;
    c->defer_lbd = lbd_pre;
    c->lb.skipScaled = !(getenv("OLF_KEEP_SCALED") && atoi(getenv("OLF_KEEP_SCALED")) != 0);      // (nothing behind the fused kernel reads the working image)
    const int rcl = olf_line_extract_dev(c, d_images, n_images, o->kls, o->ldesc, o->lcounts, c->stream2);
    c->mark_front = false; c->defer_lbd = false; c->lb.skipScaled = false;
    OLF_TRY(rcl);
    if (!lbd_pre) {
        OLF_TRY(olf_stereo_lines_dev(c, n_pairs, o->kls, o->ldesc, o->lcounts, o->lmatches12, o->ldisp, o->lle, c->stream2));
        OLF_HIP_CHECK(hipEventRecord(c->ev_join, c->stream2));
    }
    if (sched == 1) OLF_HIP_CHECK(hipStreamWaitEvent(s, c->ev_front, 0));
    c->orb_wait_after = sched >= 2 ? sched - 1 : 0;
    c->lbd_pre = lbd_pre;
    const int rco = olf_orb_extract_dev(c, d_images, n_images, o->kps, o->desc, o->counts, s);
    c->orb_wait_after = 0; c->lbd_pre = false;
    OLF_TRY(rco);
    if (lbd_pre) {
        OLF_HIP_CHECK(hipStreamWaitEvent(c->stream2, c->ev_lbd, 0));
        { StageScope t(c, c->stream2, ST_LINE_LBD); OLF_TRY(launch_line_select_lbd(c->line.geom, c->lb, d_images, c->W, n_images, o->kls, o->ldesc, o->lcounts, c->stream2, true)); }
        OLF_TRY(olf_stereo_lines_dev(c, n_pairs, o->kls, o->ldesc, o->lcounts, o->lmatches12, o->ldisp, o->lle, c->stream2));
        OLF_HIP_CHECK(hipEventRecord(c->ev_join, c->stream2));
    }
    OLF_TRY(olf_stereo_points_dev(c, n_pairs, o->kps, o->desc, o->counts, o->uright, o->depth, s));
    // the join.  Deferred (olf_ctx_set_deferred_join): the caller's stream goes on with the ORB outputs -- key points, descriptors, stereo points are complete on
    // it here -- while the line path's tail (selection, LBD, line stereo: 20 ms of a 3072-pair batch) still runs on the line stream; olf_stereo_frames_join_dev
    // (or the next call) makes the stream wait for it
    if (c->deferred_join) {
        c->join_pending = true;
        c->pend_stream = s;
        const size_t lc = (size_t)c->line.geom.outCap;
        c->pend[0] = {reinterpret_cast<const uint8_t*>(o->kls), (size_t)n_images * lc * sizeof(olf_keyline)};
        c->pend[1] = {reinterpret_cast<const uint8_t*>(o->ldesc), (size_t)n_images * lc * OLF_DESC_BYTES};
        c->pend[2] = {reinterpret_cast<const uint8_t*>(o->lcounts), (size_t)n_images * sizeof(int32_t)};
        c->pend[3] = {reinterpret_cast<const uint8_t*>(o->lmatches12), (size_t)n_pairs * lc * sizeof(int32_t)};
        c->pend[4] = {reinterpret_cast<const uint8_t*>(o->ldisp), (size_t)n_pairs * lc * 2 * sizeof(float)};
        c->pend[5] = {reinterpret_cast<const uint8_t*>(o->lle), (size_t)n_pairs * lc * 3 * sizeof(double)};
    } else OLF_HIP_CHECK(hipStreamWaitEvent(s, c->ev_join, 0));
    return OLF_OK;
}

int olf_ctx_set_deferred_join(olf_ctx* c, int on)
{
    if (!c) return OLF_ERR_INVALID;
    c->deferred_join = on != 0;
    return OLF_OK;
}

int olf_stereo_frames_join_dev(olf_ctx* c, void* stream)
{
    if (!c) return OLF_ERR_INVALID;
    return join_if_pending(c, stream ? (hipStream_t)stream : c->stream);
}

int olf_stereo_frames(olf_ctx* c, const uint8_t* images, int n_pairs, const olf_frame_buffers* o)
{
    if (!c || !images || !o) { set_error("olf_stereo_frames: null argument"); return OLF_ERR_INVALID; }
    if (n_pairs < 0 || 2 * n_pairs > c->max_images) return OLF_ERR_CAPACITY;
    if (n_pairs == 0) return OLF_OK;
    const size_t npx = (size_t)c->W * c->H, cap = c->orb.geom.outCap, lcap = c->line.geom.outCap, ni = 2 * (size_t)n_pairs;
    olf_frame_buffers d;
    d.kps = c->d_kps; d.desc = c->d_desc; d.counts = c->d_counts; d.uright = c->d_uright; d.depth = c->d_depth;
    d.kls = c->d_kls; d.ldesc = c->d_ldesc; d.lcounts = c->d_lcounts; d.lmatches12 = c->d_lm12; d.ldisp = c->d_ldisp; d.lle = c->d_lle;
    c->input_event = nullptr;           // (the upload below is on the context's stream: the line stream must fork from it, whatever event an earlier caller left)
    OLF_HIP_CHECK(hipMemcpyAsync(c->d_images, images, npx * ni, hipMemcpyHostToDevice, c->stream));
    OLF_TRY(olf_stereo_frames_dev(c, c->d_images, n_pairs, &d, c->stream));
    OLF_TRY(olf_stereo_frames_join_dev(c, c->stream));      // (a context with the deferred join on: the copies below read the line outputs)
    hipStream_t s = c->stream;
    if (c->d_outslab && (int)ni == c->max_images) {
        // one copy for all eleven arrays, then host copies out of the pinned slab (the arrays are laid out for exactly this many images)
        OLF_HIP_CHECK(hipMemcpyAsync(c->h_outslab, c->d_outslab, c->outslab_bytes, hipMemcpyDeviceToHost, s));
        OLF_HIP_CHECK(hipStreamSynchronize(s));
        const uint8_t* h = c->h_outslab;
        memcpy(o->kps, h + c->outslab_off[0], cap * ni * sizeof(olf_keypoint)); memcpy(o->desc, h + c->outslab_off[1], cap * ni * OLF_DESC_BYTES);
        memcpy(o->counts, h + c->outslab_off[2], ni * sizeof(int)); memcpy(o->uright, h + c->outslab_off[3], cap * n_pairs * sizeof(float));
        memcpy(o->depth, h + c->outslab_off[4], cap * n_pairs * sizeof(float)); memcpy(o->kls, h + c->outslab_off[5], lcap * ni * sizeof(olf_keyline));
        memcpy(o->ldesc, h + c->outslab_off[6], lcap * ni * OLF_DESC_BYTES); memcpy(o->lcounts, h + c->outslab_off[7], ni * sizeof(int));
        memcpy(o->lmatches12, h + c->outslab_off[8], lcap * n_pairs * sizeof(int)); memcpy(o->ldisp, h + c->outslab_off[9], lcap * n_pairs * 2 * sizeof(float));
        memcpy(o->lle, h + c->outslab_off[10], lcap * n_pairs * 3 * sizeof(double));
        return check_status(c);
    }
    OLF_HIP_CHECK(hipMemcpyAsync(o->kps, d.kps, cap * ni * sizeof(olf_keypoint), hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipMemcpyAsync(o->desc, d.desc, cap * ni * OLF_DESC_BYTES, hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipMemcpyAsync(o->counts, d.counts, ni * sizeof(int), hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipMemcpyAsync(o->uright, d.uright, cap * n_pairs * sizeof(float), hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipMemcpyAsync(o->depth, d.depth, cap * n_pairs * sizeof(float), hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipMemcpyAsync(o->kls, d.kls, lcap * ni * sizeof(olf_keyline), hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipMemcpyAsync(o->ldesc, d.ldesc, lcap * ni * OLF_DESC_BYTES, hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipMemcpyAsync(o->lcounts, d.lcounts, ni * sizeof(int), hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipMemcpyAsync(o->lmatches12, d.lmatches12, lcap * n_pairs * sizeof(int), hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipMemcpyAsync(o->ldisp, d.ldisp, lcap * n_pairs * 2 * sizeof(float), hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipMemcpyAsync(o->lle, d.lle, lcap * n_pairs * 3 * sizeof(double), hipMemcpyDeviceToHost, s));
    OLF_HIP_CHECK(hipStreamSynchronize(s));
    return check_status(c);
}

}  // extern "C"
