// api.cpp -- the extern "C" boundary of liborbline_hip.so (include/orbline.h).
// Owns the device context; every compute entry point launches HIP kernels -- there is no CPU
// fallback: without a gfx950 device olf_ctx_create fails with OLF_ERR_NODEVICE.
#include "olf_internal.hpp"
#include "../../include/orbline.h"
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
    std::vector<void*> allocs;
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
    p->orb.nfeatures = 2000; p->orb.scale_factor = 1.2f; p->orb.nlevels = 8; p->orb.ini_th_fast = 20; p->orb.min_th_fast = 7;
    p->line.lsd_nfeatures = 500; p->line.min_line_length = 0.025; p->line.lsd_refine = 0; p->line.lsd_scale = 1.2;
    p->line.lsd_sigma_scale = 0.6; p->line.lsd_quant = 2.0; p->line.lsd_ang_th = 22.5; p->line.lsd_log_eps = 1.0;
    p->line.lsd_density_th = 0.6; p->line.lsd_n_bins = 1024;
    p->stereo.fx = 718.856f; p->stereo.bf = 386.1448f; p->stereo.matching_s_ws = 10; p->stereo.line_sim_th = 0.75;
    p->stereo.min_ratio_12_l = 0.9; p->stereo.min_disp = 1.0; p->stereo.line_horiz_th = 0.1; p->stereo.stereo_overlap_th = 0.75;
    p->stereo.ls_min_disp_ratio = 0.7; p->stereo.best_lr_matches = 1;
    return OLF_OK;
}

void olf_ctx_destroy(olf_ctx* c)
{
    if (!c) return;
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int olf_ctx_create(const olf_params* p, int width, int height, int max_images, olf_ctx** out)
{
    if (!p || !out || width < 64 || height < 64 || max_images < 1) { set_error("olf_ctx_create: bad argument"); return OLF_ERR_INVALID; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("olf_ctx_create: no HIP device visible (this library has no CPU path)");
        return OLF_ERR_NODEVICE;
    }
    olf_ctx* c = new olf_ctx();
    c->params = *p; c->W = width; c->H = height; c->max_images = max_images;
    int rc = c->orb.build(p->orb, width, height);
    if (rc != OLF_OK) { set_error("olf_ctx_create: image size / ORB parameters not supported"); delete c; return rc; }
    auto fail = [&](int code) { olf_ctx_destroy(c); return code; };
    if (hipGetDevice(&c->device) != hipSuccess) return fail(OLF_ERR_HIP);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return fail(OLF_ERR_HIP); }
    const OrbGeom& g = c->orb.geom;
    const size_t n = (size_t)max_images;
    OrbDeviceBufs& b = c->ob;
#define A(ptr, count) if ((rc = dev_alloc(c, &(ptr), (count))) != OLF_OK) return fail(rc)
    A(b.pyr, n * g.pyrBytes); A(b.blur, n * g.pyrBytes); A(b.score, n * g.pyrBytes);
    A(b.cells, n * g.totalCells * g.cellCap); A(b.cellCount, n * g.totalCells);
    A(b.cand, n * g.candTotal); A(b.candNode, n * g.candTotal); A(b.candCount, n * g.nlevels);
    A(b.lvlKp, n * g.kpTotal); A(b.lvlCount, n * g.nlevels); A(b.lvlAngle, n * g.kpTotal);
    A(b.rx, c->orb.rx.size() + 1); A(b.ry, c->orb.ry.size() + 1); A(b.geom, 1); A(b.status, 4);
    A(c->d_images, n * width * height); A(c->d_kps, n * g.outCap); A(c->d_desc, n * g.outCap * OLF_DESC_BYTES); A(c->d_counts, n);
#undef A
    if (hipMemcpy(b.rx, c->orb.rx.data(), c->orb.rx.size() * sizeof(ResizeCoef), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b.ry, c->orb.ry.data(), c->orb.ry.size() * sizeof(ResizeCoef), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b.geom, &g, sizeof(OrbGeom), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(b.status, 0, 16) != hipSuccess || hipMemset(b.score, 0, n * g.pyrBytes) != hipSuccess) {
        set_error("olf_ctx_create: table upload failed");
        return fail(OLF_ERR_HIP);
    }
    *out = c;
    return OLF_OK;
}

int olf_ctx_synchronize(olf_ctx* c)
{
    if (!c) return OLF_ERR_INVALID;
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
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
    if (n_images < 0 || n_images > c->max_images) { set_error("olf_orb_extract_dev: n_images exceeds the context capacity"); return OLF_ERR_CAPACITY; }
    if (n_images == 0) return OLF_OK;   // ORBextractor::operator() returns silently on an empty image
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    const OrbGeom& g = c->orb.geom;
    c->last_n_images = n_images;
    OLF_TRY(launch_orb_pyramid(g, c->ob, d_images, n_images, s));
    OLF_TRY(launch_orb_fast(g, c->ob, n_images, s));
    OLF_TRY(launch_orb_octree(g, c->ob, n_images, s));
    OLF_TRY(launch_orb_blur(g, c->ob, n_images, s));
    OLF_TRY(launch_orb_describe(g, c->ob, n_images, d_kps, d_desc, d_counts, g.outCap, s));
    return OLF_OK;
}

static int check_status(olf_ctx* c)
{
    int st[4] = {0, 0, 0, 0};
    OLF_HIP_CHECK(hipMemcpy(st, c->ob.status, sizeof(st), hipMemcpyDeviceToHost));
    if (st[0]) {
        (void)hipMemset(c->ob.status, 0, 16);
        set_error("device capacity overflow, flags=" + std::to_string(st[0]));
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

int olf_orb_pyramid_level(olf_ctx* c, int image, int level, int blurred, uint8_t* dst)
{
    if (!c || !dst || image < 0 || image >= c->max_images || level < 0 || level >= c->orb.geom.nlevels) return OLF_ERR_INVALID;
    const LevelGeom& L = c->orb.geom.lv[level];
    const uint8_t* src = (blurred ? c->ob.blur : c->ob.pyr) + (size_t)image * c->orb.geom.pyrBytes + L.offset;
    OLF_HIP_CHECK(hipStreamSynchronize(c->stream));
    OLF_HIP_CHECK(hipMemcpy2D(dst, L.w, src, L.pitch, L.w, L.h, hipMemcpyDeviceToHost));
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

}  // extern "C"
