// lsd.hip -- cv::LineSegmentDetector (OpenCV 3.4 lsd.cpp, LSD_REFINE_NONE) as called by
// LSDDetectorC::detectImpl (reference Thirdparty/line_descriptor/src/LSDDetector_custom.cpp:227-324), gfx950.
//
// Data-parallel front half (one pass each): sigma-0.6 blur, x1.2 bilinear upsample, 2x2 gradient.  The reference keeps fp64 modgrad +
// angle per pixel (16 B); both are pure functions of the integer gradient pair, so a pixel is one 32-bit word (gx:11 | gy:11 | ISO |
// NOTDEF | USED) and everything derived from (gx, gy) -- level-line angle, its cos / sin -- comes from per-context tables indexed by
// the packed pair (k_lsd_angle_table), bit-identical to evaluating the reference's expressions.
// Pseudo-ordering: every defined pixel becomes a key ((n_bins-1-bin) << 22 | address), emitted in raster order; a stable segmented
// radix sort (rocPRIM) over the 10 bin bits yields "bins high to low, raster order inside a bin".
// Region growing is order-dependent by construction (seed order, running region angle, shared `used` map), so one wave per image
// (k_lsd_grow, "the agent") replays it sequentially: the 3x3 neighbourhoods of up to 8 FIFO entries are examined in parallel lanes and
// only the accept chain is serial -- run as speculative rounds that need one fastAtan2 per round instead of one per accepted pixel.
// Regions that are large enough are logged and fitted afterwards, in parallel, by k_lsd_rect / k_lsd_emit (region2rect + KeyLine).
// Parallelism comes from the batch: thousands of images in flight, up to 8 agents per SIMD.
#include "lsd_device.hpp"

namespace olf {


__device__ __forceinline__ int refl101(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * (n - 1) - p;
    return p;
}

int launch_gauss7_img(const uint8_t* src, int srcPitch, size_t srcStride, uint8_t* dst, int dstPitch, size_t dstStride, int W, int H,
                      const LineGeom& g, int which, int n_images, hipStream_t s)
{
    if (!which && g.lsdWideR > 3) return launch_sep_wide(src, srcStride, srcPitch, dst, dstStride, dstPitch, W, H, g.lsdWide, g.lsdWideR, n_images, s);
    return launch_sep7(src, srcStride, srcPitch, dst, dstStride, dstPitch, W, H, which ? g.lbdTaps : g.lsdTaps, n_images, s);
}

// x1.2 bilinear upsample (cv::resize INTER_LINEAR, App. A.2)
__global__ __launch_bounds__(256) void k_lsd_upsample(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                      const LineGeom* __restrict__ gp, const ResizeCoef* __restrict__ rx,
                                                      const ResizeCoef* __restrict__ ry)
{
    const LineGeom& g = *gp;
    const int img = blockIdx.y;
    const int quads = (g.Ws + 3) >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= quads * g.Hs) return;
    const int dy = idx / quads, dx0 = (idx - dy * quads) * 4;
    const uint8_t* s = src + (size_t)img * g.pitchW * g.H;
    const ResizeCoef cy = ry[dy];
    const int y0 = min(max((int)cy.ofs, 0), g.H - 1), y1 = min(max((int)cy.ofs + 1, 0), g.H - 1);
    const uint8_t* S0 = s + (size_t)y0 * g.pitchW;
    const uint8_t* S1 = s + (size_t)y1 * g.pitchW;
    const int b0 = cy.a0, b1 = cy.a1;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int dx = dx0 + k;
        if (dx < g.Ws) {
            const ResizeCoef cx = rx[dx];
            const int sx = cx.ofs, sx1 = min(sx + 1, g.W - 1);
            const int h0 = S0[sx] * cx.a0 + S0[sx1] * cx.a1;
            const int h1 = S1[sx] * cx.a0 + S1[sx1] * cx.a1;
            // INTER_LINEAR: 11-bit coefficients, OpenCV's two-step shift; INTER_LINEAR_EXACT (C.10): 8.8 x 0.8 fixed point, one rounding
            const int v = g.resizeExact ? (h0 * b0 + h1 * b1 + 32768) >> 16 : (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            out |= (uint32_t)(v & 0xff) << (8 * k);
        }
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)img * g.pitchS * g.Hs + (size_t)dy * g.pitchS + dx0) = out;
}

// ll_angle, first half: gradient pair per pixel + per-image max of gx^2+gy^2 over defined pixels.
// One block covers LG_CHUNK consecutive pixels of one image (a single atomic per block).
constexpr int LG_CHUNK = 4096;
__global__ __launch_bounds__(256) void k_lsd_grad(const uint8_t* __restrict__ scaled, uint32_t* __restrict__ grad,
                                                  const LineGeom* __restrict__ gp, int* __restrict__ maxN, int* __restrict__ chunkCnt)
{
    __shared__ int s_max[4], s_def[4];
    const LineGeom& g = *gp;
    const int img = blockIdx.y;
    int n = 0, ndef = 0;
    const uint8_t* sc = scaled + (size_t)img * g.pitchS * g.Hs;
    // 4 consecutive pixels per thread and step: two 8-byte (unaligned) loads per image row instead of 16 byte loads, one 16-byte store
    uint32_t* gout = grad + (size_t)img * g.Ps;
#pragma unroll 2
    for (int k = 0; k < LG_CHUNK / 1024; ++k) {
        const int idx0 = blockIdx.x * LG_CHUNK + (k * 256 + threadIdx.x) * 4;
        if (idx0 >= g.Ps) break;
        const int y = idx0 / g.Ws, x = idx0 - y * g.Ws;
        uint32_t packed[4] = {kNotDef, kNotDef, kNotDef, kNotDef};
        if (idx0 + 3 < g.Ps && x + 8 <= g.Ws && y < g.Hs - 1) {
            const uint8_t* r0 = sc + (size_t)y * g.pitchS + x;
            unsigned long long w0, w1;
            __builtin_memcpy(&w0, r0, 8);
            __builtin_memcpy(&w1, r0 + g.pitchS, 8);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int A = (int)((w0 >> (8 * p)) & 0xff), B = (int)((w0 >> (8 * p + 8)) & 0xff);
                const int C = (int)((w1 >> (8 * p)) & 0xff), D = (int)((w1 >> (8 * p + 8)) & 0xff);
                const int DA = D - A, BC = B - C;
                const int gx = DA + BC, gy = DA - BC;
                const int nn = gx * gx + gy * gy;
                if (nn >= g.nThr) { packed[p] = pack_g(gx, gy); n = max(n, nn); ++ndef; }
                else packed[p] = pack_g(gx, gy) | kNotDef;      // undefined, but its bin is still needed by the std::sort seed order (lsd_seedsort.hip)
            }
        } else {
            int yy = y, xx = x;
            for (int p = 0; p < 4; ++p, ++xx) {
                if (xx >= g.Ws) { xx -= g.Ws; ++yy; }
                if (idx0 + p < g.Ps && xx < g.Ws - 1 && yy < g.Hs - 1) {
                    const uint8_t* r0 = sc + (size_t)yy * g.pitchS + xx;
                    const uint8_t* r1 = r0 + g.pitchS;
                    const int DA = (int)r1[1] - (int)r0[0], BC = (int)r0[1] - (int)r1[0];
                    const int gx = DA + BC, gy = DA - BC;
                    const int nn = gx * gx + gy * gy;
                    if (nn >= g.nThr) { packed[p] = pack_g(gx, gy); n = max(n, nn); ++ndef; }
                    else packed[p] = pack_g(gx, gy) | kNotDef;
                }
            }
        }
        if (idx0 + 3 < g.Ps && ((reinterpret_cast<uintptr_t>(gout + idx0) & 15) == 0))
            *reinterpret_cast<uint4*>(gout + idx0) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        else
            for (int p = 0; p < 4 && idx0 + p < g.Ps; ++p) gout[idx0 + p] = packed[p];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n = max(n, __shfl_xor(n, o)); ndef += __shfl_xor(ndef, o); }
    if ((threadIdx.x & 63) == 0) { s_max[threadIdx.x >> 6] = n; s_def[threadIdx.x >> 6] = ndef; }
    __syncthreads();
    if (threadIdx.x == 0) {
        n = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
        if (n > 0) atomicMax(&maxN[img * 32], n);   // counters padded to one per 128-byte line
        chunkCnt[(size_t)img * gridDim.x + blockIdx.x] = s_def[0] + s_def[1] + s_def[2] + s_def[3];   // defined pixels of this chunk
    }
}

// LSD's x1.2 enlargement (cv::resize INTER_LINEAR) and ll_angle's gradient in ONE pass (round 4): the register-only resize of filters.hip
// (k_resize_strip: one thread = 4 output columns x 8 rows, a wave per block, source rows requested one output row ahead) computes a fifth column and a
// ninth row, and the 2 x 2 differences of the values it holds are the gradient words -- the working image is still written (the debug entry and
// LSD_REFINE_ADV's rect_nfa read nothing of it, but olf_lsd_debug_scaled does) and never read back.  Saves k_lsd_grad's pass (its loads, unpacking and
// index arithmetic: 0.49 M instructions per image) and a kernel boundary.  Only for the std::sort seed order, which does not need k_lsd_grad's per-chunk
// counts of defined pixels.
#ifndef OLF_UG_ROWS
#define OLF_UG_ROWS 8
#endif
constexpr int UG_ROWS = OLF_UG_ROWS;
__global__ __launch_bounds__(64) void k_lsd_upgrad(const uint8_t* __restrict__ src, uint8_t* __restrict__ scaled, uint32_t* __restrict__ grad,
                                                   const LineGeom* __restrict__ gp, const ResizeCoef* __restrict__ rx, const ResizeCoef* __restrict__ ry,
                                                   int* __restrict__ maxN, int nsx, int writeScaled)
{
    const LineGeom& g = *gp;
    const int q = blockIdx.x * 64 + threadIdx.x;
    const int img = blockIdx.z;
    const int sw = g.W, sh = g.H, dw = g.Ws, dh = g.Hs, srcPitch = g.pitchW;
    int nmax = 0;
    if (q < nsx) {
        const int dx0 = 4 * q, dy0 = blockIdx.y * UG_ROWS;
        const uint8_t* s = src + (size_t)img * g.pitchW * g.H;
        uint8_t* d = scaled + (size_t)img * g.pitchS * g.Hs + dx0;
        uint32_t* gout = grad + (size_t)img * g.Ps;
        typedef unsigned short us2 __attribute__((ext_vector_type(2)));
        uint32_t sel[5], coef[5];
        const int base = min((int)rx[dx0].ofs, srcPitch - 8);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const ResizeCoef c = rx[min(dx0 + k, dw - 1)];
            const int o0 = (int)c.ofs - base, o1 = min((int)c.ofs + 1, sw - 1) - base;
            sel[k] = (uint32_t)o0 | 0x0c00u | ((uint32_t)o1 << 16) | 0x0c000000u;
            coef[k] = (uint32_t)(uint16_t)c.a0 | ((uint32_t)(uint16_t)c.a1 << 16);
        }
        struct Raw { uint32_t w[2]; };
        auto load_row = [&](int r) -> Raw { Raw v; __builtin_memcpy(v.w, s + (size_t)r * srcPitch + base, 8); return v; };
        auto hrow = [&](const Raw& v, uint32_t* h) {
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const uint32_t pr = __builtin_amdgcn_perm(v.w[1], v.w[0], sel[k]);
                us2 va, vb;
                __builtin_memcpy(&va, &pr, 4); __builtin_memcpy(&vb, &coef[k], 4);
                h[k] = __builtin_amdgcn_udot2(va, vb, 0u, false) >> 4;
            }
        };
        auto rows_of = [&](int dy, int& r0, int& r1, uint32_t& b0, uint32_t& b1) {
            const ResizeCoef cy = ry[min(dy, dh - 1)];
            r0 = min(max((int)cy.ofs, 0), sh - 1); r1 = min(max((int)cy.ofs + 1, 0), sh - 1);
            b0 = (uint32_t)cy.a0; b1 = (uint32_t)cy.a1;
        };
        int r0, r1, ra = -1, rb = -1;
        uint32_t b0, b1;
        rows_of(dy0, r0, r1, b0, b1);
        Raw p0 = load_row(r0), p1 = load_row(r1);
        uint32_t hA[5] = {0, 0, 0, 0, 0}, hB[5] = {0, 0, 0, 0, 0};
        int prev[5] = {0, 0, 0, 0, 0};       // the working image's row above the current one, columns dx0 .. dx0 + 4
#pragma unroll
        for (int rr = 0; rr <= UG_ROWS; ++rr) {
            const int dy = dy0 + rr;          // (rr == UG_ROWS: the row below the strip, for the last row's differences only)
            if (dy > dh || (rr > 0 && dy - 1 >= dh)) break;
            int n0, n1;
            uint32_t c0, c1;
            rows_of(dy + 1, n0, n1, c0, c1);
            Raw q0 = p0, q1 = p1;
            if (rr < UG_ROWS) {
                if (n0 != r0 && n0 != r1) q0 = load_row(n0);
                if (n1 != n0 && n1 != r1) q1 = load_row(n1);
                __builtin_amdgcn_sched_barrier(0);
            }
            uint32_t h0[5], h1[5];
            if (r0 == ra) { for (int k = 0; k < 5; ++k) h0[k] = hA[k]; }
            else if (r0 == rb) { for (int k = 0; k < 5; ++k) h0[k] = hB[k]; }
            else hrow(p0, h0);
            if (r1 == r0) { for (int k = 0; k < 5; ++k) h1[k] = h0[k]; }
            else if (r1 == rb) { for (int k = 0; k < 5; ++k) h1[k] = hB[k]; }
            else hrow(p1, h1);
            int cur[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) cur[k] = (int)(((((b0 * h0[k]) >> 16) + ((b1 * h1[k]) >> 16) + 2u) >> 2) & 0xffu);
            if (writeScaled && rr < UG_ROWS && dy < dh) {
                const uint32_t out = (uint32_t)cur[0] | ((uint32_t)cur[1] << 8) | ((uint32_t)cur[2] << 16) | ((uint32_t)cur[3] << 24);
                __builtin_memcpy(d + (size_t)dy * g.pitchS, &out, 4);      // (columns beyond Ws inside the last quad are scratch bytes of the padded pitch)
            }
            if (rr > 0) {
                // ll_angle's differences for row dy - 1: A = (x, y), B = (x + 1, y), C = (x, y + 1), D = (x + 1, y + 1)
                const int y = dy - 1;
                uint32_t packed[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int A = prev[k], B = prev[k + 1], Cc = cur[k], D = cur[k + 1];
                    const int DA = D - A, BC = B - Cc;
                    const int gx = DA + BC, gy = DA - BC;
                    const int nn = gx * gx + gy * gy;
                    const bool inside = dx0 + k < dw - 1 && y < dh - 1;
                    packed[k] = inside ? (pack_g(gx, gy) | (nn >= g.nThr ? 0u : kNotDef)) : kNotDef;
                    if (inside && nn >= g.nThr) nmax = max(nmax, nn);
                }
                // (one 16-byte store per quad and row: four dword stores per lane wrote every sector in four partial pieces)
                uint32_t* o = gout + (size_t)y * dw + dx0;
                const uintptr_t oa = reinterpret_cast<uintptr_t>(o);
                if (dx0 + 3 < dw && (oa & 15) == 0) *reinterpret_cast<uint4*>(o) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
                else if (dx0 + 3 < dw && (oa & 7) == 0) {
                    reinterpret_cast<uint2*>(o)[0] = make_uint2(packed[0], packed[1]); reinterpret_cast<uint2*>(o)[1] = make_uint2(packed[2], packed[3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (dx0 + k < dw) o[k] = packed[k];
                }
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) prev[k] = cur[k];
            ra = r0; rb = r1;
#pragma unroll
            for (int k = 0; k < 5; ++k) { hA[k] = h0[k]; hB[k] = h1[k]; }
            r0 = n0; r1 = n1; b0 = c0; b1 = c1; p0 = q0; p1 = q1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
    if (threadIdx.x == 0 && nmax > 0) atomicMax(&maxN[img * 32], nmax);   // counters padded to one per 128-byte line
}

// ll_angle, second half, and the isolated-seed test, in one pass over the gradient words.
//
// Keys: bin of every defined pixel -> sort key.  The keys are emitted in raster order (each wave compacts a contiguous quarter of the chunk,
// chunk bases come from k_lsd_grad's per-chunk counts), so the pseudo-ordering only has to order the 10 bin bits with a stable sort: equal
// bins stay in raster order, which is the reference's list order.
//
// Isolated seeds: a seed whose 8 neighbours are all undefined or not aligned with the seed's own angle can never grow: its region is the seed
// alone (region_grow tests every neighbour against reg_angle == the seed angle and nothing is ever added), whatever has been used before.
// That is a static property of the gradient field; it is flagged here (bit ISO of the gradient word) so that the sequential growth can retire
// such seeds without running a growth step.
//
// The block first evaluates the level-line angle (fastAtan2(gx, -gy), degrees) of every pixel of its chunk and of one image row (+1 pixel)
// either side into LDS -- the neighbours' angles come from there, not from a second pass over memory -- and blocks are numbered so that
// the chunks of an image follow each other on ONE XCD (workgroups are dealt round-robin to the 8 XCDs, each with an L2 of its own): the
// halo rows are then L2 hits.
// (the level-line angle of an undefined pixel, and of every position outside the image, is NaN in the block's LDS copy: every comparison with it is false and
// v_max / v_min skip it, so the isolated-seed test needs no bounds predicates at all -- the image's last column and last row are undefined by construction
// (ll_angle), which makes the row-major wrap-around neighbours of columns 0 and Ws - 2 undefined pixels too)
constexpr int KEYS_THREADS = 512;      // 8 waves share the 36 KB of LDS a chunk needs: 4 blocks = 32 waves per CU (256 threads: 14.9 ms per 6144 images, 512: 11.5, 1024: 15.5)
// ALLKEYS (convention C.9, variant 1 -- OpenCV >= 3.3): the key of EVERY pixel with x < Ws - 1, y < Hs - 1, defined or not, at its raster position
// y * (Ws - 1) + x of `keys` -- the vector ll_angle hands to std::sort; lsd_seedsort.hip replays that sort and writes the seed list and its
// length, so the compacted emission below is skipped.
// The bin of a pixel, int(sqrt(n / 4.0) * bin_coef) in double: decided in float wherever the float product is farther than 10^-3 from an integer (its error is
// below 2.6 x 10^-4: one ulp of v_sqrt_f32, the coefficient's and the product's rounding, t <= 1023), the reference's double expression only for the rest
__device__ __forceinline__ int lsd_bin(int n, double bin_coef, float bin_coef_half_f)
{
    const float tf = __builtin_amdgcn_sqrtf((float)n) * bin_coef_half_f;
    const int b = (int)tf;
    const float fr = tf - (float)b;
    if (n != 0 && (fr < 1e-3f || fr > 0.999f)) return (int)(sqrt_quarter(n) * bin_coef);
    return b;
}

// CH: pixels per block (CH for the compacted emission, whose chunk counts come from k_lsd_grad; ALLKEYS may take larger chunks: the two halo rows a
// block evaluates on top of its chunk are 73 % extra at 4096 pixels and a 1490-pixel row, 36 % at 8192)
// WIDE (lsd_wide.hip): lsd_n_bins > 1024 or a working image of 2^22 pixels and more -- the key is the 64-bit word (n_bins - 1 - bin) << 32 | address (the key
// buffers hold 8 bytes per pixel then), the bin is the reference's double expression throughout (the float shortcut's error bound assumes bins below 1024) and
// the row of a pixel comes from a real division (the multiply-shift pair is exact below 2^22 only)
template <bool OWNER, bool ALLKEYS, int CH, bool WIDE = false>
__global__ __launch_bounds__(KEYS_THREADS) void k_lsd_keys(uint32_t* __restrict__ gradAll, const LineGeom* __restrict__ gp,
                                                  const int* __restrict__ maxN, const int* __restrict__ chunkCnt, uint32_t* __restrict__ keys,
                                                  int* __restrict__ keyCount, uint32_t* __restrict__ owner, const float* __restrict__ angDeg,
                                                  int nChunks, int total)
{
    constexpr int NWV = KEYS_THREADS / 64, SPAN = CH / NWV;
    extern __shared__ float s_deg[];           // [CH + 2 * Ws + 2]: image positions c0 - Ws - 1 .. c0 + CH + Ws (NaN outside the image)
    __shared__ uint16_t s_list[CH];      // the chunk's defined pixels (offset in the chunk), per wave quarter, raster order
    __shared__ int s_wcnt[NWV], s_base;
    const LineGeom& g = *gp;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // XCD-aware numbering: blocks L, L + 8, L + 16, ... (one XCD) take consecutive (image, chunk) pairs
    const int L = blockIdx.x, per = total >> 3;
    const int V = L < per * 8 ? (L & 7) * per + (L >> 3) : L;
    const int img = V / nChunks, chunk = V - img * nChunks;
    // (fields of *gp used inside the loops are copied out: the compiler reloads them behind every store otherwise)
    const int Ws = g.Ws, Hs = g.Hs, Ps = g.Ps, nBins1 = g.nBins - 1;
    const uint32_t divM = g.divWsM, divS = g.divWsS;
    const float alignDeg = g.alignDeg;
    const double precG = g.prec;
    uint32_t* grad = gradAll + (size_t)img * Ps;
    const int c0 = chunk * CH;
    const int lo = c0 - Ws - 1, hi = c0 + CH + Ws + 1;      // (may leave the image at either end: NaN there)
    const float kNaN = __builtin_nanf("");
    if (threadIdx.x == 0) s_base = 0;
    const double max_grad = sqrt((double)maxN[img * 32] / 4.0);
    const double bin_coef = (max_grad > 0) ? (double)nBins1 / max_grad : 0;
    const float bin_coef_half_f = 0.5f * (float)bin_coef;
    // (NB independent loads in flight per thread, then their table lookups: the pass is latency bound otherwise)
    constexpr int NB = 8;
    // ALLKEYS with a chunk of exactly NB x KEYS_THREADS pixels (the batch's form): the chunk's own pixels -- thread t takes c0 + t + 512 u, no position tests but the
    // image's end -- with their keys, then the two halo rows (angles only).  One predicate per key (inside the image's key area and not in the last column), one
    // branch for the rare double-precision bin: 25 % fewer instructions than the general loop below, which tests every element against the chunk and the area
    // with nested branches (1.30 M -> instructions per image in profiles/r6z_*; the kernel is issue bound)
    constexpr bool SPLIT = ALLKEYS && !WIDE && CH == NB * KEYS_THREADS;
    if (SPLIT) {
        uint32_t* kall = keys + (size_t)img * Ps;
        const int keyEnd = min(Ps, (Hs - 1) * Ws);          // y < Hs - 1  <=>  idx < (Hs - 1) Ws
        {
            uint32_t p[NB];
            float d[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) { const int idx = c0 + (int)threadIdx.x + u * KEYS_THREADS; p[u] = idx < Ps ? grad[idx] : kNotDef; }
#pragma unroll
            for (int u = 0; u < NB; ++u) d[u] = (p[u] & kNotDef) ? kNaN : angDeg[p[u] & 0x3fffffu];
#pragma unroll
            for (int u = 0; u < NB; ++u) s_deg[Ws + 1 + (int)threadIdx.x + u * KEYS_THREADS] = d[u];      // (position idx - lo)
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int idx = c0 + (int)threadIdx.x + u * KEYS_THREADS;
                const int y = (int)(__umulhi((uint32_t)idx, divM) >> divS), x = idx - y * Ws;
                const int gx = unpack_gx(p[u]), gy = unpack_gy(p[u]);
                const int n = gx * gx + gy * gy;
                const float tf = __builtin_amdgcn_sqrtf((float)n) * bin_coef_half_f;
                int bin = (int)tf;
                const float fr = tf - (float)bin;
                const bool valid = (idx < keyEnd) & (x < Ws - 1);
                if (valid & (n != 0) & ((fr < 1e-3f) | (fr > 0.999f))) bin = (int)(sqrt_quarter(n) * bin_coef);      // (lsd_bin: the reference's double expression near an integer)
                if (valid) kall[(uint32_t)(idx - y)] = ((uint32_t)(nBins1 - bin) << 22) | (uint32_t)idx;             // (y (Ws - 1) + x = idx - y)
            }
        }
        // the row (+ 1 pixel) above the chunk, then the one below: s_deg positions h and CH + Ws + 1 + h
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const int first = side ? c0 + CH : lo, qBase = side ? CH + Ws + 1 : 0;
            constexpr int NH = 4;
            for (int h0 = (int)threadIdx.x; h0 < Ws + 1; h0 += NH * KEYS_THREADS) {
                uint32_t p[NH];
                float d[NH];
#pragma unroll
                for (int u = 0; u < NH; ++u) { const int h = h0 + u * KEYS_THREADS, idx = first + h; p[u] = (h < Ws + 1) & ((unsigned)idx < (unsigned)Ps) ? grad[idx] : kNotDef; }
#pragma unroll
                for (int u = 0; u < NH; ++u) d[u] = (p[u] & kNotDef) ? kNaN : angDeg[p[u] & 0x3fffffu];
#pragma unroll
                for (int u = 0; u < NH; ++u) if (h0 + u * KEYS_THREADS < Ws + 1) s_deg[qBase + h0 + u * KEYS_THREADS] = d[u];
            }
        }
    } else
    for (int i0 = lo + (int)threadIdx.x; i0 < hi; i0 += NB * KEYS_THREADS) {
        uint32_t p[NB];
        float d[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) { const int idx = i0 + u * KEYS_THREADS; p[u] = (unsigned)idx < (unsigned)Ps && idx < hi ? grad[idx] : kNotDef; }
#pragma unroll
        for (int u = 0; u < NB; ++u) d[u] = (p[u] & kNotDef) ? kNaN : angDeg[p[u] & 0x3fffffu];      // fastAtan2(gx, -gy), tabulated per context
#pragma unroll
        for (int u = 0; u < NB; ++u) if (i0 + u * KEYS_THREADS < hi) s_deg[i0 + u * KEYS_THREADS - lo] = d[u];
        if (ALLKEYS) {
            // the std::sort key of every pixel of the chunk itself (not of the halo rows), while its gradient word is in a register: a second pass over
            // the chunk would load every word again, one dependent L2 round trip per pixel and thread (16.5 -> 12 ms per 6144 images)
            uint32_t* kall = keys + (size_t)img * Ps;
            unsigned long long* kallW = reinterpret_cast<unsigned long long*>(keys) + (size_t)img * Ps;
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int idx = i0 + u * KEYS_THREADS;
                if (idx >= c0 && idx < min(Ps, c0 + CH)) {
                    const int y = WIDE ? idx / Ws : (int)(__umulhi((uint32_t)idx, divM) >> divS), x = idx - y * Ws;
                    if (x < Ws - 1 && y < Hs - 1) {
                        const int gx = unpack_gx(p[u]), gy = unpack_gy(p[u]);
                        if (WIDE) {
                            const int bin = (int)(sqrt_quarter(gx * gx + gy * gy) * bin_coef);
                            kallW[(size_t)y * (Ws - 1) + x] = ((unsigned long long)(uint32_t)(nBins1 - bin) << 32) | (unsigned long long)(uint32_t)idx;
                        } else {
                        const int bin = lsd_bin(gx * gx + gy * gy, bin_coef, bin_coef_half_f);
                        kall[y * (Ws - 1) + x] = ((uint32_t)(nBins1 - bin) << 22) | (uint32_t)idx;
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (!ALLKEYS) {   // keys of the chunks before this one
        int part = 0;
        for (int c = threadIdx.x; c < chunk; c += KEYS_THREADS) part += chunkCnt[(size_t)img * nChunks + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0 && part) atomicAdd(&s_base, part);
    }
    // the defined pixels of the chunk, compacted in raster order: every wave counts its eighth, then writes it behind the waves before it (one dense list:
    // the loop below then reads entry t, no search for the wave a position belongs to)
    unsigned long long defM[SPAN / 64];
    {
        int wc = 0;
#pragma unroll
        for (int k = 0; k < SPAN / 64; ++k) {
            const float dv = s_deg[c0 + wv * SPAN + k * 64 + lane - lo];
            defM[k] = wave_vote(dv == dv);          // (not NaN; positions beyond the image are NaN too)
            wc += __popcll(defM[k]);
        }
        if (lane == 0) s_wcnt[wv] = wc;
    }
    __syncthreads();
    int n3 = 0, wbase = 0;
#pragma unroll
    for (int v = 0; v < NWV; ++v) { const int cv = s_wcnt[v]; if (v < wv) wbase += cv; n3 += cv; }
#pragma unroll
    for (int k = 0; k < SPAN / 64; ++k) {
        if (wave_bit(defM[k])) s_list[wbase + wave_rank_below(defM[k])] = (uint16_t)(wv * SPAN + k * 64 + lane);
        wbase += __popcll(defM[k]);
    }
    __syncthreads();
    uint32_t* kout = keys + (size_t)img * Ps + s_base;
    unsigned long long* koutW = reinterpret_cast<unsigned long long*>(keys) + (size_t)img * Ps + s_base;
    // (ALLKEYS: a per-block table of the undefined pixels' bins -- gx^2 + gy^2 < nThr -- with the defined ones keyed by the dense loop below was
    // slower, 22.8 against 16.9 ms per 6144 images: the dense loop's stores then scatter and nearly every wave still holds a defined pixel)
    // dense over the defined pixels: bin -> key, and the isolated-seed test against the neighbours' angles in LDS
    for (int t = threadIdx.x; t < n3; t += KEYS_THREADS) {
        const int li = s_list[t];
        const int idx = c0 + li;
        uint32_t p = 0;
        int bin = 0;
        if (!ALLKEYS) {      // (ALLKEYS: the key went out with the first pass; the word is only needed again for the rare isolated seed)
            p = grad[idx];
            const int gx = unpack_gx(p), gy = unpack_gy(p);
            bin = (int)(sqrt_quarter(gx * gx + gy * gy) * bin_coef);
        }
        const float* sd = s_deg + (idx - lo);
        const float deg0 = sd[0];
        // isaligned() on two angles in degrees a, b: with t = | |a - b| - 180 | it is t >= 180 - ang_th (|a - b| <= ang_th, or >= 360 - ang_th after the
        // wrap), i.e. d = t - alignDeg >= 0; an undefined or outside neighbour gives NaN.  Decided in float wherever d is farther than 10^-3 degrees from 0
        // (the float differences are good to 10^-4, the reference's double radians to 10^-13); the exact double form only for the rest
        float dmax = -1.f, margin = 1.0f;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (q == 4) continue;
            const float dn = sd[((q / 3) - 1) * Ws + (q % 3) - 1];
            const float dd = f_sub(fabsf(f_sub(fabsf(f_sub(deg0, dn)), 180.f)), alignDeg);
            dmax = fmaxf(dmax, dd);
            margin = fminf(margin, fabsf(dd));
        }
        bool iso = !(dmax >= 0.f);
        if (margin < 1e-3f || alignDeg < 0.f) {
            const double a0 = d_mul((double)deg0, kDegToRads);
            iso = true;
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                if (q == 4) continue;
                const float dn = sd[((q / 3) - 1) * Ws + (q % 3) - 1];
                if (dn != dn) continue;
                double n_theta = d_sub(a0, d_mul((double)dn, kDegToRads));
                if (n_theta < 0) n_theta = -n_theta;
                if (n_theta > kM32PI) { n_theta = d_sub(n_theta, kM2PI); if (n_theta < 0) n_theta = -n_theta; }
                if (n_theta <= precG) iso = false;
            }
        }
        if (iso) grad[idx] = (ALLKEYS ? grad[idx] : p) | kIso;       // other blocks only read the NOTDEF bit and the gradient pair of this word
        if (OWNER) owner[(size_t)img * Ps + idx] = 0xffffffffu;       // nobody has claimed the pixel (multi-wave growth, lsd_grow.hip)
        if (!ALLKEYS) { if (WIDE) koutW[t] = ((unsigned long long)(uint32_t)(nBins1 - bin) << 32) | (unsigned long long)(uint32_t)idx;
                        else kout[t] = ((uint32_t)(nBins1 - bin) << 22) | (uint32_t)idx; }
    }
    if (!ALLKEYS && chunk == nChunks - 1 && threadIdx.x == 0) keyCount[img * 32] = s_base + n3;
}

// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ double grad_angle(int gx, int gy) { return d_mul((double)dev_fastAtan2((float)gx, (float)(-gy)), kDegToRads); }

__device__ __forceinline__ bool is_aligned(double a, double theta, double prec)
{
    // |x| instead of "if (x < 0) x = -x": identical for every x that reaches a comparison (only the sign of a zero could differ), and the
    // absolute value folds into the next instruction's source modifier
    double n_theta = fabs(d_sub(theta, a));
    if (n_theta > kM32PI) n_theta = fabs(d_sub(n_theta, kM2PI));
    return n_theta <= prec;
}

// sin/cos of a double in [0, 2*pi + eps] (all the path ever asks for): Cody-Waite reduction by pi/2 and the fdlibm
// kernel polynomials, < 1 ulp like ocml's sincos but a third of the instructions (no large-argument path).  The
// reference evaluates libm's double cos/sin here and immediately rounds the float accumulation, so any sub-ulp
// accurate double result gives the same float except when it straddles a rounding boundary.
__device__ __forceinline__ void sincos_2pi(double x, double* sn, double* cs)
{
    const int k = (int)__fma_rn(x, 0.63661977236758134308, 0.5);
    double r = __fma_rn(-(double)k, 1.57079632679489655800e+00, x);
    r = __fma_rn(-(double)k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = __fma_rn(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __fma_rn(z, ps, 2.75573137070700676789e-06);
    ps = __fma_rn(z, ps, -1.98412698298579493134e-04);
    ps = __fma_rn(z, ps, 8.33333333332248946124e-03);
    ps = __fma_rn(z, ps, -1.66666666666666324348e-01);
    const double s0 = __fma_rn(r * z, ps, r);
    double pc = __fma_rn(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __fma_rn(z, pc, -2.75573143513906633035e-07);
    pc = __fma_rn(z, pc, 2.48015872894767294178e-05);
    pc = __fma_rn(z, pc, -1.38888888888741095749e-03);
    pc = __fma_rn(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double c0 = w + (((1.0 - w) - hz) + z * z * pc);
    const bool sw = k & 1;
    double so = sw ? c0 : s0, co = sw ? s0 : c0;
    if (k & 2) so = -so;
    if ((k + 1) & 2) co = -co;
    *sn = so; *cs = co;
}

// The level-line angle of a pixel and its cos / sin are pure functions of the integer gradient pair, so they are tabulated once per
// context over the packed 22-bit (gx:11 | gy:11) field of the gradient word -- 84 MB, image independent, shared by every agent; real
// images touch a small, L2 / Infinity-Cache resident part of it.  This takes fastAtan2 and the double sincos out of the agent's
// per-iteration instruction stream (the agent is VALU-issue bound), at the price of one more dependent load.
__global__ __launch_bounds__(256) void k_lsd_angle_table(float* __restrict__ angDeg, AngEnt* __restrict__ ent, int libmFloat)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const int gx = unpack_gx(i), gy = unpack_gy(i);
    const float deg = dev_fastAtan2((float)gx, (float)(-gy));
    const double ang = d_mul((double)deg, kDegToRads);
    double sn, cs;
    sincos_2pi((double)(float)ang, &sn, &cs);
    // convention C.6, float overloads: what region_grow adds is cosf / sinf of the float angle (glibc's, bit for bit: device_math.hpp); a float added
    // to the float sum is the same as its double added and rounded once, so the entry keeps its layout
    if (libmFloat) { cs = (double)glibc_cosf((float)ang); sn = (double)glibc_sinf((float)ang); }
    angDeg[i] = deg;
    // region_grow starts its sums at float(cos(reg_angle)), float(sin(reg_angle)) with the seed's angle as a double
    double s0, c0;
    sincos_2pi(ang, &s0, &c0);
    AngEnt e; e.cs = cs; e.sn = sn; e.ang = ang; e.seed = make_float2((float)c0, (float)s0);
    ent[i] = e;
}

int launch_lsd_angle_table(LineDeviceBufs& b, int libmFloat, hipStream_t s)
{
    hipLaunchKernelGGL(k_lsd_angle_table, dim3((1u << 22) / 256), dim3(256), 0, s, b.angDeg, reinterpret_cast<AngEnt*>(b.angEnt), libmFloat);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}


// debug / test: fdiv_unscaled against the compiler's IEEE division on pseudo-random operand pairs 0 <= a <= b drawn from the agent's
// operand range (sums of up to ~10^5 unit vectors, plus tiny and equal operands); counts the bit mismatches
__global__ __launch_bounds__(256) void k_fdiv_sweep(unsigned long long seed, int per_thread, unsigned long long* __restrict__ mismatches)
{
    unsigned long long st = seed ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(blockIdx.x * 256 + threadIdx.x + 1));
    unsigned long long bad = 0;
    for (int i = 0; i < per_thread; ++i) {
        st ^= st >> 12; st ^= st << 25; st ^= st >> 27;
        const unsigned long long r = st * 0x2545F4914F6CDD1Dull;
        // exponents: b in [2^-40, 2^17), a = b * u with u in [0, 1] at several granularities (random mantissas)
        const int eb = (int)(r & 63) - 45;                                   // -45 .. 18
        float b = ldexpf(1.0f + (float)((r >> 8) & 0x7fffff) * 1.1920929e-7f, eb);
        float a = ldexpf(1.0f + (float)((r >> 32) & 0x7fffff) * 1.1920929e-7f, eb - (int)((r >> 56) & 31));
        if (((r >> 61) & 7) == 0) a = b;
        if (((r >> 61) & 7) == 1) a = 0.f;
        if (a > b) a = b;
        b = __fadd_rn(b, (float)2.2204460492503131e-16);                    // the "+ eps" of fastAtan2's denominator
        const float q0 = __fdiv_rn(a, b), q1 = fdiv_unscaled(a, b);
        bad += __float_as_uint(q0) != __float_as_uint(q1);
    }
    if (bad) atomicAdd(mismatches, bad);
}

// debug / test: sqrt_quarter against the compiler's sqrt, and lsd_bin against the double expression, on every n in [0, count)
__global__ __launch_bounds__(256) void k_sqrtq_sweep(int count, unsigned long long* __restrict__ mismatches)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= count) return;
    const double a = sqrt((double)n / 4.0), b = sqrt_quarter(n);
    if (__double_as_longlong(a) != __double_as_longlong(b)) atomicAdd(mismatches, 1ull);
    // ... and the float-first bin of k_lsd_keys against the reference's double expression, for 97 image maxima between 36 and the largest norm a pair of
    // 8-bit differences can have (every n that can occur under each of them)
    int bad = 0;
    for (int c = 0; c < 97; ++c) {
        const int mN = 36 + c * 5418;      // ... 520 164 (gx, gy in [-510, 510])
        if (n > mN) continue;
        const double bin_coef = 1023.0 / sqrt((double)mN / 4.0);
        bad += lsd_bin(n, bin_coef, 0.5f * (float)bin_coef) != (int)(a * bin_coef);
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

// debug / test: the growth agent's cheap alignment test (k_lsd_grow, PF bit 16) against the reference's expression.  Every thread takes pseudo-random table
// entries (the candidate pixel: angle a, direction d) and float sums S -- lengths from 0.6 to 6000, directions concentrated within 3 mrad of a +- prec, where
// the two tests can disagree, the rest uniform -- and counts: out[0] decisions the cheap test called certain that differ from
// |fastAtan2(S) * DEG2RAD - a| (wrapped) <= prec, out[1] decisions it left to the reference's expression, out[2] all.
__global__ __launch_bounds__(256) void k_align_sweep(const LineGeom* __restrict__ gp, const AngEnt* __restrict__ ent, unsigned long long seed, int per_thread,
                                                     unsigned long long* __restrict__ out)
{
    const LineGeom& g = *gp;
    const float tanLo = g.alignTanLo, tanHi = g.alignTanHi;
    const double prec = g.prec, precWrap = g.precWrap;
    unsigned long long x = seed ^ ((unsigned long long)(blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull);
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    next(); next();
    unsigned long long bad = 0, und = 0;
    for (int k = 0; k < per_thread; ++k) {
        const unsigned long long r = next(), r2 = next();
        // a gradient pair as images have them (|g| <= 510; small gradients are the common ones)
        const int span = (r & 3) ? 40 : 510;
        const int gx = (int)((r >> 2) % (2 * span + 1)) - span, gy = (int)((r >> 24) % (2 * span + 1)) - span;
        const AngEnt e = ent[pack_g(gx, gy)];
        const double u = (double)(r2 & 0xfffff) / 1048576.0;               // [0, 1)
        const double len = 0.6 * exp(9.2 * (double)((r2 >> 20) & 0xfffff) / 1048576.0);
        double th;
        if ((r2 >> 40) & 3) th = e.ang + (((r2 >> 42) & 1) ? prec : -prec) + (u - 0.5) * 6e-3;
        else th = u * kM2PI;
        const float sx = (float)(len * cos(th)), sy = (float)(len * sin(th));
        // the reference: region angle = fastAtan2 of the float sums, compared in double
        const double reg_angle = d_mul((double)agent_fastAtan2(sy, sx), kDegToRads);
        const double nth = fabs(d_sub(reg_angle, e.ang));
        const bool exact = nth <= prec || nth >= precWrap;
        const float dot = __fmaf_rn(sy, e.seed.y, __fmul_rn(sx, e.seed.x));
        const float crs = __fmaf_rn(sx, e.seed.y, -__fmul_rn(sy, e.seed.x));
        const bool sa = fabsf(crs) <= __fmaf_rn(tanLo, dot, -1e-4f), sn = fabsf(crs) >= __fmaf_rn(tanHi, dot, 1e-4f);
        if (sa && sn) ++bad;
        else if (sa || sn) bad += sa != exact;
        else ++und;
    }
    atomicAdd(out, bad); atomicAdd(out + 1, und); atomicAdd(out + 2, (unsigned long long)per_thread);
}

int launch_align_sweep(const LineDeviceBufs& b, unsigned long long seed, int blocks, int per_thread, unsigned long long* d_out, hipStream_t s)
{
    hipLaunchKernelGGL(k_align_sweep, dim3(blocks), dim3(256), 0, s, b.geom, reinterpret_cast<const AngEnt*>(b.angEnt), seed, per_thread, d_out);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

int launch_sqrtq_sweep(int count, unsigned long long* d_mismatches, hipStream_t s)
{
    hipLaunchKernelGGL(k_sqrtq_sweep, dim3((count + 255) / 256), dim3(256), 0, s, count, d_mismatches);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

int launch_fdiv_sweep(unsigned long long seed, int blocks, int per_thread, unsigned long long* d_mismatches, hipStream_t s)
{
    hipLaunchKernelGGL(k_fdiv_sweep, dim3(blocks), dim3(256), 0, s, seed, per_thread, d_mismatches);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

struct SegCand { float e0, e1, e2, e3, length; int keep; };

// LSDDetectorC::detectImpl's clamping of the Vec4f end points and the length filter (LSDDetector_custom.cpp:270-289), from the rectangle's end points
__device__ __forceinline__ SegCand rect_to_cand(const LineGeom& g, double x1, double y1, double x2, double y2)
{
    x1 = d_add(x1, 0.5); y1 = d_add(y1, 0.5); x2 = d_add(x2, 0.5); y2 = d_add(y2, 0.5);
    if (g.scale != 1) { x1 = x1 / g.scale; y1 = y1 / g.scale; x2 = x2 / g.scale; y2 = y2 / g.scale; }
    float e0 = (float)x1, e1 = (float)y1, e2 = (float)x2, e3 = (float)y2;
    const int cols = g.W, rows = g.H;
    if (e0 < 0) e0 = 0;
    if (e0 >= cols) e0 = (float)cols - 1.0f;
    if (e2 < 0) e2 = 0;
    if (e2 >= cols) e2 = (float)cols - 1.0f;
    if (e1 < 0) e1 = 0;
    if (e1 >= rows) e1 = (float)rows - 1.0f;
    if (e3 < 0) e3 = 0;
    if (e3 >= rows) e3 = (float)rows - 1.0f;
    const double dxe = (double)f_sub(e0, e2), dye = (double)f_sub(e1, e3);
    const double length = (double)(float)sqrt(d_add(d_mul(dxe, dxe), d_mul(dye, dye)));
    SegCand c; c.e0 = e0; c.e1 = e1; c.e2 = e2; c.e3 = e3; c.length = (float)length; c.keep = length > g.minLength;
    return c;
}

// ---- LSD_REFINE_STD inside the agent (cv LineSegmentDetectorImpl::refine / reduce_region_radius, convention C.14; oracle/line_oracle.cpp).
// The region's pixel log is (x | y << 16, gradient word) pairs in growth order; every lane runs the same scalar loops over it (uniform addresses:
// one request per load), so the sums are formed in exactly the reference's order.
struct AgentRect { double x1, y1, x2, y2, width, theta, dx, dy, prec, p; };

__device__ __forceinline__ double log_modgrad(uint32_t w) { const int gx = unpack_gx(w), gy = unpack_gy(w); return sqrt((double)(gx * gx + gy * gy) / 4.0); }

__device__ __forceinline__ AgentRect agent_region2rect(const uint2* lg, int n, double reg_angle, double prec)
{
    AgentRect rec;
    double x = 0, y = 0, sum = 0;
    for (int q = 0; q < n; ++q) {
        const uint2 e = lg[q];
        const double wt = log_modgrad(e.y);
        x = d_add(x, d_mul((double)(int)(e.x & 0xffffu), wt));
        y = d_add(y, d_mul((double)(int)(e.x >> 16), wt));
        sum = d_add(sum, wt);
    }
    x = x / sum; y = y / sum;
    double Ixx = 0, Iyy = 0, Ixy = 0;
    for (int q = 0; q < n; ++q) {
        const uint2 e = lg[q];
        const double wt = log_modgrad(e.y);
        const double ex = d_sub((double)(int)(e.x & 0xffffu), x), ey = d_sub((double)(int)(e.x >> 16), y);
        Ixx = d_add(Ixx, d_mul(d_mul(ey, ey), wt));
        Iyy = d_add(Iyy, d_mul(d_mul(ex, ex), wt));
        Ixy = d_sub(Ixy, d_mul(d_mul(ex, ey), wt));
    }
    const double dI = d_sub(Ixx, Iyy);
    const double lambda = d_mul(0.5, d_sub(d_add(Ixx, Iyy), sqrt(d_add(d_mul(dI, dI), d_mul(d_mul(4.0, Ixy), Ixy)))));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)dev_fastAtan2((float)d_sub(lambda, Ixx), (float)Ixy)
                                           : (double)dev_fastAtan2((float)Ixy, (float)d_sub(lambda, Iyy));
    theta = d_mul(theta, kDegToRads);
    {
        double diff = d_sub(theta, reg_angle);
        while (diff <= -kPI) diff = d_add(diff, kM2PI);
        while (diff > kPI) diff = d_sub(diff, kM2PI);
        if (fabs(diff) > prec) theta = d_add(theta, kPI);
    }
    double ddx, ddy;
    sincos(theta, &ddy, &ddx);
    double l_min = 0, l_max = 0, w_min = 0, w_max = 0;
    for (int q = 0; q < n; ++q) {
        const uint32_t pk = lg[q].x;
        const double rdx = d_sub((double)(int)(pk & 0xffffu), x), rdy = d_sub((double)(int)(pk >> 16), y);
        const double lq = d_add(d_mul(rdx, ddx), d_mul(rdy, ddy));
        const double wq = d_add(d_mul(-rdx, ddy), d_mul(rdy, ddx));
        l_max = fmax(l_max, lq); l_min = fmin(l_min, lq);      // ("if (l > l_max) .. else if (l < l_min) .." from (0, 0))
        w_max = fmax(w_max, wq); w_min = fmin(w_min, wq);
    }
    rec.x1 = d_add(x, d_mul(l_min, ddx)); rec.y1 = d_add(y, d_mul(l_min, ddy));
    rec.x2 = d_add(x, d_mul(l_max, ddx)); rec.y2 = d_add(y, d_mul(l_max, ddy));
    rec.width = d_sub(w_max, w_min);
    if (rec.width < 1.0) rec.width = 1.0;
    rec.theta = theta; rec.dx = ddx; rec.dy = ddy; rec.prec = prec; rec.p = 0;      // (p: set by the caller, only the NFA stage reads it)
    return rec;
}

// ---- LSD_REFINE_ADV: rect_improve / rect_nfa / nfa / log_gamma (convention C.14; oracle/line_oracle.cpp).  The numbers of false alarms are
// compared with each other and with log_eps only; their libm calls (log, exp, pow, sinh, log10) are the device library's, so a decision can differ
// from the host's where two of these doubles are closer than the libraries' last-bit differences.
__device__ __forceinline__ double agent_log_gamma(double x)
{
    if (x > 15.0) return 0.918938533204673 + (x - 0.5) * log(x) - x + 0.5 * x * log(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
    const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
    double a = (x + 0.5) * log(x + 5.5) - (x + 5.5);
    double b = 0;
    for (int n = 0; n < 7; ++n) {
        a -= log(x + (double)n);
        b += q[n] * pow(x, (double)n);
    }
    return a + log(b);
}

__device__ __forceinline__ double agent_nfa(int n, int k, double p, double LOG_NT)
{
    if (n == 0 || k == 0) return -LOG_NT;
    if (n == k) return -LOG_NT - (double)n * log10(p);
    const double p_term = p / (1 - p);
    const double log1term = agent_log_gamma((double)n + 1) - agent_log_gamma((double)k + 1) - agent_log_gamma((double)(n - k) + 1) + (double)k * log(p) +
                            (double)(n - k) * log(1.0 - p);
    double term = exp(log1term);
    {   // double_equal(term, 0)
        bool eq = term == 0.0;
        if (!eq) { double am = fabs(term); if (am < 2.2250738585072014e-308) am = 2.2250738585072014e-308; eq = (fabs(term) / am) <= (100.0 * 2.2204460492503131e-16); }
        if (eq) {
            if (k > n * p) return -log1term / 2.30258509299404568402 - LOG_NT;
            return -LOG_NT;
        }
    }
    double bin_tail = term;
    for (int i = k + 1; i <= n; ++i) {
        const double bin_term = (double)(n - i + 1) / (double)i;
        const double mult_term = bin_term * p_term;
        term *= mult_term;
        bin_tail += term;
        if (bin_term < 1) {
            const double err = term * ((1 - pow(mult_term, (double)(n - i + 1))) / (1 - mult_term) - 1);
            if (err < 0.1 * fabs(-log10(bin_tail) - LOG_NT) * bin_tail) break;
        }
    }
    return -log10(bin_tail) - LOG_NT;
}

// rect_nfa: the rectangle's corners (truncated to int), ordered by x then y; rows from the lowest to the highest corner, the left and right ends of a
// row advance by the edge steps (integer divisions, and tailp's x where its y is meant: as in the original).  The pixels of a row are counted across the
// lanes (the counts do not depend on the order).
__device__ __forceinline__ double agent_rect_nfa(const LineGeom& g, const uint32_t* __restrict__ grad, const AngEnt* __restrict__ ent, const AgentRect& rec, int lane)
{
    const double half_width = rec.width / 2.0;
    const double dyhw = rec.dy * half_width, dxhw = rec.dx * half_width;
    int ex[4], ey[4];
    ex[0] = (int)(rec.x1 - dyhw); ey[0] = (int)(rec.y1 + dxhw);
    ex[1] = (int)(rec.x2 - dyhw); ey[1] = (int)(rec.y2 + dxhw);
    ex[2] = (int)(rec.x2 + dyhw); ey[2] = (int)(rec.y2 - dxhw);
    ex[3] = (int)(rec.x1 + dyhw); ey[3] = (int)(rec.y1 - dxhw);
    // std::sort of four (x, y) pairs, x ascending then y ascending (equal pairs are indistinguishable): insertion sort
    for (int i = 1; i < 4; ++i)
        for (int j = i; j > 0 && (ex[j] == ex[j - 1] ? ey[j] < ey[j - 1] : ex[j] < ex[j - 1]); --j) {
            const int tx = ex[j], ty = ey[j]; ex[j] = ex[j - 1]; ey[j] = ey[j - 1]; ex[j - 1] = tx; ey[j - 1] = ty;
        }
    int imin = 0, imax = 0;
    for (int i = 1; i < 4; ++i) {
        if (ey[imin] > ey[i]) imin = i;
        if (ey[imax] < ey[i]) imax = i;
    }
    unsigned taken = 1u << imin;
    int il = -1;
    for (int i = 0; i < 4; ++i)
        if (!((taken >> i) & 1u)) { if (il < 0) il = i; else if (ex[il] > ex[i]) il = i; }
    taken |= 1u << il;
    int ir = -1;
    for (int i = 0; i < 4; ++i)
        if (!((taken >> i) & 1u)) { if (ir < 0) ir = i; else if (ex[ir] < ex[i]) ir = i; }
    taken |= 1u << ir;
    int it = -1;
    for (int i = 0; i < 4; ++i)
        if (!((taken >> i) & 1u)) { if (it < 0) it = i; else if (ex[it] > ex[i]) it = i; }
    const int myx = ex[imin], myy = ey[imin], lx = ex[il], ly = ey[il], rx = ex[ir], ry = ey[ir], tx = ex[it];
    const double flstep = (myy != ly) ? (double)((myx - lx) / (myy - ly)) : 0;
    const double slstep = (ly != tx) ? (double)((lx - tx) / (ly - tx)) : 0;
    const double frstep = (myy != ry) ? (double)((myx - rx) / (myy - ry)) : 0;
    const double srstep = (ry != tx) ? (double)((rx - tx) / (ry - tx)) : 0;
    double lstep = flstep, rstep = frstep;
    double left_x = myx, right_x = myx;
    int total_pts = 0, alg_pts = 0;
    const int Ws = g.Ws, Hs = g.Hs;
    for (int y = myy; y <= ey[imax]; ++y) {
        if (y < 0 || y >= Hs) continue;
        const int xa = (int)left_x, xb = (int)right_x;
        for (int x0 = xa; x0 <= xb; x0 += 64) {
            const int x = x0 + lane;
            const bool in = x <= xb && x >= 0 && x < Ws;
            bool al = false;
            if (in) {
                const uint32_t w = grad[y * Ws + x];
                if (!(w & kNotDef)) {
                    double nt = fabs(d_sub(rec.theta, ent[w & 0x3fffffu].ang));
                    if (nt > kM32PI) nt = fabs(d_sub(nt, kM2PI));
                    al = nt <= rec.prec;
                }
            }
            total_pts += __popcll(wave_vote(in));
            alg_pts += __popcll(wave_vote(al));
        }
        if (y >= ly) lstep = slstep;
        if (y >= ry) rstep = srstep;
        left_x += lstep;
        right_x += rstep;
    }
    return agent_nfa(total_pts, alg_pts, rec.p, g.logNT);
}

__device__ __forceinline__ double agent_rect_improve(const LineGeom& g, const uint32_t* __restrict__ grad, const AngEnt* __restrict__ ent, AgentRect& rec, int lane)
{
    const double delta = 0.5, delta_2 = delta / 2.0, LOG_EPS = g.logEps;
    double log_nfa = agent_rect_nfa(g, grad, ent, rec, lane);
    if (log_nfa > LOG_EPS) return log_nfa;
    AgentRect r = rec;
    for (int n = 0; n < 5; ++n) {
        r.p /= 2;
        r.prec = r.p * kPI;
        const double v = agent_rect_nfa(g, grad, ent, r, lane);
        if (v > log_nfa) { log_nfa = v; rec = r; }
    }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n)
        if ((r.width - delta) >= 0.5) {
            r.width -= delta;
            const double v = agent_rect_nfa(g, grad, ent, r, lane);
            if (v > log_nfa) { rec = r; log_nfa = v; }
        }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n)
        if ((r.width - delta) >= 0.5) {
            r.x1 += -r.dy * delta_2; r.y1 += r.dx * delta_2; r.x2 += -r.dy * delta_2; r.y2 += r.dx * delta_2;
            r.width -= delta;
            const double v = agent_rect_nfa(g, grad, ent, r, lane);
            if (v > log_nfa) { rec = r; log_nfa = v; }
        }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n)
        if ((r.width - delta) >= 0.5) {
            r.x1 -= -r.dy * delta_2; r.y1 -= r.dx * delta_2; r.x2 -= -r.dy * delta_2; r.y2 -= r.dx * delta_2;
            r.width -= delta;
            const double v = agent_rect_nfa(g, grad, ent, r, lane);
            if (v > log_nfa) { rec = r; log_nfa = v; }
        }
    if (log_nfa > LOG_EPS) return log_nfa;
    r = rec;
    for (int n = 0; n < 5; ++n)
        if ((r.width - delta) >= 0.5) {
            r.p /= 2;
            r.prec = r.p * kPI;
            const double v = agent_rect_nfa(g, grad, ent, r, lane);
            if (v > log_nfa) { rec = r; log_nfa = v; }
        }
    return log_nfa;
}

__device__ __forceinline__ double agent_dist(double x1, double y1, double x2, double y2)
{
    const double ax = d_sub(x2, x1), ay = d_sub(y2, y1);
    return sqrt(d_add(d_mul(ax, ax), d_mul(ay, ay)));
}
__device__ __forceinline__ double agent_density(int n, const AgentRect& r) { return (double)n / d_mul(agent_dist(r.x1, r.y1, r.x2, r.y2), r.width); }

constexpr int RING = 256;    // FIFO window of the growing region kept in LDS (LDS is kept small: 24 agents share a CU with other kernels)
constexpr int PEND = 1024;   // hash table of pixels whose USED store may not be visible to a load yet (512 entries: 122.1 ms per 6144 images, 1024: 121.4 --
                             // fewer collisions, fewer flushes, fewer seed windows gathered twice; 5 KB of LDS per agent)


// (single-wave workgroup: LDS operations of one wave execute in order, so __builtin_amdgcn_wave_barrier() -- a
// compiler-only barrier -- is enough between a lane-0 LDS write and the other lanes' reads; no s_barrier / vmcnt wait)
// One wave replays cv::LineSegmentDetector's seed loop for one image.  Visibility of the wave's own stores:
// FIFO entries are read back from an LDS ring, and a pixel whose USED bit was just stored is also entered in
// an LDS hash table that every `used` test consults, so no memory fence is needed per step; a fence is only
// issued when a table slot is about to be reused by a different pixel (and before region2rect).
// The agent is a long dependent chain (LDS ring -> gradient word -> angle table -> accept chain): a single wave spends ~3/4 of its time
// waiting, so throughput comes from interleaving waves.  With region2rect moved to k_lsd_rect the agent needs 64 VGPRs, i.e. up to
// 8 agents per SIMD (8192 per GPU) and room for other kernels beside them.
// REFINE (lsd_refine = LSD_REFINE_STD): every region of minRegSize pixels is fitted and, if its density is below the threshold, un-used, grown again
// under the tolerance tau derived from its angles, and shrunk (reduce_region_radius) -- all inside the seed loop, because the pixels it gives back are
// seeds and neighbours of later regions.  The agent then writes the segment candidates itself (candAll) and k_lsd_rect is not launched.
// PF (REFINE = 0 only; bits, A/B through OLF_GROW_PF): 1 = the seed windows as a software pipeline (keys two windows ahead, the windows' gradient words one
// window ahead; a flush of the pending table first folds the table into the seed masks, so no window is ever gathered twice), 2 = the rows above and
// below every live seed of a window are requested when the window starts (a region's first 3x3 gather then finds them in the cache instead of in HBM),
// 4 = the row beyond every candidate of a growth step is requested beside its table entry (the next step's gather).
// 8 = the WINDOW PHASE of a region start (round 6): the 7 x 7 pixels around the seed are gathered ONCE, one lane per pixel (word, pending-table slot, table
// entry), and the leading FIFO entries -- every entry within Chebyshev distance 2 of the seed, whose 3 x 3 lies inside the window -- are replayed from
// registers: the candidates of an entry are (3 x 3 mask << lane) & live & aligned, lane order inside the window IS the reference's raster order, an accept
// is the plain sequential step, an entry without candidates costs a handful of scalar instructions instead of a gather.  The accepted pixels are published
// in one batch (USED bits, pending table; ring and log only if the region goes on or reaches minRegSize).  The general loop takes over at the first entry
// on the window's outer ring, or when OLF_WIN_MAXPEND entries are pending (there its 8 entries per gather and its speculative rounds pay).  On the bench
// scene 10.4 k regions per image have 8.4 k non-isolated starts; 6.9 k of them end inside the window (tools/grow_region_model.py), and the general loop's
// 38.5 k iterations per image become 8.4 k window gathers + < 20 k iterations.
// 16 = the CHEAP ALIGNMENT TEST (round 6): the reference's region angle is cv::fastAtan2 of the float sums -- 33 vector instructions for a value that is only
// ever compared with the candidates' angles.  A candidate whose direction d = (cos a, sin a) (AngEnt::seed) satisfies |S x d| <= tan(prec - m) S.d is aligned
// whatever fastAtan2's approximation error (m bounds it, host_tables.cpp), one with |S x d| >= tan(prec + m) S.d is not: four multiply-adds, two more and two
// compares per lane decide all but the candidates within 0.035 degrees of the tolerance (about one decision in a thousand), and only for those is the region
// angle evaluated and the reference's double expression taken.  The sums themselves stay the reference's float chain, so the angle can be formed at any time:
// it is formed where the reference's expression is needed and at the end of a logged region (a region's first angle is its seed's own, not fastAtan2 of the seed sums).
template <int REFINE, int PF>      // REFINE 0: LSD_REFINE_NONE, 1: STD, 2: ADV
// (REFINE = 2: rect_improve / rect_nfa / nfa inlined with the AgentRect in registers need 200 VGPRs: two agents per SIMD, no scratch, no generic-pointer
// loads; as four out-of-line functions with a stack object they were 128 VGPRs + 496 B of scratch -- 16 % slower up to 2048 images, 7 % faster at 4096)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(REFINE == 2 ? 2 : 4, 8))) void k_lsd_grow(const LineGeom* __restrict__ gp, uint32_t* __restrict__ gradAll,
                                                 const uint32_t* __restrict__ keysAll, const int* __restrict__ keyCount,
                                                 uint32_t* __restrict__ regionAll, RegionRec* __restrict__ recsAll, int* __restrict__ regCount,
                                                 int* __restrict__ status, const AngEnt* __restrict__ ent, int* __restrict__ growFmt,
                                                 SegCand* __restrict__ candAll, int retry)
{
    OLF_SET_AGENT_PRIO();
    // PF bit 32 = WIDE (lsd_wide.hip: lsd_n_bins > 1024 or a working image of 2^22 pixels and more): the seed list holds plain 32-bit addresses, 2 Ps words per image
    constexpr bool WIDE = (PF & 32) != 0;
    constexpr uint32_t kAddrMask = WIDE ? 0xffffffffu : 0x3fffffu;
    __shared__ uint32_t s_ring[RING];
    __shared__ __attribute__((aligned(16))) int s_pend[PEND];
    const LineGeom& g = *gp;
    const int img = blockIdx.x, lane = threadIdx.x;
    // growFmt != nullptr: the launch behind the multi-wave kernel -- only the images that kernel gave up (-1) are grown here, and marked 1
    // (= contiguous log) for k_lsd_rect_mixed
    if (growFmt && growFmt[img] != -1) return;
    const int Ws = g.Ws, Hs = g.Hs;
    uint32_t* grad = gradAll + (size_t)img * g.Ps;
    const uint32_t* keys = keysAll + (size_t)img * g.Ps * (WIDE ? 2 : 1);
    // the pixel log: (x | y << 16, gradient word) per pixel -- k_lsd_rect needs the gradient norm of every region pixel and reads it from here
    // instead of gathering the word again
    // The log: the image's own (g.logCap entries -- every pixel, or half of them in a batch context) or, in the RETRY launch, a full-size block of the spill
    // arena.  An image whose logged regions outgrow its own log keeps running with its log writes clamped to the last entry (`over`: its regions are garbage,
    // its USED marks are not) and is marked g.spillOf[img] = -1 at the end; the retry launch behind this one clears the image's USED bits and grows it again
    // from its first seed; an exhausted arena is a capacity error (status 8) like every other fixed-capacity buffer.  The hot loop pays a compare and a
    // v_min per commit.  (Built first and removed: switching logs inside the loop -- 106 scalar registers, the agent lost its eighth wave per SIMD; leaving
    // the image with a `return` at the overflow -- exits out of the five-deep loop nest made the structurizer rebuild it with 300 more instructions.)
    uint2* reg = reinterpret_cast<uint2*>(regionAll + (size_t)img * g.regionStride);
    int logCap = g.logCap;
    if (retry) {
        if (g.spillOf[img] != -1) return;
        int blk = 0;
        if (lane == 0) blk = atomicAdd(g.spillCtl, 1);
        blk = __builtin_amdgcn_readfirstlane(blk);
        if (blk >= g.spillBlocks) { if (lane == 0) { atomicOr(status, 8); regCount[img] = 0; g.spillOf[img] = 0; } return; }
        reg = reinterpret_cast<uint2*>(g.spillArena) + (size_t)blk * g.Ps;
        logCap = g.Ps;
        if (lane == 0) g.spillOf[img] = blk + 1;
        for (int q = lane; q < g.Ps; q += 64) gradAll[(size_t)img * g.Ps + q] &= ~kUsed;      // what the first attempt marked
        __threadfence_block();
    }
    const uint32_t logLast = (uint32_t)logCap - 1u;
    bool over = false;
#ifdef OLF_NO_LOGCHECK      // (A/B builds only: the log is assumed to fit)
#define LOG_ROOM(NEED) do { } while (0)
#define LOG_AT(IDX) reg[rbase + (IDX)]
#else
#define LOG_ROOM(NEED) do { if (rbase + (NEED) > logCap) over = true; } while (0)
#define LOG_AT(IDX) reg[min((uint32_t)(rbase + (IDX)), logLast)]
#endif
    RegionRec* recs = recsAll + (size_t)img * g.maxRegions;
    const int nkeys = keyCount[img * 32];
    const double prec = g.prec, precWrap = g.precWrap;
    // (copies: read through `g` inside the seed loop they are loaded again for every region -- the stores in between may alias the geometry block)
    const int minRegSize = g.minRegSize, maxRegions = g.maxRegions;
    const uint32_t divM = g.divWsM; const int divS = g.divWsS;
// (the table is cleared with 16-byte LDS stores: 4 instructions instead of a 16-trip loop of 7 -- a flush happens 1.5 k times per image)
#define PEND_CLEAR() do { _Pragma("unroll") for (int _i = 0; _i < PEND / 256; ++_i) reinterpret_cast<int4*>(s_pend)[_i * 64 + lane] = make_int4(-1, -1, -1, -1); } while (0)
    PEND_CLEAR();
    __builtin_amdgcn_wave_barrier();
    int nreg = 0, rbase = 0;
    int flushEpoch = 0;        // number of PEND_FLUSHes so far: between two flushes every pixel this wave has marked USED is in s_pend
#ifdef OLF_STATS
    long long st_rounds = 0, st_k = 0, st_t = 0, st_full = 0, st_single = 0, st_rounds_big = 0, st_k_big = 0, st_t_big = 0;
    long long st_mem = 0, st_flush = 0, st_iters = 0, st_deep1 = 0, st_deep2 = 0, st_cand = 0, st_regions = 0;
    long long st_win = 0, st_seedl = 0, st_regl = 0, st_acc = 0, st_isos = 0, st_logged = 0, st_winLive = 0, st_first = 0, st_cand1 = 0, st_acc1 = 0;
    long long st_wentries = 0, st_whand = 0, st_wdone = 0, st_wpend = 0;
#endif
#ifdef OLF_TIMING2
    long long p_ring = 0, p_gather = 0, p_table = 0, p_chain = 0, p_commit = 0, p_n = 0, ps;
#define PSTAMP(acc) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long _t = __builtin_readcyclecounter(); acc += _t - ps; ps = _t; } while (0)
#else
#define PSTAMP(acc)
#endif
#ifdef OLF_TIMING
    long long t_seed = 0, t_small = 0, t_big = 0, t_rect = 0, n_small = 0, n_big = 0, it_small = 0, it_big = 0; long long t0 = __builtin_readcyclecounter();
#endif

#ifdef OLF_STATS
#define ST_FLUSH ++st_flush;
#else
#define ST_FLUSH
#endif
    constexpr bool PIPE = !REFINE && (PF & 1), PFSEED = !REFINE && (PF & 2), PFCAND = !REFINE && (PF & 4), WIN = !REFINE && (PF & 8), CHEAP = !REFINE && (PF & 16);

    const float tanLo = g.alignTanLo, tanHi = g.alignTanHi;      // (a tolerance too wide for the folded test -- alignTanLo < 0 -- takes the kernel without the bit: launch_lsd_grow)
    constexpr float kAlDelta = 1e-4f;      // sums shorter than this decide nothing (fastAtan2 of a vanishing vector is dominated by its epsilon)
// the lanes whose direction DIR is aligned / not aligned with the sums (SX, SY) for certain (garbage in the lanes that hold no table entry)
#define ALIGN_CHEAP(SX, SY, DIR, SA, SN) do { const float _dot = __fmaf_rn((SY), (DIR).y, __fmul_rn((SX), (DIR).x)); \
                                              const float _crs = __fmaf_rn((SX), (DIR).y, -__fmul_rn((SY), (DIR).x)); \
                                              SA = wave_vote(fabsf(_crs) <= __fmaf_rn(tanLo, _dot, -kAlDelta)); \
                                              SN = wave_vote(fabsf(_crs) >= __fmaf_rn(tanHi, _dot, kAlDelta)); } while (0)
// reg_angle as the reference has it at this point (CHEAP keeps the sums only: a region's first angle is its seed's own, every later one fastAtan2 of the sums)
#define ENSURE_ANGLE() do { if (CHEAP) reg_angle = n == 1 ? rlane_d(seedAng, l) : d_mul((double)agent_fastAtan2(sumdy, sumdx), kDegToRads); } while (0)
#ifndef OLF_WIN_MAXPEND
#define OLF_WIN_MAXPEND 16
#endif
    // window phase: lane L < 49 is pixel (seed.x + L % 7 - 3, seed.y + L / 7 - 3)
    constexpr unsigned long long kW49 = (1ull << 49) - 1ull, kSeedBit = 1ull << 24, kM3 = 7ull | (7ull << 7) | (7ull << 14);
    constexpr unsigned long long kRow5 = 0x3eull, kD2 = (kRow5 << 7) | (kRow5 << 14) | (kRow5 << 21) | (kRow5 << 28) | (kRow5 << 35);      // rows, columns 1 .. 5
    // (one register for both offsets: the agent sits at the 64-register line of eight waves per SIMD)
    const int wpk = ((lane % 7 - 3) & 0xffff) | ((lane / 7 - 3) << 16);
// PIPE: what the table holds is folded into the seed masks of the current and of the next window before it is cleared -- the masks then stay complete
// (a window's words as loaded + every pixel marked since), and a window is never gathered again
#define PEND_FLUSH() do { ST_FLUSH ++flushEpoch; \
                          if (PIPE) { mask &= ~wave_vote(s_pend[addr & (PEND - 1)] == addr); deadN |= wave_vote(s_pend[addrN & (PEND - 1)] == addrN); } \
                          __threadfence_block(); PEND_CLEAR(); __builtin_amdgcn_wave_barrier(); } while (0)
// wave-uniform: set the USED bit of pixel A (its current word is W)
#define MARK_USED(A, W) do { const int _slot = (A) & (PEND - 1); if (s_pend[_slot] != -1) PEND_FLUSH(); \
                             if (lane == 0) { grad[(A)] = (W) | kUsed; s_pend[_slot] = (A); } __builtin_amdgcn_wave_barrier(); } while (0)

    // The seed windows are a chain of dependent loads (key -> gradient word -> table entries); the key of the next window is fetched one
    // window ahead and the table entries of every growable seed of a window at its start, so a region start waits for neither.
    // PIPE: keyNext holds the keys of the window after next, (addrN, wN) the addresses and gradient words of the next window (requested one window
    // ahead: the 64 spatially random words of a window are a full memory round trip), deadN its seeds found in the pending table at a flush since
    uint32_t keyNext = lane < nkeys ? keys[lane] : 0u;
    int addrN = 0;
    uint32_t wN = kUsed;
    unsigned long long deadN = 0;
    if (PIPE) {
        addrN = lane < nkeys ? (int)(keyNext & kAddrMask) : 0;
        wN = lane < nkeys ? grad[addrN] : kUsed;
        keyNext = 64 + lane < nkeys ? keys[64 + lane] : 0u;
    }
    for (int base = 0; base < nkeys; base += 64) {
        const bool valid = base + lane < nkeys;
        int addr;
        uint32_t wseed;
        unsigned long long dead = 0;
        if (PIPE) {
            addr = addrN; wseed = wN; dead = deadN;
            const bool validN = base + 64 + lane < nkeys;
            addrN = validN ? (int)(keyNext & kAddrMask) : 0;
            // (the new requests go out BEHIND the wait for the old ones: tied to the arrival of this window's words, or the compiler hoists the key
            // load above that wait and every window waits for a request it has just made)
            int kidx = base + 128 + lane;
            asm volatile("" : "+v"(kidx), "+v"(addrN) : "v"(wseed));
            wN = validN ? grad[addrN] : kUsed;
            keyNext = kidx < nkeys ? keys[kidx] : 0u;
            deadN = 0;
        } else {
            addr = valid ? (int)(keyNext & kAddrMask) : 0;
            keyNext = base + 64 + lane < nkeys ? keys[base + 64 + lane] : 0u;
            wseed = valid ? grad[addr] : kUsed;
        }
        int maskEpoch = flushEpoch;      // the window's USED bits as loaded here are complete up to this flush count
        const bool isoSeed = (wseed & kIso) != 0;
#ifdef OLF_STATS
        ++st_win; st_seedl += __popcll(wave_vote(valid)); st_isos += __popcll(wave_vote(valid && isoSeed));
#endif
        // (kNotDef: the std::sort seed list also holds the undefined pixels of the smallest defined bin; ll_angle's seed loop skips them)
        unsigned long long mask = wave_vote(valid && !(wseed & (kUsed | kNotDef)) && s_pend[addr & (PEND - 1)] != addr) & ~dead;
        // PFSEED: a region's first step gathers the 3x3 around its seed; the seed's own row came with the window, the rows above and below are
        // requested here for every live seed, so that (all but the window's first region) find them in the cache
#ifdef OLF_STATS
        if (mask) ++st_winLive;
#endif
        uint32_t pfA = 0, pfB = 0;
        if (PFSEED && wave_bit(mask) && !isoSeed) { pfA = grad[max(addr - Ws, 0)]; pfB = grad[min(addr + Ws, g.Ps - 1)]; }
        // region_grow starts at the seed's own angle and at sums (cos, sin) of it (double argument, unlike the added pixels): both
        // are per-(gx, gy) table entries
        double seedAng = 0;
        float2 seedSum = make_float2(0.f, 0.f);
        if (wave_bit(mask) && !isoSeed) { const AngEnt* t = ent + (wseed & 0x3fffffu); seedAng = t->ang; seedSum = t->seed; }
        unsigned long long tabM = mask;      // lanes whose table entries are loaded (REFINE: the mask can gain lanes)
        const unsigned long long isoWin = wave_vote(isoSeed);
        while (mask) {
            // isolated seeds ahead of the first growable one are one-pixel regions: mark them all at once
            if (const unsigned long long isoM = isoWin & mask) {      // (nothing to do -- and three instructions -- for a window without isolated seeds left)
                const unsigned long long grow = mask & ~isoM;
                const unsigned long long lead = isoM & (grow ? ((1ull << __builtin_ctzll(grow)) - 1ull) : ~0ull);
                if (lead) {
                    const bool mine = wave_bit(lead);
                    const int slot = addr & (PEND - 1);
                    if (wave_vote(mine && s_pend[slot] != -1)) PEND_FLUSH();
                    if (mine) { grad[addr] = wseed | kUsed; s_pend[slot] = addr; }
                    __builtin_amdgcn_wave_barrier();
                    { unsigned long long bad = wave_vote(mine && s_pend[slot] != addr);
                      while (bad) { PEND_FLUSH(); if (wave_bit(bad)) s_pend[slot] = addr; __builtin_amdgcn_wave_barrier(); bad = wave_vote(wave_bit(bad) && s_pend[slot] != addr); } }
                    mask &= ~lead;
                    if (!mask) break;
                }
            }
            const int l = __builtin_ctzll(mask);
            const int seed = rlane(addr, l);
            // ---- region_grow ------------------------------------------------------------------
#ifdef OLF_STATS
            ++st_regions;
#endif
            // the seed's word as loaded with the window: only its USED bit can have changed since, and the mask says it has not
            const uint32_t pseed = (uint32_t)rlane((int)wseed, l);
            // REFINE: the growth below runs a second time from the same seed under the tolerance tau (cv refine())
            double precC = prec;
            bool regrown = false, accept = false;
            AgentRect rec = {0, 0, 0, 0, 0};
            int n;
            double reg_angle;
          for (;;) {
            n = 1;
            reg_angle = rlane_d(seedAng, l);
            float sumdx = __int_as_float(rlane(__float_as_int(seedSum.x), l)), sumdy = __int_as_float(rlane(__float_as_int(seedSum.y), l));
            // (seed / Ws through the host's exact multiply-shift pair: 4 scalar instructions where the compiler's division by a run-time value takes 20)
            const uint32_t sy0 = WIDE ? (uint32_t)seed / (uint32_t)Ws : __umulhi((uint32_t)seed, divM) >> divS, sx0 = (uint32_t)seed - sy0 * (uint32_t)Ws;
            int i = 0;
            if (WIN) {
                // ---- window phase: gather the 7 x 7 around the seed, one lane per pixel
                const int wx = (int)sx0 + (int)(short)(wpk & 0xffff), wy = (int)sy0 + (wpk >> 16);
                const unsigned long long winIn = kW49 & wave_vote((unsigned)wx < (unsigned)Ws) & wave_vote((unsigned)wy < (unsigned)Hs);
                const int wa = wave_bit(winIn) ? wy * Ws + wx : 0;
                const uint32_t ww = grad[(uint32_t)wa];
                const int wslot = wa & (PEND - 1);
                const int wpend = s_pend[wslot];
                unsigned long long live = winIn & wave_vote(!(ww & (kUsed | kNotDef))) & wave_vote(wpend != wa) & ~kSeedBit;
                double wang, wcs, wsn;
                float2 wdir;
                if (CHEAP) asm volatile("" : "=v"(wcs), "=v"(wsn), "=v"(wdir.x), "=v"(wdir.y)); else asm volatile("" : "=v"(wang), "=v"(wcs), "=v"(wsn));
                if (wave_bit(live)) { const AngEnt* t = ent + (ww & 0x3fffffu); wcs = t->cs; wsn = t->sn; if (CHEAP) wdir = t->seed; else wang = t->ang; }
                int fidx = lane == 24 ? 0 : -1;          // this pixel's place in the FIFO
                unsigned long long accM = kSeedBit;      // the region's pixels inside the window
#ifdef OLF_STATS
                ++st_first; st_cand1 += __popcll(live);
#endif
// (lanes aligned with the region angle, live or not; lanes that hold no table entry give garbage -- every use is masked with `live`)
#define WIN_ALIGNED(TH) ({ const double _n = fabs(d_sub((TH), wang)); wave_vote(_n <= prec) | wave_vote(_n >= precWrap); })
// CHEAP: the cheap test on the sums (SX, SY); a live pixel it cannot decide sends every lane to the reference's expression under the angle TH (evaluated there only)
#define WIN_ALIGNED_C(SX, SY, TH) ({ unsigned long long _sa, _sn; ALIGN_CHEAP(SX, SY, wdir, _sa, _sn); \
                                     if (live & ~(_sa | _sn)) { asm volatile("" : "=v"(wang)); uint32_t _wx = ww; asm volatile("" : "+v"(_wx)); if (wave_bit(live)) wang = ent[_wx & 0x3fffffu].ang; _sa = WIN_ALIGNED(TH); } _sa; })
                unsigned long long alM = CHEAP ? WIN_ALIGNED_C(sumdx, sumdy, reg_angle) : WIN_ALIGNED(reg_angle);
                // FIFO entries [i, lim) are replayed here: lim = min(n, place of the first pixel on the window's outer ring -- its 3 x 3 looks outside);
                // a long FIFO is what the general loop's 8 entries per gather are for
                int lim = 1, ringAt = 1 << 20;
                for (;;) {
                    const int p = __builtin_ctzll(wave_vote(fidx == i));
                    unsigned long long cand = (kM3 << (p - 8)) & live;      // the entry's live neighbours, in lane order = the reference's visiting order
                    unsigned long long al = cand & alM;
                    if (al) {
                        do {
                            const int n0 = n;
                            unsigned long long acc1, after;
                            if ((al & (al - 1ull)) == 0) {
                                // one aligned neighbour: the plain sequential step
                                const int c = __builtin_ctzll(al);
                                acc1 = al; after = ~((2ull << c) - 1ull);
                                fidx = wave_bit(al) ? n : fidx;
                                ++n;
                                const double cs_c = rlane_d(wcs, c), sn_c = rlane_d(wsn, c);
                                sumdx = (float)d_add((double)sumdx, cs_c);
                                sumdy = (float)d_add((double)sumdy, sn_c);
                                if (!CHEAP) reg_angle = d_mul((double)agent_fastAtan2(sumdy, sumdx), kDegToRads);
                            } else {
                                // several: speculate that they are accepted in lane order (the general loop's round, without its duplicate views -- a pixel is one
                                // lane here).  Lane L keeps the sums in force at its turn; lane 63, never a candidate, ends with the sums after all of them.
                                float sx = sumdx, sy = sumdy, bsx = sumdx, bsy = sumdy;
                                unsigned long long todo = al;
                                do {
                                    const int c = __builtin_ctzll(todo);
                                    todo &= todo - 1ull;
                                    const double cs_c = rlane_d(wcs, c), sn_c = rlane_d(wsn, c);
                                    sx = (float)d_add((double)sx, cs_c);
                                    sy = (float)d_add((double)sy, sn_c);
                                    if (lane > c) { bsx = sx; bsy = sy; }
                                } while (todo);
                                const int c0 = __builtin_ctzll(al);
                                double thOwn = 0;
                                unsigned long long reM;
                                if (CHEAP) {
                                    // (lanes up to c0 hold the sums before the round: their test is alM's)
                                    reM = WIN_ALIGNED_C(bsx, bsy, ({ ENSURE_ANGLE(); lane <= c0 ? reg_angle : d_mul((double)agent_fastAtan2(bsy, bsx), kDegToRads); }));
                                } else {
                                    thOwn = d_mul((double)agent_fastAtan2(bsy, bsx), kDegToRads);
                                    const double thg = lane <= c0 ? reg_angle : thOwn;
                                    reM = WIN_ALIGNED(thg);
                                }
                                const unsigned long long mis = (reM ^ alM) & cand;          // a decision that differs under the angle that governs it
                                const int mLane = mis ? __builtin_ctzll(mis) : 63;
                                const unsigned long long bm = mis ? ((1ull << mLane) - 1ull) : ~0ull;
                                acc1 = al & bm; after = ~bm;
                                fidx = wave_bit(acc1) ? n + wave_rank_below(acc1) : fidx;
                                n += __popcll(acc1);
                                // the state after the committed accepts: what the first undecided lane sees before its turn, or lane 63 after all of them
                                sumdx = __int_as_float(rlane(__float_as_int(bsx), mLane));
                                sumdy = __int_as_float(rlane(__float_as_int(bsy), mLane));
                                if (!CHEAP) reg_angle = rlane_d(thOwn, mLane);
                            }
                            live &= ~acc1; accM |= acc1;
                            if (const unsigned long long rm = acc1 & ~kD2) ringAt = min(ringAt, n0 + (int)__popcll(acc1 & ((1ull << __builtin_ctzll(rm)) - 1ull)));      // ((int): min(int, unsigned) resolves to the double overload)
                            alM = CHEAP ? WIN_ALIGNED_C(sumdx, sumdy, ({ ENSURE_ANGLE(); reg_angle; })) : WIN_ALIGNED(reg_angle);
                            cand &= after;                                   // the entry's later neighbours, under the new angle
                            al = cand & alM;
                        } while (al);
                        lim = n - i > OLF_WIN_MAXPEND ? i + 1 : min(n, ringAt);
                    }
                    if (++i >= lim) break;
                }
                const bool handover = i < n;
#undef WIN_ALIGNED
#undef WIN_ALIGNED_C
#ifdef OLF_STATS
                st_acc1 += n - 1; st_wentries += i; if (handover) { ++st_whand; st_wpend += n - i; } else ++st_wdone;
#endif
                // ---- publish the batch: USED bits + pending table; FIFO ring and log only where somebody will read them
                const bool mine = wave_bit(accM);
                if (wave_vote(wpend != -1) & accM) PEND_FLUSH();
                if (mine) { grad[wa] = ww | kUsed; s_pend[wslot] = wa; }
                __builtin_amdgcn_wave_barrier();
                { unsigned long long bad = wave_vote(mine && s_pend[wslot] != wa);
                  while (bad) { PEND_FLUSH(); if (wave_bit(bad)) s_pend[wslot] = wa; __builtin_amdgcn_wave_barrier(); bad = wave_vote(wave_bit(bad) && s_pend[wslot] != wa); } }
                if (handover || n >= minRegSize) {
                    LOG_ROOM(n);
                    if (mine) { const uint32_t xy = (uint32_t)wx | ((uint32_t)wy << 16); s_ring[fidx & (RING - 1)] = xy; LOG_AT(fidx) = make_uint2(xy, ww); }
                    __builtin_amdgcn_wave_barrier();
                }
            } else {
                MARK_USED(seed, pseed);
                LOG_ROOM(1);
                if (lane == 0) { const uint32_t pk = sx0 | (sy0 << 16); s_ring[0] = pk; LOG_AT(0) = make_uint2(pk, pseed); }
                __builtin_amdgcn_wave_barrier();
            }
#ifdef OLF_TIMING
            { long long t1 = __builtin_readcyclecounter(); t_seed += t1 - t0; t0 = t1; }
#endif
            // the check that two pixels accepted in one iteration did not hash to one table slot is read back after the commit's writes but only
            // looked at below the next iteration's ring read (one LDS round trip less on the agent's dependent chain)
            bool chkOn = false;
            int chkA = -1, chkV = -1;
// (a pixel that lost its slot to another pixel of the same batch is entered again once the table has been flushed: the table -- and with it the seed masks
// a flush folds it into -- must hold EVERY pixel marked since the window's words were loaded)
#define PEND_VERIFY() do { if (chkOn) { unsigned long long _bad = wave_vote(chkV != chkA); \
                               while (_bad) { PEND_FLUSH(); const int _s = chkA & (PEND - 1); if (wave_bit(_bad)) s_pend[_s] = chkA; __builtin_amdgcn_wave_barrier(); \
                                              _bad = wave_vote(wave_bit(_bad) && s_pend[_s] != chkA); } \
                               chkOn = false; } } while (0)
            while (i < n) {
#ifdef OLF_TIMING
                if (n >= minRegSize) ++it_big; else ++it_small;
#endif
                const int nb = min(8, n - i);
                const int e = lane >> 3, k = (lane & 7) + ((lane & 7) >= 4 ? 1 : 0);      // 8 FIFO entries x 8 neighbours (k = 4 is the entry's own pixel)
                const int pfOff = (k / 3 - 1) * Ws;
#ifdef OLF_STATS
                ++st_iters; if (n - i >= 14) ++st_deep1; if (n - i >= 21) ++st_deep2; if (n - i > RING) ++st_mem;
#endif
                // one predicate, no nested regions: every lane forms an address (0 when it has nothing to look at) and loads; only the
                // table lookups, which cost real cache traffic, are skipped for non-candidates
#ifdef OLF_TIMING2
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ps = __builtin_readcyclecounter(); ++p_n;
#endif
                // candidates as lane masks (scalar registers) from here on: lane 8 e + j looks at neighbour j of FIFO entry e, the first nb entries count
                const unsigned long long geo = ~0ull >> (64 - 8 * nb);      // (nb in [1, 8]: one shift instead of shift + not + two selects)
                // (the ring read is unconditional and the memory read a rare wave-uniform branch: as one conditional expression the two became a
                // generic-pointer flat load, in front of which the compiler waits for the previous iteration's stores to be acknowledged)
                uint32_t rp = s_ring[(i + e) & (RING - 1)];
                asm volatile("" : "+v"(rp));        // (keeps the two loads from being merged again)
                PEND_VERIFY();
                if (n - i > RING) { __threadfence_block(); rp = LOG_AT(i + (wave_bit(geo) ? e : 0)).x; __builtin_amdgcn_s_waitcnt(0x0F70); }   // window left the ring: read the FIFO from memory
                const int xx = (int)(rp & 0xffffu) + (k % 3) - 1, yy = (int)(rp >> 16) + (k / 3) - 1;
                const unsigned long long inImg = geo & wave_vote((unsigned)xx < (unsigned)Ws) & wave_vote((unsigned)yy < (unsigned)Hs);
                const int a = wave_bit(inImg) ? yy * Ws + xx : 0;
                PSTAMP(p_ring);
                const uint32_t pw = grad[(uint32_t)a];      // (an unsigned index: the 32-bit offset form of the load, no sign extension and 64-bit add)
                const int pendv = s_pend[a & (PEND - 1)];      // (unconditional: issued beside the gradient load instead of behind it; the commit reuses it)
                const int xy = xx | (yy << 16);
                unsigned long long cm = inImg & wave_vote(!(pw & (kUsed | kNotDef))) & wave_vote(pendv != a);
                // (left undefined for the lanes that are no candidates: nothing below looks at them, and three 64-bit zero moves per step are saved)
                double ang, cs, sn;
                float2 dir;
                if (CHEAP) asm volatile("" : "=v"(cs), "=v"(sn), "=v"(dir.x), "=v"(dir.y)); else asm volatile("" : "=v"(ang), "=v"(cs), "=v"(sn));
                PSTAMP(p_gather);
                uint32_t pfC = 0;
                if (wave_bit(cm)) {
                    const AngEnt* t = ent + (pw & 0x3fffffu);      // one 32-byte sector per candidate
                    cs = t->cs; sn = t->sn;
                    if (CHEAP) dir = t->seed; else ang = t->ang;
                    // PFCAND: an accepted candidate is a FIFO entry of the next step, whose gather reaches one row further out (issued behind the table
                    // loads: vector memory returns in order)
                    if (PFCAND) { const int ar = a + pfOff; if ((unsigned)ar < (unsigned)g.Ps && pfOff != 0) pfC = grad[ar]; }
                }
                // candidates in lane order = the reference's visiting order.  Under a fixed reg_angle every lane tests
                // its own candidate at once; the first aligned one is accepted (everything before it is rejected under
                // that same angle, as in the reference), the angle is updated and the rest is re-tested.
                PSTAMP(p_table);
                unsigned long long acc = 0;
#ifdef OLF_STATS
                st_cand += __popcll(cm); if (i == 0) { ++st_first; st_cand1 += __popcll(cm); }
#endif
                const int n0 = n;
                while (cm) {
                    // isaligned(): n = |theta - a|; if (n > 3pi/2) n = |n - 2pi|; n <= prec.  For n in (3pi/2, 2pi + prec] the subtraction is exact
                    // (Sterbenz), so the wrapped test is n >= 2pi - prec, with that bound rounded up to a double on the host (precWrap);
                    // angles lie in [0, 2pi], so n never exceeds 2pi + prec.
                    // (votes per comparison, combined as lane masks: a vote on the combined predicate costs a v_cndmask + v_cmp pair on top)
                    unsigned long long wasM;
                    if (CHEAP) {
                        unsigned long long sN;
                        ALIGN_CHEAP(sumdx, sumdy, dir, wasM, sN);
                        if (cm & ~(wasM | sN)) {      // a candidate within the margin of the tolerance: the reference's expression for everybody
                            ENSURE_ANGLE();
                            double angx;
                            asm volatile("" : "=v"(angx));
                            { uint32_t pwx = pw; asm volatile("" : "+v"(pwx)); if (wave_bit(cm)) angx = ent[pwx & 0x3fffffu].ang; }      // (opaque: the address arithmetic stays on this path)
                            const double nth = fabs(d_sub(reg_angle, angx));
                            wasM = wave_vote(nth <= prec) | wave_vote(nth >= precWrap);
                        }
                    } else {
                    const double nth = fabs(d_sub(reg_angle, ang));
                    // (REFINE: the tolerance of a second growth is computed on the device, so the wrapped test keeps its original form)
                    wasM = REFINE ? wave_vote((nth > kM32PI ? fabs(d_sub(nth, kM2PI)) : nth) <= precC)
                                  : wave_vote(nth <= prec) | wave_vote(nth >= precWrap);
                    }
                    const unsigned long long al = wasM & cm;      // cm only ever holds live candidates
                    if (!al) break;
#ifndef OLF_GROW_SINGLE_MAX
#define OLF_GROW_SINGLE_MAX 2      // (two aligned candidates: two plain steps are 110 instructions, a speculative round 139; from three on the round wins -- profiles/r4p_stages.txt)
#endif
                    if (OLF_GROW_SINGLE_MAX == 1 ? (al & (al - 1ull)) == 0 : __popcll(al) <= OLF_GROW_SINGLE_MAX) {
                        // a single aligned candidate: the plain sequential step
#ifdef OLF_STATS
                        ++st_single;
#endif
                        const int c = __builtin_ctzll(al);
                        cm &= ~((2ull << c) - 1ull);                 // c and everything before it is decided
                        const int a_c = rlane(a, c);
                        const double cs_c = rlane_d(cs, c), sn_c = rlane_d(sn, c);
                        acc |= 1ull << c;
                        ++n;
                        sumdx = (float)d_add((double)sumdx, cs_c);
                        sumdy = (float)d_add((double)sumdy, sn_c);
                        if (!CHEAP) reg_angle = d_mul((double)agent_fastAtan2(sumdy, sumdx), kDegToRads);
                        cm &= ~wave_vote(a == a_c);                   // the same pixel seen through another FIFO entry of this batch
                        continue;
                    }
                    // Several candidates are aligned under the current (exact) angle.  Speculate that they are accepted in lane order: the
                    // running sums after each accept are a cheap sequential chain (lane j keeps the sums after accept j), the region angle
                    // after each of them is then ONE fastAtan2 evaluated in parallel lanes, and every candidate is re-tested against the
                    // angle that governs it (the one after the accepts before it).  The speculation is exact up to the first candidate
                    // whose decision differs from the one under the initial angle; everything before it is committed -- at least the first
                    // accept, whose governing angle is the initial, exact one -- and the rest is classified again.
                    unsigned long long todo = al;
                    // lane L keeps the sums as they are when its turn comes -- after every speculated accept at a lower lane -- so the angle that governs
                    // its test is one fastAtan2 of its own registers (no cross-lane exchange); `spare`, a lane that holds no live candidate, keeps the
                    // sums after ALL of them (the region's new state when the whole round commits)
                    float sx = sumdx, sy = sumdy, bsx = sumdx, bsy = sumdy;
                    int dupLane = 64;                                            // the speculated accept this lane is another view of (its lane; 64: none)
                    while (todo) {
                        const int c = __builtin_ctzll(todo);
                        const int a_c = rlane(a, c);
                        const double cs_c = rlane_d(cs, c), sn_c = rlane_d(sn, c);
                        sx = (float)d_add((double)sx, cs_c);
                        sy = (float)d_add((double)sy, sn_c);
                        if (lane > c) { bsx = sx; bsy = sy; }
                        const unsigned long long twM = wave_vote(a == a_c);          // c and the other FIFO entries' views of the same pixel
                        if (wave_bit(twM)) dupLane = c;                              // (a lane can only ever equal one accepted pixel; for c itself dupLane == lane)
                        todo &= ~twM;
                    }
                    // (another view always sits at a higher lane than the accept it duplicates: the lowest lane of a pixel is the one that was picked)
                    const unsigned long long dupM = wave_vote(dupLane < lane), spec = al & ~dupM;
                    const int c0 = __builtin_ctzll(spec);
                    const unsigned long long freeM = ~cm;
                    const int spare = freeM ? __builtin_ctzll(freeM) : 0;
                    if (freeM && lane == spare) { bsx = sx; bsy = sy; }
                    double thOwn = 0;
                    unsigned long long reM;
                    if (CHEAP) {
                        // every candidate against the sums in force at its turn (lanes up to c0: the sums before the round, i.e. wasM's test again)
                        unsigned long long sN;
                        ALIGN_CHEAP(bsx, bsy, dir, reM, sN);
                        if (cm & ~dupM & ~(reM | sN)) {
                            ENSURE_ANGLE();
                            double angx;
                            asm volatile("" : "=v"(angx));
                            { uint32_t pwx = pw; asm volatile("" : "+v"(pwx)); if (wave_bit(cm)) angx = ent[pwx & 0x3fffffu].ang; }
                            const double thg = lane <= c0 ? reg_angle : d_mul((double)agent_fastAtan2(bsy, bsx), kDegToRads);
                            const double n2 = fabs(d_sub(thg, angx));
                            reM = wave_vote(n2 <= prec) | wave_vote(n2 >= precWrap);
                        }
                    } else {
                    thOwn = d_mul((double)agent_fastAtan2(bsy, bsx), kDegToRads);     // the angle in force at this lane's turn (spare: after all accepts)
                    const double thg = lane <= c0 ? reg_angle : thOwn;                              // nothing accepted before this lane: the current, exact angle
                    const double n2 = fabs(d_sub(thg, ang));
                    reM = REFINE ? wave_vote((n2 > kM32PI ? fabs(d_sub(n2, kM2PI)) : n2) <= precC)
                                 : wave_vote(n2 <= prec) | wave_vote(n2 >= precWrap);
                    }
                    const unsigned long long mis = (reM ^ wasM) & ~dupM & cm;
                    const int mLane = mis ? __builtin_ctzll(mis) : 64;               // everything below the first changed decision is decided
                    const unsigned long long bm = mis ? ((1ull << mLane) - 1ull) : ~0ull;
                    const unsigned long long okAcc = spec & bm, left = spec & ~bm;
                    const int t = __popcll(okAcc);
#ifdef OLF_STATS
                    ++st_rounds; st_k += __popcll(spec); st_t += t; if (!left) ++st_full; if (n >= 64) { ++st_rounds_big; st_k_big += __popcll(spec); st_t_big += t; }
#endif
                    acc |= okAcc;
                    n += t;
                    cm &= ~bm;
                    cm &= ~wave_vote(dupLane < mLane);                              // other views of the committed pixels (committed = speculated below mLane)
                    // the region's state after the t committed accepts: what the first uncommitted speculated lane sees before its turn, or the spare lane
                    if (left || freeM) {
                        const int src = left ? __builtin_ctzll(left) : spare;
                        sumdx = __int_as_float(rlane(__float_as_int(bsx), src));
                        sumdy = __int_as_float(rlane(__float_as_int(bsy), src));
                        if (!CHEAP) reg_angle = rlane_d(thOwn, src);
                    } else {      // all 64 lanes held live candidates and all of the round committed (never seen on images): the final angle on its own
                        sumdx = sx; sumdy = sy;
                        if (!CHEAP) reg_angle = d_mul((double)agent_fastAtan2(sy, sx), kDegToRads);
                    }
                }
                PSTAMP(p_chain);
                // vmcnt(0) while only loads can be outstanding (they have long returned): behind this point the stores below are the only vector
                // memory operations in flight, so the next iteration's ring read and address arithmetic need not wait for their acknowledgement
                // (without it the compiler waits at the loop head -- a table load of a lane that was no candidate may still target a live register)
                __builtin_amdgcn_s_waitcnt(0x0F70);
                if (PFCAND) asm volatile("" :: "v"(pfC));
                // the accepted lanes publish their pixel: USED bit, FIFO slot (ring + memory), pending-visibility table
                if (acc) {
                    LOG_ROOM(n);
                    const bool mine = wave_bit(acc);
                    const int slot = a & (PEND - 1);
                    if (wave_vote(pendv != -1) & acc) PEND_FLUSH();        // (nothing has written the table since the gather read the lane's slot)
                    if (mine) {
                        const int idx = n0 + wave_rank_below(acc);
                        grad[a] = pw | kUsed;
                        s_ring[idx & (RING - 1)] = (uint32_t)xy;
                        LOG_AT(idx) = make_uint2((uint32_t)xy, pw);
                        s_pend[slot] = a;
                    }
                    __builtin_amdgcn_wave_barrier();
                    // two accepted pixels of this batch hashing to one slot: only one survived in the table -> make both visible (PEND_VERIFY)
                    chkA = mine ? a : -1;
                    chkV = mine ? s_pend[slot] : -1;
                    chkOn = true;
                }
                i += nb;
                __builtin_amdgcn_wave_barrier();
                PSTAMP(p_commit);
            }
            PEND_VERIFY();
#undef PEND_VERIFY
            if (CHEAP && n >= minRegSize) ENSURE_ANGLE();      // (a region that will be logged: the reference's final region angle)
            if (!REFINE) break;
            if (over) break;                                 // (the log is clamped garbage from here on: the retry launch redoes the image)
            const uint2* lg = reg + rbase;
            if (!regrown) {
                if (n < minRegSize) break;
                __threadfence_block();                       // the log of this region is read back from memory
                rec = agent_region2rect(lg, n, reg_angle, prec);
                if (agent_density(n, rec) >= g.densityTh) { accept = true; break; }
                // refine(): statistics of the angles within rec.width of the seed; every pixel of the region becomes NOTUSED again
                const uint2 e0 = lg[0];
                const double xc = (double)(int)(e0.x & 0xffffu), yc = (double)(int)(e0.x >> 16);
                const double ang_c = ent[e0.y & 0x3fffffu].ang;
                double sum = 0, s_sum = 0;
                int cnt = 0;
                for (int q = 0; q < n; ++q) {
                    const uint2 e = lg[q];
                    if (agent_dist(xc, yc, (double)(int)(e.x & 0xffffu), (double)(int)(e.x >> 16)) < rec.width) {
                        double ad = d_sub(ent[e.y & 0x3fffffu].ang, ang_c);
                        while (ad <= -kPI) ad = d_add(ad, kM2PI);
                        while (ad > kPI) ad = d_sub(ad, kM2PI);
                        sum = d_add(sum, ad);
                        s_sum = d_add(s_sum, d_mul(ad, ad));
                        ++cnt;
                    }
                }
                for (int q = lane; q < n; q += 64) { const uint2 e = lg[q]; grad[(int)(e.x >> 16) * Ws + (int)(e.x & 0xffffu)] = e.y; }      // (the logged word has no USED bit)
                PEND_FLUSH();
                const double mean = sum / (double)cnt;
                precC = d_mul(2.0, sqrt(d_add(d_sub(s_sum, d_mul(d_mul(2.0, mean), sum)) / (double)cnt, d_mul(mean, mean))));
                regrown = true;
                continue;
            }
            // the second growth is through
            if (n < 2) break;
            __threadfence_block();
            rec = agent_region2rect(lg, n, reg_angle, prec);
            double density = agent_density(n, rec);
            if (density >= g.densityTh) { accept = true; break; }
            {   // reduce_region_radius(): shrink the radius by 25 % until the density is reached; a removed pixel is NOTUSED again and its place in the
                // list is taken by the last pixel (std::swap + pop_back), which the next rectangle's sums then meet in that order
                uint2* lw = reg + rbase;
                const uint2 e0 = lw[0];
                const double xc = (double)(int)(e0.x & 0xffffu), yc = (double)(int)(e0.x >> 16);
                const double ra1 = d_sub(rec.x1, xc), rb1 = d_sub(rec.y1, yc), ra2 = d_sub(rec.x2, xc), rb2 = d_sub(rec.y2, yc);
                const double radSq1 = d_add(d_mul(ra1, ra1), d_mul(rb1, rb1)), radSq2 = d_add(d_mul(ra2, ra2), d_mul(rb2, rb2));
                double radSq = radSq1 > radSq2 ? radSq1 : radSq2;
                bool ok = true;
                while (density < g.densityTh) {
                    radSq = d_mul(radSq, 0.5625);
                    for (int q = 0; q < n; ++q) {
                        const uint2 e = lw[q];
                        const double ax = d_sub((double)(int)(e.x & 0xffffu), xc), ay = d_sub((double)(int)(e.x >> 16), yc);
                        if (d_add(d_mul(ax, ax), d_mul(ay, ay)) > radSq) {
                            const uint2 last = lw[n - 1];
                            if (lane == 0) { grad[(int)(e.x >> 16) * Ws + (int)(e.x & 0xffffu)] = e.y; lw[q] = last; }
                            __threadfence_block();
                            --n; --q;
                        }
                    }
                    if (n < 2) { ok = false; break; }
                    rec = agent_region2rect(lw, n, reg_angle, prec);
                    density = agent_density(n, rec);
                }
                PEND_FLUSH();
                accept = ok;
            }
            break;
          }
            if (REFINE) {
                if (REFINE >= 2 && accept) {        // LSD_REFINE_ADV: keep the rectangle only if its (improved) number of false alarms is meaningful
                    rec.p = g.pProb;
                    __threadfence_block();
                    if (agent_rect_improve(g, grad, ent, rec, lane) <= g.logEps) accept = false;
                }
                if (accept) {
                    if (nreg < maxRegions) {
                        const SegCand cnd = rect_to_cand(g, rec.x1, rec.y1, rec.x2, rec.y2);
                        if (lane == 0) {
                            candAll[(size_t)img * maxRegions + nreg] = cnd;
                        }
                        ++nreg;
                    } else if (lane == 0) atomicOr(status, 8);
                }
            }
#ifdef OLF_TIMING
            { long long t1 = __builtin_readcyclecounter(); if (n >= minRegSize) { t_big += t1 - t0; ++n_big; } else { t_small += t1 - t0; ++n_small; } t0 = t1; }
#endif
            if (!REFINE && n >= minRegSize) {
                // a region large enough to become a segment: keep its pixel list (the log only moves forward for these) and record
                // (start, size, final region angle); k_lsd_rect fits all rectangles of the batch in parallel afterwards
                if (nreg < maxRegions) {
                    if (lane == 0) {
                        RegionRec rr; rr.start = rbase; rr.n = n; rr.angle = reg_angle;
                        recs[nreg] = rr;
                    }
                    ++nreg;
                    rbase += n;
                } else if (lane == 0) atomicOr(status, 8);
            }
#ifdef OLF_TIMING
            { long long t1 = __builtin_readcyclecounter(); t_rect += t1 - t0; t0 = t1; }
#endif
            // seeds later in this 64-key window may have been consumed by the region just grown
#ifdef OLF_STATS
            st_acc += n; if (n >= minRegSize) st_logged += n; if (n > 1) st_regl += __popcll(wave_vote(valid && lane > l));
#endif
            if (REFINE && maskEpoch != flushEpoch) {
                // pixels may have been given back since the window was loaded (refine un-uses whole regions): seeds of this window that were USED
                // then can be seeds now -- look at every later seed's word again, and fetch the table entries of the ones that are new
                const uint32_t wnow = valid ? grad[addr] : kUsed;
                mask = wave_vote(valid && lane > l && !(wnow & (kUsed | kNotDef)) && s_pend[addr & (PEND - 1)] != addr);
                if (wave_bit(mask)) wseed = wnow;
                if (wave_bit(mask & ~tabM) && !isoSeed) { const AngEnt* t = ent + (wseed & 0x3fffffu); seedAng = t->ang; seedSum = t->seed; }
                tabM |= mask;
                maskEpoch = flushEpoch;
            }
            else if (n == 1) mask &= mask - 1;
            else if (PIPE || maskEpoch == flushEpoch) {
                // no flush since the window's words were loaded: whatever has been marked since is still in the pending table -- no need to
                // gather the 64 (spatially random) seed words again
                mask &= ~((2ull << l) - 1ull) & wave_vote(s_pend[addr & (PEND - 1)] != addr);
            } else {
                mask = wave_vote(valid && lane > l && !(grad[addr] & (kUsed | kNotDef)) && s_pend[addr & (PEND - 1)] != addr);
                maskEpoch = flushEpoch;
            }
        }
        if (PFSEED) asm volatile("" :: "v"(pfA), "v"(pfB));      // (the requests are only ever waited for here)
    }
#undef ALIGN_CHEAP
#undef ENSURE_ANGLE
#undef LOG_ROOM
#undef LOG_AT
#undef MARK_USED
#undef PEND_FLUSH
#undef PEND_CLEAR
#ifdef OLF_TIMING
    if (lane == 0 && img == 0) { long long* o = reinterpret_cast<long long*>(status + 16); o[0] = t_seed; o[1] = t_small; o[2] = t_big; o[3] = t_rect; o[4] = n_small; o[5] = n_big; o[6] = it_small; o[7] = it_big; }
#endif
#ifdef OLF_TIMING2
    if (lane == 0 && img == 0) { long long* o = reinterpret_cast<long long*>(status + 16); o[0] = p_ring; o[1] = p_gather; o[2] = p_table; o[3] = p_chain; o[4] = p_commit; o[5] = p_n; }
#endif
#ifdef OLF_STATS
    if (lane == 0 && img == 0) { long long* o = reinterpret_cast<long long*>(status + 16); o[0] = st_rounds; o[1] = st_k; o[2] = st_t; o[3] = st_full; o[4] = st_single; o[5] = st_rounds_big; o[6] = st_k_big; o[7] = st_t_big;
        o[8] = st_flush; o[9] = st_iters; o[10] = st_deep1; o[11] = st_deep2; o[12] = st_cand; o[13] = st_regions; o[14] = st_mem;
        o[15] = nkeys; o[16] = st_win; o[17] = st_seedl; o[18] = st_regl; o[19] = st_acc; o[20] = st_isos; o[21] = st_logged; o[22] = st_winLive; o[23] = st_first; o[24] = st_cand1;
        o[25] = st_acc1; o[26] = st_wentries; o[27] = st_whand; o[28] = st_wdone; o[29] = st_wpend; }
#endif
    if (lane == 0) { regCount[img] = over ? 0 : nreg; if (growFmt) growFmt[img] = 1; if (g.spillOf && !retry) g.spillOf[img] = over ? -1 : 0; }
}

// ---------------------------------------------------------------------------------------------
// region2rect + the Vec4f end points of LSDDetectorC::detectImpl for every logged region (cv LSD region2rect / get_theta,
// LSDDetector_custom.cpp:270-289).  Regions are independent once grown: one thread per region, its sums run over the region's pixels in
// growth order (exactly the reference's order).  The segment candidate (clamped end points, length, keep flag) goes to a 24-byte record;
// k_lsd_emit then walks each image's candidates in detection order and writes the KeyLines that pass the length filter.

// CHAINED: the pixel lists are chains of 32-pixel chunks (lsd_grow.hip; rr.start = first chunk id) instead of one contiguous log per image.
template <bool CHAINED>
__device__ __forceinline__ void lsd_rect_region(const LineGeom& g, int img, int r, const uint32_t* __restrict__ gradAll,
                                                const uint32_t* __restrict__ regionAll, const RegionRec* __restrict__ recsAll,
                                                SegCand* __restrict__ candAll, const int* __restrict__ linksAll, int nChunks, const int* __restrict__ spillOf,
                                                const uint32_t* __restrict__ spillArena)
{
    const uint32_t* grad = gradAll + (size_t)img * g.Ps;
    const RegionRec rr = recsAll[(size_t)img * g.maxRegions + r];
    const uint32_t* px_list = CHAINED ? regionAll + (size_t)img * nChunks * 32 : nullptr;
    // (pixel, gradient word) pairs of the one-wave agent: in the image's own log or, for an image the retry launch grew, in its block of the spill arena
    // (spillOf: null in contexts whose log holds every pixel)
    const int sblk = CHAINED || !spillOf ? 0 : spillOf[img];
    const uint2* log2 = CHAINED ? nullptr
                      : (sblk > 0 ? reinterpret_cast<const uint2*>(spillArena) + (size_t)(sblk - 1) * g.Ps
                                  : reinterpret_cast<const uint2*>(regionAll + (size_t)img * g.regionStride)) + rr.start;
    const int* links = CHAINED ? linksAll + (size_t)img * nChunks : nullptr;
    const int n = rr.n, Ws = g.Ws;
    int cid = rr.start, nxt = -1;
    // U pixels of the list starting at position q0 (U divides the chunk size)
#define RECT_BLOCK(q0) (px_list + (size_t)cid * 32 + ((q0) & 31))
#define RECT_STEP(q0) do { if (CHAINED) { if (((q0) & 31) == 0) { if (q0) cid = nxt; nxt = links[cid]; } } } while (0)
    // both passes are chains of dependent loads (pixel list -> gradient word); 8 pixels are fetched per step so that the loads of a
    // step are in flight together, the additions stay strictly in growth order
    constexpr int U = 16;
    double x = 0, y = 0, sum = 0;
    for (int q0 = 0; q0 < n; q0 += U) {
        uint32_t rp[U], p[U];
        if (CHAINED) {
            RECT_STEP(q0);
            const uint32_t* blk = RECT_BLOCK(q0);
#pragma unroll
            for (int u = 0; u < U; ++u) rp[u] = q0 + u < n ? blk[u] : 0u;
#pragma unroll
            for (int u = 0; u < U; ++u) p[u] = grad[(int)(rp[u] >> 16) * Ws + (int)(rp[u] & 0xffffu)];
        } else {
            // (two log entries per 16-byte load: a wave's 64 lists are 64 different cache lines per instruction, and the fit waits for the texture path)
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                uint4 e = make_uint4(0u, 0u, 0u, 0u);
                if (q0 + u + 1 < n) __builtin_memcpy(&e, log2 + q0 + u, 16);
                else if (q0 + u < n) { const uint2 e1 = log2[q0 + u]; e.x = e1.x; e.y = e1.y; }
                rp[u] = e.x; p[u] = e.y; rp[u + 1] = e.z; p[u + 1] = e.w;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (q0 + u < n) {
                const int px = (int)(rp[u] & 0xffffu), py = (int)(rp[u] >> 16);
                const int gx = unpack_gx(p[u]), gy = unpack_gy(p[u]);
                const double wt = sqrt_quarter(gx * gx + gy * gy);      // (= sqrt(n / 4.0) bit for bit: tests/test_device_math_gpu.py)
                x = d_add(x, d_mul((double)px, wt));
                y = d_add(y, d_mul((double)py, wt));
                sum = d_add(sum, wt);
            }
        }
    }
    x = x / sum; y = y / sum;
    double Ixx = 0, Iyy = 0, Ixy = 0;
    cid = rr.start;
    for (int q0 = 0; q0 < n; q0 += U) {
        uint32_t rp[U], p[U];
        if (CHAINED) {
            RECT_STEP(q0);
            const uint32_t* blk = RECT_BLOCK(q0);
#pragma unroll
            for (int u = 0; u < U; ++u) rp[u] = q0 + u < n ? blk[u] : 0u;
#pragma unroll
            for (int u = 0; u < U; ++u) p[u] = grad[(int)(rp[u] >> 16) * Ws + (int)(rp[u] & 0xffffu)];
        } else {
            // (two log entries per 16-byte load: a wave's 64 lists are 64 different cache lines per instruction, and the fit waits for the texture path)
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                uint4 e = make_uint4(0u, 0u, 0u, 0u);
                if (q0 + u + 1 < n) __builtin_memcpy(&e, log2 + q0 + u, 16);
                else if (q0 + u < n) { const uint2 e1 = log2[q0 + u]; e.x = e1.x; e.y = e1.y; }
                rp[u] = e.x; p[u] = e.y; rp[u + 1] = e.z; p[u + 1] = e.w;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (q0 + u < n) {
                const int px = (int)(rp[u] & 0xffffu), py = (int)(rp[u] >> 16);
                const int gx = unpack_gx(p[u]), gy = unpack_gy(p[u]);
                const double wt = sqrt_quarter(gx * gx + gy * gy);      // (= sqrt(n / 4.0) bit for bit: tests/test_device_math_gpu.py)
                const double ex = d_sub((double)px, x), ey = d_sub((double)py, y);
                Ixx = d_add(Ixx, d_mul(d_mul(ey, ey), wt));
                Iyy = d_add(Iyy, d_mul(d_mul(ex, ex), wt));
                Ixy = d_sub(Ixy, d_mul(d_mul(ex, ey), wt));
            }
        }
    }
    const double dI = d_sub(Ixx, Iyy);
    const double lambda = d_mul(0.5, d_sub(d_add(Ixx, Iyy), sqrt(d_add(d_mul(dI, dI), d_mul(d_mul(4.0, Ixy), Ixy)))));
    double theta = (fabs(Ixx) > fabs(Iyy)) ? (double)dev_fastAtan2((float)d_sub(lambda, Ixx), (float)Ixy)
                                           : (double)dev_fastAtan2((float)Ixy, (float)d_sub(lambda, Iyy));
    theta = d_mul(theta, kDegToRads);
    {
        double diff = d_sub(theta, rr.angle);
        while (diff <= -kPI) diff = d_add(diff, kM2PI);
        while (diff > kPI) diff = d_sub(diff, kM2PI);
        if (fabs(diff) > g.prec) theta = d_add(theta, kPI);
    }
    double ddx, ddy;
    sincos(theta, &ddy, &ddx);
    // extent along the axis: "if (l > l_max) .. else if (l < l_min) .." from (0, 0) is max(0, max l) / min(0, min l)
    double l_min = 0, l_max = 0;
    cid = rr.start;
    for (int q0 = 0; q0 < n; q0 += U) {
        uint32_t rp[U];
        if (CHAINED) {
            RECT_STEP(q0);
            const uint32_t* blk = RECT_BLOCK(q0);
#pragma unroll
            for (int u = 0; u < U; ++u) rp[u] = q0 + u < n ? blk[u] : 0u;
        } else {
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                uint4 e = make_uint4(0u, 0u, 0u, 0u);
                if (q0 + u + 1 < n) __builtin_memcpy(&e, log2 + q0 + u, 16);
                else if (q0 + u < n) e.x = log2[q0 + u].x;
                rp[u] = e.x; rp[u + 1] = e.z;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (q0 + u < n) {
                const double rdx = d_sub((double)(int)(rp[u] & 0xffffu), x), rdy = d_sub((double)(int)(rp[u] >> 16), y);
                const double lq = d_add(d_mul(rdx, ddx), d_mul(rdy, ddy));
                l_max = fmax(l_max, lq); l_min = fmin(l_min, lq);
            }
        }
    }
    double x1 = d_add(x, d_mul(l_min, ddx)), y1 = d_add(y, d_mul(l_min, ddy));
    double x2 = d_add(x, d_mul(l_max, ddx)), y2 = d_add(y, d_mul(l_max, ddy));
    const SegCand c = rect_to_cand(g, x1, y1, x2, y2);
    candAll[(size_t)img * g.maxRegions + r] = c;
#undef RECT_BLOCK
#undef RECT_STEP
}

template <bool CHAINED>
__global__ __launch_bounds__(256) void k_lsd_rect(const LineGeom* __restrict__ gp, const uint32_t* __restrict__ gradAll,
                                                  const uint32_t* __restrict__ regionAll, const RegionRec* __restrict__ recsAll,
                                                  const int* __restrict__ regCount, SegCand* __restrict__ candAll,
                                                  const int* __restrict__ linksAll, int nChunks, const int* __restrict__ spillOf, const uint32_t* __restrict__ spillArena)
{
    const int img = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
    if (r >= regCount[img]) return;
    lsd_rect_region<CHAINED>(*gp, img, r, gradAll, regionAll, recsAll, candAll, linksAll, nChunks, spillOf, spillArena);
}

// after the multi-wave growth: chunk chains, except for the images that kernel gave up and the one-wave agent grew again (growFmt 1)
__global__ __launch_bounds__(256) void k_lsd_rect_mixed(const LineGeom* __restrict__ gp, const uint32_t* __restrict__ gradAll,
                                                        const uint32_t* __restrict__ regionAll, const RegionRec* __restrict__ recsAll,
                                                        const int* __restrict__ regCount, SegCand* __restrict__ candAll,
                                                        const int* __restrict__ linksAll, int nChunks, const int* __restrict__ growFmt, const int* __restrict__ spillOf,
                                                        const uint32_t* __restrict__ spillArena)
{
    const int img = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
    if (r >= regCount[img]) return;
    if (growFmt[img] == 1) lsd_rect_region<false>(*gp, img, r, gradAll, regionAll, recsAll, candAll, linksAll, nChunks, spillOf, spillArena);
    else lsd_rect_region<true>(*gp, img, r, gradAll, regionAll, recsAll, candAll, linksAll, nChunks, spillOf, spillArena);
}

// LSDDetectorC::detectImpl, Vec4f -> KeyLine (LSDDetector_custom.cpp:290-307): one workgroup per image, candidates in detection order,
// ordered compaction of the ones that pass the length filter (class_id = position in the output).
__global__ __launch_bounds__(256) void k_lsd_emit(const LineGeom* __restrict__ gp, const SegCand* __restrict__ candAll, const int* __restrict__ regCount,
                                                  olf_keyline* __restrict__ rawLines, int* __restrict__ rawCount, int* __restrict__ status)
{
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const LineGeom& g = *gp;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const SegCand* cand = candAll + (size_t)img * g.maxRegions;
    olf_keyline* out = rawLines + (size_t)img * g.maxDetect;
    const int nreg = regCount[img];
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int r0 = 0; r0 < nreg; r0 += 256) {
        const int r = r0 + tid;
        SegCand c; c.keep = 0;
        if (r < nreg) c = cand[r];
        const unsigned long long km = wave_vote(c.keep != 0);
        if (lane == 0) s_wave[wave] = __popcll(km);
        __syncthreads();
        int pos = s_base + wave_rank_below(km);
        for (int w = 0; w < wave; ++w) pos += s_wave[w];
        if (c.keep) {
            if (pos < g.maxDetect) {
                const float e0 = c.e0, e1 = c.e1, e2 = c.e2, e3 = c.e3;
                olf_keyline kl;
                kl.startPointX = e0; kl.startPointY = e1; kl.endPointX = e2; kl.endPointY = e3;
                kl.sPointInOctaveX = e0; kl.sPointInOctaveY = e1; kl.ePointInOctaveX = e2; kl.ePointInOctaveY = e3;
                kl.lineLength = c.length;
                const int rx1 = __float2int_rn(e0), ry1 = __float2int_rn(e1), rx2 = __float2int_rn(e2), ry2 = __float2int_rn(e3);
                kl.numOfPixels = max(abs(rx2 - rx1), abs(ry2 - ry1)) + 1;
                kl.angle = g.libmFloat ? glibc_atan2f(f_sub(e3, e1), f_sub(e2, e0)) : (float)atan2((double)f_sub(e3, e1), (double)f_sub(e2, e0));      // convention C.6
                kl.class_id = pos; kl.octave = 0;
                kl.size = f_mul(f_sub(e2, e0), f_sub(e3, e1));
                kl.response = f_div(kl.lineLength, (float)max(g.W, g.H));
                kl.pt_x = f_div(f_add(e2, e0), 2.0f); kl.pt_y = f_div(f_add(e3, e1), 2.0f);
                out[pos] = kl;
            } else atomicOr(status, 8);
        }
        __syncthreads();
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (tid == 0) rawCount[img] = min(s_base, g.maxDetect);
}

// ---------------------------------------------------------------------------------------------
int launch_lsd_sort(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s);
int launch_lsd_seedsort(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s, int nOverride, int kthrOverride, int depthOverride);
int launch_lsd_sort_wide(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s, int nOverride, long long kthrOverride, int depthOverride, int fullOverride);

int lsd_grow_waves(int n_images);
// the growth kernel a batch of n_images takes: 0 the one-wave agent, > 0 waves per image of the multi-wave kernel
// (lsd_refine = STD runs in the one-wave agent only)
// (a call of more images than the owner words are allocated for -- kMwMaxImages -- takes the one-wave agent whatever is forced)
static int lsd_grow_path(const LineGeom& g, const LineDeviceBufs& b, int n_images) { return (g.refine || g.wide || n_images > b.ownerImages) ? 0 : b.forceNW >= 0 ? b.forceNW : lsd_grow_waves(n_images); }

int launch_lsd_front(const LineGeom& g, LineDeviceBufs& b, const uint8_t* d_in, int in_pitch, int n_images, hipStream_t s)
{
    OLF_HIP_CHECK(hipMemsetAsync(b.maxN, 0, (size_t)n_images * 32 * sizeof(int), s));
    OLF_HIP_CHECK(hipMemsetAsync(b.keyCount, 0, (size_t)n_images * 32 * sizeof(int), s));
    { int rc = launch_gauss7_img(d_in, in_pitch, (size_t)in_pitch * g.H, b.lsdBlur, g.pitchW, (size_t)g.pitchW * g.H, g.W, g.H, g, 0, n_images, s);
      if (rc != OLF_OK) return rc; }
    // OLF_UPGRAD=0: enlargement and gradient as two kernels (A/B measurements)
    static const bool upgrad = [] { const char* e = getenv("OLF_UPGRAD"); return !e || atoi(e) != 0; }();
    const bool fused = upgrad && (g.resizeTiled & 4) && g.seedOrder == 1;
    if (fused) {
        const int nsx = (g.Ws + 3) / 4;
        hipLaunchKernelGGL(k_lsd_upgrad, dim3((nsx + 63) / 64, (g.Hs + UG_ROWS - 1) / UG_ROWS, n_images), dim3(64), 0, s, b.lsdBlur, b.scaled, b.grad, b.geom, b.rx, b.ry,
                           b.maxN, nsx, b.skipScaled ? 0 : 1);
    } else if (g.resizeTiled) {
        int rc = launch_resize_tiled(b.lsdBlur, (size_t)g.pitchW * g.H, g.pitchW, g.W, g.H, b.scaled, (size_t)g.pitchS * g.Hs, g.pitchS, g.Ws, g.Hs, b.rx,
                                     b.ry, n_images, s, (g.resizeTiled & 2) != 0);
        if (rc != OLF_OK) return rc;
    } else {
        const int quads = ((g.Ws + 3) >> 2) * g.Hs;
        hipLaunchKernelGGL(k_lsd_upsample, dim3((quads + 255) / 256, n_images), dim3(256), 0, s, b.lsdBlur, b.scaled, b.geom, b.rx, b.ry);
    }
    if (!fused) hipLaunchKernelGGL(k_lsd_grad, dim3((g.Ps + LG_CHUNK - 1) / LG_CHUNK, n_images), dim3(256), 0, s, b.scaled, b.grad, b.geom, b.maxN, b.chunkCnt);
    {
        // (ALLKEYS, OLF_KEYS_CHUNK=8192: 8192-pixel chunks -- the stage alone 62.6 against 63.4 ms per 6144 images, the two-stream step 231-236 against 232-234:
        // two blocks of 61 KB per CU overlap worse with the pyramid beside them than four of 36 KB; 4096 stays the default, profiles/r4y_keys_chunk_ab.txt)
        static const int envCh = getenv("OLF_KEYS_CHUNK") ? atoi(getenv("OLF_KEYS_CHUNK")) : 4096;
        const bool big = g.seedOrder == 1 && envCh == 8192 && (size_t)(8192 + 2 * g.Ws + 2) * sizeof(float) + 8192 * 2 + 64 <= 64 * 1024;
        const int CHK = big ? 8192 : LG_CHUNK;
        const int nChunks = (g.Ps + CHK - 1) / CHK, total = nChunks * n_images;
        const size_t lds = (size_t)(CHK + 2 * g.Ws + 2) * sizeof(float);
        if (lds > 60 * 1024) { set_error("LSD image wider than the key kernel's LDS window"); return OLF_ERR_CAPACITY; }
        const bool ow = lsd_grow_path(g, b, n_images) != 0;
        if (g.wide) {      // 64-bit keys, both conventions into keysA (lsd_wide.hip sorts them in place and lists the addresses in keysB)
            if (g.seedOrder == 1) hipLaunchKernelGGL((k_lsd_keys<false, true, LG_CHUNK, true>), dim3(total), dim3(KEYS_THREADS), lds, s, b.grad, b.geom, b.maxN, b.chunkCnt, b.keysA, b.keyCount, b.owner, b.angDeg, nChunks, total);
            else hipLaunchKernelGGL((k_lsd_keys<false, false, LG_CHUNK, true>), dim3(total), dim3(KEYS_THREADS), lds, s, b.grad, b.geom, b.maxN, b.chunkCnt, b.keysA, b.keyCount, b.owner, b.angDeg, nChunks, total);
            OLF_HIP_CHECK(hipGetLastError());
            if (b.sortEvent) OLF_HIP_CHECK(hipEventRecord(b.sortEvent, s));
            return launch_lsd_sort_wide(g, b, n_images, s, -1, -1, -1, -1);
        }
#define KEYS_LAUNCH(OW, AK, C, KBUF) hipLaunchKernelGGL((k_lsd_keys<OW, AK, C>), dim3(total), dim3(KEYS_THREADS), lds, s, b.grad, b.geom, b.maxN, b.chunkCnt, KBUF, b.keyCount, b.owner, b.angDeg, nChunks, total)
        if (g.seedOrder == 1) {
            if (big) { if (ow) KEYS_LAUNCH(true, true, 8192, b.keysA); else KEYS_LAUNCH(false, true, 8192, b.keysA); }
            else { if (ow) KEYS_LAUNCH(true, true, LG_CHUNK, b.keysA); else KEYS_LAUNCH(false, true, LG_CHUNK, b.keysA); }
        } else {
            if (ow) KEYS_LAUNCH(true, false, LG_CHUNK, b.keysB); else KEYS_LAUNCH(false, false, LG_CHUNK, b.keysB);
        }
#undef KEYS_LAUNCH
    }
    OLF_HIP_CHECK(hipGetLastError());
    if (b.sortEvent) OLF_HIP_CHECK(hipEventRecord(b.sortEvent, s));
    // the seed order: bins high to low; inside a bin raster order (a stable radix sort of the defined pixels' keys) or libstdc++'s std::sort order
    // over all pixels (convention C.9)
    { int rc = g.seedOrder == 1 ? launch_lsd_seedsort(g, b, n_images, s, -1, -1, -1) : launch_lsd_sort(g, b, n_images, s); if (rc != OLF_OK) return rc; }
    return OLF_OK;
}

int launch_lsd_grow_mw(const LineGeom& g, LineDeviceBufs& b, int n_images, int nw, int E, int G, hipStream_t s);

// workgroups (CUs) per image of the multi-wave growth: the drop-in's online shape -- one stereo pair per call -- leaves 254 CUs idle with one workgroup per image
// (OLF_LSD_GROUPS forces 1 / 2 / 4 for A/B runs)
int lsd_grow_groups(int n_images, int nw)
{
    static const int forced = [] { const char* e = getenv("OLF_LSD_GROUPS"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 2 || v == 4) ? v : 0; }();
    if (nw < 16) return 1;
    if (forced) return n_images <= kMgMaxImages ? forced : 1;      // (launch_lsd_grow halves it until every group fits a CU of its own)
    // (two groups, not four: one pair 8.45 against 8.95 ms, 8 pairs 10.3 against 10.8 -- the further a group runs ahead of the commit order the more of what it grows
    // is taken from it again by older seeds, profiles/r5a_growth_groups.txt)
    if (n_images <= kMgMaxImages) return 2;
    return 1;
}

// waves per image of the multi-wave growth: as many as keep the chip full (8 waves per SIMD x 1024 SIMDs) without leaving a small batch
// to a handful of waves; 0 selects the one-wave agent of round 1 (kept for A/B measurements, OLF_LSD_NW=0)
int lsd_grow_waves(int n_images)
{
    static const int forced = [] { const char* e = getenv("OLF_LSD_NW"); return e ? std::max(-1, std::min(16, atoi(e))) : -1; }();
    if (forced >= 0) return forced;
    if (n_images <= 512) return 16;
    if (n_images <= 1024) return 8;
    if (n_images <= 1536) return 4;      // (8 waves x 1280 images no longer fit the 8192 wave slots: the 1080p batch takes 203 ms with 8, 188 with 4; KITTI size: equal)
    if (n_images <= 2048) return 8;
    if (n_images <= kMwMaxImages) return 4;
    return 0;      // big batches are throughput bound, and there the one-wave agent does the least work per image
}

int launch_lsd_grow(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s)
{
    const int nw = lsd_grow_path(g, b, n_images);
    if (b.spillCtl) OLF_HIP_CHECK(hipMemsetAsync(b.spillCtl, 0, sizeof(int), s));      // blocks of the spill arena handed out in this call
    b.chained = nw != 0;
    if (nw > 0) {
        // OLF_LSD_ROB: reorder-buffer entries for experiments (a power of two in [128, 512]; anything else is ignored)
        static const int envE = [] { const char* e = getenv("OLF_LSD_ROB"); const int v = e ? atoi(e) : 0; return (v == 128 || v == 256 || v == 512 || v == 1024) ? v : 0; }();
        int E = b.forceE > 0 ? b.forceE : envE > 0 ? envE : (nw >= 16 ? 512 : nw >= 8 ? 256 : 128);
        int G = b.forceG > 0 ? b.forceG : lsd_grow_groups(n_images, nw);
        const int pool = b.poolChunks > 0 ? std::min(b.poolChunks, b.nChunks) : b.nChunks;
        // (every group of an image has to be resident -- a group spins on its partners' watermarks -- i.e. one workgroup per CU of THIS device)
        static const int nCU = [] { int dev = 0, n = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 64; return n; }();
        while (G > 1 && (!b.mg || n_images > b.mgImages || n_images * G > nCU || pool / G < E + 64)) G >>= 1;
        const int rc = launch_lsd_grow_mw(g, b, n_images, nw, E, G, s);
        if (rc != OLF_OK) return rc;
        // an image whose chunk pool or region log ran out under the multi-wave kernel (it re-runs regions, so it needs more of both than the
        // sequential replay) is grown again by the one-wave agent, whose log cannot overflow; every other workgroup of this launch exits at once
        hipLaunchKernelGGL((k_lsd_grow<0, 0>), dim3(n_images), dim3(64), 0, s, b.geom, b.grad, b.keysB, b.keyCount, b.region,
                           reinterpret_cast<RegionRec*>(b.keysA), b.regCount, b.status, reinterpret_cast<const AngEnt*>(b.angEnt), b.growFmt, (SegCand*)nullptr, 0);
        OLF_HIP_CHECK(hipGetLastError());
        return OLF_OK;
    }
    // a log smaller than the image (batch contexts, olf_debug_lsd_log_cap): a second launch grows the images that outgrew theirs again, on blocks of the spill
    // arena; every other workgroup of it exits at once
    for (int RETRY = 0; RETRY < (g.logCap < g.Ps ? 2 : 1); ++RETRY) {
    // (lsd_refine: the candidates go to keysA -- keysB still holds the seed list the agent is reading; launch_lsd_rect emits from there)
    if (g.wide) {
        // (64-bit sort keys: the agent reading 32-bit addresses, the capacity path of lsd_wide.hip; lsd_ang_th > 80 degrees takes it without the cheap alignment test)
#define GROWW(RF) hipLaunchKernelGGL((k_lsd_grow<RF, (RF ? 32 : 32 | 27)>), dim3(n_images), dim3(64), 0, s, b.geom, b.grad, b.keysB, b.keyCount, b.region, \
                           RF ? (RegionRec*)nullptr : reinterpret_cast<RegionRec*>(b.keysA), b.regCount, b.status, reinterpret_cast<const AngEnt*>(b.angEnt), (int*)nullptr, \
                           RF ? reinterpret_cast<SegCand*>(b.keysA) : (SegCand*)nullptr, RETRY)
        if (g.refine >= 2) GROWW(2); else if (g.refine) GROWW(1); else if (g.alignTanLo < 0.f) hipLaunchKernelGGL((k_lsd_grow<0, 32 | 11>), dim3(n_images), dim3(64), 0, s, b.geom, b.grad, b.keysB, b.keyCount, b.region,
                           reinterpret_cast<RegionRec*>(b.keysA), b.regCount, b.status, reinterpret_cast<const AngEnt*>(b.angEnt), (int*)nullptr, (SegCand*)nullptr, RETRY); else GROWW(0);
#undef GROWW
    } else if (g.refine >= 2)
        hipLaunchKernelGGL((k_lsd_grow<2, 0>), dim3(n_images), dim3(64), 0, s, b.geom, b.grad, b.keysB, b.keyCount, b.region,
                           (RegionRec*)nullptr, b.regCount, b.status, reinterpret_cast<const AngEnt*>(b.angEnt), (int*)nullptr, reinterpret_cast<SegCand*>(b.keysA), RETRY);
    else if (g.refine)
        hipLaunchKernelGGL((k_lsd_grow<1, 0>), dim3(n_images), dim3(64), 0, s, b.geom, b.grad, b.keysB, b.keyCount, b.region,
                           (RegionRec*)nullptr, b.regCount, b.status, reinterpret_cast<const AngEnt*>(b.angEnt), (int*)nullptr, reinterpret_cast<SegCand*>(b.keysA), RETRY);
    else {
        static const int pfEnv = [] { const char* e = getenv("OLF_GROW_PF"); return e ? atoi(e) : 27; }();
        const int pf = g.alignTanLo < 0.f ? (pfEnv & ~16) : pfEnv;      // (ang_th > 80 degrees: no cheap alignment test)
#define GROW0(PFV) hipLaunchKernelGGL((k_lsd_grow<0, PFV>), dim3(n_images), dim3(64), 0, s, b.geom, b.grad, b.keysB, b.keyCount, b.region, \
                           reinterpret_cast<RegionRec*>(b.keysA), b.regCount, b.status, reinterpret_cast<const AngEnt*>(b.angEnt), (int*)nullptr, (SegCand*)nullptr, RETRY)
        if (pf == 0) GROW0(0); else if (pf == 3) GROW0(3); else if (pf == 11) GROW0(11); else if (pf == 19) GROW0(19); else GROW0(27);
#undef GROW0
    }
    }
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

int launch_lsd_rect(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s)
{
    if (g.refine) {      // the agent has fitted the rectangles itself (candidates in keysA)
        hipLaunchKernelGGL(k_lsd_emit, dim3(n_images), dim3(256), 0, s, b.geom, reinterpret_cast<const SegCand*>(b.keysA), b.regCount, b.rawLines,
                           b.rawCount, b.status);
        OLF_HIP_CHECK(hipGetLastError());
        return OLF_OK;
    }
    // the sorted keys (keysB) are dead once the agents are done: the 24-byte segment candidates live there
    if (b.chained)
        hipLaunchKernelGGL(k_lsd_rect_mixed, dim3((g.rectGrid + 255) / 256, n_images), dim3(256), 0, s, b.geom, b.grad, b.region,
                           reinterpret_cast<const RegionRec*>(b.keysA), b.regCount, reinterpret_cast<SegCand*>(b.keysB), b.links, b.nChunks, b.growFmt, g.logCap < g.Ps ? b.spillOf : (const int*)nullptr, b.spill);
    else
        // (one thread per region in index order.  Dealing an image's regions out by size -- one block per image, (size, index) keys sorted in LDS, so that a wave's 64
        // lists have similar lengths -- was built and measured: 10.2 against 8.1 ms per 6144 images (profiles/r4ac_rect_sorted_ab.txt); the fit waits for its list
        // loads, not for the longest list of its wave)
        hipLaunchKernelGGL(k_lsd_rect<false>, dim3((g.rectGrid + 255) / 256, n_images), dim3(256), 0, s, b.geom, b.grad, b.region,
                           reinterpret_cast<const RegionRec*>(b.keysA), b.regCount, reinterpret_cast<SegCand*>(b.keysB), b.links, b.nChunks, g.logCap < g.Ps ? b.spillOf : (const int*)nullptr, b.spill);
    hipLaunchKernelGGL(k_lsd_emit, dim3(n_images), dim3(256), 0, s, b.geom, reinterpret_cast<const SegCand*>(b.keysB), b.regCount, b.rawLines,
                       b.rawCount, b.status);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
