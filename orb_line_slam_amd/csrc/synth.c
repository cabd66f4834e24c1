/* synth.c -- integer-only seeded synthetic stereo generator.
 *
 * The reference ships no images (KITTI / EuRoC are external, Examples/PL/PL_stereo_kitti.cc:47-58
 * reads them from a user path), so parity tests and bench.py run on images made here.
 * Everything is integer arithmetic on a xorshift64* stream, so the CPU container and the
 * GPU box produce identical bytes for a given (seed, W, H).
 *
 * Scene: smooth ramp background, K = W*H/2500 rotated rectangles (corners -> FAST/ORB,
 * edges -> LSD), K/2 thin strokes (long line segments), K small high-contrast blobs.
 * Left = scene + noiseL.  Right(x,y) = scene(x + d(y), y) + noiseR with the integer
 * disparity d(y) = 4 + 60*y/H (columns beyond the edge replicated), so stereo matches
 * exist with a known disparity field.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s; } rng_t;

static inline uint64_t rng_next(rng_t* r)
{
    uint64_t x = r->s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    r->s = x;
    return x * 0x2545F4914F6CDD1DULL;
}
/* uniform integer in [lo, hi] */
static inline int rng_range(rng_t* r, int lo, int hi)
{
    return lo + (int)((rng_next(r) >> 33) % (uint64_t)(hi - lo + 1));
}

static int isqrt_i(int v)
{
    int r = 0;
    while ((r + 1) * (r + 1) <= v) ++r;
    return r;
}

/* paint the rotated rectangle centred (cx,cy), axis (a,b) (un-normalised), half extents (hu,hv) */
static void paint_rect(uint8_t* img, int W, int H, int cx, int cy, int a, int b, int hu, int hv, int grey)
{
    int n = isqrt_i(a * a + b * b);
    if (n == 0) { a = 1; b = 0; n = 1; }
    int ext = hu + hv + 2;
    int x0 = cx - ext < 0 ? 0 : cx - ext, x1 = cx + ext >= W ? W - 1 : cx + ext;
    int y0 = cy - ext < 0 ? 0 : cy - ext, y1 = cy + ext >= H ? H - 1 : cy + ext;
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
            int dx = x - cx, dy = y - cy;
            int u = dx * a + dy * b, v = -dx * b + dy * a;
            if (u < 0) u = -u;
            if (v < 0) v = -v;
            if (u <= hu * n && v <= hv * n) img[(size_t)y * W + x] = (uint8_t)grey;
        }
}

/* scene 0: the default mix; scene 1 ("long"): fewer, larger shapes and longer strokes, so that the detected segments average about
 * 0.08 * W pixels, the length SURVEY App. D's byte model assumes for a KITTI frame */
/* the same shape with anti-aliased borders: pixels near the border take the grey value in proportion to their covered area (4 x 4
 * sub-samples), like the edges of a real camera image -- an aliased staircase breaks the LSD regions of a long edge into short pieces */
static void paint_rect_aa(uint8_t* img, int W, int H, int cx, int cy, int a, int b, int hu, int hv, int grey)
{
    int n = isqrt_i(a * a + b * b);
    if (n == 0) { a = 1; b = 0; n = 1; }
    int ext = hu + hv + 3;
    int x0 = cx - ext < 0 ? 0 : cx - ext, x1 = cx + ext >= W ? W - 1 : cx + ext;
    int y0 = cy - ext < 0 ? 0 : cy - ext, y1 = cy + ext >= H ? H - 1 : cy + ext;
    const int margin = (a < 0 ? -a : a) + (b < 0 ? -b : b);
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
            int dx = x - cx, dy = y - cy;
            int u = dx * a + dy * b, v = -dx * b + dy * a;
            int au = u < 0 ? -u : u, av = v < 0 ? -v : v;
            if (au > hu * n + margin || av > hv * n + margin) continue;
            if (au <= hu * n - margin && av <= hv * n - margin) { img[(size_t)y * W + x] = (uint8_t)grey; continue; }
            int c = 0;                                  /* coverage in sixteenths: sub-sample centres at (x + (2i-3)/8, y + (2j-3)/8) */
            for (int j = 0; j < 4; ++j)
                for (int i = 0; i < 4; ++i) {
                    int sx = 8 * dx + 2 * i - 3, sy = 8 * dy + 2 * j - 3;
                    int su = sx * a + sy * b, sv = -sx * b + sy * a;
                    if (su < 0) su = -su;
                    if (sv < 0) sv = -sv;
                    if (su <= 8 * hu * n && sv <= 8 * hv * n) ++c;
                }
            if (c) img[(size_t)y * W + x] = (uint8_t)(((int)img[(size_t)y * W + x] * (16 - c) + grey * c + 8) / 16);
        }
}

static void make_scene(uint8_t* scene, int W, int H, uint64_t seed, int kind)
{
    rng_t r; r.s = 0x9E3779B97F4A7C15ULL ^ (seed + 1);
    if (r.s == 0) r.s = 1;
    for (int i = 0; i < 8; ++i) rng_next(&r);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            scene[(size_t)y * W + x] = (uint8_t)(32 + (x * 96) / W + (y * 64) / H);
    int K = (W * H) / 2500;
    if (kind == 1) {
        int Kl = K / 5 > 4 ? K / 5 : 4;
        for (int k = 0; k < Kl; ++k) {                  /* large rotated rectangles */
            int cx = rng_range(&r, 0, W - 1), cy = rng_range(&r, 0, H - 1);
            int a = rng_range(&r, -64, 64), b = rng_range(&r, -64, 64);
            int hu = rng_range(&r, 40, 220), hv = rng_range(&r, 30, 160);
            int g = rng_range(&r, 16, 240);
            paint_rect_aa(scene, W, H, cx, cy, a, b, hu, hv, g);
        }
        for (int k = 0; k < 3 * Kl; ++k) {              /* long strokes */
            int cx = rng_range(&r, 0, W - 1), cy = rng_range(&r, 0, H - 1);
            int a = rng_range(&r, -64, 64), b = rng_range(&r, -64, 64);
            int hl = rng_range(&r, 80, W / 2 > 81 ? W / 2 : 81);
            int hw = rng_range(&r, 0, 2);
            int g = rng_range(&r, 0, 255);
            paint_rect_aa(scene, W, H, cx, cy, a, b, hl, hw, g);
        }
        for (int k = 0; k < K; ++k) {                   /* tiny blobs: corners for FAST, sides below the minimum line length */
            int cx = rng_range(&r, 0, W - 1), cy = rng_range(&r, 0, H - 1);
            int a = rng_range(&r, -64, 64), b = rng_range(&r, -64, 64);
            int hu = rng_range(&r, 1, 3), hv = rng_range(&r, 1, 3);
            int g = rng_range(&r, 0, 255);
            paint_rect(scene, W, H, cx, cy, a, b, hu, hv, g);
        }
        return;
    }
    if (kind == 2) {
        /* "bars": one long thin anti-aliased bar per cell of a jittered grid, nearly horizontal so that bars do not cross (a crossing cuts
         * both edges), two long straight edges each: the 500 strongest segments then average close to 0.08 * W pixels -- the line length
         * SURVEY App. D's byte model assumes for a KITTI frame -- and small blobs in the gaps between the bar rows feed the corner detector */
        const int cw = W / 9, ch = 15;
        for (int cy0 = ch / 2; cy0 + ch / 2 <= H; cy0 += ch)
            for (int cx0 = ((cy0 / ch) & 1) * (cw / 2); cx0 < W; cx0 += cw) {
                int cx = cx0 + cw / 2 + rng_range(&r, -6, 6), cy = cy0 + rng_range(&r, -1, 1);
                int a = 64, b = rng_range(&r, -2, 2);
                int hl = rng_range(&r, (cw * 36) / 100, (cw * 46) / 100), hw = 2;
                int base = scene[(size_t)(cy < H ? cy : H - 1) * W + (cx < W ? cx : W - 1)];
                int g = base > 128 ? rng_range(&r, 0, base - 70) : rng_range(&r, base + 70, 255);
                paint_rect_aa(scene, W, H, cx, cy, a, b, hl, hw, g);
                for (int q = 0; q < 3; ++q) {           /* blobs on the boundary between this row of cells and the next */
                    int bx = cx0 + rng_range(&r, 0, cw - 1), by = cy0 + ch / 2 + rng_range(&r, -1, 0);
                    int ba = rng_range(&r, -64, 64), bb = rng_range(&r, -64, 64);
                    int bg = rng_range(&r, 0, 255);
                    if (bx < W && by < H) paint_rect(scene, W, H, bx, by, ba, bb, rng_range(&r, 1, 2), rng_range(&r, 1, 2), bg);
                }
            }
        return;
    }
    for (int k = 0; k < K; ++k) {                       /* rotated rectangles */
        int cx = rng_range(&r, 0, W - 1), cy = rng_range(&r, 0, H - 1);
        int a = rng_range(&r, -64, 64), b = rng_range(&r, -64, 64);
        int hu = rng_range(&r, 6, 60), hv = rng_range(&r, 6, 60);
        int g = rng_range(&r, 16, 240);
        paint_rect(scene, W, H, cx, cy, a, b, hu, hv, g);
    }
    for (int k = 0; k < K / 2; ++k) {                   /* strokes */
        int cx = rng_range(&r, 0, W - 1), cy = rng_range(&r, 0, H - 1);
        int a = rng_range(&r, -64, 64), b = rng_range(&r, -64, 64);
        int hl = rng_range(&r, 20, W / 8 > 21 ? W / 8 : 21);
        int hw = rng_range(&r, 0, 1);
        int g = rng_range(&r, 0, 255);
        paint_rect(scene, W, H, cx, cy, a, b, hl, hw, g);
    }
    for (int k = 0; k < K; ++k) {                       /* small blobs */
        int cx = rng_range(&r, 0, W - 1), cy = rng_range(&r, 0, H - 1);
        int a = rng_range(&r, -64, 64), b = rng_range(&r, -64, 64);
        int hu = rng_range(&r, 2, 6), hv = rng_range(&r, 2, 6);
        int g = rng_range(&r, 0, 255);
        paint_rect(scene, W, H, cx, cy, a, b, hu, hv, g);
    }
}

static void add_noise(uint8_t* dst, const uint8_t* src, int n, rng_t* r)
{
    for (int i = 0; i < n; ++i) {
        int v = (int)src[i] + rng_range(r, -3, 3);
        dst[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

/* left/right: W*H bytes each, row-major, stride W.  Returns 0, or -1 on bad arguments. */
int olf_synth_stereo_scene(uint64_t seed, int W, int H, int kind, uint8_t* left, uint8_t* right)
{
    if (W < 64 || H < 64 || !left || !right || kind < 0 || kind > 2) return -1;
    uint8_t* scene = (uint8_t*)malloc((size_t)W * H);
    uint8_t* shifted = (uint8_t*)malloc((size_t)W * H);
    if (!scene || !shifted) { free(scene); free(shifted); return -1; }
    make_scene(scene, W, H, seed, kind);
    for (int y = 0; y < H; ++y) {
        int d = 4 + (60 * y) / H;
        for (int x = 0; x < W; ++x) {
            int xs = x + d; if (xs >= W) xs = W - 1;
            shifted[(size_t)y * W + x] = scene[(size_t)y * W + xs];
        }
    }
    rng_t rl, rr;
    rl.s = 0xD1B54A32D192ED03ULL ^ (seed * 2 + 1);
    rr.s = 0xABCDEF0123456789ULL ^ (seed * 2 + 2);
    if (rl.s == 0) rl.s = 1;
    if (rr.s == 0) rr.s = 1;
    add_noise(left, scene, W * H, &rl);
    add_noise(right, shifted, W * H, &rr);
    free(scene); free(shifted);
    return 0;
}

int olf_synth_stereo(uint64_t seed, int W, int H, uint8_t* left, uint8_t* right) { return olf_synth_stereo_scene(seed, W, H, 0, left, right); }
