// lbd.hip -- the rest of Lineextractor::operator() (reference src/LineExtractor.cc:55-66), gfx950:
//   top-N by response (std::sort + resize + class_id renumbering, :56-65; convention C.3: stable order)
//   BinaryDescriptor::compute -> computeImpl / computeSobel / computeLBD / binaryConversion
//   (Thirdparty/line_descriptor/src/binary_descriptor_custom.cpp:350-412, :539-687, :1026-1372).
// The LBD float chains are order-sensitive (sequential += over the support-region columns, then over
// its 63 rows), so the mapping is: one thread per (line, support row) walks the row's columns in the
// reference order; one thread per line folds the 63 rows into the 9 bands and finishes the descriptor.
// No FMA contraction anywhere (f_mul/f_add are __fmul_rn/__fadd_rn).
#include "line_internal.hpp"
#include "device_math.hpp"

namespace olf {

#define OLF_TRY_RC(expr) do { int _rc = (expr); if (_rc != OLF_OK) return _rc; } while (0)

__constant__ int8_t c_bandPairs[64] = {
#include "lbd_band_pairs.inc"
};

// ---------------------------------------------------------------------------------------------
constexpr int LS_LDS_KEYS = 4096;      // raw segments sorted in LDS (32 KB: two blocks per CU on the line tail; a KITTI image has a few hundred); longer lists are sorted in the key buffer

// bitonic sort of sortN 64-bit keys, descending, by one 256-thread workgroup; `sk` in LDS or -- a list too long for it (noise images, lsd_scale 2) --
// in a per-image slice of global memory (all waves of a workgroup share their CU's L1, so a barrier orders their global accesses too)
template <typename P>
__device__ __forceinline__ void ls_bitonic_desc(P sk, int sortN, int tid)
{
    for (int k = 2; k <= sortN; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < sortN; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = sk[i], b = sk[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (a < b) : (a > b)) { sk[i] = b; sk[ixj] = a; }
                }
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(256) void k_line_select(const LineGeom* __restrict__ gp, const olf_keyline* __restrict__ rawLines,
                                                     const int* __restrict__ rawCount, olf_keyline* __restrict__ kls, int* __restrict__ counts,
                                                     unsigned long long* __restrict__ scratchAll, size_t scratchStride)
{
    extern __shared__ unsigned long long skeys[];
    const LineGeom& g = *gp;
    const int img = blockIdx.x, tid = threadIdx.x;
    const int R = min(rawCount[img], g.maxDetect);
    const olf_keyline* raw = rawLines + (size_t)img * g.maxDetect;
    olf_keyline* out = kls + (size_t)img * g.outCap;
    if (!(R > g.nFeatures && g.nFeatures != 0)) {
        const int n = min(R, g.outCap);
        for (int i = tid; i < n; i += 256) out[i] = raw[i];
        if (tid == 0) counts[img] = n;
        return;
    }
    int sortN = 64;
    while (sortN < R) sortN <<= 1;
    const bool inLds = sortN <= LS_LDS_KEYS;          // (wave-uniform)
    unsigned long long* gk = scratchAll + (size_t)img * scratchStride;
    for (int i = tid; i < sortN; i += 256) {
        unsigned long long k = 0;
        if (i < R) k = ((unsigned long long)__float_as_uint(raw[i].response) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
        if (inLds) skeys[i] = k; else gk[i] = k;   // responses are >= 0, so the float bit pattern orders like the value
    }
    __syncthreads();
    if (inLds) ls_bitonic_desc(skeys, sortN, tid); else ls_bitonic_desc(gk, sortN, tid);
    for (int i = tid; i < g.nFeatures; i += 256) {
        const unsigned idx = 0xffffffffu - (unsigned)((inLds ? skeys[i] : gk[i]) & 0xffffffffull);
        olf_keyline kl = raw[idx];
        kl.class_id = i;
        out[i] = kl;
    }
    if (tid == 0) counts[img] = g.nFeatures;
}

// cv::Sobel(CV_16S, ksize 3, BORDER_REFLECT_101) of the sigma-1 blurred image, dx and dy packed
__global__ __launch_bounds__(256) void k_sobel3(const uint8_t* __restrict__ blur, uint32_t* __restrict__ dxdy, const LineGeom* __restrict__ gp)
{
    // 8 pixels per thread: three rows x four aligned 32-bit words (the pixels' two words and one neighbour either side) instead of 48 byte loads,
    // two 16-byte stores (the rows of dxdy are padded to a multiple of 4 pixels).  Input rows are pitch-padded to 64 bytes, so the word loads
    // never leave the row.  BORDER_REFLECT_101 at the image frame (cv::Sobel).
    const LineGeom& g = *gp;
    const int img = blockIdx.y;
    const int wo = (g.W + 7) >> 3;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= wo * g.H) return;
    const int y = t / wo, x0 = (t - y * wo) * 8;
    const uint8_t* b = blur + (size_t)img * g.pitchW * g.H;
    const int ym = y == 0 ? 1 : y - 1, yp = y == g.H - 1 ? g.H - 2 : y + 1;
    int v[3][10];            // rows (ym, y, yp) x columns x0-1 .. x0+8
    const int rows[3] = {ym, y, yp};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint8_t* row = b + (size_t)rows[r] * g.pitchW;
        const uint32_t w1 = *reinterpret_cast<const uint32_t*>(row + x0);
        const uint32_t w2 = x0 + 4 < g.pitchW ? *reinterpret_cast<const uint32_t*>(row + x0 + 4) : 0u;
        const uint32_t w0 = x0 > 0 ? *reinterpret_cast<const uint32_t*>(row + x0 - 4) : 0u;
        const uint32_t w3 = x0 + 8 < g.pitchW ? *reinterpret_cast<const uint32_t*>(row + x0 + 8) : 0u;
        v[r][0] = (int)(w0 >> 24);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[r][1 + k] = (int)((w1 >> (8 * k)) & 0xffu); v[r][5 + k] = (int)((w2 >> (8 * k)) & 0xffu); }
        v[r][9] = (int)(w3 & 0xffu);
        if (x0 == 0) v[r][0] = v[r][2];                                  // x = 0: x-1 -> 1
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (x0 + k == g.W - 1) v[r][k + 2] = v[r][k];                // x = W-1: x+1 -> W-2
    }
    uint32_t o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int dx = (v[0][k + 2] - v[0][k]) + 2 * (v[1][k + 2] - v[1][k]) + (v[2][k + 2] - v[2][k]);
        const int dy = (v[2][k] + 2 * v[2][k + 1] + v[2][k + 2]) - (v[0][k] + 2 * v[0][k + 1] + v[0][k + 2]);
        o[k] = ((uint32_t)dx & 0xffffu) | ((uint32_t)dy << 16);
    }
    uint32_t* out = dxdy + (size_t)img * g.pitchD * g.H + (size_t)y * g.pitchD + x0;      // 16-byte aligned: pitchD and x0 are multiples of 4
    if (x0 < g.pitchD) *reinterpret_cast<uint4*>(out) = make_uint4(o[0], o[1], o[2], o[3]);
    if (x0 + 4 < g.pitchD) *reinterpret_cast<uint4*>(out + 4) = make_uint4(o[4], o[5], o[6], o[7]);
}

// Per line, once: the direction vector dL = (cos, sin)(direction) and the start of each of the 63 support-region rows -- a sequential float chain from the
// first row's start (sCorX0 -= dL[1]; sCorY0 += dL[0] per row, computeLBD :1143-1150).  One thread per line (round 4): with one thread per (line, row)
// every one of the 63 threads evaluated the double-precision sin / cos and walked the chain up to its own row -- a third of k_lbd_rows's instructions.
// starts: [image][line][64] float4 slots of which .x/.y = (sCorX0, sCorY0) of row h and, in slot 63, (dL0, dL1).
__global__ __launch_bounds__(64) void k_lbd_prep(const LineGeom* __restrict__ gp, const olf_keyline* __restrict__ kls, const int* __restrict__ counts,
                                                 float2* __restrict__ starts)
{
    const LineGeom& g = *gp;
    const int img = blockIdx.y, li = blockIdx.x * 64 + threadIdx.x;
    if (li >= counts[img]) return;
    const olf_keyline kl = kls[(size_t)img * g.outCap + li];
    const short heightOfLSP = 63;
    const short lengthOfLSP = (short)kl.numOfPixels;
    const short halfWidth = (short)((lengthOfLSP - 1) / 2);
    const short halfHeight = (short)((heightOfLSP - 1) / 2);
    const float midX = (float)(0.5 * (double)f_add(kl.sPointInOctaveX, kl.ePointInOctaveX));
    const float midY = (float)(0.5 * (double)f_add(kl.sPointInOctaveY, kl.ePointInOctaveY));
    // convention C.6: the C functions on doubles, or the float overloads (glibc's cosf / sinf bit for bit, device_math.hpp; kl.angle lies in [-pi, pi])
    const float dL0 = g.libmFloat ? glibc_cosf(kl.angle) : (float)cos((double)kl.angle);
    const float dL1 = g.libmFloat ? glibc_sinf(kl.angle) : (float)sin((double)kl.angle);
    float sCorX0 = f_add(f_add(f_mul(-dL0, (float)halfWidth), f_mul(dL1, (float)halfHeight)), midX);
    float sCorY0 = f_add(f_sub(f_mul(-dL1, (float)halfWidth), f_mul(dL0, (float)halfHeight)), midY);
    float2* o = starts + ((size_t)img * g.outCap + li) * 64;
    for (int h = 0; h < 63; ++h) {
        o[h] = make_float2(sCorX0, sCorY0);
        sCorX0 = f_sub(sCorX0, dL1); sCorY0 = f_add(sCorY0, dL0);
    }
    o[63] = make_float2(dL0, dL1);
}

// one wave per line, one lane per support-region row: the four weighted row sums of computeLBD (:1143-1196).
// The sums of a row are sequential float chains over its samples, so a lane walks its row; with the lanes of a wave being the 63 rows of a line, a sample
// step of a mostly HORIZONTAL line touches 63 image rows -- 63 cache lines per load instruction, and the kernel ran at the texture path's rate of one line per
// cycle (12 ms per 6144 x 500 lines, 24 GB of useful samples).  For such lines the 16 samples x 63 rows of a step are loaded TRANSPOSED: every lane writes its
// row's 16 pixel offsets to LDS, the wave loads them as (4 rows x 16 consecutive samples) per instruction -- 4 to 8 cache lines -- and hands the values back
// through the same LDS words.  Steep lines keep the direct form (their rows are horizontal: 63 neighbouring pixels per instruction as it is).
constexpr int LR_U = 16, LR_P = LR_U + 1;      // samples per step; row pitch of the LDS tile in words (17: lanes = rows and lanes = samples both hit distinct banks)
template <bool LEAN>      // LEAN: the image is smaller than 32768 - 256 pixels a side (chosen at the launch): the lean form of the sample coordinates below
__global__ __launch_bounds__(256) void k_lbd_rows(const LineGeom* __restrict__ gp, const uint32_t* __restrict__ dxdyAll,
                                                  const olf_keyline* __restrict__ kls, const int* __restrict__ counts,
                                                  const float2* __restrict__ starts, float4* __restrict__ rowSums, int transposeFlat)
{
    __shared__ uint32_t s_t[4][64 * LR_P];
    const LineGeom& g = *gp;
    const int img = blockIdx.y, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = blockIdx.x * 4 + wv;
    if (li >= counts[img]) return;
    const bool rowAct = lane < 63;
    const int hID = rowAct ? lane : 62;          // (lane 63 shadows row 62: valid coordinates, nothing stored)
    const uint32_t* dxdy = dxdyAll + (size_t)img * g.pitchD * g.H;
    const short lengthOfLSP = (short)kls[(size_t)img * g.outCap + li].numOfPixels;
    const int realWidth = g.pitchD;      // row stride of the gradient image (the reference's realWidth, padded to a multiple of 4 pixels)
    const short imageWidth = (short)(g.W - 1), imageHeight = (short)(g.H - 1);
    const float2* st = starts + ((size_t)img * g.outCap + li) * 64;
    const float2 dL = st[63], s0 = st[hID];
    const float dL0 = dL.x, dL1 = dL.y;
    const float dO0 = -dL1, dO1 = dL0;
    float sCorX = s0.x, sCorY = s0.y;
    float pgdL = 0, ngdL = 0, pgdO = 0, ngdO = 0;
    const bool flat = transposeFlat && fabsf(dL0) >= fabsf(dL1);      // (wave-uniform: one line per wave)
    uint32_t* tb = s_t[wv];
    // the sample coordinates are a cheap sequential float chain, the sums a sequential one on the loaded values: 16 samples are
    // addressed and loaded per step so that their loads are in flight together, then accumulated in order
    constexpr int U = LR_U;
    for (int w0 = 0; w0 < lengthOfLSP; w0 += U) {
        uint32_t p[U];
        int off[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int xCor, yCor;
            if (LEAN) {
                // round() = half away from zero, (short), clamp to [0, size - 1] -- with coordinates that stay within 256 pixels of the image (a support region is
                // 63 rows around a line inside the image) the cast is the identity, every negative value clamps to 0 whichever way its tie went, and a positive
                // tie is round-to-nearest-even bumped when it went down: v_rndne, the exact remainder, one compare, one select, v_med3 -- 7 instructions per
                // coordinate where the general form takes 12 (k_lbd_rows runs alone at the end of the step: its instructions are the step's time)
                const float rx = __builtin_rintf(sCorX), ry = __builtin_rintf(sCorY);
                const float bx = f_sub(sCorX, rx) == 0.5f ? f_add(rx, 1.0f) : rx, by = f_sub(sCorY, ry) == 0.5f ? f_add(ry, 1.0f) : ry;
                xCor = min(max((int)bx, 0), (int)imageWidth); yCor = min(max((int)by, 0), (int)imageHeight);
            } else {
            int tc = (int)(short)roundf(sCorX);
            xCor = tc < 0 ? 0 : (tc > imageWidth ? imageWidth : tc);
            tc = (int)(short)roundf(sCorY);
            yCor = tc < 0 ? 0 : (tc > imageHeight ? imageHeight : tc);
            }
            off[u] = yCor * realWidth + xCor;           // (coordinates past the row end are clamped into the image, the value is unused)
            sCorX = f_add(sCorX, dL0);
            sCorY = f_add(sCorY, dL1);
        }
        if (flat) {
#pragma unroll
            for (int u = 0; u < U; ++u) tb[lane * LR_P + u] = (uint32_t)off[u];
            __builtin_amdgcn_wave_barrier();
            const int a0 = (lane >> 4) * LR_P + (lane & 15);
            uint32_t v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = dxdy[tb[a0 + 4 * k * LR_P]];
#pragma unroll
            for (int k = 0; k < 16; ++k) tb[a0 + 4 * k * LR_P] = v[k];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < U; ++u) p[u] = tb[lane * LR_P + u];
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) p[u] = dxdy[off[u]];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (w0 + u < lengthOfLSP) {
                const float dx = (float)(int)(int16_t)(p[u] & 0xffffu), dy = (float)(int)(int16_t)(p[u] >> 16);
                const float gDL = f_add(f_mul(dx, dL0), f_mul(dy, dL1));
                const float gDO = f_add(f_mul(dx, dO0), f_mul(dy, dO1));
                // "if (g > 0) p += g; else n -= g;": adding / subtracting a zero is exact, so both sums can be updated unconditionally
                pgdL = f_add(pgdL, fmaxf(gDL, 0.f)); ngdL = f_sub(ngdL, fminf(gDL, 0.f));
                pgdO = f_add(pgdO, fmaxf(gDO, 0.f)); ngdO = f_sub(ngdO, fminf(gDO, 0.f));
            }
        }
    }
    if (rowAct) {
        const float cg = g.gaussCoefG[hID];
        rowSums[((size_t)img * g.outCap + li) * 63 + hID] = make_float4(f_mul(cg, pgdL), f_mul(cg, ngdL), f_mul(cg, pgdO), f_mul(cg, ngdO));
    }
}

// one thread per line: 63 rows -> 9 bands (:1201-1240), means/stds (:1256-1280), normalise / clip / renormalise
// (:1283-1341), 32 x binaryConversion (:401-412, :645-667)
__global__ __launch_bounds__(64) void k_lbd_desc(const LineGeom* __restrict__ gp, const float4* __restrict__ rowSums,
                                                 const int* __restrict__ counts, uint8_t* __restrict__ desc)
{
    const LineGeom& g = *gp;
    const int img = blockIdx.y;
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li >= counts[img]) return;
    const float4* rs = rowSums + ((size_t)img * g.outCap + li) * 63;
    float bs[9][8];   // pgdL, ngdL, pgdL2, ngdL2, pgdO, ngdO, pgdO2, ngdO2
#pragma unroll
    for (int b = 0; b < 9; ++b)
#pragma unroll
        for (int k = 0; k < 8; ++k) bs[b][k] = 0.f;
    for (int hID = 0; hID < 63; ++hID) {
        const float4 r = rs[hID];
        const float pL = r.x, nL = r.y, pO = r.z, nO = r.w;
        const float pL2 = f_mul(pL, pL), nL2 = f_mul(nL, nL), pO2 = f_mul(pO, pO), nO2 = f_mul(nO, nO);
        const int band = hID / 7, m = hID - band * 7;
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            // order of the reference: own band (coef L[m+7]), band above (L[m+14]), band below (L[m])
            const int b = which == 0 ? band : (which == 1 ? band - 1 : band + 1);
            if (b < 0 || b >= 9) continue;
            const float c = g.gaussCoefL[which == 0 ? m + 7 : (which == 1 ? m + 14 : m)];
            const float cc = f_mul(c, c);
#pragma unroll
            for (int bb = 0; bb < 9; ++bb)
                if (bb == b) {
                    bs[bb][0] = f_add(bs[bb][0], f_mul(c, pL));
                    bs[bb][1] = f_add(bs[bb][1], f_mul(c, nL));
                    bs[bb][2] = f_add(bs[bb][2], f_mul(cc, pL2));
                    bs[bb][3] = f_add(bs[bb][3], f_mul(cc, nL2));
                    bs[bb][4] = f_add(bs[bb][4], f_mul(c, pO));
                    bs[bb][5] = f_add(bs[bb][5], f_mul(c, nO));
                    bs[bb][6] = f_add(bs[bb][6], f_mul(cc, pO2));
                    bs[bb][7] = f_add(bs[bb][7], f_mul(cc, nO2));
                }
        }
    }
    float d[72];
    const float invN2 = (float)(1.0 / (7 * 2.0)), invN3 = (float)(1.0 / (7 * 3.0));
#pragma unroll
    for (int b = 0; b < 9; ++b) {
        const float invN = (b == 0 || b == 8) ? invN2 : invN3;
        float temp = f_mul(bs[b][0], invN);
        d[8 * b] = temp;
        d[8 * b + 4] = (float)sqrt((double)f_sub(f_mul(bs[b][2], invN), f_mul(temp, temp)));
        temp = f_mul(bs[b][1], invN);
        d[8 * b + 1] = temp;
        d[8 * b + 5] = (float)sqrt((double)f_sub(f_mul(bs[b][3], invN), f_mul(temp, temp)));
        temp = f_mul(bs[b][4], invN);
        d[8 * b + 2] = temp;
        d[8 * b + 6] = (float)sqrt((double)f_sub(f_mul(bs[b][6], invN), f_mul(temp, temp)));
        temp = f_mul(bs[b][5], invN);
        d[8 * b + 3] = temp;
        d[8 * b + 7] = (float)sqrt((double)f_sub(f_mul(bs[b][7], invN), f_mul(temp, temp)));
    }
    float tempM = 0, tempS = 0;
#pragma unroll
    for (int b = 0; b < 9; ++b) {
#pragma unroll
        for (int k = 0; k < 4; ++k) tempM = f_add(tempM, f_mul(d[8 * b + k], d[8 * b + k]));
#pragma unroll
        for (int k = 4; k < 8; ++k) tempS = f_add(tempS, f_mul(d[8 * b + k], d[8 * b + k]));
    }
    if (g.libmFloat) { tempM = f_div(1.0f, sqrtf(tempM)); tempS = f_div(1.0f, sqrtf(tempS)); }      // convention C.6: float sqrt, float divide
    else { tempM = (float)(1 / sqrt((double)tempM)); tempS = (float)(1 / sqrt((double)tempS)); }
#pragma unroll
    for (int b = 0; b < 9; ++b) {
#pragma unroll
        for (int k = 0; k < 4; ++k) d[8 * b + k] = f_mul(d[8 * b + k], tempM);
#pragma unroll
        for (int k = 4; k < 8; ++k) d[8 * b + k] = f_mul(d[8 * b + k], tempS);
    }
#pragma unroll
    for (int i = 0; i < 72; ++i)
        if ((double)d[i] > 0.4) d[i] = (float)0.4;
    float temp = 0;
#pragma unroll
    for (int i = 0; i < 72; ++i) temp = f_add(temp, f_mul(d[i], d[i]));
    temp = g.libmFloat ? f_div(1.0f, sqrtf(temp)) : (float)(1 / sqrt((double)temp));
#pragma unroll
    for (int i = 0; i < 72; ++i) d[i] = f_mul(d[i], temp);
    uint8_t* o = desc + ((size_t)img * g.outCap + li) * OLF_DESC_BYTES;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        const int p = c_bandPairs[2 * c], q = c_bandPairs[2 * c + 1];
        unsigned r = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float f1 = 0, f2 = 0;
#pragma unroll
            for (int bb = 0; bb < 9; ++bb) { if (bb == p) f1 = d[8 * bb + i]; if (bb == q) f2 = d[8 * bb + i]; }
            if (f1 > f2) r += 1u << i;
        }
        o[c] = (uint8_t)r;
    }
}

// LBD gradient images: GaussianBlur(5x5, sigma 1) then Sobel (computeGaussianPyramid / computeSobel) -- they depend on the input images only, so the
// fused entry runs them on the ORB stream in the shadow of the seed ordering (api.cpp, schedule 5)
// OLF_LBD_T=0: every line through the direct loads (A/B)
static int lbd_rows_transpose() { static const int v = !(getenv("OLF_LBD_T") && atoi(getenv("OLF_LBD_T")) == 0); return v; }

int launch_lbd_dense(const LineGeom& g, const LineDeviceBufs& b, const uint8_t* d_in, int in_pitch, int n_images, hipStream_t s)
{
    OLF_TRY_RC(launch_gauss7_img(d_in, in_pitch, (size_t)in_pitch * g.H, b.lbdBlur, g.pitchW, (size_t)g.pitchW * g.H, g.W, g.H, g, 1, n_images, s));
    hipLaunchKernelGGL(k_sobel3, dim3((((g.W + 7) >> 3) * g.H + 255) / 256, n_images), dim3(256), 0, s, b.lbdBlur, b.dxdy, b.geom);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

int launch_line_select_lbd(const LineGeom& g, const LineDeviceBufs& b, const uint8_t* d_in, int in_pitch, int n_images,
                           olf_keyline* d_kls, uint8_t* d_desc, int* d_counts, hipStream_t s, bool denseDone)
{
    int sortN = 64;
    while (sortN < g.maxDetect) sortN <<= 1;
    // (longer lists than LS_LDS_KEYS are sorted in the key buffer, which is dead once the segments have been emitted: 4 Ps bytes per image, and
    // maxDetect <= Ps / 48 keeps 8 bytes x the next power of two below that)
    hipLaunchKernelGGL(k_line_select, dim3(n_images), dim3(256), std::min(sortN, LS_LDS_KEYS) * sizeof(unsigned long long), s, b.geom, b.rawLines, b.rawCount, d_kls,
                       d_counts, reinterpret_cast<unsigned long long*>(b.keysA), (size_t)g.Ps / 2);
    if (!denseDone) OLF_TRY_RC(launch_lbd_dense(g, b, d_in, in_pitch, n_images, s));
    hipLaunchKernelGGL(k_lbd_prep, dim3((g.outCap + 63) / 64, n_images), dim3(64), 0, s, b.geom, d_kls, d_counts, reinterpret_cast<float2*>(b.lbdStarts));
    if (g.W + 256 < 32768 && g.H + 256 < 32768)
        hipLaunchKernelGGL(k_lbd_rows<true>, dim3((g.outCap + 3) / 4, n_images), dim3(256), 0, s, b.geom, b.dxdy, d_kls, d_counts,
                           reinterpret_cast<const float2*>(b.lbdStarts), reinterpret_cast<float4*>(b.rowSums), lbd_rows_transpose());
    else
        hipLaunchKernelGGL(k_lbd_rows<false>, dim3((g.outCap + 3) / 4, n_images), dim3(256), 0, s, b.geom, b.dxdy, d_kls, d_counts,
                           reinterpret_cast<const float2*>(b.lbdStarts), reinterpret_cast<float4*>(b.rowSums), lbd_rows_transpose());
    hipLaunchKernelGGL(k_lbd_desc, dim3((g.outCap + 63) / 64, n_images), dim3(64), 0, s, b.geom, reinterpret_cast<const float4*>(b.rowSums),
                       d_counts, d_desc);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

// BinaryDescriptor::compute on key lines already in d_kls/d_counts (no LSD, no selection)
int launch_lbd_only(const LineGeom& g, const LineDeviceBufs& b, const uint8_t* d_in, int in_pitch, int n_images, const olf_keyline* d_kls,
                    uint8_t* d_desc, const int* d_counts, hipStream_t s)
{
    OLF_TRY_RC(launch_lbd_dense(g, b, d_in, in_pitch, n_images, s));
    hipLaunchKernelGGL(k_lbd_prep, dim3((g.outCap + 63) / 64, n_images), dim3(64), 0, s, b.geom, d_kls, d_counts, reinterpret_cast<float2*>(b.lbdStarts));
    if (g.W + 256 < 32768 && g.H + 256 < 32768)
        hipLaunchKernelGGL(k_lbd_rows<true>, dim3((g.outCap + 3) / 4, n_images), dim3(256), 0, s, b.geom, b.dxdy, d_kls, d_counts,
                           reinterpret_cast<const float2*>(b.lbdStarts), reinterpret_cast<float4*>(b.rowSums), lbd_rows_transpose());
    else
        hipLaunchKernelGGL(k_lbd_rows<false>, dim3((g.outCap + 3) / 4, n_images), dim3(256), 0, s, b.geom, b.dxdy, d_kls, d_counts,
                           reinterpret_cast<const float2*>(b.lbdStarts), reinterpret_cast<float4*>(b.rowSums), lbd_rows_transpose());
    hipLaunchKernelGGL(k_lbd_desc, dim3((g.outCap + 63) / 64, n_images), dim3(64), 0, s, b.geom, reinterpret_cast<const float4*>(b.rowSums),
                       d_counts, d_desc);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
