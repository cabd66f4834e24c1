// orb_describe.hip -- per-key-point stages of ORBextractor::operator() (reference
// src/ORBextractor.cc): IC_Angle / computeOrientation (:79-106, :474-481, cv::fastAtan2 App. A.5),
// computeOrbDescriptor (:110-149) on the blurred level, the scale-up of pt (:1097-1103) and the
// level-ascending concatenation (:1078-1106).  One 64-lane wave per key point: the 749-pixel
// intensity-centroid disc is summed 2 rows per pass, the 256 steered-BRIEF tests are 4 ballots.
// The kernel is bound by the texture-address / L1 pipeline (hundreds of cache-line lookups per key point when every lane gathers single
// bytes), so the 37x37 blurred patch the rotated pattern can reach is staged in LDS with row-coalesced 32-bit loads and gathered from
// there; the test pattern (one 32-bit word per test) and umax sit in LDS as well.
#include "olf_internal.hpp"
#include "device_math.hpp"

namespace olf {

__constant__ __attribute__((aligned(16))) int8_t c_pattern[1024] = {
#include "orb_pattern_31.inc"
};

__global__ __launch_bounds__(256) void k_describe(const OrbGeom* __restrict__ gp, const uint8_t* __restrict__ pyr,
                                                  const uint8_t* __restrict__ blur, const uint32_t* __restrict__ lvlKp,
                                                  const int* __restrict__ lvlCount, olf_keypoint* __restrict__ kps,
                                                  uint8_t* __restrict__ desc, int* __restrict__ counts, int out_cap,
                                                  int* __restrict__ status)
{
    OLF_SET_GUEST_PRIO();
    constexpr int PR = 18, PW = 2 * PR + 1, PDW = 10;      // patch radius (|rotated pattern coordinate| <= round(13 * sqrt 2) = 18), 37 rows of 10 dwords
    __shared__ uint32_t s_pat[256];
    __shared__ uint32_t s_w0[256], s_w1[256];             // IC_Angle disc as byte weights per (row, dword) slot: 1 / (u + 16) inside, 0 outside
    __shared__ uint32_t s_patch[4][PW * PDW];
    const OrbGeom& g = *gp;
    const int img = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slot = blockIdx.x * 4 + wv;
    s_pat[threadIdx.x] = reinterpret_cast<const uint32_t*>(c_pattern)[threadIdx.x];
    {   // slot t = (row r = t / 8, dword j = t % 8) covers columns u = 4j - 16 .. 4j - 13 of row v = r - 15 (column -16 is padding)
        const int r = threadIdx.x >> 3, j = threadIdx.x & 7, v = r - kHalfPatch;
        uint32_t w0 = 0, w1 = 0;
        if (r <= 2 * kHalfPatch) {
            const int d = g.umax[v < 0 ? -v : v];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int u = 4 * j + b - 16;
                if (u >= -d && u <= d) { w0 |= 1u << (8 * b); w1 |= (uint32_t)(u + 16) << (8 * b); }
            }
        }
        s_w0[threadIdx.x] = w0; s_w1[threadIdx.x] = w1;
    }
    __syncthreads();
    const int* lc = lvlCount + img * g.nlevels;
    if (slot == 0 && lane == 0) {
        int tot = 0;
        for (int l = 0; l < g.nlevels; ++l) tot += lc[l];
        if (tot > out_cap) { atomicOr(status, 4); tot = out_cap; }
        counts[img] = tot;
    }
    if (slot >= g.kpTotal) return;
    int level = 0;
    while (level + 1 < g.nlevels && slot >= g.lv[level + 1].kpBase) ++level;
    const LevelGeom& L = g.lv[level];
    const int i = slot - L.kpBase;
    if (i >= lc[level]) return;
    int outIdx = i;
    for (int l = 0; l < level; ++l) outIdx += lc[l];
    if (outIdx >= out_cap) return;
    const uint32_t p = lvlKp[(size_t)img * g.kpTotal + slot];
    const int cx = (int)(p >> 20) + kMinBorder, cy = (int)((p >> 8) & 0xfff) + kMinBorder, score = (int)(p & 0xff);

    // ---- IC_Angle on the un-blurred level
    const uint8_t* im = pyr + (size_t)img * g.pyrBytes + L.offset;
    // m10 = sum u*I, m01 = sum v*I over the disc (:79-106).  The 31 rows are read as 8 (unaligned) dwords each; with the byte weights
    // above a dword contributes v_dot4_u32_u8(I, w0) to the row sum and v_dot4_u32_u8(I, w1) to sum (u+16)*I.
    int m10 = 0, m01 = 0, sI = 0;
    const uint8_t* ic = im + (size_t)(cy - kHalfPatch) * L.pitch + cx - 16;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int slot = pass * 64 + lane;
        if (slot < (2 * kHalfPatch + 1) * 8) {
            const int r = slot >> 3, j = slot & 7;
            uint32_t I;
            __builtin_memcpy(&I, ic + (size_t)r * L.pitch + 4 * j, 4);
            const int rs = (int)__builtin_amdgcn_udot4(I, s_w0[slot], 0u, false);
            m10 = (int)__builtin_amdgcn_udot4(I, s_w1[slot], (uint32_t)m10, false);
            sI += rs;
            m01 += (r - kHalfPatch) * rs;
        }
    }
    m10 -= 16 * sI;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m10 += __shfl_xor(m10, o);
        m01 += __shfl_xor(m01, o);
    }
    const float angle = dev_fastAtan2((float)m01, (float)m10);

    // ---- steered BRIEF on the blurred level
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float rad = __fmul_rn(angle, factorPI);
    const float a = glibc_cosf(rad), b = glibc_sinf(rad);
    // stage rows cy-18 .. cy+18, bytes x0 .. x0+39 (x0 = (cx-18) rounded down to a dword) of the blurred level
    const uint8_t* bl = blur + (size_t)img * g.pyrBytes + L.offset;
    const int x0 = (cx - PR) & ~3, xs = cx - x0;          // patch column of the key point
    uint32_t* patch = s_patch[wv];
#pragma unroll
    for (int k = 0; k < (PW * PDW + 63) / 64; ++k) {
        const int idx = k * 64 + lane;
        if (idx < PW * PDW) {
            const int row = idx / PDW, col = idx - row * PDW;
            const int xo = min(x0 + 4 * col, L.pitch - 4);             // the clamped dword only holds bytes no test can reach
            patch[idx] = *reinterpret_cast<const uint32_t*>(bl + (size_t)(cy - PR + row) * L.pitch + xo);
        }
    }
    __builtin_amdgcn_wave_barrier();
    const uint8_t* pb = reinterpret_cast<const uint8_t*>(patch) + PR * (PDW * 4) + xs;
    unsigned long long bits[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t pw = s_pat[j * 64 + lane];
        const float x0f = (float)(int8_t)(pw & 0xff), y0f = (float)(int8_t)((pw >> 8) & 0xff);
        const float x1f = (float)(int8_t)((pw >> 16) & 0xff), y1f = (float)(int8_t)(pw >> 24);
        const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(x0f, b), __fmul_rn(y0f, a)));
        const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(x0f, a), __fmul_rn(y0f, b)));
        const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(x1f, b), __fmul_rn(y1f, a)));
        const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(x1f, a), __fmul_rn(y1f, b)));
        const int t0 = pb[r0 * (PDW * 4) + c0], t1 = pb[r1 * (PDW * 4) + c1];
        bits[j] = wave_vote(t0 < t1);
    }
    if (lane == 0) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(desc + ((size_t)img * out_cap + outIdx) * OLF_DESC_BYTES);
        d[0] = bits[0]; d[1] = bits[1]; d[2] = bits[2]; d[3] = bits[3];
        olf_keypoint k;
        k.x = (float)cx; k.y = (float)cy;
        if (level != 0) { k.x = __fmul_rn(k.x, L.scale); k.y = __fmul_rn(k.y, L.scale); }
        k.size = (float)L.patch_size; k.angle = angle; k.response = (float)score; k.octave = level; k.class_id = -1;
        kps[(size_t)img * out_cap + outIdx] = k;
    }
}

int launch_orb_describe(const OrbGeom& g, const OrbDeviceBufs& b, int n_images, olf_keypoint* d_kps, uint8_t* d_desc,
                        int* d_counts, int out_cap, hipStream_t s)
{
    hipLaunchKernelGGL(k_describe, dim3((g.kpTotal + 3) / 4, n_images), dim3(256), 0, s, b.geom, b.pyr, b.blur, b.lvlKp, b.lvlCount,
                       d_kps, d_desc, d_counts, out_cap, b.status);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
