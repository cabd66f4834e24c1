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

#ifndef OLF_DESC_KPW
#define OLF_DESC_KPW 1
#endif
constexpr int DESC_KPW = OLF_DESC_KPW;      // key point slots per wave: the per-lane constants (16 pattern floats, 8 disc weight words) and the block's set-up are paid once for four

__global__ __launch_bounds__(256) void k_describe(const OrbGeom* __restrict__ gp, const uint8_t* __restrict__ pyr,
                                                  const uint8_t* __restrict__ blur, const uint32_t* __restrict__ lvlKp,
                                                  const int* __restrict__ lvlCount, olf_keypoint* __restrict__ kps,
                                                  uint8_t* __restrict__ desc, int* __restrict__ counts, int out_cap,
                                                  int* __restrict__ status)
{
    OLF_SET_GUEST_PRIO();
    constexpr int PR = 18, PW = 2 * PR + 1, PDW = 10;      // patch radius (|rotated pattern coordinate| <= round(13 * sqrt 2) = 18), 37 rows of 10 dwords
    __shared__ float4 s_patf[256];                        // the test pattern as floats (x0, y0, x1, y1): one 16-byte LDS read per test, no unpacking
    __shared__ uint32_t s_w0[256], s_w1[256];             // IC_Angle disc as byte weights per (row, dword) slot: 1 / (u + 16) inside, 0 outside
    __shared__ uint32_t s_patch[4][PW * PDW];
    const OrbGeom& g = *gp;
    const int img = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    {
        const uint32_t pw = reinterpret_cast<const uint32_t*>(c_pattern)[threadIdx.x];
        s_patf[threadIdx.x] = make_float4((float)(int8_t)(pw & 0xff), (float)(int8_t)((pw >> 8) & 0xff), (float)(int8_t)((pw >> 16) & 0xff), (float)(int8_t)(pw >> 24));
    }
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
    const int nlevels = g.nlevels, kpTotal = g.kpTotal;
    const int* lc = lvlCount + img * nlevels;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int tot = 0;
        for (int l = 0; l < nlevels; ++l) tot += lc[l];
        if (tot > out_cap) { atomicOr(status, 4); tot = out_cap; }
        counts[img] = tot;
    }
    // (the per-lane constants -- 16 pattern floats, 8 disc weight words -- are read from LDS where they are used: held in registers across the key point they
    // cost 24 VGPRs, i.e. two of the eight waves a SIMD can hold, and the kernel lives on its occupancy)
    // XCD-aware numbering: workgroups are dealt round-robin to the 8 XCDs (each with an L2 of its own), so the blocks x, x + 8, x + 16, ... of an image share
    // one; they take CONSECUTIVE groups of key point slots -- neighbours in the octree's output order, whose 37 x 37 patches overlap -- instead of every eighth
    // (OLF_DESC_XCD=0 in the environment of the build's A/B: -DOLF_DESC_XCD=0)
#ifndef OLF_DESC_XCD
#define OLF_DESC_XCD 1
#endif
    const int nbx = (int)gridDim.x, per = nbx >> 3;
    const int vblock = (OLF_DESC_XCD && (int)blockIdx.x < per * 8) ? ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int r6 = lane / PDW, c10 = lane - r6 * PDW;      // staging: lanes 0 .. 59 take six rows of ten dwords per step
    uint32_t* patch = s_patch[wv];
#pragma unroll 1
    for (int q = 0; q < DESC_KPW; ++q) {
        const int slot = (vblock * 4 + wv) * DESC_KPW + q;
        if (slot >= kpTotal) break;
        // which level the slot belongs to, its rank there and its place in the output: lane l holds level l's first slot and count (ONE round trip of two vector
        // loads beside the key point word -- as loops over scalar loads, "while (slot >= lv[level + 1].kpBase)" and "outIdx += lc[l]", these were up to fourteen
        // dependent memory round trips in front of everything else a wave does)
        const uint32_t p = lvlKp[(size_t)img * kpTotal + slot];
        const bool lv_lane = lane < nlevels;
        const int base_l = lv_lane ? g.lv[lane].kpBase : 0x7fffffff, cnt_l = lv_lane ? lc[lane] : 0;
        const int level = (int)__popcll(wave_vote(lv_lane && lane > 0 && slot >= base_l));
        const int i = slot - __builtin_amdgcn_readlane(base_l, level);
        if (i >= __builtin_amdgcn_readlane(cnt_l, level)) continue;
        const int outIdx = i + wave_sum_i32(lane < level ? cnt_l : 0);
        if (outIdx >= out_cap) continue;
        const LevelGeom& L = g.lv[level];
        const int cx = (int)(p >> 20) + kMinBorder, cy = (int)((p >> 8) & 0xfff) + kMinBorder, score = (int)(p & 0xff);
        const int pitch = L.pitch;

        // the blurred patch of the descriptor depends on the position only: rows cy-18 .. cy+18, bytes x0 .. x0+39 (x0 = (cx-18) rounded down to a dword) are
        // requested here, six rows of ten dwords per step, and travel while the orientation is computed (the kernel is bound by its dependent memory round
        // trips -- key point word, disc rows, patch rows -- at the occupancy 74 VGPRs allow, not by its instructions)
        const uint8_t* bl = blur + (size_t)img * g.pyrBytes + L.offset;
        const int x0 = (cx - PR) & ~3, xs = cx - x0;          // patch column of the key point
        uint32_t stg[(PW + 5) / 6];
        {
            const int xo = min(x0 + 4 * c10, pitch - 4);             // the clamped dword only holds bytes no test can reach
            const uint32_t stBase = (uint32_t)((cy - PR + r6) * pitch + xo);
#pragma unroll
            for (int k = 0; k < (PW + 5) / 6; ++k) {
                stg[k] = 0u;
                if (lane < 60 && 6 * k + r6 < PW) stg[k] = *reinterpret_cast<const uint32_t*>(bl + (stBase + (uint32_t)(6 * k * pitch)));
            }
        }

        // ---- IC_Angle on the un-blurred level
        // m10 = sum u*I, m01 = sum v*I over the disc (:79-106).  The 31 rows are read as 8 (unaligned) dwords each; with the byte weights
        // above a dword contributes v_dot4_u32_u8(I, w0) to the row sum and v_dot4_u32_u8(I, w1) to sum (u+16)*I.
        const uint8_t* im = pyr + (size_t)img * g.pyrBytes + L.offset;
        int m10 = 0, m01 = 0, sI = 0;
        const uint32_t icBase = (uint32_t)((cy - kHalfPatch + (lane >> 3)) * pitch + cx - 16 + 4 * (lane & 7));
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            if (pass * 64 + lane < (2 * kHalfPatch + 1) * 8) {
                uint32_t I;
                __builtin_memcpy(&I, im + (icBase + (uint32_t)(pass * 8 * pitch)), 4);
                const int rs = (int)__builtin_amdgcn_udot4(I, s_w0[pass * 64 + lane], 0u, false);
                m10 = (int)__builtin_amdgcn_udot4(I, s_w1[pass * 64 + lane], (uint32_t)m10, false);
                sI += rs;
                m01 += (pass * 8 + (lane >> 3) - kHalfPatch) * rs;
            }
        }
        m10 -= 16 * sI;
        m10 = wave_sum_i32(m10);
        m01 = wave_sum_i32(m01);
        const float angle = dev_fastAtan2((float)m01, (float)m10);

        // ---- steered BRIEF on the blurred level
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        const float rad = __fmul_rn(angle, factorPI);
        const float a = glibc_cosf(rad), b = glibc_sinf(rad);
        __builtin_amdgcn_wave_barrier();                      // (the previous key point's tests have read the patch)
#pragma unroll
        for (int k = 0; k < (PW + 5) / 6; ++k)
            if (lane < 60 && 6 * k + r6 < PW) patch[60 * k + lane] = stg[k];
        __builtin_amdgcn_wave_barrier();
        const uint8_t* pb = reinterpret_cast<const uint8_t*>(patch) + PR * (PDW * 4) + xs;
        unsigned long long bits[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 pt = s_patf[j * 64 + lane];
            const int r0 = __float2int_rn(__fadd_rn(__fmul_rn(pt.x, b), __fmul_rn(pt.y, a)));
            const int c0 = __float2int_rn(__fsub_rn(__fmul_rn(pt.x, a), __fmul_rn(pt.y, b)));
            const int r1 = __float2int_rn(__fadd_rn(__fmul_rn(pt.z, b), __fmul_rn(pt.w, a)));
            const int c1 = __float2int_rn(__fsub_rn(__fmul_rn(pt.z, a), __fmul_rn(pt.w, b)));
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
}

int launch_orb_describe(const OrbGeom& g, const OrbDeviceBufs& b, int n_images, olf_keypoint* d_kps, uint8_t* d_desc,
                        int* d_counts, int out_cap, hipStream_t s)
{
    hipLaunchKernelGGL(k_describe, dim3((g.kpTotal + 4 * DESC_KPW - 1) / (4 * DESC_KPW), n_images), dim3(256), 0, s, b.geom, b.pyr, b.blur, b.lvlKp, b.lvlCount,
                       d_kps, d_desc, d_counts, out_cap, b.status);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
