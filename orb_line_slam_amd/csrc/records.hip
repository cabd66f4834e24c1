// records.hip -- the trimmed wire record of a batch of stereo frames (the payload of the multi-GPU gather to rank 0) and a
// plain copy kernel (the measured HBM ceiling bench.py prints next to the 8 TB/s specification).
//
// The fused entry writes fixed-capacity arrays (olf_frame_buffers: [image][capacity][row]); what travels over xGMI is only the rows in
// use.  Record layout (byte offsets are multiples of 16):
//   header  : 16 x u32  magic 'OLFR', n_pairs, orb capacity, line capacity, total key points, total key lines, total left key points,
//                       total left key lines, 8 x reserved (0)
//   counts  : i32 [2 n_pairs], lcounts : i32 [2 n_pairs]
//   sections: kps (28 B rows), desc (32), uright (4), depth (4), kls (68), ldesc (32), lmatches12 (4), ldisp (8), lle (24), each starting
//             at the next multiple of 16: the used rows of image 0, image 1, ... back to back (per-pair sections -- uright, depth,
//             lmatches12, ldisp, lle: pair 0, pair 1, ...; their row count is the LEFT image's count, as in the reference's per-Frame
//             vectors mvuRight / mvDepth / mvDisparity_l / mvle_l)
// The same layout is produced on the host by orb_line_slam_amd/records.py (pack_records), which the CPU tests and the verifier use.
#include "olf_internal.hpp"
#include "../../include/orbline.h"

namespace olf {

constexpr int kNSec = 9;
// row bytes and whether a section is per image (rows = that image's count) or per pair (rows = the left image's count); line: uses lcounts
__constant__ int c_rowBytes[kNSec] = {28, 32, 4, 4, 68, 32, 4, 8, 24};
static const int h_rowBytes[kNSec] = {28, 32, 4, 4, 68, 32, 4, 8, 24};
__constant__ int c_perPair[kNSec] = {0, 0, 1, 1, 0, 0, 1, 1, 1};
__constant__ int c_isLine[kNSec] = {0, 0, 0, 0, 1, 1, 1, 1, 1};

struct PackArgs {
    const uint8_t* src[kNSec];
    const int32_t* counts; const int32_t* lcounts;
    int n_pairs, cap, lcap;
};

// one block: prefix sums of the counts -> row offset of every image in every section class, header, counts
__global__ __launch_bounds__(1024) void k_pack_plan(PackArgs a, uint32_t* __restrict__ dst, unsigned long long dst_capacity, int* __restrict__ rowOfs /* [4][2n] */,
                                                    unsigned long long* __restrict__ bytes_out, int* __restrict__ status)
{
    __shared__ int s_part[1024];
    __shared__ int s_tot[4];
    const int n = 2 * a.n_pairs, tid = threadIdx.x;
    // classes: 0 = per-image ORB rows, 1 = per-pair ORB rows (left count at even images, 0 at odd), 2 = per-image line rows, 3 = per-pair line rows
    for (int cls = 0; cls < 4; ++cls) {
        const int32_t* cnt = cls < 2 ? a.counts : a.lcounts;
        const int per = (n + 1023) / 1024;
        int sum = 0;
        for (int k = 0; k < per; ++k) { const int i = tid * per + k; if (i < n) sum += ((cls & 1) && (i & 1)) ? 0 : cnt[i]; }
        s_part[tid] = sum;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) { int v = tid >= o ? s_part[tid - o] : 0; __syncthreads(); s_part[tid] += v; __syncthreads(); }
        int run = tid ? s_part[tid - 1] : 0;
        for (int k = 0; k < per; ++k) { const int i = tid * per + k; if (i < n) { rowOfs[cls * n + i] = run; run += ((cls & 1) && (i & 1)) ? 0 : cnt[i]; } }
        if (tid == 1023) s_tot[cls] = s_part[1023];
        __syncthreads();
    }
    if (tid == 0) {
        unsigned long long off = 64 + (unsigned long long)(2 * n) * 4;
        off = (off + 15) & ~15ull;
        dst[0] = 0x52464c4fu; dst[1] = (uint32_t)a.n_pairs; dst[2] = (uint32_t)a.cap; dst[3] = (uint32_t)a.lcap;
        dst[4] = (uint32_t)s_tot[0]; dst[5] = (uint32_t)s_tot[2]; dst[6] = (uint32_t)s_tot[1]; dst[7] = (uint32_t)s_tot[3];
        for (int s = 8; s < 16; ++s) dst[s] = 0u;
        for (int s = 0; s < kNSec; ++s) {
            const int cls = c_isLine[s] * 2 + c_perPair[s];
            off += (unsigned long long)s_tot[cls] * c_rowBytes[s];
            off = (off + 15) & ~15ull;
        }
        *bytes_out = off;
        if (off > dst_capacity) atomicOr(status, 32);
    }
    for (int i = tid; i < n; i += 1024) { dst[16 + i] = (uint32_t)a.counts[i]; dst[16 + n + i] = (uint32_t)a.lcounts[i]; }
}

// grid (image, section): the used rows of that image in that section, copied as dwords
__global__ __launch_bounds__(256) void k_pack_rows(PackArgs a, uint8_t* __restrict__ dst, unsigned long long dst_capacity, const int* __restrict__ rowOfs,
                                                   const unsigned long long* __restrict__ bytes_out)
{
    const int img = blockIdx.x, s = blockIdx.y, n = 2 * a.n_pairs;
    if (*bytes_out > dst_capacity) return;
    const int cls = c_isLine[s] * 2 + c_perPair[s];
    if (c_perPair[s] && (img & 1)) return;
    const int rows = (c_isLine[s] ? a.lcounts : a.counts)[img];
    const int rb = c_rowBytes[s], capRows = c_isLine[s] ? a.lcap : a.cap;
    // section base: recompute the running offset exactly as k_pack_plan did
    unsigned long long off = 64 + (unsigned long long)(2 * n) * 4;
    off = (off + 15) & ~15ull;
    const uint32_t* hdr = reinterpret_cast<const uint32_t*>(dst);
    const unsigned tot[4] = {hdr[4], hdr[6], hdr[5], hdr[7]};
    for (int q = 0; q < s; ++q) { off += (unsigned long long)tot[c_isLine[q] * 2 + c_perPair[q]] * c_rowBytes[q]; off = (off + 15) & ~15ull; }
    const size_t srcImg = c_perPair[s] ? (size_t)(img >> 1) : (size_t)img;
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(a.src[s] + srcImg * capRows * rb);
    uint32_t* dp = reinterpret_cast<uint32_t*>(dst + off + (unsigned long long)rowOfs[cls * n + img] * rb);
    const int words = rows * rb / 4;
    for (int i = threadIdx.x; i < words; i += 256) dp[i] = sp[i];
}

int launch_pack_records(const olf_frame_buffers& fb, int n_pairs, int cap, int lcap, uint8_t* d_dst, size_t dst_capacity, int* d_rowOfs,
                        unsigned long long* d_bytes, int* d_status, hipStream_t s)
{
    PackArgs a;
    const void* src[kNSec] = {fb.kps, fb.desc, fb.uright, fb.depth, fb.kls, fb.ldesc, fb.lmatches12, fb.ldisp, fb.lle};
    for (int i = 0; i < kNSec; ++i) a.src[i] = static_cast<const uint8_t*>(src[i]);
    a.counts = fb.counts; a.lcounts = fb.lcounts; a.n_pairs = n_pairs; a.cap = cap; a.lcap = lcap;
    hipLaunchKernelGGL(k_pack_plan, dim3(1), dim3(1024), 0, s, a, reinterpret_cast<uint32_t*>(d_dst), (unsigned long long)dst_capacity, d_rowOfs, d_bytes, d_status);
    hipLaunchKernelGGL(k_pack_rows, dim3(2 * n_pairs, kNSec), dim3(256), 0, s, a, d_dst, (unsigned long long)dst_capacity, d_rowOfs, d_bytes);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

size_t pack_records_bound(int n_pairs, int cap, int lcap)
{
    size_t b = 64 + (size_t)4 * n_pairs * 4 + 16;
    for (int s = 0; s < kNSec; ++s) {
        const bool perPair = s == 2 || s == 3 || s >= 6, line = s >= 4;
        b += (size_t)(perPair ? n_pairs : 2 * n_pairs) * (line ? lcap : cap) * h_rowBytes[s] + 16;
    }
    return b;
}

// ---- copy kernel: one 16-byte element per thread, no loop -- the shape that reaches the part's practical copy rate (tools/micro/copy_sweep.hip
// on an MI355X: 6.2 TB/s read + write, the 6.29 TB/s of MI355X_MICROARCH.md; grid-stride variants with 256 .. 32 768 blocks 4.3 - 5.5 TB/s,
// hipMemcpyDtoD 4.8 - 5.2 TB/s)
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

int launch_copy16(const void* src, void* dst, size_t bytes, hipStream_t s)
{
    const size_t n16 = bytes / 16;
    if (n16 == 0) return OLF_OK;
    if ((n16 + 255) / 256 > 0x7fffffffull) { set_error("launch_copy16: buffer too large for one launch"); return OLF_ERR_INVALID; }
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, static_cast<const uint4*>(src), static_cast<uint4*>(dst), n16);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

// ---- map-point mask of freshly matched stereo points: 4 depths in, 4 mask bytes out per thread (the last thread takes the 0 .. 3 left over)
__global__ __launch_bounds__(256) void k_depth_mask(const float* __restrict__ depth, uint8_t* __restrict__ mask, size_t n)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 4 <= n) {
        const float4 z = *reinterpret_cast<const float4*>(depth + i);
        *reinterpret_cast<uint32_t*>(mask + i) = (z.x > 0.f ? 1u : 0u) | (z.y > 0.f ? 0x100u : 0u) | (z.z > 0.f ? 0x10000u : 0u) | (z.w > 0.f ? 0x1000000u : 0u);
    } else {
        for (size_t k = i; k < n; ++k) mask[k] = depth[k] > 0.f;
    }
}

int launch_depth_mask(const float* depth, uint8_t* mask, size_t n, hipStream_t s)
{
    if (n == 0) return OLF_OK;
    if ((reinterpret_cast<uintptr_t>(depth) & 15) || (reinterpret_cast<uintptr_t>(mask) & 3)) {
        set_error("launch_depth_mask: depth must be 16-byte and mask 4-byte aligned"); return OLF_ERR_INVALID; }
    const size_t nt = (n + 3) / 4;
    if ((nt + 255) / 256 > 0x7fffffffull) { set_error("launch_depth_mask: buffer too large for one launch"); return OLF_ERR_INVALID; }
    hipLaunchKernelGGL(k_depth_mask, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s, depth, mask, n);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
