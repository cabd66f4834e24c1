// lsd_grow.hip -- region growing of cv::LineSegmentDetector (OpenCV 3.4 lsd.cpp region_grow, restated in oracle/line_oracle.cpp:121-148)
// with SEVERAL waves per image, bit-identical to the sequential seed loop.
//
// The reference grows one region after the other: seeds in pseudo-order (rank = position in the sorted key list), every region sees the
// pixels used by all regions before it.  Here NW waves of one workgroup grow regions of the same image concurrently as ORDERED
// SPECULATION (tools/lsd_sim.cpp is the model this was designed on; it shows 12-15x fewer serial steps with 16 waves):
//   * every pixel has an owner word: FREE, or the tag (rank << 10 | ROB slot) of the region that claimed it.  Claims are atomicMin, so
//     the older (lower-rank) region always wins a pixel.
//   * seeds enter a reorder buffer (ROB, a ring in LDS) in rank order and leave it in rank order (commit).  The watermark `wm` is the rank
//     of the oldest unresolved seed: an owner whose rank is below it is final, i.e. "used" in the reference's sense.
//   * a region that would accept a pixel claimed by an older, not yet final region cannot know whether that pixel will stay used: it
//     yields -- releases its claims and is re-run once the older region is final.  A pixel claimed by an older region that is NOT aligned
//     with the running region angle is rejected either way, so it is no conflict.
//   * a region that loses a pixel to an older one (the atomicMin of the thief returns its tag) is told so through its ROB entry and is
//     re-run as well.  A region is valid at commit time iff nobody stole from it: everything it saw as used was final when it saw it.
//   The oldest unresolved seed never waits for anything, so there is always progress; the committed regions are exactly the sequential ones.
// Inside a region the growth step is the wave agent of round 1 (up to 7 FIFO entries x 8 neighbours per iteration, speculative accept
// rounds that verify every decision against the exact running angle).
// SEVERAL WORKGROUPS PER IMAGE (MG, round 5 -- the drop-in's one-pair shape, src/Frame.cc:164-171): G workgroups (one CU each) grow one image.  Seeds are dealt to
// the groups in rank-interleaved windows of 1024 seeds; every group has its OWN reorder buffer in LDS over its own windows and a local watermark (rank of its oldest
// unresolved seed) that it publishes in global memory.  What crosses a CU boundary is relaxed AGENT-scope traffic only -- owner words (atomicMin / compare-and-swap /
// sc1 loads, which bypass the CU's L1), the published watermarks, and steal notices for regions of another group (one word per reorder-buffer slot in global memory) --
// ordered by s_waitcnt vmcnt(0) on the issuing wave; no cache write-back or invalidate is ever needed because no plain store is read by another group before the
// kernel ends.  A group's head entry commits when it is DONE, older than every other group's watermark AS READ AT THE PREVIOUS COMMIT, and its notice word read NOW
// is clear: a region that stole from it finished -- notice performed -- before its group's watermark passed it, so a notice can never arrive behind the commit.
// Logged regions go to a per-group staging list with their rank and are merged by rank behind the kernel (k_mg_merge).  tools/lsd_sim.cpp Sim2 is the model.
// Region pixel lists live in 32-pixel chunks (chunk id < E: the dedicated first chunk of a ROB slot; else from a per-image pool) chained
// through links[]; k_lsd_rect walks the chains.
#include "lsd_device.hpp"

namespace olf {

constexpr int MW_SLOT_BITS = 10;
constexpr uint32_t MW_FREE = 0xffffffffu;
constexpr int MW_RING = 256;          // FIFO window of a growing region kept in LDS, per wave
constexpr int MW_DIR = 128;           // chunk directory per wave (ordinal -> chunk id) for reading the FIFO from memory
enum { ST_EMPTY = 0, ST_READY = 1, ST_PARKED = 2, ST_GROWING = 3, ST_DONE = 4, ST_DEAD = 5 };
enum { C_HEAD = 0, C_TAIL, C_DISPNEXT, C_WM, C_LOCKDISP, C_LOCKCOMMIT, C_LOCKALLOC, C_FREETOP, C_POOLTOP, C_NREG, C_ABORT, C_FREEHEAD, C_OMIN, C_WML, C_SCAN, C_SCANVER, C_N };

#define WG_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define WG_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
// Global memory shared by the waves of one workgroup (owner words, chunk lists written by another wave): workgroup-scope atomics -- coherent
// in the L2 of the XCD the workgroup runs on.  Agent scope would be wrong for the job: on a multi-XCD part it means "coherent across the
// XCDs' L2s", i.e. every load / atomic goes out to the fabric and every release fence writes the L2 back (measured: 16x slower in batch).
#define AG_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define AG_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

// MG (several workgroups per image): what another CU reads or writes goes through agent-scope relaxed atomics (loads carry sc1 and bypass the L1)
template <bool MG> __device__ __forceinline__ uint32_t own_load(const uint32_t* p)
{
    return MG ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <bool MG> __device__ __forceinline__ uint32_t own_min(uint32_t* p, uint32_t v)
{
    return MG ? __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// owner T -> FREE (only if the region still holds the pixel)
template <bool MG> __device__ __forceinline__ void own_unclaim(uint32_t* p, uint32_t T)
{
    uint32_t exp = T;
    if (MG) __hip_atomic_compare_exchange_strong(p, &exp, 0xffffffffu, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_compare_exchange_strong(p, &exp, 0xffffffffu, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t ag_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ag_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ag_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ag_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// release towards the other waves of the workgroup; MG: towards other CUs as well -- every memory operation of this wave has been performed (the compiler's
// workgroup fence does not wait for the vector-memory counter when the workgroup sits on one CU)
// (the protocol's premise is a gfx9 property: stores and atomics without return are counted in vmcnt -- gfx10 and later count them in vscnt, where this wait
// would order nothing; tools/micro/mp_litmus.hip tests the exact pattern between CUs of one and of two XCDs)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "lsd_grow.hip: the cross-CU ordering (relaxed agent-scope accesses + s_waitcnt vmcnt(0)) is written for gfx9-class ISAs (gfx950)"
#endif
template <bool MG> __device__ __forceinline__ void rel_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (MG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// wave-uniform read of an LDS word (every lane reads the same address)
__device__ __forceinline__ int lds_u(int* p) { return uni(WG_LOAD(p)); }
__device__ __forceinline__ uint32_t lds_u(uint32_t* p) { return (uint32_t)uni((int)WG_LOAD(p)); }

// head, tail, dispNext, watermark: one 16-byte LDS read, wave-uniform
struct MwCtl { int head, tail, dispNext; uint32_t wm; };
typedef int mw_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ MwCtl mw_ctl(int* ctl)
{
    // (the four words change independently; every use tolerates a stale or torn snapshot -- see the callers)
    // (two 8-byte relaxed atomic loads: a volatile vector read through the generic pointer became a flat load with system-scope cache bits and a
    // vmcnt(0) either side of it -- several hundred cycles for 16 bytes of LDS, at every pick and dispatch)
    const unsigned long long lo = __hip_atomic_load(reinterpret_cast<unsigned long long*>(ctl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned long long hi = __hip_atomic_load(reinterpret_cast<unsigned long long*>(ctl) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    MwCtl r; r.head = uni((int)(unsigned)lo); r.tail = uni((int)(unsigned)(lo >> 32)); r.dispNext = uni((int)(unsigned)hi); r.wm = (uint32_t)uni((int)(unsigned)(hi >> 32));
    return r;
}

__device__ __forceinline__ bool try_lock(int* l, int lane)
{
    int ok = 0;
    if (lane == 0) { int exp = 0; ok = __hip_atomic_compare_exchange_strong(l, &exp, 1, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1 : 0; }
    return uni(ok) != 0;
}
__device__ __forceinline__ void unlock(int* l, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(l, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// -DOLF_MW_PROF: cycle counters per phase, summed over the waves of image 0 into status[16..] (tools/prof_mw.py)
#ifdef OLF_MW_PROF
#define PROF_DECL long long pf_t = __builtin_readcyclecounter(), pf_all[2 * PF_N] = {0}; long long* pf_acc = pf_all
#define PROF(i) do { const long long _t = __builtin_readcyclecounter(); pf_acc[i] += _t - pf_t; pf_t = _t; } while (0)
#define PROF_CNT(i) (++pf_acc[i])
#else
#define PROF_DECL
#define PROF(i)
#define PROF_CNT(i)
#endif
enum { PF_COMMIT = 0, PF_PICK, PF_DISPATCH, PF_PROLOGUE, PF_GATHER, PF_CHAIN, PF_CLAIM, PF_FINISH, PF_IDLE, PF_NRUN, PF_NITER, PF_NFAIL, PF_F_INVAL, PF_F_CONTEST, PF_F_OLDER, PF_F_DUP, PF_N };

struct MwCtx {
    // LDS
    int* ctl;
    int* eRank; int* eState; uint32_t* eBlock; uint32_t* eInval; int* eN; uint32_t* eSeed; double* eAng; float* eDeg; float* eSx; float* eSy;
    uint32_t* ring; int* dir;
    // global, per image
    uint32_t* owner; const uint32_t* grad; const uint32_t* keys; uint32_t* chunks; int* links; RegionRec* recs; int* status;
    int E, mask, nkeys, nChunks, Ws, Hs, minRegSize, maxRegions, lane;
    // several workgroups per image (MG): this group, the group count, the first chunk id of this group's share of the pool (slot chunks cb .. cb + E, pool behind),
    // the image's control words / notice words [G][E] in global memory and this group's staging list of logged regions
    int G, grp, cb, ws;         // ws: log2 of the seed window (64 .. 1024 seeds)
    int* gctl; uint32_t* gInvalAll; int* sRank; RegionRec* sRec;
    const unsigned long long* tl0;      // (-DOLF_MW_PROF) start of the launch, image 0 only
};

// global control words of an image grown by several workgroups (ints): the groups' published watermarks on lines of their own, the abort flag, regions logged per group
constexpr int MG_MAX_G = 4, MG_MAX_E = 1024;
constexpr int MGC_WM = 0, MGC_ABORT = 16 * MG_MAX_G, MGC_NREG = MGC_ABORT + 16, MGC_WORDS = MGC_NREG + 16;
constexpr int MG_INF = 0x7fffffff;
constexpr size_t MG_INVAL_OFF = 1024, MG_RANK_OFF = MG_INVAL_OFF + (size_t)MG_MAX_G * MG_MAX_E * 4;
static_assert(MGC_WORDS * 4 <= (int)MG_INVAL_OFF, "control words");
__host__ __device__ inline size_t mg_rec_off(int maxRegions) { return (MG_RANK_OFF + (size_t)MG_MAX_G * maxRegions * 4 + 15) & ~(size_t)15; }
size_t lsd_grow_mg_stride(int maxRegions) { return (mg_rec_off(maxRegions) + (size_t)MG_MAX_G * maxRegions * sizeof(RegionRec) + 255) & ~(size_t)255; }
// group of the seed of rank r
__device__ __forceinline__ int mg_group(uint32_t rank, int G, int ws) { return (int)((rank >> ws) % (uint32_t)G); }

template <bool MG> __device__ __forceinline__ void mw_abort(const MwCtx& c)
{
    if (c.lane == 0) { WG_STORE(c.ctl + C_ABORT, 1); if (MG) ag_store(c.gctl + MGC_ABORT, 1); }
}

// chunk from the pool: never-used ones by a bump counter; recycled ones (given back by regions that were given up) from a free list that is
// threaded through links[] itself (head and count in LDS, under a lock), so recycling never loses a chunk however many are given back.
// -1 when the pool is exhausted
__device__ __forceinline__ int mw_alloc(const MwCtx& c)
{
    int id = -1;
    if (lds_u(c.ctl + C_FREETOP) > 0) {
        int spins = 0;
        while (!try_lock(c.ctl + C_LOCKALLOC, c.lane)) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 22)) return -1; }
        if (c.lane == 0) {
            const int ft = WG_LOAD(c.ctl + C_FREETOP);
            if (ft > 0) {
                id = WG_LOAD(c.ctl + C_FREEHEAD);
                WG_STORE(c.ctl + C_FREEHEAD, AG_LOAD(c.links + id));
                WG_STORE(c.ctl + C_FREETOP, ft - 1);
            }
        }
        unlock(c.ctl + C_LOCKALLOC, c.lane);
        id = uni(id);
        if (id >= 0) return id;
    }
    if (c.lane == 0) id = __hip_atomic_fetch_add(c.ctl + C_POOLTOP, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    id = uni(id);
    return id < c.nChunks ? id : -1;
}
// (the caller has read links[id] -- the next chunk of the list it is walking -- before it gives `id` back)
__device__ __forceinline__ void mw_free(const MwCtx& c, int id)
{
    int spins = 0;
    while (!try_lock(c.ctl + C_LOCKALLOC, c.lane)) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 22)) return; }
    if (c.lane == 0) {
        const int ft = WG_LOAD(c.ctl + C_FREETOP);
        AG_STORE(c.links + id, ft > 0 ? WG_LOAD(c.ctl + C_FREEHEAD) : -1);
        WG_STORE(c.ctl + C_FREEHEAD, id);
        WG_STORE(c.ctl + C_FREETOP, ft + 1);
    }
    unlock(c.ctl + C_LOCKALLOC, c.lane);
}

// un-claim the n pixels of the list of `slot` (pixel 0 is the seed; 1.. from the chunk chain) and give its pool chunks back
template <bool MG>
__device__ __forceinline__ void mw_release(const MwCtx& c, int slot, int n, uint32_t T)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the list may have been written by this wave a moment ago
    if (c.lane == 0) own_unclaim<MG>(c.owner + (WG_LOAD(c.eSeed + slot) & 0x3fffffu), T);
    int cid = c.cb + slot;
    for (int q0 = 0; q0 < n && cid >= 0; q0 += 32) {
        const int q = q0 + c.lane;
        if (c.lane < 32 && q >= 1 && q < n) {
            const uint32_t xy = AG_LOAD(c.chunks + (size_t)cid * 32 + c.lane);
            const int a = (int)(xy >> 16) * c.Ws + (int)(xy & 0xffffu);
            own_unclaim<MG>(c.owner + a, T);
        }
        const int nx = (q0 + 32 < n) ? uni(AG_LOAD(c.links + cid)) : -1;
        if (cid >= c.cb + c.E) mw_free(c, cid);
        cid = nx;
    }
    rel_fence<MG>();      // the pixels are free before the entry changes state
}

// The scan hint of mw_pick (C_SCAN): every entry below it is settled -- DEAD, or DONE with no steal noted -- and a settled entry only ever changes when a steal
// is noted for it.  Whoever notes a steal (after setting the entry's notice word) bumps C_SCANVER and pulls the hint back to that entry; a pick that raised the
// hint from a scan made before the notice sees the new version and pulls the hint back to the head.
__device__ __forceinline__ void mw_unsettle(const MwCtx& c, int slot)
{
    __hip_atomic_fetch_add(c.ctl + C_SCANVER, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int h = WG_LOAD(c.ctl + C_HEAD);
    __hip_atomic_fetch_min(c.ctl + C_SCAN, h + ((slot - h) & c.mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// commit DONE / DEAD entries at the head of the ROB, in rank order, up to 64 per call; logs the regions that are large enough.
// `ctl` is the caller's snapshot (the cheap test whether the head entry can go at all is made on it, without the lock).  One workgroup per image.
__device__ __forceinline__ void mw_commit(const MwCtx& c, const MwCtl& cv)
{
    if (cv.head >= cv.tail) return;
    {
        const int st0 = WG_LOAD(c.eState + (cv.head & c.mask));
        const uint32_t iv0 = WG_LOAD(c.eInval + (cv.head & c.mask));
        if (!(uni(st0) == ST_DEAD || (uni(st0) == ST_DONE && (uint32_t)uni((int)iv0) == MW_FREE))) return;
    }
    if (!try_lock(c.ctl + C_LOCKCOMMIT, c.lane)) return;
    const MwCtl cl = mw_ctl(c.ctl);
    int h = cl.head;
    {
        const int i = h + c.lane, slot = i & c.mask;
        const bool in = i < cl.tail;
        const int st = in ? WG_LOAD(c.eState + slot) : (int)ST_EMPTY;
        const uint32_t iv = in ? WG_LOAD(c.eInval + slot) : 0u;
        const int n = in ? WG_LOAD(c.eN + slot) : 0;
        const bool can = st == ST_DEAD || (st == ST_DONE && iv == MW_FREE);       // DONE with a steal noted: has to be re-run first
        const unsigned long long cm = wave_vote(can);
        const int run = cm == ~0ull ? 64 : __builtin_ctzll(~cm);
        const unsigned long long low = run >= 64 ? ~0ull : ((1ull << run) - 1ull);
        unsigned long long big = wave_vote(st == ST_DONE && n >= c.minRegSize) & low;
        int nr = lds_u(c.ctl + C_NREG);
        while (big) {
            const int l = __builtin_ctzll(big);
            big &= big - 1ull;
            const int sl = (h + l) & c.mask;
            const int nc = nr < c.maxRegions ? mw_alloc(c) : -1;
            if (nc < 0) { mw_abort<false>(c); break; }       // pool or region log exhausted: the image is grown again by the one-wave agent
            // the first 32 pixels sit in the slot's own chunk, which the next seed in this slot will overwrite: move them to a pool chunk
            if (c.lane < 32) c.chunks[(size_t)nc * 32 + c.lane] = AG_LOAD(c.chunks + (size_t)sl * 32 + c.lane);
            if (c.lane == 0) {
                c.links[nc] = AG_LOAD(c.links + sl);
                RegionRec rr; rr.start = nc; rr.n = rlane(n, l); rr.angle = c.eAng[sl];
                c.recs[nr] = rr;
            }
            ++nr;
        }
        if (c.lane < run) WG_STORE(c.eState + slot, (int)ST_EMPTY);
        h += run;
        // watermark: rank of the oldest unresolved seed.  The dispatcher publishes tail before dispNext and the snapshot `cl` is older than
        // the tail read here, so an empty ROB with a stale dispNext can only give a watermark that is too low, which is safe.
        const int t2 = lds_u(c.ctl + C_TAIL);
        const int wm = h < t2 ? lds_u(c.eRank + (h & c.mask)) : cl.dispNext;
        if (c.lane == 0) { WG_STORE(c.ctl + C_NREG, nr); WG_STORE(c.ctl + C_HEAD, h); WG_STORE(c.ctl + C_WM, wm); }
    }
    unlock(c.ctl + C_LOCKCOMMIT, c.lane);
}

// Several workgroups per image: one call = one look at the other groups (their watermarks, the abort flag), every entry at the head of this group's buffer that
// has become final committed (any number of 64-entry steps), the steal notices of the rest of the buffer swept into LDS, the group's own watermark published.
// A finished region is final when every older seed of the other groups is (its rank below their watermarks) and nobody stole from it: the notice words are
// loaded AFTER the watermarks have arrived -- a thief's notice is performed before its group's watermark passes the thief, so a notice cannot arrive behind the
// commit.  With four or more waves the group's wave 0 does nothing else (k_lsd_grow_mw): the hand-over of the commit order at a window boundary is then two
// memory round trips behind the other group's publish, not several passes of busy waves through their loop (profiles/r5a_commit_trace.txt: 40-60 us per window).
__device__ __forceinline__ void mw_commit_mg(const MwCtx& c)
{
    if (!try_lock(c.ctl + C_LOCKCOMMIT, c.lane)) return;
    int wmo = MG_INF, ab = 0;
    if (c.lane < c.G && c.lane != c.grp) wmo = ag_load(c.gctl + MGC_WM + 16 * c.lane);
    if (c.lane == 0) ab = ag_load(c.gctl + MGC_ABORT);
    uint32_t omin = (uint32_t)MG_INF;
    for (int g2 = 0; g2 < c.G; ++g2) omin = min(omin, (uint32_t)rlane(wmo, g2));
    if (uni(ab)) { if (c.lane == 0) WG_STORE(c.ctl + C_ABORT, 1); unlock(c.ctl + C_LOCKCOMMIT, c.lane); return; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (the notice words below are read behind the watermarks)
    const MwCtl cl = mw_ctl(c.ctl);
    int h = cl.head;
    uint32_t* gi = c.gInvalAll + (size_t)c.grp * c.E;
    int nr = lds_u(c.ctl + C_NREG);
    bool stop = false;
    while (!stop && h < cl.tail) {
        const int i = h + c.lane, slot = i & c.mask;
        const bool in = i < cl.tail;
        const int st = in ? WG_LOAD(c.eState + slot) : (int)ST_EMPTY;
        const uint32_t iv = in ? WG_LOAD(c.eInval + slot) : 0u;
        const int n = in ? WG_LOAD(c.eN + slot) : 0;
        const uint32_t rk = in ? (uint32_t)WG_LOAD(c.eRank + slot) : 0u;
        const bool pre = st == ST_DEAD || (st == ST_DONE && iv == MW_FREE && rk < omin);
        const unsigned long long pm = wave_vote(pre);
        const int run0 = pm == ~0ull ? 64 : __builtin_ctzll(~pm);
        if (run0 == 0) break;
        uint32_t giv = MW_FREE;
        if (c.lane < run0 && st == ST_DONE) giv = ag_load(gi + slot);
        if (giv != MW_FREE) { __hip_atomic_fetch_min(c.eInval + slot, giv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); mw_unsettle(c, slot); }
        const unsigned long long cm = wave_vote(c.lane < run0 && giv == MW_FREE);
        const int run = cm == ~0ull ? 64 : __builtin_ctzll(~cm);
        const unsigned long long low = run >= 64 ? ~0ull : ((1ull << run) - 1ull);
        unsigned long long big = wave_vote(st == ST_DONE && n >= c.minRegSize) & low;
        while (big) {
            const int l = __builtin_ctzll(big);
            big &= big - 1ull;
            const int sl = (h + l) & c.mask;
            const int nc = nr < c.maxRegions ? mw_alloc(c) : -1;
            if (nc < 0) { mw_abort<true>(c); stop = true; break; }
            if (c.lane < 32) c.chunks[(size_t)nc * 32 + c.lane] = AG_LOAD(c.chunks + (size_t)(c.cb + sl) * 32 + c.lane);
            if (c.lane == 0) {
                c.links[nc] = AG_LOAD(c.links + c.cb + sl);
                RegionRec rr; rr.start = nc; rr.n = rlane(n, l); rr.angle = c.eAng[sl];
                c.sRec[nr] = rr; c.sRank[nr] = rlane((int)rk, l);
            }
            ++nr;
        }
        if (c.lane < run) WG_STORE(c.eState + slot, (int)ST_EMPTY);
        h += run;
        if (c.lane == 0) WG_STORE(c.ctl + C_HEAD, h);          // (room for the dispatcher at once)
        if (run < 64) break;
    }
    // the notices of everything still in the buffer (regions that are told early give their claims up early)
    for (int i2 = h + c.lane; i2 < cl.tail; i2 += 64) {
        const int s2 = i2 & c.mask;
        const uint32_t v = ag_load(gi + s2);
        if (v != MW_FREE && v < WG_LOAD(c.eInval + s2)) { __hip_atomic_fetch_min(c.eInval + s2, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); mw_unsettle(c, s2); }
    }
    // watermark of the group: rank of its oldest unresolved seed (see mw_commit for the order of the reads); what the group may treat as final lies below
    // every group's watermark
    const int t2 = lds_u(c.ctl + C_TAIL);
    int wm = h < t2 ? lds_u(c.eRank + (h & c.mask)) : cl.dispNext;
    if (wm >= c.nkeys) wm = MG_INF;              // this group has nothing left
    if (wm != lds_u(c.ctl + C_WML) && c.lane == 0) { ag_store(c.gctl + MGC_WM + 16 * c.grp, wm); WG_STORE(c.ctl + C_WML, wm); }
    if (c.lane == 0) { WG_STORE(c.ctl + C_OMIN, (int)omin); WG_STORE(c.ctl + C_NREG, nr); WG_STORE(c.ctl + C_WM, min(wm, (int)omin)); }
#ifdef OLF_MW_TRACE
    // (-DOLF_MW_PROF -DOLF_MW_TRACE, image 0 of a context of four running two: a record per poll that moved the head or found it waiting, into the owner words of image 3)
    if (c.tl0) {
        int* tr = reinterpret_cast<int*>(c.owner) + 3 * (size_t)c.Ws * c.Hs + (size_t)c.grp * 300000;
        const int i2 = h + c.lane, s2 = i2 & c.mask;
        const int st2 = i2 < t2 ? WG_LOAD(c.eState + s2) : (int)ST_EMPTY;
        const int nG = (int)__popcll(wave_vote(st2 == ST_GROWING)), nR = (int)__popcll(wave_vote(st2 == ST_READY)), nP = (int)__popcll(wave_vote(st2 == ST_PARKED)), nD = (int)__popcll(wave_vote(st2 == ST_DONE || st2 == ST_DEAD));
        if (c.lane == 0) {
            const int k = tr[0];
            if (k < 29000 && (h != cl.head || (k & 7) == 0 || true)) {
                int* r = tr + 10 + 10 * k;
                r[0] = (int)(__builtin_amdgcn_s_memrealtime() - *c.tl0); r[1] = h; r[2] = t2; r[3] = h < t2 ? c.eRank[h & c.mask] : -1; r[4] = h < t2 ? c.eState[h & c.mask] : -1;
                r[5] = h - cl.head; r[6] = nG | (nR << 8) | (nP << 16) | (nD << 24); r[7] = cl.dispNext; r[8] = (int)omin; r[9] = h < t2 ? (int)c.eBlock[h & c.mask] : 0;
                tr[0] = k + 1;
            }
        }
    }
#endif
    unlock(c.ctl + C_LOCKCOMMIT, c.lane);
}

// tell the region whose tag is `victim` that the region of `rank` took one of its pixels (MG: a region of another group is told through its notice word in
// global memory, which that group's commit sweeps into its LDS entry)
template <bool MG>
__device__ __forceinline__ void mw_notify(const MwCtx& c, uint32_t victim, uint32_t rank)
{
    const uint32_t vs = victim & ((1u << MW_SLOT_BITS) - 1u);
    if (MG) {
        const int vg = mg_group(victim >> MW_SLOT_BITS, c.G, c.ws);
        if (vg != c.grp) { __hip_atomic_fetch_min(c.gInvalAll + (size_t)vg * c.E + vs, rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
    }
    __hip_atomic_fetch_min(c.eInval + vs, rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    mw_unsettle(c, (int)vs);
}

// the next seeds -> ROB entries for those that are not already consumed by a final region.  1: progress (or somebody else is at it), 0: nothing to do.
// A call looks at up to MW_DK steps of 64 seeds at once -- every load of every step in flight together -- and inserts as many WHOLE steps as the buffer has room
// for: behind the first ten thousand seeds nearly every seed of a step has been consumed already, and a dispatcher that takes one step per dependent round of
// loads (keys -> owner word + gradient word -> table entry, 2.2 us) is what the rest of the image then waits for (profiles/r5a_growth_timeline.txt).
// The loads are done before the lock is taken, the lock only covers the insertion; isolated seeds (k_lsd_iso) are claimed right here and enter the ROB as
// finished one-pixel regions.  MG: the group's seeds are the windows w of 2^ws ranks with w % G == grp; a call never straddles a window.
#ifndef MW_DK
#define MW_DK 1
#endif
template <bool MG>
__device__ __forceinline__ int mw_dispatch(const MwCtx& c, const MwCtl& cv, const float* __restrict__ angDeg, const AngEnt* __restrict__ ent)
{
    const int dn0 = cv.dispNext;
    if (dn0 >= c.nkeys) return 0;
    if (cv.tail - cv.head > c.E - 64) return 0;
    int K = min(MW_DK, (c.nkeys - dn0 + 63) >> 6);
    if (MG) K = min(K, ((1 << c.ws) - (dn0 & ((1 << c.ws) - 1))) >> 6);
    int addr[MW_DK]; uint32_t o[MW_DK], w[MW_DK]; float deg[MW_DK]; float2 ss[MW_DK];
#pragma unroll
    for (int k = 0; k < MW_DK; ++k) {
        const int rank = dn0 + 64 * k + c.lane;
        addr[k] = (k < K && rank < c.nkeys) ? (int)(c.keys[rank] & 0x3fffffu) : -1;
    }
#pragma unroll
    for (int k = 0; k < MW_DK; ++k) {
        o[k] = addr[k] >= 0 ? own_load<MG>(c.owner + addr[k]) : 0u;
        w[k] = addr[k] >= 0 ? c.grad[addr[k]] : (uint32_t)kNotDef;
    }
#pragma unroll
    for (int k = 0; k < MW_DK; ++k) {
        deg[k] = 0.f; ss[k] = make_float2(0.f, 0.f);
        // region_grow starts at the seed's angle and at (cos, sin) of it (not needed for a seed that the snapshot's watermark already shows consumed)
        const bool gone = o[k] != MW_FREE && (o[k] >> MW_SLOT_BITS) < cv.wm;
        if (addr[k] >= 0 && !(w[k] & (kIso | kNotDef)) && !gone) { deg[k] = angDeg[w[k] & 0x3fffffu]; ss[k] = ent[w[k] & 0x3fffffu].seed; }
    }
    if (!try_lock(c.ctl + C_LOCKDISP, c.lane)) return 1;
    const MwCtl cl = mw_ctl(c.ctl);
    const int dn = cl.dispNext, t = cl.tail, h = cl.head;
    int done = 0, total = 0;                    // steps inserted, entries inserted
    bool claim[MW_DK]; int sl[MW_DK];
    if (dn == dn0 && t - h <= c.E - 64) {
        const uint32_t wm = cl.wm;
        const int room = c.E - (t - h);
#pragma unroll
        for (int k = 0; k < MW_DK; ++k) {
            claim[k] = false; sl[k] = 0;
            if (k < K && done == k) {
                const int rank = dn0 + 64 * k + c.lane;
                const bool iso = (w[k] & kIso) != 0;
                // (kNotDef: the std::sort seed list also holds the undefined pixels of the smallest defined bin; they are never seeds)
                // (MG: a region of another group's LATER window may hold the seed already -- a younger owner counts as free here; the claim takes the pixel from it and tells it)
                const bool older = o[k] != MW_FREE && (o[k] >> MW_SLOT_BITS) < (uint32_t)rank;
                const bool gone = older && (o[k] >> MW_SLOT_BITS) < wm;
                // (a seed whose table entry was skipped because the older snapshot showed it consumed is consumed under the newer watermark too)
                const bool live = addr[k] >= 0 && !(w[k] & kNotDef) && !gone;
                const unsigned long long m = wave_vote(live);
                const int cnt = (int)__popcll(m);
                if (total + cnt <= room) {
                    if (live) {
                        const int s = (t + total + __popcll(m & ((1ull << c.lane) - 1ull))) & c.mask;
                        sl[k] = s;
                        claim[k] = iso && !older;
                        c.eRank[s] = rank;
                        c.eSeed[s] = (uint32_t)addr[k] | (iso ? 0x80000000u : 0u);
                        c.eInval[s] = MW_FREE;
                        if (MG) ag_store(c.gInvalAll + (size_t)c.grp * c.E + s, MW_FREE);      // (performed before the tag of this entry can be in any owner word: rel_fence below)
                        c.eN[s] = 0;
                        c.eDeg[s] = deg[k]; c.eSx[s] = ss[k].x; c.eSy[s] = ss[k].y;
                        c.eBlock[s] = older ? (o[k] >> MW_SLOT_BITS) : 0u;
                        c.eState[s] = older ? (int)ST_PARKED : claim[k] ? (int)ST_GROWING : (int)ST_READY;
                    }
                    total += cnt;
                    done = k + 1;
                }
            }
        }
        rel_fence<MG>();
        if (c.lane == 0) { WG_STORE(c.ctl + C_TAIL, t + total); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        int nx = dn + 64 * done;
        if (MG && (nx & ((1 << c.ws) - 1)) == 0) nx += (c.G - 1) << c.ws;
        if (c.lane == 0) { WG_STORE(c.ctl + C_DISPNEXT, min(nx, c.nkeys)); }
#ifdef OLF_MW_PROF
        // when (10 ns ticks since the first block of the launch started) the dispatcher of image 0 crosses every 4096th rank: status[160 + rank / 4096]
        if (c.lane == 0 && c.tl0) {
            atomicAdd(c.status + 158, total);
            const int kb = ((dn + 64 * done) >> 12) - 1;
            if (((dn >> 12) != ((dn + 64 * done) >> 12)) && kb < 30) {
                c.status[160 + kb] = (int)(__builtin_amdgcn_s_memrealtime() - *c.tl0);
                c.status[190 + kb] = atomicAdd(c.status + 158, 0); c.status[220 + kb] = atomicAdd(c.status + 159, 0);
            }
        }
#endif
    }
    unlock(c.ctl + C_LOCKDISP, c.lane);
#pragma unroll
    for (int k = 0; k < MW_DK; ++k) {
        if (k < done && claim[k]) {
            const int rank = dn0 + 64 * k + c.lane, s = sl[k];
            const uint32_t T = ((uint32_t)rank << MW_SLOT_BITS) | (uint32_t)s;
            const uint32_t old = own_min<MG>(c.owner + addr[k], T);
            if (old < T) {          // an older region took it in the meantime: consumed or to be re-examined once that region is final
                c.eBlock[s] = old >> MW_SLOT_BITS;
                __hip_atomic_store(c.eState + s, (int)ST_PARKED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                if (old != MW_FREE) mw_notify<MG>(c, old, (uint32_t)rank);     // a younger seed of a later window was quicker: the pixel is ours now
                c.eN[s] = 1;
                if (MG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the notice is performed before the entry can commit and the watermark pass it
                __hip_atomic_store(c.eState + s, (int)ST_DONE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    return 1;
}

// lowest-rank entry that can be (re-)run now; returns its slot with the entry in state GROWING (its previous state in *prev), or -1.
// The scan starts at the hint (see mw_unsettle): with several workgroups per image a buffer is mostly finished regions waiting for another group's watermark.
template <bool MG>
__device__ __forceinline__ int mw_pick(const MwCtx& c, const MwCtl& cv, int* prev)
{
    const int t = cv.tail;
    const uint32_t wm = cv.wm;
    const int ver0 = lds_u(c.ctl + C_SCANVER);
    const int hint0 = lds_u(c.ctl + C_SCAN);
    const int h = max(cv.head, hint0);
    int settledTo = h;              // every entry in [h, settledTo) was seen settled
    bool open = true;               // ... and nothing unsettled has been seen yet
    int found = -1;
    for (int base = h; base < t && found < 0; base += 64) {
        const int i = base + c.lane;
        const int slot = i & c.mask;
        int st = i < t ? WG_LOAD(c.eState + slot) : (int)ST_EMPTY;
        uint32_t bl = WG_LOAD(c.eBlock + slot);
        const uint32_t iv = WG_LOAD(c.eInval + slot);
#ifndef OLF_MW_NO_BULK
        {
            // seeds parked on a region that is final now -- all of this step at once: one look at the seed's owner word decides "consumed for good" (DEAD, the
            // common case behind the first few thousand seeds) without a region run of its own (pick, prologue, claim, finish: 6-7 k cycles each); a seed that
            // turns out free goes back with blocker 0, which the ordinary path below takes; one that another unfinished region holds waits for that one
            const bool pk = st == ST_PARKED && bl != 0u && bl < wm;
            if (wave_vote(pk)) {
                bool mine = false;
                if (pk) { int exp = ST_PARKED; mine = __hip_atomic_compare_exchange_strong(c.eState + slot, &exp, (int)ST_GROWING, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
                if (mine) {
                    const uint32_t rk = (uint32_t)WG_LOAD(c.eRank + slot);
                    const uint32_t o = own_load<MG>(c.owner + (WG_LOAD(c.eSeed + slot) & 0x3fffffu));
                    const bool older = o != MW_FREE && (o >> MW_SLOT_BITS) < rk;
                    const bool gone = older && (o >> MW_SLOT_BITS) < wm;
                    bl = older ? (o >> MW_SLOT_BITS) : 0u;
                    st = gone ? (int)ST_DEAD : (int)ST_PARKED;
                    if (!gone) WG_STORE(c.eBlock + slot, bl);
                    __hip_atomic_store(c.eState + slot, st, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
#endif
        if (open) {
            const unsigned long long sm = wave_vote(st == ST_DEAD || (st == ST_DONE && iv == MW_FREE));
            const int run = sm == ~0ull ? 64 : __builtin_ctzll(~sm);
            settledTo = min(base + run, t);
            open = run == 64;
        }
        const bool el = st == ST_READY || (st == ST_PARKED && bl < wm) || (st == ST_DONE && iv != MW_FREE && iv < wm);
        unsigned long long m = wave_vote(el);
        while (m) {
            const int l = __builtin_ctzll(m);
            const int s = (base + l) & c.mask;
            int exp = rlane(st, l);
            int ok = 0;
            if (c.lane == 0) ok = __hip_atomic_compare_exchange_strong(c.eState + s, &exp, (int)ST_GROWING, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1 : 0;
            if (uni(ok)) { *prev = rlane(st, l); found = s; break; }
            m &= m - 1ull;
        }
    }
    if (settledTo > hint0) {
        if (c.lane == 0) {
            int exp = hint0;
            if (__hip_atomic_compare_exchange_strong(c.ctl + C_SCAN, &exp, settledTo, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) &&
                WG_LOAD(c.ctl + C_SCANVER) != ver0)
                __hip_atomic_fetch_min(c.ctl + C_SCAN, WG_LOAD(c.ctl + C_HEAD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    return found;
}

// grow (or re-grow) the region of ROB entry `slot`, which this wave holds in state GROWING
#ifdef OLF_MW_PROF
#define RUN_PROF_ARGS , long long& pf_t, long long* pf_acc
#define RUN_PROF_PASS , pf_t, pf_acc
#else
#define RUN_PROF_ARGS
#define RUN_PROF_PASS
#endif
template <bool MG>
__device__ __forceinline__ void mw_run(const MwCtx& c, int slot, int prev, int wv, double prec, double precWrap, const AngEnt* __restrict__ ent RUN_PROF_ARGS)
{
    PROF_CNT(PF_NRUN);
#ifdef OLF_MW_PROF
    if (c.tl0 && c.lane == 0) atomicAdd(c.status + 159, 1);
#endif
    const int lane = c.lane, Ws = c.Ws, Hs = c.Hs;
    const int rank_v = WG_LOAD(c.eRank + slot);
    const uint32_t sw_v = WG_LOAD(c.eSeed + slot);
    const float deg_v = c.eDeg[slot], sx_v = c.eSx[slot], sy_v = c.eSy[slot];
    const uint32_t rank = (uint32_t)uni(rank_v);
    const uint32_t T = (rank << MW_SLOT_BITS) | (uint32_t)slot;
    const uint32_t sw = (uint32_t)uni((int)sw_v);
    const int seed = (int)(sw & 0x3fffffu);
    uint32_t* ring = c.ring + wv * MW_RING;
    int* dir = c.dir + wv * MW_DIR;
    // a claim whose result has not been looked at yet (this lane's atomicMin of the previous iteration; lane 0: the seed's)
    uint32_t pOld = MW_FREE;
    bool pMine = false;
    if (prev != ST_READY) {
        // re-run: the seed may have been consumed in the meantime, and a finished region that was stolen from still holds its claims
        const int nOld = lds_u(c.eN + slot);
        if (nOld > 0) { mw_release<MG>(c, slot, nOld, T); if (lane == 0) WG_STORE(c.eN + slot, 0); }
        if (lane == 0) { WG_STORE(c.eInval + slot, MW_FREE); if (MG) ag_store(c.gInvalAll + (size_t)c.grp * c.E + slot, MW_FREE); }      // steals from here on concern this run
        rel_fence<MG>();
        uint32_t old0 = 0;
        if (lane == 0) old0 = own_min<MG>(c.owner + seed, T);
        old0 = (uint32_t)uni((int)old0);
        if (old0 < T) {
            // claimed by an older region: consumed if that one is final, else wait for it
            const bool fin = (old0 >> MW_SLOT_BITS) < (uint32_t)lds_u(c.ctl + C_WM);
            if (lane == 0) {
                if (!fin) WG_STORE(c.eBlock + slot, old0 >> MW_SLOT_BITS);
                __hip_atomic_store(c.eState + slot, fin ? (int)ST_DEAD : (int)ST_PARKED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            return;
        }
        if (old0 != MW_FREE && lane == 0) mw_notify<MG>(c, old0, rank);
        if (sw & 0x80000000u) {
            // no neighbour is aligned with the seed's own angle (k_lsd_iso): the region is the seed alone, whatever is used around it
            if (MG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) { WG_STORE(c.eN + slot, 1); __hip_atomic_store(c.eState + slot, (int)ST_DONE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
            return;
        }
    } else if (lane == 0) {
        // first run of a seed that was free when it was dispatched: claim it and look at the answer with the first iteration's loads
        pOld = own_min<MG>(c.owner + seed, T);
        pMine = true;
    }
    const int own = c.cb + slot;        // the slot's own chunk
    const uint32_t seedXY = (uint32_t)(seed % Ws) | ((uint32_t)(seed / Ws) << 16);
    double reg_angle = d_mul((double)deg_v, kDegToRads);
    float sumdx = sx_v, sumdy = sy_v;
    if (lane == 0) { ring[0] = seedXY; c.chunks[(size_t)own * 32] = seedXY; c.links[own] = -1; dir[0] = own; }
    __builtin_amdgcn_wave_barrier();
    int n = 1, i = 0, cur = own;
    uint32_t blocker = MW_FREE;
    bool fail = false, seedLost = false;      // seedLost: the seed itself had been taken by an older region when this run claimed it
    // results of the claims issued one iteration ago: a pixel an older region had taken between this region's test and its claim means the
    // decisions since were made on a pixel that was not available -> yield (pOld == T: the own claim was not visible to the gather that
    // followed it and the pixel was added twice -- not observed, but detected rather than assumed away); a younger region's pixel is
    // ours now and that region is told
#define MW_PENDING() do { \
        if (pMine && pOld != MW_FREE && pOld > T) mw_notify<MG>(c, pOld, rank); \
        const unsigned long long _bad = wave_vote(pMine && pOld <= T); \
        pMine = false; \
        if (_bad) { const uint32_t _o = (uint32_t)rlane((int)pOld, __builtin_ctzll(_bad)); blocker = _o == T ? 0u : (_o >> MW_SLOT_BITS); fail = true; \
                    seedLost = n == 1 && i == 0 && _bad == 1ull; \
                    if (_o == T) PROF_CNT(PF_F_DUP); else PROF_CNT(PF_F_OLDER); } } while (0)
    PROF(PF_PROLOGUE);
    while (i < n) {
        PROF_CNT(PF_NITER);
        const uint32_t iv_v = WG_LOAD(c.eInval + slot), wm_v = (uint32_t)WG_LOAD(c.ctl + C_WM);
        const uint32_t iv = (uint32_t)uni((int)iv_v), wm = (uint32_t)uni((int)wm_v);
        if (iv != MW_FREE) { blocker = iv; fail = true; PROF_CNT(PF_F_INVAL); break; }
        // 8 FIFO entries x 8 neighbours per iteration, candidates as lane masks in scalar registers (as in the one-wave agent, lsd.hip)
        const int nb = min(8, n - i);
        const int e = lane >> 3, k = (lane & 7) + ((lane & 7) >= 4 ? 1 : 0);
        const unsigned long long geo = nb == 8 ? ~0ull : (1ull << (8 * nb)) - 1ull;
        // (as in the one-wave agent, lsd.hip: the ring read unconditional, the memory read a rare branch that waits for itself -- merged into one
        // load the two make the compiler drain the vector-memory counter, i.e. wait for the previous iteration's claims, before every gather)
        uint32_t rp = ring[(i + e) & (MW_RING - 1)];
        asm volatile("" : "+v"(rp));
        if (n - i > MW_RING) {
            // the window left the ring: read the FIFO from the chunk chain
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const int q = i + (wave_bit(geo) ? e : 0);
            const int ord = q >> 5;
            int cid;
            if (ord < MW_DIR) cid = dir[ord];
            else { cid = dir[MW_DIR - 1]; for (int o2 = MW_DIR - 1; o2 < ord; ++o2) cid = AG_LOAD(c.links + cid); }
            rp = AG_LOAD(c.chunks + (size_t)cid * 32 + (q & 31));
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        const int xx = (int)(rp & 0xffffu) + (k % 3) - 1, yy = (int)(rp >> 16) + (k / 3) - 1;
        const unsigned long long inImg = geo & wave_vote((unsigned)xx < (unsigned)Ws) & wave_vote((unsigned)yy < (unsigned)Hs);
        const int a = wave_bit(inImg) ? yy * Ws + xx : 0;
        const uint32_t pw = c.grad[a];
        const uint32_t o = own_load<MG>(c.owner + a);
        const int xy = xx | (yy << 16);
        // not mine, not used by a final region; an older, unfinished claim stays a candidate ("contested")
        const unsigned long long olderM = wave_vote(o < T);
        unsigned long long cm = inImg & wave_vote(!(pw & kNotDef)) & wave_vote(o != T) & ~(olderM & wave_vote((o >> MW_SLOT_BITS) < wm));
        const unsigned long long conM = cm & olderM;
        MW_PENDING();
        if (fail) break;
        double ang = 0, cs = 0, sn = 0;
        if (wave_bit(cm)) {
            const uint32_t ti = pw & 0x3fffffu;
            const AngEnt* t = ent + ti;
            ang = t->ang;
            cs = t->cs; sn = t->sn;
        }
        PROF(PF_GATHER);
        unsigned long long acc = 0;
        const int n0 = n;
        while (cm) {
            // isaligned(): see k_lsd_grow (lsd.hip) -- wrapped test against precWrap, candidates in lane order = the reference's visiting order
            const double nth = fabs(d_sub(reg_angle, ang));
            const unsigned long long wasM = wave_vote(nth <= prec) | wave_vote(nth >= precWrap);
            const unsigned long long al = wasM & cm;
            if (!al) break;
            if ((al & (al - 1ull)) == 0) {
                const int cc = __builtin_ctzll(al);
                cm &= ~((2ull << cc) - 1ull);
                const int a_c = rlane(a, cc);
                const double cs_c = rlane_d(cs, cc), sn_c = rlane_d(sn, cc);
                acc |= 1ull << cc;
                ++n;
                sumdx = (float)d_add((double)sumdx, cs_c);
                sumdy = (float)d_add((double)sumdy, sn_c);
                reg_angle = d_mul((double)agent_fastAtan2(sumdy, sumdx), kDegToRads);
                cm &= ~wave_vote(a == a_c);
                continue;
            }
            // speculative round (see lsd.hip): all aligned candidates assumed accepted in lane order, one fastAtan2 for all their angles, every
            // decision re-tested against the angle that governs it, commit up to the first decision that changes
            unsigned long long todo = al, spec = 0;
            float sx = sumdx, sy = sumdy, psx = 0.f, psy = 0.f;
            int dupStep = 64, j = 0;
            while (todo) {
                const int cc = __builtin_ctzll(todo);
                const int a_c = rlane(a, cc);
                const double cs_c = rlane_d(cs, cc), sn_c = rlane_d(sn, cc);
                sx = (float)d_add((double)sx, cs_c);
                sy = (float)d_add((double)sy, sn_c);
                if (lane == j) { psx = sx; psy = sy; }
                const bool tw = a == a_c;
                if (tw && lane != cc) dupStep = j;
                todo &= ~wave_vote(tw);
                spec |= 1ull << cc;
                ++j;
            }
            const double th = d_mul((double)agent_fastAtan2(psy, psx), kDegToRads);
            const int gq = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(spec >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)spec, 0u));
            double thg = shfl_d(th, max(gq - 1, 0));
            if (gq == 0) thg = reg_angle;
            const double n2 = fabs(d_sub(thg, ang));
            const unsigned long long reM = wave_vote(n2 <= prec) | wave_vote(n2 >= precWrap);
            const unsigned long long mis = (reM ^ wasM) & ~wave_vote(dupStep < gq) & cm;
            const unsigned long long bm = mis ? ((1ull << __builtin_ctzll(mis)) - 1ull) : ~0ull;
            const unsigned long long okAcc = spec & bm;
            const int tt = __popcll(okAcc);
            acc |= okAcc;
            n += tt;
            cm &= ~bm;
            cm &= ~wave_vote(dupStep < tt);
            sumdx = __int_as_float(rlane(__float_as_int(psx), tt - 1));
            sumdy = __int_as_float(rlane(__float_as_int(psy), tt - 1));
            reg_angle = rlane_d(th, tt - 1);
        }
        PROF(PF_CHAIN);
        // vmcnt(0) while only loads can be in flight (they have returned): what crosses the back edge is then exactly this iteration's claims and
        // list stores, and the next gather is issued in front of their results
        __builtin_amdgcn_s_waitcnt(0x0F70);
        if (acc & conM) {
            // the reference would add a pixel that an older, unfinished region holds right now: yield to that region
            blocker = (uint32_t)rlane((int)(o >> MW_SLOT_BITS), __builtin_ctzll(acc & conM));
            n = n0; fail = true; PROF_CNT(PF_F_CONTEST); break;
        }
        if (acc) {
            // chunks for list positions n0 .. n-1 (at most two new ones: an iteration adds <= 64 pixels)
            const int curOrd = (n0 - 1) >> 5, lastOrd = (n - 1) >> 5;
            int new1 = -1, new2 = -1;
            if (lastOrd > curOrd) {
                new1 = mw_alloc(c);
                if (lastOrd > curOrd + 1 && new1 >= 0) new2 = mw_alloc(c);
                if (new1 < 0 || (lastOrd > curOrd + 1 && new2 < 0)) {
                    mw_abort<MG>(c);
                    n = n0; blocker = 0; fail = true; break;
                }
                if (lane == 0) {
                    c.links[cur] = new1;
                    c.links[new1] = new2;
                    if (new2 >= 0) c.links[new2] = -1;
                    if (curOrd + 1 < MW_DIR) dir[curOrd + 1] = new1;
                    if (new2 >= 0 && curOrd + 2 < MW_DIR) dir[curOrd + 2] = new2;
                }
            }
            const bool mine = wave_bit(acc);
            if (mine) {
                const int idx = n0 + wave_rank_below(acc);
                pOld = own_min<MG>(c.owner + a, T);
                pMine = true;
                ring[idx & (MW_RING - 1)] = (uint32_t)xy;
                const int ord = idx >> 5;
                const int cid = ord == curOrd ? cur : (ord == curOrd + 1 ? new1 : new2);
                c.chunks[(size_t)cid * 32 + (idx & 31)] = (uint32_t)xy;
            }
            if (lastOrd > curOrd) cur = new2 >= 0 ? new2 : new1;
        }
        i += nb;
        __builtin_amdgcn_wave_barrier();
        PROF(PF_CLAIM);
    }
    MW_PENDING();      // the last iteration's claims (steals are reported even when the region is given up)
#undef MW_PENDING
    if (fail) {
        PROF_CNT(PF_NFAIL);
        // the seed consumed by a region that is final already: resolved for good; anything else waits for the region it ran into
        const bool dead = seedLost && blocker < (uint32_t)lds_u(c.ctl + C_WM);
        if (!seedLost) mw_release<MG>(c, slot, n, T);
        if (lane == 0) {
            WG_STORE(c.eN + slot, 0);
            WG_STORE(c.eBlock + slot, blocker);
            __hip_atomic_store(c.eState + slot, dead ? (int)ST_DEAD : (int)ST_PARKED, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        PROF(PF_FINISH);
        return;
    }
    rel_fence<MG>();          // list + links are in memory (MG: claims and steal notices performed) before the entry says DONE
    if (lane == 0) {
        WG_STORE(c.eN + slot, n);
        c.eAng[slot] = reg_angle;
        __hip_atomic_store(c.eState + slot, (int)ST_DONE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    PROF(PF_FINISH);
}

size_t lsd_grow_mw_lds_bytes(int nw, int E)
{
    return (size_t)(C_N + 1) * 4 + (size_t)E * (9 * 4 + 8) + 8 + (size_t)nw * (MW_RING + MW_DIR) * 4;
}

// MG = false: one workgroup per image (blockIdx.x = image).  MG = true: G workgroups per image; block b serves image (b / 8 / G) * 8 + b % 8 as group (b / 8) % G, so that
// the groups of an image land on ONE XCD under the observed round-robin placement (block b -> XCD b % 8) and meet in that XCD's L2 -- a speed matter only: every
// cross-group access is an agent-scope atomic, correct under any placement.  `scatter` (olf_debug_lsd_scatter, tests) deals the groups of an image to CONSECUTIVE
// blocks instead -- image b / G, group b % G: different XCDs under that placement -- so that tests/test_lsd_grow_gpu.py can hold the claim to a bit-exact result.
template <bool MG>
__global__ __launch_bounds__(1024) void k_lsd_grow_mw(const LineGeom* __restrict__ gp, const uint32_t* __restrict__ gradAll, uint32_t* __restrict__ ownerAll,
                                                      const uint32_t* __restrict__ keysAll, const int* __restrict__ keyCount,
                                                      uint32_t* __restrict__ chunksAll, int* __restrict__ linksAll, RegionRec* __restrict__ recsAll,
                                                      int* __restrict__ regCount, int* __restrict__ status, const float* __restrict__ angDeg,
                                                      const AngEnt* __restrict__ ent, int E, int nChunks, int poolLimit, int* __restrict__ growFmt,
                                                      unsigned char* __restrict__ mgAll, size_t mgStride, int G, int n_images, int wsBits, int ahead, int scatter)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const LineGeom& g = *gp;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int img = blockIdx.x, grp = 0;
    if (MG) {
        const int k = blockIdx.x >> 3;
        img = (k / G) * 8 + (blockIdx.x & 7); grp = k % G;
        if (scatter) { img = blockIdx.x / G; grp = blockIdx.x % G; }
        if (img >= n_images) return;
    }
    MwCtx c;
    {
        unsigned char* p = smem;
        c.eAng = reinterpret_cast<double*>(p); p += (size_t)E * 8;
        c.ctl = reinterpret_cast<int*>(p); p += (C_N + 1) * 4;
        c.eRank = reinterpret_cast<int*>(p); p += (size_t)E * 4;
        c.eState = reinterpret_cast<int*>(p); p += (size_t)E * 4;
        c.eBlock = reinterpret_cast<uint32_t*>(p); p += (size_t)E * 4;
        c.eInval = reinterpret_cast<uint32_t*>(p); p += (size_t)E * 4;
        c.eN = reinterpret_cast<int*>(p); p += (size_t)E * 4;
        c.eSeed = reinterpret_cast<uint32_t*>(p); p += (size_t)E * 4;
        c.eDeg = reinterpret_cast<float*>(p); p += (size_t)E * 4;
        c.eSx = reinterpret_cast<float*>(p); p += (size_t)E * 4;
        c.eSy = reinterpret_cast<float*>(p); p += (size_t)E * 4;
        c.ring = reinterpret_cast<uint32_t*>(p); p += (size_t)nw * MW_RING * 4;
        c.dir = reinterpret_cast<int*>(p);
    }
    c.owner = ownerAll + (size_t)img * g.Ps;
    c.grad = gradAll + (size_t)img * g.Ps;
    c.keys = keysAll + (size_t)img * g.Ps;
    c.chunks = chunksAll + (size_t)img * nChunks * 32;
    c.links = linksAll + (size_t)img * nChunks;
    c.recs = recsAll + (size_t)img * g.maxRegions;
    c.status = status;
    c.E = E; c.mask = E - 1; c.nkeys = keyCount[img * 32];
    c.Ws = g.Ws; c.Hs = g.Hs;
    c.minRegSize = g.minRegSize; c.maxRegions = g.maxRegions; c.lane = lane;
    c.G = 1; c.grp = 0; c.cb = 0; c.ws = 10; c.gctl = nullptr; c.gInvalAll = nullptr; c.sRank = nullptr; c.sRec = nullptr;
    c.nChunks = poolLimit;      // (the bound of mw_alloc only: the per-image stride of chunks / links is nChunks)
    if (MG) {
        unsigned char* mg = mgAll + (size_t)img * mgStride;
        c.G = G; c.grp = grp; c.ws = wsBits;
        const int share = poolLimit / G;            // every group allocates from its own part of the image's chunk pool (ids stay image-wide: k_lsd_rect walks them)
        c.cb = grp * share; c.nChunks = c.cb + share;
        c.gctl = reinterpret_cast<int*>(mg);
        c.gInvalAll = reinterpret_cast<uint32_t*>(mg + MG_INVAL_OFF);
        c.sRank = reinterpret_cast<int*>(mg + MG_RANK_OFF) + (size_t)grp * g.maxRegions;
        c.sRec = reinterpret_cast<RegionRec*>(mg + mg_rec_off(g.maxRegions)) + (size_t)grp * g.maxRegions;
    }
    c.tl0 = nullptr;
#ifdef OLF_MW_PROF
    if (img == 0) {
        unsigned long long* t0 = reinterpret_cast<unsigned long long*>(status + 156);
        if (threadIdx.x == 0) atomicCAS(t0, 0ull, (unsigned long long)__builtin_amdgcn_s_memrealtime());
        c.tl0 = t0;
    }
#endif
    for (int q = threadIdx.x; q < E; q += blockDim.x) c.eState[q] = ST_EMPTY;
    if (threadIdx.x < C_N + 1) c.ctl[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        c.ctl[C_POOLTOP] = c.cb + E;              // chunk ids cb .. cb + E are the ROB slots' own chunks
        if (MG) { c.ctl[C_DISPNEXT] = min(grp << wsBits, c.nkeys); c.ctl[C_WML] = -1; }
    }
    __syncthreads();
    const double prec = g.prec, precWrap = g.precWrap;
    const bool dedicated = MG && nw >= 4 && !(ahead & 0x10000);        // wave 0 of the group commits and publishes, the others grow (bit 16 of `ahead`: OLF_MW_NO_COMMIT_WAVE, A/B)
    ahead &= 0xffff;
    int idle = 0, cwHead = -1, cwTail = -1;
    PROF_DECL;
    for (;;) {
        MwCtl cv = mw_ctl(c.ctl);
#ifdef OLF_MW_PROF
#ifndef OLF_MW_PROF_LATE
#define OLF_MW_PROF_LATE 40960
#endif
        pf_acc = pf_all + (cv.dispNext >= OLF_MW_PROF_LATE ? PF_N : 0);
#endif
        if (lds_u(c.ctl + C_ABORT)) break;
        if (MG && dedicated && wv == 0) {
            // the group's commit wave: nothing but the hand-over of the commit order
            if (cv.dispNext >= c.nkeys && cv.head == cv.tail) { const MwCtl c2 = mw_ctl(c.ctl); if (c2.dispNext >= c.nkeys && c2.head == c2.tail) break; }
            mw_commit_mg(c);
            PROF(PF_COMMIT);
            __builtin_amdgcn_s_sleep(1);
            // the guard counts polls without any movement of the buffer (a single region of a few hundred thousand pixels keeps a wave busy for a second)
            if (cv.head != cwHead || cv.tail != cwTail) { cwHead = cv.head; cwTail = cv.tail; idle = 0; }
            if (++idle > (1 << 23)) { if (lane == 0) atomicOr(status, 16); mw_abort<MG>(c); break; }
            continue;
        }
        if (!MG) mw_commit(c, cv); else if (!dedicated) mw_commit_mg(c);
        PROF(PF_COMMIT);
        // supply ahead of demand: any wave tops the reorder buffer up BEFORE it looks for a region whenever the buffer is less than half full, so that the waves do
        // not run dry together and spin through commit / pick / dispatch while one of them waits for the dispatcher's loads (profiles/r4zz_mw_dispatch_ahead_ab.txt)
        if (nw > 1 && cv.dispNext < c.nkeys && cv.tail - cv.head < ahead && lds_u(c.ctl + C_LOCKDISP) == 0) {
            if (mw_dispatch<MG>(c, cv, angDeg, ent)) cv = mw_ctl(c.ctl);
            PROF(PF_DISPATCH);
        }
        int prev = ST_READY;
        const int slot = mw_pick<MG>(c, cv, &prev);
        PROF(PF_PICK);
        if (slot >= 0) { idle = 0; mw_run<MG>(c, slot, prev, wv, prec, precWrap, ent RUN_PROF_PASS); continue; }
        const int dsp = mw_dispatch<MG>(c, cv, angDeg, ent);
        PROF(PF_DISPATCH);
        if (dsp) continue;
        // nothing to run, nothing to dispatch: finished, or waiting for other waves' regions
        if (cv.dispNext >= c.nkeys && cv.head == cv.tail) {
            // (snapshot: dispNext is published after tail, and nothing is inserted once dispNext has reached nkeys)
            const MwCtl c2 = mw_ctl(c.ctl);
            if (c2.dispNext >= c.nkeys && c2.head == c2.tail) break;
        }
        __builtin_amdgcn_s_sleep(8);
        PROF(PF_IDLE);
        if (++idle > (1 << 21)) {
            // the guard: what the first wave to give up saw goes to status[32..] (olf_debug_status)
            if (lane == 0 && img == 0 && atomicCAS(status + 28 + grp, 0, 1) == 0) {
                int* d = status + 64 + grp * 24;
                const int hs = cv.head & c.mask;
                d[0] = img; d[1] = grp; d[2] = cv.head; d[3] = cv.tail; d[4] = cv.dispNext; d[5] = (int)cv.wm;
                d[6] = c.ctl[C_OMIN]; d[7] = c.ctl[C_WML]; d[8] = c.eState[hs]; d[9] = c.eRank[hs]; d[10] = (int)c.eInval[hs];
                d[11] = (int)c.eBlock[hs]; d[12] = c.nkeys; d[13] = c.eN[hs];
                if (MG) { for (int q = 0; q < c.G; ++q) d[14 + q] = ag_load(c.gctl + MGC_WM + 16 * q); d[18] = (int)ag_load(c.gInvalAll + (size_t)grp * E + hs); d[19] = ag_load(c.gctl + MGC_ABORT); }
                d[20] = c.ctl[C_LOCKCOMMIT]; d[21] = c.ctl[C_LOCKDISP]; d[22] = (int)ag_load(c.owner + (c.eSeed[hs] & 0x3fffffu)); d[23] = (int)c.eSeed[hs];
            }
            if (lane == 0) atomicOr(status, 16);
            mw_abort<MG>(c); break;
        }
    }
#ifdef OLF_MW_PROF
    if (img == 0 && threadIdx.x == 0) status[250 + grp] = (int)(__builtin_amdgcn_s_memrealtime() - *c.tl0);
    // (two sets: the whole launch at status[16..], the part behind seed rank OLF_MW_PROF_LATE at status[100..])
    if (img == 0 && lane == 0) for (int q = 0; q < PF_N; ++q) {
        atomicAdd(reinterpret_cast<unsigned long long*>(status + 16) + q, (unsigned long long)(pf_all[q] + pf_all[PF_N + q]));
        atomicAdd(reinterpret_cast<unsigned long long*>(status + 100) + q, (unsigned long long)pf_all[PF_N + q]);
    }
#endif
    __syncthreads();
    // growFmt: 0 = this image's regions are chunk chains; -1 = given up (chunk pool or region log exhausted, or the idle guard): nothing of
    // the image's gradient words was modified (claims live in the owner words), so k_lsd_grow replays it from the start (launch_lsd_grow)
    if (threadIdx.x == 0) {
        const bool ab = c.ctl[C_ABORT] != 0;
        if (MG) {
            // this group is through: nothing of it is unresolved any more (the other groups may still be waiting for exactly that); k_mg_merge writes regCount / growFmt
            ag_store(c.gctl + MGC_WM + 16 * grp, MG_INF);
            c.gctl[MGC_NREG + grp] = c.ctl[C_NREG];
            if (ab) ag_store(c.gctl + MGC_ABORT, 1);
        } else { regCount[img] = ab ? 0 : c.ctl[C_NREG]; growFmt[img] = ab ? -1 : 0; }
    }
}

// the groups' staging lists (each in rank order) -> the image's region log in rank order = the order the sequential loop logs them in
__global__ __launch_bounds__(256) void k_mg_merge(const LineGeom* __restrict__ gp, unsigned char* __restrict__ mgAll, size_t mgStride, int G, RegionRec* __restrict__ recsAll,
                                                  int* __restrict__ regCount, int* __restrict__ growFmt)
{
    const LineGeom& g = *gp;
    const int img = blockIdx.x;
    unsigned char* mg = mgAll + (size_t)img * mgStride;
    const int* gctl = reinterpret_cast<const int*>(mg);
    const int* ranks = reinterpret_cast<const int*>(mg + MG_RANK_OFF);
    const RegionRec* srec = reinterpret_cast<const RegionRec*>(mg + mg_rec_off(g.maxRegions));
    int cnt[MG_MAX_G], total = 0;
    for (int q = 0; q < MG_MAX_G; ++q) { cnt[q] = q < G ? gctl[MGC_NREG + q] : 0; total += cnt[q]; }
    if (gctl[MGC_ABORT] != 0 || total > g.maxRegions) {
        if (threadIdx.x == 0) { regCount[img] = 0; growFmt[img] = -1; }
        return;
    }
    RegionRec* out = recsAll + (size_t)img * g.maxRegions;
    for (int q = 0; q < G; ++q) {
        const int* rq = ranks + (size_t)q * g.maxRegions;
        for (int j = threadIdx.x; j < cnt[q]; j += blockDim.x) {
            const int r = rq[j];
            int pos = j;
            for (int o = 0; o < G; ++o) {
                if (o == q) continue;
                const int* ro = ranks + (size_t)o * g.maxRegions;
                int lo = 0, hi = cnt[o];
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (ro[mid] < r) lo = mid + 1; else hi = mid; }
                pos += lo;
            }
            out[pos] = srec[(size_t)q * g.maxRegions + j];
        }
    }
    if (threadIdx.x == 0) { regCount[img] = total; growFmt[img] = 0; }
}

int launch_lsd_grow_mw(const LineGeom& g, LineDeviceBufs& b, int n_images, int nw, int E, int G, hipStream_t s)
{
    const size_t lds = lsd_grow_mw_lds_bytes(nw, E);
    // up to 46 KB (16 waves, 512 entries) fits the 64 KB a launch may ask for without a function attribute; a 1024-entry buffer (70 KB) needs the attribute,
    // which is a property of the function ON THE CURRENT DEVICE: set per device, remembered per device
    if (lds > 64 * 1024) {
        static bool done[64] = {};
        int dev = 0;
        OLF_HIP_CHECK(hipGetDevice(&dev));
        if (lds > 100 * 1024 || dev < 0 || dev >= 64) { set_error("launch_lsd_grow_mw: LDS"); return OLF_ERR_INVALID; }
        if (!done[dev]) {
            OLF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lsd_grow_mw<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
            OLF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lsd_grow_mw<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
            done[dev] = true;
        }
    }
    const int poolLimit = b.poolChunks > 0 ? std::min(b.poolChunks, b.nChunks) : b.nChunks;
    // entries in the buffer below which any wave tops it up before it looks for a region (OLF_MW_AHEAD for A/B runs)
    static const int envAhead = [] { const char* e = getenv("OLF_MW_AHEAD"); return e ? atoi(e) : 0; }();
    static const int noCw = getenv("OLF_MW_NO_COMMIT_WAVE") ? 0x10000 : 0;
    const int ahead = (envAhead > 0 ? std::min(envAhead, E - 64) : E / 2) | noCw;
    if (G > 1) {
        // OLF_LSD_WS: log2 of the seed window the groups are dealt (6 .. 10), for A/B runs
        static const int envWs = [] { const char* e = getenv("OLF_LSD_WS"); const int v = e ? atoi(e) : 0; return (v >= 6 && v <= 13) ? v : 0; }();
        const int wsBits = envWs ? envWs : 10;
        if (G > MG_MAX_G || E > MG_MAX_E || !b.mg || n_images > b.mgImages || poolLimit / G < E + 64) { set_error("launch_lsd_grow_mw: groups"); return OLF_ERR_INVALID; }
        // the control words of every image start at zero (watermarks 0 = nothing final yet; the notice words are set when a slot is filled)
        OLF_HIP_CHECK(hipMemset2DAsync(b.mg, b.mgStride, 0, MG_INVAL_OFF, (size_t)n_images, s));
        const int blocks = ((n_images + 7) / 8) * 8 * G;
        static const int envScatter = getenv("OLF_LSD_SCATTER") ? atoi(getenv("OLF_LSD_SCATTER")) : 0;       // (tools/stress_mg.py under scatter)
        const int scatter = b.scatter || envScatter;
        hipLaunchKernelGGL(k_lsd_grow_mw<true>, dim3(blocks), dim3(64 * nw), lds, s, b.geom, b.grad, b.owner, b.keysB, b.keyCount, b.region, b.links,
                           reinterpret_cast<RegionRec*>(b.keysA), b.regCount, b.status, b.angDeg, reinterpret_cast<const AngEnt*>(b.angEnt), E,
                           b.nChunks, poolLimit, b.growFmt, b.mg, b.mgStride, G, n_images, wsBits, ahead, scatter);
        hipLaunchKernelGGL(k_mg_merge, dim3(n_images), dim3(256), 0, s, b.geom, b.mg, b.mgStride, G, reinterpret_cast<RegionRec*>(b.keysA), b.regCount, b.growFmt);
    } else
        hipLaunchKernelGGL(k_lsd_grow_mw<false>, dim3(n_images), dim3(64 * nw), lds, s, b.geom, b.grad, b.owner, b.keysB, b.keyCount, b.region, b.links,
                           reinterpret_cast<RegionRec*>(b.keysA), b.regCount, b.status, b.angDeg, reinterpret_cast<const AngEnt*>(b.angEnt), E,
                           b.nChunks, poolLimit, b.growFmt, (unsigned char*)nullptr, (size_t)0, 1, n_images, 10, ahead, 0);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
