// lsd_grow_lanes.hip -- region growing of cv::LineSegmentDetector (oracle/line_oracle.cpp:121-148) for BIG batches: one wave per image, one
// LANE per region, up to 64 regions of an image growing at once.
//
// Same ordered speculation as lsd_grow.hip (rank-valued owner words claimed with atomicMin, a reorder buffer committed in rank order, a
// younger region yields to an older unfinished one it runs into, a region that loses a pixel is re-run; tools/lsd_sim.cpp), but the unit
// that grows a region is a lane instead of a wave.  The sequential work of region_grow -- per added pixel two float<-double additions and
// one fastAtan2 -- is paid once per wave INSTRUCTION in the one-wave agent (lsd.hip) and in the multi-wave kernel; here one instruction
// stream serves up to 64 regions, which is what makes a large batch cheaper per image (the batch is throughput bound: DESIGN.md 3.4).
//   lane state machine, one transition per step:
//     k == 8: take the next pixel of the region's list, load its 3 x 3 neighbourhood (gradient word, owner, level-line angle: three 12-byte
//             row loads per array) and classify the eight neighbours (undefined / mine / used by a final region -> skip; free or held by a
//             younger region -> candidate; held by an older unfinished region -> contested candidate);
//     k <  8: find the first candidate at or after position k that is aligned with the running region angle -- neighbours are visited in
//             the reference's order (row by row), every one tested against the angle after all additions before it, exactly as in the
//             reference -- and add it (claim, sums, fastAtan2, list); a contested one makes the region yield.
//   wave-cooperative, between steps: commit at the ROB head, hand eligible ROB entries to idle lanes (re-runs first, then fresh seeds),
//   dispatch the next 64 keys.  One wave owns an image, so the ROB needs no locks.
#include "lsd_device.hpp"

namespace olf {

#ifdef OLF_LN_DEBUG
__device__ int g_ln_log[4 * 65536];
__device__ int g_ln_logn;
#define LN_LOG(a, b, c, d) do { if (img == 0) { const int _q = atomicAdd(&g_ln_logn, 1); if (_q < 65536) { g_ln_log[4 * _q] = (a); g_ln_log[4 * _q + 1] = (b); g_ln_log[4 * _q + 2] = (c); g_ln_log[4 * _q + 3] = (d); } } } while (0)
#else
#define LN_LOG(a, b, c, d)
#endif

constexpr int LN_E = 1024;                    // ROB entries (= the 10 slot bits of an owner tag); chunk ids below LN_E are the slots' first chunks
constexpr int LN_SLOT_BITS = 10;
constexpr uint32_t LN_FREE = 0xffffffffu;
constexpr uint32_t LN_RETRY = 0xfffffffeu;    // "blocker" of a region that may be re-run at once
enum { LS_EMPTY = 0, LS_READY = 1, LS_PARKED = 2, LS_GROWING = 3, LS_DONE = 4, LS_DEAD = 5 };

#define LW_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define LW_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)

struct LnRec { int rank; uint32_t seed; float deg, sx, sy; int n; double ang; };      // 32 bytes, one per ROB slot and image (global memory)

__device__ __forceinline__ int lu(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
    return (uint32_t)lu((int)v);
}

// three consecutive words at a 4-byte aligned address
struct W3 { uint32_t a, b, c; };
__device__ __forceinline__ W3 load3(const uint32_t* p)
{
    W3 r;
    __builtin_memcpy(&r, p, 12);
    return r;
}
__device__ __forceinline__ W3 load3_wg(const uint32_t* p)      // owner words: another lane's atomic may have changed them -> not from the L1
{
    W3 r;
    r.a = LW_LOAD(p); r.b = LW_LOAD(p + 1); r.c = LW_LOAD(p + 2);
    return r;
}
__device__ __forceinline__ uint32_t pick3(const W3& w, int idx) { return idx == 0 ? w.a : idx == 1 ? w.b : w.c; }

__global__ __launch_bounds__(64) void k_lsd_grow_lanes(const LineGeom* __restrict__ gp, const uint32_t* __restrict__ gradAll, uint32_t* __restrict__ ownerAll,
                                                        const uint32_t* __restrict__ degAll, const uint32_t* __restrict__ keysAll,
                                                        const int* __restrict__ keyCount, uint32_t* __restrict__ chunksAll, int* __restrict__ linksAll,
                                                        LnRec* __restrict__ robAll, RegionRec* __restrict__ recsAll, int* __restrict__ regCount,
                                                        int* __restrict__ status, const float* __restrict__ angDeg, const AngEnt* __restrict__ ent,
                                                        int nChunks, int maxLanes)
{
    __shared__ int eState[LN_E];
    __shared__ uint32_t eInval[LN_E];          // DONE: lowest rank that stole from the region (FREE: none); PARKED: rank of the region it waits for
    __shared__ int s_tmp[64];
    __shared__ int s_pool;
    const LineGeom& g = *gp;
    const int img = blockIdx.x, lane = threadIdx.x;
    const int Ws = g.Ws, Hs = g.Hs, mask = LN_E - 1;
    const uint32_t* grad = gradAll + (size_t)img * g.Ps;
    uint32_t* owner = ownerAll + (size_t)img * g.Ps;
    const uint32_t* degp = degAll + (size_t)img * g.Ps;
    const uint32_t* keys = keysAll + (size_t)img * g.Ps;
    uint32_t* chunks = chunksAll + (size_t)img * nChunks * 32;
    int* links = linksAll + (size_t)img * nChunks;
    LnRec* rob = robAll + (size_t)img * LN_E;
    RegionRec* recs = recsAll + (size_t)img * g.maxRegions;
    const int nkeys = keyCount[img * 32];
    const double prec = g.prec, precWrap = g.precWrap;
    for (int q = lane; q < LN_E; q += 64) { eState[q] = LS_EMPTY; eInval[q] = LN_FREE; }
    if (lane == 0) s_pool = LN_E;
    __builtin_amdgcn_wave_barrier();

    // wave-uniform control state
    int head = 0, tail = 0, readyCur = 0, dispNext = 0, nreg = 0, idleSteps = 0;
    uint32_t wm = 0;
    bool reScan = false, fatal = false;
    uint32_t minBlk = LN_FREE;      // lower bound of the ranks the parked / stolen-from entries wait for: the ROB is only searched for re-runs once the watermark has passed it
    // lane state
    int slot = -1, n = 0, i = 0, k = 8, cur = 0, rchunk = 0, ex = 0, ey = 0, prevState = LS_READY;
    uint32_t T = 0, rank = 0, seedw = 0;
    double reg_angle = 0;
    float sumdx = 0.f, sumdy = 0.f;
    float nbDeg[8];
    uint32_t nbGxgy[8];
    uint32_t candM = 0, conM = 0;
    uint32_t pOld = LN_FREE;
    int pMine = 0;                 // 2: claim issued in this step, 1: issued one step ago (its result is looked at now), 0: none pending
    bool pSeed = false;            // the pending claim is the seed's own
    int relN = 0;                  // > 0: the list of a previous run of this entry still has to be released before the region starts
    bool starting = false;

    int step = 0;
#ifdef OLF_LN_DEBUG
    int dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DBG(i, v) dbg[i] += (v)
#else
#define DBG(i, v)
#endif
    for (; step < (1 << 22); ++step) {
        // ---- (1) commit at the head, in rank order -------------------------------------------------------------------------------------
        if (head < tail) {
            const int idx = head + lane, sl = idx & mask;
            const bool in = idx < tail;
            const int st = in ? eState[sl] : (int)LS_EMPTY;
            const uint32_t iv = in ? eInval[sl] : 0u;
            const bool can = st == LS_DEAD || (st == LS_DONE && iv == LN_FREE);
            const unsigned long long cm = __ballot(can);
            const int run = cm == ~0ull ? 64 : __builtin_ctzll(~cm);
            if (run > 0) {
                const int nn = (in && st == LS_DONE && lane < run) ? LW_LOAD(&rob[sl].n) : 0;
                unsigned long long big = __ballot(nn >= g.minRegSize);
                while (big) {
                    const int l = __builtin_ctzll(big);
                    big &= big - 1ull;
                    const int s2 = (head + l) & mask;
                    int nc = -1;
                    if (nreg < g.maxRegions) { if (lane == 0) nc = __hip_atomic_fetch_add(&s_pool, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); nc = lu(nc); }
                    if (nc < 0 || nc >= nChunks) { if (lane == 0) atomicOr(status, 8); fatal = true; break; }
                    // the first 32 pixels sit in the slot's own chunk, which the next seed in this slot will overwrite: move them to a pool chunk
                    if (lane < 32) chunks[(size_t)nc * 32 + lane] = LW_LOAD(chunks + (size_t)s2 * 32 + lane);
                    if (lane == 0) {
                        links[nc] = LW_LOAD(links + s2);
                        RegionRec rr; rr.start = nc; rr.n = rlane(nn, l);
                        rr.angle = __hip_atomic_load(&rob[s2].ang, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        recs[nreg] = rr;
                    }
                    ++nreg;
                }
                if (lane < run) eState[sl] = LS_EMPTY;
                head += run; DBG(0, run);
                readyCur = max(readyCur, head);
                const uint32_t nwm = head < tail ? (uint32_t)lu(LW_LOAD(&rob[head & mask].rank)) : (uint32_t)dispNext;
                if (nwm != wm) { wm = nwm; reScan = true; }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (fatal) break;
        // ---- (2) work for idle lanes: re-runs (older) first, then fresh seeds, then the next window of keys -----------------------------
        const bool worker = lane < maxLanes;               // (debug: fewer than 64 lanes grow regions)
        unsigned long long idleM = __ballot(slot < 0 && worker);
        if (idleM && reScan && (wm > minBlk || minBlk == LN_RETRY)) {
            bool left = false;
            uint32_t newMin = LN_FREE;
            // (the whole ROB: an entry that was parked when it was dispatched may lie beyond the cursor of the fresh seeds)
            for (int base = head; base < tail && idleM; base += 64) {
                const int idx = base + lane, sl = idx & mask;
                const int st = idx < tail ? eState[sl] : (int)LS_EMPTY;
                const uint32_t bl = eInval[sl];
                const bool waiting = (st == LS_PARKED || st == LS_DONE) && bl != LN_FREE;
                const bool el = waiting && (bl < wm || bl == LN_RETRY);
                newMin = min(newMin, (waiting && !el) ? bl : LN_FREE);
                const unsigned long long em = __ballot(el);
                if (!em) continue;
                if (el) s_tmp[__popcll(em & ((1ull << lane) - 1ull))] = sl | (st << 16);
                __builtin_amdgcn_wave_barrier();
                const int ne = __popcll(em), ni = __popcll(idleM);
                const int myIdle = __popcll(idleM & ((1ull << lane) - 1ull));
                DBG(1, min(ne, ni));
                if (slot < 0 && worker && myIdle < ne) { const int v = s_tmp[myIdle]; slot = v & 0xffff; prevState = v >> 16; starting = true; eState[slot] = LS_GROWING; }
                if (ne > ni) left = true;
                idleM = __ballot(slot < 0 && worker);
                __builtin_amdgcn_wave_barrier();
            }
            if (!left && idleM) { reScan = false; minBlk = wave_min(newMin); }      // a complete pass: what is still waiting, waits for at least this
        }
        for (int pass = 0; pass < 2 && idleM; ++pass) {
            // fresh seeds: READY entries at or after readyCur
            while (idleM && readyCur < tail) {
                const int idx = readyCur + lane, sl = idx & mask;
                const bool rd = idx < tail && eState[sl] == LS_READY;
                const unsigned long long rm = __ballot(rd);
                const int span = min(64, tail - readyCur);
                if (!rm) { readyCur += span; continue; }
                if (rd) s_tmp[__popcll(rm & ((1ull << lane) - 1ull))] = lane;
                __builtin_amdgcn_wave_barrier();
                const int nr = __popcll(rm), ni = __popcll(idleM);
                const int take = min(nr, ni);
                const int myIdle = __popcll(idleM & ((1ull << lane) - 1ull));
                int lastTaken = 0; DBG(2, take);
                if (slot < 0 && worker && myIdle < take) { const int e = s_tmp[myIdle]; slot = (readyCur + e) & mask; prevState = LS_READY; starting = true; eState[slot] = LS_GROWING; }
                lastTaken = s_tmp[take - 1];
                __builtin_amdgcn_wave_barrier();
                readyCur += take == nr ? span : lu(lastTaken) + 1;
                idleM = __ballot(slot < 0 && worker);
            }
            // the next 64 keys -> ROB entries for the seeds that are not already consumed by a final region
            if (!idleM || pass == 1 || dispNext >= nkeys || tail - head > LN_E - 64) break;
            {
                const int rk = dispNext + lane;
                const bool valid = rk < nkeys;
                const int addr = valid ? (int)(keys[rk] & 0x3fffffu) : 0;
                const uint32_t o = valid ? LW_LOAD(owner + addr) : 0u;
                const uint32_t w = valid ? grad[addr] : 0u;
                const bool iso = (w & kIso) != 0;
                const bool live = valid && !(o != LN_FREE && (o >> LN_SLOT_BITS) < wm);
                const unsigned long long m = __ballot(live);
                if (live) {
                    const int s = (tail + __popcll(m & ((1ull << lane) - 1ull))) & mask;
                    LnRec r;
                    r.rank = rk; r.seed = (uint32_t)addr | (iso ? 0x80000000u : 0u); r.n = 0; r.ang = 0;
                    r.deg = iso ? 0.f : angDeg[w & 0x3fffffu];
                    const float2 ss = iso ? make_float2(0.f, 0.f) : ent[w & 0x3fffffu].seed;
                    r.sx = ss.x; r.sy = ss.y;
                    rob[s] = r;
                    int st = LS_READY;
                    uint32_t inv = LN_FREE;
                    if (o != LN_FREE) { st = LS_PARKED; inv = o >> LN_SLOT_BITS; }
#ifndef OLF_LN_NOISO
                    else if (iso) {
                        // isolated seed (k_lsd_iso): the region is the seed alone -- claim it here
                        const uint32_t Ts = ((uint32_t)rk << LN_SLOT_BITS) | (uint32_t)s;
                        const uint32_t old = __hip_atomic_fetch_min(owner + addr, Ts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (old < Ts) { st = LS_PARKED; inv = old >> LN_SLOT_BITS; }
                        else {
                            if (old != LN_FREE) __hip_atomic_fetch_min(eInval + (old & mask), (uint32_t)rk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            st = LS_DONE;
                            LW_STORE(&rob[s].n, 1);
                        }
                    }
#endif
                    eInval[s] = inv;
                    eState[s] = st;
                }
                { const uint32_t pb = wave_min(live && eState[(tail + __popcll(m & ((1ull << lane) - 1ull))) & mask] == LS_PARKED ? eInval[(tail + __popcll(m & ((1ull << lane) - 1ull))) & mask] : LN_FREE); if (pb < minBlk) { minBlk = pb; reScan = true; } }
                tail += (int)__popcll(m); DBG(3, 1);
                dispNext = min(dispNext + 64, nkeys);
                if (head == tail) { wm = (uint32_t)dispNext; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---- (3) lanes that were just handed an entry: read its record, start (or first release what a previous run left) ---------------
        if (starting) {
            // (the slot's previous occupant may still sit in this CU's L1: read the record from the L2)
            const uint32_t* rw = reinterpret_cast<const uint32_t*>(&rob[slot]);
            rank = LW_LOAD(rw); seedw = LW_LOAD(rw + 1);
            const float rdeg = __uint_as_float(LW_LOAD(rw + 2));
            sumdx = __uint_as_float(LW_LOAD(rw + 3)); sumdy = __uint_as_float(LW_LOAD(rw + 4));
            T = (rank << LN_SLOT_BITS) | (uint32_t)slot;
            relN = prevState == LS_READY ? 0 : (int)LW_LOAD(rw + 5);
            reg_angle = d_mul((double)rdeg, kDegToRads);
        }
        // release of lists (re-run of a finished region that was stolen from; regions given up in this step are released further down)
        while (__ballot(starting && relN > 0)) {
            if (starting && relN > 0) {
                // pixel 0 is the seed; 1 .. relN-1 from the chunk chain
                int c = slot;
                for (int q = 1; q < relN; ++q) {
                    if ((q & 31) == 0) c = LW_LOAD(links + c);
                    const uint32_t xy = LW_LOAD(chunks + (size_t)c * 32 + (q & 31));
                    uint32_t exp = T;
                    __hip_atomic_compare_exchange_strong(owner + ((int)(xy >> 16) * Ws + (int)(xy & 0xffffu)), &exp, LN_FREE, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                uint32_t exp = T;
                __hip_atomic_compare_exchange_strong(owner + (seedw & 0x3fffffu), &exp, LN_FREE, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                LW_STORE(&rob[slot].n, 0);
                relN = 0;
            }
        }
        if (starting) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            eInval[slot] = LN_FREE;                       // steals from here on concern this run
            const int seed = (int)(seedw & 0x3fffffu);
            pOld = __hip_atomic_fetch_min(owner + seed, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pMine = 2; pSeed = true;
            const uint32_t seedXY = (uint32_t)(seed % Ws) | ((uint32_t)(seed / Ws) << 16);
            chunks[(size_t)slot * 32] = seedXY;
            links[slot] = -1;
            n = 1; i = 0; k = 8; cur = slot; rchunk = slot;
            LN_LOG(1, (int)rank, prevState, relN);
            starting = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        // ---- (4) one transition per growing lane -------------------------------------------------------------------------------------
        bool fail = false, seedLost = false, told = false;
        uint32_t blocker = LN_FREE;
        const bool active = slot >= 0;
        if (active) {
            const uint32_t iv = eInval[slot];
            if (iv != LN_FREE) { fail = true; blocker = iv; }
            if (pMine == 2) pMine = 1;
            else if (pMine == 1) {
                // the claim issued one step ago: an older region had taken the pixel in between -> the decisions since were made on a pixel that was
                // not available: yield; a younger region's pixel is ours now, and that region is told
                if (pOld != LN_FREE && pOld > T) { __hip_atomic_fetch_min(eInval + (pOld & mask), rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); told = true; }
                if (pOld <= T) { fail = true; blocker = pOld == T ? LN_RETRY : (pOld >> LN_SLOT_BITS); seedLost = pSeed && pOld != T; }
                pMine = 0;
            }
        }
        bool finished = false;
        if (active && !fail) {
            if ((seedw & 0x80000000u) != 0u) finished = pMine == 0;        // isolated seed (re-run): the claim held, the region is the seed alone
            else if (k == 8) {
                if (i == n) finished = pMine == 0;                         // (the last addition's claim is looked at first)
                else {
                    if (i > 0 && (i & 31) == 0) rchunk = LW_LOAD(links + rchunk);
                    const uint32_t xy = LW_LOAD(chunks + (size_t)rchunk * 32 + (i & 31));
                    ++i;
                    ex = (int)(xy & 0xffffu); ey = (int)(xy >> 16);
                    const int xb = min(max(ex - 1, 0), Ws - 3), sh = (ex - 1) - xb;      // sh: -1 at the left border, +1 at the right one
                    candM = 0; conM = 0;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const int yy = ey + r - 1;
                        const bool rowOk = yy >= 0 && yy < Hs;
                        const size_t ro = (size_t)(rowOk ? yy : ey) * Ws + xb;
                        const W3 gw = load3(grad + ro), dw = load3(degp + ro);
                        const W3 ow = load3_wg(owner + ro);
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            if (r == 1 && c == 1) continue;
                            const int j = r * 3 + c - (r * 3 + c > 4 ? 1 : 0);
                            const int idx = c + sh;
                            const bool ok = rowOk && idx >= 0 && idx <= 2;
                            const uint32_t pw = pick3(gw, idx), o = pick3(ow, idx);
                            nbDeg[j] = __uint_as_float(pick3(dw, idx));
                            nbGxgy[j] = pw & 0x3fffffu;
                            const bool cand = ok && !(pw & kNotDef) && o != T && !(o < T && (o >> LN_SLOT_BITS) < wm);
                            if (cand) { candM |= 1u << j; if (o < T) conM |= 1u << j; }
                        }
                    }
                    k = 0;
                }
            }
        }
        // (a lane holds one unverified claim at a time: in its first step that is the seed's, so nothing is added before the next step)
        if (active && !fail && !finished && k < 8 && pMine == 0) {
            // the first candidate at or after k that is aligned with the running angle (isaligned(): see k_lsd_grow)
            uint32_t al = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double nth = fabs(d_sub(reg_angle, d_mul((double)nbDeg[j], kDegToRads)));
                if (nth <= prec || nth >= precWrap) al |= 1u << j;
            }
            al &= candM & ~((1u << k) - 1u);
            if (!al) k = 8;
            else {
                const int js = __builtin_ctz(al);
                const int jj = js < 4 ? js : js + 1;
                const int ax = ex + (jj % 3) - 1, ay = ey + (jj / 3) - 1;
                const int a = ay * Ws + ax;
                if ((conM >> js) & 1u) {
                    // the reference would add a pixel that an older, unfinished region holds right now: yield to that region
                    fail = true;
                    const uint32_t o = LW_LOAD(owner + a);
                    blocker = o < T ? (o >> LN_SLOT_BITS) : LN_RETRY;          // (released in the meantime: simply try again)
                } else {
                    uint32_t ti = nbGxgy[0];
#pragma unroll
                    for (int j = 1; j < 8; ++j) ti = js == j ? nbGxgy[j] : ti;
                    const AngEnt* tp = ent + ti; const double2 t = make_double2(tp->cs, tp->sn);
                    sumdx = (float)d_add((double)sumdx, t.x);
                    sumdy = (float)d_add((double)sumdy, t.y);
                    reg_angle = d_mul((double)agent_fastAtan2(sumdy, sumdx), kDegToRads);
                    pOld = __hip_atomic_fetch_min(owner + a, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    pMine = 1; pSeed = false;
                    if ((n & 31) == 0) {
                        const int nc = __hip_atomic_fetch_add(&s_pool, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (nc >= nChunks) { atomicOr(status, 8); fail = true; blocker = LN_RETRY; fatal = true; }
                        else { links[cur] = nc; links[nc] = -1; cur = nc; }
                    }
                    if (!fail) { chunks[(size_t)cur * 32 + (n & 31)] = (uint32_t)ax | ((uint32_t)ay << 16); ++n; }
                    k = js + 1;
                }
            }
        }
        if (__ballot(fatal)) { fatal = true; break; }
        // ---- (5) regions that are complete / given up ------------------------------------------------------------------------------------
        if (finished) {
            LN_LOG(2, (int)rank, n, step);
            LW_STORE(&rob[slot].n, (seedw & 0x80000000u) ? 1 : n);
            __hip_atomic_store(&rob[slot].ang, reg_angle, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        DBG(4, (int)__popcll(__ballot(finished))); DBG(5, (int)__popcll(__ballot(fail)));
        if (finished) { eState[slot] = LS_DONE; slot = -1; }
        if (fail) {
            LN_LOG(3, (int)rank, n, (int)blocker);
            if (pMine == 1 && pOld != LN_FREE && pOld > T && pOld != T) {}   // (a failing lane has issued no claim in this step)
            pMine = 0;
            {   // (also when the seed itself was lost: the start step may already have added a neighbour before the seed's claim came back)
                int c = slot;
                for (int q = 1; q < n; ++q) {
                    if ((q & 31) == 0) c = LW_LOAD(links + c);
                    const uint32_t xy = LW_LOAD(chunks + (size_t)c * 32 + (q & 31));
                    uint32_t exp = T;
                    __hip_atomic_compare_exchange_strong(owner + ((int)(xy >> 16) * Ws + (int)(xy & 0xffffu)), &exp, LN_FREE, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                uint32_t exp = T;
                __hip_atomic_compare_exchange_strong(owner + (seedw & 0x3fffffu), &exp, LN_FREE, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            LW_STORE(&rob[slot].n, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        {
            const bool dead = fail && seedLost && blocker != LN_RETRY && blocker < wm;
            const bool park = fail && !dead;
            if (park) { eInval[slot] = blocker; eState[slot] = LS_PARKED; }
            if (dead) eState[slot] = LS_DEAD;
            const unsigned long long pm = __ballot(park);
            if (pm) { const uint32_t pb = wave_min(park ? (blocker == LN_RETRY ? 0u : blocker) : LN_FREE); if (pb < minBlk) minBlk = pb; reScan = true; }
            if (__ballot(told)) { const uint32_t tb = wave_min(told ? rank : LN_FREE); if (tb < minBlk) minBlk = tb; reScan = true; }
            if (fail) slot = -1;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- (6) done? -----------------------------------------------------------------------------------------------------------------
        if (!__ballot(slot >= 0)) {
            if (dispNext >= nkeys && head == tail) break;
            if (++idleSteps > (1 << 20)) { if (lane == 0) atomicOr(status, 16); fatal = true; break; }      // cannot happen: the oldest seed never waits
        } else idleSteps = 0;
    }
#ifdef OLF_LN_DEBUG
    if (img == 0) {
        if (lane == 0) {
            status[16] = head; status[17] = tail; status[18] = readyCur; status[19] = dispNext; status[20] = (int)wm; status[21] = nreg; status[22] = nkeys;
            status[23] = step; status[24] = fatal; status[25] = head < tail ? eState[head & mask] : -1; status[26] = head < tail ? (int)eInval[head & mask] : -1;
            for (int q = 0; q < 8; ++q) status[40 + q] = dbg[q];
            status[27] = head < tail ? rob[head & mask].rank : -1; status[28] = reScan; status[29] = head < tail ? rob[head & mask].n : -1;
        }
        status[64 + lane] = slot; status[128 + lane] = k | (i << 8); status[192 + lane] = n;
    }
#endif
    if (lane == 0) regCount[img] = fatal ? 0 : nreg;
}

#ifdef OLF_LN_DEBUG
extern "C" int olf_debug_ln_log(int* out, int cap)
{
    int n = 0;
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_ln_logn), 4);
    const int m = n < cap ? n : cap;
    if (m > 65536) return -1;
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ln_log), (size_t)16 * (m < 65536 ? m : 65536));
    const int z = 0;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ln_logn), &z, 4);
    return n;
}
#endif

int launch_lsd_grow_lanes(const LineGeom& g, LineDeviceBufs& b, int n_images, hipStream_t s)
{
    const int maxLanes = b.forceE > 0 && b.forceE <= 64 ? b.forceE : 64;
    hipLaunchKernelGGL(k_lsd_grow_lanes, dim3(n_images), dim3(64), 0, s, b.geom, b.grad, b.owner, b.deg, b.keysB, b.keyCount, b.region, b.links,
                       reinterpret_cast<LnRec*>(b.rob), reinterpret_cast<RegionRec*>(b.keysA), b.regCount, b.status, b.angDeg, reinterpret_cast<const AngEnt*>(b.angEnt), b.nChunks, maxLanes);
    OLF_HIP_CHECK(hipGetLastError());
    return OLF_OK;
}

}  // namespace olf
