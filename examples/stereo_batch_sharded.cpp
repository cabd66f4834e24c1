// stereo_batch_sharded.cpp -- the batched offline mode behind the C ABI, for a C++ caller: one process per GPU, independent stereo pairs frame-sharded over the
// ranks, the packed feature records gathered to rank 0 with RCCL point-to-point over xGMI.  The shape of the reference's example driver
// (Examples/PL/PL_stereo_kitti.cc:66-108: load the image lists, loop over the frames, hand each pair to the tracker) with the per-frame feature work replaced by
// this library's batched entry; what rank 0 receives is what `Frame::Frame` would have computed for every pair (include/orbline.h, olf_frames_pack_dev).
//
//   make -C examples                     (needs hipcc + librccl; links ../orb_line_slam_amd/csrc/liborbline_hip.so and libolf_synth.so)
//   for r in 0 1 ... N-1: RANK=$r WORLD_SIZE=$N LOCAL_RANK=$r OLF_NCCL_ID_FILE=/tmp/olf_nccl_id ./stereo_batch_sharded [pairs per step] [steps] [W] [H] &
//
// No collective on the data path (src/Frame.cc:136-221 touches nothing outside its frame).  Per step and rank: olf_stereo_frames_dev -> olf_frames_pack_dev
// (trimmed record, device) -> the record's size joins a one-word ncclAllGather on the communication stream, copied to pinned host memory behind it -> ONE STEP
// LATER, while the next batch is being computed, the record goes to rank 0 (ncclSend; rank 0: grouped ncclRecv from every peer, each over its own link) with the
// sizes that have long arrived -- the host never waits for the device inside a step.  World size 1 runs the same code without a peer.
// Buffers are double: the pack of step k + 2 waits (stream s behind an event on the communication stream) until step k's record has left its buffer, and rank 0
// receives step k into recv[k % 2].  --verify (any position): rank 0 recomputes every rank's shard of the job from the seeds after the last step and compares
// the records it received (its own: the record it packed) byte for byte -- SURVEY 8(e) "Verification"; every rank then clears its record buffer before it packs,
// so that the alignment gaps between the record's sections are zero on both sides.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../include/orbline.h"

extern "C" int olf_synth_stereo(uint64_t seed, int W, int H, uint8_t* left, uint8_t* right);      // libolf_synth.so (csrc/synth.c): SURVEY 8(d)'s generator

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define NCCLCHK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "rank %d: %s: %s\n", g_rank, #x, ncclGetErrorString(r_)); exit(3); } } while (0)
#define OLFCHK(x) do { int r_ = (x); if (r_ != OLF_OK) { fprintf(stderr, "rank %d: %s: status %d: %s\n", g_rank, #x, r_, olf_last_error()); exit(4); } } while (0)
static int g_rank = 0;

static int env_int(const char* k, int dflt) { const char* v = getenv(k); return v ? atoi(v) : dflt; }

// contiguous block of frames of a rank (orb_line_slam_amd/distributed.py shard_range)
static void shard_range(long n, int rank, int world, long* lo, long* hi)
{
    const long base = n / world, rem = n % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
}

// the communicator's id from rank 0 to the others through a file on the node (one node: the 8 GPUs of north_star's batched mode)
static ncclUniqueId exchange_id(int rank, int world)
{
    ncclUniqueId id;
    const char* path = getenv("OLF_NCCL_ID_FILE");
    std::string p = path ? path : "/tmp/olf_nccl_id";
    if (rank == 0) {
        NCCLCHK(ncclGetUniqueId(&id));
        if (world > 1) {
            FILE* f = fopen((p + ".tmp").c_str(), "wb");
            if (!f || fwrite(&id, sizeof id, 1, f) != 1) { fprintf(stderr, "cannot write %s\n", p.c_str()); exit(5); }
            fclose(f);
            rename((p + ".tmp").c_str(), p.c_str());
        }
        return id;
    }
    for (int tries = 0; tries < 600; ++tries) {
        FILE* f = fopen(p.c_str(), "rb");
        if (f) { const size_t n = fread(&id, sizeof id, 1, f); fclose(f); if (n == 1) return id; }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    fprintf(stderr, "rank %d: no communicator id in %s after 60 s\n", rank, p.c_str());
    exit(5);
}

int main(int argc, char** argv)
{
    const int rank = env_int("RANK", 0), world = env_int("WORLD_SIZE", 1), local = env_int("LOCAL_RANK", rank);
    g_rank = rank;
    bool verify = false;
    std::vector<const char*> pos;
    for (int a = 1; a < argc; ++a) { if (!strcmp(argv[a], "--verify")) verify = true; else pos.push_back(argv[a]); }
    const int pairsTotal = pos.size() > 0 ? atoi(pos[0]) : 64 * world, steps = pos.size() > 1 ? atoi(pos[1]) : 3;
    const int W = pos.size() > 2 ? atoi(pos[2]) : 1242, H = pos.size() > 3 ? atoi(pos[3]) : 375;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (local >= ndev && world > 1) { fprintf(stderr, "rank %d: no GPU %d on this node (one device per rank)\n", rank, local); return 1; }
    HIPCHK(hipSetDevice(local % ndev));
    long lo, hi;
    shard_range(pairsTotal, rank, world, &lo, &hi);
    const int B = (int)(hi - lo), Bmax = (pairsTotal + world - 1) / world;

    ncclComm_t comm;
    const ncclUniqueId id = exchange_id(rank, world);
    NCCLCHK(ncclCommInitRank(&comm, world, id, rank));

    olf_params P;
    OLFCHK(olf_default_params(&P));
    P.orb.nfeatures = 2000; P.line.lsd_nfeatures = 500;
    olf_ctx* ctx = nullptr;
    OLFCHK(olf_ctx_create(&P, W, H, 2 * (Bmax > 0 ? Bmax : 1), &ctx));
    const size_t cap = (size_t)olf_orb_capacity(ctx), lcap = (size_t)olf_line_capacity(ctx), npx = (size_t)W * H;

    // this rank's frames: synthetic stereo pairs (frame f of the job has seed 7000 + f, whatever rank it lands on), resident in HBM
    std::vector<uint8_t> host((size_t)2 * (B > 0 ? B : 1) * npx);
    for (int q = 0; q < B; ++q) olf_synth_stereo(7000 + (uint64_t)(lo + q), W, H, host.data() + (size_t)2 * q * npx, host.data() + (size_t)(2 * q + 1) * npx);
    uint8_t* d_images = nullptr;
    HIPCHK(hipMalloc(&d_images, host.size()));
    HIPCHK(hipMemcpy(d_images, host.data(), host.size(), hipMemcpyHostToDevice));

    olf_frame_buffers fb;
    const size_t ni = 2 * (size_t)(Bmax > 0 ? Bmax : 1), np = ni / 2;
    HIPCHK(hipMalloc((void**)&fb.kps, ni * cap * sizeof(olf_keypoint))); HIPCHK(hipMalloc((void**)&fb.desc, ni * cap * 32)); HIPCHK(hipMalloc((void**)&fb.counts, ni * 4));
    HIPCHK(hipMalloc((void**)&fb.uright, np * cap * 4)); HIPCHK(hipMalloc((void**)&fb.depth, np * cap * 4));
    HIPCHK(hipMalloc((void**)&fb.kls, ni * lcap * sizeof(olf_keyline))); HIPCHK(hipMalloc((void**)&fb.ldesc, ni * lcap * 32)); HIPCHK(hipMalloc((void**)&fb.lcounts, ni * 4));
    HIPCHK(hipMalloc((void**)&fb.lmatches12, np * lcap * 4)); HIPCHK(hipMalloc((void**)&fb.ldisp, np * lcap * 8)); HIPCHK(hipMalloc((void**)&fb.lle, np * lcap * 24));

    hipStream_t s, sc;
    int prLeast = 0, prGreatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest));
    HIPCHK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, prGreatest));      // (a priority level of its own: streams of one level can share a hardware queue)
    const size_t bound = olf_frames_pack_bound(ctx, Bmax > 0 ? Bmax : 1);
    uint8_t* packed[2]; uint64_t* d_bytes[2]; uint64_t* d_sizes[2]; uint64_t* h_sizes[2]; hipEvent_t packedEv[2], sizesEv[2], sentEv[2];
    for (int k = 0; k < 2; ++k) {
        HIPCHK(hipMalloc(&packed[k], bound)); HIPCHK(hipMalloc((void**)&d_bytes[k], 8)); HIPCHK(hipMalloc((void**)&d_sizes[k], 8 * (size_t)world));
        HIPCHK(hipHostMalloc((void**)&h_sizes[k], 8 * (size_t)world, hipHostMallocDefault));
        HIPCHK(hipEventCreateWithFlags(&packedEv[k], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&sizesEv[k], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&sentEv[k], hipEventDisableTiming));
    }
    std::vector<uint8_t*> recv[2] = {std::vector<uint8_t*>(world, nullptr), std::vector<uint8_t*>(world, nullptr)};
    if (rank == 0) for (int k = 0; k < 2; ++k) for (int r = 1; r < world; ++r) HIPCHK(hipMalloc(&recv[k][r], bound));

    unsigned long long gathered = 0;
    // send / receive the records of step k (sizes asked for when the step was packed)
    auto transfer = [&](int k) {
        HIPCHK(hipEventSynchronize(sizesEv[k % 2]));            // complete long ago in a pipelined run
        const uint64_t* sz = h_sizes[k % 2];
        NCCLCHK(ncclGroupStart());
        if (rank == 0) { for (int r = 1; r < world; ++r) if (sz[r]) NCCLCHK(ncclRecv(recv[k % 2][r], sz[r], ncclUint8, r, comm, sc)); }
        else if (sz[rank]) NCCLCHK(ncclSend(packed[k % 2], sz[rank], ncclUint8, 0, comm, sc));
        NCCLCHK(ncclGroupEnd());
        // packed[k % 2] / d_bytes[k % 2] are free again once the communication stream has passed this point (the size exchange of step k sits in front of it
        // on the same stream): the pack of step k + 2 waits for it
        HIPCHK(hipEventRecord(sentEv[k % 2], sc));
        for (int r = 0; r < world; ++r) gathered += sz[r];
    };
    HIPCHK(hipDeviceSynchronize());
    NCCLCHK(ncclAllReduce(d_sizes[0], d_sizes[0], 1, ncclUint64, ncclSum, comm, sc));      // builds the rings before the clock starts
    HIPCHK(hipStreamSynchronize(sc));
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < steps; ++k) {
        if (B > 0) OLFCHK(olf_stereo_frames_dev(ctx, d_images, B, &fb, s));
        if (k >= 2) HIPCHK(hipStreamWaitEvent(s, sentEv[k % 2], 0));                          // step k - 2's record has left the buffer
        if (verify) HIPCHK(hipMemsetAsync(packed[k % 2], 0, bound, s));
        if (B > 0) OLFCHK(olf_frames_pack_dev(ctx, &fb, B, packed[k % 2], bound, d_bytes[k % 2], s));
        else HIPCHK(hipMemsetAsync(d_bytes[k % 2], 0, 8, s));                                // (an empty shard takes part in the size exchange with 0 bytes)
        HIPCHK(hipEventRecord(packedEv[k % 2], s));
        HIPCHK(hipStreamWaitEvent(sc, packedEv[k % 2], 0));
        NCCLCHK(ncclAllGather(d_bytes[k % 2], d_sizes[k % 2], 1, ncclUint64, comm, sc));
        HIPCHK(hipMemcpyAsync(h_sizes[k % 2], d_sizes[k % 2], 8 * (size_t)world, hipMemcpyDeviceToHost, sc));
        HIPCHK(hipEventRecord(sizesEv[k % 2], sc));
        if (k > 0) transfer(k - 1);                                                          // step k is running underneath
    }
    transfer(steps - 1);
    HIPCHK(hipStreamSynchronize(sc));
    HIPCHK(hipStreamSynchronize(s));
    OLFCHK(olf_ctx_synchronize(ctx));
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // --verify: what rank 0 holds of the last step against a recomputation of every rank's shard on this GPU (frame f of the job has seed 7000 + f)
    int verified = -1;
    if (verify && rank == 0) {
        verified = 1;
        const int last = (steps - 1) % 2;
        const uint64_t* sz = h_sizes[last];
        std::vector<uint8_t> got(bound), mine(bound);
        uint8_t* d_ver = nullptr; uint64_t* d_vb = nullptr;
        HIPCHK(hipMalloc(&d_ver, bound)); HIPCHK(hipMalloc((void**)&d_vb, 8));
        for (int r = 0; r < world; ++r) {
            long rlo, rhi;
            shard_range(pairsTotal, r, world, &rlo, &rhi);
            const int Br = (int)(rhi - rlo);
            uint64_t nb = 0;
            if (Br > 0) {
                for (int q = 0; q < Br; ++q) olf_synth_stereo(7000 + (uint64_t)(rlo + q), W, H, host.data() + (size_t)2 * q * npx, host.data() + (size_t)(2 * q + 1) * npx);
                HIPCHK(hipMemcpy(d_images, host.data(), (size_t)2 * Br * npx, hipMemcpyHostToDevice));
                OLFCHK(olf_stereo_frames_dev(ctx, d_images, Br, &fb, s));
                HIPCHK(hipMemsetAsync(d_ver, 0, bound, s));
                OLFCHK(olf_frames_pack_dev(ctx, &fb, Br, d_ver, bound, d_vb, s));
                HIPCHK(hipStreamSynchronize(s));
                HIPCHK(hipMemcpy(&nb, d_vb, 8, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(mine.data(), d_ver, nb, hipMemcpyDeviceToHost));
            }
            bool same = nb == sz[r];
            if (same && nb) {
                HIPCHK(hipMemcpy(got.data(), r == 0 ? packed[last] : recv[last][r], nb, hipMemcpyDeviceToHost));
                same = memcmp(got.data(), mine.data(), nb) == 0;
            }
            if (!same) { verified = 0; fprintf(stderr, "verify: the record of rank %d (%llu bytes received, %llu recomputed) differs\n", r, (unsigned long long)sz[r], (unsigned long long)nb); }
        }
        OLFCHK(olf_ctx_synchronize(ctx));
        HIPCHK(hipFree(d_ver)); HIPCHK(hipFree(d_vb));
    }
    char vtxt[64] = "null";
    if (verified >= 0) snprintf(vtxt, sizeof vtxt, "{\"ranks\": %d, \"identical\": %s}", world, verified ? "true" : "false");
    if (rank == 0)
        printf("{\"ranks\": %d, \"pairs_per_step\": %d, \"steps\": %d, \"stereo_frames_per_s\": %.1f, \"record_bytes_all_ranks\": %llu, \"image\": \"%dx%d\", \"verify\": %s}\n", world, pairsTotal,
               steps, (double)pairsTotal * steps / dt, gathered, W, H, vtxt);
    if (verified == 0) return 6;
    NCCLCHK(ncclCommDestroy(comm));
    olf_ctx_destroy(ctx);
    return 0;
}
