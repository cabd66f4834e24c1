#!/usr/bin/env python3
"""bench.py -- stereo frames/s for extract + match (ORB + LBD) on synthetic stereo, 1..8 GPUs (BASELINE.json's metric).

One "step" = one pass of the whole hot path over one batch of B stereo pairs per GPU, inputs already resident in HBM:
olf_stereo_frames_dev (ExtractORB x2, ExtractLine x2, ComputeStereoMatches, ComputeStereoMatches_Lines; reference src/Frame.cc:136-221)
+ the frame-to-frame LBD match (match(), src/LineMatcher.cpp:104-132) and the frame-to-frame ORB match against the previous
frame of the batch: ORBmatcher::SearchByBoW (src/ORBmatcher.cc:161-290) batched on the device, ComputeBoW included (BASELINE.json config 3;
--no-bow: a dense kNN stand-in).  Workloads (--config): C2 640x480 1000+200, C3 KITTI 1242x375 2000+500 (default: the configuration
the metric is quoted on), C4 EuRoC 752x480 1200+500, C5 1920x1080 4000+1000.
The step issues the two matchers in the tracker's order (src/Tracking.cc:963-970 before :1296-1308): the fused call with the deferred join
(olf_ctx_set_deferred_join: the point-side outputs are complete on the stream, the line path's tail still runs on the context's line stream),
SearchByBoW on the point features, the join, the line matcher -- every kernel of a step is complete when its timed region ends
(--no-deferred-join: the joined call of the C ABI's default).  The harness runs on a stream of its own (the default stream's handle is NULL,
which the C ABI reads as "the context's stream").

Multi-GPU: frames are independent -> every rank processes its own B pairs (weak scaling, no data-path collective).
The design's one communication step runs INSIDE the timed region: after every step each rank packs the trimmed feature record of its
batch (csrc/records.hip) and sends it to rank 0 (orb_line_slam_amd/distributed.py: one-word all_gather of the sizes + point-to-point
sends, on a side stream, overlapped with the next step's kernels).  --verify: rank 0 re-runs every other rank's input (same seeds) and
byte-compares the records it received with its own.
--force-dist: the same N > 1 code path with ONE rank, so that the RCCL backend itself (bring-up, collectives on device tensors, barrier) runs on a 1-GPU box.

The JSON line also carries: the roofline of the dominant stage (the one with the largest stand-alone time; algorithmic bytes / HIP-event
duration of its launch inside the timed region; `traffic` = 2 x FETCH_SIZE + WRITE_SIZE and `valu_issue` = its vector instructions against the
part's issue rate, both from PMC summaries under profiles/ that carry the hash of the sources they were collected on), `roofline.stages`:
every stage run alone, a measured copy-kernel ceiling next to the 8 TB/s specification, host-to-host latency of small batches (1, 8, 128
pairs per call), the PCIe-inclusive rate of the double-buffered offline pipeline, and the CPU oracle timed on this box's host cores in two
shapes (A: 4 threads per frame, one frame at a time, like the reference, 20 warm-up + 200 timed frames with a per-stage breakdown; B: one
frame per thread as a thread-count sweep against the box's CPU quota).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29501 bench.py --gpus 8 --steps 5 --warmup 2
"""
import argparse
import ctypes as C
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {   # BASELINE.json configs / SURVEY.md 8(d): size, ORB features, LBD lines, fx, bf, default pairs per GPU per step
    "C2": dict(w=640, h=480, nf=1000, nl=200, fx=435.2047, bf=47.9064, pairs=3072),
    "C3": dict(w=1242, h=375, nf=2000, nl=500, fx=718.856, bf=386.1448, pairs=3072),
    "C4": dict(w=752, h=480, nf=1200, nl=500, fx=435.2047, bf=47.9064, pairs=3072),
    "C5": dict(w=1920, h=1080, nf=4000, nl=1000, fx=1050.0, bf=126.0, pairs=1536),
}
STAGE_KERNEL = {"orb_pyramid": "olf::k_resize_tiled", "orb_fast": "olf::k_fast_score", "orb_fast_cells": "olf::k_fast_score", "orb_octree": "olf::k_octree", "orb_blur": "olf::k_sep7",
                "orb_describe": "olf::k_describe", "stereo_points": "olf::k_stereo_match", "lsd_front": "olf::k_lsd_keys", "lsd_grow": "olf::k_lsd_grow<0",
                "lsd_rect": "olf::k_lsd_rect", "line_select_lbd": "olf::k_lbd_rows", "stereo_lines": "olf::k_lines_dist", "match_bf": "olf::k_knn2"}


# every kernel of a stage (name prefixes as rocprofv3 prints them, "void " stripped): the PMC traffic summary of the whole step is folded into roofline.stages with it.
# k_sep7_strip serves three stages (ORB blur, the LSD blur of lsd_front, the LBD blur of line_select_lbd) and is reported under orb_blur, its largest user.
STAGE_KERNELS = {"orb_pyramid": ["olf::k_ingest", "olf::k_resize_strip", "olf::k_resize_tiled"], "orb_fast_cells": ["olf::k_fast_score", "olf::k_cells_sort", "olf::k_fast_"],
                 "orb_octree": ["olf::k_octree"], "orb_blur": ["olf::k_sep7"], "orb_describe": ["olf::k_describe", "olf::k_ic_angle", "olf::k_orb_assemble", "olf::k_assemble"],
                 "stereo_points": ["olf::k_stereo_"], "lsd_front": ["olf::k_lsd_upgrad", "olf::k_lsd_keys", "olf::k_lsd_seedsort", "olf::k_lsd_grad", "olf::k_top_", "olf::k_radix_"],
                 "lsd_grow": ["olf::k_lsd_grow", "olf::k_mg_merge"], "lsd_rect": ["olf::k_lsd_rect", "olf::k_lsd_emit"],
                 "line_select_lbd": ["olf::k_line_select", "olf::k_sobel3", "olf::k_lbd_"], "stereo_lines": ["olf::k_lines_"],
                 "match": ["olf::k_knn2", "olf::k_ratio_mutual", "olf::k_search_by_bow", "olf::k_bow_", "olf::k_match_", "olf::k_depth_mask", "olf::k_transform"]}


def synthetic_vocabulary(k, L, sample_desc, seed=4242):
    """A full k-ary vocabulary tree of depth L in the arrays ORBVocabulary.from_arrays takes (parent, is_leaf, descriptor, weight), nodes in
    breadth-first order like TemplatedVocabulary::loadFromTextFile builds them (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1425).  The
    reference's ORBvoc.txt (k = 10, L = 6, 10^6 words) is missing from its checkout: the node descriptors here are drawn from the
    batch's own ORB descriptors, so a descent spreads the features over the level-2 nodes the way a trained tree does."""
    n = (k ** (L + 1) - 1) // (k - 1)
    parent = (np.arange(n, dtype=np.int64) - 1) // k
    parent[0] = -1
    leaf = np.zeros(n, np.uint8)
    leaf[n - k ** L:] = 1
    rng = np.random.default_rng(seed)
    desc = sample_desc[rng.integers(0, len(sample_desc), n)].copy()
    flip = rng.integers(0, 256, n)
    desc[np.arange(n), flip // 8] ^= (1 << (flip % 8)).astype(np.uint8)
    weight = np.where(leaf > 0, rng.uniform(0.5, 9.0, n), 0.0)
    return parent.astype(np.int32), leaf, desc, weight


def source_hash():
    """Identifies the kernel sources a PMC summary was collected on (.git does not travel to the GPU box)."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "orb_line_slam_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.hpp")) + glob.glob(os.path.join(d, "*.cpp"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(W, H, N, NL, mean_len, nlevels=8, sf=1.2):
    """SURVEY.md Appendix D per image and stage, re-evaluated for this build's layouts (DESIGN.md "byte model"): LSD keeps the integer
    gradient pair (4 B/px) instead of fp32 modgrad + angle (8 B/px); the pseudo-sort reads it once and writes 4-byte keys; region growing
    reads the gradient word (4) and reads + writes the used / owner state (2, as the reference's byte map); the std::sort seed order (round 3) reads
    the key of every pixel once more and writes the seed list (about a quarter of the pixels): 4 Ps + Ps."""
    lv, s = [], 1.0
    for _ in range(nlevels):
        lv.append((int(round(W / s)), int(round(H / s))))
        s *= sf
    P = [w * h for w, h in lv]
    Pp, P0, P7 = sum(P), P[0], P[-1]
    Ps = int(round(W * 1.2)) * int(round(H * 1.2))
    st = {"orb_pyramid": P0 + (Pp - P0) + (Pp - P7), "orb_fast": Pp, "orb_octree": 0, "orb_blur": 2 * Pp, "orb_describe": (749 + 512 + 32 + 28) * N,
          "stereo_points": 0, "lsd_front": 2 * P0 + (P0 + Ps) + (Ps + 4 * Ps) + (4 * Ps + 4 * Ps) + (4 * Ps + Ps), "lsd_grow": (4 + 2) * Ps, "lsd_rect": 0,
          "line_select_lbd": 2 * P0 + (P0 + 4 * P0) + 63 * mean_len * 4 * NL + (32 + 68) * NL, "stereo_lines": 0, "match_bf": 0}
    orb = st["orb_pyramid"] + st["orb_fast"] + st["orb_blur"] + st["orb_describe"]
    lsd = st["lsd_front"] + st["lsd_grow"]
    return dict(stage=st, orb=orb, lsd=lsd, lbd=st["line_select_lbd"], pair=2 * (orb + lsd + st["line_select_lbd"]), Ps=Ps, Pp=Pp)


def cpu_baseline(W, H, params, seconds_budget=20.0):
    """The CPU oracle (oracle/liboracle_fast.so, -O3 -march=native -ffp-contract=off; the reference itself cannot be built: OpenCV / Eigen /
    Pangolin are not in the image) timed on this box's host cores.  Mode A, the reference's shape: 4 std::threads per frame
    (src/Frame.cc:164-171), frames one at a time -- the per-frame latency.  Mode B, throughput: one frame per core on all cores."""
    import subprocess
    subprocess.run(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "liboracle_fast.so"], check=True)   # -march=native: always rebuilt on the box that runs it
    L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_fast.so"))
    from orb_line_slam_amd import synth
    times, stages, t_all, seed = [], [], time.time(), 1000
    L.orc_frame_stage_ms.argtypes = [C.c_void_p]
    L.orc_frame_orb_stage_ms.argtypes = [C.c_void_p]
    orb12, orb_stages = np.zeros(12, np.float64), []
    WARM, TIMED = 20, 200                       # at least 200 timed frames after 20 warm-up frames
    st8 = np.zeros(8, np.float64)
    while len(times) < WARM + TIMED:
        l, r = synth.stereo_pair(seed, W, H)
        seed += 1
        n = [C.c_int() for _ in range(4)]
        t = time.perf_counter()
        rc = L.orc_stereo_frame(l.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), W, H, C.byref(params), 4,
                                None, None, C.byref(n[0]), None, None, C.byref(n[1]), 1 << 20, None, None,
                                None, None, C.byref(n[2]), None, None, C.byref(n[3]), 1 << 20, None, None, None)
        times.append(time.perf_counter() - t)
        assert rc == 0
        L.orc_frame_stage_ms(st8.ctypes.data_as(C.c_void_p))
        stages.append(st8.copy())
        L.orc_frame_orb_stage_ms(orb12.ctypes.data_as(C.c_void_p))
        orb_stages.append(orb12.copy())
        if time.time() - t_all > max(seconds_budget, 10.0) * 3 and len(times) >= WARM + 20:      # a very slow host: keep the default run bounded
            break
    times = np.array(times[WARM:]); stages = np.array(stages[WARM:])
    med = float(np.median(times))
    sm = np.median(stages, axis=0)
    om = np.median(np.array(orb_stages[WARM:]), axis=0)
    out = {"value": round(1.0 / med, 3), "unit": "stereo frames/s", "cores": 4, "kind": "port",
           "sample": f"{len(times)} synthetic {W}x{H} stereo pairs after {WARM} warm-up frames, one at a time, 4 threads/frame like src/Frame.cc:164-171, "
                     f"median {med * 1e3:.1f} ms/frame (mean {times.mean() * 1e3:.1f}, p95 {np.percentile(times, 95) * 1e3:.1f})",
           "host_cores": os.cpu_count(),
           # wall time per stage, median over the timed frames (the four extractions run concurrently on their own threads, the two stereo
           # matchers after them on the calling thread): the frame time is about max(extractions) + stereo points + stereo lines
           "stages_ms": {"orb_left": round(sm[0], 2), "orb_right": round(sm[1], 2), "lsd_left": round(sm[2], 2), "lbd_left": round(sm[3], 2),
                         "lsd_right": round(sm[4], 2), "lbd_right": round(sm[5], 2), "stereo_points": round(sm[6], 2), "stereo_lines": round(sm[7], 2),
                         # orb_left by stage (the port's FAST has FAST_t<16>'s opposite-pair pre-test since round 6)
                         **{"orb_left_" + k: round(float(om[i]), 2) for i, k in enumerate(("pyramid", "fast", "octree", "ic_angle", "blur", "rbrief"))}}}
    # Mode B: throughput, one whole frame per thread at a time (threads = 1 inside a frame), at 1, 4, 16, 64, ... all host threads: does it scale?
    try:
        cores = len(os.sched_getaffinity(0))
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            quota = None if q == "max" else round(int(q) / int(per), 2)
        except Exception:
            pass
        nd = 16
        imgs = synth.stereo_batch(2000, nd, W, H)
        L.orc_stereo_frames_throughput.restype = C.c_double
        L.orc_stereo_frames_throughput.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        sweep = {}
        for th in sorted({1, 4, 16, 64, cores} & set(range(1, cores + 1))):
            frames = max(4 * th, 8)              # four frames per worker: the first one warms its arena
            fps = float(L.orc_stereo_frames_throughput(imgs.ctypes.data_as(C.c_void_p), nd, W, H, C.byref(params), th, frames))
            sweep[str(th)] = {"fps": round(fps, 2), "fps_per_thread": round(fps / th, 3), "frames": frames}
        best_th = max(sweep, key=lambda k: sweep[k]["fps"])
        best = sweep[best_th]
        out["mode_b"] = {"value": best["fps"], "unit": "stereo frames/s", "cores": int(best_th), "host_threads": cores, "cpu_quota_cores": quota, "threads_sweep": sweep,
                         "sample": f"one frame per thread, 4 frames per thread ({nd} distinct pairs), thread counts {list(sweep)}; affinity {cores} of "
                                   f"{os.cpu_count()} logical CPUs, cgroup cpu.max quota {quota}; per-thread malloc arenas keep freed pages (oracle/frame_oracle.cpp)"}
    except Exception as e:   # the throughput leg is extra information: never lose the line over it
        out["mode_b"] = {"error": str(e)}
    return out


def small_batch_latency(params, W, H, sizes=(1, 8, 128)):
    """Host buffers in, host buffers out (olf_stereo_frames, what a Frame constructor would call): milliseconds per call."""
    import orb_line_slam_amd as ola
    from orb_line_slam_amd import synth
    out = {}
    for n in sizes:
        fe = ola.StereoFrontEnd(params, W, H, max_pairs=n)
        imgs = synth.stereo_batch(11, min(n, 16), W, H)
        imgs = np.tile(imgs, ((n + 15) // 16, 1, 1))[:2 * n].copy()
        for _ in range(3 if n <= 8 else 1):      # (the first calls load code objects and set function attributes)
            fe.frames(imgs)
        ts = []
        for _ in range(9 if n <= 8 else 3):
            t = time.perf_counter()
            fe.frames(imgs)
            ts.append(time.perf_counter() - t)
        out[str(n)] = round(float(np.median(ts)) * 1e3, 3)      # median of the calls (rounds 1-4: mean of 5 behind one warm-up call)
        fe.ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C3")
    ap.add_argument("--pairs", type=int, default=0, help="stereo pairs per GPU per step (0: the configuration's default)")
    ap.add_argument("--distinct", type=int, default=512, help="distinct synthetic pairs per rank (tiled to --pairs)")
    ap.add_argument("--order", choices=("shuffled", "tiled"), default="shuffled", help="how the batch is drawn from the distinct pairs")
    ap.add_argument("--scene", choices=("default", "long", "bars"), default="default", help="long: fewer, larger shapes; bars: long thin bars -> key lines of about 0.08*W pixels (SURVEY App. D model)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bow", action="store_true", help="frame-to-frame ORB matching by the dense kNN stand-in of rounds 1-2 instead of SearchByBoW")
    ap.add_argument("--no-deferred-join", action="store_true", help="join the two streams at the end of olf_stereo_frames_dev (the default of the C ABI) instead of behind the "
                                                                    "point features' matcher (olf_ctx_set_deferred_join, include/orbline.h)")
    ap.add_argument("--pipeline", action="store_true", help="hand the context an input-ready event (olf_ctx_set_input_event, include/orbline.h): the line path of step k + 1 then starts "
                                                            "beside the tail of step k instead of behind it (measured: no gain, profiles/r4q_pipelined_steps_ab.txt)")
    ap.add_argument("--no-isolated", action="store_true", help="skip the pass that runs every stage alone (counter collections that must see exactly the timed steps)")
    ap.add_argument("--no-extras", action="store_true", help="skip the copy ceiling, small-batch latency and PCIe-inclusive legs")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--gather", choices=("overlap", "sync", "off"), default="overlap", help="N>1: gather of the feature records to rank 0 (inside the timed region)")
    ap.add_argument("--seed-order", type=int, choices=(0, 1), default=None, help="convention C.9: order of the LSD seeds inside a gradient bin (default: the library's)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="N>1: nccl = RCCL over xGMI (device buffers); gloo = the same gather staged through host memory (lets 2 ranks share one GPU: the N>1 code path on a 1-GPU box)")
    ap.add_argument("--force-dist", action="store_true", help="run the N>1 code path (process group bring-up, batch-size agreement, pack, size exchange, gather, per-rank timing) "
                    "even with one rank: on a 1-GPU box this is the only way the RCCL backend itself executes (no peer, so no point-to-point transfer)")
    ap.add_argument("--verify", action="store_true", help="N>1: rank 0 recomputes every rank's records of the last step and byte-compares them with what it received")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak", help="N>1: weak = every rank runs the configuration's batch (the driver's contract, value ~ N x); "
                    "strong = the configuration's batch is the whole job and is split over the ranks (fixed total work: value / value(N=1) / N is the scaling efficiency)")
    ap.add_argument("--camera-rgb", type=int, default=1, choices=(0, 1), help="--images with colour files: the yaml's Camera.RGB (src/Tracking.cc:193-218; 1 in every KITTI / EuRoC "
                    "configuration of the reference: CV_RGB2GRAY applied to imread's BGR data)")
    ap.add_argument("--images", default=None, help="recorded stereo sequence instead of synthetic input: a KITTI sequence directory (times.txt, image_0, image_1 -- "
                    "Examples/PL/PL_stereo_kitti.cc LoadImages), a EuRoC mav0 directory or any directory with left / right image folders (orb_line_slam_amd/sequence.py); "
                    "the image size comes from the files, the feature counts from --config; fewer pairs than the batch are tiled")
    ap.add_argument("--sequence", type=int, default=0, metavar="RUN", help="synthetic input as runs of RUN consecutive frames of one scene (frame k of a run = the scene "
                    "shifted by 2k pixels), so that the frame-to-frame matchers see related frames; 0: independent scenes (SURVEY 8(d)'s generator)")
    args = ap.parse_args()

    import torch
    import orb_line_slam_amd as ola
    from orb_line_slam_amd import _lib, synth, records
    from orb_line_slam_amd._lib import FrameBuffers, check, lib

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    multi = world > 1 or args.force_dist
    if args.force_dist and world == 1:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path")
    dev_index = local_rank % torch.cuda.device_count()        # more ranks than GPUs only with --backend gloo (ranks then share a device)
    if multi and args.backend == "nccl" and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node (RCCL needs one device per rank; --backend gloo lets ranks share one)")
    torch.cuda.set_device(dev_index)
    dist = None
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # Bring-up under a deadline: a rendezvous that never completes (a rank that died, a wrong MASTER_PORT) or an RCCL that cannot build its rings
        # (two ranks on one device, no peer access between the GPUs, IPC handles refused -- HSA_ENABLE_IPC_MODE_LEGACY=0 must be set on this pool)
        # otherwise hangs until the driver's own clock kills the run; here it ends with a message that says which step failed
        import datetime, signal
        limit = int(os.environ.get("OLF_DIST_TIMEOUT", "240"))
        stage_name = ["rendezvous (init_process_group)"]

        def _deadline(signum, frame):
            raise SystemExit(f"rank {rank}/{world}: distributed bring-up timed out after {limit} s in: {stage_name[0]} "
                             f"(backend {args.backend}, MASTER_ADDR={os.environ.get('MASTER_ADDR')}, MASTER_PORT={os.environ.get('MASTER_PORT')}, device {dev_index} of {torch.cuda.device_count()})")
        signal.signal(signal.SIGALRM, _deadline)
        signal.alarm(limit)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), timeout=datetime.timedelta(seconds=limit))
            else:
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=limit))
            # the first collective builds every ring / channel: if RCCL refuses the topology this is where it says so
            stage_name[0] = "first all_reduce (RCCL communicator / xGMI rings)" if args.backend == "nccl" else "first all_reduce (gloo)"
            probe = torch.ones(1, dtype=torch.int64, device=torch.device("cuda", dev_index) if args.backend == "nccl" else torch.device("cpu"))
            dist.all_reduce(probe)
            if args.backend == "nccl":
                torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError(f"all_reduce of ones gave {int(probe.item())}, expected {world}")
            if args.backend == "nccl":
                # one rank per device: two ranks that landed on the same GPU (a launcher that did not set LOCAL_RANK) deadlock RCCL later, inside the gather
                ids = [None] * world
                dist.all_gather_object(ids, (os.uname().nodename, str(torch.cuda.get_device_properties(dev_index).uuid) if hasattr(torch.cuda.get_device_properties(dev_index), "uuid") else dev_index))
                if len(set(ids)) != world:
                    raise RuntimeError(f"ranks share a device: {ids} (RCCL needs one GPU per rank; use --backend gloo to let ranks share one)")
        except SystemExit:
            raise
        except Exception as e:
            raise SystemExit(f"rank {rank}/{world}: distributed bring-up failed in {stage_name[0]}: {type(e).__name__}: {e}")
        finally:
            signal.alarm(0)
    dev = torch.device("cuda", dev_index)
    # The path runs on torch's current stream, and that must not be the legacy default stream: its handle is NULL, which the C ABI reads as "the context's
    # own stream" (include/orbline.h) -- the harness's events (pack -> gather hand-over) and copies would then not be ordered with the path's work.
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    cdev = dev if (dist is None or args.backend == "nccl") else torch.device("cpu")      # where the scalar reductions of the harness live

    cfg = CONFIGS[args.config]
    W, H = cfg["w"], cfg["h"]
    seq = None
    if args.images:
        from orb_line_slam_amd.sequence import StereoSequence
        seq = StereoSequence(args.images, limit=args.pairs or cfg["pairs"], camera_rgb=bool(args.camera_rgb))
        W, H = seq.width, seq.height
    B = args.pairs or cfg["pairs"]
    if args.scaling == "strong" and world > 1:
        # strong scaling: the configuration's batch is the WHOLE job, split over the ranks (shard_range's contiguous blocks, every rank the same size so that the
        # context capacity is one number); value = that fixed total / max-over-ranks time, so N GPUs can show an efficiency below 1
        B = max(64, (B + world - 1) // world)
    # the batch lives in HBM (context buffers + outputs, 58 MB of context per KITTI pair, tools/mem_per_pair.py, plus the outputs): shrink it if this GPU has less free memory than the
    # batch needs, and use the same size on every rank
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    # (batch contexts, > 1024 pairs: 42.5 MB per KITTI pair, 172 MB per 1080p pair since round 6 -- log sized by a bound, no owner words, work images inside the key
    # buffers; smaller contexts keep the full-size log and the owner words: 55 MB per KITTI pair)
    per_pair = (43e6 if B > 1024 else 62e6) * (W * H) / (1242 * 375) * (1.15 if multi else 1.0)     # + packed records (double buffered) when they are gathered
    fit = int((free_b - 6e9) / per_pair) // 64 * 64
    if fit < B:
        print(f"[rank {rank}] {free_b / 1e9:.0f} GB free: {B} pairs per step do not fit, using {max(fit, 64)}", file=sys.stderr, flush=True)
        B = max(fit, 64)
    if multi:
        tb = torch.tensor([B], dtype=torch.int64, device=cdev)
        dist.all_reduce(tb, op=dist.ReduceOp.MIN)
        B = int(tb.item())
    params = _lib.default_params()
    params.orb.nfeatures, params.line.lsd_nfeatures = cfg["nf"], cfg["nl"]
    params.stereo.fx, params.stereo.bf = cfg["fx"], cfg["bf"]
    if args.seed_order is not None:
        params.line.conv_seed_order = args.seed_order
    ctx = _lib.Context(params, W, H, 2 * B)
    cap, lcap = ctx.orb_capacity, ctx.line_capacity

    # synthetic input: `distinct` seeded pairs per rank, tiled to B pairs, resident in HBM before timing
    nd = min(args.distinct, B)

    def make_input(r, run_len=None, scene=None):
        run_len = args.sequence if run_len is None else run_len
        scene = args.scene if scene is None else scene
        if seq is not None:
            # every rank reads its own contiguous share of the recording (frame-sharded like SURVEY 8(e)), tiled to B pairs when it is shorter
            from orb_line_slam_amd.distributed import shard_range
            lo, hi = shard_range(len(seq), r, world)
            sub_seq = StereoSequence(left=seq.left[lo:hi] or seq.left[:1], right=seq.right[lo:hi] or seq.right[:1], camera_rgb=bool(args.camera_rgb))
            host = np.concatenate([b for b, _ in sub_seq.batches(B)]).reshape(-1, 2, H, W)
            reps = (B + len(host) - 1) // len(host)
            return torch.from_numpy(np.tile(host, (reps, 1, 1, 1))[:B].reshape(2 * B, H, W).copy()).to(dev)
        host = synth.stereo_batch(7000 + 100000 * r, nd, W, H, scene=scene)
        if run_len > 1:
            # runs of consecutive frames: frame k of run j = scene j (both images) shifted by 2k pixels -- related frames for the frame-to-frame matchers
            R = run_len
            out_h = np.empty((2 * B, H, W), np.uint8)
            for i in range(B):
                j, k = (i // R) % nd, i % R
                out_h[2 * i] = np.roll(host[2 * j], 2 * k, axis=1)
                out_h[2 * i + 1] = np.roll(host[2 * j + 1], 2 * k, axis=1)
            return torch.from_numpy(out_h).to(dev)
        # B pairs drawn from the nd distinct ones in a seeded random order (a plain tiling would put identical images at a fixed period,
        # which lines them up with the hardware's round-robin placement of workgroups -- an artefact no real sequence has)
        order = np.random.default_rng(1234 + r).permutation(np.arange(B) % nd) if args.order == "shuffled" else np.arange(B) % nd
        idx = np.stack([2 * order, 2 * order + 1], 1).reshape(-1)
        return torch.from_numpy(host[idx].copy()).to(dev)
    imgs = make_input(rank)
    # the inputs are resident before the timed region starts: their "ready" event lets the line stream of a step start beside the previous step's tail
    in_ev = torch.cuda.Event(); in_ev.record(); torch.cuda.synchronize()
    if not args.no_deferred_join:
        check(lib().olf_ctx_set_deferred_join(ctx.handle, 1), "olf_ctx_set_deferred_join")

    def z(shape, dt):
        return torch.zeros(shape, dtype=dt, device=dev)
    kps = z((2 * B, cap, 28), torch.uint8); desc = z((2 * B, cap, 32), torch.uint8); counts = z((2 * B,), torch.int32)
    ur = z((B, cap), torch.float32); dp = z((B, cap), torch.float32)
    kls = z((2 * B, lcap, 68), torch.uint8); ldesc = z((2 * B, lcap, 32), torch.uint8); lcounts = z((2 * B,), torch.int32)
    lm = z((B, lcap), torch.int32); ldisp = z((B, lcap, 2), torch.float32); lle = z((B, lcap, 3), torch.float64)
    f2f_lines = z((B, 2 * lcap), torch.int32); f2f_orb = z((B, 2 * cap), torch.int32)
    fb = FrameBuffers(*[t.data_ptr() for t in (kps, desc, counts, ur, dp, kls, ldesc, lcounts, lm, ldisp, lle)])
    Lh = lib()
    nnr_l = float(params.stereo.min_ratio_12_l)
    kf_valid = z((B, cap), torch.uint8); f2f_n = z((B,), torch.int32)
    voc = None
    if not args.no_bow:
        # vocabulary of ORBvoc's shape from the descriptors of a few frames of this input (built once, outside the timed region)
        check(Lh.olf_orb_extract_dev(ctx.handle, imgs.data_ptr(), min(2 * B, 16), kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), torch.cuda.current_stream().cuda_stream), "olf_orb_extract_dev")
        torch.cuda.synchronize()
        cn0 = counts[:16].cpu().numpy(); de0 = desc[:16].cpu().numpy()
        sample = np.concatenate([de0[i, :cn0[i]] for i in range(min(2 * B, 16))])
        from orb_line_slam_amd.vocabulary import ORBVocabulary
        voc = ORBVocabulary.from_arrays(10, 6, *synthetic_vocabulary(10, 6, sample), context=ctx)

    def step(images):
        s = torch.cuda.current_stream().cuda_stream
        if args.pipeline:
            ctx.set_input_event(in_ev)      # (one-shot: every call consumes its input event)
        check(Lh.olf_stereo_frames_dev(ctx.handle, images.data_ptr(), B, C.byref(fb), s), "olf_stereo_frames_dev")
        if B > 1 and voc is not None and not args.no_deferred_join:
            # the tracker's order (src/Tracking.cc:963-970, then :1296-1308): the point features' matcher first -- it needs nothing of the line path, whose tail
            # is still running on the context's line stream (olf_ctx_set_deferred_join) -- then the join, then the line matcher
            check(Lh.olf_stereo_points_mask_dev(ctx.handle, dp.data_ptr(), B * cap, kf_valid.data_ptr(), s), "olf_stereo_points_mask_dev")
            check(Lh.olf_search_by_bow_batch_dev(ctx.handle, voc._h, B, 2, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), kf_valid.data_ptr(), None,
                                                 0.7, 1, 4, f2f_orb.data_ptr(), f2f_n.data_ptr(), s), "olf_search_by_bow_batch_dev")
            check(Lh.olf_stereo_frames_join_dev(ctx.handle, s), "olf_stereo_frames_join_dev")
            check(Lh.olf_match_bf_dev(ctx.handle, ldesc.data_ptr() + 2 * lcap * 32, lcounts.data_ptr() + 8, 2 * lcap, 2, ldesc.data_ptr(),
                                      lcounts.data_ptr(), 2 * lcap, 2, B - 1, nnr_l, 1, f2f_lines.data_ptr(), s), "olf_match_bf_dev(lines)")
            return
        check(Lh.olf_stereo_frames_join_dev(ctx.handle, s), "olf_stereo_frames_join_dev")
        if B > 1:
            # frame i (left image 2i) against frame i-1: match(last.mDescriptors_Line, cur.mDescriptors_Line) (src/Tracking.cc:1308)
            check(Lh.olf_match_bf_dev(ctx.handle, ldesc.data_ptr() + 2 * lcap * 32, lcounts.data_ptr() + 8, 2 * lcap, 2, ldesc.data_ptr(),
                                      lcounts.data_ptr(), 2 * lcap, 2, B - 1, nnr_l, 1, f2f_lines.data_ptr(), s), "olf_match_bf_dev(lines)")
            if voc is not None:
                # configuration 3's matcher: cur.ComputeBoW() + ORBmatcher(0.7).SearchByBoW(previous frame as key frame, cur) (src/Tracking.cc:963-970),
                # batched on the device; the key frame's map points = its stereo points (mvDepth > 0, src/Tracking.cc:586-588)
                check(Lh.olf_stereo_points_mask_dev(ctx.handle, dp.data_ptr(), B * cap, kf_valid.data_ptr(), s), "olf_stereo_points_mask_dev")
                check(Lh.olf_search_by_bow_batch_dev(ctx.handle, voc._h, B, 2, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), kf_valid.data_ptr(), None,
                                                     0.7, 1, 4, f2f_orb.data_ptr(), f2f_n.data_ptr(), s), "olf_search_by_bow_batch_dev")
            else:
                # dense ORB kNN(2)+ratio+mutual against the previous frame (--no-bow: the round-1/2 stand-in for SearchByBoW)
                check(Lh.olf_match_bf_dev(ctx.handle, desc.data_ptr() + 2 * cap * 32, counts.data_ptr() + 8, 2 * cap, 2, desc.data_ptr(),
                                          counts.data_ptr(), 2 * cap, 2, B - 1, 0.7, 1, f2f_orb.data_ptr(), s), "olf_match_bf_dev(orb)")

    # ---- the gather of the feature records to rank 0 (N > 1) ---------------------------------------------------------------------
    gather_on = multi and args.gather != "off"
    gstat = {"bytes": 0, "seconds": 0.0, "last": None}
    if gather_on:
        from orb_line_slam_amd.distributed import gather_records, SizeExchange
        szx = [None, None]
        bound = int(Lh.olf_frames_pack_bound(ctx.handle, B))
        # the trimmed record is about half of the bound at the configured feature counts; a record that does not fit is reported by ctx.synchronize()
        pk_bytes = bound if args.verify else bound // 2 + (1 << 20)
        packed = [torch.empty(pk_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        nbytes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(2)]
        ev = [torch.cuda.Event() for _ in range(2)]
        recv = [torch.empty(pk_bytes, dtype=torch.uint8, device=dev) if (rank == 0 and r != 0) else None for r in range(world)]
        comm_stream = torch.cuda.Stream(device=dev, priority=-1) if args.gather == "overlap" else torch.cuda.current_stream()      # (its own priority level: streams of one level can share a hardware queue, profiles/r4aq_stream_queue_aliasing.txt)

        def pack(k, exchange=True):
            s = torch.cuda.current_stream().cuda_stream
            check(Lh.olf_frames_pack_dev(ctx.handle, C.byref(fb), B, packed[k % 2].data_ptr(), pk_bytes, nbytes[k % 2].data_ptr(), s), "olf_frames_pack_dev")
            ev[k % 2].record()
            # the sizes of all ranks' records: asked for now (the device counter goes into the all_gather as it is), read when the record is sent one step later
            if exchange:      # (a collective: every rank or none -- the --verify replay on rank 0 packs without it)
                szx[k % 2] = SizeExchange(nbytes[k % 2], dist, comm_stream)

        def comm(k):
            # ordered after the pack of step k; the main stream is already running step k+1 underneath
            t = time.perf_counter()
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(ev[k % 2])
                recs, sizes = gather_records(packed[k % 2], None, dist, 0, recv if rank == 0 else None, sizes=szx[k % 2].sizes(pk_bytes))
                comm_stream.synchronize()
            gstat["seconds"] += time.perf_counter() - t
            gstat["bytes"] += sum(sizes) - sizes[0]
            gstat["last"] = (recs, sizes)

    def run_steps(n):
        for k in range(n):
            step(imgs)
            if gather_on:
                pack(k)
                if k > 0:
                    comm(k - 1)
        if gather_on and n > 0:
            comm(n - 1)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    barrier()
    gstat["bytes"], gstat["seconds"] = 0, 0.0
    ctx.profile(True)
    t0 = time.perf_counter()
    run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    ctx.synchronize()                        # outside the timed region: raises if any step overflowed a fixed-capacity device buffer
    prof = ctx.profile_read()
    ctx.profile(False)
    rank_dt = None
    if dist is not None:
        # every rank's own time for the K steps (the first real multi-GPU run must be diagnosable from one line: a slow rank, a slow link), then the
        # maximum, which is the job's time
        tl = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(tl, torch.tensor([dt], dtype=torch.float64, device=cdev))
        rank_dt = [float(x.item()) for x in tl]
        tt = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # every stage on its own: the four extraction / matching entries back to back on one stream (what OLF_ONE_STREAM=1 makes of a step), outside the
    # timed region -- in the two-stream step the stages slow each other down, so only these times can be set against a stage's own bytes
    alone = None
    if rank == 0 and not args.no_isolated:
        ctx.profile(True)
        s0 = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            check(Lh.olf_line_extract_dev(ctx.handle, imgs.data_ptr(), 2 * B, kls.data_ptr(), ldesc.data_ptr(), lcounts.data_ptr(), s0), "olf_line_extract_dev")
            check(Lh.olf_stereo_lines_dev(ctx.handle, B, kls.data_ptr(), ldesc.data_ptr(), lcounts.data_ptr(), lm.data_ptr(), ldisp.data_ptr(), lle.data_ptr(), s0), "olf_stereo_lines_dev")
            check(Lh.olf_orb_extract_dev(ctx.handle, imgs.data_ptr(), 2 * B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), s0), "olf_orb_extract_dev")
            check(Lh.olf_stereo_points_dev(ctx.handle, B, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), ur.data_ptr(), dp.data_ptr(), s0), "olf_stereo_points_dev")
        torch.cuda.synchronize()
        alone = {k: v[0] / 2 for k, v in ctx.profile_read().items() if v[1]}
        ctx.profile(False)

    # the companion line: the same step on runs of six consecutive frames of one scene, so that the frame-to-frame matchers (SearchByBoW, LBD match) meet related
    # frames as they do on a recording -- SURVEY 8(d)'s generator (independent scenes) is the headline, this is what it leaves out
    companion, companion_long = None, None
    if rank == 0 and world == 1 and not args.no_extras and seq is None and args.sequence == 0 and B > 1:
        keep = imgs
        imgs = make_input(rank, 6)
        run_steps(1); barrier()
        tq = time.perf_counter()
        run_steps(3); barrier()
        dq = (time.perf_counter() - tq) / 3
        ctx.synchronize()
        companion = {"workload": "runs of 6 consecutive frames per scene (bench.py --sequence 6)", "steps": 3, "ms_per_step": round(dq * 1e3, 3), "value": round(B / dq, 1),
                     "unit": "stereo frames/s"}
        if voc is not None:
            companion["search_by_bow_mean_matches"] = round(float(f2f_n[:B - 1].float().mean().item()), 1)
        # the second companion: long thin bars -> key lines of about 0.08 * W pixels, SURVEY App. D's model of a street scene (the headline's scene makes
        # key lines a third as long); the line-length-sensitive stages -- growth of long regions, rectangle fit, LBD -- with their times in the step
        # 
        if args.scene == "default":
            imgs = make_input(rank, 0, "bars")
            run_steps(1); barrier()
            ctx.profile(True)
            tq = time.perf_counter()
            run_steps(3); barrier()
            dq = (time.perf_counter() - tq) / 3
            ctx.synchronize()
            pq = ctx.profile_read(); ctx.profile(False)
            nq = min(64, 2 * B)
            klq = kls[:nq].cpu().numpy().view(ola.KEYLINE_DTYPE).reshape(nq, lcap); lcq = lcounts[:nq].cpu().numpy()
            companion_long = {"workload": "long thin bars (bench.py --scene bars)", "steps": 3, "ms_per_step": round(dq * 1e3, 3), "value": round(B / dq, 1), "unit": "stereo frames/s",
                              "mean_line_pixels": round(float(np.mean([klq[i, :lcq[i]]["numOfPixels"].mean() for i in range(nq) if lcq[i] > 0])), 1),
                              "survey_appD_line_pixels": round(0.08 * W, 1), "mean_keylines_per_image": round(float(lcounts.float().mean().item()), 1),
                              "stages_ms_per_step": {k: round(v[0] / 3, 3) for k, v in pq.items() if v[1] and k in ("lsd_front", "lsd_grow", "lsd_rect", "line_select_lbd", "stereo_lines")}}
        imgs = keep
        run_steps(1); barrier()             # (the outputs the checks below read belong to the headline input again)

    verify = None
    if gather_on and args.verify:
        # rank 0 runs every other rank's input itself and compares the record it received in the last step, byte for byte
        if rank == 0:
            recs, sizes = gstat["last"]
            verify = {"ranks": world, "identical": True}
            for r in range(1, world):
                other = make_input(r)
                torch.cuda.synchronize()          # (the input event set above speaks for `imgs`: another input has to be complete before the call)
                step(other)
                pack(0, exchange=False)
                torch.cuda.synchronize()
                n = int(nbytes[0].item())
                mine = records.merge_records([packed[0][:n].cpu().numpy().tobytes()])         # normalises the padding between sections
                theirs = records.merge_records([recs[r].cpu().numpy().tobytes()])
                if mine != theirs:
                    verify["identical"] = False
                    verify.setdefault("mismatch", []).append(r)
        dist.barrier()

    out = None
    if rank == 0:
        total_pairs = world * B * args.steps
        fps = total_pairs / dt
        nk = counts.float().mean().item(); nkl = lcounts.float().mean().item()
        kl_np = kls.cpu().numpy().view(ola.KEYLINE_DTYPE).reshape(2 * B, lcap)
        lc_np = lcounts.cpu().numpy()
        mean_len = float(np.mean([kl_np[i, :lc_np[i]]["numOfPixels"].mean() for i in range(min(2 * B, 256)) if lc_np[i] > 0]))
        ab = algorithmic_bytes(W, H, nk, nkl, mean_len)
        stages = {k: {"ms_per_step": v[0] / max(args.steps, 1), "calls": v[1]} for k, v in prof.items() if v[1]}
        # dominant stage: the largest stand-alone time over ALL stages (inside the two-stream step a dense stage that runs in the growth agents'
        # shadow -- FAST -- is stretched to nearly their duration and would pass for the dominant one); its launch duration is the one measured
        # inside the timed region
        dom = max(alone, key=alone.get) if alone else max(stages, key=lambda k: stages[k]["ms_per_step"])
        if dom not in stages: dom = max(stages, key=lambda k: stages[k]["ms_per_step"])
        dom_ms = stages[dom]["ms_per_step"] / max(stages[dom]["calls"] // args.steps, 1)
        per_launch_bytes = ab["stage"].get("orb_fast" if dom == "orb_fast_cells" else dom, 0) * 2 * B
        achieved = per_launch_bytes / (dom_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: 2 x FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes (profiles/*_pmc_hbm_traffic.json, per
        # image; the factor 2 is the calibration of profiles/r3_fetch_calibration.txt: the counter tallies 128-byte read requests at 64 bytes), valid only
        # for the sources it was collected on: the summary carries a hash of csrc/, anything else gives null
        traffic, sh = None, source_hash()
        try:
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")), reverse=True):
                pm = json.load(open(f))
                # (rocprofv3 prints a template kernel as "void olf::name<args>(...)": the summary's keys are that text up to the parenthesis)
                key = next((k for k in pm["kernels"] if k.replace("void ", "").startswith(STAGE_KERNEL[dom])), None)
                if pm.get("source_hash") == sh and key:
                    traffic = int(pm["kernels"][key]["bytes_per_image"] * 2 * B)
                    break
        except Exception:
            traffic = None
        # ... and of every stage: the summary of the same command (one step at this batch) covers every kernel of the step, ORB, stereo and matcher kernels included
        stage_traffic, kernel_traffic = {}, {}
        try:
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")), reverse=True):
                pm = json.load(open(f))
                if pm.get("source_hash") != sh:
                    continue
                for k, v in pm["kernels"].items():
                    kn = k.replace("void ", "")
                    per_step = int(v.get("bytes_per_image_step", v["bytes_per_image"]))      # all launches of a step (FAST: one per level)
                    kernel_traffic[kn] = per_step
                    st_of = next((st for st, pre in STAGE_KERNELS.items() if any(kn.startswith(q) for q in pre)), None)
                    if st_of:
                        stage_traffic[st_of] = stage_traffic.get(st_of, 0) + per_step
                break
        except Exception:
            stage_traffic, kernel_traffic = {}, {}
        # the same kernel against the part's vector-instruction issue rate: SQ_INSTS_VALU per image (profiles/*_valu_budget.json, same source-hash rule)
        # x images per launch / launch duration, over 256 CUs x 4 SIMDs x one wave64 vector instruction per 4 cycles at 2.4 GHz = 614.4 G/s
        valu, valu_step = None, None
        try:
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_valu_budget.json")), reverse=True):
                vb = json.load(open(f))
                key = next((k for k in vb["kernels"] if k.replace("void ", "").startswith(STAGE_KERNEL[dom])), None)
                if vb.get("source_hash") == sh and key:
                    rate = vb["kernels"][key]["valu_per_image"] * 2 * B / (dom_ms * 1e-3) / 1e9
                    valu = {"achieved": round(rate, 1), "peak": 614.4, "unit": "G wave-instructions/s", "frac": round(rate / 614.4, 4),
                            "valu_per_image": int(vb["kernels"][key]["valu_per_image"]), "salu_per_image": int(vb["kernels"][key]["salu_per_image"])}
                    # the whole step against the same ceiling: every kernel's vector instructions per image x images per step / issue rate = the time the
                    # step would take if it did nothing but issue them (scalar instructions of the one-wave agents issue in the same slots, DESIGN 3.9:
                    # the second figure counts them too)
                    tv = sum(k["valu_per_image"] for k in vb["kernels"].values()); ts = sum(k["salu_per_image"] for k in vb["kernels"].values())
                    step_ms = dt / args.steps * 1e3
                    valu_step = {"valu_per_image": int(tv), "salu_per_image": int(ts), "issue_floor_ms": round(tv * 2 * B / 614.4e9 * 1e3, 2),
                                 "frac": round(tv * 2 * B / 614.4e9 * 1e3 / step_ms, 4), "frac_with_scalar": round((tv + ts) * 2 * B / 614.4e9 * 1e3 / step_ms, 4)}
                    break
        except Exception:
            valu = None
        out = {
            "metric": f"stereo frames/s extract+match (ORB+LBD), {'KITTI ' if args.config == 'C3' else ''}{W}x{H}", "value": round(fps, 2), "unit": "stereo frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "u8", "data": ("recorded: " + os.path.basename(os.path.normpath(args.images))) if seq is not None else "synthetic",
            "config": {"workload": f"{args.config}: {W}x{H} stereo, {cfg['nf']} ORB + {cfg['nl']} LBD per image, extract + stereo point/line match + "
                                   f"f2f LBD match + " + ("SearchByBoW vs the previous frame (synthetic k=10 L=6 vocabulary, ComputeBoW included)" if voc is not None else "f2f dense ORB kNN match"), "pairs_per_gpu_per_step": B, "parallelism": f"frame-sharded x{world}",
                       "distinct_pairs": (len(seq) if seq is not None else nd), "order": ("recorded" if seq is not None else f"runs of {args.sequence}" if args.sequence > 1 else args.order), "scene": args.scene, "mean_keypoints_per_image": round(nk, 1), "mean_keylines_per_image": round(nkl, 1),
                       "mean_line_pixels": round(mean_len, 1), "source_hash": sh},
            "roofline": {"bound": "hbm", "kernel": STAGE_KERNEL[dom], "stage": dom, "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(achieved / 8000.0, 6), "traffic": traffic, "algorithmic_bytes_per_launch": int(per_launch_bytes),
                         "avg_launch_ms": round(dom_ms, 4), "path_bytes_per_pair": int(ab["pair"]),
                         "path_frac_of_hbm_peak": round(ab["pair"] * fps / world / 8e12, 6), "valu_issue": valu, "valu_issue_step": valu_step},
            "stages_ms_per_step": {k: round(v["ms_per_step"], 3) for k, v in stages.items()},
        }
        if rank_dt is not None:
            ms = [t / args.steps * 1e3 for t in rank_dt]
            out["per_rank_ms_per_step"] = {"min": round(min(ms), 3), "max": round(max(ms), 3), "slowest_rank": int(np.argmax(ms)), "ranks": [round(m, 3) for m in ms]}
        if alone:
            # per stage: algorithmic bytes of a launch over the whole batch / the stage's time when it runs alone -> GB/s and fraction of the 8 TB/s
            # peak (stages without a byte model -- octree, matchers, region2rect -- carry their time only)
            rs = {}
            for k, ms in alone.items():
                by = ab["stage"].get("orb_fast" if k == "orb_fast_cells" else k, 0) * 2 * B
                rs[k] = {"ms_alone": round(ms, 3)}
                if by:
                    rs[k].update(GBps=round(by / (ms * 1e-3) / 1e9, 1), frac=round(by / (ms * 1e-3) / 1e9 / 8000.0, 4))
                if k in stage_traffic:
                    # counter traffic (2 x FETCH_SIZE + WRITE_SIZE per launch of the whole batch) next to the algorithmic bytes: the ratio is the re-read factor
                    rs[k]["traffic"] = stage_traffic[k] * 2 * B
                    if by:
                        rs[k]["traffic_over_algorithmic"] = round(stage_traffic[k] * 2 * B / by, 2)
            if "match" in stage_traffic:
                rs["match"] = {"traffic": stage_traffic["match"] * 2 * B}
            out["roofline"]["stages"] = rs
            if kernel_traffic:
                out["roofline"]["kernel_traffic_bytes_per_image"] = kernel_traffic
            out["roofline"]["sum_alone_ms"] = round(sum(alone.values()), 2)
        if gather_on:
            out["gather"] = {"mode": args.gather, "bytes_per_step": int(gstat["bytes"] / max(args.steps, 1)),
                             "GBps": round(gstat["bytes"] / max(gstat["seconds"], 1e-9) / 1e9, 3),
                             "rank0_ingest_GBps": round(gstat["bytes"] / dt / 1e9, 3),       # bytes rank 0 received per second of the timed region
                             "host_seconds_per_step": round(gstat["seconds"] / max(args.steps, 1), 5), "verify": verify}
        if not args.no_extras:
            # measured ceiling of a plain copy kernel, next to the 8 TB/s of the specification
            try:
                g = C.c_double()
                check(Lh.olf_debug_copy_bandwidth(ctx.handle, 1 << 30, 20, C.byref(g)), "olf_debug_copy_bandwidth")
                out["roofline"]["copy_kernel_GBps"] = round(g.value, 1)
                out["roofline"]["frac_of_copy_kernel"] = round(achieved / g.value, 6)
            except Exception as e:
                out["roofline"]["copy_kernel_GBps"] = None
                print(f"copy ceiling failed: {e}", file=sys.stderr)
    if rank == 0 and out is not None and companion is not None:
        out["companion_sequence6"] = companion
    if rank == 0 and out is not None and companion_long is not None:
        out["companion_long_lines"] = companion_long
    if rank == 0 and out is not None and voc is not None:
        out["config"]["search_by_bow_mean_matches"] = round(float(f2f_n[:B - 1].float().mean().item()), 1)
    del imgs, kps, desc, ur, dp, kls, ldesc, lm, ldisp, lle, f2f_lines, f2f_orb
    if voc is not None:
        voc.clear()
    ctx.close()
    torch.cuda.empty_cache()
    if rank == 0:
        if not args.no_extras:
            try:
                out["pair_latency_ms"] = small_batch_latency(params, W, H)
            except Exception as e:
                out["pair_latency_ms"] = {"error": str(e)}
            try:
                from orb_line_slam_amd.pipeline import pcie_inclusive_rate
                # the second headline: host images in, host results out, at the full batch, steady state over 10 batches (a reader that decodes into the
                # pinned staging buffer); and the convenience shape that copies pageable arrays into it first
                out["pcie_inclusive"] = pcie_inclusive_rate(params, W, H, pairs=B, batches=10, producer="pinned")
                out["pcie_inclusive"]["frac_of_value"] = round(out["pcie_inclusive"]["value"] / out["value"], 4)
                out["pcie_inclusive"]["pageable_input"] = pcie_inclusive_rate(params, W, H, pairs=B, batches=4, producer="pageable")["value"]
            except Exception as e:
                out["pcie_inclusive"] = {"error": str(e)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cb = cpu_baseline(W, H, params, args.cpu_seconds)
            # the GPU figure against both CPU shapes, side by side: Mode A = the reference's own (one frame at a time, 4 threads), Mode B = the box's
            # cores all busy with whole frames; and the drop-in's online shape (one pair per call) against Mode A's frame time
            cb["gpu_over_cpu"] = {"batched_vs_mode_a": round(out["value"] / cb["value"], 1),
                                  "batched_vs_mode_b": round(out["value"] / cb["mode_b"]["value"], 1) if cb.get("mode_b", {}).get("value") else None,
                                  "one_pair_vs_mode_a": round(1e3 / cb["value"] / out["pair_latency_ms"]["1"], 2) if isinstance(out.get("pair_latency_ms", {}).get("1"), float) else None}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
