#!/usr/bin/env python3
"""bench.py -- stereo frames/s for extract + match (ORB + LBD) on KITTI-size synthetic stereo, 1..8 GPUs.

One "step" = one pass of the whole hot path over one batch of B stereo pairs per GPU, inputs already
resident in HBM: olf_stereo_frames_dev (ExtractORB x2, ExtractLine x2, ComputeStereoMatches,
ComputeStereoMatches_Lines) + the frame-to-frame LBD match (match(), src/LineMatcher.cpp:104-132) and
the frame-to-frame dense ORB kNN match against the previous frame of the batch.
Workload = BASELINE.json configs[2]: 1242x375, 2000 ORB + 500 LBD (the config the metric is quoted on).

Multi-GPU: frames are independent -> every rank processes its own B pairs (weak scaling,
no data-path collective); rank 0 prints ONE JSON line with the whole-job aggregate.  The gather of the
per-rank feature records to rank 0 is exercised once after the timed region (it is not part of a step).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29501 \
        bench.py --gpus 8 --steps 5 --warmup 2
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def algorithmic_bytes(W, H, N, NL, mean_len, nlevels=8, sf=1.2):
    """SURVEY.md Appendix D, re-evaluated for this build's layouts (DESIGN.md "byte model"):
    LSD keeps the integer gradient pair (4 B/px) instead of fp32 modgrad+angle (8 B/px); the pseudo-sort
    reads it once and writes 4-byte keys for the defined pixels (budgeted for all pixels); region growing
    reads grad (4) + r/w used (2)."""
    lv = []
    s = 1.0
    for _ in range(nlevels):
        lv.append((int(round(W / s)), int(round(H / s))))
        s *= sf
    P = [w * h for w, h in lv]
    Pp, P0, P7 = sum(P), P[0], P[-1]
    Ws, Hs = int(round(W * 1.2)), int(round(H * 1.2))
    Ps = Ws * Hs
    orb = P0 + (Pp - P0) + (Pp - P7) + Pp + 2 * Pp + 749 * N + (512 + 32 + 28) * N
    lsd = 2 * P0 + (P0 + Ps) + (Ps + 4 * Ps) + (4 * Ps + 4 * Ps) + (4 * Ps + 2 * Ps)
    lbd = 2 * P0 + (P0 + 4 * P0) + 63 * mean_len * 4 * NL + (32 + 68) * NL
    grow = (4 + 2) * Ps
    return dict(orb=orb, lsd=lsd, lbd=lbd, pair=2 * (orb + lsd + lbd), grow_per_image=grow, Ps=Ps, Pp=Pp)


def cpu_baseline(W, H, params, seconds_budget=20.0):
    """The CPU oracle (oracle/liboracle_fast.so, -O3 -march=native -ffp-contract=off) timed on this box's host
    cores in the reference's shape: 4 std::threads per frame (src/Frame.cc:164-171), frames one at a time."""
    import subprocess
    subprocess.run(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "liboracle_fast.so"], check=True)   # -march=native: always rebuilt on the box that runs it
    L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_fast.so"))
    from orb_line_slam_amd import synth
    times = []
    t_all = time.time()
    seed = 1000
    while True:
        l, r = synth.stereo_pair(seed, W, H)
        seed += 1
        n = [C.c_int() for _ in range(4)]
        t = time.perf_counter()
        rc = L.orc_stereo_frame(l.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p), W, H, C.byref(params), 4,
                                None, None, C.byref(n[0]), None, None, C.byref(n[1]), 1 << 20, None, None,
                                None, None, C.byref(n[2]), None, None, C.byref(n[3]), 1 << 20, None, None, None)
        times.append(time.perf_counter() - t)
        assert rc == 0
        if (time.time() - t_all > seconds_budget and len(times) >= 8) or len(times) >= 400:
            break
    times = np.array(times[2:])   # drop warm-up
    med = float(np.median(times))
    return {"value": round(1.0 / med, 3), "unit": "stereo frames/s", "cores": 4, "kind": "port",
            "sample": f"{len(times)} synthetic {W}x{H} stereo pairs, one at a time, 4 threads/frame like src/Frame.cc:164-171, "
                      f"median {med * 1e3:.1f} ms/frame (mean {times.mean() * 1e3:.1f})", "host_cores": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=3072, help="stereo pairs per GPU per step")
    ap.add_argument("--width", type=int, default=1242)
    ap.add_argument("--height", type=int, default=375)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--lines", type=int, default=500)
    ap.add_argument("--distinct", type=int, default=32, help="distinct synthetic pairs (tiled to --pairs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    import torch
    import orb_line_slam_amd as ola
    from orb_line_slam_amd import _lib, synth
    from orb_line_slam_amd._lib import FrameBuffers, check, lib

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU path")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    W, H, B = args.width, args.height, args.pairs
    # the batch lives in HBM (context buffers + outputs, about 62 MB per KITTI pair): shrink it if this GPU has less free memory than the
    # default batch needs, and use the same size on every rank
    free_b, _total_b = torch.cuda.mem_get_info(dev)
    per_pair = 62e6 * (W * H) / (1242 * 375)
    fit = int((free_b - 6e9) / per_pair) // 256 * 256
    if fit < B:
        print(f"[rank {rank}] {free_b / 1e9:.0f} GB free: {B} pairs per step do not fit, using {max(fit, 256)}", file=sys.stderr, flush=True)
        B = max(fit, 256)
    if world > 1:
        tb = torch.tensor([B], dtype=torch.int64, device=dev)
        dist.all_reduce(tb, op=dist.ReduceOp.MIN)
        B = int(tb.item())
    params = _lib.default_params()
    params.orb.nfeatures, params.line.lsd_nfeatures = args.features, args.lines
    ctx = _lib.Context(params, W, H, 2 * B)
    cap, lcap = ctx.orb_capacity, ctx.line_capacity

    # synthetic input: `distinct` seeded pairs per rank, tiled to B pairs, resident in HBM before timing
    nd = min(args.distinct, B)
    host = synth.stereo_batch(7000 + 1000 * rank, nd, W, H)
    reps = (B + nd - 1) // nd
    imgs = torch.from_numpy(np.tile(host, (reps, 1, 1))[:2 * B].copy()).to(dev)

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

    def step():
        s = torch.cuda.current_stream().cuda_stream
        check(Lh.olf_stereo_frames_dev(ctx.handle, imgs.data_ptr(), B, C.byref(fb), s), "olf_stereo_frames_dev")
        if B > 1:
            # frame i (left image 2i) against frame i-1: match(last.mDescriptors_Line, cur.mDescriptors_Line) (src/Tracking.cc:1308)
            check(Lh.olf_match_bf_dev(ctx.handle, ldesc.data_ptr() + 2 * lcap * 32, lcounts.data_ptr() + 8, 2 * lcap, 2, ldesc.data_ptr(),
                                      lcounts.data_ptr(), 2 * lcap, 2, B - 1, nnr_l, 1, f2f_lines.data_ptr(), s), "olf_match_bf_dev(lines)")
            # dense ORB kNN(2)+ratio+mutual against the previous frame (BF stand-in for SearchByBoW without a vocabulary)
            check(Lh.olf_match_bf_dev(ctx.handle, desc.data_ptr() + 2 * cap * 32, counts.data_ptr() + 8, 2 * cap, 2, desc.data_ptr(),
                                      counts.data_ptr(), 2 * cap, 2, B - 1, 0.7, 1, f2f_orb.data_ptr(), s), "olf_match_bf_dev(orb)")

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ctx.synchronize()                        # outside the timed region: raises if any step overflowed a fixed-capacity device buffer
    prof = ctx.profile_read()
    ctx.profile(False)
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # the only collective of the design: gather the (trimmed) feature records to rank 0, outside the timed region
        try:
            from orb_line_slam_amd.distributed import gather_counts
            gather_counts(counts, lcounts, dist)
        except Exception as e:   # the gather is not part of the measurement: report, do not lose the bench line
            print(f"[rank {rank}] gather_to_rank0 failed: {e}", file=sys.stderr, flush=True)

    if rank == 0:
        total_pairs = world * B * args.steps
        fps = total_pairs / dt
        nk = counts.float().mean().item(); nkl = lcounts.float().mean().item()
        kl_np = kls.cpu().numpy().view(ola.KEYLINE_DTYPE).reshape(2 * B, lcap)
        lc_np = lcounts.cpu().numpy()
        mean_len = float(np.mean([kl_np[i, :lc_np[i]]["numOfPixels"].mean() for i in range(min(2 * B, 64)) if lc_np[i] > 0]))
        ab = algorithmic_bytes(W, H, nk, nkl, mean_len)
        stages = {k: {"ms_per_step": v[0] / max(args.steps, 1), "calls": v[1]} for k, v in prof.items() if v[1]}
        dom = max(("lsd_grow", "orb_octree", "orb_describe"), key=lambda k: stages.get(k, {"ms_per_step": 0})["ms_per_step"])
        dom_ms = stages[dom]["ms_per_step"] / max(stages[dom]["calls"] // args.steps, 1)
        per_launch_bytes = {"lsd_grow": ab["grow_per_image"] * 2 * B,
                            "orb_octree": 0, "orb_describe": (749 + 512 + 32 + 28) * nk * 2 * B}[dom]
        achieved = per_launch_bytes / (dom_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes (profiles/*_pmc_hbm_traffic.json,
        # KiB counters, per image), scaled to this launch's image count; null when no PMC summary has been committed for the kernel.
        traffic = None
        try:
            import glob
            pm = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm_traffic.json")))[-1]))    # latest committed PMC summary
            kname = {"lsd_grow": "olf::k_lsd_grow", "orb_octree": "olf::k_octree", "orb_describe": "olf::k_describe"}[dom]
            if kname in pm["kernels"]:
                traffic = int(pm["kernels"][kname]["bytes_per_image"] * 2 * B)
        except Exception:
            traffic = None
        out = {
            "metric": "stereo frames/s extract+match (ORB+LBD), KITTI 1242x375", "value": round(fps, 2), "unit": "stereo frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W}x{H} stereo, {args.features} ORB + {args.lines} LBD per image, extract + stereo point/line match + "
                                   f"f2f LBD match + f2f dense ORB kNN match", "pairs_per_gpu_per_step": B, "parallelism": f"frame-sharded x{world}",
                       "mean_keypoints_per_image": round(nk, 1), "mean_keylines_per_image": round(nkl, 1), "mean_line_pixels": round(mean_len, 1)},
            "roofline": {"bound": "hbm", "kernel": {"lsd_grow": "olf::k_lsd_grow", "orb_octree": "olf::k_octree", "orb_describe": "olf::k_describe"}[dom],
                         "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 6), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(per_launch_bytes), "avg_launch_ms": round(dom_ms, 4),
                         "path_bytes_per_pair": int(ab["pair"]), "path_frac_of_hbm_peak": round(ab["pair"] * fps / world / 8e12, 6)},
            "stages_ms_per_step": {k: round(v["ms_per_step"], 3) for k, v in stages.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(W, H, params, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
