"""The N > 1 path of bench.py on ONE GPU: two ranks (torch.distributed.run, --backend gloo) share cuda:0, each runs its shard of the stereo
pairs through the HIP path, packs the trimmed feature records on the device (olf_frames_pack_dev), gathers them to rank 0 inside the timed
region (orb_line_slam_amd/distributed.py gather_records, staged through host memory because gloo moves host buffers) and --verify makes
rank 0 recompute the other rank's records from the same seeds and byte-compare them.  On an 8-GPU node the same code runs with
--backend nccl (RCCL over xGMI) on the device buffers; this test removes the 'never executed' risk from everything but the transport."""
import json
import os
import socket
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("gather", ["overlap", "sync", "overlap-strong"])
def test_two_ranks_on_one_gpu_gather_and_verify(gather):
    # ("overlap-strong": --scaling strong -- the configuration's batch is the whole job, split over the ranks)
    strong = gather.endswith("-strong")
    gather = gather.split("-")[0]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--pairs", "64", "--distinct", "64", "--steps", "3", "--warmup", "1",
           "--gather", gather, "--verify", "--no-cpu-baseline", "--no-extras"] + (["--scaling", "strong", "--pairs", "128"] if strong else [])
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]            # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == ("strong" if strong else "weak") and d["value"] > 0
    g = d["gather"]
    assert g["verify"] == {"ranks": 2, "identical": True}, g
    assert g["bytes_per_step"] > 64 * 50_000                      # rank 1's trimmed records really travelled (about 0.2 MB per pair)
    # the whole-job value counts both ranks' pairs (strong: 128 pairs split into 64 + 64)
    assert d["config"]["pairs_per_gpu_per_step"] == 64
    assert abs(d["value"] - 2 * 64 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-3
    # one line must be enough to diagnose the first real multi-GPU run: every rank's own step time and what rank 0 took in
    pr = d["per_rank_ms_per_step"]
    assert len(pr["ranks"]) == 2 and pr["min"] <= pr["max"] and abs(pr["max"] - d["ms_per_step"]) / d["ms_per_step"] < 1e-3 and pr["slowest_rank"] in (0, 1)
    assert g["rank0_ingest_GBps"] > 0


def test_distributed_bring_up_fails_fast_with_a_message():
    """bench.py --gpus N must not hang when the job cannot come up.  (a) RCCL with more ranks than GPUs on the node: refused before any collective.
    (b) a rank whose peers never arrive: the rendezvous deadline (OLF_DIST_TIMEOUT) ends it with the step it was stuck in."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="1", LOCAL_RANK="1", WORLD_SIZE="2")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env["MASTER_PORT"] = str(port)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pairs", "64", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"]
    import torch
    if torch.cuda.device_count() < 2:
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
        assert out.returncode != 0 and b"no GPU 1 on this node" in out.stderr, out.stderr.decode()[-1500:]
    env.update(RANK="0", LOCAL_RANK="0", OLF_DIST_TIMEOUT="15")
    import time
    t = time.time()
    out = subprocess.run(cmd + ["--backend", "gloo"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and (b"timed out" in out.stderr or b"bring-up failed" in out.stderr), out.stderr.decode()[-1500:]
    assert b"rendezvous" in out.stderr and time.time() - t < 200


def test_rccl_backend_single_rank():
    """The RCCL backend itself on a 1-GPU box: bench.py --force-dist runs the N > 1 code path with one rank -- init_process_group("nccl", device_id=...),
    the probe all_reduce, the one-device-per-rank check (all_gather_object), the batch-size agreement, pack, the size exchange (all_gather of device
    tensors), the barrier and the per-rank / MAX timing reductions all execute in librccl; only the point-to-point transfer has no peer to go to."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="VERSION")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--backend", "nccl", "--pairs", "64", "--distinct", "64", "--steps", "3", "--warmup", "1",
           "--verify", "--no-cpu-baseline", "--no-extras", "--no-isolated"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["gather"]["verify"] == {"ranks": 1, "identical": True} and d["gather"]["bytes_per_step"] == 0
    assert len(d["per_rank_ms_per_step"]["ranks"]) == 1


def test_rccl_point_to_point_on_device_buffers_self_loop():
    """gather_records' transfer primitive -- batch_isend_irecv of device uint8 tensors on a side stream -- through RCCL with the only peer a 1-GPU box has:
    the rank itself (tools/rccl_self_p2p_probe.py).  Group semantics, stream ordering and the work handles run as on a node; the xGMI link does not."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_self_p2p_probe.py"), str(8 << 20)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    d = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith("{")][0])
    assert d["identical"] is True and d["bytes"] == 8 << 20 and len(d["seconds"]) == 4



def test_cpp_driver_links_rccl_directly_single_rank():
    """examples/stereo_batch_sharded.cpp: the batched mode behind the C ABI for a C++ caller -- one process per GPU, shard_range, olf_stereo_frames_dev,
    olf_frames_pack_dev, ncclAllGather of the record sizes, grouped ncclSend / ncclRecv of the records one step late.  With the one rank a 1-GPU box has: the communicator,
    the all-gather and the pipelined loop run in librccl; the line it prints must account for every pair and a non-empty record per step."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "stereo_batch_sharded")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "examples")], check=True)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([exe, "16", "3", "1242", "375"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["ranks"] == 1 and line["pairs_per_step"] == 16 and line["steps"] == 3 and line["stereo_frames_per_s"] > 0
    assert line["record_bytes_all_ranks"] > 3 * 16 * 50000          # (about 0.2 MB per KITTI pair, trimmed)


def test_cpp_driver_verify_single_rank():
    """examples/stereo_batch_sharded --verify (SURVEY 8(e) "Verification"): after the last step rank 0 recomputes every rank's shard from the seeds
    and memcmp's the records.  Four steps, so that the double-buffer hand-over (the pack of step k + 2 waits for step k's record to leave its buffer) runs."""
    import json, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "stereo_batch_sharded")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "examples")], check=True)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([exe, "12", "4", "640", "480", "--verify"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["verify"] == {"ranks": 1, "identical": True}, line
    assert line["pairs_per_step"] == 12 and line["steps"] == 4 and line["record_bytes_all_ranks"] > 0


_WORKER8 = r'''
import os, sys, ctypes as C
root, n_frames = sys.argv[1], int(sys.argv[2])
sys.path.insert(0, root)
import numpy as np, torch, torch.distributed as dist
from orb_line_slam_amd import _lib, synth, records
from orb_line_slam_amd.distributed import shard_range, gather_records
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)                                      # every rank on the one GPU of the box
dev = torch.device("cuda", 0)
W, H = 640, 480
p = _lib.default_params(); p.orb.nfeatures, p.line.lsd_nfeatures = 1000, 200
L = _lib.lib()

def record_of(lo, hi, capacity):
    """the trimmed record of job frames [lo, hi) (frame f has seed 300 + f) from a context of `capacity` pairs"""
    n = hi - lo
    ctx = _lib.Context(p, W, H, 2 * max(capacity, 1))
    cap, lcap = ctx.orb_capacity, ctx.line_capacity
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    m = max(n, 1)
    t = [z((2 * m, cap, 28), torch.uint8), z((2 * m, cap, 32), torch.uint8), z((2 * m,), torch.int32), z((m, cap), torch.float32), z((m, cap), torch.float32),
         z((2 * m, lcap, 68), torch.uint8), z((2 * m, lcap, 32), torch.uint8), z((2 * m,), torch.int32), z((m, lcap), torch.int32), z((m, lcap, 2), torch.float32),
         z((m, lcap, 3), torch.float64)]
    fb = _lib.FrameBuffers(*[x.data_ptr() for x in t])
    s = torch.cuda.current_stream().cuda_stream
    bound = L.olf_frames_pack_bound(ctx.handle, m)
    dst = torch.zeros(bound, dtype=torch.uint8, device=dev)
    nbytes = torch.zeros(1, dtype=torch.int64, device=dev)
    if n > 0:
        imgs = torch.from_numpy(np.concatenate([synth.stereo_batch(300 + f, 1, W, H) for f in range(lo, hi)])).to(dev)
        _lib.check(L.olf_stereo_frames_dev(ctx.handle, imgs.data_ptr(), n, C.byref(fb), s), "olf_stereo_frames_dev")
    _lib.check(L.olf_frames_pack_dev(ctx.handle, C.byref(fb), n, dst.data_ptr(), bound, nbytes.data_ptr(), s), "olf_frames_pack_dev")
    torch.cuda.synchronize(); ctx.synchronize()
    out = dst[:int(nbytes.item())].clone()
    ctx.close()
    return out

lo, hi = shard_range(n_frames, rank, world)
mine = record_of(lo, hi, hi - lo)
recs, sizes = gather_records(mine, mine.numel(), dist, 0)
if rank == 0:
    assert len(recs) == world and [int(s) for s in sizes] == [r.numel() for r in recs]
    shards = [shard_range(n_frames, r, world) for r in range(world)]
    assert len({b - a for a, b in shards}) > 1, "the job must give uneven shards"
    merged = records.merge_records([r.cpu().numpy().tobytes() for r in recs])
    whole = records.merge_records([record_of(0, n_frames, n_frames).cpu().numpy().tobytes()])
    assert merged == whole, "the gathered records differ from the one-rank run"
    pr = records.parse_records(merged)
    assert pr["n_pairs"] == n_frames and pr["counts"].min() > 0
    print("GATHER8_OK", n_frames, [b - a for a, b in shards], len(merged))
dist.barrier(); dist.destroy_process_group()
'''


def test_eight_ranks_share_the_gpu_uneven_shards(tmp_path):
    """The driver's 8-rank shape on the one GPU of the box (gloo; the records are staged through host memory): 27 stereo pairs frame-sharded 4,4,4,3,3,3,3,3 by
    shard_range, every rank runs the HIP path on its shard and packs its record on the device, rank 0 gathers point to point; the merge of the eight records must be
    byte-identical to the record of the same job on one rank (the CPU suite has the same shape with fixture rows)."""
    script = tmp_path / "worker8.py"
    script.write_text(_WORKER8)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, "27"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[-1500:] for o in outs]
    assert "GATHER8_OK 27 [4, 4, 4, 3, 3, 3, 3, 3]" in outs[0], outs[0][-1500:]
