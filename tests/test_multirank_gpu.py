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
