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


@pytest.mark.parametrize("gather", ["overlap", "sync"])
def test_two_ranks_on_one_gpu_gather_and_verify(gather):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--pairs", "64", "--distinct", "64", "--steps", "3", "--warmup", "1",
           "--gather", gather, "--verify", "--no-cpu-baseline", "--no-extras"]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()[-2000:]            # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    g = d["gather"]
    assert g["verify"] == {"ranks": 2, "identical": True}, g
    assert g["bytes_per_step"] > 64 * 50_000                      # rank 1's trimmed records really travelled (about 0.2 MB per pair)
    # the whole-job value counts both ranks' pairs
    assert abs(d["value"] - 2 * 64 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-3
