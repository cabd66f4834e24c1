"""RCCL point-to-point on device buffers with ONE rank: the record transfer of orb_line_slam_amd/distributed.py::gather_records is
batch_isend_irecv(P2POp(isend / irecv, device uint8 tensor, peer)).  A 1-GPU box has no peer, but RCCL accepts a grouped send + receive whose peer is the
rank itself, which runs the same torch / RCCL code (group start / end, the P2P kernel on the communicator's stream, the work handles' wait) on the same kind of
buffers; what it cannot show is the xGMI link.  Prints one JSON line.  Usage: python tools/rccl_self_p2p_probe.py [bytes]"""
import json
import os
import sys
import time
import datetime
import torch
import torch.distributed as dist

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64 << 20
for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29541"), ("RANK", "0"), ("WORLD_SIZE", "1")):
    os.environ.setdefault(k, v)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=60))
src = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev)
dst = torch.zeros(n, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream(dev, priority=-1)
ev = torch.cuda.Event(); ev.record()
times = []
for it in range(4):
    dst.zero_(); ev.record()
    t = time.perf_counter()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, dst[:n - it], 0), dist.P2POp(dist.isend, src[:n - it], 0)]):
            w.wait()
        side.synchronize()
    times.append(time.perf_counter() - t)
    ok = bool(torch.equal(dst[:n - it], src[:n - it])) and (it == 0 or int(dst[n - it:].sum().item()) == 0)
    if not ok:
        break
print(json.dumps({"backend": "nccl (RCCL)", "world": 1, "bytes": n, "identical": ok, "seconds": [round(x, 5) for x in times],
                  "GBps_last": round(n / times[-1] / 1e9, 2)}), flush=True)
dist.destroy_process_group()
