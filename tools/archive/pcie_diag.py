"""The PCIe-inclusive leg of bench.py on its own: python tools/pcie_diag.py [own_stream 0/1] (OLF_PIPE_EVENT=1: input event per batch)"""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from orb_line_slam_amd import _lib
from orb_line_slam_amd.pipeline import pcie_inclusive_rate
if len(sys.argv) > 1 and sys.argv[1] == "1":
    torch.cuda.set_stream(torch.cuda.Stream())
from orb_line_slam_amd.pipeline import _gpu_local_cpus
lc = _gpu_local_cpus(torch, torch.device('cuda', 0))
print('gpu-local cpus:', (min(lc), max(lc), len(lc)) if lc else None, 'affinity:', len(os.sched_getaffinity(0)), flush=True)
p = _lib.default_params()
for prod in ("pinned", "pageable", "pinned"):
    r = pcie_inclusive_rate(p, 1242, 375, pairs=3072, batches=8, producer=prod)
    print(sys.argv[1:], os.environ.get("OLF_PIPE_EVENT", "0"), prod, r["value"], r["batch_interval_ms"], r["whole_run"], flush=True)
