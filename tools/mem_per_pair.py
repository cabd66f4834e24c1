import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
os.environ["OLF_ALLOC_TRACE"] = "1"
W, H, NF, NL, N = (int(v) for v in (sys.argv[1:6] + ["1242", "375", "2000", "500", "1024"][len(sys.argv) - 1:]))
from orb_line_slam_amd import _lib
torch.cuda.init()
f0, tot = torch.cuda.mem_get_info()
p = _lib.default_params(); p.orb.nfeatures, p.line.lsd_nfeatures = NF, NL
ctx = _lib.Context(p, W, H, 2 * N)
f1, _ = torch.cuda.mem_get_info()
print("total GB", tot / 1e9, "free before", f0 / 1e9, f"context for {N} pairs of {W}x{H}: GB", (f0 - f1) / 1e9, "-> MB per pair", (f0 - f1) / N / 1e6)
