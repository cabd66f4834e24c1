import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from orb_line_slam_amd import _lib
torch.cuda.init()
f0, tot = torch.cuda.mem_get_info()
p = _lib.default_params(); p.orb.nfeatures, p.line.lsd_nfeatures = 2000, 500
ctx = _lib.Context(p, 1242, 375, 2 * 1024)
f1, _ = torch.cuda.mem_get_info()
print("total GB", tot / 1e9, "free before", f0 / 1e9, "context for 1024 pairs: GB", (f0 - f1) / 1e9, "-> MB per pair", (f0 - f1) / 1024 / 1e6)
