"""Diagnostic for tests/test_line_gpu.py::test_pipelined_batches_with_input_event: which outputs of which call / image differ between sequential and pipelined calls."""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
from orb_line_slam_amd import synth, _lib
from orb_line_slam_amd._lib import FrameBuffers, check as chk, lib
import oracle_lib as oracle
w, h, n = 1242, 375, int(sys.argv[1]) if len(sys.argv) > 1 else 192
p = oracle.full_params(2000, 500, 718.856, 386.1448)
dev = torch.device("cuda", 0)
inputs = [torch.from_numpy(np.tile(synth.stereo_batch(7400 + 10 * k, 8, w, h), (n // 8, 1, 1))).to(dev) for k in range(2)]
ctx = _lib.Context(p, w, h, 2 * n)
cap, lcap = ctx.orb_capacity, ctx.line_capacity
names = ["kps", "desc", "counts", "ur", "dp", "kls", "ldesc", "lcounts", "lm", "ldisp", "lle"]
spec = [((2 * n, cap, 28), torch.uint8), ((2 * n, cap, 32), torch.uint8), ((2 * n,), torch.int32), ((n, cap), torch.float32), ((n, cap), torch.float32),
        ((2 * n, lcap, 68), torch.uint8), ((2 * n, lcap, 32), torch.uint8), ((2 * n,), torch.int32), ((n, lcap), torch.int32), ((n, lcap, 2), torch.float32),
        ((n, lcap, 3), torch.float64)]
out = [torch.zeros(sh, dtype=dt, device=dev) for sh, dt in spec]
fb = FrameBuffers(*[t.data_ptr() for t in out])
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st); s = st.cuda_stream
order = [0, 1, 0, 1]
def run(pipelined):
    got = []
    for t in out: t.zero_()
    ev = torch.cuda.Event(); ev.record(); torch.cuda.synchronize()
    ctx.set_input_event(ev if pipelined else None)
    for k in order:
        chk(lib().olf_stereo_frames_dev(ctx.handle, inputs[k].data_ptr(), n, C.byref(fb), s), "olf_stereo_frames_dev")
        got.append([t.clone() for t in out])
        if not pipelined: torch.cuda.synchronize()
    torch.cuda.synchronize(); ctx.synchronize(); ctx.set_input_event(None)
    return [[t.cpu().numpy() for t in g] for g in got]
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    ref, pip = run(False), run(True)
    for i in range(len(order)):
        for nm, a, b in zip(names, ref[i], pip[i]):
            if a.tobytes() != b.tobytes():
                rows = [r for r in range(a.shape[0]) if a[r].tobytes() != b[r].tobytes()]
                eq = [k for k in range(len(order)) if ref[k][names.index(nm)].tobytes() == b.tobytes()]
                print("rep", rep, "call", i, nm, "differing rows", len(rows), "of", a.shape[0], "first", rows[:6], "| equals sequential call(s)", eq, "| all zero", not b.any(), flush=True)
    ref2 = run(False)
    same = all(a.tobytes() == b.tobytes() for g1, g2 in zip(ref, ref2) for a, b in zip(g1, g2))
    print("rep", rep, "sequential run repeatable:", same, flush=True)
print("done")
