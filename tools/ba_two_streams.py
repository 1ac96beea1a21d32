"""Linearisation steps alternating between two problems on two streams against one problem on one
stream (python3 tools/ba_two_streams.py [workload])."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_problem
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
probs = []
for s in streams:
    sc, ba = make_problem(name, 0, 1)
    ba.set_stream(s.cuda_stream)
    dc = torch.full((6 * sc.Nc,), 1e-4, dtype=torch.float64, device="cuda")
    dp = torch.full((3 * sc.Np,), 1e-4, dtype=torch.float64, device="cuda")
    pc = C.cast(C.c_void_p(dc.data_ptr()), C.POINTER(C.c_double))
    pp = C.cast(C.c_void_p(dp.data_ptr()), C.POINTER(C.c_double))
    probs.append((sc, ba, dc, dp, pc, pp))
torch.cuda.synchronize()
for label, k in (("one stream", 1), ("two streams", 2)):
    for it in range(20):
        sc, ba, dc, dp, pc, pp = probs[it % k]
        assert ba.lib.vs_ba_linearize(ba.h, pc, pp) == 0
    torch.cuda.synchronize()
    steps = 400
    t = time.perf_counter()
    for it in range(steps):
        sc, ba, dc, dp, pc, pp = probs[it % k]
        ba.lib.vs_ba_linearize(ba.h, pc, pp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    print(name, label, "us per step", round(dt * 1e6, 2), "obs/s %.3e" % (probs[0][0].No / dt))
