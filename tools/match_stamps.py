"""Timeline of one workgroup of the matcher's score kernel (needs a library built with
-DVS_MATCH_STAMPS, selected through VSLAM_HIP_LIB).  Per slot every wave records four stamps:
slot entered, stage wait done (before the barrier), barrier passed, DMA issued; the time to the
next slot's first stamp is the slot's work (a K-step or half a reduction)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from visual_slam_amd import matcher as vm, _lib
pairs, n = 64, 4096
mt = vm.Matcher(pairs, n, n)
g = torch.Generator(device="cuda"); g.manual_seed(1)
keep = []
for p in range(pairs):
    d0 = torch.randn(n, 256, device="cuda", generator=g); d0 = d0 / d0.norm(dim=1, keepdim=True)
    d1 = torch.randn(n, 256, device="cuda", generator=g); d1 = d1 / d1.norm(dim=1, keepdim=True)
    keep.append((d0, d1)); mt.set_pair(p, (d0.data_ptr(), n), (d1.data_ptr(), n))
torch.cuda.synchronize()
mt.profile_run(pairs, 0.7, iters=3)
W, S = 8, 128
out = np.zeros(W * S, np.uint64)
lib = _lib.load()
rc = lib.vs_debug_match_stamps(out.ctypes.data_as(C.POINTER(C.c_ulonglong)))
assert rc == 0
t0 = out[out > 0].min()
for w in range(W):
    st = out[w * S:(w + 1) * S].astype(np.int64)
    st = st[st > 0] - int(t0)
    print("wave", w, "slots: [enter, wait, barrier, dma | work]")
    for i in range(0, len(st) - 4, 4):
        e, wt, b, d = st[i:i + 4]
        print("   at %7d: wait %5d barrier %5d dma %5d work %5d" % (e, wt - e, b - wt, d - b, st[i + 4] - d))
