"""Timeline of one workgroup of the matcher's score kernel (needs a library built with
-DVS_MATCH_STAMPS, selected through VSLAM_HIP_LIB).  Stamps per tile: tile start, then per K-step
(after the barrier, after the last MFMA was issued), after the column pass, after the row pass."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from visual_slam_amd import matcher as vm, _lib
pairs, n = 64, 4096
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4       # K-steps per tile of the build
mt = vm.Matcher(pairs, n, n)
g = torch.Generator(device="cuda"); g.manual_seed(1)
keep = []
for p in range(pairs):
    d0 = torch.randn(n, 256, device="cuda", generator=g); d0 = d0 / d0.norm(dim=1, keepdim=True)
    d1 = torch.randn(n, 256, device="cuda", generator=g); d1 = d1 / d1.norm(dim=1, keepdim=True)
    keep.append((d0, d1)); mt.set_pair(p, (d0.data_ptr(), n), (d1.data_ptr(), n))
torch.cuda.synchronize()
mt.profile_run(pairs, 0.7, iters=3)
out = np.zeros(4 * 64, np.uint64)
lib = _lib.load()
rc = lib.vs_debug_match_stamps(out.ctypes.data_as(C.POINTER(C.c_ulonglong)))
assert rc == 0
per_tile = 1 + 2 * steps + 2
for w in range(4):
    st = out[w * 64:(w + 1) * 64].astype(np.int64)
    st = st[st > 0]
    d = np.diff(st)
    print("wave", w, "deltas (cycles):")
    for t in range(0, len(d), per_tile):
        print("   ", d[t:t + per_tile].tolist())
