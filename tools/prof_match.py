"""Runs the batched matcher a few times for profiling (python3 tools_prof_match.py [pairs] [iters])."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_slam_amd import matcher as vm
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 4096
mt = vm.Matcher(pairs, n, n)
g = torch.Generator(device="cuda"); g.manual_seed(1)
keep = []
for p in range(pairs):
    d0 = torch.randn(n, 256, device="cuda", generator=g); d0 = d0 / d0.norm(dim=1, keepdim=True)
    d1 = torch.randn(n, 256, device="cuda", generator=g)
    n_in = int(n * 0.7)
    src = torch.randperm(n, device="cuda", generator=g)[:n_in]
    d1[:n_in] = d0[src] + (0.3 / 16.0) * torch.randn(n_in, 256, device="cuda", generator=g)
    d1 = d1[torch.randperm(n, device="cuda", generator=g)]
    d1 = (d1 / d1.norm(dim=1, keepdim=True)).contiguous()
    keep.append((d0, d1)); mt.set_pair(p, (d0.data_ptr(), n), (d1.data_ptr(), n))
torch.cuda.synchronize()
for _ in range(int(os.environ.get('VS_PROF_WARM', '100'))):
    mt.run(pairs, 0.7)
torch.cuda.synchronize()
ms = mt.profile_run(pairs, 0.7, iters=iters)
print('pairs', pairs, 'stages ms', [round(float(x), 3) for x in ms], 'rescans', mt.rescans())
