"""Matcher batches alternating between two handles on two streams against one handle on one stream
(python3 tools/match_two_streams.py): do the post-processing kernels of one batch overlap with the
score kernel of the next?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visual_slam_amd import matcher as vm
pairs, n = 64, 4096
g = torch.Generator(device="cuda"); g.manual_seed(1)
data = []
for p in range(pairs):
    d0 = torch.randn(n, 256, device="cuda", generator=g); d0 = d0 / d0.norm(dim=1, keepdim=True)
    d1 = torch.randn(n, 256, device="cuda", generator=g)
    n_in = int(n * 0.7)
    src = torch.randperm(n, device="cuda", generator=g)[:n_in]
    d1[:n_in] = d0[src] + (0.3 / 16.0) * torch.randn(n_in, 256, device="cuda", generator=g)
    d1 = d1[torch.randperm(n, device="cuda", generator=g)]
    d1 = (d1 / d1.norm(dim=1, keepdim=True)).contiguous()
    data.append((d0, d1))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ms = []
for s in streams:
    m = vm.Matcher(pairs, n, n)
    m.set_stream(s.cuda_stream)
    for p, (d0, d1) in enumerate(data):
        m.set_pair(p, (d0.data_ptr(), n), (d1.data_ptr(), n))
    ms.append(m)
torch.cuda.synchronize()
for label, k in (("one stream", 1), ("two streams", 2)):
    for it in range(4):
        ms[it % k].run(pairs, 0.7)
    torch.cuda.synchronize()
    steps = 40
    t = time.perf_counter()
    for it in range(steps):
        ms[it % k].run(pairs, 0.7)
    torch.cuda.synchronize()
    print(label, "ms per batch", round((time.perf_counter() - t) / steps * 1e3, 4))
