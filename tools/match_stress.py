"""Determinism of the batched matcher: the same descriptors in every pair of a batch, run repeatedly;
counts (run, pair) results whose nearest-neighbour arrays differ from the exact CPU argmin."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_slam_amd import matcher as vm, synth
from oracle import oracle as orc
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
d0, d1 = synth.make_descriptors(n, n, seed=41, duplicates=4)
dm = orc.distance_matrix(d0, d1)
e0 = dm.argmin(axis=1).astype(np.int32); e1 = dm.argmin(axis=0).astype(np.int32)
mt = vm.Matcher(pairs, n, n)
for p in range(pairs):
    mt.set_pair(p, d0, d1)
tot = 0; badr = 0; badc = 0
for r in range(runs):
    mt.run(pairs, 0.7)
    for p in range(pairs):
        j0, j1 = mt.nn(p)
        a, b = int((j0 != e0).sum()), int((j1 != e1).sum())
        tot += 1; badr += a; badc += b
print("pairs", pairs, "n", n, "results", tot, "wrong rows", badr, "wrong cols", badc, "rescans/run", mt.rescans(), flush=True)
mt.close()
