"""Repeats one full-size one-shot match and reports any run whose output differs from the first
(python3 tools/match_stress.py [runs] [ratio])."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_slam_amd import matcher as vm, synth
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ratio = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
d0, d1 = synth.make_descriptors(4096, 4096, seed=41, duplicates=4)
ref = vm.two_way(d0, d1, 0.7, ratio)
bad = 0
for r in range(runs):
    got = vm.two_way(d0, d1, 0.7, ratio)
    if not all(np.array_equal(a, b) for a, b in zip(got, ref)):
        bad += 1
        if len(got[0]) != len(ref[0]):
            print("run", r, "lengths", len(got[0]), len(ref[0]))
            only_got = np.setdiff1d(got[0], ref[0]); only_ref = np.setdiff1d(ref[0], got[0])
            print("   rows only in this run", only_got[:8], "rows only in the first run", only_ref[:8])
        else:
            k = np.nonzero((got[0] != ref[0]) | (got[1] != ref[1]) | (got[2].view(np.uint32) != ref[2].view(np.uint32)))[0]
            print("run", r, "differs at", k[:8], [(int(got[0][i]), int(got[1][i]), float(got[2][i]), int(ref[0][i]), int(ref[1][i]), float(ref[2][i])) for i in k[:4]])
print("runs", runs, "differing", bad, flush=True)
