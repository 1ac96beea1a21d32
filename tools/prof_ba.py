"""Runs the BA linearisation a few times for profiling (python3 tools_prof_ba.py c2 [iters])."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_problem
name = sys.argv[1] if len(sys.argv) > 1 else 'c2'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
sc, ba = make_problem(name, 0, 1)
dc = np.full(6 * sc.Nc, 1e-4); dp = np.full(3 * sc.Np, 1e-4)
ms = ba.profile_linearize(dc, dp, iters=iters)
print(name, [round(float(x) * 1e3, 1) for x in ms])
