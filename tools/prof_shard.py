"""Linearisation stage times of ONE landmark shard of a BA shape on one GPU
(python3 tools/prof_shard.py c4 8 [rank]): what a rank of an N-GPU run computes per step."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bench import make_problem
name = sys.argv[1] if len(sys.argv) > 1 else 'c4'
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else world // 2
sc, ba = make_problem(name, rank, world)
dc = np.full(6 * sc.Nc, 1e-4); dp = np.full(3 * sc.Np, 1e-4)
ms = ba.profile_linearize(dc, dp, iters=200)
print(name, 'rank', rank, 'of', world, [round(float(x) * 1e3, 1) for x in ms])
