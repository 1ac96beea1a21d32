"""Times motion-only and full solves at the reference's window size (config 1), with and without
the one-launch motion-only solver: python3 tools/time_small.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_slam_amd import synth
from visual_slam_amd.ba import BAProblem

sc = synth.make_ba_config("c1", seed=0)
for label, env in (("one launch", None), ("general path", "1")):
    if env:
        os.environ["VS_BA_NO_SMALL_SOLVER"] = env
    gpu = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts_gt, sc.cam, sc.pt, sc.uv, fix_points=True)
    gpu.solve()
    ts = []
    for _ in range(20):
        gpu.set_cameras(sc.t0, sc.rpy0)
        t0 = time.perf_counter()
        g = gpu.solve()
        ts.append(time.perf_counter() - t0)
    print("motion-only c1, %-12s %.3f ms  %s" % (label, float(np.median(ts)) * 1e3, g), flush=True)
    gpu.close()
