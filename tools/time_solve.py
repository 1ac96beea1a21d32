import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from visual_slam_amd import synth
from visual_slam_amd.ba import BAProblem
for name in ("c1", "c2"):
    sc = synth.make_ba_config(name, seed=0)
    for fix in (True, False):
        pts = sc.pts_gt if fix else sc.pts0
        ref = orc.Problem(sc.K, sc.t0, sc.rpy0, pts, sc.cam, sc.pt, sc.uv, fix_points=fix)
        t0 = time.perf_counter(); r = ref.solve(); t_cpu = time.perf_counter() - t0
        gpu = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, pts, sc.cam, sc.pt, sc.uv, fix_points=fix)
        gpu.solve()
        gpu.set_cameras(sc.t0, sc.rpy0); gpu.set_points(pts)
        t0 = time.perf_counter(); g = gpu.solve(); t_gpu = time.perf_counter() - t0
        print(name, "fix_points", fix, "cpu %.1f ms (%d it)" % (t_cpu*1e3, r["iterations"]), "gpu %.1f ms (%d it) -> %.0f us/it" % (t_gpu*1e3, g["iterations"], t_gpu*1e6/max(g["iterations"],1)), flush=True)
        gpu.close()
