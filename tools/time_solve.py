"""Times whole vs_ba_solve runs (ten LM iterations of full BA) per config, problem set-up apart:
python3 tools/time_solve.py c1 c2 c4"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_slam_amd import synth
from visual_slam_amd.ba import BAProblem

for name in sys.argv[1:] or ["c4"]:
    sc = synth.make_ba_config(name, seed=0)
    t0 = time.perf_counter()
    gpu = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    build = time.perf_counter() - t0
    gpu.solve(max_num_iterations=3)
    for rep in range(2):
        gpu.set_cameras(sc.t0, sc.rpy0)
        gpu.set_points(sc.pts0)
        t0 = time.perf_counter()
        g = gpu.solve(max_num_iterations=10)
        dt = time.perf_counter() - t0
        print(name, "set-up %.2f ms, solve %.2f ms" % (build * 1e3, dt * 1e3), g, flush=True)
    gpu.close()
