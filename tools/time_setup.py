"""Times problem set-up (vs_ba_create + set_cameras/points/observations + destroy) per config,
repeated so that first-call effects (code object load, first allocations) are apart:
python3 tools/time_setup.py c1 c2 c4      (VS_BA_BUILD_TIMING=1 prints the phases of one build)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visual_slam_amd import synth
from visual_slam_amd.ba import BAProblem

for name in sys.argv[1:] or ["c1"]:
    sc = synth.make_ba_config(name, seed=0)
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        gpu = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
        t1 = time.perf_counter()
        gpu.close()
        t2 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    print(name, "set-up / destroy ms:", " ".join("%.3f/%.3f" % t for t in ts), flush=True)
