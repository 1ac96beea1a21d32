"""Times one LM step (linearize + reduced solve + back-substitution + candidate cost) per config:
python3 tools/time_lm.py c4 c5shard"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visual_slam_amd import synth
from visual_slam_amd.ba import BAProblem

for name in sys.argv[1:] or ["c2"]:
    sc = synth.make_ba_config(name, seed=0)
    ba = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    ba.lm_step(1e4)
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        dcam, dpt, mcc = ba.lm_step(1e4)
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n):
        ba.lm_step(1e4, fetch=False)
    dd = (time.perf_counter() - t0) / n
    print(name, "cams", sc.Nc, "obs", sc.No, "lm_step %.2f ms (step left on the device: %.2f ms)" % (dt * 1e3, dd * 1e3),
          "model decrease %.6g" % mcc, "|dcam| %.3g" % np.abs(dcam).max(), flush=True)
