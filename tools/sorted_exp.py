"""Linearisation stage times with the landmarks numbered at random and in keyframe order
(python3 tools/sorted_exp.py c2 c4 ...)."""
import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
torch.cuda.init()
from visual_slam_amd import synth
from visual_slam_amd.ba import BAProblem
for name in sys.argv[1:]:
    Nc, Np, k, band = synth.BA_CONFIGS[name]
    sc = synth.make_ba_scene(Nc, Np, k, band=band, seed=0)
    for mode in ("random", "sorted"):
        s2 = synth.landmarks_in_keyframe_order(sc) if mode == "sorted" else sc
        cam, pt, uv, pts0 = s2.cam, s2.pt, s2.uv, s2.pts0
        ba = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, pts0, cam, pt, uv)
        dc = np.full(6 * sc.Nc, 1e-4); dp = np.full(3 * sc.Np, 1e-4)
        ms = ba.profile_linearize(dc, dp, iters=20)
        print(name, mode, [round(float(x) * 1e3, 1) for x in ms], flush=True)
        ba.close()
