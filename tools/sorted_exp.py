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
        cam, pt, uv, pts0 = sc.cam, sc.pt, sc.uv, sc.pts0
        if mode == "sorted":
            first = np.full(sc.Np, 1 << 30); np.minimum.at(first, pt, cam)
            order = np.argsort(first, kind="stable")          # new id -> old id
            rank = np.empty_like(order); rank[order] = np.arange(sc.Np)
            pt2 = rank[pt]
            o = np.lexsort((cam, pt2))
            cam, pt, uv, pts0 = cam[o], pt2[o].astype(np.int32), uv[o], pts0[order]
        ba = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, pts0, cam, pt, uv)
        dc = np.full(6 * sc.Nc, 1e-4); dp = np.full(3 * sc.Np, 1e-4)
        ms = ba.profile_linearize(dc, dp, iters=20)
        print(name, mode, [round(float(x) * 1e3, 1) for x in ms], flush=True)
        ba.close()
