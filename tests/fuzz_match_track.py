"""Randomised parity sweep on the GPU, second part (a script, not collected by pytest:
python3 tests/fuzz_match_track.py [seed] [n]).  Matcher: batches of pairs of different sizes in one
handle, the allocation-free entry point, row shards with random splits (two threads); tracking
helpers: reprojection, searchProjection with random radii / thresholds / pre-claimed keypoints,
computePoint3d.  Everything bit for bit against the CPU oracle."""
import ctypes as C
import os
import sys
import threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import oracle as orc
from visual_slam_amd import _lib, synth
from visual_slam_amd import dist as vdist
from visual_slam_amd import matcher as vm
from visual_slam_amd import tracking as vt

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed)
lib = _lib.load()
bad = 0


def same(a, b):
    return (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            and np.array_equal(np.asarray(a[2], np.float32).view(np.uint32), np.asarray(b[2], np.float32).view(np.uint32)))


def report(kind, case, desc, ok):
    global bad
    bad += not ok
    print("%-6s case %3d %s : %s" % (kind, case, desc, "ok" if ok else "MISMATCH"), flush=True)


# --- batches: pairs of different sizes in one handle, two-way or stereo
for case in range(n_cases):
    n_pairs = int(rng.integers(1, 6))
    stereo = bool(rng.random() < 0.4)
    ratio = float(rng.choice([1.0, 0.8]))
    th = float(rng.choice([2.0, 0.9]))
    sizes = [(int(rng.integers(1, 900)), int(rng.integers(1, 900))) for _ in range(n_pairs)]
    m = vm.Matcher(n_pairs, max(s[0] for s in sizes), max(s[1] for s in sizes))
    want = []
    for p, (n0, n1) in enumerate(sizes):
        d0, d1 = synth.make_descriptors(n0, n1, seed=int(rng.integers(1 << 30)), duplicates=int(rng.integers(0, 3)))
        if stereo:
            xy0, xy1 = synth.make_keypoints(n0, seed=case + p), synth.make_keypoints(n1, seed=case + p + 77)
            m.set_pair(p, d0, d1, xy0, xy1)
            want.append(orc.stereo_matching(d0, xy0, d1, xy1, th, ratio))
        else:
            m.set_pair(p, d0, d1)
            want.append(orc.two_way_matching(d0, d1, th, ratio))
    m.run(n_pairs, th, ratio)
    ok = all(same(m.matches(p), want[p]) for p in range(n_pairs))
    m.close()
    report("BATCH", case, "%d pairs %s stereo %d ratio %.1f" % (n_pairs, sizes, stereo, ratio), ok)

# --- allocation-free entry point (device pointers, limit 2048)
for case in range(n_cases):
    n0, n1 = int(rng.integers(1, 2049)), int(rng.integers(1, 2049))
    ratio = float(rng.choice([1.0, 0.8]))
    d0, d1 = synth.make_descriptors(n0, n1, seed=int(rng.integers(1 << 30)))
    t0, t1 = torch.from_numpy(d0).cuda(), torch.from_numpy(d1).cuda()
    oi = torch.zeros(n0, dtype=torch.int32, device="cuda"); oj = torch.zeros(n0, dtype=torch.int32, device="cuda")
    od = torch.zeros(n0, dtype=torch.float32, device="cuda"); on = torch.zeros(1, dtype=torch.int32, device="cuda")
    rc = lib.vs_match_two_way_noalloc(C.c_void_p(t0.data_ptr()), n0, C.c_void_p(t1.data_ptr()), n1, C.c_float(0.9),
                                      C.c_float(ratio), C.c_void_p(oi.data_ptr()), C.c_void_p(oj.data_ptr()),
                                      C.c_void_p(od.data_ptr()), C.c_void_p(on.data_ptr()), None)
    torch.cuda.synchronize()
    n = int(on.item())
    got = (oi[:n].cpu().numpy(), oj[:n].cpu().numpy(), od[:n].cpu().numpy())
    report("NOALLOC", case, "%d x %d ratio %.1f" % (n0, n1, ratio), rc == 0 and same(got, orc.two_way_matching(d0, d1, 0.9, ratio)))


# --- row shards, two threads, random split
class TwoRankMin:
    def __init__(self):
        self.barrier = threading.Barrier(2, timeout=60)
        self.stash = [None, None]

    def hook(self, rank):
        def fn(ptr, n):
            t = vdist.device_tensor(ptr, n)
            self.stash[rank] = t.cpu().numpy().view(np.uint64).copy()
            self.barrier.wait()
            both = np.minimum(self.stash[0], self.stash[1])
            t.copy_(torch.from_numpy(both.view(np.float64)))
            torch.cuda.synchronize()
            self.barrier.wait()
        return fn


for case in range(n_cases):
    n0, n1 = int(rng.integers(2, 1200)), int(rng.integers(1, 1200))
    split = int(rng.integers(0, n0 + 1))
    ratio = float(rng.choice([1.0, 0.8]))
    d0, d1 = synth.make_descriptors(n0, n1, seed=int(rng.integers(1 << 30)), duplicates=int(rng.integers(0, 4)))
    red = TwoRankMin()
    bounds = [0, split, n0]
    ms = []
    for r in range(2):
        lo, hi = bounds[r], bounds[r + 1]
        m = vm.Matcher(1, max(hi - lo, 1), n1)
        m.set_pair(0, d0[lo:hi], d1)
        m.set_min_allreduce(red.hook(r))
        ms.append(m)
    out, err = [None, None], [None, None]

    def work(r):
        try:
            ms[r].run_sharded(1, bounds[r], 0.9, ratio)
            out[r] = ms[r].matches(0)
        except BaseException as e:
            err[r] = e
    th_ = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    for t in th_:
        t.start()
    for t in th_:
        t.join(120)
    for m in ms:
        m.close()
    ok = err[0] is None and err[1] is None
    if ok:
        got = [np.concatenate([out[0][k], out[1][k]]) for k in range(3)]
        ok = same(got, orc.two_way_matching(d0, d1, 0.9, ratio))
    report("SHARD", case, "%d x %d split %d ratio %.1f" % (n0, n1, split, ratio), ok)

# --- tracking helpers
for case in range(n_cases):
    n_tp, n_keys = int(rng.integers(1, 1200)), int(rng.integers(1, 1500))
    pose, K4, Pw, desc, key_xy, key_desc = synth.make_tracking_frame(n_tp, n_keys, seed=int(rng.integers(1 << 30)),
                                                                     rivals=float(rng.random() * 0.5))
    re = orc.reproject_points(pose, K4, Pw)
    ok = np.array_equal(vt.reproject_points(pose, K4, Pw), re)
    radius = float(rng.choice([5.0, 30.0, 200.0]))
    accept = float(rng.choice([0.3, 0.8, 2.5]))
    claimed = (rng.random(n_keys) < 0.2).astype(np.uint8)
    want_t, want_c = orc.search_projection(re, desc, key_xy, key_desc, 2 * K4[2], 2 * K4[3], radius, accept, claimed)
    got_t, got_c = vt.search_projection(pose, K4, Pw, desc, key_xy, key_desc, radius, accept, claimed)
    ok = ok and np.array_equal(got_t, want_t) and np.array_equal(got_c, want_c)
    report("TRACK", case, "%d points %d keys radius %.0f accept %.1f" % (n_tp, n_keys, radius, accept), ok)

print("mismatches:", bad)
sys.exit(1 if bad else 0)
