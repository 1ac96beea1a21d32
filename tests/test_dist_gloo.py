"""CPU, world_size 2 over gloo: the landmark partition and the all-reduce of per-shard reduced
camera systems reproduce the unsharded system.  Each rank's shard is linearised by the CPU oracle
(test infrastructure standing in for the kernels); the partition rule and the reduction code are
the ones the GPU path uses (visual_slam_amd/dist.py)."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import oracle as orc
    from visual_slam_amd import dist as vd
    from visual_slam_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = synth.make_ba_config("c1", seed=21)
    m = vd.shard_mask(sc.cam, sc.pt, sc.Np, sc.Nc, rank, world)
    dc = np.full(6 * sc.Nc, 0.3); dp = np.full(3 * sc.Np, 0.3)
    p = orc.Problem(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam[m], sc.pt[m], sc.uv[m])
    s = p.schur(dc=None, dp=dp)
    buf = np.concatenate([s["S"].ravel(), s["rhs"], [p.cost()], [float(m.sum())]])
    vd.allreduce_sum_(buf)
    n = 6 * sc.Nc
    S = buf[:n * n].reshape(n, n) + np.diag(dc)       # camera damping after the cross-rank sum
    rhs = buf[n * n:n * n + n]
    whole = orc.Problem(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    w = whole.schur(dc=dc, dp=dp)
    err_S = np.abs(S - w["S"]).max() / np.abs(w["S"]).max()
    err_r = np.abs(rhs - w["rhs"]).max() / np.abs(w["rhs"]).max()
    err_c = abs(buf[n * n + n] - whole.cost()) / whole.cost()
    out[rank] = (err_S, err_r, err_c, buf[-1], float(m.sum()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_sum_to_the_whole():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    total = None
    for r in range(world):
        err_S, err_r, err_c, n_all, n_mine = out[r]
        assert err_S < 1e-12 and err_r < 1e-12 and err_c < 1e-12
        total = n_all
        assert 0 < n_mine < n_all
    from visual_slam_amd import synth
    assert total == synth.make_ba_config("c1", seed=21).No      # every edge owned exactly once


def test_partition_is_contiguous_in_the_internal_order_and_balanced():
    from visual_slam_amd import dist as vd
    from visual_slam_amd import synth
    sc = synth.make_ba_scene(20, 5000, 5, band=8, seed=3)
    new2old = vd.landmark_order(sc.cam, sc.pt, sc.Np, sc.Nc)
    assert np.array_equal(np.sort(new2old), np.arange(sc.Np))
    first = np.full(sc.Np, sc.Nc); np.minimum.at(first, sc.pt, sc.cam)
    assert np.all(np.diff(first[new2old]) >= 0)                       # by first camera, stable
    for world in (1, 2, 3, 8):
        owner = vd.landmark_owner(sc.cam, sc.pt, sc.Np, sc.Nc, world)
        assert np.all(np.diff(owner[new2old]) >= 0) and owner.min() == 0 and owner.max() == world - 1
        counts = [int(vd.shard_mask(sc.cam, sc.pt, sc.Np, sc.Nc, r, world).sum()) for r in range(world)]
        assert sum(counts) == sc.No
        assert max(counts) - min(counts) <= 2 * 8 + sc.No // (50 * world)
        # a shard sees a contiguous range of cameras: its landmarks start in one stretch of the trajectory
        if world > 1:
            lo = [first[owner == r].min() for r in range(world)]
            assert lo == sorted(lo)


def _match_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    from visual_slam_amd import dist as vd
    from visual_slam_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d0, d1 = synth.make_descriptors(300, 280, seed=4)
    d0[200] = d0[17]                                          # a tie across the shard boundary
    bounds = [0, 130, 300]
    lo, hi = bounds[rank], bounds[rank + 1]
    dm = orc.distance_matrix(d0[lo:hi], d1)                   # this rank's rows, all columns
    keys = torch.from_numpy(vd.column_keys(dm, lo))
    dist.all_reduce(keys, op=dist.ReduceOp.MIN)               # the exchange step of the sharded matcher
    out[rank] = vd.matches_from_keys(dm, lo, keys.numpy(), 0.9)
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_matcher_keys_reduce_to_the_whole_pair():
    """World 2 over gloo: packed column keys + MIN all-reduce reproduce twoWayMatching of the whole
    pair (the oracle), first-minimum ties included."""
    from oracle import oracle as orc
    from visual_slam_amd import synth
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    port = 31500 + os.getpid() % 2000
    mp.spawn(_match_worker, args=(world, port, out), nprocs=world, join=True)
    d0, d1 = synth.make_descriptors(300, 280, seed=4)
    d0[200] = d0[17]
    wi, wj, wd = orc.two_way_matching(d0, d1, 0.9)
    gi = np.concatenate([out[0][0], out[1][0]]); gj = np.concatenate([out[0][1], out[1][1]])
    gd = np.concatenate([out[0][2], out[1][2]])
    assert len(wi) > 50 and np.array_equal(gi, wi) and np.array_equal(gj, wj)
    assert np.array_equal(gd.view(np.uint32), wd.view(np.uint32))
