"""GPU test of the row-sharded matcher: one frame pair split by query rows over two "ranks" (two
matcher handles, one thread each); the minimum all-reduce of the packed column keys is played by
the two threads and a barrier.  The ranks' match lists, concatenated in rank order, must be the
whole pair's matches bit for bit (indices and distance bits), for two-way and masked stereo
matching, with the ratio test, with ties across the shard boundary and with an empty shard."""
import threading

import numpy as np
import pytest

from visual_slam_amd import dist as vdist
from visual_slam_amd import synth
from visual_slam_amd.matcher import Matcher

pytestmark = pytest.mark.gpu


class TwoRankMin:
    """Element-wise minimum of uint64 keys across two threads (through the host: the keys are
    compared as unsigned integers, which torch does not do on the device)."""

    def __init__(self, torch):
        self.torch = torch
        self.barrier = threading.Barrier(2, timeout=60)
        self.stash = [None, None]

    def hook(self, rank):
        def fn(ptr, n):
            t = vdist.device_tensor(ptr, n)              # float64 view of the same 8-byte words
            self.stash[rank] = t.cpu().numpy().view(np.uint64).copy()
            self.barrier.wait()
            both = np.minimum(self.stash[0], self.stash[1])
            t.copy_(self.torch.from_numpy(both.view(np.float64)))
            self.torch.cuda.synchronize()
            self.barrier.wait()
        return fn


def run_two(fn):
    out, err = [None, None], [None, None]

    def work(r):
        try:
            out[r] = fn(r)
        except BaseException as e:
            err[r] = e
    th = [threading.Thread(target=work, args=(r,)) for r in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    for e in err:
        if e is not None:
            raise e
    return out


def sharded_matches(torch, d0, d1, split, nn_thresh, ratio, xy0=None, xy1=None):
    red = TwoRankMin(torch)
    bounds = [0, split, len(d0)]
    ms = []
    for r in range(2):
        lo, hi = bounds[r], bounds[r + 1]
        m = Matcher(1, max(hi - lo, 1), len(d1))
        m.set_pair(0, d0[lo:hi], d1, None if xy0 is None else xy0[lo:hi], xy1)
        m.set_min_allreduce(red.hook(r))
        ms.append(m)

    def work(r):
        ms[r].run_sharded(1, bounds[r], nn_thresh, ratio)
        return ms[r].matches(0)
    parts = run_two(work)
    for m in ms:
        m.close()
    return [np.concatenate([parts[0][k], parts[1][k]]) for k in range(3)]


def same(a, b):
    return (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
            and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32)))


@pytest.mark.parametrize("n0,n1,split,ratio", [(1000, 900, 450, 1.0), (700, 1100, 123, 0.8), (300, 257, 0, 1.0),
                                              (2048, 2048, 1024, 0.8)])
def test_two_row_shards_equal_the_whole_pair(orc, n0, n1, split, ratio):
    import torch
    d0, d1 = synth.make_descriptors(n0, n1, seed=n0 + split)
    # ties across the shard boundary: identical rows on both sides compete for the same column
    if split > 3:
        d0[split + 2] = d0[split - 3]
        d0[n0 - 1] = d0[1]
    want = orc.two_way_matching(d0, d1, 0.9, ratio)
    got = sharded_matches(torch, d0, d1, split, 0.9, ratio)
    assert len(want[0]) > 50 and same(got, want)


def test_two_row_shards_of_a_masked_stereo_pair(orc):
    import torch
    n0, n1 = 800, 760
    d0, d1 = synth.make_descriptors(n0, n1, seed=5)
    xy0, xy1 = synth.make_keypoints(n0, seed=1), synth.make_keypoints(n1, seed=2)
    want = orc.stereo_matching(d0, xy0, d1, xy1, 2.0, 1.0)
    got = sharded_matches(torch, d0, d1, 311, 2.0, 1.0, xy0, xy1)
    assert len(want[0]) > 0 and same(got, want)


def test_sharded_run_with_one_rank_and_the_library_communicator(orc):
    """world = 1 through RCCL: the all-reduce is the identity and the result is the plain run's."""
    import ctypes as C
    from visual_slam_amd import _lib
    lib = _lib.load()
    d0, d1 = synth.make_descriptors(500, 400, seed=9)
    ident = (C.c_char * 128)()
    assert lib.vs_comm_unique_id(ident) == 0
    comm = C.c_void_p()
    assert lib.vs_comm_create(C.byref(comm), 0, 1, ident) == 0
    m = Matcher(1, 500, 400)
    m.set_pair(0, d0, d1)
    m.set_comm(comm)
    m.run_sharded(1, 0, 0.9, 0.8)
    got = m.matches(0)
    m.close()
    lib.vs_comm_destroy(comm)
    assert same(got, orc.two_way_matching(d0, d1, 0.9, 0.8))
