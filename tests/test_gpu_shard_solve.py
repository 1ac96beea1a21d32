"""GPU test of the landmark-sharded LM path end to end: two shard handles in one process, one
thread each, whose all-reduce hook really adds the peer's buffer (a two-rank all-reduce played
by two threads and a barrier).  Both ranks must take every LM decision identically and end with
the unsharded solve's iterations, cost and parameters; after the solve every rank holds all points."""
import threading

import numpy as np
import pytest

from visual_slam_amd import dist as vdist
from visual_slam_amd import synth
from visual_slam_amd.ba import BAProblem

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


class TwoRankAllReduce:
    """Sum across two threads: each copies its buffer aside, both wait, each adds the peer's copy.
    a + b == b + a in IEEE arithmetic, so both ranks end with bitwise identical sums."""

    def __init__(self, torch):
        self.torch = torch
        self.barrier = threading.Barrier(2, timeout=60)
        self.stash = [None, None]
        self.calls = [0, 0]

    def hook(self, rank):
        def fn(ptr, n):
            t = vdist.device_tensor(ptr, n)
            self.stash[rank] = t.clone()
            self.torch.cuda.synchronize()
            self.barrier.wait()
            t += self.stash[1 - rank]
            self.torch.cuda.synchronize()
            self.barrier.wait()
            self.calls[rank] += 1
        return fn


def run_two_ranks(fn):
    out, err = [None, None], [None, None]

    def work(r):
        try:
            out[r] = fn(r)
        except BaseException as e:       # surfaced in the main thread
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


@pytest.mark.parametrize("fix_points", [False, True])
def test_two_sharded_handles_solve_like_one(fix_points):
    import torch
    sc = synth.make_ba_config("c1", seed=13)
    pts = sc.pts_gt if fix_points else sc.pts0
    whole = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, pts, sc.cam, sc.pt, sc.uv, fix_points=fix_points)
    dw, pw, mw = whole.lm_step(1e4)
    ar = TwoRankAllReduce(torch)
    parts = [BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, pts, sc.cam, sc.pt, sc.uv, fix_points=fix_points,
                                   rank=r, world=2) for r in range(2)]
    for r in range(2):
        parts[r].set_allreduce(ar.hook(r))
    steps = run_two_ranks(lambda r: parts[r].lm_step(1e4))
    owner = vdist.landmark_owner(sc.cam, sc.pt, sc.Np, sc.Nc, 2)
    for r in range(2):
        d, p, m = steps[r]
        assert rel(d, dw) < 1e-8 and abs(m - mw) < 1e-9 * abs(mw)       # summed scalars are global
        mine = owner == r
        if not fix_points:
            assert rel(p[mine], pw[mine]) < 1e-8                        # a rank steps its own landmarks
            assert not p[~mine].any()
    assert np.array_equal(steps[0][0], steps[1][0])                     # cameras bitwise equal on both ranks
    assert ar.calls[0] == ar.calls[1] and ar.calls[0] >= 3
    # full solve: identical decisions on both ranks, equal to the unsharded solve
    ref = whole.solve()
    outs = run_two_ranks(lambda r: parts[r].solve())
    for r in range(2):
        o = outs[r]
        assert o["iterations"] == ref["iterations"] and o["successful_steps"] == ref["successful_steps"]
        assert o["termination"] == ref["termination"]
        assert abs(o["initial_cost"] - ref["initial_cost"]) < 1e-12 * ref["initial_cost"]
        assert abs(o["final_cost"] - ref["final_cost"]) < 1e-9 * ref["final_cost"]
        t, rpy = parts[r].cameras()
        tw, rw = whole.cameras()
        assert rel(t, tw) < 1e-7 and rel(rpy, rw) < 1e-7
        assert rel(parts[r].points(), whole.points()) < 1e-7            # gathered: every rank holds all points
    assert outs[0] == outs[1]
    assert np.array_equal(parts[0].cameras()[0], parts[1].cameras()[0])
    assert np.array_equal(parts[0].points(), parts[1].points())


def test_shard_bounds_of_the_library_follow_the_python_partition():
    """vs_ba_set_observations_sharded's landmark shards (read back through which landmarks get a
    non-zero V^-1) equal visual_slam_amd.dist.landmark_owner's."""
    sc = synth.make_ba_scene(30, 1500, 5, band=12, seed=3)
    dp = np.full(3 * sc.Np, 0.1)
    for world in (2, 3, 5):
        owner = vdist.landmark_owner(sc.cam, sc.pt, sc.Np, sc.Nc, world)
        for r in range(world):
            part = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv, rank=r, world=world)
            part.linearize(None, dp)
            Vinv6, _ = part.landmark_blocks()
            live = np.nonzero(np.abs(Vinv6).max(axis=1) > 0)[0]
            seen = np.unique(sc.pt[owner[sc.pt] == r])
            assert np.array_equal(live, seen)
            part.close()


def test_native_communicator_with_one_rank():
    """vs_comm_* over RCCL with a world of one (all a single-GPU box can host): the collectives
    run on the handle's stream and the solve equals the plain one."""
    import ctypes as C
    from visual_slam_amd import _lib
    lib = _lib.load()
    ident = (C.c_char * 128)()
    _lib.check(lib.vs_comm_unique_id(ident))
    comm = C.c_void_p()
    _lib.check(lib.vs_comm_create(C.byref(comm), 0, 1, ident))
    sc = synth.make_ba_config("c1", seed=14)
    a = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    b = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    _lib.check(lib.vs_ba_set_comm(b.h, comm))
    ra, rb = a.solve(), b.solve()
    assert ra == rb and np.array_equal(a.points(), b.points())
    import torch
    x = torch.arange(1000, dtype=torch.float64, device="cuda")
    _lib.check(lib.vs_comm_allreduce_sum(comm, C.c_void_p(x.data_ptr()), C.c_int64(1000), None))
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float64))
    # the all-gather the segment exchange is built on (one rank: the receive buffer is the send buffer)
    y = torch.zeros(1000, dtype=torch.float64, device="cuda")
    _lib.check(lib.vs_comm_allgather(comm, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), C.c_int64(1000), None))
    torch.cuda.synchronize()
    assert torch.equal(y.cpu(), torch.arange(1000, dtype=torch.float64))
    _lib.check(lib.vs_ba_set_comm(b.h, None))
    b.close()
    _lib.check(lib.vs_comm_destroy(comm))


def test_sharded_handles_give_the_same_answers_when_used_twice():
    """A second solve on the same shard handles (after a first one has run its candidate-cost
    passes) starts from the same initial cost and takes the same iterates as the unsharded handle:
    the cost partials of a candidate evaluation must not leak into the next linearisation.  The
    scene has enough landmarks for rank 1's first landmark tile to lie beyond index 0."""
    import torch
    sc = synth.make_ba_scene(12, 3000, 4, band=6, seed=21)
    whole = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    ar = TwoRankAllReduce(torch)
    parts = [BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv, rank=r, world=2)
             for r in range(2)]
    for r in range(2):
        parts[r].set_allreduce(ar.hook(r))
    for round_ in range(2):
        whole.set_cameras(sc.t0, sc.rpy0); whole.set_points(sc.pts0)
        for r in range(2):
            parts[r].set_cameras(sc.t0, sc.rpy0); parts[r].set_points(sc.pts0)
        ref = whole.solve(max_num_iterations=4)
        outs = run_two_ranks(lambda r: parts[r].solve(max_num_iterations=4))
        for r in range(2):
            o = outs[r]
            assert abs(o["initial_cost"] - ref["initial_cost"]) < 1e-12 * ref["initial_cost"], (round_, r)
            assert o["iterations"] == ref["iterations"] and o["successful_steps"] == ref["successful_steps"]
            assert abs(o["final_cost"] - ref["final_cost"]) < 1e-9 * ref["final_cost"]
    # a shard with no way to sum refuses to step on its own part
    lone = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv, rank=1, world=2)
    with pytest.raises(Exception):
        lone.lm_step(1e4)


class TwoRankAllGather:
    """All-gather across two threads: each rank copies its piece into both ranks' receive buffers."""

    def __init__(self, torch):
        self.torch = torch
        self.barrier = threading.Barrier(2, timeout=60)
        self.recv = [None, None]
        self.calls = [0, 0]
        self.doubles = [0, 0]

    def hook(self, rank):
        def fn(send_ptr, recv_ptr, n):
            self.recv[rank] = vdist.device_tensor(recv_ptr, 2 * n)
            self.barrier.wait()
            send = vdist.device_tensor(send_ptr, n)
            for r in range(2):
                self.recv[r][rank * n:(rank + 1) * n] = send
            self.torch.cuda.synchronize()
            self.barrier.wait()
            self.calls[rank] += 1
            self.doubles[rank] = n
        return fn


@pytest.mark.parametrize("scene", ["c1", "banded"])
def test_segment_exchange_equals_the_all_reduce(scene):
    """vs_ba_exchange_reduced through the gather callback (each rank sends only the rows its
    landmarks touch) against the whole-buffer all-reduce: the same reduce buffer on both ranks,
    equal to the unsharded one; the solve takes the same iterates; the payload is a fraction of
    the buffer when the trajectory is long."""
    import torch
    sc = synth.make_ba_config("c1", seed=13) if scene == "c1" else synth.make_ba_scene(200, 6000, 4, band=10, seed=23)
    dc = np.full(6 * sc.Nc, 1e-3); dp = np.full(3 * sc.Np, 1e-3)
    whole = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    whole.linearize(None, dp)
    ptr_w, n = whole.reduce_buffer()
    want = vdist.device_tensor(ptr_w, n).cpu().numpy().copy()
    ag, ar = TwoRankAllGather(torch), TwoRankAllReduce(torch)
    parts = [BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv, rank=r, world=2) for r in range(2)]
    for r in range(2):
        parts[r].set_allgather(ag.hook(r))
        parts[r].set_allreduce(ar.hook(r))

    def lin_and_exchange(r):
        parts[r].linearize(None, dp)
        parts[r].exchange_reduced()
        p, m = parts[r].reduce_buffer()
        return vdist.device_tensor(p, m).cpu().numpy().copy()
    got = run_two_ranks(lin_and_exchange)
    assert np.array_equal(got[0], got[1])                         # the pieces are added in rank order on both ranks
    assert rel(got[0], want) < 1e-12
    assert ag.calls == [1, 1] and ar.calls == [0, 0]              # the buffer went through the gather only
    sizes = [parts[r].exchange_size() for r in range(2)]
    assert sizes[0][1] == sizes[1][1] == 2 * ag.doubles[0]
    if scene == "banded":
        assert max(s[0] for s in sizes) < 0.6 * n                 # each rank sends about half of the buffer
    ref = whole.solve(max_num_iterations=6)
    outs = run_two_ranks(lambda r: parts[r].solve(max_num_iterations=6))
    for r in range(2):
        assert outs[r]["iterations"] == ref["iterations"] and outs[r]["successful_steps"] == ref["successful_steps"]
        assert abs(outs[r]["final_cost"] - ref["final_cost"]) < 1e-9 * ref["final_cost"]
        assert rel(parts[r].points(), whole.points()) < 1e-7
    assert outs[0] == outs[1] and np.array_equal(parts[0].cameras()[0], parts[1].cameras()[0])
    assert ag.calls[0] > 2
