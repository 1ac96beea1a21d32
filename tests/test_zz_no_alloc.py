"""GPU test of the allocation-free matcher entry point (vs_match_two_way_noalloc): its output is
compared, bit for bit, with vs_match_two_way's on the same descriptors, on two calls in a row of
which the second has the smaller input (the static scratch array keeps the first call's contents)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc
from visual_slam_amd import _lib, synth
from visual_slam_amd import matcher as vm

pytestmark = pytest.mark.gpu


def run_noalloc(lib, torch, d0, d1, nn_thresh, ratio):
    n0, n1 = len(d0), len(d1)
    t0 = torch.from_numpy(d0).cuda()
    t1 = torch.from_numpy(d1).cuda()
    oi = torch.full((max(n0, 1),), -7, dtype=torch.int32, device="cuda")
    oj = torch.full((max(n0, 1),), -7, dtype=torch.int32, device="cuda")
    od = torch.zeros(max(n0, 1), dtype=torch.float32, device="cuda")
    on = torch.full((1,), -7, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()
    rc = lib.vs_match_two_way_noalloc(C.c_void_p(t0.data_ptr()), n0, C.c_void_p(t1.data_ptr()), n1,
                                      C.c_float(nn_thresh), C.c_float(ratio), C.c_void_p(oi.data_ptr()),
                                      C.c_void_p(oj.data_ptr()), C.c_void_p(od.data_ptr()),
                                      C.c_void_p(on.data_ptr()), C.c_void_p(stream.cuda_stream))
    _lib.check(rc)
    stream.synchronize()
    n = int(on.item())
    return oi[:n].cpu().numpy(), oj[:n].cpu().numpy(), od[:n].cpu().numpy()


def same(a, b):
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert np.array_equal(np.asarray(a[2]).view(np.uint32), np.asarray(b[2]).view(np.uint32))


def test_noalloc_matches_the_allocating_entry_point_twice_in_a_row():
    import torch
    lib = _lib.load()
    # first call: the largest size the entry point takes, with duplicated rows (ties -> exact rescans)
    d0, d1 = synth.make_descriptors(2048, 2000, seed=61, duplicates=3)
    got = run_noalloc(lib, torch, d0, d1, 0.7, 1.0)
    same(got, vm.two_way(d0, d1, 0.7, 1.0))
    same(got, orc.two_way_matching(d0, d1, 0.7))      # and with the CPU oracle directly
    # second call, smaller and ragged: nothing of the first call's scratch contents may leak in
    e0, e1 = synth.make_descriptors(300, 777, seed=62, duplicates=2)
    got = run_noalloc(lib, torch, e0, e1, 2.0, 1.0)
    same(got, vm.two_way(e0, e1, 2.0, 1.0))
    assert len(got[0]) > 0
    # ratio test variant and an empty side
    same(run_noalloc(lib, torch, e0, e1, 2.0, 0.8), vm.two_way(e0, e1, 2.0, 0.8))
    assert len(run_noalloc(lib, torch, e0[:0], e1, 2.0, 1.0)[0]) == 0


def test_noalloc_allocates_no_device_memory():
    """Free device memory is the same before and after a run of calls (the first call of the process
    may load the code object, so one call comes first)."""
    import torch
    lib = _lib.load()
    d0, d1 = synth.make_descriptors(512, 640, seed=63)
    t0 = torch.from_numpy(d0).cuda(); t1 = torch.from_numpy(d1).cuda()
    oi = torch.zeros(512, dtype=torch.int32, device="cuda"); oj = torch.zeros_like(oi)
    od = torch.zeros(512, device="cuda"); on = torch.zeros(1, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()

    def call():
        _lib.check(lib.vs_match_two_way_noalloc(C.c_void_p(t0.data_ptr()), 512, C.c_void_p(t1.data_ptr()), 640,
                                                C.c_float(0.7), C.c_float(0.8), C.c_void_p(oi.data_ptr()),
                                                C.c_void_p(oj.data_ptr()), C.c_void_p(od.data_ptr()),
                                                C.c_void_p(on.data_ptr()), C.c_void_p(stream.cuda_stream)))
    call(); stream.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        call()
    stream.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    # misaligned descriptors are rejected (they are read in place as 16-byte vectors)
    rc = lib.vs_match_two_way_noalloc(C.c_void_p(t0.data_ptr() + 4), 8, C.c_void_p(t1.data_ptr()), 8,
                                      C.c_float(0.7), C.c_float(1.0), C.c_void_p(oi.data_ptr()),
                                      C.c_void_p(oj.data_ptr()), C.c_void_p(od.data_ptr()),
                                      C.c_void_p(on.data_ptr()), C.c_void_p(stream.cuda_stream))
    assert rc == -1 and b"aligned" in lib.vs_last_error_string()


def test_noalloc_rejects_sizes_beyond_its_limit_and_host_pointers():
    import torch
    lib = _lib.load()
    big = torch.zeros(2049, 256, device="cuda")
    out = torch.zeros(2049, dtype=torch.int32, device="cuda")
    outf = torch.zeros(2049, device="cuda")
    rc = lib.vs_match_two_way_noalloc(C.c_void_p(big.data_ptr()), 2049, C.c_void_p(big.data_ptr()), 10,
                                      C.c_float(2.0), C.c_float(1.0), C.c_void_p(out.data_ptr()),
                                      C.c_void_p(out.data_ptr()), C.c_void_p(outf.data_ptr()),
                                      C.c_void_p(out.data_ptr()), None)
    assert rc == -1 and b"limit" in lib.vs_last_error_string()
    host = np.zeros((4, 256), np.float32)
    rc = lib.vs_match_two_way_noalloc(host.ctypes.data_as(C.c_void_p), 4, C.c_void_p(big.data_ptr()), 4,
                                      C.c_float(2.0), C.c_float(1.0), C.c_void_p(out.data_ptr()),
                                      C.c_void_p(out.data_ptr()), C.c_void_p(outf.data_ptr()),
                                      C.c_void_p(out.data_ptr()), None)
    assert rc == -1 and b"device pointer" in lib.vs_last_error_string()
