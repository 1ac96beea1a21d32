"""Python wrappers of the vs_matcher_* C ABI and a mirror of the reference's matcher methods
(reference: src/core/superpoint_frontend.h:28-35, src/core/superpoint_frontend.cc:110-211)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr

DESC_DIM = 256


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Matcher:
    """Batched, device-resident matcher (vs_matcher_*)."""

    def __init__(self, max_pairs, max_n0, max_n1):
        self.lib = _lib.load()
        self.h = C.c_void_p()
        self.max_pairs = int(max_pairs)
        self.sizes = {}
        check(self.lib.vs_matcher_create(C.byref(self.h), int(max_pairs), int(max_n0), int(max_n1)))

    def close(self):
        if self.h:
            self.lib.vs_matcher_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        check(self.lib.vs_matcher_set_stream(self.h, C.c_void_p(int(stream))))

    def set_pair(self, pair, desc0, desc1, xy0=None, xy1=None):
        """desc0/desc1: (n, 256) float32 numpy arrays, or (device_address, n) tuples."""
        def unpack(d):
            if isinstance(d, tuple):
                return int(d[0]), int(d[1])
            d = _f32(d).reshape(-1, DESC_DIM)
            return d, len(d)
        d0, n0 = unpack(desc0)
        d1, n1 = unpack(desc1)
        x0 = None if xy0 is None else _i32(xy0).reshape(-1, 2)
        x1 = None if xy1 is None else _i32(xy1).reshape(-1, 2)
        check(self.lib.vs_matcher_set_pair(self.h, int(pair), ptr(d0, C.c_float), n0,
                                           ptr(d1, C.c_float), n1, ptr(x0, C.c_int32),
                                           ptr(x1, C.c_int32)))
        self.sizes[int(pair)] = (n0, n1)

    def set_pairs(self, first_pair, desc0, desc1, xy0=None, xy1=None):
        """vs_matcher_set_pairs: desc0/desc1 are (pairs, n, 256) float32 numpy arrays, or
        (device_address, pairs, n, stride_in_floats) tuples; xy0/xy1 (pairs, n, 2) int32 or None."""
        def unpack(d):
            if isinstance(d, tuple):
                return int(d[0]), int(d[1]), int(d[2]), int(d[3])
            d = _f32(d)
            assert d.ndim == 3 and d.shape[2] == DESC_DIM
            return d, d.shape[0], d.shape[1], d.shape[1] * DESC_DIM
        d0, p0, n0, s0 = unpack(desc0)
        d1, p1, n1, s1 = unpack(desc1)
        assert p0 == p1
        x0 = None if xy0 is None else _i32(xy0).reshape(p0, n0, 2)
        x1 = None if xy1 is None else _i32(xy1).reshape(p0, n1, 2)
        check(self.lib.vs_matcher_set_pairs(self.h, int(first_pair), p0, ptr(d0, C.c_float), C.c_int64(s0), n0,
                                            ptr(d1, C.c_float), C.c_int64(s1), n1, ptr(x0, C.c_int32),
                                            ptr(x1, C.c_int32)))
        for p in range(p0):
            self.sizes[int(first_pair) + p] = (n0, n1)

    def run(self, n_pairs, nn_thresh=2.0, ratio=1.0):
        check(self.lib.vs_matcher_run(self.h, int(n_pairs), C.c_float(nn_thresh), C.c_float(ratio)))

    def set_comm(self, comm):
        """vs_matcher_set_comm: comm = a vs_comm handle (ctypes void pointer) or None."""
        check(self.lib.vs_matcher_set_comm(self.h, comm))

    def set_min_allreduce(self, fn):
        """vs_matcher_set_min_allreduce: fn(device_address, n) must leave the element-wise minimum
        over the ranks of the n uint64 at that device address in place."""
        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)

        def thunk(_user, buf, n):
            try:
                fn(int(buf), int(n))
                return 0
            except Exception:            # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._min_cb = proto(thunk)      # keep the callback object alive
        check(self.lib.vs_matcher_set_min_allreduce(self.h, self._min_cb, None))

    def run_sharded(self, n_pairs, row0, nn_thresh=2.0, ratio=1.0):
        """vs_matcher_run_sharded: this handle holds rows row0.. of desc0 and all of desc1."""
        check(self.lib.vs_matcher_run_sharded(self.h, int(n_pairs), int(row0), C.c_float(nn_thresh),
                                              C.c_float(ratio)))

    def profile_run(self, n_pairs, nn_thresh=2.0, ratio=1.0, iters=5):
        """Average ms of [dot+top2, merge, exact+rescan, compact, total] (HIP events)."""
        ms = np.zeros(5, np.float32)
        check(self.lib.vs_matcher_profile_run(self.h, int(n_pairs), C.c_float(nn_thresh),
                                              C.c_float(ratio), int(iters), ptr(ms, C.c_float)))
        return ms

    def matches(self, pair):
        n0 = max(self.sizes[int(pair)][0], 1)
        oi = np.zeros(n0, np.int32); oj = np.zeros(n0, np.int32); od = np.zeros(n0, np.float32)
        n = C.c_int32(0)
        check(self.lib.vs_matcher_get_matches(self.h, int(pair), ptr(oi, C.c_int32),
                                              ptr(oj, C.c_int32), ptr(od, C.c_float), C.byref(n)))
        return oi[:n.value].copy(), oj[:n.value].copy(), od[:n.value].copy()

    def nn(self, pair):
        n0, n1 = self.sizes[int(pair)]
        i0 = np.zeros(max(n0, 1), np.int32); i1 = np.zeros(max(n1, 1), np.int32)
        check(self.lib.vs_matcher_get_nn(self.h, int(pair), ptr(i0, C.c_int32), ptr(i1, C.c_int32)))
        return i0[:n0], i1[:n1]

    def rescans(self):
        n = C.c_int64(0)
        check(self.lib.vs_matcher_get_rescans(self.h, C.byref(n)))
        return n.value


def two_way(desc0, desc1, nn_thresh=2.0, ratio=1.0):
    """vs_match_two_way: one-shot form."""
    lib = _lib.load()
    d0, d1 = _f32(desc0).reshape(-1, DESC_DIM), _f32(desc1).reshape(-1, DESC_DIM)
    n0 = max(len(d0), 1)
    oi = np.zeros(n0, np.int32); oj = np.zeros(n0, np.int32); od = np.zeros(n0, np.float32)
    n = C.c_int32(0)
    check(lib.vs_match_two_way(ptr(d0, C.c_float), len(d0), ptr(d1, C.c_float), len(d1),
                               C.c_float(nn_thresh), C.c_float(ratio), ptr(oi, C.c_int32),
                               ptr(oj, C.c_int32), ptr(od, C.c_float), C.byref(n)))
    return oi[:n.value].copy(), oj[:n.value].copy(), od[:n.value].copy()


def stereo(desc0, xy0, desc1, xy1, nn_thresh=2.0, ratio=1.0):
    """vs_match_stereo: one-shot form."""
    lib = _lib.load()
    d0, d1 = _f32(desc0).reshape(-1, DESC_DIM), _f32(desc1).reshape(-1, DESC_DIM)
    x0, x1 = _i32(xy0).reshape(-1, 2), _i32(xy1).reshape(-1, 2)
    n0 = max(len(d0), 1)
    oi = np.zeros(n0, np.int32); oj = np.zeros(n0, np.int32); od = np.zeros(n0, np.float32)
    n = C.c_int32(0)
    check(lib.vs_match_stereo(ptr(d0, C.c_float), ptr(x0, C.c_int32), len(d0), ptr(d1, C.c_float),
                              ptr(x1, C.c_int32), len(d1), C.c_float(nn_thresh), C.c_float(ratio),
                              ptr(oi, C.c_int32), ptr(oj, C.c_int32), ptr(od, C.c_float), C.byref(n)))
    return oi[:n.value].copy(), oj[:n.value].copy(), od[:n.value].copy()


class SuperpointFrontend:
    """The matcher half of the reference class (keypoint extraction is out of scope).
    nn_thresh comes from config superpoint.nn_thresh (reference: config/config.yaml:4)."""

    def __init__(self, nn_thresh=2.0, ratio=1.0):
        self.nn_thresh_ = float(nn_thresh)
        self.ratio_ = float(ratio)

    def twoWayMatching(self, keys0, desc0, keys1, desc1):
        """Returns [(i, j, distance)], ordered by i (superpoint_frontend.cc:163-211).  As in the
        reference only the SIZES of keys0/keys1 are used."""
        d0 = _f32(desc0).reshape(-1, DESC_DIM)[:len(keys0)]
        d1 = _f32(desc1).reshape(-1, DESC_DIM)[:len(keys1)]
        i, j, d = two_way(d0, d1, self.nn_thresh_, self.ratio_)
        return list(zip(i.tolist(), j.tolist(), d.tolist()))

    def getStereoMatching(self, left_points, left_descs, right_points, right_descs):
        """superpoint_frontend.cc:110-161."""
        i, j, d = stereo(left_descs, left_points, right_descs, right_points, self.nn_thresh_,
                         self.ratio_)
        return list(zip(i.tolist(), j.tolist(), d.tolist()))
