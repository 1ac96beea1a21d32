"""Thin Python wrapper of the vs_ba_* C ABI (include/vslam_hip.h).  No arithmetic happens here."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import LMOptions, LMSummary, check, ptr


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class BAProblem:
    """One bundle-adjustment problem resident on the current HIP device."""

    def __init__(self, n_cameras, n_points, n_observations, K):
        self.lib = _lib.load()
        self.Nc, self.Np, self.No = int(n_cameras), int(n_points), int(n_observations)
        self.h = C.c_void_p()
        K = _f64(K)
        check(self.lib.vs_ba_create(C.byref(self.h), self.Nc, self.Np, C.c_int64(self.No),
                                    C.c_double(K[0]), C.c_double(K[1]), C.c_double(K[2]),
                                    C.c_double(K[3])))
        self._cb = None

    @classmethod
    def from_arrays(cls, K, t, rpy, pts, cam, pt, uv, fix_points=False, huber_a=0.0, rank=0,
                    world=1):
        t, rpy, pts = _f64(t).reshape(-1, 3), _f64(rpy).reshape(-1, 3), _f64(pts).reshape(-1, 3)
        p = cls(len(t), len(pts), len(cam), K)
        p.set_cameras(t, rpy)
        p.set_points(pts)
        p.set_fixed(fix_points)
        p.set_huber_first(huber_a)
        p.set_observations(cam, pt, uv, rank, world)
        return p

    def close(self):
        if self.h:
            self.lib.vs_ba_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        check(self.lib.vs_ba_set_stream(self.h, C.c_void_p(int(stream))))

    def set_cameras(self, t, rpy):
        t, rpy = _f64(t), _f64(rpy)
        assert t.size == 3 * self.Nc and rpy.size == 3 * self.Nc
        check(self.lib.vs_ba_set_cameras(self.h, ptr(t, C.c_double), ptr(rpy, C.c_double)))

    def set_points(self, xyz):
        xyz = _f64(xyz)
        assert xyz.size == 3 * self.Np
        check(self.lib.vs_ba_set_points(self.h, ptr(xyz, C.c_double)))

    def set_observations(self, cam, pt, uv, rank=0, world=1):
        cam, pt, uv = _i32(cam), _i32(pt), _f64(uv)
        assert cam.size == self.No and pt.size == self.No and uv.size == 2 * self.No
        check(self.lib.vs_ba_set_observations_sharded(self.h, ptr(cam, C.c_int32), ptr(pt, C.c_int32),
                                                      ptr(uv, C.c_double), int(rank), int(world)))

    def set_fixed(self, fix_points):
        check(self.lib.vs_ba_set_fixed(self.h, int(bool(fix_points))))

    def set_huber_first(self, a):
        check(self.lib.vs_ba_set_huber_first(self.h, C.c_double(a)))

    def evaluate(self, jac=True):
        cost = C.c_double(0)
        res = np.zeros((self.No, 2))
        Jc = np.zeros((self.No, 2, 6)) if jac else None
        Jp = np.zeros((self.No, 2, 3)) if jac else None
        check(self.lib.vs_ba_evaluate(self.h, C.byref(cost), ptr(res, C.c_double),
                                      ptr(Jc, C.c_double), ptr(Jp, C.c_double)))
        return cost.value, res, Jc, Jp

    def linearize(self, damp_c=None, damp_p=None):
        dc = None if damp_c is None else _f64(damp_c)
        dp = None if damp_p is None else _f64(damp_p)
        check(self.lib.vs_ba_linearize(self.h, ptr(dc, C.c_double), ptr(dp, C.c_double)))

    def profile_linearize(self, damp_c=None, damp_p=None, iters=20):
        """Average ms of [prep, landmark, diag, offdiag, finalize, total] (HIP events)."""
        dc = None if damp_c is None else _f64(damp_c)
        dp = None if damp_p is None else _f64(damp_p)
        ms = np.zeros(6, np.float32)
        check(self.lib.vs_ba_profile_linearize(self.h, ptr(dc, C.c_double), ptr(dp, C.c_double),
                                               int(iters), ptr(ms, C.c_float)))
        return ms

    def block_index(self):
        n = C.c_int64(0)
        check(self.lib.vs_ba_block_count(self.h, C.byref(n)))
        br = np.zeros(n.value, np.int32); bc = np.zeros(n.value, np.int32)
        check(self.lib.vs_ba_block_index(self.h, ptr(br, C.c_int32), ptr(bc, C.c_int32)))
        return br, bc

    def reduced_system(self):
        br, bc = self.block_index()
        S = np.zeros((len(br), 6, 6)); rhs = np.zeros(6 * self.Nc); cost = C.c_double(0)
        check(self.lib.vs_ba_get_reduced_system(self.h, ptr(S, C.c_double), ptr(rhs, C.c_double),
                                                C.byref(cost)))
        return br, bc, S, rhs, cost.value

    def reduced_system_dense(self):
        """Expands the stored upper-triangular blocks into a dense symmetric matrix (host)."""
        br, bc, S, rhs, cost = self.reduced_system()
        n = 6 * self.Nc
        D = np.zeros((n, n))
        for r, c, blk in zip(br, bc, S):
            D[6 * r:6 * r + 6, 6 * c:6 * c + 6] = blk
            if r != c:
                D[6 * c:6 * c + 6, 6 * r:6 * r + 6] = blk.T
        return D, rhs, cost

    def landmark_blocks(self):
        Vinv6 = np.zeros((self.Np, 6)); gp = np.zeros((self.Np, 3))
        check(self.lib.vs_ba_get_landmark_blocks(self.h, ptr(Vinv6, C.c_double), ptr(gp, C.c_double)))
        return Vinv6, gp

    def reduce_buffer(self):
        p = C.c_void_p(); n = C.c_int64(0)
        check(self.lib.vs_ba_reduce_buffer(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def set_allreduce(self, fn):
        """fn(device_ptr:int, n_doubles:int) -> None must sum the buffer across ranks in place."""
        def tramp(_user, buf, n):
            try:
                fn(int(buf), int(n))
                return 0
            except Exception:   # never unwind through the C ABI
                import traceback
                traceback.print_exc()
                return 1
        self._cb = _lib.ALLREDUCE_FN(tramp)
        check(self.lib.vs_ba_set_allreduce(self.h, self._cb, None))

    def set_allgather(self, fn):
        """fn(send_ptr:int, recv_ptr:int, n_doubles:int) -> None must place every rank's n doubles
        at recv + rank * n on every rank (vs_ba_exchange_reduced's transport)."""
        def tramp(_user, send, recv, n):
            try:
                fn(int(send), int(recv), int(n))
                return 0
            except Exception:   # never unwind through the C ABI
                import traceback
                traceback.print_exc()
                return 1
        self._cb_gather = _lib.ALLGATHER_FN(tramp)
        check(self.lib.vs_ba_set_allgather(self.h, self._cb_gather, None))

    def exchange_reduced(self):
        """Sum of the shards' partial reduce buffers, on every rank (after linearize)."""
        check(self.lib.vs_ba_exchange_reduced(self.h))

    def exchange_size(self):
        """(doubles this rank sends, doubles every rank receives) in exchange_reduced."""
        a = C.c_int64(0); b = C.c_int64(0)
        check(self.lib.vs_ba_exchange_size(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def lm_step(self, radius=1e4, fetch=True):
        """One LM step.  fetch=False leaves the step on the device (as inside vs_ba_solve) and
        returns only the model cost change."""
        mcc = C.c_double(0)
        if not fetch:
            check(self.lib.vs_ba_lm_step(self.h, C.c_double(radius), None, None, C.byref(mcc)))
            return None, None, mcc.value
        dcam = np.zeros((self.Nc, 6)); dpt = np.zeros((self.Np, 3))
        check(self.lib.vs_ba_lm_step(self.h, C.c_double(radius), ptr(dcam, C.c_double),
                                     ptr(dpt, C.c_double), C.byref(mcc)))
        return dcam, dpt, mcc.value

    def solve(self, **opts):
        o = LMOptions()
        self.lib.vs_lm_default_options(C.byref(o))
        for k, v in opts.items():
            setattr(o, k, v)
        s = LMSummary()
        check(self.lib.vs_ba_solve(self.h, C.byref(o), C.byref(s)))
        return dict(iterations=s.iterations, successful_steps=s.successful_steps,
                    termination=s.termination, initial_cost=s.initial_cost,
                    final_cost=s.final_cost, radius=s.final_radius)

    def cameras(self):
        t = np.zeros((self.Nc, 3)); rpy = np.zeros((self.Nc, 3))
        check(self.lib.vs_ba_get_cameras(self.h, ptr(t, C.c_double), ptr(rpy, C.c_double)))
        return t, rpy

    def points(self):
        xyz = np.zeros((self.Np, 3))
        check(self.lib.vs_ba_get_points(self.h, ptr(xyz, C.c_double)))
        return xyz
