"""Python wrappers of the per-frame tracking helpers of the C ABI (reference:
src/core/tracking.cc:70-78, 346-377, 378-451, 453-496; src/core/frame.cc:45-86)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, ptr

DESC_DIM = 256


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def pose7(translation, quaternion_wxyz):
    """transform::Rigid3f as the 7 floats the C ABI takes."""
    return _f32(np.concatenate([np.asarray(translation, np.float32).ravel(),
                                np.asarray(quaternion_wxyz, np.float32).ravel()]))


def reproject_points(pose, K4, xyz):
    """Tracking::reprojection for an (n, 3) array of world points -> (n, 2) int32 pixels."""
    lib = _lib.load()
    pose, K4, xyz = _f32(pose), _f32(K4), _f32(xyz).reshape(-1, 3)
    out = np.zeros((len(xyz), 2), np.int32)
    check(lib.vs_reproject_points(ptr(pose, C.c_float), ptr(K4, C.c_float), ptr(xyz, C.c_float),
                                  len(xyz), ptr(out, C.c_int32)))
    return out


def search_projection(pose, K4, tp_xyz, tp_desc, key_xy, key_desc, radius=30.0, accept=0.8,
                      claimed=None):
    """Tracking::searchProjection.  Returns (tracks, claimed)."""
    lib = _lib.load()
    pose, K4 = _f32(pose), _f32(K4)
    tp_xyz, tp_desc = _f32(tp_xyz).reshape(-1, 3), _f32(tp_desc).reshape(-1, DESC_DIM)
    key_xy, key_desc = _i32(key_xy).reshape(-1, 2), _f32(key_desc).reshape(-1, DESC_DIM)
    claimed = (np.zeros(len(key_xy), np.uint8) if claimed is None
               else np.ascontiguousarray(claimed, np.uint8).copy())
    tracks = np.full(len(tp_xyz), -1, np.int32)
    check(lib.vs_search_projection(ptr(pose, C.c_float), ptr(K4, C.c_float), ptr(tp_xyz, C.c_float),
                                   ptr(tp_desc, C.c_float), len(tp_xyz), ptr(key_xy, C.c_int32),
                                   ptr(key_desc, C.c_float), len(key_xy), C.c_float(radius),
                                   C.c_double(accept), ptr(claimed, C.c_uint8), ptr(tracks, C.c_int32)))
    return tracks, claimed


def remove_outliers(pose, K4, tp_xyz, key_xy, tracks, max_dist=20.0):
    """Tracking::removeOutlier for one frame.  Returns (tracks, n_removed)."""
    lib = _lib.load()
    pose, K4 = _f32(pose), _f32(K4)
    tp_xyz, key_xy = _f32(tp_xyz).reshape(-1, 3), _i32(key_xy).reshape(-1, 2)
    tracks = _i32(tracks).copy()
    n = C.c_int32(0)
    check(lib.vs_remove_outliers(ptr(pose, C.c_float), ptr(K4, C.c_float), ptr(tp_xyz, C.c_float),
                                 len(tp_xyz), ptr(key_xy, C.c_int32), len(key_xy),
                                 C.c_double(max_dist), ptr(tracks, C.c_int32), C.byref(n)))
    return tracks, n.value


def compute_points3d(desc0, xy0, desc1, xy1, K4, baseline_fx, pose):
    """Frame::computePoint3d for every left keypoint.  Returns (good, xyz, minloc)."""
    lib = _lib.load()
    d0, d1 = _f32(desc0).reshape(-1, DESC_DIM), _f32(desc1).reshape(-1, DESC_DIM)
    x0, x1 = _i32(xy0).reshape(-1, 2), _i32(xy1).reshape(-1, 2)
    pose, K4 = _f32(pose), _f32(K4)
    good = np.zeros(len(d0), np.uint8)
    xyz = np.zeros((len(d0), 3), np.float32)
    loc = np.full(len(d0), -1, np.int32)
    check(lib.vs_compute_points3d(ptr(d0, C.c_float), ptr(x0, C.c_int32), len(d0), ptr(d1, C.c_float),
                                  ptr(x1, C.c_int32), len(d1), ptr(K4, C.c_float),
                                  C.c_float(baseline_fx), ptr(pose, C.c_float), ptr(good, C.c_uint8),
                                  ptr(xyz, C.c_float), ptr(loc, C.c_int32)))
    return good, xyz, loc


def triangulate_tracks(pose1, pose2, K4, key_xy1, tracks1, key_xy2, tracks2, xyz):
    """Tracking::updateTrackedPoints.  Returns (xyz, updated, n_updated)."""
    lib = _lib.load()
    pose1, pose2, K4 = _f32(pose1), _f32(pose2), _f32(K4)
    k1, k2 = _i32(key_xy1).reshape(-1, 2), _i32(key_xy2).reshape(-1, 2)
    t1, t2 = _i32(tracks1), _i32(tracks2)
    out = _f32(xyz).reshape(-1, 3).copy()
    upd = np.zeros(len(out), np.uint8)
    n = C.c_int32(0)
    check(lib.vs_triangulate_tracks(ptr(pose1, C.c_float), ptr(pose2, C.c_float), ptr(K4, C.c_float),
                                    ptr(k1, C.c_int32), len(k1), ptr(t1, C.c_int32),
                                    ptr(k2, C.c_int32), len(k2), ptr(t2, C.c_int32), len(out),
                                    ptr(out, C.c_float), ptr(upd, C.c_uint8), C.byref(n)))
    return out, upd, n.value
