"""Multi-GPU plumbing for the BA path: one process per GPU, landmarks sharded across ranks, the
reduced camera system summed with one all-reduce (RCCL through torch.distributed's "nccl" backend
on GPUs; "gloo" in the CPU tests).  The reference has no distributed path; the numbering and
partition rules below are the ones libvslam_hip.so applies in vs_ba_set_observations_sharded.
"""
import numpy as np


def landmark_order(cam, pt, n_points, n_cams):
    """The library's internal landmark numbering: landmarks by the first (lowest) camera that sees
    them, the caller's order kept among equals; landmarks nobody sees last.  Returns new2old."""
    first = np.full(n_points, n_cams, dtype=np.int64)
    np.minimum.at(first, np.asarray(pt), np.asarray(cam))
    return np.argsort(first, kind="stable")


def partition_landmarks(pt, n_points, world):
    """Contiguous landmark ranges balanced by observation count, for landmarks that already are in
    the internal numbering.  Returns world+1 boundaries; rank r owns [b[r], b[r+1])."""
    pt = np.asarray(pt)
    counts = np.bincount(pt, minlength=n_points).astype(np.int64)
    ptr = np.concatenate([[0], np.cumsum(counts)])
    total = int(ptr[-1])
    bounds = np.zeros(world + 1, dtype=np.int64)
    for r in range(1, world):
        target = total * r // world
        bounds[r] = min(int(np.searchsorted(ptr, target, side="left")), n_points)
    bounds[world] = n_points
    return np.maximum.accumulate(bounds)


def landmark_owner(cam, pt, n_points, n_cams, world):
    """Rank that linearises each landmark (caller's numbering): contiguous ranges of the internal
    numbering, balanced by observation count."""
    new2old = landmark_order(cam, pt, n_points, n_cams)
    old2new = np.empty(n_points, dtype=np.int64)
    old2new[new2old] = np.arange(n_points)
    b = partition_landmarks(old2new[np.asarray(pt)], n_points, world)
    owner_new = np.searchsorted(b[1:], np.arange(n_points), side="right")
    return np.minimum(owner_new, world - 1)[old2new]


def shard_mask(cam, pt, n_points, n_cams, rank, world):
    """Edges of the caller's list that rank `rank` linearises."""
    return landmark_owner(cam, pt, n_points, n_cams, world)[np.asarray(pt)] == rank


class _DevArray:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False),
                                         "version": 2}


def device_tensor(ptr, n):
    """A torch float64 CUDA tensor aliasing `n` doubles at device address `ptr` (no copy)."""
    import torch
    return torch.as_tensor(_DevArray(ptr, n), device="cuda")


def attach_allreduce(ba, group=None):
    """Makes vs_ba_lm_step / vs_ba_solve of this rank's BAProblem sum the reduced camera system
    (and the step scalars) across the process group before damping and solving."""
    import torch
    import torch.distributed as dist

    def hook(ptr, n):
        t = device_tensor(ptr, n)
        dist.all_reduce(t, group=group)
        torch.cuda.current_stream().synchronize()

    ba.set_allreduce(hook)
    return ba


def attach_allgather(ba, group=None):
    """Makes this rank's BAProblem combine the reduce buffers by sending only the segment its
    landmarks touch (vs_ba_exchange_reduced) through torch.distributed's all_gather."""
    import torch
    import torch.distributed as dist

    def hook(send_ptr, recv_ptr, n):
        world = dist.get_world_size(group)
        send = device_tensor(send_ptr, n)
        recv = device_tensor(recv_ptr, n * world)
        dist.all_gather_into_tensor(recv, send, group=group)
        torch.cuda.current_stream().synchronize()

    ba.set_allgather(hook)
    return ba


def allreduce_sum_(array, group=None):
    """In-place sum of a numpy float64 array across ranks (host path used by the CPU tests)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(array)
    dist.all_reduce(t, group=group)
    return array


def column_keys(dist_local, row0):
    """Packed (distance bits, global row) key of every column's nearest row among the rows of one
    row shard of a frame pair (dist_local: float32 n0_local x n1).  The element-wise minimum of the
    shards' keys names the column's nearest row of the whole frame, equal distances resolving to
    the smaller row (the reference's first minimum, src/core/superpoint_frontend.cc:197-203).
    Same key as the device path's column_key_kernel; no rows -> the largest key."""
    import numpy as np
    d = np.ascontiguousarray(dist_local, dtype=np.float32)
    if d.shape[0] == 0:
        return np.full(d.shape[1], np.iinfo(np.int64).max, np.int64)
    r = d.argmin(axis=0)                                     # first minimum
    bits = d[r, np.arange(d.shape[1])].view(np.uint32).astype(np.int64)
    return (bits << 32) | (r.astype(np.int64) + int(row0))


def matches_from_keys(dist_local, row0, keys, nn_thresh):
    """This shard's matches (global i, j, d) given the reduced column keys."""
    import numpy as np
    d = np.ascontiguousarray(dist_local, dtype=np.float32)
    if d.shape[0] == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32)
    j = d.argmin(axis=1)
    i = np.arange(d.shape[0])
    best = d[i, j]
    keep = ((keys[j] & 0xffffffff) == i + int(row0)) & (best < nn_thresh)
    return i[keep] + int(row0), j[keep], best[keep]
