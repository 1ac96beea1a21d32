"""Seeded synthetic inputs of the shapes the benchmarks and parity tests use.

Bundle-adjustment scenes follow the reference's camera model (camera looks along +Y, pose =
translation + roll/pitch/yaw with R = Rz*Ry*Rx, Pc = R*Pw + t; /root/reference paths:
src/core/reprojection_error.h:32-48) and its intrinsics (config/config.yaml:12-15, stored as float).
Keypoints are rounded to integer pixels as the reference's cv::Point keys are
(src/core/optimizer.cc:30-36).  Descriptors are L2-normalised 256-d fp32 rows like SuperPoint's
(src/core/superpoint_frontend.cc:97-107).
"""
import numpy as np

FX, FY, CX, CY = (float(np.float32(v)) for v in (170.5, 170.5, 191.5, 103.4))
DESC_DIM = 256


def rpy_to_R(rpy):
    """R = Rz(yaw) Ry(pitch) Rx(roll) for an (..., 3) array of angles."""
    rpy = np.asarray(rpy, dtype=np.float64)
    sr, cr = np.sin(rpy[..., 0]), np.cos(rpy[..., 0])
    sp, cp = np.sin(rpy[..., 1]), np.cos(rpy[..., 1])
    sy, cy = np.sin(rpy[..., 2]), np.cos(rpy[..., 2])
    R = np.empty(rpy.shape[:-1] + (3, 3))
    R[..., 0, 0] = cy * cp; R[..., 0, 1] = cy * sp * sr - sy * cr; R[..., 0, 2] = cy * sp * cr + sy * sr
    R[..., 1, 0] = sy * cp; R[..., 1, 1] = sy * sp * sr + cy * cr; R[..., 1, 2] = sy * sp * cr - cy * sr
    R[..., 2, 0] = -sp; R[..., 2, 1] = cp * sr; R[..., 2, 2] = cp * cr
    return R


def project(pts, t, rpy, K=(FX, FY, CX, CY)):
    """Pixel coordinates of points (N,3) seen from poses t (N,3), rpy (N,3); also returns depth."""
    R = rpy_to_R(rpy)
    pc = np.einsum("nij,nj->ni", R, pts) + t
    u = K[0] * pc[:, 0] / pc[:, 1] + K[2]
    v = -K[1] * pc[:, 2] / pc[:, 1] + K[3]
    return np.stack([u, v], axis=1), pc[:, 1]


class BAScene:
    """cam/pt/uv observation lists plus ground-truth and perturbed initial parameters."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def Nc(self):
        return len(self.t0)

    @property
    def Np(self):
        return len(self.pts0)

    @property
    def No(self):
        return len(self.cam)


def make_ba_scene(Nc, Np, k_mean, band=None, seed=0, k_jitter=2, cam_step=0.05, noise_px=0.5,
                  pose_noise=(0.01, np.deg2rad(0.5)), pt_noise=0.01, chunk=200_000):
    """Cameras on a sideways trajectory looking along +Y; each landmark is observed by about
    `k_mean` distinct cameras drawn from a co-visibility band of `band` consecutive cameras.
    Observations are emitted landmark-major (landmark ascending, camera ascending within one)."""
    rng = np.random.default_rng(0xBA5EED + seed)
    band = Nc if band is None else min(band, Nc)
    k_lo = max(2, k_mean - k_jitter)
    k_hi = min(band, k_mean + k_jitter)
    # ground-truth cameras
    cw = np.zeros((Nc, 3))
    cw[:, 0] = np.arange(Nc) * cam_step
    cw[:, 1] = rng.normal(0, 0.02, Nc)
    cw[:, 2] = rng.normal(0, 0.02, Nc)
    rpy_gt = rng.normal(0, np.deg2rad(2.0), (Nc, 3))
    R = rpy_to_R(rpy_gt)
    t_gt = -np.einsum("nij,nj->ni", R, cw)
    # landmarks: in front of the band centre
    first = rng.integers(0, Nc - band + 1, Np)
    centre_x = (first + 0.5 * (band - 1)) * cam_step
    depth = rng.uniform(1.0, 20.0, Np)
    half_w = np.maximum(0.8 * depth * (CX / FX) - 0.5 * band * cam_step, 0.1 * depth)
    pts_gt = np.empty((Np, 3))
    pts_gt[:, 0] = centre_x + rng.uniform(-1, 1, Np) * half_w
    pts_gt[:, 1] = depth
    pts_gt[:, 2] = rng.uniform(-1, 1, Np) * 0.8 * depth * (CY / FY)
    # pick k distinct cameras per landmark from its band
    kk = rng.integers(k_lo, k_hi + 1, Np)
    cams, pts_idx = [], []
    for s in range(0, Np, chunk):
        e = min(Np, s + chunk)
        keys = rng.random((e - s, band))
        order = np.argsort(keys, axis=1)[:, :k_hi]           # random distinct offsets
        sel = np.arange(k_hi)[None, :] < kk[s:e, None]
        off = np.where(sel, order, band)                       # pad with sentinel
        off.sort(axis=1)                                       # camera ascending, sentinel last
        rows = np.repeat(np.arange(s, e), k_hi).reshape(e - s, k_hi)
        m = off < band
        cams.append((first[s:e, None] + off)[m])
        pts_idx.append(rows[m])
    cam = np.concatenate(cams).astype(np.int32)
    pt = np.concatenate(pts_idx).astype(np.int32)
    uv, _ = project(pts_gt[pt], t_gt[cam], rpy_gt[cam])
    uv = np.rint(uv + rng.normal(0, noise_px, uv.shape))
    # the initial pose is the true one moved by pose_noise[0] and turned by pose_noise[1] ABOUT THE
    # CAMERA CENTRE (t = -R c: turning R while keeping t would swing a camera that is hundreds of
    # metres from the origin by metres)
    rpy0 = rpy_gt + rng.normal(0, pose_noise[1], rpy_gt.shape)
    t0 = -np.einsum("nij,nj->ni", rpy_to_R(rpy0), cw + rng.normal(0, pose_noise[0], cw.shape))
    pts0 = pts_gt + rng.normal(0, pt_noise, pts_gt.shape)
    return BAScene(cam=cam, pt=pt, uv=uv, t0=t0, rpy0=rpy0, pts0=pts0, t_gt=t_gt, rpy_gt=rpy_gt,
                   pts_gt=pts_gt, K=np.array([FX, FY, CX, CY]))


def landmarks_in_keyframe_order(sc):
    """The same scene with the landmarks renumbered by the first camera that sees them (the order
    in which the reference creates tracked points); observations stay landmark-major."""
    first = np.full(sc.Np, np.iinfo(np.int64).max)
    np.minimum.at(first, sc.pt, sc.cam)
    order = np.argsort(first, kind="stable")                 # new id -> old id
    rank = np.empty_like(order)
    rank[order] = np.arange(sc.Np)
    pt = rank[sc.pt]
    o = np.lexsort((sc.cam, pt))
    return BAScene(cam=sc.cam[o], pt=pt[o].astype(np.int32), uv=sc.uv[o], t0=sc.t0, rpy0=sc.rpy0,
                   pts0=sc.pts0[order], t_gt=sc.t_gt, rpy_gt=sc.rpy_gt, pts_gt=sc.pts_gt[order], K=sc.K)


BA_CONFIGS = {
    # name: (Nc, Np, k_mean, band)   -- BASELINE.json configs[1], [3], one 1/8 shard of [4]
    "c1": (10, 500, 6, 10),
    "c2": (50, 20_000, 10, 50),
    "c4": (1_000, 500_000, 4, 16),
    "c5shard": (10_000, 500_000, 5, 32),
    "c5": (10_000, 4_000_000, 5, 32),          # configs[4] whole, on one GPU
}


def make_ba_config(name, seed=0):
    Nc, Np, k, band = BA_CONFIGS[name]
    return make_ba_scene(Nc, Np, k, band=band, seed=seed)


def make_tracks_scene(n_frames=10, n_kpts=500, seed=0):
    """A window in the reference's own containers: per-frame integer keypoints, a dense
    frames x points track table (-1 = not seen) and float poses/points, as Tracking hands them to
    Optimizer (src/core/tracking.cc:85-99)."""
    sc = make_ba_scene(n_frames, n_kpts, 6, band=n_frames, seed=seed)
    rng = np.random.default_rng(1234 + seed)
    tracks = -np.ones((n_frames, sc.Np), np.int32)
    keys = []
    for f in range(n_frames):
        obs = np.nonzero(sc.cam == f)[0]
        perm = rng.permutation(len(obs))
        xy = sc.uv[obs][perm].astype(np.int32)
        extra = rng.integers(0, 200, (n_kpts - len(obs), 2)).astype(np.int32) if len(obs) < n_kpts else np.zeros((0, 2), np.int32)
        keys.append(np.concatenate([xy, extra], axis=0))
        tracks[f, sc.pt[obs][perm]] = np.arange(len(obs), dtype=np.int32)
    return sc, tracks, keys


def make_descriptors(n0=4096, n1=4096, seed=0, sigma=0.3, unmatched=0.3, duplicates=0):
    """desc0: random unit rows; desc1: a permutation of noisy copies of desc0 (inlier rows) with
    a fraction of unrelated rows.  `duplicates` appends exact duplicate rows to force ties."""
    rng = np.random.default_rng(0xDE5C + seed)
    d0 = rng.standard_normal((n0, DESC_DIM)).astype(np.float32)
    d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    n_in = min(n0, int(round(n1 * (1.0 - unmatched))))
    src = rng.permutation(n0)[:n_in]
    d1 = np.empty((n1, DESC_DIM), np.float32)
    d1[:n_in] = d0[src] + sigma / np.sqrt(DESC_DIM) * rng.standard_normal((n_in, DESC_DIM)).astype(np.float32)
    d1[n_in:] = rng.standard_normal((n1 - n_in, DESC_DIM)).astype(np.float32)
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    d1 = d1[rng.permutation(n1)]
    for _ in range(duplicates):
        a, b = rng.integers(0, n1, 2)
        d1[b] = d1[a]
        a, b = rng.integers(0, n0, 2)
        d0[b] = d0[a]
    return np.ascontiguousarray(d0), np.ascontiguousarray(d1.astype(np.float32))


def make_keypoints(n, seed=0, width=383, height=207):
    rng = np.random.default_rng(0x4B50 + seed)
    return np.stack([rng.integers(0, width, n), rng.integers(0, height, n)], axis=1).astype(np.int32)


def _quat_from_rpy(rpy):
    """Unit quaternion (w, x, y, z) of Rz(yaw) * Ry(pitch) * Rx(roll)."""
    r, p, y = [0.5 * float(a) for a in rpy]
    cr, sr, cp, sp, cy, sy = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    return np.array([cy * cp * cr + sy * sp * sr, cy * cp * sr - sy * sp * cr,
                     cy * sp * cr + sy * cp * sr, sy * cp * cr - cy * sp * sr])


def _quat_R(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make_tracking_frame(n_tp=500, n_keys=600, seed=0, rivals=0.15, pixel_noise=2.0, desc_sigma=0.25):
    """One frame as Tracking::searchProjection sees it (src/core/tracking.cc:453-496): a float
    pose Tcw (7 floats: t, quaternion wxyz), float intrinsics, tracked world points with
    descriptors, and integer keypoints with descriptors.  Most keypoints are noisy reprojections
    of tracked points; `rivals` of the tracked points are near-copies of another tracked point
    (same place, similar descriptor), so that several points compete for one keypoint."""
    rng = np.random.default_rng(0x7AC + seed)
    K4 = np.array([170.5, 170.5, 191.5, 103.4], np.float32)
    q = _quat_from_rpy(rng.normal(0, 0.05, 3))
    t = rng.normal(0, 0.5, 3)
    R = _quat_R(q)
    depth = rng.uniform(1.0, 20.0, n_tp)
    u = rng.uniform(-20, 2 * K4[2] + 20, n_tp)          # a few land outside the image
    v = rng.uniform(-10, 2 * K4[3] + 10, n_tp)
    Pc = np.stack([(u - K4[2]) / K4[0] * depth, depth, -(v - K4[3]) / K4[1] * depth], axis=1)
    Pw = (Pc - t) @ R                                  # R^T (Pc - t)
    desc = rng.standard_normal((n_tp, DESC_DIM))
    desc /= np.linalg.norm(desc, axis=1, keepdims=True)
    n_riv = int(rivals * n_tp)
    for k in range(n_riv):
        a, b = rng.integers(0, n_tp, 2)
        Pw[b] = Pw[a] + rng.normal(0, 0.002, 3)
        desc[b] = desc[a] + 0.05 / np.sqrt(DESC_DIM) * rng.standard_normal(DESC_DIM)
        desc[b] /= np.linalg.norm(desc[b])
        u[b], v[b] = u[a], v[a]
    n_match = min(n_keys, int(0.8 * n_tp))
    src = rng.permutation(n_tp)[:n_match]
    key_xy = np.empty((n_keys, 2), np.int32)
    key_desc = rng.standard_normal((n_keys, DESC_DIM))
    key_xy[:n_match, 0] = np.round(u[src] + rng.normal(0, pixel_noise, n_match))
    key_xy[:n_match, 1] = np.round(v[src] + rng.normal(0, pixel_noise, n_match))
    key_desc[:n_match] = desc[src] + desc_sigma / np.sqrt(DESC_DIM) * rng.standard_normal((n_match, DESC_DIM))
    key_xy[n_match:, 0] = rng.integers(0, int(2 * K4[2]), n_keys - n_match)
    key_xy[n_match:, 1] = rng.integers(0, int(2 * K4[3]), n_keys - n_match)
    key_desc /= np.linalg.norm(key_desc, axis=1, keepdims=True)
    perm = rng.permutation(n_keys)
    pose = np.concatenate([t, q]).astype(np.float32)
    return (pose, K4, np.ascontiguousarray(Pw.astype(np.float32)), np.ascontiguousarray(desc.astype(np.float32)),
            np.ascontiguousarray(key_xy[perm]), np.ascontiguousarray(key_desc[perm].astype(np.float32)))


def make_stereo_frame(n_left=400, n_right=420, seed=0, baseline_fx=0.12 * 170.5):
    """Left/right keypoints and descriptors as Frame::computePoint3d sees them
    (src/core/frame.cc:45-86): right keypoints are the left ones shifted by a disparity (some
    <= 1, some negative), rows within a few pixels, descriptors noisy copies, plus distractors."""
    rng = np.random.default_rng(0x57E + seed)
    xy0 = make_keypoints(n_left, seed=seed + 100)
    d0 = rng.standard_normal((n_left, DESC_DIM))
    d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    n_m = min(n_right, int(0.8 * n_left))
    src = rng.permutation(n_left)[:n_m]
    disp = np.round(rng.uniform(-3, 60, n_m)).astype(np.int32)
    xy1 = np.empty((n_right, 2), np.int32)
    xy1[:n_m, 0] = xy0[src, 0] - disp
    xy1[:n_m, 1] = xy0[src, 1] + rng.integers(-2, 3, n_m)
    xy1[n_m:] = make_keypoints(n_right - n_m, seed=seed + 200)
    d1 = rng.standard_normal((n_right, DESC_DIM))
    d1[:n_m] = d0[src] + 0.3 / np.sqrt(DESC_DIM) * rng.standard_normal((n_m, DESC_DIM))
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    perm = rng.permutation(n_right)
    q = _quat_from_rpy(rng.normal(0, 0.2, 3))
    pose = np.concatenate([rng.normal(0, 1.0, 3), q]).astype(np.float32)
    K4 = np.array([170.5, 170.5, 191.5, 103.4], np.float32)
    return (np.ascontiguousarray(d0.astype(np.float32)), xy0, np.ascontiguousarray(d1[perm].astype(np.float32)),
            np.ascontiguousarray(xy1[perm]), K4, np.float32(baseline_fx), pose)


def make_two_view_tracks(n_tp=500, seed=0, baseline=1.5, pixel_noise=0.5):
    """Two frames as Tracking::updateTrackedPoints sees them (src/core/tracking.cc:378-451): float
    poses of the oldest and newest frame of the window, their integer keypoints, the per-point
    track tables (keypoint index or -1) and the current world points.  Most points have usable
    parallax; some are far away (parallax cosine above 0.9998), some lie behind a camera or
    closer than the 0.2 depth gate after triangulation, some are untracked in one frame."""
    rng = np.random.default_rng(0xD17 + seed)
    K4 = np.array([170.5, 170.5, 191.5, 103.4], np.float32)
    q1 = _quat_from_rpy(rng.normal(0, 0.03, 3)); t1 = rng.normal(0, 0.3, 3)
    q2 = _quat_from_rpy(rng.normal(0, 0.08, 3)); t2 = t1 + np.array([baseline, 0.2, 0.1]) + rng.normal(0, 0.05, 3)
    R1, R2 = _quat_R(q1), _quat_R(q2)
    depth = rng.uniform(1.0, 15.0, n_tp)
    far = rng.random(n_tp) < 0.1
    depth[far] = rng.uniform(300.0, 3000.0, far.sum())            # negligible parallax
    u = rng.uniform(20, 2 * K4[2] - 20, n_tp)
    v = rng.uniform(10, 2 * K4[3] - 10, n_tp)
    Pc1 = np.stack([(u - K4[2]) / K4[0] * depth, depth, -(v - K4[3]) / K4[1] * depth], axis=1)
    Pw = (Pc1 - t1) @ R1
    Pc2 = Pw @ R2.T + t2

    def pix(Pc):
        x = K4[0] * Pc[:, 0] / Pc[:, 1] + K4[2]
        y = -K4[1] * Pc[:, 2] / Pc[:, 1] + K4[3]
        return np.stack([x, y], axis=1)

    k1 = np.round(pix(Pc1) + rng.normal(0, pixel_noise, (n_tp, 2))).astype(np.int32)
    k2 = np.round(pix(Pc2) + rng.normal(0, pixel_noise, (n_tp, 2))).astype(np.int32)
    wild = rng.random(n_tp) < 0.1                                # mismatches: rays that diverge or
    k2[wild] += rng.integers(-150, 151, (int(wild.sum()), 2)).astype(np.int32)   # meet behind a camera
    p1, p2 = rng.permutation(n_tp), rng.permutation(n_tp)
    keys1 = np.empty_like(k1); keys1[p1] = k1
    keys2 = np.empty_like(k2); keys2[p2] = k2
    tracks1, tracks2 = p1.astype(np.int32), p2.astype(np.int32)
    tracks1[rng.random(n_tp) < 0.1] = -1
    tracks2[rng.random(n_tp) < 0.1] = -1
    pose1 = np.concatenate([t1, q1]).astype(np.float32)
    pose2 = np.concatenate([t2, q2]).astype(np.float32)
    xyz0 = (Pw + rng.normal(0, 0.05, Pw.shape)).astype(np.float32)
    return pose1, pose2, K4, keys1, tracks1, keys2, tracks2, np.ascontiguousarray(xyz0)
