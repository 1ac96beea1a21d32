"""Host-side mirror of the reference's core::Optimizer (reference: src/core/optimizer.h:24-36,
src/core/optimizer.cc) whose solve() -- an empty stub in the reference (optimizer.cc:47-49) -- runs
the MI355X path through the C ABI.  Method names, argument meaning and quirks follow the reference:

  * poses are Rigid3f = (translation float[3], quaternion (w,x,y,z) float[4]); FromPose stores
    translation + eulerAngles(0,1,2) of the rotation matrix (optimizer.cc:38-45), getPose rebuilds
    q = qz(r2)*qy(r1)*qx(r0) (optimizer.cc:51-59) -- the two conventions differ in the reference
    itself and are kept as they are;
  * addReprojectionEdges skips the LAST point (optimizer.cc:102), returns without adding anything
    for fewer than two frames (:62-63), attaches HuberLoss(100) to the first edge only (:69,128),
    holds keypoints constant (:133) and points too when fix_camera_or_point is true (:134-137).
"""
import numpy as np

from .ba import BAProblem

f32 = np.float32


class Rigid3f:
    """translation (3,) float32 + unit quaternion (w, x, y, z) float32
    (reference: src/transform/rigid_transform.h:31-104)."""

    def __init__(self, translation=(0, 0, 0), rotation=(1, 0, 0, 0)):
        self.translation = np.asarray(translation, dtype=f32).reshape(3)
        self.rotation = np.asarray(rotation, dtype=f32).reshape(4)


class TrackedPoint:
    """Only the field the optimizer reads (reference: src/core/tracked_point.h:24)."""

    def __init__(self, Pw):
        self.Pw = np.asarray(Pw, dtype=f32).reshape(3)


def quat_to_matrix_f32(q):
    """Eigen::Quaternionf::toRotationMatrix()."""
    w, x, y, z = (f32(v) for v in q)
    tx, ty, tz = f32(2) * x, f32(2) * y, f32(2) * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = f32(1)
    return np.array([[one - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, one - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, one - (txx + tyy)]], dtype=f32)


def euler_angles_012_f32(R):
    """Eigen::Matrix3f::eulerAngles(0, 1, 2) (Eigen 3.3 EulerAngles.h)."""
    R = np.asarray(R, dtype=f32)
    r0 = f32(np.arctan2(R[1, 2], R[2, 2]))
    c2 = f32(np.sqrt(R[0, 0] * R[0, 0] + R[0, 1] * R[0, 1]))
    if r0 > 0:
        r0 = f32(r0 - f32(np.pi))
        r1 = f32(np.arctan2(-R[0, 2], -c2))
    else:
        r1 = f32(np.arctan2(-R[0, 2], c2))
    s1, c1 = f32(np.sin(r0)), f32(np.cos(r0))
    r2 = f32(np.arctan2(s1 * R[2, 0] - c1 * R[1, 0], c1 * R[1, 1] - s1 * R[2, 1]))
    return np.array([-r0, -r1, -r2], dtype=f32)


def _qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx], dtype=f32)


def rpy_to_quat_f32(rpy):
    """yaw_angle * pitch_angle * roll_angle with float AngleAxis (optimizer.cc:54-57)."""
    r, p, y = (f32(v) for v in rpy)
    h = f32(0.5)
    qx = np.array([np.cos(r * h), np.sin(r * h), 0, 0], dtype=f32)
    qy = np.array([np.cos(p * h), 0, np.sin(p * h), 0], dtype=f32)
    qz = np.array([np.cos(y * h), 0, 0, np.sin(y * h)], dtype=f32)
    return _qmul(_qmul(qz, qy), qx)


class Problem:
    """Stands in for the ceres::Problem argument of addReprojectionEdges: it owns the device-side
    problem the edges are added to."""

    def __init__(self):
        self.ba = None


class Optimizer:
    HUBER_A = 100.0   # optimizer.cc:69

    def __init__(self):
        self.poses_ = []     # list of (translation double[3], rotation double[3])
        self.points_ = []    # list of double[3]
        self.keys_ = []      # per frame (n,2) double
        self._problem = None

    # -- optimizer.cc:14-17, 38-45
    def addPose(self, pose):
        rpy = euler_angles_012_f32(quat_to_matrix_f32(pose.rotation))
        self.poses_.append((pose.translation.astype(np.float64), rpy.astype(np.float64)))

    # -- optimizer.cc:19-28
    def addPoints(self, points):
        for p in points:
            self.points_.append(np.asarray(p.Pw, dtype=f32).astype(np.float64))

    # -- optimizer.cc:30-36
    def addKeys(self, points):
        k = np.asarray(points, dtype=np.int32).reshape(-1, 2)
        self.keys_.append(k.astype(np.float64))

    # -- optimizer.cc:60-145
    def addReprojectionEdges(self, problem, K, tracks, fix_camera_or_point):
        tracks = [np.asarray(t, dtype=np.int32) for t in tracks]
        if len(tracks) < 2:
            return
        if len(self.points_) == 0:
            raise ValueError("addReprojectionEdges with no points (undefined behaviour in the "
                             "reference: size_t underflow at optimizer.cc:102)")
        K = np.asarray(K, dtype=f32)
        n_pts = len(self.points_)
        tab = np.stack([t[:n_pts] for t in tracks], axis=0)          # frames x points
        pt, cam = np.nonzero(tab[:, :n_pts - 1].T != -1)             # point-major, frame ascending
        key = tab[cam, pt]
        uv = np.zeros((len(cam), 2))
        for f in np.unique(cam):
            m = cam == f
            uv[m] = self.keys_[f][key[m]]
        t = np.array([p[0] for p in self.poses_]); rpy = np.array([p[1] for p in self.poses_])
        K4 = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]], dtype=np.float64)
        problem.ba = BAProblem.from_arrays(K4, t, rpy, np.array(self.points_), cam.astype(np.int32),
                                           pt.astype(np.int32), uv,
                                           fix_points=bool(fix_camera_or_point),
                                           huber_a=self.HUBER_A)
        self._problem = problem

    # -- optimizer.cc:47-49 (empty in the reference): the ceres::Solve call of tracking.cc:101-110
    def solve(self, **options):
        if self._problem is None or self._problem.ba is None:
            return None
        ba = self._problem.ba
        summary = ba.solve(**options)
        t, rpy = ba.cameras()
        self.poses_ = [(t[i].copy(), rpy[i].copy()) for i in range(len(t))]
        pts = ba.points()
        self.points_ = [pts[i].copy() for i in range(len(pts))]
        return summary

    # -- optimizer.cc:51-59
    def getPose(self, id):
        t, rpy = self.poses_[id]
        return Rigid3f(t.astype(f32), rpy_to_quat_f32(rpy.astype(f32)))

    # -- optimizer.cc:187-190
    def getPoints(self):
        return self.points_
