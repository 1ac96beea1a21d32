"""Generates the committed golden fixtures under tests/golden/ (run once, in the build container).

The reference holds no tests or golden vectors and cannot be built here, so these fixtures pin the
oracle (and through it the HIP path) against INDEPENDENT re-derivations, none of which share code
with oracle/*.c:
  * residuals: the reference functor (src/core/reprojection_error.h:22-51) restated in torch fp64
    with Eigen's quaternion formulas; Jacobians by torch autograd of that restatement;
  * normal equations, Jacobi scaling, LM damping, the dense LM step and the Schur complement:
    numpy dense linear algebra on those autograd Jacobians;
  * matchers: numpy brute force with a sequential fp64 accumulation (np.cumsum), narrowed to
    float32, clamp, sqrt, first-minimum argmin (superpoint_frontend.cc:163-211, 110-161).
Usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from visual_slam_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_default_dtype(torch.float64)


def qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return (aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
            aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx)


def functor(point, uv, t, rpy, K):
    """reprojection_error.h:22-51 for batched inputs (N,3),(N,2),(N,3),(N,3)."""
    z = torch.zeros_like(rpy[:, 0])
    qx = (torch.cos(rpy[:, 0] / 2), torch.sin(rpy[:, 0] / 2), z, z)
    qy = (torch.cos(rpy[:, 1] / 2), z, torch.sin(rpy[:, 1] / 2), z)
    qz = (torch.cos(rpy[:, 2] / 2), z, z, torch.sin(rpy[:, 2] / 2))
    w, x, y, zq = qmul(qmul(qz, qy), qx)
    qv = torch.stack([x, y, zq], dim=1)
    uvv = 2.0 * torch.cross(qv, point, dim=1)
    pc = point + w[:, None] * uvv + torch.cross(qv, uvv, dim=1) + t
    ipx = pc[:, 0] / pc[:, 1]
    ipy = -pc[:, 2] / pc[:, 1]
    rx = K[0] * ipx + K[2]
    ry = K[1] * ipy + K[3]
    return torch.stack([rx - uv[:, 0], ry - uv[:, 1]], dim=1)


def autograd_problem(K, t, rpy, pts, cam, pt, uv):
    cam_t = torch.as_tensor(cam, dtype=torch.long)
    pt_t = torch.as_tensor(pt, dtype=torch.long)
    P = torch.tensor(pts[pt], requires_grad=True)
    T = torch.tensor(t[cam], requires_grad=True)
    Rr = torch.tensor(rpy[cam], requires_grad=True)
    UV = torch.tensor(uv)
    Kt = torch.tensor(K)
    res = functor(P, UV, T, Rr, Kt)
    Jp = np.zeros((len(cam), 2, 3)); Jt = np.zeros((len(cam), 2, 3)); Jr = np.zeros((len(cam), 2, 3))
    for r in range(2):
        g = torch.autograd.grad(res[:, r].sum(), [P, T, Rr], retain_graph=True)
        Jp[:, r] = g[0].numpy(); Jt[:, r] = g[1].numpy(); Jr[:, r] = g[2].numpy()
    del cam_t, pt_t
    return res.detach().numpy(), np.concatenate([Jt, Jr], axis=2), Jp


def dense_system(Nc, Np, cam, pt, res, Jc, Jp):
    n = 6 * Nc + 3 * Np
    J = np.zeros((2 * len(cam), n))
    for o in range(len(cam)):
        J[2 * o:2 * o + 2, 6 * cam[o]:6 * cam[o] + 6] = Jc[o]
        J[2 * o:2 * o + 2, 6 * Nc + 3 * pt[o]:6 * Nc + 3 * pt[o] + 3] = Jp[o]
    r = res.reshape(-1)
    return J, J.T @ J, J.T @ r


def make_ba():
    sc = synth.make_ba_scene(6, 120, 4, band=6, seed=7)
    K = sc.K.copy()
    res, Jc, Jp = autograd_problem(K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    Nc, Np = sc.Nc, sc.Np
    J, H, g = dense_system(Nc, Np, sc.cam, sc.pt, res, Jc, Jp)
    n = H.shape[0]
    used = np.zeros(n, bool)
    for c in np.unique(sc.cam):
        used[6 * c:6 * c + 6] = True
    for j in np.unique(sc.pt):
        used[6 * Nc + 3 * j:6 * Nc + 3 * j + 3] = True
    hd = np.diag(H).copy()
    scale = 1.0 / (1.0 + np.sqrt(hd))
    radius = 1e4
    lam = np.clip(scale ** 2 * hd, 1e-6, 1e32) / radius / scale ** 2
    Hd = H + np.diag(lam)
    idx = np.nonzero(used)[0]
    delta = np.zeros(n)
    delta[idx] = np.linalg.solve(Hd[np.ix_(idx, idx)], -g[idx])
    m = J @ delta
    mcc = -(m @ (res.reshape(-1) + 0.5 * m))
    # Schur complement of the damped system
    nc = 6 * Nc
    U, W, V = Hd[:nc, :nc], Hd[:nc, nc:], Hd[nc:, nc:]
    pidx = np.nonzero(used[nc:])[0]
    Vinv = np.zeros_like(V)
    Vinv[np.ix_(pidx, pidx)] = np.linalg.inv(V[np.ix_(pidx, pidx)])
    S = U - W @ Vinv @ W.T
    rhs = -g[:nc] + W @ Vinv @ g[nc:]
    # motion-only step (points fixed)
    Hc = H[:nc, :nc] + np.diag(lam[:nc])
    dcam_fix = np.linalg.solve(Hc, -g[:nc])
    np.savez_compressed(
        os.path.join(HERE, "ba_small.npz"), K=K, t=sc.t0, rpy=sc.rpy0, pts=sc.pts0, cam=sc.cam,
        pt=sc.pt, uv=sc.uv, res=res, Jc=Jc, Jp=Jp, cost=0.5 * float((res ** 2).sum()), lam=lam,
        scale=scale, S=S, rhs=rhs, dcam=delta[:nc].reshape(Nc, 6), dpt=delta[nc:].reshape(Np, 3),
        mcc=mcc, dcam_fix=dcam_fix.reshape(Nc, 6), radius=radius)
    print("ba_small: Nc=%d Np=%d No=%d cost=%.6f" % (Nc, Np, sc.No, 0.5 * float((res ** 2).sum())))


def brute_distance(d0, d1, xy0=None, xy1=None):
    n0, n1 = len(d0), len(d1)
    out = np.ones((n0, n1), np.float32)
    a = d0.astype(np.float64); b = d1.astype(np.float64)
    for i in range(n0):
        prod = a[i][None, :] * b                      # exact products
        dot = np.cumsum(prod, axis=1)[:, -1]          # sequential fp64 accumulation
        s = dot.astype(np.float32)
        s = np.where(s < np.float32(-1), np.float32(-1), np.where(np.float32(1) < s, np.float32(1), s))
        d = np.sqrt(np.float32(2) - np.float32(2) * s).astype(np.float32)
        if xy0 is not None:
            masked = (np.abs(xy0[i, 1] - xy1[:, 1]) > 20) | (xy0[i, 0] < xy1[:, 0])
            d = np.where(masked, np.float32(1), d)
        out[i] = d
    return out


def brute_mutual(dmat, nn_thresh):
    idx0 = np.argmin(dmat, axis=1)     # numpy argmin returns the first minimum
    idx1 = np.argmin(dmat, axis=0)
    keep = [(i, int(idx0[i]), dmat[i, idx0[i]]) for i in range(dmat.shape[0])
            if idx1[idx0[i]] == i and dmat[i, idx0[i]] < nn_thresh]
    oi = np.array([k[0] for k in keep], np.int32); oj = np.array([k[1] for k in keep], np.int32)
    od = np.array([k[2] for k in keep], np.float32)
    return idx0.astype(np.int32), idx1.astype(np.int32), oi, oj, od


def make_match():
    d0, d1 = synth.make_descriptors(64, 64, seed=1, duplicates=3)
    dm = brute_distance(d0, d1)
    i0, i1, oi, oj, od = brute_mutual(dm, np.float32(2))
    np.savez_compressed(os.path.join(HERE, "match_64.npz"), d0=d0, d1=d1, idx0=i0, idx1=i1, oi=oi,
                        oj=oj, od=od)
    d0, d1 = synth.make_descriptors(512, 384, seed=2, duplicates=8)
    dm = brute_distance(d0, d1)
    i0, i1, oi, oj, od = brute_mutual(dm, np.float32(0.7))
    np.savez_compressed(os.path.join(HERE, "match_512.npz"), d0=d0, d1=d1, idx0=i0, idx1=i1, oi=oi,
                        oj=oj, od=od, nn_thresh=np.float32(0.7))
    d0, d1 = synth.make_descriptors(200, 230, seed=3)
    xy0, xy1 = synth.make_keypoints(200, seed=3), synth.make_keypoints(230, seed=4)
    dm = brute_distance(d0, d1, xy0, xy1)
    i0, i1, oi, oj, od = brute_mutual(dm, np.float32(2))
    np.savez_compressed(os.path.join(HERE, "stereo_200.npz"), d0=d0, d1=d1, xy0=xy0, xy1=xy1,
                        idx0=i0, idx1=i1, oi=oi, oj=oj, od=od)
    print("match fixtures written; matches:", len(oi))


if __name__ == "__main__":
    make_ba()
    make_match()
