"""GPU test of the largest benchmark shape: 10k cameras, 4M landmarks, ~20M observations
(BASELINE.json configs[4]) on one GPU.

The CPU oracle cannot form the 60,000 x 60,000 reduced system, so it is evaluated on windows of
consecutive cameras: every S block (a, b) with both cameras inside a window, and the window's rows
of the right-hand side, depend only on the landmarks those cameras see (with ALL their
observations, cameras outside the window included, because V_j sums over every observation of
landmark j).  That sub-problem, with cameras renumbered compactly, is what the oracle solves.  The
whole-problem cost is compared too, and eight landmark shards' partial systems must add up to the
whole (the multi-GPU layout of this shape).  Tolerances as for config 4 (see
test_gpu_ba_fullsize.py): every sampled block within 1e-6 of the matrix scale, 99.9 % within 1e-10."""
import numpy as np
import pytest

from visual_slam_amd import synth
from visual_slam_amd.ba import BAProblem

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


@pytest.fixture(scope="module")
def c5():
    return synth.make_ba_config("c5", seed=0)


def window_subproblem(sc, c0, c1):
    """Observations of every landmark seen by a camera in [c0, c1), cameras renumbered compactly."""
    seen = np.zeros(sc.Np, bool)
    seen[sc.pt[(sc.cam >= c0) & (sc.cam < c1)]] = True
    keep = seen[sc.pt]
    cam, pt, uv = sc.cam[keep], sc.pt[keep], sc.uv[keep]
    cams = np.unique(cam)
    pts = np.unique(pt)
    return cams, pts, np.searchsorted(cams, cam).astype(np.int32), np.searchsorted(pts, pt).astype(np.int32), uv


def test_c5_windows_match_oracle_and_shards_add_up(orc, c5):
    sc = c5
    assert sc.Nc == 10_000 and sc.Np == 4_000_000 and sc.No > 19_000_000
    dc = np.full(6 * sc.Nc, 1e-4); dp = np.full(3 * sc.Np, 1e-4)
    gpu = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    gpu.linearize(dc, dp)
    br, bc, S, rhs, cost = gpu.reduced_system()
    Vinv6, gp = gpu.landmark_blocks()
    # whole-problem cost against the oracle's residuals (0.5 * sum |r|^2)
    ref_all = orc.Problem(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    # 20M terms added in different orders (sequential in the oracle, a fixed tree on the GPU):
    # n * eps bounds the difference at 2e-9 relative
    assert abs(cost - ref_all.cost()) < 1e-8 * cost
    scale = np.abs(S[:sc.Nc]).max()
    checked = 0
    for c0 in (0, 4321, 9960):                         # first cameras, middle, last cameras
        c1 = c0 + 40
        cams, pts, cam_l, pt_l, uv = window_subproblem(sc, c0, c1)
        ref = orc.Problem(sc.K, sc.t0[cams], sc.rpy0[cams], sc.pts0[pts], cam_l, pt_l, uv)
        s = ref.schur(dc=dc[:6 * len(cams)], dp=dp[:3 * len(pts)])
        nl = len(cams)
        want = s["S"].reshape(nl, 6, nl, 6).transpose(0, 2, 1, 3)
        sel = (br >= c0) & (br < c1) & (bc >= c0) & (bc < c1)
        la, lb = np.searchsorted(cams, br[sel]), np.searchsorted(cams, bc[sel])
        err = np.abs(S[sel] - want[la, lb]).max(axis=(1, 2)) / scale
        assert sel.sum() >= 40 and err.max() < 1e-6 and np.quantile(err, 0.999) < 1e-10
        # the oracle's window holds no block the GPU pattern lacks
        inside = np.zeros((nl, nl), bool)
        inside[la, lb] = True; inside[lb, la] = True
        lw = np.searchsorted(cams, np.arange(c0, c1))
        assert not np.abs(want[np.ix_(lw, lw)]).max(axis=(2, 3))[~inside[np.ix_(lw, lw)]].any()
        rows = (6 * np.arange(c0, c1)[:, None] + np.arange(6)).ravel()
        rows_l = (6 * lw[:, None] + np.arange(6)).ravel()
        assert rel(rhs[rows], s["rhs"][rows_l]) < 1e-7
        V = s["Vinv"]
        ref6 = np.stack([V[:, 0, 0], V[:, 0, 1], V[:, 0, 2], V[:, 1, 1], V[:, 1, 2], V[:, 2, 2]], 1)
        ev = np.abs(Vinv6[pts] - ref6).max(axis=1) / np.abs(ref6).max(axis=1)
        assert ev.max() < 1e-7 and np.median(ev) < 1e-12 and rel(gp[pts], s["gp"]) < 1e-11   # cameras up to 500 m from the origin: R Pw + t cancels three digits
        checked += int(sel.sum())
    assert checked > 2000
    # eight landmark shards (the 8-GPU layout): partial systems add up to the whole
    gpu.linearize(None, dp)
    _, _, S, rhs, cost = gpu.reduced_system()
    gpu.close()
    tot_S, tot_rhs, tot_cost, tot_obs = np.zeros_like(S), np.zeros_like(rhs), 0.0, 0
    for r in range(8):
        part = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv, rank=r, world=8)
        part.linearize(None, dp)
        b2, c2, Sp, rp, cp = part.reduced_system()
        assert np.array_equal(b2, br) and np.array_equal(c2, bc)
        tot_S += Sp; tot_rhs += rp; tot_cost += cp
        part.close()
    assert rel(tot_S, S) < 1e-12 and rel(tot_rhs, rhs) < 1e-12 and abs(tot_cost - cost) < 1e-12 * cost
