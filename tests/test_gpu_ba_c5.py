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


def damped_system_residual(sc, gpu, dcam):
    """|S d - rhs| / |rhs| of the damped block-sparse reduced system the handle holds after an LM step."""
    import scipy.sparse as sp
    br, bc, Sd, rhs_d, _ = gpu.reduced_system()
    n = 6 * sc.Nc
    off = br != bc
    rows = np.concatenate([br, bc[off]]); cols = np.concatenate([bc, br[off]])
    data = np.concatenate([Sd, Sd[off].transpose(0, 2, 1)])
    order = np.lexsort((cols, rows))
    indptr = np.searchsorted(rows[order], np.arange(sc.Nc + 1))
    A = sp.bsr_matrix((data[order], cols[order], indptr), shape=(n, n))
    return np.linalg.norm(A @ dcam.ravel() - rhs_d) / np.linalg.norm(rhs_d)


def test_c5_lm_step_solves_its_reduced_system(c5):
    """The LM step of the full 10k-camera problem (block cyclic reduction over 192-wide blocks,
    nine levels): the camera step satisfies the damped reduced system it was computed from, the
    model cost decreases, the landmark steps are finite."""
    sc = c5
    gpu = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    dcam, dpt, mcc = gpu.lm_step(1e4)
    assert damped_system_residual(sc, gpu, dcam) < 1e-9
    assert mcc > 0 and np.isfinite(dpt).all() and np.isfinite(dcam).all()
    # the step moves every landmark (none left at zero by a skipped tile)
    assert (np.abs(dpt).max(axis=1) > 0).mean() > 0.999
    gpu.close()


def test_cyclic_reduction_matches_band_cholesky_at_10k_cameras(monkeypatch):
    """10k cameras, 50k landmarks, band of 32: cyclic reduction (nine levels) against the sequential
    band Cholesky on the same handle, and both against the reduced system."""
    sc = synth.make_ba_scene(10_000, 50_000, 5, band=32, seed=41)
    gpu = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    d1, p1, m1 = gpu.lm_step(1e4)
    assert damped_system_residual(sc, gpu, d1) < 1e-9
    monkeypatch.setenv("VS_BA_SOLVER", "band")
    d2, p2, m2 = gpu.lm_step(1e4)
    monkeypatch.delenv("VS_BA_SOLVER")
    assert rel(d1, d2) < 1e-8 and rel(p1, p2) < 1e-8 and abs(m1 - m2) < 1e-9 * abs(m2)
    gpu.close()
