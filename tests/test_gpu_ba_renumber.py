"""The library numbers landmarks internally by the first camera that sees them.  Everything the
caller hands over or gets back per landmark stays in the caller's numbering (GPU, against the
oracle): per-landmark damping, V^-1 / g_p, LM step, points before and after the edges, a second
edge set on the same handle, and an input that already is in keyframe order."""
import numpy as np
import pytest

from oracle import oracle as orc
from visual_slam_amd import synth
from visual_slam_amd.ba import BAProblem

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def sym6(Vinv):
    return np.stack([Vinv[:, 0, 0], Vinv[:, 0, 1], Vinv[:, 0, 2], Vinv[:, 1, 1], Vinv[:, 1, 2], Vinv[:, 2, 2]], axis=1)


def scene(seed, shuffle):
    sc = synth.make_ba_scene(30, 1500, 5, band=12, seed=seed)
    if not shuffle:
        sc = synth.landmarks_in_keyframe_order(sc)
    return sc


@pytest.mark.parametrize("shuffle", [True, False])
def test_per_landmark_inputs_and_outputs_keep_the_callers_numbering(shuffle):
    sc = scene(5, shuffle)
    rng = np.random.default_rng(1)
    dc = rng.uniform(0.1, 2.0, 6 * sc.Nc)
    dp = rng.uniform(0.1, 2.0, 3 * sc.Np)                  # differs from landmark to landmark
    ref = orc.Problem(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    want = ref.schur(dc=dc, dp=dp)
    gpu = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    gpu.linearize(dc, dp)
    S, rhs, cost = gpu.reduced_system_dense()
    assert rel(S, want["S"]) < 1e-9 and rel(rhs, want["rhs"]) < 1e-9
    Vinv6, gp = gpu.landmark_blocks()
    assert rel(Vinv6, sym6(want["Vinv"])) < 1e-9 and rel(gp, want["gp"]) < 1e-9
    assert np.array_equal(gpu.points(), sc.pts0)          # untouched points come back as they went in
    rc, dcam_o, dpt_o, mcc_o, _, _ = ref.lm_step(1e4)
    dcam, dpt, mcc = gpu.lm_step(1e4)
    assert rc == 0 and rel(dcam, dcam_o) < 1e-6 and rel(dpt, dpt_o) < 1e-6 and abs(mcc - mcc_o) < 1e-8 * abs(mcc_o)
    gpu.close()


def test_points_set_after_the_edges_and_a_second_edge_set():
    sc = scene(6, True)
    gpu = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, sc.pts0, sc.cam, sc.pt, sc.uv)
    moved = sc.pts0 + 0.01
    gpu.set_points(moved)                                  # after the edges: goes through the renumbering
    assert np.array_equal(gpu.points(), moved)
    ref = orc.Problem(sc.K, sc.t0, sc.rpy0, moved, sc.cam, sc.pt, sc.uv)
    dp = np.linspace(0.1, 1.0, 3 * sc.Np)
    gpu.linearize(None, dp)
    want = ref.schur(dp=dp)
    assert rel(gpu.reduced_system_dense()[0], want["S"]) < 1e-9
    # another edge set on the same handle (other landmark order): the points held stay the caller's
    keep = np.sort(np.random.default_rng(2).permutation(sc.No))          # same edges ...
    relabel = np.random.default_rng(3).permutation(sc.Np)                # ... landmarks renamed
    pts2 = np.empty_like(moved); pts2[relabel] = moved
    gpu.set_points(pts2)
    gpu.set_observations(sc.cam[keep], relabel[sc.pt[keep]], sc.uv[keep])
    assert np.array_equal(gpu.points(), pts2)
    ref2 = orc.Problem(sc.K, sc.t0, sc.rpy0, pts2, sc.cam[keep], relabel[sc.pt[keep]], sc.uv[keep])
    dp2 = np.empty_like(dp).reshape(-1, 3); dp2[relabel] = dp.reshape(-1, 3)
    gpu.linearize(None, dp2.ravel())
    want2 = ref2.schur(dp=dp2.ravel())
    assert rel(gpu.reduced_system_dense()[0], want2["S"]) < 1e-9
    assert rel(want2["S"], want["S"]) < 1e-9                              # the same problem under other names
    out = gpu.solve()
    ref_out = ref2.solve()                                                # the oracle updates its own points in place
    assert out["iterations"] == ref_out["iterations"] and out["successful_steps"] == ref_out["successful_steps"]
    assert abs(out["final_cost"] - ref_out["final_cost"]) < 1e-9 * ref_out["final_cost"]
    assert rel(gpu.points(), ref2.pts.reshape(-1, 3)) < 1e-6
    gpu.close()
