"""GPU test of the device-side construction of the index structures (large edge sets) against the
host-side construction (small edge sets): both are forced on the same scenes and every array that
crosses the C ABI must come out bit for bit the same -- the two builds order the observations, the
camera pairs and the landmarks of a pair identically, so every sum runs in the same order."""
import numpy as np
import pytest

from visual_slam_amd import synth
from visual_slam_amd.ba import BAProblem

pytestmark = pytest.mark.gpu


def build(monkeypatch, how, sc, pts, **kw):
    monkeypatch.setenv("VS_BA_BUILD", how)
    p = BAProblem.from_arrays(sc.K, sc.t0, sc.rpy0, pts, sc.cam, sc.pt, sc.uv, **kw)
    monkeypatch.delenv("VS_BA_BUILD")
    return p


def shuffled(sc, seed):
    """The same scene with the landmarks numbered at random and the observations in random order."""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(sc.Np)              # new index of old landmark j
    order = rng.permutation(len(sc.cam))
    out = type("Scene", (), {})()
    out.K, out.t0, out.rpy0, out.Nc, out.Np = sc.K, sc.t0, sc.rpy0, sc.Nc, sc.Np
    out.pts0 = np.empty_like(sc.pts0); out.pts0[perm] = sc.pts0
    out.cam = sc.cam[order].copy(); out.pt = perm[sc.pt[order]].astype(np.int32); out.uv = sc.uv[order].copy()
    return out


@pytest.mark.parametrize("case", ["c1", "shuffled", "wide", "c2"])
@pytest.mark.parametrize("kw", [dict(), dict(fix_points=True), dict(huber_a=1.5), dict(rank=1, world=3)])
def test_device_build_equals_host_build(monkeypatch, case, kw):
    if case == "c1":
        sc = synth.make_ba_config("c1", seed=5)
    elif case == "shuffled":
        sc = shuffled(synth.make_ba_scene(40, 3000, 5, band=10, seed=6), 7)
    elif case == "wide":
        sc = shuffled(synth.make_ba_scene(120, 2500, 4, band=100, seed=8), 9)    # camera pairs far from the diagonal
    else:
        sc = synth.make_ba_config("c2", seed=1)
    dc = np.full(6 * sc.Nc, 1e-3); dp = np.full(3 * sc.Np, 1e-3)
    got = []
    for how in ("host", "device"):
        p = build(monkeypatch, how, sc, sc.pts0, **kw)
        p.linearize(dc, dp)
        br, bc, S, rhs, cost = p.reduced_system()
        Vinv6, gp = p.landmark_blocks()
        ecost, res, Jc, Jp = p.evaluate()
        step = None
        if "world" not in kw:
            step = p.lm_step(1e4)
        got.append((br, bc, S, rhs, cost, Vinv6, gp, res, Jc, Jp, np.array(p.exchange_size() + (ecost,)), step, p.points()))
        p.close()
    a, b = got
    for x, y in zip(a[:11], b[:11]):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    if a[11] is not None:
        for x, y in zip(a[11], b[11]):
            assert np.array_equal(np.asarray(x), np.asarray(y))
    assert np.array_equal(a[12], b[12])


def test_device_build_reports_bad_edges(monkeypatch):
    sc = synth.make_ba_config("c1", seed=5)
    cam = sc.cam.copy(); cam[17] = sc.Nc
    bad = type("S", (), dict(K=sc.K, t0=sc.t0, rpy0=sc.rpy0, cam=cam, pt=sc.pt, uv=sc.uv))
    with pytest.raises(Exception, match="camera index"):
        build(monkeypatch, "device", bad, sc.pts0)
    pt = sc.pt.copy(); pt[5] = -1
    bad.cam, bad.pt = sc.cam, pt
    with pytest.raises(Exception, match="point index"):
        build(monkeypatch, "device", bad, sc.pts0)
    # one landmark observed twice by the same camera
    cam = np.concatenate([sc.cam, sc.cam[:1]]); pt = np.concatenate([sc.pt, sc.pt[:1]]); uv = np.concatenate([sc.uv, sc.uv[:1]])
    bad.cam, bad.pt, bad.uv = cam, pt, uv
    with pytest.raises(Exception, match="observed twice"):
        build(monkeypatch, "device", bad, sc.pts0)


def test_points_set_before_and_after_a_device_build(monkeypatch):
    """Points handed over before the (renumbering) edge set and a second edge set on one handle."""
    sc = shuffled(synth.make_ba_scene(30, 2000, 4, band=8, seed=11, k_jitter=0), 12)
    ref = build(monkeypatch, "host", sc, sc.pts0)
    dev = build(monkeypatch, "device", sc, sc.pts0)
    assert np.array_equal(ref.points(), dev.points())
    sc2 = shuffled(synth.make_ba_scene(30, 2000, 4, band=8, seed=13, k_jitter=0), 14)
    for p, how in ((ref, "host"), (dev, "device")):
        monkeypatch.setenv("VS_BA_BUILD", how)
        p.set_observations(sc2.cam, sc2.pt, sc2.uv)
        monkeypatch.delenv("VS_BA_BUILD")
    assert np.array_equal(ref.points(), dev.points())
    ref.set_points(sc2.pts0); dev.set_points(sc2.pts0)
    a, b = ref.lm_step(1e4), dev.lm_step(1e4)
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))
