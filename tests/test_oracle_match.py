"""CPU tests: matcher oracle against the numpy brute-force fixtures and the reference's quirks."""
import numpy as np

from visual_slam_amd import synth


def check(golden_dir, name, fn):
    m = np.load(golden_dir + "/" + name)
    oi, oj, od = fn(m)
    assert np.array_equal(oi, m["oi"]) and np.array_equal(oj, m["oj"])
    assert np.array_equal(od.view(np.uint32), m["od"].view(np.uint32))
    return m


def test_two_way_golden(orc, golden_dir):
    check(golden_dir, "match_64.npz", lambda m: orc.two_way_matching(m["d0"], m["d1"], 2.0))
    check(golden_dir, "match_512.npz",
          lambda m: orc.two_way_matching(m["d0"], m["d1"], float(m["nn_thresh"])))


def test_stereo_golden(orc, golden_dir):
    check(golden_dir, "stereo_200.npz",
          lambda m: orc.stereo_matching(m["d0"], m["xy0"], m["d1"], m["xy1"], 2.0))


def test_ties_resolve_to_lowest_index(orc):
    d0, _ = synth.make_descriptors(8, 8, seed=5)
    d1 = np.stack([d0[3], d0[3], d0[1], d0[1], d0[0], d0[2], d0[2], d0[5]])
    dm = orc.distance_matrix(d0, d1)
    assert dm[3, 0] == dm[3, 1]
    oi, oj, _ = orc.two_way_matching(d0, d1)
    assert (3, 0) in list(zip(oi.tolist(), oj.tolist()))      # first of the duplicate columns
    assert (1, 2) in list(zip(oi.tolist(), oj.tolist()))


def test_masked_entries_compete_at_distance_one(orc):
    d0, d1 = synth.make_descriptors(4, 4, seed=6, unmatched=1.0)
    xy0 = np.array([[10, 10]] * 4, np.int32)
    xy1 = np.array([[50, 10]] * 4, np.int32)              # x0 < x1: everything masked
    dm = orc.distance_matrix(d0, d1, xy0, xy1)
    assert np.all(dm == np.float32(1.0))
    oi, oj, od = orc.stereo_matching(d0, xy0, d1, xy1)
    assert oi.tolist() == [0] and oj.tolist() == [0] and od.tolist() == [1.0]


def test_empty_inputs(orc):
    d0, d1 = synth.make_descriptors(4, 4, seed=1)
    oi, _, _ = orc.two_way_matching(d0[:0], d1)
    assert len(oi) == 0
    oi, _, _ = orc.two_way_matching(d0, d1[:0])
    assert len(oi) == 0


def test_ratio_is_optin(orc):
    d0, d1 = synth.make_descriptors(128, 128, seed=9)
    a = orc.two_way_matching(d0, d1, 2.0, 1.0)
    b = orc.two_way_matching(d0, d1, 2.0, 0.8)
    assert len(b[0]) <= len(a[0]) and set(b[0].tolist()) <= set(a[0].tolist())


def test_search_projection_greedy_claim(orc):
    d, _ = synth.make_descriptors(3, 3, seed=2)
    key_desc = np.stack([d[0], d[0], d[1]])
    tp_desc = np.stack([d[0], d[0], d[2]])
    key_xy = np.array([[10, 10], [11, 10], [100, 100]], np.int32)
    reproj = np.array([[10, 10], [10, 10], [500, 10]], np.int32)
    tracks, claimed = orc.search_projection(reproj, tp_desc, key_xy, key_desc, 383.0, 206.8)
    assert tracks.tolist() == [0, 1, -1]       # second point takes the still-unclaimed duplicate
    assert claimed.tolist() == [1, 1, 0]
