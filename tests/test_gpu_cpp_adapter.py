"""GPU tests of the C++ host adapters (visual_slam_amd/host): the reference's call order through
core::Optimizer / core::SuperpointFrontend, compared with the CPU oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from visual_slam_amd import optimizer as vo
from visual_slam_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "cpp", "adapter_harness")


def run(inp, tmp_path):
    fi, fo = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    open(fi, "wb").write(inp)
    r = subprocess.run([HARNESS, fi, fo], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    return open(fo, "rb").read()


@pytest.mark.parametrize("fix", [1, 0])
def test_optimizer_adapter_matches_oracle(orc, tmp_path, fix):
    assert os.path.exists(HARNESS), "build the harness with `make`"
    sc, tracks, keys = synth.make_tracks_scene(6, 200, seed=5)
    nf, npnt = tracks.shape
    poses = np.zeros((nf, 7), np.float32)
    for f in range(nf):
        poses[f, :3] = sc.t0[f]
        poses[f, 3:] = vo.rpy_to_quat_f32(sc.rpy0[f].astype(np.float32))
    pw = (sc.pts_gt if fix else sc.pts0).astype(np.float32)
    blob = struct.pack("<4i", 0, nf, npnt, fix) + poses.tobytes()
    for f in range(nf):
        blob += struct.pack("<i", len(keys[f])) + keys[f].astype(np.int32).tobytes()
    blob += tracks.astype(np.int32).tobytes() + pw.tobytes()
    out = run(blob, tmp_path)
    got_pose = np.frombuffer(out, np.float32, nf * 7).reshape(nf, 7)
    got_pts = np.frombuffer(out, np.float64, npnt * 3, offset=nf * 28).reshape(npnt, 3)
    it = struct.unpack_from("<i", out, nf * 28 + npnt * 24)[0]
    final_cost = struct.unpack_from("<d", out, nf * 28 + npnt * 24 + 4)[0]
    # the same problem through the oracle, with the adapter's own float pose codec
    o = vo.Optimizer()
    for f in range(nf):
        o.addPose(vo.Rigid3f(poses[f, :3], poses[f, 3:]))
    t = np.array([p[0] for p in o.poses_]); rpy = np.array([p[1] for p in o.poses_])
    cam, pt, uv = orc.build_edges(tracks, keys)
    ref = orc.Problem(orc.K4(), t, rpy, pw.astype(np.float64), cam, pt, uv, huber_a=100.0, fix_points=bool(fix))
    summ = ref.solve()
    assert it == summ["iterations"]
    assert abs(final_cost - summ["final_cost"]) < 1e-6 * summ["final_cost"]
    assert np.abs(got_pose[:, :3] - ref.t).max() < 1e-4
    for f in range(nf):
        q = vo.rpy_to_quat_f32(ref.rpy[f].astype(np.float32))
        assert np.abs(got_pose[f, 3:] - q).max() < 1e-5
    assert np.abs(got_pts - ref.pts).max() < 1e-5 * max(1.0, np.abs(ref.pts).max())


def test_matcher_adapter_matches_oracle(orc, tmp_path):
    d0, d1 = synth.make_descriptors(300, 260, seed=8, duplicates=2)
    blob = struct.pack("<3i", 1, 300, 260) + d0.tobytes() + d1.tobytes() + struct.pack("<f", 0.7)
    out = run(blob, tmp_path)
    n = struct.unpack_from("<i", out, 0)[0]
    rec = np.frombuffer(out, np.dtype([("i", "<i4"), ("j", "<i4"), ("d", "<f4")]), n, offset=4)
    oi, oj, od = orc.two_way_matching(d0, d1, 0.7)
    assert np.array_equal(rec["i"], oi) and np.array_equal(rec["j"], oj)
    assert np.array_equal(rec["d"].view(np.uint32), od.view(np.uint32))
    xy0, xy1 = synth.make_keypoints(300, seed=1), synth.make_keypoints(260, seed=2)
    blob = struct.pack("<3i", 2, 300, 260) + d0.tobytes() + d1.tobytes() + xy0.tobytes() + xy1.tobytes() + struct.pack("<f", 2.0)
    out = run(blob, tmp_path)
    n = struct.unpack_from("<i", out, 0)[0]
    rec = np.frombuffer(out, np.dtype([("i", "<i4"), ("j", "<i4"), ("d", "<f4")]), n, offset=4)
    oi, oj, od = orc.stereo_matching(d0, xy0, d1, xy1, 2.0)
    assert np.array_equal(rec["i"], oi) and np.array_equal(rec["j"], oj)
    assert np.array_equal(rec["d"].view(np.uint32), od.view(np.uint32))
