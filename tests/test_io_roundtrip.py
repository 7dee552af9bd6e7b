"""CPU: on-disk interchange: cereal Reconstruction archive out and in
again, BAL text out and in again."""
import os

import numpy as np
import pytest

from theiasfm_amd import abi, io, synth

FOUNTAIN = "/root/reference/data/sfm/fountain11.bin"  # only in the development container
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.exists(FOUNTAIN), reason="reference fixture not on this machine")
def test_cereal_archive_round_trip_is_byte_identical(tmp_path):
    rec = io.read_theia_reconstruction(FOUNTAIN)
    out = tmp_path / "same.bin"
    io.write_theia_reconstruction(str(out), rec)
    assert out.read_bytes() == open(FOUNTAIN, "rb").read()


@pytest.mark.skipif(not os.path.exists(FOUNTAIN), reason="reference fixture not on this machine")
def test_cereal_archive_carries_adjusted_values(tmp_path):
    rec = io.read_theia_reconstruction(FOUNTAIN)
    prob = io.flatten_reconstruction(rec)
    gold = np.load(os.path.join(HERE, "golden", "fountain11_flat.npz"))
    np.testing.assert_array_equal(prob.points, gold["points"])  # same flattening as the fixture
    rng = np.random.default_rng(0)
    prob.extrinsics += 1e-3 * rng.standard_normal(prob.extrinsics.shape)
    prob.intrinsics[0] *= 1.01
    prob.points[:, :3] += 1e-3 * rng.standard_normal((prob.num_points, 3))
    flags = np.zeros(prob.num_points, np.uint8)
    flags[::7] = 1
    io.update_reconstruction(rec, prob, track_flags=flags)
    out = tmp_path / "adjusted.bin"
    io.write_theia_reconstruction(str(out), rec)
    assert os.path.getsize(out) == os.path.getsize(FOUNTAIN)
    rec2 = io.read_theia_reconstruction(str(out))
    # the filtered tracks are gone from the flattened problem, everything else is what was written
    prob2 = io.flatten_reconstruction(rec2)
    keep = flags == 0
    np.testing.assert_array_equal(prob2.extrinsics, prob.extrinsics)
    np.testing.assert_array_equal(prob2.intrinsics, prob.intrinsics)
    np.testing.assert_array_equal(prob2.points, prob.points[keep])
    assert prob2.meta["track_ids"] == [t for t, k in zip(prob.meta["track_ids"], keep) if k]
    assert sum(not t.is_estimated for t in rec2.tracks.values()) >= int(flags.sum())


def test_bal_round_trip(tmp_path):
    P = synth.make_problem(7, 60, 260, seed=2)
    path = tmp_path / "problem.txt"
    io.write_bal(str(path), P)
    Q = io.read_bal(str(path))
    assert Q.num_cameras == P.num_cameras and Q.num_points == P.num_points
    np.testing.assert_array_equal(Q.obs_camera, P.obs_camera)
    np.testing.assert_array_equal(Q.obs_point, P.obs_point)
    np.testing.assert_allclose(Q.obs_xy, P.obs_xy, rtol=0, atol=1e-12)
    np.testing.assert_allclose(Q.extrinsics[:, :3], P.extrinsics[:, :3], rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(Q.extrinsics[:, 3:], P.extrinsics[:, 3:], rtol=1e-10, atol=1e-12)
    Kq, Kp = Q.intrinsics.reshape(-1, 7), P.intrinsics.reshape(-1, 7)
    np.testing.assert_allclose(Kq[:, [0, 5, 6]], Kp[:, [0, 5, 6]], rtol=1e-15)
    np.testing.assert_allclose(Q.points[:, :3], P.points[:, :3] / P.points[:, 3:4], rtol=1e-15)
    # and the two describe the same residuals
    from oracle import oracle
    Pn = P.copy()
    Pn.intrinsics = Pn.intrinsics.copy()
    Kn = Pn.intrinsics.reshape(-1, 7)
    Kn[:, 1], Kn[:, 2], Kn[:, 3], Kn[:, 4] = 1.0, 0.0, 0.0, 0.0  # what BAL can express
    c_p, rmse_p, _ = oracle.cost(Pn)
    c_q, rmse_q, _ = oracle.cost(Q)
    np.testing.assert_allclose(c_q, c_p, rtol=1e-9)
