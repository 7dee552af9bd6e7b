"""CPU: the C++ shim's Flatten (theiasfm_amd/host/bundle_adjuster.cc, what Optimize() sends to the
device) against the Python flatten (theiasfm_amd.io.flatten_reconstruction) on the same
reconstruction -- the reference fixture fountain11 (committed flat arrays, one shared intrinsics
group) and a mixed-model problem with shared groups and un-estimated views / tracks.  Both follow
BundleAdjustReconstruction's residual set (bundle_adjustment.cc:66-80, bundle_adjuster.cc:102-180);
they were written independently (C++ hash-container walk vs Python dict walk)."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
from theiasfm_amd import abi, io, synth  # noqa: E402

EXE = os.path.join(ROOT, "tests", "cpp", "flatten_dump")
GOLDEN = os.path.join(ROOT, "tests", "golden", "fountain11_flat.npz")


def _dump(rec: io.TheiaReconstruction, path: str):
    vids = sorted(rec.views)
    tids = sorted(rec.tracks)
    assert vids == list(range(len(vids))) and tids == list(range(len(tids)))
    gids = sorted(set(rec.view_to_group.values()))
    gindex = {g: i for i, g in enumerate(gids)}
    obs = [(v, t, xy) for v in vids for t, xy in rec.views[v].features.items()]
    with open(path, "wb") as f:
        f.write(struct.pack("<4q", len(vids), len(gids), len(tids), len(obs)))
        for v in vids:
            view = rec.views[v]
            f.write(struct.pack("<IBi6d", v, int(view.is_estimated), gindex[rec.view_to_group[v]], *view.extrinsics))
        for g in gids:
            owner = rec.views[rec.groups[g][0]]
            p = list(owner.intrinsics) + [0.0] * (10 - len(owner.intrinsics))
            f.write(struct.pack("<ii10d", owner.model, len(owner.intrinsics), *p))
        for t in tids:
            tr = rec.tracks[t]
            f.write(struct.pack("<IB4d", t, int(tr.is_estimated), *tr.point))
        for v, t, xy in obs:
            f.write(struct.pack("<II2d", v, t, xy[0], xy[1]))


def _load(path: str):
    d = open(path, "rb").read()
    nc, g, npt, no, ni = struct.unpack_from("<5q", d, 0)
    o = 40

    def take(dtype, n):
        nonlocal o
        a = np.frombuffer(d, dtype=dtype, count=n, offset=o).copy()
        o += a.nbytes
        return a
    out = dict(extrinsics=take("<f8", 6 * nc).reshape(-1, 6), camera_group=take("<i4", nc), camera_flags=take("u1", nc),
               group_model=take("<i4", g), group_offset=take("<i4", g + 1), intrinsics=take("<f8", ni),
               intrinsics_constant=take("u1", ni), points=take("<f8", 4 * npt).reshape(-1, 4),
               point_constant=take("u1", npt), obs_camera=take("<i4", no), obs_point=take("<i4", no),
               obs_xy=take("<f8", 2 * no).reshape(-1, 2), view_ids=take("<u4", nc), track_ids=take("<u4", npt))
    assert o == len(d)
    return out


def _check(rec, tmp_path, bulk, ito=abi.INTRINSICS_DEFAULT):
    entry.build_engine()
    entry.build_host_shim()
    src, dst = str(tmp_path / "rec.bin"), str(tmp_path / "flat.bin")
    _dump(rec, src)
    p = subprocess.run([EXE, src, dst, "1" if bulk else "0", str(int(ito))], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    C = _load(dst)
    P = io.flatten_reconstruction(rec, intrinsics_to_optimize=ito)
    assert list(C["view_ids"]) == P.meta["view_ids"]
    assert list(C["track_ids"]) == P.meta["track_ids"]
    for k in ("extrinsics", "camera_group", "camera_flags", "group_model", "group_offset", "intrinsics",
              "intrinsics_constant", "points", "point_constant"):
        np.testing.assert_array_equal(C[k], getattr(P, k), err_msg=k)
    # the order of the residuals is an implementation detail on both sides: compare as sets of rows
    kc = np.lexsort((C["obs_camera"], C["obs_point"]))
    kp = np.lexsort((P.obs_camera, P.obs_point))
    np.testing.assert_array_equal(C["obs_camera"][kc], P.obs_camera[kp])
    np.testing.assert_array_equal(C["obs_point"][kc], P.obs_point[kp])
    np.testing.assert_array_equal(C["obs_xy"][kc], P.obs_xy[kp])
    return C, P


@pytest.mark.parametrize("bulk", [True, False])
def test_fountain11_flattens_identically_in_cpp_and_python(tmp_path, bulk):
    z = np.load(GOLDEN)
    P0 = abi.Problem(**{k: z[k] for k in z.files})
    rec = io.reconstruction_from_problem(P0, image_size=(3072, 2048))
    C, P = _check(rec, tmp_path, bulk)
    assert C["extrinsics"].shape == (11, 6) and C["points"].shape[0] == 16616 and C["obs_camera"].shape[0] == 75022
    np.testing.assert_array_equal(P.points, P0.points)


@pytest.mark.parametrize("bulk", [True, False])
def test_mixed_models_shared_groups_and_unestimated_entries(tmp_path, bulk):
    P0 = synth.make_problem(14, 300, 1500, seed=9, scene="ring", spread=0.6, shared_group_size=3,
                            models=[(abi.PINHOLE, 0.3), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.2), (abi.FISHEYE, 0.2),
                                    (abi.FOV, 0.15), (abi.DIVISION_UNDISTORTION, 0.15)])
    rec = io.reconstruction_from_problem(P0, image_size=(1000, 800))
    # un-estimated views and tracks drop out of the residual set (bundle_adjuster.cc:106,145)
    rec.views[3].is_estimated = False
    rec.views[8].is_estimated = False
    for t in (0, 17, 44, 123):
        rec.tracks[t].is_estimated = False
    ito = abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS | abi.INTRINSICS_RADIAL_DISTORTION
    C, P = _check(rec, tmp_path, bulk, ito)
    assert 3 not in C["view_ids"] and 17 not in C["track_ids"]
    assert C["intrinsics_constant"].sum() > 0 and (C["intrinsics_constant"] == 0).sum() > 0
