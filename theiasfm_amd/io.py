"""File readers for the data either side of the BA path.

* ``read_theia_reconstruction``: TheiaSfM's cereal portable-binary
  ``Reconstruction`` (reference: src/theia/io/reconstruction_reader.cc:53-77;
  class layouts reconstruction.h:158-167, view.h:91-94, track.h:80-83,
  camera.h:206-245, camera_intrinsics_prior.h:58-107, io/eigen_serializable.h:47-59).
  Only the subset of the format the shipped fixtures use is understood
  (PINHOLE intrinsics, Camera v0/v1, CameraIntrinsicsPrior v4).
* ``write_theia_reconstruction`` / ``update_reconstruction``: the way back -- the adjusted
  values patched into the archive that was read (reference writer:
  src/theia/io/reconstruction_writer.cc), so the real Theia can consume the result.
* ``read_bal`` / ``write_bal``: Bundle-Adjustment-in-the-Large text problems converted to the
  reference's conventions exactly as its Bundler importer does
  (reference: src/theia/io/read_bundler_files.cc:94-133,169,190).
* ``flatten_reconstruction``: the residual set BundleAdjustReconstruction builds
  (reference: bundle_adjustment.cc:66-80, bundle_adjuster.cc:102-180) as a
  flattened ``Problem`` with sorted-id (deterministic) block order.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np
from scipy.spatial.transform import Rotation

from . import abi
from .abi import Problem


# ---- cereal portable binary ---------------------------------------------------
class _Reader:
    def __init__(self, data: bytes):
        self.d = data
        self.o = 0

    def take(self, fmt: str):
        v = struct.unpack_from("<" + fmt, self.d, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def string(self) -> str:
        n = self.take("Q")
        s = self.d[self.o:self.o + n].decode("utf-8")
        self.o += n
        return s

    def f64(self, n: int) -> np.ndarray:
        a = np.frombuffer(self.d, dtype="<f8", count=n, offset=self.o).copy()
        self.o += 8 * n
        return a


@dataclass
class TheiaView:
    name: str
    is_estimated: bool
    extrinsics: np.ndarray           # [6]
    intrinsics_ptr: int              # shared pointer id (same id = shared block)
    intrinsics: np.ndarray           # [7] pinhole
    image_size: tuple
    features: dict                   # track id -> (x, y)
    offsets: dict = field(default_factory=dict)  # byte offsets of the fields BA rewrites


@dataclass
class TheiaTrack:
    is_estimated: bool
    view_ids: list
    point: np.ndarray                # [4]
    offsets: dict = field(default_factory=dict)


@dataclass
class TheiaReconstruction:
    views: dict = field(default_factory=dict)      # view id -> TheiaView
    tracks: dict = field(default_factory=dict)     # track id -> TheiaTrack
    view_to_group: dict = field(default_factory=dict)
    groups: dict = field(default_factory=dict)
    raw: bytes = b""                 # the archive as read (write_theia_reconstruction patches it)
    intrinsics_offsets: dict = field(default_factory=dict)  # shared pointer id -> byte offset


def read_theia_reconstruction(path: str) -> TheiaReconstruction:
    r = _Reader(open(path, "rb").read())
    if r.take("B") != 1:
        raise ValueError("big-endian cereal archives are not supported")
    seen = set()

    def version(tag):  # cereal writes a class version the first time a type appears
        if tag in seen:
            return None
        seen.add(tag)
        return r.take("I")

    versions = {}

    def ver(tag):
        v = version(tag)
        if v is not None:
            versions[tag] = v
        return versions[tag]

    rec = TheiaReconstruction()
    ver("Reconstruction")
    r.take("I")  # next_track_id
    r.take("I")  # next_view_id
    for _ in range(r.take("Q")):
        r.string()
        r.take("I")
    poly_types = {}
    shared = {}

    def prior(n):
        ver(f"Prior{n}")
        r.take("B")
        r.f64(n)

    nviews = r.take("Q")
    for _ in range(nviews):
        vid = r.take("I")
        ver("View")
        name = r.string()
        off = {"is_estimated": r.o}
        is_est = bool(r.take("B"))
        cam_ver = ver("Camera")
        if cam_ver > 0:
            off["extrinsics"] = r.o
            ext = r.f64(6)
            pid = r.take("I")
            if pid & 0x80000000:
                poly_types[pid & 0x7FFFFFFF] = r.string()
            tname = poly_types[pid & 0x7FFFFFFF]
            if tname != "theia::PinholeCameraModel":
                raise ValueError(f"unsupported intrinsics type in archive: {tname}")
            ptr = r.take("I")
            if ptr & 0x80000000:
                if ver("PinholeCameraModel") > 0:
                    ver("CameraIntrinsicsModel")
                    n = r.take("Q")
                    rec.intrinsics_offsets[ptr & 0x7FFFFFFF] = r.o
                    shared[ptr & 0x7FFFFFFF] = r.f64(n)
                else:
                    rec.intrinsics_offsets[ptr & 0x7FFFFFFF] = r.o
                    shared[ptr & 0x7FFFFFFF] = r.f64(7)
            intr_id = ptr & 0x7FFFFFFF
            intr = shared[intr_id]
            size = r.take("ii")
        else:  # Camera v0: 13 doubles [extrinsics(6), pinhole intrinsics(7)] + size
            off["extrinsics"] = r.o
            off["intrinsics"] = r.o + 48
            p = r.f64(13)
            ext, intr = p[:6], p[6:]
            intr_id = -1 - vid
            size = r.take("ii")
        # CameraIntrinsicsPrior
        pv = ver("CameraIntrinsicsPrior")
        if pv < 4:
            raise ValueError("CameraIntrinsicsPrior versions < 4 are not supported")
        r.take("ii")
        r.string()
        prior(1); prior(2); prior(1); prior(1); prior(4); prior(2)  # noqa: E702
        prior(3); prior(3); prior(1); prior(1); prior(1)  # noqa: E702
        feats = {}
        for _ in range(r.take("Q")):
            tid = r.take("I")
            rows, cols = r.take("ii")
            xy = r.f64(rows * cols)
            feats[tid] = (float(xy[0]), float(xy[1]))
        rec.views[vid] = TheiaView(name, is_est, ext, intr_id, intr, size, feats, off)
    for _ in range(r.take("Q")):
        tid = r.take("I")
        ver("Track")
        toff = {"is_estimated": r.o}
        is_est = bool(r.take("B"))
        vids = [r.take("I") for _ in range(r.take("Q"))]
        rows, cols = r.take("ii")
        toff["point"] = r.o
        pt = r.f64(rows * cols)
        rows, cols = r.take("ii")
        r.o += rows * cols
        rec.tracks[tid] = TheiaTrack(is_est, vids, pt, toff)
    for _ in range(r.take("Q")):
        k, v = r.take("II")
        rec.view_to_group[k] = v
    for _ in range(r.take("Q")):
        g = r.take("I")
        rec.groups[g] = [r.take("I") for _ in range(r.take("Q"))]
    if r.o != len(r.d):
        raise ValueError(f"trailing bytes in archive: parsed {r.o} of {len(r.d)}")
    rec.raw = r.d
    return rec


def write_theia_reconstruction(path: str, rec: TheiaReconstruction) -> None:
    """Writes `rec` as a cereal portable-binary archive the reference can read back
    (src/theia/io/reconstruction_writer.cc; same class layouts as the reader above).

    Bundle adjustment and the steps around it change values, never structure: camera
    extrinsics and intrinsics, track points and the estimated flags.  The writer therefore
    re-emits the archive `rec` was read from with exactly those fields replaced, so
    everything this package does not model (view names, priors, colours, feature tables)
    stays bit for bit what Theia wrote; an unmodified reconstruction round-trips to an
    identical file."""
    if not rec.raw:
        raise ValueError("write_theia_reconstruction needs a reconstruction obtained from "
                         "read_theia_reconstruction")
    out = bytearray(rec.raw)
    done = set()
    for view in rec.views.values():
        struct.pack_into("<B", out, view.offsets["is_estimated"], 1 if view.is_estimated else 0)
        struct.pack_into("<6d", out, view.offsets["extrinsics"], *np.asarray(view.extrinsics, float))
        if "intrinsics" in view.offsets:  # Camera v0: private copy inside the camera
            struct.pack_into("<7d", out, view.offsets["intrinsics"], *np.asarray(view.intrinsics, float))
        elif view.intrinsics_ptr not in done:  # shared block: stored once, at its first use
            done.add(view.intrinsics_ptr)
            vals = np.asarray(view.intrinsics, float)
            struct.pack_into(f"<{len(vals)}d", out, rec.intrinsics_offsets[view.intrinsics_ptr], *vals)
    for track in rec.tracks.values():
        struct.pack_into("<B", out, track.offsets["is_estimated"], 1 if track.is_estimated else 0)
        struct.pack_into("<4d", out, track.offsets["point"], *np.asarray(track.point, float))
    with open(path, "wb") as fh:
        fh.write(bytes(out))


def update_reconstruction(rec: TheiaReconstruction, prob: Problem, track_flags=None) -> None:
    """Writes an adjusted flattened problem (from flatten_reconstruction) back into `rec`:
    what BundleAdjuster::Optimize leaves in the caller's Reconstruction.  track_flags
    (optional, per problem track; non-zero = SetEstimated(false)) applies the outcome of
    the outlier filter."""
    vids, tids = prob.meta["view_ids"], prob.meta["track_ids"]
    for ci, v in enumerate(vids):
        view = rec.views[v]
        view.extrinsics = prob.extrinsics[ci].copy()
        g = int(prob.camera_group[ci])
        a, b = prob.group_offset[g], prob.group_offset[g + 1]
        new = prob.intrinsics[a:b].copy()
        for other in rec.views.values():  # every view of the shared block sees the update
            if other.intrinsics_ptr == view.intrinsics_ptr:
                other.intrinsics = new
    for pi, t in enumerate(tids):
        rec.tracks[t].point = prob.points[pi].copy()
        if track_flags is not None and track_flags[pi]:
            rec.tracks[t].is_estimated = False


def flatten_reconstruction(rec: TheiaReconstruction,
                           intrinsics_to_optimize: int = abi.INTRINSICS_DEFAULT) -> Problem:
    """BundleAdjustReconstruction's residual set, flattened (all estimated views
    and tracks variable; sorted ids give a deterministic block order)."""
    vids = sorted(v for v, view in rec.views.items() if view.is_estimated)
    cam_index = {v: i for i, v in enumerate(vids)}
    # an intrinsics group = a shared pointer in the archive; fall back to the
    # group map when present
    keyf = (lambda v: rec.view_to_group[v]) if rec.view_to_group else \
        (lambda v: rec.views[v].intrinsics_ptr)
    gkeys = sorted({keyf(v) for v in vids})
    gindex = {k: i for i, k in enumerate(gkeys)}
    ext = np.stack([rec.views[v].extrinsics for v in vids])
    cam_group = np.array([gindex[keyf(v)] for v in vids], dtype=np.int32)
    intr = np.zeros((len(gkeys), 7))
    for v in vids:
        intr[gindex[keyf(v)]] = rec.views[v].intrinsics
    tids = sorted(t for t, tr in rec.tracks.items()
                  if tr.is_estimated and any(v in cam_index for v in tr.view_ids))
    pts = np.stack([rec.tracks[t].point for t in tids])
    oc, op, oxy = [], [], []
    for pi, t in enumerate(tids):
        for v in sorted(rec.tracks[t].view_ids):
            if v in cam_index and t in rec.views[v].features:
                oc.append(cam_index[v])
                op.append(pi)
                oxy.append(rec.views[v].features[t])
    prob = Problem(
        extrinsics=ext, camera_group=cam_group, camera_flags=np.zeros(len(vids), np.uint8),
        group_model=np.zeros(len(gkeys), np.int32),
        group_offset=np.arange(len(gkeys) + 1, dtype=np.int32) * 7,
        intrinsics=intr.reshape(-1), intrinsics_constant=np.zeros(7 * len(gkeys), np.uint8),
        points=pts, point_constant=np.zeros(len(tids), np.uint8),
        obs_camera=np.array(oc, np.int32), obs_point=np.array(op, np.int32),
        obs_xy=np.array(oxy, np.float64))
    prob.set_intrinsics_to_optimize(intrinsics_to_optimize)
    prob.meta["view_ids"] = vids
    prob.meta["track_ids"] = tids
    return prob


# ---- BAL ----------------------------------------------------------------------
def read_bal(path: str, intrinsics_to_optimize: int = abi.INTRINSICS_DEFAULT) -> Problem:
    """BAL text -> reference conventions:
    R_theia = diag(1,-1,-1) R, t' = diag(1,-1,-1) t, C = -R_theia^T t',
    PINHOLE [f,1,0,0,0,k1,k2], feature (x, -y), point homogeneous w = 1."""
    opener = open
    if path.endswith(".bz2"):
        import bz2
        opener = bz2.open
    with opener(path, "rt") as fh:
        tok = fh.read().split()
    nc, npt, nobs = int(tok[0]), int(tok[1]), int(tok[2])
    o = 3
    obs = np.array(tok[o:o + 4 * nobs], dtype=np.float64).reshape(nobs, 4)
    o += 4 * nobs
    cams = np.array(tok[o:o + 9 * nc], dtype=np.float64).reshape(nc, 9)
    o += 9 * nc
    pts = np.array(tok[o:o + 3 * npt], dtype=np.float64).reshape(npt, 3)
    flip = np.diag([1.0, -1.0, -1.0])
    Rb = Rotation.from_rotvec(cams[:, :3]).as_matrix()
    Rt = np.einsum("ij,njk->nik", flip, Rb)
    tp = cams[:, 3:6] @ flip.T
    C = -np.einsum("nji,nj->ni", Rt, tp)
    aa = Rotation.from_matrix(Rt).as_rotvec()
    intr = np.zeros((nc, 7))
    intr[:, 0] = cams[:, 6]
    intr[:, 1] = 1.0
    intr[:, 5] = cams[:, 7]
    intr[:, 6] = cams[:, 8]
    xy = obs[:, 2:4].copy()
    xy[:, 1] = -xy[:, 1]
    prob = Problem(
        extrinsics=np.concatenate([C, aa], 1), camera_group=np.arange(nc, dtype=np.int32),
        camera_flags=np.zeros(nc, np.uint8), group_model=np.zeros(nc, np.int32),
        group_offset=np.arange(nc + 1, dtype=np.int32) * 7, intrinsics=intr.reshape(-1),
        intrinsics_constant=np.zeros(7 * nc, np.uint8),
        points=np.concatenate([pts, np.ones((npt, 1))], 1),
        point_constant=np.zeros(npt, np.uint8), obs_camera=obs[:, 0].astype(np.int32),
        obs_point=obs[:, 1].astype(np.int32), obs_xy=xy)
    prob.set_intrinsics_to_optimize(intrinsics_to_optimize)
    return prob


def write_bal(path: str, prob: Problem) -> None:
    """Inverse of read_bal (PINHOLE cameras with private intrinsics, w = 1 points)."""
    if np.any(prob.group_model != abi.PINHOLE) or prob.num_groups != prob.num_cameras:
        raise ValueError("BAL holds one pinhole camera [f, k1, k2] per view")
    nc = prob.num_cameras
    flip = np.diag([1.0, -1.0, -1.0])
    Rt = Rotation.from_rotvec(prob.extrinsics[:, 3:6]).as_matrix()
    C = prob.extrinsics[:, :3]
    tp = -np.einsum("nij,nj->ni", Rt, C)          # t' = -R_theia C
    Rb = np.einsum("ij,njk->nik", flip, Rt)       # R = diag(1,-1,-1) R_theia
    t = tp @ flip.T
    aa = Rotation.from_matrix(Rb).as_rotvec()
    K = prob.intrinsics.reshape(nc, 7)[prob.camera_group]
    pts = prob.points[:, :3] / prob.points[:, 3:4]
    with open(path, "w") as fh:
        fh.write(f"{nc} {prob.num_points} {prob.num_observations}\n")
        for c, p, (x, y) in zip(prob.obs_camera, prob.obs_point, prob.obs_xy):
            fh.write(f"{int(c)} {int(p)} {x:.17g} {-y:.17g}\n")
        for i in range(nc):
            for val in (*aa[i], *t[i], K[i, 0], K[i, 5], K[i, 6]):
                fh.write(f"{val:.17g}\n")
        for X in pts:
            for val in X:
                fh.write(f"{val:.17g}\n")
