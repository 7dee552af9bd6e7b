"""On-disk formats either side of the BA path.

* ``read_theia_reconstruction`` / ``write_theia_reconstruction``: TheiaSfM's cereal
  portable-binary ``Reconstruction`` (reference reader / writer:
  src/theia/io/reconstruction_reader.cc:53-77, reconstruction_writer.cc; class layouts
  reconstruction.h:158-167, view.h:91-94, track.h:80-83, camera.h:206-245,
  camera_intrinsics_model.h:214-218 and the five model headers, camera_intrinsics_prior.h:58-136,
  io/eigen_serializable.h:47-59).  The reader keeps EVERY field (names, priors, colours, the
  order of every hash container as stored); the writer builds the archive from those fields from
  scratch -- cereal's rules restated: class version tags at the first occurrence of a type,
  polymorphic ids / names and shared-pointer ids at first use -- so an archive that was read is
  reproduced byte for byte, and a reconstruction assembled in memory
  (``reconstruction_from_problem``: a BAL problem, a synthetic scene, ...) becomes a file the
  real Theia + Ceres can load off-box.  All five camera models; Camera v0 / v1;
  CameraIntrinsicsPrior v4.
* ``read_bundler``: Bundler ``bundle.out`` (+ optional list file) converted exactly as the
  reference's importer does (src/theia/io/read_bundler_files.cc:88-200).
* ``read_bal`` / ``write_bal``: Bundle-Adjustment-in-the-Large text problems in the same
  conventions.
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

# cereal polymorphic names of the intrinsics models (CEREAL_REGISTER_TYPE in each model header)
# indexed by CameraIntrinsicsModelType (camera_intrinsics_model_type.h:45-52)
MODEL_CEREAL_NAMES = ("theia::PinholeCameraModel", "theia::PinholeRadialTangentialCameraModel",
                      "theia::FisheyeCameraModel", "theia::FOVCameraModel",
                      "theia::DivisionUndistortionCameraModel")
MODEL_PRIOR_STRINGS = ("PINHOLE", "PINHOLE_RADIAL_TANGENTIAL", "FISHEYE", "FOV", "DIVISION_UNDISTORTION")
# CEREAL_CLASS_VERSION of what the current reference writes
CURRENT_VERSIONS = {"Reconstruction": 0, "View": 0, "Camera": 1, "CameraIntrinsicsModel": 0,
                    "CameraIntrinsicsPrior": 4, "Track": 0,
                    "theia::PinholeCameraModel": 1, "theia::PinholeRadialTangentialCameraModel": 0,
                    "theia::FisheyeCameraModel": 0, "theia::FOVCameraModel": 0,
                    "theia::DivisionUndistortionCameraModel": 0}
# CameraIntrinsicsPrior v4 members in serialisation order: (name, N of Prior<N>)
PRIOR_FIELDS = (("focal_length", 1), ("principal_point", 2), ("aspect_ratio", 1), ("skew", 1),
                ("radial_distortion", 4), ("tangential_distortion", 2), ("position", 3),
                ("orientation", 3), ("latitude", 1), ("longitude", 1), ("altitude", 1))
_TAIL = (("tangential_distortion", 2), ("position", 3), ("orientation", 3), ("latitude", 1),
         ("longitude", 1), ("altitude", 1))


def _prior_layout(version: int):
    """(has image size, has model string, Prior members) of a CameraIntrinsicsPrior class version
    (camera_intrinsics_prior.h:102-136: the legacy branches keep their own member lists)."""
    if version >= 4:
        return True, True, PRIOR_FIELDS
    if version == 3:
        return True, True, (("focal_length", 1), ("aspect_ratio", 1), ("skew", 1), ("radial_distortion", 4)) + _TAIL
    if version == 2:
        return True, False, (("focal_length", 1), ("aspect_ratio", 1), ("skew", 1), ("old_radial_distortion", 2)) + _TAIL
    return version >= 1, False, (("focal_length", 1), ("ppx", 1), ("ppy", 1), ("aspect_ratio", 1), ("skew", 1),
                                 ("rd1", 1), ("rd2", 1))


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


class _Writer:
    def __init__(self):
        self.parts = []

    def put(self, fmt: str, *vals):
        self.parts.append(struct.pack("<" + fmt, *vals))

    def string(self, s: str):
        b = s.encode("utf-8")
        self.put("Q", len(b))
        self.parts.append(b)

    def f64(self, a):
        self.parts.append(np.ascontiguousarray(a, dtype="<f8").tobytes())

    def bytes(self) -> bytes:
        return b"".join(self.parts)


@dataclass
class TheiaPrior:
    """CameraIntrinsicsPrior (camera_intrinsics_prior.h:76-107)."""
    image_width: int = 0
    image_height: int = 0
    camera_intrinsics_model_type: str = "PINHOLE"
    # member name -> (is_set, [values]); members as in PRIOR_FIELDS
    priors: dict = field(default_factory=lambda: {n: (False, [0.0] * k) for n, k in PRIOR_FIELDS})


@dataclass
class TheiaView:
    name: str
    is_estimated: bool
    extrinsics: np.ndarray           # [6]
    intrinsics_ptr: int              # shared pointer id (same id = shared block); < 0: private (Camera v0)
    intrinsics: np.ndarray           # model parameters
    image_size: tuple
    features: dict                   # track id -> (x, y), in stored order
    offsets: dict = field(default_factory=dict)   # (kept for callers that looked at it; unused)
    model: int = abi.PINHOLE
    prior: TheiaPrior = field(default_factory=TheiaPrior)


@dataclass
class TheiaTrack:
    is_estimated: bool
    view_ids: list                   # in stored (unordered_set) order
    point: np.ndarray                # [4]
    offsets: dict = field(default_factory=dict)
    color: tuple = (0, 0, 0)


@dataclass
class TheiaReconstruction:
    views: dict = field(default_factory=dict)      # view id -> TheiaView, in stored order
    tracks: dict = field(default_factory=dict)     # track id -> TheiaTrack, in stored order
    view_to_group: dict = field(default_factory=dict)
    groups: dict = field(default_factory=dict)     # group id -> [view ids] in stored order
    raw: bytes = b""                 # the archive as read (informational)
    intrinsics_offsets: dict = field(default_factory=dict)
    next_track_id: int = 0
    next_view_id: int = 0
    view_names: dict = field(default_factory=dict)  # name -> view id, in stored order
    versions: dict = field(default_factory=lambda: dict(CURRENT_VERSIONS))


def read_theia_reconstruction(path: str) -> TheiaReconstruction:
    r = _Reader(open(path, "rb").read())
    if r.take("B") != 1:
        raise ValueError("big-endian cereal archives are not supported")
    versions = {}

    def ver(tag):  # cereal writes a class version the first time a type appears
        if tag not in versions:
            versions[tag] = r.take("I")
        return versions[tag]

    rec = TheiaReconstruction()
    ver("Reconstruction")
    rec.next_track_id = r.take("I")
    rec.next_view_id = r.take("I")
    for _ in range(r.take("Q")):
        name = r.string()
        rec.view_names[name] = r.take("I")
    poly_types = {}
    shared = {}

    def prior(n):
        ver(f"Prior{n}")
        is_set = bool(r.take("B"))
        return is_set, [float(x) for x in r.f64(n)]

    nviews = r.take("Q")
    for _ in range(nviews):
        vid = r.take("I")
        ver("View")
        name = r.string()
        is_est = bool(r.take("B"))
        cam_ver = ver("Camera")
        model = abi.PINHOLE
        if cam_ver > 0:
            ext = r.f64(6)
            pid = r.take("I")
            if pid & 0x80000000:
                poly_types[pid & 0x7FFFFFFF] = r.string()
            tname = poly_types[pid & 0x7FFFFFFF]
            if tname not in MODEL_CEREAL_NAMES:
                raise ValueError(f"unsupported intrinsics type in archive: {tname}")
            model = MODEL_CEREAL_NAMES.index(tname)
            ptr = r.take("I")
            intr_id = ptr & 0x7FFFFFFF
            if ptr & 0x80000000:
                # derived class (versioned), which serialises its base class (versioned) =
                # std::vector<double> parameters_; PinholeCameraModel v0 wrote the raw doubles
                if tname != MODEL_CEREAL_NAMES[0] or ver(tname) > 0:
                    if tname != MODEL_CEREAL_NAMES[0]:
                        ver(tname)
                    ver("CameraIntrinsicsModel")
                    n = r.take("Q")
                    shared[intr_id] = (model, r.f64(n))
                else:
                    shared[intr_id] = (model, r.f64(abi.INTRINSICS_SIZE[model]))
            model, intr = shared[intr_id]
            size = r.take("ii")
        else:  # Camera v0: 13 doubles [extrinsics(6), pinhole intrinsics(7)] + size
            p = r.f64(13)
            ext, intr = p[:6], p[6:]
            intr_id = -1 - vid
            size = r.take("ii")
        has_size, has_string, members = _prior_layout(ver("CameraIntrinsicsPrior"))
        pr = TheiaPrior()
        if has_size:
            pr.image_width, pr.image_height = r.take("ii")
        if has_string:
            pr.camera_intrinsics_model_type = r.string()
        pr.priors = {}
        for fname, n in members:
            pr.priors[fname] = prior(n)
        feats = {}
        for _ in range(r.take("Q")):
            tid = r.take("I")
            rows, cols = r.take("ii")
            xy = r.f64(rows * cols)
            feats[tid] = (float(xy[0]), float(xy[1]))
        rec.views[vid] = TheiaView(name, is_est, ext, intr_id, intr, size, feats, {}, model, pr)
    for _ in range(r.take("Q")):
        tid = r.take("I")
        ver("Track")
        is_est = bool(r.take("B"))
        vids = [r.take("I") for _ in range(r.take("Q"))]
        rows, cols = r.take("ii")
        pt = r.f64(rows * cols)
        rows, cols = r.take("ii")
        col = tuple(r.d[r.o:r.o + rows * cols])
        r.o += rows * cols
        rec.tracks[tid] = TheiaTrack(is_est, vids, pt, {}, col)
    for _ in range(r.take("Q")):
        k, v = r.take("II")
        rec.view_to_group[k] = v
    for _ in range(r.take("Q")):
        g = r.take("I")
        rec.groups[g] = [r.take("I") for _ in range(r.take("Q"))]
    if r.o != len(r.d):
        raise ValueError(f"trailing bytes in archive: parsed {r.o} of {len(r.d)}")
    rec.raw = r.d
    rec.versions = {**CURRENT_VERSIONS, **{k: v for k, v in versions.items() if not k.startswith("Prior")}}
    return rec


def write_theia_reconstruction(path: str, rec: TheiaReconstruction) -> None:
    """Serialises `rec` as a cereal PortableBinaryOutputArchive would
    (src/theia/io/reconstruction_writer.cc -> Reconstruction::serialize, reconstruction.h:158-167):
    little-endian flag byte; members in declaration order; a uint32 class version the first
    time each versioned type is written; containers as uint64 size + elements in the order
    held here; the polymorphic shared_ptr<CameraIntrinsicsModel> of a Camera (camera.h:206-245)
    as [polymorphic id (MSB + type name at first use) | pointer id (MSB + pointee at first use)].
    Class versions come from rec.versions (what the archive had when it was read, else the
    reference's current ones), so reading an archive and writing it back is byte identical."""
    w = _Writer()
    seen = set()

    def ver(tag, default=None):
        if tag not in seen:
            seen.add(tag)
            w.put("I", int(rec.versions.get(tag, CURRENT_VERSIONS.get(tag, 0) if default is None else default)))
        return rec.versions.get(tag, CURRENT_VERSIONS.get(tag, 0) if default is None else default)

    w.put("B", 1)
    ver("Reconstruction")
    w.put("I", int(rec.next_track_id))
    w.put("I", int(rec.next_view_id))
    names = rec.view_names if rec.view_names else {v.name: vid for vid, v in rec.views.items()}
    w.put("Q", len(names))
    for name, vid in names.items():
        w.string(name)
        w.put("I", int(vid))
    poly_ids = {}
    ptr_ids = {}

    def prior(n, val):
        ver(f"Prior{n}", 0)
        is_set, values = val
        w.put("B", 1 if is_set else 0)
        w.f64(np.asarray(values, float).reshape(n))

    w.put("Q", len(rec.views))
    for vid, view in rec.views.items():
        w.put("I", int(vid))
        ver("View")
        w.string(view.name)
        w.put("B", 1 if view.is_estimated else 0)
        cam_ver = ver("Camera")
        if cam_ver > 0:
            w.f64(np.asarray(view.extrinsics, float).reshape(6))
            tname = MODEL_CEREAL_NAMES[view.model]
            if tname not in poly_ids:
                poly_ids[tname] = len(poly_ids) + 1
                w.put("I", poly_ids[tname] | 0x80000000)
                w.string(tname)
            else:
                w.put("I", poly_ids[tname])
            key = view.intrinsics_ptr
            if key not in ptr_ids:
                ptr_ids[key] = len(ptr_ids) + 1
                w.put("I", ptr_ids[key] | 0x80000000)
                vals = np.asarray(view.intrinsics, float).reshape(-1)
                if ver(tname) > 0 or tname != MODEL_CEREAL_NAMES[0]:
                    ver("CameraIntrinsicsModel")
                    w.put("Q", vals.size)
                w.f64(vals)
            else:
                w.put("I", ptr_ids[key])
        else:
            if view.model != abi.PINHOLE:
                raise ValueError("Camera v0 archives hold pinhole cameras only (camera.h:215-220)")
            w.f64(np.concatenate([np.asarray(view.extrinsics, float).reshape(6),
                                  np.asarray(view.intrinsics, float).reshape(7)]))
        w.put("ii", int(view.image_size[0]), int(view.image_size[1]))
        has_size, has_string, members = _prior_layout(ver("CameraIntrinsicsPrior"))
        pr = view.prior
        if has_size:
            w.put("ii", int(pr.image_width), int(pr.image_height))
        if has_string:
            w.string(pr.camera_intrinsics_model_type)
        for fname, n in members:
            prior(n, pr.priors.get(fname, (False, [0.0] * n)))
        w.put("Q", len(view.features))
        for tid, (x, y) in view.features.items():
            w.put("I", int(tid))
            w.put("ii", 2, 1)
            w.put("dd", float(x), float(y))
    w.put("Q", len(rec.tracks))
    for tid, tr in rec.tracks.items():
        w.put("I", int(tid))
        ver("Track")
        w.put("B", 1 if tr.is_estimated else 0)
        w.put("Q", len(tr.view_ids))
        for v in tr.view_ids:
            w.put("I", int(v))
        w.put("ii", 4, 1)
        w.f64(np.asarray(tr.point, float).reshape(4))
        w.put("ii", 3, 1)
        w.put("BBB", *[int(c) & 0xFF for c in tr.color])
    w.put("Q", len(rec.view_to_group))
    for k, v in rec.view_to_group.items():
        w.put("II", int(k), int(v))
    w.put("Q", len(rec.groups))
    for g, vids in rec.groups.items():
        w.put("I", int(g))
        w.put("Q", len(vids))
        for v in vids:
            w.put("I", int(v))
    with open(path, "wb") as fh:
        fh.write(w.bytes())


def reconstruction_from_problem(prob: Problem, view_names=None, image_size=(0, 0)) -> TheiaReconstruction:
    """A Reconstruction holding `prob` (every view and track estimated, view id = camera
    index, track id = point index, intrinsics group id = problem group), with the current
    class versions -- the hand-off of a BAL / synthetic problem to a real Theia binary."""
    rec = TheiaReconstruction()
    nc, npt = prob.num_cameras, prob.num_points
    rec.next_view_id, rec.next_track_id = nc, npt
    feats = [dict() for _ in range(nc)]
    tviews = [[] for _ in range(npt)]
    for c, p, (x, y) in zip(prob.obs_camera.tolist(), prob.obs_point.tolist(), prob.obs_xy.tolist()):
        feats[c][p] = (x, y)
        tviews[p].append(c)
    for c in range(nc):
        g = int(prob.camera_group[c])
        a, b = int(prob.group_offset[g]), int(prob.group_offset[g + 1])
        model = int(prob.group_model[g])
        name = view_names[c] if view_names is not None else f"view_{c:06d}"
        pr = TheiaPrior(image_width=int(image_size[0]), image_height=int(image_size[1]),
                        camera_intrinsics_model_type=MODEL_PRIOR_STRINGS[model])
        rec.views[c] = TheiaView(name, True, prob.extrinsics[c].copy(), g, prob.intrinsics[a:b].copy(),
                                 tuple(image_size), feats[c], {}, model, pr)
        rec.view_names[name] = c
        rec.view_to_group[c] = g
        rec.groups.setdefault(g, []).append(c)
    for p in range(npt):
        rec.tracks[p] = TheiaTrack(True, tviews[p], prob.points[p].copy(), {}, (0, 0, 0))
    return rec


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
    ext = np.stack([rec.views[v].extrinsics for v in vids]) if vids else np.zeros((0, 6))
    cam_group = np.array([gindex[keyf(v)] for v in vids], dtype=np.int32)
    grp_model = np.zeros(len(gkeys), np.int32)
    grp_vals = [None] * len(gkeys)
    for v in vids:
        g = gindex[keyf(v)]
        grp_model[g] = rec.views[v].model
        grp_vals[g] = np.asarray(rec.views[v].intrinsics, float)
    sizes = np.array([abi.INTRINSICS_SIZE[m] for m in grp_model], dtype=np.int64)
    group_offset = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    intr = np.concatenate(grp_vals) if grp_vals else np.zeros(0)
    tids = sorted(t for t, tr in rec.tracks.items()
                  if tr.is_estimated and any(v in cam_index for v in tr.view_ids))
    pts = np.stack([rec.tracks[t].point for t in tids]) if tids else np.zeros((0, 4))
    oc, op, oxy = [], [], []
    for pi, t in enumerate(tids):
        for v in sorted(rec.tracks[t].view_ids):
            if v in cam_index and t in rec.views[v].features:
                oc.append(cam_index[v])
                op.append(pi)
                oxy.append(rec.views[v].features[t])
    prob = Problem(
        extrinsics=ext, camera_group=cam_group, camera_flags=np.zeros(len(vids), np.uint8),
        group_model=grp_model, group_offset=group_offset,
        intrinsics=intr, intrinsics_constant=np.zeros(intr.shape[0], np.uint8),
        points=pts, point_constant=np.zeros(len(tids), np.uint8),
        obs_camera=np.array(oc, np.int32), obs_point=np.array(op, np.int32),
        obs_xy=np.array(oxy, np.float64).reshape(-1, 2))
    prob.set_intrinsics_to_optimize(intrinsics_to_optimize)
    prob.meta["view_ids"] = vids
    prob.meta["track_ids"] = tids
    return prob


# ---- Bundler -------------------------------------------------------------------
def read_bundler(bundle_path: str, list_path: str | None = None) -> TheiaReconstruction:
    """Bundler `bundle.out` (v0.3) [+ image list] -> Reconstruction, as the reference's importer
    (src/theia/io/read_bundler_files.cc:88-200 on top of bundler_file_reader.cc:60-293):

    * R_theia = diag(1,-1,-1) R, t' = diag(1,-1,-1) t, C = -R_theia^T t'; PINHOLE with
      [f, 1, 0, 0, 0, k1, k2]; feature (x, -y); point homogeneous w = 1; track estimated;
    * the importer's own quirks are kept, since parity with a Theia that loaded the same file
      is the point: f, k1, k2 pass through `float` (bundler_file_reader.h:53-55), keypoint
      coordinates are truncated to integers (FeatureInfo::kpt_x / kpt_y are `int`,
      bundler_file_reader.h:67-68, cc:150-160), point colours are cast to uint8;
    * a camera with focal length <= 0 is dropped together with its observations; a track is
      skipped when fewer than two observations remain, when its position is exactly 0, or when
      a view appears in it twice (Reconstruction::AddTrack fails); track ids are consecutive
      over the tracks that were added (reconstruction.cc:283-320);
    * view names come from the list file (file name without directory; a third column is the
      EXIF focal length prior) or are "view_<index>" when no list is given."""
    with open(bundle_path, "rt") as fh:
        text = fh.read()
    lines = text.split("\n", 1)
    tok = (lines[1] if lines[0].lstrip().startswith("#") else text).split()
    ncam, npt = int(tok[0]), int(tok[1])
    o = 2
    names, focal_prior = [], []
    if list_path is not None:
        for line in open(list_path, "rt"):
            parts = line.split()
            if not parts:
                continue
            if len(parts) not in (1, 3):
                raise ValueError(f"invalid list line: {line!r}")
            names.append(parts[0].replace("\\", "/").rsplit("/", 1)[-1])
            focal_prior.append(float(np.float32(parts[2])) if len(parts) == 3 else 0.0)
        if len(names) != ncam:
            raise ValueError("the list file and the bundle file disagree on the number of cameras")
    else:
        names = [f"view_{i:06d}" for i in range(ncam)]
        focal_prior = [0.0] * ncam
    rec = TheiaReconstruction()
    flip = np.diag([1.0, -1.0, -1.0])
    removed = set()
    for i in range(ncam):
        f, k1, k2 = (float(np.float32(tok[o + j])) for j in range(3))
        Rb = np.array(tok[o + 3:o + 12], dtype=np.float64).reshape(3, 3)
        t = np.array(tok[o + 12:o + 15], dtype=np.float64)
        o += 15
        Rt = flip @ Rb
        C = -Rt.T @ (flip @ t)
        # SetOrientationFromRotationMatrix -> ceres::RotationMatrixToAngleAxis
        aa = Rotation.from_matrix(Rt).as_rotvec() if abs(np.linalg.det(Rt) - 1.0) < 1e-6 else np.zeros(3)
        intr = np.array([1.0, 1.0, 0.0, 0.0, 0.0, k1, k2])  # PinholeCameraModel defaults (f = 1)
        if f <= 0.0:
            removed.add(i)
        else:
            intr[0] = f
        pr = TheiaPrior()
        if focal_prior[i] > 0.0:
            pr.priors["focal_length"] = (True, [focal_prior[i]])
        rec.views[i] = TheiaView(names[i], True, np.concatenate([C, aa]), i, intr, (0, 0), {}, {},
                                 abi.PINHOLE, pr)
        rec.view_names[names[i]] = i
        rec.view_to_group[i] = i
        rec.groups[i] = [i]
    rec.next_view_id = ncam
    tid = 0
    for _ in range(npt):
        X = np.array(tok[o:o + 3], dtype=np.float64)
        col = tuple(int(float(c)) & 0xFF for c in tok[o + 3:o + 6])
        nv = int(tok[o + 6])
        o += 7
        obs = []
        for _ in range(nv):
            cam = int(np.float32(tok[o]))
            x, y = int(np.float32(tok[o + 2])), int(np.float32(tok[o + 3]))  # truncation, as the reference
            o += 4
            if cam not in removed:
                obs.append((cam, (float(x), float(-y))))
        if len(obs) < 2 or float(X @ X) == 0.0:
            continue
        cams = [c for c, _ in obs]
        if len(set(cams)) != len(cams):
            continue
        rec.tracks[tid] = TheiaTrack(True, cams, np.array([X[0], X[1], X[2], 1.0]), {}, col)
        for c, xy in obs:
            rec.views[c].features[tid] = xy
        tid += 1
    rec.next_track_id = tid
    for i in removed:  # Reconstruction::RemoveView
        v = rec.views.pop(i)
        rec.view_names.pop(v.name, None)
        rec.view_to_group.pop(i, None)
        rec.groups.pop(i, None)
        for t in list(v.features):
            tr = rec.tracks.get(t)
            if tr is not None and i in tr.view_ids:
                tr.view_ids.remove(i)
    return rec


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
