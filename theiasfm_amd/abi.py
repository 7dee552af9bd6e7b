"""ctypes mirror of ``include/theia_mi355_ba.h`` and the flattened-problem container.

Python is only plumbing here (tests, bench, the torch.distributed hook).  The
layout of every struct below must match the C header field for field; the
``-m "not gpu"`` suite checks sizes against the compiled library.

Reference types these mirror:
  BundleAdjustmentOptions / BundleAdjustmentSummary
      src/theia/sfm/bundle_adjustment/bundle_adjustment.h:78-133
  CameraIntrinsicsModelType  src/theia/sfm/camera/camera_intrinsics_model_type.h:45-52
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

# ---- enums ------------------------------------------------------------------
PINHOLE, PINHOLE_RADIAL_TANGENTIAL, FISHEYE, FOV, DIVISION_UNDISTORTION = range(5)
INTRINSICS_SIZE = (7, 10, 9, 5, 5)

LOSS_TRIVIAL, LOSS_HUBER, LOSS_SOFTLONE, LOSS_CAUCHY, LOSS_ARCTAN, LOSS_TUKEY = range(6)

DENSE_QR, DENSE_SCHUR, SPARSE_SCHUR, ITERATIVE_SCHUR, CGNR = 1, 3, 4, 5, 6
PRECOND_IDENTITY, PRECOND_JACOBI, PRECOND_SCHUR_JACOBI = 0, 1, 2
PRECOND_CLUSTER_JACOBI, PRECOND_CLUSTER_TRIDIAGONAL = 3, 4  # clusters = shared intrinsics block + its views (or visibility clusters); TRIDIAGONAL: + the blocks between neighbours of the degree-2 spanning forest
PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS = 18  # Ceres's own block shape (6x6 + NxN per view)
SCHUR_AUTO, SCHUR_EXPLICIT, SCHUR_IMPLICIT = 0, 1, 2
CANONICAL_VIEWS, SINGLE_LINKAGE = 0, 1  # ceres::VisibilityClusteringType

INTRINSICS_NONE = 0x00
INTRINSICS_FOCAL_LENGTH = 0x01
INTRINSICS_ASPECT_RATIO = 0x02
INTRINSICS_SKEW = 0x04
INTRINSICS_PRINCIPAL_POINTS = 0x08
INTRINSICS_RADIAL_DISTORTION = 0x10
INTRINSICS_TANGENTIAL_DISTORTION = 0x20
INTRINSICS_ALL = 0x3F
# BundleAdjustmentOptions default (bundle_adjustment.h:101-103)
INTRINSICS_DEFAULT = INTRINSICS_FOCAL_LENGTH | INTRINSICS_RADIAL_DISTORTION

CAMERA_POSITION_CONSTANT = 0x1
CAMERA_ORIENTATION_CONSTANT = 0x2

NUM_KERNEL_CLASSES = 12
KERNEL_CLASS_NAMES = (
    "linearize", "point_eliminate", "camera_diag", "schur_offdiag", "preconditioner",
    "spmv", "pcg_vector", "cholesky", "back_substitute", "update_cost", "reduce",
    "allreduce",
)

ERR_INVALID_ARGUMENT = 1
ERR_UNSUPPORTED = 5

STATUS_NAMES = {
    0: "OK", 1: "INVALID_ARGUMENT", 2: "NO_DEVICE", 3: "DEVICE", 4: "OUT_OF_MEMORY",
    5: "UNSUPPORTED", 6: "EVALUATION_FAILED", 7: "LINEAR_SOLVER", 8: "COLLECTIVE",
}


class CProblem(C.Structure):
    _fields_ = [
        ("num_cameras", C.c_int32),
        ("extrinsics", C.POINTER(C.c_double)),
        ("camera_group", C.POINTER(C.c_int32)),
        ("camera_flags", C.POINTER(C.c_uint8)),
        ("num_groups", C.c_int32),
        ("group_model", C.POINTER(C.c_int32)),
        ("group_offset", C.POINTER(C.c_int32)),
        ("intrinsics", C.POINTER(C.c_double)),
        ("intrinsics_constant", C.POINTER(C.c_uint8)),
        ("num_points", C.c_int32),
        ("points", C.POINTER(C.c_double)),
        ("point_constant", C.POINTER(C.c_uint8)),
        ("num_observations", C.c_int64),
        ("obs_camera", C.POINTER(C.c_int32)),
        ("obs_point", C.POINTER(C.c_int32)),
        ("obs_xy", C.POINTER(C.c_double)),
    ]


class COptions(C.Structure):
    _fields_ = [
        ("loss_function_type", C.c_int32),
        ("robust_loss_width", C.c_double),
        ("linear_solver_type", C.c_int32),
        ("preconditioner_type", C.c_int32),
        ("verbose", C.c_int32),
        ("num_threads", C.c_int32),
        ("max_num_iterations", C.c_int32),
        ("max_solver_time_in_seconds", C.c_double),
        ("use_inner_iterations", C.c_int32),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("eta", C.c_double),
        ("max_linear_solver_iterations", C.c_int32),
        ("min_linear_solver_iterations", C.c_int32),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("point_dof", C.c_int32),
        ("device", C.c_int32),
        ("profile_kernels", C.c_int32),
        ("residual_precision", C.c_int32),
        ("schur_mode", C.c_int32),
        ("visibility_clustering_type", C.c_int32),
        ("iteration_trace", C.POINTER(C.c_double)),
        ("iteration_trace_capacity", C.c_int32),
    ]

TRACE_STRIDE = 8
TRACE_FIELDS = ("iteration", "cost", "radius", "outcome", "candidate_cost", "model_cost_change", "linear_iterations", "step_norm")


class CSummary(C.Structure):
    _fields_ = [
        ("success", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("setup_time_in_seconds", C.c_double),
        ("solve_time_in_seconds", C.c_double),
        ("status", C.c_int32),
        ("termination", C.c_int32),
        ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("num_unsuccessful_steps", C.c_int32),
        ("num_linear_solver_iterations", C.c_int64),
        ("final_rmse", C.c_double),
        ("initial_rmse", C.c_double),
        ("num_reduced_blocks", C.c_int32),
        ("reduced_block_dim", C.c_int32),
        ("num_schur_blocks", C.c_int64),
        ("num_schur_pairs", C.c_int64),
        ("num_inner_iteration_steps", C.c_int32),
        ("num_matrix_free_iterations", C.c_int32),
        ("kernel_launches", C.c_int64 * NUM_KERNEL_CLASSES),
        ("kernel_seconds", C.c_double * NUM_KERNEL_CLASSES),
        ("message", C.c_char * 192),
        ("effective_preconditioner_type", C.c_int32),
    ]

    def as_dict(self) -> dict:
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            if name == "message":
                v = v.decode("utf-8", "replace")
            elif name in ("kernel_launches", "kernel_seconds"):
                v = list(v)
            d[name] = v
        return d


class CFilterSummary(C.Structure):
    _fields_ = [
        ("num_estimated_tracks", C.c_int64),
        ("num_bad_reprojections", C.c_int64),
        ("num_insufficient_viewing_angles", C.c_int64),
        ("seconds", C.c_double),
        ("kernel_seconds", C.c_double),
    ]


class CTrackBatchSummary(C.Structure):
    _fields_ = [
        ("num_tracks", C.c_int64),
        ("num_success", C.c_int64),
        ("total_iterations", C.c_int64),
        ("seconds", C.c_double),
        ("kernel_seconds", C.c_double),
    ]


class CTwoViewBatch(C.Structure):
    _fields_ = [
        ("num_pairs", C.c_int32),
        ("extrinsics1", C.POINTER(C.c_double)),
        ("extrinsics2", C.POINTER(C.c_double)),
        ("model1", C.POINTER(C.c_int32)),
        ("model2", C.POINTER(C.c_int32)),
        ("intrinsics1", C.POINTER(C.c_double)),
        ("intrinsics2", C.POINTER(C.c_double)),
        ("constant_intrinsics1", C.POINTER(C.c_uint8)),
        ("constant_intrinsics2", C.POINTER(C.c_uint8)),
        ("correspondence_ptr", C.POINTER(C.c_int64)),
        ("features1", C.POINTER(C.c_double)),
        ("features2", C.POINTER(C.c_double)),
        ("points", C.POINTER(C.c_double)),
    ]


class CTwoViewAngularBatch(C.Structure):
    _fields_ = [
        ("num_pairs", C.c_int32),
        ("rotation2", C.POINTER(C.c_double)),
        ("position2", C.POINTER(C.c_double)),
        ("correspondence_ptr", C.POINTER(C.c_int64)),
        ("features1", C.POINTER(C.c_double)),
        ("features2", C.POINTER(C.c_double)),
    ]


class CSelectSummary(C.Structure):
    _fields_ = [
        ("num_tracks", C.c_int64),
        ("num_selected", C.c_int64),
        ("num_selected_grid", C.c_int64),
        ("seconds", C.c_double),
        ("kernel_seconds", C.c_double),
    ]


def attach_trace(options: "COptions", capacity: int) -> np.ndarray:
    """Gives `options` a per-iteration trace buffer (tmi_ba_options.iteration_trace) and returns it as a
    [capacity, TRACE_STRIDE] array (rows beyond summary.num_iterations stay NaN).  The array owns the memory: keep it
    alive for as long as `options` is used."""
    buf = np.full((capacity, TRACE_STRIDE), np.nan)
    options.iteration_trace = buf.ctypes.data_as(C.POINTER(C.c_double))
    options.iteration_trace_capacity = capacity
    return buf


def default_options(**overrides) -> COptions:
    """BundleAdjustmentOptions defaults (bundle_adjustment.h:78-122) plus the
    Ceres defaults Theia inherits.  Mirrors tmi_ba_options_init."""
    o = COptions()
    o.loss_function_type = LOSS_TRIVIAL
    o.robust_loss_width = 2.0
    o.linear_solver_type = SPARSE_SCHUR
    o.preconditioner_type = PRECOND_SCHUR_JACOBI
    o.verbose = 0
    o.num_threads = 1
    o.max_num_iterations = 100
    o.max_solver_time_in_seconds = 3600.0
    o.use_inner_iterations = 1
    o.function_tolerance = 1e-6
    o.gradient_tolerance = 1e-10
    o.parameter_tolerance = 1e-8
    o.max_trust_region_radius = 1e12
    o.initial_trust_region_radius = 1e4
    o.min_trust_region_radius = 1e-32
    o.min_relative_decrease = 1e-3
    o.min_lm_diagonal = 1e-6
    o.max_lm_diagonal = 1e32
    o.eta = 0.1
    o.max_linear_solver_iterations = 500
    o.min_linear_solver_iterations = 0
    o.max_num_consecutive_invalid_steps = 5
    o.jacobi_scaling = 1
    o.point_dof = 4
    o.device = -1
    o.profile_kernels = 0
    o.residual_precision = 64
    o.schur_mode = 0
    o.visibility_clustering_type = 0
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise AttributeError(f"tmi_ba_options has no field {k!r}")
        setattr(o, k, v)
    return o


def intrinsics_constant_mask(model: int, bits: int) -> np.ndarray:
    """GetSubsetFromOptimizeIntrinsicsType as a 0/1 mask (1 = held constant).

    reference: pinhole_camera_model.cc:132-162, pinhole_radial_tangential_camera_model.cc:150-185,
    fisheye_camera_model.cc:142-175, fov_camera_model.cc:124-149,
    division_undistortion_camera_model.cc:126-150.  (Host-side twin of
    tmi_ba_intrinsics_constant_mask; the test-suite checks they agree.)"""
    n = INTRINSICS_SIZE[model]
    m = np.zeros(n, dtype=np.uint8)
    if bits == INTRINSICS_ALL:
        return m
    no = lambda b: 0 if (bits & b) else 1  # noqa: E731
    if model in (PINHOLE, PINHOLE_RADIAL_TANGENTIAL, FISHEYE):
        m[0] = no(INTRINSICS_FOCAL_LENGTH)
        m[1] = no(INTRINSICS_ASPECT_RATIO)
        m[2] = no(INTRINSICS_SKEW)
        m[3] = m[4] = no(INTRINSICS_PRINCIPAL_POINTS)
        if model == PINHOLE:
            m[5:7] = no(INTRINSICS_RADIAL_DISTORTION)
        elif model == PINHOLE_RADIAL_TANGENTIAL:
            m[5:8] = no(INTRINSICS_RADIAL_DISTORTION)
            m[8:10] = no(INTRINSICS_TANGENTIAL_DISTORTION)
        else:
            m[5:9] = no(INTRINSICS_RADIAL_DISTORTION)
    else:
        m[0] = no(INTRINSICS_FOCAL_LENGTH)
        m[1] = no(INTRINSICS_ASPECT_RATIO)
        m[2] = m[3] = no(INTRINSICS_PRINCIPAL_POINTS)
        m[4] = no(INTRINSICS_RADIAL_DISTORTION)
    return m


def _ptr(a: Optional[np.ndarray], ctype):
    if a is None:
        return C.cast(None, C.POINTER(ctype))
    return a.ctypes.data_as(C.POINTER(ctype))


@dataclass
class Problem:
    """Flattened reconstruction: the SoA arrays of ``tmi_ba_problem``.

    All arrays are owned here (C-contiguous numpy); ``as_c()`` hands out a
    struct of borrowed pointers that is valid while this object is alive."""

    extrinsics: np.ndarray            # [Nc, 6]  position(3), angle-axis(3)
    camera_group: np.ndarray          # [Nc] int32
    camera_flags: np.ndarray          # [Nc] uint8
    group_model: np.ndarray           # [G] int32
    group_offset: np.ndarray          # [G+1] int32
    intrinsics: np.ndarray            # [sum sizes] float64
    intrinsics_constant: np.ndarray   # [sum sizes] uint8
    points: np.ndarray                # [Np, 4]
    point_constant: np.ndarray        # [Np] uint8
    obs_camera: np.ndarray            # [No] int32
    obs_point: np.ndarray             # [No] int32
    obs_xy: np.ndarray                # [No, 2]
    meta: dict = field(default_factory=dict)

    def __post_init__(self):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)  # noqa: E731
        u8 = lambda a: np.ascontiguousarray(a, dtype=np.uint8)  # noqa: E731
        self.extrinsics = f64(self.extrinsics).reshape(-1, 6)
        self.camera_group = i32(self.camera_group)
        self.camera_flags = u8(self.camera_flags)
        self.group_model = i32(self.group_model)
        self.group_offset = i32(self.group_offset)
        self.intrinsics = f64(self.intrinsics).reshape(-1)
        self.intrinsics_constant = u8(self.intrinsics_constant)
        self.points = f64(self.points).reshape(-1, 4)
        self.point_constant = u8(self.point_constant)
        self.obs_camera = i32(self.obs_camera)
        self.obs_point = i32(self.obs_point)
        self.obs_xy = f64(self.obs_xy).reshape(-1, 2)

    # sizes
    @property
    def num_cameras(self) -> int:
        return self.extrinsics.shape[0]

    @property
    def num_groups(self) -> int:
        return self.group_model.shape[0]

    @property
    def num_points(self) -> int:
        return self.points.shape[0]

    @property
    def num_observations(self) -> int:
        return self.obs_camera.shape[0]

    def copy(self) -> "Problem":
        return Problem(
            self.extrinsics.copy(), self.camera_group.copy(), self.camera_flags.copy(),
            self.group_model.copy(), self.group_offset.copy(), self.intrinsics.copy(),
            self.intrinsics_constant.copy(), self.points.copy(), self.point_constant.copy(),
            self.obs_camera.copy(), self.obs_point.copy(), self.obs_xy.copy(), dict(self.meta))

    def set_intrinsics_to_optimize(self, bits: int) -> None:
        """Apply an OptimizeIntrinsicsType bitmask to every group
        (bundle_adjuster.cc:242-268)."""
        for g in range(self.num_groups):
            a, b = self.group_offset[g], self.group_offset[g + 1]
            self.intrinsics_constant[a:b] = intrinsics_constant_mask(int(self.group_model[g]), bits)

    def as_c(self) -> CProblem:
        p = CProblem()
        p.num_cameras = self.num_cameras
        p.extrinsics = _ptr(self.extrinsics, C.c_double)
        p.camera_group = _ptr(self.camera_group, C.c_int32)
        p.camera_flags = _ptr(self.camera_flags, C.c_uint8)
        p.num_groups = self.num_groups
        p.group_model = _ptr(self.group_model, C.c_int32)
        p.group_offset = _ptr(self.group_offset, C.c_int32)
        p.intrinsics = _ptr(self.intrinsics, C.c_double)
        p.intrinsics_constant = _ptr(self.intrinsics_constant, C.c_uint8)
        p.num_points = self.num_points
        p.points = _ptr(self.points, C.c_double)
        p.point_constant = _ptr(self.point_constant, C.c_uint8)
        p.num_observations = self.num_observations
        p.obs_camera = _ptr(self.obs_camera, C.c_int32)
        p.obs_point = _ptr(self.obs_point, C.c_int32)
        p.obs_xy = _ptr(self.obs_xy, C.c_double)
        return p

    def save(self, path: str) -> None:
        np.savez_compressed(
            path, extrinsics=self.extrinsics, camera_group=self.camera_group,
            camera_flags=self.camera_flags, group_model=self.group_model,
            group_offset=self.group_offset, intrinsics=self.intrinsics,
            intrinsics_constant=self.intrinsics_constant, points=self.points,
            point_constant=self.point_constant, obs_camera=self.obs_camera,
            obs_point=self.obs_point, obs_xy=self.obs_xy)

    @staticmethod
    def load(path: str) -> "Problem":
        z = np.load(path)
        return Problem(*(z[k] for k in (
            "extrinsics", "camera_group", "camera_flags", "group_model", "group_offset",
            "intrinsics", "intrinsics_constant", "points", "point_constant", "obs_camera",
            "obs_point", "obs_xy")))


@dataclass
class TwoViewBatch:
    """View pairs for the batched BundleAdjustTwoViews (``tmi_ba_two_view_batch``)."""

    extrinsics1: np.ndarray            # [P, 6] constant
    extrinsics2: np.ndarray            # [P, 6] in/out
    model1: np.ndarray                 # [P] int32
    model2: np.ndarray
    intrinsics1: np.ndarray            # [P, 10] zero padded
    intrinsics2: np.ndarray
    constant_intrinsics1: np.ndarray   # [P] uint8
    constant_intrinsics2: np.ndarray
    correspondence_ptr: np.ndarray     # [P + 1] int64
    features1: np.ndarray              # [N, 2]
    features2: np.ndarray
    points: np.ndarray                 # [N, 4]

    def __post_init__(self):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        self.extrinsics1 = f64(self.extrinsics1).reshape(-1, 6)
        self.extrinsics2 = f64(self.extrinsics2).reshape(-1, 6)
        self.model1 = np.ascontiguousarray(self.model1, dtype=np.int32)
        self.model2 = np.ascontiguousarray(self.model2, dtype=np.int32)
        self.intrinsics1 = f64(self.intrinsics1).reshape(-1, 10)
        self.intrinsics2 = f64(self.intrinsics2).reshape(-1, 10)
        self.constant_intrinsics1 = np.ascontiguousarray(self.constant_intrinsics1, dtype=np.uint8)
        self.constant_intrinsics2 = np.ascontiguousarray(self.constant_intrinsics2, dtype=np.uint8)
        self.correspondence_ptr = np.ascontiguousarray(self.correspondence_ptr, dtype=np.int64)
        self.features1 = f64(self.features1).reshape(-1, 2)
        self.features2 = f64(self.features2).reshape(-1, 2)
        self.points = f64(self.points).reshape(-1, 4)

    @property
    def num_pairs(self) -> int:
        return self.extrinsics1.shape[0]

    def copy(self) -> "TwoViewBatch":
        return TwoViewBatch(*[getattr(self, f).copy() for f in (
            "extrinsics1", "extrinsics2", "model1", "model2", "intrinsics1", "intrinsics2",
            "constant_intrinsics1", "constant_intrinsics2", "correspondence_ptr", "features1", "features2",
            "points")])

    def head(self, n: int) -> "TwoViewBatch":
        """the first n pairs (copies)"""
        n = min(int(n), self.num_pairs)
        m = int(self.correspondence_ptr[n])
        return TwoViewBatch(self.extrinsics1[:n].copy(), self.extrinsics2[:n].copy(), self.model1[:n].copy(),
                            self.model2[:n].copy(), self.intrinsics1[:n].copy(), self.intrinsics2[:n].copy(),
                            self.constant_intrinsics1[:n].copy(), self.constant_intrinsics2[:n].copy(),
                            self.correspondence_ptr[:n + 1].copy(), self.features1[:m].copy(),
                            self.features2[:m].copy(), self.points[:m].copy())

    def as_c(self) -> CTwoViewBatch:
        b = CTwoViewBatch()
        b.num_pairs = self.num_pairs
        b.extrinsics1 = _ptr(self.extrinsics1, C.c_double)
        b.extrinsics2 = _ptr(self.extrinsics2, C.c_double)
        b.model1 = _ptr(self.model1, C.c_int32)
        b.model2 = _ptr(self.model2, C.c_int32)
        b.intrinsics1 = _ptr(self.intrinsics1, C.c_double)
        b.intrinsics2 = _ptr(self.intrinsics2, C.c_double)
        b.constant_intrinsics1 = _ptr(self.constant_intrinsics1, C.c_uint8)
        b.constant_intrinsics2 = _ptr(self.constant_intrinsics2, C.c_uint8)
        b.correspondence_ptr = _ptr(self.correspondence_ptr, C.c_int64)
        b.features1 = _ptr(self.features1, C.c_double)
        b.features2 = _ptr(self.features2, C.c_double)
        b.points = _ptr(self.points, C.c_double)
        return b


@dataclass
class TwoViewAngularBatch:
    """View pairs for the batched BundleAdjustTwoViewsAngular (``tmi_ba_two_view_angular_batch``)."""

    rotation2: np.ndarray              # [P, 3] angle-axis, in/out
    position2: np.ndarray              # [P, 3] unit norm, in/out
    correspondence_ptr: np.ndarray     # [P + 1] int64
    features1: np.ndarray              # [N, 2] normalised image coordinates
    features2: np.ndarray

    def __post_init__(self):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)  # noqa: E731
        self.rotation2 = f64(self.rotation2).reshape(-1, 3)
        self.position2 = f64(self.position2).reshape(-1, 3)
        self.correspondence_ptr = np.ascontiguousarray(self.correspondence_ptr, dtype=np.int64)
        self.features1 = f64(self.features1).reshape(-1, 2)
        self.features2 = f64(self.features2).reshape(-1, 2)

    @property
    def num_pairs(self) -> int:
        return self.rotation2.shape[0]

    def copy(self) -> "TwoViewAngularBatch":
        return TwoViewAngularBatch(self.rotation2.copy(), self.position2.copy(), self.correspondence_ptr.copy(),
                                   self.features1.copy(), self.features2.copy())

    def head(self, n: int) -> "TwoViewAngularBatch":
        """the first n pairs (copies)"""
        n = min(int(n), self.num_pairs)
        m = int(self.correspondence_ptr[n])
        return TwoViewAngularBatch(self.rotation2[:n].copy(), self.position2[:n].copy(),
                                   self.correspondence_ptr[:n + 1].copy(), self.features1[:m].copy(),
                                   self.features2[:m].copy())

    def as_c(self) -> CTwoViewAngularBatch:
        b = CTwoViewAngularBatch()
        b.num_pairs = self.num_pairs
        b.rotation2 = _ptr(self.rotation2, C.c_double)
        b.position2 = _ptr(self.position2, C.c_double)
        b.correspondence_ptr = _ptr(self.correspondence_ptr, C.c_int64)
        b.features1 = _ptr(self.features1, C.c_double)
        b.features2 = _ptr(self.features2, C.c_double)
        return b
