"""Loader + thin ctypes wrapper of the C-ABI library (include/theia_mi355_ba.h).

The library is built in-tree by ``__graft_entry__.build()`` into
``theiasfm_amd/lib/libtheia_mi355_ba.so``.  There is no CPU fallback: if the
library is missing, or no HIP device is visible when a solve is requested, the
calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtheia_mi355_ba.so")
_lib = None

# every symbol include/theia_mi355_ba.h declares
EXPORTS = (
    "tmi_ba_version", "tmi_ba_device_count", "tmi_ba_status_string", "tmi_ba_last_error",
    "tmi_ba_options_init",
    "tmi_ba_intrinsics_size", "tmi_ba_intrinsics_constant_mask", "tmi_ba_solve",
    "tmi_ba_solver_create", "tmi_ba_solver_set_allreduce", "tmi_ba_solver_solve",
    "tmi_ba_solver_reset", "tmi_ba_solver_set_parameters", "tmi_ba_solver_download", "tmi_ba_solver_stream",
    "tmi_ba_solver_destroy", "tmi_ba_solver_evaluate", "tmi_ba_structure_stats", "tmi_ba_structure_stats_for",
    "tmi_ba_rccl_unique_id", "tmi_ba_solver_init_rccl", "tmi_ba_solver_debug_allreduce",
    "tmi_ba_solver_filter_outlier_tracks", "tmi_ba_filter_outlier_tracks",
    "tmi_ba_solver_adjust_tracks", "tmi_ba_adjust_tracks",
    "tmi_ba_solver_select_good_tracks", "tmi_ba_select_good_tracks",
    "tmi_ba_adjust_two_views", "tmi_ba_adjust_two_views_angular", "tmi_ba_solver_structure_checksums",
    "tmi_ba_solver_operator_info",
)

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)


class LibraryMissing(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the HIP engine has no CPU fallback)")
    # Load order matters: torch bundles its own libamdhip64.so (SONAME
    # libamdhip64.so.7, the same as /opt/rocm's).  Imported first, the loader
    # resolves the engine's NEEDED libamdhip64.so.7 to torch's already loaded
    # runtime, so the engine and torch share ONE HIP runtime and streams / device
    # pointers are interchangeable (needed by the all-reduce hook).  Loaded the
    # other way round the process ends up with two runtimes and torch sees no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    P, O, S = C.POINTER(abi.CProblem), C.POINTER(abi.COptions), C.POINTER(abi.CSummary)
    L.tmi_ba_version.restype = C.c_int32
    L.tmi_ba_device_count.restype = C.c_int32
    L.tmi_ba_status_string.restype = C.c_char_p
    L.tmi_ba_status_string.argtypes = [C.c_int32]
    L.tmi_ba_last_error.restype = C.c_char_p
    L.tmi_ba_options_init.argtypes = [O]
    L.tmi_ba_options_init.restype = None
    L.tmi_ba_intrinsics_size.argtypes = [C.c_int32]
    L.tmi_ba_intrinsics_size.restype = C.c_int32
    L.tmi_ba_intrinsics_constant_mask.argtypes = [C.c_int32, C.c_int32, C.c_void_p]
    L.tmi_ba_intrinsics_constant_mask.restype = C.c_int32
    L.tmi_ba_solve.argtypes = [P, O, S]
    L.tmi_ba_solve.restype = C.c_int32
    L.tmi_ba_solver_create.argtypes = [P, O, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    L.tmi_ba_solver_create.restype = C.c_int32
    L.tmi_ba_solver_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
    L.tmi_ba_solver_set_allreduce.restype = C.c_int32
    L.tmi_ba_solver_solve.argtypes = [C.c_void_p, O, S]
    L.tmi_ba_solver_solve.restype = C.c_int32
    L.tmi_ba_solver_reset.argtypes = [C.c_void_p]
    L.tmi_ba_solver_reset.restype = C.c_int32
    L.tmi_ba_solver_set_parameters.argtypes = [C.c_void_p, C.c_void_p]
    L.tmi_ba_solver_set_parameters.restype = C.c_int32
    L.tmi_ba_solver_download.argtypes = [C.c_void_p, P]
    L.tmi_ba_solver_download.restype = C.c_int32
    L.tmi_ba_solver_stream.argtypes = [C.c_void_p]
    L.tmi_ba_solver_stream.restype = C.c_void_p
    L.tmi_ba_solver_destroy.argtypes = [C.c_void_p]
    L.tmi_ba_solver_destroy.restype = None
    L.tmi_ba_solver_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.POINTER(C.c_int32)]
    L.tmi_ba_solver_evaluate.restype = C.c_int32
    L.tmi_ba_rccl_unique_id.argtypes = [C.c_void_p]
    L.tmi_ba_rccl_unique_id.restype = C.c_int32
    L.tmi_ba_solver_init_rccl.argtypes = [C.c_void_p, C.c_void_p]
    L.tmi_ba_solver_init_rccl.restype = C.c_int32
    L.tmi_ba_solver_debug_allreduce.argtypes = [C.c_void_p, C.c_double, C.POINTER(C.c_double)]
    L.tmi_ba_solver_debug_allreduce.restype = C.c_int32
    L.tmi_ba_structure_stats.argtypes = [P, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    L.tmi_ba_structure_stats.restype = C.c_int32
    L.tmi_ba_structure_stats_for.argtypes = [P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
    L.tmi_ba_structure_stats_for.restype = C.c_int32
    FS, TS = C.POINTER(abi.CFilterSummary), C.POINTER(abi.CTrackBatchSummary)
    L.tmi_ba_solver_filter_outlier_tracks.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p,
                                                      C.c_void_p, FS]
    L.tmi_ba_solver_filter_outlier_tracks.restype = C.c_int32
    L.tmi_ba_filter_outlier_tracks.argtypes = [P, C.c_int32, C.c_double, C.c_double, C.c_void_p,
                                               C.c_void_p, FS]
    L.tmi_ba_filter_outlier_tracks.restype = C.c_int32
    L.tmi_ba_solver_adjust_tracks.argtypes = [C.c_void_p, O, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, TS]
    L.tmi_ba_solver_adjust_tracks.restype = C.c_int32
    L.tmi_ba_adjust_tracks.argtypes = [P, O, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, TS]
    L.tmi_ba_adjust_tracks.restype = C.c_int32
    L.tmi_ba_adjust_two_views.argtypes = [C.POINTER(abi.CTwoViewBatch), C.c_int32, C.c_int32, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, TS]
    L.tmi_ba_adjust_two_views.restype = C.c_int32
    L.tmi_ba_adjust_two_views_angular.argtypes = [C.POINTER(abi.CTwoViewAngularBatch), C.c_int32, C.c_int32,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.POINTER(abi.CTrackBatchSummary)]
    L.tmi_ba_adjust_two_views_angular.restype = C.c_int32
    L.tmi_ba_solver_structure_checksums.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.tmi_ba_solver_structure_checksums.restype = C.c_int32
    L.tmi_ba_solver_operator_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.tmi_ba_solver_operator_info.restype = C.c_int32
    SS = C.POINTER(abi.CSelectSummary)
    L.tmi_ba_solver_select_good_tracks.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, SS]
    L.tmi_ba_solver_select_good_tracks.restype = C.c_int32
    L.tmi_ba_select_good_tracks.argtypes = [P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, SS]
    L.tmi_ba_select_good_tracks.restype = C.c_int32
    _lib = L
    return L


STRUCTURE_STAT_NAMES = ("tracks", "observations", "reduced_blocks", "block_dim", "upper_blocks",
                        "bsr_blocks", "pairs", "block_checksum", "slices", "padded_observations",
                        "observation_checksum", "pair_checksum")


def structure_stats(problem: abi.Problem, rank: int = 0, world: int = 1, forms_S: bool = True) -> dict:
    """Host-only statistics of the static structure for one rank (no GPU).  forms_S=False: the dealing of a handle whose
    operator is matrix-free (the default on several ranks) -- slices balanced by observations, not by Schur pairs."""
    L = load()
    cp = problem.as_c()
    out = (C.c_int64 * 12)()
    st = L.tmi_ba_structure_stats_for(C.byref(cp), rank, world, 1 if forms_S else 0, out)
    if st != 0:
        raise EngineError(st, "tmi_ba_structure_stats")
    return dict(zip(STRUCTURE_STAT_NAMES, list(out)))


def rccl_unique_id() -> bytes:
    """ncclGetUniqueId through the engine's run-time RCCL binding (call on rank 0)."""
    L = load()
    buf = (C.c_uint8 * 128)()
    st = L.tmi_ba_rccl_unique_id(buf)
    if st != 0:
        raise EngineError(st, "tmi_ba_rccl_unique_id")
    return bytes(buf)


class EngineError(RuntimeError):
    def __init__(self, status, where, message=""):
        self.status = status
        name = abi.STATUS_NAMES.get(status, str(status))
        if not message and _lib is not None:
            message = (_lib.tmi_ba_last_error() or b"").decode("utf-8", "replace")
        super().__init__(f"{where}: status {status} ({name}) {message}")


def solve(problem: abi.Problem, options: abi.COptions):
    """One-shot tmi_ba_solve: upload, LM on the GPU, download into `problem`."""
    L = load()
    cp = problem.as_c()
    s = abi.CSummary()
    st = L.tmi_ba_solve(C.byref(cp), C.byref(options), C.byref(s))
    return st, s


def filter_outlier_tracks(problem: abi.Problem, max_inlier_reprojection_error: float,
                          min_triangulation_angle_degrees: float, device: int = -1):
    """One-shot SetOutlierTracksToUnestimated: (flag [Np] uint8, mean squared reprojection
    error [Np], CFilterSummary)."""
    L = load()
    cp = problem.as_c()
    n = problem.num_points
    flag = np.zeros(n, dtype=np.uint8)
    mean = np.zeros(n)
    fs = abi.CFilterSummary()
    st = L.tmi_ba_filter_outlier_tracks(C.byref(cp), device, float(max_inlier_reprojection_error),
                                        float(min_triangulation_angle_degrees), flag.ctypes.data,
                                        mean.ctypes.data, C.byref(fs))
    if st != 0:
        raise EngineError(st, "tmi_ba_filter_outlier_tracks")
    return flag, mean, fs


def select_good_tracks(problem: abi.Problem, long_track_length_threshold: int,
                       image_grid_cell_size_pixels: int, min_num_optimized_tracks_per_view: int,
                       view_mask=None, device: int = -1):
    """One-shot SelectGoodTracksForBundleAdjustment: (selected [Np] uint8, truncated length
    [Np] int32, mean squared error [Np], CSelectSummary)."""
    L = load()
    cp = problem.as_c()
    n = problem.num_points
    sel = np.zeros(n, dtype=np.uint8)
    ln = np.zeros(n, dtype=np.int32)
    err = np.zeros(n)
    ss = abi.CSelectSummary()
    vm = None if view_mask is None else np.ascontiguousarray(view_mask, dtype=np.uint8)
    st = L.tmi_ba_select_good_tracks(C.byref(cp), device, long_track_length_threshold,
                                     image_grid_cell_size_pixels, min_num_optimized_tracks_per_view,
                                     None if vm is None else vm.ctypes.data, sel.ctypes.data, ln.ctypes.data, err.ctypes.data, C.byref(ss))
    if st != 0:
        raise EngineError(st, "tmi_ba_select_good_tracks")
    return sel, ln, err, ss


def adjust_tracks(problem: abi.Problem, options: abi.COptions):
    """One-shot batched BundleAdjustTrack; problem.points is updated in place.  Returns
    (termination [Np] int8, iterations [Np] int32, initial cost [Np], final cost [Np],
    CTrackBatchSummary)."""
    L = load()
    cp = problem.as_c()
    n = problem.num_points
    term = np.full(n, -1, dtype=np.int8)
    iters = np.zeros(n, dtype=np.int32)
    c0 = np.zeros(n)
    c1 = np.zeros(n)
    ts = abi.CTrackBatchSummary()
    st = L.tmi_ba_adjust_tracks(C.byref(cp), C.byref(options), term.ctypes.data, iters.ctypes.data,
                                c0.ctypes.data, c1.ctypes.data, C.byref(ts))
    if st != 0:
        raise EngineError(st, "tmi_ba_adjust_tracks")
    return term, iters, c0, c1, ts


def adjust_two_views(batch: abi.TwoViewBatch, point_dof: int = 4, max_num_iterations: int = 200, device: int = -1):
    """Batched BundleAdjustTwoViews; batch.extrinsics2 / intrinsics / points are updated in place
    for the usable pairs.  Returns (termination [P] int8, iterations [P] int32, initial cost [P],
    final cost [P], CTrackBatchSummary)."""
    L = load()
    cb = batch.as_c()
    n = batch.num_pairs
    term = np.full(n, -1, dtype=np.int8)
    iters = np.zeros(n, dtype=np.int32)
    c0 = np.zeros(n)
    c1 = np.zeros(n)
    ts = abi.CTrackBatchSummary()
    st = L.tmi_ba_adjust_two_views(C.byref(cb), int(point_dof), int(max_num_iterations), int(device),
                                   term.ctypes.data, iters.ctypes.data, c0.ctypes.data, c1.ctypes.data,
                                   C.byref(ts))
    if st != 0:
        raise EngineError(st, "tmi_ba_adjust_two_views")
    return term, iters, c0, c1, ts


def adjust_two_views_angular(batch: abi.TwoViewAngularBatch, max_num_iterations: int = 200, device: int = -1):
    """Batched BundleAdjustTwoViewsAngular; batch.rotation2 / position2 are updated in place for the usable
    pairs.  Returns (termination [P] int8, iterations [P] int32, initial cost [P], final cost [P],
    CTrackBatchSummary)."""
    L = load()
    cb = batch.as_c()
    n = batch.num_pairs
    term = np.full(n, -1, dtype=np.int8)
    iters = np.zeros(n, dtype=np.int32)
    c0 = np.zeros(n)
    c1 = np.zeros(n)
    ts = abi.CTrackBatchSummary()
    st = L.tmi_ba_adjust_two_views_angular(C.byref(cb), int(max_num_iterations), int(device), term.ctypes.data,
                                           iters.ctypes.data, c0.ctypes.data, c1.ctypes.data, C.byref(ts))
    if st != 0:
        raise EngineError(st, "tmi_ba_adjust_two_views_angular")
    return term, iters, c0, c1, ts


class Solver:
    """Resident form: the problem stays in HBM across solve() calls."""

    def __init__(self, problem: abi.Problem, options: abi.COptions, rank: int = 0, world: int = 1):
        self._L = load()
        self.problem = problem
        self._cp = problem.as_c()
        self._h = C.c_void_p()
        self._cb = None
        st = self._L.tmi_ba_solver_create(C.byref(self._cp), C.byref(options), rank, world,
                                          C.byref(self._h))
        if st != 0:
            self._h = C.c_void_p()
            raise EngineError(st, "tmi_ba_solver_create")

    def set_allreduce(self, fn):
        """fn(device_ptr:int, count:int, hip_stream:int) -> 0 on success."""
        def tramp(buf, count, stream, user):
            try:
                return int(fn(buf, count, stream))
            except Exception as exc:  # never let an exception cross the C boundary
                print(f"[theiasfm_amd] all-reduce hook raised: {exc!r}", flush=True)
                return 1
        self._cb = ALLREDUCE_FN(tramp)
        st = self._L.tmi_ba_solver_set_allreduce(self._h, self._cb, None)
        if st != 0:
            raise EngineError(st, "tmi_ba_solver_set_allreduce")

    def init_rccl(self, unique_id: bytes):
        """Native RCCL transport: every rank passes the 128-byte id rank 0 obtained from
        rccl_unique_id() (ship it with torch.distributed / MPI / a file)."""
        if unique_id is None:  # drop the communicator: the all-reduce hook applies again
            st = self._L.tmi_ba_solver_init_rccl(self._h, None)
            if st != 0:
                raise EngineError(st, "tmi_ba_solver_init_rccl")
            return
        if len(unique_id) != 128:
            raise ValueError("ncclUniqueId is 128 bytes")
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        st = self._L.tmi_ba_solver_init_rccl(self._h, buf)
        if st != 0:
            raise EngineError(st, "tmi_ba_solver_init_rccl")

    def debug_allreduce(self, value: float) -> float:
        out = C.c_double(0.0)
        st = self._L.tmi_ba_solver_debug_allreduce(self._h, value, C.byref(out))
        if st != 0:
            raise EngineError(st, "tmi_ba_solver_debug_allreduce")
        return out.value

    def solve(self, options: abi.COptions):
        s = abi.CSummary()
        st = self._L.tmi_ba_solver_solve(self._h, C.byref(options), C.byref(s))
        return st, s

    def reset(self):
        st = self._L.tmi_ba_solver_reset(self._h)
        if st != 0:
            raise EngineError(st, "tmi_ba_solver_reset")

    def set_parameters(self, prob: abi.Problem):
        """new extrinsics / intrinsics / points for the resident structure (what reset() restores from then on)"""
        self._keep = prob  # the C view points into its arrays
        cp = prob.as_c()
        st = self._L.tmi_ba_solver_set_parameters(self._h, C.byref(cp))
        if st != 0:
            raise EngineError(st, "tmi_ba_solver_set_parameters")

    def download(self):
        st = self._L.tmi_ba_solver_download(self._h, C.byref(self._cp))
        if st != 0:
            raise EngineError(st, "tmi_ba_solver_download")
        return self.problem

    def filter_outlier_tracks(self, max_inlier_reprojection_error: float,
                              min_triangulation_angle_degrees: float):
        """SetOutlierTracksToUnestimated on the resident parameters (this rank's tracks)."""
        n = self.problem.num_points
        flag = np.full(n, 255, dtype=np.uint8)
        mean = np.full(n, np.nan)
        fs = abi.CFilterSummary()
        st = self._L.tmi_ba_solver_filter_outlier_tracks(
            self._h, float(max_inlier_reprojection_error), float(min_triangulation_angle_degrees),
            flag.ctypes.data, mean.ctypes.data, C.byref(fs))
        if st != 0:
            raise EngineError(st, "tmi_ba_solver_filter_outlier_tracks")
        return flag, mean, fs

    def select_good_tracks(self, long_track_length_threshold: int, image_grid_cell_size_pixels: int,
                           min_num_optimized_tracks_per_view: int, view_mask=None):
        """SelectGoodTracksForBundleAdjustment on the resident parameters (unsharded handle)."""
        n = self.problem.num_points
        sel = np.zeros(n, dtype=np.uint8)
        ln = np.zeros(n, dtype=np.int32)
        err = np.zeros(n)
        ss = abi.CSelectSummary()
        vm = None if view_mask is None else np.ascontiguousarray(view_mask, dtype=np.uint8)
        st = self._L.tmi_ba_solver_select_good_tracks(
            self._h, long_track_length_threshold, image_grid_cell_size_pixels,
            min_num_optimized_tracks_per_view, None if vm is None else vm.ctypes.data, sel.ctypes.data, ln.ctypes.data, err.ctypes.data,
            C.byref(ss))
        if st != 0:
            raise EngineError(st, "tmi_ba_solver_select_good_tracks")
        return sel, ln, err, ss

    def adjust_tracks(self, options: abi.COptions):
        """Batched BundleAdjustTrack on the resident parameters (this rank's tracks)."""
        n = self.problem.num_points
        term = np.full(n, -1, dtype=np.int8)
        iters = np.zeros(n, dtype=np.int32)
        c0 = np.zeros(n)
        c1 = np.zeros(n)
        ts = abi.CTrackBatchSummary()
        st = self._L.tmi_ba_solver_adjust_tracks(self._h, C.byref(options), term.ctypes.data,
                                                 iters.ctypes.data, c0.ctypes.data, c1.ctypes.data,
                                                 C.byref(ts))
        if st != 0:
            raise EngineError(st, "tmi_ba_solver_adjust_tracks")
        return term, iters, c0, c1, ts

    def structure_checksums(self):
        """Test hook: [24] uint64 checksums of the static structure arrays in HBM ([0] = built on the device)."""
        out = (C.c_uint64 * 24)()
        st = self._L.tmi_ba_solver_structure_checksums(self._h, out)
        if st != 0:
            raise EngineError(st, "tmi_ba_solver_structure_checksums")
        return list(out)

    def operator_info(self) -> dict:
        """Which kernels this handle runs (tmi_ba_solver_operator_info)."""
        out = (C.c_int32 * 8)()
        st = self._L.tmi_ba_solver_operator_info(self._h, out)
        if st != 0:
            raise EngineError(st, "tmi_ba_solver_operator_info")
        return dict(one_sweep_product=bool(out[0]), position_columns_formed=bool(out[1]), direct_camera_side=bool(out[2]),
                    adaptive=bool(out[3]), implicit=bool(out[4]), break_even=int(out[5]), cluster_handle=bool(out[6]),
                    compact_planes=bool(out[7]))

    @property
    def stream(self) -> int:
        return int(self._L.tmi_ba_solver_stream(self._h) or 0)

    def evaluate(self, point_dof: int):
        """Device residuals [N,2], reduced camera Jacobians [N,2,D], shared-intrinsics
        Jacobians [N,2,D], point Jacobians [N,2,point_dof], valid [N] in the caller's
        observation order, and D."""
        n = self.problem.num_observations
        bd = C.c_int32(0)
        # D is not known before the call: allocate for the largest block (16)
        r = np.zeros((n, 2))
        A = np.zeros(n * 2 * 16)
        A1 = np.zeros(n * 2 * 16)
        Jp = np.zeros((n, 2, point_dof))
        valid = np.zeros(n, dtype=np.uint8)
        st = self._L.tmi_ba_solver_evaluate(self._h, r.ctypes.data, A.ctypes.data, A1.ctypes.data,
                                            Jp.ctypes.data, valid.ctypes.data, C.byref(bd))
        if st != 0:
            raise EngineError(st, "tmi_ba_solver_evaluate")
        D = bd.value
        return r, A[: n * 2 * D].reshape(n, 2, D), A1[: n * 2 * D].reshape(n, 2, D), Jp, valid, D

    def close(self):
        if self._h:
            self._L.tmi_ba_solver_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
