"""ORACLE -- TEST INFRASTRUCTURE ONLY (see ba_oracle.h).

ctypes binding of oracle/liboracle_ba.so for tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg.  The product package never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from theiasfm_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_ba.so")
_lib = None


def _stale() -> bool:
    if not os.path.exists(_LIB_PATH):
        return True
    t = os.path.getmtime(_LIB_PATH)
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h")) or f == "Makefile"]
    srcs.append(os.path.join(_HERE, "..", "include", "theia_mi355_ba.h"))
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in srcs)


def build(force: bool = False) -> str:
    if force or _stale():
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_ba_solve.argtypes = [C.POINTER(abi.CProblem), C.POINTER(abi.COptions),
                                      C.POINTER(abi.CSummary)]
        L.oracle_ba_solve.restype = C.c_int32
        L.oracle_inner_sweep.argtypes = [C.POINTER(abi.CProblem), C.POINTER(abi.COptions)]
        L.oracle_inner_sweep.restype = C.c_int32
        L.oracle_set_inner_order.argtypes = [C.c_int32]
        L.oracle_set_inner_order.restype = None
        L.oracle_ba_evaluate.argtypes = [C.POINTER(abi.CProblem), C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_ba_evaluate.restype = C.c_int32
        L.oracle_ba_cost.argtypes = [C.POINTER(abi.CProblem), C.POINTER(abi.COptions),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.oracle_ba_cost.restype = C.c_int64
        L.oracle_project_point.argtypes = [C.c_int32] + [C.c_void_p] * 4
        L.oracle_project_point.restype = C.c_double
        L.oracle_camera_to_pixel.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_pixel_to_camera.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_camera_to_pixel_batch.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.oracle_pixel_to_camera_batch.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.oracle_loss.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_void_p]
        L.oracle_intrinsics_constant_mask.argtypes = [C.c_int32, C.c_int32, C.c_void_p]
        L.oracle_intrinsics_constant_mask.restype = C.c_int32
        L.oracle_num_threads.restype = C.c_int32
        L.oracle_set_num_threads.argtypes = [C.c_int32]
        L.oracle_set_num_threads.restype = None
        L.oracle_select_good_tracks.argtypes = [C.POINTER(abi.CProblem), C.c_int32, C.c_int32, C.c_int32,
                                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_select_good_tracks.restype = C.c_int32
        L.oracle_adjust_tracks.argtypes = [C.POINTER(abi.CProblem), C.POINTER(abi.COptions),
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_adjust_tracks.restype = C.c_int32
        L.oracle_adjust_two_views.argtypes = [C.POINTER(abi.CTwoViewBatch), C.c_int32, C.c_int32,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_adjust_two_views.restype = C.c_int32
        L.oracle_adjust_two_views_angular.argtypes = [C.POINTER(abi.CTwoViewAngularBatch), C.c_int32,
                                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_adjust_two_views_angular.restype = C.c_int32
        L.oracle_angular_epipolar_error.argtypes = [C.c_void_p] * 5
        L.oracle_angular_epipolar_error.restype = C.c_int32
        L.oracle_sufficient_triangulation_angle.argtypes = [C.c_void_p, C.c_int64, C.c_double]
        L.oracle_sufficient_triangulation_angle.restype = C.c_int32
        L.oracle_filter_outlier_tracks.argtypes = [C.POINTER(abi.CProblem), C.c_double, C.c_double,
                                                   C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_filter_outlier_tracks.restype = C.c_int32
        _lib = L
    return _lib


def solve(problem: abi.Problem, options: abi.COptions):
    """Runs the CPU LM on a copy-free view of `problem` (updated in place)."""
    cp = problem.as_c()
    s = abi.CSummary()
    st = lib().oracle_ba_solve(C.byref(cp), C.byref(options), C.byref(s))
    return st, s


def last_visibility_clusters(num_cameras: int) -> np.ndarray:
    """cluster of every camera in the last clustered solve (CLUSTER_JACOBI without shared intrinsics blocks)"""
    out = np.full(num_cameras, -2, dtype=np.int32)
    L = lib()
    L.oracle_last_visibility_clusters.argtypes = [C.c_void_p, C.c_int32]
    L.oracle_last_visibility_clusters.restype = C.c_int32
    n = L.oracle_last_visibility_clusters(out.ctypes.data, num_cameras)
    assert n == num_cameras, n
    return out


def last_tridiagonal_segments(num_cameras: int):
    """(segment, position of the view's cluster in the segment) of every camera in the last CLUSTER_TRIDIAGONAL solve"""
    seg = np.full(num_cameras, -2, dtype=np.int32)
    pos = np.full(num_cameras, -2, dtype=np.int32)
    L = lib()
    L.oracle_last_tridiagonal_segments.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.oracle_last_tridiagonal_segments.restype = C.c_int32
    n = L.oracle_last_tridiagonal_segments(seg.ctypes.data, pos.ctypes.data, num_cameras)
    assert n == num_cameras, n
    return seg, pos


def inner_sweep(problem: abi.Problem, options: abi.COptions):
    """One coordinate-descent sweep (Ceres inner iterations) in place."""
    cp = problem.as_c()
    st = lib().oracle_inner_sweep(C.byref(cp), C.byref(options))
    assert st == 0, st


def set_inner_order(order: int):
    """0 = reference order (extrinsics, intrinsics, points); 1 = intrinsics first (test hook)."""
    lib().oracle_set_inner_order(int(order))


def evaluate(problem: abi.Problem):
    """residuals [N,2], dual-number Jacobians [N,2,20], valid [N]."""
    n = problem.num_observations
    r = np.zeros((n, 2))
    J = np.zeros((n, 2, 20))
    v = np.zeros(n, dtype=np.uint8)
    cp = problem.as_c()
    lib().oracle_ba_evaluate(C.byref(cp), r.ctypes.data, J.ctypes.data, v.ctypes.data)
    return r, J, v


def cost(problem: abi.Problem, options: abi.COptions | None = None):
    c, rm = C.c_double(), C.c_double()
    cp = problem.as_c()
    o = options if options is not None else abi.default_options()
    bad = lib().oracle_ba_cost(C.byref(cp), C.byref(o), C.byref(c), C.byref(rm))
    return c.value, rm.value, bad


def camera_to_pixel(model, K, pt):
    K = np.ascontiguousarray(K, dtype=np.float64)
    pt = np.ascontiguousarray(pt, dtype=np.float64)
    out = np.zeros(2)
    lib().oracle_camera_to_pixel(model, K.ctypes.data, pt.ctypes.data, out.ctypes.data)
    return out


def pixel_to_camera(model, K, px):
    K = np.ascontiguousarray(K, dtype=np.float64)
    px = np.ascontiguousarray(px, dtype=np.float64)
    out = np.zeros(3)
    lib().oracle_pixel_to_camera(model, K.ctypes.data, px.ctypes.data, out.ctypes.data)
    return out


def camera_to_pixel_batch(model, K, pts):
    K = np.ascontiguousarray(K, dtype=np.float64)
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((pts.shape[0], 2))
    lib().oracle_camera_to_pixel_batch(model, K.ctypes.data, pts.ctypes.data, pts.shape[0],
                                       out.ctypes.data)
    return out


def pixel_to_camera_batch(model, K, px):
    K = np.ascontiguousarray(K, dtype=np.float64)
    px = np.ascontiguousarray(px, dtype=np.float64).reshape(-1, 2)
    out = np.zeros((px.shape[0], 3))
    lib().oracle_pixel_to_camera_batch(model, K.ctypes.data, px.ctypes.data, px.shape[0],
                                       out.ctypes.data)
    return out


def project_point(model, ext, K, X):
    ext, K, X = (np.ascontiguousarray(a, dtype=np.float64) for a in (ext, K, X))
    out = np.zeros(2)
    d = lib().oracle_project_point(model, ext.ctypes.data, K.ctypes.data, X.ctypes.data,
                                   out.ctypes.data)
    return out, d


def loss(kind, width, s):
    out = np.zeros(3)
    lib().oracle_loss(kind, width, s, out.ctypes.data)
    return out


def intrinsics_constant_mask(model, bits):
    out = np.zeros(abi.INTRINSICS_SIZE[model], dtype=np.uint8)
    lib().oracle_intrinsics_constant_mask(model, bits, out.ctypes.data)
    return out


def sufficient_triangulation_angle(rays, min_angle_degrees) -> bool:
    rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 3)
    return bool(lib().oracle_sufficient_triangulation_angle(rays.ctypes.data, rays.shape[0],
                                                            float(min_angle_degrees)))


def filter_outlier_tracks(problem: abi.Problem, max_inlier_reprojection_error,
                          min_triangulation_angle_degrees):
    """flag [Np] (0 kept / 1 bad reprojection / 2 insufficient angle), mean squared
    reprojection error [Np], counts (estimated, bad reprojections, insufficient angles)."""
    n = problem.num_points
    flag = np.zeros(n, dtype=np.uint8)
    mean = np.zeros(n)
    counts = np.zeros(3, dtype=np.int64)
    cp = problem.as_c()
    lib().oracle_filter_outlier_tracks(C.byref(cp), float(max_inlier_reprojection_error),
                                       float(min_triangulation_angle_degrees), flag.ctypes.data,
                                       mean.ctypes.data, counts.ctypes.data)
    return flag, mean, counts


def adjust_tracks(problem: abi.Problem, options: abi.COptions):
    """Per-track BundleAdjustTrack on the CPU; problem.points updated in place."""
    n = problem.num_points
    term = np.full(n, -1, dtype=np.int8)
    iters = np.zeros(n, dtype=np.int32)
    c0 = np.zeros(n)
    c1 = np.zeros(n)
    cp = problem.as_c()
    st = lib().oracle_adjust_tracks(C.byref(cp), C.byref(options), term.ctypes.data,
                                    iters.ctypes.data, c0.ctypes.data, c1.ctypes.data)
    assert st == 0, st
    return term, iters, c0, c1


def adjust_two_views(batch: abi.TwoViewBatch, point_dof: int = 4, max_num_iterations: int = 200):
    """BundleAdjustTwoViews pair by pair on the CPU; the batch is updated in place."""
    n = batch.num_pairs
    term = np.full(n, -1, dtype=np.int8)
    iters = np.zeros(n, dtype=np.int32)
    c0 = np.zeros(n)
    c1 = np.zeros(n)
    cb = batch.as_c()
    st = lib().oracle_adjust_two_views(C.byref(cb), int(point_dof), int(max_num_iterations), term.ctypes.data,
                                       iters.ctypes.data, c0.ctypes.data, c1.ctypes.data)
    assert st == 0, st
    return term, iters, c0, c1


def adjust_two_views_angular(batch: abi.TwoViewAngularBatch, max_num_iterations: int = 200):
    """BundleAdjustTwoViewsAngular pair by pair on the CPU; the batch is updated in place."""
    n = batch.num_pairs
    term = np.full(n, -1, dtype=np.int8)
    iters = np.zeros(n, dtype=np.int32)
    c0 = np.zeros(n)
    c1 = np.zeros(n)
    cb = batch.as_c()
    st = lib().oracle_adjust_two_views_angular(C.byref(cb), int(max_num_iterations), term.ctypes.data,
                                               iters.ctypes.data, c0.ctypes.data, c1.ctypes.data)
    assert st == 0, st
    return term, iters, c0, c1


def angular_epipolar_error(rotation, position, f1, f2):
    """AngularEpipolarError of one correspondence; None where the functor returns false."""
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (rotation, position, f1, f2)]
    out = np.zeros(1)
    ok = lib().oracle_angular_epipolar_error(*[v.ctypes.data for v in a], out.ctypes.data)
    return float(out[0]) if ok else None


def select_good_tracks(problem: abi.Problem, long_track_length_threshold: int,
                       image_grid_cell_size_pixels: int, min_num_optimized_tracks_per_view: int,
                       view_mask=None):
    """selected [Np] uint8, truncated length [Np] int32, mean squared error [Np]."""
    n = problem.num_points
    sel = np.zeros(n, dtype=np.uint8)
    ln = np.zeros(n, dtype=np.int32)
    err = np.zeros(n)
    cp = problem.as_c()
    vm = None if view_mask is None else np.ascontiguousarray(view_mask, dtype=np.uint8)
    st = lib().oracle_select_good_tracks(C.byref(cp), long_track_length_threshold,
                                         image_grid_cell_size_pixels,
                                         min_num_optimized_tracks_per_view,
                                         None if vm is None else vm.ctypes.data, sel.ctypes.data,
                                         ln.ctypes.data, err.ctypes.data)
    assert st == 0, st
    return sel, ln, err


def num_threads() -> int:
    return lib().oracle_num_threads()


def set_num_threads(n: int) -> None:
    lib().oracle_set_num_threads(int(n))
