"""Seeded synthetic bundle-adjustment problems at the BASELINE.json sizes.

No BAL / 1DSfM file exists in the container, so the benchmark and
the parity tests run on synthetic reconstructions with the named camera /
track / observation counts.  The scene is generated directly in the
reference's conventions:

  * camera = [position C(3), angle-axis w(3)] with q = R(w) (X - C)
    (reference: src/theia/sfm/camera/reprojection_error.h:62-83, camera.h:195-200),
    the camera looks along +z of its own frame;
  * PINHOLE intrinsics [f, ar, skew, px, py, k1, k2] (pinhole_camera_model.h:86-94),
    BAL-like: own intrinsics group per view, principal point (0, 0)
    (Bundler/BAL -> Theia convention);
  * homogeneous points with w = 1 (track.h:66-67).

Config 1 follows the recipe of the reference's own synthetic test scenes
(sfm/global_pose_estimation/nonlinear_position_estimator_test.cc:66-74,166-200):
positions 10*U(-1,1)^3, angle-axis 0.2*U(-1,1)^3, f = 800, pp = (500, 500),
points U(-1,1)^3 + (0,0,20), every view observes every track.

The larger configs are a "landmark" scene: cameras on a jittered ring looking
at a ball of points; each track is seen by k cameras drawn from a window of
the ring (k - 2 ~ Geometric so that mean k = N_obs / N_pts, window width
heavy-tailed), which gives a banded-plus-tail co-visibility structure whose
reduced camera matrix fill is controlled by ``spread``.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation

from . import abi
from .abi import Problem

# (cameras, points, observations) of BASELINE.json configs[0..3]
CONFIG_SIZES = {
    "tiny": (3, 100, 300),
    "ladybug49": (49, 7776, 31843),
    "alamo": (570, 140000, 900000),
    "venice1778": (1778, 993923, 5001946),
    # the same sizes with 0.2 % of the tracks seen by 24..400 views (not a BASELINE config: a stress
    # case for the long-track paths, cf. SURVEY 8d "max k a few hundred")
    "venice1778_heavy": (1778, 993923, 5001946),
    # the same sizes with SEQUENCE structure (scene "street"): S is a band of ~10 % fill, PCG needs tens of
    # iterations per LM iteration -- what real collections look like, where the ring scene is a turntable
    "venice1778_street": (1778, 993923, 5001946),
}


# ---- numpy projection (generator side only) ----------------------------------
def _distort(model: int, K: np.ndarray, q: np.ndarray) -> np.ndarray:
    """Vectorised CameraToPixelCoordinates for observations that all use `model`.
    K: [n, size], q: [n, 3] camera-frame points.  Used to synthesise pixels."""
    if model == abi.FISHEYE:
        x, y, z = q[:, 0], q[:, 1], q[:, 2]
        r = np.sqrt(x * x + y * y)
        th = np.arctan2(r, np.abs(z))
        t2 = th * th
        thd = th * (1 + K[:, 5] * t2 + K[:, 6] * t2**2 + K[:, 7] * t2**3 + K[:, 8] * t2**4)
        s = np.where(r * r < 1e-8, 1.0, thd / np.maximum(r, 1e-300)) * np.where(z < 0, -1.0, 1.0)
        s = np.where(r * r < 1e-8, 1.0, s)
        dx, dy = s * x, s * y
        return np.stack([K[:, 0] * dx + K[:, 2] * dy + K[:, 3], K[:, 0] * K[:, 1] * dy + K[:, 4]], 1)
    nx, ny = q[:, 0] / q[:, 2], q[:, 1] / q[:, 2]
    r2 = nx * nx + ny * ny
    if model == abi.PINHOLE:
        d = 1 + r2 * (K[:, 5] + K[:, 6] * r2)
        dx, dy = nx * d, ny * d
        return np.stack([K[:, 0] * dx + K[:, 2] * dy + K[:, 3], K[:, 0] * K[:, 1] * dy + K[:, 4]], 1)
    if model == abi.PINHOLE_RADIAL_TANGENTIAL:
        rd = 1 + K[:, 5] * r2 + K[:, 6] * r2**2 + K[:, 7] * r2**3
        tx = K[:, 9] * (r2 + 2 * nx * nx) + 2 * K[:, 8] * nx * ny
        ty = K[:, 8] * (r2 + 2 * ny * ny) + 2 * K[:, 9] * nx * ny
        dx, dy = nx * rd + tx, ny * rd + ty
        return np.stack([K[:, 0] * dx + K[:, 2] * dy + K[:, 3], K[:, 0] * K[:, 1] * dy + K[:, 4]], 1)
    if model == abi.FOV:
        w = K[:, 4]
        ru = np.sqrt(r2)
        with np.errstate(divide="ignore", invalid="ignore"):
            rd = np.where(
                w < 1e-3, w * w * r2 / 3 - w * w / 12 + 1,
                np.where(r2 < 1e-3,
                         -2 * np.tan(w / 2) * (4 * r2 * np.tan(w / 2) ** 2 - 3) / (3 * w),
                         np.arctan(2 * ru * np.tan(w / 2)) / (ru * w)))
        return np.stack([K[:, 0] * rd * nx + K[:, 2], K[:, 0] * K[:, 1] * rd * ny + K[:, 3]], 1)
    # division undistortion
    ux, uy = K[:, 0] * nx, K[:, 0] * K[:, 1] * ny
    ru2 = ux * ux + uy * uy
    k = K[:, 4]
    denom = 2 * k * ru2
    inner = 1 - 4 * k * ru2
    with np.errstate(divide="ignore", invalid="ignore"):
        sc = np.where((np.abs(denom) < np.finfo(float).eps) | (inner < 0), 1.0,
                      (1 - np.sqrt(np.maximum(inner, 0))) / denom)
    return np.stack([ux * sc + K[:, 2], uy * sc + K[:, 3]], 1)


def project(problem: Problem, obs_index=None) -> np.ndarray:
    """Pixels of every observation at the problem's current parameters."""
    cam = problem.obs_camera if obs_index is None else problem.obs_camera[obs_index]
    pt = problem.obs_point if obs_index is None else problem.obs_point[obs_index]
    Rm = Rotation.from_rotvec(problem.extrinsics[:, 3:6]).as_matrix()
    X = problem.points[pt]
    a = X[:, :3] - X[:, 3:4] * problem.extrinsics[cam, :3]
    q = np.einsum("nij,nj->ni", Rm[cam], a)
    out = np.empty((cam.shape[0], 2))
    grp = problem.camera_group[cam]
    models = problem.group_model[grp]
    for m in np.unique(models):
        sel = np.nonzero(models == m)[0]
        n = abi.INTRINSICS_SIZE[m]
        K = problem.intrinsics[problem.group_offset[grp[sel]][:, None] + np.arange(n)[None, :]]
        out[sel] = _distort(int(m), K, q[sel])
    return out


# ---- visibility -------------------------------------------------------------
def _track_lengths(rng, n_cameras, n_points, n_obs, heavy_tail=0.0):
    mean_k = n_obs / n_points
    if mean_k >= n_cameras:
        return np.full(n_points, n_cameras, dtype=np.int64)
    p = 1.0 / max(mean_k - 1.0, 1.0 + 1e-9)
    k = 1 + rng.geometric(min(p, 1.0), n_points).astype(np.int64)
    k = np.clip(k, 2, n_cameras)
    if heavy_tail > 0.0:
        # a fraction `heavy_tail` of the tracks is seen by tens to hundreds of views (landmark
        # points of internet photo collections; SURVEY 8d config 4: "max k a few hundred")
        long = np.flatnonzero(rng.random(n_points) < heavy_tail)
        k[long] = np.clip(np.rint(24.0 * (1.0 + rng.pareto(1.3, long.size))), 24, min(n_cameras, 400)).astype(np.int64)
    # steer the total to exactly n_obs
    for _ in range(64):
        diff = int(n_obs - k.sum())
        if diff == 0:
            break
        if diff > 0:
            cand = np.nonzero(k < n_cameras)[0]
            pick = rng.choice(cand, size=min(diff, cand.size), replace=False)
            k[pick] += 1
        else:
            cand = np.nonzero(k > 2)[0]
            pick = rng.choice(cand, size=min(-diff, cand.size), replace=False)
            k[pick] -= 1
    if k.sum() != n_obs:
        raise ValueError("cannot realise the requested observation count")
    return k


def _visibility(rng, n_cameras, n_points, n_obs, spread, heavy_tail=0.0):
    k = _track_lengths(rng, n_cameras, n_points, n_obs, heavy_tail)
    # window width w = m*k cameras, m heavy tailed, w <= n_cameras
    m_max = np.maximum(n_cameras // k, 1)
    m = np.rint(spread * n_cameras / k * rng.lognormal(0.0, 0.9, n_points)).astype(np.int64)
    m = np.clip(m, 1, m_max)
    start = rng.integers(0, n_cameras, n_points)
    pt = np.repeat(np.arange(n_points, dtype=np.int64), k)
    first = np.cumsum(k) - k
    j = np.arange(n_obs, dtype=np.int64) - np.repeat(first, k)
    mm = np.repeat(m, k)
    off = j * mm + (rng.random(n_obs) * mm).astype(np.int64)
    cam = (np.repeat(start, k) + off) % n_cameras
    return cam.astype(np.int32), pt.astype(np.int32), k


def _visibility_street(rng, n_cameras, n_points, n_obs, spread, heavy_tail=0.0, max_window=0.14):
    """Sequence structure: the views of a track are k of the w = m k CONSECUTIVE views around the track's own place on
    the path (m >= 1, log-normal around `spread` n / k, w capped at max_window n), so two views share tracks only
    when they are neighbours on the path -- the reduced camera matrix is a band (plus the wrap-around of the closed
    path), its fill about 2 E[w] / n.  Track lengths: the geometric body of _track_lengths with a short Pareto tail
    (landmarks seen from up to ~7 % of the path), the shape of the BAL track-length histograms (mode 2, mean ~5,
    maximum a few hundred).  Returns (camera, point, k, centre view of every track, window of every track)."""
    k = _track_lengths(rng, n_cameras, n_points, n_obs, 0.0)
    if heavy_tail > 0.0:
        long = np.flatnonzero(rng.random(n_points) < heavy_tail)
        k[long] = np.clip(np.rint(16.0 * (1.0 + rng.pareto(1.6, long.size))), 16, max(int(0.07 * n_cameras), 16)).astype(np.int64)
        for _ in range(64):  # steer the total back to exactly n_obs on the short tracks
            diff = int(n_obs - k.sum())
            if diff == 0:
                break
            cand = np.flatnonzero((k < 12) if diff > 0 else ((k > 2) & (k < 12)))
            pick = rng.choice(cand, size=min(abs(diff), cand.size), replace=False)
            k[pick] += 1 if diff > 0 else -1
        if k.sum() != n_obs:
            raise ValueError("cannot realise the requested observation count")
    w_cap = max(int(max_window * n_cameras), 2)
    m = np.maximum(spread * n_cameras / k * rng.lognormal(0.0, 0.5, n_points), 1.0)
    w = np.minimum(np.maximum(np.rint(m * k).astype(np.int64), k), max(w_cap, int(k.max())))
    centre = rng.integers(0, n_cameras, n_points)
    pt = np.repeat(np.arange(n_points, dtype=np.int64), k)
    first = np.cumsum(k) - k
    j = np.arange(n_obs, dtype=np.int64) - np.repeat(first, k)
    # k distinct views out of the window: the j-th of k equal strata of the window, a random view inside the stratum
    ww, kk = np.repeat(w, k), np.repeat(k, k)
    lo = (j * ww) // kk
    hi = np.maximum(((j + 1) * ww) // kk, lo + 1)
    off = lo + (rng.random(n_obs) * (hi - lo)).astype(np.int64)
    cam = (np.repeat(centre - w // 2, k) + off) % n_cameras
    return cam.astype(np.int32), pt.astype(np.int32), k, centre, w


# ---- scenes -----------------------------------------------------------------
def _look_at(C, target, up=np.array([0.0, 1.0, 0.0])):
    z = target - C
    z /= np.linalg.norm(z, axis=1, keepdims=True)
    x = np.cross(np.broadcast_to(up, z.shape), z)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    y = np.cross(z, x)
    Rm = np.stack([x, y, z], axis=1)  # rows are the camera axes in world coordinates
    return Rotation.from_matrix(Rm).as_rotvec()


def make_problem(n_cameras: int, n_points: int, n_obs: int, seed: int, *,
                 scene: str = "ring", spread: float = 0.12, pixel_noise: float = 0.5,
                 perturb: float = 1.0, models=None, shared_group_size: int = 1, shared_group_sizes=None,
                 intrinsics_to_optimize: int = abi.INTRINSICS_DEFAULT, heavy_tail: float = 0.0) -> Problem:
    """Build a seeded synthetic problem.

    scene "allsee": the reference test recipe (config 1), every view sees every track.
    scene "ring":   landmark scene with windowed visibility (configs 2-4).
    models: optional list of (camera_model, fraction); default all PINHOLE.
    shared_group_size: >1 makes consecutive cameras share an intrinsics group.
    shared_group_sizes: (lo, hi) -- groups of consecutive cameras with sizes drawn log-uniformly from [lo, hi]
             ("intrinsics groups of 1-200 views"; a group of one view is a private group).
    perturb: scale of the initial perturbation away from the generating
             parameters (1.0 = SURVEY 8d: points/positions 0.25 % of the scene
             depth, angle-axis 0.005 rad)."""
    rng = np.random.default_rng(seed)
    if scene == "allsee":
        C = 10.0 * rng.uniform(-1, 1, (n_cameras, 3))
        aa = 0.2 * rng.uniform(-1, 1, (n_cameras, 3))
        X = rng.uniform(-1, 1, (n_points, 3))
        # keep the cloud in front of every camera: depth offset as in the
        # reference recipe, scaled with the camera spread
        X[:, 2] += 20.0 + 10.0
        cam = np.tile(np.arange(n_cameras, dtype=np.int32), n_points)
        pt = np.repeat(np.arange(n_points, dtype=np.int32), n_cameras)
        if n_obs != n_cameras * n_points:
            raise ValueError("allsee scene needs n_obs == n_cameras * n_points")
        depth = 20.0
        f = np.full(n_cameras, 800.0)
        pp = np.full((n_cameras, 2), 500.0)
        k1 = np.zeros(n_cameras)
        k2 = np.zeros(n_cameras)
    elif scene == "ring":
        radius = 100.0
        phi = (np.arange(n_cameras) + rng.uniform(-0.3, 0.3, n_cameras)) * (2 * np.pi / n_cameras)
        rr = radius * (1.0 + rng.uniform(-0.08, 0.08, n_cameras))
        C = np.stack([rr * np.cos(phi), radius * rng.uniform(-0.15, 0.15, n_cameras),
                      rr * np.sin(phi)], 1)
        target = radius * 0.1 * rng.normal(size=(n_cameras, 3))
        aa = _look_at(C, target)
        X = radius * 0.3 * rng.uniform(-1, 1, (n_points, 3))
        cam, pt, _ = _visibility(rng, n_cameras, n_points, n_obs, spread, heavy_tail)
        depth = radius
        f = rng.uniform(600, 900, n_cameras)
        pp = np.zeros((n_cameras, 2))
        k1 = rng.uniform(-0.1, 0.0, n_cameras)
        k2 = rng.uniform(0.0, 0.02, n_cameras)
    elif scene == "street":
        # A closed path (a circle whose perimeter is one unit per view) photographed from the inside out: every view
        # looks away from the centre at facades / landmarks beyond the path, a track sits in front of the middle of
        # its window of consecutive views at a depth proportional to the window (a wide window = a far landmark), so
        # every view of the window has it inside a ~100 degree field of view.  Neighbouring views share most tracks,
        # distant views none: the sequence structure of real collections -- S is a band, and the
        # chain's long-wavelength bending modes are what PCG with a block-Jacobi preconditioner is slow on.
        cam, pt, _, centre, w = _visibility_street(rng, n_cameras, n_points, n_obs, spread, heavy_tail)
        R0 = n_cameras / (2 * np.pi)
        phi = (np.arange(n_cameras) + rng.uniform(-0.2, 0.2, n_cameras)) * (2 * np.pi / n_cameras)
        rr = R0 + rng.uniform(-0.3, 0.3, n_cameras)
        C = np.stack([rr * np.cos(phi), rng.uniform(-0.3, 0.3, n_cameras), rr * np.sin(phi)], 1)
        outward = np.stack([np.cos(phi), np.zeros(n_cameras), np.sin(phi)], 1)
        aa = _look_at(C, C + 10.0 * outward + rng.normal(0.0, 0.6, (n_cameras, 3)))
        z = np.maximum(6.0, 0.55 * w) * rng.uniform(1.0, 1.6, n_points)       # depth beyond the path
        pphi = (centre + rng.uniform(-0.5, 0.5, n_points)) * (2 * np.pi / n_cameras)
        X = np.stack([(R0 + z) * np.cos(pphi), z * rng.uniform(-0.35, 0.35, n_points), (R0 + z) * np.sin(pphi)], 1)
        depth = 12.0
        f = rng.uniform(600, 900, n_cameras)
        pp = np.zeros((n_cameras, 2))
        k1 = rng.uniform(-0.1, 0.0, n_cameras)
        k2 = rng.uniform(0.0, 0.02, n_cameras)
    else:
        raise ValueError(scene)

    # intrinsics groups
    if shared_group_sizes is not None:
        lo, hi = shared_group_sizes
        grng = np.random.default_rng(seed + 7919)  # its own stream: the scene does not depend on the grouping
        camera_group = np.empty(n_cameras, dtype=np.int32)
        c = g = 0
        while c < n_cameras:
            size = int(round(np.exp(grng.uniform(np.log(lo), np.log(hi)))))
            size = max(lo, min(hi, size, n_cameras - c))
            camera_group[c:c + size] = g
            c += size
            g += 1
    elif shared_group_size > 1:
        camera_group = (np.arange(n_cameras) // shared_group_size).astype(np.int32)
    else:
        camera_group = np.arange(n_cameras, dtype=np.int32)
    n_groups = int(camera_group.max()) + 1 if n_cameras else 0
    group_model = np.zeros(n_groups, dtype=np.int32)
    if models:
        edges = np.cumsum([fr for _, fr in models])
        u = rng.random(n_groups)
        for g in range(n_groups):
            group_model[g] = models[int(np.searchsorted(edges, u[g] * edges[-1]))][0]
    sizes = np.array([abi.INTRINSICS_SIZE[m] for m in group_model], dtype=np.int32)
    group_offset = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    intr = np.zeros(int(group_offset[-1]))
    rep = np.zeros(n_groups, dtype=np.int64)  # representative camera of each group
    rep[camera_group[::-1]] = np.arange(n_cameras)[::-1]
    for g in range(n_groups):
        o, m, c = group_offset[g], group_model[g], rep[g]
        if m in (abi.PINHOLE, abi.PINHOLE_RADIAL_TANGENTIAL, abi.FISHEYE):
            intr[o:o + 5] = [f[c], 1.0, 0.0, pp[c, 0], pp[c, 1]]
            if m == abi.PINHOLE:
                intr[o + 5:o + 7] = [k1[c], k2[c]]
            elif m == abi.PINHOLE_RADIAL_TANGENTIAL:
                intr[o + 5:o + 10] = [k1[c], k2[c], 0.1 * k2[c], 1e-3 * rng.normal(), 1e-3 * rng.normal()]
            else:
                intr[o + 5:o + 9] = [0.3 * k1[c], 0.3 * k2[c], 0.0, 0.0]
        elif m == abi.FOV:
            intr[o:o + 5] = [f[c], 1.0, pp[c, 0], pp[c, 1], rng.uniform(0.05, 0.4)]
        else:
            intr[o:o + 5] = [f[c], 1.0, pp[c, 0], pp[c, 1], rng.uniform(-2e-7, 0.0)]

    points = np.concatenate([X, np.ones((n_points, 1))], 1)
    prob = Problem(
        extrinsics=np.concatenate([C, aa], 1), camera_group=camera_group,
        camera_flags=np.zeros(n_cameras, np.uint8), group_model=group_model,
        group_offset=group_offset, intrinsics=intr,
        intrinsics_constant=np.zeros(intr.shape[0], np.uint8), points=points,
        point_constant=np.zeros(n_points, np.uint8), obs_camera=cam, obs_point=pt,
        obs_xy=np.zeros((cam.shape[0], 2)))
    prob.set_intrinsics_to_optimize(intrinsics_to_optimize)

    # exact projections + pixel noise
    xy = project(prob)
    xy += pixel_noise * rng.normal(size=xy.shape)
    prob.obs_xy = np.ascontiguousarray(xy)
    prob.meta["truth"] = dict(extrinsics=prob.extrinsics.copy(), intrinsics=prob.intrinsics.copy(),
                              points=prob.points.copy())
    # initial perturbation
    s = 0.0025 * depth * perturb
    prob.points[:, :3] += s * rng.normal(size=(n_points, 3))
    prob.extrinsics[:, :3] += s * rng.normal(size=(n_cameras, 3))
    prob.extrinsics[:, 3:] += 0.005 * perturb * rng.normal(size=(n_cameras, 3))
    prob.meta.update(dict(seed=seed, scene=scene, spread=spread))
    return prob


CONFIG5_INTRINSICS = (abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS | abi.INTRINSICS_RADIAL_DISTORTION
                      | abi.INTRINSICS_TANGENTIAL_DISTORTION)


def config5(**kw) -> Problem:
    """BASELINE.json configs[4] / SURVEY 8(d) config 5: the config-4 topology (1778 views, 993 923 tracks, 5 001 946
    observations), cameras 50 % PINHOLE / 25 % PINHOLE_RADIAL_TANGENTIAL / 25 % FISHEYE, shared intrinsics groups of 2-200
    consecutive views (log-uniform sizes; "groups of 1-200 views (shared)" -- a group of one is not shared),
    intrinsics_to_optimize = ALL minus skew and aspect ratio."""
    kw.setdefault("models", [(abi.PINHOLE, 0.5), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.25), (abi.FISHEYE, 0.25)])
    kw.setdefault("shared_group_sizes", (2, 200))
    kw.setdefault("intrinsics_to_optimize", CONFIG5_INTRINSICS)
    return config("venice1778", **kw)


def config(name: str, **kw) -> Problem:
    """The BASELINE.json configurations by name."""
    nc, npt, nobs = CONFIG_SIZES[name]
    if name == "tiny":
        return make_problem(nc, npt, nobs, seed=1, scene="allsee", **kw)
    seeds = {"ladybug49": 49, "alamo": 570, "venice1778": 1778, "venice1778_heavy": 1778}
    spreads = {"ladybug49": 0.35, "alamo": 0.15, "venice1778": 0.12, "venice1778_heavy": 0.12, "venice1778_street": 0.025}
    kw.setdefault("spread", spreads[name])
    if name == "venice1778_heavy":
        kw.setdefault("heavy_tail", 0.002)
    if name == "venice1778_street":
        kw.setdefault("heavy_tail", 0.002)
        return make_problem(nc, npt, nobs, seed=1778, scene="street", **kw)
    return make_problem(nc, npt, nobs, seed=seeds[name], scene="ring", **kw)


def make_two_view_batch(n_pairs: int, seed: int, *, min_corr: int = 30, max_corr: int = 400,
                        models=None, free_intrinsics: float = 0.0, pixel_noise: float = 0.5,
                        point_noise: float = 0.02) -> abi.TwoViewBatch:
    """Seeded view pairs as two_view_match_geometric_verification.cc hands them to
    BundleAdjustTwoViews (:285): camera 1 at the origin looking along +z, camera 2 a baseline away
    with a small rotation, correspondences of points 4-10 units in front, pixels with noise, the
    3D points as a (noisy) triangulation.  free_intrinsics: fraction of the pairs whose focal
    lengths are optimised (TwoViewBundleAdjustmentOptions::constant_camera*_intrinsics = false)."""
    rng = np.random.default_rng(seed)
    counts = rng.integers(min_corr, max_corr + 1, n_pairs)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    N = int(ptr[-1])
    e1 = np.zeros((n_pairs, 6))
    e2 = np.zeros((n_pairs, 6))
    e2[:, :3] = rng.normal(size=(n_pairs, 3)) * [1.0, 0.2, 0.2] + [1.0, 0, 0]
    e2[:, 3:] = 0.15 * rng.normal(size=(n_pairs, 3))
    m1 = np.zeros(n_pairs, np.int32)
    m2 = np.zeros(n_pairs, np.int32)
    if models:
        edges = np.cumsum([fr for _, fr in models])
        for arr in (m1, m2):
            u = rng.random(n_pairs) * edges[-1]
            arr[:] = [models[int(np.searchsorted(edges, x))][0] for x in u]

    def intrinsics(model):
        K = np.zeros((n_pairs, 10))
        f = rng.uniform(600, 900, n_pairs)
        for p in range(n_pairs):
            m = model[p]
            if m in (abi.PINHOLE, abi.PINHOLE_RADIAL_TANGENTIAL, abi.FISHEYE):
                K[p, :5] = [f[p], 1.0, 0.0, 500.0, 400.0]
                if m == abi.PINHOLE:
                    K[p, 5:7] = [-0.05, 0.01]
                elif m == abi.PINHOLE_RADIAL_TANGENTIAL:
                    K[p, 5:10] = [-0.05, 0.01, 0.001, 1e-3, -1e-3]
                else:
                    K[p, 5:9] = [-0.02, 0.003, 0.0, 0.0]
            elif m == abi.FOV:
                K[p, :5] = [f[p], 1.0, 500.0, 400.0, 0.2]
            else:
                K[p, :5] = [f[p], 1.0, 500.0, 400.0, -1e-7]
        return K

    k1, k2 = intrinsics(m1), intrinsics(m2)
    X = np.concatenate([rng.uniform(-2, 2, (N, 2)), rng.uniform(4, 10, (N, 1)), np.ones((N, 1))], 1)
    pair_of = np.repeat(np.arange(n_pairs), counts)

    def proj(ext, model, K):
        Rm = Rotation.from_rotvec(ext[pair_of, 3:6]).as_matrix()
        a = X[:, :3] - ext[pair_of, :3]
        q = np.einsum("nij,nj->ni", Rm, a)
        out = np.empty((N, 2))
        for m in np.unique(model):
            sel = np.nonzero(model[pair_of] == m)[0]
            out[sel] = _distort(int(m), K[pair_of[sel], :abi.INTRINSICS_SIZE[m]], q[sel])
        return out

    f1 = proj(e1, m1, k1) + pixel_noise * rng.normal(size=(N, 2))
    f2 = proj(e2, m2, k2) + pixel_noise * rng.normal(size=(N, 2))
    pts = X.copy()
    pts[:, :3] += point_noise * rng.normal(size=(N, 3))
    e2_0 = e2.copy()
    e2_0[:, :3] += 0.02 * rng.normal(size=(n_pairs, 3))
    e2_0[:, 3:] += 0.004 * rng.normal(size=(n_pairs, 3))
    free = rng.random(n_pairs) < free_intrinsics
    c1 = np.where(free, 0, 1).astype(np.uint8)
    c2 = np.where(free, 0, 1).astype(np.uint8)
    k1_0, k2_0 = k1.copy(), k2.copy()
    k1_0[free, 0] *= 1.01
    k2_0[free, 0] *= 0.99
    return abi.TwoViewBatch(e1, e2_0, m1, m2, k1_0, k2_0, c1, c2, ptr, f1, f2, pts)


def make_two_view_angular_batch(n_pairs: int, seed: int, *, min_corr: int = 30, max_corr: int = 300,
                                noise: float = 1e-3, rotation_error: float = 0.03, position_error: float = 0.08,
                                rotation_scale: float = 0.05):
    """Seeded view pairs for BundleAdjustTwoViewsAngular: view 1 at the origin, view 2 at a unit-norm
    position with a small rotation, points 3-9 units in front of both; features are NORMALISED image
    coordinates (angular_epipolar_error.h works on calibrated rays) with Gaussian noise; the starting
    relative pose is the true one perturbed.  Returns (batch, true rotation [P, 3], true position [P, 3])."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    counts = rng.integers(min_corr, max_corr + 1, n_pairs)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    N = int(ptr[-1])
    # small relative rotations: the functor's second term is (R f2) . T (R^T f2) as the reference writes it
    # (angular_epipolar_error.h:71-72), which only stays positive -- and the residual zero at the true pose --
    # while R f2 and R^T f2 are close
    rot_true = rotation_scale * rng.normal(size=(n_pairs, 3))
    pos_true = rng.normal(size=(n_pairs, 3)) * [1.0, 0.3, 0.3] + [1.5, 0, 0]
    pos_true /= np.linalg.norm(pos_true, axis=1, keepdims=True)
    f1 = np.zeros((N, 2))
    f2 = np.zeros((N, 2))
    for p in range(n_pairs):
        n = int(counts[p])
        X = np.stack([rng.uniform(-2.5, 2.5, n), rng.uniform(-2.0, 2.0, n), rng.uniform(3.0, 9.0, n)], 1)
        R = Rotation.from_rotvec(rot_true[p])
        X2 = R.apply(X - pos_true[p])
        f1[ptr[p]:ptr[p + 1]] = X[:, :2] / X[:, 2:3]
        f2[ptr[p]:ptr[p + 1]] = X2[:, :2] / X2[:, 2:3]
    f1 += noise * rng.normal(size=f1.shape)
    f2 += noise * rng.normal(size=f2.shape)
    rot0 = rot_true + rotation_error * rng.normal(size=rot_true.shape)
    pos0 = pos_true + position_error * rng.normal(size=pos_true.shape)
    pos0 /= np.linalg.norm(pos0, axis=1, keepdims=True)
    return abi.TwoViewAngularBatch(rot0, pos0, ptr, f1, f2), rot_true, pos_true
