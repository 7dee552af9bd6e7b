"""CPU: the algebra behind the engine's compact planes (theiasfm_amd/csrc/kernels.h compact_forward /
compact_ax / compact_at / compact_backward) against the reference functor's own Jacobian.

`oracle.evaluate` differentiates ReprojectionError (reprojection_error.h:51-95, pinhole_camera_model.h:181-257) with dual
numbers -- no analytic formula of this repository is involved.  For a PINHOLE view with unit aspect ratio and zero skew the
engine does not store the 2 x 9 block d r / d [C, angle-axis, f, k1, k2] but forms it from the point block
Jp' = d r / d X[0:3], the normalised image point p_n, the track and the view:

    A_pos = -w Jp'            A_rot = -Jp' [X - w C]x R^T Jl            A_int = p_n [dist, f r^2, f r^4]

and applies it through a transformed view vector and per-view moment sums.  This file restates those maps in numpy
and holds them to the dual-number Jacobian: the columns themselves, A x for random x, A^T t for random t, with Jacobi
scales on every column -- including a view with zero rotation (the first-order branch of AngleAxisRotatePoint)."""
import numpy as np
import pytest

from oracle import oracle
from theiasfm_amd import abi, synth


def skew(a):
    return np.array([[0.0, -a[2], a[1]], [a[2], 0.0, -a[0]], [-a[1], a[0], 0.0]])


def rodrigues(w):
    th = np.linalg.norm(w)
    K = skew(w)
    if th * th <= np.finfo(float).eps:
        return np.eye(3) + K, np.eye(3)  # the first-order branch: R = I + [w]x, the executed derivative has Jl = I
    R = np.eye(3) + np.sin(th) / th * K + (1.0 - np.cos(th)) / th ** 2 * K @ K
    Jl = np.eye(3) + (1.0 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K
    return R, Jl


def view_record(prob, c, s_c):
    """what the prepared camera record holds for view c: R, C, Jl diag(s_rot), f, k1, k2 and the column scales"""
    C0 = prob.extrinsics[c, :3]
    R, Jl = rodrigues(prob.extrinsics[c, 3:6])
    K = prob.intrinsics[prob.group_offset[prob.camera_group[c]]:][:7]
    return dict(R=R, C=C0, Jl_s=Jl * s_c[3:6][None, :], f=K[0], k1=K[5], k2=K[6], s_pos=s_c[0:3], s_f=s_c[6], s_k1=s_c[7],
                s_k2=s_c[8])


def compact_forward(P, x):
    eta = P["R"].T @ (P["Jl_s"] @ x[3:6])
    kappa = np.cross(eta, P["C"]) + P["s_pos"] * x[0:3]
    a0 = P["s_f"] * x[6]
    return np.concatenate([kappa, eta, [a0, P["k1"] * a0 + P["f"] * P["s_k1"] * x[7], P["k2"] * a0 + P["f"] * P["s_k2"] * x[8]]])


def compact_ax(g, X, isp, Jp_st, pn, r2=None):
    c = (np.cross(g[3:6], X[:3]) - X[3] * g[0:3]) * isp
    r2 = pn @ pn if r2 is None else r2
    return Jp_st @ c + pn * (g[6] + r2 * (g[7] + r2 * g[8]))


def compact_at(X, isp, Jp_st, pn, t, r2=None):
    h = isp * (Jp_st.T @ t)
    r2, pt = (pn @ pn if r2 is None else r2), pn @ t
    return np.concatenate([-X[3] * h, np.cross(X[:3], h), [pt, r2 * pt, r2 * r2 * pt]])


def compact_backward(P, s):
    y = np.empty(9)
    y[0:3] = P["s_pos"] * s[0:3]
    y[3:6] = P["Jl_s"].T @ (P["R"] @ (s[3:6] + np.cross(P["C"], s[0:3])))
    y[6] = P["s_f"] * (s[6] + P["k1"] * s[7] + P["k2"] * s[8])
    y[7] = P["s_k1"] * P["f"] * s[7]
    y[8] = P["s_k2"] * P["f"] * s[8]
    return y


@pytest.fixture(scope="module")
def scene():
    prob = synth.make_problem(6, 60, 300, seed=5, scene="ring", spread=0.5)
    prob.extrinsics[0, 3:6] = 0.0  # a view with zero rotation: the first-order branch
    prob.points[:, 3] = 1.0 + 0.1 * np.random.default_rng(1).standard_normal(prob.num_points)  # w != 1: homogeneous points
    r, J, valid = oracle.evaluate(prob)
    assert valid.all()
    return prob, J


def full_block(J):
    """d r / d [C, angle-axis, f, k1, k2] and d r / d X[0:3] of one observation from the dual-number Jacobian
    [ext(6) | intrinsics(10) | X(4)]"""
    return np.concatenate([J[:, 0:6], J[:, [6, 11, 12]]], axis=1), J[:, 16:19]


def normalised_point(prob, o):
    c, p = prob.obs_camera[o], prob.obs_point[o]
    R, _ = rodrigues(prob.extrinsics[c, 3:6])
    X = prob.points[p]
    q = R @ (X[:3] - X[3] * prob.extrinsics[c, :3])
    return q[:2] / q[2]


def test_columns_of_the_camera_block_from_the_point_block(scene):
    prob, J = scene
    for o in range(prob.num_observations):
        c, p = prob.obs_camera[o], prob.obs_point[o]
        A, Jp = full_block(J[o])
        X = prob.points[p]
        C0 = prob.extrinsics[c, :3]
        R, Jl = rodrigues(prob.extrinsics[c, 3:6])
        K = prob.intrinsics[prob.group_offset[prob.camera_group[c]]:][:7]
        assert K[1] == 1.0 and K[2] == 0.0  # unit aspect ratio, zero skew: what the compact form requires
        a = X[:3] - X[3] * C0
        scale = np.abs(A).max()
        np.testing.assert_allclose(-X[3] * Jp, A[:, 0:3], rtol=0, atol=1e-11 * scale)
        np.testing.assert_allclose(-Jp @ skew(a) @ R.T @ Jl, A[:, 3:6], rtol=0, atol=1e-10 * scale)
        pn = normalised_point(prob, o)
        r2 = pn @ pn
        dist = 1.0 + K[5] * r2 + K[6] * r2 * r2
        np.testing.assert_allclose(np.outer(pn, [dist, K[0] * r2, K[0] * r2 * r2]), A[:, 6:9], rtol=0, atol=1e-10 * scale)


def test_products_through_the_transformed_vector_and_the_moments(scene):
    prob, J = scene
    rng = np.random.default_rng(7)
    s_c = 1.0 / (1.0 + rng.uniform(0.0, 50.0, size=(prob.num_cameras, 9)))   # Jacobi scales of the camera columns
    s_p = 1.0 / (1.0 + rng.uniform(0.0, 50.0, size=(prob.num_points, 3)))    # ... and of the point columns
    x = rng.standard_normal((prob.num_cameras, 9))
    g = np.stack([compact_forward(view_record(prob, c, s_c[c]), x[c]) for c in range(prob.num_cameras)])
    y_full = np.zeros((prob.num_cameras, 9))
    moments = np.zeros((prob.num_cameras, 9))
    for o in range(prob.num_observations):
        c, p = prob.obs_camera[o], prob.obs_point[o]
        A, Jp = full_block(J[o])
        A_s, Jp_st = A * s_c[c][None, :], Jp * s_p[p][None, :]  # what the engine's planes would hold / hold
        X, isp, pn = prob.points[p], 1.0 / s_p[p], normalised_point(prob, o)
        u = compact_ax(g[c], X, isp, Jp_st, pn)
        np.testing.assert_allclose(u, A_s @ x[c], rtol=0, atol=1e-10 * np.abs(A_s).max() * np.abs(x[c]).max())
        t = rng.standard_normal(2)
        y_full[c] += A_s.T @ t
        moments[c] += compact_at(X, isp, Jp_st, pn, t)
    for c in range(prob.num_cameras):
        y = compact_backward(view_record(prob, c, s_c[c]), moments[c])
        np.testing.assert_allclose(y, y_full[c], rtol=0, atol=1e-10 * max(1.0, np.abs(y_full[c]).max()))


def test_the_maps_are_adjoint(scene):
    """<A x, t> = <x, A^T t> through the compact maps alone (no reference to the stored block)"""
    prob, J = scene
    rng = np.random.default_rng(11)
    c = 2
    s_c = 1.0 / (1.0 + rng.uniform(0.0, 50.0, size=9))
    P = view_record(prob, c, s_c)
    x = rng.standard_normal(9)
    lhs, mom = 0.0, np.zeros(9)
    for o in np.flatnonzero(prob.obs_camera == c):
        p = prob.obs_point[o]
        _, Jp = full_block(J[o])
        isp = np.ones(3)
        pn = normalised_point(prob, o)
        t = rng.standard_normal(2)
        lhs += compact_ax(compact_forward(P, x), prob.points[p], isp, Jp, pn) @ t
        mom += compact_at(prob.points[p], isp, Jp, pn, t)
    assert abs(lhs - x @ compact_backward(P, mom)) <= 1e-10 * max(1.0, abs(lhs))


def test_a_robust_loss_corrects_the_point_block_and_the_stored_pair(scene):
    """Ceres' corrector multiplies every column pair of an observation by one 2 x 2 matrix C: with Jp and p_n corrected and
    r^2 of the UNcorrected point handed over separately (DeviceView::compact == 2) the compact maps give C A x and (C A)^T t"""
    prob, J = scene
    rng = np.random.default_rng(13)
    s_c = 1.0 / (1.0 + rng.uniform(0.0, 50.0, size=(prob.num_cameras, 9)))
    s_p = 1.0 / (1.0 + rng.uniform(0.0, 50.0, size=(prob.num_points, 3)))
    for o in range(0, prob.num_observations, 7):
        c, p = prob.obs_camera[o], prob.obs_point[o]
        A, Jp = full_block(J[o])
        r = rng.standard_normal(2)
        C = 0.8 * (np.eye(2) - 0.3 * np.outer(r, r) / (r @ r))  # sqrt(rho') (I - alpha r r^T / |r|^2)
        A_s, Jp_st = C @ (A * s_c[c][None, :]), C @ (Jp * s_p[p][None, :])
        pn = normalised_point(prob, o)
        X, isp = prob.points[p], 1.0 / s_p[p]
        P = view_record(prob, c, s_c[c])
        x, t = rng.standard_normal(9), rng.standard_normal(2)
        u = compact_ax(compact_forward(P, x), X, isp, Jp_st, C @ pn, r2=pn @ pn)
        np.testing.assert_allclose(u, A_s @ x, rtol=0, atol=1e-10 * np.abs(A_s).max() * np.abs(x).max())
        y = compact_backward(P, compact_at(X, isp, Jp_st, C @ pn, t, r2=pn @ pn))
        np.testing.assert_allclose(y, A_s.T @ t, rtol=0, atol=1e-10 * max(1.0, np.abs(A_s).max()))
