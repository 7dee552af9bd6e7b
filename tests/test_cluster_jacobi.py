"""CLUSTER_JACOBI on problems with shared intrinsics blocks (theia_mi355_ba.h, cluster_precond.h): a cluster is a shared
block together with the views that share it, inverted exactly.

CPU: the oracle's restatement -- far fewer PCG iterations than SCHUR_JACOBI, the same LM trajectory within the
tolerance PCG's inexact solves allow, fall-back to SCHUR_JACOBI without shared blocks.
GPU: the device (gather from the blocks of S, batched tile-dataflow Cholesky, dataflow substitution per PCG iteration)
against the oracle: same iteration counts, cost 1e-9."""
import numpy as np
import pytest

from oracle import oracle
from theiasfm_amd import abi, synth

BITS = (abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS | abi.INTRINSICS_RADIAL_DISTORTION
        | abi.INTRINSICS_TANGENTIAL_DISTORTION)
MODELS = [(abi.PINHOLE, 0.5), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.25), (abi.FISHEYE, 0.25)]


def shared_problem(n_views=60, groups=(2, 14), seed=5, per_view=800):
    # BASELINE config 5 in small: mixed models, shared groups of several sizes, every intrinsic but skew / aspect free
    return synth.make_problem(n_views, n_views * 160, n_views * per_view, seed=seed, scene="ring", spread=0.2, models=MODELS,
                              shared_group_sizes=groups, intrinsics_to_optimize=BITS)


def options(pre, **kw):
    o = dict(linear_solver_type=abi.ITERATIVE_SCHUR, point_dof=3, use_inner_iterations=0, preconditioner_type=pre,
             schur_mode=abi.SCHUR_EXPLICIT, max_num_iterations=8, function_tolerance=-1.0, gradient_tolerance=-1.0,
             parameter_tolerance=-1.0)
    o.update(kw)
    return abi.default_options(**o)


def test_oracle_cluster_jacobi_cuts_the_pcg_iterations():
    prob = shared_problem()
    res = {}
    for pre in (abi.PRECOND_SCHUR_JACOBI, abi.PRECOND_CLUSTER_JACOBI, abi.PRECOND_CLUSTER_TRIDIAGONAL):
        p = prob.copy()
        st, s = oracle.solve(p, options(pre))
        assert st == 0 and s.success == 1
        res[pre] = (s, p)
    jac, clu = res[abi.PRECOND_SCHUR_JACOBI][0], res[abi.PRECOND_CLUSTER_JACOBI][0]
    assert clu.num_linear_solver_iterations * 3 < jac.num_linear_solver_iterations
    # both solve the same reduced systems to the same forcing tolerance: the trajectories agree to what that leaves
    assert abs(clu.final_cost - jac.final_cost) < 1e-4 * jac.final_cost
    assert clu.num_successful_steps == jac.num_successful_steps
    tri = res[abi.PRECOND_CLUSTER_TRIDIAGONAL][0]
    # (the oracle's OpenMP reductions are not bit-reproducible from run to run)
    assert abs(tri.final_cost - clu.final_cost) <= 1e-12 * clu.final_cost
    assert tri.num_linear_solver_iterations == clu.num_linear_solver_iterations


def test_oracle_cluster_jacobi_without_shared_blocks_is_schur_jacobi():
    prob = synth.config("ladybug49")
    outs = []
    for pre in (abi.PRECOND_SCHUR_JACOBI, abi.PRECOND_CLUSTER_JACOBI):
        p = prob.copy()
        st, s = oracle.solve(p, options(pre, max_num_iterations=5))
        assert st == 0
        outs.append((s.final_cost, s.num_linear_solver_iterations, p.extrinsics.copy()))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-12 * outs[0][0] and outs[0][1] == outs[1][1]
    assert np.abs(outs[0][2] - outs[1][2]).max() <= 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["mixed_sizes", "one_big_cluster", "pairs", "huber_dof4"])
def test_device_cluster_jacobi_matches_oracle(case):
    from theiasfm_amd import lib
    kw = {}
    if case == "mixed_sizes":
        prob = shared_problem()
    elif case == "one_big_cluster":
        prob = shared_problem(n_views=90, groups=(60, 60), seed=9, per_view=500)   # 60 views: 368 unknowns, 6 tile rows
    elif case == "pairs":
        prob = shared_problem(n_views=40, groups=(2, 2), seed=3)
    else:
        prob = shared_problem(n_views=36, groups=(3, 9), seed=11)
        kw = dict(loss_function_type=abi.LOSS_HUBER, robust_loss_width=3.0, point_dof=4)
    # the reference's stopping rules on (1e-6 / 1e-10 / 1e-8): iterations at the fixed point, where accepting a step is a
    # matter of round-off, would make the step counts of two correct implementations differ
    kw.update(function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, max_num_iterations=12)
    o = options(abi.PRECOND_CLUSTER_JACOBI, **kw)
    a, b = prob.copy(), prob.copy()
    st_d, s_d = lib.solve(a, o)
    st_o, s_o = oracle.solve(b, o)
    assert st_d == st_o == 0, (s_d.message, s_o.message)
    assert s_d.num_iterations == s_o.num_iterations and s_d.num_successful_steps == s_o.num_successful_steps
    assert abs(int(s_d.num_linear_solver_iterations) - int(s_o.num_linear_solver_iterations)) <= 1
    assert abs(s_d.final_cost - s_o.final_cost) <= 1e-9 * s_o.final_cost
    assert abs(s_d.final_rmse - s_o.final_rmse) <= 1e-9
    assert np.abs(a.extrinsics - b.extrinsics).max() <= 1e-6 * 100.0
    assert np.abs(a.intrinsics - b.intrinsics).max() <= 1e-6 * max(1.0, np.abs(b.intrinsics).max())
    # and it is doing something: SCHUR_JACOBI on the same problem needs several times the PCG iterations
    c = prob.copy()
    st_j, s_j = lib.solve(c, options(abi.PRECOND_SCHUR_JACOBI, **kw))
    assert st_j == 0 and s_j.num_linear_solver_iterations > 2 * s_d.num_linear_solver_iterations


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [abi.SCHUR_IMPLICIT, abi.SCHUR_AUTO])
def test_device_cluster_jacobi_with_the_matrix_free_operator(mode):
    """schur_mode auto / implicit: the operator is matrix-free and only the blocks INSIDE the clusters are formed for the
    preconditioner to factor; same PCG, same trajectory as with the formed S."""
    from theiasfm_amd import lib
    prob = shared_problem()
    a, b, c = prob.copy(), prob.copy(), prob.copy()
    tol = dict(function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, max_num_iterations=12)
    st_d, s_d = lib.solve(a, options(abi.PRECOND_CLUSTER_JACOBI, schur_mode=mode, **tol))
    st_e, s_e = lib.solve(c, options(abi.PRECOND_CLUSTER_JACOBI, schur_mode=abi.SCHUR_EXPLICIT, **tol))
    st_o, s_o = oracle.solve(b, options(abi.PRECOND_CLUSTER_JACOBI, **tol))
    assert st_d == st_e == st_o == 0, (s_d.message, s_o.message)
    assert s_d.num_matrix_free_iterations == s_d.num_iterations and s_e.num_matrix_free_iterations == 0
    assert 0 < s_d.num_schur_blocks < s_e.num_schur_blocks       # the clusters' blocks only
    assert s_d.num_iterations == s_o.num_iterations and s_d.num_successful_steps == s_o.num_successful_steps
    assert abs(int(s_d.num_linear_solver_iterations) - int(s_o.num_linear_solver_iterations)) <= 1
    assert abs(s_d.final_cost - s_o.final_cost) <= 1e-9 * s_o.final_cost
    assert abs(s_d.final_cost - s_e.final_cost) <= 1e-9 * s_e.final_cost
    assert np.abs(a.extrinsics - b.extrinsics).max() <= 1e-6 * 100.0


@pytest.mark.gpu
def test_device_cluster_jacobi_needs_its_blocks():
    """a handle created for SCHUR_JACOBI with the matrix-free operator has no blocks of S at all: asking that handle for
    CLUSTER_JACOBI at solve time falls back to SCHUR_JACOBI instead of failing"""
    from theiasfm_amd import lib
    prob = shared_problem(n_views=30, groups=(2, 8), seed=2)
    o_j = options(abi.PRECOND_SCHUR_JACOBI, schur_mode=abi.SCHUR_IMPLICIT, max_num_iterations=4)
    o_c = options(abi.PRECOND_CLUSTER_JACOBI, schur_mode=abi.SCHUR_IMPLICIT, max_num_iterations=4)
    sv = lib.Solver(prob.copy(), o_j)
    st1, s1 = sv.solve(o_j)
    sv.reset()
    st2, s2 = sv.solve(o_c)
    sv.close()
    assert st1 == st2 == 0
    assert s1.final_cost == s2.final_cost and s1.num_linear_solver_iterations == s2.num_linear_solver_iterations


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", [abi.SCHUR_AUTO, abi.SCHUR_EXPLICIT])
def test_sharded_cluster_jacobi_matches_single_rank(world, mode):
    """tracks sharded over `world` ranks (threads on one device, the all-reduce hook sums their buffers): the clusters'
    blocks of S are partial sums on every rank and travel in the all-reduced system like the rest of it; every rank
    factors the same clusters and ends where the single-rank solve ends"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_sharded import _run_sharded
    from theiasfm_amd import lib
    prob = shared_problem(n_views=40, groups=(2, 12), seed=4)
    tol = dict(function_tolerance=1e-6, gradient_tolerance=1e-10, parameter_tolerance=1e-8, max_num_iterations=12)
    o = options(abi.PRECOND_CLUSTER_JACOBI, schur_mode=mode, **tol)
    single = prob.copy()
    st1, s1 = lib.solve(single, o)
    assert st1 == 0
    results = _run_sharded(prob, o, world)
    for st_r, s_r in results:
        assert st_r == 0 and s_r.success == 1
        assert s_r.num_iterations == s1.num_iterations and s_r.num_successful_steps == s1.num_successful_steps
        assert abs(int(s_r.num_linear_solver_iterations) - int(s1.num_linear_solver_iterations)) <= 1
        assert abs(s_r.final_cost - s1.final_cost) <= 1e-9 * s1.final_cost
        assert s_r.final_cost == results[0][1].final_cost
