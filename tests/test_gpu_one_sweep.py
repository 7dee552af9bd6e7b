"""-m gpu: the one-sweep matrix-free product (theiasfm_amd/csrc/mf_chunks.h, opt-in through TMI_BA_MF_ONE_SWEEP) against
the two-pass product it is meant to replace and against the oracle: same operator q = S p (Ceres'
ImplicitSchurComplement behind ceres::Solve, bundle_adjuster.cc:205), so identical PCG / LM iteration counts and
results to round-off, over the shapes its work items take -- packs of short slices, whole slices, slices cut into 2 .. 64
pieces (heavy tail: up to 400 views per track), tracks of more views than the row slots hold (a wavefront per track),
items of one unit and of several, constant cameras / tracks, block sizes 6 / 9 / 16 and both point
parameterisations."""
import os

import numpy as np
import pytest

from oracle import oracle
from theiasfm_amd import abi, lib, synth

pytestmark = pytest.mark.gpu


def solve(prob, one_sweep, **kw):
    old = os.environ.pop("TMI_BA_MF_ONE_SWEEP", None)
    if one_sweep:
        os.environ["TMI_BA_MF_ONE_SWEEP"] = "1"
    try:
        p = prob.copy()
        st, s = lib.solve(p, abi.default_options(linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_IMPLICIT, **kw))
        return st, s, p
    finally:
        os.environ.pop("TMI_BA_MF_ONE_SWEEP", None)
        if old is not None:
            os.environ["TMI_BA_MF_ONE_SWEEP"] = old


def with_landmarks(prob, n_long, seed):
    """+ n_long tracks near the middle of the scene that EVERY view sees (observed where the problem's current
    parameters project them, half a pixel of noise): longer than any row-slot arrangement holds when there are more
    than 512 views (256 for the 16-wide blocks), i.e. the wavefront-per-track units of mf_chunks.h"""
    rng = np.random.default_rng(seed)
    nc, np0, no0 = prob.num_cameras, prob.num_points, prob.num_observations
    centre = (prob.points[:, :3] / prob.points[:, 3:4]).mean(0)
    X = np.concatenate([centre + 0.02 * np.abs(centre).max() * rng.standard_normal((n_long, 3)), np.ones((n_long, 1))], 1)
    prob.points = np.ascontiguousarray(np.concatenate([prob.points, X]))
    prob.point_constant = np.concatenate([prob.point_constant, np.zeros(n_long, np.uint8)])
    prob.obs_camera = np.concatenate([prob.obs_camera, np.tile(np.arange(nc, dtype=np.int32), n_long)])
    prob.obs_point = np.concatenate([prob.obs_point, np.repeat(np.arange(np0, np0 + n_long, dtype=np.int32), nc)])
    prob.obs_xy = np.ascontiguousarray(np.concatenate([prob.obs_xy, np.zeros((n_long * nc, 2))]))
    new = np.arange(no0, no0 + n_long * nc)
    prob.obs_xy[new] = synth.project(prob, new) + 0.5 * rng.standard_normal((new.size, 2))
    return prob


CASES = {
    "ladybug49": lambda: (synth.config("ladybug49"), dict(point_dof=3)),
    "dof4": lambda: (synth.make_problem(40, 6000, 30000, seed=5, scene="ring", spread=0.5), dict(point_dof=4)),
    # tracks of 24-300 views: 16 and 64 lanes per track, slices with tail rows
    "heavy_tail": lambda: (synth.make_problem(320, 30000, 190000, seed=7, scene="ring", spread=0.6, heavy_tail=0.01), dict(point_dof=3)),
    # extrinsics only (D = 6), Huber
    "d6_huber": lambda: (synth.make_problem(60, 9000, 50000, seed=9, scene="ring", spread=0.4, intrinsics_to_optimize=abi.INTRINSICS_NONE),
                         dict(point_dof=3, loss_function_type=abi.LOSS_HUBER, robust_loss_width=2.0)),
    # radial-tangential cameras with every intrinsic free but skew / aspect ratio: D = 6 + 8 -> 16
    # 560 views, six tracks that all of them see (+ the 24-400 view tail): more rows than 64 lanes per track hold
    "landmarks": lambda: (with_landmarks(synth.make_problem(560, 12000, 70000, seed=13, scene="ring", spread=0.5, heavy_tail=0.01), 6, 14),
                          dict(point_dof=3)),
    # 16-wide blocks keep one row per wavefront in registers: the same beyond 256 views
    "d16_landmarks": lambda: (with_landmarks(synth.make_problem(300, 6000, 36000, seed=15, scene="ring", spread=0.5,
                                                                models=[(abi.PINHOLE_RADIAL_TANGENTIAL, 1.0)],
                                                                intrinsics_to_optimize=abi.INTRINSICS_ALL & ~(abi.INTRINSICS_SKEW | abi.INTRINSICS_ASPECT_RATIO)), 5, 16),
                              dict(point_dof=3)),
    "d16": lambda: (synth.make_problem(30, 4000, 22000, seed=11, scene="ring", spread=0.5, models=[(abi.PINHOLE_RADIAL_TANGENTIAL, 1.0)],
                                       intrinsics_to_optimize=abi.INTRINSICS_ALL & ~(abi.INTRINSICS_SKEW | abi.INTRINSICS_ASPECT_RATIO)),
                    dict(point_dof=4)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_one_sweep_product_equals_two_pass_and_oracle(name):
    prob, kw = CASES[name]()
    if name == "heavy_tail":
        prob.camera_flags[0] = abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT  # a view without a block
        prob.point_constant[::7] = 1
    kw = dict(max_num_iterations=6, use_inner_iterations=0, **kw)
    st2, s2, p2 = solve(prob, False, **kw)
    st1, s1, p1 = solve(prob, True, **kw)
    assert st1 == st2 == 0, (s1.message, s2.message)
    assert s1.num_matrix_free_iterations == s1.num_iterations == s2.num_iterations
    assert s1.num_linear_solver_iterations == s2.num_linear_solver_iterations
    assert s1.num_successful_steps == s2.num_successful_steps
    assert abs(s1.final_cost - s2.final_cost) <= 1e-10 * s2.final_cost
    scale = max(1.0, np.abs(p2.extrinsics).max())
    assert np.abs(p1.extrinsics - p2.extrinsics).max() <= 1e-8 * scale
    assert np.abs(p1.intrinsics - p2.intrinsics).max() <= 1e-8 * max(1.0, np.abs(p2.intrinsics).max())
    if kw["point_dof"] == 3:
        assert np.abs(p1.points - p2.points).max() <= 1e-7 * scale
    b = prob.copy()
    st_o, s_o = oracle.solve(b, abi.default_options(linear_solver_type=abi.ITERATIVE_SCHUR, **kw))
    assert st_o == 0 and s_o.num_iterations == s1.num_iterations
    assert abs(s1.final_cost - s_o.final_cost) <= 1e-9 * s_o.final_cost


def test_one_sweep_product_is_reproducible_to_the_bit():
    prob, kw = CASES["heavy_tail"]()
    kw = dict(max_num_iterations=4, use_inner_iterations=0, **kw)
    _, s_a, p_a = solve(prob, True, **kw)
    _, s_b, p_b = solve(prob, True, **kw)
    assert s_a.final_cost == s_b.final_cost and (p_a.extrinsics == p_b.extrinsics).all() and (p_a.points == p_b.points).all()


@pytest.mark.parametrize("world", [2, 3])
def test_one_sweep_product_on_sharded_handles(world):
    """Every rank of a sharded solve builds its own units / items / slots from its shard of the slices (the engine
    switches the one-sweep product on per rank, from 350 k observations): `world` handles in threads on the one
    device, the hook summing their buffers where RCCL would -- each rank ends where the single-rank solve ends."""
    import threading

    import torch

    from test_gpu_sharded import EmulatedAllReduce
    torch.cuda.init()
    prob, kw = CASES["heavy_tail"]()
    opts = abi.default_options(linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_IMPLICIT, max_num_iterations=5,
                               use_inner_iterations=0, **kw)
    st1, s1, _ = solve(prob, True, max_num_iterations=5, use_inner_iterations=0, **kw)
    assert st1 == 0
    os.environ["TMI_BA_MF_ONE_SWEEP"] = "1"
    try:
        emu = EmulatedAllReduce(world)
        solvers = []
        for r in range(world):
            sv = lib.Solver(prob.copy(), opts, rank=r, world=world)
            sv.set_allreduce(emu.hook(r))
            solvers.append(sv)
    finally:
        os.environ.pop("TMI_BA_MF_ONE_SWEEP", None)
    results = [None] * world

    def run(r):
        results[r] = solvers[r].solve(opts)

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert all(not t.is_alive() for t in threads) and emu.calls > 0
    for st_r, s_r in results:
        assert st_r == 0 and s_r.num_iterations == s1.num_iterations
        assert s_r.num_linear_solver_iterations == s1.num_linear_solver_iterations
        assert abs(s_r.final_cost - s1.final_cost) <= 1e-9 * s1.final_cost
        assert s_r.final_cost == results[0][1].final_cost
    for sv in solvers:
        sv.close()
