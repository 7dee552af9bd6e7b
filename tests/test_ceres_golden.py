"""Consumes real Ceres output when it exists (tools/make_ceres_golden.md): the LM / Schur / PCG layer
against TheiaSfM's own BundleAdjustReconstruction, produced off-box.  Until someone runs the recipe
the golden files are absent and these tests skip; the input side (archive written from scratch,
readable, equal to the synthetic problem) is checked here and now."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import oracle
from theiasfm_amd import abi, io, synth

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "ceres")
CASES = sorted(glob.glob(os.path.join(GOLD, "*_ceres.json")))
RECIPE = "no Ceres golden committed yet: run tools/make_ceres_golden.md on a machine with TheiaSfM"
SOLVERS = {"DENSE_SCHUR": abi.DENSE_SCHUR, "SPARSE_SCHUR": abi.SPARSE_SCHUR, "ITERATIVE_SCHUR": abi.ITERATIVE_SCHUR}


def test_committed_input_archive_is_the_synthetic_problem():
    rec = io.read_theia_reconstruction(os.path.join(GOLD, "tiny_input.bin"))
    P = io.flatten_reconstruction(rec)
    Q = synth.config("tiny")
    np.testing.assert_array_equal(P.extrinsics, Q.extrinsics)
    np.testing.assert_array_equal(P.intrinsics, Q.intrinsics)
    np.testing.assert_array_equal(P.points, Q.points)
    assert P.num_observations == Q.num_observations
    assert rec.versions["Camera"] == 1 and rec.versions["CameraIntrinsicsPrior"] == 4


def _load(case):
    meta = json.load(open(case))
    name = os.path.basename(case).split("_")[0]
    prob = synth.config(name)
    out = io.flatten_reconstruction(io.read_theia_reconstruction(case[:-5] + ".bin"))
    opts = dict(linear_solver_type=SOLVERS[meta["linear_solver"]], point_dof=4,
                use_inner_iterations=int(meta["use_inner_iterations"]),
                max_num_iterations=int(meta["max_num_iterations"]))
    return meta, prob, out, opts


def _compare(meta, solved, summary, ceres_out):
    assert summary.success == 1 and meta["success"] == 1
    assert abs(summary.initial_cost - meta["initial_cost"]) <= 1e-9 * meta["initial_cost"]
    assert abs(summary.final_cost - meta["final_cost"]) <= 1e-6 * meta["final_cost"]
    c_ref, rmse_ref, _ = oracle.cost(ceres_out)
    assert abs(summary.final_rmse - rmse_ref) <= 1e-6
    scale = np.abs(ceres_out.extrinsics[:, :3]).max()
    assert np.abs(solved.extrinsics[:, :3] - ceres_out.extrinsics[:, :3]).max() <= 1e-6 * scale


@pytest.mark.skipif(not CASES, reason=RECIPE)
@pytest.mark.parametrize("case", CASES or ["-"])
def test_oracle_reaches_the_minimum_ceres_found(case):
    meta, prob, ceres_out, opts = _load(case)
    st, s = oracle.solve(prob, abi.default_options(**opts))
    assert st == 0
    _compare(meta, prob, s, ceres_out)


@pytest.mark.gpu
@pytest.mark.skipif(not CASES, reason=RECIPE)
@pytest.mark.parametrize("case", CASES or ["-"])
def test_device_reaches_the_minimum_ceres_found(case):
    from theiasfm_amd import lib
    meta, prob, ceres_out, opts = _load(case)
    st, s = lib.solve(prob, abi.default_options(**opts))
    assert st == 0
    _compare(meta, prob, s, ceres_out)
