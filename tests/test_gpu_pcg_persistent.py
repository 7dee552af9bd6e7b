"""The persistent PCG launch (pcg_persist.h: a whole PCG solve on the formed S in one launch, grid barriers instead of
kernel boundaries) against the launch-per-step loop it replaces (TMI_BA_PCG_PERSISTENT=0) and against the oracle:
same PCG iteration counts in every LM iteration, same trajectory, bit-reproducible from run to run."""
import os

import numpy as np
import pytest

from oracle import oracle
from theiasfm_amd import abi, synth

pytestmark = pytest.mark.gpu


def run(prob, persistent, **kw):
    from theiasfm_amd import lib
    o = abi.default_options(**{**dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, schur_mode=abi.SCHUR_EXPLICIT,
                                      use_inner_iterations=0), **kw})
    trace = abi.attach_trace(o, o.max_num_iterations + 1)
    old = os.environ.get("TMI_BA_PCG_PERSISTENT")
    os.environ["TMI_BA_PCG_PERSISTENT"] = "1" if persistent else "0"
    try:
        p = prob.copy()
        st, s = lib.solve(p, o)
    finally:
        if old is None:
            del os.environ["TMI_BA_PCG_PERSISTENT"]
        else:
            os.environ["TMI_BA_PCG_PERSISTENT"] = old
    assert st == 0, bytes(s.message)
    return p, s, trace[:s.num_iterations].copy()


CASES = {
    # long PCG solves (sequence structure): residual resets every 10th iteration, hundreds of iterations
    "street": (lambda: synth.make_problem(300, 60000, 300000, seed=5, scene="street", spread=0.025, heavy_tail=0.002),
               dict(max_num_iterations=7)),
    "street_parameter_blocks_dof4": (lambda: synth.make_problem(120, 20000, 100000, seed=6, scene="street", spread=0.03),
                                     dict(max_num_iterations=6, preconditioner_type=abi.PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS, point_dof=4)),
    "ring_huber": (lambda: synth.config("ladybug49"), dict(max_num_iterations=8, loss_function_type=abi.LOSS_HUBER, robust_loss_width=2.0)),
    "ring_16wide": (lambda: synth.make_problem(40, 6000, 30000, seed=3, scene="ring", spread=0.4, models=[(abi.PINHOLE_RADIAL_TANGENTIAL, 1.0)],
                                               intrinsics_to_optimize=abi.INTRINSICS_ALL), dict(max_num_iterations=6)),
    "capped": (lambda: synth.make_problem(300, 60000, 300000, seed=5, scene="street", spread=0.025),
               dict(max_num_iterations=6, max_linear_solver_iterations=25, min_linear_solver_iterations=3)),
}


@pytest.mark.parametrize("name", list(CASES))
def test_persistent_pcg_equals_the_launch_per_step_loop_and_the_oracle(name):
    make, kw = CASES[name]
    prob = make()
    pa, sa, ta = run(prob, True, **kw)
    pb, sb, tb = run(prob, False, **kw)
    assert sa.num_iterations == sb.num_iterations and sa.num_successful_steps == sb.num_successful_steps
    assert np.array_equal(ta[:, 3], tb[:, 3])                       # accepted / rejected sequence
    assert np.array_equal(ta[:, 6], tb[:, 6]), (ta[:, 6], tb[:, 6])  # PCG iterations of every LM iteration
    assert abs(sa.final_cost - sb.final_cost) <= 1e-9 * sb.final_cost
    assert np.abs(pa.extrinsics - pb.extrinsics).max() <= 1e-6 * max(1.0, np.abs(pb.extrinsics).max())
    # bit-reproducible from run to run
    pc, sc, tc = run(prob, True, **kw)
    assert sc.final_cost == sa.final_cost and np.array_equal(pc.extrinsics, pa.extrinsics) and np.array_equal(pc.points, pa.points)
    # the oracle
    o = abi.default_options(point_dof=kw.get("point_dof", 3), linear_solver_type=abi.ITERATIVE_SCHUR, use_inner_iterations=0,
                            **{k: v for k, v in kw.items() if k != "point_dof"})
    to = abi.attach_trace(o, o.max_num_iterations + 1)
    po = prob.copy()
    st, so = oracle.solve(po, o)
    assert st == 0 and so.num_iterations == sa.num_iterations
    assert np.array_equal(to[:so.num_iterations, 3], ta[:, 3])
    assert np.abs(to[:so.num_iterations, 6] - ta[:, 6]).max() <= 1     # a Q-tolerance decision on the last bit may differ
    assert abs(so.final_cost - sa.final_cost) <= 1e-9 * so.final_cost
