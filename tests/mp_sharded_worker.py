"""Worker of tests/test_gpu_sharded.py::test_two_processes_one_gpu: one rank of a
world-size-2 sharded solve; both ranks share cuda:0, the all-reduce goes through gloo."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from theiasfm_amd import abi, dist, lib, synth  # noqa: E402


def problem(shared):
    if shared:
        # free intrinsics shared by groups of three views, mixed camera models (BASELINE config 5)
        return synth.make_problem(18, 1200, 6000, seed=71, scene="ring", spread=0.4, shared_group_size=3,
                                  models=[(abi.PINHOLE, 0.5), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.25), (abi.FISHEYE, 0.25)],
                                  intrinsics_to_optimize=abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_RADIAL_DISTORTION)
    return synth.config("ladybug49")


def options(mode, solver, inner):
    return abi.default_options(linear_solver_type=solver, point_dof=3, schur_mode=mode, use_inner_iterations=inner,
                               device=0)


def main():
    mode = int(sys.argv[1])
    solver = int(sys.argv[2]) if len(sys.argv) > 2 else abi.ITERATIVE_SCHUR
    inner = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    shared = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    rank, world, _ = dist.init_from_env(backend="gloo")
    prob = problem(shared)
    opts = options(mode, solver, inner)
    sv = lib.Solver(prob, opts, rank, world)
    sv.set_allreduce(dist.make_staged_allreduce())
    st, s = sv.solve(opts)
    out = sv.download()
    print("RESULT " + json.dumps(dict(rank=rank, status=st, cost=s.final_cost, rmse=s.final_rmse,
                                      iters=s.num_iterations, pairs=int(s.num_schur_pairs), pcg=int(s.num_linear_solver_iterations),
                                      inner_sweeps=int(s.num_inner_iteration_steps),
                                      ext=out.extrinsics.tolist(), intr=out.intrinsics.tolist(),
                                      ext0=out.extrinsics[0].tolist())), flush=True)
    sv.close()


if __name__ == "__main__":
    main()
