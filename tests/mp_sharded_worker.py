"""Worker of tests/test_gpu_sharded.py::test_two_processes_one_gpu: one rank of a
world-size-2 sharded solve; both ranks share cuda:0, the all-reduce goes through gloo."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from theiasfm_amd import abi, dist, lib, synth  # noqa: E402


def main():
    mode = int(sys.argv[1])
    rank, world, _ = dist.init_from_env(backend="gloo")
    prob = synth.config("ladybug49")
    opts = abi.default_options(linear_solver_type=abi.ITERATIVE_SCHUR, point_dof=3, schur_mode=mode,
                               use_inner_iterations=0,
                               device=0)
    sv = lib.Solver(prob, opts, rank, world)
    sv.set_allreduce(dist.make_staged_allreduce())
    st, s = sv.solve(opts)
    out = sv.download()
    print("RESULT " + json.dumps(dict(rank=rank, status=st, cost=s.final_cost, rmse=s.final_rmse,
                                      iters=s.num_iterations, pairs=int(s.num_schur_pairs),
                                      ext0=out.extrinsics[0].tolist())), flush=True)
    sv.close()


if __name__ == "__main__":
    main()
