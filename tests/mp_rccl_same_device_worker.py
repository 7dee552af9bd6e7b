"""Worker of tests/test_gpu_bench_multirank.py::test_native_rccl_refuses_two_ranks_on_one_device_cleanly: two ranks on
cuda:0 try the engine's native RCCL binding (must fail cleanly: one communicator rank per device), then solve through
the host-staged gloo hook."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from theiasfm_amd import abi, dist, lib, synth  # noqa: E402


def main():
    rank, world, _ = dist.init_from_env(backend="gloo")
    prob = synth.config("ladybug49")
    opts = abi.default_options(linear_solver_type=abi.ITERATIVE_SCHUR, point_dof=3, max_num_iterations=3, device=0)
    sv = lib.Solver(prob, opts, rank, world)
    native = dist.init_native_rccl(sv, rank, world)
    if not native:
        sv.set_allreduce(dist.make_staged_allreduce())
    st, s = sv.solve(opts)
    print("RESULT " + json.dumps(dict(rank=rank, native=bool(native), status=int(st), cost=s.final_cost)), flush=True)
    sv.close()


if __name__ == "__main__":
    main()
