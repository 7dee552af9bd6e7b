"""Fill the device memory with NaNs (torch allocations, freed again), then solve in the SAME process: a read of memory
nobody initialised shows up as a failed or different solve.  usage: python tools/poison_probe.py [workload] [gb]"""
import sys

sys.path.insert(0, ".")
import torch  # noqa: E402

from theiasfm_amd import abi, lib, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "venice1778_heavy"
gb = float(sys.argv[2]) if len(sys.argv) > 2 else 24.0
prob = synth.config(wl)
base = dict(point_dof=int(sys.argv[3]) if len(sys.argv) > 3 else 3, linear_solver_type=abi.ITERATIVE_SCHUR, use_inner_iterations=0,
            function_tolerance=-1.0, gradient_tolerance=-1.0, parameter_tolerance=-1.0)


def solve(tag):
    o = abi.default_options(max_num_iterations=6, **base)
    s = lib.Solver(prob.copy(), o, 0, 1)
    st, sm = s.solve(o)
    print(tag, int(st), sm.message, int(sm.num_iterations), int(sm.num_linear_solver_iterations), repr(sm.final_cost), flush=True)
    s.close()
    return st, sm.final_cost


a = solve("clean ")
chunks = [torch.full((int(2 ** 27),), float("nan"), dtype=torch.float64, device="cuda") for _ in range(int(gb))]
torch.cuda.synchronize()
del chunks
torch.cuda.empty_cache()
b = solve("poison")
assert a == b, (a, b)
print("same result on poisoned memory")
