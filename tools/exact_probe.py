"""The exact reduced solve (SPARSE_SCHUR / DENSE_SCHUR: dense Cholesky of S on the device) beyond the reference's
1000-view policy limit: Venice-sized problem, n = 9 x 1778 = 16 002 unknowns.  Prints one JSON line."""
import json
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402,F401

from theiasfm_amd import abi, lib, synth  # noqa: E402

prob = synth.config(sys.argv[1] if len(sys.argv) > 1 else "venice1778_heavy")
base = dict(point_dof=3, linear_solver_type=abi.SPARSE_SCHUR, function_tolerance=-1.0, gradient_tolerance=-1.0,
            parameter_tolerance=-1.0, use_inner_iterations=0)
s = lib.Solver(prob, abi.default_options(max_num_iterations=1, **base))
s.solve(abi.default_options(max_num_iterations=1, **base))
s.reset()
t0 = time.perf_counter()
st, sm = s.solve(abi.default_options(max_num_iterations=4, profile_kernels=1, **base))
el = time.perf_counter() - t0
d = sm.as_dict()
ks = {n: (l, round(1e3 * sec / max(l, 1), 3)) for n, l, sec in zip(abi.KERNEL_CLASS_NAMES, d["kernel_launches"], d["kernel_seconds"]) if l}
print(json.dumps(dict(cameras=prob.num_cameras, unknowns=int(sm.num_reduced_blocks) * int(sm.reduced_block_dim), status=st,
                      iterations=int(sm.num_iterations), ms_per_iteration=round(1e3 * el / max(1, sm.num_iterations), 2),
                      final_cost=sm.final_cost, ms_per_launch=ks)))
s.close()
