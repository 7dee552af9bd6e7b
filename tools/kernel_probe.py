"""Per-class kernel time of a few LM iterations on the bench workload, whatever the solve's outcome (timing
experiments whose numerics are deliberately broken still print).
usage: python tools/kernel_probe.py [workload] [steps]"""
import json
import sys

sys.path.insert(0, ".")
from theiasfm_amd import abi, lib, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "venice1778_heavy"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
prob = synth.config(wl)
base = dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, use_inner_iterations=0, max_linear_solver_iterations=20)
s = lib.Solver(prob.copy(), abi.default_options(max_num_iterations=2, **base), 0, 1)
s.solve(abi.default_options(max_num_iterations=2, **base))
s.reset()
o = abi.default_options(max_num_iterations=steps, profile_kernels=1, function_tolerance=0.0, gradient_tolerance=0.0,
                        parameter_tolerance=0.0, **base)
st, sm = s.solve(o)
d = sm.as_dict()
print(json.dumps({"status": int(st), "its": int(sm.num_iterations), "pcg": int(sm.num_linear_solver_iterations),
                  "us": {k: round(1e6 * sec / max(1, l), 1) for k, l, sec in zip(abi.KERNEL_CLASS_NAMES, d["kernel_launches"], d["kernel_seconds"]) if l}}))
s.close()
