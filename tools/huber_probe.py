"""Venice size with the reference's application loss (HUBER, applications/build_reconstruction_flags.txt:117-121): ms per
LM iteration and per-class kernel time through the specialised + compact path, the specialised bodies with the stored
block (TMI_BA_COMPACT_ROBUST=0) and the generic bodies (TMI_BA_LINEARIZE_GENERIC=1).
usage: python tools/huber_probe.py [width]"""
import json
import os
import subprocess
import sys

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ".")
    from theiasfm_amd import abi, lib, synth
    width = float(sys.argv[2])
    prob = synth.config("venice1778_heavy")
    base = dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, use_inner_iterations=0, loss_function_type=abi.LOSS_HUBER,
                robust_loss_width=width, function_tolerance=-1.0, gradient_tolerance=-1.0, parameter_tolerance=-1.0)
    s = lib.Solver(prob.copy(), abi.default_options(max_num_iterations=2, **base), 0, 1)
    s.solve(abi.default_options(max_num_iterations=2, **base))
    s.reset()
    import time
    t0 = time.perf_counter()
    st, sm = s.solve(abi.default_options(max_num_iterations=10, **base))
    dt = time.perf_counter() - t0
    s.reset()
    st2, sm2 = s.solve(abi.default_options(max_num_iterations=10, profile_kernels=1, **base))
    d = sm2.as_dict()
    print(json.dumps(dict(status=int(st), its=int(sm.num_iterations), pcg=int(sm.num_linear_solver_iterations),
                          ms_per_iter=round(1e3 * dt / max(1, sm.num_iterations), 3), final_cost=sm.final_cost,
                          compact=s.operator_info()["compact_planes"],
                          us={k: round(1e6 * sec / max(1, l), 1) for k, l, sec in zip(abi.KERNEL_CLASS_NAMES, d["kernel_launches"], d["kernel_seconds"]) if l})))
    s.close()
    sys.exit(0)

width = sys.argv[1] if len(sys.argv) > 1 else "10.0"
for tag, env in (("specialised + compact", {}), ("specialised, stored block", {"TMI_BA_COMPACT_ROBUST": "0"}),
                 ("generic bodies", {"TMI_BA_LINEARIZE_GENERIC": "1"})):
    p = subprocess.run([sys.executable, __file__, "--child", width], capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
    print(tag, (p.stdout.strip().splitlines() or [p.stderr[-300:]])[-1], flush=True)
