"""Venice size with every second view constant (a partial adjustment, bundle_adjuster.cc:141-180): ms per LM iteration through
the specialised + compact path and through the generic bodies (TMI_BA_LINEARIZE_GENERIC=1).  usage: python tools/partial_probe.py"""
import json
import os
import subprocess
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ".")
    from theiasfm_amd import abi, lib, synth
    prob = synth.config("venice1778_heavy")
    for c in range(0, prob.num_cameras, 2):
        prob.camera_flags[c] = abi.CAMERA_POSITION_CONSTANT | abi.CAMERA_ORIENTATION_CONSTANT
        g = prob.camera_group[c]
        prob.intrinsics_constant[prob.group_offset[g]:prob.group_offset[g + 1]] = 1
    base = dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, use_inner_iterations=0, function_tolerance=-1.0,
                gradient_tolerance=-1.0, parameter_tolerance=-1.0)
    s = lib.Solver(prob.copy(), abi.default_options(max_num_iterations=2, **base), 0, 1)
    s.solve(abi.default_options(max_num_iterations=2, **base))
    s.reset()
    t0 = time.perf_counter()
    st, sm = s.solve(abi.default_options(max_num_iterations=10, **base))
    dt = time.perf_counter() - t0
    print(json.dumps(dict(status=int(st), its=int(sm.num_iterations), pcg=int(sm.num_linear_solver_iterations),
                          ms_per_iter=round(1e3 * dt / max(1, sm.num_iterations), 3), final_cost=sm.final_cost,
                          compact=s.operator_info()["compact_planes"])))
    s.close()
    sys.exit(0)

for tag, env in (("specialised + compact", {}), ("generic bodies", {"TMI_BA_LINEARIZE_GENERIC": "1"})):
    p = subprocess.run([sys.executable, __file__, "--child"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
    print(tag, (p.stdout.strip().splitlines() or [p.stderr[-300:]])[-1], flush=True)
