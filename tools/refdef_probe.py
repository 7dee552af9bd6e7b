"""Probe of the reference-default operating point at Venice size on one GPU: Ceres-shaped SCHUR_JACOBI
(per parameter block), 4-dof homogeneous points, inner iterations on -- what theia::BundleAdjustReconstruction
passes when the caller changes nothing (bundle_adjustment.h:78-122, reconstruction_estimator_utils.cc:110-133) --
and the relaxations between it and the headline, with per-class kernel time.
usage: python tools/refdef_probe.py [steps] [case ...]"""
import json
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

from theiasfm_amd import abi, lib, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
want = set(sys.argv[2:])
prob = synth.config("venice1778_heavy")
PB = abi.PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS
CASES = [
    # name, point_dof, preconditioner, inner, schur_mode
    ("headline", 3, abi.PRECOND_SCHUR_JACOBI, 0, 0),
    ("pb_dof3_auto", 3, PB, 0, 0),
    ("pb_dof3_explicit", 3, PB, 0, 1),
    ("pb_dof3_implicit", 3, PB, 0, 2),
    ("pb_dof4_auto", 4, PB, 0, 0),
    ("pb_dof4_explicit", 4, PB, 0, 1),
    ("pb_dof4_implicit", 4, PB, 0, 2),
    ("refdef_auto", 4, PB, 1, 0),
    ("refdef_explicit", 4, PB, 1, 1),
    ("refdef_implicit", 4, PB, 1, 2),
]
for name, dof, pre, inner, mode in CASES:
    if want and name not in want:
        continue
    base = dict(point_dof=dof, linear_solver_type=abi.ITERATIVE_SCHUR, preconditioner_type=pre,
                function_tolerance=-1.0, gradient_tolerance=-1.0, parameter_tolerance=-1.0, schur_mode=mode,
                use_inner_iterations=inner)
    s = lib.Solver(prob.copy(), abi.default_options(max_num_iterations=2, **base), 0, 1)
    s.solve(abi.default_options(max_num_iterations=2, **base))
    s.reset()
    out = {}
    for prof in (0, 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, sm = s.solve(abi.default_options(max_num_iterations=steps, profile_kernels=prof, **base))
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        s.reset()
        d = sm.as_dict()
        if prof == 0:
            out = dict(case=name, its=int(sm.num_iterations), ms_per_iter=round(1e3 * el / max(1, sm.num_iterations), 3),
                       pcg=int(sm.num_linear_solver_iterations), matrix_free_its=int(sm.num_matrix_free_iterations),
                       sweeps=int(sm.num_inner_iteration_steps), accepted=int(sm.num_successful_steps),
                       final_cost=sm.final_cost, status=int(st))
        else:
            out["kernels_ms_per_iter"] = {n: (l, round(1e3 * sec / max(1, sm.num_iterations), 3))
                                          for n, l, sec in zip(abi.KERNEL_CLASS_NAMES, d["kernel_launches"], d["kernel_seconds"]) if l}
    print(json.dumps(out), flush=True)
    s.close()
