"""Probe of the sequence-structured workload (synth scene "street", venice1778_street: S a band of ~10 % fill, tens to
hundreds of PCG iterations per LM iteration) on one GPU: per-iteration PCG counts, wall time, per-class kernel time
for each operator mode.
usage: python tools/street_probe.py [steps] [case ...]"""
import json
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from theiasfm_amd import abi, lib, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
want = set(sys.argv[2:])
prob = synth.config("venice1778_street")
PB = abi.PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS
CASES = [
    ("merged_auto", 3, abi.PRECOND_SCHUR_JACOBI, 0),
    ("merged_explicit", 3, abi.PRECOND_SCHUR_JACOBI, 1),
    ("merged_implicit", 3, abi.PRECOND_SCHUR_JACOBI, 2),
    ("pb_dof4_auto", 4, PB, 0),
    ("exact_sparse_schur", 3, abi.PRECOND_SCHUR_JACOBI, -1),
]
for name, dof, pre, mode in CASES:
    if want and name not in want:
        continue
    base = dict(point_dof=dof, linear_solver_type=abi.ITERATIVE_SCHUR if mode >= 0 else abi.SPARSE_SCHUR,
                preconditioner_type=pre, schur_mode=max(mode, 0), use_inner_iterations=0)
    t0 = time.perf_counter()
    s = lib.Solver(prob.copy(), abi.default_options(max_num_iterations=2, **base), 0, 1)
    create = time.perf_counter() - t0
    s.solve(abi.default_options(max_num_iterations=2, **base))
    s.reset()
    out = {}
    for prof in (0, 1):
        o = abi.default_options(max_num_iterations=steps, profile_kernels=prof, **base)
        tr = abi.attach_trace(o, steps + 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st, sm = s.solve(o)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        s.reset()
        d = sm.as_dict()
        n = int(sm.num_iterations)
        if prof == 0:
            out = dict(case=name, create_s=round(create, 3), its=n, ms_per_iter=round(1e3 * el / max(1, n), 3), solve_ms=round(1e3 * el, 2),
                       pcg=int(sm.num_linear_solver_iterations), pcg_per_it=tr[:n, 6].astype(int).tolist(),
                       us_per_pcg_it=round(1e6 * el / max(1, int(sm.num_linear_solver_iterations)), 1),
                       matrix_free_its=int(sm.num_matrix_free_iterations), accepted=int(sm.num_successful_steps),
                       upper_blocks=int(sm.num_schur_blocks), pairs=int(sm.num_schur_pairs),
                       cost=[sm.initial_cost, sm.final_cost], rmse=[sm.initial_rmse, sm.final_rmse], status=int(st),
                       message=bytes(sm.message).split(b"\0")[0].decode())
        else:
            out["kernels_ms_per_iter"] = {k: (l, round(1e3 * sec / max(1, n), 3))
                                          for k, l, sec in zip(abi.KERNEL_CLASS_NAMES, d["kernel_launches"], d["kernel_seconds"]) if l}
    print(json.dumps(out), flush=True)
    s.close()
