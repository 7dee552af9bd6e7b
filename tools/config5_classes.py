"""GPU: per-kernel-class times of BASELINE config 5 at Venice size (mixed camera models, shared intrinsics groups, fp32
residual evaluation) on one GPU.  usage: python tools/config5_classes.py [lo hi | size] [precision] [schur_mode]"""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
from theiasfm_amd import abi, lib, synth

bits = abi.INTRINSICS_FOCAL_LENGTH | abi.INTRINSICS_PRINCIPAL_POINTS | abi.INTRINSICS_RADIAL_DISTORTION | abi.INTRINSICS_TANGENTIAL_DISTORTION
a = [int(x) for x in sys.argv[1:]]
kw = dict(shared_group_sizes=(a[0], a[1])) if len(a) >= 2 and a[1] > 0 else dict(shared_group_size=(a[0] if a else 8))
prec = a[2] if len(a) > 2 else 32
pre = a[4] if len(a) > 4 else 2
mode = a[3] if len(a) > 3 else 0
t0 = time.time()
P = synth.config5(**kw) if "shared_group_sizes" in kw else synth.config("venice1778", models=[(abi.PINHOLE, 0.5), (abi.PINHOLE_RADIAL_TANGENTIAL, 0.25), (abi.FISHEYE, 0.25)],
                 intrinsics_to_optimize=bits, **kw)
t_gen = time.time() - t0
o = dict(point_dof=3, linear_solver_type=abi.ITERATIVE_SCHUR, residual_precision=prec, use_inner_iterations=0, schur_mode=mode, preconditioner_type=pre,
         function_tolerance=-1.0, gradient_tolerance=-1.0, parameter_tolerance=-1.0)
t0 = time.time()
s = lib.Solver(P, abi.default_options(max_num_iterations=2, **o))
t_create = time.time() - t0
s.solve(abi.default_options(max_num_iterations=2, **o))
s.reset()
st, sm = s.solve(abi.default_options(max_num_iterations=8, profile_kernels=1, **o))
rows = {n: (int(sm.kernel_launches[i]), round(1e3 * sm.kernel_seconds[i], 3)) for i, n in enumerate(abi.KERNEL_CLASS_NAMES) if sm.kernel_launches[i]}
s.reset()
st, sm2 = s.solve(abi.default_options(max_num_iterations=8, **o))
s.close()
print(json.dumps(dict(groups=P.num_groups, status=st, its=int(sm2.num_iterations), ms_per_iteration=round(1e3 * sm2.solve_time_in_seconds / max(1, sm2.num_iterations), 3),
                      pcg=int(sm2.num_linear_solver_iterations), D=int(sm2.reduced_block_dim), blocks=int(sm2.num_reduced_blocks), upper_blocks=int(sm2.num_schur_blocks),
                      pairs=int(sm2.num_schur_pairs), matrix_free_iterations=int(sm2.num_matrix_free_iterations), final_rmse=sm2.final_rmse, create_s=round(t_create, 2), gen_s=round(t_gen, 1),
                      classes_launches_ms=rows)))
