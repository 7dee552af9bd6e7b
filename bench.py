#!/usr/bin/env python3
"""Benchmark of the MI355X bundle-adjustment engine (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--workload venice1778]

A "step" is one Levenberg-Marquardt iteration (linearize, eliminate tracks,
build + solve the reduced camera system, back-substitute, trial cost) of the
synthetic BAL-Venice-1778-sized problem (1778 cameras / 993 923 tracks /
5 001 946 observations, fp64, 9-dof cameras, 3-dof points, ITERATIVE_SCHUR as
the reference's own policy picks for >= 1000 views,
reconstruction_estimator_utils.cc:121-125).  The timed region is ONE
tmi_ba_solver_solve call running exactly K iterations (tolerances zeroed) on
inputs already resident in HBM; W warm-up iterations run first and the
parameters are reset.  value = N_obs * K / wall time = observations/s over all
ranks (strong scaling: the problem is fixed, tracks are sharded over ranks).

For N > 1 launch with
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(cls: str, n_obs: int, n_cam: int, n_pts: int, dc: int, dp: int, nnzb: int):
    """Algorithmic HBM bytes of ONE launch of a kernel class, SURVEY 8(d):
    observations stream 24 B (u, v, 2 x int32), parameters 8 B per scalar,
    normal-equation blocks once, 8 d_c^2 B per structurally non-zero upper block
    of the reduced camera matrix.  Intermediates (Jacobians, W, Y) are not
    algorithmic."""
    sym = lambda d: d * (d + 1) // 2  # noqa: E731
    if cls == "linearize":
        return n_obs * 24 + n_cam * 8 * dc + n_pts * 8 * dp
    if cls == "point_eliminate":
        return n_pts * 8 * (sym(dp) + dp)
    if cls == "camera_diag":
        return n_cam * 8 * (sym(dc) + dc)
    if cls == "schur_offdiag":
        return 2 * nnzb * 8 * dc * dc
    if cls == "spmv":
        return nnzb * 8 * dc * dc + 2 * n_cam * 8 * dc
    if cls == "pcg_vector":
        return 4 * n_cam * 8 * dc
    if cls == "back_substitute":
        return n_obs * 24 + n_cam * 8 * dc + n_pts * 8 * (sym(dp) + 2 * dp)
    if cls == "update_cost":
        return n_obs * 24 + n_cam * 8 * dc + n_pts * 8 * dp
    return 0


def pmc_traffic(kernel_class: str, workload: str, world: int):
    """HBM bytes per launch of the kernel class from the committed rocprofv3 PMC passes
    (profiles/pmc_latest.json, written by tools/summarize_profile.py from separate
    --pmc FETCH_SIZE / --pmc WRITE_SIZE runs of this same command).  Correction per
    MI355X_MICROARCH.md (HBM section): both counters are in KiB and FETCH_SIZE reports
    half of a wide read stream on gfx950, so traffic = 2 * FETCH_SIZE + WRITE_SIZE.
    None when no matching profile is committed (other workload / GPU count)."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    if d.get("workload") != workload or world != 1:
        return None
    v = d.get("classes", {}).get(kernel_class)
    return None if v is None else int(v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="venice1778",
                    choices=["tiny", "ladybug49", "alamo", "venice1778", "venice1778_heavy"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=2)
    ap.add_argument("--transport", default="rccl", choices=["rccl", "torch", "staged"],
                    help="rccl: ncclAllReduce issued by the engine; torch: torch.distributed (nccl) hook; staged: "
                         "gloo through host memory with every rank on GPU 0 -- a functional check of the "
                         "multi-process path on a single-GPU box, not a measurement")
    ap.add_argument("--residual-precision", type=int, default=64, choices=[64, 32],
                    help="32 = evaluate residuals/Jacobians in fp32, accumulate in fp64 (config 5)")
    ap.add_argument("--schur-mode", default="auto", choices=["auto", "explicit", "implicit"],
                    help="ITERATIVE_SCHUR: form S explicitly (one all-reduce of S per LM iteration) "
                         "or apply it implicitly (one small all-reduce per PCG iteration); "
                         "auto = explicit on one GPU, implicit on several")
    args = ap.parse_args()

    import numpy as np
    import torch

    from theiasfm_amd import abi, dist, lib, synth
    import __graft_entry__ as entry

    rank, world, local = dist.init_from_env(backend="gloo" if args.transport == "staged" else None)
    if args.transport == "staged":
        local = 0
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                  "python -m torch.distributed.run --nproc-per-node N", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    if rank == 0:
        entry.build_engine()
    if world > 1:
        torch.distributed.barrier()

    t0 = time.perf_counter()
    prob = synth.config(args.workload)
    t_gen = time.perf_counter() - t0
    n_obs, n_cam, n_pts = prob.num_observations, prob.num_cameras, prob.num_points
    # the reference's solver-type policy (reconstruction_estimator_utils.cc:110-133)
    if n_cam >= 1000:
        solver_type, solver_name = abi.ITERATIVE_SCHUR, "ITERATIVE_SCHUR/SCHUR_JACOBI"
    elif n_cam >= 150:
        solver_type, solver_name = abi.SPARSE_SCHUR, "SPARSE_SCHUR (exact, dense Cholesky of S)"
    else:
        solver_type, solver_name = abi.DENSE_SCHUR, "DENSE_SCHUR (exact)"
    schur_mode = {"auto": 0, "explicit": 1, "implicit": 2}[args.schur_mode]
    # use_inner_iterations = 0: a step is the trust-region iteration proper (the coordinate-descent
    # sweep Ceres can add after each step is measured by the parity tests, not here), on the
    # device and in the CPU baseline alike
    base = dict(point_dof=3, linear_solver_type=solver_type, function_tolerance=0.0,
                gradient_tolerance=0.0, parameter_tolerance=0.0, device=local, schur_mode=schur_mode,
                residual_precision=args.residual_precision, use_inner_iterations=0)
    opts = abi.default_options(max_num_iterations=max(args.warmup, 1), **base)
    prob0 = prob.copy() if (world == 1 and not args.no_cpu_baseline) else None  # Solver.download() writes into `prob`
    t0 = time.perf_counter()
    solver = lib.Solver(prob, opts, rank, world)
    transport = "none"
    if world > 1:
        # RCCL over xGMI: natively from the engine (ncclAllReduce on its own stream); the
        # torch.distributed hook is the fallback (and --transport torch forces it)
        if args.transport == "staged":
            solver.set_allreduce(dist.make_staged_allreduce())
            transport = "gloo, staged through host memory (functional check only)"
        elif args.transport == "rccl" and dist.init_native_rccl(solver, rank, world):
            transport = "rccl (native, ncclAllReduce from the engine)"
        else:
            solver.set_allreduce(dist.make_device_allreduce())
            transport = "rccl via torch.distributed hook"
    t_create = time.perf_counter() - t0

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    # Warm-up (W iterations), then an untimed pass of the same K iterations with every kernel
    # class timed (HIP events on the engine's stream): it gives the per-class table and tells
    # which class dominates.  The timed region carries events for THAT class only (two event
    # records per launch of every class cost a few % of an iteration).
    if args.warmup > 0:
        opts_w = abi.default_options(max_num_iterations=args.warmup, **base)
        st, s = solver.solve(opts_w)
        if st != 0:
            raise RuntimeError(f"warm-up solve failed: {st} {s.message!r}")
        solver.reset()
    opts_p = abi.default_options(max_num_iterations=args.steps, profile_kernels=1, **base)
    st_p, s_p = solver.solve(opts_p)
    if st_p != 0:
        raise RuntimeError(f"profiling pass failed: {st_p} {s_p.message!r}")
    d_p = s_p.as_dict()
    secs = list(s_p.kernel_seconds)
    secs[abi.KERNEL_CLASS_NAMES.index("allreduce")] = 0.0
    dom_idx = max(range(len(secs)), key=lambda i: secs[i])
    solver.reset()

    opts_t = abi.default_options(max_num_iterations=args.steps,
                                 profile_kernels=(1 << dom_idx) if dom_idx > 0 else 1, **base)
    sync_all()
    t0 = time.perf_counter()
    st, s = solver.solve(opts_t)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        torch.distributed.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.transport == "staged" else "cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    if st != 0:
        raise RuntimeError(f"timed solve failed: {st} {s.message!r}")
    steps_run = int(s.num_iterations)
    d = s.as_dict()

    if rank != 0:
        solver.close()
        return

    dc, dp = int(s.reduced_block_dim), 3
    nnzb = int(s.num_schur_blocks)
    def table(dd):
        rows = []
        for name, launches, sec in zip(abi.KERNEL_CLASS_NAMES, dd["kernel_launches"], dd["kernel_seconds"]):
            if launches == 0 or sec <= 0.0:
                continue
            # per-rank launch: this rank's share of the observations / tracks
            ab = algorithmic_bytes(name, n_obs // world, n_cam, n_pts // world, dc, dp, nnzb)
            avg = sec / launches
            rows.append(dict(kernel=name, launches=int(launches), total_ms=round(sec * 1e3, 4),
                             avg_us=round(avg * 1e6, 2), algorithmic_bytes_per_launch=int(ab),
                             achieved_GBs=round(ab / avg / 1e9, 2) if avg > 0 else None))
        return rows

    kernels = table(d_p)
    dom_name = abi.KERNEL_CLASS_NAMES[dom_idx]
    timed_rows = [k for k in table(d) if k["kernel"] == dom_name]
    dom = timed_rows[0] if timed_rows else max((k for k in kernels if k["kernel"] != "allreduce"),
                                               key=lambda k: k["total_ms"])
    roofline = dict(bound="hbm", kernel=dom["kernel"], achieved=dom["achieved_GBs"], peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(dom["achieved_GBs"] / HBM_PEAK_GBS, 5),
                    traffic=pmc_traffic(dom["kernel"], args.workload, world),
                    launches=dom["launches"], avg_us=dom["avg_us"],
                    algorithmic_bytes_per_launch=dom["algorithmic_bytes_per_launch"],
                    measured="HIP events on the engine's stream inside the timed region")

    out = dict(
        metric="ba_observations_per_sec", value=n_obs * steps_run / elapsed, unit="observations/s",
        n_gpus=world, steps=steps_run, warmup=args.warmup, ms_per_step=1e3 * elapsed / max(steps_run, 1),
        higher_is_better=True, scaling="strong", vs_baseline=None,
        dtype="f64" if args.residual_precision == 64 else "f32 residuals/Jacobians, f64 accumulation",
        data="synthetic",
        config=dict(workload=f"{args.workload}-synthetic", cameras=n_cam, tracks=n_pts,
                    observations=n_obs, camera_dof=dc, point_dof=dp, linear_solver=solver_name,
                    loss="TRIVIAL",
                    schur_operator=("implicit (matrix-free)" if int(s.num_schur_pairs) == 0 and
                                    solver_type == abi.ITERATIVE_SCHUR else "explicit block-sparse S"),
                    parallelism=f"tracks sharded x{world}", transport=transport),
        lm_iterations_per_sec=steps_run / elapsed,
        pcg_iterations=int(s.num_linear_solver_iterations),
        initial_cost=s.initial_cost, final_cost=s.final_cost, initial_rmse=s.initial_rmse,
        final_rmse=s.final_rmse, accepted_steps=int(s.num_successful_steps),
        schur_blocks_upper=nnzb, schur_pairs=int(s.num_schur_pairs),
        setup_seconds=dict(generate=round(t_gen, 3), create_upload=round(t_create, 3)),
        roofline=roofline, kernels=kernels,
        kernels_note="per-class table: separate untimed pass of the same iterations with every class timed")
    if steps_run != args.steps:
        out["note"] = f"solver stopped after {steps_run} of {args.steps} iterations: {d['message']}"
    # Whole-iteration figure of SURVEY 8(d): B_iter = B_lin + B_schur + B_pcg + B_back + B_cost with the
    # measured block count and PCG iterations (intermediates -- Jacobians, Y -- are not algorithmic)
    sym = lambda n: n * (n + 1) // 2  # noqa: E731
    n_pcg = int(s.num_linear_solver_iterations) / max(steps_run, 1)
    b_lin = n_obs * 24 + n_cam * 8 * dc + n_pts * 8 * dp + n_cam * 8 * (sym(dc) + dc) + n_pts * 8 * (sym(dp) + dp)
    b_schur = 2 * nnzb * 8 * dc * dc
    b_pcg = n_pcg * (nnzb * 8 * dc * dc + 6 * n_cam * 8 * dc)
    b_back = n_obs * 24 + n_cam * 8 * dc + n_pts * 8 * (sym(dp) + 2 * dp)
    b_cost = n_obs * 24 + n_cam * 8 * dc + n_pts * 8 * dp
    b_iter = b_lin + b_schur + b_pcg + b_back + b_cost
    out["iteration_roofline"] = dict(
        algorithmic_bytes_per_lm_iteration=int(b_iter), pcg_iterations_per_lm_iteration=round(n_pcg, 2),
        achieved_GBs=round(b_iter * steps_run / elapsed / 1e9, 1), peak_GBs=HBM_PEAK_GBS * world,
        frac=round(b_iter * steps_run / elapsed / 1e9 / (HBM_PEAK_GBS * world), 5),
        note="SURVEY 8(d) formula; the kernels move ~19 GB per iteration (profiles/), ~4x these bytes, because "
             "Jacobian blocks and the per-observation Schur factors are stored and gathered rather than recomputed")

    if world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the in-repo oracle (Ceres-semantics restatement; the real
        # Theia + Ceres cannot be built offline) on the SAME problem and options
        # for a bounded number of LM iterations, all host cores via OpenMP.
        from oracle import oracle
        iters = max(1, args.cpu_iters)
        cpu_opts = abi.default_options(max_num_iterations=iters, **{**base, "device": -1})
        ref = prob0.copy()
        tc = time.perf_counter()
        st_o, s_o = oracle.solve(ref, cpu_opts)
        t_cpu = time.perf_counter() - tc
        # the device on the same bounded sample, for a full-size parity figure
        solver.reset()
        dev_opts = abi.default_options(max_num_iterations=iters, **base)
        st_d, s_d = solver.solve(dev_opts)
        out["cpu_baseline"] = dict(
            value=n_obs * int(s_o.num_iterations) / s_o.solve_time_in_seconds, unit="observations/s",
            cores=oracle.num_threads(), kind="port",
            sample=f"{int(s_o.num_iterations)} LM iterations of the full {args.workload} problem, same options "
                   f"(solve {s_o.solve_time_in_seconds:.2f} s + setup {s_o.setup_time_in_seconds:.2f} s, wall {t_cpu:.2f} s)",
            final_cost=s_o.final_cost, final_rmse=s_o.final_rmse)
        out["parity_sample"] = dict(
            iterations=int(s_o.num_iterations), device_cost=s_d.final_cost, oracle_cost=s_o.final_cost,
            rel_cost_diff=abs(s_d.final_cost - s_o.final_cost) / s_o.final_cost,
            rmse_abs_diff=abs(s_d.final_rmse - s_o.final_rmse))
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        # The steps either side of the BA on the same resident
        # problem, outside the timed region: kernel time from HIP events, the oracle beside it.
        side = {}
        adjusted = solver.download().copy()
        flag_d, _, fs = solver.filter_outlier_tracks(4.0, 2.0)
        flag_d, _, fs = solver.filter_outlier_tracks(4.0, 2.0)  # second launch: warm
        tc = time.perf_counter()
        flag_o, _, counts = oracle.filter_outlier_tracks(adjusted, 4.0, 2.0)
        t_f = time.perf_counter() - tc
        side["outlier_filter"] = dict(
            kernel_us=round(fs.kernel_seconds * 1e6, 1), call_ms=round(fs.seconds * 1e3, 3),
            observations_per_s=n_obs / fs.kernel_seconds, cpu_port_observations_per_s=n_obs / t_f,
            flags_equal=bool((flag_d == flag_o).all()), removed=int(counts[1] + counts[2]))
        solver.reset()
        trk_opts = abi.default_options(max_num_iterations=10, **{**base, "function_tolerance": 1e-6,
                                                               "parameter_tolerance": 1e-8,
                                                               "gradient_tolerance": 1e-10})
        term_d, it_d, _, c1_d, ts = solver.adjust_tracks(trk_opts)
        # CPU port on a bounded sample: the first 100 000 tracks (the rest are marked constant,
        # which the per-track oracle skips)
        n_cpu = min(100_000, n_pts)
        ref2 = prob0.copy()
        ref2.point_constant = ref2.point_constant.copy()
        ref2.point_constant[n_cpu:] = 1
        tc = time.perf_counter()
        term_o, it_o, _, c1_o = oracle.adjust_tracks(ref2, trk_opts)
        t_t = time.perf_counter() - tc
        sm = slice(0, n_cpu)
        side["batched_track_ba"] = dict(
            kernel_ms=round(ts.kernel_seconds * 1e3, 3), tracks=int(ts.num_tracks),
            lm_iterations=int(ts.total_iterations), tracks_per_s=ts.num_tracks / ts.kernel_seconds,
            cpu_port_tracks_per_s=n_cpu / t_t, cpu_port_sample=f"first {n_cpu} tracks, {oracle.num_threads()} threads",
            termination_mismatches=int((term_d[sm] != term_o[sm]).sum()),
            iteration_mismatches=int((it_d[sm] != it_o[sm]).sum()),
            final_cost_rel_diff_above_1e9=int((np.abs(c1_d[sm] - c1_o[sm]) > 1e-9 * np.maximum(c1_o[sm], 1e-12)).sum()),
            sample_final_cost=dict(device=float(c1_d[sm].sum()), oracle=float(c1_o[sm].sum())))
        solver.reset()
        sel_d, ln_d, err_d, ss = solver.select_good_tracks(10, 100, 100)
        tc = time.perf_counter()
        sel_o, _, _ = oracle.select_good_tracks(prob0, 10, 100, 100)
        t_s = time.perf_counter() - tc
        side["track_selection"] = dict(
            statistics_kernel_us=round(ss.kernel_seconds * 1e6, 1), call_ms=round(ss.seconds * 1e3, 2),
            cpu_port_ms=round(t_s * 1e3, 1), selected=int(ss.num_selected), of=int(ss.num_tracks),
            selection_equal=bool((sel_d == sel_o).all()))
        out["side_kernels"] = side
    solver.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
