#!/usr/bin/env python3
"""Bundle-adjust a Bundle-Adjustment-in-the-Large problem file on the MI355X engine.

  python tools/ba_bal.py problem-1778-993923-pre.txt.bz2 [--out adjusted.txt] [--iterations 50]
                         [--solver auto|dense|sparse|iterative] [--loss trivial|huber|cauchy] [--width 2.0]
                         [--no-inner-iterations] [--filter MAX_ERR MIN_ANGLE]

The file is converted to the reference's conventions exactly as its Bundler importer does
(theiasfm_amd/io.py:read_bal); the solver type defaults to the reference's own policy
(reconstruction_estimator_utils.cc:110-133).  Needs a GPU: there is no CPU fallback."""
import argparse
import sys
import time

sys.path.insert(0, __file__.rsplit("/", 2)[0])

from theiasfm_amd import abi, io, lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("bal")
    ap.add_argument("--out")
    ap.add_argument("--iterations", type=int, default=50)
    ap.add_argument("--solver", default="auto", choices=["auto", "dense", "sparse", "iterative"])
    ap.add_argument("--loss", default="trivial", choices=["trivial", "huber", "softlone", "cauchy", "arctan", "tukey"])
    ap.add_argument("--width", type=float, default=2.0)
    ap.add_argument("--no-inner-iterations", action="store_true")
    ap.add_argument("--filter", nargs=2, type=float, metavar=("MAX_ERR_PX", "MIN_ANGLE_DEG"),
                    help="after BA, flag outlier tracks (SetOutlierTracksToUnestimated) and adjust once more without them")
    args = ap.parse_args()

    t0 = time.perf_counter()
    prob = io.read_bal(args.bal)
    print(f"read {prob.num_cameras} cameras, {prob.num_points} points, {prob.num_observations} observations "
          f"in {time.perf_counter() - t0:.1f} s")
    if args.solver == "auto":
        solver = (abi.ITERATIVE_SCHUR if prob.num_cameras >= 1000 else
                  abi.SPARSE_SCHUR if prob.num_cameras >= 150 else abi.DENSE_SCHUR)
    else:
        solver = {"dense": abi.DENSE_SCHUR, "sparse": abi.SPARSE_SCHUR, "iterative": abi.ITERATIVE_SCHUR}[args.solver]
    loss = ["trivial", "huber", "softlone", "cauchy", "arctan", "tukey"].index(args.loss)
    opts = abi.default_options(point_dof=3, linear_solver_type=solver, max_num_iterations=args.iterations,
                               loss_function_type=loss, robust_loss_width=args.width,
                               use_inner_iterations=0 if args.no_inner_iterations else 1, verbose=1)

    def report(tag, s):
        print(f"{tag}: {s.message.decode()} | {s.num_iterations} LM iterations ({s.num_successful_steps} accepted, "
              f"{s.num_inner_iteration_steps} with an inner sweep), {s.num_linear_solver_iterations} PCG | "
              f"cost {s.initial_cost:.6e} -> {s.final_cost:.6e} | RMSE {s.initial_rmse:.4f} -> {s.final_rmse:.4f} px | "
              f"setup {s.setup_time_in_seconds:.2f} s, solve {s.solve_time_in_seconds:.3f} s")

    sv = lib.Solver(prob, opts)
    st, s = sv.solve(opts)
    report("BA", s)
    if st != 0:
        sys.exit(1)
    if args.filter:
        flag, mean, fs = sv.filter_outlier_tracks(args.filter[0], args.filter[1])
        print(f"filter: {fs.num_bad_reprojections} tracks with bad reprojections, "
              f"{fs.num_insufficient_viewing_angles} with insufficient viewing angles "
              f"of {fs.num_estimated_tracks} ({fs.kernel_seconds * 1e6:.0f} us)")
        adjusted = sv.download()
        sv.close()
        keep = flag[adjusted.obs_point] == 0
        adjusted.obs_camera, adjusted.obs_point, adjusted.obs_xy = (
            adjusted.obs_camera[keep], adjusted.obs_point[keep], adjusted.obs_xy[keep])
        st, s = lib.solve(adjusted, opts)
        report("BA without the flagged tracks", s)
        prob = adjusted
    else:
        prob = sv.download()
        sv.close()
    if args.out:
        io.write_bal(args.out, prob)
        print("wrote", args.out)


if __name__ == "__main__":
    main()
