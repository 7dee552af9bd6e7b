/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 *
 * CPU restatement of the reference's bundle-adjustment path:
 *   - residual functor + 5 camera models: restated from /root/reference
 *     (citations on every function in reprojection_tmpl.h);
 *   - Jacobians by forward-mode dual numbers, as Ceres autodiff does
 *     (create_reprojection_error_cost_function.h:60-89) -- an independent
 *     check of the device path's analytic Jacobians;
 *   - the numerics behind ceres::Solve (bundle_adjuster.cc:205): Ceres Solver
 *     is an EXTERNAL dependency absent from /root/reference, un-vendored and
 *     un-pinned (CMakeLists.txt:152; API usage bounds it to 1.12 <= v < 2.2).
 *     Its published trust-region Levenberg-Marquardt / Schur / PCG algorithm
 *     is restated from the Ceres 1.14 sources' documented behaviour (SURVEY
 *     App. B).
 *
 * PINNING STATUS: the projection functions are pinned by the reference's own
 * round-trip tests (pinhole_camera_model_test.cc:218-298 and siblings,
 * restated in tests/test_oracle_camera_models.py) and by the known-answer
 * cost/RMSE of the reference fixture data/sfm/fountain11.bin
 * (tests/golden/).  The LM / Schur / PCG layer is "PARITY UNPINNED": the
 * reference holds no test that pins BundleAdjustReconstruction output
 * numerically and neither Ceres nor the reference can be built
 * here (Eigen, Ceres, glog, gflags, SuiteSparse absent).
 *
 * The problem / options / summary structs are the public ones from
 * include/theia_mi355_ba.h so the same flattened inputs feed both sides.
 */
#ifndef ORACLE_BA_ORACLE_H_
#define ORACLE_BA_ORACLE_H_

#include "../include/theia_mi355_ba.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Ceres-semantics LM on the CPU.  In/out arrays of `problem` are updated when
 * the solution is usable.  Returns a tmi_ba_status. */
int32_t oracle_ba_solve(tmi_ba_problem* problem, const tmi_ba_options* options,
                        tmi_ba_summary* summary);

/* One inner-iteration (coordinate descent) sweep at the problem's current parameters, in
 * place: extrinsics blocks, intrinsics blocks, points -- the reversed linear-solver ordering
 * the reference hands to Ceres (bundle_adjuster.cc:193-200, groups :346-371). */
int32_t oracle_inner_sweep(tmi_ba_problem* problem, const tmi_ba_options* options);
/* Test hook: 1 swaps the first two sets (intrinsics before extrinsics), 0 restores the
 * reference order.  Exists so tests can show the orders are distinguishable. */
void oracle_set_inner_order(int32_t order);

/* Per-observation evaluation with dual numbers at the problem's current
 * parameters, in the caller's observation order:
 *   residuals [2N]; jac_full [2*20*N] row-major 2x20 with columns
 *   [ext(6) | intrinsics(10, zero padded) | point(4)]; valid [N].
 * Any output may be NULL. */
int32_t oracle_ba_evaluate(const tmi_ba_problem* problem, double* residuals,
                           double* jac_full, uint8_t* valid);

/* 1/2 sum rho(|r|^2) and un-robustified RMSE at the current parameters.
 * Returns number of invalid observations (cost excludes them). */
int64_t oracle_ba_cost(const tmi_ba_problem* problem, const tmi_ba_options* options,
                       double* cost, double* rmse);

/* Camera::ProjectPoint (reference camera.cc:204-213): pixel and depth. */
double oracle_project_point(int32_t model, const double* extrinsics,
                            const double* intrinsics, const double* point4,
                            double* pixel2);

/* CameraToPixelCoordinates / PixelToCameraCoordinates of one model, used to
 * restate the reference's round-trip tests. */
void oracle_camera_to_pixel(int32_t model, const double* intrinsics,
                            const double* point3, double* pixel2);
void oracle_pixel_to_camera(int32_t model, const double* intrinsics,
                            const double* pixel2, double* point3);

void oracle_camera_to_pixel_batch(int32_t model, const double* intrinsics, const double* points3,
                                  int64_t n, double* pixels2);
void oracle_pixel_to_camera_batch(int32_t model, const double* intrinsics, const double* pixels2,
                                  int64_t n, double* points3);

/* Loss function rho[3] = {rho, rho', rho''} (ceres/loss_function.cc). */
void oracle_loss(int32_t type, double width, double s, double rho[3]);

/* GetSubsetFromOptimizeIntrinsicsType restated (same contract as
 * tmi_ba_intrinsics_constant_mask). */
int32_t oracle_intrinsics_constant_mask(int32_t model, int32_t bitmask, uint8_t* mask);

/* SufficientTriangulationAngle (triangulation.cc:236-250) over n unit rays. */
int32_t oracle_sufficient_triangulation_angle(const double* rays3, int64_t n,
                                              double min_triangulation_angle_degrees);

/* SetOutlierTracksToUnestimated (set_outlier_tracks_to_unestimated.cc:62-133) over the
 * flattened problem.  flag[num_points]: 0 kept, 1 bad reprojection / behind a camera,
 * 2 insufficient viewing angle; mean_sq_error[num_points] (may be NULL; value at the
 * point the reference's loop stops); counts = {estimated, bad reprojections,
 * insufficient angles}. */
int32_t oracle_filter_outlier_tracks(const tmi_ba_problem* problem,
                                     double max_inlier_reprojection_error,
                                     double min_triangulation_angle_degrees, uint8_t* flag,
                                     double* mean_sq_error, int64_t counts[3]);

/* theia::BundleAdjustTrack (bundle_adjustment.cc:96-107) run once per non-constant track,
 * each on its own sub-problem with the observing cameras constant.  problem->points is
 * updated for usable solutions.  termination: 0/1/2 as in tmi_ba_summary, 3 evaluation
 * failed at the start, -1 not adjusted.  Outputs may be NULL. */
int32_t oracle_adjust_tracks(tmi_ba_problem* problem, const tmi_ba_options* options,
                             int8_t* termination, int32_t* iterations, double* initial_cost,
                             double* final_cost);

/* SelectGoodTracksForBundleAdjustment (select_good_tracks_for_bundle_adjustment.cc:81-327)
 * over the flattened problem; selected[num_points] = 1 for the tracks to optimise;
 * view_mask[num_cameras] (NULL = all) restricts the views that select (:280-327).
 * stats_len / stats_err (may be NULL): truncated track length and mean squared
 * reprojection error per track (:81-110). */
int32_t oracle_select_good_tracks(const tmi_ba_problem* problem, int32_t long_track_length_threshold,
                                  int32_t image_grid_cell_size_pixels,
                                  int32_t min_num_optimized_tracks_per_view,
                                  const uint8_t* view_mask, uint8_t* selected,
                                  int32_t* stats_len, double* stats_err);

/* theia::BundleAdjustTwoViews for every pair of the batch (bundle_adjust_two_views.cc:113-191),
 * through the LM above.  Same arrays and codes as tmi_ba_adjust_two_views. */
int32_t oracle_adjust_two_views(tmi_ba_two_view_batch* batch, int32_t point_dof, int32_t max_num_iterations,
                                int8_t* termination, int32_t* iterations, double* initial_cost,
                                double* final_cost);

/* BundleAdjustTwoViewsAngular pair by pair (bundle_adjust_two_views.cc:193-240,
 * angular_epipolar_error.h:47-89, unit_norm_three_vector_parameterization.h:45-63). */
int32_t oracle_adjust_two_views_angular(tmi_ba_two_view_angular_batch* batch, int32_t max_num_iterations,
                                        int8_t* termination, int32_t* iterations, double* initial_cost,
                                        double* final_cost);
/* residual of one correspondence at (rotation, position): returns 0 where the functor returns false */
int32_t oracle_angular_epipolar_error(const double* rotation, const double* position, const double* f1,
                                      const double* f2, double* residual);

/* test hook: the visibility cluster (CLUSTER_JACOBI without shared intrinsics blocks) of every camera in the last
 * oracle_ba_solve that built them; returns the number of cameras or -1 */
int32_t oracle_last_visibility_clusters(int32_t* out, int32_t n);
/* test hook of CLUSTER_TRIDIAGONAL: per camera the segment (chain of clusters along the degree-2 maximum spanning forest
 * of the cluster graph) its view block belongs to and the position of its cluster in that chain, -1: in no segment */
int32_t oracle_last_tridiagonal_segments(int32_t* segment, int32_t* ordinal, int32_t n);
int32_t oracle_num_threads(void);
/* OpenMP threads used by the calls that follow (bench.py: single-thread baseline). */
void oracle_set_num_threads(int32_t n);

#ifdef __cplusplus
}
#endif
#endif
