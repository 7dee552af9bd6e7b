// Per-track kernels of the steps either side of the full bundle adjustment
// (SURVEY 8(f) rows 1 and 3), on the resident track-major SELL-64 layout: thread
// (slice s, lane t) owns track 64 s + t and walks its observations, exactly as
// `linearize` and `cost` do.
//
//   outlier_filter_kernel   SetOutlierTracksToUnestimated
//                           (set_outlier_tracks_to_unestimated.cc:62-133)
//   track_lm_kernel         BundleAdjustTrack for every track at once
//                           (bundle_adjustment.cc:96-107, called per track from
//                           estimate_track.cc:238-246): one independent
//                           Levenberg-Marquardt problem per thread, all cameras constant
#pragma once
#include <hip/hip_runtime.h>

#include "camera_models.h"
#include "device_view.h"
#include "kernels.h"

namespace tmi {

// ------------------------------------------------------------------------------------
// flag[lp]: 0 kept, 1 bad reprojection (mean squared error above the threshold or a
// projection behind a camera), 2 insufficient viewing angle.  mean_sq[lp]: the mean the
// reference compares (its value where the reference's loop stops).
// The angle test is the reference's O(k^2) scan with early exit (triangulation.cc:236-250);
// ray j is recomputed from the camera centre (three L2-resident loads) instead of stored.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void unit_ray(const double* __restrict__ ext, int cam, const double Xh[3],
                                         double r[3]) {
  r[0] = Xh[0] - ext[(size_t)cam * 6 + 0];
  r[1] = Xh[1] - ext[(size_t)cam * 6 + 1];
  r[2] = Xh[2] - ext[(size_t)cam * 6 + 2];
  const double n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
  if (n2 > 0.0) {  // Eigen normalized()
    const double n = sqrt(n2);
    r[0] /= n;
    r[1] /= n;
    r[2] /= n;
  }
}

__global__ __launch_bounds__(256) void outlier_filter_kernel(DeviceView v, double max_sq, double cos_min,
                                                             unsigned char* __restrict__ flag,
                                                             double* __restrict__ mean_sq) {
  // long slices: 16 lanes per track (kernels.h track_map) -- the pair scan of a 400-view track
  // is 80 000 ray pairs, shared here by 16 lanes
  const TrackMap tm = track_map(v);
  if (!tm.valid) return;
  const int lp = tm.lp;
  const int k = tm.k;
  const size_t base = tm.base;
  double X[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) X[i] = v.pts[(size_t)lp * 4 + i];
  double behind = 0.0, sum = 0.0, nproj = 0.0;
  for (int j = tm.j0; j < k; j += tm.jstep) {
    const size_t e = base + (size_t)j * 64;
    const int cam = v.obs_cam[e];
    const int4 rec = v.cam_rec[cam];
    const double* Kp = v.intr + rec.y;
    const int nk = rec.z;
    double Kv[10], E[6], px[2];
#pragma unroll
    for (int i = 0; i < 10; ++i) Kv[i] = (i < nk) ? Kp[i] : 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) E[i] = v.ext[(size_t)cam * 6 + i];
    const double depth = project_point_depth(rec.x, E, Kv, X, px);
    if (depth < 0) {  // :101-105
      behind = 1.0;
      break;
    }
    const double dx = px[0] - v.obs_xy[2 * e], dy = px[1] - v.obs_xy[2 * e + 1];
    sum += dx * dx + dy * dy;
    nproj += 1.0;
  }
  behind = group_sum(behind, tm.wide);
  sum = group_sum(sum, tm.wide);
  nproj = group_sum(nproj, tm.wide);
  const double mean = sum / nproj;
  int f = 0;
  if (behind > 0.0 || mean > max_sq) {
    f = 1;
  } else {
    const double Xh[3] = {X[0] / X[3], X[1] / X[3], X[2] / X[3]};
    double sufficient = 0.0;
    for (int i = tm.j0; i < k && sufficient == 0.0; i += tm.jstep) {
      double ri[3];
      unit_ray(v.ext, v.obs_cam[base + (size_t)i * 64], Xh, ri);
      for (int j = i + 1; j < k; ++j) {
        double rj[3];
        unit_ray(v.ext, v.obs_cam[base + (size_t)j * 64], Xh, rj);
        if (ri[0] * rj[0] + ri[1] * rj[1] + ri[2] * rj[2] < cos_min) {
          sufficient = 1.0;
          break;
        }
      }
    }
    if (group_sum(sufficient, tm.wide) == 0.0) f = 2;
  }
  if (tm.leader) {
    flag[lp] = (unsigned char)f;
    if (mean_sq) mean_sq[lp] = mean;
  }
}

// ------------------------------------------------------------------------------------
// Track statistics of SelectGoodTracksForBundleAdjustment
// (select_good_tracks_for_bundle_adjustment.cc:81-110): number of observations and mean
// squared reprojection error per track -- every observation counts, no cheirality test.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void track_stats_kernel(DeviceView v, const double* __restrict__ prep,
                                                          int* __restrict__ count, double* __restrict__ mean_sq) {
  // thread per track, 16 lanes per track on the long slices (kernels.h track_map); cameras from their
  // prepared records (the rotation matrix instead of Rodrigues per observation)
  const TrackMap tm = track_map(v);
  if (!tm.valid) return;
  const int lp = tm.lp;
  const int k = tm.k;
  double X[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) X[i] = v.pts[(size_t)lp * 4 + i];
  double sum = 0.0;
  for (int j = tm.j0; j < k; j += tm.jstep) {
    const size_t e = tm.base + (size_t)j * 64;
    const int cam = v.obs_cam[e];
    double px[2];
    project_point_prepared(v.cam_rec[cam].x, prep + (size_t)cam * kPrepStride, X, px);
    const double dx = px[0] - v.obs_xy[2 * e], dy = px[1] - v.obs_xy[2 * e + 1];
    sum += dx * dx + dy * dy;
  }
  sum = group_sum(sum, tm.wide);
  if (tm.leader) {
    count[lp] = k;
    mean_sq[lp] = sum / (double)k;
  }
}

// ------------------------------------------------------------------------------------
// Batched single-track bundle adjustment.
// ------------------------------------------------------------------------------------
struct TrackLmArgs {
  int loss_type;
  double loss_width;
  int jacobi_scaling;
  int max_num_iterations;
  int max_num_consecutive_invalid_steps;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_radius, max_radius, min_radius;
  double min_relative_decrease;
  double lm_lo, lm_hi;
};

// One pass over a track's observations at point X: robustified, column-scaled point
// Jacobian accumulated into V0 = sum Jp^T Jp (packed upper), g = sum Jp^T r, cost.
// Returns false if any residual cannot be evaluated.
// The cameras are constant here, so both passes read the PREPARED camera records (camera_models.h:
// R, C, K evaluated once per camera) instead of running Rodrigues' formula per observation and per LM
// iteration.  tm: thread per track, or 16 lanes per track on the long slices (kernels.h track_map) --
// the sums are then finished over the 16 lanes with a fixed butterfly and every lane of the group holds
// the same totals, so the trust-region loop below runs replicated and never diverges inside a group.
// UMODEL >= 0: every camera of the problem has this camera model -- the five-way model switch folds (the generic
// track_lm_kernel needs all 256 registers: one wavefront per SIMD) and the per-observation gather of cam_rec goes
template <int DP, int UMODEL = -1>
__device__ __forceinline__ bool track_linearize(const DeviceView& v, const double* __restrict__ prep,
                                                const TrackMap& tm, const double X[4], const double sp[DP],
                                                int loss_type, double loss_width, double V0[sym_size(DP)],
                                                double g[DP], double* cost) {
  constexpr int NS = sym_size(DP);
#pragma unroll
  for (int i = 0; i < NS; ++i) V0[i] = 0.0;
#pragma unroll
  for (int a = 0; a < DP; ++a) g[a] = 0.0;
  double c = 0.0;
  double bad = 0.0;
  for (int j = tm.j0; j < tm.k; j += tm.jstep) {
    const size_t e = tm.base + (size_t)j * 64;
    const int cam = v.obs_cam[e];
    const double* P = prep + (size_t)cam * kPrepStride;
    double r[2], Jext[2][6], Jint[2][10], Jpt[2][4];
    const bool ok = reprojection_error_prepared<true, double>(UMODEL >= 0 ? UMODEL : v.cam_rec[cam].x, P, X, v.obs_xy[2 * e],
                                                              v.obs_xy[2 * e + 1], r, Jext, Jint, Jpt);
    if (!ok) {
      bad = 1.0;
      continue;
    }
    const double sq = r[0] * r[0] + r[1] * r[1];
    double sqrt_rho1 = 1.0, asn = 0.0, rscale = 1.0;
    if (loss_type != 0) {
      double rho[3];
      loss_eval(loss_type, loss_width, sq, rho);
      c += 0.5 * rho[0];
      sqrt_rho1 = sqrt(rho[1]);
      rscale = sqrt_rho1;
      if (!(sq == 0.0 || rho[2] <= 0.0)) {
        const double Dd = 1.0 + 2.0 * sq * rho[2] / rho[1];
        const double alpha = 1.0 - sqrt(Dd);
        rscale = sqrt_rho1 / (1.0 - alpha);
        asn = alpha / sq;
      }
    } else {
      c += 0.5 * sq;
    }
    double J0[DP], J1[DP];
#pragma unroll
    for (int a = 0; a < DP; ++a) {
      double j0 = Jpt[0][a], j1 = Jpt[1][a];
      if (loss_type != 0) {
        const double rtj = j0 * r[0] + j1 * r[1];
        j0 = sqrt_rho1 * (j0 - asn * r[0] * rtj);
        j1 = sqrt_rho1 * (j1 - asn * r[1] * rtj);
      }
      J0[a] = j0 * sp[a];
      J1[a] = j1 * sp[a];
    }
    const double r0 = r[0] * rscale, r1 = r[1] * rscale;
#pragma unroll
    for (int a = 0; a < DP; ++a) {
#pragma unroll
      for (int b = a; b < DP; ++b) V0[sym_idx(a, b, DP)] += J0[a] * J0[b] + J1[a] * J1[b];
      g[a] += J0[a] * r0 + J1[a] * r1;
    }
  }
#pragma unroll
  for (int i = 0; i < NS; ++i) V0[i] = group_sum(V0[i], tm.wide);
#pragma unroll
  for (int a = 0; a < DP; ++a) g[a] = group_sum(g[a], tm.wide);
  *cost = group_sum(c, tm.wide);
  return group_sum(bad, tm.wide) == 0.0;
}

template <int UMODEL = -1>
__device__ __forceinline__ bool track_cost(const DeviceView& v, const double* __restrict__ prep, const TrackMap& tm,
                                           const double X[4], int loss_type, double loss_width, double* cost) {
  double c = 0.0;
  double bad = 0.0;
  for (int j = tm.j0; j < tm.k; j += tm.jstep) {
    const size_t e = tm.base + (size_t)j * 64;
    const int cam = v.obs_cam[e];
    const double* P = prep + (size_t)cam * kPrepStride;
    double r[2];
    double (*nul6)[6] = nullptr;
    double Jint[2][10];
    double (*nul4)[4] = nullptr;
    const bool ok = reprojection_error_prepared<false, double>(UMODEL >= 0 ? UMODEL : v.cam_rec[cam].x, P, X, v.obs_xy[2 * e],
                                                               v.obs_xy[2 * e + 1], r, nul6, Jint, nul4);
    if (!ok) {
      bad = 1.0;
      continue;
    }
    const double sq = r[0] * r[0] + r[1] * r[1];
    if (loss_type != 0) {
      double rho[3];
      loss_eval(loss_type, loss_width, sq, rho);
      c += 0.5 * rho[0];
    } else {
      c += 0.5 * sq;
    }
  }
  *cost = group_sum(c, tm.wide);
  return group_sum(bad, tm.wide) == 0.0;
}

// termination[lp]: 0 CONVERGENCE, 1 NO_CONVERGENCE (iteration limit), 2 FAILURE,
// 3 residual evaluation failed at the start point, -1 not a problem (padding, constant or
// unobserved track).  The point is written back unless the code is 2 or 3 (Ceres'
// IsSolutionUsable, bundle_adjuster.cc:213-216).  The trust-region loop is the one the
// full solver runs (engine.hip / Ceres 1.14 TrustRegionMinimizer) with an empty camera
// side: the step is -(V + D)^-1 g on the track's own 2k x DP Jacobian.
template <int DP, int UMODEL = -1>
__global__ __launch_bounds__(256) void track_lm_kernel(DeviceView v, const double* __restrict__ prep, TrackLmArgs A,
                                                       signed char* __restrict__ termination,
                                                       int* __restrict__ iterations,
                                                       double* __restrict__ initial_cost,
                                                       double* __restrict__ final_cost) {
  constexpr int NS = sym_size(DP);
  const TrackMap tm = track_map(v);
  if (!tm.valid) return;
  const int lp = tm.lp;
  const int k = tm.k;
  if (k == 0 || v.pt_const[lp]) {  // uniform over the lanes that share a track
    if (tm.leader) {
      termination[lp] = -1;
      iterations[lp] = 0;
      initial_cost[lp] = 0.0;
      final_cost[lp] = 0.0;
    }
    return;
  }
  double X[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) X[i] = v.pts[(size_t)lp * 4 + i];
  double sp[DP];
#pragma unroll
  for (int a = 0; a < DP; ++a) sp[a] = 1.0;
  double V0[NS], g[DP], cost;
  // iteration zero
  bool ok = track_linearize<DP, UMODEL>(v, prep, tm, X, sp, A.loss_type, A.loss_width, V0, g, &cost);
  if (tm.leader) initial_cost[lp] = cost;
  if (!ok) {
    if (tm.leader) {
      termination[lp] = 3;
      iterations[lp] = 0;
      final_cost[lp] = cost;
    }
    return;
  }
  double gmax = 0.0;
#pragma unroll
  for (int a = 0; a < DP; ++a) gmax = fmax(gmax, fabs(g[a]));
  if (A.jacobi_scaling) {
#pragma unroll
    for (int a = 0; a < DP; ++a) sp[a] = 1.0 / (1.0 + sqrt(V0[sym_idx(a, a, DP)]));
    track_linearize<DP, UMODEL>(v, prep, tm, X, sp, A.loss_type, A.loss_width, V0, g, &cost);
  }
  double x_norm = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2] + X[3] * X[3]);
  double radius = A.initial_radius, decrease_factor = 2.0;
  int invalid_run = 0, iter = 0, term = 1;
  if (gmax <= A.gradient_tolerance) {
    term = 0;
  } else {
    for (;;) {
      if (iter >= A.max_num_iterations) break;
      ++iter;
      // (V0 + D) y = g by Cholesky
      double Lm[DP][DP];
      bool step_ok = true;
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const double dj = V0[sym_idx(j, j, DP)];
        double d = dj + fmin(fmax(dj, A.lm_lo), A.lm_hi) / radius;
#pragma unroll
        for (int m = 0; m < j; ++m) d -= Lm[j][m] * Lm[j][m];
        if (!(d > 0.0)) {
          step_ok = false;
          d = 1.0;
        }
        const double l = sqrt(d);
        Lm[j][j] = l;
        const double il = 1.0 / l;
#pragma unroll
        for (int i = j + 1; i < DP; ++i) {
          double t = V0[sym_idx(j, i, DP)];
#pragma unroll
          for (int m = 0; m < j; ++m) t -= Lm[i][m] * Lm[j][m];
          Lm[i][j] = t * il;
        }
      }
      double y[DP];
      double mcc = 0.0;
      if (step_ok) {
        // forward / backward substitution
        double z[DP];
#pragma unroll
        for (int i = 0; i < DP; ++i) {
          double t = g[i];
#pragma unroll
          for (int m = 0; m < i; ++m) t -= Lm[i][m] * z[m];
          z[i] = t / Lm[i][i];
        }
#pragma unroll
        for (int i = DP - 1; i >= 0; --i) {
          double t = z[i];
#pragma unroll
          for (int m = i + 1; m < DP; ++m) t -= Lm[m][i] * y[m];
          y[i] = t / Lm[i][i];
        }
        // model cost change of the step d = -y:  y^T g - 1/2 y^T V0 y
        double yg = 0.0, yVy = 0.0;
#pragma unroll
        for (int a = 0; a < DP; ++a) {
          yg += y[a] * g[a];
          double t = 0.0;
#pragma unroll
          for (int b = 0; b < DP; ++b) t += V0[a <= b ? sym_idx(a, b, DP) : sym_idx(b, a, DP)] * y[b];
          yVy += y[a] * t;
        }
        mcc = yg - 0.5 * yVy;
        if (!(mcc > 0.0)) step_ok = false;
      }
      if (!step_ok) {  // HandleInvalidStep
        if (++invalid_run >= A.max_num_consecutive_invalid_steps) {
          term = 2;
          break;
        }
        radius /= decrease_factor;
        decrease_factor *= 2.0;
        if (radius < A.min_radius) {
          term = 0;
          break;
        }
        continue;
      }
      invalid_run = 0;
      double Xc[4] = {X[0], X[1], X[2], X[3]};
      double step_sq = 0.0;
#pragma unroll
      for (int a = 0; a < DP; ++a) {
        const double d = -y[a] * sp[a];
        Xc[a] += d;
        step_sq += d * d;
      }
      double cand_cost;
      if (!track_cost<UMODEL>(v, prep, tm, Xc, A.loss_type, A.loss_width, &cand_cost)) cand_cost = 1.7976931348623157e308;
      if (sqrt(step_sq) <= A.parameter_tolerance * (x_norm + A.parameter_tolerance)) {
        term = 0;
        break;
      }
      const double cost_change = cost - cand_cost;
      if (fabs(cost_change) <= A.function_tolerance * cost) {
        term = 0;
        break;
      }
      const double relative_decrease = cost_change / mcc;
      if (relative_decrease > A.min_relative_decrease) {  // HandleSuccessfulStep
#pragma unroll
        for (int i = 0; i < 4; ++i) X[i] = Xc[i];
        x_norm = sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2] + X[3] * X[3]);
        track_linearize<DP, UMODEL>(v, prep, tm, X, sp, A.loss_type, A.loss_width, V0, g, &cost);
        gmax = 0.0;
#pragma unroll
        for (int a = 0; a < DP; ++a) gmax = fmax(gmax, fabs(g[a] / sp[a]));
        radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * relative_decrease - 1.0, 3.0));
        radius = fmin(A.max_radius, radius);
        decrease_factor = 2.0;
        if (gmax <= A.gradient_tolerance) {
          term = 0;
          break;
        }
      } else {
        radius /= decrease_factor;
        decrease_factor *= 2.0;
      }
      if (radius < A.min_radius) {
        term = 0;
        break;
      }
    }
  }
  if (!tm.leader) return;
  termination[lp] = (signed char)term;
  iterations[lp] = iter;
  final_cost[lp] = cost;
  if (term != 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v.pts[(size_t)lp * 4 + i] = X[i];
  }
}

}  // namespace tmi
