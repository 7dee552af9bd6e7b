// Batched theia::BundleAdjustTwoViews (reference
// src/theia/sfm/bundle_adjustment/bundle_adjust_two_views.cc:113-191, called once per view pair
// from two_view_match_geometric_verification.cc:285): ONE WAVEFRONT PER VIEW PAIR runs the pair's
// whole trust-region solve on the device.
//
// The reference problem of a pair: camera 1 extrinsics constant, camera 2 extrinsics free (6),
// each camera's intrinsics either constant or free in the focal length only
// (AddCameraParametersToProblem :66-98), every correspondence a homogeneous 4-vector point seen
// by both cameras (ReprojectionError residuals, no loss), points eliminated first (ordering
// groups 0 / 1 / 2, :147-174), DENSE_SCHUR, at most 200 iterations and otherwise Ceres'
// DEFAULT solver options (SetSolverOptions :58-68 touches nothing else: tolerances
// 1e-6 / 1e-10 / 1e-8, radius 1e4 <= 1e16, no inner iterations).  The LM below is the loop of
// engine.hip (Ceres 1.14 TrustRegionMinimizer semantics) on that shape:
//   reduced camera system  c = [f1 | C2, w2 | f2]   (8 columns, absent ones masked),
//   lane i walks correspondences i, i + 64, ...: per point V (DP x DP), g_p, W (8 x DP),
//   S = sum (Jc^T Jc - W V'^-1 W^T), g~ = sum (Jc^T r - W V'^-1 g_p) reduced over the wave by
//   shuffles, an 8 x 8 Cholesky per lane (redundantly), then per point the back-substitution, the
//   model cost change, the candidate and its cost.  Jacobians are RECOMPUTED in the second pass
//   instead of stored: the only per-point state in memory is the Jacobi scale and the candidate.
#pragma once
#include <hip/hip_runtime.h>

#include "camera_models.h"
#include "kernels.h"

namespace tmi {

constexpr int kTvNC = 8;  // reduced columns: f1, C2 (3), w2 (3), f2

struct TwoViewBatch {
  int num_pairs;
  const double* ext1;      // [6 P]
  double* ext2;            // [6 P] in/out
  const int* model1;       // [P]
  const int* model2;
  double* intr1;           // [10 P] zero padded, in/out (focal length)
  double* intr2;
  const unsigned char* const1;  // [P] TwoViewBundleAdjustmentOptions::constant_camera1_intrinsics
  const unsigned char* const2;
  const long long* corr_ptr;    // [P + 1]
  const double* feat1;     // [2 N]
  const double* feat2;
  double* points;          // [4 N] in/out
  double* points_c;        // [4 N] scratch: candidate points
  double* scale_p;         // [4 N] scratch: Jacobi scales of the point columns
};

struct TwoViewArgs {
  int point_dof;           // 4 = the reference (homogeneous, no parameterization), 3 holds w fixed
  int max_num_iterations;  // 200 in the reference
  int jacobi_scaling;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_radius, max_radius, min_radius, min_relative_decrease, lm_lo, lm_hi;
  int max_num_consecutive_invalid_steps;
};

// symmetric 8 x 8 Cholesky solve in registers: S y = b.  Returns false if not positive definite.
__device__ __forceinline__ bool tv_solve(const double (&S)[sym_size(kTvNC)], const double (&b)[kTvNC],
                                         double (&y)[kTvNC]) {
  constexpr int N = kTvNC;
  double L[N][N];
  bool pd = true;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double d = S[sym_idx(j, j, N)];
#pragma unroll
    for (int m = 0; m < j; ++m) d -= L[j][m] * L[j][m];
    if (!(d > 0.0)) {
      pd = false;
      d = 1.0;
    }
    const double l = sqrt(d);
    L[j][j] = l;
    const double il = 1.0 / l;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double t = S[sym_idx(j, i, N)];
#pragma unroll
      for (int m = 0; m < j; ++m) t -= L[i][m] * L[j][m];
      L[i][j] = t * il;
    }
  }
  double z[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double t = b[i];
#pragma unroll
    for (int m = 0; m < i; ++m) t -= L[i][m] * z[m];
    z[i] = t / L[i][i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double t = z[i];
#pragma unroll
    for (int m = i + 1; m < N; ++m) t -= L[m][i] * y[m];
    y[i] = t / L[i][i];
  }
  return pd;
}

// Jacobian blocks of one correspondence at (E1, K1, E2, K2, X): camera columns Jc (2 obs x 2
// rows x 8), point columns Jp (2 obs x 2 rows x DP), residuals r (4).  cmask: free columns;
// sc: column scales (camera), sp: point scales.  Returns false if either evaluation fails.
template <int DP>
__device__ __forceinline__ bool tv_jacobians(int model1, int model2, const double* E1, const double* K1,
                                             const double* E2, const double* K2, const double X[4],
                                             const double* f1, const double* f2, unsigned cmask,
                                             const double (&sc)[kTvNC], const double (&sp)[DP],
                                             double (&r)[4], double (&a1)[2], double (&a2)[2][7],
                                             double (&Jp)[2][2][DP]) {
  double rr[2], Jext[2][6], Jint[2][10], Jpt[2][4];
  bool ok = reprojection_error<true, double>(model1, E1, K1, X, f1[0], f1[1], rr, Jext, Jint, Jpt);
  r[0] = rr[0];
  r[1] = rr[1];
  a1[0] = (cmask & 1u) ? Jint[0][0] * sc[0] : 0.0;
  a1[1] = (cmask & 1u) ? Jint[1][0] * sc[0] : 0.0;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int a = 0; a < DP; ++a) Jp[0][q][a] = Jpt[q][a] * sp[a];
  ok = reprojection_error<true, double>(model2, E2, K2, X, f2[0], f2[1], rr, Jext, Jint, Jpt) && ok;
  r[2] = rr[0];
  r[3] = rr[1];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
#pragma unroll
    for (int c = 0; c < 6; ++c) a2[q][c] = Jext[q][c] * sc[1 + c];
    a2[q][6] = (cmask & 0x80u) ? Jint[q][0] * sc[7] : 0.0;
#pragma unroll
    for (int a = 0; a < DP; ++a) Jp[1][q][a] = Jpt[q][a] * sp[a];
  }
  return ok;
}

// termination[pair]: 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE, 3 evaluation failed at the start
// point, -1 no correspondences.  Parameters are written back unless the code is 2 or 3.
template <int DP>
__global__ __launch_bounds__(256) void two_view_lm_kernel(TwoViewBatch B, TwoViewArgs A,
                                                          signed char* __restrict__ termination,
                                                          int* __restrict__ iterations,
                                                          double* __restrict__ initial_cost,
                                                          double* __restrict__ final_cost) {
  constexpr int NC = kTvNC, NSC = sym_size(NC), NSP = sym_size(DP);
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pair >= B.num_pairs) return;  // wave-uniform; no workgroup barrier below
  const long long c0 = B.corr_ptr[pair], c1 = B.corr_ptr[pair + 1];
  if (c1 <= c0) {
    if (lane == 0) {
      termination[pair] = -1;
      iterations[pair] = 0;
      initial_cost[pair] = 0.0;
      final_cost[pair] = 0.0;
    }
    return;
  }
  const int model1 = B.model1[pair], model2 = B.model2[pair];
  double E1[6], E2[6], K1[10], K2[10], E2c[6], K1c[10], K2c[10];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    E1[i] = B.ext1[(size_t)pair * 6 + i];
    E2[i] = B.ext2[(size_t)pair * 6 + i];
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    K1[i] = B.intr1[(size_t)pair * 10 + i];
    K2[i] = B.intr2[(size_t)pair * 10 + i];
  }
  const bool free1 = !B.const1[pair], free2 = !B.const2[pair];
  const unsigned cmask = (free1 ? 1u : 0u) | 0x7eu | (free2 ? 0x80u : 0u);
  const int nk1 = model1 == 0 ? 7 : model1 == 1 ? 10 : model1 == 2 ? 9 : 5;
  const int nk2 = model2 == 0 ? 7 : model2 == 1 ? 10 : model2 == 2 ? 9 : 5;
  double sc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) sc[c] = 1.0;

  // |x|^2 over every coordinate of every non-constant parameter block
  auto state_norm_sq = [&](const double* e2, const double* k1, const double* k2, const double* pts) {
    double acc = 0.0;
    for (long long q = c0 + lane; q < c1; q += 64)
#pragma unroll
      for (int a = 0; a < 4; ++a) acc += pts[4 * q + a] * pts[4 * q + a];
    acc = wave_sum(acc);
#pragma unroll
    for (int a = 0; a < 6; ++a) acc += e2[a] * e2[a];
    if (free1)
      for (int a = 0; a < nk1; ++a) acc += k1[a] * k1[a];
    if (free2)
      for (int a = 0; a < nk2; ++a) acc += k2[a] * k2[a];
    return acc;
  };

  // pass A: normal equations at (E2, K1, K2, points) for 1 / radius = inv_radius.
  //   out: S (damped reduced matrix), gt (reduced gradient), udiag, gc (camera gradient), cost,
  //   gmax_p (max |point gradient / scale|), flags
  double S[NSC], gt[NC], udiag[NC], gc[NC];
  double cost = 0.0, gmax_p = 0.0;
  bool eval_ok = true, point_ok = true;
  auto build = [&](double inv_radius, bool scales_from_here) {
    double Sa[NSC], gta[NC], ud[NC], gca[NC];
#pragma unroll
    for (int i = 0; i < NSC; ++i) Sa[i] = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) gta[c] = ud[c] = gca[c] = 0.0;
    double cacc = 0.0, gm = 0.0;
    bool ok_all = true, pd_all = true;
    for (long long q = c0 + lane; q < c1; q += 64) {
      double X[4], sp[DP];
#pragma unroll
      for (int a = 0; a < 4; ++a) X[a] = B.points[4 * q + a];
#pragma unroll
      for (int a = 0; a < DP; ++a) sp[a] = scales_from_here ? 1.0 : B.scale_p[4 * q + a];
      double r[4], a1[2], a2[2][7], Jp[2][2][DP];
      const bool ok = tv_jacobians<DP>(model1, model2, E1, K1, E2, K2, X, B.feat1 + 2 * q, B.feat2 + 2 * q,
                                       cmask, sc, sp, r, a1, a2, Jp);
      if (!ok) {
        ok_all = false;
        continue;
      }
      cacc += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
      // point block
      double V[NSP], g[DP];
#pragma unroll
      for (int a = 0; a < DP; ++a) {
#pragma unroll
        for (int b = a; b < DP; ++b)
          V[sym_idx(a, b, DP)] = Jp[0][0][a] * Jp[0][0][b] + Jp[0][1][a] * Jp[0][1][b] +
                                 Jp[1][0][a] * Jp[1][0][b] + Jp[1][1][a] * Jp[1][1][b];
        g[a] = Jp[0][0][a] * r[0] + Jp[0][1][a] * r[1] + Jp[1][0][a] * r[2] + Jp[1][1][a] * r[3];
        gm = fmax(gm, fabs(g[a] / sp[a]));
      }
      if (scales_from_here) {
#pragma unroll
        for (int a = 0; a < DP; ++a) B.scale_p[4 * q + a] = 1.0 / (1.0 + sqrt(V[sym_idx(a, a, DP)]));
      }
      // camera side, unreduced
      ud[0] += a1[0] * a1[0] + a1[1] * a1[1];
      gca[0] += a1[0] * r[0] + a1[1] * r[1];
#pragma unroll
      for (int c = 0; c < 7; ++c) {
        ud[1 + c] += a2[0][c] * a2[0][c] + a2[1][c] * a2[1][c];
        gca[1 + c] += a2[0][c] * r[2] + a2[1][c] * r[3];
      }
      // V' = V + D, Cholesky, inverse
      double Lm[DP][DP];
      bool pd = true;
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        const double dj = V[sym_idx(j, j, DP)];
        double d = dj + fmin(fmax(dj, A.lm_lo), A.lm_hi) * inv_radius;
#pragma unroll
        for (int m = 0; m < j; ++m) d -= Lm[j][m] * Lm[j][m];
        if (!(d > 0.0)) {
          pd = false;
          d = 1.0;
        }
        const double l = sqrt(d);
        Lm[j][j] = l;
        const double il = 1.0 / l;
#pragma unroll
        for (int i = j + 1; i < DP; ++i) {
          double t = V[sym_idx(j, i, DP)];
#pragma unroll
          for (int m = 0; m < j; ++m) t -= Lm[i][m] * Lm[j][m];
          Lm[i][j] = t * il;
        }
      }
      if (!pd) pd_all = false;
      double Li[DP][DP];
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        Li[j][j] = 1.0 / Lm[j][j];
#pragma unroll
        for (int i = j + 1; i < DP; ++i) {
          double t = 0.0;
#pragma unroll
          for (int m = j; m < i; ++m) t -= Lm[i][m] * Li[m][j];
          Li[i][j] = t / Lm[i][i];
        }
      }
      // Q = L^-1 Jp^T per residual row: the rank-DP factor Y = Jc^T Q^T of the Schur term
      double Q[4][DP];  // rows: obs1 r0, obs1 r1, obs2 r0, obs2 r1
#pragma unroll
      for (int row = 0; row < 4; ++row)
#pragma unroll
        for (int b = 0; b < DP; ++b) {
          double t = 0.0;
#pragma unroll
          for (int a = 0; a <= b; ++a) t += Li[b][a] * Jp[row >> 1][row & 1][a];
          Q[row][b] = t;
        }
      double zq[DP];  // L^-1 g_p
#pragma unroll
      for (int b = 0; b < DP; ++b) {
        double t = 0.0;
#pragma unroll
        for (int a = 0; a <= b; ++a) t += Li[b][a] * g[a];
        zq[b] = t;
      }
      // Y (NC x DP): row 0 from observation 1, rows 1..7 from observation 2
      double Y[NC][DP];
#pragma unroll
      for (int b = 0; b < DP; ++b) {
        Y[0][b] = a1[0] * Q[0][b] + a1[1] * Q[1][b];
#pragma unroll
        for (int c = 0; c < 7; ++c) Y[1 + c][b] = a2[0][c] * Q[2][b] + a2[1][c] * Q[3][b];
      }
      // S += Jc^T Jc - Y Y^T ;  gt += Jc^T r - Y zq
#pragma unroll
      for (int i = 0; i < NC; ++i) {
#pragma unroll
        for (int j = i; j < NC; ++j) {
          double t = 0.0;
          if (i == 0 && j == 0) t = a1[0] * a1[0] + a1[1] * a1[1];
          if (i >= 1 && j >= 1) t = a2[0][i - 1] * a2[0][j - 1] + a2[1][i - 1] * a2[1][j - 1];
#pragma unroll
          for (int b = 0; b < DP; ++b) t -= Y[i][b] * Y[j][b];
          Sa[sym_idx(i, j, NC)] += t;
        }
        double t = (i == 0) ? a1[0] * r[0] + a1[1] * r[1] : a2[0][i - 1] * r[2] + a2[1][i - 1] * r[3];
#pragma unroll
        for (int b = 0; b < DP; ++b) t -= Y[i][b] * zq[b];
        gta[i] += t;
      }
    }
#pragma unroll
    for (int i = 0; i < NSC; ++i) S[i] = wave_sum(Sa[i]);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      gt[c] = wave_sum(gta[c]);
      udiag[c] = wave_sum(ud[c]);
      gc[c] = wave_sum(gca[c]);
    }
    cost = wave_sum(cacc);
    gmax_p = wave_max(gm);
    eval_ok = __all(ok_all ? 1 : 0) != 0;
    point_ok = __all(pd_all ? 1 : 0) != 0;
    // LM diagonal of the camera columns; masked columns become the identity
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      if (cmask & (1u << c))
        S[sym_idx(c, c, NC)] += fmin(fmax(udiag[c], A.lm_lo), A.lm_hi) * inv_radius;
      else
        S[sym_idx(c, c, NC)] = 1.0;
    }
  };

  // ---- iteration zero ----
  build(1.0, /*scales_from_here=*/true);
  int term = 1, iter = 0;
  const double cost0 = cost;
  if (!eval_ok) {
    if (lane == 0) {
      termination[pair] = 3;
      iterations[pair] = 0;
      initial_cost[pair] = cost0;
      final_cost[pair] = cost0;
    }
    return;
  }
  if (A.jacobi_scaling) {
#pragma unroll
    for (int c = 0; c < NC; ++c) sc[c] = 1.0 / (1.0 + sqrt(udiag[c]));
  } else {
    for (long long q = c0 + lane; q < c1; q += 64)
#pragma unroll
      for (int a = 0; a < DP; ++a) B.scale_p[4 * q + a] = 1.0;
  }
  double x_norm = sqrt(state_norm_sq(E2, K1, K2, B.points));
  double radius = A.initial_radius, decrease_factor = 2.0;
  int invalid_run = 0;
  bool need_gradient_check = true;
  for (;;) {
    if (iter >= A.max_num_iterations) break;
    ++iter;
    const double inv_radius = 1.0 / radius;
    build(inv_radius, false);
    if (need_gradient_check) {
      double gmax = gmax_p;
#pragma unroll
      for (int c = 0; c < NC; ++c)
        if (cmask & (1u << c)) gmax = fmax(gmax, fabs(gc[c] / sc[c]));
      need_gradient_check = false;
      if (!(gmax > A.gradient_tolerance)) {
        term = 0;
        --iter;
        break;
      }
    }
    double yc[NC];
    bool usable = tv_solve(S, gt, yc) && point_ok;
#pragma unroll
    for (int c = 0; c < NC; ++c)
      if (!(cmask & (1u << c))) yc[c] = 0.0;
    // candidate cameras
    double step_c = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const double d = -yc[1 + i] * sc[1 + i];
      E2c[i] = E2[i] + d;
      step_c += d * d;
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      K1c[i] = K1[i];
      K2c[i] = K2[i];
    }
    if (free1) {
      const double d = -yc[0] * sc[0];
      K1c[0] += d;
      step_c += d * d;
    }
    if (free2) {
      const double d = -yc[7] * sc[7];
      K2c[0] += d;
      step_c += d * d;
    }
    // pass B: back-substitution, model cost change, candidate points and their cost
    double mcc = 0.0, step_p = 0.0, cand_cost = 0.0;
    bool cand_ok = true;
    if (usable) {
      double macc = 0.0, sacc = 0.0, cacc = 0.0;
      bool ok_all = true;
      for (long long q = c0 + lane; q < c1; q += 64) {
        double X[4], sp[DP];
#pragma unroll
        for (int a = 0; a < 4; ++a) X[a] = B.points[4 * q + a];
#pragma unroll
        for (int a = 0; a < DP; ++a) sp[a] = B.scale_p[4 * q + a];
        double r[4], a1[2], a2[2][7], Jp[2][2][DP];
        tv_jacobians<DP>(model1, model2, E1, K1, E2, K2, X, B.feat1 + 2 * q, B.feat2 + 2 * q, cmask, sc, sp, r,
                         a1, a2, Jp);
        // u = Jc yc per residual row
        double u[4];
        u[0] = a1[0] * yc[0];
        u[1] = a1[1] * yc[0];
        u[2] = 0.0;
        u[3] = 0.0;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          u[2] += a2[0][c] * yc[1 + c];
          u[3] += a2[1][c] * yc[1 + c];
        }
        double V[NSP], w[DP];
#pragma unroll
        for (int a = 0; a < DP; ++a) {
#pragma unroll
          for (int b = a; b < DP; ++b)
            V[sym_idx(a, b, DP)] = Jp[0][0][a] * Jp[0][0][b] + Jp[0][1][a] * Jp[0][1][b] +
                                   Jp[1][0][a] * Jp[1][0][b] + Jp[1][1][a] * Jp[1][1][b];
          // g_p - W^T yc
          w[a] = Jp[0][0][a] * (r[0] - u[0]) + Jp[0][1][a] * (r[1] - u[1]) + Jp[1][0][a] * (r[2] - u[2]) +
                 Jp[1][1][a] * (r[3] - u[3]);
        }
        double Lm[DP][DP];
#pragma unroll
        for (int j = 0; j < DP; ++j) {
          const double dj = V[sym_idx(j, j, DP)];
          double d = dj + fmin(fmax(dj, A.lm_lo), A.lm_hi) * inv_radius;
#pragma unroll
          for (int m = 0; m < j; ++m) d -= Lm[j][m] * Lm[j][m];
          if (!(d > 0.0)) d = 1.0;
          const double l = sqrt(d);
          Lm[j][j] = l;
          const double il = 1.0 / l;
#pragma unroll
          for (int i = j + 1; i < DP; ++i) {
            double t = V[sym_idx(j, i, DP)];
#pragma unroll
            for (int m = 0; m < j; ++m) t -= Lm[i][m] * Lm[j][m];
            Lm[i][j] = t * il;
          }
        }
        double z[DP], yp[DP];
#pragma unroll
        for (int i = 0; i < DP; ++i) {
          double t = w[i];
#pragma unroll
          for (int m = 0; m < i; ++m) t -= Lm[i][m] * z[m];
          z[i] = t / Lm[i][i];
        }
#pragma unroll
        for (int i = DP - 1; i >= 0; --i) {
          double t = z[i];
#pragma unroll
          for (int m = i + 1; m < DP; ++m) t -= Lm[m][i] * yp[m];
          yp[i] = t / Lm[i][i];
        }
        // model residual m = J delta, delta = -y
#pragma unroll
        for (int row = 0; row < 4; ++row) {
          double m = u[row];
#pragma unroll
          for (int a = 0; a < DP; ++a) m += Jp[row >> 1][row & 1][a] * yp[a];
          m = -m;
          macc -= m * (r[row] + 0.5 * m);
        }
        double Xc[4] = {X[0], X[1], X[2], X[3]};
#pragma unroll
        for (int a = 0; a < DP; ++a) {
          const double d = -yp[a] * sp[a];
          Xc[a] += d;
          sacc += d * d;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) B.points_c[4 * q + a] = Xc[a];
        double rr[2];
        double (*nul6)[6] = nullptr;
        double Jint[2][10];
        double (*nul4)[4] = nullptr;
        bool ok = reprojection_error<false, double>(model1, E1, K1c, Xc, B.feat1[2 * q], B.feat1[2 * q + 1], rr,
                                                    nul6, Jint, nul4);
        double c2 = rr[0] * rr[0] + rr[1] * rr[1];
        ok = reprojection_error<false, double>(model2, E2c, K2c, Xc, B.feat2[2 * q], B.feat2[2 * q + 1], rr, nul6,
                                               Jint, nul4) && ok;
        c2 += rr[0] * rr[0] + rr[1] * rr[1];
        if (ok) cacc += 0.5 * c2;
        else ok_all = false;
      }
      mcc = wave_sum(macc);
      step_p = wave_sum(sacc);
      cand_cost = wave_sum(cacc);
      cand_ok = __all(ok_all ? 1 : 0) != 0;
      if (!(mcc > 0.0)) usable = false;
    }
    if (!usable) {  // HandleInvalidStep
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
    if (!cand_ok) cand_cost = 1.7976931348623157e308;
    const double step_norm = sqrt(step_c + step_p);
    if (step_norm <= A.parameter_tolerance * (x_norm + A.parameter_tolerance)) {
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
      for (int i = 0; i < 6; ++i) E2[i] = E2c[i];
      K1[0] = K1c[0];
      K2[0] = K2c[0];
      for (long long q = c0 + lane; q < c1; q += 64)
#pragma unroll
        for (int a = 0; a < 4; ++a) B.points[4 * q + a] = B.points_c[4 * q + a];
      cost = cand_cost;
      x_norm = sqrt(state_norm_sq(E2, K1, K2, B.points));
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * relative_decrease - 1.0, 3.0));
      radius = fmin(A.max_radius, radius);
      decrease_factor = 2.0;
      need_gradient_check = true;
    } else {
      radius /= decrease_factor;
      decrease_factor *= 2.0;
    }
    if (radius < A.min_radius) {
      term = 0;
      break;
    }
  }
  if (lane == 0) {
    termination[pair] = (signed char)term;
    iterations[pair] = iter;
    initial_cost[pair] = cost0;
    final_cost[pair] = cost;
    if (term != 2) {
#pragma unroll
      for (int i = 0; i < 6; ++i) B.ext2[(size_t)pair * 6 + i] = E2[i];
      B.intr1[(size_t)pair * 10] = K1[0];
      B.intr2[(size_t)pair * 10] = K2[0];
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Batched theia::BundleAdjustTwoViewsAngular (bundle_adjust_two_views.cc:193-240): relative
// rotation (angle-axis) and unit-norm relative position of a view pair from one AngularEpipolarError
// residual per correspondence (angular_epipolar_error.h:47-89), position on the unit sphere through
// AutoDiffLocalParameterization<UnitNormThreeVectorParameterization, 3, 3>
// (unit_norm_three_vector_parameterization.h:45-63).  ONE WAVEFRONT PER PAIR: lanes walk the
// correspondences, residual and its 6 derivatives come from forward-mode dual numbers (what Ceres'
// autodiff evaluates), J^T J / J^T r are reduced over the wave and every lane solves the 6 x 6 damped
// system itself, so the trust-region state machine (Ceres 1.14 semantics, the loop of track_lm_kernel)
// runs replicated without divergence.
// ------------------------------------------------------------------------------------------------
namespace tva {
struct D6 {
  double a;
  double v[6];
};
__device__ __forceinline__ D6 cst(double c) {
  D6 r;
  r.a = c;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = 0.0;
  return r;
}
__device__ __forceinline__ D6 var(double c, int k) {
  D6 r = cst(c);
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = (i == k) ? 1.0 : 0.0;
  return r;
}
__device__ __forceinline__ D6 operator+(const D6& f, const D6& g) {
  D6 r;
  r.a = f.a + g.a;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = f.v[i] + g.v[i];
  return r;
}
__device__ __forceinline__ D6 operator-(const D6& f, const D6& g) {
  D6 r;
  r.a = f.a - g.a;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = f.v[i] - g.v[i];
  return r;
}
__device__ __forceinline__ D6 operator-(const D6& f) {
  D6 r;
  r.a = -f.a;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = -f.v[i];
  return r;
}
__device__ __forceinline__ D6 operator*(const D6& f, const D6& g) {
  D6 r;
  r.a = f.a * g.a;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return r;
}
__device__ __forceinline__ D6 operator*(const D6& f, double c) {
  D6 r;
  r.a = f.a * c;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = f.v[i] * c;
  return r;
}
__device__ __forceinline__ D6 operator/(const D6& f, const D6& g) {
  D6 r;
  const double gi = 1.0 / g.a;
  const double fr = f.a * gi;
  r.a = fr;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = (f.v[i] - fr * g.v[i]) * gi;
  return r;
}
__device__ __forceinline__ D6 dsqrt(const D6& f) {
  D6 r;
  const double t = sqrt(f.a);
  const double s = 1.0 / (2.0 * t);
  r.a = t;
#pragma unroll
  for (int i = 0; i < 6; ++i) r.v[i] = f.v[i] * s;
  return r;
}
__device__ __forceinline__ void dsincos(const D6& f, D6* s, D6* c) {
  double sv, cv;
  sincos(f.a, &sv, &cv);
  s->a = sv;
  c->a = cv;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    s->v[i] = cv * f.v[i];
    c->v[i] = -sv * f.v[i];
  }
}

// ceres::AngleAxisToRotationMatrix (published ceres/rotation.h, 1.x) on duals
__device__ __forceinline__ void rotation(const D6 (&aa)[3], D6 (&R)[3][3]) {
  const D6 one = cst(1.0);
  const D6 theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2.a > kDblEpsilon) {
    const D6 theta = dsqrt(theta2);
    const D6 wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    D6 s, c;
    dsincos(theta, &s, &c);
    const D6 omc = one - c;
    R[0][0] = c + wx * wx * omc;
    R[1][0] = wz * s + wx * wy * omc;
    R[2][0] = -(wy * s) + wx * wz * omc;
    R[0][1] = wx * wy * omc - wz * s;
    R[1][1] = c + wy * wy * omc;
    R[2][1] = wx * s + wy * wz * omc;
    R[0][2] = wy * s + wx * wz * omc;
    R[1][2] = -(wx * s) + wy * wz * omc;
    R[2][2] = c + wz * wz * omc;
  } else {
    R[0][0] = one;    R[0][1] = -aa[2]; R[0][2] = aa[1];
    R[1][0] = aa[2];  R[1][1] = one;    R[1][2] = -aa[0];
    R[2][0] = -aa[1]; R[2][1] = aa[0];  R[2][2] = one;
  }
}

// angular_epipolar_error.h:52-84 with R and T = I - t t^T already formed.  false where the functor is.
__device__ __forceinline__ bool residual(const D6 (&R)[3][3], const D6 (&T)[3][3], const D6 (&t)[3], double f1x,
                                         double f1y, double f2x, double f2y, D6* out) {
  const double f1[3] = {f1x, f1y, 1.0}, f2[3] = {f2x, f2y, 1.0};
  D6 Rf2[3], Rtf2[3], Tf1[3], TRtf2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Rf2[i] = R[i][0] * f2[0] + R[i][1] * f2[1] + R[i][2] * f2[2];
    Rtf2[i] = R[0][i] * f2[0] + R[1][i] * f2[1] + R[2][i] * f2[2];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Tf1[i] = T[i][0] * f1[0] + T[i][1] * f1[1] + T[i][2] * f1[2];
    TRtf2[i] = T[i][0] * Rtf2[0] + T[i][1] * Rtf2[1] + T[i][2] * Rtf2[2];
  }
  D6 a = Tf1[0] * f1[0] + Tf1[1] * f1[1] + Tf1[2] * f1[2];
  a = a + Rf2[0] * TRtf2[0] + Rf2[1] * TRtf2[1] + Rf2[2] * TRtf2[2];
  const D6 cx = Rtf2[2] * f1[1] - Rtf2[1] * f1[2];
  const D6 cy = Rtf2[0] * f1[2] - Rtf2[2] * f1[0];
  const D6 cz = Rtf2[1] * f1[0] - Rtf2[0] * f1[1];
  const D6 b = t[0] * cx + t[1] * cy + t[2] * cz;
  const D6 sq = (a * a) * 0.25 - b * b;
  if (sq.a < 0.0) return false;
  *out = a * 0.5 - dsqrt(sq);
  return true;
}

// UnitNormThreeVectorParameterization::operator()
__device__ __forceinline__ void unit_plus(const double x[3], const double d[3], double out[3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i) out[i] = x[i] + d[i];
  const double sq = out[0] * out[0] + out[1] * out[1] + out[2] * out[2];
  if (sq > 0.0) {
    const double nrm = sqrt(sq);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] /= nrm;
  }
}

// One pass over the pair's correspondences at x = [rotation | position].  JAC: also H = Js^T Js (packed
// upper), g = Js^T r of the SCALED LOCAL Jacobian.  All lanes return the same sums.  false on failure.
template <bool JAC>
__device__ __forceinline__ bool linearize(const double (&x)[6], const double* __restrict__ f1,
                                          const double* __restrict__ f2, long long n, const double (&sc)[6],
                                          double (&H)[21], double (&g)[6], double* cost, int lane) {
  // d Plus(x, delta) / d delta at 0 of the position block, by duals (AutoDiffLocalParameterization)
  double P[3][3];
  if (JAC) {
    D6 xp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) xp[i] = var(x[3 + i], i);
    const D6 sq = xp[0] * xp[0] + xp[1] * xp[1] + xp[2] * xp[2];
    if (sq.a > 0.0) {
      const D6 nrm = dsqrt(sq);
#pragma unroll
      for (int i = 0; i < 3; ++i) xp[i] = xp[i] / nrm;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) P[i][k] = xp[i].v[k];
  }
  D6 rot[3], t[3], R[3][3], T[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    rot[i] = JAC ? var(x[i], i) : cst(x[i]);
    t[i] = JAC ? var(x[3 + i], 3 + i) : cst(x[3 + i]);
  }
  rotation(rot, R);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T[i][j] = cst(i == j ? 1.0 : 0.0) - t[i] * t[j];
  double Ha[21], ga[6], c = 0.0, bad = 0.0;
#pragma unroll
  for (int i = 0; i < 21; ++i) Ha[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i) ga[i] = 0.0;
  for (long long q = lane; q < n; q += 64) {
    D6 e;
    if (!residual(R, T, t, f1[2 * q], f1[2 * q + 1], f2[2 * q], f2[2 * q + 1], &e)) {
      bad = 1.0;
      continue;
    }
    c += 0.5 * e.a * e.a;
    if (JAC) {
      double J[6];
#pragma unroll
      for (int k = 0; k < 3; ++k) J[k] = e.v[k] * sc[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) J[3 + k] = (e.v[3] * P[0][k] + e.v[4] * P[1][k] + e.v[5] * P[2][k]) * sc[3 + k];
      int idx = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int b = a; b < 6; ++b) Ha[idx++] += J[a] * J[b];
        ga[a] += J[a] * e.a;
      }
    }
  }
  if (JAC) {
#pragma unroll
    for (int i = 0; i < 21; ++i) H[i] = wave_sum(Ha[i]);
#pragma unroll
    for (int i = 0; i < 6; ++i) g[i] = wave_sum(ga[i]);
  }
  *cost = wave_sum(c);
  return wave_sum(bad) == 0.0;
}

// (H + diag) y = g by Cholesky in registers; false if not positive definite
__device__ __forceinline__ bool solve6(const double (&H)[21], const double (&dadd)[6], const double (&g)[6],
                                       double (&y)[6]) {
  double L[6][6];
  bool pd = true;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = H[sym_idx(j, j, 6)] + dadd[j];
#pragma unroll
    for (int m = 0; m < j; ++m) d -= L[j][m] * L[j][m];
    if (!(d > 0.0)) {
      pd = false;
      d = 1.0;
    }
    const double l = sqrt(d);
    L[j][j] = l;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double t = H[sym_idx(j, i, 6)];
#pragma unroll
      for (int m = 0; m < j; ++m) t -= L[i][m] * L[j][m];
      L[i][j] = t / l;
    }
  }
  double z[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double t = g[i];
#pragma unroll
    for (int m = 0; m < i; ++m) t -= L[i][m] * z[m];
    z[i] = t / L[i][i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double t = z[i];
#pragma unroll
    for (int m = i + 1; m < 6; ++m) t -= L[m][i] * y[m];
    y[i] = t / L[i][i];
  }
  return pd;
}
}  // namespace tva

struct TwoViewAngularBatch {
  int num_pairs;
  double* rot2;             // [3 P] in/out
  double* pos2;             // [3 P] in/out
  const long long* corr_ptr;
  const double* feat1;      // [2 N]
  const double* feat2;
};

__global__ __launch_bounds__(256) void two_view_angular_kernel(TwoViewAngularBatch B, TwoViewArgs A,
                                                               signed char* __restrict__ termination,
                                                               int* __restrict__ iterations,
                                                               double* __restrict__ initial_cost,
                                                               double* __restrict__ final_cost) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= B.num_pairs) return;
  const long long c0 = B.corr_ptr[p], n = B.corr_ptr[p + 1] - c0;
  if (n <= 0) {
    if (lane == 0) {
      termination[p] = -1;
      iterations[p] = 0;
      initial_cost[p] = 0.0;
      final_cost[p] = 0.0;
    }
    return;
  }
  const double* f1 = B.feat1 + 2 * c0;
  const double* f2 = B.feat2 + 2 * c0;
  double x[6];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    x[i] = B.rot2[3 * (size_t)p + i];
    x[3 + i] = B.pos2[3 * (size_t)p + i];
  }
  double sc[6] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0}, H[21], g[6], cost = 0.0;
  if (!tva::linearize<true>(x, f1, f2, n, sc, H, g, &cost, lane)) {
    if (lane == 0) {
      termination[p] = 3;
      iterations[p] = 0;
      initial_cost[p] = 0.0;
      final_cost[p] = 0.0;
    }
    return;
  }
  if (lane == 0) initial_cost[p] = cost;
  double gmax = 0.0;
#pragma unroll
  for (int a = 0; a < 6; ++a) gmax = fmax(gmax, fabs(g[a]));
#pragma unroll
  for (int a = 0; a < 6; ++a) sc[a] = 1.0 / (1.0 + sqrt(H[sym_idx(a, a, 6)]));
  tva::linearize<true>(x, f1, f2, n, sc, H, g, &cost, lane);
  double x_norm = 0.0;
#pragma unroll
  for (int a = 0; a < 6; ++a) x_norm += x[a] * x[a];
  x_norm = sqrt(x_norm);
  double radius = A.initial_radius, decrease_factor = 2.0;
  int invalid_run = 0, iter = 0, term = 1;
  if (gmax <= A.gradient_tolerance) {
    term = 0;
  } else {
    for (;;) {
      if (iter >= A.max_num_iterations) break;
      ++iter;
      double dadd[6], y[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) dadd[a] = fmin(fmax(H[sym_idx(a, a, 6)], A.lm_lo), A.lm_hi) / radius;
      bool step_ok = tva::solve6(H, dadd, g, y);
      double mcc = 0.0;
      if (step_ok) {
        double yg = 0.0, yHy = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          yg += y[a] * g[a];
          double t = 0.0;
#pragma unroll
          for (int b = 0; b < 6; ++b) t += H[a <= b ? sym_idx(a, b, 6) : sym_idx(b, a, 6)] * y[b];
          yHy += y[a] * t;
        }
        mcc = yg - 0.5 * yHy;
        if (!(mcc > 0.0)) step_ok = false;
      }
      if (!step_ok) {
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
      double xc[6], d[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) d[a] = -y[a] * sc[a];
#pragma unroll
      for (int a = 0; a < 3; ++a) xc[a] = x[a] + d[a];
      tva::unit_plus(x + 3, d + 3, xc + 3);
      double step_sq = 0.0;
#pragma unroll
      for (int a = 0; a < 6; ++a) step_sq += (xc[a] - x[a]) * (xc[a] - x[a]);
      double cand, Hd[21], gd[6];
      if (!tva::linearize<false>(xc, f1, f2, n, sc, Hd, gd, &cand, lane)) cand = 1.7976931348623157e308;
      if (sqrt(step_sq) <= A.parameter_tolerance * (x_norm + A.parameter_tolerance)) {
        term = 0;
        break;
      }
      const double cost_change = cost - cand;
      if (fabs(cost_change) <= A.function_tolerance * cost) {
        term = 0;
        break;
      }
      const double rd = cost_change / mcc;
      if (rd > A.min_relative_decrease) {
#pragma unroll
        for (int a = 0; a < 6; ++a) x[a] = xc[a];
        x_norm = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) x_norm += x[a] * x[a];
        x_norm = sqrt(x_norm);
        if (!tva::linearize<true>(x, f1, f2, n, sc, H, g, &cost, lane)) {
          term = 2;
          break;
        }
        gmax = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) gmax = fmax(gmax, fabs(g[a] / sc[a]));
        radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rd - 1.0, 3.0));
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
  if (lane != 0) return;
  termination[p] = (signed char)term;
  iterations[p] = iter;
  final_cost[p] = cost;
  if (term != 2) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      B.rot2[3 * (size_t)p + i] = x[i];
      B.pos2[3 * (size_t)p + i] = x[3 + i];
    }
  }
}

}  // namespace tmi
