// Inner iterations of the trust-region loop (Ceres CoordinateDescentMinimizer, called from
// TrustRegionMinimizer::DoInnerIterationsIfNeeded; Theia turns them on by default,
// bundle_adjustment.h:112, bundle_adjuster.cc:74).  One sweep minimises, with everything else
// held constant,
//   set 1  every intrinsics block (one per intrinsics group, shared or private),
//   set 2  every extrinsics block (one per view),
//   set 3  every point                       (track_lm_kernel in track_kernels.h),
// each block with its own Levenberg-Marquardt run under Ceres' default options (50 iterations,
// tolerances 1e-6 / 1e-10 / 1e-8, radius 1e4 <= 1e16, Jacobi scaling, DENSE_QR).  The blocks of
// a set share no residual, so a set is a batch of independent small problems:
//   inner_eval_kernel    one workgroup per VIEW walks that view's observations and reduces,
//                        for the block the view belongs to, cost [+ g = J^T r, H = J^T J over
//                        the block's free columns] at the block's current or candidate value;
//   inner_step_kernel    one wave per block: sums the per-view partials (a shared intrinsics
//                        block has many views), finishes iteration zero / an accepted step
//                        (gradient test, Jacobi scaling), solves (H + D) y = g, writes the
//                        candidate and the model cost change;
//   inner_decide_kernel  one wave per block: candidate cost, Ceres' tolerance / acceptance /
//                        radius rules, state update.
// The host repeats eval(J) -> step -> eval(cost) -> decide until every block of the set has
// terminated (engine.hip).  The per-block state machine is the LM loop of tmi_ba_solver_solve
// with an empty point side; oracle/ba_oracle.c runs the same loop on one-block sub-problems.
#pragma once
#include <hip/hip_runtime.h>

#include "camera_models.h"
#include "device_view.h"
#include "kernels.h"

namespace tmi {

constexpr int kInnerMaxN = 10;                                  // widest block (intrinsics)
constexpr int kInnerNS = kInnerMaxN * (kInnerMaxN + 1) / 2;     // packed upper triangle
constexpr int kInnerPart = 1 + kInnerMaxN + kInnerNS + 1;       // per-view partial: cost, g, H, invalid vote

// everything one set (KIND 0 extrinsics, 1 intrinsics) needs, passed by value
struct InnerSet {
  int kind;
  int nblocks;
  // static
  const int* view_block;   // [Nc] block of the view in this set or -1
  const int* blk_views_ptr;  // [nblocks+1] views of a block ...
  const int* blk_views;      //   ... listed here
  const int* blk_n;        // [nblocks] free coordinates
  const signed char* blk_cols;  // [nblocks][kInnerMaxN] parameter index of free coordinate a
  const int* blk_param;    // [nblocks] offset of the block's parameters in the array below
  const int* blk_size;     // [nblocks] number of parameters of the block (6 or the model size)
  // view-major observation index of the whole problem
  const int* vo_ptr;       // [Nc+1]
  const int* vo_e;         // element in the track-major layout (pixel, camera)
  const int* vo_lp;        // padded track index (point)
  const double* vo_xy;     // [.][2] the pixel, in the same order (streamed; obs_xy[vo_e] is a 16-byte gather per observation)
  // parameters: x lives in the candidate arrays of the outer loop (ext_c / intr_c / pts_c)
  double* x;               // ext_c (kind 0) or intr_c (kind 1)
  double* xc;              // candidate of the sub-problems, same indexing
  const double* x0;        // value at the start of the set (restored when a block FAILS)
  // per-view partials and per-block state
  double* part;            // [Nc][kInnerPart]; the last slot counts residuals that could not be
                           //   evaluated (a double so that the all-reduce can sum it)
  double* H;               // [nblocks][kInnerNS] unscaled J^T J
  double* g;               // [nblocks][kInnerMaxN] unscaled J^T r
  double* scale;           // [nblocks][kInnerMaxN]
  double* st_d;            // [nblocks][8] cost, radius, decrease_factor, x_norm, mcc, step_norm
  int* st_i;               // [nblocks][8] iter, invalid_run, done, need_lin, term, fresh, started
  int* active;             // device counter of blocks still running
  int loss_type;
  double loss_width;
};
enum { ISD_COST = 0, ISD_RADIUS = 1, ISD_DF = 2, ISD_XNORM = 3, ISD_MCC = 4, ISD_STEP = 5 };
enum { ISI_ITER = 0, ISI_INVALID = 1, ISI_DONE = 2, ISI_NEED_LIN = 3, ISI_TERM = 4, ISI_FRESH = 5 };

// ---- evaluation: one workgroup per view ----------------------------------------------
// UMODEL >= 0: every camera of the problem has this camera model (DeviceView::uniform_pinhole_default: PINHOLE) -- the
// five-way model switch folds at compile time; the generic Jacobian passes need all 256 registers (one wavefront per SIMD)
// NFREE > 0: no block of the set has more free coordinates (the default mask of a PINHOLE view frees f, k1, k2: 3 -- ten
// accumulators per thread instead of 66 in the intrinsics Jacobian pass)
template <int KIND, bool JAC, int UMODEL = -1, int NFREE = 0>
__global__ __launch_bounds__(256) void inner_eval_kernel(DeviceView v, InnerSet S) {
  constexpr int NMAX = NFREE > 0 ? NFREE : (KIND == 0 ? 6 : kInnerMaxN);
  constexpr int NSX = NMAX * (NMAX + 1) / 2;
  constexpr int NV = JAC ? 1 + NMAX + NSX : 1;
  const int cam = blockIdx.x;
  const int b = S.view_block[cam];
  if (b < 0) return;
  if (S.st_i[b * 8 + ISI_DONE]) return;
  if (JAC && !S.st_i[b * 8 + ISI_NEED_LIN]) return;
  if (!JAC && !(S.st_d[b * 8 + ISD_MCC] > 0.0)) return;  // the step was invalid: no candidate
  const int n = S.blk_n[b];
  const int grp = v.cam_grp[cam];
  const int model = UMODEL >= 0 ? UMODEL : v.grp_model[grp];
  const int nk = v.grp_off[grp + 1] - v.grp_off[grp];
  // the block's parameters come from x (linearisation point) or xc (candidate); everything
  // else from the outer candidate arrays
  double Kv[10], E[6];
  {
    const double* src = JAC ? S.x : S.xc;
    const double* Ep = (KIND == 0) ? src + (size_t)cam * 6 : v.ext_c + (size_t)cam * 6;
    const double* Kp = (KIND == 1) ? src + v.grp_off[grp] : v.intr_c + v.grp_off[grp];
#pragma unroll
    for (int i = 0; i < 6; ++i) E[i] = Ep[i];
#pragma unroll
    for (int i = 0; i < 10; ++i) Kv[i] = (i < nk) ? Kp[i] : 0.0;
  }
  // the view's camera is fixed over this pass: its prepared record (camera_models.h: R, C, K and the
  // left Jacobian of SO(3), unit column scales) is built once per workgroup instead of running
  // Rodrigues' formula per observation
  __shared__ double Pcam[kPrepStride];
  if (threadIdx.x == 0) {
    double ones[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ones[i] = 1.0;
    prepare_camera_record(E, Kv, 10, ones, Pcam);
  }
  __syncthreads();
  signed char cols[NMAX];
#pragma unroll
  for (int a = 0; a < NMAX; ++a) cols[a] = (a < n) ? S.blk_cols[b * kInnerMaxN + a] : (signed char)0;
  double acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.0;
  bool bad = false;
  for (int q = S.vo_ptr[cam] + threadIdx.x; q < S.vo_ptr[cam + 1]; q += 256) {
    const int lp = S.vo_lp[q];
    const double2 xy = *reinterpret_cast<const double2*>(S.vo_xy + 2 * (size_t)q);
    double X[4];
    {
      const double2 a = *reinterpret_cast<const double2*>(v.pts_c + (size_t)lp * 4);
      const double2 b = *reinterpret_cast<const double2*>(v.pts_c + (size_t)lp * 4 + 2);
      X[0] = a.x; X[1] = a.y; X[2] = b.x; X[3] = b.y;
    }
    double r[2], Jext[2][6], Jint[2][10], Jpt[2][4];
    const bool ok = reprojection_error_prepared<JAC, double>(model, Pcam, X, xy.x, xy.y, r, Jext, Jint, Jpt);
    if (!ok) {
      bad = true;
      continue;
    }
    const double sq = r[0] * r[0] + r[1] * r[1];
    double sqrt_rho1 = 1.0, asn = 0.0, rscale = 1.0;
    if (S.loss_type != 0) {
      double rho[3];
      loss_eval(S.loss_type, S.loss_width, sq, rho);
      acc[0] += 0.5 * rho[0];
      if (JAC) {
        sqrt_rho1 = sqrt(rho[1]);
        rscale = sqrt_rho1;
        if (!(sq == 0.0 || rho[2] <= 0.0)) {
          const double Dd = 1.0 + 2.0 * sq * rho[2] / rho[1];
          const double alpha = 1.0 - sqrt(Dd);
          rscale = sqrt_rho1 / (1.0 - alpha);
          asn = alpha / sq;
        }
      }
    } else {
      acc[0] += 0.5 * sq;
    }
    if (JAC) {
      double J0[NMAX], J1[NMAX];
#pragma unroll
      for (int a = 0; a < NMAX; ++a) {
        double j0 = 0.0, j1 = 0.0;
        if (a < n) {
          const int c = cols[a];
          // select without dynamic register indexing
#pragma unroll
          for (int cc = 0; cc < (KIND == 0 ? 6 : 10); ++cc)
            if (cc == c) {
              j0 = (KIND == 0) ? Jext[0][cc < 6 ? cc : 0] : Jint[0][cc];
              j1 = (KIND == 0) ? Jext[1][cc < 6 ? cc : 0] : Jint[1][cc];
            }
          if (S.loss_type != 0) {
            const double rtj = j0 * r[0] + j1 * r[1];
            j0 = sqrt_rho1 * (j0 - asn * r[0] * rtj);
            j1 = sqrt_rho1 * (j1 - asn * r[1] * rtj);
          }
        }
        J0[a] = j0;
        J1[a] = j1;
      }
      const double r0 = r[0] * rscale, r1 = r[1] * rscale;
#pragma unroll
      for (int a = 0; a < NMAX; ++a) {
        acc[1 + a] += J0[a] * r0 + J1[a] * r1;
#pragma unroll
        for (int c2 = a; c2 < NMAX; ++c2) acc[1 + NMAX + sym_idx(a, c2, NMAX)] += J0[a] * J0[c2] + J1[a] * J1[c2];
      }
    }
  }
  // workgroup reduction in a fixed order
  __shared__ double sh[4][NV];
  __shared__ int shbad[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double t = wave_sum(acc[i]);
    if (lane == 0) sh[w][i] = t;
  }
  const int wb = __any(bad) ? 1 : 0;
  if (lane == 0) shbad[w] = wb;
  __syncthreads();
  double* out = S.part + (size_t)cam * kInnerPart;
  for (int i = threadIdx.x; i < NV; i += 256) {
    const double t = sh[0][i] + sh[1][i] + sh[2][i] + sh[3][i];
    // store in the kInnerMaxN-wide layout whatever NMAX is
    int dst;
    if (i == 0) {
      dst = 0;
    } else if (i < 1 + NMAX) {
      dst = i;
    } else {
      // (a, c2) of the NMAX-packed triangle -> kInnerMaxN-packed triangle
      int idx = i - 1 - NMAX, a = 0;
      while (idx >= NMAX - a) {
        idx -= NMAX - a;
        ++a;
      }
      dst = 1 + kInnerMaxN + sym_idx(a, a + idx, kInnerMaxN);
    }
    out[dst] = t;
  }
  if (threadIdx.x == 0) out[kInnerPart - 1] = (shbad[0] | shbad[1] | shbad[2] | shbad[3]) ? 1.0 : 0.0;
}

// ---- per-block helpers (lane 0 of the block's wave) --------------------------------------
__device__ __forceinline__ double inner_x_norm(const double* x, int size) {
  double s = 0.0;
  for (int i = 0; i < size; ++i) s += x[i] * x[i];
  return sqrt(s);
}

__device__ __forceinline__ void inner_finish(const InnerSet& S, int b, int term) {
  S.st_i[b * 8 + ISI_TERM] = term;
  S.st_i[b * 8 + ISI_DONE] = 1;
  if (term >= 2) {  // FAILURE / evaluation failure: the block keeps its starting value
    const int off = S.blk_param[b], size = S.blk_size[b];
    for (int i = 0; i < size; ++i) S.x[off + i] = S.x0[off + i];
  }
}

// one wave per block
__global__ __launch_bounds__(64) void inner_step_kernel(InnerSet S) {
  __shared__ double tot[kInnerPart];
  __shared__ int anybad;
  const int b = blockIdx.x;
  const int lane = threadIdx.x;
  int* si = S.st_i + b * 8;
  double* sd = S.st_d + b * 8;
  if (si[ISI_DONE]) return;
  const int n = S.blk_n[b];
  const int off = S.blk_param[b], size = S.blk_size[b];
  double* Hb = S.H + (size_t)b * kInnerNS;
  double* gb = S.g + (size_t)b * kInnerMaxN;
  double* sc = S.scale + (size_t)b * kInnerMaxN;
  if (si[ISI_NEED_LIN]) {
    // sum the per-view partials of the linearisation (fixed order: the block's view list)
    if (lane == 0) anybad = 0;
    __syncthreads();
    for (int i = lane; i < kInnerPart; i += 64) {
      double t = 0.0;
      for (int q = S.blk_views_ptr[b]; q < S.blk_views_ptr[b + 1]; ++q) t += S.part[(size_t)S.blk_views[q] * kInnerPart + i];
      tot[i] = t;
    }
    __syncthreads();
    if (lane == 0 && tot[kInnerPart - 1] > 0.0) anybad = 1;
    __syncthreads();
  }
  if (lane != 0) return;
  if (si[ISI_NEED_LIN]) {
    si[ISI_NEED_LIN] = 0;
    if (anybad) {
      // only possible at the start point (a candidate with an invalid residual is never
      // accepted): "residual evaluation failed" -> the block is left alone
      inner_finish(S, b, 3);
      return;
    }
    sd[ISD_COST] = tot[0];
    for (int a = 0; a < kInnerMaxN; ++a) gb[a] = tot[1 + a];
    for (int i = 0; i < kInnerNS; ++i) Hb[i] = tot[1 + kInnerMaxN + i];
    if (si[ISI_FRESH]) {
      // iteration zero: Jacobi scaling from the unscaled column norms; |x|
      si[ISI_FRESH] = 0;
      for (int a = 0; a < n; ++a) sc[a] = 1.0 / (1.0 + sqrt(Hb[sym_idx(a, a, kInnerMaxN)]));
      sd[ISD_XNORM] = inner_x_norm(S.x + off, size);
    }
    // gradient tolerance on max |J^T r| of the unscaled problem
    double gmax = 0.0;
    for (int a = 0; a < n; ++a) gmax = fmax(gmax, fabs(gb[a]));
    if (gmax <= 1e-10) {
      inner_finish(S, b, 0);
      return;
    }
  }
  // ---- one trust-region step ---------------------------------------------------------
  if (si[ISI_ITER] >= 50) {
    inner_finish(S, b, 1);
    return;
  }
  si[ISI_ITER]++;
  const double radius = sd[ISD_RADIUS];
  double A[kInnerMaxN][kInnerMaxN], gs[kInnerMaxN], y[kInnerMaxN];
  for (int a = 0; a < n; ++a) {
    gs[a] = gb[a] * sc[a];
    for (int c = a; c < n; ++c) A[a][c] = A[c][a] = Hb[sym_idx(a, c, kInnerMaxN)] * sc[a] * sc[c];
  }
  double Hs_diag[kInnerMaxN];
  for (int a = 0; a < n; ++a) {
    Hs_diag[a] = A[a][a];
    A[a][a] += fmin(fmax(A[a][a], 1e-6), 1e32) / radius;
  }
  // Cholesky A = L L^T in place (lower), then two triangular solves
  bool ok = true;
  for (int j = 0; j < n && ok; ++j) {
    double d = A[j][j];
    for (int m = 0; m < j; ++m) d -= A[j][m] * A[j][m];
    if (!(d > 0.0)) {
      ok = false;
      break;
    }
    const double l = sqrt(d);
    A[j][j] = l;
    for (int i = j + 1; i < n; ++i) {
      double t = A[i][j];
      for (int m = 0; m < j; ++m) t -= A[i][m] * A[j][m];
      A[i][j] = t / l;
    }
  }
  double mcc = 0.0;
  if (ok) {
    double z[kInnerMaxN];
    for (int i = 0; i < n; ++i) {
      double t = gs[i];
      for (int m = 0; m < i; ++m) t -= A[i][m] * z[m];
      z[i] = t / A[i][i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double t = z[i];
      for (int m = i + 1; m < n; ++m) t -= A[m][i] * y[m];
      y[i] = t / A[i][i];
    }
    // model cost change of d = -y: y^T g - 1/2 y^T H y (scaled H without the damping)
    double yg = 0.0, yHy = 0.0;
    for (int a = 0; a < n; ++a) {
      yg += y[a] * gs[a];
      double t = 0.0;
      for (int c = 0; c < n; ++c) {
        const double h = (a == c) ? Hs_diag[a] : Hb[a < c ? sym_idx(a, c, kInnerMaxN) : sym_idx(c, a, kInnerMaxN)] * sc[a] * sc[c];
        t += h * y[c];
      }
      yHy += y[a] * t;
    }
    mcc = yg - 0.5 * yHy;
    if (!(mcc > 0.0)) ok = false;
  }
  if (!ok) {  // HandleInvalidStep
    if (++si[ISI_INVALID] >= 5) {
      inner_finish(S, b, 2);
      return;
    }
    sd[ISD_RADIUS] = radius / sd[ISD_DF];
    sd[ISD_DF] *= 2.0;
    if (sd[ISD_RADIUS] < 1e-32) inner_finish(S, b, 0);
    sd[ISD_MCC] = -1.0;  // tells inner_decide there is no candidate
    return;
  }
  si[ISI_INVALID] = 0;
  for (int i = 0; i < size; ++i) S.xc[off + i] = S.x[off + i];
  double step_sq = 0.0;
  for (int a = 0; a < n; ++a) {
    const double d = -y[a] * sc[a];
    S.xc[off + S.blk_cols[b * kInnerMaxN + a]] += d;
    step_sq += d * d;
  }
  sd[ISD_MCC] = mcc;
  sd[ISD_STEP] = sqrt(step_sq);
}

__global__ __launch_bounds__(64) void inner_decide_kernel(InnerSet S) {
  const int b = blockIdx.x;
  int* si = S.st_i + b * 8;
  double* sd = S.st_d + b * 8;
  if (si[ISI_DONE]) return;
  if (threadIdx.x != 0) return;
  if (!(sd[ISD_MCC] > 0.0)) {  // the step was invalid: nothing to decide
    atomicAdd(S.active, 1);
    return;
  }
  const int off = S.blk_param[b], size = S.blk_size[b];
  double cand_cost = 0.0;
  bool bad = false;
  for (int q = S.blk_views_ptr[b]; q < S.blk_views_ptr[b + 1]; ++q) {
    cand_cost += S.part[(size_t)S.blk_views[q] * kInnerPart];
    bad = bad || S.part[(size_t)S.blk_views[q] * kInnerPart + kInnerPart - 1] > 0.0;
  }
  if (bad) cand_cost = 1.7976931348623157e308;
  const double cost = sd[ISD_COST];
  if (sd[ISD_STEP] <= 1e-8 * (sd[ISD_XNORM] + 1e-8)) {  // parameter tolerance: candidate dropped
    inner_finish(S, b, 0);
    return;
  }
  const double cost_change = cost - cand_cost;
  if (fabs(cost_change) <= 1e-6 * cost) {  // function tolerance: candidate dropped
    inner_finish(S, b, 0);
    return;
  }
  const double relative_decrease = cost_change / sd[ISD_MCC];
  if (relative_decrease > 1e-3) {
    for (int i = 0; i < size; ++i) S.x[off + i] = S.xc[off + i];
    sd[ISD_XNORM] = inner_x_norm(S.x + off, size);
    sd[ISD_RADIUS] = fmin(1e16, sd[ISD_RADIUS] / fmax(1.0 / 3.0, 1.0 - pow(2.0 * relative_decrease - 1.0, 3.0)));
    sd[ISD_DF] = 2.0;
    si[ISI_NEED_LIN] = 1;  // the next eval pass re-linearises (cost, gradient test, H, g)
  } else {
    sd[ISD_RADIUS] /= sd[ISD_DF];
    sd[ISD_DF] *= 2.0;
  }
  if (sd[ISD_RADIUS] < 1e-32) {
    inner_finish(S, b, 0);
    return;
  }
  atomicAdd(S.active, 1);
}

__global__ void inner_init_kernel(InnerSet S) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= S.nblocks) return;
  int* si = S.st_i + b * 8;
  double* sd = S.st_d + b * 8;
  for (int i = 0; i < 8; ++i) {
    si[i] = 0;
    sd[i] = 0.0;
  }
  sd[ISD_RADIUS] = 1e4;
  sd[ISD_DF] = 2.0;
  si[ISI_NEED_LIN] = 1;
  si[ISI_FRESH] = 1;
  // (a block nobody observes has a zero gradient and stops at its first gradient test; the
  // observation counts are per rank and cannot decide that here)
  if (S.blk_n[b] == 0) si[ISI_DONE] = 1;
}

// |a - b|^2 over n doubles (one workgroup, fixed order): the step norm of an LM iteration that
// ran an inner sweep is measured on the parameters themselves
__global__ __launch_bounds__(1024) void diff_sq_kernel(const double* __restrict__ a, const double* __restrict__ b,
                                                       long long n, double* __restrict__ out) {
  __shared__ double sh[1024];
  double acc = 0.0;
  for (long long i = threadIdx.x; i < n; i += 1024) {
    const double d = a[i] - b[i];
    acc += d * d;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sh[0];
}

// per-track partial sums [ |pts - pts_c|^2 , |pts_c|^2 over the live tracks ] (cf. update_points)
__global__ __launch_bounds__(256) void points_diff_kernel(DeviceView v, int nblocks, double* partial) {
  const int lp = blockIdx.x * 256 + threadIdx.x;
  double acc[2] = {0.0, 0.0};
  if (lp < v.Np_pad) {
    const bool live = v.pt_k[lp] > 0 && !v.pt_const[lp];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const double x = v.pts_c[(size_t)lp * 4 + a];
      const double d = x - v.pts[(size_t)lp * 4 + a];
      acc[0] += d * d;
      if (live) acc[1] += x * x;
    }
  }
  block_sum_store<2>(acc, partial, nblocks);
}

// |x|^2 over every coordinate of every non-constant camera-side block of the candidate
// (the second half of update_cameras_kernel, for a candidate the inner sweep has moved)
__global__ __launch_bounds__(1024) void cameras_norm_kernel(DeviceView v, double* out) {
  __shared__ double sh[16];
  double xn = 0.0;
  for (int rb = threadIdx.x; rb < v.Nrb; rb += 1024) {
    const int cam = v.rb_cam[rb];
    bool intr = true;  // a shared intrinsics block
    if (cam >= 0) {
      const unsigned m = v.cam_mask[cam];
      if (m & 0x3f)
        for (int a = 0; a < 6; ++a) xn += v.ext_c[(size_t)cam * 6 + a] * v.ext_c[(size_t)cam * 6 + a];
      intr = (m >> 6) != 0;
    }
    if (intr) {
      const int g = v.rb_grp[rb];
      for (int a = v.grp_off[g]; a < v.grp_off[g + 1]; ++a) xn += v.intr_c[a] * v.intr_c[a];
    }
  }
  const double x2 = block1024_sum(xn, sh);
  if (threadIdx.x == 0) out[0] = x2;
}

}  // namespace tmi
