// Device-side reprojection residual and ANALYTIC Jacobians for the five
// TheiaSfM camera models (fp64, gfx950).
//
// What is computed follows the reference functor and models
//   ReprojectionError<M>::operator()   src/theia/sfm/camera/reprojection_error.h:51-95
//   PinholeCameraModel                 pinhole_camera_model.h:181-210,241-257
//   PinholeRadialTangentialCameraModel pinhole_radial_tangential_camera_model.h:190-219,250-291
//   FisheyeCameraModel                 fisheye_camera_model.h:162-187,223-267
//   FOVCameraModel                     fov_camera_model.h:155-182,211-260
//   DivisionUndistortionCameraModel    division_undistortion_camera_model.h:172-202,256-289
// but where the reference differentiates with ceres::Jet dual numbers
// (create_reprojection_error_cost_function.h:60-89) this file carries closed
// form derivatives, switching on the same value predicates the Jets would
// (theta^2 > DBL_EPSILON; fisheye r^2 < 1e-8, z < 0; FOV omega < 1e-3,
// r^2 < 1e-3; division |2 k r^2| < eps or 1 - 4 k r^2 < 0).  The oracle's
// dual numbers (oracle/jet.h) are the independent check.
//
// All small arrays are indexed with compile-time constants after unrolling so
// they live in VGPRs (a runtime-indexed array would be demoted to scratch).
#pragma once
#include <hip/hip_runtime.h>

namespace tmi {

constexpr double kDblEpsilon = 2.220446049250313e-16;

// Intrinsics layouts (include/theia_mi355_ba.h):
//  0 PINHOLE [f ar s px py k1 k2]            1 RADTAN [f ar s px py k1 k2 k3 t1 t2]
//  2 FISHEYE [f ar s px py k1 k2 k3 k4]      3 FOV [f ar px py w]   4 DIVISION [f ar px py k]

// Rodrigues rotation q = R(w) a, as ceres::AngleAxisRotatePoint executes it
// (call site reprojection_error.h:81-83).  If JAC: R (= dq/da) and dq/dw.
template <bool JAC>
__device__ __forceinline__ void rotate_point(const double w[3], const double a[3], double q[3],
                                             double R[3][3], double dqdw[3][3]) {
  const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const double wxa[3] = {w[1] * a[2] - w[2] * a[1], w[2] * a[0] - w[0] * a[2],
                         w[0] * a[1] - w[1] * a[0]};
  if (theta2 > kDblEpsilon) {
    const double theta = sqrt(theta2);
    double s, c;
    sincos(theta, &s, &c);
    const double inv_theta = 1.0 / theta;
    const double A1 = s * inv_theta;                 // sin(t)/t
    const double B1 = (1.0 - c) * inv_theta * inv_theta;  // (1-cos t)/t^2
    const double wa = w[0] * a[0] + w[1] * a[1] + w[2] * a[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] = a[i] * c + wxa[i] * A1 + w[i] * (wa * B1);
    if (JAC) {
      // R = c I + A1 [w]x + B1 w w^T
      R[0][0] = c + B1 * w[0] * w[0];
      R[0][1] = -A1 * w[2] + B1 * w[0] * w[1];
      R[0][2] = A1 * w[1] + B1 * w[0] * w[2];
      R[1][0] = A1 * w[2] + B1 * w[1] * w[0];
      R[1][1] = c + B1 * w[1] * w[1];
      R[1][2] = -A1 * w[0] + B1 * w[1] * w[2];
      R[2][0] = -A1 * w[1] + B1 * w[2] * w[0];
      R[2][1] = A1 * w[0] + B1 * w[2] * w[1];
      R[2][2] = c + B1 * w[2] * w[2];
      const double dA1 = (c * theta - s) * inv_theta * inv_theta;  // d(sin t / t)/dt
      const double dB1 = (s * theta - 2.0 * (1.0 - c)) * inv_theta * inv_theta * inv_theta;
      // e_k x a
      const double exa[3][3] = {{0.0, -a[2], a[1]}, {a[2], 0.0, -a[0]}, {-a[1], a[0], 0.0}};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double wk = w[k] * inv_theta;  // d theta / d w_k
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          double v = a[i] * (-s * wk) + exa[k][i] * A1 + wxa[i] * (dA1 * wk) +
                     w[i] * (a[k] * B1 + wa * dB1 * wk);
          if (i == k) v += wa * B1;
          dqdw[i][k] = v;
        }
      }
    }
  } else {
    // first-order branch: q = a + w x a
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] = a[i] + wxa[i];
    if (JAC) {
      R[0][0] = 1.0;   R[0][1] = -w[2]; R[0][2] = w[1];
      R[1][0] = w[2];  R[1][1] = 1.0;   R[1][2] = -w[0];
      R[2][0] = -w[1]; R[2][1] = w[0];  R[2][2] = 1.0;
      dqdw[0][0] = 0.0;   dqdw[0][1] = a[2];  dqdw[0][2] = -a[1];
      dqdw[1][0] = -a[2]; dqdw[1][1] = 0.0;   dqdw[1][2] = a[0];
      dqdw[2][0] = a[1];  dqdw[2][1] = -a[0]; dqdw[2][2] = 0.0;
    }
  }
}

// pixel = K-matrix-with-skew applied to the distorted point d, and its chain
// rule pieces (pinhole_camera_model.h:206-209 and the two siblings).
// dd_dq: d(distorted)/dq [2][3]; dd_dk: d(distorted)/d(distortion params).
template <bool JAC, int NDIST>
__device__ __forceinline__ void apply_k_skew(const double* K, const double d[2], double px[2],
                                             const double dd_dq[2][3], const double dd_dk[2][NDIST],
                                             double dpdq[2][3], double dpdK[2][10]) {
  const double f = K[0], ar = K[1], sk = K[2];
  px[0] = f * d[0] + sk * d[1] + K[3];
  px[1] = f * ar * d[1] + K[4];
  if (JAC) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dpdq[0][j] = f * dd_dq[0][j] + sk * dd_dq[1][j];
      dpdq[1][j] = f * ar * dd_dq[1][j];
    }
    dpdK[0][0] = d[0];  dpdK[1][0] = ar * d[1];  // f
    dpdK[0][1] = 0.0;   dpdK[1][1] = f * d[1];   // aspect ratio
    dpdK[0][2] = d[1];  dpdK[1][2] = 0.0;        // skew
    dpdK[0][3] = 1.0;   dpdK[1][3] = 0.0;        // px
    dpdK[0][4] = 0.0;   dpdK[1][4] = 1.0;        // py
#pragma unroll
    for (int j = 0; j < NDIST; ++j) {
      dpdK[0][5 + j] = f * dd_dk[0][j] + sk * dd_dk[1][j];
      dpdK[1][5 + j] = f * ar * dd_dk[1][j];
    }
#pragma unroll
    for (int j = 5 + NDIST; j < 10; ++j) { dpdK[0][j] = 0.0; dpdK[1][j] = 0.0; }
  }
}

// n = q.xy / q.z and dn/dq
template <bool JAC>
__device__ __forceinline__ void normalize_point(const double q[3], double n[2], double dn[2][3]) {
  const double iz = 1.0 / q[2];
  n[0] = q[0] * iz;
  n[1] = q[1] * iz;
  if (JAC) {
    dn[0][0] = iz;  dn[0][1] = 0.0; dn[0][2] = -n[0] * iz;
    dn[1][0] = 0.0; dn[1][1] = iz;  dn[1][2] = -n[1] * iz;
  }
}

// dd_dq = dd_dn (2x2) * dn_dq (2x3)
__device__ __forceinline__ void chain_2x2_2x3(const double ddn[2][2], const double dn[2][3],
                                              double out[2][3]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) out[i][j] = ddn[i][0] * dn[0][j] + ddn[i][1] * dn[1][j];
}

template <bool JAC>
__device__ __forceinline__ void project_pinhole(const double* K, const double q[3], double px[2],
                                                double dpdq[2][3], double dpdK[2][10]) {
  double n[2], dn[2][3];
  normalize_point<JAC>(q, n, dn);
  const double r2 = n[0] * n[0] + n[1] * n[1];
  const double dd = 1.0 + r2 * (K[5] + K[6] * r2);
  const double d[2] = {n[0] * dd, n[1] * dd};
  double dd_dq[2][3], dd_dk[2][2];
  if (JAC) {
    const double g = 2.0 * (K[5] + 2.0 * K[6] * r2);  // d(dd)/d(r2) * 2
    const double ddn[2][2] = {{dd + n[0] * n[0] * g, n[0] * n[1] * g},
                              {n[1] * n[0] * g, dd + n[1] * n[1] * g}};
    chain_2x2_2x3(ddn, dn, dd_dq);
    dd_dk[0][0] = n[0] * r2;      dd_dk[1][0] = n[1] * r2;
    dd_dk[0][1] = n[0] * r2 * r2; dd_dk[1][1] = n[1] * r2 * r2;
  }
  apply_k_skew<JAC, 2>(K, d, px, dd_dq, dd_dk, dpdq, dpdK);
}

template <bool JAC>
__device__ __forceinline__ void project_radtan(const double* K, const double q[3], double px[2],
                                               double dpdq[2][3], double dpdK[2][10]) {
  double n[2], dn[2][3];
  normalize_point<JAC>(q, n, dn);
  const double x = n[0], y = n[1];
  const double r2 = x * x + y * y;
  const double k1 = K[5], k2 = K[6], k3 = K[7], t1 = K[8], t2 = K[9];
  const double rd = 1.0 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
  const double tx = t2 * (r2 + 2.0 * x * x) + 2.0 * t1 * x * y;
  const double ty = t1 * (r2 + 2.0 * y * y) + 2.0 * t2 * x * y;
  const double d[2] = {x * rd + tx, y * rd + ty};
  double dd_dq[2][3], dd_dk[2][5];
  if (JAC) {
    const double g = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r2 * r2;  // d rd / d r2
    const double ddn[2][2] = {
        {rd + 2.0 * x * x * g + 6.0 * t2 * x + 2.0 * t1 * y, 2.0 * x * y * g + 2.0 * t2 * y + 2.0 * t1 * x},
        {2.0 * x * y * g + 2.0 * t1 * x + 2.0 * t2 * y, rd + 2.0 * y * y * g + 6.0 * t1 * y + 2.0 * t2 * x}};
    chain_2x2_2x3(ddn, dn, dd_dq);
    dd_dk[0][0] = x * r2;           dd_dk[1][0] = y * r2;
    dd_dk[0][1] = x * r2 * r2;      dd_dk[1][1] = y * r2 * r2;
    dd_dk[0][2] = x * r2 * r2 * r2; dd_dk[1][2] = y * r2 * r2 * r2;
    dd_dk[0][3] = 2.0 * x * y;      dd_dk[1][3] = r2 + 2.0 * y * y;  // t1
    dd_dk[0][4] = r2 + 2.0 * x * x; dd_dk[1][4] = 2.0 * x * y;       // t2
  }
  apply_k_skew<JAC, 5>(K, d, px, dd_dq, dd_dk, dpdq, dpdK);
}

template <bool JAC>
__device__ __forceinline__ void project_fisheye(const double* K, const double q[3], double px[2],
                                                double dpdq[2][3], double dpdK[2][10]) {
  const double x = q[0], y = q[1], z = q[2];
  const double r2 = x * x + y * y;
  double d[2], dd_dq[2][3], dd_dk[2][4];
  if (r2 < 1e-8) {  // fisheye_camera_model.h:243: pass the raw x, y through
    d[0] = x;
    d[1] = y;
    if (JAC) {
      dd_dq[0][0] = 1.0; dd_dq[0][1] = 0.0; dd_dq[0][2] = 0.0;
      dd_dq[1][0] = 0.0; dd_dq[1][1] = 1.0; dd_dq[1][2] = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) { dd_dk[0][j] = 0.0; dd_dk[1][j] = 0.0; }
    }
  } else {
    const double r = sqrt(r2);
    const double az = fabs(z);
    const double th = atan2(r, az);
    const double t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const double poly = 1.0 + K[5] * t2 + K[6] * t4 + K[7] * t6 + K[8] * t8;
    const double thd = th * poly;
    const double sgn = (z < 0.0) ? -1.0 : 1.0;  // :263
    const double h = sgn * thd / r;
    d[0] = h * x;
    d[1] = h * y;
    if (JAC) {
      const double inv = 1.0 / (r2 + z * z);
      const double dth_dr = az * inv;
      const double dth_dz = -r * inv * ((z < 0.0) ? -1.0 : 1.0);  // d|z|/dz, Jet abs: z<0 ? -1 : +1
      const double dthd = 1.0 + 3.0 * K[5] * t2 + 5.0 * K[6] * t4 + 7.0 * K[7] * t6 + 9.0 * K[8] * t8;
      // h = sgn * thd(th(r,z)) / r
      const double dh_dr = sgn * (dthd * dth_dr / r - thd / r2);
      const double dh_dz = sgn * dthd * dth_dz / r;
      const double dr_dx = x / r, dr_dy = y / r;
      dd_dq[0][0] = h + x * dh_dr * dr_dx; dd_dq[0][1] = x * dh_dr * dr_dy; dd_dq[0][2] = x * dh_dz;
      dd_dq[1][0] = y * dh_dr * dr_dx; dd_dq[1][1] = h + y * dh_dr * dr_dy; dd_dq[1][2] = y * dh_dz;
      const double base = sgn * th / r;
      dd_dk[0][0] = base * t2 * x; dd_dk[1][0] = base * t2 * y;
      dd_dk[0][1] = base * t4 * x; dd_dk[1][1] = base * t4 * y;
      dd_dk[0][2] = base * t6 * x; dd_dk[1][2] = base * t6 * y;
      dd_dk[0][3] = base * t8 * x; dd_dk[1][3] = base * t8 * y;
    }
  }
  apply_k_skew<JAC, 4>(K, d, px, dd_dq, dd_dk, dpdq, dpdK);
}

template <bool JAC>
__device__ __forceinline__ void project_fov(const double* K, const double q[3], double px[2],
                                            double dpdq[2][3], double dpdK[2][10]) {
  double n[2], dn[2][3];
  normalize_point<JAC>(q, n, dn);
  const double f = K[0], ar = K[1], om = K[4];
  const double r2 = n[0] * n[0] + n[1] * n[1];
  double rd, drd_dr2 = 0.0, drd_dom = 0.0;
  if (om < 1e-3) {  // fov_camera_model.h:227
    rd = (om * om * r2) / 3.0 - om * om / 12.0 + 1.0;
    if (JAC) {
      drd_dr2 = om * om / 3.0;
      drd_dom = 2.0 * om * r2 / 3.0 - om / 6.0;
    }
  } else if (r2 < 1e-3) {  // :236
    const double T = tan(om / 2.0);
    const double num = -2.0 * T * (4.0 * r2 * T * T - 3.0);
    rd = num / (3.0 * om);
    if (JAC) {
      drd_dr2 = -8.0 * T * T * T / (3.0 * om);
      const double dT = 0.5 * (1.0 + T * T);
      const double dnum = (-24.0 * r2 * T * T + 6.0) * dT;
      drd_dom = dnum / (3.0 * om) - num / (3.0 * om * om);
    }
  } else {  // :249-254
    const double ru = sqrt(r2);
    const double T = tan(om / 2.0);
    const double m = 2.0 * ru * T;
    const double at = atan(m);
    rd = at / (ru * om);
    if (JAC) {
      const double im = 1.0 / (1.0 + m * m);
      const double drd_dru = (2.0 * T * im) / (ru * om) - at / (r2 * om);
      drd_dr2 = drd_dru / (2.0 * ru);
      const double dT = 0.5 * (1.0 + T * T);
      drd_dom = (2.0 * ru * dT * im) / (ru * om) - at / (ru * om * om);
    }
  }
  const double d[2] = {rd * n[0], rd * n[1]};
  px[0] = f * d[0] + K[2];
  px[1] = f * ar * d[1] + K[3];
  if (JAC) {
    const double g = 2.0 * drd_dr2;
    const double ddn[2][2] = {{rd + n[0] * n[0] * g, n[0] * n[1] * g},
                              {n[1] * n[0] * g, rd + n[1] * n[1] * g}};
    double dd_dq[2][3];
    chain_2x2_2x3(ddn, dn, dd_dq);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dpdq[0][j] = f * dd_dq[0][j];
      dpdq[1][j] = f * ar * dd_dq[1][j];
    }
    dpdK[0][0] = d[0]; dpdK[1][0] = ar * d[1];
    dpdK[0][1] = 0.0;  dpdK[1][1] = f * d[1];
    dpdK[0][2] = 1.0;  dpdK[1][2] = 0.0;
    dpdK[0][3] = 0.0;  dpdK[1][3] = 1.0;
    dpdK[0][4] = f * n[0] * drd_dom; dpdK[1][4] = f * ar * n[1] * drd_dom;
#pragma unroll
    for (int j = 5; j < 10; ++j) { dpdK[0][j] = 0.0; dpdK[1][j] = 0.0; }
  }
}

template <bool JAC>
__device__ __forceinline__ void project_division(const double* K, const double q[3], double px[2],
                                                 double dpdq[2][3], double dpdK[2][10]) {
  double n[2], dn[2][3];
  normalize_point<JAC>(q, n, dn);
  const double f = K[0], ar = K[1], k = K[4];
  const double fy = f * ar;
  const double u[2] = {f * n[0], fy * n[1]};
  const double r2 = u[0] * u[0] + u[1] * u[1];
  const double denom = 2.0 * k * r2;
  const double inner = 1.0 - 4.0 * k * r2;
  double scale = 1.0, dsc_dr2 = 0.0, dsc_dk = 0.0;
  if (!(fabs(denom) < kDblEpsilon || inner < 0.0)) {  // division_undistortion_camera_model.h:281
    const double sq = sqrt(inner);
    scale = (1.0 - sq) / denom;
    if (JAC) {
      dsc_dr2 = 1.0 / (sq * r2) - scale / r2;
      dsc_dk = 1.0 / (k * sq) - scale / k;
    }
  }
  px[0] = u[0] * scale + K[2];
  px[1] = u[1] * scale + K[3];
  if (JAC) {
    const double g = 2.0 * dsc_dr2;
    // d(distorted)/du
    const double ddu[2][2] = {{scale + u[0] * u[0] * g, u[0] * u[1] * g},
                              {u[1] * u[0] * g, scale + u[1] * u[1] * g}};
    // du/dq = diag(f, fy) dn/dq
    double du_dq[2][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { du_dq[0][j] = f * dn[0][j]; du_dq[1][j] = fy * dn[1][j]; }
    chain_2x2_2x3(ddu, du_dq, dpdq);
    // d/df: du/df = (n0, ar n1); d/dar: du/dar = (0, f n1)
    const double duf[2] = {n[0], ar * n[1]};
    const double dua[2] = {0.0, f * n[1]};
    dpdK[0][0] = ddu[0][0] * duf[0] + ddu[0][1] * duf[1];
    dpdK[1][0] = ddu[1][0] * duf[0] + ddu[1][1] * duf[1];
    dpdK[0][1] = ddu[0][0] * dua[0] + ddu[0][1] * dua[1];
    dpdK[1][1] = ddu[1][0] * dua[0] + ddu[1][1] * dua[1];
    dpdK[0][2] = 1.0; dpdK[1][2] = 0.0;
    dpdK[0][3] = 0.0; dpdK[1][3] = 1.0;
    dpdK[0][4] = u[0] * dsc_dk; dpdK[1][4] = u[1] * dsc_dk;
#pragma unroll
    for (int j = 5; j < 10; ++j) { dpdK[0][j] = 0.0; dpdK[1][j] = 0.0; }
  }
}

// CreateReprojectionErrorCostFunction's dispatch
// (create_reprojection_error_cost_function.h:51-96).
template <bool JAC>
__device__ __forceinline__ void project(int model, const double* K, const double q[3], double px[2],
                                        double dpdq[2][3], double dpdK[2][10]) {
  switch (model) {
    case 0: project_pinhole<JAC>(K, q, px, dpdq, dpdK); break;
    case 1: project_radtan<JAC>(K, q, px, dpdq, dpdK); break;
    case 2: project_fisheye<JAC>(K, q, px, dpdq, dpdK); break;
    case 3: project_fov<JAC>(K, q, px, dpdq, dpdK); break;
    default: project_division<JAC>(K, q, px, dpdq, dpdK); break;
  }
}

// The full residual (reprojection_error.h:51-95).  Returns false where the
// reference functor does (|X - w C|^2 < 1e-8, :75-77).
//   Jext [2][6]  d r / d [C, angle-axis]
//   Jint [2][10] d r / d intrinsics (model order, zero padded)
//   Jpt  [2][4]  d r / d X (homogeneous)
template <bool JAC>
__device__ __forceinline__ bool reprojection_error(int model, const double* ext, const double* K,
                                                   const double* X, double fx, double fy,
                                                   double r[2], double Jext[2][6],
                                                   double Jint[2][10], double Jpt[2][4]) {
  const double w = X[3];
  const double a[3] = {X[0] - w * ext[0], X[1] - w * ext[1], X[2] - w * ext[2]};
  if (a[0] * a[0] + a[1] * a[1] + a[2] * a[2] < 1e-8) return false;
  double q[3], R[3][3], dqdw[3][3], dpdq[2][3], px[2];
  rotate_point<JAC>(ext + 3, a, q, R, dqdw);
  project<JAC>(model, K, q, px, dpdq, Jint);
  r[0] = px[0] - fx;
  r[1] = px[1] - fy;
  if (JAC) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // M = dp/dq * R   (= d r / d X[0:3])
      double M[3];
#pragma unroll
      for (int j = 0; j < 3; ++j)
        M[j] = dpdq[i][0] * R[0][j] + dpdq[i][1] * R[1][j] + dpdq[i][2] * R[2][j];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        Jpt[i][j] = M[j];
        Jext[i][j] = -w * M[j];  // da/dC = -w I
        Jext[i][3 + j] = dpdq[i][0] * dqdw[0][j] + dpdq[i][1] * dqdw[1][j] + dpdq[i][2] * dqdw[2][j];
      }
      Jpt[i][3] = -(M[0] * ext[0] + M[1] * ext[1] + M[2] * ext[2]);  // da/dw = -C
    }
  }
  return true;
}

// ceres/loss_function.cc (1.x) restated for the device: rho, rho', rho''.
__device__ __forceinline__ void loss_eval(int type, double a, double s, double rho[3]) {
  switch (type) {
    case 1: {  // Huber
      const double b = a * a;
      if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = fmax(2.2250738585072014e-308, a / r);
        rho[2] = -rho[1] / (2.0 * s);
      } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
      }
      break;
    }
    case 2: {  // SoftLOne
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c;
      const double tmp = sqrt(sum);
      rho[0] = 2.0 * b * (tmp - 1.0);
      rho[1] = fmax(2.2250738585072014e-308, 1.0 / tmp);
      rho[2] = -(c * rho[1]) / (2.0 * sum);
      break;
    }
    case 3: {  // Cauchy
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c;
      const double inv = 1.0 / sum;
      rho[0] = b * log(sum);
      rho[1] = fmax(2.2250738585072014e-308, inv);
      rho[2] = -c * (inv * inv);
      break;
    }
    case 4: {  // Arctan
      const double b = 1.0 / (a * a);
      const double sum = 1.0 + s * s * b;
      const double inv = 1.0 / sum;
      rho[0] = a * atan2(s, a);
      rho[1] = fmax(2.2250738585072014e-308, inv);
      rho[2] = -2.0 * s * b * (inv * inv);
      break;
    }
    case 5: {  // Tukey (Ceres 1.x normalisation)
      const double a2 = a * a;
      if (s <= a2) {
        const double v = 1.0 - s / a2;
        const double v2 = v * v;
        rho[0] = a2 / 6.0 * (1.0 - v2 * v);
        rho[1] = 0.5 * v2;
        rho[2] = -1.0 / a2 * v;
      } else {
        rho[0] = a2 / 6.0; rho[1] = 0.0; rho[2] = 0.0;
      }
      break;
    }
    default:
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

}  // namespace tmi
