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

// ---- scalar-type plumbing: the functions below are templated on T = double (the
// reference's precision) or float (BASELINE config 5's fp32 residual path; the camera
// translation X - w C is always removed in fp64 first, accumulation stays fp64).
__device__ __forceinline__ double t_sqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float t_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double t_atan2(double y, double x) { return atan2(y, x); }
__device__ __forceinline__ float t_atan2(float y, float x) { return atan2f(y, x); }
__device__ __forceinline__ double t_tan(double x) { return tan(x); }
__device__ __forceinline__ float t_tan(float x) { return tanf(x); }
__device__ __forceinline__ double t_atan(double x) { return atan(x); }
__device__ __forceinline__ float t_atan(float x) { return atanf(x); }
__device__ __forceinline__ double t_fabs(double x) { return fabs(x); }
__device__ __forceinline__ float t_fabs(float x) { return fabsf(x); }
__device__ __forceinline__ void t_sincos(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ void t_sincos(float x, float* s, float* c) { sincosf(x, s, c); }

// Rodrigues rotation q = R(w) a, as ceres::AngleAxisRotatePoint executes it
// (call site reprojection_error.h:81-83).  If JAC: R (= dq/da) and dq/dw.
template <bool JAC, typename T>
__device__ __forceinline__ void rotate_point(const T w[3], const T a[3], T q[3],
                                             T R[3][3], T dqdw[3][3]) {
  const T theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  const T wxa[3] = {w[1] * a[2] - w[2] * a[1], w[2] * a[0] - w[0] * a[2],
                         w[0] * a[1] - w[1] * a[0]};
  if (theta2 > kDblEpsilon) {
    const T theta = t_sqrt(theta2);
    T s, c;
    t_sincos(theta, &s, &c);
    const T inv_theta = T(1.0) / theta;
    const T A1 = s * inv_theta;                 // sin(t)/t
    const T B1 = (T(1.0) - c) * inv_theta * inv_theta;  // (1-cos t)/t^2
    const T wa = w[0] * a[0] + w[1] * a[1] + w[2] * a[2];
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
      const T dA1 = (c * theta - s) * inv_theta * inv_theta;  // d(sin t / t)/dt
      const T dB1 = (s * theta - T(2.0) * (T(1.0) - c)) * inv_theta * inv_theta * inv_theta;
      // e_k x a
      const T exa[3][3] = {{T(0.0), -a[2], a[1]}, {a[2], T(0.0), -a[0]}, {-a[1], a[0], T(0.0)}};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const T wk = w[k] * inv_theta;  // d theta / d w_k
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          T v = a[i] * (-s * wk) + exa[k][i] * A1 + wxa[i] * (dA1 * wk) +
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
      R[0][0] = T(1.0);   R[0][1] = -w[2]; R[0][2] = w[1];
      R[1][0] = w[2];  R[1][1] = T(1.0);   R[1][2] = -w[0];
      R[2][0] = -w[1]; R[2][1] = w[0];  R[2][2] = T(1.0);
      dqdw[0][0] = T(0.0);   dqdw[0][1] = a[2];  dqdw[0][2] = -a[1];
      dqdw[1][0] = -a[2]; dqdw[1][1] = T(0.0);   dqdw[1][2] = a[0];
      dqdw[2][0] = a[1];  dqdw[2][1] = -a[0]; dqdw[2][2] = T(0.0);
    }
  }
}

// pixel = K-matrix-with-skew applied to the distorted point d, and its chain
// rule pieces (pinhole_camera_model.h:206-209 and the two siblings).
// dd_dq: d(distorted)/dq [2][3]; dd_dk: d(distorted)/d(distortion params).
template <bool JAC, int NDIST, typename T>
__device__ __forceinline__ void apply_k_skew(const T* K, const T d[2], T px[2],
                                             const T dd_dq[2][3], const T dd_dk[2][NDIST],
                                             T dpdq[2][3], T dpdK[2][10]) {
  const T f = K[0], ar = K[1], sk = K[2];
  px[0] = f * d[0] + sk * d[1] + K[3];
  px[1] = f * ar * d[1] + K[4];
  if (JAC) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dpdq[0][j] = f * dd_dq[0][j] + sk * dd_dq[1][j];
      dpdq[1][j] = f * ar * dd_dq[1][j];
    }
    dpdK[0][0] = d[0];  dpdK[1][0] = ar * d[1];  // f
    dpdK[0][1] = T(0.0);   dpdK[1][1] = f * d[1];   // aspect ratio
    dpdK[0][2] = d[1];  dpdK[1][2] = T(0.0);        // skew
    dpdK[0][3] = T(1.0);   dpdK[1][3] = T(0.0);        // px
    dpdK[0][4] = T(0.0);   dpdK[1][4] = T(1.0);        // py
#pragma unroll
    for (int j = 0; j < NDIST; ++j) {
      dpdK[0][5 + j] = f * dd_dk[0][j] + sk * dd_dk[1][j];
      dpdK[1][5 + j] = f * ar * dd_dk[1][j];
    }
#pragma unroll
    for (int j = 5 + NDIST; j < 10; ++j) { dpdK[0][j] = T(0.0); dpdK[1][j] = T(0.0); }
  }
}

// n = q.xy / q.z and dn/dq
template <bool JAC, typename T>
__device__ __forceinline__ void normalize_point(const T q[3], T n[2], T dn[2][3]) {
  const T iz = T(1.0) / q[2];
  n[0] = q[0] * iz;
  n[1] = q[1] * iz;
  if (JAC) {
    dn[0][0] = iz;  dn[0][1] = T(0.0); dn[0][2] = -n[0] * iz;
    dn[1][0] = T(0.0); dn[1][1] = iz;  dn[1][2] = -n[1] * iz;
  }
}

// dd_dq = dd_dn (2x2) * dn_dq (2x3)
template <typename T>
__device__ __forceinline__ void chain_2x2_2x3(const T ddn[2][2], const T dn[2][3],
                                              T out[2][3]) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) out[i][j] = ddn[i][0] * dn[0][j] + ddn[i][1] * dn[1][j];
}

template <bool JAC, typename T>
__device__ __forceinline__ void project_pinhole(const T* K, const T q[3], T px[2],
                                                T dpdq[2][3], T dpdK[2][10]) {
  T n[2], dn[2][3];
  normalize_point<JAC, T>(q, n, dn);
  const T r2 = n[0] * n[0] + n[1] * n[1];
  const T dd = T(1.0) + r2 * (K[5] + K[6] * r2);
  const T d[2] = {n[0] * dd, n[1] * dd};
  T dd_dq[2][3], dd_dk[2][2];
  if (JAC) {
    const T g = T(2.0) * (K[5] + T(2.0) * K[6] * r2);  // d(dd)/d(r2) * 2
    const T ddn[2][2] = {{dd + n[0] * n[0] * g, n[0] * n[1] * g},
                              {n[1] * n[0] * g, dd + n[1] * n[1] * g}};
    chain_2x2_2x3(ddn, dn, dd_dq);
    dd_dk[0][0] = n[0] * r2;      dd_dk[1][0] = n[1] * r2;
    dd_dk[0][1] = n[0] * r2 * r2; dd_dk[1][1] = n[1] * r2 * r2;
  }
  apply_k_skew<JAC, 2, T>(K, d, px, dd_dq, dd_dk, dpdq, dpdK);
}

template <bool JAC, typename T>
__device__ __forceinline__ void project_radtan(const T* K, const T q[3], T px[2],
                                               T dpdq[2][3], T dpdK[2][10]) {
  T n[2], dn[2][3];
  normalize_point<JAC, T>(q, n, dn);
  const T x = n[0], y = n[1];
  const T r2 = x * x + y * y;
  const T k1 = K[5], k2 = K[6], k3 = K[7], t1 = K[8], t2 = K[9];
  const T rd = T(1.0) + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
  const T tx = t2 * (r2 + T(2.0) * x * x) + T(2.0) * t1 * x * y;
  const T ty = t1 * (r2 + T(2.0) * y * y) + T(2.0) * t2 * x * y;
  const T d[2] = {x * rd + tx, y * rd + ty};
  T dd_dq[2][3], dd_dk[2][5];
  if (JAC) {
    const T g = k1 + T(2.0) * k2 * r2 + T(3.0) * k3 * r2 * r2;  // d rd / d r2
    const T ddn[2][2] = {
        {rd + T(2.0) * x * x * g + T(6.0) * t2 * x + T(2.0) * t1 * y, T(2.0) * x * y * g + T(2.0) * t2 * y + T(2.0) * t1 * x},
        {T(2.0) * x * y * g + T(2.0) * t1 * x + T(2.0) * t2 * y, rd + T(2.0) * y * y * g + T(6.0) * t1 * y + T(2.0) * t2 * x}};
    chain_2x2_2x3(ddn, dn, dd_dq);
    dd_dk[0][0] = x * r2;           dd_dk[1][0] = y * r2;
    dd_dk[0][1] = x * r2 * r2;      dd_dk[1][1] = y * r2 * r2;
    dd_dk[0][2] = x * r2 * r2 * r2; dd_dk[1][2] = y * r2 * r2 * r2;
    dd_dk[0][3] = T(2.0) * x * y;      dd_dk[1][3] = r2 + T(2.0) * y * y;  // t1
    dd_dk[0][4] = r2 + T(2.0) * x * x; dd_dk[1][4] = T(2.0) * x * y;       // t2
  }
  apply_k_skew<JAC, 5, T>(K, d, px, dd_dq, dd_dk, dpdq, dpdK);
}

template <bool JAC, typename T>
__device__ __forceinline__ void project_fisheye(const T* K, const T q[3], T px[2],
                                                T dpdq[2][3], T dpdK[2][10]) {
  const T x = q[0], y = q[1], z = q[2];
  const T r2 = x * x + y * y;
  T d[2], dd_dq[2][3], dd_dk[2][4];
  if (r2 < T(1e-8)) {  // fisheye_camera_model.h:243: pass the raw x, y through
    d[0] = x;
    d[1] = y;
    if (JAC) {
      dd_dq[0][0] = T(1.0); dd_dq[0][1] = T(0.0); dd_dq[0][2] = T(0.0);
      dd_dq[1][0] = T(0.0); dd_dq[1][1] = T(1.0); dd_dq[1][2] = T(0.0);
#pragma unroll
      for (int j = 0; j < 4; ++j) { dd_dk[0][j] = T(0.0); dd_dk[1][j] = T(0.0); }
    }
  } else {
    const T r = t_sqrt(r2);
    const T az = t_fabs(z);
    const T th = t_atan2(r, az);
    const T t2 = th * th, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const T poly = T(1.0) + K[5] * t2 + K[6] * t4 + K[7] * t6 + K[8] * t8;
    const T thd = th * poly;
    const T sgn = (z < T(0.0)) ? -T(1.0) : T(1.0);  // :263
    const T h = sgn * thd / r;
    d[0] = h * x;
    d[1] = h * y;
    if (JAC) {
      const T inv = T(1.0) / (r2 + z * z);
      const T dth_dr = az * inv;
      const T dth_dz = -r * inv * ((z < T(0.0)) ? -T(1.0) : T(1.0));  // d|z|/dz, Jet abs: z<0 ? -1 : +1
      const T dthd = T(1.0) + T(3.0) * K[5] * t2 + T(5.0) * K[6] * t4 + T(7.0) * K[7] * t6 + T(9.0) * K[8] * t8;
      // h = sgn * thd(th(r,z)) / r
      const T dh_dr = sgn * (dthd * dth_dr / r - thd / r2);
      const T dh_dz = sgn * dthd * dth_dz / r;
      const T dr_dx = x / r, dr_dy = y / r;
      dd_dq[0][0] = h + x * dh_dr * dr_dx; dd_dq[0][1] = x * dh_dr * dr_dy; dd_dq[0][2] = x * dh_dz;
      dd_dq[1][0] = y * dh_dr * dr_dx; dd_dq[1][1] = h + y * dh_dr * dr_dy; dd_dq[1][2] = y * dh_dz;
      const T base = sgn * th / r;
      dd_dk[0][0] = base * t2 * x; dd_dk[1][0] = base * t2 * y;
      dd_dk[0][1] = base * t4 * x; dd_dk[1][1] = base * t4 * y;
      dd_dk[0][2] = base * t6 * x; dd_dk[1][2] = base * t6 * y;
      dd_dk[0][3] = base * t8 * x; dd_dk[1][3] = base * t8 * y;
    }
  }
  apply_k_skew<JAC, 4, T>(K, d, px, dd_dq, dd_dk, dpdq, dpdK);
}

template <bool JAC, typename T>
__device__ __forceinline__ void project_fov(const T* K, const T q[3], T px[2],
                                            T dpdq[2][3], T dpdK[2][10]) {
  T n[2], dn[2][3];
  normalize_point<JAC, T>(q, n, dn);
  const T f = K[0], ar = K[1], om = K[4];
  const T r2 = n[0] * n[0] + n[1] * n[1];
  T rd, drd_dr2 = T(0.0), drd_dom = T(0.0);
  if (om < T(1e-3)) {  // fov_camera_model.h:227
    rd = (om * om * r2) / T(3.0) - om * om / T(12.0) + T(1.0);
    if (JAC) {
      drd_dr2 = om * om / T(3.0);
      drd_dom = T(2.0) * om * r2 / T(3.0) - om / T(6.0);
    }
  } else if (r2 < T(1e-3)) {  // :236
    const T th = t_tan(om / T(2.0));  // tan(omega / 2)
    const T num = -T(2.0) * th * (T(4.0) * r2 * th * th - T(3.0));
    rd = num / (T(3.0) * om);
    if (JAC) {
      drd_dr2 = -T(8.0) * th * th * th / (T(3.0) * om);
      const T dth = T(0.5) * (T(1.0) + th * th);
      const T dnum = (-T(24.0) * r2 * th * th + T(6.0)) * dth;
      drd_dom = dnum / (T(3.0) * om) - num / (T(3.0) * om * om);
    }
  } else {  // :249-254
    const T ru = t_sqrt(r2);
    const T th = t_tan(om / T(2.0));
    const T m = T(2.0) * ru * th;
    const T at = t_atan(m);
    rd = at / (ru * om);
    if (JAC) {
      const T im = T(1.0) / (T(1.0) + m * m);
      const T drd_dru = (T(2.0) * th * im) / (ru * om) - at / (r2 * om);
      drd_dr2 = drd_dru / (T(2.0) * ru);
      const T dth = T(0.5) * (T(1.0) + th * th);
      drd_dom = (T(2.0) * ru * dth * im) / (ru * om) - at / (ru * om * om);
    }
  }
  const T d[2] = {rd * n[0], rd * n[1]};
  px[0] = f * d[0] + K[2];
  px[1] = f * ar * d[1] + K[3];
  if (JAC) {
    const T g = T(2.0) * drd_dr2;
    const T ddn[2][2] = {{rd + n[0] * n[0] * g, n[0] * n[1] * g},
                              {n[1] * n[0] * g, rd + n[1] * n[1] * g}};
    T dd_dq[2][3];
    chain_2x2_2x3(ddn, dn, dd_dq);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      dpdq[0][j] = f * dd_dq[0][j];
      dpdq[1][j] = f * ar * dd_dq[1][j];
    }
    dpdK[0][0] = d[0]; dpdK[1][0] = ar * d[1];
    dpdK[0][1] = T(0.0);  dpdK[1][1] = f * d[1];
    dpdK[0][2] = T(1.0);  dpdK[1][2] = T(0.0);
    dpdK[0][3] = T(0.0);  dpdK[1][3] = T(1.0);
    dpdK[0][4] = f * n[0] * drd_dom; dpdK[1][4] = f * ar * n[1] * drd_dom;
#pragma unroll
    for (int j = 5; j < 10; ++j) { dpdK[0][j] = T(0.0); dpdK[1][j] = T(0.0); }
  }
}

template <bool JAC, typename T>
__device__ __forceinline__ void project_division(const T* K, const T q[3], T px[2],
                                                 T dpdq[2][3], T dpdK[2][10]) {
  T n[2], dn[2][3];
  normalize_point<JAC, T>(q, n, dn);
  const T f = K[0], ar = K[1], k = K[4];
  const T fy = f * ar;
  const T u[2] = {f * n[0], fy * n[1]};
  const T r2 = u[0] * u[0] + u[1] * u[1];
  const T denom = T(2.0) * k * r2;
  const T inner = T(1.0) - T(4.0) * k * r2;
  T scale = T(1.0), dsc_dr2 = T(0.0), dsc_dk = T(0.0);
  if (!(t_fabs(denom) < kDblEpsilon || inner < T(0.0))) {  // division_undistortion_camera_model.h:281
    const T sq = t_sqrt(inner);
    scale = (T(1.0) - sq) / denom;
    if (JAC) {
      dsc_dr2 = T(1.0) / (sq * r2) - scale / r2;
      dsc_dk = T(1.0) / (k * sq) - scale / k;
    }
  }
  px[0] = u[0] * scale + K[2];
  px[1] = u[1] * scale + K[3];
  if (JAC) {
    const T g = T(2.0) * dsc_dr2;
    // d(distorted)/du
    const T ddu[2][2] = {{scale + u[0] * u[0] * g, u[0] * u[1] * g},
                              {u[1] * u[0] * g, scale + u[1] * u[1] * g}};
    // du/dq = diag(f, fy) dn/dq
    T du_dq[2][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { du_dq[0][j] = f * dn[0][j]; du_dq[1][j] = fy * dn[1][j]; }
    chain_2x2_2x3(ddu, du_dq, dpdq);
    // d/df: du/df = (n0, ar n1); d/dar: du/dar = (0, f n1)
    const T duf[2] = {n[0], ar * n[1]};
    const T dua[2] = {T(0.0), f * n[1]};
    dpdK[0][0] = ddu[0][0] * duf[0] + ddu[0][1] * duf[1];
    dpdK[1][0] = ddu[1][0] * duf[0] + ddu[1][1] * duf[1];
    dpdK[0][1] = ddu[0][0] * dua[0] + ddu[0][1] * dua[1];
    dpdK[1][1] = ddu[1][0] * dua[0] + ddu[1][1] * dua[1];
    dpdK[0][2] = T(1.0); dpdK[1][2] = T(0.0);
    dpdK[0][3] = T(0.0); dpdK[1][3] = T(1.0);
    dpdK[0][4] = u[0] * dsc_dk; dpdK[1][4] = u[1] * dsc_dk;
#pragma unroll
    for (int j = 5; j < 10; ++j) { dpdK[0][j] = T(0.0); dpdK[1][j] = T(0.0); }
  }
}

// CreateReprojectionErrorCostFunction's dispatch
// (create_reprojection_error_cost_function.h:51-96).
template <bool JAC, typename T>
__device__ __forceinline__ void project(int model, const T* K, const T q[3], T px[2],
                                        T dpdq[2][3], T dpdK[2][10]) {
  switch (model) {
    case 0: project_pinhole<JAC, T>(K, q, px, dpdq, dpdK); break;
    case 1: project_radtan<JAC, T>(K, q, px, dpdq, dpdK); break;
    case 2: project_fisheye<JAC, T>(K, q, px, dpdq, dpdK); break;
    case 3: project_fov<JAC, T>(K, q, px, dpdq, dpdK); break;
    default: project_division<JAC, T>(K, q, px, dpdq, dpdK); break;
  }
}

// The full residual (reprojection_error.h:51-95).  Returns false where the
// reference functor does (|X - w C|^2 < 1e-8, :75-77).
//   Jext [2][6]  d r / d [C, angle-axis]
//   Jint [2][10] d r / d intrinsics (model order, zero padded)
//   Jpt  [2][4]  d r / d X (homogeneous)
// Parameters and the feature are always fp64.  The camera translation is removed in
// fp64 (the one place where fp32 would cancel catastrophically); rotation, projection
// and the Jacobian chain then run in T.
template <bool JAC, typename T>
__device__ __forceinline__ bool reprojection_error(int model, const double* ext, const double* K,
                                                   const double* X, double fx, double fy,
                                                   T r[2], T Jext[2][6],
                                                   T Jint[2][10], T Jpt[2][4]) {
  const double wd = X[3];
  const double ad[3] = {X[0] - wd * ext[0], X[1] - wd * ext[1], X[2] - wd * ext[2]};
  if (ad[0] * ad[0] + ad[1] * ad[1] + ad[2] * ad[2] < 1e-8) return false;
  const T w = (T)wd;
  const T a[3] = {(T)ad[0], (T)ad[1], (T)ad[2]};
  const T aa[3] = {(T)ext[3], (T)ext[4], (T)ext[5]};
  const T C[3] = {(T)ext[0], (T)ext[1], (T)ext[2]};
  T Kt[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) Kt[i] = (T)K[i];
  T q[3], R[3][3], dqdw[3][3], dpdq[2][3], px[2];
  rotate_point<JAC, T>(aa, a, q, R, dqdw);
  project<JAC, T>(model, Kt, q, px, dpdq, Jint);
  r[0] = (T)((double)px[0] - fx);
  r[1] = (T)((double)px[1] - fy);
  if (JAC) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // M = dp/dq * R   (= d r / d X[0:3])
      T M[3];
#pragma unroll
      for (int j = 0; j < 3; ++j)
        M[j] = dpdq[i][0] * R[0][j] + dpdq[i][1] * R[1][j] + dpdq[i][2] * R[2][j];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        Jpt[i][j] = M[j];
        Jext[i][j] = -w * M[j];  // da/dC = -w I
        Jext[i][3 + j] = dpdq[i][0] * dqdw[0][j] + dpdq[i][1] * dqdw[1][j] + dpdq[i][2] * dqdw[2][j];
      }
      Jpt[i][3] = -(M[0] * C[0] + M[1] * C[1] + M[2] * C[2]);  // da/dw = -C
    }
  }
  return true;
}

// ------------------------------------------------------------------------------
// Prepared per-camera record (camera_prepare_kernel, kernels.h).  The rotation
// q = R(w) a is evaluated N_obs times per pass but depends on the CAMERA only:
// sqrt, sincos and the divisions of Rodrigues' formula run once per camera per
// parameter set; the per-observation work is two 3x3 products.  The derivative
// wrt the angle-axis uses the left Jacobian of SO(3): R(w + d) = exp([Jl d]x) R(w), so
//   dq/dw_k = Jl[:,k] x q   and   dpdq_r . dq/dw_k = Jl[:,k] . (q x dpdq_r).
// In the first-order branch of ceres::AngleAxisRotatePoint (theta^2 <= DBL_EPSILON:
// q = a + w x a, reprojection_error.h:81 / ceres rotation.h) the executed expression has
// dq/dw_k = e_k x a: there Jl = I and the cross product takes a instead of q (`small`).
//   [0..8]   R row-major            [9..11]  C              [12..21] intrinsics (zero padded)
//   [22]     small-angle flag       [23]     -
//   [24..32] Jl diag(scale_w)       [33..35] scale of the position columns
//   [36..45] scale of the intrinsics columns                [46..47] -
// The first 24 doubles are all a residual-only pass needs.
// ------------------------------------------------------------------------------
constexpr int kPrepStride = 48;
constexpr int kPrepCostWords = 24;

__device__ __forceinline__ void prepare_camera_record(const double* __restrict__ ext,
                                                      const double* __restrict__ K, int nk,
                                                      const double* __restrict__ scale16,
                                                      double* __restrict__ out) {
  const double w[3] = {ext[3], ext[4], ext[5]};
  const double theta2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double R[9], Jl[9], small = 0.0;
  if (theta2 > kDblEpsilon) {
    const double theta = sqrt(theta2);
    double s, c;
    sincos(theta, &s, &c);
    const double inv_theta = 1.0 / theta;
    const double A1 = s * inv_theta;
    const double B1 = (1.0 - c) * inv_theta * inv_theta;
    const double C1 = (theta - s) * inv_theta * inv_theta * inv_theta;
    // R = c I + A1 [w]x + B1 w w^T  (the same expression rotate_point builds)
    R[0] = c + B1 * w[0] * w[0];
    R[1] = -A1 * w[2] + B1 * w[0] * w[1];
    R[2] = A1 * w[1] + B1 * w[0] * w[2];
    R[3] = A1 * w[2] + B1 * w[1] * w[0];
    R[4] = c + B1 * w[1] * w[1];
    R[5] = -A1 * w[0] + B1 * w[1] * w[2];
    R[6] = -A1 * w[1] + B1 * w[2] * w[0];
    R[7] = A1 * w[0] + B1 * w[2] * w[1];
    R[8] = c + B1 * w[2] * w[2];
    // Jl = I + B1 [w]x + C1 [w]x^2,  [w]x^2 = w w^T - theta^2 I
    const double d = 1.0 - C1 * theta2;
    Jl[0] = d + C1 * w[0] * w[0];
    Jl[1] = -B1 * w[2] + C1 * w[0] * w[1];
    Jl[2] = B1 * w[1] + C1 * w[0] * w[2];
    Jl[3] = B1 * w[2] + C1 * w[1] * w[0];
    Jl[4] = d + C1 * w[1] * w[1];
    Jl[5] = -B1 * w[0] + C1 * w[1] * w[2];
    Jl[6] = -B1 * w[1] + C1 * w[2] * w[0];
    Jl[7] = B1 * w[0] + C1 * w[2] * w[1];
    Jl[8] = d + C1 * w[2] * w[2];
  } else {
    R[0] = 1.0;   R[1] = -w[2]; R[2] = w[1];
    R[3] = w[2];  R[4] = 1.0;   R[5] = -w[0];
    R[6] = -w[1]; R[7] = w[0];  R[8] = 1.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) Jl[i] = (i % 4 == 0) ? 1.0 : 0.0;
    small = 1.0;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) out[i] = R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) out[9 + i] = ext[i];
#pragma unroll
  for (int i = 0; i < 10; ++i) out[12 + i] = (i < nk) ? K[i] : 0.0;
  out[22] = small;
  out[23] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int k = 0; k < 3; ++k) out[24 + 3 * i + k] = Jl[3 * i + k] * scale16[3 + k];
#pragma unroll
  for (int i = 0; i < 3; ++i) out[33 + i] = scale16[i];
#pragma unroll
  for (int i = 0; i < 10; ++i) out[36 + i] = scale16[6 + i];
  out[46] = 0.0;
  out[47] = 0.0;
}

// reprojection_error on a prepared record P (global memory or LDS).  Same value predicates
// and outputs as reprojection_error; with JAC the angle-axis columns of Jext and nothing
// else come out already multiplied by their Jacobi scale (folded into Jl).
template <bool JAC, typename T>
__device__ __forceinline__ bool reprojection_error_prepared(int model, const double* P,
                                                            const double* X, double fx, double fy,
                                                            T r[2], T Jext[2][6], T Jint[2][10],
                                                            T Jpt[2][4]) {
  const double wd = X[3];
  const double ad[3] = {X[0] - wd * P[9], X[1] - wd * P[10], X[2] - wd * P[11]};
  if (ad[0] * ad[0] + ad[1] * ad[1] + ad[2] * ad[2] < 1e-8) return false;
  const T a[3] = {(T)ad[0], (T)ad[1], (T)ad[2]};
  T Rm[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Rm[i] = (T)P[i];
  T q[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) q[i] = Rm[3 * i] * a[0] + Rm[3 * i + 1] * a[1] + Rm[3 * i + 2] * a[2];
  T Kt[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) Kt[i] = (T)P[12 + i];
  T dpdq[2][3], px[2];
  project<JAC, T>(model, Kt, q, px, dpdq, Jint);
  r[0] = (T)((double)px[0] - fx);
  r[1] = (T)((double)px[1] - fy);
  if (JAC) {
    const T w = (T)wd;
    const bool small = P[22] != 0.0;
    const T p[3] = {small ? a[0] : q[0], small ? a[1] : q[1], small ? a[2] : q[2]};
    const T C[3] = {(T)P[9], (T)P[10], (T)P[11]};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      T M[3];
#pragma unroll
      for (int j = 0; j < 3; ++j)
        M[j] = dpdq[i][0] * Rm[j] + dpdq[i][1] * Rm[3 + j] + dpdq[i][2] * Rm[6 + j];
      // c = p x dpdq_i
      const T c[3] = {p[1] * dpdq[i][2] - p[2] * dpdq[i][1], p[2] * dpdq[i][0] - p[0] * dpdq[i][2],
                      p[0] * dpdq[i][1] - p[1] * dpdq[i][0]};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        Jpt[i][j] = M[j];
        Jext[i][j] = -w * M[j];
        Jext[i][3 + j] = c[0] * (T)P[24 + j] + c[1] * (T)P[27 + j] + c[2] * (T)P[30 + j];
      }
      Jpt[i][3] = -(M[0] * C[0] + M[1] * C[1] + M[2] * C[2]);
    }
  }
  return true;
}

// Camera::ProjectPoint (camera.cc:204-213): pixel and depth = rotated_z / w; unlike the
// residual functor it has no degenerate-point test.
__device__ __forceinline__ double project_point_depth(int model, const double* ext,
                                                      const double* K, const double* X,
                                                      double px[2]) {
  const double a[3] = {X[0] - X[3] * ext[0], X[1] - X[3] * ext[1], X[2] - X[3] * ext[2]};
  const double aa[3] = {ext[3], ext[4], ext[5]};
  double q[3], R[3][3], dqdw[3][3], dpdq[2][3], dpdK[2][10];
  rotate_point<false, double>(aa, a, q, R, dqdw);
  project<false, double>(model, K, q, px, dpdq, dpdK);
  return q[2] / X[3];
}

// Camera::ProjectPoint on a prepared record (R, C, K): pixel only, no degenerate-point test
__device__ __forceinline__ void project_point_prepared(int model, const double* P, const double* X, double px[2]) {
  const double a[3] = {X[0] - X[3] * P[9], X[1] - X[3] * P[10], X[2] - X[3] * P[11]};
  double q[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) q[i] = P[3 * i] * a[0] + P[3 * i + 1] * a[1] + P[3 * i + 2] * a[2];
  double Kt[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) Kt[i] = P[12 + i];
  double dpdq[2][3], dpdK[2][10];
  project<false, double>(model, Kt, q, px, dpdq, dpdK);
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
