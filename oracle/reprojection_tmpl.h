/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Restatement of the reference's templated residual functor and the five
 * camera models' CameraToPixelCoordinates / DistortPoint.  This file is
 * included twice by ba_oracle.c: once with T = double (cost evaluation) and
 * once with T = jet (Jacobians), mirroring the reference's `template
 * <typename T>`.  The including file defines:
 *   T, FN(name), CST(c), ADD SUB MUL DIV NEG, SQRT COS SIN TAN ATAN ATAN2 ABS,
 *   VAL(x) (scalar part; branch predicates use values only, as Jet
 *   comparisons do).
 * Every function cites the reference lines it follows.
 */

/* ceres::AngleAxisRotatePoint -- external (ceres/rotation.h, Ceres 1.x);
 * call site: reference src/theia/sfm/camera/reprojection_error.h:81-83 and
 * camera.cc:206-208.  Published algorithm restated. */
static void FN(angle_axis_rotate_point)(const T aa[3], const T pt[3], T result[3]) {
  const T theta2 = ADD(ADD(MUL(aa[0], aa[0]), MUL(aa[1], aa[1])), MUL(aa[2], aa[2]));
  if (VAL(theta2) > DBL_EPSILON) {
    /* Away from zero, use the Rodrigues formula
     *   result = pt cos + (w x pt) sin + w (w . pt) (1 - cos) */
    const T theta = SQRT(theta2);
    const T costheta = COS(theta);
    const T sintheta = SIN(theta);
    const T theta_inverse = DIV(CST(1.0), theta);
    const T w[3] = {MUL(aa[0], theta_inverse), MUL(aa[1], theta_inverse),
                    MUL(aa[2], theta_inverse)};
    const T w_cross_pt[3] = {SUB(MUL(w[1], pt[2]), MUL(w[2], pt[1])),
                             SUB(MUL(w[2], pt[0]), MUL(w[0], pt[2])),
                             SUB(MUL(w[0], pt[1]), MUL(w[1], pt[0]))};
    const T tmp = MUL(ADD(ADD(MUL(w[0], pt[0]), MUL(w[1], pt[1])), MUL(w[2], pt[2])),
                      SUB(CST(1.0), costheta));
    for (int i = 0; i < 3; ++i)
      result[i] = ADD(ADD(MUL(pt[i], costheta), MUL(w_cross_pt[i], sintheta)),
                      MUL(w[i], tmp));
  } else {
    /* Near zero, first order Taylor: R = I + hat(w) */
    const T w_cross_pt[3] = {SUB(MUL(aa[1], pt[2]), MUL(aa[2], pt[1])),
                             SUB(MUL(aa[2], pt[0]), MUL(aa[0], pt[2])),
                             SUB(MUL(aa[0], pt[1]), MUL(aa[1], pt[0]))};
    for (int i = 0; i < 3; ++i) result[i] = ADD(pt[i], w_cross_pt[i]);
  }
}

/* Shared tail of the three "K matrix with skew" models:
 * reference pinhole_camera_model.h:206-209,
 *           pinhole_radial_tangential_camera_model.h:215-218,
 *           fisheye_camera_model.h:183-186.
 * K[0]=f K[1]=ar K[2]=skew K[3]=px K[4]=py. */
static void FN(apply_calibration_skew)(const T* K, const T d[2], T pixel[2]) {
  pixel[0] = ADD(ADD(MUL(K[0], d[0]), MUL(K[2], d[1])), K[3]);
  pixel[1] = ADD(MUL(MUL(K[0], K[1]), d[1]), K[4]);
}

/* PinholeCameraModel::DistortPoint, reference pinhole_camera_model.h:241-257 */
static void FN(pinhole_distort)(const T* K, const T u[2], T d[2]) {
  const T r_sq = ADD(MUL(u[0], u[0]), MUL(u[1], u[1]));
  const T dd = ADD(CST(1.0), MUL(r_sq, ADD(K[5], MUL(K[6], r_sq))));
  d[0] = MUL(u[0], dd);
  d[1] = MUL(u[1], dd);
}
/* PinholeCameraModel::CameraToPixelCoordinates, pinhole_camera_model.h:181-210 */
static void FN(pinhole_project)(const T* K, const T pt[3], T pixel[2]) {
  const T n[2] = {DIV(pt[0], pt[2]), DIV(pt[1], pt[2])};
  T d[2];
  FN(pinhole_distort)(K, n, d);
  FN(apply_calibration_skew)(K, d, pixel);
}

/* PinholeRadialTangentialCameraModel::DistortPoint,
 * reference pinhole_radial_tangential_camera_model.h:250-291.
 * K[5..7]=k1..k3, K[8]=t1, K[9]=t2 (:91-102). */
static void FN(radtan_distort)(const T* K, const T u[2], T d[2]) {
  const T r_sq = ADD(MUL(u[0], u[0]), MUL(u[1], u[1]));
  const T rd = ADD(ADD(ADD(CST(1.0), MUL(K[5], r_sq)), MUL(MUL(K[6], r_sq), r_sq)),
                   MUL(MUL(MUL(K[7], r_sq), r_sq), r_sq));
  const T tangential_x =
      ADD(MUL(K[9], ADD(r_sq, MUL(MUL(CST(2.0), u[0]), u[0]))),
          MUL(MUL(MUL(CST(2.0), K[8]), u[0]), u[1]));
  const T tangential_y =
      ADD(MUL(K[8], ADD(r_sq, MUL(MUL(CST(2.0), u[1]), u[1]))),
          MUL(MUL(MUL(CST(2.0), K[9]), u[0]), u[1]));
  d[0] = ADD(MUL(u[0], rd), tangential_x);
  d[1] = ADD(MUL(u[1], rd), tangential_y);
}
/* ...::CameraToPixelCoordinates, pinhole_radial_tangential_camera_model.h:190-219 */
static void FN(radtan_project)(const T* K, const T pt[3], T pixel[2]) {
  const T n[2] = {DIV(pt[0], pt[2]), DIV(pt[1], pt[2])};
  T d[2];
  FN(radtan_distort)(K, n, d);
  FN(apply_calibration_skew)(K, d, pixel);
}

/* FisheyeCameraModel::DistortPoint, reference fisheye_camera_model.h:223-267.
 * Takes the full 3-D point.  K[5..8] = k1..k4 (:67-77). */
static void FN(fisheye_distort)(const T* K, const T pt[3], T d[2]) {
  const T r_sq = ADD(MUL(pt[0], pt[0]), MUL(pt[1], pt[1]));
  if (VAL(r_sq) < 1e-8) { /* kVerySmallNumber, :227,243 */
    d[0] = pt[0];
    d[1] = pt[1];
    return;
  }
  const T r_numerator = SQRT(r_sq);
  const T theta = ATAN2(r_numerator, ABS(pt[2]));
  const T theta_sq = MUL(theta, theta);
  const T t4 = MUL(theta_sq, theta_sq);
  const T t6 = MUL(t4, theta_sq);
  const T t8 = MUL(t6, theta_sq);
  const T theta_d =
      MUL(theta, ADD(ADD(ADD(ADD(CST(1.0), MUL(K[5], theta_sq)), MUL(K[6], t4)),
                         MUL(K[7], t6)),
                     MUL(K[8], t8)));
  d[0] = DIV(MUL(theta_d, pt[0]), r_numerator);
  d[1] = DIV(MUL(theta_d, pt[1]), r_numerator);
  if (VAL(pt[2]) < 0.0) { /* :263-266 */
    d[0] = NEG(d[0]);
    d[1] = NEG(d[1]);
  }
}
/* FisheyeCameraModel::CameraToPixelCoordinates, fisheye_camera_model.h:162-187 */
static void FN(fisheye_project)(const T* K, const T pt[3], T pixel[2]) {
  T d[2];
  FN(fisheye_distort)(K, pt, d);
  FN(apply_calibration_skew)(K, d, pixel);
}

/* FOVCameraModel::DistortPoint, reference fov_camera_model.h:211-260.
 * K = [f, ar, px, py, omega] (:69-75). */
static void FN(fov_distort)(const T* K, const T u[2], T d[2]) {
  const T omega = K[4];
  const T r_u_sq = ADD(MUL(u[0], u[0]), MUL(u[1], u[1]));
  T r_d;
  if (VAL(omega) < 1e-3) { /* :227 */
    r_d = ADD(SUB(DIV(MUL(MUL(omega, omega), r_u_sq), CST(3.0)),
                  DIV(MUL(omega, omega), CST(12.0))),
              CST(1.0));
  } else if (VAL(r_u_sq) < 1e-3) { /* :236 */
    const T tan_half_omega = TAN(DIV(omega, CST(2.0)));
    r_d = DIV(MUL(MUL(CST(-2.0), tan_half_omega),
                  SUB(MUL(MUL(MUL(CST(4.0), r_u_sq), tan_half_omega), tan_half_omega),
                      CST(3.0))),
              MUL(CST(3.0), omega));
  } else { /* :249-254 */
    const T r_u = SQRT(r_u_sq);
    r_d = DIV(ATAN(MUL(MUL(CST(2.0), r_u), TAN(DIV(omega, CST(2.0))))), MUL(r_u, omega));
  }
  d[0] = MUL(r_d, u[0]);
  d[1] = MUL(r_d, u[1]);
}
/* FOVCameraModel::CameraToPixelCoordinates, fov_camera_model.h:155-182 (no skew) */
static void FN(fov_project)(const T* K, const T pt[3], T pixel[2]) {
  const T n[2] = {DIV(pt[0], pt[2]), DIV(pt[1], pt[2])};
  T d[2];
  FN(fov_distort)(K, n, d);
  const T focal_length_y = MUL(K[0], K[1]);
  pixel[0] = ADD(MUL(K[0], d[0]), K[2]);
  pixel[1] = ADD(MUL(focal_length_y, d[1]), K[3]);
}

/* DivisionUndistortionCameraModel::DistortPoint,
 * reference division_undistortion_camera_model.h:256-289.
 * K = [f, ar, px, py, k] (:76-82). */
static void FN(division_distort)(const T* K, const T u[2], T d[2]) {
  const T r_u_sq = ADD(MUL(u[0], u[0]), MUL(u[1], u[1]));
  const T k = K[4];
  const T denom = MUL(MUL(CST(2.0), k), r_u_sq);
  const T inner_sqrt = SUB(CST(1.0), MUL(MUL(CST(4.0), k), r_u_sq));
  if (fabs(VAL(denom)) < DBL_EPSILON || VAL(inner_sqrt) < 0.0) { /* :281 */
    d[0] = u[0];
    d[1] = u[1];
  } else {
    const T scale = DIV(SUB(CST(1.0), SQRT(inner_sqrt)), denom);
    d[0] = MUL(u[0], scale);
    d[1] = MUL(u[1], scale);
  }
}
/* ...::CameraToPixelCoordinates, division_undistortion_camera_model.h:172-202:
 * focal length first, then distortion, then principal point. */
static void FN(division_project)(const T* K, const T pt[3], T pixel[2]) {
  const T n[2] = {DIV(pt[0], pt[2]), DIV(pt[1], pt[2])};
  const T focal_length_y = MUL(K[0], K[1]);
  T u[2];
  u[0] = MUL(K[0], n[0]);
  u[1] = MUL(focal_length_y, n[1]);
  T d[2];
  FN(division_distort)(K, u, d);
  pixel[0] = ADD(d[0], K[2]);
  pixel[1] = ADD(d[1], K[3]);
}

/* CreateReprojectionErrorCostFunction's dispatch on the model type,
 * reference create_reprojection_error_cost_function.h:51-96. */
static void FN(camera_to_pixel)(int model, const T* K, const T pt[3], T pixel[2]) {
  switch (model) {
    case 0: FN(pinhole_project)(K, pt, pixel); break;
    case 1: FN(radtan_project)(K, pt, pixel); break;
    case 2: FN(fisheye_project)(K, pt, pixel); break;
    case 3: FN(fov_project)(K, pt, pixel); break;
    default: FN(division_project)(K, pt, pixel); break;
  }
}

/* ReprojectionError<CameraModel>::operator(),
 * reference src/theia/sfm/camera/reprojection_error.h:51-95.
 * ext = [C(3), angle_axis(3)] (camera.h:195-200).  Returns 0 where the
 * reference functor returns false (:75-77). */
static int FN(reprojection_error)(int model, const T* ext, const T* K, const T* X,
                                  const double feature[2], T res[2]) {
  /* Remove the translation (:62-64). */
  const T adjusted[3] = {SUB(X[0], MUL(X[3], ext[0])), SUB(X[1], MUL(X[3], ext[1])),
                         SUB(X[2], MUL(X[3], ext[2]))};
  const T sq = ADD(ADD(MUL(adjusted[0], adjusted[0]), MUL(adjusted[1], adjusted[1])),
                   MUL(adjusted[2], adjusted[2]));
  if (VAL(sq) < 1e-8) return 0; /* kVerySmallNumber (:59,75) */
  T rotated[3];
  FN(angle_axis_rotate_point)(ext + 3, adjusted, rotated); /* :81-83 */
  T reprojection[2];
  FN(camera_to_pixel)(model, K, rotated, reprojection); /* :87-89 */
  res[0] = SUB(reprojection[0], CST(feature[0])); /* :92-93 */
  res[1] = SUB(reprojection[1], CST(feature[1]));
  return 1;
}
