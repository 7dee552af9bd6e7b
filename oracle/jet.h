/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Forward-mode dual numbers ("Jets") restating how the reference obtains its
 * Jacobians: Theia wraps ReprojectionError<Model> in
 * ceres::AutoDiffCostFunction<..., 2, 6, kIntrinsicsSize, 4>
 * (reference: src/theia/sfm/camera/create_reprojection_error_cost_function.h:60-89),
 * i.e. Ceres evaluates the functor on ceres::Jet<double, 6+N+4>.
 * Ceres itself is an external, un-vendored, un-pinned dependency
 * (reference: CMakeLists.txt:152), so the Jet arithmetic below restates the
 * published ceres/jet.h rules (Ceres Solver 1.x, BSD): f = a + v.eps,
 * eps^2 = 0; comparisons use the scalar part only.
 *
 * Fixed width JN = 20 = 6 extrinsics + 10 (largest intrinsics block) + 4
 * (homogeneous point).  Slots: [0,6) extrinsics, [6,16) intrinsics, [16,20)
 * point.
 */
#ifndef ORACLE_JET_H_
#define ORACLE_JET_H_

#include <math.h>
#include <string.h>

#define JN 20

typedef struct jet {
  double a;
  double v[JN];
} jet;

static inline jet jet_const(double c) {
  jet r;
  r.a = c;
  memset(r.v, 0, sizeof(r.v));
  return r;
}
static inline jet jet_var(double c, int k) {
  jet r = jet_const(c);
  r.v[k] = 1.0;
  return r;
}
static inline jet jet_neg(jet f) {
  jet r;
  r.a = -f.a;
  for (int i = 0; i < JN; ++i) r.v[i] = -f.v[i];
  return r;
}
static inline jet jet_add(jet f, jet g) {
  jet r;
  r.a = f.a + g.a;
  for (int i = 0; i < JN; ++i) r.v[i] = f.v[i] + g.v[i];
  return r;
}
static inline jet jet_sub(jet f, jet g) {
  jet r;
  r.a = f.a - g.a;
  for (int i = 0; i < JN; ++i) r.v[i] = f.v[i] - g.v[i];
  return r;
}
/* jet.h: f*g = (f.a g.a, f.a g.v + f.v g.a) */
static inline jet jet_mul(jet f, jet g) {
  jet r;
  r.a = f.a * g.a;
  for (int i = 0; i < JN; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return r;
}
/* jet.h: f/g with g_inverse = 1/g.a, f_rel = f.a*g_inverse:
 *        (f_rel, (f.v - f_rel g.v) g_inverse) */
static inline jet jet_div(jet f, jet g) {
  jet r;
  const double g_inverse = 1.0 / g.a;
  const double f_rel = f.a * g_inverse;
  r.a = f_rel;
  for (int i = 0; i < JN; ++i) r.v[i] = (f.v[i] - f_rel * g.v[i]) * g_inverse;
  return r;
}
static inline jet jet_sqrt(jet f) {
  jet r;
  const double tmp = sqrt(f.a);
  const double two_a_inverse = 1.0 / (2.0 * tmp);
  r.a = tmp;
  for (int i = 0; i < JN; ++i) r.v[i] = f.v[i] * two_a_inverse;
  return r;
}
static inline jet jet_cos(jet f) {
  jet r;
  const double s = -sin(f.a);
  r.a = cos(f.a);
  for (int i = 0; i < JN; ++i) r.v[i] = s * f.v[i];
  return r;
}
static inline jet jet_sin(jet f) {
  jet r;
  const double c = cos(f.a);
  r.a = sin(f.a);
  for (int i = 0; i < JN; ++i) r.v[i] = c * f.v[i];
  return r;
}
static inline jet jet_tan(jet f) {
  jet r;
  const double tan_a = tan(f.a);
  const double tmp = 1.0 + tan_a * tan_a;
  r.a = tan_a;
  for (int i = 0; i < JN; ++i) r.v[i] = tmp * f.v[i];
  return r;
}
static inline jet jet_atan(jet f) {
  jet r;
  const double tmp = 1.0 / (1.0 + f.a * f.a);
  r.a = atan(f.a);
  for (int i = 0; i < JN; ++i) r.v[i] = tmp * f.v[i];
  return r;
}
/* jet.h: atan2(g, f) = atan(g/f): tmp = 1/(f.a^2+g.a^2);
 *        (atan2(g.a, f.a), tmp (-g.a f.v + f.a g.v)) */
static inline jet jet_atan2(jet g, jet f) {
  jet r;
  const double tmp = 1.0 / (f.a * f.a + g.a * g.a);
  r.a = atan2(g.a, f.a);
  for (int i = 0; i < JN; ++i) r.v[i] = tmp * (-g.a * f.v[i] + f.a * g.v[i]);
  return r;
}
/* jet.h: abs(f) = f.a < 0 ? -f : f */
static inline jet jet_abs(jet f) { return f.a < 0.0 ? jet_neg(f) : f; }

#endif /* ORACLE_JET_H_ */
