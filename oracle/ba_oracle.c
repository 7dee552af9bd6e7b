/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * See ba_oracle.h for scope, provenance and pinning status.
 */
#include "ba_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "jet.h"

/* ---- instantiate the residual template for double and for jets ---------- */
#define T double
#define FN(name) name##_d
#define CST(c) (c)
#define ADD(a, b) ((a) + (b))
#define SUB(a, b) ((a) - (b))
#define MUL(a, b) ((a) * (b))
#define DIV(a, b) ((a) / (b))
#define NEG(a) (-(a))
#define SQRT(a) sqrt(a)
#define COS(a) cos(a)
#define SIN(a) sin(a)
#define TAN(a) tan(a)
#define ATAN(a) atan(a)
#define ATAN2(a, b) atan2(a, b)
#define ABS(a) fabs(a)
#define VAL(a) (a)
#include "reprojection_tmpl.h"
#undef T
#undef FN
#undef CST
#undef ADD
#undef SUB
#undef MUL
#undef DIV
#undef NEG
#undef SQRT
#undef COS
#undef SIN
#undef TAN
#undef ATAN
#undef ATAN2
#undef ABS
#undef VAL

#define T jet
#define FN(name) name##_j
#define CST(c) jet_const(c)
#define ADD(a, b) jet_add(a, b)
#define SUB(a, b) jet_sub(a, b)
#define MUL(a, b) jet_mul(a, b)
#define DIV(a, b) jet_div(a, b)
#define NEG(a) jet_neg(a)
#define SQRT(a) jet_sqrt(a)
#define COS(a) jet_cos(a)
#define SIN(a) jet_sin(a)
#define TAN(a) jet_tan(a)
#define ATAN(a) jet_atan(a)
#define ATAN2(a, b) jet_atan2(a, b)
#define ABS(a) jet_abs(a)
#define VAL(x_) ((x_).a)
#include "reprojection_tmpl.h"
#undef T
#undef FN
#undef CST
#undef ADD
#undef SUB
#undef MUL
#undef DIV
#undef NEG
#undef SQRT
#undef COS
#undef SIN
#undef TAN
#undef ATAN
#undef ATAN2
#undef ABS
#undef VAL

/* ORACLE_TIMING=1: wall-clock of the phases of oracle_ba_solve on stderr (where the CPU baseline's time goes) */
#define OT_LAP(name) do { if (getenv("ORACLE_TIMING")) { const double n_ = now_s(); fprintf(stderr, "[oracle] %-22s %.3f s\n", name, n_ - ot_); ot_ = n_; } } while (0)
static double now_s(void);
static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int32_t oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int32_t n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

static int model_size(int model) {
  /* kIntrinsicsSize: pinhole_camera_model.h:84, pinhole_radial_tangential_
   * camera_model.h:89, fisheye_camera_model.h:65, fov_camera_model.h:67,
   * division_undistortion_camera_model.h:74 */
  static const int n[5] = {7, 10, 9, 5, 5};
  return (model >= 0 && model < 5) ? n[model] : -1;
}

/* ---- GetSubsetFromOptimizeIntrinsicsType --------------------------------- */
/* reference: pinhole_camera_model.cc:132-162,
 * pinhole_radial_tangential_camera_model.cc:150-185,
 * fisheye_camera_model.cc:142-175, fov_camera_model.cc:124-149,
 * division_undistortion_camera_model.cc:126-150. */
int32_t oracle_intrinsics_constant_mask(int32_t model, int32_t bits, uint8_t* mask) {
  const int n = model_size(model);
  if (n < 0 || !mask) return -1;
  memset(mask, 0, (size_t)n);
  if (bits == TMI_BA_INTRINSICS_ALL) return n;
  const int no_f = !(bits & TMI_BA_INTRINSICS_FOCAL_LENGTH);
  const int no_ar = !(bits & TMI_BA_INTRINSICS_ASPECT_RATIO);
  const int no_skew = !(bits & TMI_BA_INTRINSICS_SKEW);
  const int no_pp = !(bits & TMI_BA_INTRINSICS_PRINCIPAL_POINTS);
  const int no_rad = !(bits & TMI_BA_INTRINSICS_RADIAL_DISTORTION);
  const int no_tan = !(bits & TMI_BA_INTRINSICS_TANGENTIAL_DISTORTION);
  switch (model) {
    case 0: /* [f, ar, skew, px, py, k1, k2] */
      mask[0] = no_f; mask[1] = no_ar; mask[2] = no_skew;
      mask[3] = mask[4] = no_pp; mask[5] = mask[6] = no_rad;
      break;
    case 1: /* [f, ar, skew, px, py, k1, k2, k3, t1, t2] */
      mask[0] = no_f; mask[1] = no_ar; mask[2] = no_skew;
      mask[3] = mask[4] = no_pp; mask[5] = mask[6] = mask[7] = no_rad;
      mask[8] = mask[9] = no_tan;
      break;
    case 2: /* [f, ar, skew, px, py, k1..k4] */
      mask[0] = no_f; mask[1] = no_ar; mask[2] = no_skew;
      mask[3] = mask[4] = no_pp;
      mask[5] = mask[6] = mask[7] = mask[8] = no_rad;
      break;
    default: /* FOV / division: [f, ar, px, py, w|k] (no skew term) */
      mask[0] = no_f; mask[1] = no_ar; mask[2] = mask[3] = no_pp; mask[4] = no_rad;
      break;
  }
  return n;
}

/* ---- projection helpers --------------------------------------------------- */
void oracle_camera_to_pixel(int32_t model, const double* K, const double* pt, double* px) {
  camera_to_pixel_d(model, K, pt, px);
}

/* UndistortPoint of each model + PixelToCameraCoordinates (used only to
 * restate the reference round-trip tests).
 * reference: pinhole_camera_model.h:212-239,259-296;
 * pinhole_radial_tangential_camera_model.h:221-248,293-355;
 * fisheye_camera_model.h:189-216,269-335; fov_camera_model.h:184-209,262-306;
 * division_undistortion_camera_model.h:226-254,291-310. */
static void undistort_iterative(int model, const double* K, const double d[2], double u[2]) {
  u[0] = d[0];
  u[1] = d[1];
  for (int it = 0; it < 100; ++it) { /* kNumUndistortionIterations */
    const double p0 = u[0], p1 = u[1];
    const double r_sq = u[0] * u[0] + u[1] * u[1];
    if (model == 0) {
      const double dd = 1.0 + r_sq * (K[5] + K[6] * r_sq);
      u[0] = d[0] / dd;
      u[1] = d[1] / dd;
    } else if (model == 1) {
      const double rd = 1.0 + K[5] * r_sq + K[6] * r_sq * r_sq + K[7] * r_sq * r_sq * r_sq;
      const double tx = K[9] * (r_sq + 2.0 * u[0] * u[0]) + 2.0 * K[8] * u[0] * u[1];
      const double ty = K[8] * (r_sq + 2.0 * u[1] * u[1]) + 2.0 * K[9] * u[0] * u[1];
      u[0] = (d[0] - tx) / rd;
      u[1] = (d[1] - ty) / rd;
    } else { /* fisheye */
      const double r = sqrt(r_sq);
      if (r < 1e-8) {
        u[0] = d[0];
        u[1] = d[1];
        return;
      }
      const double theta = atan2(r, 1.0);
      const double t2 = theta * theta;
      const double theta_d = theta * (1.0 + K[5] * t2 + K[6] * t2 * t2 + K[7] * t2 * t2 * t2 +
                                      K[8] * t2 * t2 * t2 * t2);
      u[0] = r * d[0] / theta_d;
      u[1] = r * d[1] / theta_d;
    }
    if (fabs(u[0] - p0) < 1e-10 && fabs(u[1] - p1) < 1e-10) break;
  }
}

void oracle_pixel_to_camera(int32_t model, const double* K, const double* px, double* pt) {
  double d[2], u[2];
  if (model <= 2) {
    const double fy = K[0] * K[1];
    d[1] = (px[1] - K[4]) / fy;
    d[0] = (px[0] - K[3] - d[1] * K[2]) / K[0];
    undistort_iterative(model, K, d, u);
    pt[0] = u[0];
    pt[1] = u[1];
  } else if (model == 3) {
    const double fy = K[0] * K[1];
    d[0] = (px[0] - K[2]) / K[0];
    d[1] = (px[1] - K[3]) / fy;
    const double omega = K[4];
    const double r_d_sq = d[0] * d[0] + d[1] * d[1];
    double r_u;
    if (omega < 1e-3) {
      r_u = (omega * omega * r_d_sq) / 3.0 - omega * omega / 12.0 + 1.0;
    } else if (r_d_sq < 1e-3) {
      r_u = (omega * (omega * omega * r_d_sq + 3.0)) / (6.0 * tan(omega / 2.0));
    } else {
      const double r_d = sqrt(r_d_sq);
      r_u = tan(r_d * omega) / (2.0 * r_d * tan(omega / 2.0));
    }
    pt[0] = r_u * d[0];
    pt[1] = r_u * d[1];
  } else {
    const double fy = K[0] * K[1];
    d[0] = px[0] - K[2];
    d[1] = px[1] - K[3];
    const double r_d_sq = d[0] * d[0] + d[1] * d[1];
    const double undistortion = 1.0 / (1.0 + K[4] * r_d_sq);
    pt[0] = d[0] * undistortion / K[0];
    pt[1] = d[1] * undistortion / fy;
  }
  pt[2] = 1.0;
}

/* batched forms so the reference's grid tests can be restated cheaply */
void oracle_camera_to_pixel_batch(int32_t model, const double* K, const double* pts, int64_t n,
                                  double* px) {
  for (int64_t i = 0; i < n; ++i) camera_to_pixel_d(model, K, pts + 3 * i, px + 2 * i);
}
void oracle_pixel_to_camera_batch(int32_t model, const double* K, const double* px, int64_t n,
                                  double* pts) {
  for (int64_t i = 0; i < n; ++i) oracle_pixel_to_camera(model, K, px + 2 * i, pts + 3 * i);
}

/* Camera::ProjectPoint, reference camera.cc:204-213 */
double oracle_project_point(int32_t model, const double* ext, const double* K,
                            const double* X, double* pixel) {
  const double adjusted[3] = {X[0] - X[3] * ext[0], X[1] - X[3] * ext[1], X[2] - X[3] * ext[2]};
  double rotated[3];
  angle_axis_rotate_point_d(ext + 3, adjusted, rotated);
  camera_to_pixel_d(model, K, rotated, pixel);
  return rotated[2] / X[3];
}

/* ---- robust losses --------------------------------------------------------- */
/* ceres/loss_function.cc (Ceres 1.14) restated; constructed by
 * CreateLossFunction, reference create_loss_function.cc:42-71.
 * TukeyLoss uses the 1.x normalisation rho(0)'=1/2 (SURVEY App. B notes the
 * constant is version dependent). */
void oracle_loss(int32_t type, double a, double s, double rho[3]) {
  switch (type) {
    case TMI_BA_LOSS_HUBER: {
      const double b = a * a;
      if (s > b) {
        const double r = sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = fmax(DBL_MIN, a / r);
        rho[2] = -rho[1] / (2.0 * s);
      } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
      }
      break;
    }
    case TMI_BA_LOSS_SOFTLONE: {
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c;
      const double tmp = sqrt(sum);
      rho[0] = 2.0 * b * (tmp - 1.0);
      rho[1] = fmax(DBL_MIN, 1.0 / tmp);
      rho[2] = -(c * rho[1]) / (2.0 * sum);
      break;
    }
    case TMI_BA_LOSS_CAUCHY: {
      const double b = a * a, c = 1.0 / b;
      const double sum = 1.0 + s * c;
      const double inv = 1.0 / sum;
      rho[0] = b * log(sum);
      rho[1] = fmax(DBL_MIN, inv);
      rho[2] = -c * (inv * inv);
      break;
    }
    case TMI_BA_LOSS_ARCTAN: {
      const double b = 1.0 / (a * a);
      const double sum = 1.0 + s * s * b;
      const double inv = 1.0 / sum;
      rho[0] = a * atan2(s, a);
      rho[1] = fmax(DBL_MIN, inv);
      rho[2] = -2.0 * s * b * (inv * inv);
      break;
    }
    case TMI_BA_LOSS_TUKEY: {
      const double a_squared = a * a;
      if (s <= a_squared) {
        const double value = 1.0 - s / a_squared;
        const double value_sq = value * value;
        rho[0] = a_squared / 6.0 * (1.0 - value_sq * value);
        rho[1] = 0.5 * value_sq;
        rho[2] = -1.0 / a_squared * value;
      } else {
        rho[0] = a_squared / 6.0; rho[1] = 0.0; rho[2] = 0.0;
      }
      break;
    }
    default:
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
  }
}

/* ---- evaluation of one observation with jets -------------------------------- */
static int eval_obs_jet(int model, const double* ext, const double* K, int nk, const double* X,
                        const double* feat, double r[2], double J[2][JN]) {
  jet e[6], k[TMI_BA_MAX_INTRINSICS], x[4], res[2];
  for (int i = 0; i < 6; ++i) e[i] = jet_var(ext[i], i);
  for (int i = 0; i < nk; ++i) k[i] = jet_var(K[i], 6 + i);
  for (int i = nk; i < TMI_BA_MAX_INTRINSICS; ++i) k[i] = jet_const(0.0);
  for (int i = 0; i < 4; ++i) x[i] = jet_var(X[i], 16 + i);
  if (!reprojection_error_j(model, e, k, x, feat, res)) return 0;
  r[0] = res[0].a;
  r[1] = res[1].a;
  memcpy(J[0], res[0].v, sizeof(double) * JN);
  memcpy(J[1], res[1].v, sizeof(double) * JN);
  return 1;
}

int32_t oracle_ba_evaluate(const tmi_ba_problem* P, double* residuals, double* jac_full,
                           uint8_t* valid) {
  if (!P) return TMI_BA_ERR_INVALID_ARGUMENT;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < P->num_observations; ++i) {
    const int c = P->obs_camera[i], p = P->obs_point[i];
    const int g = P->camera_group[c];
    const int model = P->group_model[g];
    const double* K = P->intrinsics + P->group_offset[g];
    double r[2] = {0, 0}, J[2][JN];
    memset(J, 0, sizeof(J));
    const int ok = eval_obs_jet(model, P->extrinsics + 6 * c, K, model_size(model),
                                P->points + 4 * p, P->obs_xy + 2 * i, r, J);
    if (residuals) {
      residuals[2 * i] = r[0];
      residuals[2 * i + 1] = r[1];
    }
    if (jac_full) memcpy(jac_full + 2 * JN * i, J, sizeof(J));
    if (valid) valid[i] = (uint8_t)ok;
  }
  return TMI_BA_OK;
}

int64_t oracle_ba_cost(const tmi_ba_problem* P, const tmi_ba_options* O, double* cost,
                       double* rmse) {
  double c = 0.0, ss = 0.0;
  int64_t bad = 0;
  const int lt = O ? O->loss_function_type : 0;
  const double lw = O ? O->robust_loss_width : 1.0;
#pragma omp parallel for schedule(static) reduction(+ : c, ss, bad)
  for (int64_t i = 0; i < P->num_observations; ++i) {
    const int cam = P->obs_camera[i], p = P->obs_point[i];
    const int g = P->camera_group[cam];
    double r[2];
    if (!reprojection_error_d(P->group_model[g], P->extrinsics + 6 * cam,
                              P->intrinsics + P->group_offset[g], P->points + 4 * p,
                              P->obs_xy + 2 * i, r)) {
      ++bad;
      continue;
    }
    const double s = r[0] * r[0] + r[1] * r[1];
    double rho[3];
    oracle_loss(lt, lw, s, rho);
    c += 0.5 * rho[0];
    ss += s;
  }
  if (cost) *cost = c;
  if (rmse) *rmse = P->num_observations ? sqrt(ss / (double)P->num_observations) : 0.0;
  return bad;
}

/* ---- open-addressing map (block key -> block index) -------------------------- */
typedef struct {
  int64_t* keys;
  int64_t* vals;
  int64_t cap;
  int64_t n;
} hmap;
static uint64_t hmix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
static void hmap_init(hmap* h, int64_t cap) {
  int64_t c = 64;
  while (c < cap) c <<= 1;
  h->cap = c;
  h->n = 0;
  h->keys = (int64_t*)malloc(sizeof(int64_t) * c);
  h->vals = (int64_t*)malloc(sizeof(int64_t) * c);
  for (int64_t i = 0; i < c; ++i) h->keys[i] = -1;
}
static void hmap_free(hmap* h) { free(h->keys); free(h->vals); }
static int64_t hmap_get(const hmap* h, int64_t key) {
  int64_t i = (int64_t)(hmix((uint64_t)key) & (uint64_t)(h->cap - 1));
  while (h->keys[i] != -1) {
    if (h->keys[i] == key) return h->vals[i];
    i = (i + 1) & (h->cap - 1);
  }
  return -1;
}
static void hmap_put_raw(hmap* h, int64_t key, int64_t val) {
  int64_t i = (int64_t)(hmix((uint64_t)key) & (uint64_t)(h->cap - 1));
  while (h->keys[i] != -1) i = (i + 1) & (h->cap - 1);
  h->keys[i] = key;
  h->vals[i] = val;
  h->n++;
}
static void hmap_put(hmap* h, int64_t key, int64_t val) {
  if (2 * (h->n + 1) > h->cap) {
    hmap nh;
    hmap_init(&nh, h->cap * 2);
    for (int64_t i = 0; i < h->cap; ++i)
      if (h->keys[i] != -1) hmap_put_raw(&nh, h->keys[i], h->vals[i]);
    hmap_free(h);
    *h = nh;
  }
  hmap_put_raw(h, key, val);
}

/* ---- solver state ------------------------------------------------------------- */
#define MAXC 16 /* max camera-side columns per observation: 6 + 10 */

typedef struct {
  const tmi_ba_problem* P;
  const tmi_ba_options* O;
  int Nc, G, Np, dp;
  int64_t No;
  /* working parameters */
  double *ext, *intr, *pts;
  /* free-column maps */
  int *n_ext, *ext_idx;   /* [Nc], [Nc*6] */
  int *n_intr, *intr_idx; /* [G], [G*10]  */
  int* grp_private;       /* [G] 1 if exactly one camera uses the group */
  /* reduced blocks */
  int nrb, nr;
  int *rb_dim, *rb_off;
  int* rb_split; /* columns of a view's block that belong to the extrinsics parameter block */
  int *cam_rb, *grp_rb; /* -1 = none */
  /* observations sorted by point */
  int64_t* order;  /* sorted position -> caller index */
  int64_t* pt_ptr; /* [Np+1] */
  /* per observation (sorted order) */
  double *r, *Jc, *Jp; /* 2, 2*MAXC, 2*4 per obs */
  double* Ep;          /* Jp (V+D)^-1, 2*4 per obs */
  /* scaling and LM diagonal */
  double *scale_c, *scale_p; /* [nr], [Np*dp] */
  double *diag_c, *diag_p;   /* squared column norms of the scaled Jacobian */
  /* per point */
  double *Vinv, *gp, *tp; /* dp*dp, dp, dp */
  /* reduced system */
  hmap bmap;
  int64_t nblk;
  int* blk_i;
  int* blk_j;
  int64_t* blk_off;
  int64_t* row_ptr; /* CSR over blocks by block row */
  int64_t* row_blk;
  /* the observations of every reduced block row (a view's block: the view's observations; a shared intrinsics block:
   * those of all its views), as sorted positions k, ascending; and the point of every sorted position */
  int64_t* rb_obs_ptr; /* [nrb + 1] */
  int64_t* rb_obs;
  int* k_pt;           /* [No] */
  int* row_lpt;        /* [nrb] block rows by descending number of observations: the order threads take them in */
  double* S;
  int64_t S_len;
  double *gc, *rhs, *yc, *yp;
  double* dense; /* nr*nr when an exact solve is requested */
  int64_t pcg_iters;
  /* CLUSTER_JACOBI without shared intrinsics blocks: the visibility cluster of every reduced block (-1: none),
   * computed at the first PCG solve (visibility_clusters) */
  int* vis_cluster;
  int vis_ncl;
  /* CLUSTER_TRIDIAGONAL: the segments (chains of clusters along the degree-2 maximum spanning forest of the cluster
   * graph), computed at the first PCG solve (tridiagonal_segments): members of segment c are tri_rb[tri_ptr[c] ..
   * tri_ptr[c + 1]), tri_ord = the position of the member's cluster in its segment */
  int tri_nseg;
  int* tri_ptr;
  int* tri_rb;
  int* tri_ord;
} ost;

static int obs_parts(const ost* s, int c, int* rb0, int* n0, int* rb1, int* n1) {
  const int g = s->P->camera_group[c];
  *rb0 = s->cam_rb[c];
  *n0 = (*rb0 >= 0) ? s->rb_dim[*rb0] : 0;
  *rb1 = s->grp_private[g] ? -1 : s->grp_rb[g];
  *n1 = (*rb1 >= 0) ? s->rb_dim[*rb1] : 0;
  return *n0 + *n1;
}

/* Evaluate all observations.  with_jac: fill r (robustified), Jc, Jp (robustified,
 * Jacobi-scaled if apply_scale).  Returns cost via *cost, sum of squared raw
 * residuals via *ss, and the number of invalid observations. */
static int64_t evaluate(ost* s, int with_jac, int apply_scale, double* cost, double* ss_out) {
  const tmi_ba_problem* P = s->P;
  double c = 0.0, ss = 0.0;
  int64_t bad = 0;
  const int lt = s->O->loss_function_type;
  const double lw = s->O->robust_loss_width;
#pragma omp parallel for schedule(static) reduction(+ : c, ss, bad)
  for (int64_t k = 0; k < s->No; ++k) {
    const int64_t i = s->order[k];
    const int cam = P->obs_camera[i], p = P->obs_point[i];
    const int g = P->camera_group[cam];
    const int model = P->group_model[g];
    const double* K = s->intr + P->group_offset[g];
    double r[2];
    if (!with_jac) {
      if (!reprojection_error_d(model, s->ext + 6 * cam, K, s->pts + 4 * p, P->obs_xy + 2 * i,
                                r)) {
        ++bad;
        continue;
      }
      const double sq = r[0] * r[0] + r[1] * r[1];
      double rho[3];
      oracle_loss(lt, lw, sq, rho);
      c += 0.5 * rho[0];
      ss += sq;
      continue;
    }
    double J[2][JN];
    double* Jc = s->Jc + 2 * MAXC * k;
    double* Jp = s->Jp + 8 * k;
    memset(Jc, 0, sizeof(double) * 2 * MAXC);
    memset(Jp, 0, sizeof(double) * 8);
    if (!eval_obs_jet(model, s->ext + 6 * cam, K, model_size(model), s->pts + 4 * p,
                      P->obs_xy + 2 * i, r, J)) {
      ++bad;
      s->r[2 * k] = s->r[2 * k + 1] = 0.0;
      continue;
    }
    const double sq = r[0] * r[0] + r[1] * r[1];
    double rho[3];
    oracle_loss(lt, lw, sq, rho);
    c += 0.5 * rho[0];
    ss += sq;
    /* gather the free columns: [ext free | intr free] (merged block) or
     * [ext free] then [intr free] (shared group = second block) */
    int col = 0;
    if (s->cam_rb[cam] >= 0 || !s->grp_private[g]) {
      for (int a = 0; a < s->n_ext[cam]; ++a, ++col) {
        Jc[col] = J[0][s->ext_idx[6 * cam + a]];
        Jc[MAXC + col] = J[1][s->ext_idx[6 * cam + a]];
      }
    }
    if (s->grp_private[g] ? (s->cam_rb[cam] >= 0) : (s->grp_rb[g] >= 0)) {
      for (int a = 0; a < s->n_intr[g]; ++a, ++col) {
        Jc[col] = J[0][6 + s->intr_idx[10 * g + a]];
        Jc[MAXC + col] = J[1][6 + s->intr_idx[10 * g + a]];
      }
    }
    if (!P->point_constant || !P->point_constant[p]) {
      for (int a = 0; a < s->dp; ++a) {
        Jp[a] = J[0][16 + a];
        Jp[4 + a] = J[1][16 + a];
      }
    }
    /* ceres/corrector.cc restated (Triggs' correction): the Jacobian is
     * corrected first, with the UNcorrected residual, then the residual */
    {
      const double sqrt_rho1 = sqrt(rho[1]);
      double alpha_sq_norm = 0.0, residual_scaling = sqrt_rho1;
      if (!(sq == 0.0 || rho[2] <= 0.0)) {
        const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
        const double alpha = 1.0 - sqrt(D);
        residual_scaling = sqrt_rho1 / (1.0 - alpha);
        alpha_sq_norm = alpha / sq;
      }
      if (lt != TMI_BA_LOSS_TRIVIAL) {
        for (int a = 0; a < col; ++a) {
          const double rtj = Jc[a] * r[0] + Jc[MAXC + a] * r[1];
          Jc[a] = sqrt_rho1 * (Jc[a] - alpha_sq_norm * r[0] * rtj);
          Jc[MAXC + a] = sqrt_rho1 * (Jc[MAXC + a] - alpha_sq_norm * r[1] * rtj);
        }
        for (int a = 0; a < s->dp; ++a) {
          const double rtj = Jp[a] * r[0] + Jp[4 + a] * r[1];
          Jp[a] = sqrt_rho1 * (Jp[a] - alpha_sq_norm * r[0] * rtj);
          Jp[4 + a] = sqrt_rho1 * (Jp[4 + a] - alpha_sq_norm * r[1] * rtj);
        }
        r[0] *= residual_scaling;
        r[1] *= residual_scaling;
      }
    }
    if (apply_scale) {
      int rb0, n0, rb1, n1;
      obs_parts(s, cam, &rb0, &n0, &rb1, &n1);
      for (int a = 0; a < n0; ++a) {
        const double sc = s->scale_c[s->rb_off[rb0] + a];
        Jc[a] *= sc;
        Jc[MAXC + a] *= sc;
      }
      for (int a = 0; a < n1; ++a) {
        const double sc = s->scale_c[s->rb_off[rb1] + a];
        Jc[n0 + a] *= sc;
        Jc[MAXC + n0 + a] *= sc;
      }
      for (int a = 0; a < s->dp; ++a) {
        const double sc = s->scale_p[(int64_t)p * s->dp + a];
        Jp[a] *= sc;
        Jp[4 + a] *= sc;
      }
    }
    s->r[2 * k] = r[0];
    s->r[2 * k + 1] = r[1];
  }
  *cost = c;
  if (ss_out) *ss_out = ss;
  return bad;
}

/* Column sums over the observations, camera side: row by row over the row's observations (rb_obs: ascending sorted
 * position, the order the one serial loop over all observations added them in -- same sums bit for bit, on every
 * thread count), point side: point by point.  sq = 1: squared column norms of the current (possibly scaled) Jacobian;
 * sq = 0: the gradient J^T r. */
static void column_sums(ost* s, int sq, double* dc, double* dpn) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int bq = 0; bq < s->nrb; ++bq) {
    const int bi = s->row_lpt[bq];
    const int ni = s->rb_dim[bi];
    double* out = dc + s->rb_off[bi];
    for (int a = 0; a < ni; ++a) out[a] = 0.0;
    for (int64_t q = s->rb_obs_ptr[bi]; q < s->rb_obs_ptr[bi + 1]; ++q) {
      const int64_t k = s->rb_obs[q];
      int rb0, n0, rb1, n1;
      obs_parts(s, s->P->obs_camera[s->order[k]], &rb0, &n0, &rb1, &n1);
      const double* Jc = s->Jc + 2 * MAXC * k + (rb0 == bi ? 0 : n0);
      const double w0 = sq ? 0.0 : s->r[2 * k], w1 = sq ? 0.0 : s->r[2 * k + 1];
      for (int a = 0; a < ni; ++a)
        out[a] += sq ? Jc[a] * Jc[a] + Jc[MAXC + a] * Jc[MAXC + a] : Jc[a] * w0 + Jc[MAXC + a] * w1;
    }
  }
#pragma omp parallel for schedule(dynamic, 1024)
  for (int p = 0; p < s->Np; ++p) {
    double* out = dpn + (int64_t)p * s->dp;
    for (int a = 0; a < s->dp; ++a) out[a] = 0.0;
    for (int64_t k = s->pt_ptr[p]; k < s->pt_ptr[p + 1]; ++k) {
      const double* Jp = s->Jp + 8 * k;
      const double w0 = sq ? 0.0 : s->r[2 * k], w1 = sq ? 0.0 : s->r[2 * k + 1];
      for (int a = 0; a < s->dp; ++a)
        out[a] += sq ? Jp[a] * Jp[a] + Jp[4 + a] * Jp[4 + a] : Jp[a] * w0 + Jp[4 + a] * w1;
    }
  }
}

/* squared column norms of the current (possibly scaled) Jacobian */
static void column_sqnorms(ost* s, double* dc, double* dpn) { column_sums(s, 1, dc, dpn); }

/* gradient J^T r (of the current Jacobian) -> gc [nr], gp [Np*dp] */
static void gradient(ost* s, double* gc, double* gp) { column_sums(s, 0, gc, gp); }

static int64_t block_lookup(const ost* s, int bi, int bj) {
  return hmap_get(&s->bmap, ((int64_t)bi << 32) | (int64_t)(uint32_t)bj);
}

static void block_insert(ost* s, int bi, int bj, int64_t* cap) {
  const int64_t key = ((int64_t)bi << 32) | (int64_t)(uint32_t)bj;
  if (hmap_get(&s->bmap, key) >= 0) return;
  if (s->nblk == *cap) {
    *cap *= 2;
    s->blk_i = (int*)realloc(s->blk_i, sizeof(int) * (size_t)*cap);
    s->blk_j = (int*)realloc(s->blk_j, sizeof(int) * (size_t)*cap);
    s->blk_off = (int64_t*)realloc(s->blk_off, sizeof(int64_t) * (size_t)*cap);
  }
  s->blk_i[s->nblk] = bi;
  s->blk_j[s->nblk] = bj;
  s->blk_off[s->nblk] = s->S_len;
  s->S_len += (int64_t)s->rb_dim[bi] * s->rb_dim[bj];
  hmap_put(&s->bmap, key, s->nblk);
  s->nblk++;
}

/* structure of the reduced camera matrix: every pair of reduced blocks that
 * share a track, plus the diagonal and the ext/intrinsics cross blocks */
static void build_structure(ost* s) {
  int64_t cap = 1024;
  s->blk_i = (int*)malloc(sizeof(int) * (size_t)cap);
  s->blk_j = (int*)malloc(sizeof(int) * (size_t)cap);
  s->blk_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)cap);
  s->nblk = 0;
  s->S_len = 0;
  hmap_init(&s->bmap, 1024);
  for (int b = 0; b < s->nrb; ++b) block_insert(s, b, b, &cap);
  int* rbs = (int*)malloc(sizeof(int) * 2 * 65536);
  int rbs_cap = 2 * 65536;
  for (int p = 0; p < s->Np; ++p) {
    const int64_t k0 = s->pt_ptr[p], k1 = s->pt_ptr[p + 1];
    if (2 * (k1 - k0) > rbs_cap) {
      rbs_cap = (int)(2 * (k1 - k0));
      rbs = (int*)realloc(rbs, sizeof(int) * (size_t)rbs_cap);
    }
    int n = 0;
    for (int64_t k = k0; k < k1; ++k) {
      int rb0, n0, rb1, n1;
      obs_parts(s, s->P->obs_camera[s->order[k]], &rb0, &n0, &rb1, &n1);
      if (rb0 >= 0) rbs[n++] = rb0;
      if (rb1 >= 0) rbs[n++] = rb1;
    }
    for (int a = 0; a < n; ++a)
      for (int b = 0; b < n; ++b) block_insert(s, rbs[a], rbs[b], &cap);
  }
  free(rbs);
  /* CSR by block row */
  s->row_ptr = (int64_t*)calloc((size_t)s->nrb + 1, sizeof(int64_t));
  s->row_blk = (int64_t*)malloc(sizeof(int64_t) * (size_t)s->nblk);
  for (int64_t b = 0; b < s->nblk; ++b) s->row_ptr[s->blk_i[b] + 1]++;
  for (int i = 0; i < s->nrb; ++i) s->row_ptr[i + 1] += s->row_ptr[i];
  int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)s->nrb);
  memcpy(fill, s->row_ptr, sizeof(int64_t) * (size_t)s->nrb);
  for (int64_t b = 0; b < s->nblk; ++b) s->row_blk[fill[s->blk_i[b]]++] = b;
  free(fill);
  s->S = (double*)malloc(sizeof(double) * (size_t)s->S_len);
  /* observations by reduced block row (counting sort over the sorted positions: ascending k within a row) */
  s->k_pt = (int*)malloc(sizeof(int) * (size_t)(s->No + 1));
  s->rb_obs_ptr = (int64_t*)calloc((size_t)s->nrb + 2, sizeof(int64_t));
  int64_t total = 0;
  for (int p = 0; p < s->Np; ++p)
    for (int64_t k = s->pt_ptr[p]; k < s->pt_ptr[p + 1]; ++k) {
      s->k_pt[k] = p;
      int rb0, n0, rb1, n1;
      obs_parts(s, s->P->obs_camera[s->order[k]], &rb0, &n0, &rb1, &n1);
      if (rb0 >= 0) { s->rb_obs_ptr[rb0 + 1]++; ++total; }
      if (rb1 >= 0) { s->rb_obs_ptr[rb1 + 1]++; ++total; }
    }
  for (int i = 0; i < s->nrb; ++i) s->rb_obs_ptr[i + 1] += s->rb_obs_ptr[i];
  s->rb_obs = (int64_t*)malloc(sizeof(int64_t) * (size_t)(total + 1));
  int64_t* at = (int64_t*)malloc(sizeof(int64_t) * (size_t)(s->nrb + 1));
  memcpy(at, s->rb_obs_ptr, sizeof(int64_t) * (size_t)s->nrb);
  for (int64_t k = 0; k < s->No; ++k) {
    int rb0, n0, rb1, n1;
    obs_parts(s, s->P->obs_camera[s->order[k]], &rb0, &n0, &rb1, &n1);
    if (rb0 >= 0) s->rb_obs[at[rb0]++] = k;
    if (rb1 >= 0) s->rb_obs[at[rb1]++] = k;
  }
  free(at);
  /* rows by descending work (counting sort on the observation count would do; nrb is small: insertion into buckets
   * by a simple sort).  With dynamic scheduling the longest rows then start first and the last ones are short. */
  s->row_lpt = (int*)malloc(sizeof(int) * (size_t)(s->nrb + 1));
  for (int i = 0; i < s->nrb; ++i) s->row_lpt[i] = i;
  {
    /* stable merge sort by descending count (deterministic order for equal counts) */
    int* tmp = (int*)malloc(sizeof(int) * (size_t)(s->nrb + 1));
    for (int w = 1; w < s->nrb; w *= 2) {
      for (int lo = 0; lo < s->nrb; lo += 2 * w) {
        const int mid = lo + w < s->nrb ? lo + w : s->nrb, hi = lo + 2 * w < s->nrb ? lo + 2 * w : s->nrb;
        int a = lo, b = mid, o = lo;
        while (a < mid || b < hi) {
          const int64_t ca = a < mid ? s->rb_obs_ptr[s->row_lpt[a] + 1] - s->rb_obs_ptr[s->row_lpt[a]] : -1;
          const int64_t cb = b < hi ? s->rb_obs_ptr[s->row_lpt[b] + 1] - s->rb_obs_ptr[s->row_lpt[b]] : -1;
          if (b >= hi || (a < mid && ca >= cb)) tmp[o++] = s->row_lpt[a++];
          else tmp[o++] = s->row_lpt[b++];
        }
      }
      memcpy(s->row_lpt, tmp, sizeof(int) * (size_t)s->nrb);
    }
    free(tmp);
  }
}

/* small SPD inverse via Cholesky (n <= 4).  Returns 0 if not positive definite. */
static int spd_inverse(int n, const double* A, double* Ainv) {
  double L[16];
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    if (!(d > 0.0)) return 0;
    L[j * n + j] = sqrt(d);
    for (int i = j + 1; i < n; ++i) {
      double v = A[i * n + j];
      for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = v / L[j * n + j];
    }
  }
  /* invert L, then Ainv = L^-T L^-1 */
  double Li[16];
  memset(Li, 0, sizeof(Li));
  for (int j = 0; j < n; ++j) {
    Li[j * n + j] = 1.0 / L[j * n + j];
    for (int i = j + 1; i < n; ++i) {
      double v = 0.0;
      for (int k = j; k < i; ++k) v -= L[i * n + k] * Li[k * n + j];
      Li[i * n + j] = v / L[i * n + i];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double v = 0.0;
      for (int k = (i > j ? i : j); k < n; ++k) v += Li[k * n + i] * Li[k * n + j];
      Ainv[i * n + j] = v;
    }
  return 1;
}

/* Build S = U + Dc - W (V + Dp)^-1 W^T and rhs = gc - W (V+Dp)^-1 gp.
 * Dc, Dp = clamp(diag)/radius.  Returns 0 on a singular point block. */
static int build_reduced(ost* s, double radius) {
  const int dp = s->dp;
  const double lo = s->O->min_lm_diagonal, hi = s->O->max_lm_diagonal;
  int ok = 1;
  /* phase 1: per point (V + Dp)^-1, t_p, E_i = Jp_i (V+Dp)^-1 */
#pragma omp parallel for schedule(dynamic, 256)
  for (int p = 0; p < s->Np; ++p) {
    double V[16], g[4];
    memset(V, 0, sizeof(V));
    memset(g, 0, sizeof(g));
    for (int64_t k = s->pt_ptr[p]; k < s->pt_ptr[p + 1]; ++k) {
      const double* Jp = s->Jp + 8 * k;
      for (int a = 0; a < dp; ++a) {
        for (int b = 0; b < dp; ++b) V[a * dp + b] += Jp[a] * Jp[b] + Jp[4 + a] * Jp[4 + b];
        g[a] += Jp[a] * s->r[2 * k] + Jp[4 + a] * s->r[2 * k + 1];
      }
    }
    for (int a = 0; a < dp; ++a) {
      double d = s->diag_p[(int64_t)p * dp + a];
      d = fmin(fmax(d, lo), hi);
      V[a * dp + a] += d / radius;
    }
    double* Vi = s->Vinv + (int64_t)p * dp * dp;
    if (!spd_inverse(dp, V, Vi)) {
#pragma omp atomic write
      ok = 0;
      continue;
    }
    for (int a = 0; a < dp; ++a) {
      s->gp[(int64_t)p * dp + a] = g[a];
      double t = 0.0;
      for (int b = 0; b < dp; ++b) t += Vi[a * dp + b] * g[b];
      s->tp[(int64_t)p * dp + a] = t;
    }
    for (int64_t k = s->pt_ptr[p]; k < s->pt_ptr[p + 1]; ++k) {
      const double* Jp = s->Jp + 8 * k;
      double* E = s->Ep + 8 * k;
      for (int row = 0; row < 2; ++row)
        for (int b = 0; b < dp; ++b) {
          double v = 0.0;
          for (int a = 0; a < dp; ++a) v += Jp[4 * row + a] * Vi[a * dp + b];
          E[4 * row + b] = v;
        }
    }
  }
  if (!ok) return 0;
  memset(s->S, 0, sizeof(double) * (size_t)s->S_len);
  memset(s->rhs, 0, sizeof(double) * (size_t)s->nr);
  /* phase 2: accumulate, ROW BY ROW: a thread takes a block row, maps the row's columns to its blocks once (no
   * hashing per update) and walks the row's observations; every update of the row lands in the row's own blocks
   * (~1 MB: they stay in cache) and in its own slice of rhs.  (Until round 4 every thread walked all the points and
   * updated the rows it owned through the hash map: 78 % of an LM iteration on the bench problem, ORACLE_TIMING.) */
#pragma omp parallel
  {
    int64_t* col_blk = (int64_t*)malloc(sizeof(int64_t) * (size_t)(s->nrb + 1));
    for (int i = 0; i < s->nrb; ++i) col_blk[i] = -1;
#pragma omp for schedule(dynamic, 1)
    for (int bq = 0; bq < s->nrb; ++bq) {
      const int bi = s->row_lpt[bq];
      for (int64_t q = s->row_ptr[bi]; q < s->row_ptr[bi + 1]; ++q) col_blk[s->blk_j[s->row_blk[q]]] = s->row_blk[q];
      const int ni = s->rb_dim[bi];
      double* rhs = s->rhs + s->rb_off[bi];
      for (int64_t qi = s->rb_obs_ptr[bi]; qi < s->rb_obs_ptr[bi + 1]; ++qi) {
        const int64_t ki = s->rb_obs[qi];
        const int p = s->k_pt[ki];
        const int64_t k0 = s->pt_ptr[p], k1 = s->pt_ptr[p + 1];
        const double* tp = s->tp + (int64_t)p * dp;
        const int cam_i = s->P->obs_camera[s->order[ki]];
        int rb[2], n[2], c0[2];
        obs_parts(s, cam_i, &rb[0], &n[0], &rb[1], &n[1]);
        c0[0] = 0;
        c0[1] = n[0];
        const int pi = (rb[0] == bi) ? 0 : 1;  /* which part of the observation's camera-side block this row is */
        const double* Ji = s->Jc + 2 * MAXC * ki;
        const double* Jpi = s->Jp + 8 * ki;
        const double* Ei = s->Ep + 8 * ki;
        /* reduced residual r~ = r - Jp t_p */
        double rt[2] = {s->r[2 * ki], s->r[2 * ki + 1]};
        for (int a = 0; a < dp; ++a) {
          rt[0] -= Jpi[a] * tp[a];
          rt[1] -= Jpi[4 + a] * tp[a];
        }
        const double* Ai0 = Ji + c0[pi];
        const double* Ai1 = Ji + MAXC + c0[pi];
        for (int a = 0; a < ni; ++a) rhs[a] += Ai0[a] * rt[0] + Ai1[a] * rt[1];
        /* U: this observation's own blocks (pi, pj) */
        for (int pj = 0; pj < 2; ++pj) {
          if (rb[pj] < 0) continue;
          double* B = s->S + s->blk_off[col_blk[rb[pj]]];
          const int nj = n[pj];
          const double* Aj0 = Ji + c0[pj];
          const double* Aj1 = Ji + MAXC + c0[pj];
          for (int a = 0; a < ni; ++a)
            for (int b = 0; b < nj; ++b) B[a * nj + b] += Ai0[a] * Aj0[b] + Ai1[a] * Aj1[b];
        }
        /* Schur: - A_i^T (E_i Jp_j^T) A_j over every observation j of the track */
        for (int64_t kj = k0; kj < k1; ++kj) {
          const int cam_j = s->P->obs_camera[s->order[kj]];
          int rbj[2], nj_[2], cj0[2];
          obs_parts(s, cam_j, &rbj[0], &nj_[0], &rbj[1], &nj_[1]);
          cj0[0] = 0;
          cj0[1] = nj_[0];
          const double* Jj = s->Jc + 2 * MAXC * kj;
          const double* Jpj = s->Jp + 8 * kj;
          double M[4] = {0, 0, 0, 0};
          for (int a = 0; a < dp; ++a) {
            M[0] += Ei[a] * Jpj[a];
            M[1] += Ei[a] * Jpj[4 + a];
            M[2] += Ei[4 + a] * Jpj[a];
            M[3] += Ei[4 + a] * Jpj[4 + a];
          }
          for (int pj = 0; pj < 2; ++pj) {
            if (rbj[pj] < 0) continue;
            const int nj = nj_[pj];
            const double* Aj0 = Jj + cj0[pj];
            const double* Aj1 = Jj + MAXC + cj0[pj];
            double* B = s->S + s->blk_off[col_blk[rbj[pj]]];
            for (int b = 0; b < nj; ++b) {
              const double t0 = M[0] * Aj0[b] + M[1] * Aj1[b];
              const double t1 = M[2] * Aj0[b] + M[3] * Aj1[b];
              for (int a = 0; a < ni; ++a) B[a * nj + b] -= Ai0[a] * t0 + Ai1[a] * t1;
            }
          }
        }
      }
      for (int64_t q = s->row_ptr[bi]; q < s->row_ptr[bi + 1]; ++q) col_blk[s->blk_j[s->row_blk[q]]] = -1;
    }
    free(col_blk);
  }
  /* LM diagonal on the camera blocks */
  for (int b = 0; b < s->nrb; ++b) {
    double* B = s->S + s->blk_off[block_lookup(s, b, b)];
    const int n = s->rb_dim[b];
    for (int a = 0; a < n; ++a) {
      double d = s->diag_c[s->rb_off[b] + a];
      d = fmin(fmax(d, lo), hi);
      B[a * n + a] += d / radius;
    }
  }
  return 1;
}

static void spmv(const ost* s, const double* x, double* y) {
#pragma omp parallel for schedule(dynamic, 8)
  for (int i = 0; i < s->nrb; ++i) {
    const int ni = s->rb_dim[i];
    double acc[MAXC];
    for (int a = 0; a < ni; ++a) acc[a] = 0.0;
    for (int64_t q = s->row_ptr[i]; q < s->row_ptr[i + 1]; ++q) {
      const int64_t b = s->row_blk[q];
      const int j = s->blk_j[b], nj = s->rb_dim[j];
      const double* B = s->S + s->blk_off[b];
      const double* xj = x + s->rb_off[j];
      for (int a = 0; a < ni; ++a) {
        double v = 0.0;
        for (int c = 0; c < nj; ++c) v += B[a * nj + c] * xj[c];
        acc[a] += v;
      }
    }
    for (int a = 0; a < ni; ++a) y[s->rb_off[i] + a] = acc[a];
  }
}

static double dot(const double* a, const double* b, int n) {
  double v = 0.0;
  for (int i = 0; i < n; ++i) v += a[i] * b[i];
  return v;
}

/* dense SPD inverse of a small block via Cholesky (n <= MAXC) into Minv */
static int block_inverse(int n, const double* A, double* Ainv) {
  double L[MAXC * MAXC];
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
    if (!(d > 0.0)) return 0;
    L[j * n + j] = sqrt(d);
    for (int i = j + 1; i < n; ++i) {
      double v = A[i * n + j];
      for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = v / L[j * n + j];
    }
  }
  for (int c = 0; c < n; ++c) {
    double y[MAXC];
    for (int i = 0; i < n; ++i) {
      double v = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) v -= L[i * n + k] * y[k];
      y[i] = v / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double v = y[i];
      for (int k = i + 1; k < n; ++k) v -= L[k * n + i] * Ainv[k * n + c];
      Ainv[i * n + c] = v / L[i * n + i];
    }
  }
  return 1;
}

static int* g_last_vis = NULL;
static int g_last_vis_n = 0;
/* test hook of CLUSTER_TRIDIAGONAL: segment and position of its cluster inside the segment, per camera (-1: the view
 * block is in no segment), of the last solve that built them */
static int* g_last_tri = NULL; /* [2 * n]: segment, ordinal */
static int g_last_tri_n = 0;
int32_t oracle_last_tridiagonal_segments(int32_t* segment, int32_t* ordinal, int32_t n) {
  int rc = -1;
#pragma omp critical(oracle_last_vis)
  if (g_last_tri && n >= g_last_tri_n) {
    for (int c = 0; c < g_last_tri_n; ++c) {
      segment[c] = g_last_tri[2 * c];
      ordinal[c] = g_last_tri[2 * c + 1];
    }
    rc = g_last_tri_n;
  }
  return rc;
}

int32_t oracle_last_visibility_clusters(int32_t* out, int32_t n) {
  int rc = -1;
#pragma omp critical(oracle_last_vis)
  if (g_last_vis && n >= g_last_vis_n) {
    for (int c = 0; c < g_last_vis_n; ++c) out[c] = g_last_vis[c];
    rc = g_last_vis_n;
  }
  return rc;
}

/* The views of a problem WITHOUT shared intrinsics blocks clustered by visibility, as
 * ceres::VisibilityBasedPreconditioner::ClusterCameras does for CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL (the reference only
 * passes the two enums on: bundle_adjuster.cc:59-63, bundle_adjustment.h:86-89; Ceres 1.14 restated from its sources'
 * published behaviour -- visibility.cc, canonical_views_clustering.cc, single_linkage_clustering.cc,
 * visibility_based_preconditioner.cc; PARITY UNPINNED):
 *   vertices: the camera-side parameter blocks -- the extrinsics of a view and, if it has free intrinsics, its
 *   intrinsics block; every vertex carries a self edge of weight 1; two blocks are joined with weight
 *   |points both see| / sqrt(|points of a| |points of b|) over the non-constant points, so the two blocks of one view
 *   (same visibility) sit at similarity 1;
 *   CANONICAL_VIEWS (type 0): centres are added greedily -- the vertex whose promotion raises the summed similarity
 *   of its neighbours to their centre most, minus 3 per centre (kCanonicalViewsSizePenaltyWeight; the similarity
 *   penalty is 0, the view score weight 0) -- until the best gain is <= 0 and there are at least 3 centres; a vertex
 *   belongs to the centre it is most similar to, one that touches no centre to cluster (index mod #clusters);
 *   SINGLE_LINKAGE (type 1): connected components of the edges with similarity >= 0.9.
 * Ties: the lower index.  cluster[rb] = cluster of the view's blocks (they always share one).  Returns #clusters. */
static int visibility_clusters(ost* s, int type, int* cluster) {
  const int n = s->nrb;
  double* ntr = (double*)calloc((size_t)n + 1, sizeof(double));
  double* cnt = (double*)calloc((size_t)s->nblk + 1, sizeof(double)); /* per block (i, j), both orientations */
  int* mult = (int*)calloc((size_t)n + 1, sizeof(int));
  int* rbs = (int*)malloc(sizeof(int) * (size_t)(s->Nc + 1));
  for (int p = 0; p < s->Np; ++p) {
    if (s->P->point_constant && s->P->point_constant[p]) continue;
    int m = 0;
    for (int64_t k = s->pt_ptr[p]; k < s->pt_ptr[p + 1]; ++k) {
      const int rb = s->cam_rb[s->P->obs_camera[s->order[k]]];
      if (rb >= 0) rbs[m++] = rb;
    }
    for (int a = 0; a < m; ++a) {
      ntr[rbs[a]] += 1.0;
      for (int b = 0; b < m; ++b)
        if (a != b) {
          const int64_t q = block_lookup(s, rbs[a], rbs[b]);
          if (q >= 0) cnt[q] += 1.0;
        }
    }
  }
  free(rbs);
  for (int b = 0; b < n; ++b) mult[b] = (s->rb_split[b] > 0 ? 1 : 0) + (s->rb_dim[b] > s->rb_split[b] ? 1 : 0);
  /* the neighbours of a vertex in ascending order (block rows are in insertion order): sums of similarities are then
   * taken in a defined order, whatever built the block structure */
  int64_t* nbr = (int64_t*)malloc(sizeof(int64_t) * (size_t)(s->nblk + 1));
  memcpy(nbr, s->row_blk, sizeof(int64_t) * (size_t)s->nblk);
  for (int i = 0; i < n; ++i) { /* insertion sort by column: rows are short or nearly sorted */
    for (int64_t a = s->row_ptr[i] + 1; a < s->row_ptr[i + 1]; ++a) {
      const int64_t q = nbr[a];
      int64_t b = a - 1;
      while (b >= s->row_ptr[i] && s->blk_j[nbr[b]] > s->blk_j[q]) {
        nbr[b + 1] = nbr[b];
        --b;
      }
      nbr[b + 1] = q;
    }
  }
#define VC_W(q) (cnt[q] / sqrt(ntr[s->blk_i[q]] * ntr[s->blk_j[q]]))
  int ncl = 0;
  for (int b = 0; b < n; ++b) cluster[b] = -1;
  if (type == 1) {
    int* parent = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    for (int b = 0; b < n; ++b) parent[b] = b;
    for (int i = 0; i < n; ++i)
      for (int64_t e = s->row_ptr[i]; e < s->row_ptr[i + 1]; ++e) {
        const int64_t q = nbr[e];
        const int j = s->blk_j[q];
        if (j <= i || cnt[q] <= 0.0 || VC_W(q) < 0.9) continue;
        int a = i, b = j;
        while (parent[a] != a) a = parent[a];
        while (parent[b] != b) b = parent[b];
        if (a != b) parent[a > b ? a : b] = a > b ? b : a;
      }
    int* id = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    for (int b = 0; b < n; ++b) id[b] = -1;
    for (int b = 0; b < n; ++b) {
      if (mult[b] == 0) continue;
      int r = b;
      while (parent[r] != r) r = parent[r];
      if (id[r] < 0) id[r] = ncl++;
      cluster[b] = id[r];
    }
    free(id);
    free(parent);
  } else {
    double* sim = (double*)calloc((size_t)n + 1, sizeof(double));
    int* to_center = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    int* promoted = (int*)calloc((size_t)n + 1, sizeof(int)); /* blocks of the view that are centres */
    int* center_id = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    int n_centers = 0;
    for (int b = 0; b < n; ++b) to_center[b] = center_id[b] = -1;
    for (;;) {
      int best = -1;
      double best_gain = -DBL_MAX;
      for (int v = 0; v < n; ++v) {
        if (promoted[v] >= mult[v]) continue;
        double gain = (1.0 > sim[v]) ? mult[v] * (1.0 - sim[v]) : 0.0; /* its self edge and its sibling block */
        for (int64_t e = s->row_ptr[v]; e < s->row_ptr[v + 1]; ++e) {
          const int64_t q = nbr[e];
          const int u = s->blk_j[q];
          if (u == v || cnt[q] <= 0.0) continue;
          const double w = VC_W(q);
          if (w > sim[u]) gain += mult[u] * (w - sim[u]);
        }
        gain -= 3.0;
        if (gain > best_gain) {
          best_gain = gain;
          best = v;
        }
      }
      if (best < 0) break;
      if (best_gain <= 0.0 && n_centers >= 3) break;
      ++n_centers;
      ++promoted[best];
      if (center_id[best] < 0) center_id[best] = ncl++;
      if (1.0 > sim[best]) {
        sim[best] = 1.0;
        to_center[best] = best;
      }
      for (int64_t e = s->row_ptr[best]; e < s->row_ptr[best + 1]; ++e) {
        const int64_t q = nbr[e];
        const int u = s->blk_j[q];
        if (u == best || cnt[q] <= 0.0) continue;
        const double w = VC_W(q);
        if (w > sim[u]) {
          sim[u] = w;
          to_center[u] = best;
        }
      }
    }
    if (ncl == 0) ncl = 1;
    for (int b = 0; b < n; ++b) {
      if (mult[b] == 0) continue;
      cluster[b] = to_center[b] >= 0 ? center_id[to_center[b]] : b % ncl;
    }
    free(sim);
    free(to_center);
    free(promoted);
    free(center_id);
  }
#undef VC_W
  free(nbr);
  free(ntr);
  free(cnt);
  free(mult);
  return ncl;
}

/* CLUSTER_TRIDIAGONAL (ceres::CLUSTER_TRIDIAGONAL, bundle_adjustment.h:86-89; the reference only passes the enum on,
 * bundle_adjuster.cc:59-63 -- Ceres 1.14 visibility_based_preconditioner.cc ComputeClusterTridiagonalSparsity /
 * CreateClusterGraph / graph_algorithms.h Degree2MaximumSpanningForest restated; PARITY UNPINNED like the rest of the
 * Ceres layer).  Where CLUSTER_JACOBI keeps the cluster pairs (i, i), the tridiagonal variant also keeps the pairs
 * (i, j) that are edges of a degree-2 maximum spanning forest of the cluster graph:
 *   vertices  the clusters of CLUSTER_JACOBI -- with shared intrinsics blocks {shared block, its views}, otherwise the
 *             visibility clusters -- and every reduced block outside them as a cluster of its own; numbered by their
 *             lowest reduced block;
 *   edges     between two clusters that see a common (non-constant) track, weight = the number of such tracks;
 *   forest    the edges in decreasing order of (weight, lower vertex, higher vertex) -- Ceres sorts the pairs
 *             <weight, <v1, v2>> with reverse iterators --, an edge is taken unless one of its ends has two edges
 *             already or both ends are connected already.
 * The components of the forest are paths; the preconditioner of a path is its block-tridiagonal matrix (the clusters'
 * principal submatrices of S and the blocks of S between neighbours on the path), factored exactly.  A path is walked
 * from its end with the lower cluster number; it is cut where the next cluster would take the matrix beyond
 * TMI_BA_MAX_CLUSTER_DIM unknowns (a dense factor here, a sparse one in Ceres -- engine and oracle cut alike), a
 * cluster that alone is beyond it keeps its SCHUR_JACOBI blocks, and a segment that is one single reduced block IS
 * its SCHUR_JACOBI block. */
static void tridiagonal_segments(ost* s, int have_shared) {
  const int n = s->nrb;
  int* cl_of = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  for (int b = 0; b < n; ++b) cl_of[b] = -1;
  int ncl = 0;
  if (have_shared) {
    for (int g = 0; g < s->G; ++g) {
      if (s->grp_rb[g] < 0) continue;
      for (int c = 0; c < s->Nc; ++c)
        if (s->P->camera_group[c] == g && s->cam_rb[c] >= 0) cl_of[s->cam_rb[c]] = ncl;
      cl_of[s->grp_rb[g]] = ncl;
      ++ncl;
    }
  } else {
    for (int b = 0; b < n; ++b)
      if (s->vis_cluster[b] >= 0) {
        cl_of[b] = s->vis_cluster[b];
        if (cl_of[b] + 1 > ncl) ncl = cl_of[b] + 1;
      }
  }
  for (int b = 0; b < n; ++b)
    if (cl_of[b] < 0) cl_of[b] = ncl++;
  /* number the clusters by their lowest reduced block (empty ids of the visibility clustering drop out) */
  {
    int* lowest = (int*)malloc(sizeof(int) * (size_t)(ncl + 1));
    int* renum = (int*)malloc(sizeof(int) * (size_t)(ncl + 1));
    for (int c = 0; c < ncl; ++c) lowest[c] = renum[c] = -1;
    int m = 0;
    for (int b = 0; b < n; ++b)
      if (lowest[cl_of[b]] < 0) {
        lowest[cl_of[b]] = b;
        renum[cl_of[b]] = m++;
      }
    for (int b = 0; b < n; ++b) cl_of[b] = renum[cl_of[b]];
    ncl = m;
    free(lowest);
    free(renum);
  }
  int* cdim = (int*)calloc((size_t)ncl + 1, sizeof(int));
  for (int b = 0; b < n; ++b) cdim[cl_of[b]] += s->rb_dim[b];
  /* weights: tracks seen from both clusters */
  double* W = (double*)calloc((size_t)ncl * ncl + 1, sizeof(double));
  {
    int* seen = (int*)malloc(sizeof(int) * (size_t)(s->Nc + 2));
    for (int p = 0; p < s->Np; ++p) {
      if (s->P->point_constant && s->P->point_constant[p]) continue;
      int m = 0;
      for (int64_t k = s->pt_ptr[p]; k < s->pt_ptr[p + 1]; ++k) {
        const int rb = s->cam_rb[s->P->obs_camera[s->order[k]]];
        if (rb < 0) continue;
        const int c = cl_of[rb];
        int dup = 0;
        for (int a = 0; a < m; ++a)
          if (seen[a] == c) dup = 1;
        if (!dup) seen[m++] = c;
      }
      for (int a = 0; a < m; ++a)
        for (int b = 0; b < m; ++b)
          if (seen[a] < seen[b]) W[(int64_t)seen[a] * ncl + seen[b]] += 1.0;
    }
    free(seen);
  }
  /* edges, sorted: weight descending, then (a, b) descending */
  int64_t ne = 0;
  for (int64_t i = 0; i < (int64_t)ncl * ncl; ++i)
    if (W[i] > 0.0) ++ne;
  int* ea = (int*)malloc(sizeof(int) * (size_t)(ne + 1));
  int* eb = (int*)malloc(sizeof(int) * (size_t)(ne + 1));
  {
    int64_t k = 0;
    for (int a = 0; a < ncl; ++a)
      for (int b = a + 1; b < ncl; ++b)
        if (W[(int64_t)a * ncl + b] > 0.0) {
          ea[k] = a;
          eb[k] = b;
          ++k;
        }
    /* insertion sort is quadratic: a simple merge-free heap sort on the index instead */
    int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(ne + 1));
    for (int64_t i = 0; i < ne; ++i) idx[i] = i;
#define TRI_BEFORE(x, y)                                                                                   \
  (W[(int64_t)ea[x] * ncl + eb[x]] != W[(int64_t)ea[y] * ncl + eb[y]]                                      \
       ? W[(int64_t)ea[x] * ncl + eb[x]] > W[(int64_t)ea[y] * ncl + eb[y]]                                 \
       : (ea[x] != ea[y] ? ea[x] > ea[y] : eb[x] > eb[y]))
    /* heap sort: idx ends in "before" order */
    for (int64_t start = ne / 2 - 1; start >= 0; --start) {
      int64_t root = start;
      for (;;) {
        int64_t child = 2 * root + 1;
        if (child >= ne) break;
        if (child + 1 < ne && TRI_BEFORE(idx[child], idx[child + 1])) ++child; /* the LATER one is the larger heap key */
        if (TRI_BEFORE(idx[root], idx[child])) {
          const int64_t t = idx[root]; idx[root] = idx[child]; idx[child] = t;
          root = child;
        } else {
          break;
        }
      }
    }
    for (int64_t end = ne - 1; end > 0; --end) {
      { const int64_t t = idx[0]; idx[0] = idx[end]; idx[end] = t; }
      int64_t root = 0;
      for (;;) {
        int64_t child = 2 * root + 1;
        if (child >= end) break;
        if (child + 1 < end && TRI_BEFORE(idx[child], idx[child + 1])) ++child;
        if (TRI_BEFORE(idx[root], idx[child])) {
          const int64_t t = idx[root]; idx[root] = idx[child]; idx[child] = t;
          root = child;
        } else {
          break;
        }
      }
    }
#undef TRI_BEFORE
    /* forest */
    int* deg = (int*)calloc((size_t)ncl + 1, sizeof(int));
    int* nb = (int*)malloc(sizeof(int) * (size_t)(2 * ncl + 2));
    int* parent = (int*)malloc(sizeof(int) * (size_t)(ncl + 1));
    for (int c = 0; c < ncl; ++c) parent[c] = c;
    for (int64_t q = 0; q < ne; ++q) {
      const int a = ea[idx[q]], b = eb[idx[q]];
      if (deg[a] == 2 || deg[b] == 2) continue;
      int ra = a, rb2 = b;
      while (parent[ra] != ra) ra = parent[ra];
      while (parent[rb2] != rb2) rb2 = parent[rb2];
      if (ra == rb2) continue;
      nb[2 * a + deg[a]++] = b;
      nb[2 * b + deg[b]++] = a;
      if (rb2 < ra) { const int t = ra; ra = rb2; rb2 = t; }
      parent[rb2] = ra;
    }
    /* paths, from the end with the lower number; segments */
    int* visited = (int*)calloc((size_t)ncl + 1, sizeof(int));
    int* path = (int*)malloc(sizeof(int) * (size_t)(ncl + 1));
    s->tri_ptr = (int*)calloc((size_t)ncl + 2, sizeof(int));
    s->tri_rb = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    s->tri_ord = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    s->tri_nseg = 0;
    for (int c0 = 0; c0 < ncl; ++c0) {
      if (visited[c0] || deg[c0] == 2) continue;
      int len = 0, prev = -1, cur = c0;
      while (cur >= 0) {
        visited[cur] = 1;
        path[len++] = cur;
        int next = -1;
        for (int k = 0; k < deg[cur]; ++k)
          if (nb[2 * cur + k] != prev) next = nb[2 * cur + k];
        prev = cur;
        cur = next;
      }
      /* cut into segments */
      int i = 0;
      while (i < len) {
        if (cdim[path[i]] > TMI_BA_MAX_CLUSTER_DIM) { ++i; continue; } /* keeps its SCHUR_JACOBI blocks */
        int j = i, dim = 0;
        while (j < len && cdim[path[j]] <= TMI_BA_MAX_CLUSTER_DIM && dim + cdim[path[j]] <= TMI_BA_MAX_CLUSTER_DIM) dim += cdim[path[j++]];
        /* members: cluster by cluster, reduced blocks ascending */
        int m = s->tri_ptr[s->tri_nseg];
        const int m0 = m;
        for (int q = i; q < j; ++q)
          for (int b = 0; b < n; ++b)
            if (cl_of[b] == path[q]) {
              s->tri_rb[m] = b;
              s->tri_ord[m] = q - i;
              ++m;
            }
        if (m - m0 >= 2) { /* a single reduced block is its own SCHUR_JACOBI block */
          s->tri_ptr[s->tri_nseg + 1] = m;
          s->tri_nseg++;
        }
        i = j;
      }
    }
    free(visited); free(path); free(deg); free(nb); free(parent); free(idx);
  }
  free(ea); free(eb); free(W); free(cdim); free(cl_of);
  {
    int* mine = (int*)malloc(sizeof(int) * (size_t)(2 * s->Nc + 2));
    int* seg_of = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    int* ord_of = (int*)malloc(sizeof(int) * (size_t)(n + 1));
    for (int b = 0; b < n; ++b) seg_of[b] = ord_of[b] = -1;
    for (int g = 0; g < s->tri_nseg; ++g)
      for (int a = s->tri_ptr[g]; a < s->tri_ptr[g + 1]; ++a) {
        seg_of[s->tri_rb[a]] = g;
        ord_of[s->tri_rb[a]] = s->tri_ord[a];
      }
    for (int c = 0; c < s->Nc; ++c) {
      mine[2 * c] = s->cam_rb[c] >= 0 ? seg_of[s->cam_rb[c]] : -1;
      mine[2 * c + 1] = s->cam_rb[c] >= 0 ? ord_of[s->cam_rb[c]] : -1;
    }
    free(seg_of);
    free(ord_of);
#pragma omp critical(oracle_last_vis)
    {
      free(g_last_tri);
      g_last_tri = mine;
      g_last_tri_n = s->Nc;
    }
  }
}

/* ceres/conjugate_gradients_solver.cc (Ceres 1.14) restated, applied to the
 * explicit reduced system with the SCHUR_JACOBI preconditioner (inverse of
 * the diagonal blocks of S).  r_tolerance = -1, q_tolerance = eta
 * (LevenbergMarquardtStrategy::ComputeStep).  Returns 1 ok, 0 failure. */
static int solve_pcg(ost* s) {
  const int n = s->nr;
  double* x = s->yc;
  double* r = (double*)malloc(sizeof(double) * (size_t)n * 4);
  double *z = r + n, *pvec = r + 2 * n, *tmp = r + 3 * n;
  double* Minv = (double*)malloc(sizeof(double) * (size_t)s->nrb * MAXC * MAXC);
  int ok = 1;
  for (int b = 0; b < s->nrb; ++b) {
    const int nb = s->rb_dim[b];
    const double* B = s->S + s->blk_off[block_lookup(s, b, b)];
    if (s->O->preconditioner_type == TMI_BA_PRECOND_IDENTITY) {
      double* M = Minv + (int64_t)b * MAXC * MAXC;
      for (int a = 0; a < nb * nb; ++a) M[a] = 0.0;
      for (int a = 0; a < nb; ++a) M[a * nb + a] = 1.0;
    } else if (s->O->preconditioner_type == TMI_BA_PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS) {
      /* ceres/schur_jacobi_preconditioner.cc: the block diagonal has one block per
       * e-eliminated PARAMETER block, i.e. extrinsics and intrinsics of a view apart */
      double T[MAXC * MAXC];
      const int sp = s->rb_split[b];
      for (int i = 0; i < nb; ++i)
        for (int j = 0; j < nb; ++j) T[i * nb + j] = ((i < sp) != (j < sp)) ? 0.0 : B[i * nb + j];
      if (!block_inverse(nb, T, Minv + (int64_t)b * MAXC * MAXC)) ok = 0;
    } else if (!block_inverse(nb, B, Minv + (int64_t)b * MAXC * MAXC)) {
      ok = 0;
    }
  }
  if (!ok) {
    free(r);
    free(Minv);
    return 0;
  }
  /* CLUSTER_JACOBI (ceres::CLUSTER_JACOBI, bundle_adjustment.h:87-89) on a problem with shared intrinsics blocks: Ceres
   * clusters the cameras by visibility and inverts the block diagonal of S over the clusters exactly; here a cluster
   * is a shared intrinsics block together with the views that share it (theia_mi355_ba.h) -- the principal submatrix
   * of S over {views of g, g}, factored densely (Cholesky), views of private groups keep their own block.  A
   * cluster whose matrix is not positive definite falls back to its diagonal blocks. */
  const int tri = s->O->preconditioner_type == TMI_BA_PRECOND_CLUSTER_TRIDIAGONAL;
  const int clustered = s->O->preconditioner_type == TMI_BA_PRECOND_CLUSTER_JACOBI || tri;
  int ncl = 0;
  int* cl_ptr = NULL;     /* [ncl + 1] into cl_rb */
  int* cl_rb = NULL;      /* member reduced blocks: the views of the group ascending, then the group's block */
  int* cl_ok = NULL;
  double** cl_L = NULL;   /* dense lower factors */
  int* cl_n = NULL;
  /* Without shared intrinsics blocks the clusters are Ceres' visibility clusters of the views (visibility_clusters). */
  int have_shared = 0;
  for (int g = 0; g < s->G; ++g)
    if (s->grp_rb[g] >= 0) have_shared = 1;
  int n_cand = s->G; /* candidate clusters: one per group, or one per visibility cluster */
  if (clustered && !have_shared) {
    if (!s->vis_cluster) {
      s->vis_cluster = (int*)malloc(sizeof(int) * (size_t)(s->nrb + 1));
      s->vis_ncl = visibility_clusters(s, s->O->visibility_clustering_type, s->vis_cluster);
      /* test hook (oracle_last_visibility_clusters): the cluster of every CAMERA of the last clustered solve */
      int* mine = (int*)malloc(sizeof(int) * (size_t)(s->Nc + 1));
      for (int c = 0; c < s->Nc; ++c) mine[c] = s->cam_rb[c] >= 0 ? s->vis_cluster[s->cam_rb[c]] : -1;
      /* (concurrent oracle_ba_solve calls: the hook is swapped under a lock) */
#pragma omp critical(oracle_last_vis)
      {
        free(g_last_vis);
        g_last_vis = mine;
        g_last_vis_n = s->Nc;
      }
    }
    n_cand = s->vis_ncl;
  }
  /* CLUSTER_TRIDIAGONAL: the "clusters" factored below are the segments of tridiagonal_segments -- chains of clusters
   * with the blocks of S between neighbours.  Dropping blocks can cost positive definiteness: Ceres then halves every
   * off-diagonal cluster-pair cell and factors once more (VisibilityBasedPreconditioner::UpdateImpl /
   * ScaleOffDiagonalCells); a second failure fails the preconditioner update and with it the linear solve. */
  if (tri) {
    if (!s->tri_ptr) tridiagonal_segments(s, have_shared);
    n_cand = s->tri_nseg;
  }
  int* cl_ord = NULL;
  int tri_attempt = 0;
  double off_scale = 1.0;
  /* (test hooks shared with the engine: the factors of the two attempts, tests/test_cluster_jacobi.py) */
  if (tri && getenv("TMI_BA_TEST_TRI_SCALE0")) off_scale = atof(getenv("TMI_BA_TEST_TRI_SCALE0"));
tri_again:
  if (clustered) {
    if (tri) cl_ord = (int*)malloc(sizeof(int) * (size_t)(s->Nc + s->G + 1));
    cl_ptr = (int*)calloc((size_t)n_cand + 2, sizeof(int));
    cl_rb = (int*)malloc(sizeof(int) * (size_t)(s->Nc + s->G + 1));
    cl_ok = (int*)calloc((size_t)n_cand + 1, sizeof(int));
    cl_L = (double**)calloc((size_t)n_cand + 1, sizeof(double*));
    cl_n = (int*)calloc((size_t)n_cand + 1, sizeof(int));
    for (int g = 0; g < n_cand; ++g) {
      int m = cl_ptr[ncl];
      if (tri) {
        for (int a = s->tri_ptr[g]; a < s->tri_ptr[g + 1]; ++a) {
          cl_ord[m] = s->tri_ord[a];
          cl_rb[m++] = s->tri_rb[a];
        }
      } else if (have_shared) {
        if (s->grp_rb[g] < 0) continue;
        for (int c = 0; c < s->Nc; ++c)
          if (s->P->camera_group[c] == g && s->cam_rb[c] >= 0) cl_rb[m++] = s->cam_rb[c];
        cl_rb[m++] = s->grp_rb[g];
      } else {
        for (int b = 0; b < s->nrb; ++b)
          if (s->vis_cluster[b] == g) cl_rb[m++] = b;
        if (m - cl_ptr[ncl] < 2) continue; /* a cluster of one view is its own (whole) block: Minv has it */
      }
      int nc = 0;
      for (int a = cl_ptr[ncl]; a < m; ++a) nc += s->rb_dim[cl_rb[a]];
      /* a cluster is inverted as a DENSE matrix: beyond TMI_BA_MAX_CLUSTER_DIM unknowns its blocks stay SCHUR_JACOBI
       * blocks (engine and oracle alike; Ceres factors the cluster matrices sparsely) */
      if (nc > TMI_BA_MAX_CLUSTER_DIM) continue;
      cl_ptr[ncl + 1] = m;
      cl_n[ncl] = nc;
      double* C = (double*)calloc((size_t)nc * nc, sizeof(double));
      int oi = 0;
      for (int a = cl_ptr[ncl]; a < m; ++a) {
        const int bi = cl_rb[a], ni = s->rb_dim[bi];
        int oj = 0;
        for (int bb = cl_ptr[ncl]; bb <= a; ++bb) {
          const int bj = cl_rb[bb], nj = s->rb_dim[bj];
          const int64_t q = block_lookup(s, bi, bj);
          /* (tridiagonal: the blocks inside a cluster and between neighbouring clusters of the chain only) */
          const int dord = tri ? cl_ord[a] - cl_ord[bb] : 0;
          if (q >= 0 && dord >= -1 && dord <= 1) {
            const double* B = s->S + s->blk_off[q];
            const double sc = dord != 0 ? off_scale : 1.0;
            for (int i = 0; i < ni; ++i)
              for (int j = 0; j < nj; ++j) C[(int64_t)(oi + i) * nc + oj + j] = sc * B[i * nj + j];
          }
          oj += nj;
        }
        oi += ni;
      }
      /* in-place lower Cholesky */
      int pd = 1;
      for (int j = 0; j < nc && pd; ++j) {
        double* Cj = C + (int64_t)j * nc;
        double d = Cj[j];
        for (int k = 0; k < j; ++k) d -= Cj[k] * Cj[k];
        if (!(d > 0.0) || !isfinite(d)) { pd = 0; break; }
        const double ljj = sqrt(d);
        Cj[j] = ljj;
        for (int i = j + 1; i < nc; ++i) {
          double* Ci = C + (int64_t)i * nc;
          double v = Ci[j];
          for (int k = 0; k < j; ++k) v -= Ci[k] * Cj[k];
          Ci[j] = v / ljj;
        }
      }
      cl_ok[ncl] = pd;
      cl_L[ncl] = C;
      ++ncl;
    }
  }
  if (tri) {
    int all_pd = 1;
    for (int c = 0; c < ncl; ++c)
      if (!cl_ok[c]) all_pd = 0;
    if (!all_pd) {
      for (int c = 0; c < ncl; ++c) free(cl_L[c]);
      free(cl_ptr); free(cl_rb); free(cl_ok); free(cl_L); free(cl_n); free(cl_ord);
      cl_ptr = cl_rb = cl_ok = cl_n = cl_ord = NULL;
      cl_L = NULL;
      ncl = 0;
      if (tri_attempt == 0) {
        tri_attempt = 1;
        off_scale = getenv("TMI_BA_TEST_TRI_SCALE1") ? atof(getenv("TMI_BA_TEST_TRI_SCALE1")) : 0.5;
        goto tri_again;
      }
      free(r);
      free(Minv);
      return 0; /* "Preconditioner update failed.": LINEAR_SOLVER_FAILURE, the step is invalid */
    }
  }
  const double* bref = s->rhs;
  memset(x, 0, sizeof(double) * (size_t)n);
  memcpy(r, bref, sizeof(double) * (size_t)n); /* r = b - A*0 */
  const double norm_b = sqrt(dot(bref, bref, n));
  const double tol_r = -1.0 * norm_b; /* r_tolerance = -1: disabled */
  (void)tol_r;
  double rho = 1.0;
  double Q0 = -1.0 * 0.0; /* x = 0 */
  const int max_it = s->O->max_linear_solver_iterations;
  const int min_it = s->O->min_linear_solver_iterations;
  const double q_tol = s->O->eta;
  int it;
  for (it = 1;; ++it) {
    /* z = M^-1 r */
    for (int b = 0; b < s->nrb; ++b) {
      const int nb = s->rb_dim[b], off = s->rb_off[b];
      const double* M = Minv + (int64_t)b * MAXC * MAXC;
      for (int a = 0; a < nb; ++a) {
        double v = 0.0;
        for (int c = 0; c < nb; ++c) v += M[a * nb + c] * r[off + c];
        z[off + a] = v;
      }
    }
    for (int c = 0; c < ncl; ++c) {
      if (!cl_ok[c]) continue;
      const int nc = cl_n[c];
      const double* L = cl_L[c];
      double* y = tmp; /* scratch (free between the residual resets) */
      int o = 0;
      for (int a = cl_ptr[c]; a < cl_ptr[c + 1]; ++a) {
        const int b = cl_rb[a];
        for (int i = 0; i < s->rb_dim[b]; ++i) y[o++] = r[s->rb_off[b] + i];
      }
      for (int i = 0; i < nc; ++i) {
        double v = y[i];
        const double* Li = L + (int64_t)i * nc;
        for (int k = 0; k < i; ++k) v -= Li[k] * y[k];
        y[i] = v / Li[i];
      }
      for (int i = nc - 1; i >= 0; --i) {
        double v = y[i];
        for (int k = i + 1; k < nc; ++k) v -= L[(int64_t)k * nc + i] * y[k];
        y[i] = v / L[(int64_t)i * nc + i];
      }
      o = 0;
      for (int a = cl_ptr[c]; a < cl_ptr[c + 1]; ++a) {
        const int b = cl_rb[a];
        for (int i = 0; i < s->rb_dim[b]; ++i) z[s->rb_off[b] + i] = y[o++];
      }
    }
    const double last_rho = rho;
    rho = dot(r, z, n);
    if (rho == 0.0 || !isfinite(rho)) { ok = 0; break; }
    if (it == 1) {
      memcpy(pvec, z, sizeof(double) * (size_t)n);
    } else {
      const double beta = rho / last_rho;
      if (beta == 0.0 || !isfinite(beta)) { ok = 0; break; }
      for (int i = 0; i < n; ++i) pvec[i] = z[i] + beta * pvec[i];
    }
    double* q = z;
    spmv(s, pvec, q);
    const double pq = dot(pvec, q, n);
    if (pq <= 0.0 || !isfinite(pq)) break; /* LINEAR_SOLVER_NO_CONVERGENCE: x is kept */
    const double alpha = rho / pq;
    if (!isfinite(alpha)) { ok = 0; break; }
    for (int i = 0; i < n; ++i) x[i] += alpha * pvec[i];
    if (it % 10 == 0) { /* residual_reset_period */
      spmv(s, x, tmp);
      for (int i = 0; i < n; ++i) r[i] = bref[i] - tmp[i];
    } else {
      for (int i = 0; i < n; ++i) r[i] -= alpha * q[i];
    }
    /* Q1 = x'Ax - 2 b'x = -x'(b + r) */
    double Q1 = 0.0;
    for (int i = 0; i < n; ++i) Q1 -= x[i] * (bref[i] + r[i]);
    const double zeta = it * (Q1 - Q0) / Q1;
    if (zeta < q_tol && it >= min_it) break;
    Q0 = Q1;
    if (it >= max_it) break;
  }
  s->pcg_iters += it;
  free(r);
  free(Minv);
  for (int c = 0; c < ncl; ++c) free(cl_L[c]);
  free(cl_ptr); free(cl_rb); free(cl_ok); free(cl_L); free(cl_n); free(cl_ord);
  return ok;
}

/* dense Cholesky solve of the reduced system (DENSE_SCHUR / SPARSE_SCHUR /
 * DENSE_QR all solve the same normal equations exactly) */
static int solve_dense(ost* s) {
  const int n = s->nr;
  if (!s->dense) s->dense = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* A = s->dense;
  memset(A, 0, sizeof(double) * (size_t)n * n);
  for (int64_t b = 0; b < s->nblk; ++b) {
    const int bi = s->blk_i[b], bj = s->blk_j[b];
    const int ni = s->rb_dim[bi], nj = s->rb_dim[bj];
    const double* B = s->S + s->blk_off[b];
    for (int a = 0; a < ni; ++a)
      for (int c = 0; c < nj; ++c)
        A[(int64_t)(s->rb_off[bi] + a) * n + s->rb_off[bj] + c] = B[a * nj + c];
  }
  /* row-major lower Cholesky (Cholesky-Crout, dot products over rows) */
  for (int j = 0; j < n; ++j) {
    double* Aj = A + (int64_t)j * n;
    double d = Aj[j];
    for (int k = 0; k < j; ++k) d -= Aj[k] * Aj[k];
    if (!(d > 0.0) || !isfinite(d)) return 0;
    const double ljj = sqrt(d);
    Aj[j] = ljj;
#pragma omp parallel for schedule(static) if (n - j > 256)
    for (int i = j + 1; i < n; ++i) {
      double* Ai = A + (int64_t)i * n;
      double v = Ai[j];
      for (int k = 0; k < j; ++k) v -= Ai[k] * Aj[k];
      Ai[j] = v / ljj;
    }
  }
  double* y = s->yc;
  for (int i = 0; i < n; ++i) {
    double v = s->rhs[i];
    const double* Ai = A + (int64_t)i * n;
    for (int k = 0; k < i; ++k) v -= Ai[k] * y[k];
    y[i] = v / Ai[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double v = y[i];
    for (int k = i + 1; k < n; ++k) v -= A[(int64_t)k * n + i] * y[k];
    y[i] = v / A[(int64_t)i * n + i];
  }
  return 1;
}

/* back substitution y_p = (V+Dp)^-1 (g_p - W^T y_c) and the model cost change
 *   -(J d).(r + J d / 2) with d = -y  (TrustRegionMinimizer) */
static double back_substitute(ost* s) {
  const int dp = s->dp;
  double mcc = 0.0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : mcc)
  for (int p = 0; p < s->Np; ++p) {
    double v[4] = {0, 0, 0, 0};
    for (int a = 0; a < dp; ++a) v[a] = s->gp[(int64_t)p * dp + a];
    for (int64_t k = s->pt_ptr[p]; k < s->pt_ptr[p + 1]; ++k) {
      const int cam = s->P->obs_camera[s->order[k]];
      int rb0, n0, rb1, n1;
      obs_parts(s, cam, &rb0, &n0, &rb1, &n1);
      const double* Jc = s->Jc + 2 * MAXC * k;
      const double* Jp = s->Jp + 8 * k;
      double u0 = 0.0, u1 = 0.0;
      for (int a = 0; a < n0; ++a) {
        u0 += Jc[a] * s->yc[s->rb_off[rb0] + a];
        u1 += Jc[MAXC + a] * s->yc[s->rb_off[rb0] + a];
      }
      for (int a = 0; a < n1; ++a) {
        u0 += Jc[n0 + a] * s->yc[s->rb_off[rb1] + a];
        u1 += Jc[MAXC + n0 + a] * s->yc[s->rb_off[rb1] + a];
      }
      for (int a = 0; a < dp; ++a) v[a] -= Jp[a] * u0 + Jp[4 + a] * u1;
    }
    const double* Vi = s->Vinv + (int64_t)p * dp * dp;
    double* yp = s->yp + (int64_t)p * dp;
    for (int a = 0; a < dp; ++a) {
      double t = 0.0;
      for (int b = 0; b < dp; ++b) t += Vi[a * dp + b] * v[b];
      yp[a] = t;
    }
    /* model residual m = J d = -(Jc yc + Jp yp) per observation */
    for (int64_t k = s->pt_ptr[p]; k < s->pt_ptr[p + 1]; ++k) {
      const int cam = s->P->obs_camera[s->order[k]];
      int rb0, n0, rb1, n1;
      obs_parts(s, cam, &rb0, &n0, &rb1, &n1);
      const double* Jc = s->Jc + 2 * MAXC * k;
      const double* Jp = s->Jp + 8 * k;
      double m0 = 0.0, m1 = 0.0;
      for (int a = 0; a < n0; ++a) {
        m0 += Jc[a] * s->yc[s->rb_off[rb0] + a];
        m1 += Jc[MAXC + a] * s->yc[s->rb_off[rb0] + a];
      }
      for (int a = 0; a < n1; ++a) {
        m0 += Jc[n0 + a] * s->yc[s->rb_off[rb1] + a];
        m1 += Jc[MAXC + n0 + a] * s->yc[s->rb_off[rb1] + a];
      }
      for (int a = 0; a < dp; ++a) {
        m0 += Jp[a] * yp[a];
        m1 += Jp[4 + a] * yp[a];
      }
      m0 = -m0;
      m1 = -m1;
      mcc -= m0 * (s->r[2 * k] + 0.5 * m0) + m1 * (s->r[2 * k + 1] + 0.5 * m1);
    }
  }
  return mcc;
}

static void free_state(ost* s) {
  free(s->ext); free(s->intr); free(s->pts);
  free(s->n_ext); free(s->ext_idx); free(s->n_intr); free(s->intr_idx); free(s->grp_private);
  free(s->rb_dim); free(s->rb_off); free(s->rb_split); free(s->cam_rb); free(s->grp_rb); free(s->vis_cluster);
  free(s->tri_ptr); free(s->tri_rb); free(s->tri_ord);
  free(s->order); free(s->pt_ptr);
  free(s->r); free(s->Jc); free(s->Jp); free(s->Ep);
  free(s->scale_c); free(s->scale_p); free(s->diag_c); free(s->diag_p);
  free(s->Vinv); free(s->gp); free(s->tp);
  if (s->bmap.keys) hmap_free(&s->bmap);
  free(s->blk_i); free(s->blk_j); free(s->blk_off); free(s->row_ptr); free(s->row_blk);
  free(s->rb_obs_ptr); free(s->rb_obs); free(s->k_pt); free(s->row_lpt);
  free(s->S); free(s->gc); free(s->rhs); free(s->yc); free(s->yp); free(s->dense);
}

static int validate(const tmi_ba_problem* P) {
  if (!P || P->num_cameras < 0 || P->num_groups < 0 || P->num_points < 0 ||
      P->num_observations < 0)
    return 0;
  if (P->num_cameras && (!P->extrinsics || !P->camera_group)) return 0;
  if (P->num_groups && (!P->group_model || !P->group_offset || !P->intrinsics)) return 0;
  if (P->num_points && !P->points) return 0;
  if (P->num_observations && (!P->obs_camera || !P->obs_point || !P->obs_xy)) return 0;
  for (int c = 0; c < P->num_cameras; ++c)
    if (P->camera_group[c] < 0 || P->camera_group[c] >= P->num_groups) return 0;
  for (int g = 0; g < P->num_groups; ++g) {
    const int n = model_size(P->group_model[g]);
    if (n < 0 || P->group_offset[g + 1] - P->group_offset[g] != n) return 0;
  }
  for (int64_t i = 0; i < P->num_observations; ++i)
    if (P->obs_camera[i] < 0 || P->obs_camera[i] >= P->num_cameras || P->obs_point[i] < 0 ||
        P->obs_point[i] >= P->num_points)
      return 0;
  return 1;
}

static double state_norm(const ost* s) {
  /* ||x|| over every coordinate of every non-constant parameter block */
  double v = 0.0;
  for (int c = 0; c < s->Nc; ++c)
    if (s->n_ext[c] > 0)
      for (int a = 0; a < 6; ++a) v += s->ext[6 * c + a] * s->ext[6 * c + a];
  for (int g = 0; g < s->G; ++g)
    if (s->n_intr[g] > 0)
      for (int a = s->P->group_offset[g]; a < s->P->group_offset[g + 1]; ++a)
        v += s->intr[a] * s->intr[a];
  for (int p = 0; p < s->Np; ++p)
    if (!s->P->point_constant || !s->P->point_constant[p])
      for (int a = 0; a < 4; ++a) v += s->pts[4 * (int64_t)p + a] * s->pts[4 * (int64_t)p + a];
  return sqrt(v);
}

int32_t oracle_adjust_tracks(tmi_ba_problem* P, const tmi_ba_options* O, int8_t* termination,
                             int32_t* iterations, double* initial_cost, double* final_cost);

/* ---- inner iterations -------------------------------------------------------------
 * Ceres' CoordinateDescentMinimizer (coordinate_descent_minimizer.cc, 1.14) as the
 * TrustRegionMinimizer calls it from DoInnerIterationsIfNeeded when
 * Solver::Options::use_inner_iterations is set (Theia: BundleAdjustmentOptions::
 * use_inner_iterations = true, bundle_adjustment.h:112; bundle_adjuster.cc:74).  Ceres is
 * external to the reference: restated from the 1.14 sources' documented behaviour.
 *
 * Theia sets the ordering explicitly (bundle_adjuster.cc:193-200): a copy of the linear
 * solver ordering -- group 0 tracks, 1 intrinsics, 2 extrinsics (bundle_adjuster.cc:346-371)
 * -- with ParameterBlockOrdering::Reverse() applied, so the coordinate descent visits the
 * groups as  extrinsics blocks -> intrinsics blocks -> points.  Each group is an independent
 * set (a residual touches one extrinsics, one intrinsics and one point block), which is what
 * Ceres requires of a user-supplied inner_iteration_ordering.
 *
 * Every block of a set is minimised on its own with all other blocks constant:
 * TrustRegionMinimizer with default Minimizer::Options (50 iterations, function / gradient /
 * parameter tolerance 1e-6 / 1e-10 / 1e-8, LM radius 1e4 <= 1e16, DENSE_QR, Jacobi scaling),
 * the residual blocks keep their loss function. */
static void inner_options(const tmi_ba_options* O, tmi_ba_options* o2) {
  tmi_ba_options d;
  memset(&d, 0, sizeof(d));
  d.loss_function_type = O->loss_function_type;
  d.robust_loss_width = O->robust_loss_width;
  d.linear_solver_type = TMI_BA_DENSE_QR;
  d.preconditioner_type = O->preconditioner_type;
  d.num_threads = 1;
  d.max_num_iterations = 50;
  d.max_solver_time_in_seconds = 1e9;
  d.use_inner_iterations = 0;
  d.function_tolerance = 1e-6;
  d.gradient_tolerance = 1e-10;
  d.parameter_tolerance = 1e-8;
  d.max_trust_region_radius = 1e16;
  d.initial_trust_region_radius = 1e4;
  d.min_trust_region_radius = 1e-32;
  d.min_relative_decrease = 1e-3;
  d.min_lm_diagonal = 1e-6;
  d.max_lm_diagonal = 1e32;
  d.eta = 0.1;
  d.max_linear_solver_iterations = 500;
  d.max_num_consecutive_invalid_steps = 5;
  d.jacobi_scaling = 1;
  d.point_dof = O->point_dof;
  d.device = -1;
  d.residual_precision = 64;
  *o2 = d;
}

/* Minimise over ONE camera-side block: the intrinsics of group `g` (kind 1; the views
 * listed in cams[] share it) or the extrinsics of camera cams[0] (kind 0).  V: the
 * problem whose parameter arrays hold the current inner iterate (updated on success). */
static void inner_solve_camera_block(const tmi_ba_problem* V, const tmi_ba_options* o2, int kind,
                                     int g, const int* cams, int ncams, const int64_t* cam_ptr,
                                     const int64_t* cam_obs) {
  int64_t nobs = 0;
  for (int a = 0; a < ncams; ++a) nobs += cam_ptr[cams[a] + 1] - cam_ptr[cams[a]];
  if (nobs == 0) return;
  const int n_intr = V->group_offset[g + 1] - V->group_offset[g];
  double* ext = (double*)malloc(sizeof(double) * 6 * (size_t)ncams);
  int32_t* cgrp = (int32_t*)calloc((size_t)ncams, sizeof(int32_t));
  uint8_t* cflag = (uint8_t*)malloc((size_t)ncams);
  int32_t gmodel = V->group_model[g];
  int32_t goff[2] = {0, n_intr};
  double intr[10];
  uint8_t iconst[10];
  memcpy(intr, V->intrinsics + V->group_offset[g], sizeof(double) * (size_t)n_intr);
  for (int a = 0; a < n_intr; ++a)
    iconst[a] = (kind == 1) ? (V->intrinsics_constant ? V->intrinsics_constant[V->group_offset[g] + a] : 0) : 1;
  double* pts = (double*)malloc(sizeof(double) * 4 * (size_t)nobs);
  uint8_t* pconst = (uint8_t*)malloc((size_t)nobs);
  int32_t* ocam = (int32_t*)malloc(sizeof(int32_t) * (size_t)nobs);
  int32_t* opt = (int32_t*)malloc(sizeof(int32_t) * (size_t)nobs);
  double* oxy = (double*)malloc(sizeof(double) * 2 * (size_t)nobs);
  int64_t k = 0;
  for (int a = 0; a < ncams; ++a) {
    const int c = cams[a];
    memcpy(ext + 6 * a, V->extrinsics + 6 * (size_t)c, sizeof(double) * 6);
    cflag[a] = (kind == 0) ? (V->camera_flags ? V->camera_flags[c] : 0)
                           : (uint8_t)(TMI_BA_CAMERA_POSITION_CONSTANT | TMI_BA_CAMERA_ORIENTATION_CONSTANT);
    for (int64_t q = cam_ptr[c]; q < cam_ptr[c + 1]; ++q, ++k) {
      const int64_t o = cam_obs[q];
      /* every observation brings its own (constant) copy of the point: no index map needed */
      memcpy(pts + 4 * k, V->points + 4 * (size_t)V->obs_point[o], sizeof(double) * 4);
      pconst[k] = 1;
      ocam[k] = a;
      opt[k] = (int32_t)k;
      oxy[2 * k] = V->obs_xy[2 * o];
      oxy[2 * k + 1] = V->obs_xy[2 * o + 1];
    }
  }
  tmi_ba_problem Q;
  memset(&Q, 0, sizeof(Q));
  Q.num_cameras = ncams; Q.extrinsics = ext; Q.camera_group = cgrp; Q.camera_flags = cflag;
  Q.num_groups = 1; Q.group_model = &gmodel; Q.group_offset = goff; Q.intrinsics = intr;
  Q.intrinsics_constant = iconst;
  Q.num_points = (int32_t)nobs; Q.points = pts; Q.point_constant = pconst;
  Q.num_observations = nobs; Q.obs_camera = ocam; Q.obs_point = opt; Q.obs_xy = oxy;
  tmi_ba_summary sm;
  oracle_ba_solve(&Q, o2, &sm);
  if (sm.success) {
    if (kind == 0)
      memcpy(V->extrinsics + 6 * (size_t)cams[0], ext, sizeof(double) * 6);
    else
      memcpy(V->intrinsics + V->group_offset[g], intr, sizeof(double) * (size_t)n_intr);
  }
  free(ext); free(cgrp); free(cflag); free(pts); free(pconst); free(ocam); free(opt); free(oxy);
}

/* Test hook: 0 = the reference's order (extrinsics, intrinsics, points); 1 = intrinsics before
 * extrinsics (the order a round-1 misreading used; kept only so tests can show that the two
 * orders are distinguishable and that the device follows the reference's). */
static int g_inner_order = 0;
void oracle_set_inner_order(int32_t order) { g_inner_order = order ? 1 : 0; }

/* One sweep of the coordinate descent over the parameter arrays of V (in place). */
static void inner_iterations(tmi_ba_problem* V, const tmi_ba_options* O) {
  tmi_ba_options o2;
  inner_options(O, &o2);
  const int Nc = V->num_cameras, G = V->num_groups;
  const int64_t No = V->num_observations;
  int64_t* cam_ptr = (int64_t*)calloc((size_t)Nc + 2, sizeof(int64_t));
  int64_t* cam_obs = (int64_t*)malloc(sizeof(int64_t) * (size_t)(No > 0 ? No : 1));
  for (int64_t o = 0; o < No; ++o) cam_ptr[V->obs_camera[o] + 2]++;
  for (int c = 0; c < Nc; ++c) cam_ptr[c + 2] += cam_ptr[c + 1];
  for (int64_t o = 0; o < No; ++o) cam_obs[cam_ptr[V->obs_camera[o] + 1]++] = o;
  /* views of each group */
  int* gptr = (int*)calloc((size_t)G + 2, sizeof(int));
  int* gcam = (int*)malloc(sizeof(int) * (size_t)(Nc > 0 ? Nc : 1));
  for (int c = 0; c < Nc; ++c) gptr[V->camera_group[c] + 2]++;
  for (int g = 0; g < G; ++g) gptr[g + 2] += gptr[g + 1];
  for (int c = 0; c < Nc; ++c) gcam[gptr[V->camera_group[c] + 1]++] = c;
  for (int pass = 0; pass < 2; ++pass) {
    const int kind = g_inner_order ? 1 - pass : pass;  /* 0 extrinsics, 1 intrinsics */
    if (kind == 0) {
      /* set 1: extrinsics blocks with a free coordinate */
#pragma omp parallel for schedule(dynamic, 1)
      for (int c = 0; c < Nc; ++c) {
        const int f = V->camera_flags ? V->camera_flags[c] : 0;
        if ((f & TMI_BA_CAMERA_POSITION_CONSTANT) && (f & TMI_BA_CAMERA_ORIENTATION_CONSTANT)) continue;
        inner_solve_camera_block(V, &o2, 0, V->camera_group[c], &c, 1, cam_ptr, cam_obs);
      }
    } else {
      /* set 2: intrinsics blocks with a free coordinate */
#pragma omp parallel for schedule(dynamic, 1)
      for (int g = 0; g < G; ++g) {
        int nfree = 0;
        for (int a = V->group_offset[g]; a < V->group_offset[g + 1]; ++a)
          nfree += !(V->intrinsics_constant && V->intrinsics_constant[a]);
        if (nfree == 0 || gptr[g + 1] == gptr[g]) continue;
        inner_solve_camera_block(V, &o2, 1, g, gcam + gptr[g], gptr[g + 1] - gptr[g], cam_ptr, cam_obs);
      }
    }
  }
  /* set 3: the points, each against its (now constant) cameras */
  oracle_adjust_tracks(V, &o2, NULL, NULL, NULL, NULL);
  free(cam_ptr); free(cam_obs); free(gptr); free(gcam);
}

/* One coordinate-descent sweep at the problem's current parameters (in place); test entry. */
int32_t oracle_inner_sweep(tmi_ba_problem* P, const tmi_ba_options* O) {
  if (!O || !validate(P)) return TMI_BA_ERR_INVALID_ARGUMENT;
  inner_iterations(P, O);
  return TMI_BA_OK;
}

int32_t oracle_ba_solve(tmi_ba_problem* P, const tmi_ba_options* O, tmi_ba_summary* sum) {
  if (!O || !sum) return TMI_BA_ERR_INVALID_ARGUMENT;
  memset(sum, 0, sizeof(*sum));
  sum->termination = 2;
  if (!validate(P)) {
    sum->status = TMI_BA_ERR_INVALID_ARGUMENT;
    snprintf(sum->message, sizeof(sum->message), "invalid problem");
    return sum->status;
  }
  const double t_start = now_s();
  ost S_;
  ost* s = &S_;
  memset(s, 0, sizeof(*s));
  s->P = P;
  s->O = O;
  s->Nc = P->num_cameras;
  s->G = P->num_groups;
  s->Np = P->num_points;
  s->No = P->num_observations;
  s->dp = (O->point_dof == 3) ? 3 : 4;
  const int dp = s->dp;
  const int n_intr_total = s->G ? P->group_offset[s->G] : 0;
  s->ext = (double*)malloc(sizeof(double) * 6 * (size_t)(s->Nc + 1));
  s->intr = (double*)malloc(sizeof(double) * (size_t)(n_intr_total + 1));
  s->pts = (double*)malloc(sizeof(double) * 4 * (size_t)(s->Np + 1));
  memcpy(s->ext, P->extrinsics, sizeof(double) * 6 * (size_t)s->Nc);
  memcpy(s->intr, P->intrinsics, sizeof(double) * (size_t)n_intr_total);
  memcpy(s->pts, P->points, sizeof(double) * 4 * (size_t)s->Np);

  /* free-column maps (SubsetParameterization semantics,
   * bundle_adjuster.cc:242-334) */
  s->n_ext = (int*)calloc((size_t)s->Nc + 1, sizeof(int));
  s->ext_idx = (int*)calloc(6 * (size_t)s->Nc + 1, sizeof(int));
  s->n_intr = (int*)calloc((size_t)s->G + 1, sizeof(int));
  s->intr_idx = (int*)calloc(10 * (size_t)s->G + 1, sizeof(int));
  s->grp_private = (int*)calloc((size_t)s->G + 1, sizeof(int));
  int* grp_count = (int*)calloc((size_t)s->G + 1, sizeof(int));
  for (int c = 0; c < s->Nc; ++c) {
    const int f = P->camera_flags ? P->camera_flags[c] : 0;
    int n = 0;
    if (!(f & TMI_BA_CAMERA_POSITION_CONSTANT))
      for (int a = 0; a < 3; ++a) s->ext_idx[6 * c + n++] = a;
    if (!(f & TMI_BA_CAMERA_ORIENTATION_CONSTANT))
      for (int a = 3; a < 6; ++a) s->ext_idx[6 * c + n++] = a;
    s->n_ext[c] = n;
    grp_count[P->camera_group[c]]++;
  }
  for (int g = 0; g < s->G; ++g) {
    int n = 0;
    for (int a = P->group_offset[g]; a < P->group_offset[g + 1]; ++a)
      if (!P->intrinsics_constant || !P->intrinsics_constant[a])
        s->intr_idx[10 * g + n++] = a - P->group_offset[g];
    s->n_intr[g] = n;
    s->grp_private[g] = (grp_count[g] == 1);
  }
  free(grp_count);
  /* reduced blocks: one merged [ext | intr] block per camera whose group is
   * private, else an extrinsics block per camera plus one block per shared
   * group */
  s->cam_rb = (int*)malloc(sizeof(int) * (size_t)(s->Nc + 1));
  s->grp_rb = (int*)malloc(sizeof(int) * (size_t)(s->G + 1));
  s->rb_dim = (int*)malloc(sizeof(int) * (size_t)(s->Nc + s->G + 1));
  s->rb_off = (int*)malloc(sizeof(int) * (size_t)(s->Nc + s->G + 2));
  s->rb_split = (int*)malloc(sizeof(int) * (size_t)(s->Nc + s->G + 1));
  s->nrb = 0;
  s->nr = 0;
  for (int c = 0; c < s->Nc; ++c) {
    const int g = P->camera_group[c];
    const int d = s->n_ext[c] + (s->grp_private[g] ? s->n_intr[g] : 0);
    if (d > 0) {
      s->cam_rb[c] = s->nrb;
      s->rb_dim[s->nrb] = d;
      s->rb_split[s->nrb] = s->n_ext[c];
      s->rb_off[s->nrb] = s->nr;
      s->nr += d;
      s->nrb++;
    } else {
      s->cam_rb[c] = -1;
    }
  }
  for (int g = 0; g < s->G; ++g) {
    if (!s->grp_private[g] && s->n_intr[g] > 0) {
      s->grp_rb[g] = s->nrb;
      s->rb_dim[s->nrb] = s->n_intr[g];
      s->rb_split[s->nrb] = 0;
      s->rb_off[s->nrb] = s->nr;
      s->nr += s->n_intr[g];
      s->nrb++;
    } else {
      s->grp_rb[g] = -1;
    }
  }
  s->rb_off[s->nrb] = s->nr;

  /* sort observations by point (counting sort, stable) */
  s->order = (int64_t*)malloc(sizeof(int64_t) * (size_t)(s->No + 1));
  s->pt_ptr = (int64_t*)calloc((size_t)s->Np + 2, sizeof(int64_t));
  for (int64_t i = 0; i < s->No; ++i) s->pt_ptr[P->obs_point[i] + 1]++;
  for (int p = 0; p < s->Np; ++p) s->pt_ptr[p + 1] += s->pt_ptr[p];
  {
    int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)(s->Np + 1));
    memcpy(fill, s->pt_ptr, sizeof(int64_t) * (size_t)s->Np);
    for (int64_t i = 0; i < s->No; ++i) s->order[fill[P->obs_point[i]]++] = i;
    free(fill);
  }
  s->r = (double*)calloc(2 * (size_t)s->No + 2, sizeof(double));
  s->Jc = (double*)calloc(2 * MAXC * (size_t)s->No + 2, sizeof(double));
  s->Jp = (double*)calloc(8 * (size_t)s->No + 2, sizeof(double));
  s->Ep = (double*)calloc(8 * (size_t)s->No + 2, sizeof(double));
  s->scale_c = (double*)malloc(sizeof(double) * (size_t)(s->nr + 1));
  s->scale_p = (double*)malloc(sizeof(double) * ((size_t)s->Np * dp + 1));
  s->diag_c = (double*)malloc(sizeof(double) * (size_t)(s->nr + 1));
  s->diag_p = (double*)malloc(sizeof(double) * ((size_t)s->Np * dp + 1));
  s->Vinv = (double*)malloc(sizeof(double) * ((size_t)s->Np * dp * dp + 1));
  s->gp = (double*)malloc(sizeof(double) * ((size_t)s->Np * dp + 1));
  s->tp = (double*)malloc(sizeof(double) * ((size_t)s->Np * dp + 1));
  s->gc = (double*)malloc(sizeof(double) * (size_t)(s->nr + 1));
  s->rhs = (double*)malloc(sizeof(double) * (size_t)(s->nr + 1));
  s->yc = (double*)calloc((size_t)s->nr + 1, sizeof(double));
  s->yp = (double*)calloc((size_t)s->Np * dp + 1, sizeof(double));
  for (int i = 0; i < s->nr; ++i) s->scale_c[i] = 1.0;
  for (int64_t i = 0; i < (int64_t)s->Np * dp; ++i) s->scale_p[i] = 1.0;
  double ot_ = now_s();
  build_structure(s);
  OT_LAP("build_structure");
  sum->num_reduced_blocks = s->nrb;
  sum->reduced_block_dim = 0;
  for (int b = 0; b < s->nrb; ++b)
    if (s->rb_dim[b] > sum->reduced_block_dim) sum->reduced_block_dim = s->rb_dim[b];
  sum->num_schur_blocks = (s->nblk + s->nrb) / 2;
  sum->setup_time_in_seconds = now_s() - t_start;
  const double t_solve = now_s();

  const int iterative =
      (O->linear_solver_type == TMI_BA_ITERATIVE_SCHUR || O->linear_solver_type == TMI_BA_CGNR);

  /* ---- iteration zero (TrustRegionMinimizer::IterationZero) ---- */
  double cost, ss;
  int64_t bad = evaluate(s, 1, 0, &cost, &ss);
  OT_LAP("evaluate (jets)");
  if (bad) {
    sum->status = TMI_BA_ERR_EVALUATION_FAILED;
    snprintf(sum->message, sizeof(sum->message),
             "residual evaluation failed at the start point (%lld observations)", (long long)bad);
    free_state(s);
    return sum->status;
  }
  sum->initial_cost = cost;
  sum->initial_rmse = s->No ? sqrt(ss / (double)s->No) : 0.0;
  /* gradient of the unscaled problem */
  double* gfull_p = (double*)malloc(sizeof(double) * ((size_t)s->Np * dp + 1));
  gradient(s, s->gc, gfull_p);
  double gmax = 0.0;
  for (int i = 0; i < s->nr; ++i) gmax = fmax(gmax, fabs(s->gc[i]));
  for (int64_t i = 0; i < (int64_t)s->Np * dp; ++i) gmax = fmax(gmax, fabs(gfull_p[i]));
  if (O->jacobi_scaling) {
    /* jacobian_scaling = 1 / (1 + sqrt(squared column norm)), computed once */
    column_sqnorms(s, s->diag_c, s->diag_p);
    for (int i = 0; i < s->nr; ++i) s->scale_c[i] = 1.0 / (1.0 + sqrt(s->diag_c[i]));
    for (int64_t i = 0; i < (int64_t)s->Np * dp; ++i)
      s->scale_p[i] = 1.0 / (1.0 + sqrt(s->diag_p[i]));
    OT_LAP("gradient + norms");
    evaluate(s, 1, 1, &cost, &ss);
    OT_LAP("evaluate (scaled)");
  }
  double x_norm = state_norm(s);
  double radius = O->initial_trust_region_radius;
  double decrease_factor = 2.0;
  int reuse_diagonal = 0;
  int invalid_run = 0;
  int inner_enabled = O->use_inner_iterations ? 1 : 0;
  int iter = 0;
  int termination = 1; /* NO_CONVERGENCE unless stated */
  const char* why = "maximum number of iterations reached";
  double* cand_ext = (double*)malloc(sizeof(double) * 6 * (size_t)(s->Nc + 1));
  double* cand_intr = (double*)malloc(sizeof(double) * (size_t)(n_intr_total + 1));
  double* cand_pts = (double*)malloc(sizeof(double) * 4 * (size_t)(s->Np + 1));

  if (gmax <= O->gradient_tolerance) {
    termination = 0;
    why = "gradient tolerance reached at the start point";
  } else {
    for (;;) {
      if (iter >= O->max_num_iterations) break;
      if (now_s() - t_start >= O->max_solver_time_in_seconds) {
        why = "maximum solver time reached";
        break;
      }
      ++iter;
      ot_ = now_s();
      const double radius_used = radius;
      const int64_t pcg_before = s->pcg_iters;
      /* one row of the optional trace (theia_mi355_ba.h, tmi_ba_options.iteration_trace) */
#define ORACLE_TRACE(outcome, cand, mcc, step)                                                   \
      do {                                                                                       \
        if (O->iteration_trace && iter <= O->iteration_trace_capacity) {                         \
          double* row_ = O->iteration_trace + (size_t)(iter - 1) * TMI_BA_TRACE_STRIDE;          \
          row_[0] = iter; row_[1] = cost; row_[2] = radius_used; row_[3] = (outcome);            \
          row_[4] = (cand); row_[5] = (mcc); row_[6] = (double)(s->pcg_iters - pcg_before);      \
          row_[7] = (step);                                                                      \
        }                                                                                        \
      } while (0)
      if (!reuse_diagonal) column_sqnorms(s, s->diag_c, s->diag_p);
      OT_LAP("column norms");
      int step_ok = build_reduced(s, radius);
      OT_LAP("build_reduced");
      if (step_ok) step_ok = iterative ? solve_pcg(s) : solve_dense(s);
      OT_LAP("linear solve");
      double model_cost_change = 0.0;
      if (step_ok) {
        model_cost_change = back_substitute(s);
        OT_LAP("back_substitute");
        if (!(model_cost_change > 0.0)) step_ok = 0;
      }
      if (!step_ok) {
        /* HandleInvalidStep */
        ORACLE_TRACE(-1.0, NAN, model_cost_change, 0.0);
        if (++invalid_run >= O->max_num_consecutive_invalid_steps) {
          termination = 2;
          why = "too many consecutive invalid steps";
          break;
        }
        radius /= decrease_factor;
        decrease_factor *= 2.0;
        reuse_diagonal = 1;
        sum->num_unsuccessful_steps++;
        if (radius < O->min_trust_region_radius) {
          termination = 0;
          why = "minimum trust region radius reached";
          break;
        }
        continue;
      }
      invalid_run = 0;
      /* candidate = x + scale .* (-y) on the free coordinates */
      memcpy(cand_ext, s->ext, sizeof(double) * 6 * (size_t)s->Nc);
      memcpy(cand_intr, s->intr, sizeof(double) * (size_t)n_intr_total);
      memcpy(cand_pts, s->pts, sizeof(double) * 4 * (size_t)s->Np);
      double step_sq = 0.0;
      for (int c = 0; c < s->Nc; ++c) {
        const int rb = s->cam_rb[c];
        if (rb < 0) continue;
        const int g = P->camera_group[c];
        int col = 0;
        for (int a = 0; a < s->n_ext[c]; ++a, ++col) {
          const double d = -s->yc[s->rb_off[rb] + col] * s->scale_c[s->rb_off[rb] + col];
          cand_ext[6 * c + s->ext_idx[6 * c + a]] += d;
          step_sq += d * d;
        }
        if (s->grp_private[g])
          for (int a = 0; a < s->n_intr[g]; ++a, ++col) {
            const double d = -s->yc[s->rb_off[rb] + col] * s->scale_c[s->rb_off[rb] + col];
            cand_intr[P->group_offset[g] + s->intr_idx[10 * g + a]] += d;
            step_sq += d * d;
          }
      }
      for (int g = 0; g < s->G; ++g) {
        const int rb = s->grp_rb[g];
        if (rb < 0) continue;
        for (int a = 0; a < s->n_intr[g]; ++a) {
          const double d = -s->yc[s->rb_off[rb] + a] * s->scale_c[s->rb_off[rb] + a];
          cand_intr[P->group_offset[g] + s->intr_idx[10 * g + a]] += d;
          step_sq += d * d;
        }
      }
      for (int p = 0; p < s->Np; ++p) {
        if (P->point_constant && P->point_constant[p]) continue;
        for (int a = 0; a < dp; ++a) {
          const double d = -s->yp[(int64_t)p * dp + a] * s->scale_p[(int64_t)p * dp + a];
          cand_pts[4 * (int64_t)p + a] += d;
          step_sq += d * d;
        }
      }
      /* evaluate the candidate cost */
      double *sv_e = s->ext, *sv_i = s->intr, *sv_p = s->pts;
      s->ext = cand_ext;
      s->intr = cand_intr;
      s->pts = cand_pts;
      double cand_cost, cand_ss;
      const int64_t cbad = evaluate(s, 0, 0, &cand_cost, &cand_ss);
      s->ext = sv_e;
      s->intr = sv_i;
      s->pts = sv_p;
      if (cbad) cand_cost = DBL_MAX;
      /* DoInnerIterationsIfNeeded (trust_region_minimizer.cc) */
      int inner_useful = 0;
      if (inner_enabled && cand_cost < DBL_MAX) {
        double* b_ext = (double*)malloc(sizeof(double) * 6 * (size_t)(s->Nc + 1));
        double* b_intr = (double*)malloc(sizeof(double) * (size_t)(n_intr_total + 1));
        double* b_pts = (double*)malloc(sizeof(double) * 4 * (size_t)(s->Np + 1));
        memcpy(b_ext, cand_ext, sizeof(double) * 6 * (size_t)s->Nc);
        memcpy(b_intr, cand_intr, sizeof(double) * (size_t)n_intr_total);
        memcpy(b_pts, cand_pts, sizeof(double) * 4 * (size_t)s->Np);
        tmi_ba_problem V = *P;
        V.extrinsics = cand_ext;
        V.intrinsics = cand_intr;
        V.points = cand_pts;
        inner_iterations(&V, O);
        s->ext = cand_ext;
        s->intr = cand_intr;
        s->pts = cand_pts;
        double inner_cost, inner_ss;
        const int64_t ibad = evaluate(s, 0, 0, &inner_cost, &inner_ss);
        s->ext = sv_e;
        s->intr = sv_i;
        s->pts = sv_p;
        if (ibad) {
          /* "Inner iteration failed": the trust-region candidate stands */
          memcpy(cand_ext, b_ext, sizeof(double) * 6 * (size_t)s->Nc);
          memcpy(cand_intr, b_intr, sizeof(double) * (size_t)n_intr_total);
          memcpy(cand_pts, b_pts, sizeof(double) * 4 * (size_t)s->Np);
        } else {
          model_cost_change += cand_cost - inner_cost;
          inner_useful = inner_cost < cost;
          const double progress = 1.0 - inner_cost / cand_cost;
          inner_enabled = progress > 1e-3; /* inner_iteration_tolerance */
          cand_cost = inner_cost;
          /* step norm of the combined step, in the ambient space */
          step_sq = 0.0;
          for (int64_t i = 0; i < 6 * (int64_t)s->Nc; ++i) step_sq += (cand_ext[i] - s->ext[i]) * (cand_ext[i] - s->ext[i]);
          for (int64_t i = 0; i < n_intr_total; ++i) step_sq += (cand_intr[i] - s->intr[i]) * (cand_intr[i] - s->intr[i]);
          for (int64_t i = 0; i < 4 * (int64_t)s->Np; ++i) step_sq += (cand_pts[i] - s->pts[i]) * (cand_pts[i] - s->pts[i]);
          sum->num_inner_iteration_steps++;
        }
        free(b_ext); free(b_intr); free(b_pts);
      }
      /* ParameterToleranceReached */
      const double step_norm = sqrt(step_sq);
      if (step_norm <= O->parameter_tolerance * (x_norm + O->parameter_tolerance)) {
        ORACLE_TRACE(2.0, cand_cost, model_cost_change, step_norm);
        termination = 0;
        why = "parameter tolerance reached";
        break;
      }
      /* FunctionToleranceReached */
      const double cost_change = cost - cand_cost;
      if (fabs(cost_change) <= O->function_tolerance * cost) {
        ORACLE_TRACE(3.0, cand_cost, model_cost_change, step_norm);
        termination = 0;
        why = "function tolerance reached";
        break;
      }
      const double relative_decrease = cost_change / model_cost_change;
      ORACLE_TRACE((inner_useful || relative_decrease > O->min_relative_decrease) ? 1.0 : 0.0, cand_cost,
                   model_cost_change, step_norm);
      /* IsStepSuccessful: a net decrease through the inner iterations also accepts */
      if (inner_useful || relative_decrease > O->min_relative_decrease) {
        /* HandleSuccessfulStep */
        memcpy(s->ext, cand_ext, sizeof(double) * 6 * (size_t)s->Nc);
        memcpy(s->intr, cand_intr, sizeof(double) * (size_t)n_intr_total);
        memcpy(s->pts, cand_pts, sizeof(double) * 4 * (size_t)s->Np);
        x_norm = state_norm(s);
        double c2;
        evaluate(s, 1, 1, &c2, &ss);
        cost = c2;
        /* gradient of the unscaled problem: g_j = (J_scaled^T r)_j / scale_j */
        gradient(s, s->gc, gfull_p);
        gmax = 0.0;
        for (int i = 0; i < s->nr; ++i) gmax = fmax(gmax, fabs(s->gc[i] / s->scale_c[i]));
        for (int64_t i = 0; i < (int64_t)s->Np * dp; ++i)
          gmax = fmax(gmax, fabs(gfull_p[i] / s->scale_p[i]));
        sum->num_successful_steps++;
        radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * relative_decrease - 1.0, 3));
        radius = fmin(O->max_trust_region_radius, radius);
        decrease_factor = 2.0;
        reuse_diagonal = 0;
        if (gmax <= O->gradient_tolerance) {
          termination = 0;
          why = "gradient tolerance reached";
          break;
        }
      } else {
        radius /= decrease_factor;
        decrease_factor *= 2.0;
        reuse_diagonal = 1;
        sum->num_unsuccessful_steps++;
      }
      if (radius < O->min_trust_region_radius) {
        termination = 0;
        why = "minimum trust region radius reached";
        break;
      }
      if (O->verbose)
        fprintf(stderr, "[oracle] it %3d cost %.10e radius %.3e pcg %lld\n", iter, cost, radius,
                (long long)s->pcg_iters);
    }
  }
  free(gfull_p);
  free(cand_ext);
  free(cand_intr);
  free(cand_pts);

  sum->termination = termination;
  sum->num_iterations = iter;
  sum->num_linear_solver_iterations = s->pcg_iters;
  sum->final_cost = cost;
  sum->success = (termination != 2);
  sum->status = (termination == 2) ? TMI_BA_ERR_LINEAR_SOLVER : TMI_BA_OK;
  snprintf(sum->message, sizeof(sum->message), "%s", why);
  if (sum->success) {
    memcpy(P->extrinsics, s->ext, sizeof(double) * 6 * (size_t)s->Nc);
    memcpy(P->intrinsics, s->intr, sizeof(double) * (size_t)n_intr_total);
    memcpy(P->points, s->pts, sizeof(double) * 4 * (size_t)s->Np);
  }
  {
    double c3, rm;
    oracle_ba_cost(P, O, &c3, &rm);
    sum->final_rmse = rm;
  }
  sum->solve_time_in_seconds = now_s() - t_solve;
  free_state(s);
  return sum->status;
}

/* ---- post-BA outlier filter -------------------------------- */
/* SufficientTriangulationAngle, reference triangulation/triangulation.cc:236-250;
 * DegToRad: math/util.h:48-56 */
int32_t oracle_sufficient_triangulation_angle(const double* rays3, int64_t n,
                                              double min_triangulation_angle_degrees) {
  const double cos_of_min_angle = cos(min_triangulation_angle_degrees * (M_PI / 180.0));
  for (int64_t i = 0; i < n; ++i)
    for (int64_t j = i + 1; j < n; ++j) {
      const double d = rays3[3 * i] * rays3[3 * j] + rays3[3 * i + 1] * rays3[3 * j + 1] +
                       rays3[3 * i + 2] * rays3[3 * j + 2];
      if (d < cos_of_min_angle) return 1;
    }
  return 0;
}

/* SetOutlierTracksToUnestimated, reference set_outlier_tracks_to_unestimated.cc:62-133,
 * over the flattened problem (every camera and every track of `P` is "estimated").
 * flag: 0 kept, 1 bad reprojection (mean squared error above the threshold, or a
 * projection behind a camera :101-105), 2 insufficient viewing angle (:120-125).
 * Observations of a track are visited in the caller's order (the reference follows
 * unordered_set iteration; the outcome does not depend on it apart from the rounding
 * of the sum).  counts = {estimated tracks, bad reprojections, insufficient angles}. */
int32_t oracle_filter_outlier_tracks(const tmi_ba_problem* P, double max_inlier_reprojection_error,
                                     double min_triangulation_angle_degrees, uint8_t* flag,
                                     double* mean_sq_error, int64_t counts[3]) {
  const double max_sq = max_inlier_reprojection_error * max_inlier_reprojection_error;
  const int64_t Np = P->num_points, No = P->num_observations;
  int64_t* ptr = (int64_t*)calloc((size_t)Np + 2, sizeof(int64_t));
  int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(No > 0 ? No : 1));
  for (int64_t o = 0; o < No; ++o) ptr[P->obs_point[o] + 2]++;
  for (int64_t p = 0; p < Np; ++p) ptr[p + 2] += ptr[p + 1];
  for (int64_t o = 0; o < No; ++o) idx[ptr[P->obs_point[o] + 1]++] = o;
  int64_t nbad = 0, nangle = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : nbad, nangle)
  for (int64_t p = 0; p < Np; ++p) {
    const double* X = P->points + 4 * p;
    const int64_t b = ptr[p], e = ptr[p + 1];
    double* rays = (double*)malloc(sizeof(double) * 3 * (size_t)(e - b + 1));
    int64_t nr = 0;
    int estimated = 1;
    int64_t nproj = 0;
    double sum = 0.0;
    for (int64_t q = b; q < e; ++q) {
      const int64_t o = idx[q];
      const int cam = P->obs_camera[o];
      const double* ext = P->extrinsics + 6 * (size_t)cam;
      const int g = P->camera_group[cam];
      /* :87-89 ray = hnormalized(point) - position, normalized */
      double r[3] = {X[0] / X[3] - ext[0], X[1] / X[3] - ext[1], X[2] / X[3] - ext[2]};
      const double n2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      if (n2 > 0.0) {
        const double n = sqrt(n2);
        r[0] /= n; r[1] /= n; r[2] /= n;
      }
      rays[3 * nr] = r[0]; rays[3 * nr + 1] = r[1]; rays[3 * nr + 2] = r[2];
      ++nr;
      double px[2];
      const double depth = oracle_project_point(P->group_model[g], ext,
                                                P->intrinsics + P->group_offset[g], X, px);
      if (depth < 0) {
        estimated = 0;
        break;
      }
      const double dx = px[0] - P->obs_xy[2 * o], dy = px[1] - P->obs_xy[2 * o + 1];
      sum += dx * dx + dy * dy;
      ++nproj;
    }
    int f = 0;
    double mean = sum / (double)nproj;
    if (!estimated) {
      f = 1;
    } else if (mean > max_sq) {
      f = 1;
    } else if (!oracle_sufficient_triangulation_angle(rays, nr, min_triangulation_angle_degrees)) {
      f = 2;
    }
    if (f == 1) ++nbad;
    if (f == 2) ++nangle;
    if (flag) flag[p] = (uint8_t)f;
    if (mean_sq_error) mean_sq_error[p] = mean;
    free(rays);
  }
  if (counts) {
    counts[0] = Np;
    counts[1] = nbad;
    counts[2] = nangle;
  }
  free(ptr);
  free(idx);
  return TMI_BA_OK;
}

/* ---- batched BundleAdjustTrack -------------------------------- */
/* theia::BundleAdjustTrack, reference bundle_adjustment.cc:96-107 -> BundleAdjuster::AddTrack
 * (bundle_adjuster.cc:141-180): the residuals of one track, every observing camera's
 * extrinsics and intrinsics constant, the track variable; solved with the LM above.
 * Run once per non-constant track of `P` (the way estimate_track.cc:238-246 calls it), each
 * on its own sub-problem.  termination: 0/1/2 as tmi_ba_summary, 3 = evaluation failed at
 * the start point, -1 = not adjusted (constant or unobserved track). */
int32_t oracle_adjust_tracks(tmi_ba_problem* P, const tmi_ba_options* O, int8_t* termination,
                             int32_t* iterations, double* initial_cost, double* final_cost) {
  if (!validate(P) || !O) return TMI_BA_ERR_INVALID_ARGUMENT;
  const int64_t Np = P->num_points, No = P->num_observations;
  int64_t* ptr = (int64_t*)calloc((size_t)Np + 2, sizeof(int64_t));
  int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * (size_t)(No > 0 ? No : 1));
  for (int64_t o = 0; o < No; ++o) ptr[P->obs_point[o] + 2]++;
  for (int64_t p = 0; p < Np; ++p) ptr[p + 2] += ptr[p + 1];
  for (int64_t o = 0; o < No; ++o) idx[ptr[P->obs_point[o] + 1]++] = o;
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t p = 0; p < Np; ++p) {
    const int64_t b = ptr[p], e = ptr[p + 1];
    const int k = (int)(e - b);
    if (k == 0 || (P->point_constant && P->point_constant[p])) {
      if (termination) termination[p] = -1;
      if (iterations) iterations[p] = 0;
      if (initial_cost) initial_cost[p] = 0.0;
      if (final_cost) final_cost[p] = 0.0;
      continue;
    }
    /* sub-problem: the k observing cameras, one private copy of their intrinsics each */
    double* ext = (double*)malloc(sizeof(double) * 6 * (size_t)k);
    int32_t* cgrp = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);
    uint8_t* cflag = (uint8_t*)malloc((size_t)k);
    int32_t* gmodel = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);
    int32_t* goff = (int32_t*)malloc(sizeof(int32_t) * ((size_t)k + 1));
    double* intr = (double*)malloc(sizeof(double) * 10 * (size_t)k);
    uint8_t* iconst = (uint8_t*)malloc(10 * (size_t)k);
    int32_t* ocam = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);
    int32_t* opt = (int32_t*)calloc((size_t)k, sizeof(int32_t));
    double* oxy = (double*)malloc(sizeof(double) * 2 * (size_t)k);
    double X[4];
    uint8_t pconst = 0;
    memcpy(X, P->points + 4 * p, sizeof(X));
    goff[0] = 0;
    for (int j = 0; j < k; ++j) {
      const int64_t o = idx[b + j];
      const int cam = P->obs_camera[o];
      const int g = P->camera_group[cam];
      const int n = P->group_offset[g + 1] - P->group_offset[g];
      memcpy(ext + 6 * j, P->extrinsics + 6 * (size_t)cam, sizeof(double) * 6);
      cgrp[j] = j;
      cflag[j] = TMI_BA_CAMERA_POSITION_CONSTANT | TMI_BA_CAMERA_ORIENTATION_CONSTANT;
      gmodel[j] = P->group_model[g];
      memcpy(intr + goff[j], P->intrinsics + P->group_offset[g], sizeof(double) * (size_t)n);
      memset(iconst + goff[j], 1, (size_t)n);
      goff[j + 1] = goff[j] + n;
      ocam[j] = j;
      oxy[2 * j] = P->obs_xy[2 * o];
      oxy[2 * j + 1] = P->obs_xy[2 * o + 1];
    }
    tmi_ba_problem Q;
    memset(&Q, 0, sizeof(Q));
    Q.num_cameras = k; Q.extrinsics = ext; Q.camera_group = cgrp; Q.camera_flags = cflag;
    Q.num_groups = k; Q.group_model = gmodel; Q.group_offset = goff; Q.intrinsics = intr;
    Q.intrinsics_constant = iconst;
    Q.num_points = 1; Q.points = X; Q.point_constant = &pconst;
    Q.num_observations = k; Q.obs_camera = ocam; Q.obs_point = opt; Q.obs_xy = oxy;
    tmi_ba_options o2 = *O;
    o2.linear_solver_type = TMI_BA_DENSE_QR; /* bundle_adjustment.cc:100-101 */
    o2.use_inner_iterations = 0;
    o2.verbose = 0;
    tmi_ba_summary sm;
    const int32_t st = oracle_ba_solve(&Q, &o2, &sm);
    int8_t t = (int8_t)sm.termination;
    if (st == TMI_BA_ERR_EVALUATION_FAILED) t = 3;
    if (termination) termination[p] = t;
    if (iterations) iterations[p] = sm.num_iterations;
    if (initial_cost) initial_cost[p] = sm.initial_cost;
    if (final_cost) final_cost[p] = sm.final_cost;
    if (t == 0 || t == 1) memcpy(P->points + 4 * p, X, sizeof(X));
    free(ext); free(cgrp); free(cflag); free(gmodel); free(goff); free(intr); free(iconst);
    free(ocam); free(opt); free(oxy);
  }
  free(ptr);
  free(idx);
  return TMI_BA_OK;
}

/* ---- pre-BA track sub-sampling -------------------------------- */
/* SelectGoodTracksForBundleAdjustment, reference
 * select_good_tracks_for_bundle_adjustment.cc:81-327, over the flattened problem (every
 * view and track estimated).  The reference iterates unordered containers; this
 * restatement fixes the orders it leaves open -- views ascending, ties of the grid-cell
 * minimum to the smaller track index -- and keeps its comparators as written:
 *   - a grid cell keeps the MINIMUM of (truncated track length, mean squared reprojection
 *     error) (CompareGridCellElements :65-69 with std::min_element :187-190);
 *   - the per-view top-up ranks candidates with std::pair's default operator< on
 *     (TrackId, statistics), i.e. by track id (:236-247).
 * stats_len / stats_err (optional outputs): the track statistics (:81-110). */
typedef struct { int len; double err; } trk_stat;

static int stat_less(const trk_stat* a, const trk_stat* b) {
  if (a->len != b->len) return a->len < b->len;
  return a->err < b->err;
}

typedef struct { int64_t key; int track; } cell_ent;

static int cmp_i32(const void* a, const void* b) {
  const int x = *(const int*)a, y = *(const int*)b;
  return (x > y) - (x < y);
}

int32_t oracle_select_good_tracks(const tmi_ba_problem* P, int32_t long_track_length_threshold,
                                  int32_t image_grid_cell_size_pixels,
                                  int32_t min_num_optimized_tracks_per_view,
                                  const uint8_t* view_mask, uint8_t* selected,
                                  int32_t* stats_len, double* stats_err) {
  if (!validate(P) || !selected || image_grid_cell_size_pixels <= 0) return TMI_BA_ERR_INVALID_ARGUMENT;
  const int64_t Np = P->num_points, No = P->num_observations;
  const int Nc = P->num_cameras;
  trk_stat* st = (trk_stat*)calloc((size_t)Np + 1, sizeof(trk_stat));
  /* ComputeStatisticsForTrack :81-110: every observation counts, no cheirality test */
  {
    double* sum = (double*)calloc((size_t)Np + 1, sizeof(double));
    int* cnt = (int*)calloc((size_t)Np + 1, sizeof(int));
    for (int64_t o = 0; o < No; ++o) {
      const int cam = P->obs_camera[o], p = P->obs_point[o];
      const int g = P->camera_group[cam];
      double px[2];
      oracle_project_point(P->group_model[g], P->extrinsics + 6 * (size_t)cam,
                           P->intrinsics + P->group_offset[g], P->points + 4 * (size_t)p, px);
      const double dx = px[0] - P->obs_xy[2 * o], dy = px[1] - P->obs_xy[2 * o + 1];
      sum[p] += dx * dx + dy * dy;
      cnt[p]++;
    }
    for (int64_t p = 0; p < Np; ++p) {
      st[p].len = cnt[p] < long_track_length_threshold ? cnt[p] : long_track_length_threshold;
      st[p].err = sum[p] / (double)cnt[p];
      if (stats_len) stats_len[p] = st[p].len;
      if (stats_err) stats_err[p] = st[p].err;
    }
    free(sum);
    free(cnt);
  }
  memset(selected, 0, (size_t)Np);
  /* observations by view */
  int64_t* vptr = (int64_t*)calloc((size_t)Nc + 2, sizeof(int64_t));
  int64_t* vobs = (int64_t*)malloc(sizeof(int64_t) * (size_t)(No > 0 ? No : 1));
  for (int64_t o = 0; o < No; ++o) vptr[P->obs_camera[o] + 2]++;
  for (int c = 0; c < Nc; ++c) vptr[c + 2] += vptr[c + 1];
  for (int64_t o = 0; o < No; ++o) vobs[vptr[P->obs_camera[o] + 1]++] = o;
  const double inv_grid_cell_size = 1.0 / image_grid_cell_size_pixels; /* :158 */
  /* SelectBestTracksFromEachImageGridCell :150-196 */
  for (int c = 0; c < Nc; ++c) {
    if (view_mask && !view_mask[c]) continue; /* the overload with a view set :280-327 */
    const int64_t b = vptr[c], e = vptr[c + 1];
    const int64_t n = e - b;
    if (n == 0) continue;
    int64_t cap = 16;
    while (cap < 2 * n) cap *= 2;
    cell_ent* tab = (cell_ent*)malloc(sizeof(cell_ent) * (size_t)cap);
    for (int64_t i = 0; i < cap; ++i) tab[i].track = -1;
    for (int64_t q = b; q < e; ++q) {
      const int64_t o = vobs[q];
      const int t = P->obs_point[o];
      const int cx = (int)(P->obs_xy[2 * o] * inv_grid_cell_size);     /* .cast<int>() :174 */
      const int cy = (int)(P->obs_xy[2 * o + 1] * inv_grid_cell_size);
      const int64_t key = ((int64_t)cx << 32) ^ (int64_t)(uint32_t)cy;
      uint64_t h = (uint64_t)key * 0x9E3779B97F4A7C15ULL;
      int64_t i = (int64_t)(h >> 20) & (cap - 1);
      while (tab[i].track >= 0 && tab[i].key != key) i = (i + 1) & (cap - 1);
      if (tab[i].track < 0) {
        tab[i].key = key;
        tab[i].track = t;
      } else {
        const int cur = tab[i].track;
        if (stat_less(&st[t], &st[cur]) || (!stat_less(&st[cur], &st[t]) && t < cur)) tab[i].track = t;
      }
    }
    for (int64_t i = 0; i < cap; ++i)
      if (tab[i].track >= 0) selected[tab[i].track] = 1;
    free(tab);
  }
  /* SelectTopRankedTracksInView :201-249, views in ascending order */
  for (int c = 0; c < Nc; ++c) {
    if (view_mask && !view_mask[c]) continue;
    const int64_t b = vptr[c], e = vptr[c + 1];
    int num_optimized = 0, num_estimated = (int)(e - b);
    for (int64_t q = b; q < e; ++q) num_optimized += selected[P->obs_point[vobs[q]]];
    if (num_optimized >= min_num_optimized_tracks_per_view) continue; /* the early return :224-226 */
    if (num_optimized == num_estimated) continue;
    int needed = min_num_optimized_tracks_per_view - num_optimized;
    if (needed > num_estimated - num_optimized) needed = num_estimated - num_optimized;
    int* cand = (int*)malloc(sizeof(int) * (size_t)(num_estimated + 1));
    int nc = 0;
    for (int64_t q = b; q < e; ++q) {
      const int t = P->obs_point[vobs[q]];
      if (!selected[t]) cand[nc++] = t;
    }
    qsort(cand, (size_t)nc, sizeof(int), cmp_i32); /* pair<TrackId, ...> default order */
    for (int i = 0; i < needed && i < nc; ++i) selected[cand[i]] = 1;
    free(cand);
  }
  free(vptr);
  free(vobs);
  free(st);
  return TMI_BA_OK;
}

/* ---- BundleAdjustTwoViews (reference bundle_adjust_two_views.cc:113-191), one pair at a time -----
 * The pair is an ordinary problem for the LM above: two cameras (the first with constant
 * extrinsics), two private intrinsics groups whose only free coordinate is the focal length when
 * TwoViewBundleAdjustmentOptions says so (:66-98: SubsetParameterization over indices 1..n-1), one
 * point per correspondence observed by both, DENSE_SCHUR, 200 iterations, Ceres' default
 * tolerances and radius bounds (SetSolverOptions :58-68 sets nothing else), no loss, no inner
 * iterations. */
int32_t oracle_adjust_two_views(tmi_ba_two_view_batch* B, int32_t point_dof, int32_t max_num_iterations,
                                int8_t* termination, int32_t* iterations, double* initial_cost,
                                double* final_cost) {
  if (!B) return TMI_BA_ERR_INVALID_ARGUMENT;
  tmi_ba_options o;
  memset(&o, 0, sizeof(o));
  o.loss_function_type = TMI_BA_LOSS_TRIVIAL;
  o.robust_loss_width = 2.0;
  o.linear_solver_type = TMI_BA_DENSE_SCHUR;
  o.preconditioner_type = TMI_BA_PRECOND_SCHUR_JACOBI;
  o.num_threads = 1;
  o.max_num_iterations = max_num_iterations;
  o.max_solver_time_in_seconds = 1e9;
  o.use_inner_iterations = 0;
  o.function_tolerance = 1e-6;
  o.gradient_tolerance = 1e-10;
  o.parameter_tolerance = 1e-8;
  o.max_trust_region_radius = 1e16;
  o.initial_trust_region_radius = 1e4;
  o.min_trust_region_radius = 1e-32;
  o.min_relative_decrease = 1e-3;
  o.min_lm_diagonal = 1e-6;
  o.max_lm_diagonal = 1e32;
  o.eta = 0.1;
  o.max_linear_solver_iterations = 500;
  o.max_num_consecutive_invalid_steps = 5;
  o.jacobi_scaling = 1;
  o.point_dof = point_dof;
  o.device = -1;
  o.residual_precision = 64;
#pragma omp parallel for schedule(dynamic, 1)
  for (int p = 0; p < B->num_pairs; ++p) {
    const int64_t c0 = B->correspondence_ptr[p], n = B->correspondence_ptr[p + 1] - c0;
    if (n <= 0) {
      if (termination) termination[p] = -1;
      if (iterations) iterations[p] = 0;
      if (initial_cost) initial_cost[p] = 0.0;
      if (final_cost) final_cost[p] = 0.0;
      continue;
    }
    const int n1 = model_size(B->model1[p]), n2 = model_size(B->model2[p]);
    double ext[12], intr[20];
    memcpy(ext, B->extrinsics1 + 6 * (size_t)p, sizeof(double) * 6);
    memcpy(ext + 6, B->extrinsics2 + 6 * (size_t)p, sizeof(double) * 6);
    memcpy(intr, B->intrinsics1 + 10 * (size_t)p, sizeof(double) * (size_t)n1);
    memcpy(intr + n1, B->intrinsics2 + 10 * (size_t)p, sizeof(double) * (size_t)n2);
    int32_t cgrp[2] = {0, 1}, gmodel[2] = {B->model1[p], B->model2[p]}, goff[3] = {0, n1, n1 + n2};
    uint8_t cflag[2] = {(uint8_t)(TMI_BA_CAMERA_POSITION_CONSTANT | TMI_BA_CAMERA_ORIENTATION_CONSTANT), 0};
    uint8_t iconst[20];
    memset(iconst, 1, sizeof(iconst));
    if (B->constant_intrinsics1 && !B->constant_intrinsics1[p]) iconst[0] = 0;
    if (B->constant_intrinsics2 && !B->constant_intrinsics2[p]) iconst[n1] = 0;
    int32_t* ocam = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n);
    int32_t* opt = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n);
    double* oxy = (double*)malloc(sizeof(double) * 4 * (size_t)n);
    for (int64_t q = 0; q < n; ++q) {
      ocam[2 * q] = 0;
      ocam[2 * q + 1] = 1;
      opt[2 * q] = opt[2 * q + 1] = (int32_t)q;
      oxy[4 * q] = B->features1[2 * (c0 + q)];
      oxy[4 * q + 1] = B->features1[2 * (c0 + q) + 1];
      oxy[4 * q + 2] = B->features2[2 * (c0 + q)];
      oxy[4 * q + 3] = B->features2[2 * (c0 + q) + 1];
    }
    tmi_ba_problem Q;
    memset(&Q, 0, sizeof(Q));
    Q.num_cameras = 2; Q.extrinsics = ext; Q.camera_group = cgrp; Q.camera_flags = cflag;
    Q.num_groups = 2; Q.group_model = gmodel; Q.group_offset = goff; Q.intrinsics = intr;
    Q.intrinsics_constant = iconst;
    Q.num_points = (int32_t)n; Q.points = B->points + 4 * (size_t)c0; Q.point_constant = NULL;
    Q.num_observations = 2 * n; Q.obs_camera = ocam; Q.obs_point = opt; Q.obs_xy = oxy;
    tmi_ba_summary sm;
    const int32_t st = oracle_ba_solve(&Q, &o, &sm);
    int8_t t = (int8_t)sm.termination;
    if (st == TMI_BA_ERR_EVALUATION_FAILED) t = 3;
    if (termination) termination[p] = t;
    if (iterations) iterations[p] = sm.num_iterations;
    if (initial_cost) initial_cost[p] = sm.initial_cost;
    if (final_cost) final_cost[p] = sm.final_cost;
    if (sm.success) {  /* the points were updated in place by the solve */
      memcpy(B->extrinsics2 + 6 * (size_t)p, ext + 6, sizeof(double) * 6);
      B->intrinsics1[10 * (size_t)p] = intr[0];
      B->intrinsics2[10 * (size_t)p] = intr[n1];
    }
    free(ocam); free(opt); free(oxy);
  }
  return TMI_BA_OK;
}

/* ==========================================================================================
 * BundleAdjustTwoViewsAngular (bundle_adjust_two_views.cc:193-240): relative rotation (angle
 * axis) and unit-norm relative position of a view pair from AngularEpipolarError residuals.
 * The functor (angular_epipolar_error.h:47-89) and the parameterization
 * (unit_norm_three_vector_parameterization.h:45-63) are evaluated on dual numbers, as Ceres'
 * AutoDiffCostFunction / AutoDiffLocalParameterization do; jet slots 0-2 = rotation, 3-5 =
 * position (or the local increment).  ceres::AngleAxisToRotationMatrix restated from the
 * published ceres/rotation.h (1.x): Rodrigues above theta^2 = DBL_EPSILON, first order below.
 * The trust-region loop is the one of the full solver (Ceres 1.14 TrustRegionMinimizer
 * semantics) on a dense 6-column local Jacobian; with two parameter blocks that
 * meet in every residual DENSE_SCHUR eliminates one 3-block and solves for the other, i.e. it
 * solves the same 6 x 6 damped normal equations.
 * ========================================================================================== */
static void aa_to_rotation_jet(const jet aa[3], jet R[3][3]) {
  const jet one = jet_const(1.0);
  const jet theta2 = jet_add(jet_add(jet_mul(aa[0], aa[0]), jet_mul(aa[1], aa[1])), jet_mul(aa[2], aa[2]));
  if (theta2.a > 2.220446049250313e-16) {
    const jet theta = jet_sqrt(theta2);
    const jet wx = jet_div(aa[0], theta), wy = jet_div(aa[1], theta), wz = jet_div(aa[2], theta);
    const jet c = jet_cos(theta), s = jet_sin(theta), omc = jet_sub(one, c);
    R[0][0] = jet_add(c, jet_mul(jet_mul(wx, wx), omc));
    R[1][0] = jet_add(jet_mul(wz, s), jet_mul(jet_mul(wx, wy), omc));
    R[2][0] = jet_add(jet_neg(jet_mul(wy, s)), jet_mul(jet_mul(wx, wz), omc));
    R[0][1] = jet_sub(jet_mul(jet_mul(wx, wy), omc), jet_mul(wz, s));
    R[1][1] = jet_add(c, jet_mul(jet_mul(wy, wy), omc));
    R[2][1] = jet_add(jet_mul(wx, s), jet_mul(jet_mul(wy, wz), omc));
    R[0][2] = jet_add(jet_mul(wy, s), jet_mul(jet_mul(wx, wz), omc));
    R[1][2] = jet_add(jet_neg(jet_mul(wx, s)), jet_mul(jet_mul(wy, wz), omc));
    R[2][2] = jet_add(c, jet_mul(jet_mul(wz, wz), omc));
  } else {
    R[0][0] = one;          R[0][1] = jet_neg(aa[2]); R[0][2] = aa[1];
    R[1][0] = aa[2];        R[1][1] = one;            R[1][2] = jet_neg(aa[0]);
    R[2][0] = jet_neg(aa[1]); R[2][1] = aa[0];        R[2][2] = one;
  }
}

/* angular_epipolar_error.h:52-84 on jets.  Returns 0 where the functor returns false. */
static int angular_error_jet(const jet rot[3], const jet t[3], const double f1d[2], const double f2d[2], jet* out) {
  const jet f1[3] = {jet_const(f1d[0]), jet_const(f1d[1]), jet_const(1.0)};
  const jet f2[3] = {jet_const(f2d[0]), jet_const(f2d[1]), jet_const(1.0)};
  jet R[3][3];
  aa_to_rotation_jet(rot, R);
  /* translation_term = I - t t^T */
  jet T[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[i][j] = jet_sub(jet_const(i == j ? 1.0 : 0.0), jet_mul(t[i], t[j]));
  jet Rf2[3], Rtf2[3], Tf1[3], TRtf2[3];
  for (int i = 0; i < 3; ++i) {
    Rf2[i] = jet_add(jet_add(jet_mul(R[i][0], f2[0]), jet_mul(R[i][1], f2[1])), jet_mul(R[i][2], f2[2]));
    Rtf2[i] = jet_add(jet_add(jet_mul(R[0][i], f2[0]), jet_mul(R[1][i], f2[1])), jet_mul(R[2][i], f2[2]));
  }
  for (int i = 0; i < 3; ++i) {
    Tf1[i] = jet_add(jet_add(jet_mul(T[i][0], f1[0]), jet_mul(T[i][1], f1[1])), jet_mul(T[i][2], f1[2]));
    TRtf2[i] = jet_add(jet_add(jet_mul(T[i][0], Rtf2[0]), jet_mul(T[i][1], Rtf2[1])), jet_mul(T[i][2], Rtf2[2]));
  }
  jet a = jet_const(0.0);
  for (int i = 0; i < 3; ++i) a = jet_add(a, jet_mul(f1[i], Tf1[i]));
  for (int i = 0; i < 3; ++i) a = jet_add(a, jet_mul(Rf2[i], TRtf2[i]));
  /* b_sqrt = t . (f1 x R^T f2) */
  const jet cx = jet_sub(jet_mul(f1[1], Rtf2[2]), jet_mul(f1[2], Rtf2[1]));
  const jet cy = jet_sub(jet_mul(f1[2], Rtf2[0]), jet_mul(f1[0], Rtf2[2]));
  const jet cz = jet_sub(jet_mul(f1[0], Rtf2[1]), jet_mul(f1[1], Rtf2[0]));
  const jet b = jet_add(jet_add(jet_mul(t[0], cx), jet_mul(t[1], cy)), jet_mul(t[2], cz));
  const jet sq = jet_sub(jet_div(jet_mul(a, a), jet_const(4.0)), jet_mul(b, b));
  if (sq.a < 0.0) return 0;
  *out = jet_sub(jet_div(a, jet_const(2.0)), jet_sqrt(sq));
  return 1;
}

int32_t oracle_angular_epipolar_error(const double* rotation, const double* position, const double* f1,
                                      const double* f2, double* residual) {
  jet r[3], t[3], e;
  for (int i = 0; i < 3; ++i) {
    r[i] = jet_const(rotation[i]);
    t[i] = jet_const(position[i]);
  }
  if (!angular_error_jet(r, t, f1, f2, &e)) return 0;
  *residual = e.a;
  return 1;
}

/* UnitNormThreeVectorParameterization::operator() on plain doubles */
static void unit_norm_plus(const double x[3], const double d[3], double out[3]) {
  for (int i = 0; i < 3; ++i) out[i] = x[i] + d[i];
  const double sq = out[0] * out[0] + out[1] * out[1] + out[2] * out[2];
  if (sq > 0.0) {
    const double nrm = sqrt(sq);
    for (int i = 0; i < 3; ++i) out[i] /= nrm;
  }
}

/* residuals (and, if H, the scaled local normal equations H = Js^T Js packed upper, g = Js^T r) at
 * x = [rotation | position]; sc = column scales of the local Jacobian.  Returns 0 on evaluation failure. */
static int angular_linearize(const double x[6], const double* f1, const double* f2, int64_t n, const double sc[6],
                             double* H, double* g, double* cost) {
  /* local parameterization Jacobian of the position block: d Plus(x, delta) / d delta at 0, by jets
   * (AutoDiffLocalParameterization) */
  double P[3][3];
  {
    jet xp[3];
    for (int i = 0; i < 3; ++i) xp[i] = jet_add(jet_const(x[3 + i]), jet_var(0.0, i));
    const jet sq = jet_add(jet_add(jet_mul(xp[0], xp[0]), jet_mul(xp[1], xp[1])), jet_mul(xp[2], xp[2]));
    if (sq.a > 0.0) {
      const jet nrm = jet_sqrt(sq);
      for (int i = 0; i < 3; ++i) xp[i] = jet_div(xp[i], nrm);
    }
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) P[i][k] = xp[i].v[k];
  }
  jet rot[3], t[3];
  for (int i = 0; i < 3; ++i) {
    rot[i] = jet_var(x[i], i);
    t[i] = jet_var(x[3 + i], 3 + i);
  }
  if (H) {
    for (int i = 0; i < 21; ++i) H[i] = 0.0;
    for (int i = 0; i < 6; ++i) g[i] = 0.0;
  }
  double c = 0.0;
  for (int64_t q = 0; q < n; ++q) {
    jet e;
    if (!angular_error_jet(rot, t, f1 + 2 * q, f2 + 2 * q, &e)) return 0;
    c += 0.5 * e.a * e.a;
    if (!H) continue;
    double J[6];
    for (int k = 0; k < 3; ++k) J[k] = e.v[k] * sc[k];
    for (int k = 0; k < 3; ++k)
      J[3 + k] = (e.v[3] * P[0][k] + e.v[4] * P[1][k] + e.v[5] * P[2][k]) * sc[3 + k];
    int idx = 0;
    for (int a = 0; a < 6; ++a) {
      for (int b = a; b < 6; ++b) H[idx++] += J[a] * J[b];
      g[a] += J[a] * e.a;
    }
  }
  *cost = c;
  return 1;
}

static int chol6_solve(const double* H, const double* diag_add, const double* g, double* y) {
  double A[6][6], L[6][6];
  int idx = 0;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) {
      A[a][b] = A[b][a] = H[idx++];
    }
  for (int a = 0; a < 6; ++a) A[a][a] += diag_add[a];
  for (int j = 0; j < 6; ++j) {
    double d = A[j][j];
    for (int m = 0; m < j; ++m) d -= L[j][m] * L[j][m];
    if (!(d > 0.0)) return 0;
    L[j][j] = sqrt(d);
    for (int i = j + 1; i < 6; ++i) {
      double t = A[i][j];
      for (int m = 0; m < j; ++m) t -= L[i][m] * L[j][m];
      L[i][j] = t / L[j][j];
    }
  }
  double z[6];
  for (int i = 0; i < 6; ++i) {
    double t = g[i];
    for (int m = 0; m < i; ++m) t -= L[i][m] * z[m];
    z[i] = t / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double t = z[i];
    for (int m = i + 1; m < 6; ++m) t -= L[m][i] * y[m];
    y[i] = t / L[i][i];
  }
  return 1;
}

int32_t oracle_adjust_two_views_angular(tmi_ba_two_view_angular_batch* B, int32_t max_num_iterations,
                                        int8_t* termination, int32_t* iterations, double* initial_cost,
                                        double* final_cost) {
  if (!B || B->num_pairs < 0) return TMI_BA_ERR_INVALID_ARGUMENT;
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double max_radius = 1e16, min_radius = 1e-32, min_relative_decrease = 1e-3, lm_lo = 1e-6, lm_hi = 1e32;
#pragma omp parallel for schedule(dynamic, 4)
  for (int p = 0; p < B->num_pairs; ++p) {
    const int64_t c0 = B->correspondence_ptr[p], n = B->correspondence_ptr[p + 1] - c0;
    int8_t term = 1;
    int iter = 0;
    double cost = 0.0, cost0 = 0.0;
    if (n <= 0) {
      term = -1;
    } else {
      const double *f1 = B->features1 + 2 * c0, *f2 = B->features2 + 2 * c0;
      double x[6];
      for (int i = 0; i < 3; ++i) {
        x[i] = B->rotation2[3 * (size_t)p + i];
        x[3 + i] = B->position2[3 * (size_t)p + i];
      }
      double sc[6] = {1, 1, 1, 1, 1, 1}, H[21], g[6];
      if (!angular_linearize(x, f1, f2, n, sc, H, g, &cost)) {
        term = 3;
        cost0 = cost;
      } else {
        cost0 = cost;
        double gmax = 0.0;
        for (int a = 0; a < 6; ++a) gmax = fmax(gmax, fabs(g[a]));
        {
          /* Jacobi scaling from the start point: 1 / (1 + ||column||) */
          int idx = 0;
          for (int a = 0; a < 6; ++a) {
            sc[a] = 1.0 / (1.0 + sqrt(H[idx]));
            idx += 6 - a;
          }
          angular_linearize(x, f1, f2, n, sc, H, g, &cost);
        }
        double x_norm = 0.0;
        for (int a = 0; a < 6; ++a) x_norm += x[a] * x[a];
        x_norm = sqrt(x_norm);
        double radius = 1e4, decrease_factor = 2.0;
        int invalid_run = 0;
        if (gmax <= gradient_tolerance) {
          term = 0;
        } else {
          for (;;) {
            if (iter >= max_num_iterations) break;
            ++iter;
            double dadd[6], y[6];
            {
              int idx = 0;
              for (int a = 0; a < 6; ++a) {
                dadd[a] = fmin(fmax(H[idx], lm_lo), lm_hi) / radius;
                idx += 6 - a;
              }
            }
            int step_ok = chol6_solve(H, dadd, g, y);
            double mcc = 0.0;
            if (step_ok) {
              double yg = 0.0, yHy = 0.0, Hf[6][6];
              int idx = 0;
              for (int a = 0; a < 6; ++a)
                for (int b = a; b < 6; ++b) Hf[a][b] = Hf[b][a] = H[idx++];
              for (int a = 0; a < 6; ++a) {
                yg += y[a] * g[a];
                double t = 0.0;
                for (int b = 0; b < 6; ++b) t += Hf[a][b] * y[b];
                yHy += y[a] * t;
              }
              mcc = yg - 0.5 * yHy;
              if (!(mcc > 0.0)) step_ok = 0;
            }
            if (!step_ok) {
              if (++invalid_run >= 5) {
                term = 2;
                break;
              }
              radius /= decrease_factor;
              decrease_factor *= 2.0;
              if (radius < min_radius) {
                term = 0;
                break;
              }
              continue;
            }
            invalid_run = 0;
            double xc[6], d[6];
            for (int a = 0; a < 6; ++a) d[a] = -y[a] * sc[a];
            for (int a = 0; a < 3; ++a) xc[a] = x[a] + d[a];
            unit_norm_plus(x + 3, d + 3, xc + 3);
            double step_sq = 0.0;
            for (int a = 0; a < 6; ++a) step_sq += (xc[a] - x[a]) * (xc[a] - x[a]);
            double cand;
            if (!angular_linearize(xc, f1, f2, n, sc, NULL, NULL, &cand)) cand = 1.7976931348623157e308;
            if (sqrt(step_sq) <= parameter_tolerance * (x_norm + parameter_tolerance)) {
              term = 0;
              break;
            }
            const double cost_change = cost - cand;
            if (fabs(cost_change) <= function_tolerance * cost) {
              term = 0;
              break;
            }
            const double rd = cost_change / mcc;
            if (rd > min_relative_decrease) {
              for (int a = 0; a < 6; ++a) x[a] = xc[a];
              x_norm = 0.0;
              for (int a = 0; a < 6; ++a) x_norm += x[a] * x[a];
              x_norm = sqrt(x_norm);
              if (!angular_linearize(x, f1, f2, n, sc, H, g, &cost)) {
                term = 2;
                break;
              }
              gmax = 0.0;
              for (int a = 0; a < 6; ++a) gmax = fmax(gmax, fabs(g[a] / sc[a]));
              radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rd - 1.0, 3.0));
              radius = fmin(max_radius, radius);
              decrease_factor = 2.0;
              if (gmax <= gradient_tolerance) {
                term = 0;
                break;
              }
            } else {
              radius /= decrease_factor;
              decrease_factor *= 2.0;
            }
            if (radius < min_radius) {
              term = 0;
              break;
            }
          }
        }
        if (term != 2) {
          for (int i = 0; i < 3; ++i) {
            B->rotation2[3 * (size_t)p + i] = x[i];
            B->position2[3 * (size_t)p + i] = x[3 + i];
          }
        }
      }
    }
    if (termination) termination[p] = term;
    if (iterations) iterations[p] = iter;
    if (initial_cost) initial_cost[p] = cost0;
    if (final_cost) final_cost[p] = cost;
  }
  return TMI_BA_OK;
}
