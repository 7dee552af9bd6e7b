// Minimal stand-in for the few ceres/types.h enumerations that appear in
// TheiaSfM's BundleAdjustmentOptions (reference:
// src/theia/sfm/bundle_adjustment/bundle_adjustment.h:86-89).  Enumerator names
// and order follow Ceres Solver 1.x so that code written against the reference
// (`options.linear_solver_type = ceres::ITERATIVE_SCHUR;`) compiles unchanged.
// No Ceres functionality lives here: the numerics are the HIP engine behind
// include/theia_mi355_ba.h.
#ifndef THEIA_MI355_CERES_TYPES_SHIM_H_
#define THEIA_MI355_CERES_TYPES_SHIM_H_
namespace ceres {
enum LinearSolverType {
  DENSE_NORMAL_CHOLESKY,
  DENSE_QR,
  SPARSE_NORMAL_CHOLESKY,
  DENSE_SCHUR,
  SPARSE_SCHUR,
  ITERATIVE_SCHUR,
  CGNR
};
enum PreconditionerType { IDENTITY, JACOBI, SCHUR_JACOBI, CLUSTER_JACOBI, CLUSTER_TRIDIAGONAL };
enum VisibilityClusteringType { CANONICAL_VIEWS, SINGLE_LINKAGE };
}  // namespace ceres
#endif
