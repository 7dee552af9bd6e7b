/*
 * theia_mi355_ba.h -- C ABI of the MI355X-native bundle-adjustment engine.
 *
 * This is the drop-in boundary for TheiaSfM's full-reconstruction bundle
 * adjustment.  Theia has no FFI / plugin registry for this path: the seam is
 * the C++ API
 *     BundleAdjustReconstruction / BundleAdjustPartialReconstruction
 *         (reference: src/theia/sfm/bundle_adjustment/bundle_adjustment.h:136-155)
 *     class BundleAdjuster { AddView, AddTrack, Optimize }
 *         (reference: src/theia/sfm/bundle_adjustment/bundle_adjuster.h:60-77)
 * whose Optimize() is hard-wired to ceres::Solve
 *         (reference: src/theia/sfm/bundle_adjustment/bundle_adjuster.cc:205).
 * The entry points below are what a maintainer would bind in place of that
 * ceres::Solve call: the host side flattens the Reconstruction into the SoA
 * arrays of tmi_ba_problem, calls tmi_ba_solve(), and copies the in/out arrays
 * back (see INTEGRATION.md for the binding, include/theia/ for the host shim).
 *
 * Plain C, plain pointers and sizes.  No torch / Eigen / Ceres types.
 * All floating point is IEEE fp64, all indices int32 (observation count int64).
 * The library never keeps a caller pointer past the return of a call, never
 * aborts and never throws across this boundary: every entry point returns a
 * tmi_ba_status.
 */
#ifndef THEIA_MI355_BA_H_
#define THEIA_MI355_BA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMI_BA_VERSION_MAJOR 0
#define TMI_BA_VERSION_MINOR 1

/* ---- status codes ------------------------------------------------------ */
typedef enum tmi_ba_status {
  TMI_BA_OK = 0,
  TMI_BA_ERR_INVALID_ARGUMENT = 1, /* null pointer, bad index, bad enum      */
  TMI_BA_ERR_NO_DEVICE = 2,        /* no gfx950 device visible               */
  TMI_BA_ERR_DEVICE = 3,           /* a HIP runtime call failed              */
  TMI_BA_ERR_OUT_OF_MEMORY = 4,
  TMI_BA_ERR_UNSUPPORTED = 5,      /* problem shape the device path lacks    */
  TMI_BA_ERR_EVALUATION_FAILED = 6,/* residual undefined at the start point  */
  TMI_BA_ERR_LINEAR_SOLVER = 7,    /* reduced system could not be solved     */
  TMI_BA_ERR_COLLECTIVE = 8        /* the all-reduce callback reported error */
} tmi_ba_status;

/* ---- enums mirrored from the reference ---------------------------------- */

/* CameraIntrinsicsModelType,
 * reference: src/theia/sfm/camera/camera_intrinsics_model_type.h:45-52.
 * Parameter order inside one intrinsics group:
 *   PINHOLE (7)                  [f, ar, skew, px, py, k1, k2]
 *       reference: pinhole_camera_model.h:86-94
 *   PINHOLE_RADIAL_TANGENTIAL(10)[f, ar, skew, px, py, k1, k2, k3, t1, t2]
 *       reference: pinhole_radial_tangential_camera_model.h:91-102
 *   FISHEYE (9)                  [f, ar, skew, px, py, k1, k2, k3, k4]
 *       reference: fisheye_camera_model.h:67-77
 *   FOV (5)                      [f, ar, px, py, omega]
 *       reference: fov_camera_model.h:69-75
 *   DIVISION_UNDISTORTION (5)    [f, ar, px, py, k]
 *       reference: division_undistortion_camera_model.h:76-82            */
typedef enum tmi_ba_camera_model {
  TMI_BA_PINHOLE = 0,
  TMI_BA_PINHOLE_RADIAL_TANGENTIAL = 1,
  TMI_BA_FISHEYE = 2,
  TMI_BA_FOV = 3,
  TMI_BA_DIVISION_UNDISTORTION = 4
} tmi_ba_camera_model;
#define TMI_BA_MAX_INTRINSICS 10
#define TMI_BA_EXTRINSICS_SIZE 6 /* [C(3), angle_axis(3)], camera.h:195-200 */

/* LossFunctionType,
 * reference: src/theia/sfm/bundle_adjustment/create_loss_function.h:51-58 */
typedef enum tmi_ba_loss {
  TMI_BA_LOSS_TRIVIAL = 0,
  TMI_BA_LOSS_HUBER = 1,
  TMI_BA_LOSS_SOFTLONE = 2,
  TMI_BA_LOSS_CAUCHY = 3,
  TMI_BA_LOSS_ARCTAN = 4,
  TMI_BA_LOSS_TUKEY = 5
} tmi_ba_loss;

/* ceres::LinearSolverType values Theia passes through
 * (reference: bundle_adjustment.h:86; bundle_adjustment.cc:86,100 force
 * DENSE_QR for single view / single track problems).  The device path solves
 * the reduced camera system exactly (blocked dense Cholesky of the explicit
 * Schur complement) for DENSE_QR/DENSE_SCHUR/SPARSE_SCHUR and with
 * preconditioned conjugate gradients for ITERATIVE_SCHUR/CGNR.             */
typedef enum tmi_ba_linear_solver {
  TMI_BA_DENSE_QR = 1,
  TMI_BA_DENSE_SCHUR = 3,
  TMI_BA_SPARSE_SCHUR = 4,
  TMI_BA_ITERATIVE_SCHUR = 5,
  TMI_BA_CGNR = 6
} tmi_ba_linear_solver;

/* ceres::PreconditionerType (bundle_adjustment.h:87).  Substitutions made by the device
 * path, all of them in the PRECONDITIONER only (the reduced system that PCG solves, its
 * stopping rule and therefore the LM step it converges to are the same in every mode; what
 * changes is the number of PCG iterations an LM step takes and the low-order bits of the
 * truncated solve):
 *   SCHUR_JACOBI (Theia's default): the inverse of the block diagonal of the reduced camera
 *     matrix S with ONE block per view = [extrinsics | private intrinsics] merged (up to
 *     16 x 16).  Ceres builds its block diagonal per PARAMETER block, i.e. a 6 x 6
 *     extrinsics block and a separate N x N intrinsics block per view
 *     (schur_jacobi_preconditioner.cc); the merged block keeps the extrinsics-intrinsics
 *     coupling of a view, is a strictly stronger preconditioner and costs the same to apply.
 *   SCHUR_JACOBI_PARAMETER_BLOCKS (extension value, not a ceres enumerator): exactly Ceres's
 *     shape -- the cross terms between a view's extrinsics and intrinsics columns are dropped
 *     before the inversion.  Meant for block-for-block comparisons of PCG trajectories with
 *     real Ceres output (tests/test_ceres_golden.py); slower to converge than the default.
 *   JACOBI maps to SCHUR_JACOBI.
 *   CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL: Ceres clusters the cameras by visibility
 *     (visibility_clustering_type) and inverts the block diagonal of S over the clusters exactly.  Here the
 *     clusters are the intrinsics groups: a cluster = a shared intrinsics block together with the views that share
 *     it, its principal submatrix of S factored densely every LM iteration (6 n_views + <= 10 unknowns) and
 *     applied with two triangular solves per PCG iteration; views of private groups keep their SCHUR_JACOBI block.
 *     Sharing intrinsics puts about three near-degenerate directions per shared block into the block-Jacobi
 *     preconditioned system (principal point against a coherent rotation of the block's views, ...): 45-80 PCG
 *     iterations per LM iteration with SCHUR_JACOBI, 4-5 with the clusters.  Needs the cluster's blocks of S only:
 *     with schur_mode auto (and implicit) the operator PCG applies is the matrix-free one and just the block pairs
 *     INSIDE a cluster are formed for the preconditioner; schur_mode explicit forms all of S.
 *     On problems WITHOUT shared intrinsics blocks the views are clustered by visibility as Ceres'
 *     VisibilityBasedPreconditioner does (visibility_clustering_type below: the Schur-complement graph with
 *     edge weights |tracks seen by both| / sqrt(|tracks of a| |tracks of b|) over the parameter blocks, then
 *     canonical views with size penalty 3 / similarity penalty 0 / at least 3 centres, or single linkage at 0.9;
 *     restated from Ceres 1.14 -- parity unpinned like the rest of the Ceres layer) and every cluster's principal
 *     submatrix of S is inverted exactly; this needs the formed S (schur_mode auto picks it; with schur_mode
 *     implicit, or on several ranks, the solve keeps the SCHUR_JACOBI blocks and says so in
 *     tmi_ba_summary.effective_preconditioner_type).
 *     CLUSTER_TRIDIAGONAL (round 6, cluster_chains.h): Ceres' tridiagonal variant also keeps the blocks between clusters that are neighbours in a degree-2 maximum spanning forest of the cluster graph (vertices: the
 *     clusters above plus every other reduced block as a cluster of its own; edge weight: the number of non-constant
 *     tracks both clusters see; edges taken in decreasing order of (weight, lower end, higher end) unless an end has two
 *     edges already or the ends are connected).  The forest's components are paths; every path is factored exactly as
 *     a block-tridiagonal matrix (a dense factor whose tiles outside the band stay zero: a path is cut where it would
 *     pass TMI_BA_MAX_CLUSTER_DIM unknowns).  Dropping the other blocks can cost positive definiteness: as in Ceres the
 *     off-diagonal cluster-pair cells are then halved and the factorisation repeated once; a second failure fails the
 *     linear solve (an invalid LM step).  Needs the formed S and one rank -- schur_mode auto forms S, schur_mode implicit
 *     and sharded solves keep the SCHUR_JACOBI blocks (effective_preconditioner_type says so).  A handle serves the
 *     preconditioner it was created for (the chains are part of its structure).  Restated from Ceres 1.14
 *     (visibility_based_preconditioner.cc, graph_algorithms.h); parity unpinned like the rest of the Ceres layer.
 *     A cluster launch that cannot become co-resident (device shared with another process) retires
 *     the clusters for that solve: PCG continues with the SCHUR_JACOBI blocks.
 *   Intrinsics shared by several views form their own reduced block in every mode.  */
/* A cluster of CLUSTER_JACOBI is inverted as a dense matrix (n^3 / 3 flops and n^2 / 2 doubles per LM iteration): a
 * cluster with more unknowns than this keeps the SCHUR_JACOBI blocks of its views (Ceres factors its cluster matrices
 * sparsely and has no such limit). */
#define TMI_BA_MAX_CLUSTER_DIM 4096

typedef enum tmi_ba_preconditioner {
  TMI_BA_PRECOND_IDENTITY = 0,
  TMI_BA_PRECOND_JACOBI = 1,
  TMI_BA_PRECOND_SCHUR_JACOBI = 2,
  TMI_BA_PRECOND_CLUSTER_JACOBI = 3,
  TMI_BA_PRECOND_CLUSTER_TRIDIAGONAL = 4,
  TMI_BA_PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS = 18
} tmi_ba_preconditioner;

/* OptimizeIntrinsicsType bit flags, reference: bundle_adjustment.h:65-76 */
enum {
  TMI_BA_INTRINSICS_NONE = 0x00,
  TMI_BA_INTRINSICS_FOCAL_LENGTH = 0x01,
  TMI_BA_INTRINSICS_ASPECT_RATIO = 0x02,
  TMI_BA_INTRINSICS_SKEW = 0x04,
  TMI_BA_INTRINSICS_PRINCIPAL_POINTS = 0x08,
  TMI_BA_INTRINSICS_RADIAL_DISTORTION = 0x10,
  TMI_BA_INTRINSICS_TANGENTIAL_DISTORTION = 0x20,
  TMI_BA_INTRINSICS_ALL = 0x3f
};

/* camera_flags bits: which half of the 6 extrinsics is held constant
 * (reference: bundle_adjuster.cc:304-334; both bits = SetCameraExtrinsicsConstant,
 * which is also what AddTrack does to cameras that were not added through
 * AddView, bundle_adjuster.cc:156-168).                                     */
enum {
  TMI_BA_CAMERA_POSITION_CONSTANT = 0x1,
  TMI_BA_CAMERA_ORIENTATION_CONSTANT = 0x2
};

/* ---- the flattened problem ---------------------------------------------- */
/* Caller-owned structure-of-arrays view of the residual set
 *   {(view, track): view estimated, track estimated, view observes track}
 * that BundleAdjuster::AddView/AddTrack build (bundle_adjuster.cc:102-180).
 * Arrays marked in/out are overwritten with the optimised values when the
 * solve produced a usable solution (Ceres semantics: CONVERGENCE or
 * NO_CONVERGENCE, bundle_adjuster.cc:218) and left untouched otherwise.     */
typedef struct tmi_ba_problem {
  /* cameras (one per optimised or anchoring view) */
  int32_t num_cameras;
  double* extrinsics;            /* in/out [6*num_cameras]                   */
  const int32_t* camera_group;   /* [num_cameras] intrinsics group index     */
  const uint8_t* camera_flags;   /* [num_cameras] TMI_BA_CAMERA_* bits       */

  /* shared intrinsics groups (reference: reconstruction.h:93-106; one Ceres
   * parameter block per group because Camera holds a shared_ptr, camera.h:247) */
  int32_t num_groups;
  const int32_t* group_model;    /* [num_groups] tmi_ba_camera_model         */
  const int32_t* group_offset;   /* [num_groups+1] offsets into intrinsics   */
  double* intrinsics;            /* in/out [group_offset[num_groups]]        */
  const uint8_t* intrinsics_constant; /* same length; 1 = held constant
                                    (GetSubsetFromOptimizeIntrinsicsType and
                                    the whole-block rule bundle_adjuster.cc:242-287) */

  /* tracks: homogeneous 3-D points (reference: track.h:66-67) */
  int32_t num_points;
  double* points;                /* in/out [4*num_points]                    */
  const uint8_t* point_constant; /* [num_points] 1 = SetTrackConstant        */

  /* observations: Feature = pixel (x, y) (reference: feature.h) */
  int64_t num_observations;
  const int32_t* obs_camera;     /* [num_observations]                       */
  const int32_t* obs_point;      /* [num_observations]                       */
  const double* obs_xy;          /* [2*num_observations]                     */
} tmi_ba_problem;

/* ---- options: BundleAdjustmentOptions field for field -------------------- */
/* reference: src/theia/sfm/bundle_adjustment/bundle_adjustment.h:78-122.
 * intrinsics_to_optimize / constant_camera_* live in the flattened problem
 * (intrinsics_constant, camera_flags); tmi_ba_intrinsics_constant_mask()
 * below converts the bitmask.                                               */
typedef struct tmi_ba_options {
  int32_t loss_function_type;        /* tmi_ba_loss, default TRIVIAL         */
  double robust_loss_width;          /* 2.0                                   */
  int32_t linear_solver_type;        /* tmi_ba_linear_solver, SPARSE_SCHUR    */
  int32_t preconditioner_type;       /* tmi_ba_preconditioner, SCHUR_JACOBI   */
  int32_t verbose;                   /* 0                                     */
  int32_t num_threads;               /* accepted, unused by the device path   */
  int32_t max_num_iterations;        /* 100                                   */
  double max_solver_time_in_seconds; /* 3600                                  */
  int32_t use_inner_iterations;      /* 1 (reference default, bundle_adjustment.h:112): after
                                        every trust-region step one coordinate-descent sweep
                                        over extrinsics blocks, intrinsics blocks, points
                                        (Ceres inner iterations in the reversed solver
                                        ordering, bundle_adjuster.cc:193-200), until their relative gain
                                        drops below 1e-3; evaluated in fp64 whatever
                                        residual_precision says                  */
  double function_tolerance;         /* 1e-6                                  */
  double gradient_tolerance;         /* 1e-10                                 */
  double parameter_tolerance;        /* 1e-8                                  */
  double max_trust_region_radius;    /* 1e12                                  */

  /* Ceres defaults Theia does not override; exposed so the
   * benchmark and the tests can pin them.                                    */
  double initial_trust_region_radius; /* 1e4                                  */
  double min_trust_region_radius;     /* 1e-32                                */
  double min_relative_decrease;       /* 1e-3                                 */
  double min_lm_diagonal;             /* 1e-6                                 */
  double max_lm_diagonal;             /* 1e32                                 */
  double eta;                         /* 0.1  (PCG forcing, q-tolerance)      */
  int32_t max_linear_solver_iterations; /* 500                                */
  int32_t min_linear_solver_iterations; /* 0                                  */
  int32_t max_num_consecutive_invalid_steps; /* 5                             */
  int32_t jacobi_scaling;             /* 1                                    */

  /* extensions of the device path */
  int32_t point_dof;       /* 4 = reference-exact homogeneous points (no
                              parameterization, bundle_adjuster.cc:379-385);
                              3 = hold w fixed (the north-star 2x3 blocks)   */
  int32_t device;          /* HIP device ordinal; -1 = current device        */
  int32_t profile_kernels; /* 1 = bracket every kernel class with HIP events
                              and report per-class times in the summary      */
  int32_t residual_precision; /* 64 = fp64 throughout (reference precision).
                              32 = residuals and Jacobian blocks evaluated in fp32
                              (camera translation still removed in fp64), loss
                              correction and all accumulation (J^T J, gradients,
                              cost) in fp64 -- BASELINE config 5.  Fixed at
                              tmi_ba_solver_create.                                */
  int32_t schur_mode;      /* ITERATIVE_SCHUR only.  1 = explicit: form the block
                              sparse reduced camera matrix S (Schur complement) and
                              run PCG on it; with several GPUs S is all-reduced once
                              per LM iteration.  2 = implicit: S is never formed, every
                              PCG product walks the observations (Ceres'
                              ImplicitSchurComplement), with several GPUs one small
                              all-reduce (the reduced vector) per PCG iteration.
                              0 = auto: implicit on several GPUs; on one GPU both
                              operators are kept resident and every LM iteration takes
                              the cheaper one for the PCG length it expects (forming S
                              pays off after a few products; same result to round-off). */
  int32_t visibility_clustering_type;
                           /* CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL on a problem WITHOUT shared
                              intrinsics blocks: how the views are clustered (ceres::
                              VisibilityClusteringType, bundle_adjustment.h:88-89; Theia's default is
                              CANONICAL_VIEWS).  0 = CANONICAL_VIEWS, 1 = SINGLE_LINKAGE.          */
  double* iteration_trace; /* optional (NULL = none): HOST buffer of iteration_trace_capacity rows of
                              TMI_BA_TRACE_STRIDE doubles, one row per trust-region iteration (the first
                              summary.num_iterations rows, fewer if the buffer is shorter):
                              [0] iteration (1-based), [1] cost at the iteration's linearisation point,
                              [2] trust-region radius the step was computed with, [3] outcome (1 accepted,
                              0 rejected, -1 invalid step, 2 parameter tolerance reached, 3 function
                              tolerance reached -- 2 and 3 drop the candidate), [4] candidate cost (NaN for an
                              invalid step), [5] model cost change, [6] linear-solver (PCG) iterations of this
                              iteration, [7] step norm.  What ceres::Solver::Summary::iterations holds; the
                              reference reads none of it (bundle_adjuster.cc:203-218) -- it exists so that
                              tests can hold the trust-region trajectory against an independent model
                              (tests/trajectory_model.py).  On a sharded solve every rank writes the same rows. */
  int32_t iteration_trace_capacity;
} tmi_ba_options;
#define TMI_BA_TRACE_STRIDE 8

/* ---- summary: BundleAdjustmentSummary + device-path extras --------------- */
/* reference: bundle_adjustment.h:125-133 */
#define TMI_BA_NUM_KERNEL_CLASSES 12
typedef struct tmi_ba_summary {
  int32_t success;               /* IsSolutionUsable()                       */
  double initial_cost;           /* 1/2 sum rho(|r|^2)                       */
  double final_cost;
  double setup_time_in_seconds;  /* flatten-side preprocessing + upload      */
  double solve_time_in_seconds;  /* the LM loop, device-resident inputs      */

  int32_t status;                /* tmi_ba_status                            */
  int32_t termination;           /* 0 convergence, 1 no convergence (limits),
                                    2 failure                                */
  int32_t num_iterations;        /* LM iterations run (accepted + rejected)  */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int64_t num_linear_solver_iterations; /* PCG iterations over the solve     */
  double final_rmse;             /* sqrt(sum |r|^2 / num_obs), un-robustified */
  double initial_rmse;
  int32_t num_reduced_blocks;    /* camera blocks of the reduced system      */
  int32_t reduced_block_dim;     /* D (uniform, zero padded)                 */
  int64_t num_schur_blocks;      /* structurally non-zero D x D blocks of S,
                                    upper triangle incl. diagonal            */
  int64_t num_schur_pairs;       /* observation pairs feeding off-diagonal S */
  int32_t num_inner_iteration_steps; /* LM iterations that ran an inner-iteration sweep */
  int32_t num_matrix_free_iterations; /* LM iterations whose PCG ran on the matrix-free operator
                                         (all of them with schur_mode 2; some with auto on one rank) */
  /* per kernel class (index = tmi_ba_kernel_class): launches and total
   * device time from HIP events; filled when options.profile_kernels != 0   */
  int64_t kernel_launches[TMI_BA_NUM_KERNEL_CLASSES];
  double kernel_seconds[TMI_BA_NUM_KERNEL_CLASSES];
  char message[192];
  /* the preconditioner PCG actually ran with in the last LM iteration (tmi_ba_preconditioner_type; 0 for the exact
   * solvers): differs from options.preconditioner_type where the header above says a request is served by another
   * one -- CLUSTER_JACOBI without usable clusters (several ranks, schur_mode implicit without shared blocks, a
   * cluster launch that could not become co-resident, clusters that do not fit) keeps its SCHUR_JACOBI blocks,
   * JACOBI runs as SCHUR_JACOBI.  A caller can tell which trajectory it got. */
  int32_t effective_preconditioner_type;
} tmi_ba_summary;

typedef enum tmi_ba_kernel_class {
  TMI_BA_K_LINEARIZE = 0,      /* residuals + Jacobian blocks per observation */
  TMI_BA_K_POINT_ELIMINATE = 1,/* per-track V, g_p, (V+D)^-1, Y = W L^-T     */
  TMI_BA_K_CAMERA_DIAG = 2,    /* per-camera U, g~, diagonal S block          */
  TMI_BA_K_SCHUR_OFFDIAG = 3,  /* off-diagonal S blocks from pair lists       */
  TMI_BA_K_PRECONDITIONER = 4, /* invert diagonal blocks                      */
  TMI_BA_K_SPMV = 5,           /* q = S p (PCG)                               */
  TMI_BA_K_PCG_VECTOR = 6,     /* PCG dots / axpys / preconditioner apply     */
  TMI_BA_K_CHOLESKY = 7,       /* dense reduced-system factor + solve         */
  TMI_BA_K_BACK_SUBSTITUTE = 8,/* delta_p, model cost change                  */
  TMI_BA_K_UPDATE_COST = 9,    /* x+ = x + delta, trial cost                  */
  TMI_BA_K_REDUCE = 10,        /* small reductions / bookkeeping              */
  TMI_BA_K_ALLREDUCE = 11      /* time spent inside the all-reduce callback   */
} tmi_ba_kernel_class;

/* ---- multi-GPU hook ------------------------------------------------------- */
/* When the tracks are sharded over several processes (one per GPU) each rank
 * builds the reduced camera system from its own tracks; the engine then calls
 * this hook once per LM iteration on the concatenated device buffer
 *   [S blocks | U diagonal | reduced gradient | cost terms]
 * and once (a few doubles) for the trial cost.  The hook must sum `count`
 * doubles in place across ranks (RCCL all-reduce over xGMI in practice) on
 * `stream` or synchronise it itself.  Return 0 on success.                  */
typedef int (*tmi_ba_allreduce_fn)(void* device_buffer, int64_t count,
                                   void* hip_stream, void* user);

/* ---- entry points --------------------------------------------------------- */
typedef struct tmi_ba_solver tmi_ba_solver; /* opaque, device-resident problem */

/* Library / device discovery. */
int32_t tmi_ba_version(void);           /* major*1000 + minor                 */
int32_t tmi_ba_device_count(void);      /* gfx950 devices visible, <0 = error */
const char* tmi_ba_status_string(int32_t status);
/* Human-readable detail of the last failed call on the calling thread. */
const char* tmi_ba_last_error(void);

/* Fill `opts` with the reference defaults (bundle_adjustment.h:78-122 plus
 * the Ceres defaults of SURVEY App. B). */
void tmi_ba_options_init(tmi_ba_options* opts);

/* Number of intrinsic parameters of a camera model (kIntrinsicsSize), or -1. */
int32_t tmi_ba_intrinsics_size(int32_t camera_model);

/* GetSubsetFromOptimizeIntrinsicsType for one model
 * (reference: pinhole_camera_model.cc:132-162 and the four siblings):
 * writes 1 into mask[i] for every parameter held constant under the
 * OptimizeIntrinsicsType bitmask.  mask has tmi_ba_intrinsics_size() entries. */
int32_t tmi_ba_intrinsics_constant_mask(int32_t camera_model,
                                        int32_t intrinsics_to_optimize,
                                        uint8_t* mask);

/* One-shot: upload, solve, download.  The replacement for ceres::Solve at
 * bundle_adjuster.cc:205.  Re-entrant; serialises per device.               */
int32_t tmi_ba_solve(tmi_ba_problem* problem, const tmi_ba_options* options,
                     tmi_ba_summary* summary);

/* Resident form (used by the benchmark so the timed region starts with the
 * inputs already in HBM, and by pipelines that run BA -> filter -> BA):
 *   create   : validate, build the static structure (track-major and
 *              camera-major orders, Schur block structure, pair lists),
 *              upload.  `rank`/`world` select the contiguous shard of tracks
 *              this process owns (0/1 for single GPU).
 *   solve    : run LM on the resident parameters (may be called repeatedly).
 *   reset    : restore the parameters uploaded at create time (or by the last set_parameters).
 *   set_parameters : new values of extrinsics / intrinsics / points for the SAME residual set and
 *              constancy flags (the caller moved cameras or re-triangulated tracks between two
 *              BAs): uploads them and makes them what `reset` restores; the structure, the
 *              layouts and every buffer stay resident.
 *   download : copy the current parameters into the caller's in/out arrays.  */
int32_t tmi_ba_solver_create(const tmi_ba_problem* problem,
                             const tmi_ba_options* options, int32_t rank,
                             int32_t world, tmi_ba_solver** out);
int32_t tmi_ba_solver_set_allreduce(tmi_ba_solver* s, tmi_ba_allreduce_fn fn,
                                    void* user);
/* Native RCCL transport (optional, instead of the hook).  The engine resolves
 * ncclGetUniqueId / ncclCommInitRank / ncclAllReduce at run time from the librccl
 * already loaded in the process (PyTorch's) or from /opt/rocm, and issues
 * ncclAllReduce(buf, buf, n, ncclDouble, ncclSum, comm, engine stream) itself:
 *   rank 0 : tmi_ba_rccl_unique_id(id)  -> ship the 128 bytes to every rank
 *   all    : tmi_ba_solver_init_rccl(solver, id)   (rank / world from create)
 * A hook set with tmi_ba_solver_set_allreduce is ignored once RCCL is initialised;
 * tmi_ba_solver_init_rccl(solver, NULL) destroys the communicator again (fall back to the hook). */
int32_t tmi_ba_rccl_unique_id(uint8_t id[128]);
int32_t tmi_ba_solver_init_rccl(tmi_ba_solver* s, const uint8_t id[128]);
/* Test aid: sums the solver's 8-double scalar buffer through the configured
 * transport (even for world = 1) after filling it with `value`; returns the
 * first element in *out. */
int32_t tmi_ba_solver_debug_allreduce(tmi_ba_solver* s, double value, double* out);
int32_t tmi_ba_solver_solve(tmi_ba_solver* s, const tmi_ba_options* options,
                            tmi_ba_summary* summary);
int32_t tmi_ba_solver_reset(tmi_ba_solver* s);
int32_t tmi_ba_solver_set_parameters(tmi_ba_solver* s, const tmi_ba_problem* problem);
int32_t tmi_ba_solver_download(tmi_ba_solver* s, tmi_ba_problem* problem);
/* The HIP stream the engine launches on (so callers can record their own
 * events on it). */
void* tmi_ba_solver_stream(tmi_ba_solver* s);
void tmi_ba_solver_destroy(tmi_ba_solver* s);

/* Per-observation evaluation on the device, exposed for parity tests:
 * residuals [2*N], the reduced camera Jacobian blocks [2*D*N] (row major
 * 2 x D, columns = free extrinsics then free PRIVATE intrinsics of the observing
 * camera), the Jacobian w.r.t. the free intrinsics the camera SHARES with other
 * views [2*D*N] (zero when it shares none) and point Jacobian blocks
 * [2*point_dof*N], in the caller's observation order, without loss correction
 * or Jacobi scaling.  valid[i] = 0 where the reference functor returns false
 * (reprojection_error.h:75-77).  Any output pointer may be NULL.            */
int32_t tmi_ba_solver_evaluate(tmi_ba_solver* s, double* residuals,
                               double* jac_camera, double* jac_shared,
                               double* jac_point, uint8_t* valid,
                               int32_t* block_dim);

/* ---- steps either side of the full adjustment ----------------------- */

/* Post-BA outlier filter: theia::SetOutlierTracksToUnestimated
 * (src/theia/sfm/set_outlier_tracks_to_unestimated.cc:62-133; callers
 * global_reconstruction_estimator.cc:259-263, incremental_reconstruction_estimator.cc:525,592).
 * Every camera and track of the flattened problem counts as estimated.  Per track:
 *   flag 1  a projection lies behind its camera (Camera::ProjectPoint depth < 0, :101-105) or
 *           the mean squared reprojection error exceeds max_inlier_reprojection_error^2 (:110-115)
 *   flag 2  no pair of viewing rays subtends min_triangulation_angle_degrees
 *           (SufficientTriangulationAngle, triangulation.cc:236-250; :120-125)
 *   flag 0  the track stays estimated.
 * The caller applies the flags (Track::SetEstimated(false)); the return value of the
 * reference is num_bad_reprojections + num_insufficient_viewing_angles. */
typedef struct tmi_ba_filter_summary {
  int64_t num_estimated_tracks;
  int64_t num_bad_reprojections;
  int64_t num_insufficient_viewing_angles;
  double seconds;        /* wall time of the call */
  double kernel_seconds; /* the device kernel alone (HIP events) */
} tmi_ba_filter_summary;

/* On the parameters resident in `solver` (e.g. right after tmi_ba_solver_solve: nothing is
 * uploaded).  track_flag / track_mean_sq_error are indexed by the caller's track index,
 * [num_points]; entries of tracks owned by other ranks are left untouched and the counts
 * cover this rank's tracks.  Either output may be NULL. */
int32_t tmi_ba_solver_filter_outlier_tracks(tmi_ba_solver* solver,
                                            double max_inlier_reprojection_error,
                                            double min_triangulation_angle_degrees,
                                            uint8_t* track_flag, double* track_mean_sq_error,
                                            tmi_ba_filter_summary* summary);

/* One-shot form: uploads the problem, filters, frees everything. device < 0 = current. */
int32_t tmi_ba_filter_outlier_tracks(const tmi_ba_problem* problem, int32_t device,
                                     double max_inlier_reprojection_error,
                                     double min_triangulation_angle_degrees, uint8_t* track_flag,
                                     double* track_mean_sq_error, tmi_ba_filter_summary* summary);

/* Batched theia::BundleAdjustTrack (bundle_adjustment.cc:96-107, called once per track from
 * estimate_track.cc:238-246): every non-constant track of the problem is adjusted on its own
 * with all cameras and intrinsics held constant -- one independent trust-region problem per
 * GPU thread, same Levenberg-Marquardt semantics and options as tmi_ba_solve (the linear
 * solve is the track's own point_dof x point_dof system; linear_solver_type is irrelevant,
 * max_solver_time_in_seconds is not enforced).
 * track_termination[num_points]: 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE, 3 residual
 * evaluation failed at the start point, -1 not adjusted (constant or unobserved track).
 * BundleAdjustmentSummary::success of the per-track call == (termination is 0 or 1); points
 * are updated exactly in that case.  All per-track outputs may be NULL. */
typedef struct tmi_ba_track_batch_summary {
  int64_t num_tracks;       /* tracks adjusted */
  int64_t num_success;
  int64_t total_iterations; /* sum of LM iterations over the tracks */
  double seconds;
  double kernel_seconds;
} tmi_ba_track_batch_summary;

int32_t tmi_ba_solver_adjust_tracks(tmi_ba_solver* solver, const tmi_ba_options* options,
                                    int8_t* track_termination, int32_t* track_iterations,
                                    double* track_initial_cost, double* track_final_cost,
                                    tmi_ba_track_batch_summary* summary);

/* One-shot form; problem->points is updated in place. */
int32_t tmi_ba_adjust_tracks(tmi_ba_problem* problem, const tmi_ba_options* options,
                             int8_t* track_termination, int32_t* track_iterations,
                             double* track_initial_cost, double* track_final_cost,
                             tmi_ba_track_batch_summary* summary);

/* Pre-BA track sub-sampling: theia::SelectGoodTracksForBundleAdjustment
 * (src/theia/sfm/select_good_tracks_for_bundle_adjustment.cc:251-327; callers
 * global_reconstruction_estimator.cc:475-486, incremental_reconstruction_estimator.cc:497-515).
 * Track statistics (:81-110: observation count truncated at long_track_length_threshold, mean
 * squared reprojection error over all observations) are computed on the device; the
 * selection logic runs on the host:
 *   1. per view, the features are binned into image_grid_cell_size_pixels cells and the
 *      track with the minimum (truncated length, mean error) of every occupied cell is
 *      selected (:150-196 with the comparator at :65-69);
 *   2. per view, if fewer than min_num_optimized_tracks_per_view of its tracks are selected,
 *      the not yet selected ones with the smallest track index are added (:201-249 ranks
 *      std::pair<TrackId, statistics> with the default operator<).
 * The reference iterates unordered containers; the engine fixes what that leaves open: views
 * in ascending index order, grid-cell ties to the smaller track index.
 * view_mask[num_cameras] (NULL = every view): the views whose features take part in steps 1
 * and 2 -- the reference's overload with an explicit view set (:280-327); the statistics of a
 * track always cover all of its observations (:81-110).
 * selected[num_points] (required): 1 = optimise this track.  stats_len / stats_err
 * [num_points] may be NULL.  Needs an unsharded handle (world == 1). */
typedef struct tmi_ba_select_summary {
  int64_t num_tracks;
  int64_t num_selected;
  int64_t num_selected_grid; /* selected by the grid step alone */
  double seconds;
  double kernel_seconds;     /* the statistics kernel (HIP events) */
} tmi_ba_select_summary;

int32_t tmi_ba_solver_select_good_tracks(tmi_ba_solver* solver, int32_t long_track_length_threshold,
                                         int32_t image_grid_cell_size_pixels,
                                         int32_t min_num_optimized_tracks_per_view,
                                         const uint8_t* view_mask, uint8_t* selected,
                                         int32_t* stats_len, double* stats_err,
                                         tmi_ba_select_summary* summary);

int32_t tmi_ba_select_good_tracks(const tmi_ba_problem* problem, int32_t device,
                                  int32_t long_track_length_threshold,
                                  int32_t image_grid_cell_size_pixels,
                                  int32_t min_num_optimized_tracks_per_view,
                                  const uint8_t* view_mask, uint8_t* selected,
                                  int32_t* stats_len, double* stats_err,
                                  tmi_ba_select_summary* summary);

/* Batched theia::BundleAdjustTwoViews (src/theia/sfm/bundle_adjustment/bundle_adjust_two_views.cc:
 * 113-191; called once per verified view pair from two_view_match_geometric_verification.cc:285):
 * every pair is an independent small bundle adjustment -- camera 1 extrinsics constant, camera 2
 * extrinsics free, each camera's intrinsics constant or free in the focal length only (:66-98),
 * one homogeneous point per correspondence seen by both cameras, no loss, DENSE_SCHUR, at most
 * max_num_iterations (the reference: 200) iterations, otherwise Ceres' default solver options
 * (SetSolverOptions :58-68).  One wavefront per pair runs the pair's whole trust-region solve.
 * Arrays are caller-owned; extrinsics2, intrinsics1/2 (focal length) and points are updated in place
 * for the pairs whose termination is 0 or 1 (Ceres' IsSolutionUsable).
 *   point_dof: 4 = the reference (no point parameterization); 3 holds the homogeneous w fixed.
 *   pair_termination / pair_iterations / pair_initial_cost / pair_final_cost [num_pairs] may be NULL;
 *   termination: 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE, 3 residual evaluation failed at the
 *   start point, -1 pair without correspondences. */
typedef struct tmi_ba_two_view_batch {
  int32_t num_pairs;
  const double* extrinsics1;            /* [6 * num_pairs] held constant                       */
  double* extrinsics2;                  /* [6 * num_pairs] in/out                              */
  const int32_t* model1;                /* [num_pairs] tmi_ba_camera_model                     */
  const int32_t* model2;
  double* intrinsics1;                  /* [10 * num_pairs] model order, zero padded; in/out   */
  double* intrinsics2;
  const uint8_t* constant_intrinsics1;  /* [num_pairs] TwoViewBundleAdjustmentOptions flags;   */
  const uint8_t* constant_intrinsics2;  /*   NULL = constant (the reference default)           */
  const int64_t* correspondence_ptr;    /* [num_pairs + 1] ranges into the arrays below        */
  const double* features1;              /* [2 * N] pixel in image 1                            */
  const double* features2;              /* [2 * N] pixel in image 2                            */
  double* points;                       /* [4 * N] homogeneous points, in/out                  */
} tmi_ba_two_view_batch;

int32_t tmi_ba_adjust_two_views(tmi_ba_two_view_batch* batch, int32_t point_dof,
                                int32_t max_num_iterations, int32_t device,
                                int8_t* pair_termination, int32_t* pair_iterations,
                                double* pair_initial_cost, double* pair_final_cost,
                                tmi_ba_track_batch_summary* summary);

/* ---- batched BundleAdjustTwoViewsAngular --------------------------------------------------
 * reference: bundle_adjust_two_views.cc:193-240 -- the relative pose of a view pair from
 * its correspondences alone: parameters TwoViewInfo::rotation_2 (angle-axis, 3) and
 * TwoViewInfo::position_2 (3, kept on the unit sphere by
 * AutoDiffLocalParameterization<UnitNormThreeVectorParameterization, 3, 3>,
 * unit_norm_three_vector_parameterization.h:45-63), one AngularEpipolarError residual per
 * correspondence (angular_epipolar_error.h:47-89; features in normalised image coordinates), no
 * loss, DENSE_SCHUR with the ordering left to Ceres, at most 200 iterations, Ceres' default
 * tolerances.  One wavefront per pair on the device.  Termination codes as for
 * tmi_ba_adjust_two_views; rotation2 / position2 are written back for usable solutions. */
typedef struct tmi_ba_two_view_angular_batch {
  int32_t num_pairs;
  double* rotation2;                    /* [3 * num_pairs] in/out                              */
  double* position2;                    /* [3 * num_pairs] in/out, unit norm                   */
  const int64_t* correspondence_ptr;    /* [num_pairs + 1] ranges into the arrays below        */
  const double* features1;              /* [2 * N] normalised image coordinates in view 1      */
  const double* features2;              /* [2 * N] ... in view 2                               */
} tmi_ba_two_view_angular_batch;

int32_t tmi_ba_adjust_two_views_angular(tmi_ba_two_view_angular_batch* batch, int32_t max_num_iterations,
                                        int32_t device, int8_t* pair_termination, int32_t* pair_iterations,
                                        double* pair_initial_cost, double* pair_final_cost,
                                        tmi_ba_track_batch_summary* summary);

/* Test hook: FNV-1a checksums of the static structure arrays resident in HBM -- built in HBM by
 * sort / scan kernels (one rank, no shared intrinsics blocks; TMI_BA_HOST_SETUP=1 disables) or on
 * host threads otherwise.  out[0] = 1 when the device built it; the other slots are documented at
 * the definition (engine.hip).  The two builders must agree array for array. */
int32_t tmi_ba_solver_structure_checksums(tmi_ba_solver* solver, uint64_t out[24]);

/* Which kernels a handle will run -- decided at create from the problem's shape and size (bench.py reports the
 * bytes of the kernel that ran instead of re-deriving the engine's rules):
 *   out[0] the one-sweep matrix-free product is built (mf_chunks.h)        out[1] position columns formed from Jp
 *   out[2] matrix-free LM iterations build the camera side without camera-major records (direct_diag.h)
 *   out[3] schur_mode auto chooses the operator per LM iteration           out[4] S is never formed (implicit)
 *   out[5] PCG length up to which the matrix-free operator is taken (auto)
 *   out[6] the handle holds clusters for CLUSTER_JACOBI (0: such a request keeps the SCHUR_JACOBI blocks, and
 *          tmi_ba_summary.effective_preconditioner_type says so)
 *   out[7] the planes of the handle's last linearisation are COMPACT: on the all-PINHOLE / default-mask / TRIVIAL-loss
 *          problem with unit aspect ratio and zero skew the camera block is formed from the point block, the normalised
 *          image point, the track and the view instead of stored (TMI_BA_COMPACT_PLANES=0 switches it off)              */
int32_t tmi_ba_solver_operator_info(tmi_ba_solver* solver, int32_t out[8]);

/* Host-only: statistics of the static structure the engine would build for
 * rank `rank` of `world` (no GPU needed).  Used by the CPU tests of the track
 * sharding: out[0] tracks owned, out[1] observations owned, out[2] reduced
 * blocks, out[3] block dimension D, out[4] upper off-diagonal blocks of S,
 * out[5] BSR blocks, out[6] observation pairs owned, out[7] checksum of the
 * (rank independent) block list, out[8] slices, out[9] padded observations,
 * out[10] checksum of the owned caller observation indices, out[11] sum over
 * owned pairs of a (slot independent) pair key.  Returns a tmi_ba_status. */
int32_t tmi_ba_structure_stats(const tmi_ba_problem* problem, int32_t rank, int32_t world,
                               int64_t out[12]);
/* The same for the dealing a handle actually uses: slices go to the ranks longest-work-first, and "work" is Schur
 * pairs + 5 x observations where the handle forms S (forms_S = 1: schur_mode explicit, exact solvers -- what
 * tmi_ba_structure_stats reports) but the track's observations where the operator is matrix-free (forms_S = 0: the
 * default of a solve on several ranks, schur_mode auto / implicit). */
int32_t tmi_ba_structure_stats_for(const tmi_ba_problem* problem, int32_t rank, int32_t world, int32_t forms_S,
                                   int64_t out[12]);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* THEIA_MI355_BA_H_ */
