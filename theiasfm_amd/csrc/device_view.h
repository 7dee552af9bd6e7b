// Plain-pointer view of one resident problem in HBM, passed by value to every
// kernel.  Layout notes (all fp64 unless stated):
//
//  parameters        ext[Nc][6]  intr[sum sizes]  pts[Np_pad][4]
//  track-major SELL  element e = slice_ptr[s] + 64 j + t  (track 64 s + t, its j-th obs)
//    pm_r   [2][No_pad]      robustified residual
//    pm_A   [2 D][No_pad]    plane 2*col+row: reduced camera Jacobian (scaled, robustified)
//    pm_Jp  [2 DP][No_pad]   plane 2*col+row: point Jacobian
//  camera-major (slot = obs_cpos[e], contiguous per reduced block)
//    cm_Y   [Nslots][YS]     Y = A^T Jp L^-T  (D x DP row-major, YS = D*DP rounded up to even)
//    cm_A   [Nslots][AS]     A row 0 (D), A row 1 (D), N = I - Q Q^T (3), r~(2), r(2); when free intrinsics are shared
//                            between views: A rows, Q rows (2 x 4), r~, r, A1 row 0 (D), A1 row 1 (D)  (kernels.h, sh_off_*)
//  reduced system
//    red    [nub*D*D | Nrb*D*D | Nrb*D | Nrb*D | Nrb*D | 8]   the all-reduce buffer:
//           upper blocks, raw diagonal blocks, U diagonal, reduced gradient g~,
//           camera gradient g_c, scalars
//    S is symmetric and lives in `red` itself: upper blocks red[ub + u D^2] (rows contiguous)
//    Sdiag  [Nrb][D*D]       diagonal blocks with the LM diagonal added
//    tbuf   [nub][D]         transposed block products of the symmetric SpMV
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

namespace tmi {

// SpMV rows pass: a wave streams kSpmvTrips trips of 64 / D upper blocks (its chunk);
// shared by the host work list (structure.cpp) and the kernel (kernels.h)
constexpr int kSpmvTrips = 8;
// free columns of [extrinsics (bits 0-5) | intrinsics in model order (bits 6-15)] for a PINHOLE view under the reference's
// default intrinsics_to_optimize = FOCAL_LENGTH | RADIAL_DISTORTION (bundle_adjustment.h:95; pinhole_camera_model.h:86-94:
// f = intrinsic 0, k1 = 5, k2 = 6): what BAL problems and an unchanged Theia caller have for every view
constexpr unsigned kPinholeDefaultMask = 0x3Fu | (1u << 6) | (1u << 11) | (1u << 12);
// per-track kernels: slices whose longest track has at least this many observations are run
// with 16 lanes per track (kernels.h, track_map); the larger value applies when a rank holds
// >= 5000 slices (structure.cpp)
constexpr int kWideK = 12;
constexpr int kWideKLarge = 20;
// ... and slices whose longest track has at least this many with 64 lanes (a whole wavefront) per track
constexpr int kUltraK = 96;

struct DeviceView {
  int Nc, G, Np_pad, nslices, Nrb, D, DP;
  int n_ultra;         // leading slices run with 64 lanes per track (n_ultra <= n_wide)
  int n_wide;          // leading slices run with 16 (the first n_ultra: 64) lanes per track
  int n_track_blocks;  // grid of the per-track kernels: 16 n_ultra + 4 (n_wide - n_ultra) + ceil((nslices - n_wide) / 4)
  int Ncam_rb;     // blocks [0, Ncam_rb) are cameras, [Ncam_rb, Nrb) shared intrinsics groups
  int has_shared;
  int No_pad;
  int Nslots;
  int nub, nnzb;
  int n_order;  // entries of ub_order (multiple of 32)
  long long npairs;

  // parameters (current and candidate)
  double* ext;
  double* intr;
  double* pts;
  double* ext_c;
  double* intr_c;
  double* pts_c;

  // static structure
  const int* slice_ptr;
  const int* pt_k;
  const unsigned char* pt_const;
  const int* obs_cam;
  const double* obs_xy;  // [No_pad][2]
  const int* obs_cpos;
  const int* obs_rb;      // [No_pad] reduced block of the observation's view (cam_rb[obs_cam], -1: constant view):
                          //   one dependent load less in the per-track sweeps that only need the block
  const int* slot_track;  // [Nslots] track (position in the rank's slice order) of a camera-major slot
                          //   (matrix-free product without shared blocks; null otherwise)
  const int* cam_grp;
  const int4* cam_rec;         // [Nc] {camera model, intrinsics offset, #intrinsics, free-column mask}
  int uniform_pinhole_default; // every camera: PINHOLE, free columns = extrinsics + f + k1 + k2 (kPinholeDefaultMask) -- or
                               //   none at all (a constant view has no block)
  const int* cam_rb;
  const unsigned* cam_mask;
  const int* grp_model;
  const int* grp_off;
  const int* rb_cam;           // camera of a camera block, -1 for a shared intrinsics block
  const int* rb_grp;           // group whose intrinsics the block's intrinsics columns address
  const int* cam_grb;          // [Nc] block of the camera's shared free intrinsics or -1
  const unsigned* grp_mask;    // [G] free intrinsics bits of a shared group
  const int* obs_gslot;        // [No_pad] slot of the (track, shared block) record or -1
  const unsigned char* obs_gflag;  // bit0 first / bit1 last observation of its run
  const int* cam_cross_u;      // [Nc] upper block (camera block, shared block) or -1
  const int* grp_cam_ptr;      // views of each shared block
  const int* grp_cams;
  const signed char* rb_cols;  // [Nrb][D]
  int write_y;                 // point_eliminate writes the Y records (explicit S, or shared intrinsics blocks)
  const int* cam_ptr;
  const int* urow_ptr;
  const int* ub_i;
  const int* ub_j;
  const int* ucol_ptr;
  const int* ucol_u;
  const int* spc_row;   // SpMV rows pass work list: chunk -> block row, first upper block
  const int* spc_u0;
  const int* spc_rptr;  // [Nrb+1] chunks of a block row
  int n_spc;
  const long long* pair_ptr;
  const int* pair_i;
  const int* pair_j;
  const int* ub_order;

  // work arrays
  double* pm_r;
  double* pm_A;
  // drop_pos (round 4): the three POSITION columns of the camera block are not stored in pm_A (its planes 0..5 are
  // holes nobody touches).  With every block's position free and no constant point they are, column by column,
  //   A_pos[a] = -w Jp[a] scale_c[a] / scale_p[a]   (reprojection_error.h: d r / d C = -w d r / d X; the loss
  // corrector is the same linear map on both), so the consumers form them from the Jp planes they read anyway:
  // pos_coef[a][track] = -w / scale_p[a] (pos_coef_kernel, after every linearize), the view's scale_c[a] is folded
  // into the vector a product gathers (xs, pos_scale_kernel) and into the reduce launch.  48 of the 208 plane bytes
  // per observation less for linearize, point_eliminate, back_substitute and every matrix-free product.
  int drop_pos;
  double* pos_coef;  // [3][Np_pad]
  double* xs;        // [Nrb D] the vector of the running product / back-substitution, position entries times scale_c
  // compact (round 6, last session): on the all-PINHOLE / default-mask / TRIVIAL-loss problem with unit aspect ratio and
  // zero skew the 2 x 9 camera block is a function of the point block Jp (2 x 3, stored anyway), the normalised image
  // point p_n (2 doubles), the track's X and the VIEW's R, C, Jl, f, k1, k2 and column scales:
  //   A_pos  = -w Jp' S_pos                      Jp' = Jp diag(1 / scale_p) = d r / d X[0:3]     (drop_pos, above)
  //   A_rot  = -Jp' [X - w C]x R^T Jl S_rot      (d q / d w_k = Jl[:,k] x q, q = R (X - w C): camera_models.h)
  //   A_int  = p_n [dist s_f, f r^2 s_k1, f r^4 s_k2],  r^2 = |p_n|^2, dist = 1 + k1 r^2 + k2 r^4
  // so pm_A holds ONE 16-byte plane pair (p_n) instead of six, the product / back-substitution gather a TRANSFORMED
  // view vector [kappa | eta | a0 a1 a2] (compact_forward, kernels.h) instead of [s_pos x_pos | x_rot | x_int] and form
  //   A x = Jp' (eta x X - w kappa) + p_n (a0 + a1 r^2 + a2 r^4),
  // and A^T t is accumulated per view as moments [-w h | X x h | (p_n . t)(1, r^2, r^4)], h = Jp'^T t, which
  // reduce_kernel maps back (compact_backward).  80 of the 160 plane bytes per observation less for linearize,
  // back_substitute and every matrix-free product.  Set per linearize by the engine (solve); 0: the full planes.
  int compact;
  int sums_ready;   // Vraw / gp already hold this linearisation's V = sum Jp^T Jp and g_p (the compact linearize): point_eliminate
                    //   (REC = false) loads them instead of sweeping the Jp and r planes
  double* cp_trk;   // [7][Np_pad] planes X0 X1 X2 w 1/scale_p[0..2] of the linearisation the planes belong to (linearize)
  double* xz;       // [Nrb D] the transformed z of a PCG step (pcg_step: xs <- xz + beta xs)
  double* pm_Jp;
  double* pm_A1;    // [2 D][No_pad] shared-intrinsics Jacobian columns (has_shared only)
  int planes_fp32;  // round 5: residual_precision = 32 on a problem with shared intrinsics blocks STORES pm_r / pm_A / pm_A1 /
                    //   pm_Jp as float (same tile layout in elements; the allocations stay sized for doubles)
  double* cam_part; // [Ncam_rb][2 D^2 + 3 D] per-view sums for the shared blocks
  double* cm_Y;
  double* cm_A;
  double* cm_R;  // tail records {N, r~, r} when there are no shared intrinsics blocks (kernels.h asa_of)
  // direct_diag (round 5, direct_diag.h): this LM iteration is matrix-free with the one-sweep product, so point_eliminate
  // writes NO camera-major records; it leaves trk_rec and camera_diag_direct re-evaluates the observations view by view
  int direct_diag;
  double* trk_rec;  // [Np_pad][16 (3-dof) | 24 (4-dof)] {X, scale_p, L^-1, t_p}
  double* scale_c;  // [Nrb][D]
  double* scale_cam; // [Nc][16] scale_c expanded to the columns [ext(6) | intr(10)] of every view
                     //   (0 on constant columns): linearize indexes it statically
  double* scale_p;  // [Np_pad][DP]
  double* prep;     // [Nc][kPrepStride] prepared camera records of (ext, intr)     (camera_models.h)
  double* prep_c;   // ... of the candidate (ext_c, intr_c); swapped with prep on acceptance
  double* Vinv;     // [DP(DP+1)/2][Np_pad]  planes, symmetric inverse of V + Dp
  double* gp;       // [DP][Np_pad]
  double* Linv;     // [DP(DP+1)/2][Np_pad]  planes, L^-1 (lower; plane sym_idx(a, b), a <= b, holds element (b, a))
  double* Vraw;     // [DP(DP+1)/2][Np_pad]  planes, V = Jp^T Jp without the damping
  double* yp;       // [DP][Np_pad]
  double* red;      // all-reduce buffer
  double* Sdiag;    // [Nrb][D*D] diagonal blocks + LM diagonal
  double* tbuf;     // [nub][D]
  double* rbuf;     // [n_spc][D] per-chunk partial row products of the SpMV
  double* Minv;     // [Nrb][D*D] inverse diagonal blocks
  double* rhs;      // [Nrb*D]
  double* yc;       // [Nrb*D]
  double* cg_r;
  double* cg_z;
  double* cg_p;
  double* cg_q;
  double* cg_t;
  double* partial;  // per-workgroup partial sums (several lanes of scalars)
  double* scal;     // device scalars
  int* flags;       // device flags (invalid residual, singular block, ...)
  double* dotbuf;   // [Nrb] per-block partial dot products of the product kernels
  int* ticket;      // [4][kTicketStride] arrival counters of the "last workgroup finishes the reduction" kernels
  int* pcg_done;    // set by pcg_step once PCG has stopped: pcg_p, enqueued behind it, returns at once
};

// host-visible copy of the device scalars (pinned, mapped, coherent memory): published by a
// kernel with system-scope stores + a release store of `seq`, polled by the host
struct HostMirror {
  double scal[32];
  double red[8];
  int flags[8];
  unsigned long long seq;
};

// offsets into `red`
struct RedLayout {
  long long ub, diag, udiag, gt, gc, scalars, total;
};
inline RedLayout red_layout(long long nub, int Nrb, int D) {
  RedLayout L;
  L.ub = 0;
  L.diag = L.ub + nub * D * D;
  L.udiag = L.diag + (long long)Nrb * D * D;
  L.gt = L.udiag + (long long)Nrb * D;
  L.gc = L.gt + (long long)Nrb * D;
  L.scalars = L.gc + (long long)Nrb * D;
  L.total = L.scalars + 8;
  return L;
}

// slots in DeviceView::scal
enum {
  SC_COST = 0,      // 1/2 sum rho at the linearisation point
  SC_SS = 1,        // sum of squared raw residuals
  SC_CAND_COST = 2,
  SC_CAND_SS = 3,
  SC_MCC = 4,       // model cost change
  SC_STEP_SQ = 5,
  SC_XNORM_SQ = 6,
  SC_GMAX = 7,
  SC_RHO = 8,       // PCG scalars
  SC_LAST_RHO = 9,
  SC_PQ = 10,
  SC_ALPHA = 11,
  SC_Q0 = 12,
  SC_Q1 = 13,
  SC_ZETA = 14,
  SC_BNORM = 15,
  SC_GMAX_P = 16,
  SC_RHO_BAD = 17,  // the next iteration's rho is zero or not finite
  SC_II_DEXT = 18,  // inner iterations: |x - candidate|^2 over extrinsics / intrinsics,
  SC_II_DINTR = 19,
  SC_II_XC = 22,    //   |candidate|^2 over the non-constant camera blocks
  SC_PCG_STOP = 24, // pcg_step: its stopping test held (what DeviceView::pcg_done says, for the host)
  SC_COUNT = 32
};
// DeviceView::flags
enum { FL_INVALID = 0, FL_SINGULAR_POINT = 1, FL_SINGULAR_BLOCK = 2, FL_PCG_FAIL = 3, FL_CHOL_ABORT = 4, FL_COUNT = 8 };

}  // namespace tmi
