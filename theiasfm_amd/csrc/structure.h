// Host-side static structure of one bundle-adjustment problem: everything that
// does not change over the Levenberg-Marquardt iterations.  Built once per
// tmi_ba_solver_create (counted in setup time, like Ceres' preprocessor in
// bundle_adjuster.cc:210-211) and uploaded to HBM.
//
//   * reduced camera blocks ("rblocks"): the free extrinsics columns of a
//     camera merged with the free intrinsics columns of its (private)
//     intrinsics group, zero padded to a uniform dimension D;
//   * track-major order: tracks sorted by length and packed in slices of 64
//     (one wavefront) so that "thread t handles track 64 s + t, observation j"
//     reads element  slice_ptr[s] + 64 j + t  -- perfectly coalesced;
//   * camera-major order: a slot per observation grouped by rblock, the order
//     the per-camera reductions and the Schur pair gathers read;
//   * the block structure of the reduced camera matrix S (symmetric storage:
//     upper blocks by rows + a column index for the transposed products) and, per structurally non-zero upper block (bi < bj), the
//     list of observation pairs (slot_i, slot_j) whose tracks are seen by both
//     cameras -- S_ij = - sum_pairs Y_i Y_j^T is then a gather, with no atomics
//     and a fixed summation order (bit-reproducible).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/theia_mi355_ba.h"

namespace tmi {

struct Structure {
  int D = 0;        // uniform (padded) reduced block dimension
  int Nc = 0, G = 0;
  int Np_total = 0;  // tracks in the whole problem
  int Np = 0;        // tracks owned by this rank
  int nslices = 0;   // ceil(Np / 64)
  int Np_pad = 0;    // 64 * nslices
  int64_t No = 0;    // observations owned by this rank
  int64_t No_pad = 0;
  int Nrb = 0;
  int rank = 0, world = 1;

  // rblocks: [0, Ncam_rb) belong to cameras, [Ncam_rb, Nrb) to shared intrinsics groups
  bool has_shared = false;
  int Ncam_rb = 0;
  std::vector<int> cam_rb;        // [Nc] rblock of a camera or -1
  std::vector<int> cam_grb;       // [Nc] rblock of the camera's SHARED free intrinsics or -1
  std::vector<int> rb_cam;        // [Nrb] camera of a camera block, -1 for a shared intrinsics block
  std::vector<int> rb_grp;        // [Nrb] intrinsics group the block's intrinsics columns belong to
  std::vector<uint32_t> grp_mask; // [G] free intrinsics of a SHARED group (0 otherwise)
  std::vector<int> cam_cross_u;   // [Nc] upper block (cam_rb, cam_grb) or -1
  std::vector<int> grp_cam_ptr;   // [Nrb-Ncam_rb+1] views of each shared block ...
  std::vector<int> grp_cams;      //   ... listed here
  std::vector<int> obs_gslot;     // [No_pad] slot of the (track, shared block) record or -1
  std::vector<uint8_t> obs_gflag; // [No_pad] bit0 first / bit1 last observation of its run
  std::vector<int> rb_dim;        // [Nrb] true (unpadded) dimension
  std::vector<int8_t> rb_cols;    // [Nrb*D] 0..5 extrinsics idx, 6+j intrinsics idx, -1 padding
  std::vector<uint32_t> cam_mask; // [Nc] bit c set = column c of [ext(6) | intr(10)] is free
  std::vector<uint32_t> grp_free; // [G] free intrinsics bits of a group (shared or private)
  std::vector<int> cam_group;     // [Nc] intrinsics group of a camera
  std::vector<int> group_offset;  // [G+1] offsets into the intrinsics array

  // track-major (SELL-64) layout
  std::vector<int> pt_orig;       // [Np_pad] caller's track index or -1
  std::vector<int> pt_k;          // [Np_pad] track length (0 = padding)
  std::vector<int> slice_ptr;     // [nslices+1]
  int n_wide = 0;                 // leading slices with >= kWideK observations per track (device_view.h)
  int n_ultra = 0;                // leading slices with >= kUltraK observations per track (64 lanes per track)
  std::vector<int> obs_cam;       // [No_pad] camera or -1 (padding)
  std::vector<double> obs_xy;     // [2*No_pad]
  std::vector<int> obs_cpos;      // [No_pad] camera-major slot or -1
  std::vector<int64_t> obs_orig;  // [No_pad] caller's observation index or -1
  std::vector<uint8_t> pt_const;  // [Np_pad]
  std::vector<int> unobserved;    // tracks without observations (listed on rank 0 only)

  // camera-major layout
  int64_t Nslots = 0;
  std::vector<int> cam_ptr;       // [Nrb+1] slot ranges per rblock

  // reduced camera matrix structure (identical on every rank)
  int64_t nub = 0;                // upper off-diagonal blocks (bi < bj)
  std::vector<int> ub_i, ub_j;    // [nub] sorted by (bi, bj)
  int64_t nnzb = 0;               // blocks of S counting both triangles + diagonal (statistics)
  // S is stored symmetric: the upper off-diagonal blocks in (bi, bj) order (rows are
  // contiguous) plus the diagonal blocks.
  std::vector<int> urow_ptr;      // [Nrb+1] upper blocks of block row i: [urow_ptr[i], urow_ptr[i+1])
  std::vector<int> ucol_ptr;      // [Nrb+1] upper blocks in block column j ...
  std::vector<int> ucol_u;        // [nub]   ... listed here, ascending bi
  std::vector<int> spc_row, spc_u0;  // SpMV rows pass: chunk -> (block row, first upper block)
  std::vector<int> spc_rptr;      // [Nrb+1] chunks of block row i
  // pair lists (this rank's tracks only)
  int64_t npairs = 0;
  std::vector<int64_t> pair_ptr;  // [nub+1]
  std::vector<int> pair_i, pair_j;  // [npairs] camera-major slots
  std::vector<int> ub_order;      // launch order of the upper blocks (XCD-aware, -1 = padding)

  std::string error;
};

// Camera-side part only (argument checks, reduced blocks, column maps); used by both builders.
int build_blocks(const tmi_ba_problem* P, int rank, int world, Structure* out);

// Returns TMI_BA_OK or an error status (message in out->error).
// want_pairs = false skips the block structure of S and the pair lists (implicit Schur
// operator: S is never formed).

// want_pairs = 2: only the blocks INSIDE a cluster {shared intrinsics block, its views} and their pair lists -- what
// CLUSTER_JACOBI needs when the operator itself is matrix-free (cluster_precond.h); without shared blocks: none.
int build_structure(const tmi_ba_problem* P, int rank, int world, Structure* out,
                    int want_pairs = 1);

}  // namespace tmi
