// Host driver + C ABI of the MI355X bundle-adjustment engine.
//
// tmi_ba_solver_solve() is the replacement for the ceres::Solve call at
// src/theia/sfm/bundle_adjustment/bundle_adjuster.cc:205: a trust-region
// Levenberg-Marquardt loop with Ceres' semantics (restated from
// Ceres 1.14's TrustRegionMinimizer / LevenbergMarquardtStrategy /
// ConjugateGradientsSolver since Ceres itself is an external dependency of the
// reference) whose every numeric phase is a HIP kernel from kernels.h.  The
// host only sequences launches and takes the accept / reject decisions from a
// handful of scalars read back once per iteration (once per PCG iteration
// inside the linear solve).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/theia_mi355_ba.h"
#include "dense_cholesky.h"
#include "dense_cholesky_df.h"
#include "cluster_chains.h"
#include "cluster_precond.h"
#include "kernels.h"
#include "mf_chunks.h"
#include "direct_diag.h"
#include "pcg_persist.h"
#include "track_kernels.h"
#include "inner_kernels.h"
#include "two_view_kernels.h"
#include "select_kernels.h"
#include "structure_gpu.h"
#include <hipcub/hipcub.hpp>
#include "structure.h"

#ifndef TMI_LIN_OCC_COMPACT
#define TMI_LIN_OCC_COMPACT 3  // ... and its compact instantiation (45 KB of LDS: three workgroups fit a CU)
#endif
#ifndef TMI_LIN_OCC
#define TMI_LIN_OCC 2  // workgroups per CU the specialised linearize is compiled for (A/B builds: -DTMI_LIN_OCC=3)
#endif
namespace tmi {

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define TMI_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      char buf_[256];                                                                  \
      snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
               __FILE__, __LINE__);                                                    \
      s->error = buf_;                                                                 \
      return (e_ == hipErrorOutOfMemory) ? TMI_BA_ERR_OUT_OF_MEMORY : TMI_BA_ERR_DEVICE; \
    }                                                                                  \
  } while (0)

// ---- per (D, DP) launch table --------------------------------------------------
struct Launch {
  // evaluation at a parameter set through its prepared camera records; `sums` / `flag_dst`:
  // where the kernel's last workgroup leaves the grid-wide sums / the invalid-residual vote
  // norms != 0 (fp64 evaluation only): no planes; cost sums and scale_p from the unscaled column norms (kernels.h, NORMS)
  void (*linearize)(const DeviceView&, hipStream_t, const double* prep, int, double, int, double* sums, int norms);
  void (*track_records_points_only)(const DeviceView&, hipStream_t, const double* pts);
  void (*cost)(const DeviceView&, hipStream_t, const double* prep, const double* pts, int, double, int, int,
               double* partial, double* sums, double* flag_dst);
  void (*point_scale)(const DeviceView&, hipStream_t, int);
  void (*shared_blocks)(const DeviceView&, hipStream_t, RedLayout);
  void (*cross_add)(const DeviceView&, hipStream_t, RedLayout);
  void (*point_eliminate)(const DeviceView&, hipStream_t, double, double, double, int, double*, double*, double,
                          double*);
  void (*camera_diag)(const DeviceView&, hipStream_t, RedLayout, int, double*);
  // direct_diag.h: the same sums without camera-major records (matrix-free iterations, no shared intrinsics blocks)
  void (*camera_diag_direct)(const DeviceView&, hipStream_t, RedLayout, const ddg::Plan&, const double* prep, int, double,
                             int skip_reduce);
  // direct_diag.h, camera_finish_kernel: [the chunk sums of camera_diag_direct ->] diagonal blocks + LM diagonal ->
  // SCHUR_JACOBI inverses -> start of PCG, one launch
  void (*camera_finish)(const DeviceView&, hipStream_t, RedLayout, const ddg::Plan&, double, double, double, int want_gmax,
                        int mode, int direct);
  void (*schur_offdiag)(const DeviceView&, hipStream_t, RedLayout);
  void (*expand_scale)(const DeviceView&, hipStream_t);
  void (*expand)(const DeviceView&, hipStream_t, RedLayout, double, double, double, int want_gmax);
  void (*precond)(const DeviceView&, hipStream_t, int);
  void (*cluster_gather)(hipStream_t, const clp::GatherEntry*, int, const clp::ClusterDesc*, const double*, const double*, double*, double);
  void (*cluster_rz)(const DeviceView&, hipStream_t, int, double*);
  // dot: the product kernel also leaves x . y at y[Nrb D]
  void (*spmv)(const DeviceView&, hipStream_t, const double*, const double*, double*, int dot);
  void (*implicit_spmv)(const DeviceView&, hipStream_t, RedLayout, const double*, double*, double*,
                        double*, double, double, double, int, int, int dot);
  // the one-sweep matrix-free product of mf_chunks.h (no shared intrinsics blocks)
  // xs_ready: DeviceView::xs already holds x with the position entries scaled (pcg_init / pcg_p leave it for cg_p)
  // ev_a / ev_b (may be null): HIP events that take the start of the first and the end of the last kernel of the product
  void (*mf_product)(const DeviceView&, const mfc::View&, hipStream_t, RedLayout, const double*, double*, double, double,
                     double, int, int dot, int xs_ready, hipEvent_t ev_a, hipEvent_t ev_b, const int* guard);
  void (*pcg_step)(const DeviceView&, hipStream_t, const double* b, int it, int nb, double eta, int min_it,
                   int max_it, const double* red8, HostMirror* mirror, unsigned long long seq, const int* guard);
  void (*pcg_a)(const DeviceView&, hipStream_t, int, int);
  void (*pcg_b2)(const DeviceView&, hipStream_t, const double*, int, int, double*);
  void (*back_substitute)(const DeviceView&, hipStream_t, int, double*, double* sums);
  void (*update_cameras)(const DeviceView&, hipStream_t, double* out, double* prep_c);
  void (*pcg_init)(const DeviceView&, hipStream_t, const double* b, int nb);
  void (*pcg_persistent)(const DeviceView&, hipStream_t, int grid, const ppcg::Args&);
  void (*dense_gather)(const DeviceView&, hipStream_t, const double*, double*, int);
  void (*tile_gather)(const DeviceView&, hipStream_t, const double*, const double*, double*, int);
};

template <int D, int DP, bool SH>
Launch make_launch(bool fp32) {
  Launch L;
  if (fp32) {
    L.linearize = [](const DeviceView& v, hipStream_t st, const double* prep, int lt, double lw, int nb, double* sums, int) {
      hipLaunchKernelGGL((linearize_kernel<D, DP, SH, float, 2>), dim3(nb), dim3(256), 0, st, v, prep, lt, lw, nb, sums);
    };
    L.cost = [](const DeviceView& v, hipStream_t st, const double* prep, const double* p, int lt, double lw, int fl,
                int nb, double* partial, double* sums, double* flag_dst) {
      hipLaunchKernelGGL((cost_kernel<DP, float>), dim3(nb), dim3(256), 0, st, v, prep, p, lt, lw, fl, nb, partial, sums, flag_dst);
    };
  } else {
    L.linearize = [](const DeviceView& v, hipStream_t st, const double* prep, int lt, double lw, int nb, double* sums, int norms) {
      if constexpr (!SH) {
        if (norms) {
          // start of a solve: cost + the Jacobi scales of the point columns, no planes (kernels.h, NORMS)
          bool special = false;
          if constexpr (D == 9) {
            if (v.uniform_pinhole_default && lt == 0) {
              hipLaunchKernelGGL((linearize_kernel<D, DP, SH, double, 2, double, 0, kPinholeDefaultMask, true, true>), dim3(nb), dim3(256),
                                 0, st, v, prep, lt, lw, nb, sums);
              special = true;
            } else if (v.uniform_pinhole_default) {  // (a robust loss: the specialised body with the corrector left in)
              hipLaunchKernelGGL((linearize_kernel<D, DP, SH, double, 2, double, 0, kPinholeDefaultMask, true, true, false, true>), dim3(nb),
                                 dim3(256), 0, st, v, prep, lt, lw, nb, sums);
              special = true;
            }
          }
          if (!special)
            hipLaunchKernelGGL((linearize_kernel<D, DP, SH, double, 2, double, -1, 0u, false, true>), dim3(nb), dim3(256), 0, st, v,
                               prep, lt, lw, nb, sums);
          return;
        }
      }
      if constexpr (!SH && D == 9) {
        // every camera a PINHOLE with extrinsics + focal length + two radial terms free (the BAL / reference default,
        // bundle_adjustment.h:95) and no robust loss: the specialised body (kernels.h, UMODEL / UMASK)
        if (v.uniform_pinhole_default && lt == 0) {
          if (v.drop_pos && v.compact)  // compact planes (device_view.h): p_n instead of the camera block
            hipLaunchKernelGGL((linearize_kernel<D, DP, SH, double, TMI_LIN_OCC_COMPACT, double, 0, kPinholeDefaultMask, true, false, true>), dim3(nb),
                               dim3(256), 0, st, v, prep, lt, lw, nb, sums);
          else if (v.drop_pos)
            hipLaunchKernelGGL((linearize_kernel<D, DP, SH, double, TMI_LIN_OCC, double, 0, kPinholeDefaultMask, true>), dim3(nb), dim3(256), 0, st, v,
                               prep, lt, lw, nb, sums);
          else
            hipLaunchKernelGGL((linearize_kernel<D, DP, SH, double, 2, double, 0, kPinholeDefaultMask, false>), dim3(nb), dim3(256), 0, st, v,
                               prep, lt, lw, nb, sums);
          return;
        }
        // ... with a robust loss (the reference's application flags: HUBER): the same bodies with the corrector left in;
        // compact planes then hold [C p_n | r^2 .] per observation (DeviceView::compact == 2)
        if (v.uniform_pinhole_default && v.drop_pos) {
          if (v.compact)
            // (two workgroups per CU: with the corrector the body spills at the 168 registers three would allow)
            hipLaunchKernelGGL((linearize_kernel<D, DP, SH, double, TMI_LIN_OCC, double, 0, kPinholeDefaultMask, true, false, true, true>),
                               dim3(nb), dim3(256), 0, st, v, prep, lt, lw, nb, sums);
          else
            hipLaunchKernelGGL((linearize_kernel<D, DP, SH, double, TMI_LIN_OCC, double, 0, kPinholeDefaultMask, true, false, false, true>),
                               dim3(nb), dim3(256), 0, st, v, prep, lt, lw, nb, sums);
          return;
        }
      }
      hipLaunchKernelGGL((linearize_kernel<D, DP, SH, double, 2>), dim3(nb), dim3(256), 0, st, v, prep, lt, lw, nb, sums);
    };
    L.cost = [](const DeviceView& v, hipStream_t st, const double* prep, const double* p, int lt, double lw, int fl,
                int nb, double* partial, double* sums, double* flag_dst) {
      hipLaunchKernelGGL((cost_kernel<DP, double>), dim3(nb), dim3(256), 0, st, v, prep, p, lt, lw, fl, nb, partial, sums, flag_dst);
    };
  }
  L.point_scale = [](const DeviceView& v, hipStream_t st, int nb) {
    hipLaunchKernelGGL((point_scale_kernel<DP>), dim3(nb), dim3(256), 0, st, v);
  };
  L.track_records_points_only = [](const DeviceView& v, hipStream_t st, const double* pts) {
    const long long n = (long long)v.Np_pad * (ddg::trk_stride(DP) / 2);
    if (n && v.trk_rec)
      hipLaunchKernelGGL((ddg::track_records_points_only_kernel<DP>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v, pts);
  };
  L.shared_blocks = [](const DeviceView& v, hipStream_t st, RedLayout R) {
    if (!SH || v.Nrb == v.Ncam_rb) return;
    hipLaunchKernelGGL((camera_group_partials_kernel<D, DP>), dim3(v.Ncam_rb), dim3(64), 0, st, v);
    hipLaunchKernelGGL((group_reduce_kernel<D>), dim3(v.Nrb - v.Ncam_rb), dim3(256), 0, st, v, R);
  };
  L.cross_add = [](const DeviceView& v, hipStream_t st, RedLayout R) {
    if (!SH || v.Nrb == v.Ncam_rb) return;
    const int n = v.Ncam_rb * D * D;
    hipLaunchKernelGGL((cross_add_kernel<D>), dim3((n + 255) / 256), dim3(256), 0, st, v, R);
  };
  L.point_eliminate = [](const DeviceView& v, hipStream_t st, double ir, double lo, double hi, int nb,
                         double* pm, double* vote, double gtol, double* gvote) {
    if constexpr (!SH) {
      if (v.direct_diag) {
        hipLaunchKernelGGL((point_eliminate_kernel<D, DP, SH, false>), dim3(nb), dim3(256), 0, st, v, ir, lo, hi, nb, pm,
                           vote, gtol, gvote);
        return;
      }
    }
    hipLaunchKernelGGL((point_eliminate_kernel<D, DP, SH>), dim3(nb), dim3(256), 0, st, v, ir, lo, hi, nb, pm,
                       vote, gtol, gvote);
  };
  L.camera_diag = [](const DeviceView& v, hipStream_t st, RedLayout R, int max_chunks, double* chunk_partial) {
    // max_chunks > 0: the shared blocks' raw diagonals come from the chunked kernels (big shared blocks)
    const bool chunked = SH && max_chunks > 0 && v.Nrb > v.Ncam_rb;
    if (v.Nrb) hipLaunchKernelGGL((camera_diag_kernel<D, DP, SH>), dim3(v.Nrb), dim3(64), 0, st, v, R, chunked ? 1 : 0);
    if (chunked) {
      const int ns = v.Nrb - v.Ncam_rb;
      hipLaunchKernelGGL((shared_diag_partial_kernel<D, DP>), dim3(ns * max_chunks), dim3(64), 0, st, v, max_chunks, chunk_partial);
      hipLaunchKernelGGL((shared_diag_reduce_kernel<D>), dim3(ns), dim3(64), 0, st, v, R, max_chunks, chunk_partial);
    }
  };
  L.camera_diag_direct = [](const DeviceView& v, hipStream_t st, RedLayout R, const ddg::Plan& pl, const double* prep, int lt,
                            double lw, int skip_reduce) {
    if constexpr (!SH && D <= ddg::kMaxD) {
      if (pl.n_chunks) {
        bool special = false;
        if constexpr (D == 9) {
          if (v.uniform_pinhole_default && lt == 0) {  // (as linearize: every camera PINHOLE / default mask, no robust loss)
            hipLaunchKernelGGL((ddg::camera_diag_direct_kernel<D, DP, 0, kPinholeDefaultMask>), dim3(pl.n_chunks), dim3(64), 0, st, v, pl,
                               prep, lt, lw);
            special = true;
          } else if (v.uniform_pinhole_default) {  // ... with a robust loss: the same body, corrector left in
            hipLaunchKernelGGL((ddg::camera_diag_direct_kernel<D, DP, 0, kPinholeDefaultMask, true>), dim3(pl.n_chunks), dim3(64), 0, st,
                               v, pl, prep, lt, lw);
            special = true;
          }
        }
        if (!special)
          hipLaunchKernelGGL((ddg::camera_diag_direct_kernel<D, DP>), dim3(pl.n_chunks), dim3(64), 0, st, v, pl, prep, lt, lw);
      }
      // (skip_reduce: camera_finish sums the chunks itself)
      if (v.Nrb && !skip_reduce)
        hipLaunchKernelGGL((ddg::camera_diag_direct_reduce_kernel<D>), dim3(v.Nrb), dim3(64), 0, st, v, R, pl);
    }
  };
  L.camera_finish = [](const DeviceView& v, hipStream_t st, RedLayout R, const ddg::Plan& pl, double ir, double lo, double hi,
                       int want_gmax, int mode, int direct) {
    if (!v.Nrb) return;
    if constexpr (!SH && D <= ddg::kMaxD) {
      if (direct) {
        hipLaunchKernelGGL((ddg::camera_finish_kernel<D, true>), dim3(v.Nrb), dim3(64), 0, st, v, R, pl, ir, lo, hi, want_gmax, mode);
        return;
      }
    }
    hipLaunchKernelGGL((ddg::camera_finish_kernel<D, false>), dim3(v.Nrb), dim3(64), 0, st, v, R, pl, ir, lo, hi, want_gmax, mode);
  };
  L.expand_scale = [](const DeviceView& v, hipStream_t st) {
    if (v.Nc) hipLaunchKernelGGL((expand_camera_scale_kernel<D>), dim3((v.Nc + 255) / 256), dim3(256), 0, st, v);
  };
  L.schur_offdiag = [](const DeviceView& v, hipStream_t st, RedLayout R) {
    if (!v.nub) return;
    const dim3 grid((v.n_order + 4 * kSchurBlocksPerWave - 1) / (4 * kSchurBlocksPerWave));
    // without shared intrinsics blocks: the [A | Q] records (no Y record exists unless asked for)
    if (!SH && !v.write_y) hipLaunchKernelGGL((schur_offdiag_aq_kernel<D, DP>), grid, dim3(256), 0, st, v, R);
    else hipLaunchKernelGGL((schur_offdiag_kernel<D, DP>), grid, dim3(256), 0, st, v, R);
  };
  L.expand = [](const DeviceView& v, hipStream_t st, RedLayout R, double ir, double lo, double hi, int want_gmax) {
    const int n2 = v.Nrb * D * D;
    if (n2)
      hipLaunchKernelGGL((finish_diag_kernel<D>), dim3((n2 + 255) / 256), dim3(256), 0, st, v, R, ir,
                         lo, hi, want_gmax);
  };
  L.cluster_gather = [](hipStream_t st, const clp::GatherEntry* ge, int n, const clp::ClusterDesc* desc, const double* ub,
                        const double* Sdiag, double* tiles, double off_scale) {
    if (n) hipLaunchKernelGGL((clp::cluster_gather_kernel<D>), dim3((n + 3) / 4), dim3(256), 0, st, ge, n, desc, ub, Sdiag, tiles, off_scale);
  };
  L.cluster_rz = [](const DeviceView& v, hipStream_t st, int nb, double* partial) {
    hipLaunchKernelGGL((clp::cluster_rz_kernel<D>), dim3(nb), dim3(256), 0, st, v, nb, partial);
  };
  L.precond = [](const DeviceView& v, hipStream_t st, int mode) {
    if (v.Nrb) hipLaunchKernelGGL((precond_invert_kernel<D>), dim3(v.Nrb), dim3(64), 0, st, v, mode);
  };
  L.spmv = [](const DeviceView& v, hipStream_t st, const double* ub, const double* x, double* y, int dot) {
    if (!v.Nrb) return;
    if (v.n_spc) hipLaunchKernelGGL((spmv_rows_kernel<D>), dim3((v.n_spc + 3) / 4), dim3(256), 0, st, v, ub, x);
    hipLaunchKernelGGL((spmv_cols_kernel<D>), dim3(v.Nrb), dim3(256), 0, st, v, x, y, dot);
  };
  L.pcg_step = [](const DeviceView& v, hipStream_t st, const double* b, int it, int nb, double eta, int min_it,
                  int max_it, const double* red8, HostMirror* mirror, unsigned long long seq, const int* guard) {
    hipLaunchKernelGGL((pcg_step_kernel<D>), dim3(nb), dim3(kPcgStepThreads), 0, st, v, b, it, nb, eta, min_it,
                       max_it, red8, mirror, seq, guard);
  };
  L.implicit_spmv = [](const DeviceView& v, hipStream_t st, RedLayout R, const double* x, double* y,
                       double* w1, double* w2, double ir, double lo, double hi, int add_diag, int nb, int dot) {
    // work array: w1 = zhat, 4 doubles per track (w2 unused)
    if (!v.Nrb) return;
    if (!SH) {
      hipLaunchKernelGGL((implicit_tracks_q_kernel<D, DP>), dim3(nb), dim3(256), 0, st, v, x, w1);
      hipLaunchKernelGGL((implicit_cameras_q_kernel<D, DP>), dim3(8 * ((v.Nrb + 7) / 8)), dim3(64), 0, st, v, R, x, w1, y, ir, lo, hi,
                         add_diag, dot);
      return;
    }
    hipLaunchKernelGGL((implicit_tracks_sq_kernel<D, DP>), dim3(nb), dim3(256), 0, st, v, x, w1);
    // cam_part is free between two builds of the normal equations: the per-view partial
    // products of the shared intrinsics blocks live in its head
    hipLaunchKernelGGL((implicit_cameras_sq_kernel<D, DP>), dim3(v.Ncam_rb), dim3(64), 0, st, v, R, x, w1, y,
                       ir, lo, hi, add_diag, v.cam_part);
    if (v.Nrb > v.Ncam_rb)
      hipLaunchKernelGGL((implicit_groups_kernel<D>), dim3(v.Nrb - v.Ncam_rb), dim3(64), 0, st, v, R, x, v.cam_part,
                         y, ir, lo, hi, add_diag);
  };
  L.mf_product = [](const DeviceView& v, const mfc::View& m0, hipStream_t st, RedLayout R, const double* x, double* y,
                    double ir, double lo, double hi, int add_diag, int dot, int xs_ready, hipEvent_t ev_a, hipEvent_t ev_b,
                    const int* guard) {
    if (!v.Nrb) return;
    mfc::View m = m0;
    m.guard = guard;
    if (v.drop_pos) {
      // the position columns of the planes are formed from Jp (device_view.h): the product gathers x with the
      // position entries times the views' column scales, the reduce launch scales the position entries of the sums
      const int n = v.Nrb * D;
      if (!xs_ready) hipLaunchKernelGGL((pos_scale_kernel<D>), dim3((n + 255) / 256), dim3(256), 0, st, v, x, v.xs);
      bool cp_done = false;
      if constexpr (D == 9) {
        if (m.n_items && v.compact == 2) {
          hipExtLaunchKernelGGL((mfc::product_kernel<D, DP, true, 2>), dim3(m.n_items), dim3(mfc::kThreads), 0, st, ev_a, nullptr, 0, v,
                                m, (const double*)v.xs);
          cp_done = true;
        } else if (m.n_items && v.compact) {
          hipExtLaunchKernelGGL((mfc::product_kernel<D, DP, true, 1>), dim3(m.n_items), dim3(mfc::kThreads), 0, st, ev_a, nullptr, 0, v,
                                m, (const double*)v.xs);
          cp_done = true;
        }
      }
      if (m.n_items && !cp_done)
        hipExtLaunchKernelGGL((mfc::product_kernel<D, DP, true>), dim3(m.n_items), dim3(mfc::kThreads), 0, st, ev_a, nullptr, 0, v, m,
                              (const double*)v.xs);
    } else if (m.n_items) {
      hipExtLaunchKernelGGL((mfc::product_kernel<D, DP, false>), dim3(m.n_items), dim3(mfc::kThreads), 0, st, ev_a, nullptr, 0, v, m, x);
    }
    hipExtLaunchKernelGGL((mfc::reduce_kernel<D>), dim3(8 * ((v.Nrb + 7) / 8)), dim3(256), 0, st, m.n_items ? nullptr : ev_a, ev_b, 0,
                          v, m, R, x, y, ir, lo, hi, add_diag, dot);
  };
  L.pcg_a = [](const DeviceView& v, hipStream_t st, int n, int it) {
    hipLaunchKernelGGL((pcg_a_kernel<D>), dim3(1), dim3(1024), 0, st, v, n, it);
  };
  L.pcg_b2 = [](const DeviceView& v, hipStream_t st, const double* b, int mode, int nb, double* partial) {
    hipLaunchKernelGGL((pcg_b2_kernel<D>), dim3(nb), dim3(256), 0, st, v, b, mode, nb, partial);
  };
  L.back_substitute = [](const DeviceView& v, hipStream_t st, int nb, double* partial, double* sums) {
    // (drop_pos: the scaled copy of y_c the kernel gathers is update_cameras' -- the engine launches that one first;
    //  the candidate points and their step / norm sums come out of this launch too: no update_points launch)
    hipLaunchKernelGGL((back_substitute_kernel<D, DP, SH>), dim3(nb), dim3(256), 0, st, v, nb, partial, sums);
  };
  L.update_cameras = [](const DeviceView& v, hipStream_t st, double* out, double* prep_c) {
    const int n = v.Nc + (SH ? v.Nrb - v.Ncam_rb : 0);
    hipLaunchKernelGGL((update_cameras_kernel<D, SH>), dim3((std::max(n, 1) + 255) / 256), dim3(256), 0, st, v, out, prep_c);
    // shared intrinsics: a view's record needs its group's candidate, written by another thread
    if (SH && v.Nc)
      hipLaunchKernelGGL(camera_prepare_kernel, dim3((v.Nc + 255) / 256), dim3(256), 0, st, v, v.ext_c, v.intr_c, prep_c);
  };
  L.pcg_init = [](const DeviceView& v, hipStream_t st, const double* b, int nb) {
    hipLaunchKernelGGL((pcg_init_kernel<D>), dim3(nb), dim3(kPcgStepThreads), 0, st, v, b, nb);
  };
  L.pcg_persistent = [](const DeviceView& v, hipStream_t st, int grid, const ppcg::Args& a) {
    hipLaunchKernelGGL((ppcg::pcg_persistent_kernel<D>), dim3(grid), dim3(ppcg::kThreads), 0, st, v, a);
  };
  if constexpr (SH) {
    if (fp32) {
      // fp32 evaluation on a problem with shared intrinsics blocks (BASELINE config 5): the per-observation planes are
      // STORED in fp32 as well (DeviceView::planes_fp32; sums and everything per track / per view stay fp64) -- the
      // kernels that touch the planes, in their float instantiation
      L.linearize = [](const DeviceView& v, hipStream_t st, const double* prep, int lt, double lw, int nb, double* sums, int) {
        hipLaunchKernelGGL((linearize_kernel<D, DP, SH, float, 2, float>), dim3(nb), dim3(256), 0, st, v, prep, lt, lw, nb, sums);
      };
      L.point_scale = [](const DeviceView& v, hipStream_t st, int nb) {
        hipLaunchKernelGGL((point_scale_kernel<DP, float>), dim3(nb), dim3(256), 0, st, v);
      };
      L.point_eliminate = [](const DeviceView& v, hipStream_t st, double ir, double lo, double hi, int nb, double* pm,
                             double* vote, double gtol, double* gvote) {
        hipLaunchKernelGGL((point_eliminate_kernel<D, DP, SH, true, float>), dim3(nb), dim3(256), 0, st, v, ir, lo, hi, nb, pm,
                           vote, gtol, gvote);
      };
      L.implicit_spmv = [](const DeviceView& v, hipStream_t st, RedLayout R, const double* x, double* y, double* w1, double* w2,
                           double ir, double lo, double hi, int add_diag, int nb, int dot) {
        if (!v.Nrb) return;
        hipLaunchKernelGGL((implicit_tracks_sq_kernel<D, DP, float>), dim3(nb), dim3(256), 0, st, v, x, w1);
        hipLaunchKernelGGL((implicit_cameras_sq_kernel<D, DP, float>), dim3(v.Ncam_rb), dim3(64), 0, st, v, R, x, w1, y,
                           ir, lo, hi, add_diag, v.cam_part);
        if (v.Nrb > v.Ncam_rb)
          hipLaunchKernelGGL((implicit_groups_kernel<D>), dim3(v.Nrb - v.Ncam_rb), dim3(64), 0, st, v, R, x, v.cam_part,
                             y, ir, lo, hi, add_diag);
      };
      L.back_substitute = [](const DeviceView& v, hipStream_t st, int nb, double* partial, double* sums) {
        hipLaunchKernelGGL((back_substitute_kernel<D, DP, SH, float>), dim3(nb), dim3(256), 0, st, v, nb, partial, sums);
      };
      // ... and the [A | Q | r~ | r | A1] records (cm_A) are fp32 too: their writer is point_eliminate above, their readers
      L.shared_blocks = [](const DeviceView& v, hipStream_t st, RedLayout R) {
        if (v.Nrb == v.Ncam_rb) return;
        hipLaunchKernelGGL((camera_group_partials_kernel<D, DP, float>), dim3(v.Ncam_rb), dim3(64), 0, st, v);
        hipLaunchKernelGGL((group_reduce_kernel<D>), dim3(v.Nrb - v.Ncam_rb), dim3(256), 0, st, v, R);
      };
      L.camera_diag = [](const DeviceView& v, hipStream_t st, RedLayout R, int max_chunks, double* chunk_partial) {
        const bool chunked = max_chunks > 0 && v.Nrb > v.Ncam_rb;
        if (v.Nrb) hipLaunchKernelGGL((camera_diag_kernel<D, DP, SH, float>), dim3(v.Nrb), dim3(64), 0, st, v, R, chunked ? 1 : 0);
        if (chunked) {
          const int ns = v.Nrb - v.Ncam_rb;
          hipLaunchKernelGGL((shared_diag_partial_kernel<D, DP>), dim3(ns * max_chunks), dim3(64), 0, st, v, max_chunks, chunk_partial);
          hipLaunchKernelGGL((shared_diag_reduce_kernel<D>), dim3(ns), dim3(64), 0, st, v, R, max_chunks, chunk_partial);
        }
      };

    }
  }
  L.tile_gather = [](const DeviceView& v, hipStream_t st, const double* ub, const double* rhs, double* tiles, int n) {
    const long long total = ((long long)v.nub + v.Nrb) * D * D + n;
    if (total)
      hipLaunchKernelGGL((cdf::tile_gather_kernel<D>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                         st, v, ub, rhs, tiles, n);
  };
  L.dense_gather = [](const DeviceView& v, hipStream_t st, const double* ub, double* A, int n) {
    const long long total = ((long long)v.nub + v.Nrb) * D * D;
    if (total)
      hipLaunchKernelGGL((dense_gather_kernel<D>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                         st, v, ub, A, n);
  };
  return L;
}

static bool get_launch(int D, int DP, bool shared, bool fp32, Launch* out) {
#define TMI_CASE(d, p)                                                                 \
  if (D == d && DP == p) {                                                             \
    *out = shared ? make_launch<d, p, true>(fp32) : make_launch<d, p, false>(fp32);   \
    return true;                                                                       \
  }
  TMI_CASE(6, 3) TMI_CASE(6, 4) TMI_CASE(9, 3) TMI_CASE(9, 4)
  TMI_CASE(12, 3) TMI_CASE(12, 4) TMI_CASE(16, 3) TMI_CASE(16, 4)
#undef TMI_CASE
  return false;
}

}  // namespace tmi

using namespace tmi;

// ---- the opaque solver ----------------------------------------------------------
namespace tmi {
static_assert(SC_COUNT == 32 && FL_COUNT == 8, "HostMirror layout (device_view.h)");
// top of an LM iteration: the device flags and the eight all-reduced scalars start at zero (one launch; two fill commands before)
__global__ void iteration_begin_kernel(int* __restrict__ flags, double* __restrict__ sc8) {
  const int t = threadIdx.x;
  if (t < FL_COUNT) flags[t] = 0;
  if (t < 8) sc8[t] = 0.0;
}
__global__ void publish_kernel(const double* __restrict__ scal, const double* __restrict__ red8,
                               const int* __restrict__ flags, HostMirror* m, unsigned long long seq) {
  const int t = threadIdx.x;  // 64 threads
  if (t < SC_COUNT) m->scal[t] = scal[t];
  if (t < 8) m->red[t] = red8[t];
  if (t < FL_COUNT) m->flags[t] = flags[t];
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(&m->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void flag_to_scalar_kernel(const int* __restrict__ flag, double* __restrict__ dst) {
  if (threadIdx.x == 0) *dst = (*flag) ? 1.0 : 0.0;
}
}  // namespace tmi

struct tmi_ba_solver {
  Structure st;
  Launch launch;
  DeviceView v;
  RedLayout RL;
  int DP = 4;
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<void*> allocs;
  // host mirror of the device scalars: pinned, mapped, coherent.  A one-workgroup kernel
  // publishes them and bumps `seq`; the host polls `seq` (no copy commands, no stream sync)
  HostMirror* h_mirror = nullptr;
  HostMirror* d_mirror = nullptr;  // device address of the same memory
  unsigned long long mirror_seq = 0;
  double* h_scal = nullptr;   // = h_mirror->scal
  double* h_red = nullptr;    // = h_mirror->red: the 8 all-reduced scalars at the tail of `red`
  int* h_flags = nullptr;     // = h_mirror->flags
  // initial parameters (for reset) in device order, on the host and resident in HBM
  std::vector<double> ext0, intr0, pts0;
  std::vector<int> grp_off_h;  // [G + 1] offsets of the groups' intrinsics in intr0
  double *d_ext0 = nullptr, *d_intr0 = nullptr, *d_pts0 = nullptr;
  int n_intr = 0;
  // extra device arrays not in the view
  double* d_pm_u = nullptr;
  double* d_cm_t = nullptr;   // implicit Schur operator: t_i per camera-major slot (shared intrinsics blocks only)
  bool need_slot_track = false;
  mfc::View mf = {};          // one-sweep matrix-free product (mf_chunks.h); mf_ok: built and in use
  bool mf_ok = false;
  ddg::Plan dd = {};          // camera side without camera-major records (direct_diag.h); direct_ok: built
  bool direct_ok = false;
  // camera_finish_kernel (direct_diag.h) instead of the separate reduce / finish_diag / precond / pcg_init launches;
  // TMI_BA_FUSED_FINISH=0 keeps the separate launches (tests hold the two to each other)
  bool fuse_finish_ok = true;
  // the product class hands its HIP events to the launches (Timed, attach); TMI_BA_ATTACH_EVENTS=0 records them on the stream
  bool attach_events = true;
  // PCG: the next iteration is enqueued before the host has read the current one's stopping test (solve_reduced_pcg);
  // TMI_BA_PCG_SPECULATE=0 switches it off
  bool pcg_speculate = true;
  bool fast_start_ok = true;  // TMI_BA_FAST_START=0: linearize + point_scale + point_eliminate before the scaled linearize
  bool compact_env = true;    // TMI_BA_COMPACT_PLANES=0: always the full planes (device_view.h, DeviceView::compact)
  bool fuse_sums = true;      // TMI_BA_FUSE_TRACK_SUMS=0: point_eliminate sweeps the planes for V and g_p as before
  bool compact_robust = true; // TMI_BA_COMPACT_ROBUST=0: compact planes for the TRIVIAL loss only
  bool cost_by_view = false;   // ... and the trial cost view by view (every observation owns a slot)
  bool cost_warm = true;       // ... which also reads linearize's observation stream into the Infinity Cache (TMI_BA_COST_WARM=0: off)
  bool implicit = false;      // S is never formed (schur_mode)
  bool adaptive = false;      // schur_mode auto on one rank: both operators are resident and every LM iteration
                              // takes the cheaper one for the PCG length it expects (see solve)
  bool implicit_now = false;  // the operator of the current LM iteration
  bool y_records = false;     // point_eliminate writes Y records (shared blocks, or the older Schur kernels by env)
  int n_implicit_iterations = 0;
  int adaptive_break_even_override = -1;
  int adaptive_break_even = 4;  // PCG iterations up to which the matrix-free operator is the cheaper one
  double cur_inv_radius = 0.0;
  double time_vote = 0.0;     // this rank's "solver time exceeded" vote (source of a small async copy)
  const tmi_ba_options* cur_opts = nullptr;
  double* d_partial_max = nullptr;
  double* d_dense = nullptr;  // n_r x n_r when an exact solve is requested (launch-per-panel path)
  // tile-dataflow Cholesky (dense_cholesky_df.h): tiles, inverse diagonal factors, flags; one epoch per solve
  double* d_df_tiles = nullptr;
  double* d_df_linv = nullptr;
  int* d_df_flags = nullptr;
  int df_epoch = 0;
  int num_cus = 0;
  // CLUSTER_JACOBI over the shared intrinsics blocks (cluster_precond.h)
  clp::Plan cl_plan;
  int shared_diag_chunks = 0;   // > 0: chunks per shared block of the two-step raw diagonal (kernels.h, shared_diag_*)
  double* d_shared_diag_partial = nullptr;
  bool cluster_blocks = false;  // the matrix-free operator with the clusters' blocks of S formed beside it
  bool cl_built = false;      // plan + device buffers exist
  int vis_type = 0;           // visibility_clustering_type the clusters were built with (create)
  bool vis_clusters = false;  // no shared intrinsics blocks: the clusters are Ceres' visibility clusters of the views
  std::vector<std::vector<int> > vis_members;  // ... their reduced blocks, ascending (build_visibility_clusters)
  std::vector<int> vis_cluster_of_rb;          // ... and the cluster of every reduced block (-1: none), singletons included
  // CLUSTER_TRIDIAGONAL (cluster_chains.h): the handle's "clusters" are chains of clusters with the blocks of S between
  // neighbours; built at create, factored with a retry (off-diagonal cells halved) as Ceres does
  bool tri = false;
  chains::Segments tri_seg;
  bool cl_failed = false;  // the last factorisation of a tridiagonal handle failed twice: the linear solve fails
  bool cl_active = false;     // the current LM iteration's PCG applies it
  // persistent PCG on the formed S (pcg_persist.h): buffers made on first use; off after an aborted launch
  bool ppcg_ready = false, ppcg_off = false;
  int ppcg_grid = 0;
  double *d_ppcg_p = nullptr, *d_ppcg_x = nullptr, *d_ppcg_partial = nullptr;
  int* d_ppcg_bar = nullptr;
  long long* d_ppcg_prof = nullptr;  // TMI_BA_PPCG_PROF: per-phase ticks of the persistent PCG launch, printed at destroy
  bool cl_unavailable = false;  // the clusters' tiles could not be allocated: SCHUR_JACOBI for the life of the handle
  bool cl_retired = false;    // a cluster launch gave up in this solve (device shared with another process): SCHUR_JACOBI for the rest of it
  clp::ClusterDesc* d_cl_desc = nullptr;
  clp::GatherEntry* d_cl_ge = nullptr;
  int* d_cl_idx = nullptr;
  double *d_cl_tiles = nullptr, *d_cl_linv = nullptr, *d_cl_vec = nullptr;
  int *d_cl_flags = nullptr, *d_cl_bad = nullptr;
  int cl_epoch = 0;
  int nblocks_slices = 0;     // ceil(nslices / 4): grid of the thread-per-track side kernels
  int nblocks_tracks = 0;     // grid of the per-track kernels of the solve (DeviceView::n_track_blocks)
  int nblocks_points = 0;
  tmi_ba_allreduce_fn allreduce = nullptr;
  void* allreduce_user = nullptr;
  void* nccl_comm = nullptr;  // native RCCL communicator (tmi_ba_solver_init_rccl)
  // profiling
  unsigned prof_mask = 0;
  struct Ev { int cls; hipEvent_t a, b; };
  std::vector<Ev> events;
  size_t ev_used = 0;
  int64_t launches[TMI_BA_NUM_KERNEL_CLASSES] = {0};
  std::string error;
  double setup_seconds = 0.0;
  // inner iterations (inner_kernels.h): built on first use
  struct InnerCtx {
    bool ready = false;
    InnerSet set[2];  // 0 extrinsics blocks, 1 intrinsics blocks
    double* x0[2] = {nullptr, nullptr};
    double* xc[2] = {nullptr, nullptr};
    double* bak_ext = nullptr;
    double* bak_intr = nullptr;
    double* bak_pts = nullptr;
    int* d_active = nullptr;
    int* h_active = nullptr;  // pinned
  } inner;
  // per-track side kernels (outlier filter, batched track adjustment): parameters + SELL
  // layout only when `light`; output arrays allocated on first use
  bool light = false;
  unsigned char* d_trk_flag = nullptr;
  double* d_trk_mean = nullptr;
  signed char* d_trk_term = nullptr;
  int* d_trk_iter = nullptr;
  double* d_trk_c0 = nullptr;
  double* d_trk_c1 = nullptr;
  // device side of the filter / selection bookkeeping (select_kernels.h), built on first use
  int* d_pt_orig = nullptr;               // [Np_pad] caller's track index or -1
  unsigned char* d_out_u8 = nullptr;      // [Np_total] flags / selection in the caller's order
  double* d_out_f64 = nullptr;            // [Np_total]
  int* d_out_i32 = nullptr;               // [Np_total]
  int* d_counters = nullptr;              // [4]
  int* h_counters = nullptr;              // pinned
  unsigned char* h_stage = nullptr;       // pinned staging for the per-track outputs: [u8 | i32 | f64] x Np_total
                                          //   (a copy straight into pageable caller memory makes the runtime
                                          //   pin those pages first: 24 ms for 9 MB, measured)
  int* d_vbox = nullptr;                  // [Nc][4]
  long long* d_cell_off = nullptr;        // [Nc + 1]
  long long* h_cell_total = nullptr;      // pinned
  long long cell_capacity = 0;
  std::vector<void*> cell_allocs;         // the three cell arrays (re-allocated when the grid grows)
  unsigned* d_cell_len = nullptr;
  unsigned long long* d_cell_err = nullptr;
  unsigned* d_cell_trk = nullptr;
  unsigned* d_sel = nullptr;              // [Np_total]
  unsigned long long* d_vt_keys = nullptr;  // [No_pad] (view << 32 | track) sorted; static per handle
  long long* d_vt_ptr = nullptr;          // [Nc + 1]
  unsigned char* d_view_mask = nullptr;   // [Nc]
  int* d_vcount = nullptr;                // [2 Nc]: selected tracks per view after the grid phase | views needing a top-up
  // device-built structure (structure_gpu.h): the big layout arrays exist in HBM only; host copies
  // are fetched on demand (inner iterations, tmi_ba_solver_evaluate)
  bool device_structure = false;
  long long* d_obs_orig = nullptr;        // [No_pad] caller's observation index or -1
  int n_order_dev = 0, n_spc_dev = 0;
};

namespace {


template <class T>
int dev_alloc(tmi_ba_solver* s, T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  TMI_HIP(hipMalloc((void**)p, n * sizeof(T)));
  s->allocs.push_back((void*)*p);
  return TMI_BA_OK;
}
template <class T>
int dev_upload(tmi_ba_solver* s, T** p, const std::vector<T>& h) {
  int rc = dev_alloc(s, p, h.size());
  if (rc) return rc;
  if (!h.empty()) TMI_HIP(hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return TMI_BA_OK;
}

// HIP events around the launches of one kernel class.  attach = true: the events are not recorded here (a record is a
// barrier packet of its own on the stream: ~7 us of idle device per record, 2.7 % of the headline iteration with the two
// records per product) but handed to the launches -- hipExtLaunchKernelGGL takes the start time of the first kernel
// and the end time of the last one from the dispatch packets' own completion signals (start() / stop()).
struct Timed {
  tmi_ba_solver* s;
  int cls;
  bool on;
  bool attached;
  tmi_ba_solver::Ev* ev = nullptr;
  Timed(tmi_ba_solver* s_, int cls_, bool attach = false) : s(s_), cls(cls_), attached(attach) {
    s->launches[cls]++;
    on = (s->prof_mask >> cls) & 1u;
    if (on) {
      if (s->ev_used == s->events.size()) {
        tmi_ba_solver::Ev e;
        e.cls = cls;
        hipEventCreate(&e.a);
        hipEventCreate(&e.b);
        s->events.push_back(e);
      }
      ev = &s->events[s->ev_used++];
      ev->cls = cls;
      if (!attached) hipEventRecord(ev->a, s->stream);
    }
  }
  hipEvent_t start() const { return on && attached ? ev->a : nullptr; }
  hipEvent_t stop() const { return on && attached ? ev->b : nullptr; }
  ~Timed() {
    if (on && !attached) hipEventRecord(ev->b, s->stream);
  }
};

void select_mirror(tmi_ba_solver* s, int slot) {
  s->h_scal = s->h_mirror[slot].scal;
  s->h_red = s->h_mirror[slot].red;
  s->h_flags = s->h_mirror[slot].flags;
}

int readback(tmi_ba_solver* s) {
  select_mirror(s, 0);
  const unsigned long long seq = ++s->mirror_seq;
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, s->stream, s->v.scal,
                     s->v.red + s->RL.scalars, s->v.flags, s->d_mirror, seq);
  volatile unsigned long long* p = &s->h_mirror->seq;
  for (unsigned spin = 1;; ++spin) {
    if (*p == seq) break;
    if ((spin & 0xfffu) == 0) {
      // safety net: once the stream has drained the data is in host memory regardless
      const hipError_t q = hipStreamQuery(s->stream);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) {
        s->error = std::string("hipStreamQuery: ") + hipGetErrorString(q);
        return TMI_BA_ERR_DEVICE;
      }
    }
    __builtin_ia32_pause();
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  // a kernel that could not be launched (bad configuration, out of resources) shows up here
  const hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    s->error = std::string("kernel launch failed: ") + hipGetErrorString(le);
    return TMI_BA_ERR_DEVICE;
  }
  return TMI_BA_OK;
}

// ---- RCCL, resolved at run time (no link-time dependency; inside a PyTorch process the
// already loaded librccl is reused so there is a single RCCL in the process)
struct Rccl {
  struct Id { char b[128]; };  // ncclUniqueId (passed by value to ncclCommInitRank)
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
  std::string error;
};
Rccl& rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r;
  tried = true;
  void* h = nullptr;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!h)
    for (const char* n : names)
      if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) {
    r.error = "librccl not found";
    return r;
  }
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
  if (!r.ok) r.error = "librccl lacks the expected symbols";
  return r;
}

int transport_allreduce(tmi_ba_solver* s, double* buf, int64_t count) {
  Timed t(s, TMI_BA_K_ALLREDUCE);
  if (s->nccl_comm) {
    // ncclDouble = 8, ncclSum = 0 (nccl.h)
    const int rc = rccl().AllReduce(buf, buf, (size_t)count, 8, 0, s->nccl_comm, s->stream);
    if (rc != 0) {
      s->error = std::string("ncclAllReduce failed: ") +
                 (rccl().GetErrorString ? rccl().GetErrorString(rc) : "error");
      return TMI_BA_ERR_COLLECTIVE;
    }
    return TMI_BA_OK;
  }
  if (!s->allreduce) return TMI_BA_OK;
  const int rc = s->allreduce((void*)buf, count, (void*)s->stream, s->allreduce_user);
  if (rc != 0) {
    s->error = "all-reduce callback failed";
    return TMI_BA_ERR_COLLECTIVE;
  }
  return TMI_BA_OK;
}

int do_allreduce(tmi_ba_solver* s, double* buf, int64_t count) {
  if (s->st.world <= 1 || (!s->allreduce && !s->nccl_comm)) return TMI_BA_OK;
  return transport_allreduce(s, buf, count);
}

thread_local std::string g_last_error;  // message of the last failed call on this thread

void set_message(tmi_ba_summary* sum, const char* m) { snprintf(sum->message, sizeof(sum->message), "%s", m); }

}  // namespace

// ---- device-side structure build (structure_gpu.h) -------------------------------------------
namespace {
int bits_for(unsigned long long n) {  // bits needed to hold values 0 .. n - 1 (at least 1)
  int b = 1;
  while (b < 63 && (1ull << b) < n) ++b;
  return b;
}
struct TempPool {
  std::vector<void*> v;
  ~TempPool() {
    for (void* p : v) hipFree(p);
  }
  template <class T>
  hipError_t get(T** p, size_t n) {
    *p = nullptr;
    const hipError_t e = hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T));
    if (e == hipSuccess) v.push_back((void*)*p);
    return e;
  }
};
}  // namespace

static bool device_setup_possible(const Structure& st, int world, int64_t No, bool want_pairs) {
  if (getenv("TMI_BA_HOST_SETUP")) return false;
  // sharded handles: the block set of S is global, which the device pass over the local layout does not
  // see -- they take the device path when S is not formed (the matrix-free operator, the default for world > 1)
  if ((world != 1 && want_pairs) || st.has_shared || st.Nc < 1 || st.Np_total < 1 || No < 1) return false;
  if (No >= (int64_t)1 << 31) return false;
  if (want_pairs) {
    // the block set of S goes through an Nrb x Nrb presence map (13 bytes per entry in flags, scan and
    // ids): fine up to tens of thousands of views on a 288 GB device, 32-bit scans bound it at 46 340
    const int64_t n2 = (int64_t)st.Nrb * st.Nrb;
    if (n2 >= 2000000000LL) return false;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (double)free_b < 16.0 * (double)n2 + 4e9) return false;
  }
  if (bits_for((unsigned)st.Nc) + bits_for((unsigned)st.Np_total) > 62) return false;
  return true;
}

// s->st holds what build_blocks left (camera side); fills the rest of s->st that the host needs and
// the static-structure pointers of s->v.  Orders match structure.cpp element for element.
static int build_structure_device(tmi_ba_solver* s, const tmi_ba_problem* P, bool want_pairs) {
  using namespace tmi::sg;
  Structure& st = s->st;
  DeviceView& v = s->v;
  hipStream_t stream = s->stream;
  const bool timing = getenv("TMI_BA_SETUP_TIMING") != nullptr;
  double t_phase = now_s();
  auto lap = [&](const char* what) {
    if (!timing) return;
    hipStreamSynchronize(stream);
    const double now = now_s();
    fprintf(stderr, "[tmi_ba setup/device] %-24s %.3f s\n", what, now - t_phase);
    t_phase = now;
  };
  const int64_t No = P->num_observations;
  const int Np = st.Np_total, Nc = st.Nc, Nrb = st.Nrb;
  TempPool tmp;
  void* cub_tmp = nullptr;
  size_t cub_cap = 0;
  auto cub_reserve = [&](size_t bytes) -> int {
    if (bytes <= cub_cap) return TMI_BA_OK;
    if (cub_tmp) hipFree(cub_tmp);
    cub_tmp = nullptr;
    cub_cap = 0;
    TMI_HIP(hipMalloc(&cub_tmp, bytes + 256));
    cub_cap = bytes + 256;
    return TMI_BA_OK;
  };
  struct CubFree {
    void** p;
    ~CubFree() {
      if (*p) hipFree(*p);
    }
  } cub_free{&cub_tmp};
  int rc;
#define SG_SORT_PAIRS(kin, kout, vin, vout, n, bits)                                                        \
  do {                                                                                                      \
    size_t bytes_ = 0;                                                                                      \
    TMI_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes_, kin, kout, vin, vout, (int)(n), 0, bits, stream)); \
    if ((rc = cub_reserve(bytes_))) return rc;                                                              \
    TMI_HIP(hipcub::DeviceRadixSort::SortPairs(cub_tmp, bytes_, kin, kout, vin, vout, (int)(n), 0, bits, stream)); \
  } while (0)
#define SG_EXCLUSIVE_SUM(in, out, n)                                                                        \
  do {                                                                                                      \
    size_t bytes_ = 0;                                                                                      \
    TMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes_, in, out, (int)(n), stream));                  \
    if ((rc = cub_reserve(bytes_))) return rc;                                                              \
    TMI_HIP(hipcub::DeviceScan::ExclusiveSum(cub_tmp, bytes_, in, out, (int)(n), stream));                  \
  } while (0)
  auto nb = [](long long n) { return dim3((unsigned)std::max<long long>((n + 255) / 256, 1)); };  // kernels bound-check

  // ---- observations to the device, track lengths, argument check
  int *d_ocam, *d_opt, *d_klen, *d_bad;
  double* d_oxy;
  TMI_HIP(tmp.get(&d_ocam, (size_t)No));
  TMI_HIP(tmp.get(&d_opt, (size_t)No));
  TMI_HIP(tmp.get(&d_oxy, (size_t)2 * No));
  TMI_HIP(tmp.get(&d_klen, (size_t)Np + 1));
  TMI_HIP(tmp.get(&d_bad, 1));
  TMI_HIP(hipMemcpyAsync(d_ocam, P->obs_camera, (size_t)No * sizeof(int), hipMemcpyHostToDevice, stream));
  TMI_HIP(hipMemcpyAsync(d_opt, P->obs_point, (size_t)No * sizeof(int), hipMemcpyHostToDevice, stream));
  TMI_HIP(hipMemcpyAsync(d_oxy, P->obs_xy, (size_t)2 * No * sizeof(double), hipMemcpyHostToDevice, stream));
  TMI_HIP(hipMemsetAsync(d_klen, 0, ((size_t)Np + 1) * sizeof(int), stream));
  TMI_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), stream));
  hipLaunchKernelGGL(hist_kernel, nb(No), dim3(256), 0, stream, d_ocam, d_opt, (long long)No, Nc, Np, d_klen, d_bad);
  lap("upload + histogram");

  // ---- observations sorted by (track, camera); duplicates
  const int cam_bits = bits_for((unsigned)Nc), pt_bits = bits_for((unsigned)Np);
  unsigned long long *d_ok_in, *d_ok_out;
  unsigned *d_ov_in, *d_tobs;
  TMI_HIP(tmp.get(&d_ok_in, (size_t)No));
  TMI_HIP(tmp.get(&d_ok_out, (size_t)No));
  TMI_HIP(tmp.get(&d_ov_in, (size_t)No));
  TMI_HIP(tmp.get(&d_tobs, (size_t)No));
  hipLaunchKernelGGL(obs_keys_kernel, nb(No), dim3(256), 0, stream, d_ocam, d_opt, (long long)No, cam_bits, d_ok_in, d_ov_in);
  SG_SORT_PAIRS(d_ok_in, d_ok_out, d_ov_in, d_tobs, No, cam_bits + pt_bits);
  hipLaunchKernelGGL(dup_check_kernel, nb(No), dim3(256), 0, stream, d_ok_out, (long long)No, d_bad);
  // tptr = exclusive scan of the lengths (64-bit)
  long long *d_klen64, *d_tptr;
  TMI_HIP(tmp.get(&d_klen64, (size_t)Np + 1));
  TMI_HIP(tmp.get(&d_tptr, (size_t)Np + 1));
  hipLaunchKernelGGL(widen_kernel, nb(Np + 1), dim3(256), 0, stream, d_klen, Np + 1, d_klen64);
  SG_EXCLUSIVE_SUM(d_klen64, d_tptr, Np + 1);
  // ---- tracks by descending length (stable)
  unsigned long long *d_tk_in, *d_tk_out;
  int *d_tv_in, *d_order;
  TMI_HIP(tmp.get(&d_tk_in, (size_t)Np));
  TMI_HIP(tmp.get(&d_tk_out, (size_t)Np));
  TMI_HIP(tmp.get(&d_tv_in, (size_t)Np));
  TMI_HIP(tmp.get(&d_order, (size_t)Np));
  hipLaunchKernelGGL(track_keys_kernel, nb(Np), dim3(256), 0, stream, d_klen, Np, d_tptr, d_ok_out, cam_bits,
                     d_tk_in, d_tv_in);
  SG_SORT_PAIRS(d_tk_in, d_tk_out, d_tv_in, d_order, Np, 32 + cam_bits);
  std::vector<int> order((size_t)Np), klen((size_t)Np);
  int bad = 0;
  TMI_HIP(hipMemcpyAsync(order.data(), d_order, (size_t)Np * sizeof(int), hipMemcpyDeviceToHost, stream));
  TMI_HIP(hipMemcpyAsync(klen.data(), d_klen, (size_t)Np * sizeof(int), hipMemcpyDeviceToHost, stream));
  TMI_HIP(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, stream));
  TMI_HIP(hipStreamSynchronize(stream));
  lap("sorts (observations, tracks)");
  if (bad == 1) {
    s->error = "observation index out of range";
    return TMI_BA_ERR_INVALID_ARGUMENT;
  }
  if (bad == 2) {
    s->error = "a track is observed twice by the same view";
    return TMI_BA_ERR_INVALID_ARGUMENT;
  }
  // ---- host: slices (O(#tracks)); a sharded handle keeps the slices the longest-processing-time-first
  // deal of structure.cpp gives its rank (same rule, same result)
  int n_active = 0;
  while (n_active < Np && klen[order[n_active]] > 0) ++n_active;
  st.unobserved.clear();
  if (st.rank == 0) st.unobserved.assign(order.begin() + n_active, order.end());  // ascending (stable sort)
  const int gslices = (n_active + 63) / 64;
  std::vector<int> local_slices;
  if (st.world > 1) {
    std::vector<double> load((size_t)st.world, 0.0);
    for (int gs = 0; gs < gslices; ++gs) {
      double w = 0.0;
      for (int t = 0; t < 64; ++t) {
        const int idx = gs * 64 + t;
        if (idx < n_active) {
          const double k = klen[order[idx]];
          w += want_pairs ? 0.5 * k * (k - 1.0) + 5.0 * k : k + 0.5;  // (structure.cpp: observations when S is not formed)
        }
      }
      int best = 0;
      for (int r = 1; r < st.world; ++r)
        if (load[r] < load[best]) best = r;
      if (best == st.rank) local_slices.push_back(gs);
      load[best] += w;
    }
  } else {
    local_slices.resize((size_t)gslices);
    for (int gs = 0; gs < gslices; ++gs) local_slices[gs] = gs;
  }
  st.nslices = (int)local_slices.size();
  st.Np_pad = st.nslices * 64;
  st.Np = 0;
  int64_t No_loc = 0;
  st.pt_orig.assign(st.Np_pad, -1);
  st.pt_k.assign(st.Np_pad, 0);
  st.pt_const.assign(st.Np_pad, 0);
  st.slice_ptr.assign(st.nslices + 1, 0);
  for (int sl = 0; sl < st.nslices; ++sl)
    for (int t = 0; t < 64; ++t) {
      const int idx = local_slices[sl] * 64 + t;
      if (idx >= n_active) continue;
      const int p = order[idx], lp = sl * 64 + t;
      st.pt_orig[lp] = p;
      st.pt_k[lp] = klen[p];
      st.pt_const[lp] = (P->point_constant && P->point_constant[p]) ? 1 : 0;
      st.Np++;
      No_loc += klen[p];
    }
  for (int sl = 0; sl < st.nslices; ++sl) {
    const int K = st.pt_k[sl * 64];  // sorted by descending length: the slice's first track is its longest
    const int64_t next = (int64_t)st.slice_ptr[sl] + (int64_t)K * 64;
    if (next > 0x7fffffff) {
      s->error = "too many observations on one rank for 32-bit slot indices";
      return TMI_BA_ERR_UNSUPPORTED;
    }
    st.slice_ptr[sl + 1] = (int)next;
  }
  st.n_wide = 0;
  {
    int wide_k = st.nslices >= 5000 ? kWideKLarge : kWideK;
    if (const char* e = getenv("TMI_BA_WIDE_K")) wide_k = std::max(1, atoi(e));
    while (st.n_wide < st.nslices && ((st.slice_ptr[st.n_wide + 1] - st.slice_ptr[st.n_wide]) >> 6) >= wide_k) ++st.n_wide;
    int ultra_k = kUltraK;
    if (const char* e = getenv("TMI_BA_ULTRA_K")) ultra_k = std::max(1, atoi(e));
    st.n_ultra = 0;
    while (st.n_ultra < st.n_wide && ((st.slice_ptr[st.n_ultra + 1] - st.slice_ptr[st.n_ultra]) >> 6) >= ultra_k) ++st.n_ultra;
  }
  st.No_pad = st.slice_ptr[st.nslices];
  st.No = No_loc;
  const int64_t Npad = st.No_pad;
  {
    int* pi;
    unsigned char* pc;
    if ((rc = dev_upload(s, &pi, st.slice_ptr))) return rc;
    v.slice_ptr = pi;
    if ((rc = dev_upload(s, &pi, st.pt_orig))) return rc;
    s->d_pt_orig = pi;
    if ((rc = dev_upload(s, &pi, st.pt_k))) return rc;
    v.pt_k = pi;
    if ((rc = dev_upload(s, &pc, st.pt_const))) return rc;
    v.pt_const = pc;
  }
  // ---- layout + slots
  int *d_obs_cam, *d_obs_cpos, *d_cam_rb, *d_cam_ptr;
  double* d_obs_xy;
  if ((rc = dev_alloc(s, &d_obs_cam, (size_t)Npad))) return rc;
  if ((rc = dev_alloc(s, &d_obs_xy, (size_t)2 * Npad))) return rc;
  if ((rc = dev_alloc(s, &d_obs_cpos, (size_t)Npad))) return rc;
  if ((rc = dev_alloc(s, &s->d_obs_orig, (size_t)Npad))) return rc;
  if ((rc = dev_alloc(s, &d_cam_ptr, (size_t)Nrb + 2))) return rc;
  TMI_HIP(tmp.get(&d_cam_rb, (size_t)Nc));
  TMI_HIP(hipMemcpyAsync(d_cam_rb, st.cam_rb.data(), (size_t)Nc * sizeof(int), hipMemcpyHostToDevice, stream));
  TMI_HIP(hipMemsetAsync(d_obs_cam, 0xff, (size_t)std::max<int64_t>(Npad, 1) * sizeof(int), stream));
  TMI_HIP(hipMemsetAsync(d_obs_xy, 0, (size_t)std::max<int64_t>(2 * Npad, 1) * sizeof(double), stream));
  TMI_HIP(hipMemsetAsync(d_obs_cpos, 0xff, (size_t)std::max<int64_t>(Npad, 1) * sizeof(int), stream));
  TMI_HIP(hipMemsetAsync(s->d_obs_orig, 0xff, (size_t)std::max<int64_t>(Npad, 1) * sizeof(long long), stream));
  long long *d_ptk64, *d_tstart;
  TMI_HIP(tmp.get(&d_ptk64, (size_t)st.Np_pad + 1));
  TMI_HIP(tmp.get(&d_tstart, (size_t)st.Np_pad + 1));
  TMI_HIP(hipMemsetAsync(d_ptk64, 0, ((size_t)st.Np_pad + 1) * sizeof(long long), stream));
  hipLaunchKernelGGL(widen_kernel, nb(st.Np_pad), dim3(256), 0, stream, v.pt_k, st.Np_pad, d_ptk64);
  SG_EXCLUSIVE_SUM(d_ptk64, d_tstart, st.Np_pad + 1);
  unsigned *d_sk_in, *d_sk_out, *d_se_in, *d_se_out;
  TMI_HIP(tmp.get(&d_sk_in, (size_t)No));
  TMI_HIP(tmp.get(&d_sk_out, (size_t)No));
  TMI_HIP(tmp.get(&d_se_in, (size_t)No));
  TMI_HIP(tmp.get(&d_se_out, (size_t)No));
  hipLaunchKernelGGL(layout_kernel, nb(st.Np_pad), dim3(256), 0, stream, st.Np_pad, s->d_pt_orig, v.pt_k, v.slice_ptr,
                     d_tptr, d_tobs, d_ocam, d_oxy, d_cam_rb, Nrb, d_tstart, d_obs_cam, d_obs_xy, s->d_obs_orig,
                     d_sk_in, d_se_in);
  SG_SORT_PAIRS(d_sk_in, d_sk_out, d_se_in, d_se_out, No_loc, bits_for((unsigned)Nrb + 1));
  hipLaunchKernelGGL(slot_assign_kernel, nb(No_loc), dim3(256), 0, stream, d_sk_out, d_se_out, (long long)No_loc, Nrb, d_obs_cpos);
  hipLaunchKernelGGL((lower_bound_kernel<unsigned, int>), nb(Nrb + 1), dim3(256), 0, stream, d_sk_out, (long long)No_loc,
                     (long long)Nrb, d_cam_ptr);
  int nslots = 0;
  TMI_HIP(hipMemcpyAsync(&nslots, d_cam_ptr + Nrb, sizeof(int), hipMemcpyDeviceToHost, stream));
  TMI_HIP(hipStreamSynchronize(stream));
  st.Nslots = nslots;
  v.obs_cam = d_obs_cam;
  v.obs_xy = d_obs_xy;
  v.obs_cpos = d_obs_cpos;
  v.cam_ptr = d_cam_ptr;
  lap("layout + slots");

  st.nub = 0;
  st.npairs = 0;
  st.nnzb = st.Nrb;
  s->n_order_dev = 0;
  s->n_spc_dev = 0;
  int *d_urow_ptr, *d_ucol_ptr, *d_spc_rptr;
  if ((rc = dev_alloc(s, &d_urow_ptr, (size_t)Nrb + 2))) return rc;
  if ((rc = dev_alloc(s, &d_ucol_ptr, (size_t)Nrb + 2))) return rc;
  if ((rc = dev_alloc(s, &d_spc_rptr, (size_t)Nrb + 2))) return rc;
  TMI_HIP(hipMemsetAsync(d_urow_ptr, 0, ((size_t)Nrb + 2) * sizeof(int), stream));
  TMI_HIP(hipMemsetAsync(d_ucol_ptr, 0, ((size_t)Nrb + 2) * sizeof(int), stream));
  TMI_HIP(hipMemsetAsync(d_spc_rptr, 0, ((size_t)Nrb + 2) * sizeof(int), stream));
  v.urow_ptr = d_urow_ptr;
  v.ucol_ptr = d_ucol_ptr;
  v.spc_rptr = d_spc_rptr;
  {
    // placeholders; replaced below when the pair structure is built
    int* pi;
    long long* pl;
    if ((rc = dev_alloc(s, &pi, 1))) return rc;
    v.ub_i = v.ub_j = v.ucol_u = v.spc_row = v.spc_u0 = v.pair_i = v.pair_j = v.ub_order = pi;
    if ((rc = dev_alloc(s, &pl, 2))) return rc;
    TMI_HIP(hipMemsetAsync(pl, 0, 2 * sizeof(long long), stream));
    v.pair_ptr = pl;
  }
  if (!want_pairs || Nrb == 0) return TMI_BA_OK;

  // ---- block set of S: dense presence map -> sorted upper list
  const long long n2 = (long long)Nrb * Nrb;
  unsigned char* d_present;
  int *d_pres_i, *d_pos, *d_blk_id;
  TMI_HIP(tmp.get(&d_present, (size_t)n2));
  TMI_HIP(tmp.get(&d_pres_i, (size_t)n2 + 1));
  TMI_HIP(tmp.get(&d_pos, (size_t)n2 + 1));
  TMI_HIP(tmp.get(&d_blk_id, (size_t)n2));
  TMI_HIP(hipMemsetAsync(d_present, 0, (size_t)n2, stream));
  TMI_HIP(hipMemsetAsync(d_pres_i, 0, ((size_t)n2 + 1) * sizeof(int), stream));
  int n_long = 0;  // tracks are sorted by descending length
  while (n_long < st.Np_pad && st.pt_k[n_long] >= kLongK) ++n_long;
  if (n_long > 0)
    hipLaunchKernelGGL(block_flags_kernel<64>, dim3((n_long + 3) / 4), dim3(256), 0, stream, 0, n_long, v.pt_k, v.pt_const,
                       v.slice_ptr, d_obs_cam, d_cam_rb, Nrb, d_present);
  if (st.Np_pad > n_long)
    hipLaunchKernelGGL(block_flags_kernel<1>, nb(st.Np_pad - n_long), dim3(256), 0, stream, n_long, st.Np_pad, v.pt_k,
                       v.pt_const, v.slice_ptr, d_obs_cam, d_cam_rb, Nrb, d_present);
  hipLaunchKernelGGL(flags_to_int_kernel, nb(n2), dim3(256), 0, stream, d_present, n2, d_pres_i);
  SG_EXCLUSIVE_SUM(d_pres_i, d_pos, n2 + 1);
  int nub = 0;
  TMI_HIP(hipMemcpyAsync(&nub, d_pos + n2, sizeof(int), hipMemcpyDeviceToHost, stream));
  TMI_HIP(hipStreamSynchronize(stream));
  st.nub = nub;
  st.nnzb = 2 * (int64_t)nub + Nrb;
  int *d_ub_i, *d_ub_j;
  if ((rc = dev_alloc(s, &d_ub_i, (size_t)nub))) return rc;
  if ((rc = dev_alloc(s, &d_ub_j, (size_t)nub))) return rc;
  hipLaunchKernelGGL(block_list_kernel, nb(n2), dim3(256), 0, stream, d_present, d_pos, Nrb, d_ub_i, d_ub_j, d_blk_id);
  v.ub_i = d_ub_i;
  v.ub_j = d_ub_j;
  lap("block set of S");

  // ---- pair lists
  long long *d_pcnt, *d_poff;
  TMI_HIP(tmp.get(&d_pcnt, (size_t)st.Np_pad + 1));
  TMI_HIP(tmp.get(&d_poff, (size_t)st.Np_pad + 1));
  TMI_HIP(hipMemsetAsync(d_pcnt, 0, ((size_t)st.Np_pad + 1) * sizeof(long long), stream));
  hipLaunchKernelGGL(pair_count_kernel, nb(st.Np_pad), dim3(256), 0, stream, st.Np_pad, v.pt_k, v.pt_const, v.slice_ptr,
                     d_obs_cpos, d_pcnt);
  SG_EXCLUSIVE_SUM(d_pcnt, d_poff, st.Np_pad + 1);
  long long npairs = 0;
  TMI_HIP(hipMemcpyAsync(&npairs, d_poff + st.Np_pad, sizeof(long long), hipMemcpyDeviceToHost, stream));
  TMI_HIP(hipStreamSynchronize(stream));
  if (npairs >= ((long long)1 << 31)) {
    s->error = "too many observation pairs for the device-side structure build";
    return TMI_BA_ERR_UNSUPPORTED;
  }
  st.npairs = npairs;
  long long* d_pair_ptr;
  int *d_pair_i, *d_pair_j;
  if ((rc = dev_alloc(s, &d_pair_ptr, (size_t)nub + 2))) return rc;
  if ((rc = dev_alloc(s, &d_pair_i, (size_t)npairs))) return rc;
  if ((rc = dev_alloc(s, &d_pair_j, (size_t)npairs))) return rc;
  {
    TempPool ptmp;  // the pair staging arrays are the largest temporaries: release them early
    unsigned *d_uk_in, *d_uk_out;
    unsigned long long *d_pv_in, *d_pv_out;
    TMI_HIP(ptmp.get(&d_uk_in, (size_t)npairs));
    TMI_HIP(ptmp.get(&d_uk_out, (size_t)npairs));
    TMI_HIP(ptmp.get(&d_pv_in, (size_t)npairs));
    TMI_HIP(ptmp.get(&d_pv_out, (size_t)npairs));
    if (n_long > 0)
    {
      // tracks are sorted by descending length: [0, n_huge) do not fit the LDS-staged kernel
      int n_huge = 0;
      while (n_huge < n_long && st.pt_k[n_huge] > kEmitCap) ++n_huge;
      if (n_huge > 0)
        hipLaunchKernelGGL(pair_emit_kernel<64>, dim3((n_huge + 3) / 4), dim3(256), 0, stream, 0, n_huge, v.pt_k,
                           v.pt_const, v.slice_ptr, d_obs_cam, d_cam_rb, d_obs_cpos, d_blk_id, Nrb, d_poff, d_uk_in, d_pv_in);
      if (n_long > n_huge)
        hipLaunchKernelGGL(pair_emit_long_kernel, dim3((n_long - n_huge + 3) / 4), dim3(256), 0, stream, n_huge, n_long,
                           v.pt_k, v.pt_const, v.slice_ptr, d_obs_cam, d_cam_rb, d_obs_cpos, d_blk_id, Nrb, d_poff, d_uk_in,
                           d_pv_in);
    }
    if (st.Np_pad > n_long)
      hipLaunchKernelGGL(pair_emit_kernel<1>, nb(st.Np_pad - n_long), dim3(256), 0, stream, n_long, st.Np_pad, v.pt_k,
                         v.pt_const, v.slice_ptr, d_obs_cam, d_cam_rb, d_obs_cpos, d_blk_id, Nrb, d_poff, d_uk_in, d_pv_in);
    if (npairs > 0) {
      SG_SORT_PAIRS(d_uk_in, d_uk_out, d_pv_in, d_pv_out, npairs, bits_for((unsigned)std::max(nub, 1)));
      hipLaunchKernelGGL(split_pairs_kernel, nb(npairs), dim3(256), 0, stream, d_pv_out, npairs, d_pair_i, d_pair_j);
    }
    hipLaunchKernelGGL((lower_bound_kernel<unsigned, long long>), nb(nub + 1), dim3(256), 0, stream, d_uk_out, npairs,
                       (long long)nub, d_pair_ptr);
    TMI_HIP(hipStreamSynchronize(stream));
  }
  v.pair_ptr = d_pair_ptr;
  v.pair_i = d_pair_i;
  v.pair_j = d_pair_j;
  lap("pair lists");

  // ---- row / column indices of the symmetric storage, SpMV chunks, launch order
  hipLaunchKernelGGL((lower_bound_kernel<int, int>), nb(Nrb + 1), dim3(256), 0, stream, d_ub_i, (long long)nub,
                     (long long)Nrb, d_urow_ptr);
  {
    unsigned *d_k_in, *d_k_out;
    int *d_iota, *d_ucol_u;
    TMI_HIP(tmp.get(&d_k_in, (size_t)nub));
    TMI_HIP(tmp.get(&d_k_out, (size_t)nub));
    TMI_HIP(tmp.get(&d_iota, (size_t)nub));
    if ((rc = dev_alloc(s, &d_ucol_u, (size_t)nub))) return rc;
    hipLaunchKernelGGL(iota_kernel, nb(nub), dim3(256), 0, stream, d_iota, nub);
    if (nub > 0) {
      TMI_HIP(hipMemcpyAsync(d_k_in, d_ub_j, (size_t)nub * sizeof(int), hipMemcpyDeviceToDevice, stream));
      SG_SORT_PAIRS(d_k_in, d_k_out, d_iota, d_ucol_u, nub, bits_for((unsigned)Nrb));
    }
    hipLaunchKernelGGL((lower_bound_kernel<unsigned, int>), nb(Nrb + 1), dim3(256), 0, stream, d_k_out, (long long)nub,
                       (long long)Nrb, d_ucol_ptr);
    v.ucol_u = d_ucol_u;
    // chunks
    const int chunk = kSpmvTrips * (64 / std::max(st.D, 1));
    int* d_cnt;
    TMI_HIP(tmp.get(&d_cnt, (size_t)Nrb + 1));
    TMI_HIP(hipMemsetAsync(d_cnt, 0, ((size_t)Nrb + 1) * sizeof(int), stream));
    hipLaunchKernelGGL(spc_count_kernel, nb(Nrb), dim3(256), 0, stream, d_urow_ptr, Nrb, chunk, d_cnt);
    SG_EXCLUSIVE_SUM(d_cnt, d_spc_rptr, Nrb + 1);
    int n_spc = 0;
    TMI_HIP(hipMemcpyAsync(&n_spc, d_spc_rptr + Nrb, sizeof(int), hipMemcpyDeviceToHost, stream));
    TMI_HIP(hipStreamSynchronize(stream));
    int *d_spc_row, *d_spc_u0;
    if ((rc = dev_alloc(s, &d_spc_row, (size_t)n_spc))) return rc;
    if ((rc = dev_alloc(s, &d_spc_u0, (size_t)n_spc))) return rc;
    hipLaunchKernelGGL(spc_fill_kernel, nb(Nrb), dim3(256), 0, stream, d_urow_ptr, d_spc_rptr, Nrb, chunk, d_spc_row, d_spc_u0);
    v.spc_row = d_spc_row;
    v.spc_u0 = d_spc_u0;
    s->n_spc_dev = n_spc;
    // launch order: (row & 7, row, pair count descending, block index)
    unsigned *d_k1_in, *d_k1_out, *d_k2_in, *d_k2_out;
    int *d_u1, *d_u2;
    long long* d_qstart;
    TMI_HIP(tmp.get(&d_k1_in, (size_t)nub));
    TMI_HIP(tmp.get(&d_k1_out, (size_t)nub));
    TMI_HIP(tmp.get(&d_k2_in, (size_t)nub));
    TMI_HIP(tmp.get(&d_k2_out, (size_t)nub));
    TMI_HIP(tmp.get(&d_u1, (size_t)nub));
    TMI_HIP(tmp.get(&d_u2, (size_t)nub));
    TMI_HIP(tmp.get(&d_qstart, 16));
    if (nub > 0) {
      hipLaunchKernelGGL(order_key1_kernel, nb(nub), dim3(256), 0, stream, d_pair_ptr, nub, d_k1_in);
      SG_SORT_PAIRS(d_k1_in, d_k1_out, d_iota, d_u1, nub, 32);
      hipLaunchKernelGGL(order_key2_kernel, nb(nub), dim3(256), 0, stream, d_u1, d_ub_i, nub, d_k2_in);
      SG_SORT_PAIRS(d_k2_in, d_k2_out, d_u1, d_u2, nub, 27);
    }
    hipLaunchKernelGGL(queue_start_kernel, dim3(1), dim3(64), 0, stream, d_k2_out, (long long)nub, d_qstart);
    long long qstart[9];
    TMI_HIP(hipMemcpyAsync(qstart, d_qstart, sizeof(qstart), hipMemcpyDeviceToHost, stream));
    TMI_HIP(hipStreamSynchronize(stream));
    long long longest = 0;
    for (int x = 0; x < 8; ++x) longest = std::max(longest, qstart[x + 1] - qstart[x]);
    const long long groups = (longest + 15) / 16;
    const long long n_order = groups * 8 * 16;
    int4* d_hdr;
    if ((rc = dev_alloc(s, &d_hdr, (size_t)std::max<long long>(n_order, 1)))) return rc;
    TMI_HIP(hipMemsetAsync(d_hdr, 0xff, (size_t)std::max<long long>(n_order, 1) * sizeof(int4), stream));
    if (nub > 0)
      hipLaunchKernelGGL(order_fill_kernel, nb(nub), dim3(256), 0, stream, d_u2, d_k2_out, nub, d_qstart, d_pair_ptr, d_hdr);
    v.ub_order = reinterpret_cast<const int*>(d_hdr);
    s->n_order_dev = (int)n_order;
  }
  TMI_HIP(hipStreamSynchronize(stream));
  lap("indices, chunks, launch order");
#undef SG_SORT_PAIRS
#undef SG_EXCLUSIVE_SUM
  return TMI_BA_OK;
}

// host copies of the layout arrays of a device-built structure, for the few host consumers
static int materialize_host_layout(tmi_ba_solver* s) {
  Structure& st = s->st;
  if (!s->device_structure || !st.obs_cam.empty() || st.No_pad == 0) return TMI_BA_OK;
  st.obs_cam.resize((size_t)st.No_pad);
  st.obs_cpos.resize((size_t)st.No_pad);
  st.obs_orig.resize((size_t)st.No_pad);
  st.obs_xy.resize((size_t)2 * st.No_pad);
  std::vector<long long> oo((size_t)st.No_pad);
  TMI_HIP(hipMemcpy(st.obs_cam.data(), s->v.obs_cam, (size_t)st.No_pad * sizeof(int), hipMemcpyDeviceToHost));
  TMI_HIP(hipMemcpy(st.obs_cpos.data(), s->v.obs_cpos, (size_t)st.No_pad * sizeof(int), hipMemcpyDeviceToHost));
  TMI_HIP(hipMemcpy(st.obs_xy.data(), s->v.obs_xy, (size_t)2 * st.No_pad * sizeof(double), hipMemcpyDeviceToHost));
  TMI_HIP(hipMemcpy(oo.data(), s->d_obs_orig, (size_t)st.No_pad * sizeof(long long), hipMemcpyDeviceToHost));
  for (int64_t e = 0; e < st.No_pad; ++e) st.obs_orig[e] = oo[e];
  return TMI_BA_OK;
}

// ---- C ABI -------------------------------------------------------------------------
extern "C" {

int32_t tmi_ba_version(void) { return TMI_BA_VERSION_MAJOR * 1000 + TMI_BA_VERSION_MINOR; }

const char* tmi_ba_last_error(void) { return g_last_error.c_str(); }

int32_t tmi_ba_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return -1;
  return n;
}

const char* tmi_ba_status_string(int32_t st) {
  switch (st) {
    case TMI_BA_OK: return "ok";
    case TMI_BA_ERR_INVALID_ARGUMENT: return "invalid argument";
    case TMI_BA_ERR_NO_DEVICE: return "no HIP device";
    case TMI_BA_ERR_DEVICE: return "HIP runtime error";
    case TMI_BA_ERR_OUT_OF_MEMORY: return "out of device memory";
    case TMI_BA_ERR_UNSUPPORTED: return "unsupported problem shape";
    case TMI_BA_ERR_EVALUATION_FAILED: return "residual evaluation failed at the start point";
    case TMI_BA_ERR_LINEAR_SOLVER: return "linear solver failure";
    case TMI_BA_ERR_COLLECTIVE: return "collective failure";
    default: return "unknown status";
  }
}

void tmi_ba_options_init(tmi_ba_options* o) {
  if (!o) return;
  memset(o, 0, sizeof(*o));
  // BundleAdjustmentOptions, bundle_adjustment.h:78-122
  o->loss_function_type = TMI_BA_LOSS_TRIVIAL;
  o->robust_loss_width = 2.0;
  o->linear_solver_type = TMI_BA_SPARSE_SCHUR;
  o->preconditioner_type = TMI_BA_PRECOND_SCHUR_JACOBI;
  o->verbose = 0;
  o->num_threads = 1;
  o->max_num_iterations = 100;
  o->max_solver_time_in_seconds = 3600.0;
  o->use_inner_iterations = 1;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->max_trust_region_radius = 1e12;
  // Ceres defaults Theia does not override
  o->initial_trust_region_radius = 1e4;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->eta = 0.1;
  o->max_linear_solver_iterations = 500;
  o->min_linear_solver_iterations = 0;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->point_dof = 4;
  o->device = -1;
  o->profile_kernels = 0;
  o->residual_precision = 64;
  o->schur_mode = 0;
  o->visibility_clustering_type = 0;
  o->iteration_trace = nullptr;
  o->iteration_trace_capacity = 0;
}

int32_t tmi_ba_intrinsics_size(int32_t model) {
  static const int n[5] = {7, 10, 9, 5, 5};
  return (model >= 0 && model < 5) ? n[model] : -1;
}

// GetSubsetFromOptimizeIntrinsicsType: pinhole_camera_model.cc:132-162,
// pinhole_radial_tangential_camera_model.cc:150-185, fisheye_camera_model.cc:142-175,
// fov_camera_model.cc:124-149, division_undistortion_camera_model.cc:126-150.
int32_t tmi_ba_intrinsics_constant_mask(int32_t model, int32_t bits, uint8_t* mask) {
  const int n = tmi_ba_intrinsics_size(model);
  if (n < 0 || !mask) return -1;
  memset(mask, 0, (size_t)n);
  if (bits == TMI_BA_INTRINSICS_ALL) return n;
  const uint8_t cf = !(bits & TMI_BA_INTRINSICS_FOCAL_LENGTH);
  const uint8_t ca = !(bits & TMI_BA_INTRINSICS_ASPECT_RATIO);
  const uint8_t cs = !(bits & TMI_BA_INTRINSICS_SKEW);
  const uint8_t cp = !(bits & TMI_BA_INTRINSICS_PRINCIPAL_POINTS);
  const uint8_t cr = !(bits & TMI_BA_INTRINSICS_RADIAL_DISTORTION);
  const uint8_t ct = !(bits & TMI_BA_INTRINSICS_TANGENTIAL_DISTORTION);
  if (model <= TMI_BA_FISHEYE) {
    mask[0] = cf; mask[1] = ca; mask[2] = cs; mask[3] = cp; mask[4] = cp;
    for (int i = 5; i < n; ++i) mask[i] = cr;
    if (model == TMI_BA_PINHOLE_RADIAL_TANGENTIAL) { mask[8] = ct; mask[9] = ct; }
  } else {
    mask[0] = cf; mask[1] = ca; mask[2] = cp; mask[3] = cp; mask[4] = cr;
  }
  return n;
}

void tmi_ba_solver_destroy(tmi_ba_solver* s) {
  if (s && s->d_ppcg_prof) {
    long long h[48];
    if (hipMemcpy(h, s->d_ppcg_prof, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
      static const char* names[8] = {"A rows", "barrier 1", "B cols + p.q", "reduce p.q", "-", "C update (+ resets)", "reduce Q1 rho", "-"};
      for (int wg = 0; wg < 3; ++wg) {
        fprintf(stderr, "[tmi_ba ppcg] %s workgroup, ms:", wg == 0 ? "first" : wg == 1 ? "middle" : "last");
        for (int i = 0; i < 7; ++i)
          if (i != 4) fprintf(stderr, " %s %.3f |", names[i], 1e-5 * (double)h[16 * wg + i]);
        fprintf(stderr, "\n");
      }
    }
  }
#ifdef TMI_LIN_PROFILE
  {
    unsigned long long h[8] = {0};
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lin_prof), sizeof(h)) == hipSuccess && h[5]) {
      const double t = (double)h[5];
      fprintf(stderr, "[linearize profile] per wave-trip (cycles): loads %.0f | stage A %.0f | eval A %.0f | stage B %.0f | columns + stores %.0f  (wave-trips %.0f, waves %.0f)\n",
              h[0] / t, h[1] / t, h[2] / t, h[3] / t, h[4] / t, t, (double)h[6]);
    }
  }
#endif
#ifdef TMI_MF_PROFILE
  if (s && s->mf_ok && s->mf.prof) {
    std::vector<long long> h((size_t)s->mf.n_items * 16);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), s->mf.prof, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double t[13] = {0}, units = 0, items = 0;
    for (int i = s->mf.nwb; i < s->mf.n_items; ++i) {
      for (int k = 0; k < 10; ++k) t[k] += (double)h[(size_t)i * 16 + k];
      t[11] += (double)h[(size_t)i * 16 + 11];
      t[12] += (double)h[(size_t)i * 16 + 12];
      units += (double)h[(size_t)i * 16 + 10];
      items += 1;
    }
    fprintf(stderr, "[mf profile] per narrow unit (cycles): issue %.0f | wait+u+wpart %.0f | bar1 %.0f | z %.0f | bar2 %.0f | t,v %.0f | bar3 %.0f | runs %.0f | long+bar %.0f | bar5 %.0f  (units %.0f); per item: before loop %.0f, write-out %.0f\n",
            t[0] / units, t[1] / units, t[2] / units, t[3] / units, t[4] / units, t[5] / units, t[6] / units, t[7] / units,
            t[8] / units, t[9] / units, units, t[11] / items, t[12] / items);
  }
#endif
  if (!s) return;
  hipSetDevice(s->device);
  if (s->stream) hipStreamSynchronize(s->stream);
  if (s->nccl_comm) rccl().CommDestroy(s->nccl_comm);
  for (auto& e : s->events) {
    hipEventDestroy(e.a);
    hipEventDestroy(e.b);
  }
  for (void* p : s->allocs) hipFree(p);
  if (s->h_mirror) hipHostFree(s->h_mirror);
  if (s->inner.h_active) hipHostFree(s->inner.h_active);
  if (s->h_counters) hipHostFree(s->h_counters);
  if (s->h_stage) hipHostFree(s->h_stage);
  if (s->h_cell_total) hipHostFree(s->h_cell_total);
  for (void* p : s->cell_allocs) hipFree(p);
  if (s->stream) hipStreamDestroy(s->stream);
  delete s;
}


// ---- static structure of the one-sweep matrix-free product (mf_chunks.h): units, runs, items, slots ----------
// From the resident layout (slice_ptr, obs_rb): works for host- and device-built structures and for every rank of a
// sharded handle.  Leaves s->mf_ok = false (the two-pass product of kernels.h stays in use) when the problem does not
// fit the kernel: a thread-per-track slice longer than kMaxNarrowK rows.
static int build_mf_chunks(tmi_ba_solver* s) {
  using namespace tmi::mfc;
  s->mf_ok = false;
  int rc_early = TMI_BA_OK;
  // Which product: measured on MI355X (profiles/r04_one_sweep_experiment.md, same-box pairs on venice1778_heavy, ms per
  // LM iteration, two-pass / one-sweep) a rank that holds the whole problem, a half, a quarter, an eighth, a sixteenth
  // of it takes 3.93 / 2.98, 2.33 / 1.77, 1.41 / 1.16, 0.77 / 0.70, 0.93 / 0.96.  The shards of a sharded handle are
  // dealt by work, not by observations (the rank with the longest tracks of eight holds ~450 k of 5.0 M), so: the
  // one-sweep product from 350 k observations per rank, the two-pass product below; TMI_BA_MF_ONE_SWEEP=1 / =0 forces
  // either (tests, A/B).
  {
    const char* e = getenv("TMI_BA_MF_ONE_SWEEP");
    const bool want = e ? atoi(e) != 0 : s->st.No >= 350000;
    if (!want) return TMI_BA_OK;
  }
  Structure& st = s->st;
  DeviceView& v = s->v;
  hipStream_t stream = s->stream;
  if (st.has_shared || st.Nrb == 0 || st.No_pad == 0 || st.nslices == 0) return TMI_BA_OK;
  const int Nrb = st.Nrb, D = st.D;
  auto rows_of = [&](int sl) { return (st.slice_ptr[sl + 1] - st.slice_ptr[sl]) >> 6; };
  // slices (descending length) with more rows than 64 lanes per track can hold in the row slots: a wavefront per
  // track, four tracks a unit (they are among the `ultra` slices of track_map); every other slice is a narrow unit
  // or is cut into narrow units
  const int max_rows = 64 * kWaves * reg_rows(D);
  int n_old = 0;
  while (n_old < st.nslices && rows_of(n_old) > max_rows) ++n_old;
  if (n_old > st.n_ultra) return TMI_BA_OK;  // (cannot happen: kUltraK <= max_rows)
  const int nub = 16 * n_old, nwb = nub;
  // narrow units: a slice, or a pack of consecutive slices of the same (small) length whose rows together fill the
  // kWaves * reg_rows(D) row slots of the kernel
  std::vector<int4> unit_desc;  // {first element, rows, rows of one slice | log2 L << 16, first slice}
  {
    const int slots = kWaves * reg_rows(D);
    for (int sl = n_old; sl < st.nslices;) {
      const int K = rows_of(sl);
      if (K > slots) {
        // a long slice: L pieces of 64 / L tracks, L lanes per track (mf_chunks.h)
        int lsh = 0;
        while ((slots << lsh) < K) ++lsh;
        for (int q = 0; q < (1 << lsh); ++q)
          unit_desc.push_back(make_int4(st.slice_ptr[sl] + q * (64 >> lsh), K, K | (lsh << 16), sl));
        ++sl;
        continue;
      }
      int G = 1;
      if (K > 0 && 2 * K <= slots)
        while (G < slots / K && sl + G < st.nslices && rows_of(sl + G) == K) ++G;
      unit_desc.push_back(make_int4(st.slice_ptr[sl], G * K, K, sl));
      sl += G;
    }
  }
  const int n_units = nwb + (int)unit_desc.size();
  int4* d_unit_desc;
  if ((rc_early = dev_upload(s, &d_unit_desc, unit_desc))) return rc_early;
  if (!s->num_cus) {
    hipDeviceProp_t prop;
    TMI_HIP(hipGetDeviceProperties(&prop, s->device));
    s->num_cus = prop.multiProcessorCount;
  }
  const long long n = st.No_pad;
  TempPool tmp;
  void* cub_tmp = nullptr;
  size_t cub_cap = 0;
  struct CubFree {
    void** p;
    ~CubFree() {
      if (*p) hipFree(*p);
    }
  } cub_free{&cub_tmp};
  auto cub_reserve = [&](size_t bytes) -> int {
    if (bytes <= cub_cap) return TMI_BA_OK;
    if (cub_tmp) hipFree(cub_tmp);
    cub_tmp = nullptr;
    cub_cap = 0;
    TMI_HIP(hipMalloc(&cub_tmp, bytes + 256));
    cub_cap = bytes + 256;
    return TMI_BA_OK;
  };
  int rc;
#define MF_SORT_PAIRS(kin, kout, vin, vout, cnt, bits)                                                      \
  do {                                                                                                      \
    size_t bytes_ = 0;                                                                                      \
    TMI_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes_, kin, kout, vin, vout, (int)(cnt), 0, bits, stream)); \
    if ((rc = cub_reserve(bytes_))) return rc;                                                              \
    TMI_HIP(hipcub::DeviceRadixSort::SortPairs(cub_tmp, bytes_, kin, kout, vin, vout, (int)(cnt), 0, bits, stream)); \
  } while (0)
#define MF_EXCLUSIVE_SUM(in, out, cnt)                                                                      \
  do {                                                                                                      \
    size_t bytes_ = 0;                                                                                      \
    TMI_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes_, in, out, (int)(cnt), stream));                \
    if ((rc = cub_reserve(bytes_))) return rc;                                                              \
    TMI_HIP(hipcub::DeviceScan::ExclusiveSum(cub_tmp, bytes_, in, out, (int)(cnt), stream));                \
  } while (0)
  auto nb = [](long long c) { return dim3((unsigned)std::max<long long>((c + 255) / 256, 1)); };

  // ---- runs: the observations sorted by (unit, view block)
  const unsigned long long invalid = (unsigned long long)n_units * (unsigned)Nrb;
  const int key_bits = bits_for(invalid + 1);
  unsigned long long *d_k_in, *d_k_out;
  int *d_v_in, *d_v_out, *d_head, *d_pos, *d_nvalid;
  TMI_HIP(tmp.get(&d_k_in, (size_t)n));
  TMI_HIP(tmp.get(&d_k_out, (size_t)n));
  TMI_HIP(tmp.get(&d_v_in, (size_t)n));
  TMI_HIP(tmp.get(&d_v_out, (size_t)n));
  TMI_HIP(tmp.get(&d_head, (size_t)n + 1));
  TMI_HIP(tmp.get(&d_pos, (size_t)n + 1));
  TMI_HIP(tmp.get(&d_nvalid, 1));
  hipLaunchKernelGGL(unit_keys_kernel, dim3(n_units), dim3(256), 0, stream, v, nub, nwb, Nrb, d_unit_desc, invalid, d_k_in, d_v_in);
  MF_SORT_PAIRS(d_k_in, d_k_out, d_v_in, d_v_out, n, key_bits);
  int n_valid = (int)n;
  TMI_HIP(hipMemcpyAsync(d_nvalid, &n_valid, sizeof(int), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(heads_kernel, nb(n + 1), dim3(256), 0, stream, d_k_out, n, invalid, d_head, d_nvalid);
  MF_EXCLUSIVE_SUM(d_head, d_pos, n + 1);
  int n_runs = 0;
  TMI_HIP(hipMemcpyAsync(&n_runs, d_pos + n, sizeof(int), hipMemcpyDeviceToHost, stream));
  TMI_HIP(hipMemcpyAsync(&n_valid, d_nvalid, sizeof(int), hipMemcpyDeviceToHost, stream));
  TMI_HIP(hipStreamSynchronize(stream));
  if (n_runs == 0) return TMI_BA_OK;
  int *d_run_first, *d_unit_run_ptr;
  unsigned long long* d_run_key;
  TMI_HIP(tmp.get(&d_run_first, (size_t)n_runs + 1));
  TMI_HIP(tmp.get(&d_run_key, (size_t)n_runs));
  TMI_HIP(tmp.get(&d_unit_run_ptr, (size_t)n_units + 1));
  hipLaunchKernelGGL(run_fill_kernel, nb(n), dim3(256), 0, stream, d_k_out, d_head, d_pos, n, d_run_first, d_run_key);
  TMI_HIP(hipMemcpyAsync(d_run_first + n_runs, &n_valid, sizeof(int), hipMemcpyHostToDevice, stream));
  hipLaunchKernelGGL(lower_bound_stride_kernel, nb(n_units + 1), dim3(256), 0, stream, d_run_key, (long long)n_runs,
                     (unsigned long long)Nrb, n_units, d_unit_run_ptr);

  // ---- items: wide / ultra units one each; the narrow slices in groups of about equal size, cut where the slice
  // length changes (the track order restarts at the lowest view there) and regrouped smaller when an item sees more
  // views than its accumulators hold
  unsigned long long *d_sk_in, *d_sk_out, *d_slot_key;
  int *d_sv_in, *d_sv_out, *d_head2, *d_pos2, *d_unit_item, *d_item_unit0, *d_run_slot, *d_slot_rb, *d_item_slot_ptr;
  TMI_HIP(tmp.get(&d_sk_in, (size_t)n_runs));
  TMI_HIP(tmp.get(&d_sk_out, (size_t)n_runs));
  TMI_HIP(tmp.get(&d_sv_in, (size_t)n_runs));
  TMI_HIP(tmp.get(&d_sv_out, (size_t)n_runs));
  TMI_HIP(tmp.get(&d_head2, (size_t)n_runs + 1));
  TMI_HIP(tmp.get(&d_pos2, (size_t)n_runs + 1));
  TMI_HIP(tmp.get(&d_unit_item, (size_t)n_units));
  TMI_HIP(tmp.get(&d_item_unit0, (size_t)n_units + 1));
  TMI_HIP(tmp.get(&d_run_slot, (size_t)n_runs));
  TMI_HIP(tmp.get(&d_slot_rb, (size_t)n_runs));
  TMI_HIP(tmp.get(&d_slot_key, (size_t)n_runs));
  TMI_HIP(tmp.get(&d_item_slot_ptr, (size_t)n_units + 1));
  const long long narrow_elems = (long long)st.slice_ptr[st.nslices] - st.slice_ptr[n_old];
  long long target = std::max<long long>(64, std::min<long long>(4096, narrow_elems / std::max(1, 4 * s->num_cus)));
  if (const char* e = getenv("TMI_BA_MF_ITEM")) target = std::max(64, atoi(e));  // A/B: elements per item
  // narrow items as unit ranges: about `target` elements each, cut where the slice length changes
  std::vector<std::pair<int, int>> groups;
  {
    long long have = 0;
    int last_rows = -1;
    for (size_t i = 0; i < unit_desc.size(); ++i) {
      const int u = nwb + (int)i, rows = unit_desc[i].z;
      if (last_rows < 0 || rows != last_rows || have >= target) {
        groups.emplace_back(u, u + 1);
        have = 0;
      } else {
        groups.back().second = u + 1;
      }
      have += (long long)unit_desc[i].y * (64 >> (unit_desc[i].z >> 16));
      last_rows = rows;
    }
  }
  std::vector<int> unit_item((size_t)n_units), item_unit0, item_slot_ptr;
  int n_items = 0, n_slots = 0;
  for (int attempt = 0;; ++attempt) {
    item_unit0.clear();
    for (int u = 0; u < nwb; ++u) {
      item_unit0.push_back(u);
      unit_item[u] = u;
    }
    for (const auto& g : groups) {
      for (int u = g.first; u < g.second; ++u) unit_item[u] = (int)item_unit0.size();
      item_unit0.push_back(g.first);
    }
    n_items = (int)item_unit0.size();
    item_unit0.push_back(n_units);
    TMI_HIP(hipMemcpyAsync(d_unit_item, unit_item.data(), (size_t)n_units * sizeof(int), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(slot_keys_kernel, nb(n_runs), dim3(256), 0, stream, d_run_key, n_runs, d_unit_item, Nrb, d_sk_in, d_sv_in);
    MF_SORT_PAIRS(d_sk_in, d_sk_out, d_sv_in, d_sv_out, n_runs, bits_for((unsigned long long)n_items * (unsigned)Nrb + 1));
    hipLaunchKernelGGL(heads_kernel, nb(n_runs + 1), dim3(256), 0, stream, d_sk_out, (long long)n_runs, ~0ull, d_head2, d_nvalid);
    MF_EXCLUSIVE_SUM(d_head2, d_pos2, n_runs + 1);
    TMI_HIP(hipMemcpyAsync(&n_slots, d_pos2 + n_runs, sizeof(int), hipMemcpyDeviceToHost, stream));
    hipLaunchKernelGGL(slot_fill_kernel, nb(n_runs), dim3(256), 0, stream, d_sk_out, d_head2, d_pos2, d_sv_out, n_runs, Nrb,
                       d_run_slot, d_slot_rb, d_slot_key);
    TMI_HIP(hipStreamSynchronize(stream));
    hipLaunchKernelGGL(lower_bound_stride_kernel, nb(n_items + 1), dim3(256), 0, stream, d_slot_key, (long long)n_slots,
                       (unsigned long long)Nrb, n_items, d_item_slot_ptr);
    item_slot_ptr.resize((size_t)n_items + 1);
    TMI_HIP(hipMemcpyAsync(item_slot_ptr.data(), d_item_slot_ptr, ((size_t)n_items + 1) * sizeof(int), hipMemcpyDeviceToHost, stream));
    TMI_HIP(hipStreamSynchronize(stream));
    // an item of several slices that sees more views than the accumulators hold is cut in two (an item of one slice
    // needs none: mf_chunks.h)
    std::vector<std::pair<int, int>> next;
    bool split = false;
    for (size_t g = 0; g < groups.size(); ++g) {
      const int i = nwb + (int)g, views = item_slot_ptr[i + 1] - item_slot_ptr[i];
      const int u0 = groups[g].first, u1 = groups[g].second;
      if (views > lc_max(D) && u1 - u0 > 1) {
        const int mid = u0 + (u1 - u0) / 2;
        next.emplace_back(u0, mid);
        next.emplace_back(mid, u1);
        split = true;
      } else {
        next.push_back(groups[g]);
      }
    }
    if (!split) break;
    if (attempt >= 16) return TMI_BA_OK;  // cannot happen (an item halves every round); the two-pass product serves
    groups.swap(next);
  }

  // ---- persistent arrays
  View& m = s->mf;
  memset(&m, 0, sizeof(m));
  m.n_items = n_items;
  m.n_units = n_units;
  m.n_runs = n_runs;
  m.nub = nub;
  m.nwb = nwb;
  int *p_item_unit0, *p_unit_run_ptr, *p_run_obs_ptr, *p_run_obs, *p_run_slot, *p_item_slot_ptr, *p_slot_rb, *p_obs_pos,
      *p_cam_slot_ptr, *p_cam_slots;
  if ((rc = dev_upload(s, &p_item_unit0, item_unit0))) return rc;
  if ((rc = dev_alloc(s, &p_unit_run_ptr, (size_t)n_units + 1))) return rc;
  if ((rc = dev_alloc(s, &p_run_obs_ptr, (size_t)n_runs + 1))) return rc;
  if ((rc = dev_alloc(s, &p_run_obs, (size_t)std::max(n_valid, 1)))) return rc;
  if ((rc = dev_alloc(s, &p_run_slot, (size_t)n_runs))) return rc;
  if ((rc = dev_alloc(s, &p_item_slot_ptr, (size_t)n_items + 1))) return rc;
  if ((rc = dev_alloc(s, &p_slot_rb, (size_t)n_slots))) return rc;
  if ((rc = dev_alloc(s, &p_obs_pos, (size_t)n))) return rc;
  if ((rc = dev_alloc(s, &p_cam_slot_ptr, (size_t)Nrb + 1))) return rc;
  if ((rc = dev_alloc(s, &p_cam_slots, (size_t)n_slots))) return rc;
  TMI_HIP(hipMemcpyAsync(p_unit_run_ptr, d_unit_run_ptr, ((size_t)n_units + 1) * sizeof(int), hipMemcpyDeviceToDevice, stream));
  TMI_HIP(hipMemcpyAsync(p_run_obs_ptr, d_run_first, ((size_t)n_runs + 1) * sizeof(int), hipMemcpyDeviceToDevice, stream));
  TMI_HIP(hipMemcpyAsync(p_run_obs, d_v_out, (size_t)n_valid * sizeof(int), hipMemcpyDeviceToDevice, stream));
  TMI_HIP(hipMemcpyAsync(p_run_slot, d_run_slot, (size_t)n_runs * sizeof(int), hipMemcpyDeviceToDevice, stream));
  TMI_HIP(hipMemcpyAsync(p_item_slot_ptr, d_item_slot_ptr, ((size_t)n_items + 1) * sizeof(int), hipMemcpyDeviceToDevice, stream));
  TMI_HIP(hipMemcpyAsync(p_slot_rb, d_slot_rb, (size_t)n_slots * sizeof(int), hipMemcpyDeviceToDevice, stream));
  TMI_HIP(hipMemsetAsync(p_obs_pos, 0xff, (size_t)n * sizeof(int), stream));
  hipLaunchKernelGGL(obs_pos_kernel, nb(n_valid), dim3(256), 0, stream, d_k_out, d_v_out, n_valid, Nrb, d_unit_run_ptr,
                     d_run_first, p_obs_pos);
  {
    // slots of every view block, ascending: the order mfc::reduce_kernel adds them in
    unsigned *d_ck_in, *d_ck_out;
    int* d_cv_in;
    TMI_HIP(tmp.get(&d_ck_in, (size_t)n_slots));
    TMI_HIP(tmp.get(&d_ck_out, (size_t)n_slots));
    TMI_HIP(tmp.get(&d_cv_in, (size_t)n_slots));
    hipLaunchKernelGGL(slot_rb_keys_kernel, nb(n_slots), dim3(256), 0, stream, d_slot_rb, n_slots, d_ck_in, d_cv_in);
    MF_SORT_PAIRS(d_ck_in, d_ck_out, d_cv_in, p_cam_slots, n_slots, bits_for((unsigned long long)Nrb + 1));
    hipLaunchKernelGGL(lower_bound_u32_kernel, nb(Nrb + 1), dim3(256), 0, stream, d_ck_out, n_slots, Nrb, p_cam_slot_ptr);
    TMI_HIP(hipStreamSynchronize(stream));
  }
#undef MF_SORT_PAIRS
#undef MF_EXCLUSIVE_SUM
  double *p_partial, *p_ut;
  if ((rc = dev_alloc(s, &p_partial, (size_t)n_slots * D))) return rc;
  if ((rc = dev_alloc(s, &p_ut, (size_t)2 * std::max(st.slice_ptr[n_old], 1)))) return rc;
  {
    // launch order: the wavefront-per-track units first (as before), then the narrow items by descending size --
    // the hardware hands out workgroups in index order, and the largest items (packs of the shortest tracks) used to
    // come last
    std::vector<int> order((size_t)n_items);
    std::vector<long long> elems((size_t)n_items, 0);
    for (int i = 0; i < n_items; ++i) order[i] = i;
    for (size_t g = 0; g < groups.size(); ++g)
      for (int u = groups[g].first; u < groups[g].second; ++u) {
        const int4& d = unit_desc[(size_t)(u - nwb)];
        elems[(size_t)nwb + g] += (long long)d.y * (64 >> (d.z >> 16));
      }
    std::stable_sort(order.begin() + nwb, order.end(), [&](int a, int b) { return elems[a] > elems[b]; });
    int* p_order;
    if ((rc = dev_upload(s, &p_order, order))) return rc;
    m.item_order = p_order;
  }
  {
    int4* p_hdr;
    if ((rc = dev_alloc(s, &p_hdr, (size_t)n_items * 6))) return rc;
    hipLaunchKernelGGL(item_hdr_kernel, nb(n_items), dim3(256), 0, stream, n_items, nwb, n_units, p_item_unit0, p_item_slot_ptr,
                       d_unit_desc, p_unit_run_ptr, p_run_obs_ptr, p_hdr);
    m.item_hdr = p_hdr;
  }
  m.unit_desc = d_unit_desc;
  m.item_unit0 = p_item_unit0;
  m.unit_run_ptr = p_unit_run_ptr;
  m.run_obs_ptr = p_run_obs_ptr;
  m.run_obs = p_run_obs;
  m.run_slot = p_run_slot;
  m.item_slot_ptr = p_item_slot_ptr;
  m.slot_rb = p_slot_rb;
  m.obs_pos = p_obs_pos;
  m.cam_slot_ptr = p_cam_slot_ptr;
  m.cam_slots = p_cam_slots;
  m.partial = p_partial;
#ifdef TMI_MF_PROFILE
  {
    long long* pp;
    if ((rc = dev_alloc(s, &pp, (size_t)n_items * 16))) return rc;
    TMI_HIP(hipMemsetAsync(pp, 0, (size_t)n_items * 16 * sizeof(long long), stream));
    m.prof = pp;
  }
#endif
  m.ut = p_ut;
  TMI_HIP(hipStreamSynchronize(stream));
  s->mf_ok = true;
  if (getenv("TMI_BA_SETUP_TIMING")) {
    int occ = -1;
    if (D == 9 && s->DP == 3) hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, mfc::product_kernel<9, 3, false>, mfc::kThreads, 0);
    fprintf(stderr, "[tmi_ba setup] one-sweep product: %d units, %d items, %d runs, %d slots (%.1f MB of partials), "
                    "%d workgroups per CU\n", n_units, n_items, n_runs, n_slots, 8e-6 * n_slots * D, occ);
  }
  return TMI_BA_OK;
}


// ---- CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL on a problem without shared intrinsics blocks: the views clustered by
// visibility as ceres::VisibilityBasedPreconditioner::ClusterCameras does (Ceres is external to the reference --
// bundle_adjuster.cc:59-63 only passes preconditioner_type and visibility_clustering_type on -- so this restates Ceres
// 1.14: visibility.cc CreateSchurComplementGraph, canonical_views_clustering.cc, single_linkage_clustering.cc,
// visibility_based_preconditioner.cc; parity unpinned like the rest of that layer).
//   graph   : one vertex per camera-side PARAMETER block -- a view's extrinsics and, when it has free private
//             intrinsics, its intrinsics block (identical visibility: similarity 1 between the two) --, a self edge of
//             weight 1 on every vertex, and between two blocks the weight |tracks both see| / sqrt(|tracks of a| |tracks
//             of b|) over the NON-constant tracks (constant points are no e-blocks);
//   CANONICAL_VIEWS: greedily the view whose promotion to a centre gains most -- sum over its neighbours of the
//             similarity they would win, minus size_penalty_weight = 3 (similarity_penalty_weight = 0, view_score_weight
//             = 0: visibility_based_preconditioner.cc's constants) -- until the gain is <= 0 and there are >= 3 centres;
//             every vertex joins the centre it is most similar to; a vertex that touches no centre goes to cluster
//             (its index mod #clusters), as FlattenMembershipMap does;
//   SINGLE_LINKAGE : the connected components of the edges with similarity >= 0.9.
// Ties go to the lower vertex index (Ceres iterates hash sets: its order is the STL's).  The two blocks of a view end
// up in one cluster by construction (similarity 1), so a cluster is a set of views = reduced blocks here.
static int build_visibility_clusters(tmi_ba_solver* s, const tmi_ba_problem* P, int type) {
  Structure& st = s->st;
  const int Nrb = st.Nrb, D = st.D;
  s->vis_members.clear();
  if (Nrb == 0) return TMI_BA_OK;
  // host copies of the block structure of S and of the pair counts
  std::vector<long long> pair_ptr;
  if (s->device_structure) {
    st.urow_ptr.resize((size_t)Nrb + 1);
    st.ub_j.resize((size_t)st.nub);
    pair_ptr.resize((size_t)st.nub + 1);
    TMI_HIP(hipMemcpy(st.urow_ptr.data(), s->v.urow_ptr, ((size_t)Nrb + 1) * sizeof(int), hipMemcpyDeviceToHost));
    if (st.nub) {
      TMI_HIP(hipMemcpy(st.ub_j.data(), s->v.ub_j, (size_t)st.nub * sizeof(int), hipMemcpyDeviceToHost));
      TMI_HIP(hipMemcpy(pair_ptr.data(), s->v.pair_ptr, ((size_t)st.nub + 1) * sizeof(long long), hipMemcpyDeviceToHost));
    }
  } else {
    pair_ptr.assign(st.pair_ptr.begin(), st.pair_ptr.end());
  }
  // tracks of a view (non-constant ones), parameter blocks of a view
  std::vector<double> ntr((size_t)Nrb, 0.0);
  for (int64_t i = 0; i < P->num_observations; ++i) {
    const int rb = st.cam_rb[P->obs_camera[i]];
    if (rb >= 0 && !(P->point_constant && P->point_constant[P->obs_point[i]])) ntr[rb] += 1.0;
  }
  std::vector<int> mult((size_t)Nrb, 0);
  for (int rb = 0; rb < Nrb; ++rb) {
    bool ext = false, intr = false;
    for (int c = 0; c < D; ++c) {
      const int b = st.rb_cols[(size_t)rb * D + c];
      if (b >= 0 && b < 6) ext = true;
      if (b >= 6) intr = true;
    }
    mult[rb] = (ext ? 1 : 0) + (intr ? 1 : 0);
  }
  // symmetric adjacency of the views: (neighbour, similarity)
  std::vector<std::vector<std::pair<int, double> > > adj((size_t)Nrb);
  for (int i = 0; i < Nrb; ++i)
    for (int u = st.urow_ptr[i]; u < st.urow_ptr[i + 1]; ++u) {
      const int j = st.ub_j[u];
      const double cnt = (double)(pair_ptr[u + 1] - pair_ptr[u]);
      if (cnt <= 0.0 || ntr[i] <= 0.0 || ntr[j] <= 0.0) continue;
      const double w = cnt / std::sqrt(ntr[i] * ntr[j]);
      adj[i].emplace_back(j, w);
      adj[j].emplace_back(i, w);
    }
  std::vector<int> cluster((size_t)Nrb, -1);
  int ncl = 0;
  if (type == 1) {
    // SINGLE_LINKAGE (kSingleLinkageMinSimilarity = 0.9)
    std::vector<int> parent((size_t)Nrb);
    for (int i = 0; i < Nrb; ++i) parent[i] = i;
    std::function<int(int)> find = [&](int a) { return parent[a] == a ? a : parent[a] = find(parent[a]); };
    for (int i = 0; i < Nrb; ++i)
      for (const auto& e : adj[i])
        if (e.first > i && e.second >= 0.9) {
          const int a = find(i), b = find(e.first);
          if (a != b) parent[std::max(a, b)] = std::min(a, b);
        }
    std::vector<int> id((size_t)Nrb, -1);
    for (int i = 0; i < Nrb; ++i) {
      if (mult[i] == 0) continue;
      const int r = find(i);
      if (id[r] < 0) id[r] = ncl++;
      cluster[i] = id[r];
    }
  } else {
    // CANONICAL_VIEWS on the parameter-block graph.  The blocks of a view have the same neighbours, so the state is
    // kept per view: sim[u] = similarity of u's blocks to their centre, and a view's blocks count mult[u] times.
    const double size_penalty = 3.0, similarity_penalty = 0.0;
    const int min_views = 3;
    std::vector<double> sim((size_t)Nrb, 0.0);
    std::vector<int> to_center((size_t)Nrb, -1);
    std::vector<int> centers_of_view((size_t)Nrb, 0);  // how many of the view's blocks are centres already
    std::vector<int> centers;                          // views (a view can appear twice: both of its blocks)
    for (;;) {
      int best = -1;
      double best_gain = -1.7976931348623157e308;
      for (int vtx = 0; vtx < Nrb; ++vtx) {
        if (centers_of_view[vtx] >= mult[vtx]) continue;  // no block of this view is left to promote
        double gain = 0.0;
        // its own blocks: the self edge and the edge between extrinsics and intrinsics, weight 1 each
        if (1.0 > sim[vtx]) gain += mult[vtx] * (1.0 - sim[vtx]);
        for (const auto& e : adj[vtx])
          if (e.second > sim[e.first]) gain += mult[e.first] * (e.second - sim[e.first]);
        gain -= size_penalty;
        if (similarity_penalty != 0.0)
          for (const int c : centers) {
            if (c == vtx) { gain -= similarity_penalty; continue; }
            for (const auto& e : adj[vtx])
              if (e.first == c) gain -= similarity_penalty * e.second;
          }
        if (gain > best_gain) {
          best_gain = gain;
          best = vtx;
        }
      }
      if (best < 0) break;  // every block is a centre
      if (best_gain <= 0.0 && (int)centers.size() >= min_views) break;
      centers.push_back(best);
      ++centers_of_view[best];
      if (1.0 > sim[best]) {
        sim[best] = 1.0;
        to_center[best] = best;
      }
      for (const auto& e : adj[best])
        if (e.second > sim[e.first]) {
          sim[e.first] = e.second;
          to_center[e.first] = best;
        }
    }
    // cluster ids: one per CENTRE VIEW (a view promoted twice is one cluster: its second block joined the first)
    std::vector<int> id((size_t)Nrb, -1);
    for (const int c : centers)
      if (id[c] < 0) id[c] = ncl++;
    for (int i = 0; i < Nrb; ++i) {
      if (mult[i] == 0) continue;
      cluster[i] = to_center[i] >= 0 ? id[to_center[i]] : (ncl > 0 ? i % ncl : 0);
    }
    if (ncl == 0) ncl = 1;
  }
  s->vis_cluster_of_rb = cluster;
  std::vector<std::vector<int> > members((size_t)ncl);
  for (int i = 0; i < Nrb; ++i)
    if (cluster[i] >= 0) members[cluster[i]].push_back(i);  // ascending
  for (auto& m : members)
    if (m.size() >= 2) s->vis_members.push_back(m);  // (a cluster of one view is its SCHUR_JACOBI block)
  if (getenv("TMI_BA_SETUP_TIMING")) {
    size_t big = 0;
    for (const auto& m : s->vis_members) big = std::max(big, m.size());
    fprintf(stderr, "[tmi_ba setup] visibility clusters (%s): %d clusters, %zu with more than one view, largest %zu views\n",
            type == 1 ? "SINGLE_LINKAGE" : "CANONICAL_VIEWS", ncl, s->vis_members.size(), big);
    if (Nrb <= 64) {
      fprintf(stderr, "[tmi_ba setup]   cluster of every view block:");
      for (int i = 0; i < Nrb; ++i) fprintf(stderr, " %d", cluster[i]);
      fprintf(stderr, "\n");
    }
  }
  return TMI_BA_OK;
}

static int create_impl(tmi_ba_solver* s, const tmi_ba_problem* P, const tmi_ba_options* O, int rank,
                       int world, bool light = false) {
  s->light = light;
  const double t0 = now_s();
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    s->error = "no HIP device visible (the device path has no CPU fallback)";
    return TMI_BA_ERR_NO_DEVICE;
  }
  if (O->device >= 0) {
    if (O->device >= ndev) {
      s->error = "options.device out of range";
      return TMI_BA_ERR_INVALID_ARGUMENT;
    }
    s->device = O->device;
  } else {
    TMI_HIP(hipGetDevice(&s->device));
  }
  TMI_HIP(hipSetDevice(s->device));
  if (O->point_dof != 3 && O->point_dof != 4) {
    s->error = "point_dof must be 3 or 4";
    return TMI_BA_ERR_INVALID_ARGUMENT;
  }
  if (O->residual_precision != 64 && O->residual_precision != 32 && O->residual_precision != 0) {
    s->error = "residual_precision must be 64 or 32";
    return TMI_BA_ERR_INVALID_ARGUMENT;
  }
  s->DP = O->point_dof;
  const bool iterative_type =
      (O->linear_solver_type == TMI_BA_ITERATIVE_SCHUR || O->linear_solver_type == TMI_BA_CGNR);
  if (O->schur_mode < 0 || O->schur_mode > 2) {
    s->error = "schur_mode must be 0 (auto), 1 (explicit) or 2 (implicit)";
    return TMI_BA_ERR_INVALID_ARGUMENT;
  }
  // implicit needs an iterative solver; auto = explicit on one GPU, implicit on several
  s->implicit = iterative_type && (O->schur_mode == 2 || (O->schur_mode == 0 && world > 1));
  if (light) s->implicit = false;
  s->adaptive = !light && iterative_type && O->schur_mode == 0 && world == 1;
  s->implicit_now = s->implicit;
  TMI_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  bool want_pairs = !s->implicit && !light;
  const bool setup_timing = getenv("TMI_BA_SETUP_TIMING") != nullptr;
  // the camera side on the host (tiny), then the observation-sized structure on the device when the
  // problem shape allows it (structure_gpu.h), else on host threads (structure.cpp)
  int rc = build_blocks(P, rank, world, &s->st);
  if (rc) {
    s->error = s->st.error;
    return rc;
  }
  // CLUSTER_JACOBI on a problem with shared intrinsics blocks (cluster_precond.h): with schur_mode auto the operator is
  // the matrix-free one and only the blocks INSIDE the clusters are formed (what the preconditioner factors); an
  // explicit request for the formed / the matrix-free operator is honoured, the latter with the cluster blocks as well
  // CLUSTER_TRIDIAGONAL (cluster_chains.h) also needs the blocks BETWEEN neighbouring clusters: the formed S, on one
  // rank (the cluster graph counts tracks globally).  With schur_mode implicit or on several ranks the handle keeps its
  // SCHUR_JACOBI blocks, like CLUSTER_JACOBI without shared blocks (tmi_ba_summary::effective_preconditioner_type).
  const bool tri_pre = O->preconditioner_type == TMI_BA_PRECOND_CLUSTER_TRIDIAGONAL && iterative_type && !light;
  s->tri = tri_pre && world == 1 && O->schur_mode != 2;
  const bool cluster_pre = O->preconditioner_type == TMI_BA_PRECOND_CLUSTER_JACOBI || s->tri;
  if (s->tri && s->st.has_shared) {
    s->implicit = s->implicit_now = false;
    s->adaptive = false;
    s->cluster_blocks = false;
    want_pairs = true;
  } else if (cluster_pre && s->st.has_shared && iterative_type && !light) {
    if (O->schur_mode == 0) s->implicit = s->implicit_now = true;
    s->cluster_blocks = s->implicit;
    want_pairs = !s->implicit;
  }
  // ... and on a problem WITHOUT shared blocks: Ceres' clusters of the views by visibility (build_visibility_clusters).
  // They are principal submatrices of the formed S and come from the GLOBAL co-visibility counts, which a rank of a
  // sharded handle does not have: one rank only (several ranks: SCHUR_JACOBI, as before round 4).
  // (schur_mode implicit: no formed S to take the clusters from -- SCHUR_JACOBI, as on several ranks)
  if (cluster_pre && !s->st.has_shared && iterative_type && !light && world == 1 && O->schur_mode != 2) {
    s->vis_clusters = true;
    s->vis_type = O->visibility_clustering_type;
    s->implicit = s->implicit_now = false;
    s->adaptive = false;
    want_pairs = true;
  }
  if (device_setup_possible(s->st, world, P->num_observations, want_pairs)) {
    s->device_structure = true;
    memset(&s->v, 0, sizeof(s->v));
    rc = build_structure_device(s, P, want_pairs);
    if (rc) return rc;
  } else {
    s->st = Structure();
    rc = build_structure(P, rank, world, &s->st, want_pairs ? 1 : (s->cluster_blocks ? 2 : 0));
    if (rc) {
      s->error = s->st.error;
      return rc;
    }
  }
  Structure& st = s->st;
  if (setup_timing)
    fprintf(stderr, "[tmi_ba setup] %-28s %.3f s (%s)\n", "structure total", now_s() - t0,
            s->device_structure ? "device" : "host");
  if (!get_launch(st.D, s->DP, st.has_shared, O->residual_precision == 32, &s->launch)) {
    s->error = "no kernel instantiation for this block size";
    return TMI_BA_ERR_UNSUPPORTED;
  }
  // two slots: pcg_step of iteration `it` publishes into slot it & 1, so that the speculatively
  // launched next iteration can never overwrite scalars the host is still reading; every other
  // read-back uses slot 0
  TMI_HIP(hipHostMalloc((void**)&s->h_mirror, 2 * sizeof(HostMirror), hipHostMallocMapped | hipHostMallocCoherent));
  memset(s->h_mirror, 0, 2 * sizeof(HostMirror));
  TMI_HIP(hipHostGetDevicePointer((void**)&s->d_mirror, s->h_mirror, 0));
  s->h_scal = s->h_mirror->scal;
  s->h_red = s->h_mirror->red;
  s->h_flags = s->h_mirror->flags;

  const int D = st.D, DP = s->DP;
  DeviceView& v = s->v;
  const DeviceView built = s->v;  // pointers of a device-built structure (all null otherwise)
  memset(&v, 0, sizeof(v));
  v.Nc = st.Nc; v.G = st.G; v.Np_pad = st.Np_pad; v.nslices = st.nslices; v.Nrb = st.Nrb;
  v.Ncam_rb = st.Ncam_rb; v.has_shared = st.has_shared ? 1 : 0;
  v.planes_fp32 = (st.has_shared && O->residual_precision == 32) ? 1 : 0;  // (make_launch picks the float-plane kernels)
  v.D = D; v.DP = DP; v.No_pad = (int)st.No_pad; v.Nslots = (int)st.Nslots;
  v.nub = (int)st.nub; v.nnzb = (int)st.nnzb; v.npairs = st.npairs;
  v.n_order = s->device_structure ? s->n_order_dev : (int)st.ub_order.size();
  v.n_spc = s->device_structure ? s->n_spc_dev : (int)st.spc_row.size();
  s->RL = red_layout(st.nub, st.Nrb, D);
  s->n_intr = st.G ? P->group_offset[st.G] : 0;

  // parameters in device order
  s->ext0.assign(P->extrinsics, P->extrinsics + (size_t)6 * st.Nc);
  s->intr0.assign(P->intrinsics, P->intrinsics + s->n_intr);
  s->pts0.assign((size_t)4 * st.Np_pad, 0.0);
  for (int lp = 0; lp < st.Np_pad; ++lp) {
    const int p = st.pt_orig[lp];
    for (int a = 0; a < 4; ++a) s->pts0[(size_t)4 * lp + a] = (p >= 0) ? P->points[(size_t)4 * p + a] : (a == 3 ? 1.0 : 0.0);
  }
#define UP(field, vec)                                            \
  do {                                                            \
    rc = dev_upload(s, const_cast<std::remove_const<std::remove_pointer<decltype(v.field)>::type>::type**>(&v.field), vec); \
    if (rc) return rc;                                            \
  } while (0)
  {
    double *d_ext, *d_intr, *d_pts;
    if ((rc = dev_upload(s, &d_ext, s->ext0))) return rc;
    if ((rc = dev_upload(s, &d_intr, s->intr0))) return rc;
    if ((rc = dev_upload(s, &d_pts, s->pts0))) return rc;
    v.ext = d_ext; v.intr = d_intr; v.pts = d_pts;
    if ((rc = dev_upload(s, &d_ext, s->ext0))) return rc;
    if ((rc = dev_upload(s, &d_intr, s->intr0))) return rc;
    if ((rc = dev_upload(s, &d_pts, s->pts0))) return rc;
    v.ext_c = d_ext; v.intr_c = d_intr; v.pts_c = d_pts;
    if (!light) {
      if ((rc = dev_upload(s, &s->d_ext0, s->ext0))) return rc;
      if ((rc = dev_upload(s, &s->d_intr0, s->intr0))) return rc;
      if ((rc = dev_upload(s, &s->d_pts0, s->pts0))) return rc;
    }
  }
  std::vector<int> cam_grp(P->camera_group, P->camera_group + st.Nc);
  std::vector<int> grp_model(P->group_model, P->group_model + st.G);
  std::vector<int> grp_off(P->group_offset, P->group_offset + st.G + 1);
  s->grp_off_h = grp_off;
  std::vector<signed char> rb_cols(st.rb_cols.begin(), st.rb_cols.end());
  std::vector<long long> pair_ptr(st.pair_ptr.begin(), st.pair_ptr.end());
  {
    int* p; unsigned* pu; unsigned char* pc; signed char* ps; long long* pl; double* pd;
#define UPI(dst, vec) if ((rc = dev_upload(s, &p, vec))) return rc; dst = p;
    // camera-side arrays (small, built on the host in both paths)
    UPI(v.cam_grp, cam_grp) UPI(v.cam_rb, st.cam_rb)
    UPI(v.grp_model, grp_model) UPI(v.grp_off, grp_off) UPI(v.rb_cam, st.rb_cam) UPI(v.rb_grp, st.rb_grp)
    UPI(v.cam_grb, st.cam_grb) UPI(v.cam_cross_u, st.cam_cross_u)
    UPI(v.grp_cam_ptr, st.grp_cam_ptr) UPI(v.grp_cams, st.grp_cams)
    if (s->device_structure) {
      // observation-sized arrays were built in HBM (structure_gpu.h)
      v.slice_ptr = built.slice_ptr; v.pt_k = built.pt_k; v.pt_const = built.pt_const;
      v.obs_cam = built.obs_cam; v.obs_xy = built.obs_xy; v.obs_cpos = built.obs_cpos; v.cam_ptr = built.cam_ptr;
      v.urow_ptr = built.urow_ptr; v.ub_i = built.ub_i; v.ub_j = built.ub_j; v.ucol_ptr = built.ucol_ptr;
      v.ucol_u = built.ucol_u; v.spc_row = built.spc_row; v.spc_u0 = built.spc_u0; v.spc_rptr = built.spc_rptr;
      v.pair_ptr = built.pair_ptr; v.pair_i = built.pair_i; v.pair_j = built.pair_j; v.ub_order = built.ub_order;
      UPI(v.obs_gslot, st.obs_gslot)  // empty (no shared intrinsics blocks on this path)
    } else {
    UPI(v.slice_ptr, st.slice_ptr) UPI(v.pt_k, st.pt_k) UPI(v.obs_cam, st.obs_cam)
    UPI(v.obs_cpos, st.obs_cpos) UPI(v.obs_gslot, st.obs_gslot)
    UPI(v.cam_ptr, st.cam_ptr) UPI(v.urow_ptr, st.urow_ptr) UPI(v.ub_i, st.ub_i) UPI(v.ub_j, st.ub_j)
    UPI(v.ucol_ptr, st.ucol_ptr) UPI(v.ucol_u, st.ucol_u)
    UPI(v.spc_row, st.spc_row) UPI(v.spc_u0, st.spc_u0) UPI(v.spc_rptr, st.spc_rptr)
    UPI(v.pair_i, st.pair_i) UPI(v.pair_j, st.pair_j)
    {
      // launch headers of schur_offdiag: {block, #pairs, first pair lo, first pair hi}
      std::vector<int> hdr(st.ub_order.size() * 4, -1);
      for (size_t k = 0; k < st.ub_order.size(); ++k) {
        const int u = st.ub_order[k];
        if (u < 0) continue;
        const long long p0 = st.pair_ptr[u];
        hdr[4 * k] = u;
        hdr[4 * k + 1] = (int)(st.pair_ptr[u + 1] - p0);
        hdr[4 * k + 2] = (int)(unsigned)(p0 & 0xffffffffLL);
        hdr[4 * k + 3] = (int)(p0 >> 32);
      }
      UPI(v.ub_order, hdr)
    }
    }
#undef UPI
    if ((rc = dev_upload(s, &pu, st.cam_mask))) return rc; v.cam_mask = pu;
    {
      std::vector<int4> rec((size_t)st.Nc);
      for (int c = 0; c < st.Nc; ++c) {
        const int g = cam_grp[c];
        rec[c] = make_int4(grp_model[g], grp_off[g], grp_off[g + 1] - grp_off[g], (int)st.cam_mask[c]);
      }
      int4* pr;
      if ((rc = dev_upload(s, &pr, rec))) return rc;
      v.cam_rec = pr;
      // one camera model and one free-column mask for the whole problem?  (the specialised linearize, kernels.h)
      // (a view that is constant altogether -- mask 0, no block: the views outside the subset of
      //  BundleAdjustPartialReconstruction, bundle_adjuster.cc:141-180 -- does not break the uniformity: its observations
      //  are evaluated with the same model and nobody reads a camera block of theirs)
      bool uni = st.Nc > 0 && !st.has_shared;
      bool any_free = false;
      for (int c = 0; c < st.Nc && uni; ++c) {
        uni = rec[c].x == TMI_BA_PINHOLE && ((unsigned)rec[c].w == kPinholeDefaultMask || rec[c].w == 0);
        any_free = any_free || rec[c].w != 0;
      }
      uni = uni && any_free;
      if (const char* e = getenv("TMI_BA_LINEARIZE_GENERIC")) uni = uni && atoi(e) == 0;  // A/B, tests
      v.uniform_pinhole_default = uni ? 1 : 0;
    }
    if ((rc = dev_upload(s, &pu, st.grp_mask))) return rc; v.grp_mask = pu;
    if ((rc = dev_upload(s, &pc, st.obs_gflag))) return rc; v.obs_gflag = pc;
    if ((rc = dev_upload(s, &ps, rb_cols))) return rc; v.rb_cols = ps;
    if (!s->device_structure) {
      if ((rc = dev_upload(s, &pc, st.pt_const))) return rc; v.pt_const = pc;
      if ((rc = dev_upload(s, &pl, pair_ptr))) return rc; v.pair_ptr = pl;
      if ((rc = dev_upload(s, &pd, st.obs_xy))) return rc; v.obs_xy = pd;
      if ((rc = dev_upload(s, &p, st.pt_orig))) return rc; s->d_pt_orig = p;
    }
  }
#undef UP
  const size_t N = (size_t)st.No_pad, NP = (size_t)st.Np_pad;
  const int YS = ys_of(D, DP), AS = a_alloc_of(D, DP, st.has_shared), NS = sym_size(DP);
  const int n_r = st.Nrb * D;
  s->nblocks_slices = (st.nslices + kSlicesPerBlock - 1) / kSlicesPerBlock;
  if (s->nblocks_slices < 1) s->nblocks_slices = 1;
  s->nblocks_points = (st.Np_pad + 255) / 256;
  if (s->nblocks_points < 1) s->nblocks_points = 1;
  v.n_wide = st.n_wide;
  v.n_ultra = st.n_ultra;
  v.n_track_blocks = 16 * st.n_ultra + 4 * (st.n_wide - st.n_ultra) + (st.nslices - st.n_wide + kSlicesPerBlock - 1) / kSlicesPerBlock;
  s->nblocks_tracks = std::max(v.n_track_blocks, 1);
  const int nbmax = std::max(std::max(s->nblocks_slices, s->nblocks_tracks), s->nblocks_points);
#define AL(ptr, n) if ((rc = dev_alloc(s, &ptr, (size_t)(n)))) return rc;
  // the prepared camera records serve the side kernels of a light handle too (their scale part is
  // only meaningful inside a solve)
  AL(v.scale_cam, (size_t)std::max(st.Nc, 1) * 16)
  AL(v.prep, (size_t)std::max(st.Nc, 1) * kPrepStride) AL(v.prep_c, (size_t)std::max(st.Nc, 1) * kPrepStride)
  TMI_HIP(hipMemsetAsync(v.scale_cam, 0, (size_t)std::max(st.Nc, 1) * 16 * sizeof(double), s->stream));
  if (light) {
    TMI_HIP(hipStreamSynchronize(s->stream));
    s->setup_seconds = now_s() - t0;
    return TMI_BA_OK;
  }
  AL(v.pm_r, 2 * N) AL(v.pm_A, 2 * D * N) AL(v.pm_Jp, 2 * DP * N)
  {
    // work arrays of the matrix-free product: zhat (4 doubles per track) + the slot -> track index
    const bool mf = s->implicit || s->adaptive;
    AL(s->d_pm_u, mf ? 4 * NP : 1)
    AL(s->d_cm_t, 1)
    s->need_slot_track = mf;
  }
  AL(v.pm_A1, st.has_shared ? 2 * D * N : 1) AL(v.cam_part, (size_t)std::max(st.Ncam_rb, 1) * (2 * D * D + 3 * D))
  // Y records: the shared-block sums need them; without shared blocks the Schur complement works from the
  // [A | Q] records and Y exists only for the A/B switches that select the older kernels
  if (st.has_shared) {
    // a shared block with more records than one wavefront should walk alone: the chunked raw diagonal (kernels.h)
    int64_t most = 0;
    for (int rb = st.Ncam_rb; rb < st.Nrb; ++rb) most = std::max<int64_t>(most, (int64_t)st.cam_ptr[rb + 1] - st.cam_ptr[rb]);
    if (most > 2 * kSharedDiagChunk) {
      s->shared_diag_chunks = (int)((most + kSharedDiagChunk - 1) / kSharedDiagChunk);
      AL(s->d_shared_diag_partial, (size_t)(st.Nrb - st.Ncam_rb) * s->shared_diag_chunks * sym_size(D))
    }
  }
  // (TMI_BA_SCHUR_Y: the Y-record kernel on a problem without shared blocks -- how tests/test_gpu_parity.py checks it
  // against the [A | Q] kernel on the same matrix)
  s->y_records = st.has_shared || getenv("TMI_BA_SCHUR_Y") != nullptr;
  v.write_y = (s->y_records && (!s->implicit || st.has_shared)) ? 1 : 0;
  AL(v.cm_Y, v.write_y ? (size_t)std::max<int64_t>(st.Nslots, 1) * YS : 1) AL(v.cm_A, (size_t)std::max<int64_t>(st.Nslots, 1) * AS)
  v.cm_R = v.cm_A + (size_t)std::max<int64_t>(st.Nslots, 1) * asa_of(D, DP);  // tails behind the [A | Q] records (!has_shared)
  AL(v.scale_c, std::max(n_r, 1)) AL(v.scale_p, NP * DP)
  AL(v.Vinv, NP * NS) AL(v.Linv, NP * NS) AL(v.gp, NP * DP) AL(v.Vraw, NP * NS) AL(v.yp, NP * DP)
  AL(v.red, s->RL.total) AL(v.Sdiag, (size_t)std::max(st.Nrb, 1) * D * D)
  AL(v.tbuf, (size_t)std::max<int64_t>(st.nub, 1) * D) AL(v.rbuf, (size_t)std::max(v.n_spc, 1) * D) AL(v.Minv, (size_t)std::max(st.Nrb, 1) * D * D)
  AL(v.rhs, std::max(n_r, 1)) AL(v.yc, std::max(n_r, 1)) AL(v.cg_r, std::max(n_r, 1))
  AL(v.cg_z, std::max(n_r, 1)) AL(v.cg_p, std::max(n_r, 1)) AL(v.cg_q, n_r + 8)
  AL(v.dotbuf, st.Nrb + 8) AL(v.ticket, 4 * kTicketStride) AL(v.pcg_done, 1)
  AL(v.cg_t, std::max(n_r, 1)) AL(v.partial, (size_t)4 * std::max(nbmax, (st.Nrb + 3) / 4 + 1)) AL(s->d_partial_max, nbmax)
  AL(v.scal, SC_COUNT) AL(v.flags, FL_COUNT)
#undef AL
  TMI_HIP(hipMemsetAsync(v.scal, 0, SC_COUNT * sizeof(double), s->stream));
  TMI_HIP(hipMemsetAsync(v.flags, 0, FL_COUNT * sizeof(int), s->stream));
  TMI_HIP(hipMemsetAsync(v.ticket, 0, 4 * kTicketStride * sizeof(int), s->stream));
  TMI_HIP(hipMemsetAsync(v.pcg_done, 0, sizeof(int), s->stream));
  TMI_HIP(hipMemsetAsync(v.cg_q, 0, (size_t)(n_r + 8) * sizeof(double), s->stream));
  TMI_HIP(hipMemsetAsync(v.yc, 0, std::max(n_r, 1) * sizeof(double), s->stream));
  TMI_HIP(hipMemsetAsync(v.red, 0, s->RL.total * sizeof(double), s->stream));
  TMI_HIP(hipMemsetAsync(v.cm_Y, 0, (v.write_y ? (size_t)std::max<int64_t>(st.Nslots, 1) * YS : 1) * sizeof(double), s->stream));
  TMI_HIP(hipMemsetAsync(v.cm_A, 0, (size_t)std::max<int64_t>(st.Nslots, 1) * AS * sizeof(double), s->stream));
  {
    int* orb;
    if ((rc = dev_alloc(s, &orb, (size_t)std::max<int64_t>(st.No_pad, 1)))) return rc;
    if (st.No_pad > 0)
      hipLaunchKernelGGL(obs_rb_kernel, dim3((unsigned)((st.No_pad + 255) / 256)), dim3(256), 0, s->stream, v.obs_cam, v.cam_rb,
                         st.Nc, (long long)st.No_pad, orb);
    v.obs_rb = orb;
  }
  if (s->need_slot_track && !st.has_shared) {
    rc = build_mf_chunks(s);
    if (rc) return rc;
  }
  // drop_pos (device_view.h): the position columns of the A planes formed from Jp instead of stored.  Only where the
  // identity holds for every observation and every reader of the planes knows it: the two-pass matrix-free kernels (which
  // read stored columns) cannot run, fp64 evaluation (fp32 rounds -w M and M separately), every camera block has
  // its position free (then its first three columns are the position) and no point is constant (its Jp is stored as
  // zero).  TMI_BA_DROP_POS=0 keeps the columns (A/B, tests).
  v.drop_pos = 0;
  // (the two-pass matrix-free kernels are the only other readers of the A planes: fine when they cannot run -- the
  // one-sweep product is built, or the operator is always the formed S / an exact solver)
  const bool two_pass_possible = (s->implicit || s->adaptive) && !s->mf_ok;
  if (!light && !two_pass_possible && !st.has_shared && O->residual_precision != 32 && D >= 3 && s->DP >= 3) {
    const char* e = getenv("TMI_BA_DROP_POS");
    bool ok = !(e && atoi(e) == 0);
    for (int rb = 0; rb < st.Nrb && ok; ++rb) {
      const int c = st.rb_cam[rb];
      ok = c >= 0 && (st.cam_mask[c] & 7u) == 7u;
    }
    for (int lp = 0; lp < st.Np_pad && ok; ++lp) ok = st.pt_const[lp] == 0;
    if (ok) {
      if ((rc = dev_alloc(s, &v.pos_coef, (size_t)3 * std::max(st.Np_pad, 1)))) return rc;
      if ((rc = dev_alloc(s, &v.cp_trk, (size_t)7 * std::max(st.Np_pad, 1)))) return rc;
      if ((rc = dev_alloc(s, &v.xs, (size_t)std::max(st.Nrb, 1) * D + 8))) return rc;
      if ((rc = dev_alloc(s, &v.xz, (size_t)std::max(st.Nrb, 1) * D + 8))) return rc;
      v.drop_pos = 1;
    }
  }
  if (setup_timing) fprintf(stderr, "[tmi_ba setup] position columns of the A planes: %s\n", v.drop_pos ? "formed from Jp" : "stored");
  // direct_diag.h: matrix-free iterations of the one-sweep product build the camera side without camera-major records
  // (TMI_BA_DIRECT_DIAG=0 keeps the records: A/B, tests)
  s->direct_ok = false;
  {
    const char* env = getenv("TMI_BA_FUSED_FINISH");
    s->fuse_finish_ok = !(env && env[0] == '0');
    env = getenv("TMI_BA_ATTACH_EVENTS");
    s->attach_events = !(env && env[0] == '0');
    env = getenv("TMI_BA_PCG_SPECULATE");
    s->pcg_speculate = !(env && env[0] == '0');
    env = getenv("TMI_BA_FAST_START");
    s->fast_start_ok = !(env && env[0] == '0');
    env = getenv("TMI_BA_COMPACT_PLANES");
    s->compact_env = !(env && env[0] == '0');
    env = getenv("TMI_BA_FUSE_TRACK_SUMS");
    s->fuse_sums = !(env && env[0] == '0');
    env = getenv("TMI_BA_COMPACT_ROBUST");
    s->compact_robust = !(env && env[0] == '0');
  }
  v.direct_diag = 0;
  {
    const char* e = getenv("TMI_BA_DIRECT_DIAG");
    if (s->mf_ok && !st.has_shared && O->residual_precision != 32 && !s->cluster_blocks && st.Nrb > 0 && st.Nslots > 0 && D <= ddg::kMaxD &&
        !(e && atoi(e) == 0)) {
      std::vector<int> cp((size_t)st.Nrb + 1);
      TMI_HIP(hipMemcpyAsync(cp.data(), v.cam_ptr, cp.size() * sizeof(int), hipMemcpyDeviceToHost, s->stream));
      TMI_HIP(hipStreamSynchronize(s->stream));
      std::vector<int> crb, cs0, cs1, rbc((size_t)st.Nrb + 1, 0);
      for (int rb = 0; rb < st.Nrb; ++rb) {
        rbc[rb] = (int)crb.size();
        const int n = cp[rb + 1] - cp[rb];
        if (n <= 0) continue;
        // equal shares of whole trips
        const int nch = (n + ddg::kChunkSlots - 1) / ddg::kChunkSlots;
        const int per = (((n + nch - 1) / nch) + 63) & ~63;
        for (int b = cp[rb]; b < cp[rb + 1]; b += per) {
          crb.push_back(rb);
          cs0.push_back(b);
          cs1.push_back(std::min(b + per, cp[rb + 1]));
        }
      }
      rbc[st.Nrb] = (int)crb.size();
      int *p0, *p1, *p2, *p3, *stt;
      double *xy, *part;
      if ((rc = dev_upload(s, &p0, crb))) return rc;
      if ((rc = dev_upload(s, &p1, cs0))) return rc;
      if ((rc = dev_upload(s, &p2, cs1))) return rc;
      if ((rc = dev_upload(s, &p3, rbc))) return rc;
      if ((rc = dev_alloc(s, &xy, (size_t)2 * st.Nslots))) return rc;
      if ((rc = dev_alloc(s, &part, std::max<size_t>(crb.size(), 1) * ddg::n_acc(D)))) return rc;
      if ((rc = dev_alloc(s, &v.trk_rec, (size_t)std::max(st.Np_pad, 1) * ddg::trk_stride(s->DP)))) return rc;
      if ((rc = dev_alloc(s, &stt, (size_t)st.Nslots))) return rc;
      hipLaunchKernelGGL(ddg::slot_gather_kernel, dim3(s->nblocks_tracks), dim3(256), 0, s->stream, v, stt, xy);
      v.slot_track = stt;
      s->dd.n_chunks = (int)crb.size();
      s->dd.chunk_rb = p0;
      s->dd.chunk_s0 = p1;
      s->dd.chunk_s1 = p2;
      s->dd.rb_chunk = p3;
      s->dd.cm_xy = xy;
      s->dd.part = part;
      s->direct_ok = true;
      // the trial cost view by view (ddg::cost_view_kernel): every observation must own a camera-major slot (no fully
      // constant camera) and the per-workgroup partial sums must fit where cost_kernel leaves its own
      const char* ec = getenv("TMI_BA_COST_BY_VIEW");
      if (const char* ew = getenv("TMI_BA_COST_WARM")) s->cost_warm = atoi(ew) != 0;
      s->cost_by_view = st.Nslots == st.No && ((int)crb.size() + 3) / 4 <= nbmax && !(ec && atoi(ec) == 0);
    }
  }
  if (s->adaptive) {
    // Cost model of schur_mode auto, constants measured on MI355X (profiles/r02_z, r04, r05): forming S ~61 ps per pair
    // (79 with 4-dof points), a product with S ~195 ps per upper block, a matrix-free product ~85 ps per observation in
    // two passes and 52 (57 with 4-dof points) in one sweep.  An iteration that forms S also writes and reads the
    // camera-major records, which a matrix-free one with the direct camera side (direct_diag.h) does not: ~80 ps per
    // observation (point_eliminate 0.41 against 0.11 ms, camera_diag 0.32 against 0.22 ms at Venice size).  Round 4
    // priced the one-sweep product at 70 ps (what measured best then: TMI_BA_BREAK_EVEN = 8 / 16 / 24 / 32 / 48 gave
    // 10.67 / 10.58 / 10.66 / 10.77 / 10.91 ms per LM iteration at the reference's default options); with the records
    // gone from the matrix-free iterations 15 / 32 / 64 / 128 give 9.96 / 9.56 / 9.14 / 9.17 ms.
    const double form = (s->DP == 4 ? 79.0 : 61.0) * (double)st.npairs + (s->direct_ok ? 80.0 * (double)st.No : 0.0),
                 with_s = 195.0 * (double)st.nub,
                 free = (s->mf_ok ? (s->DP == 4 ? 57.0 : 52.0) : 85.0) * (double)st.No;
    if (const char* e = getenv("TMI_BA_BREAK_EVEN")) s->adaptive_break_even_override = atoi(e);
    // (a product with S that costs more than a matrix-free one -- many views, little co-visibility: S has more
    // blocks than there are observations to walk -- never pays off: always matrix-free)
    s->adaptive_break_even = free > with_s ? (int)std::min(1.0e6, form / (free - with_s)) : 1 << 30;
    if (s->adaptive_break_even_override >= 0) s->adaptive_break_even = s->adaptive_break_even_override;
  }
  if (setup_timing) fprintf(stderr, "[tmi_ba setup] camera side of matrix-free iterations: %s\n", s->direct_ok ? "view by view from the track records (no camera-major records)" : "camera-major records");
  if (s->vis_clusters && (rc = build_visibility_clusters(s, P, O->visibility_clustering_type))) return rc;
  if (s->tri) {
    // base clusters: the ones CLUSTER_JACOBI would use on this problem; everything else is a cluster of its own
    std::vector<int> cl_of((size_t)st.Nrb, -1);
    if (st.has_shared) {
      for (int g = 0; g < st.Nrb - st.Ncam_rb; ++g) {
        for (int k = st.grp_cam_ptr[g]; k < st.grp_cam_ptr[g + 1]; ++k) {
          const int rb = st.cam_rb[st.grp_cams[k]];
          if (rb >= 0) cl_of[rb] = g;
        }
        cl_of[st.Ncam_rb + g] = g;
      }
    } else {
      cl_of = s->vis_cluster_of_rb;
      cl_of.resize((size_t)st.Nrb, -1);
    }
    s->tri_seg = chains::build(cl_of, st.rb_dim, P->num_observations, P->obs_camera, st.cam_rb, P->obs_point, P->num_points,
                               P->point_constant, TMI_BA_MAX_CLUSTER_DIM);
    if (setup_timing) {
      size_t big = 0, most = 0;
      for (size_t c = 0; c < s->tri_seg.members.size(); ++c) {
        big = std::max(big, s->tri_seg.members[c].size());
        most = std::max<size_t>(most, (size_t)s->tri_seg.ordinal[c].back() + 1);
      }
      fprintf(stderr, "[tmi_ba setup] CLUSTER_TRIDIAGONAL: %zu segments, the largest %zu reduced blocks, the longest %zu clusters\n",
              s->tri_seg.members.size(), big, most);
    }
  }
  if (s->need_slot_track && !s->mf_ok) {
    int* stt;
    if ((rc = dev_alloc(s, &stt, (size_t)std::max<int64_t>(st.Nslots, 1)))) return rc;
    hipLaunchKernelGGL(slot_track_kernel, dim3(s->nblocks_tracks), dim3(256), 0, s->stream, v, stt);
    v.slot_track = stt;
  }
  TMI_HIP(hipStreamSynchronize(s->stream));
  s->setup_seconds = now_s() - t0;
  if (setup_timing) fprintf(stderr, "[tmi_ba setup] %-28s %.3f s\n", "create total (incl. upload)", s->setup_seconds);
  return TMI_BA_OK;
}

int32_t tmi_ba_solver_create(const tmi_ba_problem* P, const tmi_ba_options* O, int32_t rank,
                             int32_t world, tmi_ba_solver** out) {
  if (!out) return TMI_BA_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!P || !O) return TMI_BA_ERR_INVALID_ARGUMENT;
  tmi_ba_solver* s = new tmi_ba_solver();
  const int rc = create_impl(s, P, O, rank, world);
  if (rc != TMI_BA_OK) {
    if (O->verbose) fprintf(stderr, "[tmi_ba] create failed: %s\n", s->error.c_str());
    g_last_error = s->error;
    tmi_ba_solver_destroy(s);
    return rc;
  }
  *out = s;
  return TMI_BA_OK;
}

int32_t tmi_ba_solver_set_allreduce(tmi_ba_solver* s, tmi_ba_allreduce_fn fn, void* user) {
  if (!s) return TMI_BA_ERR_INVALID_ARGUMENT;
  s->allreduce = fn;
  s->allreduce_user = user;
  return TMI_BA_OK;
}

void* tmi_ba_solver_stream(tmi_ba_solver* s) { return s ? (void*)s->stream : nullptr; }

int32_t tmi_ba_rccl_unique_id(uint8_t id[128]) {
  if (!id) return TMI_BA_ERR_INVALID_ARGUMENT;
  Rccl& r = rccl();
  if (!r.ok) {
    g_last_error = r.error;
    return TMI_BA_ERR_COLLECTIVE;
  }
  const int rc = r.GetUniqueId((void*)id);
  if (rc != 0) {
    g_last_error = "ncclGetUniqueId failed";
    return TMI_BA_ERR_COLLECTIVE;
  }
  return TMI_BA_OK;
}

int32_t tmi_ba_solver_init_rccl(tmi_ba_solver* s, const uint8_t id[128]) {
  if (!s) return TMI_BA_ERR_INVALID_ARGUMENT;
  if (!id) {  // drop the communicator (the caller falls back to its own all-reduce hook)
    if (s->nccl_comm) {
      hipSetDevice(s->device);
      hipStreamSynchronize(s->stream);
      rccl().CommDestroy(s->nccl_comm);
      s->nccl_comm = nullptr;
    }
    return TMI_BA_OK;
  }
  Rccl& r = rccl();
  if (!r.ok) {
    g_last_error = s->error = r.error;
    return TMI_BA_ERR_COLLECTIVE;
  }
  if (hipSetDevice(s->device) != hipSuccess) return TMI_BA_ERR_DEVICE;
  Rccl::Id uid;
  memcpy(uid.b, id, 128);
  void* comm = nullptr;
  const int rc = r.CommInitRank(&comm, s->st.world, uid, s->st.rank);
  if (rc != 0 || !comm) {
    g_last_error = s->error = std::string("ncclCommInitRank failed: ") +
                              (r.GetErrorString ? r.GetErrorString(rc) : "error");
    return TMI_BA_ERR_COLLECTIVE;
  }
  s->nccl_comm = comm;
  return TMI_BA_OK;
}

int32_t tmi_ba_solver_debug_allreduce(tmi_ba_solver* s, double value, double* out) {
  if (!s || !out) return TMI_BA_ERR_INVALID_ARGUMENT;
  if (hipSetDevice(s->device) != hipSuccess) return TMI_BA_ERR_DEVICE;
  double h[8];
  for (double& x : h) x = value;
  double* d = s->v.red + s->RL.scalars;
  if (hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, s->stream) != hipSuccess) return TMI_BA_ERR_DEVICE;
  const int rc = transport_allreduce(s, d, 8);
  if (rc) {
    g_last_error = s->error;
    return rc;
  }
  if (hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, s->stream) != hipSuccess) return TMI_BA_ERR_DEVICE;
  if (hipStreamSynchronize(s->stream) != hipSuccess) return TMI_BA_ERR_DEVICE;
  *out = h[0];
  return TMI_BA_OK;
}

int32_t tmi_ba_solver_reset(tmi_ba_solver* s) {
  if (!s) return TMI_BA_ERR_INVALID_ARGUMENT;
  TMI_HIP(hipSetDevice(s->device));
  // from the resident copy when there is one (a device-to-device copy: microseconds)
  const hipMemcpyKind kind = s->d_ext0 ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (!s->ext0.empty())
    TMI_HIP(hipMemcpyAsync(s->v.ext, s->d_ext0 ? s->d_ext0 : s->ext0.data(), s->ext0.size() * sizeof(double), kind, s->stream));
  if (!s->intr0.empty())
    TMI_HIP(hipMemcpyAsync(s->v.intr, s->d_intr0 ? s->d_intr0 : s->intr0.data(), s->intr0.size() * sizeof(double), kind, s->stream));
  if (!s->pts0.empty())
    TMI_HIP(hipMemcpyAsync(s->v.pts, s->d_pts0 ? s->d_pts0 : s->pts0.data(), s->pts0.size() * sizeof(double), kind, s->stream));
  TMI_HIP(hipStreamSynchronize(s->stream));
  return TMI_BA_OK;
}

// New parameter values for the resident problem (same structure): what `reset` restores from then on.
int32_t tmi_ba_solver_set_parameters(tmi_ba_solver* s, const tmi_ba_problem* P) {
  if (!s || !P) return TMI_BA_ERR_INVALID_ARGUMENT;
  const Structure& st = s->st;
  if (P->num_cameras != st.Nc || P->num_points != st.Np_total || P->num_groups != st.G ||
      (st.G && P->group_offset[st.G] != s->n_intr) || (st.Nc && !P->extrinsics) || (st.Np_total && !P->points) ||
      (s->n_intr && !P->intrinsics)) {
    s->error = "set_parameters: the problem does not have the shape the solver was created with";
    return TMI_BA_ERR_INVALID_ARGUMENT;
  }
  TMI_HIP(hipSetDevice(s->device));
  if (st.Nc) s->ext0.assign(P->extrinsics, P->extrinsics + (size_t)6 * st.Nc);
  if (s->n_intr) s->intr0.assign(P->intrinsics, P->intrinsics + s->n_intr);
  for (int lp = 0; lp < st.Np_pad; ++lp) {
    const int p = st.pt_orig[lp];
    if (p < 0) continue;
    for (int a = 0; a < 4; ++a) s->pts0[(size_t)4 * lp + a] = P->points[(size_t)4 * p + a];
  }
  if (s->d_ext0) {
    if (!s->ext0.empty())
      TMI_HIP(hipMemcpyAsync(s->d_ext0, s->ext0.data(), s->ext0.size() * sizeof(double), hipMemcpyHostToDevice, s->stream));
    if (!s->intr0.empty())
      TMI_HIP(hipMemcpyAsync(s->d_intr0, s->intr0.data(), s->intr0.size() * sizeof(double), hipMemcpyHostToDevice, s->stream));
    if (!s->pts0.empty())
      TMI_HIP(hipMemcpyAsync(s->d_pts0, s->pts0.data(), s->pts0.size() * sizeof(double), hipMemcpyHostToDevice, s->stream));
  }
  return tmi_ba_solver_reset(s);
}

int32_t tmi_ba_solver_download(tmi_ba_solver* s, tmi_ba_problem* P) {
  if (!s || !P) return TMI_BA_ERR_INVALID_ARGUMENT;
  if (P->num_cameras != s->st.Nc || P->num_points != s->st.Np_total) return TMI_BA_ERR_INVALID_ARGUMENT;
  TMI_HIP(hipSetDevice(s->device));
  std::vector<double> pts((size_t)4 * s->st.Np_pad);
  TMI_HIP(hipMemcpyAsync(P->extrinsics, s->v.ext, (size_t)6 * s->st.Nc * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  if (s->n_intr)
    TMI_HIP(hipMemcpyAsync(P->intrinsics, s->v.intr, (size_t)s->n_intr * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  if (!pts.empty())
    TMI_HIP(hipMemcpyAsync(pts.data(), s->v.pts, pts.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  TMI_HIP(hipStreamSynchronize(s->stream));
  for (int lp = 0; lp < s->st.Np_pad; ++lp) {
    const int p = s->st.pt_orig[lp];
    if (p < 0) continue;
    for (int a = 0; a < 4; ++a) P->points[(size_t)4 * p + a] = pts[(size_t)4 * lp + a];
  }
  return TMI_BA_OK;
}

// prepared camera records of a parameter set (camera_models.h); must follow every change of
// the extrinsics / intrinsics it is taken from and every change of the column scales
static void prepare_cameras(tmi_ba_solver* s, const double* ext, const double* intr, double* prep) {
  if (s->v.Nc)
    hipLaunchKernelGGL(camera_prepare_kernel, dim3((s->v.Nc + 255) / 256), dim3(256), 0, s->stream, s->v, ext,
                       intr, prep);
}

// ---- the linear solve of one LM iteration ------------------------------------------
// returns TMI_BA_OK; *usable = 0 for LINEAR_SOLVER_FAILURE
// q = S x: explicit (symmetric block SpMV on the formed Schur complement) or implicit
// (two passes over the observations; the reduced vector is all-reduced across ranks)
static int apply_schur(tmi_ba_solver* s, const double* x, double* y, int dot = 0, int xs_ready = 0, const int* guard = nullptr) {
  DeviceView& v = s->v;
  const int n = v.Nrb * v.D;
  if (!s->implicit_now) {
    Timed t(s, TMI_BA_K_SPMV);
    s->launch.spmv(v, s->stream, v.red + s->RL.ub, x, y, dot);
    return TMI_BA_OK;
  }
  {
    Timed t(s, TMI_BA_K_SPMV, /*attach=*/s->mf_ok && s->attach_events);
    const tmi_ba_options* O = s->cur_opts;
    const int add_diag = (s->st.world <= 1 || s->st.rank == 0) ? 1 : 0;
    if (s->mf_ok)
      s->launch.mf_product(v, s->mf, s->stream, s->RL, x, y, s->cur_inv_radius, O->min_lm_diagonal, O->max_lm_diagonal,
                           add_diag, dot, xs_ready, t.start(), t.stop(), guard);
    else
      s->launch.implicit_spmv(v, s->stream, s->RL, x, y, s->d_pm_u, s->d_cm_t, s->cur_inv_radius,
                              O->min_lm_diagonal, O->max_lm_diagonal, add_diag, s->nblocks_tracks, dot);
  }
  // with the dot product fused, x . y (this rank's share) rides behind the vector
  return do_allreduce(s, y, n + (dot ? 1 : 0));
}

// Wait until slot `slot` of the host mirror shows sequence number `seq` (published by pcg_step).
static int wait_mirror(tmi_ba_solver* s, int slot, unsigned long long seq) {
  volatile unsigned long long* p = &s->h_mirror[slot].seq;
  for (unsigned spin = 1;; ++spin) {
    if (*p == seq) break;
    if ((spin & 0xfffu) == 0) {
      const hipError_t q = hipStreamQuery(s->stream);
      if (q == hipSuccess) {
        if (*p == seq) break;
        // an idle-looking stream can still have launches in the runtime's submission batch: drain
        // for real before calling the missing sequence number an error
        if (hipStreamSynchronize(s->stream) != hipSuccess || *p != seq) {
          char buf[160];
          snprintf(buf, sizeof(buf), "PCG step did not publish its scalars (expected sequence %llu, mirror holds %llu)",
                   seq, (unsigned long long)*p);
          s->error = buf;
          return TMI_BA_ERR_DEVICE;
        }
        break;
      }
      if (q != hipErrorNotReady) {
        s->error = std::string("hipStreamQuery: ") + hipGetErrorString(q);
        return TMI_BA_ERR_DEVICE;
      }
    }
    __builtin_ia32_pause();
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  select_mirror(s, slot);
  const hipError_t le = hipGetLastError();
  if (le != hipSuccess) {
    s->error = std::string("kernel launch failed: ") + hipGetErrorString(le);
    return TMI_BA_ERR_DEVICE;
  }
  return TMI_BA_OK;
}

// big dataflow launches of one process on one device run one after the other: two of them racing for the same CUs could
// each hold a part of the device and wait for the rest (dense_cholesky_df.h); the dependency is enqueued, the host
// does not wait
static std::mutex g_df_mutex;
static hipEvent_t g_df_event[64] = {};
static int launch_coresident(tmi_ba_solver* s, int grid, const std::function<void()>& launch) {
  if (!s->num_cus) {
    hipDeviceProp_t prop;
    TMI_HIP(hipGetDeviceProperties(&prop, s->device));
    s->num_cus = prop.multiProcessorCount;
  }
  const bool big = 2 * grid > s->num_cus && s->device >= 0 && s->device < 64;
  if (!big) {
    launch();
    return TMI_BA_OK;
  }
  std::lock_guard<std::mutex> lock(g_df_mutex);
  hipEvent_t& ev = g_df_event[s->device];
  if (ev) {
    TMI_HIP(hipStreamWaitEvent(s->stream, ev, 0));
  } else {
    TMI_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  launch();
  TMI_HIP(hipEventRecord(ev, s->stream));
  return TMI_BA_OK;
}

// ---- CLUSTER_JACOBI over the shared intrinsics blocks (cluster_precond.h) ------------------------------------
static int ensure_clusters(tmi_ba_solver* s) {
  if (s->cl_built) return TMI_BA_OK;
  const Structure& st = s->st;
  if (!s->num_cus) {
    hipDeviceProp_t prop;
    TMI_HIP(hipGetDeviceProperties(&prop, s->device));
    s->num_cus = prop.multiProcessorCount;
  }
  std::vector<std::vector<int> > members = s->tri ? s->tri_seg.members : s->vis_members;  // (empty unless vis_clusters / tri)
  for (int g = 0; g < st.Nrb - st.Ncam_rb && !s->vis_clusters && !s->tri; ++g) {
    std::vector<int> m;
    for (int k = st.grp_cam_ptr[g]; k < st.grp_cam_ptr[g + 1]; ++k) m.push_back(st.cam_rb[st.grp_cams[k]]);
    std::sort(m.begin(), m.end());
    m.push_back(st.Ncam_rb + g);
    members.push_back(m);
  }
  // a cluster is factored densely: oversized ones keep their SCHUR_JACOBI blocks (TMI_BA_MAX_CLUSTER_DIM, the oracle
  // applies the same rule)
  members.erase(std::remove_if(members.begin(), members.end(),
                               [&](const std::vector<int>& m) {
                                 long long n = 0;
                                 for (const int rb : m) n += st.rb_dim[rb];
                                 return n > TMI_BA_MAX_CLUSTER_DIM;
                               }),
                members.end());
  auto lookup = [&](int bi, int bj) -> int {
    const int* b = st.ub_j.data() + st.urow_ptr[bi];
    const int* e = st.ub_j.data() + st.urow_ptr[bi + 1];
    const int* it = std::lower_bound(b, e, bj);
    return (it != e && *it == bj) ? (int)(it - st.ub_j.data()) : -1;
  };
  // (tridiagonal: only the blocks inside a cluster and between neighbours of the chain are gathered, the latter marked)
  s->cl_plan = clp::make_plan(members, st.rb_dim, st.D, lookup, s->num_cus, s->tri ? &s->tri_seg.ordinal : nullptr);
  const clp::Plan& p = s->cl_plan;
  int rc;
  if ((rc = dev_upload(s, &s->d_cl_desc, p.desc))) return rc;
  if ((rc = dev_upload(s, &s->d_cl_ge, p.entries))) return rc;
  if ((rc = dev_upload(s, &s->d_cl_idx, p.idx))) return rc;
  if ((rc = dev_alloc(s, &s->d_cl_tiles, (size_t)p.n_tiles * cdf::TILE))) return rc;
  if ((rc = dev_alloc(s, &s->d_cl_linv, (size_t)p.n_linv * cdf::TILE))) return rc;
  if ((rc = dev_alloc(s, &s->d_cl_vec, (size_t)std::max(p.n_vec, 1)))) return rc;
  if ((rc = dev_alloc(s, &s->d_cl_flags, (size_t)std::max(p.n_flags, 1)))) return rc;
  if ((rc = dev_alloc(s, &s->d_cl_bad, (size_t)std::max(p.ncl, 1)))) return rc;
  TMI_HIP(hipMemsetAsync(s->d_cl_flags, 0, (size_t)std::max(p.n_flags, 1) * sizeof(int), s->stream));
  s->cl_epoch = 0;
  s->cl_built = true;
  return TMI_BA_OK;
}

// the clusters' matrices from the blocks of S (diagonal blocks with the LM diagonal already added), factored
static int factor_clusters(tmi_ba_solver* s) {
  int rc = ensure_clusters(s);
  if (rc) return rc;
  const clp::Plan& p = s->cl_plan;
  if (p.ncl == 0) return TMI_BA_OK;
  DeviceView& v = s->v;
  s->cl_failed = false;
  // CLUSTER_TRIDIAGONAL: a chain's matrix lacks the blocks between clusters that are not neighbours, which can cost
  // positive definiteness.  Ceres (VisibilityBasedPreconditioner::UpdateImpl) then halves every off-diagonal
  // cluster-pair cell and factors once more; a second failure fails the preconditioner update and the linear solve.
  for (int attempt = 0; attempt < (s->tri ? 2 : 1); ++attempt) {
    TMI_HIP(hipMemsetAsync(s->d_cl_tiles, 0, (size_t)p.n_tiles * cdf::TILE * sizeof(double), s->stream));
    TMI_HIP(hipMemsetAsync(s->d_cl_bad, 0, (size_t)p.ncl * sizeof(int), s->stream));
    // (test hooks, read by the oracle as well: TMI_BA_TEST_TRI_SCALE0 / _SCALE1 replace the factors 1 and 1/2 of the two
    // attempts, so that tests/test_cluster_jacobi.py can drive the retry and the failure path on both sides)
    double off_scale = attempt ? 0.5 : 1.0;
    if (const char* e = getenv(attempt ? "TMI_BA_TEST_TRI_SCALE1" : "TMI_BA_TEST_TRI_SCALE0")) off_scale = atof(e);
    s->launch.cluster_gather(s->stream, s->d_cl_ge, (int)p.entries.size(), s->d_cl_desc, v.red + s->RL.ub, v.Sdiag, s->d_cl_tiles,
                             off_scale);
    if (s->cl_epoch >= 0x7ffffff0) {  // flags compare against the epoch: start over long before it wraps (as df_epoch)
      TMI_HIP(hipMemsetAsync(s->d_cl_flags, 0, (size_t)std::max(p.n_flags, 1) * sizeof(int), s->stream));
      s->cl_epoch = 0;
    }
    const int epoch = ++s->cl_epoch;
    rc = launch_coresident(s, p.grid, [&] {
      hipLaunchKernelGGL(clp::cluster_factor_kernel, dim3(p.grid), dim3(256), 0, s->stream, s->d_cl_desc, p.ncl, s->d_cl_tiles,
                         s->d_cl_linv, s->d_cl_flags, epoch, v.flags + FL_CHOL_ABORT, s->d_cl_bad);
    });
    if (rc || !s->tri) return rc;
    // (one small read-back per LM iteration: the retry is a host decision, as in Ceres)
    std::vector<int> bad((size_t)p.ncl);
    TMI_HIP(hipMemcpyAsync(bad.data(), s->d_cl_bad, (size_t)p.ncl * sizeof(int), hipMemcpyDeviceToHost, s->stream));
    TMI_HIP(hipStreamSynchronize(s->stream));
    bool any = false;
    for (const int b : bad) any = any || b != 0;
    if (!any) return TMI_BA_OK;
  }
  s->cl_failed = true;  // "Preconditioner update failed.": LINEAR_SOLVER_FAILURE, the step is invalid
  return TMI_BA_OK;
}

// z <- C^-1 r on the clustered entries (the others keep their block-Jacobi values)
static int apply_clusters(tmi_ba_solver* s, const double* r, double* z) {
  const clp::Plan& p = s->cl_plan;
  if (p.ncl == 0) return TMI_BA_OK;
  const int epoch = ++s->cl_epoch;
  return launch_coresident(s, p.grid, [&] {
    hipLaunchKernelGGL(clp::cluster_apply_kernel, dim3(p.grid), dim3(256), 0, s->stream, s->d_cl_desc, p.ncl, s->d_cl_tiles,
                       s->d_cl_linv, s->d_cl_flags, epoch, s->v.flags + FL_CHOL_ABORT, s->d_cl_bad, s->d_cl_idx, s->d_cl_vec, r, z);
  });
}

static inline double st_nub_bytes(const tmi_ba_solver* s) { return 8.0 * (double)s->st.nub * s->st.D * s->st.D; }
static int solve_reduced_pcg(tmi_ba_solver* s, const tmi_ba_options* O, int* usable, int64_t* iters, bool init_done = false) {
  DeviceView& v = s->v;
  const int n = v.Nrb * v.D;
  const double* b = v.red + s->RL.gt;
  *usable = 1;
  if (n == 0) return TMI_BA_OK;
  if (!init_done) {  // (init_done: camera_finish left x, r, z, p and rho)
    // x = 0, r = b, z = M^-1 b, p = z, rho: one multi-workgroup launch
    Timed t(s, TMI_BA_K_PCG_VECTOR);
    s->launch.pcg_init(v, s->stream, b, (v.Nrb + kPcgStepThreads / 64 - 1) / (kPcgStepThreads / 64));
    if (s->cl_active) {
      const int rcc = apply_clusters(s, v.cg_r, v.cg_z);
      if (rcc) return rcc;
      hipLaunchKernelGGL(clp::cluster_init_fix_kernel, dim3(1), dim3(1024), 0, s->stream, v, n);
    }
  }
  // Fused path (no shared intrinsics blocks): an iteration is  product (+ p.q) -> [all-reduce] -> pcg_step (whose last
  // workgroup also forms p for the next product), three launches and one poll of the host mirror.  Every tenth iteration recomputes the residual
  // (residual_reset_period) through the three-kernel path below.  (Enqueueing iteration it + 1 speculatively before
  // the scalars of iteration it are read measured no gain on MI355X -- 5.02 vs 5.04 ms per LM iteration at one GPU,
  // 1.57 vs 1.53 ms for an eighth of the tracks, profiles/r02_g -- and is gone.)
  const bool fused = !s->st.has_shared && !s->cl_active;  // (pcg_step applies the block inverses itself: no clusters)
  // The formed S: the whole solve as ONE persistent launch (pcg_persist.h) -- no launch, drain or host poll per PCG
  // iteration.  TMI_BA_PCG_PERSISTENT=0 keeps the launch-per-step loop below (tests hold the two to each other).
  if (fused && !s->implicit_now && !s->ppcg_off && v.n_spc > 0) {
    // ... while S is small: the grid of one workgroup per CU keeps ~28 KB per CU in flight, enough for a matrix the
    // Infinity Cache holds but half the rate of the launch-per-step kernels (whole-chip occupancy) on the 826 MB of the
    // ring scene (measured: 406 against 235 us per PCG iteration).  TMI_BA_PCG_PERSISTENT=1 forces it, 0 switches it off
    // (read per solve: tests switch it inside one process).
    const char* env = getenv("TMI_BA_PCG_PERSISTENT");
    const bool small_S = (double)st_nub_bytes(s) <= 256.0 * 1024 * 1024;
    if (env ? env[0] != '0' : small_S) {
      if (!s->ppcg_ready) {
        if (!s->num_cus) {
          hipDeviceProp_t prop;
          TMI_HIP(hipGetDeviceProperties(&prop, s->device));
          s->num_cus = prop.multiProcessorCount;
        }
        s->ppcg_grid = std::max(1, s->num_cus);
        // the persistent launch is an optimisation: if its (small) buffers cannot be had, the solve continues on the
        // launch-per-step loop below instead of failing -- clear the sticky HIP error and the message
        if (dev_alloc(s, &s->d_ppcg_p, (size_t)2 * n) || dev_alloc(s, &s->d_ppcg_x, (size_t)n) ||
            dev_alloc(s, &s->d_ppcg_partial, (size_t)3 * s->ppcg_grid) || dev_alloc(s, &s->d_ppcg_bar, (size_t)ppcg::kBarInts)) {
          (void)hipGetLastError();
          s->error.clear();
          s->ppcg_off = true;
          return solve_reduced_pcg(s, O, usable, iters);
        }
        s->ppcg_ready = true;
      }
      int rc0;
      ppcg::Args a;
      a.ub = v.red + s->RL.ub;
      a.b = b;
      a.pbuf = s->d_ppcg_p;
      a.xpub = s->d_ppcg_x;
      a.partial = s->d_ppcg_partial;
      a.bar = s->d_ppcg_bar;
      a.ctrl = v.flags + FL_CHOL_ABORT;
      a.eta = O->eta;
      a.min_it = O->min_linear_solver_iterations;
      a.max_it = O->max_linear_solver_iterations;
      a.red8 = v.red + s->RL.scalars;
      a.mirror = s->d_mirror;
      const unsigned long long my_seq = ++s->mirror_seq;
      a.seq = my_seq;
      a.prof = nullptr;
      static const bool want_prof = getenv("TMI_BA_PPCG_PROF") != nullptr;
      if (want_prof) {
        if (!s->d_ppcg_prof) {
          if ((rc0 = dev_alloc(s, &s->d_ppcg_prof, 48))) return rc0;
          TMI_HIP(hipMemsetAsync(s->d_ppcg_prof, 0, 48 * sizeof(long long), s->stream));
        }
        a.prof = s->d_ppcg_prof;
      }
      TMI_HIP(hipMemsetAsync(s->d_ppcg_bar, 0, ppcg::kBarInts * sizeof(int), s->stream));
      int rc;
      {
        Timed t(s, TMI_BA_K_SPMV);
        rc = launch_coresident(s, s->ppcg_grid, [&] { s->launch.pcg_persistent(v, s->stream, s->ppcg_grid, a); });
      }
      if (rc) return rc;
      if ((rc = wait_mirror(s, 0, my_seq))) return rc;
      if (s->h_flags[FL_CHOL_ABORT]) {
        // the grid could not become co-resident (another process holds part of the device): launch per step from
        // here on, this solve included
        TMI_HIP(hipMemsetAsync(v.flags + FL_CHOL_ABORT, 0, sizeof(int), s->stream));
        s->ppcg_off = true;
        return solve_reduced_pcg(s, O, usable, iters);
      }
      const int done = (int)s->h_scal[ppcg::SC_PCG_IT];
      if (done > 1) s->launches[TMI_BA_K_SPMV] += done - 1;  // the class counts products
      if (s->h_flags[FL_PCG_FAIL]) {
        *usable = 0;
      } else if (!(s->h_scal[SC_PQ] > 0.0)) {
        // LINEAR_SOLVER_NO_CONVERGENCE, x kept
      } else if ((s->h_scal[SC_ZETA] < O->eta && done >= O->min_linear_solver_iterations) ||
                 done >= O->max_linear_solver_iterations) {
      } else if (s->h_scal[SC_RHO_BAD] != 0.0) {
        *usable = 0;
      }
      *iters += done;
      return TMI_BA_OK;
    }
  }
  const int nbv = (v.Nrb + 3) / 4;
  const int nbs16 = (v.Nrb + kPcgStepThreads / 64 - 1) / (kPcgStepThreads / 64);  // workgroups of pcg_step
  // drop_pos: pcg_init and pcg_p leave the scaled copy of p the product gathers; the three-kernel path does not
  bool xs_valid = !s->cl_active;
  // One fused iteration: product (+ p.q) -> [all-reduce] -> pcg_step, which publishes to slot (it & 1) of the host mirror.
  // guard != null: the kernels return at once when the previous step's stopping test held (DeviceView::pcg_done).
  auto enqueue_step = [&](int it_, const int* guard, unsigned long long* seq_out) -> int {
    const int rcs = apply_schur(s, v.cg_p, v.cg_q, /*dot=*/1, xs_valid ? 1 : 0, guard);
    if (rcs) return rcs;
    xs_valid = true;  // (pcg_step leaves the scaled copy of the next p)
    Timed t(s, TMI_BA_K_PCG_VECTOR);
    *seq_out = ++s->mirror_seq;
    s->launch.pcg_step(v, s->stream, b, it_, nbs16, O->eta, O->min_linear_solver_iterations, O->max_linear_solver_iterations,
                       v.red + s->RL.scalars, s->d_mirror + (it_ & 1), *seq_out, guard);
    return TMI_BA_OK;
  };
  // Round 6: iteration it + 1 is enqueued BEFORE the host reads the scalars of iteration it.  The host's round trip
  // (mirror write over PCIe, poll, three launches) left the device idle for ~20 us after every pcg_step
  // (profiles/r06_a_gaps.md); now it sits behind the next product, and a solve that has stopped costs three launches
  // that return at once -- while the host is on that same round trip.  (Round 2 measured no gain from this: an LM
  // iteration took 5 ms then and every PCG iteration was four launches and a poll.)  Only with the one-sweep product
  // (its kernels carry the guard) and never across a residual-reset iteration (another launch sequence).
  // One rank only: a collective enqueued ahead cannot be skipped, a stopped solve would pay one more all-reduce.
  const bool speculate = fused && s->pcg_speculate && s->implicit_now && s->mf_ok && s->st.world == 1;
  bool pending = false;  // iteration `it` is already enqueued (speculatively, during iteration it - 1)
  unsigned long long pending_seq = 0;
  size_t spec_ev_mark = 0;
  long long spec_launches[TMI_BA_NUM_KERNEL_CLASSES];
  // a speculative iteration that found PCG stopped did nothing: its launches and event pairs do not count
  auto void_speculation = [&]() {
    if (!pending) return;
    for (size_t i = spec_ev_mark; i < s->ev_used; ++i) s->events[i].cls = -1;
    for (int c = 0; c < TMI_BA_NUM_KERNEL_CLASSES; ++c) s->launches[c] = spec_launches[c];
    pending = false;
  };
  int it;
  for (it = 1;; ++it) {
    const bool reset = (it % 10 == 0);  // residual_reset_period
    int rc;
    if (fused && !reset) {
      unsigned long long my_seq = pending_seq;
      if (!pending && (rc = enqueue_step(it, nullptr, &my_seq))) return rc;
      pending = false;
      if (speculate && (it + 1) % 10 != 0 && it + 1 <= O->max_linear_solver_iterations) {
        spec_ev_mark = s->ev_used;
        for (int c = 0; c < TMI_BA_NUM_KERNEL_CLASSES; ++c) spec_launches[c] = s->launches[c];
        if ((rc = enqueue_step(it + 1, v.pcg_done, &pending_seq))) return rc;
        pending = true;
      }
      if ((rc = wait_mirror(s, it & 1, my_seq))) return rc;
    } else {
      {
        const int rcs = apply_schur(s, v.cg_p, v.cg_q, 0, xs_valid ? 1 : 0);
        if (rcs) return rcs;
        xs_valid = false;  // (pcg_b3 writes p)
      }
      {
        Timed t(s, TMI_BA_K_PCG_VECTOR);
        hipLaunchKernelGGL(pcg_b1_kernel, dim3(1), dim3(1024), 0, s->stream, v, n);
        s->launch.pcg_b2(v, s->stream, b, reset ? 1 : 0, nbv, v.partial);
        if (s->cl_active && !reset) {
          const int rcc = apply_clusters(s, v.cg_r, v.cg_z);
          if (rcc) return rcc;
          s->launch.cluster_rz(v, s->stream, nbv, v.partial);
        }
      }
      if (reset) {
        const int rcs = apply_schur(s, v.yc, v.cg_t);
        if (rcs) return rcs;
        Timed t(s, TMI_BA_K_PCG_VECTOR);
        s->launch.pcg_b2(v, s->stream, b, 2, nbv, v.partial);
        if (s->cl_active) {
          const int rcc = apply_clusters(s, v.cg_r, v.cg_z);
          if (rcc) return rcc;
          s->launch.cluster_rz(v, s->stream, nbv, v.partial);
        }
      }
      {
        // Q1, zeta, and z / rho / p of iteration it + 1
        Timed t(s, TMI_BA_K_PCG_VECTOR);
        hipLaunchKernelGGL(pcg_b3_kernel, dim3(1), dim3(1024), 0, s->stream, v, n, it, nbv, v.partial);
      }
      rc = readback(s);
      if (rc) return rc;
    }
    {
      static const bool pcg_trace = getenv("TMI_BA_PCG_TRACE") != nullptr;  // (diagnostics: the scalars of every PCG iteration)
      if (pcg_trace)
        fprintf(stderr, "[tmi_ba pcg] it %d %s pq %.17g alpha %.17g Q1 %.17g zeta %.17g rho %.17g\n", it,
                fused && !reset ? "step" : "reset", s->h_scal[SC_PQ], s->h_scal[SC_ALPHA], s->h_scal[SC_Q1], s->h_scal[SC_ZETA],
                s->h_scal[SC_RHO]);
    }
    if (s->h_flags[FL_PCG_FAIL]) {
      *usable = 0;
      break;
    }
    if (pending && s->h_scal[SC_PCG_STOP] != 0.0 && (s->h_scal[SC_PQ] > 0.0) &&
        !((s->h_scal[SC_ZETA] < O->eta && it >= O->min_linear_solver_iterations) || it >= O->max_linear_solver_iterations ||
          s->h_scal[SC_RHO_BAD] != 0.0)) {
      // pcg_step's stopping test and the host's below are the same rules; should they ever disagree (p.q = +inf), the
      // device's answer stands: the speculative step has already returned without publishing
      break;
    }
    if (s->cl_active && s->h_flags[FL_CHOL_ABORT]) {
      // A dataflow launch of the cluster preconditioner could not become co-resident (another process holds part of
      // the device) and gave up: nothing it produced is to be trusted.  As the exact solver does (solve_reduced_dense),
      // fall back instead of failing the solve: clear the flag, retire the clusters for this solve and run this LM
      // iteration's PCG again with the SCHUR_JACOBI blocks (Minv holds them: clusters only override entries of z).
      TMI_HIP(hipMemsetAsync(v.flags + FL_CHOL_ABORT, 0, sizeof(int), s->stream));
      TMI_HIP(hipMemsetAsync(s->d_cl_flags, 0, (size_t)std::max(s->cl_plan.n_flags, 1) * sizeof(int), s->stream));
      s->cl_epoch = 0;
      s->cl_active = false;
      s->cl_retired = true;
      return solve_reduced_pcg(s, O, usable, iters);
    }
    if (!(s->h_scal[SC_PQ] > 0.0)) break;  // LINEAR_SOLVER_NO_CONVERGENCE, x kept
    if (s->h_scal[SC_ZETA] < O->eta && it >= O->min_linear_solver_iterations) break;
    if (it >= O->max_linear_solver_iterations) break;
    if (s->h_scal[SC_RHO_BAD] != 0.0) {  // rho of the next iteration is 0 / inf: FAILURE
      *usable = 0;
      break;
    }
  }
  void_speculation();
  *iters += it;
  return TMI_BA_OK;
}

static int solve_reduced_dense_panels(tmi_ba_solver* s) {
  DeviceView& v = s->v;
  const int n = v.Nrb * v.D;
  if (!s->d_dense) {
    // + the side buffer of the factored diagonal blocks (dense_cholesky.h)
    int rc = dev_alloc(s, &s->d_dense, (size_t)n * n + (size_t)kPanel * (n + kPanel));
    if (rc) return rc;
  }
  TMI_HIP(hipMemsetAsync(s->d_dense, 0, (size_t)n * n * sizeof(double), s->stream));
  s->launch.dense_gather(v, s->stream, v.red + s->RL.ub, s->d_dense, n);
  dense_cholesky_solve(s->d_dense, n, v.red + s->RL.gt, v.yc, v.cg_t, s->d_dense + (size_t)n * n,
                       v.flags + FL_SINGULAR_BLOCK, s->stream);
  return TMI_BA_OK;
}

static int solve_reduced_dense(tmi_ba_solver* s, int* usable) {
  DeviceView& v = s->v;
  const int n = v.Nrb * v.D;
  *usable = 1;
  if (n == 0) return TMI_BA_OK;
  Timed t(s, TMI_BA_K_CHOLESKY);
  // the launch-per-panel Cholesky is the fall-back of the dataflow launch (a launch that cannot become co-resident,
  // FL_CHOL_ABORT below); TMI_BA_CHOL_PANELS selects it outright so that tests/test_gpu_cholesky.py can hold it to
  // the dataflow result
  if (getenv("TMI_BA_CHOL_PANELS") != nullptr) return solve_reduced_dense_panels(s);
  if (!s->num_cus) {
    hipDeviceProp_t prop;
    TMI_HIP(hipGetDeviceProperties(&prop, s->device));
    s->num_cus = prop.multiProcessorCount;
  }
  const cdf::Plan plan = cdf::make_plan(n, s->num_cus);
  if (!s->d_df_tiles) {
    int rc;
    if ((rc = dev_alloc(s, &s->d_df_tiles, plan.tile_doubles))) return rc;
    if ((rc = dev_alloc(s, &s->d_df_linv, plan.linv_doubles))) return rc;
    if ((rc = dev_alloc(s, &s->d_df_flags, plan.flag_ints))) return rc;
    TMI_HIP(hipMemsetAsync(s->d_df_flags, 0, plan.flag_ints * sizeof(int), s->stream));
    s->df_epoch = 0;
  }
  if (++s->df_epoch == 0x7fffffff) {  // flags compare against the epoch: start over long before it wraps
    TMI_HIP(hipMemsetAsync(s->d_df_flags, 0, plan.flag_ints * sizeof(int), s->stream));
    s->df_epoch = 1;
  }
  TMI_HIP(hipMemsetAsync(s->d_df_tiles, 0, plan.tile_doubles * sizeof(double), s->stream));
  s->launch.tile_gather(v, s->stream, v.red + s->RL.ub, v.red + s->RL.gt, s->d_df_tiles, n);
  cdf::Args a;
  const size_t ntiles = cdf::tile_index(plan.T - 1, plan.T - 1) + 1;
  a.tiles = s->d_df_tiles;
  a.linv = s->d_df_linv;
  a.tflag = s->d_df_flags;
  a.dflag = s->d_df_flags + ntiles;
  a.xflag = a.dflag + plan.T;
  a.x = v.yc;
  a.ctrl = v.flags + FL_CHOL_ABORT;
  a.singular = v.flags + FL_SINGULAR_BLOCK;
  a.n = n;
  a.T = plan.T;
  a.epoch = s->df_epoch;
  a.band = plan.band;
  a.team = plan.team;
  a.G = plan.G;
  {
    const int rcl = launch_coresident(s, plan.G, [&] {
      hipLaunchKernelGGL(cdf::chol_dataflow_kernel, dim3(plan.G), dim3(256), 0, s->stream, a);
    });
    if (rcl) return rcl;
  }
  return TMI_BA_OK;
}

// ---- inner iterations: one coordinate-descent sweep over the candidate arrays -----------
static int ensure_track_outputs(tmi_ba_solver* s);

static int ensure_inner(tmi_ba_solver* s) {
  tmi_ba_solver::InnerCtx& I = s->inner;
  if (I.ready) return TMI_BA_OK;
  int rc;
  const Structure& st = s->st;
  // view-major observation index from the track-major layout, built on the device: a stable radix
  // sort of the layout's elements by camera (within a view: ascending element, as the host loop
  // it replaces produced), the view pointers by binary search, the track of every element alongside
  int *d_vo_ptr, *d_vo_e, *d_vo_lp;
  double* d_vo_xy;  // the pixels in the same order: streamed by inner_eval (a 16-byte gather by element costs a 128-byte line)
  if ((rc = dev_alloc(s, &d_vo_xy, 2 * (size_t)std::max<int64_t>(st.No_pad, 1)))) return rc;
  if ((rc = dev_alloc(s, &d_vo_ptr, (size_t)st.Nc + 2))) return rc;
  if ((rc = dev_alloc(s, &d_vo_e, (size_t)std::max<int64_t>(st.No_pad, 1)))) return rc;
  if ((rc = dev_alloc(s, &d_vo_lp, (size_t)std::max<int64_t>(st.No_pad, 1)))) return rc;
  if (st.No_pad >= ((int64_t)1 << 31)) {
    s->error = "inner iterations: more than 2^31 layout elements";
    return TMI_BA_ERR_UNSUPPORTED;
  }
  {
    using namespace tmi::sg;
    hipStream_t stream = s->stream;
    TempPool tmp;
    unsigned *d_key_in, *d_key_out;
    int *d_val_in, *d_elem_lp;
    const size_t n = (size_t)std::max<int64_t>(st.No_pad, 1);
    TMI_HIP(tmp.get(&d_key_in, n));
    TMI_HIP(tmp.get(&d_key_out, n));
    TMI_HIP(tmp.get(&d_val_in, n));
    TMI_HIP(tmp.get(&d_elem_lp, n));
    if (st.No_pad > 0) {
      hipLaunchKernelGGL(view_index_keys_kernel, dim3((st.Np_pad + 255) / 256), dim3(256), 0, stream, st.Np_pad, s->v.pt_k,
                         s->v.slice_ptr, s->v.obs_cam, st.Nc, d_key_in, d_elem_lp);
      hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((st.No_pad + 255) / 256)), dim3(256), 0, stream, d_val_in, (int)st.No_pad);
      size_t bytes = 0;
      const int bits = bits_for((unsigned)st.Nc + 1);
      TMI_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, d_key_in, d_key_out, d_val_in, d_vo_e, (int)st.No_pad, 0, bits, stream));
      void* cub_tmp = nullptr;
      TMI_HIP(tmp.get((char**)&cub_tmp, std::max<size_t>(bytes, 16)));
      TMI_HIP(hipcub::DeviceRadixSort::SortPairs(cub_tmp, bytes, d_key_in, d_key_out, d_val_in, d_vo_e, (int)st.No_pad, 0, bits, stream));
      hipLaunchKernelGGL(gather_int_kernel, dim3((unsigned)((st.No_pad + 255) / 256)), dim3(256), 0, stream, d_vo_e,
                         (long long)st.No_pad, d_elem_lp, d_vo_lp);
      hipLaunchKernelGGL(gather_double2_kernel, dim3((unsigned)((st.No_pad + 255) / 256)), dim3(256), 0, stream, d_vo_e,
                         (long long)st.No_pad, reinterpret_cast<const double2*>(s->v.obs_xy), reinterpret_cast<double2*>(d_vo_xy));
    }
    hipLaunchKernelGGL((lower_bound_kernel<unsigned, int>), dim3((st.Nc + 1 + 255) / 256), dim3(256), 0, stream, d_key_out,
                       (long long)st.No_pad, (long long)st.Nc, d_vo_ptr);
    TMI_HIP(hipStreamSynchronize(stream));
  }
  double* d_part;
  if ((rc = dev_alloc(s, &d_part, (size_t)std::max(st.Nc, 1) * kInnerPart))) return rc;
  if ((rc = dev_alloc(s, &I.d_active, 1))) return rc;
  TMI_HIP(hipHostMalloc((void**)&I.h_active, sizeof(int), hipHostMallocDefault));
  for (int kind = 0; kind < 2; ++kind) {
    std::vector<int> view_block((size_t)st.Nc, -1), bptr(1, 0), bviews, bn, bparam, bsize;
    std::vector<signed char> bcols;
    if (kind == 0) {
      for (int c = 0; c < st.Nc; ++c) {
        const unsigned m = st.cam_mask[c] & 0x3fu;
        if (!m) continue;
        view_block[c] = (int)bn.size();
        bviews.push_back(c);
        bptr.push_back((int)bviews.size());
        int n = 0;
        signed char cols[kInnerMaxN] = {0};
        for (int a = 0; a < 6; ++a)
          if (m & (1u << a)) cols[n++] = (signed char)a;
        bn.push_back(n);
        bcols.insert(bcols.end(), cols, cols + kInnerMaxN);
        bparam.push_back(6 * c);
        bsize.push_back(6);
      }
    } else {
      for (int g = 0; g < st.G; ++g) {
        const unsigned m = st.grp_free[g];
        if (!m) continue;
        const int b = (int)bn.size();
        for (int c = 0; c < st.Nc; ++c)
          if (st.cam_group[c] == g) {
            view_block[c] = b;
            bviews.push_back(c);
          }
        bptr.push_back((int)bviews.size());
        int n = 0;
        signed char cols[kInnerMaxN] = {0};
        for (int a = 0; a < kInnerMaxN; ++a)
          if (m & (1u << a)) cols[n++] = (signed char)a;
        bn.push_back(n);
        bcols.insert(bcols.end(), cols, cols + kInnerMaxN);
        bparam.push_back(st.group_offset[g]);
        bsize.push_back(st.group_offset[g + 1] - st.group_offset[g]);
      }
    }
    InnerSet& S = I.set[kind];
    memset(&S, 0, sizeof(S));
    S.kind = kind;
    S.nblocks = (int)bn.size();
    int* pi;
    signed char* pc;
    if ((rc = dev_upload(s, &pi, view_block))) return rc; S.view_block = pi;
    if ((rc = dev_upload(s, &pi, bptr))) return rc; S.blk_views_ptr = pi;
    if ((rc = dev_upload(s, &pi, bviews))) return rc; S.blk_views = pi;
    if ((rc = dev_upload(s, &pi, bn))) return rc; S.blk_n = pi;
    if ((rc = dev_upload(s, &pc, bcols))) return rc; S.blk_cols = pc;
    if ((rc = dev_upload(s, &pi, bparam))) return rc; S.blk_param = pi;
    if ((rc = dev_upload(s, &pi, bsize))) return rc; S.blk_size = pi;
    S.vo_ptr = d_vo_ptr; S.vo_e = d_vo_e; S.vo_lp = d_vo_lp; S.vo_xy = d_vo_xy;
    const size_t nx = kind == 0 ? (size_t)6 * std::max(st.Nc, 1) : (size_t)std::max(s->n_intr, 1);
    if ((rc = dev_alloc(s, &I.x0[kind], nx))) return rc;
    if ((rc = dev_alloc(s, &I.xc[kind], nx))) return rc;
    S.part = d_part;
    const size_t nb = (size_t)std::max(S.nblocks, 1);
    if ((rc = dev_alloc(s, &S.H, nb * kInnerNS))) return rc;
    if ((rc = dev_alloc(s, &S.g, nb * kInnerMaxN))) return rc;
    if ((rc = dev_alloc(s, &S.scale, nb * kInnerMaxN))) return rc;
    if ((rc = dev_alloc(s, &S.st_d, nb * 8))) return rc;
    if ((rc = dev_alloc(s, &S.st_i, nb * 8))) return rc;
    S.active = I.d_active;
  }
  if ((rc = dev_alloc(s, &I.bak_ext, (size_t)6 * std::max(st.Nc, 1)))) return rc;
  if ((rc = dev_alloc(s, &I.bak_intr, (size_t)std::max(s->n_intr, 1)))) return rc;
  if ((rc = dev_alloc(s, &I.bak_pts, (size_t)4 * std::max(st.Np_pad, 1)))) return rc;
  I.ready = true;
  return TMI_BA_OK;
}

// Moves the candidate (ext_c, intr_c, pts_c) by one sweep; the values before the sweep stay
// in the bak_* buffers.
static int run_inner_sweep(tmi_ba_solver* s, const tmi_ba_options* O) {
  int rc = ensure_inner(s);
  if (rc) return rc;
  tmi_ba_solver::InnerCtx& I = s->inner;
  const Structure& st = s->st;
  DeviceView& v = s->v;
  hipStream_t stream = s->stream;
  TMI_HIP(hipMemcpyAsync(I.bak_ext, v.ext_c, (size_t)6 * st.Nc * sizeof(double), hipMemcpyDeviceToDevice, stream));
  if (s->n_intr) TMI_HIP(hipMemcpyAsync(I.bak_intr, v.intr_c, (size_t)s->n_intr * sizeof(double), hipMemcpyDeviceToDevice, stream));
  TMI_HIP(hipMemcpyAsync(I.bak_pts, v.pts_c, (size_t)4 * st.Np_pad * sizeof(double), hipMemcpyDeviceToDevice, stream));
  // reversed linear-solver ordering (bundle_adjuster.cc:193-200 with the groups of :346-371):
  // extrinsics blocks, then intrinsics blocks, then the points
  for (int kind = 0; kind <= 1; ++kind) {
    InnerSet S = I.set[kind];
    if (S.nblocks == 0 || st.Nc == 0) continue;
    S.x = kind == 0 ? v.ext_c : v.intr_c;
    S.xc = I.xc[kind];
    S.x0 = I.x0[kind];
    S.loss_type = O->loss_function_type;
    S.loss_width = O->robust_loss_width;
    const size_t nx = kind == 0 ? (size_t)6 * st.Nc : (size_t)s->n_intr;
    TMI_HIP(hipMemcpyAsync(I.x0[kind], S.x, nx * sizeof(double), hipMemcpyDeviceToDevice, stream));
    TMI_HIP(hipMemcpyAsync(I.xc[kind], S.x, nx * sizeof(double), hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(inner_init_kernel, dim3((S.nblocks + 255) / 256), dim3(256), 0, stream, S);
    // at most 50 steps per block + the re-linearisation round of the last accepted one
    for (int round = 0; round < 52; ++round) {
      // a view's observations may be spread over the ranks: the per-view partials are summed
      // across them after each evaluation pass; the per-block kernels then run replicated
      const size_t part_bytes = (size_t)st.Nc * kInnerPart * sizeof(double);
      {
        Timed t(s, TMI_BA_K_LINEARIZE);
        TMI_HIP(hipMemsetAsync(S.part, 0, part_bytes, stream));
        // (one camera model for the whole problem: the instantiation without the model switch, inner_kernels.h)
        const bool uni = v.uniform_pinhole_default != 0;
        if (kind == 0) {
          if (uni) hipLaunchKernelGGL((inner_eval_kernel<0, true, 0>), dim3(st.Nc), dim3(256), 0, stream, v, S);
          else hipLaunchKernelGGL((inner_eval_kernel<0, true>), dim3(st.Nc), dim3(256), 0, stream, v, S);
        } else {
          if (uni) hipLaunchKernelGGL((inner_eval_kernel<1, true, 0, 3>), dim3(st.Nc), dim3(256), 0, stream, v, S);  // (f, k1, k2)
          else hipLaunchKernelGGL((inner_eval_kernel<1, true>), dim3(st.Nc), dim3(256), 0, stream, v, S);
        }
      }
      if ((rc = do_allreduce(s, S.part, (int64_t)st.Nc * kInnerPart))) return rc;
      {
        Timed t(s, TMI_BA_K_LINEARIZE);
        hipLaunchKernelGGL(inner_step_kernel, dim3(S.nblocks), dim3(64), 0, stream, S);
        TMI_HIP(hipMemsetAsync(S.part, 0, part_bytes, stream));
        // (one camera model for the whole problem: the instantiation without the model switch, inner_kernels.h)
        const bool uni = v.uniform_pinhole_default != 0;
        if (kind == 0) {
          if (uni) hipLaunchKernelGGL((inner_eval_kernel<0, false, 0>), dim3(st.Nc), dim3(256), 0, stream, v, S);
          else hipLaunchKernelGGL((inner_eval_kernel<0, false>), dim3(st.Nc), dim3(256), 0, stream, v, S);
        } else {
          if (uni) hipLaunchKernelGGL((inner_eval_kernel<1, false, 0>), dim3(st.Nc), dim3(256), 0, stream, v, S);
          else hipLaunchKernelGGL((inner_eval_kernel<1, false>), dim3(st.Nc), dim3(256), 0, stream, v, S);
        }
      }
      if ((rc = do_allreduce(s, S.part, (int64_t)st.Nc * kInnerPart))) return rc;
      TMI_HIP(hipMemsetAsync(I.d_active, 0, sizeof(int), stream));
      hipLaunchKernelGGL(inner_decide_kernel, dim3(S.nblocks), dim3(64), 0, stream, S);
      TMI_HIP(hipMemcpyAsync(I.h_active, I.d_active, sizeof(int), hipMemcpyDeviceToHost, stream));
      TMI_HIP(hipStreamSynchronize(stream));
      if (*I.h_active == 0) break;
    }
  }
  // the points, each against its (now constant) cameras: the batched single-track solver on
  // the candidate arrays, Ceres' default minimizer options
  if (st.nslices > 0) {
    TrackLmArgs A;
    A.loss_type = O->loss_function_type;
    A.loss_width = O->robust_loss_width;
    A.jacobi_scaling = 1;
    A.max_num_iterations = 50;
    A.max_num_consecutive_invalid_steps = 5;
    A.function_tolerance = 1e-6;
    A.gradient_tolerance = 1e-10;
    A.parameter_tolerance = 1e-8;
    A.initial_radius = 1e4;
    A.max_radius = 1e16;
    A.min_radius = 1e-32;
    A.min_relative_decrease = 1e-3;
    A.lm_lo = 1e-6;
    A.lm_hi = 1e32;
    if ((rc = ensure_track_outputs(s))) return rc;
    DeviceView vc = v;
    vc.ext = v.ext_c;
    vc.intr = v.intr_c;
    vc.pts = v.pts_c;
    Timed t(s, TMI_BA_K_LINEARIZE);
    prepare_cameras(s, v.ext_c, v.intr_c, v.prep_c);  // the two view sets moved the candidate cameras
    // (one camera model for the whole problem: the instantiation without the model switch, track_kernels.h)
    if (s->DP == 3 && v.uniform_pinhole_default)
      hipLaunchKernelGGL((track_lm_kernel<3, 0>), dim3(s->nblocks_tracks), dim3(256), 0, stream, vc, v.prep_c, A, s->d_trk_term,
                         s->d_trk_iter, s->d_trk_c0, s->d_trk_c1);
    else if (s->DP == 3)
      hipLaunchKernelGGL(track_lm_kernel<3>, dim3(s->nblocks_tracks), dim3(256), 0, stream, vc, v.prep_c, A, s->d_trk_term,
                         s->d_trk_iter, s->d_trk_c0, s->d_trk_c1);
    else if (v.uniform_pinhole_default)
      hipLaunchKernelGGL((track_lm_kernel<4, 0>), dim3(s->nblocks_tracks), dim3(256), 0, stream, vc, v.prep_c, A, s->d_trk_term,
                         s->d_trk_iter, s->d_trk_c0, s->d_trk_c1);
    else
      hipLaunchKernelGGL(track_lm_kernel<4>, dim3(s->nblocks_tracks), dim3(256), 0, stream, vc, v.prep_c, A, s->d_trk_term,
                         s->d_trk_iter, s->d_trk_c0, s->d_trk_c1);
  }
  return TMI_BA_OK;
}

int32_t tmi_ba_solver_solve(tmi_ba_solver* s, const tmi_ba_options* O, tmi_ba_summary* sum) {
  if (!s || !O || !sum) return TMI_BA_ERR_INVALID_ARGUMENT;
  memset(sum, 0, sizeof(*sum));
  sum->termination = 2;
  if (O->point_dof != s->DP || s->light) {
    sum->status = TMI_BA_ERR_INVALID_ARGUMENT;
    set_message(sum, s->light ? "this handle only holds the per-track side kernels"
                              : "point_dof differs from the value the solver was created with");
    return sum->status;
  }
  auto fail = [&](int rc) {
    g_last_error = s->error;
    sum->status = rc;
    sum->success = 0;
    sum->termination = 2;
    set_message(sum, s->error.empty() ? tmi_ba_status_string(rc) : s->error.c_str());
    return rc;
  };
  if (hipSetDevice(s->device) != hipSuccess) {
    s->error = "hipSetDevice failed";
    return fail(TMI_BA_ERR_DEVICE);
  }
  const double t_start = now_s();
  DeviceView& v = s->v;
  Structure& st = s->st;
  const RedLayout& RL = s->RL;
  const int D = st.D;
  const int n_r = st.Nrb * D;
  const int nbs = s->nblocks_tracks, nbp = s->nblocks_points;
  hipStream_t stream = s->stream;
  s->prof_mask = (O->profile_kernels == 1) ? 0xffffffffu : (unsigned)O->profile_kernels;
  s->cl_retired = false;
  s->ev_used = 0;
  memset(s->launches, 0, sizeof(s->launches));
  sum->num_reduced_blocks = st.Nrb;
  sum->reduced_block_dim = D;
  sum->num_schur_blocks = st.nub + st.Nrb;
  sum->num_schur_pairs = st.npairs;
  sum->setup_time_in_seconds = s->setup_seconds;
  int rc;
#define CK(x) if ((rc = (x)) != TMI_BA_OK) return fail(rc)
#define CKH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { s->error = std::string(#call) + ": " + hipGetErrorString(e_); return fail(TMI_BA_ERR_DEVICE); } } while (0)

  const int lt = O->loss_function_type;
  const double lw = O->robust_loss_width;
  const bool iterative = (O->linear_solver_type == TMI_BA_ITERATIVE_SCHUR || O->linear_solver_type == TMI_BA_CGNR);
  s->cur_opts = O;
  if (iterative && (s->st.has_shared || s->vis_clusters) &&
      ((s->tri && O->preconditioner_type == TMI_BA_PRECOND_CLUSTER_JACOBI) ||
       (!s->tri && O->preconditioner_type == TMI_BA_PRECOND_CLUSTER_TRIDIAGONAL && st.world == 1 && !s->implicit))) {
    // the clusters (CLUSTER_JACOBI) or chains of clusters (CLUSTER_TRIDIAGONAL) are part of the handle's structure
    s->error = "CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL: the solver was created for the other one";
    return fail(TMI_BA_ERR_INVALID_ARGUMENT);
  }
  if (iterative && s->vis_clusters &&
      (O->preconditioner_type == TMI_BA_PRECOND_CLUSTER_JACOBI || O->preconditioner_type == TMI_BA_PRECOND_CLUSTER_TRIDIAGONAL) &&
      O->visibility_clustering_type != s->vis_type) {
    // the clusters are part of the handle's structure (built at create from visibility_clustering_type)
    s->error = "visibility_clustering_type differs from the value the solver was created with";
    return fail(TMI_BA_ERR_INVALID_ARGUMENT);
  }
  if (s->implicit && !iterative) {
    s->error = "this solver was created for the implicit Schur operator (ITERATIVE_SCHUR); "
               "an exact linear solver type needs schur_mode = explicit at creation";
    return fail(TMI_BA_ERR_INVALID_ARGUMENT);
  }
  double* d_sc = v.red + RL.scalars;  // 8 device scalars that get all-reduced

  // ---- iteration zero ---------------------------------------------------------------
  CKH(hipMemsetAsync(v.flags, 0, FL_COUNT * sizeof(int), stream));
  CKH(hipMemsetAsync(v.yc, 0, std::max(n_r, 1) * sizeof(double), stream));
  hipLaunchKernelGGL(fill_kernel, dim3((n_r + 255) / 256 + 1), dim3(256), 0, stream, v.scale_c, (long long)n_r, 1.0);
  s->launch.expand_scale(v, stream);
  prepare_cameras(s, v.ext, v.intr, v.prep);
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((long long)st.Np_pad * s->DP + 255) / 256 + 1)), dim3(256), 0, stream, v.scale_p, (long long)st.Np_pad * s->DP, 1.0);
  // Compact planes (device_view.h): p_n instead of the stored camera block wherever every consumer of the planes forms
  // A x / A^T t from Jp, p_n, the track and the view -- the all-PINHOLE / default-mask / TRIVIAL-loss problem without
  // stored position columns, unit aspect ratio and zero skew (constant under that mask: read from the parameters the
  // solve starts from), matrix-free iterations of the one-sweep product whose camera side is built view by view
  // (no kernel reads the A planes then but the product and back_substitute).  Decided per linearize from the operator
  // the next LM iteration is expected to run; an iteration that forms S after all re-linearizes with the full planes.
  bool compact_possible = s->compact_env && v.drop_pos && v.uniform_pinhole_default && s->mf_ok && s->direct_ok && D == 9 &&
                          (lt == 0 || s->compact_robust) && iterative && !s->cluster_blocks && v.cp_trk != nullptr && !v.planes_fp32;
  if (compact_possible) {
    compact_possible = (int)s->grp_off_h.size() == st.G + 1;
    for (int g = 0; g < st.G && compact_possible; ++g) {
      const double* K = s->intr0.data() + s->grp_off_h[g];
      compact_possible = s->grp_off_h[g + 1] - s->grp_off_h[g] >= 7 && K[1] == 1.0 && K[2] == 0.0;
    }
  }
  int last_pcg_len = 0;  // PCG iterations of the previous LM iteration (0: none yet)
  auto linearize = [&](bool norms_only = false, bool full_planes = false) {
    if (!norms_only)
      v.compact = (!full_planes && compact_possible && (s->implicit || (s->adaptive && last_pcg_len <= s->adaptive_break_even)))
                      ? (lt == 0 ? 1 : 2)  // (2: a robust loss -- the planes hold the corrected point and r^2)
                      : 0;
    if (!norms_only) v.sums_ready = (v.compact && s->fuse_sums) ? 1 : 0;  // (the compact instantiation also leaves V and g_p)
    // cost and sum of squares land in d_sc[0..1] (finished by the kernel's last workgroup)
    Timed t(s, TMI_BA_K_LINEARIZE);
    // (drop_pos: the kernel also leaves -w / scale_p of every track, at the point and the scales the planes are taken at)
    s->launch.linearize(v, stream, v.prep, lt, lw, nbs, d_sc, norms_only ? 1 : 0);
  };
  // (direct_diag.h) the pre-pass below only needs the U diagonal: without records whenever the handle can
  v.direct_diag = (s->direct_ok && iterative) ? 1 : 0;
  // Round 6: where the camera side is built view by view (no camera-major records) the start of a solve writes no
  // planes before their Jacobi scales are known: the first pass is the norms-only linearize (cost + scale_p), the
  // U diagonal comes from camera_diag_direct on records that carry the points alone -- instead of a full linearize,
  // point_scale and point_eliminate (0.9 -> 0.6 ms at Venice size).  TMI_BA_FAST_START=0: the three-pass start.
  const bool fast_start = O->jacobi_scaling && v.direct_diag && s->fast_start_ok;  // (direct_ok implies fp64 evaluation)
  // (the unscaled first pass of the three-pass start keeps the full planes: its point columns are what point_scale sums,
  //  and the norms-only pass is held to that instantiation bit for bit -- tests/test_gpu_launch_sequence.py)
  linearize(fast_start, /*full_planes=*/O->jacobi_scaling != 0);
  // d_sc[0] = cost, d_sc[1] = ss, d_sc[2] = #ranks with an invalid residual
  hipLaunchKernelGGL(flag_to_scalar_kernel, dim3(1), dim3(64), 0, stream, v.flags + FL_INVALID, d_sc + 2);
  CK(do_allreduce(s, d_sc, 8));
  CK(readback(s));
  double hsc[8];
  memcpy(hsc, s->h_red, sizeof(hsc));
  const int64_t No_global_hint = st.No;  // per-rank; RMSE uses the global count below
  (void)No_global_hint;
  if (hsc[2] > 0.0) {
    s->error = "residual evaluation failed at the start point (a track lies on a camera centre)";
    return fail(TMI_BA_ERR_EVALUATION_FAILED);
  }
  double cost = hsc[0];
  // global observation count for the RMSE
  double n_obs_global = (double)st.No;
  if (st.world > 1) {
    double tmp[8] = {(double)st.No, 0, 0, 0, 0, 0, 0, 0};
    CKH(hipMemcpyAsync(d_sc, tmp, sizeof(tmp), hipMemcpyHostToDevice, stream));
    CK(do_allreduce(s, d_sc, 8));
    CKH(hipMemcpyAsync(tmp, d_sc, sizeof(tmp), hipMemcpyDeviceToHost, stream));
    CKH(hipStreamSynchronize(stream));
    n_obs_global = tmp[0];
  }
  sum->initial_cost = cost;
  sum->initial_rmse = n_obs_global > 0 ? std::sqrt(hsc[1] / n_obs_global) : 0.0;
  double final_ss = hsc[1];

  // trial cost at the candidate (prep_c, pts_c): view by view where every observation owns a slot (direct_diag.h)
  auto trial_cost = [&]() {
    if (s->cost_by_view) {
      const int nb = (s->dd.n_chunks + 3) / 4;
      hipLaunchKernelGGL(ddg::cost_view_kernel, dim3(nb), dim3(256), 0, stream, v, s->dd, v.prep_c, v.pts_c, lt, lw, FL_INVALID,
                         nb, v.partial, d_sc + 3, d_sc + 5, s->cost_warm ? 1 : 0);
    } else {
      s->launch.cost(v, stream, v.prep_c, v.pts_c, lt, lw, FL_INVALID, nbs, v.partial, d_sc + 3, d_sc + 5);
    }
  };
  // builds the camera side of the normal equations from the current linearisation
  auto build_camera_side = [&](double inv_radius, bool skip_direct_reduce = false) {
    {
      Timed t(s, TMI_BA_K_POINT_ELIMINATE);
      s->launch.point_eliminate(v, stream, inv_radius, O->min_lm_diagonal, O->max_lm_diagonal, nbs,
                                s->d_partial_max, d_sc + 6, O->gradient_tolerance, st.world > 1 ? d_sc + 0 : nullptr);
    }
    {
      Timed t(s, TMI_BA_K_CAMERA_DIAG);
      if (v.direct_diag) {
        s->launch.camera_diag_direct(v, stream, RL, s->dd, v.prep, lt, lw, skip_direct_reduce ? 1 : 0);
      } else {
        s->launch.camera_diag(v, stream, RL, s->shared_diag_chunks, s->d_shared_diag_partial);
        s->launch.shared_blocks(v, stream, RL);
      }
    }
  };
  if (O->jacobi_scaling) {
    // Jacobi scaling 1 / (1 + ||column||) from the UNSCALED Jacobian at the start point:
    // track columns directly, camera-side columns as the diagonal of J_c^T J_c which a
    // first pass of point_eliminate + camera_diag leaves in `red` (udiag).
    if (fast_start) {
      // (scale_p is the norms-only linearize's; records = the points alone)
      Timed t(s, TMI_BA_K_CAMERA_DIAG);
      s->launch.track_records_points_only(v, stream, v.pts);
      s->launch.camera_diag_direct(v, stream, RL, s->dd, v.prep, lt, lw, 0);
    } else {
      {
        Timed t(s, TMI_BA_K_REDUCE);
        s->launch.point_scale(v, stream, nbs);
      }
      build_camera_side(1.0);
    }
    CK(do_allreduce(s, v.red + RL.udiag, n_r));
    {
      Timed t(s, TMI_BA_K_REDUCE);
      if (n_r) {
        CKH(hipMemcpyAsync(v.scale_c, v.red + RL.udiag, (size_t)n_r * sizeof(double), hipMemcpyDeviceToDevice, stream));
        hipLaunchKernelGGL(camera_scale_finish_kernel, dim3((n_r + 255) / 256), dim3(256), 0, stream, v.scale_c, n_r);
        s->launch.expand_scale(v, stream);
        prepare_cameras(s, v.ext, v.intr, v.prep);
      }
    }
    linearize();
  }
  // |x| of the start point: camera part through update_cameras with y = 0
  double xnorm_cam_sq = 0.0, xnorm_pts_sq = 0.0;
  {
    Timed t(s, TMI_BA_K_UPDATE_COST);
    s->launch.update_cameras(v, stream, v.scal + SC_STEP_SQ, v.prep_c);  // y = 0: copies + |x|
    hipLaunchKernelGGL(points_norm_kernel, dim3(nbp), dim3(256), 0, stream, v, v.pts, nbp, v.partial, d_sc);
  }
  CK(do_allreduce(s, d_sc, 8));
  CK(readback(s));
  memcpy(hsc, s->h_red, sizeof(hsc));
  xnorm_cam_sq = s->h_scal[SC_STEP_SQ + 1];
  xnorm_pts_sq = hsc[0];
  double x_norm = std::sqrt(xnorm_cam_sq + xnorm_pts_sq);

  const double t_loop = now_s();
  double radius = O->initial_trust_region_radius;
  double decrease_factor = 2.0;
  int invalid_run = 0;
  int iter = 0;
  int termination = 1;
  const char* why = "maximum number of iterations reached";
  int64_t pcg_iters = 0;
  s->n_implicit_iterations = 0;
  bool need_gradient_check = true;  // after the first build and after every accepted step
  bool inner_enabled = O->use_inner_iterations != 0;
  bool time_up = false;

  for (;;) {
    if (iter >= O->max_num_iterations) break;
    // On a sharded solve every termination decision must come from all-reduced data or the
    // ranks would leave the loop at different iterations and the next collective would hang:
    // the local clocks only VOTE (summed with the trial-step scalars below), the sum decides.
    if (st.world > 1 ? time_up : (now_s() - t_start >= O->max_solver_time_in_seconds)) {
      why = "maximum solver time reached";
      break;
    }
    ++iter;
    const double inv_radius = 1.0 / radius;
    const double radius_used = radius;
    const int64_t pcg_before = pcg_iters;
    // one row of the optional trace (theia_mi355_ba.h, tmi_ba_options::iteration_trace)
    auto trace = [&](double outcome, double cand, double mcc, double step) {
      if (!O->iteration_trace || iter > O->iteration_trace_capacity) return;
      double* row = O->iteration_trace + (size_t)(iter - 1) * TMI_BA_TRACE_STRIDE;
      row[0] = iter; row[1] = cost; row[2] = radius_used; row[3] = outcome; row[4] = cand; row[5] = mcc;
      row[6] = (double)(pcg_iters - pcg_before); row[7] = step;
    };
    hipLaunchKernelGGL(iteration_begin_kernel, dim3(1), dim3(64), 0, stream, v.flags, d_sc);  // flags = 0, d_sc[0..7] = 0
    s->cur_inv_radius = inv_radius;
    if (s->adaptive && iterative && !s->st.has_shared) {
      // Forming S pays off after adaptive_break_even products (set at create from the sizes): short PCG
      // solves -- the first LM iterations, small trust regions -- run matrix-free, and point_eliminate then
      // skips the Y records.  The forecast is the previous iteration's PCG length; both operators give the
      // same product to round-off.
      s->implicit_now = last_pcg_len <= s->adaptive_break_even;
      if (s->y_records) v.write_y = s->implicit_now ? 0 : 1;
      if (s->implicit_now) s->n_implicit_iterations++;
    } else if (s->implicit && iterative) {
      s->n_implicit_iterations++;
    }
    v.direct_diag = (s->direct_ok && iterative && s->implicit_now && !s->cluster_blocks) ? 1 : 0;
    if (v.compact && !v.direct_diag) {
      // this iteration forms S (records from the full camera block): the linearisation again, with the full planes
      compact_possible = false;
      linearize();
      hipLaunchKernelGGL(iteration_begin_kernel, dim3(1), dim3(64), 0, stream, v.flags, d_sc);
    }
    // one launch for [chunk sums ->] diagonal blocks -> block inverses -> start of PCG (direct_diag.h, camera_finish_kernel)
    // wherever no cluster factorisation sits between the preconditioner blocks and PCG's first residual
    const bool cluster_handle = s->st.has_shared || s->vis_clusters;
    const bool cluster_precond = O->preconditioner_type == TMI_BA_PRECOND_CLUSTER_JACOBI ||
                                 O->preconditioner_type == TMI_BA_PRECOND_CLUSTER_TRIDIAGONAL;
    const bool fuse_finish = s->fuse_finish_ok && iterative && n_r > 0 && !(cluster_handle && cluster_precond);
    const bool fuse_direct = fuse_finish && v.direct_diag && st.world == 1;
    build_camera_side(inv_radius, fuse_direct);
    if (!s->implicit_now || s->cluster_blocks) {
      Timed t(s, TMI_BA_K_SCHUR_OFFDIAG);
      s->launch.schur_offdiag(v, stream, RL);
      s->launch.cross_add(v, stream, RL);
    }
    CK(do_allreduce(s, v.red, RL.total));  // d_sc[6] carries the singular-track votes
    const int precond_mode = O->preconditioner_type == TMI_BA_PRECOND_IDENTITY ? 1
                             : O->preconditioner_type == TMI_BA_PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS ? 2 : 0;
    if (fuse_finish) {
      Timed t(s, TMI_BA_K_PRECONDITIONER);
      s->launch.camera_finish(v, stream, RL, s->dd, inv_radius, O->min_lm_diagonal, O->max_lm_diagonal,
                              need_gradient_check ? 1 : 0, precond_mode, fuse_direct ? 1 : 0);
      s->cl_active = false;
    } else {
      // diagonal blocks + LM diagonal; with the gradient test pending also max |g_c / scale|
      // (max |g_p / scale| was finished by point_eliminate's last workgroup)
      Timed t(s, TMI_BA_K_REDUCE);
      s->launch.expand(v, stream, RL, inv_radius, O->min_lm_diagonal, O->max_lm_diagonal, need_gradient_check ? 1 : 0);
    }
    int usable = 1;
    if (iterative) {
      if (!fuse_finish) {
        Timed t(s, TMI_BA_K_PRECONDITIONER);
        s->launch.precond(v, stream, precond_mode);
        // CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL on a problem with shared intrinsics blocks: the exact inverse of every
        // {shared block, its views} cluster (needs the cluster's blocks of S: the formed operator)
        s->cl_active = (s->tri ? O->preconditioner_type == TMI_BA_PRECOND_CLUSTER_TRIDIAGONAL
                               : O->preconditioner_type == TMI_BA_PRECOND_CLUSTER_JACOBI) &&
                       (s->st.has_shared || s->vis_clusters) && (!s->implicit_now || s->cluster_blocks) && n_r > 0 &&
                       !s->cl_retired && !s->cl_unavailable;
        if (s->cl_active) {
          rc = factor_clusters(s);
          if (rc == TMI_BA_ERR_OUT_OF_MEMORY && !s->cl_built) {
            // the clusters' tiles do not fit: the handle keeps its SCHUR_JACOBI blocks -- (void) the sticky
            // HIP error, forget the message, do not try again
            (void)hipGetLastError();
            s->error.clear();
            s->cl_unavailable = true;
            s->cl_active = false;
          } else if (rc != TMI_BA_OK) {
            return fail(rc);
          }
        }
      }
      const int64_t before = pcg_iters;
      if (s->cl_active && s->cl_failed) {
        usable = 0;  // the tridiagonal preconditioner could not be factored, scaled or not: the linear solve fails (Ceres)
      } else {
        CK(solve_reduced_pcg(s, O, &usable, &pcg_iters, fuse_finish));
      }
      last_pcg_len = (int)(pcg_iters - before);
      // what this iteration's PCG ran with (tmi_ba_summary::effective_preconditioner_type): the clusters only while
      // they are active (cl_active is cleared when a cluster launch retires them mid-solve), JACOBI as SCHUR_JACOBI
      sum->effective_preconditioner_type =
          s->cl_active ? (s->tri ? TMI_BA_PRECOND_CLUSTER_TRIDIAGONAL : TMI_BA_PRECOND_CLUSTER_JACOBI)
          : O->preconditioner_type == TMI_BA_PRECOND_IDENTITY ? TMI_BA_PRECOND_IDENTITY
          : O->preconditioner_type == TMI_BA_PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS ? TMI_BA_PRECOND_SCHUR_JACOBI_PARAMETER_BLOCKS
                                                                                    : TMI_BA_PRECOND_SCHUR_JACOBI;
    } else {
      CK(solve_reduced_dense(s, &usable));
    }
    // the PCG loop ends on a readback and launches nothing after it: the mirror is current
    if (!(iterative && n_r > 0)) {
      CK(readback(s));
      if (!iterative && s->h_flags[FL_CHOL_ABORT]) {
        // the dataflow launch could not become co-resident (another process holds part of the device): clear the
        // flag and take the launch-per-panel path for this solve
        TMI_HIP(hipMemsetAsync(v.flags + FL_CHOL_ABORT, 0, sizeof(int), stream));
        {
          Timed t(s, TMI_BA_K_CHOLESKY);
          CK(solve_reduced_dense_panels(s));
        }
        CK(readback(s));
      }
    }
    // singular track blocks are voted on by every rank (summed in the all-reduce)
    if (s->h_red[6] > 0.0 || s->h_flags[FL_SINGULAR_BLOCK]) usable = 0;
    if (need_gradient_check) {
      // Gradient tolerance.  The camera part of the gradient is all-reduced, so max |g_c| is the same number on
      // every rank; the track part is this rank's own, and on a sharded solve every rank's vote on it was summed
      // by the all-reduce of the reduced system (point_eliminate left it in the scalar tail, which the mirror
      // still shows: nothing has written there since) -- no collective and no synchronisation of its own.
      double vote = (s->h_scal[SC_GMAX] > O->gradient_tolerance) ? 1.0 : 0.0;
      if (st.world > 1)
        vote += s->h_red[0];
      else
        vote += (s->h_scal[SC_GMAX_P] > O->gradient_tolerance) ? 1.0 : 0.0;
      need_gradient_check = false;
      if (vote == 0.0) {
        termination = 0;
        why = "gradient tolerance reached";
        --iter;
        break;
      }
    }
    double model_cost_change = 0.0, step_sq = 0.0, cand_cost = 0.0, cand_ss = 0.0;
    double cand_xc_sq = 0.0, cand_xp_sq = 0.0;
    bool cand_invalid = false;
    if (usable) {
      {
        // candidate cameras + their prepared records (+ the scaled copy of y_c back_substitute gathers: first)
        Timed t(s, TMI_BA_K_UPDATE_COST);
        s->launch.update_cameras(v, stream, v.scal + SC_STEP_SQ, v.prep_c);
      }
      {
        // y_p, the model cost change, the candidate points and their share of |step|^2, |x+|^2: d_sc[0..2]
        Timed t(s, TMI_BA_K_BACK_SUBSTITUTE);
        s->launch.back_substitute(v, stream, nbs, v.partial, d_sc + 0);
      }
      {
        Timed t(s, TMI_BA_K_UPDATE_COST);
        // d_sc: [mcc, step_sq_points, |x+|^2 points, cand_cost, cand_ss, invalid votes, singular
        //        track votes, time-limit votes]
        trial_cost();
      }
      if (st.world > 1 && now_s() - t_start >= O->max_solver_time_in_seconds) {
        // this rank's clock says the time limit has passed: its vote (the slot is zero otherwise, cleared at the
        // top of the iteration) is summed with the trial-step scalars below
        s->time_vote = 1.0;
        CKH(hipMemcpyAsync(d_sc + 7, &s->time_vote, sizeof(double), hipMemcpyHostToDevice, stream));
      }
      CK(do_allreduce(s, d_sc, 8));
      CK(readback(s));
      memcpy(hsc, s->h_red, sizeof(hsc));
      time_up = st.world > 1 && hsc[7] > 0.0;
      model_cost_change = hsc[0];
      step_sq = hsc[1] + s->h_scal[SC_STEP_SQ];
      cand_xp_sq = hsc[2];
      cand_xc_sq = s->h_scal[SC_STEP_SQ + 1];
      cand_cost = hsc[3];
      cand_ss = hsc[4];
      cand_invalid = hsc[5] > 0.0;
      if (!(model_cost_change > 0.0)) usable = 0;
    }
    // DoInnerIterationsIfNeeded (Ceres trust_region_minimizer.cc): one coordinate-descent
    // sweep from the trust-region candidate; its gain is credited to the model
    bool inner_useful = false;
    if (usable && inner_enabled && !cand_invalid) {
      CK(run_inner_sweep(s, O));
      {
        Timed t(s, TMI_BA_K_UPDATE_COST);
        CKH(hipMemsetAsync(v.flags + FL_INVALID, 0, sizeof(int), stream));
        prepare_cameras(s, v.ext_c, v.intr_c, v.prep_c);  // the sweep moved the candidate cameras
        trial_cost();
        hipLaunchKernelGGL(diff_sq_kernel, dim3(1), dim3(1024), 0, stream, v.ext, v.ext_c, (long long)6 * st.Nc, v.scal + SC_II_DEXT);
        hipLaunchKernelGGL(diff_sq_kernel, dim3(1), dim3(1024), 0, stream, v.intr, v.intr_c, (long long)s->n_intr, v.scal + SC_II_DINTR);
        // per-track sums go where the trial step left its own: d_sc[1] = |step|^2 over the
        // points, d_sc[2] = |x+|^2 over the points (summed over the ranks below)
        hipLaunchKernelGGL(points_diff_kernel, dim3(nbp), dim3(256), 0, stream, v, nbp, v.partial);
        hipLaunchKernelGGL(reduce_sum_kernel, dim3(2), dim3(256), 0, stream, v.partial, nbp, d_sc + 1);
        hipLaunchKernelGGL(cameras_norm_kernel, dim3(1), dim3(1024), 0, stream, v, v.scal + SC_II_XC);
      }
      CK(do_allreduce(s, d_sc, 8));
      CK(readback(s));
      const double inner_cost = s->h_red[3], inner_ss = s->h_red[4];
      if (s->h_red[5] > 0.0) {
        // "Inner iteration failed": the trust-region candidate stands
        CKH(hipMemcpyAsync(v.ext_c, s->inner.bak_ext, (size_t)6 * st.Nc * sizeof(double), hipMemcpyDeviceToDevice, stream));
        if (s->n_intr) CKH(hipMemcpyAsync(v.intr_c, s->inner.bak_intr, (size_t)s->n_intr * sizeof(double), hipMemcpyDeviceToDevice, stream));
        CKH(hipMemcpyAsync(v.pts_c, s->inner.bak_pts, (size_t)4 * st.Np_pad * sizeof(double), hipMemcpyDeviceToDevice, stream));
        prepare_cameras(s, v.ext_c, v.intr_c, v.prep_c);
      } else {
        model_cost_change += cand_cost - inner_cost;
        inner_useful = inner_cost < cost;
        inner_enabled = (1.0 - inner_cost / cand_cost) > 1e-3;  // inner_iteration_tolerance
        cand_cost = inner_cost;
        cand_ss = inner_ss;
        step_sq = s->h_scal[SC_II_DEXT] + s->h_scal[SC_II_DINTR] + s->h_red[1];
        cand_xc_sq = s->h_scal[SC_II_XC];
        cand_xp_sq = s->h_red[2];
        sum->num_inner_iteration_steps++;
      }
    }
    if (!usable) {
      // HandleInvalidStep
      trace(-1.0, std::nan(""), model_cost_change, 0.0);
      sum->num_unsuccessful_steps++;
      if (++invalid_run >= O->max_num_consecutive_invalid_steps) {
        termination = 2;
        why = "too many consecutive invalid steps";
        break;
      }
      radius /= decrease_factor;
      decrease_factor *= 2.0;
      if (radius < O->min_trust_region_radius) {
        termination = 0;
        why = "minimum trust region radius reached";
        break;
      }
      continue;
    }
    invalid_run = 0;
    if (cand_invalid) cand_cost = 1.7976931348623157e308;
    const double step_norm = std::sqrt(step_sq);
    if (step_norm <= O->parameter_tolerance * (x_norm + O->parameter_tolerance)) {
      trace(2.0, cand_cost, model_cost_change, step_norm);
      termination = 0;
      why = "parameter tolerance reached";
      break;
    }
    const double cost_change = cost - cand_cost;
    if (std::fabs(cost_change) <= O->function_tolerance * cost) {
      trace(3.0, cand_cost, model_cost_change, step_norm);
      termination = 0;
      why = "function tolerance reached";
      break;
    }
    const double relative_decrease = cost_change / model_cost_change;
    trace((inner_useful || relative_decrease > O->min_relative_decrease) ? 1.0 : 0.0, cand_cost, model_cost_change, step_norm);
    if (inner_useful || relative_decrease > O->min_relative_decrease) {  // IsStepSuccessful
      std::swap(v.ext, v.ext_c);
      std::swap(v.intr, v.intr_c);
      std::swap(v.pts, v.pts_c);
      std::swap(v.prep, v.prep_c);
      cost = cand_cost;
      final_ss = cand_ss;
      x_norm = std::sqrt(cand_xc_sq + cand_xp_sq);
      sum->num_successful_steps++;
      radius = radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
      radius = std::fmin(O->max_trust_region_radius, radius);
      decrease_factor = 2.0;
      linearize();
      need_gradient_check = true;
    } else {
      sum->num_unsuccessful_steps++;
      radius /= decrease_factor;
      decrease_factor *= 2.0;
    }
    if (radius < O->min_trust_region_radius) {
      termination = 0;
      why = "minimum trust region radius reached";
      break;
    }
    if (O->verbose && st.rank == 0)
      fprintf(stderr, "[tmi_ba] it %3d cost %.10e radius %.3e pcg %lld\n", iter, cost, radius, (long long)pcg_iters);
  }
  CKH(hipStreamSynchronize(stream));
  const double t_end = now_s();
  (void)t_loop;

  sum->termination = termination;
  sum->num_iterations = iter;
  sum->num_linear_solver_iterations = pcg_iters;
  sum->num_matrix_free_iterations = s->n_implicit_iterations;
  sum->final_cost = cost;
  sum->final_rmse = n_obs_global > 0 ? std::sqrt(final_ss / n_obs_global) : 0.0;
  sum->success = (termination != 2);
  sum->status = (termination == 2) ? TMI_BA_ERR_LINEAR_SOLVER : TMI_BA_OK;
  sum->solve_time_in_seconds = t_end - t_start;
  set_message(sum, why);
  for (int c = 0; c < TMI_BA_NUM_KERNEL_CLASSES; ++c) sum->kernel_launches[c] = s->launches[c];
  for (size_t i = 0; i < s->ev_used; ++i) {
    float ms = 0.f;
    if (s->events[i].cls < 0) continue;  // (a speculative PCG iteration that found the solve stopped)
    if (hipEventElapsedTime(&ms, s->events[i].a, s->events[i].b) == hipSuccess)
      sum->kernel_seconds[s->events[i].cls] += 1e-3 * ms;
  }
  return sum->status;
#undef CK
#undef CKH
}

int32_t tmi_ba_solve(tmi_ba_problem* P, const tmi_ba_options* O, tmi_ba_summary* sum) {
  if (!P || !O || !sum) return TMI_BA_ERR_INVALID_ARGUMENT;
  tmi_ba_solver* s = nullptr;
  memset(sum, 0, sizeof(*sum));
  const int rc = tmi_ba_solver_create(P, O, 0, 1, &s);
  if (rc != TMI_BA_OK) {
    sum->status = rc;
    sum->termination = 2;
    set_message(sum, g_last_error.empty() ? tmi_ba_status_string(rc) : g_last_error.c_str());
    return rc;
  }
  const int rc2 = tmi_ba_solver_solve(s, O, sum);
  if (sum->success) tmi_ba_solver_download(s, P);
  tmi_ba_solver_destroy(s);
  return rc2;
}

// ---- per-track side kernels ----------------------------------
static int ensure_track_outputs(tmi_ba_solver* s) {
  if (s->d_trk_flag) return TMI_BA_OK;
  const size_t n = (size_t)std::max(s->st.Np_pad, 1);
  int rc;
  if ((rc = dev_alloc(s, &s->d_trk_flag, n))) return rc;
  if ((rc = dev_alloc(s, &s->d_trk_mean, n))) return rc;
  if ((rc = dev_alloc(s, &s->d_trk_term, n))) return rc;
  if ((rc = dev_alloc(s, &s->d_trk_iter, n))) return rc;
  if ((rc = dev_alloc(s, &s->d_trk_c0, n))) return rc;
  if ((rc = dev_alloc(s, &s->d_trk_c1, n))) return rc;
  const size_t nt = (size_t)std::max(s->st.Np_total, 1);
  if ((rc = dev_alloc(s, &s->d_out_u8, nt))) return rc;
  if ((rc = dev_alloc(s, &s->d_out_f64, nt))) return rc;
  if ((rc = dev_alloc(s, &s->d_out_i32, nt))) return rc;
  if ((rc = dev_alloc(s, &s->d_counters, 4))) return rc;
  TMI_HIP(hipHostMalloc((void**)&s->h_counters, 4 * sizeof(int), hipHostMallocDefault));
  TMI_HIP(hipHostMalloc((void**)&s->h_cell_total, sizeof(long long), hipHostMallocDefault));
  TMI_HIP(hipHostMalloc((void**)&s->h_stage, nt * 16, hipHostMallocDefault));
  return TMI_BA_OK;
}

int32_t tmi_ba_solver_filter_outlier_tracks(tmi_ba_solver* s, double max_inlier_reprojection_error,
                                            double min_triangulation_angle_degrees,
                                            uint8_t* track_flag, double* track_mean_sq_error,
                                            tmi_ba_filter_summary* sum) {
  if (!s || !sum) return TMI_BA_ERR_INVALID_ARGUMENT;
  memset(sum, 0, sizeof(*sum));
  const double t0 = now_s();
  TMI_HIP(hipSetDevice(s->device));
  int rc = ensure_track_outputs(s);
  if (rc) return rc;
  const Structure& st = s->st;
  const double max_sq = max_inlier_reprojection_error * max_inlier_reprojection_error;
  const double cos_min = std::cos(min_triangulation_angle_degrees * (M_PI / 180.0));
  hipEvent_t ea, eb;
  TMI_HIP(hipEventCreate(&ea));
  TMI_HIP(hipEventCreate(&eb));
  TMI_HIP(hipEventRecord(ea, s->stream));
  if (st.nslices > 0)
    hipLaunchKernelGGL(outlier_filter_kernel, dim3(s->nblocks_tracks), dim3(256), 0, s->stream, s->v, max_sq,
                       cos_min, s->d_trk_flag, s->d_trk_mean);
  TMI_HIP(hipEventRecord(eb, s->stream));
  float ms = 0.f;
  if (st.world == 1) {
    // counts and the permutation to the caller's track order happen on the device; one copy per
    // requested output (round 1: flags + means of every slot copied out and walked on the host)
    TMI_HIP(hipMemsetAsync(s->d_counters, 0, 4 * sizeof(int), s->stream));
    if (st.Np_pad > 0)
      hipLaunchKernelGGL(filter_finish_kernel, dim3((st.Np_pad + 255) / 256), dim3(256), 0, s->stream, s->d_pt_orig,
                         st.Np_pad, s->d_trk_flag, s->d_trk_mean, track_flag ? s->d_out_u8 : nullptr,
                         track_mean_sq_error ? s->d_out_f64 : nullptr, s->d_counters);
    TMI_HIP(hipMemcpyAsync(s->h_counters, s->d_counters, 4 * sizeof(int), hipMemcpyDeviceToHost, s->stream));
    const size_t ntot = (size_t)st.Np_total;
    unsigned char* stage_u8 = s->h_stage;
    double* stage_f64 = reinterpret_cast<double*>(s->h_stage + 8 * ntot);
    if (track_flag && ntot > 0)
      TMI_HIP(hipMemcpyAsync(stage_u8, s->d_out_u8, ntot, hipMemcpyDeviceToHost, s->stream));
    if (track_mean_sq_error && ntot > 0)
      TMI_HIP(hipMemcpyAsync(stage_f64, s->d_out_f64, ntot * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    TMI_HIP(hipStreamSynchronize(s->stream));
    if (track_flag && ntot > 0) memcpy(track_flag, stage_u8, ntot);
    if (track_mean_sq_error && ntot > 0) memcpy(track_mean_sq_error, stage_f64, ntot * sizeof(double));
    hipEventElapsedTime(&ms, ea, eb);
    hipEventDestroy(ea);
    hipEventDestroy(eb);
    sum->num_estimated_tracks = s->h_counters[0];
    sum->num_bad_reprojections = s->h_counters[1];
    sum->num_insufficient_viewing_angles = s->h_counters[2];
  } else {
    std::vector<unsigned char> flag((size_t)st.Np_pad);
    std::vector<double> mean(track_mean_sq_error ? (size_t)st.Np_pad : 0);
    if (!flag.empty())
      TMI_HIP(hipMemcpyAsync(flag.data(), s->d_trk_flag, flag.size(), hipMemcpyDeviceToHost, s->stream));
    if (!mean.empty())
      TMI_HIP(hipMemcpyAsync(mean.data(), s->d_trk_mean, mean.size() * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    TMI_HIP(hipStreamSynchronize(s->stream));
    hipEventElapsedTime(&ms, ea, eb);
    hipEventDestroy(ea);
    hipEventDestroy(eb);
    for (int lp = 0; lp < st.Np_pad; ++lp) {
      const int p = st.pt_orig[lp];
      if (p < 0) continue;
      const unsigned char f = flag[lp];
      sum->num_estimated_tracks++;
      if (f == 1) sum->num_bad_reprojections++;
      if (f == 2) sum->num_insufficient_viewing_angles++;
      if (track_flag) track_flag[p] = f;
      if (track_mean_sq_error) track_mean_sq_error[p] = mean[lp];
    }
  }
  // a track nobody observes: mean = 0 / 0, no ray pair -> insufficient viewing angle
  // (set_outlier_tracks_to_unestimated.cc:108,120-125 with empty lists)
  for (const int p : st.unobserved) {
    sum->num_estimated_tracks++;
    sum->num_insufficient_viewing_angles++;
    if (track_flag) track_flag[p] = 2;
    if (track_mean_sq_error) track_mean_sq_error[p] = std::nan("");
  }
  sum->kernel_seconds = ms * 1e-3;
  sum->seconds = now_s() - t0;
  return TMI_BA_OK;
}

int32_t tmi_ba_filter_outlier_tracks(const tmi_ba_problem* P, int32_t device,
                                     double max_inlier_reprojection_error,
                                     double min_triangulation_angle_degrees, uint8_t* track_flag,
                                     double* track_mean_sq_error, tmi_ba_filter_summary* sum) {
  if (!P || !sum) return TMI_BA_ERR_INVALID_ARGUMENT;
  memset(sum, 0, sizeof(*sum));
  const double t0 = now_s();
  tmi_ba_options O;
  tmi_ba_options_init(&O);
  O.device = device;
  tmi_ba_solver* s = new tmi_ba_solver();
  int rc = create_impl(s, P, &O, 0, 1, /*light=*/true);
  if (rc == TMI_BA_OK)
    rc = tmi_ba_solver_filter_outlier_tracks(s, max_inlier_reprojection_error,
                                             min_triangulation_angle_degrees, track_flag,
                                             track_mean_sq_error, sum);
  else
    g_last_error = s->error;
  tmi_ba_solver_destroy(s);
  sum->seconds = now_s() - t0;
  return rc;
}

int32_t tmi_ba_solver_adjust_tracks(tmi_ba_solver* s, const tmi_ba_options* O, int8_t* track_termination,
                                    int32_t* track_iterations, double* track_initial_cost,
                                    double* track_final_cost, tmi_ba_track_batch_summary* sum) {
  if (!s || !O || !sum) return TMI_BA_ERR_INVALID_ARGUMENT;
  memset(sum, 0, sizeof(*sum));
  if (O->point_dof != s->DP) return TMI_BA_ERR_INVALID_ARGUMENT;
  const double t0 = now_s();
  TMI_HIP(hipSetDevice(s->device));
  int rc = ensure_track_outputs(s);
  if (rc) return rc;
  const Structure& st = s->st;
  TrackLmArgs A;
  A.loss_type = O->loss_function_type;
  A.loss_width = O->robust_loss_width;
  A.jacobi_scaling = O->jacobi_scaling;
  A.max_num_iterations = O->max_num_iterations;
  A.max_num_consecutive_invalid_steps = O->max_num_consecutive_invalid_steps;
  A.function_tolerance = O->function_tolerance;
  A.gradient_tolerance = O->gradient_tolerance;
  A.parameter_tolerance = O->parameter_tolerance;
  A.initial_radius = O->initial_trust_region_radius;
  A.max_radius = O->max_trust_region_radius;
  A.min_radius = O->min_trust_region_radius;
  A.min_relative_decrease = O->min_relative_decrease;
  A.lm_lo = O->min_lm_diagonal;
  A.lm_hi = O->max_lm_diagonal;
  hipEvent_t ea, eb;
  TMI_HIP(hipEventCreate(&ea));
  TMI_HIP(hipEventCreate(&eb));
  TMI_HIP(hipEventRecord(ea, s->stream));
  prepare_cameras(s, s->v.ext, s->v.intr, s->v.prep);  // (inside the timed region: part of the call's device work)
  if (st.nslices > 0) {
    if (s->DP == 3 && s->v.uniform_pinhole_default)
      hipLaunchKernelGGL((track_lm_kernel<3, 0>), dim3(s->nblocks_tracks), dim3(256), 0, s->stream, s->v, s->v.prep, A,
                         s->d_trk_term, s->d_trk_iter, s->d_trk_c0, s->d_trk_c1);
    else if (s->DP == 3)
      hipLaunchKernelGGL(track_lm_kernel<3>, dim3(s->nblocks_tracks), dim3(256), 0, s->stream, s->v, s->v.prep, A,
                         s->d_trk_term, s->d_trk_iter, s->d_trk_c0, s->d_trk_c1);
    else if (s->v.uniform_pinhole_default)
      hipLaunchKernelGGL((track_lm_kernel<4, 0>), dim3(s->nblocks_tracks), dim3(256), 0, s->stream, s->v, s->v.prep, A,
                         s->d_trk_term, s->d_trk_iter, s->d_trk_c0, s->d_trk_c1);
    else
      hipLaunchKernelGGL(track_lm_kernel<4>, dim3(s->nblocks_tracks), dim3(256), 0, s->stream, s->v, s->v.prep, A,
                         s->d_trk_term, s->d_trk_iter, s->d_trk_c0, s->d_trk_c1);
  }
  TMI_HIP(hipEventRecord(eb, s->stream));
  const size_t n = (size_t)st.Np_pad;
  std::vector<signed char> term(n);
  std::vector<int> iters(n);
  std::vector<double> c0(track_initial_cost ? n : 0), c1(track_final_cost ? n : 0);
  if (n) {
    TMI_HIP(hipMemcpyAsync(term.data(), s->d_trk_term, n, hipMemcpyDeviceToHost, s->stream));
    TMI_HIP(hipMemcpyAsync(iters.data(), s->d_trk_iter, n * sizeof(int), hipMemcpyDeviceToHost, s->stream));
    if (!c0.empty()) TMI_HIP(hipMemcpyAsync(c0.data(), s->d_trk_c0, n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
    if (!c1.empty()) TMI_HIP(hipMemcpyAsync(c1.data(), s->d_trk_c1, n * sizeof(double), hipMemcpyDeviceToHost, s->stream));
  }
  TMI_HIP(hipStreamSynchronize(s->stream));
  float ms = 0.f;
  hipEventElapsedTime(&ms, ea, eb);
  hipEventDestroy(ea);
  hipEventDestroy(eb);
  for (int lp = 0; lp < st.Np_pad; ++lp) {
    const int p = st.pt_orig[lp];
    if (p < 0) continue;
    const int t = term[lp];
    if (t >= 0) {
      sum->num_tracks++;
      if (t == 0 || t == 1) sum->num_success++;
      sum->total_iterations += iters[lp];
    }
    if (track_termination) track_termination[p] = (int8_t)t;
    if (track_iterations) track_iterations[p] = iters[lp];
    if (track_initial_cost) track_initial_cost[p] = c0[lp];
    if (track_final_cost) track_final_cost[p] = c1[lp];
  }
  for (const int p : st.unobserved) {
    if (track_termination) track_termination[p] = -1;
    if (track_iterations) track_iterations[p] = 0;
    if (track_initial_cost) track_initial_cost[p] = 0.0;
    if (track_final_cost) track_final_cost[p] = 0.0;
  }
  sum->kernel_seconds = ms * 1e-3;
  sum->seconds = now_s() - t0;
  return TMI_BA_OK;
}

int32_t tmi_ba_adjust_tracks(tmi_ba_problem* P, const tmi_ba_options* O, int8_t* track_termination,
                             int32_t* track_iterations, double* track_initial_cost,
                             double* track_final_cost, tmi_ba_track_batch_summary* sum) {
  if (!P || !O || !sum) return TMI_BA_ERR_INVALID_ARGUMENT;
  memset(sum, 0, sizeof(*sum));
  const double t0 = now_s();
  tmi_ba_solver* s = new tmi_ba_solver();
  int rc = create_impl(s, P, O, 0, 1, /*light=*/true);
  if (rc == TMI_BA_OK) {
    rc = tmi_ba_solver_adjust_tracks(s, O, track_termination, track_iterations, track_initial_cost,
                                     track_final_cost, sum);
    if (rc == TMI_BA_OK) {
      rc = tmi_ba_solver_download(s, P);  // cameras are constant here: only points changed
    }
  } else {
    g_last_error = s->error;
  }
  tmi_ba_solver_destroy(s);
  sum->seconds = now_s() - t0;
  return rc;
}

// BundleAdjustTwoViewsAngular for a batch of view pairs (two_view_kernels.h)
int32_t tmi_ba_adjust_two_views_angular(tmi_ba_two_view_angular_batch* Bh, int32_t max_num_iterations, int32_t device,
                                        int8_t* pair_termination, int32_t* pair_iterations,
                                        double* pair_initial_cost, double* pair_final_cost,
                                        tmi_ba_track_batch_summary* sum) {
  if (!Bh || !sum) return TMI_BA_ERR_INVALID_ARGUMENT;
  memset(sum, 0, sizeof(*sum));
  if (Bh->num_pairs < 0 || max_num_iterations < 0) return TMI_BA_ERR_INVALID_ARGUMENT;
  const int P = Bh->num_pairs;
  if (P > 0 && (!Bh->rotation2 || !Bh->position2 || !Bh->correspondence_ptr)) return TMI_BA_ERR_INVALID_ARGUMENT;
  const double t0 = now_s();
  const int64_t N = P ? Bh->correspondence_ptr[P] : 0;
  for (int p = 0; p < P; ++p)
    if (Bh->correspondence_ptr[p + 1] < Bh->correspondence_ptr[p]) return TMI_BA_ERR_INVALID_ARGUMENT;
  if (N > 0 && (!Bh->features1 || !Bh->features2)) return TMI_BA_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_last_error = "no HIP device visible (the device path has no CPU fallback)";
    return TMI_BA_ERR_NO_DEVICE;
  }
  if (device >= ndev) return TMI_BA_ERR_INVALID_ARGUMENT;
  if (P == 0) return TMI_BA_OK;
  tmi_ba_solver* s = new tmi_ba_solver();  // holder of the allocations (freed by tmi_ba_solver_destroy)
  s->light = true;
  auto done = [&](int rc) {
    if (rc != TMI_BA_OK) g_last_error = s->error;
    tmi_ba_solver_destroy(s);
    sum->seconds = now_s() - t0;
    return rc;
  };
  if (device >= 0) s->device = device;
  else if (hipGetDevice(&s->device) != hipSuccess) return done(TMI_BA_ERR_DEVICE);
  if (hipSetDevice(s->device) != hipSuccess) return done(TMI_BA_ERR_DEVICE);
  if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) return done(TMI_BA_ERR_DEVICE);
  int rc;
  auto up = [&](double** dst, const double* src, size_t n) -> int {
    double* d = nullptr;
    int r = dev_alloc(s, &d, n);
    if (r) return r;
    if (n && hipMemcpyAsync(d, src, n * sizeof(double), hipMemcpyHostToDevice, s->stream) != hipSuccess) {
      s->error = "hipMemcpyAsync failed";
      return TMI_BA_ERR_DEVICE;
    }
    *dst = d;
    return TMI_BA_OK;
  };
  std::vector<long long> cptr(Bh->correspondence_ptr, Bh->correspondence_ptr + P + 1);
  TwoViewAngularBatch B;
  memset(&B, 0, sizeof(B));
  B.num_pairs = P;
  double *d_rot, *d_pos, *d_f1, *d_f2;
  long long* d_cptr = nullptr;
  if ((rc = up(&d_rot, Bh->rotation2, (size_t)3 * P))) return done(rc);
  if ((rc = up(&d_pos, Bh->position2, (size_t)3 * P))) return done(rc);
  if ((rc = up(&d_f1, Bh->features1, (size_t)2 * N))) return done(rc);
  if ((rc = up(&d_f2, Bh->features2, (size_t)2 * N))) return done(rc);
  if ((rc = dev_alloc(s, &d_cptr, (size_t)P + 1))) return done(rc);
  if (hipMemcpyAsync(d_cptr, cptr.data(), ((size_t)P + 1) * sizeof(long long), hipMemcpyHostToDevice, s->stream) != hipSuccess)
    return done(TMI_BA_ERR_DEVICE);
  B.rot2 = d_rot; B.pos2 = d_pos; B.corr_ptr = d_cptr; B.feat1 = d_f1; B.feat2 = d_f2;
  signed char* d_term;
  int* d_iter;
  double *d_c0, *d_cf;
  if ((rc = dev_alloc(s, &d_term, (size_t)P))) return done(rc);
  if ((rc = dev_alloc(s, &d_iter, (size_t)P))) return done(rc);
  if ((rc = dev_alloc(s, &d_c0, (size_t)P))) return done(rc);
  if ((rc = dev_alloc(s, &d_cf, (size_t)P))) return done(rc);
  TwoViewArgs A;
  memset(&A, 0, sizeof(A));
  A.max_num_iterations = max_num_iterations;
  A.jacobi_scaling = 1;
  // Ceres Solver::Options defaults (SetSolverOptions, bundle_adjust_two_views.cc:57-69, overrides none of these)
  A.function_tolerance = 1e-6;
  A.gradient_tolerance = 1e-10;
  A.parameter_tolerance = 1e-8;
  A.initial_radius = 1e4;
  A.max_radius = 1e16;
  A.min_radius = 1e-32;
  A.min_relative_decrease = 1e-3;
  A.lm_lo = 1e-6;
  A.lm_hi = 1e32;
  A.max_num_consecutive_invalid_steps = 5;
  hipEvent_t ea, eb;
  if (hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) return done(TMI_BA_ERR_DEVICE);
  hipEventRecord(ea, s->stream);
  hipLaunchKernelGGL(two_view_angular_kernel, dim3((P + 3) / 4), dim3(256), 0, s->stream, B, A, d_term, d_iter, d_c0, d_cf);
  hipEventRecord(eb, s->stream);
  std::vector<signed char> term((size_t)P);
  std::vector<int> iters((size_t)P);
  std::vector<double> c0((size_t)P), cf((size_t)P), rot((size_t)3 * P), pos((size_t)3 * P);
  bool okc = hipMemcpyAsync(term.data(), d_term, (size_t)P, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  okc = okc && hipMemcpyAsync(iters.data(), d_iter, (size_t)P * sizeof(int), hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  okc = okc && hipMemcpyAsync(c0.data(), d_c0, (size_t)P * 8, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  okc = okc && hipMemcpyAsync(cf.data(), d_cf, (size_t)P * 8, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  okc = okc && hipMemcpyAsync(rot.data(), d_rot, rot.size() * 8, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  okc = okc && hipMemcpyAsync(pos.data(), d_pos, pos.size() * 8, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  const hipError_t se = hipStreamSynchronize(s->stream);
  float ms = 0.f;
  hipEventElapsedTime(&ms, ea, eb);
  hipEventDestroy(ea);
  hipEventDestroy(eb);
  if (!okc || se != hipSuccess) {
    s->error = std::string("angular two-view batch failed on the device: ") + hipGetErrorString(se);
    return done(TMI_BA_ERR_DEVICE);
  }
  for (int p = 0; p < P; ++p) {
    const int t = term[p];
    if (t >= 0) {
      sum->num_tracks++;
      if (t == 0 || t == 1) sum->num_success++;
      sum->total_iterations += iters[p];
    }
    if (t == 0 || t == 1) {  // termination != FAILURE: write back
      for (int a = 0; a < 3; ++a) {
        Bh->rotation2[(size_t)3 * p + a] = rot[(size_t)3 * p + a];
        Bh->position2[(size_t)3 * p + a] = pos[(size_t)3 * p + a];
      }
    }
    if (pair_termination) pair_termination[p] = (int8_t)t;
    if (pair_iterations) pair_iterations[p] = iters[p];
    if (pair_initial_cost) pair_initial_cost[p] = c0[p];
    if (pair_final_cost) pair_final_cost[p] = cf[p];
  }
  sum->kernel_seconds = ms * 1e-3;
  return done(TMI_BA_OK);
}

// SelectGoodTracksForBundleAdjustment (select_good_tracks_for_bundle_adjustment.cc:251-327):
// the projections (track statistics) run on the device, the per-view grid / ranking logic --
// integer compares over the view's feature list -- on the host.
int32_t tmi_ba_solver_select_good_tracks(tmi_ba_solver* s, int32_t long_track_length_threshold,
                                         int32_t image_grid_cell_size_pixels,
                                         int32_t min_num_optimized_tracks_per_view,
                                         const uint8_t* view_mask, uint8_t* selected,
                                         int32_t* stats_len, double* stats_err,
                                         tmi_ba_select_summary* sum) {
  if (!s || !sum || !selected || image_grid_cell_size_pixels <= 0) return TMI_BA_ERR_INVALID_ARGUMENT;
  memset(sum, 0, sizeof(*sum));
  const Structure& st = s->st;
  if (st.world > 1) {
    g_last_error = s->error = "track selection ranks every view's tracks: run it on an unsharded handle";
    return TMI_BA_ERR_UNSUPPORTED;
  }
  const double t0 = now_s();
  TMI_HIP(hipSetDevice(s->device));
  int rc = ensure_track_outputs(s);
  if (rc) return rc;
  hipEvent_t ea, eb;
  TMI_HIP(hipEventCreate(&ea));
  TMI_HIP(hipEventCreate(&eb));
  TMI_HIP(hipEventRecord(ea, s->stream));
  prepare_cameras(s, s->v.ext, s->v.intr, s->v.prep);
  if (st.nslices > 0)
    hipLaunchKernelGGL(track_stats_kernel, dim3(s->nblocks_tracks), dim3(256), 0, s->stream, s->v, s->v.prep,
                       s->d_trk_iter, s->d_trk_mean);
  TMI_HIP(hipEventRecord(eb, s->stream));
  const int Np = st.Np_total, Nc = st.Nc;
  hipStream_t stream = s->stream;
  // static per handle: every view's tracks sorted by track index (device radix sort)
  if (!s->d_vt_ptr) {
    if ((rc = dev_alloc(s, &s->d_vt_ptr, (size_t)Nc + 2))) return rc;
    if ((rc = dev_alloc(s, &s->d_vt_keys, (size_t)std::max<int64_t>(st.No_pad, 1)))) return rc;
    if ((rc = dev_alloc(s, &s->d_vbox, (size_t)std::max(Nc, 1) * 4))) return rc;
    if ((rc = dev_alloc(s, &s->d_cell_off, (size_t)Nc + 2))) return rc;
    if ((rc = dev_alloc(s, &s->d_sel, (size_t)std::max(Np, 1)))) return rc;
    if ((rc = dev_alloc(s, &s->d_view_mask, (size_t)std::max(Nc, 1)))) return rc;
    if ((rc = dev_alloc(s, &s->d_vcount, (size_t)2 * std::max(Nc, 1)))) return rc;
    if (st.No_pad > 0) {
      unsigned long long* keys_in = nullptr;
      TMI_HIP(hipMalloc((void**)&keys_in, (size_t)st.No_pad * sizeof(unsigned long long)));
      hipLaunchKernelGGL(select_keys_kernel, dim3(s->nblocks_slices), dim3(256), 0, stream, s->v, s->d_pt_orig, keys_in);
      size_t tmp_bytes = 0;
      hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, keys_in, s->d_vt_keys, (int)st.No_pad, 0, 64, stream);
      void* tmp = nullptr;
      TMI_HIP(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 16)));
      hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, keys_in, s->d_vt_keys, (int)st.No_pad, 0, 64, stream);
      TMI_HIP(hipStreamSynchronize(stream));
      hipFree(tmp);
      hipFree(keys_in);
    }
    hipLaunchKernelGGL(select_view_ptr_kernel, dim3((Nc + 1 + 255) / 256), dim3(256), 0, stream, s->d_vt_keys,
                       (long long)st.No_pad, Nc, s->d_vt_ptr);
  }
  SelectView S;
  memset(&S, 0, sizeof(S));
  S.Nc = Nc;
  S.Np_total = Np;
  S.view_mask = nullptr;
  if (view_mask && Nc > 0) {
    TMI_HIP(hipMemcpyAsync(s->d_view_mask, view_mask, (size_t)Nc, hipMemcpyHostToDevice, stream));
    S.view_mask = s->d_view_mask;
  }
  S.pt_orig = s->d_pt_orig;
  S.cnt = s->d_trk_iter;
  S.mean = s->d_trk_mean;
  S.long_thr = long_track_length_threshold;
  S.inv_cell = 1.0 / image_grid_cell_size_pixels;
  S.vbox = s->d_vbox;
  S.cell_off = s->d_cell_off;
  S.sel = s->d_sel;
  S.counters = s->d_counters;
  const int nb_init = (std::max(std::max(Nc, Np), 4) + 255) / 256;
  hipLaunchKernelGGL(select_init_kernel, dim3(nb_init), dim3(256), 0, stream, S);
  if (st.nslices > 0) hipLaunchKernelGGL(select_bounds_kernel, dim3(s->nblocks_tracks), dim3(256), 0, stream, s->v, S);
  hipLaunchKernelGGL(select_offsets_kernel, dim3(1), dim3(1024), 0, stream, S);
  TMI_HIP(hipMemcpyAsync(s->h_cell_total, s->d_cell_off + Nc, sizeof(long long), hipMemcpyDeviceToHost, stream));
  TMI_HIP(hipStreamSynchronize(stream));
  const long long ncells = *s->h_cell_total;
  if (ncells > ((long long)1 << 31)) {
    g_last_error = s->error = "track selection: the image grids need more than 2^31 cells (cell size too small "
                              "for the pixel range of the features)";
    hipEventDestroy(ea);
    hipEventDestroy(eb);
    return TMI_BA_ERR_UNSUPPORTED;
  }
  if (ncells > s->cell_capacity) {
    for (void* p : s->cell_allocs) hipFree(p);
    s->cell_allocs.clear();
    s->cell_capacity = 0;
    const size_t cap = (size_t)(ncells + ncells / 4 + 1024);
    TMI_HIP(hipMalloc((void**)&s->d_cell_len, cap * sizeof(unsigned)));
    s->cell_allocs.push_back(s->d_cell_len);
    TMI_HIP(hipMalloc((void**)&s->d_cell_err, cap * sizeof(unsigned long long)));
    s->cell_allocs.push_back(s->d_cell_err);
    TMI_HIP(hipMalloc((void**)&s->d_cell_trk, cap * sizeof(unsigned)));
    s->cell_allocs.push_back(s->d_cell_trk);
    s->cell_capacity = (long long)cap;
  }
  S.cell_len = s->d_cell_len;
  S.cell_err = s->d_cell_err;
  S.cell_trk = s->d_cell_trk;
  if (ncells > 0) {
    const unsigned nbc = (unsigned)((ncells + 255) / 256);
    hipLaunchKernelGGL(select_fill_cells_kernel, dim3(nbc), dim3(256), 0, stream, S, ncells);
    hipLaunchKernelGGL(select_cells_kernel<1>, dim3(s->nblocks_tracks), dim3(256), 0, stream, s->v, S);
    hipLaunchKernelGGL(select_cells_kernel<2>, dim3(s->nblocks_tracks), dim3(256), 0, stream, s->v, S);
    hipLaunchKernelGGL(select_cells_kernel<3>, dim3(s->nblocks_tracks), dim3(256), 0, stream, s->v, S);
    hipLaunchKernelGGL(select_mark_kernel, dim3(nbc), dim3(256), 0, stream, S, ncells);
  }
  if (Nc > 0 && st.No_pad > 0) {
    hipLaunchKernelGGL(select_view_count_kernel, dim3(Nc), dim3(256), 0, stream, S, s->d_vt_keys, s->d_vt_ptr, s->d_vcount);
    // the flags of all tracks as a bit vector in LDS when they fit beside the kernel's static 4 KB
    const size_t bit_bytes = ((size_t)Np + 31) / 32 * 4;
    if (bit_bytes <= 152 * 1024) {
      static bool attr_set = false;
      if (!attr_set) {
        TMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&select_topup_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        attr_set = true;
      }
      hipLaunchKernelGGL(select_topup_kernel<true>, dim3(1), dim3(kTopupThreads), bit_bytes, stream, S, s->d_vt_keys,
                         s->d_vt_ptr, s->d_vcount, s->d_vcount + Nc, min_num_optimized_tracks_per_view);
    } else {
      hipLaunchKernelGGL(select_topup_kernel<false>, dim3(1), dim3(kTopupThreads), 0, stream, S, s->d_vt_keys, s->d_vt_ptr,
                         s->d_vcount, s->d_vcount + Nc, min_num_optimized_tracks_per_view);
    }
  }
  if (Np > 0) hipLaunchKernelGGL(select_finish_kernel, dim3((Np + 255) / 256), dim3(256), 0, stream, S, s->d_out_u8);
  if ((stats_len || stats_err) && st.Np_pad > 0) {
    // tracks without observations keep length 0 / NaN error
    if (stats_len) TMI_HIP(hipMemsetAsync(s->d_out_i32, 0, (size_t)Np * sizeof(int), stream));
    if (stats_err) TMI_HIP(hipMemsetAsync(s->d_out_f64, 0xff, (size_t)Np * sizeof(double), stream));
    hipLaunchKernelGGL(scatter_track_stats_kernel, dim3((st.Np_pad + 255) / 256), dim3(256), 0, stream, s->d_pt_orig,
                       st.Np_pad, s->d_trk_iter, s->d_trk_mean, long_track_length_threshold,
                       stats_len ? s->d_out_i32 : nullptr, stats_err ? s->d_out_f64 : nullptr);
  }
  TMI_HIP(hipMemcpyAsync(s->h_counters, s->d_counters, 4 * sizeof(int), hipMemcpyDeviceToHost, stream));
  unsigned char* stage_u8 = s->h_stage;
  int* stage_i32 = reinterpret_cast<int*>(s->h_stage + 4 * (size_t)std::max(Np, 1));
  double* stage_f64 = reinterpret_cast<double*>(s->h_stage + 8 * (size_t)std::max(Np, 1));
  if (Np > 0) TMI_HIP(hipMemcpyAsync(stage_u8, s->d_out_u8, (size_t)Np, hipMemcpyDeviceToHost, stream));
  if (stats_len && Np > 0) TMI_HIP(hipMemcpyAsync(stage_i32, s->d_out_i32, (size_t)Np * sizeof(int), hipMemcpyDeviceToHost, stream));
  if (stats_err && Np > 0) TMI_HIP(hipMemcpyAsync(stage_f64, s->d_out_f64, (size_t)Np * sizeof(double), hipMemcpyDeviceToHost, stream));
  TMI_HIP(hipStreamSynchronize(stream));
  if (Np > 0) memcpy(selected, stage_u8, (size_t)Np);
  if (stats_len && Np > 0) memcpy(stats_len, stage_i32, (size_t)Np * sizeof(int));
  if (stats_err && Np > 0) memcpy(stats_err, stage_f64, (size_t)Np * sizeof(double));
  float ms = 0.f;
  hipEventElapsedTime(&ms, ea, eb);
  hipEventDestroy(ea);
  hipEventDestroy(eb);
  sum->kernel_seconds = ms * 1e-3;
  if (stats_err)  // 0xff.. is a NaN pattern; make it the quiet NaN the host path produced
    for (const int p : st.unobserved) stats_err[p] = std::nan("");
  sum->num_tracks = Np;
  sum->num_selected_grid = s->h_counters[0];
  sum->num_selected = s->h_counters[1];
  sum->seconds = now_s() - t0;
  return TMI_BA_OK;
}

int32_t tmi_ba_select_good_tracks(const tmi_ba_problem* P, int32_t device,
                                  int32_t long_track_length_threshold,
                                  int32_t image_grid_cell_size_pixels,
                                  int32_t min_num_optimized_tracks_per_view,
                                  const uint8_t* view_mask, uint8_t* selected,
                                  int32_t* stats_len, double* stats_err, tmi_ba_select_summary* sum) {
  if (!P || !sum) return TMI_BA_ERR_INVALID_ARGUMENT;
  memset(sum, 0, sizeof(*sum));
  const double t0 = now_s();
  tmi_ba_options O;
  tmi_ba_options_init(&O);
  O.device = device;
  tmi_ba_solver* s = new tmi_ba_solver();
  int rc = create_impl(s, P, &O, 0, 1, /*light=*/true);
  if (rc == TMI_BA_OK)
    rc = tmi_ba_solver_select_good_tracks(s, long_track_length_threshold, image_grid_cell_size_pixels,
                                          min_num_optimized_tracks_per_view, view_mask, selected,
                                          stats_len, stats_err, sum);
  else
    g_last_error = s->error;
  tmi_ba_solver_destroy(s);
  sum->seconds = now_s() - t0;
  return rc;
}

// ---- batched two-view bundle adjustment ------------------------------
int32_t tmi_ba_adjust_two_views(tmi_ba_two_view_batch* Bh, int32_t point_dof, int32_t max_num_iterations,
                                int32_t device, int8_t* pair_termination, int32_t* pair_iterations,
                                double* pair_initial_cost, double* pair_final_cost,
                                tmi_ba_track_batch_summary* sum) {
  if (!Bh || !sum) return TMI_BA_ERR_INVALID_ARGUMENT;
  memset(sum, 0, sizeof(*sum));
  if ((point_dof != 3 && point_dof != 4) || Bh->num_pairs < 0 || max_num_iterations < 0)
    return TMI_BA_ERR_INVALID_ARGUMENT;
  const int P = Bh->num_pairs;
  if (P > 0 && (!Bh->extrinsics1 || !Bh->extrinsics2 || !Bh->model1 || !Bh->model2 || !Bh->intrinsics1 ||
                !Bh->intrinsics2 || !Bh->correspondence_ptr))
    return TMI_BA_ERR_INVALID_ARGUMENT;
  const double t0 = now_s();
  const int64_t N = P ? Bh->correspondence_ptr[P] : 0;
  for (int p = 0; p < P; ++p) {
    if (Bh->correspondence_ptr[p + 1] < Bh->correspondence_ptr[p] || Bh->model1[p] < 0 || Bh->model1[p] > 4 ||
        Bh->model2[p] < 0 || Bh->model2[p] > 4)
      return TMI_BA_ERR_INVALID_ARGUMENT;
  }
  if (N > 0 && (!Bh->features1 || !Bh->features2 || !Bh->points)) return TMI_BA_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_last_error = "no HIP device visible (the device path has no CPU fallback)";
    return TMI_BA_ERR_NO_DEVICE;
  }
  if (device >= ndev) return TMI_BA_ERR_INVALID_ARGUMENT;
  if (P == 0) return TMI_BA_OK;
  // a throw-away holder for the allocations (freed by tmi_ba_solver_destroy)
  tmi_ba_solver* s = new tmi_ba_solver();
  s->light = true;
  auto done = [&](int rc) {
    if (rc != TMI_BA_OK) g_last_error = s->error;
    tmi_ba_solver_destroy(s);
    sum->seconds = now_s() - t0;
    return rc;
  };
  if (device >= 0) s->device = device;
  else if (hipGetDevice(&s->device) != hipSuccess) return done(TMI_BA_ERR_DEVICE);
  if (hipSetDevice(s->device) != hipSuccess) return done(TMI_BA_ERR_DEVICE);
  if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) return done(TMI_BA_ERR_DEVICE);
  int rc;
  TwoViewBatch B;
  memset(&B, 0, sizeof(B));
  B.num_pairs = P;
  auto up = [&](auto** dst, const auto* src, size_t n) -> int {
    typedef typename std::remove_const<typename std::remove_pointer<decltype(src)>::type>::type T;
    T* d = nullptr;
    int r = dev_alloc(s, &d, n);
    if (r) return r;
    if (n && hipMemcpyAsync(d, src, n * sizeof(T), hipMemcpyHostToDevice, s->stream) != hipSuccess) {
      s->error = "hipMemcpyAsync failed";
      return TMI_BA_ERR_DEVICE;
    }
    *dst = d;
    return TMI_BA_OK;
  };
  std::vector<unsigned char> c1((size_t)P, 1), c2((size_t)P, 1);
  if (Bh->constant_intrinsics1) c1.assign(Bh->constant_intrinsics1, Bh->constant_intrinsics1 + P);
  if (Bh->constant_intrinsics2) c2.assign(Bh->constant_intrinsics2, Bh->constant_intrinsics2 + P);
  std::vector<long long> cptr(Bh->correspondence_ptr, Bh->correspondence_ptr + P + 1);
  double *d_e1, *d_e2, *d_k1, *d_k2, *d_f1, *d_f2, *d_pts;
  int *d_m1, *d_m2;
  unsigned char *d_c1, *d_c2;
  long long* d_cptr;
  const size_t Nn = (size_t)std::max<int64_t>(N, 1);
  if ((rc = up(&d_e1, Bh->extrinsics1, (size_t)6 * P))) return done(rc);
  if ((rc = up(&d_e2, (const double*)Bh->extrinsics2, (size_t)6 * P))) return done(rc);
  if ((rc = up(&d_m1, Bh->model1, (size_t)P))) return done(rc);
  if ((rc = up(&d_m2, Bh->model2, (size_t)P))) return done(rc);
  if ((rc = up(&d_k1, (const double*)Bh->intrinsics1, (size_t)10 * P))) return done(rc);
  if ((rc = up(&d_k2, (const double*)Bh->intrinsics2, (size_t)10 * P))) return done(rc);
  if ((rc = up(&d_c1, (const unsigned char*)c1.data(), (size_t)P))) return done(rc);
  if ((rc = up(&d_c2, (const unsigned char*)c2.data(), (size_t)P))) return done(rc);
  if ((rc = up(&d_cptr, (const long long*)cptr.data(), (size_t)P + 1))) return done(rc);
  if ((rc = up(&d_f1, Bh->features1, (size_t)2 * N))) return done(rc);
  if ((rc = up(&d_f2, Bh->features2, (size_t)2 * N))) return done(rc);
  if ((rc = up(&d_pts, (const double*)Bh->points, (size_t)4 * N))) return done(rc);
  B.ext1 = d_e1; B.ext2 = d_e2; B.model1 = d_m1; B.model2 = d_m2; B.intr1 = d_k1; B.intr2 = d_k2;
  B.const1 = d_c1; B.const2 = d_c2; B.corr_ptr = d_cptr; B.feat1 = d_f1; B.feat2 = d_f2; B.points = d_pts;
  if ((rc = dev_alloc(s, &B.points_c, 4 * Nn))) return done(rc);
  if ((rc = dev_alloc(s, &B.scale_p, 4 * Nn))) return done(rc);
  signed char* d_term;
  int* d_iter;
  double *d_c0, *d_cf;
  if ((rc = dev_alloc(s, &d_term, (size_t)P))) return done(rc);
  if ((rc = dev_alloc(s, &d_iter, (size_t)P))) return done(rc);
  if ((rc = dev_alloc(s, &d_c0, (size_t)P))) return done(rc);
  if ((rc = dev_alloc(s, &d_cf, (size_t)P))) return done(rc);
  TwoViewArgs A;
  A.point_dof = point_dof;
  A.max_num_iterations = max_num_iterations;
  A.jacobi_scaling = 1;
  // Ceres Solver::Options defaults: bundle_adjust_two_views.cc:58-68 overrides none of these
  A.function_tolerance = 1e-6;
  A.gradient_tolerance = 1e-10;
  A.parameter_tolerance = 1e-8;
  A.initial_radius = 1e4;
  A.max_radius = 1e16;
  A.min_radius = 1e-32;
  A.min_relative_decrease = 1e-3;
  A.lm_lo = 1e-6;
  A.lm_hi = 1e32;
  A.max_num_consecutive_invalid_steps = 5;
  hipEvent_t ea, eb;
  if (hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) return done(TMI_BA_ERR_DEVICE);
  hipEventRecord(ea, s->stream);
  if (point_dof == 3)
    hipLaunchKernelGGL(two_view_lm_kernel<3>, dim3((P + 3) / 4), dim3(256), 0, s->stream, B, A, d_term, d_iter, d_c0, d_cf);
  else
    hipLaunchKernelGGL(two_view_lm_kernel<4>, dim3((P + 3) / 4), dim3(256), 0, s->stream, B, A, d_term, d_iter, d_c0, d_cf);
  hipEventRecord(eb, s->stream);
  std::vector<signed char> term((size_t)P);
  std::vector<int> iters((size_t)P);
  std::vector<double> c0((size_t)P), cf((size_t)P), e2((size_t)6 * P), k1((size_t)10 * P), k2((size_t)10 * P),
      pts((size_t)4 * N);
  bool okc = hipMemcpyAsync(term.data(), d_term, (size_t)P, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  okc = okc && hipMemcpyAsync(iters.data(), d_iter, (size_t)P * sizeof(int), hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  okc = okc && hipMemcpyAsync(c0.data(), d_c0, (size_t)P * 8, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  okc = okc && hipMemcpyAsync(cf.data(), d_cf, (size_t)P * 8, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  okc = okc && hipMemcpyAsync(e2.data(), d_e2, e2.size() * 8, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  okc = okc && hipMemcpyAsync(k1.data(), d_k1, k1.size() * 8, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  okc = okc && hipMemcpyAsync(k2.data(), d_k2, k2.size() * 8, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  if (N) okc = okc && hipMemcpyAsync(pts.data(), d_pts, pts.size() * 8, hipMemcpyDeviceToHost, s->stream) == hipSuccess;
  const hipError_t se = hipStreamSynchronize(s->stream);
  float ms = 0.f;
  hipEventElapsedTime(&ms, ea, eb);
  hipEventDestroy(ea);
  hipEventDestroy(eb);
  if (!okc || se != hipSuccess) {
    s->error = std::string("two-view batch failed on the device: ") + hipGetErrorString(se);
    return done(TMI_BA_ERR_DEVICE);
  }
  for (int p = 0; p < P; ++p) {
    const int t = term[p];
    if (t >= 0) {
      sum->num_tracks++;
      if (t == 0 || t == 1) sum->num_success++;
      sum->total_iterations += iters[p];
    }
    if (t == 0 || t == 1) {  // IsSolutionUsable: write back
      for (int a = 0; a < 6; ++a) Bh->extrinsics2[(size_t)6 * p + a] = e2[(size_t)6 * p + a];
      Bh->intrinsics1[(size_t)10 * p] = k1[(size_t)10 * p];
      Bh->intrinsics2[(size_t)10 * p] = k2[(size_t)10 * p];
      for (int64_t q = Bh->correspondence_ptr[p]; q < Bh->correspondence_ptr[p + 1]; ++q)
        for (int a = 0; a < 4; ++a) Bh->points[4 * q + a] = pts[4 * q + a];
    }
    if (pair_termination) pair_termination[p] = (int8_t)t;
    if (pair_iterations) pair_iterations[p] = iters[p];
    if (pair_initial_cost) pair_initial_cost[p] = c0[p];
    if (pair_final_cost) pair_final_cost[p] = cf[p];
  }
  sum->kernel_seconds = ms * 1e-3;
  return done(TMI_BA_OK);
}

int32_t tmi_ba_structure_stats(const tmi_ba_problem* P, int32_t rank, int32_t world, int64_t out[12]) {
  return tmi_ba_structure_stats_for(P, rank, world, /*forms_S=*/1, out);
}

int32_t tmi_ba_structure_stats_for(const tmi_ba_problem* P, int32_t rank, int32_t world, int32_t forms_S, int64_t out[12]) {
  if (!P || !out) return TMI_BA_ERR_INVALID_ARGUMENT;
  Structure st;
  // the dealing of the slices depends on what the handle will do with them (structure.cpp): work = Schur pairs +
  // 5 x observations where S is formed (want_pairs_mode 1), observations where the operator is matrix-free -- the
  // default of a sharded solve (want_pairs_mode 0)
  const int rc = build_structure(P, rank, world, &st, forms_S ? 1 : 0);
  if (rc != TMI_BA_OK) return rc;
  out[0] = st.Np; out[1] = st.No; out[2] = st.Nrb; out[3] = st.D; out[4] = st.nub;
  out[5] = st.nnzb; out[6] = st.npairs; out[8] = st.nslices; out[9] = st.No_pad;
  uint64_t h = 1469598103934665603ULL;
  for (int64_t u = 0; u < st.nub; ++u) {
    h = (h ^ (uint64_t)st.ub_i[u]) * 1099511628211ULL;
    h = (h ^ (uint64_t)st.ub_j[u]) * 1099511628211ULL;
  }
  out[7] = (int64_t)(h >> 1);
  int64_t osum = 0;
  for (int64_t e = 0; e < st.No_pad; ++e)
    if (st.obs_orig[e] >= 0) osum += st.obs_orig[e] + 1;
  out[10] = osum;
  // pair key independent of slot numbering: caller observation indices of both ends
  std::vector<int64_t> slot_obs((size_t)st.Nslots, -1);
  for (int64_t e = 0; e < st.No_pad; ++e)
    if (st.obs_cpos[e] >= 0) slot_obs[st.obs_cpos[e]] = st.obs_orig[e];
  int64_t psum = 0;
  for (int64_t k = 0; k < st.npairs; ++k)
    psum += (slot_obs[st.pair_i[k]] + 1) * 31 + (slot_obs[st.pair_j[k]] + 1) * 17;
  out[11] = psum;
  return TMI_BA_OK;
}

// Test hook: FNV-1a checksums of the static structure as it sits in HBM (whoever built it), so
// that the device builder can be compared with the host builder array for array.
//   out[0] setup path (1 device, 0 host), [1] slice_ptr, [2] pt_k, [3] pt_const, [4] obs_cam,
//   [5] obs_xy, [6] obs_cpos, [7] cam_ptr, [8] ub_i, [9] ub_j, [10] urow_ptr, [11] ucol_ptr,
//   [12] ucol_u, [13] spc_row, [14] spc_u0, [15] spc_rptr, [16] pair_ptr, [17] pair_i, [18] pair_j,
//   [19] launch headers, [20] pt_orig, [21] n_order, [22] n_spc, [23] Nslots
int32_t tmi_ba_solver_structure_checksums(tmi_ba_solver* s, uint64_t out[24]) {
  if (!s || !out) return TMI_BA_ERR_INVALID_ARGUMENT;
  TMI_HIP(hipSetDevice(s->device));
  TMI_HIP(hipStreamSynchronize(s->stream));
  const Structure& st = s->st;
  const DeviceView& v = s->v;
  auto sum = [&](const void* dev, size_t bytes, uint64_t* dst) -> int {
    std::vector<unsigned char> h(bytes);
    if (bytes) TMI_HIP(hipMemcpy(h.data(), dev, bytes, hipMemcpyDeviceToHost));
    uint64_t x = 1469598103934665603ULL;
    for (unsigned char c : h) x = (x ^ c) * 1099511628211ULL;
    *dst = x;
    return TMI_BA_OK;
  };
  int rc;
  memset(out, 0, 24 * sizeof(uint64_t));
  out[0] = s->device_structure ? 1 : 0;
  const size_t Npad = (size_t)st.Np_pad, Nop = (size_t)st.No_pad, Nrb = (size_t)st.Nrb, nub = (size_t)st.nub;
#define CS(i, ptr, n, T) if ((rc = sum(ptr, (size_t)(n) * sizeof(T), &out[i]))) return rc;
  CS(1, v.slice_ptr, st.nslices + 1, int) CS(2, v.pt_k, Npad, int) CS(3, v.pt_const, Npad, unsigned char)
  CS(4, v.obs_cam, Nop, int) CS(5, v.obs_xy, 2 * Nop, double) CS(6, v.obs_cpos, Nop, int)
  CS(7, v.cam_ptr, Nrb + 1, int) CS(20, s->d_pt_orig, Npad, int)
  if (!s->light && !s->implicit) {
    CS(8, v.ub_i, nub, int) CS(9, v.ub_j, nub, int) CS(10, v.urow_ptr, Nrb + 1, int) CS(11, v.ucol_ptr, Nrb + 1, int)
    CS(12, v.ucol_u, nub, int) CS(13, v.spc_row, v.n_spc, int) CS(14, v.spc_u0, v.n_spc, int)
    CS(15, v.spc_rptr, Nrb + 1, int) CS(16, v.pair_ptr, nub + 1, long long) CS(17, v.pair_i, st.npairs, int)
    CS(18, v.pair_j, st.npairs, int) CS(19, v.ub_order, (size_t)v.n_order * 4, int)
  }
#undef CS
  out[21] = (uint64_t)v.n_order;
  out[22] = (uint64_t)v.n_spc;
  out[23] = (uint64_t)st.Nslots;
  return TMI_BA_OK;
}

int32_t tmi_ba_solver_operator_info(tmi_ba_solver* s, int32_t out[8]) {
  if (!s || !out) return TMI_BA_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < 8; ++i) out[i] = 0;
  out[0] = s->mf_ok ? 1 : 0;
  out[1] = s->v.drop_pos ? 1 : 0;
  out[2] = s->direct_ok ? 1 : 0;
  out[3] = s->adaptive ? 1 : 0;
  out[4] = s->implicit ? 1 : 0;
  out[5] = s->adaptive ? (int32_t)std::min(s->adaptive_break_even, 1 << 30) : 0;
  // can this handle serve CLUSTER_JACOBI with clusters (shared-intrinsics clusters, or visibility clusters built at create)?
  // 0: a CLUSTER_JACOBI request keeps the SCHUR_JACOBI blocks (tmi_ba_summary::effective_preconditioner_type says so per solve)
  out[6] = ((s->st.has_shared || s->vis_clusters) && !s->cl_unavailable) ? 1 : 0;
  // the planes of the LAST linearisation are compact (device_view.h): p_n instead of the stored camera block
  out[7] = s->v.compact ? 1 : 0;
  return TMI_BA_OK;
}

int32_t tmi_ba_solver_evaluate(tmi_ba_solver* s, double* residuals, double* jac_camera,
                               double* jac_shared, double* jac_point, uint8_t* valid,
                               int32_t* block_dim) {
  if (!s || s->light) return TMI_BA_ERR_INVALID_ARGUMENT;
  TMI_HIP(hipSetDevice(s->device));
  {
    const int rcm = materialize_host_layout(s);
    if (rcm) return rcm;
  }
  DeviceView& v = s->v;
  Structure& st = s->st;
  const int D = st.D, DP = s->DP;
  const int n_r = st.Nrb * D;
  hipStream_t stream = s->stream;
  if (block_dim) *block_dim = D;
  TMI_HIP(hipMemsetAsync(v.flags, 0, FL_COUNT * sizeof(int), stream));
  hipLaunchKernelGGL(fill_kernel, dim3((n_r + 255) / 256 + 1), dim3(256), 0, stream, v.scale_c, (long long)n_r, 1.0);
  s->launch.expand_scale(v, stream);
  prepare_cameras(s, v.ext, v.intr, v.prep);
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)(((long long)st.Np_pad * DP + 255) / 256 + 1)), dim3(256), 0, stream, v.scale_p, (long long)st.Np_pad * DP, 1.0);
  // poison the residual planes so that invalid observations can be told apart
  {
    DeviceView ve = v;  // (the caller gets every column: stored, not formed from Jp)
    ve.drop_pos = 0;
    ve.compact = 0;
    v.compact = 0;  // (the planes now hold the full blocks)
    v.sums_ready = 0;
    s->launch.linearize(ve, stream, v.prep, 0, 1.0, s->nblocks_tracks, nullptr, 0);
  }
  const size_t N = (size_t)st.No_pad;
  std::vector<double> r(residuals ? 2 * N : 0), A(jac_camera ? (size_t)2 * D * N : 0),
      Jp(jac_point ? (size_t)2 * DP * N : 0);
  if (residuals) TMI_HIP(hipMemcpyAsync(r.data(), v.pm_r, r.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  if (jac_camera) TMI_HIP(hipMemcpyAsync(A.data(), v.pm_A, A.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  if (jac_point) TMI_HIP(hipMemcpyAsync(Jp.data(), v.pm_Jp, Jp.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  std::vector<double> A1((jac_shared && st.has_shared) ? (size_t)2 * D * N : 0);
  if (!A1.empty()) TMI_HIP(hipMemcpyAsync(A1.data(), v.pm_A1, A1.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
  TMI_HIP(hipStreamSynchronize(stream));
  if (v.planes_fp32) {
    // the planes hold floats (same layout in elements)
    auto widen = [](std::vector<double>& buf) {
      std::vector<float> f(buf.size());
      if (!f.empty()) memcpy(f.data(), buf.data(), f.size() * sizeof(float));
      for (size_t k = 0; k < f.size(); ++k) buf[k] = (double)f[k];
    };
    widen(r); widen(A); widen(Jp); widen(A1);
  }
  for (size_t e = 0; e < N; ++e) {
    const int64_t i = st.obs_orig[e];
    if (i < 0) continue;
    // the planes are tiled by 64 observations (kernels.h pidx)
    auto at = [e](int npl, int plane) {
      return (e >> 6) * (size_t)(npl * 64) + (size_t)(plane >> 1) * 128 + ((e & 63) << 1) + (size_t)(plane & 1);
    };
    if (residuals) {
      residuals[2 * i] = r[at(2, 0)];
      residuals[2 * i + 1] = r[at(2, 1)];
    }
    if (jac_camera)
      for (int a = 0; a < D; ++a) {
        jac_camera[(size_t)2 * D * i + a] = A[at(2 * D, 2 * a)];
        jac_camera[(size_t)2 * D * i + D + a] = A[at(2 * D, 2 * a + 1)];
      }
    if (jac_shared)
      for (int a = 0; a < D; ++a) {
        jac_shared[(size_t)2 * D * i + a] = A1.empty() ? 0.0 : A1[at(2 * D, 2 * a)];
        jac_shared[(size_t)2 * D * i + D + a] = A1.empty() ? 0.0 : A1[at(2 * D, 2 * a + 1)];
      }
    if (jac_point)
      for (int a = 0; a < DP; ++a) {
        jac_point[(size_t)2 * DP * i + a] = Jp[at(2 * DP, 2 * a)];
        jac_point[(size_t)2 * DP * i + DP + a] = Jp[at(2 * DP, 2 * a + 1)];
      }
    if (valid) valid[i] = 1;
  }
  if (valid) {
    // an invalid observation wrote zeros everywhere; flag it from the functor's own predicate
    std::vector<double> ext((size_t)6 * st.Nc), pts((size_t)4 * st.Np_pad);
    TMI_HIP(hipMemcpy(ext.data(), v.ext, ext.size() * sizeof(double), hipMemcpyDeviceToHost));
    TMI_HIP(hipMemcpy(pts.data(), v.pts, pts.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int sl = 0; sl < st.nslices; ++sl)
      for (int t = 0; t < 64; ++t) {
        const int lp = sl * 64 + t;
        for (int j = 0; j < st.pt_k[lp]; ++j) {
          const size_t e = (size_t)st.slice_ptr[sl] + (size_t)j * 64 + t;
          const int64_t i = st.obs_orig[e];
          const int c = st.obs_cam[e];
          double sq = 0.0;
          for (int a = 0; a < 3; ++a) {
            const double d = pts[(size_t)4 * lp + a] - pts[(size_t)4 * lp + 3] * ext[(size_t)6 * c + a];
            sq += d * d;
          }
          valid[i] = sq < 1e-8 ? 0 : 1;
        }
      }
  }
  return TMI_BA_OK;
}

}  // extern "C"
