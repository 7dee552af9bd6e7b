// Exact solve of the reduced camera system for the DENSE_SCHUR / SPARSE_SCHUR /
// DENSE_QR solver types: the explicit Schur complement is scattered into a dense
// n x n matrix (n = #reduced blocks * D, a few thousand below the reference's
// 1000-view switch to ITERATIVE_SCHUR, reconstruction_estimator_utils.cc:110-133)
// and factored with a blocked right-looking Cholesky (lower triangle).  Panels are 32 columns
// wide: the 32 x 32 diagonal block is factored by ONE wavefront out of registers (lane i owns row
// i, pivots and multipliers travel by v_readlane -- no barriers: a 256-thread version
// with three __syncthreads per column took 23 us), the panel below it by a thread per row, both in
// ONE launch per panel (every workgroup factors the diagonal block itself, chol_panel_kernel).  The
// trailing update -- all of the n^3 / 3 flops -- runs once per PAIR of panels as a rank-64 update
// on the f64 matrix cores (v_mfma_f64_16x16x4, 64 x 64 tiles), the second panel of a pair seeing
// the first through a narrow rank-32 update of its own 32 columns.  Forward / backward
// substitution take one launch per 64 columns, the 64 x 64 triangular solve fused into the update
// kernel (every workgroup solves it itself).  In the
// reference this is Ceres' DenseSchurComplementSolver / CHOLMOD
// (SparseSchurComplementSolver); both solve the same normal equations exactly.
#pragma once
#include <hip/hip_runtime.h>

#include "device_view.h"

namespace tmi {

constexpr int kPanel = 32;   // columns factored at a time
constexpr int kTile = 64;    // trailing-update tile / substitution block
constexpr int kTilePitch = kTile + 1;

// scatter the symmetric block storage (upper blocks in `red`, diagonal blocks in
// Sdiag) into the dense lower + upper triangles
template <int D>
__global__ __launch_bounds__(256) void dense_gather_kernel(DeviceView v, const double* __restrict__ ub,
                                                           double* __restrict__ A, int n) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n_off = (long long)v.nub * D * D;
  const long long n_all = n_off + (long long)v.Nrb * D * D;
  if (e >= n_all) return;
  if (e < n_off) {
    const int u = (int)(e / (D * D));
    const int w = (int)(e - (long long)u * (D * D));
    const int a = w / D, b = w - a * D;
    const int row = v.ub_i[u], col = v.ub_j[u];
    const double val = ub[e];
    A[(size_t)(row * D + a) * n + (size_t)col * D + b] = val;
    A[(size_t)(col * D + b) * n + (size_t)row * D + a] = val;
  } else {
    const long long f = e - n_off;
    const int rb = (int)(f / (D * D));
    const int w = (int)(f - (long long)rb * (D * D));
    const int a = w / D, b = w - a * D;
    A[(size_t)(rb * D + a) * n + (size_t)rb * D + b] = v.Sdiag[f];
  }
}

// broadcast of one lane's double to the whole wavefront (scalar registers)
__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src_lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// One panel in ONE launch: every workgroup factors the kPanel x kPanel diagonal block itself (wave 0: lane i holds row i of the block in registers,
// pivots and multipliers by v_readlane, on a copy in LDS -- a dependent chain of ~6 us that costs the same wherever
// it runs) and then solves its 256 rows of the panel against it, so the diagonal factor never crosses workgroups
// inside the launch.  The factored diagonal block goes to `diag` (kPanel x kPanel per panel), NOT back into A:
// other workgroups may still be reading the unfactored block there; chol_diag_store_kernel puts all of them
// into A once the factorisation is over.  Same arithmetic in the same order as the two-kernel form.
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ A, int n, int k0,
                                                         double* __restrict__ diag, int* __restrict__ flag) {
  __shared__ double L[kPanel][kPanel + 1];
  const int nb = min(kPanel, n - k0);
  for (int e = threadIdx.x; e < kPanel * kPanel; e += 256) {
    const int r = e / kPanel, c = e - r * kPanel;
    L[r][c] = (r < nb && c < nb && c <= r) ? A[(size_t)(k0 + r) * n + k0 + c] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    double r[kPanel];
#pragma unroll
    for (int c = 0; c < kPanel; ++c) r[c] = (lane < kPanel && c <= lane) ? L[lane][c] : 0.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < kPanel; ++j) {
      double d = lane_bcast(r[j], j);
      if (!(d > 0.0) || !isfinite(d)) {
        bad = true;
        d = 1.0;
      }
      const double dj = sqrt(d);
      r[j] = (lane == j) ? dj : r[j] / dj;
#pragma unroll
      for (int m = j + 1; m < kPanel; ++m) r[m] -= r[j] * lane_bcast(r[j], m);
    }
    if (bad && lane == 0 && blockIdx.x == 0) *flag = 1;
    if (lane < kPanel) {
      double* dst = diag + (size_t)(k0 / kPanel) * kPanel * kPanel + lane * kPanel;
#pragma unroll
      for (int c = 0; c < kPanel; ++c)
        if (c <= lane) {
          L[lane][c] = r[c];
          if (blockIdx.x == 0) dst[c] = r[c];
        }
    }
  }
  __syncthreads();
  const int i = k0 + nb + blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double* row = A + (size_t)i * n + k0;
  double x[kPanel];
#pragma unroll
  for (int c = 0; c < kPanel; ++c) {
    if (c < nb) {
      double t = row[c];
#pragma unroll
      for (int m = 0; m < c; ++m) t -= x[m] * L[c][m];
      x[c] = t / L[c][c];
    } else {
      x[c] = 0.0;
    }
  }
#pragma unroll
  for (int c = 0; c < kPanel; ++c)
    if (c < nb) row[c] = x[c];
}

// the factored diagonal blocks of every panel, from the side buffer into A
__global__ __launch_bounds__(256) void chol_diag_store_kernel(double* __restrict__ A, int n,
                                                              const double* __restrict__ diag) {
  const int e = blockIdx.x * 256 + threadIdx.x;  // element (row i, column c of its panel)
  if (e >= n * kPanel) return;
  const int i = e / kPanel, c = e - i * kPanel;
  const int p = i / kPanel, r = i - p * kPanel;
  if (c <= r && p * kPanel + c < n) A[(size_t)i * n + p * kPanel + c] = diag[(size_t)p * kPanel * kPanel + r * kPanel + c];
}

// Rank-kw update on the f64 matrix cores: A[i, j] -= sum_{k < kw} A[i, kp + k] A[j, kp + k] for the
// lower-triangle entries with i >= r0, r0 <= j < cmax, one 64 x 64 tile per workgroup (tile rows
// r0 + 64 bx, tile columns r0 + 64 by, by <= bx).  Wave w takes rows [16 w, 16 w + 16) of the tile and
// all four 16-column blocks; K = 64 (zero padded beyond kw) is 16 MFMA steps of 4.
__global__ __launch_bounds__(256) void chol_syrk_kernel(double* __restrict__ A, int n, int kp, int kw, int r0,
                                                        int cmax) {
  if (blockIdx.y > blockIdx.x) return;
  typedef double v4f64c __attribute__((ext_vector_type(4)));
  __shared__ double Pi[kTile][kTilePitch], Pj[kTile][kTilePitch];
  const int i0 = r0 + blockIdx.x * kTile, j0 = r0 + blockIdx.y * kTile;
  if (j0 >= cmax) return;
  for (int e = threadIdx.x; e < kTile * kTile; e += 256) {
    const int r = e / kTile, c = e - r * kTile;
    Pi[r][c] = (i0 + r < n && c < kw) ? A[(size_t)(i0 + r) * n + kp + c] : 0.0;
    Pj[r][c] = (j0 + r < n && c < kw) ? A[(size_t)(j0 + r) * n + kp + c] : 0.0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 15, kk = lane >> 4;
  v4f64c acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4f64c){0.0, 0.0, 0.0, 0.0};
  const int ksteps = (kw + 3) / 4;
#pragma unroll
  for (int s = 0; s < kTile / 4; ++s) {
    if (s < ksteps) {
      const double a = Pi[16 * w + li][4 * s + kk];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double b = Pj[16 * t + li][4 * s + kk];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  // C/D layout of the f64 MFMA: column = lane & 15, row = (lane >> 4) + 4 q
#pragma unroll
  for (int t = 0; t < 4; ++t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = i0 + 16 * w + kk + 4 * q, j = j0 + 16 * t + li;
      if (i < n && j <= i && j < cmax) A[(size_t)i * n + j] -= acc[t][q];
    }
  }
}

// Substitution, one launch per panel: every workgroup solves the panel's 64 x 64 triangular system
// itself (one wavefront, ~1 us) and then updates its share of the remaining right-hand side.
//   forward:  y_k = L_kk^-1 w_k;  w[i] -= L[i, panel] y_k  for the rows below the panel
//   backward: x_k = L_kk^-T y_k;  y[j] -= L[panel, j]^T x_k for the columns left of the panel
// `in` is the working vector (never written inside the panel's own range), `out` receives the solved
// panel entries -- so no workgroup reads what another one writes.
template <bool FWD>
__global__ __launch_bounds__(256) void chol_subst_kernel(const double* __restrict__ A, int n, int k0,
                                                         double* __restrict__ in, double* __restrict__ out) {
  __shared__ double L[kTile][kTilePitch];
  __shared__ double xs[kTile];
  const int nb = min(kTile, n - k0);
  for (int e = threadIdx.x; e < kTile * kTile; e += 256) {
    const int r = e / kTile, c = e - r * kTile;
    L[r][c] = (r < nb && c <= r) ? A[(size_t)(k0 + r) * n + k0 + c] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    double wv = lane < nb ? in[k0 + lane] : 0.0;
    if (FWD) {
      for (int c = 0; c < nb; ++c) {
        const double xc = __shfl(wv, c, 64) / L[c][c];
        if (lane == c) wv = xc;
        if (lane > c) wv -= L[lane][c] * xc;
      }
    } else {
      for (int c = nb - 1; c >= 0; --c) {
        const double xc = __shfl(wv, c, 64) / L[c][c];
        if (lane == c) wv = xc;
        if (lane < c) wv -= L[c][lane] * xc;
      }
    }
    xs[lane] = wv;
    if (blockIdx.x == 0 && lane < nb) out[k0 + lane] = wv;
  }
  __syncthreads();
  if (FWD) {
    const int i = k0 + nb + blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double* row = A + (size_t)i * n + k0;
    double t = 0.0;
    for (int c = 0; c < nb; ++c) t += row[c] * xs[c];
    in[i] -= t;
  } else {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= k0) return;
    double t = 0.0;
    for (int c = 0; c < nb; ++c) t += A[(size_t)(k0 + c) * n + j] * xs[c];
    in[j] -= t;
  }
}

// A (n x n, symmetric, row major) is overwritten by its Cholesky factor (lower);
// x <- A^-1 b.  *flag is set when a pivot is not positive.  tmp: n doubles of scratch.
// diag: kPanel * (n rounded up to kPanel) doubles of scratch for the factored diagonal blocks.
inline void dense_cholesky_solve(double* A, int n, const double* b, double* x, double* tmp, double* diag, int* flag,
                                 hipStream_t st) {
  auto panel = [&](int k0) {  // factor columns [k0, k0 + kPanel) given all earlier updates
    const int nb = (n - k0 < kPanel) ? n - k0 : kPanel;
    const int rem = n - k0 - nb;
    hipLaunchKernelGGL(chol_panel_kernel, dim3(rem > 0 ? (rem + 255) / 256 : 1), dim3(256), 0, st, A, n, k0, diag, flag);
  };
  for (int k0 = 0; k0 < n; k0 += kTile) {
    panel(k0);
    const int k1 = k0 + kPanel;
    if (k1 >= n) break;
    {
      // the second panel of the pair sees the first: columns [k1, k1 + 32), rows >= k1
      const int nt = (n - k1 + kTile - 1) / kTile;
      hipLaunchKernelGGL(chol_syrk_kernel, dim3(nt, 1), dim3(256), 0, st, A, n, k0, kPanel, k1, k1 + kPanel);
    }
    panel(k1);
    const int k2 = k0 + kTile;
    if (k2 >= n) break;
    const int nt = (n - k2 + kTile - 1) / kTile;
    hipLaunchKernelGGL(chol_syrk_kernel, dim3(nt, nt), dim3(256), 0, st, A, n, k0, kTile, k2, n);
  }
  hipLaunchKernelGGL(chol_diag_store_kernel, dim3((n * kPanel + 255) / 256), dim3(256), 0, st, A, n, diag);
  // forward on x (working) -> tmp (solved y), backward on tmp (working) -> x
  hipMemcpyAsync(x, b, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
  for (int k0 = 0; k0 < n; k0 += kTile) {
    const int nb = (n - k0 < kTile) ? n - k0 : kTile;
    const int rem = n - k0 - nb;
    hipLaunchKernelGGL(chol_subst_kernel<true>, dim3(1 + (rem + 255) / 256), dim3(256), 0, st, A, n, k0, x, tmp);
  }
  const int last = ((n - 1) / kTile) * kTile;
  for (int k0 = last; k0 >= 0; k0 -= kTile)
    hipLaunchKernelGGL(chol_subst_kernel<false>, dim3(1 + (k0 + 255) / 256), dim3(256), 0, st, A, n, k0, tmp, x);
}

}  // namespace tmi
