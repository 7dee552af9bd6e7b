// Exact solve of the reduced camera system for the DENSE_SCHUR / SPARSE_SCHUR /
// DENSE_QR solver types: the explicit Schur complement is scattered into a dense
// n x n matrix (n = #reduced blocks * D, a few thousand below the reference's
// 1000-view switch to ITERATIVE_SCHUR, reconstruction_estimator_utils.cc:110-133)
// and factored with a tiled right-looking Cholesky (32 x 32 tiles, lower
// triangle), followed by tiled forward / backward substitution.  In the
// reference this is Ceres' DenseSchurComplementSolver / CHOLMOD
// (SparseSchurComplementSolver); both solve the same normal equations exactly.
#pragma once
#include <hip/hip_runtime.h>

#include "device_view.h"

namespace tmi {

constexpr int kTile = 32;

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

__global__ __launch_bounds__(256) void chol_potrf_tile_kernel(double* __restrict__ A, int n, int k0,
                                                              int* __restrict__ flag) {
  __shared__ double T[kTile][kTile + 1];
  const int nb = min(kTile, n - k0);
  const int tid = threadIdx.x;
  for (int e = tid; e < kTile * kTile; e += 256) {
    const int r = e / kTile, c = e - r * kTile;
    T[r][c] = (r < nb && c < nb) ? A[(size_t)(k0 + r) * n + k0 + c] : (r == c ? 1.0 : 0.0);
  }
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    if (tid == 0) {
      double d = T[j][j];
      if (!(d > 0.0) || !isfinite(d)) {
        *flag = 1;
        d = 1.0;
      }
      T[j][j] = sqrt(d);
    }
    __syncthreads();
    if (tid > j && tid < nb) T[tid][j] /= T[j][j];
    __syncthreads();
    for (int e = tid; e < kTile * kTile; e += 256) {
      const int i = e / kTile, m = e - i * kTile;
      if (m > j && m <= i && i < nb) T[i][m] -= T[i][j] * T[m][j];
    }
    __syncthreads();
  }
  for (int e = tid; e < kTile * kTile; e += 256) {
    const int r = e / kTile, c = e - r * kTile;
    if (r < nb && c <= r) A[(size_t)(k0 + r) * n + k0 + c] = T[r][c];
  }
}

// rows i >= k0 + nb:  A[i, k0:k0+nb] <- A[i, k0:k0+nb] L_kk^-T
__global__ __launch_bounds__(256) void chol_trsm_kernel(double* __restrict__ A, int n, int k0) {
  __shared__ double L[kTile][kTile + 1];
  const int nb = min(kTile, n - k0);
  for (int e = threadIdx.x; e < kTile * kTile; e += 256) {
    const int r = e / kTile, c = e - r * kTile;
    L[r][c] = (r < nb && c <= r) ? A[(size_t)(k0 + r) * n + k0 + c] : 0.0;
  }
  __syncthreads();
  const int i = k0 + nb + blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double* row = A + (size_t)i * n + k0;
  double x[kTile];
#pragma unroll
  for (int c = 0; c < kTile; ++c) {
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
  for (int c = 0; c < kTile; ++c)
    if (c < nb) row[c] = x[c];
}

// trailing update, lower tiles only: A[I, J] -= P_I P_J^T with P = A[:, k0:k0+nb]
__global__ __launch_bounds__(256) void chol_syrk_kernel(double* __restrict__ A, int n, int k0) {
  if (blockIdx.y > blockIdx.x) return;
  __shared__ double Pi[kTile][kTile + 1], Pj[kTile][kTile + 1];
  const int nb = min(kTile, n - k0);
  const int base = k0 + nb;
  const int i0 = base + blockIdx.x * kTile, j0 = base + blockIdx.y * kTile;
  for (int e = threadIdx.x; e < kTile * kTile; e += 256) {
    const int r = e / kTile, c = e - r * kTile;
    Pi[r][c] = (i0 + r < n && c < nb) ? A[(size_t)(i0 + r) * n + k0 + c] : 0.0;
    Pj[r][c] = (j0 + r < n && c < nb) ? A[(size_t)(j0 + r) * n + k0 + c] : 0.0;
  }
  __syncthreads();
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  double c00 = 0.0, c01 = 0.0, c10 = 0.0, c11 = 0.0;
#pragma unroll
  for (int k = 0; k < kTile; ++k) {
    const double a0 = Pi[2 * ty][k], a1 = Pi[2 * ty + 1][k];
    const double b0 = Pj[2 * tx][k], b1 = Pj[2 * tx + 1][k];
    c00 += a0 * b0;
    c01 += a0 * b1;
    c10 += a1 * b0;
    c11 += a1 * b1;
  }
  const int i = i0 + 2 * ty, j = j0 + 2 * tx;
  if (i < n && j < n) A[(size_t)i * n + j] -= c00;
  if (i < n && j + 1 < n) A[(size_t)i * n + j + 1] -= c01;
  if (i + 1 < n && j < n) A[(size_t)(i + 1) * n + j] -= c10;
  if (i + 1 < n && j + 1 < n) A[(size_t)(i + 1) * n + j + 1] -= c11;
}

// forward: solve L_kk y_k = w_k in place (one wave)
__global__ __launch_bounds__(64) void chol_fwd_diag_kernel(const double* __restrict__ A, int n, int k0,
                                                           double* __restrict__ w) {
  __shared__ double y[kTile];
  const int nb = min(kTile, n - k0);
  if (threadIdx.x < kTile) y[threadIdx.x] = (threadIdx.x < nb) ? w[k0 + threadIdx.x] : 0.0;
  __syncthreads();
  for (int j = 0; j < nb; ++j) {
    if (threadIdx.x == 0) y[j] /= A[(size_t)(k0 + j) * n + k0 + j];
    __syncthreads();
    const int i = threadIdx.x;
    if (i > j && i < nb) y[i] -= A[(size_t)(k0 + i) * n + k0 + j] * y[j];
    __syncthreads();
  }
  if (threadIdx.x < nb) w[k0 + threadIdx.x] = y[threadIdx.x];
}
// rows i >= k0 + nb: w[i] -= L[i, k0:k0+nb] . w[k0:k0+nb]
__global__ __launch_bounds__(256) void chol_fwd_update_kernel(const double* __restrict__ A, int n, int k0,
                                                              double* __restrict__ w) {
  __shared__ double y[kTile];
  const int nb = min(kTile, n - k0);
  if (threadIdx.x < kTile) y[threadIdx.x] = (threadIdx.x < nb) ? w[k0 + threadIdx.x] : 0.0;
  __syncthreads();
  const int i = k0 + nb + blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double* row = A + (size_t)i * n + k0;
  double t = 0.0;
  for (int c = 0; c < nb; ++c) t += row[c] * y[c];
  w[i] -= t;
}
// backward: solve L_kk^T x_k = w_k in place
__global__ __launch_bounds__(64) void chol_bwd_diag_kernel(const double* __restrict__ A, int n, int k0,
                                                           double* __restrict__ w) {
  __shared__ double y[kTile];
  const int nb = min(kTile, n - k0);
  if (threadIdx.x < kTile) y[threadIdx.x] = (threadIdx.x < nb) ? w[k0 + threadIdx.x] : 0.0;
  __syncthreads();
  for (int j = nb - 1; j >= 0; --j) {
    if (threadIdx.x == 0) y[j] /= A[(size_t)(k0 + j) * n + k0 + j];
    __syncthreads();
    const int i = threadIdx.x;
    if (i < j) y[i] -= A[(size_t)(k0 + j) * n + k0 + i] * y[j];
    __syncthreads();
  }
  if (threadIdx.x < nb) w[k0 + threadIdx.x] = y[threadIdx.x];
}
// columns j < k0: w[j] -= sum_c L[k0 + c, j] x[k0 + c]
__global__ __launch_bounds__(256) void chol_bwd_update_kernel(const double* __restrict__ A, int n, int k0,
                                                              double* __restrict__ w) {
  __shared__ double y[kTile];
  const int nb = min(kTile, n - k0);
  if (threadIdx.x < kTile) y[threadIdx.x] = (threadIdx.x < nb) ? w[k0 + threadIdx.x] : 0.0;
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= k0) return;
  double t = 0.0;
  for (int c = 0; c < nb; ++c) t += A[(size_t)(k0 + c) * n + j] * y[c];
  w[j] -= t;
}

// A (n x n, symmetric, row major) is overwritten by its Cholesky factor (lower);
// x <- A^-1 b.  *flag is set when a pivot is not positive.
inline void dense_cholesky_solve(double* A, int n, const double* b, double* x, int* flag,
                                 hipStream_t st) {
  for (int k0 = 0; k0 < n; k0 += kTile) {
    const int nb = (n - k0 < kTile) ? n - k0 : kTile;
    hipLaunchKernelGGL(chol_potrf_tile_kernel, dim3(1), dim3(256), 0, st, A, n, k0, flag);
    const int rem = n - k0 - nb;
    if (rem > 0) {
      hipLaunchKernelGGL(chol_trsm_kernel, dim3((rem + 255) / 256), dim3(256), 0, st, A, n, k0);
      const int nt = (rem + kTile - 1) / kTile;
      hipLaunchKernelGGL(chol_syrk_kernel, dim3(nt, nt), dim3(256), 0, st, A, n, k0);
    }
  }
  hipMemcpyAsync(x, b, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st);
  for (int k0 = 0; k0 < n; k0 += kTile) {
    const int nb = (n - k0 < kTile) ? n - k0 : kTile;
    hipLaunchKernelGGL(chol_fwd_diag_kernel, dim3(1), dim3(64), 0, st, A, n, k0, x);
    const int rem = n - k0 - nb;
    if (rem > 0)
      hipLaunchKernelGGL(chol_fwd_update_kernel, dim3((rem + 255) / 256), dim3(256), 0, st, A, n, k0, x);
  }
  const int last = ((n - 1) / kTile) * kTile;
  for (int k0 = last; k0 >= 0; k0 -= kTile) {
    hipLaunchKernelGGL(chol_bwd_diag_kernel, dim3(1), dim3(64), 0, st, A, n, k0, x);
    if (k0 > 0)
      hipLaunchKernelGGL(chol_bwd_update_kernel, dim3((k0 + 255) / 256), dim3(256), 0, st, A, n, k0, x);
  }
}

}  // namespace tmi
