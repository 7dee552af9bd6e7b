// Exact solve of the reduced camera system (DENSE_SCHUR / SPARSE_SCHUR / DENSE_QR; in the reference Ceres'
// DenseSchurComplementSolver / SparseSchurComplementSolver + CHOLMOD, selected below 1000 views by
// reconstruction_estimator_utils.cc:110-133) as ONE persistent launch: a tile-dataflow Cholesky.
//
// Round 2 factored 32 columns per launch: 161 dependent launches at n = 5130, each paying dispatch and drain, 14 ms
// for 45 GFLOP (4 % of the fp64 MFMA peak).  Here the matrix is cut into 64 x 64 tiles (lower triangle, tile-packed,
// plus ONE extra row that carries the right-hand side: the last row of the factor of [[S, b], [b^T, .]] is the
// forward-substituted y = L^-1 b, so forward substitution costs nothing extra), every workgroup owns a fixed list of
// tiles and computes each of them ONCE, left-looking:
//     C_IJ = S_IJ - sum_{k<J} L_Ik L_Jk^T            f64 MFMA (v_mfma_f64_16x16x4), C in registers, k ascending
//     I == J:  L_JJ = chol(C_JJ), X_J = L_JJ^-1      in LDS (two 32 x 32 in-register factorisations + small products)
//     I >  J:  L_IJ = C_IJ X_J^T                     MFMA again (the triangular solve as a product)
// A finished tile is published with write-through stores and a flag; consumers poll the flags (agent-scope relaxed
// atomics, the hand-over idiom of kernels.h: no fences, nothing but the flag crosses workgroups unannounced).  Tiles on
// and next to the diagonal -- the critical path -- belong to dedicated workgroups that pre-accumulate while the previous
// diagonal block is being factored; everything else is dealt round-robin in column-major order, which is a topological
// order of the dependencies, so the earliest unfinished tile can always run: no deadlock as long as the grid is
// co-resident (one workgroup per CU, grid <= #CUs).  Backward substitution is a second dataflow phase of the same
// launch (block column J: x_J = X_J^T (y_J - sum_{I>J} L_IJ^T x_I)).
// The summation order of every tile is fixed (k ascending), so results are bit-reproducible whatever the timing.
// Every wait is bounded (wall clock): a grid that cannot become co-resident (two such kernels racing for the CUs of one
// device from different processes) aborts with a flag instead of hanging, and the caller falls back to the
// launch-per-panel path of dense_cholesky.h.
#pragma once
#include <hip/hip_runtime.h>

#include "device_view.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the flag hand-over below (write-through stores + s_waitcnt, no fences) is validated for gfx950 only"
#endif

namespace tmi {
namespace cdf {

constexpr int TS = 64;          // tile edge
constexpr int TP = 66;          // LDS pitch (doubles): 16 rows x 2 k-lanes of an MFMA operand read hit 32 distinct banks
constexpr int TILE = TS * TS;   // doubles per tile
constexpr int kBand = 2;        // tiles with I - J < kBand have dedicated owners
constexpr int kTeam = 16;       // workgroups per band
constexpr long long kSpinLimitTicks = 300000000ll;  // 3 s of the 100 MHz wall clock

struct Args {
  double* tiles;   // [T (T + 1) / 2][64][64] tile (I, J) at I (I + 1) / 2 + J, row major
  double* linv;    // [T][64][64] inverse of the diagonal factors
  int* tflag;      // [T (T + 1) / 2] == epoch once tile (I, J) holds L_IJ
  int* dflag;      // [T] == epoch once linv[J] is valid
  int* xflag;      // [T] == epoch once x block J is valid
  double* x;       // [n] solution
  int* ctrl;       // [0] abort (a wait timed out)
  int* singular;   // set to 1 on a non-positive pivot
  int n, T, epoch, band, team, G;
};

__host__ __device__ inline size_t tile_index(int I, int J) { return (size_t)I * (I + 1) / 2 + J; }

__device__ __forceinline__ void st_wt(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_wt(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ld_flag(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_flag(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ double bcast_lane(double v, int src_lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), src_lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src_lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// One thread waits until *f == epoch; false when the launch was aborted or the wait timed out.
__device__ __forceinline__ bool spin_until(const int* f, int epoch, int* ctrl) {
  if (ld_flag(f) == epoch) return true;
  const long long t0 = wall_clock64();
  for (int it = 0;; ++it) {
    __builtin_amdgcn_s_sleep(2);
    if (ld_flag(f) == epoch) return true;
    if ((it & 255) == 255) {
      if (ld_flag(ctrl) != 0) return false;
      if (wall_clock64() - t0 > kSpinLimitTicks) {
        st_flag(ctrl, 1);
        return false;
      }
    }
  }
}

// Wave 0 only (all 64 lanes): how many of the steps k = from, from + 1, ... < J have both operands published?
// Waits until at least one has; returns `from` on abort.
__device__ __forceinline__ int ready_prefix(const Args& a, int I, int J, int from) {
  const int lane = threadIdx.x & 63;
  const int* fi = a.tflag + tile_index(I, 0);
  const int* fj = a.tflag + tile_index(J, 0);
  const long long t0 = wall_clock64();
  for (int it = 0;; ++it) {
    const int k = from + lane;
    const bool ok = k < J && ld_flag(fi + k) == a.epoch && (I == J || ld_flag(fj + k) == a.epoch);
    const unsigned long long m = __ballot(ok);
    const int cnt = m == ~0ull ? 64 : __builtin_ctzll(~m);
    if (cnt > 0) return from + cnt;
    __builtin_amdgcn_s_sleep(2);
    if ((it & 255) == 255) {
      if (ld_flag(a.ctrl) != 0) return from;
      if (wall_clock64() - t0 > kSpinLimitTicks) {
        if (lane == 0) st_flag(a.ctrl, 1);
        return from;
      }
    }
  }
}

// stage timestamps for tools/chol_harness.hip (nothing in the product build)
#ifdef TMI_CDF_STAMPS
__device__ long long* g_cdf_stamps = nullptr;
#define CDF_STAMP(i)                                                              \
  do {                                                                            \
    if (threadIdx.x == 0 && blockIdx.x == 0 && g_cdf_stamps) g_cdf_stamps[i] = wall_clock64(); \
  } while (0)
#else
#define CDF_STAMP(i) \
  do {               \
  } while (0)
#endif

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// acc[t] (+)= sum_k sgn * Pi[16 w + .][k] * Pj[16 t + .][k] over the 64 k of the staged tiles (f64 MFMA 16x16x4;
// C layout: column = lane & 15, row = (lane >> 4) + 4 q).  Wave w owns rows [16 w, 16 w + 16) of the tile.
template <bool NEG>
__device__ __forceinline__ void tile_mma(const double (*Pi)[TP], const double (*Pj)[TP], v4d (&acc)[4]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int s = 0; s < TS / 4; ++s) {
    double av = Pi[16 * w + li][4 * s + kk];
    if (NEG) av = -av;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const double bv = Pj[16 * t + li][4 * s + kk];
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
    }
  }
}

// 1 / sqrt(d) to full precision: v_rsq_f64 and two Newton steps (the library sqrt + division are ~45 dependent
// instructions per pivot, which WAS the factorisation's critical path)
__device__ __forceinline__ double rsqrt_nr(double d) {
  double y = __builtin_amdgcn_rsq(d);
  double e = fma(-d * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  e = fma(-d * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  return y;
}

// A zero the compiler cannot see through, in a VGPR: added to a wave-uniform LDS address it keeps the loaded value in
// vector registers.  (Left alone, the compiler turns every uniform load / v_readlane of these fully unrolled 32 x 32
// routines into scalar registers, runs out of them and spills SGPRs through v_writelane: 10 us per block.)
__device__ __forceinline__ int opaque_zero() {
  int z = 0;
  asm volatile("" : "+v"(z));
  return z;
}

// 32 x 32 Cholesky of the block at (b, b) of the LDS tile P by ONE wavefront: lane i holds row b + i in registers.
// Column j, once scaled, goes to the LDS staging vector `cb` and comes back as broadcast reads (two multipliers per
// read) for the rank-1 update.  Column j + 1 is updated first and its pivot's reciprocal square root started BEFORE
// the rest of the update (that one multiplier and the pivot travel by v_readlane), so the pivot chain runs under the
// update's FMAs.  Columns >= npiv (padding, the right-hand-side row) get a unit pivot.  dinv[b + j] = 1 / L[j][j].
__device__ __forceinline__ void potrf32(double (*P)[TP], double* dinv, double* cb, int b, int npiv, int* singular) {
  const int lane = threadIdx.x & 63;
  const int vz = opaque_zero();
  double r[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) r[c] = (lane < 32 && c <= lane) ? P[b + lane][b + c] : 0.0;
  bool bad = false;
  double d = bcast_lane(r[0], 0);
  if (0 >= npiv) {
    d = 1.0;
  } else if (!(d > 0.0) || !isfinite(d)) {
    bad = true;
    d = 1.0;
  }
  double di = rsqrt_nr(d);
  double my_di = (lane == 0) ? di : 0.0;  // lane j keeps 1 / L[j][j]  (no lane-conditional store inside the loop: any
                                          // control flow there makes the compiler sink every update to its use)
  // software pipeline: the multipliers of column j + 1 are staged and their broadcast reads issued BEFORE the rank-1
  // update with column j runs, so an LDS round trip per column hides under the FMAs (cb is double buffered: 2 x 64)
  v2d l[2][16];
  r[0] *= di;
  cb[lane] = r[0];
  {
    const v2d* cbv = reinterpret_cast<const v2d*>(cb + vz);
#pragma unroll
    for (int q = 0; q < 16; ++q) l[0][q] = cbv[q];
  }
#pragma unroll
  for (int j = 0; j < 31; ++j) {
    const int cur = j & 1, nxt = cur ^ 1;
    const double cj = r[j];  // lane j: sqrt(d); lanes below: the multipliers; lanes above: 0
    // column j + 1: its update, its pivot, its scaling
    r[j + 1] -= cj * l[cur][(j + 1) >> 1][(j + 1) & 1];
    double dn = bcast_lane(r[j + 1], j + 1);
    if (j + 1 >= npiv) {
      dn = 1.0;
    } else if (!(dn > 0.0) || !isfinite(dn)) {
      bad = true;
      dn = 1.0;
    }
    di = rsqrt_nr(dn);
    my_di = (lane == j + 1) ? di : my_di;
    r[j + 1] *= di;
    if (j + 2 < 32) {
      cb[64 * nxt + lane] = r[j + 1];
      const v2d* cbv = reinterpret_cast<const v2d*>(cb + 64 * nxt + vz);
#pragma unroll
      for (int q = (j + 2) >> 1; q < 16; ++q) l[nxt][q] = cbv[q];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = j + 2; m < 32; ++m) r[m] -= cj * l[cur][m >> 1][m & 1];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (bad && lane == 0) *singular = 1;
  if (lane < 32) {
    dinv[b + lane] = my_di;
#pragma unroll
    for (int c = 0; c < 32; ++c) P[b + lane][b + c] = (c <= lane) ? r[c] : 0.0;
  }
}

// One wavefront, 64 right-hand sides against the 32 x 32 lower-triangular L at (b, b) of P (dinv = reciprocal diagonal):
//   lanes 0..31  (j):  column j of X = L^-1           (L x = e_j)           -> X[b + .][b + j]
//   lanes 32..63 (r):  row r of the block below, A10[r][:] L^-T  (x L^T = a) -> P[b + 32 + r][b + .], when `below`
// Both are the same recurrence  x[m] = t[m] / L[m][m];  t[i] -= L[i][m] x[m] (i > m)  with the SAME coefficients, read
// as LDS broadcasts two columns at a time; the dependent chain is two multiplies and an FMA per column pair.
__device__ __forceinline__ void trsolve32(double (*P)[TP], double (*X)[TP], const double* dinv, int b, bool below) {
  const int lane = threadIdx.x & 63;
  const bool lo = lane < 32;
  if (!lo && !below) return;
  const int vz = opaque_zero();
  double t[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) t[c] = lo ? (c == lane ? 1.0 : 0.0) : P[b + lane][b + c];  // b + 32 + (lane - 32)
  const double* Lb = &P[b][b] + vz;
  const double* dv = dinv + b + vz;
  // the coefficients of column pair m + 2 are fetched while pair m is applied (they are static)
  v2d l[2][32];
  v2d dd[2];
  double sub[2];
  dd[0] = *reinterpret_cast<const v2d*>(dv);
  sub[0] = Lb[TP];
#pragma unroll
  for (int i = 2; i < 32; ++i) l[0][i] = *reinterpret_cast<const v2d*>(Lb + i * TP);
#pragma unroll
  for (int m = 0; m < 32; m += 2) {
    const int cur = (m >> 1) & 1, nxt = cur ^ 1;
    if (m + 2 < 32) {
      dd[nxt] = *reinterpret_cast<const v2d*>(dv + m + 2);
      sub[nxt] = Lb[(m + 3) * TP + m + 2];
#pragma unroll
      for (int i = m + 4; i < 32; ++i) l[nxt][i] = *reinterpret_cast<const v2d*>(Lb + i * TP + m + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    const double x0 = t[m] * dd[cur][0];
    t[m] = x0;
    t[m + 1] -= sub[cur] * x0;
    const double x1 = t[m + 1] * dd[cur][1];
    t[m + 1] = x1;
#pragma unroll
    for (int i = m + 2; i < 32; ++i) {
      t[i] -= l[cur][i][0] * x0;
      t[i] -= l[cur][i][1] * x1;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (lo) {
#pragma unroll
    for (int i = 0; i < 32; ++i) X[b + i][b + lane] = t[i];
  } else {
#pragma unroll
    for (int c = 0; c < 32; ++c) P[b + lane][b + c] = t[c];
  }
}

// In LDS: P (lower triangle valid) -> its Cholesky factor L (in P, upper part zeroed) and X = L^-1 (full tile, upper part
// zero).  npiv = real pivot columns of this tile.  256 threads, every barrier is workgroup-uniform.
__device__ __forceinline__ void potrf64_inv(double (*P)[TP], double (*X)[TP], double* dinv, double* cb, int npiv, int* singular) {
  const int tid = threadIdx.x, w = tid >> 6;
  // --- block (0, 0), then X00 and L10 = A10 L00^-T in one sweep
  CDF_STAMP(0);
  if (w == 0) potrf32(P, dinv, cb, 0, npiv, singular);
  __syncthreads();
  CDF_STAMP(1);
  if (w == 0) trsolve32(P, X, dinv, 0, true);
  __syncthreads();
  CDF_STAMP(2);
  // A11 -= L10 L10^T
  {
    const int r = tid >> 3, c0 = (tid & 7) * 4;
    double o[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
    for (int m = 0; m < 32; ++m) {
      const double av = P[32 + r][m];
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] += av * P[32 + c0 + u][m];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) P[32 + r][32 + c0 + u] -= o[u];
  }
  __syncthreads();
  CDF_STAMP(4);
  // --- block (1, 1)
  if (w == 0) potrf32(P, dinv, cb, 32, npiv - 32, singular);
  // meanwhile M = L10 X00 into the (unused) upper-right quadrant of X: M[r][c] = sum_{m >= c} L10[r][m] X00[m][c]
  if (w != 0) {
    for (int e = tid - 64; e < 256; e += 192) {
      const int r = e >> 3, c0 = (e & 7) * 4;
      double o[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
      for (int m = 0; m < 32; ++m) {
        const double av = P[32 + r][m];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] += av * X[m][c0 + u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) X[r][32 + c0 + u] = o[u];
    }
  }
  __syncthreads();
  CDF_STAMP(5);
  if (w == 0) trsolve32(P, X, dinv, 32, false);
  __syncthreads();
  CDF_STAMP(6);
  // X10 = -X11 M : X10[r][c] = -sum_{m <= r} X11[r][m] M[m][c]
  {
    const int r = tid >> 3, c0 = (tid & 7) * 4;
    double o[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
    for (int m = 0; m < 32; ++m) {
      const double xv = X[32 + r][32 + m];  // zero for m > r
#pragma unroll
      for (int u = 0; u < 4; ++u) o[u] -= xv * X[m][32 + c0 + u];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      X[32 + r][c0 + u] = o[u];
      X[r][32 + c0 + u] = 0.0;
      P[r][32 + c0 + u] = 0.0;
    }
  }
  __syncthreads();
  CDF_STAMP(7);
}

// registers (MFMA C layout) <-> LDS tile
__device__ __forceinline__ void acc_to_lds(const v4d (&acc)[4], double (*P)[TP]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) P[16 * w + kk + 4 * q][16 * t + li] = acc[t][q];
}

// a whole tile global -> registers: thread t takes the 16-byte pieces t, t + 256, ... (fully coalesced)
__device__ __forceinline__ void tile_fetch(const double* __restrict__ g, v2d (&r)[8]) {
  const v2d* p = reinterpret_cast<const v2d*>(g) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 8; ++u) r[u] = p[u * 256];
}
__device__ __forceinline__ void tile_stage(const v2d (&r)[8], double (*P)[TP]) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int e = 2 * (u * 256 + (int)threadIdx.x);
    const int row = e >> 6, col = e & 63;
    P[row][col] = r[u][0];
    P[row][col + 1] = r[u][1];
  }
}
// LDS tile -> global, write-through (readers in other XCDs go to memory, never to a stale L2 line)
__device__ __forceinline__ void tile_publish(const double (*P)[TP], double* __restrict__ g) {
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int e = u * 256 + (int)threadIdx.x;
    st_wt(g + e, P[e >> 6][e & 63]);
  }
}

// tile (I, J): everything described at the top of the file.  Returns false on abort (workgroup-uniform).
__device__ __forceinline__ bool process_tile(const Args& a, int I, int J, double (*Pi)[TP], double (*Pj)[TP], double* dinv, double* cb, int* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 15, kk = lane >> 4;
  v4d acc[4];
  {
    const double* c = a.tiles + tile_index(I, J) * TILE;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[t][q] = c[(16 * w + kk + 4 * q) * TS + 16 * t + li];
  }
  int ready = 0;  // steps [0, ready) are known to be published (valid in wave 0, broadcast through sh[0])
  v2d ri[8], rj[8];
  if (J > 0) {
    if (w == 0) {
      ready = ready_prefix(a, I, J, 0);
      if (lane == 0) sh[0] = ready;
    }
    __syncthreads();
    ready = sh[0];
    if (ready == 0) return false;
    tile_fetch(a.tiles + tile_index(I, 0) * TILE, ri);
    if (I != J) tile_fetch(a.tiles + tile_index(J, 0) * TILE, rj);
  }
  for (int k = 0; k < J; ++k) {
    tile_stage(ri, Pi);
    if (I != J) tile_stage(rj, Pj);
    if (k + 1 < J && k + 1 >= ready && w == 0) {
      const int r = ready_prefix(a, I, J, k + 1);
      if (lane == 0) sh[0] = r;
    }
    __syncthreads();
    if (k + 1 < J) {
      if (k + 1 >= ready) {
        ready = sh[0];
        if (ready <= k + 1) return false;
      }
      tile_fetch(a.tiles + tile_index(I, k + 1) * TILE, ri);
      if (I != J) tile_fetch(a.tiles + tile_index(J, k + 1) * TILE, rj);
    }
    tile_mma<true>(Pi, I == J ? Pi : Pj, acc);
    __syncthreads();
  }
  if (I == J) {
    acc_to_lds(acc, Pi);
    __syncthreads();
    const int npiv = min(TS, a.n - TS * J);
    potrf64_inv(Pi, Pj, dinv, cb, npiv, a.singular);
    tile_publish(Pj, a.linv + (size_t)J * TILE);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) st_flag(a.dflag + J, a.epoch);
    CDF_STAMP(8);
    // the factor itself: only the substitution reads it (the right-hand-side row of the last tile row)
    tile_publish(Pi, a.tiles + tile_index(J, J) * TILE);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) st_flag(a.tflag + tile_index(J, J), a.epoch);
    CDF_STAMP(9);
  } else {
    if (threadIdx.x == 0) sh[0] = spin_until(a.dflag + J, a.epoch, a.ctrl) ? 1 : 0;
    acc_to_lds(acc, Pi);
    __syncthreads();
    if (!sh[0]) return false;
    tile_fetch(a.linv + (size_t)J * TILE, rj);
    tile_stage(rj, Pj);
    __syncthreads();
    v4d out[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) out[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    tile_mma<false>(Pi, Pj, out);  // L_IJ[r][c] = sum_m C[r][m] X[c][m]
    __syncthreads();
    acc_to_lds(out, Pi);
    __syncthreads();
    tile_publish(Pi, a.tiles + tile_index(I, J) * TILE);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) st_flag(a.tflag + tile_index(I, J), a.epoch);
  }
  return true;
}

// backward substitution of block column J (see the top of the file); false on abort
__device__ __forceinline__ bool substitute_block(const Args& a, int J, double (*Pi)[TP], double* xs, double* part, int* sh) {
  const int tid = threadIdx.x;
  const int Treal = (a.n + TS - 1) / TS;
  const int rl = a.n - TS * (a.T - 1);  // local row of the right-hand side in the last tile row
  const int c2 = tid & 31, rg = tid >> 5;
  double s0 = 0.0, s1 = 0.0;  // partial sums of columns 2 c2, 2 c2 + 1 over rows rg, rg + 8, ...
  // Everything that does not depend on x_I is fetched before the wait for it (the factor is final by now, the flags are
  // still checked): X_J goes to LDS up front, a tile's loads are in flight while the flag of x_I is polled.
  if (tid == 0) sh[0] = (spin_until(a.tflag + tile_index(a.T - 1, J), a.epoch, a.ctrl) && spin_until(a.dflag + J, a.epoch, a.ctrl)) ? 1 : 0;
  __syncthreads();
  if (!sh[0]) return false;
  {
    v2d r[8];
    tile_fetch(a.linv + (size_t)J * TILE, r);
    tile_stage(r, Pi);
  }
  for (int I = Treal - 1; I > J; --I) {
    __syncthreads();  // sh[0] / xs of the previous step are free
    if (tid == 0) sh[0] = spin_until(a.tflag + tile_index(I, J), a.epoch, a.ctrl) ? 1 : 0;
    __syncthreads();
    if (!sh[0]) return false;
    const v2d* p = reinterpret_cast<const v2d*>(a.tiles + tile_index(I, J) * TILE);
    v2d l[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) l[u] = p[(rg + 8 * u) * 32 + c2];
    if (tid == 0) sh[1] = spin_until(a.xflag + I, a.epoch, a.ctrl) ? 1 : 0;
    __syncthreads();
    if (!sh[1]) return false;
    if (tid < TS) xs[tid] = (TS * I + tid < a.n) ? ld_wt(a.x + TS * I + tid) : 0.0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int row = rg + 8 * u;
      s0 += l[u][0] * xs[row];
      s1 += l[u][1] * xs[row];
    }
  }
  __syncthreads();
  // y_J: the right-hand-side row of the factor
  part[rg * TS + 2 * c2] = s0;
  part[rg * TS + 2 * c2 + 1] = s1;
  __syncthreads();
  if (tid < TS) {
    double wv = (TS * J + tid < a.n) ? ld_wt(a.tiles + tile_index(a.T - 1, J) * TILE + (size_t)rl * TS + tid) : 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) wv -= part[g * TS + tid];
    xs[tid] = wv;
  }
  __syncthreads();
  // x_J[c] = sum_{r >= c} X[r][c] w[r]; 4 threads per column
  {
    const int c = tid >> 2, q = tid & 3;
    double t = 0.0;
    for (int r = q; r < TS; r += 4) t += Pi[r][c] * xs[r];
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    if (q == 0 && TS * J + c < a.n) st_wt(a.x + TS * J + c, t);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (tid == 0) st_flag(a.xflag + J, a.epoch);
  CDF_STAMP(10);
  return true;
}

__global__ __launch_bounds__(256) void chol_dataflow_kernel(Args a) {
  __shared__ double Pi[TS][TP], Pj[TS][TP];
  __shared__ double xs[TS], part[8 * TS];
  __shared__ int sh[2];
  const int g = blockIdx.x;
  const int nteam = a.band * a.team;
  // ---- phase 1: the factor (+ the forward-substituted right-hand side as its last row)
  if (g < nteam) {
    const int b = g / a.team, m = g - b * a.team;
    for (int J = m; J + b < a.T; J += a.team)
      if (!process_tile(a, J + b, J, Pi, Pj, xs, part, sh)) return;
  } else {
    const int nb = a.G - nteam;
    int J = 0;
    long long base = 0;
    for (long long idx = g - nteam;; idx += nb) {
      while (J < a.T && idx >= base + max(0, a.T - J - a.band)) {
        base += max(0, a.T - J - a.band);
        ++J;
      }
      if (J >= a.T) break;
      if (!process_tile(a, J + a.band + (int)(idx - base), J, Pi, Pj, xs, part, sh)) return;
    }
  }
  // ---- phase 2: L^T x = y
  const int Treal = (a.n + TS - 1) / TS;
  for (int J = Treal - 1 - g; J >= 0; J -= a.G)
    if (!substitute_block(a, J, Pi, xs, part, sh)) return;
}

// scatter the symmetric block storage (upper blocks in `ub`, diagonal blocks in Sdiag) into the LOWER tiles, and the
// right-hand side into row n
template <int D>
__global__ __launch_bounds__(256) void tile_gather_kernel(DeviceView v, const double* __restrict__ ub,
                                                          const double* __restrict__ rhs, double* __restrict__ tiles,
                                                          int n) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n_off = (long long)v.nub * D * D;
  const long long n_diag = n_off + (long long)v.Nrb * D * D;
  int i, j;
  double val;
  if (e < n_off) {
    const int u = (int)(e / (D * D));
    const int w = (int)(e - (long long)u * (D * D));
    const int p = w / D, q = w - p * D;
    i = v.ub_j[u] * D + q;  // the mirrored (lower) element of the upper block (row block ub_i < column block ub_j)
    j = v.ub_i[u] * D + p;
    val = ub[e];
  } else if (e < n_diag) {
    const long long f = e - n_off;
    const int rb = (int)(f / (D * D));
    const int w = (int)(f - (long long)rb * (D * D));
    const int p = w / D, q = w - p * D;
    if (q > p) return;
    i = rb * D + p;
    j = rb * D + q;
    val = v.Sdiag[f];
  } else if (e < n_diag + n) {
    i = n;
    j = (int)(e - n_diag);
    val = rhs[j];
  } else {
    return;
  }
  tiles[tile_index(i >> 6, j >> 6) * TILE + (size_t)(i & 63) * TS + (j & 63)] = val;
}

struct Plan {
  int n = 0, T = 0, band = 0, team = 0, G = 0;
  size_t tile_doubles = 0, linv_doubles = 0, flag_ints = 0;
};

inline Plan make_plan(int n, int num_cus) {
  Plan p;
  p.n = n;
  p.T = n / TS + 1;  // row n (the right-hand side) must exist
  p.band = p.T < kBand ? p.T : kBand;
  p.team = p.T < kTeam ? p.T : kTeam;
  long long bulk = 0;
  for (int J = 0; J < p.T; ++J) bulk += (p.T - J - p.band > 0) ? p.T - J - p.band : 0;
  const int nteam = p.band * p.team;
  long long nb = num_cus - nteam;
  if (nb > bulk) nb = bulk;
  if (nb < 1 && bulk > 0) nb = 1;
  p.G = nteam + (int)nb;
  const size_t ntiles = tile_index(p.T - 1, p.T - 1) + 1;
  p.tile_doubles = ntiles * TILE;
  p.linv_doubles = (size_t)p.T * TILE;
  p.flag_ints = ntiles + 2 * (size_t)p.T + 8;
  return p;
}

}  // namespace cdf
}  // namespace tmi
