// CLUSTER_JACOBI for problems with shared intrinsics blocks (ceres::CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL,
// reference options bundle_adjustment.h:86-89).  Ceres clusters the cameras by visibility and inverts the block
// diagonal of the reduced camera matrix S over the clusters exactly.  Here a cluster is a shared intrinsics block g
// together with the views that share it: the principal submatrix of S over {views of g, g} (6 n_g + <= 10 unknowns,
// 56 for a group of 8 views, 1208 for a group of 200), factored densely every LM iteration and applied with two
// triangular solves per PCG iteration.  Views of private groups keep their own block (SCHUR_JACOBI).
//
// Why: sharing intrinsics adds about three near-degenerate directions per block to the block-Jacobi preconditioned
// system -- a principal-point shift compensated by a coherent rotation of every view of the block, and the focal /
// distortion analogues -- and PCG pays 45-80 iterations per LM iteration for them (profiles/r03_c).  They live inside
// a cluster, view-view coupling included, so the exact cluster inverse removes them: 4-5 PCG iterations.
//
// Device side: the clusters' lower triangles are gathered from the block storage of S into 64 x 64 tiles and factored
// by the tile-dataflow Cholesky of dense_cholesky_df.h, all clusters in ONE launch (every workgroup serves a fixed
// range of clusters / tiles in a global topological order, so any grid size is deadlock free); the application is a
// second dataflow kernel, forward then backward substitution over the tile rows of every cluster.
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "dense_cholesky_df.h"
#include "device_view.h"

namespace tmi {
namespace clp {

using cdf::TILE;
using cdf::TP;
using cdf::TS;

struct ClusterDesc {
  long long tile0;   // first tile of the cluster's packed lower triangle (units of tiles)
  long long linv0;   // first inverse-diagonal tile
  int tflag0;        // flags of the tiles; dflag0: of the diagonal inverses; yflag0 / xflag0: of the solve's rows
  int dflag0, yflag0, xflag0;
  int n, T;          // unknowns (padding compressed away), tile rows
  int wg0, wgn;      // workgroups [wg0, wg0 + wgn) of the launches serve this cluster
  int idx0;          // first entry of the cluster's local -> global index map
  int vec0;          // first entry of the cluster's scratch vectors (64 T doubles each)
};

// one block of S to copy into a cluster's lower triangle
struct GatherEntry {
  int cluster;
  int i0, j0;        // local offsets (row block a >= column block b)
  int ni, nj;        // true dimensions
  int src;           // >= 0: upper block of S (stored as its transpose); < 0: diagonal block -1 - rb
  int offdiag;       // CLUSTER_TRIDIAGONAL: the block couples two NEIGHBOURING clusters of a chain (scaled by off_scale)
  int tr;            // the upper block is S(row part, column part) itself (the row member is the LOWER reduced block:
                     // only in the chains of CLUSTER_TRIDIAGONAL, whose members are not ascending)
};

template <int D>
__global__ __launch_bounds__(256) void cluster_gather_kernel(const GatherEntry* __restrict__ ge, int n_entries,
                                                             const ClusterDesc* __restrict__ desc,
                                                             const double* __restrict__ ub,
                                                             const double* __restrict__ Sdiag,
                                                             double* __restrict__ tiles, double off_scale) {
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= n_entries) return;
  const GatherEntry g = ge[e];
  const ClusterDesc c = desc[g.cluster];
  double* base = tiles + (size_t)c.tile0 * TILE;
  for (int w = threadIdx.x & 63; w < g.ni * g.nj; w += 64) {
    const int i = w / g.nj, j = w - i * g.nj;
    double val;
    if (g.src >= 0) {
      val = g.tr ? ub[(size_t)g.src * D * D + i * D + j]
                 : ub[(size_t)g.src * D * D + j * D + i];  // the upper block holds S(column part, row part)
      if (g.offdiag) val *= off_scale;
    } else {
      if (j > i) continue;
      val = Sdiag[(size_t)(-1 - g.src) * D * D + i * D + j];
    }
    const int gi = g.i0 + i, gj = g.j0 + j;
    base[cdf::tile_index(gi >> 6, gj >> 6) * TILE + (size_t)(gi & 63) * TS + (gj & 63)] = val;
  }
}

// Factor every cluster: workgroup w serves the clusters whose range holds it, in cluster order, taking the tiles
// w - wg0, w - wg0 + wgn, ... of the cluster's column-major tile order (dense_cholesky_df.h, process_tile).
__global__ __launch_bounds__(256) void cluster_factor_kernel(const ClusterDesc* __restrict__ desc, int ncl,
                                                             double* tiles, double* linv, int* flags, int epoch,
                                                             int* ctrl, int* cl_bad) {
  __shared__ double Pi[TS][TP], Pj[TS][TP];
  __shared__ double xs[TS], part[8 * TS];
  __shared__ int sh[2];
  const int w = blockIdx.x;
  for (int c = 0; c < ncl; ++c) {
    const ClusterDesc d = desc[c];
    if (w < d.wg0 || w >= d.wg0 + d.wgn) continue;
    cdf::Args a;
    a.tiles = tiles + (size_t)d.tile0 * TILE;
    a.linv = linv + (size_t)d.linv0 * TILE;
    a.tflag = flags + d.tflag0;
    a.dflag = flags + d.dflag0;
    a.xflag = nullptr;
    a.x = nullptr;
    a.ctrl = ctrl;
    a.singular = cl_bad + c;
    a.n = d.n;
    a.T = d.T;
    a.epoch = epoch;
    a.band = 0;
    a.team = 0;
    a.G = d.wgn;
    int J = 0;
    long long base = 0;
    for (long long idx = w - d.wg0;; idx += d.wgn) {
      while (J < d.T && idx >= base + (d.T - J)) {
        base += d.T - J;
        ++J;
      }
      if (J >= d.T) break;
      if (!cdf::process_tile(a, J + (int)(idx - base), J, Pi, Pj, xs, part, sh)) return;
    }
  }
}

// z = C^-1 r for every cluster C = L L^T: forward substitution down the tile rows (y), backward up (x), rows dealt
// to the cluster's workgroups round-robin; r and z are addressed through the cluster's index map, y and x live in
// the cluster's scratch vectors.  A cluster whose factorisation met a non-positive pivot is skipped (its entries of z
// keep the block-Jacobi values they arrive with).
__global__ __launch_bounds__(256) void cluster_apply_kernel(const ClusterDesc* __restrict__ desc, int ncl,
                                                            const double* __restrict__ tiles,
                                                            const double* __restrict__ linv, int* flags, int epoch,
                                                            int* ctrl, const int* __restrict__ cl_bad,
                                                            const int* __restrict__ idx, double* vec,
                                                            const double* __restrict__ r, double* __restrict__ z) {
  __shared__ double X[TS][TP];
  __shared__ double ws[TS], part[8 * TS];
  __shared__ int sh[2];
  const int tid = threadIdx.x, w = blockIdx.x;
  for (int c = 0; c < ncl; ++c) {
    const ClusterDesc d = desc[c];
    if (w < d.wg0 || w >= d.wg0 + d.wgn || cl_bad[c]) continue;
    const double* ct = tiles + (size_t)d.tile0 * TILE;
    const double* cl = linv + (size_t)d.linv0 * TILE;
    int* yflag = flags + d.yflag0;
    int* xflag = flags + d.xflag0;
    double* y = vec + d.vec0;
    double* x = y + (size_t)TS * d.T;
    const int* map = idx + d.idx0;
    const int g = w - d.wg0;
    // ---- forward: y_J = X_J (r_J - sum_{K<J} L_JK y_K)
    for (int J = g; J < d.T; J += d.wgn) {
      const int row = tid >> 2, q = tid & 3;
      double acc = 0.0;
      {  // X_J into LDS BEFORE the wait for the last y_K: off the chain  y_{J-1} -> y_J
        cdf::v2d t[8];
        cdf::tile_fetch(cl + (size_t)J * TILE, t);
        cdf::tile_stage(t, X);
      }
      for (int K = 0; K < J; ++K) {
        // the factor's tile is static: fetched before the wait for y_K, so only the flag, 64 doubles and the FMAs
        // sit between y_K and this row's partial sum
        const cdf::v2d* p = reinterpret_cast<const cdf::v2d*>(ct + cdf::tile_index(J, K) * TILE + (size_t)row * TS + 16 * q);
        cdf::v2d l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) l[u] = p[u];
        if (tid == 0) sh[0] = cdf::spin_until(yflag + K, epoch, ctrl) ? 1 : 0;
        __syncthreads();
        if (!sh[0]) return;
        if (tid < TS) ws[tid] = cdf::ld_wt(y + TS * K + tid);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += l[u][0] * ws[16 * q + 2 * u] + l[u][1] * ws[16 * q + 2 * u + 1];
        __syncthreads();
      }
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      if (q == 0) ws[row] = ((TS * J + row < d.n) ? r[map[TS * J + row]] : 0.0) - acc;
      __syncthreads();
      {
        double s = 0.0;
        for (int cc = q; cc <= row; cc += 4) s += X[row][cc] * ws[cc];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (q == 0) cdf::st_wt(y + TS * J + row, s);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      if (tid == 0) cdf::st_flag(yflag + J, epoch);
    }
    // ---- backward: x_J = X_J^T (y_J - sum_{I>J} L_IJ^T x_I)
    int Jlast = g + ((d.T - 1 - g) / d.wgn) * d.wgn;  // the last row this workgroup owns
    if (g >= d.T) Jlast = -1;
    for (int J = Jlast; J >= 0; J -= d.wgn) {
      const int c2 = tid & 31, rg = tid >> 5;
      double s0 = 0.0, s1 = 0.0;
      __syncthreads();  // (the previous row's reads of X are done)
      {
        cdf::v2d t[8];
        cdf::tile_fetch(cl + (size_t)J * TILE, t);
        cdf::tile_stage(t, X);
      }
      for (int I = d.T - 1; I > J; --I) {
        const cdf::v2d* p = reinterpret_cast<const cdf::v2d*>(ct + cdf::tile_index(I, J) * TILE);
        cdf::v2d l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) l[u] = p[(rg + 8 * u) * 32 + c2];
        if (tid == 0) sh[0] = cdf::spin_until(xflag + I, epoch, ctrl) ? 1 : 0;
        __syncthreads();
        if (!sh[0]) return;
        if (tid < TS) ws[tid] = cdf::ld_wt(x + TS * I + tid);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int row = rg + 8 * u;
          s0 += l[u][0] * ws[row];
          s1 += l[u][1] * ws[row];
        }
        __syncthreads();
      }
      if (tid == 0) sh[0] = cdf::spin_until(yflag + J, epoch, ctrl) ? 1 : 0;
      part[rg * TS + 2 * c2] = s0;
      part[rg * TS + 2 * c2 + 1] = s1;
      __syncthreads();
      if (!sh[0]) return;
      if (tid < TS) {
        double wv = cdf::ld_wt(y + TS * J + tid);
#pragma unroll
        for (int k = 0; k < 8; ++k) wv -= part[k * TS + tid];
        ws[tid] = wv;
      }
      __syncthreads();
      {
        const int cc = tid >> 2, q = tid & 3;
        double s = 0.0;
        for (int rr = cc + q; rr < TS; rr += 4) s += X[rr][cc] * ws[rr];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (q == 0) {
          cdf::st_wt(x + TS * J + cc, s);
          if (TS * J + cc < d.n) z[map[TS * J + cc]] = s;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      if (tid == 0) cdf::st_flag(xflag + J, epoch);
    }
  }
}

// after the cluster solve changed z: the partial sums of r.z that pcg_b3 finishes (same layout as pcg_b2's: one entry
// per group of four reduced blocks)
template <int D>
__global__ __launch_bounds__(256) void cluster_rz_kernel(DeviceView v, int nblocks, double* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rb = blockIdx.x * 4 + wv;
  __shared__ double sw[4];
  double a = 0.0;
  if (rb < v.Nrb && lane < D) a = v.cg_r[rb * D + lane] * v.cg_z[rb * D + lane];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  if (lane == 0) sw[wv] = a;
  __syncthreads();
  if (threadIdx.x == 0) partial[nblocks + blockIdx.x] = sw[0] + sw[1] + sw[2] + sw[3];
}

// after the cluster solve changed z at the start of a PCG solve: p = z, rho = r.z (what pcg_init left for the block
// diagonal), one workgroup
__global__ __launch_bounds__(1024) void cluster_init_fix_kernel(DeviceView v, int n) {
  __shared__ double sh[1024];
  double l = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const double zz = v.cg_z[i];
    v.cg_p[i] = zz;
    l += v.cg_r[i] * zz;
  }
  sh[threadIdx.x] = l;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double rho = sh[0];
    v.scal[SC_RHO] = rho;
    v.flags[FL_PCG_FAIL] = (rho == 0.0 || !isfinite(rho)) ? 1 : 0;
  }
}

// ---- host side: the plan ------------------------------------------------------------------------------------
struct Plan {
  int ncl = 0;
  std::vector<ClusterDesc> desc;
  std::vector<GatherEntry> entries;
  std::vector<int> idx;
  long long n_tiles = 0, n_linv = 0;
  int n_flags = 0, n_vec = 0, grid = 0;
};

// members[c] = the reduced blocks of cluster c in ascending order (views, then the shared block); rb_dim = true
// dimensions; ub_lookup(bi, bj) = index of the upper block (bi < bj) or -1
// ordinal (CLUSTER_TRIDIAGONAL, cluster_chains.h): position of every member's cluster in its chain -- only the blocks
// inside a cluster and between neighbouring clusters are gathered, everything else of the chain's matrix stays zero
template <class Lookup>
inline Plan make_plan(const std::vector<std::vector<int> >& members, const std::vector<int>& rb_dim, int D,
                      Lookup ub_lookup, int num_cus, const std::vector<std::vector<int> >* ordinal = nullptr) {
  Plan p;
  p.ncl = (int)members.size();
  p.desc.resize(p.ncl);
  std::vector<double> cost(p.ncl);
  double total = 0.0;
  for (int c = 0; c < p.ncl; ++c) {
    ClusterDesc& d = p.desc[c];
    std::vector<int> off(members[c].size() + 1, 0);
    for (size_t m = 0; m < members[c].size(); ++m) off[m + 1] = off[m] + rb_dim[members[c][m]];
    d.n = off.back();
    d.T = (d.n + TS - 1) / TS;
    if (d.T < 1) d.T = 1;
    d.tile0 = p.n_tiles;
    p.n_tiles += (long long)d.T * (d.T + 1) / 2;
    d.linv0 = p.n_linv;
    p.n_linv += d.T;
    d.tflag0 = p.n_flags;
    p.n_flags += d.T * (d.T + 1) / 2;
    d.dflag0 = p.n_flags;
    p.n_flags += d.T;
    d.yflag0 = p.n_flags;
    p.n_flags += d.T;
    d.xflag0 = p.n_flags;
    p.n_flags += d.T;
    d.idx0 = (int)p.idx.size();
    for (size_t m = 0; m < members[c].size(); ++m)
      for (int k = 0; k < rb_dim[members[c][m]]; ++k) p.idx.push_back(members[c][m] * D + k);
    p.idx.resize(d.idx0 + (size_t)TS * d.T, 0);  // padding entries are never dereferenced (guarded by n)
    d.vec0 = p.n_vec;
    p.n_vec += 2 * TS * d.T;
    for (size_t a = 0; a < members[c].size(); ++a) {
      for (size_t b = 0; b <= a; ++b) {
        GatherEntry g;
        g.cluster = c;
        g.i0 = off[a];
        g.j0 = off[b];
        g.ni = rb_dim[members[c][a]];
        g.nj = rb_dim[members[c][b]];
        if (g.ni == 0 || g.nj == 0) continue;
        g.offdiag = 0;
        if (ordinal) {
          const int dord = (*ordinal)[c][a] - (*ordinal)[c][b];
          if (dord < -1 || dord > 1) continue;
          g.offdiag = dord != 0 ? 1 : 0;
        }
        g.tr = 0;
        if (a == b) {
          g.src = -1 - members[c][a];
        } else if (members[c][b] < members[c][a]) {
          g.src = ub_lookup(members[c][b], members[c][a]);
          if (g.src < 0) continue;
        } else {
          g.src = ub_lookup(members[c][a], members[c][b]);
          g.tr = 1;
          if (g.src < 0) continue;
        }
        p.entries.push_back(g);
      }
    }
    cost[c] = (double)d.T * d.T * d.T / 6.0 + d.T;
    total += cost[c];
  }
  // workgroups: big clusters get several, small ones share one; never more than the device can hold at once
  int budget = num_cus;
  for (int attempt = 0; attempt < 40; ++attempt) {
    const double avg = total / budget;
    int w = 0;
    double acc = 0.0;
    for (int c = 0; c < p.ncl; ++c) {
      ClusterDesc& d = p.desc[c];
      const int tiles = d.T * (d.T + 1) / 2;
      if (cost[c] >= avg) {
        if (acc > 0.0) {
          ++w;
          acc = 0.0;
        }
        int k = (int)(cost[c] / avg);
        if (k < 1) k = 1;
        if (k > tiles) k = tiles;
        d.wg0 = w;
        d.wgn = k;
        w += k;
      } else {
        d.wg0 = w;
        d.wgn = 1;
        acc += cost[c];
        if (acc >= avg) {
          ++w;
          acc = 0.0;
        }
      }
    }
    if (acc > 0.0) ++w;
    p.grid = w;
    if (w <= num_cus) break;
    budget = budget * 9 / 10;
    if (budget < 1) budget = 1;
  }
  return p;
}

}  // namespace clp
}  // namespace tmi
