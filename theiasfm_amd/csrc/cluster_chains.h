// CLUSTER_TRIDIAGONAL (ceres::CLUSTER_TRIDIAGONAL, reference options bundle_adjustment.h:86-89; the reference only hands
// the enum to Ceres, bundle_adjuster.cc:59-63): which cluster pairs the preconditioner keeps.  Host code, run once per
// handle at tmi_ba_solver_create.
//
// Ceres 1.14 (visibility_based_preconditioner.cc: ComputeClusterTridiagonalSparsity, CreateClusterGraph,
// ForestToClusterPairs; graph_algorithms.h: Degree2MaximumSpanningForest) keeps, besides the pairs (i, i) of
// CLUSTER_JACOBI, the pairs (i, j) that are edges of a degree-2 maximum spanning forest of the cluster graph:
//   vertices  the clusters (here: the clusters CLUSTER_JACOBI uses on this problem -- {shared intrinsics block, its
//             views}, or the visibility clusters -- and every other reduced block as a cluster of its own), numbered by
//             their lowest reduced block;
//   edges     two clusters that see a common non-constant track, weight = the number of such tracks;
//   forest    edges in decreasing order of (weight, lower end, higher end) -- Ceres sorts pair<weight, pair<v1, v2>>
//             with reverse iterators --, an edge is taken unless an end has two edges already or the ends are
//             connected already.
// The components are paths.  The preconditioner of a path is its block-tridiagonal matrix; it is factored as ONE dense
// "cluster" of cluster_precond.h whose tiles outside the band stay zero (the factor of a block-tridiagonal matrix has no
// fill outside the band), so a path is cut where the next cluster would take it beyond TMI_BA_MAX_CLUSTER_DIM unknowns.
// Restated in oracle/ba_oracle.c (tridiagonal_segments); the two must agree segment for segment.  Parity with Ceres:
// unpinned, like the rest of the Ceres layer.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <unordered_map>
#include <vector>

namespace tmi {
namespace chains {

struct Segments {
  std::vector<std::vector<int> > members;  // reduced blocks of a segment: cluster by cluster along the path, ascending inside
  std::vector<std::vector<int> > ordinal;  // position of the member's cluster in its segment
};

// cl_of[rb] = base cluster of every reduced block (-1: a cluster of its own); rb_dim = true block dimensions;
// obs_rb / obs_point: reduced block of the observing view (-1: none) and track of every observation;
// point_constant may be null
inline Segments build(std::vector<int> cl_of, const std::vector<int>& rb_dim, int64_t n_obs, const int32_t* obs_cam,
                      const std::vector<int>& cam_rb, const int32_t* obs_point, int32_t n_points,
                      const uint8_t* point_constant, int max_dim) {
  const int n = (int)cl_of.size();
  int ncl = 0;
  for (int b = 0; b < n; ++b) ncl = std::max(ncl, cl_of[b] + 1);
  for (int b = 0; b < n; ++b)
    if (cl_of[b] < 0) cl_of[b] = ncl++;
  {  // number the clusters by their lowest reduced block
    std::vector<int> renum((size_t)ncl, -1);
    int m = 0;
    for (int b = 0; b < n; ++b)
      if (renum[cl_of[b]] < 0) renum[cl_of[b]] = m++;
    for (int b = 0; b < n; ++b) cl_of[b] = renum[cl_of[b]];
    ncl = m;
  }
  std::vector<long long> cdim((size_t)ncl, 0);
  for (int b = 0; b < n; ++b) cdim[cl_of[b]] += rb_dim[b];
  // tracks seen from both clusters: the observations grouped by track (counting sort), distinct clusters per track
  std::unordered_map<uint64_t, double> W;
  {
    std::vector<int64_t> ptr((size_t)n_points + 1, 0);
    for (int64_t i = 0; i < n_obs; ++i) ptr[obs_point[i] + 1]++;
    for (int32_t p = 0; p < n_points; ++p) ptr[p + 1] += ptr[p];
    std::vector<int> ocl((size_t)std::max<int64_t>(n_obs, 1));
    {
      std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
      for (int64_t i = 0; i < n_obs; ++i) {
        const int rb = cam_rb[obs_cam[i]];
        ocl[fill[obs_point[i]]++] = rb >= 0 ? cl_of[rb] : -1;
      }
    }
    std::vector<int> seen;
    for (int32_t p = 0; p < n_points; ++p) {
      if (point_constant && point_constant[p]) continue;
      seen.clear();
      for (int64_t k = ptr[p]; k < ptr[p + 1]; ++k)
        if (ocl[k] >= 0 && std::find(seen.begin(), seen.end(), ocl[k]) == seen.end()) seen.push_back(ocl[k]);
      for (size_t a = 0; a < seen.size(); ++a)
        for (size_t b = 0; b < seen.size(); ++b)
          if (seen[a] < seen[b]) W[(uint64_t)seen[a] * (uint64_t)ncl + (uint64_t)seen[b]] += 1.0;
    }
  }
  struct Edge {
    double w;
    int a, b;
  };
  std::vector<Edge> edges;
  edges.reserve(W.size());
  for (const auto& kv : W) edges.push_back(Edge{kv.second, (int)(kv.first / (uint64_t)ncl), (int)(kv.first % (uint64_t)ncl)});
  std::sort(edges.begin(), edges.end(), [](const Edge& x, const Edge& y) {
    if (x.w != y.w) return x.w > y.w;
    if (x.a != y.a) return x.a > y.a;
    return x.b > y.b;
  });
  std::vector<int> deg((size_t)ncl, 0), nb((size_t)2 * ncl, -1), parent((size_t)ncl);
  std::iota(parent.begin(), parent.end(), 0);
  auto find = [&](int a) {
    while (parent[a] != a) a = parent[a];
    return a;
  };
  for (const Edge& e : edges) {
    if (deg[e.a] == 2 || deg[e.b] == 2) continue;
    int ra = find(e.a), rb = find(e.b);
    if (ra == rb) continue;
    nb[2 * e.a + deg[e.a]++] = e.b;
    nb[2 * e.b + deg[e.b]++] = e.a;
    if (rb < ra) std::swap(ra, rb);
    parent[rb] = ra;
  }
  std::vector<std::vector<int> > cl_members((size_t)ncl);
  for (int b = 0; b < n; ++b) cl_members[cl_of[b]].push_back(b);  // ascending
  Segments out;
  std::vector<char> visited((size_t)ncl, 0);
  std::vector<int> path;
  for (int c0 = 0; c0 < ncl; ++c0) {
    if (visited[c0] || deg[c0] == 2) continue;
    path.clear();
    for (int prev = -1, cur = c0; cur >= 0;) {
      visited[cur] = 1;
      path.push_back(cur);
      int next = -1;
      for (int k = 0; k < deg[cur]; ++k)
        if (nb[2 * cur + k] != prev) next = nb[2 * cur + k];
      prev = cur;
      cur = next;
    }
    size_t i = 0;
    while (i < path.size()) {
      if (cdim[path[i]] > max_dim) {  // keeps its SCHUR_JACOBI blocks
        ++i;
        continue;
      }
      size_t j = i;
      long long dim = 0;
      while (j < path.size() && cdim[path[j]] <= max_dim && dim + cdim[path[j]] <= max_dim) dim += cdim[path[j++]];
      std::vector<int> mem, ord;
      for (size_t q = i; q < j; ++q)
        for (const int b : cl_members[path[q]]) {
          mem.push_back(b);
          ord.push_back((int)(q - i));
        }
      if (mem.size() >= 2) {  // a single reduced block is its own SCHUR_JACOBI block
        out.members.push_back(mem);
        out.ordinal.push_back(ord);
      }
      i = j;
    }
  }
  return out;
}

}  // namespace chains
}  // namespace tmi
