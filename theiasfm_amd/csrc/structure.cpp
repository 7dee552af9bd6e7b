// Host-side structure builder (see structure.h).  Plain C++17, no device code.
//
// Residual-set semantics come from the flattened problem the host shim
// produces (bundle_adjuster.cc:102-180); block constancy follows
// SetCameraExtrinsicsParameterization / SetCameraIntrinsicsParameterization
// (bundle_adjuster.cc:223-287): constant coordinates simply have no column.
#include "structure.h"
#include "device_view.h"

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <numeric>
#include <atomic>
#include <thread>

namespace tmi {

namespace {

int model_size(int m) {
  static const int n[5] = {7, 10, 9, 5, 5};
  return (m >= 0 && m < 5) ? n[m] : -1;
}

struct KeyMap {  // open addressing: uint64 key -> int value
  std::vector<uint64_t> keys;
  std::vector<int> vals;
  uint64_t mask = 0;
  int64_t n = 0;
  static uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
  }
  void init(uint64_t cap) {
    uint64_t c = 1024;
    while (c < cap) c <<= 1;
    keys.assign(c, ~0ULL);
    vals.assign(c, -1);
    mask = c - 1;
    n = 0;
  }
  void grow() {
    std::vector<uint64_t> ok;
    std::vector<int> ov;
    ok.swap(keys);
    ov.swap(vals);
    init((mask + 1) * 2);
    for (size_t i = 0; i < ok.size(); ++i)
      if (ok[i] != ~0ULL) put(ok[i], ov[i]);
  }
  // returns true if inserted
  bool put(uint64_t k, int v) {
    if (2 * (uint64_t)(n + 1) > mask + 1) grow();
    uint64_t i = mix(k) & mask;
    while (keys[i] != ~0ULL) {
      if (keys[i] == k) return false;
      i = (i + 1) & mask;
    }
    keys[i] = k;
    vals[i] = v;
    ++n;
    return true;
  }
  int* find(uint64_t k) {
    uint64_t i = mix(k) & mask;
    while (keys[i] != ~0ULL) {
      if (keys[i] == k) return &vals[i];
      i = (i + 1) & mask;
    }
    return nullptr;
  }
};

int pad_dim(int d) {
  if (d <= 6) return 6;
  if (d <= 9) return 9;
  if (d <= 12) return 12;
  return 16;
}

}  // namespace

namespace {
double wall_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// The camera side of the structure: argument checks that do not touch the observations, the
// reduced blocks and their column maps.  Shared by the host builder below and the device
// builder (structure_gpu.h), which sorts and scans the observation-sized arrays in HBM.
int build_blocks(const tmi_ba_problem* P, int rank, int world, Structure* S) {
  Structure& s = *S;
  if (!P || world < 1 || rank < 0 || rank >= world) {
    s.error = "null problem or bad rank/world";
    return TMI_BA_ERR_INVALID_ARGUMENT;
  }
  if (P->num_cameras < 0 || P->num_groups < 0 || P->num_points < 0 || P->num_observations < 0 ||
      (P->num_cameras && (!P->extrinsics || !P->camera_group)) ||
      (P->num_groups && (!P->group_model || !P->group_offset || !P->intrinsics)) ||
      (P->num_points && !P->points) ||
      (P->num_observations && (!P->obs_camera || !P->obs_point || !P->obs_xy))) {
    s.error = "null array or negative size in tmi_ba_problem";
    return TMI_BA_ERR_INVALID_ARGUMENT;
  }
  s.rank = rank;
  s.world = world;
  s.Nc = P->num_cameras;
  s.G = P->num_groups;
  s.Np_total = P->num_points;
  for (int c = 0; c < s.Nc; ++c)
    if (P->camera_group[c] < 0 || P->camera_group[c] >= s.G) {
      s.error = "camera_group out of range";
      return TMI_BA_ERR_INVALID_ARGUMENT;
    }
  for (int g = 0; g < s.G; ++g) {
    const int n = model_size(P->group_model[g]);
    if (n < 0 || P->group_offset[g + 1] - P->group_offset[g] != n) {
      s.error = "bad camera model or group_offset";
      return TMI_BA_ERR_INVALID_ARGUMENT;
    }
  }
  // ---- reduced camera blocks --------------------------------------------------
  // A view whose intrinsics group is private gets ONE block [free extrinsics | free
  // intrinsics].  Free intrinsics SHARED by several views (one Ceres block per group,
  // camera.h:247 / reconstruction.cc:113-124) become a block of their own, placed after
  // the camera blocks; the views of such a group keep an extrinsics-only block.
  std::vector<int> grp_count(s.G, 0);
  for (int c = 0; c < s.Nc; ++c) grp_count[P->camera_group[c]]++;
  std::vector<uint32_t> grp_free(s.G, 0);
  s.grp_mask.assign(s.G, 0);
  s.has_shared = false;
  for (int g = 0; g < s.G; ++g) {
    const int o = P->group_offset[g], n = P->group_offset[g + 1] - o;
    for (int a = 0; a < n; ++a)
      if (!P->intrinsics_constant || !P->intrinsics_constant[o + a]) grp_free[g] |= 1u << a;
    if (grp_free[g] && grp_count[g] > 1) {
      s.grp_mask[g] = grp_free[g];
      s.has_shared = true;
    }
  }
  s.grp_free = grp_free;
  if (s.Nc) s.cam_group.assign(P->camera_group, P->camera_group + s.Nc);
  if (P->group_offset) s.group_offset.assign(P->group_offset, P->group_offset + s.G + 1);
  else s.group_offset.assign(1, 0);
  s.cam_mask.assign(s.Nc, 0);
  s.cam_rb.assign(s.Nc, -1);
  s.cam_grb.assign(s.Nc, -1);
  int maxdim = 0;
  for (int c = 0; c < s.Nc; ++c) {
    const int f = P->camera_flags ? P->camera_flags[c] : 0;
    const int g = P->camera_group[c];
    uint32_t m = 0;
    if (!(f & TMI_BA_CAMERA_POSITION_CONSTANT)) m |= 0x07;
    if (!(f & TMI_BA_CAMERA_ORIENTATION_CONSTANT)) m |= 0x38;
    if (!s.grp_mask[g]) m |= grp_free[g] << 6;
    s.cam_mask[c] = m;
    const int d = __builtin_popcount(m);
    // a view of a shared free group keeps a (possibly empty, all padding) block: its
    // camera-major records carry the intrinsics Jacobian for the per-camera sums
    if (d > 0 || s.grp_mask[g]) {
      s.cam_rb[c] = s.Nrb++;
      s.rb_cam.push_back(c);
      s.rb_grp.push_back(g);
      s.rb_dim.push_back(d);
      maxdim = std::max(maxdim, d);
    }
  }
  s.Ncam_rb = s.Nrb;
  std::vector<int> grp_rb(s.G, -1);
  for (int g = 0; g < s.G; ++g)
    if (s.grp_mask[g]) {
      grp_rb[g] = s.Nrb++;
      s.rb_cam.push_back(-1);
      s.rb_grp.push_back(g);
      const int d = __builtin_popcount(s.grp_mask[g]);
      s.rb_dim.push_back(d);
      maxdim = std::max(maxdim, d);
    }
  for (int c = 0; c < s.Nc; ++c) s.cam_grb[c] = grp_rb[P->camera_group[c]];
  s.D = pad_dim(std::max(maxdim, 1));
  s.rb_cols.assign((size_t)s.Nrb * s.D, -1);
  for (int rb = 0; rb < s.Nrb; ++rb) {
    int col = 0;
    const uint32_t m = s.rb_cam[rb] >= 0 ? s.cam_mask[s.rb_cam[rb]] : (s.grp_mask[s.rb_grp[rb]] << 6);
    for (int b = 0; b < 16; ++b)
      if (m & (1u << b)) s.rb_cols[(size_t)rb * s.D + col++] = (int8_t)b;
  }

  return TMI_BA_OK;
}

int build_structure(const tmi_ba_problem* P, int rank, int world, Structure* S, int want_pairs_mode) {
  const bool timing = std::getenv("TMI_BA_SETUP_TIMING") != nullptr;
  double t_phase = wall_s();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const double now = wall_s();
    fprintf(stderr, "[tmi_ba setup] %-28s %.3f s\n", what, now - t_phase);
    t_phase = now;
  };
  {
    const int rc = build_blocks(P, rank, world, S);
    if (rc) return rc;
  }
  Structure& s = *S;
  // cluster-only blocks (want_pairs 2): a block (a, b) is kept when a and b lie in the same cluster {shared block, its views}
  const bool cluster_only = want_pairs_mode == 2 && s.has_shared;
  const bool want_pairs = want_pairs_mode == 1 || cluster_only;
  std::vector<int> cluster_of(s.Nrb, -1);
  if (cluster_only) {
    for (int c = 0; c < s.Nc; ++c)
      if (s.cam_rb[c] >= 0) cluster_of[s.cam_rb[c]] = s.cam_grb[c];
    for (int rb = s.Ncam_rb; rb < s.Nrb; ++rb) cluster_of[rb] = rb;
  }
  auto keep_block = [&](int a, int b) { return !cluster_only || (cluster_of[a] >= 0 && cluster_of[a] == cluster_of[b]); };
  const int64_t No_all = P->num_observations;
  for (int64_t i = 0; i < No_all; ++i)
    if (P->obs_camera[i] < 0 || P->obs_camera[i] >= s.Nc || P->obs_point[i] < 0 ||
        P->obs_point[i] >= s.Np_total) {
      s.error = "observation index out of range";
      return TMI_BA_ERR_INVALID_ARGUMENT;
    }
  lap("reduced blocks");
  // ---- tracks: lengths, order by descending length, shard ------------------------
  // Host threads for the observation-sized passes of this phase (this builder serves the shapes the device builder
  // does not: shared intrinsics blocks, sharded handles that form S).  Every result is independent of the thread
  // count: concurrent passes only fill slots whose final order a sort or a per-thread prefix fixes afterwards.
  const int n_early = (No_all < 200000) ? 1 : (int)std::max(1u, std::min(32u, std::thread::hardware_concurrency() / 2));
  auto run_early = [&](auto&& body) {
    if (n_early == 1) {
      body(0);
      return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < n_early; ++t) pool.emplace_back([&, t] { body(t); });
    for (auto& th : pool) th.join();
  };
  std::vector<int> klen(s.Np_total, 0);
  run_early([&](int t) {
    const int64_t i0 = No_all * t / n_early, i1 = No_all * (t + 1) / n_early;
    for (int64_t i = i0; i < i1; ++i) __atomic_fetch_add(&klen[P->obs_point[i]], 1, __ATOMIC_RELAXED);
  });
  int kmax = 0;
  for (int p = 0; p < s.Np_total; ++p) kmax = std::max(kmax, klen[p]);
  // CSR of observations by track (the order inside a track is fixed by the sort below)
  std::vector<int64_t> tptr(s.Np_total + 1, 0);
  for (int p = 0; p < s.Np_total; ++p) tptr[p + 1] = tptr[p] + klen[p];
  std::vector<int64_t> tobs(No_all);
  {
    std::vector<int64_t> fill(tptr.begin(), tptr.end() - 1);
    run_early([&](int t) {
      const int64_t i0 = No_all * t / n_early, i1 = No_all * (t + 1) / n_early;
      for (int64_t i = i0; i < i1; ++i) tobs[__atomic_fetch_add(&fill[P->obs_point[i]], (int64_t)1, __ATOMIC_RELAXED)] = i;
    });
  }
  // inside a track: a deterministic order; reject a view observing a track twice
  {
    std::atomic<int> twice(0);
    run_early([&](int t) {
      const int p0 = (int)((int64_t)s.Np_total * t / n_early), p1 = (int)((int64_t)s.Np_total * (t + 1) / n_early);
      for (int p = p0; p < p1; ++p) {
        auto b = tobs.begin() + tptr[p], e = tobs.begin() + tptr[p + 1];
        // observations of one shared intrinsics block are kept adjacent (their Y factors
        // are summed on the fly), then ascending camera index
        std::sort(b, e, [&](int64_t x, int64_t y) {
          const int cx = P->obs_camera[x], cy = P->obs_camera[y];
          if (s.cam_grb[cx] != s.cam_grb[cy]) return s.cam_grb[cx] < s.cam_grb[cy];
          return cx != cy ? cx < cy : x < y;
        });
        for (auto it = b; it != e && it + 1 != e; ++it)
          if (P->obs_camera[*it] == P->obs_camera[*(it + 1)]) twice.store(1, std::memory_order_relaxed);
      }
    });
    if (twice.load()) {
      s.error = "a track is observed twice by the same view";
      return TMI_BA_ERR_INVALID_ARGUMENT;
    }
  }
  // Order: descending length, then the track's lowest view index, then its index (two stable counting
  // sorts).  Tracks seen from the same views share slices, so a wave's parameter gathers and the gathers of a
  // view's slots in the matrix-free product touch neighbouring memory.  Tracks without observations take no part.
  std::vector<int> order;
  order.reserve(s.Np_total);
  {
    std::vector<int> by_cam;
    by_cam.reserve(s.Np_total);
    {
      std::vector<int> first_cam(s.Np_total, 0);
      run_early([&](int t) {
        const int p0 = (int)((int64_t)s.Np_total * t / n_early), p1 = (int)((int64_t)s.Np_total * (t + 1) / n_early);
        for (int p = p0; p < p1; ++p) {
          int c = s.Nc;
          for (int64_t q = tptr[p]; q < tptr[p + 1]; ++q) c = std::min(c, (int)P->obs_camera[tobs[q]]);
          first_cam[p] = c;
        }
      });
      std::vector<int> cpos(s.Nc + 2, 0);
      int total = 0;
      for (int p = 0; p < s.Np_total; ++p)
        if (klen[p] > 0) {
          cpos[first_cam[p] + 1]++;
          ++total;
        }
      for (int c = 0; c <= s.Nc; ++c) cpos[c + 1] += cpos[c];
      by_cam.assign(total, 0);
      for (int p = 0; p < s.Np_total; ++p)
        if (klen[p] > 0) by_cam[cpos[first_cam[p]]++] = p;
    }
    // bucket b = kmax - k  (b = 0 holds the longest tracks)
    std::vector<int> pos(kmax + 2, 0);
    for (int p : by_cam) pos[kmax - klen[p] + 1]++;
    for (int b = 0; b <= kmax; ++b) pos[b + 1] += pos[b];
    order.assign(by_cam.size(), 0);
    for (int p : by_cam) order[pos[kmax - klen[p]]++] = p;
  }
  const int n_active = (int)order.size();
  // tracks without observations are in no slice; rank 0 answers for them in the
  // per-track side kernels (outlier filter / batched track adjustment)
  s.unobserved.clear();
  if (rank == 0)
    for (int p = 0; p < s.Np_total; ++p)
      if (klen[p] == 0) s.unobserved.push_back(p);
  const int gslices = (n_active + 63) / 64;
  // Deal the slices (already in descending track-length order) to the ranks by
  // longest-processing-time-first on the estimated work of a slice:
  //   Schur pairs k(k-1)/2  +  5 per observation (the per-observation kernels cost
  //   about 5x a pair; profiles/r01_a) when S is formed; with the matrix-free operator -- the default
  //   on several ranks -- there are no pairs and the work of a track is its observations (+ 1/2 for
  //   the per-track records): the pair rule left rank 0 of 8 with 228 k of the 5.0 M observations of
  //   venice1778_heavy (the 400-view tracks) and the other seven with 580-706 k (round 5; rank 0 then
  //   also fell below the size from which the one-sweep product is built).  With observation counts
  //   every rank holds 625 k; rank 0 -- the slice of the very longest tracks -- still takes 0.87 ms per
  //   LM iteration against 0.78 ms for rank 7 (tools/scale_probe.py times first and last rank):
  //   weighting tracks of 16+ views 1.5x moved 85 k observations off rank 0 and changed nothing, its
  //   time is the pace of its 400-view tracks, which no dealing of whole tracks can split.
  //   Deterministic: ties go to the lowest rank.  (engine.hip, build_structure_device: the same rule.)
  std::vector<int> slice_rank(gslices, 0);
  {
    std::vector<double> load(world, 0.0);
    for (int gs = 0; gs < gslices; ++gs) {
      double w = 0.0;
      for (int t = 0; t < 64; ++t) {
        const int idx = gs * 64 + t;
        if (idx < n_active) {
          const double k = klen[order[idx]];
          w += (want_pairs_mode == 1) ? 0.5 * k * (k - 1.0) + 5.0 * k : k + 0.5;
        }
      }
      int best = 0;
      for (int r = 1; r < world; ++r)
        if (load[r] < load[best]) best = r;
      slice_rank[gs] = best;
      load[best] += w;
    }
  }
  std::vector<int> local_pts;
  for (int gs = 0; gs < gslices; ++gs) {
    if (slice_rank[gs] != rank) continue;
    for (int t = 0; t < 64; ++t) {
      const int idx = gs * 64 + t;
      local_pts.push_back(idx < n_active ? order[idx] : -1);
    }
  }
  s.nslices = (int)local_pts.size() / 64;
  s.Np_pad = s.nslices * 64;
  s.Np = 0;
  s.pt_orig.assign(s.Np_pad, -1);
  s.pt_k.assign(s.Np_pad, 0);
  s.pt_const.assign(s.Np_pad, 0);
  s.slice_ptr.assign(s.nslices + 1, 0);
  for (int sl = 0; sl < s.nslices; ++sl) {
    int K = 0;
    for (int t = 0; t < 64; ++t) {
      const int p = local_pts[sl * 64 + t];
      if (p >= 0) {
        s.pt_orig[sl * 64 + t] = p;
        s.pt_k[sl * 64 + t] = klen[p];
        s.pt_const[sl * 64 + t] = (P->point_constant && P->point_constant[p]) ? 1 : 0;
        K = std::max(K, klen[p]);
        s.Np++;
      }
    }
    const int64_t next = (int64_t)s.slice_ptr[sl] + (int64_t)K * 64;
    if (next > 0x7fffffff) {
      s.error = "too many observations on one rank for 32-bit slot indices";
      return TMI_BA_ERR_UNSUPPORTED;
    }
    s.slice_ptr[sl + 1] = (int)next;
  }
  // leading slices (the order is by descending length) that the per-track kernels run with 16
  // lanes per track; the shared-intrinsics kernels accumulate runs of adjacent observations
  // serially and keep one thread per track
  s.n_wide = 0;
  {
    // Measured on the Venice-sized problem (tools/exp_wide.sh): with the whole problem on one
    // GPU (15.5 k slices) the bulk hides most of the tail and a high threshold is best
    // (5.15 ms per iteration at 20-24 vs 5.26 at 12); with an eighth of the tracks the
    // longest thread-per-track slice is the critical path and 10-12 wins (1.77 vs 2.05 ms).
    int wide_k = s.nslices >= 5000 ? kWideKLarge : kWideK;
    if (const char* e = std::getenv("TMI_BA_WIDE_K")) wide_k = std::max(1, std::atoi(e));  // tuning knob
    if (!s.has_shared)
      while (s.n_wide < s.nslices && ((s.slice_ptr[s.n_wide + 1] - s.slice_ptr[s.n_wide]) >> 6) >= wide_k)
        ++s.n_wide;
    // the very long tracks take a whole wavefront each: a 400-view track is 25 dependent trips with 16
    // lanes, which alone is the run time of the per-track kernels once the tracks are spread over 8 GPUs
    int ultra_k = kUltraK;
    if (const char* e = std::getenv("TMI_BA_ULTRA_K")) ultra_k = std::max(1, std::atoi(e));
    s.n_ultra = 0;
    while (s.n_ultra < s.n_wide && ((s.slice_ptr[s.n_ultra + 1] - s.slice_ptr[s.n_ultra]) >> 6) >= ultra_k) ++s.n_ultra;
  }
  s.No_pad = s.slice_ptr[s.nslices];
  s.obs_cam.assign(s.No_pad, -1);
  s.obs_xy.assign(2 * (size_t)s.No_pad, 0.0);
  s.obs_cpos.assign(s.No_pad, -1);
  s.obs_orig.assign(s.No_pad, -1);
  s.obs_gslot.assign(s.has_shared ? s.No_pad : 0, -1);
  s.obs_gflag.assign(s.has_shared ? s.No_pad : 0, 0);
  s.No = 0;
  // camera-major slot counts: one slot per observation in its camera's block, and one
  // slot per (track, shared intrinsics block) in that block
  // (threads own contiguous ranges of the layout; a slot's number = the block's first slot + the slots of the
  // earlier threads in that block + the thread's own running count: the numbering of the sequential loop)
  std::vector<std::vector<int>> slot_cnt_t(n_early, std::vector<int>(s.Nrb + 1, 0));
  std::vector<int64_t> no_t(n_early, 0);
  run_early([&](int th) {
    std::vector<int>& slot_cnt = slot_cnt_t[th];
    const int lp0 = (int)((int64_t)s.Np_pad * th / n_early), lp1 = (int)((int64_t)s.Np_pad * (th + 1) / n_early);
    for (int lp = lp0; lp < lp1; ++lp) {
      const int p = s.pt_orig[lp];
      if (p < 0) continue;
      const int sl = lp >> 6, t = lp & 63;
      int prev_g = -1;
      for (int j = 0; j < klen[p]; ++j) {
        const int64_t i = tobs[tptr[p] + j];
        const int64_t e = (int64_t)s.slice_ptr[sl] + (int64_t)j * 64 + t;
        const int c = P->obs_camera[i];
        s.obs_cam[e] = c;
        s.obs_xy[2 * e] = P->obs_xy[2 * i];
        s.obs_xy[2 * e + 1] = P->obs_xy[2 * i + 1];
        s.obs_orig[e] = i;
        const int rb = s.cam_rb[c];
        if (rb >= 0) slot_cnt[rb + 1]++;
        const int g = s.cam_grb[c];
        if (g >= 0 && g != prev_g) slot_cnt[g + 1]++;
        prev_g = g;
        no_t[th]++;
      }
    }
  });
  for (int th = 0; th < n_early; ++th) s.No += no_t[th];
  s.cam_ptr.assign(s.Nrb + 1, 0);
  for (int rb = 0; rb < s.Nrb; ++rb) {
    int tot = 0;
    for (int th = 0; th < n_early; ++th) {
      const int c = slot_cnt_t[th][rb + 1];
      slot_cnt_t[th][rb + 1] = tot;  // slots of this block owned by earlier threads
      tot += c;
    }
    s.cam_ptr[rb + 1] = s.cam_ptr[rb] + tot;
  }
  s.Nslots = s.cam_ptr[s.Nrb];
  run_early([&](int th) {
    std::vector<int> fill(s.Nrb, 0);
    for (int rb = 0; rb < s.Nrb; ++rb) fill[rb] = s.cam_ptr[rb] + slot_cnt_t[th][rb + 1];
    const int lp0 = (int)((int64_t)s.Np_pad * th / n_early), lp1 = (int)((int64_t)s.Np_pad * (th + 1) / n_early);
    for (int lp = lp0; lp < lp1; ++lp) {
      const int p = s.pt_orig[lp];
      if (p < 0) continue;
      const int sl = lp >> 6, t = lp & 63;
      int prev_g = -1, cur_slot = -1;
      for (int j = 0; j < klen[p]; ++j) {
        const int64_t e = (int64_t)s.slice_ptr[sl] + (int64_t)j * 64 + t;
        const int c = s.obs_cam[e];
        const int rb = s.cam_rb[c];
        if (rb >= 0) s.obs_cpos[e] = fill[rb]++;
        if (!s.has_shared) continue;
        const int g = s.cam_grb[c];
        if (g >= 0) {
          int flag = 0;
          if (g != prev_g) {
            cur_slot = fill[g]++;
            flag |= 1;  // first observation of the run
          }
          s.obs_gslot[e] = cur_slot;
          // last of the run?  look ahead
          bool last = (j + 1 == klen[p]);
          if (!last) {
            const int64_t en = (int64_t)s.slice_ptr[sl] + (int64_t)(j + 1) * 64 + t;
            last = s.cam_grb[s.obs_cam[en]] != g;
          }
          if (last) flag |= 2;
          s.obs_gflag[e] = (uint8_t)flag;
        }
        prev_g = g;
      }
    }
  });

  lap("track order, layout, slots");
  // ---- block structure of S from ALL tracks (rank independent) -------------------
  KeyMap blocks;
  blocks.init(1 << 16);
  // Host threads for the two passes over the tracks (block set, pair lists): the pair lists
  // of a Venice-sized problem are 16 M entries behind 16 M hash look-ups -- single-threaded
  // they were 80 % of the set-up time.  Every result below is independent of the thread count.
  const int n_threads = (!want_pairs || No_all < 200000)
                            ? 1
                            : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
  auto run_threads = [&](auto&& body) {  // body(thread index)
    if (n_threads == 1) {
      body(0);
      return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < n_threads; ++t) pool.emplace_back([&, t] { body(t); });
    for (auto& th : pool) th.join();
  };
  // reduced blocks a track touches, ascending: camera blocks, then (once each) the shared
  // intrinsics blocks
  auto track_blocks = [&](std::vector<int>& rbs, auto&& cam_of, int k) {
    rbs.clear();
    int prev_g = -1;
    for (int j = 0; j < k; ++j) {
      const int c = cam_of(j);
      if (s.cam_rb[c] >= 0) rbs.push_back(s.cam_rb[c]);
    }
    for (int j = 0; j < k; ++j) {
      const int g = s.cam_grb[cam_of(j)];
      if (g >= 0 && g != prev_g) rbs.push_back(g);
      prev_g = g;
    }
    std::sort(rbs.begin(), rbs.end());
  };
  const bool dense_flags = want_pairs && (int64_t)s.Nrb * s.Nrb <= (int64_t)64 << 20;
  if (dense_flags) {
    // one byte per (row, column): set concurrently (every writer stores the same 1), then
    // swept in row-major order into the key map
    std::vector<uint8_t> present((size_t)s.Nrb * s.Nrb, 0);
    run_threads([&](int t) {
      std::vector<int> rbs;
      const int p0 = (int)((int64_t)s.Np_total * t / n_threads), p1 = (int)((int64_t)s.Np_total * (t + 1) / n_threads);
      for (int p = p0; p < p1; ++p) {
        if (klen[p] < 1) continue;
        if (P->point_constant && P->point_constant[p]) continue;  // no elimination, no coupling
        track_blocks(rbs, [&](int j) { return P->obs_camera[tobs[tptr[p] + j]]; }, klen[p]);
        for (size_t a = 0; a < rbs.size(); ++a)
          for (size_t b = a + 1; b < rbs.size(); ++b)
            if (keep_block(rbs[a], rbs[b]))
              __atomic_store_n(&present[(size_t)rbs[a] * s.Nrb + rbs[b]], (uint8_t)1, __ATOMIC_RELAXED);
      }
    });
    for (int a = 0; a < s.Nrb; ++a)
      for (int b = a + 1; b < s.Nrb; ++b)
        if (present[(size_t)a * s.Nrb + b]) blocks.put(((uint64_t)a << 32) | (uint32_t)b, 0);
  } else {
    std::vector<int> rbs;
    for (int p = 0; p < s.Np_total && want_pairs; ++p) {
      if (klen[p] < 1) continue;
      if (P->point_constant && P->point_constant[p]) continue;  // no elimination, no coupling
      track_blocks(rbs, [&](int j) { return P->obs_camera[tobs[tptr[p] + j]]; }, klen[p]);
      for (size_t a = 0; a < rbs.size(); ++a)
        for (size_t b = a + 1; b < rbs.size(); ++b)
          if (keep_block(rbs[a], rbs[b])) blocks.put(((uint64_t)rbs[a] << 32) | (uint32_t)rbs[b], 0);
    }
  }
  // J_c^T J_c couples a view's extrinsics with its shared intrinsics block even when
  // none of its tracks is eliminated
  if (s.has_shared && want_pairs) {
    std::vector<char> seen(s.Nc, 0);
    for (int64_t i = 0; i < No_all; ++i) seen[P->obs_camera[i]] = 1;
    for (int c = 0; c < s.Nc; ++c)
      if (seen[c] && s.cam_rb[c] >= 0 && s.cam_grb[c] >= 0)
        blocks.put(((uint64_t)s.cam_rb[c] << 32) | (uint32_t)s.cam_grb[c], 0);
  }
  std::vector<uint64_t> ukeys;
  ukeys.reserve(blocks.n);
  for (size_t i = 0; i < blocks.keys.size(); ++i)
    if (blocks.keys[i] != ~0ULL) ukeys.push_back(blocks.keys[i]);
  std::sort(ukeys.begin(), ukeys.end());
  s.nub = (int64_t)ukeys.size();
  if (s.nub * 2 + s.Nrb > 0x7fffffff) {
    s.error = "reduced camera matrix has too many blocks for 32-bit block indices";
    return TMI_BA_ERR_UNSUPPORTED;
  }
  s.ub_i.resize(s.nub);
  s.ub_j.resize(s.nub);
  for (int64_t u = 0; u < s.nub; ++u) {
    s.ub_i[u] = (int)(ukeys[u] >> 32);
    s.ub_j[u] = (int)(ukeys[u] & 0xffffffffu);
    *blocks.find(ukeys[u]) = (int)u;
  }
  // dense (row, column) -> block index table for the pair pass (a hash look-up per pair otherwise)
  std::vector<int> blk_id;
  if (dense_flags) {
    blk_id.assign((size_t)s.Nrb * s.Nrb, -1);
    for (int64_t u = 0; u < s.nub; ++u) blk_id[(size_t)s.ub_i[u] * s.Nrb + s.ub_j[u]] = (int)u;
  }
  // symmetric storage: row ranges of the (bi, bj)-sorted upper list and a column index
  s.nnzb = 2 * s.nub + s.Nrb;
  s.urow_ptr.assign(s.Nrb + 1, 0);
  s.ucol_ptr.assign(s.Nrb + 1, 0);
  for (int64_t u = 0; u < s.nub; ++u) {
    s.urow_ptr[s.ub_i[u] + 1]++;
    s.ucol_ptr[s.ub_j[u] + 1]++;
  }
  for (int i = 0; i < s.Nrb; ++i) {
    s.urow_ptr[i + 1] += s.urow_ptr[i];
    s.ucol_ptr[i + 1] += s.ucol_ptr[i];
  }
  // balanced work list of the symmetric SpMV's rows pass: a block row is cut into chunks of
  // kSpmvTrips * (64 / D) consecutive upper blocks, one wavefront per chunk (kernels.h)
  {
    const int chunk = kSpmvTrips * (64 / std::max(s.D, 1));
    s.spc_row.clear();
    s.spc_u0.clear();
    s.spc_rptr.assign(s.Nrb + 1, 0);
    for (int i = 0; i < s.Nrb; ++i) {
      for (int u = s.urow_ptr[i]; u < s.urow_ptr[i + 1]; u += chunk) {
        s.spc_row.push_back(i);
        s.spc_u0.push_back(u);
      }
      s.spc_rptr[i + 1] = (int)s.spc_row.size();
    }
  }
  s.ucol_u.assign(s.nub, 0);
  {
    std::vector<int> fill(s.ucol_ptr.begin(), s.ucol_ptr.end() - 1);
    for (int64_t u = 0; u < s.nub; ++u) s.ucol_u[fill[s.ub_j[u]]++] = (int)u;  // ascending bi
  }

  lap("block structure of S");
  // ---- pair lists from this rank's tracks ----------------------------------------
  s.pair_ptr.assign(s.nub + 1, 0);
  // the (block, slot_i, slot_j) triples of the tracks [lp0, lp1), in track order
  auto for_each_pair = [&](int lp0, int lp1, auto&& fn) {
    std::vector<std::pair<int, int>> rs;  // (block, slot) of a track's virtual observations
    for (int lp = lp0; lp < lp1 && want_pairs; ++lp) {
      const int p = s.pt_orig[lp];
      if (p < 0 || s.pt_const[lp]) continue;
      const int sl = lp >> 6, t = lp & 63;
      rs.clear();
      int prev_g = -1;
      for (int j = 0; j < klen[p]; ++j) {
        const int64_t e = (int64_t)s.slice_ptr[sl] + (int64_t)j * 64 + t;
        const int c = s.obs_cam[e];
        if (s.cam_rb[c] >= 0) rs.emplace_back(s.cam_rb[c], s.obs_cpos[e]);
        const int g = s.cam_grb[c];
        if (g >= 0 && g != prev_g) rs.emplace_back(g, s.obs_gslot[e]);
        prev_g = g;
      }
      std::sort(rs.begin(), rs.end());
      for (size_t a = 0; a < rs.size(); ++a)
        for (size_t b = a + 1; b < rs.size(); ++b) {
          if (!keep_block(rs[a].first, rs[b].first)) continue;
          const int u = blk_id.empty() ? *blocks.find(((uint64_t)rs[a].first << 32) | (uint32_t)rs[b].first)
                                       : blk_id[(size_t)rs[a].first * s.Nrb + rs[b].first];
          fn(u, rs[a].second, rs[b].second);
        }
    }
  };
  // contiguous track ranges of about equal pair work: thread t owns [cut[t], cut[t+1]).  A
  // block's pair list is ordered by track, whatever the number of threads.
  std::vector<int> cut(n_threads + 1, s.Np_pad);
  {
    std::vector<int64_t> work((size_t)s.Np_pad + 1, 0);
    for (int lp = 0; lp < s.Np_pad; ++lp) {
      const int64_t k = s.pt_k[lp];
      work[lp + 1] = work[lp] + k * (k - 1) / 2 + 1;
    }
    cut[0] = 0;
    for (int t = 1; t < n_threads; ++t)
      cut[t] = (int)(std::lower_bound(work.begin(), work.end(), work.back() * t / n_threads) - work.begin());
    for (int t = 1; t <= n_threads; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
    cut[n_threads] = s.Np_pad;
  }
  // count per (thread, block), then: pair_ptr = prefix over blocks, and the counts become the
  // thread's first position inside each block
  std::vector<std::vector<int>> cnt(n_threads);
  run_threads([&](int t) {
    cnt[t].assign((size_t)s.nub + 1, 0);
    for_each_pair(cut[t], cut[t + 1], [&](int u, int, int) { cnt[t][u]++; });
  });
  for (int64_t u = 0; u < s.nub; ++u) {
    int64_t tot = 0;
    for (int t = 0; t < n_threads; ++t) {
      const int c = cnt[t][u];
      cnt[t][u] = (int)tot;
      tot += c;
    }
    if (tot > 0x7fffffff) {
      s.error = "a block of the reduced camera matrix has too many observation pairs";
      return TMI_BA_ERR_UNSUPPORTED;
    }
    s.pair_ptr[u + 1] = s.pair_ptr[u] + tot;
  }
  s.npairs = s.pair_ptr[s.nub];
  s.pair_i.assign(s.npairs, 0);
  s.pair_j.assign(s.npairs, 0);
  run_threads([&](int t) {
    for_each_pair(cut[t], cut[t + 1], [&](int u, int si, int sj) {
      const int64_t pos = s.pair_ptr[u] + cnt[t][u]++;
      s.pair_i[pos] = si;
      s.pair_j[pos] = sj;
    });
  });
  lap("pair lists");
  // per view of a shared block: the S block (extrinsics, shared intrinsics) receiving
  // its J_c^T J_c cross term; per shared block: its views
  s.cam_cross_u.assign(s.Nc, -1);
  s.grp_cam_ptr.assign(s.Nrb - s.Ncam_rb + 1, 0);
  if (s.has_shared) {
    for (int c = 0; c < s.Nc; ++c)
      if (s.cam_rb[c] >= 0 && s.cam_grb[c] >= 0) {
        int* u = blocks.find(((uint64_t)s.cam_rb[c] << 32) | (uint32_t)s.cam_grb[c]);
        s.cam_cross_u[c] = u ? *u : -1;
        s.grp_cam_ptr[s.cam_grb[c] - s.Ncam_rb + 1]++;
      }
    for (int g = 0; g < s.Nrb - s.Ncam_rb; ++g) s.grp_cam_ptr[g + 1] += s.grp_cam_ptr[g];
    s.grp_cams.assign(s.grp_cam_ptr.back(), 0);
    std::vector<int> fill(s.grp_cam_ptr.begin(), s.grp_cam_ptr.end() - 1);
    for (int c = 0; c < s.Nc; ++c)
      if (s.cam_rb[c] >= 0 && s.cam_grb[c] >= 0) s.grp_cams[fill[s.cam_grb[c] - s.Ncam_rb]++] = c;
  }
  // Launch order of the upper blocks for schur_offdiag: one wavefront per block, four
  // per workgroup, and workgroup w runs on XCD w % 8 (observed dispatch, used for speed
  // only).  All blocks of a block row gather the SAME camera's Y records on their i
  // side, so a row is kept on one XCD (row % 8) and rows are walked in order: the i
  // side then hits that XCD's L2 and the j side (neighbouring rows look at the same
  // tracks) the memory-side Infinity Cache.  -1 entries pad the shorter queues.
  {
    std::vector<std::vector<int>> q(8);
    for (int64_t u = 0; u < s.nub; ++u) q[s.ub_i[u] & 7].push_back((int)u);
    // inside a row: longest pair lists first (they are the stragglers)
    for (auto& v : q) {
      size_t b = 0;
      while (b < v.size()) {
        size_t e = b;
        while (e < v.size() && s.ub_i[v[e]] == s.ub_i[v[b]]) ++e;
        std::stable_sort(v.begin() + b, v.begin() + e, [&](int a, int c) {
          return (s.pair_ptr[a + 1] - s.pair_ptr[a]) > (s.pair_ptr[c + 1] - s.pair_ptr[c]);
        });
        b = e;
      }
    }
    // workgroup w = m * 8 + x takes kSchurWG consecutive entries of queue x; its wave t the
    // R = kSchurWG / 4 entries [t R, t R + R) of those (engine: kSchurBlocksPerWave)
    constexpr size_t kSchurWG = 16;
    size_t longest = 0;
    for (auto& v : q) longest = std::max(longest, v.size());
    const size_t groups = (longest + kSchurWG - 1) / kSchurWG;
    s.ub_order.assign(groups * 8 * kSchurWG, -1);
    for (size_t m = 0; m < groups; ++m)
      for (int x = 0; x < 8; ++x)
        for (size_t t = 0; t < kSchurWG; ++t) {
          const size_t idx = m * kSchurWG + t;
          if (idx < q[x].size()) s.ub_order[(m * 8 + x) * kSchurWG + t] = q[x][idx];
        }
  }
  lap("launch order");
  return TMI_BA_OK;
}

}  // namespace tmi
