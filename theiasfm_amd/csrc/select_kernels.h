// Device side of SelectGoodTracksForBundleAdjustment and of the outlier filter's bookkeeping
// (reference select_good_tracks_for_bundle_adjustment.cc:146-249,
// set_outlier_tracks_to_unestimated.cc:62-133).  Round 1 ran the projections on the device and
// everything else -- per-view feature lists, per-cell minima, per-view top-up, counting, the
// permutation back to the caller's track order -- on the host (0.45 s per call at Venice size).
// Here:
//   * grid step (:146-196): each view's features fall into cells of an integer grid; the cell keeps
//     the minimum of (truncated length, mean error, track index).  No sort and no hash: a pass of
//     atomicMin / atomicMax gives every view's cell bounding box, a one-block scan turns the box
//     sizes into offsets of a dense cell array, then three passes of 32/64-bit atomicMin resolve
//     the lexicographic minimum key by key (minima are order independent => deterministic);
//   * top-up step (:201-249): views are visited IN ORDER and each visit depends on what earlier
//     visits selected, so one workgroup walks the views; a view's tracks come from a list sorted
//     by (view, track index) -- built once per handle with a device radix sort -- and "the needed
//     lowest-index unselected tracks" is a block-wide exclusive scan over that list;
//   * results are scattered to the caller's track order on the device and copied out once.
#pragma once
#include <hip/hip_runtime.h>

#include "device_view.h"
#include "kernels.h"

namespace tmi {

struct SelectView {
  int Nc;
  int Np_total;
  const unsigned char* view_mask;  // [Nc] or nullptr
  const int* pt_orig;              // [Np_pad] caller's track index or -1
  const int* cnt;                  // [Np_pad] observations per track (track_stats_kernel)
  const double* mean;              // [Np_pad] mean squared reprojection error
  int long_thr;
  double inv_cell;
  int* vbox;                       // [Nc][4] cx min, cx max, cy min, cy max
  long long* cell_off;             // [Nc + 1]
  unsigned* cell_len;              // per cell: minimum truncated length
  unsigned long long* cell_err;    //           then minimum error (IEEE bits: errors are >= 0)
  unsigned* cell_trk;              //           then minimum track index
  unsigned* sel;                   // [Np_total] 0 / 1
  int* counters;                   // [4] selected by the grid step, selected in total, ...
};

__device__ __forceinline__ void feature_cell(const DeviceView& v, const SelectView& S, size_t e, int* cx, int* cy) {
  *cx = (int)(v.obs_xy[2 * e] * S.inv_cell);  // (feature * inv_grid_cell_size).cast<int>(), :174
  *cy = (int)(v.obs_xy[2 * e + 1] * S.inv_cell);
}

__global__ __launch_bounds__(256) void select_init_kernel(SelectView S) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < S.Nc) {
    S.vbox[4 * i + 0] = 0x7fffffff;
    S.vbox[4 * i + 1] = (int)0x80000000;
    S.vbox[4 * i + 2] = 0x7fffffff;
    S.vbox[4 * i + 3] = (int)0x80000000;
  }
  if (i < S.Np_total) S.sel[i] = 0u;
  if (i < 4) S.counters[i] = 0;
}

// pass 0: cell bounding box of every view
__global__ __launch_bounds__(256) void select_bounds_kernel(DeviceView v, SelectView S) {
  const TrackMap tm = track_map(v);
  if (!tm.valid) return;
  for (int j = tm.j0; j < tm.k; j += tm.jstep) {
    const size_t e = tm.base + (size_t)j * 64;
    const int cam = v.obs_cam[e];
    if (S.view_mask && !S.view_mask[cam]) continue;
    int cx, cy;
    feature_cell(v, S, e, &cx, &cy);
    // the boxes settle after a few hundred observations: look before the atomic (a stale value only
    // costs an atomic that changes nothing)
    int* box = S.vbox + 4 * cam;
    if (cx < __atomic_load_n(box + 0, __ATOMIC_RELAXED)) atomicMin(box + 0, cx);
    if (cx > __atomic_load_n(box + 1, __ATOMIC_RELAXED)) atomicMax(box + 1, cx);
    if (cy < __atomic_load_n(box + 2, __ATOMIC_RELAXED)) atomicMin(box + 2, cy);
    if (cy > __atomic_load_n(box + 3, __ATOMIC_RELAXED)) atomicMax(box + 3, cy);
  }
}

// one block: exclusive scan of the box sizes -> cell_off[Nc + 1]
__global__ __launch_bounds__(1024) void select_offsets_kernel(SelectView S) {
  __shared__ long long part[1024];
  __shared__ long long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int c0 = 0; c0 < S.Nc; c0 += 1024) {
    const int c = c0 + threadIdx.x;
    long long n = 0;
    if (c < S.Nc && S.vbox[4 * c + 1] >= S.vbox[4 * c + 0])
      n = ((long long)S.vbox[4 * c + 1] - S.vbox[4 * c + 0] + 1) * ((long long)S.vbox[4 * c + 3] - S.vbox[4 * c + 2] + 1);
    part[threadIdx.x] = n;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const long long t = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    if (c < S.Nc) S.cell_off[c] = carry + part[threadIdx.x] - n;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) S.cell_off[S.Nc] = carry;
}

__global__ __launch_bounds__(256) void select_fill_cells_kernel(SelectView S, long long ncells) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < ncells) {
    S.cell_len[i] = 0xffffffffu;
    S.cell_err[i] = ~0ull;
    S.cell_trk[i] = 0xffffffffu;
  }
}

// passes 1..3: lexicographic minimum of (truncated length, mean error, track index) per cell
template <int PASS>
__global__ __launch_bounds__(256) void select_cells_kernel(DeviceView v, SelectView S) {
  const TrackMap tm = track_map(v);
  if (!tm.valid) return;
  const int lp = tm.lp;
  const int k = tm.k;
  if (k == 0) return;
  const int p = S.pt_orig[lp];
  const unsigned tlen = (unsigned)min(S.cnt[lp], S.long_thr);
  const unsigned long long ebits = (unsigned long long)__double_as_longlong(S.mean[lp]);
  const size_t base = tm.base;
  for (int j = tm.j0; j < k; j += tm.jstep) {
    const size_t e = base + (size_t)j * 64;
    const int cam = v.obs_cam[e];
    if (S.view_mask && !S.view_mask[cam]) continue;
    int cx, cy;
    feature_cell(v, S, e, &cx, &cy);
    const long long w = (long long)S.vbox[4 * cam + 1] - S.vbox[4 * cam + 0] + 1;
    const long long slot = S.cell_off[cam] + (long long)(cy - S.vbox[4 * cam + 2]) * w + (cx - S.vbox[4 * cam + 0]);
    // look before the atomic: most candidates lose against what the cell already holds
    if (PASS == 1) {
      if (tlen < __atomic_load_n(&S.cell_len[slot], __ATOMIC_RELAXED)) atomicMin(&S.cell_len[slot], tlen);
    } else if (PASS == 2) {
      if (S.cell_len[slot] == tlen && ebits < __atomic_load_n(&S.cell_err[slot], __ATOMIC_RELAXED))
        atomicMin(&S.cell_err[slot], ebits);
    } else {
      if (S.cell_len[slot] == tlen && S.cell_err[slot] == ebits &&
          (unsigned)p < __atomic_load_n(&S.cell_trk[slot], __ATOMIC_RELAXED))
        atomicMin(&S.cell_trk[slot], (unsigned)p);
    }
  }
}

__global__ __launch_bounds__(256) void select_mark_kernel(SelectView S, long long ncells) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= ncells) return;
  const unsigned t = S.cell_trk[i];
  if (t != 0xffffffffu && atomicExch(&S.sel[t], 1u) == 0u) atomicAdd(&S.counters[0], 1);
}

// key of the per-view track lists: (view << 32) | track index, ~0 for padding
__global__ __launch_bounds__(256) void select_keys_kernel(DeviceView v, const int* __restrict__ pt_orig,
                                                          unsigned long long* __restrict__ keys) {
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * kSlicesPerBlock + (threadIdx.x >> 6);
  if (s >= v.nslices) return;
  const int lp = s * 64 + lane;
  const int k = v.pt_k[lp];
  const int K = (v.slice_ptr[s + 1] - v.slice_ptr[s]) >> 6;
  const int p = pt_orig[lp];
  const size_t base = (size_t)v.slice_ptr[s] + lane;
  for (int j = 0; j < K; ++j) {
    const size_t e = base + (size_t)j * 64;
    keys[e] = (j < k) ? (((unsigned long long)(unsigned)v.obs_cam[e] << 32) | (unsigned)p) : ~0ull;
  }
}

// vt_ptr[c] = first sorted key of view c (binary search), c = 0..Nc
__global__ __launch_bounds__(256) void select_view_ptr_kernel(const unsigned long long* __restrict__ keys,
                                                              long long n, int Nc, long long* __restrict__ vt_ptr) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c > Nc) return;
  const unsigned long long want = (unsigned long long)(unsigned)c << 32;
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (keys[mid] < want) lo = mid + 1;
    else hi = mid;
  }
  vt_ptr[c] = lo;
}

// Top-up (:201-249): ONE workgroup visits the views in ascending order; a view with fewer than
// min_opt selected tracks (and unselected ones left) takes its lowest-index unselected tracks.
constexpr int kTopupThreads = 1024;
constexpr int kTopupPerThread = 8;  // slow path (views beyond the one-pass size): chunks of kTopupThreads * 8 tracks

// Selected tracks per view after the grid phase (one workgroup per view).  Selections only grow, so a
// view that already has its minimum here never needs the sequential top-up below.
__global__ __launch_bounds__(256) void select_view_count_kernel(SelectView S, const unsigned long long* __restrict__ keys,
                                                                const long long* __restrict__ vt_ptr,
                                                                int* __restrict__ vcount) {
  __shared__ int ws[4];
  const int c = blockIdx.x;
  const long long b = vt_ptr[c], e = vt_ptr[c + 1];
  int mine = 0;
  for (long long q = b + threadIdx.x; q < e; q += 256) mine += (int)S.sel[(unsigned)(keys[q] & 0xffffffffu)];
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) vcount[c] = ws[0] + ws[1] + ws[2] + ws[3];
}

// BITS: the selection flags of all tracks live in LDS as a bit vector for the duration of the kernel
// (Np / 8 bytes of dynamic LDS: 124 KB for a million tracks; the host picks the variant by size), so the
// per-view gather of flags -- the latency that sets the pace of this one-workgroup kernel -- never
// leaves the CU.  New selections are also stored to the global flags (write-only here).
template <bool BITS>
__global__ __launch_bounds__(kTopupThreads) void select_topup_kernel(SelectView S,
                                                                     const unsigned long long* __restrict__ keys,
                                                                     const long long* __restrict__ vt_ptr,
                                                                     const int* __restrict__ vcount,
                                                                     int* __restrict__ needy, int min_opt) {
  __shared__ int wsum[kTopupThreads / 64];
  __shared__ int sh_total;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  auto block_sum = [&](int x) {
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    __syncthreads();
    if (lane == 0) wsum[wv] = x;
    __syncthreads();
    int t = 0;
    for (int i = 0; i < kTopupThreads / 64; ++i) t += wsum[i];
    return t;
  };
  // exclusive prefix of x over the block (thread order), also returns the block total
  auto block_scan = [&](int x, int* total) {
    int inc = x;
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int before = 0, tot = 0;
    for (int i = 0; i < kTopupThreads / 64; ++i) {
      if (i < wv) before += wsum[i];
      tot += wsum[i];
    }
    *total = tot;
    return before + inc - x;
  };
  extern __shared__ unsigned sel_bits[];
  if (BITS) {
    const int nwords = (S.Np_total + 31) / 32;
    for (int wd = tid; wd < nwords; wd += kTopupThreads) {
      unsigned m = 0;
      const int p0 = wd * 32;
      if (p0 + 32 <= S.Np_total) {
        const uint4* src = reinterpret_cast<const uint4*>(S.sel + p0);  // 32 flags of 4 bytes, 16-byte aligned
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const uint4 u = src[h];
          if (u.x) m |= 1u << (4 * h);
          if (u.y) m |= 1u << (4 * h + 1);
          if (u.z) m |= 1u << (4 * h + 2);
          if (u.w) m |= 1u << (4 * h + 3);
        }
      } else {
        for (int i = 0; p0 + i < S.Np_total; ++i)
          if (S.sel[p0 + i]) m |= 1u << i;
      }
      sel_bits[wd] = m;
    }
    __syncthreads();
  }
  auto is_selected = [&](unsigned t) -> bool {
    return BITS ? ((sel_bits[t >> 5] >> (t & 31)) & 1u) != 0 : S.sel[t] != 0;
  };
  auto select = [&](unsigned t) {
    if (BITS) atomicOr(&sel_bits[t >> 5], 1u << (t & 31));
    S.sel[t] = 1u;
  };
  // the views that may need a top-up, ascending: everything else is skipped without touching memory
  __shared__ int n_needy_sh;
  int n_needy = 0;
  for (int c0 = 0; c0 < S.Nc; c0 += kTopupThreads) {
    const int c = c0 + tid;
    int want = 0;
    if (c < S.Nc && !(S.view_mask && !S.view_mask[c])) {
      const int n = (int)(vt_ptr[c + 1] - vt_ptr[c]);
      want = (n > 0 && vcount[c] < min_opt && vcount[c] < n) ? 1 : 0;
    }
    int total = 0;
    const int pos = block_scan(want, &total);
    if (want) needy[n_needy + pos] = c;
    n_needy += total;
  }
  if (tid == 0) n_needy_sh = n_needy;
  __threadfence_block();
  __syncthreads();
  n_needy = n_needy_sh;
  // Views of up to kTopupThreads * kTopupFast tracks take ONE pass: every thread holds its tracks and
  // their flags in registers, one block scan gives both the count of selected tracks and the ranks of
  // the unselected ones; the track ids of the next view are fetched while this one is decided (they do
  // not depend on the flags).  The views are a dependent chain (a top-up changes the counts of the views
  // after it), so this kernel is one workgroup and its time is views x (one gather + one scan).
  constexpr int kTopupFast = 4;
  unsigned nx_trk[kTopupFast];
  auto fetch = [&](int w, unsigned (&trk)[kTopupFast]) {
#pragma unroll
    for (int i = 0; i < kTopupFast; ++i) trk[i] = 0xffffffffu;
    if (w >= n_needy) return;
    const int c = needy[w];
    const long long b = vt_ptr[c], e = vt_ptr[c + 1];
    if (e - b > (long long)kTopupThreads * kTopupFast) return;
#pragma unroll
    for (int i = 0; i < kTopupFast; ++i) {
      const long long q = b + (long long)tid * kTopupFast + i;
      if (q < e) trk[i] = (unsigned)(keys[q] & 0xffffffffu);
    }
  };
  fetch(0, nx_trk);
  for (int w = 0; w < n_needy; ++w) {
    const int c = needy[w];
    const long long b = vt_ptr[c], e = vt_ptr[c + 1];
    const int n = (int)(e - b);
    if (n <= kTopupThreads * kTopupFast) {
      unsigned trk[kTopupFast];
      int uns[kTopupFast], local = 0;
#pragma unroll
      for (int i = 0; i < kTopupFast; ++i) {
        trk[i] = nx_trk[i];
        uns[i] = (trk[i] != 0xffffffffu && !is_selected(trk[i])) ? 1 : 0;
        local += uns[i];
      }
      fetch(w + 1, nx_trk);
      int total = 0;
      int rank = block_scan(local, &total);
      const int num_opt = n - total;
      if (num_opt < min_opt && total > 0) {
        const int needed = min(min_opt - num_opt, total);
#pragma unroll
        for (int i = 0; i < kTopupFast; ++i) {
          if (uns[i]) {
            if (rank < needed) select(trk[i]);
            ++rank;
          }
        }
      }
      __threadfence_block();
      __syncthreads();
      continue;
    }
    fetch(w + 1, nx_trk);
    // count the selected tracks of the view
    int mine = 0;
    for (long long q = b + tid; q < e; q += kTopupThreads) mine += is_selected((unsigned)(keys[q] & 0xffffffffu)) ? 1 : 0;
    const int num_opt = block_sum(mine);
    if (num_opt >= min_opt || num_opt == n) continue;
    int needed = min(min_opt - num_opt, n - num_opt);
    // the first `needed` unselected tracks in track-index order: contiguous blocks of the list per thread
    for (long long q0 = b; q0 < e && needed > 0; q0 += (long long)kTopupThreads * kTopupPerThread) {
      unsigned trk[kTopupPerThread];
      int uns[kTopupPerThread], local = 0;
#pragma unroll
      for (int i = 0; i < kTopupPerThread; ++i) {
        const long long q = q0 + (long long)tid * kTopupPerThread + i;
        trk[i] = 0;
        uns[i] = 0;
        if (q < e) {
          trk[i] = (unsigned)(keys[q] & 0xffffffffu);
          uns[i] = is_selected(trk[i]) ? 0 : 1;
        }
        local += uns[i];
      }
      int total = 0;
      int rank = block_scan(local, &total);
#pragma unroll
      for (int i = 0; i < kTopupPerThread; ++i) {
        if (uns[i]) {
          if (rank < needed) select(trk[i]);
          ++rank;
        }
      }
      needed -= min(needed, total);
    }
    __threadfence_block();
    __syncthreads();
  }
  (void)sh_total;
}

// selected flags / statistics in the caller's track order; counts the selection
__global__ __launch_bounds__(256) void select_finish_kernel(SelectView S, unsigned char* __restrict__ out_sel) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  int one = 0;
  if (p < S.Np_total) {
    one = S.sel[p] ? 1 : 0;
    out_sel[p] = (unsigned char)one;
  }
  for (int o = 32; o > 0; o >>= 1) one += __shfl_xor(one, o, 64);
  __shared__ int ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = one;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = ws[0] + ws[1] + ws[2] + ws[3];
    if (t) atomicAdd(&S.counters[1], t);
  }
}

__global__ __launch_bounds__(256) void scatter_track_stats_kernel(const int* __restrict__ pt_orig, int n_pad,
                                                                  const int* __restrict__ cnt,
                                                                  const double* __restrict__ mean, int long_thr,
                                                                  int* __restrict__ out_len,
                                                                  double* __restrict__ out_err) {
  const int lp = blockIdx.x * 256 + threadIdx.x;
  if (lp >= n_pad) return;
  const int p = pt_orig[lp];
  if (p < 0) return;
  if (out_len) out_len[p] = min(cnt[lp], long_thr);
  if (out_err) out_err[p] = mean[lp];
}

// outlier filter: flags / means in the caller's track order + the three counters
// counters: [0] estimated tracks examined, [1] bad reprojections, [2] insufficient angles
__global__ __launch_bounds__(256) void filter_finish_kernel(const int* __restrict__ pt_orig, int n_pad,
                                                            const unsigned char* __restrict__ flag,
                                                            const double* __restrict__ mean,
                                                            unsigned char* __restrict__ out_flag,
                                                            double* __restrict__ out_mean, int* __restrict__ counters) {
  const int lp = blockIdx.x * 256 + threadIdx.x;
  int n = 0, bad = 0, ang = 0;
  if (lp < n_pad) {
    const int p = pt_orig[lp];
    if (p >= 0) {
      const unsigned char f = flag[lp];
      n = 1;
      bad = f == 1;
      ang = f == 2;
      if (out_flag) out_flag[p] = f;
      if (out_mean) out_mean[p] = mean[lp];
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    n += __shfl_xor(n, o, 64);
    bad += __shfl_xor(bad, o, 64);
    ang += __shfl_xor(ang, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    if (n) atomicAdd(&counters[0], n);
    if (bad) atomicAdd(&counters[1], bad);
    if (ang) atomicAdd(&counters[2], ang);
  }
}

}  // namespace tmi
