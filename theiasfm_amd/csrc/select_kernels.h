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
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * kSlicesPerBlock + (threadIdx.x >> 6);
  if (s >= v.nslices) return;
  const int lp = s * 64 + lane;
  const int k = v.pt_k[lp];
  const size_t base = (size_t)v.slice_ptr[s] + lane;
  for (int j = 0; j < k; ++j) {
    const size_t e = base + (size_t)j * 64;
    const int cam = v.obs_cam[e];
    if (S.view_mask && !S.view_mask[cam]) continue;
    int cx, cy;
    feature_cell(v, S, e, &cx, &cy);
    atomicMin(&S.vbox[4 * cam + 0], cx);
    atomicMax(&S.vbox[4 * cam + 1], cx);
    atomicMin(&S.vbox[4 * cam + 2], cy);
    atomicMax(&S.vbox[4 * cam + 3], cy);
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
  const int lane = threadIdx.x & 63;
  const int s = blockIdx.x * kSlicesPerBlock + (threadIdx.x >> 6);
  if (s >= v.nslices) return;
  const int lp = s * 64 + lane;
  const int k = v.pt_k[lp];
  if (k == 0) return;
  const int p = S.pt_orig[lp];
  const unsigned tlen = (unsigned)min(S.cnt[lp], S.long_thr);
  const unsigned long long ebits = (unsigned long long)__double_as_longlong(S.mean[lp]);
  const size_t base = (size_t)v.slice_ptr[s] + lane;
  for (int j = 0; j < k; ++j) {
    const size_t e = base + (size_t)j * 64;
    const int cam = v.obs_cam[e];
    if (S.view_mask && !S.view_mask[cam]) continue;
    int cx, cy;
    feature_cell(v, S, e, &cx, &cy);
    const long long w = (long long)S.vbox[4 * cam + 1] - S.vbox[4 * cam + 0] + 1;
    const long long slot = S.cell_off[cam] + (long long)(cy - S.vbox[4 * cam + 2]) * w + (cx - S.vbox[4 * cam + 0]);
    if (PASS == 1) {
      atomicMin(&S.cell_len[slot], tlen);
    } else if (PASS == 2) {
      if (S.cell_len[slot] == tlen) atomicMin(&S.cell_err[slot], ebits);
    } else {
      if (S.cell_len[slot] == tlen && S.cell_err[slot] == ebits) atomicMin(&S.cell_trk[slot], (unsigned)p);
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
constexpr int kTopupPerThread = 8;  // views with more than 8192 tracks loop in chunks

__global__ __launch_bounds__(kTopupThreads) void select_topup_kernel(SelectView S,
                                                                     const unsigned long long* __restrict__ keys,
                                                                     const long long* __restrict__ vt_ptr,
                                                                     int min_opt) {
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
  for (int c = 0; c < S.Nc; ++c) {
    if (S.view_mask && !S.view_mask[c]) continue;
    const long long b = vt_ptr[c], e = vt_ptr[c + 1];
    const int n = (int)(e - b);
    if (n == 0) continue;
    // count the selected tracks of the view
    int mine = 0;
    for (long long q = b + tid; q < e; q += kTopupThreads) mine += (int)S.sel[(unsigned)(keys[q] & 0xffffffffu)];
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
          uns[i] = S.sel[trk[i]] ? 0 : 1;
        }
        local += uns[i];
      }
      int total = 0;
      int rank = block_scan(local, &total);
#pragma unroll
      for (int i = 0; i < kTopupPerThread; ++i) {
        if (uns[i]) {
          if (rank < needed) S.sel[trk[i]] = 1u;
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
  if ((threadIdx.x & 63) == 0 && one) atomicAdd(&S.counters[1], one);
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
