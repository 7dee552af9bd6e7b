// Device-side construction of the static structure (what structure.cpp builds on host threads):
// the observation-sized work -- per-track ordering, the length-sorted SELL-64 layout, camera-major
// slots, the block set of S, the pair lists and the launch / work lists derived from them -- as radix
// sorts (hipCUB), scans and gather / scatter kernels in HBM.  Only O(#tracks) and O(#slices) data
// visits the host (track order -> slices, slice pointers).  Orders are the host builder's orders
// exactly (stable sorts keyed the way the host loops iterate), so a device-built structure gives
// bit-identical solves; tests/test_gpu_setup.py compares the two element for element.
//
// Scope: one rank, no shared intrinsics blocks, Nrb^2 <= 2^26 (a dense block-presence map).  Anything
// else uses structure.cpp (TMI_BA_HOST_SETUP=1 forces it).
//
// Round 1: 0.73 s of host time at Venice size (16 threads) out of a 0.8 s create; see DESIGN.md section 3.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>
#include <vector>

#include "device_view.h"
#include "structure.h"

namespace tmi {
namespace sg {

constexpr int kLongK = 32;  // tracks at least this long get a wavefront instead of a thread

// ---- kernels ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hist_kernel(const int* __restrict__ ocam, const int* __restrict__ opt,
                                                   long long n, int Nc, int Np, int* __restrict__ klen,
                                                   int* __restrict__ bad) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = ocam[i], p = opt[i];
  if (c < 0 || c >= Nc || p < 0 || p >= Np) {
    *bad = 1;
    return;
  }
  atomicAdd(&klen[p], 1);
}

// key (track, camera) of every observation; value = its index in the caller's arrays
__global__ __launch_bounds__(256) void obs_keys_kernel(const int* __restrict__ ocam, const int* __restrict__ opt,
                                                       long long n, int cam_bits, unsigned long long* __restrict__ keys,
                                                       unsigned* __restrict__ vals) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  keys[i] = ((unsigned long long)(unsigned)opt[i] << cam_bits) | (unsigned)ocam[i];
  vals[i] = (unsigned)i;
}

__global__ __launch_bounds__(256) void dup_check_kernel(const unsigned long long* __restrict__ keys, long long n,
                                                        int* __restrict__ bad) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i + 1 < n && keys[i] == keys[i + 1]) *bad = 2;
}

// tracks by descending length, then by their lowest view index, then by index (structure.cpp): tracks seen from
// the same views end up in the same slices, so a wave's parameter gathers and the zhat gathers of a view's slots
// touch neighbouring memory.  okeys = the observation keys sorted by (track, view).
__global__ __launch_bounds__(256) void track_keys_kernel(const int* __restrict__ klen, int Np,
                                                         const long long* __restrict__ tptr,
                                                         const unsigned long long* __restrict__ okeys, int cam_bits,
                                                         unsigned long long* __restrict__ keys,
                                                         int* __restrict__ vals) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= Np) return;
  unsigned cam = 0;
  if (klen[p] > 0) cam = (unsigned)(okeys[tptr[p]] & ((1ull << cam_bits) - 1ull));
  // ascending ~k = descending k; k = 0 sorts last
  keys[p] = ((unsigned long long)(~(unsigned)klen[p]) << cam_bits) | cam;
  vals[p] = p;
}

// SELL layout + the (lp, j)-ordered key list of the slot assignment.  Thread per local track.
__global__ __launch_bounds__(256) void layout_kernel(int Np_pad, const int* __restrict__ pt_orig,
                                                     const int* __restrict__ pt_k, const int* __restrict__ slice_ptr,
                                                     const long long* __restrict__ tptr,
                                                     const unsigned* __restrict__ tobs, const int* __restrict__ ocam,
                                                     const double* __restrict__ oxy, const int* __restrict__ cam_rb,
                                                     int Nrb, const long long* __restrict__ tstart,
                                                     int* __restrict__ obs_cam, double* __restrict__ obs_xy,
                                                     long long* __restrict__ obs_orig, unsigned* __restrict__ slot_key,
                                                     unsigned* __restrict__ slot_e) {
  const int lp = blockIdx.x * 256 + threadIdx.x;
  if (lp >= Np_pad) return;
  const int p = pt_orig[lp];
  if (p < 0) return;
  const int k = pt_k[lp];
  const int sl = lp >> 6, t = lp & 63;
  const long long src = tptr[p], dst = tstart[lp];
  for (int j = 0; j < k; ++j) {
    const unsigned i = tobs[src + j];
    const long long e = (long long)slice_ptr[sl] + (long long)j * 64 + t;
    const int c = ocam[i];
    obs_cam[e] = c;
    obs_xy[2 * e] = oxy[2 * (size_t)i];
    obs_xy[2 * e + 1] = oxy[2 * (size_t)i + 1];
    obs_orig[e] = (long long)i;
    const int rb = cam_rb[c];
    slot_key[dst + j] = rb >= 0 ? (unsigned)rb : (unsigned)Nrb;
    slot_e[dst + j] = (unsigned)e;
  }
}

__global__ __launch_bounds__(256) void slot_assign_kernel(const unsigned* __restrict__ keys,
                                                          const unsigned* __restrict__ es, long long n, int Nrb,
                                                          int* __restrict__ obs_cpos) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  if ((int)keys[q] < Nrb) obs_cpos[es[q]] = (int)q;
}

// out[i] = first index in sorted `keys` (n of them) whose key >= i, for i = 0..m
template <class K, class O>
__global__ __launch_bounds__(256) void lower_bound_kernel(const K* __restrict__ keys, long long n, long long m,
                                                          O* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i > m) return;
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if ((long long)keys[mid] < i) lo = mid + 1;
    else hi = mid;
  }
  out[i] = (O)lo;
}

// Work of one track for the pair passes: its observations that own a slot, in ascending block order
// (ascending camera order; cam_rb is monotone in the camera index).  L lanes share a track.
template <int L>
struct TrackLanes {
  int lp, lane;
  __device__ TrackLanes(int first_lp) {
    if (L == 1) {
      lp = first_lp + blockIdx.x * 256 + threadIdx.x;
      lane = 0;
    } else {
      lp = first_lp + blockIdx.x * 4 + (threadIdx.x >> 6);
      lane = threadIdx.x & 63;
    }
  }
};

// present[a * Nrb + b] = 1 for every pair of blocks a < b a non-constant track sees
template <int L>
__global__ __launch_bounds__(256) void block_flags_kernel(int lp0, int lp1, const int* __restrict__ pt_k,
                                                          const unsigned char* __restrict__ pt_const,
                                                          const int* __restrict__ slice_ptr,
                                                          const int* __restrict__ obs_cam, const int* __restrict__ cam_rb,
                                                          int Nrb, unsigned char* __restrict__ present) {
  const TrackLanes<L> T(lp0);
  if (T.lp >= lp1) return;
  const int k = pt_k[T.lp];
  if (k < 2 || pt_const[T.lp]) return;
  const long long base = (long long)slice_ptr[T.lp >> 6] + (T.lp & 63);
  for (int a = 0; a < k; ++a) {
    const int ra = cam_rb[obs_cam[base + (long long)a * 64]];
    if (ra < 0) continue;
    for (int b = a + 1 + T.lane; b < k; b += L) {
      const int rb = cam_rb[obs_cam[base + (long long)b * 64]];
      if (rb >= 0) present[(size_t)ra * Nrb + rb] = 1;
    }
  }
}

__global__ __launch_bounds__(256) void flags_to_int_kernel(const unsigned char* __restrict__ f, long long n,
                                                           int* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = f[i] ? 1 : 0;
}

__global__ __launch_bounds__(256) void block_list_kernel(const unsigned char* __restrict__ present,
                                                         const int* __restrict__ pos, int Nrb, int* __restrict__ ub_i,
                                                         int* __restrict__ ub_j, int* __restrict__ blk_id) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)Nrb * Nrb) return;
  int id = -1;
  if (present[i]) {
    id = pos[i];
    ub_i[id] = (int)(i / Nrb);
    ub_j[id] = (int)(i % Nrb);
  }
  blk_id[i] = id;
}

// pairs of a track: m = its observations with a slot; m (m - 1) / 2 of them (0 for constant tracks)
__global__ __launch_bounds__(256) void pair_count_kernel(int Np_pad, const int* __restrict__ pt_k,
                                                         const unsigned char* __restrict__ pt_const,
                                                         const int* __restrict__ slice_ptr,
                                                         const int* __restrict__ obs_cpos,
                                                         long long* __restrict__ cnt) {
  const int lp = blockIdx.x * 256 + threadIdx.x;
  if (lp >= Np_pad) return;
  const int k = pt_k[lp];
  long long m = 0;
  if (k >= 2 && !pt_const[lp]) {
    const long long base = (long long)slice_ptr[lp >> 6] + (lp & 63);
    for (int a = 0; a < k; ++a) m += obs_cpos[base + (long long)a * 64] >= 0 ? 1 : 0;
  }
  cnt[lp] = m * (m - 1) / 2;
}

// (block, slot_i | slot_j << 32) of every pair, at the track's offset, pairs in (a < b) lexicographic order
template <int L>
__global__ __launch_bounds__(256) void pair_emit_kernel(int lp0, int lp1, const int* __restrict__ pt_k,
                                                        const unsigned char* __restrict__ pt_const,
                                                        const int* __restrict__ slice_ptr,
                                                        const int* __restrict__ obs_cam, const int* __restrict__ cam_rb,
                                                        const int* __restrict__ obs_cpos, const int* __restrict__ blk_id,
                                                        int Nrb, const long long* __restrict__ pair_off,
                                                        unsigned* __restrict__ ukey, unsigned long long* __restrict__ pval) {
  const TrackLanes<L> T(lp0);
  if (T.lp >= lp1) return;
  const int k = pt_k[T.lp];
  if (k < 2 || pt_const[T.lp]) return;
  const long long base = (long long)slice_ptr[T.lp >> 6] + (T.lp & 63);
  long long out = pair_off[T.lp];
  // every observation of a non-constant track owns a slot unless its view has no block: walk the
  // slotted ones as a compact sequence 0..m-1 (index ia) without materialising it
  int m = 0;
  for (int a = 0; a < k; ++a) m += obs_cpos[base + (long long)a * 64] >= 0 ? 1 : 0;
  int ia = 0;
  for (int a = 0; a < k; ++a) {
    const long long ea = base + (long long)a * 64;
    const int sa = obs_cpos[ea];
    if (sa < 0) continue;
    const int ra = cam_rb[obs_cam[ea]];
    // pairs (ia, ib), ib > ia, start at  ia (2 m - ia - 1) / 2
    const long long row0 = out + (long long)ia * (2 * m - ia - 1) / 2;
    int ib = ia + 1;
    for (int b = a + 1; b < k; ++b) {
      const long long eb = base + (long long)b * 64;
      const int sb = obs_cpos[eb];
      if (sb < 0) continue;
      if (((ib - ia - 1) % L) == T.lane) {
        const int rb = cam_rb[obs_cam[eb]];
        const long long q = row0 + (ib - ia - 1);
        ukey[q] = (unsigned)blk_id[(size_t)ra * Nrb + rb];
        pval[q] = (unsigned long long)(unsigned)sa | ((unsigned long long)(unsigned)sb << 32);
      }
      ++ib;
    }
    ++ia;
  }
}

// pair_emit for long tracks, one wavefront per track: the slotted observations (slot, block) are
// compacted into LDS first, then row ia of the pair triangle is written 64 pairs at a time.  (The
// generic kernel above walks the whole a x b triangle with one global load per step -- 36 ms for the
// few hundred-view tracks of a Venice-sized problem, against 0.2 ms here.)
constexpr int kEmitCap = 2048;  // slotted observations of a track held in LDS; longer tracks take the generic kernel
__global__ __launch_bounds__(256) void pair_emit_long_kernel(int lp0, int lp1, const int* __restrict__ pt_k,
                                                             const unsigned char* __restrict__ pt_const,
                                                             const int* __restrict__ slice_ptr,
                                                             const int* __restrict__ obs_cam, const int* __restrict__ cam_rb,
                                                             const int* __restrict__ obs_cpos, const int* __restrict__ blk_id,
                                                             int Nrb, const long long* __restrict__ pair_off,
                                                             unsigned* __restrict__ ukey, unsigned long long* __restrict__ pval) {
  __shared__ int s_slot[4][kEmitCap];
  __shared__ int s_blk[4][kEmitCap];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lp = lp0 + blockIdx.x * 4 + w;
  if (lp >= lp1) return;
  const int k = pt_k[lp];
  if (k < 2 || k > kEmitCap || pt_const[lp]) return;
  const long long base = (long long)slice_ptr[lp >> 6] + (lp & 63);
  int m = 0;
  for (int a0 = 0; a0 < k; a0 += 64) {
    const int a = a0 + lane;
    int sa = -1, ra = -1;
    if (a < k) {
      const long long ea = base + (long long)a * 64;
      sa = obs_cpos[ea];
      if (sa >= 0) ra = cam_rb[obs_cam[ea]];
    }
    const unsigned long long mask = __ballot(sa >= 0);
    if (sa >= 0) {
      const int pos = m + __popcll(mask & ((1ull << lane) - 1ull));
      s_slot[w][pos] = sa;
      s_blk[w][pos] = ra;
    }
    m += __popcll(mask);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const long long out = pair_off[lp];
  for (int ia = 0; ia + 1 < m; ++ia) {
    const int sa = s_slot[w][ia], ra = s_blk[w][ia];
    const long long row0 = out + (long long)ia * (2 * m - ia - 1) / 2;
    for (int ib = ia + 1 + lane; ib < m; ib += 64) {
      const long long q = row0 + (ib - ia - 1);
      ukey[q] = (unsigned)blk_id[(size_t)ra * Nrb + s_blk[w][ib]];
      pval[q] = (unsigned long long)(unsigned)sa | ((unsigned long long)(unsigned)s_slot[w][ib] << 32);
    }
  }
}

__global__ __launch_bounds__(256) void split_pairs_kernel(const unsigned long long* __restrict__ pval, long long n,
                                                          int* __restrict__ pi, int* __restrict__ pj) {
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= n) return;
  pi[q] = (int)(unsigned)(pval[q] & 0xffffffffull);
  pj[q] = (int)(unsigned)(pval[q] >> 32);
}

__global__ __launch_bounds__(256) void widen_kernel(const int* __restrict__ in, int n, long long* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i];
}

// qstart[x] = first index of queue x in the launch-order keys ((row & 7) << 24 | row), x = 0..8
__global__ void queue_start_kernel(const unsigned* __restrict__ keys, long long n, long long* __restrict__ qstart) {
  const int x = threadIdx.x;
  if (x > 8) return;
  const unsigned long long want = (unsigned long long)x << 24;
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if ((unsigned long long)keys[mid] < want) lo = mid + 1;
    else hi = mid;
  }
  qstart[x] = lo;
}

__global__ __launch_bounds__(256) void iota_kernel(int* __restrict__ v, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) v[i] = i;
}

// launch order keys: first by pair count descending (stable), then by (row & 7, row)
__global__ __launch_bounds__(256) void order_key1_kernel(const long long* __restrict__ pair_ptr, int nub,
                                                         unsigned* __restrict__ key) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u < nub) key[u] = ~(unsigned)(pair_ptr[u + 1] - pair_ptr[u]);
}
__global__ __launch_bounds__(256) void order_key2_kernel(const int* __restrict__ u_sorted, const int* __restrict__ ub_i,
                                                         int nub, unsigned* __restrict__ key) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nub) return;
  const unsigned row = (unsigned)ub_i[u_sorted[q]];
  key[q] = ((row & 7u) << 24) | row;
}
// hdr[slot] = {block, #pairs, first pair lo, first pair hi}; queue x starts at qstart[x]
__global__ __launch_bounds__(256) void order_fill_kernel(const int* __restrict__ u_sorted,
                                                         const unsigned* __restrict__ key2, int nub,
                                                         const long long* __restrict__ qstart,
                                                         const long long* __restrict__ pair_ptr, int4* __restrict__ hdr) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nub) return;
  const int x = (int)(key2[q] >> 24);
  const long long idx = q - qstart[x];
  const long long slot = ((idx / 16) * 8 + x) * 16 + (idx % 16);
  const int u = u_sorted[q];
  const long long p0 = pair_ptr[u];
  hdr[slot] = make_int4(u, (int)(pair_ptr[u + 1] - p0), (int)(unsigned)(p0 & 0xffffffffLL), (int)(p0 >> 32));
}

// SpMV chunk list of a block row
__global__ __launch_bounds__(256) void spc_count_kernel(const int* __restrict__ urow_ptr, int Nrb, int chunk,
                                                        int* __restrict__ cnt) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < Nrb) cnt[r] = (urow_ptr[r + 1] - urow_ptr[r] + chunk - 1) / chunk;
}
__global__ __launch_bounds__(256) void spc_fill_kernel(const int* __restrict__ urow_ptr, const int* __restrict__ rptr,
                                                       int Nrb, int chunk, int* __restrict__ spc_row,
                                                       int* __restrict__ spc_u0) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= Nrb) return;
  int c = rptr[r];
  for (int u = urow_ptr[r]; u < urow_ptr[r + 1]; u += chunk, ++c) {
    spc_row[c] = r;
    spc_u0[c] = u;
  }
}

}  // namespace sg
// view-major observation index for the inner iterations: keys (camera, Nc for padding) and
// values (element, track) of every element of the SELL layout
__global__ __launch_bounds__(256) void view_index_keys_kernel(int Np_pad, const int* __restrict__ pt_k,
                                                              const int* __restrict__ slice_ptr,
                                                              const int* __restrict__ obs_cam, int Nc,
                                                              unsigned* __restrict__ key, int* __restrict__ elem_lp) {
  const int lp = blockIdx.x * 256 + threadIdx.x;
  if (lp >= Np_pad) return;
  const int sl = lp >> 6;
  const int K = (slice_ptr[sl + 1] - slice_ptr[sl]) >> 6;
  const long long base = (long long)slice_ptr[sl] + (lp & 63);
  const int k = pt_k[lp];
  for (int j = 0; j < K; ++j) {
    const long long e = base + (long long)j * 64;
    const int cam = j < k ? obs_cam[e] : -1;
    key[e] = cam >= 0 ? (unsigned)cam : (unsigned)Nc;
    elem_lp[e] = lp;
  }
}

__global__ __launch_bounds__(256) void gather_int_kernel(const int* __restrict__ idx, long long n,
                                                         const int* __restrict__ src, int* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
// 16-byte records (the pixel of a layout element)
__global__ __launch_bounds__(256) void gather_double2_kernel(const int* __restrict__ idx, long long n,
                                                             const double2* __restrict__ src, double2* __restrict__ dst) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

}  // namespace tmi
