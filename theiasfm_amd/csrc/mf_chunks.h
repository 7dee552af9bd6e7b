// The matrix-free Schur product q = S p in ONE sweep over the track-major planes (kernel class 5, schur_mode
// implicit / auto; no shared intrinsics blocks).  Replaces Ceres' ImplicitSchurComplement::RightMultiply behind
// ceres::Solve at src/theia/sfm/bundle_adjustment/bundle_adjuster.cc:205 (ITERATIVE_SCHUR).
//
//   S p = D_c p + sum_obs A_i^T t_i,   t_i = u_i - Jp_i z_track,   u_i = A_i p_cam(i),
//   z = (V + D_p)^-1 sum_{i in track} Jp_i^T u_i = L^-T L^-1 w.
//
// Rounds 1-3 ran this as two launches -- a track-major pass over the planes (u, zhat) and a camera-major pass over
// the [A | Q] records (A^T t summed per view) -- so the camera Jacobian blocks crossed HBM twice per product
// (2.4 GB for 0.83 GB algorithmic at Venice size).  Summing A^T t per VIEW is what forced the second order.  Here
// a workgroup streams a UNIT (below) ONCE: the lane that owns an observation keeps its A block in registers until
// the track's z is known, forms v_i = A_i^T t_i and drops it into LDS at the observation's position in the
// unit's VIEW order (a static index); one thread per "run" (the observations of the unit that share a view)
// then sums consecutive LDS entries => a fixed order and no atomics, and adds the run sum into per-view
// accumulators that stay in LDS while the workgroup walks an ITEM = a range of consecutive units.  Tracks are
// ordered by (length, lowest view), so the units of an item see the same few hundred views; an item leaves one
// D-vector per distinct view ("slot") in HBM and reduce_kernel sums a view's slots in item order and adds the damping
// and p.q.  Bytes per observation: the stored columns of the A planes and the Jp planes once (16 (D + DP), 48 less
// where the position columns are formed from Jp: DeviceView::drop_pos), two int32 indices, and ~10 % of partials,
// against twice the A blocks + a 128-byte line per zhat gather before.
//
// Units: what a workgroup (kWaves wavefronts x reg_rows(D) row slots) handles between two load batches --
//   * a SELL slice of at most kWaves reg_rows(D) rows, or a pack of 2-4 consecutive slices of <= 4 rows;
//   * one of the L = 2, 4, .. 64 pieces a longer slice (up to 64 times that many rows) is cut into: 64 / L tracks,
//     L lanes per track;
//   * four tracks of a slice that is longer still: a wavefront per track, u / t through a global scratch, every run
//     writes its slot directly (an item of its own).
#pragma once
#include <hip/hip_runtime.h>

#include "device_view.h"
#include "kernels.h"

namespace tmi {
#ifndef TMI_MF_ABL
#define TMI_MF_ABL 0  // timing experiments, results wrong by construction (bit 0: no run sums, 1: x gathers from eight cached
                      // blocks, 2: the v_i not stored to LDS, 3: no slot write-out); profiles/r06_product_experiments.md
#endif
namespace mfc {

constexpr int kWaves = 4;  // two workgroups per CU (<= 80 KB of LDS each), two wavefronts per SIMD
constexpr int kThreads = 64 * kWaves;
constexpr int kLongRun = 12;                  // a run with more entries than this is summed by a whole wavefront
// row slots of a workgroup: kWaves * reg_rows(D); a row slot keeps A, Jp and u of its rows in registers until the
// track's z is known (see the narrow path of product_kernel for what a unit puts into them)
__host__ __device__ constexpr int reg_rows(int D) { return D <= 9 ? 2 : 1; }
// distinct views an item of SEVERAL slices may see (its accumulators); an item of one slice writes its run sums
// straight to its slots and has no limit
#ifndef TMI_LCM9
#define TMI_LCM9 472  // (A/B builds: -DTMI_LCM9=<views>)
#endif
__host__ __device__ constexpr int lc_max(int D) { return D <= 6 ? 704 : D <= 9 ? TMI_LCM9 : D <= 12 ? 352 : 264; }
// v_i per round: at least the kWaves * reg_rows(D) rows a pack keeps in registers
__host__ __device__ constexpr int vb_entries(int D) { return D <= 9 ? 512 : 256; }

struct View {
  int n_items, n_units, n_runs;
  int nub;  // units [0, nub): four tracks (a wavefront each) of a slice too long for the row slots
  int nwb;  // = nub since round 4 (was: + the quarters of the 16-lane slices); [nwb, n_units): narrow units
  const int* item_unit0;     // [n_items + 1] units of an item (wide / ultra: one)
  const int* item_order;     // [n_items] workgroup b takes item item_order[b]: largest first, so the last round is short
  const int4* item_hdr;      // [n_items][6] what a narrow item's prologue needs, in one place (item_hdr_kernel)
  const int4* unit_desc;     // [n_units - nwb] narrow units: {first element, rows, rows of one slice | log2 L << 16,
                             //  first slice}; L > 1: one of the L pieces (64 / L tracks each) of a long slice
  const int* unit_run_ptr;   // [n_units + 1]
  const int* run_obs_ptr;    // [n_runs + 1]
  const int* run_obs;        // element index e of every observation that has a view block, by (unit, view)
  const int* run_slot;       // [n_runs] slot the run's sum goes to
  const int* item_slot_ptr;  // [n_items + 1]
  const int* slot_rb;        // [n_slots] view block of a slot
  const int* obs_pos;        // [No_pad] position of the observation in its unit's view order, -1: no view block
  const int* cam_slot_ptr;   // [Nrb + 1] slots of a view block ...
  const int* cam_slots;      // ... ascending (item order)
  long long* prof;           // TMI_MF_PROFILE builds: per item {cycles of 10 phases, units, start-up, write-out}
  double* partial;           // [n_slots][D]
  double* ut;                // [elements of the wavefront-per-track slices][2]
  const int* guard;          // null, or DeviceView::pcg_done: a product enqueued ahead of PCG's stopping test returns at once
                             // when the test held (engine.hip, solve_reduced_pcg)
};

struct UnitShape {
  int s, t0, nt, K, sp0;
};
__device__ __forceinline__ UnitShape unit_shape(const DeviceView& v, int u, int nub, int nwb) {
  UnitShape q;
  if (u < nub) {
    q.s = u >> 4;
    q.t0 = 4 * (u & 15);
    q.nt = 4;
  } else if (u < nwb) {
    const int b = u - nub;
    q.s = v.n_ultra + (b >> 2);
    q.t0 = 16 * (b & 3);
    q.nt = 16;
  } else {
    q.s = v.n_wide + (u - nwb);
    q.t0 = 0;
    q.nt = 64;
  }
  q.sp0 = v.slice_ptr[q.s];
  q.K = (v.slice_ptr[q.s + 1] - q.sp0) >> 6;
  return q;
}
// (narrow units are slices or packs of slices: their shape comes from View::unit_desc)

// ---- structure build (once per handle; radix sorts + scans driven by engine.hip) ---------------------------
// key (unit, view block) of every element; elements without a view block sort last
__global__ __launch_bounds__(256) void unit_keys_kernel(DeviceView v, int nub, int nwb, int Nrb,
                                                        const int4* __restrict__ unit_desc, unsigned long long invalid,
                                                        unsigned long long* __restrict__ keys, int* __restrict__ vals) {
  const int u = blockIdx.x;
  UnitShape q;
  if (u < nwb) {
    q = unit_shape(v, u, nub, nwb);
  } else {  // a slice or a pack of slices: rows x 64 consecutive elements
    const int4 d = unit_desc[u - nwb];
    q.sp0 = d.x;  // (the first track's column included)
    q.K = d.y;
    q.t0 = 0;
    q.nt = 64 >> (d.z >> 16);
  }
  const int n = q.K * q.nt;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int j = i / q.nt, t = q.t0 + (i - j * q.nt);
    const int e = q.sp0 + 64 * j + t;
    const int rb = v.obs_rb[e];
    keys[e] = rb >= 0 ? (unsigned long long)u * (unsigned)Nrb + (unsigned)rb : invalid;
    vals[e] = e;
  }
}

// head[i] = 1 where a new key starts among the valid keys (head[n] = 0 closes the scan)
__global__ __launch_bounds__(256) void heads_kernel(const unsigned long long* __restrict__ skey, long long n,
                                                    unsigned long long invalid, int* __restrict__ head,
                                                    int* __restrict__ n_valid) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    head[i] = 0;
    return;
  }
  const unsigned long long k = skey[i];
  const bool valid = k != invalid;
  head[i] = (valid && (i == 0 || skey[i - 1] != k)) ? 1 : 0;
  if (!valid && (i == 0 || skey[i - 1] != invalid)) *n_valid = (int)i;
}

__global__ __launch_bounds__(256) void run_fill_kernel(const unsigned long long* __restrict__ skey,
                                                       const int* __restrict__ head, const int* __restrict__ pos,
                                                       long long n, int* __restrict__ run_first,
                                                       unsigned long long* __restrict__ run_key) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !head[i]) return;
  run_first[pos[i]] = (int)i;
  run_key[pos[i]] = skey[i];
}

// out[i] = first index of the sorted keys whose key >= i * stride, i = 0..m
__global__ __launch_bounds__(256) void lower_bound_stride_kernel(const unsigned long long* __restrict__ keys,
                                                                 long long n, unsigned long long stride, int m,
                                                                 int* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > m) return;
  const unsigned long long want = (unsigned long long)i * stride;
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (keys[mid] < want) lo = mid + 1;
    else hi = mid;
  }
  out[i] = (int)lo;
}

// (item, view block) of every run; value = the run
__global__ __launch_bounds__(256) void slot_keys_kernel(const unsigned long long* __restrict__ run_key, int n_runs,
                                                        const int* __restrict__ unit_item, int Nrb,
                                                        unsigned long long* __restrict__ keys, int* __restrict__ vals) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_runs) return;
  const unsigned long long k = run_key[r];
  const unsigned long long u = k / (unsigned)Nrb;
  keys[r] = (unsigned long long)unit_item[u] * (unsigned)Nrb + (k - u * (unsigned)Nrb);
  vals[r] = r;
}

__global__ __launch_bounds__(256) void slot_fill_kernel(const unsigned long long* __restrict__ skey,
                                                        const int* __restrict__ head, const int* __restrict__ pos,
                                                        const int* __restrict__ srun, int n_runs, int Nrb,
                                                        int* __restrict__ run_slot, int* __restrict__ slot_rb,
                                                        unsigned long long* __restrict__ slot_key) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_runs) return;
  const int slot = pos[i] + head[i] - 1;
  run_slot[srun[i]] = slot;
  if (head[i]) {
    slot_rb[slot] = (int)(skey[i] % (unsigned)Nrb);
    slot_key[slot] = skey[i];
  }
}

// position of every observation with a view block in the view order of its unit (= its index in run_obs
// minus the unit's first)
__global__ __launch_bounds__(256) void obs_pos_kernel(const unsigned long long* __restrict__ skey,
                                                      const int* __restrict__ sval, int n_valid, int Nrb,
                                                      const int* __restrict__ unit_run_ptr,
                                                      const int* __restrict__ run_first, int* __restrict__ obs_pos) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_valid) return;
  const unsigned long long u = skey[i] / (unsigned)Nrb;
  obs_pos[sval[i]] = i - run_first[unit_run_ptr[u]];
}

__global__ __launch_bounds__(256) void slot_rb_keys_kernel(const int* __restrict__ slot_rb, int n,
                                                           unsigned* __restrict__ keys, int* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  keys[i] = (unsigned)slot_rb[i];
  vals[i] = i;
}

__global__ __launch_bounds__(256) void lower_bound_u32_kernel(const unsigned* __restrict__ keys, int n, int m,
                                                              int* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i > m) return;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < (unsigned)i) lo = mid + 1;
    else hi = mid;
  }
  out[i] = lo;
}

// everything the prologue of a narrow item looks up, gathered once: {u0, u1, slot0, #slots}, the descriptors of its
// first three units, the run ranges of the first three and the observation ranges of the first two -- one
// dependent round trip at the start of a work item instead of three
__global__ __launch_bounds__(256) void item_hdr_kernel(int n_items, int nwb, int n_units, const int* __restrict__ item_unit0,
                                                       const int* __restrict__ item_slot_ptr,
                                                       const int4* __restrict__ unit_desc,
                                                       const int* __restrict__ unit_run_ptr,
                                                       const int* __restrict__ run_obs_ptr, int4* __restrict__ hdr) {
  const int item = blockIdx.x * 256 + threadIdx.x;
  if (item >= n_items) return;
  int4* h = hdr + (size_t)item * 6;
  const int u0 = item_unit0[item], u1 = item_unit0[item + 1];
  const int slot0 = item_slot_ptr[item];
  h[0] = make_int4(u0, u1, slot0, item_slot_ptr[item + 1] - slot0);
  if (item < nwb) return;
  const int4* desc = unit_desc - nwb;
  const int last_u = n_units - 1;
  h[1] = desc[u0];
  h[2] = desc[min(u0 + 1, last_u)];
  h[3] = desc[min(u0 + 2, last_u)];
  const int r0 = unit_run_ptr[u0], r1 = unit_run_ptr[u0 + 1], r2 = unit_run_ptr[min(u0 + 2, n_units)];
  h[4] = make_int4(r0, r1, r2, run_obs_ptr[r0]);
  h[5] = make_int4(run_obs_ptr[r1], 0, 0, 0);
}

// ---- the product ---------------------------------------------------------------------------------------------
// z = L^-T (L^-1 w) of track lp (Linv planes: sym_idx(a, b), a <= b, holds L^-1(b, a))
template <int DP>
__device__ __forceinline__ void track_solve(const DeviceView& v, int lp, const double (&w)[DP], double (&z)[DP]) {
  const size_t NP = (size_t)v.Np_pad;
  double Li[sym_size(DP)];
#pragma unroll
  for (int i = 0; i < sym_size(DP); ++i) Li[i] = v.Linv[(size_t)i * NP + lp];
  double zh[DP];
#pragma unroll
  for (int b = 0; b < DP; ++b) {
    double t = 0.0;
#pragma unroll
    for (int a = 0; a <= b; ++a) t += Li[sym_idx(a, b, DP)] * w[a];
    zh[b] = t;
  }
#pragma unroll
  for (int a = 0; a < DP; ++a) {
    double t = 0.0;
#pragma unroll
    for (int b = a; b < DP; ++b) t += Li[sym_idx(a, b, DP)] * zh[b];
    z[a] = t;
  }
}

// workgroup barrier that orders LDS accesses only: loads from global memory stay in flight across it (__syncthreads
// drains them: s_waitcnt vmcnt(0)).  Everything the wavefronts of the narrow path exchange goes through LDS.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The rows of the planes, which a product touches ONCE, are loaded non-temporal:
// they would otherwise push the x blocks -- gathered again and again by the units of an item, the only reuse the
// kernel has -- out of the 32 KB vector L1 (measured: 5 % of the kernel; the same hint on the index, L^-1 and run-list
// loads, which are issued a unit ahead, costs 3 %).
typedef double mf_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 mf_stream_load(const double2* p) {
  const mf_v2d t = __builtin_nontemporal_load(reinterpret_cast<const mf_v2d*>(p));
  return make_double2(t.x, t.y);
}
#define MF_STREAM_LOAD(p) mf_stream_load(p)
// DROP: the position columns of the A planes are not stored (DeviceView::drop_pos); x is then the vector with the position
// entries of every block already times the block's column scales (pos_scale_kernel), reduce_kernel applies the scales
// to the position entries of the sums.
// CP: compact planes (device_view.h, DeviceView::compact): pm_A holds p_n alone, x is the transformed vector
// [kappa | eta | a0 a1 a2] of every view (compact_forward), the sums are the moments compact_backward maps back
// (reduce_kernel).  Same units, runs, slots and summation orders as the full planes.
// (CP == 2: a robust loss -- pm_A holds [C p_n | r^2 .] per observation, 32 bytes)
template <int D, int DP, bool DROP, int CP = 0>
__global__ __launch_bounds__(kThreads, 2) void product_kernel(DeviceView v, View m, const double* __restrict__ x) {
  static_assert(!CP || (DROP && D == 9), "compact planes: the 9-wide PINHOLE block without stored position columns");
  constexpr size_t CPT = CP == 2 ? 256 : 128;  // doubles of a 64-observation tile of the compact pm_A
  constexpr int A0 = DROP ? 3 : 0;  // first stored column of the A planes
  constexpr int LCM = lc_max(D);
  constexpr int VB = vb_entries(D);
  constexpr int ROWD = 2 * D * 64;  // doubles of one row (= one 64-observation tile) of the A planes
  constexpr int ROWP = 2 * DP * 64;
  // vbuf: the v_i of a round, in view order; the row slots' shares of w (wpart) live in the same memory: they are
  // read before the z barrier, vbuf is written after it
  __shared__ double vbuf[VB * D];
  static_assert(VB * D >= reg_rows(D) * kWaves * DP * 64, "wpart fits vbuf");
  double (*wpart)[DP][64] = reinterpret_cast<double (*)[DP][64]>(vbuf);
  __shared__ double zs[kWaves][DP][64];
  __shared__ double acc[LCM * D];
  if (m.guard && *m.guard) return;
  const int item = m.item_order[blockIdx.x];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;

  if (item < m.nwb) {
    // ---- a wide / ultra unit: 16 or 64 lanes per track, u and t through the global scratch
    const UnitShape q = unit_shape(v, item, m.nub, m.nwb);
    {
      const bool ultra = item < m.nub;
      const int L = ultra ? 64 : kWideLanes;
      const int t = ultra ? q.t0 + w : q.t0 + 4 * w + (lane >> 4);
      const int j0 = ultra ? lane : (lane & (kWideLanes - 1));
      const int lp = q.s * 64 + t;
      const int k = v.pt_k[lp];
      const size_t base = (size_t)q.sp0 + t;
      double wv[DP];
#pragma unroll
      for (int a = 0; a < DP; ++a) wv[a] = 0.0;
      double pcw[3] = {0.0, 0.0, 0.0};
      if (DROP && !CP) {
#pragma unroll
        for (int a = 0; a < 3; ++a) pcw[a] = v.pos_coef[(size_t)a * v.Np_pad + lp];
      }
      double Xw[4] = {0.0, 0.0, 0.0, 0.0}, ispw[3] = {0.0, 0.0, 0.0};
      if (CP) {
#pragma unroll
        for (int a = 0; a < 4; ++a) Xw[a] = v.cp_trk[(size_t)a * v.Np_pad + lp];
#pragma unroll
        for (int a = 0; a < 3; ++a) ispw[a] = v.cp_trk[(size_t)(4 + a) * v.Np_pad + lp];
      }
      for (int j = j0; j < k; j += L) {
        const size_t e = base + (size_t)j * 64;
        const int rb = v.obs_rb[e];
        if (rb < 0) continue;
        const double* xc = x + (size_t)rb * D;
        const double* ap = v.pm_A + (e >> 6) * (size_t)ROWD + ((e & 63) << 1);
        const double* jp = v.pm_Jp + (e >> 6) * (size_t)ROWP + ((e & 63) << 1);
        double u0 = 0.0, u1 = 0.0;
        if (!CP) {
#pragma unroll
          for (int a = A0; a < D; ++a) {
            const double2 aa = *reinterpret_cast<const double2*>(ap + a * 128);
            const double xa = xc[a];
            u0 += aa.x * xa;
            u1 += aa.y * xa;
          }
        }
        double2 jj[DP];
#pragma unroll
        for (int a = 0; a < DP; ++a) jj[a] = *reinterpret_cast<const double2*>(jp + a * 128);
        if constexpr (CP) {
          double g[9];
#pragma unroll
          for (int a = 0; a < 9; ++a) g[a] = xc[a];
          const double* cpa = v.pm_A + (e >> 6) * CPT + ((e & 63) << 1);
          const double2 pn = *reinterpret_cast<const double2*>(cpa);
          const double r2 = CP == 2 ? cpa[128] : -1.0;
          const double2 j3[3] = {jj[0], jj[1], jj[2]};
          compact_ax(g, Xw, ispw, j3, pn, u0, u1, r2);
        } else if (DROP) {
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const double ca = pcw[a] * xc[a];
            u0 += jj[a < DP ? a : 0].x * ca;
            u1 += jj[a < DP ? a : 0].y * ca;
          }
        }
#pragma unroll
        for (int a = 0; a < DP; ++a) wv[a] += jj[a].x * u0 + jj[a].y * u1;
        *reinterpret_cast<double2*>(m.ut + 2 * e) = make_double2(u0, u1);
      }
#pragma unroll
      for (int a = 0; a < DP; ++a) wv[a] = group_sum(wv[a], ultra ? 2 : 1);
      if (k > 0) {
        double z[DP];
        track_solve<DP>(v, lp, wv, z);
        for (int j = j0; j < k; j += L) {
          const size_t e = base + (size_t)j * 64;
          if (v.obs_rb[e] < 0) continue;
          const double* jp = v.pm_Jp + (e >> 6) * (size_t)ROWP + ((e & 63) << 1);
          double2 ut = *reinterpret_cast<const double2*>(m.ut + 2 * e);
#pragma unroll
          for (int a = 0; a < DP; ++a) {
            const double2 jj = *reinterpret_cast<const double2*>(jp + a * 128);
            ut.x -= jj.x * z[a];
            ut.y -= jj.y * z[a];
          }
          *reinterpret_cast<double2*>(m.ut + 2 * e) = ut;
        }
      }
    }
    __syncthreads();  // the t values of this workgroup's tracks are in the scratch
    const int r1 = m.unit_run_ptr[item + 1];
    for (int r = m.unit_run_ptr[item] + (int)threadIdx.x; r < r1; r += kThreads) {
      double sum[D];
#pragma unroll
      for (int a = 0; a < D; ++a) sum[a] = 0.0;
      const int o1 = m.run_obs_ptr[r + 1];
      for (int o = m.run_obs_ptr[r]; o < o1; ++o) {
        const size_t e = (size_t)m.run_obs[o];
        const double2 t = *reinterpret_cast<const double2*>(m.ut + 2 * e);
        const double* ap = v.pm_A + (e >> 6) * (size_t)ROWD + ((e & 63) << 1);
        if (!CP) {
#pragma unroll
          for (int a = A0; a < D; ++a) {
            const double2 aa = *reinterpret_cast<const double2*>(ap + a * 128);
            sum[a] += aa.x * t.x + aa.y * t.y;
          }
        }
        if constexpr (CP) {
          const double* jp = v.pm_Jp + (e >> 6) * (size_t)ROWP + ((e & 63) << 1);
          const size_t lpe = (size_t)q.s * 64 + (e & 63);
          double Xe[4], ispe[3];
#pragma unroll
          for (int a = 0; a < 4; ++a) Xe[a] = v.cp_trk[(size_t)a * v.Np_pad + lpe];
#pragma unroll
          for (int a = 0; a < 3; ++a) ispe[a] = v.cp_trk[(size_t)(4 + a) * v.Np_pad + lpe];
          const double2 j3[3] = {*reinterpret_cast<const double2*>(jp), *reinterpret_cast<const double2*>(jp + 128),
                                 *reinterpret_cast<const double2*>(jp + 256)};
          const double* cpa = v.pm_A + (e >> 6) * CPT + ((e & 63) << 1);
          const double2 pn = *reinterpret_cast<const double2*>(cpa);
          const double r2 = CP == 2 ? cpa[128] : -1.0;
          double o[9];
          compact_at(Xe, ispe, j3, pn, t.x, t.y, o, r2);
#pragma unroll
          for (int a = 0; a < 9; ++a) sum[a] += o[a];
        } else if (DROP) {
          // the element's track: slice q.s, column e & 63
          const double* jp = v.pm_Jp + (e >> 6) * (size_t)ROWP + ((e & 63) << 1);
          const size_t lpe = (size_t)q.s * 64 + (e & 63);
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const double2 jj = *reinterpret_cast<const double2*>(jp + (a < DP ? a : 0) * 128);
            sum[a] += v.pos_coef[(size_t)a * v.Np_pad + lpe] * (jj.x * t.x + jj.y * t.y);
          }
        }
      }
      double* dst = m.partial + (size_t)m.run_slot[r] * D;
#pragma unroll
      for (int a = 0; a < D; ++a) dst[a] = sum[a];
    }
    return;
  }

  // ---- an item of narrow units
  // A narrow unit fills the kWaves * reg_rows(D) ROW SLOTS of the workgroup: row slot (w, rr) belongs to wavefront w.
  //   * a slice of at most that many rows, or a PACK of 2-4 consecutive slices of <= 4 rows (59 % of the slices of the
  //     bench problem): their tiles are consecutive in memory, so a pack is walked like one slice of G K rows whose
  //     row R belongs to the tracks of slice R / K; lane = track, a row slot holds one row;
  //   * a longer slice (up to 64 times the row slots) is cut into L = 2, 4, .. 64 units of 64 / L tracks: L lanes per track,
  //     lane = sub * (64 / L) + track, a row slot holds the L rows slot * L + sub.  (Before: the rows beyond the row
  //     slots were demand loads in dependent trips, twice per unit -- 29 % of the observations sit in such slices and
  //     their units took 45 % of the kernel's time.)
  // So every row of every narrow unit is in registers from the batch to the v_i, the unit's v_i fit vbuf, and
  // every load an iteration needs is either issued in ONE batch at its top (x, A, Jp: the only demand loads) or was
  // issued an iteration earlier (geometry, view indices, L^-1, run lists); barriers order LDS only, so those stay
  // in flight across them.
  constexpr int RR = reg_rows(D);
  constexpr int J1 = RR * kWaves;  // row slots
  constexpr int NLI = sym_size(DP);
  static_assert(VB >= J1 * 64, "the v_i of a unit fit vbuf");
  static_assert(J1 * 64 <= 2 * kThreads && kWaves == 4, "two runs per thread cover a unit");
#ifdef TMI_MF_PROFILE
  const long long tc0 = clock64();
#endif
  const int4* hdr = m.item_hdr + (size_t)item * 6;
  const int4 h0 = hdr[0], h4 = hdr[4], h5 = hdr[5];
  const int u0 = h0.x, u1 = h0.y, slot0 = h0.z, nlc = h0.w;
  // an item of ONE unit has a slot per run: the sums go straight to HBM (and it may see any number of views)
  const bool direct = (u1 - u0) == 1;
  if (!direct)
    for (int i = threadIdx.x; i < nlc * D; i += kThreads) acc[i] = 0.0;
  const int4* desc = m.unit_desc - m.nwb;  // {first element, rows, rows of one slice | log2 L << 16, first slice}
  const int last_u = m.n_units - 1;
  int4 d0 = hdr[1], d1 = hdr[2], d2 = hdr[3];
  int r0 = h4.x, r1 = h4.y, r2 = h4.z;
  int o0 = h4.w, o1 = h5.x;
  int nrb[RR], npos[RR];
  auto load_index = [&](int4 d) {
    const int lsh = d.z >> 16, t = lane & ((64 >> lsh) - 1), sub = lane >> (6 - lsh);
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
      const int R = ((w + rr * kWaves) << lsh) + sub;
      nrb[rr] = npos[rr] = -1;
      if (R < d.y) {
        const size_t e = (size_t)d.x + 64 * R + t;
        nrb[rr] = v.obs_rb[e];
        npos[rr] = m.obs_pos[e];
      }
    }
  };
  // L^-1 of the tracks wavefront w solves for: slice w of a pack; wavefront 0 for a cut slice
  double Li[NLI];
  auto load_linv = [&](int4 d) {
    const int lsh = d.z >> 16;
    const int G = lsh ? 1 : d.y / (d.z & 0xffff);
    if (w < G) {
      const size_t NP = (size_t)v.Np_pad;
      const size_t lp = (size_t)(d.w + w) * 64 + (d.x & 63) + (lane & ((64 >> lsh) - 1));
#pragma unroll
      for (int i = 0; i < NLI; ++i) Li[i] = v.Linv[(size_t)i * NP + lp];
    }
  };
  int ma[2], mb[2], ms[2];
  auto load_runs = [&](int first) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = min(first + 4 * (lane + 64 * q) + w, m.n_runs - 1);
      ma[q] = m.run_obs_ptr[r];
      mb[q] = m.run_obs_ptr[r + 1];
      ms[q] = m.run_slot[r];
    }
  };
  load_index(d0);
  load_linv(d0);
  load_runs(r0);
#ifdef TMI_MF_PROFILE
  long long tp[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_begin = tc0;
  long long tc = clock64();
#define MF_LAP(i) do { const long long n_ = clock64(); tp[i] += n_ - tc; tc = n_; } while (0)
#else
#define MF_LAP(i)
#endif
  for (int u = u0; u < u1; ++u) {
    const int rows = d0.y, K = d0.z & 0xffff, lsh = d0.z >> 16;
    const int nt = 64 >> lsh, t = lane & (nt - 1), sub = lane >> (6 - lsh);
    const int G = lsh ? 1 : rows / K;  // slices of a pack
    const size_t tile0 = (size_t)(d0.x >> 6);
    const int tl = (d0.x & 63) + t;  // the track's column of the tiles
    double2 ar[RR][D], jr[RR][DP];
    double xr[RR][D];
    double uu[RR][2];
    double pcr[RR][3];  // DROP: -w / scale_p of the row's track
    double2 pnr[RR];    // CP: p_n of the row's observation, {X, w} and 1 / scale_p of its track
    double r2r[RR];     // CP == 2: r^2 of the uncorrected point (-1: take |p_n|^2)
    double Xr[RR][4], ispr[RR][3];
    int pos[RR];
    // ---- the batch of loads
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
      const int R = ((w + rr * kWaves) << lsh) + sub;
      pos[rr] = nrb[rr] >= 0 ? npos[rr] : -1;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        xr[rr][a] = 0.0;
        ar[rr][a] = make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int a = 0; a < DP; ++a) jr[rr][a] = make_double2(0.0, 0.0);
      if (R < rows) {
        // the view's block of x: D doubles at an 8-byte aligned address, fetched 16 bytes at a time
#if TMI_MF_ABL & 2
        const double* xc = x + (size_t)(lane & 7) * D;  // (timing experiment: eight cache-resident blocks)
#else
        const double* xc = x + (size_t)max(nrb[rr], 0) * D;
#endif
#pragma unroll
        for (int a = 0; a + 1 < D; a += 2) {
          const double2_a8 t2 = *reinterpret_cast<const double2_a8*>(xc + a);
          xr[rr][a] = t2.x;
          xr[rr][a + 1] = t2.y;
        }
        if (D & 1) xr[rr][D - 1] = xc[D - 1];
        const double* ap = v.pm_A + (tile0 + R) * ROWD + 2 * tl;
        const double* jp = v.pm_Jp + (tile0 + R) * ROWP + 2 * tl;
        // streamed once: non-temporal, so that the rows do not push the x blocks (gathered again and again by the
        // units of an item) out of the vector L1
        if (!CP) {
#pragma unroll
          for (int a = A0; a < D; ++a) ar[rr][a] = MF_STREAM_LOAD(reinterpret_cast<const double2*>(ap + a * 128));
        } else {
          pnr[rr] = MF_STREAM_LOAD(reinterpret_cast<const double2*>(v.pm_A + (tile0 + R) * CPT + 2 * tl));
          if (CP == 2) r2r[rr] = MF_STREAM_LOAD(reinterpret_cast<const double2*>(v.pm_A + (tile0 + R) * CPT + 128 + 2 * tl)).x;
        }
#pragma unroll
        for (int a = 0; a < DP; ++a) jr[rr][a] = MF_STREAM_LOAD(reinterpret_cast<const double2*>(jp + a * 128));
        if (DROP) {
          // the row's track: slice R / K of a pack, the piece's own slice otherwise
          const size_t lpr = (size_t)(d0.w + (lsh ? 0 : min(R / K, G - 1))) * 64 + tl;
          if (!CP) {
#pragma unroll
            for (int a = 0; a < 3; ++a) pcr[rr][a] = v.pos_coef[(size_t)a * v.Np_pad + lpr];
          } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) Xr[rr][a] = v.cp_trk[(size_t)a * v.Np_pad + lpr];
#pragma unroll
            for (int a = 0; a < 3; ++a) ispr[rr][a] = v.cp_trk[(size_t)(4 + a) * v.Np_pad + lpr];
          }
        }
      }
      if (DROP && !(R < rows)) {
#pragma unroll
        for (int a = 0; a < 3; ++a) pcr[rr][a] = 0.0;
        if (CP) {
          pnr[rr] = make_double2(0.0, 0.0);
          r2r[rr] = 0.0;
#pragma unroll
          for (int a = 0; a < 4; ++a) Xr[rr][a] = 0.0;
#pragma unroll
          for (int a = 0; a < 3; ++a) ispr[rr][a] = 0.0;
        }
      }
    }
    MF_LAP(0);
    // one unit ahead: the view indices; three / two ahead: geometry and run ranges (wave-uniform, tiny)
    if (u + 1 < u1) load_index(d1);
    const int4 d3 = desc[min(u + 3, last_u)];
    const int r3 = m.unit_run_ptr[min(u + 3, m.n_units)];
    const int o2 = m.run_obs_ptr[r2];
    // ---- u_i = A_i x; this lane's share of w = sum Jp^T u per row slot
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
      double s0 = 0.0, s1 = 0.0;
      if (!CP) {
#pragma unroll
        for (int a = A0; a < D; ++a) {
          s0 += ar[rr][a].x * xr[rr][a];
          s1 += ar[rr][a].y * xr[rr][a];
        }
      }
      if constexpr (CP) {
        const double2 j3[3] = {jr[rr][0], jr[rr][1], jr[rr][2]};
        compact_ax(xr[rr], Xr[rr], ispr[rr], j3, pnr[rr], s0, s1, CP == 2 ? r2r[rr] : -1.0);
      } else if (DROP) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const double ca = pcr[rr][a] * xr[rr][a];
          s0 += jr[rr][a < DP ? a : 0].x * ca;
          s1 += jr[rr][a < DP ? a : 0].y * ca;
        }
      }
      if (pos[rr] < 0) {  // no observation here: whatever the loads fetched must not reach the sums
        s0 = s1 = 0.0;
#pragma unroll
        for (int a = 0; a < DP; ++a) jr[rr][a] = make_double2(0.0, 0.0);
      }
      uu[rr][0] = s0;
      uu[rr][1] = s1;
      double wv[DP];
#pragma unroll
      for (int a = 0; a < DP; ++a) wv[a] = jr[rr][a].x * s0 + jr[rr][a].y * s1;
      // a cut slice: the L lanes of a track add their rows (a fixed butterfly; every lane ends with the sum)
      for (int mk = nt; mk < 64; mk <<= 1) {
#pragma unroll
        for (int a = 0; a < DP; ++a) wv[a] += __shfl_xor(wv[a], mk, 64);
      }
#pragma unroll
      for (int a = 0; a < DP; ++a) wpart[w + rr * kWaves][a][lane] = wv[a];
    }
    MF_LAP(1);
    lds_barrier();  // wpart (and: the previous unit's run sums have left vbuf, which wpart shares its memory with)
    MF_LAP(2);
    if (w < G) {
      // slice w of a pack: its rows are the slots [w K, w K + K); a cut slice: every slot, the L lanes of the track
      double wt[DP], zh[DP];
#pragma unroll
      for (int a = 0; a < DP; ++a) wt[a] = 0.0;
      {
        // every slot is read (independent LDS loads); the slots of slice w of a pack -- of a cut slice: all of them --
        // are added, in slot order
        const int s_lo = lsh ? 0 : w * K, s_hi = lsh ? J1 : w * K + K;
        double wp[J1][DP];
#pragma unroll
        for (int sl = 0; sl < J1; ++sl)
#pragma unroll
          for (int a = 0; a < DP; ++a) wp[sl][a] = wpart[sl][a][lane];
#pragma unroll
        for (int sl = 0; sl < J1; ++sl)
#pragma unroll
          for (int a = 0; a < DP; ++a) wt[a] += (sl >= s_lo && sl < s_hi) ? wp[sl][a] : 0.0;
      }
      // z = L^-T (L^-1 w)   (Linv planes: sym_idx(a, b), a <= b, holds L^-1(b, a))
#pragma unroll
      for (int bb = 0; bb < DP; ++bb) {
        double tt = 0.0;
#pragma unroll
        for (int a = 0; a <= bb; ++a) tt += Li[sym_idx(a, bb, DP)] * wt[a];
        zh[bb] = tt;
      }
#pragma unroll
      for (int a = 0; a < DP; ++a) {
        double tt = 0.0;
#pragma unroll
        for (int bb = a; bb < DP; ++bb) tt += Li[sym_idx(a, bb, DP)] * zh[bb];
        zs[w][a][lane] = tt;
      }
    }
    if (u + 1 < u1) load_linv(d1);  // consumed an iteration from now
    MF_LAP(3);
    lds_barrier();  // zs
    MF_LAP(4);
    // t_i = u_i - Jp_i z, v_i = A_i^T t_i into LDS in the unit's view order
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
      const int g = lsh ? 0 : min((w + rr * kWaves) / K, G - 1);
#pragma unroll
      for (int a = 0; a < DP; ++a) {
        const double za = zs[g][a][lane];
        uu[rr][0] -= jr[rr][a].x * za;
        uu[rr][1] -= jr[rr][a].y * za;
      }
#if TMI_MF_ABL & 4
      if (pos[rr] >= 0 && uu[rr][0] == 1.2345e301) {  // (timing experiment: the v_i are formed but not stored)
#else
      if (pos[rr] >= 0) {
#endif
        double* dst = &vbuf[pos[rr] * D];
        if (!CP) {
#pragma unroll
          for (int a = A0; a < D; ++a) dst[a] = ar[rr][a].x * uu[rr][0] + ar[rr][a].y * uu[rr][1];
        }
        if constexpr (CP) {
          const double2 j3[3] = {jr[rr][0], jr[rr][1], jr[rr][2]};
          double o[9];
          compact_at(Xr[rr], ispr[rr], j3, pnr[rr], uu[rr][0], uu[rr][1], o, CP == 2 ? r2r[rr] : -1.0);
#pragma unroll
          for (int a = 0; a < 9; ++a) dst[a] = o[a];
        } else if (DROP) {
#pragma unroll
          for (int a = 0; a < 3; ++a)
            dst[a] = pcr[rr][a] * (jr[rr][a < DP ? a : 0].x * uu[rr][0] + jr[rr][a < DP ? a : 0].y * uu[rr][1]);
        }
      }
    }
    MF_LAP(5);
    lds_barrier();  // vbuf
    MF_LAP(6);
    // The unit's runs (at most 2 kThreads: one per observation at worst): run r0 + 4 i + w belongs to thread i of
    // wavefront w -- consecutive runs are consecutive views, and the views many tracks share (long runs) come in
    // clusters, so dealing them round keeps the wavefronts even.  A thread sums its run's (consecutive) entries; a run
    // of more than kLongRun entries (the tracks of a slice share their lowest view: 64) would hold its wavefront for
    // as many dependent trips, so the wavefront takes those together afterwards: lane g D + a sums component a of
    // the entries g, g + 64 / D, ..; the partial sums are added in g order.  Fixed orders, no atomics.
    auto put_sum = [&](const double (&sum)[D], int slot) {
      if (direct) {
        double* dst = m.partial + (size_t)slot * D;
#pragma unroll
        for (int a = 0; a < D; ++a) dst[a] = sum[a];
      } else {
        double* dst = &acc[(slot - slot0) * D];  // one run per view and unit: nobody else adds here
#pragma unroll
        for (int a = 0; a < D; ++a) dst[a] += sum[a];
      }
    };
    unsigned long long is_long[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const bool mine = r0 + 4 * (lane + 64 * q) + w < r1;
      const int b = ma[q] - o0, e = mb[q] - o0;
      is_long[q] = __ballot(mine && e - b > kLongRun);
#if TMI_MF_ABL & 1
      is_long[q] = 0;  // (timing experiment: no run sums at all)
      if (false) {
#else
      if (mine && e - b <= kLongRun) {
#endif
        double sum[D];
#pragma unroll
        for (int a = 0; a < D; ++a) sum[a] = 0.0;
        for (int p = b; p < e; ++p) {
#pragma unroll
          for (int a = 0; a < D; ++a) sum[a] += vbuf[p * D + a];
        }
        put_sum(sum, ms[q]);
      }
    }
    MF_LAP(7);
    {
      constexpr int NG = 64 / D;  // entry groups
      const int g = lane / D, a = lane - g * D;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        unsigned long long todo = is_long[q];
        while (todo) {
          const int i = __builtin_ctzll(todo);
          todo &= todo - 1;
          const int b = __builtin_amdgcn_readlane(ma[q], i) - o0, e = __builtin_amdgcn_readlane(mb[q], i) - o0;
          const int slot = __builtin_amdgcn_readlane(ms[q], i);
          double part = 0.0;
          if (g < NG)
            for (int p = b + g; p < e; p += NG) part += vbuf[p * D + a];
          double tot = part;  // (lanes a < D: group 0's partial)
#pragma unroll
          for (int gg = 1; gg < NG; ++gg) tot += __shfl(part, gg * D + a);
          if (lane < D) {
            if (direct) m.partial[(size_t)slot * D + a] = tot;
            else acc[(slot - slot0) * D + a] += tot;
          }
        }
      }
    }
    MF_LAP(8);
    lds_barrier();  // vbuf is free (wpart shares its memory)
    MF_LAP(9);
    if (u + 1 < u1) load_runs(r1);  // consumed an iteration from now
    d0 = d1;
    d1 = d2;
    d2 = d3;
    r0 = r1;
    r1 = r2;
    r2 = r3;
    o0 = o1;
    o1 = o2;
  }
#ifdef TMI_MF_PROFILE
  const long long t_loop = clock64();
#endif
#if TMI_MF_ABL & 8
  if (!direct && x[0] == 1.2345e301) {  // (timing experiment: the item's slots are not written)
#else
  if (!direct) {
#endif
    __syncthreads();  // acc
    double* out = m.partial + (size_t)slot0 * D;
    for (int i = threadIdx.x; i < nlc * D; i += kThreads) out[i] = acc[i];
  }
#ifdef TMI_MF_PROFILE
  if (threadIdx.x == 0 && m.prof) {
    long long in_loop = 0;
    for (int i = 0; i < 10; ++i) {
      m.prof[(size_t)item * 16 + i] = tp[i];
      in_loop += tp[i];
    }
    m.prof[(size_t)item * 16 + 10] = u1 - u0;
    m.prof[(size_t)item * 16 + 11] = tc - t_begin - in_loop;
    m.prof[(size_t)item * 16 + 12] = clock64() - t_loop;
  }
#endif
}

// y = D_c x + the sum of every view's slots (ascending = item order) [+ x . y behind the vector]
template <int D>
__global__ __launch_bounds__(256) void reduce_kernel(DeviceView v, View m, RedLayout L, const double* __restrict__ x,
                                                     double* __restrict__ y, double inv_radius, double lm_lo,
                                                     double lm_hi, int add_diag, int dot) {
  __shared__ double sh[4][D];
  __shared__ double prod[D];
  if (m.guard && *m.guard) return;
  const int chunk = ((int)gridDim.x) >> 3;
  const int rb = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  const bool live = rb < v.Nrb;  // the padding workgroups still take part in the dot product's ticket
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double a9[D];
#pragma unroll
  for (int a = 0; a < D; ++a) a9[a] = 0.0;
  if (live) {
    // a slot is D doubles at an 8-byte aligned address: fetched 16 bytes at a time, the slot index of the thread's next
    // trip already in flight (the two loads of a trip are dependent; a view has ~3 trips per thread).  (Round 6 measured a
    // variant with the indices and the slots of four trips loaded as two batches: 42 us against 32, not kept.)
    const int k1 = m.cam_slot_ptr[rb + 1];
    int k = m.cam_slot_ptr[rb] + (int)threadIdx.x;
    int slot = k < k1 ? m.cam_slots[k] : 0;
    while (k < k1) {
      const double* p = m.partial + (size_t)slot * D;
      const int kn = k + 256;
      if (kn < k1) slot = m.cam_slots[kn];
      double t[D];
#pragma unroll
      for (int a = 0; a + 1 < D; a += 2) {
        const double2_a8 t2 = *reinterpret_cast<const double2_a8*>(p + a);
        t[a] = t2.x;
        t[a + 1] = t2.y;
      }
      if (D & 1) t[D - 1] = p[D - 1];
#pragma unroll
      for (int a = 0; a < D; ++a) a9[a] += t[a];
      k = kn;
    }
  }
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const double t = wave_sum(a9[a]);
    if (lane == 0) sh[w][a] = t;
  }
  __syncthreads();
  if ((int)threadIdx.x < D) {
    const int a = threadIdx.x;
    double tot = sh[0][a] + sh[1][a] + sh[2][a] + sh[3][a];
    double pr = 0.0;
    if constexpr (D == 9) {
      if (live && v.compact) {
        // compact planes: the sums are the view's moments; component a of A^T t from all nine (compact_backward)
        double raw[9];
#pragma unroll
        for (int c = 0; c < 9; ++c) raw[c] = sh[0][c] + sh[1][c] + sh[2][c] + sh[3][c];
        tot = compact_backward(v.prep + (size_t)v.rb_cam[rb] * kPrepStride, raw, a);
      }
    }
    if (live) {
      // (drop_pos: the sums of the position entries still lack the view's column scale, device_view.h)
      if (v.drop_pos && a < 3) tot *= v.scale_c[(size_t)rb * D + a];
      const double xa = x[(size_t)rb * D + a];
      // the damping (and the identity on padding rows) enters once: on rank 0 when the product is all-reduced
      if (add_diag) {
        if (v.rb_cols[(size_t)rb * D + a] < 0) {
          tot = xa;
        } else {
          const double d = v.red[L.udiag + (size_t)rb * D + a];
          tot += fmin(fmax(d, lm_lo), lm_hi) * inv_radius * xa;
        }
      } else if (v.rb_cols[(size_t)rb * D + a] < 0) {
        tot = 0.0;
      }
      y[(size_t)rb * D + a] = tot;
      pr = tot * xa;
    }
    prod[a] = pr;
  }
  if (dot) {
    __syncthreads();
    double mine = 0.0;
    if (threadIdx.x == 0)
      for (int a = 0; a < D; ++a) mine += prod[a];
    double total;
    if (last_block_sum<256>(mine, v.dotbuf, v.ticket, &total) && threadIdx.x == 0) y[(size_t)v.Nrb * D] = total;
  }
}

}  // namespace mfc
}  // namespace tmi
