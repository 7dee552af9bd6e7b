// HIP kernels of the bundle-adjustment engine (gfx950, wave64, fp64).
// One Levenberg-Marquardt iteration is
//   linearize -> point_eliminate -> camera_diag + schur_offdiag -> [all-reduce]
//   -> finish_diag + precond_invert -> PCG (pcg_a / symmetric spmv / pcg_b) or dense Cholesky
//   -> back_substitute -> update -> cost
// Every kernel is HBM-bound gather/stream work; the only contraction
// (S_ij = -sum_pairs Y_i Y_j^T, inner dimension 3 x #common tracks) runs on
// the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
//
// Mathematics: SURVEY App. A/B; the reference call site these replace is
// ceres::Solve at src/theia/sfm/bundle_adjustment/bundle_adjuster.cc:205.
#pragma once
#include <hip/hip_runtime.h>

#include "camera_models.h"
#include "device_view.h"

namespace tmi {

typedef double v4f64 __attribute__((ext_vector_type(4)));
// 16-byte vector that may sit on an 8-byte boundary (global loads only need dword alignment)
typedef double double2_a8 __attribute__((ext_vector_type(2), aligned(8)));

constexpr int kWave = 64;
constexpr int kSlicesPerBlock = 4;  // 256 threads

__host__ __device__ constexpr int sym_idx(int a, int b, int n) {
  // index of (a, b), a <= b, in a row-wise packed upper triangle of an n x n matrix
  return a * n - a * (a - 1) / 2 + (b - a);
}
__host__ __device__ constexpr int sym_size(int n) { return n * (n + 1) / 2; }
// Per-observation planes are stored in TILES of 64 observations: element (plane, e) of an
// NPL-plane buffer lives at (e / 64) * NPL * 64 + (plane / 2) * 128 + 2 (e % 64) + plane % 2.  A wave working on 64
// consecutive observations then reads / writes ONE contiguous NPL * 512-byte region per buffer
// instead of NPL regions No_pad * 8 bytes apart: `linearize` used to keep 26 write streams open
// per wave and spent 63 % of its wave cycles in s_waitcnt at 2.3 TB/s of stores.
// 16-byte store with the non-temporal hint: data that is written once and read back from HBM by a
// later kernel must not write-allocate in L2 (linearize: 0.49 -> 0.36 ms with the hint alone)
typedef double nt_double2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_nt(double* dst, const double* src) {
  __builtin_nontemporal_store(*reinterpret_cast<const nt_double2*>(src), reinterpret_cast<nt_double2*>(dst));
}
// Inside a tile the two rows of a column (planes 2a and 2a + 1: the u- and v-residual rows of the
// same Jacobian column) are interleaved per lane, so both come and go in one 16-byte access per lane.
// store into a plane of storage type PT (write-once data: non-temporal)
template <typename PT>
__device__ __forceinline__ void plane_store(double v, PT* p) {
  __builtin_nontemporal_store((PT)v, p);
}
// camera-major records in their storage type (double, or float with fp32 evaluation on shared-intrinsics problems --
// DeviceView::planes_fp32): two consecutive words of a record, widened
typedef float nt_float2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 rec_ld2(const double* p) { return *reinterpret_cast<const double2*>(p); }
__device__ __forceinline__ double2 rec_ld2(const float* p) {
  const float2 t = *reinterpret_cast<const float2*>(p);
  return make_double2((double)t.x, (double)t.y);
}
// ... and the staged (fp64, LDS) pair `src` stored non-temporally at words [idx, idx + 2) of the record array `base`
template <typename RS>
__device__ __forceinline__ void rec_store2_nt(double* base, size_t idx, const double* src) {
  if constexpr (sizeof(RS) == 8) {
    store_nt(base + idx, src);
  } else {
    nt_float2 f;
    f.x = (float)src[0];
    f.y = (float)src[1];
    __builtin_nontemporal_store(f, reinterpret_cast<nt_float2*>(reinterpret_cast<float*>(base) + idx));
  }
}
template <int NPL>
__host__ __device__ __forceinline__ size_t pidx(int plane, size_t e) {
  return (e >> 6) * (size_t)(NPL * 64) + (size_t)(plane >> 1) * 128 + ((e & 63) << 1) + (size_t)(plane & 1);
}
// camera-major record strides in doubles, rounded up to whole 64-byte sectors so that a
// record never straddles an extra sector (gathers) and is written as full sectors
__host__ __device__ constexpr int ys_of(int D, int DP) { return (D * DP + 7) & ~7; }
__host__ __device__ constexpr int as_of(int D, bool SH = false) { return SH ? ((4 * D + 12 + 7) & ~7) : ((2 * D + 7 + 7) & ~7); }
// camera-major A record without shared blocks: [A row 0 (D) | A row 1 (D) | N00 N01 N11 | r~ (2) | r (2)]
// N = I - Q Q^T (Q = Jp L^-T, 2 x DP): sum A^T N A = sum (A^T A - Y Y^T), so the per-camera
// reductions read this record only and never the Y record.
// With shared intrinsics blocks (SH) ONE record per observation:
//   [A row 0 (D) | A row 1 (D) | Q row 0 (4) | Q row 1 (4) | r~ (2) | r (2) | A1 row 0 (D) | A1 row 1 (D)]
// (A1 = the Jacobian w.r.t. the view's SHARED intrinsics; N is recomputed from Q by the per-camera sums; Q is what the
// one-sweep matrix-free product needs: S x = sum A^T (A x_c + A1 x_g - Q zhat) ...) -- 48 doubles for D = 9.
__host__ __device__ constexpr int sh_off_q(int D) { return 2 * D; }
__host__ __device__ constexpr int sh_off_rt(int D) { return 2 * D + 8; }
__host__ __device__ constexpr int sh_off_r(int D) { return 2 * D + 10; }
__host__ __device__ constexpr int sh_off_a1(int D) { return 2 * D + 12; }
__host__ __device__ constexpr int a_off_n(int D) { return 2 * D; }
__host__ __device__ constexpr int a_off_rt(int D) { return 2 * D + 3; }
__host__ __device__ constexpr int a_off_r(int D) { return 2 * D + 5; }
// Without shared intrinsics blocks the A record is kept as TWO arrays in the same allocation: the A rows
// alone (stride asa_of(D): 192 B for D = 9, what the cameras pass of the matrix-free product streams 4-5
// times per LM iteration) and the 64-byte tail {N, r~, r} (cm_R) that only camera_diag reads.  Same bytes
// written by point_eliminate and read by camera_diag as the one 256-byte record they replace.
// The A rows are followed by Q = Jp L^-T (2 x DP) in the same record: Y Y'^T = A^T (Q Q'^T) A', so the Schur
// complement gathers these [A | Q] records (192 B for 9 x 3, two lines -- what a 216-byte Y record costs as
// well; 256 B against 320 B and three lines for 9 x 4) and no Y record is written at all.
__host__ __device__ constexpr int asa_of(int D, int DP) { return (2 * D + 2 * DP + 7) & ~7; }
constexpr int kRecTail = 8;  // doubles of the tail record
// doubles of cm_A per slot: one record with shared blocks, rows + tail otherwise
__host__ __device__ constexpr int a_alloc_of(int D, int DP, bool SH) {
  return SH ? as_of(D, true) : asa_of(D, DP) + kRecTail;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// ------------------------------------------------------------------------------
// Grid-wide hand-over inside ONE launch ("the last workgroup finishes the reduction").
// MI355X is eight XCDs with separate L2s: an agent-scope FENCE writes back / invalidates the whole
// L2 of the XCD (measured: a cost kernel whose 4000 workgroups each execute __threadfence() went
// from 0.16 ms to over 1 ms).  The hand-over below therefore uses no fence at all: the few values
// that cross workgroups are written and read with agent-scope RELAXED ATOMICS (write-through /
// cache-bypassing single accesses), the ticket is an agent-scope atomic add, and the only ordering
// needed -- "my stores have completed before I take the ticket" -- is a workgroup-scope release
// (s_waitcnt, no cache maintenance).  Anything bigger than a handful of scalars per workgroup
// must cross a kernel boundary instead.
// ------------------------------------------------------------------------------
// HARDWARE ASSUMPTION (gfx950): an agent-scope relaxed atomic store is a write-through store (sc1) and is complete --
// visible to agent-scope atomic loads from any XCD -- once s_waitcnt vmcnt(0) has drained it, which the workgroup-scope
// release before the ticket does.  The HIP memory model does not promise that ordering across workgroups; another
// target (or the threadgroup-split mode) would need an agent-scope release on the ticket instead.  Refuse to build
// for anything else rather than corrupt cost / gradient / dot-product scalars silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "the fence-free grid hand-over of kernels.h / dense_cholesky_df.h is validated for gfx950 only"
#endif
__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_agent(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Call from ONE thread after the workgroup's hand-over stores are done (and a __syncthreads if other
// threads made them); true in the workgroup that arrives last.  A single counter serialises one
// atomic per workgroup at one address (measured: +25 us for a 1778-workgroup kernel), so arrivals
// are counted in kTicketGroups sub-counters and only the last arrival of each sub-counter touches
// the top one.  `ticket` points at a region of kTicketStride ints: [0] top, [1 ..] sub-counters;
// all of them are back at zero when the function returns true.
constexpr int kTicketGroups = 32;
constexpr int kTicketStride = 64;
__device__ __forceinline__ bool take_ticket(int* ticket, int nblocks) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  const int g = (int)blockIdx.x & (kTicketGroups - 1);
  const int in_group = (nblocks - g + kTicketGroups - 1) / kTicketGroups;  // workgroups with this residue
  if (__hip_atomic_fetch_add(ticket + 1 + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != in_group - 1) return false;
  __hip_atomic_store(ticket + 1 + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int groups = nblocks < kTicketGroups ? nblocks : kTicketGroups;
  if (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != groups - 1) return false;
  __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// ------------------------------------------------------------------------------
// Track -> lane mapping of the per-track kernels.
// Slices are sorted by track length.  A thread per track means the longest slice sets a
// serial floor (63 dependent iterations on the Venice-sized problem) that does not shrink
// when the tracks are sharded over several GPUs.  The first v.n_wide slices (those whose
// longest track has >= kWideK observations) are therefore run "wide": a track is shared by
// 16 consecutive lanes (lane i takes observations i, i + 16, ...), per-track sums are
// finished with a fixed 16-lane butterfly, and a 256-thread workgroup covers a quarter of a
// slice.  The first v.n_ultra of those (longest track >= kUltraK) give every track a whole
// wavefront (lane i takes observations i, i + 64, ...; a workgroup covers four tracks): a
// 400-view track is 7 trips instead of 25, which is what the per-track kernels cost once the
// tracks are spread over 8 GPUs.  All other slices keep the thread-per-track mapping
// (workgroup = 4 slices).
//   grid = 16 n_ultra + 4 (n_wide - n_ultra) + ceil((nslices - n_wide) / 4)   (DeviceView::n_track_blocks)
// Which mapping a slice gets depends on the slice only (never on the rank count), so the
// summation order of a track is the same in a sharded and an unsharded run.
// ------------------------------------------------------------------------------
constexpr int kWideLanes = 16;

struct TrackMap {
  int s;        // slice
  int lp;       // padded track index 64 s + t
  int k;        // observations of the track (0 = padding)
  size_t base;  // element of observation 0: slice_ptr[s] + t; observation j at base + 64 j
  int j0, jstep;  // this lane's observations j0, j0 + jstep, ...
  int trips;    // wave-uniform trip count covering the slice
  int wide;     // 0: thread per track, 1: 16 lanes per track, 2: 64 lanes per track
  bool leader, valid;
};

// (bid, tid): the workgroup and thread of the 256-thread launch the mapping was made for -- a kernel with another launch
// shape (back_substitute_lds_kernel) walks the same virtual workgroups
__device__ __forceinline__ TrackMap track_map_at(const DeviceView& v, int bid, int tid) {
  TrackMap m;
  const int lane = tid & 63, w = tid >> 6;
  const int nub = 16 * v.n_ultra;
  const int nwb = nub + 4 * (v.n_wide - v.n_ultra);
  int t;
  if (bid < nub) {
    m.wide = 2;
    m.s = bid >> 4;
    t = 4 * (bid & 15) + w;
    m.j0 = lane;
    m.jstep = 64;
    m.leader = lane == 0;
  } else if (bid < nwb) {
    m.wide = 1;
    const int bw = bid - nub;
    m.s = v.n_ultra + (bw >> 2);
    t = 16 * (bw & 3) + 4 * w + (lane >> 4);
    m.j0 = lane & (kWideLanes - 1);
    m.jstep = kWideLanes;
    m.leader = m.j0 == 0;
  } else {
    m.wide = 0;
    m.s = v.n_wide + (bid - nwb) * kSlicesPerBlock + w;
    t = lane;
    m.j0 = 0;
    m.jstep = 1;
    m.leader = true;
  }
  m.valid = m.s < v.nslices;
  m.lp = m.s * 64 + t;
  m.k = 0;
  m.base = 0;
  m.trips = 0;
  if (m.valid) {
    m.k = v.pt_k[m.lp];
    const int sp0 = v.slice_ptr[m.s];
    m.base = (size_t)sp0 + t;
    const int K = (v.slice_ptr[m.s + 1] - sp0) >> 6;
    m.trips = m.wide == 2 ? (K + 63) / 64 : m.wide == 1 ? (K + kWideLanes - 1) / kWideLanes : K;
  }
  return m;
}
__device__ __forceinline__ TrackMap track_map(const DeviceView& v) { return track_map_at(v, (int)blockIdx.x, (int)threadIdx.x); }

// sum over the 16 lanes that share a track (identity for thread-per-track slices); every lane
// of the group receives the total; fixed butterfly => reproducible
__device__ __forceinline__ double group_sum(double x, int wide) {
  if (wide) {
    x += __shfl_xor(x, 1, 64);
    x += __shfl_xor(x, 2, 64);
    x += __shfl_xor(x, 4, 64);
    x += __shfl_xor(x, 8, 64);
    if (wide == 2) {
      x += __shfl_xor(x, 16, 64);
      x += __shfl_xor(x, 32, 64);
    }
  }
  return x;
}

// Sum NV per-thread values over a 256-thread workgroup and store them at
// partial[v * nblocks + blockIdx.x] (fixed order => reproducible).
template <int NV>
__device__ __forceinline__ void block_sum_store(const double (&v)[NV], double* partial, int nblocks) {
  __shared__ double sh[NV][kSlicesPerBlock];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double s = wave_sum(v[i]);
    if (lane == 0) sh[i][w] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < kSlicesPerBlock; ++k) s += sh[threadIdx.x][k];
    st_agent(&partial[(size_t)threadIdx.x * nblocks + blockIdx.x], s);
  }
}

// block_sum_store + the grid-wide finish in the same launch: the last workgroup to arrive adds the
// per-workgroup partials exactly as reduce_sum_kernel would (thread t takes i = t, t + 256, ...,
// then the same tree), so results are bit-identical to the two-launch form.  dst == nullptr keeps
// the partials only.  flag_src / flag_dst: optionally the last workgroup also turns a device
// flag into the double 0 / 1 that rides in the all-reduced scalar tail.
template <int NV>
__device__ __forceinline__ void block_sum_finish(const double (&v)[NV], double* partial, int nblocks,
                                                 int* ticket, double* dst, const int* flag_src = nullptr,
                                                 double* flag_dst = nullptr) {
  block_sum_store<NV>(v, partial, nblocks);
  if (dst == nullptr) return;
  __shared__ double fin_sh[256];
  __shared__ int fin_last;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (threadIdx.x == 0) fin_last = take_ticket(ticket, nblocks) ? 1 : 0;
  __syncthreads();
  if (!fin_last) return;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const double* src = partial + (size_t)q * nblocks;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) acc += ld_agent(&src[i]);
    fin_sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) fin_sh[threadIdx.x] += fin_sh[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) dst[q] = fin_sh[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (flag_dst) *flag_dst = ld_agent(flag_src) ? 1.0 : 0.0;
  }
}

// dst[v] = sum_{i<n} src[v*n + i]   (grid = #values, block = 256)
__global__ void reduce_sum_kernel(const double* __restrict__ src, int n, double* __restrict__ dst) {
  __shared__ double sh[256];
  const double* s = src + (size_t)blockIdx.x * n;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += s[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) dst[blockIdx.x] = sh[0];
}
__global__ void reduce_max_kernel(const double* __restrict__ src, int n, double* __restrict__ dst) {
  __shared__ double sh[256];
  const double* s = src + (size_t)blockIdx.x * n;
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc = fmax(acc, s[i]);
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) dst[blockIdx.x] = sh[0];
}

__global__ void fill_kernel(double* p, long long n, double v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// scale_c (per reduced column) expanded per view to the 16 source columns [ext(6) | intr(10)];
// launched whenever scale_c changes (start of a solve, after the Jacobi-scaling pre-pass)
template <int D>
__global__ void expand_camera_scale_kernel(DeviceView v) {
  const int cam = blockIdx.x * blockDim.x + threadIdx.x;
  if (cam >= v.Nc) return;
  const unsigned mask = v.cam_mask[cam];
  const int rb = v.cam_rb[cam];
  const int grb = v.has_shared ? v.cam_grb[cam] : -1;
  const unsigned gmask = grb >= 0 ? v.grp_mask[v.cam_grp[cam]] : 0u;
  int dst = 0, dst1 = 0;
  for (int c = 0; c < 16; ++c) {
    double sc = 0.0;
    if (mask & (1u << c)) {
      sc = rb >= 0 ? v.scale_c[(size_t)rb * D + dst] : 0.0;
      ++dst;
    } else if (c >= 6 && (gmask & (1u << (c - 6)))) {
      sc = v.scale_c[(size_t)grb * D + dst1];
      ++dst1;
    }
    v.scale_cam[(size_t)cam * 16 + c] = sc;
  }
}

// ------------------------------------------------------------------------------
// Evaluation path: camera_prepare + linearize + cost.
// linearize (kernel class 0) replaces the N_obs AutoDiffCostFunction evaluations of hot loop 1
// (reprojection_error.h:51-95 under Jets) with analytic Jacobians, applies the loss correction
// (ceres corrector.cc) and the Jacobi column scaling, and writes the reduced blocks as SoA planes.
//
// camera_prepare_kernel: one thread per view turns [C, angle-axis], the view's intrinsics and
// its 16 column scales into the 48-double record of camera_models.h (R, C, K, Jl diag(scale),
// scales).  N_c threads; everything transcendental about the rotation happens here.
//
// The per-observation kernels are track-major, so the 64 lanes of a wave need 64 DIFFERENT
// camera records per trip.  A lane reading its record straight from global memory costs one
// L1 tag lookup per lane per 16-byte load (64 lookups per instruction, 12 instructions for
// half a record); instead the wave copies the 64 half records (192 B each) into LDS
// COOPERATIVELY -- consecutive lanes fetch consecutive 16-byte chunks, ~5 records per load
// instruction -- and every lane then reads its own record from LDS on demand, which also keeps
// the camera data out of the register file until the expression that uses it.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void camera_prepare_kernel(DeviceView v, const double* __restrict__ ext,
                                                             const double* __restrict__ intr,
                                                             double* __restrict__ prep) {
  const int cam = blockIdx.x * 256 + threadIdx.x;
  if (cam >= v.Nc) return;
  const int4 rec = v.cam_rec[cam];
  prepare_camera_record(ext + (size_t)cam * 6, intr + rec.y, rec.z, v.scale_cam + (size_t)cam * 16,
                        prep + (size_t)cam * kPrepStride);
}

constexpr int kStageWords = kPrepCostWords;  // doubles of one staged half record
constexpr int kStagePitch = kStageWords + 2; // +2 doubles: conflict-free 16-byte LDS accesses

// lane l wants words [off, off + kStageWords) of record idx (idx < 0: nothing); on return
// st[l * kStagePitch + i] holds word off + i of lane l's record.  Wave-collective.
// WORDS: doubles staged per record (default: the half record), the LDS pitch is WORDS + 2
template <int NB = 2, int WORDS = kStageWords>
__device__ __forceinline__ void stage_camera_records(const double* __restrict__ prep, int off, int idx,
                                                     double* st, int lane) {
  constexpr int CH = WORDS / 2;
  constexpr int kStagePitch = WORDS + 2;  // (shadows the default pitch)
  // the previous contents may still be being read by other lanes of this wave
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // Every chunk is loaded UNCONDITIONALLY (a lane without a record fetches record 0; nobody reads what it stages) and
  // all loads of a batch are issued before the first LDS store.  With the load inside `if (id >= 0)` every chunk was
  // a basic block of its own -- global_load, s_waitcnt vmcnt(0), ds_write, twelve times per staging: 24 dependent
  // memory round trips per trip of linearize (round 6: in-kernel phase counters put 44 k of a trip's 54 k cycles
  // into the two stagings, 1.5 k into the evaluation).  Two batches of CH / 2 keep the transient registers at 24.
  constexpr int PER = CH / NB;
  static_assert(CH % NB == 0 && (PER == 5 || PER == 6 || PER == 12), "batches");
#pragma unroll
  for (int bt = 0; bt < NB; ++bt) {
    double2 t[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int f = (bt * PER + q) * 64 + lane;
      const int rec = f / CH, part = f - rec * CH;
      const int id = max(__shfl(idx, rec, 64), 0);
      t[q] = *reinterpret_cast<const double2*>(prep + (size_t)id * kPrepStride + off + 2 * part);
    }
    // (left alone, the compiler sinks every load back to its store -- the kernel is at the register limit: ONE empty
    // asm statement that names all six values keeps the batch together: they must all be in registers there)
    // ("memory": the loads of the second half of a 12-chunk batch stay in front of it too)
    if constexpr (PER == 5)
      asm volatile("" : "+v"(t[0].x), "+v"(t[0].y), "+v"(t[1].x), "+v"(t[1].y), "+v"(t[2].x), "+v"(t[2].y), "+v"(t[3].x),
                        "+v"(t[3].y), "+v"(t[4].x), "+v"(t[4].y) : : "memory");
    else
      asm volatile("" : "+v"(t[0].x), "+v"(t[0].y), "+v"(t[1].x), "+v"(t[1].y), "+v"(t[2].x), "+v"(t[2].y), "+v"(t[3].x),
                        "+v"(t[3].y), "+v"(t[4].x), "+v"(t[4].y), "+v"(t[PER > 5 ? 5 : 0].x), "+v"(t[PER > 5 ? 5 : 0].y) : : "memory");
    if constexpr (PER == 12)
      asm volatile("" : "+v"(t[6].x), "+v"(t[6].y), "+v"(t[7].x), "+v"(t[7].y), "+v"(t[8].x), "+v"(t[8].y), "+v"(t[9].x),
                        "+v"(t[9].y), "+v"(t[10].x), "+v"(t[10].y), "+v"(t[11].x), "+v"(t[11].y));
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int f = (bt * PER + q) * 64 + lane;
      const int rec = f / CH, part = f - rec * CH;
      *reinterpret_cast<double2*>(st + rec * kStagePitch + 2 * part) = t[q];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- compact planes (device_view.h, DeviceView::compact): the per-view maps between a view's block of a reduced
// vector and what the product / back-substitution gather, and between the per-view moment sums and the block of A^T t.
// P = the view's prepared record (camera_models.h: R [0..8], C [9..11], K [12..21], Jl diag(s_rot) [24..32],
// s_pos [33..35], intrinsics scales [36..45]; PINHOLE: f = K[0], k1 = K[5], k2 = K[6]).
//   forward:  x = [x_pos | x_rot | x_f x_k1 x_k2]  ->  g = [kappa | eta | a0 a1 a2]
//     eta = R^T Jl_s x_rot, kappa = eta x C + s_pos .* x_pos,
//     a0 = s_f x_f, a1 = k1 a0 + f s_k1 x_k1, a2 = k2 a0 + f s_k2 x_k2
//   (first-order branch of AngleAxisRotatePoint, theta^2 <= DBL_EPSILON: R = I + [w]x is not orthogonal and the executed
//    derivative takes a instead of q; R^T for R^-1 is then off by O(|w|) <= 1.5e-8 relative -- exact for w = 0)
__device__ __forceinline__ void compact_forward(const double* __restrict__ P, const double (&x)[9], double (&g)[9]) {
  double xi[3], eta[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) xi[i] = P[24 + 3 * i] * x[3] + P[24 + 3 * i + 1] * x[4] + P[24 + 3 * i + 2] * x[5];
#pragma unroll
  for (int a = 0; a < 3; ++a) eta[a] = P[a] * xi[0] + P[3 + a] * xi[1] + P[6 + a] * xi[2];
  const double C0 = P[9], C1 = P[10], C2 = P[11];
  g[0] = eta[1] * C2 - eta[2] * C1 + P[33] * x[0];
  g[1] = eta[2] * C0 - eta[0] * C2 + P[34] * x[1];
  g[2] = eta[0] * C1 - eta[1] * C0 + P[35] * x[2];
  g[3] = eta[0];
  g[4] = eta[1];
  g[5] = eta[2];
  const double f = P[12], k1 = P[17], k2 = P[18];
  const double a0 = P[36] * x[6];
  g[6] = a0;
  g[7] = k1 * a0 + f * (P[41] * x[7]);
  g[8] = k2 * a0 + f * (P[42] * x[8]);
}
//   backward: s = per-view sums [sum -w h | sum X x h | sum (p_n . t) (1, r^2, r^4)], h = Jp'^T t  ->  y = A^T t summed
//     y_pos = s_pos .* s[0..2]   (reduce_kernel's drop_pos line),  y_rot = Jl_s^T R (s[3..5] + C x s[0..2]),
//     y_f = s_f (m0 + k1 m1 + k2 m2), y_k1 = s_k1 f m1, y_k2 = s_k2 f m2.   Component a only (one thread per component).
__device__ __forceinline__ double compact_backward(const double* __restrict__ P, const double (&s)[9], int a) {
  if (a < 3) return s[a];  // (the caller applies scale_c: the same line as for the full planes)
  const double f = P[12], k1 = P[17], k2 = P[18];
  if (a == 6) return P[36] * (s[6] + k1 * s[7] + k2 * s[8]);
  if (a == 7) return P[41] * (f * s[7]);
  if (a == 8) return P[42] * (f * s[8]);
  const double C0 = P[9], C1 = P[10], C2 = P[11];
  const double v0 = s[3] + (C1 * s[2] - C2 * s[1]);
  const double v1 = s[4] + (C2 * s[0] - C0 * s[2]);
  const double v2 = s[5] + (C0 * s[1] - C1 * s[0]);
  const int k = a - 3;
  double y = 0.0;
#pragma unroll
  for (int m = 0; m < 3; ++m) y += P[24 + 3 * m + k] * (P[3 * m] * v0 + P[3 * m + 1] * v1 + P[3 * m + 2] * v2);
  return y;
}
// A x of one observation from the gathered g, the track's {X, w, 1 / scale_p} and the observation's Jp (interleaved
// rows) and p_n
// (a robust loss -- DeviceView::compact == 2 -- multiplies every column pair of the block by the observation's 2 x 2
//  corrector C: Jp is stored corrected anyway, pn is then C p_n, and r^2 = |p_n|^2 of the UNcorrected point comes from
//  its own plane: r2 >= 0; r2 < 0 means "take |pn|^2", the TRIVIAL loss)
__device__ __forceinline__ void compact_ax(const double (&g)[9], const double (&X)[4], const double (&isp)[3],
                                           const double2 (&jp)[3], double2 pn, double& u0, double& u1, double r2_in = -1.0) {
  const double w = X[3];
  const double c0 = (g[4] * X[2] - g[5] * X[1] - w * g[0]) * isp[0];
  const double c1 = (g[5] * X[0] - g[3] * X[2] - w * g[1]) * isp[1];
  const double c2 = (g[3] * X[1] - g[4] * X[0] - w * g[2]) * isp[2];
  const double r2 = r2_in >= 0.0 ? r2_in : pn.x * pn.x + pn.y * pn.y;
  const double pr = g[6] + r2 * (g[7] + r2 * g[8]);
  u0 = jp[0].x * c0 + jp[1].x * c1 + jp[2].x * c2 + pn.x * pr;
  u1 = jp[0].y * c0 + jp[1].y * c1 + jp[2].y * c2 + pn.y * pr;
}
// the nine moment contributions of one observation: [-w h | X x h | (p_n . t) (1, r^2, r^4)], h = Jp'^T t
__device__ __forceinline__ void compact_at(const double (&X)[4], const double (&isp)[3], const double2 (&jp)[3], double2 pn,
                                           double t0, double t1, double (&o)[9], double r2_in = -1.0) {
  const double h0 = isp[0] * (jp[0].x * t0 + jp[0].y * t1);
  const double h1 = isp[1] * (jp[1].x * t0 + jp[1].y * t1);
  const double h2 = isp[2] * (jp[2].x * t0 + jp[2].y * t1);
  const double w = X[3];
  o[0] = -w * h0;
  o[1] = -w * h1;
  o[2] = -w * h2;
  o[3] = X[1] * h2 - X[2] * h1;
  o[4] = X[2] * h0 - X[0] * h2;
  o[5] = X[0] * h1 - X[1] * h0;
  const double r2 = r2_in >= 0.0 ? r2_in : pn.x * pn.x + pn.y * pn.y;
  const double pt = pn.x * t0 + pn.y * t1;
  o[6] = pt;
  o[7] = r2 * pt;
  o[8] = r2 * r2 * pt;
}

// the transformed block of view block rb of x into xs (one thread)
__device__ __forceinline__ void compact_forward_view(const DeviceView& v, int rb, const double (&xv)[9], double* __restrict__ xs) {
  double g[9];
  compact_forward(v.prep + (size_t)v.rb_cam[rb] * kPrepStride, xv, g);
#pragma unroll
  for (int a = 0; a < 9; ++a) st_agent(&xs[(size_t)rb * 9 + a], g[a]);  // (pcg_step's last workgroup reads xz: agent scope)
}
// ... where lane a < 9 of a wavefront holds component a of the block (every lane of the wavefront calls)
// (lanes 0..8 each form the block -- the loads of the record are the same addresses, broadcast -- and keep their own entry)
__device__ __forceinline__ double compact_forward_wave_value(const DeviceView& v, int rb, double val, int lane) {
  double xv[9];
#pragma unroll
  for (int a = 0; a < 9; ++a) xv[a] = __shfl(val, a, 64);
  double mine = 0.0;
  if (lane < 9) {
    double g[9];
    compact_forward(v.prep + (size_t)v.rb_cam[rb] * kPrepStride, xv, g);
    mine = g[0];
#pragma unroll
    for (int a = 1; a < 9; ++a) mine = (lane == a) ? g[a] : mine;
  }
  return mine;
}
__device__ __forceinline__ void compact_forward_wave(const DeviceView& v, int rb, double val, int lane, double* __restrict__ xs) {
  const double mine = compact_forward_wave_value(v, rb, val, lane);
  if (lane < 9) st_agent(&xs[(size_t)rb * 9 + lane], mine);
}

// linearize (kernel class 0).
// Per trip: stage [R C K flag] -> value, projection Jacobian, M = dp/dq R, c = p x dp/dq;
// stage [Jl scale] -> angle-axis columns, scaling and the plane stores.
// OCC = minimum workgroups per CU the register allocation must allow (2: 256 registers per
// lane, a handful of doubles spilled in the rarely taken camera-model branches).
#ifdef TMI_LIN_PROFILE
__device__ unsigned long long g_lin_prof[8];
#define LIN_LAP(i) do { const long long n_ = clock64(); lp_[i] += n_ - lt_; lt_ = n_; } while (0)
#else
#define LIN_LAP(i)
#endif
// UMODEL >= 0: every camera of the problem has camera model UMODEL and the free-column mask UMASK (the engine checks
// at create): the model switch and the column compaction fold at compile time, the intrinsics columns nobody stores are
// never formed and the per-observation gather of cam_rec goes -- ~100 registers less than the generic body.
// UDROP: the handle does not store the position columns (DeviceView::drop_pos is set): known at compile time, M is dead
// once the point block is out.
// NORMS (round 6): the start of a solve needs the cost and the squared column norms of the UNSCALED Jacobian (Ceres'
// jacobi_scaling, 1 / (1 + ||column||)) before the planes can be written with their scales: this instantiation stores no
// plane at all -- it sums the point block's column norms per track and leaves scale_p itself (point_scale_kernel's job,
// same sums in the same order), never stages the second half record, and takes a third of the time of the full pass.
// COMPACT (device_view.h): the specialised instantiation stores p_n instead of the camera block -- the second half record
// is never staged, no column of A is formed -- and leaves the track's {X, w, 1 / scale_p} for the consumers.
template <int D, int DP, bool SH, typename RT, int OCC, typename PT = double, int UMODEL = -1, unsigned UMASK = 0u,
          bool UDROP = false, bool NORMS = false, bool COMPACT = false, bool ULOSS = false>
__global__ __launch_bounds__(256, OCC) void linearize_kernel(DeviceView v, const double* __restrict__ prep,
                                                             int loss_type_arg, double loss_width, int nblocks,
                                                             double* __restrict__ sums) {
  // (the specialised instantiation folds the TRIVIAL loss unless ULOSS: the reference's application flags ask for HUBER,
  //  applications/build_reconstruction_flags.txt:117 -- the same bodies with the corrector left in)
  const int loss_type = (UMODEL >= 0 && !ULOSS) ? 0 : loss_type_arg;
  // COMPACT: R, C, f, the principal point, k1, k2 are words 0..18 of the record -- 20 staged words, 45 instead of 53 KB of
  // LDS: a third workgroup fits a CU
  constexpr int SW = COMPACT ? 20 : kStageWords;
  constexpr int kStagePitch = SW + 2;  // (shadows the default pitch)
  __shared__ __attribute__((aligned(16))) double stage[kSlicesPerBlock][64 * kStagePitch];
  const TrackMap tm = track_map(v);
  PT* const pmR = reinterpret_cast<PT*>(v.pm_r);   // (the planes in their storage type: double, or float with
  PT* const pmA = reinterpret_cast<PT*>(v.pm_A);   //  fp32 evaluation on shared-intrinsics problems -- DeviceView::planes_fp32)
  PT* const pmA1 = reinterpret_cast<PT*>(v.pm_A1);
  PT* const pmJp = reinterpret_cast<PT*>(v.pm_Jp);
  (void)pmR; (void)pmA; (void)pmA1; (void)pmJp;
  const int lane = threadIdx.x & 63;
  double* st = stage[threadIdx.x >> 6];
  const double* P = st + lane * kStagePitch;
  double acc[2] = {0.0, 0.0};
  auto PST = [&](double val, PT* ptr) {
    if constexpr (!NORMS) plane_store<PT>(val, ptr);
  };
  double n2[DP];  // NORMS: squared norms of the point block's columns over the track's observations
#pragma unroll
  for (int a = 0; a < DP; ++a) n2[a] = 0.0;
  // COMPACT: V = sum Jp^T Jp and g_p = sum Jp^T r of the track, in point_eliminate's order -- that kernel then skips its
  // sweep over the Jp and r planes (DeviceView::sums_ready)
  double Vt[sym_size(DP)], gt[DP];
#pragma unroll
  for (int i = 0; i < sym_size(DP); ++i) Vt[i] = 0.0;
#pragma unroll
  for (int a = 0; a < DP; ++a) gt[a] = 0.0;
  const int lp = tm.lp;
  const int k = tm.k;  // 0 for padding tracks and beyond the last slice
  double X[4] = {0.0, 0.0, 0.0, 1.0};
  double sp[DP];
#pragma unroll
  for (int a = 0; a < DP; ++a) sp[a] = 0.0;
  bool pconst = false;
  if (tm.valid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) X[i] = v.pts[(size_t)lp * 4 + i];
    pconst = v.pt_const[lp] != 0;
#pragma unroll
    for (int a = 0; a < DP; ++a) sp[a] = NORMS ? 1.0 : v.scale_p[(size_t)lp * DP + a];
    // drop_pos (device_view.h): pos_coef[a][track] = -w / scale_p[a] at the point and the scales these planes are taken at
    // (round 6: was a launch of its own after every linearize)
    // (COMPACT: no consumer of these planes reads pos_coef -- they take {X, w, 1 / scale_p} below)
    if (!NORMS && !COMPACT && !SH && (UDROP || v.drop_pos) && tm.leader) {
#pragma unroll
      for (int a = 0; a < 3; ++a) v.pos_coef[(size_t)a * v.Np_pad + lp] = -X[3] / sp[a];
    }
    if constexpr (COMPACT) {
      if (tm.leader) {
        const size_t NPc = (size_t)v.Np_pad;
#pragma unroll
        for (int a = 0; a < 4; ++a) v.cp_trk[(size_t)a * NPc + lp] = X[a];
#pragma unroll
        for (int a = 0; a < 3; ++a) v.cp_trk[(size_t)(4 + a) * NPc + lp] = 1.0 / sp[a];
      }
    }
  }
  int cam_next = (tm.j0 < k) ? v.obs_cam[tm.base + (size_t)tm.j0 * 64] : -1;
#ifdef TMI_LIN_PROFILE
  long long lp_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long lt_ = clock64();
#endif
  for (int trip = 0; trip < tm.trips; ++trip) {
    const int j = tm.j0 + trip * tm.jstep;
    const bool act = j < k;
    const size_t e = tm.base + (size_t)j * 64;
    const int cam = act ? cam_next : -1;
    if (j + tm.jstep < k) cam_next = v.obs_cam[e + (size_t)tm.jstep * 64];
    int4 rec = make_int4(0, 0, 0, 0);
    double fx = 0.0, fy = 0.0;
    if (act) {
      if (UMODEL < 0) rec = v.cam_rec[cam];
      const double2 f2 = *reinterpret_cast<const double2*>(v.obs_xy + 2 * e);
      fx = f2.x;
      fy = f2.y;
    }
    if (UMODEL >= 0) {
      rec.x = UMODEL;
      rec.w = (int)UMASK;
    }
    LIN_LAP(0);
    stage_camera_records<2, SW>(prep, 0, cam, st, lane);
    LIN_LAP(1);
    // ---- phase A: value, dp/dq, dp/dK; M = dp/dq R; c = p x dp/dq ----
    bool ok = false;
    RT Jint[2][10], M[2][3], cx[2][3], Jp3[2];
    double r[2] = {0.0, 0.0};
    double pn0 = 0.0, pn1 = 0.0;  // COMPACT: the normalised image point
    if (act) {
      const double wd = X[3];
      const double ad[3] = {X[0] - wd * P[9], X[1] - wd * P[10], X[2] - wd * P[11]};
      ok = !(ad[0] * ad[0] + ad[1] * ad[1] + ad[2] * ad[2] < 1e-8);
      if (ok) {
        const RT a[3] = {(RT)ad[0], (RT)ad[1], (RT)ad[2]};
        RT q[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
          q[i] = (RT)P[3 * i] * a[0] + (RT)P[3 * i + 1] * a[1] + (RT)P[3 * i + 2] * a[2];
        RT Kt[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) Kt[i] = (12 + i < SW) ? (RT)P[12 + i] : (RT)0.0;  // (COMPACT stages 20 words: PINHOLE has 7)
        RT dpdq[2][3], px[2];
        project<true, RT>(rec.x, Kt, q, px, dpdq, Jint);
        r[0] = (double)(RT)((double)px[0] - fx);
        r[1] = (double)(RT)((double)px[1] - fy);
        if constexpr (COMPACT) {
          pn0 = (double)q[0] / (double)q[2];
          pn1 = (double)q[1] / (double)q[2];
        }
        const bool small = COMPACT ? false : P[COMPACT ? 0 : 22] != 0.0;  // (only the rotation columns need it: not formed when COMPACT)
        const RT p[3] = {small ? a[0] : q[0], small ? a[1] : q[1], small ? a[2] : q[2]};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int jj = 0; jj < 3; ++jj)
            M[i][jj] = dpdq[i][0] * (RT)P[jj] + dpdq[i][1] * (RT)P[3 + jj] + dpdq[i][2] * (RT)P[6 + jj];
          cx[i][0] = p[1] * dpdq[i][2] - p[2] * dpdq[i][1];
          cx[i][1] = p[2] * dpdq[i][0] - p[0] * dpdq[i][2];
          cx[i][2] = p[0] * dpdq[i][1] - p[1] * dpdq[i][0];
          Jp3[i] = -(M[i][0] * (RT)P[9] + M[i][1] * (RT)P[10] + M[i][2] * (RT)P[11]);
        }
      }
    }
    // The loss corrector and the planes that need nothing of the second half record -- the point block and the residual --
    // go out BEFORE it is staged: M then dies here where the position columns are not stored (UDROP), and fewer values
    // cross the staging (round 6).
    double sqrt_rho1 = 1.0, asn = 0.0, rscale = 1.0;
    if (act && ok) {
      const double sq = r[0] * r[0] + r[1] * r[1];
      if (loss_type != 0) {
        double rho[3];
        loss_eval(loss_type, loss_width, sq, rho);
        acc[0] += 0.5 * rho[0];
        sqrt_rho1 = sqrt(rho[1]);
        rscale = sqrt_rho1;
        if (!(sq == 0.0 || rho[2] <= 0.0)) {
          const double Dd = 1.0 + 2.0 * sq * rho[2] / rho[1];
          const double alpha = 1.0 - sqrt(Dd);
          rscale = sqrt_rho1 / (1.0 - alpha);
          asn = alpha / sq;
        }
      } else {
        acc[0] += 0.5 * sq;
      }
      acc[1] += sq;
      double Js0[DP], Js1[DP];  // COMPACT: the stored (scaled) point block
#pragma unroll
      for (int a = 0; a < DP; ++a) {
        double j0 = 0.0, j1 = 0.0;
        if (!pconst) {
          j0 = (a < 3) ? (double)M[0][a < 3 ? a : 0] : (double)Jp3[0];
          j1 = (a < 3) ? (double)M[1][a < 3 ? a : 0] : (double)Jp3[1];
        }
        if (loss_type != 0) {
          const double rtj = j0 * r[0] + j1 * r[1];
          j0 = sqrt_rho1 * (j0 - asn * r[0] * rtj);
          j1 = sqrt_rho1 * (j1 - asn * r[1] * rtj);
        }
        PST(j0 * sp[a], &pmJp[pidx<2 * DP>((2 * a), e)]);
        PST(j1 * sp[a], &pmJp[pidx<2 * DP>((2 * a + 1), e)]);
        if constexpr (NORMS) {
          const double s0 = j0 * sp[a], s1 = j1 * sp[a];  // (what point_scale_kernel reads back from the planes)
          n2[a] += s0 * s0 + s1 * s1;
        }
        if constexpr (COMPACT) {
          Js0[a] = j0 * sp[a];
          Js1[a] = j1 * sp[a];
        }
      }
      PST(r[0] * rscale, &pmR[pidx<2>(0, e)]);
      PST(r[1] * rscale, &pmR[pidx<2>(1, e)]);
      if constexpr (COMPACT) {
        const double r0s = r[0] * rscale, r1s = r[1] * rscale;
#pragma unroll
        for (int a = 0; a < DP; ++a) {
#pragma unroll
          for (int b = a; b < DP; ++b) Vt[sym_idx(a, b, DP)] += Js0[a] * Js0[b] + Js1[a] * Js1[b];
          gt[a] += Js0[a] * r0s + Js1[a] * r1s;
        }
      }
    }
    LIN_LAP(2);
    if constexpr (NORMS) {
      if (act && !ok) v.flags[FL_INVALID] = 1;
      continue;  // nothing of the camera block is stored: no second half record
    }
    if constexpr (COMPACT) {
      // the camera block is p_n (one plane pair of pm_A): no second half record either
      if (act) {
        if (!ok) {
          v.flags[FL_INVALID] = 1;
          for (int d = 0; d < 2 * DP; ++d) PST(0.0, &pmJp[pidx<2 * DP>(d, e)]);
          PST(0.0, &pmR[pidx<2>(0, e)]);
          PST(0.0, &pmR[pidx<2>(1, e)]);
        }
        if constexpr (ULOSS) {
          // a robust loss: the corrected point C p_n (C = sqrt(rho') (I - alpha r r^T / |r|^2), as on every column pair)
          // and, in a second pair, r^2 of the uncorrected one
          double q0 = pn0, q1 = pn1;
          if (loss_type != 0) {
            const double rtj = q0 * r[0] + q1 * r[1];
            q0 = sqrt_rho1 * (q0 - asn * r[0] * rtj);
            q1 = sqrt_rho1 * (q1 - asn * r[1] * rtj);
          }
          PST(ok ? q0 : 0.0, &pmA[pidx<4>(0, e)]);
          PST(ok ? q1 : 0.0, &pmA[pidx<4>(1, e)]);
          PST(ok ? pn0 * pn0 + pn1 * pn1 : 0.0, &pmA[pidx<4>(2, e)]);
          PST(0.0, &pmA[pidx<4>(3, e)]);
        } else {
          PST(ok ? pn0 : 0.0, &pmA[pidx<2>(0, e)]);
          PST(ok ? pn1 : 0.0, &pmA[pidx<2>(1, e)]);
        }
      }
      continue;
    }
    stage_camera_records(prep, kStageWords, cam, st, lane);
    LIN_LAP(3);
    // ---- phase B: P[0..8] = Jl diag(scale_w), P[9..11] = position scales, P[12..21] = intrinsics scales
    if (!act) continue;
    if (!ok) {
      v.flags[FL_INVALID] = 1;
      for (int d = 0; d < 2 * D; ++d) PST(0.0, &pmA[pidx<2 * D>(d, e)]);
      if (SH)
        for (int d = 0; d < 2 * D; ++d) PST(0.0, &pmA1[pidx<2 * D>(d, e)]);
      for (int d = 0; d < 2 * DP; ++d) PST(0.0, &pmJp[pidx<2 * DP>(d, e)]);
      PST(0.0, &pmR[pidx<2>(0, e)]);
      PST(0.0, &pmR[pidx<2>(1, e)]);
      continue;
    }
    const double wneg = -X[3];
    const unsigned mask = (unsigned)rec.w;
    // reduced camera block: free columns of [ext(6) | intr(10)], compacted
    int dst = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (mask & (1u << c)) {
        double j0, j1, scl;
        if (c < 3) {
          j0 = wneg * (double)M[0][c < 3 ? c : 0];
          j1 = wneg * (double)M[1][c < 3 ? c : 0];
          scl = P[9 + (c < 3 ? c : 0)];
        } else if (c < 6) {
          const int kk = (c >= 3 && c < 6) ? c - 3 : 0;
          j0 = (double)(cx[0][0] * (RT)P[kk] + cx[0][1] * (RT)P[3 + kk] + cx[0][2] * (RT)P[6 + kk]);
          j1 = (double)(cx[1][0] * (RT)P[kk] + cx[1][1] * (RT)P[3 + kk] + cx[1][2] * (RT)P[6 + kk]);
          scl = 1.0;  // folded into Jl
        } else {
          j0 = (double)Jint[0][c >= 6 ? c - 6 : 0];
          j1 = (double)Jint[1][c >= 6 ? c - 6 : 0];
          scl = P[12 + (c >= 6 ? c - 6 : 0)];
        }
        if (loss_type != 0) {
          const double rtj = j0 * r[0] + j1 * r[1];
          j0 = sqrt_rho1 * (j0 - asn * r[0] * rtj);
          j1 = sqrt_rho1 * (j1 - asn * r[1] * rtj);
        }
        if (!(c < 3 && !SH && (UDROP || v.drop_pos))) {  // (drop_pos: the position columns are not stored, device_view.h)
          PST(j0 * scl, &pmA[pidx<2 * D>((2 * dst), e)]);
          PST(j1 * scl, &pmA[pidx<2 * D>((2 * dst + 1), e)]);
        }
        ++dst;
      }
    }
    for (int d = 2 * dst; d < 2 * D; ++d) PST(0.0, &pmA[pidx<2 * D>(d, e)]);
    if (SH) {
      // free intrinsics shared between views: their columns go to the group's own block
      const int grb = v.cam_grb[cam];
      int dst1 = 0;
      if (grb >= 0) {
        const unsigned gmask = v.grp_mask[v.cam_grp[cam]];
#pragma unroll
        for (int c = 0; c < 10; ++c) {
          if (gmask & (1u << c)) {
            double j0 = (double)Jint[0][c], j1 = (double)Jint[1][c];
            if (loss_type != 0) {
              const double rtj = j0 * r[0] + j1 * r[1];
              j0 = sqrt_rho1 * (j0 - asn * r[0] * rtj);
              j1 = sqrt_rho1 * (j1 - asn * r[1] * rtj);
            }
            const double scl = P[12 + c];
            PST(j0 * scl, &pmA1[pidx<2 * D>((2 * dst1), e)]);
            PST(j1 * scl, &pmA1[pidx<2 * D>((2 * dst1 + 1), e)]);
            ++dst1;
          }
        }
      }
      for (int d = 2 * dst1; d < 2 * D; ++d) PST(0.0, &pmA1[pidx<2 * D>(d, e)]);
    }
    LIN_LAP(4);
  }
#ifdef TMI_LIN_PROFILE
  if ((threadIdx.x & 63) == 0) {
    for (int i = 0; i < 5; ++i) atomicAdd(&g_lin_prof[i], (unsigned long long)lp_[i]);
    atomicAdd(&g_lin_prof[5], (unsigned long long)tm.trips);
    atomicAdd(&g_lin_prof[6], 1ull);
  }
#endif
  if constexpr (NORMS) {
#pragma unroll
    for (int a = 0; a < DP; ++a) {
      const double t = group_sum(n2[a], tm.wide);
      if (tm.valid && tm.leader) v.scale_p[(size_t)lp * DP + a] = 1.0 / (1.0 + sqrt(t));
    }
  }
  if constexpr (COMPACT) {
    const size_t NPc = (size_t)v.Np_pad;
#pragma unroll
    for (int i = 0; i < sym_size(DP); ++i) {
      const double t = group_sum(Vt[i], tm.wide);
      if (tm.valid && tm.leader && k > 0) v.Vraw[(size_t)i * NPc + lp] = t;
    }
#pragma unroll
    for (int a = 0; a < DP; ++a) {
      const double t = group_sum(gt[a], tm.wide);
      if (tm.valid && tm.leader && k > 0) v.gp[(size_t)a * NPc + lp] = t;
    }
  }
  block_sum_finish<2>(acc, v.partial, nblocks, v.ticket + 2 * kTicketStride, sums);
}

// drop_pos: xs = x with the position entries of every view block times the block's column scales (device_view.h)

template <int D>
__global__ __launch_bounds__(256) void pos_scale_kernel(DeviceView v, const double* __restrict__ x, double* __restrict__ xs) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if constexpr (D == 9) {
    if (v.compact) {  // compact planes: the transformed blocks, a thread per view
      if (i < v.Nrb) {
        double xv[9];
#pragma unroll
        for (int a = 0; a < 9; ++a) xv[a] = x[(size_t)i * 9 + a];
        compact_forward_view(v, i, xv, xs);
      }
      return;
    }
  }
  if (i >= v.Nrb * D) return;
  const int a = i % D;
  xs[i] = a < 3 ? x[i] * v.scale_c[i] : x[i];
}

// cost only, at a prepared parameter set (kernel class 9): hot loop 1, residual-only.
template <int DP, typename RT>
__global__ __launch_bounds__(256) void cost_kernel(DeviceView v, const double* __restrict__ prep,
                                                   const double* __restrict__ pts, int loss_type,
                                                   double loss_width, int flag_slot, int nblocks,
                                                   double* partial, double* __restrict__ sums,
                                                   double* __restrict__ flag_dst) {
  __shared__ __attribute__((aligned(16))) double stage[kSlicesPerBlock][64 * kStagePitch];
  const TrackMap tm = track_map(v);
  const int lane = threadIdx.x & 63;
  double* st = stage[threadIdx.x >> 6];
  const double* P = st + lane * kStagePitch;
  double acc[2] = {0.0, 0.0};
  const int k = tm.k;
  double X[4] = {0.0, 0.0, 0.0, 1.0};
  if (tm.valid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) X[i] = pts[(size_t)tm.lp * 4 + i];
  }
  int cam_next = (tm.j0 < k) ? v.obs_cam[tm.base + (size_t)tm.j0 * 64] : -1;
  for (int trip = 0; trip < tm.trips; ++trip) {
    const int j = tm.j0 + trip * tm.jstep;
    const bool act = j < k;
    const size_t e = tm.base + (size_t)j * 64;
    const int cam = act ? cam_next : -1;
    if (j + tm.jstep < k) cam_next = v.obs_cam[e + (size_t)tm.jstep * 64];
    int model = 0;
    double fx = 0.0, fy = 0.0;
    if (act) {
      model = v.cam_rec[cam].x;
      const double2 f2 = *reinterpret_cast<const double2*>(v.obs_xy + 2 * e);
      fx = f2.x;
      fy = f2.y;
    }
    stage_camera_records(prep, 0, cam, st, lane);
    if (!act) continue;
    RT rr[2];
    RT (*nul6)[6] = nullptr;
    RT Jint[2][10];
    RT (*nul4)[4] = nullptr;
    const bool ok = reprojection_error_prepared<false, RT>(model, P, X, fx, fy, rr, nul6, Jint, nul4);
    if (!ok) {
      st_agent(&v.flags[flag_slot], 1);
      continue;
    }
    const double r[2] = {(double)rr[0], (double)rr[1]};
    const double sq = r[0] * r[0] + r[1] * r[1];
    if (loss_type != 0) {
      double rho[3];
      loss_eval(loss_type, loss_width, sq, rho);
      acc[0] += 0.5 * rho[0];
    } else {
      acc[0] += 0.5 * sq;
    }
    acc[1] += sq;
  }
  block_sum_finish<2>(acc, partial, nblocks, v.ticket + 2 * kTicketStride, sums, v.flags + flag_slot, flag_dst);
}

// ------------------------------------------------------------------------------
// Jacobi scaling (Ceres jacobi_scaling): 1 / (1 + ||column||), computed once
// from the unscaled Jacobian at the start point.
// ------------------------------------------------------------------------------
template <int DP, typename PT = double>
__global__ __launch_bounds__(256) void point_scale_kernel(DeviceView v) {
  const TrackMap tm = track_map(v);
  PT* const pmR = reinterpret_cast<PT*>(v.pm_r);   // (the planes in their storage type: double, or float with
  PT* const pmA = reinterpret_cast<PT*>(v.pm_A);   //  fp32 evaluation on shared-intrinsics problems -- DeviceView::planes_fp32)
  PT* const pmA1 = reinterpret_cast<PT*>(v.pm_A1);
  PT* const pmJp = reinterpret_cast<PT*>(v.pm_Jp);
  (void)pmR; (void)pmA; (void)pmA1; (void)pmJp;
  if (!tm.valid) return;
  const int lp = tm.lp;
  const int k = tm.k;
  const size_t base = tm.base;
  const size_t N = (size_t)v.No_pad;
  double n2[DP];
#pragma unroll
  for (int a = 0; a < DP; ++a) n2[a] = 0.0;
  for (int j = tm.j0; j < k; j += tm.jstep) {
    const size_t e = base + (size_t)j * 64;
#pragma unroll
    for (int a = 0; a < DP; ++a) {
      const double j0 = pmJp[pidx<2 * DP>((2 * a), e)], j1 = pmJp[pidx<2 * DP>((2 * a + 1), e)];
      n2[a] += j0 * j0 + j1 * j1;
    }
  }
#pragma unroll
  for (int a = 0; a < DP; ++a) {
    const double t = group_sum(n2[a], tm.wide);
    if (tm.leader) v.scale_p[(size_t)lp * DP + a] = 1.0 / (1.0 + sqrt(t));
  }
}

// scale_c holds the (all-reduced) squared column norms U_aa of the camera side, taken
// from a first unscaled pass of point_eliminate + camera_diag at the start point.
__global__ void camera_scale_finish_kernel(double* scale_c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) scale_c[i] = 1.0 / (1.0 + sqrt(scale_c[i]));
}

// ------------------------------------------------------------------------------
// point_eliminate (kernel class 1): per track V = sum Jp^T Jp, g_p, LM damping,
// Cholesky of the DP x DP block in registers, t_p = (V+Dp)^-1 g_p; then per
// observation Y = A^T (Jp L^-T) and the reduced residual r~ = r - Jp t_p, written
// to the camera-major records.  This is the e-block half of Ceres'
// SchurEliminator (hot loop 2) restricted to what the camera side needs.
// partial: [gmax_p] (max) per block.
// ------------------------------------------------------------------------------
// REC = false (direct_diag.h, DeviceView::direct_diag): the per-track part only -- no camera-major records, no LDS
// staging, a quarter of the registers, so that the sweep over the Jp and r planes runs at full occupancy.
template <int D, int DP, bool SH, bool REC = true, typename PT = double>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void point_eliminate_kernel(DeviceView v, double inv_radius,
                                                              double lm_lo, double lm_hi, int nblocks,
                                                              double* partial_max, double* singular_vote,
                                                              double grad_tol, double* grad_vote) {
  constexpr int NS = sym_size(DP);
  PT* const pmR = reinterpret_cast<PT*>(v.pm_r);   // (the planes in their storage type: double, or float with
  PT* const pmA = reinterpret_cast<PT*>(v.pm_A);   //  fp32 evaluation on shared-intrinsics problems -- DeviceView::planes_fp32)
  PT* const pmA1 = reinterpret_cast<PT*>(v.pm_A1);
  PT* const pmJp = reinterpret_cast<PT*>(v.pm_Jp);
  (void)pmR; (void)pmA; (void)pmA1; (void)pmJp;
  constexpr int YS = ys_of(D, DP);
  constexpr int ASA = asa_of(D, DP);
  constexpr int AS = SH ? as_of(D, true) : ASA + kRecTail;  // staged A record: [A rows | Q | pad][N r~ r] without
                                                          // shared blocks, the one record of as_of otherwise
  constexpr int TO = SH ? 2 * D : ASA;                    // where {N, r~, r} start in it
  constexpr int STP = (YS > AS ? YS : AS) + 2;  // LDS record pitch, +2 doubles: conflict-free b64/b128
  __shared__ __attribute__((aligned(16))) double stage[REC ? kSlicesPerBlock : 1][REC ? 32 : 1][REC ? STP : 2];
  __shared__ int stage_cpos[REC ? kSlicesPerBlock : 1][REC ? 32 : 1];
  constexpr int TRS = DP == 3 ? 16 : 24;  // doubles of a track record (direct_diag.h, trk_stride)
  constexpr int TPITCH = TRS + 2;
  __shared__ __attribute__((aligned(16))) double tstage[REC ? 1 : kSlicesPerBlock][REC ? 2 : 64 * TPITCH];
  const int lane = threadIdx.x & 63;
  const TrackMap tm = track_map(v);
  double gmax = 0.0;
  if (tm.valid) {
    const int lp = tm.lp;
    const int k = tm.k;
    const size_t base = tm.base;
    const size_t N = (size_t)v.No_pad;
    const size_t NP = (size_t)v.Np_pad;
    bool have_tp = false;
    double tp_s[DP], Li_s[DP][DP];
    double trec[REC ? 2 : TRS];  // (REC = false) the track's record for the view-by-view camera side
#pragma unroll
    for (int i = 0; i < (REC ? 2 : TRS); ++i) trec[i] = 0.0;
    if (k > 0) {
      double V[NS], g[DP];
#pragma unroll
      for (int i = 0; i < NS; ++i) V[i] = 0.0;
#pragma unroll
      for (int a = 0; a < DP; ++a) g[a] = 0.0;
      const bool ready = !REC && v.sums_ready;  // (the compact linearize left V and g_p: no sweep over the planes)
      for (int j = tm.j0; j < k && !ready; j += tm.jstep) {
        const size_t e = base + (size_t)j * 64;
        double J0[DP], J1[DP];
#pragma unroll
        for (int a = 0; a < DP; ++a) {
          J0[a] = pmJp[pidx<2 * DP>((2 * a), e)];
          J1[a] = pmJp[pidx<2 * DP>((2 * a + 1), e)];
        }
        const double r0 = pmR[pidx<2>(0, e)], r1 = pmR[pidx<2>(1, e)];
#pragma unroll
        for (int a = 0; a < DP; ++a) {
#pragma unroll
          for (int b = a; b < DP; ++b) V[sym_idx(a, b, DP)] += J0[a] * J0[b] + J1[a] * J1[b];
          g[a] += J0[a] * r0 + J1[a] * r1;
        }
      }
      // wide slices: finish the sums over the 16 lanes that share the track; every lane of
      // the group then factors the same DP x DP block (redundantly), the leader stores it
#pragma unroll
      for (int i = 0; i < NS; ++i) V[i] = group_sum(V[i], tm.wide);
#pragma unroll
      for (int a = 0; a < DP; ++a) g[a] = group_sum(g[a], tm.wide);
      if (ready) {
#pragma unroll
        for (int i = 0; i < NS; ++i) V[i] = v.Vraw[(size_t)i * NP + lp];
#pragma unroll
        for (int a = 0; a < DP; ++a) g[a] = v.gp[(size_t)a * NP + lp];
      }
      // the undamped V = Jp^T Jp is kept for the model cost change (back_substitute_kernel)
      if (tm.leader && !ready) {
#pragma unroll
        for (int i = 0; i < NS; ++i) v.Vraw[(size_t)i * NP + lp] = V[i];
      }
      // LM damping and Cholesky V + Dp = L L^T (L lower, packed by rows into Lm[a][b], b <= a)
      double Lm[DP][DP];
      bool pd = true;
#pragma unroll
      for (int a = 0; a < DP; ++a) {
        const double d = V[sym_idx(a, a, DP)];
        V[sym_idx(a, a, DP)] = d + fmin(fmax(d, lm_lo), lm_hi) * inv_radius;
        if (tm.leader && !ready) v.gp[(size_t)a * NP + lp] = g[a];
        gmax = fmax(gmax, fabs(g[a] / v.scale_p[(size_t)lp * DP + a]));
      }
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        double d = V[sym_idx(j, j, DP)];
#pragma unroll
        for (int m = 0; m < j; ++m) d -= Lm[j][m] * Lm[j][m];
        if (!(d > 0.0)) {
          pd = false;
          d = 1.0;
        }
        const double l = sqrt(d);
        Lm[j][j] = l;
        const double il = 1.0 / l;
#pragma unroll
        for (int i = j + 1; i < DP; ++i) {
          double t = V[sym_idx(j, i, DP)];
#pragma unroll
          for (int m = 0; m < j; ++m) t -= Lm[i][m] * Lm[j][m];
          Lm[i][j] = t * il;
        }
      }
      if (!pd) {
        v.flags[FL_SINGULAR_POINT] = 1;
        *singular_vote = 1.0;  // lives in the all-reduced scalar tail: every rank sees it
      }
      // Li = L^-1 (lower)
      double Li[DP][DP];
#pragma unroll
      for (int j = 0; j < DP; ++j) {
        Li[j][j] = 1.0 / Lm[j][j];
#pragma unroll
        for (int i = j + 1; i < DP; ++i) {
          double t = 0.0;
#pragma unroll
          for (int m = j; m < i; ++m) t -= Lm[i][m] * Li[m][j];
          Li[i][j] = t / Lm[i][i];
        }
      }
      // L^-1 itself serves the matrix-free product (zhat = L^-1 sum Jp^T u, implicit_tracks_q_kernel)
      if (tm.leader) {
#pragma unroll
        for (int a = 0; a < DP; ++a)
#pragma unroll
          for (int b = a; b < DP; ++b) v.Linv[(size_t)sym_idx(a, b, DP) * NP + lp] = Li[b][a];
      }
      // Vinv = Li^T Li (symmetric), t_p = Vinv g
      double Vi[NS], tp[DP];
#pragma unroll
      for (int a = 0; a < DP; ++a)
#pragma unroll
        for (int b = a; b < DP; ++b) {
          double t = 0.0;
#pragma unroll
          for (int m = b; m < DP; ++m) t += Li[m][a] * Li[m][b];
          Vi[sym_idx(a, b, DP)] = t;
          if (tm.leader) v.Vinv[(size_t)sym_idx(a, b, DP) * NP + lp] = t;
        }
#pragma unroll
      for (int a = 0; a < DP; ++a) {
        double t = 0.0;
#pragma unroll
        for (int b = 0; b < DP; ++b) t += Vi[a <= b ? sym_idx(a, b, DP) : sym_idx(b, a, DP)] * g[b];
        tp[a] = t;
      }
      have_tp = true;
#pragma unroll
      for (int a = 0; a < DP; ++a) {
        tp_s[a] = tp[a];
#pragma unroll
        for (int b = 0; b <= a; ++b) Li_s[a][b] = Li[a][b];
      }
      if (!REC) {
        // direct_diag.h: the camera side re-evaluates the observations view by view and needs, per track,
        // { X, L^-1 diag(scale_p), scale_p . t_p } in one record (scale 0: constant point, its Jp is zero)
        const bool pc = v.pt_const[lp] != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) trec[i] = v.pts[(size_t)lp * 4 + i];
#pragma unroll
        for (int a = 0; a < DP; ++a) {
          const double sa = pc ? 0.0 : v.scale_p[(size_t)lp * DP + a];
          trec[4 + NS + a] = tp[a] * sa;
#pragma unroll
          for (int b = a; b < DP; ++b) trec[4 + sym_idx(a, b, DP)] = Li[b][a] * sa;
        }
      }
    }
    if (!REC) {
      if (tm.wide == 0) {
        // a wavefront's 64 records are consecutive in memory: through LDS, so that consecutive lanes store consecutive
        // 16 bytes (a lane storing its own 128-byte record 16 bytes at a time writes partial sectors)
        double* ts = &tstage[threadIdx.x >> 6][0];
#pragma unroll
        for (int i = 0; i < TRS; i += 2)
          *reinterpret_cast<double2*>(ts + lane * TPITCH + i) = make_double2(trec[i], trec[i + 1]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double* out = v.trk_rec + (size_t)tm.s * 64 * TRS;
        for (int c = lane; c < 64 * (TRS / 2); c += 64) {
          const int rc = c / (TRS / 2), part = c - rc * (TRS / 2);
          *reinterpret_cast<double2*>(out + (size_t)rc * TRS + 2 * part) = *reinterpret_cast<const double2*>(ts + rc * TPITCH + 2 * part);
        }
      } else if (tm.leader && k > 0) {
        double* out = v.trk_rec + (size_t)lp * TRS;
#pragma unroll
        for (int i = 0; i < TRS; i += 2) *reinterpret_cast<double2*>(out + i) = make_double2(trec[i], trec[i + 1]);
      }
    }
    // ---- second pass: Y = A^T (Jp L^-T), r~ = r - Jp t_p into the camera-major records.
    // Each lane builds its record in registers; the wave then stages 32 records at a
    // time in LDS and writes them out with consecutive lanes covering consecutive 16 B,
    // i.e. whole 64-byte sectors per record (scattered 8/16-byte stores cost 2-4x the
    // bytes in HBM write traffic, profiles/r01_a).  The trip count K is wave uniform.
    double* st = &stage[REC ? threadIdx.x >> 6 : 0][0][0];
    int* scp = &stage_cpos[REC ? threadIdx.x >> 6 : 0][0];
    double Yg[SH ? YS : 1];  // running sum of the shared block's Y over the current run
#pragma unroll
    for (int i = 0; i < (SH ? YS : 1); ++i) Yg[i] = 0.0;
    // (direct_diag: no camera-major records at all -- nothing but camera_diag would read them in a matrix-free iteration)
    const int rec_trips = REC ? tm.trips : 0;
    for (int trip = 0; trip < rec_trips; ++trip) {
      const int j = tm.j0 + trip * tm.jstep;
      const size_t e = base + (size_t)j * 64;
      int cpos = -1, gslot = -1, gflag = 0;
      if (have_tp && j < k) {
        cpos = v.obs_cpos[e];
        if (SH) {
          gslot = v.obs_gslot[e];
          gflag = v.obs_gflag[e];
        }
      }
      double Yv[YS], Av[AS];
      if (cpos >= 0) {
        double J0[DP], J1[DP], Q0[DP], Q1[DP];
#pragma unroll
        for (int a = 0; a < DP; ++a) {
          J0[a] = pmJp[pidx<2 * DP>((2 * a), e)];
          J1[a] = pmJp[pidx<2 * DP>((2 * a + 1), e)];
        }
        const double r0 = pmR[pidx<2>(0, e)], r1 = pmR[pidx<2>(1, e)];
        double rt0 = r0, rt1 = r1;
#pragma unroll
        for (int a = 0; a < DP; ++a) {
          rt0 -= J0[a] * tp_s[a];
          rt1 -= J1[a] * tp_s[a];
        }
        // Q_row = Li * Jp_row^T
#pragma unroll
        for (int b = 0; b < DP; ++b) {
          double q0 = 0.0, q1 = 0.0;
#pragma unroll
          for (int a = 0; a <= b; ++a) {
            q0 += Li_s[b][a] * J0[a];
            q1 += Li_s[b][a] * J1[a];
          }
          Q0[b] = q0;
          Q1[b] = q1;
        }
        double pc[3] = {0.0, 0.0, 0.0};
        if (!SH && v.drop_pos) {
          const int rb = v.obs_rb[e];
#pragma unroll
          for (int a = 0; a < 3; ++a) pc[a] = v.pos_coef[(size_t)a * NP + lp] * v.scale_c[(size_t)(rb < 0 ? 0 : rb) * D + a];
        }
#pragma unroll
        for (int a = 0; a < D; ++a) {
          double a0, a1;
          if (!SH && a < 3 && v.drop_pos) {  // the position columns are formed from Jp (device_view.h)
            a0 = J0[a < DP ? a : 0] * pc[a < 3 ? a : 0];
            a1 = J1[a < DP ? a : 0] * pc[a < 3 ? a : 0];
          } else {
            a0 = pmA[pidx<2 * D>((2 * a), e)];
            a1 = pmA[pidx<2 * D>((2 * a + 1), e)];
          }
          Av[a] = a0;
          Av[D + a] = a1;
#pragma unroll
          for (int b = 0; b < DP; ++b) Yv[a * DP + b] = a0 * Q0[b] + a1 * Q1[b];
        }
#pragma unroll
        for (int i = D * DP; i < YS; ++i) Yv[i] = 0.0;
        if (!SH) {
          double n00 = 1.0, n01 = 0.0, n11 = 1.0;
#pragma unroll
          for (int b = 0; b < DP; ++b) {
            n00 -= Q0[b] * Q0[b];
            n01 -= Q0[b] * Q1[b];
            n11 -= Q1[b] * Q1[b];
          }
          Av[TO] = n00;
          Av[TO + 1] = n01;
          Av[TO + 2] = n11;
          Av[TO + 3] = rt0;
          Av[TO + 4] = rt1;
          Av[TO + 5] = r0;
          Av[TO + 6] = r1;
        } else {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            Av[sh_off_q(D) + b] = b < DP ? Q0[b < DP ? b : 0] : 0.0;
            Av[sh_off_q(D) + 4 + b] = b < DP ? Q1[b < DP ? b : 0] : 0.0;
          }
          Av[sh_off_rt(D)] = rt0;
          Av[sh_off_rt(D) + 1] = rt1;
          Av[sh_off_r(D)] = r0;
          Av[sh_off_r(D) + 1] = r1;
        }
        if (!SH) {
#pragma unroll
          for (int b = 0; b < DP; ++b) {
            Av[2 * D + b] = Q0[b];
            Av[2 * D + DP + b] = Q1[b];
          }
#pragma unroll
          for (int i = 2 * D + 2 * DP; i < ASA; ++i) Av[i] = 0.0;
          Av[TO + 7] = 0.0;
        }
        if (SH) {
          // shared intrinsics block: Y1 = A1^T Q summed over the track's observations of
          // that block (they are adjacent); A1 rides in the A record for the per-view sums
          if (gflag & 1) {
#pragma unroll
            for (int i = 0; i < YS; ++i) Yg[i] = 0.0;
          }
#pragma unroll
          for (int a = 0; a < D; ++a) {
            double a0 = 0.0, a1 = 0.0;
            if (gslot >= 0) {
              a0 = pmA1[pidx<2 * D>((2 * a), e)];
              a1 = pmA1[pidx<2 * D>((2 * a + 1), e)];
            }
            Av[sh_off_a1(D) + a] = a0;
            Av[sh_off_a1(D) + D + a] = a1;
#pragma unroll
            for (int b = 0; b < DP; ++b) Yg[a * DP + b] += a0 * Q0[b] + a1 * Q1[b];
          }
#pragma unroll
          for (int i = 4 * D + 12; i < AS; ++i) Av[i] = 0.0;
        }
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        // Y records of lanes [32 half, 32 half + 32) -- only the explicit Schur complement (and the
        // shared-block sums) read them; the matrix-free operator works from the A records
        if ((lane >> 5) == half) {
          scp[lane & 31] = cpos;
          if (cpos >= 0 && v.write_y) {
#pragma unroll
            for (int i = 0; i < YS; i += 2)
              *reinterpret_cast<double2*>(st + (lane & 31) * STP + i) = make_double2(Yv[i], Yv[i + 1]);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (v.write_y) {
          for (int c = lane; c < 32 * (YS / 2); c += 64) {
            const int rec = c / (YS / 2), part = c - rec * (YS / 2);
            const int cp = scp[rec];
            if (cp >= 0)
              store_nt(v.cm_Y + (size_t)cp * YS + 2 * part, st + rec * STP + 2 * part);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // A records of the same lanes
        if ((lane >> 5) == half && cpos >= 0) {
#pragma unroll
          for (int i = 0; i < AS; i += 2)
            *reinterpret_cast<double2*>(st + (lane & 31) * STP + i) = make_double2(Av[i], Av[i + 1]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (SH) {
          for (int c = lane; c < 32 * (AS / 2); c += 64) {
            const int rec = c / (AS / 2), part = c - rec * (AS / 2);
            const int cp = scp[rec];
            if (cp >= 0)
              rec_store2_nt<PT>(v.cm_A, (size_t)cp * AS + 2 * part, st + rec * STP + 2 * part);  // (SH: PT is the records' type too)
          }
        } else {
          // [A rows | Q] (whole sectors) and the tail
          for (int c = lane; c < 32 * (ASA / 2); c += 64) {
            const int rec = c / (ASA / 2), part = c - rec * (ASA / 2);
            const int cp = scp[rec];
            if (cp >= 0)
              store_nt(v.cm_A + (size_t)cp * ASA + 2 * part, st + rec * STP + 2 * part);
          }
          for (int c = lane; c < 32 * (kRecTail / 2); c += 64) {
            const int rec = c / (kRecTail / 2), part = c - rec * (kRecTail / 2);
            const int cp = scp[rec];
            if (cp >= 0)
              store_nt(v.cm_R + (size_t)cp * kRecTail + 2 * part, st + rec * STP + ASA + 2 * part);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (SH) {
          // completed (track, shared block) sums of the same lanes
          const bool emit = cpos >= 0 && gslot >= 0 && (gflag & 2);
          if ((lane >> 5) == half) {
            scp[lane & 31] = emit ? gslot : -1;
            if (emit) {
#pragma unroll
              for (int i = 0; i < YS; i += 2)
                *reinterpret_cast<double2*>(st + (lane & 31) * STP + i) = make_double2(Yg[i], Yg[i + 1]);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          for (int c = lane; c < 32 * (YS / 2); c += 64) {
            const int rec = c / (YS / 2), part = c - rec * (YS / 2);
            const int cp = scp[rec];
            if (cp >= 0)
              store_nt(v.cm_Y + (size_t)cp * YS + 2 * part, st + rec * STP + 2 * part);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      }
    }
  }
  // block max of the point gradient
  __shared__ double shm[kSlicesPerBlock];
  const double wm = wave_max(gmax);
  if (lane == 0) shm[threadIdx.x >> 6] = wm;
  __syncthreads();
  __shared__ int pe_last;
  if (threadIdx.x == 0) {
    double m = shm[0];
    for (int i = 1; i < kSlicesPerBlock; ++i) m = fmax(m, shm[i]);
    st_agent(&partial_max[blockIdx.x], m);
    pe_last = take_ticket(v.ticket + 2 * kTicketStride, nblocks) ? 1 : 0;
  }
  __syncthreads();
  if (pe_last) {
    // the last workgroup finishes max |g_p / scale| (what reduce_max_kernel did in a launch of its own)
    double m = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) m = fmax(m, ld_agent(&partial_max[i]));
    const double wm2 = wave_max(m);
    __syncthreads();
    if (lane == 0) shm[threadIdx.x >> 6] = wm2;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = shm[0];
      for (int i = 1; i < kSlicesPerBlock; ++i) t = fmax(t, shm[i]);
      v.scal[SC_GMAX_P] = t;
      // sharded solves: this rank's vote in the gradient-tolerance test rides in the scalar tail of the reduced
      // system and is summed by the all-reduce that follows (the camera part of the gradient is all-reduced
      // itself, so its maximum needs no vote)
      if (grad_vote) *grad_vote = (t > grad_tol) ? 1.0 : 0.0;
    }
  }
}

// ------------------------------------------------------------------------------
// camera_diag (kernel class 2): one wavefront per reduced block walks that
// camera's camera-major records and reduces, in registers + wave shuffles,
//   S_cc(raw) = sum A^T N A (= sum A^T A - Y Y^T),  U diag = sum diag(A^T A),
//   g~ = sum A^T r~  (reduced gradient),  g_c = sum A^T r  (camera gradient).
// ------------------------------------------------------------------------------
// Raw diagonal of a SHARED intrinsics block, - sum Y Y^T over its (track, block) records, in two steps: a block
// shared by 200 views has 10^5-10^6 records and ONE wavefront walking them (camera_diag's way) takes milliseconds, so
// the records are cut into chunks of kSharedDiagChunk, a wavefront per chunk leaves its partial sum, and a second
// launch adds the partials in chunk order (fixed order: bit-reproducible) -- camera_diag then skips those blocks.
constexpr int kSharedDiagChunk = 4096;
template <int D, int DP>
__global__ __launch_bounds__(64) void shared_diag_partial_kernel(DeviceView v, int max_chunks, double* __restrict__ partial) {
  constexpr int NS = sym_size(D);
  constexpr int YS = ys_of(D, DP);
  const int g = blockIdx.x / max_chunks, ch = blockIdx.x - g * max_chunks;
  const int rb = v.Ncam_rb + g;
  const int s0 = v.cam_ptr[rb] + ch * kSharedDiagChunk;
  const int s1 = min(s0 + kSharedDiagChunk, v.cam_ptr[rb + 1]);
  if (s0 >= s1) return;
  double Ss[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) Ss[i] = 0.0;
  for (int s = s0 + threadIdx.x; s < s1; s += 64) {
    const double* yrec = v.cm_Y + (size_t)s * YS;
    double Y[YS];
#pragma unroll
    for (int i = 0; i < YS; i += 2) {
      const double2 t = *reinterpret_cast<const double2*>(yrec + i);
      Y[i] = t.x;
      Y[i + 1] = t.y;
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = a; b < D; ++b) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < DP; ++c) t -= Y[a * DP + c] * Y[b * DP + c];
        Ss[sym_idx(a, b, D)] += t;
      }
  }
  double* out = partial + (size_t)blockIdx.x * NS;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const double t = wave_sum(Ss[i]);
    if (threadIdx.x == 0) out[i] = t;
  }
}
template <int D>
__global__ __launch_bounds__(64) void shared_diag_reduce_kernel(DeviceView v, RedLayout L, int max_chunks,
                                                                const double* __restrict__ partial) {
  constexpr int NS = sym_size(D);
  const int g = blockIdx.x, rb = v.Ncam_rb + g;
  const int nch = (v.cam_ptr[rb + 1] - v.cam_ptr[rb] + kSharedDiagChunk - 1) / kSharedDiagChunk;
  double* diag = v.red + L.diag + (size_t)rb * D * D;
  for (int a = 0; a < D; ++a)
    for (int b = a + (int)threadIdx.x; b < D; b += 64) {
      double t = 0.0;
      for (int ch = 0; ch < nch; ++ch) t += partial[((size_t)g * max_chunks + ch) * NS + sym_idx(a, b, D)];
      diag[a * D + b] = t;
      diag[b * D + a] = t;
    }
}

template <int D, int DP, bool SH, typename RS = double>
__global__ __launch_bounds__(64) void camera_diag_kernel(DeviceView v, RedLayout L, int shared_elsewhere) {
  constexpr int NS = sym_size(D);
  constexpr int AS = as_of(D, SH);
  const int rb = blockIdx.x;
  double Ss[NS], Ud[D], gt[D], gc[D];
#pragma unroll
  for (int i = 0; i < NS; ++i) Ss[i] = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) Ud[a] = gt[a] = gc[a] = 0.0;
  if (SH && rb >= v.Ncam_rb) {
    // shared intrinsics block: its slots are (track, block) records whose Y is a SUM over the
    // track's observations of the block (point_eliminate), so Y Y^T has no per-observation
    // N form; the J^T J part arrives through group_reduce.  Raw diagonal = - sum Y Y^T
    // (by shared_diag_partial / shared_diag_reduce when the blocks are big: shared_elsewhere).
    constexpr int YS = ys_of(D, DP);
    for (int s = v.cam_ptr[rb] + threadIdx.x; s < v.cam_ptr[rb + 1] && !shared_elsewhere; s += 64) {
      const double* yrec = v.cm_Y + (size_t)s * YS;
      double Y[YS];
#pragma unroll
      for (int i = 0; i < YS; i += 2) {
        const double2 t = *reinterpret_cast<const double2*>(yrec + i);
        Y[i] = t.x;
        Y[i + 1] = t.y;
      }
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = a; b < D; ++b) {
          double t = 0.0;
#pragma unroll
          for (int c = 0; c < DP; ++c) t -= Y[a * DP + c] * Y[b * DP + c];
          Ss[sym_idx(a, b, D)] += t;
        }
    }
  } else
  for (int s = v.cam_ptr[rb] + threadIdx.x; s < v.cam_ptr[rb + 1]; s += 64) {
    double rec[2 * D + 8];
    if (SH) {
      // [A rows | Q | r~ | r | ...]: N = I - Q Q^T is formed here, the tail goes where the other layout has it
      const RS* arec = reinterpret_cast<const RS*>(v.cm_A) + (size_t)s * AS;
      double w12[12];
#pragma unroll
      for (int i = 0; i < 2 * D; i += 2) {
        const double2 t = rec_ld2(arec + i);
        rec[i] = t.x;
        rec[i + 1] = t.y;
      }
#pragma unroll
      for (int i = 0; i < 12; i += 2) {
        const double2 t = rec_ld2(arec + 2 * D + i);
        w12[i] = t.x;
        w12[i + 1] = t.y;
      }
      double n00 = 1.0, n01 = 0.0, n11 = 1.0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        n00 -= w12[b] * w12[b];
        n01 -= w12[b] * w12[4 + b];
        n11 -= w12[4 + b] * w12[4 + b];
      }
      rec[a_off_n(D)] = n00;
      rec[a_off_n(D) + 1] = n01;
      rec[a_off_n(D) + 2] = n11;
      rec[a_off_rt(D)] = w12[8];
      rec[a_off_rt(D) + 1] = w12[9];
      rec[a_off_r(D)] = w12[10];
      rec[a_off_r(D) + 1] = w12[11];
    } else {
      const double* arec = v.cm_A + (size_t)s * asa_of(D, DP);
      const double* trec = v.cm_R + (size_t)s * kRecTail;
#pragma unroll
      for (int i = 0; i < 2 * D; i += 2) {
        const double2 t = *reinterpret_cast<const double2*>(arec + i);
        rec[i] = t.x;
        rec[i + 1] = t.y;
      }
#pragma unroll
      for (int i = 0; i < kRecTail; i += 2) {
        const double2 t = *reinterpret_cast<const double2*>(trec + i);
        rec[2 * D + i] = t.x;
        rec[2 * D + i + 1] = t.y;
      }
    }
    const double n00 = rec[a_off_n(D)], n01 = rec[a_off_n(D) + 1], n11 = rec[a_off_n(D) + 2];
    const double rt0 = rec[a_off_rt(D)], rt1 = rec[a_off_rt(D) + 1];
    const double r0 = rec[a_off_r(D)], r1 = rec[a_off_r(D) + 1];
    double B0[D], B1[D];
#pragma unroll
    for (int a = 0; a < D; ++a) {
      B0[a] = n00 * rec[a] + n01 * rec[D + a];
      B1[a] = n01 * rec[a] + n11 * rec[D + a];
    }
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const double a0 = rec[a], a1 = rec[D + a];
#pragma unroll
      for (int b = a; b < D; ++b) Ss[sym_idx(a, b, D)] += a0 * B0[b] + a1 * B1[b];
      Ud[a] += a0 * a0 + a1 * a1;
      gt[a] += a0 * rt0 + a1 * rt1;
      gc[a] += a0 * r0 + a1 * r1;
    }
  }
  double* diag = v.red + L.diag + (size_t)rb * D * D;
#pragma unroll
  for (int a = 0; a < D; ++a) {
#pragma unroll
    for (int b = a; b < D; ++b) {
      const double t = wave_sum(Ss[sym_idx(a, b, D)]);
      if (threadIdx.x == 0) {
        diag[a * D + b] = t;
        diag[b * D + a] = t;
      }
    }
    const double u = wave_sum(Ud[a]), g1 = wave_sum(gt[a]), g2 = wave_sum(gc[a]);
    if (threadIdx.x == 0) {
      v.red[L.udiag + (size_t)rb * D + a] = u;
      v.red[L.gt + (size_t)rb * D + a] = g1;
      v.red[L.gc + (size_t)rb * D + a] = g2;
    }
  }
}

// ------------------------------------------------------------------------------
// schur_offdiag (kernel class 3): S_ij = - sum_{tracks seen by both} Y_i Y_j^T.
// One wavefront per structurally non-zero upper block; the pair list is a flat
// K = DP * #pairs contraction fed to v_mfma_f64_16x16x4_f64:
//   A[i][k] = Y_i[i][k % DP] of pair k / DP,  B[k][j] = Y_j[j][k % DP].
// Lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; the 16 x 16 result
// has row = (l >> 4) + 4 reg, col = l & 15 (f64 MFMA layout).  No atomics, the
// summation order is the pair list's (bit-reproducible).
// ------------------------------------------------------------------------------
constexpr int kSchurBlocksPerWave = 4;


// schur_offdiag, LDS-staged version (default).  Same launch slots, headers, pair lists, MFMA
// contraction and summation order as the gather kernel above; what changes is how the Y
// records reach the matrix cores.  There every lane fetched single doubles of "its" matrix
// element straight from global memory (36 useful 8-byte lanes per load instruction, 24 load
// instructions per 16 pairs, every instruction a fresh set of L1 tag look-ups).  Here the wave
// copies the 2 x 16 records of a chunk of 16 pairs into LDS with 16-byte loads in which
// consecutive lanes cover consecutive parts of a record (7 load instructions per 16 pairs for
// 9 x 3 records), and the MFMA operands are read from LDS.  The loads of chunk t + 1 are in flight
// in registers and the pair indices of chunk t + 2 are being fetched while chunk t is contracted.
//
// The chunk sequence of a wave runs across its R launch slots; its state is two SGPRs (slot r,
// first pair c0) and the slot headers stay in the lanes they were loaded into (lane r = slot r,
// v_readlane with a scalar lane select).  Records of pairs past the end of a block are zero-filled
// in LDS, so the contraction runs whole groups of four MFMA steps without per-step branches and
// the operand reads of a group are issued together.  (An earlier version kept the headers in
// scalar arrays picked by select chains and branched around every MFMA step: it spent 1.36 ms of
// its 1.82 ms on venice1778_heavy with the record loads switched OFF, profiles/r02_l.)
constexpr int kSchurPairsPerChunk = 16;

template <int D, int DP>
__global__ __launch_bounds__(256) void schur_offdiag_kernel(DeviceView v, RedLayout L) {
  constexpr int YS = ys_of(D, DP);
  constexpr int R = kSchurBlocksPerWave;
  constexpr int PC = kSchurPairsPerChunk;
  constexpr int PARTS = (D * DP + 1) / 2;  // 16-byte parts of a record that carry data
  constexpr int PITCH = ys_of(D, DP) + 2;          // doubles; keeps the 16-byte parts aligned
  constexpr int NL = (2 * PC * PARTS + 63) / 64;   // 16-byte loads per lane per chunk
  constexpr int STEPS = PC * DP / 4;               // MFMA steps (K = 4 each) of a full chunk
  constexpr int GS = 4;                            // MFMA steps per group (uniform branch per group)
  static_assert((PC * DP) % 4 == 0 && STEPS % GS == 0, "chunk K must be whole groups of MFMA steps");
  static_assert(R <= 64, "slot headers live one per lane");
  __shared__ __attribute__((aligned(16))) double lds_all[4][2 * PC * PITCH];
  const int lane = threadIdx.x & 63;
  double* lds = lds_all[threadIdx.x >> 6];
  const long long first = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (first >= v.n_order) return;  // wave-uniform; no workgroup barrier below
  // lane r holds launch slot r: {block, #pairs, first pair lo, first pair hi}
  int4 hq = make_int4(-1, 0, 0, 0);
  if (lane < R && first + lane < v.n_order) hq = reinterpret_cast<const int4*>(v.ub_order)[first + lane];
  // chunks a slot contributes, in pairs: a block without pairs on this rank (sharded solves) still
  // has ONE chunk, of zero pairs, so that its zeros get written
  const int vspan = hq.x >= 0 ? max(hq.y, 1) : 0;
  auto slot_span = [&](int r) { return __builtin_amdgcn_readlane(vspan, r); };
  auto slot_pairs = [&](int r) { return __builtin_amdgcn_readlane(hq.y, r); };
  auto slot_first = [&](int r) {
    return ((long long)(unsigned)__builtin_amdgcn_readlane(hq.z, r)) | ((long long)__builtin_amdgcn_readlane(hq.w, r) << 32);
  };
  // per-lane LDS offsets of the MFMA operands of step h: K index k = 4 h + (lane >> 4) is
  // (pair k / DP, component k % DP); row / column = lane & 15
  const int i = lane & 15, kk = lane >> 4;
  const bool row_ok = i < D;
  int koff[STEPS];
#pragma unroll
  for (int h = 0; h < STEPS; ++h) {
    const int k = 4 * h + kk;
    const int pr = k / DP;
    koff[h] = pr * PITCH + (row_ok ? i : 0) * DP + (k - pr * DP);
  }
  // chunk = (slot r, first pair c0, pairs n); r = R past the end.  All wave-uniform.
  struct Chunk {
    int r, c0, n;
  };
  auto make_chunk = [&](int r, int c0) {
    r = __builtin_amdgcn_readfirstlane(r);
    c0 = __builtin_amdgcn_readfirstlane(c0);
    while (r < R && c0 >= slot_span(r)) {
      ++r;
      c0 = 0;
    }
    Chunk c;
    c.r = r;
    c.c0 = c0;
    c.n = r < R ? max(0, min(PC, slot_pairs(r) - c0)) : 0;
    return c;
  };
  // lanes [0, PC) hold the i-side slots of the chunk's pairs, lanes [PC, 2 PC) the j-side slots
  auto load_slots = [&](const Chunk& c) {
    int sl = 0;
    if (c.n > 0) {
      const long long q = slot_first(c.r) + c.c0;
      if (lane < c.n) sl = v.pair_i[q + lane];
      else if (lane >= PC && lane < PC + c.n) sl = v.pair_j[q + lane - PC];
    }
    return sl;
  };
  double2 pf[NL];
  auto issue_loads = [&](const Chunk& c, int slots) {
    // all cross-lane slot reads first, then the loads back to back
    int sl[NL];
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int rec = (it * 64 + lane) / PARTS;
      sl[it] = __shfl(slots, rec < 2 * PC ? rec : 0, 64);
    }
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int f = it * 64 + lane;
      const int rec = f / PARTS, part = f - rec * PARTS;
      const int pr = rec >= PC ? rec - PC : rec;
      double2 t = make_double2(0.0, 0.0);
      if (rec < 2 * PC && pr < c.n)
        t = *reinterpret_cast<const double2*>(v.cm_Y + (size_t)sl[it] * YS + 2 * part);
      pf[it] = t;
    }
  };
  Chunk A = make_chunk(0, 0);
  int slots_a = load_slots(A);
  issue_loads(A, slots_a);
  Chunk B = make_chunk(A.r, A.c0 + PC);
  int slots_b = load_slots(B);
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  while (A.r < R) {
    // registers -> LDS (the previous chunk's operands have been consumed)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int f = it * 64 + lane;
      const int rec = f / PARTS, part = f - rec * PARTS;
      if (rec < 2 * PC) *reinterpret_cast<double2*>(lds + rec * PITCH + 2 * part) = pf[it];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // next chunk's records and the chunk after's indices go out before the contraction
    if (B.n > 0) issue_loads(B, slots_b);
    const Chunk C = make_chunk(B.r, B.c0 + PC);
    const int slots_c = load_slots(C);
    if (A.c0 == 0) acc = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int ksteps = (A.n * DP + 3) / 4;
#pragma unroll
    for (int g = 0; g < STEPS / GS; ++g) {
      if (g * GS < ksteps) {
        double a[GS], b[GS];
#pragma unroll
        for (int h = 0; h < GS; ++h) {
          a[h] = lds[koff[g * GS + h]];
          b[h] = lds[PC * PITCH + koff[g * GS + h]];
        }
#pragma unroll
        for (int h = 0; h < GS; ++h) {
          a[h] = row_ok ? a[h] : 0.0;
          b[h] = row_ok ? b[h] : 0.0;
        }
#pragma unroll
        for (int h = 0; h < GS; ++h) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[h], b[h], acc, 0, 0, 0);
      }
    }
    if (A.c0 + PC >= slot_span(A.r)) {
      // last chunk of the block: write it out
      double* out = v.red + L.ub + (size_t)__builtin_amdgcn_readlane(hq.x, A.r) * D * D;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = (lane >> 4) + 4 * q;
        if (row < D && i < D) out[row * D + i] = -acc[q];
      }
    }
    A = B;
    B = C;
    slots_b = slots_c;
  }
}

// schur_offdiag on the [A | Q] records (default without shared intrinsics blocks):
//   S_ij = - sum_pairs Y_i Y_j^T = - sum_pairs A_i^T (Q_i Q_j^T) A_j,
// a K = 2 per pair contraction: A operand = the two rows of A_i, B operand = M A_j with the 2 x 2
// M = Q_i Q_j^T of the pair.  Same launch slots, chunking (16 pairs), software pipeline and record
// staging as schur_offdiag_kernel above; after a chunk's records are in LDS lanes 0..15 form the
// chunk's M matrices, then 8 MFMA steps (instead of 12 for 9 x 3, 16 for 9 x 4) run with the B
// operand assembled from two A_j entries and one row of M.  No Y record exists on this path:
// point_eliminate writes 256 B less per observation and the gathered record is 192 B (9 x 3) or
// 256 B (9 x 4: two lines where the Y record takes three).
template <int D, int DP>
__global__ __launch_bounds__(256) void schur_offdiag_aq_kernel(DeviceView v, RedLayout L) {
  constexpr int RS = asa_of(D, DP);                 // record stride in cm_A (doubles)
  constexpr int R = kSchurBlocksPerWave;
  constexpr int PC = kSchurPairsPerChunk;
  constexpr int PARTS = (2 * D + 2 * DP + 1) / 2;   // 16-byte parts of a record that carry data
  constexpr int PITCH = RS + 2;                     // LDS pitch (doubles), keeps 16-byte alignment
  constexpr int NL = (2 * PC * PARTS + 63) / 64;
  constexpr int STEPS = PC * 2 / 4;                 // MFMA steps (K = 4 = two pairs) of a full chunk
  constexpr int GS = 4;
  static_assert(STEPS % GS == 0, "whole groups of MFMA steps");
  __shared__ __attribute__((aligned(16))) double lds_all[4][2 * PC * PITCH + 4 * PC];
  const int lane = threadIdx.x & 63;
  double* lds = lds_all[threadIdx.x >> 6];
  double* ldsM = lds + 2 * PC * PITCH;              // [PC][2][2]
  const long long first = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (first >= v.n_order) return;
  int4 hq = make_int4(-1, 0, 0, 0);
  if (lane < R && first + lane < v.n_order) hq = reinterpret_cast<const int4*>(v.ub_order)[first + lane];
  const int vspan = hq.x >= 0 ? max(hq.y, 1) : 0;
  auto slot_span = [&](int r) { return __builtin_amdgcn_readlane(vspan, r); };
  auto slot_pairs = [&](int r) { return __builtin_amdgcn_readlane(hq.y, r); };
  auto slot_first = [&](int r) {
    return ((long long)(unsigned)__builtin_amdgcn_readlane(hq.z, r)) | ((long long)__builtin_amdgcn_readlane(hq.w, r) << 32);
  };
  // MFMA operand addressing: K index k = 4 h + kk is (pair 2 h + (kk >> 1), row r = kk & 1)
  const int i = lane & 15, kk = lane >> 4;
  const bool row_ok = i < D;
  const int ic = row_ok ? i : 0;
  const int pr0 = kk >> 1, rr = kk & 1;
  const int off_a = pr0 * PITCH + rr * D + ic;          // + 2 h PITCH: A_i[pair][r][i]
  const int off_j = (PC + pr0) * PITCH + ic;            // + 2 h PITCH: A_j[pair][0][i], + D: row 1
  const int off_m = pr0 * 4 + 2 * rr;                   // + 8 h: M[pair][r][0 .. 1]
  struct Chunk {
    int r, c0, n;
  };
  auto make_chunk = [&](int r, int c0) {
    r = __builtin_amdgcn_readfirstlane(r);
    c0 = __builtin_amdgcn_readfirstlane(c0);
    while (r < R && c0 >= slot_span(r)) {
      ++r;
      c0 = 0;
    }
    Chunk c;
    c.r = r;
    c.c0 = c0;
    c.n = r < R ? max(0, min(PC, slot_pairs(r) - c0)) : 0;
    return c;
  };
  auto load_slots = [&](const Chunk& c) {
    int sl = 0;
    if (c.n > 0) {
      const long long q = slot_first(c.r) + c.c0;
      if (lane < c.n) sl = v.pair_i[q + lane];
      else if (lane >= PC && lane < PC + c.n) sl = v.pair_j[q + lane - PC];
    }
    return sl;
  };
  double2 pf[NL];
  auto issue_loads = [&](const Chunk& c, int slots) {
    int sl[NL];
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int rec = (it * 64 + lane) / PARTS;
      sl[it] = __shfl(slots, rec < 2 * PC ? rec : 0, 64);
    }
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int f = it * 64 + lane;
      const int rec = f / PARTS, part = f - rec * PARTS;
      const int pr = rec >= PC ? rec - PC : rec;
      double2 t = make_double2(0.0, 0.0);
      if (rec < 2 * PC && pr < c.n) t = *reinterpret_cast<const double2*>(v.cm_A + (size_t)sl[it] * RS + 2 * part);
      pf[it] = t;
    }
  };
  Chunk A = make_chunk(0, 0);
  int slots_a = load_slots(A);
  issue_loads(A, slots_a);
  Chunk B = make_chunk(A.r, A.c0 + PC);
  int slots_b = load_slots(B);
  v4f64 acc = {0.0, 0.0, 0.0, 0.0};
  while (A.r < R) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NL; ++it) {
      const int f = it * 64 + lane;
      const int rec = f / PARTS, part = f - rec * PARTS;
      if (rec < 2 * PC) *reinterpret_cast<double2*>(lds + rec * PITCH + 2 * part) = pf[it];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (B.n > 0) issue_loads(B, slots_b);
    const Chunk C = make_chunk(B.r, B.c0 + PC);
    const int slots_c = load_slots(C);
    // M = Q_i Q_j^T of every pair of the chunk (zero records give zero)
    if (lane < PC) {
      const double* qi = lds + lane * PITCH + 2 * D;
      const double* qj = lds + (PC + lane) * PITCH + 2 * D;
      double m00 = 0.0, m01 = 0.0, m10 = 0.0, m11 = 0.0;
#pragma unroll
      for (int c = 0; c < DP; ++c) {
        const double a0 = qi[c], a1 = qi[DP + c], b0 = qj[c], b1 = qj[DP + c];
        m00 += a0 * b0;
        m01 += a0 * b1;
        m10 += a1 * b0;
        m11 += a1 * b1;
      }
      *reinterpret_cast<double2*>(ldsM + 4 * lane) = make_double2(m00, m01);
      *reinterpret_cast<double2*>(ldsM + 4 * lane + 2) = make_double2(m10, m11);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (A.c0 == 0) acc = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int ksteps = (A.n * 2 + 3) / 4;
#pragma unroll
    for (int g = 0; g < STEPS / GS; ++g) {
      if (g * GS < ksteps) {
        double a[GS], b[GS];
#pragma unroll
        for (int h = 0; h < GS; ++h) {
          const int hh = g * GS + h;
          a[h] = lds[off_a + 2 * hh * PITCH];
          const double2 m = *reinterpret_cast<const double2*>(ldsM + off_m + 8 * hh);
          const double j0 = lds[off_j + 2 * hh * PITCH], j1 = lds[off_j + 2 * hh * PITCH + D];
          b[h] = m.x * j0 + m.y * j1;
        }
#pragma unroll
        for (int h = 0; h < GS; ++h) {
          a[h] = row_ok ? a[h] : 0.0;
          b[h] = row_ok ? b[h] : 0.0;
        }
#pragma unroll
        for (int h = 0; h < GS; ++h) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[h], b[h], acc, 0, 0, 0);
      }
    }
    if (A.c0 + PC >= slot_span(A.r)) {
      double* out = v.red + L.ub + (size_t)__builtin_amdgcn_readlane(hq.x, A.r) * D * D;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = (lane >> 4) + 4 * q;
        if (row < D && i < D) out[row * D + i] = -acc[q];
      }
    }
    A = B;
    B = C;
    slots_b = slots_c;
  }
}

// ------------------------------------------------------------------------------
// finish_diag (timed as TMI_BA_K_REDUCE): diagonal blocks of S = (all-reduced)
// raw diagonal block + LM diagonal clamp(U_aa) / radius; padding rows get 1.
// The off-diagonal blocks need no post-processing: S is symmetric and the
// upper blocks are used where schur_offdiag (and the all-reduce) left them.
// ------------------------------------------------------------------------------
// want_gmax: also scal[SC_GMAX] = max |g_c / scale| over the free camera columns (the gradient
// tolerance test), finished by the last workgroup.
template <int D>
__global__ __launch_bounds__(256) void finish_diag_kernel(DeviceView v, RedLayout L, double inv_radius,
                                                          double lm_lo, double lm_hi, int want_gmax) {
  __shared__ double fd_sh[4];
  __shared__ int fd_last;
  const int e = blockIdx.x * 256 + threadIdx.x;
  double gm = 0.0;
  if (e < v.Nrb * D * D) {
    const int rb = e / (D * D);
    const int w = e - rb * (D * D);
    const int a = w / D, b = w - a * D;
    double val = v.red[L.diag + e];
    if (a == b) {
      if (v.rb_cols[(size_t)rb * D + a] < 0) {
        val = 1.0;
      } else {
        const double d = v.red[L.udiag + (size_t)rb * D + a];
        val += fmin(fmax(d, lm_lo), lm_hi) * inv_radius;
        if (want_gmax) gm = fabs(v.red[L.gc + (size_t)rb * D + a] / v.scale_c[(size_t)rb * D + a]);
      }
    }
    v.Sdiag[e] = val;
  }
  if (!want_gmax) return;
  const double wm = wave_max(gm);
  if ((threadIdx.x & 63) == 0) fd_sh[threadIdx.x >> 6] = wm;
  __syncthreads();
  if (threadIdx.x == 0) {
    st_agent(&v.dotbuf[blockIdx.x], fmax(fmax(fd_sh[0], fd_sh[1]), fmax(fd_sh[2], fd_sh[3])));
    fd_last = take_ticket(v.ticket + 2 * kTicketStride, (int)gridDim.x) ? 1 : 0;
  }
  __syncthreads();
  if (!fd_last) return;
  double m = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) m = fmax(m, ld_agent(&v.dotbuf[i]));
  const double wm2 = wave_max(m);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) fd_sh[threadIdx.x >> 6] = wm2;
  __syncthreads();
  if (threadIdx.x == 0) {
    v.scal[SC_GMAX] = fmax(fmax(fd_sh[0], fd_sh[1]), fmax(fd_sh[2], fd_sh[3]));
  }
}

// ------------------------------------------------------------------------------
// precond_invert (kernel class 4): SCHUR_JACOBI = inverse of the diagonal
// blocks of S, one wavefront per block, matrix staged in LDS.
// ------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(64) void precond_invert_kernel(DeviceView v, int mode) {
  // mode 0: merged [extrinsics | intrinsics] block per view, 1: identity, 2: one block per
  // PARAMETER block as ceres's SchurJacobiPreconditioner builds it (the extrinsics x intrinsics
  // cross terms of a view are dropped before the inversion)
  __shared__ double M[D * D];
  __shared__ int bad;
  const int rb = blockIdx.x;
  const int t = threadIdx.x;
  const double* src = v.Sdiag + (size_t)rb * D * D;
  const signed char* cols = v.rb_cols + (size_t)rb * D;
  for (int e = t; e < D * D; e += 64) {
    double m = src[e];
    if (mode == 2) {
      const int ci = cols[e / D], cj = cols[e % D];
      if (ci >= 0 && cj >= 0 && ((ci < 6) != (cj < 6))) m = 0.0;
    }
    M[e] = m;
  }
  if (t == 0) bad = 0;
  __syncthreads();
  double* out = v.Minv + (size_t)rb * D * D;
  const int identity = mode == 1;
  if (identity) {
    for (int e = t; e < D * D; e += 64) out[e] = (e / D == e % D) ? 1.0 : 0.0;
    return;
  }
  for (int j = 0; j < D; ++j) {
    if (t == 0) {
      const double d = M[j * D + j];
      if (!(d > 0.0)) {
        bad = 1;
        M[j * D + j] = 1.0;
      } else {
        M[j * D + j] = sqrt(d);
      }
    }
    __syncthreads();
    if (t > j && t < D) M[t * D + j] /= M[j * D + j];
    __syncthreads();
    // trailing update: (i, m), j < m <= i
    for (int e = t; e < D * D; e += 64) {
      const int i = e / D, m = e - i * D;
      if (m > j && m <= i) M[e] -= M[i * D + j] * M[m * D + j];
    }
    __syncthreads();
  }
  if (t < D) {
    // column t of (L L^T)^-1
    double y[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double s = (i == t) ? 1.0 : 0.0;
      for (int m = 0; m < i; ++m) s -= M[i * D + m] * y[m];
      y[i] = s / M[i * D + i];
    }
#pragma unroll
    for (int i = D - 1; i >= 0; --i) {
      double s = y[i];
      for (int m = i + 1; m < D; ++m) s -= M[m * D + i] * y[m];
      y[i] = s / M[i * D + i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) out[i * D + t] = y[i];
  }
  if (t == 0 && bad) v.flags[FL_SINGULAR_BLOCK] = 1;
}

// ------------------------------------------------------------------------------
// spmv (kernel class 5): q = S p with S stored SYMMETRIC (upper blocks + diagonal),
// the PCG hot kernel (hot loop 3).  Purely HBM-bound: every upper block (8 D^2
// bytes) is read exactly once per product -- the algorithmic minimum.
//
//  rows pass: the upper blocks of a block row are cut into chunks of kSpmvTrips * G
//    consecutive blocks (G = 64 / D), ONE WAVEFRONT PER CHUNK (host work list spc_*): every
//    wave streams the same amount of S regardless of how long its row is (row lengths of
//    the upper triangle run from ~N to 0).  Per trip the wave fetches G blocks = G D^2
//    contiguous doubles with fully coalesced 16-byte loads and parks them in its own LDS
//    area; lane (g, c) then reads COLUMN c of block g = (i, j) once and uses it twice:
//      t_j[c]  = sum_r U[r][c] x_i[r]     -> tbuf[u][c]   (the transposed product for y_j)
//      a[r]   += U[r][c] x_j[c]           (r = 0..D-1: this lane's share of y_i)
//    and the D accumulators are summed over the wave at the end of the chunk -> rbuf.
//    (Reading rows and columns straight from global memory costs ~10 cache-line accesses
//    per line of S in the vector L1 and caps the kernel at half the HBM rate.)
//  cols pass: y_j = Sdiag_j x_j + sum over the chunks of row j of rbuf
//                 + sum over the blocks of column j of tbuf[u]        (fixed orders).
// No atomics anywhere: bit-reproducible.
// ------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void spmv_rows_kernel(DeviceView v, const double* __restrict__ ub,
                                                        const double* __restrict__ x) {
  constexpr int G = 64 / D;
  constexpr int BLK = D * D;
  constexpr int NW = G * BLK;            // doubles per trip
  constexpr int NLD = (NW + 127) / 128;  // 16-byte loads per lane per trip
  constexpr int PITCH = (NW + 1) & ~1;
  __shared__ __attribute__((aligned(16))) double sblk[4][PITCH];  // private to each wave
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int cidx = blockIdx.x * 4 + w;
  if (cidx >= v.n_spc) return;  // no workgroup barrier below
  const int row = v.spc_row[cidx];
  const int u0 = v.spc_u0[cidx];
  const int u1 = min(u0 + kSpmvTrips * G, v.urow_ptr[row + 1]);
  const int g = lane / D, c = lane - g * D;
  double xi[D], a[D];
#pragma unroll
  for (int r = 0; r < D; ++r) {
    xi[r] = x[(size_t)row * D + r];
    a[r] = 0.0;
  }
  double* __restrict__ tb = v.tbuf;

  double2_a8 ld[NLD];
  double xjc = 0.0;
  auto fetch = [&](int ub0) {
    const int nv = max(0, min(G, u1 - ub0)) * BLK;  // valid doubles of this trip
    const double* src = ub + (size_t)ub0 * BLK;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = 2 * (64 * i + lane);
      double2_a8 t2 = {0.0, 0.0};
      if (e + 1 < nv) {
        t2 = *reinterpret_cast<const double2_a8*>(src + e);
      } else if (e < nv) {
        t2.x = src[e];
      }
      ld[i] = t2;
    }
    const int u = ub0 + g;
    xjc = (g < G && u < u1) ? x[(size_t)v.ub_j[u] * D + c] : 0.0;
  };
  fetch(u0);
#pragma unroll 1
  for (int ub0 = u0; ub0 < u1; ub0 += G) {
    // LDS operations of one wave execute in order and the staging area is the wave's own:
    // no workgroup barrier, only compiler fences around the exchange
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = 2 * (64 * i + lane);
      if (e < PITCH) *reinterpret_cast<double2_a8*>(&sblk[w][e]) = ld[i];
    }
    const double xj = xjc;
    __builtin_amdgcn_wave_barrier();
    if (ub0 + G < u1) fetch(ub0 + G);  // the next trip's loads fly while this one is consumed
    const int u = ub0 + g;
    if (g < G && u < u1) {
      const double* blk = &sblk[w][g * BLK];
      double tt = 0.0;
#pragma unroll
      for (int r = 0; r < D; ++r) {
        const double e = blk[r * D + c];
        tt += e * xi[r];
        a[r] += e * xj;
      }
      tb[(size_t)u * D + c] = tt;
    }
    __builtin_amdgcn_wave_barrier();
  }
  // sum the D accumulators over the wave (fixed butterfly => reproducible)
#pragma unroll
  for (int r = 0; r < D; ++r) a[r] = wave_sum(a[r]);
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < D; ++r) v.rbuf[(size_t)cidx * D + r] = a[r];
  }
}

// Deterministic grid-wide sum of one value per workgroup without a second launch: every
// workgroup stores its value, takes a ticket, and the LAST one to arrive adds them up in a fixed
// order (thread t takes elements t, t + T, ...; then a fixed tree).  Returns true in the last
// workgroup only (all of its threads), with the total in *total.
template <int T>
__device__ __forceinline__ bool last_block_sum(double mine, double* __restrict__ slots, int* ticket, double* total) {
  __shared__ double lb_sh[T];
  __shared__ int lb_last;
  if (threadIdx.x == 0) {
    st_agent(&slots[blockIdx.x], mine);
    lb_last = take_ticket(ticket, (int)gridDim.x) ? 1 : 0;
  }
  __syncthreads();
  if (!lb_last) return false;
  double acc = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += T) acc += ld_agent(&slots[i]);
  lb_sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = T / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) lb_sh[threadIdx.x] += lb_sh[threadIdx.x + o];
    __syncthreads();
  }
  *total = lb_sh[0];
  return true;
}

// dot != 0: also y[Nrb D] = x . y (the p.q of a PCG iteration), summed by the last workgroup;
template <int D>
__global__ __launch_bounds__(256) void spmv_cols_kernel(DeviceView v, const double* __restrict__ x,
                                                        double* __restrict__ y, int dot) {
  constexpr int G = 64 / D;
  __shared__ double part[4][G][D];
  const int col = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane / D, r = lane - g * D;
  if (g < G) {
    // the column's transposed products, gathered: every trip is two dependent loads (block index, then its entry of
    // tbuf), so four trips are kept in flight -- a column of the bench problem is ~25 trips of 28 blocks, and one
    // round trip per trip was the run time of this kernel (56 us; the sum order is unchanged)
    double acc = 0.0;
    const int k0 = v.ucol_ptr[col], k1 = v.ucol_ptr[col + 1];
    constexpr int STEP = 4 * G;
    int k = k0 + w * G + g;
    for (; k + 3 * STEP < k1; k += 4 * STEP) {
      const int u0 = v.ucol_u[k], u1 = v.ucol_u[k + STEP], u2 = v.ucol_u[k + 2 * STEP], u3 = v.ucol_u[k + 3 * STEP];
      const double t0 = v.tbuf[(size_t)u0 * D + r], t1 = v.tbuf[(size_t)u1 * D + r];
      const double t2 = v.tbuf[(size_t)u2 * D + r], t3 = v.tbuf[(size_t)u3 * D + r];
      acc += t0;
      acc += t1;
      acc += t2;
      acc += t3;
    }
    for (; k < k1; k += STEP) acc += v.tbuf[(size_t)v.ucol_u[k] * D + r];
    part[w][g][r] = acc;
  }
  __syncthreads();
  if (threadIdx.x < D) {
    double s = 0.0;
    const double* dg = v.Sdiag + (size_t)col * D * D + threadIdx.x * D;
#pragma unroll
    for (int cc = 0; cc < D; ++cc) s += dg[cc] * x[(size_t)col * D + cc];
    for (int c = v.spc_rptr[col]; c < v.spc_rptr[col + 1]; ++c) s += v.rbuf[(size_t)c * D + threadIdx.x];
#pragma unroll
    for (int ww = 0; ww < 4; ++ww)
#pragma unroll
      for (int gg = 0; gg < G; ++gg) s += part[ww][gg][threadIdx.x];
    y[(size_t)col * D + threadIdx.x] = s;
    if (dot) part[0][0][threadIdx.x] = s * x[(size_t)col * D + threadIdx.x];
  }
  if (dot) {
    __syncthreads();
    double mine = 0.0;
    if (threadIdx.x == 0)
      for (int a = 0; a < D; ++a) mine += part[0][0][a];
    double tot;
    if (last_block_sum<256>(mine, v.dotbuf, v.ticket, &tot) && threadIdx.x == 0) y[(size_t)v.Nrb * D] = tot;
  }
}

// ------------------------------------------------------------------------------
// Implicit Schur operator (kernel class 5, schur_mode = implicit): q = S p without
// ever forming S (Ceres' ImplicitSchurComplement, the ITERATIVE_SCHUR path):
//   S p = D_c p + sum_obs A_i^T t_i,   t_i = u_i - Jp_i z_track,
//   u_i = A_i p_cam(i),   z = (V + D_p)^-1 sum_{i in track} Jp_i^T u_i.
//  tracks pass : thread per track (SELL), writes t_i to the observation's camera-major slot
//  cameras pass: one wave per camera block reduces A_i^T t_i over its records (fixed order)
// Every product streams the observations once (track-major planes) plus the camera-major
// A records; with several GPUs only the reduced vector q is all-reduced.
// ------------------------------------------------------------------------------
// With shared intrinsics blocks the same ONE sweep (round 2 walked a track's observations twice, kept u planes and
// scattered 16-byte t records): u_i = A_i x_c(i) + A1_i x_g(i) with x_g the view's shared block, zhat as below; the
// cameras pass recomputes u_i from its record (which carries A, A1 and Q) and leaves the view's partial of the shared
// block for implicit_groups to add up in the block's view order.
template <int D, int DP, typename PT = double>
__global__ __launch_bounds__(256) void implicit_tracks_sq_kernel(DeviceView v, const double* __restrict__ x,
                                                                 double* __restrict__ zhat) {
  const TrackMap tm = track_map(v);
  PT* const pmR = reinterpret_cast<PT*>(v.pm_r);   // (the planes in their storage type: double, or float with
  PT* const pmA = reinterpret_cast<PT*>(v.pm_A);   //  fp32 evaluation on shared-intrinsics problems -- DeviceView::planes_fp32)
  PT* const pmA1 = reinterpret_cast<PT*>(v.pm_A1);
  PT* const pmJp = reinterpret_cast<PT*>(v.pm_Jp);
  (void)pmR; (void)pmA; (void)pmA1; (void)pmJp;
  if (!tm.valid) return;
  const int lp = tm.lp;
  const int k = tm.k;
  if (k == 0) return;  // uniform over the lanes that share a track
  const size_t base = tm.base;
  const size_t NP = (size_t)v.Np_pad;
  double w[DP];
#pragma unroll
  for (int a = 0; a < DP; ++a) w[a] = 0.0;
  int cam_next = (tm.j0 < k) ? v.obs_cam[base + (size_t)tm.j0 * 64] : -1;
  for (int j = tm.j0; j < k; j += tm.jstep) {
    const size_t e = base + (size_t)j * 64;
    const int cam = cam_next;
    if (j + tm.jstep < k) cam_next = v.obs_cam[e + (size_t)tm.jstep * 64];
    if (cam < 0) continue;
    const int rb = v.cam_rb[cam], grb = v.cam_grb[cam];
    double u0 = 0.0, u1 = 0.0;
    if (rb >= 0) {
      const double* xc = x + (size_t)rb * D;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const double xa = xc[a];
        u0 += pmA[pidx<2 * D>((2 * a), e)] * xa;
        u1 += pmA[pidx<2 * D>((2 * a + 1), e)] * xa;
      }
    }
    if (grb >= 0) {
      const double* xg = x + (size_t)grb * D;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const double xa = xg[a];
        u0 += pmA1[pidx<2 * D>((2 * a), e)] * xa;
        u1 += pmA1[pidx<2 * D>((2 * a + 1), e)] * xa;
      }
    }
#pragma unroll
    for (int a = 0; a < DP; ++a)
      w[a] += pmJp[pidx<2 * DP>((2 * a), e)] * u0 + pmJp[pidx<2 * DP>((2 * a + 1), e)] * u1;
  }
#pragma unroll
  for (int a = 0; a < DP; ++a) w[a] = group_sum(w[a], tm.wide);
  if (tm.leader) {
    double z[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < DP; ++b) {
      double t = 0.0;
#pragma unroll
      for (int a = 0; a <= b; ++a) t += v.Linv[(size_t)sym_idx(a, b, DP) * NP + lp] * w[a];
      z[b] = t;
    }
    *reinterpret_cast<double2*>(zhat + (size_t)lp * 4) = make_double2(z[0], z[1]);
    *reinterpret_cast<double2*>(zhat + (size_t)lp * 4 + 2) = make_double2(z[2], z[3]);
  }
}

// ------------------------------------------------------------------------------
// The matrix-free product without shared intrinsics blocks, from the [A | Q] records:
//   S x = sum_i A_i^T (A_i x_c(i) - Q_i zhat_t(i)) + D_c x,   zhat_t = sum_{i in t} Q_i^T A_i x_c(i) = L^-1 sum Jp_i^T u_i
// (Q = Jp L^-T, so Jp (V+D)^-1 Jp^T = Q Q^T).  The tracks pass is ONE sweep over the planes that
// leaves 32 bytes per track (zhat); the cameras pass streams the records it streams anyway (Q rides
// in them), recomputes u_i = A_i x_c -- x_c is the workgroup's own block -- and gathers zhat of the
// slot's track.  No u planes, no t records, no second sweep over the track's observations:
// 196 + ~230 bytes per observation instead of 310 + 224.
// ------------------------------------------------------------------------------
template <int D, int DP>
__global__ __launch_bounds__(256) void implicit_tracks_q_kernel(DeviceView v, const double* __restrict__ x,
                                                                double* __restrict__ zhat) {
  const TrackMap tm = track_map(v);
  if (!tm.valid) return;
  const int lp = tm.lp;
  const int k = tm.k;
  if (k == 0) return;  // uniform over the lanes that share a track
  const size_t base = tm.base;
  const size_t NP = (size_t)v.Np_pad;
  double w[DP];
#pragma unroll
  for (int a = 0; a < DP; ++a) w[a] = 0.0;
  // the block index of the NEXT observation is fetched while this one is processed, so an iteration
  // is one memory round trip (planes + the block's x, all independent) instead of three dependent ones
  int rb_next = (tm.j0 < k) ? v.obs_rb[base + (size_t)tm.j0 * 64] : -1;
  for (int j = tm.j0; j < k; j += tm.jstep) {
    const size_t e = base + (size_t)j * 64;
    const int rb = rb_next;
    if (j + tm.jstep < k) rb_next = v.obs_rb[e + (size_t)tm.jstep * 64];
    if (rb < 0) continue;
    const double* xc = x + (size_t)rb * D;
    double u0 = 0.0, u1 = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const double xa = xc[a];
      u0 += v.pm_A[pidx<2 * D>((2 * a), e)] * xa;
      u1 += v.pm_A[pidx<2 * D>((2 * a + 1), e)] * xa;
    }
#pragma unroll
    for (int a = 0; a < DP; ++a)
      w[a] += v.pm_Jp[pidx<2 * DP>((2 * a), e)] * u0 + v.pm_Jp[pidx<2 * DP>((2 * a + 1), e)] * u1;
  }
#pragma unroll
  for (int a = 0; a < DP; ++a) w[a] = group_sum(w[a], tm.wide);
  if (tm.leader) {
    double z[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int b = 0; b < DP; ++b) {
      double t = 0.0;
#pragma unroll
      for (int a = 0; a <= b; ++a) t += v.Linv[(size_t)sym_idx(a, b, DP) * NP + lp] * w[a];
      z[b] = t;
    }
    *reinterpret_cast<double2*>(zhat + (size_t)lp * 4) = make_double2(z[0], z[1]);
    *reinterpret_cast<double2*>(zhat + (size_t)lp * 4 + 2) = make_double2(z[2], z[3]);
  }
}

// reduced block of every observation's view (built once per structure)
__global__ __launch_bounds__(256) void obs_rb_kernel(const int* __restrict__ obs_cam, const int* __restrict__ cam_rb,
                                                     int Nc, long long n, int* __restrict__ obs_rb) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const int cam = obs_cam[e];
  obs_rb[e] = (cam >= 0 && cam < Nc) ? cam_rb[cam] : -1;
}

// slot -> track of the rank's camera-major records (built once per structure)
__global__ __launch_bounds__(256) void slot_track_kernel(DeviceView v, int* __restrict__ slot_track) {
  const TrackMap tm = track_map(v);
  if (!tm.valid) return;
  for (int j = tm.j0; j < tm.k; j += tm.jstep) {
    const int cpos = v.obs_cpos[tm.base + (size_t)j * 64];
    if (cpos >= 0) slot_track[cpos] = tm.lp;
  }
}

template <int D, int DP>
__global__ __launch_bounds__(64) void implicit_cameras_q_kernel(DeviceView v, RedLayout L,
                                                                const double* __restrict__ x,
                                                                const double* __restrict__ zhat,
                                                                double* __restrict__ y, double inv_radius,
                                                                double lm_lo, double lm_hi, int add_diag, int dot) {
  constexpr int ASA = asa_of(D, DP);
  // Workgroups are dealt to the 8 XCDs round-robin: XCD x takes the views [x chunk, (x + 1) chunk), so the views
  // in flight on one XCD are neighbours and the zhat lines of the tracks they share are fetched into that XCD's
  // L2 once (the track order keeps the tracks of neighbouring views together).  Grid = 8 chunk workgroups.
  const int chunk = ((int)gridDim.x) >> 3;
  const int rb = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  const bool live = rb < v.Nrb;  // the padding workgroups still take part in the dot product's ticket
  double xc[D], acc[D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    xc[a] = live ? x[(size_t)rb * D + a] : 0.0;
    acc[a] = 0.0;
  }
  const int s_end = live ? v.cam_ptr[rb + 1] : 0;
  for (int s = (live ? v.cam_ptr[rb] : 0) + threadIdx.x; s < s_end; s += 64) {
    const double* arec = v.cm_A + (size_t)s * ASA;
    double rec[2 * D + 2 * DP];
#pragma unroll
    for (int i = 0; i < 2 * D + 2 * DP; i += 2) {
      const double2 t = *reinterpret_cast<const double2*>(arec + i);
      rec[i] = t.x;
      rec[i + 1] = t.y;
    }
    const double* zt = zhat + (size_t)v.slot_track[s] * 4;
    const double2 z01 = *reinterpret_cast<const double2*>(zt);
    const double2 z23 = *reinterpret_cast<const double2*>(zt + 2);
    const double z[4] = {z01.x, z01.y, z23.x, z23.y};
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      t0 += rec[a] * xc[a];
      t1 += rec[D + a] * xc[a];
    }
#pragma unroll
    for (int b = 0; b < DP; ++b) {
      t0 -= rec[2 * D + b] * z[b];
      t1 -= rec[2 * D + DP + b] * z[b];
    }
#pragma unroll
    for (int a = 0; a < D; ++a) acc[a] += rec[a] * t0 + rec[D + a] * t1;
  }
  double my_dot = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double tot = wave_sum(acc[a]);
    if (threadIdx.x == 0 && live) {
      // the damping (and the identity on padding rows) enters once: on rank 0 when the
      // product is all-reduced afterwards
      if (add_diag) {
        if (v.rb_cols[(size_t)rb * D + a] < 0) {
          tot = xc[a];
        } else {
          const double d = v.red[L.udiag + (size_t)rb * D + a];
          tot += fmin(fmax(d, lm_lo), lm_hi) * inv_radius * xc[a];
        }
      } else if (v.rb_cols[(size_t)rb * D + a] < 0) {
        tot = 0.0;
      }
      y[(size_t)rb * D + a] = tot;
      my_dot += tot * xc[a];
    }
  }
  if (dot) {
    // this rank's share of x . y rides behind the product vector and is all-reduced with it
    double total;
    if (last_block_sum<64>(my_dot, v.dotbuf, v.ticket, &total) && threadIdx.x == 0) y[(size_t)v.Nrb * D] = total;
  }
}

// cameras pass with shared intrinsics blocks: one wave per VIEW block (shared blocks: implicit_groups_kernel)
template <int D, int DP, typename RS = double>
__global__ __launch_bounds__(64) void implicit_cameras_sq_kernel(DeviceView v, RedLayout L,
                                                                 const double* __restrict__ x,
                                                                 const double* __restrict__ zhat,
                                                                 double* __restrict__ y, double inv_radius,
                                                                 double lm_lo, double lm_hi, int add_diag,
                                                                 double* __restrict__ grp_part) {
  constexpr int AS = as_of(D, true);
  const int rb = blockIdx.x;
  if (rb >= v.Ncam_rb) return;
  const int cam = v.rb_cam[rb];
  const int grb = cam >= 0 ? v.cam_grb[cam] : -1;
  double xc[D], xg[D], acc[D], acc1[D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    xc[a] = x[(size_t)rb * D + a];
    xg[a] = grb >= 0 ? x[(size_t)grb * D + a] : 0.0;
    acc[a] = acc1[a] = 0.0;
  }
  for (int s = v.cam_ptr[rb] + threadIdx.x; s < v.cam_ptr[rb + 1]; s += 64) {
    const RS* arec = reinterpret_cast<const RS*>(v.cm_A) + (size_t)s * AS;
    double rec[4 * D + 12];
#pragma unroll
    for (int i = 0; i < 4 * D + 12; i += 2) {
      const double2 t = rec_ld2(arec + i);
      rec[i] = t.x;
      rec[i + 1] = t.y;
    }
    const double* zt = zhat + (size_t)v.slot_track[s] * 4;
    const double2 z01 = *reinterpret_cast<const double2*>(zt);
    const double2 z23 = *reinterpret_cast<const double2*>(zt + 2);
    const double z[4] = {z01.x, z01.y, z23.x, z23.y};
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      t0 += rec[a] * xc[a] + rec[sh_off_a1(D) + a] * xg[a];
      t1 += rec[D + a] * xc[a] + rec[sh_off_a1(D) + D + a] * xg[a];
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      t0 -= rec[sh_off_q(D) + b] * z[b];
      t1 -= rec[sh_off_q(D) + 4 + b] * z[b];
    }
#pragma unroll
    for (int a = 0; a < D; ++a) {
      acc[a] += rec[a] * t0 + rec[D + a] * t1;
      acc1[a] += rec[sh_off_a1(D) + a] * t0 + rec[sh_off_a1(D) + D + a] * t1;
    }
  }
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const double tot1 = wave_sum(acc1[a]);
    if (threadIdx.x == 0) grp_part[(size_t)rb * D + a] = tot1;
  }
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double tot = wave_sum(acc[a]);
    if (threadIdx.x == 0) {
      // the damping (and the identity on padding rows) enters once: on rank 0 when the
      // product is all-reduced afterwards
      if (add_diag) {
        if (v.rb_cols[(size_t)rb * D + a] < 0) {
          tot = xc[a];
        } else {
          const double d = v.red[L.udiag + (size_t)rb * D + a];
          tot += fmin(fmax(d, lm_lo), lm_hi) * inv_radius * xc[a];
        }
      } else if (v.rb_cols[(size_t)rb * D + a] < 0) {
        tot = 0.0;
      }
      y[(size_t)rb * D + a] = tot;
    }
  }
}

// shared intrinsics blocks of the implicit product: sum of the per-view partials in the
// block's view order (fixed), plus the damping on rank 0
template <int D>
__global__ __launch_bounds__(64) void implicit_groups_kernel(DeviceView v, RedLayout L,
                                                             const double* __restrict__ x,
                                                             const double* __restrict__ grp_part,
                                                             double* __restrict__ y, double inv_radius,
                                                             double lm_lo, double lm_hi, int add_diag) {
  const int gi = blockIdx.x;
  const int grb = v.Ncam_rb + gi;
  const int a = threadIdx.x;
  if (a >= D) return;
  double tot = 0.0;
  for (int k = v.grp_cam_ptr[gi]; k < v.grp_cam_ptr[gi + 1]; ++k) tot += grp_part[(size_t)v.cam_rb[v.grp_cams[k]] * D + a];
  if (add_diag) {
    const double xa = x[(size_t)grb * D + a];
    if (v.rb_cols[(size_t)grb * D + a] < 0) {
      tot = xa;
    } else {
      const double d = v.red[L.udiag + (size_t)grb * D + a];
      tot += fmin(fmax(d, lm_lo), lm_hi) * inv_radius * xa;
    }
  } else if (v.rb_cols[(size_t)grb * D + a] < 0) {
    tot = 0.0;
  }
  y[(size_t)grb * D + a] = tot;
}

// ------------------------------------------------------------------------------
// PCG vector kernels (class 6), single 1024-thread workgroup each so that dot
// products, the scalar recurrences and the vector updates of one half-step
// share a launch and need no host round trip.  Restates
// ceres/conjugate_gradients_solver.cc (1.14).
// ------------------------------------------------------------------------------
__device__ __forceinline__ double block1024_sum(double v, double* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double s = wave_sum(v);
  __syncthreads();
  if (lane == 0) sh[w] = s;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) t += sh[i];
  return t;
}

// x = 0, r = b, Q0 = 0, rho = 1
__global__ __launch_bounds__(1024) void pcg_begin_kernel(DeviceView v, const double* __restrict__ b, int n) {
  for (int i = threadIdx.x; i < n; i += 1024) {
    v.yc[i] = 0.0;
    v.cg_r[i] = b[i];
  }
  if (threadIdx.x == 0) {
    v.scal[SC_RHO] = 1.0;
    v.scal[SC_Q0] = 0.0;
    v.flags[FL_PCG_FAIL] = 0;
    *v.pcg_done = 0;
    v.ticket[0] = 0;
    v.ticket[1] = 0;
  }
}

// z = M^-1 r; rho = r.z; p = z (+ beta p)
template <int D>
__global__ __launch_bounds__(1024) void pcg_a_kernel(DeviceView v, int n, int it) {
  __shared__ double sh[16];
  double local = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const int rb = i / D, a = i - rb * D;
    const double* M = v.Minv + (size_t)rb * D * D + a * D;
    const double* rr = v.cg_r + (size_t)rb * D;
    double z = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) z += M[c] * rr[c];
    v.cg_z[i] = z;
    local += z * v.cg_r[i];
  }
  const double rho = block1024_sum(local, sh);
  const double last_rho = v.scal[SC_RHO];
  __syncthreads();
  if (threadIdx.x == 0) {
    v.scal[SC_LAST_RHO] = last_rho;
    v.scal[SC_RHO] = rho;
    if (rho == 0.0 || !isfinite(rho)) v.flags[FL_PCG_FAIL] = 1;
  }
  const double beta = rho / last_rho;
  if (it > 1 && (beta == 0.0 || !isfinite(beta)) && threadIdx.x == 0) v.flags[FL_PCG_FAIL] = 1;
  for (int i = threadIdx.x; i < n; i += 1024)
    v.cg_p[i] = (it == 1) ? v.cg_z[i] : v.cg_z[i] + beta * v.cg_p[i];
}

// One PCG iteration after q = S p (ceres/conjugate_gradients_solver.cc restated) is
// three light launches instead of one long single-workgroup kernel:
//   pcg_b1 (1 workgroup)   pq = p.q, alpha = rho / pq
//   pcg_b2 (many)          x += alpha p; r -= alpha q (or r = b - S x on a residual
//                          reset); z = M^-1 r with one wave per reduced block (the
//                          block's r is exchanged by wave shuffles); partial sums of
//                          Q1 = -x.(b + r) and rho' = r.z
//   pcg_b3 (1 workgroup)   Q1, zeta = it (Q1 - Q0) / Q1, rho', beta, p = z + beta p
// mode 0: regular; mode 1: update x only (before the reset product); mode 2: r = b - t.
__global__ __launch_bounds__(1024) void pcg_b1_kernel(DeviceView v, int n) {
  __shared__ double sh[16];
  double local = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) local += v.cg_p[i] * v.cg_q[i];
  const double pq = block1024_sum(local, sh);
  if (threadIdx.x == 0) {
    v.scal[SC_PQ] = pq;
    const double alpha = v.scal[SC_RHO] / pq;
    v.scal[SC_ALPHA] = alpha;
    if (pq > 0.0 && isfinite(pq) && !isfinite(alpha)) v.flags[FL_PCG_FAIL] = 1;
  }
}

template <int D>
__global__ __launch_bounds__(256) void pcg_b2_kernel(DeviceView v, const double* __restrict__ b, int mode,
                                                     int nblocks, double* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int rb = blockIdx.x * 4 + (threadIdx.x >> 6);
  const double pq = v.scal[SC_PQ];
  double acc[2] = {0.0, 0.0};
  // LINEAR_SOLVER_NO_CONVERGENCE (pq <= 0): nothing moves, the host stops
  if (rb < v.Nrb && pq > 0.0 && isfinite(pq)) {
    const double alpha = v.scal[SC_ALPHA];
    const int i = rb * D + lane;
    double rn = 0.0;
    if (lane < D) {
      double x = v.yc[i];
      if (mode != 2) {
        x += alpha * v.cg_p[i];
        v.yc[i] = x;
      }
      if (mode == 0) {
        rn = v.cg_r[i] - alpha * v.cg_q[i];
        v.cg_r[i] = rn;
      } else if (mode == 2) {
        rn = b[i] - v.cg_t[i];
        v.cg_r[i] = rn;
      }
      if (mode != 1) acc[0] = -x * (b[i] + rn);
    }
    if (mode != 1) {
      double z = 0.0;
      const double* M = v.Minv + (size_t)rb * D * D + (lane < D ? lane : 0) * D;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        const double rc = __shfl(rn, c, 64);
        if (lane < D) z += M[c] * rc;
      }
      if (lane < D) {
        v.cg_z[i] = z;
        acc[1] = rn * z;
      }
    }
  }
  block_sum_store<2>(acc, partial, nblocks);
}

__global__ __launch_bounds__(1024) void pcg_b3_kernel(DeviceView v, int n, int it, int nblocks,
                                                      const double* __restrict__ partial) {
  __shared__ double sh[16];
  double l0 = 0.0, l1 = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 1024) {
    l0 += partial[i];
    l1 += partial[nblocks + i];
  }
  const double Q1 = block1024_sum(l0, sh);
  const double rho = block1024_sum(l1, sh);
  const double last_rho = v.scal[SC_RHO];
  const double pq = v.scal[SC_PQ];
  __syncthreads();
  if (!(pq > 0.0) || !isfinite(pq)) {
    if (threadIdx.x == 0) v.scal[SC_ZETA] = -1.0;
    return;
  }
  if (threadIdx.x == 0) {
    const double Q0 = v.scal[SC_Q0];
    v.scal[SC_Q1] = Q1;
    v.scal[SC_ZETA] = it * (Q1 - Q0) / Q1;
    v.scal[SC_Q0] = Q1;
    v.scal[SC_LAST_RHO] = last_rho;
    v.scal[SC_RHO] = rho;
    v.scal[SC_RHO_BAD] = (rho == 0.0 || !isfinite(rho) || !isfinite(rho / last_rho)) ? 1.0 : 0.0;
  }
  const double beta = rho / last_rho;
  for (int i = threadIdx.x; i < n; i += 1024) v.cg_p[i] = v.cg_z[i] + beta * v.cg_p[i];
}

// One PCG iteration after q = S p (replaces pcg_b1 + pcg_b2 + pcg_b3 + publish when the product
// kernel delivered p.q behind the product vector): pcg_step -- every workgroup forms alpha itself,
// updates x, r and z = M^-1 r of its blocks and hands over its partial sums of Q1 and rho'; the last
// workgroup to arrive finishes the sums in a fixed order, does the scalar recurrences, decides
// whether PCG stops, forms p = z + beta p for the next product (round 6: until then a launch of its own, pcg_p --
// z is handed over like the partial sums: agent-scope stores by the workgroups that form it, agent-scope loads by the
// last one, see the hand-over note at the top; p and its scaled copy then cross the kernel boundary to the product)
// and publishes the scalars to the host mirror.
constexpr int kPcgStepThreads = 1024;
template <int D>
__global__ __launch_bounds__(kPcgStepThreads) void pcg_step_kernel(DeviceView v, const double* __restrict__ b, int it,
                                                       int nblocks, double eta, int min_it, int max_it,
                                                       const double* __restrict__ red8, HostMirror* mirror,
                                                       unsigned long long seq, const int* guard) {
  constexpr int T = kPcgStepThreads, W = T / 64;
  // (guard: a step enqueued before the host had read the previous one's stopping test -- nothing to do once that held)
  if (guard && *guard) return;
  __shared__ double sh[2][T];
  __shared__ double shw[2][W];
  __shared__ double pub[SC_COUNT];
  __shared__ int pubf[FL_COUNT];
  __shared__ int last;
  __shared__ double beta_sh;
  __shared__ int stop_sh;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rb = blockIdx.x * W + wv;
  const int n = v.Nrb * D;
  const double pq = v.cg_q[n];
  const bool ok = pq > 0.0 && isfinite(pq);
  const double rho = v.scal[SC_RHO];
  const double alpha = rho / pq;
  double acc[2] = {0.0, 0.0};
  if (rb < v.Nrb && ok) {
    const int i = rb * D + lane;
    double rn = 0.0;
    double z = 0.0;
    if (lane < D) {
      const double x = v.yc[i] + alpha * v.cg_p[i];
      v.yc[i] = x;
      rn = v.cg_r[i] - alpha * v.cg_q[i];
      v.cg_r[i] = rn;
      acc[0] = -x * (b[i] + rn);
    }
    const double* M = v.Minv + (size_t)rb * D * D + (lane < D ? lane : 0) * D;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      const double rc = __shfl(rn, c, 64);
      if (lane < D) z += M[c] * rc;
    }
    if (lane < D) {
      st_agent(&v.cg_z[i], z);  // (read by the last workgroup below)
      acc[1] = rn * z;
    }
    if constexpr (D == 9) {
      // compact planes: the transformed block of z, by the wavefront that formed it -- the map is linear, so the
      // transformed p = z + beta p is xz + beta xs, element by element (like p itself)
      if (v.compact) compact_forward_wave(v, rb, z, lane, v.xz);
    }
  }
  {
    const double s0 = wave_sum(acc[0]), s1 = wave_sum(acc[1]);
    if (lane == 0) {
      shw[0][wv] = s0;
      shw[1][wv] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double t0 = 0.0, t1 = 0.0;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        t0 += shw[0][k];
        t1 += shw[1][k];
      }
      st_agent(&v.partial[blockIdx.x], t0);
      st_agent(&v.partial[(size_t)nblocks + blockIdx.x], t1);
      last = take_ticket(v.ticket + 1 * kTicketStride, nblocks) ? 1 : 0;
    }
    __syncthreads();
  }
  if (!last) return;
  double l0 = 0.0, l1 = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += T) {
    l0 += ld_agent(&v.partial[i]);
    l1 += ld_agent(&v.partial[(size_t)nblocks + i]);
  }
  sh[0][threadIdx.x] = l0;
  sh[1][threadIdx.x] = l1;
  if (threadIdx.x < SC_COUNT) pub[threadIdx.x] = v.scal[threadIdx.x];
  if (threadIdx.x < FL_COUNT) pubf[threadIdx.x] = v.flags[threadIdx.x];
  __syncthreads();
  for (int o = T / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double Q1 = sh[0][0], rho_new = sh[1][0];
    auto set = [&](int slot, double val) {
      v.scal[slot] = val;
      pub[slot] = val;
    };
    set(SC_PQ, pq);
    set(SC_ALPHA, alpha);
    if (ok && !isfinite(alpha)) {
      v.flags[FL_PCG_FAIL] = 1;
      pubf[FL_PCG_FAIL] = 1;
    }
    double zeta = -1.0, rho_bad = 0.0;
    if (ok) {
      const double Q0 = v.scal[SC_Q0];
      zeta = it * (Q1 - Q0) / Q1;
      set(SC_Q1, Q1);
      set(SC_Q0, Q1);
      set(SC_LAST_RHO, rho);
      set(SC_RHO, rho_new);
      rho_bad = (rho_new == 0.0 || !isfinite(rho_new) || !isfinite(rho_new / rho)) ? 1.0 : 0.0;
      set(SC_RHO_BAD, rho_bad);
    }
    set(SC_ZETA, zeta);
    // the host's stopping rules (solve_reduced_pcg): pcg_p, enqueued behind this kernel, has nothing to do once they hold
    const bool stop = pubf[FL_PCG_FAIL] != 0 || !ok || (zeta < eta && it >= min_it) || it >= max_it || rho_bad != 0.0;
    *v.pcg_done = stop ? 1 : 0;
    set(SC_PCG_STOP, stop ? 1.0 : 0.0);
    stop_sh = stop ? 1 : 0;
    beta_sh = ok ? rho_new / rho : 0.0;
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (!stop_sh) {
    // p = z + beta p, beta = rho' / rho, and (drop_pos) the copy the next product gathers: position entries times the
    // column scales (pos_scale_kernel's job for any other vector).  Eight elements per thread and trip, every load of a
    // trip issued before the first use: the agent-scope loads of z cost a round trip to the L2 each, and the element-by-
    // element loop paid sixteen of them one after the other (31 -> 27 us per PCG iteration at Venice size).
    // (Measured and NOT kept: every workgroup forming the p of its own blocks after a bounded wait for the last
    //  workgroup's beta -- 26 us, but launches of several handles sharing a device can fill it with waiting workgroups.)
    const double beta = beta_sh;
    bool cpx = false;
    if constexpr (D == 9) cpx = v.compact != 0;
    constexpr int U = 8;
    for (int i0 = t; i0 < n; i0 += U * T) {
      double zz[U], pp[U], gg[U], xx[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = min(i0 + u * T, n - 1);
        zz[u] = ld_agent(&v.cg_z[i]);
        pp[u] = v.cg_p[i];
        gg[u] = cpx ? ld_agent(&v.xz[i]) : 0.0;
        xx[u] = cpx ? v.xs[i] : (v.drop_pos ? v.scale_c[i] : 0.0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = i0 + u * T;
        if (i < n) {
          const double pn = zz[u] + beta * pp[u];
          v.cg_p[i] = pn;
          if (cpx) v.xs[i] = gg[u] + beta * xx[u];  // (compact planes: xs = the transformed p, see above)
          else if (v.drop_pos) v.xs[i] = (i % D) < 3 ? pn * xx[u] : pn;
        }
      }
    }
  }
  // publish (what publish_kernel does)
  if (t < SC_COUNT) mirror->scal[t] = pub[t];
  if (t < 8) mirror->red[t] = red8[t];
  if (t < FL_COUNT) mirror->flags[t] = pubf[t];
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(&mirror->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Start of a PCG solve in one launch (pcg_begin + the first pcg_a): x = 0, r = b, z = M^-1 b,
// p = z, rho = r.z finished by the last workgroup, which also resets the PCG state.
template <int D>
__global__ __launch_bounds__(kPcgStepThreads) void pcg_init_kernel(DeviceView v, const double* __restrict__ b,
                                                                   int nblocks) {
  constexpr int T = kPcgStepThreads, W = T / 64;
  __shared__ double sh[T];
  __shared__ double shw[W];
  __shared__ int last;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int rb = blockIdx.x * W + wv;
  double acc = 0.0;
  if (rb < v.Nrb) {
    const int i = rb * D + lane;
    double rn = 0.0;
    if (lane < D) {
      rn = b[i];
      v.yc[i] = 0.0;
      v.cg_r[i] = rn;
    }
    double z = 0.0;
    const double* M = v.Minv + (size_t)rb * D * D + (lane < D ? lane : 0) * D;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      const double rc = __shfl(rn, c, 64);
      if (lane < D) z += M[c] * rc;
    }
    bool cpx = false;
    if constexpr (D == 9) cpx = v.compact != 0;
    if (lane < D) {
      v.cg_z[i] = z;
      v.cg_p[i] = z;
      if (v.drop_pos && !cpx) v.xs[i] = lane < 3 ? z * v.scale_c[i] : z;
      acc = rn * z;
    }
    if constexpr (D == 9) {
      if (cpx) compact_forward_wave(v, rb, z, lane, v.xs);
    }
  }
  const double s0 = wave_sum(acc);
  if (lane == 0) shw[wv] = s0;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < W; ++k) t += shw[k];
    st_agent(&v.partial[blockIdx.x], t);
    last = take_ticket(v.ticket + 1 * kTicketStride, nblocks) ? 1 : 0;
  }
  __syncthreads();
  if (!last) return;
  double l0 = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += T) l0 += ld_agent(&v.partial[i]);
  sh[threadIdx.x] = l0;
  __syncthreads();
  for (int o = T / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double rho = sh[0];
    v.scal[SC_LAST_RHO] = 1.0;
    v.scal[SC_RHO] = rho;
    v.scal[SC_Q0] = 0.0;
    v.flags[FL_PCG_FAIL] = (rho == 0.0 || !isfinite(rho)) ? 1 : 0;
    *v.pcg_done = 0;
  }
}

// ------------------------------------------------------------------------------
// Shared intrinsics blocks (free intrinsics used by several views; kernel class 2).
// camera_group_partials: one wave per view of a shared block walks the view's records
//   and forms  C = sum A0^T A1 (cross block),  G = sum A1^T A1,  sum A1^T r~,
//   sum A1^T r,  diag(G).
// group_reduce: adds the views' G / gradients to the shared block's diagonal entries of
//   `red` (fixed order), cross_add puts C into the (view, shared block) upper block.
// cam_part[view block] = [C (D^2) | G (D^2) | gt (D) | gc (D) | ud (D)]
// ------------------------------------------------------------------------------
template <int D, int DP, typename RS = double>
__global__ __launch_bounds__(64) void camera_group_partials_kernel(DeviceView v) {
  constexpr int AS = as_of(D, true);
  const int rb = blockIdx.x;
  const int cam = v.rb_cam[rb];
  double* out = v.cam_part + (size_t)rb * (2 * D * D + 3 * D);
  if (cam < 0 || v.cam_grb[cam] < 0) return;
  {
    double C[D][D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) C[a][b] = 0.0;
    for (int s = v.cam_ptr[rb] + threadIdx.x; s < v.cam_ptr[rb + 1]; s += 64) {
      const RS* rec = reinterpret_cast<const RS*>(v.cm_A) + (size_t)s * AS;
      double A0[2][D], A1[2][D];
#pragma unroll
      for (int a = 0; a < D; ++a) {
        A0[0][a] = rec[a];
        A0[1][a] = rec[D + a];
        A1[0][a] = rec[sh_off_a1(D) + a];
        A1[1][a] = rec[sh_off_a1(D) + D + a];
      }
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < D; ++b) C[a][b] += A0[0][a] * A1[0][b] + A0[1][a] * A1[1][b];
    }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < D; ++b) {
        const double t = wave_sum(C[a][b]);
        if (threadIdx.x == 0) out[a * D + b] = t;
      }
  }
  {
    double G[sym_size(D)], gt[D], gc[D], ud[D];
#pragma unroll
    for (int i = 0; i < sym_size(D); ++i) G[i] = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) gt[a] = gc[a] = ud[a] = 0.0;
    for (int s = v.cam_ptr[rb] + threadIdx.x; s < v.cam_ptr[rb + 1]; s += 64) {
      const RS* rec = reinterpret_cast<const RS*>(v.cm_A) + (size_t)s * AS;
      double A1[2][D];
#pragma unroll
      for (int a = 0; a < D; ++a) {
        A1[0][a] = rec[sh_off_a1(D) + a];
        A1[1][a] = rec[sh_off_a1(D) + D + a];
      }
      const double rt0 = rec[sh_off_rt(D)], rt1 = rec[sh_off_rt(D) + 1], r0 = rec[sh_off_r(D)], r1 = rec[sh_off_r(D) + 1];
#pragma unroll
      for (int a = 0; a < D; ++a) {
#pragma unroll
        for (int b = a; b < D; ++b) G[sym_idx(a, b, D)] += A1[0][a] * A1[0][b] + A1[1][a] * A1[1][b];
        gt[a] += A1[0][a] * rt0 + A1[1][a] * rt1;
        gc[a] += A1[0][a] * r0 + A1[1][a] * r1;
        ud[a] += A1[0][a] * A1[0][a] + A1[1][a] * A1[1][a];
      }
    }
    double* g = out + D * D;
#pragma unroll
    for (int a = 0; a < D; ++a) {
#pragma unroll
      for (int b = a; b < D; ++b) {
        const double t = wave_sum(G[sym_idx(a, b, D)]);
        if (threadIdx.x == 0) {
          g[a * D + b] = t;
          g[b * D + a] = t;
        }
      }
      const double t1 = wave_sum(gt[a]), t2 = wave_sum(gc[a]), t3 = wave_sum(ud[a]);
      if (threadIdx.x == 0) {
        out[2 * D * D + a] = t1;
        out[2 * D * D + D + a] = t2;
        out[2 * D * D + 2 * D + a] = t3;
      }
    }
  }
}

template <int D>
__global__ __launch_bounds__(256) void group_reduce_kernel(DeviceView v, RedLayout L) {
  const int gi = blockIdx.x;          // shared block index - Ncam_rb
  const int grb = v.Ncam_rb + gi;
  constexpr int NE = D * D + 3 * D;
  for (int e = threadIdx.x; e < NE; e += 256) {
    double acc = 0.0;
    for (int k = v.grp_cam_ptr[gi]; k < v.grp_cam_ptr[gi + 1]; ++k) {
      const int rb = v.cam_rb[v.grp_cams[k]];
      acc += v.cam_part[(size_t)rb * (2 * D * D + 3 * D) + D * D + e];
    }
    if (e < D * D)
      v.red[L.diag + (size_t)grb * D * D + e] += acc;
    else if (e < D * D + D)
      v.red[L.gt + (size_t)grb * D + (e - D * D)] += acc;
    else if (e < D * D + 2 * D)
      v.red[L.gc + (size_t)grb * D + (e - D * D - D)] += acc;
    else
      v.red[L.udiag + (size_t)grb * D + (e - D * D - 2 * D)] += acc;
  }
}

template <int D>
__global__ __launch_bounds__(256) void cross_add_kernel(DeviceView v, RedLayout L) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= v.Ncam_rb * D * D) return;
  const int rb = e / (D * D);
  const int cam = v.rb_cam[rb];
  const int u = v.cam_cross_u[cam];
  if (u < 0) return;
  v.red[L.ub + (size_t)u * D * D + (e - rb * D * D)] += v.cam_part[(size_t)rb * (2 * D * D + 3 * D) + (e - rb * D * D)];
}

// ------------------------------------------------------------------------------
// back_substitute (kernel class 8): y_p = (V+Dp)^-1 (g_p - W^T y_c) per track and
// the model cost change -(J d).(r + J d / 2), d = -y (TrustRegionMinimizer), in ONE sweep over
// the track's observations.  With u_i = A_i y_c and t_i = u_i + Jp_i y_p = -(J d)_i:
//   sum_i t_i.r_i - |t_i|^2 / 2
//     = sum_i (u_i.r_i - |u_i|^2 / 2)  +  y_p.g_p - y_p.(sum_i Jp_i^T u_i) - y_p^T V y_p / 2
// -- g_p and the undamped V = sum Jp^T Jp are point_eliminate's, sum Jp^T u is g_p - w -- so the
// second sweep (u written and read back, Jp and r read again) is not needed.
// ------------------------------------------------------------------------------
template <int D, int DP, bool SH, typename PT = double>
__global__ __launch_bounds__(256) void back_substitute_kernel(DeviceView v, int nblocks, double* partial,
                                                              double* __restrict__ sums) {
  constexpr int NS = sym_size(DP);
  PT* const pmR = reinterpret_cast<PT*>(v.pm_r);   // (the planes in their storage type: double, or float with
  PT* const pmA = reinterpret_cast<PT*>(v.pm_A);   //  fp32 evaluation on shared-intrinsics problems -- DeviceView::planes_fp32)
  PT* const pmA1 = reinterpret_cast<PT*>(v.pm_A1);
  PT* const pmJp = reinterpret_cast<PT*>(v.pm_Jp);
  (void)pmR; (void)pmA; (void)pmA1; (void)pmJp;
  const TrackMap tm = track_map(v);
  double acc[3] = {0.0, 0.0, 0.0};
  if (tm.valid) {
    const int lp = tm.lp;
    const int k = tm.k;
    const size_t base = tm.base;
    const size_t NP = (size_t)v.Np_pad;
    double ypv[DP];
#pragma unroll
    for (int a = 0; a < DP; ++a) ypv[a] = 0.0;
    if (k > 0) {
      double w[DP];
#pragma unroll
      for (int a = 0; a < DP; ++a) w[a] = tm.leader ? v.gp[(size_t)a * NP + lp] : 0.0;
      double ur = 0.0, uu = 0.0;
      double pc[3] = {0.0, 0.0, 0.0};  // drop_pos: -w / scale_p of this track (device_view.h)
      if (!SH && v.drop_pos && !v.compact) {
#pragma unroll
        for (int a = 0; a < 3; ++a) pc[a] = v.pos_coef[(size_t)a * NP + lp];
      }
      // compact planes (device_view.h): the track's {X, w} and 1 / scale_p of the linearisation
      double Xt[4] = {0.0, 0.0, 0.0, 0.0}, ispt[3] = {0.0, 0.0, 0.0};
      bool cpx = false;
      if constexpr (!SH && D == 9) {
        if (v.compact) {
          cpx = true;
#pragma unroll
          for (int a = 0; a < 4; ++a) Xt[a] = v.cp_trk[(size_t)a * NP + lp];
#pragma unroll
          for (int a = 0; a < 3; ++a) ispt[a] = v.cp_trk[(size_t)(4 + a) * NP + lp];
        }
      }
      // (the block index of the next observation is in flight while this one is processed)
      int rb_next = (tm.j0 < k) ? v.obs_rb[base + (size_t)tm.j0 * 64] : -1;
      for (int j = tm.j0; j < k; j += tm.jstep) {
        const size_t e = base + (size_t)j * 64;
        const int rb = rb_next;
        if (j + tm.jstep < k) rb_next = v.obs_rb[e + (size_t)tm.jstep * 64];
        const int cam = SH ? v.obs_cam[e] : 0;
        double u0 = 0.0, u1 = 0.0;
        if (rb >= 0) {
          if (!SH && v.drop_pos) {
            // the position columns from Jp (device_view.h); xs = y_c with the position entries times scale_c
            // (the view's block of y_c: D doubles at an 8-byte aligned address, gathered 16 bytes at a time -- five
            // instructions of 64 different lines instead of nine)
            const double* ycp = v.xs + (size_t)rb * D;
            double yv[D];
#pragma unroll
            for (int a = 0; a + 1 < D; a += 2) {
              const double2_a8 t2 = *reinterpret_cast<const double2_a8*>(ycp + a);
              yv[a] = t2.x;
              yv[a + 1] = t2.y;
            }
            if (D & 1) yv[D - 1] = ycp[D - 1];
            if constexpr (!SH && D == 9) {
              if (cpx) {
                // xs = the transformed block of y_c (update_cameras): A y_c from Jp, p_n and the track
                const double2 j3[3] = {make_double2((double)pmJp[pidx<2 * DP>(0, e)], (double)pmJp[pidx<2 * DP>(1, e)]),
                                       make_double2((double)pmJp[pidx<2 * DP>(2, e)], (double)pmJp[pidx<2 * DP>(3, e)]),
                                       make_double2((double)pmJp[pidx<2 * DP>(4, e)], (double)pmJp[pidx<2 * DP>(5, e)])};
                if (v.compact == 2) {  // a robust loss: [C p_n | r^2 .] per observation
                  const double2 pn = make_double2((double)pmA[pidx<4>(0, e)], (double)pmA[pidx<4>(1, e)]);
                  compact_ax(yv, Xt, ispt, j3, pn, u0, u1, (double)pmA[pidx<4>(2, e)]);
                } else {
                  const double2 pn = make_double2((double)pmA[pidx<2>(0, e)], (double)pmA[pidx<2>(1, e)]);
                  compact_ax(yv, Xt, ispt, j3, pn, u0, u1);
                }
              }
            }
#pragma unroll
            for (int a = 0; a < D && !cpx; ++a) {
              const double ya = yv[a];
              if (a < 3) {
                const double t = ya * pc[a];
                u0 += pmJp[pidx<2 * DP>((2 * (a < DP ? a : 0)), e)] * t;
                u1 += pmJp[pidx<2 * DP>((2 * (a < DP ? a : 0) + 1), e)] * t;
              } else {
                u0 += pmA[pidx<2 * D>((2 * a), e)] * ya;
                u1 += pmA[pidx<2 * D>((2 * a + 1), e)] * ya;
              }
            }
          } else {
            const double* yc = v.yc + (size_t)rb * D;
#pragma unroll
            for (int a = 0; a < D; ++a) {
              const double ya = yc[a];
              u0 += pmA[pidx<2 * D>((2 * a), e)] * ya;
              u1 += pmA[pidx<2 * D>((2 * a + 1), e)] * ya;
            }
          }
        }
        if (SH) {
          const int grb = v.cam_grb[cam];
          if (grb >= 0) {
            const double* yg = v.yc + (size_t)grb * D;
#pragma unroll
            for (int a = 0; a < D; ++a) {
              const double ya = yg[a];
              u0 += pmA1[pidx<2 * D>((2 * a), e)] * ya;
              u1 += pmA1[pidx<2 * D>((2 * a + 1), e)] * ya;
            }
          }
        }
        ur += u0 * pmR[pidx<2>(0, e)] + u1 * pmR[pidx<2>(1, e)];
        uu += u0 * u0 + u1 * u1;
#pragma unroll
        for (int a = 0; a < DP; ++a)
          w[a] -= pmJp[pidx<2 * DP>((2 * a), e)] * u0 + pmJp[pidx<2 * DP>((2 * a + 1), e)] * u1;
      }
      acc[0] = ur - 0.5 * uu;
#pragma unroll
      for (int a = 0; a < DP; ++a) w[a] = group_sum(w[a], tm.wide);
      if (tm.leader) {
        double Vi[NS], Vr[NS], yp[DP];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          Vi[i] = v.Vinv[(size_t)i * NP + lp];
          Vr[i] = v.Vraw[(size_t)i * NP + lp];
        }
#pragma unroll
        for (int a = 0; a < DP; ++a) {
          double t = 0.0;
#pragma unroll
          for (int b = 0; b < DP; ++b) t += Vi[a <= b ? sym_idx(a, b, DP) : sym_idx(b, a, DP)] * w[b];
          yp[a] = t;
          ypv[a] = t;
          v.yp[(size_t)a * NP + lp] = t;
        }
        double lin = 0.0, quad = 0.0;
#pragma unroll
        for (int a = 0; a < DP; ++a) {
          // y_p.g_p - y_p.(g_p - w) = y_p.w
          lin += yp[a] * w[a];
          double t = 0.0;
#pragma unroll
          for (int b = 0; b < DP; ++b) t += Vr[a <= b ? sym_idx(a, b, DP) : sym_idx(b, a, DP)] * yp[b];
          quad += yp[a] * t;
        }
        acc[0] += lin - 0.5 * quad;
      }
    }
    if (tm.leader) {
      // the candidate point x - scale .* y on the free coordinates and the tracks' share of |step|^2 and |x+|^2
      // (round 6: update_points_kernel's job, here where y_p is in registers -- one launch and a sweep over
      // y_p / the points less per LM iteration; a track without observations or a constant point is copied)
      const bool live = k > 0 && !v.pt_const[lp];
      const double2 x01 = *reinterpret_cast<const double2*>(v.pts + (size_t)lp * 4);
      const double2 x23 = *reinterpret_cast<const double2*>(v.pts + (size_t)lp * 4 + 2);
      double x[4] = {x01.x, x01.y, x23.x, x23.y};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (live && a < DP) {
          const double d = -ypv[a < DP ? a : 0] * v.scale_p[(size_t)lp * DP + (a < DP ? a : 0)];
          x[a] += d;
          acc[1] += d * d;
        }
        if (live) acc[2] += x[a] * x[a];
      }
      *reinterpret_cast<double2*>(v.pts_c + (size_t)lp * 4) = make_double2(x[0], x[1]);
      *reinterpret_cast<double2*>(v.pts_c + (size_t)lp * 4 + 2) = make_double2(x[2], x[3]);
    }
  }
  // sums[0] = model cost change, sums[1] = |step|^2 over the points, sums[2] = |x+|^2 over the points
  block_sum_finish<3>(acc, partial, nblocks, v.ticket + 2 * kTicketStride, sums);
}

// ------------------------------------------------------------------------------
// update (kernel class 9): candidate = x - scale .* y on the free coordinates.
// Points: by back_substitute_kernel, where y_p is in registers (partial sums [step^2, |x+|^2] for the tracks).
// Cameras: update_cameras_kernel (the camera part is replicated on every rank).
// ------------------------------------------------------------------------------
// |x|^2 over the live tracks of a parameter set
__global__ __launch_bounds__(256) void points_norm_kernel(DeviceView v, const double* __restrict__ pts,
                                                          int nblocks, double* partial,
                                                          double* __restrict__ sums) {
  const int lp = blockIdx.x * 256 + threadIdx.x;
  double acc[1] = {0.0};
  if (lp < v.Np_pad && v.pt_k[lp] > 0 && !v.pt_const[lp]) {
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[0] += pts[(size_t)lp * 4 + a] * pts[(size_t)lp * 4 + a];
  }
  block_sum_finish<1>(acc, partial, nblocks, v.ticket + 2 * kTicketStride, sums);
}

// Candidate cameras in ONE multi-workgroup launch (round 1: two copy commands + a single
// 1024-thread workgroup, 47 us): thread c < Nc owns view c -- ext_c = ext + d on its free
// extrinsics, intr_c = intr + d on its PRIVATE free intrinsics (d = -y * scale), a constant
// group is copied by each of its views (identical values) -- threads beyond own the shared
// intrinsics blocks.  out[0] = |step|^2 (cameras), out[1] = |x+|^2 over every coordinate of every
// non-constant camera-side block, finished by the last workgroup.  Without shared intrinsics the
// thread also writes the candidate's prepared record (camera_models.h), saving the separate
// camera_prepare launch.  Column of parameter bit a inside its block = popcount of the lower free bits
// (rb_cols is filled in increasing bit order, structure.cpp).
template <int D, bool SH>
__global__ __launch_bounds__(256) void update_cameras_kernel(DeviceView v, double* __restrict__ out,
                                                             double* __restrict__ prep_c) {
  __shared__ double uc_sh[2][4];
  __shared__ int uc_last;
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int n_groups_sh = v.Nrb - v.Ncam_rb;
  double step = 0.0, xn = 0.0;
  if (c < v.Nc) {
    const unsigned mask = v.cam_mask[c];
    const int rb = v.cam_rb[c];
    const int4 rec = v.cam_rec[c];
    double e[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double x = v.ext[(size_t)c * 6 + a];
      if (mask & (1u << a)) {
        const int col = __popc(mask & ((1u << a) - 1u));
        const double d = -v.yc[(size_t)rb * D + col] * v.scale_c[(size_t)rb * D + col];
        x += d;
        step += d * d;
      }
      e[a] = x;
      v.ext_c[(size_t)c * 6 + a] = x;
    }
    const bool shared_free = SH && v.cam_grb[c] >= 0;
    double K[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      double x = 0.0;
      if (j < rec.z) {
        x = v.intr[rec.y + j];
        if (mask & (1u << (6 + j))) {
          const int col = __popc(mask & ((1u << (6 + j)) - 1u));
          const double d = -v.yc[(size_t)rb * D + col] * v.scale_c[(size_t)rb * D + col];
          x += d;
          step += d * d;
        }
        if (!shared_free) v.intr_c[rec.y + j] = x;
      }
      K[j] = x;
    }
    if (rb >= 0) {
      if (mask & 0x3fu)
#pragma unroll
        for (int a = 0; a < 6; ++a) xn += e[a] * e[a];
      if (mask >> 6)
#pragma unroll
        for (int j = 0; j < 10; ++j) xn += K[j] * K[j];
    }
    if (!SH) prepare_camera_record(e, K, rec.z, v.scale_cam + (size_t)c * 16, prep_c + (size_t)c * kPrepStride);
    if (!SH && v.drop_pos && rb >= 0) {
      // drop_pos: the copy of y_c back_substitute gathers, position entries times the column scales (round 6:
      // pos_scale_kernel's job -- this launch runs BEFORE back_substitute and owns the view's block anyway)
      bool cpx = false;
      if constexpr (D == 9) {
        if (v.compact) {  // compact planes: the transformed block (at the linearisation's record, v.prep)
          double yv[9];
#pragma unroll
          for (int a = 0; a < 9; ++a) yv[a] = v.yc[(size_t)rb * 9 + a];
          compact_forward_view(v, rb, yv, v.xs);
          cpx = true;
        }
      }
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const size_t i = (size_t)rb * D + a;
        const double y = v.yc[i];
        if (!cpx) v.xs[i] = a < 3 ? y * v.scale_c[i] : y;
      }
    }
  } else if (SH && c < v.Nc + n_groups_sh) {
    const int grb = v.Ncam_rb + (c - v.Nc);
    const int g = v.rb_grp[grb];
    const unsigned gmask = v.grp_mask[g];
    const int o = v.grp_off[g], nk = v.grp_off[g + 1] - o;
    for (int j = 0; j < nk; ++j) {
      double x = v.intr[o + j];
      if (gmask & (1u << j)) {
        const int col = __popc(gmask & ((1u << j) - 1u));
        const double d = -v.yc[(size_t)grb * D + col] * v.scale_c[(size_t)grb * D + col];
        x += d;
        step += d * d;
      }
      v.intr_c[o + j] = x;
      xn += x * x;
    }
  }
  const double s0 = wave_sum(step), s1 = wave_sum(xn);
  if ((threadIdx.x & 63) == 0) {
    uc_sh[0][threadIdx.x >> 6] = s0;
    uc_sh[1][threadIdx.x >> 6] = s1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nb = (int)gridDim.x;
    st_agent(&v.dotbuf[blockIdx.x], uc_sh[0][0] + uc_sh[0][1] + uc_sh[0][2] + uc_sh[0][3]);
    st_agent(&v.dotbuf[nb + blockIdx.x], uc_sh[1][0] + uc_sh[1][1] + uc_sh[1][2] + uc_sh[1][3]);
    uc_last = take_ticket(v.ticket + 2 * kTicketStride, nb) ? 1 : 0;
  }
  __syncthreads();
  if (!uc_last) return;
  if (threadIdx.x == 0) {
    const int nb = (int)gridDim.x;
    double t0 = 0.0, t1 = 0.0;
    for (int i = 0; i < nb; ++i) {
      t0 += ld_agent(&v.dotbuf[i]);
      t1 += ld_agent(&v.dotbuf[nb + i]);
    }
    out[0] = t0;
    out[1] = t1;
  }
}

}  // namespace tmi
