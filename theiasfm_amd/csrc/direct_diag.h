// Camera side of the normal equations WITHOUT camera-major records (round 5).
//
// In a matrix-free LM iteration (the one-sweep product of mf_chunks.h) nothing but camera_diag reads the camera-major
// [A | Q] records + tails that point_eliminate writes: 1.47 GB written and 1.31 GB read back per LM iteration on the
// Venice-sized problem to obtain 0.77 MB of diagonal blocks, U diagonal, g~ and g_c (profiles/r04_e_summary.md:
// point_eliminate 493 us, camera_diag 243 us).  The records are a transposition -- per-track data (the factor of
// V + D_p, t_p) meets per-view sums -- and the transposition is cheaper done on the INPUTS than on the Jacobians:
//
//   * point_eliminate (DeviceView::direct_diag) stops after the per-track part and leaves ONE record per track,
//       trk_rec[lp] = { X (4), L^-1 diag(scale_p) (DP (DP + 1) / 2), scale_p . t_p (DP) }
//     (the Jacobi scales of the point block folded into the factor: Q = (Jp diag(s)) L^-T = Jp (L^-1 diag(s))^T and
//     Jp diag(s) t_p = Jp (s . t_p), so the scaled Jp is never formed; s = 0 for a constant point, whose Jp is zero)
//     -- 128 B for 3-dof points, 192 B for 4-dof -- 1 M tracks = 128 MB, resident in the Infinity Cache;
//   * camera_diag_direct walks a view's slots (one wavefront per chunk of a view): the pixel and the track of a slot
//     are static and streamed (20 B per observation), the track's record is gathered, and the observation's residual
//     and Jacobian blocks are RE-EVALUATED from the view's prepared record, which is wave-uniform here -- no staging of
//     64 different camera records per trip as in the track-major linearize, the camera-model switch does not diverge
//     -- followed by the same loss corrector, column scaling and compaction as linearize_kernel, Q = Jp L^-T,
//     N = I - Q Q^T, r~ = r - Jp t_p and the sums of camera_diag_kernel;
//   * a chunk leaves its partial sums, camera_diag_direct_reduce adds a view's chunks in chunk order (fixed order:
//     bit-reproducible) and writes the raw diagonal block, U diagonal, g~ and g_c where camera_diag_kernel does.
//
// Reference: this is the f-block diagonal of Ceres' SchurEliminator::Eliminate (called under ceres::Solve,
// bundle_adjuster.cc:205) for the cost functions of reprojection_error.h:51-95.
#pragma once
#include "kernels.h"

namespace tmi {
namespace ddg {

constexpr int kChunkSlots = 1024;  // slots of a view one wavefront walks (16 trips)
constexpr int kMaxD = 12;  // (16-wide blocks: S_cc, g~ and g_c do not fit one 16 x 16 accumulator; they keep the records)
#ifndef TMI_DD_EXP
#define TMI_DD_EXP 0  // timing experiments (1: no matrix-core loop, 2: track records from a cache-resident window, 3: no evaluation)
#endif
#ifdef TMI_DD_PROFILE
#define DDP(i) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof[i] += t_ - tprev; tprev = t_; }
#else
#define DDP(i)
#endif
#ifndef TMI_DD_WAVES
#define TMI_DD_WAVES 2  // wavefronts per SIMD the register allocation of camera_diag_direct must allow
#endif

__host__ __device__ constexpr int trk_stride(int DP) { return DP == 3 ? 16 : 24; }
__host__ __device__ constexpr int trk_off_li(int) { return 4; }
__host__ __device__ constexpr int trk_off_tp(int DP) { return 4 + sym_size(DP); }
__host__ __device__ constexpr int trk_used(int DP) { return (4 + sym_size(DP) + DP + 1) & ~1; }  // doubles of a record that are read
__host__ __device__ constexpr int n_acc(int D) { return sym_size(D) + 3 * D; }

struct Plan {
  int n_chunks;
  const int* chunk_rb;   // [n_chunks] view block of a chunk
  const int* chunk_s0;   // [n_chunks + 1] ... its slots [s0, s1)
  const int* chunk_s1;
  const int* rb_chunk;   // [Nrb + 1] chunks of a view block
  const double* cm_xy;   // [Nslots][2] pixel of a slot
  double* part;          // [n_chunks][n_acc(D)]
};

// slot -> (track, pixel), built once per structure
__global__ __launch_bounds__(256) void slot_gather_kernel(DeviceView v, int* __restrict__ slot_track,
                                                          double* __restrict__ cm_xy) {
  const TrackMap tm = track_map(v);
  if (!tm.valid) return;
  for (int j = tm.j0; j < tm.k; j += tm.jstep) {
    const size_t e = tm.base + (size_t)j * 64;
    const int cpos = v.obs_cpos[e];
    if (cpos >= 0) {
      slot_track[cpos] = tm.lp;
      *reinterpret_cast<double2*>(cm_xy + 2 * (size_t)cpos) = *reinterpret_cast<const double2*>(v.obs_xy + 2 * e);
    }
  }
}

// Start of a solve (round 6): the Jacobi scales of the camera columns need the U diagonal of the UNSCALED Jacobian, i.e.
// camera_diag_direct before any track has been eliminated -- its records then carry the point only (L^-1 = 0: Q = 0,
// N = I; what the launch leaves besides the U diagonal is not used).  Replaces a point_eliminate pass over planes that
// the norms-only linearize (kernels.h, NORMS) no longer writes.
template <int DP>
__global__ __launch_bounds__(256) void track_records_points_only_kernel(DeviceView v, const double* __restrict__ pts) {
  constexpr int TR = trk_stride(DP);
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // one 16-byte piece of a record per thread
  const long long n = (long long)v.Np_pad * (TR / 2);
  if (i >= n) return;
  const long long lp = i / (TR / 2);
  const int part = (int)(i - lp * (TR / 2));
  double2 val = make_double2(0.0, 0.0);
  if (part < 2) val = *reinterpret_cast<const double2*>(pts + lp * 4 + 2 * part);
  *reinterpret_cast<double2*>(v.trk_rec + lp * TR + 2 * part) = val;
}

// The sums run on v_mfma_f64_4x4x4f64: FOUR independent 4 x 4 x 4 products per instruction (16 cycles; the 16 x 16 x 4
// form takes 64 cycles and a 9 x 11 result would use 99 of its 256 outputs).  Layout found with tools/mfma_probe
// (lane = 16 g + 4 b + e): block b takes A_b[i = e][k = g] and B_b[k = g][j = e] and holds D_b[i = g][j = e].
// S_cc is symmetric and g~, g_c sit in the last column block(s), so only the 4 x 4 blocks (I, J >= I) of
// A^T [N A | r~ | r] are formed: 6 for D = 9, 3 for D = 6, 9 for D = 12.
__host__ __device__ constexpr int blk_rows(int D) { return (D + 3) / 4; }
__host__ __device__ constexpr int blk_cols(int D) { return (D + 2 + 3) / 4; }
__host__ __device__ constexpr int blk_count(int D) {
  int n = 0;
  for (int I = 0; I < blk_rows(D); ++I)
    for (int J = I; J < blk_cols(D); ++J) ++n;
  return n;
}
__host__ __device__ constexpr int blk_I(int t, int D) {
  int n = 0;
  for (int I = 0; I < blk_rows(D); ++I)
    for (int J = I; J < blk_cols(D); ++J) {
      if (n == t) return I;
      ++n;
    }
  return 0;
}
__host__ __device__ constexpr int blk_J(int t, int D) {
  int n = 0;
  for (int I = 0; I < blk_rows(D); ++I)
    for (int J = I; J < blk_cols(D); ++J) {
      if (n == t) return J;
      ++n;
    }
  return 0;
}
__host__ __device__ constexpr int gcd_c(int a, int b) { return b == 0 ? a : gcd_c(b, a % b); }

// LDS staging of one trip (64 slots) for the matrix cores: per slot the two rows of the compacted camera block A
// (2 x D) and of B' = [N A | r~ | r] (2 x (D + 2)); pitches are odd so that the per-slot writes (lane = slot) and the
// per-step operand reads (lane = (k, column)) both spread over the banks
template <int D>
struct StageDims {
  static constexpr int PA = 2 * D + 1;
  static constexpr int PB = 2 * (D + 2) + 1;
};

// UMODEL >= 0 (as linearize_kernel's): every camera is of model UMODEL with the free-column mask UMASK and the loss is
// TRIVIAL -- the model switch, the column compaction and the corrector fold at compile time
// (ULOSS: the specialised body with the loss corrector left in, as linearize_kernel's)
template <int D, int DP, int UMODEL = -1, unsigned UMASK = 0u, bool ULOSS = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(TMI_DD_WAVES, TMI_DD_WAVES))) void camera_diag_direct_kernel(DeviceView v, Plan pl, const double* __restrict__ prep,
                                                                int loss_type_arg, double loss_width) {
  const int loss_type = (UMODEL >= 0 && !ULOSS) ? 0 : loss_type_arg;
  static_assert(D + 2 <= 16, "one 16 x 16 accumulator holds S_cc (D x D), g~ and g_c");
  constexpr int NS = sym_size(D);
  constexpr int TR = trk_stride(DP);
  constexpr int TU = trk_used(DP);
  constexpr int NA = n_acc(D);
  constexpr int PA = StageDims<D>::PA, PB = StageDims<D>::PB;
  __shared__ double Ls[64 * PA + 64 * PB];  // staged A rows | staged B' rows
  double* const As = Ls;
  double* const Bs = Ls + 64 * PA;
  const int lane = threadIdx.x;
  const int ch = blockIdx.x;
  const int rb = pl.chunk_rb[ch];
  const int s0 = pl.chunk_s0[ch], s1 = pl.chunk_s1[ch];
  const int cam = v.rb_cam[rb];
  const int4 rec = v.cam_rec[cam];
  const unsigned mask = UMODEL >= 0 ? UMASK : (unsigned)__builtin_amdgcn_readfirstlane(rec.w);
  const int model = UMODEL >= 0 ? UMODEL : __builtin_amdgcn_readfirstlane(rec.x);
  // the view's prepared record: a wave-uniform address, so its words arrive through the scalar cache into SGPRs
  // (read ONCE, before the loop: behind the LDS fences of a trip the compiler would fetch every word again at its use,
  // a scalar-cache round trip each, with two wavefronts per SIMD to hide it).  The Jacobi scales of the position and
  // intrinsics columns [33..45] are not needed here: they are applied to the SUMS by the reduce launch
  // (S = diag(s) (sum A'^T N A') diag(s) for A = A' diag(s)).
  double Pl[kPrepStride];
  {
    const double* __restrict__ Pg = prep + (size_t)__builtin_amdgcn_readfirstlane(cam) * kPrepStride;
#pragma unroll
    for (int i = 0; i < 33; ++i) Pl[i] = Pg[i];
#pragma unroll
    for (int i = 33; i < kPrepStride; ++i) Pl[i] = 0.0;
  }
  // padding columns of the staged blocks stay zero for the whole launch
  for (int i = lane; i < 64 * PA + 64 * PB; i += 64) Ls[i] = 0.0;
  // Work items of a trip = (slot pair p, block t), item w = p NB + t goes to instruction w / 4, block w % 4; the
  // assignment repeats every PER instructions (SP slot pairs), so a lane keeps PER accumulators: acc[r] is element
  // (g, e) of block type (4 r + b) % NB, summed over the slot pairs that land there.
  constexpr int NB = blk_count(D);
  constexpr int LCM = NB * 4 / gcd_c(NB, 4);
  constexpr int PER = LCM / 4;   // instructions per period
  constexpr int SP = LCM / NB;   // slot pairs per period
  constexpr int NQ = 32 / SP;    // periods per trip
  static_assert(32 % SP == 0, "a trip is a whole number of periods");
  double acc[PER], usq[PER];
  int oA[PER], oB[PER];
  const int mg = lane >> 4, mb = (lane >> 2) & 3, me = lane & 3;
#pragma unroll
  for (int r = 0; r < PER; ++r) {
    acc[r] = 0.0;
    usq[r] = 0.0;
    const int w = 4 * r + mb, p0 = w / NB, t = w - p0 * NB;
    int I = 0, J = 0;
#pragma unroll
    for (int tt = 0; tt < NB; ++tt)
      if (t == tt) {
        I = blk_I(tt, D);
        J = blk_J(tt, D);
      }
    const int slot0 = 2 * p0 + (mg >> 1), row = mg & 1;
    const int ca = 4 * I + me, cb = 4 * J + me;
    // (an operand column that is padding reads the slot's pad word -- the pitch is one more than the two rows --,
    // which nobody writes: zero, and every lane steps by the same compile-time stride)
    oA[r] = slot0 * PA + (ca < D ? row * D + ca : 2 * D);
    oB[r] = 64 * PA + slot0 * PB + (cb < D + 2 ? row * (D + 2) + cb : 2 * (D + 2));
  }
  __syncthreads();

  double* Al = As + lane * PA;
  double* Bl = Bs + lane * PB;

  const int trips = (s1 - s0 + 63) >> 6;
  // software pipeline: the track record of trip t + 1 and the (track, pixel) of trip t + 2 are in flight during trip t
  int lp_n = -1;
  double2 xy_n = make_double2(0.0, 0.0);
  double Tn[TU];
#pragma unroll
  for (int i = 0; i < TU; ++i) Tn[i] = 0.0;
  {
    const int s = s0 + lane;
    if (s < s1) {
      lp_n = v.slot_track[s];
      xy_n = *reinterpret_cast<const double2*>(pl.cm_xy + 2 * (size_t)s);
      const double* tr = v.trk_rec + (size_t)(TMI_DD_EXP == 2 ? (lp_n & 1023) : lp_n) * TR;
#pragma unroll
      for (int i = 0; i < TU; i += 2) {
        const double2 t = *reinterpret_cast<const double2*>(tr + i);
        Tn[i] = t.x;
        Tn[i + 1] = t.y;
      }
    }
  }
  int lp_nn = -1;
  double2 xy_nn = make_double2(0.0, 0.0);
  if (trips > 1) {
    const int s = s0 + 64 + lane;
    if (s < s1) {
      lp_nn = v.slot_track[s];
      xy_nn = *reinterpret_cast<const double2*>(pl.cm_xy + 2 * (size_t)s);
    }
  }
#ifdef TMI_DD_PROFILE
  unsigned long long prof[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#endif
  for (int trip = 0; trip < trips; ++trip) {
    DDP(5)
    const bool act = lp_n >= 0;
    const double2 xy = xy_n;
    double T[TU];
#pragma unroll
    for (int i = 0; i < TU; ++i) T[i] = Tn[i];
    // next trip's record, the trip after's indices
    lp_n = lp_nn;
    xy_n = xy_nn;
    if (lp_n >= 0) {
      const double* tr = v.trk_rec + (size_t)(TMI_DD_EXP == 2 ? (lp_n & 1023) : lp_n) * TR;
#pragma unroll
      for (int i = 0; i < TU; i += 2) {
        const double2 t = *reinterpret_cast<const double2*>(tr + i);
        Tn[i] = t.x;
        Tn[i + 1] = t.y;
      }
    }
    lp_nn = -1;
    {
      const int s = s0 + (trip + 2) * 64 + lane;
      if (s < s1) {
        lp_nn = v.slot_track[s];
        xy_nn = *reinterpret_cast<const double2*>(pl.cm_xy + 2 * (size_t)s);
      }
    }
    DDP(0)
    // ---- residual and Jacobian blocks of this lane's observation (linearize_kernel's expressions) ----
    double r[2], Jext[2][6], Jint[2][10], Jpt[2][4];
    bool ok = false;
#if TMI_DD_EXP == 3
    ok = act;
    r[0] = T[0] - xy.x;
    r[1] = T[1] - xy.y;
#pragma unroll
    for (int j = 0; j < 6; ++j) { Jext[0][j] = T[j & 3]; Jext[1][j] = T[(j + 1) & 3]; }
#pragma unroll
    for (int j = 0; j < 10; ++j) { Jint[0][j] = T[j & 3] * 0.5; Jint[1][j] = T[(j + 1) & 3] * 0.25; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { Jpt[0][j] = T[j]; Jpt[1][j] = T[3 - j]; }
#else
    if (act) ok = reprojection_error_prepared<true, double>(model, Pl, T, xy.x, xy.y, r, Jext, Jint, Jpt);
#endif
    DDP(1)
    // an observation that does not count (past the chunk: the last trip only; |X - w C|^2 < 1e-8) contributes zero rows
    const double live = ok ? 1.0 : 0.0;
    if (!__all(ok)) {
      if (!ok) {
        r[0] = r[1] = 0.0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
          for (int j = 0; j < 6; ++j) Jext[i][j] = 0.0;
#pragma unroll
          for (int j = 0; j < 10; ++j) Jint[i][j] = 0.0;
#pragma unroll
          for (int j = 0; j < 4; ++j) Jpt[i][j] = 0.0;
        }
      }
    }
    double sqrt_rho1 = 1.0, asn = 0.0, rscale = 1.0;
    if (loss_type != 0 && ok) {
      const double sq = r[0] * r[0] + r[1] * r[1];
      double rho[3];
      loss_eval(loss_type, loss_width, sq, rho);
      sqrt_rho1 = sqrt(rho[1]);
      rscale = sqrt_rho1;
      if (!(sq == 0.0 || rho[2] <= 0.0)) {
        const double Dd = 1.0 + 2.0 * sq * rho[2] / rho[1];
        const double alpha = 1.0 - sqrt(Dd);
        rscale = sqrt_rho1 / (1.0 - alpha);
        asn = alpha / sq;
      }
    }
    double J0[DP], J1[DP];
#pragma unroll
    for (int a = 0; a < DP; ++a) {
      double j0 = Jpt[0][a < 4 ? a : 0], j1 = Jpt[1][a < 4 ? a : 0];
      if (loss_type != 0) {
        const double rtj = j0 * r[0] + j1 * r[1];
        j0 = sqrt_rho1 * (j0 - asn * r[0] * rtj);
        j1 = sqrt_rho1 * (j1 - asn * r[1] * rtj);
      }
      J0[a] = j0;  // (unscaled: the scales ride in the track's factor)
      J1[a] = j1;
    }
    const double r0 = r[0] * rscale, r1 = r[1] * rscale;
    // Q = Jp L^-T, N = I - Q Q^T, r~ = r - Jp t_p   (point_eliminate_kernel's second pass)
    double rt0 = r0, rt1 = r1;
#pragma unroll
    for (int a = 0; a < DP; ++a) {
      rt0 -= J0[a] * T[trk_off_tp(DP) + a];
      rt1 -= J1[a] * T[trk_off_tp(DP) + a];
    }
    double n00 = 1.0, n01 = 0.0, n11 = 1.0;
#pragma unroll
    for (int b = 0; b < DP; ++b) {
      double q0 = 0.0, q1 = 0.0;
#pragma unroll
      for (int a = 0; a <= b; ++a) {
        const double li = T[trk_off_li(DP) + sym_idx(a, b, DP)];  // L^-1 (b, a)
        q0 += li * J0[a];
        q1 += li * J1[a];
      }
      n00 -= q0 * q0;
      n01 -= q0 * q1;
      n11 -= q1 * q1;
    }
    // ---- the free columns of [ext(6) | intr(10)], compacted: column dst of the staged A and B' rows ----
    int dst = 0;  // (wave uniform)
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (mask & (1u << c)) {
        double j0, j1;  // (zero where the observation does not count: the blocks were cleared above)
        if (c < 6) {
          j0 = Jext[0][c < 6 ? c : 0];
          j1 = Jext[1][c < 6 ? c : 0];
        } else {
          j0 = Jint[0][c >= 6 ? c - 6 : 0];
          j1 = Jint[1][c >= 6 ? c - 6 : 0];
        }
        if (loss_type != 0) {
          const double rtj = j0 * r[0] + j1 * r[1];
          j0 = sqrt_rho1 * (j0 - asn * r[0] * rtj);
          j1 = sqrt_rho1 * (j1 - asn * r[1] * rtj);
        }
        Al[dst] = j0;
        Al[D + dst] = j1;
        Bl[dst] = n00 * j0 + n01 * j1;
        Bl[(D + 2) + dst] = n01 * j0 + n11 * j1;
        ++dst;
      }
    }
    DDP(2)
    Bl[D] = rt0 * live;
    Bl[D + 1] = r0 * live;
    Bl[(D + 2) + D] = rt1 * live;
    Bl[(D + 2) + D + 1] = r1 * live;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    DDP(3)
    // ---- sum over the 64 slots on the matrix cores: K = 4 per step = the two rows of two slots ----
#pragma unroll
    for (int q = 0; q < (TMI_DD_EXP == 1 ? 0 : NQ); ++q) {
      double a[PER], b[PER];
#pragma unroll
      for (int r = 0; r < PER; ++r) {
        a[r] = Ls[oA[r] + q * (2 * SP * PA)];
        b[r] = Ls[oB[r] + q * (2 * SP * PB)];
      }
#pragma unroll
      for (int r = 0; r < PER; ++r) {
        acc[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[r], b[r], acc[r], 0, 0, 0);
        usq[r] += a[r] * a[r];
      }
    }
    DDP(4)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#ifdef TMI_DD_PROFILE
  if (lane == 0 && (ch == 7 || ch == pl.n_chunks / 2 || ch == pl.n_chunks - 9))
    printf("[ddg profile] chunk %d trips %d: cycles per trip: loads %llu | eval %llu | Jp Q N columns %llu | stage + barrier %llu | matrix cores %llu | loop %llu\n",
           ch, trips, prof[0] / trips, prof[1] / trips, prof[2] / trips, prof[3] / trips, prof[4] / trips, prof[5] / trips);
#endif
  // ---- a lane's PER accumulators -> the chunk's NA sums.  Several (r, b) carry the same block type: they are added
  // in LDS in a fixed order (rounds over r, two sub-rounds over b so that no two lanes of a round share an entry).
  double* const Out = Ls;  // (the staging area is free now)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int i = lane; i < NA; i += 64) Out[i] = 0.0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int r = 0; r < PER; ++r) {
    // U diagonal: sum over k (the four g) of a lane's squares, diagonal block types only
    double u = usq[r];
    u += __shfl_xor(u, 16, 64);
    u += __shfl_xor(u, 32, 64);
    int bI = 0, bJ = 0;
    {
      const int w = 4 * r + mb, t = w % NB;
#pragma unroll
      for (int tt = 0; tt < NB; ++tt)
        if (t == tt) {
          bI = blk_I(tt, D);
          bJ = blk_J(tt, D);
        }
    }
    const int row = 4 * bI + mg, col = 4 * bJ + me;
    int idx = -1;
    if (row < D) {
      if (col < D) {
        if (row <= col) idx = sym_idx(row, col, D);
      } else if (col == D) {
        idx = NS + D + row;
      } else if (col == D + 1) {
        idx = NS + 2 * D + row;
      }
    }
    const int uidx = (bI == bJ && mg == 0 && col < D) ? NS + col : -1;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if ((mb < 3) == (half == 0)) {
        if (idx >= 0) Out[idx] += acc[r];
        if (uidx >= 0) Out[uidx] += u;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  double* out = pl.part + (size_t)ch * NA;
  for (int i = lane; i < NA; i += 64) out[i] = Out[i];
}

// ------------------------------------------------------------------------------
// Trial cost view by view (kernel class 9): the residual of reprojection_error.h:51-95 at the candidate, summed.
// The track-major cost_kernel gathers a 192-byte half record of a DIFFERENT camera for each lane of a trip (its time
// is the L2 gather); here the camera record is wave-uniform (SGPRs), a lane streams its slot's pixel and track index
// and gathers the 32-byte candidate point.  Needs every observation to own a slot (no fully constant camera).
// A workgroup = 4 wavefronts = 4 chunks of the plan; sums finished by the last workgroup as in cost_kernel.
// ------------------------------------------------------------------------------
// warm (Infinity Cache): the trial cost is followed -- when the step is accepted, the usual case -- by linearize, whose
// 0.8 GB of plane stores leave its own reads (the track-major pixels and camera indices, 20 B per observation) queueing
// behind a saturated write stream: 350 -> 404 us when nothing had touched them since the last iteration (the track-major
// cost_kernel used to, as a side effect).  Every wavefront here also reads its share of those two arrays, which this
// kernel's own gathers leave room for, so that linearize finds them in the memory-side cache.
__global__ __launch_bounds__(256) void cost_view_kernel(DeviceView v, Plan pl, const double* __restrict__ prep,
                                                        const double* __restrict__ pts, int loss_type, double loss_width,
                                                        int flag_slot, int nblocks, double* partial,
                                                        double* __restrict__ sums, double* __restrict__ flag_dst, int warm) {
  const int lane = threadIdx.x & 63;
  const int ch = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
  double acc[2] = {0.0, 0.0};
  double wsum = 0.0;
  if (warm && ch < pl.n_chunks) {
    // 16-byte pieces of obs_xy (2 doubles per observation) and obs_cam (1 int): this chunk's contiguous share
    const size_t n16 = (size_t)v.No_pad + (size_t)v.No_pad / 4;
    const size_t per = (n16 + pl.n_chunks - 1) / pl.n_chunks;
    const size_t b0 = (size_t)ch * per, b1 = b0 + per < n16 ? b0 + per : n16;
    const double2* xy16 = reinterpret_cast<const double2*>(v.obs_xy);
    const double2* cam16 = reinterpret_cast<const double2*>(v.obs_cam);
    for (size_t i = b0 + lane; i < b1; i += 64) {
      const double2 t = i < (size_t)v.No_pad ? xy16[i] : cam16[i - (size_t)v.No_pad];
      wsum += t.x;
    }
  }
  if (ch < pl.n_chunks) {
    const int rb = pl.chunk_rb[ch];
    const int s0 = pl.chunk_s0[ch], s1 = pl.chunk_s1[ch];
    const int cam = __builtin_amdgcn_readfirstlane(v.rb_cam[rb]);
    const int model = __builtin_amdgcn_readfirstlane(v.cam_rec[cam].x);
    double P[kPrepCostWords];
    {
      const double* __restrict__ Pg = prep + (size_t)cam * kPrepStride;
#pragma unroll
      for (int i = 0; i < kPrepCostWords; ++i) P[i] = Pg[i];
    }
    const int trips = (s1 - s0 + 63) >> 6;
    int lp_n = -1, lp_nn = -1;
    double2 xy_n = make_double2(0.0, 0.0), xy_nn = make_double2(0.0, 0.0);
    double Xn[4] = {0.0, 0.0, 0.0, 1.0};
    {
      const int s = s0 + lane;
      if (s < s1) {
        lp_n = v.slot_track[s];
        xy_n = *reinterpret_cast<const double2*>(pl.cm_xy + 2 * (size_t)s);
        const double2 a = *reinterpret_cast<const double2*>(pts + (size_t)lp_n * 4);
        const double2 b = *reinterpret_cast<const double2*>(pts + (size_t)lp_n * 4 + 2);
        Xn[0] = a.x; Xn[1] = a.y; Xn[2] = b.x; Xn[3] = b.y;
      }
      const int s2 = s + 64;
      if (s2 < s1) {
        lp_nn = v.slot_track[s2];
        xy_nn = *reinterpret_cast<const double2*>(pl.cm_xy + 2 * (size_t)s2);
      }
    }
    for (int trip = 0; trip < trips; ++trip) {
      const bool act = lp_n >= 0;
      const double2 xy = xy_n;
      const double X[4] = {Xn[0], Xn[1], Xn[2], Xn[3]};
      lp_n = lp_nn;
      xy_n = xy_nn;
      if (lp_n >= 0) {
        const double2 a = *reinterpret_cast<const double2*>(pts + (size_t)lp_n * 4);
        const double2 b = *reinterpret_cast<const double2*>(pts + (size_t)lp_n * 4 + 2);
        Xn[0] = a.x; Xn[1] = a.y; Xn[2] = b.x; Xn[3] = b.y;
      }
      lp_nn = -1;
      {
        const int s = s0 + (trip + 2) * 64 + lane;
        if (s < s1) {
          lp_nn = v.slot_track[s];
          xy_nn = *reinterpret_cast<const double2*>(pl.cm_xy + 2 * (size_t)s);
        }
      }
      if (!act) continue;
      double rr[2];
      double (*nul6)[6] = nullptr;
      double Jint[2][10];
      double (*nul4)[4] = nullptr;
      const bool ok = reprojection_error_prepared<false, double>(model, P, X, xy.x, xy.y, rr, nul6, Jint, nul4);
      if (!ok) {
        st_agent(&v.flags[flag_slot], 1);
        continue;
      }
      const double sq = rr[0] * rr[0] + rr[1] * rr[1];
      if (loss_type != 0) {
        double rho[3];
        loss_eval(loss_type, loss_width, sq, rho);
        acc[0] += 0.5 * rho[0];
      } else {
        acc[0] += 0.5 * sq;
      }
      acc[1] += sq;
    }
  }
  if (wsum == -1.2345678e300) acc[1] += wsum;  // (never: keeps the warming loads alive)
  block_sum_finish<2>(acc, partial, nblocks, v.ticket + 2 * kTicketStride, sums, v.flags + flag_slot, flag_dst);
}

// a view's chunks summed in chunk order; results where camera_diag_kernel leaves them
template <int D>
__global__ __launch_bounds__(64) void camera_diag_direct_reduce_kernel(DeviceView v, RedLayout L, Plan pl) {
  constexpr int NS = sym_size(D);
  constexpr int NA = n_acc(D);
  const int rb = blockIdx.x;
  const int c0 = pl.rb_chunk[rb], c1 = pl.rb_chunk[rb + 1];
  double* diag = v.red + L.diag + (size_t)rb * D * D;
  // Jacobi scale of a compacted column: the angle-axis columns carry theirs already (folded into Jl of the prepared
  // record, camera_models.h), position and intrinsics columns take scale_c
  __shared__ double f[D];
  if (threadIdx.x < D) {
    const int c = v.rb_cols[(size_t)rb * D + threadIdx.x];
    f[threadIdx.x] = (c >= 3 && c < 6) ? 1.0 : (c < 0 ? 0.0 : v.scale_c[(size_t)rb * D + threadIdx.x]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < NA; i += 64) {
    double t = 0.0;
    for (int c = c0; c < c1; ++c) t += pl.part[(size_t)c * NA + i];
    if (i < NS) {
      int a = 0, rem = i;
      while (rem >= D - a) {
        rem -= D - a;
        ++a;
      }
      const int b = a + rem;
      t *= f[a] * f[b];
      diag[a * D + b] = t;
      diag[b * D + a] = t;
    } else if (i < NS + D) {
      const int a = i - NS;
      v.red[L.udiag + (size_t)rb * D + a] = t * (f[a] * f[a]);
    } else if (i < NS + 2 * D) {
      const int a = i - NS - D;
      v.red[L.gt + (size_t)rb * D + a] = t * f[a];
    } else {
      const int a = i - NS - 2 * D;
      v.red[L.gc + (size_t)rb * D + a] = t * f[a];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 6: the per-block tail of the normal equations and the start of PCG in ONE launch.  Until then four launches --
// camera_diag_direct_reduce, finish_diag, precond_invert, pcg_init -- of ~6-10 us each plus the ~5 us of idle device
// between two dependent launches, for 1 778 blocks of 81 doubles (profiles/r06_a_gaps.md).  One wavefront per reduced
// block, everything block-local stays in LDS / registers:
//   (DIRECT: camera_diag_direct ran, one rank) the block's chunk partials summed in chunk order -> raw diagonal block,
//       U diagonal, g~, g_c in `red`, exactly as camera_diag_direct_reduce_kernel leaves them;
//   diagonal block of S = raw + clamp(U_aa) / radius -> Sdiag (finish_diag_kernel), max |g_c / scale| for the gradient test;
//   SCHUR_JACOBI inverse of the block -> Minv (precond_invert_kernel; mode 0 merged block, 1 identity, 2 parameter blocks);
//   start of PCG (pcg_init_kernel): x = 0, r = b = g~, z = M^-1 r, p = z (+ the scaled copy the product gathers),
//       rho = r . z.
// rho and max |g_c| are finished by the last workgroup (one ticket, fence-free hand-over of kernels.h).
// Not used with the cluster preconditioners (their factorisation sits between precond and pcg_init) nor by the exact
// solvers (no preconditioner): those keep the separate launches.
// ------------------------------------------------------------------------------------------------------------------
template <int D, bool DIRECT>
__global__ __launch_bounds__(64) void camera_finish_kernel(DeviceView v, RedLayout L, Plan pl, double inv_radius, double lm_lo,
                                                           double lm_hi, int want_gmax, int mode) {
  constexpr int NS = sym_size(D);
  constexpr int NA = n_acc(D);
  __shared__ double Sraw[D * D];
  __shared__ double M[D * D];
  __shared__ double ud[D], gt[D], gc[D], f[D];
  __shared__ int bad;
  __shared__ int cf_last;
  const int rb = blockIdx.x;
  const int t = threadIdx.x;
  const int nb = (int)gridDim.x;
  const signed char* cols = v.rb_cols + (size_t)rb * D;
  if (t == 0) bad = 0;
  if (DIRECT) {
    const int c0 = pl.rb_chunk[rb], c1 = pl.rb_chunk[rb + 1];
    if (t < D) {
      const int c = cols[t];
      f[t] = (c >= 3 && c < 6) ? 1.0 : (c < 0 ? 0.0 : v.scale_c[(size_t)rb * D + t]);
    }
    __syncthreads();
    double* diag = v.red + L.diag + (size_t)rb * D * D;
    for (int i = t; i < NA; i += 64) {
      double s = 0.0;
      for (int c = c0; c < c1; ++c) s += pl.part[(size_t)c * NA + i];
      if (i < NS) {
        int a = 0, rem = i;
        while (rem >= D - a) {
          rem -= D - a;
          ++a;
        }
        const int b = a + rem;
        s *= f[a] * f[b];
        diag[a * D + b] = s;
        diag[b * D + a] = s;
        Sraw[a * D + b] = s;
        Sraw[b * D + a] = s;
      } else if (i < NS + D) {
        const int a = i - NS;
        s *= f[a] * f[a];
        v.red[L.udiag + (size_t)rb * D + a] = s;
        ud[a] = s;
      } else if (i < NS + 2 * D) {
        const int a = i - NS - D;
        s *= f[a];
        v.red[L.gt + (size_t)rb * D + a] = s;
        gt[a] = s;
      } else {
        const int a = i - NS - 2 * D;
        s *= f[a];
        v.red[L.gc + (size_t)rb * D + a] = s;
        gc[a] = s;
      }
    }
  } else {
    for (int e = t; e < D * D; e += 64) Sraw[e] = v.red[L.diag + (size_t)rb * D * D + e];
    if (t < D) {
      ud[t] = v.red[L.udiag + (size_t)rb * D + t];
      gt[t] = v.red[L.gt + (size_t)rb * D + t];
      gc[t] = v.red[L.gc + (size_t)rb * D + t];
    }
  }
  __syncthreads();
  // ---- finish_diag: the block of S, the preconditioner's input (mode 2: cross terms of a view dropped) ----
  for (int e = t; e < D * D; e += 64) {
    const int a = e / D, b = e - a * D;
    double val = Sraw[e];
    if (a == b) {
      if (cols[a] < 0)
        val = 1.0;
      else
        val += fmin(fmax(ud[a], lm_lo), lm_hi) * inv_radius;
    }
    v.Sdiag[(size_t)rb * D * D + e] = val;
    if (mode == 2) {
      const int ci = cols[a], cj = cols[b];
      if (ci >= 0 && cj >= 0 && ((ci < 6) != (cj < 6))) val = 0.0;
    }
    M[e] = val;
  }
  double gm = 0.0;
  if (want_gmax && t < D && cols[t] >= 0) gm = fabs(gc[t] / v.scale_c[(size_t)rb * D + t]);
  gm = wave_max(gm);
  __syncthreads();
  // ---- precond_invert: M <- (L L^T)^-1 of the block (identity for mode 1) ----
  double* out = v.Minv + (size_t)rb * D * D;
  if (mode == 1) {
    for (int e = t; e < D * D; e += 64) {
      const double one = (e / D == e % D) ? 1.0 : 0.0;
      out[e] = one;
      M[e] = one;
    }
    __syncthreads();
  } else {
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
      for (int e = t; e < D * D; e += 64) {
        const int i = e / D, m = e - i * D;
        if (m > j && m <= i) M[e] -= M[i * D + j] * M[m * D + j];
      }
      __syncthreads();
    }
    double y[D];
    if (t < D) {
      // column t of (L L^T)^-1
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
    __syncthreads();  // (every lane is done with the factor)
    if (t < D) {
#pragma unroll
      for (int i = 0; i < D; ++i) M[i * D + t] = y[i];
    }
    __syncthreads();
    if (t == 0 && bad) v.flags[FL_SINGULAR_BLOCK] = 1;
  }
  // ---- pcg_init: x = 0, r = b, z = M^-1 b, p = z ----
  double acc = 0.0;
  double zc = 0.0;
  bool cpx = false;
  if constexpr (D == 9) cpx = v.compact != 0;
  if (t < D) {
    const size_t i = (size_t)rb * D + t;
    const double rn = gt[t];
    double z = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) z += M[t * D + c] * gt[c];
    v.yc[i] = 0.0;
    v.cg_r[i] = rn;
    v.cg_z[i] = z;
    v.cg_p[i] = z;
    if (v.drop_pos && !cpx) v.xs[i] = t < 3 ? z * v.scale_c[i] : z;
    acc = rn * z;
    zc = z;
  }
  if constexpr (D == 9) {
    if (cpx) compact_forward_wave(v, rb, zc, t, v.xs);  // compact planes: the transformed block of p (kernels.h)
  }
  acc = wave_sum(acc);
  if (t == 0) {
    st_agent(&v.dotbuf[rb], acc);
    st_agent(&v.partial[rb], gm);
    cf_last = take_ticket(v.ticket + 1 * kTicketStride, nb) ? 1 : 0;
  }
  __syncthreads();
  if (!cf_last) return;
  // rho in pcg_init_kernel's summation order (workgroups of kPcgStepThreads / 64 blocks added in block order, then that
  // kernel's 1024-leaf tree), so that this launch and the separate ones give the same bits
  __shared__ double tree[kPcgStepThreads];
  constexpr int W16 = kPcgStepThreads / 64;
  const int nb16 = (nb + W16 - 1) / W16;
  for (int i = t; i < kPcgStepThreads; i += 64) {
    double l0 = 0.0;
    for (int g = i; g < nb16; g += kPcgStepThreads) {
      double tt = 0.0;
#pragma unroll
      for (int k = 0; k < W16; ++k) {
        const int rbk = g * W16 + k;
        tt += rbk < nb ? ld_agent(&v.dotbuf[rbk]) : 0.0;
      }
      l0 += tt;
    }
    tree[i] = l0;
  }
  double g0 = 0.0;
  for (int i = t; i < nb; i += 64) g0 = fmax(g0, ld_agent(&v.partial[i]));
  g0 = wave_max(g0);
  __syncthreads();
  for (int o = kPcgStepThreads / 2; o > 0; o >>= 1) {
    for (int i = t; i < o; i += 64) tree[i] += tree[i + o];
    __syncthreads();
  }
  const double r0 = tree[0];
  if (t == 0) {
    v.scal[SC_LAST_RHO] = 1.0;
    v.scal[SC_RHO] = r0;
    v.scal[SC_Q0] = 0.0;
    v.flags[FL_PCG_FAIL] = (r0 == 0.0 || !isfinite(r0)) ? 1 : 0;
    *v.pcg_done = 0;
    if (want_gmax) v.scal[SC_GMAX] = g0;
  }
}

}  // namespace ddg
}  // namespace tmi
