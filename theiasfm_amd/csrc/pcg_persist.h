// A whole PCG solve on the FORMED reduced camera matrix as ONE persistent launch (kernel class 5 + 6, schur_mode
// explicit / auto).  Replaces, behind ceres::Solve at src/theia/sfm/bundle_adjustment/bundle_adjuster.cc:205
// (ITERATIVE_SCHUR), the launch-per-step loop of engine.hip::solve_reduced_pcg -- spmv_rows, spmv_cols, pcg_step, pcg_p
// and a poll of the host mirror per PCG iteration.
//
// Why (round 5, profiles/r05_a_street_probe.jsonl): on a problem with SEQUENCE structure (scene "street": S a band of
// 14 % fill = 145 MB, 100-440 PCG iterations per LM iteration -- what real collections look like) a PCG iteration took
// 102 us of which the kernels were 57 (product) + 17 (vector) and the rest dispatch, drain and the host round trip:
// the LM iteration was 12.4 ms, 10 of them this loop.  The ring scene of the headline hides that (4.8 PCG iterations of
// 250 us each).  Here the grid (one workgroup per CU, co-resident) stays on the device for the whole solve and the
// phases of an iteration are separated by grid barriers instead of kernel boundaries:
//
//   A  rows pass of q = S p over this wavefront's chunks of S (the symmetric block storage is read once; kernels.h,
//      spmv_rows): p is not stored -- p_k = z_{k-1} + beta_k p_{k-1} is formed from the two published vectors where it
//      is gathered, which saves the barrier a stored p would need
//   -- barrier --
//   B  cols pass for the view blocks this wavefront owns: q_j = Sdiag_j p_j + row partials + transposed partials
//      (fixed orders), and the workgroup's share of p.q
//   -- barrier --
//   C  alpha; x += alpha p, r -= alpha q, z = M^-1 r for the owned blocks (z and p_k are published), shares of
//      Q1 = -x.(b + r) and rho' = r.z          [every 10th iteration: r = b - S x through two more passes, Ceres'
//      residual_reset_period]
//   -- barrier --
//   D  every workgroup finishes the sums in the same fixed order, so all of them take the same decision: zeta =
//      k (Q1 - Q0) / Q1 < eta stops (ConjugateGradientsSolver's Q tolerance); workgroup 0 publishes the scalars to
//      the host mirror, the host reads ONE flag per solve.
//
// Data that crosses workgroups inside the launch (tbuf, rbuf, z, p, x at a reset, the partial sums) is written with
// write-through stores and read with agent-scope loads -- the hand-over idiom of kernels.h / dense_cholesky_df.h
// (hardware assumption spelled out there; gfx950 only) -- because the XCDs' L2s are not coherent with each other
// within a kernel.  Everything a view block's owner alone touches (x, r, q) is plain memory.  No atomics in the
// arithmetic, fixed summation orders: bit-reproducible.  Every wait is bounded: a grid that cannot become co-resident
// aborts with FL_CHOL_ABORT and the caller falls back to the launch-per-step loop.
#pragma once
#include <hip/hip_runtime.h>

#include "dense_cholesky_df.h"
#include "device_view.h"
#include "kernels.h"

namespace tmi {
namespace ppcg {

constexpr int kWaves = 8;
constexpr int kThreads = 64 * kWaves;
constexpr int kBarInts = 2 * 8 * 1024;  // the grid barrier's records: two sets of 32 bytes per workgroup (grid <= 1024)
constexpr int kResetPeriod = 10;  // ConjugateGradientsSolver: residual_reset_period
constexpr int kDepth = 2;         // trips of S a wavefront keeps in flight (same box, street scene, us per PCG iteration:
                                  // 1: 108.4, 2: 88.2, 3: 94.4, 5: 106.6 -- the loads of a trip live in registers)
// chunks per wavefront whose header and block columns stay on chip over the solve; entries per lane of the first owned
// column whose tbuf indices do (both in LDS: the wide blocks leave less of it)
__host__ __device__ constexpr int chunks_cached(int D) { return D <= 9 ? 3 : 1; }
__host__ __device__ constexpr int col_cached(int D) { return D <= 9 ? 20 : 8; }

struct Args {
  const double* ub;   // upper blocks of S
  const double* b;    // right-hand side
  double* pbuf;       // [2][n] p_k lives in pbuf[k & 1]
  double* xpub;       // [n] x, published at a residual reset
  double* partial;    // [3][grid] workgroup shares of p.q, Q1, rho'
  int* bar;           // grid barrier: one Rec per workgroup (zeroed per launch)
  int* ctrl;          // abort flag (DeviceView::flags + FL_CHOL_ABORT)
  double eta;
  int min_it, max_it;
  const double* red8;
  HostMirror* mirror;
  unsigned long long seq;
  long long* prof;    // optional (TMI_BA_PPCG_PROF): [3][16] ticks of the 100 MHz clock per phase: first, middle, last workgroup
};

// slot of DeviceView::scal the launch leaves its iteration count in
constexpr int SC_PCG_IT = 23;

// Grid barrier and grid-wide sum in one step.  Every workgroup leaves a RECORD {v0, v1, guard} with three plain
// write-through stores, guard = bits(v0) ^ bits(v1) ^ epoch * K: a reader that finds the guard matching the values it read
// knows all three words are this epoch's, in whatever order the stores arrived -- so nothing has to be acknowledged between
// "the sums are written" and "the flag is written" (a store round trip each in a counter-and-flag barrier: 6-7 us per
// barrier measured with 256 workgroups), and the sums need no load of their own afterwards.  Wavefront 0 of EVERY
// workgroup polls all the records (grid / 64 per lane) and adds them up in the same fixed order (lane l takes records
// l, l + 64, ...; then the butterfly of wave_sum), so all workgroups get the same bits.  The bulk write-through stores a
// thread made before the call (tbuf, rbuf, z, p) are acknowledged before its workgroup's record is written, i.e. they
// are visible to agent-scope loads after the call.  false: aborted (workgroup-uniform).
struct Rec {
  double v0, v1;
  unsigned long long guard, pad;
};
constexpr unsigned long long kGuardMul = 0x9E3779B97F4A7C15ull;

__device__ __forceinline__ void st_agent_u64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_agent_u64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// s0, s1: this THREAD's shares (summed over the wavefront, then over the wavefronts in order); wsum: [2][kWaves] LDS
__device__ __forceinline__ bool grid_reduce(const Args& a, int& epoch, double s0, double s1, double (*wsum)[kWaves],
                                            double* sh_val, int* sh_ok, double* t0, double* t1) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (lane == 0) {
    wsum[0][w] = s0;
    wsum[1][w] = s1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's bulk stores have been acknowledged
  __syncthreads();
  ++epoch;
  if (w == 0) {
    const int G = (int)gridDim.x;
    // two sets of records, by the parity of the epoch: a workgroup that is already through this barrier writes its NEXT
    // record into the other set, so a slower workgroup still polling this epoch never finds a record replaced
    Rec* rec = reinterpret_cast<Rec*>(a.bar) + (size_t)(epoch & 1) * G;
    const unsigned long long tag = (unsigned long long)epoch * kGuardMul;
    if (lane == 0) {
      double m0 = 0.0, m1 = 0.0;
#pragma unroll
      for (int k = 0; k < kWaves; ++k) {
        m0 += wsum[0][k];
        m1 += wsum[1][k];
      }
      Rec* mine = rec + blockIdx.x;
      st_agent(&mine->v0, m0);
      st_agent(&mine->v1, m1);
      st_agent_u64(&mine->guard, (unsigned long long)__double_as_longlong(m0) ^ (unsigned long long)__double_as_longlong(m1) ^ tag);
    }
    bool ok = true;
    double a0 = 0.0, a1 = 0.0;
    const long long tstart = wall_clock64();
    for (int it = 0;; ++it) {
      bool all = true;
      a0 = a1 = 0.0;
      for (int i = lane; i < G; i += 64) {
        const double r0 = ld_agent(&rec[i].v0), r1 = ld_agent(&rec[i].v1);
        const unsigned long long gd = ld_agent_u64(&rec[i].guard);
        all = all && (((unsigned long long)__double_as_longlong(r0) ^ (unsigned long long)__double_as_longlong(r1) ^ gd) == tag);
        a0 += r0;
        a1 += r1;
      }
      if (__ballot(all) == ~0ull) break;
      __builtin_amdgcn_s_sleep(1);
      if ((it & 255) == 255) {
        if (cdf::ld_flag(a.ctrl) != 0 || wall_clock64() - tstart > cdf::kSpinLimitTicks) {
          if (lane == 0) cdf::st_flag(a.ctrl, 1);
          ok = false;
          break;
        }
      }
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if (lane == 0) {
      sh_val[0] = a0;
      sh_val[1] = a1;
      *sh_ok = ok ? 1 : 0;
    }
  }
  __syncthreads();
  if (t0) *t0 = sh_val[0];
  if (t1) *t1 = sh_val[1];
  const bool ok = *sh_ok != 0;
  __syncthreads();  // (sh_val / wsum are free again)
  return ok;
}

template <int D>
__global__ __launch_bounds__(kThreads) void pcg_persistent_kernel(DeviceView v, Args a) {
  constexpr int G = 64 / D;
  constexpr int BLK = D * D;
  constexpr int NW = G * BLK;
  constexpr int NLD = (NW + 127) / 128;
  constexpr int PITCH = (NW + 1) & ~1;
  constexpr int T = kSpmvTrips;
  constexpr int kChunksCached = chunks_cached(D), kColCached = col_cached(D);
  __shared__ __attribute__((aligned(16))) double sblk[kWaves][PITCH];
  __shared__ double wsum[2][kWaves];
  __shared__ double own_rows[kWaves][2][D][D + 1];  // Sdiag and M^-1 of the register-resident view block of a wavefront
  __shared__ int own_cols[kWaves][kColCached][64];  // tbuf indices of its column (entry m of lane l), -1: none
  // The chunks [ch_lo, ch_hi) of S belong to this workgroup, dealt round its wavefronts: every workgroup streams the same
  // share of S (dealt by index over the whole grid, the first workgroups held three chunks per wavefront where the last
  // held two: 45 against 27 us per pass).  Header and block columns of the first kLocalCached chunks stay in LDS.
  constexpr int kLocalCached = kWaves * kChunksCached;
  __shared__ int own_ubj[kLocalCached][T][64];  // block columns (trip t, lane l)
  __shared__ int own_hdr[kLocalCached][4];      // {block row, first block, end}
  __shared__ double sh_val[2];
  __shared__ int sh_ok;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nwg = (int)gridDim.x;
  const int n_waves = nwg * kWaves;
  // (equal shares of the WORK, not of the chunks: a chunk costs about three trips' worth of latency plus its trips --
  // the last chunk of a block row is partly filled and the rows of the upper triangle get shorter towards the end; by
  // chunk count the first workgroup took twice as long as the last, by block count the last 30 % longer than the first)
  auto first_chunk_at = [&](long long work) {  // first chunk c with 3 c + (blocks before c) / G >= work
    int lo = 0, hi = v.n_spc;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (3ll * mid + v.spc_u0[mid] / G < work) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  };
  const long long work_total = 3ll * v.n_spc + v.nub / G;
  const int ch_lo = blockIdx.x == 0 ? 0 : first_chunk_at(work_total * blockIdx.x / nwg);
  const int ch_hi = (int)blockIdx.x == nwg - 1 ? v.n_spc : first_chunk_at(work_total * (blockIdx.x + 1) / nwg);
  const int own0 = w * nwg + (int)blockIdx.x;    // view blocks: own0, own0 + n_waves, ... (consecutive blocks sit in
                                                 // different workgroups: the column lengths of the upper triangle grow
                                                 // with the block index)
  const int n = v.Nrb * D;
  const int g = lane / D, c = lane - g * D;
  const int ld_lane = lane < D ? lane : 0;
  double* __restrict__ tb = v.tbuf;
  double* pcur = nullptr;
  const double* pprev = nullptr;
  int epoch = 0;
  double rho = v.scal[SC_RHO], Q0 = 0.0, beta = 0.0;
  double pq = 0.0, alpha = 0.0, zeta = -1.0, rho_bad = 0.0;
  bool fail = v.flags[FL_PCG_FAIL] != 0, aborted = false;
  int it = 0;

  // ---- what never changes over the solve, fetched once: the headers and block columns of this wavefront's first
  // chunks, the tbuf indices of its first view block's column, that block's rows of Sdiag and M^-1 and its state
#pragma unroll 1
  for (int q = w; q < kLocalCached; q += kWaves) {
    const int cidx = ch_lo + q;
#pragma unroll
    for (int t = 0; t < T; ++t) own_ubj[q][t][lane] = 0;
    if (cidx < ch_hi) {
      const int row = v.spc_row[cidx], u0 = v.spc_u0[cidx];
      const int u1 = min(u0 + T * G, v.urow_ptr[row + 1]);
      if (lane == 0) {
        own_hdr[q][0] = row;
        own_hdr[q][1] = u0;
        own_hdr[q][2] = u1;
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int u = u0 + t * G + g;
        if (g < G && u < u1) own_ubj[q][t][lane] = v.ub_j[u];
      }
    }
  }
  __syncthreads();
  const bool own = own0 < v.Nrb;
  const int oi = own ? own0 * D + ld_lane : 0;  // this lane's entry of the owned block (lanes < D)
  int col_k0 = 0, col_k1 = 0, col_c0 = 0, col_c1 = 0;
  double x_own = 0.0, r_own = 0.0, z_own = 0.0, p_own = 0.0, b_own = 0.0;
#pragma unroll
  for (int m = 0; m < kColCached; ++m) own_cols[w][m][lane] = -1;
  if (own) {
    col_k0 = v.ucol_ptr[own0];
    col_k1 = v.ucol_ptr[own0 + 1];
    col_c0 = v.spc_rptr[own0];
    col_c1 = v.spc_rptr[own0 + 1];
#pragma unroll
    for (int m = 0; m < kColCached; ++m) {
      const int k = col_k0 + g + G * m;
      if (g < G && k < col_k1) own_cols[w][m][lane] = v.ucol_u[k];
    }
    if (lane < D) {
#pragma unroll
      for (int cc = 0; cc < D; ++cc) {
        own_rows[w][0][lane][cc] = v.Sdiag[(size_t)own0 * BLK + lane * D + cc];
        own_rows[w][1][lane][cc] = v.Minv[(size_t)own0 * BLK + lane * D + cc];
      }
      x_own = v.yc[oi];
      r_own = v.cg_r[oi];
      z_own = v.cg_z[oi];
      b_own = a.b[oi];
    }
  }

  // the vector of a product: mode 0: p_k = z + beta pprev (k = 1: z), mode 1: the published x
  auto vec = [&](int mode, int i) -> double {
    if (mode) return ld_agent(a.xpub + i);
    const double z = ld_agent(v.cg_z + i);
    return pprev ? z + beta * ld_agent(pprev + i) : z;
  };
  // ---- rows pass, one chunk (kernels.h, spmv_rows_kernel): every gathered entry of the vector is requested up front
  // (one round trip for the whole chunk), kDepth trips of S are in flight; tbuf and rbuf are read by other workgroups
  auto do_chunk = [&](int mode, int cidx, int row, int u0, int u1, const int (&ubj)[T]) {
    double xi[D], acc[D], xj[T];
#pragma unroll
    for (int r = 0; r < D; ++r) {
      xi[r] = cdf::bcast_lane(vec(mode, row * D + r), 0);  // the same value in every lane: kept in scalar registers
      acc[r] = 0.0;
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int u = u0 + t * G + g;
      xj[t] = (g < G && u < u1) ? vec(mode, ubj[t] * D + c) : 0.0;
    }
    double2_a8 ld[kDepth][NLD];
    auto fetch = [&](int t, double2_a8 (&dst)[NLD]) {
      const int ub0 = u0 + t * G;
      const int nv = max(0, min(G, u1 - ub0)) * BLK;
      const double* src = a.ub + (size_t)ub0 * BLK;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int e = 2 * (64 * i + lane);
        double2_a8 t2 = {0.0, 0.0};
        if (e + 1 < nv) {
          t2 = *reinterpret_cast<const double2_a8*>(src + e);
        } else if (e < nv) {
          t2.x = src[e];
        }
        dst[i] = t2;
      }
    };
#pragma unroll
    for (int t = 0; t < kDepth; ++t) fetch(t, ld[t]);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int ub0 = u0 + t * G;
      if (ub0 < u1) {  // wave uniform
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          const int e = 2 * (64 * i + lane);
          if (e < PITCH) *reinterpret_cast<double2_a8*>(&sblk[w][e]) = ld[t % kDepth][i];
        }
        __builtin_amdgcn_wave_barrier();
        if (t + kDepth < T) fetch(t + kDepth, ld[t % kDepth]);
        const int u = ub0 + g;
        if (g < G && u < u1) {
          const double* blk = &sblk[w][g * BLK];
          double tt = 0.0;
#pragma unroll
          for (int r = 0; r < D; ++r) {
            const double e = blk[r * D + c];
            tt += e * xi[r];
            acc[r] += e * xj[t];
          }
          st_agent(tb + (size_t)u * D + c, tt);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
#pragma unroll
    for (int r = 0; r < D; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < D; ++r) st_agent(v.rbuf + (size_t)cidx * D + r, acc[r]);
    }
  };
  auto rows_pass = [&](int mode) {
#pragma unroll 1
    for (int q = w; ch_lo + q < ch_hi; q += kWaves) {
      const int cidx = ch_lo + q;
      int row, u0, u1, ubj[T];
      if (q < kLocalCached) {
        row = own_hdr[q][0];
        u0 = own_hdr[q][1];
        u1 = own_hdr[q][2];
#pragma unroll
        for (int t = 0; t < T; ++t) ubj[t] = own_ubj[q][t][lane];
      } else {
        row = v.spc_row[cidx];
        u0 = v.spc_u0[cidx];
        u1 = min(u0 + T * G, v.urow_ptr[row + 1]);
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const int u = u0 + t * G + g;
          ubj[t] = (g < G && u < u1) ? v.ub_j[u] : 0;
        }
      }
      do_chunk(mode, cidx, row, u0, u1, ubj);
    }
  };
  // ---- cols pass for a view block by one wavefront: lanes r < D return (S x)_j[r]; xj_lane = the block's own entries
  // of the vector (lanes < D).  The transposed partials of the column: kColCached entries per lane with their indices
  // given (one round trip), the rest in batches of eight (index, entry); the row's chunk partials four at a time.
  auto cols_block = [&](int k0, int k1, int c0, int c1, bool have_cu, const double* sd, double xj_lane) -> double {
    // every load of the block is requested before the first one is waited for: the row's first chunk partials ...
    double rv[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) rv[m] = (lane < D && c0 + m < c1) ? ld_agent(v.rbuf + (size_t)(c0 + m) * D + lane) : 0.0;
    // ... and the column's transposed partials
    double acc = 0.0;
    if (g < G) {
      int k = k0 + g;
      if (have_cu) {
        double tv[kColCached];
#pragma unroll
        for (int m = 0; m < kColCached; ++m) {
          const int cu = own_cols[w][m][lane];
          tv[m] = cu >= 0 ? ld_agent(tb + (size_t)cu * D + c) : 0.0;
        }
#pragma unroll
        for (int m = 0; m < kColCached; ++m) acc += tv[m];
        k += G * kColCached;
      }
      for (; k < k1; k += 8 * G) {
        int q[8];
        double tv[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) q[m] = (k + m * G < k1) ? v.ucol_u[k + m * G] : -1;
#pragma unroll
        for (int m = 0; m < 8; ++m) tv[m] = q[m] >= 0 ? ld_agent(tb + (size_t)q[m] * D + c) : 0.0;
#pragma unroll
        for (int m = 0; m < 8; ++m) acc += tv[m];
      }
    }
    double s = 0.0;
#pragma unroll
    for (int cc = 0; cc < D; ++cc) {
      const double xc = __shfl(xj_lane, cc, 64);
      s += sd[cc] * xc;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) s += rv[m];
    if (lane < D)
      for (int ch = c0 + 4; ch < c1; ++ch) s += ld_agent(v.rbuf + (size_t)ch * D + lane);
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
      const double t = __shfl(acc, gg * D + ld_lane, 64);
      s += t;
    }
    return s;  // (meaningful in lanes < D)
  };
  // z = M^-1 r of a block (r in lanes < D; mrow = this lane's row of the block's inverse)
  auto precond = [&](const double* mrow, double rn) -> double {
    double z = 0.0;
#pragma unroll
    for (int cc = 0; cc < D; ++cc) {
      const double rc = __shfl(rn, cc, 64);
      z += mrow[cc] * rc;
    }
    return z;
  };
  const double* sd_row = &own_rows[w][0][ld_lane][0];
  const double* mi_row = &own_rows[w][1][ld_lane][0];
  long long tprev = a.prof ? wall_clock64() : 0;
  const int prof_slot = blockIdx.x == 0 ? 0 : ((int)blockIdx.x == nwg / 2 ? 1 : ((int)blockIdx.x == nwg - 1 ? 2 : -1));
  auto lap = [&](int k) {
    if (a.prof && prof_slot >= 0 && threadIdx.x == 0) {
      const long long t = wall_clock64();
      a.prof[16 * prof_slot + k] += t - tprev;
      tprev = t;
    }
  };
  while (!fail) {
    ++it;
    pcur = a.pbuf + (size_t)(it & 1) * n;
    const bool reset = (it % kResetPeriod) == 0;
    // ---- A
    rows_pass(0);
    lap(0);
    if (!grid_reduce(a, epoch, 0.0, 0.0, wsum, sh_val, &sh_ok, nullptr, nullptr)) { aborted = true; break; }
    lap(1);
    // ---- B: q of the owned blocks, p.q.  The first owned block lives in registers (p_own = this iteration's p)
    double my_pq = 0.0, q_own = 0.0;
#pragma unroll 1
    for (int j = own0; j < v.Nrb; j += n_waves) {
      const bool first = j == own0;
      const int i = j * D + ld_lane;
      double pj;
      if (first) {
        p_own = pprev ? z_own + beta * p_own : z_own;
        pj = lane < D ? p_own : 0.0;
      } else {
        pj = lane < D ? vec(0, i) : 0.0;
      }
      // (a further block's rows of Sdiag / M^-1 come straight from memory)
      const double* sd_tmp = v.Sdiag + (size_t)j * BLK + ld_lane * D;
      const double qj = first ? cols_block(col_k0, col_k1, col_c0, col_c1, true, sd_row, pj)
                              : cols_block(v.ucol_ptr[j], v.ucol_ptr[j + 1], v.spc_rptr[j], v.spc_rptr[j + 1], false, sd_tmp, pj);
      if (first) q_own = qj;
      else if (lane < D) v.cg_q[i] = qj;  // owner only
      if (lane < D) my_pq += pj * qj;
    }
    lap(2);
    if (!grid_reduce(a, epoch, my_pq, 0.0, wsum, sh_val, &sh_ok, &pq, nullptr)) { aborted = true; break; }
    lap(3);
    // ---- C
    const bool ok = pq > 0.0 && isfinite(pq);
    alpha = rho / pq;
    if (!ok) break;  // LINEAR_SOLVER_NO_CONVERGENCE: nothing moves (every workgroup takes this branch)
    if (!isfinite(alpha)) {
      fail = true;
      break;
    }
    double my_q1 = 0.0, my_rho = 0.0;
#pragma unroll 1
    for (int j = own0; j < v.Nrb; j += n_waves) {
      const bool first = j == own0;
      const int i = j * D + ld_lane;
      double rn = 0.0;
      if (first) {
        if (lane < D) {
          x_own += alpha * p_own;
          st_agent(pcur + i, p_own);
          if (reset) {
            st_agent(a.xpub + i, x_own);
          } else {
            r_own -= alpha * q_own;
            my_q1 += -x_own * (b_own + r_own);
            rn = r_own;
          }
        }
      } else if (lane < D) {
        const double pj = vec(0, i);
        const double x = v.yc[i] + alpha * pj;
        v.yc[i] = x;
        st_agent(pcur + i, pj);
        if (reset) {
          st_agent(a.xpub + i, x);
        } else {
          rn = v.cg_r[i] - alpha * v.cg_q[i];
          v.cg_r[i] = rn;
          my_q1 += -x * (a.b[i] + rn);
        }
      }
      if (!reset) {
        const double z = precond(first ? mi_row : v.Minv + (size_t)j * BLK + ld_lane * D, rn);
        if (first) z_own = z;
        if (lane < D) {
          st_agent(v.cg_z + i, z);
          my_rho += rn * z;
        }
      }
    }
    if (reset) {
      // r = b - S x: two more passes over S with x as the vector
      if (!grid_reduce(a, epoch, 0.0, 0.0, wsum, sh_val, &sh_ok, nullptr, nullptr)) { aborted = true; break; }
      rows_pass(1);
      if (!grid_reduce(a, epoch, 0.0, 0.0, wsum, sh_val, &sh_ok, nullptr, nullptr)) { aborted = true; break; }
#pragma unroll 1
      for (int j = own0; j < v.Nrb; j += n_waves) {
        const bool first = j == own0;
        const int i = j * D + ld_lane;
        const double* sd_tmp = v.Sdiag + (size_t)j * BLK + ld_lane * D;
        const double xj = lane < D ? (first ? x_own : v.yc[i]) : 0.0;
        const double bj = lane < D ? (first ? b_own : a.b[i]) : 0.0;
        const double t = first ? cols_block(col_k0, col_k1, col_c0, col_c1, true, sd_row, xj)
                               : cols_block(v.ucol_ptr[j], v.ucol_ptr[j + 1], v.spc_rptr[j], v.spc_rptr[j + 1], false, sd_tmp, xj);
        const double rn = lane < D ? bj - t : 0.0;
        if (lane < D) my_q1 += -xj * (bj + rn);
        if (first) r_own = rn;
        else if (lane < D) v.cg_r[i] = rn;
        const double z = precond(first ? mi_row : v.Minv + (size_t)j * BLK + ld_lane * D, rn);
        if (first) z_own = z;
        if (lane < D) {
          st_agent(v.cg_z + i, z);
          my_rho += rn * z;
        }
      }
    }
    lap(5);
    // ---- D
    double Q1, rho_new;
    if (!grid_reduce(a, epoch, my_q1, my_rho, wsum, sh_val, &sh_ok, &Q1, &rho_new)) { aborted = true; break; }
    lap(6);
    zeta = it * (Q1 - Q0) / Q1;
    Q0 = Q1;
    rho_bad = (rho_new == 0.0 || !isfinite(rho_new) || !isfinite(rho_new / rho)) ? 1.0 : 0.0;
    beta = rho_new / rho;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      v.scal[SC_Q1] = Q1;
      v.scal[SC_Q0] = Q1;
      v.scal[SC_LAST_RHO] = rho;
      v.scal[SC_RHO] = rho_new;
    }
    rho = rho_new;
    pprev = pcur;
    if ((zeta < a.eta && it >= a.min_it) || it >= a.max_it || rho_bad != 0.0) break;
  }
  // the register-resident block's solution goes back to memory (the other blocks' x never left it)
  if (own && lane < D) v.yc[oi] = x_own;
  // ---- the scalars the host decides on, published as pcg_step_kernel publishes them
  if (blockIdx.x != 0) return;
  __shared__ double pub[SC_COUNT];
  __shared__ int pubf[FL_COUNT];
  __syncthreads();
  if (threadIdx.x == 0) {
    v.scal[SC_PQ] = pq;
    v.scal[SC_ALPHA] = alpha;
    v.scal[SC_ZETA] = (pq > 0.0 && isfinite(pq)) ? zeta : -1.0;
    v.scal[SC_RHO_BAD] = rho_bad;
    v.scal[SC_PCG_IT] = (double)it;
    if (fail) v.flags[FL_PCG_FAIL] = 1;
    (void)aborted;  // the abort flag itself is DeviceView::flags[FL_CHOL_ABORT], set by whoever timed out
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < SC_COUNT) pub[t] = v.scal[t];
  if (t < FL_COUNT) pubf[t] = t == FL_CHOL_ABORT ? cdf::ld_flag(v.flags + t) : v.flags[t];
  __syncthreads();
  if (t < SC_COUNT) a.mirror->scal[t] = pub[t];
  if (t < 8) a.mirror->red[t] = a.red8[t];
  if (t < FL_COUNT) a.mirror->flags[t] = pubf[t];
  __threadfence_system();
  __syncthreads();
  if (t == 0) __hip_atomic_store(&a.mirror->seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace ppcg
}  // namespace tmi
