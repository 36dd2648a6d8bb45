// ieskf_k1.hip — the LIST kernel of the split IESKF path (ieskf_split.h): iterations `split_iters` .. end of
// performIESKF (SE:465-600) without a grid.
//
// One 256-thread workgroup per scan pair, one lane per query, queries packed densely (plane queries
// first, then line queries: every wave is full), ~7 KB of LDS: four scans are resident per CU and fill
// each other's barriers and serial tails.  Per iteration and query:
//   de-skew (SE:1066-1080)  ->  nearest neighbour among the listed candidates (exact (distance, index)
//   order of the kd-tree restatement)  ->  second / third point among the listed candidates the
//   reference's index walk would visit (SE:859-910, 983-1024: forward part bounded by the QUERY count,
//   ring classes, first-seen-wins as a (distance, visit rank) key)  ->  certificate: every unlisted
//   point is farther than the chosen one (ieskf_split.h)  ->  plane / line row (SE:917-951, 1031-1061)
//   ->  28 sums (wave butterflies)  ->  6x6 solve, boxPlus, stop tests (SE:542-580).
// A decision that cannot be certified is redone by the whole workgroup as an exhaustive search of the
// query's target cloud in global memory — the reference's literal rules, exact by construction.
//
// Results are identical to the persistent kernel's: the same f32 distances decide, the same keys break
// ties, the rows and the update are the same code (ieskf_device.h, ieskf_rowsum.h).
#include <hip/hip_runtime.h>

#include "ieskf_rowsum.h"
#include "ieskf_split.h"

namespace lins {
namespace k1 {

constexpr int kBlock = 256, kWaves = kBlock / 64;
constexpr int kBackRank = 0x40000000;
constexpr int kBruteCap = kBlock;  // (one entry per lane and round)
constexpr int kPool = 1792;      // candidates of the scan kept in LDS (28 KB: four workgroups still fit a CU)
constexpr int kMaxQ = 512;       // queries of a split scan (lins_capi.hip: <= 320 plane + 192 line)
constexpr unsigned short kNotInLds = 0xFFFF;

struct OutRec {  // (layout of the persistent kernels' record, lins_capi.hip reads it)
  double residual_norm, update_norm;
  int iters, converged, diverged, m_surf, m_corner, pad[3];
};

struct Lds {
  double P[324];
  double trig[kSincCosTab];  // series coefficients of the de-skew (as in the grid kernel: the two must de-skew alike)
  IterConst ic;
  double filt[19];
  double sums[28];
  double partial[kWaves * 28];
  double aug[3][42];
  double stage[22];  // the update's results on their way to the other waves: linState_ (19), |r|, |r| kept, |dx|
  int stage_flags[2];  // diverged, converged
  double res_prev, res_last, upd_norm;
  unsigned long long red[kWaves][3];
  int ring_start[2][kRingsBinned + 1];
  int m_surf, m_corner, iter, conv, div;
  int nbrute, brute_total, gcount;
  long long prof[8];  // phase clocks of the PROF variant
  // The scan's candidate lists, copied ONCE from the hand-off buffer (L2) and read by every iteration from here:
  // query q's list is pool[qoff[q] .. + count).  qoff = kNotInLds: that list did not fit, or the list kernel has
  // re-gathered it since — it is read from global memory.
  float4 pool[kPool];
  unsigned short qoff[kMaxQ];
  int scan_tmp[8];
  unsigned short brute_q[kBruteCap];
};
__shared__ Lds g;

// candidate of the index walk around nearest neighbour j1 on ring rho (sorted cloud): forward part
// (j1, f_hi) in ascending order first, then the backward part [b_lo, j1) in descending order
struct Walk {
  int j1, f_hi, b_lo;
};
__device__ __forceinline__ Walk make_walk(const int* ring_start, int n, int nq, int j1, int rho) {
  const int fend = nq < n ? nq : n;
  const int r_hi = rho + 3 < kRingsBinned ? rho + 3 : kRingsBinned;
  const int r_lo = rho - 2 > 0 ? rho - 2 : 0;
  return Walk{j1, fend < ring_start[r_hi] ? fend : ring_start[r_hi], ring_start[r_lo]};
}
__device__ __forceinline__ bool walk_rank(const Walk& w, int j, int& rank) {
  const bool fwd = (unsigned)(j - w.j1 - 1) < (unsigned)(w.f_hi > w.j1 + 1 ? w.f_hi - w.j1 - 1 : 0);
  const bool bwd = (unsigned)(j - w.b_lo) < (unsigned)(w.j1 - w.b_lo);
  rank = fwd ? j - w.j1 : kBackRank + (w.j1 - j);
  return fwd || bwd;
}
__device__ __forceinline__ unsigned long long pack_key(float d, int key) {
  return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)key;
}
// (ieskf_lds_impl.h's certificate test: d_now < (lb - drift)^2 with slack for the f32 roundings)
__device__ __forceinline__ bool certified(float d_now, float lb, float drift) {
  return sqrtf(d_now) * (1.f + 4e-6f) + 2e-6f < (lb - drift * (1.f + 4e-6f)) * (1.f - 4e-6f);
}
// The list kernel rewrites lists and claims of queries it had to search exhaustively; the per-CU vector L1 is
// not coherent with those stores, so everything read from the hand-off buffers bypasses it (L2-served).
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_fresh(const float4* p) {
  const v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long k) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)k, m), hi = __shfl_xor((unsigned)(k >> 32), m);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    k = o < k ? o : k;
  }
  return k;
}

// ---------------------------------------------------------------------------
// serial tail of one iteration: the persistent kernel's solve_and_update (ieskf_lds_impl.h) on this
// kernel's LDS block — (sigma^2 I + A P_SS) w = g + A d_S, dx = d - P[:,S] w, NaN / divergence /
// convergence tests, boxPlus (SE:542-580), then the constants of the next iteration.
// ---------------------------------------------------------------------------
// (scalars by value: a reference to the kernel's parameter struct would keep the whole struct in scratch)
__device__ __noinline__ void solve_and_update(double prm_r2, int prm_fixed_iters, int prm_pad, int tid, int iter) {
  Lds& L = g;
  const int lane = tid & 63, wave = tid >> 6;
  // ---- wave 0: the 6 x 7 system, its solution (spread over the wave), dx, the stop tests and boxPlus; the new
  // linearisation state is STAGED in LDS (the old one may still be read by the other waves until the barrier)
  if (wave == 0) {
    double v = 0.0;
    if (lane < 42) {
      const int i = lane / 7, j = lane % 7;
      if (j < 6) {
        v = (i == j ? prm_r2 : 0.0);
#pragma unroll
        for (int k = 0; k < 6; ++k) v += sym6(L.sums, i, k) * L.P[sidx(k) * 18 + sidx(j)];
      } else {
        v = L.sums[21 + i];
#pragma unroll
        for (int k = 0; k < 6; ++k) v += sym6(L.sums, i, k) * L.ic.d[sidx(k)];
      }
    }
    // (what dx needs from LDS besides the solution is read BEFORE the solve: the reads then wait behind nothing)
    double pls[6] = {0, 0, 0, 0, 0, 0}, dl = 0;
    if (lane < 18) {
#pragma unroll
      for (int k = 0; k < 6; ++k) pls[k] = L.P[lane * 18 + sidx(k)];
      dl = L.ic.d[lane];
    }
    double wsol[6];
    wave_gj_solve6(v, lane, wsol);
    double dxi = 0;
    if (lane < 18) {
      double sacc = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) sacc += pls[k] * wsol[k];
      dxi = dl - sacc;
    }
    double lin[19];
#pragma unroll
    for (int k = 0; k < 19; ++k) lin[k] = L.ic.lin[k];
    double dth[3] = {0, 0, 0};
    bool has_nan = false;
    double un = 0;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      const double vk = readlane_f64(dxi, k);
      has_nan = has_nan || isnan(vk);
      un += vk * vk;
      if (k >= 6 && k < 9)
        dth[k - 6] = vk;
      else
        lin[k < 6 ? k : k + 1] += vk;
    }
    un = sqrt(un);
    const double rn = sqrt(L.sums[27]);
    double res_prev = L.res_prev;
    int div = 0, conv = 0;
    if (has_nan) {
      div = 2, un = L.upd_norm;
    } else if (rn > res_prev * 10) {
      div = 1, un = L.upd_norm;
    } else {
      const Q4 qn = qnormalized(qmul(Q4{lin[6], lin[7], lin[8], lin[9]}, axis2quat_fast(V3{dth[0], dth[1], dth[2]})));
      lin[6] = qn.w, lin[7] = qn.x, lin[8] = qn.y, lin[9] = qn.z;
      if (un <= 1e-2 && !prm_fixed_iters) conv = 1;
      res_prev = rn;
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 19; ++k) L.stage[k] = lin[k];
      L.stage[19] = rn, L.stage[20] = res_prev, L.stage[21] = un;
      L.stage_flags[0] = div, L.stage_flags[1] = conv;
    }
  }
  __syncthreads();  // every reader of the old linearisation state is done; the staged one is visible
  const int div = L.stage_flags[0];
  if (wave < 3 && !div && !(prm_pad & 8192)) {
    // the constants of the next iteration, one wave each: linState_ + R^T | phi, Rinvleft(-phi)^T | x_filter (-) x_lin
    const Q4 q{L.stage[6], L.stage[7], L.stage[8], L.stage[9]};
    if (wave == 0) {
      const M3 Rt = mtrans(qmat(q));
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 19; ++k) L.ic.lin[k] = L.stage[k];
        L.ic.Rt = Rt;
      }
    } else if (wave == 1) {
      V3 phi;
      M3 Gt;
      phi_and_Gt(q, phi, Gt);
      if (lane == 0) L.ic.phi = phi, L.ic.Gt = Gt;
    } else {
      const Q4 qf{L.filt[6], L.filt[7], L.filt[8], L.filt[9]};
      const V3 da = quat2axis_fast(qmul(qinverse(q), qf));
      if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          L.ic.d[0 + k] = L.filt[0 + k] - L.stage[0 + k];
          L.ic.d[3 + k] = L.filt[3 + k] - L.stage[3 + k];
          L.ic.d[9 + k] = L.filt[10 + k] - L.stage[10 + k];
          L.ic.d[12 + k] = L.filt[13 + k] - L.stage[13 + k];
          L.ic.d[15 + k] = L.filt[16 + k] - L.stage[16 + k];
        }
        L.ic.d[6] = da.x, L.ic.d[7] = da.y, L.ic.d[8] = da.z;
      }
    }
  }
  if (tid == 0) {
    L.res_last = L.stage[19], L.res_prev = L.stage[20], L.upd_norm = L.stage[21];
    L.conv = L.stage_flags[1], L.div = div;
    L.iter = iter + 1;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------
// Exhaustive search of ONE query by the whole workgroup — the reference's literal rules on the cloud
// in its original order: nearest neighbour = minimum of (distance, index) below the search radius
// (SE:847-851; skipped when the list certified it: known_j1 >= 0), second / third point = minima of
// (distance, visit rank) over the indices the walk reaches, by ring class (SE:859-910, 983-1024).
// Returns original indices (-1: none) to every thread.
// Afterwards the query's class claims are RE-ESTABLISHED so that the next iterations decide from the list
// again: a second sweep of the walk's index range appends every candidate within r2 / r3 = winner's
// distance + margin (or the search radius + margin when a class has no winner) of the query's current
// position; the owner lane then moves the class anchor there (SplitQ::b, jc).
// ---------------------------------------------------------------------------
struct BruteOut {
  int j1, j2, j3;
  int new_count;  // list length after the re-gather (> kSplitK: it did not fit, nothing was claimed)
  float r2, r3;
};
__device__ __noinline__ BruteOut brute_query(const float4* __restrict__ tg, int n, int nq, bool is_surf, float thr, float margin,
                                             float sx, float sy, float sz, int known_j1, int old_count, float4* __restrict__ cb,
                                             int stride) {
  Lds& L = g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long sentinel = (unsigned long long)__float_as_uint(thr) << 32;
  BruteOut r{-1, -1, -1, kSplitK + 1, 0.f, 0.f};
  __syncthreads();  // (L.red / L.gcount of the previous query have been read)
  if (tid == 0) L.gcount = old_count;
  if (known_j1 >= 0) {  // the nearest neighbour was certified from the list: only the walk is open (uniform)
    r.j1 = known_j1;
  } else {
    unsigned long long k1 = sentinel;
    for (int j0 = tid; j0 < n; j0 += 4 * kBlock) {  // (four loads in flight per trip)
      float4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = tg[j0 + u * kBlock < n ? j0 + u * kBlock : n - 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (j0 + u * kBlock >= n) break;
        const unsigned long long k = pack_key(sqdist3(t[u].x, t[u].y, t[u].z, sx, sy, sz), j0 + u * kBlock);
        k1 = k < k1 ? k : k1;
      }
    }
    k1 = wave_min_u64(k1);
    if (lane == 0) L.red[wave][0] = k1;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kWaves; ++w) k1 = L.red[w][0] < k1 ? L.red[w][0] : k1;
    r.j1 = k1 < sentinel ? (int)(unsigned)k1 : -1;
  }
  if (r.j1 < 0) return r;  // (uniform)
  const int j1 = r.j1, rho = ring_of(tg[j1].w);
  const Walk w = make_walk(L.ring_start[is_surf ? 0 : 1], n, nq, j1, rho);
  unsigned long long k2 = sentinel, k3 = sentinel;
  const int w_hi = w.f_hi > j1 ? w.f_hi : j1;  // (the forward part may be empty: its end is bounded by the QUERY count)
  // One sweep of the walk's index range, kBruteHold points per thread kept in registers (their loads all in flight
  // together: a dependent load per point would cost a memory round trip each) — the re-gather below reuses them;
  // a range beyond kBlock * kBruteHold points (more than ~3 full rings) sweeps the rest from memory again.
  constexpr int kBruteHold = 8;
  float4 held[kBruteHold];
#pragma unroll
  for (int u = 0; u < kBruteHold; ++u) {
    const int j = w.b_lo + tid + u * kBlock;
    held[u] = j < w_hi ? tg[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto visit = [&](int j, const float4& t) {
    int rank;
    if (!walk_rank(w, j, rank)) return;  // (j1 itself)
    const bool on_rho = ring_of(t.w) == rho;
    const unsigned long long k = pack_key(sqdist3(t.x, t.y, t.z, sx, sy, sz), rank);
    if (is_surf ? on_rho : !on_rho)
      k2 = k < k2 ? k : k2;
    else if (is_surf)
      k3 = k < k3 ? k : k3;
  };
#pragma unroll
  for (int u = 0; u < kBruteHold; ++u) {
    const int j = w.b_lo + tid + u * kBlock;
    if (j < w_hi) visit(j, held[u]);
  }
  for (int j = w.b_lo + tid + kBruteHold * kBlock; j < w_hi; j += kBlock) visit(j, tg[j]);
  k2 = wave_min_u64(k2), k3 = wave_min_u64(k3);
  if (lane == 0) L.red[wave][1] = k2, L.red[wave][2] = k3;
  __syncthreads();
#pragma unroll
  for (int ww = 0; ww < kWaves; ++ww) {
    k2 = L.red[ww][1] < k2 ? L.red[ww][1] : k2;
    k3 = L.red[ww][2] < k3 ? L.red[ww][2] : k3;
  }
  auto index_of = [&](unsigned long long k) {
    if (!(k < sentinel)) return -1;
    const int rank = (int)(unsigned)k;
    return rank >= kBackRank ? j1 - (rank - kBackRank) : j1 + rank;
  };
  r.j2 = index_of(k2), r.j3 = index_of(k3);
  // re-gather: radii from the exact winners (what split_gather does in the grid kernel)
  auto radius = [&](unsigned long long k) { return sqrtf(k < sentinel ? __uint_as_float((unsigned)(k >> 32)) : thr) * (1.f + 2e-6f) + margin; };
  r.r2 = radius(k2), r.r3 = is_surf ? radius(k3) : 0.f;
  const float t2 = r.r2 * r.r2 * (1.f + 4e-6f), t3 = r.r3 * r.r3 * (1.f + 4e-6f);
  auto regather = [&](int j, const float4& t) {
    int rank;
    if (!walk_rank(w, j, rank)) return;
    const int ring = ring_of(t.w);
    const bool cls2 = is_surf ? ring == rho : ring != rho;
    if (!cls2 && !is_surf) return;
    if (sqdist3(t.x, t.y, t.z, sx, sy, sz) <= (cls2 ? t2 : t3)) {
      const int k = atomicAdd(&L.gcount, 1);
      if (k < kSplitK) cb[(size_t)k * stride] = split_pack(t.x, t.y, t.z, j, ring);
    }
  };
#pragma unroll
  for (int u = 0; u < kBruteHold; ++u) {
    const int j = w.b_lo + tid + u * kBlock;
    if (j < w_hi) regather(j, held[u]);
  }
  for (int j = w.b_lo + tid + kBruteHold * kBlock; j < w_hi; j += kBlock) regather(j, tg[j]);
  __syncthreads();
  r.new_count = L.gcount;
  return r;
}

// ---------------------------------------------------------------------------
template <bool PROF>
__global__ __launch_bounds__(kBlock, 4) void ieskf_k1_kernel(
    DevParams prm, const ScanDesc* __restrict__ descs, const float4* __restrict__ arena, const SplitScan* __restrict__ hand,
    SplitQ* hq, float4* hcand, const double* __restrict__ state_in,
    const double* __restrict__ cov_in, double* __restrict__ state_out, double* __restrict__ a6_out, OutRec* __restrict__ out,
    lins_pose_record* __restrict__ poses, int scan_id_base, lins_corr* __restrict__ dump, int dump_iter,
    long long* __restrict__ prof_buf) {
  Lds& L = g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int scan = blockIdx.x;
  const SplitScan* hs = hand + scan;
  if (hs->status != SPLIT_CONTINUE) return;  // finished (converged / diverged / out of iterations) in the grid kernel
  const long long t_begin = PROF ? clock64() : 0;
  const ScanDesc sd = descs[scan];
  const int total = sd.n_surf_q + sd.n_corner_q;

  for (int k = tid; k < 324; k += kBlock) L.P[k] = cov_in[(size_t)scan * 324 + k];
  if (tid < 19) {
    L.filt[tid] = state_in[(size_t)scan * 19 + tid];
    L.ic.lin[tid] = hs->lin[tid];
  }
  if (tid >= 64 && tid < 64 + 2 * (kRingsBinned + 1)) {
    const int k = tid - 64;
    L.ring_start[k / (kRingsBinned + 1)][k % (kRingsBinned + 1)] = hs->ring_start[k / (kRingsBinned + 1)][k % (kRingsBinned + 1)];
  }
  if (tid < 28) L.sums[tid] = 0;
  if (tid == 0) {
    L.res_prev = hs->res_prev, L.res_last = hs->res_last, L.upd_norm = hs->upd_norm;
    L.iter = hs->iter, L.conv = 0, L.div = 0, L.m_surf = 0, L.m_corner = 0, L.brute_total = 0;
    for (int k = 0; k < 8; ++k) L.prof[k] = 0;
  }
  __syncthreads();
  {  // candidate lists -> LDS (coalesced: lane = query, one list slot per trip)
    int run = 0;
    for (int base = 0; base < total; base += kBlock) {
      const int slot = base + tid;
      int cnt = 0;
      if (slot < total) cnt = __float_as_int(ld_fresh(reinterpret_cast<const float4*>(hq + sd.slot_base + slot) + 2).y) & 0xFF;
      const int off = run + block_exclusive_scan(cnt, tid, L.scan_tmp);
      if (tid == kBlock - 1) L.scan_tmp[6] = off + cnt;
      if (slot < total) {
        const bool fits = off + cnt <= kPool;
        L.qoff[slot] = fits ? (unsigned short)off : kNotInLds;
        if (fits) {
          const float4* cb = hcand + (size_t)kSplitK * sd.slot_base + slot;
          for (int k = 0; k < cnt; ++k) L.pool[off + k] = ld_fresh(cb + (size_t)k * total);
        }
      }
      __syncthreads();
      run = L.scan_tmp[6];
      __syncthreads();
    }
  }
  if (tid == 64) lins_sinc_cos_table(L.trig);
  if (tid < 64) {  // wave 0, lane-redundant: constants of the first iteration
    IterConst ic;
    double filt[19];
    for (int k = 0; k < 19; ++k) ic.lin[k] = L.ic.lin[k], filt[k] = L.filt[k];
    make_iter_const_tail(filt, ic);
    if (tid == 0) {
      L.ic.phi = ic.phi, L.ic.Rt = ic.Rt, L.ic.Gt = ic.Gt;
      for (int k = 0; k < 18; ++k) L.ic.d[k] = ic.d[k];
    }
  }
  __syncthreads();

  const float thr = prm.nearest_f;
  const unsigned long long sentinel = (unsigned long long)__float_as_uint(thr) << 32;
  for (;;) {
    const int iter = L.iter;
    if (iter >= prm.num_iter || L.conv || L.div) break;
    __syncthreads();  // everyone has read the loop state before it is rewritten
    if (tid == 0) L.m_surf = 0, L.m_corner = 0;
    double acc = 0;
    int ms = 0, mc = 0;
    long long pt[6] = {PROF ? clock64() : 0, 0, 0, 0, 0, 0};  // phase clocks of the PROF variant (lane 0 of wave 0 reports)
    for (int base = 0; base < total; base += kBlock) {
      const int slot = base + tid;
      const bool active = slot < total;
      const bool is_surf = slot < sd.n_surf_q;
      const int qi = is_surf ? slot : slot - sd.n_surf_q;
      const int nq = is_surf ? sd.n_surf_q : sd.n_corner_q, nt = is_surf ? sd.n_surf_t : sd.n_corner_t;
      const int* ring_start = L.ring_start[is_surf ? 0 : 1];
      if (tid == 0) L.nbrute = 0;
      __syncthreads();
      float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
      QueryOut o;
      o.accepted = 0;
      o.c[0] = o.c[1] = o.c[2] = o.c[3] = 0.f;
      o.sel[0] = o.sel[1] = o.sel[2] = 0.f;
      float4 t1 = q, t2 = q, t3 = q;  // the three target points
      int j1 = -1, j2 = -1, j3 = -1;
      int my_count = 0;  // length of this query's candidate list
      int known1 = -1;  // certified nearest neighbour of a query whose second / third point need the exhaustive search
      int why = 0;  // debug: why the decision was not certified (1 no candidate, 2 NN, 3 ring, 4 second, 5 third point)
      bool was_brute = false;
      const V3 phi = L.ic.phi;
      if (active) {
        q = arena[(is_surf ? sd.off_surf_q : sd.off_corner_q) + qi];
        const V3 t{L.ic.lin[0], L.ic.lin[1], L.ic.lin[2]};
        if (prm.pad & 1024)
          o.sel[0] = q.x, o.sel[1] = q.y, o.sel[2] = q.z;  // (timing experiments: pad bits 1024..16384 drop one phase each)
        else
          transform_to_start(prm, phi, t, q, o.sel[0], o.sel[1], o.sel[2], g.trig);
        if (PROF) pt[1] = clock64();
        const float sx = o.sel[0], sy = o.sel[1], sz = o.sel[2];
        SplitQ mq;
        {
          const float4* mp = reinterpret_cast<const float4*>(hq + sd.slot_base + slot);
          const float4 m0 = ld_fresh(mp), m1 = ld_fresh(mp + 1), m2 = ld_fresh(mp + 2);
          mq.ax = m0.x, mq.ay = m0.y, mq.az = m0.z, mq.r_nn = m0.w;
          mq.bx = m1.x, mq.by = m1.y, mq.bz = m1.z, mq.r2 = m1.w;
          mq.r3 = m2.x, mq.meta = __float_as_int(m2.y), mq.j1 = __float_as_int(m2.z), mq.jc = __float_as_int(m2.w);
        }
        if (prm.pad & 16384) mq.meta = 0;
        const int count = mq.meta & 0xFF, rho0 = (mq.meta >> 8) & 0xFF, flags = mq.meta >> 16;
        const float4* cb = hcand + (size_t)kSplitK * sd.slot_base + slot;
        const int loff = L.qoff[slot];
        const bool in_lds = loff != kNotInLds;
        auto cand = [&](int k) { return in_lds ? L.pool[loff + k] : ld_fresh(cb + (size_t)k * total); };
        my_count = count;
        auto dist_from = [&](float x, float y, float z) {
          const float ex = sx - x, ey = sy - y, ez = sz - z;
          return sqrtf(ex * ex + ey * ey + ez * ez);
        };
        const float drift = dist_from(mq.ax, mq.ay, mq.az), drift_c = dist_from(mq.bx, mq.by, mq.bz);
        // --- one pass over the candidates (their loads issued four at a time): the nearest neighbour = minimum of
        // (distance, index) strictly below the radius, and — speculating that it is still the anchor's j1 — the
        // second / third point = minima of (distance, visit rank) over the candidates the walk around j1 reaches.
        // A changed nearest neighbour repeats the pass for the walk around the new one.
        unsigned long long k1 = sentinel, k2 = sentinel, k3 = sentinel;
        int c1 = -1, c2 = -1, c3 = -1;
        Walk w = make_walk(ring_start, nt, nq, mq.j1 >= 0 ? mq.j1 : 0, rho0);
        int rho = rho0;
        auto pass = [&](bool with_nn) {
          k2 = k3 = sentinel, c2 = c3 = -1;
#pragma unroll 1
          for (int k0 = 0; k0 < count; k0 += 4) {
            float4 cv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) cv[u] = cand(k0 + u < count ? k0 + u : count - 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              if (k0 + u >= count) break;
              const float4 c = cv[u];
              const int bits = __float_as_int(c.w), j = bits & 0xFFFF, ring = (bits >> 16) & 0xFF;
              const float d = sqdist3(c.x, c.y, c.z, sx, sy, sz);
              if (with_nn) {
                const unsigned long long key = pack_key(d, j);
                if (key < k1) k1 = key, c1 = k0 + u;
              }
              int rank;
              if (!walk_rank(w, j, rank)) continue;
              const unsigned long long key = pack_key(d, rank);
              const bool on_rho = ring == rho;
              if (is_surf ? on_rho : !on_rho) {
                if (key < k2) k2 = key, c2 = k0 + u;
              } else if (is_surf) {
                if (key < k3) k3 = key, c3 = k0 + u;
              }
            }
          }
        };
        pass(true);
        bool ok = c1 >= 0 && certified(__uint_as_float((unsigned)(k1 >> 32)), mq.r_nn, drift);
        // nobody listed inside the search radius, and everybody unlisted provably outside it: still no neighbour
        // (SE:851 drops the feature) — j1 = j2 = j3 = -1 stand
        const bool still_none = c1 < 0 && certified(thr, mq.r_nn, drift);
        if (!ok && !still_none) why = c1 < 0 ? 1 : 2;
        if (still_none) ok = true;
        if (ok && c1 >= 0) {
          t1 = cand(c1);
          j1 = __float_as_int(t1.w) & 0xFFFF;
          rho = (__float_as_int(t1.w) >> 16) & 0xFF;
          ok = mq.jc >= 0 || rho == rho0;  // (the grid kernel's ring claims are stated relative to rho0)
          if (!ok) why = 3;
          if (j1 != mq.j1) {  // the nearest neighbour moved: the walk's index intervals moved with it
            w = make_walk(ring_start, nt, nq, j1, rho);
            pass(false);
          }
          // is there anything at all the walk could visit in a class?  (index ranges only: ring_start is sorted)
          const int rs = ring_start[rho], re = ring_start[rho + 1];
          const bool any_same = j1 > (w.b_lo > rs ? w.b_lo : rs) || (j1 + 1 < (w.f_hi < re ? w.f_hi : re));
          const bool any_other = rs > w.b_lo || w.f_hi > re;
          auto judge = [&](int cw, unsigned long long kw, float rc, bool none_flag, bool any) {
            if (!any) return true;  // no index of this class is reachable: "none" whatever the geometry
            // claims the list kernel re-established hold for nearest neighbour jc; of the grid kernel's, a NONE claim
            // holds for the anchor's nearest neighbour only
            const bool holds = mq.jc >= 0 ? j1 == mq.jc : !(none_flag && j1 != mq.j1);
            return certified(cw >= 0 ? __uint_as_float((unsigned)(kw >> 32)) : thr, holds ? rc : 0.f, drift_c);
          };
          if (is_surf) {
            if (ok && !judge(c2, k2, mq.r2, (flags & SPLITQ_NONE2) != 0, any_same)) ok = false, why = 4;
            if (ok && !judge(c3, k3, mq.r3, (flags & SPLITQ_NONE3) != 0, any_other)) ok = false, why = 5;
          } else {
            if (ok && !judge(c2, k2, mq.r2, (flags & SPLITQ_NONE2) != 0, any_other)) ok = false, why = 4;
          }
          if (c2 >= 0) t2 = cand(c2), j2 = __float_as_int(t2.w) & 0xFFFF;
          if (c3 >= 0) t3 = cand(c3), j3 = __float_as_int(t3.w) & 0xFFFF;
        }
        if (!ok && (prm.pad & 64)) ok = true, j1 = j2 = j3 = -1;  // (timing aid: no exhaustive searches — wrong results)
        if (!ok) {  // not certified: exhaustive search below (of the walk only when the nearest neighbour is certified)
          const int k = atomicAdd(&L.nbrute, 1);
          if (k < kBruteCap) L.brute_q[k] = (unsigned short)tid;
          known1 = why >= 3 ? j1 : -1;
          j1 = j2 = j3 = -2;
          was_brute = true;
        }
      }
      if (PROF) pt[2] = clock64();
      __syncthreads();
      const int nb = L.nbrute;  // (uniform)
      for (int b = 0; b < nb; ++b) {
        const int owner = L.brute_q[b];  // (order of the atomics: irrelevant, every entry is handled)
        const int oslot = base + owner;
        const bool osurf = oslot < sd.n_surf_q;
        // the owner's de-skewed query travels through LDS (red is free between queries)
        if (tid == owner) {
          L.aug[0][0] = (double)o.sel[0], L.aug[0][1] = (double)o.sel[1], L.aug[0][2] = (double)o.sel[2];
          L.aug[0][3] = (double)known1, L.aug[0][4] = (double)my_count;
        }
        __syncthreads();
        const float bx = (float)L.aug[0][0], by = (float)L.aug[0][1], bz = (float)L.aug[0][2];
        const int bknown = (int)L.aug[0][3], bcount = (int)L.aug[0][4];
        const float4* tg = arena + (osurf ? sd.off_surf_t : sd.off_corner_t);
        const BruteOut br = brute_query(tg, osurf ? sd.n_surf_t : sd.n_corner_t, osurf ? sd.n_surf_q : sd.n_corner_q, osurf, thr,
                                        prm.split_margin, bx, by, bz, bknown, bcount,
                                        hcand + (size_t)kSplitK * sd.slot_base + oslot, total);
        if (tid == owner) {
          j1 = br.j1, j2 = br.j2, j3 = br.j3;
          if (br.j1 >= 0) t1 = tg[br.j1];
          if (br.j2 >= 0) t2 = tg[br.j2];
          if (br.j3 >= 0) t3 = tg[br.j3];
          if (br.j1 >= 0 && br.new_count <= kSplitK) {  // the class claims now hold around here, for this nearest neighbour
            SplitQ* mq = hq + sd.slot_base + oslot;
            mq->bx = o.sel[0], mq->by = o.sel[1], mq->bz = o.sel[2], mq->r2 = br.r2, mq->r3 = br.r3;
            mq->meta = (mq->meta & ~0xFF) | br.new_count;
            mq->jc = br.j1;
            L.qoff[oslot] = kNotInLds;  // (the list grew in the hand-off buffer: read it from there from now on)
          }
        }
        __syncthreads();
      }
      if (nb && tid == 0) L.brute_total += nb;
      if (PROF) pt[3] = clock64();
      double row[7] = {0, 0, 0, 0, 0, 0, 0};
      asm volatile("" ::: "memory");  // (R^T, G^T are read from LDS here instead of hoisted over the list pass and spilled)
      if (active) {
        if (is_surf) {
          if (j1 >= 0 && j2 >= 0 && j3 >= 0) surf_row(prm, iter, o.sel[0], o.sel[1], o.sel[2], t1, t2, t3, o);
        } else if (j1 >= 0 && j2 >= 0) {
          corner_row(prm, iter, o.sel[0], o.sel[1], o.sel[2], t1, t2, o);
        }
        if (o.accepted) {
          const V3 cv{(double)o.c[0], (double)o.c[1], (double)o.c[2]};
          const V3 u = cross(V3{(double)q.x, (double)q.y, (double)q.z}, mvec(L.ic.Rt, cv));
          const V3 a = mvec(L.ic.Gt, u);
          row[0] = cv.x, row[1] = cv.y, row[2] = cv.z, row[3] = a.x, row[4] = a.y, row[5] = a.z;
          row[6] = prm.lidar_scale * (double)o.c[3];
          if (is_surf)
            ++ms;
          else
            ++mc;
        }
        if (dump && iter == dump_iter) {
          lins_corr r;
          r.ind1 = j1, r.ind2 = j2, r.ind3 = is_surf ? j3 : -1;
          r.accepted = o.accepted | (was_brute ? 256 : 0) | (why << 9) | ((hq[sd.slot_base + slot].meta & 0xFF) << 16) | (((hq[sd.slot_base + slot].meta >> 16) & 3) << 24);  // (debug bits)
          for (int k = 0; k < 4; ++k) r.coeff[k] = o.c[k];
          r.sel[0] = o.sel[0], r.sel[1] = o.sel[1], r.sel[2] = o.sel[2], r.sel[3] = q.w;
          dump[sd.slot_base + slot] = r;
        }
      }
      if (PROF) pt[4] = clock64();
      if (!(prm.pad & 2048)) acc += wave_reduce_rows(row, lane);
      if (PROF) pt[5] = clock64();
    }
    {
      const int sidx28 = reduce_sum_index(lane);
      if (sidx28 >= 0) L.partial[wave * 28 + sidx28] = acc;
    }
    if (ms) atomicAdd(&L.m_surf, ms);
    if (mc) atomicAdd(&L.m_corner, mc);
    __syncthreads();
    if (tid < 28) {
      double sacc = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) sacc += L.partial[w * 28 + tid];
      L.sums[tid] = sacc;
    }
    __syncthreads();
    const long long ts0 = PROF ? clock64() : 0;
    solve_and_update(prm.r2, prm.fixed_iters, prm.pad, tid, iter);
    if (PROF && tid == 0) {
      // [0] de-skew [1] meta + list pass + certificates [2] wait + exhaustive searches [3] rows [4] wave reduction
      // [5] barriers + fold of the partials [6] solve + update   (thread 0's view, summed over the iterations)
      long long* pp = L.prof;  // (LDS: a global accumulator would put its own memory latency into the next phase)
      const long long te = clock64();
      pp[0] += pt[1] - pt[0], pp[1] += pt[2] - pt[1], pp[2] += pt[3] - pt[2], pp[3] += pt[4] - pt[3], pp[4] += pt[5] - pt[4];
      pp[5] += ts0 - pt[5], pp[6] += te - ts0, pp[7] += 1;
    }
  }

  // ---- hand-off to the Joseph kernel / the caller (SE:585-598), as the persistent kernel does ----
  const int div = L.div;
  if (tid < 19) state_out[(size_t)scan * 19 + tid] = div ? L.filt[tid] : L.ic.lin[tid];
  if (tid < 21) a6_out[(size_t)scan * 21 + tid] = L.sums[tid];
  if (tid == 0) {
    OutRec r;
    r.residual_norm = L.res_last, r.update_norm = L.upd_norm;
    r.iters = L.iter, r.converged = L.conv, r.diverged = div;
    r.m_surf = L.m_surf, r.m_corner = L.m_corner;
    r.pad[0] = hs->dbg[0], r.pad[1] = hs->dbg[1], r.pad[2] = L.brute_total;
    out[scan] = r;
    if (PROF) {
      for (int k = 0; k < 8; ++k) prof_buf[(size_t)scan * 16 + k] = L.prof[k];
      prof_buf[(size_t)scan * 16 + 12] = clock64() - t_begin;
    }
  }
  if (poses && tid < 32) {
    lins_pose_record* pr = poses + scan;
    const double* st = div ? L.filt : L.ic.lin;
    if (tid < 19) pr->state[tid] = st[tid];
    if (tid == 19) pr->residual_norm = L.res_last;
    if (tid == 20) {
      pr->iters = L.iter, pr->converged = L.conv, pr->diverged = div;
      pr->m_surf = L.m_surf, pr->m_corner = L.m_corner, pr->scan_id = scan_id_base + scan;
      pr->pad[0] = pr->pad[1] = 0;
    }
  }
}

}  // namespace k1

size_t split_scan_size() { return sizeof(SplitScan); }
size_t split_q_size() { return sizeof(SplitQ); }
size_t split_cand_slots() { return kSplitK; }

void launch_k1(hipStream_t stream, int n, const DevParams& prm, const ScanDesc* descs, const float4* arena, const void* hand,
               const void* hq, const float4* hcand, const double* state_in, const double* cov_in, double* state_out, double* a6,
               void* out, lins_pose_record* poses, int scan_id_base, lins_corr* dump, int dump_iter, long long* prof) {
  if (prof)
    hipLaunchKernelGGL((k1::ieskf_k1_kernel<true>), dim3(n), dim3(k1::kBlock), 0, stream, prm, descs, arena, (const SplitScan*)hand,
                       (SplitQ*)hq, (float4*)hcand, state_in, cov_in, state_out, a6, (k1::OutRec*)out, poses, scan_id_base, dump,
                       dump_iter, prof);
  else
    hipLaunchKernelGGL((k1::ieskf_k1_kernel<false>), dim3(n), dim3(k1::kBlock), 0, stream, prm, descs, arena, (const SplitScan*)hand,
                       (SplitQ*)hq, (float4*)hcand, state_in, cov_in, state_out, a6, (k1::OutRec*)out, poses, scan_id_base, dump,
                       dump_iter, prof);
}

}  // namespace lins
