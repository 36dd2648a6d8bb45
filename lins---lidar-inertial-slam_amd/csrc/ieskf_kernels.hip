// ieskf_kernels.hip — CDNA4 (gfx950) kernels of the IESKF update path.
//
// ieskf_persistent_kernel: one workgroup (5 wave64) owns one scan pair for the
// WHOLE iterated update (<= NUM_ITER iterations, no host round trips, SE:475-583):
//
//   once          counting-sort both target clouds into (ring x azimuth-column)
//                 grids — a coarse range image of the previous scan (ieskf_binned.h)
//   per iteration
//     all lanes   one query feature each: de-skew (f64) -> exact NN + index walk
//                 (f32, grid-pruned) -> plane/line residual + Jacobian (f64 -> f32)
//                 -> H row (c, a = G^T (p x R^T c), r) into an LDS slot [A2-A5]
//     280 lanes   28 f64 sums (upper triangle of the 6x6 H^T H, H^T r, r^T r):
//                 fixed-shape two-stage tree (10 strided groups -> ordered fold),
//                 bit-reproducible from run to run
//     <=42 lanes  (sigma^2 I + A P_SS) w = (g + A d) by pivoted elimination in LDS,
//                 dx = d - P[:,S] w                                        [A6]
//     wave 0      NaN / divergence / convergence tests, boxPlus, next constants [A7]
// ieskf_joseph_kernel: once per scan after the loop, the Joseph covariance update
//   with the last iteration's A (SE:594-598) — split off so that its 18x18 algebra
//   does not dictate the register budget (occupancy) of the search loop.
//
// No dense contraction anywhere => MFMA unused; the path is gather/VALU work over
// HBM-resident clouds that are read from HBM once per update and re-gathered from
// L2 on every iteration.

#include <hip/hip_runtime.h>

#include <type_traits>

#include "ieskf_binned.h"
#include "ieskf_device.h"
#include "ieskf_joseph.h"

namespace lins {

struct OutRec {
  double residual_norm, update_norm;
  int iters, converged, diverged, m_surf, m_corner, pad[3];
};

// row vector v = (c0 c1 c2 a0 a1 a2 r); sum k accumulates v[A[k]] * v[B[k]]:
// [0..20] upper triangle of H^T H (row-major), [21..26] H^T r, [27] r^T r
__constant__ unsigned char kPairA[28] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2,
                                         2, 3, 3, 3, 4, 4, 5, 0, 1, 2, 3, 4, 5, 6};
__constant__ unsigned char kPairB[28] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4,
                                         5, 3, 4, 5, 4, 5, 5, 6, 6, 6, 6, 6, 6, 6};

struct NoBins {};

// LDS views of the two target grids of the scan this workgroup owns
struct ScanBins {
  CloudBins surf, corner;
};

// ---------------------------------------------------------------------------
// one query feature -> (indices, accepted, coeff)
// ---------------------------------------------------------------------------
template <int SEARCH>
__device__ __forceinline__ void process_surf(const DevParams& prm, const ScanDesc& sd,
                                             const float4* __restrict__ arena, const ScanBins* sb, const V3& phi,
                                             const V3& t, int iter, bool do_search, const float4& q, QueryOut& o) {
  const float4* tg = arena + sd.off_surf_t;
  transform_to_start(prm, phi, t, q, o.sel[0], o.sel[1], o.sel[2]);
  o.accepted = 0;
  o.c[0] = o.c[1] = o.c[2] = o.c[3] = 0.f;
  if (do_search) {
    int j1;
    float d1;
    o.j1 = o.j2 = o.j3 = -1;
    if (prm.pad & 2) {
      j1 = -1, d1 = 0;
    } else if (SEARCH == SEARCH_BINNED && sd.surf_sorted) {
      nn_binned(sb->surf, o.sel[0], o.sel[1], o.sel[2], prm.nearest_f, ring_of(q.w), j1, d1);
      if (j1 >= 0 && (double)d1 < prm.nearest && !(prm.pad & 1)) {
        o.j1 = j1;
        walk_surf_binned(sb->surf, sd.n_surf_q, prm.nearest_f, j1, ring_of(tg[j1].w), o.sel[0], o.sel[1], o.sel[2],
                         o.j2, o.j3);
      }
    } else {
      nn_brute(tg, sd.n_surf_t, o.sel[0], o.sel[1], o.sel[2], j1, d1);
      if (j1 >= 0 && (double)d1 < prm.nearest) {
        o.j1 = j1;
        walk_surf_literal(tg, sd.n_surf_t, sd.n_surf_q, prm.nearest_f, j1, o.sel[0], o.sel[1], o.sel[2], o.j2, o.j3);
      }
    }
  }
  if (o.j2 >= 0 && o.j3 >= 0)
    surf_row(prm, iter, o.sel[0], o.sel[1], o.sel[2], tg[o.j1], tg[o.j2], tg[o.j3], o);
}

template <int SEARCH>
__device__ __forceinline__ void process_corner(const DevParams& prm, const ScanDesc& sd,
                                               const float4* __restrict__ arena, const ScanBins* sb, const V3& phi,
                                               const V3& t, int iter, bool do_search, const float4& q, QueryOut& o) {
  const float4* tg = arena + sd.off_corner_t;
  transform_to_start(prm, phi, t, q, o.sel[0], o.sel[1], o.sel[2]);
  o.accepted = 0;
  o.c[0] = o.c[1] = o.c[2] = o.c[3] = 0.f;
  if (do_search) {
    int j1;
    float d1;
    o.j1 = o.j2 = -1;
    o.j3 = -1;
    if (prm.pad & 2) {
      j1 = -1, d1 = 0;
    } else if (SEARCH == SEARCH_BINNED && sd.corner_sorted) {
      nn_binned(sb->corner, o.sel[0], o.sel[1], o.sel[2], prm.nearest_f, ring_of(q.w), j1, d1);
      if (j1 >= 0 && (double)d1 < prm.nearest && !(prm.pad & 1)) {
        o.j1 = j1;
        walk_corner_binned(sb->corner, sd.n_corner_q, prm.nearest_f, j1, ring_of(tg[j1].w), o.sel[0], o.sel[1],
                           o.sel[2], o.j2);
      }
    } else {
      nn_brute(tg, sd.n_corner_t, o.sel[0], o.sel[1], o.sel[2], j1, d1);
      if (j1 >= 0 && (double)d1 < prm.nearest) {
        o.j1 = j1;
        walk_corner_literal(tg, sd.n_corner_t, sd.n_corner_q, prm.nearest_f, j1, o.sel[0], o.sel[1], o.sel[2], o.j2);
      }
    }
  }
  if (o.j2 >= 0) corner_row(prm, iter, o.sel[0], o.sel[1], o.sel[2], tg[o.j1], tg[o.j2], o);
}

// slot -> query (surf slots first, then corner: the reference's concat order SE:499-504)
template <int SEARCH>
__device__ __forceinline__ void process_slot(const DevParams& prm, const ScanDesc& sd,
                                             const float4* __restrict__ arena, const ScanBins* sb, const V3& phi,
                                             const V3& t, int iter, bool do_search, int slot,
                                             int4* __restrict__ idx_store, QueryOut& o, float4& q, bool& is_surf) {
  is_surf = slot < sd.n_surf_q;
  int i = is_surf ? slot : slot - sd.n_surf_q;
  q = arena[(is_surf ? sd.off_surf_q : sd.off_corner_q) + i];
  if (!do_search) {
    int4 s = idx_store[sd.slot_base + slot];
    o.j1 = s.x, o.j2 = s.y, o.j3 = s.z;
  }
  if (is_surf)
    process_surf<SEARCH>(prm, sd, arena, sb, phi, t, iter, do_search, q, o);
  else
    process_corner<SEARCH>(prm, sd, arena, sb, phi, t, iter, do_search, q, o);
  if (do_search && prm.icp_freq > 1) idx_store[sd.slot_base + slot] = make_int4(o.j1, o.j2, o.j3, 0);
}

// One round of <= kRowsCap query slots: each lane turns its query into an H row in LDS.
template <int SEARCH>
__device__ __forceinline__ void correspondence_round(const DevParams& prm, const ScanDesc& sd,
                                                     const float4* __restrict__ arena, const ScanBins* sb,
                                                     const IterConst& ic, int iter, bool do_search, int base,
                                                     int total, int4* __restrict__ idx_store, double* rows, int tid,
                                                     int& ms, int& mc, lins_corr* __restrict__ dump) {
  const int slot = base + tid;
  double row[7] = {0, 0, 0, 0, 0, 0, 0};
  if (slot < total) {
    V3 phi = ic.phi;
    V3 t{ic.lin[0], ic.lin[1], ic.lin[2]};
    QueryOut o;
    float4 q;
    bool is_surf;
    process_slot<SEARCH>(prm, sd, arena, sb, phi, t, iter, do_search, slot, idx_store, o, q, is_surf);
    if (o.accepted) {
      // H row (SE:526-531): pos block c^T, att block c^T (-R [p]x) Rinvleft(-phi) = (G^T (p x R^T c))^T
      V3 c{(double)o.c[0], (double)o.c[1], (double)o.c[2]};
      V3 u = cross(V3{(double)q.x, (double)q.y, (double)q.z}, mvec(ic.Rt, c));
      V3 a = mvec(ic.Gt, u);
      row[0] = c.x, row[1] = c.y, row[2] = c.z;
      row[3] = a.x, row[4] = a.y, row[5] = a.z;
      row[6] = prm.lidar_scale * (double)o.c[3];
      if (is_surf)
        ++ms;
      else
        ++mc;
    }
    if (dump) {
      lins_corr r;
      r.ind1 = o.j1, r.ind2 = o.j2, r.ind3 = is_surf ? o.j3 : -1, r.accepted = o.accepted;
      for (int k = 0; k < 4; ++k) r.coeff[k] = o.c[k];
      r.sel[0] = o.sel[0], r.sel[1] = o.sel[1], r.sel[2] = o.sel[2], r.sel[3] = q.w;
      dump[slot] = r;
    }
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) rows[tid * 7 + k] = row[k];
}

// rows -> 28 sums.  Group g folds rows g, g+G, ... in order; then the G group partials
// are folded in order.  Same tree every run => deterministic.
__device__ __forceinline__ void accumulate_rows(const double* rows, int nrows, int tid, double& acc) {
  int g = tid >> 5, k = tid & 31;
  if (k < 28) {
    int a = kPairA[k], b = kPairB[k];
    for (int r = g; r < nrows; r += kRedGroups) acc += rows[r * 7 + a] * rows[r * 7 + b];
  }
}

// Build the (ring x column) grids of this scan's two target clouds (binned search only).
template <int SEARCH>
__device__ __forceinline__ void setup_bins(const ScanDesc& sd, const float4* __restrict__ arena,
                                           float4* __restrict__ binned, BinStorage* bs, ScanBins* sb, int tid) {
  if (SEARCH != SEARCH_BINNED) return;
  build_az_edges(bs->az_edge, tid);
  if (sd.surf_sorted)
    build_cloud_bins(arena + sd.off_surf_t, sd.n_surf_t, kAzSurf, bs->cell_surf, bs->ring_start[0], bs->el_bits[0],
                     bs->el[0], binned + sd.off_surf_t, tid, bs->scan_tmp);
  if (sd.corner_sorted)
    build_cloud_bins(arena + sd.off_corner_t, sd.n_corner_t, kAzCorner, bs->cell_corner, bs->ring_start[1],
                     bs->el_bits[1], bs->el[1], binned + sd.off_corner_t, tid, bs->scan_tmp);
  if (tid == 0) {
    sb->surf = CloudBins{bs->cell_surf, bs->ring_start[0], &bs->el[0][0][0], bs->az_edge, binned + sd.off_surf_t,
                         kAzSurf, 1, sd.n_surf_t};
    sb->corner = CloudBins{bs->cell_corner, bs->ring_start[1], &bs->el[1][0][0], bs->az_edge,
                           binned + sd.off_corner_t, kAzCorner, kAzSurf / kAzCorner, sd.n_corner_t};
  }
  __syncthreads();
}


// ---------------------------------------------------------------------------
// LDS layout of the persistent kernel
// ---------------------------------------------------------------------------
struct Shared {
  IterConst ic;
  double filt[19];
  double P[324];
  double rows[kRowsCap * 7];
  double partial[kRedGroups * 28];
  double sums[28];
  double aug[6 * 7];
  double w[6];
  double dx[18];
  double res_prev, res_last, upd_norm;
  int piv[6], used[6];
  int m_surf, m_corner;
  int iter, conv, div, pad;
};

// ---------------------------------------------------------------------------
// persistent IESKF kernel: grid = scans, block = kBlock
// ---------------------------------------------------------------------------
template <int SEARCH>
__global__ __launch_bounds__(kBlock, 4) void ieskf_persistent_kernel(
    DevParams prm, const ScanDesc* __restrict__ descs, const float4* __restrict__ arena,
    const double* __restrict__ state_in, const double* __restrict__ cov_in, double* __restrict__ state_out,
    double* __restrict__ a6_out, OutRec* __restrict__ out, int4* __restrict__ idx_store,
    lins_pose_record* __restrict__ poses, int scan_id_base, float4* __restrict__ binned,
    long long* __restrict__ prof) {
  __shared__ Shared sh;
  __shared__ std::conditional_t<SEARCH == SEARCH_BINNED, BinStorage, NoBins> bstore;
  __shared__ ScanBins sbins;
  const int tid = threadIdx.x;
  const int scan = blockIdx.x;
  const ScanDesc sd = descs[scan];
  const int total = sd.n_surf_q + sd.n_corner_q;
  // optional phase profile (prof != nullptr): shader-clock ticks per phase, per workgroup:
  // [0] setup+grid build [1] correspondence (critical path) [2] reduction [3] solve
  // [4] state update [5] total [6..10] per-wave correspondence time
  long long pt[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const long long t_begin = prof ? clock64() : 0;

  for (int k = tid; k < 324; k += kBlock) sh.P[k] = cov_in[(size_t)scan * 324 + k];
  if (tid < 19) {
    double v = state_in[(size_t)scan * 19 + tid];
    sh.filt[tid] = v;
    sh.ic.lin[tid] = v;
  }
  if (tid < 28) sh.sums[tid] = 0;
  if (tid == 0) {
    sh.res_prev = 1e6, sh.res_last = 0, sh.upd_norm = 0;
    sh.iter = 0, sh.conv = 0, sh.div = 0, sh.m_surf = 0, sh.m_corner = 0;
  }
  __syncthreads();
  if (tid < 64) {  // wave 0, lane-redundant: constants of iteration 0
    IterConst ic;
    double filt[19];
    for (int k = 0; k < 19; ++k) ic.lin[k] = filt[k] = sh.filt[k];
    make_iter_const(filt, ic);
    if (tid == 0) {
      sh.ic.phi = ic.phi, sh.ic.Rt = ic.Rt, sh.ic.Gt = ic.Gt;
      for (int k = 0; k < 18; ++k) sh.ic.d[k] = ic.d[k];
    }
  }
  setup_bins<SEARCH>(sd, arena, binned, (BinStorage*)&bstore, &sbins, tid);  // ends with a barrier
  __syncthreads();
  if (prof) pt[0] = clock64() - t_begin;

  for (;;) {
    const int iter = sh.iter;
    if (iter >= prm.num_iter || sh.conv || sh.div) break;
    __syncthreads();  // everyone has read the loop state before it is rewritten
    if (tid == 0) sh.m_surf = 0, sh.m_corner = 0;

    const bool do_search = (iter % prm.icp_freq) == 0;
    double acc = 0;
    int ms = 0, mc = 0;
    long long t0 = prof ? clock64() : 0, t1 = t0, t2 = t0, t3 = t0;
    for (int base = 0; base < total; base += kRowsCap) {
      correspondence_round<SEARCH>(prm, sd, arena, &sbins, sh.ic, iter, do_search, base, total, idx_store,
                                   sh.rows, tid, ms, mc, nullptr);
      if (prof) pt[6] += clock64() - t0;  // this wave's own search time (kept per lane, lane 0 of each wave reports)
      __syncthreads();
      if (prof) t1 = clock64();
      int nrows = total - base < kRowsCap ? total - base : kRowsCap;
      accumulate_rows(sh.rows, nrows, tid, acc);
      __syncthreads();
    }
    if ((tid & 31) < 28) sh.partial[(tid >> 5) * 28 + (tid & 31)] = acc;
    if (ms) atomicAdd(&sh.m_surf, ms);
    if (mc) atomicAdd(&sh.m_corner, mc);
    __syncthreads();
    if (tid < 28) {
      double s = 0;
#pragma unroll
      for (int g = 0; g < kRedGroups; ++g) s += sh.partial[g * 28 + tid];
      sh.sums[tid] = s;
    }
    __syncthreads();
    if (prof) t2 = clock64();

    // (sigma^2 I + A P_SS) w = g + A d_S     (push-through form of SE:542-549)
    if (tid < 36) {
      int i = tid / 6, j = tid - i * 6;
      double t = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) t += sym6(sh.sums, i, k) * sh.P[sidx(k) * 18 + sidx(j)];
      sh.aug[i * 7 + j] = t + (i == j ? prm.r2 : 0.0);
    } else if (tid < 42) {
      int i = tid - 36;
      double z = sh.sums[21 + i];
#pragma unroll
      for (int k = 0; k < 6; ++k) z += sym6(sh.sums, i, k) * sh.ic.d[sidx(k)];
      sh.aug[i * 7 + 6] = z;
    }
    __syncthreads();
    block_solve6(sh.aug, 7, sh.w, sh.piv, sh.used, tid);
    if (tid < 18) {  // dx = d - P[:,S] w
      double s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += sh.P[tid * 18 + sidx(k)] * sh.w[k];
      sh.dx[tid] = sh.ic.d[tid] - s;
    }
    __syncthreads();
    if (prof) t3 = clock64();

    if (tid < 64) {  // wave 0, lane-redundant: SE:552-580 + constants of the next iteration
      double dx[18];
      for (int k = 0; k < 18; ++k) dx[k] = sh.dx[k];
      double rn = sqrt(sh.sums[27]);
      bool has_nan = false;
      for (int k = 0; k < 18; ++k)
        if (isnan(dx[k])) has_nan = true;
      int div = 0, conv = 0;
      IterConst ic;
      for (int k = 0; k < 19; ++k) ic.lin[k] = sh.ic.lin[k];
      double un = sh.upd_norm, res_prev = sh.res_prev;
      if (has_nan) {
        div = 2;
      } else if (rn > res_prev * 10) {
        div = 1;
      } else {
        box_plus_inplace(ic.lin, dx);
        un = 0;
        for (int k = 0; k < 18; ++k) un += dx[k] * dx[k];
        un = sqrt(un);
        if (un <= 1e-2 && !prm.fixed_iters) conv = 1;
        res_prev = rn;
        double filt[19];
        for (int k = 0; k < 19; ++k) filt[k] = sh.filt[k];
        make_iter_const(filt, ic);
      }
      if (tid == 0) {
        if (!div) {
          for (int k = 0; k < 19; ++k) sh.ic.lin[k] = ic.lin[k];
          sh.ic.phi = ic.phi, sh.ic.Rt = ic.Rt, sh.ic.Gt = ic.Gt;
          for (int k = 0; k < 18; ++k) sh.ic.d[k] = ic.d[k];
        }
        sh.res_last = rn, sh.res_prev = res_prev, sh.upd_norm = un;
        sh.conv = conv, sh.div = div;
        sh.iter = iter + 1;
      }
    }
    __syncthreads();
    if (prof) {
      long long t4 = clock64();
      pt[1] += t1 - t0, pt[2] += t2 - t1, pt[3] += t3 - t2, pt[4] += t4 - t3;
    }
  }
  if (prof) {
    pt[5] = clock64() - t_begin;
    if (tid == 0)
      for (int k = 0; k < 6; ++k) prof[(size_t)scan * 16 + k] = pt[k];
    if ((tid & 63) == 0) prof[(size_t)scan * 16 + 6 + (tid >> 6)] = pt[6];
  }

  // ---- hand-off to the Joseph kernel / the caller (SE:585-598) ---------------
  const int div = sh.div;
  if (tid < 19) state_out[(size_t)scan * 19 + tid] = div ? sh.filt[tid] : sh.ic.lin[tid];
  if (tid < 21) a6_out[(size_t)scan * 21 + tid] = sh.sums[tid];  // A of the LAST executed iteration
  if (tid == 0) {
    OutRec r;
    r.residual_norm = sh.res_last, r.update_norm = sh.upd_norm;
    r.iters = sh.iter, r.converged = sh.conv, r.diverged = div;
    r.m_surf = sh.m_surf, r.m_corner = sh.m_corner;
    r.pad[0] = r.pad[1] = r.pad[2] = 0;
    out[scan] = r;
  }
  if (poses && tid < 32) {
    lins_pose_record* pr = poses + scan;
    const double* st = div ? sh.filt : sh.ic.lin;
    if (tid < 19) pr->state[tid] = st[tid];
    if (tid == 19) pr->residual_norm = sh.res_last;
    if (tid == 20) {
      pr->iters = sh.iter, pr->converged = sh.conv, pr->diverged = div;
      pr->m_surf = sh.m_surf, pr->m_corner = sh.m_corner, pr->scan_id = scan_id_base + scan;
      pr->pad[0] = pr->pad[1] = 0;
    }
  }
}

// ---------------------------------------------------------------------------
// Joseph covariance update (SE:594-598) in the reduced form, one workgroup per scan:
//   N = sigma^2 I + A P_SS,  Y = N^-1 A,  Z = Y N^-T
//   KH = P[:,S] Y E_S^T,  K R K^T = sigma^2 P[:,S] Z P[:,S]^T
//   P+ = (I - KH) P (I - KH)^T + K R K^T, symmetrised (enforceSymmetry).
// Diverged scans get the un-updated Pk_ back (SE:592).
// ---------------------------------------------------------------------------
constexpr int kJosephBlock = 128;

__global__ __launch_bounds__(kJosephBlock) void ieskf_joseph_kernel(DevParams prm, const double* __restrict__ cov_in,
                                                                    const double* __restrict__ a6_in,
                                                                    const OutRec* __restrict__ out,
                                                                    double* __restrict__ cov_out) {
  __shared__ double P[324], A[21];
  __shared__ JosephScratch scratch;
  const int tid = threadIdx.x, scan = blockIdx.x;
  for (int k = tid; k < 324; k += kJosephBlock) P[k] = cov_in[(size_t)scan * 324 + k];
  if (tid < 21) A[tid] = a6_in[(size_t)scan * 21 + tid];
  __syncthreads();
  joseph_update<kJosephBlock>(prm.r2, out[scan].diverged != 0, P, A, scratch, cov_out + (size_t)scan * 324, tid);
}

// ---------------------------------------------------------------------------
// single pass kernel (BASELINE.json configs[1]: device correspondences + reduction,
// host-side 18x18 solve).  One correspondence + residual/Jacobian pass for a
// caller-supplied linearisation state; optionally dumps per-query records and/or the
// 28 sums.
// ---------------------------------------------------------------------------
template <int SEARCH>
__global__ __launch_bounds__(kBlock) void ieskf_pass_kernel(
    DevParams prm, const ScanDesc* __restrict__ descs, const float4* __restrict__ arena,
    const double* __restrict__ lin_state, const double* __restrict__ filt_state, int iter,
    int4* __restrict__ idx_store, lins_corr* __restrict__ dump, double* __restrict__ sums_out,
    int* __restrict__ counts_out, float4* __restrict__ binned) {
  __shared__ std::conditional_t<SEARCH == SEARCH_BINNED, BinStorage, NoBins> bstore;
  __shared__ ScanBins sbins;
  __shared__ IterConst ic;
  __shared__ double rows[kRowsCap * 7];
  __shared__ double partial[kRedGroups * 28];
  __shared__ int cnt[2];
  const int tid = threadIdx.x, scan = blockIdx.x;
  const ScanDesc sd = descs[scan];
  const int total = sd.n_surf_q + sd.n_corner_q;
  if (tid < 64) {
    IterConst c;
    double filt[19];
    for (int k = 0; k < 19; ++k) c.lin[k] = lin_state[(size_t)scan * 19 + k], filt[k] = filt_state[(size_t)scan * 19 + k];
    make_iter_const(filt, c);
    if (tid == 0) {
      ic = c;
      cnt[0] = cnt[1] = 0;
    }
  }
  __syncthreads();
  setup_bins<SEARCH>(sd, arena, binned, (BinStorage*)&bstore, &sbins, tid);
  double acc = 0;
  int ms = 0, mc = 0;
  for (int base = 0; base < total; base += kRowsCap) {
    correspondence_round<SEARCH>(prm, sd, arena, &sbins, ic, iter, true, base, total, idx_store, rows, tid, ms,
                                 mc, dump ? dump + sd.slot_base : nullptr);
    __syncthreads();
    int nrows = total - base < kRowsCap ? total - base : kRowsCap;
    accumulate_rows(rows, nrows, tid, acc);
    __syncthreads();
  }
  if ((tid & 31) < 28) partial[(tid >> 5) * 28 + (tid & 31)] = acc;
  if (ms) atomicAdd(&cnt[0], ms);
  if (mc) atomicAdd(&cnt[1], mc);
  __syncthreads();
  if (sums_out && tid < 28) {
    double s = 0;
#pragma unroll
    for (int g = 0; g < kRedGroups; ++g) s += partial[g * 28 + tid];
    sums_out[(size_t)scan * 28 + tid] = s;
  }
  if (counts_out && tid < 2) counts_out[scan * 2 + tid] = cnt[tid];
}

// ---------------------------------------------------------------------------
// launchers (called from lins_capi.hip)
// ---------------------------------------------------------------------------
void launch_persistent(hipStream_t stream, int n, const DevParams& prm, const ScanDesc* descs,
                       const float4* arena, const double* state_in, const double* cov_in, double* state_out,
                       double* cov_out, double* a6, void* out, int4* idx_store, lins_pose_record* poses,
                       int scan_id_base, float4* binned, long long* prof) {
  if (prm.search == SEARCH_BINNED)
    hipLaunchKernelGGL(ieskf_persistent_kernel<SEARCH_BINNED>, dim3(n), dim3(kBlock), 0, stream, prm, descs, arena,
                       state_in, cov_in, state_out, a6, (OutRec*)out, idx_store, poses, scan_id_base, binned, prof);
  else
    hipLaunchKernelGGL(ieskf_persistent_kernel<SEARCH_BRUTE>, dim3(n), dim3(kBlock), 0, stream, prm, descs, arena,
                       state_in, cov_in, state_out, a6, (OutRec*)out, idx_store, poses, scan_id_base, binned, prof);
  hipLaunchKernelGGL(ieskf_joseph_kernel, dim3(n), dim3(kJosephBlock), 0, stream, prm, cov_in, a6, (const OutRec*)out,
                     cov_out);
}

void launch_joseph(hipStream_t stream, int n, const DevParams& prm, const double* cov_in, const double* a6,
                   const void* out, double* cov_out) {
  hipLaunchKernelGGL(ieskf_joseph_kernel, dim3(n), dim3(kJosephBlock), 0, stream, prm, cov_in, a6, (const OutRec*)out,
                     cov_out);
}

void launch_pass(hipStream_t stream, int n, const DevParams& prm, const ScanDesc* descs, const float4* arena,
                 const double* lin_state, const double* filt_state, int iter, int4* idx_store, lins_corr* dump,
                 double* sums_out, int* counts_out, float4* binned) {
  if (prm.search == SEARCH_BINNED)
    hipLaunchKernelGGL(ieskf_pass_kernel<SEARCH_BINNED>, dim3(n), dim3(kBlock), 0, stream, prm, descs, arena,
                       lin_state, filt_state, iter, idx_store, dump, sums_out, counts_out, binned);
  else
    hipLaunchKernelGGL(ieskf_pass_kernel<SEARCH_BRUTE>, dim3(n), dim3(kBlock), 0, stream, prm, descs, arena,
                       lin_state, filt_state, iter, idx_store, dump, sums_out, counts_out, binned);
}

// ---------------------------------------------------------------------------
// updatePointCloud's re-projection (SE:1116-1139): transformToEnd (SE:1083-1101) per point.
// One thread per point, blockIdx.y = cloud; pure streaming (16 B in, 16 or 32 B out per point)
// with ~300 f64 flops of de-skew per point in between.
// ---------------------------------------------------------------------------
struct ReprojectJob {
  long long off;  // first point of the cloud in the in / out arenas
  int n, has_yzx;
  double t[3], q[4];
  double inv_period;
};

constexpr int kToEndPts = 8;
__global__ __launch_bounds__(256) void transform_to_end_kernel(const ReprojectJob* __restrict__ jobs,
                                                               const float4* __restrict__ in,
                                                               float4* __restrict__ out_xyz,
                                                               float4* __restrict__ out_yzx) {
  const ReprojectJob jb = jobs[blockIdx.y];
  const ToEnd te = make_to_end(V3{jb.t[0], jb.t[1], jb.t[2]}, Q4{jb.q[0], jb.q[1], jb.q[2], jb.q[3]}, jb.inv_period);
  // kToEndPts points per thread, their reads in flight together: the per-cloud constants (a libm quat2axis, a matrix)
  // cost as much as a dozen points, and with one point per thread (rounds 1-3) they were the kernel
  for (int i0 = blockIdx.x * blockDim.x * kToEndPts + threadIdx.x; i0 < jb.n; i0 += gridDim.x * blockDim.x * kToEndPts) {
    float4 pi[kToEndPts];
#pragma unroll
    for (int u = 0; u < kToEndPts; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      pi[u] = in[jb.off + (i < jb.n ? i : jb.n - 1)];
    }
#pragma unroll
    for (int u = 0; u < kToEndPts; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      if (i >= jb.n) break;
      const V3 p2 = to_end_point(te, (double)pi[u].x, (double)pi[u].y, (double)pi[u].z, pi[u].w);
      const float x = (float)p2.x, y = (float)p2.y, z = (float)p2.z;
      out_xyz[jb.off + i] = make_float4(x, y, z, pi[u].w);
      if (jb.has_yzx) out_yzx[jb.off + i] = make_float4(y, z, x, pi[u].w);
    }
  }
}

void launch_transform_to_end(hipStream_t stream, int n_jobs, int max_n, const void* jobs, const float4* in,
                             float4* out_xyz, float4* out_yzx) {
  int gx = (max_n + 256 * kToEndPts - 1) / (256 * kToEndPts);
  if (gx < 1) gx = 1;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(transform_to_end_kernel, dim3(gx, n_jobs), dim3(256), 0, stream, (const ReprojectJob*)jobs, in,
                     out_xyz, out_yzx);
}
size_t reproject_job_size() { return sizeof(ReprojectJob); }

// updatePointCloud's re-projection for device-resident streams (lins_streams_step): the pose is read
// from the update's output states on the device, the clouds are rewritten in place (SE:1122-1131)
struct StreamCloud {
  long long off;  // first point in the stream arena
  int n, stream;  // points, index of the state to use
};
__global__ __launch_bounds__(256) void reproject_in_place_kernel(const StreamCloud* __restrict__ jobs,
                                                                 const double* __restrict__ states, float4* __restrict__ arena,
                                                                 double inv_period) {
  const StreamCloud jb = jobs[blockIdx.y];
  const double* st = states + (size_t)jb.stream * 19;
  const ToEnd te = make_to_end(V3{st[0], st[1], st[2]}, Q4{st[6], st[7], st[8], st[9]}, inv_period);
  for (int i0 = blockIdx.x * blockDim.x * kToEndPts + threadIdx.x; i0 < jb.n; i0 += gridDim.x * blockDim.x * kToEndPts) {
    float4 pi[kToEndPts];
#pragma unroll
    for (int u = 0; u < kToEndPts; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      pi[u] = arena[jb.off + (i < jb.n ? i : jb.n - 1)];
    }
#pragma unroll
    for (int u = 0; u < kToEndPts; ++u) {
      const int i = i0 + u * (int)blockDim.x;
      if (i >= jb.n) break;
      const V3 p2 = to_end_point(te, (double)pi[u].x, (double)pi[u].y, (double)pi[u].z, pi[u].w);
      arena[jb.off + i] = make_float4((float)p2.x, (float)p2.y, (float)p2.z, pi[u].w);
    }
  }
}
void launch_reproject_in_place(hipStream_t stream, int n_jobs, int max_n, const void* jobs, const double* states,
                               float4* arena, double inv_period) {
  int gx = (max_n + 256 * kToEndPts - 1) / (256 * kToEndPts);
  gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
  hipLaunchKernelGGL(reproject_in_place_kernel, dim3(gx, n_jobs), dim3(256), 0, stream, (const StreamCloud*)jobs, states,
                     arena, inv_period);
}
size_t stream_cloud_size() { return sizeof(StreamCloud); }

// device-copy ceiling probe (lins_debug_stream_copy): what a pure streaming kernel reaches on this box
__global__ __launch_bounds__(256) void stream_copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
void launch_stream_copy(hipStream_t stream, const float4* in, float4* out, size_t n) {
  hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 16), dim3(256), 0, stream, in, out, n);
}

size_t out_rec_size() { return sizeof(OutRec); }

}  // namespace lins
