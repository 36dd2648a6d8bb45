// ieskf_kernels.hip — CDNA4 (gfx950) kernels of the IESKF update path.
//
// One workgroup (4 wave64) owns one scan pair for the WHOLE iterated update
// (persistent over <= NUM_ITER iterations, no host round trips, SE:475-583):
//
//   wave 0        per-iteration constants  (phi, R^T, Rinvleft(-phi), x_f (-) x_lin)
//   all lanes     one query feature each: de-skew (f64) -> exact NN + index walk
//                 (f32) -> plane/line residual + Jacobian (f64 -> f32) -> row
//                 (c, u = p x R^T c, r) into an LDS slot   [A2, A3, A4, A5]
//   224 lanes     28 f64 sums of the 7x7 outer products, fixed-shape two-stage
//                 tree (8 strided groups -> ordered fold) => bit-reproducible
//   wave 0        6x6 pivoted solve of (sigma^2 I + A_SS P_SS), dx, NaN /
//                 divergence / convergence tests, boxPlus                   [A6, A7]
//   all lanes     after the loop: Joseph covariance update, 18x18 in LDS.
//
// No dense contraction anywhere => MFMA unused; the path is HBM/VALU work.
// Inputs are read from HBM once per scan (targets stay L2-resident across
// iterations); the only per-iteration global traffic is the target gather.

#include <hip/hip_runtime.h>

#include <type_traits>

#include "ieskf_binned.h"
#include "ieskf_device.h"

namespace lins {

struct OutRec {
  double residual_norm, update_norm;
  int iters, converged, diverged, m_surf, m_corner, pad[3];
};

__constant__ unsigned char kPairA[28] = {0, 0, 0, 1, 1, 2, 0, 0, 0, 1, 1, 1, 2, 2,
                                         2, 3, 3, 3, 4, 4, 5, 0, 1, 2, 3, 4, 5, 6};
__constant__ unsigned char kPairB[28] = {0, 1, 2, 1, 2, 2, 3, 4, 5, 3, 4, 5, 3, 4,
                                         5, 3, 4, 5, 4, 5, 5, 6, 6, 6, 6, 6, 6, 6};

// ---------------------------------------------------------------------------
// one query feature -> (indices, accepted, coeff)
// ---------------------------------------------------------------------------
struct NoBins {};

// LDS views of the two target grids of the scan this workgroup owns
struct ScanBins {
  CloudBins surf, corner;
};

template <int SEARCH>
__device__ __forceinline__ void process_surf(const DevParams& prm, const ScanDesc& sd,
                                             const float4* __restrict__ arena, const ScanBins* sb, const V3& phi,
                                             const V3& t, int iter, bool do_search, int i, const float4& q,
                                             QueryOut& o) {
  const float4* tg = arena + sd.off_surf_t;
  transform_to_start(prm, phi, t, q, o.sel[0], o.sel[1], o.sel[2]);
  o.accepted = 0;
  o.c[0] = o.c[1] = o.c[2] = o.c[3] = 0.f;
  if (do_search) {
    int j1;
    float d1;
    o.j1 = o.j2 = o.j3 = -1;
    if (SEARCH == SEARCH_BINNED && sd.surf_sorted) {
      nn_binned(sb->surf, o.sel[0], o.sel[1], o.sel[2], prm.nearest_f, ring_of(q.w), j1, d1);
      if (j1 >= 0 && (double)d1 < prm.nearest) {
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
                                               const V3& t, int iter, bool do_search, int i, const float4& q,
                                               QueryOut& o) {
  const float4* tg = arena + sd.off_corner_t;
  transform_to_start(prm, phi, t, q, o.sel[0], o.sel[1], o.sel[2]);
  o.accepted = 0;
  o.c[0] = o.c[1] = o.c[2] = o.c[3] = 0.f;
  if (do_search) {
    int j1;
    float d1;
    o.j1 = o.j2 = -1;
    o.j3 = -1;
    if (SEARCH == SEARCH_BINNED && sd.corner_sorted) {
      nn_binned(sb->corner, o.sel[0], o.sel[1], o.sel[2], prm.nearest_f, ring_of(q.w), j1, d1);
      if (j1 >= 0 && (double)d1 < prm.nearest) {
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
    process_surf<SEARCH>(prm, sd, arena, sb, phi, t, iter, do_search, i, q, o);
  else
    process_corner<SEARCH>(prm, sd, arena, sb, phi, t, iter, do_search, i, q, o);
  if (do_search && prm.icp_freq > 1) idx_store[sd.slot_base + slot] = make_int4(o.j1, o.j2, o.j3, 0);
}

// ---------------------------------------------------------------------------
// LDS layout of the persistent kernel
// ---------------------------------------------------------------------------
struct Shared {
  IterConst ic;
  double filt[19];
  double P[324];
  double rows[kRowsCap * 7];  // later reused for the Joseph update (IKH, T)
  double partial[kRedGroups * 28];
  double sums[28];
  double A6[36];
  double Y[36], Zt[36];
  double dx[18];
  double res_prev, res_last, upd_norm;
  int m_surf, m_corner;
  int iter, conv, div, pad;
};

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

// rows -> 28 sums.  Group g folds rows g, g+8, ... in order; then the 8 group
// partials are folded in order.  Same tree every run => deterministic.
__device__ __forceinline__ void accumulate_rows(const double* rows, int nrows, int tid, double& acc) {
  int g = tid >> 5, k = tid & 31;
  if (k < 28) {
    int a = kPairA[k], b = kPairB[k];
    for (int r = g; r < nrows; r += kRedGroups) acc += rows[r * 7 + a] * rows[r * 7 + b];
  }
}

template <int SEARCH>
__device__ __forceinline__ void correspondence_round(const DevParams& prm, const ScanDesc& sd,
                                                     const float4* __restrict__ arena, const ScanBins* sb,
                                                     const IterConst& ic,
                                                     int iter, bool do_search, int base, int total,
                                                     int4* __restrict__ idx_store, double* rows, int tid,
                                                     int& ms, int& mc, lins_corr* __restrict__ dump) {
  V3 phi = ic.phi;
  V3 t{ic.lin[0], ic.lin[1], ic.lin[2]};
#pragma unroll 1
  for (int h = 0; h < kRowsCap / kBlock; ++h) {
    int local = h * kBlock + tid;
    int slot = base + local;
    double row[7] = {0, 0, 0, 0, 0, 0, 0};
    if (slot < total) {
      QueryOut o;
      float4 q;
      bool is_surf;
      process_slot<SEARCH>(prm, sd, arena, sb, phi, t, iter, do_search, slot, idx_store, o, q, is_surf);
      if (o.accepted) {
        V3 c{(double)o.c[0], (double)o.c[1], (double)o.c[2]};
        V3 w = mvec(ic.Rt, c);
        V3 u = cross(V3{(double)q.x, (double)q.y, (double)q.z}, w);
        row[0] = c.x, row[1] = c.y, row[2] = c.z;
        row[3] = u.x, row[4] = u.y, row[5] = u.z;
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
    if (rows) {
#pragma unroll
      for (int k = 0; k < 7; ++k) rows[local * 7 + k] = row[k];
    }
  }
}

// ---------------------------------------------------------------------------
// persistent IESKF kernel: grid = scans, block = 256
// ---------------------------------------------------------------------------
template <int SEARCH>
__global__ __launch_bounds__(kBlock) void ieskf_persistent_kernel(
    DevParams prm, const ScanDesc* __restrict__ descs, const float4* __restrict__ arena,
    const double* __restrict__ state_in, const double* __restrict__ cov_in, double* __restrict__ state_out,
    double* __restrict__ cov_out, OutRec* __restrict__ out, int4* __restrict__ idx_store,
    lins_pose_record* __restrict__ poses, int scan_id_base, float4* __restrict__ binned) {
  __shared__ Shared sh;
  __shared__ std::conditional_t<SEARCH == SEARCH_BINNED, BinStorage, NoBins> bstore;
  __shared__ ScanBins sbins;
  const int tid = threadIdx.x;
  const int scan = blockIdx.x;
  const ScanDesc sd = descs[scan];
  const int total = sd.n_surf_q + sd.n_corner_q;

  for (int k = tid; k < 324; k += kBlock) sh.P[k] = cov_in[(size_t)scan * 324 + k];
  if (tid < 19) {
    double v = state_in[(size_t)scan * 19 + tid];
    sh.filt[tid] = v;
    sh.ic.lin[tid] = v;
  }
  if (tid == 0) {
    sh.res_prev = 1e6, sh.res_last = 0, sh.upd_norm = 0;
    sh.iter = 0, sh.conv = 0, sh.div = 0, sh.m_surf = 0, sh.m_corner = 0;
  }
  __syncthreads();
  setup_bins<SEARCH>(sd, arena, binned, (BinStorage*)&bstore, &sbins, tid);

  for (;;) {
    const int iter = sh.iter;
    if (iter >= prm.num_iter || sh.conv || sh.div) break;
    if (tid < 64) {  // wave 0, lane-redundant scalar work
      IterConst ic;
      for (int k = 0; k < 19; ++k) ic.lin[k] = sh.ic.lin[k];
      double filt[19];
      for (int k = 0; k < 19; ++k) filt[k] = sh.filt[k];
      make_iter_const(filt, ic);
      if (tid == 0) {
        sh.ic.phi = ic.phi;
        sh.ic.Rt = ic.Rt;
        sh.ic.G = ic.G;
        for (int k = 0; k < 18; ++k) sh.ic.d[k] = ic.d[k];
        sh.m_surf = 0, sh.m_corner = 0;
      }
    }
    __syncthreads();

    const bool do_search = (iter % prm.icp_freq) == 0;
    double acc = 0;
    int ms = 0, mc = 0;
    for (int base = 0; base < total; base += kRowsCap) {
      correspondence_round<SEARCH>(prm, sd, arena, &sbins, sh.ic, iter, do_search, base, total, idx_store,
                                   sh.rows, tid, ms, mc, nullptr);
      __syncthreads();
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

    if (tid < 64) {  // wave 0: solve + state update (SE:542-580), lane-redundant
      double sums[28];
      for (int k = 0; k < 28; ++k) sums[k] = sh.sums[k];
      double A6[36], g6[6], d[18], dx[18];
      M3 G = sh.ic.G;
      for (int k = 0; k < 18; ++k) d[k] = sh.ic.d[k];
      sums_to_normal(sums, G, A6, g6);
      update_reduced(prm.r2, sh.P, A6, g6, d, dx);
      double rn = sqrt(sums[27]);
      bool has_nan = false;
      for (int k = 0; k < 18; ++k)
        if (isnan(dx[k])) has_nan = true;
      int div = 0, conv = 0;
      double lin[19];
      for (int k = 0; k < 19; ++k) lin[k] = sh.ic.lin[k];
      double un = sh.upd_norm, res_prev = sh.res_prev;
      if (has_nan) {
        div = 2;
      } else if (rn > res_prev * 10) {
        div = 1;
      } else {
        box_plus_inplace(lin, dx);
        un = 0;
        for (int k = 0; k < 18; ++k) un += dx[k] * dx[k];
        un = sqrt(un);
        if (un <= 1e-2 && !prm.fixed_iters) conv = 1;
        res_prev = rn;
      }
      if (tid == 0) {
        for (int k = 0; k < 36; ++k) sh.A6[k] = A6[k];
        for (int k = 0; k < 19; ++k) sh.ic.lin[k] = lin[k];
        sh.res_last = rn;
        sh.res_prev = res_prev;
        sh.upd_norm = un;
        sh.conv = conv, sh.div = div;
        sh.iter = iter + 1;
      }
    }
    __syncthreads();
  }

  // ---- after the loop (SE:585-598) ----------------------------------------
  const int div = sh.div;
  if (div) {
    // the caller runs the ICP fallback; hand back the un-updated filter state / Pk_
    for (int k = tid; k < 324; k += kBlock) cov_out[(size_t)scan * 324 + k] = sh.P[k];
    if (tid < 19) state_out[(size_t)scan * 19 + tid] = sh.filt[tid];
  } else {
    // Joseph update with the LAST executed iteration's A (SE:594-598), reduced form:
    //   KH = P[:,S] Y E_S^T,  Y = N^-1 A ;  K R K^T = sigma^2 P[:,S] (Y N^-T) P[:,S]^T
    if (tid < 64) {
      double n[6][12];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          double t = 0;
#pragma unroll
          for (int k = 0; k < 6; ++k) t += sh.A6[i * 6 + k] * sh.P[sidx(k) * 18 + sidx(j)];
          n[i][j] = t + (i == j ? prm.r2 : 0.0);
          n[i][6 + j] = sh.A6[i * 6 + j];
        }
      double nn[6][6];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) nn[i][j] = n[i][j];
      lu_solve6<6>(n);  // Y = N^-1 A
      double z[6][12];
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          z[i][j] = nn[i][j];
          z[i][6 + j] = n[j][6 + i];  // Y^T
        }
      lu_solve6<6>(z);  // Zt = N^-1 Y^T  (Z = Y N^-T)
      if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = 0; j < 6; ++j) sh.Y[i * 6 + j] = n[i][6 + j], sh.Zt[i * 6 + j] = z[i][6 + j];
      }
    }
    __syncthreads();
    double* IKH = sh.rows;         // 324
    double* T = sh.rows + 324;     // 324
    double* PSZ = sh.rows + 648;   // 18 x 6
    for (int e = tid; e < 324; e += kBlock) {
      int i = e / 18, j = e % 18;
      double v = (i == j) ? 1.0 : 0.0;
      // column j of KH is non-zero only for j in S
      int kj = (j < 3) ? j : ((j >= 6 && j < 9) ? j - 3 : -1);
      if (kj >= 0) {
        double s = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) s += sh.P[i * 18 + sidx(k)] * sh.Y[k * 6 + kj];
        v -= s;
      }
      IKH[e] = v;
    }
    for (int e = tid; e < 108; e += kBlock) {
      int i = e / 6, j = e % 6;
      double s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += sh.P[i * 18 + sidx(k)] * sh.Zt[j * 6 + k];
      PSZ[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < 324; e += kBlock) {
      int i = e / 18, j = e % 18;
      double s = 0;
      for (int k = 0; k < 18; ++k) s += IKH[i * 18 + k] * sh.P[k * 18 + j];
      T[e] = s;
    }
    __syncthreads();
    double* O = sh.rows + 756;  // 324
    for (int e = tid; e < 324; e += kBlock) {
      int i = e / 18, j = e % 18;
      double s = 0;
      for (int k = 0; k < 18; ++k) s += T[i * 18 + k] * IKH[j * 18 + k];
      double kk = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) kk += PSZ[i * 6 + k] * sh.P[j * 18 + sidx(k)];
      O[e] = s + prm.r2 * kk;
    }
    __syncthreads();
    for (int e = tid; e < 324; e += kBlock) {
      int i = e / 18, j = e % 18;
      cov_out[(size_t)scan * 324 + e] = 0.5 * (O[i * 18 + j] + O[j * 18 + i]);
    }
    if (tid < 19) state_out[(size_t)scan * 19 + tid] = sh.ic.lin[tid];
  }
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
// single pass kernels (BASELINE.json configs[1]: device correspondences, host solve)
// ---------------------------------------------------------------------------
// One correspondence + residual/Jacobian pass for a caller-supplied linearisation
// state; optionally dumps per-query records and/or the 28 reduced sums.
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
                       double* cov_out, void* out, int4* idx_store, lins_pose_record* poses, int scan_id_base,
                       float4* binned) {
  if (prm.search == SEARCH_BINNED)
    hipLaunchKernelGGL(ieskf_persistent_kernel<SEARCH_BINNED>, dim3(n), dim3(kBlock), 0, stream, prm, descs, arena,
                       state_in, cov_in, state_out, cov_out, (OutRec*)out, idx_store, poses, scan_id_base, binned);
  else
    hipLaunchKernelGGL(ieskf_persistent_kernel<SEARCH_BRUTE>, dim3(n), dim3(kBlock), 0, stream, prm, descs, arena,
                       state_in, cov_in, state_out, cov_out, (OutRec*)out, idx_store, poses, scan_id_base, binned);
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

size_t out_rec_size() { return sizeof(OutRec); }

}  // namespace lins
