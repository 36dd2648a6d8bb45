// ieskf_binned.h — exact, pruned correspondence search over a (ring x azimuth-column)
// grid of the target cloud: a coarse range image of the previous scan's features.
//
// The reference finds the closest target with a kd-tree (SE:847, 973) and the
// second/third points by walking neighbouring indices gated by ring id
// (SE:859-910, 983-1024).  Here the targets of one scan are counting-sorted once
// per update into cells (ring r = int(intensity), column a = floor of the azimuth),
// and every query visits only the cells whose lower distance bound does not exceed
// its current best:
//   ring bound    all points of ring r lie between two elevation cones measured
//                 from the data (min/max of atan2(z, rho) over the ring);
//   column bound  all points of the not-yet-visited columns on one side lie beyond
//                 the vertical half-plane through the last visited column edge.
// Both bounds are conservative (angular slack for the f32 atan2/sincos/rounding
// errors, see kSlack), ties are resolved on explicit (distance, index) /
// (distance, visit-rank) keys, so the result is the SAME triplet the exhaustive
// search + literal walk returns — independent of the order points sit in a cell.
//
// Requirements: target cloud ring-sorted with ring ids < kRingsBinned (VLP-16:
// LINE_NUM = 16, yaml:9).  Otherwise the scan falls back to the exact brute path.
#pragma once

#include "ieskf_device.h"

namespace lins {

constexpr int kRingsBinned = 16;
constexpr float kPiF = 3.14159265358979f;
constexpr float kSlack = 1.2e-5f;  // rad: >5x the worst-case angular error of a cell assignment

struct CloudBins {
  const unsigned* cell_end;  // LDS [kRingsBinned * naz]: exclusive end of each cell in `binned`
  const int* ring_start;     // LDS [kRingsBinned + 1]
  const float* el;           // LDS [kRingsBinned][4]: (cos lo, sin lo, cos hi, sin hi)
  const float2* az_edge;     // LDS [kAzSurf + 1]: unit rays of the column edges (finest grid)
  const float4* binned;      // global: (x, y, z, bits(original index)), cell-major
  int naz, az_stride, n;
};

__device__ __forceinline__ int az_bin(float x, float y, int naz) {
  int a = (int)((atan2f(y, x) + kPiF) * ((float)naz * (0.5f / kPiF)));
  return a < 0 ? 0 : (a >= naz ? naz - 1 : a);
}

__device__ __forceinline__ int ordered_int(float f) {
  int b = __float_as_int(f);
  return b >= 0 ? b : b ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float ordered_float(int b) { return __int_as_float(b >= 0 ? b : b ^ 0x7FFFFFFF); }

// squared lower bound of the distance from q (2-D) to everything beyond the ray u,
// on the side away from q; shrunk by the angular slack so that it stays a bound
// under every rounding on the way.
__device__ __forceinline__ float ray_bound_sq(float qx, float qy, float qn, float2 u) {
  float t = qx * u.x + qy * u.y;
  float perp = fabsf(qx * u.y - qy * u.x);
  float lb = (t > 0.f ? perp : qn) - kSlack * qn - 1e-7f;
  lb = lb > 0.f ? lb : 0.f;
  return lb * lb * (1.f - 1e-6f);
}

// ring r's elevation wedge vs the query (rho_q, z_q)
__device__ __forceinline__ float ring_bound_sq(const float* el, float rho, float z, float qn3) {
  float lo_c = el[0], lo_s = el[1], hi_c = el[2], hi_s = el[3];
  float above_hi = hi_c * z - hi_s * rho;  // > 0: query is above the upper cone
  float below_lo = lo_s * rho - lo_c * z;  // > 0: query is below the lower cone
  if (above_hi > 0.f) return ray_bound_sq(rho, z, qn3, make_float2(hi_c, hi_s));
  if (below_lo > 0.f) return ray_bound_sq(rho, z, qn3, make_float2(lo_c, lo_s));
  return 0.f;
}

// All points of binned[s, e): four independent 16-byte loads in flight per step (the
// gather is latency-bound; a column window is one contiguous span of the cell-major array).
template <class F>
__device__ __forceinline__ void for_span_points(const float4* __restrict__ b, unsigned s, unsigned e, F f) {
  for (unsigned p = s; p < e; p += 4) {
    const unsigned last = e - 1;
    float4 t0 = b[p];
    float4 t1 = b[p + 1 < last ? p + 1 : last];
    float4 t2 = b[p + 2 < last ? p + 2 : last];
    float4 t3 = b[p + 3 < last ? p + 3 : last];
    f(t0);
    if (p + 1 < e) f(t1);
    if (p + 2 < e) f(t2);
    if (p + 3 < e) f(t3);
  }
}

// columns [lo, hi] of ring-row `base` (lo, hi may run past the wrap on either side)
template <class F>
__device__ __forceinline__ void for_columns(const CloudBins& cb, int base, int lo, int hi, F f) {
  const int naz = cb.naz;
  if (lo > hi) return;
  auto span = [&](int a, int b) {  // 0 <= a <= b < naz
    int c0 = base + a, c1 = base + b;
    for_span_points(cb.binned, c0 ? cb.cell_end[c0 - 1] : 0u, cb.cell_end[c1], f);
  };
  // normalise: lo into [0, naz), at most naz columns, at most two contiguous pieces
  int len = hi - lo;
  if (len >= naz - 1) {
    span(0, naz - 1);
    return;
  }
  lo %= naz;
  if (lo < 0) lo += naz;
  hi = lo + len;
  if (hi < naz) {
    span(lo, hi);
  } else {
    span(lo, naz - 1);
    span(0, hi - naz);
  }
}

// Visit ring r: first the seed window a0-1..a0+1, then — with the bound tightened by
// the seed — the columns to the right and to the left whose edge bound does not exceed
// `bound()`.  The window is fixed before each span is read (a superset of what a
// cell-by-cell adaptive walk would touch, never a subset), so a ring costs at most
// three dependent gather rounds instead of one per cell.
template <class Bound, class F>
__device__ __forceinline__ void scan_ring(const CloudBins& cb, int r, int a0, float qx, float qy, float qn,
                                          Bound bound, F f) {
  const int naz = cb.naz, base = r * naz, half = naz / 2;
  for_columns(cb, base, a0 - 1, a0 + 1, f);
  // column a0+k is needed iff the bound of its lower edge (a0+k) is within reach
  int kr = 1;
  while (kr < half) {
    int e = a0 + kr + 1;
    if (ray_bound_sq(qx, qy, qn, cb.az_edge[(e <= naz ? e : e - naz) * cb.az_stride]) > bound()) break;
    ++kr;
  }
  // column a0-k is needed iff the bound of its upper edge (a0-k+1) is within reach
  int kl = 1;
  while (kl < half - 1) {
    int e = a0 - kl;
    if (ray_bound_sq(qx, qy, qn, cb.az_edge[(e >= 0 ? e : e + naz) * cb.az_stride]) > bound()) break;
    ++kl;
  }
  for_columns(cb, base, a0 + 2, a0 + kr, f);
  for_columns(cb, base, a0 - kl, a0 - 2, f);
}

// ---- pass 1: exact nearest neighbour (lowest index wins ties) -----------------------
__device__ __forceinline__ void nn_binned(const CloudBins& cb, float sx, float sy, float sz, float thr, int rq,
                                          int& best_j, float& best_d) {
  best_j = -1;
  best_d = thr;  // anything >= thr is "no correspondence" (SE:851): prune at thr
  const float rho = sqrtf(sx * sx + sy * sy);
  const float qn3 = sqrtf(rho * rho + sz * sz);
  const int a0 = az_bin(sx, sy, cb.naz);
  rq = rq < 0 ? 0 : (rq >= kRingsBinned ? kRingsBinned - 1 : rq);
  for (int i = 0; i < 2 * kRingsBinned; ++i) {  // rq, rq+1, rq-1, rq+2, ...
    int r = rq + ((i & 1) ? (i + 1) / 2 : -(i / 2));
    if (r < 0 || r >= kRingsBinned) continue;
    if (cb.ring_start[r + 1] == cb.ring_start[r]) continue;
    if (ring_bound_sq(cb.el + 4 * r, rho, sz, qn3) > best_d) continue;
    scan_ring(
        cb, r, a0, sx, sy, rho, [&]() { return best_d; },
        [&](const float4& t) {
          int j = __float_as_int(t.w);
          float d = sqdist3(t.x, t.y, t.z, sx, sy, sz);
          if (d < best_d || (d == best_d && j < best_j)) best_d = d, best_j = j;
        });
  }
}

// ---- pass 2: the index walk as a masked, rank-keyed argmin ----------------------------
constexpr int kBackRank = 0x40000000;

// Surf (SE:859-910).  Sorted cloud => forward candidates are (j1, f_hi), backward
// [b_lo, j1); class 2 = ring rho, class 3 = the other rings of the window.
__device__ __forceinline__ void walk_surf_binned(const CloudBins& cb, int nq, float thr, int j1, int rho, float sx,
                                                 float sy, float sz, int& m2, int& m3) {
  const int fend = nq < cb.n ? nq : cb.n;
  const int r_hi = rho + 3 < kRingsBinned ? rho + 3 : kRingsBinned;
  const int r_lo = rho - 2 > 0 ? rho - 2 : 0;
  const int f_hi = fend < cb.ring_start[r_hi] ? fend : cb.ring_start[r_hi];
  const int b_lo = cb.ring_start[r_lo];
  float d2 = thr, d3 = thr;
  int k2 = 0, k3 = 0;
  m2 = m3 = -1;
  const float rho_q = sqrtf(sx * sx + sy * sy);
  const float qn3 = sqrtf(rho_q * rho_q + sz * sz);
  const int a0 = az_bin(sx, sy, cb.naz);
  for (int i = 0; i < 5; ++i) {  // rho, rho-1, rho+1, rho-2, rho+2: near rings tighten the bounds first
    const int r = rho + ((i & 1) ? -((i + 1) / 2) : i / 2);
    if (r < r_lo || r >= r_hi) continue;
    const int rs = cb.ring_start[r], re = cb.ring_start[r + 1];
    const bool fwd = (j1 + 1 > rs ? j1 + 1 : rs) < (f_hi < re ? f_hi : re);
    const bool bwd = (b_lo > rs ? b_lo : rs) < (j1 < re ? j1 : re);
    if (!fwd && !bwd) continue;
    const bool c2 = (r == rho);
    if (ring_bound_sq(cb.el + 4 * r, rho_q, sz, qn3) > (c2 ? d2 : d3)) continue;
    scan_ring(
        cb, r, a0, sx, sy, rho_q, [&]() { return c2 ? d2 : d3; },
        [&](const float4& t) {
          int j = __float_as_int(t.w);
          bool ok = (j > j1 && j < f_hi) || (j < j1 && j >= b_lo);
          if (!ok) return;
          int rank = j > j1 ? j - j1 : kBackRank + (j1 - j);
          float d = sqdist3(t.x, t.y, t.z, sx, sy, sz);
          if (c2) {
            if (d < d2 || (d == d2 && m2 >= 0 && rank < k2)) d2 = d, m2 = j, k2 = rank;
          } else {
            if (d < d3 || (d == d3 && m3 >= 0 && rank < k3)) d3 = d, m3 = j, k3 = rank;
          }
        });
  }
}

// Corner (SE:983-1024): second point on a different ring.
__device__ __forceinline__ void walk_corner_binned(const CloudBins& cb, int nq, float thr, int j1, int rho, float sx,
                                                   float sy, float sz, int& m2) {
  const int fend = nq < cb.n ? nq : cb.n;
  const int r_hi = rho + 3 < kRingsBinned ? rho + 3 : kRingsBinned;
  const int r_lo = rho - 2 > 0 ? rho - 2 : 0;
  const int f_hi = fend < cb.ring_start[r_hi] ? fend : cb.ring_start[r_hi];
  const int b_lo = cb.ring_start[r_lo];
  float d2 = thr;
  int k2 = 0;
  m2 = -1;
  const float rho_q = sqrtf(sx * sx + sy * sy);
  const float qn3 = sqrtf(rho_q * rho_q + sz * sz);
  const int a0 = az_bin(sx, sy, cb.naz);
  for (int i = 1; i < 5; ++i) {  // rho-1, rho+1, rho-2, rho+2
    const int r = rho + ((i & 1) ? -((i + 1) / 2) : i / 2);
    if (r < r_lo || r >= r_hi) continue;
    const int rs = cb.ring_start[r], re = cb.ring_start[r + 1];
    const bool any = r > rho ? ((j1 + 1 > rs ? j1 + 1 : rs) < (f_hi < re ? f_hi : re))
                             : ((b_lo > rs ? b_lo : rs) < (j1 < re ? j1 : re));
    if (!any) continue;
    if (ring_bound_sq(cb.el + 4 * r, rho_q, sz, qn3) > d2) continue;
    scan_ring(
        cb, r, a0, sx, sy, rho_q, [&]() { return d2; },
        [&](const float4& t) {
          int j = __float_as_int(t.w);
          bool ok = (j > j1 && j < f_hi) || (j < j1 && j >= b_lo);
          if (!ok) return;
          int rank = j > j1 ? j - j1 : kBackRank + (j1 - j);
          float d = sqdist3(t.x, t.y, t.z, sx, sy, sz);
          if (d < d2 || (d == d2 && m2 >= 0 && rank < k2)) d2 = d, m2 = j, k2 = rank;
        });
  }
}

// ---------------------------------------------------------------------------
// grid build: one workgroup, once per scan per update
// ---------------------------------------------------------------------------
struct BinStorage {  // LDS
  unsigned cell_surf[kRingsBinned * kAzSurf];
  unsigned cell_corner[kRingsBinned * kAzCorner];
  int ring_start[2][kRingsBinned + 1];
  float el[2][kRingsBinned][4];
  int el_bits[2][kRingsBinned][2];
  float2 az_edge[kAzSurf + 1];
  int scan_tmp[8];
};

// exclusive prefix sum of one value per thread over the 256-thread block
__device__ __forceinline__ int block_exclusive_scan(int v, int tid, int* tmp /*>=5 ints*/) {
  int lane = tid & 63, wave = tid >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int n = __shfl_up(incl, o, 64);
    if (lane >= o) incl += n;
  }
  if (lane == 63) tmp[wave] = incl;
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += tmp[w];
  __syncthreads();
  return off + incl - v;
}

__device__ __forceinline__ void build_cloud_bins(const float4* __restrict__ tg, int n, int naz, unsigned* cell,
                                                 int* ring_start, int (*el_bits)[2], float (*el)[4],
                                                 float4* __restrict__ binned, int tid, int* tmp) {
  const int ncell = kRingsBinned * naz;
  for (int c = tid; c < ncell; c += kBlock) cell[c] = 0;
  if (tid < kRingsBinned) el_bits[tid][0] = 0x7FFFFFFF, el_bits[tid][1] = (int)0x80000000;
  __syncthreads();
  for (int j = tid; j < n; j += kBlock) {
    float4 p = tg[j];
    int r = ring_of(p.w);
    int a = az_bin(p.x, p.y, naz);
    atomicAdd(&cell[r * naz + a], 1u);
    int eb = ordered_int(atan2f(p.z, sqrtf(p.x * p.x + p.y * p.y)));
    atomicMin(&el_bits[r][0], eb);
    atomicMax(&el_bits[r][1], eb);
  }
  __syncthreads();
  // exclusive scan over the cells: thread t owns cells [t*per, (t+1)*per)
  const int per = (ncell + kBlock - 1) / kBlock;
  const int c_lo = tid * per < ncell ? tid * per : ncell;
  const int c_hi = c_lo + per < ncell ? c_lo + per : ncell;
  int local = 0;
  for (int c = c_lo; c < c_hi; ++c) local += (int)cell[c];
  int run = block_exclusive_scan(local, tid, tmp);
  for (int c = c_lo; c < c_hi; ++c) {
    int cnt = (int)cell[c];
    cell[c] = (unsigned)run;  // start offset, used as the scatter cursor
    run += cnt;
  }
  __syncthreads();
  for (int j = tid; j < n; j += kBlock) {
    float4 p = tg[j];
    int r = ring_of(p.w);
    int a = az_bin(p.x, p.y, naz);
    unsigned pos = atomicAdd(&cell[r * naz + a], 1u);  // order inside a cell is irrelevant (keyed ties)
    binned[pos] = make_float4(p.x, p.y, p.z, __int_as_float(j));
  }
  __syncthreads();  // cursors now hold the exclusive END of every cell
  if (tid <= kRingsBinned) ring_start[tid] = tid == 0 ? 0 : (int)cell[tid * naz - 1];
  if (tid < kRingsBinned) {
    float lo = ordered_float(el_bits[tid][0]) - kSlack, hi = ordered_float(el_bits[tid][1]) + kSlack;
    el[tid][0] = cosf(lo), el[tid][1] = sinf(lo), el[tid][2] = cosf(hi), el[tid][3] = sinf(hi);
  }
  __syncthreads();
}

__device__ __forceinline__ void build_az_edges(float2* az_edge, int tid) {
  for (int a = tid; a <= kAzSurf; a += kBlock) {
    float th = -kPiF + (float)a * (2.f * kPiF / (float)kAzSurf);
    az_edge[a] = make_float2(cosf(th), sinf(th));
  }
}

}  // namespace lins
