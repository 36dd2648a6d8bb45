// ieskf_device.h — device-side building blocks of the IESKF update path.
//
// Behavioural contract, relative to /root/reference/lins/include/StateEstimator.hpp:
//   transformToStart                         1066-1080
//   findCorrespondingSurfFeatures            829-953   (search, walk, plane row)
//   findCorrespondingCornerFeatures          955-1063  (search, walk, line row)
//   H row / residual                         507-532
//   gain / increment / Joseph update         542-549, 594-598 (reduced 6x6 form,
//                                            SURVEY.md §8a A6 — never inverts P)
// Arithmetic rules that make the index sets and f32 rows bit-comparable with the
// reference's CPU path: distances in f32 as ((dx*dx+dy*dy)+dz*dz) with contraction
// off, geometry in f64 rounded to f32 at the same points, strict '<' first-seen
// tie rules restated as lexicographic (distance, visit-rank) minima.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/lins_ieskf.h"
#include "lins_math.h"

namespace lins {

constexpr int kBlock = 320;          // threads per workgroup (5 waves): one round covers a typical
                                     // VLP-16 scan's ~260 query features (caps: 192 + 144)
constexpr int kRowsCap = kBlock;     // LDS row slots per reduction round
constexpr int kRedGroups = kBlock / 32;  // partial-sum groups per reduction
constexpr int kAzSurf = 128;         // azimuth columns per ring, surf targets
constexpr int kAzCorner = 64;        // azimuth columns per ring, corner targets
constexpr int kMaxRing = LINS_MAX_RING;

enum { SEARCH_BRUTE = 0, SEARCH_BINNED = 1, SEARCH_LDS = 2, SEARCH_LDS3 = 3, SEARCH_MR = 4, SEARCH_AUTO = 5 };

struct ScanDesc {  // one IESKF problem in the device arena (offsets in points)
  int off_surf_q, n_surf_q;
  int off_corner_q, n_corner_q;
  int off_surf_t, n_surf_t;
  int off_corner_t, n_corner_t;
  int surf_sorted, corner_sorted;  // targets ring-sorted (host-validated)?
  int slot_base;                   // first per-query scratch slot of this scan
  int pad;
};

struct DevParams {
  int num_iter, icp_freq, fixed_iters, search;
  double r2;          // LIDAR_STD^2
  double lidar_scale;
  double inv_period;  // (double)(1.f / SCAN_PERIOD)
  double nearest;     // NEAREST_FEATURE_SEARCH_SQ_DIST
  float nearest_f;
  int pad;            // debug / profiling flags (LINS_DEBUG_SKIP)
  float margin_cold;  // certificate margins of the LDS search [m] (ieskf_lds.hip)
  float margin_warm;
  float reseed_drift;  // [m] a warm search whose query moved farther than this since its last search seeds like a cold one (ieskf_lds_lean.h)
  int pad2;
};

// The iterations an update is cut at: k x at for k = 1 .. max_cuts, none afterwards (the last part runs to the end).
// Shared by the kernel (where an item ends) and the host (how many items a launch has).
// (at == -1: max_cuts is a MASK of the iterations an update is cut at — bit i: a part ends before iteration i, i < 31 —
// for cuts that are not evenly spaced, e.g. 0x92 = after the cold iteration, then at 4 and 7)
__host__ __device__ inline int relay_next_cut(int iter, int at, int max_cuts) {
  if (at < 0) {
    const unsigned m = iter >= 30 ? 0u : ((unsigned)max_cuts >> (iter + 1)) << (iter + 1);
#if defined(__HIP_DEVICE_COMPILE__)
    return m ? __ffs((int)m) - 1 : 0x7FFFFFFF;
#else
    return m ? __builtin_ctz(m) : 0x7FFFFFFF;
#endif
  }
  const int k = iter / at + 1;
  return k <= max_cuts ? k * at : 0x7FFFFFFF;
}
__host__ __device__ inline int relay_max_parts(int num_iter, int at, int max_cuts) {  // parts an update of num_iter iterations can have
  int parts = 1;
  for (int c = relay_next_cut(0, at, max_cuts); c < num_iter; c = relay_next_cut(c, at, max_cuts)) ++parts;
  return parts;
}

// Several-part updates of the batch kernel (ieskf_lds_impl.h "relay" + "work items"): the update of every scan of a large batch
// is cut (relay_next_cut) into `parts` parts; the launch has one workgroup per (scan, part), which draws its item by ticket.
struct RelayArgs {
  int at = 0, cuts = 0, parts = 0, gen = 0;
  int cap = 0;            // scans the flag array holds
  int slots = 0;          // workgroups of the batch kernel resident at once on the device
  int spins = 1 << 21;    // polls (~1 us each) a part waits for its hand-over before it gives up (reported by lins_sync)
  double* hdr = nullptr;  // per scan: 64 doubles of loop state
  int* lane = nullptr;    // per scan: the carried state of every query lane (ieskf_lds_impl.h CarryWords)
  int* queue = nullptr;   // ticket counters + one flag per scan (ieskf_lds_impl.h kQ*)
  int* err = nullptr;     // per context: waits that ran out
};

struct IterConst {  // per-iteration constants, hoisted (the reference recomputes per point)
  double lin[19];   // linState_
  V3 phi;           // Quat2axis(linState_.qbn_)
  M3 Rt;            // R(q)^T
  M3 Gt;            // Rinvleft(-phi)^T
  double d[18];     // filterState (-) linState_
};

struct QueryOut {
  int j1, j2, j3, accepted;
  float c[4];
  float sel[3];
};

// ---------------------------------------------------------------------------
__device__ __forceinline__ float sqdist3(float tx, float ty, float tz, float sx, float sy, float sz) {
  // contraction is off: three roundings, ((dx dx + dy dy) + dz dz), as on the CPU.  (dx, dy) travel as one packed pair
  // — v_pk_add_f32 / v_pk_mul_f32 of gfx950, the same IEEE operations in the same order: -0.5 % on the batch kernel
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f d = v2f{tx, ty} - v2f{sx, sy};
  const v2f q = d * d;
  const float dz = tz - sz;
  return (q.x + q.y) + dz * dz;
}

__device__ __forceinline__ int ring_of(float intensity) { return (int)intensity; }

// transformToStart (SE:1066-1080) with phi hoisted.
// (`tab`: the series coefficients of lins_sinc_cos_table in memory — LDS in the LDS kernels — or nullptr for libm)
__device__ __forceinline__ void transform_to_start(const DevParams& prm, const V3& phi, const V3& t,
                                                   const float4& pi, float& ox, float& oy, float& oz,
                                                   const double* tab = nullptr) {
  float frac = pi.w - (float)(int)pi.w;
  double s = prm.inv_period * (double)frac;
  Q4 r = tab ? axis2quat_tab(s * phi, tab) : axis2quat(s * phi);
  V3 p1 = qrot(r, V3{(double)pi.x, (double)pi.y, (double)pi.z}) + s * t;
  ox = (float)p1.x, oy = (float)p1.y, oz = (float)p1.z;
}

// plane row (SE:917-951)
__device__ __forceinline__ void surf_row(const DevParams& prm, int iter, float sx, float sy, float sz,
                                         const float4& t1, const float4& t2, const float4& t3, QueryOut& o) {
  V3 p0{sx, sy, sz}, p1{t1.x, t1.y, t1.z}, p2{t2.x, t2.y, t2.z}, p3{t3.x, t3.y, t3.z};
  V3 m = cross(p1 - p2, p1 - p3);
  double r = dot(p0 - p1, m);
  double mn = norm(m);
  float res = (float)(r / mn);
  V3 jac = m / mn;
  float s = 1;
  if (iter >= prm.icp_freq) {
    float n2 = sx * sx + sy * sy + sz * sz;
    s = (float)(1 - 1.8 * (double)fabsf(res) / (double)sqrtf(sqrtf(n2)));
  }
  if (s > 0.1 && res != 0) {
    o.accepted = 1;
    o.c[0] = (float)((double)s * jac.x);
    o.c[1] = (float)((double)s * jac.y);
    o.c[2] = (float)((double)s * jac.z);
    o.c[3] = s * res;
  }
}

// line row (SE:1031-1061)
__device__ __forceinline__ void corner_row(const DevParams& prm, int iter, float sx, float sy, float sz,
                                           const float4& t1, const float4& t2, QueryOut& o) {
  V3 p0{sx, sy, sz}, p1{t1.x, t1.y, t1.z}, p2{t2.x, t2.y, t2.z};
  V3 P = cross(p0 - p1, p0 - p2);
  float r = (float)norm(P);
  float d12 = (float)norm(p1 - p2);
  float res = r / d12;
  V3 v = p2 - p1;
  double den = (double)(d12 * r);
  V3 jac{(P.y * v.z - P.z * v.y) / den, (P.z * v.x - P.x * v.z) / den, (P.x * v.y - P.y * v.x) / den};
  float s = 1;
  if (iter >= prm.icp_freq) s = (float)(1 - 1.8 * (double)fabsf(res));
  if (s > 0.1 && res != 0) {
    o.accepted = 1;
    o.c[0] = (float)((double)s * jac.x);
    o.c[1] = (float)((double)s * jac.y);
    o.c[2] = (float)((double)s * jac.z);
    o.c[3] = s * res;
  }
}

// ---------------------------------------------------------------------------
// exact all-pairs search + literal index walk (the reference's own control flow)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void nn_brute(const float4* __restrict__ tg, int nt, float sx, float sy,
                                         float sz, int& best_j, float& best_d) {
  best_j = -1;
  best_d = INFINITY;
  for (int j = 0; j < nt; ++j) {  // wave-uniform address: one broadcast load per step
    float4 t = tg[j];
    float d = sqdist3(t.x, t.y, t.z, sx, sy, sz);
    if (d < best_d) best_d = d, best_j = j;  // ascending j, strict < : lowest index wins
  }
}

__device__ __forceinline__ void walk_surf_literal(const float4* __restrict__ tg, int nt, int nq,
                                                  float thr, int closest, float sx, float sy, float sz,
                                                  int& m2, int& m3) {
  m2 = m3 = -1;
  int ring = ring_of(tg[closest].w);
  float d2 = thr, d3 = thr;
  int fend = nq < nt ? nq : nt;
  for (int j = closest + 1; j < fend; ++j) {
    float4 t = tg[j];
    int rj = ring_of(t.w);
    if ((double)rj > ring + 2.5) break;
    float d = sqdist3(t.x, t.y, t.z, sx, sy, sz);
    if (rj <= ring) {
      if (d < d2) d2 = d, m2 = j;
    } else {
      if (d < d3) d3 = d, m3 = j;
    }
  }
  for (int j = closest - 1; j >= 0; --j) {
    float4 t = tg[j];
    int rj = ring_of(t.w);
    if ((double)rj < ring - 2.5) break;
    float d = sqdist3(t.x, t.y, t.z, sx, sy, sz);
    if (rj >= ring) {
      if (d < d2) d2 = d, m2 = j;
    } else {
      if (d < d3) d3 = d, m3 = j;
    }
  }
}

__device__ __forceinline__ void walk_corner_literal(const float4* __restrict__ tg, int nt, int nq,
                                                    float thr, int closest, float sx, float sy, float sz,
                                                    int& m2) {
  m2 = -1;
  int ring = ring_of(tg[closest].w);
  float d2 = thr;
  int fend = nq < nt ? nq : nt;
  for (int j = closest + 1; j < fend; ++j) {
    float4 t = tg[j];
    int rj = ring_of(t.w);
    if ((double)rj > ring + 2.5) break;
    float d = sqdist3(t.x, t.y, t.z, sx, sy, sz);
    if (rj > ring && d < d2) d2 = d, m2 = j;
  }
  for (int j = closest - 1; j >= 0; --j) {
    float4 t = tg[j];
    int rj = ring_of(t.w);
    if ((double)rj < ring - 2.5) break;
    float d = sqdist3(t.x, t.y, t.z, sx, sy, sz);
    if (rj < ring && d < d2) d2 = d, m2 = j;
  }
}

// ---------------------------------------------------------------------------
// per-iteration constants from the linearisation state (wave-uniform)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void make_iter_const(const double* filt, IterConst& ic) {
  Q4 q{ic.lin[6], ic.lin[7], ic.lin[8], ic.lin[9]};
  ic.phi = quat2axis(q);
  ic.Rt = mtrans(qmat(q));
  ic.Gt = mtrans(rinvleft(V3{-ic.phi.x, -ic.phi.y, -ic.phi.z}));
  // boxMinus(filter, lin), KF:84-94
  Q4 qf{filt[6], filt[7], filt[8], filt[9]};
  V3 da = quat2axis(qmul(qinverse(q), qf));
  for (int k = 0; k < 3; ++k) {
    ic.d[0 + k] = filt[0 + k] - ic.lin[0 + k];
    ic.d[3 + k] = filt[3 + k] - ic.lin[3 + k];
    ic.d[9 + k] = filt[10 + k] - ic.lin[10 + k];
    ic.d[12 + k] = filt[13 + k] - ic.lin[13 + k];
    ic.d[15 + k] = filt[16 + k] - ic.lin[16 + k];
  }
  ic.d[6] = da.x, ic.d[7] = da.y, ic.d[8] = da.z;
}

// boxPlus (KF:71-81) applied in place to a 19-vector
__device__ __forceinline__ void box_plus_inplace(double* s, const double* dx) {
  for (int k = 0; k < 3; ++k) {
    s[0 + k] += dx[0 + k];
    s[3 + k] += dx[3 + k];
    s[10 + k] += dx[9 + k];
    s[13 + k] += dx[12 + k];
    s[16 + k] += dx[15 + k];
  }
  Q4 q = qnormalized(qmul(Q4{s[6], s[7], s[8], s[9]}, axis2quat(V3{dx[6], dx[7], dx[8]})));
  s[6] = q.w, s[7] = q.x, s[8] = q.y, s[9] = q.z;
}

__device__ __forceinline__ int sidx(int k) { return k < 3 ? k : k + 3; }  // {0,1,2,6,7,8}
// position of (i, j), i <= j, in the row-major upper triangle of a 6x6
__device__ __forceinline__ int tri6(int i, int j) { return 6 * i - (i * (i - 1)) / 2 + (j - i); }
__device__ __forceinline__ double sym6(const double* tri, int i, int j) {
  return i <= j ? tri[tri6(i, j)] : tri[tri6(j, i)];
}

}  // namespace lins
