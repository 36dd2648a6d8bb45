// ieskf_lds_impl.h — the LDS-resident IESKF kernel: the fast path for VLP-16 sized scans.
// Included by ieskf_lds.hip and ieskf_lds_mr.hip, which instantiate it in two shapes:
//
//   full  (ieskf_lds.hip)     the whole scan in LDS, one 1024-thread workgroup per CU (it declares
//                             ~159 KB of the CU's 160 KB), three lanes per query: the shortest
//                             critical path for ONE scan (single-scan latency);
//   mr    (ieskf_lds_mr.hip)  "multi-resident": only the first LINS_LDS_CAP grid positions (the
//                             corner cloud and the low surf rings, where nearly every search
//                             ends) live in LDS, the rest in a grid-sorted copy in global memory
//                             that the same loops fall through to; 512 threads, one OWNER lane
//                             per query (the queries spread over all waves), no row slots — two
//                             workgroups (= independent scans) fit one CU and hide each other's
//                             barriers and serial tails (batch throughput).  After the cold
//                             iteration only the queries whose certificates fail search, and a
//                             wave serves each of those with up to kCoopMaxLanes of its lanes
//                             (coop_map / coop_lanes below).
//
// One workgroup owns one scan pair for the whole iterated update.  Both target clouds of
// the scan come counting-sorted into (ring x azimuth-column) grids of 16-byte records
// (x, y, z, original index bits — one ds_read_b128 per candidate) — the scan's search index, built once per target
// cloud by grid_index_kernel (ieskf_grid.h: the reference's kd-tree build, setInputCloud in updatePointCloud,
// SE:1156-1160, is outside performIESKF too).  The kernel copies the index's tables and its first LINS_LDS_CAP
// records into LDS in one coalesced pass; after that every iteration's correspondence search is LDS traffic
// (positions beyond the cap: rare L2 reads of the sorted copy) — the candidate windows are staged in LDS, not
// re-gathered from L2.
//
// The kernel is bound by instruction issue and latency (~60 % VALU-busy
// with 31 of 64 lanes at work), not by HBM, so what round 2 did to it is mostly about the
// instructions a wave executes: arithmetic that only feeds pruning bounds or certificates
// takes the hardware's 1-ulp sqrt / rcp (bound_sqrtf, bound_divf), azimuth columns a
// 12-instruction atan2 (az_bin_lds), the rotation maps their short series (lins_math.h), the
// 6x6 solve a Gauss-Jordan over one wave (ieskf_rowsum.h) — each with the argument why the
// results cannot change, or the tolerance under which they may, next to the code.
//
// Parameters supplied by the including translation unit:
//   LINS_LDS_NS         namespace of this instantiation
//   LINS_LDS_CAP        grid positions resident in LDS
//   LINS_LDS_NMAX       target points a scan may have (> CAP: hybrid LDS / global storage)
//   LINS_LDS_SCANBATCH  points whose LDS reads are in flight together in the scan loops
//   LINS_LDS_WAVES      waves of the largest workgroup shape instantiated
//   LINS_LDS_MINW       minimum waves per SIMD the register allocation must allow
//   LINS_LDS_BYTES      LDS budget of one workgroup, checked at compile time
//
//   per iteration
//     3 lanes / query   de-skew (f64, redundantly per lane) -> exact NN + index walk on
//                       the LDS grid, the ring windows of one query split over its
//                       three lanes and merged with wave shuffles on (distance, key)
//     1 lane / query    plane / line residual + Jacobian (f64 -> f32) -> H row in registers
//     every wave        28 f64 sums by a fixed-shape register butterfly (ieskf_rowsum.h
//                       wave_reduce_rows), then an ordered fold over the wave partials
//     42 lanes of wave 0  6x6 Gauss-Jordan, one element per lane (ieskf_rowsum.h wave_gj_solve6), dx
//     wave 0            NaN / divergence / convergence, boxPlus, next constants
//   16 waves x 21 queries = 336 queries per round = the VLP-16 caps (144 flat + 192 sharp).
//
// Scans that do not fit (more than kNpCap target points, ring ids >= 16, unsorted
// rings) take the global-memory kernel in ieskf_kernels.hip — same results.

#include <hip/hip_runtime.h>

#include "ieskf_binned.h"
#include "ieskf_device.h"
#include "ieskf_grid.h"
#include "icp_math.h"
#include "icp_wave.h"
#include "ieskf_rowsum.h"

#ifndef LINS_PERSIST
#define LINS_PERSIST 0
#endif
#ifndef LINS_PRIO_SHIFT
#define LINS_PRIO_SHIFT 0  // (0: off)
#endif
#ifndef LINS_PRIO_LEVEL
#define LINS_PRIO_LEVEL 3
#endif
#ifndef LINS_WALK_CACHE
#define LINS_WALK_CACHE 1  // (the second / third points of a query's previous nearest neighbour kept for its return: see the kernel)
#endif
#ifndef LINS_SPREAD_S
// waves the plane / line queries of a 512-thread workgroup are spread over.  Measured on the batch workload (91 plane
// + 164 line queries on average), one GPU call: 5 / 3 (round 1's choice) 0.692 ms, 4 / 4 0.685, 4 / 3 0.705, 3 / 3 0.711,
// 3 / 5 0.712, 2 / 3 0.760 — using fewer than all eight waves never pays.
#define LINS_SPREAD_S 5  // (round 3, with LINS_COOP_COLD = 2: see there)
#define LINS_SPREAD_C 3
#endif

namespace lins {
namespace LINS_LDS_NS {

struct OutRec {
  double residual_norm, update_norm;
  int iters, converged, diverged, m_surf, m_corner, pad[3];
};

// The kernel's ONE parameter (by value: it lies at offset 0 of the kernarg segment).  Round 3 passed these as 28
// separate arguments; the compiler loads every argument at the kernel's entry and keeps it in scalar registers to the
// end — ~60 of the 106 SGPRs a wave has, which the loop then paid for in SGPR spills (568 v_readlane / v_writelane).
// What the loop needs is read through `ka.`; what only the prologue or the epilogue needs (result pointers, the
// relay's buffers) is read where it is used through cold_args(): a laundered pointer to the same kernarg bytes, so
// that those s_loads stay where they are written instead of being hoisted to the entry.
struct KernelArgs {
  DevParams prm;
  const ScanDesc* descs;
  const int* order;
  const GridTables* tabs;
  const double* state_in;
  const double* cov_in;
  const double* lin_in;
  double* state_out;
  double* a6_out;
  double* cov_out;
  OutRec* out;
  lins_pose_record* poses;
  double* sums_out;
  int* counts_out;
  long long* prof_buf;
  double* relay_hdr;
  int* relay_lane;
  int* queue;      // the launch's ticket counters and per-scan flags (kQ* below)
  int* relay_err;  // one word per context (pinned host memory): raised when a queue wait ran out (checked at lins_sync)
  unsigned* walk_cache;  // per query slot 8 words: the second / third points of the nearest neighbour the query had BEFORE (walk cache below); may be null
  int run_gen;           // number of this launch (tags of the walk cache: an entry of an earlier launch does not count)
  int iter_arg, scan_id_base, relay_n, relay_at, relay_cuts, relay_gen;
  int relay_cap;    // scans the context's flag array holds (the debug trace lies behind it)
  int relay_items;  // (persistent launch, LINS_PERSIST) items of the launch: the workgroups draw tickets until they run out
  int relay_spins;  // polls of ~1 us a part waits for its hand-over before it gives up (and the launch reports it)
};
typedef const KernelArgs __attribute__((address_space(4))) * ColdArgs;
__device__ __forceinline__ ColdArgs cold_args() {
  auto p = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));  // (launder: a load through p cannot move above this point)
  return (ColdArgs)p;
}

constexpr int kMaxLWaves = LINS_LDS_WAVES;  // waves of the largest workgroup shape instantiated
#ifndef LINS_LDS_BATCH_BLOCK
#define LINS_LDS_BATCH_BLOCK 512
#endif
constexpr int kBatchBlock = LINS_LDS_BATCH_BLOCK;  // threads of the batch shape (the instantiation with the spread layout, the tickets and the relay)
constexpr int kNpCap = LINS_LDS_CAP;   // grid positions (corner cloud first, then surf ring-major) resident in LDS
constexpr int kNpMax = LINS_LDS_NMAX;  // target points of an eligible scan
constexpr bool kHybrid = kNpMax > kNpCap;  // positions >= kNpCap live in the sorted global copy
constexpr int kScanBatch = LINS_LDS_SCANBATCH;  // points per trip of the scan loops
static_assert(kNpMax >= kNpCap && kNpMax <= kGridNpMax, "positions are u16, indices u16; the index kernel's cap");

typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct LdsStore {
  float4 pt[kNpCap];  // grid-sorted targets, one 16-byte record (x, y, z, original index bits) per position
  GridTables gt;  // cell ends, ring ranges, elevation wedges: copied from the scan's prebuilt index (ieskf_grid.h)
  double P[324];
  // Series coefficients of the de-skew's axis2Quat (lins_math.h axis2quat_tab), read from here where they are used: as
  // literals the compiler keeps all eighteen in registers across the search loop (6 -> 42 spilled registers, +7 %);
  // from LDS the short series replaces libm's sin / cos + sqrt + three divisions per query and iteration: -2.2 %.
  double trig[kSincCosTab];
  IterConst ic;
  double filt[19];
  double sums[28];
  double partial[kMaxLWaves * 28 > kIcpWorkspace ? kMaxLWaves * 28 : kIcpWorkspace];  // one partial 28-vector per wave (and the ICP step's workspace)
  double aug[3][42];  // one staging copy of [N | z] per solving wave
  double res_prev, res_last, upd_norm;
  int scan_tmp[kMaxLWaves + 4];
  int m_surf, m_corner, iter, conv, div, pad;
  long long prof_acc[16];  // phase profile accumulators of the PROF variant (written by thread 0)
#ifdef LINS_PROF_TAIL
  long long prof_tail[2];  // stamps inside solve_wave0 (system built, solved)
#endif
#ifdef LINS_PROF2
  int prof2[64];  // per wave x phase ticks of the iterations >= LINS_PROF2 (lane 0 of each wave; lins_debug_wave_phases)
  int prof3[32];  // per wave: [0] wave-iterations with a nearest-neighbour search [1] ... with a walk [2] searches [3] walks
#endif
  int dbg[4];  // [0] certificate disagreements (verify mode) [1] NN searches skipped [2] walks skipped
};
static_assert(sizeof(LdsStore) <= LINS_LDS_BYTES, "LDS budget of one workgroup");

// The one LDS block of the workgroup.  File scope so that out-of-line device functions address
// it as LDS (ds_* instructions) instead of through a generic pointer.
__shared__ LdsStore g_lds;

struct LCloud {  // one target cloud's grid (all pointers into LDS)
  const unsigned short* cell_end;  // this cloud's cells (absolute positions)
  const int* ring_start;
  const float2* el_ang;
  int naz, stride, base, n;
  const float4* gs;  // hybrid storage: grid-sorted copy (x, y, z, index bits) of positions >= n_lds
  int n_lds;
};

// ---- one grid position -> point.  Positions below n_lds are LDS-resident; in the hybrid layout
// the others are read from the sorted global copy (L2): rare — the searches end in the low rings.
__device__ __forceinline__ void pt_xyz(const LdsStore& L, const LCloud& c, int p, float& x, float& y, float& z) {
  if (!kHybrid || p < c.n_lds) {
    const float4 v = L.pt[p];
    x = v.x, y = v.y, z = v.z;
  } else {
    const float4 g = c.gs[p];
    x = g.x, y = g.y, z = g.z;
  }
}
__device__ __forceinline__ int pt_idx(const LdsStore& L, const LCloud& c, int p) {
  if (!kHybrid || p < c.n_lds) return __float_as_int(L.pt[p].w);
  return __float_as_int(c.gs[p].w);
}
__device__ __forceinline__ float pt_sqdist(const LdsStore& L, const LCloud& c, int p, float sx, float sy, float sz) {
  float x, y, z;
  pt_xyz(L, c, p, x, y, z);
  return sqdist3(x, y, z, sx, sy, sz);
}

// polar view of a de-skewed query, computed once and shared by both search passes
struct QueryPolar {
  float rho, qn3, el, inv_unused;
  int a0_surf_or_corner;
};

// Running best of a search: (distance, tie key) packed so that ONE unsigned 64-bit compare
// is the reference's "strict < , first seen wins" rule: distances are non-negative floats
// (their bit patterns order like the values, NaN above everything), the tie key is the
// original index (pass 1: lowest index wins) or the visit rank (pass 2).  Initialised to
// (threshold, 0): a candidate must be strictly closer than the threshold (SE:851, 856).
struct Best {
  unsigned long long k;   // winner: (distance bits, tie key); pos < 0: none yet (k = threshold sentinel)
  int pos, ring;
  unsigned long long k2;  // runner-up, kept with its identity so that a near-tie can be re-decided
  int pos2, ring2;        //   later from two distance evaluations instead of a search
  float omin;             // smallest squared distance of every OTHER candidate seen
  __device__ __forceinline__ float d() const { return __uint_as_float((unsigned)(k >> 32)); }
  __device__ __forceinline__ int key() const { return (int)(unsigned)k; }
};
__device__ __forceinline__ Best best_init(float thr) {
  // (no runner-up yet: key (+inf, 0xFFFFFFFF), so that its distance word reads +inf)
  return Best{(unsigned long long)__float_as_uint(thr) << 32, -1, -1, 0x7F800000FFFFFFFFull, -1, -1, INFINITY};
}
__device__ __forceinline__ unsigned long long pack_key(float d, int key) {
  return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)key;
}
// insert candidate (k, pos, ring) into the top-2 list; equal keys are the same point seen again
__device__ __forceinline__ void insert_key(Best& b, unsigned long long k, int pos, int ring) {
  if (k < b.k) {
    if (b.pos2 >= 0) b.omin = fminf(b.omin, __uint_as_float((unsigned)(b.k2 >> 32)));
    b.k2 = b.k, b.pos2 = b.pos, b.ring2 = b.ring;  // (a dethroned sentinel stays a sentinel: pos2 < 0)
    b.k = k, b.pos = pos, b.ring = ring;
  } else if (k > b.k) {
    if (k < b.k2) {
      if (b.pos2 >= 0) b.omin = fminf(b.omin, __uint_as_float((unsigned)(b.k2 >> 32)));
      b.k2 = k, b.pos2 = pos, b.ring2 = ring;
    } else if (k > b.k2) {
      b.omin = fminf(b.omin, __uint_as_float((unsigned)(k >> 32)));
    }
  }
}
__device__ __forceinline__ void consider(Best& b, float d, int key, int pos, int ring) {
  insert_key(b, pack_key(d, key), pos, ring);
}
// The scan loops' form of consider(): nearly every scanned point is farther than the current
// runner-up and only lowers `omin` — one compare and a select, no branch; the 64-bit key logic runs
// for the few that can enter the top two (ties on the runner-up's distance included).  `ok` masks
// points that are not candidates at all (walk rank filter, batch padding).
// (a branch-free insertion was measured: + 11 % — profiles/history/kernel_notes.md #insert)
__device__ __forceinline__ void consider_scan(Best& b, bool ok, float d, int key, int pos, int ring) {
  const bool cand = ok && d <= __uint_as_float((unsigned)(b.k2 >> 32));
  b.omin = (ok && !cand) ? fminf(b.omin, d) : b.omin;
  if (cand) insert_key(b, pack_key(d, key), pos, ring);
}
__device__ __forceinline__ void merge_from_lane(Best& b, int src_lane) {
  unsigned lo = __shfl((unsigned)b.k, src_lane), hi = __shfl((unsigned)(b.k >> 32), src_lane);
  unsigned lo2 = __shfl((unsigned)b.k2, src_lane), hi2 = __shfl((unsigned)(b.k2 >> 32), src_lane);
  int pos = __shfl(b.pos, src_lane), ring = __shfl(b.ring, src_lane);
  int pos2 = __shfl(b.pos2, src_lane), ring2 = __shfl(b.ring2, src_lane);
  float om = __shfl(b.omin, src_lane);
  b.omin = fminf(b.omin, om);
  if (pos >= 0) insert_key(b, ((unsigned long long)hi << 32) | lo, pos, ring);
  if (pos2 >= 0) insert_key(b, ((unsigned long long)hi2 << 32) | lo2, pos2, ring2);
}

// fold the partial results of a query's lanes (consecutive, starting at lane_base) into every one of them.
// LANES = 3: the three-lane kernel; LANES = 0: ln lanes, a power of two chosen at run time.
template <int LANES>
__device__ __forceinline__ void merge_query_lanes(Best& b, int lane_base, int role, int ln) {
  if (LANES == 3) {
    merge_from_lane(b, lane_base + (role + 1) % 3);
    merge_from_lane(b, lane_base + (role + 2) % 3);
  } else if (LANES != 1) {
    const int n = LANES ? LANES : ln;
#pragma unroll 1
    for (int m = 1; m < n; m <<= 1) merge_from_lane(b, lane_base + (role ^ m));
  }
}

// Square roots that only feed pruning windows and certificates — every one of them under a relative slack of 1e-6 or
// more — may come straight from v_sqrt_f32 (1 ulp) instead of the correctly rounded sequence (~10 instructions).
// (every use multiplies the result by (1 +- 1e-6) or more in the safe direction: cert_lb / certified, reach.)  With
// bound_divf below: -2.4 % on the batch kernel.  The same treatment of the queries' rho / |q| / elevation (a
// 16-instruction atan for |z| <= rho) measured +-0: not kept.
__device__ __forceinline__ float bound_sqrtf(float x) { return __builtin_amdgcn_sqrtf(x); }

// ... and a quotient that only has to be an UPPER bound: the hardware reciprocal (1 ulp) times a factor that covers it.
__device__ __forceinline__ float bound_divf(float x, float y) { return x * __frcp_rn(y) * (1.f + 3e-7f); }

// ---- certificates: skipping a search that provably returns the same answer -----------------
// After a search run with its pruning bound inflated by `margin` metres, every candidate other
// than the winner is at least  lb = min(sqrt(omin), sqrt(d_best) + margin)  away from the query
// (scanned ones: measured; pruned ones: beyond the inflated bound).  If the de-skewed query has
// since moved by `drift`, any other candidate is still at least lb - drift away, so while
//     dist(query, winner)  <  lb - drift        (with slack for the f32 roundings)
// the winner is the unique strict minimum of the reference's comparison and the search can be
// skipped.  With no winner (nothing inside the search radius) the same test against the radius
// certifies that there is still none.
__device__ __forceinline__ float cert_lb(const Best& b, float thr, float margin) {
  return fminf(bound_sqrtf(b.omin), bound_sqrtf(b.pos >= 0 ? b.d() : thr) + margin);
}
__device__ __forceinline__ bool certified(float d_now, float lb, float drift) {
  return bound_sqrtf(d_now) * (1.f + 4e-6f) + 2e-6f < (lb - drift * (1.f + 4e-6f)) * (1.f - 4e-6f);
}

// ---- azimuth column of a point on the LDS grid ---------------------------------------------
// (az_bin_lds, the column function shared by the build and the queries: ieskf_grid.h)

// ---- columns / windows on the LDS grid ----------------------------------------------------
// All lanes of a wave run the SAME code on different (ring, column-range) data: every
// scan goes through scan_cols(), whose point loop exists once per call site, so splitting
// a query's ring windows over three lanes is real parallelism, not serialised branches.
#ifndef LINS_GLOB_BATCH
#define LINS_GLOB_BATCH 1
#endif
constexpr int kGlobBatch = LINS_GLOB_BATCH;
struct Spans {  // the grid positions of a column window: [s0, e0) and, when the window wraps past the last column, [s1, e1)
  int s0, e0, s1, e1, spans;
  __device__ __forceinline__ int count() const { return (e0 - s0) + (e1 - s1); }
};
__device__ __forceinline__ Spans spans_of(const LCloud& c, int r, int lo, int hi) {
  int s0 = 0, e0 = 0, s1 = 0, e1 = 0, spans = 1;  // (a window that wraps past the last column has a second span)
  if (lo <= hi) {
    const int naz = c.naz, row = r * naz;
    int len = hi - lo;
    if (len >= naz - 1) lo = 0, len = naz - 1;  // at most naz columns
    lo &= naz - 1;  // (naz is a power of two: the non-negative remainder without an integer division, ~20 instructions)
    hi = lo + len;
    const int c0 = row + lo, c1 = row + (hi < naz ? hi : naz - 1);
    s0 = c0 ? (int)c.cell_end[c0 - 1] : c.base;
    e0 = (int)c.cell_end[c1];
    if (hi >= naz) {  // wrapped tail: columns 0 .. hi-naz
      spans = 2;
      s1 = row ? (int)c.cell_end[row - 1] : c.base;
      e1 = (int)c.cell_end[row + hi - naz];
    }
  }
  return Spans{s0, e0, s1, e1, spans};
}
template <class F>
__device__ __forceinline__ void scan_spans(const LdsStore& L, const LCloud& c, const Spans& w, F f) {
#pragma unroll 1
  for (int k = 0; k < w.spans; ++k) {
    const int s = k ? w.s1 : w.s0, e = k ? w.e1 : w.e0;
    const int el = kHybrid ? (e < c.n_lds ? e : c.n_lds) : e;
    // kScanBatch points per trip, all their LDS reads issued before the first is consumed (the
    // loop is latency bound: one dependent LDS round trip per trip instead of per point); the last
    // trip re-reads the final point for its padding lanes and masks them
#pragma unroll 1
    for (int p = s; p < el; p += kScanBatch) {
      float x[kScanBatch], y[kScanBatch], z[kScanBatch];
      int j[kScanBatch];
#pragma unroll
      for (int u = 0; u < kScanBatch; ++u) {
        const int pu = p + u < el ? p + u : el - 1;
        const float4 v = L.pt[pu];
        x[u] = v.x, y[u] = v.y, z[u] = v.z, j[u] = __float_as_int(v.w);
      }
#pragma unroll
      for (int u = 0; u < kScanBatch; ++u) f(x[u], y[u], z[u], j[u], p + u, p + u < el);
    }
    if (kHybrid) {
      // the positions past the resident ones, from the sorted global copy (L2), one record per trip (four in flight: + 2.0 %,
      // profiles/history/kernel_notes.md #globbatch)
#pragma unroll 1
      for (int p = s > c.n_lds ? s : c.n_lds; p < e; p += kGlobBatch) {
        float4 g[kGlobBatch];
#pragma unroll
        for (int u = 0; u < kGlobBatch; ++u) g[u] = c.gs[p + u < e ? p + u : e - 1];
#pragma unroll
        for (int u = 0; u < kGlobBatch; ++u) f(g[u].x, g[u].y, g[u].z, __float_as_int(g[u].w), p + u, p + u < e);
      }
    }
  }
}

template <class F>
__device__ __forceinline__ void scan_cols(const LdsStore& L, const LCloud& c, int r, int lo, int hi, F f) {
  scan_spans(L, c, spans_of(c, r, lo, hi), f);
}

// How many columns either side of a0 can hold a point within sqrt(bound) of the query:
// a point at azimuth difference D from the query is at least rho*sin(D) away (rho for
// D >= 90 deg), so D <= asin(sqrt(bound)/rho); the query sits anywhere inside its own
// column, hence the +2 (one for its offset, one for rounding) — a superset, never less.
// asin(s) <= s + (pi/2 - 1) s^3 on [0, 1] (every term of asin's series beyond s is <= its
// coefficient times s^3, and the coefficients sum to pi/2 - 1): a cheap upper bound is all the
// pruning needs — a slightly wider window never changes the result.
__device__ __forceinline__ float asin_ub(float s) { return s + 0.5707964f * s * s * s; }

__device__ __forceinline__ int reach(const LCloud& c, float rho, float sqrt_bound) {
  const int half = c.naz / 2;
  float s = bound_divf(sqrt_bound * (1.f + 1e-6f) + kSlack * rho + 1e-6f, rho);  // rho == 0 -> inf/nan -> all columns
  if (!(s < 1.f)) return half;
  int k = (int)(asin_ub(s) * (1.f + 1e-6f) * ((float)c.naz * (0.5f / kPiF))) + 2;
  return k < half ? k : half;
}

__device__ __forceinline__ bool ring_nonempty(const LCloud& c, int r) {
  return r >= 0 && r < kRingsBinned && c.ring_start[r + 1] > c.ring_start[r];
}

// ring r can hold a point within sqrt(bound) of the query only if the query's elevation
// is within delta = asin(sqrt(bound)/|q|) of the ring's elevation wedge (a point at
// elevation difference g < 90 deg is at least |q| sin g away, |q| beyond that)
__device__ __forceinline__ float reach_elev(float qn3, float sqrt_bound) {
  float s = bound_divf(sqrt_bound * (1.f + 1e-6f) + kSlack * qn3 + 1e-6f, qn3);
  return s < 1.f ? asin_ub(s) * (1.f + 1e-6f) + kSlack : 4.f;  // 4 rad > any elevation difference
}
__device__ __forceinline__ bool ring_in_reach(const LCloud& c, int r, float el_q, float delta) {
  const float2 w = c.el_ang[r];  // one 8-byte LDS read; empty rings fail both tests
  return el_q >= w.x - delta && el_q <= w.y + delta;
}

// ---- azimuth windows from the query's COLUMN COORDINATE ---------------------------------------------------------------------
// reach() above — the windows of the three-lane single-scan kernel and of the correspondence pass / ICP shapes, whose register budget the two extra window words do not fit: measured 181.5 -> 196.3 us per single-scan update with them — opens a0 - K .. a0 + K with K = floor(D / w) + 2 columns (D = asin(sqrt(bound) / rho), w the
// column width): one column for the query's place inside its own column, one for the errors of the column function — five
// columns of slack around a window whose real width is 2 D / w.  With the query's column coordinate gq = g(q) / w itself
// (the real number az_bin_lds truncates) the window is exact up to the column function's error: a point within angular
// distance D of the query has |g(p) - g(q)| <= D + 2 eps (eps = 4e-3 rad bounds lins_atan2_coarse, lins_math.h), so its
// column lies in floor(gq - D / w - e) .. floor(gq + D / w + e), e = 2 eps / w + 2e-3 — on average 2 D / w + 1.3 columns
// instead of 2 floor(D / w) + 5: about half the points of a typical window.  A superset decision like every other pruning
// step: results cannot change.
struct ColWin {
  int lo, hi;  // columns, unwrapped (spans_of reduces them)
};
__device__ __forceinline__ float az_col_f(float x, float y, int naz) { return (lins_atan2_coarse(y, x) + kPiF) * ((float)naz * (0.5f / kPiF)); }
__device__ __forceinline__ int az_col_of(float gq, int naz) {  // az_bin_lds(): the column the build puts the point in
  const int a = (int)gq;
  return a < 0 ? 0 : (a >= naz ? naz - 1 : a);
}
__device__ __forceinline__ ColWin reach_cols(const LCloud& c, float rho, float sqrt_bound, float gq) {
  const float s = bound_divf(sqrt_bound * (1.f + 1e-6f) + kSlack * rho + 1e-6f, rho);  // rho == 0 -> inf/nan -> all columns
  const int half = c.naz / 2;
  if (!(s < 1.f)) return ColWin{(int)gq - half, (int)gq + half};
  const float cpr = (float)c.naz * (0.5f / kPiF);  // columns per radian
  const float d = asin_ub(s) * (1.f + 1e-6f) * cpr + (2.f * 4e-3f) * cpr + 2e-3f;
  const float lo = floorf(gq - d), hi = floorf(gq + d);
  return (hi - lo >= (float)(c.naz - 1)) ? ColWin{(int)gq - half, (int)gq + half} : ColWin{(int)lo, (int)hi};
}

// ---- pass 1: exact NN; with LANES = 3 the ring windows of a query are split over its lanes
template <int LANES>
__device__ __forceinline__ Best nn_lds(const LdsStore& L, const LCloud& c, float sx, float sy, float sz,
                                       const QueryPolar& qp, float thr, float margin, int rq, int ln, int role,
                                       int lane_base, int warm_pos, int warm_ring) {
  const int LN = LANES ? LANES : ln;  // lanes of this query (LANES = 0: a run-time power of two, wave-uniform)
  Best b = best_init(thr);
  // Warm start (iterations >= 1): last iteration's nearest neighbour is still a candidate, and its
  // distance to the re-de-skewed query bounds the search from the start — the seed scan is skipped
  // and the windows are minimal.  It only tightens bounds; the exact arg-min is still taken over
  // every cell that could beat it, so the result is the same as a cold search.
  const bool warm = warm_pos >= 0;
  if (warm)
    consider(b, pt_sqdist(L, c, warm_pos, sx, sy, sz), pt_idx(L, c, warm_pos), warm_pos, warm_ring);
  const float rho = qp.rho, qn3 = qp.qn3, el_q = qp.el;
  const int a0 = qp.a0_surf_or_corner;
  rq = rq < 0 ? 0 : (rq >= kRingsBinned ? kRingsBinned - 1 : rq);
  int rcur = rq;
  auto f = [&](float x, float y, float z, int j, int p, bool ok) {
    consider_scan(b, ok, sqdist3(x, y, z, sx, sy, sz), j, p, rcur);
  };
  const bool own = ring_nonempty(c, rq);
  // cold seed: columns a0-1..a0+1 of the query's own ring and of its two neighbours, one ring per
  // lane, merged before the bound is fixed — a query that sits between rings (or whose own ring is
  // empty there) still starts from a real neighbour instead of sweeping the whole search radius
  {
    const int rs = LN == 1 ? rq : rq + (role == 0 ? 0 : (role == 1 ? 1 : -1));
    const bool seed = !warm && role < 3 && ring_nonempty(c, rs);
    rcur = rs;
    scan_cols(L, c, rs, seed ? a0 - 1 : 1, seed ? a0 + 1 : 0, f);
    if (LN == 1) {
#pragma unroll 1
      for (int dr = -1; dr <= 1; dr += 2) {
        const bool sd2 = !warm && ring_nonempty(c, rq + dr);
        rcur = rq + dr;
        scan_cols(L, c, rq + dr, sd2 ? a0 - 1 : 1, sd2 ? a0 + 1 : 0, f);
      }
    } else if (!warm) {  // (the lanes of a query agree on `warm`)
      merge_query_lanes<LANES>(b, lane_base, role, ln);
    }
    rcur = rq;
  }
  const int cin = warm ? 0 : 2;  // first column either side that the seed has not covered
  const float B = b.d();  // fixed bound for everything below (conservative: >= the final best)
  const float sqrtB = bound_sqrtf(B) + margin;  // pruning bound inflated by the certificate margin
  const int K = reach(c, rho, sqrtB);
  const float delta = reach_elev(qn3, sqrtB);
  // This lane's tasks as a bit mask (bit i <-> task t = role + LANES i): task 0 = own ring right
  // of the seed, 1 = own ring left of it, 2.. = the other rings rq+1, rq-1, rq+2, ...  The ring
  // tests are independent LDS reads, issued together; only surviving tasks enter the scan loop.
  constexpr int kTasks = 2 + 2 * (kRingsBinned - 1), kPerLane = LANES ? (kTasks + LANES - 1) / LANES : kTasks;
  unsigned todo = 0;
  // One lane per query (the cold iteration): the thirty "other ring" tasks are simply the sixteen rings — bit 2 + r
  // stands for ring r, tested with a compile-time index (half the tests, no task -> ring arithmetic).  The order the
  // tasks are visited in does not matter: the bound is fixed, winner / runner-up / omin are order-free.
  const bool by_ring = LANES == 0 && LN == 1;  // (wave-uniform)
  if (by_ring) {
    todo = (own && K >= cin) ? 3u : 0u;
#pragma unroll
    for (int r = 0; r < kRingsBinned; ++r) todo |= (r != rq && ring_in_reach(c, r, el_q, delta)) ? (4u << r) : 0u;
  } else {
#pragma unroll
    for (int i = 0; i < kPerLane; ++i) {
      const int t = role + LN * i;
      if (LANES == 0 && t >= kTasks) break;
      const int k = t - 2, off = (k >> 1) + 1;
      const int r = t < 2 ? rq : rq + ((k & 1) ? -off : off);
      // (the ring reads issued unconditionally: + 0.75 %, profiles/history/kernel_notes.md #reachtests)
      bool go;
      if (t < 2)
        go = own && K >= cin;
      else
        go = t < kTasks && r >= 0 && r < kRingsBinned && ring_in_reach(c, r < 0 ? 0 : (r >= kRingsBinned ? kRingsBinned - 1 : r), el_q, delta);
      todo |= go ? (1u << i) : 0u;
    }
  }
#pragma unroll 1
  while (todo) {
    const int i = __ffs(todo) - 1;
    todo &= todo - 1;
    const int t = by_ring ? (i < 2 ? i : 2) : role + LN * i;
    const int k = t - 2, off = (k >> 1) + 1;
    const int r = by_ring ? (i < 2 ? rq : i - 2) : (t < 2 ? rq : rq + ((k & 1) ? -off : off));
    rcur = r;
    // own ring: right part (with the centre column when warm) / left part; other rings: whole window
    scan_cols(L, c, r, t == 0 ? a0 + cin : a0 - K, t == 1 ? a0 - (warm ? 1 : 2) : a0 + K, f);
  }
  merge_query_lanes<LANES>(b, lane_base, role, ln);
  return b;
}

// ---- wave-cooperative searches: which lanes need one, and who serves whom ------------------------
#ifndef LINS_COOP_LANES
#define LINS_COOP_LANES 8  // (re-timed at the end of round 2: 4, 8, 16 within 0.2 %)
#endif
#ifndef LINS_COOP_COLD
// lanes per search in the cold iteration, where every query searches.  A wave gives each of its n searches
// 64 / n lanes rounded down to a power of two, so this only bites when a wave holds <= 32 queries: with the plane
// queries spread over five waves (~20 each) their searches — the long ones: five rings, dense ground — get two lanes,
// the line queries (three waves, ~55 each) keep one.  Measured on the round-3 batch (A/B builds, one GPU call):
// 1 lane, 4 + 4 waves 0.7348 ms; 2 lanes, 4 + 4 waves 0.7326; 2 lanes, 5 + 3 waves 0.7025; 4 lanes 0.7300 (4 + 4).
// (Round 2 measured 1 lane best — on the scrambled range images of the old generator, 91 plane queries per scan.)
#define LINS_COOP_COLD 2
#endif
constexpr int kCoopMaxLanes = LINS_COOP_LANES;  // (measured on the batch workload: 2 -> 8.4, 4 -> 8.9, 8 -> 8.9, 16 -> 8.85 M it/s)
struct CoopMap {
  int n;          // searches needed in this wave
  int rank;       // this lane's position in the permutation (needing lanes first, in lane order)
  int owner_map;  // lane r holds the lane id of rank r
};
__device__ __forceinline__ CoopMap coop_map(bool need, int lane) {
  const unsigned long long m = __ballot(need);
  const int n = __popcll(m);
  const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
  const int rank = need ? below : n + (lane - below);  // a permutation of 0..63: every lane receives exactly one push
  return CoopMap{n, rank, __builtin_amdgcn_ds_permute(rank << 2, lane)};
}
// lanes per search when n searches share a wave: 64 / n rounded down to a power of two (1: everyone for himself)
__device__ __forceinline__ int coop_lanes(int n, int cap) {
  const int l = n <= 1 ? 64 : 64 >> (32 - __clz(n - 1));  // = 64 / n rounded down to a power of two, without the division
  return l > cap ? cap : l;
}

constexpr int kBackRankL = 0x40000000;

struct WalkCtx {
  int j1, f_hi, b_lo;
};
__device__ __forceinline__ bool walk_ring_has_candidates(const LCloud& c, const WalkCtx& w, int r) {
  if (r < 0 || r >= kRingsBinned) return false;
  const int rs = c.ring_start[r], re = c.ring_start[r + 1];
  const bool fwd = (w.j1 + 1 > rs ? w.j1 + 1 : rs) < (w.f_hi < re ? w.f_hi : re);
  const bool bwd = (w.b_lo > rs ? w.b_lo : rs) < (w.j1 < re ? w.j1 : re);
  return fwd || bwd;
}

// candidate filter + visit rank of the index walk: forward part (j1, f_hi) in ascending
// order first, then the backward part [b_lo, j1) in descending order
__device__ __forceinline__ bool walk_rank(const WalkCtx& w, int j, int& rank) {
  const bool fwd = (unsigned)(j - w.j1 - 1) < (unsigned)(w.f_hi > w.j1 + 1 ? w.f_hi - w.j1 - 1 : 0);
  const bool bwd = (unsigned)(j - w.b_lo) < (unsigned)(w.j1 - w.b_lo);  // b_lo <= j1 always
  rank = fwd ? j - w.j1 : kBackRankL + (w.j1 - j);
  return fwd || bwd;
}

// index intervals of the walk around nearest neighbour j1 on ring rho (sorted cloud):
// forward (j1, min(N_query, N_target, start[rho+3])), backward [start[rho-2], j1)
__device__ __forceinline__ WalkCtx make_walk_ctx(const LCloud& c, int nq, int j1, int rho) {
  const int fend = nq < c.n ? nq : c.n;
  const int r_hi = rho + 3 < kRingsBinned ? rho + 3 : kRingsBinned;
  const int r_lo = rho - 2 > 0 ? rho - 2 : 0;
  return WalkCtx{j1, fend < c.ring_start[r_hi] ? fend : c.ring_start[r_hi], c.ring_start[r_lo]};
}

// One walk task: candidates of ring r for the running best `cur` (rank-keyed).  full:
// seed window + both extensions with the tightened bound; !full: extensions only (the
// seed of that ring was scanned by all lanes before).  Same code for every lane.
// seed_first: scan the seed window first and tighten the bound; centre_done: columns a0-1..a0+1
// were already scanned (by the all-lane class-2 seed) — otherwise the extensions include them.
__device__ __forceinline__ void walk_task(const LdsStore& L, const LCloud& c, const WalkCtx& w, int r, bool seed_first,
                                          bool centre_done, int a0, float sx, float sy, float sz, float rho_q,
                                          float qn3, float el_q, float margin, Best& cur) {
  bool go = walk_ring_has_candidates(c, w, r);
  if (go) go = ring_in_reach(c, r, el_q, reach_elev(qn3, bound_sqrtf(cur.d()) + margin));
  auto f = [&](float x, float y, float z, int j, int p, bool ok) {
    int rank;
    const bool in_walk = walk_rank(w, j, rank);
    consider_scan(cur, ok && in_walk, sqdist3(x, y, z, sx, sy, sz), rank, p, 0);
  };
  const bool seed = go && seed_first;
  scan_cols(L, c, r, seed ? a0 - 1 : 1, seed ? a0 + 1 : 0, f);
  const bool done = seed_first || centre_done;
  // Widen progressively: the window needed for the current bound, but at most 4x the width already
  // covered per round — when the seed window was empty the bound tightens as soon as the first real
  // candidate shows up, instead of one sweep over the whole search radius.
  int kk = done ? 1 : -1;  // columns a0-kk..a0+kk are covered (-1: nothing yet)
  const int half = c.naz / 2;
  // (one scan per round / the whole window at once: + 6.8 % / +- 0.5 %, profiles/history/kernel_notes.md #walkrounds)
#pragma unroll 1
  for (int round = 0; round < 8 && go; ++round) {
    const int K = reach(c, rho_q, bound_sqrtf(cur.d()) + margin);
    if (K <= kk) break;
    const int nk = kk < 1 ? K : (K < 4 * kk ? K : 4 * kk);  // (a bound from a warm candidate: one shot)
    scan_cols(L, c, r, a0 + kk + 1, a0 + nk, f);
    scan_cols(L, c, r, a0 - nk, a0 - (kk < 0 ? 1 : kk + 1), f);
    kk = nk;
    if (kk >= half) break;
  }
}

// ---- pass 2 (SE:859-910 surf, SE:983-1024 corner) ----------------------------------------
// surf:   class 2 = ring rho (second point), class 3 = rings rho+-1, rho+-2 (third point)
// corner: class 2 = rings rho+-1, rho+-2 (second point on a different ring), no class 3
template <int LANES>
__device__ __forceinline__ void walk_lds(const LdsStore& L, const LCloud& c, bool is_surf, int nq, float thr, int j1,
                                         int rho, float sx, float sy, float sz, const QueryPolar& qp, float margin,
                                         int ln, int role, int lane_base, int warm2, int warm3, bool check_class,
                                         Best& c2, Best& c3) {
  const int LN = LANES ? LANES : ln;
  const WalkCtx w = make_walk_ctx(c, nq, j1, rho);
  c2 = best_init(thr);
  c3 = best_init(thr);
  const float rho_q = qp.rho, qn3 = qp.qn3, el_q = qp.el;
  const int a0 = qp.a0_surf_or_corner;
  // warm start: last iteration's second / third point are candidates whose distances bound the walk from the
  // start.  Same nearest neighbour => same index intervals and classes; after the nearest neighbour moved
  // (check_class) they still qualify when they lie in the new intervals and on a ring of the right class.
  auto warm_cand = [&](Best& b, int pos, bool on_ring_rho) {
    int rank;
    if (pos < 0) return;
    const int j = pt_idx(L, c, pos);
    if (!walk_rank(w, j, rank)) return;
    if (check_class) {
      int r = 0;  // ring of index j: the last ring that starts at or before it
#pragma unroll
      for (int step = kRingsBinned / 2; step > 0; step >>= 1)
        if (c.ring_start[r + step] <= j) r += step;
      if ((r == rho) != on_ring_rho) return;
    }
    consider(b, pt_sqdist(L, c, pos, sx, sy, sz), rank, pos, 0);
  };
  warm_cand(c2, warm2, is_surf);  // second point: ring rho for planes, another ring for lines
  warm_cand(c3, warm3, false);    // third point (planes only): another ring
  const bool w2 = c2.pos >= 0, w3 = c3.pos >= 0;
  if (is_surf && !w2) {  // class-2 seed on ring rho: all lanes of the query
    bool go = walk_ring_has_candidates(c, w, rho);
    auto f = [&](float x, float y, float z, int j, int p, bool ok) {
      int rank;
      const bool in_walk = walk_rank(w, j, rank);
      consider_scan(c2, ok && in_walk, sqdist3(x, y, z, sx, sy, sz), rank, p, 0);
    };
    scan_cols(L, c, rho, go ? a0 - 1 : 1, go ? a0 + 1 : 0, f);
  }
  // tasks (data, not code), dealt round-robin to the query's lanes:
  //   surf    t0: rho (class 2, extensions)  t1: rho-1  t2: rho-2  t3: rho+1  t4: rho+2  (class 3)
  //   corner  t0: rho-1  t1: rho+1  t2: rho-2  t3: rho+2
  constexpr int kPerLane = LANES ? (5 + LANES - 1) / LANES : 5;
#pragma unroll 1
  for (int i = 0; i < kPerLane; ++i) {
    const int t = role + LN * i;
    if (LANES == 0 && t > 4) break;
    int dr;
    if (is_surf)
      dr = t == 0 ? 0 : (t == 1 ? -1 : (t == 2 ? -2 : (t == 3 ? 1 : (t == 4 ? 2 : 99))));
    // (this order suits the three-lane shape: the forward bound j < N_query, SE:859, leaves rho, rho -1, rho -2 as the only
    // rings with candidates for nearly every query — one per lane; walk_lean deals adjacent rings first for its 1 / 2 / 4 / 8 lanes)
    else
      dr = t == 0 ? -1 : (t == 1 ? 1 : (t == 2 ? -2 : (t == 3 ? 2 : 99)));
    const bool use2 = !is_surf || dr == 0;
    const bool have_bound = use2 ? w2 : w3;  // a warm bound replaces the per-ring seed scan
    const bool seed_first = !have_bound && (!is_surf || dr != 0);
    const bool centre_done = is_surf && dr == 0 && !w2;
    Best cur = use2 ? c2 : c3;
    walk_task(L, c, w, rho + dr, seed_first, centre_done, a0, sx, sy, sz, rho_q, qn3, el_q, margin, cur);
    if (use2)
      c2 = cur;
    else
      c3 = cur;
  }
  merge_query_lanes<LANES>(c2, lane_base, role, ln);
  merge_query_lanes<LANES>(c3, lane_base, role, ln);
}

// ---- the cooperative searches of one wave (LANES == 1 shapes): what the kernel's loop calls -------------------------
// The cm.n searches a wave needs are served coop_lanes() lanes each; inputs travel from the owner lanes to the serving
// lanes and the results back by wave shuffles (no LDS, no barrier).  Called by every lane of the wave (wave-uniform cm.n).
struct NnOut {
  int pos, ring, pos2, ring2;
  float lb;
};
struct WalkOut {
  int r2, r2b, r3, r3b;
  float lb2, lb3;
};
__device__ __forceinline__ QueryPolar polar_of(float sx, float sy, float sz, int naz) {
  QueryPolar qp;
  qp.rho = sqrtf(sx * sx + sy * sy);
  qp.qn3 = sqrtf(qp.rho * qp.rho + sz * sz);
  qp.el = atan2f(sz, qp.rho);
  qp.inv_unused = 0.f;
  qp.a0_surf_or_corner = az_bin_lds(sx, sy, naz);
  return qp;
}
__device__ __forceinline__ NnOut coop_nn(const LdsStore& L, const LCloud& c, const CoopMap& cm, int coop_cap, bool need_nn, int lane, float sx,
                                         float sy, float sz, const QueryPolar& qp, int rq, int a1, int ra1, float thr, float margin,
                                         bool skip) {
  // lanes per search: as many as the wave can give each of its cm.n searches, a power of two
  const int ln = coop_lanes(cm.n, coop_cap);
  const int wrole = lane & (ln - 1), wbase = lane - wrole, item = lane / ln;
  bool valid = need_nn;
  float isx = sx, isy = sy, isz = sz;
  QueryPolar iq = qp;
  int i_rq = rq, i_a1 = a1, i_ra1 = ra1;  // (the kind is the wave-round's: nothing to ship)
  if (ln > 1) {  // the inputs of search `item` travel from its owner to the ln lanes that serve it
    valid = item < cm.n;
    const int owner = __shfl(cm.owner_map, valid ? item : 0);
    isx = __shfl(sx, owner), isy = __shfl(sy, owner), isz = __shfl(sz, owner);
    iq.rho = __shfl(qp.rho, owner), iq.qn3 = __shfl(qp.qn3, owner), iq.el = __shfl(qp.el, owner);
    iq.a0_surf_or_corner = __shfl(qp.a0_surf_or_corner, owner);
    i_rq = __shfl(i_rq, owner);
    i_a1 = __shfl(a1, owner), i_ra1 = __shfl(ra1, owner);
  }
  Best bb = best_init(thr);
  if (valid && !skip)  // (profiling aid: LINS_DEBUG_SKIP=2 skips the search, 1 skips the walk)
    bb = nn_lds<0>(L, c, isx, isy, isz, iq, thr, margin, i_rq, ln, wrole, wbase, i_a1, i_ra1);
  NnOut r{bb.pos, bb.ring, bb.pos2, bb.ring2, cert_lb(bb, thr, margin)};
  if (ln > 1) {  // hand back: the owner of rank r reads the first lane of group r
    const int src = (cm.rank * ln) & 63;
    r.pos = __shfl(r.pos, src), r.ring = __shfl(r.ring, src);
    r.pos2 = __shfl(r.pos2, src), r.ring2 = __shfl(r.ring2, src), r.lb = __shfl(r.lb, src);
  }
  return r;
}
__device__ __forceinline__ WalkOut coop_walk(const LdsStore& L, const LCloud& c, bool is_surf, int nq, const CoopMap& cm, int coop_cap,
                                             bool need_walk, int lane, float sx, float sy, float sz, const QueryPolar& qp, int j1, int rho1,
                                             int w2, int w3, bool nn_changed, float thr, float margin, bool skip) {
  const int ln = coop_lanes(cm.n, coop_cap);
  const int wrole = lane & (ln - 1), wbase = lane - wrole, item = lane / ln;
  bool valid = need_walk;
  float isx = sx, isy = sy, isz = sz;
  QueryPolar iq = qp;
  int i_j1 = j1, i_rho1 = rho1, i_w2 = w2, i_w3 = w3, i_chk = nn_changed;
  if (ln > 1) {
    valid = item < cm.n;
    const int owner = __shfl(cm.owner_map, valid ? item : 0);
    isx = __shfl(sx, owner), isy = __shfl(sy, owner), isz = __shfl(sz, owner);
    iq.rho = __shfl(qp.rho, owner), iq.qn3 = __shfl(qp.qn3, owner), iq.el = __shfl(qp.el, owner);
    iq.a0_surf_or_corner = __shfl(qp.a0_surf_or_corner, owner);
    i_j1 = __shfl(j1, owner), i_rho1 = __shfl(rho1, owner);
    i_w2 = __shfl(w2, owner), i_w3 = __shfl(w3, owner), i_chk = __shfl((int)nn_changed, owner);
  }
  Best c2 = best_init(thr), c3 = c2;
  if (valid && !skip)
    walk_lds<0>(L, c, is_surf, nq, thr, i_j1, i_rho1, isx, isy, isz, iq, margin, ln, wrole, wbase, i_w2, i_w3, i_chk != 0, c2, c3);
  WalkOut r{c2.pos, c2.pos2, c3.pos, c3.pos2, cert_lb(c2, thr, margin), cert_lb(c3, thr, margin)};
  if (ln > 1) {
    const int src = (cm.rank * ln) & 63;
    r.r2 = __shfl(r.r2, src), r.r2b = __shfl(r.r2b, src), r.r3 = __shfl(r.r3, src), r.r3b = __shfl(r.r3b, src);
    r.lb2 = __shfl(r.lb2, src), r.lb3 = __shfl(r.lb3, src);
  }
  return r;
}
// ---- the register-lean search core of the batch kernel (round 6): Top2, nn_lean, walk_lean, the carry records ----------
constexpr int kRelayLanes = 512;                    // query slots of a scan's carry records
constexpr int kRelayRegionInts = 4 * kRelayLanes * 4;  // ints per scan in KernelArgs::relay_lane: [4][512] 16-byte words, by QUERY slot
constexpr int kRelayLaneInts = kRelayRegionInts;
#ifdef LINS_LDS_LEAN
#include "ieskf_lds_lean.h"
constexpr bool kLeanBuild = true;
#else
constexpr bool kLeanBuild = false;
#endif
// one target cloud's grid view of the workgroup's LDS block (cs / cc of the kernel)
__device__ __forceinline__ LCloud make_cloud(const LdsStore& L, bool is_surf, const float4* gs, int n_corner_t, int n_surf_t, int n_lds) {
  return is_surf ? LCloud{L.gt.cell_end + kCellsCorner, L.gt.ring_start[0], &L.gt.el_ang[0][0], kAzSurf, 1, n_corner_t, n_surf_t, gs, n_lds}
                 : LCloud{L.gt.cell_end, L.gt.ring_start[1], &L.gt.el_ang[1][0], kAzCorner, kAzSurf / kAzCorner, 0, n_corner_t, gs, n_lds};
}
// ---- grid load: the scan's prebuilt index (grid_index_kernel, ieskf_grid.hip — the reference's setInputCloud,
// SE:1156-1160, outside performIESKF) into LDS: the tables and the first n_lds records of the sorted copy as 16-byte
// words, every read of a thread in flight before its first LDS write (~1 HBM round trip per scan).  Ends with a
// barrier.
// Every move is unconditional on a CLAMPED index (the threads past the end re-copy the last word: same value to the
// same address), so that nothing ties a read to a branch: written with guards, the compiler sinks each read into its
// guard and waits for it there — nine dependent HBM round trips instead of one.
// Inlined since round 5 (-2 %).  Rounds 3-4 kept it out of line because, inlined, the 1024 x 3 correspondence-pass
// instantiation returned grid position 0 as the second point of every line query.  Root cause (round 6, tools/repro/README.md):
// a register-allocator fault of this compiler — the copies of a live-range split and spill stores placed at the top of the
// block where a divergent branch ends IN FRONT of its `s_or_b64 exec`, so the lanes that skipped the branch keep stale
// registers (there: the partner lane of merge_query_lanes<3>).  Where the allocator splits depends on register pressure —
// inlining this function is one of the things that move it.  tools/check_exec_prologue.py finds the shape in the shipped
// library; __graft_entry__.build() and tests/test_build_check.py refuse a library that has it.
template <int BLOCK>
#ifndef LINS_GRID_INLINE
#define LINS_GRID_INLINE 1
#endif
#if LINS_GRID_INLINE
__device__ __forceinline__
#else
__device__ __noinline__
#endif
    void
    load_lds_grid(const GridTables* __restrict__ tab, const float4* __restrict__ gs, int n_lds, int tid) {
  LdsStore& L = g_lds;
  constexpr int kTabPer = (kGridTableWords + BLOCK - 1) / BLOCK, kPtPer = (kNpCap + BLOCK - 1) / BLOCK;
  constexpr int kChunk = kPtPer < 12 ? kPtPer : 12;
  const uint4* src = reinterpret_cast<const uint4*>(tab);
  uint4* dst = reinterpret_cast<uint4*>(&L.gt);
  uint4 tw[kTabPer];
#pragma unroll
  for (int u = 0; u < kTabPer; ++u) tw[u] = src[min(tid + u * BLOCK, kGridTableWords - 1)];
  if (n_lds > 0) {  // (uniform)
#pragma unroll
    for (int k0 = 0; k0 < kPtPer; k0 += kChunk) {
      float4 pbuf[kChunk];
#pragma unroll
      for (int u = 0; u < kChunk; ++u) pbuf[u] = gs[min(tid + (k0 + u) * BLOCK, n_lds - 1)];
#pragma unroll
      for (int u = 0; u < kChunk; ++u) L.pt[min(tid + (k0 + u) * BLOCK, n_lds - 1)] = pbuf[u];
    }
  }
#pragma unroll
  for (int u = 0; u < kTabPer; ++u) dst[min(tid + u * BLOCK, kGridTableWords - 1)] = tw[u];
  __syncthreads();
}

#include "ieskf_lds_tail.h"  // the serial tail of an iteration (solve, next constants, ICP step) and the Joseph epilogue

// ---------------------------------------------------------------------------
// the kernel.  PASS_ONLY: one correspondence pass at a caller-supplied linearisation
// state (lins_correspondences / lins_reduce_pass), dumping records / sums.
// ---------------------------------------------------------------------------
// Hand-over words of the relay (below) cross workgroups — possibly XCDs, whose L2s are not coherent with one another for
// ordinary accesses inside a kernel — as agent-scope relaxed atomics / sc1 buffer accesses: each access carries the cache
// policy that makes it coherent at device scope (sc1), instead of fences that write back / invalidate a whole L2 per
// workgroup (measured: with __threadfence() on both sides the launch took 0.97 instead of 0.67 ms).
__device__ __forceinline__ void relay_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int relay_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void relay_st(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<long long*>(p), __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double relay_ld(const double* p) {
  return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int relay_add(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// A scan's flag only ever rises within a launch (16 gen + next part ... 16 gen + 15 = finished): raised with an atomic max.
#ifndef LINS_RELAY_RELEASE
#define LINS_RELAY_RELEASE 1  // (the flag is raised with release / read with acquire semantics at agent scope; 0: the relaxed forms of rounds 3-5 — A/B knob)
#endif
__device__ __forceinline__ void relay_raise(int* p, int v) {
  __hip_atomic_fetch_max(p, v, LINS_RELAY_RELEASE ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int relay_ld_acq(const int* p) {
  return __hip_atomic_load(p, LINS_RELAY_RELEASE ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the work items of a batch launch: tickets (the batch shape only; KernelArgs::queue, relay_n > 0) ---------------------
// A batch with more scans than the device has workgroup slots is launched with one workgroup per WORK ITEM — (scan, part):
// "run scan s from cut p to cut p + 1" (relay_next_cut, ieskf_device.h) — and a LIST of the items (KernelArgs::order): every
// part 0 in the host's longest-expected-first order, then every part 1 in that order, ...  A workgroup does not take the
// item of its index: it DRAWS A TICKET when it starts (an atomic counter) and takes the item the list has there.  Part
// p > 0 waits for the flag of its scan to say that part p - 1 has handed the loop state over (relay_out below), or that
// the scan is finished (stop rule met in an earlier part: nothing to do).
//   Why the wait cannot starve its producer, whatever order the dispatcher hands workgroups out in: the item a workgroup
//   waits for has a smaller ticket (the list names a part behind the part before it), tickets are drawn by workgroups
//   that have STARTED, so its holder is resident — running, or itself waiting for a still smaller ticket; the chain ends
//   at a part 0, which never waits.  Rounds 3-4 took item blockIdx.x and relied on blocks being handed out in index
//   order, with a bounded spin and a whole-update fallback ("solo") in case they were not; both are gone.  The bound on
//   the wait stays as a safety net that REPORTS (relay_err -> lins_sync fails): a launch never spins for ever.
//   Q[kQHead]        tickets drawn
//   Q[kQExited]      workgroups that left: the last one out resets both counters for the next launch
//   Q[kQFlags + s]   flag of scan s: 16 gen + p once part p - 1 has handed over, 16 gen + 15 once the update is finished;
//                    gen numbers the launches of the context, so the flags are never reset (the host clears them when
//                    gen would overflow)
// All of these are device-scope atomics (the parts of a scan may sit on different XCDs); the hand-over data is complete at
// the memory side before the flag rises, and the taker reads it with device-coherent loads.
// (a FIFO of continuations and a priority FIFO for the late parts were built and measured slower / equal in round 5:
// profiles/r05_batch_kernel_variants.md, profiles/history/kernel_notes.md #fifo)
constexpr int kQHead = 0, kQExited = 32, kQFlags = 64;  // (ints; the counters on lines of their own)

// (The per-query loop state of an update lives in the scan's carry records — ieskf_lds_lean.h — so a part hands over only the
// header below; rounds 3-5 packed 13 words per lane into the same buffer at every cut: history at 51c48b0.)

// One work item: the update of scan `scan` — from its start (cont = false) or from the loop state another workgroup
// handed over (cont = true: the batch shape under the work queue) — to its end or to the next cut.  Returns true when
// the scan is finished (results written), false when it was handed over.
template <int BLOCK, int LANES, bool PASS_ONLY, bool PROF, bool ICP, bool KNOBS>
__device__ __forceinline__ bool ieskf_lds_update(const KernelArgs ka, const float4* __restrict__ arena, const float4* __restrict__ sorted,
                                                 int4* __restrict__ idx_store, lins_corr* __restrict__ dump, const int scan, const int part) {
  const bool cont = part > 0;
  const DevParams prm = ka.prm;
  const int relay_n = ka.relay_n, relay_at = ka.relay_at;
  const int pad = KNOBS ? prm.pad : 0;  // (debug / counting flags: a constant 0 in the production instantiations)
  constexpr bool prof = PROF;  // phase profile compiled in only for the debug variant
  constexpr int kLBlock = BLOCK, kQPerWave = 64 / LANES, kQPerRound = (BLOCK / 64) * kQPerWave;
  static_assert(BLOCK >= 256 && BLOCK / 64 <= kMaxLWaves, "block shape");
  // (LANES > 1 is dispatched for single-round scans only, see effective_search() in lins_capi.hip)
  LdsStore& L = g_lds;
  // optional phase profile: [0] setup+grid build [1] correspondence [2] reduction [3] solve [4] update [5] total
  // accumulators live in LDS (thread 0 only) so that the profiled variant keeps the register
  // allocation of the production one: [0] setup [1] corr [2] reduce [3] solve [4] update [5] total
  // [6..9] thread 0's own de-skew / NN / walk / geometry  [10..15] per-iteration time of iterations 0..5
  if (prof && threadIdx.x < 16) g_lds.prof_acc[threadIdx.x] = 0;
#ifdef LINS_PROF2
  if (prof && threadIdx.x < 64) g_lds.prof2[threadIdx.x] = 0;
  if (prof && threadIdx.x < 32) g_lds.prof3[threadIdx.x] = 0;
#define PROF2_ADD(ph, dt)                                                                   \
  do {                                                                                      \
    if (prof && lane == 0 && iter >= LINS_PROF2 && wave < 8) g_lds.prof2[wave * 8 + (ph)] += (int)(dt); \
  } while (0)
#else
#define PROF2_ADD(ph, dt) \
  do {                    \
    (void)sizeof(dt);     \
  } while (0)
#endif
  const long long t_begin = prof ? clock64() : 0, t_wall_begin = prof ? wall_clock64() : 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave: a scalar)
  // Several-part updates (the batch shape only; relay_n = scans of the launch, 0 = off): the update of a scan is cut every
  // relay_at iterations into parts that run as separate work items of one launch (the tickets above), so that a launch of
  // 2 x (slots) scans is several rounds of shorter jobs instead of two rounds of whole updates: the end of the launch (slots
  // idle while the last whole updates finish) shrinks with the job size — 0.665 -> 0.615 ms per 1024 scans x 10 iterations in
  // parts of four iterations (tools/relay_sweep.py; parts of 2: 0.647, 3: 0.625, 5: 0.641, 6: 0.623, 7: 0.626, 8: 0.641).
  // A part takes over the loop state through global memory (relay_out / the take-over below).  Same arithmetic in the same
  // order: results do not depend on the cuts, bit for bit (tests/test_gpu_parity.py
  // test_two_part_updates_return_the_whole_updates_bits).
  // the register-lean form of the correspondence phase (ieskf_lds_lean.h): per-query state in the scan's carry records
  constexpr bool kLean = kLeanBuild && LANES == 1 && !PASS_ONLY && !ICP;
  constexpr bool kRelay = kLean && BLOCK == kBatchBlock;  // (the several-part updates rest on those records)
  const ScanDesc sd = ka.descs[scan];
  const int total = sd.n_surf_q + sd.n_corner_q;
  // hybrid storage: this scan's slice of the sorted copy (same offsets as its targets in the arena:
  // the corner targets follow the surf targets, so positions 0 .. n_all-1 fit)
  const float4* const gs = sorted + sd.off_surf_t;
  const int n_all_t = sd.n_surf_t + sd.n_corner_t, n_lds = n_all_t < kNpCap ? n_all_t : kNpCap;

  for (int k = tid; k < 324; k += kLBlock) L.P[k] = (PASS_ONLY || ICP) ? 0.0 : ka.cov_in[(size_t)scan * 324 + k];
  if (tid < 19) {
    double v = ka.state_in[(size_t)scan * 19 + tid];
    L.filt[tid] = v;
    L.ic.lin[tid] = PASS_ONLY ? ka.lin_in[(size_t)scan * 19 + tid] : v;
  }
  if (tid < 28) L.sums[tid] = 0;
  if (tid == 64) lins_sinc_cos_table(L.trig);
  if (tid == 0) {
    L.res_prev = 1e6, L.res_last = 0, L.upd_norm = 0;
    L.iter = PASS_ONLY ? ka.iter_arg : 0, L.conv = 0, L.div = 0, L.m_surf = 0, L.m_corner = 0;
    L.dbg[0] = L.dbg[1] = L.dbg[2] = L.dbg[3] = 0;
  }
  __syncthreads();
  if (tid < 64 && !(kRelay && cont)) {  // wave 0, lane-redundant: constants of the first iteration (a later part takes them over)
    IterConst ic;
    double filt[19];
    for (int k = 0; k < 19; ++k) ic.lin[k] = L.ic.lin[k], filt[k] = L.filt[k];
    make_iter_const_tail(filt, ic);
    if (tid == 0) {
      L.ic.phi = ic.phi, L.ic.Rt = ic.Rt, L.ic.Gt = ic.Gt;
      for (int k = 0; k < 18; ++k) L.ic.d[k] = ic.d[k];
    }
  }
  load_lds_grid<BLOCK>(ka.tabs + scan, gs, n_lds, tid);  // ends with a barrier
  if (prof && tid == 0) L.prof_acc[0] = clock64() - t_begin;

  const LCloud cs{L.gt.cell_end + kCellsCorner, L.gt.ring_start[0], &L.gt.el_ang[0][0], kAzSurf, 1, sd.n_corner_t, sd.n_surf_t,
                  gs, n_lds};
  const LCloud cc{L.gt.cell_end, L.gt.ring_start[1], &L.gt.el_ang[1][0], kAzCorner, kAzSurf / kAzCorner, 0, sd.n_corner_t,
                  gs, n_lds};
  const int role = lane % LANES, lane_base = lane - role, q_in_wave = lane / LANES;
  const bool lane_used = lane < kQPerWave * LANES;
  // ---- query -> lane layout of the one-lane-per-query shapes: WAVE-ROUNDS -------------------------------------------------
  // A wave takes, per round, up to 64 queries of ONE kind (plane or line), so that the kind, the target cloud's grid
  // (LCloud) and every branch on them are wave-uniform: scalar selects and branches instead of per-lane selects with both
  // clouds' parameters live across the search code.  Two layouts:
  //   spread  the plane queries (the costlier kind: two walks over five rings) evenly over the first kSpreadSurf waves,
  //           the line queries over the rest, one round — a wave runs the union of its lanes' search paths, so the
  //           queries are spread as thin as the waves allow; the lane <-> query mapping is fixed for the whole update;
  //   rounds  query sets that do not fit one round: wave-rounds of 64 dealt to the waves round after round (nothing is
  //           carried between iterations then).
  // (a third, denser layout for the late iterations: + 1.9 ... + 9 %, profiles/history/kernel_notes.md #denselate)
  struct WrLay {
    int n_wr, k;  // wave-rounds of the scan, this wave's wave-round (>= n_wr: none)
    bool kind_s;  // its kind
    int slot;     // this lane's query: plane queries first, then line queries
    bool active;
  };
  constexpr int kWaves = BLOCK / 64, kSpreadSurf = LINS_SPREAD_S > 0 && BLOCK == kBatchBlock ? LINS_SPREAD_S : (kWaves * 5 + 4) / 8;
  constexpr int kWs = kSpreadSurf;
  constexpr int kWc = LINS_SPREAD_C > 0 && BLOCK == kBatchBlock ? LINS_SPREAD_C : kWaves - kWs;
  // The spread layout's shares.  The queries of a kind come sorted by ring and what a search costs goes with the ring (the
  // far ground rings' walks are the longest phase of every iteration: tools/wave_phases.py), so equal COUNTS leave the
  // last plane wave the slowest in half of the workgroups; the waves of a kind take contiguous blocks whose sizes follow
  // LINS_SPREAD_WS / _WC instead (weights; equal weights = equal counts).
#ifndef LINS_SPREAD_WS
#define LINS_SPREAD_WS 30, 24, 20, 14, 12  // (round 5, one GPU call: equal counts 0.5770 ms; 27,23,20,16,14 0.5721; 30,24,20,14,12 0.5686; 33,25,19,12,11 0.5740; weights on the line waves too: slower)
#endif
#ifndef LINS_SPREAD_WC
#define LINS_SPREAD_WC 1, 1, 1
#endif
  constexpr int kWtS[] = {LINS_SPREAD_WS, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1}, kWtC[] = {LINS_SPREAD_WC, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
  constexpr int kSpreadMax = 16;  // waves of one kind at most
  static_assert(LANES != 1 || kWaves <= kSpreadMax + 1, "spread layout: weights for every wave");
  constexpr bool kWeighted = BLOCK == kBatchBlock && LINS_SPREAD_S > 0;  // (the batch shape; the other one-lane shapes deal equal counts)
  // first query of wave-round j of a kind with n queries over nw waves: floor(n * (w_0 + ... + w_{j-1}) / (w_0 + ... + w_{nw-1}))
  auto share_start = [&](const int* wt, int nw, int n, int j) {
    int tot = 0, cum = 0;
#pragma unroll
    for (int k = 0; k < kSpreadMax; ++k) tot += k < nw ? (kWeighted ? wt[k] : 1) : 0, cum += k < j ? (kWeighted ? wt[k] : 1) : 0;
    return j >= nw ? n : (n * cum) / tot;
  };
  bool fits = kWs < kWaves;
#pragma unroll
  for (int k = 0; k < kSpreadMax; ++k) {
    if (k < kWs) fits = fits && share_start(kWtS, kWs, sd.n_surf_q, k + 1) - share_start(kWtS, kWs, sd.n_surf_q, k) <= 64;
    if (k < kWc) fits = fits && share_start(kWtC, kWc, sd.n_corner_q, k + 1) - share_start(kWtC, kWc, sd.n_corner_q, k) <= 64;
  }
  const bool spread_ok = fits;  // the one-round spread layout holds the scan
  auto wr_layout = [&](int rnd) {
    WrLay y;
    y.k = rnd * kWaves + wave;
    if (spread_ok) {
#ifdef LINS_SPREAD_ROLES  // (experiment of round 6, +-0: which share a wave takes — a map from the wave to its place in the list "plane shares near to far, then line shares", e.g. -DLINS_SPREAD_ROLES=0,5,1,6,2,7,3,4)
      constexpr int kRole[] = {LINS_SPREAD_ROLES, 8, 9, 10, 11, 12, 13, 14, 15, 16};
      const int role_k = (kWeighted && y.k < kWs + kWc && y.k < 8) ? kRole[y.k] : y.k;
#else
      const int role_k = y.k;
#endif
      y.n_wr = kWs + kWc, y.kind_s = role_k < kWs;
      const int j = y.kind_s ? role_k : role_k - kWs, nw = y.kind_s ? kWs : kWc, n = y.kind_s ? sd.n_surf_q : sd.n_corner_q;
      const int q0 = share_start(y.kind_s ? kWtS : kWtC, nw, n, j), q1 = share_start(y.kind_s ? kWtS : kWtC, nw, n, j + 1);
      y.slot = (y.kind_s ? 0 : sd.n_surf_q) + q0 + lane, y.active = y.k < y.n_wr && lane < q1 - q0;
    } else {
      const int nws = (sd.n_surf_q + 63) >> 6;
      y.n_wr = nws + ((sd.n_corner_q + 63) >> 6), y.kind_s = y.k < nws;
      const int q0 = y.kind_s ? y.k * 64 : (y.k - nws) * 64;
      const int left = (y.kind_s ? sd.n_surf_q : sd.n_corner_q) - q0;
      y.slot = (y.kind_s ? 0 : sd.n_surf_q) + q0 + lane, y.active = y.k < y.n_wr && lane < (left < 64 ? left : 64);
    }
    return y;
  };

  // last iteration's triplet of this lane's query (grid positions) for the warm start; only
  // meaningful when the scan needs a single round (the lane <-> query mapping is then fixed)
  int a1 = -1, b1c = -1, ra1 = -1, rb1 = -1;  // nearest neighbour: tracked winner / runner-up (+ their rings)
  int a2 = -1, b2c = -1, a3 = -1, b3c = -1;    // second and third point: tracked winner / runner-up
  int sel1 = -1;                               // last iteration's nearest neighbour
  // certificates of those three selections (see cert_lb / certified): lower bounds of every other
  // candidate's distance and the query positions they were established at
  float lb1 = 0.f, lb2 = 0.f, lb3 = 0.f, certA[3] = {0.f, 0.f, 0.f}, certB[3] = {0.f, 0.f, 0.f};
  bool have_cert = false;
  bool searched = false;  // a search iteration has run: certificates and warm candidates exist (uniform)
  constexpr int kRelayHdr = 64;  // doubles per scan: IterConst (58), res_prev, res_last, upd_norm, then 6 ints
  static_assert(sizeof(IterConst) == 58 * sizeof(double), "relay header layout");
  static_assert(kGridNpMax < 32768, "grid positions travel as signed 16-bit words");
  if (kRelay && cont) {  // take-over: the loop state the part before left (the barrier of the grid load has passed)
    const double* h = ka.relay_hdr + (size_t)scan * kRelayHdr;
    if (tid < 58) reinterpret_cast<double*>(&L.ic)[tid] = relay_ld(h + tid);
    if (tid == 64) L.res_prev = relay_ld(h + 58), L.res_last = relay_ld(h + 59), L.upd_norm = relay_ld(h + 60);
    if (tid == 65) {
      const int* hi = reinterpret_cast<const int*>(h + 61);
      L.iter = relay_ld(hi), L.dbg[0] = relay_ld(hi + 1), L.dbg[1] = relay_ld(hi + 2), L.dbg[2] = relay_ld(hi + 3), L.dbg[3] = relay_ld(hi + 4);
    }
    searched = true;  // (the carry records lie where the part before left them)
    __syncthreads();
  }
  bool relay_out = false;
  // (ticketed launch) this item ends at the scan's next cut
  const int cut_at = (kRelay && relay_n > 0) ? relay_next_cut(L.iter, relay_at, ka.relay_cuts) : 0x7FFFFFFF;
#ifdef LINS_PROF2
  int dbg_nn = 0, dbg_walk = 0, dbg_walk_mask = 0, dbg_slot = -1;  // this lane's query: searches / walks in the iterations >= LINS_PROF2
#endif

  for (;;) {
    const int iter = L.iter;
    if (!PASS_ONLY && (iter >= prm.num_iter || L.conv || L.div)) break;
    if (kRelay && iter >= cut_at) {
      relay_out = true;
      break;
    }
    __syncthreads();  // everyone has read the loop state before it is rewritten
    if (tid == 0) L.m_surf = 0, L.m_corner = 0;

    const bool do_search = PASS_ONLY || (iter % prm.icp_freq) == 0;
    double acc = 0;
    int ms = 0, mc = 0;
    long long t0 = prof ? clock64() : 0, t1 = t0, t2 = t0, t3 = t0;
    // Query -> lane layout.  LANES == 1: wave-rounds (wr_layout above).  LANES == 3: when both kinds fit one round with the
    // corner queries starting on a wave boundary, do that: no wave then mixes plane and line code paths.
    int surf_waves = (sd.n_surf_q + kQPerWave - 1) / kQPerWave;
    const bool aligned = surf_waves * kQPerWave + sd.n_corner_q <= kQPerRound;
    const int n_wr = LANES == 1 ? wr_layout(0).n_wr : 0;
    const int span = LANES == 1 ? ((n_wr + kWaves - 1) / kWaves) * kQPerRound
                                : (aligned ? surf_waves * kQPerWave + sd.n_corner_q : total);  // row slots in use
    int4 held = make_int4(0, 0, 0, -1);  // (ICP, ICP_FREQ > 1, single round) a corner triplet waiting for the plane-row count
    int part_k = wave;  // the slot of this wave's partial sums: its wave-round
    for (int base = 0; base < span; base += kQPerRound) {
      const int vslot = base + wave * kQPerWave + q_in_wave;  // position in the (padded) layout
      int slot = vslot;                                        // query index: surf first, then corner
      bool active = lane_used && vslot < span;
      bool kind_s = true;  // (LANES == 1) this wave-round's kind, wave-uniform
      int wr_k = 0;        // (LANES == 1) this wave's wave-round
      if (LANES == 1) {
        const WrLay y = wr_layout(base / kQPerRound);
        wr_k = y.k, kind_s = y.kind_s, slot = y.slot, active = y.active;
        if (base == 0) part_k = wave;  // (= y.k of the first round)
      } else if (aligned) {
        if (wave < surf_waves)
          active = active && vslot < sd.n_surf_q;
        else
          slot = vslot - surf_waves * kQPerWave + sd.n_surf_q;
      }
      double row[7] = {0, 0, 0, 0, 0, 0, 0};
      if (span > kQPerRound) {  // several rounds: the lane <-> query mapping changes, nothing carries over
        a1 = b1c = ra1 = rb1 = a2 = b2c = a3 = b3c = sel1 = -1, have_cert = false;
        lb1 = lb2 = lb3 = 0.f;
      }
      if constexpr (kLean) {
#ifdef LINS_LDS_LEAN
#include "ieskf_lds_lean_body.inc"
#endif
      } else if constexpr (LANES == 1) {
        // ---- one owner lane per query, wave-cooperative searches --------------------------------------
        // The owner de-skews its query and tests the certificates; the queries of the wave that do need a
        // search are then served several lanes at a time (inputs and results travel by bpermute, no LDS
        // and no barrier): a wave's search time follows the number of searches it really has to run, not
        // the union of 64 independent control flows.
        const bool is_surf = kind_s;  // (wave-uniform)
        const int qi = is_surf ? slot : slot - sd.n_surf_q;
        const LCloud& c = is_surf ? cs : cc;
        const float thr = prm.nearest_f;
        const bool single_round = span <= kQPerRound;  // (the lane <-> query mapping is fixed)
        const bool warm_iter = single_round && searched;  // certificates / warm candidates exist (uniform)
        const float margin = warm_iter ? prm.margin_warm : prm.margin_cold;
        const bool verify = (pad & 8) != 0;  // test aid: search anyway and count disagreements
        // the cold iteration searches for every query: one lane each (more lanes per wave cost more than the shorter
        // chains give back — measured); afterwards only the uncertified queries search, up to kCoopMaxLanes lanes each
        const int coop_cap = warm_iter ? kCoopMaxLanes : LINS_COOP_COLD;
        const unsigned long long kNone = ~0ull;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        V3 phi = L.ic.phi;
        QueryOut o;
        o.accepted = 0;
        o.c[0] = o.c[1] = o.c[2] = o.c[3] = 0.f;
        o.sel[0] = o.sel[1] = o.sel[2] = 0.f;
        QueryPolar qp = {0.f, 0.f, 0.f, 0.f, 0};
        bool qp_ready = false;  // (wave-uniform)
        int p1 = -1, p2 = -1, p3 = -1;
        long long s0 = prof ? clock64() : 0, s1 = s0, s2 = s0;
        if (active) {
          if (pad & 0x400000)  // (counting aid: no query load)
            q = make_float4(1.f + lane, 2.f, 0.5f, 3.25f);
          else
            q = arena[(is_surf ? sd.off_surf_q : sd.off_corner_q) + qi];
          V3 t{L.ic.lin[0], L.ic.lin[1], L.ic.lin[2]};
          if (pad & 0x100000)  // (counting aid: no de-skew)
            o.sel[0] = q.x, o.sel[1] = q.y, o.sel[2] = q.z;
          else
            transform_to_start(prm, phi, t, q, o.sel[0], o.sel[1], o.sel[2], g_lds.trig);
        }
        if (prof) {
          s1 = clock64();
#ifndef LINS_PROF_WAVES
          if (tid == 0) L.prof_acc[6] += s1 - s0;
#endif
          PROF2_ADD(0, s1 - s0);
        }
        auto dist_to = [&](int pos) { return pt_sqdist(L, c, pos, o.sel[0], o.sel[1], o.sel[2]); };
        auto drift_from = [&](const float* cp) {
          float ex = o.sel[0] - cp[0], ey = o.sel[1] - cp[1], ez = o.sel[2] - cp[2];
          return bound_sqrtf(ex * ex + ey * ey + ez * ez);
        };
        if (do_search) {
          if (!single_round) a1 = b1c = a2 = b2c = a3 = b3c = sel1 = -1;
          // --- nearest neighbour: certificate (owner) -----------------------------------------------------
          // Per selection the last search left two tracked candidates — the winner A and the runner-up B
          // (grid positions, -1 = absent) — and a lower bound lb for the distance of every other
          // candidate.  While everybody else is certified to stay farther than the closer of A and B,
          // the selection is re-decided between those two from their distances alone (same strict
          // (distance, key) order as the search); otherwise the search runs again, warm-started from A.
          bool need_nn = false, said = false, flip = false;
          int pred = -1;
          if (active && (pad & 0x80000) && warm_iter) {  // (counting aid: every certificate holds, unchecked)
            pred = a1, said = true;
          } else if (active) {
            const float da = a1 >= 0 ? dist_to(a1) : INFINITY, db = b1c >= 0 ? dist_to(b1c) : INFINITY;
            bool ok = warm_iter && !(pad & 16) && certified(fminf(fminf(da, db), thr), lb1, drift_from(certA));
            const unsigned long long ka = da < thr ? pack_key(da, pt_idx(L, c, a1)) : kNone;
            const unsigned long long kb = db < thr ? pack_key(db, pt_idx(L, c, b1c)) : kNone;
            flip = kb < ka;
            pred = (flip ? kb : ka) == kNone ? -1 : (flip ? b1c : a1);
            said = ok;
            if (verify) ok = false;
            need_nn = !ok;
          }
          // --- nearest neighbour: the searches of this wave, coop_lanes() lanes each ---------------------------
          {
            const CoopMap cm = coop_map(need_nn, lane);
#ifdef LINS_PROF2
            if (prof && lane == 0 && iter >= LINS_PROF2 && cm.n) g_lds.prof3[wave * 4 + 0] += 1, g_lds.prof3[wave * 4 + 2] += cm.n;
#endif
            int r_pos = -1, r_ring = -1, r_pos2 = -1, r_ring2 = -1;
            float r_lb = 0.f;
            if (cm.n) {  // (wave-uniform)
              // The polar view of the query (two square roots, an atan2f, the column) only feeds the searches: made here,
              // by every lane of a wave that searches — wave-uniform, so no lane waits for another's branch; a wave
              // whose selections are all certified (nearly every wave of a late iteration) never computes it.
              if (!(pad & 0x200000)) qp = polar_of(o.sel[0], o.sel[1], o.sel[2], c.naz), qp_ready = true;
              const NnOut r = coop_nn(L, c, cm, coop_cap, need_nn, lane, o.sel[0], o.sel[1], o.sel[2], qp, ring_of(q.w), a1, ra1, thr, margin, (pad & 2) != 0);
              r_pos = r.pos, r_ring = r.ring, r_pos2 = r.pos2, r_ring2 = r.ring2, r_lb = r.lb;
            }
            if (need_nn) {
              p1 = r_pos;  // (a winner beat the threshold sentinel, so its distance is < thr, SE:851)
              if (said && p1 != pred) atomicAdd(&L.dbg[0], 1);
              a1 = r_pos, ra1 = r_ring, b1c = r_pos2, rb1 = r_ring2;
              lb1 = r_lb;
              certA[0] = o.sel[0], certA[1] = o.sel[1], certA[2] = o.sel[2];
            } else if (active) {
              p1 = pred;
              if (flip) {  // the runner-up took over: swap the two tracked candidates
                const int tp = a1, tr = ra1;
                a1 = b1c, ra1 = rb1, b1c = tp, rb1 = tr;
              }
              atomicAdd(&L.dbg[1], 1);
            }
          }
          const bool nn_changed = p1 != sel1;
          // Walk cache.  A query that sits on the bisector of two target points changes its nearest neighbour back and
          // forth — the ICP flip-flop: the correspondence flips, the row with it, the state moves by a micron, the
          // correspondence flips back — and every change asks for the walk again, because the candidate sets of the second /
          // third point hang on the nearest neighbour (its index and ring, SE:859-910).  Round 4 measured it: 0.8 % of the
          // plane queries walk in EVERY late iteration and account for 60 % of the late walks, each ~20 k ticks on the
          // critical path of its workgroup (tools/wave_phases.py).  So the tracked candidates and certificates of the walk
          // are kept PER NEAREST NEIGHBOUR: on a change, the outgoing neighbour's set goes to the query's entry in global
          // memory and, if the entry holds the set of the incoming one, that set comes back and is judged like any other —
          // it was established by an exact walk for exactly this neighbour over the same clouds, so its certificate says
          // what it said then.  (Tags carry the launch number: nothing survives a run; device-coherent accesses: the parts
          // of a cut update may sit on different XCDs.)
          bool set_ok = !nn_changed;  // the carried second / third points belong to this nearest neighbour
          if (LINS_WALK_CACHE && !PASS_ONLY && warm_iter && nn_changed && active && ka.walk_cache && !(pad & 0x800000)) {  // (LINS_DEBUG_SKIP bit 0x800000: no walk cache)
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(ka.walk_cache + (size_t)sd.slot_base * 8, 0, total * 32, 0x00020000);
            const unsigned gen16 = (unsigned)ka.run_gen << 16;
            const v4u e0 = __builtin_amdgcn_raw_buffer_load_b128(rs, slot * 32, 0, 16);
            const v4u e1 = __builtin_amdgcn_raw_buffer_load_b128(rs, slot * 32 + 16, 0, 16);
            if (sel1 >= 0) {  // (the carried set is sel1's: a walk or a judged set stood behind every selected neighbour)
              __builtin_amdgcn_raw_buffer_store_b128(v4u{gen16 | (unsigned)sel1, ((unsigned)a2 & 0xFFFFu) | ((unsigned)b2c << 16),
                                                         ((unsigned)a3 & 0xFFFFu) | ((unsigned)b3c << 16), __float_as_uint(lb2)},
                                                     rs, slot * 32, 0, 16);
              __builtin_amdgcn_raw_buffer_store_b128(v4u{__float_as_uint(lb3), __float_as_uint(certB[0]), __float_as_uint(certB[1]),
                                                         __float_as_uint(certB[2])},
                                                     rs, slot * 32 + 16, 0, 16);
            }
            if (p1 >= 0 && e0.x == (gen16 | (unsigned)p1)) {
              a2 = (int)(short)(e0.y & 0xFFFFu), b2c = (int)e0.y >> 16, a3 = (int)(short)(e0.z & 0xFFFFu), b3c = (int)e0.z >> 16;
              lb2 = __uint_as_float(e0.w), lb3 = __uint_as_float(e1.x);
              certB[0] = __uint_as_float(e1.y), certB[1] = __uint_as_float(e1.z), certB[2] = __uint_as_float(e1.w);
              set_ok = true;
            }
          }
          if (active) sel1 = p1;
          if (prof) {
            s2 = clock64();
#ifndef LINS_PROF_WAVES
            if (tid == 0) L.prof_acc[7] += s2 - s1;
#endif
            PROF2_ADD(1, s2 - s1);
          }
          // --- second / third point: certificate (owner) --------------------------------------------------
          bool need_walk = false, flip2 = false, flip3 = false, said23 = false;
          int pred2 = -1, pred3 = -1, j1 = -1;
          if (active && p1 >= 0) {
            j1 = pt_idx(L, c, p1);  // (p1 >= 0 => p1 is candidate A, on ring ra1)
            need_walk = !set_ok || !warm_iter;
            if (!need_walk && (pad & 0x80000)) {
              pred2 = a2, pred3 = a3, said23 = true;
            } else if (!need_walk) {
              const WalkCtx w = make_walk_ctx(c, is_surf ? sd.n_surf_q : sd.n_corner_q, j1, ra1);
              const float dB = drift_from(certB);
              auto judge = [&](int pa, int pb, float lb, int& pd, bool& fl) {
                const float da = pa >= 0 ? dist_to(pa) : INFINITY, db = pb >= 0 ? dist_to(pb) : INFINITY;
                int rka = 0, rkb = 0;
                if (pa >= 0) walk_rank(w, pt_idx(L, c, pa), rka);
                if (pb >= 0) walk_rank(w, pt_idx(L, c, pb), rkb);
                const unsigned long long ka = da < thr ? pack_key(da, rka) : kNone;
                const unsigned long long kb = db < thr ? pack_key(db, rkb) : kNone;
                fl = kb < ka;
                pd = (fl ? kb : ka) == kNone ? -1 : (fl ? pb : pa);
                return certified(fminf(fminf(da, db), thr), lb, dB);
              };
              bool ok23 = judge(a2, b2c, lb2, pred2, flip2);
              if (is_surf) ok23 = judge(a3, b3c, lb3, pred3, flip3) && ok23;
              said23 = ok23;
              need_walk = !ok23 || verify;
            }
          }
          // --- second / third point: the walks of this wave --------------------------------------------------
          {
            const CoopMap cm = coop_map(need_walk, lane);
#ifdef LINS_PROF2
            if (prof && lane == 0 && iter >= LINS_PROF2 && cm.n) g_lds.prof3[wave * 4 + 1] += 1, g_lds.prof3[wave * 4 + 3] += cm.n;
#endif
            int r2 = -1, r2b = -1, r3 = -1, r3b = -1;
            float r_lb2 = 0.f, r_lb3 = 0.f;
            if (cm.n) {
              if (!qp_ready && !(pad & 0x200000)) qp = polar_of(o.sel[0], o.sel[1], o.sel[2], c.naz);
              const int nq = is_surf ? sd.n_surf_q : sd.n_corner_q;
              const WalkOut r = coop_walk(L, c, is_surf, nq, cm, coop_cap, need_walk, lane, o.sel[0], o.sel[1], o.sel[2], qp, j1, ra1, a2, a3, !set_ok, thr,
                                              margin, (pad & 1) != 0);
              r2 = r.r2, r2b = r.r2b, r3 = r.r3, r3b = r.r3b, r_lb2 = r.lb2, r_lb3 = r.lb3;
            }
#ifdef LINS_PROF2
            if (prof && iter >= LINS_PROF2 && active) dbg_slot = sd.slot_base + slot, dbg_walk += need_walk, dbg_nn += need_nn, dbg_walk_mask |= (need_walk ? 1 : 0) << iter | (nn_changed ? 1 : 0) << (16 + iter);
#endif
            if (need_walk) {
              if (said23 && (r2 != pred2 || (is_surf && r3 != pred3))) atomicAdd(&L.dbg[0], 1);
              p2 = r2, p3 = r3;
              a2 = r2, b2c = r2b, a3 = r3, b3c = r3b;
              lb2 = r_lb2, lb3 = r_lb3;
              certB[0] = o.sel[0], certB[1] = o.sel[1], certB[2] = o.sel[2];
            } else if (active && p1 >= 0) {
              p2 = pred2, p3 = pred3;
              if (flip2) {
                const int tp = a2;
                a2 = b2c, b2c = tp;
              }
              if (flip3) {
                const int tp = a3;
                a3 = b3c, b3c = tp;
              }
              atomicAdd(&L.dbg[2], 1);
            }
          }
#ifndef LINS_PROF_WAVES
          if (prof && tid == 0) L.prof_acc[8] += clock64() - s2;
#endif
          if (active && prm.icp_freq > 1) {
            // ICP mode: estimateTransform searches the corners only after >= 10 plane rows were accepted
            // (SE:1175-1178) — a corner triplet is committed after the reduction, once that count is known
            if (ICP && !is_surf)
              held = make_int4(p1, p2, p3, sd.slot_base + slot);
            else
              idx_store[sd.slot_base + slot] = make_int4(p1, p2, p3, 0);
          }
        } else if (active) {
          int4 s = idx_store[sd.slot_base + slot];
          p1 = s.x, p2 = s.y, p3 = s.z;
        }
        long long s3 = prof ? clock64() : 0;
        if (prof && do_search) PROF2_ADD(2, s3 - s2);
        // The iteration constants the rows need (R^T, G^T: 36 registers' worth, the same for every lane) are read from
        // LDS HERE.  Without the fence the compiler hoists those reads to the top of the iteration, finds no
        // registers for them across the search code, and moves them through scratch: ~12 scratch stores and as many
        // waited-for scratch loads per wave and iteration — the bulk of a late iteration's time.
        asm volatile("" ::: "memory");
        if (active && !(pad & 0x20000)) {  // (counting aid: LINS_DEBUG_SKIP bit 0x20000 drops the rows)
          // (a selected point is the first of its two tracked candidates: p1 == a1, p2 == a2, p3 == a3 when they exist)
          auto pt4 = [&](int pos) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            pt_xyz(L, c, pos, v.x, v.y, v.z);
            return v;
          };
          if (is_surf) {
            if (p1 >= 0 && p2 >= 0 && p3 >= 0)
              surf_row(prm, iter, o.sel[0], o.sel[1], o.sel[2], pt4(p1), pt4(p2), pt4(p3), o);
          } else if (p1 >= 0 && p2 >= 0) {
            corner_row(prm, iter, o.sel[0], o.sel[1], o.sel[2], pt4(p1), pt4(p2), o);
          }
          if (o.accepted) {
            if (ICP) {  // Gauss-Newton row of the fallback (SE:1246-1257): [c^T(-R(s phi)[p]x), c^T | -0.05 res]
              const IcpRow ir = icp_row_dev(prm.inv_period, phi.x, phi.y, phi.z, q.x, q.y, q.z, q.w, o.c[0], o.c[1], o.c[2], o.c[3]);
#pragma unroll
              for (int k = 0; k < 7; ++k) row[k] = ir.v[k];
            } else {
              V3 cv{(double)o.c[0], (double)o.c[1], (double)o.c[2]};
              V3 u = cross(V3{(double)q.x, (double)q.y, (double)q.z}, mvec(L.ic.Rt, cv));
              V3 a = mvec(L.ic.Gt, u);
              row[0] = cv.x, row[1] = cv.y, row[2] = cv.z, row[3] = a.x, row[4] = a.y, row[5] = a.z;
              row[6] = prm.lidar_scale * (double)o.c[3];
            }
            if (is_surf)
              ++ms;
            else
              ++mc;
          }
          if (PASS_ONLY && dump) {
            lins_corr r;
            r.ind1 = p1 >= 0 ? pt_idx(L, c, p1) : -1;
            r.ind2 = p2 >= 0 ? pt_idx(L, c, p2) : -1;
            r.ind3 = (is_surf && p3 >= 0) ? pt_idx(L, c, p3) : -1;
            r.accepted = o.accepted;
            for (int k = 0; k < 4; ++k) r.coeff[k] = o.c[k];
            r.sel[0] = o.sel[0], r.sel[1] = o.sel[1], r.sel[2] = o.sel[2], r.sel[3] = q.w;
            dump[sd.slot_base + slot] = r;
          }
        }
#ifndef LINS_PROF_WAVES
        if (prof && tid == 0) L.prof_acc[9] += clock64() - s3;
#endif
        if (prof) PROF2_ADD(3, clock64() - s3);
      } else if (active) {
        const bool is_surf = slot < sd.n_surf_q;
        const int qi = is_surf ? slot : slot - sd.n_surf_q;
        const float4 q = arena[(is_surf ? sd.off_surf_q : sd.off_corner_q) + qi];
        const LCloud& c = is_surf ? cs : cc;
        V3 phi = L.ic.phi;
        V3 t{L.ic.lin[0], L.ic.lin[1], L.ic.lin[2]};
        QueryOut o;
        long long s0 = prof ? clock64() : 0, s1 = s0, s2 = s0;
        transform_to_start(prm, phi, t, q, o.sel[0], o.sel[1], o.sel[2], g_lds.trig);
        if (prof) {
          s1 = clock64();
#ifndef LINS_PROF_WAVES
          if (tid == 0) L.prof_acc[6] += s1 - s0;
#endif
        }
        o.accepted = 0;
        o.c[0] = o.c[1] = o.c[2] = o.c[3] = 0.f;
        int p1 = -1, p2 = -1, p3 = -1;  // grid positions of the three target points
        if (do_search) {
          const bool single_round = span <= kQPerRound;
          if (!single_round) a1 = b1c = a2 = b2c = a3 = b3c = sel1 = -1, have_cert = false;
          QueryPolar qp;
          qp.rho = sqrtf(o.sel[0] * o.sel[0] + o.sel[1] * o.sel[1]);
          qp.qn3 = sqrtf(qp.rho * qp.rho + o.sel[2] * o.sel[2]);
          qp.el = atan2f(o.sel[2], qp.rho);
          qp.inv_unused = 0.f;
          qp.a0_surf_or_corner = az_bin_lds(o.sel[0], o.sel[1], c.naz);
          const float thr = prm.nearest_f;
          const float margin = have_cert ? prm.margin_warm : prm.margin_cold;
          auto dist_to = [&](int pos) { return pt_sqdist(L, c, pos, o.sel[0], o.sel[1], o.sel[2]); };
          auto drift_from = [&](const float* cp) {
            float ex = o.sel[0] - cp[0], ey = o.sel[1] - cp[1], ez = o.sel[2] - cp[2];
            return bound_sqrtf(ex * ex + ey * ey + ez * ez);
          };
          // Per selection the last search left two tracked candidates — the winner A and the
          // runner-up B (grid positions, -1 = absent) — and a lower bound lb for the distance of every
          // other candidate.  While everybody else is certified to stay farther than the closer of
          // A and B, the selection is re-decided between those two from their distances alone
          // (same strict (distance, key) order as the search); otherwise the search runs again,
          // warm-started from A.
          const unsigned long long kNone = ~0ull;
          const bool verify = (pad & 8) != 0;  // test aid: search anyway and count disagreements
          // --- nearest neighbour -----------------------------------------------------------------------
          {
            const float da = a1 >= 0 ? dist_to(a1) : INFINITY, db = b1c >= 0 ? dist_to(b1c) : INFINITY;
            bool ok = have_cert && !(pad & 16) && certified(fminf(fminf(da, db), thr), lb1, drift_from(certA));
            const unsigned long long ka = da < thr ? pack_key(da, pt_idx(L, c, a1)) : kNone;
            const unsigned long long kb = db < thr ? pack_key(db, pt_idx(L, c, b1c)) : kNone;
            const bool flip = kb < ka;
            const int pred = (flip ? kb : ka) == kNone ? -1 : (flip ? b1c : a1);
            const bool said = ok;
            if (verify) ok = false;
            if (!ok) {
              if ((pad & 32) && __ffsll(__ballot(1)) - 1 == lane) atomicAdd(&L.dbg[3], 1 + (iter >= 3 ? 1000 : 0));
              Best bb = best_init(thr);
              if (!(pad & 2))  // (profiling aid: LINS_DEBUG_SKIP=2 skips the search, 1 skips the walk)
                bb = nn_lds<LANES>(L, c, o.sel[0], o.sel[1], o.sel[2], qp, thr, margin, ring_of(q.w), LANES, role, lane_base,
                                   a1, ra1);
              p1 = bb.pos;  // (a winner beat the threshold sentinel, so its distance is < thr, SE:851)
              if (said && p1 != pred && role == 0) atomicAdd(&L.dbg[0], 1);
              a1 = bb.pos, ra1 = bb.ring, b1c = bb.pos2, rb1 = bb.ring2;
              lb1 = cert_lb(bb, thr, margin);
              certA[0] = o.sel[0], certA[1] = o.sel[1], certA[2] = o.sel[2];
            } else {
              p1 = pred;
              if (flip) {  // the runner-up took over: swap the two tracked candidates
                const int tp = a1, tr = ra1;
                a1 = b1c, ra1 = rb1, b1c = tp, rb1 = tr;
              }
              if (role == 0) atomicAdd(&L.dbg[1], 1);
            }
          }
          const bool nn_changed = p1 != sel1;
          sel1 = p1;
          if (prof) {
            s2 = clock64();
#ifndef LINS_PROF_WAVES
            if (tid == 0) L.prof_acc[7] += s2 - s1;
#endif
          }
          // --- second / third point ------------------------------------------------------------------
          if (p1 >= 0) {
            const int j1 = pt_idx(L, c, p1), rho1 = ra1;  // (p1 >= 0 => p1 is candidate A)
            bool need_walk = nn_changed || !have_cert;
            int pred2 = -1, pred3 = -1;
            bool flip2 = false, flip3 = false, said23 = false;
            if (!need_walk) {
              const WalkCtx w = make_walk_ctx(c, is_surf ? sd.n_surf_q : sd.n_corner_q, j1, rho1);
              const float dB = drift_from(certB);
              auto judge = [&](int pa, int pb, float lb, int& pred, bool& flip) {
                const float da = pa >= 0 ? dist_to(pa) : INFINITY, db = pb >= 0 ? dist_to(pb) : INFINITY;
                int rka = 0, rkb = 0;
                if (pa >= 0) walk_rank(w, pt_idx(L, c, pa), rka);
                if (pb >= 0) walk_rank(w, pt_idx(L, c, pb), rkb);
                const unsigned long long ka = da < thr ? pack_key(da, rka) : kNone;
                const unsigned long long kb = db < thr ? pack_key(db, rkb) : kNone;
                flip = kb < ka;
                pred = (flip ? kb : ka) == kNone ? -1 : (flip ? pb : pa);
                return certified(fminf(fminf(da, db), thr), lb, dB);
              };
              bool ok23 = judge(a2, b2c, lb2, pred2, flip2);
              if (is_surf) ok23 = judge(a3, b3c, lb3, pred3, flip3) && ok23;
              said23 = ok23;
              need_walk = !ok23 || verify;
            }
            if (need_walk) {
              if ((pad & 32) && __ffsll(__ballot(1)) - 1 == lane) atomicAdd(&L.dbg[0], 1 + (iter >= 3 ? 1000 : 0));
              Best c2 = best_init(thr), c3 = c2;
              if (!(pad & 1))
                walk_lds<LANES>(L, c, is_surf, is_surf ? sd.n_surf_q : sd.n_corner_q, thr, j1, rho1, o.sel[0], o.sel[1],
                                o.sel[2], qp, margin, LANES, role, lane_base, a2, a3, nn_changed, c2, c3);
              if (said23 && (c2.pos != pred2 || (is_surf && c3.pos != pred3)) && role == 0) atomicAdd(&L.dbg[0], 1);
              p2 = c2.pos, p3 = c3.pos;
              a2 = c2.pos, b2c = c2.pos2, a3 = c3.pos, b3c = c3.pos2;
              lb2 = cert_lb(c2, thr, margin), lb3 = cert_lb(c3, thr, margin);
              certB[0] = o.sel[0], certB[1] = o.sel[1], certB[2] = o.sel[2];
            } else {
              p2 = pred2, p3 = pred3;
              if (flip2) {
                const int tp = a2;
                a2 = b2c, b2c = tp;
              }
              if (flip3) {
                const int tp = a3;
                a3 = b3c, b3c = tp;
              }
              if (role == 0) atomicAdd(&L.dbg[2], 1);
            }
          }
          have_cert = single_round;
#ifndef LINS_PROF_WAVES
          if (prof && tid == 0) L.prof_acc[8] += clock64() - s2;
#endif
          if (prm.icp_freq > 1 && role == 0) {
            // ICP mode: estimateTransform searches the corners only after >= 10 plane rows were accepted
            // (SE:1175-1178) — a corner triplet is committed after the reduction, once that count is known
            if (ICP && !is_surf)
              held = make_int4(p1, p2, p3, sd.slot_base + slot);
            else
              idx_store[sd.slot_base + slot] = make_int4(p1, p2, p3, 0);
          }
        } else {
          int4 s = idx_store[sd.slot_base + slot];
          p1 = s.x, p2 = s.y, p3 = s.z;
        }
        long long s3 = prof ? clock64() : 0;
        asm volatile("" ::: "memory");  // (R^T, G^T are read from LDS here, not hoisted and spilled: see the one-lane path)
        if (role == 0) {
          auto pt4 = [&](int pos) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            pt_xyz(L, c, pos, v.x, v.y, v.z);
            return v;
          };
          if (is_surf) {
            if (p2 >= 0 && p3 >= 0) surf_row(prm, iter, o.sel[0], o.sel[1], o.sel[2], pt4(p1), pt4(p2), pt4(p3), o);
          } else if (p2 >= 0) {
            corner_row(prm, iter, o.sel[0], o.sel[1], o.sel[2], pt4(p1), pt4(p2), o);
          }
          if (o.accepted) {
            if (ICP) {  // Gauss-Newton row of the fallback (SE:1246-1257): [c^T(-R(s phi)[p]x), c^T | -0.05 res]
              const IcpRow ir = icp_row_dev(prm.inv_period, phi.x, phi.y, phi.z, q.x, q.y, q.z, q.w, o.c[0], o.c[1], o.c[2], o.c[3]);
#pragma unroll
              for (int k = 0; k < 7; ++k) row[k] = ir.v[k];
            } else {
              V3 cv{(double)o.c[0], (double)o.c[1], (double)o.c[2]};
              V3 u = cross(V3{(double)q.x, (double)q.y, (double)q.z}, mvec(L.ic.Rt, cv));
              V3 a = mvec(L.ic.Gt, u);
              row[0] = cv.x, row[1] = cv.y, row[2] = cv.z, row[3] = a.x, row[4] = a.y, row[5] = a.z;
              row[6] = prm.lidar_scale * (double)o.c[3];
            }
            if (is_surf)
              ++ms;
            else
              ++mc;
          }
          if (PASS_ONLY && dump) {
            lins_corr r;
            r.ind1 = p1 >= 0 ? pt_idx(L, c, p1) : -1;
            r.ind2 = p2 >= 0 ? pt_idx(L, c, p2) : -1;
            r.ind3 = (is_surf && p3 >= 0) ? pt_idx(L, c, p3) : -1;
            r.accepted = o.accepted;
            for (int k = 0; k < 4; ++k) r.coeff[k] = o.c[k];
            r.sel[0] = o.sel[0], r.sel[1] = o.sel[1], r.sel[2] = o.sel[2], r.sel[3] = q.w;
            dump[sd.slot_base + slot] = r;
          }
        }
#ifndef LINS_PROF_WAVES
        if (prof && tid == 0) L.prof_acc[9] += clock64() - s3;
#endif
      }
      if (prof) t1 = clock64();
#ifdef LINS_PROF_WAVES  // (experiment: per-wave correspondence time of the late iterations in slots 6..13)
      if (prof && lane == 0 && iter >= LINS_PROF_WAVES) L.prof_acc[6 + wave] += t1 - t0;
#endif
      if (!(pad & 0x40000)) {  // no LDS, no barrier: the rows never leave registers
        const long long r0 = prof ? clock64() : 0;
        const double red = wave_reduce_rows(row, lane);
        if (prof) PROF2_ADD(4, clock64() - r0);
        acc += red;
      }
    }
    if (do_search) searched = true;
    {
      const int sidx28 = reduce_sum_index(lane);
      if (sidx28 >= 0) L.partial[part_k * 28 + sidx28] = acc;
    }
    if (ms) atomicAdd(&L.m_surf, ms);
    if (mc) atomicAdd(&L.m_corner, mc);
    const long long b0 = prof ? clock64() : 0;
    __syncthreads();
    const long long b1 = prof ? clock64() : 0;
    PROF2_ADD(5, b1 - b0);
    if (tid < 28) {
      double sacc = 0;
#pragma unroll
      for (int g = 0; g < BLOCK / 64; ++g) sacc += L.partial[g * 28 + tid];
      L.sums[tid] = sacc;
    }
    __syncthreads();
    if (prof) t2 = clock64();
    PROF2_ADD(6, t2 - b1);
    if (PASS_ONLY && (pad & 4) && !L.conv) {
      // test aid: repeat the pass with the warm start fed by the first one — a warm search at
      // the same state must return the very same triplets (exercises the tightest bounds)
      __syncthreads();
      if (tid == 0) L.conv = 1;
      continue;
    }
    if (PASS_ONLY) {
      if (ka.sums_out && tid < 28) ka.sums_out[(size_t)scan * 28 + tid] = L.sums[tid];
      if (ka.counts_out && tid == 0) ka.counts_out[scan * 2] = L.m_surf, ka.counts_out[scan * 2 + 1] = L.m_corner;
      return true;
    }

    if (ICP && held.w >= 0 && L.m_surf >= 10) idx_store[held.w] = make_int4(held.x, held.y, held.z, 0);
    if (ICP) {
      if (tid < 64) icp_solve_and_update(tid, iter);
      __syncthreads();
    } else
      t3 = solve_and_update(prm.r2, prm.fixed_iters, pad, tid, iter, prof);
    if (prof) {
      long long t4 = clock64();
      PROF2_ADD(7, t4 - t2);
      if (tid == 0) {
        L.prof_acc[1] += t1 - t0, L.prof_acc[2] += t2 - t1, L.prof_acc[3] += t3 - t2, L.prof_acc[4] += t4 - t3;
#ifndef LINS_PROF_WAVES
        if (iter < 3) L.prof_acc[10 + iter] = t4 - t0;
#endif
      }
    }
  }
  const ColdArgs kp = cold_args();  // (what only the epilogue needs is loaded here, not kept in SGPRs across the loop)
#define KP(f) kp->f
#ifdef LINS_PROF2
  if (prof && dbg_slot >= 0) idx_store[dbg_slot] = make_int4(dbg_nn, dbg_walk, dbg_walk_mask, ra1 | (a3 >= 0 ? 0x100 : 0));
#endif
  if (prof && tid == 0) {
    long long* prof_out = KP(prof_buf);
    L.prof_acc[5] = clock64() - t_begin;
    // residency probe: [13] = HW_ID | XCC_ID << 32, [14] / [15] = start / end on the 100 MHz wall clock
#ifndef LINS_PROF_WAVES
    L.prof_acc[13] = (long long)(unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                     ((long long)(unsigned)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
#endif
    L.prof_acc[14] = t_wall_begin, L.prof_acc[15] = wall_clock64();
    for (int k = 0; k < 16; ++k) prof_out[(size_t)scan * 16 + k] = L.prof_acc[k];
#ifdef LINS_PROF2
    int* ext = reinterpret_cast<int*>(prof_out + (size_t)gridDim.x * 16) + (size_t)scan * 64;  // (whole updates: one workgroup per scan)
    // (the reduction phase, ~230 ticks a wave, gives its slot to the search counts: [w][4] = wave-iterations with a
    // nearest-neighbour search | with a walk, both << 16 ... kept simple: phase 4 of wave w = walks << 16 | wave-iterations with a walk,
    // phase 4 is re-read by tools/wave_phases.py)
    for (int k = 0; k < 64; ++k) ext[k] = (k & 7) == 4 ? (L.prof3[(k >> 3) * 4 + 1] | (L.prof3[(k >> 3) * 4 + 3] << 12) | (L.prof3[(k >> 3) * 4 + 0] << 20) | (L.prof3[(k >> 3) * 4 + 2] << 26)) : L.prof2[k];
#endif
  }

  if (kRelay && relay_out) {  // the second part of this scan's update continues from here (see relay_in)
    double* h = KP(relay_hdr) + (size_t)scan * kRelayHdr;
    if (tid < 58) relay_st(h + tid, reinterpret_cast<const double*>(&L.ic)[tid]);
    if (tid == 64) relay_st(h + 58, L.res_prev), relay_st(h + 59, L.res_last), relay_st(h + 60, L.upd_norm);
    if (tid == 65) {
      int* hi = reinterpret_cast<int*>(h + 61);
      relay_st(hi, L.iter), relay_st(hi + 1, L.dbg[0]), relay_st(hi + 2, L.dbg[1]), relay_st(hi + 3, L.dbg[2]), relay_st(hi + 4, L.dbg[3]);
    }
    // every store of this wave has completed — write-through stores: at the memory side — before the barrier, the queue
    // slot after it: whoever pops the continuation sees the hand-over
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) relay_raise(KP(queue) + kQFlags + scan, KP(relay_gen) * 16 + part + 1);
    return false;
  }
  if (kRelay && relay_n > 0 && tid == 0)  // this part finished the scan: the later ones have nothing to do
    relay_raise(KP(queue) + kQFlags + scan, KP(relay_gen) * 16 + 15);

  // ---- hand-off to the Joseph kernel / the caller (SE:585-598) ---------------
  const int div = L.div;
  if (ICP) {  // filterState with rn_, qbn_ replaced (SE:590-592); the covariance is the caller's, un-updated
    if (tid < 19) KP(state_out)[(size_t)scan * 19 + tid] = (tid < 3 || (tid >= 6 && tid < 10)) ? L.ic.lin[tid] : L.filt[tid];
  } else {
    if (tid < 19) KP(state_out)[(size_t)scan * 19 + tid] = div ? L.filt[tid] : L.ic.lin[tid];
    if (KP(a6_out) && tid < 21) KP(a6_out)[(size_t)scan * 21 + tid] = L.sums[tid];
  }
  if (tid == 0) {
    OutRec r;
    r.residual_norm = L.res_last, r.update_norm = L.upd_norm;
    r.iters = L.iter, r.converged = L.conv, r.diverged = div;
    r.m_surf = L.m_surf, r.m_corner = L.m_corner;
    r.pad[0] = L.dbg[0], r.pad[1] = (pad & 32) ? L.dbg[3] : L.dbg[1], r.pad[2] = L.dbg[2];
    KP(out)[scan] = r;
  }
  if (KP(poses) && tid < 32) {
    lins_pose_record* pr = KP(poses) + scan;
    const double* st = div ? L.filt : L.ic.lin;
    if (tid < 19) pr->state[tid] = st[tid];
    if (tid == 19) pr->residual_norm = L.res_last;
    if (tid == 20) {
      pr->iters = L.iter, pr->converged = L.conv, pr->diverged = div;
      pr->m_surf = L.m_surf, pr->m_corner = L.m_corner, pr->scan_id = KP(scan_id_base) + scan;
      pr->pad[0] = pr->pad[1] = 0;
    }
  }
  if (!ICP && KP(cov_out)) joseph_epilogue<BLOCK>(KP(prm).r2, div, KP(cov_out) + (size_t)scan * 324, tid);
  return true;
}

// KNOBS: the counting aids and test modes of LINS_DEBUG_SKIP (DevParams::pad) compiled in.  The production instantiations
// of the update kernels are built without them — twenty-odd tests of a run-time word in the hottest code of a kernel that is
// short of scalar registers; the launchers take the KNOBS twin whenever pad != 0 (tools/, the certificate tests).
template <int BLOCK, int LANES, bool PASS_ONLY, bool PROF, bool ICP = false, bool KNOBS = true>
#if defined(LINS_LDS_NUMVGPR)
// (A/B aid: a register budget without the occupancy that goes with it — the compiler relaxes waves-per-SIMD to what the LDS allows)
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_num_vgpr(LINS_LDS_NUMVGPR))) void ieskf_lds_kernel(
#elif LINS_LDS_MINW > 1
// (second argument: waves per SIMD the register allocation must allow — the update kernels' budget; the correspondence pass
// and the ICP fallback, which run once per scan, take what their workgroup shape leaves them)
__global__ __launch_bounds__(BLOCK, (PASS_ONLY || ICP) ? 1 : LINS_LDS_MINW) void ieskf_lds_kernel(
#else
__global__ __launch_bounds__(BLOCK) void ieskf_lds_kernel(
#endif
    const KernelArgs ka, const float4* __restrict__ arena, const float4* __restrict__ sorted, int4* __restrict__ idx_store,
    lins_corr* __restrict__ dump) {
  // (the four pointers the loop reads and writes through stay parameters of their own: only a parameter carries
  // `noalias`, and without it the register allocation of every instantiation got worse)
  constexpr bool kQueue = kLeanBuild && BLOCK == kBatchBlock && LANES == 1 && !PASS_ONLY && !ICP;  // (the shape that has the relay)
  const bool queued = kQueue && ka.relay_n > 0;  // (uniform) a workgroup of a ticketed batch launch, see "work items" above
  int item;  // scan | part << 27; -1: nothing to do
#ifdef LINS_QUEUE_TRACE  // (debug builds, tools/queue_trace.py: per workgroup start / item in hand / end on the 100 MHz clock)
  const long long qt0 = wall_clock64();
  long long qt1 = qt0;
#endif
  // (Rounds 5 and 6 also ran the ticket draw as a LOOP — as many persistent workgroups as the device holds, each drawing
  // tickets until the list is exhausted, no workgroup turnover between items and no static share of the items per XCD
  // (-DLINS_PERSIST=1): round 5 0.5767 against 0.5636 ms with 15 spilled registers; round 6, on the register-lean
  // correspondence phase, 0.5754 against 0.5563 ms with 10 (fresh copies of the parameters inside the loop, so that nothing is
  // hoisted out of it: 38) — the loop around the update costs more than the turnover it saves.)
#if LINS_PERSIST
  // (round 6, with the register-lean correspondence phase: as many workgroups as the device holds, each drawing tickets
  // until the list is exhausted — no workgroup turnover between items)
#pragma unroll 1
  for (;;) {
#endif
  if (kQueue && queued) {
    if (threadIdx.x == 0) {
      int* const Q = ka.queue;
      const int ticket = relay_add(Q + kQHead, 1);
#if LINS_PERSIST
      if (ticket >= ka.relay_items) {
        g_lds.scan_tmp[0] = -2;
      } else {
#endif
      int it = ka.order[ticket];
      const int part = it >> 27, scan = it & 0x7FFFFFF;
      g_lds.scan_tmp[1] = ticket - part * ka.relay_n;  // the scan's place in the launch order (longest-expected-first)
      if (part) {
        const int want = ka.relay_gen * 16 + part;
        int f = relay_ld(Q + kQFlags + scan);
        for (int spins = 0; f < want; ++spins) {
          if (spins >= ka.relay_spins) {  // (cannot happen on a healthy device, see above: reported, not worked around)
            __hip_atomic_fetch_add(ka.relay_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
          }
          __builtin_amdgcn_s_sleep(32);
          f = relay_ld(Q + kQFlags + scan);
        }
        if (LINS_RELAY_RELEASE && f == want) f = relay_ld_acq(Q + kQFlags + scan);  // (the acquire that pairs with the raise)
        if (f != want) it = -1;  // the scan is finished (stop rule, divergence) — or the wait ran out
      }
      g_lds.scan_tmp[0] = it;
#if LINS_PERSIST
      }
#endif
    }
    __syncthreads();
    item = g_lds.scan_tmp[0];
#if LINS_PRIO_SHIFT > 0
    // The launch ends with the CHAIN of its slowest updates (tools/queue_trace.py: the parts of the scan with the largest
    // prior translation take 2.2-2.5 x the mean and follow one another); the scans the host expects to be the longest — the
    // first n >> LINS_PRIO_SHIFT of the launch order — run their waves at raised issue priority: their chain shortens at
    // the expense of the co-resident updates, which have slack.
    if (g_lds.scan_tmp[1] < (ka.relay_n >> LINS_PRIO_SHIFT)) __builtin_amdgcn_s_setprio(LINS_PRIO_LEVEL);
#endif
#ifdef LINS_QUEUE_TRACE
    qt1 = wall_clock64();
#endif
    __syncthreads();  // (scan_tmp is the update's from here)
  } else {
    // one workgroup per scan, in the host's launch order (lins_capi.hip launch_order: longest-expected-first, so that
    // the dispatcher, which hands workgroups out in index order as slots free up, ends the launch with the short ones)
    item = ka.order ? ka.order[blockIdx.x] : (int)blockIdx.x;
  }
#if LINS_PERSIST
  if (kQueue && queued && item == -2) break;
#endif
  if (item >= 0)
    ieskf_lds_update<BLOCK, LANES, PASS_ONLY, PROF, ICP, KNOBS>(ka, arena, sorted, idx_store, dump, item & 0x7FFFFFF, kQueue ? item >> 27 : 0);
#if LINS_PERSIST
  if (!(kQueue && queued)) break;
  __syncthreads();  // (the workgroup's LDS block is the next item's)
  }
#endif
  if (kQueue && queued && threadIdx.x == 0) {  // the last workgroup out leaves the counters at zero for the next launch
    int* const Q = cold_args()->queue;
#ifdef LINS_QUEUE_TRACE
    {
      long long* tr = reinterpret_cast<long long*>(Q + kQFlags + cold_args()->relay_cap) + 4 * (size_t)blockIdx.x;
      tr[0] = qt0, tr[1] = qt1, tr[2] = wall_clock64(), tr[3] = item;
    }
#endif
    if (relay_add(Q + kQExited, 1) == (int)gridDim.x - 1) relay_st(Q + kQHead, 0), relay_st(Q + kQExited, 0);
  }
}

#undef KP
#undef PROF2_ADD
}  // namespace LINS_LDS_NS
}  // namespace lins
