// frontend_kernels.hip — the feature front-end of StateEstimator on the device:
// what turns one segmented VLP-16 cloud (+ cloud_msgs/cloud_info) into the four feature clouds the
// IESKF update reads.  Restates, in this order,
//   undistortPcl        SE:619-654   relative-time tag of every point
//   calculateSmoothness SE:656-678   11-tap range stencil over the FLAT segmented-cloud index
//   markOccludedPoints  SE:680-713   occlusion / parallel-beam masks
//   extractFeatures     SE:719-827   per ring, 6 sectors: sort by curvature, greedy picks with
//                                    neighbour suppression, less-flat collection, VoxelGrid 0.2 m
// One 1024-thread workgroup per scan (16 waves = one wave per ring for the sequential greedy
// part).  The per-point stencils read the range / column arrays coalesced in index order — the
// organised cloud is ring-major; flags and columns of the whole scan live in LDS for the greedy
// picks, the per-sector sort is a 512-key and the per-ring VoxelGrid sort a 2048-key bitonic network
// per wave with the keys in registers (wave shuffles across lanes).
//
// Sort keys carry the point index as tie-break ((|diffRange| bits, index): the reference's
// std::sort leaves the order of equal curvatures unspecified; the host restatement uses the same
// total order), so the picks are reproducible and identical on both sides.

#include <hip/hip_runtime.h>

#include "../../include/lins_host.h"
#include "lins_math.h"

namespace lins {

#ifndef LINS_FE_D2_GROUP
#define LINS_FE_D2_GROUP 2  // the centroid pass: a wave takes the 64-position chunks of the less-flat cloud in this many groups of consecutive chunks
#endif
#ifndef LINS_FE_COMPACT_MAX
#define LINS_FE_COMPACT_MAX 64  // edge candidates of a sector up to which they are dealt one per lane
#endif
constexpr int kFeRows = LINS_LINE_NUM;
constexpr int kFeMaxN = LINS_CLOUD_MAX;  // 28 800 cells
constexpr int kFeBlock = 1024;
constexpr int kSectorCap = 512;          // >= points of one sector (a ring has <= 1800 -> <= 300 + margins)
constexpr int kRingCap = 2048;           // >= less-flat points of one ring
constexpr int kPickStride = 32;          // per sector: [0..1] sharp, [2..21] less sharp (incl. sharp), [22..25] flat, [26..28] counts

struct FeScan {  // device view of one lins_segmented_scan + its outputs
  long long off;     // first point in the point / range / col / ground arenas
  int n;
  int start_ring[kFeRows], end_ring[kFeRows];
  float start_ori, end_ori, ori_diff;
  int pad;
  long long o_sharp, o_less_sharp, o_flat, o_less_flat;  // where the four feature clouds go (points from `out`)
};

struct FeLds {
  union {
    struct {
      unsigned char flags[kFeMaxN + 16];  // bit 0 picked (cloudNeighborPicked); bits 1-2 cloudLabel: 0 = 0, 1 = 1 (less
                                          // sharp), 2 = 2 (sharp), 3 = -1 (flat); bit 3 ground
      unsigned short col[kFeMaxN + 16];
      unsigned long long skey[kFeRows][kSectorCap];  // per wave: 4 KB of work space (the less-flat stage's index lists; rounds 1-3: the sector sort's keys)
    } a;                                              // stencils, masks, sector picks
    unsigned short vso[kFeRows][kRingCap];            // VoxelGrid: per ring, sorted (run start << 15) | order
  };
  int first_half_end;      // first point with ori - startOri > pi (halfPassed flips after it)
  int ring_m[kFeRows];     // less-flat points of each ring
  int ring_base[kFeRows];  // the ring's first sector's start: what its kept-point list is relative to
  int ring_out[kFeRows];   // voxels (= output points) of each ring
  int ring_off[kFeRows + 1];
  int chunk_off[kFeRows + 1];  // D2: 64-position chunks of the rings' sorted lists, prefix sum
  int bad;
};
static_assert(sizeof(FeLds) <= 160 * 1024, "LDS budget");

__shared__ FeLds g_fe;
#ifdef LINS_FE_PROF
__device__ long long g_fe_prof[16 * 8];
#endif

__device__ __forceinline__ int fe_ordered_int(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float fe_ordered_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

template <int N>
struct FeInt {
  static constexpr int value = N;
};

__device__ __forceinline__ unsigned long long fe_shfl_xor(unsigned long long v, int m) {
  const unsigned lo = __shfl_xor((unsigned)v, m), hi = __shfl_xor((unsigned)(v >> 32), m);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned fe_shfl_xor(unsigned v, int m) { return __shfl_xor(v, m); }

// The largest (kMax) or smallest 64-bit key of a wave, in every lane, without the LDS crossbar: four DPP butterflies inside a
// row of 16 lanes (quad swaps, then the two mirrors), then the row swaps of gfx950 (v_permlane16_swap / v_permlane32_swap
// with both operands the same value: each lane sees its partner row's).  ~30 VALU instructions; the six ds_bpermute pairs
// it replaces each waited for an LDS round trip.
typedef unsigned fe_v2u __attribute__((ext_vector_type(2)));
template <bool kMax>
__device__ __forceinline__ unsigned long long fe_pick64(unsigned long long a, unsigned long long b) {
  return kMax ? (a > b ? a : b) : (a < b ? a : b);
}
template <int CTRL>
__device__ __forceinline__ unsigned long long fe_dpp64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)v, CTRL, 0xF, 0xF, true);
  const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(v >> 32), CTRL, 0xF, 0xF, true);
  return ((unsigned long long)hi << 32) | lo;
}
template <bool kMax>
__device__ __forceinline__ unsigned long long wave_best64(unsigned long long v) {
  v = fe_pick64<kMax>(v, fe_dpp64<0xB1>(v));   // quad_perm [1, 0, 3, 2]
  v = fe_pick64<kMax>(v, fe_dpp64<0x4E>(v));   // quad_perm [2, 3, 0, 1]
  v = fe_pick64<kMax>(v, fe_dpp64<0x141>(v));  // row_half_mirror
  v = fe_pick64<kMax>(v, fe_dpp64<0x140>(v));  // row_mirror
  {
    const fe_v2u l = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    const fe_v2u h = __builtin_amdgcn_permlane16_swap((unsigned)(v >> 32), (unsigned)(v >> 32), false, false);
    v = fe_pick64<kMax>(((unsigned long long)h.x << 32) | l.x, ((unsigned long long)h.y << 32) | l.y);
  }
  {
    const fe_v2u l = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    const fe_v2u h = __builtin_amdgcn_permlane32_swap((unsigned)(v >> 32), (unsigned)(v >> 32), false, false);
    v = fe_pick64<kMax>(((unsigned long long)h.x << 32) | l.x, ((unsigned long long)h.y << 32) | l.y);
  }
  return v;
}

// The largest 32-bit value of a wave, in every lane (the same moves as wave_best64, one v_max_u32 each).
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0u, v, 0x141, 0xF, 0xF, true));
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0u, v, 0x140, 0xF, 0xF, true));
  {
    const fe_v2u s = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = max(s.x, s.y);
  }
  {
    const fe_v2u s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v = max(s.x, s.y);
  }
  return v;
}

// v of lane (l ^ LM), LM a power of two below 64, as VALU moves: DPP inside a row of 16 (quad swaps; xor 4 as two
// bank-masked row shifts; xor 8 as a row rotation), the row swaps of gfx950 beyond.  (__shfl_xor is a ds_bpermute: an LDS
// instruction and a wait each — the 2048-key network of the VoxelGrid stage issues 672 of them per ring.)
template <int LM>
__device__ __forceinline__ unsigned fe_xor_lane(unsigned v, int lane) {
  if constexpr (LM == 1) {
    return __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xF, 0xF, true);
  } else if constexpr (LM == 2) {
    return __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xF, 0xF, true);
  } else if constexpr (LM == 4) {
    unsigned r = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);  // row_shl:4 into banks 0, 2: lane i <- i + 4
    r = __builtin_amdgcn_update_dpp(r, v, 0x114, 0xF, 0xA, false);           // row_shr:4 into banks 1, 3: lane i <- i - 4
    return r;
  } else if constexpr (LM == 8) {
    return __builtin_amdgcn_update_dpp(0u, v, 0x128, 0xF, 0xF, true);  // row_ror:8
  } else if constexpr (LM == 16) {
    const fe_v2u s = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // x: rows (0, 0, 2, 2), y: rows (1, 1, 3, 3)
    return (lane & 16) ? s.x : s.y;
  } else {
    static_assert(LM == 32, "lane distance");
    const fe_v2u s = __builtin_amdgcn_permlane32_swap(v, v, false, false);  // x: halves (lo, lo), y: (hi, hi)
    return (lane & 32) ? s.x : s.y;
  }
}
template <int LM>
__device__ __forceinline__ unsigned long long fe_xor_lane(unsigned long long v, int lane) {
  return ((unsigned long long)fe_xor_lane<LM>((unsigned)(v >> 32), lane) << 32) | fe_xor_lane<LM>((unsigned)v, lane);
}
// Bitonic sort of 64 * P keys by one wave, the keys in registers (lane l owns positions l P .. l P + P - 1):
// compare-exchanges whose partner lies inside the lane's own block are register selects, the others one lane-xor move
// per key (both lanes of a pair evaluate it and keep the min or the max) — no LDS round trips, no fences.  The network
// is unrolled by template recursion: every register index and every lane distance is a compile-time constant.
// The median of three: with c = 0 it is min(a, b), with c = ~0 max(a, b) — ONE instruction (v_med3_u32) where "keep the
// smaller or the larger, depending on the lane" was a minimum, a maximum and a select.  (Written as the min / max tree the
// back end matches to it.)
__device__ __forceinline__ unsigned fe_med3(unsigned a, unsigned b, unsigned c) {
  return max(min(a, b), min(max(a, b), c));
}
template <int P, int K2, int J2, class K>
__device__ __forceinline__ void bitonic_step(K (&v)[P], int lane) {
  if constexpr (sizeof(K) == 4 && J2 < P) {  // 32-bit keys, partner inside the lane: two medians per pair
#pragma unroll
    for (int u = 0; u < P; ++u)
      if ((u & J2) == 0) {
        const unsigned c = (((lane * P + u) & K2) == 0) ? 0u : ~0u;  // ascending: the smaller key first
        const unsigned a = (unsigned)v[u], b = (unsigned)v[u | J2];
        v[u] = (K)fe_med3(a, b, c), v[u | J2] = (K)fe_med3(a, b, ~c);
      }
  } else if constexpr (sizeof(K) == 4) {  // ... in another lane: the move and one median
    constexpr int kLm = J2 / P;
    const bool lower = (lane & kLm) == 0;
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const bool up = ((lane * P + u) & K2) == 0;
      const unsigned w = fe_xor_lane<kLm>((unsigned)v[u], lane);
      v[u] = (K)fe_med3(w, (unsigned)v[u], lower == up ? 0u : ~0u);
    }
  } else if constexpr (J2 < P) {
#pragma unroll
    for (int u = 0; u < P; ++u)
      if ((u & J2) == 0) {
        const bool up = ((lane * P + u) & K2) == 0;
        const K a = v[u], b = v[u | J2];
        const bool sw = (a > b) == up;
        v[u] = sw ? b : a, v[u | J2] = sw ? a : b;
      }
  } else {
    constexpr int kLm = J2 / P;
    const bool lower = (lane & kLm) == 0;
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const bool up = ((lane * P + u) & K2) == 0;
      const K w = fe_xor_lane<kLm>(v[u], lane);
      const bool keep_min = lower == up;
      v[u] = keep_min ? (w < v[u] ? w : v[u]) : (w > v[u] ? w : v[u]);
    }
  }
  if constexpr (J2 > 1) bitonic_step<P, K2, J2 / 2, K>(v, lane);
}
template <int P, int K2, class K>
__device__ __forceinline__ void bitonic_merge(K (&v)[P], int lane) {
  bitonic_step<P, K2, K2 / 2, K>(v, lane);
  if constexpr (K2 < 64 * P) bitonic_merge<P, K2 * 2, K>(v, lane);
}
template <int P, class K>
__device__ __forceinline__ void wave_bitonic_sort(K (&v)[P], int lane) {
  bitonic_merge<P, 2, K>(v, lane);
}

__global__ __launch_bounds__(kFeBlock) void frontend_kernel(
    const FeScan* __restrict__ scans, const float4* __restrict__ cloud, const float* __restrict__ range,
    const unsigned* __restrict__ col, const unsigned char* __restrict__ ground, double scan_period,
    int* __restrict__ picks, float4* __restrict__ out, int* __restrict__ out_counts) {
  FeLds& L = g_fe;
#ifdef LINS_FE_PROF
  long long fe_t0 = clock64(), fe_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define FE_MARK(id) \
  { long long t_ = clock64(); fe_t[id] = t_ - fe_t0; fe_t0 = t_; }
#else
#define FE_MARK(id)
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int scan = blockIdx.x;
  const FeScan& sc = scans[scan];  // (read where it is used: a private copy indexed by the ring lives in scratch)
  const int n = sc.n;
  const float4* pts = cloud + sc.off;
  const float* rg = range + sc.off;
  const unsigned* cl = col + sc.off;
  const unsigned char* gd = ground + sc.off;
  int* pk = picks + (size_t)scan * kFeRows * 6 * kPickStride;
  const double kPi = 3.14159265358979323846;

  if (tid == 0) L.first_half_end = n, L.bad = 0;
  // ---- load + undistortPcl, pass 1: where does halfPassed flip?  (SE:631-638: first-half adjustment) ----
  // Round 3: no per-point array is written at all.  The relative-time tag of a point is a function of the point, its
  // index and the flip position; the few consumers (the picked points, the kept points of the less-flat cloud: each
  // point at most once) evaluate it where they read the point — rounds 1-2 stored a de-skewed copy of the whole cloud.
  const double s_ori = (double)sc.start_ori, e_ori = (double)sc.end_ori;
  // Round 4: ONE pass over the cloud does what two did (where halfPassed flips; flags and columns into LDS; then, behind a
  // barrier, the occlusion / parallel-beam masks of SE:680-713 from a second read of the ranges): the masks need a
  // point's own column and its successor's — both read from global memory here — and their marks are 32-bit LDS atomic
  // ORs, as is the ground bit, so that nothing orders a point's own initialisation against its neighbours' marks: the
  // flag words are zeroed first (LDS only), then everything is an OR.
  unsigned* fw = reinterpret_cast<unsigned*>(L.a.flags);  // (marks are idempotent bit sets: 32-bit LDS atomics)
  auto mark = [&](int i) { atomicOr(&fw[i >> 2], 1u << ((i & 3) * 8)); };
  for (int w = tid; w < (n + 16 + 3) / 4; w += kFeBlock) fw[w] = 0u;
  if (tid < 16) L.a.col[n + tid] = 0;
  __syncthreads();
  {
    int first = n;
    constexpr int kIn = 4;  // points per thread whose reads are in flight together
    // (the trip count is WAVE-UNIFORM — the test is on the wave's first index, the points are predicated by i < n below —
    // so that the wave reduction at the end of the body runs with every lane present: with `i0 < n` the wave that straddles
    // n lost its upper lanes one trip early and the shuffles read switched-off lanes)
    for (int i0 = tid; i0 - lane < n; i0 += kIn * kFeBlock) {
      // Round 5: the flip is the FIRST index that passes pi — a point behind an index already found cannot be it, so its
      // coordinates are not read and its arctangent is not taken.  The organised cloud is ring-major and ring 0 sweeps
      // the whole turn: the first trip of this loop (indices < 4096) usually finds it, the other six skip 40 of their
      // ~80 instructions per point and 16 of its 33 bytes.  (A stale `seen` only costs work: the minimum only falls.)
      const int seen = __builtin_amdgcn_readfirstlane(*(volatile int*)&L.first_half_end);
      float4 p[kIn];
      unsigned char g[kIn];
      unsigned c[kIn], c1[kIn];
      float rm[kIn], r0[kIn], rp[kIn];
#pragma unroll
      for (int u = 0; u < kIn; ++u) {
        const int i = i0 + u * kFeBlock, ic = i < n ? i : n - 1;  // (clamped reads; an empty scan does not get here)
        const int im = ic > 0 ? ic - 1 : 0, ip = ic < n - 1 ? ic + 1 : n - 1;
        if (i < seen) p[u] = pts[ic];
        g[u] = gd[ic], c[u] = cl[ic], c1[u] = cl[ip];
        rm[u] = rg[im], r0[u] = rg[ic], rp[u] = rg[ip];
      }
#pragma unroll
      for (int u = 0; u < kIn; ++u) {
        const int i = i0 + u * kFeBlock;
        if (i < n) {
          if (i < seen) {
            double ori = (double)(-lins_atan2f(p[u].y, p[u].x));
            if (ori < s_ori - kPi / 2)
              ori += 2 * kPi;
            else if (ori > s_ori + kPi * 3 / 2)
              ori -= 2 * kPi;
            if (ori - s_ori > kPi && i < first) first = i;
          }
          if (g[u]) atomicOr(&fw[i >> 2], 8u << ((i & 3) * 8));
          L.a.col[i] = (unsigned short)c[u];
          if (i >= 5 && i < n - 6) {  // markOccludedPoints (SE:680-713)
            const float d1 = r0[u], d2 = rp[u];
            int cd = (int)c1[u] - (int)c[u];
            cd = cd < 0 ? -cd : cd;
            if (cd < 10) {
              if (d1 - d2 > 0.3) {
                for (int k = 0; k <= 5; ++k) mark(i - k);
              } else if (d2 - d1 > 0.3) {
                for (int k = 1; k <= 6; ++k) mark(i + k);
              }
            }
            const float f1 = fabsf(rm[u] - r0[u]), f2 = fabsf(rp[u] - r0[u]);
            if (f1 > 0.02 * r0[u] && f2 > 0.02 * r0[u]) mark(i);
          }
        }
      }
      if (__any(first < seen)) {  // (told at once: the waves still on their way stop taking arctangents)
        for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
        if (lane == 0) atomicMin(&L.first_half_end, first);
      }
    }
  }
  __syncthreads();
  const int flip = L.first_half_end;
  FE_MARK(1)
  const double ori_diff = (double)sc.ori_diff;
  auto tag_of = [&](int i, const float4& p) {  // undistortPcl's intensity (SE:639-650) of point i
    double ori = (double)(-lins_atan2f(p.y, p.x));
    if (i <= flip) {
      if (ori < s_ori - kPi / 2)
        ori += 2 * kPi;
      else if (ori > s_ori + kPi * 3 / 2)
        ori -= 2 * kPi;
    } else {
      ori += 2 * kPi;
      if (ori < e_ori - kPi * 3 / 2)
        ori += 2 * kPi;
      else if (ori > e_ori + kPi / 2)
        ori -= 2 * kPi;
    }
    const double rel = (ori - s_ori) / ori_diff;
    return (float)((double)(int)p.w + scan_period * rel);
  };
  // calculateSmoothness (SE:656-678) is evaluated where it is consumed — the sector sort below — from the range array
  // (f32, left to right, as written in SE:660-666); cloudCurvature / cloudSmoothness are never materialised.
  auto diff_at = [&](int i) {
    float d = 0.f;
    if (i >= 5 && i < n - 5)
      d = rg[i - 5] + rg[i - 4] + rg[i - 3] + rg[i - 2] + rg[i - 1] - rg[i] * 10 + rg[i + 1] + rg[i + 2] + rg[i + 3] +
          rg[i + 4] + rg[i + 5];
    return d;
  };
  FE_MARK(2)
  // ---- extractFeatures: one wave per ring, sectors in order (marks of one sector reach the next) ---
  {
    const int ring = __builtin_amdgcn_readfirstlane(wave);
#ifdef LINS_FE_PROF
    long long p3_t[5] = {0, 0, 0, 0, 0};
    const long long p3_begin = clock64();
#endif
    for (int j = 0; j < 6; ++j) {
      const int sp = (sc.start_ring[ring] * (6 - j) + sc.end_ring[ring] * j) / 6;
      const int ep = (sc.start_ring[ring] * (5 - j) + sc.end_ring[ring] * (j + 1)) / 6 - 1;
      int* spk = pk + (ring * 6 + j) * kPickStride;
      bool skip = sp >= ep || sp < 0 || ep >= n || ep - sp > kSectorCap - 1;
      if (skip) {
        if (lane < 3) spk[26 + lane] = 0;
        continue;
      }
      int pick_entry = 0;  // lane k < 29 carries entry k of the sector's pick list: one 116-byte store at the end
      const int m = ep - sp;  // [sp, ep) is visited in curvature order — ep itself keeps its place (SE:739-740)
      // cloudSmoothness[i].ind is i only where the stencil ran, [5, n - 5); elsewhere the value-initialised 0
      auto smooth_ind = [&](int i) { return (i >= 5 && i < n - 5) ? i : 0; };
      // Round 4: NO SORT.  The reference sorts the sector by curvature and walks the order from the top (edges) and from
      // the bottom (planes), taking the first candidate that is still eligible; flags only gain bits, so "the first
      // eligible candidate in descending order" is the eligible candidate with the LARGEST key — an arg-max over the
      // wave per pick (at most 20 + 4 picks a sector; 6.4 on the bench scans) instead of a 128 / 256 / 512-key bitonic
      // network per sector (rounds 1-3: ~1.7 k of the ~2.3 k instructions a sector cost, 168 LDS-crossbar shuffles).
      // Keys as before: (|diffRange| bits, index) — the index breaks the ties std::sort leaves open, the host restatement
      // sorts by the same total order.  Element e = u * 64 + lane of the sector lives in this lane's u-th register
      // (coalesced stencil reads); per lane a bit mask of its edge candidates (curvature > 0.5, not ground, not picked
      // when the sector begins) and of its plane candidates (curvature < 0.5, ground, not picked).  While the sector is
      // worked on, cloudNeighborPicked changes only through this wave's own picks — the picked index and the run of
      // neighbours it marks — so the masks are kept up to date in registers (drop_range) and a round reads no flag at
      // all: local best (registers), wave arg-max (VALU), the pick's column-gap test (the one LDS round trip), marks.
      constexpr int kPmax = kSectorCap / 64;
#ifdef LINS_FE_PROF
      long long p3a = clock64();
#endif
      unsigned dbits[kPmax];
      unsigned ecand = 0, pcand = 0;
#pragma unroll
      for (int u = 0; u < kPmax; ++u) {
        dbits[u] = 0;
        if (u * 64 < m) {  // (wave-uniform)
          const int e = u * 64 + lane;
          if (e < m) {
            const float d = fabsf(diff_at(sp + e));
            const double c = (double)d * (double)d;  // cloudCurvature (SE:668), compared as the reference compares it
            const unsigned char f = L.a.flags[smooth_ind(sp + e)];
            dbits[u] = __float_as_uint(d);
            ecand |= (c > 0.5 && !(f & 9)) ? 1u << u : 0u;
            pcand |= (c < 0.5 && (f & 8)) ? 1u << u : 0u;  // (picked or not is looked up when the edges are done)
          }
        }
      }
      const int ind_ep = smooth_ind(ep);
      const float d_ep = fabsf(diff_at(ep));
      const double c_ep = (double)d_ep * (double)d_ep;
#ifdef LINS_FE_PROF
      asm volatile("" ::"v"(ecand), "v"(pcand), "v"(d_ep) : "memory");
      long long p3b = clock64();
      p3_t[0] += p3b - p3a;
#endif
      {
        // Flags change by 32-bit LDS atomic ORs with no return value (`mark`, above) and are read as bytes: the LDS unit
        // takes a wave's instructions in order, so a flag read issued after an OR sees it — nothing to wait for; what
        // is needed is that the COMPILER keeps the order (it does for accesses that may alias; the barrier makes it plain).
        auto wave_sync = [&] {
          asm volatile("" ::: "memory");
          __builtin_amdgcn_wave_barrier();
        };
        auto set_bits = [&](int i, unsigned bits) { atomicOr(&fw[i >> 2], bits << ((i & 3) * 8)); };
        // A pick: its own flag (label, picked) and cloudNeighborPicked of the +-5 neighbours up to the first column gap > 10
        // (SE:764-779), in one step.  Lane t <= 10 stands for point ind - 5 + t: it reads that point's column, takes its
        // right neighbour's by a row shift, and bit t of one ballot says "gap between t and t + 1" (a point outside the
        // cloud is a gap: the reference's loops stop there); the first gap either side of t = 5 bounds the marked run, and
        // every lane of the run ORs its point's flag (the LDS unit takes a wave's instructions in order: a later flag read
        // sees it).  Round 5: one LDS read and one OR per lane, no branch (rounds 3-4: two reads per neighbour behind three
        // nested bounds tests, the pick's own flag as a separate instruction).
        // this lane's candidates whose point index lies in [lo, hi] are picked now: out of the mask.  Only the blocks
        // of 64 elements the range touches are looked at (wave-uniform test; a range that reaches index 0 — where the
        // elements outside the stencil's reach point, smooth_ind — looks at all).
        auto drop_range = [&](unsigned& cand, int lo, int hi) {
#pragma unroll
          for (int u = 0; u < kPmax; ++u) {
            if (u * 64 < m && (lo <= 0 || (sp + u * 64 <= hi && sp + u * 64 + 63 >= lo))) {
              const int ind = smooth_ind(sp + u * 64 + lane);
              cand &= (ind >= lo && ind <= hi) ? ~(1u << u) : ~0u;
            }
          }
        };
        int mk_lo = 0, mk_hi = -1;  // the run of indices the last pick marked (itself included)
        auto pick_and_mark = [&](int ind, unsigned own_bits, bool nbrs) {
          const int j = ind - 5 + lane;
          const bool in = lane <= 10 && j >= 0 && j < n;
          const int c = (int)L.a.col[in ? j : 0];
          const int cn = __builtin_amdgcn_update_dpp(0, c, 0x101, 0xF, 0xF, true);  // row_shl:1: lane t reads lane t + 1
          const int inn = __builtin_amdgcn_update_dpp(0, (int)in, 0x101, 0xF, 0xF, true);
          const int d = cn - c;
          const unsigned gm = (unsigned)__ballot(!in || !inn || (d < 0 ? -d : d) > 10) & 0x3FFu;
          const unsigned gf = gm >> 5, gb = gm & 0x1Fu;  // forward: bits 0..4 = neighbours +1..+5; backward: bit 4..0 = -1..-5
          int reach_f = gf ? __builtin_ctz(gf) : 5, reach_b = gb ? __builtin_clz(gb) - 27 : 5;  // neighbours marked per side
          if (!nbrs) reach_f = 0, reach_b = 0;
          const int t = lane - 5;
          if (lane <= 10 && t >= -reach_b && t <= reach_f) set_bits(j, t == 0 ? own_bits : 1u);
          mk_lo = ind - reach_b, mk_hi = ind + reach_f;
        };
        // the best key among this lane's candidates (kMax: largest, else smallest), then over the wave
        auto best_of = [&](unsigned cand, auto is_max) {
          constexpr bool kMax = decltype(is_max)::value != 0;
          unsigned long long best = kMax ? 0ull : ~0ull;
#pragma unroll
          for (int u = 0; u < kPmax; ++u) {
            if (u * 64 < m) {
              const unsigned long long k = ((unsigned long long)dbits[u] << 32) | (unsigned)smooth_ind(sp + u * 64 + lane);
              if ((cand >> u) & 1u) best = fe_pick64<kMax>(best, k);
            }
          }
          // (two steps as in the edge loop below: the 32-bit extreme of |diffRange| over the wave, its owner by ballot; the
          // 64-bit keys only when two lanes share it)
          const unsigned hi = (unsigned)(best >> 32);
          const unsigned ex = kMax ? wave_max_u32(hi) : ~wave_max_u32(~hi);
          const unsigned long long owners = __ballot(hi == ex);
          if ((owners & (owners - 1ull)) == 0ull)
            return ((unsigned long long)ex << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)best, __ffsll((long long)owners) - 1);
          return wave_best64<kMax>(hi == ex ? best : (kMax ? 0ull : ~0ull));
        };
        int n_sharp = 0, n_ls = 0, n_flat = 0;
        // edges: largest curvature first, at most 2 sharp + 18 less sharp (SE:743-780)
        auto edge_pick = [&](int pind) {
          // cloudLabel 2 / 1, picked: ORed in — an eligible edge candidate (not ground, not picked) carries no label yet
          pick_and_mark(pind, ((n_ls < 2 ? 2u : 1u) << 1) | 1u, true);
          pick_entry = ((n_ls < 2 && lane == n_sharp) || lane == 2 + n_ls) ? pind : pick_entry;
          n_sharp += n_ls < 2 ? 1 : 0;
          ++n_ls;
          wave_sync();
        };
        if (c_ep > 0.5 && !(L.a.flags[ind_ep] & 9)) {  // position ep comes first whatever its curvature
          edge_pick(ind_ep);
          drop_range(ecand, mk_lo, mk_hi);
        }
        // how many edge candidates the sector has (wave-uniform), and this lane's first one's rank among them
        int n_ec = 0, my_rank[kPmax];
#pragma unroll
        for (int u = 0; u < kPmax; ++u) {
          my_rank[u] = 0;
          if (u * 64 < m) {
            const unsigned long long bm = __ballot((ecand >> u) & 1u);
            my_rank[u] = n_ec + __popcll(bm & ((1ull << lane) - 1ull));
            n_ec += __popcll(bm);
          }
        }
        if (n_ec > 0 && n_ec <= LINS_FE_COMPACT_MAX) {
          // ONE candidate per lane (the usual case: a sector has a few dozen points of curvature > 0.5): through the
          // wave's 4 KB of LDS work space into lane order; a round is then the wave arg-max of one register, the pick, and
          // a three-instruction range test — the block masks are not touched again.
          unsigned long long* sk = L.a.skey[ring];
#pragma unroll
          for (int u = 0; u < kPmax; ++u)
            if (u * 64 < m && ((ecand >> u) & 1u)) sk[my_rank[u]] = ((unsigned long long)dbits[u] << 32) | (unsigned)smooth_ind(sp + u * 64 + lane);
          wave_sync();
          unsigned long long ck = lane < n_ec ? sk[lane] : 0ull;
          while (n_ls < 20) {  // the 21st eligible candidate would only end the loop (SE:757-759)
            // (Round 5: the arg-max in two steps — the largest |diffRange| as a 32-bit wave maximum, 14 instructions; its
            // owner by ballot.  Only when two candidates share it does the 64-bit key, 44 instructions, decide by index.)
            const unsigned hi = (unsigned)(ck >> 32);
            const unsigned mx = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_max_u32(hi));  // (every lane holds it: say so)
            if (!mx) break;  // (an edge candidate's key is > 0: its |diffRange| is)
            const unsigned long long owners = __ballot(hi == mx);
            int pind;
            if ((owners & (owners - 1ull)) == 0ull)
              pind = __builtin_amdgcn_readlane((int)(unsigned)ck, __ffsll((long long)owners) - 1);
            else
              pind = __builtin_amdgcn_readfirstlane((int)(unsigned)wave_best64<true>(hi == mx ? ck : 0ull));
            edge_pick(pind);
            const int ci = (int)(unsigned)ck;
            ck = (ci >= mk_lo && ci <= mk_hi) ? 0ull : ck;
          }
        } else if (n_ec > 0) {
          while (n_ls < 20) {
            const unsigned long long best = best_of(ecand, FeInt<1>{});
            if (!best) break;
            edge_pick((int)(unsigned)best);
            drop_range(ecand, mk_lo, mk_hi);
          }
        }
#ifdef LINS_FE_PROF
        long long p3c = clock64();
        p3_t[1] += p3c - p3b;
        p3_t[3] += n_ls;
#endif
        // planes: smallest curvature first, ground points only, at most 4; the 4th is not marked (SE:782-813)
        auto plane_pick = [&](int pind) {
          const bool last = n_flat + 1 >= 4;
          pick_and_mark(pind, (3u << 1) | (last ? 0u : 1u), !last);  // cloudLabel -1 (+ picked and the neighbours unless the 4th)
          pick_entry = lane == 22 + n_flat ? pind : pick_entry;
          ++n_flat;
          wave_sync();
        };
        if (__any(pcand != 0)) {
          {  // the plane candidates the edge picks (and everything before this sector) have left unpicked
            unsigned char fl[kPmax];
#pragma unroll
            for (int u = 0; u < kPmax; ++u) fl[u] = u * 64 < m ? L.a.flags[smooth_ind(sp + u * 64 + lane)] : (unsigned char)1;
#pragma unroll
            for (int u = 0; u < kPmax; ++u) pcand &= (fl[u] & 1) ? ~(1u << u) : ~0u;
          }
          while (n_flat < 4) {
            const unsigned long long best = best_of(pcand, FeInt<0>{});
            if (best == ~0ull) break;
            plane_pick((int)(unsigned)best);
            drop_range(pcand, mk_lo, mk_hi);
          }
        }
        if (n_flat < 4 && c_ep < 0.5) {  // position ep comes last
          const unsigned char f = L.a.flags[ind_ep];
          if ((f & 8) && !(f & 1)) plane_pick(ind_ep);
        }
#ifdef LINS_FE_PROF
        p3_t[2] += clock64() - p3c;
        p3_t[4] += n_flat;
#endif
        if (lane == 26) pick_entry = n_sharp;
        if (lane == 27) pick_entry = n_ls;
        if (lane == 28) pick_entry = n_flat;
        if (lane < 29) spk[lane] = pick_entry;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#ifdef LINS_FE_PROF
    if (lane == 0 && scan == 0) {
      long long* o = g_fe_prof + ring * 8;
      o[0] = p3_t[0], o[1] = p3_t[1], o[2] = p3_t[2], o[3] = p3_t[3], o[4] = p3_t[4], o[5] = sc.end_ring[ring] - sc.start_ring[ring], o[6] = clock64() - p3_begin;
    }
#endif
  }
  __threadfence_block();
  __syncthreads();

  FE_MARK(3)
  auto label_le0 = [&](int k) {  // cloudLabel <= 0: untouched (0) or flat (-1)
    const int b = (L.a.flags[k] >> 1) & 3;
    return b == 0 || b == 3;
  };

  // ---- feature clouds in the reference's order: rings, sectors, pick order -------------------------
  {
    int* cnt = reinterpret_cast<int*>(L.a.skey);  // [3][96] counts, then [3][96] exclusive offsets (sort buffers idle here)
    constexpr int kSec = kFeRows * 6;
    if (tid < kSec) {
      const int* spk = pk + tid * kPickStride;
      cnt[tid] = spk[26], cnt[kSec + tid] = spk[27], cnt[2 * kSec + tid] = spk[28];
    }
    __syncthreads();
    if (wave < 3) {  // exclusive offsets of the 96 sectors' counts: a wave per kind, two sectors per lane (48 lanes)
      static_assert(kSec <= 128, "two sectors per lane");
      const int a = 2 * lane < kSec ? cnt[wave * kSec + 2 * lane] : 0, b = 2 * lane + 1 < kSec ? cnt[wave * kSec + 2 * lane + 1] : 0;
      int incl = a + b;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int nb = __shfl_up(incl, o, 64);
        if (lane >= o) incl += nb;
      }
      if (2 * lane < kSec) cnt[(3 + wave) * kSec + 2 * lane] = incl - a - b;
      if (2 * lane + 1 < kSec) cnt[(3 + wave) * kSec + 2 * lane + 1] = incl - b;
      if (lane == 63) out_counts[scan * 4 + wave] = incl;
    }
    __syncthreads();
    auto und_pt = [&](int i) {  // the de-skewed point: coordinates as they came, the relative-time tag as intensity (SE:649-650)
      float4 q = pts[i];
      q.w = tag_of(i, q);
      return q;
    };
    for (int t = tid; t < kSec * 26; t += kFeBlock) {
      const int s2 = t / 26, k = t - s2 * 26;
      const int* spk = pk + s2 * kPickStride;
      if (k < 2) {
        if (k < spk[26]) out[sc.o_sharp + cnt[3 * kSec + s2] + k] = und_pt(spk[k]);
      } else if (k < 22) {
        if (k - 2 < spk[27]) out[sc.o_less_sharp + cnt[4 * kSec + s2] + (k - 2)] = und_pt(spk[k]);
      } else {
        if (k - 22 < spk[28]) out[sc.o_flat + cnt[5 * kSec + s2] + (k - 22)] = und_pt(spk[k]);
      }
    }
    __syncthreads();
  }

  FE_MARK(4)
  // ---- less-flat cloud: per ring, every point of its sectors with label <= 0 (SE:815-820) ... -----
  // D0, one wave per ring: the kept points' positions (relative to the ring's first sector) as a compact u16 list in
  // LDS — the sector sort buffers are idle.  Round 3: the points themselves are not touched here (rounds 1-2 copied
  // the kept points to a compact global list and read that list back twice).
  float4* olf = out + sc.o_less_flat;
  unsigned short* const kept = reinterpret_cast<unsigned short*>(L.a.skey[wave]);  // [kRingCap] of this ring
  static_assert(sizeof(L.a.skey[0]) >= kRingCap * sizeof(unsigned short), "index list of a ring");
  {
    const int ring = __builtin_amdgcn_readfirstlane(wave);
    int m = 0, base = -1;
    for (int j = 0; j < 6; ++j) {
      const int sp = (sc.start_ring[ring] * (6 - j) + sc.end_ring[ring] * j) / 6;
      const int ep = (sc.start_ring[ring] * (5 - j) + sc.end_ring[ring] * (j + 1)) / 6 - 1;
      if (sp >= ep || sp < 0 || ep >= n || ep - sp > kSectorCap - 1) continue;
      if (base < 0) base = sp;
      for (int c0 = sp; c0 <= ep; c0 += 64) {
        const int k = c0 + lane;
        const bool keep = k <= ep && label_le0(k);
        const unsigned long long mask = __ballot(keep);
        if (keep) {
          const int pos = m + __popcll(mask & ((1ull << lane) - 1ull));
          if (pos < kRingCap) kept[pos] = (unsigned short)(k - base);
        }
        m += __popcll(mask);
      }
    }
    if (lane == 0) {
      L.ring_m[ring] = m, L.ring_base[ring] = base < 0 ? 0 : base;
      if (m > kRingCap || (m > 0 && sc.end_ring[ring] - base >= 2048)) L.bad = 1;  // cannot happen for a 16 x 1800 sensor; refuse rather than truncate silently
    }
  }
  __threadfence_block();
  __syncthreads();  // (every reader of the flags is done: the voxel orders below overlay them)
  if (L.bad) {
    if (tid == 0) out_counts[scan * 4 + 3] = -1;
    return;
  }
  FE_MARK(5)
  // D1, pcl::VoxelGrid 0.2 m with all-field averaging, output ordered by voxel index (SE:189, 822-825):
  // one wave per ring; 2048 keys (voxel index << 11 | order) sorted in registers, then only the order
  // and a run-start bit per sorted position go to LDS.  Sorted position e belongs to lane e / 32.
  // The kept points are read ONCE here: each lane packs the voxel coordinates of its points (11 + 11 + 10 bits around
  // zero: +-204 m / +-102 m) into one register per point, the wave folds the bounding box VoxelGrid needs
  // (getMinMax3D), and the keys are built from the packed coordinates.  A ring that does not pack (never a VLP-16's)
  // takes the same route with a second read of its points.
  // Two phases: every ring sorts and counts its voxels first; after one block barrier the rings' output
  // offsets are known and the centroids go straight to their final place.
  {
    {
      const int ring = __builtin_amdgcn_readfirstlane(wave);
      const int m = L.ring_m[ring], base = L.ring_base[ring];
      unsigned short* vs = L.vso[wave];
      int nvox = 0;
      if (m > 0) {
        const float inv = 1.0f / 0.2f;
        auto voxel_grid = [&](auto tag) {
          constexpr int kP = decltype(tag)::value;
          int mine = 0;
          unsigned pkd[kP];  // (iz + 512) << 22 | (iy + 1024) << 11 | (ix + 1024); ~0u: no point
          int mnx = 0x7FFFFFFF, mny = 0x7FFFFFFF, mnz = 0x7FFFFFFF, mxx = (int)0x80000000, mxy = (int)0x80000000, mxz = (int)0x80000000;
          bool fits = true;
#pragma unroll
          for (int u = 0; u < kP; ++u) {
            const int e = u * 64 + lane;  // (unsorted: any placement will do — this one reads the points coalesced)
            pkd[u] = ~0u;
            if ((u & 7) == 0) asm volatile("" ::: "memory");  // (eight point reads in flight, not all kP: registers)
            if (e < m) {
              const float4 p = pts[base + (int)kept[e]];
              const int ix = (int)floorf(p.x * inv), iy = (int)floorf(p.y * inv), iz = (int)floorf(p.z * inv);
              mnx = min(mnx, ix), mny = min(mny, iy), mnz = min(mnz, iz);
              mxx = max(mxx, ix), mxy = max(mxy, iy), mxz = max(mxz, iz);
              fits = fits && ix >= -1024 && ix < 1024 && iy >= -1024 && iy < 1024 && iz >= -512 && iz < 511;
              pkd[u] = ((unsigned)(iz + 512) << 22) | ((unsigned)(iy + 1024) << 11) | (unsigned)(ix + 1024);
            }
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) {
            mnx = min(mnx, __shfl_xor(mnx, o)), mny = min(mny, __shfl_xor(mny, o)), mnz = min(mnz, __shfl_xor(mnz, o));
            mxx = max(mxx, __shfl_xor(mxx, o)), mxy = max(mxy, __shfl_xor(mxy, o)), mxz = max(mxz, __shfl_xor(mxz, o));
          }
          const bool packed = __all(fits);
          const long long dx = (long long)mxx - mnx + 1, dy = (long long)mxy - mny + 1, dz = (long long)mxz - mnz + 1;
          const bool narrow = dx * dy * dz < (1ll << 21) - 1;  // every key below the 32-bit padding value
          // PCL's linear voxel index of this lane's u-th point: from the packed coordinates (the common case: 32-bit
          // arithmetic throughout), or — a ring whose box needs 64-bit keys or whose coordinates do not pack — from a
          // second read of the point
          unsigned startmask = 0;  // bit u: this lane's u-th sorted position starts a voxel
          auto sort_and_mark = [&](auto key_zero, auto from_pack) {
            using K = decltype(key_zero);  // 32-bit keys when (voxel index << 11 | order) fits: half the shuffles and compares
            K kv[kP];
#pragma unroll
            for (int u = 0; u < kP; ++u) {
              const int e = u * 64 + lane;
              K k = (K)~key_zero;
              if (e < m) {
                if (decltype(from_pack)::value) {
                  const int ix = (int)(pkd[u] & 2047u) - 1024, iy = (int)((pkd[u] >> 11) & 2047u) - 1024, iz = (int)(pkd[u] >> 22) - 512;
                  k = (K)(((unsigned)((ix - mnx) + (iy - mny) * (int)dx + (iz - mnz) * (int)(dx * dy)) << 11) | (unsigned)e);
                } else {
                  const float4 p = pts[base + (int)kept[e]];
                  const long long ix = (long long)floorf(p.x * inv), iy = (long long)floorf(p.y * inv), iz = (long long)floorf(p.z * inv);
                  k = (K)(((unsigned long long)((ix - mnx) + (iy - mny) * dx + (iz - mnz) * dx * dy) << 11) | (unsigned)e);
                }
              }
              kv[u] = k;
            }
            wave_bitonic_sort<kP, K>(kv, lane);
            // run starts: the voxel index differs from the predecessor's (the previous lane's last key for u = 0)
            const unsigned plo = __shfl_up((unsigned)kv[kP - 1], 1), phi = __shfl_up((unsigned)((unsigned long long)kv[kP - 1] >> 32), 1);
            K prev = (K)(((unsigned long long)phi << 32) | plo);
#pragma unroll
            for (int u = 0; u < kP; ++u) {
              const int e = lane * kP + u;
              const bool start = e < m && (e == 0 || (prev >> 11) != (kv[u] >> 11));
              // sorted position e -> (run start, the point's position relative to the ring's base): D2 needs no second lookup
              vs[e] = (unsigned short)((start ? 0x8000u : 0u) | (e < m ? (unsigned)kept[kv[u] & 2047u] : 0u));
              mine += start ? 1 : 0;
              startmask |= start ? (1u << u) : 0u;
              prev = kv[u];
            }
          };

          if (narrow && packed)
            sort_and_mark(0u, FeInt<1>{});
          else if (narrow)
            sort_and_mark(0u, FeInt<0>{});
          else
            sort_and_mark(0ull, FeInt<0>{});
          int incl = mine;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const int nb = __shfl_up(incl, o, 64);
            if (lane >= o) incl += nb;
          }
          nvox = __shfl(incl, 63);
          // the list of kept points has been folded into vs: its place now takes, for every run start, the voxel's
          // position in the ring's output (every lane of the wave has finished reading the list)
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          int run = incl - mine;
#pragma unroll
          for (int u = 0; u < kP; ++u)
            if ((startmask >> u) & 1u) kept[lane * kP + u] = (unsigned short)run++;
        };
        static_assert(kRingCap == 2048, "sort sizes below");
        if (m <= 512)
          voxel_grid(FeInt<8>{});
        else if (m <= 1024)
          voxel_grid(FeInt<16>{});
        else
          voxel_grid(FeInt<32>{});
      }
      if (lane == 0) L.ring_out[ring] = nvox;
    }
    __threadfence_block();
    __syncthreads();
  }
  FE_MARK(6)
  if (tid == 0) {
    int run = 0;
    for (int r = 0; r < kFeRows; ++r) L.ring_off[r] = run, run += L.ring_out[r];
    L.ring_off[kFeRows] = run;
    out_counts[scan * 4 + 3] = run;
    int ch = 0;
    for (int r = 0; r < kFeRows; ++r) L.chunk_off[r] = ch, ch += (L.ring_m[r] + 63) >> 6;
    L.chunk_off[kFeRows] = ch;
  }
  __syncthreads();
  // D2: one centroid per run start — f32 sums of all four fields in stable (original) order — at the ring's final
  // offset + the voxel's position in the ring.  The whole workgroup takes the rings one after the other (round 3; one
  // wave per ring before): a thread per sorted position, so the gathers of all sixteen waves fall into ONE ring's points
  // at a time (they stay in the caches between the first touch of a line and the last) and neighbouring threads write
  // neighbouring centroids.
  // (Round 4: the 64-position chunks of all rings are dealt to the waves as ONE list — ring after ring, a ring's last
  // chunks together with the next ring's first — instead of a pass of the workgroup per ring, whose second step kept
  // 1024 threads for the ~200 positions a ring has beyond the first 1024: 32 steps a wave -> ~19.)
#ifdef LINS_FE_PROF
  long long d2_t[4] = {0, 0, 0, 0};
#endif
  // (Round 5: a wave takes kD2Group CONSECUTIVE chunks at a time and carries the partial sums of a voxel that goes on beyond
  // a chunk's last lane into the next chunk — lane 0 continues it — instead of finishing it alone, one dependent gather and
  // one arctangent per point: with ~5 points a voxel nearly every chunk ended in such a run, 37 k of the phase's 87 k
  // clocks.  Only a group's last chunk still finishes its last run alone.)
  const int n_chunks = L.chunk_off[kFeRows];
  const int kD2Group = max(1, (n_chunks + (kFeBlock / 64) * LINS_FE_D2_GROUP - 1) / ((kFeBlock / 64) * LINS_FE_D2_GROUP));
  for (int g0 = wave * kD2Group; g0 < n_chunks; g0 += (kFeBlock / 64) * kD2Group) {
    float cx = 0.f, cy = 0.f, cz = 0.f, ci = 0.f;  // the carried run (wave-uniform): sums so far, points so far, output slot
    int c_cnt = 0, c_slot = -1;
    for (int ch = g0; ch < min(g0 + kD2Group, n_chunks); ++ch) {
      // the chunk's ring: lane r < 16 tests ring r's first chunk, the ballot counts (chunk_off is non-decreasing)
      const int ring = __popcll(__ballot(lane >= 1 && lane < kFeRows && L.chunk_off[lane] <= ch));
      const int m = L.ring_m[ring], base = L.ring_base[ring];
      const unsigned short* vs = L.vso[ring];
      const unsigned short* slot = reinterpret_cast<const unsigned short*>(L.a.skey[ring]);
      float4* dst = olf + L.ring_off[ring];
      // A wave takes 64 consecutive sorted positions per step: every lane reads ITS point and forms its tag (one gather
      // and one arctangent per lane, no divergence); a voxel's lanes are neighbours, so its first lane collects the
      // others' values left to right with shuffles — the sums in the order VoxelGrid adds them (ascending original
      // index).  (Issuing the next step's gather before working on the current one was measured: no faster — the other
      // waves cover the latency.)
      const int e0 = (ch - L.chunk_off[ring]) * 64;
      const int e = e0 + lane;
      const bool valid = e < m;
      const unsigned v = valid ? (unsigned)vs[e] : 0x8000u;
      const bool start = valid && (v & 0x8000u);
      const bool head = lane == 0 && c_slot >= 0 && !start;  // (a carried run is of this ring: it crossed a chunk end inside it)
      const bool st = start || head;
      const int i = base + (int)(v & 2047u);
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
      float tg = 0.f;
#ifdef LINS_FE_PROF
      long long d2a = clock64();
#endif
      if (valid) p = pts[i];
#ifdef LINS_FE_PROF
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      long long d2b = clock64();
      d2_t[0] += d2b - d2a;
#endif
      if (valid) tg = tag_of(i, p);
#ifdef LINS_FE_PROF
      asm volatile("" ::"v"(tg) : "memory");
      long long d2c = clock64();
      d2_t[1] += d2c - d2b;
#endif
      const unsigned long long bounds = __ballot(start || !valid);  // where a run ends: the next start, or the end of the ring
      const unsigned long long above = lane < 63 ? bounds & ~((2ull << lane) - 1ull) : 0ull;
      const int nxt = above ? __ffsll((long long)above) - 1 : 64;
      const int len_in = nxt - lane;  // this run's points inside the step (meaningful for start lanes)
      float sx = (head ? cx : 0.f) + p.x, sy = (head ? cy : 0.f) + p.y, sz = (head ? cz : 0.f) + p.z, si = (head ? ci : 0.f) + tg;
      for (int d = 1; __any(st && d < len_in); ++d) {
        const float ax = __shfl_down(p.x, d), ay = __shfl_down(p.y, d), az = __shfl_down(p.z, d), at = __shfl_down(tg, d);
        if (st && d < len_in) sx += ax, sy += ay, sz += az, si += at;
      }
#ifdef LINS_FE_PROF
      asm volatile("" ::"v"(sx) : "memory");
      long long d2d = clock64();
      d2_t[2] += d2d - d2c;
      d2_t[3] += 1;
#endif
      int total = len_in + (head ? c_cnt : 0);
      const int my_slot = head ? c_slot : (valid ? (int)slot[e] : 0);
      // does the run that reaches the last lane go on in the next chunk?  (wave-uniform)
      const unsigned long long stm = __ballot(st);
      const bool goes_on = stm != 0ull && e0 + 64 < m && !(vs[e0 + 64] & 0x8000u);
      const int last_st = stm ? 63 - __builtin_clzll(stm) : 0;  // its first lane (the run above it reaches lane 63: valid lanes throughout)
      const bool carry = goes_on && ch + 1 < min(g0 + kD2Group, n_chunks);
      if (st && !(goes_on && lane == last_st)) {
        const float cnt = (float)total;
        dst[my_slot] = make_float4(sx / cnt, sy / cnt, sz / cnt, si / cnt);
      }
      if (carry) {
        cx = __shfl(sx, last_st), cy = __shfl(sy, last_st), cz = __shfl(sz, last_st), ci = __shfl(si, last_st);
        c_cnt = __shfl(total, last_st), c_slot = __shfl(my_slot, last_st);
      } else {
        c_slot = -1;
        if (goes_on && lane == last_st) {  // a group's last chunk: its last run is finished by its first lane alone
          int j = e0 + 64;
          while (j < m && !(vs[j] & 0x8000u)) {
            const int i2 = base + (int)(vs[j] & 2047u);
            const float4 p2 = pts[i2];
            sx += p2.x, sy += p2.y, sz += p2.z, si += tag_of(i2, p2);
            ++j;
          }
          total += j - (e0 + 64);
          const float cnt = (float)total;
          dst[my_slot] = make_float4(sx / cnt, sy / cnt, sz / cnt, si / cnt);
        }
      }
    }
  }
#ifdef LINS_FE_PROF
  FE_MARK(7)
  if (tid < 16 && scan == 0)
    printf("FE scan 0 ring %d: keys %lld edges %lld (%lld picks) planes %lld (%lld picks), n %lld, whole %lld\n", tid, g_fe_prof[tid * 8], g_fe_prof[tid * 8 + 1], g_fe_prof[tid * 8 + 3],
           g_fe_prof[tid * 8 + 2], g_fe_prof[tid * 8 + 4], g_fe_prof[tid * 8 + 5], g_fe_prof[tid * 8 + 6]);
  if (tid == 0 && (scan & 255) == 0)
    printf("FE scan %d: load %lld flip %lld tags+stencil+masks %lld sort+picks %lld labels %lld compact %lld voxel grid %lld gaps %lld | D2 wave 0: gather %lld tag %lld sums %lld chunks %lld\n", scan, fe_t[0],
           fe_t[1], fe_t[2], fe_t[3], fe_t[4], fe_t[5], fe_t[6], fe_t[7], d2_t[0], d2_t[1], d2_t[2], d2_t[3]);
#endif
}

void launch_frontend(hipStream_t stream, int n_scans, const void* scans, const float4* cloud, const float* range,
                     const unsigned* col, const unsigned char* ground, double scan_period, int* picks, float4* out,
                     int* out_counts) {
  hipLaunchKernelGGL(frontend_kernel, dim3(n_scans), dim3(kFeBlock), 0, stream, (const FeScan*)scans, cloud, range, col,
                     ground, scan_period, picks, out, out_counts);
}
size_t fe_scan_size() { return sizeof(FeScan); }
int fe_pick_stride() { return kFeRows * 6 * kPickStride; }

}  // namespace lins
