// frontend_kernels.hip — the feature front-end of StateEstimator on the device:
// what turns one segmented VLP-16 cloud (+ cloud_msgs/cloud_info) into the four feature clouds the
// IESKF update reads.  Restates, in this order,
//   undistortPcl        SE:619-654   relative-time tag of every point
//   calculateSmoothness SE:656-678   11-tap range stencil over the FLAT segmented-cloud index
//   markOccludedPoints  SE:680-713   occlusion / parallel-beam masks
//   extractFeatures     SE:719-827   per ring, 6 sectors: sort by curvature, greedy picks with
//                                    neighbour suppression, less-flat collection, VoxelGrid 0.2 m
// Round 5: TWO kernels of (scan, ring) workgroups instead of one 1024-thread workgroup per scan (rounds 1-4).
//   fe_ring_kernel   one WAVE per (scan, ring), ~10 KB of LDS: the ring's masks, its six sectors' picks, its less-flat
//                    points' voxel order.  Nothing in it looks at another ring (the masks of a ring's first and last
//                    points come from a halo of 6 points either side, read from global memory), so there is no barrier
//                    and sixteen rings of whatever scans share a CU, each in its own phase: the point reads of one
//                    overlap the pick rounds of another (the one-workgroup kernel ran every CU of the device through
//                    the same phase at the same time — HBM-bound passes followed by passes that left it idle).
//   fe_out_kernel    256 threads per (scan, ring): offsets from the rings' counts, the three picked clouds, the less-flat
//                    cloud's centroids; the relative-time tags are formed here, where a point is written.
// The per-point stencils read the range / column arrays coalesced in index order — the organised cloud is ring-major.
//
// Sort keys carry the point index as tie-break ((|diffRange| bits, index): the reference's
// std::sort leaves the order of equal curvatures unspecified; the host restatement uses the same
// total order), so the picks are reproducible and identical on both sides.

#include <hip/hip_runtime.h>

#include <cstdlib>

#include "../../include/lins_host.h"
#include "lins_math.h"

namespace lins {

#ifndef LINS_FE_D2_GROUP
#define LINS_FE_D2_GROUP 1  // the centroid pass: a wave takes a ring's 64-position chunks in this many groups of consecutive chunks
#endif
#ifndef LINS_FE_COMPACT_MAX
#define LINS_FE_COMPACT_MAX 64  // edge candidates of a sector up to which they are dealt one per lane
#endif
constexpr int kFeRows = LINS_LINE_NUM;
constexpr int kOutBlock = 256;           // fe_out_kernel: four waves per (scan, ring)
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

constexpr int kHalo = 6;         // how far a point's occlusion mark reaches (SE:693-704)
constexpr int kRingWin = 1856;   // a ring's window in LDS: its <= 1800 cells + the 4 + 6 points cloud_info's ring indices leave out + 2 halos + alignment
struct FeRingLds {               // fe_ring_kernel: one wave, one ring
  union {
    struct {
      unsigned flagw[kRingWin / 4];  // a byte per point of the window: bit 0 picked (cloudNeighborPicked); bits 1-2 cloudLabel: 0 = 0,
                                     // 1 = 1 (less sharp), 2 = 2 (sharp), 3 = -1 (flat); bit 3 ground
      unsigned short col[kRingWin];
    } a;                             // masks, sector picks, less-flat collection
    unsigned hk[1024];               // VoxelGrid: the run heads' keys in order of appearance
    unsigned short vs[kRingCap];     // VoxelGrid: sorted (voxel start << 15) | point position
  };
  unsigned long long work[kSectorCap];          // 4 KB: a sector's edge candidates; the kept points' positions (u16[kRingCap]); the voxels' output slots
  unsigned long long headbits[kRingCap / 64];   // VoxelGrid: bit e: kept point e opens a run of consecutive points of one voxel
};
static_assert(sizeof(FeRingLds) <= 10240, "sixteen rings per CU");
static_assert(sizeof(unsigned long long) * kSectorCap >= kRingCap * sizeof(unsigned short), "index list of a ring");

struct FeRingInfo {  // what fe_ring_kernel leaves per (scan, ring) for fe_out_kernel
  int m;             // less-flat points of the ring before VoxelGrid (-1: a ring beyond the kernel's limits)
  int base;          // the ring's first sector's start: positions in `order` are relative to it; the list sits at order[off + base ..]
  int nvox;          // voxels = less-flat output points of the ring
  int flip;          // (ring 0) first point with ori - startOri > pi (halfPassed flips after it), n if none
  int n_sharp, n_less_sharp, n_flat, pad;  // the ring's six sectors' picks
};

__shared__ FeRingLds g_ring;
template <class T>
struct FeWin {  // a window array addressed by the cloud's flat index
  T* p;
  int first;
  __device__ __forceinline__ T& operator[](int i) const { return p[i - first]; }
};

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
// per key (both lanes of a pair evaluate it and keep the min or the max) — no LDS round trips, no fences.
// Round 5: the network is a LOOP over its merge levels (rounds 3-4 unrolled all of it by template recursion: 66 steps of
// 32 keys are 80 KB of straight-line code per instantiation, half a megabyte for the kernel).  With one wave per ring
// every wave of a CU is somewhere else in the code, and straight-line code is fetched once per wave, not once per CU:
// a level's body — the six possible lane distances behind a switch, the in-lane steps unrolled — is a few KB that stay
// in the instruction cache.
template <int P, int LM, class K>
__device__ __forceinline__ void bitonic_cross(K (&v)[P], int lane, bool up) {
  const bool keep_min = ((lane & LM) == 0) == up;
#pragma unroll
  for (int u = 0; u < P; ++u) {
    const K w = fe_xor_lane<LM>(v[u], lane);
    v[u] = keep_min ? (w < v[u] ? w : v[u]) : (w > v[u] ? w : v[u]);
  }
}
template <int P, int J2, class K>
__device__ __forceinline__ void bitonic_inlane(K (&v)[P], int lane, int k2) {
  if constexpr (J2 >= 1) {
    if (J2 < k2) {  // (wave-uniform)
#pragma unroll
      for (int u = 0; u < P; ++u)
        if ((u & J2) == 0) {
          const bool up = (((lane * P) | u) & k2) == 0;
          const K a = v[u], b = v[u | J2];
          const bool sw = (a > b) == up;
          v[u] = sw ? b : a, v[u | J2] = sw ? a : b;
        }
    }
    bitonic_inlane<P, J2 / 2, K>(v, lane, k2);
  }
}
// (One copy per size and key type in the binary, called from wherever a sort is needed: the arguments travel through
// registers, what is live across the call is saved around it — once per ring.)
template <int P, class K>
__device__ __noinline__ void wave_bitonic_sort_mem(K (&v)[P], int lane);
template <int P, class K>
__device__ __forceinline__ void wave_bitonic_sort(K (&v)[P], int lane) {  // (the caller's keys stay registers: a copy goes through memory)
  K t[P];
#pragma unroll
  for (int u = 0; u < P; ++u) t[u] = v[u];
  wave_bitonic_sort_mem<P, K>(t, lane);
#pragma unroll
  for (int u = 0; u < P; ++u) v[u] = t[u];
}
template <int P, class K>
__device__ __noinline__ void wave_bitonic_sort_mem(K (&v)[P], int lane) {
#pragma unroll 1
  for (int k2 = 2; k2 <= 64 * P; k2 <<= 1) {
    const bool up = ((lane * P) & k2) == 0;  // (for the lane-crossing steps k2 >= 2 P: the direction is the lane's)
#pragma unroll 1
    for (int lm = (k2 >> 1) / P; lm >= 1; lm >>= 1) {
      switch (lm) {
        case 32: bitonic_cross<P, 32, K>(v, lane, up); break;
        case 16: bitonic_cross<P, 16, K>(v, lane, up); break;
        case 8: bitonic_cross<P, 8, K>(v, lane, up); break;
        case 4: bitonic_cross<P, 4, K>(v, lane, up); break;
        case 2: bitonic_cross<P, 2, K>(v, lane, up); break;
        default: bitonic_cross<P, 1, K>(v, lane, up); break;
      }
    }
    bitonic_inlane<P, P / 2, K>(v, lane, k2);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// fe_ring_kernel: one wave per (scan, ring).  Leaves per sector its pick list (`picks`, 32 ints), per ring its counts
// (FeRingInfo) and the voxel order of its less-flat points: order[off + base + e], e < m, = (voxel's slot in the ring's
// output << 16) | (voxel start << 15) | the point's position relative to base — what the centroid pass walks.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void fe_ring_kernel(
    const FeScan* __restrict__ scans, const float4* __restrict__ cloud, const float* __restrict__ range,
    const unsigned* __restrict__ col, const unsigned char* __restrict__ ground, int* __restrict__ picks,
    unsigned* __restrict__ order, FeRingInfo* __restrict__ info) {
  FeRingLds& L = g_ring;
  const int lane = threadIdx.x;
  const int scan = blockIdx.x / kFeRows, ring = blockIdx.x % kFeRows;
#ifdef LINS_FE_PROF
  long long fe_t0 = clock64(), fe_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define FE_MARK(id) \
  { long long t_ = clock64(); fe_t[id] += t_ - fe_t0; fe_t0 = t_; }
#else
#define FE_MARK(id)
#endif
  const FeScan& sc = scans[scan];
  const int n = sc.n;
  const float4* pts = cloud + sc.off;
  const float* rg = range + sc.off;
  const unsigned* cl = col + sc.off;
  const unsigned char* gd = ground + sc.off;
  int* pk = picks + (size_t)scan * kFeRows * 6 * kPickStride;
  const double kPi = 3.14159265358979323846;

  // ---- undistortPcl, pass 1 (ring 0's wave): where does halfPassed flip?  (SE:631-638: first-half adjustment) ----
  // The relative-time tag of a point is a function of the point, its index and the flip position; its consumers (the
  // picked points, the kept points of the less-flat cloud) evaluate it in fe_out_kernel where they write the point.
  // The flip is the FIRST index that passes pi: the organised cloud is ring-major and ring 0 sweeps the whole turn, so
  // the search stops after a few hundred points (it walks the whole cloud only when no point passes).
  int flip = n;
  if (ring == 0) {
    const double s_ori = (double)sc.start_ori;
    constexpr int kIn = 4;
    for (int i0 = 0; i0 < n && flip == n; i0 += kIn * 64) {
      float4 p[kIn];
#pragma unroll
      for (int u = 0; u < kIn; ++u) {
        const int i = i0 + u * 64 + lane;
        p[u] = pts[i < n ? i : n - 1];
      }
      int first = n;
#pragma unroll
      for (int u = kIn - 1; u >= 0; --u) {
        const int i = i0 + u * 64 + lane;
        double ori = (double)(-lins_atan2f(p[u].y, p[u].x));
        if (ori < s_ori - kPi / 2)
          ori += 2 * kPi;
        else if (ori > s_ori + kPi * 3 / 2)
          ori -= 2 * kPi;
        if (ori - s_ori > kPi && i < n) first = i;
      }
      for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o));
      flip = first;
    }
  }

  FE_MARK(0)
  // ---- the ring's window: its points [start - 4, end + 6) (cloud_info's ring indices leave 5 points out either side,
  // IP:395-410) and a halo of kHalo points, whose occlusion marks reach into the ring ----
  const int w0 = max(0, min(sc.start_ring[ring], n) - 4 - kHalo) & ~3;  // (a multiple of 4: a flag word of the window is a flag word of the cloud)
  const int w1 = max(w0, min(n, sc.end_ring[ring] + 6 + kHalo));
  FeRingInfo* const my = info + blockIdx.x;
  if (w1 - w0 > kRingWin) {  // never a VLP-16's ring (1800 cells): refused, not truncated
    if (lane == 0) my->m = -1, my->base = 0, my->nvox = 0, my->flip = flip, my->n_sharp = 0, my->n_less_sharp = 0, my->n_flat = 0, my->pad = 0;
    return;
  }
  const FeWin<unsigned> fw{L.a.flagw, w0 >> 2};                                              // fw[i >> 2]: the flag word of point i
  const FeWin<unsigned char> flb{reinterpret_cast<unsigned char*>(L.a.flagw), w0};         // flb[i]: its flag byte
  const FeWin<unsigned short> colb{L.a.col, w0};                                             // colb[i]: its column
  // markOccludedPoints (SE:680-713) over the window, as bit masks: a chunk of 64 consecutive points is a wave, a condition
  // a ballot.  Pass 1 reads a point's column and range and both neighbours' (global memory, coalesced, eight chunks in
  // flight) and keeps, per chunk, four 64-bit masks in the registers of the lane of the chunk's number (a window has at
  // most 29 chunks): A — the point hides its successor's surface: it and the five before it are marked (SE:693-697);
  // B — the successor hides it: the six behind it (SE:698-704); C — a beam nearly parallel to the surface: itself
  // (SE:707-711); G — ground.  Pass 2 smears A downwards and B upwards across the chunk borders (shifts of the chunk's
  // and its neighbour's masks: scalar instructions) and every lane stores its point's flag byte once — no atomics, no
  // zeroed array, no loop per mark.  Conditions of points outside the window do not exist (they belong to other rings'
  // waves; the halo covers every source that reaches this ring).
  {
    static_assert(kRingWin <= 64 * 64, "a chunk per lane");
    unsigned long long keep_a = 0ull, keep_b = 0ull, keep_c = 0ull, keep_g = 0ull;
    const int n_ch = (w1 - w0 + 63) >> 6;
    constexpr int kIn = 8;
    for (int c0 = 0; c0 < n_ch; c0 += kIn) {
      unsigned char g[kIn];
      unsigned c[kIn], c1[kIn];
      float rm[kIn], r0[kIn], rp[kIn];
#pragma unroll
      for (int u = 0; u < kIn; ++u) {
        const int i = w0 + (c0 + u) * 64 + lane, ic = i < w1 ? i : w1 - 1;  // (clamped reads; an empty window does not get here)
        const int im = ic > 0 ? ic - 1 : 0, ip = ic < n - 1 ? ic + 1 : n - 1;
        g[u] = gd[ic], c[u] = cl[ic], c1[u] = cl[ip];
        rm[u] = rg[im], r0[u] = rg[ic], rp[u] = rg[ip];
      }
#pragma unroll
      for (int u = 0; u < kIn; ++u) {
        const int i = w0 + (c0 + u) * 64 + lane;
        bool ca = false, cb = false, cc = false;
        if (i < w1) {
          colb[i] = (unsigned short)c[u];
          if (i >= 5 && i < n - 6) {
            const float d1 = r0[u], d2 = rp[u];
            int cd = (int)c1[u] - (int)c[u];
            cd = cd < 0 ? -cd : cd;
            if (cd < 10) {
              ca = d1 - d2 > 0.3;
              cb = !ca && d2 - d1 > 0.3;
            }
            const float f1 = fabsf(rm[u] - r0[u]), f2 = fabsf(rp[u] - r0[u]);
            cc = f1 > 0.02 * r0[u] && f2 > 0.02 * r0[u];
          }
        }
        const unsigned long long ma = __ballot(ca), mb = __ballot(cb), mc = __ballot(cc), mg = __ballot(i < w1 && g[u] != 0);
        if (lane == c0 + u) keep_a = ma, keep_b = mb, keep_c = mc, keep_g = mg;
      }
    }
    auto chunk_mask = [&](unsigned long long keep, int c) {  // chunk c's mask (0 outside the window), in every lane
      const int cc = c < 0 ? 0 : (c > 63 ? 63 : c);
      const unsigned lo = __builtin_amdgcn_readlane((unsigned)keep, cc), hi = __builtin_amdgcn_readlane((unsigned)(keep >> 32), cc);
      return (c < 0 || c >= n_ch) ? 0ull : ((unsigned long long)hi << 32) | lo;
    };
    for (int c = 0; c < n_ch; ++c) {
      const unsigned long long a0 = chunk_mask(keep_a, c), a1 = chunk_mask(keep_a, c + 1);
      const unsigned long long b0 = chunk_mask(keep_b, c), bp = chunk_mask(keep_b, c - 1);
      unsigned long long mk = chunk_mask(keep_c, c) | a0;
#pragma unroll
      for (int k = 1; k <= 5; ++k) mk |= (a0 >> k) | (a1 << (64 - k));  // a source marks the five points before it
#pragma unroll
      for (int k = 1; k <= 6; ++k) mk |= (b0 << k) | (bp >> (64 - k));  // ... the six behind it
      const unsigned long long gm = chunk_mask(keep_g, c);
      const int i = w0 + c * 64 + lane;
      if (i < w1) flb[i] = (unsigned char)(((mk >> lane) & 1ull) | (((gm >> lane) & 1ull) << 3));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // calculateSmoothness (SE:656-678) is evaluated where it is consumed — a sector's candidate keys below — from the range
  // array (f32, left to right, as written in SE:660-666); cloudCurvature / cloudSmoothness are never materialised.
  auto diff_at = [&](int i) {
    float d = 0.f;
    if (i >= 5 && i < n - 5)
      d = rg[i - 5] + rg[i - 4] + rg[i - 3] + rg[i - 2] + rg[i - 1] - rg[i] * 10 + rg[i + 1] + rg[i + 2] + rg[i + 3] +
          rg[i + 4] + rg[i + 5];
    return d;
  };
  // ---- extractFeatures: the ring's sectors in order (marks of one sector reach the next) ---
  FE_MARK(1)
  int r_sharp = 0, r_ls = 0, r_flat = 0;
  {
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
            const unsigned char f = flb[smooth_ind(sp + e)];
            dbits[u] = __float_as_uint(d);
            ecand |= (c > 0.5 && !(f & 9)) ? 1u << u : 0u;
            pcand |= (c < 0.5 && (f & 8)) ? 1u << u : 0u;  // (picked or not is looked up when the edges are done)
          }
        }
      }
      const int ind_ep = smooth_ind(ep);
      const float d_ep = fabsf(diff_at(ep));
      const double c_ep = (double)d_ep * (double)d_ep;
      {
        // Flags change by 32-bit LDS atomic ORs with no return value (`mark`, above) and are read as bytes: the LDS unit
        // takes a wave's instructions in order, so a flag read issued after an OR sees it — nothing to wait for; what
        // is needed is that the COMPILER keeps the order (it does for accesses that may alias; the barrier makes it plain).
        auto wave_sync = [&] {
          asm volatile("" ::: "memory");
          __builtin_amdgcn_wave_barrier();
        };
        auto set_bits = [&](int i, unsigned bits) {
          if (i >= w0 && i < w1) atomicOr(&fw[i >> 2], bits << ((i & 3) * 8));
        };
        auto col_gap = [&](int a, int b) {
          if (a < 0 || b < 0 || a >= n || b >= n) return 1000;
          const int g = (int)colb[a] - (int)colb[b];
          return g < 0 ? -g : g;
        };
        // cloudNeighborPicked of the +-5 neighbours up to the first column gap > 10 (SE:764-779): lanes 0-4
        // look forward, 5-9 backward, the break position comes from a ballot
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
        auto mark_nbrs = [&](int ind) {
          const bool fwd = lane < 5, bwd = lane >= 5 && lane < 10;
          const int l = fwd ? lane + 1 : -(lane - 5 + 1);
          const bool gap = (fwd || bwd) && col_gap(ind + l, ind + l + (fwd ? -1 : 1)) > 10;
          const unsigned long long gm = __ballot(gap);
          const int stop_f = __ffsll((long long)(gm & 0x1Full)), stop_b = __ffsll((long long)((gm >> 5) & 0x1Full));
          const int reach_f = stop_f ? stop_f - 1 : 5, reach_b = stop_b ? stop_b - 1 : 5;  // neighbours marked per side
          if ((fwd && lane < reach_f) || (bwd && lane - 5 < reach_b)) set_bits(ind + l, 1u);
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
          return wave_best64<kMax>(best);
        };
        int n_sharp = 0, n_ls = 0, n_flat = 0;
        // edges: largest curvature first, at most 2 sharp + 18 less sharp (SE:743-780)
        auto edge_pick = [&](int pind) {
          // cloudLabel 2 / 1, picked: ORed in — an eligible edge candidate (not ground, not picked) carries no label yet
          if (lane == 0) set_bits(pind, ((n_ls < 2 ? 2u : 1u) << 1) | 1u);
          if ((n_ls < 2 && lane == n_sharp) || lane == 2 + n_ls) pick_entry = pind;
          n_sharp += n_ls < 2 ? 1 : 0;
          ++n_ls;
          wave_sync();
          mark_nbrs(pind);
          wave_sync();
        };
        if (c_ep > 0.5 && !(flb[ind_ep] & 9)) {  // position ep comes first whatever its curvature
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
          unsigned long long* sk = L.work;
#pragma unroll
          for (int u = 0; u < kPmax; ++u)
            if (u * 64 < m && ((ecand >> u) & 1u)) sk[my_rank[u]] = ((unsigned long long)dbits[u] << 32) | (unsigned)smooth_ind(sp + u * 64 + lane);
          wave_sync();
          unsigned long long ck = lane < n_ec ? sk[lane] : 0ull;
          while (n_ls < 20) {  // the 21st eligible candidate would only end the loop (SE:757-759)
            const unsigned long long best = wave_best64<true>(ck);
            if (!best) break;  // (an edge candidate's key is > 0: its |diffRange| is)
            edge_pick((int)(unsigned)best);
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
        // planes: smallest curvature first, ground points only, at most 4; the 4th is not marked (SE:782-813)
        auto plane_pick = [&](int pind) {
          const bool last = n_flat + 1 >= 4;
          if (lane == 0) set_bits(pind, (3u << 1) | (last ? 0u : 1u));  // cloudLabel -1 (+ picked unless the 4th)
          if (lane == 22 + n_flat) pick_entry = pind;
          ++n_flat;
          wave_sync();
          if (!last) {
            mark_nbrs(pind);
            wave_sync();
          }
        };
        if (__any(pcand != 0)) {
          {  // the plane candidates the edge picks (and everything before this sector) have left unpicked
            unsigned char fl[kPmax];
#pragma unroll
            for (int u = 0; u < kPmax; ++u) fl[u] = u * 64 < m ? flb[smooth_ind(sp + u * 64 + lane)] : (unsigned char)1;
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
          const unsigned char f = flb[ind_ep];
          if ((f & 8) && !(f & 1)) plane_pick(ind_ep);
        }
        r_sharp += n_sharp, r_ls += n_ls, r_flat += n_flat;
        if (lane == 26) pick_entry = n_sharp;
        if (lane == 27) pick_entry = n_ls;
        if (lane == 28) pick_entry = n_flat;
        if (lane < 29) spk[lane] = pick_entry;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  FE_MARK(2)
  auto label_le0 = [&](int k) {  // cloudLabel <= 0: untouched (0) or flat (-1)
    const int b = (flb[k] >> 1) & 3;
    return b == 0 || b == 3;
  };
  // ---- less-flat cloud: per ring, every point of its sectors with label <= 0 (SE:815-820) ... -----
  // D0, the kept points' positions (relative to the ring's first sector) as a compact u16 list in
  // LDS — the sector sort buffers are idle.  Round 3: the points themselves are not touched here (rounds 1-2 copied
  // the kept points to a compact global list and read that list back twice).
  unsigned short* const kept = reinterpret_cast<unsigned short*>(L.work);  // [kRingCap]
  int m = 0, base = -1;
  {
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
    if (m > kRingCap || (m > 0 && sc.end_ring[ring] - base >= 2048)) m = -1;  // cannot happen for a 16 x 1800 sensor; refuse rather than truncate silently
    base = base < 0 ? 0 : base;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();  // (every read of the flags is done: the voxel orders below overlay them)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // D1, pcl::VoxelGrid 0.2 m with all-field averaging, output ordered by voxel index (SE:189, 822-825):
  // one wave per ring; 2048 keys (voxel index << 11 | order) sorted in registers, then only the order
  // and a run-start bit per sorted position go to LDS.  Sorted position e belongs to lane e / 32.
  // The kept points are read ONCE here: each lane packs the voxel coordinates of its points (11 + 11 + 10 bits around
  // zero: +-204 m / +-102 m) into one register per point, the wave folds the bounding box VoxelGrid needs
  // (getMinMax3D), and the keys are built from the packed coordinates.  A ring that does not pack (never a VLP-16's)
  // takes the same route with a second read of its points.
  // The centroids are fe_out_kernel's: the rings' output offsets are known once every ring has counted its voxels.
  FE_MARK(3)
  int nvox = 0;
  {
    {
      unsigned short* vs = L.vs;
      unsigned long long* const hbw = L.headbits;
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
          // Round 5: RUNS, not points, are sorted.  Along a ring consecutive kept points mostly fall into the same 0.2 m
          // voxel (beam spacing 3.5 cm at 10 m), and a run of them stays together under the (voxel index, order) sort, in
          // order.  So only the first point of every run — a "head" — goes through the bitonic network (64 * 2 .. 16 keys
          // instead of 64 * 8 .. 32; the network is the one VALU-saturated phase of the kernel), the sorted heads' run
          // lengths give every run its place by a prefix sum, and each head writes its run's points behind it.  Two runs
          // of one voxel (the beam left it and came back) end up adjacent, the earlier first: the same order.
          int heads = 0;
          if (narrow && packed) {
            unsigned* hk = reinterpret_cast<unsigned*>(vs);  // [1024] the heads' keys in order of appearance (vs is written after the sort)
            unsigned carry_vox = ~0u;                        // the voxel of element u * 64 - 1
#pragma unroll
            for (int u = 0; u < kP; ++u) {
              if (u * 64 < m) {  // (wave-uniform)
                const int e = u * 64 + lane;
                if (e < m) {
                  const int ix = (int)(pkd[u] & 2047u) - 1024, iy = (int)((pkd[u] >> 11) & 2047u) - 1024, iz = (int)(pkd[u] >> 22) - 512;
                  pkd[u] = ((unsigned)((ix - mnx) + (iy - mny) * (int)dx + (iz - mnz) * (int)(dx * dy)) << 11) | (unsigned)e;
                }
                const unsigned vox = pkd[u] >> 11;  // (no point: 0x1FFFFF, above every voxel index of a narrow box)
                unsigned pv = __shfl_up(vox, 1);
                if (lane == 0) pv = carry_vox;
                carry_vox = __shfl(vox, 63);
                const bool head = e < m && vox != pv;
                const unsigned long long bm = __ballot(head);
                if (head) {
                  const int r = heads + __popcll(bm & ((1ull << lane) - 1ull));
                  if (r < 1024) hk[r] = pkd[u];
                }
                if (lane == 0) hbw[u] = bm;
                heads += __popcll(bm);
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
          auto sort_heads = [&](auto qtag) {
            constexpr int kQ = decltype(qtag)::value;
            const unsigned* hk = reinterpret_cast<const unsigned*>(vs);
            const unsigned long long* hb = hbw;
            unsigned kv[kQ];
#pragma unroll
            for (int u = 0; u < kQ; ++u) kv[u] = u * 64 + lane < heads ? hk[u * 64 + lane] : ~0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (every lane has its keys: vs, the same bytes, is written below)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            FE_MARK(4)
            wave_bitonic_sort<kQ, unsigned>(kv, lane);
            FE_MARK(5)
            // sorted head lane * kQ + u: its run's length (to the next head in order of appearance, or the ring's end),
            // whether it opens a voxel (the sorted predecessor is of another one)
            unsigned prev = __shfl_up(kv[kQ - 1], 1);
            int len[kQ], my_len = 0;
#pragma unroll
            for (int u = 0; u < kQ; ++u) {
              const int sidx = lane * kQ + u;
              len[u] = 0;
              if (sidx < heads) {
                const int e = (int)(kv[u] & 2047u), q = e + 1;
                int nx = m;
                const unsigned long long x = q < m ? hb[q >> 6] >> (q & 63) : 0ull;  // (words from m / 64 rounded up on are not written)
                if (x) {
                  nx = q + __ffsll((long long)x) - 1;
                } else {
                  for (int w = (q >> 6) + 1; w < (m + 63) >> 6; ++w)
                    if (hb[w]) {
                      nx = w * 64 + __ffsll((long long)hb[w]) - 1;
                      break;
                    }
                }
                len[u] = nx - e;
                const bool start = sidx == 0 || (prev >> 11) != (kv[u] >> 11);
                mine += start ? 1 : 0;
                startmask |= start ? (1u << u) : 0u;
                prev = kv[u];
              }
              my_len += len[u];
            }
            int incl_len = my_len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
              const int nb = __shfl_up(incl_len, o, 64);
              if (lane >= o) incl_len += nb;
            }
            // Every run's points behind its head, without a loop per run: a head leaves a marker at its position — the
            // distance from a sorted position of its run to the point's place in the kept list, e - pos, is the same for
            // the whole run — then every lane walks its own block of consecutive positions, carrying the last marker
            // along (the block's first positions take the last marker of the lanes before: one shuffle), and replaces
            // each position by (voxel start, kept[position + distance]).  Marker: bit 15 set, bit 14 voxel start, bits
            // 0-11 distance + 2048.
            const int blk = (((m + 63) >> 6) + 3) & ~3;  // positions per lane (a multiple of 4: 64-bit LDS words), <= 32
            unsigned long long* const vw = reinterpret_cast<unsigned long long*>(vs);
            for (int g = 0; g < blk; g += 4) vw[(lane * blk + g) >> 2] = 0ull;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int pos = incl_len - my_len;
#pragma unroll
            for (int u = 0; u < kQ; ++u) {
              const int e = (int)(kv[u] & 2047u);
              if (len[u] > 0) vs[pos] = (unsigned short)(0x8000u | (((startmask >> u) & 1u) ? 0x4000u : 0u) | (unsigned)(e - pos + 2048));
              pos += len[u];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
              const int p0 = lane * blk;
              unsigned last = 0u;  // this block's last marker
              for (int g = 0; g < blk; g += 4) {
                const unsigned long long w = vw[(p0 + g) >> 2];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const unsigned x = (unsigned)(w >> (16 * k)) & 0xFFFFu;
                  last = (x & 0x8000u) ? x : last;
                }
              }
              const unsigned long long below = __ballot(last != 0u) & ((1ull << lane) - 1ull);
              const unsigned from_left = __shfl(last, below ? 63 - __builtin_clzll(below) : 0);
              unsigned cur = below ? from_left : 0u;
              for (int g = 0; g < blk; g += 4) {
                const unsigned long long w = vw[(p0 + g) >> 2];
                unsigned long long o = 0ull;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                  const unsigned x = (unsigned)(w >> (16 * k)) & 0xFFFFu;
                  const bool mk = (x & 0x8000u) != 0u;
                  cur = mk ? x : cur;
                  const int q = p0 + g + k;
                  unsigned val = 0u;
                  if (q < m) val = (unsigned)kept[q + (int)(cur & 0xFFFu) - 2048] | ((mk && (x & 0x4000u)) ? 0x8000u : 0u);
                  o |= (unsigned long long)val << (16 * k);
                }
                vw[(p0 + g) >> 2] = o;
              }
            }
            int incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
              const int nb = __shfl_up(incl, o, 64);
              if (lane >= o) incl += nb;
            }
            nvox = __shfl(incl, 63);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // (the list of kept points has been folded into vs: see the other route below)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            int run = incl - mine;
            pos = incl_len - my_len;
#pragma unroll
            for (int u = 0; u < kQ; ++u) {
              if ((startmask >> u) & 1u) kept[pos] = (unsigned short)run++;
              pos += len[u];
            }
          };
          if (narrow && packed && heads <= 1024 && heads * 2 <= kP * 64) {  // (fewer keys per lane than the points' network)
            if (heads <= 128)
              sort_heads(FeInt<2>{});
            else if (heads <= 256)
              sort_heads(FeInt<4>{});
            else if (heads <= 512)
              sort_heads(FeInt<8>{});
            else
              sort_heads(FeInt<16>{});
            return;
          }
          auto sort_and_mark = [&](auto key_zero, auto from_pack) {
            using K = decltype(key_zero);  // 32-bit keys when (voxel index << 11 | order) fits: half the shuffles and compares
            K kv[kP];
#pragma unroll
            for (int u = 0; u < kP; ++u) {
              const int e = u * 64 + lane;
              K k = (K)~key_zero;
              if (e < m) {
                if (decltype(from_pack)::value) {
                  k = (K)pkd[u];  // (the key was formed in place above)
                } else {
                  const float4 p = pts[base + (int)kept[e]];
                  const long long ix = (long long)floorf(p.x * inv), iy = (long long)floorf(p.y * inv), iz = (long long)floorf(p.z * inv);
                  k = (K)(((unsigned long long)((ix - mnx) + (iy - mny) * dx + (iz - mnz) * dx * dy) << 11) | (unsigned)e);
                }
              }
              kv[u] = k;
            }
            FE_MARK(4)
            wave_bitonic_sort<kP, K>(kv, lane);
            FE_MARK(5)
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
    }
  }
  FE_MARK(6)
  // ---- what fe_out_kernel needs ----
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (m > 0) {
    unsigned* const ord = order + sc.off + base;
    for (int e = lane; e < m; e += 64) {
      const unsigned v = L.vs[e];
      ord[e] = v | ((v & 0x8000u) ? (unsigned)kept[e] << 16 : 0u);
    }
  }
  if (lane == 0) my->m = m, my->base = base, my->nvox = nvox, my->flip = flip, my->n_sharp = r_sharp, my->n_less_sharp = r_ls, my->n_flat = r_flat, my->pad = 0;
#ifdef LINS_FE_PROF
  FE_MARK(7)
  if (lane == 0 && scan == 0)
    printf("FE ring %2d: flip %6lld masks %6lld picks %7lld compact %6lld | voxel: read+heads %6lld sort %6lld fill %6lld | out %6lld | m %d picks %d\n", ring, fe_t[0], fe_t[1], fe_t[2],
           fe_t[3], fe_t[4], fe_t[5], fe_t[6], fe_t[7], m, r_ls + r_flat);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// fe_out_kernel: 256 threads per (scan, ring).  The ring's place in the four feature clouds follows from the counts of the
// rings before it (sixteen FeRingInfo records); then
//  * the picked clouds in the reference's order — rings, sectors, pick order (SE:743-813) — one thread per pick slot;
//  * D2, the less-flat cloud: one centroid per voxel — f32 sums of all four fields in stable (original) order (PCL's
//    VoxelGrid with all-field averaging, SE:189, 822-825) — at the ring's offset + the voxel's slot.
__global__ __launch_bounds__(kOutBlock) void fe_out_kernel(
    const FeScan* __restrict__ scans, const float4* __restrict__ cloud, double scan_period, const int* __restrict__ picks,
    const unsigned* __restrict__ order, const FeRingInfo* __restrict__ info, float4* __restrict__ out, int* __restrict__ out_counts) {
  __shared__ int s_pick[6 * kPickStride];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int scan = blockIdx.x / kFeRows, ring = blockIdx.x % kFeRows;
  const FeScan& sc = scans[scan];
  const float4* pts = cloud + sc.off;
  const FeRingInfo* inf = info + (size_t)scan * kFeRows;
  const double kPi = 3.14159265358979323846;
  // the rings before this one (every wave for itself: sixteen records, a wave scan — no barrier)
  int off_sharp, off_ls, off_flat, off_lf, m, base;
  bool bad;
  {
    const bool in = lane < kFeRows;
    const int rm = in ? inf[lane].m : 0, rv = in ? inf[lane].nvox : 0;
    const int a = in ? inf[lane].n_sharp : 0, b = in ? inf[lane].n_less_sharp : 0, c = in ? inf[lane].n_flat : 0;
    bad = __any(rm < 0);
    int ia = a, ib = b, ic = c, iv = rv;
#pragma unroll
    for (int o = 1; o < kFeRows; o <<= 1) {
      const int na = __shfl_up(ia, o, 64), nb = __shfl_up(ib, o, 64), nc = __shfl_up(ic, o, 64), nv = __shfl_up(iv, o, 64);
      if (lane >= o) ia += na, ib += nb, ic += nc, iv += nv;
    }
    off_sharp = __shfl(ia - a, ring), off_ls = __shfl(ib - b, ring), off_flat = __shfl(ic - c, ring), off_lf = __shfl(iv - rv, ring);
    m = __shfl(rm, ring), base = in ? inf[ring].base : 0;
    base = __shfl(base, 0);
    if (ring == 0 && wave == 0 && lane == kFeRows - 1) {
      out_counts[scan * 4 + 0] = ia, out_counts[scan * 4 + 1] = ib, out_counts[scan * 4 + 2] = ic;
      out_counts[scan * 4 + 3] = bad ? -1 : iv;  // (-1: a ring beyond the kernels' limits, LINS_E_UNSUPPORTED)
    }
  }
  if (bad) return;
  const int flip = inf[0].flip;
  const double s_ori = (double)sc.start_ori, e_ori = (double)sc.end_ori, ori_diff = (double)sc.ori_diff;
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
  // ---- the picked clouds: the ring's six sectors' lists (29 ints each) through LDS, a thread per slot ----
  static_assert(kOutBlock >= 6 * kPickStride && kOutBlock >= 6 * 26, "a thread per pick slot");
  if (tid < 6 * kPickStride) s_pick[tid] = picks[((size_t)scan * kFeRows + ring) * 6 * kPickStride + tid];
  __syncthreads();
  if (tid < 6 * 26) {
    const int s2 = tid / 26, k = tid - s2 * 26;
    const int* spk = s_pick + s2 * kPickStride;
    int before_sharp = 0, before_ls = 0, before_flat = 0;  // the ring's sectors before this one
    for (int j = 0; j < s2; ++j) before_sharp += s_pick[j * kPickStride + 26], before_ls += s_pick[j * kPickStride + 27], before_flat += s_pick[j * kPickStride + 28];
    long long dst = -1;
    if (k < 2) {
      if (k < spk[26]) dst = sc.o_sharp + off_sharp + before_sharp + k;
    } else if (k < 22) {
      if (k - 2 < spk[27]) dst = sc.o_less_sharp + off_ls + before_ls + (k - 2);
    } else {
      if (k - 22 < spk[28]) dst = sc.o_flat + off_flat + before_flat + (k - 22);
    }
    if (dst >= 0) {  // the de-skewed point: coordinates as they came, the relative-time tag as intensity (SE:649-650)
      const int i = spk[k];
      float4 q = pts[i];
      q.w = tag_of(i, q);
      out[dst] = q;
    }
  }
  // ---- D2: the ring's centroids ----
  // A wave takes 64 consecutive sorted positions per step: every lane reads ITS point and forms its tag (one gather and
  // one arctangent per lane, no divergence); a voxel's lanes are neighbours, so its first lane collects the others' values
  // left to right with shuffles — the sums in the order VoxelGrid adds them (ascending original index).  A wave takes
  // CONSECUTIVE chunks and carries the partial sums of a voxel that goes on beyond a chunk's last lane into the next
  // chunk — lane 0 continues it; only a group's last chunk finishes its last run alone, one dependent gather and one
  // arctangent per point (with ~5 points a voxel nearly every chunk ends in such a run).
  if (m <= 0) return;
  const unsigned* ord = order + sc.off + base;
  float4* dst = out + sc.o_less_flat + off_lf;
  const int n_chunks = (m + 63) >> 6;
  const int group = max(1, (n_chunks + (kOutBlock / 64) * LINS_FE_D2_GROUP - 1) / ((kOutBlock / 64) * LINS_FE_D2_GROUP));
  for (int g0 = wave * group; g0 < n_chunks; g0 += (kOutBlock / 64) * group) {
    float cx = 0.f, cy = 0.f, cz = 0.f, ci = 0.f;  // the carried run (wave-uniform): sums so far, points so far, output slot
    int c_cnt = 0, c_slot = -1;
    const int g1 = min(g0 + group, n_chunks);
    // two deep: while chunk c is worked on, chunk c + 1's points and chunk c + 2's entries are in flight (a chunk is a
    // chain of two dependent reads — the entry, then the point it names — and a wave has only a handful of chunks)
    unsigned v_cur = g0 * 64 + lane < m ? ord[g0 * 64 + lane] : 0x8000u;
    unsigned v_next = g0 * 64 + 64 + lane < m ? ord[g0 * 64 + 64 + lane] : 0x8000u;
    float4 p_cur = make_float4(0.f, 0.f, 0.f, 0.f);
    if (g0 * 64 + lane < m) p_cur = pts[base + (int)(v_cur & 2047u)];
    for (int ch = g0; ch < g1; ++ch) {
      const int e0 = ch * 64, e = e0 + lane;
      const bool valid = e < m;
      const unsigned v = v_cur;
      float4 p = p_cur;
      const unsigned v_after = e + 128 < m ? ord[e + 128] : 0x8000u;
      p_cur = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e + 64 < m && ch + 1 < g1) p_cur = pts[base + (int)(v_next & 2047u)];
      const unsigned v_first = __shfl(v_next, 0);  // (the next chunk's first entry tells whether this chunk's last run goes on)
      v_cur = v_next, v_next = v_after;
      const bool start = valid && (v & 0x8000u);
      const bool head = lane == 0 && c_slot >= 0 && !start;
      const bool st = start || head;
      const int i = base + (int)(v & 2047u);
      float tg = 0.f;
      if (valid) tg = tag_of(i, p);
      const unsigned long long bounds = __ballot(start || !valid);  // where a run ends: the next start, or the end of the ring
      const unsigned long long above = lane < 63 ? bounds & ~((2ull << lane) - 1ull) : 0ull;
      const int nxt = above ? __ffsll((long long)above) - 1 : 64;
      const int len_in = nxt - lane;  // this run's points inside the step (meaningful for start lanes)
      float sx = (head ? cx : 0.f) + p.x, sy = (head ? cy : 0.f) + p.y, sz = (head ? cz : 0.f) + p.z, si = (head ? ci : 0.f) + tg;
      for (int d = 1; __any(st && d < len_in); ++d) {
        const float ax = __shfl_down(p.x, d), ay = __shfl_down(p.y, d), az = __shfl_down(p.z, d), at = __shfl_down(tg, d);
        if (st && d < len_in) sx += ax, sy += ay, sz += az, si += at;
      }
      int total = len_in + (head ? c_cnt : 0);
      const int my_slot = head ? c_slot : (int)(v >> 16);
      // does the run that reaches the last lane go on in the next chunk?  (wave-uniform)
      const unsigned long long stm = __ballot(st);
      const bool goes_on = stm != 0ull && e0 + 64 < m && !(v_first & 0x8000u);
      const int last_st = stm ? 63 - __builtin_clzll(stm) : 0;  // its first lane (the run above it reaches lane 63: valid lanes throughout)
      const bool carry = goes_on && ch + 1 < g1;
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
          while (j < m) {
            const unsigned v2 = ord[j];
            if (v2 & 0x8000u) break;
            const int i2 = base + (int)(v2 & 2047u);
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
}

void launch_frontend(hipStream_t stream, int n_scans, const void* scans, const float4* cloud, const float* range,
                     const unsigned* col, const unsigned char* ground, double scan_period, int* picks, unsigned* order,
                     void* ring_info, float4* out, int* out_counts) {
  if (n_scans <= 0) return;
  static const int stages = [] {  // (debug knob: time one kernel without the other — tools/frontend_rate.py)
    const char* g = std::getenv("LINS_ENABLE_DEBUG_KNOBS");
    const char* e = std::getenv("LINS_FE_STAGES");
    return g && g[0] == '1' && e ? std::atoi(e) : 3;
  }();
  if (stages & 1)
  hipLaunchKernelGGL(fe_ring_kernel, dim3(n_scans * kFeRows), dim3(64), 0, stream, (const FeScan*)scans, cloud, range, col, ground,
                     picks, order, (FeRingInfo*)ring_info);
  if (stages & 2)
  hipLaunchKernelGGL(fe_out_kernel, dim3(n_scans * kFeRows), dim3(kOutBlock), 0, stream, (const FeScan*)scans, cloud, scan_period,
                     picks, order, (const FeRingInfo*)ring_info, out, out_counts);
}
size_t fe_scan_size() { return sizeof(FeScan); }
int fe_pick_stride() { return kFeRows * 6 * kPickStride; }
size_t fe_ring_info_size() { return sizeof(FeRingInfo) * kFeRows; }

}  // namespace lins
