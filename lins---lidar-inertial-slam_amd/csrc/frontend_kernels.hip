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
// loops (latency bound, one lane per ring), the per-sector sort is a 512-key bitonic network per
// wave in LDS, the per-ring VoxelGrid a 2048-key bitonic network over the whole workgroup.
//
// Sort keys carry the point index as tie-break ((|diffRange| bits, index): the reference's
// std::sort leaves the order of equal curvatures unspecified; the host restatement uses the same
// total order), so the picks are reproducible and identical on both sides.

#include <hip/hip_runtime.h>

#include "../../include/lins_host.h"
#include "lins_math.h"

namespace lins {

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
};

struct FeLds {
  unsigned char flags[kFeMaxN + 16];   // bit 0 picked (cloudNeighborPicked); bits 1-2 cloudLabel: 0 = 0, 1 = 1 (less
                                       // sharp), 2 = 2 (sharp), 3 = -1 (flat); bit 3 ground
  unsigned short col[kFeMaxN + 16];
  union {
    unsigned long long skey[kFeRows][kSectorCap];  // per wave: (|diffRange| bits << 32) | index
    struct {
      unsigned long long vkey[kRingCap];  // (voxel index << 11) | order
      float4 vpt[kRingCap];
    } vox;
  };
  int first_half_end;   // first point with ori - startOri > pi (halfPassed flips after it)
  int bmin[3], bmax[3];  // ordered-int min / max of the ring's less-flat points
  int scan_tmp[20];
  int ring_m;           // less-flat points of the current ring
  int out_base;         // less-flat points written so far
};
static_assert(sizeof(FeLds) <= 160 * 1024, "LDS budget");

__shared__ FeLds g_fe;

__device__ __forceinline__ int fe_ordered_int(float f) {
  int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float fe_ordered_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

// exclusive prefix sum of one value per thread over the block; total in *total (LDS)
__device__ __forceinline__ int fe_block_scan(int v, int tid, int* tmp) {
  const int lane = tid & 63, wave = tid >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int nb = __shfl_up(incl, o, 64);
    if (lane >= o) incl += nb;
  }
  if (lane == 63) tmp[wave] = incl;
  __syncthreads();
  int off = 0, tot = 0;
  for (int w = 0; w < kFeBlock / 64; ++w) {
    if (w < wave) off += tmp[w];
    tot += tmp[w];
  }
  __syncthreads();
  tmp[18] = tot;  // (every thread writes the same value)
  return off + incl - v;
}


__global__ __launch_bounds__(kFeBlock) void frontend_kernel(
    const FeScan* __restrict__ scans, const float4* __restrict__ cloud, const float* __restrict__ range,
    const unsigned* __restrict__ col, const unsigned char* __restrict__ ground, double scan_period,
    float4* __restrict__ und, float* __restrict__ diff, int* __restrict__ picks, float4* __restrict__ out_sharp,
    float4* __restrict__ out_less_sharp, float4* __restrict__ out_flat, float4* __restrict__ out_less_flat,
    int* __restrict__ out_counts) {
  FeLds& L = g_fe;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int scan = blockIdx.x;
  const FeScan sc = scans[scan];
  const int n = sc.n;
  const float4* pts = cloud + sc.off;
  const float* rg = range + sc.off;
  const unsigned* cl = col + sc.off;
  const unsigned char* gd = ground + sc.off;
  float4* un = und + sc.off;
  float* df = diff + sc.off;
  int* pk = picks + (size_t)scan * kFeRows * 6 * kPickStride;
  const double kPi = 3.14159265358979323846;

  if (tid == 0) L.first_half_end = n, L.out_base = 0;
  for (int i = tid; i < n + 16; i += kFeBlock) {
    L.flags[i] = i < n && gd[i] ? 8 : 0;
    L.col[i] = i < n ? (unsigned short)cl[i] : 0;
  }
  __syncthreads();

  // ---- undistortPcl, pass 1: where does halfPassed flip?  (SE:631-638: first-half adjustment) ----
  const double s_ori = (double)sc.start_ori, e_ori = (double)sc.end_ori;
  for (int i = tid; i < n; i += kFeBlock) {
    const float4 p = pts[i];
    double ori = (double)(-atan2f(p.y, p.x));
    if (ori < s_ori - kPi / 2)
      ori += 2 * kPi;
    else if (ori > s_ori + kPi * 3 / 2)
      ori -= 2 * kPi;
    if (ori - s_ori > kPi) atomicMin(&L.first_half_end, i);
  }
  __syncthreads();
  const int flip = L.first_half_end;
  // ---- pass 2: relative time tag; smoothness stencil; masks -------------------------------------
  for (int i = tid; i < n; i += kFeBlock) {
    float4 p = pts[i];
    double ori = (double)(-atan2f(p.y, p.x));
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
    const double rel = (ori - s_ori) / (double)sc.ori_diff;
    p.w = (float)((double)(int)p.w + scan_period * rel);
    un[i] = p;
    float d = 0.f;
    if (i >= 5 && i < n - 5)  // calculateSmoothness (f32, left to right, as written in SE:660-666)
      d = rg[i - 5] + rg[i - 4] + rg[i - 3] + rg[i - 2] + rg[i - 1] - rg[i] * 10 + rg[i + 1] + rg[i + 2] + rg[i + 3] +
          rg[i + 4] + rg[i + 5];
    df[i] = d;
  }
  unsigned* fw = reinterpret_cast<unsigned*>(L.flags);  // (marks are idempotent bit sets: 32-bit LDS atomics)
  auto mark = [&](int i) { atomicOr(&fw[i >> 2], 1u << ((i & 3) * 8)); };
  for (int i = tid; i < n; i += kFeBlock) {
    if (i >= 5 && i < n - 6) {  // markOccludedPoints (SE:680-713)
      const float d1 = rg[i], d2 = rg[i + 1];
      int cd = (int)(cl[i + 1] - cl[i]);
      cd = cd < 0 ? -cd : cd;
      if (cd < 10) {
        if (d1 - d2 > 0.3) {
          for (int k = 0; k <= 5; ++k) mark(i - k);
        } else if (d2 - d1 > 0.3) {
          for (int k = 1; k <= 6; ++k) mark(i + k);
        }
      }
      const float f1 = fabsf(rg[i - 1] - rg[i]), f2 = fabsf(rg[i + 1] - rg[i]);
      if (f1 > 0.02 * rg[i] && f2 > 0.02 * rg[i]) mark(i);
    }
  }
  __syncthreads();

  // ---- extractFeatures: one wave per ring, sectors in order (marks of one sector reach the next) ---
  {
    const int ring = wave;
    unsigned long long* key = L.skey[ring];
    for (int j = 0; j < 6; ++j) {
      const int sp = (sc.start_ring[ring] * (6 - j) + sc.end_ring[ring] * j) / 6;
      const int ep = (sc.start_ring[ring] * (5 - j) + sc.end_ring[ring] * (j + 1)) / 6 - 1;
      int* spk = pk + (ring * 6 + j) * kPickStride;
      bool skip = sp >= ep || sp < 0 || ep >= n || ep - sp > kSectorCap - 1;
      if (skip) {
        if (lane < 3) spk[26 + lane] = 0;
        continue;
      }
      const int m = ep - sp;  // the sort covers [sp, ep) — ep itself keeps its place (SE:739-740)
      // cloudSmoothness[i].ind is i only where the stencil ran, [5, n - 5); elsewhere the value-initialised 0
      auto smooth_ind = [&](int i) { return (i >= 5 && i < n - 5) ? i : 0; };
      for (int e = lane; e < kSectorCap; e += 64)
        key[e] = e < m ? ((unsigned long long)__float_as_uint(fabsf(df[sp + e])) << 32) | (unsigned)smooth_ind(sp + e) : ~0ull;
      // bitonic sort, ascending (wave-local: LDS accesses of one wave are ordered by the fences)
      for (int k2 = 2; k2 <= kSectorCap; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          for (int e = lane; e < kSectorCap; e += 64) {
            const int partner = e ^ j2;
            if (partner > e) {
              const unsigned long long a = key[e], b = key[partner];
              const bool up = (e & k2) == 0;
              if ((a > b) == up) key[e] = b, key[partner] = a;
            }
          }
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (lane == 0) {
        auto sorted_ind = [&](int k) { return k == ep ? smooth_ind(ep) : (int)(unsigned)key[k - sp]; };
        auto col_gap = [&](int a, int b) {
          if (a < 0 || b < 0 || a >= n || b >= n) return 1000;
          const int g = (int)L.col[a] - (int)L.col[b];
          return g < 0 ? -g : g;
        };
        auto mark_nbrs = [&](int ind) {
          for (int l = 1; l <= 5; ++l) {
            if (col_gap(ind + l, ind + l - 1) > 10) break;
            L.flags[ind + l] |= 1;
          }
          for (int l = -1; l >= -5; --l) {
            if (col_gap(ind + l, ind + l + 1) > 10) break;
            L.flags[ind + l] |= 1;
          }
        };
        auto curv_of = [&](int ind) {
          const double d = (double)df[ind];
          return d * d;
        };
        int n_sharp = 0, n_ls = 0, n_flat = 0, largest = 0;
        for (int k = ep; k >= sp; --k) {  // edges: largest curvature first (SE:743-779)
          const int ind = sorted_ind(k);
          const unsigned char f = L.flags[ind];
          if (!(f & 1) && curv_of(ind) > 0.5 && !(f & 8)) {
            ++largest;
            if (largest <= 2) {
              L.flags[ind] = (unsigned char)((f & ~6) | (2 << 1));  // cloudLabel 2
              spk[n_sharp++] = ind;
              spk[2 + n_ls++] = ind;
            } else if (largest <= 20) {
              L.flags[ind] = (unsigned char)((f & ~6) | (1 << 1));  // cloudLabel 1
              spk[2 + n_ls++] = ind;
            } else {
              break;
            }
            L.flags[ind] |= 1;
            mark_nbrs(ind);
          }
        }
        int smallest = 0;
        for (int k = sp; k <= ep; ++k) {  // planes: smallest curvature first, ground only (SE:782-813)
          const int ind = sorted_ind(k);
          const unsigned char f = L.flags[ind];
          if (!(f & 1) && curv_of(ind) < 0.5 && (f & 8)) {
            L.flags[ind] = (unsigned char)(f | (3 << 1));  // cloudLabel -1
            spk[22 + n_flat++] = ind;
            if (++smallest >= 4) break;
            L.flags[ind] |= 1;
            mark_nbrs(ind);
          }
        }
        spk[26] = n_sharp, spk[27] = n_ls, spk[28] = n_flat;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  __threadfence_block();
  __syncthreads();

  auto label_le0 = [&](int k) {  // cloudLabel <= 0: untouched (0) or flat (-1)
    const int b = (L.flags[k] >> 1) & 3;
    return b == 0 || b == 3;
  };

  // ---- feature clouds in the reference's order: rings, sectors, pick order -------------------------
  {
    int* cnt = reinterpret_cast<int*>(L.vox.vkey);  // [3][96] counts, then [3][96] exclusive offsets (region idle here)
    constexpr int kSec = kFeRows * 6;
    if (tid < kSec) {
      const int* spk = pk + tid * kPickStride;
      cnt[tid] = spk[26], cnt[kSec + tid] = spk[27], cnt[2 * kSec + tid] = spk[28];
    }
    __syncthreads();
    if (tid < 3) {
      int run = 0;
      for (int s2 = 0; s2 < kSec; ++s2) {
        cnt[(3 + tid) * kSec + s2] = run;
        run += cnt[tid * kSec + s2];
      }
      out_counts[scan * 4 + tid] = run;
    }
    __syncthreads();
    for (int t = tid; t < kSec * 26; t += kFeBlock) {
      const int s2 = t / 26, k = t - s2 * 26;
      const int* spk = pk + s2 * kPickStride;
      if (k < 2) {
        if (k < spk[26]) out_sharp[(size_t)scan * 192 + cnt[3 * kSec + s2] + k] = un[spk[k]];
      } else if (k < 22) {
        if (k - 2 < spk[27]) out_less_sharp[(size_t)scan * 1920 + cnt[4 * kSec + s2] + (k - 2)] = un[spk[k]];
      } else {
        if (k - 22 < spk[28]) out_flat[(size_t)scan * 384 + cnt[5 * kSec + s2] + (k - 22)] = un[spk[k]];
      }
    }
    __syncthreads();
  }

  // ---- less-flat cloud: per ring, every point of its sectors with label <= 0, VoxelGrid 0.2 m ------
  float4* olf = out_less_flat + (size_t)scan * kFeMaxN;
  for (int ring = 0; ring < kFeRows; ++ring) {
    // the ring's sector spans, in order, concatenated: positions [lo_j, hi_j]
    int my_k[2] = {-1, -1};
    {
      // enumerate candidate indices of this ring: the union of its valid sectors is contiguous per sector;
      // thread t takes candidates t and t + 1024 of the concatenation
      int base = 0;
      for (int j = 0; j < 6; ++j) {
        const int sp = (sc.start_ring[ring] * (6 - j) + sc.end_ring[ring] * j) / 6;
        const int ep = (sc.start_ring[ring] * (5 - j) + sc.end_ring[ring] * (j + 1)) / 6 - 1;
        if (sp >= ep || sp < 0 || ep >= n || ep - sp > kSectorCap - 1) continue;
        const int len = ep - sp + 1;
        for (int u = 0; u < 2; ++u) {
          const int c = tid + u * kFeBlock - base;
          if (c >= 0 && c < len) my_k[u] = sp + c;
        }
        base += len;
      }
    }
    bool keep[2];
    for (int u = 0; u < 2; ++u) {
      keep[u] = my_k[u] >= 0 && label_le0(my_k[u]);

    }
    // order inside the ring = candidate order: thread t's first candidate precedes every candidate of
    // higher threads, its second (t + 1024) follows all first candidates
    const int first_cnt = keep[0] ? 1 : 0, second_cnt = keep[1] ? 1 : 0;
    const int pos0 = fe_block_scan(first_cnt, tid, L.scan_tmp);
    const int tot0 = L.scan_tmp[18];
    __syncthreads();
    const int pos1 = fe_block_scan(second_cnt, tid, L.scan_tmp);
    const int m = tot0 + L.scan_tmp[18];
    __syncthreads();
    if (tid < 3) L.bmin[tid] = 0x7FFFFFFF, L.bmax[tid] = (int)0x80000000;
    __syncthreads();
    if (m > kRingCap) {  // cannot happen for a 16 x 1800 sensor; refuse rather than truncate silently
      if (tid == 0) out_counts[scan * 4 + 3] = -1;
      return;
    }
    for (int u = 0; u < 2; ++u)
      if (keep[u]) {
        const float4 p = un[my_k[u]];
        const int pos = u == 0 ? pos0 : tot0 + pos1;
        L.vox.vpt[pos] = p;
        atomicMin(&L.bmin[0], fe_ordered_int(p.x)), atomicMax(&L.bmax[0], fe_ordered_int(p.x));
        atomicMin(&L.bmin[1], fe_ordered_int(p.y)), atomicMax(&L.bmax[1], fe_ordered_int(p.y));
        atomicMin(&L.bmin[2], fe_ordered_int(p.z)), atomicMax(&L.bmax[2], fe_ordered_int(p.z));
      }
    __syncthreads();
    if (m > 0) {
      const float inv = 1.0f / 0.2f;
      int minb[3], maxb[3];
      for (int a = 0; a < 3; ++a) {
        minb[a] = (int)floorf(fe_ordered_float(L.bmin[a]) * inv);
        maxb[a] = (int)floorf(fe_ordered_float(L.bmax[a]) * inv);
      }
      const long long dx = maxb[0] - minb[0] + 1, dy = maxb[1] - minb[1] + 1;
      for (int e = tid; e < kRingCap; e += kFeBlock) {
        unsigned long long k = ~0ull;
        if (e < m) {
          const float4 p = L.vox.vpt[e];
          const long long ix = (long long)floorf(p.x * inv) - minb[0];
          const long long iy = (long long)floorf(p.y * inv) - minb[1];
          const long long iz = (long long)floorf(p.z * inv) - minb[2];
          k = ((unsigned long long)(ix + iy * dx + iz * dx * dy) << 11) | (unsigned)e;
        }
        L.vox.vkey[e] = k;
      }
      // bitonic sort of 2048 keys over the block (stable: the order is part of the key)
      for (int k2 = 2; k2 <= kRingCap; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
          __syncthreads();
          for (int e = tid; e < kRingCap; e += kFeBlock) {
            const int partner = e ^ j2;
            if (partner > e) {
              const unsigned long long a = L.vox.vkey[e], b = L.vox.vkey[partner];
              const bool up = (e & k2) == 0;
              if ((a > b) == up) L.vox.vkey[e] = b, L.vox.vkey[partner] = a;
            }
          }
        }
      __syncthreads();
      // one thread per voxel run: centroid of all four fields, f32 sums in stable order
      int starts = 0;
      float4 cen[2];
      bool is_start[2] = {false, false};
      for (int u = 0; u < 2; ++u) {
        const int e = tid + u * kFeBlock;
        if (e < m) {
          const unsigned long long ke = L.vox.vkey[e] >> 11;
          if (e == 0 || (L.vox.vkey[e - 1] >> 11) != ke) {
            float sx = 0, sy = 0, sz = 0, si = 0;
            int j = e;
            while (j < m && (L.vox.vkey[j] >> 11) == ke) {
              const float4 p = L.vox.vpt[(int)(L.vox.vkey[j] & 2047u)];
              sx += p.x, sy += p.y, sz += p.z, si += p.w;
              ++j;
            }
            const float cnt = (float)(j - e);
            cen[u] = make_float4(sx / cnt, sy / cnt, sz / cnt, si / cnt);
            is_start[u] = true;
            ++starts;
          }
        }
      }
      // output order = voxel index order = sorted order: element e before e' if e < e'
      const int p0 = fe_block_scan(is_start[0] ? 1 : 0, tid, L.scan_tmp);
      const int t0 = L.scan_tmp[18];
      __syncthreads();
      const int p1 = fe_block_scan(is_start[1] ? 1 : 0, tid, L.scan_tmp);
      const int t1 = L.scan_tmp[18];
      __syncthreads();
      const int ob = L.out_base;
      if (is_start[0]) olf[ob + p0] = cen[0];
      if (is_start[1]) olf[ob + t0 + p1] = cen[1];
      __syncthreads();
      if (tid == 0) L.out_base = ob + t0 + t1;
      (void)starts;
    }
    __syncthreads();
  }
  if (tid == 0) out_counts[scan * 4 + 3] = L.out_base;
}

void launch_frontend(hipStream_t stream, int n_scans, const void* scans, const float4* cloud, const float* range,
                     const unsigned* col, const unsigned char* ground, double scan_period, float4* und, float* diff,
                     int* picks, float4* out_sharp, float4* out_less_sharp, float4* out_flat, float4* out_less_flat,
                     int* out_counts) {
  hipLaunchKernelGGL(frontend_kernel, dim3(n_scans), dim3(kFeBlock), 0, stream, (const FeScan*)scans, cloud, range, col,
                     ground, scan_period, und, diff, picks, out_sharp, out_less_sharp, out_flat, out_less_flat, out_counts);
}
size_t fe_scan_size() { return sizeof(FeScan); }
int fe_pick_stride() { return kFeRows * 6 * kPickStride; }

}  // namespace lins
