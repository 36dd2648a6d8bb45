// ieskf_grid.hip — grid_index_kernel: the search index of a scan's target clouds, built once where the clouds
// arrive (the reference's kdtree->setInputCloud, SE:1156-1160 — see ieskf_grid.h).  One 512-thread workgroup per
// scan: a single coalesced pass over the scan's targets for the histogram and the rings' elevation wedges, a block scan
// over the 3072 cells, a second pass (L2 hits) for the scatter; the sorted copy and the tables go to HBM.
#include <hip/hip_runtime.h>

#include "ieskf_grid.h"
#include "ieskf_rowsum.h"

#ifndef LINS_GRID_PF
#define LINS_GRID_PF 8  // steps of the grid build whose point reads are in flight together, in the histogram pass (measured
                        // inside the update kernel: 1 -> 4: -0.9 %, 8: as 4) and in the scatter pass (-0.5 %)
#endif

namespace lins {
namespace {

// (1024-thread workgroups halve the registers per thread — 62 instead of 110 VGPRs, twice the waves per CU — and change
// nothing: 0.091 against 0.093 ms per 1024 scans, tools/index_time.py; the build moves 2 x 143 MB in that time, 3.1 TB/s,
// and is held up by its two dependent passes and their LDS atomics, not by occupancy)
#ifndef LINS_GRID_BLOCK
#define LINS_GRID_BLOCK 512
#endif
constexpr int kGBlock = LINS_GRID_BLOCK;

struct GridLds {
  GridTables gt;
  int el_bits[2][kRingsBinned][2];
  int scan_tmp[kGBlock / 64 + 4];
};

// REPROJECT (round 4, the reference's updatePointCloud as ONE pass: SE:1116-1161): the clouds are the scan's less-sharp /
// less-flat clouds as the front-end left them; every point is taken to the scan end with the stream's final state
// (transformToEnd, SE:1083-1101) where the histogram pass reads it, written back in place — the next update's rows and
// the any-size kernels read the arena — and binned as what it has become.  One read of the new clouds instead of two
// kernels' two (lins_streams_step ran reproject_in_place_kernel, then this kernel at the start of the next step).
template <bool REPROJECT>
__global__ __launch_bounds__(kGBlock) void grid_index_kernel(const ScanDesc* __restrict__ descs, const float4* arena, float4* arena_rw,
                                                             float4* __restrict__ sorted, GridTables* __restrict__ tab,
                                                             const double* __restrict__ states, double inv_period) {
  __shared__ GridLds L;
  const int tid = threadIdx.x;
  const ScanDesc sd = descs[blockIdx.x];
  ToEnd te;
  if (REPROJECT) {
    const double* st = states + (size_t)blockIdx.x * 19;
    te = make_to_end(V3{st[0], st[1], st[2]}, Q4{st[6], st[7], st[8], st[9]}, inv_period);
  }
  float4* const gsorted = sorted + sd.off_surf_t;
  // cell c's counter is the u16 half (c & 1) of word c >> 1 of cell_end: counts and positions stay
  // below 2^16, so a half never carries into its neighbour
  unsigned* cnt32 = L.gt.cell_word;
  constexpr int ncell = kCellsSurf + kCellsCorner;
  static_assert(ncell % 2 == 0, "two counters per word");
  for (int c = tid; c < ncell / 2; c += kGBlock) cnt32[c] = 0;
  if (tid < 2 * kRingsBinned) {
    L.el_bits[tid / kRingsBinned][tid % kRingsBinned][0] = 0x7FFFFFFF;
    L.el_bits[tid / kRingsBinned][tid % kRingsBinned][1] = (int)0x80000000;
  }
  if (tid < 2) L.gt.pad[tid] = 0;
  __syncthreads();
  const float4* ts = arena + sd.off_surf_t;
  const float4* tc = arena + sd.off_corner_t;
  const int n_all = sd.n_surf_t + sd.n_corner_t;
  // Ownership: wave w owns the points [w * 64 P, (w + 1) * 64 P), P = ceil(n_all / BLOCK), lane l of it the points
  // w * 64 P + 64 k + l — a wave's loads are coalesced, and its consecutive steps are consecutive 64-point runs of the
  // ring-sorted cloud, so the (cloud, ring) key of a step changes every ~7 steps instead of every step: see the
  // elevation wedge below.  The cell of each point is kept in a register between the histogram pass and the scatter
  // pass (no second atan2, no second classification).
  constexpr int kPerThread = (kGridNpMax + kGBlock - 1) / kGBlock;
  const int per_lane = (n_all + kGBlock - 1) / kGBlock;
  const int lane = tid & 63, j_first = (tid >> 6) * 64 * per_lane + lane;
  int cell_of[kPerThread];
  // Elevation wedge of a ring (min / max of z / rho over its points; atan is monotone, so the wedge is the atan of the
  // extreme ratios — two atanf per ring at the end instead of an atan2f per point).  The 64 points of a wave's step
  // nearly always lie on one ring of one cloud, and so do the steps before and after: every lane folds its points of
  // the current RUN of equal keys into two registers, and only when the key changes (or at the end) does the wave
  // fold the 64 partial results (six cross-lane steps) and ONE lane update LDS.
  int run_key = -1, run_lo = 0x7FFFFFFF, run_hi = (int)0x80000000;
  auto flush_run = [&]() {
    if (run_key >= 0) {  // (wave-uniform)
      int lo = run_lo, hi = run_hi;
      lo = min(lo, xor_lane_i32<1>(lo, lane)), hi = max(hi, xor_lane_i32<1>(hi, lane));
      lo = min(lo, xor_lane_i32<2>(lo, lane)), hi = max(hi, xor_lane_i32<2>(hi, lane));
      lo = min(lo, xor_lane_i32<4>(lo, lane)), hi = max(hi, xor_lane_i32<4>(hi, lane));
      lo = min(lo, xor_lane_i32<8>(lo, lane)), hi = max(hi, xor_lane_i32<8>(hi, lane));
      lo = min(lo, xor_lane_i32<16>(lo, lane)), hi = max(hi, xor_lane_i32<16>(hi, lane));
      lo = min(lo, xor_lane_i32<32>(lo, lane)), hi = max(hi, xor_lane_i32<32>(hi, lane));
      if (lane == 0) {
        atomicMin(&L.el_bits[run_key / kRingsBinned][run_key % kRingsBinned][0], lo);
        atomicMax(&L.el_bits[run_key / kRingsBinned][run_key % kRingsBinned][1], hi);
      }
    }
    run_lo = 0x7FFFFFFF, run_hi = (int)0x80000000;
  };
  // The point reads of LINS_GRID_PF consecutive steps are issued together (clamped addresses, no branch in between):
  // a wave pays the HBM latency once per chunk instead of once per step.
  constexpr int kChunk = LINS_GRID_PF;
#pragma unroll
  for (int k0 = 0; k0 < kPerThread; k0 += kChunk) {
    float4 pbuf[kChunk];
    if (k0 < per_lane) {  // (wave-uniform)
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const int j = j_first + (k0 + u) * 64, jc = j < n_all ? j : n_all - 1;
        pbuf[u] = jc < sd.n_surf_t ? ts[jc] : tc[jc - sd.n_surf_t];
      }
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int k = k0 + u;
      if (k >= kPerThread) break;
      const int j = j_first + k * 64;
      cell_of[k] = -1;
      if (k < per_lane) {  // (wave-uniform)
        int eb = 0, ek = -1;  // elevation bits, (cloud, ring) key of this lane's point
        if (j < n_all) {
          const bool is_s = j < sd.n_surf_t;
          float4 p = pbuf[u];
          if (REPROJECT) {
            const V3 e = to_end_point(te, (double)p.x, (double)p.y, (double)p.z, p.w);
            p.x = (float)e.x, p.y = (float)e.y, p.z = (float)e.z;
            (is_s ? arena_rw + sd.off_surf_t : arena_rw + sd.off_corner_t - sd.n_surf_t)[j] = p;
          }
          const int r = ring_of(p.w), naz = is_s ? kAzSurf : kAzCorner;
          const int cell = (is_s ? kCellsCorner : 0) + r * naz + az_bin_lds(p.x, p.y, naz);
          cell_of[k] = cell;
          atomicAdd(&cnt32[cell >> 1], 1u << ((cell & 1) * 16));
          // z / rho through the hardware's reciprocal square root (1 ulp): the ratio is off by < 2^-22 relative, the
          // elevation by < 1.2e-7 rad — two orders below kSlack.  rho = 0: +-inf / 0 (atanf gives +-pi/2 / 0); a
          // ratio that overflows to +-inf widens the wedge, never narrows it.
          const float rho2 = p.x * p.x + p.y * p.y;
          eb = ordered_int(rho2 > 0.f ? p.z * __frsqrt_rn(rho2) : (p.z > 0.f ? INFINITY : (p.z < 0.f ? -INFINITY : 0.f)));
          ek = (is_s ? 0 : kRingsBinned) + r;
        }
        const unsigned long long have = __ballot(ek >= 0);
        if (have) {  // (wave-uniform)
          const int ek0 = __builtin_amdgcn_readlane(ek, __ffsll((long long)have) - 1);
          if (__all(ek == ek0 || ek < 0)) {
            if (ek0 != run_key) {
              flush_run();
              run_key = ek0;
            }
            if (ek >= 0) run_lo = min(run_lo, eb), run_hi = max(run_hi, eb);
          } else {  // a step that straddles rings: its lanes update LDS themselves
            flush_run();
            run_key = -1;
            if (ek >= 0) {
              atomicMin(&L.el_bits[ek / kRingsBinned][ek % kRingsBinned][0], eb);
              atomicMax(&L.el_bits[ek / kRingsBinned][ek % kRingsBinned][1], eb);
            }
          }
        }
      }
    }
  }
  flush_run();
  __syncthreads();
  // exclusive scan over all cells: corner cells first, so surf positions start at n_corner_t
  constexpr int per = (ncell + kGBlock - 1) / kGBlock;
  const int c_lo = tid * per < ncell ? tid * per : ncell;
  const int c_hi = c_lo + per < ncell ? c_lo + per : ncell;
  int local = 0;
  for (int c = c_lo; c < c_hi; ++c) local += (int)L.gt.cell_end[c];
  int run = block_exclusive_scan(local, tid, L.scan_tmp);
  for (int c = c_lo; c < c_hi; ++c) {
    int n = (int)L.gt.cell_end[c];
    L.gt.cell_end[c] = (unsigned short)run;  // (16-bit stores: neighbours in the same word are not touched)
    run += n;
  }
  __syncthreads();
  // (the second read of the points — L2 hits — chunked like the first: one round trip per chunk, not per step)
#pragma unroll
  for (int k0 = 0; k0 < kPerThread; k0 += kChunk) {
    float4 pbuf[kChunk];
    if (k0 < per_lane) {  // (wave-uniform)
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        const int j = j_first + (k0 + u) * 64, jc = j < n_all ? j : n_all - 1;
        pbuf[u] = jc < sd.n_surf_t ? ts[jc] : tc[jc - sd.n_surf_t];
      }
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int k = k0 + u;
      if (k >= kPerThread) break;
      const int j = j_first + k * 64;
      if (cell_of[k] >= 0) {
        const bool is_s = j < sd.n_surf_t;
        const int jj = is_s ? j : j - sd.n_surf_t;
        const float4 p = pbuf[u];
        const int cell = cell_of[k], sh = (cell & 1) * 16;
        // order inside a cell is irrelevant (keyed ties)
        const int pos = (int)((atomicAdd(&cnt32[cell >> 1], 1u << sh) >> sh) & 0xFFFFu);
        gsorted[pos] = make_float4(p.x, p.y, p.z, __int_as_float(jj));
      }
    }
  }
  __syncthreads();  // cell_end[c] is now the exclusive END of cell c
  if (tid <= kRingsBinned) {
    L.gt.ring_start[1][tid] = tid == 0 ? 0 : (int)L.gt.cell_end[tid * kAzCorner - 1];
    L.gt.ring_start[0][tid] = (int)L.gt.cell_end[kCellsCorner + tid * kAzSurf - 1] - sd.n_corner_t;
  }
  if (tid < 2 * kRingsBinned) {
    int cl = tid / kRingsBinned, r = tid % kRingsBinned;
    // (atanf of the ratio vs the queries' atan2f(z, rho): a few ulp of an angle below 0.3 rad, ~1e-7 — inside kSlack)
    float lo = atanf(ordered_float(L.el_bits[cl][r][0])) - kSlack, hi = atanf(ordered_float(L.el_bits[cl][r][1])) + kSlack;
    const bool empty = L.el_bits[cl][r][0] == 0x7FFFFFFF;  // no point touched the ring's min/max
    L.gt.el_ang[cl][r] = empty ? make_float2(INFINITY, -INFINITY) : make_float2(lo, hi);
  }
  __syncthreads();
  const uint4* src = reinterpret_cast<const uint4*>(&L.gt);
  uint4* dst = reinterpret_cast<uint4*>(tab + blockIdx.x);
  for (int k = tid; k < kGridTableWords; k += kGBlock) dst[k] = src[k];
}

}  // namespace

void launch_grid_index(hipStream_t stream, int n, const ScanDesc* descs, const float4* arena, float4* gsorted,
                       GridTables* tab) {
  if (n <= 0) return;
  hipLaunchKernelGGL(grid_index_kernel<false>, dim3(n), dim3(kGBlock), 0, stream, descs, arena, nullptr, gsorted, tab, nullptr, 0.0);
}
void launch_reproject_and_index(hipStream_t stream, int n, const ScanDesc* descs, float4* arena, float4* gsorted, GridTables* tab,
                                const double* states, double inv_period) {
  if (n <= 0) return;
  hipLaunchKernelGGL(grid_index_kernel<true>, dim3(n), dim3(kGBlock), 0, stream, descs, arena, arena, gsorted, tab, states, inv_period);
}

}  // namespace lins
