// ieskf_grid.h — the search index of one scan's target clouds, as it lies in HBM between the kernel that
// builds it and the update kernels that use it.
//
// The reference builds its search index — two kd-trees over the last scan's less-sharp / less-flat clouds — where
// those clouds are produced: kdtreeCorner_/kdtreeSurf_->setInputCloud at the end of updatePointCloud() (SE:1156-1160;
// SE:363-364 for the first scan), i.e. once per target cloud and OUTSIDE performIESKF (SE:465-600), which only
// queries them (SE:847, 973).  The device path has the same split since the end of round 3:
//
//   grid_index_kernel (ieskf_grid.hip)   = setInputCloud: both target clouds of a scan counting-sorted into the
//                                          (ring x azimuth-column) grid — a grid-sorted copy of the points as
//                                          16-byte records (x, y, z, original index bits) + the tables below;
//                                          run where the target clouds arrive (lins_batch_upload, a chunk's copy of
//                                          lins_ieskf_update_batch, a streams step);
//   ieskf_lds_kernel (ieskf_lds_impl.h)  = performIESKF: loads the tables and the first kNpCap records into LDS
//                                          (one coalesced pass) and iterates.
//
// Rounds 1-3 built the grid inside the update kernel (histogram, block scan, scatter: 96 us per 1024-scan launch
// at two workgroups per CU); the stand-alone build runs at whatever occupancy its 7 KB of LDS allow and is paid
// once per target cloud however often the cloud is searched (update, ICP fallback, correspondence passes).
#pragma once

#include "ieskf_binned.h"
#include "ieskf_device.h"

namespace lins {

constexpr int kCellsSurf = kRingsBinned * kAzSurf, kCellsCorner = kRingsBinned * kAzCorner;
constexpr int kGridNpMax = 12288;  // target points of a scan the grid kernels take (positions and indices are u16)

// Tables of one scan's grid: what the searches read besides the points.  The same bytes in HBM (written by
// grid_index_kernel) and in LDS (the update kernels copy them with 16-byte moves).
struct alignas(16) GridTables {
  // exclusive end (absolute grid position) per cell, corner cells first; during the build the same
  // words are the histogram / scatter counters (two u16 counters per 32-bit LDS atomic).  A union,
  // and the library is built with -fno-strict-aliasing: the 16- and 32-bit views DO alias.
  union {
    unsigned short cell_end[kCellsCorner + kCellsSurf];
    unsigned cell_word[(kCellsCorner + kCellsSurf) / 2];
  };
  float2 el_ang[2][kRingsBinned];       // elevation wedge of each ring as angles (lo - slack, hi + slack);
                                        // an empty ring gets (+inf, -inf): never within reach
  int ring_start[2][kRingsBinned + 1];  // per cloud (0 = surf, 1 = corner), in (original) index space
  int pad[2];
};
static_assert(sizeof(GridTables) % 16 == 0, "copied as 16-byte words");
constexpr int kGridTableWords = (int)(sizeof(GridTables) / 16);

// The grid only has to be CONSISTENT: a point within angular distance D of a query must sit within the query's
// window of +-K columns, K = reach().  With columns of width w and a column function floor(g(theta) / w) whose angle
// g is off by at most eps, |g(p) - g(q)| <= D + 2 eps, so the two columns differ by at most floor((D + 2 eps) / w) + 1
// <= floor(D / w) + 2 as long as 2 eps < w — which is what reach() adds (its "+ 2": one column for the query's offset
// inside its own column, one for rounding).  The finest columns are 2 pi / 128 = 0.049 rad wide, so an angle good to
// 4e-3 rad is enough: lins_atan2_coarse (a dozen instructions, lins_math.h) instead of atan2f (~45) for the
// points of the build and for every query of every iteration.  Results cannot change: every pruning decision stays a
// superset decision, and ties are resolved on explicit keys.  (One definition for the build and the queries.)
static_assert((kAzSurf & (kAzSurf - 1)) == 0 && (kAzCorner & (kAzCorner - 1)) == 0, "column counts are powers of two");
__device__ __forceinline__ int az_bin_lds(float x, float y, int naz) {
  int a = (int)((lins_atan2_coarse(y, x) + kPiF) * ((float)naz * (0.5f / kPiF)));
  return a < 0 ? 0 : (a >= naz ? naz - 1 : a);
}

// the build of scans [0, n) of `descs` on `stream`: sorted copy into gsorted (same offsets as the targets in the
// arena: positions 0 .. n_all - 1 at off_surf_t), tables into tab[scan]
// updatePointCloud in one kernel: the clouds `descs` name as targets are re-projected in place with states[scan] (19
// doubles each: t at 0, q at 6) and indexed as re-projected (ieskf_grid.hip grid_index_kernel<true>)
void launch_reproject_and_index(hipStream_t stream, int n, const ScanDesc* descs, float4* arena, float4* gsorted, GridTables* tab,
                                const double* states, double inv_period);
void launch_grid_index(hipStream_t stream, int n, const ScanDesc* descs, const float4* arena, float4* gsorted,
                       GridTables* tab);

}  // namespace lins
