// ieskf_lds_mr.hip — "multi-resident" instantiation of the LDS IESKF kernel (ieskf_lds_impl.h): the
// batch-throughput path.  512-thread workgroups (8 waves: two per SIMD, so the placement of a
// workgroup's waves is balanced whatever SIMD the dispatcher starts on), one owner lane per query (the
// searches of the warm iterations are served by several lanes, see ieskf_lds_impl.h), whose
// LDS holds only the first 4208 grid positions of the scan's prebuilt search index (ieskf_grid.h) — the
// corner cloud and the low surf rings, where nearly every search ends; the rest of the grid is read from
// the index's sorted copy in global memory, which the same loops fall through to.  Batches beyond the
// device's workgroup slots run every update as several workgroups of the launch, which draw their parts by ticket (the kernel's relay).  At < 80 KB of LDS and 128 VGPRs two independent scans are
// resident per CU and fill each other's barriers and serial tails (measured with the HW_ID /
// wall-clock probe of the PROF variant, tools/residency.py).  Results are identical to the
// full-residency kernel: the same loops run over the same grid, only the storage of a position
// differs.
//
// Shapes that were measured and dropped: 384-thread workgroups (6 waves land 2,2,1,1 on the four
// SIMDs, every workgroup starting on the same one, so a second / third workgroup only fits with
// <= 128 / 80 VGPRs even when LDS would allow it); three workgroups per CU at 96 VGPRs (588 B of
// scratch per lane: no faster than two).
#include <algorithm>
#define LINS_LDS_NS lds_mr
// (16-byte point records (x, y, z, original index bits): a candidate is ONE ds_read_b128; round 1's 14-byte SoA layout
// held 4736 instead of 4224 positions but cost four reads per candidate: +2.9 % kernel time, removed in round 3)
#ifndef LINS_MR_CAP
#define LINS_MR_CAP 4208  // (4224 until 16 positions made room for the de-skew's coefficient table)
#endif
#define LINS_LDS_CAP LINS_MR_CAP
#define LINS_LDS_NMAX 12288
#ifndef LINS_LDS_SCANBATCH
#define LINS_LDS_SCANBATCH 2  // (measured: 1 -> 7.9, 2 -> 8.1, 3 -> 8.0, 4 -> 7.75, 8 -> 7.1 M it/s at 128 VGPRs;
                              // re-timed at the end of round 2: 3 -> -0.5 % (noise, 18 spilled registers), 4 -> +2 %)
#endif
#ifndef LINS_MR_BLOCK
#define LINS_MR_BLOCK 512
#endif
#define LINS_LDS_WAVES (LINS_MR_BLOCK / 64)
#define LINS_LDS_BATCH_BLOCK LINS_MR_BLOCK
#ifndef LINS_MR_MINW
#define LINS_MR_MINW 4
#endif
#define LINS_LDS_MINW LINS_MR_MINW
#ifndef LINS_MR_LDSBYTES
#define LINS_MR_LDSBYTES 80896
#endif
#define LINS_LDS_BYTES LINS_MR_LDSBYTES
#ifndef LINS_MR_LEAN
#define LINS_MR_LEAN 1  // (0: the round-5 form of the correspondence phase, per-lane carried state in registers — A/B)
#endif
#if LINS_MR_LEAN
#define LINS_LDS_LEAN 1
#endif
#include "ieskf_lds_impl.h"

namespace lins {

// one KernelArgs by value (ieskf_lds_impl.h): fields not named by a launcher are zero
template <class K>
static void launch_args(K kernel, int grid, int block, hipStream_t stream, const lds_mr::KernelArgs& ka, const float4* arena, const float4* sorted,
                        int4* idx_store, lins_corr* dump = nullptr) {
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, ka, arena, sorted, idx_store, dump);
}

int lds_mr_np_cap() { return lds_mr::kNpMax; }
bool lds_mr_has_parts() { return lds_mr::kLeanBuild; }  // (the several-part updates rest on the carry records of the lean form)
int lds_mr_queue_flags_offset() { return lds_mr::kQFlags; }

// workgroups of the production batch kernel that are resident at once on the current device (larger batches are cut into parts)
int lds_mr_resident_workgroups(int n_cu) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lds_mr::ieskf_lds_kernel<LINS_MR_BLOCK, 1, false, false, false, false>, LINS_MR_BLOCK, 0) != hipSuccess ||
      per_cu < 1)
    per_cu = 2;
  return per_cu * n_cu;
}

// relay != nullptr: every update is cut (relay_next_cut) into relay->parts parts, the launch has one workgroup per (scan, part)
// and `order` lists them (every part 0, then every part 1, ...: scan | part << 27); relay->gen numbers the launch (the
// per-scan flags are never reset: a flag of an earlier launch is smaller)
void launch_lds_mr(hipStream_t stream, int n, const DevParams& prm, const ScanDesc* descs, const int* order, const float4* arena,
                   const float4* sorted, const GridTables* tabs, const double* state_in, const double* cov_in, double* state_out, double* a6,
                   double* cov_out, void* out, int4* idx_store, lins_pose_record* poses, int scan_id_base, long long* prof,
                   const RelayArgs* relay, unsigned* walk_cache, int run_gen, int* carry) {
  // (carry: the carry records of these n scans — 4 x 512 16-byte words each, ieskf_lds_lean.h; every launch needs them)
  lds_mr::KernelArgs ka{};
  ka.relay_lane = carry;
  ka.prm = prm, ka.descs = descs, ka.order = order, ka.tabs = tabs;
  ka.state_in = state_in, ka.cov_in = cov_in, ka.state_out = state_out, ka.a6_out = a6, ka.cov_out = cov_out;
  ka.out = (lds_mr::OutRec*)out, ka.poses = poses, ka.scan_id_base = scan_id_base, ka.prof_buf = prof;
  int grid = n;
  ka.walk_cache = walk_cache, ka.run_gen = run_gen;
  if (relay) {
    ka.relay_n = n, ka.relay_at = relay->at, ka.relay_cuts = relay->cuts, ka.relay_gen = relay->gen, ka.relay_spins = relay->spins, ka.relay_cap = relay->cap;
    ka.relay_hdr = relay->hdr, ka.queue = relay->queue, ka.relay_err = relay->err;
    if (!carry) ka.relay_lane = relay->lane;
    grid = relay->parts * n;
    ka.relay_items = grid;
#if LINS_PERSIST
    grid = std::min(grid, relay->slots);
#endif
  }
  if (prof)
    launch_args(lds_mr::ieskf_lds_kernel<LINS_MR_BLOCK, 1, false, true>, grid, LINS_MR_BLOCK, stream, ka, arena, sorted, idx_store);
  else {
    if (prm.pad)  // (counting aids / test modes asked for: the instantiation that has them)
      launch_args(lds_mr::ieskf_lds_kernel<LINS_MR_BLOCK, 1, false, false, false, true>, grid, LINS_MR_BLOCK, stream, ka, arena, sorted, idx_store);
    else
      launch_args(lds_mr::ieskf_lds_kernel<LINS_MR_BLOCK, 1, false, false, false, false>, grid, LINS_MR_BLOCK, stream, ka, arena, sorted, idx_store);
  }
}

// ICP / Gauss-Newton fallback (estimateTransform, SE:1163-1320) on the same grid and searches:
// state_in = the pose to start from (the filter's), state_out = that state with rn_, qbn_ replaced
void launch_lds_mr_icp(hipStream_t stream, int n, const DevParams& prm, const ScanDesc* descs, const float4* arena,
                       const float4* sorted, const GridTables* tabs, const double* state_in, double* state_out, void* out, int4* idx_store) {
  lds_mr::KernelArgs ka{};
  ka.prm = prm, ka.descs = descs, ka.tabs = tabs;
  ka.state_in = state_in, ka.cov_in = state_in /*unused: no covariance on this path*/, ka.state_out = state_out;
  ka.out = (lds_mr::OutRec*)out;
  launch_args(lds_mr::ieskf_lds_kernel<LINS_MR_BLOCK, 1, false, false, true>, n, LINS_MR_BLOCK, stream, ka, arena, sorted, idx_store);
}

void launch_lds_mr_pass(hipStream_t stream, int n, const DevParams& prm, const ScanDesc* descs, const float4* arena,
                        const float4* sorted, const GridTables* tabs, const double* lin_state, const double* filt_state, int iter,
                        int4* idx_store, lins_corr* dump, double* sums_out, int* counts_out) {
  lds_mr::KernelArgs ka{};
  ka.prm = prm, ka.descs = descs, ka.tabs = tabs;
  ka.state_in = filt_state, ka.lin_in = lin_state, ka.iter_arg = iter;
  ka.sums_out = sums_out, ka.counts_out = counts_out;
  launch_args(lds_mr::ieskf_lds_kernel<LINS_MR_BLOCK, 1, true, false>, n, LINS_MR_BLOCK, stream, ka, arena, sorted, idx_store, dump);
}

}  // namespace lins
