// ieskf_lds_mr.hip — "multi-resident" instantiation of the LDS IESKF kernel (ieskf_lds_impl.h): the
// batch-throughput path.  512-thread workgroups (8 waves: two per SIMD, so the placement of a
// workgroup's waves is balanced whatever SIMD the dispatcher starts on), one owner lane per query (the
// searches of the warm iterations are served by several lanes, see ieskf_lds_impl.h), whose
// LDS holds only the first 4208 grid positions of the scan's prebuilt search index (ieskf_grid.h) — the
// corner cloud and the low surf rings, where nearly every search ends; the rest of the grid is read from
// the index's sorted copy in global memory, which the same loops fall through to.  Batches beyond the
// device's workgroup slots run every update as consecutive workgroups of the launch (the kernel's relay).  At < 80 KB of LDS and 128 VGPRs two independent scans are
// resident per CU and fill each other's barriers and serial tails (measured with the HW_ID /
// wall-clock probe of the PROF variant, tools/residency.py).  Results are identical to the
// full-residency kernel: the same loops run over the same grid, only the storage of a position
// differs.
//
// Shapes that were measured and dropped: 384-thread workgroups (6 waves land 2,2,1,1 on the four
// SIMDs, every workgroup starting on the same one, so a second / third workgroup only fits with
// <= 128 / 80 VGPRs even when LDS would allow it); three workgroups per CU at 96 VGPRs (588 B of
// scratch per lane: no faster than two).
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
#define LINS_LDS_WAVES 8
#ifndef LINS_MR_MINW
#define LINS_MR_MINW 4
#endif
#define LINS_LDS_MINW LINS_MR_MINW
#define LINS_MR_BLOCK 512
#ifndef LINS_MR_LDSBYTES
#define LINS_MR_LDSBYTES 80896
#endif
#define LINS_LDS_BYTES LINS_MR_LDSBYTES
#include "ieskf_lds_impl.h"

namespace lins {

#define LINS_LAUNCH(NS, B, LN, PR)                                                                                  \
  hipLaunchKernelGGL((NS::ieskf_lds_kernel<B, LN, false, PR>), dim3(grid), dim3(B), 0, stream, prm, descs, order, arena, sorted, tabs, \
                     state_in, cov_in, (const double*)nullptr, 0, state_out, a6, cov_out, (NS::OutRec*)out, idx_store, poses,  \
                     scan_id_base, (lins_corr*)nullptr, (double*)nullptr, (int*)nullptr, prof, RELAY_ARGS)
#define LINS_LAUNCH_PASS(NS, B, LN)                                                                                    \
  hipLaunchKernelGGL((NS::ieskf_lds_kernel<B, LN, true, false>), dim3(n), dim3(B), 0, stream, prm, descs, order, arena, sorted, tabs, \
                     filt_state, (const double*)nullptr, lin_state, iter, (double*)nullptr, (double*)nullptr,           \
                     (double*)nullptr, (NS::OutRec*)nullptr, idx_store, (lins_pose_record*)nullptr, 0, dump, sums_out, counts_out,        \
                     (long long*)nullptr, 0, 0, 0, 0, (double*)nullptr, (int*)nullptr, (int*)nullptr)

int lds_mr_np_cap() { return lds_mr::kNpMax; }

// relay_hdr != nullptr: every update is cut every relay_at iterations into relay_parts workgroups of the launch, see
// the kernel; relay_gen numbers the launch (the per-scan flags are never reset: a flag of an earlier launch is smaller)
void launch_lds_mr(hipStream_t stream, int n, const DevParams& prm, const ScanDesc* descs, const int* order, const float4* arena,
                   const float4* sorted, const GridTables* tabs, const double* state_in, const double* cov_in, double* state_out, double* a6,
                   double* cov_out, void* out, int4* idx_store, lins_pose_record* poses, int scan_id_base, long long* prof,
                   int relay_at, int relay_parts, int relay_gen, double* relay_hdr, int* relay_lane, int* relay_flag) {
  const int relay_n = relay_hdr ? n : 0, grid = relay_n ? relay_parts * n : n;
#define RELAY_ARGS relay_n, relay_at, relay_parts, relay_gen, relay_hdr, relay_lane, relay_flag
  if (prof)
    LINS_LAUNCH(lds_mr, LINS_MR_BLOCK, 1, true);
  else
    LINS_LAUNCH(lds_mr, LINS_MR_BLOCK, 1, false);
}

// ICP / Gauss-Newton fallback (estimateTransform, SE:1163-1320) on the same grid and searches:
// state_in = the pose to start from (the filter's), state_out = that state with rn_, qbn_ replaced
void launch_lds_mr_icp(hipStream_t stream, int n, const DevParams& prm, const ScanDesc* descs, const float4* arena,
                       const float4* sorted, const GridTables* tabs, const double* state_in, double* state_out, void* out, int4* idx_store) {
  hipLaunchKernelGGL((lds_mr::ieskf_lds_kernel<512, 1, false, false, true>), dim3(n), dim3(512), 0, stream, prm, descs,
                     (const int*)nullptr, arena, sorted, tabs, state_in, state_in /*unused: no covariance on this path*/, (const double*)nullptr, 0,
                     state_out, (double*)nullptr, (double*)nullptr, (lds_mr::OutRec*)out, idx_store, (lins_pose_record*)nullptr, 0,
                     (lins_corr*)nullptr, (double*)nullptr, (int*)nullptr, (long long*)nullptr, 0, 0, 0, 0, (double*)nullptr,
                     (int*)nullptr, (int*)nullptr);
}

void launch_lds_mr_pass(hipStream_t stream, int n, const DevParams& prm, const ScanDesc* descs, const float4* arena,
                        const float4* sorted, const GridTables* tabs, const double* lin_state, const double* filt_state, int iter,
                        int4* idx_store, lins_corr* dump, double* sums_out, int* counts_out) {
  const int* order = nullptr;
  LINS_LAUNCH_PASS(lds_mr, 512, 1);
}

}  // namespace lins
