// ieskf_lds.hip — full-residency instantiation of the LDS IESKF kernel (ieskf_lds_impl.h): the
// whole scan in LDS when it has <= 8960 target points (what a VLP-16 front-end emits on most scans), one workgroup
// per CU; the few positions beyond that — up to 12288 — live in the grid-sorted global copy the same loops fall
// through to (the high surf rings, rarely searched), so a large scan no longer falls back to the global-memory
// kernel (round 3: the clean range images of the new generator carry ~8700 target points on average).  Two shapes:
//   lanes = 3: 1024 threads, three lanes per query (shortest critical path per iteration: the
//              single-scan latency path)
//   lanes = 1: 384 threads, one lane per query  (fewest instructions issued per update)
#define LINS_LDS_NS lds_full
#define LINS_LDS_CAP 8960
#define LINS_LDS_NMAX 12288
#ifndef LINS_LDS_SCANBATCH
#define LINS_LDS_SCANBATCH 4
#endif
#define LINS_LDS_WAVES 16
#define LINS_LDS_MINW 1
#define LINS_LDS_BYTES 163840
#ifndef LINS_FULL_LEAN
// (1: the one-lane shape's update kernels in the register-lean form of ieskf_lds_lean.h.  Round 6 measured the single-scan
// update through a lean 1024 x 1 shape — sixteen waves, up to eight lanes per cold search: 216 us against 182 us of the
// three-lane shape and 188 us of the batch kernel's 512 x 1 on one scan: more waves, more barrier and per-wave fixed cost.)
#define LINS_FULL_LEAN 0
#endif
#if LINS_FULL_LEAN
#define LINS_LDS_LEAN 1
#endif
#ifndef LINS_FULL_BLOCK1
#define LINS_FULL_BLOCK1 384  // threads of the one-lane-per-query shape
#endif
#include "ieskf_lds_impl.h"

namespace lins {

template <class K>
static void launch_args(K kernel, int grid, int block, hipStream_t stream, const lds_full::KernelArgs& ka, const float4* arena, const float4* sorted,
                        int4* idx_store, lins_corr* dump = nullptr) {
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, ka, arena, sorted, idx_store, dump);
}

int lds_np_cap() { return lds_full::kNpMax; }

// (one workgroup per CU: nothing to order; the batch shape's several-part updates do not exist here)
void launch_lds(hipStream_t stream, int n, const DevParams& prm, int lanes, const ScanDesc* descs,
                const float4* arena, const float4* sorted, const GridTables* tabs, const double* state_in, const double* cov_in, double* state_out, double* a6,
                double* cov_out, void* out, int4* idx_store, lins_pose_record* poses, int scan_id_base, long long* prof, int* carry) {
  lds_full::KernelArgs ka{};
  ka.relay_lane = carry;  // (the carry records of these n scans: the one-lane shape's per-query state, ieskf_lds_lean.h)
  ka.prm = prm, ka.descs = descs, ka.tabs = tabs;
  ka.state_in = state_in, ka.cov_in = cov_in, ka.state_out = state_out, ka.a6_out = a6, ka.cov_out = cov_out;
  ka.out = (lds_full::OutRec*)out, ka.poses = poses, ka.scan_id_base = scan_id_base, ka.prof_buf = prof;
  if (lanes == 3) {
    if (prof)
      launch_args(lds_full::ieskf_lds_kernel<1024, 3, false, true>, n, 1024, stream, ka, arena, sorted, idx_store);
    else if (prm.pad)
      launch_args(lds_full::ieskf_lds_kernel<1024, 3, false, false, false, true>, n, 1024, stream, ka, arena, sorted, idx_store);
    else
      launch_args(lds_full::ieskf_lds_kernel<1024, 3, false, false, false, false>, n, 1024, stream, ka, arena, sorted, idx_store);
  } else {
    if (prof)
      launch_args(lds_full::ieskf_lds_kernel<LINS_FULL_BLOCK1, 1, false, true>, n, LINS_FULL_BLOCK1, stream, ka, arena, sorted, idx_store);
    else
      launch_args(lds_full::ieskf_lds_kernel<LINS_FULL_BLOCK1, 1, false, false>, n, LINS_FULL_BLOCK1, stream, ka, arena, sorted, idx_store);
  }
}

void launch_lds_pass(hipStream_t stream, int n, const DevParams& prm, int lanes, const ScanDesc* descs,
                     const float4* arena, const float4* sorted, const GridTables* tabs, const double* lin_state, const double* filt_state, int iter,
                     int4* idx_store, lins_corr* dump, double* sums_out, int* counts_out) {
  lds_full::KernelArgs ka{};
  ka.prm = prm, ka.descs = descs, ka.tabs = tabs;
  ka.state_in = filt_state, ka.lin_in = lin_state, ka.iter_arg = iter;
  ka.sums_out = sums_out, ka.counts_out = counts_out;
  if (lanes == 3)
    launch_args(lds_full::ieskf_lds_kernel<1024, 3, true, false>, n, 1024, stream, ka, arena, sorted, idx_store, dump);
  else
    launch_args(lds_full::ieskf_lds_kernel<384, 1, true, false>, n, 384, stream, ka, arena, sorted, idx_store, dump);
}

}  // namespace lins
