// ieskf_lds_tail.hip — the TAIL instantiation of the LDS IESKF kernel (ieskf_lds_impl.h): the last part of every update
// of a large batch as a launch of its own behind the batch kernel's (ieskf_lds_mr.hip) on the same stream.
//
// Why.  After its first few iterations an update is a chain of short phases — de-skew, certificate tests (> 99 % of the
// searches are certified away), rows, reduction, a one-wave solve — that leaves most of a CU idle, and the batch kernel's
// 80 KB of LDS and 8 waves x 128 VGPRs put only two such chains on a CU.  The late iterations need neither: the points
// they touch are the 2-6 tracked candidates of each query, read from the index's grid-sorted copy in L2; the carried
// per-query state (13 words) lives in LDS by QUERY, so a wave takes the queries of one of the head's wave-rounds, then
// of another: 256 threads, < 40 KB of LDS => four scans per CU instead of two.  The rare search that is not certified
// runs through the same loops with (almost) no resident positions — the hybrid fall-through to the sorted copy.
//
// Same code as the head (one template), same query -> lane layout inside a wave-round, same reduction tree and fold
// order: the results are the head's bits (tests/test_gpu_parity.py).
#define LINS_LDS_NS lds_tail
#define LINS_LDS_TAIL 1
#ifndef LINS_TAIL_CAP
#define LINS_TAIL_CAP 448  // resident grid positions: most of the corner cloud (and the Joseph epilogue's 5.5 KB of scratch)
#endif
#define LINS_LDS_CAP LINS_TAIL_CAP
#define LINS_LDS_NMAX 12288
#ifndef LINS_LDS_SCANBATCH
#define LINS_LDS_SCANBATCH 2
#endif
#define LINS_LDS_WAVES 8  // (partial sums: one slot per wave-round of the head)
#ifndef LINS_TAIL_MINW
#define LINS_TAIL_MINW 4
#endif
#define LINS_LDS_MINW LINS_TAIL_MINW
#ifndef LINS_TAIL_LDSBYTES
#define LINS_TAIL_LDSBYTES 40960  // 160 KB / 4
#endif
#define LINS_LDS_BYTES LINS_TAIL_LDSBYTES
#include "ieskf_lds_impl.h"

namespace lins {

int lds_tail_max_queries() { return lds_tail::kTailSlots; }

// `relay`: the hand-over buffers and numbers of the head's launch; this launch continues part relay->launched
void launch_lds_tail(hipStream_t stream, int n, const DevParams& prm, const ScanDesc* descs, const int* order, const float4* arena,
                     const float4* sorted, const GridTables* tabs, const double* state_in, const double* cov_in, double* state_out, double* a6,
                     double* cov_out, void* out, int4* idx_store, lins_pose_record* poses, int scan_id_base, const RelayArgs& relay,
                     unsigned* walk_cache, int run_gen) {
  lds_tail::KernelArgs ka{};
  ka.prm = prm, ka.descs = descs, ka.order = order, ka.tabs = tabs;
  ka.state_in = state_in, ka.cov_in = cov_in, ka.state_out = state_out, ka.a6_out = a6, ka.cov_out = cov_out;
  ka.out = (lds_tail::OutRec*)out, ka.poses = poses, ka.scan_id_base = scan_id_base;
  ka.relay_n = n, ka.relay_at = relay.at, ka.relay_parts = relay.parts, ka.relay_gen = relay.gen, ka.relay_spins = relay.spins;
  ka.relay_hdr = relay.hdr, ka.relay_lane = relay.lane, ka.relay_flag = relay.flag, ka.relay_err = relay.err;
  ka.tail_part = relay.launched, ka.walk_cache = walk_cache, ka.run_gen = run_gen;
  if (prm.pad)
    hipLaunchKernelGGL((lds_tail::ieskf_lds_kernel<256, 1, false, false, false, true>), dim3(n), dim3(256), 0, stream, ka, arena, sorted,
                       idx_store, (lins_corr*)nullptr);
  else
    hipLaunchKernelGGL((lds_tail::ieskf_lds_kernel<256, 1, false, false, false, false>), dim3(n), dim3(256), 0, stream, ka, arena, sorted,
                       idx_store, (lins_corr*)nullptr);
}

}  // namespace lins
