// ieskf_lds.hip — full-residency instantiation of the LDS IESKF kernel (ieskf_lds_impl.h): the
// whole scan (<= 8960 target points) in LDS, one workgroup per CU.  Two shapes:
//   lanes = 3: 1024 threads, three lanes per query (shortest critical path per iteration: the
//              single-scan latency path)
//   lanes = 1: 384 threads, one lane per query  (fewest instructions issued per update)
#define LINS_LDS_NS lds_full
#define LINS_LDS_CAP 8960
#define LINS_LDS_NMAX 8960
#ifndef LINS_LDS_SCANBATCH
#define LINS_LDS_SCANBATCH 4
#endif
// rows -> 28 sums in registers (no row slots in LDS, two barriers fewer per round) and 16-byte point records: the
// 19 KB of slots the register reduction frees pay for the 2 extra bytes per point (-4 % / -7 % batch time for the
// 3-lane / 1-lane shape, single-scan update 181 -> 170 us)
#define LINS_LDS_REGREDUCE 1
#define LINS_LDS_WAVES 16
#define LINS_LDS_AOS 1
#define LINS_LDS_MINW 1
#define LINS_LDS_BYTES 163840
#include "ieskf_lds_impl.h"

namespace lins {

#define LINS_LAUNCH(NS, B, LN, PR)                                                                                  \
  hipLaunchKernelGGL((NS::ieskf_lds_kernel<B, LN, false, PR>), dim3(n), dim3(B), 0, stream, prm, descs, arena, sorted, \
                     state_in, cov_in, (const double*)nullptr, 0, state_out, a6, (NS::OutRec*)out, idx_store, poses,  \
                     scan_id_base, (lins_corr*)nullptr, (double*)nullptr, (int*)nullptr, prof)
#define LINS_LAUNCH_PASS(NS, B, LN)                                                                                    \
  hipLaunchKernelGGL((NS::ieskf_lds_kernel<B, LN, true, false>), dim3(n), dim3(B), 0, stream, prm, descs, arena, sorted, \
                     filt_state, (const double*)nullptr, lin_state, iter, (double*)nullptr, (double*)nullptr,           \
                     (NS::OutRec*)nullptr, idx_store, (lins_pose_record*)nullptr, 0, dump, sums_out, counts_out,        \
                     (long long*)nullptr)

int lds_np_cap() { return lds_full::kNpMax; }

void launch_lds(hipStream_t stream, int n, const DevParams& prm, int lanes, const ScanDesc* descs,
                const float4* arena, const double* state_in, const double* cov_in, double* state_out, double* a6,
                void* out, int4* idx_store, lins_pose_record* poses, int scan_id_base, long long* prof) {
  float4* sorted = nullptr;
  if (lanes == 3) {
    if (prof)
      LINS_LAUNCH(lds_full, 1024, 3, true);
    else
      LINS_LAUNCH(lds_full, 1024, 3, false);
  } else {
    if (prof)
      LINS_LAUNCH(lds_full, 384, 1, true);
    else
      LINS_LAUNCH(lds_full, 384, 1, false);
  }
}

void launch_lds_pass(hipStream_t stream, int n, const DevParams& prm, int lanes, const ScanDesc* descs,
                     const float4* arena, const double* lin_state, const double* filt_state, int iter,
                     int4* idx_store, lins_corr* dump, double* sums_out, int* counts_out) {
  float4* sorted = nullptr;
  if (lanes == 3)
    LINS_LAUNCH_PASS(lds_full, 1024, 3);
  else
    LINS_LAUNCH_PASS(lds_full, 384, 1);
}

}  // namespace lins
