// ieskf_host.cpp — performIESKF() as lins_fusion_node sees it: the GPU IESKF
// loop, and when the filter diverges the reference's fallback
// (StateEstimator.hpp:585-592): estimateTransform (SE:1163-1196) = up to
// NUM_ITER rounds of {correspondence pass, 6-DoF Gauss-Newton step
// calculateTransformation (SE:1198-1320)} from the filter's pose, covariance
// left un-updated — one device kernel (lins_icp_update_batch).
//
// Clouds the grid kernels cannot take (unsorted rings, ring ids >= 16) keep the
// earlier split: correspondences + rows from the exhaustive HIP kernels
// (lins_correspondences()), the 6x6 Gauss-Newton step (icp_math.h, the same code
// the device tail runs) on the host.

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../../include/lins_host.h"
#include "../icp_math.h"
#include "../lins_ctx_priv.h"
#include "../lins_math.h"

using namespace lins;

namespace {

// calculateTransformation (SE:1198-1320); true = converged
bool gauss_newton_step(const lins_params& prm, double* t, Q4& q, const lins_scan_pair& in,
                       const std::vector<lins_corr>& cs, const std::vector<lins_corr>& cc, int iter) {
  double JTJ[36] = {0}, JTb[6] = {0};
  const V3 phi = quat2axis(q);
  const double inv_period = (double)(1.f / prm.scan_period);
  auto add_row = [&](const lins_point& kp, const lins_corr& c) {
    double J[6], b;
    icp_row(inv_period, phi, kp.x, kp.y, kp.z, kp.intensity, c.coeff, J, b);
    for (int a = 0; a < 6; ++a) {
      for (int e = 0; e < 6; ++e) JTJ[a * 6 + e] += J[a] * J[e];
      JTb[a] += J[a] * b;
    }
  };
  for (int i = 0; i < in.n_surf_flat; ++i)
    if (cs[i].accepted) add_row(lins_point_load(in.surf_flat, in.point_stride_bytes, i), cs[i]);
  for (int i = 0; i < in.n_corner_sharp; ++i)
    if (cc[i].accepted) add_row(lins_point_load(in.corner_sharp, in.point_stride_bytes, i), cc[i]);
  double x[6];
  double ws[kIcpWorkspace];
  icp_gn_solve(JTJ, JTb, iter, x, ws);
  return icp_apply(x, t, q);
}

}  // namespace

extern "C" int lins_host_perform_ieskf(lins_ctx* ctx, const lins_params* prm, const lins_scan_pair* in,
                                        lins_result* out, int32_t* used_icp_fallback) {
  if (!ctx || !prm || !in || !out) return LINS_E_ARG;
  // the device loop runs with the parameters the context was created with; the host half of the fallback must not
  // run with others (the argument exists for callers that keep their own copy: it has to be the same values)
  if (std::memcmp(prm, lins::ctx_params(ctx), sizeof(lins_params)) != 0) return LINS_E_ARG;
  if (used_icp_fallback) *used_icp_fallback = 0;
  int rc = lins_ieskf_update(ctx, in, out);
  if (rc != LINS_OK || !out->diverged) return rc;
  // ---- "======Using ICP Method======" (SE:585-592) --------------------------
  if (used_icp_fallback) *used_icp_fallback = 1;
  // the whole fallback in one kernel (LINS_ICP_HOST=1, a test aid, forces the split path below)
  const char* force_host = std::getenv("LINS_ICP_HOST");
  if (!(force_host && force_host[0] == '1')) {
    lins_result icp;
    rc = lins_icp_update_batch(ctx, 1, in, &icp);
    if (rc == LINS_OK) {  // keep what the filter reported (iters, diverged, norms); pose from the ICP
      std::memcpy(out->state, icp.state, sizeof icp.state);
      std::memcpy(out->cov, in->cov, sizeof in->cov);  // Pk_ un-updated
      return LINS_OK;
    }
    if (rc != LINS_E_UNSUPPORTED) return rc;
  }
  // clouds the grid kernels cannot take: device correspondences + host Gauss-Newton
  double lin[LINS_STATE_DIM];
  std::memcpy(lin, in->state, sizeof lin);
  double t[3] = {lin[0], lin[1], lin[2]};
  Q4 q{lin[6], lin[7], lin[8], lin[9]};
  std::vector<lins_corr> cs(std::max(in->n_surf_flat, 1)), cc(std::max(in->n_corner_sharp, 1));
  for (int iter = 0; iter < prm->num_iter; ++iter) {
    lin[0] = t[0], lin[1] = t[1], lin[2] = t[2];
    lin[6] = q.w, lin[7] = q.x, lin[8] = q.y, lin[9] = q.z;
    rc = lins_correspondences(ctx, in, lin, iter, cs.data(), cc.data());
    if (rc != LINS_OK) return rc;
    int ms = 0, mc = 0;
    for (int i = 0; i < in->n_surf_flat; ++i) ms += cs[i].accepted;
    for (int i = 0; i < in->n_corner_sharp; ++i) mc += cc[i].accepted;
    if (ms < 10) continue;  // SE:1175-1178
    if (mc < 5) continue;   // SE:1181-1184
    if (gauss_newton_step(*prm, t, q, *in, cs, cc, iter)) break;
  }
  std::memcpy(out->state, in->state, sizeof in->state);  // filterState with rn_, qbn_ replaced
  out->state[0] = t[0], out->state[1] = t[1], out->state[2] = t[2];
  out->state[6] = q.w, out->state[7] = q.x, out->state[8] = q.y, out->state[9] = q.z;
  std::memcpy(out->cov, in->cov, sizeof in->cov);  // Pk_ un-updated
  return LINS_OK;
}
