// debug_kernels.hip — direct device evaluation of the small-math building blocks of the update path, for unit
// tests against the CPU oracle (lins_debug_math; not part of the drop-in surface).  One thread per item.
//   op 0  Quat2axis(q)                      MU:75-88        in 4 (w x y z)   out 3
//   op 1  axis2Quat(v)                      MU:61-73        in 3             out 4
//   op 2  Rinvleft(v)                       MU:304-321      in 3             out 9 (row-major)
//   op 3  GlobalState::boxPlus(x, dx)       KF:71-81        in 19 + 18       out 19
//   op 4  GlobalState::boxMinus(a, b)=a(-)b KF:84-94        in 19 + 19       out 18
//   op 5  the kernels' own per-iteration constants from a linearisation state (ieskf_rowsum.h phi_and_Gt, the
//         sin/cos-free route the persistent kernels take): in 4 (q)          out 3 (phi) + 9 (Rinvleft(-phi)^T)
//   op 6  transformToStart                  SE:1066-1080    in 19 (linState_) + 4 (point x y z intensity) + 1 (scan period)
//                                                           out 3 (the f32 results, widened)
//   op 7 / 8    reg_solve6 in one lane / wave_solve6 over a wave (round 1's elimination + back-substitution)
//   op 11 / 12  gj_solve6 in one lane (lins_solve6.h, the definition) / wave_gj_solve6 over a wave (what the kernels run)
//   op 13 / 14  axis2quat_fast / quat2axis_fast (lins_math.h): the short-series forms of op 1 / op 0 the serial tail uses
//   op 15       phi_and_Gt_general: op 5 with the small-rotation shortcut disabled
//   op 16 / 17  icp_gn_solve in one lane (icp_math.h, the definition) / wave_icp_gn_solve over a wave (icp_wave.h, what the
//               ICP kernel runs): in 36 (J^T J) + 6 (J^T b) + 1 (round), out 6
//   op 18 / 19  lm_step_from_sums by one thread (lm_math.h, the definition) / over a wave (lm_wave.h, what map_lm_kernel
//               runs) — the kernel lives in map_kernels.hip (that file's build flags): in 72, out 44
#include <hip/hip_runtime.h>

#include "ieskf_device.h"
#include "ieskf_rowsum.h"
#include "icp_wave.h"

namespace lins {

__global__ void debug_math_kernel(int op, int n, int n_in, int n_out, const double* __restrict__ in, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* a = in + (size_t)i * n_in;
  double* o = out + (size_t)i * n_out;
  if (op == 0) {
    const V3 v = quat2axis(Q4{a[0], a[1], a[2], a[3]});
    o[0] = v.x, o[1] = v.y, o[2] = v.z;
  } else if (op == 1) {
    const Q4 q = axis2quat(V3{a[0], a[1], a[2]});
    o[0] = q.w, o[1] = q.x, o[2] = q.y, o[3] = q.z;
  } else if (op == 2) {
    const M3 m = rinvleft(V3{a[0], a[1], a[2]});
    for (int k = 0; k < 9; ++k) o[k] = m.m[k];
  } else if (op == 3) {
    double s[19];
    for (int k = 0; k < 19; ++k) s[k] = a[k];
    box_plus_inplace(s, a + 19);
    for (int k = 0; k < 19; ++k) o[k] = s[k];
  } else if (op == 4) {
    IterConst ic;
    for (int k = 0; k < 19; ++k) ic.lin[k] = a[19 + k];  // d = filter (-) lin
    make_iter_const(a, ic);
    for (int k = 0; k < 18; ++k) o[k] = ic.d[k];
  } else if (op == 5) {
    V3 phi;
    M3 gt;
    phi_and_Gt(Q4{a[0], a[1], a[2], a[3]}, phi, gt);
    o[0] = phi.x, o[1] = phi.y, o[2] = phi.z;
    for (int k = 0; k < 9; ++k) o[3 + k] = gt.m[k];
  } else if (op == 7) {  // one lane's reg_solve6 and (below, whole wave) wave_solve6 on the same 6 x 7 system
    double m[6][7], x6[6];
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 7; ++c) m[r][c] = a[r * 7 + c];
    reg_solve6(m, x6);
    for (int k = 0; k < 6; ++k) o[k] = x6[k];
  } else if (op == 11) {
    double m[6][7], x6[6];
    for (int r = 0; r < 6; ++r)
      for (int c = 0; c < 7; ++c) m[r][c] = a[r * 7 + c];
    gj_solve6(m, x6);
    for (int k = 0; k < 6; ++k) o[k] = x6[k];
  } else if (op == 13) {
    const Q4 q = axis2quat_fast(V3{a[0], a[1], a[2]});
    o[0] = q.w, o[1] = q.x, o[2] = q.y, o[3] = q.z;
  } else if (op == 14) {
    const V3 v = quat2axis_fast(Q4{a[0], a[1], a[2], a[3]});
    o[0] = v.x, o[1] = v.y, o[2] = v.z;
  } else if (op == 15) {
    V3 phi;
    M3 gt;
    phi_and_Gt_general(Q4{a[0], a[1], a[2], a[3]}, phi, gt);
    o[0] = phi.x, o[1] = phi.y, o[2] = phi.z;
    for (int k = 0; k < 9; ++k) o[3 + k] = gt.m[k];
  } else if (op == 6) {
    DevParams prm{};
    prm.inv_period = (double)(1.f / (float)a[23]);
    const V3 phi = quat2axis(Q4{a[6], a[7], a[8], a[9]});
    float x, y, z;
    transform_to_start(prm, phi, V3{a[0], a[1], a[2]}, make_float4((float)a[19], (float)a[20], (float)a[21], (float)a[22]), x, y, z);
    o[0] = x, o[1] = y, o[2] = z;
  }
}

// op 100..: cycle counts of the building blocks (microbenchmark): out[0] = shader cycles per call, one 256-thread
// workgroup per launch block, `n` blocks (to load the CU with several), 64 dependent repetitions
__global__ __launch_bounds__(256) void debug_cycles_kernel(int op, const double* __restrict__ in, double* __restrict__ out) {
  const int tid = threadIdx.x;
  DevParams prm{};
  prm.inv_period = 10.0, prm.icp_freq = 1, prm.lidar_scale = 1.0;
  V3 phi{in[0], in[1], in[2]}, t{in[3], in[4], in[5]};
  float4 q = make_float4((float)in[6] + tid * 0.01f, (float)in[7], (float)in[8], 3.05f);
  float acc = 0.f;
  double dacc = 0.0;
  const long long t0 = clock64();
#pragma unroll 1
  for (int r = 0; r < 64; ++r) {
    if (op == 100) {  // transformToStart
      float x, y, z;
      transform_to_start(prm, phi, t, q, x, y, z);
      q.x = x * 0.999f + acc * 1e-9f, acc += y + z;
    } else if (op == 101) {  // plane row
      QueryOut o;
      o.accepted = 0;
      o.c[0] = o.c[1] = o.c[2] = o.c[3] = 0.f;
      surf_row(prm, 1, q.x, q.y, q.z, make_float4(1.f + acc, 2.f, 3.f, 0.f), make_float4(1.5f, 2.2f, 3.f, 0.f), make_float4(1.f, 2.6f, 3.1f, 0.f), o);
      acc += o.c[0] * 1e-3f + o.c[3];
    } else if (op == 102) {  // 28 sums of a wave
      double row[7] = {dacc + 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0 + tid};
      dacc += wave_reduce_rows(row, tid & 63) * 1e-9;
    } else if (op == 103) {  // 6x6 solve in one lane
      if ((tid & 63) == 0) {
        double a[6][7], x6[6];
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 7; ++j) a[i][j] = (i == j ? 10.0 : 0.1 * (i + j)) + dacc;
        reg_solve6(a, x6);
        dacc += x6[0] * 1e-9;
      }
    } else if (op == 105) {  // 6x6 solve spread over the wave
      const int l = tid & 63, i = l / 7, j = l % 7;
      double x6[6];
      wave_solve6(l < 42 ? (i == j ? 10.0 : 0.1 * (i + j)) + dacc : 0.0, l, x6);
      dacc += x6[0] * 1e-9;
    } else if (op == 106) {  // the Gauss-Jordan solve spread over the wave (what the kernels run)
      const int l = tid & 63, i = l / 7, j = l % 7;
      double x6[6];
      wave_gj_solve6(l < 42 ? (i == j ? 10.0 : 0.1 * (i + j)) + dacc : 0.0, l, x6);
      dacc += x6[0] * 1e-9;
    } else if (op == 107) {  // next iteration's constants by the general (atan2) route
      V3 p2;
      M3 gt;
      phi_and_Gt_general(Q4{1.0 - dacc, 0.01, 0.02, 0.03 + dacc}, p2, gt);
      dacc += (p2.x + gt.m[1]) * 1e-9;
    } else if (op == 108) {  // boxPlus' quaternion step, libm route / short series
      const Q4 r = axis2quat(V3{0.001 + dacc, 0.002, 0.003});
      dacc += (r.x + r.w) * 1e-9;
    } else if (op == 109) {
      const Q4 r = axis2quat_fast(V3{0.001 + dacc, 0.002, 0.003});
      dacc += (r.x + r.w) * 1e-9;
    } else if (op == 110) {  // boxMinus' rotation part, libm route / short series
      const V3 r = quat2axis(Q4{1.0 - dacc, 0.001, 0.002, 0.003});
      dacc += r.x * 1e-9;
    } else if (op == 111) {
      const V3 r = quat2axis_fast(Q4{1.0 - dacc, 0.001, 0.002, 0.003});
      dacc += r.x * 1e-9;
    } else if (op == 104) {  // next iteration's constants
      V3 p2;
      M3 gt;
      phi_and_Gt(Q4{1.0 - dacc, 0.01, 0.02, 0.03 + dacc}, p2, gt);
      dacc += (p2.x + gt.m[1]) * 1e-9;
    }
  }
  const long long t1 = clock64();
  if (tid == 0) out[blockIdx.x * 2] = (double)(t1 - t0) / 64.0, out[blockIdx.x * 2 + 1] = (double)acc + dacc;
}

// op 8 / 12: wave_solve6 / wave_gj_solve6, one system per WAVE (64 threads per item): out 6
__global__ void debug_wave_solve_kernel(int n, int gj, const double* __restrict__ in, double* __restrict__ out) {
  const int item = blockIdx.x, l = threadIdx.x;
  double x6[6];
  if (gj)
    wave_gj_solve6(l < 42 ? in[(size_t)item * 42 + l] : 0.0, l, x6);
  else
    wave_solve6(l < 42 ? in[(size_t)item * 42 + l] : 0.0, l, x6);
  if (l < 6) out[(size_t)item * 6 + l] = x6[l];
}
// op 9: wave_reduce_rows (one wave per item, 64 rows of 7 in, the 28 sums out); op 10: the same tree with plain
// __shfl_xor exchanges — the two must agree bit for bit (the cross-lane paths of xor_lane_i32 are then right)
__device__ __forceinline__ double reduce_rows_ref(const double (&row)[7], int lane) {
  constexpr int A[28] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 0, 1, 2, 3, 4, 5, 6};
  constexpr int B[28] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5, 6, 6, 6, 6, 6, 6, 6};
  double v[28];
  for (int k = 0; k < 28; ++k) v[k] = row[A[k]] * row[B[k]];
  int cnt = 14;
  for (int m = 32; m >= 2; m >>= 1) {
    const bool up = (lane & m) != 0;
    if (m == 8) v[7] = 0.0;
    for (int i = 0; i < cnt; ++i) {
      const double lo = v[i], hi = v[i + cnt];
      v[i] = (up ? hi : lo) + shfl_xor_f64(up ? lo : hi, m);
    }
    cnt = m == 16 ? 4 : cnt / 2;
  }
  return v[0] + shfl_xor_f64(v[0], 1);
}
__global__ void debug_reduce_rows_kernel(int op, const double* __restrict__ in, double* __restrict__ out) {
  const int item = blockIdx.x, l = threadIdx.x;
  double row[7];
  for (int k = 0; k < 7; ++k) row[k] = in[((size_t)item * 64 + l) * 7 + k];
  const double acc = op == 9 ? wave_reduce_rows(row, l) : reduce_rows_ref(row, l);
  const int idx = reduce_sum_index(l);
  if (idx >= 0) out[(size_t)item * 28 + idx] = acc;
}
void launch_debug_reduce_rows(hipStream_t stream, int op, int n, const double* in, double* out) {
  hipLaunchKernelGGL(debug_reduce_rows_kernel, dim3(n), dim3(64), 0, stream, op, in, out);
}

// op 16 / 17: icp_gn_solve (icp_math.h, one lane, workspace in LDS — the definition) / wave_icp_gn_solve (icp_wave.h,
// what the ICP kernel runs) on the same system: in 36 (J^T J row-major) + 6 (J^T b) + 1 (round index), out 6
__global__ void debug_icp_gn_kernel(int wave_version, const double* __restrict__ in, double* __restrict__ out) {
  __shared__ double ws[kIcpWorkspace + 48];
  const int item = blockIdx.x, l = threadIdx.x;
  const double* a = in + (size_t)item * 43;
  double* JTJ = ws + kIcpWorkspace;
  if (l < 42) JTJ[l] = a[l];
  __syncthreads();
  const int iter = (int)a[42];
  if (wave_version) {
    double x[6];
    wave_icp_gn_solve(JTJ, JTJ + 36, iter, l, x, ws);
    if (l == 0)
      for (int k = 0; k < 6; ++k) out[(size_t)item * 6 + k] = x[k];
  } else if (l == 0) {
    double* x = JTJ + 42;
    icp_gn_solve(JTJ, JTJ + 36, iter, x, ws);
    for (int k = 0; k < 6; ++k) out[(size_t)item * 6 + k] = x[k];
  }
}
void launch_debug_icp_gn(hipStream_t stream, int n, int wave_version, const double* in, double* out) {
  hipLaunchKernelGGL(debug_icp_gn_kernel, dim3(n), dim3(64), 0, stream, wave_version, in, out);
}

void launch_debug_wave_solve(hipStream_t stream, int n, int gj, const double* in, double* out) {
  hipLaunchKernelGGL(debug_wave_solve_kernel, dim3(n), dim3(64), 0, stream, n, gj, in, out);
}

void launch_debug_cycles(hipStream_t stream, int op, int blocks, const double* in, double* out) {
  hipLaunchKernelGGL(debug_cycles_kernel, dim3(blocks), dim3(256), 0, stream, op, in, out);
}

void launch_debug_math(hipStream_t stream, int op, int n, int n_in, int n_out, const double* in, double* out) {
  hipLaunchKernelGGL(debug_math_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, op, n, n_in, n_out, in, out);
}

}  // namespace lins
