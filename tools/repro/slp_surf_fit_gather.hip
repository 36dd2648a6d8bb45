// Stand-alone attempt at the map_kernels.hip SLP sensitivity (ROCm 7.2 clang, gfx950) WITH the context the fit alone lacks
// (tools/repro/slp_qr5x3.hip equals the host): the neighbours come out of a top-5 insertion network as grid positions,
// are gathered as float4 loads from global memory into px / py / pz, and the fit's result is stored into a record next to
// the indices — as in map_corr_kernel (csrc/map_kernels.hip:100-160; LM:1464-1518).  One thread per query; the device
// result against the same source compiled for the host.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off                     -o fit_slp   tools/repro/slp_surf_fit_gather.hip
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -o fit_noslp tools/repro/slp_surf_fit_gather.hip
// exit code 1 = the device differs from the host.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../lins---lidar-inertial-slam_amd/csrc/map_math.h"

struct Rec {
  float sel[3], coeff[4], sq5;
  int ind[5], accepted;
};
constexpr int kPts = 64;  // points of one wall patch

__host__ __device__ inline Rec fit_query(const float4* gp, float sx, float sy, float sz, int which) {
  unsigned long long key[5];
  int pos[5];
  for (int k = 0; k < 5; ++k) key[k] = ~0ull, pos[k] = 0;
#ifdef REPRO_SIMPLE_POS  // (reduction aid: the five neighbours are simply points 0..4)
  for (int k = 0; k < 5; ++k) key[k] = (unsigned long long)k, pos[k] = k;
  for (int i = kPts; i < kPts; ++i) {
#else
  for (int i = 0; i < kPts; ++i) {
#endif
    const float4 t = gp[i];
    const float dx = t.x - sx, dy = t.y - sy, dz = t.z - sz;
    const float d = (dx * dx + dy * dy) + dz * dz;
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long ck = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
#else
    unsigned u;
    __builtin_memcpy(&u, &d, 4);
    unsigned long long ck = ((unsigned long long)u << 32) | (unsigned)i;
#endif
    int cp = i;
    if (!(ck < key[4])) continue;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const bool before = ck < key[k];
      const unsigned long long tk = key[k];
      const int tp = pos[k];
      key[k] = before ? ck : tk, pos[k] = before ? cp : tp;
      ck = before ? tk : ck, cp = before ? tp : cp;
    }
  }
  Rec r;
  r.sel[0] = sx, r.sel[1] = sy, r.sel[2] = sz;
  r.accepted = 0;
  r.coeff[0] = r.coeff[1] = r.coeff[2] = r.coeff[3] = 0.f;
  unsigned hi = (unsigned)(key[4] >> 32);
  float sq5;
  __builtin_memcpy(&sq5, &hi, 4);
#ifdef REPRO_SIMPLE_POS
  sq5 = 0.5f;
#endif
  if (sq5 < 1.0) {
    float px[5], py[5], pz[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const float4 t = gp[pos[k]];
      px[k] = t.x, py[k] = t.y, pz[k] = t.z;
      r.ind[k] = (int)(unsigned)key[k];
    }
    r.sq5 = sq5;
    float c[4];
#ifdef REPRO_SURF_ONLY  // (reduction aid: the plane fit alone)
    r.accepted = lins::map_surf_fit(px, py, pz, sx, sy, sz, c);
#else
    r.accepted = which == 0 ? lins::map_corner_fit(px, py, pz, sx, sy, sz, c) : lins::map_surf_fit(px, py, pz, sx, sy, sz, c);
#endif
    r.coeff[0] = c[0], r.coeff[1] = c[1], r.coeff[2] = c[2], r.coeff[3] = c[3];
  } else {
    for (int k = 0; k < 5; ++k) r.ind[k] = -1;
    r.sq5 = INFINITY;
  }
  return r;
}
#ifdef REPRO_DUMP  // (reduction aid: the 5 x 3 system after the Householder steps, and the solution, per query)
struct Dump {
  float a[15], b[5], x[3];
};
__host__ __device__ inline Dump qr_dump(const float4* gp) {
  float px[5], py[5], pz[5];
  for (int k = 0; k < 5; ++k) {
    const float4 t = gp[k];
    px[k] = t.x, py[k] = t.y, pz[k] = t.z;
  }
  Dump d;
  float A[15], B[5] = {-1, -1, -1, -1, -1}, X[3];
  for (int j = 0; j < 5; j++) A[j * 3 + 0] = px[j], A[j * 3 + 1] = py[j], A[j * 3 + 2] = pz[j];
  lins::map_qr_5x3(A, B, X);
  for (int i = 0; i < 15; ++i) d.a[i] = A[i];
  for (int i = 0; i < 5; ++i) d.b[i] = B[i];
  for (int i = 0; i < 3; ++i) d.x[i] = X[i];
  return d;
}
__global__ void kern_dump(const float4* pts, Dump* out, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = qr_dump(pts + (size_t)k * kPts);
}
#endif
__global__ void kern(const float4* pts, const float4* q, Rec* out, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = fit_query(pts + (size_t)k * kPts, q[k].x, q[k].y, q[k].z, (int)q[k].w);
}
int main() {
  const int n = 4096;
  std::vector<float4> pts((size_t)n * kPts), q(n);
  unsigned s = 4242u;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f; };
  for (int k = 0; k < n; ++k) {
    const int wall = k % 3;  // 0: x = const, 1: y = const, 2: z = const
    const float c0 = 2.f + 6.f * rnd(), u0 = -3.f + 6.f * rnd(), v0 = -1.f + 2.f * rnd();
    for (int i = 0; i < kPts; ++i) {
      const float u = u0 + 0.25f * (float)(i % 8) + 0.05f * rnd(), v = v0 + 0.25f * (float)(i / 8) + 0.05f * rnd(), w = c0 + 0.01f * rnd();
      pts[(size_t)k * kPts + i] = wall == 0 ? make_float4(w, u, v, 0.f) : (wall == 1 ? make_float4(u, w, v, 0.f) : make_float4(u, v, w, 0.f));
    }
    const float qu = u0 + 0.9f, qv = v0 + 0.9f, qw = c0 + 0.05f;
    q[k] = wall == 0 ? make_float4(qw, qu, qv, 1.f) : (wall == 1 ? make_float4(qu, qw, qv, 1.f) : make_float4(qu, qv, qw, 1.f));
  }
  float4 *dp, *dq;
  Rec* dr;
  hipMalloc(&dp, pts.size() * sizeof(float4)), hipMalloc(&dq, n * sizeof(float4)), hipMalloc(&dr, n * sizeof(Rec));
  hipMemcpy(dp, pts.data(), pts.size() * sizeof(float4), hipMemcpyHostToDevice);
  hipMemcpy(dq, q.data(), n * sizeof(float4), hipMemcpyHostToDevice);
  kern<<<n / 256, 256>>>(dp, dq, dr, n);
  std::vector<Rec> h(n);
  hipMemcpy(h.data(), dr, n * sizeof(Rec), hipMemcpyDeviceToHost);
  int bad = 0, acc = 0;
  for (int k = 0; k < n; ++k) {
    const Rec w = fit_query(pts.data() + (size_t)k * kPts, q[k].x, q[k].y, q[k].z, (int)q[k].w);
    bool same = w.accepted == h[k].accepted;
    for (int i = 0; i < 4; ++i) same = same && w.coeff[i] == h[k].coeff[i];
    for (int i = 0; i < 5; ++i) same = same && w.ind[i] == h[k].ind[i];
    acc += w.accepted;
    if (!same && bad++ < 4)
      printf("query %d (wall %d): device ok=%d (%g %g %g | %g)   host ok=%d (%g %g %g | %g)\n", k, k % 3, h[k].accepted, h[k].coeff[0], h[k].coeff[1],
             h[k].coeff[2], h[k].coeff[3], w.accepted, w.coeff[0], w.coeff[1], w.coeff[2], w.coeff[3]);
  }
  printf("%d of %d fits differ from the host build of the same source (%d accepted on the host)\n", bad, n, acc);
#ifdef REPRO_DUMP
  {
    Dump* dd;
    hipMalloc(&dd, n * sizeof(Dump));
    kern_dump<<<n / 256, 256>>>(dp, dd, n);
    std::vector<Dump> hd(n);
    hipMemcpy(hd.data(), dd, n * sizeof(Dump), hipMemcpyDeviceToHost);
    int cnt[23] = {0};
    for (int k = 0; k < n; ++k) {
      const Dump w = qr_dump(pts.data() + (size_t)k * kPts);
      for (int i = 0; i < 15; ++i) cnt[i] += w.a[i] != hd[k].a[i];
      for (int i = 0; i < 5; ++i) cnt[15 + i] += w.b[i] != hd[k].b[i];
      for (int i = 0; i < 3; ++i) cnt[20 + i] += w.x[i] != hd[k].x[i];
      if (k == 1) {
        printf("query 1 device A:"); for (int i = 0; i < 15; ++i) printf(" %g", hd[k].a[i]); printf("\n");
        printf("query 1 host   A:"); for (int i = 0; i < 15; ++i) printf(" %g", w.a[i]); printf("\n");
        printf("query 1 device B:"); for (int i = 0; i < 5; ++i) printf(" %g", hd[k].b[i]); printf("  X: %g %g %g\n", hd[k].x[0], hd[k].x[1], hd[k].x[2]);
        printf("query 1 host   B:"); for (int i = 0; i < 5; ++i) printf(" %g", w.b[i]); printf("  X: %g %g %g\n", w.x[0], w.x[1], w.x[2]);
      }
    }
    printf("entries that differ (of %d): A:", n); for (int i = 0; i < 15; ++i) printf(" %d", cnt[i]);
    printf(" | B:"); for (int i = 0; i < 5; ++i) printf(" %d", cnt[15 + i]);
    printf(" | X: %d %d %d\n", cnt[20], cnt[21], cnt[22]);
  }
#endif
  return bad != 0;
}
