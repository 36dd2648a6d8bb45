// Does a HIP graph shorten a launch-latency-bound chain like lins_ieskf_update of ONE scan?  The chain: two pinned H2D copies
// (144 KB clouds, 3 KB priors), a short kernel (the search index), a ~135 us kernel (the update), one D2H copy (3 KB), a wait.
// Prints the median wall time of the chain enqueued call by call and launched as an instantiated graph.
// build + run: hipcc -O2 --offload-arch=gfx950 tools/experiments/graph_latency.hip -o /tmp/graph_latency && /tmp/graph_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void short_kernel(const float4* in, float4* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
__global__ void spin_kernel(const double* in, double* out, long long ticks) {  // one workgroup busy for `ticks` of the 100 MHz clock
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x < 343) out[threadIdx.x] = in[threadIdx.x] + 1.0;
}
int main() {
  const size_t nb_cloud = 144 << 10, nb_meta = 3 << 10;
  char *h_cloud, *h_meta, *h_back, *d_cloud, *d_sorted, *d_meta, *d_out;
  CK(hipHostMalloc((void**)&h_cloud, nb_cloud)); CK(hipHostMalloc((void**)&h_meta, nb_meta)); CK(hipHostMalloc((void**)&h_back, nb_meta));
  CK(hipMalloc((void**)&d_cloud, nb_cloud)); CK(hipMalloc((void**)&d_sorted, nb_cloud)); CK(hipMalloc((void**)&d_meta, nb_meta)); CK(hipMalloc((void**)&d_out, nb_meta));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n4 = (int)(nb_cloud / 16);
  auto chain = [&](bool events) -> int {
    CK(hipMemcpyAsync(d_cloud, h_cloud, nb_cloud, hipMemcpyHostToDevice, st));
    CK(hipMemcpyAsync(d_meta, h_meta, nb_meta, hipMemcpyHostToDevice, st));
    short_kernel<<<(n4 + 255) / 256, 256, 0, st>>>((const float4*)d_cloud, (float4*)d_sorted, n4);
    if (events) CK(hipEventRecord(e0, st));
    spin_kernel<<<1, 1024, 0, st>>>((const double*)d_meta, (double*)d_out, 13500);
    if (events) CK(hipEventRecord(e1, st));
    CK(hipMemcpyAsync(h_back, d_out, nb_meta, hipMemcpyDeviceToHost, st));
    return 0;
  };
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  for (int ev = 0; ev < 2; ++ev) {
    std::vector<double> a, b;
    for (int r = 0; r < 60; ++r) {
      auto t0 = std::chrono::steady_clock::now();
      if (chain(ev)) return 1;
      CK(hipStreamSynchronize(st));
      a.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    if (chain(ev)) return 1;
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 60; ++r) {
      auto t0 = std::chrono::steady_clock::now();
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      b.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    float ms = 0; if (ev) CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%s timing events: call by call %.1f us (min %.1f), graph %.1f us (min %.1f); the long kernel by its events %.1f us\n", ev ? "with" : "without", med(a), a[0], med(b), b[0], ms * 1e3);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
