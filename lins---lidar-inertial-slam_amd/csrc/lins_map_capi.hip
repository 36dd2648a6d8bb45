// lins_map_capi.hip — C ABI of the scan-to-map row (include/lins_map.h): host orchestration around
// map_kernels.hip.  Per problem the two map clouds are bucketed into 1 m cells once (host counting sort,
// threaded over problems); each round the host forms the trigonometry of the current transform (libm, as the
// reference does, LM:579-592 / 1524-1529), the device evaluates every query (5-NN, fits, rows, 28 sums),
// and the host takes the 6-DoF Gauss-Newton step of LMOptimization in f32 (QR solve, degeneracy projection
// of round 0, stop rule) — 6x6 scalar algebra per problem.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/lins_map.h"
#include "lins_ctx_priv.h"
#include "lm_math.h"
#include "map_math.h"

namespace lins {
void launch_map_selfcheck(hipStream_t, float*);
void launch_map_corr(hipStream_t, int, int, const void*, const void*, const float4*, const int*, const float4*,
                     lins_map_corr*, double*);
void launch_map_grid(hipStream_t, int, const void*, const float4*, float4*, int*);
void launch_map_lm(hipStream_t, int, int, int, void*, void*, const double*, lins_map_result*, void*);
size_t map_dev_size();
size_t map_round_size();
size_t map_grid_job_size();
size_t map_carry_size();
int map_block();
}  // namespace lins
using namespace lins;

namespace {

struct MapGridHost {
  long long off_pts, off_cells;
  int cmin[3], cdim[3];
};
struct MapDevHost {
  MapGridHost g[2];
  long long off_q, off_rec;
  int n_q[2];
  int active, pad;
};
struct MapGridJobHost {  // (map_kernels.hip's MapGridJob)
  long long off_raw, off_pts, off_cells;
  int n, ncell;
  int cmin[3], cdim[3];
};

struct MapState {
  float4 *d_raw = nullptr, *d_pts = nullptr, *d_q = nullptr;
  float4* h_q = nullptr;  // pinned staging of the queries of one call (one copy instead of two per problem)
  size_t cap_hq = 0;
  int* d_cells = nullptr;
  lins_map_corr* d_rec = nullptr;
  double* d_partials = nullptr;
  void *d_probs = nullptr, *d_rounds = nullptr, *d_jobs = nullptr, *d_carry = nullptr;
  lins_map_result* d_results = nullptr;
  size_t cap_raw = 0, cap_pts = 0, cap_q = 0, cap_cells = 0, cap_partials = 0;
  int cap_probs = 0;
  // the maps resident on the device (gridded): sizes per problem of the last upload, for LINS_MAP_REUSE
  std::vector<int> resident_sizes;
  std::vector<MapDevHost> resident_dev;
  float ms = 0.f;
  uint64_t queries = 0;
  bool selfcheck_done = false;
};

void map_state_free(void* p) {
  MapState* m = (MapState*)p;
  (void)hipFree(m->d_raw), (void)hipFree(m->d_pts), (void)hipFree(m->d_q), (void)hipFree(m->d_cells), (void)hipFree(m->d_rec);
  (void)hipFree(m->d_partials), (void)hipFree(m->d_probs), (void)hipFree(m->d_rounds), (void)hipFree(m->d_jobs);
  (void)hipFree(m->d_carry), (void)hipFree(m->d_results), (void)hipHostFree(m->h_q);
  delete m;
}

#define MAP_TRY(ctx, expr)                                  \
  do {                                                      \
    hipError_t e__ = (expr);                                \
    if (e__ != hipSuccess) return ctx_fail_hip(ctx, e__, #expr); \
  } while (0)

template <class T>
int grow(lins_ctx* ctx, T** p, size_t* cap, size_t need) {
  if (*cap >= need) return LINS_OK;
  (void)hipFree(*p);
  *p = nullptr, *cap = 0;
  MAP_TRY(ctx, hipMalloc((void**)p, need * sizeof(T)));
  *cap = need;
  return LINS_OK;
}

template <class F>
int parallel_for(int n, F fn) {
  const unsigned hw = std::thread::hardware_concurrency();
  const int T = std::max(1, std::min({16, (int)(hw ? hw : 1), n}));
  std::vector<int> rc(n, 0);
  std::atomic<int> next{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < T; ++t)
    pool.emplace_back([&] {
      for (int k; (k = next.fetch_add(1)) < n;) rc[k] = fn(k);
    });
  for (auto& th : pool) th.join();
  for (int k = 0; k < n; ++k)
    if (rc[k]) return rc[k];
  return 0;
}

// input contract of a map cloud + its bounding box in 1 m cells (the counting sort itself runs on the device)
int cloud_box(const lins_point* p, int n, int* cmin, int* cdim, long long* ncell) {
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    if (!std::isfinite(p[i].x) || !std::isfinite(p[i].y) || !std::isfinite(p[i].z)) return LINS_E_INPUT;
    if (std::fabs(p[i].x) > 1e6f || std::fabs(p[i].y) > 1e6f || std::fabs(p[i].z) > 1e6f) return LINS_E_INPUT;
    const int c[3] = {(int)std::floor(p[i].x), (int)std::floor(p[i].y), (int)std::floor(p[i].z)};
    for (int a = 0; a < 3; ++a) lo[a] = i ? std::min(lo[a], c[a]) : c[a], hi[a] = i ? std::max(hi[a], c[a]) : c[a];
  }
  *ncell = 1;
  for (int a = 0; a < 3; ++a) cmin[a] = lo[a], cdim[a] = hi[a] - lo[a] + 1, *ncell *= cdim[a];
  if (*ncell > (1ll << 26)) return LINS_E_CAPACITY;  // a map far larger than a local one (LM keeps ~50 key frames)
  return LINS_OK;
}

// Brings n problems onto the device: queries always; the maps (raw upload + device gridding) unless every problem
// carries LINS_MAP_REUSE and the resident maps have the same sizes — the local map of the mapping node only changes
// with its key frames.  Fills the device descriptors; active[k] = precondition of LM:1636.
int map_upload(lins_ctx* ctx, MapState* m, int n, const lins_map_problem* in, std::vector<MapDevHost>& dev, int* max_q) {
  bool reuse = (int)m->resident_dev.size() == n && n > 0;
  for (int k = 0; k < n; ++k) {
    const lins_map_problem& p = in[k];
    if (p.n_map_corner < 0 || p.n_map_surf < 0 || p.n_scan_corner < 0 || p.n_scan_surf < 0) return LINS_E_ARG;
    if ((p.n_map_corner && !p.map_corner) || (p.n_map_surf && !p.map_surf) || (p.n_scan_corner && !p.scan_corner) ||
        (p.n_scan_surf && !p.scan_surf))
      return LINS_E_ARG;
    if (!(p.reserved[0] & LINS_MAP_REUSE) || !reuse || m->resident_sizes[2 * k] != p.n_map_corner ||
        m->resident_sizes[2 * k + 1] != p.n_map_surf)
      reuse = false;
  }
  hipStream_t st = ctx_stream(ctx);
  int rc;
  if (!reuse) {
    m->resident_dev.clear(), m->resident_sizes.clear();
    std::vector<MapGridJobHost> jobs((size_t)n * 2);
    rc = parallel_for(2 * n, [&](int j) {
      const lins_map_problem& p = in[j / 2];
      MapGridJobHost& jb = jobs[j];
      jb.n = (j & 1) ? p.n_map_surf : p.n_map_corner;
      long long ncell = 1;
      const int r = cloud_box((j & 1) ? p.map_surf : p.map_corner, jb.n, jb.cmin, jb.cdim, &ncell);
      jb.ncell = (int)ncell;
      return r;
    });
    if (rc) return rc;
    size_t tot_pts = 0, tot_cells = 0;
    dev.assign(n, MapDevHost{});
    for (int k = 0; k < n; ++k)
      for (int w = 0; w < 2; ++w) {
        MapGridJobHost& jb = jobs[(size_t)k * 2 + w];
        jb.off_raw = jb.off_pts = (long long)tot_pts, jb.off_cells = (long long)tot_cells;
        dev[k].g[w].off_pts = jb.off_pts, dev[k].g[w].off_cells = jb.off_cells;
        for (int a = 0; a < 3; ++a) dev[k].g[w].cmin[a] = jb.cmin[a], dev[k].g[w].cdim[a] = jb.cdim[a];
        tot_pts += (size_t)jb.n, tot_cells += 2 * ((size_t)jb.ncell + 1);  // starts + scratch cursors
      }
    if ((rc = grow(ctx, &m->d_raw, &m->cap_raw, std::max<size_t>(tot_pts, 1)))) return rc;
    if ((rc = grow(ctx, &m->d_pts, &m->cap_pts, std::max<size_t>(tot_pts, 1)))) return rc;
    if ((rc = grow(ctx, &m->d_cells, &m->cap_cells, std::max<size_t>(tot_cells, 1)))) return rc;
    if (m->cap_probs < n) {
      (void)hipFree(m->d_probs), (void)hipFree(m->d_rounds), (void)hipFree(m->d_jobs), (void)hipFree(m->d_carry), (void)hipFree(m->d_results);
      m->d_probs = m->d_rounds = m->d_jobs = m->d_carry = nullptr, m->d_results = nullptr, m->cap_probs = 0;
      MAP_TRY(ctx, hipMalloc(&m->d_probs, (size_t)n * sizeof(MapDevHost)));
      MAP_TRY(ctx, hipMalloc(&m->d_rounds, (size_t)n * map_round_size()));
      MAP_TRY(ctx, hipMalloc(&m->d_jobs, (size_t)n * 2 * sizeof(MapGridJobHost)));
      MAP_TRY(ctx, hipMalloc(&m->d_carry, (size_t)n * map_carry_size()));
      MAP_TRY(ctx, hipMalloc((void**)&m->d_results, (size_t)n * sizeof(lins_map_result)));
      m->cap_probs = n;
    }
    for (int k = 0; k < n; ++k) {
      if (in[k].n_map_corner)
        MAP_TRY(ctx, hipMemcpyAsync(m->d_raw + jobs[(size_t)k * 2].off_raw, in[k].map_corner, (size_t)in[k].n_map_corner * sizeof(float4), hipMemcpyHostToDevice, st));
      if (in[k].n_map_surf)
        MAP_TRY(ctx, hipMemcpyAsync(m->d_raw + jobs[(size_t)k * 2 + 1].off_raw, in[k].map_surf, (size_t)in[k].n_map_surf * sizeof(float4), hipMemcpyHostToDevice, st));
    }
    MAP_TRY(ctx, hipMemcpyAsync(m->d_jobs, jobs.data(), jobs.size() * sizeof(MapGridJobHost), hipMemcpyHostToDevice, st));
    launch_map_grid(st, 2 * n, m->d_jobs, m->d_raw, m->d_pts, m->d_cells);
    MAP_TRY(ctx, hipGetLastError());
    MAP_TRY(ctx, hipStreamSynchronize(st));  // (jobs goes out of scope)
    for (int k = 0; k < n; ++k) m->resident_sizes.push_back(in[k].n_map_corner), m->resident_sizes.push_back(in[k].n_map_surf);
  } else {
    dev = m->resident_dev;
  }
  size_t tot_q = 0;
  *max_q = 0;
  for (int k = 0; k < n; ++k) {
    dev[k].off_q = dev[k].off_rec = (long long)tot_q;
    dev[k].n_q[0] = in[k].n_scan_corner, dev[k].n_q[1] = in[k].n_scan_surf;
    dev[k].active = in[k].n_map_corner > 10 && in[k].n_map_surf > 100;  // LM:1636
    tot_q += (size_t)dev[k].n_q[0] + dev[k].n_q[1];
    *max_q = std::max(*max_q, dev[k].n_q[0] + dev[k].n_q[1]);
  }
  if (m->cap_hq < std::max<size_t>(tot_q, 1)) {
    (void)hipHostFree(m->h_q);
    m->h_q = nullptr, m->cap_hq = 0;
    MAP_TRY(ctx, hipHostMalloc((void**)&m->h_q, std::max<size_t>(tot_q, 1) * sizeof(float4)));
    m->cap_hq = std::max<size_t>(tot_q, 1);
  }
  rc = parallel_for(n, [&](int k) {  // input contract + packing into the pinned staging buffer
    float4* dst = m->h_q + dev[k].off_q;
    for (int i = 0; i < in[k].n_scan_corner + in[k].n_scan_surf; ++i) {
      const lins_point& q = i < in[k].n_scan_corner ? in[k].scan_corner[i] : in[k].scan_surf[i - in[k].n_scan_corner];
      if (!std::isfinite(q.x) || !std::isfinite(q.y) || !std::isfinite(q.z)) return (int)LINS_E_INPUT;
      dst[i] = make_float4(q.x, q.y, q.z, q.intensity);
    }
    return (int)LINS_OK;
  });
  if (rc) return rc;
  if (m->cap_q < std::max<size_t>(tot_q, 1)) {
    (void)hipFree(m->d_q), (void)hipFree(m->d_rec);
    m->d_q = nullptr, m->d_rec = nullptr, m->cap_q = 0;
    MAP_TRY(ctx, hipMalloc((void**)&m->d_q, std::max<size_t>(tot_q, 1) * sizeof(float4)));
    MAP_TRY(ctx, hipMalloc((void**)&m->d_rec, std::max<size_t>(tot_q, 1) * sizeof(lins_map_corr)));
    m->cap_q = std::max<size_t>(tot_q, 1);
  }
  if (tot_q) MAP_TRY(ctx, hipMemcpyAsync(m->d_q, m->h_q, tot_q * sizeof(float4), hipMemcpyHostToDevice, st));
  m->resident_dev = dev;
  return LINS_OK;
}

MapState* state_of(lins_ctx* ctx) {
  void** slot = ctx_map_slot(ctx, map_state_free);
  if (!*slot) *slot = new MapState();
  return (MapState*)*slot;
}

// once per context: the device plane fit on a known wall (see map_selfcheck_kernel)
int map_selfcheck(lins_ctx* ctx, MapState* m) {
  if (m->selfcheck_done) return LINS_OK;
  float* d = nullptr;
  float h[5] = {0, 0, 0, 0, 0};
  MAP_TRY(ctx, hipMalloc((void**)&d, sizeof h));
  launch_map_selfcheck(ctx_stream(ctx), d);
  hipError_t e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx_stream(ctx));
  if (e == hipSuccess) e = hipStreamSynchronize(ctx_stream(ctx));
  (void)hipFree(d);
  if (e != hipSuccess) return ctx_fail_hip(ctx, e, "map self-check");
  // normal (0, -1, 0) scaled by the weight s ~ 0.97, signed distance -0.05 scaled likewise, accepted
  const bool ok = h[4] == 1.f && std::fabs(h[0]) < 1e-3f && std::fabs(h[2]) < 1e-3f && h[1] < -0.9f && std::fabs(h[3] + 0.0485f) < 2e-3f;
  if (!ok) return ctx_fail_hip(ctx, hipErrorUnknown, "scan-to-map plane fit self-check failed: the 5x3 QR is miscompiled (toolchain / flags changed?)");
  m->selfcheck_done = true;
  return LINS_OK;
}

}  // namespace

extern "C" {

int lins_scan2map_batch(lins_ctx* ctx, int n, const lins_map_problem* in, lins_map_result* out) {
  if (!ctx || n < 0 || (n && (!in || !out))) return LINS_E_ARG;
  if (n == 0) return LINS_OK;
  static_assert(sizeof(MapDevHost) == 112 && sizeof(MapRoundParams) == 64 && sizeof(lins_map_corr) == 56, "layouts");
  if (map_dev_size() != sizeof(MapDevHost) || map_round_size() != sizeof(MapRoundParams) ||
      map_grid_job_size() != sizeof(MapGridJobHost))
    return LINS_E_STATE;
  MAP_TRY(ctx, hipSetDevice(ctx_device(ctx)));
  MapState* m = state_of(ctx);
  int rc = map_selfcheck(ctx, m);
  if (rc) return rc;
  std::vector<MapDevHost> dev;
  int max_q = 0;
  rc = map_upload(ctx, m, n, in, dev, &max_q);
  if (rc) {
    m->resident_dev.clear(), m->resident_sizes.clear();
    return rc;
  }
  const int bpp = std::max(1, (max_q + map_block() - 1) / map_block());
  if ((rc = grow(ctx, &m->d_partials, &m->cap_partials, (size_t)n * bpp * 28))) return rc;
  hipStream_t st = ctx_stream(ctx);
  hipEvent_t e0, e1;
  ctx_events(ctx, &e0, &e1);
  for (int k = 0; k < n; ++k) {
    std::memcpy(out[k].transform, in[k].transform, sizeof out[k].transform);
    out[k].iters = 0, out[k].converged = 0, out[k].degenerate = 0, out[k].n_sel = 0;
  }
  // the ten rounds of scan2MapOptimization (LM:1640-1647) back to back on the device: correspondences + rows + sums,
  // then the 6x6 step, which also writes the next round's rotation terms and retires converged problems
  MAP_TRY(ctx, hipMemcpyAsync(m->d_probs, dev.data(), (size_t)n * sizeof(MapDevHost), hipMemcpyHostToDevice, st));
  MAP_TRY(ctx, hipMemcpyAsync(m->d_results, out, (size_t)n * sizeof(lins_map_result), hipMemcpyHostToDevice, st));
  MAP_TRY(ctx, hipEventRecord(e0, st));
  launch_map_lm(st, n, -1, bpp, m->d_probs, m->d_rounds, m->d_partials, m->d_results, m->d_carry);
  for (int iter = 0; iter < 10; ++iter) {
    launch_map_corr(st, n, bpp, m->d_probs, m->d_rounds, m->d_pts, m->d_cells, m->d_q, m->d_rec, m->d_partials);
    launch_map_lm(st, n, iter, bpp, m->d_probs, m->d_rounds, m->d_partials, m->d_results, m->d_carry);
  }
  MAP_TRY(ctx, hipGetLastError());
  MAP_TRY(ctx, hipEventRecord(e1, st));
  MAP_TRY(ctx, hipMemcpyAsync(out, m->d_results, (size_t)n * sizeof(lins_map_result), hipMemcpyDeviceToHost, st));
  MAP_TRY(ctx, hipStreamSynchronize(st));
  MAP_TRY(ctx, hipEventElapsedTime(&m->ms, e0, e1));
  m->queries = 0;
  for (int k = 0; k < n; ++k) m->queries += (uint64_t)out[k].iters * ((uint64_t)dev[k].n_q[0] + dev[k].n_q[1]);
  return LINS_OK;
}

int lins_map_correspondences(lins_ctx* ctx, const lins_map_problem* in, lins_map_corr* corner, lins_map_corr* surf) {
  if (!ctx || !in || (in->n_scan_corner && !corner) || (in->n_scan_surf && !surf)) return LINS_E_ARG;
  if (map_dev_size() != sizeof(MapDevHost) || map_round_size() != sizeof(MapRoundParams)) return LINS_E_STATE;
  MAP_TRY(ctx, hipSetDevice(ctx_device(ctx)));
  MapState* m = state_of(ctx);
  int rc = map_selfcheck(ctx, m);
  if (rc) return rc;
  std::vector<MapDevHost> dev;
  int max_q = 0;
  rc = map_upload(ctx, m, 1, in, dev, &max_q);
  if (rc) {
    m->resident_dev.clear(), m->resident_sizes.clear();
    return rc;
  }
  dev[0].active = 1;  // a single pass is evaluated whatever the map sizes
  const int bpp = std::max(1, (max_q + map_block() - 1) / map_block());
  if ((rc = grow(ctx, &m->d_partials, &m->cap_partials, (size_t)bpp * 28))) return rc;
  hipStream_t st = ctx_stream(ctx);
  const MapRoundParams rd = lm_make_round(in->transform);
  MAP_TRY(ctx, hipMemcpyAsync(m->d_probs, dev.data(), sizeof(MapDevHost), hipMemcpyHostToDevice, st));
  MAP_TRY(ctx, hipMemcpyAsync(m->d_rounds, &rd, sizeof rd, hipMemcpyHostToDevice, st));
  launch_map_corr(st, 1, bpp, m->d_probs, m->d_rounds, m->d_pts, m->d_cells, m->d_q, m->d_rec, m->d_partials);
  MAP_TRY(ctx, hipGetLastError());
  if (in->n_scan_corner)
    MAP_TRY(ctx, hipMemcpyAsync(corner, m->d_rec, (size_t)in->n_scan_corner * sizeof(lins_map_corr), hipMemcpyDeviceToHost, st));
  if (in->n_scan_surf)
    MAP_TRY(ctx, hipMemcpyAsync(surf, m->d_rec + in->n_scan_corner, (size_t)in->n_scan_surf * sizeof(lins_map_corr), hipMemcpyDeviceToHost, st));
  MAP_TRY(ctx, hipStreamSynchronize(st));
  return LINS_OK;
}

int lins_last_map_stats(lins_ctx* ctx, float* kernel_ms, uint64_t* queries) {
  if (!ctx) return LINS_E_ARG;
  MapState* m = state_of(ctx);
  if (kernel_ms) *kernel_ms = m->ms;
  if (queries) *queries = m->queries;
  return LINS_OK;
}

}  // extern "C"
