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
#include "map_math.h"

namespace lins {
void launch_map_corr(hipStream_t, int, int, const void*, const void*, const float4*, const int*, const float4*,
                     lins_map_corr*, double*);
size_t map_dev_size();
size_t map_round_size();
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
struct MapRoundHost {
  MapAssoc as;
  MapTrig tg;
  float pad;
};

struct MapState {
  float4 *d_pts = nullptr, *d_q = nullptr;
  int* d_cells = nullptr;
  lins_map_corr* d_rec = nullptr;
  double* d_partials = nullptr;
  void *d_probs = nullptr, *d_rounds = nullptr;
  size_t cap_pts = 0, cap_q = 0, cap_cells = 0, cap_partials = 0;
  int cap_probs = 0;
  float ms = 0.f;
  uint64_t queries = 0;
};

void map_state_free(void* p) {
  MapState* m = (MapState*)p;
  (void)hipFree(m->d_pts), (void)hipFree(m->d_q), (void)hipFree(m->d_cells), (void)hipFree(m->d_rec);
  (void)hipFree(m->d_partials), (void)hipFree(m->d_probs), (void)hipFree(m->d_rounds);
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

// ---- LMOptimization's 6x6 algebra in f32 (LM:1583-1632), the product's own copy of the fixed sequences -------
void lm_eig6(float* a, float* w, float* V) {  // cyclic Jacobi: w descending, rows of V = eigenvectors
  const int N = 6;
  float v[36];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) v[i * N + j] = i == j ? 1.f : 0.f;
  for (int sweep = 0; sweep < 60; ++sweep) {
    float off = 0.f, diag = 0.f;
    for (int i = 0; i < N; ++i) {
      diag += std::fabs(a[i * N + i]);
      for (int j = i + 1; j < N; ++j) off += std::fabs(a[i * N + j]);
    }
    if (!(off > 1e-12f * diag)) break;
    for (int p = 0; p < N; ++p)
      for (int q = p + 1; q < N; ++q) {
        const float apq = a[p * N + q];
        if (apq == 0.f) continue;
        const float theta = (a[q * N + q] - a[p * N + p]) / (2.f * apq);
        const float t = (theta >= 0.f ? 1.f : -1.f) / (std::fabs(theta) + std::sqrt(theta * theta + 1.f));
        const float c = 1.f / std::sqrt(t * t + 1.f), s = t * c;
        for (int k = 0; k < N; ++k) {
          const float x = a[k * N + p], y = a[k * N + q];
          a[k * N + p] = c * x - s * y, a[k * N + q] = s * x + c * y;
        }
        for (int k = 0; k < N; ++k) {
          const float x = a[p * N + k], y = a[q * N + k];
          a[p * N + k] = c * x - s * y, a[q * N + k] = s * x + c * y;
        }
        for (int k = 0; k < N; ++k) {
          const float x = v[k * N + p], y = v[k * N + q];
          v[k * N + p] = c * x - s * y, v[k * N + q] = s * x + c * y;
        }
      }
  }
  int ord[6] = {0, 1, 2, 3, 4, 5};
  for (int i = 1; i < N; ++i)
    for (int j = i; j > 0 && a[ord[j] * N + ord[j]] > a[ord[j - 1] * N + ord[j - 1]]; --j) std::swap(ord[j], ord[j - 1]);
  for (int i = 0; i < N; ++i) {
    w[i] = a[ord[i] * N + ord[i]];
    for (int k = 0; k < N; ++k) V[i * N + k] = v[k * N + ord[i]];
  }
}

void lm_qr6(float* a, float* b, float* x) {  // Householder QR solve of the 6x6 system (a, b destroyed)
  const int N = 6;
  for (int k = 0; k < N; ++k) {
    float nrm2 = 0.f;
    for (int i = k; i < N; ++i) nrm2 += a[i * N + k] * a[i * N + k];
    const float nrm = std::sqrt(nrm2);
    if (nrm == 0.f) continue;
    const float alpha = a[k * N + k] >= 0.f ? -nrm : nrm;
    float v[6];
    for (int i = 0; i < N; ++i) v[i] = i >= k ? a[i * N + k] : 0.f;
    v[k] -= alpha;
    float vv = 0.f;
    for (int i = k; i < N; ++i) vv += v[i] * v[i];
    if (vv == 0.f) continue;
    for (int j = k; j < N; ++j) {
      float s = 0.f;
      for (int i = k; i < N; ++i) s += v[i] * a[i * N + j];
      s = 2.f * s / vv;
      for (int i = k; i < N; ++i) a[i * N + j] -= s * v[i];
    }
    float s = 0.f;
    for (int i = k; i < N; ++i) s += v[i] * b[i];
    s = 2.f * s / vv;
    for (int i = k; i < N; ++i) b[i] -= s * v[i];
  }
  for (int i = N - 1; i >= 0; --i) {
    float s = b[i];
    for (int j = i + 1; j < N; ++j) s -= a[i * N + j] * x[j];
    x[i] = s / a[i * N + i];
  }
}

void lm_inv6(const float* A, float* inv) {  // Gauss-Jordan, partial pivoting
  const int n = 6;
  float a[36];
  std::memcpy(a, A, sizeof a);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) inv[i * n + j] = i == j ? 1.f : 0.f;
  for (int k = 0; k < n; ++k) {
    int p = k;
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(a[i * n + k]) > std::fabs(a[p * n + k])) p = i;
    if (p != k)
      for (int j = 0; j < n; ++j) std::swap(a[k * n + j], a[p * n + j]), std::swap(inv[k * n + j], inv[p * n + j]);
    const float d = a[k * n + k];
    for (int j = 0; j < n; ++j) a[k * n + j] /= d, inv[k * n + j] /= d;
    for (int i = 0; i < n; ++i) {
      if (i == k) continue;
      const float f = a[i * n + k];
      for (int j = 0; j < n; ++j) a[i * n + j] -= f * a[k * n + j], inv[i * n + j] -= f * inv[k * n + j];
    }
  }
}

struct LmCarry {
  bool degenerate = false;
  float P[36];
};

// the step from the 28 sums (upper triangle of A^T A, A^T b, row count); true = converged (LM:1583-1632)
bool lm_step_from_sums(const double* sums, int iter, float* T, LmCarry& st) {
  if ((int)sums[27] < 50) return false;  // LM:1530-1532
  float A[36], B[6], X[6], Aq[36], Bq[6];
  int t = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 6; ++j) A[i * 6 + j] = A[j * 6 + i] = (float)sums[t++];
  for (int i = 0; i < 6; ++i) B[i] = (float)sums[21 + i];
  std::memcpy(Aq, A, sizeof A), std::memcpy(Bq, B, sizeof B);
  lm_qr6(Aq, Bq, X);
  if (iter == 0) {
    float Ae[36], E[6], V[36], V2[36], Vi[36];
    std::memcpy(Ae, A, sizeof A);
    lm_eig6(Ae, E, V);
    std::memcpy(V2, V, sizeof V);
    st.degenerate = false;
    for (int i = 5; i >= 0; i--) {
      if (E[i] < 100) {
        for (int j = 0; j < 6; j++) V2[i * 6 + j] = 0;
        st.degenerate = true;
      } else {
        break;
      }
    }
    lm_inv6(V, Vi);
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) {
        float s = 0.f;
        for (int k = 0; k < 6; ++k) s += Vi[i * 6 + k] * V2[k * 6 + j];
        st.P[i * 6 + j] = s;
      }
  }
  if (st.degenerate) {
    float X2[6];
    std::memcpy(X2, X, sizeof X);
    for (int i = 0; i < 6; ++i) {
      float s = 0.f;
      for (int k = 0; k < 6; ++k) s += st.P[i * 6 + k] * X2[k];
      X[i] = s;
    }
  }
  for (int i = 0; i < 6; ++i) T[i] += X[i];
  auto rad2deg = [](float a) { return (float)(a * 57.29578f); };
  const float deltaR = std::sqrt(std::pow(rad2deg(X[0]), 2) + std::pow(rad2deg(X[1]), 2) + std::pow(rad2deg(X[2]), 2));
  const float deltaT = std::sqrt(std::pow(X[3] * 100, 2) + std::pow(X[4] * 100, 2) + std::pow(X[5] * 100, 2));
  return deltaR < 0.05 && deltaT < 0.05;
}

MapRoundHost make_round(const float* T) {
  MapRoundHost r;
  r.as.cRoll = std::cos(T[0]), r.as.sRoll = std::sin(T[0]);
  r.as.cPitch = std::cos(T[1]), r.as.sPitch = std::sin(T[1]);
  r.as.cYaw = std::cos(T[2]), r.as.sYaw = std::sin(T[2]);
  r.as.tX = T[3], r.as.tY = T[4], r.as.tZ = T[5];
  r.tg.srx = std::sin(T[0]), r.tg.crx = std::cos(T[0]);
  r.tg.sry = std::sin(T[1]), r.tg.cry = std::cos(T[1]);
  r.tg.srz = std::sin(T[2]), r.tg.crz = std::cos(T[2]);
  r.pad = 0.f;
  return r;
}

// counting sort of one map cloud into 1 m cells of the integer lattice
struct HostGrid {
  int cmin[3] = {0, 0, 0}, cdim[3] = {1, 1, 1};
  std::vector<int> cells;    // ncell + 1 starts
  std::vector<float4> pts;   // sorted, w = original index bits
};
int build_grid(const lins_point* p, int n, HostGrid& g) {
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) {
    if (!std::isfinite(p[i].x) || !std::isfinite(p[i].y) || !std::isfinite(p[i].z)) return LINS_E_INPUT;
    if (std::fabs(p[i].x) > 1e6f || std::fabs(p[i].y) > 1e6f || std::fabs(p[i].z) > 1e6f) return LINS_E_INPUT;
    const int c[3] = {(int)std::floor(p[i].x), (int)std::floor(p[i].y), (int)std::floor(p[i].z)};
    for (int a = 0; a < 3; ++a) lo[a] = i ? std::min(lo[a], c[a]) : c[a], hi[a] = i ? std::max(hi[a], c[a]) : c[a];
  }
  long long ncell = 1;
  for (int a = 0; a < 3; ++a) g.cmin[a] = lo[a], g.cdim[a] = hi[a] - lo[a] + 1, ncell *= g.cdim[a];
  if (ncell > (1ll << 26)) return LINS_E_CAPACITY;  // a map far larger than a local one (LM keeps ~50 key frames)
  g.cells.assign((size_t)ncell + 1, 0);
  auto cell_of = [&](const lins_point& q) {
    const int cx = (int)std::floor(q.x) - g.cmin[0], cy = (int)std::floor(q.y) - g.cmin[1], cz = (int)std::floor(q.z) - g.cmin[2];
    return ((size_t)cz * g.cdim[1] + cy) * g.cdim[0] + cx;
  };
  for (int i = 0; i < n; ++i) ++g.cells[cell_of(p[i]) + 1];
  for (size_t c = 0; c < (size_t)ncell; ++c) g.cells[c + 1] += g.cells[c];
  g.pts.resize(n);
  std::vector<int> fill(g.cells.begin(), g.cells.end() - 1);
  for (int i = 0; i < n; ++i) {
    float w;
    std::memcpy(&w, &i, 4);
    g.pts[fill[cell_of(p[i])]++] = make_float4(p[i].x, p[i].y, p[i].z, w);
  }
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

// uploads n problems (grids, queries); fills the device descriptors.  active[k] = precondition of LM:1636
int map_upload(lins_ctx* ctx, MapState* m, int n, const lins_map_problem* in, std::vector<MapDevHost>& dev, int* max_q) {
  for (int k = 0; k < n; ++k) {
    const lins_map_problem& p = in[k];
    if (p.n_map_corner < 0 || p.n_map_surf < 0 || p.n_scan_corner < 0 || p.n_scan_surf < 0) return LINS_E_ARG;
    if ((p.n_map_corner && !p.map_corner) || (p.n_map_surf && !p.map_surf) || (p.n_scan_corner && !p.scan_corner) ||
        (p.n_scan_surf && !p.scan_surf))
      return LINS_E_ARG;
  }
  std::vector<HostGrid> grids((size_t)n * 2);
  int rc = parallel_for(2 * n, [&](int j) {
    const lins_map_problem& p = in[j / 2];
    return (j & 1) ? build_grid(p.map_surf, p.n_map_surf, grids[j]) : build_grid(p.map_corner, p.n_map_corner, grids[j]);
  });
  if (rc) return rc;
  size_t tot_pts = 0, tot_cells = 0, tot_q = 0;
  dev.assign(n, MapDevHost{});
  *max_q = 0;
  for (int k = 0; k < n; ++k) {
    for (int w = 0; w < 2; ++w) {
      const HostGrid& g = grids[(size_t)k * 2 + w];
      dev[k].g[w].off_pts = (long long)tot_pts, dev[k].g[w].off_cells = (long long)tot_cells;
      for (int a = 0; a < 3; ++a) dev[k].g[w].cmin[a] = g.cmin[a], dev[k].g[w].cdim[a] = g.cdim[a];
      tot_pts += g.pts.size(), tot_cells += g.cells.size();
    }
    dev[k].off_q = dev[k].off_rec = (long long)tot_q;
    dev[k].n_q[0] = in[k].n_scan_corner, dev[k].n_q[1] = in[k].n_scan_surf;
    dev[k].active = in[k].n_map_corner > 10 && in[k].n_map_surf > 100;  // LM:1636
    for (int i = 0; i < in[k].n_scan_corner + in[k].n_scan_surf; ++i) {
      const lins_point& q = i < in[k].n_scan_corner ? in[k].scan_corner[i] : in[k].scan_surf[i - in[k].n_scan_corner];
      if (!std::isfinite(q.x) || !std::isfinite(q.y) || !std::isfinite(q.z)) return LINS_E_INPUT;
    }
    tot_q += (size_t)dev[k].n_q[0] + dev[k].n_q[1];
    *max_q = std::max(*max_q, dev[k].n_q[0] + dev[k].n_q[1]);
  }
  if ((rc = grow(ctx, &m->d_pts, &m->cap_pts, std::max<size_t>(tot_pts, 1)))) return rc;
  if ((rc = grow(ctx, &m->d_cells, &m->cap_cells, std::max<size_t>(tot_cells, 1)))) return rc;
  if (m->cap_q < std::max<size_t>(tot_q, 1)) {
    (void)hipFree(m->d_q), (void)hipFree(m->d_rec);
    m->d_q = nullptr, m->d_rec = nullptr, m->cap_q = 0;
    MAP_TRY(ctx, hipMalloc((void**)&m->d_q, std::max<size_t>(tot_q, 1) * sizeof(float4)));
    MAP_TRY(ctx, hipMalloc((void**)&m->d_rec, std::max<size_t>(tot_q, 1) * sizeof(lins_map_corr)));
    m->cap_q = std::max<size_t>(tot_q, 1);
  }
  if (m->cap_probs < n) {
    (void)hipFree(m->d_probs), (void)hipFree(m->d_rounds);
    m->d_probs = nullptr, m->d_rounds = nullptr, m->cap_probs = 0;
    MAP_TRY(ctx, hipMalloc(&m->d_probs, (size_t)n * sizeof(MapDevHost)));
    MAP_TRY(ctx, hipMalloc(&m->d_rounds, (size_t)n * sizeof(MapRoundHost)));
    m->cap_probs = n;
  }
  hipStream_t st = ctx_stream(ctx);
  for (int k = 0; k < n; ++k) {
    for (int w = 0; w < 2; ++w) {
      const HostGrid& g = grids[(size_t)k * 2 + w];
      if (!g.pts.empty())
        MAP_TRY(ctx, hipMemcpyAsync(m->d_pts + dev[k].g[w].off_pts, g.pts.data(), g.pts.size() * sizeof(float4), hipMemcpyHostToDevice, st));
      MAP_TRY(ctx, hipMemcpyAsync(m->d_cells + dev[k].g[w].off_cells, g.cells.data(), g.cells.size() * sizeof(int), hipMemcpyHostToDevice, st));
    }
    if (in[k].n_scan_corner)
      MAP_TRY(ctx, hipMemcpyAsync(m->d_q + dev[k].off_q, in[k].scan_corner, (size_t)in[k].n_scan_corner * sizeof(float4), hipMemcpyHostToDevice, st));
    if (in[k].n_scan_surf)
      MAP_TRY(ctx, hipMemcpyAsync(m->d_q + dev[k].off_q + in[k].n_scan_corner, in[k].scan_surf, (size_t)in[k].n_scan_surf * sizeof(float4), hipMemcpyHostToDevice, st));
  }
  MAP_TRY(ctx, hipStreamSynchronize(st));  // (the host grids go out of scope)
  return LINS_OK;
}

MapState* state_of(lins_ctx* ctx) {
  void** slot = ctx_map_slot(ctx, map_state_free);
  if (!*slot) *slot = new MapState();
  return (MapState*)*slot;
}

}  // namespace

extern "C" {

int lins_scan2map_batch(lins_ctx* ctx, int n, const lins_map_problem* in, lins_map_result* out) {
  if (!ctx || n < 0 || (n && (!in || !out))) return LINS_E_ARG;
  if (n == 0) return LINS_OK;
  static_assert(sizeof(MapDevHost) == 112 && sizeof(MapRoundHost) == 64 && sizeof(lins_map_corr) == 56, "layouts");
  if (map_dev_size() != sizeof(MapDevHost) || map_round_size() != sizeof(MapRoundHost)) return LINS_E_STATE;
  MAP_TRY(ctx, hipSetDevice(ctx_device(ctx)));
  MapState* m = state_of(ctx);
  std::vector<MapDevHost> dev;
  int max_q = 0;
  int rc = map_upload(ctx, m, n, in, dev, &max_q);
  if (rc) return rc;
  const int bpp = std::max(1, (max_q + map_block() - 1) / map_block());
  if ((rc = grow(ctx, &m->d_partials, &m->cap_partials, (size_t)n * bpp * 28))) return rc;
  hipStream_t st = ctx_stream(ctx);
  hipEvent_t e0, e1;
  ctx_events(ctx, &e0, &e1);
  std::vector<LmCarry> carry(n);
  std::vector<MapRoundHost> rounds(n);
  std::vector<double> partials((size_t)n * bpp * 28);
  for (int k = 0; k < n; ++k) {
    std::memcpy(out[k].transform, in[k].transform, sizeof out[k].transform);
    out[k].iters = 0, out[k].converged = 0, out[k].degenerate = 0, out[k].n_sel = 0;
  }
  m->ms = 0.f, m->queries = 0;
  for (int iter = 0; iter < 10; ++iter) {
    int n_active = 0;
    for (int k = 0; k < n; ++k) {
      rounds[k] = make_round(out[k].transform);
      n_active += dev[k].active;
      if (dev[k].active) m->queries += (uint64_t)dev[k].n_q[0] + dev[k].n_q[1];
    }
    if (!n_active) break;
    MAP_TRY(ctx, hipMemcpyAsync(m->d_probs, dev.data(), (size_t)n * sizeof(MapDevHost), hipMemcpyHostToDevice, st));
    MAP_TRY(ctx, hipMemcpyAsync(m->d_rounds, rounds.data(), (size_t)n * sizeof(MapRoundHost), hipMemcpyHostToDevice, st));
    MAP_TRY(ctx, hipEventRecord(e0, st));
    launch_map_corr(st, n, bpp, m->d_probs, m->d_rounds, m->d_pts, m->d_cells, m->d_q, m->d_rec, m->d_partials);
    MAP_TRY(ctx, hipGetLastError());
    MAP_TRY(ctx, hipEventRecord(e1, st));
    MAP_TRY(ctx, hipMemcpyAsync(partials.data(), m->d_partials, partials.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    MAP_TRY(ctx, hipStreamSynchronize(st));
    float ms = 0.f;
    MAP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    m->ms += ms;
    for (int k = 0; k < n; ++k) {
      if (!dev[k].active) continue;
      double sums[28] = {0};
      for (int b = 0; b < bpp; ++b)  // the per-block partials in block order
        for (int t = 0; t < 28; ++t) sums[t] += partials[((size_t)k * bpp + b) * 28 + t];
      out[k].n_sel = (int)sums[27];
      out[k].iters = iter + 1;
      if (lm_step_from_sums(sums, iter, out[k].transform, carry[k])) out[k].converged = 1, dev[k].active = 0;
      out[k].degenerate = carry[k].degenerate ? 1 : 0;
    }
  }
  return LINS_OK;
}

int lins_map_correspondences(lins_ctx* ctx, const lins_map_problem* in, lins_map_corr* corner, lins_map_corr* surf) {
  if (!ctx || !in || (in->n_scan_corner && !corner) || (in->n_scan_surf && !surf)) return LINS_E_ARG;
  if (map_dev_size() != sizeof(MapDevHost) || map_round_size() != sizeof(MapRoundHost)) return LINS_E_STATE;
  MAP_TRY(ctx, hipSetDevice(ctx_device(ctx)));
  MapState* m = state_of(ctx);
  std::vector<MapDevHost> dev;
  int max_q = 0;
  int rc = map_upload(ctx, m, 1, in, dev, &max_q);
  if (rc) return rc;
  dev[0].active = 1;  // a single pass is evaluated whatever the map sizes
  const int bpp = std::max(1, (max_q + map_block() - 1) / map_block());
  if ((rc = grow(ctx, &m->d_partials, &m->cap_partials, (size_t)bpp * 28))) return rc;
  hipStream_t st = ctx_stream(ctx);
  const MapRoundHost rd = make_round(in->transform);
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
