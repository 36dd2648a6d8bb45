// map_kernels.hip — scan-to-map correspondences on the device: cornerOptimization /
// surfOptimization (LM:1351-1521) and the rows + normal-equation sums of LMOptimization (LM:1543-1584).
//
// The local map (1e4 .. 1e5 points per cloud) is bucketed into 1 m cells of the integer lattice
// (host-built counting sort, once per problem: the reference rebuilds its kd-trees per scan too,
// LM:1637-1638).  A correspondence only counts when the FIFTH neighbour is closer than 1 m
// (pointSearchSqDis[4] < 1.0, LM:1360 / 1464), so the exact 5-NN it needs lies in the 27 cells around
// the query's cell: kMapLanes lanes per query scan those cells, each keeps the five smallest
// (squared distance, index) keys it saw — the order FLANN's result is restated with — the lists are
// merged by shuffles, and the query runs the eigen- / plane-fit of map_math.h in registers.  The accepted rows go straight into the 21 + 6 sums
// of A^T A, A^T b (f64, fixed-shape tree per block; the host adds the per-block partials in order).
#include <hip/hip_runtime.h>

#include "lm_math.h"
#include "lm_wave.h"
#include "map_math.h"

namespace lins {

struct MapGrid {  // one cloud of one problem
  long long off_pts;    // first sorted point (x, y, z, original index bits) in the point arena
  long long off_cells;  // first of (ncell + 1) cell starts in the cell arena (positions relative to off_pts)
  int cmin[3], cdim[3];
};
struct MapDev {  // one problem
  MapGrid g[2];      // 0 corner map, 1 surf map
  long long off_q;   // queries: corner scan points, then surf scan points
  long long off_rec; // lins_map_corr records, same order
  int n_q[2];
  int active;        // 0: finished / precondition not met — its blocks return at once
  int pad;
};
struct MapRound {
  MapAssoc as;
  MapTrig tg;
  float pad;
};

constexpr int kMapBlock = 256;
constexpr int kMapLanes = 8;  // lanes per query: they share the scan of its 27 cells, then merge their five-best lists
constexpr int kMapQPerBlock = kMapBlock / kMapLanes;

__global__ __launch_bounds__(kMapBlock) void map_corr_kernel(const MapDev* __restrict__ probs, const MapRound* __restrict__ rounds,
                                                             const float4* __restrict__ pts, const int* __restrict__ cells,
                                                             const float4* __restrict__ queries, lins_map_corr* __restrict__ recs,
                                                             double* __restrict__ partials, int blocks_per_problem, int n_problems) {
  // XCD-aware block -> (problem, block) mapping.  Workgroup b of a launch runs on XCD b % 8 (observed placement; it only
  // matters for speed), every XCD has an L2 of its own, and all blocks of a problem read the same ~0.5 MB of map: with
  // the problem index in blockIdx.y (rounds 1-4) the blocks of one problem were dealt over all eight XCDs and every L2
  // fetched every map — 2-4 x the maps' bytes per dispatch (profiles/r04_rocprofv3_pmc_aux.txt: 33 MB counted for 17 MB).
  // Here the blocks b, b + 8, b + 16, ... of one XCD walk through the problems x, x + 8, x + 16, ... block by block: a
  // problem's map is fetched into ONE L2 (four problems of 0.5 MB per XCD at a time when 32 are in flight).
  const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
  const int prob = (kk / blocks_per_problem) * 8 + xcd, blk = kk % blocks_per_problem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (prob >= n_problems) return;  // (the problem count is not a multiple of eight: this XCD's last group is short)
  const MapDev pd = probs[prob];
  double* out = partials + ((size_t)prob * blocks_per_problem + blk) * 28;
  if (!pd.active) {
    if (tid < 28) out[tid] = 0.0;
    return;
  }
  const MapRound rd = rounds[prob];
  const int q = blk * kMapQPerBlock + tid / kMapLanes, sub = tid % kMapLanes, nq = pd.n_q[0] + pd.n_q[1];
  double v[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) v[k] = 0.0;
  if (q < nq) {
    const int which = q < pd.n_q[0] ? 0 : 1;
    const MapGrid g = pd.g[which];
    const float4 po = queries[pd.off_q + q];
    float sx, sy, sz;
    map_associate(rd.as, po.x, po.y, po.z, sx, sy, sz);
    // the five smallest keys (squared-distance bits << 32 | original index): distances are >= 0, so the
    // unsigned 64-bit order is the (distance, index) order; pos = where the point sits in the sorted map
    const unsigned long long kNoKey = ((unsigned long long)0x7F800000u << 32) | 0xFFFFFFFFull;
    unsigned long long key[5] = {kNoKey, kNoKey, kNoKey, kNoKey, kNoKey};
    int pos[5] = {-1, -1, -1, -1, -1};
    const int cx = (int)floorf(sx) - g.cmin[0], cy = (int)floorf(sy) - g.cmin[1], cz = (int)floorf(sz) - g.cmin[2];
    const float4* gp = pts + g.off_pts;
    const int* gc = cells + g.off_cells;
    for (int dz = -1; dz <= 1; ++dz)
      for (int dy = -1; dy <= 1; ++dy) {
        const int iz = cz + dz, iy = cy + dy;
        if (iz < 0 || iz >= g.cdim[2] || iy < 0 || iy >= g.cdim[1]) continue;
        // the three x-neighbours are consecutive cells: one contiguous span of points
        int x0 = cx - 1, x1 = cx + 1;
        x0 = x0 < 0 ? 0 : x0, x1 = x1 >= g.cdim[0] ? g.cdim[0] - 1 : x1;
        if (x0 > x1) continue;
        const int row = (iz * g.cdim[1] + iy) * g.cdim[0];
        const int s = gc[row + x0], e = gc[row + x1 + 1];
        for (int p = s + sub; p < e; p += kMapLanes) {  // (a group's lanes read consecutive points)
          const float4 t = gp[p];
          const float ddx = sx - t.x, ddy = sy - t.y, ddz = sz - t.z;
          const float d = (ddx * ddx + ddy * ddy) + ddz * ddz;
          unsigned long long ck = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(t.w);
          if (!(ck < key[4])) continue;
          int cp = p;  // insertion, unrolled so that the arrays stay in registers
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            const bool before = ck < key[k];
            const unsigned long long tk = key[k];
            const int tp = pos[k];
            key[k] = before ? ck : tk, pos[k] = before ? cp : tp;
            ck = before ? tk : ck, cp = before ? tp : cp;
          }
        }
      }
    // fold the lanes' lists: after log2(kMapLanes) exchanges every lane of the query holds the same five best
    // (the lists are disjoint — every point was scanned by exactly one lane — so nothing is inserted twice)
#pragma unroll
    for (int m = 1; m < kMapLanes; m <<= 1) {
      unsigned long long ok[5];
      int op[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) ok[k] = __shfl_xor(key[k], m), op[k] = __shfl_xor(pos[k], m);
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        unsigned long long ck = ok[i];
        int cp = op[i];
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
    }
    lins_map_corr r;
    r.sel[0] = sx, r.sel[1] = sy, r.sel[2] = sz;
    r.accepted = 0;
    r.coeff[0] = r.coeff[1] = r.coeff[2] = r.coeff[3] = 0.f;
    const float sq5 = __uint_as_float((unsigned)(key[4] >> 32));
    if (sq5 < 1.0) {
      float px[5], py[5], pz[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const float4 t = gp[pos[k]];
        px[k] = t.x, py[k] = t.y, pz[k] = t.z;
        r.ind[k] = (int)(unsigned)key[k];
      }
#ifdef LINS_MAP_GATHER_BARRIER  // (canary experiments: the gathered neighbours are opaque to the optimiser from here)
#pragma unroll
      for (int k = 0; k < 5; ++k) asm volatile("" : "+v"(px[k]), "+v"(py[k]), "+v"(pz[k]));
#endif
      r.sq5 = sq5;
      float c[4];
      r.accepted = which == 0 ? map_corner_fit(px, py, pz, sx, sy, sz, c) : map_surf_fit(px, py, pz, sx, sy, sz, c);
      r.coeff[0] = c[0], r.coeff[1] = c[1], r.coeff[2] = c[2], r.coeff[3] = c[3];
      if (r.accepted && sub == 0) {  // (the fit ran on every lane of the query, identically: one of them counts)
        float row[6], b;
        map_lm_row(rd.tg, po.x, po.y, po.z, c, row, b);
        int t = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
          for (int j = i; j < 6; ++j) v[t++] = (double)row[i] * (double)row[j];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[21 + i] = (double)row[i] * (double)b;
        v[27] = 1.0;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 5; ++k) r.ind[k] = -1;
      r.sq5 = INFINITY;
    }
    if (sub == 0) recs[pd.off_rec + q] = r;
  }
  // 28 sums: wave butterfly, then the four waves in order
  __shared__ double wsum[kMapBlock / 64][28];
#pragma unroll
  for (int k = 0; k < 28; ++k) {
    double x = v[k];
    for (int o = 32; o >= kMapLanes; o >>= 1) {  // (only lane 0 of a query carries a row)
      const int lo = __shfl_xor(__double2loint(x), o), hi = __shfl_xor(__double2hiint(x), o);
      x += __hiloint2double(hi, lo);
    }
    if (lane == 0) wsum[wave][k] = x;
  }
  __syncthreads();
  if (tid < 28) {
    double s = 0.0;
    for (int w = 0; w < kMapBlock / 64; ++w) s += wsum[w][tid];
    out[tid] = s;
  }
}

// ---------------------------------------------------------------------------
// The local map's 1 m gridding on the device (what the reference's kdtree->setInputCloud stands for, LM:1637-1638):
// one 1024-thread workgroup per cloud counting-sorts the raw points into the cells of the integer lattice — zero
// the cursors, histogram (L2 atomics), exclusive scan, scatter.  The order inside a cell is whatever the atomics
// give: the 5-NN is decided on (distance, index) keys, so it does not matter.
// ---------------------------------------------------------------------------
struct MapGridJob {
  long long off_raw;   // raw points of this cloud in the staging arena
  long long off_pts;   // sorted points
  long long off_cells; // ncell + 1 starts (relative to off_pts), followed by ncell + 1 scratch cursors
  int n, ncell;
  int cmin[3], cdim[3];
};
constexpr int kGridBlock = 1024;

__global__ __launch_bounds__(kGridBlock) void map_grid_kernel(const MapGridJob* __restrict__ jobs, const float4* __restrict__ raw,
                                                              float4* __restrict__ pts, int* __restrict__ cells) {
  const MapGridJob jb = jobs[blockIdx.x];
  const int tid = threadIdx.x;
  int* const starts = cells + jb.off_cells;
  int* const cursor = starts + jb.ncell + 1;
  const float4* const src = raw + jb.off_raw;
  for (int c = tid; c <= jb.ncell; c += kGridBlock) cursor[c] = 0;
  __syncthreads();
  auto cell_of = [&](const float4& q) {
    const int cx = (int)floorf(q.x) - jb.cmin[0], cy = (int)floorf(q.y) - jb.cmin[1], cz = (int)floorf(q.z) - jb.cmin[2];
    return (cz * jb.cdim[1] + cy) * jb.cdim[0] + cx;
  };
  for (int i = tid; i < jb.n; i += kGridBlock) atomicAdd(&cursor[cell_of(src[i])], 1);
  __syncthreads();
  // exclusive scan: every thread owns a contiguous run of cells
  __shared__ int chunk_sum[kGridBlock];
  const int per = (jb.ncell + kGridBlock - 1) / kGridBlock;
  const int c_lo = tid * per < jb.ncell ? tid * per : jb.ncell, c_hi = c_lo + per < jb.ncell ? c_lo + per : jb.ncell;
  int local = 0;
  for (int c = c_lo; c < c_hi; ++c) local += cursor[c];
  chunk_sum[tid] = local;
  __syncthreads();
  for (int o = 1; o < kGridBlock; o <<= 1) {  // Hillis-Steele inclusive scan of the chunk sums
    const int v = tid >= o ? chunk_sum[tid - o] : 0;
    __syncthreads();
    chunk_sum[tid] += v;
    __syncthreads();
  }
  int run = chunk_sum[tid] - local;
  for (int c = c_lo; c < c_hi; ++c) {
    const int cnt = cursor[c];
    starts[c] = run, cursor[c] = run;
    run += cnt;
  }
  if (tid == 0) starts[jb.ncell] = jb.n;
  __syncthreads();
  float4* const dst = pts + jb.off_pts;
  for (int i = tid; i < jb.n; i += kGridBlock) {
    const float4 p = src[i];
    const int pos = atomicAdd(&cursor[cell_of(p)], 1);
    dst[pos] = make_float4(p.x, p.y, p.z, __int_as_float(i));
  }
}

// ---------------------------------------------------------------------------
// LMOptimization's step between two correspondence rounds (LM:1583-1632), one thread per problem: the per-block
// partials are added in block order, the 6x6 step of lm_math.h moves the transform, and the next round's rotation
// terms / the problem's "still running" flag are written where the correspondence kernel reads them — the ten
// rounds of scan2MapOptimization run back to back without the host.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void map_lm_kernel(int n, int iter, int blocks_per_problem, MapDev* __restrict__ probs,
                                                    MapRound* __restrict__ rounds, const double* __restrict__ partials,
                                                    lins_map_result* __restrict__ results, LmCarry* __restrict__ carry) {
  // One WAVE per problem (round 3: lm_wave.h — element (i, j) of the 6 x 6 matrices in lane 6 i + j; rounds 1-2 ran
  // lm_math.h's one-thread definition here, ~100 us for round 0).  Everything outside the matrices is uniform.
  __shared__ double sums[28];
  const int k = blockIdx.x, lane = threadIdx.x;
  lins_map_result r = results[k];
  if (iter < 0) {  // initialisation: rotation terms of the given transform, counters
    r.iters = 0, r.converged = 0, r.degenerate = 0, r.n_sel = 0;
    if (lane == 0) results[k] = r, carry[k].degenerate = 0;
  } else {
    if (!probs[k].active) return;  // (uniform)
    if (lane < 28) {
      double s = 0.0;
      for (int b = 0; b < blocks_per_problem; ++b) s += partials[((size_t)k * blocks_per_problem + b) * 28 + lane];
      sums[lane] = s;
    }
    __syncthreads();
    r.n_sel = (int)sums[27];
    r.iters = iter + 1;
    float T[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) T[t] = r.transform[t];
    int degenerate = 0;
    const bool conv = wave_lm_step_from_sums(sums, iter, T, carry + k, lane, degenerate);
#pragma unroll
    for (int t = 0; t < 6; ++t) r.transform[t] = T[t];
    if (conv) r.converged = 1;
    r.degenerate = degenerate;
    if (lane == 0) {
      if (conv) probs[k].active = 0;
      results[k] = r;
    }
  }
  const MapRoundParams p = lm_make_round(r.transform);
  if (lane == 0) {
    MapRound rd;
    rd.as = p.as, rd.tg = p.tg, rd.pad = 0.f;
    rounds[k] = rd;
  }
}

// Debug aid (tests/test_gpu_math.py): lm_step_from_sums by one thread (lm_math.h, the definition) / over a wave
// (lm_wave.h, what map_lm_kernel runs) on the same input: 28 sums, round, T[6], carry.degenerate, carry.P[36] (72
// doubles) -> converged, T[6], degenerate, P[36] (44 doubles)
__global__ __launch_bounds__(64) void debug_lm_step_kernel(int wave_version, const double* __restrict__ in, double* __restrict__ out,
                                                           LmCarry* __restrict__ scratch) {
  __shared__ double sums[28];
  const int item = blockIdx.x, lane = threadIdx.x;
  const double* a = in + (size_t)item * 72;
  if (lane < 28) sums[lane] = a[lane];
  LmCarry* c = scratch + item;
  if (lane < 36) c->P[lane] = (float)a[36 + lane];
  if (lane == 0) c->degenerate = (int)a[35];
  __syncthreads();
  const int iter = (int)a[28];
  float T[6];
  for (int t = 0; t < 6; ++t) T[t] = (float)a[29 + t];
  bool conv = false;
  if (wave_version) {
    int deg = 0;
    conv = wave_lm_step_from_sums(sums, iter, T, c, lane, deg);
  } else if (lane == 0) {
    LmCarry cc = *c;
    conv = lm_step_from_sums(sums, iter, T, cc);
    *c = cc;
  }
  __syncthreads();
  if (lane == 0) {
    double* o = out + (size_t)item * 44;
    o[0] = conv ? 1.0 : 0.0;
    for (int t = 0; t < 6; ++t) o[1 + t] = (double)T[t];
    o[7] = (double)c->degenerate;
    for (int t = 0; t < 36; ++t) o[8 + t] = (double)c->P[t];
  }
}
void launch_debug_lm_step(hipStream_t stream, int n, int wave_version, const double* in, double* out, void* scratch) {
  hipLaunchKernelGGL(debug_lm_step_kernel, dim3(n), dim3(64), 0, stream, wave_version, in, out, (LmCarry*)scratch);
}

void launch_map_grid(hipStream_t stream, int n_jobs, const void* jobs, const float4* raw, float4* pts, int* cells) {
  hipLaunchKernelGGL(map_grid_kernel, dim3(n_jobs), dim3(kGridBlock), 0, stream, (const MapGridJob*)jobs, raw, pts, cells);
}
void launch_map_lm(hipStream_t stream, int n, int iter, int blocks_per_problem, void* probs, void* rounds, const double* partials,
                   lins_map_result* results, void* carry) {
  hipLaunchKernelGGL(map_lm_kernel, dim3(n), dim3(64), 0, stream, n, iter, blocks_per_problem, (MapDev*)probs,
                     (MapRound*)rounds, partials, results, (LmCarry*)carry);
}
size_t map_grid_job_size() { return sizeof(MapGridJob); }
size_t map_carry_size() { return sizeof(LmCarry); }

void launch_map_corr(hipStream_t stream, int n_problems, int blocks_per_problem, const void* probs, const void* rounds,
                     const float4* pts, const int* cells, const float4* queries, lins_map_corr* recs, double* partials) {
  // (one-dimensional grid: 8 x blocks_per_problem x ceil(n_problems / 8) — the XCD-aware mapping of the kernel)
  hipLaunchKernelGGL(map_corr_kernel, dim3(8 * blocks_per_problem * ((n_problems + 7) / 8)), dim3(kMapBlock), 0, stream, (const MapDev*)probs,
                     (const MapRound*)rounds, pts, cells, queries, recs, partials, blocks_per_problem, n_problems);
}
// Start-up self-check of the plane fit: five points of the wall y = 2 and a query 5 cm in front of it must give a
// normal along y.  (ROCm 7.2's SLP vectoriser loses the y column of the unrolled 5 x 3 QR at -O2 and above — this file
// is built with -fno-slp-vectorize; the check catches a toolchain or flag change that brings the miscompile back.)
__global__ void map_selfcheck_kernel(float* __restrict__ out) {
  const float px[5] = {0.3f, -0.4f, 0.1f, 0.5f, -0.2f}, py[5] = {2.f, 2.f, 2.f, 2.f, 2.f}, pz[5] = {0.2f, 0.1f, -0.3f, -0.1f, 0.4f};
  float c[4];
  const int acc = map_surf_fit(px, py, pz, 0.05f, 2.05f, 0.02f, c);
  out[0] = c[0], out[1] = c[1], out[2] = c[2], out[3] = c[3], out[4] = (float)acc;
}
void launch_map_selfcheck(hipStream_t stream, float* out) { hipLaunchKernelGGL(map_selfcheck_kernel, dim3(1), dim3(1), 0, stream, out); }
size_t map_dev_size() { return sizeof(MapDev); }
size_t map_round_size() { return sizeof(MapRound); }
int map_block() { return kMapQPerBlock; }  // queries per block

}  // namespace lins
