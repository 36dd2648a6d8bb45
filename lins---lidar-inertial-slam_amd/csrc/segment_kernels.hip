// segment_kernels.hip — image_projection_node on the device: raw VLP-16 cloud -> segmented cloud +
// cloud_info, the input of the feature front-end (frontend_kernels.hip).  Restates
//   findStartEndAngle IP:191-203, projectPointCloud IP:205-241, groundRemoval IP:243-287,
//   cloudSegmentation IP:289-334, labelComponents IP:336-415
// one 1024-thread workgroup per scan.
//
// The reference labels segments with a sequential BFS over the 16 x 1800 range image, seeds in
// raster order, neighbours from a table whose (-1) entries are stored in uint8 (IP:72, 133-144): the
// "row above" neighbour never exists, the "left" neighbour is column + 255 (column 0 once that
// passes 1800), so the adjacency is DIRECTED: right (wrapping), +255 (-> 0), down.  A cell joins the
// flood of the first seed that reaches it, and a cell is a seed iff no earlier flood reached it.
// Because reachability is transitive, that is exactly: label(x) = the smallest raster index among
// the cells from which x is reachable (itself included) — a min-label propagation along the
// directed edges, which is order-free and runs in parallel: every thread owns a contiguous run of
// cells and pulls the minimum over the in-edges of each, run after run, until nothing changes.
// Segment validity (>= 30 cells, or >= 5 cells whose pushed neighbours span >= 3 rows, IP:398-406)
// comes from per-label counters; only validity survives into the outputs, not the label values.

#include <hip/hip_runtime.h>

#include <cfloat>

#include "../../include/lins_host.h"
#include "lins_math.h"

namespace lins {

constexpr int kSgRows = LINS_LINE_NUM, kSgCols = LINS_SCAN_NUM, kSgCells = kSgRows * kSgCols;
constexpr int kSgBlock = 1024;
constexpr int kSgGroundScanInd = 5;

struct FeScanOut {  // the head of FeScan (frontend_kernels.hip): what this kernel fills in
  long long off;
  int n;
  int start_ring[kSgRows], end_ring[kSgRows];
  float start_ori, end_ori, ori_diff;
};
constexpr size_t kFeScanStride = 192;  // sizeof(FeScan)

struct SgRaw {
  long long off;  // first raw point of the scan
  int n;
  int pad;
};

struct SgConsts {
  float sin_ax, cos_ax, sin_ay, cos_ay;  // sin / cos of segmentAlphaX / Y as the host computes them
  float theta;                           // segmentTheta
};

struct SgLds {
  signed char ground[kSgCells];    // groundMat: 0, 1, -1
  unsigned char edges[kSgCells];   // bit 0 right, bit 1 "+255", bit 2 down; bit 7 eligible (labelMat == 0 at the start)
  unsigned short label[kSgCells];  // smallest raster index that reaches the cell (0xFFFF: not eligible)
  int changed;
  int scan_tmp[20];
  int ring_count[kSgRows + 1];
  int n_outlier;
};
static_assert(sizeof(SgLds) <= 160 * 1024, "LDS budget");
__shared__ SgLds g_sg;

__device__ __forceinline__ int sg_block_scan(int v, int tid, int* tmp) {  // exclusive; total in tmp[18]
  const int lane = tid & 63, wave = tid >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int nb = __shfl_up(incl, o, 64);
    if (lane >= o) incl += nb;
  }
  if (lane == 63) tmp[wave] = incl;
  __syncthreads();
  int off = 0, tot = 0;
  for (int w = 0; w < kSgBlock / 64; ++w) {
    if (w < wave) off += tmp[w];
    tot += tmp[w];
  }
  __syncthreads();
  tmp[18] = tot;
  return off + incl - v;
}

__global__ __launch_bounds__(kSgBlock) void segment_kernel(const SgRaw* __restrict__ raws, const float4* __restrict__ raw,
                                                           SgConsts k, int* __restrict__ cellidx, float4* __restrict__ full,
                                                           float* __restrict__ rangeMat, int* __restrict__ seg_count,
                                                           int* __restrict__ seg_rows, unsigned char* __restrict__ fe_scans,
                                                           float4* __restrict__ out_cloud, float* __restrict__ out_range,
                                                           unsigned* __restrict__ out_col, unsigned char* __restrict__ out_ground,
                                                           int* __restrict__ out_outliers) {
  SgLds& L = g_sg;
#ifdef LINS_SG_PROF
  long long sg_t0 = clock64();
  int sg_sweeps = 0;
#endif
  const int tid = threadIdx.x, scan = blockIdx.x;
  const SgRaw rw = raws[scan];
  const float4* pts = raw + rw.off;
  const int n = rw.n;
  int* ci = cellidx + (size_t)scan * kSgCells;
  float4* fl = full + (size_t)scan * kSgCells;
  float* rm = rangeMat + (size_t)scan * kSgCells;
  int* cnt = seg_count + (size_t)scan * kSgCells;
  int* rws = seg_rows + (size_t)scan * kSgCells;
  FeScanOut* fo = reinterpret_cast<FeScanOut*>(fe_scans + (size_t)scan * kFeScanStride);
  const double kPi = 3.14159265358979323846;

  for (int c = tid; c < kSgCells; c += kSgBlock) ci[c] = -1, cnt[c] = 0, rws[c] = 0;
  if (tid == 0) L.n_outlier = 0;
  __syncthreads();

#ifdef LINS_SG_PROF
  if (tid == 0 && scan == 0) { long long t_ = clock64(); printf("SG %d %lld\n", 0, t_ - sg_t0); sg_t0 = t_; }
#endif
  // ---- projectPointCloud (IP:205-241): the LAST point that falls into a cell owns it -------------
#pragma unroll 4
  for (int i = tid; i < n; i += kSgBlock) {
    const float4 p = pts[i];
    const float vert = (float)((double)(lins_atan2f(p.z, sqrtf(p.x * p.x + p.y * p.y)) * 180) / kPi);
    const float rowf = (vert + (15.0f + 0.1f)) / 2.0f;
    if (rowf < 0 || rowf >= kSgRows) continue;
    const int row = (int)rowf;
    const float horizon = (float)((double)(lins_atan2f(p.x, p.y) * 180) / kPi);
    int colm = (int)(-round(((double)horizon - 90.0) / (double)0.2f) + kSgCols / 2);
    if (colm >= kSgCols) colm -= kSgCols;
    if (colm < 0 || colm >= kSgCols) continue;
    atomicMax(&ci[colm + row * kSgCols], i);
  }
  __threadfence_block();
  __syncthreads();
#pragma unroll 4
  for (int c = tid; c < kSgCells; c += kSgBlock) {
    const int i = ci[c];
    float4 p = make_float4(NAN, NAN, NAN, -1.f);
    float r = FLT_MAX;
    if (i >= 0) {
      p = pts[i];
      r = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
      const int row = c / kSgCols, colm = c - row * kSgCols;
      p.w = (float)((double)(float)row + (double)(float)colm / 10000.0);
    }
    fl[c] = p, rm[c] = r;
  }
  __threadfence_block();
  __syncthreads();

#ifdef LINS_SG_PROF
  if (tid == 0 && scan == 0) { long long t_ = clock64(); printf("SG %d %lld\n", 1, t_ - sg_t0); sg_t0 = t_; }
#endif
  // ---- groundRemoval (IP:243-278): one thread per column, rows bottom-up (a row is rewritten by the next) ----
  for (int c = tid; c < kSgCells; c += kSgBlock) L.ground[c] = 0;
  __syncthreads();
  for (int j = tid; j < kSgCols; j += kSgBlock)
    for (int i = 0; i < kSgGroundScanInd; ++i) {
      const int lo = j + i * kSgCols, up = j + (i + 1) * kSgCols;
      const float4 a = fl[lo], b = fl[up];
      if (a.w == -1.f || b.w == -1.f) {
        L.ground[lo] = -1;
        continue;
      }
      const float dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
      const float angle = (float)((double)(lins_atan2f(dz, sqrtf(dx * dx + dy * dy)) * 180) / kPi);
      if (fabsf(angle - 0.0f) <= 10) L.ground[lo] = 1, L.ground[up] = 1;
    }
  __syncthreads();

#ifdef LINS_SG_PROF
  if (tid == 0 && scan == 0) { long long t_ = clock64(); printf("SG %d %lld\n", 2, t_ - sg_t0); sg_t0 = t_; }
#endif
  // ---- adjacency of labelComponents (IP:336-415) as three edge bits per eligible cell ------------------
#pragma unroll 4
  for (int c = tid; c < kSgCells; c += kSgBlock) {
    const bool elig = !(L.ground[c] == 1 || rm[c] == FLT_MAX);
    L.edges[c] = elig ? 0x80 : 0;
    L.label[c] = elig ? (unsigned short)c : (unsigned short)0xFFFF;
  }
  __syncthreads();
  auto target = [&](int c, int dir) {  // dir 0: (0, +1)   1: (0, +255)   2: (+1, 0);  -1 if outside
    const int r = c / kSgCols, col = c - r * kSgCols;
    if (dir == 2) return r + 1 < kSgRows ? c + kSgCols : -1;
    int tc = col + (dir == 0 ? 1 : 255);
    if (tc >= kSgCols) tc = 0;
    return r * kSgCols + tc;
  };
#pragma unroll 4
  for (int c = tid; c < kSgCells; c += kSgBlock) {
    if (!(L.edges[c] & 0x80)) continue;
    unsigned char e = 0x80;
    const float rc = rm[c];
    for (int dir = 0; dir < 3; ++dir) {
      const int t = target(c, dir);
      if (t < 0 || !(L.edges[t] & 0x80)) continue;
      const float rt = rm[t];
      const float d1 = fmaxf(rc, rt), d2 = fminf(rc, rt);
      const float sa = dir == 2 ? k.sin_ay : k.sin_ax, ca = dir == 2 ? k.cos_ay : k.cos_ax;
      const float angle = lins_atan2f(d2 * sa, d1 - d2 * ca);
      if (angle > k.theta) e |= (unsigned char)(1 << dir);
    }
    L.edges[c] = e;  // (only this thread writes the low bits of its cells; bit 7 is read by others and unchanged)
  }
  __syncthreads();

#ifdef LINS_SG_PROF
  if (tid == 0 && scan == 0) { long long t_ = clock64(); printf("SG %d %lld\n", 3, t_ - sg_t0); sg_t0 = t_; }
#endif
  // ---- min-label propagation: pull over the in-edges, own run of cells in raster order, until stable ----
  constexpr int kRun = (kSgCells + kSgBlock - 1) / kSgBlock;  // 29 cells per thread
  const int c_lo = tid * kRun < kSgCells ? tid * kRun : kSgCells;
  const int c_hi = c_lo + kRun < kSgCells ? c_lo + kRun : kSgCells;
  for (int sweep = 0; sweep < 4096; ++sweep) {
    if (tid == 0) L.changed = 0;
    __syncthreads();
    bool ch = false;
    {  // column 0 of ring r collects the "+255" edges of every column whose col + 255 passes 1800: one wave per ring
      const int r = tid >> 6, lane = tid & 63;
      const int c0 = r * kSgCols;
      if (r < kSgRows && (L.edges[c0] & 0x80)) {
        unsigned best = 0xFFFFu;
        for (int cc = kSgCols - 255 + lane; cc < kSgCols; cc += 64)
          if (L.edges[c0 + cc] & 2) best = min(best, (unsigned)L.label[c0 + cc]);
        for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
        if (lane == 0 && best < L.label[c0]) L.label[c0] = (unsigned short)best, ch = true;
      }
    }
    __syncthreads();
    for (int c = c_lo; c < c_hi; ++c) {
      if (!(L.edges[c] & 0x80)) continue;
      const int r = c / kSgCols, col = c - r * kSgCols;
      unsigned short best = L.label[c];
      // in-edges: (r, col - 1) right [col 0: (r, 1799)], (r, col - 255) "+255" [col >= 255], (r - 1, col) down
      const int pl = col ? c - 1 : c + kSgCols - 1;
      if (L.edges[pl] & 1) best = min(best, L.label[pl]);
      if (col >= 255 && (L.edges[c - 255] & 2)) best = min(best, L.label[c - 255]);
      if (r > 0 && (L.edges[c - kSgCols] & 4)) best = min(best, L.label[c - kSgCols]);
      if (best != L.label[c]) L.label[c] = best, ch = true;
    }
    if (ch) L.changed = 1;
    __syncthreads();
#ifdef LINS_SG_PROF
    ++sg_sweeps;
#endif
    if (!L.changed) break;
    __syncthreads();
  }

#ifdef LINS_SG_PROF
  if (tid == 0 && scan == 0) { long long t_ = clock64(); printf("SG %d %lld\n", 4, t_ - sg_t0); sg_t0 = t_; }
#endif
  // ---- segment validity (IP:398-406): size, and rows of the cells that were pushed as neighbours -------
#pragma unroll 4
  for (int c = tid; c < kSgCells; c += kSgBlock) {
    if (!(L.edges[c] & 0x80)) continue;
    const int s = L.label[c];
    atomicAdd(&cnt[s], 1);
    if (c != s) atomicOr(&rws[s], 1 << (c / kSgCols));
  }
  __threadfence_block();
  __syncthreads();

#ifdef LINS_SG_PROF
  if (tid == 0 && scan == 0) { long long t_ = clock64(); printf("SG %d %lld\n", 5, t_ - sg_t0); sg_t0 = t_; }
#endif
  // ---- cloudSegmentation emission (IP:292-321): ring-major, ascending column -----------------------------
  auto decide = [&](int c, bool& is_ground) {  // emitted? (also counts the outliers of invalid segments)
    const int i = c / kSgCols, j = c - i * kSgCols;
    is_ground = false;
    if (L.edges[c] & 0x80) {
      const int s = L.label[c], np = cnt[s];
      const bool feasible = np >= 30 || (np >= 5 && __popc(rws[s]) >= 3);
      if (!feasible) {
        if (i > kSgGroundScanInd && j % 5 == 0) atomicAdd(&L.n_outlier, 1);
        return false;
      }
      return true;
    }
    if (L.ground[c] == 1) {
      is_ground = true;
      return !(j % 5 != 0 && j > 5 && j < kSgCols - 5);
    }
    return false;
  };
  int mine = 0;
  for (int c = c_lo; c < c_hi; ++c) {
    bool g;
    mine += decide(c, g) ? 1 : 0;
  }
  // (decide() counted outliers once here; the second pass below must not count again)
  const int base = sg_block_scan(mine, tid, L.scan_tmp);
  const int total = L.scan_tmp[18];
  const size_t ob = (size_t)fo->off;
  int pos = base;
  for (int c = c_lo; c < c_hi; ++c) {
    const int i = c / kSgCols, j = c - i * kSgCols;
    if (j == 0) L.ring_count[i] = pos;  // points emitted before ring i
    bool emit = false, is_ground = false;
    if (L.edges[c] & 0x80) {
      const int s = L.label[c], np = cnt[s];
      emit = np >= 30 || (np >= 5 && __popc(rws[s]) >= 3);
    } else if (L.ground[c] == 1) {
      is_ground = true;
      emit = !(j % 5 != 0 && j > 5 && j < kSgCols - 5);
    }
    if (emit) {
      out_cloud[ob + pos] = fl[c], out_range[ob + pos] = rm[c], out_col[ob + pos] = (unsigned)j;
      out_ground[ob + pos] = is_ground ? 1 : 0;
      ++pos;
    }
  }
  if (tid == 0) L.ring_count[kSgRows] = total;
  __syncthreads();
  if (tid < kSgRows) {
    fo->start_ring[tid] = L.ring_count[tid] - 1 + 5;
    fo->end_ring[tid] = L.ring_count[tid + 1] - 1 - 5;
  }
#ifdef LINS_SG_PROF
  if (tid == 0 && scan == 0) { long long t_ = clock64(); printf("SG %d %lld sweeps %d\n", 6, t_ - sg_t0, sg_sweeps); }
#endif
  if (tid == 0) {
    fo->n = total;
    // findStartEndAngle (IP:191-203), including the y(last) / x(second-to-last) mix
    float so = -lins_atan2f(pts[0].y, pts[0].x);
    float eo = (float)((double)(-lins_atan2f(pts[n - 1].y, pts[n - 2].x)) + 2 * kPi);
    if (eo - so > 3 * kPi)
      eo = (float)((double)eo - 2 * kPi);
    else if (eo - so < kPi)
      eo = (float)((double)eo + 2 * kPi);
    fo->start_ori = so, fo->end_ori = eo, fo->ori_diff = eo - so;
    out_outliers[scan] = L.n_outlier;
  }
}

void launch_segment(hipStream_t stream, int n_scans, const void* raws, const float4* raw, float sin_ax, float cos_ax,
                    float sin_ay, float cos_ay, float theta, int* cellidx, float4* full, float* rangeMat, int* seg_count,
                    int* seg_rows, void* fe_scans, float4* out_cloud, float* out_range, unsigned* out_col,
                    unsigned char* out_ground, int* out_outliers) {
  SgConsts k{sin_ax, cos_ax, sin_ay, cos_ay, theta};
  hipLaunchKernelGGL(segment_kernel, dim3(n_scans), dim3(kSgBlock), 0, stream, (const SgRaw*)raws, raw, k, cellidx, full,
                     rangeMat, seg_count, seg_rows, (unsigned char*)fe_scans, out_cloud, out_range, out_col, out_ground,
                     out_outliers);
}
size_t sg_raw_size() { return sizeof(SgRaw); }

}  // namespace lins
