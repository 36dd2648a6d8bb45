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
  unsigned char flags[kSgCells];  // bit 0 right, bit 1 "+255", bit 2 down edge; bit 3 groundMat == 1; bit 7 eligible (labelMat == 0)
  union {
    unsigned own[kSgCells];  // 1 + index of the point that owns the cell (0: no return) — during the projection
    float range[kSgCells];   // rangeMat — until the adjacency is built
    struct {
      unsigned short label[kSgCells];  // smallest raster index that reaches the cell (0xFFFF: not eligible)
      unsigned cnt2[kSgCells / 2];     // cells per label, two u16 counters per word (a count never exceeds 28 800)
    } seg;                             // — afterwards
    unsigned short emitted[kSgCells];  // the emitted cells in output order — once validity is decided
  } u;
  int changed;
  int scan_tmp[20];
  int ring_count[kSgRows + 1];
  int n_outlier;
};
static_assert(sizeof(SgLds) <= 160 * 1024 && kSgCells % 2 == 0, "LDS budget");
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

// The images live in LDS, in the same 115 KB one after the other: the owner image (the projection's atomicMax
// target), rangeMat, then the labels + their counters, then the list of emitted cells.  The only per-scan global
// scratch is a copy of the owner image (written once, read back by ground removal and the emission) and the row
// masks of the small segments (zeroed at the segment roots only).  fullCloud is never materialised: a cell's point is read
// back through its index where the reference reads fullCloud (ground removal, emission).
__global__ __launch_bounds__(kSgBlock) void segment_kernel(const SgRaw* __restrict__ raws, const float4* __restrict__ raw,
                                                           SgConsts k, unsigned* __restrict__ cellidx, int* __restrict__ seg_rows,
                                                           unsigned char* __restrict__ fe_scans, float4* __restrict__ out_cloud,
                                                           float* __restrict__ out_range, unsigned* __restrict__ out_col,
                                                           unsigned char* __restrict__ out_ground, int* __restrict__ out_outliers) {
  SgLds& L = g_sg;
#ifdef LINS_SG_PROF
  long long sg_t0 = clock64();
  int sg_sweeps = 0;
  long long sg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SG_MARK(id) \
  { long long t_ = clock64(); sg_t[id] = t_ - sg_t0; sg_t0 = t_; }
#else
#define SG_MARK(id)
#endif
  const int tid = threadIdx.x, scan = blockIdx.x;
  const SgRaw rw = raws[scan];
  const float4* pts = raw + rw.off;
  const int n = rw.n;
  // per cell: 1 + index of the point that owns it, 0 = no return (written once, from the LDS image the projection builds)
  unsigned* ci = cellidx + (size_t)scan * kSgCells;
  int* rws = seg_rows + (size_t)scan * kSgCells;
  FeScanOut* fo = reinterpret_cast<FeScanOut*>(fe_scans + (size_t)scan * kFeScanStride);
  const double kPi = 3.14159265358979323846;

  for (int c = tid; c < kSgCells; c += kSgBlock) L.u.own[c] = 0u, L.flags[c] = 0;
  if (tid == 0) L.n_outlier = 0;
  __syncthreads();
  SG_MARK(0)

  // ---- projectPointCloud (IP:205-241): the LAST point that falls into a cell owns it -------------
  // Clouds of up to kSgPts points per thread (a VLP-16 scan has 28 800) keep every point's cell in registers, so
  // that the range image and, later, the output are written point by point with coalesced reads of the raw cloud
  // (the raw cloud is in firing order, the images ring-major: gathering points cell by cell touches a 64-byte line
  // per 16-byte point).  Larger clouds take the cell-by-cell path.
  constexpr int kSgPts = 32;
  const bool by_point = n <= kSgPts * kSgBlock;
  int cell_of[kSgPts];
  unsigned owner_mask = 0;  // bit k: this thread's k-th point owns its cell
  auto cell_of_point = [&](const float4& p) {
    const float vert = (float)((double)(lins_atan2f(p.z, sqrtf(p.x * p.x + p.y * p.y)) * 180) / kPi);
    const float rowf = (vert + (15.0f + 0.1f)) / 2.0f;
    // size_t rowIdn = rowf (IP:207, 220-221): the conversion truncates towards zero, so (-1, 0) is row 0; anything
    // <= -1 (or NaN) becomes a huge index on x86-64 and fails the `>= LINE_NUM` test
    if (!(rowf > -1.0f) || rowf >= kSgRows) return -1;
    const int row = (int)rowf;
    const float horizon = (float)((double)(lins_atan2f(p.x, p.y) * 180) / kPi);
    int colm = (int)(-round(((double)horizon - 90.0) / (double)0.2f) + kSgCols / 2);
    if (colm >= kSgCols) colm -= kSgCols;
    if (colm < 0 || colm >= kSgCols) return -1;
    return colm + row * kSgCols;
  };
  if (by_point) {
#pragma unroll
    for (int k = 0; k < kSgPts; ++k) {
      const int i = tid + k * kSgBlock;
      cell_of[k] = i < n ? cell_of_point(pts[i]) : -1;
      if (cell_of[k] >= 0) atomicMax(&L.u.own[cell_of[k]], (unsigned)(i + 1));
    }
  } else {
#pragma unroll 4
    for (int i = tid; i < n; i += kSgBlock) {
      const int cell = cell_of_point(pts[i]);
      if (cell >= 0) atomicMax(&L.u.own[cell], (unsigned)(i + 1));
    }
  }
  __syncthreads();
  // ---- groundRemoval (IP:243-278): one thread per column, rows bottom-up (a row is rewritten by the next) ----
  // (straight off the owner image while it is still in LDS: rounds 1-2 ran it after the range image had taken the
  // owner image's place and read the owners back from a global copy)
  for (int j = tid; j < kSgCols; j += kSgBlock) {
    int gi[kSgGroundScanInd + 1];
#pragma unroll
    for (int i = 0; i <= kSgGroundScanInd; ++i) gi[i] = (int)L.u.own[j + i * kSgCols] - 1;
    float4 gp[kSgGroundScanInd + 1];
#pragma unroll
    for (int i = 0; i <= kSgGroundScanInd; ++i) gp[i] = pts[gi[i] >= 0 ? gi[i] : 0];
    int g[kSgGroundScanInd + 1];
#pragma unroll
    for (int i = 0; i <= kSgGroundScanInd; ++i) g[i] = 0;
#pragma unroll
    for (int i = 0; i < kSgGroundScanInd; ++i) {
      if (gi[i] < 0 || gi[i + 1] < 0) {  // fullCloud intensity -1: no point in one of the two cells
        g[i] = -1;
        continue;
      }
      const float dx = gp[i + 1].x - gp[i].x, dy = gp[i + 1].y - gp[i].y, dz = gp[i + 1].z - gp[i].z;
      const float angle = (float)((double)(lins_atan2f(dz, sqrtf(dx * dx + dy * dy)) * 180) / kPi);
      if (fabsf(angle - 0.0f) <= 10) g[i] = 1, g[i + 1] = 1;
    }
#pragma unroll
    for (int i = 0; i <= kSgGroundScanInd; ++i)
      if (g[i] == 1) L.flags[j + i * kSgCols] = 8;
  }
  __syncthreads();  // (every reader of the owner image is done: its bytes may become the range image)


  // the owner image's LDS bytes become rangeMat (FLT_MAX: no return); only the cell-by-cell path of oversized clouds
  // keeps a global copy of it (its emission reads the owners back)
  if (by_point) {
    for (int c = tid; c < kSgCells; c += kSgBlock)
      if (!L.u.own[c]) L.u.range[c] = FLT_MAX;
    __syncthreads();
    // (a cell's word is rewritten by its owner only; the other points of that cell compare it with their own
    // index + 1 and see either the owner's index or range bits — a range >= 0.1 m is no index — never their own)
#pragma unroll
    for (int k = 0; k < kSgPts; ++k) {
      const int c = cell_of[k], i = tid + k * kSgBlock;
      if (c >= 0 && L.u.own[c] == (unsigned)(i + 1)) {
        const float4 p = pts[i];
        L.u.range[c] = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);
        owner_mask |= 1u << k;
      }
    }
  } else {
    constexpr int kFill = 8;
    for (int c0 = tid; c0 < kSgCells; c0 += kSgBlock * kFill) {
      unsigned o[kFill];
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const int c = c0 + u * kSgBlock;
        o[u] = c < kSgCells ? L.u.own[c] : 0u;
      }
      float4 p[kFill];
#pragma unroll
      for (int u = 0; u < kFill; ++u) p[u] = pts[o[u] ? o[u] - 1 : 0];
#pragma unroll
      for (int u = 0; u < kFill; ++u) {
        const int c = c0 + u * kSgBlock;
        if (c < kSgCells) {
          ci[c] = o[u];
          L.u.range[c] = o[u] ? sqrtf(p[u].x * p[u].x + p[u].y * p[u].y + p[u].z * p[u].z) : FLT_MAX;
        }
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  SG_MARK(1)
  SG_MARK(2)

  // ---- adjacency of labelComponents (IP:336-415) as three edge bits per eligible cell ------------------
  auto target = [&](int c, int dir) {  // dir 0: (0, +1)   1: (0, +255)   2: (+1, 0);  -1 if outside
    const int r = c / kSgCols, col = c - r * kSgCols;
    if (dir == 2) return r + 1 < kSgRows ? c + kSgCols : -1;
    int tc = col + (dir == 0 ? 1 : 255);
    if (tc >= kSgCols) tc = 0;
    return r * kSgCols + tc;
  };
  // eligible = labelMat 0 at the start: not ground, has a return (bit 3 is final here, the range image too)
  auto eligible = [&](int c) { return !((L.flags[c] & 8) || L.u.range[c] == FLT_MAX); };
#pragma unroll 2
  for (int c = tid; c < kSgCells; c += kSgBlock) {
    if (!eligible(c)) continue;
    unsigned char e = 0x80;
    const float rc = L.u.range[c];
#pragma unroll
    for (int dir = 0; dir < 3; ++dir) {
      const int t = target(c, dir);
      if (t < 0 || !eligible(t)) continue;
      const float rt = L.u.range[t];
      const float d1 = fmaxf(rc, rt), d2 = fminf(rc, rt);
      const float sa = dir == 2 ? k.sin_ay : k.sin_ax, ca = dir == 2 ? k.cos_ay : k.cos_ax;
      const float angle = lins_atan2f(d2 * sa, d1 - d2 * ca);
      if (angle > k.theta) e |= (unsigned char)(1 << dir);
    }
    L.flags[c] = e;  // (an eligible cell has bit 3 clear; others read only bit 3 of this byte, which stays 0)
  }
  __syncthreads();  // the range image is dead from here: its bytes become the labels and their counters
  for (int c = tid; c < kSgCells; c += kSgBlock) L.u.seg.label[c] = (L.flags[c] & 0x80) ? (unsigned short)c : (unsigned short)0xFFFF;
  for (int w = tid; w < kSgCells / 2; w += kSgBlock) L.u.seg.cnt2[w] = 0;
  __syncthreads();
  SG_MARK(3)

  // ---- min-label propagation: pull over the in-edges, own run of cells in raster order, until stable ----
  unsigned short* label = L.u.seg.label;
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
      if (r < kSgRows && (L.flags[c0] & 0x80)) {
        unsigned best = 0xFFFFu;
        for (int cc = kSgCols - 255 + lane; cc < kSgCols; cc += 64)
          if (L.flags[c0 + cc] & 2) best = min(best, (unsigned)label[c0 + cc]);
        for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
        if (lane == 0 && best < label[c0]) label[c0] = (unsigned short)best, ch = true;
      }
    }
    __syncthreads();
    for (int c = c_lo; c < c_hi; ++c) {
      if (!(L.flags[c] & 0x80)) continue;
      const int r = c / kSgCols, col = c - r * kSgCols;
      unsigned short best = label[c];
      // in-edges: (r, col - 1) right [col 0: (r, 1799)], (r, col - 255) "+255" [col >= 255], (r - 1, col) down
      const int pl = col ? c - 1 : c + kSgCols - 1;
      if (L.flags[pl] & 1) best = min(best, label[pl]);
      if (col >= 255 && (L.flags[c - 255] & 2)) best = min(best, label[c - 255]);
      if (r > 0 && (L.flags[c - kSgCols] & 4)) best = min(best, label[c - kSgCols]);
      if (best != label[c]) label[c] = best, ch = true;
    }
    if (ch) L.changed = 1;
    __syncthreads();
#ifdef LINS_SG_PROF
    ++sg_sweeps;
#endif
    if (!L.changed) break;
    __syncthreads();
  }
  SG_MARK(4)

  // ---- segment validity (IP:398-406): size, and rows of the cells that were pushed as neighbours -------
  auto count_of = [&](int s) { return (int)((L.u.seg.cnt2[s >> 1] >> ((s & 1) * 16)) & 0xFFFFu); };
  for (int c = tid; c < kSgCells; c += kSgBlock)
    if (L.flags[c] & 0x80) {
      const int s = label[c];
      atomicAdd(&L.u.seg.cnt2[s >> 1], 1u << ((s & 1) * 16));
      if (s == c) rws[c] = 0;  // a segment's row mask lives at its root: zeroed here, not for all 28 800 cells
    }
  __threadfence_block();
  __syncthreads();
  // the row test only decides segments of 5 .. 29 cells: only those touch the global row masks
  for (int c = tid; c < kSgCells; c += kSgBlock)
    if (L.flags[c] & 0x80) {
      const int s = label[c];
      if (c != s) {
        const int np = count_of(s);
        if (np >= 5 && np < 30) atomicOr(&rws[s], 1 << (c / kSgCols));
      }
    }
  __threadfence_block();
  __syncthreads();
  SG_MARK(5)

  // ---- cloudSegmentation emission (IP:292-321): ring-major, ascending column -----------------------------
  auto feasible = [&](int c) {
    const int s = label[c], np = count_of(s);
    return np >= 30 || (np >= 5 && __popc(rws[s]) >= 3);
  };
  int mine = 0;
  unsigned emit_mask = 0;  // bit k: cell c_lo + k is emitted (kRun <= 32)
  static_assert(kRun <= 32, "emit mask");
  for (int c = c_lo; c < c_hi; ++c) {
    const int i = c / kSgCols, j = c - i * kSgCols;
    bool emit = false;
    if (L.flags[c] & 0x80) {
      emit = feasible(c);
      if (!emit && i > kSgGroundScanInd && j % 5 == 0) atomicAdd(&L.n_outlier, 1);
    } else if (L.flags[c] & 8) {
      emit = !(j % 5 != 0 && j > 5 && j < kSgCols - 5);
    }
    if (emit) ++mine, emit_mask |= 1u << (c - c_lo);
  }
  const int base = sg_block_scan(mine, tid, L.scan_tmp);
  const int total = L.scan_tmp[18];
  const size_t ob = (size_t)fo->off;
  // (sg_block_scan's barriers: every thread is done with the labels and counters — their bytes take the list)
  int pos = base;
  for (int c = c_lo; c < c_hi; ++c) {
    if (c % kSgCols == 0) L.ring_count[c / kSgCols] = pos;  // points emitted before ring i
    if (emit_mask & (1u << (c - c_lo))) {
      if (by_point)
        L.u.emitted[c] = (unsigned short)pos++, L.flags[c] |= 0x40;  // cell -> output position, bit 6: emitted
      else
        L.u.emitted[pos++] = (unsigned short)c;                       // output position -> cell
    }
  }
  __syncthreads();
  auto put = [&](int o, int c, float4 p) {
    const int i = c / kSgCols, j = c - i * kSgCols;
    out_range[ob + o] = sqrtf(p.x * p.x + p.y * p.y + p.z * p.z);  // rangeMat's expression (IP:230)
    p.w = (float)((double)(float)i + (double)(float)j / 10000.0);  // fullCloud intensity (IP:234)
    out_cloud[ob + o] = p, out_col[ob + o] = (unsigned)j;
    out_ground[ob + o] = (L.flags[c] & 8) ? 1 : 0;
  };
  if (by_point) {  // every owner of an emitted cell writes its point: coalesced reads of the raw cloud
#pragma unroll
    for (int k = 0; k < kSgPts; ++k)
      if ((owner_mask >> k) & 1u) {
        const int c = cell_of[k];
        if (L.flags[c] & 0x40) put((int)L.u.emitted[c], c, pts[tid + k * kSgBlock]);
      }
  } else {  // one thread per emitted point: coalesced stores, gathers of the owning points
#pragma unroll 4
    for (int o = tid; o < total; o += kSgBlock) {
      const int c = L.u.emitted[o];
      put(o, c, pts[(int)ci[c] - 1]);
    }
  }
  if (tid == 0) L.ring_count[kSgRows] = total;
  __syncthreads();
  if (tid < kSgRows) {
    fo->start_ring[tid] = L.ring_count[tid] - 1 + 5;
    fo->end_ring[tid] = L.ring_count[tid + 1] - 1 - 5;
  }
#ifdef LINS_SG_PROF
  SG_MARK(6)
  if (tid == 0 && (scan & 255) == 0)
    printf("SG scan %d: init %lld project+range %lld ground %lld adjacency %lld labels %lld (%d sweeps) validity %lld emission %lld\n", scan,
           sg_t[0], sg_t[1], sg_t[2], sg_t[3], sg_t[4], sg_sweeps, sg_t[5], sg_t[6]);
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
                    float sin_ay, float cos_ay, float theta, unsigned* cellidx, int* seg_rows, void* fe_scans, float4* out_cloud,
                    float* out_range, unsigned* out_col, unsigned char* out_ground, int* out_outliers) {
  SgConsts k{sin_ax, cos_ax, sin_ay, cos_ay, theta};
  hipLaunchKernelGGL(segment_kernel, dim3(n_scans), dim3(kSgBlock), 0, stream, (const SgRaw*)raws, raw, k, cellidx, seg_rows,
                     (unsigned char*)fe_scans, out_cloud, out_range, out_col, out_ground, out_outliers);
}
size_t sg_raw_size() { return sizeof(SgRaw); }

}  // namespace lins
