// frontend.cpp — host-side feature front-end: turns one raw VLP-16 cloud into
// the four feature clouds the IESKF update reads.
//
// Behavioural mirror (own data layout: flat arrays over the 16 x 1800 grid) of
//   image_projection_node   /root/reference/lins/src/image_projection_node.cpp
//       findStartEndAngle 191-203, projectPointCloud 205-241, groundRemoval
//       243-287, cloudSegmentation 289-334, labelComponents 336-415
//   StateEstimator          /root/reference/lins/include/StateEstimator.hpp
//       undistortPcl 619-654, calculateSmoothness 656-678, markOccludedPoints
//       680-713, extractFeatures 719-827 (+ pcl::VoxelGrid 0.2 m, SE:189,822-825)
//   transformToEnd          StateEstimator.hpp:1083-1101
// Constants: parameters.h:79-92, exp_port.yaml:9-13.
//
// This is the CPU restatement: it FEEDS the hot path with realistically shaped inputs and is
// what the device front-end (frontend_kernels.hip, lins_extract_features_batch) is checked against.

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../../include/lins_host.h"
#include "../lins_math.h"

using namespace lins;

namespace {

constexpr int kRows = LINS_LINE_NUM, kCols = LINS_SCAN_NUM, kCells = kRows * kCols;
constexpr float kAngResX = 0.2f, kAngResY = 2.0f, kAngBottom = 15.0f + 0.1f;
constexpr int kGroundScanInd = 5;
constexpr float kSensorMountAngle = 0.0f;
constexpr float kSegmentTheta = 1.0472f;
constexpr int kSegmentValidPointNum = 5, kSegmentValidLineNum = 3;
const float kSegmentAlphaX = kAngResX / 180.0 * M_PI, kSegmentAlphaY = kAngResY / 180.0 * M_PI;
constexpr double kEdgeThreshold = 0.5, kSurfThreshold = 0.5;
constexpr float kLeaf = 0.2f;

struct Segmented {
  std::vector<lins_point> cloud;     // ring-major, ascending column
  std::vector<uint8_t> ground;       // segmentedCloudGroundFlag
  std::vector<uint32_t> col;         // segmentedCloudColInd
  std::vector<float> range;          // segmentedCloudRange
  int start_ring[kRows], end_ring[kRows];
  float start_ori, end_ori, ori_diff;
  int n_outlier = 0;
};

void project_and_segment(const lins_point* raw, int n, Segmented& seg) {
  std::vector<lins_point> full(kCells, lins_point{NAN, NAN, NAN, -1.f});
  std::vector<float> range(kCells, FLT_MAX);
  std::vector<int8_t> ground(kCells, 0);
  std::vector<int> label(kCells, 0);

  // findStartEndAngle (IP:191-203) — including the y(last)/x(second-to-last) mix
  seg.start_ori = -lins_atan2f(raw[0].y, raw[0].x);
  seg.end_ori = -lins_atan2f(raw[n - 1].y, raw[n - 2].x) + 2 * M_PI;
  if (seg.end_ori - seg.start_ori > 3 * M_PI)
    seg.end_ori -= 2 * M_PI;
  else if (seg.end_ori - seg.start_ori < M_PI)
    seg.end_ori += 2 * M_PI;
  seg.ori_diff = seg.end_ori - seg.start_ori;

  // projectPointCloud (IP:205-241)
  for (int i = 0; i < n; ++i) {
    lins_point p = raw[i];
    float vert = lins_atan2f(p.z, std::sqrt(p.x * p.x + p.y * p.y)) * 180 / M_PI;
    float rowf = (vert + kAngBottom) / kAngResY;
    // size_t rowIdn = rowf (IP:207, 220-221): truncation towards zero, so (-1, 0) is row 0; <= -1 or NaN turns
    // into a huge index on x86-64 and is dropped by the `>= LINE_NUM` test
    if (!(rowf > -1.0f) || rowf >= kRows) continue;
    int row = (int)rowf;
    float horizon = lins_atan2f(p.x, p.y) * 180 / M_PI;
    int colm = (int)(-std::round((horizon - 90.0) / kAngResX) + kCols / 2);
    if (colm >= kCols) colm -= kCols;
    if (colm < 0 || colm >= kCols) continue;
    float r = std::sqrt(p.x * p.x + p.y * p.y + p.z * p.z);
    range[colm + row * kCols] = r;
    p.intensity = (float)row + (float)colm / 10000.0;
    full[colm + row * kCols] = p;
  }

  // groundRemoval (IP:243-278)
  for (int j = 0; j < kCols; ++j)
    for (int i = 0; i < kGroundScanInd; ++i) {
      int lo = j + i * kCols, up = j + (i + 1) * kCols;
      if (full[lo].intensity == -1 || full[up].intensity == -1) {
        ground[lo] = -1;
        continue;
      }
      float dx = full[up].x - full[lo].x, dy = full[up].y - full[lo].y, dz = full[up].z - full[lo].z;
      float angle = lins_atan2f(dz, std::sqrt(dx * dx + dy * dy)) * 180 / M_PI;
      if (std::fabs(angle - kSensorMountAngle) <= 10) ground[lo] = 1, ground[up] = 1;
    }
  for (int c = 0; c < kCells; ++c)
    if (ground[c] == 1 || range[c] == FLT_MAX) label[c] = -1;

  // cloudSegmentation / labelComponents (IP:289-415): BFS over the neighbour list
  int label_count = 1;
  std::vector<int> queue(kCells), pushed(kCells);
  // The reference stores its (-1,0),(0,1),(0,-1),(1,0) offsets in
  // std::pair<uint8_t,uint8_t> (IP:72,133-144), so -1 is 255: the "row above"
  // neighbour always falls out of range and the "left" neighbour is column+255
  // (or column 0 once that passes SCAN_NUM, IP:365-366).  Kept as is.
  const int nbr[4][2] = {{255, 0}, {0, 1}, {0, 255}, {1, 0}};
  for (int r0 = 0; r0 < kRows; ++r0)
    for (int c0 = 0; c0 < kCols; ++c0) {
      if (label[c0 + r0 * kCols] != 0) continue;
      bool line_seen[kRows] = {false};
      int qs = 0, qe = 0, np = 0;
      queue[qe++] = c0 + r0 * kCols;
      pushed[np++] = c0 + r0 * kCols;
      while (qs < qe) {
        int cell = queue[qs++];
        int fr = cell / kCols, fc = cell % kCols;
        label[cell] = label_count;
        for (auto& d : nbr) {
          int tr = fr + d[0], tc = fc + d[1];
          if (tr < 0 || tr >= kRows) continue;
          if (tc < 0) tc = kCols - 1;
          if (tc >= kCols) tc = 0;
          int tcell = tc + tr * kCols;
          if (label[tcell] != 0) continue;
          float d1 = std::max(range[cell], range[tcell]), d2 = std::min(range[cell], range[tcell]);
          float alpha = d[0] == 0 ? kSegmentAlphaX : kSegmentAlphaY;
          float angle = lins_atan2f(d2 * std::sin(alpha), d1 - d2 * std::cos(alpha));
          if (angle > kSegmentTheta) {
            queue[qe++] = tcell;
            label[tcell] = label_count;
            line_seen[tr] = true;
            pushed[np++] = tcell;
          }
        }
      }
      bool feasible = false;
      if (np >= 30)
        feasible = true;
      else if (np >= kSegmentValidPointNum) {
        int lines = 0;
        for (bool b : line_seen) lines += b;
        if (lines >= kSegmentValidLineNum) feasible = true;
      }
      if (feasible)
        ++label_count;
      else
        for (int k = 0; k < np; ++k) label[pushed[k]] = 999999;
    }

  // emission (IP:292-321)
  seg.cloud.clear(), seg.ground.clear(), seg.col.clear(), seg.range.clear();
  seg.n_outlier = 0;
  int count = 0;
  for (int i = 0; i < kRows; ++i) {
    seg.start_ring[i] = count - 1 + 5;
    for (int j = 0; j < kCols; ++j) {
      int c = j + i * kCols;
      if (label[c] > 0 || ground[c] == 1) {
        if (label[c] == 999999) {
          if (i > kGroundScanInd && j % 5 == 0) seg.n_outlier++;
          continue;
        }
        if (ground[c] == 1 && j % 5 != 0 && j > 5 && j < kCols - 5) continue;
        seg.ground.push_back(ground[c] == 1);
        seg.col.push_back(j);
        seg.range.push_back(range[c]);
        seg.cloud.push_back(full[c]);
        ++count;
      }
    }
    seg.end_ring[i] = count - 1 - 5;
  }
}

// pcl::VoxelGrid with leaf 0.2 and all-field averaging; output ordered by voxel index
void voxel_grid(const std::vector<lins_point>& in, std::vector<lins_point>& out) {
  out.clear();
  if (in.empty()) return;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (auto& p : in) {
    mn[0] = std::min(mn[0], p.x), mn[1] = std::min(mn[1], p.y), mn[2] = std::min(mn[2], p.z);
    mx[0] = std::max(mx[0], p.x), mx[1] = std::max(mx[1], p.y), mx[2] = std::max(mx[2], p.z);
  }
  const float inv = 1.0f / kLeaf;
  int minb[3], maxb[3];
  for (int a = 0; a < 3; ++a) {
    minb[a] = (int)std::floor(mn[a] * inv);
    maxb[a] = (int)std::floor(mx[a] * inv);
  }
  long long dx = maxb[0] - minb[0] + 1, dy = maxb[1] - minb[1] + 1;
  struct Key {
    long long idx;
    int pt;
  };
  std::vector<Key> keys(in.size());
  for (size_t i = 0; i < in.size(); ++i) {
    long long ix = (long long)std::floor(in[i].x * inv) - minb[0];
    long long iy = (long long)std::floor(in[i].y * inv) - minb[1];
    long long iz = (long long)std::floor(in[i].z * inv) - minb[2];
    keys[i] = {ix + iy * dx + iz * dx * dy, (int)i};
  }
  std::stable_sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.idx < b.idx; });
  size_t i = 0;
  while (i < keys.size()) {
    size_t j = i;
    float sx = 0, sy = 0, sz = 0, si = 0;
    while (j < keys.size() && keys[j].idx == keys[i].idx) {
      const lins_point& p = in[keys[j].pt];
      sx += p.x, sy += p.y, sz += p.z, si += p.intensity;
      ++j;
    }
    float n = (float)(j - i);
    out.push_back({sx / n, sy / n, sz / n, si / n});
    i = j;
  }
}

void extract(const Segmented& seg, double scan_period, lins_features* out) {
  const int n = (int)seg.cloud.size();
  // undistortPcl (SE:619-654): relative-time tagging (IMU_LIDAR_EXTRINSIC_ANGLE = 0)
  std::vector<lins_point> und(n);
  bool half_passed = false;
  for (int i = 0; i < n; ++i) {
    lins_point p = seg.cloud[i];
    double ori = -lins_atan2f(p.y, p.x);
    if (!half_passed) {
      if (ori < seg.start_ori - M_PI / 2)
        ori += 2 * M_PI;
      else if (ori > seg.start_ori + M_PI * 3 / 2)
        ori -= 2 * M_PI;
      if (ori - seg.start_ori > M_PI) half_passed = true;
    } else {
      ori += 2 * M_PI;
      if (ori < seg.end_ori - M_PI * 3 / 2)
        ori += 2 * M_PI;
      else if (ori > seg.end_ori + M_PI / 2)
        ori -= 2 * M_PI;
    }
    double rel = (ori - seg.start_ori) / seg.ori_diff;
    p.intensity = (float)((int)seg.cloud[i].intensity + scan_period * rel);
    und[i] = p;
  }

  // calculateSmoothness (SE:656-678)
  std::vector<double> curv(std::max(n, 1), 0.0);
  std::vector<int> picked(std::max(n, 1), 0), lab(std::max(n, 1), 0);
  struct Smooth {
    double value;
    int ind;
  };
  std::vector<Smooth> smooth(std::max(n, 1), Smooth{0.0, 0});
  const std::vector<float>& rg = seg.range;
  for (int i = 5; i < n - 5; ++i) {
    double d = rg[i - 5] + rg[i - 4] + rg[i - 3] + rg[i - 2] + rg[i - 1] - rg[i] * 10 + rg[i + 1] +
               rg[i + 2] + rg[i + 3] + rg[i + 4] + rg[i + 5];
    curv[i] = d * d;
    smooth[i] = {curv[i], i};
  }
  // markOccludedPoints (SE:680-713)
  for (int i = 5; i < n - 6; ++i) {
    float d1 = rg[i], d2 = rg[i + 1];
    int cd = std::abs((int)(seg.col[i + 1] - seg.col[i]));
    if (cd < 10) {
      if (d1 - d2 > 0.3) {
        for (int k = 0; k <= 5; ++k) picked[i - k] = 1;
      } else if (d2 - d1 > 0.3) {
        for (int k = 1; k <= 6; ++k) picked[i + k] = 1;
      }
    }
    float f1 = std::fabs(rg[i - 1] - rg[i]), f2 = std::fabs(rg[i + 1] - rg[i]);
    if (f1 > 0.02 * rg[i] && f2 > 0.02 * rg[i]) picked[i] = 1;
  }

  // extractFeatures (SE:719-827)
  out->n_corner_sharp = out->n_corner_less_sharp = out->n_surf_flat = out->n_surf_less_flat = 0;
  std::vector<lins_point> ring_less_flat, ring_ds;
  auto col_gap = [&](int a, int b) {
    if (a < 0 || b < 0 || a >= n || b >= n) return 1000;  // guard (reference reads unchecked)
    return std::abs((int)(seg.col[a] - seg.col[b]));
  };
  auto mark_nbrs = [&](int ind) {
    for (int l = 1; l <= 5; ++l) {
      if (col_gap(ind + l, ind + l - 1) > 10) break;
      picked[ind + l] = 1;
    }
    for (int l = -1; l >= -5; --l) {
      if (col_gap(ind + l, ind + l + 1) > 10) break;
      picked[ind + l] = 1;
    }
  };
  for (int i = 0; i < kRows; ++i) {
    ring_less_flat.clear();
    for (int j = 0; j < 6; ++j) {
      int sp = (seg.start_ring[i] * (6 - j) + seg.end_ring[i] * j) / 6;
      int ep = (seg.start_ring[i] * (5 - j) + seg.end_ring[i] * (j + 1)) / 6 - 1;
      if (sp >= ep) continue;
      if (sp < 0 || ep >= n) continue;  // guard
      // (the reference's comparator looks at the value only, which leaves equal curvatures in an
      // unspecified order; ties are broken by index here and in the device kernel)
      std::sort(smooth.begin() + sp, smooth.begin() + ep, [](const Smooth& a, const Smooth& b) {
        return a.value < b.value || (a.value == b.value && a.ind < b.ind);
      });
      int largest = 0;
      for (int k = ep; k >= sp; --k) {
        int ind = smooth[k].ind;
        if (picked[ind] == 0 && curv[ind] > kEdgeThreshold && !seg.ground[ind]) {
          ++largest;
          if (largest <= 2) {
            lab[ind] = 2;
            if (out->n_corner_sharp < 192) out->corner_sharp[out->n_corner_sharp++] = und[ind];
            if (out->n_corner_less_sharp < 1920) out->corner_less_sharp[out->n_corner_less_sharp++] = und[ind];
          } else if (largest <= 20) {
            lab[ind] = 1;
            if (out->n_corner_less_sharp < 1920) out->corner_less_sharp[out->n_corner_less_sharp++] = und[ind];
          } else {
            break;
          }
          picked[ind] = 1;
          mark_nbrs(ind);
        }
      }
      int smallest = 0;
      for (int k = sp; k <= ep; ++k) {
        int ind = smooth[k].ind;
        if (picked[ind] == 0 && curv[ind] < kSurfThreshold && seg.ground[ind]) {
          lab[ind] = -1;
          if (out->n_surf_flat < LINS_MAX_QUERY) out->surf_flat[out->n_surf_flat++] = und[ind];
          if (++smallest >= 4) break;
          picked[ind] = 1;
          mark_nbrs(ind);
        }
      }
      for (int k = sp; k <= ep; ++k)
        if (lab[k] <= 0) ring_less_flat.push_back(und[k]);
    }
    voxel_grid(ring_less_flat, ring_ds);
    for (auto& p : ring_ds)
      if (out->n_surf_less_flat < LINS_CLOUD_MAX) out->surf_less_flat[out->n_surf_less_flat++] = p;
  }
  out->n_segmented = n;
  out->n_outlier = seg.n_outlier;
}

}  // namespace

extern "C" {

int lins_frontend_extract(const lins_point* raw, int n_raw, double scan_period, lins_features* out) {
  if (!raw || !out || n_raw < 2) return LINS_E_ARG;
  if (!out->corner_sharp || !out->corner_less_sharp || !out->surf_flat || !out->surf_less_flat)
    return LINS_E_ARG;
  Segmented seg;
  project_and_segment(raw, n_raw, seg);
  extract(seg, scan_period, out);
  return LINS_OK;
}

int lins_frontend_segment(const lins_point* raw, int n_raw, lins_point* cloud, float* range, uint32_t* col,
                          uint8_t* ground, lins_segmented_scan* out) {
  if (!raw || !cloud || !range || !col || !ground || !out || n_raw < 2) return LINS_E_ARG;
  Segmented seg;
  project_and_segment(raw, n_raw, seg);
  const int n = (int)seg.cloud.size();
  if (n > LINS_CLOUD_MAX) return LINS_E_CAPACITY;
  for (int i = 0; i < n; ++i) cloud[i] = seg.cloud[i], range[i] = seg.range[i], col[i] = seg.col[i], ground[i] = seg.ground[i];
  out->cloud = cloud, out->range = range, out->col = col, out->ground = ground, out->n = n;
  for (int r = 0; r < kRows; ++r) out->start_ring[r] = seg.start_ring[r], out->end_ring[r] = seg.end_ring[r];
  out->start_ori = seg.start_ori, out->end_ori = seg.end_ori, out->ori_diff = seg.ori_diff;
  out->n_outlier = seg.n_outlier;
  return LINS_OK;
}

int lins_frontend_extract_segmented(const lins_segmented_scan* in, double scan_period, lins_features* out) {
  if (!in || !out || in->n < 0 || in->n > LINS_CLOUD_MAX) return LINS_E_ARG;
  if (!out->corner_sharp || !out->corner_less_sharp || !out->surf_flat || !out->surf_less_flat)
    return LINS_E_ARG;
  Segmented seg;
  seg.cloud.assign(in->cloud, in->cloud + in->n);
  seg.range.assign(in->range, in->range + in->n);
  seg.col.assign(in->col, in->col + in->n);
  seg.ground.assign(in->ground, in->ground + in->n);
  for (int r = 0; r < kRows; ++r) seg.start_ring[r] = in->start_ring[r], seg.end_ring[r] = in->end_ring[r];
  seg.start_ori = in->start_ori, seg.end_ori = in->end_ori, seg.ori_diff = in->ori_diff;
  seg.n_outlier = in->n_outlier;
  extract(seg, scan_period, out);
  return LINS_OK;
}

float lins_host_atan2f(float y, float x) { return lins_atan2f(y, x); }

void lins_transform_to_end(const double* t, const double* q, double scan_period, const lins_point* in,
                           int n, lins_point* out) {
  // SE:1083-1101: to start with the interpolated pose, then to end with the full pose
  V3 tt{t[0], t[1], t[2]};
  Q4 qq{q[0], q[1], q[2], q[3]};
  V3 phi = quat2axis(qq);
  Q4 qinv = qinverse(qq);
  for (int i = 0; i < n; ++i) {
    lins_point pi = in[i];
    float frac = pi.intensity - (float)(int)pi.intensity;
    double s = (double)(1.f / scan_period) * (double)frac;
    V3 p1 = qrot(axis2quat(s * phi), V3{pi.x, pi.y, pi.z}) + s * tt;
    V3 p2 = qrot(qinv, p1 - tt);
    out[i] = {(float)p2.x, (float)p2.y, (float)p2.z, pi.intensity};
  }
}

}  // extern "C"
