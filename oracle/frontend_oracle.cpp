// frontend_oracle.cpp — CPU restatement of the stages either side of the IESKF update, for CHECKING the device
// kernels of those rows (SURVEY.md §8f-2, §8f-3 and the projection / segmentation stage before them).
//
// TEST INFRASTRUCTURE ONLY: loaded by tests/ and tools/ through ctypes, never linked by the product.
// PARITY: the feature stage (SE:619-827) and transformToEnd are pinned through oracle/_ref since round 3 (tests/test_ref.py holds
// the product's host restatement against the reference's own compiled undistortPcl ... extractFeatures; this libm-based
// checker is held against that restatement by tests/test_frontend_oracle.py).  The image projection / segmentation part
// (IP:174-415) is pinned the same way: oracle/ref_ip_driver.cpp compiles image_projection_node.cpp verbatim and tests/test_ref.py
// holds the host restatement against what the node publishes, bit for bit.
//
// Independence: this file includes NOTHING from csrc/ (in particular not csrc/lins_math.h, whose fixed-sequence
// lins_atan2f the device kernels and the product's host restatement share) and nothing from include/.  Every
// angle goes through this box's libm exactly where the reference calls it (atan2f / sinf / cosf on float
// arguments: with the reference's headers the float overloads of <cmath> are the ones its unqualified calls
// resolve to; sin / cos / atan2 on doubles in the quaternion helpers), so that what the device kernels are
// compared with is the arithmetic of the reference's own node on this machine.
//
// Restated (paths relative to /root/reference/lins/):
//   src/image_projection_node.cpp
//     findStartEndAngle 191-203 · projectPointCloud 205-241 · groundRemoval 243-287 · cloudSegmentation 289-334
//     labelComponents 336-415 (with the std::pair<uint8_t, uint8_t> neighbour table of lines 72, 133-144: the
//     stored -1 reads back as 255, so "up" never passes the row test and "left" is "255 columns to the right")
//   include/StateEstimator.hpp
//     undistortPcl 619-654 (rotatePoint with IMU_LIDAR_EXTRINSIC_ANGLE = 0, exp_port.yaml:7: the identity)
//     calculateSmoothness 656-678 · markOccludedPoints 680-713 · extractFeatures 719-827 · transformToEnd 1083-1101
//   include/parameters.h:82-92 (constants), config/exp_config/exp_port.yaml:9-13 (LINE_NUM, SCAN_NUM, thresholds)
// Third-party behaviour restated from its published algorithm (pcl 1.7 / 1.8, un-pinned by the reference):
//   pcl::removeNaNFromPointCloud (drop points with a non-finite coordinate)
//   pcl::VoxelGrid::applyFilter, leaf 0.2 (SE:189): f32 min / max box, min_b = floor(min * inv_leaf), index =
//     sum_k (floor(x_k * inv_leaf_k) - min_b_k) * mul_k, voxels in ascending index, centroid = f32 sums / n.
// Left open by the reference and FIXED here (stated, not pinned): std::sort's order of equal curvatures (here:
// by position), the order in which VoxelGrid's unstable sort leaves the points of one voxel (here: input order),
// what segmentedCloudColInd[-1] holds when the first sector's position 4 (a default Smooth: value 0, ind 0) gets
// picked and its backward neighbour loop steps off the array (here: a column gap, nothing marked),
// size_t conversion of a negative row (here: x86-64's cvttss2si — (-1, 0) truncates to row 0, anything <= -1
// becomes a huge index and is dropped, IP:220-221), `abs` of a float angle (here: the float overload).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

const int LINE_NUM = 16, SCAN_NUM = 1800;
const float ang_res_x = 0.2, ang_res_y = 2.0, ang_bottom = 15.0 + 0.1;
const int groundScanInd = 5;
const float sensorMountAngle = 0.0;
const float segmentTheta = 1.0472;
const int segmentValidPointNum = 5, segmentValidLineNum = 3;
const float segmentAlphaX = ang_res_x / 180.0 * M_PI;
const float segmentAlphaY = ang_res_y / 180.0 * M_PI;
const double EDGE_THRESHOLD = 0.5, SURF_THRESHOLD = 0.5;

struct P4 {
  float x, y, z, intensity;
};

}  // namespace

extern "C" {

// ---- image_projection_node: raw cloud (firing order) -> segmented cloud + cloud_info ----------------------
// Outputs sized for LINE_NUM * SCAN_NUM entries; returns the segmented size.  label_out (optional, LINE_NUM *
// SCAN_NUM ints) receives labelMat for inspection.
int fo_segment(const float* raw_xyzi, int n_raw, float* seg_xyzi, float* seg_range, uint32_t* seg_col, uint8_t* seg_ground,
               int32_t* start_ring, int32_t* end_ring, float* orientation3, int32_t* n_outlier, float* outlier_xyzi,
               int32_t* label_out) {
  // copyPointCloud: removeNaNFromPointCloud
  std::vector<P4> in;
  in.reserve(n_raw);
  for (int i = 0; i < n_raw; ++i) {
    const float* p = raw_xyzi + 4 * (size_t)i;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
    in.push_back(P4{p[0], p[1], p[2], p[3]});
  }
  if (in.size() < 2) return -1;
  // findStartEndAngle (the message fields are float32)
  float startOrientation = -atan2f(in[0].y, in[0].x);
  float endOrientation = (float)((double)(-atan2f(in[in.size() - 1].y, in[in.size() - 2].x)) + 2 * M_PI);
  if (endOrientation - startOrientation > 3 * M_PI) {
    endOrientation = (float)((double)endOrientation - 2 * M_PI);
  } else if (endOrientation - startOrientation < M_PI) {
    endOrientation = (float)((double)endOrientation + 2 * M_PI);
  }
  const float orientationDiff = endOrientation - startOrientation;
  orientation3[0] = startOrientation, orientation3[1] = endOrientation, orientation3[2] = orientationDiff;

  // resetParameters
  std::vector<float> rangeMat((size_t)LINE_NUM * SCAN_NUM, FLT_MAX);
  std::vector<int8_t> groundMat((size_t)LINE_NUM * SCAN_NUM, 0);
  std::vector<int32_t> labelMat((size_t)LINE_NUM * SCAN_NUM, 0);
  const float nan = std::numeric_limits<float>::quiet_NaN();
  std::vector<P4> fullCloud((size_t)LINE_NUM * SCAN_NUM, P4{nan, nan, nan, -1.f});

  // projectPointCloud
  for (size_t i = 0; i < in.size(); ++i) {
    P4 thisPoint{in[i].x, in[i].y, in[i].z, 0.f};
    const float verticalAngle =
        (float)((double)(atan2f(thisPoint.z, sqrtf(thisPoint.x * thisPoint.x + thisPoint.y * thisPoint.y)) * 180) / M_PI);
    const float rowf = (verticalAngle + ang_bottom) / ang_res_y;
    // size_t rowIdn = rowf (IP:220): cvttss2si — truncation towards zero for rowf > -1, "indefinite" (huge) otherwise
    if (!(rowf > -1.0f) || !(rowf < 9.0e18f)) continue;
    const size_t rowIdn = (size_t)(int64_t)rowf;
    if (rowIdn >= (size_t)LINE_NUM) continue;
    const float horizonAngle = (float)((double)(atan2f(thisPoint.x, thisPoint.y) * 180) / M_PI);
    const double colv = -round((horizonAngle - 90.0) / ang_res_x) + SCAN_NUM / 2;
    if (!(colv > -1.0)) continue;  // (cannot happen for finite angles: colv is in [450, 2250])
    size_t columnIdn = (size_t)(int64_t)colv;
    if (columnIdn >= (size_t)SCAN_NUM) columnIdn -= SCAN_NUM;
    if (columnIdn >= (size_t)SCAN_NUM) continue;
    const float range = sqrtf(thisPoint.x * thisPoint.x + thisPoint.y * thisPoint.y + thisPoint.z * thisPoint.z);
    rangeMat[rowIdn * SCAN_NUM + columnIdn] = range;
    thisPoint.intensity = (float)((float)rowIdn + (float)columnIdn / 10000.0);
    fullCloud[columnIdn + rowIdn * SCAN_NUM] = thisPoint;
  }

  // groundRemoval
  for (size_t j = 0; j < (size_t)SCAN_NUM; ++j) {
    for (size_t i = 0; i < (size_t)groundScanInd; ++i) {
      const size_t lowerInd = j + i * SCAN_NUM, upperInd = j + (i + 1) * SCAN_NUM;
      if (fullCloud[lowerInd].intensity == -1 || fullCloud[upperInd].intensity == -1) {
        groundMat[i * SCAN_NUM + j] = -1;
        continue;
      }
      const float diffX = fullCloud[upperInd].x - fullCloud[lowerInd].x;
      const float diffY = fullCloud[upperInd].y - fullCloud[lowerInd].y;
      const float diffZ = fullCloud[upperInd].z - fullCloud[lowerInd].z;
      const float angle = (float)((double)(atan2f(diffZ, sqrtf(diffX * diffX + diffY * diffY)) * 180) / M_PI);
      if (std::abs(angle - sensorMountAngle) <= 10) {
        groundMat[i * SCAN_NUM + j] = 1;
        groundMat[(i + 1) * SCAN_NUM + j] = 1;
      }
    }
  }
  for (size_t k = 0; k < (size_t)LINE_NUM * SCAN_NUM; ++k)
    if (groundMat[k] == 1 || rangeMat[k] == FLT_MAX) labelMat[k] = -1;

  // cloudSegmentation / labelComponents
  int labelCount = 1;
  const uint8_t nb[4][2] = {{(uint8_t)-1, 0}, {0, 1}, {0, (uint8_t)-1}, {1, 0}};  // pair<uint8_t, uint8_t> of IP:72
  std::vector<uint16_t> queueX((size_t)LINE_NUM * SCAN_NUM), queueY((size_t)LINE_NUM * SCAN_NUM);
  std::vector<uint16_t> pushedX((size_t)LINE_NUM * SCAN_NUM), pushedY((size_t)LINE_NUM * SCAN_NUM);
  for (int row = 0; row < LINE_NUM; ++row) {
    for (int col = 0; col < SCAN_NUM; ++col) {
      if (labelMat[(size_t)row * SCAN_NUM + col] != 0) continue;
      bool lineCountFlag[16] = {false};
      queueX[0] = row, queueY[0] = col;
      int queueSize = 1, queueStartInd = 0, queueEndInd = 1;
      pushedX[0] = row, pushedY[0] = col;
      int allPushedIndSize = 1;
      while (queueSize > 0) {
        const int fromIndX = queueX[queueStartInd], fromIndY = queueY[queueStartInd];
        --queueSize;
        ++queueStartInd;
        labelMat[(size_t)fromIndX * SCAN_NUM + fromIndY] = labelCount;
        for (int it = 0; it < 4; ++it) {
          const int thisIndX = fromIndX + nb[it][0];
          int thisIndY = fromIndY + nb[it][1];
          if (thisIndX < 0 || thisIndX >= LINE_NUM) continue;
          if (thisIndY < 0) thisIndY = SCAN_NUM - 1;
          if (thisIndY >= SCAN_NUM) thisIndY = 0;
          if (labelMat[(size_t)thisIndX * SCAN_NUM + thisIndY] != 0) continue;
          const float ra = rangeMat[(size_t)fromIndX * SCAN_NUM + fromIndY], rb = rangeMat[(size_t)thisIndX * SCAN_NUM + thisIndY];
          const float d1 = std::max(ra, rb), d2 = std::min(ra, rb);
          const float alpha = nb[it][0] == 0 ? segmentAlphaX : segmentAlphaY;
          const float angle = atan2f(d2 * sinf(alpha), (d1 - d2 * cosf(alpha)));
          if (angle > segmentTheta) {
            queueX[queueEndInd] = thisIndX, queueY[queueEndInd] = thisIndY;
            ++queueSize;
            ++queueEndInd;
            labelMat[(size_t)thisIndX * SCAN_NUM + thisIndY] = labelCount;
            lineCountFlag[thisIndX] = true;
            pushedX[allPushedIndSize] = thisIndX, pushedY[allPushedIndSize] = thisIndY;
            ++allPushedIndSize;
          }
        }
      }
      bool feasibleSegment = false;
      if (allPushedIndSize >= 30) {
        feasibleSegment = true;
      } else if (allPushedIndSize >= segmentValidPointNum) {
        int lineCount = 0;
        for (int i = 0; i < LINE_NUM; ++i)
          if (lineCountFlag[i]) ++lineCount;
        if (lineCount >= segmentValidLineNum) feasibleSegment = true;
      }
      if (feasibleSegment) {
        ++labelCount;
      } else {
        for (int i = 0; i < allPushedIndSize; ++i) labelMat[(size_t)pushedX[i] * SCAN_NUM + pushedY[i]] = 999999;
      }
    }
  }
  if (label_out) std::memcpy(label_out, labelMat.data(), labelMat.size() * sizeof(int32_t));

  int sizeOfSegCloud = 0, outliers = 0;
  for (size_t i = 0; i < (size_t)LINE_NUM; ++i) {
    start_ring[i] = sizeOfSegCloud - 1 + 5;
    for (size_t j = 0; j < (size_t)SCAN_NUM; ++j) {
      const int32_t lab = labelMat[i * SCAN_NUM + j];
      const bool ground = groundMat[i * SCAN_NUM + j] == 1;
      if (lab > 0 || ground) {
        if (lab == 999999) {
          if (i > (size_t)groundScanInd && j % 5 == 0) {
            if (outlier_xyzi) std::memcpy(outlier_xyzi + 4 * (size_t)outliers, &fullCloud[j + i * SCAN_NUM], 16);
            ++outliers;
          }
          continue;
        }
        if (ground) {
          if (j % 5 != 0 && j > 5 && j < (size_t)SCAN_NUM - 5) continue;
        }
        seg_ground[sizeOfSegCloud] = ground ? 1 : 0;
        seg_col[sizeOfSegCloud] = (uint32_t)j;
        seg_range[sizeOfSegCloud] = rangeMat[i * SCAN_NUM + j];
        std::memcpy(seg_xyzi + 4 * (size_t)sizeOfSegCloud, &fullCloud[j + i * SCAN_NUM], 16);
        ++sizeOfSegCloud;
      }
    }
    end_ring[i] = sizeOfSegCloud - 1 - 5;
  }
  *n_outlier = outliers;
  return sizeOfSegCloud;
}

// ---- pcl::VoxelGrid (leaf 0.2, all fields, no minimum count) on one ring's candidates --------------------
static void voxel_grid(const std::vector<P4>& in, std::vector<P4>& out) {
  out.clear();
  if (in.empty()) return;
  const float leaf = 0.2f;
  const float inv = 1.0f / leaf;  // Eigen::Array4f::Ones() / leaf_size_
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (const P4& p : in) {  // getMinMax3D: finite points only
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    mn[0] = std::min(mn[0], p.x), mn[1] = std::min(mn[1], p.y), mn[2] = std::min(mn[2], p.z);
    mx[0] = std::max(mx[0], p.x), mx[1] = std::max(mx[1], p.y), mx[2] = std::max(mx[2], p.z);
  }
  int min_b[3], max_b[3], div_b[3], mul[3];
  for (int k = 0; k < 3; ++k) {
    min_b[k] = (int)floorf(mn[k] * inv), max_b[k] = (int)floorf(mx[k] * inv);
    div_b[k] = max_b[k] - min_b[k] + 1;
  }
  mul[0] = 1, mul[1] = div_b[0], mul[2] = div_b[0] * div_b[1];
  struct Entry {
    int idx;
    unsigned pos;
  };
  std::vector<Entry> idx;
  idx.reserve(in.size());
  for (unsigned i = 0; i < in.size(); ++i) {
    const P4& p = in[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    const int i0 = (int)(floorf(p.x * inv) - (float)min_b[0]);
    const int i1 = (int)(floorf(p.y * inv) - (float)min_b[1]);
    const int i2 = (int)(floorf(p.z * inv) - (float)min_b[2]);
    idx.push_back(Entry{i0 * mul[0] + i1 * mul[1] + i2 * mul[2], i});
  }
  std::stable_sort(idx.begin(), idx.end(), [](const Entry& a, const Entry& b) { return a.idx < b.idx; });
  size_t first = 0;
  while (first < idx.size()) {
    size_t last = first + 1;
    while (last < idx.size() && idx[last].idx == idx[first].idx) ++last;
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    for (size_t k = first; k < last; ++k) {
      const P4& p = in[idx[k].pos];
      sx += p.x, sy += p.y, sz += p.z, si += p.intensity;
    }
    const float n = (float)(last - first);
    out.push_back(P4{sx / n, sy / n, sz / n, si / n});
    first = last;
  }
}

// ---- StateEstimator's feature stage on a segmented scan -------------------------------------------------------
// Outputs: undistorted cloud (n points, time-tagged), the four feature clouds (caps: 192 / 1920 / 1024 / LINE_NUM *
// SCAN_NUM points) and their sizes in counts[4] = sharp, less sharp, flat, less flat.  Returns 0.
int fo_features(const float* seg_xyzi, const float* seg_range, const uint32_t* seg_col, const uint8_t* seg_ground, int n,
                const int32_t* start_ring, const int32_t* end_ring, const float* orientation3, double scan_period,
                float* undist_xyzi, float* sharp, float* less_sharp, float* flat, float* less_flat, int32_t* counts) {
  const float startOrientation = orientation3[0], endOrientation = orientation3[1], orientationDiff = orientation3[2];
  std::vector<P4> und(n);
  // undistortPcl
  bool halfPassed = false;
  for (int i = 0; i < n; ++i) {
    const float* d = seg_xyzi + 4 * (size_t)i;
    P4 point{d[0], d[1], d[2], d[3]};  // rotatePoint with a zero extrinsic angle: R = I in double, exact
    double ori = -atan2f(point.y, point.x);
    if (!halfPassed) {
      if (ori < startOrientation - M_PI / 2)
        ori += 2 * M_PI;
      else if (ori > startOrientation + M_PI * 3 / 2)
        ori -= 2 * M_PI;
      if (ori - startOrientation > M_PI) halfPassed = true;
    } else {
      ori += 2 * M_PI;
      if (ori < endOrientation - M_PI * 3 / 2)
        ori += 2 * M_PI;
      else if (ori > endOrientation + M_PI / 2)
        ori -= 2 * M_PI;
    }
    const double relTime = (ori - startOrientation) / orientationDiff;
    point.intensity = (float)(int(d[3]) + scan_period * relTime);
    und[i] = point;
  }
  std::memcpy(undist_xyzi, und.data(), (size_t)n * 16);

  // calculateSmoothness (the sum is a float expression, SE:660-670)
  const size_t cap = std::max((size_t)n, (size_t)LINE_NUM * SCAN_NUM);
  std::vector<double> curvature(cap, 0.0);
  std::vector<int> picked(cap, 0), label(cap, 0);
  struct Smooth {
    double value;
    size_t ind;
  };
  std::vector<Smooth> smooth(cap, Smooth{0.0, 0});
  const float* r = seg_range;
  for (int i = 5; i < n - 5; ++i) {
    const double diffRange = r[i - 5] + r[i - 4] + r[i - 3] + r[i - 2] + r[i - 1] - r[i] * 10 + r[i + 1] + r[i + 2] + r[i + 3] +
                             r[i + 4] + r[i + 5];
    curvature[i] = diffRange * diffRange;
    picked[i] = 0, label[i] = 0;
    smooth[i].value = curvature[i], smooth[i].ind = i;
  }
  // markOccludedPoints
  for (int i = 5; i < n - 6; ++i) {
    const float depth1 = r[i], depth2 = r[i + 1];
    const int columnDiff = std::abs(int(seg_col[i + 1] - seg_col[i]));
    if (columnDiff < 10) {
      if (depth1 - depth2 > 0.3) {
        for (int k = -5; k <= 0; ++k) picked[i + k] = 1;
      } else if (depth2 - depth1 > 0.3) {
        for (int k = 1; k <= 6; ++k) picked[i + k] = 1;
      }
    }
    const float diff1 = std::abs(r[i - 1] - r[i]), diff2 = std::abs(r[i + 1] - r[i]);
    if (diff1 > 0.02 * r[i] && diff2 > 0.02 * r[i]) picked[i] = 1;
  }
  // extractFeatures
  int n_sharp = 0, n_less_sharp = 0, n_flat = 0, n_less_flat = 0;
  auto put = [](float* dst, int& cnt, int cap_pts, const P4& p) {
    if (cnt < cap_pts) std::memcpy(dst + 4 * (size_t)cnt, &p, 16);
    ++cnt;
  };
  auto col_gap = [&](int a, int b) { return std::abs(int(seg_col[a] - seg_col[b])); };
  std::vector<P4> ringCand, ringDS;
  for (int i = 0; i < LINE_NUM; ++i) {
    ringCand.clear();
    for (int j = 0; j < 6; ++j) {
      const int sp = (start_ring[i] * (6 - j) + end_ring[i] * j) / 6;
      const int ep = (start_ring[i] * (5 - j) + end_ring[i] * (j + 1)) / 6 - 1;
      if (sp >= ep) continue;
      std::stable_sort(smooth.begin() + sp, smooth.begin() + ep, [](const Smooth& a, const Smooth& b) { return a.value < b.value; });
      int largestPickedNum = 0;
      for (int k = ep; k >= sp; --k) {
        const int ind = (int)smooth[k].ind;
        if (picked[ind] == 0 && curvature[ind] > EDGE_THRESHOLD && seg_ground[ind] == 0) {
          ++largestPickedNum;
          if (largestPickedNum <= 2) {
            label[ind] = 2;
            put(sharp, n_sharp, 192, und[ind]);
            put(less_sharp, n_less_sharp, 1920, und[ind]);
          } else if (largestPickedNum <= 20) {
            label[ind] = 1;
            put(less_sharp, n_less_sharp, 1920, und[ind]);
          } else {
            break;
          }
          picked[ind] = 1;
          for (int l = 1; l <= 5; ++l) {
            if (col_gap(ind + l, ind + l - 1) > 10) break;
            picked[ind + l] = 1;
          }
          for (int l = -1; l >= -5; --l) {
            if (ind + l < 0) break;  // (the reference reads segmentedCloudColInd[-1] here: whatever precedes the vector, a gap)
            if (col_gap(ind + l, ind + l + 1) > 10) break;
            picked[ind + l] = 1;
          }
        }
      }
      int smallestPickedNum = 0;
      for (int k = sp; k <= ep; ++k) {
        const int ind = (int)smooth[k].ind;
        if (picked[ind] == 0 && curvature[ind] < SURF_THRESHOLD && seg_ground[ind] != 0) {
          label[ind] = -1;
          put(flat, n_flat, 1024, und[ind]);
          ++smallestPickedNum;
          if (smallestPickedNum >= 4) break;
          picked[ind] = 1;
          for (int l = 1; l <= 5; ++l) {
            if (col_gap(ind + l, ind + l - 1) > 10) break;
            picked[ind + l] = 1;
          }
          for (int l = -1; l >= -5; --l) {
            if (ind + l < 0) break;  // (the reference reads segmentedCloudColInd[-1] here: whatever precedes the vector, a gap)
            if (col_gap(ind + l, ind + l + 1) > 10) break;
            picked[ind + l] = 1;
          }
        }
      }
      for (int k = sp; k <= ep; ++k)
        if (label[k] <= 0) ringCand.push_back(und[k]);
    }
    voxel_grid(ringCand, ringDS);
    for (const P4& p : ringDS) put(less_flat, n_less_flat, LINE_NUM * SCAN_NUM, p);
  }
  counts[0] = n_sharp, counts[1] = n_less_sharp, counts[2] = n_flat, counts[3] = n_less_flat;
  return 0;
}

// ---- transformToEnd (SE:1083-1101) for n points; t = linState_.rn_, q = linState_.qbn_ (w, x, y, z) ----------
void fo_transform_to_end(const double* t, const double* q, double scan_period, const float* in, int n, float* out) {
  auto qmul_vec = [](const double* qq, const double* v, double* o) {  // Eigen: v + 2w(u x v) + 2 u x (u x v)
    const double ux = qq[1], uy = qq[2], uz = qq[3];
    double cx = uy * v[2] - uz * v[1], cy = uz * v[0] - ux * v[2], cz = ux * v[1] - uy * v[0];
    cx += cx, cy += cy, cz += cz;
    o[0] = v[0] + qq[0] * cx + (uy * cz - uz * cy);
    o[1] = v[1] + qq[0] * cy + (uz * cx - ux * cz);
    o[2] = v[2] + qq[0] * cz + (ux * cy - uy * cx);
  };
  // Quat2axis(linState_.qbn_) (math_utils.h:75-88)
  double phi[3] = {q[1], q[2], q[3]};
  const double mag = sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (mag >= 1e-10) {
    double ang = 2.0 * atan2(mag, q[0]);
    while (ang >= M_PI) ang -= 2.0 * M_PI;
    while (ang < -M_PI) ang += 2.0 * M_PI;
    for (int k = 0; k < 3; ++k) phi[k] = phi[k] / mag * ang;
  }
  const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double qinv[4] = {q[0] / n2, -q[1] / n2, -q[2] / n2, -q[3] / n2};  // Eigen::Quaternion::inverse
  for (int i = 0; i < n; ++i) {
    const float* pi = in + 4 * (size_t)i;
    const double s = (1.f / scan_period) * (pi[3] - int(pi[3]));  // SCAN_PERIOD is a double: float / double, then double * float
    const double p2[3] = {pi[0], pi[1], pi[2]};
    // axis2Quat(s * phi) (math_utils.h:61-73, 43-59)
    const double v[3] = {s * phi[0], s * phi[1], s * phi[2]};
    const double theta = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double r21[4] = {1, 0, 0, 0};
    if (!(theta < 1e-10)) {
      const double m = sin(theta / 2.0f);
      r21[0] = cos(theta / 2.0f), r21[1] = v[0] / theta * m, r21[2] = v[1] / theta * m, r21[3] = v[2] / theta * m;
    }
    double p1[3];
    qmul_vec(r21, p2, p1);
    for (int k = 0; k < 3; ++k) p1[k] += s * t[k];
    const double d[3] = {p1[0] - t[0], p1[1] - t[1], p1[2] - t[2]};
    double e[3];
    qmul_vec(qinv, d, e);
    float* po = out + 4 * (size_t)i;
    po[0] = (float)e[0], po[1] = (float)e[1], po[2] = (float)e[2], po[3] = pi[3];
  }
}

}  // extern "C"
