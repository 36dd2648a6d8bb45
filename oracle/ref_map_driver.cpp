// ref_map_driver.cpp — the REFERENCE'S OWN lidar_mapping_node.cpp compiled verbatim into oracle/_ref/liblins_ref.so: its
// scan-to-map optimisation (cornerOptimization, surfOptimization, LMOptimization, scan2MapOptimization, LM:1351-1652,
// with pointAssociateToMap LM:579-607) run on caller-provided map and scan clouds.
//
// TEST INFRASTRUCTURE ONLY (oracle/): loaded by tests/ through oracle/ref.py; never by the product.
//
// The node is one class, MappingHandler, around ros::NodeHandle, GTSAM, tf and PCL; with the stand-in headers of
// oracle/ref_shim all of it parses and the part run here executes: the kd-tree is the exact 5-NN stand-in, the three
// OpenCV calls (cv::eigen, cv::solve(DECOMP_QR), Mat products / inv) run the restated numerics of
// lins_ref_shim/cv_restated.h — OpenCV itself is not on this machine, so what this pins is everything the reference
// itself wrote: the association, the line / plane coefficient formulas and their gates, the 6 x 6 normal equations'
// rows, the degeneracy projection, the update and the stop rule.  The class keeps its working clouds private; the
// driver reaches them by compiling the node's text with `private` spelled `public` (after every standard and stand-in
// header has been included under its own guards) — no line of the node is changed; its main() is renamed, never called.
#include <parameters.h>

#include <gtsam/lins_ref_gtsam.h>

#include <cstring>
#include <deque>
#include <iostream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/lins_map.h"

#define private public
#define main lins_ref_lidar_mapping_node_main
#include <lidar_mapping_node.cpp>
#undef main
#undef private

namespace {

void fill(pcl::PointCloud<PointType>::Ptr& c, const lins_point* p, int n) {
  c->points.resize(n);
  for (int i = 0; i < n; ++i) c->points[i].x = p[i].x, c->points[i].y = p[i].y, c->points[i].z = p[i].z, c->points[i].intensity = p[i].intensity;
  c->width = (std::uint32_t)n, c->height = 1;
}

void load(MappingHandler& h, const lins_map_problem* in) {
  fill(h.laserCloudCornerFromMapDS, in->map_corner, in->n_map_corner);
  fill(h.laserCloudSurfFromMapDS, in->map_surf, in->n_map_surf);
  fill(h.laserCloudCornerLastDS, in->scan_corner, in->n_scan_corner);
  fill(h.laserCloudSurfTotalLastDS, in->scan_surf, in->n_scan_surf);
  h.laserCloudCornerFromMapDSNum = in->n_map_corner, h.laserCloudSurfFromMapDSNum = in->n_map_surf;
  h.laserCloudCornerLastDSNum = in->n_scan_corner, h.laserCloudSurfTotalLastDSNum = in->n_scan_surf;
  for (int i = 0; i < 6; ++i) h.transformTobeMapped[i] = in->transform[i];
}

}  // namespace

extern "C" {

// one pass of cornerOptimization + surfOptimization at in->transform: the rows they push (laserCloudOri / coeffSel, corner
// rows first), at most cap of them.  -> number of rows
int ref_map_rows(const lins_map_problem* in, lins_point* ori, float* coeff4, int cap) {
  if (!in || !ori || !coeff4) return -1;
  parameter::LINE_NUM = 16, parameter::SCAN_NUM = 1800;
  ros::NodeHandle nh, pnh("~");
  MappingHandler h(nh, pnh);
  load(h, in);
  h.kdtreeCornerFromMap->setInputCloud(h.laserCloudCornerFromMapDS);
  h.kdtreeSurfFromMap->setInputCloud(h.laserCloudSurfFromMapDS);
  h.laserCloudOri->clear(), h.coeffSel->clear();
  h.cornerOptimization(0);
  h.surfOptimization(0);
  const int n = (int)h.laserCloudOri->points.size();
  for (int i = 0; i < n && i < cap; ++i) {
    const PointType &o = h.laserCloudOri->points[i], &c = h.coeffSel->points[i];
    ori[i].x = o.x, ori[i].y = o.y, ori[i].z = o.z, ori[i].intensity = o.intensity;
    coeff4[4 * i] = c.x, coeff4[4 * i + 1] = c.y, coeff4[4 * i + 2] = c.z, coeff4[4 * i + 3] = c.intensity;
  }
  return n;
}

// scan2MapOptimization.  The transform comes from the node's own scan2MapOptimization(); the counters (rounds run,
// converged, rows of the last round, degenerate) from a second object driven round by round through the same three
// member functions — whose transform must equal the first's (-9 otherwise).
int ref_scan2map(const lins_map_problem* in, lins_map_result* out) {
  if (!in || !out) return -1;
  parameter::LINE_NUM = 16, parameter::SCAN_NUM = 1800;
  ros::NodeHandle nh, pnh("~");
  MappingHandler whole(nh, pnh), steps(nh, pnh);
  load(whole, in), load(steps, in);
  whole.scan2MapOptimization();
  std::memset(out, 0, sizeof *out);
  if (steps.laserCloudCornerFromMapDSNum > 10 && steps.laserCloudSurfFromMapDSNum > 100) {  // (LM:1636)
    steps.kdtreeCornerFromMap->setInputCloud(steps.laserCloudCornerFromMapDS);
    steps.kdtreeSurfFromMap->setInputCloud(steps.laserCloudSurfFromMapDS);
    for (int iter = 0; iter < 10; ++iter) {
      steps.laserCloudOri->clear(), steps.coeffSel->clear();
      steps.cornerOptimization(iter);
      steps.surfOptimization(iter);
      out->n_sel = (int)steps.laserCloudOri->points.size();
      out->iters = iter + 1;
      if (steps.LMOptimization(iter)) {
        out->converged = 1;
        break;
      }
    }
  }
  out->degenerate = steps.isDegenerate ? 1 : 0;
  for (int i = 0; i < 6; ++i) out->transform[i] = whole.transformTobeMapped[i];
  for (int i = 0; i < 6; ++i)
    if (std::memcmp(&whole.transformTobeMapped[i], &steps.transformTobeMapped[i], sizeof(float)) != 0) return -9;
  return 0;
}

}  // extern "C"
