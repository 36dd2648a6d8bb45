// synth.cpp — seeded synthetic VLP-16 scan pairs + canned IMU.
//
// There is no dataset in the reference tree (the sample rosbag is an external
// link, README.md:55) and no network, so the workload is manufactured: a
// parametric scene is ray-cast on the 16 x 1800 VLP-16 firing grid from a moving
// sensor, the raw (motion-distorted) clouds go through the restated front-end
// (frontend.cpp) to become the four feature clouds performIESKF() reads, and 40
// IMU samples at 400 Hz go through the restated StatePredictor
// (state_predictor.cpp) to become its prior (x, P).
//
// Scene: ground z=-1.8 m, 40 x 30 m room with 6 m walls, 24 poles r=0.15 m,
// 8 boxes 2x2x2 m.  Sensor: rings -15..+15 deg step 2 deg (parameters.h:82-84),
// 1800 firings of 0.2 deg, clockwise, range noise N(0, 0.02 m), max 100 m.  Like a real sensor's, the firing
// azimuths are generic: firing f sits at (f + c_f) * 0.2 deg with c_f = phase + 0.1 (f / 1800 - 1/2) + e_f — a seeded
// per-scan phase in (-0.3, 0.3) of a column, a slow drift of a tenth of a column per revolution (a rotor turning 56 ppm
// slow) and a per-firing jitter e_f ~ N(0, 0.004) clamped to +-0.01.  So |c_f| < 0.36: no firing sits within 0.14
// column (0.028 deg) of a rounding edge of image_projection_node's column index (IP:225), and the firings a quarter,
// half and three-quarter turn after the first one — which sit exactly on the reference's unwrap thresholds
// (SE:632-645: start - pi/2, start + pi, start + 3 pi/2, end - 3 pi/2, end + pi/2) when the spacing is exactly 0.2 deg
// — are >= 0.005 column (1.7e-5 rad) off them, two orders above any atan2f's error.  (Rounds 1-2 fired at exactly
// (f + 0.5) * 0.2 deg: every point on a column edge, where the column hangs on the last bit of whichever atan2f is used.)
// Motion: planar, constant forward speed U(0,10) m/s and yaw rate U(-0.5,0.5).
//
// Scene family B (round 5, lins_synth_*_scene(1, ...): "open"): the opposite of the room in everything the search
// structures and the feature front-end could have been tuned to — no room: open ground to the 100 m range limit (the
// upper rings see sky); ~60 trunks (vertical cylinders of radius 0.15 .. 0.45 m) scattered over 80 x 80 m; six far wall
// segments (10 .. 25 m long, 5 m high, 35 .. 70 m away, any orientation); 30 % of the firings that hit something return
// nothing (seeded per ray); and one 2 x 2 x 2 m box that MOVES at up to 5 m/s — every ray sees it where it is at its
// own firing time, so its points disagree with the rigid motion the filter estimates.  Same sensor, motion and prior.

#include <array>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../../include/lins_host.h"
#include "../lins_math.h"

using namespace lins;

namespace {

struct Rng {  // splitmix64
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
  double uni(double a, double b) { return a + (b - a) * uni(); }
  double gauss() {
    double u1 = uni(), u2 = uni();
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
  }
};

constexpr double kGroundZ = -1.8, kWallTop = 4.2, kHalfX = 20.0, kHalfY = 15.0;
constexpr double kPoleR = 0.15, kBoxHalf = 1.0, kMaxRange = 100.0, kMinRange = 0.5;
constexpr double kScanPeriod = 0.1;
constexpr int kPoles = 24, kBoxes = 8, kImuPerScan = 40;

struct Scene {
  double pole[kPoles][2];
  double box[kBoxes][2];
  double x0, y0, yaw0, speed, yaw_rate;
  // family B ("open"): no room, these instead
  bool open = false;
  double drop = 0.0;                       // probability that a ray that hit something returns nothing
  std::vector<std::array<double, 3>> trunk;  // x, y, radius
  std::vector<std::array<double, 4>> wall;   // segment x0, y0, x1, y1 (height kFarWallTop)
  double mbox[4] = {0, 0, 0, 0};           // moving box: position at tau = 0, velocity
};
constexpr double kFarWallTop = 3.2;  // (5 m above the ground)

Scene make_scene(uint32_t seed, uint32_t scan_index) {
  Rng r(((uint64_t)seed << 32) ^ (uint64_t)scan_index * 0x9E3779B1u ^ 0xA5A5A5A5ull);
  Scene s;
  s.x0 = r.uni(-5, 5);
  s.y0 = r.uni(-4, 4);
  s.yaw0 = r.uni(-M_PI, M_PI);
  s.speed = r.uni(0, 10);
  s.yaw_rate = r.uni(-0.5, 0.5);
  auto place = [&](double* xy, double margin) {
    for (;;) {
      double x = r.uni(-kHalfX + margin, kHalfX - margin), y = r.uni(-kHalfY + margin, kHalfY - margin);
      double dx = x - s.x0, dy = y - s.y0;
      if (dx * dx + dy * dy < 4.5 * 4.5) continue;  // keep the sensor's 0.2 s path clear
      xy[0] = x, xy[1] = y;
      return;
    }
  };
  for (auto& p : s.pole) place(p, 1.0);
  for (auto& b : s.box) place(b, 2.0);
  return s;
}

Scene make_open_scene(uint32_t seed, uint32_t scan_index) {
  Rng r(((uint64_t)seed << 32) ^ (uint64_t)scan_index * 0x9E3779B1u ^ 0x0B0B0B0B0Bull);
  Scene s;
  s.open = true, s.drop = 0.30;
  s.x0 = r.uni(-5, 5), s.y0 = r.uni(-4, 4), s.yaw0 = r.uni(-M_PI, M_PI);
  s.speed = r.uni(0, 10), s.yaw_rate = r.uni(-0.5, 0.5);
  for (auto& p : s.pole) p[0] = p[1] = 1e9;  // (unused: far outside the range limit)
  for (auto& b : s.box) b[0] = b[1] = 1e9;
  for (int k = 0; k < 60; ++k)
    for (;;) {
      const double x = s.x0 + r.uni(-40, 40), y = s.y0 + r.uni(-40, 40), rad = r.uni(0.15, 0.45);
      const double dx = x - s.x0, dy = y - s.y0;
      if (dx * dx + dy * dy < 4.5 * 4.5) continue;  // keep the sensor's 0.2 s path clear
      s.trunk.push_back({x, y, rad});
      break;
    }
  for (int k = 0; k < 6; ++k) {
    const double dist = r.uni(35, 70), bearing = r.uni(-M_PI, M_PI), len = r.uni(10, 25), dir = r.uni(-M_PI, M_PI);
    const double cx = s.x0 + dist * std::cos(bearing), cy = s.y0 + dist * std::sin(bearing);
    s.wall.push_back({cx - 0.5 * len * std::cos(dir), cy - 0.5 * len * std::sin(dir), cx + 0.5 * len * std::cos(dir), cy + 0.5 * len * std::sin(dir)});
  }
  for (;;) {  // the moving box: 6 .. 15 m away at tau = 0, any heading, 1 .. 5 m/s; never within 3 m of the sensor's start
    const double dist = r.uni(6, 15), bearing = r.uni(-M_PI, M_PI), v = r.uni(1, 5), h = r.uni(-M_PI, M_PI);
    s.mbox[0] = s.x0 + dist * std::cos(bearing), s.mbox[1] = s.y0 + dist * std::sin(bearing);
    s.mbox[2] = v * std::cos(h), s.mbox[3] = v * std::sin(h);
    const double ex = s.mbox[0] + 0.2 * s.mbox[2] - s.x0, ey = s.mbox[1] + 0.2 * s.mbox[3] - s.y0;
    if (ex * ex + ey * ey > 5.5 * 5.5) break;
  }
  return s;
}

Scene make_scene_of(int family, uint32_t seed, uint32_t scan_index) { return family == 1 ? make_open_scene(seed, scan_index) : make_scene(seed, scan_index); }

// planar pose of the sensor at absolute time tau (since the start of scan 0)
void pose_at(const Scene& s, double tau, double& x, double& y, double& yaw) {
  double w = s.yaw_rate, v = s.speed;
  yaw = s.yaw0 + w * tau;
  double lx, ly;  // displacement in the frame of tau = 0
  if (std::fabs(w) < 1e-9) {
    lx = v * tau, ly = 0;
  } else {
    lx = v / w * std::sin(w * tau), ly = v / w * (1 - std::cos(w * tau));
  }
  x = s.x0 + std::cos(s.yaw0) * lx - std::sin(s.yaw0) * ly;
  y = s.y0 + std::sin(s.yaw0) * lx + std::cos(s.yaw0) * ly;
}

// an axis-aligned 2 x 2 x 2 m box standing on the ground at (bx, by): entry distance of the ray, or +inf
double cast_box(double bx, double by, const double o[3], const double d[3]) {
  double lo[3] = {bx - kBoxHalf, by - kBoxHalf, kGroundZ};
  double hi[3] = {bx + kBoxHalf, by + kBoxHalf, kGroundZ + 2 * kBoxHalf};
  double t0 = 0, t1 = INFINITY;
  for (int k = 0; k < 3; ++k) {
    if (std::fabs(d[k]) < 1e-12) {
      if (o[k] < lo[k] || o[k] > hi[k]) return INFINITY;
    } else {
      double ta = (lo[k] - o[k]) / d[k], tb = (hi[k] - o[k]) / d[k];
      if (ta > tb) std::swap(ta, tb);
      t0 = std::max(t0, ta), t1 = std::min(t1, tb);
      if (t0 > t1) return INFINITY;
    }
  }
  return t0;
}

// nearest hit distance of the ray o + r d (|d| = 1) fired at time tau, or +inf
double cast(const Scene& s, const double o[3], const double d[3], double tau = 0.0) {
  double best = INFINITY;
  auto consider = [&](double t) {
    if (t > 1e-6 && t < best) best = t;
  };
  if (d[2] < -1e-12) consider((kGroundZ - o[2]) / d[2]);
  if (s.open) {  // family B: trunks, far wall segments, the moving box
    const double a2 = d[0] * d[0] + d[1] * d[1];
    if (a2 > 1e-12)
      for (auto& p : s.trunk) {
        const double fx = o[0] - p[0], fy = o[1] - p[1];
        const double b = fx * d[0] + fy * d[1], c = fx * fx + fy * fy - p[2] * p[2];
        const double disc = b * b - a2 * c;
        if (disc < 0) continue;
        const double t = (-b - std::sqrt(disc)) / a2;
        if (t <= 0) continue;
        const double z = o[2] + t * d[2];
        if (z >= kGroundZ && z <= kWallTop) consider(t);
      }
    for (auto& w : s.wall) {  // ray against the vertical rectangle over the segment
      const double ex = w[2] - w[0], ey = w[3] - w[1];
      const double den = d[0] * ey - d[1] * ex;
      if (std::fabs(den) < 1e-12) continue;
      const double fx = w[0] - o[0], fy = w[1] - o[1];
      const double t = (fx * ey - fy * ex) / den, u = (fx * d[1] - fy * d[0]) / den;
      if (t <= 0 || u < 0 || u > 1) continue;
      const double z = o[2] + t * d[2];
      if (z >= kGroundZ && z <= kFarWallTop) consider(t);
    }
    consider(cast_box(s.mbox[0] + tau * s.mbox[2], s.mbox[1] + tau * s.mbox[3], o, d));
    return best;
  }
  // walls (bounded in the other horizontal axis and in z)
  for (int sgn = -1; sgn <= 1; sgn += 2) {
    if (std::fabs(d[0]) > 1e-12) {
      double t = (sgn * kHalfX - o[0]) / d[0];
      if (t > 0) {
        double y = o[1] + t * d[1], z = o[2] + t * d[2];
        if (std::fabs(y) <= kHalfY && z >= kGroundZ && z <= kWallTop) consider(t);
      }
    }
    if (std::fabs(d[1]) > 1e-12) {
      double t = (sgn * kHalfY - o[1]) / d[1];
      if (t > 0) {
        double x = o[0] + t * d[0], z = o[2] + t * d[2];
        if (std::fabs(x) <= kHalfX && z >= kGroundZ && z <= kWallTop) consider(t);
      }
    }
  }
  // poles: vertical cylinders
  double a = d[0] * d[0] + d[1] * d[1];
  if (a > 1e-12)
    for (auto& p : s.pole) {
      double fx = o[0] - p[0], fy = o[1] - p[1];
      double b = fx * d[0] + fy * d[1], c = fx * fx + fy * fy - kPoleR * kPoleR;
      double disc = b * b - a * c;
      if (disc < 0) continue;
      double t = (-b - std::sqrt(disc)) / a;
      if (t <= 0) continue;
      double z = o[2] + t * d[2];
      if (z >= kGroundZ && z <= kWallTop) consider(t);
    }
  // boxes: axis-aligned, on the ground
  for (auto& bx : s.box) {
    double lo[3] = {bx[0] - kBoxHalf, bx[1] - kBoxHalf, kGroundZ};
    double hi[3] = {bx[0] + kBoxHalf, bx[1] + kBoxHalf, kGroundZ + 2 * kBoxHalf};
    double t0 = 0, t1 = INFINITY;
    bool miss = false;
    for (int k = 0; k < 3 && !miss; ++k) {
      if (std::fabs(d[k]) < 1e-12) {
        if (o[k] < lo[k] || o[k] > hi[k]) miss = true;
      } else {
        double ta = (lo[k] - o[k]) / d[k], tb = (hi[k] - o[k]) / d[k];
        if (ta > tb) std::swap(ta, tb);
        t0 = std::max(t0, ta), t1 = std::min(t1, tb);
        if (t0 > t1) miss = true;
      }
    }
    if (!miss) consider(t0);
  }
  return best;
}

// raw distorted cloud of scan k (k-th 0.1 s interval), firing order
int raw_scan(const Scene& s, uint32_t seed, uint32_t scan_index, int k, lins_point* out, int cap) {
  Rng noise(((uint64_t)seed << 32) ^ ((uint64_t)scan_index << 8) ^ (uint64_t)(k + 1) * 0xD1B54A32D192ED03ull);
  Rng azr(((uint64_t)seed << 32) ^ ((uint64_t)scan_index << 9) ^ (uint64_t)(k + 1) * 0x9FB21C651E98DF25ull);
  const double phase = azr.uni(-0.3, 0.3);  // of a column, per scan
  int n = 0;
  for (int f = 0; f < LINS_SCAN_NUM; ++f) {
    double tau = k * kScanPeriod + kScanPeriod * f / LINS_SCAN_NUM;
    double sx, sy, yaw;
    pose_at(s, tau, sx, sy, yaw);
    double jit = 0.004 * azr.gauss();
    jit = jit > 0.01 ? 0.01 : (jit < -0.01 ? -0.01 : jit);
    jit += 0.1 * ((double)f / LINS_SCAN_NUM - 0.5);
    // clockwise from -x; IP:225 rounds (f + phase + jit) - 900: the firing lands in column f (mod wrap), off the edges
    double az = M_PI - (f + phase + jit) * (2 * M_PI / LINS_SCAN_NUM);
    for (int l = 0; l < LINS_LINE_NUM; ++l) {
      double el = (-15.0 + 2.0 * l) * M_PI / 180.0;
      double ds[3] = {std::cos(el) * std::cos(az), std::cos(el) * std::sin(az), std::sin(el)};
      double dw[3] = {std::cos(yaw) * ds[0] - std::sin(yaw) * ds[1], std::sin(yaw) * ds[0] + std::cos(yaw) * ds[1], ds[2]};
      double o[3] = {sx, sy, 0.0};
      double r = cast(s, o, dw, tau);
      double nz = noise.gauss();  // always drawn: keeps streams aligned across hits/misses
      const bool lost = s.drop > 0 && noise.uni() < s.drop;  // (family B only: the room's streams are what they were)
      if (!(r < kMaxRange) || r < kMinRange || lost) continue;
      r += 0.02 * nz;
      if (n >= cap) return -1;
      out[n++] = {(float)(r * ds[0]), (float)(r * ds[1]), (float)(r * ds[2]), 0.f};
    }
  }
  return n;
}

// ---- a SEQUENCE of scans along one trajectory (the in-situ test of the drop-in boundary: tests/test_gpu_sequence.py) ----
// The same scene model; the sensor drives a circle of radius 3 .. 5 m around the scene's start point at 1 .. 3 m/s, so
// that any number of consecutive 0.1 s sweeps stays inside the room; poles and boxes keep 2.5 m off the circle.
constexpr uint32_t kSeqTag = 0x5E9u;
Scene make_seq_scene(uint32_t seed) {
  Rng r(((uint64_t)seed << 32) ^ 0x5E95E95E9ull);
  Scene s;
  s.x0 = r.uni(-4, 4), s.y0 = r.uni(-3, 3), s.yaw0 = r.uni(-M_PI, M_PI);
  s.speed = r.uni(1, 3);
  const double radius = r.uni(3, 5);
  s.yaw_rate = (r.uni() < 0.5 ? -1.0 : 1.0) * s.speed / radius;
  // centre of the circle: to the left (yaw_rate > 0) or right of the start heading
  const double sg = s.yaw_rate > 0 ? 1.0 : -1.0;
  const double cx = s.x0 - sg * radius * std::sin(s.yaw0), cy = s.y0 + sg * radius * std::cos(s.yaw0);
  auto place = [&](double* xy, double margin) {
    for (;;) {
      double x = r.uni(-kHalfX + margin, kHalfX - margin), y = r.uni(-kHalfY + margin, kHalfY - margin);
      double d = std::sqrt((x - cx) * (x - cx) + (y - cy) * (y - cy));
      if (std::fabs(d - radius) < 2.5 + margin) continue;  // keep the path clear
      xy[0] = x, xy[1] = y;
      return;
    }
  };
  for (auto& p : s.pole) place(p, 0.5);
  for (auto& b : s.box) place(b, 1.5);
  return s;
}

struct FeatBuf {
  std::vector<lins_point> cs, cls, sf, slf;
  lins_features f;
  FeatBuf() : cs(192), cls(1920), sf(LINS_MAX_QUERY), slf(LINS_CLOUD_MAX) {
    f.corner_sharp = cs.data(), f.corner_less_sharp = cls.data();
    f.surf_flat = sf.data(), f.surf_less_flat = slf.data();
    f.n_corner_sharp = f.n_corner_less_sharp = f.n_surf_flat = f.n_surf_less_flat = 0;
    f.n_segmented = f.n_outlier = 0;
  }
};

// relative pose over one scan interval: end frame expressed in the start frame
void rel_pose(const Scene& s, double t[3], double q[4]) {
  double w = s.yaw_rate, v = s.speed, T = kScanPeriod;
  double psi = w * T;
  if (std::fabs(w) < 1e-9) {
    t[0] = v * T, t[1] = 0;
  } else {
    t[0] = v / w * std::sin(psi), t[1] = v / w * (1 - std::cos(psi));
  }
  t[2] = 0;
  q[0] = std::cos(psi / 2), q[1] = 0, q[2] = 0, q[3] = std::sin(psi / 2);
}

}  // namespace

extern "C" {

int lins_synth_raw_scan_scene(int scene, uint32_t seed, uint32_t scan_index, int k, lins_point* out, int cap) {
  if (!out || k < 0 || scene < 0 || scene > 1) return LINS_E_ARG;
  Scene s = make_scene_of(scene, seed, scan_index);
  int n = raw_scan(s, seed, scan_index, k, out, cap);
  return n < 0 ? LINS_E_CAPACITY : n;
}
int lins_synth_raw_scan(uint32_t seed, uint32_t scan_index, int k, lins_point* out, int cap) {
  return lins_synth_raw_scan_scene(0, seed, scan_index, k, out, cap);
}

/* scan k (k = 0, 1, ...: the k-th 0.1 s sweep) of sequence `seed`: raw distorted cloud, firing order */
int lins_synth_seq_raw_scan(uint32_t seed, int k, lins_point* out, int cap) {
  if (!out || k < 0) return LINS_E_ARG;
  Scene s = make_seq_scene(seed);
  int n = raw_scan(s, seed, kSeqTag, k, out, cap);
  return n < 0 ? LINS_E_CAPACITY : n;
}

/* the 40 IMU samples (400 Hz) of sweep k: specific force and angular rate of the planar constant-twist motion in the
 * sensor frame + constant biases (INIT_BA / INIT_BW + a seeded offset) + white noise; acc, gyr: 40 x 3 */
int lins_synth_seq_imu(uint32_t seed, int k, double* acc, double* gyr) {
  if (!acc || !gyr || k < 0) return LINS_E_ARG;
  Scene s = make_seq_scene(seed);
  Rng rb(((uint64_t)seed << 32) ^ 0xB1A5ull);
  const double ba0[3] = {-0.015774, 0.143237, -0.0263845}, bw0[3] = {-0.00275058, -0.000165954, 0.00262913};
  double ba[3], bw[3];
  for (int i = 0; i < 3; ++i) ba[i] = ba0[i] + 0.01 * rb.gauss(), bw[i] = bw0[i] + 0.0005 * rb.gauss();
  Rng r(((uint64_t)seed << 32) ^ ((uint64_t)(k + 1) * 0x9E3779B97F4A7C15ull));
  for (int i = 0; i < kImuPerScan; ++i) {
    acc[i * 3 + 0] = ba[0] + 0.05 * r.gauss(), acc[i * 3 + 1] = s.speed * s.yaw_rate + ba[1] + 0.05 * r.gauss();
    acc[i * 3 + 2] = 9.81 + ba[2] + 0.05 * r.gauss();
    gyr[i * 3 + 0] = bw[0] + 0.002 * r.gauss(), gyr[i * 3 + 1] = bw[1] + 0.002 * r.gauss();
    gyr[i * 3 + 2] = s.yaw_rate + bw[2] + 0.002 * r.gauss();
  }
  return kImuPerScan;
}

/* ground truth: planar pose (x, y, yaw) of the sensor at time tau since the start of sweep 0, speed, yaw rate */
int lins_synth_seq_truth(uint32_t seed, double tau, double* xyyaw, double* speed, double* yaw_rate) {
  if (!xyyaw) return LINS_E_ARG;
  Scene s = make_seq_scene(seed);
  pose_at(s, tau, xyyaw[0], xyyaw[1], xyyaw[2]);
  if (speed) *speed = s.speed;
  if (yaw_rate) *yaw_rate = s.yaw_rate;
  return LINS_OK;
}

int lins_synth_generate(uint32_t seed, uint32_t scan_index, lins_synth_pair* out) { return lins_synth_generate_scene(0, seed, scan_index, out); }

int lins_synth_generate_scene(int scene, uint32_t seed, uint32_t scan_index, lins_synth_pair* out) {
  if (!out || !out->surf_flat || !out->corner_sharp || !out->surf_last || !out->corner_last || scene < 0 || scene > 1)
    return LINS_E_ARG;
  Scene s = make_scene_of(scene, seed, scan_index);
  std::vector<lins_point> raw(LINS_CLOUD_MAX);
  FeatBuf last, cur;
  int n0 = raw_scan(s, seed, scan_index, 0, raw.data(), LINS_CLOUD_MAX);
  if (n0 < 2) return LINS_E_INPUT;
  int rc = lins_frontend_extract(raw.data(), n0, kScanPeriod, &last.f);
  if (rc) return rc;
  int n1 = raw_scan(s, seed, scan_index, 1, raw.data(), LINS_CLOUD_MAX);
  if (n1 < 2) return LINS_E_INPUT;
  rc = lins_frontend_extract(raw.data(), n1, kScanPeriod, &cur.f);
  if (rc) return rc;
  out->n_raw_last = n0, out->n_raw_new = n1;

  // targets: previous scan's less-sharp / less-flat clouds re-projected to its end
  // with the true motion — what updatePointCloud() leaves after a converged update
  double t[3], q[4];
  rel_pose(s, t, q);
  lins_transform_to_end(t, q, kScanPeriod, last.f.corner_less_sharp, last.f.n_corner_less_sharp, out->corner_last);
  lins_transform_to_end(t, q, kScanPeriod, last.f.surf_less_flat, last.f.n_surf_less_flat, out->surf_last);
  out->n_corner_last = last.f.n_corner_less_sharp;
  out->n_surf_last = last.f.n_surf_less_flat;
  std::memcpy(out->corner_sharp, cur.f.corner_sharp, sizeof(lins_point) * cur.f.n_corner_sharp);
  std::memcpy(out->surf_flat, cur.f.surf_flat, sizeof(lins_point) * cur.f.n_surf_flat);
  out->n_corner_sharp = cur.f.n_corner_sharp;
  out->n_surf_flat = cur.f.n_surf_flat;
  std::memcpy(out->true_t, t, sizeof t);
  std::memcpy(out->true_q, q, sizeof q);
  out->speed = s.speed, out->yaw_rate = s.yaw_rate;

  // prior: StatePredictor over scan 0's IMU, reset(1), then scan 1's 40 samples
  Rng r(((uint64_t)seed << 32) ^ (uint64_t)scan_index * 0x85EBCA6Bu ^ 0x5EEDull);
  const double ba0[3] = {-0.015774, 0.143237, -0.0263845};       // INIT_BA yaml:65-69
  const double bw0[3] = {-0.00275058, -0.000165954, 0.00262913}; // INIT_BW yaml:71-75
  double ba_true[3], bw_true[3];
  for (int i = 0; i < 3; ++i) ba_true[i] = ba0[i] + 0.01 * r.gauss(), bw_true[i] = bw0[i] + 0.0005 * r.gauss();
  lins_filter_params fp;
  lins_filter_default_params(&fp);
  lins_filter filt;
  double v_body[3] = {s.speed + 0.05 * r.gauss(), 0.05 * r.gauss(), 0.02 * r.gauss()};
  lins_filter_init(&filt, &fp, v_body, ba0, bw0);
  const double dt = kScanPeriod / kImuPerScan;
  for (int scan = 0; scan < 2; ++scan) {
    for (int i = 0; i < kImuPerScan; ++i) {
      // specific force / angular rate of the planar constant-twist motion
      double acc[3] = {0 + ba_true[0] + 0.05 * r.gauss(), s.speed * s.yaw_rate + ba_true[1] + 0.05 * r.gauss(),
                       9.81 + ba_true[2] + 0.05 * r.gauss()};
      double gyr[3] = {bw_true[0] + 0.002 * r.gauss(), bw_true[1] + 0.002 * r.gauss(),
                       s.yaw_rate + bw_true[2] + 0.002 * r.gauss()};
      lins_filter_predict(&filt, dt, acc, gyr);
    }
    if (scan == 0) lins_filter_reset1(&filt);
  }
  std::memcpy(out->state, filt.state, sizeof filt.state);
  std::memcpy(out->cov, filt.cov, sizeof filt.cov);
  return LINS_OK;
}

}  // extern "C"
