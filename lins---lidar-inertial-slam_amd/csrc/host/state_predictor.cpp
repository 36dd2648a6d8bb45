// state_predictor.cpp — host-side IMU propagation producing the prior (x, P)
// that the IESKF update starts from.
//
// Behavioural mirror of filter::StatePredictor, /root/reference/lins/include/
// KalmanFilter.hpp: predict 125-186, initializeCovariance 247-311, reset(1)
// 320-352; unit constants parameters.h:63-71.  400 Hz x 18x18 serial algebra —
// stays on the host by design.

#include <cmath>
#include <cstring>

#include "../../../include/lins_host.h"
#include "../lins_math.h"

using namespace lins;

namespace {
constexpr double kG0 = 9.81;
constexpr double kDeg = M_PI / 180.0;
constexpr double kDph = kDeg / 3600.0;
const double kDpsh = kDeg / std::sqrt(3600.0);
constexpr double kUg = kG0 / 1000.0 / 1000.0;

struct St {
  V3 p, v;
  Q4 q;
  V3 ba, bw, g;
};
St load(const double* s) {
  return {{s[0], s[1], s[2]}, {s[3], s[4], s[5]}, {s[6], s[7], s[8], s[9]},
          {s[10], s[11], s[12]}, {s[13], s[14], s[15]}, {s[16], s[17], s[18]}};
}
void store(const St& st, double* s) {
  double v[19] = {st.p.x, st.p.y, st.p.z, st.v.x, st.v.y, st.v.z, st.q.w, st.q.x, st.q.y, st.q.z,
                  st.ba.x, st.ba.y, st.ba.z, st.bw.x, st.bw.y, st.bw.z, st.g.x, st.g.y, st.g.z};
  std::memcpy(s, v, sizeof v);
}
inline void set_block(double* M, int n, int r, int c, const M3& b, double scale = 1.0) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[(r + i) * n + c + j] = scale * b.m[i * 3 + j];
}
inline M3 get_block(const double* M, int n, int r, int c) {
  M3 b;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) b.m[i * 3 + j] = M[(r + i) * n + c + j];
  return b;
}
const M3 kI3{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
}  // namespace

extern "C" {

void lins_filter_default_params(lins_filter_params* p) {
  // lins/config/exp_config/exp_port.yaml:29-62
  p->acc_n = 70000, p->gyr_n = 0.1, p->acc_w = 500, p->gyr_w = 0.05;
  for (int i = 0; i < 3; ++i) p->init_pos_std[i] = p->init_vel_std[i] = p->init_att_std[i] = 0.0;
  p->init_acc_std[0] = 0.01, p->init_acc_std[1] = 0.01, p->init_acc_std[2] = 0.02;
  p->init_gyr_std[0] = p->init_gyr_std[1] = p->init_gyr_std[2] = 0.002;
}

void lins_filter_init(lins_filter* f, const lins_filter_params* p, const double* vn, const double* ba,
                      const double* bw) {
  std::memset(f, 0, sizeof *f);
  f->prm = *p;
  St s{{0, 0, 0}, {vn[0], vn[1], vn[2]}, {1, 0, 0, 0}, {ba[0], ba[1], ba[2]}, {bw[0], bw[1], bw[2]},
       {0, 0, -kG0}};
  store(s, f->state);
  // initializeCovariance(0), KF:247-286
  double* C = f->cov;
  for (int i = 0; i < 3; ++i) {
    C[(0 + i) * 18 + 0 + i] = p->init_pos_std[i] * p->init_pos_std[i];
    C[(3 + i) * 18 + 3 + i] = p->init_vel_std[i] * p->init_vel_std[i];
    double a = p->init_att_std[i] * kDeg;
    C[(6 + i) * 18 + 6 + i] = a * a;
    C[(9 + i) * 18 + 9 + i] = p->init_acc_std[i] * p->init_acc_std[i];
    C[(12 + i) * 18 + 12 + i] = p->init_gyr_std[i] * p->init_gyr_std[i];
    C[(15 + i) * 18 + 15 + i] = 0.01;
  }
  // noise_, KF:263-266, 307-311
  double peba = std::pow(p->acc_n * kUg, 2), pebg = std::pow(p->gyr_n * kDph, 2);
  double pweba = std::pow(p->acc_w * kUg, 2), pwebg = std::pow(p->gyr_w * kDpsh, 2);
  for (int i = 0; i < 3; ++i) {
    f->noise[(0 + i) * 12 + 0 + i] = peba;
    f->noise[(3 + i) * 12 + 3 + i] = pebg;
    f->noise[(6 + i) * 12 + 6 + i] = pweba;
    f->noise[(9 + i) * 12 + 9 + i] = pwebg;
  }
}

void lins_filter_predict(lins_filter* f, double dt, const double* acc_, const double* gyr_) {
  V3 acc{acc_[0], acc_[1], acc_[2]}, gyr{gyr_[0], gyr_[1], gyr_[2]};
  if (!f->has_imu) {  // KF:130-134
    f->has_imu = 1;
    std::memcpy(f->acc_last, acc_, 3 * sizeof(double));
    std::memcpy(f->gyr_last, gyr_, 3 * sizeof(double));
  }
  V3 acc_last{f->acc_last[0], f->acc_last[1], f->acc_last[2]};
  V3 gyr_last{f->gyr_last[0], f->gyr_last[1], f->gyr_last[2]};
  St s = load(f->state);
  // mid-point integration, KF:137-147
  V3 un_acc_0 = qrot(s.q, acc_last - s.ba) + s.g;
  V3 un_gyr = 0.5 * (gyr_last + gyr) - s.bw;
  s.q = qnormalized(qmul(s.q, axis2quat(dt * un_gyr)));
  V3 un_acc_1 = qrot(s.q, acc - s.ba) + s.g;
  V3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
  s.p = s.p + dt * s.v + (0.5 * dt * dt) * un_acc;
  s.v = s.v + dt * un_acc;

  // F_t / G_t blocks, KF:149-169
  static thread_local double Ft[324], Ft2[324], F[324], T[324], Gt[18 * 12], GQ[18 * 12];
  std::memset(Ft, 0, sizeof Ft);
  M3 R = qmat(s.q);
  M3 negR;
  for (int k = 0; k < 9; ++k) negR.m[k] = -R.m[k];
  set_block(Ft, 18, 0, 3, kI3);
  set_block(Ft, 18, 3, 6, mmul(negR, skew(acc - s.ba)));
  set_block(Ft, 18, 3, 9, negR);
  set_block(Ft, 18, 3, 15, kI3);
  set_block(Ft, 18, 6, 6, skew(gyr - s.bw), -1.0);
  set_block(Ft, 18, 6, 12, kI3, -1.0);
  std::memset(Gt, 0, sizeof Gt);
  set_block(Gt, 12, 3, 0, negR, dt);
  set_block(Gt, 12, 6, 3, kI3, -dt);
  set_block(Gt, 12, 9, 6, kI3, dt);
  set_block(Gt, 12, 12, 9, kI3, dt);
  // F = I + Ft dt + 0.5 Ft Ft dt dt, KF:173
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) {
      double a = 0;
      for (int k = 0; k < 18; ++k) a += Ft[i * 18 + k] * Ft[k * 18 + j];
      Ft2[i * 18 + j] = a;
    }
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j)
      F[i * 18 + j] = (i == j ? 1.0 : 0.0) + Ft[i * 18 + j] * dt + 0.5 * Ft2[i * 18 + j] * dt * dt;
  // P = F P F^T + G Q G^T, symmetrised, KF:176-178
  double* P = f->cov;
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) {
      double a = 0;
      for (int k = 0; k < 18; ++k) a += F[i * 18 + k] * P[k * 18 + j];
      T[i * 18 + j] = a;
    }
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 12; ++j) {
      double a = 0;
      for (int k = 0; k < 12; ++k) a += Gt[i * 12 + k] * f->noise[k * 12 + j];
      GQ[i * 12 + j] = a;
    }
  static thread_local double Pn[324];
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) {
      double a = 0;
      for (int k = 0; k < 18; ++k) a += T[i * 18 + k] * F[j * 18 + k];
      double b = 0;
      for (int k = 0; k < 12; ++k) b += GQ[i * 12 + k] * Gt[j * 12 + k];
      Pn[i * 18 + j] = a + b;
    }
  for (int i = 0; i < 18; ++i)
    for (int j = 0; j < 18; ++j) P[i * 18 + j] = 0.5 * (Pn[i * 18 + j] + Pn[j * 18 + i]);

  store(s, f->state);
  f->time += dt;
  std::memcpy(f->acc_last, acc_, 3 * sizeof(double));
  std::memcpy(f->gyr_last, gyr_, 3 * sizeof(double));
}

void lins_filter_reset1(lins_filter* f) {
  // KF:320-352.  Note (reference quirk, kept): q is set to identity BEFORE gn_
  // is rotated, so the gravity rotation is a no-op; only its norm is reset.
  St s = load(f->state);
  const lins_filter_params& p = f->prm;
  M3 vel = get_block(f->cov, 18, 3, 3), accb = get_block(f->cov, 18, 9, 9);
  M3 gyrb = get_block(f->cov, 18, 12, 12), gra = get_block(f->cov, 18, 15, 15);
  M3 R = qmat(s.q), Rt = mtrans(R);  // q.inverse()*M*q on matrices == R^T M R
  std::memset(f->cov, 0, sizeof f->cov);
  for (int i = 0; i < 3; ++i) {
    f->cov[(0 + i) * 18 + 0 + i] = p.init_pos_std[i] * p.init_pos_std[i];
    double a = p.init_att_std[i] * kDeg;
    f->cov[(6 + i) * 18 + 6 + i] = a * a;
  }
  set_block(f->cov, 18, 3, 3, mmul(mmul(Rt, vel), R));
  set_block(f->cov, 18, 9, 9, accb);
  set_block(f->cov, 18, 12, 12, gyrb);
  set_block(f->cov, 18, 15, 15, mmul(mmul(Rt, gra), R));
  s.p = {0, 0, 0};
  s.v = qrot(qinverse(s.q), s.v);
  s.q = {1, 0, 0, 0};
  s.g = qrot(qinverse(s.q), s.g);
  s.g = (9.81 / norm(s.g)) * s.g;
  store(s, f->state);
}

}  // extern "C"
