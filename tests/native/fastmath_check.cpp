// Host build of the short-series rotation helpers (csrc/lins_math.h) and of the Gauss-Jordan solve
// (csrc/lins_solve6.h) — the same source the device compiles, with correctly rounded fma / division, so the bits
// are the device's — measured against long-double libm (binary128 where the reference cancels) and against the textbook routes.  Prints one line of
// "name value" pairs; tests/test_fastmath.py asserts on them.  Usage: fastmath_check [seed]
#include <math.h>
#include <quadmath.h>
#include <stdio.h>
#include <stdlib.h>

#include "lins_math.h"
#include "lins_solve6.h"

using namespace lins;

static double ulp_of(double x) { return nextafter(fabs(x), INFINITY) - fabs(x); }
static double rnd() { return (double)rand() / RAND_MAX; }

int main(int argc, char** argv) {
  srand(argc > 1 ? atoi(argv[1]) : 1);
  // 1. series vs long double libm, over their stated ranges
  double e_sinc = 0, e_cos = 0, e_atanc = 0, e_q = 0;
  for (int k = 0; k < 200000; ++k) {
    const double h = 0.5 * pow(rnd(), 3.0) + (k % 7 == 0 ? 1e-9 * rnd() : 0.0);  // dense near 0, up to 0.5
    double s, c;
    lins_sinc_cos_small(h * h, s, c);
    const long double hl = sqrtl((long double)(h * h));
    const long double sl = hl > 0 ? sinl(hl) / hl : 1.0L, cl = cosl(hl);
    e_sinc = fmax(e_sinc, fabs((double)(s - sl)) / ulp_of((double)sl));
    e_cos = fmax(e_cos, fabs((double)(c - cl)) / ulp_of((double)cl));
    const double t = 0.125 * pow(rnd(), 2.0);
    double A, Q;
    lins_atanc_small(t * t, A, Q);
    const long double tl = sqrtl((long double)(t * t));
    const long double al = tl > 0 ? atanl(tl) / tl : 1.0L;
    e_atanc = fmax(e_atanc, fabs((double)(A - al)) / ulp_of((double)al));
    if (t > 1e-6) {  // ((1 - A) / z cancels: the reference is formed in binary128)
      const __float128 tq = sqrtq((__float128)(t * t)), qq = ((__float128)1.0 - atanq(tq) / tq) / (tq * tq);
      e_q = fmax(e_q, fabs((double)((__float128)Q - qq)) / ulp_of((double)qq));
    }
  }
  // 2. the maps against the textbook routes of the same header (libm sin / cos / atan2): componentwise, in units of
  // the ulp of the LARGEST component (the tests' tolerance is absolute), inside and outside the series' ranges
  double e_a2q = 0, e_q2a = 0, e_phi = 0, e_gt = 0;
  int n_small = 0, n_general = 0;
  for (int k = 0; k < 200000; ++k) {
    const double sc = k % 4 == 0 ? 3.0 : (k % 4 == 1 ? 0.3 : (k % 4 == 2 ? 1e-3 : 1e-8));
    const V3 v{sc * (rnd() - 0.5), sc * (rnd() - 0.5), sc * (rnd() - 0.5)};
    const Q4 a = axis2quat(v), b = axis2quat_fast(v);
    e_a2q = fmax(e_a2q, fmax(fmax(fabs(a.w - b.w), fabs(a.x - b.x)), fmax(fabs(a.y - b.y), fabs(a.z - b.z))) / ulp_of(1.0));
    Q4 q = a;
    if (k % 5 == 0) q.w = -q.w;                         // w < 0: the general route
    if (k % 11 == 0) q = {q.w * 1.7, q.x * 1.7, q.y * 1.7, q.z * 1.7};  // not normalised: both routes are scale free
    const V3 p = quat2axis(q), pf = quat2axis_fast(q);
    const double pm = fmax(fmax(fabs(p.x), fabs(p.y)), fmax(fabs(p.z), 1e-300));
    e_q2a = fmax(e_q2a, fmax(fmax(fabs(p.x - pf.x), fabs(p.y - pf.y)), fabs(p.z - pf.z)) / ulp_of(pm));
    V3 phi{0, 0, 0};
    M3 gt{{0, 0, 0, 0, 0, 0, 0, 0, 0}};
    if (phi_and_gt_small(q, phi, gt)) {
      ++n_small;
      e_phi = fmax(e_phi, fmax(fmax(fabs(p.x - phi.x), fabs(p.y - phi.y)), fabs(p.z - phi.z)) / ulp_of(pm));
      const M3 r = mtrans(rinvleft(V3{-p.x, -p.y, -p.z}));
      for (int i = 0; i < 9; ++i) e_gt = fmax(e_gt, fabs(r.m[i] - gt.m[i]) / ulp_of(1.0));
    } else {
      ++n_general;
    }
  }
  // 2b. the de-skew's table form (coefficients read from memory: LDS on the device) is the literal form, bit for bit
  int tab_diff = 0;
  {
    double tab[kSincCosTab];
    lins_sinc_cos_table(tab);
    for (int k = 0; k < 200000; ++k) {
      const double sc = k % 3 == 0 ? 2.5 : (k % 3 == 1 ? 0.05 : 1e-9);
      const V3 v{sc * (rnd() - 0.5), sc * (rnd() - 0.5), sc * (rnd() - 0.5)};
      const Q4 a = axis2quat_fast(v), b = axis2quat_tab(v, tab);
      if (!(a.w == b.w && a.x == b.x && a.y == b.y && a.z == b.z)) ++tab_diff;
    }
  }
  // 3. Gauss-Jordan: residual of random systems (diagonally dominant, pivoting forced, mixed scales)
  double e_res = 0;
  for (int k = 0; k < 20000; ++k) {
    double a[6][7], a0[6][7], x[6];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 7; ++j) a[i][j] = (rnd() - 0.5) * 2.0;
    if (k % 3 == 0)
      for (int i = 0; i < 6; ++i) a[i][i] += 8.0;
    if (k % 3 == 1) a[0][0] = 1e-12;
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 7; ++j) a0[i][j] = a[i][j];
    gj_solve6(a, x);
    if (k % 3 == 0) {  // (well conditioned: the residual is a rounding bound)
      for (int i = 0; i < 6; ++i) {
        double r = -a0[i][6];
        for (int j = 0; j < 6; ++j) r += a0[i][j] * x[j];
        e_res = fmax(e_res, fabs(r));
      }
    }
  }
  // 4. a dump of systems + solutions for numpy (stdout line 2..): 50 systems
  printf("sinc_ulp %.3f cos_ulp %.3f atanc_ulp %.3f atanq_ulp %.3f a2q_ulp %.3f q2a_ulp %.3f phi_ulp %.3f gt_ulp %.3f n_small %d n_general %d tab_diff %d gj_res %.3e\n",
         e_sinc, e_cos, e_atanc, e_q, e_a2q, e_q2a, e_phi, e_gt, n_small, n_general, tab_diff, e_res);
  for (int k = 0; k < 50; ++k) {
    double a[6][7], x[6];
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 7; ++j) a[i][j] = (rnd() - 0.5) * (k % 2 ? 2.0 : 2e3) + (i == j && k % 5 == 0 ? 5.0 : 0.0);
    if (k % 7 == 3) a[0][0] = 0.0;  // a zero pivot candidate: exchange needed
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 7; ++j) printf("%.17g ", a[i][j]);
    gj_solve6(a, x);
    for (int i = 0; i < 6; ++i) printf("%.17g ", x[i]);
    printf("\n");
  }
  return 0;
}
