"""A hand-built scan pair that takes the reference's residual blow-up branch (SE:566-570, diverged == 1).

Iteration 0 accepts exactly ONE plane row with a 1 mm residual — every other query sits 10 m from any target
(prior t_x = +10 m, all queries stamped at the end of the sweep so the full translation applies).  That plane is
almost parallel to x (normal ~ (1e-4, 0, 1)) and x is the only direction the prior covariance leaves free
(P_xx = 1e6, everything else 0), so the 1 mm residual is explained by a -10 m jump of t_x.  In iteration 1 the
other queries have landed 0.1 m above their (horizontal) planes: ~20 rows of ~0.096 each, |r| = 0.43 > 10 |r_0|.
Targets are ring-sorted with ring ids 0 / 1, so every grid kernel accepts the pair."""
import numpy as np


def make_diverging_pair(pkg, n_other=20):
    f = np.float32
    p3, p2, p1 = [], [], []  # ring 0 third points; ring 1 second / first points (second before first: backward walk)
    queries = []
    t_prior = np.array([10.0, 0.0, 0.0])
    for k in range(n_other + 1):
        c = np.array([0.0, 25.0 * k, -1.5])
        tilt = 1e-4 if k == 0 else 0.0  # plane z = c_z - tilt * (x - c_x)
        pts = []
        for dx, dy in ((0.0, 0.0), (-0.4, 0.35), (0.45, 0.3)):
            pts.append([c[0] + dx, c[1] + dy, c[2] - tilt * dx])
        a, b, d = (np.array(v) for v in pts)
        p1.append(a), p2.append(b), p3.append(d)
        n = np.cross(a - b, a - d)
        n /= np.linalg.norm(n)
        off = 1e-3 if k == 0 else 0.1
        sel = a + np.array([0.03, 0.02, 0.0]) + off * n * np.sign(n[2])  # closest to the first point, `off` above the plane
        raw = sel - (t_prior if k == 0 else np.array([0.1, 0.0, 0.0]))  # k > 0: in place once t_x has dropped to ~0.1
        queries.append(raw)
    surf_last = np.zeros((3 * (n_other + 1), 4), dtype=f)
    surf_last[: n_other + 1, :3] = np.array(p3)
    surf_last[: n_other + 1, 3] = 0.05  # ring 0
    for k in range(n_other + 1):
        surf_last[n_other + 1 + 2 * k, :3] = p2[k]
        surf_last[n_other + 2 + 2 * k, :3] = p1[k]
    surf_last[n_other + 1:, 3] = 1.05  # ring 1
    surf_flat = np.zeros((n_other + 1, 4), dtype=f)
    surf_flat[:, :3] = np.array(queries)
    surf_flat[:, 3] = f(1.0) + f(0.1)  # ring 1, relative time 1: s = 10 * frac = 1
    state = np.zeros(19)
    state[0:3] = t_prior
    state[6] = 1.0
    state[18] = -9.81
    cov = np.zeros((18, 18))
    cov[0, 0] = 1e6
    empty = np.zeros((0, 4), dtype=f)
    return pkg.ScanPair(surf_flat, empty, surf_last, empty, state, cov)
