"""Synthetic scan-to-map problems (planes + line features) for the scan-to-map row's tests."""
import numpy as np


def rot(rx, ry, rz):
    """pointAssociateToMap's rotation (LM:594-607): rotate about z, then x, then y."""
    cz, sz, cx, sx, cy, sy = np.cos(rz), np.sin(rz), np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    return Ry @ Rx @ Rz


def make_problem(defs, seed, n_map_surf=20000, n_map_corner=3000, n_scan_surf=900, n_scan_corner=250, noise=0.01,
                 perturb=(0.01, 0.05)):
    rng = np.random.default_rng(seed)
    # a room: floor, ceiling-less, four walls (surf); vertical and horizontal edges (corner)
    def plane(n, origin, u, v):
        a, b = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
        return origin + a[:, None] * u + b[:, None] * v
    L, W, H = 30.0, 20.0, 6.0
    o = np.array([-L / 2, -W / 2, -1.5])
    ex, ey, ez = np.array([L, 0, 0.0]), np.array([0, W, 0.0]), np.array([0, 0, H])
    parts = [plane(n_map_surf // 5, o, ex, ey), plane(n_map_surf // 5, o, ex, ez), plane(n_map_surf // 5, o + ey, ex, ez),
             plane(n_map_surf // 5, o, ey, ez), plane(n_map_surf - 4 * (n_map_surf // 5), o + ex, ey, ez)]
    map_surf = np.concatenate(parts) + rng.normal(0, noise, (n_map_surf, 3))
    edges = []
    corners = [o, o + ex, o + ey, o + ex + ey]
    per = n_map_corner // 8
    for c0 in corners:
        edges.append(c0 + rng.uniform(0, 1, per)[:, None] * ez)
    for a, d in ((o, ex), (o + ey, ex), (o, ey), (o + ex, ey)):
        edges.append(a + rng.uniform(0, 1, per)[:, None] * d)
    map_corner = np.concatenate(edges) + rng.normal(0, noise / 2, (8 * per, 3))
    # true transform (sensor -> map) and the scan = map points seen from the sensor frame
    T_true = np.concatenate([rng.normal(0, 0.05, 3), rng.normal(0, 0.5, 3)])
    R = rot(*T_true[:3])
    def to_sensor(pm):
        return (pm - T_true[3:]) @ R  # R^T (p - t)
    ss = to_sensor(map_surf[rng.choice(n_map_surf, n_scan_surf, replace=False)]) + rng.normal(0, noise, (n_scan_surf, 3))
    sc = to_sensor(map_corner[rng.choice(len(map_corner), n_scan_corner, replace=False)]) + rng.normal(0, noise, (n_scan_corner, 3))
    T0 = T_true + np.concatenate([rng.normal(0, perturb[0], 3), rng.normal(0, perturb[1], 3)])
    pad = lambda a: np.concatenate([a, np.zeros((len(a), 1))], 1).astype(np.float32)
    return defs.MapProblem(pad(map_corner), pad(map_surf), pad(sc), pad(ss), T0.astype(np.float32)), T_true


def make_corridor(defs, seed, n_map_surf=12000, n_map_corner=1500, n_scan_surf=900, n_scan_corner=200, noise=0.01):
    """A corridor along x: floor and two side walls, edges along x only — nothing constrains the translation along
    x, so LMOptimization's eigen-test must flag the problem as degenerate (LM:1589-1614)."""
    rng = np.random.default_rng(seed)
    L, W, H = 40.0, 4.0, 3.0
    n3 = n_map_surf // 3
    floor = np.stack([rng.uniform(-L / 2, L / 2, n3), rng.uniform(-W / 2, W / 2, n3), np.full(n3, -1.5)], 1)
    wl = np.stack([rng.uniform(-L / 2, L / 2, n3), np.full(n3, -W / 2), rng.uniform(-1.5, -1.5 + H, n3)], 1)
    n_r = n_map_surf - 2 * n3
    wr = np.stack([rng.uniform(-L / 2, L / 2, n_r), np.full(n_r, W / 2), rng.uniform(-1.5, -1.5 + H, n_r)], 1)
    map_surf = np.concatenate([floor, wl, wr]) + rng.normal(0, noise, (n_map_surf, 3))
    per = n_map_corner // 4
    edges = [np.stack([rng.uniform(-L / 2, L / 2, per), np.full(per, y), np.full(per, z)], 1)
             for y in (-W / 2, W / 2) for z in (-1.5, -1.5 + H)]
    map_corner = np.concatenate(edges) + rng.normal(0, noise / 2, (4 * per, 3))
    T_true = np.concatenate([rng.normal(0, 0.02, 3), rng.normal(0, 0.2, 3)])
    R = rot(*T_true[:3])
    to_sensor = lambda pm: (pm - T_true[3:]) @ R
    near = lambda a: a[np.abs(a[:, 0]) < L / 2 - 3]  # keep the scan away from the corridor's open ends
    ms, mc = near(map_surf), near(map_corner)
    ss = to_sensor(ms[rng.choice(len(ms), n_scan_surf, replace=False)]) + rng.normal(0, noise, (n_scan_surf, 3))
    sc = to_sensor(mc[rng.choice(len(mc), n_scan_corner, replace=False)]) + rng.normal(0, noise, (n_scan_corner, 3))
    T0 = T_true + np.concatenate([rng.normal(0, 0.005, 3), rng.normal(0, 0.03, 3)])
    pad = lambda a: np.concatenate([a, np.zeros((len(a), 1))], 1).astype(np.float32)
    return defs.MapProblem(pad(map_corner), pad(map_surf), pad(sc), pad(ss), T0.astype(np.float32)), T_true
