"""SECOND, independently written CPU restatement of the IESKF update path (numpy / pure Python).

TEST INFRASTRUCTURE ONLY — imported by tests/ (never by the product package, never timed).
Written in rounds 1-2, when the reference could not be run here at all; since round 3 oracle/lins_oracle.cpp is pinned
to the reference's own compiled text (oracle/_ref, tests/test_ref.py) and this file is pinned through it.

Why it exists: the only other ground truth is oracle/lins_oracle.cpp.  This file was written straight from the
reference's lines, sharing no code and no helper with that oracle (nor with csrc/), so that a misreading of the
reference would have to be made twice, independently, to go unnoticed: tests/test_np_oracle.py asserts that the
two restatements produce the same index triplets, accepted sets, rows and posterior on the golden pairs.

Restated (paths relative to /root/reference/lins/include/):
  transformToStart                       StateEstimator.hpp:1066-1080
  findCorrespondingSurfFeatures          StateEstimator.hpp:829-953
  findCorrespondingCornerFeatures        StateEstimator.hpp:955-1063
  performIESKF (dense M x M form)        StateEstimator.hpp:465-600
  GlobalState boxPlus / boxMinus         KalmanFilter.hpp:71-94
  wrap_pi, enforceSymmetry, axis2Quat, Quat2axis, skew, Rinvleft
                                         math_utils.h:27-41, 43-73, 75-88, 196-204, 304-321
Eigen pieces the reference leans on, restated from their documented formulas: Quaternion product,
Quaternion * vector (v + 2w(u x v) + 2 u x (u x v)), toRotationMatrix, inverse, normalized, LLT solve.
The kd-tree's nearestKSearch(1) is the exact nearest neighbour in f32 squared distance (FLANN L2_Simple
accumulates ((dx^2) + dy^2) + dz^2 in float); ties go to the lowest index.

Float semantics follow the C++ expression types line by line (float where the reference computes in float,
double elsewhere); libm calls go through Python's math module (glibc on this box), not numpy's vector loops.
"""
import math

import numpy as np

F = np.float32


# ---- math_utils.h -------------------------------------------------------------------------------
def wrap_pi(x):  # MU:27-37
    while x >= math.pi:
        x -= 2.0 * math.pi
    while x < -math.pi:
        x += 2.0 * math.pi
    return x


def skew(v):  # MU:196-204
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def axis2quat(vec):  # MU:61-73 -> MU:43-59 ; quaternion as (w, x, y, z)
    theta = math.sqrt(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2])
    if theta < 1e-10:
        return np.array([1.0, 0.0, 0.0, 0.0])
    a = vec / theta
    m = math.sin(theta / 2.0)
    return np.array([math.cos(theta / 2.0), a[0] * m, a[1] * m, a[2] * m])


def quat2axis(q):  # MU:75-88
    mag = math.sqrt(q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    v = np.array([q[1], q[2], q[3]])
    if mag >= 1e-10:
        v = v / mag
        v = v * wrap_pi(2.0 * math.atan2(mag, q[0]))
    return v


def rinvleft(axis):  # MU:304-321
    theta = math.sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2])
    if theta < 1e-10:
        return np.eye(3)
    h = theta / 2.0
    a = axis / theta
    s = h * (math.cos(h) / math.sin(h))
    return s * np.eye(3) + (1.0 - s) * np.outer(a, a) - h * skew(a)


# ---- Eigen::Quaterniond ---------------------------------------------------------------------------
def qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                     a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])


def qrotate(q, v):
    u = np.array([q[1], q[2], q[3]])
    uv = np.cross(u, v)
    uv = uv + uv
    return v + q[0] * uv + np.cross(u, uv)


def qinverse(q):
    n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]
    return np.array([q[0], -q[1], -q[2], -q[3]]) / n2


def qnormalized(q):
    return q / math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])


def qmatrix(q):
    w, x, y, z = q
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1.0 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1.0 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1.0 - (txx + tyy)]])


# ---- KalmanFilter.hpp: GlobalState as a 19-vector (p, v, q(wxyz), ba, bw, g) ---------------------------
def box_plus(s, dx):  # KF:71-81 ; dx blocks: pos 0, vel 3, att 6, acc 9, gyr 12, gra 15
    o = np.array(s, dtype=np.float64)
    o[0:3] = s[0:3] + dx[0:3]
    o[3:6] = s[3:6] + dx[3:6]
    o[10:13] = s[10:13] + dx[9:12]
    o[13:16] = s[13:16] + dx[12:15]
    o[6:10] = qnormalized(qmul(s[6:10], axis2quat(np.array(dx[6:9]))))
    o[16:19] = s[16:19] + dx[15:18]
    return o


def box_minus(a, b):  # KF:84-94: a (-) b
    d = np.zeros(18)
    d[0:3] = a[0:3] - b[0:3]
    d[3:6] = a[3:6] - b[3:6]
    d[9:12] = a[10:13] - b[10:13]
    d[12:15] = a[13:16] - b[13:16]
    d[6:9] = quat2axis(qmul(qinverse(b[6:10]), a[6:10]))
    d[15:18] = a[16:19] - b[16:19]
    return d


# ---- StateEstimator.hpp ------------------------------------------------------------------------------
def transform_to_start(prm, lin, pt):  # SE:1066-1080 ; pt = (x, y, z, intensity) float32
    inten = F(pt[3])
    frac = F(inten - F(int(inten)))  # float - int -> float
    s = float(F(F(1.0) / F(prm.scan_period))) * float(frac)  # (1.f / SCAN_PERIOD) is a float, the product a double
    p2 = np.array([float(pt[0]), float(pt[1]), float(pt[2])])
    phi = quat2axis(lin[6:10])
    r21 = axis2quat(s * phi)  # (".normalized()" on the next line of the reference discards its result)
    p1 = qrotate(r21, p2) + s * lin[0:3]
    return np.array([F(p1[0]), F(p1[1]), F(p1[2]), inten], dtype=np.float32)


def _sqdist(tg, sel):
    """f32 ((dx*dx + dy*dy) + dz*dz) of every target to sel."""
    dx = tg[:, 0] - sel[0]
    dy = tg[:, 1] - sel[1]
    dz = tg[:, 2] - sel[2]
    return (dx * dx + dy * dy) + dz * dz  # float32 arrays: every operation rounds to float


def _ring(tg):
    return tg[:, 3].astype(np.int32)  # int(intensity): truncation


def surf_search(prm, tg, nq, sel):  # SE:844-913
    d = _sqdist(tg, sel)
    closest, m2, m3 = -1, -1, -1
    j1 = int(np.argmin(d))  # (first minimum = lowest index)
    if d[j1] < F(prm.nearest_sq_dist):
        closest = j1
        ring = _ring(tg)
        scan = int(ring[j1])
        d2 = d3 = F(prm.nearest_sq_dist)
        for j in range(closest + 1, nq):  # bounded by the QUERY count (SE:859)
            if j >= len(tg):
                break  # (the reference would read out of bounds: guarded like every other restatement here)
            if ring[j] > scan + 2.5:
                break
            if ring[j] <= scan:
                if d[j] < d2:
                    d2, m2 = d[j], j
            else:
                if d[j] < d3:
                    d3, m3 = d[j], j
        for j in range(closest - 1, -1, -1):
            if ring[j] < scan - 2.5:
                break
            if ring[j] >= scan:
                if d[j] < d2:
                    d2, m2 = d[j], j
            else:
                if d[j] < d3:
                    d3, m3 = d[j], j
    return closest, m2, m3


def corner_search(prm, tg, nq, sel):  # SE:970-1028
    d = _sqdist(tg, sel)
    closest, m2 = -1, -1
    j1 = int(np.argmin(d))
    if d[j1] < F(prm.nearest_sq_dist):
        closest = j1
        ring = _ring(tg)
        scan = int(ring[j1])
        d2 = F(prm.nearest_sq_dist)
        for j in range(closest + 1, nq):
            if j >= len(tg):
                break
            if ring[j] > scan + 2.5:
                break
            if ring[j] > scan and d[j] < d2:
                d2, m2 = d[j], j
        for j in range(closest - 1, -1, -1):
            if ring[j] < scan - 2.5:
                break
            if ring[j] < scan and d[j] < d2:
                d2, m2 = d[j], j
    return closest, m2


def surf_row(prm, it, sel, t1, t2, t3):  # SE:917-951 -> (accepted, coeff[4] float32)
    p0, p1, p2, p3 = (np.array([float(v[0]), float(v[1]), float(v[2])]) for v in (sel, t1, t2, t3))
    m_vec = skew(p1 - p2) @ (p1 - p3)
    r = float((p0 - p1) @ m_vec)
    m = math.sqrt(m_vec[0] * m_vec[0] + m_vec[1] * m_vec[1] + m_vec[2] * m_vec[2])
    res = F(r / m)
    jac = m_vec / m
    s = F(1.0)
    if it >= prm.icp_freq:
        n2 = F(F(F(sel[0]) * F(sel[0]) + F(sel[1]) * F(sel[1])) + F(sel[2]) * F(sel[2]))
        s = F(1.0 - 1.8 * float(abs(res)) / float(np.sqrt(np.sqrt(n2))))  # sqrt(sqrt(float)) stays float
    if s > 0.1 and res != 0:
        return 1, np.array([F(float(s) * jac[0]), F(float(s) * jac[1]), F(float(s) * jac[2]), F(s * res)], dtype=np.float32)
    return 0, np.zeros(4, dtype=np.float32)


def corner_row(prm, it, sel, t1, t2):  # SE:1031-1061
    p0, p1, p2 = (np.array([float(v[0]), float(v[1]), float(v[2])]) for v in (sel, t1, t2))
    pv = skew(p0 - p1) @ (p0 - p2)
    r = F(math.sqrt(pv[0] * pv[0] + pv[1] * pv[1] + pv[2] * pv[2]))
    dv = p1 - p2
    d12 = F(math.sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]))
    with np.errstate(divide="ignore", invalid="ignore"):
        res = F(r / d12)
        jac = (pv @ skew(p2 - p1)) / float(F(d12 * r))
    s = F(1.0)
    if it >= prm.icp_freq:
        s = F(1.0 - 1.8 * float(abs(res)))
    if s > 0.1 and res != 0:
        return 1, np.array([F(float(s) * jac[0]), F(float(s) * jac[1]), F(float(s) * jac[2]), F(s * res)], dtype=np.float32)
    return 0, np.zeros(4, dtype=np.float32)


def correspondences(prm, pair, lin, it):
    """One pass of both feature kinds at linearisation state `lin`: per query (ind1, ind2, ind3, accepted, coeff, sel)."""
    out = {}
    for kind, q, tg in (("surf", pair.surf_flat, pair.surf_last), ("corner", pair.corner_sharp, pair.corner_last)):
        q = np.asarray(q).view(np.float32).reshape(-1, 4)
        tg = np.asarray(tg).view(np.float32).reshape(-1, 4)
        n = len(q)
        rec = dict(ind=np.full((n, 3), -1, dtype=np.int32), acc=np.zeros(n, dtype=np.int32),
                   coeff=np.zeros((n, 4), dtype=np.float32), sel=np.zeros((n, 4), dtype=np.float32))
        for i in range(n):
            sel = transform_to_start(prm, lin, q[i])
            rec["sel"][i] = sel
            if len(tg) == 0:
                continue
            if kind == "surf":
                c, m2, m3 = surf_search(prm, tg, n, sel)
                rec["ind"][i] = (c, m2, m3)
                if m2 >= 0 and m3 >= 0:
                    rec["acc"][i], rec["coeff"][i] = surf_row(prm, it, sel, tg[c], tg[m2], tg[m3])
            else:
                c, m2 = corner_search(prm, tg, n, sel)
                rec["ind"][i] = (c, m2, -1)
                if m2 >= 0:
                    rec["acc"][i], rec["coeff"][i] = corner_row(prm, it, sel, tg[c], tg[m2])
        out[kind] = rec
    return out


def perform_ieskf(prm, pair):
    """SE:465-583, 594-598 (the ICP fallback of a diverged filter is not restated here).
    -> dict(state, cov, iters, converged, diverged, m_surf, m_corner, trace=[per-iteration correspondences])."""
    pk = np.array(pair.cov, dtype=np.float64).reshape(18, 18)
    filt = np.array(pair.state, dtype=np.float64)
    lin = filt.copy()
    residual_norm = 1e6
    converged = diverged = False
    trace = []
    kk = hk = None
    m_surf = m_corner = 0
    iters = 0
    sq = np.asarray(pair.surf_flat).view(np.float32).reshape(-1, 4)
    cq = np.asarray(pair.corner_sharp).view(np.float32).reshape(-1, 4)
    for it in range(prm.num_iter):
        if converged or diverged:
            break
        iters = it + 1
        c = correspondences(prm, pair, lin, it)
        trace.append(c)
        keyp = np.concatenate([sq[c["surf"]["acc"] == 1], cq[c["corner"]["acc"] == 1]])  # surf rows first (SE:500-503)
        coff = np.concatenate([c["surf"]["coeff"][c["surf"]["acc"] == 1], c["corner"]["coeff"][c["corner"]["acc"] == 1]])
        m_surf, m_corner = int(c["surf"]["acc"].sum()), int(c["corner"]["acc"].sum())
        m = len(keyp)
        hk = np.zeros((m, 18))
        residual = np.zeros(m)
        axis = quat2axis(lin[6:10])
        rmat = qmatrix(lin[6:10])
        g = rinvleft(-axis)
        for i in range(m):  # SE:516-532
            p2 = np.array([float(keyp[i][0]), float(keyp[i][1]), float(keyp[i][2])])
            cx = np.array([float(coff[i][0]), float(coff[i][1]), float(coff[i][2])])
            residual[i] = prm.lidar_scale * float(coff[i][3])
            hk[i, 6:9] = cx @ (-rmat @ skew(p2)) @ g
            hk[i, 0:3] = cx
        rk = (prm.lidar_std * prm.lidar_std) * np.eye(m)
        py = hk @ pk @ hk.T + rk  # SE:542-546
        if m:
            low = np.linalg.cholesky(py)
            pyinv = np.linalg.solve(low.T, np.linalg.solve(low, np.eye(m)))
        else:
            pyinv = np.zeros((0, 0))
        kk = pk @ hk.T @ pyinv
        dif = box_minus(filt, lin)
        upd = -kk @ (residual + hk @ dif) + dif  # SE:548-549
        if np.isnan(upd).any():  # SE:552-563
            diverged = True
            break
        rn = float(np.sqrt(residual @ residual))
        if rn > residual_norm * 10:  # SE:566-570
            diverged = True
            break
        lin = box_plus(lin, upd)  # SE:573
        if float(np.sqrt(upd @ upd)) <= 1e-2 and not prm.fixed_iters:  # (fixed_iters: the throughput mode of lins_params)
            converged = True
        residual_norm = rn
    if diverged:
        state, cov = filt, pk
    else:  # SE:594-598
        ikh = np.eye(18) - kk @ hk
        cov = ikh @ pk @ ikh.T + kk @ ((prm.lidar_std * prm.lidar_std) * np.eye(len(hk))) @ kk.T
        cov = 0.5 * (cov + cov.T)
        state = lin
    return dict(state=state, cov=cov, iters=iters, converged=int(converged), diverged=int(diverged), m_surf=m_surf,
                m_corner=m_corner, trace=trace)
