"""Direct device tests of the small-math building blocks through lins_debug_math, against
the CPU oracle's own helpers: Quat2axis / axis2Quat / Rinvleft (MU:61-88, 304-321), GlobalState boxPlus / boxMinus
(KF:71-94), the kernels' sin/cos-free route to phi and Rinvleft(-phi)^T, and transformToStart (SE:1066-1080) —
including the branches the end-to-end tests never reach: |theta| < 1e-10, w < 0 (angle beyond pi, wrap_pi),
rotations close to pi."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg, ieskf):
    c = ieskf.IeskfContext(pkg.default_params(), max_batch=1, max_targets=1024)
    yield c
    c.close()


def dev(ieskf, ctx, op, x, n_out):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros((len(x), n_out))
    L = ieskf.lib()
    L.lins_debug_math.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.lins_debug_math.restype = C.c_int
    assert L.lins_debug_math(ctx._h, op, len(x), x.ctypes.data, x.shape[1], out.ctypes.data, n_out) == 0
    return out


def quats(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    # special cases: identity, tiny vector parts around the 1e-10 branch, w < 0, w = 0, rotation by ~pi
    extra = [[1, 0, 0, 0], [1, 3e-11, 0, 0], [1, 0, 2e-10, 0], [1, 9.99e-11, 0, 0], [-1, 1e-11, 0, 0], [-0.6, 0.8, 0, 0],
             [0, 1, 0, 0], [1e-9, 0.6, 0.8, 0], [-1e-9, 0, 0.6, 0.8], [0.5, -0.5, 0.5, -0.5]]
    return np.concatenate([np.array(extra, dtype=np.float64), q])


def vecs(rng, n):
    v = rng.normal(size=(n, 3)) * rng.choice([1e-12, 1e-6, 1e-2, 0.5, 2.0], size=(n, 1))
    extra = [[0, 0, 0], [5e-11, 0, 0], [0, 1.0001e-10, 0], [0, 0, 9.999e-11], [3.0, 0, 0], [0, 3.14159, 0], [0, 0, -3.2]]
    return np.concatenate([np.array(extra, dtype=np.float64), v])


def test_quat2axis_axis2quat_rinvleft(ieskf, oracle, ctx):
    rng = np.random.default_rng(11)
    q = quats(rng, 200)
    got = dev(ieskf, ctx, 0, q, 3)
    want = np.array([oracle.quat2axis(x) for x in q])
    assert np.abs(got - want).max() <= 1e-14  # (device atan2 vs glibc: last ulp of an angle <= 2 pi)
    v = vecs(rng, 200)
    got = dev(ieskf, ctx, 1, v, 4)
    want = np.array([oracle.axis2quat(x) for x in v])
    assert np.abs(got - want).max() <= 1e-15
    assert np.array_equal(got[0], [1, 0, 0, 0]) and np.array_equal(got[1], [1, 0, 0, 0]) and np.array_equal(got[3], [1, 0, 0, 0])
    assert got[2, 0] != 1.0 or got[2, 2] != 0.0  # |v| = 1.0001e-10 is past the branch: a real rotation
    got = dev(ieskf, ctx, 2, v, 9)
    want = np.array([oracle.rinvleft(x) for x in v]).reshape(-1, 9)
    scale = np.maximum(1.0, np.abs(want).max(axis=1, keepdims=True))
    assert (np.abs(got - want) / scale).max() <= 1e-13  # (cot(theta/2) near theta = 2 pi is large: relative)
    assert np.array_equal(got[0].reshape(3, 3), np.eye(3)) and np.array_equal(got[1].reshape(3, 3), np.eye(3))


def test_box_plus_box_minus(ieskf, oracle, ctx):
    rng = np.random.default_rng(12)
    n = 200
    q = quats(rng, n - 10)
    s = rng.normal(size=(n, 19))
    s[:, 6:10] = q
    dx = rng.normal(size=(n, 18)) * rng.choice([1e-12, 1e-4, 0.1, 1.5], size=(n, 1))
    dx[:3, 6:9] = [[0, 0, 0], [5e-11, 0, 0], [0, 0, 2e-10]]
    got = dev(ieskf, ctx, 3, np.concatenate([s, dx], axis=1), 19)
    want = np.array([oracle.box_plus(a, b) for a, b in zip(s, dx)])
    assert np.abs(got - want).max() <= 1e-14
    assert np.abs(np.linalg.norm(got[:, 6:10], axis=1) - 1).max() <= 1e-15  # boxPlus normalises (KF:78)
    b = rng.normal(size=(n, 19))
    b[:, 6:10] = quats(rng, n - 10)
    b[:5, 6:10] = s[:5, 6:10]  # equal attitudes: Quat2axis of the identity (the < 1e-10 branch)
    got = dev(ieskf, ctx, 4, np.concatenate([s, b], axis=1), 18)
    want = np.array([oracle.box_minus(x, y) for x, y in zip(s, b)])
    assert np.abs(got - want).max() <= 1e-13
    # (x [+] d) [-] x = d on the device alone, for increments below pi
    d_small = rng.normal(size=(n, 18)) * 0.3
    plus = dev(ieskf, ctx, 3, np.concatenate([s, d_small], axis=1), 19)
    back = dev(ieskf, ctx, 4, np.concatenate([plus, s], axis=1), 18)
    assert np.abs(back - d_small).max() <= 1e-12


def test_kernel_iteration_constants_route(ieskf, oracle, ctx):
    """phi_and_Gt (what the persistent kernels really run between iterations: h cot h from the quaternion's own
    half-angle, no sin / cos) against the reference's formulas Quat2axis + Rinvleft(-phi)."""
    rng = np.random.default_rng(13)
    q = quats(rng, 300)
    got = dev(ieskf, ctx, 5, q, 12)
    phi = np.array([oracle.quat2axis(x) for x in q])
    assert np.abs(got[:, :3] - phi).max() <= 1e-14
    gt = np.array([oracle.rinvleft(-p).T for p in phi]).reshape(-1, 9)
    scale = np.maximum(1.0, np.abs(gt).max(axis=1, keepdims=True))
    assert (np.abs(got[:, 3:] - gt) / scale).max() <= 1e-12


def test_transform_to_start(pkg, ieskf, oracle, ctx):
    rng = np.random.default_rng(14)
    n = 400
    prm = pkg.default_params()
    lin = rng.normal(size=(n, 19))
    lin[:, 6:10] = quats(rng, n - 10)
    lin[:10, 6:10] = [1, 0, 0, 0]  # no rotation at all: the theta < 1e-10 branch of axis2Quat(s * phi)
    pts = np.zeros((n, 4), dtype=np.float32)
    pts[:, :3] = rng.normal(size=(n, 3)) * 20
    pts[:, 3] = rng.integers(0, 16, size=n) + rng.uniform(-0.02, 0.12, size=n)  # ring + 0.1 * relTime (SE:649-650)
    pts[:5, 3] = [0.0, 3.0, -0.01, 15.1, 7.05]
    x = np.concatenate([lin, pts.astype(np.float64), np.full((n, 1), prm.scan_period)], axis=1)
    got = dev(ieskf, ctx, 6, x, 3).astype(np.float32)
    want = np.array([oracle.transform_to_start(prm, l, p)[0, :3] for l, p in zip(lin, pts)], dtype=np.float32)
    ulp = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64))
    assert ulp.max() <= 1 and (ulp > 0).mean() <= 1e-3  # (f64 sin / cos of ocml vs glibc under an f32 rounding)


def test_wave_solve6_is_bit_identical_to_the_one_lane_elimination(ieskf, ctx):
    """The 6 x 6 pivoted elimination spread over a wave (ieskf_rowsum.h wave_solve6) performs, element by element,
    the operations of the one-lane reg_solve6: same pivots, same roundings, the same bits — also on systems that
    force row exchanges, and with a NaN in the system (the divergence test of SE:552-563 must see the same NaNs)."""
    rng = np.random.default_rng(21)
    sys_ = rng.normal(size=(300, 6, 7))
    sys_[:100, np.arange(6), np.arange(6)] += 8.0        # diagonally dominant: no exchanges
    sys_[100:150, 0, 0] = 1e-12                            # forces pivoting at the first step
    sys_[150:160, 2, :] = sys_[150:160, 1, :]              # singular: inf / nan results, still the same ones
    sys_[160, 3, 4] = np.nan
    x = sys_.reshape(300, 42)
    a = dev(ieskf, ctx, 7, x, 6)
    b = dev(ieskf, ctx, 8, x, 6)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint64), b[~np.isnan(b)].view(np.uint64))
    ok = np.isfinite(a).all(axis=1)
    res = np.einsum("nij,nj->ni", sys_[ok][:, :, :6], a[ok]) - sys_[ok][:, :, 6]
    assert np.abs(res[:100]).max() <= 1e-12


def small_quats(rng, n):
    """unit quaternions of SMALL rotations (w > 0, tan(angle / 2) <= 1/8: the short-series range) and of rotations
    just outside it, plus the 1e-10 branch"""
    ang = rng.uniform(0, 0.3, size=n) * rng.choice([1e-6, 1e-3, 0.1, 1.0], size=n)
    ax = rng.normal(size=(n, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    q = np.concatenate([np.cos(ang / 2)[:, None], ax * np.sin(ang / 2)[:, None]], axis=1)
    extra = [[1, 0, 0, 0], [1, 3e-11, 0, 0], [1, 0, 1.0001e-10, 0], [1, 0.124, 0, 0], [1, 0.1251, 0, 0], [2.0, 0.02, -0.04, 0.06]]
    return np.concatenate([np.array(extra, dtype=np.float64), q])


def test_short_series_forms_of_the_serial_tail(ieskf, oracle, ctx):
    """axis2quat_fast / quat2axis_fast / the small-rotation branch of phi_and_Gt (lins_math.h): what the wave that
    walks from one iteration to the next really evaluates, against the oracle's libm formulas (MU:61-88, 304-321)
    — inside the series' range, just outside it (the libm fallback), and across the 1e-10 branches."""
    rng = np.random.default_rng(31)
    v = np.concatenate([vecs(rng, 200), rng.normal(size=(200, 3)) * 0.2])
    got = dev(ieskf, ctx, 13, v, 4)
    want = np.array([oracle.axis2quat(x) for x in v])
    assert np.abs(got - want).max() <= 1e-15
    assert np.array_equal(got[0], [1, 0, 0, 0]) and np.array_equal(got[1], [1, 0, 0, 0]) and np.array_equal(got[3], [1, 0, 0, 0])
    q = np.concatenate([quats(rng, 100), small_quats(rng, 400)])
    got = dev(ieskf, ctx, 14, q, 3)
    want = np.array([oracle.quat2axis(x) for x in q])
    assert np.abs(got - want).max() <= 1e-14
    got = dev(ieskf, ctx, 5, q, 12)
    ref = dev(ieskf, ctx, 15, q, 12)  # the same constants by the general (atan2) route, on the device
    assert np.abs(got[:, :3] - want).max() <= 1e-14
    gt = np.array([oracle.rinvleft(-p).T for p in want]).reshape(-1, 9)
    scale = np.maximum(1.0, np.abs(gt).max(axis=1, keepdims=True))
    assert (np.abs(got[:, 3:] - gt) / scale).max() <= 1e-12
    small = (q[:, 0] > 0) & ((q[:, 1:] ** 2).sum(axis=1) <= q[:, 0] ** 2 / 64) & ((q[:, 1:] ** 2).sum(axis=1) >= 1e-20)
    assert small.sum() > 200 and (~small).sum() > 100
    assert np.array_equal(got[~small], ref[~small])                # outside the range: the general route itself
    assert np.abs(got[small] - ref[small]).max() <= 2e-15          # inside: the two routes agree to rounding


def test_gauss_jordan_solve_of_the_kernels(ieskf, ctx):
    """wave_gj_solve6 (ieskf_rowsum.h, what the kernels run) returns the bits of the scalar definition gj_solve6
    (lins_solve6.h) — row exchanges, singular systems and a NaN included — and solves: against numpy, and against
    round 1's elimination + back-substitution within the conditioning."""
    rng = np.random.default_rng(22)
    sys_ = rng.normal(size=(400, 6, 7))
    sys_[:100, np.arange(6), np.arange(6)] += 8.0
    sys_[100:150, 0, 0] = 1e-12
    sys_[150:160, 2, :] = sys_[150:160, 1, :]
    sys_[160, 3, 4] = np.nan
    sys_[161:200] *= 10.0 ** rng.integers(-6, 7, size=(39, 1, 1))
    sys_[200:230, 1, 1] = 0.0
    x = sys_.reshape(400, 42)
    a = dev(ieskf, ctx, 11, x, 6)
    b = dev(ieskf, ctx, 12, x, 6)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(np.uint64), b[~np.isnan(b)].view(np.uint64))
    assert np.isnan(b[160]).any()  # a NaN in the system reaches the solution (SE:552-563 must see it)
    old = dev(ieskf, ctx, 8, x, 6)
    ok = np.isfinite(b).all(axis=1) & np.isfinite(old).all(axis=1)
    ok[150:161] = False
    want = np.stack([np.linalg.solve(m[:, :6], m[:, 6]) for m in sys_[ok]])
    cond = np.array([np.linalg.cond(m[:, :6]) for m in sys_[ok]])
    tol = 4e-15 * np.maximum(cond, 10.0) * np.abs(want).max(axis=1)
    assert (np.abs(b[ok] - want).max(axis=1) <= tol).all()
    assert (np.abs(b[ok] - old[ok]).max(axis=1) <= 2 * tol).all()


def test_row_reduction_on_the_valu_lane_paths_matches_the_shuffle_tree(ieskf, ctx):
    """wave_reduce_rows exchanges lanes with DPP / lane-swap instructions (ieskf_rowsum.h xor_lane_i32); the same
    tree written with __shfl_xor must give the same bits, and both the 28 sums of the 64 rows."""
    rng = np.random.default_rng(5)
    rows = rng.normal(size=(50, 64, 7)) * 10.0 ** rng.integers(-3, 4, size=(50, 64, 1))
    x = rows.reshape(50, 448)
    a = dev(ieskf, ctx, 9, x, 28)
    b = dev(ieskf, ctx, 10, x, 28)
    assert np.array_equal(a.view(np.uint64), b.view(np.uint64))
    A = [0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 0, 1, 2, 3, 4, 5, 6]
    B = [0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5, 6, 6, 6, 6, 6, 6, 6]
    want = np.stack([(rows[:, :, i] * rows[:, :, j]).sum(axis=1) for i, j in zip(A, B)], axis=1)
    scale = np.stack([np.abs(rows[:, :, i] * rows[:, :, j]).sum(axis=1) for i, j in zip(A, B)], axis=1)
    assert (np.abs(a - want) <= 1e-13 * scale).all()


def test_icp_gauss_newton_step_over_a_wave_is_bit_identical_to_the_one_lane_definition(ieskf, ctx):
    """icp_wave.h (a matrix column per lane: what the ICP kernel runs) against icp_math.h's icp_gn_solve run by one
    lane (the definition, shared with the host shim): column-pivoted Householder QR with Eigen's rank rule, and on round 0
    the Jacobi eigen-decomposition + degenerate-direction projection (SE:1264-1302) — the same bits on well-conditioned
    normal equations, on systems that force column exchanges, on rank-deficient ones and on ones with eigenvalues
    below the degeneracy threshold (10)."""
    rng = np.random.default_rng(77)
    n = 400
    J = rng.normal(size=(n, 40, 6)) * rng.choice([0.3, 1.0, 5.0], size=(n, 1, 6))
    J[100:160, :, 2] *= 1e-3            # a weak direction: eigenvalue < 10 -> projection on round 0
    J[160:200, :, 4] = J[160:200, :, 1]  # exactly rank-deficient (two equal columns)
    J[200:220, :, 0] = 0.0               # a zero column
    b = rng.normal(size=(n, 40)) * 0.05
    A = np.einsum("nki,nkj->nij", J, J)
    g = np.einsum("nki,nk->ni", J, b)
    A[220:230] *= 1e-6                   # every eigenvalue below the threshold
    rounds = np.where(np.arange(n) % 2 == 0, 0.0, 3.0)
    x = np.concatenate([A.reshape(n, 36), g, rounds[:, None]], axis=1)
    one = dev(ieskf, ctx, 16, x, 6)
    wav = dev(ieskf, ctx, 17, x, 6)
    assert np.array_equal(np.isnan(one), np.isnan(wav))
    assert np.array_equal(one[~np.isnan(one)].view(np.uint64), wav[~np.isnan(wav)].view(np.uint64))
    # and the definition does what it says on the plain systems: later rounds solve the normal equations
    ok = (rounds == 3.0) & (np.arange(n) < 100)
    sol = np.linalg.solve(A[ok], g[ok][:, :, None])[:, :, 0]
    assert np.abs(one[ok] - sol).max() <= 1e-9 * np.abs(sol).max()


def test_lm_step_over_a_wave_is_bit_identical_to_the_one_thread_definition(ieskf, ctx):
    """lm_wave.h (element (i, j) of the 6 x 6 matrices in lane 6 i + j: what map_lm_kernel runs) against lm_math.h's
    lm_step_from_sums run by one thread (the definition, shared with the host) — Householder QR every round; on round 0
    the cyclic-Jacobi eigen-decomposition, the Gauss-Jordan inverse with row exchanges, the degeneracy projection
    (LM:1583-1632): the same bits of the transform, the stop flag, isDegenerate and matP on well-conditioned normal
    equations, on ones with eigenvalues below the threshold (100), on rank-deficient ones, with fewer than 50 rows, and
    on later rounds that apply a carried projection."""
    rng = np.random.default_rng(91)
    n = 480
    J = (rng.normal(size=(n, 300, 6)) * rng.choice([1.0, 2.0, 4.0], size=(n, 1, 6))).astype(np.float32).astype(np.float64)
    J[100:180, :, 2] *= 1e-2             # a weak direction: eigenvalue < 100 -> projection
    J[180:220, :, 4] = J[180:220, :, 1]   # exactly rank-deficient (two equal columns)
    J[220:240, :, 0] = 0.0                # a zero column
    b = rng.normal(size=(n, 300)) * 0.02
    A = np.einsum("nki,nkj->nij", J, J)
    g = np.einsum("nki,nk->ni", J, b)
    A[240:260] *= 1e-4                    # every eigenvalue below the threshold
    iu = np.triu_indices(6)
    sums = np.zeros((n, 28))
    sums[:, :21] = A[:, iu[0], iu[1]]
    sums[:, 21:27] = g
    sums[:, 27] = 300
    sums[260:270, 27] = 49                # too few rows: nothing happens (LM:1530)
    rounds = np.where(np.arange(n) % 3 == 0, 0.0, np.arange(n) % 3 + 1.0)
    T0 = rng.normal(size=(n, 6)) * 0.1
    deg_in = (rng.random(n) < 0.5).astype(np.float64)  # carried isDegenerate / matP of the later rounds
    P_in = (np.eye(6)[None] + rng.normal(size=(n, 6, 6)) * 0.2).reshape(n, 36)
    x = np.concatenate([sums, rounds[:, None], T0, deg_in[:, None], P_in], axis=1)
    one = dev(ieskf, ctx, 18, x, 44)
    wav = dev(ieskf, ctx, 19, x, 44)
    assert np.array_equal(np.isnan(one), np.isnan(wav))
    assert np.array_equal(one[~np.isnan(one)].view(np.uint64), wav[~np.isnan(wav)].view(np.uint64))
    r0 = rounds == 0
    assert one[r0 & (np.arange(n) >= 100) & (np.arange(n) < 260), 7].all() and not one[r0 & (np.arange(n) < 100), 7].any()
    # and the definition does what it says: a well-conditioned later round without a projection solves the normal equations
    ok = (~r0) & (deg_in == 0) & (np.arange(n) < 100)
    sol = np.linalg.solve(A[ok], g[ok][:, :, None])[:, :, 0]
    got = one[ok, 1:7] - T0[ok].astype(np.float32)
    assert np.abs(got - sol).max() <= 2e-3 * np.abs(sol).max()
