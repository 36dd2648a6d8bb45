"""ctypes binding of the CPU oracle (oracle/liblins_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the product package.
Pinned against oracle/_ref — the reference's own headers compiled verbatim — by tests/test_ref.py (see lins_oracle.h).
"""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_defs = importlib.import_module("lins---lidar-inertial-slam_amd._ctypes_defs")
Params, ScanPairC, ResultC, Result, Point = _defs.Params, _defs.ScanPairC, _defs.ResultC, _defs.Result, _defs.Point
CORR_DTYPE = _defs.CORR_DTYPE

FORM_DENSE, FORM_REDUCED = 0, 1
NN_KDTREE, NN_BRUTE = 0, 1

_LIB = None


class TraceC(C.Structure):
    _fields_ = [
        ("max_iters", C.c_int32),
        ("surf", C.c_void_p),
        ("corner", C.c_void_p),
        ("lin_state", C.POINTER(C.c_double)),
        ("dx", C.POINTER(C.c_double)),
        ("sums28", C.POINTER(C.c_double)),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        p = os.path.join(_HERE, "liblins_oracle.so")
        if not os.path.exists(p):
            build()
        L = C.CDLL(p)
        dp = C.POINTER(C.c_double)
        L.oracle_correspondences.argtypes = [C.POINTER(Params), C.POINTER(ScanPairC), dp, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p]
        L.oracle_ieskf.argtypes = [C.POINTER(Params), C.POINTER(ScanPairC), C.c_int, C.c_int, C.POINTER(ResultC),
                                   C.POINTER(TraceC)]
        L.oracle_icp.argtypes = [C.POINTER(Params), C.POINTER(ScanPairC), dp, dp, C.c_int, C.POINTER(C.c_int32)]
        L.oracle_perform_ieskf.argtypes = [C.POINTER(Params), C.POINTER(ScanPairC), C.c_int, C.c_int,
                                           C.POINTER(ResultC)]
        L.oracle_nn.argtypes = [C.POINTER(Point), C.c_int, C.POINTER(Point), C.c_int, C.c_int,
                                C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        L.oracle_bench.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(ScanPairC), C.c_int, C.c_int, C.c_int,
                                   dp, C.POINTER(C.c_uint64)]
        for f in (L.oracle_correspondences, L.oracle_ieskf, L.oracle_icp, L.oracle_perform_ieskf, L.oracle_nn,
                  L.oracle_bench):
            f.restype = C.c_int
        L.oracle_quat2axis.argtypes = [dp, dp]
        L.oracle_axis2quat.argtypes = [dp, dp]
        L.oracle_rinvleft.argtypes = [dp, dp]
        L.oracle_box_plus.argtypes = [dp, dp, dp]
        L.oracle_box_minus.argtypes = [dp, dp, dp]
        L.oracle_transform_to_start.argtypes = [C.POINTER(Params), dp, C.POINTER(Point), C.POINTER(Point)]
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _vecfn(name, x, nout):
    x = np.ascontiguousarray(x, dtype=np.float64)
    o = np.zeros(nout)
    getattr(lib(), name)(_d(x), _d(o))
    return o


def quat2axis(q):
    return _vecfn("oracle_quat2axis", q, 3)


def axis2quat(a):
    return _vecfn("oracle_axis2quat", a, 4)


def rinvleft(a):
    return _vecfn("oracle_rinvleft", a, 9).reshape(3, 3)


def box_plus(s, dx):
    s = np.ascontiguousarray(s, dtype=np.float64)
    dx = np.ascontiguousarray(dx, dtype=np.float64)
    o = np.zeros(19)
    lib().oracle_box_plus(_d(s), _d(dx), _d(o))
    return o


def box_minus(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    o = np.zeros(18)
    lib().oracle_box_minus(_d(a), _d(b), _d(o))
    return o


def transform_to_start(prm, lin_state, pts):
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 4)
    lin_state = np.ascontiguousarray(lin_state, dtype=np.float64)
    out = np.empty_like(pts)
    for i in range(len(pts)):
        lib().oracle_transform_to_start(C.byref(prm), _d(lin_state),
                                        pts[i:i + 1].ctypes.data_as(C.POINTER(Point)),
                                        out[i:i + 1].ctypes.data_as(C.POINTER(Point)))
    return out


def nn(targets, queries, mode=NN_BRUTE):
    t = np.ascontiguousarray(targets, dtype=np.float32).reshape(-1, 4)
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 4)
    idx = np.zeros(len(q), dtype=np.int32)
    d = np.zeros(len(q), dtype=np.float32)
    lib().oracle_nn(t.ctypes.data_as(C.POINTER(Point)), len(t), q.ctypes.data_as(C.POINTER(Point)), len(q), mode,
                    idx.ctypes.data_as(C.POINTER(C.c_int32)), d.ctypes.data_as(C.POINTER(C.c_float)))
    return idx, d


def correspondences(prm, pair, lin_state, it, mode=NN_BRUTE):
    c = pair.as_c()
    lin_state = np.ascontiguousarray(lin_state, dtype=np.float64)
    surf = np.zeros(c.n_surf_flat, dtype=CORR_DTYPE)
    corner = np.zeros(c.n_corner_sharp, dtype=CORR_DTYPE)
    rc = lib().oracle_correspondences(C.byref(prm), C.byref(c), _d(lin_state), it, mode, surf.ctypes.data,
                                      corner.ctypes.data)
    assert rc == 0, rc
    return surf, corner


def ieskf(prm, pair, form=FORM_DENSE, mode=NN_KDTREE, trace=False):
    c = pair.as_c()
    r = ResultC()
    if not trace:
        rc = lib().oracle_ieskf(C.byref(prm), C.byref(c), form, mode, C.byref(r), None)
        assert rc == 0, rc
        return Result(r)
    k = prm.num_iter
    tr = dict(surf=np.zeros((k, c.n_surf_flat), dtype=CORR_DTYPE), corner=np.zeros((k, c.n_corner_sharp), dtype=CORR_DTYPE),
              lin_state=np.zeros((k, 19)), dx=np.zeros((k, 18)), sums28=np.zeros((k, 28)))
    t = TraceC(k, tr["surf"].ctypes.data, tr["corner"].ctypes.data, _d(tr["lin_state"]), _d(tr["dx"]), _d(tr["sums28"]))
    rc = lib().oracle_ieskf(C.byref(prm), C.byref(c), form, mode, C.byref(r), C.byref(t))
    assert rc == 0, rc
    return Result(r), tr


def perform_ieskf(prm, pair, form=FORM_DENSE, mode=NN_KDTREE):
    c = pair.as_c()
    r = ResultC()
    rc = lib().oracle_perform_ieskf(C.byref(prm), C.byref(c), form, mode, C.byref(r))
    assert rc == 0, rc
    return Result(r)


def icp(prm, pair, t, q, mode=NN_KDTREE):
    c = pair.as_c()
    t = np.array(t, dtype=np.float64)
    q = np.array(q, dtype=np.float64)
    it = C.c_int32(0)
    rc = lib().oracle_icp(C.byref(prm), C.byref(c), _d(t), _d(q), mode, C.byref(it))
    assert rc == 0, rc
    return t, q, it.value


def bench(prm, pairs, form=FORM_DENSE, mode=NN_KDTREE, threads=1):
    arr = _defs.pairs_to_c(pairs)
    sec = C.c_double(0)
    its = C.c_uint64(0)
    rc = lib().oracle_bench(C.byref(prm), len(pairs), arr, form, mode, threads, C.byref(sec), C.byref(its))
    assert rc == 0, rc
    return sec.value, its.value


# ---- scan-to-map row (oracle/map_oracle.cpp) ------------------------------------------------
def map_correspondences(problem):
    c = problem.as_c()
    corner = np.zeros(len(problem.scan_corner), dtype=_defs.MAP_CORR_DTYPE)
    surf = np.zeros(len(problem.scan_surf), dtype=_defs.MAP_CORR_DTYPE)
    L = lib()
    L.oracle_map_correspondences.argtypes = [C.POINTER(_defs.MapProblemC), C.c_void_p, C.c_void_p]
    rc = L.oracle_map_correspondences(C.byref(c), corner.ctypes.data, surf.ctypes.data)
    assert rc == 0, rc
    return corner, surf


def scan2map(problem):
    c = problem.as_c()
    r = _defs.MapResultC()
    L = lib()
    L.oracle_scan2map.argtypes = [C.POINTER(_defs.MapProblemC), C.POINTER(_defs.MapResultC)]
    rc = L.oracle_scan2map(C.byref(c), C.byref(r))
    assert rc == 0, rc
    return dict(transform=np.array(r.transform[:], dtype=np.float32), iters=r.iters, converged=r.converged,
                degenerate=r.degenerate, n_sel=r.n_sel)


# ---- stages either side of the update (oracle/frontend_oracle.cpp: independent of csrc/, libm angles) ----------
CLOUD_MAX = 16 * 1800


def fe_segment(raw):
    """image_projection_node (IP:191-415): raw cloud (n, 4) float32 in firing order ->
    dict(cloud (m, 4), range, col, ground, start_ring, end_ring, orientation (start, end, diff), n_outlier, label)."""
    raw = np.ascontiguousarray(raw, dtype=np.float32).reshape(-1, 4)
    L = lib()
    cloud = np.zeros((CLOUD_MAX, 4), np.float32)
    rng = np.zeros(CLOUD_MAX, np.float32)
    col = np.zeros(CLOUD_MAX, np.uint32)
    ground = np.zeros(CLOUD_MAX, np.uint8)
    sr, er = np.zeros(16, np.int32), np.zeros(16, np.int32)
    ori = np.zeros(3, np.float32)
    nout = C.c_int32(0)
    label = np.zeros(CLOUD_MAX, np.int32)
    L.fo_segment.restype = C.c_int
    L.fo_segment.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 7 + [C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]
    m = L.fo_segment(raw.ctypes.data, len(raw), cloud.ctypes.data, rng.ctypes.data, col.ctypes.data, ground.ctypes.data,
                     sr.ctypes.data, er.ctypes.data, ori.ctypes.data, C.byref(nout), None, label.ctypes.data)
    assert m >= 0, m
    return dict(cloud=cloud, range=rng, col=col, ground=ground, n=m, start_ring=sr, end_ring=er, orientation=ori,
                n_outlier=nout.value, label=label.reshape(16, 1800))


def fe_features(seg, scan_period=0.1):
    """StateEstimator's feature stage (SE:619-827) on a segmented scan given as fe_segment's dict (arrays of
    CLOUD_MAX entries, the first n valid) -> dict(undistorted, corner_sharp, corner_less_sharp, surf_flat, surf_less_flat)."""
    L = lib()
    n = int(seg["n"])
    und = np.zeros((max(n, 1), 4), np.float32)
    bufs = [np.zeros((c, 4), np.float32) for c in (192, 1920, 1024, CLOUD_MAX)]
    counts = np.zeros(4, np.int32)
    arrs = [np.ascontiguousarray(seg[k]) for k in ("cloud", "range", "col", "ground")]
    sr = np.ascontiguousarray(seg["start_ring"], dtype=np.int32)
    er = np.ascontiguousarray(seg["end_ring"], dtype=np.int32)
    ori = np.ascontiguousarray(seg["orientation"], dtype=np.float32)
    L.fo_features.restype = C.c_int
    L.fo_features.argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3 + [C.c_double] + [C.c_void_p] * 6
    rc = L.fo_features(arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data, n, sr.ctypes.data,
                       er.ctypes.data, ori.ctypes.data, scan_period, und.ctypes.data, *[b.ctypes.data for b in bufs],
                       counts.ctypes.data)
    assert rc == 0, rc
    names = ("corner_sharp", "corner_less_sharp", "surf_flat", "surf_less_flat")
    out = {k: b[:c].copy() for k, b, c in zip(names, bufs, counts)}
    out["undistorted"] = und[:n]
    return out


def fe_transform_to_end(t, q, pts, scan_period=0.1):
    """transformToEnd (SE:1083-1101) of every point with pose (t, q = w x y z)."""
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 4)
    out = np.zeros_like(pts)
    t = np.ascontiguousarray(t, dtype=np.float64)
    q = np.ascontiguousarray(q, dtype=np.float64)
    L = lib()
    L.fo_transform_to_end.restype = None
    L.fo_transform_to_end.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p]
    L.fo_transform_to_end(t.ctypes.data, q.ctypes.data, scan_period, pts.ctypes.data, len(pts), out.ctypes.data)
    return out
