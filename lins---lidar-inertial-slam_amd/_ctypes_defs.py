"""ctypes mirrors of the PODs in include/lins_ieskf.h and include/lins_host.h.

Shared by the product bindings (ieskf.py, host.py) and by the oracle's test
binding (oracle/oracle.py) — it describes the C ABI only, no behaviour.
"""
import ctypes as C

import numpy as np

STATE_DIM = 19
ERR_DIM = 18
MAX_QUERY = 1024
CLOUD_MAX = 16 * 1800

LINS_OK = 0


class Point(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("intensity", C.c_float)]


class Params(C.Structure):
    _fields_ = [
        ("num_iter", C.c_int32),
        ("icp_freq", C.c_int32),
        ("fixed_iters", C.c_int32),
        ("reserved", C.c_int32),
        ("lidar_std", C.c_double),
        ("lidar_scale", C.c_double),
        ("nearest_sq_dist", C.c_double),
        ("scan_period", C.c_double),
    ]


def default_params(num_iter=30, fixed_iters=0, icp_freq=1):
    """exp_port.yaml:11-20 values."""
    return Params(num_iter, icp_freq, fixed_iters, 0, 0.01, 1.0, 25.0, 0.1)


class ScanPairC(C.Structure):
    _fields_ = [
        ("surf_flat", C.POINTER(Point)),
        ("corner_sharp", C.POINTER(Point)),
        ("surf_less_flat_last", C.POINTER(Point)),
        ("corner_less_sharp_last", C.POINTER(Point)),
        ("n_surf_flat", C.c_int32),
        ("n_corner_sharp", C.c_int32),
        ("n_surf_last", C.c_int32),
        ("n_corner_last", C.c_int32),
        ("point_stride_bytes", C.c_int32),  # 0 / 16: packed points; 32: pcl::PointXYZI arrays
        ("reserved", C.c_int32),
        ("state", C.c_double * STATE_DIM),
        ("cov", C.c_double * (ERR_DIM * ERR_DIM)),
    ]


class ResultC(C.Structure):
    _fields_ = [
        ("state", C.c_double * STATE_DIM),
        ("cov", C.c_double * (ERR_DIM * ERR_DIM)),
        ("residual_norm", C.c_double),
        ("update_norm", C.c_double),
        ("iters", C.c_int32),
        ("converged", C.c_int32),
        ("diverged", C.c_int32),
        ("m_surf", C.c_int32),
        ("m_corner", C.c_int32),
        ("reserved", C.c_int32 * 3),
    ]


class PoseRecordC(C.Structure):
    _fields_ = [
        ("state", C.c_double * STATE_DIM),
        ("residual_norm", C.c_double),
        ("iters", C.c_int32),
        ("converged", C.c_int32),
        ("diverged", C.c_int32),
        ("m_surf", C.c_int32),
        ("m_corner", C.c_int32),
        ("scan_id", C.c_int32),
        ("pad", C.c_int32 * 2),
    ]


assert C.sizeof(PoseRecordC) == 192

CORR_DTYPE = np.dtype(
    [("ind1", "<i4"), ("ind2", "<i4"), ("ind3", "<i4"), ("accepted", "<i4"), ("coeff", "<f4", (4,)), ("sel", "<f4", (4,))]
)
assert CORR_DTYPE.itemsize == 48

POSE_DTYPE = np.dtype(
    [
        ("state", "<f8", (STATE_DIM,)),
        ("residual_norm", "<f8"),
        ("iters", "<i4"),
        ("converged", "<i4"),
        ("diverged", "<i4"),
        ("m_surf", "<i4"),
        ("m_corner", "<i4"),
        ("scan_id", "<i4"),
        ("pad", "<i4", (2,)),
    ]
)
assert POSE_DTYPE.itemsize == 192


def _pts(a):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4)
    return a


class ScanPair:
    """One IESKF problem: queries, targets, prior state + covariance."""

    def __init__(self, surf_flat, corner_sharp, surf_last, corner_last, state, cov, meta=None):
        self.surf_flat = _pts(surf_flat)
        self.corner_sharp = _pts(corner_sharp)
        self.surf_last = _pts(surf_last)
        self.corner_last = _pts(corner_last)
        self.state = np.ascontiguousarray(state, dtype=np.float64).reshape(STATE_DIM)
        self.cov = np.ascontiguousarray(cov, dtype=np.float64).reshape(ERR_DIM, ERR_DIM)
        self.meta = meta or {}

    def sizes(self):
        return (len(self.corner_sharp), len(self.surf_flat), len(self.corner_last), len(self.surf_last))

    def bytes_per_iter(self):
        """Algorithmic bytes of one iteration."""
        return 16 * sum(self.sizes()) + 8 * 19 + 8 * 28

    def as_c(self):
        c = ScanPairC()
        self.fill_c(c)
        return c

    def fill_c(self, c):
        c.surf_flat = self.surf_flat.ctypes.data_as(C.POINTER(Point))
        c.corner_sharp = self.corner_sharp.ctypes.data_as(C.POINTER(Point))
        c.surf_less_flat_last = self.surf_last.ctypes.data_as(C.POINTER(Point))
        c.corner_less_sharp_last = self.corner_last.ctypes.data_as(C.POINTER(Point))
        c.n_surf_flat = len(self.surf_flat)
        c.n_corner_sharp = len(self.corner_sharp)
        c.n_surf_last = len(self.surf_last)
        c.n_corner_last = len(self.corner_last)
        C.memmove(c.state, self.state.ctypes.data, 8 * STATE_DIM)
        C.memmove(c.cov, self.cov.ctypes.data, 8 * ERR_DIM * ERR_DIM)


def pairs_to_c(pairs):
    arr = (ScanPairC * len(pairs))()
    for i, p in enumerate(pairs):
        p.fill_c(arr[i])
    return arr


def pairs_strided(pairs):
    """The same pairs with their clouds as pcl::PointXYZI lays them out (32 bytes a point: x, y, z, pad, intensity, 3 pads;
    parameters.h:52) and point_stride_bytes = 32: what a lins_fusion_node passes without repacking.  Returns the ScanPairC
    array and the arrays that own the memory (keep them alive)."""
    arr = pairs_to_c(pairs)
    keep = []
    for i, p in enumerate(pairs):
        for name, src in (("surf_flat", p.surf_flat), ("corner_sharp", p.corner_sharp), ("surf_less_flat_last", p.surf_last),
                          ("corner_less_sharp_last", p.corner_last)):
            wide = np.full((len(src), 8), np.float32(-77.0))  # (pads hold a value no cloud has: read by mistake, it shows)
            wide[:, 0:3], wide[:, 4] = src[:, 0:3], src[:, 3]
            keep.append(wide)
            setattr(arr[i], name, wide.ctypes.data_as(C.POINTER(Point)))
        arr[i].point_stride_bytes = 32
    return arr, keep


class Result:
    def __init__(self, rc):
        self.state = np.array(rc.state[:], dtype=np.float64)
        self.cov = np.array(rc.cov[:], dtype=np.float64).reshape(ERR_DIM, ERR_DIM)
        self.residual_norm = rc.residual_norm
        self.update_norm = rc.update_norm
        self.iters = rc.iters
        self.converged = rc.converged
        self.diverged = rc.diverged
        self.m_surf = rc.m_surf
        self.m_corner = rc.m_corner
        self.reserved = tuple(rc.reserved[:])  # debug counters of the kernels (certificate / exhaustive-search statistics)

    def __repr__(self):
        return (
            f"Result(iters={self.iters}, conv={self.converged}, div={self.diverged}, "
            f"m=({self.m_surf},{self.m_corner}), p={self.state[:3]}, |r|={self.residual_norm:.4g})"
        )


# ---- scan-to-map row (include/lins_map.h) -------------------------------------------------
class MapProblemC(C.Structure):
    _fields_ = [("map_corner", C.POINTER(Point)), ("map_surf", C.POINTER(Point)), ("scan_corner", C.POINTER(Point)),
                ("scan_surf", C.POINTER(Point)), ("n_map_corner", C.c_int32), ("n_map_surf", C.c_int32),
                ("n_scan_corner", C.c_int32), ("n_scan_surf", C.c_int32), ("transform", C.c_float * 6),
                ("reserved", C.c_int32 * 2)]


class MapResultC(C.Structure):
    _fields_ = [("transform", C.c_float * 6), ("iters", C.c_int32), ("converged", C.c_int32),
                ("degenerate", C.c_int32), ("n_sel", C.c_int32)]


MAP_CORR_DTYPE = np.dtype([("ind", np.int32, 5), ("accepted", np.int32), ("coeff", np.float32, 4),
                           ("sel", np.float32, 3), ("sq5", np.float32)])


class MapProblem:
    """numpy-owned scan-to-map problem; as_c() gives the ctypes view."""

    def __init__(self, map_corner, map_surf, scan_corner, scan_surf, transform):
        f = lambda a: np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 4)
        self.map_corner, self.map_surf, self.scan_corner, self.scan_surf = f(map_corner), f(map_surf), f(scan_corner), f(scan_surf)
        self.transform = np.asarray(transform, dtype=np.float32).copy()
        self.reuse_resident_map = False  # LINS_MAP_REUSE: the maps are the previous call's (lins_map.h)

    def as_c(self):
        c = MapProblemC()
        c.reserved[0] = 1 if self.reuse_resident_map else 0
        pp = lambda a: a.ctypes.data_as(C.POINTER(Point))
        c.map_corner, c.map_surf, c.scan_corner, c.scan_surf = pp(self.map_corner), pp(self.map_surf), pp(self.scan_corner), pp(self.scan_surf)
        c.n_map_corner, c.n_map_surf = len(self.map_corner), len(self.map_surf)
        c.n_scan_corner, c.n_scan_surf = len(self.scan_corner), len(self.scan_surf)
        c.transform[:] = [float(v) for v in self.transform]
        return c
