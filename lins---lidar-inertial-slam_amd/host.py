"""Bindings for liblins_host.so (pure-CPU host pieces, include/lins_host.h):
StatePredictor mirror, feature front-end, transformToEnd, synthetic scan pairs.
"""
import ctypes as C
import os

import numpy as np

from ._ctypes_defs import (CLOUD_MAX, ERR_DIM, MAX_QUERY, STATE_DIM, Point, ScanPair)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SYNTH_SEED = 0x4C494E53  # "LINS"


class FilterParams(C.Structure):
    _fields_ = [
        ("acc_n", C.c_double), ("gyr_n", C.c_double), ("acc_w", C.c_double), ("gyr_w", C.c_double),
        ("init_pos_std", C.c_double * 3), ("init_vel_std", C.c_double * 3), ("init_att_std", C.c_double * 3),
        ("init_acc_std", C.c_double * 3), ("init_gyr_std", C.c_double * 3),
    ]


class Filter(C.Structure):
    _fields_ = [
        ("state", C.c_double * STATE_DIM),
        ("cov", C.c_double * (ERR_DIM * ERR_DIM)),
        ("noise", C.c_double * 144),
        ("acc_last", C.c_double * 3), ("gyr_last", C.c_double * 3),
        ("time", C.c_double),
        ("has_imu", C.c_int32), ("pad", C.c_int32),
        ("prm", FilterParams),
    ]


class Features(C.Structure):
    _fields_ = [
        ("corner_sharp", C.POINTER(Point)), ("n_corner_sharp", C.c_int32),
        ("corner_less_sharp", C.POINTER(Point)), ("n_corner_less_sharp", C.c_int32),
        ("surf_flat", C.POINTER(Point)), ("n_surf_flat", C.c_int32),
        ("surf_less_flat", C.POINTER(Point)), ("n_surf_less_flat", C.c_int32),
        ("n_segmented", C.c_int32), ("n_outlier", C.c_int32),
    ]


class SegmentedScanC(C.Structure):
    """lins_segmented_scan: segmented cloud + cloud_msgs/cloud_info (include/lins_host.h)."""
    _fields_ = [("cloud", C.POINTER(Point)), ("range", C.POINTER(C.c_float)), ("col", C.POINTER(C.c_uint32)),
                ("ground", C.POINTER(C.c_uint8)), ("n", C.c_int32), ("start_ring", C.c_int32 * 16),
                ("end_ring", C.c_int32 * 16), ("start_ori", C.c_float), ("end_ori", C.c_float),
                ("ori_diff", C.c_float), ("n_outlier", C.c_int32)]


class Segmented:
    """numpy-owned segmented scan; .c is the ctypes view (pointers into the arrays kept alive here)."""

    def __init__(self, cloud, rng, col, ground, c):
        self.cloud, self.range, self.col, self.ground, self.c = cloud, rng, col, ground, c

    @property
    def n(self):
        return self.c.n


def _features_buffers():
    cs, pcs = _buf(192)
    cls, pcls = _buf(1920)
    sf, psf = _buf(MAX_QUERY)
    slf, pslf = _buf(CLOUD_MAX)
    f = Features()
    f.corner_sharp, f.corner_less_sharp, f.surf_flat, f.surf_less_flat = pcs, pcls, psf, pslf
    return f, (cs, cls, sf, slf)


def _features_dict(f, bufs):
    cs, cls, sf, slf = bufs
    return dict(corner_sharp=cs[: f.n_corner_sharp].copy(), corner_less_sharp=cls[: f.n_corner_less_sharp].copy(),
                surf_flat=sf[: f.n_surf_flat].copy(), surf_less_flat=slf[: f.n_surf_less_flat].copy(),
                n_segmented=f.n_segmented, n_outlier=f.n_outlier)


def frontend_segment(raw):
    """image_projection_node on the host: raw cloud -> segmented scan (lins_frontend_segment)."""
    raw = np.ascontiguousarray(raw, dtype=np.float32).reshape(-1, 4)
    cloud = np.zeros((CLOUD_MAX, 4), np.float32)
    rng = np.zeros(CLOUD_MAX, np.float32)
    col = np.zeros(CLOUD_MAX, np.uint32)
    ground = np.zeros(CLOUD_MAX, np.uint8)
    c = SegmentedScanC()
    L = lib()
    L.lins_frontend_segment.argtypes = [C.POINTER(Point), C.c_int, C.POINTER(Point), C.POINTER(C.c_float),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(SegmentedScanC)]
    rc = L.lins_frontend_segment(raw.ctypes.data_as(C.POINTER(Point)), len(raw), cloud.ctypes.data_as(C.POINTER(Point)),
                                 rng.ctypes.data_as(C.POINTER(C.c_float)), col.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 ground.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(c))
    if rc != 0:
        raise RuntimeError(f"lins_frontend_segment failed: {rc}")
    return Segmented(cloud, rng, col, ground, c)


def segmented_from_arrays(cloud, rng, col, ground, n, start_ring, end_ring, orientation, n_outlier=0):
    """A Segmented (lins_segmented_scan view) over caller-provided arrays — e.g. another implementation's output."""
    cloud = np.ascontiguousarray(cloud, dtype=np.float32).reshape(-1, 4)
    rng = np.ascontiguousarray(rng, dtype=np.float32)
    col = np.ascontiguousarray(col, dtype=np.uint32)
    ground = np.ascontiguousarray(ground, dtype=np.uint8)
    c = SegmentedScanC()
    c.cloud = cloud.ctypes.data_as(C.POINTER(Point))
    c.range = rng.ctypes.data_as(C.POINTER(C.c_float))
    c.col = col.ctypes.data_as(C.POINTER(C.c_uint32))
    c.ground = ground.ctypes.data_as(C.POINTER(C.c_uint8))
    c.n = int(n)
    for k in range(16):
        c.start_ring[k], c.end_ring[k] = int(start_ring[k]), int(end_ring[k])
    c.start_ori, c.end_ori, c.ori_diff = float(orientation[0]), float(orientation[1]), float(orientation[2])
    c.n_outlier = int(n_outlier)
    return Segmented(cloud, rng, col, ground, c)


def frontend_extract_segmented(seg, scan_period=0.1):
    """StateEstimator's feature stage on the host (the CPU restatement of the device front-end)."""
    f, bufs = _features_buffers()
    L = lib()
    L.lins_frontend_extract_segmented.argtypes = [C.POINTER(SegmentedScanC), C.c_double, C.POINTER(Features)]
    rc = L.lins_frontend_extract_segmented(C.byref(seg.c), scan_period, C.byref(f))
    if rc != 0:
        raise RuntimeError(f"lins_frontend_extract_segmented failed: {rc}")
    return _features_dict(f, bufs)


class SynthPairC(C.Structure):
    _fields_ = [
        ("surf_flat", C.POINTER(Point)), ("n_surf_flat", C.c_int32),
        ("corner_sharp", C.POINTER(Point)), ("n_corner_sharp", C.c_int32),
        ("surf_last", C.POINTER(Point)), ("n_surf_last", C.c_int32),
        ("corner_last", C.POINTER(Point)), ("n_corner_last", C.c_int32),
        ("state", C.c_double * STATE_DIM),
        ("cov", C.c_double * (ERR_DIM * ERR_DIM)),
        ("true_t", C.c_double * 3), ("true_q", C.c_double * 4),
        ("speed", C.c_double), ("yaw_rate", C.c_double),
        ("n_raw_last", C.c_int32), ("n_raw_new", C.c_int32),
    ]


def lib_path():
    return os.path.join(_HERE, "liblins_host.so")


def lib():
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError(f"{p} is missing — run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(p)
        L.lins_synth_generate.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(SynthPairC)]
        L.lins_synth_generate.restype = C.c_int
        L.lins_synth_raw_scan.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.POINTER(Point), C.c_int]
        L.lins_synth_raw_scan.restype = C.c_int
        L.lins_synth_generate_scene.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(SynthPairC)]
        L.lins_synth_generate_scene.restype = C.c_int
        L.lins_synth_raw_scan_scene.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(Point), C.c_int]
        L.lins_synth_raw_scan_scene.restype = C.c_int
        dp = C.POINTER(C.c_double)
        L.lins_synth_seq_raw_scan.argtypes = [C.c_uint32, C.c_int, C.POINTER(Point), C.c_int]
        L.lins_synth_seq_imu.argtypes = [C.c_uint32, C.c_int, dp, dp]
        L.lins_synth_seq_truth.argtypes = [C.c_uint32, C.c_double, dp, dp, dp]
        for f in (L.lins_synth_seq_raw_scan, L.lins_synth_seq_imu, L.lins_synth_seq_truth):
            f.restype = C.c_int
        L.lins_frontend_extract.argtypes = [C.POINTER(Point), C.c_int, C.c_double, C.POINTER(Features)]
        L.lins_frontend_extract.restype = C.c_int
        L.lins_transform_to_end.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
                                            C.POINTER(Point), C.c_int, C.POINTER(Point)]
        L.lins_transform_to_end.restype = None
        L.lins_filter_default_params.argtypes = [C.POINTER(FilterParams)]
        L.lins_filter_init.argtypes = [C.POINTER(Filter), C.POINTER(FilterParams)] + [C.POINTER(C.c_double)] * 3
        L.lins_filter_predict.argtypes = [C.POINTER(Filter), C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.lins_filter_reset1.argtypes = [C.POINTER(Filter)]
        for f in (L.lins_filter_default_params, L.lins_filter_init, L.lins_filter_predict, L.lins_filter_reset1):
            f.restype = None
        _LIB = L
    return _LIB


def _buf(n):
    a = np.zeros((n, 4), dtype=np.float32)
    return a, a.ctypes.data_as(C.POINTER(Point))


def synth_pair(index, seed=SYNTH_SEED, scene=0):
    """Seeded synthetic scan pair `index` as a ScanPair.  scene: 0 = the room, 1 = the open scene family
    (trunks, far wall segments, lost returns, a moving box: csrc/host/synth.cpp)."""
    sf, psf = _buf(MAX_QUERY)
    cs, pcs = _buf(MAX_QUERY)
    sl, psl = _buf(CLOUD_MAX)
    cl, pcl = _buf(1920)
    sp = SynthPairC()
    sp.surf_flat, sp.corner_sharp, sp.surf_last, sp.corner_last = psf, pcs, psl, pcl
    rc = lib().lins_synth_generate_scene(scene, seed, index, C.byref(sp))
    if rc != 0:
        raise RuntimeError(f"lins_synth_generate_scene({scene}) failed: {rc}")
    meta = dict(true_t=np.array(sp.true_t[:]), true_q=np.array(sp.true_q[:]), speed=sp.speed,
                yaw_rate=sp.yaw_rate, n_raw=(sp.n_raw_last, sp.n_raw_new), index=index, seed=seed, scene=scene)
    return ScanPair(sf[: sp.n_surf_flat].copy(), cs[: sp.n_corner_sharp].copy(), sl[: sp.n_surf_last].copy(),
                    cl[: sp.n_corner_last].copy(), np.array(sp.state[:]), np.array(sp.cov[:]), meta)


def synth_batch(n, start=0, seed=SYNTH_SEED, scene=0):
    return [synth_pair(start + i, seed, scene) for i in range(n)]


def synth_raw_scan(index, k, seed=SYNTH_SEED, scene=0):
    a, p = _buf(CLOUD_MAX)
    n = lib().lins_synth_raw_scan_scene(scene, seed, index, k, p, CLOUD_MAX)
    if n < 0:
        raise RuntimeError(f"lins_synth_raw_scan failed: {n}")
    return a[:n].copy()


def synth_seq_raw_scan(seq, k):
    """Raw cloud (firing order) of sweep k of the seeded scan SEQUENCE `seq` (one trajectory: lins_synth_seq_raw_scan)."""
    a, p = _buf(CLOUD_MAX)
    n = lib().lins_synth_seq_raw_scan(seq, k, p, CLOUD_MAX)
    if n < 0:
        raise RuntimeError(f"lins_synth_seq_raw_scan failed: {n}")
    return a[:n].copy()


def synth_seq_imu(seq, k):
    """(acc, gyr): the 40 IMU samples (400 Hz) of sweep k of sequence `seq`, each 40 x 3."""
    acc, gyr = np.zeros((40, 3)), np.zeros((40, 3))
    dp = C.POINTER(C.c_double)
    n = lib().lins_synth_seq_imu(seq, k, acc.ctypes.data_as(dp), gyr.ctypes.data_as(dp))
    if n != 40:
        raise RuntimeError(f"lins_synth_seq_imu failed: {n}")
    return acc, gyr


def synth_seq_truth(seq, tau):
    """(x, y, yaw, speed, yaw_rate) of the sensor at time tau [s] since the start of sweep 0."""
    xyy, v, w = np.zeros(3), C.c_double(0), C.c_double(0)
    rc = lib().lins_synth_seq_truth(seq, tau, xyy.ctypes.data_as(C.POINTER(C.c_double)), C.byref(v), C.byref(w))
    if rc != 0:
        raise RuntimeError(f"lins_synth_seq_truth failed: {rc}")
    return xyy[0], xyy[1], xyy[2], v.value, w.value


def frontend_extract(raw, scan_period=0.1):
    raw = np.ascontiguousarray(raw, dtype=np.float32).reshape(-1, 4)
    cs, pcs = _buf(192)
    cls, pcls = _buf(1920)
    sf, psf = _buf(MAX_QUERY)
    slf, pslf = _buf(CLOUD_MAX)
    f = Features()
    f.corner_sharp, f.corner_less_sharp, f.surf_flat, f.surf_less_flat = pcs, pcls, psf, pslf
    rc = lib().lins_frontend_extract(raw.ctypes.data_as(C.POINTER(Point)), len(raw), scan_period, C.byref(f))
    if rc != 0:
        raise RuntimeError(f"lins_frontend_extract failed: {rc}")
    return dict(corner_sharp=cs[: f.n_corner_sharp].copy(), corner_less_sharp=cls[: f.n_corner_less_sharp].copy(),
                surf_flat=sf[: f.n_surf_flat].copy(), surf_less_flat=slf[: f.n_surf_less_flat].copy(),
                n_segmented=f.n_segmented, n_outlier=f.n_outlier)


def transform_to_end(t, q, pts, scan_period=0.1):
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 4)
    out = np.empty_like(pts)
    t = (C.c_double * 3)(*t)
    q = (C.c_double * 4)(*q)
    lib().lins_transform_to_end(t, q, scan_period, pts.ctypes.data_as(C.POINTER(Point)), len(pts),
                                out.ctypes.data_as(C.POINTER(Point)))
    return out
