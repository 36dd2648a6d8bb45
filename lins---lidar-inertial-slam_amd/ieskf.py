"""Binding of liblins_ieskf.so — the HIP IESKF update path behind the C ABI of
include/lins_ieskf.h.  Host-side mirror of the reference's call surface for this
path (StateEstimator::performIESKF and the two correspondence functions).

There is NO CPU fallback here: a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

from ._ctypes_defs import (CORR_DTYPE, POSE_DTYPE, Params, Point, PoseRecordC, Result, ResultC, ScanPairC, default_params,
                           pairs_to_c)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

EXPORTS = [
    "lins_create", "lins_destroy", "lins_strerror", "lins_last_hip_error", "lins_set_search",
    "lins_ieskf_update", "lins_ieskf_update_batch", "lins_batch_upload", "lins_batch_run", "lins_sync",
    "lins_batch_download", "lins_last_kernel_ms", "lins_batch_bytes_per_iter", "lins_batch_total_iters",
    "lins_correspondences", "lins_reduce_pass", "lins_host_perform_ieskf", "lins_transform_to_end_batch",
    "lins_last_reproject_stats", "lins_icp_update_batch", "lins_extract_features_batch", "lins_last_frontend_stats",
    "lins_streams_init", "lins_streams_step", "lins_streams_stats", "lins_streams_peek", "lins_segment_batch",
    "lins_last_segment_ms", "lins_streams_step_raw", "lins_map_correspondences", "lins_scan2map_batch",
    "lins_last_map_stats", "lins_last_search", "lins_kernel_ms_history", "lins_set_pipelined", "lins_set_launch_queues", "lins_runs_span_ms", "lins_launch_ms_history",
    "lins_rccl_unique_id", "lins_rccl_init", "lins_pose_allgather", "lins_rccl_destroy", "lins_last_index_ms", "lins_last_cut",
    "lins_batch_map",
]


class LinsError(RuntimeError):
    pass


class ReprojectJob(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out_xyz", C.c_void_p), ("out_yzx", C.c_void_p), ("n", C.c_int32),
                ("reserved", C.c_int32), ("t", C.c_double * 3), ("q", C.c_double * 4)]


def lib_path():
    # (LINS_IESKF_LIB: A/B timing of two builds of the library in one GPU call, tools/ab_timing.py)
    return os.environ.get("LINS_IESKF_LIB") or os.path.join(_HERE, "liblins_ieskf.so")


def lib():
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise LinsError(f"{p} is missing: the HIP extension was not built (run __graft_entry__.build()); "
                            "there is no CPU fallback for this path")
        L = C.CDLL(p)
        vp, dp = C.c_void_p, C.POINTER(C.c_double)
        L.lins_create.argtypes = [C.POINTER(Params), C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
        L.lins_destroy.argtypes = [vp]
        L.lins_destroy.restype = None
        L.lins_strerror.argtypes = [C.c_int]
        L.lins_strerror.restype = C.c_char_p
        L.lins_last_hip_error.argtypes = [vp]
        L.lins_last_hip_error.restype = C.c_char_p
        L.lins_set_search.argtypes = [vp, C.c_char_p]
        L.lins_ieskf_update.argtypes = [vp, C.POINTER(ScanPairC), C.POINTER(ResultC)]
        L.lins_ieskf_update_batch.argtypes = [vp, C.c_int, C.POINTER(ScanPairC), C.POINTER(ResultC)]
        L.lins_batch_upload.argtypes = [vp, C.c_int, C.POINTER(ScanPairC)]
        L.lins_batch_run.argtypes = [vp, vp, C.c_int32]
        L.lins_sync.argtypes = [vp]
        L.lins_batch_download.argtypes = [vp, C.c_int, C.POINTER(ResultC)]
        L.lins_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.lins_kernel_ms_history.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
        L.lins_last_index_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.lins_set_pipelined.argtypes = [vp, C.c_int]
        L.lins_set_launch_queues.argtypes = [vp, C.c_int]
        L.lins_runs_span_ms.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
        L.lins_launch_ms_history.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
        L.lins_rccl_unique_id.argtypes = [vp, C.c_void_p]
        L.lins_rccl_init.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
        L.lins_pose_allgather.argtypes = [vp, vp, C.c_int, vp]
        L.lins_rccl_destroy.argtypes = [vp]
        L.lins_batch_bytes_per_iter.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.lins_batch_total_iters.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.lins_correspondences.argtypes = [vp, C.POINTER(ScanPairC), dp, C.c_int, vp, vp]
        L.lins_reduce_pass.argtypes = [vp, C.POINTER(ScanPairC), dp, C.c_int, dp, C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32)]
        L.lins_host_perform_ieskf.argtypes = [vp, C.POINTER(Params), C.POINTER(ScanPairC), C.POINTER(ResultC),
                                              C.POINTER(C.c_int32)]
        L.lins_icp_update_batch.argtypes = [vp, C.c_int, C.POINTER(ScanPairC), C.POINTER(ResultC)]
        L.lins_transform_to_end_batch.argtypes = [vp, C.c_int, C.POINTER(ReprojectJob)]
        L.lins_last_reproject_stats.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        for name in EXPORTS:
            if name not in ("lins_destroy", "lins_strerror", "lins_last_hip_error", "lins_last_search"):
                if os.environ.get("LINS_IESKF_LIB") and not hasattr(L, name):
                    continue  # (an older build under A/B timing)
                getattr(L, name).restype = C.c_int
        _LIB = L
    return _LIB


class IeskfContext:
    """lins_ctx wrapper: one HIP stream + device arena on one GPU."""

    def __init__(self, params=None, device=0, max_batch=1, max_targets=16384, search="auto"):
        self.params = params if params is not None else default_params()
        self._h = C.c_void_p()
        self._check(lib().lins_create(C.byref(self.params), device, max_batch, max_targets, C.byref(self._h)))
        self.max_batch = max_batch
        self.set_search(search)
        self._n = 0

    def _check(self, rc):
        if rc != 0:
            msg = lib().lins_strerror(rc).decode()
            if rc == -2 and self._h:
                msg += ": " + lib().lins_last_hip_error(self._h).decode()
            raise LinsError(f"liblins_ieskf error {rc}: {msg}")

    def close(self):
        if self._h:
            lib().lins_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_search(self, mode):
        self._check(lib().lins_set_search(self._h, mode.encode()))
        self.search = mode

    # -- StateEstimator::performIESKF ------------------------------------------------
    def update(self, pair):
        c = pair.as_c()
        r = ResultC()
        self._check(lib().lins_ieskf_update(self._h, C.byref(c), C.byref(r)))
        return Result(r)

    def update_batch(self, pairs, arr=None):
        """lins_ieskf_update_batch; `arr`: a prepared ScanPairC array for these pairs (pairs_strided / map_batch)."""
        arr = pairs_to_c(pairs) if arr is None else arr
        res = (ResultC * len(pairs))()
        self._check(lib().lins_ieskf_update_batch(self._h, len(pairs), arr, res))
        return [Result(r) for r in res]

    def map_batch(self, pairs):
        """lins_batch_map for the cloud sizes of `pairs`, then the clouds written into the context's pinned staging arena
        where the library wants them — standing in for a caller whose feature extraction writes there in the first place.
        Returns the ScanPairC array that points at them: passed to upload / update_batch, the library's staging copy is
        skipped."""
        n = len(pairs)
        counts = np.array([[len(p.surf_flat), len(p.corner_sharp), len(p.surf_last), len(p.corner_last)] for p in pairs], dtype=np.int32)
        clouds = (C.POINTER(Point) * (4 * n))()
        self._check(lib().lins_batch_map(self._h, n, counts.ctypes.data_as(C.POINTER(C.c_int32)), clouds))
        arr = pairs_to_c(pairs)
        for i, p in enumerate(pairs):
            for c, (name, src) in enumerate((("surf_flat", p.surf_flat), ("corner_sharp", p.corner_sharp),
                                             ("surf_less_flat_last", p.surf_last), ("corner_less_sharp_last", p.corner_last))):
                if len(src):
                    C.memmove(clouds[4 * i + c], src.ctypes.data, src.nbytes)
                setattr(arr[i], name, clouds[4 * i + c])
        return arr

    def perform_ieskf(self, pair):
        """performIESKF as the node sees it: GPU loop + ICP fallback on divergence."""
        c = pair.as_c()
        r = ResultC()
        used = C.c_int32(0)
        self._check(lib().lins_host_perform_ieskf(self._h, C.byref(self.params), C.byref(c), C.byref(r), C.byref(used)))
        return Result(r), bool(used.value)

    # -- scan-to-map row (include/lins_map.h) -----------------------------------------------------
    def map_correspondences(self, problem):
        from ._ctypes_defs import MAP_CORR_DTYPE, MapProblemC

        c = problem.as_c()
        corner = np.zeros(len(problem.scan_corner), dtype=MAP_CORR_DTYPE)
        surf = np.zeros(len(problem.scan_surf), dtype=MAP_CORR_DTYPE)
        L = lib()
        L.lins_map_correspondences.argtypes = [C.c_void_p, C.POINTER(MapProblemC), C.c_void_p, C.c_void_p]
        self._check(L.lins_map_correspondences(self._h, C.byref(c), corner.ctypes.data, surf.ctypes.data))
        return corner, surf

    def scan2map_batch(self, problems):
        from ._ctypes_defs import MapProblemC, MapResultC

        n = len(problems)
        arr = (MapProblemC * n)(*[p.as_c() for p in problems])
        res = (MapResultC * n)()
        L = lib()
        L.lins_scan2map_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(MapProblemC), C.POINTER(MapResultC)]
        self._check(L.lins_scan2map_batch(self._h, n, arr, res))
        return [dict(transform=np.array(r.transform[:], dtype=np.float32), iters=r.iters, converged=r.converged,
                     degenerate=r.degenerate, n_sel=r.n_sel) for r in res]

    def map_stats(self):
        ms, q = C.c_float(0), C.c_uint64(0)
        L = lib()
        L.lins_last_map_stats.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        self._check(L.lins_last_map_stats(self._h, C.byref(ms), C.byref(q)))
        return ms.value, q.value

    # -- image_projection_node on the device: raw clouds -> segmented scans --------------------
    def segment_batch(self, raws):
        """raws: list of (n,4) f32 raw clouds in firing order.  Returns a list of host.Segmented."""
        import importlib

        host = importlib.import_module(__package__ + ".host")
        n = len(raws)
        raws = [np.ascontiguousarray(r, dtype=np.float32).reshape(-1, 4) for r in raws]
        ptrs = (C.POINTER(host.Point) * n)(*[r.ctypes.data_as(C.POINTER(host.Point)) for r in raws])
        counts = (C.c_int32 * n)(*[len(r) for r in raws])
        out = (host.SegmentedScanC * n)()
        keep = []
        for k in range(n):
            cloud = np.zeros((host.CLOUD_MAX, 4), np.float32)
            rng = np.zeros(host.CLOUD_MAX, np.float32)
            col = np.zeros(host.CLOUD_MAX, np.uint32)
            ground = np.zeros(host.CLOUD_MAX, np.uint8)
            out[k].cloud = cloud.ctypes.data_as(C.POINTER(host.Point))
            out[k].range = rng.ctypes.data_as(C.POINTER(C.c_float))
            out[k].col = col.ctypes.data_as(C.POINTER(C.c_uint32))
            out[k].ground = ground.ctypes.data_as(C.POINTER(C.c_uint8))
            keep.append((cloud, rng, col, ground))
        L = lib()
        L.lins_segment_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(host.Point)), C.POINTER(C.c_int32),
                                         C.POINTER(host.SegmentedScanC)]
        self._check(L.lins_segment_batch(self._h, n, ptrs, counts, out))
        res = []
        for k in range(n):
            c = host.SegmentedScanC()
            C.memmove(C.byref(c), C.byref(out[k]), C.sizeof(c))
            res.append(host.Segmented(*keep[k], c))
        return res

    def segment_ms(self):
        ms = C.c_float(0)
        L = lib()
        L.lins_last_segment_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        self._check(L.lins_last_segment_ms(self._h, C.byref(ms)))
        return ms.value

    # -- StateEstimator's feature front-end on the device (undistortPcl .. extractFeatures) ----
    def extract_features_batch(self, segs, scan_period=0.1):
        """segs: list of host.Segmented.  Returns a list of dicts like host.frontend_extract()."""
        import importlib

        host = importlib.import_module(__package__ + ".host")
        n = len(segs)
        arr = (host.SegmentedScanC * n)(*[s.c for s in segs])
        feats = (host.Features * n)()
        keep = []
        for k in range(n):
            f, bufs = host._features_buffers()
            feats[k] = f
            keep.append(bufs)
        L = lib()
        L.lins_extract_features_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(host.SegmentedScanC), C.c_double,
                                                  C.POINTER(host.Features)]
        self._check(L.lins_extract_features_batch(self._h, n, arr, scan_period, feats))
        return [host._features_dict(feats[k], keep[k]) for k in range(n)]

    def frontend_stats(self):
        ms, b = C.c_float(0), C.c_uint64(0)
        L = lib()
        L.lins_last_frontend_stats.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        self._check(L.lins_last_frontend_stats(self._h, C.byref(ms), C.byref(b)))
        return ms.value, b.value

    # -- device-resident streams: front-end -> update -> re-projection per scan -----------
    def streams_init(self, n):
        L = lib()
        L.lins_streams_init.argtypes = [C.c_void_p, C.c_int]
        self._check(L.lins_streams_init(self._h, n))
        self._streams = n

    def streams_step(self, segs, prior_state, prior_cov, scan_period=0.1):
        """segs: one host.Segmented per stream; prior_state (n,19), prior_cov (n,18,18).
        Returns (results, feature_counts (n,4) = sharp, less sharp, flat, less flat)."""
        import importlib

        host = importlib.import_module(__package__ + ".host")
        n = self._streams
        assert len(segs) == n
        arr = (host.SegmentedScanC * n)(*[s.c for s in segs])
        ps = np.ascontiguousarray(prior_state, dtype=np.float64).reshape(n, 19)
        pc = np.ascontiguousarray(prior_cov, dtype=np.float64).reshape(n, 324)
        res = (ResultC * n)()
        counts = np.zeros((n, 4), np.int32)
        L = lib()
        L.lins_streams_step.argtypes = [C.c_void_p, C.POINTER(host.SegmentedScanC), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.c_double, C.POINTER(ResultC), C.POINTER(C.c_int32)]
        self._check(L.lins_streams_step(self._h, arr, ps.ctypes.data_as(C.POINTER(C.c_double)),
                                        pc.ctypes.data_as(C.POINTER(C.c_double)), scan_period, res,
                                        counts.ctypes.data_as(C.POINTER(C.c_int32))))
        self._n = 0
        return [Result(r) for r in res], counts

    def streams_step_raw(self, raws, prior_state, prior_cov, scan_period=0.1):
        """Like streams_step, from raw clouds (firing order): image projection / segmentation on the device too."""
        import importlib

        host = importlib.import_module(__package__ + ".host")
        n = self._streams
        assert len(raws) == n
        raws = [np.ascontiguousarray(r, dtype=np.float32).reshape(-1, 4) for r in raws]
        ptrs = (C.POINTER(host.Point) * n)(*[r.ctypes.data_as(C.POINTER(host.Point)) for r in raws])
        cnts = (C.c_int32 * n)(*[len(r) for r in raws])
        ps = np.ascontiguousarray(prior_state, dtype=np.float64).reshape(n, 19)
        pc = np.ascontiguousarray(prior_cov, dtype=np.float64).reshape(n, 324)
        res = (ResultC * n)()
        counts = np.zeros((n, 4), np.int32)
        L = lib()
        L.lins_streams_step_raw.argtypes = [C.c_void_p, C.POINTER(C.POINTER(host.Point)), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(ResultC),
                                            C.POINTER(C.c_int32)]
        self._check(L.lins_streams_step_raw(self._h, ptrs, cnts, ps.ctypes.data_as(C.POINTER(C.c_double)),
                                            pc.ctypes.data_as(C.POINTER(C.c_double)), scan_period, res,
                                            counts.ctypes.data_as(C.POINTER(C.c_int32))))
        self._n = 0
        return [Result(r) for r in res], counts

    def streams_stats(self):
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        L = lib()
        L.lins_streams_stats.argtypes = [C.c_void_p] + [C.POINTER(C.c_float)] * 3
        self._check(L.lins_streams_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def streams_peek(self, stream, which):
        buf = np.zeros((28800, 4), np.float32)
        L = lib()
        L.lins_streams_peek.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        n = L.lins_streams_peek(self._h, stream, which, buf.ctypes.data, len(buf))
        self._check(min(n, 0))
        return buf[:n].copy()

    def icp_update_batch(self, pairs):
        """estimateTransform (the ICP fallback) on the device, from each pair's state pose."""
        arr = pairs_to_c(pairs)
        res = (ResultC * len(pairs))()
        self._check(lib().lins_icp_update_batch(self._h, len(pairs), arr, res))
        self._n = 0
        return [Result(r) for r in res]

    # -- staged batch form -------------------------------------------------------------
    def upload(self, pairs, arr=None):
        arr = pairs_to_c(pairs) if arr is None else arr
        self._check(lib().lins_batch_upload(self._h, len(pairs), arr))
        self._n = len(pairs)

    def run(self, poses_ptr=None, scan_id_base=0):
        self._check(lib().lins_batch_run(self._h, C.c_void_p(poses_ptr) if poses_ptr else None, scan_id_base))

    def sync(self):
        self._check(lib().lins_sync(self._h))

    def download(self, n=None):
        n = self._n if n is None else n
        res = (ResultC * n)()
        self._check(lib().lins_batch_download(self._h, n, res))
        return [Result(r) for r in res]

    def last_kernel_ms(self):
        ms = C.c_float(0)
        self._check(lib().lins_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def last_index_ms(self):
        """HIP-event time (ms) of the search-index build of the last upload (the reference's kd-tree build, SE:1156-1160)."""
        ms = C.c_float(0)
        self._check(lib().lins_last_index_ms(self._h, C.byref(ms)))
        return ms.value

    def last_cut(self):
        """(parts, queue_timeouts): pieces every update of the last run() was cut into (1 = whole updates), and the waits at
        the work queue that ran out over the life of the context (0 in a healthy process)."""
        parts, tail = C.c_int(0), C.c_int(0)
        self._check(lib().lins_last_cut(self._h, C.byref(parts), C.byref(tail)))
        return parts.value, tail.value

    def kernel_ms_history(self, n):
        """HIP-event times (ms) of the update kernels of the last n run() calls, oldest first."""
        ms = (C.c_float * n)()
        self._check(lib().lins_kernel_ms_history(self._h, n, ms))
        return [float(v) for v in ms]

    def set_pipelined(self, on=True):
        """Pipelined staged mode: Joseph kernel / pose gather of run k beside the update kernel of run k + 1."""
        self._check(lib().lins_set_pipelined(self._h, 1 if on else 0))

    def runs_span_ms(self, n):
        """GPU time the last n runs took together (first launch's start to last launch's end, both launch queues)."""
        v = C.c_float(0)
        self._check(lib().lins_runs_span_ms(self._h, int(n), C.byref(v)))
        return float(v.value)

    def launch_ms_history(self, n):
        """durations of the launches of the last n runs, each by its own queue's events: [(first, second or 0.0), ...]"""
        a = (C.c_float * (2 * n))()
        self._check(lib().lins_launch_ms_history(self._h, int(n), a))
        return [(float(a[2 * k]), float(a[2 * k + 1])) for k in range(n)]

    def set_launch_queues(self, queues):
        """2 (default): a batch beyond the device's slots runs as whole-update launches on two streams; 1: one launch, several-part updates."""
        self._check(lib().lins_set_launch_queues(self._h, int(queues)))

    # -- multi-GPU: RCCL all-gather of the pose records through the C ABI (include/lins_ieskf.h) ------
    def rccl_unique_id(self):
        buf = (C.c_char * 128)()
        self._check(lib().lins_rccl_unique_id(self._h, buf))
        return bytes(buf)

    def rccl_init(self, uid, rank, world):
        assert len(uid) == 128
        self._check(lib().lins_rccl_init(self._h, C.c_char_p(uid), rank, world))

    def pose_allgather(self, d_local_ptr, n_records, d_all_ptr):
        self._check(lib().lins_pose_allgather(self._h, C.c_void_p(d_local_ptr), n_records, C.c_void_p(d_all_ptr)))

    def rccl_destroy(self):
        self._check(lib().lins_rccl_destroy(self._h))

    def last_search(self):
        """Kernel family the last batch / pass actually ran (after "auto" and the eligibility fall-backs)."""
        f = lib().lins_last_search
        f.restype = C.c_char_p
        f.argtypes = [C.c_void_p]
        return f(self._h).decode()

    def bytes_per_iter(self):
        b = C.c_uint64(0)
        self._check(lib().lins_batch_bytes_per_iter(self._h, C.byref(b)))
        return b.value

    def total_iters(self):
        b = C.c_uint64(0)
        self._check(lib().lins_batch_total_iters(self._h, C.byref(b)))
        return b.value

    # -- StateEstimator::updatePointCloud's re-projection (transformToEnd) -----------------
    def transform_to_end(self, clouds, poses, yzx=True):
        """clouds: list of (n,4) f32 arrays; poses: list of (t[3], q[4]).  Returns (xyz, yzx) lists."""
        jobs = (ReprojectJob * len(clouds))()
        keep = []
        for k, (cl, (t, q)) in enumerate(zip(clouds, poses)):
            cl = np.ascontiguousarray(cl, dtype=np.float32).reshape(-1, 4)
            o1 = np.empty_like(cl)
            o2 = np.empty_like(cl) if yzx else None
            keep.append((cl, o1, o2))
            jobs[k].inp, jobs[k].out_xyz = cl.ctypes.data, o1.ctypes.data
            jobs[k].out_yzx = o2.ctypes.data if yzx else None
            jobs[k].n = len(cl)
            jobs[k].t[:] = list(t)
            jobs[k].q[:] = list(q)
        self._check(lib().lins_transform_to_end_batch(self._h, len(clouds), jobs))
        self._n = 0
        return [k[1] for k in keep], [k[2] for k in keep]

    def reproject_stats(self):
        ms, b = C.c_float(0), C.c_uint64(0)
        self._check(lib().lins_last_reproject_stats(self._h, C.byref(ms), C.byref(b)))
        return ms.value, b.value

    # -- findCorrespondingSurfFeatures / findCorrespondingCornerFeatures ---------------
    def correspondences(self, pair, lin_state, it):
        c = pair.as_c()
        lin = np.ascontiguousarray(lin_state, dtype=np.float64)
        surf = np.zeros(c.n_surf_flat, dtype=CORR_DTYPE)
        corner = np.zeros(c.n_corner_sharp, dtype=CORR_DTYPE)
        self._check(lib().lins_correspondences(self._h, C.byref(c), lin.ctypes.data_as(C.POINTER(C.c_double)), it,
                                               surf.ctypes.data, corner.ctypes.data))
        return surf, corner

    def reduce_pass(self, pair, lin_state, it):
        c = pair.as_c()
        lin = np.ascontiguousarray(lin_state, dtype=np.float64)
        sums = np.zeros(28)
        ms, mc = C.c_int32(0), C.c_int32(0)
        self._check(lib().lins_reduce_pass(self._h, C.byref(c), lin.ctypes.data_as(C.POINTER(C.c_double)), it,
                                           sums.ctypes.data_as(C.POINTER(C.c_double)), C.byref(ms), C.byref(mc)))
        return sums, ms.value, mc.value


def host_solve_from_sums(params, pair, lin_state, sums):
    """BASELINE.json configs[1]: device correspondences + reduction, host-side
    18x18 solve.  Faithful dense algebra on the 18-state system embedded from the
    28 sums: dx = -W P (g + A d) + d with W = (sigma^2 I + P A)^-1."""
    lin = np.asarray(lin_state, dtype=np.float64)
    filt = pair.state
    P = pair.cov

    def quat2axis(q):
        v = q[1:4]
        m = np.linalg.norm(v)
        if m < 1e-10:
            return v.copy()
        a = 2.0 * np.arctan2(m, q[0])
        a = (a + np.pi) % (2 * np.pi) - np.pi
        return v / m * a

    def qmul(a, b):
        return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                         a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                         a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                         a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])

    ql, qf = lin[6:10], filt[6:10]
    qinv = np.array([ql[0], -ql[1], -ql[2], -ql[3]]) / np.dot(ql, ql)
    d = np.zeros(18)
    d[0:3] = filt[0:3] - lin[0:3]
    d[3:6] = filt[3:6] - lin[3:6]
    d[6:9] = quat2axis(qmul(qinv, qf))
    d[9:12] = filt[10:13] - lin[10:13]
    d[12:15] = filt[13:16] - lin[13:16]
    d[15:18] = filt[16:19] - lin[16:19]
    S = [0, 1, 2, 6, 7, 8]
    A6 = np.zeros((6, 6))
    A6[np.triu_indices(6)] = sums[:21]
    A6 = A6 + np.triu(A6, 1).T
    A = np.zeros((18, 18))
    A[np.ix_(S, S)] = A6
    g = np.zeros(18)
    g[S] = sums[21:27]
    r2 = params.lidar_std ** 2
    W = np.linalg.inv(r2 * np.eye(18) + P @ A)
    dx = -W @ P @ (g + A @ d) + d
    return dx, A, W
