#!/usr/bin/env python
"""How far the product's front-end rows are from the reference's own arithmetic (libm angles, size_t row
truncation): the independent oracle (oracle/frontend_oracle.cpp) against
  * the product's host restatement (csrc/host/frontend.cpp, shares lins_atan2f with the kernels)   [CPU, always]
  * the device kernels (lins_segment_batch, lins_extract_features_batch)                            [with --gpu]
on n seeded synthetic raw scans: differing range-image cells, segmented points, ground flags, feature picks, time tags.
usage: tools/frontend_vs_libm.py [n_scans] [--gpu]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "lins---lidar-inertial-slam_amd"
host = importlib.import_module(PKG + ".host")
from oracle import oracle  # noqa: E402


def compare_segment(o, cloud, rng, col, ground, n, sr, er, ori, n_out):
    """-> dict of difference counts between the oracle's segmented scan `o` and another implementation's."""
    d = {}
    d["size_differs"] = int(o["n"] != n)
    key_o = (np.floor(o["cloud"][: o["n"], 3]).astype(np.int64) * 1800 + o["col"][: o["n"]].astype(np.int64))
    key_x = (np.floor(cloud[:n, 3]).astype(np.int64) * 1800 + col[:n].astype(np.int64))
    so, sx = set(key_o.tolist()), set(key_x.tolist())
    d["cells_only_in_one"] = len(so ^ sx)
    common = sorted(so & sx)
    io = {k: i for i, k in enumerate(key_o.tolist())}
    ix = {k: i for i, k in enumerate(key_x.tolist())}
    a = np.array([io[k] for k in common], dtype=np.int64)
    b = np.array([ix[k] for k in common], dtype=np.int64)
    d["common_cells"] = len(common)
    d["xyz_differs"] = int((o["cloud"][a, :3] != cloud[b, :3]).any(axis=1).sum())
    d["range_differs"] = int((o["range"][a] != rng[b]).sum())
    d["ground_flag_differs"] = int((o["ground"][a] != ground[b]).sum())
    d["ring_index_differs"] = int((o["start_ring"] != sr).sum() + (o["end_ring"] != er).sum())
    d["orientation_ulp_max"] = int(np.abs(o["orientation"].view(np.int32).astype(np.int64) - np.asarray(ori, np.float32).view(np.int32).astype(np.int64)).max())
    d["outlier_count_differs"] = int(o["n_outlier"] != n_out)
    return d


def compare_features(fo, fx):
    d = {}
    for k in ("corner_sharp", "corner_less_sharp", "surf_flat", "surf_less_flat"):
        a, b = fo[k], fx[k]
        d[k + "_count_differs"] = int(len(a) != len(b))
        m = min(len(a), len(b))
        same_xyz = (a[:m, :3] == b[:m, :3]).all(axis=1)
        d[k + "_points_differ"] = int((~same_xyz).sum() + abs(len(a) - len(b)))
        tag = np.abs(a[:m, 3].view(np.int32).astype(np.int64) - b[:m, 3].view(np.int32).astype(np.int64))
        d[k + "_time_tag_ulp>0"] = int((tag[same_xyz] > 0).sum())
        d[k + "_time_tag_ulp_max"] = int(tag[same_xyz].max(initial=0))  # (ulp distances explode for tags next to zero: ring 0, relTime ~ 0)
        d[k + "_time_tag_absdiff_max"] = float(np.abs(a[:m, 3] - b[:m, 3])[same_xyz].max(initial=0))
        d[k + "_total"] = int(len(a))
    return d


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 64
    gpu = "--gpu" in sys.argv
    if gpu:
        import torch  # noqa: F401  (first: one HIP runtime in the process)
    tot_seg_host, tot_feat_host, tot_seg_dev, tot_feat_dev = {}, {}, {}, {}
    raws = [host.synth_raw_scan(i // 2, i % 2) for i in range(n)]
    turn = 0.0
    for a in sys.argv:
        if a.startswith("--turn="):
            turn = float(a.split("=")[1])
    if turn:  # rotate the clouds about z (rounds 1-2: the synthetic sensor fired exactly ON the column edges of IP:225)
        ca, sa = np.float32(np.cos(np.radians(turn))), np.float32(np.sin(np.radians(turn)))
        for r in raws:
            x, y = r[:, 0].copy(), r[:, 1].copy()
            r[:, 0], r[:, 1] = ca * x - sa * y, sa * x + ca * y
    print(f"clouds turned by {turn} deg about z" if turn else "clouds as generated (round 3: seeded azimuth phase + drift + jitter, "
          "no firing within 0.14 column of a rounding edge of IP:225)")

    def acc(t, d):
        for k, v in d.items():
            t[k] = max(t.get(k, 0), v) if k.endswith("_max") else t.get(k, 0) + v

    segs_o = []
    for raw in raws:
        o = oracle.fe_segment(raw)
        segs_o.append(o)
        h = host.frontend_segment(raw)
        acc(tot_seg_host, compare_segment(o, h.cloud, h.range, h.col, h.ground, h.n, np.array(h.c.start_ring[:]), np.array(h.c.end_ring[:]),
                                          [h.c.start_ori, h.c.end_ori, h.c.ori_diff], h.c.n_outlier))
        # feature stage on the SAME segmented input (the oracle's), so that the stage is compared in isolation
        fo = oracle.fe_features(o)
        hs = host.segmented_from_arrays(o["cloud"], o["range"], o["col"], o["ground"], o["n"], o["start_ring"], o["end_ring"], o["orientation"], o["n_outlier"])
        fh = host.frontend_extract_segmented(hs)
        acc(tot_feat_host, compare_features(fo, fh))
    print(f"{n} synthetic raw scans; oracle = reference arithmetic (libm atan2f / sinf / cosf, size_t row truncation)")
    print("projection + ground + segmentation, product host restatement vs oracle:", tot_seg_host)
    print("feature stage on identical segmented input, product host restatement vs oracle:", tot_feat_host)
    if gpu:
        ieskf = importlib.import_module(PKG + ".ieskf")
        pkg = importlib.import_module(PKG)
        with ieskf.IeskfContext(pkg.default_params(), max_batch=min(n, 256), max_targets=16384) as c:
            for lo in range(0, n, 256):
                chunk = raws[lo:lo + 256]
                segs = c.segment_batch(chunk)
                for o, s in zip(segs_o[lo:lo + 256], segs):
                    acc(tot_seg_dev, compare_segment(o, s.cloud, s.range, s.col, s.ground, s.n, np.array(s.c.start_ring[:]), np.array(s.c.end_ring[:]),
                                                     [s.c.start_ori, s.c.end_ori, s.c.ori_diff], s.c.n_outlier))
                hs = [host.segmented_from_arrays(o["cloud"], o["range"], o["col"], o["ground"], o["n"], o["start_ring"], o["end_ring"], o["orientation"], o["n_outlier"])
                      for o in segs_o[lo:lo + 256]]
                feats = c.extract_features_batch(hs)
                for o, f in zip(segs_o[lo:lo + 256], feats):
                    acc(tot_feat_dev, compare_features(oracle.fe_features(o), f))
        print("projection + ground + segmentation, DEVICE kernels vs oracle:", tot_seg_dev)
        print("feature stage on identical segmented input, DEVICE kernel vs oracle:", tot_feat_dev)


if __name__ == "__main__":
    main()
