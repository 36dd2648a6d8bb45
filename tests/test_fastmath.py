"""The serial tail's short-series rotation maps (csrc/lins_math.h: axis2quat_fast, quat2axis_fast, phi_and_gt_small)
and its Gauss-Jordan solve (csrc/lins_solve6.h) as a HOST build of the device's own source — fma, division and sqrt
are correctly rounded on both sides, so these are the device's bits — against long-double / binary128 libm, against
the textbook routes of MU:61-88, 304-321, and against numpy.  (tests/test_gpu_math.py runs the device build.)"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "lins---lidar-inertial-slam_amd", "csrc")


@pytest.fixture(scope="module")
def report(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fastmath") / "fastmath_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", CSRC,
                           os.path.join(ROOT, "tests", "native", "fastmath_check.cpp"), "-lquadmath", "-o", exe])
    out = subprocess.check_output([exe, "7"]).decode().splitlines()
    tok = out[0].split()
    vals = {tok[i]: float(tok[i + 1]) for i in range(0, len(tok), 2)}
    rows = np.array([[float(x) for x in line.split()] for line in out[1:]])
    return vals, rows


def test_series_are_correct_to_the_last_ulp(report):
    v, _ = report
    # sin(h)/h, cos(h) on |h| <= 0.5; atan(t)/t and (1 - atan(t)/t)/t^2 on t <= 1/8
    assert v["sinc_ulp"] <= 1.0 and v["cos_ulp"] <= 1.0 and v["atanc_ulp"] <= 1.0 and v["atanq_ulp"] <= 1.5


def test_fast_maps_agree_with_the_libm_routes(report):
    v, _ = report
    assert v["n_small"] > 50000 and v["n_general"] > 50000  # both branches of every map were exercised
    # componentwise distance to the libm route in ulps of the largest component (each route is itself ~2 ulp from
    # the true value: an atan2 or sin / cos, a square root, divisions)
    assert v["a2q_ulp"] <= 2.0 and v["q2a_ulp"] <= 4.0 and v["phi_ulp"] <= 4.0 and v["gt_ulp"] <= 4.0
    # the de-skew's form with its coefficients in a table (axis2quat_tab: LDS on the device) returns axis2quat_fast's bits
    assert v["tab_diff"] == 0


def test_gauss_jordan_against_numpy(report):
    v, rows = report
    assert v["gj_res"] <= 1e-14  # residual of the diagonally dominant systems
    a = rows[:, :42].reshape(-1, 6, 7)
    x = rows[:, 42:]
    want = np.stack([np.linalg.solve(m[:, :6], m[:, 6]) for m in a])
    cond = np.array([np.linalg.cond(m[:, :6]) for m in a])
    err = np.abs(x - want).max(axis=1) / np.abs(want).max(axis=1)
    assert (err <= 1e-15 * np.maximum(cond, 10.0) * 20).all(), (err / cond).max()
