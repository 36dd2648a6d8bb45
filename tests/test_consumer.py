"""The drop-in boundary has a real C++ consumer: tests/consumer/fusion_node_stub.cpp is INTEGRATION.md §2's binding
compiled as a C++11 translation unit (reference types stubbed: no Eigen / PCL / ROS here).
  * (CPU) the snippet in INTEGRATION.md and the code in the stub are the same text (two documented substitutions);
    the stub compiles against include/ with -std=c++11 -Wall -Wextra;
  * (GPU) built, linked against liblins_ieskf.so and run: performIESKF through lins_host_perform_ieskf returns what
    the Python harness gets from the same C ABI."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "consumer", "fusion_node_stub.cpp")
PKG_DIR = os.path.join(ROOT, "lins---lidar-inertial-slam_amd")


def snippet_from_stub():
    src = open(STUB).read()
    body = src.split("// BEGIN INTEGRATION.md section 2\n")[1].split("  // END INTEGRATION.md section 2")[0]
    lines = [l[2:] if l.startswith("  ") else l for l in body.splitlines()]  # (member functions: one indent level less)
    text = "\n".join(lines)
    # the two places where the stub stands in for Eigen / PCL
    text = text.replace("static const lins_point* points_of(const Cloud& c)", "static const lins_point* points_of(const pcl::PointCloud<PointType>& c)")
    text = re.sub(r"copy_covariance_row_major\(filter_->covariance_, in\.cov\);\s*// Eigen: (.*)", r"\1;", text)
    text = re.sub(r"copy_covariance_row_major\(out\.cov, Pk_\);\s*// Eigen: (.*)", r"\1;", text)
    return text.strip() + "\n"


def test_integration_md_snippet_is_the_compiled_code():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc.split("## 2. Binding")[1].split("```cpp\n")[1].split("```")[0]
    want = '#include "lins_host.h"   // pulls in lins_ieskf.h\n\n// members, created once (e.g. in the StateEstimator constructor, SE:187):\n' + snippet_from_stub()
    assert block == want


def test_consumer_compiles_as_cxx11(tmp_path):
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", STUB,
                           "-o", str(tmp_path / "stub.o")])


@pytest.mark.gpu
def test_consumer_runs_perform_ieskf_through_the_c_abi(pkg, ieskf, host, tmp_path):
    exe = str(tmp_path / "fusion_node_stub")
    subprocess.check_call(["g++", "-std=c++11", "-O1", "-I", os.path.join(ROOT, "include"), STUB, "-L", PKG_DIR, "-llins_ieskf",
                           "-Wl,-rpath," + PKG_DIR, "-o", exe])
    for idx in (0, 3):
        pair = host.synth_pair(idx)
        with open(tmp_path / "pair.bin", "wb") as f:
            f.write(np.array([len(pair.surf_flat), len(pair.corner_sharp), len(pair.surf_last), len(pair.corner_last)], np.int32).tobytes())
            for c in (pair.surf_flat, pair.corner_sharp, pair.surf_last, pair.corner_last):
                f.write(np.ascontiguousarray(c).tobytes())
            f.write(pair.state.tobytes())
            f.write(pair.cov.tobytes())
        env = dict(os.environ, LD_LIBRARY_PATH=PKG_DIR + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        subprocess.check_call([exe, str(tmp_path / "pair.bin"), str(tmp_path / "post.bin")], env=env)
        got = np.fromfile(tmp_path / "post.bin", dtype=np.float64)
        prm = pkg.default_params(num_iter=30)
        with ieskf.IeskfContext(prm, max_batch=1, max_targets=16 * 1800) as c:
            want, used = c.perform_ieskf(pair)
        assert not used
        assert np.array_equal(got[:19], want.state) and np.array_equal(got[19:].reshape(18, 18), want.cov)
