"""The C ABI boundary without a GPU: both shared libraries load, export every symbol that
include/*.h declares, and the ctypes mirrors have the C structs' sizes."""
import ctypes as C
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DECL = re.compile(r"^\s*(?:const\s+)?(?:int|void|char\s*\*|const char\s*\*)\s*\*?\s*(lins_\w+)\s*\(", re.M)


def declared(header):
    return sorted(set(DECL.findall(open(os.path.join(ROOT, "include", header)).read())))


def test_headers_declare_the_expected_surface():
    names = declared("lins_ieskf.h")
    for must in ("lins_create", "lins_destroy", "lins_ieskf_update", "lins_ieskf_update_batch", "lins_batch_upload",
                 "lins_batch_run", "lins_batch_download", "lins_correspondences", "lins_reduce_pass", "lins_strerror"):
        assert must in names
    assert "lins_host_perform_ieskf" in declared("lins_host.h")


def test_ieskf_library_exports_every_declared_symbol(ieskf):
    L = ieskf.lib()  # loads liblins_ieskf.so (cross-compiled for gfx950; loading needs no GPU)
    for name in declared("lins_ieskf.h") + declared("lins_map.h") + ["lins_host_perform_ieskf"]:
        assert hasattr(L, name), f"liblins_ieskf.so does not export {name}"
    assert L.lins_strerror(0) == b"ok" and b"capacity" in L.lins_strerror(-3)


DEVICE_SIDE_OF_HOST_HEADER = ("lins_host_perform_ieskf", "lins_extract_features_batch", "lins_last_frontend_stats",
                              "lins_streams_init", "lins_streams_step", "lins_streams_stats", "lins_streams_peek",
                              "lins_segment_batch", "lins_last_segment_ms", "lins_streams_step_raw")


def test_host_library_exports_every_declared_symbol(host):
    L = host.lib()
    for name in declared("lins_host.h"):
        if name in DEVICE_SIDE_OF_HOST_HEADER:
            continue  # live in liblins_ieskf.so (they drive the GPU path)
        assert hasattr(L, name), f"liblins_host.so does not export {name}"


def test_device_entries_of_the_host_header_are_in_the_hip_library(ieskf):
    L = ieskf.lib()
    for name in DEVICE_SIDE_OF_HOST_HEADER:
        assert name in declared("lins_host.h") and hasattr(L, name), name


def test_no_compute_without_a_device(pkg, ieskf):
    """Creating a context without a GPU fails loudly (there is no CPU fallback)."""
    import pytest
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(ieskf.LinsError, match="-5|-2"):
        ieskf.IeskfContext(pkg.default_params())


def test_ctypes_mirrors_match_the_c_structs(pkg, host):
    defs = __import__("importlib").import_module("lins---lidar-inertial-slam_amd._ctypes_defs")
    src = r"""
#include <stdio.h>
#include "lins_host.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(lins_point), sizeof(lins_params), sizeof(lins_scan_pair),
         sizeof(lins_result), sizeof(lins_pose_record), sizeof(lins_corr), sizeof(lins_filter), sizeof(lins_features),
         sizeof(lins_synth_pair));
  return 0;
}
"""
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    want = [C.sizeof(defs.Point), C.sizeof(defs.Params), C.sizeof(defs.ScanPairC), C.sizeof(defs.ResultC),
            C.sizeof(defs.PoseRecordC), defs.CORR_DTYPE.itemsize, C.sizeof(host.Filter), C.sizeof(host.Features),
            C.sizeof(host.SynthPairC)]
    assert sizes == want


def test_strided_point_access_and_field_offsets_of_the_scan_pair():
    """lins_scan_pair.point_stride_bytes (0 / 16: packed lins_point arrays; 32: pcl::PointXYZI as it lies in memory, PH:52):
    the header's one definition of the access, lins_point_load, reads x, y, z at bytes 0 / 4 / 8 and the intensity at
    byte 12 resp. 16 — compiled as C against include/ — and the new fields sit where the ctypes mirror puts them."""
    defs = __import__("importlib").import_module("lins---lidar-inertial-slam_amd._ctypes_defs")
    src = r"""
#include <stddef.h>
#include <stdio.h>
#include "lins_ieskf.h"
int main(void) {
  float packed[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  float wide[16] = {1, 2, 3, -77, 4, -77, -77, -77, 5, 6, 7, -77, 8, -77, -77, -77};
  int ok = 1;
  for (int s = 0; s < 3; ++s) {
    const int stride = s == 0 ? 0 : (s == 1 ? 16 : 32);
    const lins_point* base = (const lins_point*)(stride == 32 ? wide : packed);
    for (int i = 0; i < 2; ++i) {
      const lins_point p = lins_point_load(base, stride, i);
      ok = ok && p.x == 1 + 4 * i && p.y == 2 + 4 * i && p.z == 3 + 4 * i && p.intensity == 4 + 4 * i;
    }
  }
  printf("%d %zu %zu %zu\n", ok, offsetof(lins_scan_pair, point_stride_bytes), offsetof(lins_scan_pair, state), offsetof(lins_scan_pair, cov));
  return 0;
}
"""
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        ok, o_stride, o_state, o_cov = [int(x) for x in subprocess.check_output([exe]).split()]
    assert ok == 1
    assert (o_stride, o_state, o_cov) == (defs.ScanPairC.point_stride_bytes.offset, defs.ScanPairC.state.offset, defs.ScanPairC.cov.offset)
