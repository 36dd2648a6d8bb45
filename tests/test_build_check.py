"""The shipped HIP library is free of the register-allocator fault of ROCm 7.2 clang that tools/check_exec_prologue.py
describes (per-lane copies / spill stores in front of the `s_or_b64 exec` that ends a divergent branch — the round-3/4
"load_lds_grid inlining" failure of the 1024 x 3 correspondence pass, root-caused in round 6), and the checker itself
tells the two shapes apart."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_exec_prologue", os.path.join(ROOT, "tools", "check_exec_prologue.py"))
cep = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cep)

FAULTY = """
kernel_a:                               ; @kernel_a
; %bb.0:
\ts_and_saveexec_b64 s[8:9], vcc
\ts_cbranch_execz .LBB0_2
; %bb.1:
\tv_add_f32_e32 v1, v2, v3
.LBB0_2:
\tscratch_store_dword off, v100, off      ; 4-byte Folded Spill
\tv_mov_b32_e32 v86, v96
\ts_or_b64 exec, exec, s[8:9]
\tds_bpermute_b32 v10, v86, v80
"""
# the same block with the restore first, a branch BODY merged with its restore (entered with execnz), and a join that
# opens a region of its own before any per-lane work: all fine
CLEAN = """
kernel_b:                               ; @kernel_b
; %bb.0:
\ts_and_saveexec_b64 s[8:9], vcc
\ts_cbranch_execz .LBB1_2
; %bb.1:
\tv_add_f32_e32 v1, v2, v3
.LBB1_2:
\tv_readlane_b32 s0, v127, 26
\ts_or_b64 exec, exec, s[8:9]
\tv_mov_b32_e32 v86, v96
\ts_and_saveexec_b64 s[10:11], vcc
\ts_cbranch_execnz .LBB1_4
\ts_branch .LBB1_5
.LBB1_4:
\tv_mov_b32_e32 v48, v31
\tv_mov_b32_e32 v45, v28
.LBB1_5:
\ts_or_b64 exec, exec, s[10:11]
\ts_and_saveexec_b64 s[12:13], vcc
\ts_cbranch_execz .LBB1_7
; %bb.6:
\tv_add_f32_e32 v1, v2, v3
.LBB1_7:
\ts_andn2_saveexec_b64 s[12:13], s[12:13]
\tv_rsq_f32_e32 v4, v5
\ts_or_b64 exec, exec, s[12:13]
"""


def test_checker_tells_the_faulty_join_from_the_legitimate_shapes():
    bad = cep.check_lines(FAULTY.splitlines())
    assert len(bad) == 1 and bad[0][0] == "kernel_a" and bad[0][1] == ".LBB0_2"
    assert bad[0][2] == ["scratch_store_dword off, v100, off", "v_mov_b32_e32 v86, v96"]
    assert cep.check_lines(CLEAN.splitlines()) == []


@pytest.mark.skipif(not os.path.exists(cep.OBJDUMP) or shutil.which("cp") is None, reason="needs llvm-objdump of the ROCm image")
def test_shipped_library_has_no_per_lane_work_in_front_of_an_exec_restore():
    lib = os.path.join(ROOT, "lins---lidar-inertial-slam_amd", "liblins_ieskf.so")
    if not os.path.exists(lib):
        pytest.skip("library not built (python -c 'import __graft_entry__ as g; g.build()')")
    sites = cep.check(lib)
    assert sites == [], "\n".join(f"{f} {b}: {' | '.join(i[:6])}" for f, b, i in sites)
