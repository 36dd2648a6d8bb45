"""The drop-in boundary exercised where it would live: INSIDE the reference's own state machine, over a scan sequence.

oracle/ref_seq_driver.cpp compiles the reference's StateEstimator.hpp verbatim — processImu, processPCL, processScan,
integrateTransformation, filter_->reset(1), updatePointCloud, the kd-tree rebuilds: every statement the reference's own —
with ONE call swapped by a macro around the #include: `performIESKF();` (SE:443) reaches the INTEGRATION.md section 2
binding, written with the reference's real types, which calls lins_host_perform_ieskf of liblins_ieskf.so through the
C ABI.  60 consecutive synthetic sweeps along one trajectory + their IMU go through processImu / processPCL twice —
the unmodified reference, and the reference with the GPU path in its loop — and every scan's flags and globalState_
must agree (VERDICT r03 "missing" #2: per-pair parity says nothing about 58 re-linearisations, kd-tree rebuilds and
resets in a row).  Second test: the same sequence, raw clouds in, through the DEVICE-RESIDENT chain
(lins_streams_step_raw: segmentation, feature front-end, update, re-projection, clouds staying in HBM) with the
product's StatePredictor mirror between the scans, against the reference's two nodes."""
import ctypes as C
import importlib
import os

import numpy as np
import pytest

import seq_common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref_seq():
    from oracle import ref_seq as r

    if not r.available():
        if os.environ.get("LINS_REQUIRE_REF") == "1":
            pytest.fail("LINS_REQUIRE_REF=1 and oracle/_ref/liblins_ref_seq.so did not travel with the snapshot")
        pytest.skip("oracle/_ref/liblins_ref_seq.so did not travel and /root/reference is not here to build it")
    r.lib()
    return r


@pytest.mark.parametrize("seq", [11, 7])
def test_sixty_scans_through_the_references_state_machine_with_the_gpu_path_in_its_loop(pkg, host, ieskf, ref_seq, seq):
    prm = pkg.default_params(num_iter=30)
    inputs = seq_common.sequence_inputs(host, seq, 60)
    want = seq_common.run(ref_seq, prm, inputs)
    assert sum(r.ran_update for r in want) >= 58
    with ieskf.IeskfContext(prm, max_batch=1, max_targets=16 * 1800) as ctx:  # lins_create(..., LINE_NUM * SCAN_NUM)
        fn = ieskf.lib().lins_host_perform_ieskf
        got = seq_common.run(ref_seq, prm, inputs, hook=(fn, ctx._h))
        assert ctx.last_search() in ("lds", "lds1", "mr", "binned")  # (the device kernels did run behind the hook)
    worst_p, worst_a = seq_common.compare(want, got, pos_tol=1e-5, ang_tol=1e-6)
    print(f"sequence {seq}: 60 scans, largest globalState_ difference {worst_p:.2e} m, {worst_a:.2e} rad")


def test_sixty_raw_scans_through_the_device_resident_chain_against_the_references_two_nodes(pkg, host, ieskf, ref_seq):
    """Reference side: image_projection_node (compiled verbatim, oracle/ref.py segment) -> processPCL.  Device side:
    lins_streams_step_raw — raw cloud in, posterior out, clouds resident in HBM between the scans — with the host
    StatePredictor mirror (lins_filter_*) doing what processImu / filter_->update / reset(1) do between two scans.  The
    device chain starts where the reference's two-scan bootstrap (SE:331-425: ICP + IMU pre-integration, host code out of
    scope) ends: from the reference's filter after its second scan; from then on the two chains only share their inputs."""
    from oracle import ref

    seq, n_scans = 11, 60
    prm = pkg.default_params(num_iter=30)
    raws = [host.synth_seq_raw_scan(seq, k) for k in range(n_scans)]
    imus = [host.synth_seq_imu(seq, k) for k in range(n_scans)]
    inputs = [(0.1 * (k + 1), imus[k][0], imus[k][1], ref.segment(raws[k])) for k in range(n_scans)]
    want = seq_common.run(ref_seq, prm, inputs)
    L = host.lib()
    fp = host.FilterParams()
    L.lins_filter_default_params(C.byref(fp))
    filt = host.Filter()
    dp = C.POINTER(C.c_double)
    z3 = (C.c_double * 3)(0, 0, 0)
    L.lins_filter_init(C.byref(filt), C.byref(fp), z3, z3, z3)
    boot = want[1]  # after processSecondScan
    for i in range(19):
        filt.state[i] = boot.filter_state[i]
    for i in range(324):
        filt.cov[i] = boot.filter_cov[i]
    for i in range(3):
        filt.acc_last[i], filt.gyr_last[i] = boot.imu_last[i], boot.imu_last[3 + i]
    filt.has_imu = 1
    worst_p = worst_a = 0.0
    with ieskf.IeskfContext(prm, max_batch=1, max_targets=16 * 1800) as ctx:
        ctx.streams_init(1)
        # scan 1 is the stream's first: no update, its clouds re-projected with the reference's initial relative pose
        ctx.streams_step_raw([raws[1]], np.array(boot.lin_state)[None], np.eye(18)[None] * 1e-4)
        for k in range(2, n_scans):
            acc, gyr = imus[k]
            for i in range(40):
                L.lins_filter_predict(C.byref(filt), 0.1 / 40, acc[i].ctypes.data_as(dp), gyr[i].ctypes.data_as(dp))
            res, _ = ctx.streams_step_raw([raws[k]], np.array(filt.state[:])[None], np.array(filt.cov[:]).reshape(1, 18, 18))
            r, w = res[0], want[k]
            assert (r.iters, r.converged, r.diverged, r.m_surf, r.m_corner) == (w.iters, w.converged, w.diverged, w.m_surf, w.m_corner), k
            lw = np.array(w.lin_state)
            dpos, dang = float(np.abs(r.state[:3] - lw[:3]).max()), seq_common.quat_angle(r.state[6:10], lw[6:10])
            worst_p, worst_a = max(worst_p, dpos), max(worst_a, dang)
            assert dpos <= 1e-5 and dang <= 1e-6, (k, dpos, dang)
            for i in range(19):  # filter_->update(linState_, Pk_) (SE:598), then reset(1) (SE:446)
                filt.state[i] = r.state[i]
            cov = r.cov.reshape(324)
            for i in range(324):
                filt.cov[i] = cov[i]
            L.lins_filter_reset1(C.byref(filt))
            assert np.abs(np.array(filt.state[:]) - np.array(w.filter_state)).max() <= 1e-5, k
    print(f"device-resident chain, 58 scans: largest relative-pose difference {worst_p:.2e} m, {worst_a:.2e} rad")
