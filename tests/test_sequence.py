"""The in-situ checker on the CPU: the reference's own state machine over a 60-scan sequence, performIESKF() swapped (a
macro around the #include of StateEstimator.hpp, oracle/ref_seq_driver.cpp) for the INTEGRATION.md section 2 binding —
here calling the CPU oracle through the signature of lins_host_perform_ieskf.  What this pins without a GPU: the swap
itself, the binding's packing / unpacking with the reference's real types, and that the oracle's 1e-16 .. 1e-12
per-update differences stay bounded through 58 re-linearisations, kd-tree rebuilds and filter resets
(tests/test_gpu_sequence.py runs the same sequence through the C ABI on the device)."""
import ctypes as C
import os

import numpy as np
import pytest

import seq_common


@pytest.fixture(scope="module")
def ref_seq():
    from oracle import ref_seq as r

    if not r.available():
        if os.environ.get("LINS_REQUIRE_REF") == "1":
            pytest.fail("LINS_REQUIRE_REF=1 and oracle/_ref/liblins_ref_seq.so is neither built nor buildable here")
        pytest.skip("oracle/_ref/liblins_ref_seq.so not built and /root/reference not present")
    r.lib()
    return r


def test_the_reference_state_machine_runs_the_sequence(pkg, host, ref_seq):
    """first scan -> second scan (ICP initialisation, SE:376-425) -> running; every later scan reaches performIESKF,
    converges in a handful of iterations and follows the ground-truth circle (planar: x, y, yaw) to a few centimetres."""
    prm = pkg.default_params(num_iter=30)
    inputs = seq_common.sequence_inputs(host, 7, 30)
    recs = seq_common.run(ref_seq, prm, inputs)
    assert [r.status for r in recs[:3]] == [1, 3, 3] and all(r.status == 3 for r in recs[1:])
    assert all(r.ran_update and r.converged and not r.diverged for r in recs[2:])
    x0, y0, yaw0, v, w = host.synth_seq_truth(7, 0.1)  # the global frame is the first scan's (end of sweep 0)
    xe, ye, yawe, _, _ = host.synth_seq_truth(7, 0.1 * len(recs))
    c, s = np.cos(-yaw0), np.sin(-yaw0)
    want = np.array([c * (xe - x0) - s * (ye - y0), s * (xe - x0) + c * (ye - y0)])
    got = np.array(recs[-1].global_state[:2])
    assert np.linalg.norm(got - want) < 0.15 * max(1.0, v * 0.1 * len(recs)), (got, want)


def test_sequence_through_the_swapped_call_equals_the_unmodified_reference(pkg, host, oracle, ref_seq):
    prm = pkg.default_params(num_iter=30)
    inputs = seq_common.sequence_inputs(host, 11, 60)
    want = seq_common.run(ref_seq, prm, inputs)
    fn = oracle.lib().oracle_perform_ieskf_hook
    got = seq_common.run(ref_seq, prm, inputs, hook=(fn, None))
    worst_p, worst_a = seq_common.compare(want, got)
    assert sum(r.ran_update for r in want) >= 58
    print(f"60 scans: largest globalState_ difference {worst_p:.2e} m, {worst_a:.2e} rad")
