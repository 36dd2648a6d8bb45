"""bench.py's contract on the GPU box: one JSON line, the last line on stdout, with the roofline object — also
when the RCCL path is exercised (forced on one GPU: the same code the N > 1 launches run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, port):
    env = dict(os.environ, **extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--steps", "3", "--warmup", "1", "--batch", "320"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    return json.loads(lines[-1])  # the LAST line must be the result


def test_bench_prints_one_json_line_last():
    d = _run({}, 0)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["higher_is_better"] is True
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert abs(d["value"] - 320 * 10 * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert "inputs resident in HBM" in d["config"]["workload"]
    assert d["roofline"]["kernel_ms"] <= d["ms_per_step"]


def test_bench_checks_its_own_results_and_times_the_reference():
    """The default command's CPU leg on a small sample: parity_checked compares the timed batch's GPU results with the
    oracle on the same scans; cpu_baseline is the reference's own code when oracle/_ref travelled."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "300", "--cpu-sample",
                        "16", "--no-extras"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    d = json.loads([l for l in p.stdout.decode().splitlines() if l.strip()][-1])
    pc = d["parity_checked"]
    assert pc["ok"] and pc["flags_equal"] and pc["scans"] == 16 and pc["max_dp"] <= 1e-6 and pc["max_rel_dP"] <= 1e-9
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["kind"] in ("reference", "port")
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "liblins_ref.so")):
        assert cb["kind"] == "reference" and cb["port"]["reduced_all_cores"]["value"] > 0


def test_bench_with_the_rccl_gather_forced_on_one_gpu():
    d = _run({"LINS_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577", "RANK": "0", "WORLD_SIZE": "1",
              "LOCAL_RANK": "0"}, 29577)
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert "all-gather" in d["config"]["parallelism"]


RCCL_SCRIPT = r"""
import importlib, sys
import numpy as np
import torch  # first: the process then runs on ONE HIP runtime (the one PyTorch brings), like bench.py
sys.path.insert(0, %r)
PKG = "lins---lidar-inertial-slam_amd"
pkg = importlib.import_module(PKG); host = importlib.import_module(PKG + ".host"); ieskf = importlib.import_module(PKG + ".ieskf")
defs = importlib.import_module(PKG + "._ctypes_defs")
prm = pkg.default_params(num_iter=10, fixed_iters=1)
batch = host.synth_batch(%d, start=5000)
with ieskf.IeskfContext(prm, max_batch=len(batch), max_targets=16384, search="mr") as c:
    c.upload(batch)
    c.run(); c.sync()
    plain = c.download()
    c.rccl_init(c.rccl_unique_id(), 0, 1)
    c.set_pipelined(True)
    poses = [torch.zeros(len(batch) * 192, dtype=torch.uint8, device="cuda") for _ in range(2)]
    gathered = [torch.zeros(len(batch) * 192, dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for k in range(4):
        c.run(poses[k & 1].data_ptr(), 7000)
        c.pose_allgather(poses[k & 1].data_ptr(), len(batch), gathered[k & 1].data_ptr())
    c.sync()
    piped = c.download()
    rec = np.frombuffer(gathered[1].cpu().numpy().tobytes(), dtype=defs.POSE_DTYPE)
    c.set_pipelined(False)
    c.rccl_destroy()
assert np.array_equal(rec["scan_id"], 7000 + np.arange(len(batch)))
for a, b, r in zip(plain, piped, rec):
    assert np.array_equal(a.state, b.state) and np.array_equal(a.cov, b.cov)
    assert np.array_equal(r["state"], a.state) and (r["iters"], r["m_surf"], r["m_corner"]) == (a.iters, a.m_surf, a.m_corner)
print("RCCL_GATHER_OK")
"""


@pytest.mark.parametrize("n_scans,queues", [(300, "one launch"), (700, "two launch queues")])
def test_pose_allgather_through_the_c_abi_world_of_one(n_scans, queues):
    """lins_rccl_unique_id / _init / lins_pose_allgather / _destroy (include/lins_ieskf.h) in the pipelined staged mode:
    the records RCCL delivers are the records the update kernel wrote, for the LAST of four back-to-back runs — for a batch
    within the device's workgroup slots (one launch per run) and for one beyond them with every run dealt to the context's
    two launch queues (lins_set_launch_queues; forced with the debug knob so that it does not hang on timing): the gather
    then waits for both queues without joining them."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if queues != "one launch":
        env.update(LINS_ENABLE_DEBUG_KNOBS="1", LINS_SPLIT_STREAMS="2")
    p = subprocess.run([sys.executable, "-c", RCCL_SCRIPT % (ROOT, n_scans)], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0 and b"RCCL_GATHER_OK" in p.stdout, p.stderr.decode()[-3000:]


def test_driver_smoke_entry_point_runs():
    """__graft_entry__.smoke() — what the driver runs before the bench — on the stock scan 0 with its default context."""
    import __graft_entry__ as g

    g.smoke()
