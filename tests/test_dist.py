"""The N > 1 path on CPU: scan sharding + the all-gather of the fixed-size pose records over
torch.distributed with the gloo backend, world_size 2 (on the GPU box the same code runs over
RCCL).  No compute here — the records are synthetic — the point is ordering, raggedness
and completeness of the exchange."""
import importlib
import os
import socket

import numpy as np
import pytest

PKG = "lins---lidar-inertial-slam_amd"


def test_shard_ranges_partition_the_batch():
    dist_mod = importlib.import_module(PKG + ".dist")
    for n in (0, 1, 7, 8, 1024, 8191):
        for world in (1, 2, 3, 8):
            spans = [dist_mod.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        dist_mod.shard_range(8, 2, 2)


def _worker(rank, world, port, n_total, ret):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dist_mod = importlib.import_module(PKG + ".dist")
        defs = importlib.import_module(PKG + "._ctypes_defs")
        lo, hi = dist_mod.shard_range(n_total, rank, world)
        rec = np.zeros(hi - lo, dtype=defs.POSE_DTYPE)
        rec["scan_id"] = np.arange(lo, hi)
        rec["state"][:, 0] = np.arange(lo, hi) * 0.5  # something recognisable per scan
        rec["iters"] = 10
        local = torch.from_numpy(rec.view(np.uint8).copy())
        out = dist_mod.gather_pose_records(local, n_total)
        ok = (len(out) == n_total and np.array_equal(out["scan_id"], np.arange(n_total))
              and np.array_equal(out["state"][:, 0], np.arange(n_total) * 0.5) and (out["iters"] == 10).all())
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 1])
def test_pose_gather_world_size_2_gloo(n_total):
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(0) is True and ret.get(1) is True


def _pipeline_worker(rank, world, port, n_total, steps, ret):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dist_mod = importlib.import_module(PKG + ".dist")
        defs = importlib.import_module(PKG + "._ctypes_defs")
        pipe = dist_mod.PoseGatherPipeline(n_total, rank, world, device="cpu")
        seen = []
        for k in range(steps):
            b, buf = pipe.begin_step()  # (waits for the gather that last read this buffer)
            rec = np.zeros(pipe.hi - pipe.lo, dtype=defs.POSE_DTYPE)
            rec["scan_id"] = np.arange(pipe.lo, pipe.hi)
            rec["state"][:, 0] = np.arange(pipe.lo, pipe.hi) * 0.5 + 1000.0 * k
            rec["iters"] = 10
            rec["m_corner"] = k
            buf[: rec.nbytes] = torch.from_numpy(rec.view(np.uint8).copy())  # "the update of step k"
            pipe.gather_newest()  # step k - 1 travels now
            pipe.end_step(b)
            if k >= 2:  # the gather of step k - 2 has been waited for by begin_step: its buffer must hold step k - 2
                pass
        pipe.drain()
        out = pipe.records()  # raises when out of order
        ok = (len(out) == n_total and np.array_equal(out["scan_id"], np.arange(n_total))
              and np.array_equal(out["state"][:, 0], np.arange(n_total) * 0.5 + 1000.0 * (steps - 1))
              and int(out["iters"].sum()) == 10 * n_total and (out["m_corner"] == steps - 1).all())
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total,steps", [(8, 5), (7, 4), (1, 3), (9, 1)])
def test_benchmarked_double_buffered_gather_world_size_2_gloo(n_total, steps):
    """bench.py's own exchange step (dist.PoseGatherPipeline): ordering, raggedness, completeness and that the
    records returned are the LAST step's — the code path the N > 1 bench runs over RCCL."""
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_pipeline_worker, args=(r, 2, port, n_total, steps, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(0) is True and ret.get(1) is True


def _real_records_worker(rank, world, port, n_total, ret):
    """Each rank runs the (CPU) oracle on ITS shard of seeded scan pairs, packs the results the way the kernel writes
    them on the device (records_from_results) and takes part in the gather; rank 0 compares what arrived with a
    single-process run over the whole batch."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module(PKG)
        host = importlib.import_module(PKG + ".host")
        dist_mod = importlib.import_module(PKG + ".dist")
        from oracle import oracle

        prm = pkg.default_params(num_iter=10, fixed_iters=1)
        lo, hi = dist_mod.shard_range(n_total, rank, world)
        mine = [oracle.ieskf(prm, host.synth_pair(600 + i), oracle.FORM_REDUCED, oracle.NN_KDTREE) for i in range(lo, hi)]
        pipe = dist_mod.PoseGatherPipeline(n_total, rank, world, device="cpu")
        b, buf = pipe.begin_step()
        rec = dist_mod.records_from_results(mine, scan_id_base=lo)
        buf[: rec.nbytes] = torch.from_numpy(rec.view(np.uint8).copy())
        pipe.gather_newest()
        pipe.end_step(b)
        pipe.drain()
        got = pipe.records()  # raises when out of order
        ok = True
        if rank == 0:
            whole = [oracle.ieskf(prm, host.synth_pair(600 + i), oracle.FORM_REDUCED, oracle.NN_KDTREE) for i in range(n_total)]
            want = dist_mod.records_from_results(whole, scan_id_base=0)
            ok = got.tobytes() == want.tobytes() and int(got["iters"].sum()) == 10 * n_total
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_two_ranks_gather_real_results_of_different_shards_world_size_2_gloo():
    """SURVEY.md section 8e / BASELINE.json configs[4] in miniature: a batch of 5 scan pairs sharded 3 + 2 over two ranks,
    every rank's real results (not synthetic records) gathered, bit-equal to the single-process run."""
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_real_records_worker, args=(r, 2, port, 5, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get(0) is True and ret.get(1) is True


def _bench(*argv, env=None):
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], cwd=root, env=e,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


def test_bench_launches_its_own_ranks_when_no_launcher_is_around():
    """`python bench.py --gpus 2` with WORLD_SIZE unset re-executes itself under torch.distributed.run (dry run: gloo,
    synthetic records, no GPU): two ranks, one JSON line, the last thing on stdout."""
    import json

    p = _bench("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--batch", "5")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    d = json.loads(lines[-1])
    assert d == {"dry_run": True, "n_gpus": 2, "steps": 3, "warmup": 1, "records": 10, "ordered": True}


def test_bench_honours_an_external_launcher_and_rejects_a_mismatch():
    p = _bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and b"WORLD_SIZE=1" in p.stderr


def test_bench_says_in_one_line_when_the_node_has_too_few_gpus():
    import torch

    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    p = _bench("--gpus", str(n + 2), "--steps", "1")
    assert p.returncode != 0
    assert f"needs {n + 2} GPUs on this node, found {n}".encode() in p.stderr


def test_records_from_results_layout(pkg):
    dist_mod = importlib.import_module(PKG + ".dist")
    defs = importlib.import_module(PKG + "._ctypes_defs")
    rc = defs.ResultC()
    rc.state[0] = 1.5
    rc.iters, rc.m_surf = 7, 70
    rec = dist_mod.records_from_results([defs.Result(rc)], scan_id_base=40)
    assert rec.dtype.itemsize == 192 and rec["scan_id"][0] == 40 and rec["iters"][0] == 7 and rec["state"][0, 0] == 1.5
