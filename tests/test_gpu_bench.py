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


def test_bench_with_the_rccl_gather_forced_on_one_gpu():
    d = _run({"LINS_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577", "RANK": "0", "WORLD_SIZE": "1",
              "LOCAL_RANK": "0"}, 29577)
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert "all-gather" in d["config"]["parallelism"]
